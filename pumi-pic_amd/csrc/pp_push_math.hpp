// pp_push_math.hpp -- the per-particle arithmetic of the elliptical push, shared by the plain
// push kernels (pp_push.hip) and the fused push+walk kernel (pp_search.hip).
// Reference: setPosition lambda, test/ellipticalPush.hpp:51-67.  cos/sin go through the
// deterministic ppg::sincos_det so host oracle and device agree bit for bit.
#pragma once
#include "pp_geom.hpp"

namespace ppm {

constexpr double kPi = 3.14159265358979323846;  // M_PI

// rad = phi + deg * (centerFactor/class) * pi/180 ; (x,y) on the particle's ellipse
PPD void elliptical_advance(int cls, float phi, float b, double h, double k, double d, double deg,
                            double& x, double& y, double& rad) {
  const double centerFactor = cls == 1 ? 0.01 : 1.0;
  const double distByClass = centerFactor * (double)1.0 / cls;
  const double degP = deg * distByClass;
  const double a = b * d;
  rad = phi + degP * kPi / 180.0;
  double sn, cs;
  ppg::sincos_det(rad, sn, cs);
  x = a * cs + h;
  y = b * sn + k;
}

// 3-D tokamak restatement: the same advance in the local (R,Z) half-plane, then a
// rigid rotation of that half-plane about the Z axis by the same class-scaled angle.  The
// toroidal direction comes from the current position (x0,y0) -- no atan2:
// x_tgt = (R'/hypot(x0,y0)) * Rot(dphi) (x0,y0).
PPD void toroidal_advance(int cls, float phi, float b, double x0, double y0, double h, double k,
                          double d, double deg, double& tx, double& ty, double& tz, double& rad) {
  const double centerFactor = cls == 1 ? 0.01 : 1.0;
  const double distByClass = centerFactor * (double)1.0 / cls;
  const double degP = deg * distByClass;
  const double a = b * d;
  const double dphi = degP * kPi / 180.0;
  rad = phi + dphi;
  double sn, cs, st, ct;
  ppg::sincos_det(rad, sn, cs);
  ppg::sincos_det(dphi, st, ct);
  const double Rn = a * cs + h;
  const double Zn = b * sn + k;
  const double r0 = sqrt(x0 * x0 + y0 * y0);
  const double sc = Rn / r0;  // radial scale; (x0,y0) rotated by the class angle
  tx = sc * (x0 * ct - y0 * st);
  ty = sc * (x0 * st + y0 * ct);
  tz = Zn;
}

// ---- row-hoisted forms used by the row-tiled kernel: everything that depends only on the parent
// element's class is evaluated once per row; the expression trees (and therefore the bits) are
// the same as in elliptical_advance / toroidal_advance above.
struct ClassTerm {
  double dphi;    // degP * pi / 180
  double st, ct;  // sincos_det(dphi) (3-D only)
};
PPD ClassTerm class_term(int cls, double deg, bool need_trig) {
  ClassTerm t;
  const double centerFactor = cls == 1 ? 0.01 : 1.0;
  const double distByClass = centerFactor * (double)1.0 / cls;
  const double degP = deg * distByClass;
  t.dphi = degP * kPi / 180.0;
  t.st = 0;
  t.ct = 1;
  if (need_trig) ppg::sincos_det(t.dphi, t.st, t.ct);
  return t;
}
PPD void elliptical_point(const ClassTerm& t, float phi, float b, double h, double k, double d,
                          double& x, double& y, double& rad) {
  const double a = b * d;
  rad = phi + t.dphi;
  double sn, cs;
  ppg::sincos_det(rad, sn, cs);
  x = a * cs + h;
  y = b * sn + k;
}
PPD void toroidal_point(const ClassTerm& t, float phi, float b, double x0, double y0, double h,
                        double k, double d, double& tx, double& ty, double& tz, double& rad) {
  const double a = b * d;
  rad = phi + t.dphi;
  double sn, cs;
  ppg::sincos_det(rad, sn, cs);
  const double Rn = a * cs + h;
  const double Zn = b * sn + k;
  const double r0 = sqrt(x0 * x0 + y0 * y0);
  const double sc = Rn / r0;
  tx = sc * (x0 * t.ct - y0 * t.st);
  ty = sc * (x0 * t.st + y0 * t.ct);
  tz = Zn;
}

// velocity update of pushBoris (src/pumipic_push.hpp:28-58; charge = 1, amu = 10 as hard-coded there)
PPD ppg::V3 boris_velocity(ppg::V3 vel, ppg::V3 eField, ppg::V3 bField, double dt) {
  using namespace ppg;
  const double charge = 1, amu = 10;
  const double bFieldMag = norm(bField);
  const double qPrime = charge * 1.60217662e-19 / (amu * 1.6737236e-27) * dt * 0.5;
  const double coeff = 2.0 * qPrime / (1.0 + (qPrime * bFieldMag) * (qPrime * bFieldMag));
  const V3 qpE = mul(eField, qPrime);
  const V3 vMinus = sub(vel, qpE);
  const V3 vPrime = add(vMinus, mul(cross(vMinus, bField), qPrime));
  vel = add(vMinus, mul(cross(vPrime, bField), coeff));
  return add(vel, qpE);
}

}  // namespace ppm
