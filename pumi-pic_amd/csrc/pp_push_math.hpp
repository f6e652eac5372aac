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
// toroidal direction comes from the current position (x0,y0)/hypot -- no atan2.
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
  const double ux = x0 / r0, uy = y0 / r0;
  tx = Rn * (ux * ct - uy * st);
  ty = Rn * (ux * st + uy * ct);
  tz = Zn;
}

}  // namespace ppm
