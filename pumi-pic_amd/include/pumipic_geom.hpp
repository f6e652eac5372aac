// pumipic_geom.hpp (csrc/pp_geom.hpp forwards here) -- device geometry of the particle walk (gfx950).
//
// Arithmetic order follows the reference so every sign test / arg-min is bit-identical to a
// Kokkos::Serial run built without FMA contraction (this library is compiled with
// -ffp-contract=off):
//   barycentric_tri / barycentric_tet        src/pumipic_adjacency.tpp:23-69
//   find_barycentric_tet                     src/pumipic_adjacency.hpp:97-133
//   barycentric_coords_tet                   src/pumipic_adjacency.hpp:136-159
//   ray_intersects_triangle / line_edge_2d   src/pumipic_adjacency.tpp:152-218
//   line_triangle_intx_simple                src/pumipic_adjacency.hpp:163-183,230-273
//   all_positive / min3 / min_index / max_index  src/pumipic_utils.hpp:78-92,125-149
//   isFaceFlipped / getFaceMap               src/pumipic_utils.hpp:489-507
// Omega_h primitives (cross, inner_product, norm, are_close, simplex templates) are restated
// from their published definitions (Omega_h scorec-v10.8.4; not under the reference tree).
#pragma once
#include <hip/hip_runtime.h>

#define PPD __host__ __device__ __forceinline__

namespace ppg {

struct V2 {
  double x, y;
};
struct V3 {
  double x, y, z;
};

PPD V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PPD V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PPD V3 mul(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
PPD V3 divs(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
PPD V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PPD double dot(V3 a, V3 b) {
  double c = a.x * b.x;
  c += a.y * b.y;
  c += a.z * b.z;
  return c;
}
PPD double norm(V3 a) { return sqrt(dot(a, a)); }
PPD V3 normalize(V3 a) { return divs(a, norm(a)); }
PPD V2 sub(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
PPD double dot(V2 a, V2 b) {
  double c = a.x * b.x;
  c += a.y * b.y;
  return c;
}
PPD double cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
PPD V2 perp(V2 a) { return {-a.y, a.x}; }

// tet face f -> local verts ; tri edge e -> local verts (Omega_h simplex_down_template)
PPD int tet_face_vert(int f, int j) {
  // {0,2,1},{0,1,3},{1,2,3},{2,0,3} packed 2 bits each
  const unsigned packed = (0u | (2u << 2) | (1u << 4)) | ((0u | (1u << 2) | (3u << 4)) << 6) |
                          ((1u | (2u << 2) | (3u << 4)) << 12) |
                          ((2u | (0u << 2) | (3u << 4)) << 18);
  return (int)((packed >> (6 * f + 2 * j)) & 3u);
}
PPD int tri_edge_vert(int e, int j) { return (e + j) % 3; }  // {0,1},{1,2},{2,0}
PPD int face_map(int i) {                                      // utils.hpp:489-493
  const unsigned packed = 2u | (1u << 2) | (1u << 4) | (3u << 6) | (2u << 8) | (3u << 10) |
                          (0u << 12) | (3u << 14);
  return (int)((packed >> (2 * i)) & 3u);
}

constexpr double kEpsilon = 1e-10;  // src/pumipic_constants.hpp:6

// Omega_h are_close(a, 0, tol, tol) || a > 0   (utils.hpp:82)
// are_close's relative difference for b == 0 is |a|/|a|: exactly 1.0 for every finite non-zero a
// and NaN for +-inf / NaN, so the quotient is never formed: (|a|/|a| <= tol) == (finite && 1 <= tol)
PPD bool gtez(double a, double tol) {
  const double am = fabs(a);
  // (non-short-circuit & and |: the operands have no side effects, and on the device a short-circuit chain is a chain
  // of exec-mask branches around three instructions each)
  const bool close = (am <= tol) | ((am < __builtin_inf()) & (1.0 <= tol));
  return close | (a > 0);
}
PPD bool all_positive3(const double* a, double tol) {
  bool p = true;
  for (int i = 0; i < 3; ++i) p = p & gtez(a[i], tol);
  return p;
}
PPD bool all_positive4(const double* a, double tol) {
  bool p = true;
  for (int i = 0; i < 4; ++i) p = p & gtez(a[i], tol);
  return p;
}
PPD int min3(const double* a) {
  int idx = (a[0] < a[1]) ? 0 : 1;
  idx = (a[idx] < a[2]) ? idx : 2;
  return idx;
}
PPD int min_index4(const double* a) {
  int ind = 0;
  double mn = a[0];
  for (int i = 0; i < 3; ++i)
    if (mn > a[i + 1]) {
      mn = a[i + 1];
      ind = i + 1;
    }
  return ind;
}
PPD int max_index4(const double* a) {
  int ind = 0;
  double mx = a[0];
  for (int i = 0; i < 3; ++i)
    if (mx < a[i + 1]) {
      mx = a[i + 1];
      ind = i + 1;
    }
  return ind;
}

// measure_elements_real restated (Omega_h triangle_area_from_basis / tet_volume_from_basis)
PPD double tri_area(const V2 p[3]) { return cross(sub(p[1], p[0]), sub(p[2], p[0])) / 2.0; }
PPD double tet_volume(const V3 p[4]) {
  return dot(cross(sub(p[1], p[0]), sub(p[2], p[0])), sub(p[3], p[0])) / 6.0;
}

PPD void barycentric_tri(double parentArea, const V2 fc[3], V2 pos, double bcc[3]) {
  for (int i = 0; i < 3; ++i) {
    const V2 k = fc[tri_edge_vert(i, 0)];
    const V2 l = fc[tri_edge_vert(i, 1)];
    const double area = cross(sub(l, k), sub(pos, k)) / 2.0;
    bcc[i] = area / parentArea;
  }
}
PPD void tet_face_vals(const V3 M[4], V3 pos, double vals[4]) {
  for (int f = 0; f < 4; ++f) {
    const V3 a = M[tet_face_vert(f, 0)], b = M[tet_face_vert(f, 1)], c = M[tet_face_vert(f, 2)];
    vals[f] = dot(sub(pos, a), cross(sub(c, a), sub(b, a)));
  }
}
PPD bool barycentric_tet(double parentVol, const V3 M[4], V3 pos, double bcc[4]) {
  double vals[4];
  for (int i = 0; i < 4; ++i) bcc[i] = -1;
  tet_face_vals(M, pos, vals);
  double inv_vol = 0.0;
  if (parentVol > 0)
    inv_vol = 1.0 / parentVol;
  else
    return false;
  for (int i = 0; i < 4; ++i) bcc[i] = inv_vol * vals[i];
  return true;
}
PPD bool find_barycentric_tet(const V3 M[4], V3 pos, double bcc[4]) {
  double vals[4];
  for (int i = 0; i < 4; ++i) bcc[i] = -1;
  tet_face_vals(M, pos, vals);
  const V3 a = M[0], b = M[2], c = M[1];  // face 0 = {0,2,1}
  const double vol6 = dot(sub(M[3], M[0]), cross(sub(c, a), sub(b, a)));
  double inv_vol = 0.0;
  if (vol6 > 1.0e-20)
    inv_vol = 1.0 / vol6;
  else
    return false;
  for (int i = 0; i < 4; ++i) bcc[i] = inv_vol * vals[i];
  return true;
}

// adjacency.hpp:136-159: vals scaled by 1/6 first, then 1/vol * vals (vol from the element basis)
PPD bool barycentric_coords_tet(const V3 M[4], V3 pos, double bcc[4], double tol) {
  double vals[4];
  for (int f = 0; f < 4; ++f) {
    const V3 a = M[tet_face_vert(f, 0)], b = M[tet_face_vert(f, 1)], c = M[tet_face_vert(f, 2)];
    vals[f] = 1.0 / 6.0 * dot(sub(pos, a), cross(sub(c, a), sub(b, a)));
    bcc[f] = 0;
  }
  const double vol = tet_volume(M);
  if (vol < tol) return false;
  for (int f = 0; f < 4; ++f) bcc[f] = 1.0 / vol * vals[f];
  return true;
}

#define PPG_KMAX(a, b) (((a) < (b)) ? (b) : (a))
#define PPG_KMIN(a, b) (((b) < (a)) ? (b) : (a))

PPD bool ray_intersects_triangle(const V3 fv[3], V3 orig, V3 dest, V3& xpoint, double tol,
                                 int flip, double& dproj, double& closeness, double& param) {
  const V3 edge1 = sub(fv[2 - flip], fv[0]);
  const V3 edge2 = sub(fv[flip + 1], fv[0]);
  const V3 displacement = sub(dest, orig);
  const double seg_length = norm(displacement);
  const V3 dir = divs(displacement, seg_length);
  const V3 faceNorm = cross(edge2, edge1);
  const V3 pvec = cross(dir, edge2);
  dproj = dot(dir, faceNorm);
  const double invdet = 1.0 / dproj;
  const V3 tvec = sub(orig, fv[0]);
  const double u = invdet * dot(tvec, pvec);
  const V3 qvec = cross(tvec, edge1);
  const double v = invdet * dot(dir, qvec);
  const double t = invdet * dot(edge2, qvec);
  param = t / seg_length;
  xpoint = add(orig, mul(dir, t));
  const double m1 = PPG_KMIN(fabs(u), fabs(1 - u));
  const double m2 = PPG_KMIN(fabs(v), fabs(1 - v));
  const double m3 = PPG_KMIN(fabs(u + v), fabs(1 - u - v));
  const double mm = PPG_KMAX(m1, m2);
  closeness = PPG_KMAX(mm, m3);
  return (dproj >= tol) && (t >= -tol) && (u >= -tol) && (v >= -tol) && (u + v <= 1.0 + 2 * tol);
}

PPD bool line_edge_2d(const V2 ev[2], V2 orig, V2 dest, V2& xpoint, double tol, int flip) {
  const V2 a = ev[flip], b = ev[!flip];
  const V2 path = sub(dest, orig);
  const V2 edge = sub(b, a);
  const V2 nrm = perp(edge);
  const V2 nrmp = perp(path);
  const double det = -dot(nrm, path);
  const double s = dot(nrmp, sub(orig, a));
  const double t = dot(nrm, sub(orig, a));
  xpoint.x = orig.x + (t / det) * path.x;
  xpoint.y = orig.y + (t / det) * path.y;
  return det >= tol && s >= -tol && s <= det + tol && t >= -tol && t <= det + tol;
}

PPD bool find_barycentric_tri_simple(const V3 abc[3], V3 xpoint, double bc[3]) {
  const V3 a = abc[0], b = abc[1], c = abc[2];
  const V3 cr = mul(cross(sub(b, a), sub(c, a)), 1 / 2.0);
  const V3 nrm = normalize(cr);
  const double area = dot(nrm, cr);
  if (fabs(area) < 1e-20) return false;
  const double fac = 1 / (area * 2.0);
  bc[0] = fac * dot(nrm, cross(sub(b, a), sub(xpoint, a)));
  bc[1] = fac * dot(nrm, cross(sub(c, b), sub(xpoint, b)));
  bc[2] = fac * dot(nrm, cross(sub(xpoint, a), sub(c, a)));
  return true;
}
PPD bool line_triangle_intx_simple(const V3 abc[3], V3 origin, V3 dest, V3& xpoint,
                                   double& dproj, bool reverse, double tol) {
  xpoint = {0, 0, 0};
  bool found = false;
  const V3 line = sub(dest, origin);
  V3 normv = cross(sub(abc[1], abc[0]), sub(abc[2], abc[0]));
  if (reverse) normv = mul(normv, -1);
  const V3 snorm_unit = normalize(normv);
  const double dist2plane = dot(sub(abc[0], origin), snorm_unit);
  const double proj_end = dot(snorm_unit, sub(dest, abc[0]));
  if (dist2plane >= -tol && proj_end >= -tol) {
    dproj = dot(line, snorm_unit);
    const double par_t = (dproj > 0) ? dist2plane / dproj : 0;
    xpoint = add(origin, mul(line, par_t));
    if (dproj > 0) {
      double bcc[3];
      const bool res = find_barycentric_tri_simple(abc, xpoint, bcc);
      if (res && bcc[0] >= 0 && bcc[0] <= 1 && bcc[1] >= 0 && bcc[1] <= 1 && bcc[2] >= 0 &&
          bcc[2] <= 1)
        found = true;
    }
  }
  return found;
}

PPD bool is_edge_flipped(const int ev2v[2], const int facev2v[3]) {
  const int index = (ev2v[0] == facev2v[0]) ? 1 : (ev2v[0] == facev2v[1]) ? 2 : 0;
  return ev2v[1] != facev2v[index];
}
PPD bool is_face_flipped(int fi, const int fv2v[3], const int tetv2v[4]) {
  const int m1 = face_map(fi * 2), m2 = face_map(fi * 2 + 1);
  const int index = (fv2v[0] == tetv2v[m1]) ? 1 : (fv2v[1] == tetv2v[m1]) ? 2 : 0;
  return tetv2v[m2] != fv2v[index];
}

// ---- deterministic sincos: Cody-Waite 3-term pi/2 reduction + degree-13/14 minimax kernels
// (published fdlibm algorithm).  Evaluated with plain IEEE mul/add in a fixed order; the oracle
// evaluates the same expression tree, so positions agree bit for bit (<= 1 ulp from libm).
PPD double ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x;
  const double v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
PPD double kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double w = z * z;
  const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  const double hz = 0.5 * z;
  const double ww = 1.0 - hz;
  return ww + (((1.0 - ww) - hz) + (z * r - x * y));
}
PPD int dexp(double x) {
  union {
    double d;
    unsigned long long u;
  } c;
  c.d = x;
  return (int)((c.u >> 52) & 0x7ff);
}
PPD void sincos_det(double x, double& s, double& c) {
  const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11,
               PIO2_2 = 6.07710050630396597660e-11, PIO2_2T = 2.02226624879595063154e-21,
               PIO2_3 = 2.02226624871116645580e-21, PIO2_3T = 8.47842766036889956997e-32,
               INVPIO2 = 6.36619772367581382433e-01;
  if (!(fabs(x) < 1.0e9)) {
    s = c = __builtin_nan("");
    return;
  }
  double y0, y1;
  int n;
  if (fabs(x) <= 0.78539816339744830962) {
    y0 = x;
    y1 = 0.0;
    n = 0;
  } else {
    const double fn = rint(x * INVPIO2);
    double r = x - fn * PIO2_1;
    double w = fn * PIO2_1T;
    const int ex = dexp(x);
    y0 = r - w;
    if (ex - dexp(y0) > 16) {
      double t = r;
      w = fn * PIO2_2;
      r = t - w;
      w = fn * PIO2_2T - ((t - r) - w);
      y0 = r - w;
      if (ex - dexp(y0) > 49) {
        t = r;
        w = fn * PIO2_3;
        r = t - w;
        w = fn * PIO2_3T - ((t - r) - w);
        y0 = r - w;
      }
    }
    y1 = (r - y0) - w;
    n = (int)((long long)fn & 3);
  }
  const double sn = ksin(y0, y1), cs = kcos(y0, y1);
  // quadrant n: (s, c) = (sn, cs), (cs, -sn), (-sn, -cs), (-cs, sn) -- as selects (a four-way switch is four
  // exec-mask branches per call on the device)
  const bool odd = (n & 1) != 0;
  const double ss = odd ? cs : sn, cc = odd ? sn : cs;
  s = (n & 2) ? -ss : ss;
  c = ((n + 1) & 2) ? -cc : cc;
}

}  // namespace ppg
