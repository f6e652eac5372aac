// pumipic_wall.hpp -- wall-interaction geometry a boundary functor calls per particle, restated on
// raw double[3] so the same header serves the library's kernels and USER lambdas run through
// ps::parallel_for.
//
//   TriRegion                                  src/pumipic_adjacency.hpp:812-821
//   closest_point_on_triangle_wnormal          src/pumipic_adjacency.hpp:824-906
//   closest_point_on_triangle                  src/pumipic_adjacency.hpp:910-1009
//
// Arithmetic follows the reference expression by expression (compile with -ffp-contract=off).
// Two reference traits are kept on purpose and stated here: the _wnormal form narrows its
// scalar intermediates to `float` (hpp:840-846,863,873,883,892-894); closest_point_on_triangle
// does not write *reg in its EDGEAB branch (hpp:951-958), so *reg keeps the caller's value there.
#pragma once
#include <hip/hip_runtime.h>

#ifndef PPG_INLINE
#define PPG_INLINE __host__ __device__ inline
#endif

namespace pumipic {

enum TriRegion { VTXA, VTXB, VTXC, EDGEAB, EDGEAC, EDGEBC, TRIFACE, NREGIONS };

namespace wall_detail {
PPG_INLINE double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
PPG_INLINE void sub3(const double a[3], const double b[3], double r[3]) {
  r[0] = a[0] - b[0];
  r[1] = a[1] - b[1];
  r[2] = a[2] - b[2];
}
PPG_INLINE void cross3(const double a[3], const double b[3], double r[3]) {
  r[0] = a[1] * b[2] - a[2] * b[1];
  r[1] = a[2] * b[0] - a[0] * b[2];
  r[2] = a[0] * b[1] - a[1] * b[0];
}
}  // namespace wall_detail

// abc = 3 vertices x 3 coordinates (row-major); q = closest point of the triangle to p
PPG_INLINE void closest_point_on_triangle_wnormal(const double abc[9], const double p[3],
                                                  double q[3], int* reg = nullptr) {
  using namespace wall_detail;
  const double *a = abc, *b = abc + 3, *c = abc + 6;
  double ab[3], ac[3], bc[3], pa[3], pb[3], pc[3], amb[3], amc[3], bmc[3];
  sub3(b, a, ab);
  sub3(c, a, ac);
  sub3(c, b, bc);
  sub3(p, a, pa);
  sub3(p, b, pb);
  sub3(p, c, pc);
  sub3(a, b, amb);
  sub3(a, c, amc);
  sub3(b, c, bmc);
  const float snom = (float)dot3(pa, ab);
  const float sdenom = (float)dot3(pb, amb);
  const float tnom = (float)dot3(pa, ac);
  const float tdenom = (float)dot3(pc, amc);
  if (snom <= 0.0 && tnom <= 0.0) {
    if (reg) *reg = VTXA;
    for (int i = 0; i < 3; ++i) q[i] = a[i];
    return;
  }
  const float unom = (float)dot3(pb, bc);
  const float udenom = (float)dot3(pc, bmc);
  if (sdenom <= 0.0 && unom <= 0.0) {
    if (reg) *reg = VTXB;
    for (int i = 0; i < 3; ++i) q[i] = b[i];
    return;
  }
  if (tdenom <= 0.0 && udenom <= 0.0) {
    if (reg) *reg = VTXC;
    for (int i = 0; i < 3; ++i) q[i] = c[i];
    return;
  }
  double n[3], amp[3], bmp[3], cmp[3], t0[3];
  cross3(ab, ac, n);
  sub3(a, p, amp);
  sub3(b, p, bmp);
  sub3(c, p, cmp);
  cross3(amp, bmp, t0);
  const float vc = (float)dot3(n, t0);
  if (vc <= 0.0 && snom >= 0.0 && sdenom >= 0.0) {
    const double s = snom / (snom + sdenom);  // float division, widened for the product
    for (int i = 0; i < 3; ++i) q[i] = a[i] + s * ab[i];
    if (reg) *reg = EDGEAB;
    return;
  }
  double t1[3];
  cross3(bmp, cmp, t1);
  const float va = (float)dot3(n, t1);
  if (va <= 0.0 && unom >= 0.0 && udenom >= 0.0) {
    const double u = unom / (unom + udenom);
    for (int i = 0; i < 3; ++i) q[i] = b[i] + u * bc[i];
    if (reg) *reg = EDGEBC;
    return;
  }
  double t2[3];
  cross3(cmp, amp, t2);
  const float vb = (float)dot3(n, t2);
  if (vb <= 0.0 && tnom >= 0.0 && tdenom >= 0.0) {
    const double t = tnom / (tnom + tdenom);
    for (int i = 0; i < 3; ++i) q[i] = a[i] + t * ac[i];
    if (reg) *reg = EDGEAC;
    return;
  }
  const float u = va / (va + vb + vc);
  const float v = vb / (va + vb + vc);
  const float w = (float)(1.0 - u - v);
  for (int i = 0; i < 3; ++i) q[i] = (double)u * a[i] + (double)v * b[i] + (double)w * c[i];
  if (reg) *reg = TRIFACE;
}

// Ericson, Real-Time Collision Detection (2005), as restated by the reference
PPG_INLINE void closest_point_on_triangle(const double abc[9], const double ptp[3], double ptq[3],
                                          int* reg = nullptr) {
  using namespace wall_detail;
  const double *pta = abc, *ptb = abc + 3, *ptc = abc + 6;
  double vab[3], vac[3], vap[3];
  sub3(ptb, pta, vab);
  sub3(ptc, pta, vac);
  sub3(ptp, pta, vap);
  const double d1 = dot3(vab, vap);
  const double d2 = dot3(vac, vap);
  if (d1 <= 0 && d2 <= 0) {
    for (int i = 0; i < 3; ++i) ptq[i] = pta[i];
    if (reg) *reg = VTXA;
    return;
  }
  double vbp[3];
  sub3(ptp, ptb, vbp);
  const double d3 = dot3(vab, vbp);
  const double d4 = dot3(vac, vbp);
  if (d3 >= 0 && d4 <= d3) {
    for (int i = 0; i < 3; ++i) ptq[i] = ptb[i];
    if (reg) *reg = VTXB;
    return;
  }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    const double v = d1 / (d1 - d3);
    for (int i = 0; i < 3; ++i) ptq[i] = v * vab[i] + pta[i];
    return;  // *reg untouched (see header)
  }
  double vcp[3];
  sub3(ptp, ptc, vcp);
  const double d5 = dot3(vab, vcp);
  const double d6 = dot3(vac, vcp);
  if (d6 >= 0 && d5 <= d6) {
    for (int i = 0; i < 3; ++i) ptq[i] = ptc[i];
    if (reg) *reg = VTXC;
    return;
  }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    const double w = d2 / (d2 - d6);
    for (int i = 0; i < 3; ++i) ptq[i] = w * vac[i] + pta[i];
    if (reg) *reg = EDGEAC;
    return;
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; ++i) ptq[i] = ptb[i] + w * (ptc[i] - ptb[i]);
    if (reg) *reg = EDGEBC;
    return;
  }
  const double inv = 1.0 / (va + vb + vc);
  const double v = vb * inv;
  const double w = vc * inv;
  for (int i = 0; i < 3; ++i) ptq[i] = pta[i] + v * vab[i] + w * vac[i];
  if (reg) *reg = TRIFACE;
}

}  // namespace pumipic
