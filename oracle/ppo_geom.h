/*
 * ppo_geom.h -- ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the small-vector geometry the PUMI-PIC hot path is written in.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
 * under oracle/.  The product (pumi-pic_amd/) never includes, links or calls this.
 *
 * The arithmetic (operand order, association, where the divides are) follows the
 * reference line by line so that sign tests and arg-min decisions are bit-identical
 * to a Kokkos::Serial build compiled without FMA contraction:
 *   src/pumipic_adjacency.tpp:23-69,152-218   (barycentric_tri/tet, ray/segment/edge tests)
 *   src/pumipic_adjacency.hpp:75-94,97-133,136-159,163-183,230-273
 *   src/pumipic_utils.hpp:78-92,125-149,489-507,565-572
 * Third-party arithmetic that is NOT under /root/reference (Omega_h scorec-v10.8.4,
 * call sites adjacency.tpp:29-36,52-56,158-177): cross, inner_product, norm, perp,
 * simplex_down_template, triangle_area_from_basis, tet_volume_from_basis, are_close.
 * Those are restated here from their published definitions; they are pinned by the
 * reference's own KATs (src/unit_tests.hpp:101-177) in tests/test_oracle_kats.py.
 *
 * Build with -ffp-contract=off (see oracle/Makefile).
 */
#ifndef PPO_GEOM_H
#define PPO_GEOM_H
#include <math.h>

typedef struct { double v[2]; } ppo_v2;
typedef struct { double v[3]; } ppo_v3;

/* Omega_h simplex_down_template: tet face -> 3 tet-local verts; tri edge -> 2 verts.
 * Documented in-tree at adjacency.hpp:48-66 and utils.hpp:488,567,642. */
static const int PPO_TET_FACE[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};
static const int PPO_TRI_EDGE[3][2] = {{0, 1}, {1, 2}, {2, 0}};
/* simplex_opposite_template(3,2,f): vertex opposite tet face f (adjacency.hpp:51) */
static const int PPO_TET_OPP[4] = {3, 2, 0, 1};

static inline ppo_v3 ppo_sub3(ppo_v3 a, ppo_v3 b) {
  ppo_v3 c = {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}};
  return c;
}
static inline ppo_v3 ppo_add3(ppo_v3 a, ppo_v3 b) {
  ppo_v3 c = {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}};
  return c;
}
static inline ppo_v3 ppo_scale3(ppo_v3 a, double s) {
  ppo_v3 c = {{a.v[0] * s, a.v[1] * s, a.v[2] * s}};
  return c;
}
static inline ppo_v3 ppo_div3(ppo_v3 a, double s) {
  ppo_v3 c = {{a.v[0] / s, a.v[1] / s, a.v[2] / s}};
  return c;
}
/* Omega_h cross(Vector<3>,Vector<3>) */
static inline ppo_v3 ppo_cross3(ppo_v3 a, ppo_v3 b) {
  ppo_v3 c = {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2],
               a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
  return c;
}
/* Omega_h inner_product: c = a0*b0; c += a1*b1; c += a2*b2 */
static inline double ppo_dot3(ppo_v3 a, ppo_v3 b) {
  double c = a.v[0] * b.v[0];
  c += a.v[1] * b.v[1];
  c += a.v[2] * b.v[2];
  return c;
}
static inline double ppo_norm3(ppo_v3 a) { return sqrt(ppo_dot3(a, a)); }
static inline ppo_v3 ppo_normalize3(ppo_v3 a) { return ppo_div3(a, ppo_norm3(a)); }

static inline ppo_v2 ppo_sub2(ppo_v2 a, ppo_v2 b) {
  ppo_v2 c = {{a.v[0] - b.v[0], a.v[1] - b.v[1]}};
  return c;
}
static inline double ppo_dot2(ppo_v2 a, ppo_v2 b) {
  double c = a.v[0] * b.v[0];
  c += a.v[1] * b.v[1];
  return c;
}
/* Omega_h cross(Vector<2>,Vector<2>) (scalar) and perp */
static inline double ppo_cross2(ppo_v2 a, ppo_v2 b) { return a.v[0] * b.v[1] - a.v[1] * b.v[0]; }
static inline ppo_v2 ppo_perp2(ppo_v2 a) {
  ppo_v2 c = {{-a.v[1], a.v[0]}};
  return c;
}

/* Omega_h measure_elements_real for simplices: basis b_i = p_{i+1}-p_0;
 * triangle_area_from_basis = cross(b0,b1)/2 ; tet_volume_from_basis = (cross(b0,b1).b2)/6 */
static inline double ppo_tri_area(const ppo_v2 p[3]) {
  return ppo_cross2(ppo_sub2(p[1], p[0]), ppo_sub2(p[2], p[0])) / 2.0;
}
static inline double ppo_tet_volume(const ppo_v3 p[4]) {
  ppo_v3 b0 = ppo_sub3(p[1], p[0]), b1 = ppo_sub3(p[2], p[0]), b2 = ppo_sub3(p[3], p[0]);
  return ppo_dot3(ppo_cross3(b0, b1), b2) / 6.0;
}

/* Omega_h are_close(a,b,tol,floor) = rel_diff_with_floor(a,b,floor) <= tol */
static inline int ppo_are_close(double a, double b, double tol, double floor_) {
  double am = fabs(a), bm = fabs(b);
  if (am <= floor_ && bm <= floor_) return 1; /* rel diff 0.0 <= tol */
  double mx = (bm > am) ? bm : am;
  return (fabs(b - a) / mx) <= tol;
}

#define PPO_EPSILON 1e-10 /* src/pumipic_constants.hpp:6 */

/* utils.hpp:78-86 */
static inline int ppo_all_positive(const double* a, int n, double tol) {
  int isPos = 1;
  for (int i = 0; i < n; ++i) {
    const int gtez = ppo_are_close(a[i], 0.0, tol, tol) || a[i] > 0;
    isPos = isPos && gtez;
  }
  return isPos;
}
/* utils.hpp:88-92 */
static inline int ppo_min3(const double* a) {
  int idx = (a[0] < a[1]) ? 0 : 1;
  idx = (a[idx] < a[2]) ? idx : 2;
  return idx;
}
/* utils.hpp:125-136 */
static inline int ppo_min_index(const double* a, int n) {
  int ind = 0;
  double mn = a[0];
  for (int i = 0; i < n - 1; ++i)
    if (mn > a[i + 1]) {
      mn = a[i + 1];
      ind = i + 1;
    }
  return ind;
}
/* utils.hpp:138-149 (beg = 0) */
static inline int ppo_max_index(const double* a, int n) {
  int ind = 0;
  double mx = a[0];
  for (int i = 0; i < n - 1; ++i)
    if (mx < a[i + 1]) {
      mx = a[i + 1];
      ind = i + 1;
    }
  return ind;
}

/* adjacency.tpp:23-39 -- bcc[i] is the area coordinate of edge i (edge-major order) */
static inline void ppo_barycentric_tri(double parentArea, const ppo_v2 fc[3], ppo_v2 pos,
                                       double bcc[3]) {
  for (int i = 0; i < 3; ++i) {
    const ppo_v2 k = fc[PPO_TRI_EDGE[i][0]];
    const ppo_v2 l = fc[PPO_TRI_EDGE[i][1]];
    const double area = ppo_cross2(ppo_sub2(l, k), ppo_sub2(pos, k)) / 2.0;
    bcc[i] = area / parentArea;
  }
}
/* adjacency.hpp:75-94 with vertex_major shift */
static inline void ppo_barycentric_tri_vm(double parentArea, const ppo_v2 fc[3], ppo_v2 pos,
                                          double bcc[3], int vertex_major) {
  const int vshift = vertex_major ? 1 : 0;
  for (int i = 0; i < 3; ++i) {
    const ppo_v2 k = fc[PPO_TRI_EDGE[(i + vshift) % 3][0]];
    const ppo_v2 l = fc[PPO_TRI_EDGE[(i + vshift) % 3][1]];
    const double area = ppo_cross2(ppo_sub2(l, k), ppo_sub2(pos, k)) / 2.0;
    bcc[i] = area / parentArea;
  }
}

/* common numerators: vals[f] = (p-a).((c-a)x(b-a)) over template faces (tpp:51-57) */
static inline void ppo_tet_face_vals(const ppo_v3 M[4], ppo_v3 pos, double vals[4]) {
  for (int f = 0; f < 4; ++f) {
    const ppo_v3 a = M[PPO_TET_FACE[f][0]], b = M[PPO_TET_FACE[f][1]], c = M[PPO_TET_FACE[f][2]];
    const ppo_v3 vab = ppo_sub3(b, a), vac = ppo_sub3(c, a), vap = ppo_sub3(pos, a);
    vals[f] = ppo_dot3(vap, ppo_cross3(vac, vab));
  }
}
/* adjacency.tpp:41-69: bcc = vals / parentVol (NOTE: sums to 6 when fed the true volume,
 * SURVEY F7 -- replicated, only signs and arg-min are consumed) */
static inline int ppo_barycentric_tet(double parentVol, const ppo_v3 M[4], ppo_v3 pos,
                                      double bcc[4]) {
  double vals[4];
  for (int i = 0; i < 4; ++i) bcc[i] = -1;
  ppo_tet_face_vals(M, pos, vals);
  double inv_vol = 0.0;
  if (parentVol > 0)
    inv_vol = 1.0 / parentVol;
  else
    return 0;
  for (int i = 0; i < 4; ++i) bcc[i] = inv_vol * vals[i];
  return 1;
}
/* adjacency.hpp:97-133: divides by vol6 computed from face 0 and its opposite vertex */
static inline int ppo_find_barycentric_tet(const ppo_v3 M[4], ppo_v3 pos, double bcc[4]) {
  double vals[4];
  for (int i = 0; i < 4; ++i) bcc[i] = -1;
  const double tol = 1.0e-20;
  ppo_tet_face_vals(M, pos, vals);
  const ppo_v3 a = M[PPO_TET_FACE[0][0]], b = M[PPO_TET_FACE[0][1]], c = M[PPO_TET_FACE[0][2]];
  const ppo_v3 cross_ac_ab = ppo_cross3(ppo_sub3(c, a), ppo_sub3(b, a));
  const double vol6 = ppo_dot3(ppo_sub3(M[PPO_TET_OPP[0]], M[0]), cross_ac_ab);
  double inv_vol = 0.0;
  if (vol6 > tol)
    inv_vol = 1.0 / vol6;
  else
    return 0;
  for (int i = 0; i < 4; ++i) bcc[i] = inv_vol * vals[i];
  return 1;
}
/* adjacency.hpp:136-159: vals scaled by 1/6 first, then 1/vol * vals */
static inline int ppo_barycentric_coords_tet(const ppo_v3 M[4], ppo_v3 pos, double bcc[4],
                                             double tol) {
  double vals[4];
  for (int f = 0; f < 4; ++f) {
    const ppo_v3 a = M[PPO_TET_FACE[f][0]], b = M[PPO_TET_FACE[f][1]], c = M[PPO_TET_FACE[f][2]];
    const ppo_v3 vab = ppo_sub3(b, a), vac = ppo_sub3(c, a), vap = ppo_sub3(pos, a);
    vals[f] = 1.0 / 6.0 * ppo_dot3(vap, ppo_cross3(vac, vab));
    bcc[f] = 0;
  }
  const double vol = ppo_tet_volume(M);
  if (vol < tol) return 0;
  for (int f = 0; f < 4; ++f) bcc[f] = 1.0 / vol * vals[f];
  return 1;
}

/* adjacency.hpp:163-183 */
static inline int ppo_find_barycentric_tri_simple(const ppo_v3 abc[3], ppo_v3 xpoint,
                                                  double bc[3]) {
  const ppo_v3 a = abc[0], b = abc[1], c = abc[2];
  const ppo_v3 cross = ppo_scale3(ppo_cross3(ppo_sub3(b, a), ppo_sub3(c, a)), 1 / 2.0);
  const ppo_v3 nrm = ppo_normalize3(cross);
  const double area = ppo_dot3(nrm, cross);
  if (fabs(area) < 1e-20) return 0;
  const double fac = 1 / (area * 2.0);
  bc[0] = fac * ppo_dot3(nrm, ppo_cross3(ppo_sub3(b, a), ppo_sub3(xpoint, a)));
  bc[1] = fac * ppo_dot3(nrm, ppo_cross3(ppo_sub3(c, b), ppo_sub3(xpoint, b)));
  bc[2] = fac * ppo_dot3(nrm, ppo_cross3(ppo_sub3(xpoint, a), ppo_sub3(c, a)));
  return 1;
}

/* adjacency.hpp:230-273 (legacy intersection used by the 3-D legacy search) */
static inline int ppo_line_triangle_intx_simple(const ppo_v3 abc[3], ppo_v3 origin, ppo_v3 dest,
                                                ppo_v3* xpoint, double* dproj, int reverse,
                                                double tol) {
  for (int i = 0; i < 3; ++i) xpoint->v[i] = 0;
  int found = 0;
  const ppo_v3 line = ppo_sub3(dest, origin);
  const ppo_v3 edge0 = ppo_sub3(abc[1], abc[0]);
  const ppo_v3 edge1 = ppo_sub3(abc[2], abc[0]);
  ppo_v3 normv = ppo_cross3(edge0, edge1);
  if (reverse) normv = ppo_scale3(normv, -1);
  const ppo_v3 snorm_unit = ppo_normalize3(normv);
  const double dist2plane = ppo_dot3(ppo_sub3(abc[0], origin), snorm_unit);
  const ppo_v3 plane2dest = ppo_sub3(dest, abc[0]);
  const double proj_end = ppo_dot3(snorm_unit, plane2dest);
  if (dist2plane >= -tol && proj_end >= -tol) {
    *dproj = ppo_dot3(line, snorm_unit);
    const double par_t = (*dproj > 0) ? dist2plane / *dproj : 0;
    *xpoint = ppo_add3(origin, ppo_scale3(line, par_t));
    if (*dproj > 0) {
      double bcc[3];
      const int res = ppo_find_barycentric_tri_simple(abc, *xpoint, bcc);
      if (res && bcc[0] >= 0 && bcc[0] <= 1 && bcc[1] >= 0 && bcc[1] <= 1 && bcc[2] >= 0 &&
          bcc[2] <= 1)
        found = 1;
    }
  }
  return found;
}

/* adjacency.tpp:152-178 -- Moller-Trumbore on the unit direction; flip selects which
 * stored face vertices span edge1/edge2 so the normal points out of the current tet. */
static inline int ppo_ray_intersects_triangle(const ppo_v3 fv[3], ppo_v3 orig, ppo_v3 dest,
                                              ppo_v3* xpoint, double tol, int flip,
                                              double* dproj, double* closeness, double* param) {
  const int vtx1 = 2 - flip;
  const int vtx2 = flip + 1;
  const ppo_v3 edge1 = ppo_sub3(fv[vtx1], fv[0]);
  const ppo_v3 edge2 = ppo_sub3(fv[vtx2], fv[0]);
  const ppo_v3 displacement = ppo_sub3(dest, orig);
  const double seg_length = ppo_norm3(displacement);
  const ppo_v3 dir = ppo_div3(displacement, seg_length);
  const ppo_v3 faceNorm = ppo_cross3(edge2, edge1);
  const ppo_v3 pvec = ppo_cross3(dir, edge2);
  *dproj = ppo_dot3(dir, faceNorm);
  const double invdet = 1.0 / *dproj;
  const ppo_v3 tvec = ppo_sub3(orig, fv[0]);
  const double u = invdet * ppo_dot3(tvec, pvec);
  const ppo_v3 qvec = ppo_cross3(tvec, edge1);
  const double v = invdet * ppo_dot3(dir, qvec);
  const double t = invdet * ppo_dot3(edge2, qvec);
  *param = t / seg_length;
  *xpoint = ppo_add3(orig, ppo_scale3(dir, t));
  /* Kokkos::max/min/fabs == fmax/fmin/fabs for non-NaN; with NaN Kokkos::max(a,b)=(a<b)?b:a */
#define PPO_KMAX(a, b) (((a) < (b)) ? (b) : (a))
#define PPO_KMIN(a, b) (((b) < (a)) ? (b) : (a))
  {
    const double m1 = PPO_KMIN(fabs(u), fabs(1 - u));
    const double m2 = PPO_KMIN(fabs(v), fabs(1 - v));
    const double m3 = PPO_KMIN(fabs(u + v), fabs(1 - u - v));
    const double mm = PPO_KMAX(m1, m2);
    *closeness = PPO_KMAX(mm, m3);
  }
  return (*dproj >= tol) && (t >= -tol) && (u >= -tol) && (v >= -tol) && (u + v <= 1.0 + 2 * tol);
}
/* adjacency.tpp:192-201 */
static inline int ppo_line_segment_intersects_triangle(const ppo_v3 fv[3], ppo_v3 orig,
                                                       ppo_v3 dest, ppo_v3* xpoint, double tol,
                                                       int flip, double* dproj,
                                                       double* closeness, double* param) {
  const int hit =
      ppo_ray_intersects_triangle(fv, orig, dest, xpoint, tol, flip, dproj, closeness, param);
  return hit && *param <= 1 + tol;
}
/* adjacency.tpp:204-218 */
static inline int ppo_line_edge_2d(const ppo_v2 ev[2], ppo_v2 orig, ppo_v2 dest, ppo_v2* xpoint,
                                   double tol, int flip) {
  const int vtx1 = flip;
  const int vtx2 = !flip;
  const ppo_v2 path = ppo_sub2(dest, orig);
  const ppo_v2 edge = ppo_sub2(ev[vtx2], ev[vtx1]);
  const ppo_v2 norm = ppo_perp2(edge);
  const ppo_v2 normp = ppo_perp2(path);
  const double det = -ppo_dot2(norm, path);
  const double s = ppo_dot2(normp, ppo_sub2(orig, ev[vtx1]));
  const double t = ppo_dot2(norm, ppo_sub2(orig, ev[vtx1]));
  xpoint->v[0] = orig.v[0] + (t / det) * path.v[0];
  xpoint->v[1] = orig.v[1] + (t / det) * path.v[1];
  return det >= tol && s >= -tol && s <= det + tol && t >= -tol && t <= det + tol;
}

/* utils.hpp:489-507 */
static inline int ppo_face_map(int i) {
  static const int fmap[8] = {2, 1, 1, 3, 2, 3, 0, 3};
  return fmap[i];
}
static inline int ppo_is_edge_flipped(int ei, const int ev2v[2], const int facev2v[3]) {
  (void)ei;
  const int index = (ev2v[0] == facev2v[0]) ? 1 : (ev2v[0] == facev2v[1]) ? 2 : 0;
  return ev2v[1] != facev2v[index];
}
static inline int ppo_is_face_flipped(int fi, const int fv2v[3], const int tetv2v[4]) {
  const int matInd1 = ppo_face_map(fi * 2);
  const int matInd2 = ppo_face_map(fi * 2 + 1);
  const int index = (fv2v[0] == tetv2v[matInd1]) ? 1 : (fv2v[1] == tetv2v[matInd1]) ? 2 : 0;
  return tetv2v[matInd2] != fv2v[index];
}

#endif
