/*
 * ppo_mesh.c -- ORACLE (test infrastructure, NOT product code).
 *
 * Derives, from (coords, elem2verts), the Omega_h adjacency arrays the hot path reads
 * (call sites: src/pumipic_adjacency.tpp:238-241,394-396,489,497-501;
 *  src/pumipic_adjacency.hpp:568-574,1030-1036; test/gyroScatter.hpp:137,148-152).
 * Omega_h itself is not under /root/reference, so entity NUMBERING of derived sides is this
 * build's own canonical choice (documented in DESIGN.md): a side gets the next id the first
 * time it is met walking elements in id order and local sides in template order, and stores its
 * vertices in the template order of that first element.  Any consistent choice yields identical
 * element ids / intersection points (SURVEY "hard parts: face orientation").
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "ppo.h"
#include "ppo_geom.h"

typedef struct {
  int k[3];
  long occ; /* elem*(dim+1)+local side */
} side_rec;

static int cmp_side(const void* a, const void* b) {
  const side_rec* x = (const side_rec*)a;
  const side_rec* y = (const side_rec*)b;
  for (int i = 0; i < 3; ++i) {
    if (x->k[i] != y->k[i]) return (x->k[i] < y->k[i]) ? -1 : 1;
  }
  return (x->occ < y->occ) ? -1 : (x->occ > y->occ);
}
static int cmp_long(const void* a, const void* b) {
  long x = *(const long*)a, y = *(const long*)b;
  return (x < y) ? -1 : (x > y);
}
static void sort3(int* k, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (k[j] < k[i]) {
        int t = k[i];
        k[i] = k[j];
        k[j] = t;
      }
}

ppo_mesh* ppo_mesh_create(int dim, int nverts, const double* coords, int nelems,
                          const int* elem2verts, const int* class_id) {
  if (dim != 2 && dim != 3) return NULL;
  const int nv = dim + 1; /* verts per elem == sides per elem */
  ppo_mesh* m = (ppo_mesh*)calloc(1, sizeof(ppo_mesh));
  m->dim = dim;
  m->nverts = nverts;
  m->nelems = nelems;
  m->coords = (double*)malloc(sizeof(double) * (size_t)nverts * dim);
  memcpy(m->coords, coords, sizeof(double) * (size_t)nverts * dim);
  m->elem2verts = (int*)malloc(sizeof(int) * (size_t)nelems * nv);
  memcpy(m->elem2verts, elem2verts, sizeof(int) * (size_t)nelems * nv);
  m->class_id = (int*)calloc((size_t)(nelems > 0 ? nelems : 1), sizeof(int));
  if (class_id) memcpy(m->class_id, class_id, sizeof(int) * (size_t)nelems);

  const long nrec = (long)nelems * nv;
  side_rec* rec = (side_rec*)malloc(sizeof(side_rec) * (size_t)(nrec > 0 ? nrec : 1));
  for (int e = 0; e < nelems; ++e)
    for (int ls = 0; ls < nv; ++ls) {
      side_rec* r = &rec[(long)e * nv + ls];
      r->k[2] = -1;
      for (int j = 0; j < dim; ++j) {
        const int lv = (dim == 3) ? PPO_TET_FACE[ls][j] : PPO_TRI_EDGE[ls][j];
        r->k[j] = elem2verts[(long)e * nv + lv];
      }
      sort3(r->k, dim);
      r->occ = (long)e * nv + ls;
    }
  qsort(rec, (size_t)nrec, sizeof(side_rec), cmp_side);
  /* group -> first occurrence */
  long ngroups = 0;
  long* first = (long*)malloc(sizeof(long) * (size_t)(nrec > 0 ? nrec : 1));
  for (long i = 0; i < nrec;) {
    long j = i + 1;
    while (j < nrec && rec[j].k[0] == rec[i].k[0] && rec[j].k[1] == rec[i].k[1] &&
           rec[j].k[2] == rec[i].k[2])
      ++j;
    if (j - i > 2) {
      fprintf(stderr, "ppo_mesh_create: non-manifold side shared by %ld elements\n", j - i);
    }
    first[ngroups++] = rec[i].occ;
    i = j;
  }
  long* first_sorted = (long*)malloc(sizeof(long) * (size_t)(ngroups > 0 ? ngroups : 1));
  memcpy(first_sorted, first, sizeof(long) * (size_t)ngroups);
  qsort(first_sorted, (size_t)ngroups, sizeof(long), cmp_long);
  /* id of a group = rank of its first occurrence; map occ -> id via an occ-indexed table */
  int* occ2side = (int*)malloc(sizeof(int) * (size_t)(nrec > 0 ? nrec : 1));
  for (long i = 0; i < nrec; ++i) occ2side[i] = -1;
  for (long s = 0; s < ngroups; ++s) occ2side[first_sorted[s]] = (int)s;
  m->nsides = (int)ngroups;
  m->elem2sides = (int*)malloc(sizeof(int) * (size_t)(nrec > 0 ? nrec : 1));
  m->side2verts = (int*)malloc(sizeof(int) * (size_t)(ngroups > 0 ? ngroups : 1) * dim);
  m->side2elems_off = (int*)calloc((size_t)ngroups + 1, sizeof(int));
  m->side_exposed = (signed char*)calloc((size_t)(ngroups > 0 ? ngroups : 1), 1);
  {
    long g = 0;
    for (long i = 0; i < nrec;) {
      long j = i + 1;
      while (j < nrec && rec[j].k[0] == rec[i].k[0] && rec[j].k[1] == rec[i].k[1] &&
             rec[j].k[2] == rec[i].k[2])
        ++j;
      const int sid = occ2side[first[g]];
      for (long q = i; q < j; ++q) m->elem2sides[rec[q].occ] = sid;
      m->side2elems_off[sid + 1] = (int)(j - i);
      m->side_exposed[sid] = (j - i == 1);
      const long occ = first[g];
      const int e = (int)(occ / nv), ls = (int)(occ % nv);
      for (int q = 0; q < dim; ++q) {
        const int lv = (dim == 3) ? PPO_TET_FACE[ls][q] : PPO_TRI_EDGE[ls][q];
        m->side2verts[(long)sid * dim + q] = elem2verts[(long)e * nv + lv];
      }
      ++g;
      i = j;
    }
  }
  for (long s = 0; s < ngroups; ++s) m->side2elems_off[s + 1] += m->side2elems_off[s];
  m->side2elems = (int*)malloc(sizeof(int) * (size_t)(nrec > 0 ? nrec : 1));
  {
    int* fill = (int*)calloc((size_t)ngroups + 1, sizeof(int));
    for (int e = 0; e < nelems; ++e) /* ascending element id per side */
      for (int ls = 0; ls < nv; ++ls) {
        const int sid = m->elem2sides[(long)e * nv + ls];
        m->side2elems[m->side2elems_off[sid] + fill[sid]++] = e;
      }
    free(fill);
  }
  /* dual graph */
  m->dual_off = (int*)calloc((size_t)nelems + 1, sizeof(int));
  for (int e = 0; e < nelems; ++e) {
    int c = 0;
    for (int ls = 0; ls < nv; ++ls) c += !m->side_exposed[m->elem2sides[(long)e * nv + ls]];
    m->dual_off[e + 1] = m->dual_off[e] + c;
  }
  m->dual_elems = (int*)malloc(sizeof(int) * (size_t)(m->dual_off[nelems] > 0 ? m->dual_off[nelems] : 1));
  for (int e = 0; e < nelems; ++e) {
    int c = m->dual_off[e];
    for (int ls = 0; ls < nv; ++ls) {
      const int sid = m->elem2sides[(long)e * nv + ls];
      if (m->side_exposed[sid]) continue;
      const int a = m->side2elems[m->side2elems_off[sid]];
      const int b = m->side2elems[m->side2elems_off[sid] + 1];
      m->dual_elems[c++] = (a == e) ? b : a;
    }
  }
  /* vert -> elems (ascending elem id) */
  m->vert2elems_off = (int*)calloc((size_t)nverts + 1, sizeof(int));
  for (long i = 0; i < nrec; ++i) m->vert2elems_off[elem2verts[i] + 1]++;
  for (int v = 0; v < nverts; ++v) m->vert2elems_off[v + 1] += m->vert2elems_off[v];
  m->vert2elems = (int*)malloc(sizeof(int) * (size_t)(nrec > 0 ? nrec : 1));
  {
    int* fill = (int*)calloc((size_t)nverts + 1, sizeof(int));
    for (int e = 0; e < nelems; ++e)
      for (int lv = 0; lv < nv; ++lv) {
        const int v = elem2verts[(long)e * nv + lv];
        m->vert2elems[m->vert2elems_off[v] + fill[v]++] = e;
      }
    free(fill);
  }
  /* measures */
  m->elem_measure = (double*)malloc(sizeof(double) * (size_t)(nelems > 0 ? nelems : 1));
  for (int e = 0; e < nelems; ++e) {
    if (dim == 2) {
      ppo_v2 p[3];
      for (int i = 0; i < 3; ++i) {
        const int v = elem2verts[(long)e * 3 + i];
        p[i].v[0] = coords[(long)v * 2];
        p[i].v[1] = coords[(long)v * 2 + 1];
      }
      m->elem_measure[e] = ppo_tri_area(p);
    } else {
      ppo_v3 p[4];
      for (int i = 0; i < 4; ++i) {
        const int v = elem2verts[(long)e * 4 + i];
        for (int c = 0; c < 3; ++c) p[i].v[c] = coords[(long)v * 3 + c];
      }
      m->elem_measure[e] = ppo_tet_volume(p);
    }
  }
  free(rec);
  free(first);
  free(first_sorted);
  free(occ2side);
  return m;
}

void ppo_mesh_destroy(ppo_mesh* m) {
  if (!m) return;
  free(m->coords);
  free(m->elem2verts);
  free(m->class_id);
  free(m->elem2sides);
  free(m->side2verts);
  free(m->side2elems_off);
  free(m->side2elems);
  free(m->side_exposed);
  free(m->elem_measure);
  free(m->dual_off);
  free(m->dual_elems);
  free(m->vert2elems_off);
  free(m->vert2elems);
  free(m);
}

/* adjacency.tpp:418-428: tol = max(1e-15/min_measure, 1e-8) */
double ppo_compute_tolerance_from_area(const ppo_mesh* m) {
  double min_area = INFINITY; /* Kokkos::Min identity = DBL_MAX; same result for non-empty meshes */
  for (int e = 0; e < m->nelems; ++e)
    if (m->elem_measure[e] < min_area) min_area = m->elem_measure[e];
  const double t = 1e-15 / min_area;
  return (t < 1e-8) ? 1e-8 : t; /* Kokkos::max(a,b) = (a<b)?b:a */
}

/* ---- KAT entry points */
void ppo_kat_barycentric_tet(const double Mflat[12], const double p[3], double parentVol,
                             double bcc_new[4], double bcc_old[4], double bcc_coords[4]) {
  ppo_v3 M[4], pos;
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 3; ++c) M[i].v[c] = Mflat[i * 3 + c];
  for (int c = 0; c < 3; ++c) pos.v[c] = p[c];
  ppo_barycentric_tet(parentVol, M, pos, bcc_new);
  ppo_find_barycentric_tet(M, pos, bcc_old);
  ppo_barycentric_coords_tet(M, pos, bcc_coords, 0);
}
void ppo_kat_barycentric_tri(const double fc[6], const double p[2], double area, double bcc[3]) {
  ppo_v2 f[3], pos = {{p[0], p[1]}};
  for (int i = 0; i < 3; ++i) {
    f[i].v[0] = fc[2 * i];
    f[i].v[1] = fc[2 * i + 1];
  }
  ppo_barycentric_tri(area, f, pos, bcc);
}
int ppo_kat_ray_triangle(const double fv[9], const double o[3], const double d[3], double tol,
                         int flip, int segment, double xpoint[3], double* dproj,
                         double* closeness, double* param) {
  ppo_v3 f[3], orig, dest, xp;
  for (int i = 0; i < 3; ++i)
    for (int c = 0; c < 3; ++c) f[i].v[c] = fv[i * 3 + c];
  for (int c = 0; c < 3; ++c) {
    orig.v[c] = o[c];
    dest.v[c] = d[c];
  }
  int r = segment ? ppo_line_segment_intersects_triangle(f, orig, dest, &xp, tol, flip, dproj,
                                                         closeness, param)
                  : ppo_ray_intersects_triangle(f, orig, dest, &xp, tol, flip, dproj, closeness,
                                                param);
  for (int c = 0; c < 3; ++c) xpoint[c] = xp.v[c];
  return r;
}
int ppo_kat_line_edge_2d(const double ev[4], const double o[2], const double d[2], double tol,
                         int flip, double xpoint[2]) {
  ppo_v2 e[2] = {{{ev[0], ev[1]}}, {{ev[2], ev[3]}}};
  ppo_v2 orig = {{o[0], o[1]}}, dest = {{d[0], d[1]}}, xp;
  int r = ppo_line_edge_2d(e, orig, dest, &xp, tol, flip);
  xpoint[0] = xp.v[0];
  xpoint[1] = xp.v[1];
  return r;
}
int ppo_kat_line_triangle_simple(const double abc[9], const double o[3], const double d[3],
                                 int reverse, double tol, double xpoint[3], double* dproj) {
  ppo_v3 f[3], orig, dest, xp;
  for (int i = 0; i < 3; ++i)
    for (int c = 0; c < 3; ++c) f[i].v[c] = abc[i * 3 + c];
  for (int c = 0; c < 3; ++c) {
    orig.v[c] = o[c];
    dest.v[c] = d[c];
  }
  *dproj = 0;
  int r = ppo_line_triangle_intx_simple(f, orig, dest, &xp, dproj, reverse, tol);
  for (int c = 0; c < 3; ++c) xpoint[c] = xp.v[c];
  return r;
}
int ppo_kat_all_positive(const double* a, int n, double tol) { return ppo_all_positive(a, n, tol); }
int ppo_kat_min3(const double* a) { return ppo_min3(a); }
int ppo_kat_min_index(const double* a, int n) { return ppo_min_index(a, n); }
int ppo_kat_max_index(const double* a, int n) { return ppo_max_index(a, n); }

/* ---- shared deterministic sincos (published fdlibm algorithm: k_sin.c, k_cos.c, e_rem_pio2.c
 * medium-range path).  Plain IEEE ops in a fixed order; the HIP kernel mirrors it exactly. */
static const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11,
                    PIO2_2 = 6.07710050630396597660e-11, PIO2_2T = 2.02226624879595063154e-21,
                    PIO2_3 = 2.02226624871116645580e-21, PIO2_3T = 8.47842766036889956997e-32,
                    INVPIO2 = 6.36619772367581382433e-01;
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

static inline double ksin(double x, double y) {
  const double z = x * x;
  const double v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
static inline double kcos(double x, double y) {
  const double z = x * x;
  const double w = z * z;
  const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  const double hz = 0.5 * z;
  const double ww = 1.0 - hz;
  return ww + (((1.0 - ww) - hz) + (z * r - x * y));
}
static inline int expo(double x) {
  union {
    double d;
    unsigned long long u;
  } c;
  c.d = x;
  return (int)((c.u >> 52) & 0x7ff);
}
void ppo_sincos(double x, double* s, double* c) {
  double y0, y1;
  int n;
  if (!(fabs(x) < 1.0e9)) { /* outside the supported domain (and NaN/Inf): defined result */
    *s = NAN;
    *c = NAN;
    return;
  }
  if (fabs(x) <= 0.78539816339744830962) { /* pi/4 */
    y0 = x;
    y1 = 0.0;
    n = 0;
  } else {
    const double fn = rint(x * INVPIO2);
    double r = x - fn * PIO2_1;
    double w = fn * PIO2_1T;
    const int ex = expo(x);
    y0 = r - w;
    if (ex - expo(y0) > 16) { /* need 2nd iteration */
      double t = r;
      w = fn * PIO2_2;
      r = t - w;
      w = fn * PIO2_2T - ((t - r) - w);
      y0 = r - w;
      if (ex - expo(y0) > 49) { /* 3rd iteration */
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
  switch (n & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

/* Threads of the `#pragma omp` loops (the per-particle loops of the pushes and walks; every
 * iteration touches its own slot only, so the results do not depend on the thread count).
 * The library starts with ONE thread (Kokkos::Serial semantics); bench.py raises it for the
 * all-cores baseline of SURVEY 8(d). */
#ifdef _OPENMP
#include <omp.h>
void ppo_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int ppo_max_threads(void) { return omp_get_num_procs(); }
#else
void ppo_set_threads(int n) { (void)n; }
int ppo_max_threads(void) { return 1; }
#endif
