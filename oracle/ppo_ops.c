/*
 * ppo_ops.c -- ORACLE (test infrastructure, NOT product code).
 *
 * Pushes, gyro scatter (+ ring map build), density, gather helper and the ps_combo160 pseudo
 * push, restated from:
 *   test/ellipticalPush.hpp:10-70          src/pumipic_push.hpp:17-75
 *   test/pseudoPushAndSearch.cpp:87-119,340-374
 *   test/gyroScatter.hpp:28-229            test/pseudoXGCm.cpp:102-114
 *   performance_tests/ps_combo160.cpp:152-178
 *   src/pumipic_adjacency.hpp:772-790
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include "ppo.h"
#include "ppo_geom.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static void* xcalloc(size_t n, size_t s) {
  void* p = calloc(n ? n : 1, s ? s : 1);
  if (!p) {
    fprintf(stderr, "ppo: out of memory\n");
    exit(EXIT_FAILURE);
  }
  return p;
}

#define MD(ps, m) ((double*)(ps)->data[m])
#define MF(ps, m) ((float*)(ps)->data[m])
#define MI(ps, m) ((int*)(ps)->data[m])
#define ML(ps, m) ((long*)(ps)->data[m])

typedef struct {
  int cap;
  int* elem;
  unsigned char* mask;
} slots;
static slots get_slots(const ppo_ps* ps) {
  slots s;
  s.cap = ps->capacity;
  s.elem = (int*)xcalloc((size_t)s.cap, sizeof(int));
  s.mask = (unsigned char*)xcalloc((size_t)s.cap, 1);
  ppo_ps_slot_info(ps, s.elem, s.mask);
  if (ps->num_ptcls == 0) /* parallel_for is a no-op then */
    for (int i = 0; i < s.cap; ++i) {
      s.elem[i] = -1;
      s.mask[i] = 0;
    }
  return s;
}
static void free_slots(slots* s) {
  free(s->elem);
  free(s->mask);
}

/* test/ellipticalPush.hpp:10-33 */
void ppo_elliptical_setup(ppo_ps* ps, int m_x, int m_b, int m_phi, double h, double k, double d) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  for (int pid = 0; pid < s.cap; ++pid) {
    if (!s.mask[pid]) continue;
    const double w = MD(ps, m_x)[pid];
    const double z = MD(ps, m_x)[A + pid];
    const double phi = atan2(d * (z - k), w - h);
    const double b = (z - k) / sin(phi);
    MF(ps, m_phi)[pid] = (float)phi;
    MF(ps, m_b)[pid] = (float)b;
  }
  free_slots(&s);
}

static void trig(int mode, double x, double* sn, double* cs) {
  if (mode == 0) {
    *sn = sin(x);
    *cs = cos(x);
  } else {
    ppo_sincos(x, sn, cs);
  }
}

/* test/ellipticalPush.hpp:36-70 */
void ppo_elliptical_push(ppo_ps* ps, const ppo_mesh* mesh, int m_xtgt, int m_b, int m_phi,
                         double h, double k, double d, double deg, int trigmode) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
#pragma omp parallel for schedule(static)
  for (int pid = 0; pid < s.cap; ++pid) {
    if (!s.mask[pid]) continue;
    const int e = s.elem[pid];
    const double centerFactor = mesh->class_id[e] == 1 ? 0.01 : 1.0;
    const double distByClass = centerFactor * (double)1.0 / mesh->class_id[e];
    const double degP = deg * distByClass;
    const float phi = MF(ps, m_phi)[pid];
    const float b = MF(ps, m_b)[pid];
    const double a = b * d;
    const double rad = phi + degP * M_PI / 180.0;
    double sn, cs;
    trig(trigmode, rad, &sn, &cs);
    const double x = a * cs + h;
    const double y = b * sn + k;
    MD(ps, m_xtgt)[pid] = x;
    MD(ps, m_xtgt)[A + pid] = y;
    MF(ps, m_phi)[pid] = (float)rad;
  }
  free_slots(&s);
}

/* 3-D tokamak restatement (SURVEY 8(d) "3-D north-star variant"): the same elliptical advance
 * in the particle's local (R,Z) half-plane, plus a rigid rotation of that half-plane about the
 * Z axis by the same class-scaled angle.  The toroidal direction is taken from the CURRENT
 * position (x,y)/hypot so no atan2 is needed and every op is IEEE-exact apart from the shared
 * sincos: x_tgt = (R'/hypot(x0,y0)) * (x0*c - y0*s, x0*s + y0*c), z_tgt = Z'. */
void ppo_toroidal_push(ppo_ps* ps, const ppo_mesh* mesh, int m_x, int m_xtgt, int m_b, int m_phi,
                       double h, double k, double d, double deg, int trigmode) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
#pragma omp parallel for schedule(static)
  for (int pid = 0; pid < s.cap; ++pid) {
    if (!s.mask[pid]) continue;
    const int e = s.elem[pid];
    const double centerFactor = mesh->class_id[e] == 1 ? 0.01 : 1.0;
    const double distByClass = centerFactor * (double)1.0 / mesh->class_id[e];
    const double degP = deg * distByClass;
    const float phi = MF(ps, m_phi)[pid];
    const float b = MF(ps, m_b)[pid];
    const double a = b * d;
    const double dphi = degP * M_PI / 180.0;
    const double rad = phi + dphi;
    double sn, cs, st, ct;
    trig(trigmode, rad, &sn, &cs);
    trig(trigmode, dphi, &st, &ct);
    const double Rn = a * cs + h;
    const double Zn = b * sn + k;
    const double x0 = MD(ps, m_x)[pid];
    const double y0 = MD(ps, m_x)[A + pid];
    const double r0 = sqrt(x0 * x0 + y0 * y0);
    const double sc = Rn / r0;
    MD(ps, m_xtgt)[pid] = sc * (x0 * ct - y0 * st);
    MD(ps, m_xtgt)[A + pid] = sc * (x0 * st + y0 * ct);
    MD(ps, m_xtgt)[2 * A + pid] = Zn;
    MF(ps, m_phi)[pid] = (float)rad;
  }
  free_slots(&s);
}

/* test/pseudoPushAndSearch.cpp:104-115 (ptclUnique_d is all zeros) */
void ppo_linear_push(ppo_ps* ps, int m_x, int m_xtgt, double distance, double dx, double dy,
                     double dz) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  const double disp[4] = {distance, dx, dy, dz};
  for (int pid = 0; pid < s.cap; ++pid) {
    if (!s.mask[pid]) continue;
    double dir[3];
    dir[0] = disp[0] * disp[1];
    dir[1] = disp[0] * disp[2];
    dir[2] = disp[0] * disp[3];
    for (int i = 0; i < 3; ++i)
      MD(ps, m_xtgt)[i * A + pid] = MD(ps, m_x)[i * A + pid] + dir[i] + 0.0;
  }
  free_slots(&s);
}

/* src/pumipic_push.hpp:26-71, over n particles */
void ppo_push_boris(int n, double* x, double* y, double* z, double* xp, double* yp, double* zp,
                    double* vx, double* vy, double* vz, const double* ex, const double* ey,
                    const double* ez, const double* br, const double* bt, const double* bz,
                    double dt) {
  for (int pid = 0; pid < n; ++pid) {
    ppo_v3 vel = {{vx[pid], vy[pid], vz[pid]}};
    const ppo_v3 eField = {{ex[pid], ey[pid], ez[pid]}};
    const ppo_v3 bField = {{br[pid], bt[pid], bz[pid]}};
    const double charge = 1, amu = 10;
    /* osh_mag: Omega_h-side helper not in tree; Euclidean norm */
    const double bFieldMag = ppo_norm3(bField);
    const double qPrime = charge * 1.60217662e-19 / (amu * 1.6737236e-27) * dt * 0.5;
    const double coeff = 2.0 * qPrime / (1.0 + (qPrime * bFieldMag) * (qPrime * bFieldMag));
    const ppo_v3 qpE = ppo_scale3(eField, qPrime);
    const ppo_v3 vMinus = ppo_sub3(vel, qpE);
    const ppo_v3 vmxB = ppo_cross3(vMinus, bField);
    const ppo_v3 qpVmxB = ppo_scale3(vmxB, qPrime);
    const ppo_v3 vPrime = ppo_add3(vMinus, qpVmxB);
    const ppo_v3 vpxB = ppo_cross3(vPrime, bField);
    const ppo_v3 cVpxB = ppo_scale3(vpxB, coeff);
    vel = ppo_add3(vMinus, cVpxB);
    vel = ppo_add3(vel, qpE);
    const double pre[3] = {xp[pid], yp[pid], zp[pid]};
    xp[pid] = x[pid];
    yp[pid] = y[pid];
    zp[pid] = z[pid];
    x[pid] = pre[0] + vel.v[0] * dt;
    y[pid] = pre[1] + vel.v[1] * dt;
    z[pid] = pre[2] + vel.v[2] * dt;
    vx[pid] = vel.v[0];
    vy[pid] = vel.v[1];
    vz[pid] = vel.v[2];
  }
}

/* test/pseudoXGCm.cpp:102-114: every visited slot, masked or not */
void ppo_update_positions(ppo_ps* ps, int m_x, int m_xtgt) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  for (int pid = 0; pid < s.cap; ++pid) {
    if (s.elem[pid] < 0) continue;
    for (int i = 0; i < 3; ++i) {
      MD(ps, m_x)[i * A + pid] = MD(ps, m_xtgt)[i * A + pid];
      MD(ps, m_xtgt)[i * A + pid] = 0;
    }
  }
  free_slots(&s);
}

/* performance_tests/ps_combo160.cpp:158-178 */
void ppo_pseudo_push160(ppo_ps* ps, const double* parentElmData) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  for (int p = 0; p < s.cap; ++p) {
    const int e = s.elem[p];
    if (e < 0) continue;
    if (s.mask[p]) {
      for (int i = 0; i < 17; ++i) {
        double v = 10.3;
        v = v * v * v / sqrt((double)p) / sqrt((double)e) + parentElmData[e];
        MD(ps, 0)[i * A + p] = v;
      }
      for (int i = 0; i < 4; ++i) MI(ps, 1)[i * A + p] = 4 * p + i;
      ML(ps, 2)[p] = p;
    } else {
      for (int i = 0; i < 17; ++i) MD(ps, 0)[i * A + p] = 0;
      for (int i = 0; i < 4; ++i) MI(ps, 1)[i * A + p] = -1;
      ML(ps, 2)[p] = 0;
    }
  }
  free_slots(&s);
}

/* test/gyroScatter.hpp:101-166 with searchAndBuildMap :28-95.  The reference loads the ring
 * points into a scratch SCS and calls search_mesh_2d (maxLoops 100); the per-point walk is the
 * same function of (start element, target), so the single-point walker (hpp:1160-1252) is used. */
void ppo_create_gyro_ring_mappings(const ppo_mesh* mesh, double rmax, int gnr, int gppr,
                                   double theta_deg, int trigmode, int* forward_map,
                                   int* backward_map) {
  const int nverts = mesh->nverts;
  const long num_points = (long)nverts * gnr * gppr;
  const double torad = M_PI / 180;
  const int dim = mesh->dim, nvpe = dim + 1;
  for (long id = 0; id < num_points; ++id) {
    const int point_id = (int)(id % gppr);
    const long id2 = id / gppr;
    const int ring_id = (int)(id2 % gnr);
    const int vert_id = (int)(id2 / gnr);
    const double radius = rmax * (ring_id + 1) / gnr;
    const double deg = theta_deg + (((double)point_id) / gppr * 360);
    const double rad = deg * torad;
    double sn, cs;
    trig(trigmode, rad, &sn, &cs);
    const int start_elem = mesh->vert2elems[mesh->vert2elems_off[vert_id]];
    int parent;
    if (dim == 2) {
      double pt[2];
      pt[0] = mesh->coords[(size_t)vert_id * 2] + radius * cs;
      pt[1] = mesh->coords[(size_t)vert_id * 2 + 1] + radius * sn;
      /* centroid of the start element = average(vtxCoords) = ((p0+p1)+p2)/3 (unused by the walk) */
      double orig[2] = {0, 0};
      for (int c = 0; c < 2; ++c) {
        double acc = mesh->coords[(size_t)mesh->elem2verts[(size_t)start_elem * 3] * 2 + c];
        acc = acc + mesh->coords[(size_t)mesh->elem2verts[(size_t)start_elem * 3 + 1] * 2 + c];
        acc = acc + mesh->coords[(size_t)mesh->elem2verts[(size_t)start_elem * 3 + 2] * 2 + c];
        orig[c] = acc / 3;
      }
      int loops = 0;
      parent = ppo_search_mesh_2d_pt(mesh, orig, pt, (int)id, start_elem, &loops, 100);
    } else {
      const double xv = mesh->coords[(size_t)vert_id * 3], yv = mesh->coords[(size_t)vert_id * 3 + 1],
                   zv = mesh->coords[(size_t)vert_id * 3 + 2];
      const double Rv = sqrt(xv * xv + yv * yv);
      const double Rp = Rv + radius * cs;
      const double sc = Rp / Rv;
      const double pt[3] = {sc * xv, sc * yv, zv + radius * sn};
      parent = ppo_search_mesh_3d_pt(mesh, pt, start_elem, 100);
    }
    for (int i = 0; i < nvpe; ++i) {
      const int v = (parent >= 0) ? mesh->elem2verts[(size_t)parent * nvpe + i] : -1;
      forward_map[id * nvpe + i] = v;
      backward_map[id * nvpe + i] = v; /* forward/backward projections are identical (:125-134) */
    }
  }
}

/* test/gyroScatter.hpp:168-229 */
void ppo_gyro_scatter(const ppo_mesh* mesh, const ppo_ps* ps, const int* v2v, double rmax, int gnr,
                      int gppr, double* scatter_w) {
  const int nvpe = mesh->dim + 1; /* 3 = literal; 4 = documented tet variant */
  const double ringWidth = rmax / gnr;
  double* ring_accum = (double*)xcalloc((size_t)gnr * mesh->nverts, sizeof(double));
  slots s = get_slots(ps);
  for (int pid = 0; pid < s.cap; ++pid) {
    if (!s.mask[pid]) continue;
    const int e = s.elem[pid];
    const double ptclRadius = ringWidth * 1.125;
    int ringDown = 0;
    for (int i = 2; i <= gnr; i++) ringDown += (ptclRadius >= ringWidth * i);
    const int ringUp = ringDown + 1;
    for (int i = 0; i < nvpe; ++i) {
      const int v = mesh->elem2verts[(size_t)e * nvpe + i];
      ring_accum[(size_t)v * gnr + ringUp] += 1;
      ring_accum[(size_t)v * gnr + ringDown] += 1;
    }
  }
  for (int v = 0; v < mesh->nverts; ++v) scatter_w[v] = 0;
  for (int v = 0; v < mesh->nverts; ++v) {
    const long vtxIdx = (long)v * gnr * gppr;
    for (int ring = 0; ring < gnr; ++ring) {
      const double accumRingVal = ring_accum[(size_t)v * gnr + ring] / gppr;
      for (int pt = 0; pt < gppr; ++pt) {
        const long ptIdx = nvpe * (vtxIdx + (long)ring * gppr + pt);
        for (int elmVtx = 0; elmVtx < nvpe; ++elmVtx) {
          const int mappedVtx = v2v[ptIdx + elmVtx];
          if (mappedVtx >= 0) scatter_w[mappedVtx] += accumRingVal;
        }
      }
    }
  }
  free(ring_accum);
  free_slots(&s);
}

/* test/gyroScatter.hpp:168-229 with the particle radius the reference leaves as a TODO (:184
 * "const auto ptclRadius = ringWidth*1.125; //TODO compute the radius") taken per particle, and an
 * optional per-particle weight instead of the literal 1.  Same ring selection (:186-191) and the same
 * two additions per element vertex (:193-200); a particle whose upper ring would be >= gnr (the
 * reference asserts ringUp < gnr, :190) contributes to its lower ring only and is counted in
 * *num_clipped.  radius / weight are slot-indexed; weight == NULL means 1. */
void ppo_gyro_scatter_radius(const ppo_mesh* mesh, const ppo_ps* ps, const double* radius, const double* weight,
                             const int* v2v, double rmax, int gnr, int gppr, double* scatter_w,
                             int* num_clipped) {
  const int nvpe = mesh->dim + 1;
  const double ringWidth = rmax / gnr;
  double* ring_accum = (double*)xcalloc((size_t)gnr * mesh->nverts, sizeof(double));
  slots s = get_slots(ps);
  int clipped = 0;
  for (int pid = 0; pid < s.cap; ++pid) {
    if (!s.mask[pid]) continue;
    const int e = s.elem[pid];
    const double ptclRadius = radius[pid];
    const double w = weight ? weight[pid] : 1.0;
    int ringDown = 0;
    for (int i = 2; i <= gnr; i++) ringDown += (ptclRadius >= ringWidth * i);
    const int ringUp = ringDown + 1;
    if (ringUp >= gnr) ++clipped;
    for (int i = 0; i < nvpe; ++i) {
      const int v = mesh->elem2verts[(size_t)e * nvpe + i];
      if (ringUp < gnr) ring_accum[(size_t)v * gnr + ringUp] += w;
      ring_accum[(size_t)v * gnr + ringDown] += w;
    }
  }
  for (int v = 0; v < mesh->nverts; ++v) scatter_w[v] = 0;
  for (int v = 0; v < mesh->nverts; ++v) {
    const long vtxIdx = (long)v * gnr * gppr;
    for (int ring = 0; ring < gnr; ++ring) {
      const double accumRingVal = ring_accum[(size_t)v * gnr + ring] / gppr;
      for (int pt = 0; pt < gppr; ++pt) {
        const long ptIdx = nvpe * (vtxIdx + (long)ring * gppr + pt);
        for (int elmVtx = 0; elmVtx < nvpe; ++elmVtx) {
          const int mappedVtx = v2v[ptIdx + elmVtx];
          if (mappedVtx >= 0) scatter_w[mappedVtx] += accumRingVal;
        }
      }
    }
  }
  if (num_clipped) *num_clipped = clipped;
  free(ring_accum);
  free_slots(&s);
}

/* test/pseudoPushAndSearch.cpp:340-374 (counts every visited slot, masked or not) */
void ppo_avg_ptcl_density(const ppo_mesh* mesh, const ppo_ps* ps, double* elem_cnt,
                          double* vert_density) {
  slots s = get_slots(ps);
  for (int e = 0; e < mesh->nelems; ++e) elem_cnt[e] = 0;
  for (int pid = 0; pid < s.cap; ++pid)
    if (s.elem[pid] >= 0 && s.elem[pid] < mesh->nelems) elem_cnt[s.elem[pid]] += 1;
  for (int v = 0; v < mesh->nverts; ++v) {
    const int first = mesh->vert2elems_off[v];
    const int deg = mesh->vert2elems_off[v + 1] - first;
    double val = 0.00;
    for (int j = 0; j < deg; ++j) val += elem_cnt[mesh->vert2elems[first + j]];
    vert_density[v] = val / deg;
  }
  free_slots(&s);
}

/* src/pumipic_adjacency.hpp:772-790.  The reference gathers a 4-vector of field values and
 * indexes it with d*dof+comp, which is only in range for dof==1; that case is restated. */
double ppo_interpolate_tet_vtx(const ppo_mesh* mesh, const double* field, int elem,
                               const double bcc[4], int dof, int comp) {
  (void)dof;
  (void)comp;
  double val = 0;
  for (int fi = 0; fi < 4; ++fi) {
    const int d = PPO_TET_OPP[fi];
    const int v = mesh->elem2verts[(size_t)elem * 4 + d];
    val = val + bcc[fi] * field[v];
  }
  return val;
}

/* ------------------------------------------------------------------ gather side
 * src/pumipic_adjacency.hpp:772-809 and src/pumipic_utils.hpp:186-241,375-454, one particle at a
 * time over the structure (the library's pp_gather_tet_vtx / pp_interp*). */
static int find_bcc_tet(const ppo_mesh* mesh, int elem, ppo_v3 pos, double bcc[4]) {
  ppo_v3 M[4];
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 3; ++c)
      M[i].v[c] = mesh->coords[(size_t)mesh->elem2verts[(size_t)elem * 4 + i] * 3 + c];
  return ppo_find_barycentric_tet(M, pos, bcc);
}
void ppo_gather_tet_vtx(const ppo_mesh* mesh, const ppo_ps* ps, int m_x, const int* elem_ids,
                        const double* field, int dof, double* out, int* num_degenerate) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  int bad = 0;
  for (int pid = 0; pid < s.cap; ++pid) {
    const int e = !s.mask[pid] ? -1 : (elem_ids ? elem_ids[pid] : s.elem[pid]);
    double bcc[4];
    int ok = e >= 0;
    if (ok) {
      ppo_v3 pos = {{MD(ps, m_x)[pid], MD(ps, m_x)[A + pid], MD(ps, m_x)[2 * A + pid]}};
      ok = find_bcc_tet(mesh, e, pos, bcc);
      if (!ok) ++bad;
    }
    for (int c = 0; c < dof; ++c) {
      double val = 0;
      if (ok)
        for (int fi = 0; fi < 4; ++fi) {
          const int v = mesh->elem2verts[(size_t)e * 4 + PPO_TET_OPP[fi]];
          val = val + bcc[fi] * field[(size_t)v * dof + c];
        }
      out[(size_t)c * s.cap + pid] = val;
    }
  }
  if (num_degenerate) *num_degenerate = bad;
  free_slots(&s);
}
static double interp2d_field_pt(const double* data, double gridx0, double gridz0, double dx, double dz,
                                int nx, int nz, const double pos[3], int cyl, int nComp, int comp) {
  if (nx * nz == 1) return data[comp];
  double fxz = 0, fx_z1 = 0, fx_z2 = 0;
  double dim1 = pos[0];
  const double z = pos[2];
  if (cyl) dim1 = sqrt(pos[0] * pos[0] + pos[1] * pos[1]);
  int i = (int)floor((dim1 - gridx0) / dx);
  int j = (int)floor((z - gridz0) / dz);
  if (i < 0) i = 0;
  if (j < 0) j = 0;
  const double gridXi = gridx0 + i * dx, gridXip1 = gridx0 + (i + 1) * dx;
  const double gridZj = gridz0 + j * dz, gridZjp1 = gridz0 + (j + 1) * dz;
  if (i >= nx - 1 && j >= nz - 1) {
    fxz = data[(nx - 1 + (nz - 1) * nx) * nComp + comp];
  } else if (i >= nx - 1) {
    fx_z1 = data[(nx - 1 + j * nx) * nComp + comp];
    fx_z2 = data[(nx - 1 + (j + 1) * nx) * nComp + comp];
    fxz = ((gridZjp1 - z) * fx_z1 + (z - gridZj) * fx_z2) / dz;
  } else if (j >= nz - 1) {
    fx_z1 = data[(i + (nz - 1) * nx) * nComp + comp];
    fx_z2 = data[(i + (nz - 1) * nx) * nComp + comp];
    fxz = ((gridXip1 - dim1) * fx_z1 + (dim1 - gridXi) * fx_z2) / dx;
  } else {
    fx_z1 = ((gridXip1 - dim1) * data[(i + j * nx) * nComp + comp] +
             (dim1 - gridXi) * data[(i + 1 + j * nx) * nComp + comp]) / dx;
    fx_z2 = ((gridXip1 - dim1) * data[(i + (j + 1) * nx) * nComp + comp] +
             (dim1 - gridXi) * data[(i + 1 + (j + 1) * nx) * nComp + comp]) / dx;
    fxz = ((gridZjp1 - z) * fx_z1 + (z - gridZj) * fx_z2) / dz;
  }
  return fxz;
}
static double i2d_base(double d1, double d2, double g1, double g2, double v, double dv) {
  return (d1 * (g2 - v) + d2 * (v - g1)) / dv;
}
/* interpolate2d_field -> interpolate2d (utils.hpp:258-322), cylSymm already applied to x */
static double interp2d_field2_pt(const double* data, double gridx0, double gridz0, double dx, double dz,
                                 int nx, int nz, const double pos[3], int cyl, int nComp, int comp) {
  if (nx <= 1 && nz <= 1) return data[comp];
  double x = pos[0];
  const double z = pos[2];
  if (cyl) x = sqrt(x * x + pos[1] * pos[1]);
  int i = (int)floor((x - gridx0) / dx);
  int j = (int)floor((z - gridz0) / dz);
  if (i < 0) i = 0;
  if (j < 0) j = 0;
  const double gXi = gridx0 + i * dx, gXip1 = gridx0 + (i + 1) * dx;
  const double gZj = gridz0 + j * dz, gZjp1 = gridz0 + (j + 1) * dz;
  if (i >= nx - 1 && j >= nz - 1) return data[(nx - 1 + (nz - 1) * nx) * nComp + comp];
  if (i >= nx - 1)
    return i2d_base(data[(nx - 1 + j * nx) * nComp + comp], data[(nx - 1 + (j + 1) * nx) * nComp + comp],
                    z - gZj, gZjp1 - z, z, dz);
  if (j >= nz - 1)
    return i2d_base(data[(i + (nz - 1) * nx) * nComp + comp], data[(i + (nz - 1) * nx) * nComp + comp],
                    x - gXi, gXip1 - x, x, dx);
  const double f1 = i2d_base(data[(i + j * nx) * nComp + comp], data[(i + 1 + j * nx) * nComp + comp], gXi,
                             gXip1, x, dx);
  const double f2 = i2d_base(data[(i + (j + 1) * nx) * nComp + comp],
                             data[(i + 1 + (j + 1) * nx) * nComp + comp], gXi, gXip1, x, dx);
  return i2d_base(f1, f2, gZj, gZjp1, z, dz);
}
void ppo_interp2d_field(const ppo_ps* ps, int m_x, const double* data, double gridx0, double gridz0,
                        double dx, double dz, int nx, int nz, int cyl, int ncomp, int comp, double* out) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  for (int pid = 0; pid < s.cap; ++pid) {
    out[pid] = 0;
    if (!s.mask[pid]) continue;
    const double pos[3] = {MD(ps, m_x)[pid], MD(ps, m_x)[A + pid], MD(ps, m_x)[2 * A + pid]};
    out[pid] = interp2d_field_pt(data, gridx0, gridz0, dx, dz, nx, nz, pos, cyl, ncomp, comp);
  }
  free_slots(&s);
}
void ppo_interp2d_vector(const ppo_ps* ps, int m_x, const double* data3, double gridx0, double gridz0,
                         double dx, double dz, int nx, int nz, int cyl, double* out) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  for (int pid = 0; pid < s.cap; ++pid) {
    double f[3] = {0, 0, 0};
    if (s.mask[pid]) {
      const double pos[3] = {MD(ps, m_x)[pid], MD(ps, m_x)[A + pid], MD(ps, m_x)[2 * A + pid]};
      for (int i = 0; i < 3; ++i)
        f[i] = interp2d_field2_pt(data3, gridx0, gridz0, dx, dz, nx, nz, pos, cyl, 3, i);
      if (cyl) {
        const double theta = atan2(pos[1], pos[0]);
        const double f0 = f[0], f1 = f[1];
        f[0] = cos(theta) * f0 - sin(theta) * f1;
        f[1] = sin(theta) * f0 + cos(theta) * f1;
      }
    }
    for (int c = 0; c < 3; ++c) out[(size_t)c * s.cap + pid] = f[c];
  }
  free_slots(&s);
}
void ppo_interp3d_field(const ppo_ps* ps, int m_x, int nx, int ny, int nz, const double* gridx,
                        const double* gridy, const double* gridz, const double* data, double* out) {
  slots s = get_slots(ps);
  const size_t A = (size_t)ps->alloc;
  const double dx = gridx[1] - gridx[0];
  const double dy = ny > 1 ? gridy[1] - gridy[0] : 1.0, dz = nz > 1 ? gridz[1] - gridz[0] : 1.0;
  for (int pid = 0; pid < s.cap; ++pid) {
    out[pid] = 0;
    if (!s.mask[pid]) continue;
    const double x = MD(ps, m_x)[pid], y = MD(ps, m_x)[A + pid], z = MD(ps, m_x)[2 * A + pid];
    int i = (int)floor((x - gridx[0]) / dx);
    int j = ny > 1 ? (int)floor((y - gridy[0]) / dy) : 0;
    int k = nz > 1 ? (int)floor((z - gridz[0]) / dz) : 0;
    i = (i < 0) ? 0 : ((i >= nx - 1) ? (nx - 2) : i);
    j = (j < 0 || ny <= 1) ? 0 : ((j >= ny - 1) ? (ny - 2) : j);
    k = (k < 0 || nz <= 1) ? 0 : ((k >= nz - 1) ? (nz - 2) : k);
#define BASEG(di) i2d_base(data[(di)], data[(di) + 1], gridx[i], gridx[i + 1], x, dx)
    const double fx_z0 = BASEG(i + j * nx + k * nx * ny);
    double fxyz = fx_z0;
    if (nz > 1) {
      const double fx_z1 = BASEG(i + j * nx + (k + 1) * nx * ny);
      const double fxz0 = i2d_base(fx_z0, fx_z1, gridz[k], gridz[k + 1], z, dz);
      fxyz = fxz0;
      if (ny > 1) {
        const double fxy_z0 = BASEG(i + (j + 1) * nx + k * nx * ny);
        const double fxy_z1 = BASEG(i + (j + 1) * nx + (k + 1) * nx * ny);
        const double fxz1 = i2d_base(fxy_z0, fxy_z1, gridz[k], gridz[k + 1], z, dz);
        fxyz = i2d_base(fxz0, fxz1, gridy[j], gridy[j + 1], y, dy);
      }
    }
#undef BASEG
    out[pid] = fxyz;
  }
  free_slots(&s);
}

/* ------------------------------------------------------------------ closest point on a triangle
 * closest_point_on_triangle_wnormal src/pumipic_adjacency.hpp:824-906 (scalar intermediates are
 * `float` in the reference and here); closest_point_on_triangle :910-1009 */
static ppo_v3 v3_of(const double* p) {
  ppo_v3 r = {{p[0], p[1], p[2]}};
  return r;
}
static void put3(double q[3], ppo_v3 v) {
  q[0] = v.v[0];
  q[1] = v.v[1];
  q[2] = v.v[2];
}
static void closest_wnormal(ppo_v3 a, ppo_v3 b, ppo_v3 c, ppo_v3 p, double q[3], int* reg) {
  const ppo_v3 ab = ppo_sub3(b, a), ac = ppo_sub3(c, a), bc = ppo_sub3(c, b);
  const float snom = (float)ppo_dot3(ppo_sub3(p, a), ab);
  const float sdenom = (float)ppo_dot3(ppo_sub3(p, b), ppo_sub3(a, b));
  const float tnom = (float)ppo_dot3(ppo_sub3(p, a), ac);
  const float tdenom = (float)ppo_dot3(ppo_sub3(p, c), ppo_sub3(a, c));
  if (snom <= 0.0 && tnom <= 0.0) {
    if (reg) *reg = 0;
    put3(q, a);
    return;
  }
  const float unom = (float)ppo_dot3(ppo_sub3(p, b), bc);
  const float udenom = (float)ppo_dot3(ppo_sub3(p, c), ppo_sub3(b, c));
  if (sdenom <= 0.0 && unom <= 0.0) {
    if (reg) *reg = 1;
    put3(q, b);
    return;
  }
  if (tdenom <= 0.0 && udenom <= 0.0) {
    if (reg) *reg = 2;
    put3(q, c);
    return;
  }
  const ppo_v3 n = ppo_cross3(ppo_sub3(b, a), ppo_sub3(c, a));
  const float vc = (float)ppo_dot3(n, ppo_cross3(ppo_sub3(a, p), ppo_sub3(b, p)));
  if (vc <= 0.0 && snom >= 0.0 && sdenom >= 0.0) {
    put3(q, ppo_add3(a, ppo_scale3(ab, (double)(snom / (snom + sdenom)))));
    if (reg) *reg = 3;
    return;
  }
  const float va = (float)ppo_dot3(n, ppo_cross3(ppo_sub3(b, p), ppo_sub3(c, p)));
  if (va <= 0.0 && unom >= 0.0 && udenom >= 0.0) {
    put3(q, ppo_add3(b, ppo_scale3(bc, (double)(unom / (unom + udenom)))));
    if (reg) *reg = 5;
    return;
  }
  const float vb = (float)ppo_dot3(n, ppo_cross3(ppo_sub3(c, p), ppo_sub3(a, p)));
  if (vb <= 0.0 && tnom >= 0.0 && tdenom >= 0.0) {
    put3(q, ppo_add3(a, ppo_scale3(ac, (double)(tnom / (tnom + tdenom)))));
    if (reg) *reg = 4;
    return;
  }
  const float u = va / (va + vb + vc);
  const float v = vb / (va + vb + vc);
  const float w = (float)(1.0 - u - v);
  put3(q, ppo_add3(ppo_add3(ppo_scale3(a, (double)u), ppo_scale3(b, (double)v)), ppo_scale3(c, (double)w)));
  if (reg) *reg = 6;
}
static void closest_plain(ppo_v3 pta, ppo_v3 ptb, ppo_v3 ptc, ppo_v3 ptp, double q[3], int* reg) {
  const ppo_v3 vab = ppo_sub3(ptb, pta), vac = ppo_sub3(ptc, pta), vap = ppo_sub3(ptp, pta);
  const double d1 = ppo_dot3(vab, vap), d2 = ppo_dot3(vac, vap);
  if (d1 <= 0 && d2 <= 0) {
    put3(q, pta);
    if (reg) *reg = 0;
    return;
  }
  const ppo_v3 vbp = ppo_sub3(ptp, ptb);
  const double d3 = ppo_dot3(vab, vbp), d4 = ppo_dot3(vac, vbp);
  if (d3 >= 0 && d4 <= d3) {
    put3(q, ptb);
    if (reg) *reg = 1;
    return;
  }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    const double v = d1 / (d1 - d3);
    put3(q, ppo_add3(ppo_scale3(vab, v), pta));
    return; /* region not reported in this branch (hpp:951-958) */
  }
  const ppo_v3 vcp = ppo_sub3(ptp, ptc);
  const double d5 = ppo_dot3(vab, vcp), d6 = ppo_dot3(vac, vcp);
  if (d6 >= 0 && d5 <= d6) {
    put3(q, ptc);
    if (reg) *reg = 2;
    return;
  }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    const double w = d2 / (d2 - d6);
    put3(q, ppo_add3(ppo_scale3(vac, w), pta));
    if (reg) *reg = 4;
    return;
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    put3(q, ppo_add3(ptb, ppo_scale3(ppo_sub3(ptc, ptb), w)));
    if (reg) *reg = 5;
    return;
  }
  const double inv = 1.0 / (va + vb + vc);
  const double v = vb * inv, w = vc * inv;
  put3(q, ppo_add3(ppo_add3(pta, ppo_scale3(vab, v)), ppo_scale3(vac, w)));
  if (reg) *reg = 6;
}
void ppo_closest_point_on_triangle(const double abc[9], const double p[3], int wnormal, double q[3],
                                   int* reg) {
  const ppo_v3 a = v3_of(abc), b = v3_of(abc + 3), c = v3_of(abc + 6), pp = v3_of(p);
  if (wnormal)
    closest_wnormal(a, b, c, pp, q, reg);
  else
    closest_plain(a, b, c, pp, q, reg);
}

/* ------------------------------------------------------------------ PICpart safe zone / buffer
 * src/pumipic_part_construct.cpp:387-468 (BFS, bfsBufferLayers, bfsSafeInward), kernel by kernel */
static int bridge_adj(const ppo_mesh* mesh, int bridge_dim, const int** off, const int** vals) {
  if (bridge_dim == 0) {
    *off = mesh->vert2elems_off;
    *vals = mesh->vert2elems;
    return mesh->nverts;
  }
  *off = mesh->side2elems_off;
  *vals = mesh->side2elems;
  return mesh->nsides;
}
static void bfs_sweep(int nb, const int* off, const int* vals, const int* visited, int* next) {
  for (int b = 0; b < nb; ++b) { /* :387-405 */
    const int deg = off[b + 1] - off[b], first = off[b];
    int is_visited_here = 0;
    for (int j = 0; j < deg; ++j)
      if (visited[vals[first + j]]) is_visited_here = 1;
    const int loops = deg * is_visited_here;
    for (int j = 0; j < loops; ++j) next[vals[first + j]] = 1;
  }
}
void ppo_bfs_buffer_layers(const ppo_mesh* mesh, int bridge_dim, int rank, int comm_size,
                           int safe_layers, int ghost_layers, const int* owner,
                           unsigned char* is_safe, int* has_part) {
  const int ne = mesh->nelems;
  int* visited = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  int* next = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  for (int e = 0; e < ne; ++e) { /* initVisit :414-418 */
    visited[e] = (owner[e] == rank);
    is_safe[e] = (unsigned char)visited[e];
    next[e] = visited[e];
  }
  for (int r = 0; r < comm_size; ++r) has_part[r] = 0;
  has_part[rank] = 1; /* initSelfPart */
  const int *off, *vals;
  const int nb = bridge_adj(mesh, bridge_dim, &off, &vals);
  for (int i = 0; i < ghost_layers || i < safe_layers; ++i) {
    bfs_sweep(nb, off, vals, visited, next);
    for (int e = 0; e < ne; ++e) { /* copyVisit :427-435 */
      visited[e] = next[e];
      if (i == safe_layers - 1) is_safe[e] = (unsigned char)next[e];
      if (i < ghost_layers && visited[e]) has_part[owner[e]] = 1;
    }
  }
  free(visited);
  free(next);
}
void ppo_bfs_safe_inward(const ppo_mesh* mesh, int bridge_dim, int rank, int safe_layers,
                         const int* owner, const int* has_part, unsigned char* safe) {
  const int ne = mesh->nelems;
  int* visited = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  int* next = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  for (int e = 0; e < ne; ++e) visited[e] = next[e] = !has_part[owner[e]]; /* :444-448 */
  const int *off, *vals;
  const int nb = bridge_adj(mesh, bridge_dim, &off, &vals);
  for (int i = 0; i < safe_layers; ++i) {
    bfs_sweep(nb, off, vals, visited, next);
    for (int e = 0; e < ne; ++e) visited[e] = next[e];
  }
  for (int e = 0; e < ne; ++e) safe[e] = (unsigned char)(!visited[e] || owner[e] == rank); /* :461-466 */
  free(visited);
  free(next);
}
