// pp_gather.hip -- gather side: mesh / grid field -> particle, one thread per slot.
//   interpolateTetVtx / interpolate3dFieldTet / findBCCoordsInTet   src/pumipic_adjacency.hpp:772-809
//   interpolate2dField / interp2dVector / interpolate3d_field       src/pumipic_utils.hpp:186-454
// The per-particle arithmetic lives in include/pumipic_gather.hpp (shared with user lambdas).
#include "../include/pumipic_gather.hpp"
#include "../include/pumipic_wall.hpp"
#include "pp_geom.hpp"
#include "pp_internal.hpp"
#include "pp_push_math.hpp"

namespace {
using pp::grid_for;
using pp::kBlock;

__global__ void k_gather_tet_vtx(int capacity, const unsigned char* __restrict__ mask,
                                 const int* __restrict__ slot_elem, const int* __restrict__ elem_ids,
                                 const double* __restrict__ x, long long stride,
                                 const double* __restrict__ coords, const int* __restrict__ e2v,
                                 const double* __restrict__ field, int dof, double* __restrict__ out,
                                 int* __restrict__ bad) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = !mask[pid] ? -1 : (elem_ids ? elem_ids[pid] : slot_elem[pid]);
  double bcc[4];
  bool ok = e >= 0;
  if (ok) {
    const double pos[3] = {x[pid], x[stride + pid], x[2 * stride + pid]};
    ok = pumipic::findBCCoordsInTet(coords, e2v, pos, e, bcc);
    if (!ok) atomicAdd(bad, 1);  // OMEGA_H_CHECK(res==1), adjacency.hpp:805
  }
  for (int c = 0; c < dof; ++c)
    out[(size_t)c * capacity + pid] = ok ? pumipic::interpolateTetVtx(e2v, field, e, bcc, dof, c) : 0.0;
}
__global__ void k_interp2d_field(int capacity, const unsigned char* __restrict__ mask,
                                 const double* __restrict__ x, long long stride,
                                 const double* __restrict__ data, double gridx0, double gridz0,
                                 double dx, double dz, int nx, int nz, int cyl, int ncomp, int comp,
                                 double* __restrict__ out) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  double v = 0;
  if (mask[pid]) {
    const double pos[3] = {x[pid], x[stride + pid], x[2 * stride + pid]};
    v = pumipic::interpolate2dField(data, gridx0, gridz0, dx, dz, nx, nz, pos, cyl != 0, ncomp, comp);
  }
  out[pid] = v;
}
__global__ void k_interp2d_vector(int capacity, const unsigned char* __restrict__ mask,
                                  const double* __restrict__ x, long long stride,
                                  const double* __restrict__ data3, double gridx0, double gridz0,
                                  double dx, double dz, int nx, int nz, int cyl,
                                  double* __restrict__ out) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  double f[3] = {0, 0, 0};
  if (mask[pid]) {
    const double pos[3] = {x[pid], x[stride + pid], x[2 * stride + pid]};
    pumipic::interp2dVector(data3, gridx0, gridz0, dx, dz, nx, nz, pos, f, cyl != 0);
  }
  for (int c = 0; c < 3; ++c) out[(size_t)c * capacity + pid] = f[c];
}
__global__ void k_interp3d_field(int capacity, const unsigned char* __restrict__ mask,
                                 const double* __restrict__ x, long long stride, int nx, int ny,
                                 int nz, const double* __restrict__ gridx,
                                 const double* __restrict__ gridy, const double* __restrict__ gridz,
                                 const double* __restrict__ data, double* __restrict__ out) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  out[pid] = mask[pid] ? pumipic::interpolate3d_field(x[pid], x[stride + pid], x[2 * stride + pid], nx,
                                                     ny, nz, gridx, gridy, gridz, data)
                       : 0.0;
}

int xmember(const pp_ps* ps, int m_x, const char* what, const double** x) {
  if (int rc = pp::ps_ready(ps)) return rc;
  if (m_x < 0 || m_x >= ps->nmembers) {
    pp::set_error(std::string(what) + ": member index out of range");
    return PP_EINVAL;
  }
  const int s = ps->member_map[m_x];
  if (ps->member_bytes[s] != 8 || ps->member_ncomp[s] < 3) {
    pp::set_error(std::string(what) + ": the position member must be double[3]");
    return PP_EINVAL;
  }
  *x = (const double*)ps->data[s].p;
  return PP_OK;
}
}  // namespace

// Gather + Boris push in one pass: E from a 3-dof vertex field of the particle's tet
// (interpolate3dFieldTet, adjacency.hpp:793-799), B from an (R,Z) grid (interp2dVector,
// utils.hpp:437-454), then pushBoris (pumipic_push.hpp:17-75) on the particle's own members.
// Equal, value for value, to pp_gather_tet_vtx + pp_interp2d_vector + pp_push_boris.
__global__ void k_boris_fields(int capacity, const unsigned char* __restrict__ mask,
                               const int* __restrict__ slot_elem, const int* __restrict__ elem_ids,
                               double* __restrict__ x, double* __restrict__ xp, double* __restrict__ v,
                               long long stride, const double* __restrict__ coords,
                               const int* __restrict__ e2v, const double* __restrict__ efield,
                               const double* __restrict__ bgrid, double gridx0, double gridz0, double dx,
                               double dz, int nx, int nz, int cyl, double dt, int* __restrict__ bad) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  const int e = elem_ids ? elem_ids[pid] : slot_elem[pid];
  const double pos[3] = {x[pid], x[stride + pid], x[2 * stride + pid]};
  double E[3] = {0, 0, 0}, B[3], bcc[4];
  if (e >= 0) {
    if (pumipic::findBCCoordsInTet(coords, e2v, pos, e, bcc))
      pumipic::interpolate3dFieldTet(e2v, efield, e, bcc, E);
    else
      atomicAdd(bad, 1);
  }
  pumipic::interp2dVector(bgrid, gridx0, gridz0, dx, dz, nx, nz, pos, B, cyl != 0);
  using ppg::V3;
  const V3 vel = ppm::boris_velocity(V3{v[pid], v[stride + pid], v[2 * stride + pid]}, V3{E[0], E[1], E[2]},
                                     V3{B[0], B[1], B[2]}, dt);
  const double p0 = xp[pid], p1 = xp[stride + pid], p2 = xp[2 * stride + pid];
  xp[pid] = pos[0];
  xp[stride + pid] = pos[1];
  xp[2 * stride + pid] = pos[2];
  x[pid] = p0 + vel.x * dt;
  x[stride + pid] = p1 + vel.y * dt;
  x[2 * stride + pid] = p2 + vel.z * dt;
  v[pid] = vel.x;
  v[stride + pid] = vel.y;
  v[2 * stride + pid] = vel.z;
}

// closest_point_on_triangle[_wnormal] over n (triangle, point) pairs; tri_stride 0 = one triangle
__global__ void k_closest_point(int n, const double* __restrict__ tris, int tri_stride,
                                const double* __restrict__ pts, int wnormal,
                                double* __restrict__ out, int* __restrict__ region) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double abc[9], p[3], q[3];
  for (int k = 0; k < 9; ++k) abc[k] = tris[(size_t)i * tri_stride + k];
  for (int k = 0; k < 3; ++k) p[k] = pts[(size_t)i * 3 + k];
  int reg = region ? region[i] : -1;  // EDGEAB of the plain form leaves the caller's value
  if (wnormal)
    pumipic::closest_point_on_triangle_wnormal(abc, p, q, &reg);
  else
    pumipic::closest_point_on_triangle(abc, p, q, &reg);
  for (int k = 0; k < 3; ++k) out[(size_t)i * 3 + k] = q[k];
  if (region) region[i] = reg;
}

// ray_intersects_triangle / line_segment_intersects_triangle (adjacency.tpp:152-201) over n (triangle,
// origin, destination) triples
__global__ void k_ray_triangle(int n, const double* __restrict__ tris, int tri_stride,
                               const double* __restrict__ orig, const double* __restrict__ dest, double tol,
                               const int* __restrict__ flip, int flip_all, int segment, int* __restrict__ hit,
                               double* __restrict__ xpoint, double* __restrict__ out3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ppg::V3 fv[3];
  for (int k = 0; k < 3; ++k)
    fv[k] = {tris[(size_t)i * tri_stride + 3 * k], tris[(size_t)i * tri_stride + 3 * k + 1],
             tris[(size_t)i * tri_stride + 3 * k + 2]};
  const ppg::V3 o{orig[(size_t)i * 3], orig[(size_t)i * 3 + 1], orig[(size_t)i * 3 + 2]};
  const ppg::V3 d{dest[(size_t)i * 3], dest[(size_t)i * 3 + 1], dest[(size_t)i * 3 + 2]};
  ppg::V3 xp;
  double dproj, closeness, param;
  bool h = ppg::ray_intersects_triangle(fv, o, d, xp, tol, flip ? flip[i] : flip_all, dproj, closeness, param);
  if (segment) h = h && param <= 1 + tol;  // tpp:192-201
  hit[i] = h ? 1 : 0;
  if (xpoint) {
    xpoint[(size_t)i * 3] = xp.x;
    xpoint[(size_t)i * 3 + 1] = xp.y;
    xpoint[(size_t)i * 3 + 2] = xp.z;
  }
  if (out3) {
    out3[(size_t)i * 3] = dproj;
    out3[(size_t)i * 3 + 1] = closeness;
    out3[(size_t)i * 3 + 2] = param;
  }
}

extern "C" {

int pp_ray_intersects_triangle(int n, const double* tris_dev, int tri_stride, const double* orig_dev,
                               const double* dest_dev, double tol, const int* flip_dev, int flip_all, int segment,
                               int* hit_dev, double* xpoint_dev, double* dproj_closeness_param_dev) {
  PP_REQUIRE(n >= 0 && tris_dev && orig_dev && dest_dev && hit_dev && (tri_stride == 0 || tri_stride >= 9),
             "pp_ray_intersects_triangle: bad argument");
  if (n == 0) return PP_OK;
  k_ray_triangle<<<grid_for(n), kBlock, 0, pp::stream()>>>(n, tris_dev, tri_stride, orig_dev, dest_dev, tol, flip_dev,
                                                          flip_all, segment, hit_dev, xpoint_dev,
                                                          dproj_closeness_param_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_gather_tet_vtx(const pp_mesh* mesh, const pp_ps* ps, int m_x, const int* elem_ids_dev,
                      const double* field_dev, int dof, double* out_dev, int* num_degenerate) {
  PP_REQUIRE(mesh && ps && field_dev && out_dev, "pp_gather_tet_vtx: null argument");
  PP_REQUIRE(mesh->dim == 3 && dof >= 1, "pp_gather_tet_vtx: needs a tet mesh and dof >= 1");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_gather_tet_vtx: structure/mesh element mismatch");
  const double* x;
  int rc = xmember(ps, m_x, "pp_gather_tet_vtx", &x);
  if (rc) return rc;
  if (num_degenerate) *num_degenerate = 0;
  if (ps->capacity == 0) return PP_OK;
  static pp::DevBuf* s_bad = new pp::DevBuf();
  PP_HIP_CHECK(s_bad->reserve(sizeof(int)));
  PP_HIP_CHECK(hipMemsetAsync(s_bad->p, 0, sizeof(int), pp::stream()));
  k_gather_tet_vtx<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), elem_ids_dev, x,
      ps->stride, mesh->d_coords.as<double>(), mesh->d_elem2verts.as<int>(), field_dev, dof, out_dev,
      s_bad->as<int>());
  PP_LAUNCH_CHECK();
  if (num_degenerate) {
    PP_HIP_CHECK(hipMemcpyAsync(num_degenerate, s_bad->p, sizeof(int), hipMemcpyDeviceToHost, pp::stream()));
    PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  }
  return PP_OK;
}

int pp_interp2d_field(const pp_ps* ps, int m_x, const double* data_dev, double gridx0, double gridz0,
                      double dx, double dz, int nx, int nz, int cyl_symm, int ncomp, int comp,
                      double* out_dev) {
  PP_REQUIRE(ps && data_dev && out_dev, "pp_interp2d_field: null argument");
  PP_REQUIRE(dx > 0 && dz > 0 && nx >= 1 && nz >= 1 && ncomp >= 1 && comp >= 0 && comp < ncomp,
             "pp_interp2d_field: bad grid (the reference checks dx > 0 && dz > 0, utils.hpp:205)");
  const double* x;
  int rc = xmember(ps, m_x, "pp_interp2d_field", &x);
  if (rc) return rc;
  if (ps->capacity == 0) return PP_OK;
  k_interp2d_field<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), x, ps->stride, data_dev, gridx0, gridz0, dx, dz, nx,
      nz, cyl_symm, ncomp, comp, out_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_interp2d_vector(const pp_ps* ps, int m_x, const double* data3_dev, double gridx0, double gridz0,
                       double dx, double dz, int nx, int nz, int cyl_symm, double* out_dev) {
  PP_REQUIRE(ps && data3_dev && out_dev, "pp_interp2d_vector: null argument");
  PP_REQUIRE(dx > 0 && dz > 0 && nx >= 1 && nz >= 1, "pp_interp2d_vector: bad grid");
  const double* x;
  int rc = xmember(ps, m_x, "pp_interp2d_vector", &x);
  if (rc) return rc;
  if (ps->capacity == 0) return PP_OK;
  k_interp2d_vector<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), x, ps->stride, data3_dev, gridx0, gridz0, dx, dz, nx,
      nz, cyl_symm, out_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_interp3d_field(const pp_ps* ps, int m_x, int nx, int ny, int nz, const double* gridx_dev,
                      const double* gridy_dev, const double* gridz_dev, const double* data_dev,
                      double* out_dev) {
  PP_REQUIRE(ps && gridx_dev && gridy_dev && gridz_dev && data_dev && out_dev,
             "pp_interp3d_field: null argument");
  PP_REQUIRE(nx >= 2 && ny >= 1 && nz >= 1, "pp_interp3d_field: needs nx >= 2");
  const double* x;
  int rc = xmember(ps, m_x, "pp_interp3d_field", &x);
  if (rc) return rc;
  if (ps->capacity == 0) return PP_OK;
  k_interp3d_field<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), x, ps->stride, nx, ny, nz, gridx_dev, gridy_dev,
      gridz_dev, data_dev, out_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_boris_push_fields(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xprev, int m_v,
                         const int* elem_ids_dev, const double* efield_vtx_dev, const double* bgrid_dev,
                         double gridx0, double gridz0, double dx, double dz, int nx, int nz, int cyl_symm,
                         double dt, int* num_degenerate) {
  PP_REQUIRE(mesh && ps && efield_vtx_dev && bgrid_dev, "pp_boris_push_fields: null argument");
  PP_REQUIRE(mesh->dim == 3, "pp_boris_push_fields: needs a tet mesh");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_boris_push_fields: structure/mesh element mismatch");
  PP_REQUIRE(dx > 0 && dz > 0 && nx >= 1 && nz >= 1 && dt > 0,
             "pp_boris_push_fields: bad grid or dt (utils.hpp:205, pumipic_push.hpp:38)");
  PP_REQUIRE(m_x != m_xprev && m_x != m_v && m_xprev != m_v, "pp_boris_push_fields: members must differ");
  const double *x, *xp, *v;
  int rc;
  if ((rc = xmember(ps, m_x, "pp_boris_push_fields x", &x))) return rc;
  if ((rc = xmember(ps, m_xprev, "pp_boris_push_fields x_prev", &xp))) return rc;
  if ((rc = xmember(ps, m_v, "pp_boris_push_fields v", &v))) return rc;
  if (num_degenerate) *num_degenerate = 0;
  if (ps->capacity == 0) return PP_OK;
  static pp::DevBuf* s_bad = new pp::DevBuf();
  PP_HIP_CHECK(s_bad->reserve(sizeof(int)));
  PP_HIP_CHECK(hipMemsetAsync(s_bad->p, 0, sizeof(int), pp::stream()));
  k_boris_fields<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), elem_ids_dev,
      (double*)x, (double*)xp, (double*)v, ps->stride, mesh->d_coords.as<double>(),
      mesh->d_elem2verts.as<int>(), efield_vtx_dev, bgrid_dev, gridx0, gridz0, dx, dz, nx, nz, cyl_symm,
      dt, s_bad->as<int>());
  PP_LAUNCH_CHECK();
  if (num_degenerate) {
    PP_HIP_CHECK(hipMemcpyAsync(num_degenerate, s_bad->p, sizeof(int), hipMemcpyDeviceToHost, pp::stream()));
    PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  }
  return PP_OK;
}

int pp_closest_point_on_triangle(int n, const double* tris_dev, int tri_stride,
                                 const double* pts_dev, int wnormal, double* out_dev,
                                 int* region_dev) {
  PP_REQUIRE(n >= 0 && tris_dev && pts_dev && out_dev, "pp_closest_point_on_triangle: null argument");
  PP_REQUIRE(tri_stride == 0 || tri_stride >= 9,
             "pp_closest_point_on_triangle: tri_stride is 0 (one triangle) or >= 9 doubles");
  if (n == 0) return PP_OK;
  k_closest_point<<<grid_for(n), kBlock, 0, pp::stream()>>>(n, tris_dev, tri_stride, pts_dev, wnormal,
                                                           out_dev, region_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // extern "C"
