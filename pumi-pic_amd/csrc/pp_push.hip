// pp_push.hip -- particle pushes (one streaming kernel each; see pp_search.hip for the fused
// push+walk kernel that is the hot path).
//   ellipticalPush::setup / push      test/ellipticalPush.hpp:10-70
//   linear push lambda                test/pseudoPushAndSearch.cpp:104-115
//   pushBoris                         src/pumipic_push.hpp:17-75 (launched over n)
//   updatePtclPositions               test/pseudoXGCm.cpp:102-114
//   pseudoPush                        performance_tests/ps_combo160.cpp:158-178
#include "pp_geom.hpp"
#include "pp_internal.hpp"
#include "pp_push_math.hpp"

namespace {
using pp::grid_for;
using pp::kBlock;

__global__ void k_elliptical_setup(int capacity, const unsigned char* __restrict__ mask,
                                   const double* __restrict__ x, long long stride,
                                   float* __restrict__ pb, float* __restrict__ pphi, double h,
                                   double k, double d) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  const double w = x[pid];
  const double z = x[stride + pid];
  // setup runs once; libm-grade atan2/sin from the device library (compared at float32 ulp)
  const double phi = atan2(d * (z - k), w - h);
  const double b = (z - k) / sin(phi);
  pphi[pid] = (float)phi;
  pb[pid] = (float)b;
}

__global__ void k_elliptical_push(int capacity, const unsigned char* __restrict__ mask,
                                  const int* __restrict__ slot_elem,
                                  const int* __restrict__ class_id, double* __restrict__ xt,
                                  long long stride, const float* __restrict__ pb,
                                  float* __restrict__ pphi, double h, double k, double d,
                                  double deg) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  double x, y, rad;
  ppm::elliptical_advance(class_id[slot_elem[pid]], pphi[pid], pb[pid], h, k, d, deg, x, y, rad);
  xt[pid] = x;
  xt[stride + pid] = y;
  pphi[pid] = (float)rad;
}

__global__ void k_toroidal_push(int capacity, const unsigned char* __restrict__ mask,
                                const int* __restrict__ slot_elem,
                                const int* __restrict__ class_id, const double* __restrict__ x0,
                                double* __restrict__ xt, long long stride,
                                const float* __restrict__ pb, float* __restrict__ pphi, double h,
                                double k, double d, double deg) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  double tx, ty, tz, rad;
  ppm::toroidal_advance(class_id[slot_elem[pid]], pphi[pid], pb[pid], x0[pid], x0[stride + pid], h,
                        k, d, deg, tx, ty, tz, rad);
  xt[pid] = tx;
  xt[stride + pid] = ty;
  xt[2 * stride + pid] = tz;
  pphi[pid] = (float)rad;
}

__global__ void k_linear_push(int capacity, const unsigned char* __restrict__ mask,
                              const double* __restrict__ x, double* __restrict__ xt,
                              long long stride, double distance, double dx, double dy, double dz) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  const double dir0 = distance * dx, dir1 = distance * dy, dir2 = distance * dz;
  xt[pid] = x[pid] + dir0 + 0.0;
  xt[stride + pid] = x[stride + pid] + dir1 + 0.0;
  xt[2 * stride + pid] = x[2 * stride + pid] + dir2 + 0.0;
}

__global__ void k_boris(int n, double* x, double* y, double* z, double* xp, double* yp, double* zp,
                        double* vx, double* vy, double* vz, const double* ex, const double* ey,
                        const double* ez, const double* br, const double* bt, const double* bz,
                        double dt) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= n) return;
  using namespace ppg;
  const V3 vel = ppm::boris_velocity(V3{vx[pid], vy[pid], vz[pid]}, V3{ex[pid], ey[pid], ez[pid]},
                                     V3{br[pid], bt[pid], bz[pid]}, dt);
  const double p0 = xp[pid], p1 = yp[pid], p2 = zp[pid];
  xp[pid] = x[pid];
  yp[pid] = y[pid];
  zp[pid] = z[pid];
  x[pid] = p0 + vel.x * dt;
  y[pid] = p1 + vel.y * dt;
  z[pid] = p2 + vel.z * dt;
  vx[pid] = vel.x;
  vy[pid] = vel.y;
  vz[pid] = vel.z;
}

__global__ void k_update_positions(int capacity, const int* __restrict__ slot_elem,
                                   double* __restrict__ x, double* __restrict__ xt,
                                   long long stride) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || slot_elem[pid] < 0) return;
  for (int i = 0; i < 3; ++i) {
    x[i * stride + pid] = xt[i * stride + pid];
    xt[i * stride + pid] = 0;
  }
}

__global__ void k_pseudo_push160(int capacity, const unsigned char* __restrict__ mask,
                                 const int* __restrict__ slot_elem, double* __restrict__ dbls,
                                 int* __restrict__ nums, long long* __restrict__ lint,
                                 long long stride, const double* __restrict__ ped) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= capacity) return;
  const int e = slot_elem[p];
  if (e < 0) return;
  // write-once streams (22 of them): non-temporal stores.  Live and padded lanes store through
  // the SAME instructions (values selected, not branched): two half-masked stores per stream would
  // leave the CU as partial lines, and partial lines are read-modify-writes at the HBM.
  const bool live = mask[p];
  double v = 10.3;
  v = v * v * v / sqrt((double)p) / sqrt((double)e) + ped[e];
  v = live ? v : 0.0;
  for (int i = 0; i < 17; ++i) __builtin_nontemporal_store(v, dbls + i * stride + p);
  for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(live ? 4 * p + i : -1, nums + i * stride + p);
  __builtin_nontemporal_store(live ? (long long)p : 0ll, lint + p);
}

int check_member(const pp_ps* ps, int m, int bytes, int ncomp, const char* what) {
  if (int rc = pp::ps_ready(ps)) return rc;  // a member that is only logically zero gets its zeros now
  if (m < 0 || m >= ps->nmembers) {
    pp::set_error(std::string(what) + ": member index out of range");
    return PP_EINVAL;
  }
  const int s = ps->member_map[m];
  if (ps->member_bytes[s] != bytes || ps->member_ncomp[s] < ncomp) {
    pp::set_error(std::string(what) + ": member has the wrong type for this operator");
    return PP_EINVAL;
  }
  return PP_OK;
}
#define PP_MEMBER(ps, m, T) ((T*)(ps)->data[(ps)->member_map[m]].p)
}  // namespace

extern "C" {

int pp_elliptical_setup(pp_ps* ps, int m_x, int m_b, int m_phi, double h, double k, double d) {
  PP_REQUIRE(ps, "pp_elliptical_setup: null ps");
  int rc;
  if ((rc = check_member(ps, m_x, 8, 2, "pp_elliptical_setup x"))) return rc;
  if ((rc = check_member(ps, m_b, 4, 1, "pp_elliptical_setup b"))) return rc;
  if ((rc = check_member(ps, m_phi, 4, 1, "pp_elliptical_setup phi"))) return rc;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  k_elliptical_setup<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), PP_MEMBER(ps, m_x, double), ps->stride,
      PP_MEMBER(ps, m_b, float), PP_MEMBER(ps, m_phi, float), h, k, d);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_elliptical_push(pp_ps* ps, const pp_mesh* mesh, int m_xtgt, int m_b, int m_phi, double h,
                       double k, double d, double deg) {
  pp::Range rg_("ellipticalPush");
  PP_REQUIRE(ps && mesh, "pp_elliptical_push: null argument");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_elliptical_push: structure/mesh element mismatch");
  int rc;
  if ((rc = check_member(ps, m_xtgt, 8, 2, "pp_elliptical_push x_tgt"))) return rc;
  if ((rc = check_member(ps, m_b, 4, 1, "pp_elliptical_push b"))) return rc;
  if ((rc = check_member(ps, m_phi, 4, 1, "pp_elliptical_push phi"))) return rc;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  k_elliptical_push<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps),
      mesh->d_class_id.as<int>(), PP_MEMBER(ps, m_xtgt, double), ps->stride,
      PP_MEMBER(ps, m_b, float), PP_MEMBER(ps, m_phi, float), h, k, d, deg);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_toroidal_push(pp_ps* ps, const pp_mesh* mesh, int m_x, int m_xtgt, int m_b, int m_phi,
                     double h, double k, double d, double deg) {
  PP_REQUIRE(ps && mesh, "pp_toroidal_push: null argument");
  PP_REQUIRE(ps->num_elems == mesh->nelems, "pp_toroidal_push: structure/mesh element mismatch");
  int rc;
  if ((rc = check_member(ps, m_x, 8, 3, "pp_toroidal_push x"))) return rc;
  if ((rc = check_member(ps, m_xtgt, 8, 3, "pp_toroidal_push x_tgt"))) return rc;
  if ((rc = check_member(ps, m_b, 4, 1, "pp_toroidal_push b"))) return rc;
  if ((rc = check_member(ps, m_phi, 4, 1, "pp_toroidal_push phi"))) return rc;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  k_toroidal_push<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps),
      mesh->d_class_id.as<int>(), PP_MEMBER(ps, m_x, double), PP_MEMBER(ps, m_xtgt, double),
      ps->stride, PP_MEMBER(ps, m_b, float), PP_MEMBER(ps, m_phi, float), h, k, d, deg);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_linear_push(pp_ps* ps, int m_x, int m_xtgt, double distance, double dx, double dy,
                   double dz) {
  PP_REQUIRE(ps, "pp_linear_push: null ps");
  int rc;
  if ((rc = check_member(ps, m_x, 8, 3, "pp_linear_push x"))) return rc;
  if ((rc = check_member(ps, m_xtgt, 8, 3, "pp_linear_push x_tgt"))) return rc;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  k_linear_push<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), PP_MEMBER(ps, m_x, double),
      PP_MEMBER(ps, m_xtgt, double), ps->stride, distance, dx, dy, dz);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_push_boris(int n, double* x, double* y, double* z, double* xp, double* yp, double* zp,
                  double* vx, double* vy, double* vz, const double* ex, const double* ey,
                  const double* ez, const double* br, const double* bt, const double* bz,
                  double dt) {
  PP_REQUIRE(n >= 0 && dt > 0, "pp_push_boris: n >= 0 and dt > 0 required (push.hpp:37)");
  if (n == 0) return PP_OK;
  k_boris<<<grid_for(n), kBlock, 0, pp::stream()>>>(n, x, y, z, xp, yp, zp, vx, vy, vz, ex, ey, ez,
                                                    br, bt, bz, dt);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_update_positions(pp_ps* ps, int m_x, int m_xtgt) {
  PP_REQUIRE(ps, "pp_update_positions: null ps");
  int rc;
  if ((rc = check_member(ps, m_x, 8, 3, "pp_update_positions x"))) return rc;
  if ((rc = check_member(ps, m_xtgt, 8, 3, "pp_update_positions x_tgt"))) return rc;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  k_update_positions<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, pp::slot_elem(ps), PP_MEMBER(ps, m_x, double),
      PP_MEMBER(ps, m_xtgt, double), ps->stride);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_pseudo_push160(pp_ps* ps, const double* parent_elm_data_dev) {
  PP_REQUIRE(ps && parent_elm_data_dev, "pp_pseudo_push160: null argument");
  int rc;
  // The pass writes every component of every member of every slot it iterates.  When the particles still sit in the
  // records of the last rebuild (pp_ps::lazy_rec == 3) the pass that would copy them into the member arrays first is
  // pointless: the records are simply given up.
  if (ps->lazy_rec == 3 && ps->zero_pending < 0 && ps->nmembers == 3 && ps->member_bytes[0] == 8 &&
      ps->member_ncomp[0] == 17 && ps->member_bytes[1] == 4 && ps->member_ncomp[1] == 4 && ps->member_bytes[2] == 8 &&
      ps->member_ncomp[2] == 1 && ps->member_map[0] == 0 && ps->member_map[1] == 1 && ps->member_map[2] == 2)
    ps->lazy_rec = 0;
  if ((rc = check_member(ps, 0, 8, 17, "pp_pseudo_push160 dbls"))) return rc;
  if ((rc = check_member(ps, 1, 4, 4, "pp_pseudo_push160 nums"))) return rc;
  if ((rc = check_member(ps, 2, 8, 1, "pp_pseudo_push160 lint"))) return rc;
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  k_pseudo_push160<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps),
      PP_MEMBER(ps, 0, double), PP_MEMBER(ps, 1, int), PP_MEMBER(ps, 2, long long), ps->stride,
      parent_elm_data_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // extern "C"
