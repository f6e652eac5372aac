// gyroScatter.hpp -- test/gyroScatter.hpp:1-258 on the particle_structs mirror: the same free functions
// with the same arguments (mesh, particle structure, ring map, tag names), so the step loop of
// pseudoXGCm reads as in the reference.  The ring maps and the two-stage scatter are library calls
// (pp_create_gyro_ring_mappings, pp_gyro_scatter); the reduction over ranks is Mesh::reduceCommArray.
#ifndef GYRO_SCATTER_H
#define GYRO_SCATTER_H
#include <cstdio>
#include <string>
#include "pseudoXGCmTypes.hpp"

namespace {
o::Real gyro_rmax = 0.038;  // max ring radius
o::LO gyro_num_rings = 3;
o::LO gyro_points_per_ring = 8;
o::Real gyro_theta = 0;
}  // namespace

inline void setGyroConfig(o::Real rmax, o::LO nrings, o::LO pointsPerRing, o::Real theta) {
  gyro_rmax = rmax;
  gyro_num_rings = nrings;
  gyro_points_per_ring = pointsPerRing;
  gyro_theta = theta;
}

inline void printGyroConfig() {
  printf("gyro rmax num_rings points_per_ring theta %f %d %d %f\n", gyro_rmax, gyro_num_rings,
         gyro_points_per_ring, gyro_theta);
}

/* Build gyro-avg mapping (gyroScatter.hpp:101-166): ring points around every vertex, projected
   (identity, the reference's TODO), located by search_mesh_2d; 3 (triangles) / 4 (tets) mapped
   vertices per ring point, -1 outside the domain */
inline void createGyroRingMappings(o::Mesh* mesh, o::LOs& forward_map, o::LOs& backward_map) {
  pp_range_push("xgcm_createGyroRingMappings");
  const size_t n = (size_t)mesh->nverts() * gyro_num_rings * gyro_points_per_ring * (mesh->dim() + 1);
  o::Write<o::LO> fwd(n), bkwd(n);
  pumipic::pp_check(pp_create_gyro_ring_mappings(mesh->handle(), gyro_rmax, gyro_num_rings, gyro_points_per_ring,
                                                 gyro_theta, fwd.data(), bkwd.data()),
                    "createGyroRingMappings");
  forward_map = fwd;
  backward_map = bkwd;
  pp_range_pop();
}

/* gyroScatter.hpp:168-229: accumulate every particle to the rings of its element's vertices, scatter the
   rings to the mapped vertices, store the result as the vertex tag `scatterTagName` */
inline void gyroScatter(o::Mesh* mesh, PS* ptcls, o::LOs v2v, std::string scatterTagName) {
  const auto btime = pumipic::pumipic_prebarrier();
  pumipic::Timer timer;
  pp_range_push("xgcm_gyroScatter");
  o::Write<o::Real> scatter_w((size_t)mesh->nverts());
  pumipic::pp_check(pp_gyro_scatter(mesh->handle(), ptcls->handle(), v2v.data(), gyro_rmax, gyro_num_rings,
                                    gyro_points_per_ring, scatter_w.data()),
                    "gyroScatter");
  mesh->set_tag(o::VERT, scatterTagName, o::Reals(scatter_w));
  pumipic::RecordTime("gyro scatter", timer.seconds(), btime);
  pp_range_pop();
}

/* gyroScatter.hpp:231-258: interleave the two fields and SUM them over the ranks */
inline void gyroSync(p::Mesh& picparts, const std::string& fwdTagName, const std::string& bkwdTagName,
                     const std::string& syncTagName) {
  const auto btime = pumipic::pumipic_prebarrier();
  pumipic::Timer timer;
  pp_range_push("xgcm_gyroSync");
  Omega_h::Write<Omega_h::Real> sync_array = picparts.createCommArray(0, 2, Omega_h::Real(0.0));
  Omega_h::Mesh* mesh = picparts.mesh();
  Omega_h::Read<Omega_h::Real> fwdTag = mesh->get_array<Omega_h::Real>(0, fwdTagName);
  Omega_h::Read<Omega_h::Real> bkwdTag = mesh->get_array<Omega_h::Real>(0, bkwdTagName);
  pumipic::pp_check(pp_gyro_sync_pack(mesh->nverts(), fwdTag.data(), bkwdTag.data(), sync_array.data()),
                    "setSyncArray");
  pumipic::Timer reducetimer;
  picparts.reduceCommArray(0, p::Mesh::Op::SUM_OP, sync_array);
  const auto rtime = reducetimer.seconds();
  mesh->set_tag(0, syncTagName, Omega_h::Reals(sync_array));
  pumipic::RecordTime("gyro sync", timer.seconds(), btime);
  pumipic::RecordTime("gyro reduction", rtime);
  pp_range_pop();
}
#endif
