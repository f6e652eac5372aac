// pumipic_utils.hpp -- the reference's DEVICE helpers of the search, under their names and argument lists
// (src/pumipic_adjacency.hpp:75-290, src/pumipic_adjacency.tpp:23-218,419-430, src/pumipic_utils.hpp:78-149,489-507),
// for user lambdas and for the reference's own tests (test/test_adj.cpp, test/moller_trumbore_line_tri_test.cpp,
// src/unit_tests.hpp).  Every function is an ADAPTER: it unpacks the Omega_h-style arguments and calls the function of
// pumipic_geom.hpp that the library's search kernels call, so a user's check and the library's walk evaluate the same
// expression tree (compile user code with -ffp-contract=off, as the library is, for bit-equal results).
#pragma once
#include "pumipic_geom.hpp"
#include "compat/Omega_h_mesh.hpp"

namespace pumipic {
namespace geom_detail {
PPD ppg::V3 v3(const o::Vector<3>& a) { return {a[0], a[1], a[2]}; }
PPD ppg::V2 v2(const o::Vector<2>& a) { return {a[0], a[1]}; }
PPD o::Vector<3> ov(ppg::V3 a) {
  o::Vector<3> r;
  r[0] = a.x;
  r[1] = a.y;
  r[2] = a.z;
  return r;
}
}  // namespace geom_detail

constexpr double EPSILON = ppg::kEpsilon;  // src/pumipic_constants.hpp:6

// ---- particle position of a slot (adjacency.hpp:275-289)
template <typename Segment>
OMEGA_H_DEVICE o::Vector<3> makeVector3(int pid, Segment xyz) {
  o::Vector<3> v;
  for (int i = 0; i < 3; ++i) v[i] = xyz(pid, i);
  return v;
}
template <typename Segment>
OMEGA_H_DEVICE o::Vector<2> makeVector2(int pid, Segment xyz) {
  o::Vector<2> v;
  for (int i = 0; i < 2; ++i) v[i] = xyz(pid, i);
  return v;
}
OMEGA_H_DEVICE o::Vector<3> makeVector3FromArray(const o::Real (&arr)[3]) {
  o::Vector<3> v;
  for (int i = 0; i < 3; ++i) v[i] = arr[i];
  return v;
}

// ---- sign tests and arg-min / arg-max (utils.hpp:78-149)
template <class Vec>
OMEGA_H_DEVICE bool all_positive(const Vec a, o::Real tol = EPSILON) {
  bool p = true;
  for (int i = 0; i < a.size(); ++i) p = p & ppg::gtez(a[i], tol);
  return p;
}
OMEGA_H_DEVICE o::LO min3(o::Vector<3> a) { return ppg::min3(a.data()); }
template <class T>
OMEGA_H_DEVICE o::LO min_index(const T& a, o::LO n, o::LO beg = 0) {
  o::LO ind = beg;
  auto mn = a[beg];
  for (o::LO i = beg; i < n - 1; ++i)
    if (mn > a[i + 1]) {
      mn = a[i + 1];
      ind = i + 1;
    }
  return ind;
}
template <class T>
OMEGA_H_DEVICE o::LO max_index(const T& a, o::LO n, o::LO beg = 0) {
  o::LO ind = beg;
  auto mx = a[beg];
  for (o::LO i = beg; i < n - 1; ++i)
    if (mx < a[i + 1]) {
      mx = a[i + 1];
      ind = i + 1;
    }
  return ind;
}

// ---- side orientation (utils.hpp:489-507)
OMEGA_H_DEVICE o::LO getFaceMap(const o::LO i) { return ppg::face_map(i); }
OMEGA_H_DEVICE bool isFaceFlipped(const o::LO, const o::Few<o::LO, 2>& ev2v, const o::Few<o::LO, 3>& facev2v) {
  return ppg::is_edge_flipped(ev2v.data(), facev2v.data());
}
OMEGA_H_DEVICE bool isFaceFlipped(const o::LO fi, const o::Few<o::LO, 3>& fv2v, const o::Few<o::LO, 4>& tetv2v) {
  return ppg::is_face_flipped(fi, fv2v.data(), tetv2v.data());
}

// ---- barycentric coordinates (adjacency.tpp:23-69, adjacency.hpp:75-133,163-183): one per SIDE of the element
// (the coordinate of side i belongs to the vertex opposite to it)
OMEGA_H_DEVICE void barycentric_tri(const o::Real parentArea, const o::Matrix<2, 3>& faceCoords, const o::Vector<2>& pos,
                                    o::Vector<3>& bcc) {
  const ppg::V2 fc[3] = {geom_detail::v2(faceCoords[0]), geom_detail::v2(faceCoords[1]), geom_detail::v2(faceCoords[2])};
  ppg::barycentric_tri(parentArea, fc, geom_detail::v2(pos), bcc.data());
}
OMEGA_H_DEVICE void barycentric_tri(const o::Reals triArea, const o::Matrix<2, 3>& faceCoords, const o::Vector<2>& pos,
                                    o::Vector<3>& bcc, const int searchElm, const bool vertex_major = false) {
  o::Vector<3> e;
  barycentric_tri(triArea[searchElm], faceCoords, pos, e);
  for (int i = 0; i < 3; ++i) bcc[i] = e[(i + (vertex_major ? 1 : 0)) % 3];
}
OMEGA_H_DEVICE bool barycentric_tet(const o::Real parentVol, const o::Matrix<3, 4>& mat, const o::Vector<3>& pos,
                                    o::Vector<4>& bcc) {
  const ppg::V3 M[4] = {geom_detail::v3(mat[0]), geom_detail::v3(mat[1]), geom_detail::v3(mat[2]), geom_detail::v3(mat[3])};
  return ppg::barycentric_tet(parentVol, M, geom_detail::v3(pos), bcc.data());
}
OMEGA_H_DEVICE bool find_barycentric_tet(const o::Matrix<3, 4>& mat, const o::Vector<3>& pos, o::Vector<4>& bcc,
                                         bool = false) {
  const ppg::V3 M[4] = {geom_detail::v3(mat[0]), geom_detail::v3(mat[1]), geom_detail::v3(mat[2]), geom_detail::v3(mat[3])};
  return ppg::find_barycentric_tet(M, geom_detail::v3(pos), bcc.data());
}
OMEGA_H_DEVICE bool find_barycentric_tri_simple(const o::Few<o::Vector<3>, 3>& abc, const o::Vector<3>& xpoint,
                                                o::Vector<3>& bc) {
  const ppg::V3 t[3] = {geom_detail::v3(abc[0]), geom_detail::v3(abc[1]), geom_detail::v3(abc[2])};
  return ppg::find_barycentric_tri_simple(t, geom_detail::v3(xpoint), bc.data());
}

// ---- Moller-Trumbore (adjacency.tpp:152-202) and the 2-D segment / edge test (:204-218)
OMEGA_H_DEVICE bool ray_intersects_triangle(const o::Few<o::Vector<3>, 3>& faceVerts, const o::Vector<3>& orig,
                                            const o::Vector<3>& dest, o::Vector<3>& xpoint, const o::Real tol,
                                            const o::LO flip, o::Real& dproj, o::Real& closeness,
                                            o::Real& intersection_parametric_coord) {
  const ppg::V3 fv[3] = {geom_detail::v3(faceVerts[0]), geom_detail::v3(faceVerts[1]), geom_detail::v3(faceVerts[2])};
  ppg::V3 xp;
  const bool hit = ppg::ray_intersects_triangle(fv, geom_detail::v3(orig), geom_detail::v3(dest), xp, tol, flip, dproj,
                                                closeness, intersection_parametric_coord);
  xpoint = geom_detail::ov(xp);
  return hit;
}
OMEGA_H_DEVICE bool moller_trumbore_line_triangle(const o::Few<o::Vector<3>, 3>& faceVerts, const o::Vector<3>& orig,
                                                  const o::Vector<3>& dest, o::Vector<3>& xpoint, const o::Real tol,
                                                  const o::LO flip, o::Real& dproj, o::Real& closeness) {
  o::Real t;
  return ray_intersects_triangle(faceVerts, orig, dest, xpoint, tol, flip, dproj, closeness, t);
}
OMEGA_H_DEVICE bool line_segment_intersects_triangle(const o::Few<o::Vector<3>, 3>& faceVerts, const o::Vector<3>& orig,
                                                     const o::Vector<3>& dest, o::Vector<3>& xpoint, const o::Real tol,
                                                     const o::LO flip, o::Real& dproj, o::Real& closeness,
                                                     o::Real& intersection_parametric_coord) {
  const bool ray = ray_intersects_triangle(faceVerts, orig, dest, xpoint, tol, flip, dproj, closeness,
                                           intersection_parametric_coord);
  return ray && intersection_parametric_coord <= 1 + tol;
}
OMEGA_H_DEVICE bool line_edge_2d(const o::Few<o::Vector<2>, 2>& edgeVerts, const o::Vector<2>& orig,
                                 const o::Vector<2>& dest, o::Vector<2>& xpoint, o::Real tol, o::LO flip) {
  const ppg::V2 ev[2] = {geom_detail::v2(edgeVerts[0]), geom_detail::v2(edgeVerts[1])};
  ppg::V2 xp;
  const bool hit = ppg::line_edge_2d(ev, geom_detail::v2(orig), geom_detail::v2(dest), xp, tol, flip);
  xpoint[0] = xp.x;
  xpoint[1] = xp.y;
  return hit;
}

// ---- the search tolerance of a mesh (adjacency.tpp:418-430): max(1e-15 / smallest element measure, 1e-8)
template <typename Array>
inline o::Real compute_tolerance_from_area(Array elmArea) {
  const o::Real min_area = o::get_min(o::Reals(elmArea));  // (host fold of the copy: set-up code)
  const o::Real tol = 1e-15 / min_area > 1e-8 ? 1e-15 / min_area : 1e-8;
  printInfo("Min area is: %.15f, Planned tol is %.15f\n", min_area, tol);
  return tol;
}

// ---- check_initial_parents (adjacency.tpp:71-148): every live, unfinished particle must lie in elem_ids[slot];
// those that do not are counted, their element set to -1 and their done flag raised.  (The walk kernels make this
// test themselves, fused; this is the reference's stand-alone form, which its tests call as the judge of a search.)
template <class ParticleType, typename Segment3d, typename SegmentInt>
inline o::LO check_initial_parents(o::Mesh mesh, ParticleStructure<ParticleType>* ptcls, Segment3d x_ps_orig,
                                   SegmentInt pids, o::Write<o::LO> elem_ids, o::Write<o::LO> ptcl_done,
                                   o::Reals elmArea, const o::Real& tol_, bool debug = false) {
  const int dim = mesh.dim();
  const o::LOs elm2verts = mesh.ask_elem_verts();
  const o::Reals coords = mesh.coords();
  const int rank = mesh.comm()->rank();
  const o::Real tol = tol_;
  o::Write<o::LO> numNotInElem(1, 0, "search_numNotInElem");
  auto checkParent = PS_LAMBDA(const int e, const int pid, const int mask) {
    if (mask > 0 && !ptcl_done[pid]) {
      const o::LO searchElm = elem_ids[pid];
      OMEGA_H_CHECK(searchElm >= 0);
      bool inside;
      if (dim == 2) {
        const auto elmCoords = o::gather_vectors<3, 2>(coords, o::gather_verts<3>(elm2verts, searchElm));
        o::Vector<3> bcc;
        barycentric_tri(elmArea[searchElm], elmCoords, makeVector2(pid, x_ps_orig), bcc);
        inside = all_positive(bcc, tol);
        if (!inside && debug)
          printInfo("%d Particle not in element! ptcl %d: %d elem %d => %d bcc %.15f %.15f %.15f\n", rank, pid,
                    (int)pids(pid), e, searchElm, bcc[0], bcc[1], bcc[2]);
      } else {
        const auto elmCoords = o::gather_vectors<4, 3>(coords, o::gather_verts<4>(elm2verts, searchElm));
        o::Vector<4> bcc;
        barycentric_tet(elmArea[searchElm], elmCoords, makeVector3(pid, x_ps_orig), bcc);
        inside = all_positive(bcc, tol);
        if (!inside && debug)
          printInfo("%d Particle not in element! ptcl %d: %d elem %d => %d bcc %.15f %.15f %.15f %.15f\n", rank, pid,
                    (int)pids(pid), e, searchElm, bcc[0], bcc[1], bcc[2], bcc[3]);
      }
      if (!inside) {
        Kokkos::atomic_add(&(numNotInElem[0]), 1);
        elem_ids[pid] = -1;
        ptcl_done[pid] = 1;
      }
    }
  };
  parallel_for(ptcls, checkParent, "search_checkParent");
  const o::HostWrite<o::LO> numNotInElem_h(numNotInElem);
  if (numNotInElem_h[0] > 0)
    printError("[WARNING] Rank %d: %d particles are not located in their starting elements. Deleting them...\n", rank,
               numNotInElem_h[0]);
  return numNotInElem_h[0];
}
}  // namespace pumipic
#include "pumipic_profiling.hpp"
