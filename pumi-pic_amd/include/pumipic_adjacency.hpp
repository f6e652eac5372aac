// pumipic_adjacency.hpp -- C++ host mirror of the search / push / scatter operators over the C-ABI.
//
//   search_mesh (new, 2-D/3-D)   src/pumipic_adjacency.hpp:37-45, adjacency.tpp:641-654
//   search_mesh_2d               src/pumipic_adjacency.hpp:1011-1020
//   search_mesh (legacy 3-D)     src/pumipic_adjacency.hpp:558-562
//   search_mesh_3d               src/pumipic_adjacency.hpp:314-324
//   trace_particle_through_mesh  src/pumipic_adjacency.tpp:460-615 (functor hook), :617-639
//   migrate_ptcls / migrate_lb_ptcls   src/pumipic_ptcl_ops.hpp:53-85 (single rank: rebuild)
//   RecordTime / SummarizeTime   support/ppTiming.hpp:34-75
//   gather-side interpolation    src/pumipic_adjacency.hpp:772-809, src/pumipic_utils.hpp:186-454
//                                (pumipic_gather.hpp, raw-pointer signatures)
// Omega_h::Mesh is replaced by pumipic::Mesh (a handle that owns the derived adjacency and the
// packed walk records on the device); Omega_h::Write<T>/Read<T> by pumipic::View<T>.
#pragma once
#include <chrono>
#include <map>
#include "particle_structs.hpp"
#include "pumipic_wall.hpp"    // closest_point_on_triangle[_wnormal] (device-inline)
#include "pumipic_gather.hpp"  // interpolateTetVtx, interpolate2dField, ... (device-inline)

namespace Omega_h {
typedef int LO;
typedef double Real;
typedef int ClassId;
template <class T>
using Write = pumipic::View<T>;
template <class T>
using Read = pumipic::View<T>;
typedef Read<LO> LOs;
typedef Read<Real> Reals;
}  // namespace Omega_h
namespace o = Omega_h;

namespace pumipic {

typedef double fp_t;
typedef fp_t Vector3d[3];

class Mesh {
 public:
  Mesh(int dim, const std::vector<double>& coords, const std::vector<int>& elem2verts,
       const std::vector<int>& class_id) {
    h_ = pp_mesh_create(dim, (int)(coords.size() / dim), coords.data(),
                        (int)(elem2verts.size() / (dim + 1)), elem2verts.data(),
                        class_id.empty() ? nullptr : class_id.data());
    if (!h_) pp_check(PP_EHIP, "pumipic::Mesh");
    pp_check(pp_mesh_info(h_, &dim_, &nverts_, &nelems_, &nsides_), "pp_mesh_info");
  }
  ~Mesh() {
    if (h_) (void)pp_mesh_destroy(h_);
  }
  Mesh(const Mesh&) = delete;
  Mesh& operator=(const Mesh&) = delete;
  int dim() const { return dim_; }
  int nverts() const { return nverts_; }
  int nelems() const { return nelems_; }
  int nsides() const { return nsides_; }
  pp_mesh* handle() const { return h_; }
  // read-only device views for user kernels (Omega_h: ask_elem_verts(), coords(),
  // get_array<ClassId>(dim,"class_id"), ask_up(0,dim), measure_elements_real ...)
  o::LOs ask_elem_verts() const { return view<int>(PP_MESH_ELEM2VERTS); }
  o::Reals coords() const { return view<double>(PP_MESH_COORDS); }
  o::LOs class_ids() const { return view<int>(PP_MESH_CLASS_ID); }
  o::LOs verts2elems_offsets() const { return view<int>(PP_MESH_VERT2ELEMS_OFF); }
  o::LOs verts2elems() const { return view<int>(PP_MESH_VERT2ELEMS); }
  o::Reals elem_measures() const { return view<double>(PP_MESH_ELEM_MEASURE); }
  // ask_up(dim-1, dim) {a2ab, ab2b}, mark_exposed_sides, ask_down(dim, dim-1).ab2b
  o::LOs sides2elems_offsets() const { return view<int>(PP_MESH_SIDE2ELEMS_OFF); }
  o::LOs sides2elems() const { return view<int>(PP_MESH_SIDE2ELEMS); }
  View<signed char> side_is_exposed() const { return view<signed char>(PP_MESH_SIDE_EXPOSED); }
  o::LOs elems2sides() const { return view<int>(PP_MESH_ELEM2SIDES); }
  double tolerance() const { return pp_mesh_tolerance(h_); }

 private:
  template <class T>
  View<T> view(int which) const {
    size_t n = 0;
    const void* p = pp_mesh_array_dev(h_, which, &n);
    return View<T>::wrap((T*)p, n);
  }
  pp_mesh* h_ = nullptr;
  int dim_ = 0, nverts_ = 0, nelems_ = 0, nsides_ = 0;
};

// ---------------------------------------------------------------- timing (ppTiming.hpp)
inline std::map<std::string, std::pair<double, int>>& timing_table() {
  static std::map<std::string, std::pair<double, int>> t;
  return t;
}
inline void RecordTime(const std::string& name, double seconds, double = 0) {
  auto& e = timing_table()[name];
  e.first += seconds;
  e.second += 1;
}
inline void SummarizeTime() {
  fprintf(stderr, "%-40s %14s %8s %14s\n", "Timing", "Total(s)", "Calls", "Avg(s)");
  for (auto& kv : timing_table())
    fprintf(stderr, "%-40s %14.6f %8d %14.6f\n", kv.first.c_str(), kv.second.first, kv.second.second,
            kv.second.first / kv.second.second);
}
struct Timer {  // Kokkos::Timer stand-in; seconds() drains the stream first, like Kokkos::fence
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void reset() { t0 = std::chrono::steady_clock::now(); }
  double seconds() const {
    (void)pp_sync();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
};

// ---------------------------------------------------------------- searches
template <class ParticleStruct, typename CurrentCoordView, typename TargetCoordView, typename SegmentInt>
bool search_mesh_2d(Mesh& mesh, ParticleStruct* ptcls, CurrentCoordView x_ps_d, TargetCoordView xtgt_ps_d,
                    SegmentInt pid_d, o::Write<o::LO> elem_ids, int looplimit = 0, bool = false) {
  Timer timer;
  int found = 1;
  pp_check(pp_search_mesh_2d(mesh.handle(), ptcls->handle(), x_ps_d.member(), xtgt_ps_d.member(),
                             pid_d.member(), elem_ids.data(), looplimit, &found),
           "search_mesh_2d");
  RecordTime("pumipic search_2d", timer.seconds());
  return found != 0;
}

template <class ParticleType, typename Segment3d, typename SegmentInt>
bool search_mesh(Mesh& mesh, ParticleStructure<ParticleType>* ptcls, Segment3d x_ps_orig,
                 Segment3d x_ps_tgt, SegmentInt pids, o::Write<o::LO>& elem_ids,
                 bool requireIntersection, o::Write<o::LO>& inter_faces,
                 o::Write<o::Real>& inter_points, int looplimit = 0, int = 0) {
  Timer timer;
  const size_t cap = (size_t)ptcls->capacity();
  int seeded = 1;
  if (elem_ids.size() == 0) {  // adjacency.tpp:504-515
    elem_ids = o::Write<o::LO>(cap, -1);
    seeded = 0;
  }
  if (requireIntersection && (inter_points.size() == 0 || inter_faces.size() == 0)) {
    inter_points = o::Write<o::Real>((size_t)mesh.dim() * cap, 0);
    inter_faces = o::Write<o::LO>(cap, -1);
  }
  int found = 1, notin = 0;
  pp_check(pp_search_mesh(mesh.handle(), ptcls->handle(), x_ps_orig.member(), x_ps_tgt.member(),
                          pids.member(), elem_ids.data(), seeded, requireIntersection ? 1 : 0,
                          inter_faces.data(), inter_points.data(), looplimit, &found, &notin),
           "search_mesh");
  RecordTime("pumipic search_mesh", timer.seconds());
  return found != 0;
}

// trace_particle_through_mesh with a caller-supplied functor (adjacency.tpp:460-615).  `func` is
// called once per walk iteration between find_exit_face and set_new_element (tpp:561-565) with
//   (mesh, ptcls, elem_ids, inter_faces, lastExit, inter_points, ptcl_done, x_ps_orig, x_ps_tgt)
// and may run any device code (ps::parallel_for lambdas) on those arrays.  The walk runs kernel by
// kernel through the pp_trace_* entry points; search_mesh above is the fused form of the same walk
// with RemoveParticleOnGeometricModelExit.
template <class ParticleType, typename Segment3d, typename SegmentInt, typename Func>
bool trace_particle_through_mesh(Mesh& mesh, ParticleStructure<ParticleType>* ptcls,
                                 Segment3d x_ps_orig, Segment3d x_ps_tgt, SegmentInt pids,
                                 o::Write<o::LO>& elem_ids, bool requireIntersection,
                                 o::Write<o::LO>& inter_faces, o::Write<o::Real>& inter_points,
                                 int looplimit, bool /*debug*/, Func& func) {
  Timer timer;
  (void)pids;
  const size_t cap = (size_t)ptcls->capacity();
  o::Write<o::LO> ptcl_done(cap, 0);   // tpp:486
  o::Write<o::LO> lastExit(cap, -1);   // tpp:488
  int seeded = 1;
  if (elem_ids.size() == 0) {  // tpp:504-515
    elem_ids = o::Write<o::LO>(cap, -1);
    seeded = 0;
  }
  if (requireIntersection && (inter_points.size() == 0 || inter_faces.size() == 0)) {
    inter_points = o::Write<o::Real>((size_t)mesh.dim() * cap, 0);
    inter_faces = o::Write<o::LO>(cap, -1);
  }
  int notin = 0;
  pp_check(pp_trace_begin(mesh.handle(), ptcls->handle(), x_ps_orig.member(), x_ps_tgt.member(),
                          elem_ids.data(), seeded, requireIntersection ? 1 : 0, inter_faces.data(),
                          inter_points.data(), ptcl_done.data(), lastExit.data(), &notin),
           "trace_particle_through_mesh: begin");
  bool found = false;
  int loops = 0;
  while (!found) {
    pp_check(pp_trace_find_exit_face(mesh.handle(), ptcls->handle(), x_ps_orig.member(),
                                     x_ps_tgt.member(), elem_ids.data(), ptcl_done.data(),
                                     lastExit.data(), inter_points.data(), requireIntersection ? 0 : 1),
             "trace_particle_through_mesh: find_exit_face");
    func(mesh, ptcls, elem_ids, inter_faces, lastExit, inter_points, ptcl_done, x_ps_orig, x_ps_tgt);
    int left = 0;
    pp_check(pp_trace_set_new_element(mesh.handle(), ptcls->handle(), elem_ids.data(),
                                      ptcl_done.data(), lastExit.data(), &left),
             "trace_particle_through_mesh: set_new_element");
    found = (left == 0);
    ++loops;
    if (looplimit && loops >= looplimit) {  // tpp:583-606
      int nf = 0;
      pp_check(pp_trace_not_found(ptcls->handle(), elem_ids.data(), ptcl_done.data(), &nf),
               "trace_particle_through_mesh: not found");
      fprintf(stderr, "ERROR: loop limit %d exceeded. %d particles were not found. Deleting them...\n",
              looplimit, nf);
      break;
    }
  }
  RecordTime("pumipic search_mesh", timer.seconds());
  return found;
}

// the default functor (adjacency.tpp:617-639)
template <typename ParticleType, typename Segment3d>
struct RemoveParticleOnGeometricModelExit {
  RemoveParticleOnGeometricModelExit(Mesh&, bool requireIntersection)
      : requireIntersection_(requireIntersection) {}
  void operator()(Mesh& mesh, ParticleStructure<ParticleType>* ptcls, o::Write<o::LO>& elem_ids,
                  o::Write<o::LO>& inter_faces, o::Write<o::LO>& lastExit, o::Write<o::Real>&,
                  o::Write<o::LO>& ptcl_done, Segment3d, Segment3d) const {
    pp_check(pp_trace_check_model_intersection(mesh.handle(), ptcls->handle(), elem_ids.data(),
                                               ptcl_done.data(), lastExit.data(),
                                               requireIntersection_ ? 1 : 0, inter_faces.data()),
             "check_model_intersection");
  }

 private:
  bool requireIntersection_;
};

// legacy 3-D overload: chosen when argument 7 is a Write<Real> (adjacency.hpp:558-562)
template <class ParticleType, typename Segment3d, typename SegmentInt>
bool search_mesh(Mesh& mesh, ParticleStructure<ParticleType>* ptcls, Segment3d x_ps_d,
                 Segment3d xtgt_ps_d, SegmentInt pid_d, o::Write<o::LO>& elem_ids,
                 o::Write<o::Real>& xpoints_d, o::Write<o::LO>& xface_d, int looplimit = 0, int = 0) {
  const size_t cap = (size_t)ptcls->capacity();
  int seeded = 1;
  if (elem_ids.size() == 0) {
    elem_ids = o::Write<o::LO>(cap);
    seeded = 0;
  }
  int found = 1;
  pp_check(pp_search_mesh_legacy3d(mesh.handle(), ptcls->handle(), x_ps_d.member(),
                                   xtgt_ps_d.member(), pid_d.member(), elem_ids.data(), seeded,
                                   xpoints_d.data(), xface_d.data(), looplimit, &found),
           "search_mesh (legacy)");
  if (found == -2) {  // OMEGA_H_CHECK(false), adjacency.hpp:622-626
    fprintf(stderr, "Warning: Particle not in this element at loops=0\n");
    abort();
  }
  return found == 1;
}

// search_mesh_3d (adjacency.hpp:314-324)
template <class ParticleStruct, typename CurrentCoordView, typename TargetCoordView, typename SegmentInt>
bool search_mesh_3d(Mesh& mesh, ParticleStruct* ptcls, CurrentCoordView x_ps_d,
                    TargetCoordView xtgt_ps_d, SegmentInt pid_d, o::Write<o::LO>& elem_ids,
                    o::Write<o::Real>& xpoints_d, o::Write<o::LO>& xface_d, int looplimit = 0,
                    int = 0) {
  Timer timer;
  int seeded = 1;
  if (elem_ids.size() == 0) {
    elem_ids = o::Write<o::LO>((size_t)ptcls->capacity());
    seeded = 0;
  }
  int found = 1;
  pp_check(pp_search_mesh_3d(mesh.handle(), ptcls->handle(), x_ps_d.member(), xtgt_ps_d.member(),
                             pid_d.member(), elem_ids.data(), seeded, xpoints_d.data(),
                             xface_d.data(), looplimit, &found),
           "search_mesh_3d");
  if (found == -2) abort();  // OMEGA_H_CHECK(false), adjacency.hpp:373-379
  RecordTime("Search Mesh 3d", timer.seconds());
  return found == 1;
}

// single-rank form of migrate_lb_ptcls / migrate_ptcls (pumipic_ptcl_ops.hpp:53-85):
// setUnsafeProcs leaves every particle on this rank, the balancer returns immediately
// (pumipic_lb.hpp:353-358), migrate() falls through to rebuild().
template <class PS>
void migrate_lb_ptcls(Mesh&, PS* ptcls, o::LOs elems, float /*tol*/, float = 0.5) {
  Timer t;
  ptcls->rebuild(elems);
  RecordTime("migration", t.seconds());
}
template <class PS>
void migrate_ptcls(Mesh& m, PS* ptcls, o::LOs elems) {
  migrate_lb_ptcls(m, ptcls, elems, 1.0f);
}

}  // namespace pumipic
namespace p = pumipic;
namespace ps = particle_structs;
