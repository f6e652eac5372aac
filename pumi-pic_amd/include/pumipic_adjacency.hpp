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
#include <algorithm>
#include <fstream>
#include <stdexcept>
#include <chrono>
#include <map>
#define PP_ADJACENCY_IN_PROGRESS  // (particle_structs.hpp leaves the Kokkos facade to the end of this header)
#include "particle_structs.hpp"
#include "pumipic_wall.hpp"    // closest_point_on_triangle[_wnormal] (device-inline)
#include "pumipic_gather.hpp"  // interpolateTetVtx, interpolate2dField, ... (device-inline)

namespace pumipic {
class Mesh;
}
namespace Omega_h {
typedef int LO;
typedef long GO;
typedef double Real;
typedef int ClassId;
typedef int Int;
typedef signed char Byte;
typedef signed char I8;
template <class T>
using Write = pumipic::View<T>;
template <class T>
using Read = pumipic::View<T>;
typedef Read<LO> LOs;
typedef Read<GO> GOs;
typedef Read<Real> Reals;
enum { VERT = 0, EDGE = 1, FACE = 2, REGION = 3 };
// Omega_h::Adj as the drivers read it: a2ab (offsets; empty for a downward adjacency of fixed degree), ab2b (values)
struct Adj {
  Read<LO> a2ab, ab2b;
};
struct TagBase {};  // (Mesh::get_tagbase: a tag's existence)
// Omega_h::parallel_for(n, OMEGA_H_LAMBDA(LO i){...}, name) over a plain index range, on the library's
// stream (Omega_h_for.hpp); HostWrite / HostRead: a host copy of a device array
#define OMEGA_H_LAMBDA [=] __host__ __device__
#define OMEGA_H_DEVICE __host__ __device__ inline
#define OMEGA_H_INLINE __host__ __device__ inline
template <class F>
__global__ void oh_parallel_for_kernel(int n, F f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f(i);
}
template <class F>
void parallel_for(LO n, F f, const char* = "") {
  if (n <= 0) return;
  hipLaunchKernelGGL(oh_parallel_for_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)pp_stream(), (int)n, f);
}
template <class T>
class HostWrite {
 public:
  HostWrite() {}
  explicit HostWrite(size_t n) : h_(n) {}
  HostWrite(const std::vector<T>& v) : h_(v) {}  // (HostRead<LO>(picparts.nentsOffsets(dim)): already on the host)
  HostWrite(const pumipic::View<T>& v) : h_(v.size()) {
    pumipic::pp_check(pp_sync(), "HostWrite sync");
    h_ = v.to_host();
  }
  T& operator[](size_t i) { return h_[i]; }
  const T& operator[](size_t i) const { return h_[i]; }
  size_t size() const { return h_.size(); }
  T* data() { return h_.data(); }
  pumipic::View<T> write() const {  // Omega_h::Write<T>(HostWrite<T>)
    pumipic::View<T> v = pumipic::View<T>::uninitialized(h_.size());
    v.from_host(h_.data());
    return v;
  }
  operator pumipic::View<T>() const { return write(); }

 private:
  std::vector<T> h_;
};
template <class T>
HostWrite(const pumipic::View<T>&) -> HostWrite<T>;  // `auto h = o::HostWrite(device_array);`
template <class T>
using HostRead = HostWrite<T>;
// The drivers hold an Omega_h::Mesh* (the serial mesh inside the PICpart) next to the pumipic::Mesh
// (the PICpart); here one handle plays both parts.
typedef pumipic::Mesh Mesh;
}  // namespace Omega_h
namespace o = Omega_h;

namespace pumipic {

typedef double fp_t;
typedef fp_t Vector3d[3];

// pumipic::Input (src/pumipic_input.hpp:8-75): the full mesh, the partition vector (per element, or per
// classification id) and the buffer / safe rules of the PICparts
class Input {
 public:
  enum Method { INVALID = -1, FULL, BFS, MINIMUM, NONE };
  enum Ownership { PARTITION, CLASSIFICATION };
  inline Input(Mesh& mesh, Ownership rule, const std::vector<int>& partition_vector, Method bufferMethod_,
               Method safeMethod_, pp_comm* comm_ = nullptr);
  // (the reference passes the owners as Omega_h::Write<LO>: test/search2d.cpp:193-195)
  inline Input(Mesh& mesh, Ownership rule, const View<int>& partition_vector, Method bufferMethod_,
               Method safeMethod_, pp_comm* comm_ = nullptr);
  // (src/pumipic_input.cpp:23-111: the owners from a file -- `.ptn`: one owner per element; `.cpn`: the number of
  //  classification ids, then `id owner` pairs, ownership by classification.  One rank: everything on rank 0.)
  inline Input(Mesh& mesh, const char* partition_filename, Method bufferMethod_, Method safeMethod_,
               pp_comm* comm_ = nullptr);
  Ownership getRule() const { return ownership_rule; }
  const std::vector<int>& getPartition() const { return partition; }
  static Method getMethod(std::string s) {  // pumipic_input.cpp:139-150
    for (auto& c : s) c = (char)toupper(c);
    if (s == "FULL") return FULL;
    if (s == "BFS") return BFS;
    if (s == "MINIMUM") return MINIMUM;
    if (s == "NONE") return NONE;
    return INVALID;
  }
  int bridge_dim = 0;       // bridge dimension of the BFS (0 = vertices, dim-1 = sides)
  int bufferBFSLayers = 3;  // layers of the buffer (Method BFS)
  int safeBFSLayers = 1;    // layers of the safe zone (Method BFS)

 private:
  friend class Mesh;
  Mesh& m;
  Ownership ownership_rule;
  std::vector<int> partition;
  Method bufferMethod, safeMethod;
  pp_comm* comm;
};

// Omega_h::CommPtr as the drivers use it (`picparts.comm()->rank()`, `lib.world()->size()`); converts to the
// library's communicator handle
struct CommPtr {
  pp_comm* c = nullptr;
  CommPtr() {}
  CommPtr(pp_comm* c_) : c(c_) {}
  operator pp_comm*() const { return c; }
  const CommPtr* operator->() const { return this; }
  int rank() const { return pp_comm_rank(c); }
  int size() const { return pp_comm_size(c); }
  pp_comm* get_impl() const { return c; }
  void barrier() const { pp_check(pp_comm_barrier(c), "Comm::barrier"); }
};

class Mesh {
 public:
  Mesh() {}  // an empty handle, filled by pumipic::read / move assignment
  // PICparts from an Input (Mesh::Mesh(Input&), src/pumipic_part_construct.cpp:75-118): this object is the
  // part -- its own pp_mesh unless the buffer is FULL -- with the numberings and the exchange plan of
  // reduceCommArray (pp_picpart, include/pumipic_hip.h)
  explicit Mesh(Input& in) {
    comm_ = in.comm ? in.comm : comm_world();
    std::vector<int> owner;
    if (in.ownership_rule == Input::CLASSIFICATION) {
      owner.resize((size_t)in.m.nelems());
      pp_check(pp_owner_by_classification(in.m.handle(), in.partition.data(), (int)in.partition.size(),
                                          pp_comm_rank(comm_), owner.data()), "setOwnerByClassification");
    } else {
      owner = in.partition;
    }
    init_part(in.m, owner, (int)in.bufferMethod, (int)in.safeMethod, in.bridge_dim, in.bufferBFSLayers,
              in.safeBFSLayers, comm_);
  }
  // PICparts with a core and the whole mesh as buffer and safe zone (src/pumipic_part_construct.cpp:42-52)
  Mesh(Mesh& full_mesh, const std::vector<int>& partition_vector, pp_comm* comm = nullptr) {
    init_part(full_mesh, partition_vector, PP_PART_FULL, PP_PART_FULL, 0, 0, 0, comm);
  }
  // ... with all parts within buffer_layers of the core as buffer, the core plus safe_layers as safe zone
  // (:54-73; "Ghost layers must be >= safe layers")
  Mesh(Mesh& full_mesh, const std::vector<int>& partition_vector, int buffer_layers, int safe_layers,
       pp_comm* comm = nullptr) {
    if (buffer_layers < safe_layers) {
      fprintf(stderr, "Ghost layers must be >= safe layers\n");
      throw 1;
    }
    init_part(full_mesh, partition_vector, PP_PART_BFS, PP_PART_BFS, 0, buffer_layers, safe_layers, comm);
  }
  // (the same with the owners in a device array: test/test_comm_array.cpp:56)
  Mesh(Mesh& full_mesh, const View<int>& partition_vector, int buffer_layers, int safe_layers, pp_comm* comm = nullptr)
      : Mesh(full_mesh, (pp_check(pp_sync(), "pumipic::Mesh"), partition_vector.to_host()), buffer_layers, safe_layers,
             comm) {}
  // (Mesh(Omega_h::Mesh& full_mesh, Omega_h::LOs partition_vector), pumipic_mesh.hpp:27-31: owners in a device array)
  Mesh(Mesh& full_mesh, const View<int>& partition_vector, pp_comm* comm = nullptr) {
    pp_check(pp_sync(), "pumipic::Mesh");
    init_part(full_mesh, partition_vector.to_host(), PP_PART_FULL, PP_PART_FULL, 0, 0, 0, comm);
  }
  Mesh(int dim, const std::vector<double>& coords, const std::vector<int>& elem2verts,
       const std::vector<int>& class_id) {
    h_ = pp_mesh_create(dim, (int)(coords.size() / dim), coords.data(),
                        (int)(elem2verts.size() / (dim + 1)), elem2verts.data(),
                        class_id.empty() ? nullptr : class_id.data());
    if (!h_) pp_check(PP_EHIP, "pumipic::Mesh");
    own_ = std::make_shared<Owned>();
    own_->mesh = h_;
    pp_check(pp_mesh_info(h_, &dim_, &nverts_, &nelems_, &nsides_), "pp_mesh_info");
  }
  ~Mesh() {
    // (copies share the handles, as copies of an Omega_h::Mesh share its arrays: the LAST holder dumps and destroys)
    if (h_ && own_ && own_.use_count() == 1 && getenv("PP_DUMP_ON_DELETE")) {  // the real tags (scatter fields), see ParticleStructure::dumpOnDelete
      (void)pp_sync();
      for (auto& kv : real_tags_) {
        std::string name = kv.first;
        for (auto& ch : name)
          if (ch == ':' || ch == '/') ch = '_';
        const std::string fn = std::string(getenv("PP_DUMP_ON_DELETE")) + "_tag_" + name + "_r" + std::to_string(rank()) + ".f64";
        const std::vector<double> h = kv.second.to_host();
        if (FILE* f = fopen(fn.c_str(), "wb")) {
          fwrite(h.data(), sizeof(double), h.size(), f);
          fclose(f);
        }
      }
      for (auto& kv : int_tags_) {  // (has_particles of test/pseudoPushAndSearch.cpp: which elements ever held a particle)
        if (kv.first.find(":storage") != std::string::npos) continue;
        std::string name = kv.first;
        for (auto& ch : name)
          if (ch == ':' || ch == '/') ch = '_';
        const std::string fn = std::string(getenv("PP_DUMP_ON_DELETE")) + "_itag_" + name + "_r" + std::to_string(rank()) + ".i32";
        const std::vector<int> h = kv.second.to_host();
        if (FILE* f = fopen(fn.c_str(), "wb")) {
          fwrite(h.data(), sizeof(int), h.size(), f);
          fclose(f);
        }
      }
    }
  }
  // Omega_h::Mesh is passed BY VALUE through the reference's tests (test/test_adj.cpp:29,45,...): a copy is another
  // holder of the same device mesh (and part, and balancer) with its own tag table
  Mesh(const Mesh&) = default;
  Mesh& operator=(const Mesh&) = default;
  Mesh(Mesh&& o) noexcept { swap(o); }
  Mesh& operator=(Mesh&& o) noexcept {
    swap(o);
    return *this;
  }
  void swap(Mesh& o) {
    std::swap(h_, o.h_);
    own_.swap(o.own_);
    std::swap(part_, o.part_);
    std::swap(dim_, o.dim_);
    std::swap(nverts_, o.nverts_);
    std::swap(nelems_, o.nelems_);
    std::swap(nsides_, o.nsides_);
    std::swap(comm_, o.comm_);
    std::swap(owners_, o.owners_);
    std::swap(safe_, o.safe_);
    std::swap(gids_, o.gids_);
    real_tags_.swap(o.real_tags_);
    int_tags_.swap(o.int_tags_);
    parent_.swap(o.parent_);
    recipe_.swap(o.recipe_);
    std::swap(safe_ints_, o.safe_ints_);
    std::swap(safe_ints_src_, o.safe_ints_src_);
    std::swap(vert2sides_off_, o.vert2sides_off_);
    std::swap(vert2sides_, o.vert2sides_);
  }
  // a part cut from a full mesh that nobody else holds: the part takes the full mesh with it
  void keep_alive(Mesh&& full_mesh) { parent_ = std::make_shared<Mesh>(std::move(full_mesh)); }
  int dim() const { return dim_; }
  int nverts() const { return nverts_; }
  int nelems() const { return nelems_; }
  int nsides() const { return nsides_; }
  int nedges() const { return nents(1); }
  int nfaces() const { return nents(2); }
  int nregions() const { return dim_ == 3 ? nelems_ : 0; }
  // Omega_h::Mesh::ask_down(high, low) / get_adj(high, low) / ask_up(low, high) for the pairs the library derives:
  // element -> vertices / sides (/ edges of a tet), side -> vertices; vertex / side (/ tet edge) -> elements
  o::Adj ask_down(int high, int low) const {
    o::Adj a;
    if (high == dim_ && low == 0) a.ab2b = view<int>(PP_MESH_ELEM2VERTS);
    else if (high == dim_ && low == dim_ - 1) a.ab2b = view<int>(PP_MESH_ELEM2SIDES);
    else if (high == dim_ - 1 && low == 0) a.ab2b = view<int>(PP_MESH_SIDE2VERTS);
    else if (dim_ == 3 && high == 3 && low == 1) a.ab2b = view<int>(PP_MESH_ELEM2EDGES);
    else if (dim_ == 3 && high == 1 && low == 0) a.ab2b = view<int>(PP_MESH_EDGE2VERTS);
    else no_adjacency(high, low);
    return a;
  }
  o::Adj get_adj(int from, int to) const { return from > to ? ask_down(from, to) : ask_up(from, to); }
  o::Adj ask_up(int low, int high) const {
    o::Adj a;
    if (low == 0 && high == dim_) {
      a.a2ab = view<int>(PP_MESH_VERT2ELEMS_OFF);
      a.ab2b = view<int>(PP_MESH_VERT2ELEMS);
    } else if (low == dim_ - 1 && high == dim_) {
      a.a2ab = view<int>(PP_MESH_SIDE2ELEMS_OFF);
      a.ab2b = view<int>(PP_MESH_SIDE2ELEMS);
    } else if (dim_ == 3 && low == 1 && high == 3) {
      a.a2ab = view<int>(PP_MESH_EDGE2ELEMS_OFF);
      a.ab2b = view<int>(PP_MESH_EDGE2ELEMS);
    } else if (dim_ == 2 && low == 0 && high == 1) {
      // vertex -> edges of a triangle mesh (test/test_adj.cpp:174 walks them to pick a direction): inverted once on
      // the host from side -> vertices, edges of a vertex in ascending edge id as Omega_h's invert_adj orders them
      const_cast<Mesh*>(this)->ensure_vert2sides();
      a.a2ab = View<int>::wrap(vert2sides_off_.data(), (size_t)nverts_ + 1);
      a.ab2b = View<int>::wrap(vert2sides_.data(), (size_t)nsides_ * 2);
    } else {
      no_adjacency(low, high);
    }
    return a;
  }
  // Omega_h::Mesh::ask_verts_of(dim) / ask_dual(): element -> elements across a side
  o::LOs ask_verts_of(int edim) const { return ask_down(edim, 0).ab2b; }
  o::Adj ask_dual() const {
    o::Adj a;
    a.a2ab = view<int>(PP_MESH_DUAL_OFF);
    a.ab2b = view<int>(PP_MESH_DUAL_ELEMS);
    return a;
  }
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

  // ---- PICpart attributes (src/pumipic_mesh.hpp:40-130).  The mesh is fully buffered (every rank holds
  // all of it, the reference's Input::FULL); elements are owned in contiguous blocks, and the safe zone
  // is the own block plus `safe_layers` breadth-first layers (bfsBufferLayers,
  // pumipic_part_construct.cpp:407-437; 0 = BASELINE's rule: a particle migrates as soon as it
  // leaves its owner's block).
  void partition(pp_comm* comm, int safe_layers = 0) {
    comm_ = comm;
    const int world = pp_comm_size(comm), rank = pp_comm_rank(comm);
    std::vector<int> own((size_t)nelems_);
    for (int e = 0; e < nelems_; ++e) own[(size_t)e] = (int)((long long)e * world / (nelems_ > 0 ? nelems_ : 1));
    owners_ = o::Write<o::LO>((size_t)std::max(nelems_, 1));
    owners_.from_host(own.data());
    safe_ = View<unsigned char>((size_t)std::max(nelems_, 1));
    if (safe_layers > 0 && world > 1) {
      std::vector<int> has_part((size_t)world, 0);
      pp_check(pp_bfs_buffer_layers(h_, 0, rank, world, safe_layers, safe_layers, owners_.data(), safe_.data(),
                                    has_part.data()), "bfsBufferLayers");
    } else {
      std::vector<unsigned char> sf((size_t)nelems_);
      for (int e = 0; e < nelems_; ++e) sf[(size_t)e] = own[(size_t)e] == rank;
      safe_.from_host(sf.data());
    }
  }
  CommPtr comm() const { return CommPtr(comm_ ? comm_ : comm_world()); }
  int rank() const { return pp_comm_rank(comm()); }
  int num_ranks() const { return pp_comm_size(comm()); }
  pp_picpart* picpart() const { return part_; }
  // the part's particle balancer (Mesh::ptclBalancer, pumipic_mesh.hpp:76; the reference builds it at the end of
  // constructPICPart, here on first use); null for a mesh that was not built from an Input
  pp_balancer* ptclBalancerHandle() {
    if (!part_) return nullptr;
    if (!own_->bal) {
      own_->bal = pp_balancer_create(part_);
      if (!own_->bal) pp_check(PP_EHIP, "ParticleBalancer");
    }
    return own_->bal;
  }
  bool isFullMesh() const {
    if (!part_) return true;
    int full = 1;
    pp_check(pp_picpart_info(part_, &full, nullptr, nullptr, nullptr), "isFullMesh");
    return full != 0;
  }
  // Omega_h Mesh::nents(dim): vertices, edges (the sides of a triangle mesh; derived on first use for tets),
  // faces / sides, elements
  int nents(int edim) const {
    if (edim == 0) return nverts_;
    if (edim == dim_) return nelems_;
    if (edim == dim_ - 1) return nsides_;
    const int n = pp_mesh_num_edges(part_ ? pp_picpart_mesh(part_) : h_);
    pp_check(n < 0 ? n : 0, "nents(1)");
    return n;
  }
  int numBuffers(int /*edim*/) const {  // parts held, self included (pumipic_mesh.hpp:43)
    int nb = num_ranks();
    if (part_) pp_check(pp_picpart_info(part_, nullptr, &nb, nullptr, nullptr), "numBuffers");
    return nb;
  }
  std::vector<int> bufferedRanks(int edim) const {
    std::vector<int> r((size_t)num_ranks());
    int n = 0;
    if (part_) {
      pp_check(pp_picpart_buffered_ranks(part_, edim, r.data(), &n), "bufferedRanks");
    } else {
      for (int q = 0; q < num_ranks(); ++q)
        if (q != rank()) r[(size_t)n++] = q;
    }
    r.resize((size_t)n);
    return r;
  }
  // rankLocalIndex / commArrayIndex / nentsOffsets (pumipic_mesh.hpp:55-59), parts built from an Input
  o::LOs rankLocalIndex(int edim) { return part_view<o::LO>(PP_PART_RANK_LIDS, edim); }
  o::LOs commArrayIndex(int edim) { return part_view<o::LO>(PP_PART_COMM_INDEX, edim); }
  // (a device array, as the reference's Omega_h::LOs: test/test_file.cpp:93-96 reads it inside a lambda,
  //  test/test_comm_array.cpp:106 through a HostRead)
  o::LOs nentsOffsets(int edim) {
    std::vector<int> off((size_t)num_ranks() + 1, 0);
    // (a dimension above the mesh's: the reference keeps an empty array there, test/test_comm_array.cpp:106 asks for 3)
    if (part_ && edim <= dim_) pp_check(pp_picpart_nents_offsets(part_, edim, off.data()), "nentsOffsets");
    o::Write<o::LO> d(off.size());
    d.from_host(off.data());
    return d;
  }
  o::LOs entOwners(int dim) {
    if (part_) return part_view<o::LO>(PP_PART_OWNERS, dim);
    ensure_partition(dim);
    return owners_;
  }
  // safeTag(): the reference hands out Omega_h::LOs (one int per element, src/pumipic_mesh.hpp:66); the library keeps
  // the tag as bytes.  The returned object is the byte view (data(), size(), operator[]) and converts to an int view
  // on demand (made once per mesh: `Omega_h::LOs is_safe = picparts.safeTag();`, test/search2d.cpp:60).
  struct SafeTag {
    View<unsigned char> bytes;
    Mesh* owner;
    operator View<unsigned char>() const { return bytes; }
    operator View<int>() const { return owner->safe_as_ints(bytes); }
    PP_INLINE unsigned char* data() const { return bytes.data(); }
    PP_INLINE size_t size() const { return bytes.size(); }
    PP_INLINE unsigned char& operator[](size_t i) const { return bytes[i]; }
    std::vector<unsigned char> to_host() const { return bytes.to_host(); }
  };
  SafeTag safeTag() {
    if (part_) return SafeTag{part_view<unsigned char>(PP_PART_SAFE, dim_), this};
    ensure_partition(dim_);
    return SafeTag{safe_, this};
  }
  View<int> safe_as_ints(const View<unsigned char>& bytes) {
    if (safe_ints_.size() != bytes.size() || safe_ints_src_ != bytes.data()) {
      safe_ints_ = View<int>::uninitialized(std::max(bytes.size(), (size_t)1));
      const unsigned char* b = bytes.data();
      int* out = safe_ints_.data();
      o::parallel_for((o::LO)bytes.size(), [=] __host__ __device__(o::LO i) { out[i] = b[i]; }, "safeTag");
      safe_ints_src_ = bytes.data();
    }
    return View<int>::wrap(safe_ints_.data(), bytes.size());
  }
  o::GOs globalIds(int dim) {  // full-mesh replica without an Input: global id == local id
    if (part_) return part_view<o::GO>(PP_PART_GIDS, dim);
    if (gids_.size() == 0) {
      std::vector<o::GO> g((size_t)nelems_);
      for (int e = 0; e < nelems_; ++e) g[(size_t)e] = e;
      gids_ = o::Write<o::GO>((size_t)std::max(nelems_, 1));
      gids_.from_host(g.data());
    }
    return gids_;
  }
  // createCommArray / reduceCommArray (src/pumipic_mesh.hpp:92-110, pumipic_comm.cpp:222-246): every
  // entity is buffered on every rank, so the reduction is the all-reduce of the whole array
  // (a part built from an Input goes through the owners: pp_picpart_reduce, fan-in / fan-out, every Op,
  // o::LO and o::Real -- pumipic_comm.cpp:249-440)
  enum Op { SUM_OP, MAX_OP, MIN_OP, BCAST_OP };
  template <class T>
  o::Write<T> createCommArray(int edim, int num_entries_per_entity, T default_value) {
    return o::Write<T>((size_t)nents(edim) * num_entries_per_entity, default_value);
  }
  void reduceCommArray(int edim, Op op, o::Write<o::Real> array) {
    if (part_) {
      reduce_part(edim, op, PP_T_F64, array.data(), array.size());
      return;
    }
    if (op != SUM_OP) {
      fprintf(stderr, "reduceCommArray: MAX / MIN / BCAST need a part built from a pumipic::Input\n");
      exit(EXIT_FAILURE);
    }
    pp_check(pp_allreduce_sum(comm(), array.data(), (int64_t)array.size()), "reduceCommArray");
  }
  void reduceCommArray(int edim, Op op, o::Write<o::LO> array) {
    if (!part_) {
      fprintf(stderr, "reduceCommArray<LO>: needs a part built from a pumipic::Input\n");
      exit(EXIT_FAILURE);
    }
    reduce_part(edim, op, PP_T_I32, array.data(), array.size());
  }
  // ---- mesh tags the drivers use (Omega_h::Mesh::add_tag / set_tag / get_array)
  template <class T>
  void add_tag(int edim, const std::string& name, int /*ncomps*/, View<T> values) {
    set_tag(edim, name, values);
  }
  void set_tag(int edim, const std::string& name, View<double> values) { real_tags_[key(edim, name)] = values; }
  void set_tag(int edim, const std::string& name, View<int> values) { int_tags_[key(edim, name)] = values; }
  bool has_tag(int edim, const std::string& name) const {
    return real_tags_.count(key(edim, name)) || int_tags_.count(key(edim, name)) ||
           (name == "class_id" && edim == dim_);
  }
  template <class T>
  View<T> get_array(int edim, const std::string& name) {
    return get_array_impl(edim, name, (T*)nullptr);
  }
  // Omega_h::Mesh::get_tagbase(dim, name): only its existence is asked for (test/test_full_mesh.cpp:44)
  const o::TagBase* get_tagbase(int edim, const std::string& name) {
    static const o::TagBase tag;
    if (!(has_tag(edim, name) || ((name == "global" || name == "global_serial") && edim == dim_))) {
      fprintf(stderr, "mesh has no tag %s on dimension %d\n", name.c_str(), edim);
      exit(EXIT_FAILURE);
    }
    return &tag;
  }
  Mesh* mesh() { return this; }             // picparts.mesh()
  Mesh* operator->() { return this; }       // picparts->dim()
  o::LOs ask_verts_of_elems() const { return view<int>(PP_MESH_ELEM2VERTS); }

 private:
  static std::string key(int edim, const std::string& name) { return std::to_string(edim) + ":" + name; }
  View<double> get_array_impl(int edim, const std::string& name, double*) {
    auto it = real_tags_.find(key(edim, name));
    if (it == real_tags_.end()) {
      fprintf(stderr, "mesh has no real tag %s on dimension %d\n", name.c_str(), edim);
      exit(EXIT_FAILURE);
    }
    return it->second;
  }
  // "global" of a serial mesh / "global_serial" of a part (pumipic_part_construct.cpp renames the tag): the ids of
  // the full mesh's entities
  View<long> get_array_impl(int edim, const std::string& name, long*) {
    if (name != "global" && name != "global_serial") {
      fprintf(stderr, "mesh has no 64-bit integer tag %s on dimension %d\n", name.c_str(), edim);
      exit(EXIT_FAILURE);
    }
    return globalIds(edim);
  }
  View<int> get_array_impl(int edim, const std::string& name, int*) {
    if (name == "class_id" && edim == dim_) return class_ids();
    auto it = int_tags_.find(key(edim, name));
    if (it == int_tags_.end()) {
      fprintf(stderr, "mesh has no integer tag %s on dimension %d\n", name.c_str(), edim);
      exit(EXIT_FAILURE);
    }
    return it->second;
  }
  void ensure_vert2sides() {
    if (vert2sides_off_.size() != 0) return;
    pp_check(pp_sync(), "ask_up(0, 1)");
    const std::vector<int> s2v = view<int>(PP_MESH_SIDE2VERTS).to_host();
    std::vector<int> off((size_t)nverts_ + 1, 0), adj((size_t)nsides_ * 2);
    for (int v : s2v) ++off[(size_t)v + 1];
    for (int v = 0; v < nverts_; ++v) off[(size_t)v + 1] += off[(size_t)v];
    std::vector<int> fill(off.begin(), off.end() - 1);
    for (int sd = 0; sd < nsides_; ++sd)
      for (int k = 0; k < 2; ++k) adj[(size_t)fill[(size_t)s2v[(size_t)sd * 2 + k]]++] = sd;
    vert2sides_off_ = View<int>(off.size());
    vert2sides_off_.from_host(off.data());
    vert2sides_ = View<int>(std::max(adj.size(), (size_t)1));
    vert2sides_.from_host(adj.data());
  }
  void ensure_partition(int) {
    if (owners_.size() == 0) partition(comm(), 0);
  }
  [[noreturn]] void no_adjacency(int a, int b) const {
    fprintf(stderr, "pumipic::Mesh: the adjacency %d -> %d of a %d-D mesh is not derived by this library\n", a, b, dim_);
    exit(EXIT_FAILURE);
  }
  template <class T>
  View<T> view(int which) const {
    size_t n = 0;
    const void* p = pp_mesh_array_dev(h_, which, &n);
    return View<T>::wrap((T*)p, n);
  }
  void init_part(Mesh& full, const std::vector<int>& owner, int buffer_method, int safe_method, int bridge_dim,
                 int buffer_layers, int safe_layers, pp_comm* comm) {
    comm_ = comm ? comm : comm_world();
    if ((int)owner.size() != full.nelems()) {
      fprintf(stderr, "pumipic::Mesh: the partition vector holds %zu owners for %d elements\n", owner.size(),
              full.nelems());
      exit(EXIT_FAILURE);
    }
    recipe_ = std::make_shared<PartRecipe>(PartRecipe{full.handle(), owner, buffer_method, safe_method, bridge_dim,
                                                      buffer_layers, safe_layers});
    part_ = pp_picpart_create(full.handle(), owner.data(), buffer_method, safe_method, bridge_dim, buffer_layers,
                              safe_layers, comm_);
    if (!part_) pp_check(PP_EHIP, "pumipic::Mesh (PICparts)");
    own_ = std::make_shared<Owned>();
    own_->part = part_;
    h_ = const_cast<pp_mesh*>(pp_picpart_mesh(part_));
    pp_check(pp_mesh_info(h_, &dim_, &nverts_, &nelems_, &nsides_), "pp_mesh_info");
    // a part that buffers the whole mesh numbers its entities as the full mesh does: the full mesh's tags (the side
    // classification a reader attached, fields a driver added) are the part's too, as the reference's parts carry them
    if (nverts_ == full.nverts() && nelems_ == full.nelems() && nsides_ == full.nsides()) {
      real_tags_ = full.real_tags_;
      int_tags_ = full.int_tags_;
    }
  }
  void reduce_part(int edim, Op op, int dtype, void* data, size_t n) {
    const int ne = nents(edim);
    if (ne <= 0) return;
    if (n % (size_t)ne != 0) {  // pumipic_comm.cpp:253-256
      fprintf(stderr, "Comm array size does not match the expected size for dimension %d\n", edim);
      return;
    }
    pp_check(pp_picpart_reduce(part_, edim, (int)op, dtype, (int)(n / (size_t)ne), data), "reduceCommArray");
  }
  template <class T>
  View<T> part_view(int which, int edim) const {
    size_t n = 0;
    const void* p = pp_picpart_array_dev(part_, which, edim, &n);
    if (!p && n) pp_check(PP_EINVAL, "picpart array");
    return View<T>::wrap((T*)p, n);
  }
  // what the last holder destroys: the library's mesh (a mesh made from arrays) or the part (its mesh goes with it)
  struct Owned {
    pp_mesh* mesh = nullptr;
    pp_picpart* part = nullptr;
    pp_balancer* bal = nullptr;
    ~Owned() {
      if (bal) (void)pp_balancer_destroy(bal);
      if (part) (void)pp_picpart_destroy(part);
      if (mesh) (void)pp_mesh_destroy(mesh);
    }
  };
  std::shared_ptr<Owned> own_;
  // how a part was cut (pumipic::write stores it; the full mesh must be alive when write is called)
  struct PartRecipe {
    const pp_mesh* full;
    std::vector<int> owner;
    int buffer_method, safe_method, bridge_dim, buffer_layers, safe_layers;
  };
  std::shared_ptr<PartRecipe> recipe_;
  friend void write(Mesh& picparts, const char* prefix);
  friend void read_parts_container(const std::string& fn, pp_comm* comm, Mesh* picparts);
  pp_mesh* h_ = nullptr;
  pp_picpart* part_ = nullptr;
  View<int> vert2sides_off_, vert2sides_;
  int dim_ = 0, nverts_ = 0, nelems_ = 0, nsides_ = 0;
  pp_comm* comm_ = nullptr;
  o::Write<o::LO> owners_;
  View<unsigned char> safe_;
  o::Write<o::GO> gids_;
  std::map<std::string, View<double>> real_tags_;
  std::map<std::string, View<int>> int_tags_;
  std::shared_ptr<Mesh> parent_;
  View<int> safe_ints_;
  const unsigned char* safe_ints_src_ = nullptr;
};

inline Input::Input(Mesh& mesh, Ownership rule, const std::vector<int>& partition_vector, Method bufferMethod_,
                    Method safeMethod_, pp_comm* comm_)
    : m(mesh), ownership_rule(rule), partition(partition_vector), bufferMethod(bufferMethod_),
      safeMethod(safeMethod_), comm(comm_) {
  if (bufferMethod == NONE) {  // pumipic_input.cpp:122-126
    fprintf(stderr, "[WARNING] bufferMethod given as NONE, setting to MINIMUM\n");
    bufferMethod = MINIMUM;
  }
  if (bufferMethod == MINIMUM) bufferBFSLayers = 0;  // :133-136
  if (safeMethod == MINIMUM) safeBFSLayers = 0;
}

inline Input::Input(Mesh& mesh, const char* partition_filename, Method bufferMethod_, Method safeMethod_, pp_comm* comm_)
    : Input(mesh, PARTITION, std::vector<int>((size_t)mesh.nelems(), 0), bufferMethod_, safeMethod_, comm_) {
  pp_comm* c = comm ? comm : comm_world();
  if (pp_comm_size(c) <= 1) return;
  const std::string fn(partition_filename);
  const size_t dot = fn.find_last_of('.');
  if (dot == std::string::npos) {
    printError("Filename provided has no extension (%s)", partition_filename);
    throw std::runtime_error("Filename has no extension");
  }
  const std::string ext = fn.substr(dot + 1);
  std::ifstream in_str(partition_filename);
  if (ext != "ptn" && ext != "cpn") {
    printError("Only .ptn and .cpn partitions are supported");
    throw std::runtime_error("Invalid partition file extension");
  }
  if (!in_str) {
    if (!pp_comm_rank(c)) printError("Cannot open file %s\n", partition_filename);
    throw std::runtime_error("Cannot open file");
  }
  int own;
  if (ext == "ptn") {
    size_t index = 0;
    while (in_str >> own && index < partition.size()) partition[index++] = own;
  } else {
    ownership_rule = CLASSIFICATION;
    int size = 0, cid;
    in_str >> size;
    partition.assign((size_t)size + 1, 0);
    while (in_str >> cid >> own)
      if (cid >= 0 && cid <= size) partition[(size_t)cid] = own;
  }
}

inline Input::Input(Mesh& mesh, Ownership rule, const View<int>& partition_vector, Method bufferMethod_,
                    Method safeMethod_, pp_comm* comm_)
    : Input(mesh, rule, (pp_check(pp_sync(), "pumipic::Input"), partition_vector.to_host()), bufferMethod_, safeMethod_,
            comm_) {}

// ---------------------------------------------------------------- pumipic::write / the container pumipic::read takes back
// (src/pumipic_mesh.hpp:150-151, pumipic_file.cpp:44-204.)  The reference writes, per rank, the part's mesh as an
// Omega_h .osh directory and its comm arrays as a .ppm file; .osh is defined by Omega_h's sources, which are not in the
// reference tree.  This library's container holds what the parts were CUT from -- the full mesh, the owner of every
// element, the buffer / safe rules -- in one file <prefix>_<ranks>.pparts, and read() cuts them again: the part a rank
// reads equals the part it wrote, entity for entity (test/test_file.cpp compares every array).  Rank 0 writes.
inline void write(Mesh& picparts, const char* prefix) {
  if (!picparts.recipe_) {
    fprintf(stderr, "pumipic::write: the mesh was not built from a pumipic::Input / partition vector\n");
    exit(EXIT_FAILURE);
  }
  pp_comm* c = picparts.comm();
  if (pp_comm_rank(c) == 0) {
    const Mesh::PartRecipe& r = *picparts.recipe_;
    pp_check(pp_sync(), "pumipic::write");
    int dim = 0, nv = 0, ne = 0, ns = 0;
    pp_check(pp_mesh_info(r.full, &dim, &nv, &ne, &ns), "pumipic::write");
    auto host = [&](int which, size_t bytes_per, std::vector<char>& out) {
      size_t n = 0;
      const void* p = pp_mesh_array_dev(r.full, which, &n);
      out.resize(n * bytes_per);
      if (n) pp_check(pp_memcpy_d2h(out.data(), p, out.size()), "pumipic::write");
    };
    std::vector<char> coords, e2v, cls;
    host(PP_MESH_COORDS, sizeof(double), coords);
    host(PP_MESH_ELEM2VERTS, sizeof(int), e2v);
    host(PP_MESH_CLASS_ID, sizeof(int), cls);
    const std::string fn = std::string(prefix) + "_" + std::to_string(pp_comm_size(c)) + ".pparts";
    FILE* f = fopen(fn.c_str(), "wb");
    if (!f) {
      fprintf(stderr, "pumipic::write: cannot open %s\n", fn.c_str());
      exit(EXIT_FAILURE);
    }
    const int hdr[10] = {0x50505054, dim, nv, ne, r.buffer_method, r.safe_method, r.bridge_dim, r.buffer_layers,
                         r.safe_layers, pp_comm_size(c)};
    fwrite(hdr, sizeof(int), 10, f);
    fwrite(coords.data(), 1, coords.size(), f);
    fwrite(e2v.data(), 1, e2v.size(), f);
    fwrite(cls.data(), 1, cls.size(), f);
    fwrite(r.owner.data(), sizeof(int), r.owner.size(), f);
    fclose(f);
  }
  pp_check(pp_comm_barrier(c), "pumipic::write");
}
inline void read_parts_container(const std::string& fn, pp_comm* comm, Mesh* picparts) {
  FILE* f = fopen(fn.c_str(), "rb");
  int hdr[10] = {0};
  bool ok = f && fread(hdr, sizeof(int), 10, f) == 10 && hdr[0] == 0x50505054 && hdr[9] == pp_comm_size(comm);
  std::vector<double> coords;
  std::vector<int> e2v, cls, owner;
  if (ok) {  // (sizes a file of this length can hold)
    fseek(f, 0, SEEK_END);
    const long long len = ftell(f);
    fseek(f, 10 * (long)sizeof(int), SEEK_SET);
    ok = (hdr[1] == 2 || hdr[1] == 3) && hdr[2] >= 0 && hdr[3] >= 0 &&
         40ll + 8ll * hdr[2] * hdr[1] + 4ll * hdr[3] * (hdr[1] + 3) <= len;
  }
  if (ok) {
    const int dim = hdr[1];
    coords.resize((size_t)hdr[2] * dim);
    e2v.resize((size_t)hdr[3] * (dim + 1));
    cls.resize((size_t)hdr[3]);
    owner.resize((size_t)hdr[3]);
    ok = fread(coords.data(), sizeof(double), coords.size(), f) == coords.size() &&
         fread(e2v.data(), sizeof(int), e2v.size(), f) == e2v.size() &&
         fread(cls.data(), sizeof(int), cls.size(), f) == cls.size() &&
         fread(owner.data(), sizeof(int), owner.size(), f) == owner.size();
  }
  if (f) fclose(f);
  if (!ok) {
    fprintf(stderr, "%s: not a parts container written by pumipic::write for %d ranks\n", fn.c_str(), pp_comm_size(comm));
    exit(EXIT_FAILURE);
  }
  Mesh full(hdr[1], coords, e2v, cls);
  Mesh part;
  part.init_part(full, owner, hdr[4], hdr[5], hdr[6], hdr[7], hdr[8], comm);
  picparts->swap(part);
  picparts->keep_alive(std::move(full));  // (the part refers to the full mesh it was cut from)
}

// ---------------------------------------------------------------- pumipic::ParticleBalancer
// (src/pumipic_lb.hpp:33-118) over pp_balancer: sbars, weights and the selection as in the reference, the
// balancing step (EnGPar there) is the library's diffusion on the gathered weight table.
class ParticleBalancer {
 public:
  explicit ParticleBalancer(Mesh& picparts) {
    if (!picparts.picpart()) {
      fprintf(stderr, "ParticleBalancer: the mesh must be built from a pumipic::Input\n");
      exit(EXIT_FAILURE);
    }
    b_ = pp_balancer_create(picparts.picpart());
    if (!b_) pp_check(PP_EHIP, "ParticleBalancer");
  }
  ~ParticleBalancer() {
    if (b_) (void)pp_balancer_destroy(b_);
  }
  ParticleBalancer(const ParticleBalancer&) = delete;
  ParticleBalancer& operator=(const ParticleBalancer&) = delete;
  // new_procs of the particles that move is rewritten; particles pushed out of the safe zone must already
  // carry their owner (pumipic_lb.hpp:41-52)
  template <class PS>
  void repartition(Mesh& /*picparts*/, PS* ptcls, double tol, View<int> new_elems, View<int> new_procs,
                   double step_factor = 0.3) {
    pp_check(pp_balancer_repartition(b_, ptcls->handle(), tol, new_elems.data(), new_procs.data(), step_factor),
             "ParticleBalancer::repartition");
  }
  // array form (:54-64): particles per element of the part -> new process per particle, element-major
  View<int> partition(Mesh& /*picparts*/, View<int> ptcls_per_elem, double tol, double step_factor = 0.3,
                      int /*selection_iterations*/ = 5) {
    pp_check(pp_sync(), "ParticleBalancer::partition");
    std::vector<int> ppe = ptcls_per_elem.to_host();
    long long np = 0;
    for (int n : ppe) np += n;
    std::vector<int> procs((size_t)std::max<long long>(np, 1));
    pp_check(pp_balancer_partition(b_, ppe.data(), tol, step_factor, procs.data()), "ParticleBalancer::partition");
    View<int> out((size_t)np);
    out.from_host(procs.data());
    return out;
  }
  o::LOs getSbarIDs(Mesh& /*picparts*/) const {
    size_t n = 0;
    const int* p = pp_balancer_sbar_ids_dev(b_, &n);
    return View<int>::wrap(const_cast<int*>(p), n);
  }
  pp_balancer* handle() const { return b_; }

 private:
  pp_balancer* b_ = nullptr;
};

// ---------------------------------------------------------------- timing (support/ppTiming.hpp:34-75)
struct TimingEntry {
  double time = 0, prebarrier = 0;
  int count = 0, order = 0;
};
inline std::map<std::string, TimingEntry>& timing_table() {
  static std::map<std::string, TimingEntry> t;
  return t;
}
inline int& timing_verbosity() {
  static int v = 0;
  return v;
}
inline int& timing_enabled() {
  static int e = -1;  // -1: default (rank 0 of the world records)
  return e;
}
inline void SetTimingVerbosity(int v) { timing_verbosity() = v; }
inline void EnableTiming() { timing_enabled() = 1; }
inline void DisableTiming() { timing_enabled() = 0; }
inline bool isTiming() {
  return timing_enabled() == 1 || (timing_enabled() == -1 && pp_comm_rank(comm_world()) == 0);
}
inline void RecordTime(const std::string& name, double seconds, double prebarrierTime = 0.0) {
  if (!isTiming()) return;
  auto& t = timing_table();
  auto it = t.find(name);
  if (it == t.end()) {
    it = t.emplace(name, TimingEntry()).first;
    it->second.order = (int)t.size();
  }
  it->second.time += seconds;
  it->second.prebarrier += prebarrierTime;
  it->second.count += 1;
  if (timing_verbosity() >= 1)
    fprintf(stderr, "%d %s (seconds) %f pre-barrier (seconds) %f\n", pp_comm_rank(comm_world()), name.c_str(),
            seconds, prebarrierTime);
}
// start / end of an operation a structure times itself (particle_structs.hpp: rebuild, migrate): a host clock, as the
// reference's Kokkos::Timer is; with PP_TIMER_FENCE=1 (timer_fences(), below) the stream is drained on both sides, so
// that the row holds the operation's device time
inline bool timer_fences();
inline double op_timer_start() {
  if (!isTiming()) return 0.0;
  if (timer_fences()) (void)pp_sync();
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void op_timer_record(const std::string& name, double t0) {
  if (!isTiming()) return;
  if (timer_fences()) (void)pp_sync();
  RecordTime(name, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0);
}
enum TimingSortOption { SORT_ALPHA, SORT_ORDER, SORT_LONGEST, SORT_SHORTEST };
inline std::vector<std::pair<std::string, TimingEntry>> sorted_timing(TimingSortOption sort) {
  std::vector<std::pair<std::string, TimingEntry>> v(timing_table().begin(), timing_table().end());
  if (sort == SORT_ORDER)
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.order < b.second.order; });
  else if (sort == SORT_LONGEST)
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.time > b.second.time; });
  else if (sort == SORT_SHORTEST)
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second.time < b.second.time; });
  return v;
}
inline void SummarizeTime(TimingSortOption sort = SORT_ALPHA) {
  if (timing_verbosity() < 0 || !isTiming()) return;
  fprintf(stderr, "Timing Summary %d\n%-40s %14s %8s %14s\n", pp_comm_rank(comm_world()), "Operation", "Total Time",
          "Calls", "Average Time");
  for (auto& kv : sorted_timing(sort))
    fprintf(stderr, "%-40s %14.6f %8d %14.6f\n", kv.first.c_str(), kv.second.time, kv.second.count,
            kv.second.time / kv.second.count);
}
// collective: max / min (with the rank that holds it) / average over the ranks that record
inline void SummarizeTimeAcrossProcesses(TimingSortOption sort = SORT_ALPHA) {
  pp_comm* c = comm_world();
  const int n = pp_comm_size(c), me = pp_comm_rank(c);
  // the operation names of rank 0 define the table (the ranks run the same program)
  auto mine = sorted_timing(sort);
  std::vector<double> row(2 * 64, -1.0);
  std::vector<std::string> names;
  for (auto& kv : mine)
    if (names.size() < 64) names.push_back(kv.first);
  for (size_t i = 0; i < names.size(); ++i) {
    row[2 * i] = isTiming() ? mine[i].second.time : -1.0;
    row[2 * i + 1] = mine[i].second.count;
  }
  std::vector<double> all((size_t)n * row.size());
  pp_check(pp_comm_allgather_host(c, row.data(), all.data(), (int)(row.size() * sizeof(double))),
           "SummarizeTimeAcrossProcesses");
  if (me != 0 || timing_verbosity() < 0) return;
  int nt = 0;
  for (int r = 0; r < n; ++r) nt += all[(size_t)r * row.size()] >= 0;
  fprintf(stderr, "Reduced Timing Summary with %d ranks\n%-40s %22s %22s %14s %10s\n", nt, "Operation",
          "Max Time (max proc)", "Min Time (min proc)", "Average Time", "Call Count");
  for (size_t i = 0; i < names.size(); ++i) {
    double mx = -1, mn = 1e300, sum = 0;
    int rmx = 0, rmn = 0, cnt = 0;
    for (int r = 0; r < n; ++r) {
      const double t = all[(size_t)r * row.size() + 2 * i];
      if (t < 0) continue;
      if (t > mx) mx = t, rmx = r;
      if (t < mn) mn = t, rmn = r;
      sum += t;
      ++cnt;
    }
    if (cnt)
      fprintf(stderr, "%-40s %14.6f (%5d) %14.6f (%5d) %14.6f %10d\n", names[i].c_str(), mx, rmx, mn, rmn, sum / cnt,
              (int)all[2 * i + 1]);
  }
}
// prebarrier (support/ppTiming.hpp, SCS_rebuild.h:124): optional barrier before a timed operation
inline int& prebarrier_enabled() {
  static int e = 0;
  return e;
}
inline void enable_prebarrier() { prebarrier_enabled() = 1; }
inline double pumipic_prebarrier(pp_comm* c = nullptr) {
  if (!prebarrier_enabled()) return 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  pp_check(pp_comm_barrier(c ? c : comm_world()), "prebarrier");
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
inline double prebarrier() { return pumipic_prebarrier(); }
// Kokkos::fence(): the host waits for everything enqueued on the library stream
inline void fence() { pp_check(pp_sync(), "fence"); }
// Kokkos::Timer: a host clock.  Like the reference's, seconds() does NOT wait for the device (test/ellipticalPush.hpp:39,68
// records the launch of an asynchronous kernel); callers that want device time fence first, as the reference does
// (performance_tests/ps_combo160.cpp:180-184).  PP_TIMER_FENCE=1 in the environment (or SetTimerFence(true)) makes every
// seconds() fence: per-operation tables that hold device time, at the price of one host wait per recorded operation.
inline int& timer_fence_flag() {
  static int f = (getenv("PP_TIMER_FENCE") && atoi(getenv("PP_TIMER_FENCE"))) ? 1 : 0;
  return f;
}
inline void SetTimerFence(bool on) { timer_fence_flag() = on ? 1 : 0; }
inline bool timer_fences() { return timer_fence_flag() != 0; }
struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void reset() { t0 = std::chrono::steady_clock::now(); }
  double seconds() const {
    if (timer_fence_flag()) (void)pp_sync();
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
  // PP_SEARCH_DUMP=<prefix> (a hook of the mirror headers, like PP_DUMP_ON_DELETE: the reference's test programs cannot
  // be touched): every call writes its inputs and outputs to <prefix>_call<k>_*.bin, so that a particle one of those
  // programs complains about can be handed to the oracle (tools/replay_search_dump.py)
  static int dump_call = 0;
  const char* dump = getenv("PP_SEARCH_DUMP");
  auto dump_file = [&](const char* what, const void* host, size_t bytes) {
    const std::string fn = std::string(dump) + "_call" + std::to_string(dump_call) + "_" + what + ".bin";
    if (FILE* f = fopen(fn.c_str(), "wb")) {
      fwrite(host, 1, bytes, f);
      fclose(f);
    }
  };
  auto dump_member = [&](const char* what, int m, size_t bytes_per) {
    const pp_ps_info_t inf = ptcls->info();
    std::vector<char> h((size_t)inf.stride * bytes_per);
    pp_check(pp_ps_member_to_host(ptcls->handle(), m, h.data()), "PP_SEARCH_DUMP");
    dump_file(what, h.data(), h.size());
  };
  if (dump) {
    pp_check(pp_sync(), "PP_SEARCH_DUMP");
    const pp_ps_info_t inf = ptcls->info();
    const int hdr[6] = {(int)cap, (int)inf.stride, seeded, requireIntersection ? 1 : 0, looplimit, mesh.dim()};
    dump_file("hdr", hdr, sizeof(hdr));
    dump_member("xo", x_ps_orig.member(), 3 * sizeof(double));
    dump_member("xt", x_ps_tgt.member(), 3 * sizeof(double));
    std::vector<int> slot_elem(cap), ein = elem_ids.to_host();
    std::vector<unsigned char> mask(cap);
    pp_check(pp_ps_layout_to_host(ptcls->handle(), nullptr, nullptr, nullptr, nullptr, mask.data(), slot_elem.data()),
             "PP_SEARCH_DUMP");
    dump_file("mask", mask.data(), mask.size());
    dump_file("elem", slot_elem.data(), slot_elem.size() * sizeof(int));
    dump_file("ein", ein.data(), ein.size() * sizeof(int));
  }
  int found = 1, notin = 0;
  pp_check(pp_search_mesh(mesh.handle(), ptcls->handle(), x_ps_orig.member(), x_ps_tgt.member(),
                          pids.member(), elem_ids.data(), seeded, requireIntersection ? 1 : 0,
                          inter_faces.data(), inter_points.data(), looplimit, &found, &notin),
           "search_mesh");
  if (dump) {
    pp_check(pp_sync(), "PP_SEARCH_DUMP");
    const std::vector<int> eout = elem_ids.to_host();
    dump_file("eout", eout.data(), eout.size() * sizeof(int));
    if (requireIntersection) {
      const std::vector<int> fc = inter_faces.to_host();
      const std::vector<double> pt = inter_points.to_host();
      dump_file("face", fc.data(), fc.size() * sizeof(int));
      dump_file("pts", pt.data(), pt.size() * sizeof(double));
    }
    ++dump_call;
  }
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

// setUnsafeProcs (src/pumipic_ptcl_ops.hpp:32-52): a particle whose new element is not safe on this
// part goes to that element's owner
template <class PS>
void setUnsafeProcs(Mesh& mesh, PS* ptcls, o::LOs elems, typename PS::kkLidView new_elems,
                    typename PS::kkLidView new_procs) {
  pp_check(pp_set_unsafe_procs(ptcls->handle(), elems.data(), mesh.safeTag().data(),
                               mesh.entOwners(mesh.dim()).data(), mesh.rank(), new_elems.data(),
                               new_procs.data()),
           "setUnsafeProcs");
}
// migrate_ptcls / migrate_lb_ptcls (src/pumipic_ptcl_ops.hpp:53-85).  On parts built from a pumipic::Input
// migrate_lb_ptcls runs the part's balancer between setUnsafeProcs and the migration, as the reference does
// (ParticleBalancer::repartition, pumipic_lb.hpp:352-362); a mesh partitioned with Mesh::partition (the
// full-mesh replica without an Input) has none: particles go to the owner of their element.
// (`dist`: the reference's two functions migrate over the world Distributor; a caller that holds the subset
// form -- self + the buffered ranks of its part, test/pseudoXGCm.cpp:390-396 -- may pass it: the migration is
// then checked against it, ParticleStructure::migrate)
// One rank: setUnsafeProcs copies `elems` and names this rank for every particle, and migrate() is rebuild()
// (scs/SCS_migrate.h:20-25) -- the rebuild reads `elems` itself, without the pass that would write the two routing
// arrays (45 us and two 50 MB arrays per step at 10 M particles).
template <class PS>
inline bool migrate_on_one_rank(Mesh& mesh, PS* ptcls, o::LOs elems, const Distributor<>* dist) {
  if (pp_comm_size(dist ? dist->comm() : (pp_comm*)mesh.comm()) != 1) return false;
  Timer init_timer;
  RecordTime("migration_init", init_timer.seconds());
  Timer migrate_timer;
  ptcls->rebuild(View<lid_t>::wrap(const_cast<lid_t*>(elems.data()), elems.size()));
  RecordTime("migration", migrate_timer.seconds());
  return true;
}
template <class PS>
void migrate_ptcls(Mesh& mesh, PS* ptcls, o::LOs elems, const Distributor<>* dist = nullptr) {
  if (migrate_on_one_rank(mesh, ptcls, elems, dist)) return;
  Timer init_timer;
  const size_t cap = (size_t)std::max(ptcls->capacity(), 1);
  // (setUnsafeProcs writes every slot of both arrays)
  auto new_elems = PS::kkLidView::uninitialized(cap), new_procs = PS::kkLidView::uninitialized(cap);
  setUnsafeProcs(mesh, ptcls, elems, new_elems, new_procs);
  RecordTime("migration_init", init_timer.seconds());
  Timer migrate_timer;
  ptcls->migrate(new_elems, new_procs, dist ? *dist : Distributor<>(mesh.comm()));
  RecordTime("migration", migrate_timer.seconds());
}
template <class PS>
void migrate_lb_ptcls(Mesh& mesh, PS* ptcls, o::LOs elems, float tol, float step_factor = 0.5,
                      const Distributor<>* dist = nullptr) {
  pp_balancer* balancer = mesh.ptclBalancerHandle();
  if (!balancer) {
    migrate_ptcls(mesh, ptcls, elems, dist);
    return;
  }
  Timer init_timer;
  const size_t cap = (size_t)std::max(ptcls->capacity(), 1);
  // (setUnsafeProcs writes every slot of both arrays)
  auto new_elems = PS::kkLidView::uninitialized(cap), new_procs = PS::kkLidView::uninitialized(cap);
  setUnsafeProcs(mesh, ptcls, elems, new_elems, new_procs);
  RecordTime("migration_init", init_timer.seconds());
  Timer balance_timer;
  pp_check(pp_balancer_repartition(balancer, ptcls->handle(), tol, new_elems.data(), new_procs.data(), step_factor),
           "ParticleBalancer::repartition");
  RecordTime("migration_balance", balance_timer.seconds());
  Timer migrate_timer;
  ptcls->migrate(new_elems, new_procs, dist ? *dist : Distributor<>(mesh.comm()));
  RecordTime("migration", migrate_timer.seconds());
}

}  // namespace pumipic
namespace p = pumipic;
namespace ps = particle_structs;
// The reference declares the device helpers of the search (barycentric_*, ray_intersects_triangle, check_initial_parents,
// ...) in THIS header; here they are adapters in pumipic_utils.hpp over Omega_h-style vectors, which need the Kokkos and
// Omega_h facades complete.  Whichever of the three headers a translation unit names first, the last one to finish
// pulls in the rest.
#define PP_ADJACENCY_BODY_DONE
#include "compat/Kokkos_Core.hpp"
#ifdef PP_KOKKOS_CORE_DONE
#include "compat/Omega_h_mesh.hpp"
#endif
