// particle_structs.hpp -- C++ host mirror of PUMI-PIC's particle_structs operator API over the
// C-ABI of libpumipic_hip.so (include/pumipic_hip.h).  Header-only; compile user code with hipcc.
//
// Mirrors (names, argument meaning, error behaviour):
//   MemberTypes / MemberTypeAtIndex      particle_structs/src/support/MemberTypes.h:24-73
//   Segment<T>  seg(pid) / seg(pid,i)    particle_structs/src/support/Segment.h:29-98
//   ParticleStructure<DataTypes>         particle_structs/src/particle_structure.hpp:63-104
//   SellCSigma, SCS_Input                particle_structs/src/scs/SellCSigma.h:52-141, scs_input.hpp:27-36
//   CSR, CSR_Input                       particle_structs/src/csr/CSR.hpp:37-69, CSR_input.hpp
//   ps::parallel_for(ps, lambda, name)   particle_structs/src/ps_for.hpp:5-31
//   ps::copy<MSpace>(ps)                 particle_structs/src/ps_for.hpp:33-55 (device -> host snapshot)
//   ParticleStructure::getPIDs           particle_structs/src/ps_for.hpp:57-85
//   printFormat                          scs/SellCSigma.h:403-463, csr/CSR.hpp:232-266
//   createMemberViews/getMemberView/destroyViews  support/MemberTypeLibraries.h:33-41
//   PS_LAMBDA, lid_t, gid_t              support/ppMacros.h:3-13, support/ppTypes.h:5-30
// Kokkos::View<T*> is replaced by pumipic::View<T> (device array with shared ownership); a
// Kokkos::TeamPolicy argument is replaced by pumipic::TeamPolicy{league, team} whose team size
// is the chunk height C (64 = one CDNA wavefront).
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <initializer_list>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <map>
#include <vector>
#include "../../include/pumipic_hip.h"

#define PS_LAMBDA [=] __host__ __device__
#define PP_INLINE __host__ __device__ inline
#define PP_DEVICE __device__ inline

namespace pumipic {

typedef int lid_t;
typedef long int gid_t;

// Memory / execution space tags (support/ppTypes.h:13-30: DefaultMemSpace = the default execution space's memory
// space).  There is one space here -- the MI355X the library was initialised on -- so the tags carry no behaviour;
// they exist so that `ParticleStructure<Types, MemSpace>`, `Distributor<Space>` and `PS::execution_space` spell the
// same in user code.
struct DeviceSpace {
  typedef DeviceSpace memory_space;
  typedef DeviceSpace execution_space;
  typedef DeviceSpace device_type;
  static const char* name() { return "HIP (gfx950)"; }
};
struct HostSpace {
  typedef HostSpace memory_space;
  typedef HostSpace execution_space;
  typedef HostSpace device_type;
  static const char* name() { return "Host"; }
};
typedef DeviceSpace DefaultMemSpace;

// support/ppPrint.h:20-38, ppPrint.cpp:5-34: the library's two output streams and the printf-style writers on them
// (printInfo is silent inside device code, as the reference's is)
inline FILE*& pp_stdout_ref() {
  static FILE* f = stdout;
  return f;
}
inline FILE*& pp_stderr_ref() {
  static FILE* f = stderr;
  return f;
}
inline FILE* getStdout() { return pp_stdout_ref(); }
inline FILE* getStderr() { return pp_stderr_ref(); }
inline void setStdout(FILE* out) {
  if (!out) abort();  // assert(out != NULL)
  pp_stdout_ref() = out;
}
inline void setStderr(FILE* err) {
  if (!err) abort();
  pp_stderr_ref() = err;
}
inline void printError(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(getStderr(), "[ERROR]");
  vfprintf(getStderr(), fmt, ap);
  va_end(ap);
}
template <class... Args>
__host__ __device__ inline void printInfo(const char* fmt, Args... args) {  // (device code has no varargs: a pack)
#if !defined(__HIP_DEVICE_COMPILE__)
  if constexpr (sizeof...(Args) == 0)
    fputs(fmt, getStdout());
  else
    fprintf(getStdout(), fmt, args...);
#else
  (void)fmt;
  ((void)args, ...);
#endif
}

inline void pp_check(int rc, const char* what) {
  if (rc != PP_OK) {  // the reference aborts on unrecoverable errors (ppAssert.cpp:10-13)
    fprintf(stderr, "%s failed: %s\n", what, pp_last_error());
    exit(EXIT_FAILURE);
  }
}

// ---------------------------------------------------------------- host copy of a device array
// (what Kokkos::View<T*>::host_mirror_type is to the reference's tests: deviceToHost(view)(i), (i, j) for the [n][k]
//  arrays of a member with k components)
template <class T>
class HostMirror {
 public:
  HostMirror() : ncomp_(1) {}
  explicit HostMirror(size_t n, int ncomp = 1) : h_(std::make_shared<std::vector<T>>(n * (size_t)ncomp)), n_(n), ncomp_(ncomp) {}
  T& operator()(size_t i) const { return (*h_)[i]; }
  T& operator[](size_t i) const { return (*h_)[i]; }
  T& operator()(size_t i, size_t j) const { return (*h_)[j * n_ + i]; }  // (component-major, like the device array)
  size_t size() const { return n_; }
  size_t extent(int) const { return n_; }
  T* data() const { return h_ ? h_->data() : nullptr; }

 private:
  std::shared_ptr<std::vector<T>> h_;
  size_t n_ = 0;
  int ncomp_;
};

// ---------------------------------------------------------------- device array (Kokkos::View<T*>)
template <class T>
class View {
 public:
  typedef T value_type;
  typedef T non_const_value_type;
  typedef HostMirror<T> host_mirror_type;
  typedef HostMirror<T> HostMirror;
  View() : p_(nullptr), n_(0) {}
  explicit View(size_t n) { alloc(n, true); }
  View(const std::string&, size_t n) { alloc(n, true); }
  View(size_t n, T init) {  // Omega_h::Write<T>(n, value): filled on the device, on the library stream
    alloc(n, false);
    if (n) pp_check(fill_(init), "View fill");
  }
  View(size_t n, T init, const std::string&) : View(n, init) {}  // Omega_h::Write<T>(n, value, name)
  View(std::initializer_list<T> l) {                             // Omega_h::Write<T>({a, b, c, d})
    alloc(l.size(), false);
    from_host(l.begin());
  }
  View(size_t n, const std::string&) { alloc(n, true); }         // Omega_h::Write<T>(n, name)
  // Kokkos::View<T*>(Kokkos::ViewAllocateWithoutInitializing(name), n): for arrays whose every entry is written
  // before it is read (a 50 MB fill per 10 M slots otherwise)
  static View uninitialized(size_t n) {
    View v;
    v.alloc(n, false);
    return v;
  }
  static View wrap(T* dev, size_t n) {  // non-owning
    View v;
    v.p_ = dev;
    v.n_ = n;
    return v;
  }
  PP_INLINE T& operator()(size_t i) const { return p_[i]; }
  PP_INLINE T& operator[](size_t i) const { return p_[i]; }
  PP_INLINE size_t size() const { return n_; }
  PP_INLINE T* data() const { return p_; }
  std::vector<T> to_host() const {
    std::vector<T> h(n_);
    if (n_) pp_check(pp_memcpy_d2h(h.data(), p_, n_ * sizeof(T)), "View d2h");
    return h;
  }
  void from_host(const T* h) {
    if (n_) pp_check(pp_memcpy_h2d(p_, h, n_ * sizeof(T)), "View h2d");
  }

 private:
  void alloc(size_t n, bool zero) {
    n_ = n;
    p_ = (T*)pp_malloc(n * sizeof(T));  // (pooled: include/pumipic_hip.h)
    if (!p_) pp_check(PP_EHIP, "View allocation");
    own_ = std::shared_ptr<void>((void*)p_, [](void* q) { (void)pp_free(q); });
    // Kokkos::View zero-initialises; stream-ordered like everything else that touches the array
    if (zero && n) pp_check(pp_memset(p_, 0, n * sizeof(T)), "View memset");
  }
  int fill_(const T& v) {
    if constexpr (sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8) {
      return pp_fill(p_, &v, (int)sizeof(T), n_);
    } else {
      std::vector<T> h(n_, v);
      return pp_memcpy_h2d(p_, h.data(), n_ * sizeof(T));
    }
  }
  T* p_;
  size_t n_;
  std::shared_ptr<void> own_;
};

// support/SupportKK.h:55-110: host array -> device view, device view -> host copy, the last entry of a device view
template <class T>
inline void hostToDevice(View<T> view, const T* data) { view.from_host(data); }
template <class T, class U, class = typename std::enable_if<!std::is_same<T, U>::value>::type>
inline void hostToDevice(View<T> view, const U* data) {  // (a host array of another arithmetic type: converted)
  std::vector<T> t(view.size());
  for (size_t i = 0; i < t.size(); ++i) t[i] = (T)data[i];
  view.from_host(t.data());
}
template <class T>
inline HostMirror<T> deviceToHost(View<T> view) {
  pp_check(pp_sync(), "deviceToHost");
  HostMirror<T> h(view.size());
  if (view.size()) pp_check(pp_memcpy_d2h(h.data(), view.data(), view.size() * sizeof(T)), "deviceToHost");
  return h;
}
template <class ViewT>
inline typename ViewT::value_type getLastValue(ViewT view) {
  typename ViewT::value_type v = typename ViewT::value_type();
  if (view.size() == 0) return v;
  pp_check(pp_sync(), "getLastValue");
  pp_check(pp_memcpy_d2h(&v, view.data() + (view.size() - 1), sizeof(v)), "getLastValue");
  return v;
}

// the timing table of the mirror (pumipic_adjacency.hpp: RecordTime / Timer; defined there, after this header)
inline double op_timer_start();
inline void op_timer_record(const std::string& name, double t0);

// ---------------------------------------------------------------- member type lists
// MemberTypes<T0, T1, ...>: the compile-time list of a particle's members (support/MemberTypes.h:6-62 names the same
// things: ::size, ::memsize = bytes of one particle, ::sizeToIndex<N>() = bytes of the members in front of member N,
// MemberTypeAtIndex<N, List>::type).  C++17 pack expansions instead of recursive templates.

template <typename... Types>
struct MemberTypes {
  static constexpr std::size_t size = sizeof...(Types);
  static constexpr std::size_t memsize = (std::size_t(0) + ... + sizeof(Types));
  template <std::size_t N>
  static constexpr std::size_t sizeToIndex() {
    static_assert(N <= sizeof...(Types), "sizeToIndex: no such member");
    constexpr std::size_t bytes[sizeof...(Types) + 1] = {sizeof(Types)..., 0};
    std::size_t before = 0;
    for (std::size_t i = 0; i < N; ++i) before += bytes[i];
    return before;
  }
};
template <std::size_t N, typename DataTypes>
struct MemberTypeAtIndex;
template <std::size_t N, typename... Types>
struct MemberTypeAtIndex<N, MemberTypes<Types...>> {
  // (std::tuple is only named, never instantiated: array members such as double[3] are fine)
  using type = typename std::tuple_element<N, std::tuple<Types...>>::type;
};
template <class T>
struct BaseType {
  using type = T;
  static constexpr int size = 1;
};
template <class T, std::size_t N>
struct BaseType<T[N]> {
  using type = typename BaseType<T>::type;
  static constexpr int size = (int)N * BaseType<T>::size;
};
template <typename DataTypes>
struct MemberMeta;
template <typename... Types>
struct MemberMeta<MemberTypes<Types...>> {
  static std::vector<int> bytes() { return {(int)sizeof(typename BaseType<Types>::type)...}; }
  static std::vector<int> ncomp() { return {BaseType<Types>::size...}; }
};

// ---------------------------------------------------------------- Segment (ptcls->get<N>())
// support/Segment.h:29-98.  operator()(pid), (pid,i), (pid,i,j), (pid,i,j,k) by the rank of Type, getComponents(pid)
// -> SubSegment (Segment.h:101-176: [i], (), (i), (i,j), (i,j,k) of ONE particle).  Components of a member are
// flattened row-major (T[A][B]: component i*B + j), each component one device array of `stride` slots.
template <class T>
struct ExtentsOf {
  static constexpr int e1 = 1, e2 = 1;
};
template <class T, std::size_t A>
struct ExtentsOf<T[A]> {  // size of one step of the first index = product of the remaining extents
  static constexpr int e1 = BaseType<T>::size, e2 = ExtentsOf<T>::e1;
};
template <typename Type>
class SubSegment;
template <typename Type>
class Segment {
 public:
  using Base = typename BaseType<Type>::type;
  Segment() : p_(nullptr), stride_(0), member_(-1) {}
  Segment(Base* p, long long stride, int member) : p_(p), stride_(stride), member_(member) {}
  PP_INLINE Base& operator()(const int& pid) const { return p_[pid]; }
  PP_INLINE Base& operator()(const int& pid, const int& i) const { return p_[(long long)i * stride_ + pid]; }
  PP_INLINE Base& operator()(const int& pid, const int& i, const int& j) const {
    return p_[(long long)(i * ExtentsOf<Type>::e1 + j) * stride_ + pid];
  }
  PP_INLINE Base& operator()(const int& pid, const int& i, const int& j, const int& k) const {
    return p_[(long long)(i * ExtentsOf<Type>::e1 + j * ExtentsOf<Type>::e2 + k) * stride_ + pid];
  }
  PP_INLINE SubSegment<Type> getComponents(const int& particle_index) const {
    return SubSegment<Type>(p_ + particle_index, stride_);
  }
  int member() const { return member_; }  // which member of the structure this accessor views
  PP_INLINE Base* data() const { return p_; }
  PP_INLINE long long stride() const { return stride_; }

 private:
  Base* p_;
  long long stride_;
  int member_;
};
template <typename Type>
class SubSegment {
 public:
  using Base = typename BaseType<Type>::type;
  PP_INLINE SubSegment(Base* first, long long stride) : p_(first), stride_(stride) {}
  PP_INLINE Base& operator[](const int& i) const { return p_[(long long)i * stride_]; }
  PP_INLINE Base& operator()() const { return p_[0]; }
  PP_INLINE Base& operator()(const int& i) const { return p_[(long long)i * stride_]; }
  PP_INLINE Base& operator()(const int& i, const int& j) const {
    return p_[(long long)(i * ExtentsOf<Type>::e1 + j) * stride_];
  }
  PP_INLINE Base& operator()(const int& i, const int& j, const int& k) const {
    return p_[(long long)(i * ExtentsOf<Type>::e1 + j * ExtentsOf<Type>::e2 + k) * stride_];
  }

 private:
  Base* p_;
  long long stride_;
};

// MTVs: per-member device arrays [ncomp][n] used to hand new particles to a structure
typedef void** MemberTypeViews;
template <typename DataTypes>
struct MTVHeader {  // stored in front of the pointer table
  int n;
};
template <typename DataTypes>
MemberTypeViews createMemberViews(int n) {
  const auto b = MemberMeta<DataTypes>::bytes();
  const auto c = MemberMeta<DataTypes>::ncomp();
  void** v = new void*[DataTypes::size + 1];
  v[DataTypes::size] = (void*)(long)n;
  for (std::size_t m = 0; m < DataTypes::size; ++m) {
    v[m] = pp_malloc((size_t)std::max(n, 1) * c[m] * b[m]);
    if (!v[m]) pp_check(PP_EHIP, "createMemberViews");
  }
  return v;
}
template <typename DataTypes, std::size_t N>
Segment<typename MemberTypeAtIndex<N, DataTypes>::type> getMemberView(MemberTypeViews v) {
  using T = typename MemberTypeAtIndex<N, DataTypes>::type;
  return Segment<T>((typename BaseType<T>::type*)v[N], (long long)(long)v[DataTypes::size], (int)N);
}
template <typename DataTypes>
void destroyViews(MemberTypeViews v) {
  if (!v) return;
  for (std::size_t m = 0; m < DataTypes::size; ++m) (void)pp_free(v[m]);
  delete[] v;
}
// hostToDevice / deviceToHost on a member view of an MTV (particle_structs/test/read_particles.hpp:55-61,82-88): the
// host side is an array of n values of Type (n x k for a member with k components, particle-major), the device side
// component-major [k][n]
template <class Type>
inline void hostToDevice(Segment<Type> seg, const Type* data) {
  typedef typename BaseType<Type>::type B;
  const size_t n = (size_t)seg.stride(), k = (size_t)BaseType<Type>::size;
  if (n == 0) return;
  std::vector<B> t(n * k);
  const B* src = (const B*)data;
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < k; ++j) t[j * n + i] = src[i * k + j];
  pp_check(pp_memcpy_h2d(seg.data(), t.data(), t.size() * sizeof(B)), "hostToDevice");
}
template <class Type>
inline HostMirror<typename BaseType<Type>::type> deviceToHost(Segment<Type> seg) {
  typedef typename BaseType<Type>::type B;
  const size_t n = (size_t)seg.stride(), k = (size_t)BaseType<Type>::size;
  pp_check(pp_sync(), "deviceToHost");
  HostMirror<B> h(n, (int)k);
  if (n) pp_check(pp_memcpy_d2h(h.data(), seg.data(), n * k * sizeof(B)), "deviceToHost");
  return h;
}

// ---------------------------------------------------------------- policy / distributor stand-ins
struct TeamPolicy {
  int league, team;
  TeamPolicy(int l, int t) : league(l), team(t) {}
  int team_size() const { return team; }
};
inline TeamPolicy TeamPolicyAuto(int league_size, int team_size) {  // team_policy.hpp:4-11
  return TeamPolicy(league_size, team_size < 64 ? 64 : team_size);  // wave64: C = 64
}
// The process-wide communicator (MPI_COMM_WORLD of the reference): created on first use from the
// launcher's environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, PP_COMM=rccl|tcp); a
// single-rank communicator when the program was started without a launcher.
inline pp_comm* comm_world() {
  static pp_comm* c = nullptr;
  if (!c) {
    c = pp_comm_create_env();
    if (!c) pp_check(PP_EHIP, "pp_comm_create_env");
  }
  return c;
}
// support/psDistributor.hpp:10-138.  A Distributor names the communicator a structure migrates over and,
// optionally, the subset of ranks it exchanges with (`Distributor(nr, rnks)`: self + the buffered ranks of a
// PICpart in test/pseudoXGCm.cpp:390-396, the two neighbours in particle_structs/test/test_migrate.cpp:60-64).
// The exchange here is one grouped send/recv per peer with a non-zero count whatever the subset, so the subset
// changes no message; what it keeps is the reference's contract -- index(process) is defined for the listed
// ranks only -- which ParticleStructure::migrate checks before anything moves (a particle bound for a rank
// outside the subset is an error here, undefined behaviour there).
template <typename Space = DefaultMemSpace>
class Distributor {
 public:
  Distributor() : comm_(nullptr) {}
  explicit Distributor(pp_comm* c) : comm_(c) {}
  Distributor(int nr, const int* rnks, pp_comm* c = nullptr) : comm_(c) { setRanks(nr, rnks); }
  template <class ViewT, class = decltype(std::declval<const ViewT&>().size()),
            class = decltype(std::declval<const ViewT&>()[0])>
  explicit Distributor(const ViewT& rnks, pp_comm* c = nullptr) : comm_(c) {
    setRanks(rnks);
  }
  template <class OtherSpace>
  Distributor(const Distributor<OtherSpace>& o) : comm_(o.comm_), ranks_(o.ranks_), index_(o.index_) {}
  void setRanks(int nr, const int* rnks) {
    ranks_.assign(rnks, rnks + nr);
    buildMap();
  }
  template <class ViewT>
  void setRanks(const ViewT& rnks) {
    set_from(rnks, 0);
    buildMap();
  }
  void buildMap() {
    index_.clear();
    for (size_t i = 0; i < ranks_.size(); ++i) index_[ranks_[i]] = (int)i;
  }
  pp_comm* comm() const { return comm_ ? comm_ : comm_world(); }
  pp_comm* mpi_comm() const { return comm(); }
  bool isWorld() const { return ranks_.empty(); }
  int num_ranks() const { return isWorld() ? pp_comm_size(comm()) : (int)ranks_.size(); }
  int rank_host(int i) const { return isWorld() ? i : ranks_[(size_t)i]; }
  int rank(int i) const { return rank_host(i); }
  // position of `process` in the subset; -1 when it is not listed (the reference: undefined)
  int index(int process) const {
    if (isWorld()) return process;
    const auto it = index_.find(process);
    return it == index_.end() ? -1 : it->second;
  }

 private:
  template <class>
  friend class Distributor;
  // a device View is read back; any host container with size() and operator[] is copied
  template <class ViewT>
  auto set_from(const ViewT& rnks, int) -> decltype(rnks.to_host(), void()) {
    const auto h = rnks.to_host();
    ranks_.assign(h.begin(), h.end());
  }
  template <class ViewT>
  void set_from(const ViewT& rnks, long) {
    ranks_.resize(rnks.size());
    for (size_t i = 0; i < ranks_.size(); ++i) ranks_[i] = rnks[i];
  }
  pp_comm* comm_;
  std::vector<int> ranks_;
  std::map<int, int> index_;
};

enum PaddingStrategy { PAD_EVENLY = 0, PAD_PROPORTIONALLY = 1, PAD_INVERSELY = 2 };

// ---------------------------------------------------------------- ParticleStructure
template <class DataTypes, typename MemSpace = DefaultMemSpace>
class ParticleStructure {
 public:
  typedef DataTypes Types;
  typedef MemSpace memory_space;
  typedef typename MemSpace::execution_space execution_space;
  typedef typename MemSpace::device_type device_type;
  typedef View<lid_t> kkLidView;
  typedef View<gid_t> kkGidView;
  typedef MemberTypeViews MTVs;
  template <std::size_t N>
  using DataType = typename MemberTypeAtIndex<N, DataTypes>::type;
  template <std::size_t N>
  using Slice = Segment<DataType<N>>;
  typedef typename kkLidView::host_mirror_type kkLidHostMirror;
  typedef typename kkGidView::host_mirror_type kkGidHostMirror;
  typedef ParticleStructure<DataTypes, HostSpace> HostMirror;
  template <typename Space2>
  using Mirror = ParticleStructure<DataTypes, Space2>;

  ParticleStructure() : h_(nullptr), name_("ptcls") {}
  // adopts a library handle (ps::copy<DeviceSpace>(host copy) makes its structure this way)
  ParticleStructure(pp_ps* adopted, const std::string& name) : h_(adopted), name_(name) {}
  virtual ~ParticleStructure() {
    if (h_ && getenv("PP_DUMP_ON_DELETE")) dumpOnDelete(getenv("PP_DUMP_ON_DELETE"));
    if (h_) (void)pp_ps_destroy(h_);
  }
  const std::string& getName() const { return name_; }
  lid_t nElems() const { return info().num_elems; }
  lid_t nPtcls() const { return info().num_ptcls; }
  lid_t capacity() const { return info().capacity; }
  lid_t numRows() const { return info().num_rows; }
  pp_ps* handle() const { return h_; }

  // accessor invalidated by rebuild/migrate, like the reference's (drivers re-get after rebuild)
  template <std::size_t N>
  Slice<N> get() {
    if (nPtcls() == 0) return Slice<N>(nullptr, 0, (int)N);  // (a rank may hold nothing yet: the operators still
                                                             //  need to know WHICH member this accessor names)
    using B = typename BaseType<DataType<N>>::type;
    return Slice<N>((B*)pp_ps_member_ptr(h_, (int)N), pp_ps_member_stride(h_), (int)N);
  }
  virtual void rebuild(kkLidView new_element, kkLidView new_particle_elements = kkLidView(),
                       MTVs new_particle_info = NULL) {
    const double t0 = op_timer_start();
    pp_check(pp_ps_rebuild(h_, new_element.data(), (int)new_particle_elements.size(),
                           new_particle_elements.data(), (const void* const*)new_particle_info),
             "ParticleStructure::rebuild");
    // (the rows the reference's structures add to the timing table: scs/SCS_rebuild.h:177,312, csr/CSR_rebuild.hpp:116)
    op_timer_record(timing_label() + " rebuild", t0);
  }
  std::string timing_label() const { return info().kind == PP_SCS ? name_ : std::string("CSR"); }
  // SellCSigma::migrate / CSR::migrate (scs/SCS_migrate.h:5-222): particles whose new_process is
  // another rank are packed, exchanged over the distributor's communicator and enter the
  // receiver's rebuild as new particles; one rank -> plain rebuild (:20-25)
  virtual void migrate(kkLidView new_element, kkLidView new_process, Distributor<MemSpace> dist = Distributor<MemSpace>(),
                       kkLidView new_particle_elements = kkLidView(), MTVs new_particle_info = NULL) {
    const double t0 = op_timer_start();
    if (!dist.isWorld() && pp_comm_size(dist.comm()) > 1) {  // every leaving particle goes to a rank of the subset
      // The check is COLLECTIVE: a rank that found a violation and left alone would leave its peers waiting
      // in the exchange (round-3 advisor) -- every rank contributes its verdict to one host all-gather and
      // all of them stop together, each naming the ranks at fault.
      const int world = pp_comm_size(dist.comm()), self = pp_comm_rank(dist.comm());
      int bad = 0;
      if (capacity() > 0) {
        std::vector<int> sends((size_t)world, 0);
        pp_check(pp_ps_migrate_count(h_, new_element.data(), new_process.data(), self, world, sends.data()),
                 "ParticleStructure::migrate (send counts)");
        for (int r = 0; r < world; ++r)
          if (r != self && sends[(size_t)r] > 0 && dist.index(r) < 0) {
            fprintf(stderr, "[ERROR] ParticleStructure::migrate: %d particle(s) bound for rank %d, which the "
                            "Distributor does not list\n", sends[(size_t)r], r);
            bad = 1;
          }
      }
      std::vector<int> verdicts((size_t)world, 0);
      pp_check(pp_comm_allgather_host(dist.comm(), &bad, verdicts.data(), (int)sizeof(int)),
               "ParticleStructure::migrate (Distributor verdicts)");
      for (int r = 0; r < world; ++r)
        if (verdicts[(size_t)r]) {
          if (r != self)
            fprintf(stderr, "[ERROR] ParticleStructure::migrate: rank %d has particles for a rank its Distributor "
                            "does not list; stopping with it\n", r);
          bad = 1;
        }
      if (bad) pp_check(PP_EINVAL, "ParticleStructure::migrate (Distributor rank subset)");
    }
    pp_check(pp_ps_migrate_scatter(h_, -1, -1, new_element.data(), new_process.data(), dist.comm(),
                                   (int)new_particle_elements.size(), new_particle_elements.data(),
                                   (const void* const*)new_particle_info, nullptr, 0, nullptr, 0, nullptr,
                                   nullptr, 0.0, 2, 1),
             "ParticleStructure::migrate");
    // (scs/SCS_migrate.h:21,122,218, csr/CSR_migrate.hpp:30,130,225 -- there without the rebuild at its end, which
    //  has a row of its own; here the exchange and the re-layout are one call)
    op_timer_record(timing_label() + " particle migration", t0);
  }
  virtual void printMetrics() const {
    const pp_ps_info_t i = info();
    if (i.kind == PP_SCS) {
      int pc = 0, psl = 0, er = 0;
      pp_check(pp_ps_metrics(h_, &pc, &psl, &er), "printMetrics");
      printf("Metrics 0, C %d, V %d, sigma %d\nNelems %d, Nchunks %d, Nslices %d, Nptcls %d, "
             "Capacity %d\nPadded Cells <Tot %%> %d %.3f\nPadded Slices <Tot %%> %d %.3f\n"
             "Empty Rows <Tot %%> %d %.3f\n",
             i.C, i.V, i.sigma, i.num_elems, i.num_chunks, i.num_slices, i.num_ptcls, i.capacity, pc,
             pc * 100.0 / (i.capacity ? i.capacity : 1), psl, psl * 100.0 / (i.num_slices ? i.num_slices : 1),
             er, er * 100.0 / (i.num_rows ? i.num_rows : 1));
    } else {
      printf("Metrics (Rank 0)\nNumber of Elements %d, Number of Particles %d, Capacity %d\n",
             i.num_elems, i.num_ptcls, i.capacity);
    }
  }
  // printFormat (scs/SellCSigma.h:403-463, csr/CSR.hpp:232-266): the structure's layout as text on stdout --
  // chunks with their elements(gids), slices with the particle mask of every slot; CSR: one `1` per particle
  virtual void printFormat(const char* prefix = "") const {
    const pp_ps_info_t i = info();
    std::vector<int64_t> gids((size_t)std::max(i.num_elems, 1));
    const int ngids = pp_ps_gids_to_host(h_, gids.data());
    std::stringstream ss;
    char buffer[1000];
    if (i.kind == PP_SCS) {
      std::vector<int> off((size_t)i.num_slices + 1), s2c((size_t)std::max(i.num_slices, 1)), r2e((size_t)std::max(i.num_rows, 1));
      std::vector<unsigned char> mask((size_t)std::max(i.capacity, 1));
      pp_check(pp_ps_layout_to_host(h_, off.data(), s2c.data(), r2e.data(), nullptr, mask.data(), nullptr), "printFormat");
      snprintf(buffer, sizeof(buffer), "%s\nParticle Structures Sell-C-Sigma C: %d sigma: %d V: %d.\nNumber of Elements: %d.\n"
               "Number of Particles: %d.\nNumber of Chunks: %d.\nNumber of Slices: %d.\n", prefix, i.C, i.sigma, i.V,
               i.num_elems, i.num_ptcls, i.num_chunks, i.num_slices);
      ss << buffer;
      int last_chunk = -1;
      for (int sl = 0; sl < i.num_slices; ++sl) {
        const int chunk = s2c[(size_t)sl];
        if (chunk != last_chunk) {
          last_chunk = chunk;
          ss << "  Chunk " << chunk << ". Elements" << (ngids > 0 ? "(GID)" : "") << ":";
          for (int row = chunk * i.C; row < (chunk + 1) * i.C; ++row) {
            const int elem = r2e[(size_t)row];
            ss << " " << elem;
            if (ngids > 0) ss << "(" << (elem < ngids ? (long)gids[(size_t)elem] : -1L) << ")";  // (padding rows: no gid)
          }
          ss << "\n";
        }
        ss << "    Slice " << sl;
        for (int j = off[(size_t)sl]; j < off[(size_t)sl + 1]; ++j) {
          if ((j - off[(size_t)sl]) % i.C == 0) ss << " |";
          ss << " " << (int)mask[(size_t)j];
        }
        ss << "\n";
      }
    } else {
      std::vector<int> off((size_t)i.num_elems + 1);
      pp_check(pp_ps_layout_to_host(h_, off.data(), nullptr, nullptr, nullptr, nullptr, nullptr), "printFormat");
      snprintf(buffer, sizeof(buffer), "%s\nParticle Structures CSR\nNumber of Elements: %d.\nNumber of Particles: %d.",
               prefix, i.num_elems, i.num_ptcls);
      ss << buffer;
      for (int e = 1; e <= i.num_elems; ++e) {
        if (off[(size_t)e] == off[(size_t)e - 1]) continue;
        if (ngids > 0)
          snprintf(buffer, sizeof(buffer), "\n  Element %2d(%2ld) |", e - 1, (long)gids[(size_t)e - 1]);
        else
          snprintf(buffer, sizeof(buffer), "\n  Element %2d |", e - 1);
        ss << buffer;
        for (int j = off[(size_t)e - 1]; j < off[(size_t)e]; ++j) ss << " 1";
      }
      ss << "\n";
    }
    std::cout << ss.str();
  }
  // getPIDs (ps_for.hpp:57-85): offsets[e] .. offsets[e+1] index the slots of element e's live particles in pids
  // (the reference fills each element's range in the order of its atomics; here in slot order)
  template <typename ViewT>
  void getPIDs(ViewT& pids, ViewT& offsets) {
    offsets = ViewT("offsets", (size_t)nElems() + 1);
    pids = ViewT("pids", (size_t)std::max(nPtcls(), 1));
    pp_check(pp_ps_get_pids(h_, offsets.data(), pids.data()), "getPIDs");
    if (nPtcls() == 0) pids = ViewT();
  }
  pp_ps_info_t info() const {
    pp_ps_info_t i;
    pp_check(pp_ps_info(h_, &i), "pp_ps_info");
    return i;
  }
  // structure->parallel_for(lambda, name) (scs/SellCSigma.h:110-111): the member form of ps::parallel_for
  template <typename FunctionType>
  void parallel_for(FunctionType& fn, std::string name = "");

 protected:
  // PP_DUMP_ON_DELETE=<prefix>: a structure writes itself to <prefix>_ps_<name>_r<rank>_<serial>_{meta.txt, mask.u8,
  // elem.i32, m<k>.bin} when it is deleted -- how the tests read the final state of a driver whose source they
  // must not touch (the reference's own test/pseudoXGCm.cpp, compiled unchanged: tests/test_gpu_refdrivers.py)
  void dumpOnDelete(const char* prefix) {
    static int serial = 0;
    const pp_ps_info_t i = info();
    const auto b = MemberMeta<DataTypes>::bytes();
    const auto c = MemberMeta<DataTypes>::ncomp();
    char base[1024];
    snprintf(base, sizeof(base), "%s_ps_%s_r%d_%d", prefix, name_.c_str(), pp_comm_rank(comm_world()), serial++);
    auto put = [&](const std::string& suffix, const void* data, size_t bytes) {
      FILE* f = fopen((std::string(base) + suffix).c_str(), "wb");
      if (!f) return;
      fwrite(data, 1, bytes, f);
      fclose(f);
    };
    const size_t cap = (size_t)std::max(i.capacity, 1);
    std::vector<unsigned char> mask(cap);
    std::vector<int> slot_elem(cap);
    if (i.capacity > 0)
      pp_check(pp_ps_layout_to_host(h_, nullptr, nullptr, nullptr, nullptr, mask.data(), slot_elem.data()), "dump");
    put("_mask.u8", mask.data(), (size_t)i.capacity);
    put("_elem.i32", slot_elem.data(), (size_t)i.capacity * sizeof(int));
    for (std::size_t m = 0; m < DataTypes::size; ++m) {
      std::vector<char> buf((size_t)std::max<int64_t>(i.stride, 1) * c[m] * b[m]);
      if (i.capacity > 0) pp_check(pp_ps_member_to_host(h_, (int)m, buf.data()), "dump");
      put("_m" + std::to_string(m) + ".bin", buf.data(), (size_t)i.stride * c[m] * b[m]);
    }
    std::stringstream meta;
    meta << i.capacity << " " << (long long)i.stride << " " << i.num_ptcls << " " << i.num_elems << " " << DataTypes::size;
    for (std::size_t m = 0; m < DataTypes::size; ++m) meta << " " << b[m] << " " << c[m];
    meta << "\n";
    put("_meta.txt", meta.str().data(), meta.str().size());
  }
  pp_ps* h_;
  std::string name_;
};

template <class DataTypes, typename MemSpace = DefaultMemSpace>
class SellCSigma;
template <class DataTypes, typename MemSpace = DefaultMemSpace>
class SCS_Input {
 public:
  typedef View<lid_t> kkLidView;
  typedef View<gid_t> kkGidView;
  SCS_Input(TeamPolicy& p, lid_t sigma, lid_t V_, lid_t ne_, lid_t np_, kkLidView ppe_,
            kkGidView eg, kkLidView pes = kkLidView(), MemberTypeViews info = NULL)
      : policy(p), sig(sigma), V(V_), ne(ne_), np(np_), ppe(ppe_), e_gids(eg), particle_elms(pes),
        p_info(info) {
    name = "ptcls";
  }
  bool always_realloc = false;
  double minimize_size = .8;
  double shuffle_padding = 0.1;
  double extra_padding = 0.05;
  PaddingStrategy padding_strat = PAD_EVENLY;
  std::string name;
  TeamPolicy policy;
  lid_t sig, V, ne, np;
  kkLidView ppe;
  kkGidView e_gids;
  kkLidView particle_elms;
  MemberTypeViews p_info;
};

template <class DataTypes, typename MemSpace>
class SellCSigma : public ParticleStructure<DataTypes, MemSpace> {
 public:
  typedef View<lid_t> kkLidView;
  typedef View<gid_t> kkGidView;
  typedef MemberTypeViews MTVs;
  typedef SCS_Input<DataTypes, MemSpace> Input_T;
  SellCSigma(TeamPolicy& p, lid_t sigma, lid_t vertical_chunk_size, lid_t num_elements,
             lid_t num_particles, kkLidView particles_per_element, kkGidView element_gids,
             kkLidView particle_elements = kkLidView(), MTVs particle_info = NULL) {
    construct(p.team_size(), sigma, vertical_chunk_size, num_elements, num_particles,
              particles_per_element, element_gids, particle_elements, particle_info, PAD_EVENLY, 0.1,
              0.05);
  }
  SellCSigma(Input_T& in) {
    this->name_ = in.name;
    construct(in.policy.team_size(), in.sig, in.V, in.ne, in.np, in.ppe, in.e_gids, in.particle_elms,
              in.p_info, in.padding_strat, in.shuffle_padding, in.extra_padding);
  }
  lid_t C() const { return this->info().C; }
  lid_t V() const { return this->info().V; }
  // scs/SellCSigma.h:92: whether rebuild first tries to keep the layout and move only the particles that change
  // element (reshuffle, SCS_rebuild.h:4-119); default on (:236)
  void setShuffling(bool newS) { pp_check(pp_ps_set_shuffling(this->h_, newS ? 1 : 0), "setShuffling"); }

 private:
  void construct(int C, lid_t sigma, lid_t V, lid_t ne, lid_t np, kkLidView ppe, kkGidView gids,
                 kkLidView particle_elements, MTVs particle_info, int pad, double shuffle,
                 double extra) {
    const auto b = MemberMeta<DataTypes>::bytes();
    const auto c = MemberMeta<DataTypes>::ncomp();
    std::vector<lid_t> ppe_h = ppe.to_host();
    std::vector<gid_t> g_h = gids.to_host();
    std::vector<int64_t> g64(g_h.begin(), g_h.end());
    const bool with_info = particle_elements.size() > 0 && particle_info != NULL;
    if (with_info) std::fill(ppe_h.begin(), ppe_h.end(), 0);  // particles enter through rebuild
    this->h_ = pp_ps_create_scs(C, sigma, V, ne, with_info ? 0 : np, ppe_h.data(),
                                g64.empty() ? nullptr : g64.data(), pad, shuffle, extra,
                                (int)DataTypes::size, b.data(), c.data(), nullptr, nullptr);
    if (!this->h_) pp_check(PP_EHIP, "SellCSigma construction");
    if (with_info) {  // initSCSData (SCS_buildFns.h:205-232): device-side placement
      kkLidView none(1);
      pp_check(pp_ps_rebuild(this->h_, none.data(), (int)particle_elements.size(),
                             particle_elements.data(), (const void* const*)particle_info),
               "SellCSigma initial particles");
    }
  }
};

// csr/CSR_input.hpp:10-42
template <class DataTypes, typename MemSpace = DefaultMemSpace>
class CSR_Input {
 public:
  typedef View<lid_t> kkLidView;
  typedef View<gid_t> kkGidView;
  typedef MemberTypeViews MTVs;
  typedef TeamPolicy PolicyType;
  CSR_Input(PolicyType& p, lid_t num_elements, lid_t num_particles, kkLidView particles_per_element,
            kkGidView element_gids, kkLidView particle_elements = kkLidView(), MTVs particle_info = NULL)
      : policy(p), ne(num_elements), np(num_particles), ppe(particles_per_element), e_gids(element_gids),
        particle_elems(particle_elements), p_info(particle_info) {
    name = "ptcls";
  }
  bool always_realloc = false;   // (the library's swap buffers grow on demand and are kept: nothing to choose)
  double minimize_size = 0.8;
  double padding_amount = 1.05;  // capacity = padding_amount * num_ptcls
  std::string name;
  PolicyType policy;
  lid_t ne, np;
  kkLidView ppe;
  kkGidView e_gids;
  kkLidView particle_elems;
  MTVs p_info;
};

template <class DataTypes, typename MemSpace = DefaultMemSpace>
class CSR : public ParticleStructure<DataTypes, MemSpace> {
 public:
  typedef View<lid_t> kkLidView;
  typedef View<gid_t> kkGidView;
  typedef MemberTypeViews MTVs;
  typedef CSR_Input<DataTypes, MemSpace> Input_T;
  CSR(Input_T& in) {  // csr/CSR.hpp:146-156
    this->name_ = in.name;
    construct(in.ne, in.np, in.ppe, in.e_gids, in.particle_elems, in.p_info, in.padding_amount);
  }
  CSR(TeamPolicy&, lid_t num_elements, lid_t num_particles, kkLidView particles_per_element,
      kkGidView element_gids, kkLidView particle_elements = kkLidView(), MTVs particle_info = NULL) {
    construct(num_elements, num_particles, particles_per_element, element_gids, particle_elements, particle_info, 1.05);
  }

 private:
  void construct(lid_t num_elements, lid_t num_particles, kkLidView particles_per_element, kkGidView element_gids,
                 kkLidView particle_elements, MTVs particle_info, double padding_amount) {
    const auto b = MemberMeta<DataTypes>::bytes();
    const auto c = MemberMeta<DataTypes>::ncomp();
    std::vector<lid_t> ppe_h = particles_per_element.to_host();
    std::vector<gid_t> g_h = element_gids.to_host();
    std::vector<int64_t> g64(g_h.begin(), g_h.end());
    const bool with_info = particle_elements.size() > 0 && particle_info != NULL;
    if (with_info) std::fill(ppe_h.begin(), ppe_h.end(), 0);
    this->h_ = pp_ps_create_csr(num_elements, with_info ? 0 : num_particles, ppe_h.data(),
                                g64.empty() ? nullptr : g64.data(), padding_amount, (int)DataTypes::size,
                                b.data(), c.data(), nullptr, nullptr);
    if (!this->h_) pp_check(PP_EHIP, "CSR construction");
    if (with_info) {
      kkLidView none(1);
      pp_check(pp_ps_rebuild(this->h_, none.data(), (int)particle_elements.size(),
                             particle_elements.data(), (const void* const*)particle_info),
               "CSR initial particles");
    }
  }
};

// ---------------------------------------------------------------- parallel_for
// One thread per slot; fn(element_id, particle_id, mask) is called for EVERY slot the reference
// would visit, masked ones included (SellCSigma.h:545-552, CSR.hpp:198-208).  Sell-C-sigma with chunk
// height 64 (the wave): the 64 slots of a wave are the 64 rows of one column of one chunk -- the chunk comes from
// a wave-uniform load, the element from the row table (pp_ps_iteration); else from the slot -> element table.
template <class Fn>
__global__ void ps_parallel_for_kernel(int capacity, const int* __restrict__ slot_elem,
                                       const unsigned char* __restrict__ mask, Fn fn) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = slot_elem[pid];
  if (e < 0) return;
  fn(e, pid, (int)mask[pid]);
}
template <class Fn>
__global__ void ps_parallel_for_kernel_c64(int capacity, const int* __restrict__ group_chunk,
                                           const int* __restrict__ row_to_element,
                                           const unsigned char* __restrict__ mask, Fn fn) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int c = __builtin_amdgcn_readfirstlane(group_chunk[pid >> 6]);
  const int e = row_to_element[(c << 6) + (pid & 63)];
  fn(e, pid, (int)mask[pid]);
}
template <typename FunctionType, typename DataTypes, typename MemSpace>
void parallel_for(ParticleStructure<DataTypes, MemSpace>* ps, FunctionType& fn, std::string = "") {
  if (!ps || !ps->handle()) {
    fprintf(stderr, "Structure does not support parallel for\n");
    throw 1;  // ps_for.hpp:28-30
  }
  pp_ps_iter_t it;
  pp_check(pp_ps_iteration(ps->handle(), &it), "pp_ps_iteration");
  if (ps->nPtcls() == 0 || it.capacity == 0) return;  // SellCSigma.h:529
  const int block = 256, grid = (it.capacity + block - 1) / block;
  if (it.group_chunk)
    hipLaunchKernelGGL(ps_parallel_for_kernel_c64<FunctionType>, dim3(grid), dim3(block), 0,
                       (hipStream_t)pp_stream(), it.capacity, it.group_chunk, it.row_to_element, it.mask, fn);
  else
    hipLaunchKernelGGL(ps_parallel_for_kernel<FunctionType>, dim3(grid), dim3(block), 0,
                       (hipStream_t)pp_stream(), it.capacity, it.slot_elem, it.mask, fn);
}

template <class DataTypes, typename MemSpace>
template <typename FunctionType>
void ParticleStructure<DataTypes, MemSpace>::parallel_for(FunctionType& fn, std::string name) {
  pumipic::parallel_for(this, fn, name);
}

// ---------------------------------------------------------------- ps::copy<MSpace> (ps_for.hpp:33-55)
// The reference copies a structure into another memory space (SellCSigma::copy<MSpace>, scs/SellCSigma.h:336-391:
// a deep copy of every layout array and member view) -- its tests read a device structure back with it and run
// the same PS_LAMBDA on the host copy (particle_structs/test/test_structure.cpp).  Here: a HOST SNAPSHOT with the
// same read interface (nElems / nPtcls / capacity / numRows, get<N>() -> Segment over host memory, the layout arrays,
// ps::parallel_for on the host); rebuild / migrate stay with the device structure.
template <class DataTypes>
class ParticleStructure<DataTypes, HostSpace> {
 public:
  typedef DataTypes Types;
  typedef HostSpace memory_space;
  template <std::size_t N>
  using DataType = typename MemberTypeAtIndex<N, DataTypes>::type;
  template <std::size_t N>
  using Slice = Segment<DataType<N>>;
  template <class MemSpace>
  explicit ParticleStructure(ParticleStructure<DataTypes, MemSpace>* old) : name_(old->getName()), i_(old->info()) {
    const auto b = MemberMeta<DataTypes>::bytes();
    const auto c = MemberMeta<DataTypes>::ncomp();
    data_.resize(DataTypes::size);
    for (std::size_t m = 0; m < DataTypes::size; ++m) {
      data_[m].resize((size_t)std::max<int64_t>(i_.stride, 1) * c[m] * b[m]);
      if (i_.capacity > 0) pp_check(pp_ps_member_to_host(old->handle(), (int)m, data_[m].data()), "ps::copy (members)");
    }
    const bool scs = i_.kind == PP_SCS;
    offsets.resize(scs ? (size_t)i_.num_slices + 1 : (size_t)i_.num_elems + 1);
    slice_to_chunk.resize(scs ? (size_t)std::max(i_.num_slices, 1) : 1);
    row_to_element.resize(scs ? (size_t)std::max(i_.num_rows, 1) : 1);
    element_to_row.resize(scs ? (size_t)std::max(i_.num_rows, 1) : 1);
    particle_mask.resize((size_t)std::max(i_.capacity, 1));
    slot_element.resize((size_t)std::max(i_.capacity, 1));
    pp_check(pp_ps_layout_to_host(old->handle(), offsets.data(), scs ? slice_to_chunk.data() : nullptr,
                                  scs ? row_to_element.data() : nullptr, scs ? element_to_row.data() : nullptr,
                                  particle_mask.data(), slot_element.data()), "ps::copy (layout)");
    // the layout of the copy as the library holds it: a deep copy on the device, which ps::copy<DeviceSpace>(this)
    // turns into the new device structure (with the members as they are HERE at that time)
    pp_ps* t = pp_ps_clone(old->handle());
    if (!t) pp_check(PP_EHIP, "ps::copy (device twin)");
    twin_ = std::shared_ptr<pp_ps>(t, [](pp_ps* q) { (void)pp_ps_destroy(q); });
  }
  const std::string& getName() const { return name_; }
  lid_t nElems() const { return i_.num_elems; }
  lid_t nPtcls() const { return i_.num_ptcls; }
  lid_t capacity() const { return i_.capacity; }
  lid_t numRows() const { return i_.num_rows; }
  const pp_ps_info_t& info() const { return i_; }
  template <std::size_t N>
  Slice<N> get() {
    if (nPtcls() == 0) return Slice<N>(nullptr, 0, (int)N);
    using B = typename BaseType<DataType<N>>::type;
    return Slice<N>((B*)data_[N].data(), i_.stride, (int)N);
  }
  // a new device structure with this copy's layout and members (SellCSigma::copy<MSpace> towards the device)
  pp_ps* to_device() const {
    pp_ps* n = pp_ps_clone(twin_.get());
    if (!n) pp_check(PP_EHIP, "ps::copy (to the device)");
    if (i_.capacity > 0)
      for (std::size_t m = 0; m < DataTypes::size; ++m)
        pp_check(pp_ps_member_from_host(n, (int)m, data_[m].data()), "ps::copy (members to the device)");
    return n;
  }
  // layout arrays of the snapshot (SellCSigma.h:186-215 / CSR.hpp:92-93), host memory
  std::vector<lid_t> offsets, slice_to_chunk, row_to_element, element_to_row, slot_element;
  std::vector<unsigned char> particle_mask;

 private:
  std::string name_;
  pp_ps_info_t i_;
  std::vector<std::vector<char>> data_;
  std::shared_ptr<pp_ps> twin_;
};
template <class DataTypes>
using HostParticleStructure = ParticleStructure<DataTypes, HostSpace>;

// copy<MSpace>(structure): device -> host snapshot, host snapshot -> a new device structure; the same space on both
// sides is refused as in the reference (scs/SellCSigma.h:339-342)
template <typename MSpace, typename DataTypes, typename MemSpace>
ParticleStructure<DataTypes, MSpace>* copy(ParticleStructure<DataTypes, MemSpace>* old) {
  if (!old) {
    fprintf(stderr, "Structure does not support copy\n");
    throw 1;  // ps_for.hpp:52-54
  }
  if constexpr (std::is_same<MSpace, MemSpace>::value) {
    fprintf(stderr, "Copy to same memory space not supported\n");
    exit(EXIT_FAILURE);
    return nullptr;
  } else if constexpr (std::is_same<MSpace, HostSpace>::value) {
    if (!old->handle()) {
      fprintf(stderr, "Structure does not support copy\n");
      throw 1;
    }
    return new ParticleStructure<DataTypes, HostSpace>(old);
  } else {
    return new ParticleStructure<DataTypes, MSpace>(old->to_device(), old->getName());
  }
}
// the same PS_LAMBDA on the host copy: every slot in slot order
template <typename FunctionType, typename DataTypes>
void parallel_for(ParticleStructure<DataTypes, HostSpace>* ps, FunctionType& fn, std::string = "") {
  if (!ps) {
    fprintf(stderr, "Structure does not support parallel for\n");
    throw 1;
  }
  if (ps->nPtcls() == 0) return;
  for (lid_t pid = 0; pid < ps->capacity(); ++pid) {
    const lid_t e = ps->slot_element[(size_t)pid];
    if (e < 0) continue;
    fn(e, pid, (int)ps->particle_mask[(size_t)pid]);
  }
}

}  // namespace pumipic

namespace particle_structs = pumipic;

// The reference's <particle_structs.hpp> brings MPI and Kokkos with it, and its own tests and drivers spell those names
// after including nothing else (particle_structs/test/test_types.hpp:14-16): the facades over the C-ABI and HIP
#include "pumipic_mpi.hpp"
#ifndef PP_ADJACENCY_IN_PROGRESS  // (else pumipic_adjacency.hpp includes it when its own declarations are complete)
#include "compat/Kokkos_Core.hpp"
#endif
