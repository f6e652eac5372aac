// pp_internal.hpp -- shared host-side internals of libpumipic_hip.so (not part of the C-ABI).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "../../include/pumipic_hip.h"

// Laboratory switches.  The A/B knobs the kernels were tuned with are gone from the library: the variants that lost
// are deleted (profiles/r0x_rejected_* keep their patches and numbers).  What is left under this macro forces a
// FALLBACK path that is still live code -- the checked rebuild a structure takes when a buffer must grow, the
// atomic scatter of a ring map without a transpose, the unpacked Moeller-Trumbore walk of a mesh whose records
// cannot be packed -- so that tests can drive it at scale.  It reads the environment only in a build with
// -DPP_LAB (make lab); the shipped library compiles every one of them to a constant.
#ifdef PP_LAB
#define PP_LAB_ENV(name) getenv(name)
#else
#define PP_LAB_ENV(name) ((const char*)nullptr)
#endif

namespace pp {

void set_error(const std::string& msg);
unsigned long long next_version();  // pp_runtime.hip: process-wide monotonic stamp
hipStream_t stream();
// pp_scatter.hip: a device range was freed / overwritten through the C-ABI -- forget the gather
// form of any gyro ring map living there
void gyro_map_invalidate(const void* dev, size_t bytes);
void gyro_map_mesh_gone(const void* mesh);
// pp_scatter.hip: gyroScatter for `nmaps` ring maps from an explicit per-element count array (the
// histogram of a rebuild that is still in flight); forgets the ring accumulation kept for reuse
bool initialised();
// pp_runtime.hip: the device memory pool behind pp_malloc / pp_free
void* pool_alloc(size_t bytes);
int pool_free(void* dev);
int pool_trim();
void pool_set_limit(size_t bytes);
void pool_stats(size_t* live_bytes, size_t* cached_bytes, long long* hits, long long* misses);
// pp_mesh.hip: derive (once) and return the edges of a tet mesh; PP_EINVAL for a 2-D mesh

#define PP_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      pp::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ + \
                    ":" + std::to_string(__LINE__) + ")");                              \
      return PP_EHIP;                                                                   \
    }                                                                                   \
  } while (0)

#define PP_HIP_CHECK_NULL(expr)                                                         \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      pp::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ + \
                    ":" + std::to_string(__LINE__) + ")");                              \
      return nullptr;                                                                   \
    }                                                                                   \
  } while (0)

#define PP_REQUIRE(cond, msg)        \
  do {                               \
    if (!(cond)) {                   \
      pp::set_error(msg);            \
      return PP_EINVAL;              \
    }                                \
  } while (0)

#define PP_LAUNCH_CHECK() PP_HIP_CHECK(hipGetLastError())

// Simple owning device buffer (grow-only reuse to avoid hipMalloc on the hot path).
struct DevBuf {
  void* p = nullptr;     // usable pointer (base + skew)
  void* base = nullptr;  // what hipMalloc returned
  size_t bytes = 0;      // usable bytes from p
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), base(o.base), bytes(o.bytes) {
    o.p = o.base = nullptr;
    o.bytes = 0;
  }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      base = o.base;
      bytes = o.bytes;
      o.p = o.base = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  void release() {
    if (base) (void)hipFree(base);
    p = base = nullptr;
    bytes = 0;
  }
  // ensure capacity; contents are NOT preserved when it grows.  `skew` (a multiple of 512 B) shifts
  // the usable pointer off the allocator's alignment: arrays that are streamed side by side should
  // not all start at the same position of the HBM channel interleave.
  hipError_t reserve(size_t n, size_t skew = 0) {
    if (n <= bytes) return hipSuccess;
    static const bool dbg = PP_LAB_ENV("PP_ALLOC_DEBUG") != nullptr;  // (tools/r04_coldsteps.py: who re-allocates when)
    if (dbg) fprintf(stderr, "pp alloc: %zu -> %zu bytes\n", bytes, n + n / 8 + 256);
    release();
    size_t want = n + n / 8 + 256;
    hipError_t e = hipMalloc(&base, want + skew);
    if (e != hipSuccess) {  // blocks cached by the pp_malloc pool may be what is in the way
      (void)hipGetLastError();
      (void)pool_trim();
      e = hipMalloc(&base, want + skew);
    }
    if (e == hipSuccess) {
      bytes = want;
      p = (char*)base + skew;
    }
    return e;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(base, o.base);
    std::swap(bytes, o.bytes);
  }
};

constexpr int kBlock = 256;
constexpr int kTileP = 8;  // particles per row handled by one thread of the row-tiled kernels
static_assert(32 % kTileP == 0, "a tile group of the histogram (32 columns, the stay mask) is a whole number of tiles: "
                                "the over-full row's own blocks start on a group boundary (pp_ps::hot)");
inline unsigned grid_for(size_t n, int block = kBlock) {
  return (unsigned)((n + (size_t)block - 1) / (size_t)block);
}

}  // namespace pp

// ---------------------------------------------------------------------------------------------
// packed per-element walk records (DESIGN.md "data layout in HBM")
// tri : 64 B = half a 128-B line ; tet : 128 B = one line.  nbr[i] = element across local side i
// (Omega_h template order), -1 when that side is exposed.
struct alignas(16) pp_tri_rec {
  double xy[3][2];  // 48
  int nbr[3];       // 12
  int class_id;     // 4
};
struct alignas(16) pp_tet_rec {
  double xyz[4][3];  // 96
  int nbr[4];        // 16
  double vol;        // 8  measure_elements_real value
  int class_id;      // 4
  unsigned mt_code;  // 4  Moeller-Trumbore face codes: per face the stored side's vertices as tet-local indices (pp_mesh.hip)
};
static_assert(sizeof(pp_tri_rec) == 64, "tri record must be 64 B");
static_assert(sizeof(pp_tet_rec) == 128, "tet record must be 128 B");

struct pp_mesh {
  unsigned long long uid = 0;  // unique per mesh object (caches keyed on a mesh survive a new mesh at the same address)
  int dim = 0, nverts = 0, nelems = 0, nsides = 0;
  double tol = 0;  // compute_tolerance_from_area
  double unmoved_sq = 0;  // min{s : sqrt(s) >= tol}: norm(v) < tol  <=>  v.v < unmoved_sq
  // host copies (setup + to_host queries)
  std::vector<double> coords, elem_measure;
  std::vector<int> elem2verts, class_id, elem2sides, side2verts, side2elems_off, side2elems,
      dual_off, dual_elems, vert2elems_off, vert2elems;
  std::vector<signed char> side_exposed;
  // edges of a tet mesh (entity dimension 1 of a 3-D mesh; in 2-D the edges are the sides), derived on first
  // use (pp_mesh_edges): Omega_h's template order (0,1),(1,2),(2,0),(0,3),(1,3),(2,3), numbered in
  // first-seen order over (element, local edge), edge2verts in the orientation first seen
  bool mt_packed_ok = true;  // no element has the same neighbour behind two faces (k_search_mt3)
  bool edges_ready = false;
  int nedges = 0;
  std::vector<int> elem2edges, edge2verts, edge2elems_off, edge2elems;
  pp::DevBuf d_elem2edges, d_edge2verts, d_edge2elems_off, d_edge2elems;
  // device arrays
  pp::DevBuf d_coords, d_elem2verts, d_class_id, d_elem2sides, d_side2verts, d_side2elems_off,
      d_side2elems, d_side_exposed, d_elem_measure, d_dual_off, d_dual_elems, d_vert2elems_off,
      d_vert2elems, d_records;
};

namespace pp {
int mesh_edges(const pp_mesh* mesh);
}
namespace pp {
struct GyroRide;
}
// First record of row r of chunk c when the staging records are row-major inside a chunk (pp_ps::rec_rm).  The row
// pitch is the chunk width rounded up to a multiple of FOUR and every chunk gets three spare columns (3 C records), so
// that every row starts on a 128-byte line and a line holds whole records of one row only: the 64-B records of columns
// (2j, 2j+1), the 32-B records of the pseudoXGCm type (round 5) of columns (4j .. 4j+3) -- the record-fed push fetches
// them together.  The record buffer holds capacity + 3 C * nchunks records.
constexpr int kRecSpareCols = 3;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int pp_rec_pitch(int w) { return (w + 3) & ~3; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int pp_rec_row0(int chunk_start, int c, int r, int w, int C) {
  return chunk_start + kRecSpareCols * C * c + r * pp_rec_pitch(w);
}
namespace pp {
// (see pp_ps::hot) columns [c1p, w) of row `row` of chunk `chunk`, whose first slot is `start`, hold particles of
// that row only
struct HotRow {
  int on = 0;
  int chunk = 0, row = 0;
  int c1p = 0;    // multiple of 32: a tile group of the histogram, four 8-column blocks of the pack
  int w = 0;      // the row's particle count
  int start = 0;  // chunk_start[chunk]
};
}  // namespace pp

struct pp_ps {
  int kind = PP_SCS;
  int num_elems = 0, num_ptcls = 0, capacity = 0, num_rows = 0;
  int C = 1, C_max = 1, V = 1024, sigma = 1, num_chunks = 0, num_slices = 0;
  int pad_strat = PP_PAD_EVENLY;
  double shuffle_padding = 0.1, extra_padding = 0.05, minimize_size = 0.8, padding_amount = 1.05;
  int num_empty_elements = 0;
  int nmembers = 0;
  std::vector<int> member_bytes, member_ncomp;
  std::vector<int> member_map;  // logical member -> storage index (pp_ps_swap_members)
  int64_t stride = 0;           // allocated slots per component of the live buffers
  int64_t swap_stride = 0;
  std::vector<pp::DevBuf> data, swap;
  bool has_gids = false;
  pp::DevBuf d_gids;  // element -> gid (num_elems)
  // gid -> element, the structure's own map (element_gid_to_lid of the reference, SCS_migrate.h:181-187), as a
  // dense table over [0, max gid]; built at construction when the gids are not 0..ne-1 (a part of a
  // partitioned mesh), used by the migration when the caller passes no table of its own
  bool gids_identity = true;
  int64_t n_gid2lid = 0;
  pp::DevBuf d_gid2lid;
  // layout (device)
  pp::DevBuf d_offsets, d_slice_to_chunk, d_row_to_element, d_element_to_row, d_mask, d_slot_elem;
  // slot -> parent element: the row-tiled kernels of the time step never read it, so the SCS re-layout leaves it
  // unwritten (40 MB per 10 M slots) and pp::slot_elem() fills it when something asks
  mutable bool slot_elem_valid = true;
  mutable bool slot_elem_used = false;  // somebody asked for the table of the CURRENT layout (pp::slot_elem)
  // 64-slot group -> chunk (chunk height 64; pp_ps_iteration): filled on first use after a re-layout
  mutable pp::DevBuf d_group_chunk;
  mutable bool group_chunk_valid = false;
  // SCS row tiles for the row-major hot kernels: tile = (chunk, first p), kTileP columns wide.
  // A chunk's slots are contiguous: slot = chunk_start[c] + row_in_chunk + p*C, p < chunk_width[c]
  pp::DevBuf d_chunk_start, d_chunk_width, d_tiles, d_ntiles;
  int ntiles_max = 0;
  // live particles per element, kept current by construction/rebuild (gyroScatter reads it)
  pp::DevBuf d_elem_count;
  bool elem_count_valid = false;
  // stamp of the current particle->element assignment (unique across structures); bumped by every
  // construction and rebuild, lets pp_gyro_scatter reuse the ring accumulation of the previous call
  unsigned long long version = 0;
  unsigned long long last_max_key = ~0ull;  // largest layout sort key of the previous rebuild (~0 = unknown)
  int tile_p = pp::kTileP;  // columns per tile (a constant since round 5: pp::kTileP)
  // SellCSigma::tryShuffling (SellCSigma.h:92,213,236): a rebuild keeps the layout and moves only the
  // particles that change element when every row's arrivals fit its holes (SCS_rebuild.h:4-119)
  // 0 = always the full re-layout (setShuffling(false)); 1 (default) = the reference's decision
  int shuffle_mode = 1;
  long long n_reshuffles = 0, n_full_rebuilds = 0;  // how the rebuilds of this structure ended
  long long n_from_records = 0;  // full re-layouts whose first pass read the previous one's records (lazy_rec == 3)
  pp::DevBuf d_eslot0;  // first slot of every element's row in the CURRENT layout
  // a member whose content is logically all zero but has not been written yet (storage index, -1 =
  // none): x_tgt after a fused updatePtclPositions of the in-place rebuild.  Cleared without a pass
  // when the next fused push overwrites the member; any other access materialises the zeros first.
  int zero_pending = -1;
  // the THIRD component of member lazy_xt is logically zero and its plane holds old values: the 2-D record-fed push
  // writes two components of x_tgt, the third one is zero since the fused updatePtclPositions and is never written --
  // the next committing re-layout packs a zero for it, anything else writes the plane first (ps_zeros).  Round 5: the
  // push used to write those zeros and the pack to read them back, 160 MB per step of the 2-D literal.
  bool zero_z_pending = false;
  // pp_ps_set_origin_trust: the caller vouches that every live particle's position lies in the element
  // the fused push starts from (true when the structure was rebuilt from, or the ids are, the unmodified
  // result of the previous search): check_initial_parents is skipped
  bool trust_origins = false;
  // Deferred second pass of the full re-layout (DESIGN "Rebuild: the record-fed push").  After a rebuild
  // with the fused updatePtclPositions the particles sit in the 32-B staging records (+ side word) of the move's first
  // pass (s_aos_live, indexed by NEW slot); the pass that copies them into the SoA arrays is not run.
  //   lazy_rec == 1: every travelling member is in the records, the SoA arrays are stale;
  //   lazy_rec == 2: the fused push consumed the records (pp_search.hip: RECIN) and wrote every member
  //                  but lazy_x to the SoA arrays -- only member lazy_x (the origin) is still in records.
  //   lazy_rec == 3: any particle type whose record is wider than 64 B, after a rebuild without new particles
  //                  and without the commit: every member is in the records (rec_nq quads each, slot order); the
  //                  next rebuild moves records to records (round 5: the back-to-back rebuilds of ps_combo160).
  // Anything else that touches member data goes through ps_ready(), which runs the deferred pass.
  int lazy_rec = 0;
  int rec_nq = 0;  // quads per record of the live records (lazy_rec == 3)
  int lazy_x = -1, lazy_xt = -1;  // commit members of the rebuild that left the records
  pp::DevBuf s_aos_live;
  // the 4-byte member that travels BESIDE the 32-B records of the pseudoXGCm type (WordTable::side_*), record order
  pp::DevBuf s_side, s_side_live;
  // Position of a slot's record in s_aos_live.  rec_rm: ROW-MAJOR inside a chunk -- the record of (row r, column p)
  // of chunk c is number pp_rec_row0(chunk_start[c], c, r, chunk_width[c], C) + p, so the particles of a row
  // (consecutive ranks of one element) are consecutive records and the re-layout's scattered stores leave as
  // runs instead of single records (round 4); else the record index is the slot.  d_erec0 / s_erec0: first record
  // of every element's row in the current / the new layout.
  bool rec_rm = false;
  // the live records are SPLIT (2-D loop, lazy_rec 1 / 2): s_aos_live holds (x, y) per record, s_side_live the second
  // halves (pad, phi, b, id) as 16-B quads -- the 2-D push reads those only
  bool rec_split = false;
  pp::DevBuf d_erec0, s_erec0;
  // One over-full row at the end of the CURRENT layout (pseudoXGCm's remainder rule puts 60 000 particles into
  // one element, pseudoXGCm.cpp:167-222): with a full sort the rows are in ascending order of their counts, so it
  // is the last row of the last chunk, and beyond the second-largest count the chunk's other 63 rows are padding
  // -- 3.8 M slots that the histogram and the first pass of the next re-layout would visit to find 60 000
  // particles.  Set by a full re-layout (chunk height 64, one sort window, row-major records), cleared by anything
  // that changes rows in place.
  pp::HotRow hot;
  pp::DevBuf s_rs, s_holes;  // in-place rebuild: per-element counters, per-row hole lists
  // pinned landing zone of the rebuild's totals (host-mapped) + its event, and the stamp the host polls for
  void* h_totals = nullptr;
  void* ev_totals = nullptr;
  int totals_stamp = 0;
  // not_found counter of the most recent pp_push_search as the last full re-layout's totals carried it to the host
  // (-1: not carried): pp_ps_last_search_found
  int search_nf = -1;
  // the structure's own pair of search counter sets (pp_search.hip: Counters[2], used alternately), the not-found
  // counter of its most recent pp_push_search and that search's serial number in the process
  void* cnt2 = nullptr;
  int cnt2_cur = 0;
  const int* last_nf_dev = nullptr;
  unsigned long long searched_serial = 0;
  ~pp_ps() {
    if (cnt2) (void)hipFree(cnt2);
    if (h_totals) (void)hipHostFree(h_totals);
    if (ev_totals) (void)hipEventDestroy((hipEvent_t)ev_totals);
  }
  // scratch reused across rebuilds
  void* ppe_zeroed = nullptr;  // == s_ppe.p: that buffer was cleared by the previous re-layout's tail
  size_t ppe_zeroed_bytes = 0;
  const pp::GyroRide* ride = nullptr;  // set for the duration of a pp_ps_rebuild_scatter call (consumed by enqueue_layout)
  bool ride_done = false;
  int wide_skip = 0;  // full re-layouts left before the one-pass layout sort is tried again (it overflowed)
  int narrow_skip = 0;  // ... before a digit narrower than 2048 is tried again (pp_ps.hip: wide_ndig)
  pp::DevBuf s_ppe, s_keys, s_keys2, s_vals, s_vals2, s_hist, s_chunkw, s_misc, s_rowstart,
      s_newidx, s_offsets2, s_s2c2, s_r2e2, s_e2r2, s_mask2, s_slot2, s_scan, s_cstart2, s_cwidth2, s_aos, s_idx, s_ranknew, s_eslot0, s_scan2;
};

// ---------------------------------------------------------------------------------------------
// communicator (pp_comm.hip).  kind: 0 self, 1 rccl, 2 tcp, 3 host, 4 local
namespace pp {
struct LocalWorld;
struct TcpStar;
struct MigratePending {  // arguments of pp_ps_migrate_begin kept until pp_ps_migrate_end
  bool active = false;
  pp_ps* ps = nullptr;
  int m_x = -1, m_xtgt = -1;
  int* new_element = nullptr;
  int n_new = 0;
  const int* new_elems = nullptr;
  const void* const* new_info = nullptr;
  const int* gid2lid = nullptr;
  int64_t ngids = 0;
  const pp_mesh* mesh = nullptr;
  int nmaps = 0;
  std::vector<const int*> v2v;
  std::vector<double*> outs;
  double rmax = 0;
  int gnr = 2, gppr = 1;
  int rec_bytes = 0;
  std::vector<int> send_counts, recv_counts;  // particles per rank
  bool recv_known = false;
  int64_t n_send = 0;
};
}  // namespace pp
struct pp_comm {
  int kind = 0, rank = 0, nranks = 1;
  void* nccl = nullptr;  // ncclComm_t
  pp_comm_host_ops ops{};
  void* user = nullptr;
  pp::TcpStar* tcp = nullptr;            // kind tcp: owns the sockets (ops/user point into it)
  std::shared_ptr<pp::LocalWorld> world;  // kind local
  pp::DevBuf d_counts, d_allcounts, d_send, d_recv, d_small;
  void* h_pin = nullptr;  // pinned staging (counts matrix; host-staged transports: records)
  size_t h_pin_bytes = 0;
  pp::MigratePending pend;
  int pin_reserve(size_t bytes);
};
namespace pp {
// transport primitives used by the migration (pp_migrate.hip)
// counts: d_counts holds `nranks` ints on the device (particles leaving for every rank).  Fills
// send_counts and, when the transport can know them already, recv_counts (local: only at `end`).
int comm_counts(pp_comm* c, const int* d_counts, std::vector<int>& send_counts,
                std::vector<int>& recv_counts, bool* recv_known);
// records: d_send rank-major (send_counts particles of rec_bytes each); returns the received records
// (library-owned device buffer of the communicator) in rank order
int comm_exchange_records(pp_comm* c, const void* d_send, const std::vector<int>& send_counts,
                          std::vector<int>& recv_counts, int rec_bytes, void** d_recv_out, int chan = 0);
// local communicator bookkeeping (virtual ranks of one process)
// (chan: 0 migration, 1 / 2 fan-in / fan-out of pp_picpart_reduce -- independent rounds of the local world)
int local_publish(LocalWorld* w, int rank, const std::vector<int>& send_counts, const void* d_send, int rec_bytes,
                  int chan = 0);
int local_all_begun(LocalWorld* w, int chan = 0);
void local_ended(LocalWorld* w, int rank, int chan = 0);
// pp_runtime.hip: roctx ranges (no-ops unless PP_ROCTX=1)
void range_push(const char* name);
void range_pop();
struct Range {
  explicit Range(const char* n) { range_push(n); }
  ~Range() { range_pop(); }
};
}  // namespace pp

namespace pp {
// pp_ps.hip: write the zeros of a member that is only logically zero (pp_ps::zero_pending); every
// entry point that reads or exposes member data calls this first
int ps_materialize(pp_ps* ps);
int ps_zeros(pp_ps* ps);  // only the pending zeros of pp_ps::zero_pending
const int* slot_elem(const pp_ps* ps);  // d_slot_elem, filled first when the last re-layout left it out (pp_ps.hip)
const int* group_chunk(const pp_ps* ps);  // d_group_chunk (SCS, chunk height 64), else nullptr
// pp_ps.hip: exclusive scan of n ints on the library stream (three launches beyond 16 K entries); *total_dev (may be
// null) receives the sum
int scan_excl_i32(DevBuf& scratch, int n, const int* in, int* out, int* total_dev);
const int* search_not_found_dev(const pp_ps* ps);  // pp_search.hip: device address of the not_found counter of the structure's last pp_push_search
unsigned long long search_serial();
void search_counters_released(const void* cnt2);  // pp_push_search calls of the process so far
inline int ps_ready(const pp_ps* ps) {
  return (ps && (ps->zero_pending >= 0 || ps->lazy_rec || ps->zero_z_pending)) ? ps_materialize(const_cast<pp_ps*>(ps)) : PP_OK;
}
// pp_ps_rebuild_scatter with one more promise: member m_xtgt of every NEW particle is zero (arrivals of a
// migration that packed with the commit) -- the rebuild may then defer its second pass as if there were none
int ps_rebuild_scatter(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_new,
                       const int* new_elems_dev, const void* const* new_info_dev, const pp_mesh* mesh, int nmaps,
                       const int* const* v2v_dev, double* const* scatter_w_dev, double rmax, int gnr, int gppr,
                       bool new_xt_zero);
// the structure can feed the record-fed fused push with these member roles (3-D, pseudoXGCm particle type)
bool lazy_push_ok(const pp_ps* ps, int m_x, int m_xtgt, int m_b, int m_phi);
// pp_scatter.hip: gyroScatter for `nmaps` ring maps from an explicit per-element count array (the
// histogram of a rebuild that is still in flight); forgets the ring accumulation kept for reuse.
int gyro_scatter_counts(const pp_mesh* mesh, const int* cnt_dev, int nmaps, const int* const* v2v_dev,
                        double* const* out_dev, double rmax, int gnr, int gppr);
// The same scatter RIDING in the launches of the rebuild that makes the counts (pp_ps_rebuild_scatter): its two
// kernels are a few hundred blocks of latency-bound gathers, and so are the rebuild's key sweep (49 blocks at
// 10^5 elements) and layout kernel (one block) -- side by side they cost what the longer one costs.  The first
// stage runs as extra blocks of k_make_keys (both need only the finished histogram), the second as extra
// blocks of k_layout_fused (a kernel boundary later).  gyro_scatter_ride() fills the arguments when the calls
// reduce to the ring accumulation + ONE transposed-map gather (one map, or the two maps of one
// pp_create_gyro_ring_mappings call), else leaves on = 0 and the caller uses gyro_scatter_counts().
struct GyroRide {
  int on = 0;
  int nverts = 0, gnr = 0, ringDown = 0, ringUp = 0, gppr = 1;
  const int *v2e_off = nullptr, *v2e = nullptr;   // vertex -> elements
  double* ring = nullptr;                         // [nverts][gnr]
  const int *off = nullptr, *src = nullptr;       // transposed ring map
  double *out = nullptr, *out2 = nullptr;
};
int gyro_scatter_ride(const pp_mesh* mesh, int nmaps, const int* const* v2v_dev, double* const* out_dev,
                      double rmax, int gnr, int gppr, GyroRide* ride);
// first stage, thread = vertex: ring_accum[v][ringUp] = ring_accum[v][ringDown] = live particles in the elements
// around v (gyroScatter.hpp:188-197 as a gather; integer sums in double are exact in any order)
__device__ __forceinline__ void gyro_rings_body(int v, int nverts, int gnr, const int* __restrict__ v2e_off,
                                                const int* __restrict__ v2e, const int* __restrict__ cnt,
                                                int ringDown, int ringUp, double* __restrict__ ring_accum) {
  if (v >= nverts) return;
  long long n = 0;
  for (int j = v2e_off[v]; j < v2e_off[v + 1]; ++j) n += cnt[v2e[j]];
  for (int r = 0; r < gnr; ++r)
    ring_accum[(size_t)v * gnr + r] = (r == ringUp ? (double)n : 0.0) + (r == ringDown ? (double)n : 0.0);
}
// second stage, 16 lanes per target vertex (g = 16 * vertex + lane; groups of 16 do not straddle waves): the
// lanes fetch 16 list entries at once, then every lane adds them up in list order = the sequential sum
// (exact = the caller does not need the list order of the sum: every term is an integer count divided by a power of
// two -- the constant-radius scatter with gppr = 2^k: the sum is exact, bit-identical to the sequential one -- or the
// terms themselves come from unordered atomics -- the per-particle-radius scatter.  Each of the 16 lanes adds up its
// own entries and a four-step butterfly joins them, instead of sixteen shuffles per batch of sixteen entries.)
__device__ __forceinline__ void gyro_gather_body(int g, int nverts, int gppr, const int* __restrict__ off,
                                                 const int* __restrict__ src, const double* __restrict__ ring_accum,
                                                 double* __restrict__ scatter_w, double* __restrict__ scatter_w2,
                                                 bool exact = false) {
  const int t = g >> 4, sub = g & 15, lane0 = (threadIdx.x & 63) & ~15;
  const bool in = t < nverts;
  const int b = in ? off[t] : 0, e = in ? off[t + 1] : 0;
  int len = e - b;
  for (int o = 16; o < 64; o <<= 1) len = max(len, __shfl_xor(len, o));  // wave-uniform trip count
  double w = 0;
  if (exact) {  // (wave-uniform)
    for (int base = 0; base < len; base += 16) {
      const int j = b + base + sub;
      w += j < e ? ring_accum[src[j]] / gppr : 0.0;
    }
    for (int o = 8; o > 0; o >>= 1) w += __shfl_xor(w, o);
    if (in && sub == 0) {
      scatter_w[t] = w;
      if (scatter_w2) scatter_w2[t] = w;
    }
    return;
  }
  for (int base = 0; base < len; base += 16) {
    const int j = b + base + sub;
    const double val = j < e ? ring_accum[src[j]] / gppr : 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double x = __shfl(val, lane0 + k);
      if (b + base + k < e) w += x;
    }
  }
  if (in && sub == 0) {
    scatter_w[t] = w;
    if (scatter_w2) scatter_w2[t] = w;  // a second map with the same transpose: the same sums in the same order
  }
}
}  // namespace pp
