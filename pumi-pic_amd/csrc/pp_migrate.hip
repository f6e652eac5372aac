// pp_migrate.hip -- device side of particle migration.
//   setUnsafeProcs                 src/pumipic_ptcl_ops.hpp:32-52
//   SellCSigma::migrate send side  particle_structs/src/scs/SCS_migrate.h:29-137,189-196
//
// The reference packs one SoA send buffer per member type and posts T+1 host-staged MPI messages
// per peer.  Here the particles leaving for every peer are packed rank-major into one buffer per
// member (+ the element gid as int64, fixing the int truncation of SCS_migrate.h:77);
// the exchange itself is a single all-to-all-v over RCCL/xGMI issued by the host layer, and the
// received particles enter pp_ps_rebuild as "new particles" exactly like the reference
// (SCS_migrate.h:198-213).
#include <algorithm>
#include <vector>
#include "pp_internal.hpp"

namespace {
using pp::grid_for;
using pp::kBlock;

// (every slot of the capacity, like the reference's parallel_for: the slots of padding rows are masked off and
// keep this rank; no slot -> element table is read)
__global__ void k_unsafe(int capacity, const unsigned char* __restrict__ mask, const int* __restrict__ elems,
                         const unsigned char* __restrict__ safe, const int* __restrict__ owners,
                         int rank, int* __restrict__ new_elems, int* __restrict__ new_procs) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  int proc = rank;
  const int nelm = __builtin_nontemporal_load(elems + pid);
  __builtin_nontemporal_store(nelm, new_elems + pid);
  if (mask[pid] && nelm != -1 && !safe[nelm]) proc = owners[nelm];
  __builtin_nontemporal_store(proc, new_procs + pid);
}

// a particle is sent when it is live, keeps a valid new element and is routed to another rank
__global__ void k_send_count(int capacity, const unsigned char* __restrict__ mask,
                             const int* __restrict__ new_element,
                             const int* __restrict__ new_process, int rank, int nranks,
                             int* __restrict__ counts) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid] || new_element[pid] < 0) return;
  const int p = new_process[pid];
  if (p != rank && p >= 0 && p < nranks) atomicAdd(&counts[p], 1);
}

struct PackArgs {
  int nmembers;
  const void* src[8];
  void* dst[8];
  int bytes[8];
  int ncomp[8];
  long long src_stride, dst_stride;
};

__global__ void k_pack(int capacity, const unsigned char* __restrict__ mask, int* new_element,
                       const int* __restrict__ new_process, int rank, int nranks,
                       int* __restrict__ cursor, const long long* __restrict__ gids,
                       long long* __restrict__ send_gid, PackArgs a) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  const int e = new_element[pid];
  if (e < 0) return;
  const int p = new_process[pid];
  if (p == rank || p < 0 || p >= nranks) return;
  const int idx = atomicAdd(&cursor[p], 1);
  send_gid[idx] = gids ? gids[e] : (long long)e;
  for (int m = 0; m < a.nmembers; ++m) {
    const int nc = a.ncomp[m];
    if (a.bytes[m] == 8) {
      const unsigned long long* s = (const unsigned long long*)a.src[m];
      unsigned long long* d = (unsigned long long*)a.dst[m];
      for (int c = 0; c < nc; ++c) d[c * a.dst_stride + idx] = s[c * a.src_stride + pid];
    } else if (a.bytes[m] == 4) {
      const unsigned* s = (const unsigned*)a.src[m];
      unsigned* d = (unsigned*)a.dst[m];
      for (int c = 0; c < nc; ++c) d[c * a.dst_stride + idx] = s[c * a.src_stride + pid];
    } else if (a.bytes[m] == 2) {
      const unsigned short* s = (const unsigned short*)a.src[m];
      unsigned short* d = (unsigned short*)a.dst[m];
      for (int c = 0; c < nc; ++c) d[c * a.dst_stride + idx] = s[c * a.src_stride + pid];
    } else {
      const unsigned char* s = (const unsigned char*)a.src[m];
      unsigned char* d = (unsigned char*)a.dst[m];
      for (int c = 0; c < nc; ++c) d[c * a.dst_stride + idx] = s[c * a.src_stride + pid];
    }
  }
  new_element[pid] = -1;  // removeSentParticles (SCS_migrate.h:189-196)
}
// ---- one-record-per-particle form: [gid (2 words) | member words ...] padded to 16 B.  One
// buffer per step means ONE all-to-all-v over xGMI instead of one per member (the reference
// posts T+1 messages per peer, SCS_migrate.h:143-178).
constexpr int kRecMaxWords = 66;
struct RecTable {
  int nwords;       // member words (without the gid)
  int rec_words;    // padded record length in words
  const char* src[kRecMaxWords];
  int scale[kRecMaxWords];
  char* dst[kRecMaxWords];  // unpack: destination arrays [ncomp][n]
  // bit w set: word w carries a member scalar of 1 or 2 bytes (scale[w]) in its low bytes -- a `short` or `char` member
  // (particle_structs/test/test_types.hpp:12: MemberTypes<int, Vector3, short, int>) travels as one word per component
  unsigned long long narrow;
};
__device__ __forceinline__ unsigned rec_load(const RecTable& t, int w, long long i) {
  const char* p = t.src[w] + i * t.scale[w];
  if (w < 64 && (t.narrow >> w & 1ull)) return t.scale[w] == 2 ? (unsigned)*(const unsigned short*)p : (unsigned)*(const unsigned char*)p;
  return *(const unsigned*)p;
}
__device__ __forceinline__ void rec_store(const RecTable& t, int w, long long i, unsigned v) {
  char* p = t.dst[w] + i * t.scale[w];
  if (w < 64 && (t.narrow >> w & 1ull)) {
    if (t.scale[w] == 2)
      *(unsigned short*)p = (unsigned short)v;
    else
      *(unsigned char*)p = (unsigned char)v;
  } else {
    *(unsigned*)p = v;
  }
}
__global__ void k_pack_records(int capacity, const unsigned char* __restrict__ mask,
                               int* new_element, const int* __restrict__ new_process, int rank,
                               int nranks, int* __restrict__ cursor,
                               const long long* __restrict__ gids, unsigned* __restrict__ out,
                               RecTable t) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  const int e = new_element[pid];
  if (e < 0) return;
  const int p = new_process[pid];
  if (p == rank || p < 0 || p >= nranks) return;
  const int idx = atomicAdd(&cursor[p], 1);
  unsigned* r = out + (size_t)idx * t.rec_words;
  const long long g = gids ? gids[e] : (long long)e;
  r[0] = (unsigned)(g & 0xffffffffll);
  r[1] = (unsigned)((unsigned long long)g >> 32);
  for (int w = 0; w < t.nwords; ++w)
    r[2 + w] = t.src[w] ? rec_load(t, w, pid) : 0u;
  new_element[pid] = -1;  // removeSentParticles (SCS_migrate.h:189-196)
}
__global__ void k_unpack_records(int n, const unsigned* __restrict__ rec, const int* __restrict__ gid2lid,
                                 long long ngids, int* __restrict__ elems, int* bad, RecTable t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned* r = rec + (size_t)i * t.rec_words;
  const long long g = (long long)(((unsigned long long)r[1] << 32) | r[0]);
  int lid;
  if (gid2lid) {
    lid = (g >= 0 && g < ngids) ? gid2lid[g] : -1;
  } else {
    lid = (int)g;  // full-mesh replica: global id == local id
  }
  if (lid < 0) *bad = 1;
  elems[i] = lid;
  for (int w = 0; w < t.nwords; ++w) rec_store(t, w, i, r[2 + w]);
}

// ---- routing.  A per-particle atomicAdd on one of `nranks` counters serialises in the L2 (~10 ns each: 3 ms for
// 300 k leaving particles), so positions in the rank-major send buffer come from counts and a scan:
//   k_route_mark    block = 4096 consecutive slots (a thread: 4 x 4 of them): where every slot goes as ONE BYTE (0xff = stays),
//                   the block's leavers per destination counted in LDS -> block_cnt[dest][block]
//   scan            ONE exclusive scan over the rank-major [dest][block] table: entry (p, b) is the first record of
//                   block b's leavers for rank p in the send buffer (relative to rank 0's first)
//   k_route_totals  leavers per destination (what the count exchange needs on the host)
//   k_route_pack    the same block -> slot mapping, reads the byte per slot (not mask / element / safe / owner
//                   again); a block without leavers -- nearly all of them: two particles in a thousand leave a
//                   rank of configs[4] per step -- returns after 4 KB of loads; leavers take their place by LDS atomics.
// Round 5: this replaced 1024 persistent blocks sweeping the slots twice with a one-THREAD-per-destination scan of
// their counts in between -- 180 + 145 + 435 us per rank and step at configs[4] (a quarter of the step) for 62 000
// leavers among 40 M slots.
constexpr int kRouteSlots = 4096;  // slots per block: 4 groups of 4 consecutive slots per thread
constexpr int kMaxRanks = 255;  // (0xff marks a slot that stays)
struct RankStarts {
  int v[64];  // first record of every destination rank in the send buffer (nranks <= 64 by value)
};
// where a particle goes: the caller's new_process array, or -- setUnsafeProcs folded in
// (src/pumipic_ptcl_ops.hpp:32-52) -- the owner of its new element when that element is not safe here
struct RouteRule {
  const int* new_process;
  const unsigned char* safe;
  const int* owners;
};
__device__ __forceinline__ int route_dest(int pid, int capacity, const unsigned char* __restrict__ mask,
                                          const int* __restrict__ new_element, const RouteRule rr, int rank,
                                          int nranks) {
  if (pid >= capacity || !mask[pid]) return -1;
  const int e = new_element[pid];
  if (e < 0) return -1;
  const int p = rr.new_process ? rr.new_process[pid] : (rr.safe[e] ? rank : rr.owners[e]);
  return (p != rank && p >= 0 && p < nranks) ? p : -1;
}
// (thread = 4 consecutive slots: mask, new element and destination byte move as one 4-/16-/4-byte access each)
__global__ void k_route_mark(int capacity, const unsigned char* __restrict__ mask, const int* __restrict__ new_element,
                             const RouteRule rr, int rank, int nranks, int nblocks,
                             unsigned char* __restrict__ dest8, int* __restrict__ block_cnt) {
  __shared__ int h[kMaxRanks + 1];
  for (int i = threadIdx.x; i < nranks; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const bool aligned = ((size_t)new_element & 15) == 0 && ((size_t)mask & 3) == 0 && ((size_t)dest8 & 3) == 0 &&
                       !rr.new_process;
#pragma unroll
  for (int grp = 0; grp < kRouteSlots / 1024; ++grp) {
  const long long p0 = (long long)blockIdx.x * kRouteSlots + grp * 1024 + 4 * threadIdx.x;
  const bool vec = aligned && p0 + 3 < capacity;
  if (vec) {
    const uchar4 m4 = *(const uchar4*)(mask + p0);
    const int4 e4 = *(const int4*)(new_element + p0);
    const unsigned char m[4] = {m4.x, m4.y, m4.z, m4.w};
    const int e[4] = {e4.x, e4.y, e4.z, e4.w};
    unsigned char d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int p = -1;
      if (m[k] && e[k] >= 0) {
        const int q = rr.safe[e[k]] ? rank : rr.owners[e[k]];
        p = (q != rank && q >= 0 && q < nranks) ? q : -1;
      }
      d[k] = p < 0 ? (unsigned char)0xff : (unsigned char)p;
      if (p >= 0) atomicAdd(&h[p], 1);
    }
    *(uchar4*)(dest8 + p0) = make_uchar4(d[0], d[1], d[2], d[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long pid = p0 + k;
      if (pid < capacity) {
        const int p = route_dest((int)pid, capacity, mask, new_element, rr, rank, nranks);
        dest8[pid] = p < 0 ? (unsigned char)0xff : (unsigned char)p;
        if (p >= 0) atomicAdd(&h[p], 1);
      }
    }
  }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nranks; i += blockDim.x) block_cnt[(size_t)i * nblocks + blockIdx.x] = h[i];
}
__global__ void k_route_totals(int nranks, int nblocks, const int* __restrict__ S, const int* __restrict__ total,
                               int* __restrict__ counts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nranks) return;
  counts[p] = (p + 1 < nranks ? S[(size_t)(p + 1) * nblocks] : *total) - S[(size_t)p * nblocks];
}
__global__ void k_route_pack(int capacity, const unsigned char* __restrict__ dest8, int* new_element, int nranks,
                             int nblocks, const int* __restrict__ S, RankStarts rs,
                             const int* __restrict__ rank_start_dev, const long long* __restrict__ gids,
                             unsigned* __restrict__ out, RecTable t) {
  __shared__ int cur[kMaxRanks + 1];
  constexpr int NG = kRouteSlots / 1024;
  const long long lo = (long long)blockIdx.x * kRouteSlots + 4 * threadIdx.x;  // the mapping of k_route_mark
  unsigned char d[4 * NG];
  unsigned all = 0xffu;
#pragma unroll
  for (int grp = 0; grp < NG; ++grp) {
    const long long p0 = lo + grp * 1024;
    unsigned char* dg = d + 4 * grp;
    dg[0] = dg[1] = dg[2] = dg[3] = 0xff;
    if (p0 + 3 < capacity && ((size_t)dest8 & 3) == 0) {
      const uchar4 d4 = *(const uchar4*)(dest8 + p0);
      dg[0] = d4.x;
      dg[1] = d4.y;
      dg[2] = d4.z;
      dg[3] = d4.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (p0 + k < capacity) dg[k] = dest8[p0 + k];
    }
    all &= dg[0] & dg[1] & dg[2] & dg[3];
  }
  const bool mine = all != 0xffu;
  if (!__syncthreads_or(mine)) return;
  for (int i = threadIdx.x; i < nranks; i += blockDim.x)
    cur[i] = (rank_start_dev ? rank_start_dev[i] : rs.v[i]) + S[(size_t)i * nblocks + blockIdx.x] - S[(size_t)i * nblocks];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4 * NG; ++k) {
    if (d[k] == 0xff) continue;
    const int pid = (int)(lo + (k >> 2) * 1024 + (k & 3)), p = d[k];
    const int e = new_element[pid];
    const int idx = atomicAdd(&cur[p], 1);  // LDS
    unsigned* r = out + (size_t)idx * t.rec_words;
    const long long g = gids ? gids[e] : (long long)e;
    r[0] = (unsigned)(g & 0xffffffffll);
    r[1] = (unsigned)((unsigned long long)g >> 32);
    // (eight member words in flight at a time: one load -> one store at a time is a chain of ~20 memory round trips
    // per leaver, the compiler cannot reorder them across the stores to `out`)
    for (int w0 = 0; w0 < t.nwords; w0 += 8) {
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        v[k] = (w0 + k < t.nwords && t.src[w0 + k]) ? rec_load(t, w0 + k, pid) : 0u;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w0 + k < t.nwords) r[2 + w0 + k] = v[k];
    }
    new_element[pid] = -1;  // removeSentParticles (SCS_migrate.h:189-196)
  }
}

// commit_x / commit_xt >= 0: the record carries the particle AFTER updatePtclPositions (member
// commit_x is read from commit_xt's arrays, member commit_xt travels as zeros)
// shape_only: only the layout of the record is wanted (sizes, the unpack side): no member is read, so
// nothing has to be written back to the SoA arrays first.  The same holds for a committing pack when only
// the ORIGIN of the same commit pair is still in the re-layout's records (pp_ps::lazy_rec == 2): the
// committed record reads member commit_xt, never commit_x.
int build_rec_table(const pp_ps* ps, RecTable& t, int commit_x = -1, int commit_xt = -1, bool shape_only = false) {
  const bool origin_only_missing = ps->lazy_rec == 2 && ps->zero_pending < 0 && commit_x >= 0 &&
                                   commit_x == ps->lazy_x && commit_xt == ps->lazy_xt;
  if (!shape_only && !origin_only_missing)
    if (int rc = pp::ps_ready(ps)) return rc;
  // (the committed record reads all three components of member commit_xt: a third one that is only logically zero --
  // after a 2-D record-fed push -- is written now)
  if (!shape_only && ps->zero_z_pending)
    if (int rc = pp::ps_zeros(const_cast<pp_ps*>(ps))) return rc;
  int nw = 0;
  for (int m = 0; m < ps->nmembers; ++m) {
    const int s = ps->member_map[m == commit_x ? commit_xt : m];
    const int b = ps->member_bytes[s];
    PP_REQUIRE(b == 1 || b == 2 || b == 4 || b == 8, "migration records need member scalars of 1, 2, 4 or 8 bytes");
    for (int c = 0; c < ps->member_ncomp[s]; ++c)
      for (int hw = 0; hw < (b < 4 ? 1 : b / 4); ++hw) {
        PP_REQUIRE(nw < kRecMaxWords, "particle record too large for the migration pack");
        if (b < 4) {
          PP_REQUIRE(nw < 64, "a 1- or 2-byte member must lie within the first 64 words of the migration record");
          t.narrow |= 1ull << nw;
        }
        t.src[nw] = m == commit_xt ? nullptr : (const char*)ps->data[s].p + ((size_t)c * ps->stride) * b + hw * 4;
        t.scale[nw] = b;
        t.dst[nw] = nullptr;
        ++nw;
      }
  }
  t.nwords = nw;
  t.rec_words = ((2 + nw + 3) / 4) * 4;
  return PP_OK;
}
// ---- PICpart safe zone / buffer by breadth-first layers (pumipic_part_construct.cpp:387-468)
__global__ void k_bfs_init(int ne, const int* __restrict__ owner, int rank, int* __restrict__ visited,
                           int* __restrict__ next, unsigned char* __restrict__ safe) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int own = owner[e] == rank;
  visited[e] = next[e] = own;
  safe[e] = (unsigned char)own;
}
// BFS(): every bridge entity (vertex or side) with a visited adjacent element visits all of them
__global__ void k_bfs_sweep(int nbridges, const int* __restrict__ off, const int* __restrict__ vals,
                            const int* __restrict__ visited, int* __restrict__ next) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbridges) return;
  const int first = off[b], last = off[b + 1];
  bool here = false;
  for (int j = first; j < last; ++j) here |= visited[vals[j]] != 0;
  if (here)
    for (int j = first; j < last; ++j) next[vals[j]] = 1;
}
__global__ void k_bfs_copy(int ne, int i, int safe_layers, int ghost_layers,
                           const int* __restrict__ owner, int* __restrict__ visited,
                           const int* __restrict__ next, unsigned char* __restrict__ safe,
                           int* __restrict__ has_part) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int v = next[e];
  visited[e] = v;
  if (i == safe_layers - 1) safe[e] = (unsigned char)v;
  if (i < ghost_layers && v) has_part[owner[e]] = 1;
}
__global__ void k_bfs_inward_init(int ne, const int* __restrict__ owner,
                                  const int* __restrict__ has_part, int* __restrict__ visited,
                                  int* __restrict__ next) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  visited[e] = next[e] = !has_part[owner[e]];
}
__global__ void k_bfs_inward_set(int ne, const int* __restrict__ owner, int rank,
                                 const int* __restrict__ visited, unsigned char* __restrict__ safe) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  safe[e] = (unsigned char)(!visited[e] || owner[e] == rank);
}
// grow-only scratch of the per-step migration calls (library lifetime: a hipMalloc / hipFree pair
// per call costs more than the kernels, and hipFree drains the GPU)
pp::DevBuf& scratch(int i) {
  static std::vector<pp::DevBuf>* v = new std::vector<pp::DevBuf>(24);
  return (*v)[(size_t)i];
}
int bridge_adjacency(const pp_mesh* mesh, int bridge_dim, int* n, const int** off, const int** vals) {
  if (bridge_dim == 0) {
    *n = mesh->nverts;
    *off = mesh->d_vert2elems_off.as<int>();
    *vals = mesh->d_vert2elems.as<int>();
  } else if (bridge_dim == mesh->dim - 1) {
    *n = mesh->nsides;
    *off = mesh->d_side2elems_off.as<int>();
    *vals = mesh->d_side2elems.as<int>();
  } else {
    pp::set_error("PICpart BFS: bridge_dim must be 0 (vertices) or dim-1 (sides)");
    return PP_EINVAL;
  }
  return PP_OK;
}
}  // namespace

extern "C" {

int pp_ps_migrate_record_bytes(const pp_ps* ps) {
  if (!ps) return PP_EINVAL;
  RecTable t{};
  const int rc = build_rec_table(ps, t, -1, -1, true);
  return rc ? rc : t.rec_words * 4;
}

static int pack_records(const pp_ps* ps, int commit_x, int commit_xt, int* new_element_dev,
                        const int* new_process_dev, int comm_rank, int nranks,
                        const int* send_counts_host, void* send_records_dev);
int pp_ps_migrate_pack_records(const pp_ps* ps, int* new_element_dev, const int* new_process_dev,
                               int comm_rank, int nranks, const int* send_counts_host,
                               void* send_records_dev) {
  return pack_records(ps, -1, -1, new_element_dev, new_process_dev, comm_rank, nranks,
                      send_counts_host, send_records_dev);
}
int pp_ps_migrate_pack_records_commit(const pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev,
                                      const int* new_process_dev, int comm_rank, int nranks,
                                      const int* send_counts_host, void* send_records_dev) {
  PP_REQUIRE(ps && m_x >= 0 && m_xtgt >= 0 && m_x < ps->nmembers && m_xtgt < ps->nmembers && m_x != m_xtgt,
             "pp_ps_migrate_pack_records_commit: bad member index");
  PP_REQUIRE(ps->member_bytes[ps->member_map[m_x]] == 8 && ps->member_bytes[ps->member_map[m_xtgt]] == 8 &&
                 ps->member_ncomp[ps->member_map[m_x]] == ps->member_ncomp[ps->member_map[m_xtgt]],
             "pp_ps_migrate_pack_records_commit: x and x_tgt must be double members of equal shape");
  return pack_records(ps, m_x, m_xtgt, new_element_dev, new_process_dev, comm_rank, nranks,
                      send_counts_host, send_records_dev);
}
// the routing of the current arrays: destination byte per slot (scratch(6)), per-block leaver counts
// (scratch(4): [nranks][nblocks]) and their scan (scratch(5)); totals per destination to counts_dev (nranks ints)
static int route_count(const pp_ps* ps, const int* new_element_dev, const RouteRule new_process_dev,
                       int comm_rank, int nranks, int* counts_dev, int* nblocks_out) {
  PP_REQUIRE(nranks <= kMaxRanks, "migration: more than 255 ranks are not supported by the routing kernels");
  hipStream_t st = pp::stream();
  const int cap = ps->num_ptcls > 0 ? ps->capacity : 0;
  const int nblocks = std::max(1, (cap + kRouteSlots - 1) / kRouteSlots);
  *nblocks_out = nblocks;
  const size_t n = (size_t)nblocks * nranks;
  PP_REQUIRE(n < ((size_t)1 << 31), "migration: routing table too large");
  pp::DevBuf &bc = scratch(4), &S = scratch(5), &d8 = scratch(6);
  PP_HIP_CHECK(bc.reserve(sizeof(int) * n));
  PP_HIP_CHECK(S.reserve(sizeof(int) * (n + 1)));
  PP_HIP_CHECK(d8.reserve((size_t)std::max(cap, 1)));
  k_route_mark<<<nblocks, kBlock, 0, st>>>(cap, ps->d_mask.as<unsigned char>(), new_element_dev, new_process_dev,
                                           comm_rank, nranks, nblocks, d8.as<unsigned char>(), bc.as<int>());
  int rc = pp::scan_excl_i32(scratch(7), (int)n, bc.as<int>(), S.as<int>(), S.as<int>() + n);
  if (rc) return rc;
  k_route_totals<<<grid_for(nranks), kBlock, 0, st>>>(nranks, nblocks, S.as<int>(), S.as<int>() + n, counts_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
// pack pass of the same routing (route_count must have run on the same arrays)
static int route_pack(const pp_ps* ps, int commit_x, int commit_xt, int* new_element_dev,
                      const RouteRule /*new_process_dev*/, int /*comm_rank*/, int nranks, int nblocks,
                      const int* rank_start_host, void* send_records_dev) {
  RecTable t{};
  int rc = build_rec_table(ps, t, commit_x, commit_xt);
  if (rc) return rc;
  hipStream_t st = pp::stream();
  RankStarts rs{};
  const int* rs_dev = nullptr;
  if (nranks <= 64) {
    for (int r = 0; r < nranks; ++r) rs.v[r] = rank_start_host[r];
  } else {  // does not fit the kernel arguments: through device memory (one extra sync)
    pp::DevBuf& cur = scratch(0);
    PP_HIP_CHECK(cur.reserve(sizeof(int) * (size_t)nranks));
    PP_HIP_CHECK(hipMemcpyAsync(cur.p, rank_start_host, sizeof(int) * (size_t)nranks, hipMemcpyHostToDevice, st));
    PP_HIP_CHECK(hipStreamSynchronize(st));
    rs_dev = cur.as<int>();
  }
  k_route_pack<<<nblocks, kBlock, 0, st>>>(ps->num_ptcls > 0 ? ps->capacity : 0, scratch(6).as<unsigned char>(),
                                           new_element_dev, nranks, nblocks, scratch(5).as<int>(), rs, rs_dev,
                                           ps->has_gids ? ps->d_gids.as<long long>() : nullptr,
                                           (unsigned*)send_records_dev, t);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
static int pack_records(const pp_ps* ps, int commit_x, int commit_xt, int* new_element_dev,
                        const int* new_process_dev, int comm_rank, int nranks,
                        const int* send_counts_host, void* send_records_dev) {
  PP_REQUIRE(ps && new_element_dev && new_process_dev && send_counts_host && nranks > 0,
             "pp_ps_migrate_pack_records: bad argument");
  long long total = 0;
  std::vector<int> start((size_t)nranks, 0);
  for (int r = 0; r < nranks; ++r) {
    start[r] = (int)total;
    total += send_counts_host[r];
  }
  if (total == 0) return PP_OK;
  PP_REQUIRE(send_records_dev, "pp_ps_migrate_pack_records: null send buffer");
  pp::DevBuf& cnt = scratch(3);
  PP_HIP_CHECK(cnt.reserve(sizeof(int) * (size_t)nranks));
  int per_block = 0;
  const RouteRule rr{new_process_dev, nullptr, nullptr};
  int rc = route_count(ps, new_element_dev, rr, comm_rank, nranks, cnt.as<int>(), &per_block);
  if (rc) return rc;
  rc = route_pack(ps, commit_x, commit_xt, new_element_dev, rr, comm_rank, nranks, per_block,
                  start.data(), send_records_dev);
  if (rc) return rc;
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  return PP_OK;
}

int pp_ps_rebuild_records(pp_ps* ps, const int* new_element_dev, int n_recv,
                          const void* recv_records_dev, const int* gid2lid_dev, int64_t ngids) {
  return pp_ps_rebuild_records_scatter(ps, -1, -1, new_element_dev, n_recv, recv_records_dev, gid2lid_dev,
                                       ngids, nullptr, 0, nullptr, nullptr, 0.0, 2, 1);
}

// received records (+ the caller's own new particles) become the "new particles" of the rebuild
// (SCS_migrate.h:198-213)
static int rebuild_from_records(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_recv,
                                const void* recv_records_dev, const int* gid2lid_dev, int64_t ngids,
                                int n_extra, const int* extra_elems_dev, const void* const* extra_info_dev,
                                const pp_mesh* mesh, int nmaps, const int* const* v2v_dev,
                                double* const* scatter_w_dev, double rmax, int gnr, int gppr) {
  const bool plain = m_x < 0 && m_xtgt < 0 && nmaps == 0;
  const int n_tot = n_recv + n_extra;
  // the structure's own map (SCS_migrate.h:181-187) -- needed only to place particles that arrive
  // (pp_ps_migrate_begin checks the same thing before anything is packed or removed)
  if (n_recv > 0 && !gid2lid_dev && ps->has_gids && !ps->gids_identity) {
    PP_REQUIRE(ps->n_gid2lid > 0, "migration: the structure's element gids are too sparse for its own gid -> element "
                                  "table -- pass gid2lid_dev");
    gid2lid_dev = ps->d_gid2lid.as<int>();
    ngids = ps->n_gid2lid;
  }
  if (n_recv == 0)
    return plain ? pp_ps_rebuild(ps, new_element_dev, n_extra, extra_elems_dev, extra_info_dev)
                 : pp_ps_rebuild_scatter(ps, m_x, m_xtgt, new_element_dev, n_extra, extra_elems_dev,
                                         extra_info_dev, mesh, nmaps, v2v_dev, scatter_w_dev, rmax, gnr, gppr);
  RecTable t{};
  int rc = build_rec_table(ps, t, -1, -1, true);
  if (rc) return rc;
  hipStream_t st = pp::stream();
  pp::DevBuf* info = &scratch(8);  // scratch(8 + m): member m of the arriving particles
  std::vector<const void*> ptrs((size_t)ps->nmembers);
  int w = 0;
  for (int m = 0; m < ps->nmembers; ++m) {
    const int s = ps->member_map[m];
    const int b = ps->member_bytes[s], nc = ps->member_ncomp[s];
    PP_HIP_CHECK(info[m].reserve((size_t)n_tot * nc * b));
    ptrs[m] = info[m].p;
    for (int c = 0; c < nc; ++c) {
      for (int hw = 0; hw < (b < 4 ? 1 : b / 4); ++hw) t.dst[w++] = (char*)info[m].p + ((size_t)c * n_tot) * b + hw * 4;
      if (n_extra > 0)
        PP_HIP_CHECK(hipMemcpyAsync((char*)info[m].p + ((size_t)c * n_tot + n_recv) * b,
                                    (const char*)extra_info_dev[m] + ((size_t)c * n_extra) * b,
                                    (size_t)n_extra * b, hipMemcpyDeviceToDevice, st));
    }
  }
  pp::DevBuf &elems = scratch(1), &bad = scratch(2);
  PP_HIP_CHECK(elems.reserve(sizeof(int) * (size_t)n_tot));
  PP_HIP_CHECK(bad.reserve(sizeof(int)));
  PP_HIP_CHECK(hipMemsetAsync(bad.p, 0, sizeof(int), st));
  k_unpack_records<<<grid_for(n_recv), kBlock, 0, st>>>(n_recv, (const unsigned*)recv_records_dev,
                                                        gid2lid_dev, (long long)ngids,
                                                        elems.as<int>(), bad.as<int>(), t);
  PP_LAUNCH_CHECK();
  if (n_extra > 0)
    PP_HIP_CHECK(hipMemcpyAsync(elems.as<int>() + n_recv, extra_elems_dev, sizeof(int) * (size_t)n_extra,
                                hipMemcpyDeviceToDevice, st));
  // A gid without a local id unpacks as element -1, which the rebuild rejects like any inactive
  // new particle BEFORE it changes the structure: no separate host sync for the check.  The
  // scratch buffers outlive the call: the rebuild reads them in stream order.
  // (records packed with the commit carry member m_xtgt as zeros: build_rec_table -- with no particles of
  // the caller's own among the new ones the rebuild may treat that member as all zero)
  rc = plain ? pp_ps_rebuild(ps, new_element_dev, n_tot, elems.as<int>(), ptrs.data())
             : pp::ps_rebuild_scatter(ps, m_x, m_xtgt, new_element_dev, n_tot, elems.as<int>(), ptrs.data(), mesh,
                                      nmaps, v2v_dev, scatter_w_dev, rmax, gnr, gppr,
                                      /*new_xt_zero=*/m_x >= 0 && m_xtgt >= 0 && n_extra == 0);
  if (rc == PP_EINVAL) {
    int hbad = 0;
    if (hipMemcpy(&hbad, bad.p, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && hbad)
      pp::set_error("pp_ps_rebuild_records: received an element gid with no local id "
                    "(assert(valid_at(index)), SCS_migrate.h:184)");
  }
  return rc;
}

int pp_ps_rebuild_records_scatter(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_recv,
                                  const void* recv_records_dev, const int* gid2lid_dev, int64_t ngids,
                                  const pp_mesh* mesh, int nmaps, const int* const* v2v_dev,
                                  double* const* scatter_w_dev, double rmax, int gnr, int gppr) {
  PP_REQUIRE(ps && n_recv >= 0 && (n_recv == 0 || recv_records_dev),
             "pp_ps_rebuild_records: bad argument");
  return rebuild_from_records(ps, m_x, m_xtgt, new_element_dev, n_recv, recv_records_dev, gid2lid_dev, ngids,
                              0, nullptr, nullptr, mesh, nmaps, v2v_dev, scatter_w_dev, rmax, gnr, gppr);
}

// ---------------------------------------------------------------------------------------------
// SellCSigma::migrate (scs/SCS_migrate.h:5-222) behind one call
static int migrate_begin_rule(pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev, const RouteRule rr,
                              pp_comm* comm, int n_new, const int* new_elems_dev,
                              const void* const* new_info_dev, const int* gid2lid_dev, int64_t ngids,
                              const pp_mesh* mesh, int nmaps, const int* const* v2v_dev,
                              double* const* scatter_w_dev, double rmax, int gnr, int gppr);
int pp_ps_migrate_begin(pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev,
                        const int* new_process_dev, pp_comm* comm, int n_new,
                        const int* new_elems_dev, const void* const* new_info_dev,
                        const int* gid2lid_dev, int64_t ngids, const pp_mesh* mesh, int nmaps,
                        const int* const* v2v_dev, double* const* scatter_w_dev, double rmax,
                        int gnr, int gppr) {
  PP_REQUIRE(ps && (new_process_dev || ps->capacity == 0), "pp_ps_migrate: null routing arrays");
  return migrate_begin_rule(ps, m_x, m_xtgt, new_element_dev, RouteRule{new_process_dev, nullptr, nullptr}, comm,
                            n_new, new_elems_dev, new_info_dev, gid2lid_dev, ngids, mesh, nmaps, v2v_dev,
                            scatter_w_dev, rmax, gnr, gppr);
}
// migrate_lb_ptcls / migrate_ptcls (src/pumipic_ptcl_ops.hpp:53-85) as ONE call: setUnsafeProcs is the
// routing rule of the count / pack passes (no new_elems / new_procs arrays: 16 B per slot less traffic
// and one pass less), then SellCSigma::migrate.  elem_ids_dev is in/out: sent particles read -1
// afterwards.
int pp_migrate_ptcls_begin(pp_ps* ps, int m_x, int m_xtgt, int* elem_ids_dev, const unsigned char* safe_dev,
                           const int* owners_dev, pp_comm* comm, const pp_mesh* mesh, int nmaps,
                           const int* const* v2v_dev, double* const* scatter_w_dev, double rmax, int gnr,
                           int gppr) {
  PP_REQUIRE(ps && ((safe_dev && owners_dev) || ps->capacity == 0), "pp_migrate_ptcls: null safe / owner arrays");
  return migrate_begin_rule(ps, m_x, m_xtgt, elem_ids_dev, RouteRule{nullptr, safe_dev, owners_dev}, comm, 0,
                            nullptr, nullptr, nullptr, 0, mesh, nmaps, v2v_dev, scatter_w_dev, rmax, gnr, gppr);
}
int pp_migrate_ptcls(pp_ps* ps, int m_x, int m_xtgt, int* elem_ids_dev, const unsigned char* safe_dev,
                     const int* owners_dev, pp_comm* comm, const pp_mesh* mesh, int nmaps,
                     const int* const* v2v_dev, double* const* scatter_w_dev, double rmax, int gnr, int gppr) {
  PP_REQUIRE(comm, "pp_migrate_ptcls: null communicator");
  PP_REQUIRE(comm->kind != 4 || comm->nranks == 1,
             "pp_migrate_ptcls: a local communicator needs pp_migrate_ptcls_begin on every virtual rank, then "
             "pp_ps_migrate_end on every virtual rank");
  int rc = pp_migrate_ptcls_begin(ps, m_x, m_xtgt, elem_ids_dev, safe_dev, owners_dev, comm, mesh, nmaps, v2v_dev,
                                  scatter_w_dev, rmax, gnr, gppr);
  if (rc) return rc;
  return pp_ps_migrate_end(ps, comm, nullptr, nullptr);
}
static int migrate_begin_rule(pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev, const RouteRule new_process_dev,
                              pp_comm* comm, int n_new, const int* new_elems_dev,
                              const void* const* new_info_dev, const int* gid2lid_dev, int64_t ngids,
                              const pp_mesh* mesh, int nmaps, const int* const* v2v_dev,
                              double* const* scatter_w_dev, double rmax, int gnr, int gppr) {
  PP_REQUIRE(ps && comm, "pp_ps_migrate: null structure / communicator");
  PP_REQUIRE(new_element_dev || ps->capacity == 0, "pp_ps_migrate: null routing arrays");
  PP_REQUIRE(n_new >= 0 && (n_new == 0 || (new_elems_dev && new_info_dev)), "pp_ps_migrate: bad new particles");
  PP_REQUIRE(nmaps >= 0 && (nmaps == 0 || (mesh && v2v_dev && scatter_w_dev)), "pp_ps_migrate: bad scatter arguments");
  const bool commit = m_x >= 0 || m_xtgt >= 0;
  if (commit) {
    PP_REQUIRE(m_x >= 0 && m_xtgt >= 0 && m_x < ps->nmembers && m_xtgt < ps->nmembers && m_x != m_xtgt,
               "pp_ps_migrate: bad member index");
    PP_REQUIRE(ps->member_bytes[ps->member_map[m_x]] == 8 && ps->member_bytes[ps->member_map[m_xtgt]] == 8 &&
                   ps->member_ncomp[ps->member_map[m_x]] == ps->member_ncomp[ps->member_map[m_xtgt]],
               "pp_ps_migrate: x and x_tgt must be double members of equal shape");
  }
  pp::MigratePending& P = comm->pend;
  PP_REQUIRE(!P.active, "pp_ps_migrate_begin: the previous migration on this communicator was not ended");
  // arrivals travel with their element's gid: with more than one rank the receiver needs a gid -> element
  // table.  Checked here, before anything is packed or marked as sent.
  PP_REQUIRE(comm->nranks == 1 || gid2lid_dev || !ps->has_gids || ps->gids_identity || ps->n_gid2lid > 0,
             "migration: the structure's element gids are too sparse for its own gid -> element table -- pass "
             "gid2lid_dev");
  pp::Range rg("pp_ps_migrate_begin");
  P = pp::MigratePending();
  P.ps = ps;
  P.m_x = commit ? m_x : -1;
  P.m_xtgt = commit ? m_xtgt : -1;
  P.new_element = new_element_dev;
  P.n_new = n_new;
  P.new_elems = new_elems_dev;
  P.new_info = new_info_dev;
  P.gid2lid = gid2lid_dev;
  P.ngids = ngids;
  P.mesh = mesh;
  P.nmaps = nmaps;
  for (int k = 0; k < nmaps; ++k) {
    P.v2v.push_back(v2v_dev[k]);
    P.outs.push_back(scatter_w_dev[k]);
  }
  P.rmax = rmax;
  P.gnr = gnr;
  P.gppr = gppr;
  const int n = comm->nranks;
  P.send_counts.assign((size_t)n, 0);
  P.recv_counts.assign((size_t)n, 0);
  P.recv_known = true;
  if (n > 1) {
    RecTable t{};
    int rc = build_rec_table(ps, t, P.m_x, P.m_xtgt, true);
    if (rc) return rc;
    P.rec_bytes = t.rec_words * 4;
    PP_HIP_CHECK(comm->d_counts.reserve(sizeof(int) * (size_t)n));
    int per_block = 0;
    rc = route_count(ps, new_element_dev, new_process_dev, comm->rank, n, comm->d_counts.as<int>(), &per_block);
    if (rc) return rc;
    rc = pp::comm_counts(comm, comm->d_counts.as<int>(), P.send_counts, P.recv_counts, &P.recv_known);
    if (rc) return rc;
    std::vector<int> start((size_t)n, 0);
    int64_t tot = 0;
    for (int r = 0; r < n; ++r) {
      start[(size_t)r] = (int)tot;
      if (r != comm->rank) tot += P.send_counts[(size_t)r];
    }
    P.n_send = tot;
    PP_HIP_CHECK(comm->d_send.reserve((size_t)std::max<int64_t>(tot, 1) * P.rec_bytes));
    if (tot > 0) {
      rc = route_pack(ps, P.m_x, P.m_xtgt, new_element_dev, new_process_dev, comm->rank, n, per_block,
                      start.data(), comm->d_send.p);
      if (rc) return rc;
    }
    if (comm->kind == 4) {  // publish to the other virtual ranks
      pp::LocalWorld* w = comm->world.get();
      rc = pp::local_publish(w, comm->rank, P.send_counts, comm->d_send.p, P.rec_bytes);
      if (rc) return rc;
    }
  }
  P.active = true;
  return PP_OK;
}

int pp_ps_migrate_end(pp_ps* ps, pp_comm* comm, int* n_sent, int* n_received) {
  PP_REQUIRE(ps && comm, "pp_ps_migrate_end: null argument");
  pp::MigratePending& P = comm->pend;
  PP_REQUIRE(P.active && P.ps == ps, "pp_ps_migrate_end: no migration of this structure was begun on this communicator");
  pp::Range rg("pp_ps_migrate_end");
  P.active = false;
  // a virtual rank leaves the round on EVERY exit path: a failed exchange must not leave the
  // channel's begun[] flags set, or every later pp_ps_migrate_begin on that world fails
  struct LocalRoundGuard {
    pp_comm* c;
    ~LocalRoundGuard() {
      if (c->kind == 4) pp::local_ended(c->world.get(), c->rank);
    }
  } round_guard{comm};
  int rc;
  int64_t nrecv = 0;
  void* d_recv = nullptr;
  if (comm->nranks > 1) {
    if (comm->kind == 4) {
      rc = pp::local_all_begun(comm->world.get());
      if (rc) return rc;
    }
    rc = pp::comm_exchange_records(comm, comm->d_send.p, P.send_counts, P.recv_counts, P.rec_bytes, &d_recv);
    if (rc) return rc;
    for (int r = 0; r < comm->nranks; ++r)
      if (r != comm->rank) nrecv += P.recv_counts[(size_t)r];
  }
  if (n_sent) *n_sent = (int)P.n_send;
  if (n_received) *n_received = (int)nrecv;
  rc = rebuild_from_records(ps, P.m_x, P.m_xtgt, P.new_element, (int)nrecv, d_recv, P.gid2lid, P.ngids, P.n_new,
                            P.new_elems, P.new_info, P.mesh, P.nmaps, P.v2v.data(), P.outs.data(), P.rmax,
                            P.gnr, P.gppr);
  return rc;
}

int pp_ps_migrate_scatter(pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev,
                          const int* new_process_dev, pp_comm* comm, int n_new,
                          const int* new_elems_dev, const void* const* new_info_dev,
                          const int* gid2lid_dev, int64_t ngids, const pp_mesh* mesh, int nmaps,
                          const int* const* v2v_dev, double* const* scatter_w_dev, double rmax,
                          int gnr, int gppr) {
  PP_REQUIRE(comm, "pp_ps_migrate: null communicator");
  PP_REQUIRE(comm->kind != 4 || comm->nranks == 1,
             "pp_ps_migrate: a local communicator needs pp_ps_migrate_begin on every virtual rank, then "
             "pp_ps_migrate_end on every virtual rank");
  int rc = pp_ps_migrate_begin(ps, m_x, m_xtgt, new_element_dev, new_process_dev, comm, n_new, new_elems_dev,
                               new_info_dev, gid2lid_dev, ngids, mesh, nmaps, v2v_dev, scatter_w_dev, rmax, gnr,
                               gppr);
  if (rc) return rc;
  return pp_ps_migrate_end(ps, comm, nullptr, nullptr);
}

int pp_ps_migrate(pp_ps* ps, int* new_element_dev, const int* new_process_dev, pp_comm* comm) {
  return pp_ps_migrate_scatter(ps, -1, -1, new_element_dev, new_process_dev, comm, 0, nullptr, nullptr, nullptr,
                               0, nullptr, 0, nullptr, nullptr, 0.0, 2, 1);
}

int pp_set_unsafe_procs(const pp_ps* ps, const int* elems_dev, const unsigned char* safe_dev,
                        const int* owners_dev, int comm_rank, int* new_elems_dev,
                        int* new_procs_dev) {
  PP_REQUIRE(ps && elems_dev && safe_dev && owners_dev && new_elems_dev && new_procs_dev,
             "pp_set_unsafe_procs: null argument");
  if (ps->num_ptcls == 0 || ps->capacity == 0) return PP_OK;
  k_unsafe<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), elems_dev, safe_dev, owners_dev, comm_rank, new_elems_dev,
      new_procs_dev);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_ps_migrate_count(const pp_ps* ps, const int* new_element_dev, const int* new_process_dev,
                        int comm_rank, int nranks, int* send_counts_host) {
  PP_REQUIRE(ps && new_element_dev && new_process_dev && send_counts_host && nranks > 0,
             "pp_ps_migrate_count: bad argument");
  hipStream_t st = pp::stream();
  pp::DevBuf& cnt = scratch(3);
  PP_HIP_CHECK(cnt.reserve(sizeof(int) * (size_t)nranks));
  int per_block = 0;
  int rc = route_count(ps, new_element_dev, RouteRule{new_process_dev, nullptr, nullptr}, comm_rank, nranks,
                       cnt.as<int>(), &per_block);
  if (rc) return rc;
  PP_HIP_CHECK(hipMemcpyAsync(send_counts_host, cnt.p, sizeof(int) * (size_t)nranks,
                              hipMemcpyDeviceToHost, st));
  PP_HIP_CHECK(hipStreamSynchronize(st));
  return PP_OK;
}

int pp_ps_migrate_pack(const pp_ps* ps, int* new_element_dev, const int* new_process_dev,
                       int comm_rank, int nranks, const int* send_counts_host,
                       int64_t* send_gid_dev, void* const* send_info_dev) {
  PP_REQUIRE(ps && new_element_dev && new_process_dev && send_counts_host && nranks > 0,
             "pp_ps_migrate_pack: bad argument");
  long long total = 0;
  std::vector<int> start((size_t)nranks, 0);
  for (int r = 0; r < nranks; ++r) {
    start[r] = (int)total;
    total += send_counts_host[r];
  }
  if (total == 0) return PP_OK;
  PP_REQUIRE(send_gid_dev && send_info_dev, "pp_ps_migrate_pack: null send buffers");
  if (int rc = pp::ps_ready(ps)) return rc;
  hipStream_t st = pp::stream();
  pp::DevBuf& cur = scratch(0);
  PP_HIP_CHECK(cur.reserve(sizeof(int) * (size_t)nranks));
  PP_HIP_CHECK(hipMemcpyAsync(cur.p, start.data(), sizeof(int) * (size_t)nranks,
                              hipMemcpyHostToDevice, st));
  PackArgs a{};
  a.nmembers = ps->nmembers;
  for (int m = 0; m < ps->nmembers; ++m) {
    const int s = ps->member_map[m];
    a.src[m] = ps->data[s].p;
    a.dst[m] = send_info_dev[m];
    a.bytes[m] = ps->member_bytes[s];
    a.ncomp[m] = ps->member_ncomp[s];
  }
  a.src_stride = ps->stride;
  a.dst_stride = total;
  k_pack<<<grid_for(ps->capacity), kBlock, 0, st>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), new_element_dev, new_process_dev, comm_rank,
      nranks, cur.as<int>(), ps->has_gids ? ps->d_gids.as<long long>() : nullptr,
      (long long*)send_gid_dev, a);
  PP_LAUNCH_CHECK();
  PP_HIP_CHECK(hipStreamSynchronize(st));  // `start` and `cur` go out of scope
  return PP_OK;
}

int pp_bfs_buffer_layers(const pp_mesh* mesh, int bridge_dim, int comm_rank, int comm_size,
                         int safe_layers, int ghost_layers, const int* owner_dev,
                         unsigned char* is_safe_dev, int* has_part_host) {
  PP_REQUIRE(mesh && owner_dev && is_safe_dev && has_part_host, "pp_bfs_buffer_layers: null argument");
  PP_REQUIRE(comm_size > 0 && comm_rank >= 0 && comm_rank < comm_size && safe_layers >= 0 &&
                 ghost_layers >= 0,
             "pp_bfs_buffer_layers: bad rank / layer counts");
  int nb;
  const int *off, *vals;
  int rc = bridge_adjacency(mesh, bridge_dim, &nb, &off, &vals);
  if (rc) return rc;
  hipStream_t st = pp::stream();
  const int ne = mesh->nelems;
  pp::DevBuf visited, next, part;
  PP_HIP_CHECK(visited.reserve(sizeof(int) * (size_t)std::max(ne, 1)));
  PP_HIP_CHECK(next.reserve(sizeof(int) * (size_t)std::max(ne, 1)));
  PP_HIP_CHECK(part.reserve(sizeof(int) * (size_t)comm_size));
  PP_HIP_CHECK(hipMemsetAsync(part.p, 0, sizeof(int) * (size_t)comm_size, st));
  const int one = 1;  // initSelfPart
  PP_HIP_CHECK(hipMemcpyAsync(part.as<int>() + comm_rank, &one, sizeof(int), hipMemcpyHostToDevice, st));
  if (ne > 0) {
    k_bfs_init<<<grid_for(ne), kBlock, 0, st>>>(ne, owner_dev, comm_rank, visited.as<int>(),
                                               next.as<int>(), is_safe_dev);
    for (int i = 0; i < ghost_layers || i < safe_layers; ++i) {
      if (nb > 0)
        k_bfs_sweep<<<grid_for(nb), kBlock, 0, st>>>(nb, off, vals, visited.as<int>(), next.as<int>());
      k_bfs_copy<<<grid_for(ne), kBlock, 0, st>>>(ne, i, safe_layers, ghost_layers, owner_dev,
                                                 visited.as<int>(), next.as<int>(), is_safe_dev,
                                                 part.as<int>());
    }
  }
  PP_LAUNCH_CHECK();
  PP_HIP_CHECK(hipMemcpyAsync(has_part_host, part.p, sizeof(int) * (size_t)comm_size,
                              hipMemcpyDeviceToHost, st));
  PP_HIP_CHECK(hipStreamSynchronize(st));
  return PP_OK;
}

int pp_bfs_safe_inward(const pp_mesh* mesh, int bridge_dim, int comm_rank, int comm_size,
                       int safe_layers, const int* owner_dev, const int* has_part_host,
                       unsigned char* safe_dev) {
  PP_REQUIRE(mesh && owner_dev && has_part_host && safe_dev, "pp_bfs_safe_inward: null argument");
  PP_REQUIRE(comm_size > 0 && comm_rank >= 0 && comm_rank < comm_size && safe_layers >= 0,
             "pp_bfs_safe_inward: bad rank / layer count");
  int nb;
  const int *off, *vals;
  int rc = bridge_adjacency(mesh, bridge_dim, &nb, &off, &vals);
  if (rc) return rc;
  hipStream_t st = pp::stream();
  const int ne = mesh->nelems;
  if (ne == 0) return PP_OK;
  pp::DevBuf visited, next, part;
  PP_HIP_CHECK(visited.reserve(sizeof(int) * (size_t)ne));
  PP_HIP_CHECK(next.reserve(sizeof(int) * (size_t)ne));
  PP_HIP_CHECK(part.reserve(sizeof(int) * (size_t)comm_size));
  PP_HIP_CHECK(hipMemcpyAsync(part.p, has_part_host, sizeof(int) * (size_t)comm_size,
                              hipMemcpyHostToDevice, st));
  k_bfs_inward_init<<<grid_for(ne), kBlock, 0, st>>>(ne, owner_dev, part.as<int>(), visited.as<int>(),
                                                    next.as<int>());
  for (int i = 0; i < safe_layers; ++i) {
    if (nb > 0)
      k_bfs_sweep<<<grid_for(nb), kBlock, 0, st>>>(nb, off, vals, visited.as<int>(), next.as<int>());
    PP_HIP_CHECK(hipMemcpyAsync(visited.p, next.p, sizeof(int) * (size_t)ne, hipMemcpyDeviceToDevice, st));
  }
  k_bfs_inward_set<<<grid_for(ne), kBlock, 0, st>>>(ne, owner_dev, comm_rank, visited.as<int>(), safe_dev);
  PP_LAUNCH_CHECK();
  PP_HIP_CHECK(hipStreamSynchronize(st));  // the temporaries go out of scope
  return PP_OK;
}

}  // extern "C"
