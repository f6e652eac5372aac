// pp_ps_move.hpp -- the data movement of a rebuild (included by pp_ps.hip only): every member of every particle from
// its old slot to its new one.
//
// Reference: particle_structs/src/support/psMemberType.h:72-112 (CopyPSToPS / ShuffleParticles),
// MemberTypeLibraries.h:134-265 (one gather/scatter kernel per member type, each re-reading new_element /
// new_indices / the mask: SURVEY A14).  Here: ONE pass over the old layout packs each particle into a 16-byte-aligned
// record at its new position (whole 64-B sectors per particle, runs of records per row when they are stored
// row-major), a second pass over the new layout writes the SoA arrays coalesced -- or is deferred, when the next
// consumer reads the records themselves (pp_ps::lazy_rec).  Also the slot-to-slot copy for members the record
// cannot hold and the insertion of new particles.
#pragma once
#include "pp_internal.hpp"

namespace {

struct MoveArgs {
  int nmembers;
  const void* src[8];
  void* dst[8];
  int bytes[8];
  int ncomp[8];
  long long src_stride, dst_stride;
  // fused updatePtclPositions (test/pseudoXGCm.cpp:102-114): member commit_x takes the values of
  // member commit_xt, member commit_xt is written as zeros.  -1 = plain copy.
  int commit_x, commit_xt;
};
__device__ __forceinline__ void copy_members(const MoveArgs& a, long long from, long long to) {
  for (int m = 0; m < a.nmembers; ++m) {
    const int nc = a.ncomp[m];
    if (a.bytes[m] == 8) {
      const unsigned long long* s =
          (const unsigned long long*)(m == a.commit_x ? a.src[a.commit_xt] : a.src[m]);
      unsigned long long* d = (unsigned long long*)a.dst[m];
      if (m == a.commit_xt)
        for (int c = 0; c < nc; ++c) d[c * a.dst_stride + to] = 0ull;
      else
        for (int c = 0; c < nc; ++c) d[c * a.dst_stride + to] = s[c * a.src_stride + from];
    } else if (a.bytes[m] == 4) {
      const unsigned* s = (const unsigned*)a.src[m];
      unsigned* d = (unsigned*)a.dst[m];
      for (int c = 0; c < nc; ++c) d[c * a.dst_stride + to] = s[c * a.src_stride + from];
    } else if (a.bytes[m] == 2) {
      const unsigned short* s = (const unsigned short*)a.src[m];
      unsigned short* d = (unsigned short*)a.dst[m];
      for (int c = 0; c < nc; ++c) d[c * a.dst_stride + to] = s[c * a.src_stride + from];
    } else {
      const unsigned char* s = (const unsigned char*)a.src[m];
      unsigned char* d = (unsigned char*)a.dst[m];
      for (int c = 0; c < nc; ++c) d[c * a.dst_stride + to] = s[c * a.src_stride + from];
    }
  }
}
// new particles (set_new_particle + CopyViewsToViews, SCS_rebuild.h:277-289)
__global__ void k_add_scs(int n_new, const int* __restrict__ new_elems,
                          const int* __restrict__ e2r_new, int C_new, int* __restrict__ row_cursor,
                          const int* __restrict__ rank_new, const int* __restrict__ elem_slot0,
                          unsigned char* __restrict__ new_mask, MoveArgs a,
                          const int* __restrict__ go) {
  if (!*go) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int e = new_elems[i];
  // rank_new: slot from the rank the counting pass returned; else the row cursor (direct move)
  const int idx = rank_new ? elem_slot0[e] + rank_new[i] * C_new : atomicAdd(&row_cursor[e2r_new[e]], C_new);
  new_mask[idx] = 1;
  copy_members(a, i, idx);
}
// new particles of a rebuild whose second pass is deferred (pseudoXGCm particle type, committed layout of
// k_move_pack: words 0-5 member commit_x, 6 / 7 the last two 4-byte members from the back; the first one beside it)
__global__ void k_add_rec(int n_new, const int* __restrict__ new_elems, const int* __restrict__ rank_new,
                          const int* __restrict__ elem_slot0, int C_new, const unsigned long long* __restrict__ x,
                          const unsigned* __restrict__ m2, const unsigned* __restrict__ m3,
                          const unsigned* __restrict__ m4, uint4* __restrict__ aos, unsigned* __restrict__ side,
                          const int* __restrict__ go) {
  if (!*go) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  // (elem_slot0 / C_new: first slot of the row and the slot distance of consecutive ranks -- or, with row-major
  // records, the row's first record and 1)
  const long long idx = elem_slot0[new_elems[i]] + (long long)rank_new[i] * C_new;
  const unsigned long long x0 = x[i], x1 = x[(size_t)n_new + i], x2 = x[2 * (size_t)n_new + i];
  uint4* r = aos + idx * 2;
  r[0] = make_uint4((unsigned)x0, (unsigned)(x0 >> 32), (unsigned)x1, (unsigned)(x1 >> 32));
  r[1] = make_uint4((unsigned)x2, (unsigned)(x2 >> 32), m4[i], m3[i]);
  side[idx] = m2[i];
}
// Row-tiled move (SCS): thread = (old tile, row).  Stayers of the thread reserve their slots in the
// new row with ONE atomic (n_stay * C) and are written in a second sweep; movers take slots one
// by one.  Reads are coalesced 64-slot runs; every member of a particle moves in this one pass.
__global__ void k_move_tiled(const int* __restrict__ ntiles_dev, int C, int TP,
                             const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                             const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                             const unsigned char* __restrict__ mask,
                             const int* __restrict__ new_element, const int* __restrict__ e2r_new,
                             int C_new, int* __restrict__ row_cursor,
                             unsigned char* __restrict__ new_mask, MoveArgs a,
                             const int* __restrict__ go) {
  if (!*go) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tile = (int)(g / C), r = (int)(g - (long long)tile * C);
  if (tile >= *ntiles_dev) return;
  const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
  const int start = chunk_start[c] + r, pend = min(p0 + TP, chunk_width[c]);
  const int e = r2e[c * C + r];
  unsigned stay = 0;  // TP <= 32
  for (int p = p0; p < pend; ++p) {
    const int pid = start + p * C;
    if (!mask[pid]) continue;
    const int ne_ = new_element[pid];
    if (ne_ == -1) continue;
    if (ne_ == e) {
      stay |= 1u << (p - p0);
    } else {
      const int idx = atomicAdd(&row_cursor[e2r_new[ne_]], C_new);
      copy_members(a, pid, idx);
    }
  }
  if (stay) {
    int idx = atomicAdd(&row_cursor[e2r_new[e]], __popc(stay) * C_new);
    for (int p = p0; p < pend; ++p)
      if (stay & (1u << (p - p0))) {
        copy_members(a, start + p * C, idx);
        idx += C_new;
      }
  }
}
// ---- AoS-staged move (SCS).  The row order of a Sell-C-sigma structure is a function of the
// per-element counts, so after a rebuild adjacent old rows land in unrelated new rows: writing
// the members straight into the new SoA scatters 4/8-byte stores over as many cache lines
// (measured 1.6 ms for 10 M particles, ~8x write amplification).  Instead
//   pass 1  reads the old SoA coalesced, packs each particle into ONE 16-byte-aligned record and
//           writes it to aos[new_slot] (whole 32/64-B sectors per particle);
//   pass 2  walks the NEW layout, reads aos[slot] contiguously and writes the new SoA coalesced.
// Word table: record word w of slot pid lives at src[w] + pid*sscale[w] (sscale < 0: constant 0,
// used for the fused updatePtclPositions), and goes to dst[w] + slot*dscale[w].
// Entry table: 8-byte components are moved with 64-bit accesses (record words 2i, 2i+1 counted
// from the front), 4-byte components with 32-bit accesses (record words counted from the back,
// so that every register index is static); destinations that receive 0 (x_tgt of the fused
// updatePtclPositions) are never staged, pass 2 writes them as plain coalesced zero stores.
constexpr int kMax8 = 30, kMax4 = 16;
struct WordTable {
  int n8, n4, nz8, nz4;
  const char* src8[kMax8];
  char* dst8[kMax8];
  const char* src4[kMax4];
  char* dst4[kMax4];
  char* z8[8];
  char* z4[8];
  // ONE 4-byte member that travels beside the record instead of inside it (the pseudoXGCm particle: origin + phi + b
  // are 32 B, the third member would make it 36 -> a 64-B record).  Pass 1 reads it at side_src + old slot * 4 and
  // stores it to `side` + record * 4 (an argument of the kernels: an array in RECORD order); pass 2 reads it there
  // and writes side_dst + new slot * 4 (null: not wanted).
  const char* side_src;
  char* side_dst;
};
// pass 1b: one thread per old slot packs its record; the wave transposes through LDS so that
// NQ adjacent lanes store one whole record (full 64-B sectors leave the CU already merged:
// 0.21 vs 0.27 ms per 10 M random records, tools/ub_scatter.hip)
// SCS passes `rs` (rank -> slot translation: slot = start of the new row + rank*C); CSR passes the
// slot directly.
struct RankToSlot {
  const int* new_element;  // element of every source particle
  const int* elem_slot0;   // first slot of the element's new row (SCS) / new offsets (CSR)
  int step;                // slot distance between consecutive ranks: C (SCS) / 1 (CSR)
};
template <int NQ>
__global__ void k_move_pack(int capacity, const int* __restrict__ new_idx, RankToSlot rs,
                            uint4* __restrict__ aos, WordTable t, const int* __restrict__ go,
                            unsigned* __restrict__ side = nullptr) {
  if (go && !*go) return;
  __shared__ uint4 st[4][64][NQ + 1];
  __shared__ int sd[4][64];
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int rk = (pid < capacity) ? new_idx[pid] : -1;
  int idx = -1;
  if (rk >= 0) {
    // the member loads below depend only on rk >= 0: they are in flight while the two dependent
    // loads of the slot translation return
    idx = rs.elem_slot0[rs.new_element[pid]] + rk * rs.step;
    unsigned v[NQ * 4];
#pragma unroll
    for (int i = 0; i < NQ * 4; ++i) v[i] = 0u;
#pragma unroll
    for (int i = 0; i < NQ * 2; ++i)
      if (i < t.n8) {
        const unsigned long long d =  // (a null source: a component that is logically zero, pp_ps::zero_z_pending)
            t.src8[i] ? __builtin_nontemporal_load((const unsigned long long*)(t.src8[i] + (long long)pid * 8)) : 0ull;
        v[2 * i] = (unsigned)d;
        v[2 * i + 1] = (unsigned)(d >> 32);
      }
#pragma unroll
    for (int j = 0; j < (NQ * 4 < kMax4 ? NQ * 4 : kMax4); ++j)
      if (j < t.n4)
        v[NQ * 4 - 1 - j] = __builtin_nontemporal_load((const unsigned*)(t.src4[j] + (long long)pid * 4));
#pragma unroll
    for (int q = 0; q < NQ; ++q) st[w][l][q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    if (side) side[idx] = __builtin_nontemporal_load((const unsigned*)(t.side_src + (long long)pid * 4));
  }
  sd[w][l] = idx;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int item = j * 64 + l, rec = item / NQ, part = item % NQ;
    const int d = sd[w][rec];
    if (d >= 0) {  // non-temporal: the record is read once, by pass 2 (c3 -1.4 %, 160-B particles -4 %)
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const uint4 x = st[w][rec][part];
      v4u y;
      y.x = x.x;
      y.y = x.y;
      y.z = x.z;
      y.w = x.w;
      __builtin_nontemporal_store(y, (v4u*)(aos + (long long)d * NQ + part));
    }
  }
}
// The same pass for records that stay the particle data (pp_ps::rec_rm): destination = first RECORD of the new
// row + rank, i.e. the particles of a row are consecutive records.  A block holds 4 columns x 64 rows; the
// stayers of a row in those columns carry consecutive ranks (k_count_tiled hands a run its ranks in column
// order), so the transpose goes through LDS at BLOCK level and the block's records leave row by row: up to
// 256 contiguous bytes per row instead of four 64-byte stores 4 KB apart (c3: 294 -> 224 us for the pass).
template <int NQ>
__global__ void k_move_pack_rm(int capacity, const int* __restrict__ new_idx, RankToSlot rs,
                               uint4* __restrict__ aos, WordTable t, const int* __restrict__ go, int wide,
                               pp::HotRow hot = pp::HotRow{}, unsigned hot_blocks = 0u,
                               unsigned* __restrict__ side = nullptr, uint4* __restrict__ split_hot = nullptr) {
  // split_hot (NQ == 2 only, pp_ps::rec_split): quad 0 of a record goes to aos[record], quad 1 to split_hot[record]
  // -- two arrays of 16-B halves instead of one of 32-B records (the 2-D push reads the second halves only)
  if (go && !*go) return;
  __shared__ uint4 st[256][NQ + 1];
  __shared__ int sd[256];
  __shared__ unsigned ss[256];  // the member that travels beside the record (WordTable::side_src)
  // which old slot: 4 columns x 64 rows of the block's 256 consecutive slots, or (wide: chunk height 64) 8 columns x
  // 32 rows -- block pairs share 8 columns, so a row's run is up to 8 records = 512 contiguous bytes
  const int tid = threadIdx.x;
  int pid, li;  // li = LDS index: records of one row adjacent
  // (of the main blocks) XCD-aware: blockIdx i runs on XCD i % 8, each with an L2 of its own.  The stayers of a row
  // leave as ONE contiguous stretch of records, written piecewise by the blocks of the row's column groups; when those
  // blocks run on one XCD the ragged ends of neighbouring pieces meet in its L2 and leave as whole lines -- so every
  // XCD takes a contiguous eighth of the blocks (the grid is a multiple of 16)
  // (round 5, same box: c3 0.57 -> 0.53 ms, 2dc3 0.59 -> 0.54)
  const unsigned nb = gridDim.x - hot_blocks;
  const unsigned bid = (nb & 7u) ? blockIdx.x - hot_blocks  // (a grid that is no multiple of 8: the hardware's order)
                                 : (blockIdx.x & 7u) * (nb >> 3) + ((blockIdx.x - hot_blocks) >> 3);
  if (blockIdx.x < hot_blocks) {
    // (pp_ps::hot; `capacity` ends the main blocks' slots where these columns begin) 256 columns of the over-full
    // row: one run of up to 16 KB.  The first blocks of the grid, as in the histogram.
    const int p = hot.c1p + (int)blockIdx.x * 256 + tid;
    pid = p < hot.w ? hot.start + p * 64 + hot.row : 0x7fffffff;
    li = tid;
    capacity = 0x7fffffff;
  } else if (wide) {  // wide = log2(columns per block), chunk height 64: 2^wide columns x (256 >> wide) rows
    const int nrow = 256 >> wide, col = tid / nrow, row = tid - col * nrow;
    const int sub = bid & ((64 / nrow) - 1);  // which group of rows of the column block
    pid = (bid / (64 / nrow)) * (64 << wide) + col * 64 + sub * nrow + row;
    li = (row << wide) + col;
  } else {
    pid = bid * 256 + tid;
    li = (tid & 63) * 4 + (tid >> 6);
  }
  const int rk = (pid < capacity) ? new_idx[pid] : -1;
  int idx = -1;
  if (rk >= 0) {
    idx = rs.elem_slot0[rs.new_element[pid]] + rk * rs.step;
    unsigned v[NQ * 4];
#pragma unroll
    for (int i = 0; i < NQ * 4; ++i) v[i] = 0u;
#pragma unroll
    for (int i = 0; i < NQ * 2; ++i)
      if (i < t.n8) {
        const unsigned long long d =  // (a null source: a component that is logically zero, pp_ps::zero_z_pending)
            t.src8[i] ? __builtin_nontemporal_load((const unsigned long long*)(t.src8[i] + (long long)pid * 8)) : 0ull;
        v[2 * i] = (unsigned)d;
        v[2 * i + 1] = (unsigned)(d >> 32);
      }
#pragma unroll
    for (int j = 0; j < (NQ * 4 < kMax4 ? NQ * 4 : kMax4); ++j)
      if (j < t.n4)
        v[NQ * 4 - 1 - j] = __builtin_nontemporal_load((const unsigned*)(t.src4[j] + (long long)pid * 4));
#pragma unroll
    for (int q = 0; q < NQ; ++q) st[li][q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    if (side) ss[li] = __builtin_nontemporal_load((const unsigned*)(t.side_src + (long long)pid * 4));
  }
  sd[li] = idx;
  __syncthreads();
  if (side) {  // thread i stores for LDS index i: the records of a row are adjacent indices, their places adjacent words
    const int d = sd[tid];
    if (d >= 0) side[d] = ss[tid];
  }
  if (NQ == 2 && split_hot) {
    // split records: thread i stores both halves of LDS index i -- consecutive lanes write consecutive 16-B pieces
    // of ONE array (a row's run of 32 columns is 512 contiguous bytes in each)
    const int d = sd[tid];
    if (d >= 0) {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        const uint4 x = st[tid][part];
        v4u y;
        y.x = x.x;
        y.y = x.y;
        y.z = x.z;
        y.w = x.w;
        __builtin_nontemporal_store(y, (v4u*)((part ? split_hot : aos) + (long long)d));
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int item = j * 256 + tid, rec = item / NQ, part = item % NQ;  // the records of one row are adjacent items
    const int d = sd[rec];
    if (d >= 0) {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const uint4 x = st[rec][part];
      v4u y;
      y.x = x.x;
      y.y = x.y;
      y.z = x.z;
      y.w = x.w;
      __builtin_nontemporal_store(y, (v4u*)(aos + (long long)d * NQ + part));
    }
  }
}
// Pass 1 of a rebuild that FOLLOWS a rebuild whose second pass was deferred (pp_ps::lazy_rec == 3): the particles ARE
// the 16*NQ-byte records of the previous re-layout, one per old slot in slot order -- nothing was asked of the member
// arrays in between (performance_tests/ps_combo160.cpp:205-232: redistribute + rebuild, a hundred times in a row).
// A wave takes 64 consecutive old slots: lane l computes the destination of slot l, then the NQ*64 quads of the
// run are loaded as NQ fully coalesced instructions (1 KB each) and every quad goes to its record's new place --
// NQ adjacent lanes store one whole record.  No staging in LDS: the destination travels by shuffle.
// 16*NQ B read + 16*NQ B written per particle, against 2 x (16*NQ + the member bytes) for SoA -> records -> SoA.
template <int NQ>
__global__ void k_move_pack_rec(int capacity, const int* __restrict__ new_idx, RankToSlot rs,
                                const uint4* __restrict__ src, uint4* __restrict__ aos,
                                const int* __restrict__ go) {
  if (go && !*go) return;
  // XCD order (see k_move_pack_rm): consecutive blocks on one XCD, so that the stayers' runs of neighbouring blocks
  // (CSR: slot order is element order) meet in one L2 -- ps_combo160 50 k / 50 M, CSR: 5.28 -> 5.14 ms per rebuild
  unsigned vb = blockIdx.x;
  const unsigned nb8 = gridDim.x & ~7u;
  if (vb < nb8) vb = (vb & 7u) * (nb8 >> 3) + (vb >> 3);
  const int pid = vb * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & 63;
  const long long base = pid - l;  // first slot of the wave's run
  const int rk = (pid < capacity) ? new_idx[pid] : -1;
  int idx = -1;
  if (rk >= 0) idx = rs.elem_slot0[rs.new_element[pid]] + rk * rs.step;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int item = j * 64 + l, rec = item / NQ, part = item - rec * NQ;
    const int d = __shfl(idx, rec);
    if (d >= 0) {
      const v4u q = __builtin_nontemporal_load((const v4u*)(src + (base + rec) * NQ + part));
      __builtin_nontemporal_store(q, (v4u*)(aos + (long long)d * NQ + part));
    }
  }
}
template <int NQ>
__global__ void k_move_unpack(const int* __restrict__ ntiles_dev, int C, int TP,
                              const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                              const int* __restrict__ chunk_width,
                              const unsigned char* __restrict__ new_mask,
                              const uint4* __restrict__ aos, WordTable t, const int* __restrict__ go,
                              int rec_rm = 0, const unsigned* __restrict__ side = nullptr,
                              const uint4* __restrict__ split_hot = nullptr) {
  // split_hot (NQ == 2, row-major records): quad 0 of record i is aos[i], quad 1 split_hot[i] (k_move_pack_rm);
  // a table without 4-byte destinations (only the origin is wanted, pp_ps::lazy_rec == 2) leaves the second halves
  // unread
  if (!*go) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tile = (int)(g / C), r = (int)(g - (long long)tile * C);
  if (tile >= *ntiles_dev) return;
  const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
  const int start = chunk_start[c] + r, pend = min(p0 + TP, chunk_width[c]);
  const long long rbase = pp_rec_row0(chunk_start[c], c, r, chunk_width[c], C);  // row-major records (pp_ps::rec_rm)
  const bool want_side = side != nullptr && t.side_dst != nullptr;
  auto put = [&](int slot, const unsigned* w) {
#pragma unroll
    for (int i = 0; i < NQ * 2; ++i)
      if (i < t.n8)
        __builtin_nontemporal_store(((unsigned long long)w[2 * i + 1] << 32) | w[2 * i],
                                    (unsigned long long*)(t.dst8[i] + (long long)slot * 8));
#pragma unroll
    for (int j = 0; j < (NQ * 4 < kMax4 ? NQ * 4 : kMax4); ++j)
      if (j < t.n4) __builtin_nontemporal_store(w[NQ * 4 - 1 - j], (unsigned*)(t.dst4[j] + (long long)slot * 4));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < t.nz8) __builtin_nontemporal_store(0ull, (unsigned long long*)(t.z8[i] + (long long)slot * 8));
      if (i < t.nz4) __builtin_nontemporal_store(0u, (unsigned*)(t.z4[i] + (long long)slot * 4));
    }
  };
  if (NQ <= 4 && rec_rm) {
    // Row-major records of at most 64 B: a row's records of a tile are contiguous (columns (2j, 2j + 1) share a
    // 128-B line).  Read one column at a time (four loads, nine stores, the next four loads), the second half of
    // a line is asked for long after the first -- with 32 KB of lines in flight per wave the L2 has dropped it by
    // then: the PMC counters read 1 114 MB fetched for 640 MB of records (profiles/traffic_driver.json; 662 MB
    // now).  KC columns' records are loaded back to back, then stored.
    // (same-box A/B of the drop-in driver, 3 x 100 steps: 1.244 against 1.252 ms with one column at a time, 1.24
    // with four -- the pass is not bound by these bytes; the traffic is what went down)
    constexpr int KC = NQ <= 2 ? 4 : 2;  // (the records of one 128-B line)
    for (int p = p0; p < pend; p += KC) {
      bool m[KC];
      const uint4* sp = aos + (rbase + p) * NQ;
      unsigned w[KC][NQ * 4];
      unsigned sv[KC];
      bool any = false;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        m[k] = (p + k < pend) && new_mask[start + (p + k) * C] != 0;
        any = any || m[k];
      }
      if (!any) continue;
#pragma unroll
      for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (NQ == 2 && split_hot) {
            if (m[k] && (q == 0 || t.n4 > 0)) v = (q ? split_hot : aos)[rbase + p + k];
          } else if (m[k]) {
            v = sp[k * NQ + q];
          }
          w[k][4 * q] = v.x;
          w[k][4 * q + 1] = v.y;
          w[k][4 * q + 2] = v.z;
          w[k][4 * q + 3] = v.w;
        }
      if (want_side) {
#pragma unroll
        for (int k = 0; k < KC; ++k) sv[k] = m[k] ? side[rbase + p + k] : 0u;
      }
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (m[k]) {
          put(start + (p + k) * C, w[k]);
          if (want_side) __builtin_nontemporal_store(sv[k], (unsigned*)(t.side_dst + (long long)(start + (p + k) * C) * 4));
        }
    }
    return;
  }
  for (int p = p0; p < pend; ++p) {
    const int slot = start + p * C;
    if (!new_mask[slot]) continue;
    const uint4* sp = aos + (rec_rm ? rbase + p : (long long)slot) * NQ;
    unsigned w[NQ * 4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      // plain loads: the lanes of a wave read 64-B-strided records, every line serves several
      // instructions; a non-temporal hint throws that reuse away (c3 +5 %, 160-B particles +20 %)
      const uint4 v = sp[q];
      w[4 * q] = v.x;
      w[4 * q + 1] = v.y;
      w[4 * q + 2] = v.z;
      w[4 * q + 3] = v.w;
    }
    put(slot, w);
    if (want_side)
      __builtin_nontemporal_store(side[rec_rm ? rbase + p : (long long)slot], (unsigned*)(t.side_dst + (long long)slot * 4));
  }
}
// CSR staged move (same two passes as SCS): slot assignment by the element cursor, pack through
// the LDS transpose into 64-B-multiple records, then a flat pass that writes the new SoA coalesced
template <int NQ>
__global__ void k_unpack_flat(int n, const uint4* __restrict__ aos, WordTable t,
                              const int* __restrict__ go) {
  if (!*go) return;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const uint4* sp = aos + (long long)slot * NQ;
  unsigned w[NQ * 4];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const uint4 v = sp[q];
    w[4 * q] = v.x;
    w[4 * q + 1] = v.y;
    w[4 * q + 2] = v.z;
    w[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < NQ * 2; ++i)
    if (i < t.n8) *(uint2*)(t.dst8[i] + (long long)slot * 8) = make_uint2(w[2 * i], w[2 * i + 1]);
#pragma unroll
  for (int j = 0; j < (NQ * 4 < kMax4 ? NQ * 4 : kMax4); ++j)
    if (j < t.n4) *(unsigned*)(t.dst4[j] + (long long)slot * 4) = w[NQ * 4 - 1 - j];
}

}  // namespace
