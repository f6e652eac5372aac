// pp_ps.hip -- Sell-C-sigma and CSR particle structures on the device.
//
// Reference: particle_structs/src/scs/{SellCSigma.h,SCS_sort.h,SCS_buildFns.h,SCS_rebuild.h},
// particle_structs/src/csr/{CSR.hpp,CSR_buildFns.hpp,CSR_rebuild.hpp}, ps_for.hpp:65-85.
//
// Construction runs on the host (setup).  rebuild() is the per-step operation and runs on the
// device end to end: histogram of new parents -> stable LSD radix sort of elements by
// (sigma-window, count) -> chunk widths + padding -> slice/slot offsets by scan -> one fused
// move kernel that writes EVERY member of a particle in one pass (the reference launches one
// gather/scatter kernel per member type).  Two 16-byte D2H reads per rebuild
// (counts, then capacity) replace the reference's half-dozen getLastValue syncs.
//
// Deviation (documented in DESIGN.md): ties between elements with EQUAL particle counts are
// ordered by ascending element id (stable sort); Kokkos' bitonic sort_by_key_thread order is a
// third-party detail and no result depends on it.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <numeric>
#include <functional>
#include "pp_internal.hpp"
#include "pp_ps_sort.hpp"   // Totals, the layout sort, scans
#include "pp_ps_build.hpp"  // host-side construction (HostLayout, member allocation, uploads)
#include "pp_ps_move.hpp"        // the two passes that move every member of every particle (+ new particles)
#include "pp_ps_inplace.hpp"     // kernels of the in-place rebuild (the reference's reshuffle)
#include "pp_ps_distribute.hpp"  // redistribute_particles, getPIDs kernels

namespace {

using pp::grid_for;
using pp::kBlock;

// ------------------------------------------------------------------ device kernels
__global__ void k_count_added(int n_new, const int* __restrict__ new_elems, int ne,
                              int* __restrict__ ppe, Totals* tot, int* __restrict__ rank_new) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int e = new_elems[i];
  if (e < 0 || e >= ne) {
    tot->invalid = 1;
    if (rank_new) rank_new[i] = -1;
    return;
  }
  const int r = atomicAdd(&ppe[e], 1);
  if (rank_new) rank_new[i] = r;  // rank inside the new row (see k_count_tiled)
}
// The same for chunk heights that are a multiple of 4: a thread covers four adjacent rows of a tile, so the
// mask leaves as 4-byte words (k_init_slots_tiled stores single bytes: 10.7 us per 10 M slots against the
// ~3 us its 10 MB cost) and the optional slot -> element table as 16-byte quads.
struct InitSlotsArgs {
  const int* ntiles_dev;
  int C, TP;
  const int *tiles, *chunk_start, *chunk_width, *r2e, *ppe;
  int ne;
  int *slot_elem, *row_cursor, *elem_slot0;
  unsigned char* new_mask;
  const int* go;
  int* zero_next;
  int zero_words;
  int* elem_rec0;
  // non-null: the tile table and row -> element are NOT read (k_layout_tables writes them in the same launch):
  // the tile's chunk by bisection in tile_off, the row's element from the sorted index
  const int* tile_off;
  const int* index;
  int nchunks, sorted;
};
// (g: global thread index; nthreads: threads of the launch or of the block range that runs this body)
__device__ __forceinline__ void init_slots_tiled4_body(const InitSlotsArgs& a, long long g, long long nthreads) {
  const int C = a.C, TP = a.TP, ne = a.ne;
  for (long long i = g; i < a.zero_words; i += nthreads) a.zero_next[i] = 0;
  const int Q = C >> 2;
  const int tile = (int)(g / Q), r = 4 * (int)(g - (long long)tile * Q);
  if (tile >= *a.ntiles_dev) return;
  int c, p0;
  int4 e;
  if (a.tile_off) {
    int lo = 0, hi = a.nchunks - 1;  // last chunk with tile_off[c] <= tile (as k_layout_tables)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.tile_off[mid] <= tile)
        lo = mid;
      else
        hi = mid - 1;
    }
    c = lo;
    p0 = (tile - a.tile_off[lo]) * TP;
    const int row = c * C + r;
    e.x = (row < ne && a.sorted) ? a.index[row] : row;
    e.y = (row + 1 < ne && a.sorted) ? a.index[row + 1] : row + 1;
    e.z = (row + 2 < ne && a.sorted) ? a.index[row + 2] : row + 2;
    e.w = (row + 3 < ne && a.sorted) ? a.index[row + 3] : row + 3;
  } else {
    c = a.tiles[2 * tile];
    p0 = a.tiles[2 * tile + 1];
    e = *(const int4*)(a.r2e + c * C + r);
  }
  const int start = a.chunk_start[c] + r, pend = min(p0 + TP, a.chunk_width[c]);
  const int c0 = e.x < ne ? a.ppe[e.x] : 0, c1 = e.y < ne ? a.ppe[e.y] : 0, c2 = e.z < ne ? a.ppe[e.z] : 0,
            c3 = e.w < ne ? a.ppe[e.w] : 0;
  if (p0 == 0) {
    *(int4*)(a.row_cursor + c * C + r) = make_int4(start, start + 1, start + 2, start + 3);
    if (e.x < ne) a.elem_slot0[e.x] = start;
    if (e.y < ne) a.elem_slot0[e.y] = start + 1;
    if (e.z < ne) a.elem_slot0[e.z] = start + 2;
    if (e.w < ne) a.elem_slot0[e.w] = start + 3;
    if (a.elem_rec0) {  // first record of the row when the staging records are row-major inside the chunk (pp_ps::rec_rm)
      const int w = a.chunk_width[c], q0 = pp_rec_row0(a.chunk_start[c], c, r, w, C), pt = pp_rec_pitch(w);
      if (e.x < ne) a.elem_rec0[e.x] = q0;
      if (e.y < ne) a.elem_rec0[e.y] = q0 + pt;
      if (e.z < ne) a.elem_rec0[e.z] = q0 + 2 * pt;
      if (e.w < ne) a.elem_rec0[e.w] = q0 + 3 * pt;
    }
  }
  for (int p = p0; p < pend; ++p) {
    if (a.slot_elem) *(int4*)(a.slot_elem + start + p * C) = e;
    *(unsigned*)(a.new_mask + start + p * C) =
        (p < c0 ? 1u : 0u) | (p < c1 ? 0x100u : 0u) | (p < c2 ? 0x10000u : 0u) | (p < c3 ? 0x1000000u : 0u);
  }
}
__global__ void k_init_slots_tiled4(InitSlotsArgs a) {
  if (!*a.go) return;
  init_slots_tiled4_body(a, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}
// (pp_ps::hot) Histogram and ranks of the columns [c1p, w) of the over-full row: thread = four consecutive columns,
// block = 1024 of them.  The particles that stay get consecutive ranks in column order behind ONE atomic per block
// (the next re-layout stores them as one run); the ones that leave go to a handful of neighbours: an LDS table of
// up to 16 destinations, one atomic per destination and block.
constexpr int kHotCols = 4;
__device__ __forceinline__ void count_hot_row(const pp::HotRow& hot, unsigned blk, int C, const int* __restrict__ r2e,
                                              const unsigned char* __restrict__ mask,
                                              const int* __restrict__ new_element, int ne, int* __restrict__ ppe,
                                              Totals* tot, int* __restrict__ rank) {
  __shared__ int h_key[16], h_cnt[16], h_base[16];
  __shared__ int w_stay[4], s_stay_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 16) {
    h_key[tid] = -1;
    h_cnt[tid] = 0;
  }
  __syncthreads();
  const int e = r2e[hot.chunk * C + hot.row];
  const int col0 = hot.c1p + (int)(blk * 256 + tid) * kHotCols;
  const long long slot0 = (long long)hot.start + hot.row;
  int nel[kHotCols], slot[kHotCols], loc[kHotCols];  // slot: -1 none, -2 stays, -3 own atomic (rank in loc), >= 0 table entry
  int nstay = 0;
#pragma unroll
  for (int j = 0; j < kHotCols; ++j) {
    const int p = col0 + j;
    nel[j] = -1;
    if (p < hot.w) {
      const long long pid = slot0 + (long long)p * C;
      if (mask[pid]) nel[j] = new_element[pid];
    }
  }
#pragma unroll
  for (int j = 0; j < kHotCols; ++j) {
    slot[j] = -1;
    loc[j] = -1;
    const int ne_ = nel[j];
    if (ne_ == -1) continue;
    if (ne_ < 0 || ne_ >= ne) {
      tot->invalid = 1;
      continue;
    }
    if (ne_ == e) {
      slot[j] = -2;
      loc[j] = nstay++;
      continue;
    }
    int k = 0;
    for (; k < 16; ++k) {
      int cur = h_key[k];
      if (cur == -1) cur = atomicCAS(&h_key[k], -1, ne_);
      if (cur == -1 || cur == ne_) break;
    }
    if (k < 16) {
      slot[j] = k;
      loc[j] = atomicAdd(&h_cnt[k], 1);
    } else {
      slot[j] = -3;
      loc[j] = atomicAdd(&ppe[ne_], 1);
    }
  }
  // exclusive scan of the stay counts over the block (column order)
  int inc = nstay;
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(inc, o);
    if (lane >= o) inc += y;
  }
  if (lane == 63) w_stay[wave] = inc;
  __syncthreads();
  int before = inc - nstay;
  for (int w = 0; w < wave; ++w) before += w_stay[w];
  if (tid == 0) {
    const int total = w_stay[0] + w_stay[1] + w_stay[2] + w_stay[3];
    s_stay_base = total ? atomicAdd(&ppe[e], total) : 0;
  }
  if (tid < 16 && h_key[tid] != -1) h_base[tid] = atomicAdd(&ppe[h_key[tid]], h_cnt[tid]);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kHotCols; ++j) {
    const int p = col0 + j;
    if (p >= hot.w) continue;
    int rk = -1;
    if (slot[j] == -2)
      rk = s_stay_base + before + loc[j];
    else if (slot[j] == -3)
      rk = loc[j];
    else if (slot[j] >= 0)
      rk = h_base[slot[j]] + loc[j];
    rank[slot0 + (long long)p * C] = rk;
  }
}
// ---- row-tiled histogram of new parents (SCS): thread = (old tile, row).  Particles that stay in
// their element are counted in a register and leave as ONE atomic per thread, and so do the movers
// that share one of the first three other destinations of the thread's run.  Lanes of a wave are
// different rows, so same-address contention inside a wave is gone.
template <int NK, bool UNI>
__global__ void k_count_tiled(const int* __restrict__ ntiles_dev, int C, int TP, int G,
                              const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                              const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                              const unsigned char* __restrict__ mask,
                              const int* __restrict__ new_element, int ne, int* __restrict__ ppe,
                              Totals* tot, int* __restrict__ rank, int merge, pp::HotRow hot = pp::HotRow{},
                              unsigned hot_blocks = 0u) {
  // (pp_ps::hot) the columns only the over-full row has particles in: the FIRST blocks of the grid (each is a chain
  // of round trips: at the end of the grid they would be the kernel's tail)
  if (blockIdx.x < hot_blocks) {
    count_hot_row(hot, blockIdx.x, C, r2e, mask, new_element, ne, ppe, tot, rank);
    return;
  }
  // thread = (group of G consecutive tiles, row), G*TP <= 32: consecutive tiles of one chunk are
  // the same row of the same element, so the stayers of up to 32 columns cost ONE atomic (the L2
  // atomic rate, not the 5 B/particle read, bounds this kernel).  The atomics RETURN the old count:
  // that is the particle's rank inside its new row, so the histogram pass is also the slot
  // assignment (slot = row start + rank*C once the layout is known) -- one atomic per particle per
  // rebuild instead of two.
  const long long g = (long long)(blockIdx.x - hot_blocks) * blockDim.x + threadIdx.x;
  const int grp = (int)(g / C), r = (int)(g - (long long)grp * C);
  const int ntiles = *ntiles_dev;
  int cur = -1, e = -1, start = 0, run_p0 = 0;
  // Destination table of the current run (up to 32 columns of one row): slot 0 is the row's own
  // element (the stayers), slots 1-3 the first three other destinations met -- particles of a row
  // leave into the few neighbours of its element, so most movers share a destination with another
  // mover of the same thread.  bit b of a mask = column run_p0 + b goes to that key.  A particle
  // whose destination finds no slot issues its own atomic at once.
  // (NK keys; every index below is a constant after unrolling: the table stays in registers)
  int key[NK];
  unsigned mk_[NK];
  unsigned m0 = 0;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    key[k] = -1;
    mk_[k] = 0;
  }
  auto put_ranks = [&](unsigned m, int idx) {
    while (m) {
      const int b = __ffs(m) - 1;
      m &= m - 1;
      rank[start + (run_p0 + b) * C] = idx++;
    }
  };
  auto flush = [&]() {
    // every atomic of the run first (they are independent: up to 1 + NK returning atomics in flight instead of one
    // memory round trip after the other), then the ranks they returned
    int i0 = 0, ik[NK];
    if (m0) i0 = atomicAdd(&ppe[e], __popc(m0));
#pragma unroll
    for (int k = 0; k < NK; ++k) ik[k] = mk_[k] ? atomicAdd(&ppe[key[k]], __popc(mk_[k])) : 0;
    put_ranks(m0, i0);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      put_ranks(mk_[k], ik[k]);
      key[k] = -1;
      mk_[k] = 0;
    }
    m0 = 0;
  };
  for (int k = 0; k < G; ++k) {
    const int tile = grp * G + k;
    if (tile >= ntiles) break;
    int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
    if constexpr (UNI) {
      // (chunk height 64) a wave is the 64 rows of ONE tile group: chunk, first column and width are the same in every lane -- said
      // out loud, the compiler keeps them (and the bounds tests and half of the address arithmetic below) on the
      // scalar unit
      c = __builtin_amdgcn_readfirstlane(c);
      p0 = __builtin_amdgcn_readfirstlane(p0);
    }
    if (hot.on && c == hot.chunk && p0 >= hot.c1p) continue;  // (the hot blocks' columns: the other rows are padding there)
    if (c != cur) {
      flush();
      cur = c;
      start = chunk_start[c] + r;
      e = r2e[c * C + r];
      run_p0 = p0;
    }
    const int pend = min(p0 + TP, chunk_width[c]);
    for (int pb = p0; pb < pend; pb += 8) {  // 16 independent loads in flight, then the atomics
      int nel[8];
      unsigned char mk[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pid = start + (pb + j) * C;
        mk[j] = 0;
        nel[j] = -1;
        if (pb + j < pend) {
          mk[j] = mask[pid];
          nel[j] = new_element[pid];
        }
      }
      // Two passes over the batch (the kernel is bound by instruction issue, the scalar unit above all -- ~70 scalar
      // instructions and 14 branches per particle when every particle went through the table code).  First the
      // eight particles without a branch: who stays (a bit of m0), who leaves (a bit of `mv`), who is dead (-1).
      // Then ONLY the leavers, one per iteration of a loop whose trip count is the largest number of leavers any
      // lane of the wave has among its eight (one particle in five leaves: four or five iterations, not eight).
      unsigned mv = 0;
      bool anybad = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ne_ = nel[j];
        const bool in = pb + j < pend;
        const bool live = in && mk[j] && ne_ != -1;
        const bool bad = live && (ne_ < 0 || ne_ >= ne);
        const bool good = live && !bad;
        const bool stays = good && ne_ == e;
        m0 |= stays ? (1u << (pb + j - run_p0)) : 0u;
        mv |= (good && !stays) ? (1u << j) : 0u;
        anybad = anybad || bad;
        if (in && !good) rank[start + (pb + j) * C] = -1;
      }
      if (anybad) tot->invalid = 1;
      while (mv) {
        const int j = __ffs(mv) - 1;
        mv &= mv - 1;
        int ne_ = nel[0];  // nel[j] by selects (a dynamic index would put the batch into scratch)
#pragma unroll
        for (int q = 1; q < 8; ++q) ne_ = j == q ? nel[q] : ne_;
        const unsigned bit = 1u << (pb + j - run_p0);
        bool placed = false;  // (the keys fill in order: the first matching or free one ends the search)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const bool here = !placed && (key[k] == ne_ || key[k] < 0);
          key[k] = here ? ne_ : key[k];
          mk_[k] |= here ? bit : 0u;
          placed = placed || here;
        }
        if (!placed) rank[start + (pb + j) * C] = atomicAdd(&ppe[ne_], 1);  // no room in the table: its own atomic
      }
    }
  }
  // The stayers of the thread's LAST run, merged over the block: the four waves of a block are consecutive tile
  // groups -- mostly of one chunk, i.e. the same 64 rows = the same 64 counters.  Waves whose last run lies in the
  // same chunk add their stayer counts up in LDS (the LDS atomic hands each its place inside the block's range) and
  // the first of them issues ONE returning atomic per row for all: up to four times fewer of the atomics whose rate
  // bounds this kernel.  (Runs that end inside the loop -- a group that spans chunks -- flush as before.)
  __shared__ int s_last[4];
  __shared__ int s_sum[4][64];
  const int wv = threadIdx.x >> 6;
  const bool blockwise = merge && C == 64 && blockDim.x == 256;  // (lane == row)
  if (blockwise) {
    if ((threadIdx.x & 63) == 0) s_last[wv] = cur;  // -1: the wave had no tile
    s_sum[wv][threadIdx.x & 63] = 0;
    __syncthreads();
    int lead = wv;
    for (int w = wv - 1; w >= 0; --w)
      if (s_last[w] == cur) lead = w;
    const int n0 = __popc(m0);
    int loc = 0;
    if (cur >= 0 && n0) loc = atomicAdd(&s_sum[lead][r], n0);
    __syncthreads();
    if (cur >= 0 && lead == wv) {  // (wave-uniform) this wave's `e` is the element of row r of that chunk
      const int t = s_sum[wv][r];
      s_sum[wv][r] = t ? atomicAdd(&ppe[e], t) : 0;
    }
    __syncthreads();
    if (n0) {
      int idx = s_sum[lead][r] + loc;
      unsigned m = m0;
      while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        rank[start + (run_p0 + b) * C] = idx++;
      }
    }
    m0 = 0;
  }
  flush();
}
// speculative rebuild tail: may it run against buffers sized for (cap_lim, nsl_lim)?
__global__ void k_spec_check_csr(Totals* tot, int expected) {
  tot->go = (!tot->invalid && tot->active == expected) ? 1 : 0;
}
struct SpecArgs {
  int on, cap_lim, nsl_lim, C_max, keep_if_fits;
  const int* search_nf = nullptr;  // not_found counter of the last pp_push_search: travels to the host with the totals
};
__device__ __forceinline__ void spec_decide(Totals* tot, int cap_lim, int nsl_lim, int C_max, int key_bits,
                                            int keep_if_fits) {
  if (keep_if_fits && tot->n_over == 0) {  // the reference keeps the layout here: no re-layout tail
    tot->go = 0;
    return;
  }
  tot->go = (!tot->invalid && !tot->sort_bad && tot->active > 0 && tot->nonempty >= C_max && tot->capacity <= cap_lim &&
             tot->nslices <= nsl_lim && (key_bits >= 64 || (tot->max_key >> key_bits) == 0))
                ? 1
                : 0;
}
__global__ void k_spec_check(Totals* tot, int cap_lim, int nsl_lim, int C_max, int key_bits, int keep_if_fits) {
  spec_decide(tot, cap_lim, nsl_lim, C_max, key_bits, keep_if_fits);
}
// live particles and non-empty elements of the new population, from the histogram: one atomic per
// wave of ELEMENTS (a per-wave atomic on one counter in the particle-sized kernels serialises at
// ~10 ns each and used to cost more than the histogram itself)
__global__ void k_nonempty(int ne, const int* __restrict__ ppe, Totals* tot) {
  __shared__ int s_nz[4], s_sum[4];
  int nz = 0, sum = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += gridDim.x * blockDim.x) {
    const int n = ppe[i];  // grid-stride: at most 256 blocks touch the two counters
    nz += n > 0;
    sum += n;
  }
  for (int o = 32; o > 0; o >>= 1) {
    nz += __shfl_down(nz, o);
    sum += __shfl_down(sum, o);
  }
  if ((threadIdx.x & 63) == 0) {
    s_nz[threadIdx.x >> 6] = nz;
    s_sum[threadIdx.x >> 6] = sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    nz = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
    sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    if (nz) {
      atomicAdd(&tot->nonempty, nz);
      atomicAdd(&tot->active, sum);
    }
  }
}
// SellCSigma::reshuffle's test (SCS_rebuild.h:33-42) on the histogram the full re-layout needs anyway:
// does every element's new count fit the width of the chunk its row lives in TODAY?  tot->n_over
// counts the rows that do not (0 <=> the reference would keep the layout).
__global__ void k_fit_check(int ne, int C, const int* __restrict__ ppe, const int* __restrict__ e2r,
                            const int* __restrict__ chunk_width, Totals* tot) {
  int over = 0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x)
    over += ppe[e] > chunk_width[e2r[e] / C];
  for (int o = 32; o > 0; o >>= 1) over += __shfl_down(over, o);
  if ((threadIdx.x & 63) == 0 && over) atomicAdd(&tot->n_over, over);
}
// single-block reduction of the chunk widths (sum, #non-zero) -- replaces one atomic per chunk
__global__ void k_reduce_widths(int nchunks, const int* __restrict__ widths, Totals* tot) {
  __shared__ int ssum[16], scnt[16];
  int s = 0, c = 0;
  for (int i = threadIdx.x; i < nchunks; i += blockDim.x) {
    s += widths[i];
    c += widths[i] > 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_down(s, o);
    c += __shfl_down(c, o);
  }
  if ((threadIdx.x & 63) == 0) {
    ssum[threadIdx.x >> 6] = s;
    scnt[threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int S = 0, Cn = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
      S += ssum[w];
      Cn += scnt[w];
    }
    tot->cw_sum = S;
    tot->cw_cnt = Cn;
  }
}
// One block does the whole O(nchunks) part of the layout that used to be seven launches:
// width reduction (cw_sum, cw_cnt), padding (EVENLY / PROPORTIONALLY / none; INVERSELY needs the
// ordered fp sum and keeps the separate kernels), slices / slots / tiles per chunk and their three
// exclusive scans (slice_off, chunk_start, tile_off) with totals.
__global__ void __launch_bounds__(1024)
    k_layout_fused(int nchunks, int C, int V, int TP, int pad_strat, double pad,
                   int* __restrict__ widths, int* __restrict__ slice_off,
                   int* __restrict__ chunk_start, int* __restrict__ tile_off, Totals* tot,
                   int* __restrict__ ntiles_out, SpecArgs sp, int key_bits,
                   const int* __restrict__ partial, int npartial,
                   const unsigned long long* keys_sorted = nullptr, int ne = 0,
                   unsigned long long* fix_keys = nullptr, int* fix_vals = nullptr,
                   const int* __restrict__ wide_hist = nullptr, pp::GyroRide ride = pp::GyroRide{},
                   Totals* host_out = nullptr, int host_stamp = 0,
                   const int* __restrict__ wide_tail_start = nullptr, int wide_ndig = kWideDigits) {
  if (blockIdx.x > 0) {  // gyroScatter's second stage riding along (pp::GyroRide; its first stage rode k_make_keys)
    pp::gyro_gather_body((blockIdx.x - 1) * 1024 + threadIdx.x, ride.nverts, ride.gppr, ride.off, ride.src, ride.ring,
                         ride.out, ride.out2, (ride.gppr & (ride.gppr - 1)) == 0);  // (the ride is the count-based scatter)
    return;
  }
  if (fix_keys) wide_fix_tail(ne, npartial, wide_hist, fix_keys, fix_vals, tot, wide_tail_start, wide_ndig);
  __shared__ int ssum[16], scnt[16];
  __shared__ int w3[16][3];
  __shared__ int carry[3];
  __shared__ int cw_sum_s, cw_cnt_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (partial) {  // totals of the new population: the per-block sums k_make_keys left (ElemTotalsArgs::partial)
    int p0 = 0, p1 = 0, p2 = 0;
    for (int i = t; i < npartial; i += 1024) {
      p0 += partial[3 * i];
      p1 += partial[3 * i + 1];
      p2 += partial[3 * i + 2];
    }
    for (int o = 32; o > 0; o >>= 1) {
      p0 += __shfl_down(p0, o);
      p1 += __shfl_down(p1, o);
      p2 += __shfl_down(p2, o);
    }
    if (lane == 0) {
      w3[wave][0] = p0;
      w3[wave][1] = p1;
      w3[wave][2] = p2;
    }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < 16; ++w) {
        p0 += w3[w][0];
        p1 += w3[w][1];
        p2 += w3[w][2];
      }
      tot->nonempty += p0;
      tot->active += p1;
      tot->n_over += p2;
    }
    __syncthreads();
  }
  {  // ---- reduction of the unpadded widths
    int s = 0, c = 0;
    // (four strided loads in flight per thread: at 15 625 chunks -- 1 M elements -- a thread has 15 of them,
    // one cache line each, and issued one by one they are 15 memory round trips of the single block)
    for (int i0 = t; i0 < nchunks; i0 += 4 * 1024) {
      int w4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 1024;
        // keys_sorted: one sort window, keys = counts in ascending order -- a chunk's widest row is its last
        // (k_chunk_widths2 folded in: one launch less)
        w4[k] = i < nchunks ? (keys_sorted ? (int)keys_sorted[min(i * C + C - 1, ne - 1)] : widths[i]) : 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 1024;
        if (i < nchunks) {
          if (keys_sorted) widths[i] = w4[k];
          s += w4[k];
          c += w4[k] > 0;
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_down(s, o);
      c += __shfl_down(c, o);
    }
    if (lane == 0) {
      ssum[wave] = s;
      scnt[wave] = c;
    }
    __syncthreads();
    if (t == 0) {
      int S = 0, Cn = 0;
      for (int w = 0; w < 16; ++w) {
        S += ssum[w];
        Cn += scnt[w];
      }
      cw_sum_s = S;
      cw_cnt_s = Cn;
      tot->cw_sum = S;
      tot->cw_cnt = Cn;
      carry[0] = carry[1] = carry[2] = 0;
    }
    __syncthreads();
  }
  const int cw_sum = cw_sum_s, cw_cnt = cw_cnt_s;
  const int avg_pad = (pad > 0 && cw_sum > 0 && pad_strat == PP_PAD_EVENLY) ? (int)(cw_sum * pad / cw_cnt) : 0;
  constexpr int ITEMS = 4;
  for (int base = 0; base < nchunks; base += 1024 * ITEMS) {
    int v[ITEMS][3];
    int s0 = 0, s1 = 0, s2 = 0;
    for (int k = 0; k < ITEMS; ++k) {
      const int i = base + t * ITEMS + k;
      v[k][0] = v[k][1] = v[k][2] = 0;
      if (i < nchunks) {
        int w = widths[i];
        if (pad > 0 && cw_sum > 0) {
          if (pad_strat == PP_PAD_EVENLY) {
            if (w > 0) w += avg_pad;
          } else {
            w = (int)(w + w * pad);
          }
          widths[i] = w;
        }
        v[k][0] = w / V + ((w % V) != 0);
        v[k][1] = w * C;
        v[k][2] = (w + TP - 1) / TP;
      }
      s0 += v[k][0];
      s1 += v[k][1];
      s2 += v[k][2];
    }
    int i0 = s0, i1 = s1, i2 = s2;
    for (int o = 1; o < 64; o <<= 1) {
      const int y0 = __shfl_up(i0, o), y1 = __shfl_up(i1, o), y2 = __shfl_up(i2, o);
      if (lane >= o) {
        i0 += y0;
        i1 += y1;
        i2 += y2;
      }
    }
    if (lane == 63) {
      w3[wave][0] = i0;
      w3[wave][1] = i1;
      w3[wave][2] = i2;
    }
    __syncthreads();
    int o0 = carry[0], o1 = carry[1], o2 = carry[2];
    for (int w = 0; w < wave; ++w) {
      o0 += w3[w][0];
      o1 += w3[w][1];
      o2 += w3[w][2];
    }
    int r0 = o0 + i0 - s0, r1 = o1 + i1 - s1, r2 = o2 + i2 - s2;
    for (int k = 0; k < ITEMS; ++k) {
      const int i = base + t * ITEMS + k;
      if (i < nchunks) {
        slice_off[i] = r0;
        chunk_start[i] = r1;
        tile_off[i] = r2;
      }
      r0 += v[k][0];
      r1 += v[k][1];
      r2 += v[k][2];
    }
    __syncthreads();
    if (t == 1023) {
      carry[0] = r0;
      carry[1] = r1;
      carry[2] = r2;
    }
    __syncthreads();
  }
  if (t == 0) {
    tot->nslices = carry[0];
    tot->capacity = carry[1];
    *ntiles_out = carry[2];
    // (pp_ps::hot: with one sort window the keys are the counts in ascending order)
    tot->second_key1 = (keys_sorted && ne >= 2 && nchunks >= 1 && keys_sorted[ne - 2] < (1ull << 30)) ? (int)keys_sorted[ne - 2] + 1 : 0;
    tot->last_chunk_start = nchunks >= 1 ? chunk_start[nchunks - 1] : 0;
    // the speculative tail's gate (k_spec_check) rides here: one launch less per rebuild
    if (sp.on) spec_decide(tot, sp.cap_lim, sp.nsl_lim, sp.C_max, key_bits, sp.keep_if_fits);
    // the totals, final now, straight into the host's pinned landing zone: no copy dispatch (4 us + a 6 us gap
    // before the next kernel) between this kernel and the tail
    // (host_stamp != 0: the host polls pad_[0] for it instead of waiting for an event behind this kernel -- an
    // event record is a barrier packet, 6 us before the next kernel starts)
    tot->pad_[1] = sp.search_nf ? *sp.search_nf : -1;  // (pp_ps_last_search_found)
    if (host_out) {
      Totals v = *tot;
      v.pad_[0] = 0;
      *host_out = v;
      if (host_stamp) {
        __threadfence_system();
        __hip_atomic_store(&host_out->pad_[0], host_stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
// k_layout_fused on NB blocks for structures of many chunks (BASELINE configs[3]: 10^6 elements = 15 625 chunks, where
// the single block spends 51 us on ~20 dependent memory round trips; SCS_buildFns.h:47-153 does these steps with
// parallel scans).  Block b owns the chunks [b * per, (b + 1) * per).  Two grid barriers among the NB <= 16 blocks
// (2.2 us each at 16 blocks, profiles/r02_ub_gridbar.txt; the blocks of a launch are dispatched in index order, so
// the first NB are resident before any rider block):
//   phase 1  widths off the sorted keys, per-block (sum, non-zero)                      | barrier
//   phase 2  padding from the global (sum, non-zero); LOCAL exclusive scans of slices / slots / tiles | barrier
//   phase 3  add the totals of the blocks before; the last block writes the totals (and the host's copy)
// Block 0 also adds up the per-tile totals of k_make_keys, the last block orders the overflow digit of the one-pass
// sort first (its keys are the widths of the last chunks, which are that block's).
__device__ __forceinline__ void layout_grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other XCDs' stores become visible
  }
  __syncthreads();
}
__global__ void __launch_bounds__(1024)
    k_layout_multi(int NB, int nchunks, int C, int V, int TP, int pad_strat, double pad,
                   int* __restrict__ widths, int* __restrict__ slice_off,
                   int* __restrict__ chunk_start, int* __restrict__ tile_off, Totals* tot,
                   int* __restrict__ ntiles_out, SpecArgs sp, int key_bits,
                   const int* __restrict__ partial, int npartial,
                   const unsigned long long* keys_sorted, int ne,
                   unsigned long long* fix_keys, int* fix_vals,
                   const int* __restrict__ wide_hist, pp::GyroRide ride,
                   Totals* host_out, int host_stamp,
                   const int* __restrict__ wide_tail_start, int wide_ndig) {
  if ((int)blockIdx.x >= NB) {  // gyroScatter's second stage riding along (pp::GyroRide)
    pp::gyro_gather_body((blockIdx.x - NB) * 1024 + threadIdx.x, ride.nverts, ride.gppr, ride.off, ride.src, ride.ring,
                         ride.out, ride.out2, (ride.gppr & (ride.gppr - 1)) == 0);
    return;
  }
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (nchunks + NB - 1) / NB;
  const int c0 = min(b * per, nchunks), c1 = min(c0 + per, nchunks);
  __shared__ int w3[16][3];
  __shared__ int ssum[16], scnt[16];
  __shared__ int carry[3];
  if (b == NB - 1 && fix_keys) wide_fix_tail(ne, npartial, wide_hist, fix_keys, fix_vals, tot, wide_tail_start, wide_ndig);
  if (b == 0 && partial) {  // totals of the new population: the per-tile sums k_make_keys left
    int p0 = 0, p1 = 0, p2 = 0;
    for (int i = t; i < npartial; i += 1024) {
      p0 += partial[3 * i];
      p1 += partial[3 * i + 1];
      p2 += partial[3 * i + 2];
    }
    for (int o = 32; o > 0; o >>= 1) {
      p0 += __shfl_down(p0, o);
      p1 += __shfl_down(p1, o);
      p2 += __shfl_down(p2, o);
    }
    if (lane == 0) {
      w3[wave][0] = p0;
      w3[wave][1] = p1;
      w3[wave][2] = p2;
    }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < 16; ++w) {
        p0 += w3[w][0];
        p1 += w3[w][1];
        p2 += w3[w][2];
      }
      tot->nonempty += p0;
      tot->active += p1;
      tot->n_over += p2;
    }
    __syncthreads();
  }
  {  // ---- phase 1: the unpadded widths of the own chunks and their (sum, non-zero)
    int s = 0, c = 0;
    for (int i0 = c0 + t; i0 < c1; i0 += 4 * 1024) {
      int w4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 1024;
        w4[k] = i < c1 ? (keys_sorted ? (int)keys_sorted[min(i * C + C - 1, ne - 1)] : widths[i]) : 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 1024;
        if (i < c1) {
          if (keys_sorted) widths[i] = w4[k];
          s += w4[k];
          c += w4[k] > 0;
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_down(s, o);
      c += __shfl_down(c, o);
    }
    if (lane == 0) {
      ssum[wave] = s;
      scnt[wave] = c;
    }
    __syncthreads();
    if (t == 0) {
      int S = 0, Cn = 0;
      for (int w = 0; w < 16; ++w) {
        S += ssum[w];
        Cn += scnt[w];
      }
      tot->mb[b][0] = S;
      tot->mb[b][1] = Cn;
      carry[0] = carry[1] = carry[2] = 0;
    }
  }
  layout_grid_barrier(&tot->bar, (unsigned)NB);
  int cw_sum = 0, cw_cnt = 0;
  for (int q = 0; q < NB; ++q) {  // (every thread: 2 * NB cached loads)
    cw_sum += tot->mb[q][0];
    cw_cnt += tot->mb[q][1];
  }
  const int avg_pad = (pad > 0 && cw_sum > 0 && pad_strat == PP_PAD_EVENLY) ? (int)(cw_sum * pad / cw_cnt) : 0;
  // ---- phase 2: padding, then the local exclusive scans of the own range
  constexpr int ITEMS = 4;
  for (int base = c0; base < c1; base += 1024 * ITEMS) {
    int v[ITEMS][3];
    int s0 = 0, s1 = 0, s2 = 0;
    for (int k = 0; k < ITEMS; ++k) {
      const int i = base + t * ITEMS + k;
      v[k][0] = v[k][1] = v[k][2] = 0;
      if (i < c1) {
        int w = widths[i];
        if (pad > 0 && cw_sum > 0) {
          if (pad_strat == PP_PAD_EVENLY) {
            if (w > 0) w += avg_pad;
          } else {
            w = (int)(w + w * pad);
          }
          widths[i] = w;
        }
        v[k][0] = w / V + ((w % V) != 0);
        v[k][1] = w * C;
        v[k][2] = (w + TP - 1) / TP;
      }
      s0 += v[k][0];
      s1 += v[k][1];
      s2 += v[k][2];
    }
    int i0 = s0, i1 = s1, i2 = s2;
    for (int o = 1; o < 64; o <<= 1) {
      const int y0 = __shfl_up(i0, o), y1 = __shfl_up(i1, o), y2 = __shfl_up(i2, o);
      if (lane >= o) {
        i0 += y0;
        i1 += y1;
        i2 += y2;
      }
    }
    if (lane == 63) {
      w3[wave][0] = i0;
      w3[wave][1] = i1;
      w3[wave][2] = i2;
    }
    __syncthreads();
    int o0 = carry[0], o1 = carry[1], o2 = carry[2];
    for (int w = 0; w < wave; ++w) {
      o0 += w3[w][0];
      o1 += w3[w][1];
      o2 += w3[w][2];
    }
    int r0 = o0 + i0 - s0, r1 = o1 + i1 - s1, r2 = o2 + i2 - s2;
    for (int k = 0; k < ITEMS; ++k) {
      const int i = base + t * ITEMS + k;
      if (i < c1) {
        slice_off[i] = r0;
        chunk_start[i] = r1;
        tile_off[i] = r2;
      }
      r0 += v[k][0];
      r1 += v[k][1];
      r2 += v[k][2];
    }
    __syncthreads();
    if (t == 1023) {
      carry[0] = r0;
      carry[1] = r1;
      carry[2] = r2;
    }
    __syncthreads();
  }
  if (t == 0) {
    tot->mb[b][2] = carry[0];
    tot->mb[b][3] = carry[1];
    tot->mb[b][4] = carry[2];
  }
  layout_grid_barrier(&tot->bar, 2u * (unsigned)NB);
  // ---- phase 3: the totals of the blocks before this one
  int o0 = 0, o1 = 0, o2 = 0;
  for (int q = 0; q < b; ++q) {
    o0 += tot->mb[q][2];
    o1 += tot->mb[q][3];
    o2 += tot->mb[q][4];
  }
  if (b > 0)
    for (int i = c0 + t; i < c1; i += 1024) {
      slice_off[i] += o0;
      chunk_start[i] += o1;
      tile_off[i] += o2;
    }
  if (b == NB - 1) {
    __syncthreads();  // (chunk_start[nchunks - 1] below was written by another thread of this block)
    if (t == 0) {
      tot->cw_sum = cw_sum;
      tot->cw_cnt = cw_cnt;
      tot->nslices = o0 + carry[0];
      tot->capacity = o1 + carry[1];
      *ntiles_out = o2 + carry[2];
      tot->second_key1 = (keys_sorted && ne >= 2 && nchunks >= 1 && keys_sorted[ne - 2] < (1ull << 30)) ? (int)keys_sorted[ne - 2] + 1 : 0;
      tot->last_chunk_start = nchunks >= 1 ? chunk_start[nchunks - 1] : 0;
      if (sp.on) spec_decide(tot, sp.cap_lim, sp.nsl_lim, sp.C_max, key_bits, sp.keep_if_fits);
      tot->pad_[1] = sp.search_nf ? *sp.search_nf : -1;  // (pp_ps_last_search_found)
    }
  }
  // every block is past both barriers once it arrives here: the last arrival puts the counters back to zero for the
  // next launch on these totals (the re-layout's retry with another chunk height), then the totals go to the host
  __syncthreads();
  if (t == 0) {
    const unsigned d = __hip_atomic_fetch_add(&tot->bar_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (d == (unsigned)NB - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      tot->bar = 0;
      tot->bar_done = 0;
      if (host_out) {
        Totals v = *tot;
        v.pad_[0] = 0;
        *host_out = v;
        if (host_stamp) {
          __threadfence_system();
          __hip_atomic_store(&host_out->pad_[0], host_stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
}
__global__ void k_chunk_widths2(int nchunks, int C, int ne, const unsigned long long* __restrict__ keys,
                                unsigned long long base, int sorted, const int* __restrict__ ppe,
                                int* __restrict__ widths) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= nchunks) return;
  int w = 0;
  for (int r = lane; r < C; r += 64) {
    const int row = c * C + r;
    if (row < ne) w = max(w, sorted ? (int)(keys[row] % base) : ppe[row]);
  }
  for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_down(w, o));
  if (lane == 0) widths[c] = w;
}
// The four small table fills of the new layout in ONE launch (each was a ~5 us kernel): blocks
// [0,b1) tile table, [b1,b2) slice offsets, [b2,b3) row <-> element, [b3,b4) cursors of empty chunks.
struct LayoutTablesArgs {
  unsigned b1, b2, b3;
  const int* ntiles_dev;
  int nchunks, TP, C, V, nrows, ne, sorted;
  const int *tile_off, *widths, *slice_off, *chunk_start, *index;
  int *tiles, *offsets, *s2c, *r2e, *e2r, *row_cursor, *eslot0;
  const Totals* tot;
};
__device__ __forceinline__ void layout_tables_body(const LayoutTablesArgs& a, const unsigned b) {
  if (b < a.b1) {  // tile -> (chunk, first column): owning chunk by bisection in the tile prefix
    const int t = (int)(b * blockDim.x + threadIdx.x);
    if (t >= *a.ntiles_dev) return;
    int lo = 0, hi = a.nchunks - 1;  // last chunk with tile_off[c] <= t
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.tile_off[mid] <= t)
        lo = mid;
      else
        hi = mid - 1;
    }
    a.tiles[2 * t] = lo;
    a.tiles[2 * t + 1] = (t - a.tile_off[lo]) * a.TP;
  } else if (b < a.b2) {  // slice offsets and slice -> chunk
    const int c = (int)((b - a.b1) * blockDim.x + threadIdx.x);
    if (c == 0) a.offsets[a.tot->nslices] = a.tot->capacity;
    if (c >= a.nchunks) return;
    const int w = a.widths[c];
    const int ns = w / a.V + ((w % a.V) != 0);
    const int so = a.slice_off[c], start = a.chunk_start[c];
    for (int j = 0; j < ns; ++j) {
      a.offsets[so + j] = start + j * a.V * a.C;
      a.s2c[so + j] = c;
    }
  } else if (b < a.b3) {  // row <-> element
    const int i = (int)((b - a.b2) * blockDim.x + threadIdx.x);
    if (i >= a.nrows) return;
    if (i < a.ne) {
      const int e = a.sorted ? a.index[i] : i;
      a.r2e[i] = e;
      a.e2r[e] = i;
      a.eslot0[e] = a.chunk_start[i / a.C] + i % a.C;  // first slot of the element's row
    } else {
      a.r2e[i] = i;
      a.e2r[i] = i;
    }
  } else {  // rows of zero-width chunks own no tile: give them a defined cursor
    const int i = (int)((b - a.b3) * blockDim.x + threadIdx.x);
    if (i >= a.nchunks * a.C) return;
    const int c = i / a.C;
    if (a.widths[c] == 0) a.row_cursor[i] = a.chunk_start[c] + i % a.C;
  }
}
__global__ void k_layout_tables(LayoutTablesArgs a) {
  if (!a.tot->go) return;
  layout_tables_body(a, blockIdx.x);
}
// the table fills AND the slot initialisation in one launch: the slot blocks do not read the tables (InitSlotsArgs::
// tile_off / index), so nothing orders the two ranges -- one launch and ~6 us of its latency less per re-layout
__global__ void k_layout_tables_slots(LayoutTablesArgs a, InitSlotsArgs ia, unsigned table_blocks) {
  if (!a.tot->go) return;
  if (blockIdx.x < table_blocks)
    layout_tables_body(a, blockIdx.x);
  else
    init_slots_tiled4_body(ia, (long long)(blockIdx.x - table_blocks) * blockDim.x + threadIdx.x,
                           (long long)(gridDim.x - table_blocks) * blockDim.x);
}
// printMetrics (SellCSigma.h:465-524): padded cells (slots whose mask is 0) and slices that hold at least one.  Block
// (x, y) takes segment y (kMetricsSeg bytes) of slice x; the first block of a slice that finds a padded cell counts
// the slice (flag[y], zeroed by the caller).  out[0] += cells, out[1] += slices
constexpr int kMetricsSeg = 8192;
__global__ void k_slice_padding(const int* __restrict__ offsets, const unsigned char* __restrict__ mask,
                                int* __restrict__ out, int* __restrict__ flag) {
  __shared__ int s_w[4];
  const int s = blockIdx.x;
  const int lo0 = offsets[s], hi0 = offsets[s + 1];
  const int lo = lo0 + (int)blockIdx.y * kMetricsSeg, hi = min(hi0, lo + kMetricsSeg);
  if (lo >= hi0) return;
  int n = 0;
  for (int j = lo + 4 * (int)threadIdx.x; j < hi; j += 4 * 256) {
    if (j + 3 < hi && (lo & 3) == 0) {
      const unsigned m = *(const unsigned*)(mask + j);
      n += 4 - __popc(m & 0x01010101u);
    } else {
      for (int q = j; q < min(j + 4, hi); ++q) n += !mask[q];
    }
  }
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    n = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (n) {
      atomicAdd(&out[0], n);
      if (atomicExch(&flag[s], 1) == 0) atomicAdd(&out[1], 1);
    }
  }
}
// slot -> parent element of every slot of every tile (what k_init_slots_tiled leaves out, see pp::slot_elem)
__global__ void k_fill_slot_elem(const int* __restrict__ ntiles_dev, int C, int TP, const int* __restrict__ tiles,
                                 const int* __restrict__ chunk_start, const int* __restrict__ chunk_width,
                                 const int* __restrict__ r2e, int* __restrict__ slot_elem) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tile = (int)(g / C), r = (int)(g - (long long)tile * C);
  if (tile >= *ntiles_dev) return;
  const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
  const int start = chunk_start[c] + r, pend = min(p0 + TP, chunk_width[c]);
  const int e = r2e[c * C + r];
  for (int p = p0; p < pend; ++p) slot_elem[start + p * C] = e;
}
// 64-slot group -> chunk, chunk height 64 (pp_ps_iteration): thread = tile
__global__ void k_fill_group_chunk(const int* __restrict__ ntiles_dev, int TP, const int* __restrict__ tiles,
                                   const int* __restrict__ chunk_start, const int* __restrict__ chunk_width,
                                   int* __restrict__ group_chunk) {
  const int tile = blockIdx.x * blockDim.x + threadIdx.x;
  if (tile >= *ntiles_dev) return;
  const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
  const int g0 = chunk_start[c] >> 6, pend = min(p0 + TP, chunk_width[c]);
  for (int p = p0; p < pend; ++p) group_chunk[g0 + p] = c;
}
// new layout: slot -> parent element for every slot of every tile, first slot of every row
__global__ void k_init_slots_tiled(const int* __restrict__ ntiles_dev, int C, int TP,
                                   const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                                   const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                                   const int* __restrict__ ppe, int ne,
                                   int* __restrict__ slot_elem, int* __restrict__ row_cursor,
                                   int* __restrict__ elem_slot0, unsigned char* __restrict__ new_mask,
                                   const int* __restrict__ go, int* __restrict__ zero_next = nullptr,
                                   int zero_words = 0, int* __restrict__ elem_rec0 = nullptr) {
  if (!*go) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // the histogram + totals block of the NEXT rebuild (today's d_elem_count, free once this tail runs) is
  // cleared here instead of by a fill at the start of that rebuild (a fill is a ~6 us dispatch)
  for (long long i = g; i < zero_words; i += (long long)gridDim.x * blockDim.x) zero_next[i] = 0;
  const int tile = (int)(g / C), r = (int)(g - (long long)tile * C);
  if (tile >= *ntiles_dev) return;
  const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
  const int start = chunk_start[c] + r, pend = min(p0 + TP, chunk_width[c]);
  const int e = r2e[c * C + r];
  // a row's particles take its slots in column order (the row cursor starts at column 0 and
  // advances by C), so the new mask is a function of the new per-element counts: written here as
  // coalesced runs instead of one scattered byte store per moved particle
  const int cnt = e < ne ? ppe[e] : 0;
  if (p0 == 0) {
    row_cursor[c * C + r] = start;
    if (e < ne) elem_slot0[e] = start;  // first slot of the element's new row (pack: + rank*C)
    if (e < ne && elem_rec0) elem_rec0[e] = pp_rec_row0(chunk_start[c], c, r, chunk_width[c], C);
  }
  for (int p = p0; p < pend; ++p) {
    if (slot_elem) slot_elem[start + p * C] = e;
    new_mask[start + p * C] = p < cnt ? 1 : 0;
  }
}

// CSR counting sort (CSR_rebuild.hpp:62-108)
__global__ void k_move_csr(int nold, const int* __restrict__ new_element, int* __restrict__ cursor,
                           MoveArgs a, const int* __restrict__ go) {
  if (!*go) return;
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= nold) return;
  const int e = new_element[pid];
  if (e < 0) return;  // count_existing treats every negative id as removed (CSR_rebuild.hpp:36-40)
  const int idx = atomicAdd(&cursor[e], 1);
  copy_members(a, pid, idx);
}
__global__ void k_add_csr(int n_new, const int* __restrict__ new_elems, int* __restrict__ cursor,
                          MoveArgs a, const int* __restrict__ go) {
  if (!*go) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int idx = atomicAdd(&cursor[new_elems[i]], 1);
  copy_members(a, i, idx);
}
// CSR rank/histogram pass.  Consecutive CSR slots belong to the same element, so the particles of a
// wave that STAY in their element would all hit one counter (same-address returning atomics
// serialise: 5.5 ms for 50 M particles at 1000 per element).  Up to four groups of stayers per wave
// are counted with one atomic each; everything else (movers, further groups) goes one by one.
__global__ void k_count_csr(int nold, const int* __restrict__ new_element,
                            const int* __restrict__ old_element, int ne, int* __restrict__ ppe,
                            Totals* tot, int* __restrict__ rank) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const bool in = pid < nold;
  const int e = in ? new_element[pid] : -1;
  const bool valid = e > -1 && e < ne;  // negative ids are removed (CSR_rebuild.hpp:36-40)
  if (in && e >= ne) tot->invalid = 1;
  const bool stay = valid && e == old_element[pid];
  int rk = -1;
  bool counted = false;
  unsigned long long rem = __ballot(stay);
  for (int it = 0; it < 4 && rem; ++it) {
    const int leader = __builtin_ctzll(rem);
    const int key = __shfl(e, leader);
    const bool mine = stay && e == key;
    const unsigned long long m = __ballot(mine);
    int base = 0;
    if (lane == leader) base = atomicAdd(&ppe[key], __popcll(m));
    base = __shfl(base, leader);
    if (mine) {
      rk = base + __popcll(m & lt_mask);  // rank inside the new element: slot = offsets[e] + rank
      counted = true;
    }
    rem &= ~m;
  }
  if (valid && !counted) rk = atomicAdd(&ppe[e], 1);
  if (in) rank[pid] = rk;
}
// slot -> element and mask of a CSR from its offsets: a thread owns 8 slots 64 apart (coalesced
// stores), finds the element of its first slot by bisection and then walks the offsets forward.
// (One wave per element row, the obvious form, runs at 0.25 TB/s when rows are a few hundred slots.)
__global__ void k_csr_slots(int ne, const int* __restrict__ offsets, int capacity,
                            int* __restrict__ slot_elem, unsigned char* __restrict__ mask,
                            const int* __restrict__ go) {
  if (go && !*go) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long first = (g >> 6) * 512 + (g & 63);
  if (first >= capacity) return;
  const int total = offsets[ne];
  int lo = 0, hi = ne;  // largest e with offsets[e] <= first (e == ne: tail)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (offsets[mid] <= first) lo = mid; else hi = mid - 1;
  }
  int e = lo;
  for (int k = 0; k < 8; ++k) {
    const long long j = first + 64 * k;
    if (j >= capacity) break;
    if (j >= total) {
      slot_elem[j] = -1;
      mask[j] = 0;
      continue;
    }
    // largest e with offsets[e] <= j: gallop + bisection.  (A plain `while (offsets[e+1] <= j) ++e` walks every
    // empty element in between: 64 slots apart are ~60 000 elements apart in the tails of a gaussian population
    // at one particle per element -- 10 ms for this kernel, ps_combo160 1 M / 1 M dist 2.)
    if (offsets[e + 1] <= j) {
      int lo = e, step = 1;
      while (lo + step <= ne && offsets[lo + step] <= j) {
        lo += step;
        step <<= 1;
      }
      int hi = min(lo + step, ne);
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= j) lo = mid; else hi = mid;
      }
      e = lo;
    }
    slot_elem[j] = e;
    mask[j] = 1;
  }
}

int finish_layout_upload(pp_ps* ps, const HostLayout& L, const std::vector<int>& ppe) {
  // slot_elem + mask image on host
  std::vector<int> slot_elem((size_t)L.capacity, -1);
  std::vector<unsigned char> mask((size_t)L.capacity, 0);
  for (int c = 0; c < L.nchunks; ++c)
    for (int r = 0; r < L.C; ++r) {
      const int row = c * L.C + r;
      const int e = L.row_to_element[row];
      for (int p = 0; p < L.chunk_widths[c]; ++p) {
        const int pid = L.chunk_start[c] + r + p * L.C;
        slot_elem[pid] = e;
        mask[pid] = (e < ps->num_elems && ps->num_ptcls > 0) ? (p < L.ptcls[row]) : 0;
      }
    }
  (void)ppe;
  // row tiles (chunk, first p) of kTileP columns for the row-major hot kernels
  std::vector<int> tiles;
  for (int c = 0; c < L.nchunks; ++c)
    for (int p0 = 0; p0 < L.chunk_widths[c]; p0 += ps->tile_p) {
      tiles.push_back(c);
      tiles.push_back(p0);
    }
  const int ntiles = (int)(tiles.size() / 2);
  ps->ntiles_max = ntiles;
  std::vector<int> ntl(1, ntiles);
  int rc;
  std::vector<int> eslot0((size_t)std::max(ps->num_elems, 1), 0);
  for (int e = 0; e < ps->num_elems; ++e) {
    const int row = L.element_to_row[e];
    eslot0[e] = L.chunk_start[row / L.C] + row % L.C;
  }
  if ((rc = upload_vec(ps->d_eslot0, eslot0))) return rc;
  if ((rc = upload_vec(ps->d_tiles, tiles))) return rc;
  if ((rc = upload_vec(ps->d_ntiles, ntl))) return rc;
  if ((rc = upload_vec(ps->d_chunk_start, L.chunk_start))) return rc;
  if ((rc = upload_vec(ps->d_chunk_width, L.chunk_widths))) return rc;
  if ((rc = upload_vec(ps->d_offsets, L.offsets))) return rc;
  if ((rc = upload_vec(ps->d_slice_to_chunk, L.slice_to_chunk))) return rc;
  if ((rc = upload_vec(ps->d_row_to_element, L.row_to_element))) return rc;
  if ((rc = upload_vec(ps->d_element_to_row, L.element_to_row))) return rc;
  if ((rc = upload_vec(ps->d_mask, mask))) return rc;
  if ((rc = upload_vec(ps->d_slot_elem, slot_elem))) return rc;
  ps->slot_elem_valid = true;
  ps->group_chunk_valid = false;
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  return PP_OK;
}

MoveArgs make_move(const pp_ps* ps, const std::vector<pp::DevBuf>& src, int64_t src_stride,
                   std::vector<pp::DevBuf>& dst, int64_t dst_stride) {
  MoveArgs a{};
  a.nmembers = ps->nmembers;
  for (int m = 0; m < ps->nmembers; ++m) {
    a.src[m] = src[m].p;
    a.dst[m] = dst[m].p;
    a.bytes[m] = ps->member_bytes[m];
    a.ncomp[m] = ps->member_ncomp[m];
  }
  a.src_stride = src_stride;
  a.dst_stride = dst_stride;
  a.commit_x = a.commit_xt = -1;
  return a;
}

// One attempt of the device re-layout for a given chunk height.  Enqueues everything up to (and
// including) the D2H read of the totals; the caller synchronises once.
struct LayoutPlan {
  int C, nchunks, nrows;
  int key_bits;  // key bits the radix passes of this attempt covered (64 = every bit)
  bool sorted;
  bool wide;  // sorted by k_rs_pass_wide: the layout kernel orders the overflow digit
  bool totals_on_host;  // the layout kernel wrote the totals to the pinned landing zone itself
  const int* wide_tail_start;  // one-pass sort over many tiles: first output position of the overflow digit
  int wide_ndig = kWideDigits;  // digits of the one pass (2048 / 256 / 64)
  unsigned long long base;
  unsigned long long* keys;
  int* index;
  int *widths, *nsl, *nslots, *slice_off, *tile_cnt, *tile_off, *chunk_start;
};

// bits_limit > 0: sort on the low `bits_limit` key bits only (the caller predicts the largest key
// from the previous rebuild and checks the prediction against Totals::max_key afterwards)
int enqueue_layout(pp_ps* ps, int C_new, int* ppe, Totals* tot, long long key_base, LayoutPlan& L,
                   int bits_limit = 0, ElemTotalsArgs et = ElemTotalsArgs{0, 0, 1, nullptr, nullptr, nullptr},
                   SpecArgs sp = SpecArgs{0, 0, 0, 0, 0}, const LayoutTablesArgs* tables = nullptr,
                   bool allow_wide = false, Totals* host_out = nullptr, int host_stamp = 0) {
  hipStream_t st = pp::stream();
  const int ne = ps->num_elems;
  pp::GyroRide ride{};
  L.wide = false;
  L.wide_tail_start = nullptr;
  L.totals_on_host = false;
  L.C = C_new;
  L.nchunks = ne / C_new + (ne % C_new != 0);
  L.nrows = L.nchunks * C_new;
  L.sorted = ps->sigma > 1 && ne > 1;
  L.base = (unsigned long long)key_base;
  L.key_bits = 64;
  L.keys = nullptr;
  L.index = nullptr;
  if (L.sorted) {
    const int sg = std::min(ps->sigma, std::max(ne, 1));
    const int n_sigma = ne / sg;
    PP_HIP_CHECK(ps->s_keys.reserve(sizeof(unsigned long long) * (size_t)ne));
    PP_HIP_CHECK(ps->s_keys2.reserve(sizeof(unsigned long long) * (size_t)ne));
    PP_HIP_CHECK(ps->s_vals.reserve(sizeof(int) * (size_t)ne));
    PP_HIP_CHECK(ps->s_vals2.reserve(sizeof(int) * (size_t)ne));
    const int nblk = (ne + RS_TILE - 1) / RS_TILE;
    const int nseg = (nblk + kWideSeg - 1) / kWideSeg;
    const size_t hist_words = std::max(256 * (size_t)nblk * 3, ((size_t)nblk + nseg + 1) * kWideDigits);
    PP_HIP_CHECK(ps->s_hist.reserve(sizeof(int) * (hist_words + 3 * (size_t)nblk)));
    // the layout kernel below adds up the per-block totals (the inversely-padded layout has no such kernel)
    int* const partial_at = ps->s_hist.as<int>() + hist_words;
    et.partial = (et.totals && !(ps->pad_strat == PP_PAD_INVERSELY && ps->shuffle_padding > 0)) ? partial_at : nullptr;
    const bool fused_sort = nblk <= kFusedSortBlocks;
    unsigned long long maxkey = (unsigned long long)(n_sigma > 0 ? n_sigma : 1) * L.base;
    int bits = 0;
    while (bits < 64 && (maxkey >> bits)) ++bits;
    L.key_bits = 64;
    if (bits_limit > 0 && bits_limit < bits) {
      bits = bits_limit;
      L.key_bits = (bits + 7) / 8 * 8;
    }
    // one sort window (the keys are the counts): ONE counting pass, see k_rs_pass_wide
    // (up to 64 tiles every block sweeps the digit table itself; beyond that k_wide_seg / k_wide_base prefix it)
    const bool wide_sort = allow_wide && n_sigma <= 1 && ps->wide_skip == 0 &&
                           (ps->pad_strat != PP_PAD_INVERSELY || !(ps->shuffle_padding > 0));  // (needs k_layout_fused)
    // digits of the one pass, from the largest count of the previous rebuild (unknown: 2048).  A prediction that
    // fails shows as Totals::sort_bad (more than 1 024 keys in the overflow digit): the caller re-sorts with every
    // 8-bit pass and pp_ps::narrow_skip keeps the next rebuilds on 2048 digits.
    int wide_ndig = kWideDigits;
    if (wide_sort && ps->narrow_skip == 0 && ps->last_max_key != ~0ull) {
      if (ps->last_max_key <= 44) wide_ndig = 64;
      else if (ps->last_max_key <= 200) wide_ndig = 256;
    }
    static const int force_ndig = PP_LAB_ENV("PP_WIDE_NDIG") ? atoi(PP_LAB_ENV("PP_WIDE_NDIG")) : 0;  // (lab build: A/B)
    if (wide_sort && (force_ndig == 64 || force_ndig == 256 || force_ndig == 2048)) wide_ndig = force_ndig;
    const bool narrow = wide_sort && wide_ndig < kWideDigits;
    const bool wide_big = wide_sort && !fused_sort && !narrow;
    L.wide = wide_sort;
    L.wide_ndig = wide_ndig;
    L.wide_tail_start = nullptr;
    if (wide_sort) L.key_bits = 64;  // (its own check: Totals::sort_bad)
    int* const H0 = ps->s_hist.as<int>();
    // gyroScatter of a pp_ps_rebuild_scatter call rides in this launch and in the layout kernel's (pp::GyroRide)
    const bool ride_ok = ps->ride && ps->ride->on && !ps->ride_done &&
                         (ps->pad_strat != PP_PAD_INVERSELY || !(ps->shuffle_padding > 0));
    if (ride_ok) ride = *ps->ride;
    k_make_keys<<<nblk + (ride.on ? grid_for(ride.nverts) : 0), 256, 0, st>>>(ne, ppe, sg, n_sigma, L.base,
                                                 ps->s_keys.as<unsigned long long>(),
                                                 ps->s_vals.as<int>(), tot,
                                                 false, et,
                                                 (fused_sort || wide_sort) ? FusedHist{H0, nblk, wide_sort ? 1 : 0, wide_ndig}
                                                                           : FusedHist{nullptr, 0, 0},
                                                 nblk, ride);
    unsigned long long *ka = ps->s_keys.as<unsigned long long>(),
                       *kb = ps->s_keys2.as<unsigned long long>();
    int *va = ps->s_vals.as<int>(), *vb = ps->s_vals2.as<int>();
    int* hist = ps->s_hist.as<int>();
    int* hist_sc = hist + 256 * nblk;
    if (wide_sort) {
      int *seg_base = nullptr, *digit_base = nullptr;
      if (wide_big) {
        seg_base = H0 + (size_t)nblk * kWideDigits;
        k_wide_seg<<<dim3(kWideDigits / 256, nseg), 256, 0, st>>>(nblk, H0, seg_base);
        digit_base = seg_base + (size_t)nseg * kWideDigits;
        k_wide_base<<<1, 1024, 0, st>>>(nseg, seg_base, digit_base);
        L.wide_tail_start = digit_base + kWideDigits - 1;  // first output position of the overflow digit
      }
      if (wide_ndig == 64)
        k_rs_pass_wide<64><<<nblk, kWideThreads, 0, st>>>(ne, ka, va, nblk, H0, kb, vb, nullptr, nullptr);
      else if (wide_ndig == 256)
        k_rs_pass_wide<256><<<nblk, kWideThreads, 0, st>>>(ne, ka, va, nblk, H0, kb, vb, nullptr, nullptr);
      else
        k_rs_pass_wide<kWideDigits><<<nblk, kWideThreads, 0, st>>>(ne, ka, va, nblk, H0, kb, vb, seg_base, digit_base);
      std::swap(ka, kb);
      std::swap(va, vb);
    }
    for (int shift = 0; fused_sort && !wide_sort && shift < bits; shift += 8) {
      if (shift > 0) k_rs_hist<<<nblk, 256, 0, st>>>(ne, ka, shift, nblk, H0, tot);  // (pass 0: k_make_keys)
      k_rs_pass<<<nblk, 256, 0, st>>>(ne, ka, va, shift, nblk, H0, kb, vb, tot);
      std::swap(ka, kb);
      std::swap(va, vb);
    }
    for (int shift = 0; !fused_sort && !wide_sort && shift < bits; shift += 8) {
      k_rs_hist<<<nblk, 256, 0, st>>>(ne, ka, shift, nblk, hist, tot);
      if (scan_excl(ps->s_scan2, 256 * nblk, hist, hist_sc, nullptr, st, tot, shift)) return PP_EHIP;
      k_rs_scatter<<<nblk, 256, 0, st>>>(ne, ka, va, shift, nblk, hist_sc, kb, vb, tot);
      std::swap(ka, kb);
      std::swap(va, vb);
    }
    L.keys = ka;
    L.index = va;
  }
  const int nchunks = L.nchunks;
  PP_HIP_CHECK(ps->s_chunkw.reserve(sizeof(int) * (size_t)nchunks * 5 + 64));
  PP_HIP_CHECK(ps->s_cwidth2.reserve(sizeof(int) * (size_t)nchunks));
  PP_HIP_CHECK(ps->s_cstart2.reserve(sizeof(int) * (size_t)nchunks));
  L.widths = ps->s_cwidth2.as<int>();
  L.nsl = ps->s_chunkw.as<int>();
  L.nslots = L.nsl + nchunks;
  L.slice_off = L.nslots + nchunks;
  L.tile_cnt = L.slice_off + nchunks;
  L.tile_off = L.tile_cnt + nchunks;
  L.chunk_start = ps->s_cstart2.as<int>();
  const bool fused_layout = ps->pad_strat != PP_PAD_INVERSELY || !(ps->shuffle_padding > 0);
  // one sort window: the keys are the counts, ascending -- the layout kernel reads a chunk's width off its last row
  // (after the one-pass sort it has to: the overflow digit is ordered by that kernel's prologue)
  const bool widths_in_layout = fused_layout && L.sorted && ne > 0 && ne / std::min(ps->sigma, std::max(ne, 1)) <= 1;
  if (!widths_in_layout)
    k_chunk_widths2<<<grid_for((size_t)nchunks * 64), kBlock, 0, st>>>(
        nchunks, C_new, ne, L.keys, L.base, L.sorted ? 1 : 0, ppe, L.widths);
  PP_HIP_CHECK(ps->s_scan.reserve(sizeof(int)));
  // many chunks (10^6 elements): the same work on up to 16 blocks with two grid barriers (k_layout_multi)
  const int layout_blocks = nchunks >= 4096 ? std::min(kLayoutBlocksMax, (nchunks + 1023) / 1024) : 1;
  if (fused_layout && layout_blocks > 1) {
    k_layout_multi<<<layout_blocks + (ride.on ? (unsigned)(((size_t)ride.nverts * 16 + 1023) / 1024) : 0), 1024, 0, st>>>(
                                       layout_blocks, nchunks, C_new, ps->V, ps->tile_p, ps->pad_strat,
                                       ps->shuffle_padding, L.widths, L.slice_off, L.chunk_start,
                                       L.tile_off, tot, ps->s_scan.as<int>(), sp, L.key_bits,
                                       L.sorted ? et.partial : nullptr, (ne + RS_TILE - 1) / RS_TILE,
                                       widths_in_layout ? L.keys : nullptr, ne, L.wide ? L.keys : nullptr,
                                       L.wide ? L.index : nullptr, L.wide ? ps->s_hist.as<int>() : nullptr, ride,
                                       host_out, host_stamp, L.wide ? L.wide_tail_start : nullptr, L.wide_ndig);
    if (ride.on) ps->ride_done = true;
    L.totals_on_host = host_out != nullptr;
  } else if (fused_layout) {
    k_layout_fused<<<1 + (ride.on ? (unsigned)(((size_t)ride.nverts * 16 + 1023) / 1024) : 0), 1024, 0, st>>>(
                                       nchunks, C_new, ps->V, ps->tile_p, ps->pad_strat,
                                       ps->shuffle_padding, L.widths, L.slice_off, L.chunk_start,
                                       L.tile_off, tot, ps->s_scan.as<int>(), sp, L.key_bits,
                                       L.sorted ? et.partial : nullptr, (ne + RS_TILE - 1) / RS_TILE,
                                       widths_in_layout ? L.keys : nullptr, ne, L.wide ? L.keys : nullptr,
                                       L.wide ? L.index : nullptr, L.wide ? ps->s_hist.as<int>() : nullptr, ride,
                                       host_out, host_stamp, L.wide ? L.wide_tail_start : nullptr, L.wide_ndig);
    if (ride.on) ps->ride_done = true;
    L.totals_on_host = host_out != nullptr;
  } else {
    k_reduce_widths<<<1, 1024, 0, st>>>(nchunks, L.widths, tot);
    k_cw_inv_serial<<<1, 64, 0, st>>>(nchunks, L.widths, tot);
    k_apply_padding<<<grid_for(nchunks), kBlock, 0, st>>>(nchunks, ps->pad_strat,
                                                          ps->shuffle_padding, L.widths, tot);
    k_slices_and_slots<<<grid_for(nchunks), kBlock, 0, st>>>(nchunks, C_new, ps->V, L.widths, L.nsl,
                                                             L.nslots);
    k_scan_excl<<<1, 1024, 0, st>>>(nchunks, L.nsl, L.slice_off, &tot->nslices);
    k_scan_excl<<<1, 1024, 0, st>>>(nchunks, L.nslots, L.chunk_start, &tot->capacity);
    k_tile_count<<<grid_for(nchunks), kBlock, 0, st>>>(nchunks, ps->tile_p, L.widths, L.tile_cnt);
    k_scan_excl<<<1, 1024, 0, st>>>(nchunks, L.tile_cnt, L.tile_off, ps->s_scan.as<int>());
    if (sp.on) k_spec_check<<<1, 1, 0, st>>>(tot, sp.cap_lim, sp.nsl_lim, sp.C_max, L.key_bits, sp.keep_if_fits);
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// Entry table of the staged move: sources are member arrays `src[m]` with component stride
// `src_stride` (elements), destinations the swap buffers.  Returns the number of 16-B quads per
// record, or 0 when the members do not fit the staged path (other sizes than 4/8 bytes, too many).
// (lab build, PP_NO_LAZY_UNPACK=1: the second pass of every re-layout runs right away)
// Split 2-D records -- (x, y) and (pad, phi, b, id) in two arrays, the 2-D push reads the second only -- are OFF by
// default: the push gains 45 us per step at 10 M particles and the pack, storing two streams of 16-B pieces, loses 35 to
// 65 depending on the box (profiles/r06_ab_split_records*.txt: four sessions, one of which had the split form ahead).
// Lab build, PP_REC_SPLIT=1: on (the state-machine and record tests run both ways, tools/gpu_test_matrix.sh).
bool no_rec_split() {
  static const bool v = PP_LAB_ENV("PP_REC_SPLIT") == nullptr || atoi(PP_LAB_ENV("PP_REC_SPLIT")) == 0;
  return v;
}
bool no_lazy_unpack() {
  static const bool off = PP_LAB_ENV("PP_NO_LAZY_UNPACK") != nullptr;
  return off;
}
// `side_member` >= 0: that member (one 4-byte component) travels beside the record (WordTable::side_src / side_dst).
// `zero_z`: component 2 of member commit_xt's arrays is logically zero (pp_ps::zero_z_pending): the word that
// would read it gets a null source -- the pack stores 0.
// `drop_z` (with zero_z): that component does not travel at all -- its destination joins the zero list (pass 2 writes
// the zeros) and the record is one word shorter (the split 2-D record: x, y | pad, phi, b, id).
int build_word_table(const pp_ps* ps, const void* const* src, int64_t src_stride, int64_t dst_stride,
                     int commit_x, int commit_xt, WordTable& wt, int side_member = -1, bool zero_z = false,
                     bool drop_z = false) {
  wt = WordTable{};
  for (int m = 0; m < ps->nmembers; ++m) {
    const int b = ps->member_bytes[m];
    if (b != 4 && b != 8) return 0;
    for (int cc = 0; cc < ps->member_ncomp[m]; ++cc) {
      char* dst = (char*)ps->swap[m].p + ((size_t)cc * dst_stride) * b;
      if (m == side_member) {
        if (b != 4 || ps->member_ncomp[m] != 1 || m == commit_x || m == commit_xt) return 0;
        wt.side_src = (const char*)src[m];
        wt.side_dst = dst;
        continue;
      }
      if (m == commit_xt) {  // constant 0 after the fused updatePtclPositions
        int& nz = (b == 8) ? wt.nz8 : wt.nz4;
        if (nz >= 8) return 0;
        ((b == 8) ? wt.z8 : wt.z4)[nz++] = dst;
        continue;
      }
      const int sm = (m == commit_x) ? commit_xt : m;  // fused updatePtclPositions
      const char* sp = (const char*)src[sm] + ((size_t)cc * src_stride) * b;
      if (zero_z && m == commit_x && cc == 2 && b == 8) {
        if (drop_z) {
          if (wt.nz8 >= 8) return 0;
          wt.z8[wt.nz8++] = dst;
          continue;
        }
        sp = nullptr;
      }
      if (b == 8) {
        if (wt.n8 >= kMax8) return 0;
        wt.src8[wt.n8] = sp;
        wt.dst8[wt.n8++] = dst;
      } else {
        if (wt.n4 >= kMax4) return 0;
        wt.src4[wt.n4] = sp;
        wt.dst4[wt.n4++] = dst;
      }
    }
  }
  const int nw = 2 * wt.n8 + wt.n4;
  // 16-B quads per record; 3 is rounded up to 4: a 48-B record straddles 64-B sectors and the
  // scattered stores of pass 1 become read-modify-writes (measured 0.25 -> 0.40 ms per 10 M)
  int NQ = (nw + 3) / 4 == 3 ? 4 : (nw + 3) / 4;
  // the same for the 160-B ps_combo160 particle: every other 160-B record starts in the middle of a 64-B sector;
  // padded to 192 B (whole sectors) the scattered stores of pass 1 are plain writes
  if (NQ == 10) NQ = 12;
  return (NQ == 4 || NQ == 10 || NQ == 12 || (NQ >= 1 && NQ <= 3) || NQ == 6 || NQ == 8) ? NQ : 0;
}

// pinned landing zone of the rebuild totals + the event the host waits on: one per STRUCTURE (two host threads
// that rebuild different structures share nothing; round-3 advisor finding)
int totals_pin(pp_ps* ps, Totals** h_pin_out, hipEvent_t* ev_out) {
  if (!ps->h_totals) {
    PP_HIP_CHECK(hipHostMalloc(&ps->h_totals, sizeof(Totals), hipHostMallocCoherent | hipHostMallocMapped));
    memset(ps->h_totals, 0, sizeof(Totals));
    PP_HIP_CHECK(hipEventCreateWithFlags((hipEvent_t*)&ps->ev_totals, hipEventDisableTiming));
  }
  *h_pin_out = (Totals*)ps->h_totals;
  *ev_out = (hipEvent_t)ps->ev_totals;
  return PP_OK;
}

// The in-place rebuild (kernels above).  Returns 1 when the rebuild is complete, 0 when the layout
// cannot be kept (some row would overflow, nothing is left, members the staged record cannot hold:
// the structure is untouched and the caller runs the full re-layout), < 0 on error.
int scs_reshuffle(pp_ps* ps, const int* new_element, int n_new, const int* new_elems,
                  const void* const* new_info, int commit_x, int commit_xt,
                  const std::function<int(const int*)>& pre_sync) {
  if (ps->shuffle_mode <= 0) return 0;
  if (!(ps->capacity > 0 && ps->num_ptcls > 0) || !ps->elem_count_valid || ps->ntiles_max <= 0) return 0;
  if (ps->d_eslot0.bytes < sizeof(int) * (size_t)std::max(ps->num_elems, 1)) return 0;
  const bool commit = commit_x >= 0 && commit_xt >= 0;
  const int ne = ps->num_elems;
  // word table: every component that travels, source == destination buffer (in place).  With the
  // fused updatePtclPositions the buffers of x and x_tgt trade places on success: x is read from /
  // written to x_tgt's buffer, x_tgt (all zero afterwards) does not travel at all.
  WordTable wt{};
  for (int m = 0; m < ps->nmembers; ++m) {
    const int b = ps->member_bytes[m];
    if (b != 4 && b != 8) return 0;
    if (commit && m == commit_xt) continue;
    const pp::DevBuf& buf = (commit && m == commit_x) ? ps->data[commit_xt] : ps->data[m];
    for (int cc = 0; cc < ps->member_ncomp[m]; ++cc) {
      char* q = (char*)buf.p + ((size_t)cc * ps->stride) * b;
      if (b == 8) {
        if (wt.n8 >= kMax8) return 0;
        wt.src8[wt.n8] = q;
        wt.dst8[wt.n8++] = q;
      } else {
        if (wt.n4 >= kMax4) return 0;
        wt.src4[wt.n4] = q;
        wt.dst4[wt.n4++] = q;
      }
    }
  }
  const int nw = 2 * wt.n8 + wt.n4;
  const int NQ = (nw + 3) / 4 == 3 ? 4 : (nw + 3) / 4;
  if (!(NQ == 1 || NQ == 2 || NQ == 4 || NQ == 6 || NQ == 8 || NQ == 10)) return 0;
  hipStream_t st = pp::stream();
  PP_HIP_CHECK(ps->s_ppe.reserve(sizeof(int) * (size_t)std::max(ne, 1)));
  PP_HIP_CHECK(ps->s_misc.reserve(sizeof(Totals)));
  PP_HIP_CHECK(ps->s_rs.reserve(sizeof(int) * 5 * (size_t)std::max(ne, 1)));
  PP_HIP_CHECK(ps->s_idx.reserve(sizeof(int) * (size_t)ps->capacity));
  PP_HIP_CHECK(ps->s_holes.reserve(sizeof(int) * (size_t)ps->capacity));
  PP_HIP_CHECK(ps->s_aos.reserve((size_t)ps->capacity * NQ * 16));
  PP_HIP_CHECK(ps->s_ranknew.reserve(sizeof(int) * (size_t)std::max(n_new, 1)));
  PP_HIP_CHECK(hipMemsetAsync(ps->s_rs.p, 0, sizeof(int) * 5 * (size_t)std::max(ne, 1), st));
  PP_HIP_CHECK(hipMemsetAsync(ps->s_misc.p, 0, sizeof(Totals), st));
  Totals* tot = ps->s_misc.as<Totals>();
  RsCounters cn{ps->s_rs.as<int>(), ps->s_rs.as<int>() + ne, ps->s_rs.as<int>() + 2 * (size_t)ne,
                ps->s_rs.as<int>() + 3 * (size_t)ne, ps->s_rs.as<int>() + 4 * (size_t)ne};
  int* n_new_e = ps->s_ppe.as<int>();
  const int* n_old = ps->d_elem_count.as<int>();
  int* rank = ps->s_idx.as<int>();
  int* rank_new = ps->s_ranknew.as<int>();
  int* holes = ps->s_holes.as<int>();
  uint4* aos = ps->s_aos.as<uint4>();
  const int G = std::max(1, 32 / ps->tile_p);
  const unsigned grp_grid = grid_for(((size_t)ps->ntiles_max + G - 1) / G * ps->C);
#define PP_RS_TILES                                                                                  \
  ps->d_ntiles.as<int>(), ps->C, ps->tile_p, G, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(), \
      ps->d_chunk_width.as<int>(), ps->d_row_to_element.as<int>()
#define PP_RS_CASE(N)                                                                                   \
  case N:                                                                                               \
    k_rs_count<N><<<grp_grid, kBlock, 0, st>>>(PP_RS_TILES, n_old, new_element, ne, cn, tot, rank, aos, wt); \
    break;
  switch (NQ) { PP_RS_CASE(1) PP_RS_CASE(2) PP_RS_CASE(4) PP_RS_CASE(6) PP_RS_CASE(8) PP_RS_CASE(10) }
#undef PP_RS_CASE
  if (n_new > 0) k_rs_count_added<<<grid_for(n_new), kBlock, 0, st>>>(n_new, new_elems, ne, cn.arrive, tot, rank_new);
  k_rs_fit<<<std::min(grid_for(std::max(ne, 1)), 256u), kBlock, 0, st>>>(
      ne, ps->C, n_old, cn, ps->d_element_to_row.as<int>(), ps->d_chunk_width.as<int>(), n_new_e, tot);
  k_rs_go<<<1, 1, 0, st>>>(tot);
  const int* go = &tot->go;
  k_rs_plan<<<grp_grid, kBlock, 0, st>>>(PP_RS_TILES, n_old, n_new_e, new_element, ne, cn, holes, rank,
                                        ps->d_mask.as<unsigned char>(), go);
#define PP_RS_CASE(N)                                                                                     \
  case N:                                                                                                 \
    k_rs_move<N><<<grp_grid, kBlock, 0, st>>>(PP_RS_TILES, n_old, n_new_e, new_element, ne,              \
                                              ps->d_eslot0.as<int>(), holes, rank, aos, wt, go);          \
    break;
  switch (NQ) { PP_RS_CASE(1) PP_RS_CASE(2) PP_RS_CASE(4) PP_RS_CASE(6) PP_RS_CASE(8) PP_RS_CASE(10) }
#undef PP_RS_CASE
#undef PP_RS_TILES
  // x_tgt <- 0 of the fused updatePtclPositions.  Without new particles the zeros stay pending: the
  // next fused push overwrites x_tgt of every live particle (pp_push_search), anything else that
  // looks at the member materialises them first (pp::ps_ready).
  const bool lazy = commit && n_new == 0;
  if (commit && !lazy) {
    const long long nwords = (long long)ps->stride * ps->member_ncomp[commit_x];
    k_zero_gated<<<2048, kBlock, 0, st>>>((unsigned long long*)ps->data[commit_x].p, nwords, go);
  }
  if (n_new > 0) {
    MoveArgs add = make_move(ps, ps->data, ps->stride, ps->data, ps->stride);
    for (int m = 0; m < ps->nmembers; ++m) add.src[m] = new_info[m];
    add.src_stride = n_new;
    if (commit) std::swap(add.dst[commit_x], add.dst[commit_xt]);  // the buffers trade places below
    k_rs_add<<<grid_for(n_new), kBlock, 0, st>>>(n_new, new_elems, rank_new, ps->d_eslot0.as<int>(), holes, ps->C,
                                                 add, go);
  }
  PP_LAUNCH_CHECK();
  Totals* h_pin;
  hipEvent_t ev_tot;
  int rc = totals_pin(ps, &h_pin, &ev_tot);
  if (rc) return rc;
  PP_HIP_CHECK(hipMemcpyAsync(h_pin, tot, sizeof(Totals), hipMemcpyDeviceToHost, st));
  PP_HIP_CHECK(hipEventRecord(ev_tot, st));
  // the new per-element counts are final whether or not the layout can be kept: the step's
  // scatters run behind them while the host waits
  bool scattered = false;
  if (pre_sync) {
    rc = pre_sync(n_new_e);
    if (rc) return rc;
    scattered = true;
  }
  (void)scattered;
  PP_HIP_CHECK(hipEventSynchronize(ev_tot));
  const Totals h = *h_pin;
  static const bool spec_debug_rs = PP_LAB_ENV("PP_SPEC_DEBUG") != nullptr;
  if (spec_debug_rs)
    fprintf(stderr, "rebuild in place: go %d active %d nonempty %d invalid %d overflowing rows %d\n", h.go, h.active,
            h.nonempty, h.invalid, h.n_over);
  if (h.invalid) {
    pp::set_error(
        "rebuild: a particle's new element is out of range, or a new particle is marked inactive "
        "(element id -1) -- the reference exits here (SCS_rebuild.h:147-151)");
    return PP_EINVAL;
  }
  if (!h.go) return 0;
  ps->d_elem_count.swap(ps->s_ppe);
  ps->version = pp::next_version();
  ps->num_ptcls = h.active;
  ps->num_empty_elements = (ps->num_rows - ne) + (ne - h.nonempty);
  if (commit) {
    ps->data[commit_x].swap(ps->data[commit_xt]);
    ps->zero_pending = lazy ? commit_xt : -1;
  }
  ++ps->n_reshuffles;
  ps->lazy_rec = 0;  // (a commit that came in with only the origin in records: the origin is the old target now)
  ps->hot = pp::HotRow{};  // (rows changed in place)
  return 1;
}

// `pre_sync` (may be empty): work that depends on nothing but the new per-element counts, enqueued
// before the rebuild's host sync so that the GPU has something to run while the host wakes up
int scs_rebuild(pp_ps* ps, const int* new_element, int n_new, const int* new_elems,
                const void* const* new_info, int commit_x, int commit_xt,
                const std::function<int(const int*)>& pre_sync = std::function<int(const int*)>(),
                bool try_reshuffle = true, bool new_xt_zero = false) {
  hipStream_t st = pp::stream();
  const int ne = ps->num_elems;
  if (ne == 0) {  // a structure without elements holds nothing and can take nothing (the reference's "empty_ptcls" runs)
    PP_REQUIRE(n_new == 0, "rebuild: a structure without elements cannot take new particles");
    return PP_OK;
  }
  PP_REQUIRE(n_new == 0 || new_info != nullptr, "rebuild: new particles need new_info_dev");
  ps->search_nf = -1;
  // zeros left pending by the previous in-place rebuild; records left by the previous full re-layout.
  // Exception: after the record-fed push only the ORIGIN (member lazy_x) is still in records, and a
  // rebuild that commits the same pair of members never reads it.
  // Exception 2 (round 5): every member is in the previous re-layout's records (lazy_rec == 3) and this rebuild
  // commits nothing: its first pass reads them there (k_move_pack_rec).
  const bool from_rec = ps->lazy_rec == 3 && ps->zero_pending < 0 && commit_x < 0 && commit_xt < 0 && ps->rec_nq > 4;
  if (!from_rec && !(ps->lazy_rec == 2 && ps->zero_pending < 0 && (n_new == 0 || new_xt_zero) && commit_x >= 0 &&
                     commit_x == ps->lazy_x && commit_xt == ps->lazy_xt)) {
    int rc0 = pp::ps_ready(ps);
    if (rc0) return rc0;
  }
  pp::Range rg("scs_rebuild");
  // The reference's reshuffle decision (SellCSigma::setShuffling) is evaluated on the histogram of the full
  // path below, which costs nothing when the layout cannot be kept -- the normal case at 10^5 rows.
  // histogram and totals in one allocation: one fill clears both (a fill is a ~5 us dispatch)
  const size_t tot_off = (sizeof(int) * (size_t)std::max(ne, 1) + 255) / 256 * 256;
  PP_HIP_CHECK(ps->s_ppe.reserve(tot_off + sizeof(Totals)));
  // (cleared by the tail of the previous full re-layout when that is what this buffer last saw: pp_ps::ppe_zeroed)
  if (!(ps->ppe_zeroed == ps->s_ppe.p && ps->ppe_zeroed_bytes >= tot_off + sizeof(Totals)))
    PP_HIP_CHECK(hipMemsetAsync(ps->s_ppe.p, 0, tot_off + sizeof(Totals), st));
  ps->ppe_zeroed = nullptr;
  Totals* tot = (Totals*)((char*)ps->s_ppe.p + tot_off);
  int* ppe = ps->s_ppe.as<int>();
  const bool have_old = ps->capacity > 0 && ps->num_ptcls > 0;
  const unsigned old_grid = grid_for((size_t)ps->ntiles_max * ps->C);
  // tiles per thread of the count/assign kernels: G*TP <= 32 (stay bit mask)
  const int G = std::max(1, 32 / ps->tile_p);
  const unsigned grp_grid = grid_for(((size_t)ps->ntiles_max + G - 1) / G * ps->C);
  // ranks inside the new rows, returned by the histogram's atomics (old particles, then new ones)
  PP_HIP_CHECK(ps->s_idx.reserve(sizeof(int) * (size_t)std::max(ps->capacity, 1)));
  PP_HIP_CHECK(ps->s_ranknew.reserve(sizeof(int) * (size_t)std::max(n_new, 1)));
  int* rank = ps->s_idx.as<int>();
  int* rank_new = ps->s_ranknew.as<int>();
  // (pp_ps::hot) the over-full row's own columns go through their own blocks -- in the steady state of the
  // record-fed loop, where this rebuild takes the row-major staged path again (the test at the top of this function)
  static const bool no_hot = PP_LAB_ENV("PP_NO_HOT_ROW") != nullptr;  // (lab build: the over-full row through the common blocks)
  const bool steady = ps->lazy_rec == 2 && ps->zero_pending < 0 && (n_new == 0 || new_xt_zero) && commit_x >= 0 &&
                      commit_x == ps->lazy_x && commit_xt == ps->lazy_xt;
  const pp::HotRow hot_now =
      (steady && ps->rec_rm && ps->hot.on && have_old && old_grid > 0 && ps->C == 64 && !no_hot) ? ps->hot : pp::HotRow{};
  if (have_old && old_grid > 0) {
    const unsigned hot_blocks = hot_now.on ? (unsigned)((hot_now.w - hot_now.c1p + 256 * kHotCols - 1) / (256 * kHotCols)) : 0u;
    // Thin chunks (ps_combo160 at 10^6 elements / 10^6 particles: ~5 columns, ONE tile per chunk): a thread that
    // takes G tiles takes G different rows of G different chunks -- nothing to merge, and its returning atomics
    // (one L2-miss line each) queue up behind one another.  One tile per thread there: four times the threads,
    // each with one round of atomics in flight.
    const bool thin = ps->num_chunks > 0 && (long long)ps->capacity <= (long long)ps->num_chunks * ps->C * ps->tile_p;
    const int Gc = thin ? 1 : G;
    const unsigned grp_grid_c = thin ? grid_for((size_t)ps->ntiles_max * ps->C) : grp_grid;
    // (three keys beside the row's own element; six: 64.3 against 60.3 us at c3, 366 against 358 at the c5 share)
#define PP_COUNT_ARGS                                                                                      \
  ps->d_ntiles.as<int>(), ps->C, ps->tile_p, Gc, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(),       \
      ps->d_chunk_width.as<int>(), ps->d_row_to_element.as<int>(), ps->d_mask.as<unsigned char>(), new_element, \
      ne, ppe, tot, rank, 1, hot_now, hot_blocks
    if (ps->C == 64)
      k_count_tiled<3, true><<<grp_grid_c + hot_blocks, kBlock, 0, st>>>(PP_COUNT_ARGS);
    else
      k_count_tiled<3, false><<<grp_grid_c + hot_blocks, kBlock, 0, st>>>(PP_COUNT_ARGS);
#undef PP_COUNT_ARGS
  }
  if (n_new > 0) {
    k_count_added<<<grid_for(n_new), kBlock, 0, st>>>(n_new, new_elems, ne, ppe, tot, rank_new);
  }
  // 64..256 blocks: each block ends with two atomics on the same two counters (~5 ns apiece), each
  // thread strides over ne / (blocks * 256) elements
  // the reference's reshuffle decision (mode 1), on the histogram just built
  const bool decide_keep = try_reshuffle && ps->shuffle_mode >= 1 && have_old && ne > 0 && ps->elem_count_valid;
  // totals of the new population + the decision: in the sweep that makes the sort keys when the layout
  // is sorted (one launch instead of three), else their own kernels
  // (up to 256 key blocks: every block ends with three atomics on the same counters, ~10 ns each)
  const bool inv_pad = ps->pad_strat == PP_PAD_INVERSELY && ps->shuffle_padding > 0;
  const bool totals_in_keys = ps->sigma > 1 && ne > 1 && (!inv_pad || (ne + RS_TILE - 1) / RS_TILE <= 256);
  if (!totals_in_keys) {
    if (ne > 0)
      k_nonempty<<<std::min(grid_for(ne), std::min(256u, std::max(64u, (unsigned)(ne / 4096)))), kBlock, 0, st>>>(
          ne, ppe, tot);
    if (decide_keep)
      k_fit_check<<<std::min(grid_for(ne), 256u), kBlock, 0, st>>>(ne, ps->C, ppe, ps->d_element_to_row.as<int>(),
                                                                 ps->d_chunk_width.as<int>(), tot);
  }
  // Sort keys are (window, count) with count < key_base; an upper bound known on the host avoids
  // a D2H read of the live count before the layout can start.
  const long long key_base = (long long)(have_old ? ps->num_ptcls : 0) + n_new + 1;
  // radix passes: the largest key of the previous rebuild (+1 bit) predicts how many are needed;
  // a pass that turns out to be unnecessary still costs five launches (45 us of a 0.44 ms step at
  // 1 M elements / 1 M particles)
  int bits_pred = 0;
  if (ps->last_max_key != ~0ull) {
    // (margin: 1/16 of the last maximum.  A whole extra bit -- round 2 -- launched a third, idle, 8-bit pass
    // for the literal pseudoXGCm population, whose largest row holds 60 000 particles: 16 bits exactly)
    const unsigned long long guess = ps->last_max_key + ps->last_max_key / 16 + 16;
    while (bits_pred < 63 && (guess >> bits_pred)) ++bits_pred;
  }
  LayoutPlan L;
  int nchunks = ne / ps->C_max + (ne % ps->C_max != 0), nrows = nchunks * ps->C_max;
  int C_new = ps->C_max;
  int ntiles_max = 0;
  PP_HIP_CHECK(ps->s_eslot0.reserve(sizeof(int) * (size_t)std::max(ne, 1)));
  // buffers and arguments of the table fills of the new layout, for buffers sized (cap_sz, nsl_sz)
  auto make_tables = [&](int cap_sz, int nsl_sz, LayoutTablesArgs& ta) -> int {
    PP_HIP_CHECK(ps->s_r2e2.reserve(sizeof(int) * (size_t)nrows));
    PP_HIP_CHECK(ps->s_e2r2.reserve(sizeof(int) * (size_t)nrows));
    PP_HIP_CHECK(ps->s_rowstart.reserve(sizeof(int) * (size_t)nrows));
    PP_HIP_CHECK(ps->s_offsets2.reserve(sizeof(int) * ((size_t)nsl_sz + 1)));
    PP_HIP_CHECK(ps->s_s2c2.reserve(sizeof(int) * (size_t)std::max(nsl_sz, 1)));
    PP_HIP_CHECK(ps->s_mask2.reserve((size_t)std::max(cap_sz, 1)));
    PP_HIP_CHECK(ps->s_slot2.reserve(sizeof(int) * (size_t)std::max(cap_sz, 1)));
    PP_HIP_CHECK(ps->s_scan.reserve(sizeof(int)));
    // sum_c ceil(w_c/TP) <= nchunks + capacity/(C*TP): sizes the launch without reading the count
    ntiles_max = nchunks + cap_sz / (C_new * ps->tile_p) + 1;
    PP_HIP_CHECK(ps->s_newidx.reserve(sizeof(int) * 2 * (size_t)ntiles_max));  // new tile table
    ta.b1 = grid_for(ntiles_max);
    ta.b2 = ta.b1 + grid_for(std::max(nchunks, 1));
    ta.b3 = ta.b2 + grid_for(nrows);
    ta.ntiles_dev = ps->s_scan.as<int>();
    ta.nchunks = nchunks;
    ta.TP = ps->tile_p;
    ta.C = C_new;
    ta.V = ps->V;
    ta.nrows = nrows;
    ta.ne = ne;
    ta.sorted = (ps->sigma > 1 && ne > 1) ? 1 : 0;
    ta.tile_off = L.tile_off;  // (the layout's own arrays: known once it is enqueued)
    ta.widths = L.widths;
    ta.slice_off = L.slice_off;
    ta.chunk_start = L.chunk_start;
    ta.index = L.index;
    ta.tiles = ps->s_newidx.as<int>();
    ta.offsets = ps->s_offsets2.as<int>();
    ta.s2c = ps->s_s2c2.as<int>();
    ta.r2e = ps->s_r2e2.as<int>();
    ta.e2r = ps->s_e2r2.as<int>();
    ta.row_cursor = ps->s_rowstart.as<int>();
    ta.eslot0 = ps->s_eslot0.as<int>();
    ta.tot = tot;
    return PP_OK;
  };
  // ---- limits of the speculative tail (see below): known before the layout is enqueued, so that the gate
  // rides in the layout kernel
  bool spec_ok = false;
  long long cap_lim = 0, nsl_lim = 0;
  int64_t stride_fit = 0;
  // (lab build, PP_NO_SPEC_REBUILD=1: every rebuild takes the checked path a rebuild takes when some buffer must grow)
  static const bool no_spec = PP_LAB_ENV("PP_NO_SPEC_REBUILD") != nullptr;
  if (!no_spec && have_old && old_grid > 0) {
    const int nchunks0 = ne / ps->C_max + (ne % ps->C_max != 0);
    cap_lim = std::min<long long>((long long)ps->s_mask2.bytes, (long long)(ps->s_slot2.bytes / 4));
    // the component stride the swap buffers can hold today, on the spread_stride pattern
    long long fit = (int)ps->swap.size() >= ps->nmembers ? (1ll << 40) : 0;
    for (int m = 0; m < ps->nmembers && fit > 0; ++m)
      fit = std::min<long long>(fit, (long long)(ps->swap[m].bytes /
                                                 ((size_t)ps->member_ncomp[m] * ps->member_bytes[m])));
    long long sq = fit / 64;
    while (sq > 0 && sq % 32 != 17) --sq;
    stride_fit = sq * 64;
    cap_lim = std::min<long long>(cap_lim, stride_fit);
    if (cap_lim > 0) {  // staging buffer: NQ quads per slot
      const void* srcs[8];
      for (int m = 0; m < ps->nmembers; ++m) srcs[m] = ps->data[m].p;
      WordTable wt_probe{};
      const int nq = build_word_table(ps, srcs, ps->stride, stride_fit, commit_x, commit_xt, wt_probe);
      if (nq > 0) {
        // (the pseudoXGCm type with the commit: 32-B records + the side word, as enqueue_tail will decide)
        const bool compact = nq == 4 && commit_x >= 0 && commit_xt >= 0 && (n_new == 0 || new_xt_zero) &&
                             !no_lazy_unpack() && xgcm_shape(ps);
        const long long spare = (long long)kRecSpareCols * ps->C_max * nchunks0;
        cap_lim = std::min<long long>(cap_lim, (long long)(ps->s_aos.bytes / ((size_t)(compact ? 2 : nq) * 16)) - spare);
        if (compact) cap_lim = std::min<long long>(cap_lim, (long long)(ps->s_side.bytes / sizeof(unsigned)) - spare);
      }
    }
    const long long tiles_room = (long long)(ps->s_newidx.bytes / 8) - nchunks0 - 1;
    cap_lim = std::min<long long>(cap_lim, tiles_room * (long long)(ps->C_max * ps->tile_p));
    nsl_lim = std::min<long long>((long long)(ps->s_offsets2.bytes / 4) - 1, (long long)(ps->s_s2c2.bytes / 4));
    cap_lim = std::min<long long>(cap_lim, 2147483647ll / 2);
    spec_ok = cap_lim >= ps->capacity / 2 && cap_lim > 0 && nsl_lim > 0;
  }
  LayoutTablesArgs ta_spec{};
  if (spec_ok)
    if (int rct = make_tables((int)cap_lim, (int)nsl_lim, ta_spec)) return rct;
  Totals* h_pin = nullptr;
  hipEvent_t ev_tot = nullptr;
  if (int rcp = totals_pin(ps, &h_pin, &ev_tot)) return rcp;
  // the host polls the landing zone for this rebuild's stamp (no event in the stream); one poll that runs into
  // its time limit (the memory turned out not to be visible mid-stream) switches back to the event for good
  static bool poll_totals = true;
  const int stamp = poll_totals ? (ps->totals_stamp = ps->totals_stamp % 1000000 + 1) : 0;
  int rc = enqueue_layout(ps, ps->C_max, ppe, tot, key_base, L, bits_pred,
                          ElemTotalsArgs{totals_in_keys ? 1 : 0, (totals_in_keys && decide_keep) ? 1 : 0, ps->C,
                                         ps->d_element_to_row.as<int>(), ps->d_chunk_width.as<int>(), nullptr},
                          SpecArgs{spec_ok ? 1 : 0, (int)cap_lim, (int)nsl_lim, ps->C_max, decide_keep ? 1 : 0,
                                   pp::search_not_found_dev(ps)},
                          spec_ok ? &ta_spec : nullptr, /*allow_wide=*/true, h_pin, stamp);
  const bool polling = stamp != 0 && L.totals_on_host;
  if (rc) return rc;
  const int sorted_ndig = L.wide_ndig;  // (a retry below re-plans L)
  nchunks = L.nchunks;
  nrows = L.nrows;
  const int* go = &tot->go;
  int NQ = 0;
  bool lazy_zero = false, defer_unpack = false, use_rm = false, defer_wide = false, rec_split = false;
  // today's per-element counts are not read by a re-layout that commits: its tail clears them for the next one
  // slot -> element: left out (pp::slot_elem fills it when something asks) unless the table of the layout that is
  // being replaced WAS asked for -- ps_combo160's loop redistributes (reads it) before every rebuild: written here by
  // the slot blocks it costs 4 B per slot, filled on demand a launch and ~6 us per round
  const bool lazy_slot_elem = !ps->slot_elem_used;
  ps->slot_elem_used = false;
  // staging records row-major inside a chunk (pp_ps::rec_rm)
  const bool want_rm = have_old && old_grid > 0;
  if (want_rm) PP_HIP_CHECK(ps->s_erec0.reserve(sizeof(int) * (size_t)std::max(ne, 1)));
  const bool prezero = ps->d_elem_count.p && ps->d_elem_count.p != ps->s_ppe.p &&
                       ps->d_elem_count.bytes >= tot_off + sizeof(Totals);
  // Everything after the layout: new layout arrays, slot tables and the move of every member.
  // `cap_sz` / `nsl_sz` size the buffers and the launches; the kernels themselves read the true
  // counts from the device (ntiles, Totals), and all of them return at once when tot->go == 0.
  auto enqueue_tail = [&](int cap_sz, int nsl_sz, int64_t stride_fixed) -> int {
    LayoutTablesArgs ta;
    if (int rct = make_tables(cap_sz, nsl_sz, ta)) return rct;
    int* new_tiles = ps->s_newidx.as<int>();
    const int* new_ntiles = ps->s_scan.as<int>();
    InitSlotsArgs isa{new_ntiles, C_new, ps->tile_p, new_tiles, L.chunk_start, L.widths, ps->s_r2e2.as<int>(), ppe, ne,
                      lazy_slot_elem ? nullptr : ps->s_slot2.as<int>(), ps->s_rowstart.as<int>(),
                      ps->s_eslot0.as<int>(), ps->s_mask2.as<unsigned char>(), go,
                      prezero ? ps->d_elem_count.as<int>() : nullptr,
                      prezero ? (int)((tot_off + sizeof(Totals)) / sizeof(int)) : 0,
                      want_rm ? ps->s_erec0.as<int>() : nullptr, nullptr, nullptr, nchunks, ta.sorted};
    const unsigned table_blocks = ta.b3 + grid_for(nrows);
    if (C_new % 4 == 0) {  // one launch: table blocks + slot blocks (which find tile and element themselves)
      isa.tile_off = L.tile_off;
      isa.index = L.index;
      k_layout_tables_slots<<<table_blocks + grid_for((size_t)ntiles_max * (C_new / 4)), kBlock, 0, st>>>(ta, isa,
                                                                                                      table_blocks);
    } else {
      k_layout_tables<<<table_blocks, kBlock, 0, st>>>(ta);
      if (C_new % 4 == 0)
        k_init_slots_tiled4<<<grid_for((size_t)ntiles_max * (C_new / 4)), kBlock, 0, st>>>(isa);
      else
        k_init_slots_tiled<<<grid_for((size_t)ntiles_max * C_new), kBlock, 0, st>>>(
            isa.ntiles_dev, isa.C, isa.TP, isa.tiles, isa.chunk_start, isa.chunk_width, isa.r2e, isa.ppe, isa.ne,
            isa.slot_elem, isa.row_cursor, isa.elem_slot0, isa.new_mask, isa.go, isa.zero_next, isa.zero_words,
            isa.elem_rec0);
    }
    // ---- swap buffer sizing (SCS_rebuild.h:223-229)
    int64_t swap_stride = ps->swap_stride;
    if (stride_fixed > 0) {  // speculative tail: whatever the swap buffers hold today
      swap_stride = stride_fixed;
    } else if (swap_stride < cap_sz || swap_stride * ps->minimize_size < cap_sz) {
      swap_stride = (int64_t)(cap_sz * (1 + ps->extra_padding));
      if (swap_stride < cap_sz) swap_stride = cap_sz;
      swap_stride = spread_stride(swap_stride);
    }
    int rc2 = alloc_members(ps, ps->swap, swap_stride, false);
    if (rc2) return rc2;
    ps->swap_stride = swap_stride;
    // ---- move every member of every live particle
    MoveArgs mv = make_move(ps, ps->data, ps->stride, ps->swap, swap_stride);
    mv.commit_x = commit_x;
    mv.commit_xt = commit_xt;
    // record = all members (fast path needs 4/8-byte scalars, see build_word_table)
    WordTable wt{};
    NQ = 0;
    if (have_old && old_grid > 0) {
      const void* srcs[8];
      for (int m = 0; m < ps->nmembers; ++m) srcs[m] = ps->data[m].p;
      const bool zero_z = ps->zero_z_pending && commit_x >= 0 && commit_x == ps->lazy_x && commit_xt == ps->lazy_xt;
      NQ = build_word_table(ps, srcs, ps->stride, swap_stride, commit_x, commit_xt, wt, -1, zero_z);
      // x_tgt <- 0 of the fused updatePtclPositions stays pending (pp_ps::zero_pending): the next fused
      // push overwrites the member, anything else materialises the zeros first.  24 of the 60 bytes
      // pass 2 would write per particle.
      lazy_zero = NQ > 0 && commit_x >= 0 && commit_xt >= 0 && (n_new == 0 || new_xt_zero);
      if (lazy_zero) wt.nz8 = wt.nz4 = 0;
      // The second pass (records -> new SoA arrays) is deferred for the pseudoXGCm particle type: the
      // next fused push reads the records themselves (pp_search.hip: RECIN), anything else runs the
      // pass first (ps_ready).  One pass over the particles less per step of the pseudoXGCm loop.
      // (lab build, PP_NO_LAZY_UNPACK=1: pass 2 runs right away -- the path every other particle type takes)
      const bool no_defer = no_lazy_unpack();
      defer_unpack = lazy_zero && NQ == 4 && !no_defer && xgcm_shape(ps);
      if (defer_unpack) {
        // ... as 32-B records: origin, phi, b; the third member (4 bytes) travels beside them (WordTable::side_*) --
        // with it the record would be 36 B, i.e. a 64-B sector per particle, written here and read by the push
        NQ = build_word_table(ps, srcs, ps->stride, swap_stride, commit_x, commit_xt, wt, 2, zero_z);
        PP_REQUIRE(NQ == 2 && wt.side_src, "rebuild (internal): the pseudoXGCm record is not 32 bytes + one word");
        wt.nz8 = wt.nz4 = 0;
        // The 2-D loop (test/pseudoXGCm.cpp on triangles; zero_z: the last push was the 2-D record-fed one, the
        // origin's third component is known to be zero): SPLIT records -- quad 0 (x, y) to one array, quad 1
        // (pad, phi, b, id) to another.  The 2-D push reads phi, b and the id only (test/ellipticalPush.hpp:51-67 and
        // search_mesh_2d, src/pumipic_adjacency.hpp:1045-1117, never look at the origin): 16 B per particle instead
        // of the 32-B record + the side word (round-5 verdict: 0.70 GB read per step for 10 M particles).
        rec_split = zero_z && n_new == 0 && want_rm && !from_rec && !no_rec_split();
        if (rec_split) {
          NQ = build_word_table(ps, srcs, ps->stride, swap_stride, commit_x, commit_xt, wt, -1, zero_z, /*drop_z=*/true);
          PP_REQUIRE(NQ == 2 && wt.n8 == 2 && wt.n4 == 3 && !wt.side_src,
                     "rebuild (internal): the split 2-D record is not (x, y | pad, phi, b, id)");
          wt.nz8 = wt.nz4 = 0;
        }
      }
      // Records wider than 64 B (ps_combo160's 160-B particle: 192-B records), no new particles, no commit: the
      // second pass waits until somebody asks for a member -- a rebuild that follows reads the records
      // (performance_tests/ps_combo160.cpp:205-232 rebuilds a hundred times without touching a member)
      defer_wide = NQ > 4 && n_new == 0 && commit_x < 0 && commit_xt < 0 && !no_defer;
      PP_REQUIRE(!from_rec || NQ == ps->rec_nq, "rebuild (internal): the live records have another width");
    }
    if (NQ > 0) {
      // (+ one spare column per chunk: the row-major record geometry, pp_rec_row0)
      const size_t nrec = (size_t)std::max(cap_sz, 1) + (size_t)kRecSpareCols * C_new * nchunks;
      PP_HIP_CHECK(ps->s_aos.reserve(nrec * NQ * 16));
      if (defer_unpack) PP_HIP_CHECK(ps->s_side.reserve(nrec * (rec_split ? sizeof(uint4) : sizeof(unsigned))));
      uint4* aos = ps->s_aos.as<uint4>();
      unsigned* const side = (defer_unpack && !rec_split) ? ps->s_side.as<unsigned>() : nullptr;
      uint4* const split_hot = rec_split ? ps->s_side.as<uint4>() : nullptr;  // (the side buffer holds the second halves)
      const unsigned new_grid = grid_for((size_t)ntiles_max * C_new);
#define PP_UNPACK_ARGS                                                                                  \
  new_ntiles, C_new, ps->tile_p, new_tiles, L.chunk_start, L.widths, ps->s_mask2.as<unsigned char>(), aos, \
      wt, go, use_rm ? 1 : 0
#define PP_STAGED(N)                                                                             \
  case N:                                                                                        \
    if (from_rec)                                                                                \
      k_move_pack_rec<N><<<grid_for(ps->capacity), kBlock, 0, st>>>(ps->capacity, rank, rs,      \
                                                                    ps->s_aos_live.as<uint4>(), aos, go); \
    else if (use_rm)                                                                             \
      k_move_pack_rm<N><<<pack_main + pack_hot, kBlock, 0, st>>>(pack_end, rank, rs_rm, aos, wt, go, \
                                                                 ps->C == 64 ? rm_wide : 0, hot_now, pack_hot, side, \
                                                                 N == 2 ? split_hot : nullptr);  \
    else                                                                                         \
      k_move_pack<N><<<grid_for(ps->capacity), kBlock, 0, st>>>(ps->capacity, rank, rs, aos, wt, go, side); \
    if (!defer_unpack && !defer_wide) k_move_unpack<N><<<new_grid, kBlock, 0, st>>>(PP_UNPACK_ARGS); \
    break;
      const RankToSlot rs{new_element, ps->s_eslot0.as<int>(), C_new};
      // the staging records row-major inside a chunk: the particles of a row, which carry consecutive ranks,
      // are consecutive records, and the pack's block-level transpose stores them as runs
      const RankToSlot rs_rm{new_element, ps->s_erec0.as<int>(), 1};
      // log2(columns per block): a row's run of a block is 512 contiguous bytes -- 8 of the 64-B records, 16 of the
      // 32-B ones (whose side words then leave as 64-B pieces)
      // (split 2-D records are two arrays of 16-B halves; 16 columns per block as for the 32-B records -- 32 columns,
      //  512-B runs in each array, measured slower: pack 237 against 182 us, profiles/r06_ab_split_records.txt;
      //  lab build: PP_SPLIT_WIDE overrides)
      static const int split_wide = PP_LAB_ENV("PP_SPLIT_WIDE") ? atoi(PP_LAB_ENV("PP_SPLIT_WIDE")) : 4;
      const int rm_wide = rec_split ? split_wide : NQ <= 2 ? 4 : 3;
      // (only where the records stay the particle data -- the pseudoXGCm flows, where nine particles in ten keep
      // their row: c3 -3.3 %, 2dc3 -3.6 %, c5 share -3.2 %.  ps_combo160 redistributes half of its particles to
      // random elements: nothing to merge, and its second pass reads row-major records 2.5 % slower)
      // ... and for every type whose record is one 64-B half line or less, deferred or not (the drop-in loop of
      // test/pseudoXGCm.cpp rebuilds without the fused commit and reads the SoA arrays right away: pass 1 358 ->
      // row-major, pass 2 reads the same records back): two records share a 128-B line, so runs are what keeps
      // the scattered stores whole lines
      use_rm = want_rm && (defer_unpack || NQ <= 4) && !from_rec;
      PP_REQUIRE(!hot_now.on || use_rm, "rebuild (internal): the over-full row's blocks need the row-major staged path");
      // (pp_ps::hot) the main blocks end where the columns of the over-full row begin, its own blocks follow
      const int pack_end = hot_now.on ? hot_now.start + 64 * hot_now.c1p : ps->capacity;
      const unsigned pack_main = (grid_for(pack_end) + 15) / 16 * 16;
      const unsigned pack_hot = hot_now.on ? (unsigned)((hot_now.w - hot_now.c1p + 255) / 256) : 0u;
      switch (NQ) {
        PP_STAGED(1) PP_STAGED(2) PP_STAGED(3) PP_STAGED(4) PP_STAGED(6) PP_STAGED(8) PP_STAGED(10) PP_STAGED(12)
      }
#undef PP_STAGED
#undef PP_UNPACK_ARGS
    } else if (have_old && old_grid > 0)
      k_move_tiled<<<old_grid, kBlock, 0, st>>>(
          ps->d_ntiles.as<int>(), ps->C, ps->tile_p, ps->d_tiles.as<int>(),
          ps->d_chunk_start.as<int>(), ps->d_chunk_width.as<int>(), ps->d_row_to_element.as<int>(),
          ps->d_mask.as<unsigned char>(), new_element, ps->s_e2r2.as<int>(), C_new,
          ps->s_rowstart.as<int>(), ps->s_mask2.as<unsigned char>(), mv, go);
    if (n_new > 0 && defer_unpack) {
      // (pseudoXGCm type, member commit_xt of the arrivals known to be zero: they join the records)
      k_add_rec<<<grid_for(n_new), kBlock, 0, st>>>(
          n_new, new_elems, rank_new, use_rm ? ps->s_erec0.as<int>() : ps->s_eslot0.as<int>(), use_rm ? 1 : C_new,
          (const unsigned long long*)new_info[commit_x],
          (const unsigned*)new_info[2], (const unsigned*)new_info[3], (const unsigned*)new_info[4],
          ps->s_aos.as<uint4>(), ps->s_side.as<unsigned>(), go);
    } else if (n_new > 0) {
      MoveArgs add = mv;
      for (int m = 0; m < ps->nmembers; ++m) add.src[m] = new_info[m];
      add.src_stride = n_new;
      add.commit_x = add.commit_xt = -1;  // new particles arrive with their own positions
      k_add_scs<<<grid_for(n_new), kBlock, 0, st>>>(n_new, new_elems, ps->s_e2r2.as<int>(), C_new,
                                                    ps->s_rowstart.as<int>(),
                                                    NQ > 0 ? rank_new : nullptr, ps->s_eslot0.as<int>(),
                                                    ps->s_mask2.as<unsigned char>(), add, go);
    }
    return PP_OK;
  };
  // ---- speculative tail.  The host needs the new capacity and slice count to size buffers and
  // launches, which used to cost a sync in the middle of the rebuild (the GPU drained, then waited
  // for five launch latencies: ~80 us of a 1.1 ms step).  When every buffer already has room for
  // what it had room for last time, the whole tail is enqueued against those LIMITS first; a
  // one-thread kernel compares the true counts with them and clears tot->go when anything would
  // not fit (or the chunk height changes, or the input is invalid), which turns the tail into
  // no-ops.  The sync then happens once, after everything is queued.
  bool speculated = false;
  const int64_t swap_stride_before = ps->swap_stride;  // the speculative tail re-labels the swap buffers
  // The totals are final when the layout kernel ends.  They travel to pinned memory right behind it and the
  // host waits for THAT copy only: the tail (tables, slot init, the move) is enqueued behind the copy and is
  // still running when the host has finished its bookkeeping and the caller issues the next push -- the
  // host's wake-up is off the GPU's critical path (before: copy behind the tail, 27 us idle per step once the
  // scatter kernels that used to cover it rode in the rebuild's own launches).
  // (the layout kernel writes them there itself; the separate-kernel layout of PAD_INVERSELY copies them)
  if (!L.totals_on_host) PP_HIP_CHECK(hipMemcpyAsync(h_pin, tot, sizeof(Totals), hipMemcpyDeviceToHost, st));
  if (!polling) PP_HIP_CHECK(hipEventRecord(ev_tot, st));
  if (spec_ok) {  // (the gate itself ran at the end of the layout kernel)
    rc = enqueue_tail((int)cap_lim, (int)nsl_lim, stride_fit);
    if (rc) return rc;
    speculated = true;
  }
  if (pre_sync && !ps->ride_done) {  // (ride_done: the scatter rode in the key sweep and the layout kernel)
    rc = pre_sync(ppe);
    if (rc) return rc;
  }
  if (polling) {  // the only host wait of a regular rebuild
    const auto t0 = std::chrono::steady_clock::now();
    volatile int* flag = &h_pin->pad_[0];
    long spins = 0;
    while (*flag != stamp) {
      if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
        PP_HIP_CHECK(hipStreamSynchronize(st));  // everything queued has run: either the stamp is there now ...
        if (*flag != stamp) {                    // ... or kernel stores do not reach this memory mid-stream
          poll_totals = false;
          PP_HIP_CHECK(hipMemcpyAsync(h_pin, tot, sizeof(Totals), hipMemcpyDeviceToHost, st));
          PP_HIP_CHECK(hipStreamSynchronize(st));
        }
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    PP_HIP_CHECK(hipEventSynchronize(ev_tot));
  }
  Totals h = *h_pin;
  ps->search_nf = L.totals_on_host ? h.pad_[1] : -1;
  if (h.invalid) {
    ps->swap_stride = swap_stride_before;
    pp::set_error(
        "rebuild: a particle's new element is out of range, or a new particle is marked inactive "
        "(element id -1) -- the reference exits here (SCS_rebuild.h:147-151)");
    return PP_EINVAL;
  }
  if (decide_keep && h.active > 0 && h.n_over == 0) {
    // Every row's new count fits its chunk: the reference keeps the layout (SCS_rebuild.h:184-189).  The
    // speculative re-layout tail did not run (k_spec_check); the in-place path does the work.  Rare at
    // scale (some row of 10^5 overflows its padding nearly every step), common for small structures.
    ps->swap_stride = swap_stride_before;
    if (int rcm = pp::ps_ready(ps)) return rcm;  // (the in-place path moves between the member arrays)
    const int mode = ps->shuffle_mode;
    rc = scs_reshuffle(ps, new_element, n_new, new_elems, new_info, commit_x, commit_xt,
                       std::function<int(const int*)>());  // (the scatters ran behind the histogram above)
    ps->shuffle_mode = mode;
    if (rc < 0) return rc;
    if (rc == 1) return PP_OK;
    // The histogram's test is the loose one (new count <= chunk width); the reference's rule counts a row's
    // holes BEFORE its movers leave (SCS_rebuild.h:13-25), which scs_reshuffle evaluates exactly.  Loose but not
    // strict (two full rows exchanging particles: small structures) and members the in-place path cannot
    // stage (1/2-byte scalars) land here: full re-layout after all, without the decision -- correct, at the
    // price of a second histogram + sort + layout for that rebuild.
    return scs_rebuild(ps, new_element, n_new, new_elems, new_info, commit_x, commit_xt,
                       std::function<int(const int*)>(), false, new_xt_zero);
  }
  if (h.active == 0) {  // SCS_rebuild.h:168-182: no particle left -- resetMask, structure kept
    ps->swap_stride = swap_stride_before;
    if (int rcm = pp::ps_ready(ps)) return rcm;
    // the fused commit still happens: the drivers call updatePtclPositions before the rebuild
    if (commit_x >= 0 && commit_xt >= 0 && ps->num_ptcls > 0) {
      rc = pp_update_positions(ps, commit_x, commit_xt);
      if (rc) return rc;
    }
    if (ps->capacity > 0) PP_HIP_CHECK(hipMemsetAsync(ps->d_mask.p, 0, (size_t)ps->capacity, st));
    ps->hot = pp::HotRow{};
    ps->d_elem_count.swap(ps->s_ppe);  // the histogram just built: all zeros
    ps->elem_count_valid = true;
    ps->version = pp::next_version();
    ps->num_ptcls = 0;
    return PP_OK;
  }
  static const bool spec_debug = PP_LAB_ENV("PP_SPEC_DEBUG") != nullptr;
  if (spec_debug)
    fprintf(stderr, "rebuild: speculated %d go %d capacity %d (old %d) nslices %d active %d nonempty %d "
                    "key bits sorted %d max key %llu; over-full row's blocks %d (columns %d..%d)\n",
            (int)speculated, h.go, h.capacity, ps->capacity, h.nslices, h.active, h.nonempty, L.key_bits,
            h.max_key, hot_now.on, hot_now.c1p, hot_now.w);
  if (!(speculated && h.go)) {
    // chooseChunkHeight (SCS_buildFns.h:3-16): C shrinks only when fewer than C_max elements hold
    // particles -- redo the (tiny) layout with that height; same when the predicted number of
    // radix passes did not cover the largest key
    C_new = std::min(h.nonempty, ps->C_max);
    const bool sort_ok = (L.key_bits >= 64 || (h.max_key >> L.key_bits) == 0) && !h.sort_bad;
    if (C_new != ps->C_max || !sort_ok) {
      PP_HIP_CHECK(hipMemsetAsync(&tot->cw_sum, 0, sizeof(int) * 2, st));
      rc = enqueue_layout(ps, C_new, ppe, tot, key_base, L);
      if (rc) return rc;
      const int active = h.active, nonempty = h.nonempty;
      const unsigned long long max_key = h.max_key;
      PP_HIP_CHECK(hipMemcpyAsync(&h, tot, sizeof(Totals), hipMemcpyDeviceToHost, st));
      PP_HIP_CHECK(hipStreamSynchronize(st));
          h.active = active;
      h.nonempty = nonempty;
      h.max_key = std::max(h.max_key, max_key);
      nchunks = L.nchunks;
      nrows = L.nrows;
    }
    const int one = 1;
    PP_HIP_CHECK(hipMemcpyAsync(&tot->go, &one, sizeof(int), hipMemcpyHostToDevice, st));
    rc = enqueue_tail(h.capacity, h.nslices, 0);
    if (rc) return rc;
  }
  // (a population whose rows mostly exceed the one-pass sort's digit range: the 8-bit passes for a while)
  // (a narrow digit that overflowed: the next rebuilds take the 2048-digit pass; the 2048-digit pass itself
  //  overflowed: the 8-bit passes for a while)
  if (h.sort_bad && sorted_ndig < kWideDigits) ps->narrow_skip = 64;
  else if (h.sort_bad) ps->wide_skip = 64;
  else {
    if (ps->wide_skip > 0) --ps->wide_skip;
    if (ps->narrow_skip > 0) --ps->narrow_skip;
  }
  ps->last_max_key = h.max_key;
  const int new_capacity = h.capacity, new_nslices = h.nslices;
  ntiles_max = nchunks + new_capacity / (C_new * ps->tile_p) + 1;  // launch bound of the next calls
  PP_LAUNCH_CHECK();
  // ---- swap in
  ps->data.swap(ps->swap);
  std::swap(ps->stride, ps->swap_stride);
  ps->zero_pending = lazy_zero ? commit_xt : -1;
  ps->zero_z_pending = false;  // (packed as zeros above / written by ps_ready before a non-committing re-layout)
  ps->lazy_rec = 0;
  ps->rec_split = false;
  ps->hot = pp::HotRow{};
  if (defer_unpack) {  // the records of the first pass are what holds the particles now
    ps->s_aos.swap(ps->s_aos_live);
    ps->s_side.swap(ps->s_side_live);
    ps->lazy_rec = 1;
    ps->lazy_x = commit_x;
    ps->lazy_xt = commit_xt;
    ps->rec_rm = use_rm;
    ps->rec_split = rec_split;
    if (use_rm) ps->d_erec0.swap(ps->s_erec0);
    // (pp_ps::hot) one sort window: the rows are in ascending order of their counts, the last row holds the largest
    if (use_rm && C_new == 64 && h.second_key1 > 0 && h.max_key < (1ull << 30) && !no_hot) {
      const int c1p = (h.second_key1 - 1 + 31) / 32 * 32, w = (int)h.max_key;
      if (w - c1p >= 2048) {
        ps->hot.on = 1;
        ps->hot.chunk = nchunks - 1;
        ps->hot.row = (ne - 1) % 64;
        ps->hot.c1p = c1p;
        ps->hot.w = w;
        ps->hot.start = h.last_chunk_start;
      }
    }
  } else if (defer_wide && NQ > 4) {  // wide records, slot order: the particle data until a member is asked for
    ps->s_aos.swap(ps->s_aos_live);
    ps->lazy_rec = 3;
    ps->rec_nq = NQ;
    ps->rec_rm = false;
    ps->lazy_x = ps->lazy_xt = -1;
  }
  ps->d_offsets.swap(ps->s_offsets2);
  ps->d_slice_to_chunk.swap(ps->s_s2c2);
  ps->d_row_to_element.swap(ps->s_r2e2);
  ps->d_element_to_row.swap(ps->s_e2r2);
  ps->d_mask.swap(ps->s_mask2);
  ps->d_slot_elem.swap(ps->s_slot2);
  ps->slot_elem_valid = !lazy_slot_elem;
  ps->group_chunk_valid = false;
  ps->d_chunk_start.swap(ps->s_cstart2);
  ps->d_chunk_width.swap(ps->s_cwidth2);
  ps->d_tiles.swap(ps->s_newidx);
  ps->d_ntiles.swap(ps->s_scan);
  ps->d_elem_count.swap(ps->s_ppe);  // live particles per element == the histogram just built
  if (prezero) {
    ps->ppe_zeroed = ps->s_ppe.p;
    ps->ppe_zeroed_bytes = tot_off + sizeof(Totals);
  }
  ps->d_eslot0.swap(ps->s_eslot0);
  ps->elem_count_valid = true;
  ps->version = pp::next_version();
  ++ps->n_full_rebuilds;
  if (from_rec) ++ps->n_from_records;
  ps->ntiles_max = ntiles_max;
  ps->C = C_new;
  ps->num_ptcls = h.active;
  ps->num_chunks = nchunks;
  ps->num_slices = new_nslices;
  ps->capacity = new_capacity;
  ps->num_rows = nrows;
  ps->num_empty_elements = (nrows - ne) + (ne - h.nonempty);
  return PP_OK;
}

int csr_rebuild(pp_ps* ps, const int* new_element, int n_new, const int* new_elems,
                const void* const* new_info) {
  hipStream_t st = pp::stream();
  const int ne = ps->num_elems;
  if (ne == 0) {  // (see scs_rebuild)
    PP_REQUIRE(n_new == 0, "rebuild: a structure without elements cannot take new particles");
    return PP_OK;
  }
  PP_HIP_CHECK(ps->s_ppe.reserve(sizeof(int) * ((size_t)ne + 1)));
  PP_HIP_CHECK(ps->s_misc.reserve(sizeof(Totals)));
  PP_HIP_CHECK(hipMemsetAsync(ps->s_ppe.p, 0, sizeof(int) * ((size_t)ne + 1), st));
  PP_HIP_CHECK(hipMemsetAsync(ps->s_misc.p, 0, sizeof(Totals), st));
  Totals* tot = ps->s_misc.as<Totals>();
  int* ppe = ps->s_ppe.as<int>();
  // live slots are [0, offsets[ne]) == [0, num_ptcls)
  const int nold = ps->num_ptcls;
  // the counting atomics return each particle's rank inside its new element (old, then new ones)
  PP_HIP_CHECK(ps->s_idx.reserve(sizeof(int) * (size_t)std::max(nold, 1)));
  PP_HIP_CHECK(ps->s_ranknew.reserve(sizeof(int) * (size_t)std::max(n_new, 1)));
  int* rank = ps->s_idx.as<int>();
  int* rank_new = ps->s_ranknew.as<int>();
  if (nold > 0) k_count_csr<<<grid_for(nold), kBlock, 0, st>>>(nold, new_element, pp::slot_elem(ps), ne, ppe, tot, rank);
  if (n_new > 0) k_count_added<<<grid_for(n_new), kBlock, 0, st>>>(n_new, new_elems, ne, ppe, tot, rank_new);
  PP_HIP_CHECK(ps->s_offsets2.reserve(sizeof(int) * ((size_t)ne + 1)));
  if (scan_excl(ps->s_scan2, ne + 1, ppe, ps->s_offsets2.as<int>(), &tot->active, st)) return PP_EHIP;
  PP_LAUNCH_CHECK();
  if (n_new > 0) PP_REQUIRE(new_info != nullptr, "rebuild: new particles need new_info_dev");
  const int* go = &tot->go;
  int64_t new_stride = 0;
  const bool from_rec = ps->lazy_rec == 3 && ps->rec_nq > 4 && nold > 0;  // (pp_ps_rebuild left the records alone)
  bool defer_wide = false;
  int nq_used = 0;
  // everything after the counts: swap sizing, the two move passes, the slot tables (all kernels
  // return at once when tot->go == 0)
  auto enqueue_tail = [&](int on_process) -> int {
    int64_t swap_stride = ps->swap_stride;
    if (on_process > swap_stride)
      swap_stride = (int64_t)(ps->padding_amount * on_process);
    else if (on_process < ps->minimize_size * swap_stride)
      swap_stride = (int64_t)(ps->padding_amount * on_process);
    if (swap_stride < on_process) swap_stride = on_process;
    int rc = alloc_members(ps, ps->swap, swap_stride, false);
    if (rc) return rc;
    new_stride = swap_stride;
    PP_HIP_CHECK(ps->s_rowstart.reserve(sizeof(int) * ((size_t)ne + 1)));
    PP_HIP_CHECK(hipMemcpyAsync(ps->s_rowstart.p, ps->s_offsets2.p, sizeof(int) * ((size_t)ne + 1),
                                hipMemcpyDeviceToDevice, st));
    MoveArgs mv = make_move(ps, ps->data, ps->stride, ps->swap, swap_stride);
    WordTable wt{}, wt_new{};
    int NQ = 0;
    {
      const void* srcs[8];
      for (int m = 0; m < ps->nmembers; ++m) srcs[m] = ps->data[m].p;
      NQ = build_word_table(ps, srcs, ps->stride, swap_stride, -1, -1, wt);
      if (NQ > 0 && n_new > 0 && build_word_table(ps, new_info, n_new, swap_stride, -1, -1, wt_new) != NQ) NQ = 0;
    }
    if (NQ > 0 && on_process > 0) {
      PP_HIP_CHECK(ps->s_aos.reserve((size_t)on_process * NQ * 16));
      uint4* aos = ps->s_aos.as<uint4>();
      const int* off2 = ps->s_offsets2.as<int>();
      const bool no_defer = no_lazy_unpack();
      defer_wide = NQ > 4 && n_new == 0 && !no_defer;  // (see scs_rebuild)
      nq_used = NQ;
      PP_REQUIRE(!from_rec || NQ == ps->rec_nq, "rebuild (internal): the live records have another width");
#define PP_CSR_STAGED(N)                                                                         \
  case N:                                                                                        \
    if (nold > 0 && from_rec)                                                                    \
      k_move_pack_rec<N><<<grid_for(nold), kBlock, 0, st>>>(                                     \
          nold, rank, RankToSlot{new_element, off2, 1}, ps->s_aos_live.as<uint4>(), aos, go);    \
    else if (nold > 0)                                                                           \
      k_move_pack<N><<<grid_for(nold), kBlock, 0, st>>>(                                         \
          nold, rank, RankToSlot{new_element, off2, 1}, aos, wt, go);                            \
    if (n_new > 0)                                                                               \
      k_move_pack<N><<<grid_for(n_new), kBlock, 0, st>>>(                                        \
          n_new, rank_new, RankToSlot{new_elems, off2, 1}, aos, wt_new, go);                     \
    if (!defer_wide) k_unpack_flat<N><<<grid_for(on_process), kBlock, 0, st>>>(on_process, aos, wt, go); \
    break;
      switch (NQ) {
        PP_CSR_STAGED(1) PP_CSR_STAGED(2) PP_CSR_STAGED(3) PP_CSR_STAGED(4) PP_CSR_STAGED(6)
        PP_CSR_STAGED(8) PP_CSR_STAGED(10) PP_CSR_STAGED(12)
      }
#undef PP_CSR_STAGED
    } else {
      PP_REQUIRE(!from_rec, "rebuild (internal): live records but no staged path");
      if (nold > 0)
        k_move_csr<<<grid_for(nold), kBlock, 0, st>>>(nold, new_element, ps->s_rowstart.as<int>(), mv, go);
      if (n_new > 0) {
        MoveArgs add = mv;
        for (int m = 0; m < ps->nmembers; ++m) add.src[m] = new_info[m];
        add.src_stride = n_new;
        k_add_csr<<<grid_for(n_new), kBlock, 0, st>>>(n_new, new_elems, ps->s_rowstart.as<int>(), add, go);
      }
    }
    // slot -> element and mask of the NEW structure (still addressed through the swap side)
    const int new_cap = (int)swap_stride;
    PP_HIP_CHECK(ps->s_slot2.reserve(sizeof(int) * (size_t)std::max(new_cap, 1)));
    PP_HIP_CHECK(ps->s_mask2.reserve((size_t)std::max(new_cap, 1)));
    k_csr_slots<<<grid_for(((size_t)std::max(new_cap, 1) + 7) / 8 + 64), kBlock, 0, st>>>(
        ne, ps->s_offsets2.as<int>(), new_cap, ps->s_slot2.as<int>(), ps->s_mask2.as<unsigned char>(), go);
    return PP_OK;
  };
  // Speculation (see scs_rebuild): without deletions the live count is nold + n_new; the tail is
  // enqueued for that count and a one-thread kernel clears tot->go when the true count differs.
  const int guess = nold + n_new;
  bool speculated = false;
  static const bool no_spec = PP_LAB_ENV("PP_NO_SPEC_REBUILD") != nullptr;
  if (!no_spec && guess > 0) {
    k_spec_check_csr<<<1, 1, 0, st>>>(tot, guess);
    int rc = enqueue_tail(guess);
    if (rc) return rc;
    speculated = true;
  }
  Totals* h_pin = nullptr;
  hipEvent_t ev_tot = nullptr;
  if (int rcp = totals_pin(ps, &h_pin, &ev_tot)) return rcp;
  PP_HIP_CHECK(hipMemcpyAsync(h_pin, tot, sizeof(Totals), hipMemcpyDeviceToHost, st));
  PP_HIP_CHECK(hipEventRecord(ev_tot, st));
  PP_HIP_CHECK(hipEventSynchronize(ev_tot));
  const Totals h = *h_pin;
  if (h.invalid) {
    pp::set_error("rebuild: new element id out of range");
    return PP_EINVAL;
  }
  const int on_process = h.active;
  if (!(speculated && h.go)) {
    const int one = 1;
    PP_HIP_CHECK(hipMemcpyAsync(&tot->go, &one, sizeof(int), hipMemcpyHostToDevice, st));
    int rc = enqueue_tail(on_process);
    if (rc) return rc;
  }
  ps->swap_stride = new_stride;
  ps->data.swap(ps->swap);
  std::swap(ps->stride, ps->swap_stride);
  ps->d_offsets.swap(ps->s_offsets2);
  ps->d_slot_elem.swap(ps->s_slot2);
  ps->d_mask.swap(ps->s_mask2);
  ps->capacity = (int)ps->stride;
  ps->num_ptcls = on_process;
  ps->lazy_rec = 0;
  ++ps->n_full_rebuilds;
  if (from_rec) ++ps->n_from_records;
  if (defer_wide && on_process > 0) {  // the records of pass 1 are the particle data until a member is asked for
    ps->s_aos.swap(ps->s_aos_live);
    ps->lazy_rec = 3;
    ps->rec_nq = nq_used;
    ps->rec_rm = false;
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // namespace

namespace pp {
int scan_excl_i32(DevBuf& scratch, int n, const int* in, int* out, int* total_dev) {
  return scan_excl(scratch, n, in, out, total_dev, pp::stream());
}
const int* slot_elem(const pp_ps* ps) {
  ps->slot_elem_used = true;
  if (!ps->slot_elem_valid) {
    if (ps->kind == PP_SCS && ps->capacity > 0 && ps->ntiles_max > 0)
      k_fill_slot_elem<<<grid_for((size_t)ps->ntiles_max * ps->C), kBlock, 0, pp::stream()>>>(
          ps->d_ntiles.as<int>(), ps->C, ps->tile_p, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(),
          ps->d_chunk_width.as<int>(), ps->d_row_to_element.as<int>(), ps->d_slot_elem.as<int>());
    ps->slot_elem_valid = true;
  }
  return ps->d_slot_elem.as<int>();
}
const int* group_chunk(const pp_ps* ps) {
  if (ps->kind != PP_SCS || ps->C != 64 || ps->capacity <= 0 || ps->ntiles_max <= 0) return nullptr;
  if (!ps->group_chunk_valid) {
    if (ps->d_group_chunk.reserve(sizeof(int) * ((size_t)ps->capacity / 64 + 1)) != hipSuccess) return nullptr;
    k_fill_group_chunk<<<grid_for((size_t)ps->ntiles_max), kBlock, 0, pp::stream()>>>(
        ps->d_ntiles.as<int>(), ps->tile_p, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(),
        ps->d_chunk_width.as<int>(), ps->d_group_chunk.as<int>());
    ps->group_chunk_valid = true;
  }
  return ps->d_group_chunk.as<int>();
}
bool lazy_push_ok(const pp_ps* ps, int m_x, int m_xtgt, int m_b, int m_phi) {
  if (!ps || ps->lazy_rec != 1 || ps->kind != PP_SCS || !xgcm_shape(ps)) return false;
  for (int m = 0; m < 5; ++m)
    if (ps->member_map[m] != m) return false;
  return m_x == 0 && m_xtgt == 1 && m_b == 3 && m_phi == 4 && ps->lazy_x == 0 && ps->lazy_xt == 1;
}
int ps_materialize(pp_ps* ps) {
  if (ps->lazy_rec == 3) {
    // every member of every live particle is in the rec_nq-quad records of the last re-layout (slot order)
    const int NQ = ps->rec_nq;
    ps->lazy_rec = 0;
    if (ps->capacity > 0 && ps->num_ptcls > 0) {
      WordTable wt{};
      for (int m = 0; m < ps->nmembers; ++m) {  // (the order of build_word_table)
        const int b = ps->member_bytes[m];
        for (int cc = 0; cc < ps->member_ncomp[m]; ++cc) {
          char* dst = (char*)ps->data[m].p + ((size_t)cc * ps->stride) * b;
          if (b == 8)
            wt.dst8[wt.n8++] = dst;
          else
            wt.dst4[wt.n4++] = dst;
        }
      }
      static int* const go_one3 = [] {
        int* p1 = nullptr;
        const int one = 1;
        if (hipMalloc((void**)&p1, sizeof(int)) != hipSuccess ||
            hipMemcpy(p1, &one, sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
          return (int*)nullptr;
        return p1;
      }();
      PP_REQUIRE(go_one3 != nullptr, "ps_materialize: device allocation failed");
      if (ps->kind == PP_SCS) {
#define PP_MAT(N)                                                                                          \
  case N:                                                                                                  \
    k_move_unpack<N><<<grid_for((size_t)ps->ntiles_max * ps->C), kBlock, 0, pp::stream()>>>(               \
        ps->d_ntiles.as<int>(), ps->C, ps->tile_p, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(),     \
        ps->d_chunk_width.as<int>(), ps->d_mask.as<unsigned char>(), ps->s_aos_live.as<uint4>(), wt, go_one3, 0); \
    break;
        switch (NQ) { PP_MAT(6) PP_MAT(8) PP_MAT(10) PP_MAT(12) default: PP_REQUIRE(false, "ps_materialize: record width"); }
#undef PP_MAT
      } else {
#define PP_MAT(N)                                                                                          \
  case N:                                                                                                  \
    k_unpack_flat<N><<<grid_for(ps->num_ptcls), kBlock, 0, pp::stream()>>>(ps->num_ptcls, ps->s_aos_live.as<uint4>(), \
                                                                           wt, go_one3);                   \
    break;
        switch (NQ) { PP_MAT(6) PP_MAT(8) PP_MAT(10) PP_MAT(12) default: PP_REQUIRE(false, "ps_materialize: record width"); }
#undef PP_MAT
      }
      PP_LAUNCH_CHECK();
    }
    return ps_zeros(ps);
  }
  if (ps->lazy_rec) {
    // the deferred second pass of the last full re-layout: 32-B records (+ the side word) -> SoA arrays, for every
    // member that is still only in the records (all that travelled / the origin only)
    const int state = ps->lazy_rec;
    ps->lazy_rec = 0;
    const bool split = ps->rec_split;  // (x, y) and (pad, phi, b, id) in two arrays (scs_rebuild: rec_split)
    ps->rec_split = false;
    if (ps->capacity > 0 && ps->num_ptcls > 0) {
      WordTable wt{};
      for (int m = 0; m < ps->nmembers; ++m) {
        if (m == ps->lazy_xt) continue;  // logically zero (zero_pending)
        const int b = ps->member_bytes[m];
        for (int cc = 0; cc < ps->member_ncomp[m]; ++cc) {
          char* dst = (state == 2 && m != ps->lazy_x) ? nullptr : (char*)ps->data[m].p + ((size_t)cc * ps->stride) * b;
          if (split && m == ps->lazy_x && cc == 2) {  // (split 2-D records carry no third component: it is zero)
            PP_HIP_CHECK(hipMemsetAsync(dst, 0, (size_t)ps->stride * b, pp::stream()));
            continue;
          }
          if (m == 2 && !split)  // (the order of build_word_table with side_member 2)
            wt.side_dst = dst;
          else if (b == 8)
            wt.dst8[wt.n8++] = dst;
          else
            wt.dst4[wt.n4++] = dst;
        }
      }
      static int* const go_one = [] {  // a device 1 (initialised once, thread-safe: function-local static)
        int* p1 = nullptr;
        const int one = 1;
        if (hipMalloc((void**)&p1, sizeof(int)) != hipSuccess ||
            hipMemcpy(p1, &one, sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
          return (int*)nullptr;
        return p1;
      }();
      PP_REQUIRE(go_one != nullptr, "ps_materialize: device allocation failed");
      if (state == 2) wt.n4 = 0;  // (4-byte members were written by the push; 8-byte ones: only lazy_x has a target)
      k_move_unpack<2><<<grid_for((size_t)ps->ntiles_max * ps->C), kBlock, 0, pp::stream()>>>(
          ps->d_ntiles.as<int>(), ps->C, ps->tile_p, ps->d_tiles.as<int>(), ps->d_chunk_start.as<int>(),
          ps->d_chunk_width.as<int>(), ps->d_mask.as<unsigned char>(), ps->s_aos_live.as<uint4>(), wt, go_one,
          ps->rec_rm ? 1 : 0, (state == 2 || split) ? nullptr : ps->s_side_live.as<unsigned>(),
          split ? ps->s_side_live.as<uint4>() : nullptr);
      PP_LAUNCH_CHECK();
    }
  }
  return ps_zeros(ps);
}
int ps_zeros(pp_ps* ps) {
  if (ps->zero_z_pending) {  // the third component of x_tgt after a 2-D record-fed push
    ps->zero_z_pending = false;
    const int z = ps->lazy_xt;
    if (z >= 0 && z < ps->nmembers && ps->member_ncomp[z] == 3 && ps->stride > 0) {
      const size_t plane = (size_t)ps->stride * ps->member_bytes[z];
      PP_HIP_CHECK(hipMemsetAsync((char*)ps->data[z].p + 2 * plane, 0, plane, pp::stream()));
    }
  }
  const int s = ps->zero_pending;
  if (s < 0) return PP_OK;
  ps->zero_pending = -1;
  const size_t bytes = (size_t)ps->stride * ps->member_ncomp[s] * ps->member_bytes[s];
  if (bytes) PP_HIP_CHECK(hipMemsetAsync(ps->data[s].p, 0, bytes, pp::stream()));
  return PP_OK;
}
}  // namespace pp

extern "C" {

// element gids of a new structure: the device copy, and the gid -> element table when they are not 0..ne-1
static bool store_gids(pp_ps* ps, const int64_t* gids_host, int num_elems) {
  ps->has_gids = true;
  bool ok = ps->d_gids.reserve(sizeof(int64_t) * (size_t)num_elems) == hipSuccess &&
            hipMemcpy(ps->d_gids.p, gids_host, sizeof(int64_t) * (size_t)num_elems, hipMemcpyHostToDevice) == hipSuccess;
  int64_t mx = -1;
  ps->gids_identity = true;
  for (int e = 0; e < num_elems; ++e) {
    if (gids_host[e] != e) ps->gids_identity = false;
    mx = std::max<int64_t>(mx, gids_host[e]);
  }
  if (!ok || ps->gids_identity) return ok;
  if (mx >= (int64_t)1 << 30) return ok;  // too sparse for a dense table: the caller passes its own map
  std::vector<int> tab((size_t)mx + 1, -1);
  for (int e = 0; e < num_elems; ++e)
    if (gids_host[e] >= 0) tab[(size_t)gids_host[e]] = e;
  ps->n_gid2lid = mx + 1;
  return ps->d_gid2lid.reserve(sizeof(int) * tab.size()) == hipSuccess &&
         hipMemcpy(ps->d_gid2lid.p, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) == hipSuccess;
}

pp_ps* pp_ps_create_scs(int C, int sigma, int V, int num_elems, int num_ptcls,
                        const int* ppe_host, const int64_t* gids_host, int pad_strat,
                        double shuffle_padding, double extra_padding, int nmembers,
                        const int* member_bytes, const int* member_ncomp,
                        const int* particle_elements_host, const void* const* particle_info_host) {
  if (C < 1 || V < 1 || num_elems < 0 || num_ptcls < 0 || (!ppe_host && num_elems > 0)) {
    pp::set_error("pp_ps_create_scs: bad arguments");
    return nullptr;
  }
  if (!pp::initialised() && pp_init(0) != PP_OK) return nullptr;
  pp_ps* ps = new pp_ps();
  ps->kind = PP_SCS;
  if (set_members(ps, nmembers, member_bytes, member_ncomp) != PP_OK) {
    delete ps;
    return nullptr;
  }
  long total = 0;
  for (int e = 0; e < num_elems; ++e) total += ppe_host[e];
  if (total != num_ptcls) {
    pp::set_error("pp_ps_create_scs: sum(ppe) != num_ptcls");
    delete ps;
    return nullptr;
  }
  ps->num_elems = num_elems;
  ps->num_ptcls = num_ptcls;
  ps->C_max = C;
  ps->V = V;
  ps->sigma = sigma;
  ps->pad_strat = pad_strat;
  ps->shuffle_padding = shuffle_padding;
  ps->extra_padding = extra_padding;
  ps->C = choose_chunk_height(C, ppe_host, num_elems);
  HostLayout L;
  host_layout(L, ps->C, V, sigma, num_elems, ppe_host, pad_strat, shuffle_padding);
  ps->num_chunks = L.nchunks;
  ps->num_rows = L.nchunks * L.C;
  ps->num_slices = L.nslices;
  ps->capacity = L.capacity;
  ps->num_empty_elements = L.num_empty;
  int64_t cap = L.capacity;
  if (extra_padding > 0) cap = (int64_t)(int)(L.capacity * (1 + extra_padding));
  ps->stride = spread_stride(std::max<int64_t>(cap, 1));
  ps->swap_stride = ps->stride;  // the reference allocates an equal-sized swap at construction
  bool ok = alloc_members(ps, ps->data, ps->stride, true) == PP_OK;
  std::vector<int> ppe(ppe_host, ppe_host + num_elems);
  ok = ok && finish_layout_upload(ps, L, ppe) == PP_OK;
  if (ok && gids_host && num_elems > 0) ok = store_gids(ps, gids_host, num_elems);
  if (ok) {
    ok = upload_vec(ps->d_elem_count, ppe) == PP_OK && hipStreamSynchronize(pp::stream()) == hipSuccess;
    ps->elem_count_valid = ok;
    ps->version = pp::next_version();
  }
  if (ok && num_ptcls > 0 && particle_elements_host && particle_info_host) {
    // initSCSData (SCS_buildFns.h:205-232) in particle order
    std::vector<int> row_index((size_t)ps->num_rows);
    for (int i = 0; i < ps->num_rows; ++i) row_index[i] = L.chunk_start[i / L.C] + i % L.C;
    std::vector<int> slot((size_t)num_ptcls);
    for (int i = 0; i < num_ptcls; ++i) {
      const int e = particle_elements_host[i];
      if (e < 0 || e >= num_elems) {
        pp::set_error("pp_ps_create_scs: particle element out of range");
        ok = false;
        break;
      }
      const int row = L.element_to_row[e];
      slot[i] = row_index[row];
      row_index[row] += L.C;
    }
    ok = ok && upload_initial(ps, slot, num_ptcls, particle_info_host) == PP_OK;
  }
  if (!ok) {
    delete ps;
    return nullptr;
  }
  return ps;
}

pp_ps* pp_ps_create_csr(int num_elems, int num_ptcls, const int* ppe_host,
                        const int64_t* gids_host, double padding_amount, int nmembers,
                        const int* member_bytes, const int* member_ncomp,
                        const int* particle_elements_host, const void* const* particle_info_host) {
  if (num_elems < 0 || num_ptcls < 0 || (!ppe_host && num_elems > 0)) {
    pp::set_error("pp_ps_create_csr: bad arguments");
    return nullptr;
  }
  if (!pp::initialised() && pp_init(0) != PP_OK) return nullptr;
  pp_ps* ps = new pp_ps();
  ps->kind = PP_CSR;
  if (set_members(ps, nmembers, member_bytes, member_ncomp) != PP_OK) {
    delete ps;
    return nullptr;
  }
  ps->num_elems = num_elems;
  ps->num_rows = num_elems;
  ps->num_ptcls = num_ptcls;
  ps->padding_amount = padding_amount;
  std::vector<int> offsets((size_t)num_elems + 1, 0);
  for (int e = 0; e < num_elems; ++e) offsets[e + 1] = offsets[e] + ppe_host[e];
  if (offsets[num_elems] != num_ptcls) {
    pp::set_error("pp_ps_create_csr: sum(ppe) != num_ptcls");
    delete ps;
    return nullptr;
  }
  ps->capacity = (int)(offsets[num_elems] * padding_amount);
  ps->stride = std::max<int64_t>(ps->capacity, 1);
  ps->swap_stride = ps->capacity;
  bool ok = alloc_members(ps, ps->data, ps->stride, true) == PP_OK;
  ok = ok && upload_vec(ps->d_offsets, offsets) == PP_OK;
  std::vector<int> slot_elem((size_t)ps->capacity, -1);
  std::vector<unsigned char> mask((size_t)ps->capacity, 0);
  for (int e = 0; e < num_elems; ++e)
    for (int j = offsets[e]; j < offsets[e + 1]; ++j) {
      slot_elem[j] = e;
      mask[j] = 1;
    }
  ok = ok && upload_vec(ps->d_slot_elem, slot_elem) == PP_OK && upload_vec(ps->d_mask, mask) == PP_OK;
  ps->slot_elem_valid = true;
  ps->group_chunk_valid = false;
  if (ok) ok = hipStreamSynchronize(pp::stream()) == hipSuccess;
  if (ok && gids_host && num_elems > 0) ok = store_gids(ps, gids_host, num_elems);
  if (ok && num_ptcls > 0 && particle_elements_host && particle_info_host) {
    std::vector<int> row(offsets.begin(), offsets.end());
    std::vector<int> slot((size_t)num_ptcls);
    for (int i = 0; i < num_ptcls; ++i) {
      const int e = particle_elements_host[i];
      if (e < 0 || e >= num_elems) {
        pp::set_error("pp_ps_create_csr: particle element out of range");
        ok = false;
        break;
      }
      slot[i] = row[e]++;
    }
    ok = ok && upload_initial(ps, slot, num_ptcls, particle_info_host) == PP_OK;
  }
  if (!ok) {
    delete ps;
    return nullptr;
  }
  return ps;
}

// SellCSigma::copy<MSpace> / CSR::copy (scs/SellCSigma.h:336-391): a deep copy of the structure -- the same layout
// arrays, the same slots, every member -- as a new, independent structure (pending passes of the source run first)
pp_ps* pp_ps_clone(pp_ps* src) {
  if (!src) {
    pp::set_error("pp_ps_clone: null structure");
    return nullptr;
  }
  if (pp::ps_ready(src) != PP_OK) return nullptr;
  (void)pp::slot_elem(src);
  hipStream_t st = pp::stream();
  pp_ps* n = new pp_ps();
  auto dup = [&](pp::DevBuf& dst, const pp::DevBuf& from) -> bool {
    if (!from.p || from.bytes == 0) return true;
    if (dst.reserve(from.bytes) != hipSuccess) return false;
    return hipMemcpyAsync(dst.p, from.p, from.bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
  };
  n->kind = src->kind;
  n->num_elems = src->num_elems;
  n->num_ptcls = src->num_ptcls;
  n->capacity = src->capacity;
  n->num_rows = src->num_rows;
  n->C = src->C;
  n->C_max = src->C_max;
  n->V = src->V;
  n->sigma = src->sigma;
  n->num_chunks = src->num_chunks;
  n->num_slices = src->num_slices;
  n->pad_strat = src->pad_strat;
  n->shuffle_padding = src->shuffle_padding;
  n->extra_padding = src->extra_padding;
  n->minimize_size = src->minimize_size;
  n->padding_amount = src->padding_amount;
  n->num_empty_elements = src->num_empty_elements;
  n->nmembers = src->nmembers;
  n->member_bytes = src->member_bytes;
  n->member_ncomp = src->member_ncomp;
  n->member_map = src->member_map;
  n->stride = src->stride;
  n->has_gids = src->has_gids;
  n->gids_identity = src->gids_identity;
  n->n_gid2lid = src->n_gid2lid;
  n->ntiles_max = src->ntiles_max;
  n->tile_p = src->tile_p;
  n->shuffle_mode = src->shuffle_mode;
  n->elem_count_valid = src->elem_count_valid;
  n->slot_elem_valid = true;
  n->version = pp::next_version();
  bool ok = true;
  n->data.resize(src->data.size());
  n->swap.resize(src->swap.size());
  for (size_t m = 0; m < src->data.size(); ++m) ok = ok && dup(n->data[m], src->data[m]);
  ok = ok && dup(n->d_gids, src->d_gids) && dup(n->d_gid2lid, src->d_gid2lid) && dup(n->d_offsets, src->d_offsets) &&
       dup(n->d_slice_to_chunk, src->d_slice_to_chunk) && dup(n->d_row_to_element, src->d_row_to_element) &&
       dup(n->d_element_to_row, src->d_element_to_row) && dup(n->d_mask, src->d_mask) &&
       dup(n->d_slot_elem, src->d_slot_elem) && dup(n->d_chunk_start, src->d_chunk_start) &&
       dup(n->d_chunk_width, src->d_chunk_width) && dup(n->d_tiles, src->d_tiles) && dup(n->d_ntiles, src->d_ntiles) &&
       dup(n->d_elem_count, src->d_elem_count) && dup(n->d_eslot0, src->d_eslot0);
  if (!ok || hipStreamSynchronize(st) != hipSuccess) {
    pp::set_error("pp_ps_clone: device allocation or copy failed");
    delete n;
    return nullptr;
  }
  return n;
}

int pp_ps_destroy(pp_ps* ps) {
  if (ps) {
    // (queued kernels may still read the structure's buffers; the search's counter sets die with it)
    (void)hipStreamSynchronize(pp::stream());
    pp::search_counters_released(ps->cnt2);
  }
  delete ps;
  return PP_OK;
}

int pp_ps_info(const pp_ps* ps, pp_ps_info_t* out) {
  PP_REQUIRE(ps && out, "pp_ps_info: null argument");
  out->kind = ps->kind;
  out->num_elems = ps->num_elems;
  out->num_ptcls = ps->num_ptcls;
  out->capacity = ps->capacity;
  out->num_rows = ps->num_rows;
  out->C = ps->C;
  out->V = ps->V;
  out->sigma = ps->sigma;
  out->num_chunks = ps->num_chunks;
  out->num_slices = ps->num_slices;
  out->nmembers = ps->nmembers;
  out->stride = ps->stride;
  return PP_OK;
}

void* pp_ps_member_ptr(pp_ps* ps, int m) {
  if (!ps || m < 0 || m >= ps->nmembers) {
    pp::set_error("pp_ps_member_ptr: bad member index");
    return nullptr;
  }
  if (pp::ps_ready(ps)) return nullptr;
  return ps->data[ps->member_map[m]].p;
}
int64_t pp_ps_member_stride(const pp_ps* ps) { return ps ? ps->stride : 0; }

int pp_ps_swap_members(pp_ps* ps, int a, int b) {
  PP_REQUIRE(ps && a >= 0 && b >= 0 && a < ps->nmembers && b < ps->nmembers,
             "pp_ps_swap_members: bad member index");
  if (int rc = pp::ps_ready(ps)) return rc;  // (deferred passes are keyed on storage indices)
  const int sa = ps->member_map[a], sb = ps->member_map[b];
  PP_REQUIRE(ps->member_bytes[sa] == ps->member_bytes[sb] &&
                 ps->member_ncomp[sa] == ps->member_ncomp[sb],
             "pp_ps_swap_members: members differ in type");
  std::swap(ps->member_map[a], ps->member_map[b]);
  return PP_OK;
}

int pp_ps_set_shuffling(pp_ps* ps, int mode) {
  PP_REQUIRE(ps && (mode == 0 || mode == 1), "pp_ps_set_shuffling: mode must be 0 or 1");
  ps->shuffle_mode = mode;
  return PP_OK;
}
int pp_ps_rebuild_stats(const pp_ps* ps, long long* n_in_place, long long* n_full, long long* n_from_records) {
  PP_REQUIRE(ps, "pp_ps_rebuild_stats: null ps");
  if (n_in_place) *n_in_place = ps->n_reshuffles;
  if (n_full) *n_full = ps->n_full_rebuilds;
  if (n_from_records) *n_from_records = ps->n_from_records;
  return PP_OK;
}

int pp_ps_deferred_state(const pp_ps* ps, pp_ps_deferred_t* out) {
  PP_REQUIRE(ps && out, "pp_ps_deferred_state: null argument");
  out->lazy_rec = ps->lazy_rec;
  out->zero_pending = ps->zero_pending;
  out->zero_z_pending = ps->zero_z_pending ? 1 : 0;
  out->elem_count_valid = ps->elem_count_valid ? 1 : 0;
  out->slot_elem_valid = ps->slot_elem_valid ? 1 : 0;
  out->hot_row = ps->hot.on ? 1 : 0;
  return PP_OK;
}
int pp_ps_materialize(pp_ps* ps) {
  PP_REQUIRE(ps, "pp_ps_materialize: null ps");
  if (int rc = pp::ps_ready(ps)) return rc;
  (void)pp::slot_elem(ps);
  (void)pp::group_chunk(ps);
  return PP_OK;
}

int pp_ps_layout(const pp_ps* ps, pp_ps_layout_t* out) {
  PP_REQUIRE(ps && out, "pp_ps_layout: null argument");
  out->offsets = ps->d_offsets.as<int>();
  out->slice_to_chunk = ps->d_slice_to_chunk.as<int>();
  out->row_to_element = ps->d_row_to_element.as<int>();
  out->element_to_row = ps->d_element_to_row.as<int>();
  out->mask = ps->d_mask.as<unsigned char>();
  out->slot_elem = pp::slot_elem(ps);
  return PP_OK;
}

int pp_ps_iteration(const pp_ps* ps, pp_ps_iter_t* out) {
  PP_REQUIRE(ps && out, "pp_ps_iteration: null argument");
  out->capacity = ps->capacity;
  out->mask = ps->d_mask.as<unsigned char>();
  out->group_chunk = pp::group_chunk(ps);
  out->row_to_element = ps->d_row_to_element.as<int>();
  out->slot_elem = out->group_chunk ? nullptr : pp::slot_elem(ps);
  return PP_OK;
}

int pp_ps_layout_to_host(const pp_ps* ps, int* offsets, int* slice_to_chunk, int* row_to_element,
                         int* element_to_row, unsigned char* mask, int* slot_elem) {
  PP_REQUIRE(ps, "pp_ps_layout_to_host: null ps");
  if (slot_elem) (void)pp::slot_elem(ps);
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  const size_t noff = (ps->kind == PP_SCS) ? (size_t)ps->num_slices + 1 : (size_t)ps->num_elems + 1;
  if (offsets) PP_HIP_CHECK(hipMemcpy(offsets, ps->d_offsets.p, noff * sizeof(int), hipMemcpyDeviceToHost));
  if (ps->kind == PP_SCS) {
    if (slice_to_chunk && ps->num_slices)
      PP_HIP_CHECK(hipMemcpy(slice_to_chunk, ps->d_slice_to_chunk.p,
                             (size_t)ps->num_slices * sizeof(int), hipMemcpyDeviceToHost));
    if (row_to_element && ps->num_rows)
      PP_HIP_CHECK(hipMemcpy(row_to_element, ps->d_row_to_element.p,
                             (size_t)ps->num_rows * sizeof(int), hipMemcpyDeviceToHost));
    if (element_to_row && ps->num_rows)
      PP_HIP_CHECK(hipMemcpy(element_to_row, ps->d_element_to_row.p,
                             (size_t)ps->num_rows * sizeof(int), hipMemcpyDeviceToHost));
  }
  if (mask && ps->capacity)
    PP_HIP_CHECK(hipMemcpy(mask, ps->d_mask.p, (size_t)ps->capacity, hipMemcpyDeviceToHost));
  if (slot_elem && ps->capacity)
    PP_HIP_CHECK(hipMemcpy(slot_elem, pp::slot_elem(ps), (size_t)ps->capacity * sizeof(int),
                           hipMemcpyDeviceToHost));
  return PP_OK;
}

int pp_ps_last_search_found(const pp_ps* ps, int* found) {
  PP_REQUIRE(ps && found, "pp_ps_last_search_found: null argument");
  if (ps->search_nf >= 0) {
    *found = ps->search_nf == 0;
    return PP_OK;
  }
  int nf = 0;  // (the rebuild did not carry it: in place, CSR, separate-kernel layout) one host sync
  if (ps->last_nf_dev) {  // the structure's own counter set
    PP_HIP_CHECK(hipMemcpyAsync(&nf, ps->last_nf_dev, sizeof(int), hipMemcpyDeviceToHost, pp::stream()));
    PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  } else {
    // the process-wide set: it is this structure's search only if nothing was searched since
    if (ps->searched_serial == 0) {  // never searched with pp_push_search: nothing was cut off
      *found = 1;
      return PP_OK;
    }
    if (ps->searched_serial != pp::search_serial()) {
      pp::set_error("pp_ps_last_search_found: another structure was searched since this one (CSR structures share one "
                    "counter set) -- pass `found` to pp_push_search instead");
      return PP_ESTATE;
    }
    if (int rc = pp_push_search_counters(&nf, nullptr, nullptr)) return rc;
  }
  *found = nf == 0;
  return PP_OK;
}

int pp_ps_gids_to_host(const pp_ps* ps, int64_t* out_host) {
  PP_REQUIRE(ps && out_host, "pp_ps_gids_to_host: null argument");
  if (!ps->has_gids || ps->num_elems == 0) return 0;
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  PP_HIP_CHECK(hipMemcpy(out_host, ps->d_gids.p, sizeof(int64_t) * (size_t)ps->num_elems, hipMemcpyDeviceToHost));
  return ps->num_elems;
}

int pp_ps_member_to_host(pp_ps* ps, int m, void* out_host) {
  PP_REQUIRE(ps && out_host && m >= 0 && m < ps->nmembers, "pp_ps_member_to_host: bad argument");
  if (int rc = pp::ps_ready(ps)) return rc;
  const int s = ps->member_map[m];
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  PP_HIP_CHECK(hipMemcpy(out_host, ps->data[s].p,
                         (size_t)ps->stride * ps->member_ncomp[s] * ps->member_bytes[s],
                         hipMemcpyDeviceToHost));
  return PP_OK;
}
int pp_ps_member_from_host(pp_ps* ps, int m, const void* in_host) {
  PP_REQUIRE(ps && in_host && m >= 0 && m < ps->nmembers, "pp_ps_member_from_host: bad argument");
  if (int rc = pp::ps_ready(ps)) return rc;
  const int s = ps->member_map[m];
  PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  PP_HIP_CHECK(hipMemcpy(ps->data[s].p, in_host,
                         (size_t)ps->stride * ps->member_ncomp[s] * ps->member_bytes[s],
                         hipMemcpyHostToDevice));
  return PP_OK;
}

int pp_ps_rebuild(pp_ps* ps, const int* new_element_dev, int n_new, const int* new_elems_dev,
                  const void* const* new_info_dev) {
  pp::Range rg_("ps_rebuild");
  PP_REQUIRE(ps && (new_element_dev || ps->capacity == 0), "pp_ps_rebuild: null argument");
  PP_REQUIRE(n_new >= 0 && (n_new == 0 || new_elems_dev), "pp_ps_rebuild: bad new particles");
  // (wide records left by the previous rebuild -- lazy_rec == 3 -- feed this one's first pass: nothing to materialise)
  if (!(ps->lazy_rec == 3 && ps->zero_pending < 0))
    if (int rc = pp::ps_ready(ps)) return rc;
  // storage order of members may be permuted by pp_ps_swap_members: normalise first
  for (int m = 0; m < ps->nmembers; ++m)
    if (ps->member_map[m] != m) {
      // apply the permutation to the buffers so that logical == storage again
      std::vector<pp::DevBuf> tmp((size_t)ps->nmembers);
      for (int q = 0; q < ps->nmembers; ++q) tmp[q].swap(ps->data[ps->member_map[q]]);
      for (int q = 0; q < ps->nmembers; ++q) ps->data[q].swap(tmp[q]);
      std::iota(ps->member_map.begin(), ps->member_map.end(), 0);
      break;
    }
  if (ps->kind == PP_SCS)
    return scs_rebuild(ps, new_element_dev, n_new, new_elems_dev, new_info_dev, -1, -1);
  ps->elem_count_valid = false;
  ps->version = pp::next_version();
  return csr_rebuild(ps, new_element_dev, n_new, new_elems_dev, new_info_dev);
}

int pp_ps_rebuild_scatter(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_new,
                          const int* new_elems_dev, const void* const* new_info_dev,
                          const pp_mesh* mesh, int nmaps, const int* const* v2v_dev,
                          double* const* scatter_w_dev, double rmax, int gnr, int gppr) {
  return pp::ps_rebuild_scatter(ps, m_x, m_xtgt, new_element_dev, n_new, new_elems_dev, new_info_dev, mesh, nmaps,
                                v2v_dev, scatter_w_dev, rmax, gnr, gppr, false);
}
}  // extern "C"
int pp::ps_rebuild_scatter(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_new,
                           const int* new_elems_dev, const void* const* new_info_dev, const pp_mesh* mesh, int nmaps,
                           const int* const* v2v_dev, double* const* scatter_w_dev, double rmax, int gnr, int gppr,
                           bool new_xt_zero) {
  PP_REQUIRE(ps && nmaps >= 0 && (nmaps == 0 || (mesh && v2v_dev && scatter_w_dev)),
             "pp_ps_rebuild_scatter: null argument");
  PP_REQUIRE(nmaps == 0 || ps->num_elems == mesh->nelems,
             "pp_ps_rebuild_scatter: structure/mesh element mismatch");
  PP_REQUIRE(gnr >= 2 && gppr > 0, "pp_ps_rebuild_scatter: needs gnr >= 2 (ringUp < gnr, gyroScatter.hpp:190)");
  for (int k = 0; k < nmaps; ++k)
    PP_REQUIRE(v2v_dev[k] && scatter_w_dev[k], "pp_ps_rebuild_scatter: null map / output");
  const bool commit = m_x >= 0 || m_xtgt >= 0;
  if (ps->kind != PP_SCS) {  // CSR: the reference calls back to back
    int rc = commit ? pp_ps_rebuild_commit(ps, m_x, m_xtgt, new_element_dev, n_new, new_elems_dev, new_info_dev)
                    : pp_ps_rebuild(ps, new_element_dev, n_new, new_elems_dev, new_info_dev);
    for (int k = 0; k < nmaps && !rc; ++k)
      rc = pp_gyro_scatter(mesh, ps, v2v_dev[k], rmax, gnr, gppr, scatter_w_dev[k]);
    return rc;
  }
  if (commit) {
    PP_REQUIRE(m_x >= 0 && m_xtgt >= 0 && m_x < ps->nmembers && m_xtgt < ps->nmembers && m_x != m_xtgt,
               "pp_ps_rebuild_scatter: bad member index");
    PP_REQUIRE(ps->member_bytes[m_x] == 8 && ps->member_bytes[m_xtgt] == 8 &&
                   ps->member_ncomp[m_x] == ps->member_ncomp[m_xtgt],
               "pp_ps_rebuild_scatter: x and x_tgt must be double members of equal shape");
  }
  PP_REQUIRE(new_element_dev || ps->capacity == 0, "pp_ps_rebuild_scatter: null new_element");
  PP_REQUIRE(n_new >= 0 && (n_new == 0 || new_elems_dev), "pp_ps_rebuild_scatter: bad new particles");
  for (int m = 0; m < ps->nmembers; ++m)
    if (ps->member_map[m] != m) {  // normalise a pending pp_ps_swap_members permutation
      std::vector<pp::DevBuf> tmp((size_t)ps->nmembers);
      for (int q = 0; q < ps->nmembers; ++q) tmp[q].swap(ps->data[ps->member_map[q]]);
      for (int q = 0; q < ps->nmembers; ++q) ps->data[q].swap(tmp[q]);
      std::iota(ps->member_map.begin(), ps->member_map.end(), 0);
      break;
    }
  auto scatter = [&](const int* counts) -> int {
    return pp::gyro_scatter_counts(mesh, counts, nmaps, v2v_dev, scatter_w_dev, rmax, gnr, gppr);
  };
  pp::GyroRide ride{};
  if (nmaps > 0)
    if (int rc = pp::gyro_scatter_ride(mesh, nmaps, v2v_dev, scatter_w_dev, rmax, gnr, gppr, &ride)) return rc;
  ps->ride = &ride;
  ps->ride_done = false;
  const int rc = scs_rebuild(ps, new_element_dev, n_new, new_elems_dev, new_info_dev, commit ? m_x : -1,
                             commit ? m_xtgt : -1, nmaps > 0 ? std::function<int(const int*)>(scatter)
                                                             : std::function<int(const int*)>(),
                             true, new_xt_zero && commit);
  ps->ride = nullptr;
  ps->ride_done = false;
  return rc;
}
extern "C" {

int pp_ps_rebuild_commit(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_new,
                         const int* new_elems_dev, const void* const* new_info_dev) {
  PP_REQUIRE(ps, "pp_ps_rebuild_commit: null ps");
  PP_REQUIRE(m_x >= 0 && m_xtgt >= 0 && m_x < ps->nmembers && m_xtgt < ps->nmembers && m_x != m_xtgt,
             "pp_ps_rebuild_commit: bad member index");
  if (ps->kind != PP_SCS) {  // CSR: the two reference steps back to back
    int rc = pp_update_positions(ps, m_x, m_xtgt);
    if (rc) return rc;
    return pp_ps_rebuild(ps, new_element_dev, n_new, new_elems_dev, new_info_dev);
  }
  // normalise a pending pp_ps_swap_members permutation, then fuse the commit into the move
  if (ps->member_map[m_x] != m_x || ps->member_map[m_xtgt] != m_xtgt) {
    std::vector<pp::DevBuf> tmp((size_t)ps->nmembers);
    for (int q = 0; q < ps->nmembers; ++q) tmp[q].swap(ps->data[ps->member_map[q]]);
    for (int q = 0; q < ps->nmembers; ++q) ps->data[q].swap(tmp[q]);
    std::iota(ps->member_map.begin(), ps->member_map.end(), 0);
  }
  PP_REQUIRE(ps->member_bytes[m_x] == 8 && ps->member_bytes[m_xtgt] == 8 &&
                 ps->member_ncomp[m_x] == ps->member_ncomp[m_xtgt],
             "pp_ps_rebuild_commit: x and x_tgt must be double members of equal shape");
  PP_REQUIRE(new_element_dev || ps->capacity == 0, "pp_ps_rebuild_commit: null new_element");
  return scs_rebuild(ps, new_element_dev, n_new, new_elems_dev, new_info_dev, m_x, m_xtgt);
}

int pp_ps_get_pids(const pp_ps* ps, int* offsets_dev, int* pids_dev) {
  PP_REQUIRE(ps && offsets_dev && pids_dev, "pp_ps_get_pids: null argument");
  hipStream_t st = pp::stream();
  const int ne = ps->num_elems;
  pp::DevBuf ppe, cur;
  PP_HIP_CHECK(ppe.reserve(sizeof(int) * ((size_t)ne + 1)));
  PP_HIP_CHECK(cur.reserve(sizeof(int) * ((size_t)ne + 1)));
  PP_HIP_CHECK(hipMemsetAsync(ppe.p, 0, sizeof(int) * ((size_t)ne + 1), st));
  PP_HIP_CHECK(hipMemsetAsync(cur.p, 0, sizeof(int) * ((size_t)ne + 1), st));
  if (ps->capacity > 0 && ps->num_ptcls > 0)
    k_pid_count<<<grid_for(ps->capacity), kBlock, 0, st>>>(
        ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), ppe.as<int>());
  k_scan_excl<<<1, 1024, 0, st>>>(ne + 1, ppe.as<int>(), offsets_dev, nullptr);
  if (ps->capacity > 0 && ps->num_ptcls > 0)
    k_pid_set<<<grid_for(ps->capacity), kBlock, 0, st>>>(
        ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), offsets_dev,
        cur.as<int>(), pids_dev);
  PP_LAUNCH_CHECK();
  PP_HIP_CHECK(hipStreamSynchronize(st));
  return PP_OK;
}

int pp_redistribute_particles_dist(const pp_ps* ps, int strat, double percent_moved, unsigned long long seed,
                                   int* new_elems_dev) {
  PP_REQUIRE(ps && new_elems_dev, "pp_redistribute_particles: null argument");
  PP_REQUIRE(percent_moved >= 0 && percent_moved <= 1, "pp_redistribute_particles: percentMoved in [0,1]");
  PP_REQUIRE(strat >= 1 && strat <= 4,
             "pp_redistribute_particles: strategy 1 (uniform), 2 (gaussian), 3 (exponential) or 4 (GITRm approximation)");
  if (ps->capacity == 0) return PP_OK;
  const int ne = ps->num_elems;
  const int *es = nullptr, *ee = nullptr;
  if (strat == 3) {  // the conversion's logarithms, once per element count, on the host (Distribute.cpp:155-173)
    static pp::DevBuf& tab = *new pp::DevBuf();  // (never destroyed: no hipFree after the runtime is gone)
    static int tab_ne = -1;
    if (tab_ne != ne) {
      std::vector<int> h((size_t)2 * std::max(ne, 1));
      const double lambda = 1.0f;
      const double freq_max = std::log(1.0 / ne) * -1;
      for (int uni = 0; uni < ne; ++uni) {
        const double percent_elem = ((double)uni) / ne;
        const double temp = -1 / lambda * std::log(1 - percent_elem) / freq_max;
        const double temp_next = -1 / lambda * std::log(1 - percent_elem - 1.0 / ne) / freq_max;
        const double a = temp * ne, b = temp_next * ne;
        h[(size_t)uni] = (a >= 0 && a < 2147483647.0) ? (int)a : 2147483647;
        h[(size_t)ne + uni] = (b >= 0 && b < 2147483647.0) ? (int)b : 2147483647;
      }
      PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
      PP_HIP_CHECK(tab.reserve(sizeof(int) * h.size()));
      PP_HIP_CHECK(hipMemcpy(tab.p, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
      tab_ne = ne;
    }
    es = tab.as<int>();
    ee = es + ne;
  }
  k_redistribute<<<grid_for(ps->capacity), kBlock, 0, pp::stream()>>>(
      ps->capacity, ps->d_mask.as<unsigned char>(), pp::slot_elem(ps), ne, percent_moved, seed,
      new_elems_dev, strat, es, ee);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
int pp_redistribute_particles(const pp_ps* ps, double percent_moved, unsigned long long seed,
                              int* new_elems_dev) {
  return pp_redistribute_particles_dist(ps, 1, percent_moved, seed, new_elems_dev);
}

int pp_ps_metrics(const pp_ps* ps, int* padded_cells, int* padded_slices, int* empty_rows) {
  PP_REQUIRE(ps, "pp_ps_metrics: null ps");
  PP_REQUIRE(ps->kind == PP_SCS, "pp_ps_metrics: SCS only (SellCSigma.h:465-524)");
  // The reference's drivers print the metrics in EVERY iteration of their loop (test/pseudoXGCm.cpp:505-506): counted
  // on the device (one block per slice sums the zero bytes of its part of the mask), 8 bytes come back -- round 5
  // copied the whole mask to the host and counted there (12 MB and a host loop per call at 10 M particles).
  int h[2] = {0, 0};
  if (ps->num_slices > 0 && ps->capacity > 0) {
    static pp::DevBuf* s_m = new pp::DevBuf();
    const size_t words = 2 + (size_t)ps->num_slices;
    PP_HIP_CHECK(s_m->reserve(words * sizeof(int)));
    PP_HIP_CHECK(hipMemsetAsync(s_m->p, 0, words * sizeof(int), pp::stream()));
    const unsigned segs = (unsigned)(((size_t)ps->V * ps->C + kMetricsSeg - 1) / kMetricsSeg);  // (a slice is at most V x C slots)
    k_slice_padding<<<dim3((unsigned)ps->num_slices, std::max(segs, 1u)), 256, 0, pp::stream()>>>(
        ps->d_offsets.as<int>(), ps->d_mask.as<unsigned char>(), s_m->as<int>(), s_m->as<int>() + 2);
    PP_LAUNCH_CHECK();
    PP_HIP_CHECK(hipMemcpyAsync(h, s_m->p, 2 * sizeof(int), hipMemcpyDeviceToHost, pp::stream()));
    PP_HIP_CHECK(hipStreamSynchronize(pp::stream()));
  }
  if (padded_cells) *padded_cells = h[0];
  if (padded_slices) *padded_slices = h[1];
  if (empty_rows) *empty_rows = ps->num_empty_elements;
  return PP_OK;
}

}  // extern "C"
