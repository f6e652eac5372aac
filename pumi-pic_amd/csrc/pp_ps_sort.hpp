// pp_ps_sort.hpp -- the layout sort of the SCS re-layout and the scans around it (included by pp_ps.hip only).
//
// Reference: particle_structs/src/scs/SCS_sort.h (sigma sort of the rows by particle count),
// SCS_buildFns.h (chunk widths, padding, slice offsets).  Contents: the rebuild's device totals (Totals), the
// sweep that makes the sort keys and the totals of the new population (k_make_keys), the stable LSD radix sort
// in its three forms (separate launches; scan folded into the scatter for <= 64 tiles; one counting pass over an
// 11-bit digit for single-window structures + the ordering of its overflow digit), exclusive scans, the
// per-chunk kernels of the inversely padded layout.
#pragma once
#include "pp_internal.hpp"

namespace {

using pp::grid_for;
using pp::kBlock;

struct Totals {  // s_misc layout
  int active;    // live particles after the rebuild
  int nonempty;  // elements with >= 1 particle
  int invalid;   // new particle with element -1 / out of range
  int cw_sum, cw_cnt;
  int nslices, capacity;
  int go;  // speculative tail of the rebuild may run (k_spec_check); 1 on the checked path
  double cw_inv;
  unsigned long long max_key;  // largest sort key of this rebuild (radix passes above it are skipped)
  // in-place rebuild: rows whose new count exceeds their chunk width, rows that traded places,
  // "no home found for an overflowing row"
  int n_over;    // rows whose new count exceeds their chunk width (the reference then re-lays out)
  int sort_bad;  // one-pass layout sort: more keys in the overflow digit than its fix-up holds (re-sort with every pass)
  int pad_[2];   // ([0]: the stamp the host polls for)
  int second_key1;       // 1 + the second-largest count of a one-window sort (0: not known) and the first slot of
  int last_chunk_start;  // the last chunk: what pp_ps::hot is made of
  // the multi-block layout kernel (k_layout_multi): arrival counters of its two grid barriers (put back to zero by
  // the kernel's last block) and the per-block partial results: widths (sum, non-zero), slices, slots, tiles
  unsigned bar, bar_done;
  int mb[16][5];
};
constexpr int kLayoutBlocksMax = 16;

// ---- stable LSD radix sort (8-bit digits) of (key64, val32)
constexpr int RS_TILE = 2048;  // keys per block
// (also the per-element totals of the new population -- live particles, non-empty elements, rows
// that overflow their current chunk: k_nonempty and k_fit_check in the same sweep -- when `totals`)
struct ElemTotalsArgs {
  int totals;  // accumulate active / nonempty
  int fit;     // accumulate n_over against the CURRENT layout
  int C_old;
  const int *e2r_old, *chunk_width_old;
  // per-block partial sums (3 ints a block: non-empty, live, over) for the layout kernel to add up; null =
  // atomics on the totals (every block ends with three atomics on the same counters, ~10 ns each queued)
  int* partial;
};
// fused form of the radix passes (k_rs_pass / k_rs_pass_wide, ne <= kFusedSortBlocks tiles): the sweep that
// makes the keys also counts the first pass's digits per tile.
constexpr int kWideBits = 11, kWideDigits = 1 << kWideBits;
struct FusedHist {
  int* h0;   // digit counts of the first pass per tile: [256][nblk], wide: [nblk][ndig] (null = separate launches)
  int nblk;
  int wide;  // the ONE pass over one wide digit; keys of ndig - 1 and more share the last digit (k_rs_pass_wide)
  // digits of the one pass: 2048 (11 bits), or 256 / 64 when the previous rebuild's largest count says the keys are
  // small (round 6: at 10^6 elements the [tiles][2048] table is 4 MB that two extra kernels prefix, k_wide_seg /
  // k_wide_base; [tiles][64] is 125 KB that every block of the pass sums itself)
  int ndig = kWideDigits;
};
__device__ __forceinline__ int wide_digit(unsigned long long key, int ndig = kWideDigits) {
  return key < (unsigned long long)(ndig - 1) ? (int)key : ndig - 1;
}
__global__ void k_make_keys(int ne, const int* __restrict__ ppe, int sigma, int n_sigma,
                            unsigned long long base, unsigned long long* __restrict__ keys,
                            int* __restrict__ vals, Totals* tot, int no_skip, ElemTotalsArgs et,
                            FusedHist fh = FusedHist{nullptr, 0, 0}, int key_blocks = 1 << 30,
                            pp::GyroRide ride = pp::GyroRide{}) {
  if ((int)blockIdx.x >= key_blocks) {  // gyroScatter's first stage riding along (pp::GyroRide): needs only ppe
    pp::gyro_rings_body((blockIdx.x - key_blocks) * 256 + threadIdx.x, ride.nverts, ride.gnr, ride.v2e_off, ride.v2e,
                        ppe, ride.ringDown, ride.ringUp, ride.ring);
    return;
  }
  unsigned long long mx = 0;
  int nz = 0, sum = 0, over = 0;
  const int base_i = blockIdx.x * RS_TILE;
  __shared__ int s_h0[kWideDigits];
  const int dmask = fh.wide ? fh.ndig - 1 : 255;  // (digits of the first pass)
  if (fh.h0) {
    for (int d = threadIdx.x; d <= dmask; d += 256) s_h0[d] = 0;
    __syncthreads();
  }
  // (loads of the whole tile first -- counts, then the old row and chunk width of the fit test: three dependent
  // global loads per element would otherwise be paid eight times in a row)
  constexpr int R = RS_TILE / 256;
  int nn[R], cw_old[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = base_i + r * 256 + threadIdx.x;
    nn[r] = i < ne ? ppe[i] : 0;
    cw_old[r] = (et.fit && i < ne) ? et.e2r_old[i] : 0;
  }
  if (et.fit) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = base_i + r * 256 + threadIdx.x;
      cw_old[r] = i < ne ? et.chunk_width_old[cw_old[r] / et.C_old] : 0;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = base_i + r * 256 + threadIdx.x;
    if (i >= ne) break;
    int w = 0;
    if (sigma > 0) {
      w = i / sigma;
      if (w > n_sigma - 1) w = n_sigma - 1;
    }
    const int n = nn[r];
    const unsigned long long key = (unsigned long long)w * base + (unsigned long long)n;
    keys[i] = key;
    vals[i] = i;
    if (fh.h0) atomicAdd(&s_h0[fh.wide ? wide_digit(key, fh.ndig) : (int)(key & 255ull)], 1);
    mx = key > mx ? key : mx;
    nz += n > 0;
    sum += n;
    if (et.fit) over += n > cw_old[r];
  }
  if (et.totals) {
    __shared__ int s_t[4][3];
    for (int o = 32; o > 0; o >>= 1) {
      nz += __shfl_down(nz, o);
      sum += __shfl_down(sum, o);
      over += __shfl_down(over, o);
    }
    if ((threadIdx.x & 63) == 0) {
      s_t[threadIdx.x >> 6][0] = nz;
      s_t[threadIdx.x >> 6][1] = sum;
      s_t[threadIdx.x >> 6][2] = over;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      nz = s_t[0][0] + s_t[1][0] + s_t[2][0] + s_t[3][0];
      sum = s_t[0][1] + s_t[1][1] + s_t[2][1] + s_t[3][1];
      over = s_t[0][2] + s_t[1][2] + s_t[2][2] + s_t[3][2];
      if (et.partial) {
        et.partial[3 * blockIdx.x] = nz;
        et.partial[3 * blockIdx.x + 1] = sum;
        et.partial[3 * blockIdx.x + 2] = over;
      } else {
        if (nz) {
          atomicAdd(&tot->nonempty, nz);
          atomicAdd(&tot->active, sum);
        }
        if (over) atomicAdd(&tot->n_over, over);
      }
    }
    __syncthreads();
  }
  if (fh.h0) {
    __syncthreads();
    if (fh.wide)
      for (int d = threadIdx.x; d < fh.ndig; d += 256) fh.h0[(size_t)blockIdx.x * fh.ndig + d] = s_h0[d];
    else
      fh.h0[threadIdx.x * fh.nblk + blockIdx.x] = s_h0[threadIdx.x];
  }
  // the host sizes the number of 8-bit passes from an upper bound (total particles); the real
  // maximum (a per-element count) usually needs one or two passes: later passes see it and copy.
  // One atomic per 2048 keys: same-address atomics serialise at ~10 ns each.
  __shared__ unsigned long long smx[4];
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long y = __shfl_down(mx, o);
    mx = y > mx ? y : mx;
  }
  if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) mx = smx[k] > mx ? smx[k] : mx;
    if (mx) atomicMax(&tot->max_key, no_skip ? ~0ull : mx);
  }
}
__global__ void k_rs_hist(int n, const unsigned long long* __restrict__ keys, int shift, int nblk,
                          int* __restrict__ hist, const Totals* tot) {
  if ((tot->max_key >> shift) == 0) return;  // every digit of this pass is 0: identity pass
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
  for (int j = threadIdx.x; j < RS_TILE; j += 256) {
    const int i = base + j;
    if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1);
  }
  __syncthreads();
  hist[threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}
__global__ void k_rs_scatter(int n, const unsigned long long* __restrict__ keys,
                             const int* __restrict__ vals, int shift, int nblk,
                             const int* __restrict__ hist_scanned,
                             unsigned long long* __restrict__ keys_out, int* __restrict__ vals_out,
                             const Totals* tot) {
  if ((tot->max_key >> shift) == 0) {  // identity pass of the stable sort: plain copy
    for (int j = threadIdx.x; j < RS_TILE; j += 256) {
      const int i = blockIdx.x * RS_TILE + j;
      if (i < n) {
        keys_out[i] = keys[i];
        vals_out[i] = vals[i];
      }
    }
    return;
  }
  __shared__ int base_d[256];     // running count of each digit inside this tile
  __shared__ int wave_cnt[4][256];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  base_d[t] = 0;
  const int tile0 = blockIdx.x * RS_TILE;
  for (int round = 0; round < RS_TILE / 256; ++round) {
    for (int w = 0; w < 4; ++w) wave_cnt[w][t] = 0;
    __syncthreads();
    const int i = tile0 + round * 256 + t;
    const bool valid = i < n;
    unsigned long long key = 0;
    int val = 0, digit = 0;
    if (valid) {
      key = keys[i];
      val = vals[i];
      digit = (int)((key >> shift) & 255ull);
    }
    // lanes of this wave holding the same digit
    unsigned long long same = __ballot(valid);
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot(valid && ((digit >> b) & 1));
      same &= ((digit >> b) & 1) ? bal : ~bal;
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int rank_in_wave = __popcll(same & lt);
    if (valid && rank_in_wave == 0) wave_cnt[wave][digit] = __popcll(same);
    __syncthreads();
    if (valid) {
      int off = base_d[digit];
      for (int w = 0; w < wave; ++w) off += wave_cnt[w][digit];
      const int pos = hist_scanned[digit * nblk + blockIdx.x] + off + rank_in_wave;
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __syncthreads();
    base_d[t] += wave_cnt[0][t] + wave_cnt[1][t] + wave_cnt[2][t] + wave_cnt[3][t];
    __syncthreads();
  }
}

// The scatter of a radix pass with the scan of the digit table folded in (small structures: the table
// of <= kFusedSortBlocks tiles is read whole by every block).  The separate-launch form costs histogram +
// table scan (single block, 9 us) + scatter per pass (23 us at 100 800 elements,
// profiles/r03_c3_recordfed_kernel_stats.csv); here every block scans the table itself (thread d: the
// digit's total over all tiles and over the tiles before its own; one block scan over the 256 digits).
// (Also tried: the NEXT digit's histogram accumulated here with one atomic per key -- the high digit of
// nearly every key is 0, so the atomics pile onto one counter per tile: 250 us per pass.)
constexpr int kFusedSortBlocks = 64;
__global__ void k_rs_pass(int n, const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                          int shift, int nblk, const int* __restrict__ hist,
                          unsigned long long* __restrict__ keys_out, int* __restrict__ vals_out,
                          const Totals* tot) {
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const unsigned long long max_key = tot->max_key;
  if ((max_key >> shift) == 0) {  // identity pass of the stable sort: plain copy (so are all later ones)
    for (int j = t; j < RS_TILE; j += 256) {
      const int i = blockIdx.x * RS_TILE + j;
      if (i < n) {
        keys_out[i] = keys[i];
        vals_out[i] = vals[i];
      }
    }
    return;
  }
  __shared__ int gbase[256];  // first output position of this tile's keys with digit d
  __shared__ int s_part[4];
  {
    int total = 0, mine = 0;
    const int* row = hist + t * nblk;
#pragma unroll 8
    for (int b = 0; b < nblk; ++b) {  // (independent loads in flight: the loop is latency-bound)
      const int h = row[b];
      total += h;
      mine += b < (int)blockIdx.x ? h : 0;
    }
    // exclusive scan of `total` over the 256 digits
    int incl = total;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += s_part[w];
    gbase[t] = off + incl - total + mine;
  }
  __shared__ int base_d[256];  // running count of each digit inside this tile
  __shared__ int wave_cnt[4][256];
  base_d[t] = 0;
  const int tile0 = blockIdx.x * RS_TILE;
  for (int round = 0; round < RS_TILE / 256; ++round) {
    for (int w = 0; w < 4; ++w) wave_cnt[w][t] = 0;
    __syncthreads();
    const int i = tile0 + round * 256 + t;
    const bool valid = i < n;
    unsigned long long key = 0;
    int val = 0, digit = 0;
    if (valid) {
      key = keys[i];
      val = vals[i];
      digit = (int)((key >> shift) & 255ull);
    }
    unsigned long long same = __ballot(valid);  // lanes of this wave holding the same digit
    for (int b = 0; b < 8; ++b) {
      const unsigned long long bal = __ballot(valid && ((digit >> b) & 1));
      same &= ((digit >> b) & 1) ? bal : ~bal;
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int rank_in_wave = __popcll(same & lt);
    if (valid && rank_in_wave == 0) wave_cnt[wave][digit] = __popcll(same);
    __syncthreads();
    if (valid) {
      int off = base_d[digit];
      for (int w = 0; w < wave; ++w) off += wave_cnt[w][digit];
      const int pos = gbase[digit] + off + rank_in_wave;
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __syncthreads();
    base_d[t] += wave_cnt[0][t] + wave_cnt[1][t] + wave_cnt[2][t] + wave_cnt[3][t];
    __syncthreads();
  }
}
// Digit table of the one-pass sort for MANY tiles (> kFusedSortBlocks: every block sweeping the whole table
// stops paying): the tiles are cut into segments of kWideSeg; k_wide_seg turns hist[b][d] into the count of
// digit d in the earlier tiles of b's segment and leaves the segment totals, k_wide_base turns those into the
// count of digit d in the earlier segments and writes the first output position of every digit (digit-major
// order: all of digit 0, then digit 1 ...).  A tile's base for digit d is then base[d] + seg_tot[segment][d] +
// hist[b][d]: three loads per digit in k_rs_pass_wide.
constexpr int kWideSeg = 16;
__global__ void k_wide_seg(int nblk, int* __restrict__ hist, int* __restrict__ seg_tot) {
  const int d = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  const int b0 = s * kWideSeg;
  int h[kWideSeg];
#pragma unroll
  for (int j = 0; j < kWideSeg; ++j) h[j] = b0 + j < nblk ? hist[(size_t)(b0 + j) * kWideDigits + d] : 0;
  int run = 0;
#pragma unroll
  for (int j = 0; j < kWideSeg; ++j) {
    if (b0 + j < nblk) hist[(size_t)(b0 + j) * kWideDigits + d] = run;
    run += h[j];
  }
  seg_tot[(size_t)s * kWideDigits + d] = run;
}
__global__ void __launch_bounds__(1024) k_wide_base(int nseg, int* __restrict__ seg_tot, int* __restrict__ base) {
  constexpr int K = kWideDigits / 1024;
  __shared__ int s_part[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int total[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {  // thread t owns digits k*1024 + t: prefix over the segments, in place
    int run = 0;
    for (int s0 = 0; s0 < nseg; s0 += 16) {  // (16 loads in flight)
      int v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = s0 + j < nseg ? seg_tot[(size_t)(s0 + j) * kWideDigits + k * 1024 + t] : 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (s0 + j < nseg) seg_tot[(size_t)(s0 + j) * kWideDigits + k * 1024 + t] = run;
        run += v[j];
      }
    }
    total[k] = run;
  }
  int carry = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {  // exclusive scan of the digit totals in digit order
    int incl = total[k];
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int off = 0, all = 0;
    for (int w = 0; w < 16; ++w) {
      off += w < wave ? s_part[w] : 0;
      all += s_part[w];
    }
    base[k * 1024 + t] = carry + off + incl - total[k];
    carry += all;
    __syncthreads();
  }
}
// The whole sort as ONE counting pass, for structures with one sort window (the keys are the per-element
// counts): digit = min(key, 2047).  Counts of ~100 per element resolve exactly; the few rows above 2046 (the
// literal pseudoXGCm population piles its remainder into one element) land in the last digit in element order
// and are ordered by k_layout_fused's prologue (wide_fix_tail: up to 1024 of them, else Totals::sort_bad and
// the caller re-sorts with every 8-bit pass).  Two 8-bit passes + the second histogram cost 34 us at 100 800
// elements.  hist is [nblk][kWideDigits] (k_make_keys).  Ranks inside the tile
// as in k_rs_pass, with 16 waves per tile (two rounds); the per-wave counts live in a [16][2048] byte table
// whose used entries are put back to zero by their writers.
constexpr int kWideThreads = 1024;
constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }
template <int DIG>
__global__ void __launch_bounds__(kWideThreads)
    k_rs_pass_wide(int n, const unsigned long long* __restrict__ keys, const int* __restrict__ vals, int nblk,
                   const int* __restrict__ hist, unsigned long long* __restrict__ keys_out,
                   int* __restrict__ vals_out, const int* __restrict__ seg_base = nullptr,
                   const int* __restrict__ digit_base = nullptr) {
  constexpr int kWideDigits = DIG, kWideBits = ilog2_c(DIG);  // (shadow the 11-bit constants)
  constexpr int NW = kWideThreads / 64, R = RS_TILE / kWideThreads, K = DIG >= kWideThreads ? DIG / kWideThreads : 1;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  __shared__ int base_d[kWideDigits];                   // next output position of this tile's keys with digit d
  __shared__ unsigned char wave_cnt[NW][kWideDigits];   // (a wave holds at most 64 keys of one digit)
  __shared__ int s_part[NW];
  // this tile's keys first: their loads fly while the digit table is summed
  const int tile0 = blockIdx.x * RS_TILE;
  unsigned long long key[R];
  int val[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tile0 + r * kWideThreads + t;
    key[r] = i < n ? keys[i] : 0ull;
    val[r] = i < n ? vals[i] : 0;
  }
  for (int q = t; q < NW * kWideDigits / 16; q += kWideThreads) ((uint4*)&wave_cnt[0][0])[q] = make_uint4(0, 0, 0, 0);
  if constexpr (DIG < kWideThreads) {
    // narrow digit: the [tiles][DIG] table is small -- every block sums it itself whatever the number of tiles.
    // Thread t: digit t % DIG, tiles t / DIG, t / DIG + LN, ... (LN lanes of tiles per digit), then an LDS fold.
    constexpr int LN = kWideThreads / DIG;
    __shared__ int s_tot[LN][DIG], s_mine[LN][DIG];
    const int d = t % DIG, l = t / DIG;
    int total = 0, mine = 0;
    for (int b0 = l; b0 < nblk; b0 += 8 * LN) {  // (8 independent loads in flight)
      int h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = b0 + j * LN < nblk ? hist[(size_t)(b0 + j * LN) * DIG + d] : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        total += h[j];
        mine += b0 + j * LN < (int)blockIdx.x ? h[j] : 0;
      }
    }
    s_tot[l][d] = total;
    s_mine[l][d] = mine;
    __syncthreads();
    if (t < DIG) {
      total = mine = 0;
#pragma unroll
      for (int q = 0; q < LN; ++q) {
        total += s_tot[q][t];
        mine += s_mine[q][t];
      }
      int incl = total;  // exclusive scan over the DIG digits (DIG <= 256: up to four waves)
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      if (lane == 63) s_part[wave] = incl;
      s_tot[0][t] = incl - total + mine;  // (own slot: read above by this thread only)
    }
    __syncthreads();
    if (t < DIG) {
      int off = 0;
      for (int w = 0; w < wave; ++w) off += s_part[w];
      base_d[t] = off + s_tot[0][t];
    }
    __syncthreads();
  } else if (seg_base) {  // many tiles: the table was prefixed by k_wide_seg / k_wide_base
    const int* sb = seg_base + (size_t)(blockIdx.x / kWideSeg) * kWideDigits;
    const int* hb = hist + (size_t)blockIdx.x * kWideDigits;
#pragma unroll
    for (int k = 0; k < K; ++k)
      base_d[k * kWideThreads + t] = digit_base[k * kWideThreads + t] + sb[k * kWideThreads + t] + hb[k * kWideThreads + t];
    __syncthreads();
  } else if constexpr (DIG >= kWideThreads) {
    int total[K], mine[K];
#pragma unroll
    for (int k = 0; k < K; ++k) total[k] = mine[k] = 0;
    // (thread t owns digits k*1024 + t.  The table was written by the previous kernel's blocks on other XCDs:
    // every batch of loads costs a trip past the L2, so 16 rows = 32 loads are in flight at a time)
    for (int b0 = 0; b0 < nblk; b0 += 16) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int b = b0 + j;
        const int* row = hist + (size_t)min(b, nblk - 1) * kWideDigits + t;
        const bool before = b < (int)blockIdx.x;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int h = b < nblk ? row[k * kWideThreads] : 0;
          total[k] += h;
          mine[k] += before ? h : 0;
        }
      }
    }
    int carry = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {  // exclusive scan over the digits in order d = k*1024 + t
      int incl = total[k];
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
      }
      if (lane == 63) s_part[wave] = incl;
      __syncthreads();
      int off = 0, all = 0;
      for (int w = 0; w < NW; ++w) {
        const int p = s_part[w];
        off += w < wave ? p : 0;
        all += p;
      }
      base_d[k * kWideThreads + t] = carry + off + incl - total[k] + mine[k];
      carry += all;
      __syncthreads();
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool valid = tile0 + r * kWideThreads + t < n;
    const int digit = wide_digit(key[r], DIG);
    unsigned long long same = __ballot(valid);  // lanes of this wave holding the same digit
    for (int b = 0; b < kWideBits; ++b) {
      const unsigned long long bal = __ballot(valid && ((digit >> b) & 1));
      same &= ((digit >> b) & 1) ? bal : ~bal;
    }
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int rank_in_wave = __popcll(same & lt);
    const bool leader = valid && rank_in_wave == 0;
    const int cnt = __popcll(same);
    if (leader) wave_cnt[wave][digit] = (unsigned char)cnt;  // own row: zeroed by this wave's leaders last round
    __syncthreads();
    if (valid) {
      int off = base_d[digit];
      for (int w = 0; w < wave; ++w) off += wave_cnt[w][digit];
      const int pos = off + rank_in_wave;
      keys_out[pos] = key[r];
      vals_out[pos] = val[r];
    }
    __syncthreads();
    if (leader) {
      atomicAdd(&base_d[digit], cnt);
      wave_cnt[wave][digit] = 0;
    }
  }
}
// ---- single-block exclusive scan (int); total written to *total if non-null
__global__ void k_scan_excl(int n, const int* __restrict__ in, int* __restrict__ out, int* total,
                            const Totals* skip_tot = nullptr, int skip_shift = 0) {
  if (skip_tot && (skip_tot->max_key >> skip_shift) == 0) return;  // digit table of an identity pass
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  constexpr int ITEMS = 4;
  for (int base = 0; base < n; base += 1024 * ITEMS) {
    int v[ITEMS];
    int s = 0;
    for (int k = 0; k < ITEMS; ++k) {
      const int i = base + t * ITEMS + k;
      v[k] = (i < n) ? in[i] : 0;
      s += v[k];
    }
    int incl = s;
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = carry_s + woff + incl - s;
    for (int k = 0; k < ITEMS; ++k) {
      const int i = base + t * ITEMS + k;
      if (i < n) out[i] = run;
      run += v[k];
    }
    __syncthreads();
    if (t == 1023) carry_s = run;
    __syncthreads();
  }
  if (t == 0 && total) *total = carry_s;
}

// ---- three-launch scan for long arrays (radix digit tables of ~10^5 entries, CSR offsets of 10^6
// elements): per-block local scan + block totals, single-block scan of the totals, offset add.
// The single-block kernel above needs ~2.4 us per 4096 items (71 us for 125 k entries).
__global__ void __launch_bounds__(1024) k_scan_local(int n, const int* __restrict__ in, int* __restrict__ out,
                                                     int* __restrict__ block_tot, const Totals* skip_tot,
                                                     int skip_shift) {
  if (skip_tot && (skip_tot->max_key >> skip_shift) == 0) return;
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int ITEMS = 4;
  const int base = blockIdx.x * 1024 * ITEMS;
  int v[ITEMS];
  int s = 0;
  for (int k = 0; k < ITEMS; ++k) {
    const int i = base + t * ITEMS + k;
    v[k] = (i < n) ? in[i] : 0;
    s += v[k];
  }
  int incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  int run = woff + incl - s;
  for (int k = 0; k < ITEMS; ++k) {
    const int i = base + t * ITEMS + k;
    if (i < n) out[i] = run;
    run += v[k];
  }
  if (t == 1023) block_tot[blockIdx.x] = run;
}
__global__ void k_scan_add(int n, int* __restrict__ out, const int* __restrict__ block_off,
                           const Totals* skip_tot, int skip_shift) {
  if (skip_tot && (skip_tot->max_key >> skip_shift) == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_off[i / 4096];
}
// exclusive scan of n ints on the stream; *total (device, may be null) receives the sum
int scan_excl(pp::DevBuf& scratch, int n, const int* in, int* out, int* total, hipStream_t st,
              const Totals* skip_tot = nullptr, int skip_shift = 0) {
  if (n <= 16384) {
    k_scan_excl<<<1, 1024, 0, st>>>(n, in, out, total, skip_tot, skip_shift);
    return PP_OK;
  }
  const int nb = (n + 4095) / 4096;
  PP_HIP_CHECK(scratch.reserve(sizeof(int) * 2 * (size_t)nb));
  int* bt = (int*)scratch.p;
  k_scan_local<<<nb, 1024, 0, st>>>(n, in, out, bt, skip_tot, skip_shift);
  k_scan_excl<<<1, 1024, 0, st>>>(nb, bt, bt + nb, total, skip_tot, skip_shift);
  k_scan_add<<<grid_for(n), kBlock, 0, st>>>(n, out, bt + nb, skip_tot, skip_shift);
  return PP_OK;
}

// serial sum of 1/width in chunk order (only PAD_INVERSELY needs it; order-dependent in fp)
__global__ void k_cw_inv_serial(int nchunks, const int* __restrict__ widths, Totals* tot) {
  if (blockIdx.x || threadIdx.x) return;
  double s = 0;
  for (int c = 0; c < nchunks; ++c)
    if (widths[c] > 0) s += 1.0 / widths[c];
  tot->cw_inv = s;
}
__global__ void k_apply_padding(int nchunks, int pad_strat, double pad, int* __restrict__ widths,
                                const Totals* tot) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int cw_sum = tot->cw_sum;
  if (cw_sum <= 0) return;
  int w = widths[c];
  if (pad_strat == PP_PAD_EVENLY) {
    const int avg_pad = (int)(cw_sum * pad / tot->cw_cnt);
    if (w > 0) w += avg_pad;
  } else if (pad_strat == PP_PAD_PROPORTIONALLY) {
    w = (int)(w + w * pad);
  } else {
    const double cw_sum2 = cw_sum / tot->cw_inv * pad;
    if (w != 0) w = (int)(w + cw_sum2 / w);
  }
  widths[c] = w;
}
__global__ void k_slices_and_slots(int nchunks, int C, int V, const int* __restrict__ widths,
                                   int* __restrict__ nsl, int* __restrict__ nslots) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int w = widths[c];
  nsl[c] = w / V + ((w % V) != 0);
  nslots[c] = w * C;
}
__global__ void k_tile_count(int nchunks, int TP, const int* __restrict__ widths,
                             int* __restrict__ ntl) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nchunks) ntl[c] = (widths[c] + TP - 1) / TP;
}
// Prologue of the layout kernel after k_rs_pass_wide (one block of 1024 threads): the keys of the overflow
// digit sit at the end of the sorted arrays in element order; order them by key, ties by position (= the
// stable order the 8-bit passes produce).
__device__ void wide_fix_tail(int ne, int nblk, const int* __restrict__ hist, unsigned long long* keys,
                              int* vals, Totals* tot, const int* __restrict__ tail_start = nullptr,
                              int ndig = kWideDigits) {
  __shared__ unsigned long long fk[1024];
  __shared__ int fv[1024];
  __shared__ int s_n[16];
  const int t = threadIdx.x;
  int n = 0;
  if (tail_start) {  // (many tiles: the first output position of the overflow digit is in the prefixed table)
    n = ne - *tail_start;
  } else {
    for (int b = t; b < nblk; b += 1024) n += hist[(size_t)b * ndig + ndig - 1];
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((t & 63) == 0) s_n[t >> 6] = n;
    __syncthreads();
    n = 0;
    for (int w = 0; w < 16; ++w) n += s_n[w];
  }
  if (n <= 1) return;  // (block-uniform)
  if (n > 1024) {
    if (t == 0) tot->sort_bad = 1;
    return;
  }
  const int start = ne - n;
  if (t < n) {
    fk[t] = keys[start + t];
    fv[t] = vals[start + t];
  }
  __syncthreads();
  if (t < n) {
    const unsigned long long k = fk[t];
    int r = 0;
    for (int j = 0; j < n; ++j) r += (fk[j] < k) || (fk[j] == k && j < t);
    keys[start + r] = k;
    vals[start + r] = fv[t];
  }
  __syncthreads();
}
}  // namespace
