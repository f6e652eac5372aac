// pp_ps_inplace.hpp -- kernels of the in-place rebuild (included by pp_ps.hip only).
//
// Reference: particle_structs/src/scs/SCS_rebuild.h:4-119 (reshuffle), decision :160-189.
#pragma once
#include "pp_internal.hpp"
#include "pp_ps_sort.hpp"  // Totals
#include "pp_ps_move.hpp"  // WordTable, MoveArgs, copy_members

namespace {

// ------------------------------------------------------------------ in-place rebuild ("reshuffle")
// The reference first tries to keep the layout (SCS_rebuild.h:4-119, decision :160-189): when every
// row's arrivals fit into its holes -- new count <= chunk width -- offsets / slice_to_chunk /
// row_to_element / element_to_row stay as they are and only the particles that change element move.
// Same decision here; the data movement is this library's own.  Rows stay prefix-compact (the hot
// kernels rely on it: a row's live slots are its first `count` columns), so a row that shrinks
// back-fills the holes below its new count from its own tail:
//   k_rs_count  thread = (run of <= 32 columns, row): arrivals per element (atomics that RETURN the
//               arrival's rank), leavers per element, the movers' records packed to aos[slot]
//   k_rs_fit    per element: new count = old - leavers + arrivals <= chunk width ?  totals, go flag
//   k_rs_plan   per run: holes (columns below the new count that are empty or being left) are
//               enumerated into the row's hole list, tail stayers (columns at or above the new count)
//               get claim numbers behind the arrivals; new mask
//   k_rs_move   claimant k of a row takes the row's k-th hole: movers from their staged record, tail
//               stayers slot to slot (their source slots are nobody's target)
// About 16 % of the particles move (8 % change element per pseudoXGCm step, as many again back-fill)
// instead of every particle twice.
struct RsCounters {  // one int array each, num_elems long, zeroed per rebuild (n_new lives in s_ppe)
  int *arrive, *leave, *hole_cur, *tail_cur, *removed;
};
template <int NQ>
__global__ void k_rs_count(const int* __restrict__ ntiles_dev, int C, int TP, int G,
                           const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                           const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                           const int* __restrict__ n_old, const int* __restrict__ new_element, int ne,
                           RsCounters cn, Totals* tot, int* __restrict__ rank, uint4* __restrict__ aos,
                           WordTable t) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = (int)(g / C), r = (int)(g - (long long)grp * C);
  const int ntiles = *ntiles_dev;
  int cur = -1, e = -1, start = 0, run_p0 = 0, nold = 0, nl = 0, nd = 0;
  // movers of the run that share one of the first three destinations met leave as ONE atomic each
  // (a row's particles cross into the few neighbours of its element, see k_count_tiled)
  int key1 = -1, key2 = -1, key3 = -1;
  unsigned m1 = 0, m2 = 0, m3 = 0;
  auto flush_one = [&](int key, unsigned m) {
    if (!m) return;
    int idx = atomicAdd(&cn.arrive[key], __popc(m));
    while (m) {
      const int b = __ffs(m) - 1;
      m &= m - 1;
      rank[start + (run_p0 + b) * C] = idx++;
    }
  };
  auto flush = [&]() {
    flush_one(key1, m1);
    flush_one(key2, m2);
    flush_one(key3, m3);
    if (nl) atomicAdd(&cn.leave[e], nl);
    if (nd) atomicAdd(&cn.removed[e], nd);
    m1 = m2 = m3 = 0;
    key1 = key2 = key3 = -1;
    nl = nd = 0;
  };
  for (int k = 0; k < G; ++k) {
    const int tile = grp * G + k;
    if (tile >= ntiles) break;
    const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
    if (c != cur) {
      flush();
      cur = c;
      start = chunk_start[c] + r;
      e = r2e[c * C + r];
      nold = e < ne ? n_old[e] : 0;
      run_p0 = p0;
    }
    const int pend = min(min(p0 + TP, chunk_width[c]), nold);  // live columns only
    for (int pb = p0; pb < pend; pb += 8) {
      int nel[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) nel[j] = (pb + j < pend) ? new_element[start + (pb + j) * C] : e;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ne_ = nel[j];
        if (ne_ == e) continue;  // stays (or column past the live range)
        ++nl;
        if (ne_ == -1) {  // removed
          ++nd;
          continue;
        }
        if (ne_ < 0 || ne_ >= ne) {
          tot->invalid = 1;
          continue;
        }
        const int pid = start + (pb + j) * C;
        {  // stage the mover: its slot may be another particle's target
          unsigned v[NQ * 4];
#pragma unroll
          for (int i = 0; i < NQ * 4; ++i) v[i] = 0u;
#pragma unroll
          for (int i = 0; i < NQ * 2; ++i)
            if (i < t.n8) {
              const unsigned long long d = *(const unsigned long long*)(t.src8[i] + (long long)pid * 8);
              v[2 * i] = (unsigned)d;
              v[2 * i + 1] = (unsigned)(d >> 32);
            }
#pragma unroll
          for (int i = 0; i < (NQ * 4 < kMax4 ? NQ * 4 : kMax4); ++i)
            if (i < t.n4) v[NQ * 4 - 1 - i] = *(const unsigned*)(t.src4[i] + (long long)pid * 4);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            aos[(long long)pid * NQ + q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
        const unsigned bit = 1u << (pb + j - run_p0);
        if (ne_ == key1) {
          m1 |= bit;
        } else if (ne_ == key2) {
          m2 |= bit;
        } else if (ne_ == key3) {
          m3 |= bit;
        } else if (key1 < 0) {
          key1 = ne_;
          m1 = bit;
        } else if (key2 < 0) {
          key2 = ne_;
          m2 = bit;
        } else if (key3 < 0) {
          key3 = ne_;
          m3 = bit;
        } else {
          rank[pid] = atomicAdd(&cn.arrive[ne_], 1);
        }
      }
    }
  }
  flush();
}
__global__ void k_rs_count_added(int n_new, const int* __restrict__ new_elems, int ne, int* __restrict__ arrive,
                                 Totals* tot, int* __restrict__ rank_new) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int e = new_elems[i];
  if (e < 0 || e >= ne) {
    tot->invalid = 1;
    rank_new[i] = -1;
    return;
  }
  rank_new[i] = atomicAdd(&arrive[e], 1);
}
// per element: the new count and whether it fits the row (SCS_rebuild.h:33-42: new particles of a row
// against its holes, i.e. new count <= chunk width); totals by one atomic pair per block
__global__ void k_rs_fit(int ne, int C, const int* __restrict__ n_old, RsCounters cn,
                         const int* __restrict__ e2r, const int* __restrict__ chunk_width,
                         int* __restrict__ n_new, Totals* tot) {
  __shared__ int s_sum[4], s_nz[4];
  int sum = 0, nz = 0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) {
    const int n = n_old[e] - cn.leave[e] + cn.arrive[e];
    n_new[e] = n;
    sum += n;
    nz += n > 0;
    // The reference counts a row's holes BEFORE its movers leave (SCS_rebuild.h:13-25: a slot is a hole
    // when it is empty or its particle is removed; a particle that moves to another row still
    // occupies its slot), so its test is  arrivals <= width - (old count - removed).
    const int occupied = n_old[e] - cn.removed[e] + cn.arrive[e];
    if (occupied > chunk_width[e2r[e] / C]) atomicAdd(&tot->n_over, 1);  // the row overflows: no in-place rebuild
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_down(sum, o);
    nz += __shfl_down(nz, o);
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[threadIdx.x >> 6] = sum;
    s_nz[threadIdx.x >> 6] = nz;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    nz = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
    if (sum) atomicAdd(&tot->active, sum);
    if (nz) atomicAdd(&tot->nonempty, nz);
  }
}
__global__ void k_rs_go(Totals* tot) {
  tot->go = (!tot->invalid && tot->active > 0 && tot->n_over == 0) ? 1 : 0;
}
__global__ void k_rs_plan(const int* __restrict__ ntiles_dev, int C, int TP, int G,
                          const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                          const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                          const int* __restrict__ n_old, const int* __restrict__ n_new,
                          const int* __restrict__ new_element, int ne, RsCounters cn,
                          int* __restrict__ hole_tab, int* __restrict__ rank,
                          unsigned char* __restrict__ mask, const int* __restrict__ go) {
  if (!*go) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = (int)(g / C), r = (int)(g - (long long)grp * C);
  const int ntiles = *ntiles_dev;
  int cur = -1, e = -1, start = 0, run_p0 = 0, nold = 0, nnew = 0, arr = 0;
  unsigned hb = 0, tb = 0;  // holes / tail stayers of the run, bit = column - run_p0
  auto flush = [&]() {
    if (hb) {
      int h = atomicAdd(&cn.hole_cur[e], __popc(hb));
      while (hb) {
        const int b = __ffs(hb) - 1;
        hb &= hb - 1;
        hole_tab[start + h * C] = start + (run_p0 + b) * C;
        ++h;
      }
    }
    if (tb) {
      int tk = arr + atomicAdd(&cn.tail_cur[e], __popc(tb));  // claims of the arrivals come first
      while (tb) {
        const int b = __ffs(tb) - 1;
        tb &= tb - 1;
        rank[start + (run_p0 + b) * C] = tk++;
      }
    }
  };
  for (int k = 0; k < G; ++k) {
    const int tile = grp * G + k;
    if (tile >= ntiles) break;
    const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
    if (c != cur) {
      flush();
      cur = c;
      start = chunk_start[c] + r;
      e = r2e[c * C + r];
      nold = nnew = arr = 0;
      if (e < ne) {
        nold = n_old[e];
        nnew = n_new[e];
        arr = cn.arrive[e];
      }
      run_p0 = p0;
    }
    const int pend = min(min(p0 + TP, chunk_width[c]), max(nold, nnew));
    for (int p = p0; p < pend; ++p) {
      const int pid = start + p * C;
      const bool live = p < nold;
      const bool stays = live && new_element[pid] == e;
      const unsigned bit = 1u << (p - run_p0);
      if (p < nnew) {
        if (!stays) hb |= bit;
      } else if (stays) {
        tb |= bit;
      }
      if ((p < nnew) != live) mask[pid] = p < nnew ? 1 : 0;
    }
  }
  flush();
}
template <int NQ>
__global__ void k_rs_move(const int* __restrict__ ntiles_dev, int C, int TP, int G,
                          const int* __restrict__ tiles, const int* __restrict__ chunk_start,
                          const int* __restrict__ chunk_width, const int* __restrict__ r2e,
                          const int* __restrict__ n_old, const int* __restrict__ n_new,
                          const int* __restrict__ new_element, int ne, const int* __restrict__ eslot0,
                          const int* __restrict__ hole_tab, const int* __restrict__ rank,
                          const uint4* __restrict__ aos, WordTable t, const int* __restrict__ go) {
  if (!*go) return;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = (int)(g / C), r = (int)(g - (long long)grp * C);
  const int ntiles = *ntiles_dev;
  int cur = -1, e = -1, start = 0, nold = 0, nnew = 0;
  for (int k = 0; k < G; ++k) {
    const int tile = grp * G + k;
    if (tile >= ntiles) break;
    const int c = tiles[2 * tile], p0 = tiles[2 * tile + 1];
    if (c != cur) {
      cur = c;
      start = chunk_start[c] + r;
      e = r2e[c * C + r];
      nold = nnew = 0;
      if (e < ne) {
        nold = n_old[e];
        nnew = n_new[e];
      }
    }
    const int pend = min(min(p0 + TP, chunk_width[c]), nold);
    for (int p = p0; p < pend; ++p) {
      const int pid = start + p * C;
      const int ne_ = new_element[pid];
      if (ne_ == e) {
        if (p < nnew) continue;  // stays where it is
        const long long tgt = hole_tab[start + rank[pid] * C];  // back-fill a hole of the own row
#pragma unroll
        for (int i = 0; i < NQ * 2; ++i)
          if (i < t.n8)
            *(unsigned long long*)(t.dst8[i] + tgt * 8) = *(const unsigned long long*)(t.dst8[i] + (long long)pid * 8);
#pragma unroll
        for (int i = 0; i < (NQ * 4 < kMax4 ? NQ * 4 : kMax4); ++i)
          if (i < t.n4) *(unsigned*)(t.dst4[i] + tgt * 4) = *(const unsigned*)(t.dst4[i] + (long long)pid * 4);
      } else if (ne_ >= 0) {
        const long long tgt = hole_tab[eslot0[ne_] + rank[pid] * C];
        const uint4* sp = aos + (long long)pid * NQ;
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
          if (i < t.n8)
            *(unsigned long long*)(t.dst8[i] + tgt * 8) = ((unsigned long long)w[2 * i + 1] << 32) | w[2 * i];
#pragma unroll
        for (int i = 0; i < (NQ * 4 < kMax4 ? NQ * 4 : kMax4); ++i)
          if (i < t.n4) *(unsigned*)(t.dst4[i] + tgt * 4) = w[NQ * 4 - 1 - i];
      }
    }
  }
}
// new particles take the holes their arrival ranks name
__global__ void k_rs_add(int n_new, const int* __restrict__ new_elems, const int* __restrict__ rank_new,
                         const int* __restrict__ eslot0, const int* __restrict__ hole_tab, int C, MoveArgs a,
                         const int* __restrict__ go) {
  if (!*go) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int tgt = hole_tab[eslot0[new_elems[i]] + rank_new[i] * C];
  copy_members(a, i, tgt);
}
__global__ void k_zero_gated(unsigned long long* __restrict__ p, long long n, const int* __restrict__ go) {
  if (!*go) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(0ull, p + i);
}

}  // namespace
