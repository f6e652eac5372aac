// pp_ps_distribute.hpp -- redistribute_particles and getPIDs kernels (included by pp_ps.hip only).
//
// Reference: particle_structs/test/Distribute.h:28-89, Distribute.cpp (strategies 1-4), ps_for.hpp:57-85 (getPIDs).
#pragma once
#include "pp_internal.hpp"

namespace {

// redistribute_particles (particle_structs/test/Distribute.h:28-89) with uniform re-draws: every live
// particle moves with probability percentMoved to a uniformly drawn element.  The reference draws
// from a Kokkos XorShift64 pool (not reproducible run to run); here the two draws of a slot are a
// splitmix64 hash of (seed, slot), so the CPU oracle produces the same ids.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
// The re-draw by distribution strategy (distribute_particles' device forms, Distribute.cpp:76-253): 1 uniform,
// 2 gaussian(ne/2, ne/8) truncated and clamped, 3 uniform -> exponential conversion (its two logarithms per
// element come from host-made tables), 4 GITRm approximation.  Extra draws are g_j = splitmix64(h1 + j); the
// normal variate is the Irwin-Hall sum of twelve 32-bit uniforms (exact in double: host and device agree).
__device__ __forceinline__ int draw_element(int strat, int ne, unsigned long long h1, const int* __restrict__ exp_start,
                                            const int* __restrict__ exp_end) {
  if (strat == 2) {
    double S = 0;
    for (int j = 0; j < 12; ++j) S += (double)(splitmix64(h1 + (unsigned long long)j) >> 32);
    const double z = S * (1.0 / 4294967296.0) - 6.0;
    const double v = ne / 2.0 + (ne / 8.0) * z;
    int elem = (int)v;
    if (elem < 0) elem = 0;
    if (elem >= ne) elem = ne - 1;
    return elem;
  }
  if (strat == 3) {
    const int uni = (int)(h1 % (unsigned long long)ne);
    if (uni == ne - 1) return 0;
    const int start = exp_start[uni];
    const long long length = (long long)exp_end[uni] - start;
    int inside = 0;
    if (length > 1) inside = (int)(splitmix64(h1 + 1ull) % (unsigned long long)length);
    long long e = (long long)start + inside;
    if (e >= ne) e = (long long)(splitmix64(h1 + 2ull) % (unsigned long long)ne);
    return (int)e;
  }
  if (strat == 4) {
    const int cutoff = 2 * ne / 5;
    const double u = (double)(splitmix64(h1 + 1ull) >> 11) * (1.0 / 9007199254740992.0);
    const unsigned long long g = splitmix64(h1 + 2ull);
    if (u < 0.85 && cutoff > 0) return (int)(g % (unsigned long long)cutoff);
    return cutoff + (int)(g % (unsigned long long)(ne - cutoff));
  }
  return (int)(h1 % (unsigned long long)ne);
}
__global__ void k_redistribute(int capacity, const unsigned char* __restrict__ mask,
                               const int* __restrict__ slot_elem, int ne, double percent_moved,
                               unsigned long long seed, int* __restrict__ new_elems, int strat,
                               const int* __restrict__ exp_start, const int* __restrict__ exp_end) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity) return;
  const int e = slot_elem[pid];
  if (e < 0 || !mask[pid]) {
    new_elems[pid] = -1;
    return;
  }
  const unsigned long long h0 = splitmix64(seed ^ (2ull * (unsigned long long)pid));
  const double prob = (double)(h0 >> 11) * (1.0 / 9007199254740992.0);
  if (prob <= percent_moved) {
    const unsigned long long h1 = splitmix64(seed ^ (2ull * (unsigned long long)pid + 1ull));
    new_elems[pid] = draw_element(strat, ne, h1, exp_start, exp_end);
  } else {
    new_elems[pid] = e;
  }
}
__global__ void k_pid_count(int capacity, const unsigned char* __restrict__ mask,
                            const int* __restrict__ slot_elem, int* __restrict__ ppe) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid < capacity && mask[pid]) atomicAdd(&ppe[slot_elem[pid]], 1);
}
__global__ void k_pid_set(int capacity, const unsigned char* __restrict__ mask,
                          const int* __restrict__ slot_elem, const int* __restrict__ offsets,
                          int* __restrict__ cur, int* __restrict__ pids) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid < capacity && mask[pid]) {
    const int e = slot_elem[pid];
    pids[offsets[e] + atomicAdd(&cur[e], 1)] = pid;
  }
}

}  // namespace
