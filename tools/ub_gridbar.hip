// What does a grid-wide barrier inside one kernel cost against a kernel boundary?  (the sort + layout
// chain of a rebuild is ~15 dependent kernels of a few dozen blocks each, ~5 us apiece)
//   hipcc --offload-arch=gfx950 -O3 tools/ub_gridbar.hip -o tools/_ubb && tools/_ubb
// Every phase: each block writes 256 ints, the barrier (or the kernel boundary), each block reads the
// 256 ints of the NEXT block and checks them -- so the barrier has to make plain stores visible across
// XCDs, as the real chain needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    target += nblocks;
    // release at agent scope: write-back of this XCD's L2 (the __syncthreads above ordered the block's stores)
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // invalidate: the other XCDs' stores become visible
  }
  __syncthreads();
}
__global__ void __launch_bounds__(256) k_phases(int nph, int* buf, unsigned* counter, int* bad) {
  unsigned target = 0;
  const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x;
  for (int ph = 0; ph < nph; ++ph) {
    buf[(ph & 1) * nb * 256 + b * 256 + t] = ph * 1000003 + b * 256 + t;
    grid_barrier(counter, nb, target);
    const int o = (b + 1) % nb;
    const int v = buf[(ph & 1) * nb * 256 + o * 256 + t];
    if (v != ph * 1000003 + o * 256 + t) atomicAdd(bad, 1);
  }
}
__global__ void __launch_bounds__(256) k_one_write(int ph, int* buf) {
  const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x;
  buf[(ph & 1) * nb * 256 + b * 256 + t] = ph * 1000003 + b * 256 + t;
}
__global__ void __launch_bounds__(256) k_one_check(int ph, const int* buf, int* bad) {
  const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x;
  const int o = (b + 1) % nb;
  if (buf[(ph & 1) * nb * 256 + o * 256 + t] != ph * 1000003 + o * 256 + t) atomicAdd(bad, 1);
}

int main() {
  const int nph = 40;
  int* buf;
  unsigned* counter;
  int* bad;
  CK(hipMalloc(&buf, sizeof(int) * 2 * 2048 * 256));
  CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&bad, 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int nb : {16, 49, 128, 256, 489, 1024}) {
    float best_coop = 1e9f, best_sep = 1e9f;
    int h_bad = 0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemsetAsync(counter, 0, 4, st));
      CK(hipMemsetAsync(bad, 0, 4, st));
      CK(hipEventRecord(e0, st));
      k_phases<<<nb, 256, 0, st>>>(nph, buf, counter, bad);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_coop) best_coop = ms;
      int hb;
      CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      h_bad += hb;
      CK(hipMemsetAsync(bad, 0, 4, st));
      CK(hipEventRecord(e0, st));
      for (int ph = 0; ph < nph; ++ph) {
        k_one_write<<<nb, 256, 0, st>>>(ph, buf);
        k_one_check<<<nb, 256, 0, st>>>(ph, buf, bad);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_sep) best_sep = ms;
      CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      h_bad += hb;
    }
    printf("blocks %5d: one kernel, %d grid barriers: %7.2f us per phase | %d kernel pairs: %7.2f us per kernel | mismatches %d\n",
           nb, nph, best_coop * 1000.f / nph, nph, best_sep * 1000.f / (2 * nph), h_bad);
  }
  return 0;
}
