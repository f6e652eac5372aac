// cost of blocks that do (almost) nothing: tools/ub_emptyblocks.hip -- hipcc --offload-arch=gfx950 -O3 -o /tmp/ub_eb tools/ub_emptyblocks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_exit(const int* flag) {
  if (*flag) return;  // uniform scalar load, then exit
  __builtin_trap();
}
__global__ void k_load_exit(const int* __restrict__ a, int* out) {
  const int v = a[blockIdx.x * 256 + threadIdx.x];
  if (v >= 0) out[0] = v;  // never
}
__global__ void k_load_sync_exit(const int* __restrict__ a, int* out) {
  const int v = a[blockIdx.x * 256 + threadIdx.x];
  if (!__syncthreads_or(v >= 0)) return;
  out[0] = v;
}
int main() {
  const int nb[3] = {15000, 42000, 100000};
  int *a, *flag, *out;
  hipMalloc(&a, 100000 * 256 * 4);
  hipMemset(a, 0xff, 100000 * 256 * 4);
  hipMalloc(&flag, 4);
  hipMalloc(&out, 4);
  int one = 1;
  hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int v = 0; v < 3; ++v)
    for (int n : nb) {
      float best = 1e9;
      for (int rep = 0; rep < 20; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) {
          if (v == 0) k_exit<<<n, 256>>>(flag);
          if (v == 1) k_load_exit<<<n, 256>>>(a, out);
          if (v == 2) k_load_sync_exit<<<n, 256>>>(a, out);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%s %6d blocks: %.2f us per launch, %.3f ns per block\n",
             v == 0 ? "flag-exit     " : v == 1 ? "load-exit     " : "load-sync-exit", n, best * 100, best * 1e5 / n);
    }
  return 0;
}
