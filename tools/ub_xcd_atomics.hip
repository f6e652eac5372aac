// Micro-benchmark: returning atomics on ~10^5 distinct counters, device (agent) scope against
// workgroup-scope atomics on a per-XCD copy of the counters selected by the wave's real XCC id
// (s_getreg HW_REG_XCC_ID): all updates of copy x then come from XCD x, whose CUs share one L2.
// Checks that the per-XCD copies add up to the agent-scope result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7); }  // HW_REG_XCC_ID bits 0..3
__global__ void k_agent(int n, int ne, int* hist, int* rank, int local) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int e = local ? (int)((i / 100 + (hashu(i) % 5)) % ne) : (int)(hashu(i) % (unsigned)ne);
  rank[i] = atomicAdd(&hist[e], 1);
}
__global__ void k_xcd(int n, int ne, int* hist8, int* rank, int local) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int e = local ? (int)((i / 100 + (hashu(i) % 5)) % ne) : (int)(hashu(i) % (unsigned)ne);
  const int x = xcc_id();
  int r = __hip_atomic_fetch_add(&hist8[(size_t)x * ne + e], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  rank[i] = r | (x << 28);
}
int main() {
  const int n = 10000000;
  for (int ne : {100800, 998400}) for (int local = 0; local < 2; ++local) {
    int *hist, *hist8, *rank;
    CK(hipMalloc(&hist, sizeof(int) * ne)); CK(hipMalloc(&hist8, sizeof(int) * 8 * (size_t)ne)); CK(hipMalloc(&rank, sizeof(int) * n));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms_agent = 0, ms_xcd = 0;
    for (int it = 0; it < 5; ++it) {
      CK(hipMemset(hist, 0, sizeof(int) * ne)); CK(hipMemset(hist8, 0, sizeof(int) * 8 * (size_t)ne));
      hipEventRecord(a); k_agent<<<(n + 255) / 256, 256>>>(n, ne, hist, rank, local); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (it) ms_agent += ms / 4;
      hipEventRecord(a); k_xcd<<<(n + 255) / 256, 256>>>(n, ne, hist8, rank, local); hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b); if (it) ms_xcd += ms / 4;
    }
    std::vector<int> h(ne), h8(8 * (size_t)ne);
    CK(hipMemcpy(h.data(), hist, sizeof(int) * ne, hipMemcpyDeviceToHost)); CK(hipMemcpy(h8.data(), hist8, sizeof(int) * 8 * (size_t)ne, hipMemcpyDeviceToHost));
    long bad = 0; long per[8] = {0};
    for (int e = 0; e < ne; ++e) { int s = 0; for (int x = 0; x < 8; ++x) { s += h8[(size_t)x * ne + e]; per[x] += h8[(size_t)x * ne + e]; } bad += s != h[e]; }
    printf("ne %7d %s: agent-scope %.1f us (%.1f ps/atomic), per-XCD workgroup-scope %.1f us (%.1f ps/atomic), mismatching counters %ld, per-XCD totals", ne, local ? "local " : "random",
           ms_agent * 1e3, ms_agent * 1e9 / n, ms_xcd * 1e3, ms_xcd * 1e9 / n, bad);
    for (int x = 0; x < 8; ++x) printf(" %ld", per[x]);
    printf("\n");
    hipFree(hist); hipFree(hist8); hipFree(rank);
  }
  return 0;
}
