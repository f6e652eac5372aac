// Micro-benchmarks behind the rebuild design (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ub_scatter.hip -o /tmp/ub && /tmp/ub
// (1) device-scope atomicAdd throughput on ~100k counters, with / without return value
// (2) random 64-B record writes: 4 x dwordx4 per lane vs LDS-transposed full-sector stores
// (3) random 8-B stores (what a direct SoA scatter does)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                      \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void k_atomic_noret(int n, const int* __restrict__ key, int* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&cnt[key[i]], 1);
}
__global__ void k_atomic_ret(int n, const int* __restrict__ key, int* cnt, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = atomicAdd(&cnt[key[i]], 1);
}
__global__ void k_atomic_wg(int n, const int* __restrict__ key, int* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __hip_atomic_fetch_add(&cnt[key[i]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_write64_direct(int n, const int* __restrict__ dst, uint4* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4* d = out + (size_t)dst[i] * 4;
  const uint4 v = make_uint4(i, i + 1, i + 2, i + 3);
  d[0] = v;
  d[1] = v;
  d[2] = v;
  d[3] = v;
}
__global__ void k_write64_lds(int n, const int* __restrict__ dst, uint4* out) {
  __shared__ uint4 st[4][64][5];
  __shared__ int sd[4][64];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const uint4 v = make_uint4(i, i + 1, i + 2, i + 3);
  for (int q = 0; q < 4; ++q) st[w][l][q] = v;
  sd[w][l] = (i < n) ? dst[i] : -1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int j = 0; j < 4; ++j) {
    const int rec = j * 16 + (l >> 2), part = l & 3;
    const int d = sd[w][rec];
    if (d >= 0) out[(size_t)d * 4 + part] = st[w][rec][part];
  }
}
__global__ void k_write8(int n, const int* __restrict__ dst, double* out, int ncomp, size_t stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < ncomp; ++c) out[c * stride + dst[i]] = (double)i;
}
__global__ void k_copy(int n, const uint4* __restrict__ in, uint4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

template <class F>
float timeit(F f, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int n = 10'000'000, ncnt = 100'000, cap = 15'000'000;
  std::vector<int> key(n), dst(n);
  srand(1);
  for (int i = 0; i < n; ++i) key[i] = (int)(((long long)rand() * 32768 + rand()) % ncnt);
  // destination permutation-ish: random slot in [0,cap)
  for (int i = 0; i < n; ++i) dst[i] = (int)(((long long)rand() * 32768 + rand()) % cap);
  // "mostly local" destinations: slot = i + small jitter
  std::vector<int> dloc(n);
  for (int i = 0; i < n; ++i) dloc[i] = (i + (rand() % 4096)) % cap;
  int *dkey, *ddst, *dloc_d, *cnt, *out;
  uint4* big;
  double* soa;
  CK(hipMalloc(&dkey, n * 4));
  CK(hipMalloc(&ddst, n * 4));
  CK(hipMalloc(&dloc_d, n * 4));
  CK(hipMalloc(&cnt, ncnt * 4));
  CK(hipMalloc(&out, n * 4));
  CK(hipMalloc(&big, (size_t)cap * 64));
  CK(hipMalloc(&soa, (size_t)cap * 8 * 8));
  CK(hipMemcpy(dkey, key.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ddst, dst.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dloc_d, dloc.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(cnt, 0, ncnt * 4));
  const int B = 256, G = (n + B - 1) / B;
  printf("n=%d counters=%d\n", n, ncnt);
  printf("atomic no-return (agent)   : %.3f ms\n", timeit([&] { k_atomic_noret<<<G, B>>>(n, dkey, cnt); }));
  printf("atomic with return (agent) : %.3f ms\n", timeit([&] { k_atomic_ret<<<G, B>>>(n, dkey, cnt, out); }));
  printf("atomic no-return (wg scope): %.3f ms\n", timeit([&] { k_atomic_wg<<<G, B>>>(n, dkey, cnt); }));
  printf("write 64B random, direct   : %.3f ms\n", timeit([&] { k_write64_direct<<<G, B>>>(n, ddst, big); }));
  printf("write 64B random, via LDS  : %.3f ms\n", timeit([&] { k_write64_lds<<<G, B>>>(n, ddst, big); }));
  printf("write 64B local, direct    : %.3f ms\n", timeit([&] { k_write64_direct<<<G, B>>>(n, dloc_d, big); }));
  printf("write 64B local, via LDS   : %.3f ms\n", timeit([&] { k_write64_lds<<<G, B>>>(n, dloc_d, big); }));
  printf("write 8x8B random (SoA)    : %.3f ms\n", timeit([&] { k_write8<<<G, B>>>(n, ddst, soa, 8, cap); }));
  printf("write 8x8B local  (SoA)    : %.3f ms\n", timeit([&] { k_write8<<<G, B>>>(n, dloc_d, soa, 8, cap); }));
  printf("copy 320 MB (16B/lane)     : %.3f ms\n",
         timeit([&] { k_copy<<<(n * 2 + B - 1) / B, B>>>(n * 2, big, big + (size_t)n * 2); }));
  return 0;
}
