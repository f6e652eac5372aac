// Scattered record writes by record size (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ub_scatter2.hip -o /tmp/ub2 && /tmp/ub2
// 640 MB of records written to random record-aligned positions of a 1 GB buffer: R/16 adjacent lanes
// write one R-byte record (16 B each), plain and non-temporal stores.  Question: does a scattered write of
// whole 128-B lines (two 64-B records of one row side by side) beat 64-B records per byte?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int LPR, bool NT>  // lanes per record
__global__ void k_write(long long nquads, const int* __restrict__ dst, v4u* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nquads) return;
  const long long rec = i / LPR;
  const int part = (int)(i % LPR);
  v4u v;
  v.x = (unsigned)i; v.y = v.x + 1; v.z = v.x + 2; v.w = v.x + 3;
  v4u* p = out + (long long)dst[rec] * LPR + part;
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <class F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const long long bytes = 640ll << 20, cap_bytes = 1ll << 30;
  v4u* big; CK(hipMalloc(&big, cap_bytes));
  const long long nquads = bytes / 16;
  srand(1);
  for (int R : {32, 64, 128, 256, 512}) {
    const long long nrec = bytes / R, slots = cap_bytes / R;
    std::vector<int> dst(nrec);
    for (long long i = 0; i < nrec; ++i) dst[i] = (int)((((long long)rand() << 15) ^ rand()) % slots);
    int* d; CK(hipMalloc(&d, nrec * 4)); CK(hipMemcpy(d, dst.data(), nrec * 4, hipMemcpyHostToDevice));
    const int B = 256; const unsigned G = (unsigned)((nquads + B - 1) / B);
    float t0 = 0, t1 = 0;
    switch (R) {
      case 32: t0 = timeit([&] { k_write<2, false><<<G, B>>>(nquads, d, big); }); t1 = timeit([&] { k_write<2, true><<<G, B>>>(nquads, d, big); }); break;
      case 64: t0 = timeit([&] { k_write<4, false><<<G, B>>>(nquads, d, big); }); t1 = timeit([&] { k_write<4, true><<<G, B>>>(nquads, d, big); }); break;
      case 128: t0 = timeit([&] { k_write<8, false><<<G, B>>>(nquads, d, big); }); t1 = timeit([&] { k_write<8, true><<<G, B>>>(nquads, d, big); }); break;
      case 256: t0 = timeit([&] { k_write<16, false><<<G, B>>>(nquads, d, big); }); t1 = timeit([&] { k_write<16, true><<<G, B>>>(nquads, d, big); }); break;
      case 512: t0 = timeit([&] { k_write<32, false><<<G, B>>>(nquads, d, big); }); t1 = timeit([&] { k_write<32, true><<<G, B>>>(nquads, d, big); }); break;
    }
    printf("random %3d-B records, 640 MB: plain %.3f ms (%.2f TB/s)   non-temporal %.3f ms (%.2f TB/s)\n", R, t0,
           bytes / t0 / 1e9, t1, bytes / t1 / 1e9);
    CK(hipFree(d));
  }
  return 0;
}
