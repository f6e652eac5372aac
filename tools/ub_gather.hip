// Micro-benchmark behind the walk-record fetch design (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ub_gather.hip -o /tmp/ubg && /tmp/ubg
// Every lane needs one 128-B record per item at a data-dependent index (12.9 MB table, L2/MALL
// resident).  (A) each lane reads its own record with 8 x dwordx4 (64 distinct lines per
// instruction); (B) 8 lanes share one record: a wave instruction touches 8 lines, the 16-B pieces
// are transposed through LDS (XOR-swizzled, conflict-free); (C) = B with global_load_lds (no
// staging VGPRs / ds_write).  P = fraction of items whose index differs from the lane's previous.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int TP = 8;
struct alignas(16) Rec { double v[16]; };

__device__ __forceinline__ double sum_rec(const double2* p) {
  double s = 0;
  for (int i = 0; i < 8; ++i) { double2 a = p[i]; s += a.x + a.y; }
  return s;
}

// layout like the SCS tile: thread = (tile,row), item p at idx[(tile*TP + p)*64 + lane]
__global__ void __launch_bounds__(256) k_lane(int ntiles, const int* __restrict__ idx,
                                               const Rec* __restrict__ recs, double* out) {
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (t >= ntiles) return;
  int have = -1;
  double cache[16];
  for (int i = 0; i < 16; ++i) cache[i] = 0;
  for (int p = 0; p < TP; ++p) {
    const size_t s = ((size_t)t * TP + p) * 64 + lane;
    const int e = idx[s];
    if (e != have) {
      const double2* r = (const double2*)(recs + e);
      for (int i = 0; i < 8; ++i) { double2 a = r[i]; cache[2 * i] = a.x; cache[2 * i + 1] = a.y; }
      have = e;
    }
    double acc = 0;
    for (int i = 0; i < 16; ++i) acc += cache[i];
    out[s] = acc;
  }
}

__global__ void __launch_bounds__(256) k_coop(int ntiles, const int* __restrict__ idx,
                                               const Rec* __restrict__ recs, double* out) {
  __shared__ double2 st[4][64 * 8];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= ntiles) return;
  int have = -1;
  double cache[16];
  for (int i = 0; i < 16; ++i) cache[i] = 0;
  for (int p = 0; p < TP; ++p) {
    const size_t s = ((size_t)t * TP + p) * 64 + lane;
    const int e = idx[s];
    const int want = (e != have) ? e : -1;
    if (__ballot(want >= 0)) {
      for (int j = 0; j < 8; ++j) {
        const int o = 8 * j + (lane >> 3);
        const int eo = __shfl(want, o);
        const int piece = (lane & 7) ^ (o & 7);
        if (eo >= 0) st[w][o * 8 + (lane & 7)] = ((const double2*)(recs + eo))[piece];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (want >= 0) {
        for (int i = 0; i < 8; ++i) {
          const double2 a = st[w][lane * 8 + (i ^ (lane & 7))];
          cache[2 * i] = a.x;
          cache[2 * i + 1] = a.y;
        }
        have = e;
      }
      __builtin_amdgcn_wave_barrier();
    }
    double acc = 0;
    for (int i = 0; i < 16; ++i) acc += cache[i];
    out[s] = acc;
  }
}

__global__ void __launch_bounds__(256) k_glds(int ntiles, const int* __restrict__ idx,
                                               const Rec* __restrict__ recs, double* out) {
  __shared__ double2 st[4][64 * 8];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= ntiles) return;
  int have = -1;
  double cache[16];
  for (int i = 0; i < 16; ++i) cache[i] = 0;
  for (int p = 0; p < TP; ++p) {
    const size_t s = ((size_t)t * TP + p) * 64 + lane;
    const int e = idx[s];
    const int want = (e != have) ? e : -1;
    if (__ballot(want >= 0)) {
      for (int j = 0; j < 8; ++j) {
        const int o = 8 * j + (lane >> 3);
        int eo = __shfl(want, o);
        const int piece = (lane & 7) ^ (o & 7);
        if (eo < 0) eo = __shfl(e, o);  // keep the wave's LDS image dense: re-read the cached one
        __builtin_amdgcn_global_load_lds((const void*)(((const double2*)(recs + eo)) + piece),
                                         (__attribute__((address_space(3))) void*)&st[w][j * 64], 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) only
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (want >= 0) {
        for (int i = 0; i < 8; ++i) {
          const double2 a = st[w][lane * 8 + (i ^ (lane & 7))];
          cache[2 * i] = a.x;
          cache[2 * i + 1] = a.y;
        }
        have = e;
      }
      __builtin_amdgcn_wave_barrier();
    }
    double acc = 0;
    for (int i = 0; i < 16; ++i) acc += cache[i];
    out[s] = acc;
  }
}

template <class F>
float timeit(F f, int reps = 10) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int ne = 100800, ntiles = 20480;
  const size_t n = (size_t)ntiles * TP * 64;
  std::vector<Rec> hrec(ne);
  for (int e = 0; e < ne; ++e) for (int i = 0; i < 16; ++i) hrec[e].v[i] = e * 0.001 + i;
  Rec* recs; int* idx; double *o1, *o2, *o3;
  CK(hipMalloc(&recs, sizeof(Rec) * ne)); CK(hipMalloc(&idx, 4 * n));
  CK(hipMalloc(&o1, 8 * n)); CK(hipMalloc(&o2, 8 * n)); CK(hipMalloc(&o3, 8 * n));
  CK(hipMemcpy(recs, hrec.data(), sizeof(Rec) * ne, hipMemcpyHostToDevice));
  std::vector<int> h(n);
  // (D) footprint per XCD: blocks are dealt round-robin to the 8 XCDs; draw every tile's indices
  // from a window of the table chosen by (block % 8) so each XCD's L2 sees only that window
  for (int win : {ne, ne / 8, ne / 32}) {
    for (int t = 0; t < ntiles; ++t) {
      const int xcd = (t / 4) % 8;  // 4 tiles (waves) per 256-thread block
      const int base = (win == ne) ? 0 : xcd * (ne / 8);
      for (int l = 0; l < 64; ++l)
        for (int p = 0; p < TP; ++p)
          h[((size_t)t * TP + p) * 64 + l] = base + rand() % win;
    }
    CK(hipMemcpy(idx, h.data(), 4 * n, hipMemcpyHostToDevice));
    const int grid = (ntiles * 64 + 255) / 256;
    float a = timeit([&] { k_lane<<<grid, 256>>>(ntiles, idx, recs, o1); });
    float c = timeit([&] { k_glds<<<grid, 256>>>(ntiles, idx, recs, o3); });
    printf("window/XCD = %6.2f MB  all-miss  per-lane %.3f ms | global_load_lds %.3f ms\n",
           win * 128.0 / 1e6, a, c);
  }
  for (double P : {1.0, 0.5, 0.1}) {
    srand(7);
    for (int t = 0; t < ntiles; ++t)
      for (int l = 0; l < 64; ++l) {
        int e = (int)(((size_t)t * 64 + l) * 2654435761u % ne);
        for (int p = 0; p < TP; ++p) {
          if (p == 0 || rand() < P * RAND_MAX) e = (e + 1 + rand() % 7) % ne;  // a neighbour-ish element
          h[((size_t)t * TP + p) * 64 + l] = e;
        }
      }
    CK(hipMemcpy(idx, h.data(), 4 * n, hipMemcpyHostToDevice));
    const int grid = (ntiles * 64 + 255) / 256;
    float a = timeit([&] { k_lane<<<grid, 256>>>(ntiles, idx, recs, o1); });
    float b = timeit([&] { k_coop<<<grid, 256>>>(ntiles, idx, recs, o2); });
    float c = timeit([&] { k_glds<<<grid, 256>>>(ntiles, idx, recs, o3); });
    std::vector<double> r1(n), r2(n), r3(n);
    CK(hipMemcpy(r1.data(), o1, 8 * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2, 8 * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r3.data(), o3, 8 * n, hipMemcpyDeviceToHost));
    size_t bad2 = 0, bad3 = 0;
    for (size_t i = 0; i < n; ++i) { bad2 += r1[i] != r2[i]; bad3 += r1[i] != r3[i]; }
    printf("P(change)=%.1f  items=%zu  per-lane %.3f ms | coop+LDS %.3f ms (bad %zu) | global_load_lds %.3f ms (bad %zu)\n",
           P, n, a, b, bad2, c, bad3);
  }
  return 0;
}
