// Streaming skeletons for the fused push+walk kernel (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ub_stream.hip -o /tmp/ubs && /tmp/ubs
// Same arrays and bytes per slot as bench c2 (read x[3] f64, b,phi f32, mask u8, ids i32; write
// xt[3] f64, phi f32, ids i32 = 69 B), trivial arithmetic.  Which loop structure streams fastest?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
struct Arr { const double* x; double* xt; const float* b; float* phi; const unsigned char* m; int* ids; long long stride; };

__device__ __forceinline__ void body(const Arr& a, long long pid, double x, double y, double z, float b, float ph, unsigned char m, int id) {
  if (!m) return;
  a.xt[pid] = x + b; a.xt[a.stride + pid] = y * 1.5; a.xt[2 * a.stride + pid] = z - ph;
  a.phi[pid] = ph + 0.25f; a.ids[pid] = id + 1;
}
// S1: thread=(tile,row), TP columns, prefetch 1 ahead (the production structure)
template <int TP, int OCC> __global__ void __launch_bounds__(256, OCC) s_rows(Arr a, long long cap) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long tile = g >> 6; const int r = g & 63;
  const long long start = tile * 64 * TP + r;
  if (start >= cap) return;
  long long pid = start;
  double x = a.x[pid], y = a.x[a.stride + pid], z = a.x[2 * a.stride + pid]; float b = a.b[pid], ph = a.phi[pid]; unsigned char m = a.m[pid]; int id = a.ids[pid];
  for (int p = 0; p < TP; ++p) {
    const double cx = x, cy = y, cz = z; const float cb = b, cph = ph; const unsigned char cm = m; const int cid = id; const long long cpid = pid;
    if (p + 1 < TP) { pid += 64; x = a.x[pid]; y = a.x[a.stride + pid]; z = a.x[2 * a.stride + pid]; b = a.b[pid]; ph = a.phi[pid]; m = a.m[pid]; id = a.ids[pid]; }
    body(a, cpid, cx, cy, cz, cb, cph, cm, cid);
  }
}
// S2: flat, one slot per thread
template <int OCC> __global__ void __launch_bounds__(256, OCC) s_flat(Arr a, long long cap) {
  const long long pid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pid >= cap) return;
  body(a, pid, a.x[pid], a.x[a.stride + pid], a.x[2 * a.stride + pid], a.b[pid], a.phi[pid], a.m[pid], a.ids[pid]);
}
// S3: thread=(tile,row), all TP columns loaded up front, then computed/stored
template <int TP, int OCC> __global__ void __launch_bounds__(256, OCC) s_rows_all(Arr a, long long cap) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long tile = g >> 6; const int r = g & 63;
  const long long start = tile * 64 * TP + r;
  if (start >= cap) return;
  double x[TP], y[TP], z[TP]; float b[TP], ph[TP]; unsigned char m[TP]; int id[TP];
#pragma unroll
  for (int p = 0; p < TP; ++p) { const long long pid = start + p * 64; x[p] = a.x[pid]; y[p] = a.x[a.stride + pid]; z[p] = a.x[2 * a.stride + pid]; b[p] = a.b[pid]; ph[p] = a.phi[pid]; m[p] = a.m[pid]; id[p] = a.ids[pid]; }
#pragma unroll
  for (int p = 0; p < TP; ++p) body(a, start + p * 64, x[p], y[p], z[p], b[p], ph[p], m[p], id[p]);
}
// S4: rows, prefetch 2 ahead
template <int TP, int OCC> __global__ void __launch_bounds__(256, OCC) s_rows2(Arr a, long long cap) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long tile = g >> 6; const int r = g & 63;
  const long long start = tile * 64 * TP + r;
  if (start >= cap) return;
  double x[3], y[3], z[3]; float b[3], ph[3]; unsigned char m[3]; int id[3];
#pragma unroll
  for (int p = 0; p < 2; ++p) { const long long pid = start + p * 64; x[p] = a.x[pid]; y[p] = a.x[a.stride + pid]; z[p] = a.x[2 * a.stride + pid]; b[p] = a.b[pid]; ph[p] = a.phi[pid]; m[p] = a.m[pid]; id[p] = a.ids[pid]; }
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int cu = p % 3, nx = (p + 2) % 3;
    if (p + 2 < TP) { const long long pid = start + (p + 2) * 64; x[nx] = a.x[pid]; y[nx] = a.x[a.stride + pid]; z[nx] = a.x[2 * a.stride + pid]; b[nx] = a.b[pid]; ph[nx] = a.phi[pid]; m[nx] = a.m[pid]; id[nx] = a.ids[pid]; }
    body(a, start + p * 64, x[cu], y[cu], z[cu], b[cu], ph[cu], m[cu], id[cu]);
  }
}
template <class F> float timeit(F f, int reps = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const long long cap = 10485760 + 64 * 8 * 100;  // multiple of 512
  Arr a; double *x, *xt; float *b, *phi; unsigned char* m; int* ids;
  CK(hipMalloc(&x, 24 * cap)); CK(hipMalloc(&xt, 24 * cap)); CK(hipMalloc(&b, 4 * cap)); CK(hipMalloc(&phi, 4 * cap)); CK(hipMalloc(&m, cap)); CK(hipMalloc(&ids, 4 * cap));
  CK(hipMemset(x, 0, 24 * cap)); CK(hipMemset(b, 0, 4 * cap)); CK(hipMemset(phi, 0, 4 * cap)); CK(hipMemset(m, 1, cap)); CK(hipMemset(ids, 0, 4 * cap));
  a = {x, xt, b, phi, m, ids, cap};
  const double gb = 69.0 * cap / 1e9;
#define RUN(name, launch) { float t = timeit([&] { launch; }); printf("%-28s %.3f ms  %.0f GB/s\n", name, t, gb / (t * 1e-3)); }
  const unsigned gflat = (unsigned)((cap + 255) / 256);
  RUN("flat occ8", (s_flat<8><<<gflat, 256>>>(a, cap)));
  RUN("flat occ4", (s_flat<4><<<gflat, 256>>>(a, cap)));
  RUN("rows TP8 pf1 occ4", (s_rows<8, 4><<<(unsigned)(cap / 8 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP8 pf1 occ8", (s_rows<8, 8><<<(unsigned)(cap / 8 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP4 pf1 occ4", (s_rows<4, 4><<<(unsigned)(cap / 4 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP16 pf1 occ4", (s_rows<16, 4><<<(unsigned)(cap / 16 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP8 pf2 occ4", (s_rows2<8, 4><<<(unsigned)(cap / 8 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP4 all occ4", (s_rows_all<4, 4><<<(unsigned)(cap / 4 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP2 all occ4", (s_rows_all<2, 4><<<(unsigned)(cap / 2 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP2 all occ8", (s_rows_all<2, 8><<<(unsigned)(cap / 2 / 256 + 1), 256>>>(a, cap)));
  RUN("rows TP8 all occ4", (s_rows_all<8, 4><<<(unsigned)(cap / 8 / 256 + 1), 256>>>(a, cap)));
  // occupancy sweep: dynamic LDS caps resident blocks per CU (160 KB / lds) = waves per SIMD
  for (int wps : {8, 6, 5, 4, 3, 2}) {
    const size_t lds = (size_t)(160 * 1024 / wps) - 512;
    char nm[64];
    snprintf(nm, sizeof nm, "rows TP8 pf1  %d waves/SIMD", wps);
    RUN(nm, (s_rows<8, 4><<<(unsigned)(cap / 8 / 256 + 1), 256, lds>>>(a, cap)));
    snprintf(nm, sizeof nm, "rows TP8 pf2  %d waves/SIMD", wps);
    RUN(nm, (s_rows2<8, 4><<<(unsigned)(cap / 8 / 256 + 1), 256, lds>>>(a, cap)));
    snprintf(nm, sizeof nm, "flat          %d waves/SIMD", wps);
    RUN(nm, (s_flat<4><<<gflat, 256, lds>>>(a, cap)));
  }
  return 0;
}
