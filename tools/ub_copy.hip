// Ceilings for pass 2 of the staged move (k_move_unpack): read 32-B records, write 7 SoA streams.
//   hipcc --offload-arch=gfx950 -O3 tools/ub_copy.hip -o /tmp/ubc && /tmp/ubc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
struct Dst { u64* d8[6]; unsigned* d4; };

__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ s, uint4* __restrict__ d, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}
__global__ void __launch_bounds__(256) k_copy16_nt(const uint4* __restrict__ s, uint4* __restrict__ d, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  typedef unsigned v4 __attribute__((ext_vector_type(4)));
  if (i < n) __builtin_nontemporal_store(((const v4*)s)[i], (v4*)d + i);
}
__global__ void __launch_bounds__(256) k_read16(const uint4* __restrict__ s, unsigned* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { uint4 v = s[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) out[0] = 1; }
}
template <int NT> __global__ void __launch_bounds__(256) k_write7(Dst t, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 6; ++k) { if (NT) __builtin_nontemporal_store((u64)i + k, t.d8[k] + i); else t.d8[k][i] = (u64)i + k; }
  if (NT) __builtin_nontemporal_store((unsigned)i, t.d4 + i); else t.d4[i] = (unsigned)i;
}
template <int NT> __device__ __forceinline__ void put(const Dst& t, long long slot, uint4 a, uint4 b) {
  const u64 v0 = ((u64)a.y << 32) | a.x, v1 = ((u64)a.w << 32) | a.z, v2 = ((u64)b.y << 32) | b.x;
  if (NT) {
    __builtin_nontemporal_store(v0, t.d8[0] + slot); __builtin_nontemporal_store(v1, t.d8[1] + slot);
    __builtin_nontemporal_store(v2, t.d8[2] + slot); __builtin_nontemporal_store(b.w, t.d4 + slot);
    __builtin_nontemporal_store(0ull, t.d8[3] + slot); __builtin_nontemporal_store(0ull, t.d8[4] + slot);
    __builtin_nontemporal_store(0ull, t.d8[5] + slot);
  } else {
    t.d8[0][slot] = v0; t.d8[1][slot] = v1; t.d8[2][slot] = v2; t.d4[slot] = b.w;
    t.d8[3][slot] = 0; t.d8[4][slot] = 0; t.d8[5][slot] = 0;
  }
}
template <int NT> __global__ void __launch_bounds__(256) k_unpack_flat(const uint4* __restrict__ aos, Dst t, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  put<NT>(t, i, aos[2 * i], aos[2 * i + 1]);
}
// two adjacent lanes share a record?  no: lane-pair transposed read: lane reads uint4 #lane of the
// wave's 2048-B span (fully coalesced 16 B/lane), then exchanges via shuffles
template <int NT> __global__ void __launch_bounds__(256) k_unpack_coal(const uint4* __restrict__ aos, Dst t, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int l = threadIdx.x & 63;
  const long long w0 = (i - l) * 2;
  const uint4 p = aos[w0 + l], q = aos[w0 + 64 + l];
  // record r = 2 quads at 2r, 2r+1: quads 0..63 are in p (lane=quad), 64..127 in q
  const int qa = (2 * l) & 63, qb = (2 * l + 1) & 63;
  uint4 a, b;
  a.x = __shfl(l < 32 ? p.x : q.x, qa); // placeholder; replaced below
  // gather from p or q depending on the half
  const unsigned pax = __shfl(p.x, qa), pay = __shfl(p.y, qa), paz = __shfl(p.z, qa), paw = __shfl(p.w, qa);
  const unsigned qax = __shfl(q.x, qa), qay = __shfl(q.y, qa), qaz = __shfl(q.z, qa), qaw = __shfl(q.w, qa);
  const unsigned pbx = __shfl(p.x, qb), pby = __shfl(p.y, qb), pbz = __shfl(p.z, qb), pbw = __shfl(p.w, qb);
  const unsigned qbx = __shfl(q.x, qb), qby = __shfl(q.y, qb), qbz = __shfl(q.z, qb), qbw = __shfl(q.w, qb);
  const bool lo = l < 32;
  a = lo ? make_uint4(pax, pay, paz, paw) : make_uint4(qax, qay, qaz, qaw);
  b = lo ? make_uint4(pbx, pby, pbz, pbw) : make_uint4(qbx, qby, qbz, qbw);
  put<NT>(t, i, a, b);
}
// the production structure: thread = (tile,row), TP columns
template <int NT, int TP, int UNR> __global__ void __launch_bounds__(256) k_unpack_rows(const uint4* __restrict__ aos, Dst t, long long n) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long tile = g >> 6; const int r = g & 63;
  const long long start = tile * 64 * TP + r;
  if (start >= n) return;
#pragma unroll UNR
  for (int p = 0; p < TP; ++p) {
    const long long slot = start + p * 64;
    put<NT>(t, slot, aos[2 * slot], aos[2 * slot + 1]);
  }
}
// ---- production-like pass 2 (c3): 64-B records, 9 streams (3 x 8 B data, 3 x 4 B data, 3 x 8 B zero),
// mask byte per slot, optional tile table indirection
struct Dst9 { u64* d8[3]; unsigned* d4[3]; u64* z8[3]; };
template <int NT> __device__ __forceinline__ void put9(const Dst9& t, long long slot, uint4 a, uint4 b, uint4 c) {
  const u64 v0 = ((u64)a.y << 32) | a.x, v1 = ((u64)a.w << 32) | a.z, v2 = ((u64)b.y << 32) | b.x;
  if (NT) {
    __builtin_nontemporal_store(v0, t.d8[0] + slot); __builtin_nontemporal_store(v1, t.d8[1] + slot);
    __builtin_nontemporal_store(v2, t.d8[2] + slot);
    __builtin_nontemporal_store(b.z, t.d4[0] + slot); __builtin_nontemporal_store(b.w, t.d4[1] + slot);
    __builtin_nontemporal_store(c.x, t.d4[2] + slot);
    __builtin_nontemporal_store(0ull, t.z8[0] + slot); __builtin_nontemporal_store(0ull, t.z8[1] + slot);
    __builtin_nontemporal_store(0ull, t.z8[2] + slot);
  } else {
    t.d8[0][slot] = v0; t.d8[1][slot] = v1; t.d8[2][slot] = v2; t.d4[0][slot] = b.z; t.d4[1][slot] = b.w;
    t.d4[2][slot] = c.x; t.z8[0][slot] = 0; t.z8[1][slot] = 0; t.z8[2][slot] = 0;
  }
}
// MODE bit0: mask load + branch; bit1: tile table; bit2: read only 3 of the 4 quads
template <int NT, int MODE> __global__ void __launch_bounds__(256) k_unpack_prod(const uint4* __restrict__ aos, Dst9 t,
    const unsigned char* __restrict__ mask, const int* __restrict__ tiles, long long n) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  long long tile = g >> 6; const int r = g & 63;
  if (tile * 512 >= n) return;
  long long start = tile * 512 + r;
  if (MODE & 2) start = (long long)tiles[2 * tile] * 512 + tiles[2 * tile + 1] + r;
  for (int p = 0; p < 8; ++p) {
    const long long slot = start + p * 64;
    if ((MODE & 1) && !mask[slot]) continue;
    const uint4 a = aos[4 * slot], b = aos[4 * slot + 1], c = aos[4 * slot + 2];
    uint4 d = c;
    if (!(MODE & 4)) d = aos[4 * slot + 3];
    put9<NT>(t, slot, a, b, make_uint4(c.x ^ d.w, 0, 0, 0));
  }
}
// all columns' mask bytes first, then all records of the live columns, then the stores
template <int NT> __global__ void __launch_bounds__(256) k_unpack_prod_batched(const uint4* __restrict__ aos, Dst9 t,
    const unsigned char* __restrict__ mask, long long n) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long tile = g >> 6; const int r = g & 63;
  if (tile * 512 >= n) return;
  const long long start = tile * 512 + r;
  unsigned char m[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) m[p] = mask[start + p * 64];
#pragma unroll
  for (int h = 0; h < 8; h += 2) {
    uint4 a[2], b[2], c[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) if (m[h + q]) { const long long slot = start + (h + q) * 64; a[q] = aos[4 * slot]; b[q] = aos[4 * slot + 1]; c[q] = aos[4 * slot + 2]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) if (m[h + q]) put9<NT>(t, start + (h + q) * 64, a[q], b[q], make_uint4(c[q].x, 0, 0, 0));
  }
}
template <typename F> static void timeit(const char* name, double bytes, F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
  printf("%-34s %.3f ms  %5.0f GB/s\n", name, ms, bytes / ms * 1e-6);
}
int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 10485760;  // slots (10 M: partly MALL-resident; 100 M: HBM)
  uint4 *aos, *aos2; CK(hipMalloc(&aos, n * 32)); CK(hipMalloc(&aos2, n * 32));
  CK(hipMemset(aos, 1, n * 32));
  Dst t; for (int k = 0; k < 6; ++k) CK(hipMalloc(&t.d8[k], n * 8)); CK(hipMalloc(&t.d4, n * 4));
  unsigned* out; CK(hipMalloc(&out, 4));
  const int gq = (int)((2 * n + 255) / 256), gs = (int)((n + 255) / 256);
  timeit("read16 (320 MB)", 32.0 * n, [&] { k_read16<<<gq, 256>>>(aos, out, 2 * n); });
  timeit("copy16 (r+w 640 MB)", 64.0 * n, [&] { k_copy16<<<gq, 256>>>(aos, aos2, 2 * n); });
  timeit("copy16 nt store", 64.0 * n, [&] { k_copy16_nt<<<gq, 256>>>(aos, aos2, 2 * n); });
  timeit("memset 320 MB (hipMemsetAsync)", 32.0 * n, [&] { CK(hipMemsetAsync(aos2, 0, n * 32, 0)); });
  timeit("write7 plain (52 B/slot)", 52.0 * n, [&] { k_write7<0><<<gs, 256>>>(t, n); });
  timeit("write7 nt", 52.0 * n, [&] { k_write7<1><<<gs, 256>>>(t, n); });
  timeit("unpack flat plain (84 B/slot)", 84.0 * n, [&] { k_unpack_flat<0><<<gs, 256>>>(aos, t, n); });
  timeit("unpack flat nt", 84.0 * n, [&] { k_unpack_flat<1><<<gs, 256>>>(aos, t, n); });
  timeit("unpack coalesced-read nt", 84.0 * n, [&] { k_unpack_coal<1><<<gs, 256>>>(aos, t, n); });
  const int gr = (int)((n / 8 + 255) / 256);
  timeit("unpack rows TP8 nt", 84.0 * n, [&] { k_unpack_rows<1, 8, 1><<<gr, 256>>>(aos, t, n); });
  timeit("unpack rows TP8 nt unroll 4", 84.0 * n, [&] { k_unpack_rows<1, 8, 4><<<gr, 256>>>(aos, t, n); });
  timeit("unpack rows TP8 plain", 84.0 * n, [&] { k_unpack_rows<0, 8, 1><<<gr, 256>>>(aos, t, n); });
  // producer->consumer: records just written (as after pass 1), then unpacked
  timeit("copy16 then unpack flat nt", (64.0 + 84.0) * n, [&] { k_copy16<<<gq, 256>>>(aos2, aos, 2 * n); k_unpack_flat<1><<<gs, 256>>>(aos, t, n); });
  {  // production-like: 64-B records
    uint4* aos4; CK(hipMalloc(&aos4, n * 64)); CK(hipMemset(aos4, 1, n * 64));
    Dst9 d; for (int k = 0; k < 3; ++k) { CK(hipMalloc(&d.d8[k], n * 8)); CK(hipMalloc(&d.z8[k], n * 8)); CK(hipMalloc(&d.d4[k], n * 4)); }
    unsigned char* hm = (unsigned char*)malloc(n); unsigned long long x = 88172645463325252ull;
    long long live = 0;
    for (long long i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hm[i] = (x % 10) != 0; live += hm[i]; }
    unsigned char* dm; CK(hipMalloc(&dm, n)); CK(hipMemcpy(dm, hm, n, hipMemcpyHostToDevice));
    const long long ntile = n / 512;
    int* ht = (int*)malloc(ntile * 8); for (long long i = 0; i < ntile; ++i) { ht[2 * i] = (int)i; ht[2 * i + 1] = 0; }
    int* dt; CK(hipMalloc(&dt, ntile * 8)); CK(hipMemcpy(dt, ht, ntile * 8, hipMemcpyHostToDevice));
    const int gr9 = (int)((n / 8 + 255) / 256);
    const double full = (64.0 + 60.0) * n, lv = (64.0 + 60.0) * live + n;
    timeit("prod NQ4 9 streams, no mask", full, [&] { k_unpack_prod<1, 0><<<gr9, 256>>>(aos4, d, dm, dt, n); });
    timeit("prod + mask(10% dead)", lv, [&] { k_unpack_prod<1, 1><<<gr9, 256>>>(aos4, d, dm, dt, n); });
    timeit("prod + mask + tile table", lv, [&] { k_unpack_prod<1, 3><<<gr9, 256>>>(aos4, d, dm, dt, n); });
    timeit("prod + mask, 3 of 4 quads read", lv - 16.0 * live, [&] { k_unpack_prod<1, 5><<<gr9, 256>>>(aos4, d, dm, dt, n); });
    {  // mask variants: all live (isolates the load+branch), sorted-chunk pattern (live lanes are a prefix)
      unsigned char* hm2 = (unsigned char*)malloc(n); memset(hm2, 1, n);
      unsigned char* dm2; CK(hipMalloc(&dm2, n)); CK(hipMemcpy(dm2, hm2, n, hipMemcpyHostToDevice));
      timeit("prod + mask (all live)", full + n, [&] { k_unpack_prod<1, 1><<<gr9, 256>>>(aos4, d, dm2, dt, n); });
      long long live2 = 0;
      for (long long c = 0; c < n / 64; ++c) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const int k = (x % 10 == 0) ? 0 : 64 - (int)((x >> 8) % 12);
        for (int l = 0; l < 64; ++l) { hm2[c * 64 + l] = l < k; live2 += l < k; } }
      CK(hipMemcpy(dm2, hm2, n, hipMemcpyHostToDevice));
      timeit("prod + mask (prefix-live columns)", (64.0 + 60.0) * live2 + n, [&] { k_unpack_prod<1, 1><<<gr9, 256>>>(aos4, d, dm2, dt, n); });
    }
    timeit("prod plain stores + mask", lv, [&] { k_unpack_prod<0, 1><<<gr9, 256>>>(aos4, d, dm, dt, n); });
    timeit("prod batched (masks first, 2 cols)", lv - 16.0 * live, [&] { k_unpack_prod_batched<1><<<gr9, 256>>>(aos4, d, dm, n); });
  }
  return 0;
}
