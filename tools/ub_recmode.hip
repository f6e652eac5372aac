// Would the drop-in loop gain if `Segment` read and wrote the 64-B records a re-layout leaves behind, so that the
// second pass of the re-layout (k_move_unpack<4>, 292 us at 10 M particles) never ran?  (round-5 verdict, item 3)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ub_recmode.hip -o tools/_ubr && tools/_ubr
// The three lambdas of test/pseudoXGCm.cpp's step that touch members, on a structure shaped like BASELINE configs[2]
// (10 M particles in 100 352 rows, chunks of 64 rows, ~12.4 M slots), in two storage forms:
//   soa   one array per component, slot-indexed (what get<N>() hands out today: 64 lanes = 64 consecutive slots)
//   rec   64-B records, row-major inside a chunk as the re-layout's first pass leaves them: the record of
//         (row r, column p) of chunk c is rec0[c] + r * pitch[c] + p, so the 64 lanes of a wave (the 64 rows of one
//         column) sit pitch * 64 B apart -- every lane in its own cache line
//   rec_t the same records with the thread mapping turned: lane = column, wave = row (64 consecutive records)
// kernels: push (reads phi, b; writes x_tgt.x, x_tgt.y, phi), update (x <- x_tgt, x_tgt <- 0), read5 (what
// search_mesh_2d reads: x.x, x.y, x_tgt.x, x_tgt.y, id; writes 4 B)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

struct Rec {  // 64 B: the pseudoXGCm particle
  double x[3], xt[3];
  int id;
  float b, phi;
  int pad;
};
struct Layout {
  int nchunks;
  const int *chunk_start, *width, *rec0, *pitch;  // per chunk
  const int* group_chunk;                          // 64-slot group -> chunk
};
struct Soa {
  double *x, *xt;  // [3][stride]
  int* id;
  float *b, *phi;
  long long stride;
};
__device__ __forceinline__ int rec_of(const Layout& L, int pid) {
  const int c = __builtin_amdgcn_readfirstlane(L.group_chunk[pid >> 6]);
  const int p = (pid - L.chunk_start[c]) >> 6, r = pid & 63;
  return L.rec0[c] + r * L.pitch[c] + p;
}
// ---- SoA
__global__ void k_push_soa(int cap, const unsigned char* mask, Soa s, double deg) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap || !mask[pid]) return;
  const double rad = (double)s.phi[pid] + deg, b = s.b[pid];
  s.xt[pid] = 0.6 * b * cos(rad) + 1.6;
  s.xt[s.stride + pid] = b * sin(rad) + 0.02;
  s.phi[pid] = (float)rad;
}
__global__ void k_update_soa(int cap, Soa s) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap) return;
  for (int k = 0; k < 3; ++k) {
    s.x[k * s.stride + pid] = s.xt[k * s.stride + pid];
    s.xt[k * s.stride + pid] = 0;
  }
}
__global__ void k_read5_soa(int cap, const unsigned char* mask, Soa s, int* out) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap) return;
  out[pid] = mask[pid] ? (int)(s.x[pid] + s.x[s.stride + pid] + s.xt[pid] + s.xt[s.stride + pid]) + s.id[pid] : -1;
}
// ---- records, lane = row (the mapping ps::parallel_for has today)
__global__ void k_push_rec(int cap, const unsigned char* mask, Layout L, Rec* rec, double deg) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap || !mask[pid]) return;
  Rec& q = rec[rec_of(L, pid)];
  const double rad = (double)q.phi + deg, b = q.b;
  q.xt[0] = 0.6 * b * cos(rad) + 1.6;
  q.xt[1] = b * sin(rad) + 0.02;
  q.phi = (float)rad;
}
__global__ void k_update_rec(int cap, Layout L, Rec* rec) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap) return;
  Rec& q = rec[rec_of(L, pid)];
  for (int k = 0; k < 3; ++k) {
    q.x[k] = q.xt[k];
    q.xt[k] = 0;
  }
}
__global__ void k_read5_rec(int cap, const unsigned char* mask, Layout L, const Rec* rec, int* out) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap) return;
  const Rec& q = rec[rec_of(L, pid)];
  out[pid] = mask[pid] ? (int)(q.x[0] + q.x[1] + q.xt[0] + q.xt[1]) + q.id : -1;
}
// ---- records, lane = column: wave w of the launch owns 64 consecutive columns of one row (tile table: chunk, row,
// first column), so its 64 records are 4 KB of contiguous memory
struct RowTile {
  int chunk, row, p0;
};
__global__ void k_update_rec_t(int ntiles, const RowTile* tiles, Layout L, Rec* rec) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (w >= ntiles) return;
  const RowTile t = tiles[w];
  const int p = t.p0 + lane;
  if (p >= L.width[t.chunk]) return;
  Rec& q = rec[L.rec0[t.chunk] + t.row * L.pitch[t.chunk] + p];
  for (int k = 0; k < 3; ++k) {
    q.x[k] = q.xt[k];
    q.xt[k] = 0;
  }
}
__global__ void k_push_rec_t(int ntiles, const RowTile* tiles, const unsigned char* mask, Layout L, Rec* rec, double deg) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (w >= ntiles) return;
  const RowTile t = tiles[w];
  const int p = t.p0 + lane;
  if (p >= L.width[t.chunk] || !mask[L.chunk_start[t.chunk] + t.row + p * 64]) return;
  Rec& q = rec[L.rec0[t.chunk] + t.row * L.pitch[t.chunk] + p];
  const double rad = (double)q.phi + deg, b = q.b;
  q.xt[0] = 0.6 * b * cos(rad) + 1.6;
  q.xt[1] = b * sin(rad) + 0.02;
  q.phi = (float)rad;
}
// the second pass of the re-layout, for scale: records -> SoA (lane = row, as k_move_unpack reads them)
__global__ void k_unpack(int cap, const unsigned char* mask, Layout L, const Rec* rec, Soa s) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= cap || !mask[pid]) return;
  const Rec q = rec[rec_of(L, pid)];
  for (int k = 0; k < 3; ++k) {
    s.x[k * s.stride + pid] = q.x[k];
    s.xt[k * s.stride + pid] = q.xt[k];
  }
  s.id[pid] = q.id;
  s.b[pid] = q.b;
  s.phi[pid] = q.phi;
}

template <class F>
float timeit(F f, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}
template <class T>
T* dev(const std::vector<T>& h) {
  T* p = nullptr;
  CK(hipMalloc((void**)&p, h.size() * sizeof(T) + 64));
  CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

int main() {
  const int nrows = 100352, C = 64, nchunks = nrows / C;
  std::vector<int> cnt(nrows);
  srand(7);
  long long np = 0;
  for (int r = 0; r < nrows; ++r) {  // ~100 per row, sorted ascending like a full sigma sort leaves them
    cnt[r] = 70 + (int)(60.0 * r / nrows) + rand() % 3;
    np += cnt[r];
  }
  std::vector<int> cs(nchunks), wd(nchunks), r0(nchunks), pt(nchunks);
  long long cap = 0, nrec = 0;
  for (int c = 0; c < nchunks; ++c) {
    int w = 0;
    for (int r = 0; r < C; ++r) w = std::max(w, cnt[c * C + r]);
    w += 12;  // (padding: ~10 %)
    cs[c] = (int)cap;
    wd[c] = w;
    pt[c] = (w + 3) & ~3;
    r0[c] = (int)nrec;
    cap += (long long)w * C;
    nrec += (long long)pt[c] * C;
  }
  std::vector<int> gc((size_t)cap / 64);
  std::vector<unsigned char> mask((size_t)cap, 0);
  std::vector<RowTile> tiles;
  for (int c = 0; c < nchunks; ++c) {
    for (int p = 0; p < wd[c]; ++p) gc[(size_t)cs[c] / 64 + p] = c;
    for (int r = 0; r < C; ++r) {
      for (int p = 0; p < cnt[c * C + r]; ++p) mask[(size_t)cs[c] + r + (size_t)p * C] = 1;
      for (int p0 = 0; p0 < wd[c]; p0 += 64) tiles.push_back(RowTile{c, r, p0});
    }
  }
  printf("%lld particles, %lld slots, %lld records (%.2f GB as 64-B records), %zu row tiles\n", np, cap, nrec,
         nrec * 64e-9, tiles.size());
  Layout L{nchunks, dev(cs), dev(wd), dev(r0), dev(pt), dev(gc)};
  unsigned char* d_mask = dev(mask);
  RowTile* d_tiles = dev(tiles);
  Soa s;
  s.stride = cap;
  CK(hipMalloc((void**)&s.x, 3 * cap * 8));
  CK(hipMalloc((void**)&s.xt, 3 * cap * 8));
  CK(hipMalloc((void**)&s.id, cap * 4));
  CK(hipMalloc((void**)&s.b, cap * 4));
  CK(hipMalloc((void**)&s.phi, cap * 4));
  CK(hipMemset(s.x, 0, 3 * cap * 8));
  CK(hipMemset(s.xt, 0, 3 * cap * 8));
  CK(hipMemset(s.id, 0, cap * 4));
  CK(hipMemset(s.b, 0, cap * 4));
  CK(hipMemset(s.phi, 0, cap * 4));
  Rec* rec = nullptr;
  CK(hipMalloc((void**)&rec, nrec * sizeof(Rec)));
  CK(hipMemset(rec, 0, nrec * sizeof(Rec)));
  int* out = nullptr;
  CK(hipMalloc((void**)&out, cap * 4));
  const unsigned g = (unsigned)((cap + 255) / 256), gt = (unsigned)((tiles.size() * 64 + 255) / 256);
  const int icap = (int)cap, nt = (int)tiles.size();
  struct Row {
    const char* name;
    float soa, rec, rec_t;
  } rows[3];
  rows[0] = {"push   (user lambda of ellipticalPush.hpp)",
             timeit([&] { k_push_soa<<<g, 256>>>(icap, d_mask, s, 0.01); }),
             timeit([&] { k_push_rec<<<g, 256>>>(icap, d_mask, L, rec, 0.01); }),
             timeit([&] { k_push_rec_t<<<gt, 256>>>(nt, d_tiles, d_mask, L, rec, 0.01); })};
  rows[1] = {"update (updatePtclPositions)",
             timeit([&] { k_update_soa<<<g, 256>>>(icap, s); }),
             timeit([&] { k_update_rec<<<g, 256>>>(icap, L, rec); }),
             timeit([&] { k_update_rec_t<<<gt, 256>>>(nt, d_tiles, L, rec); })};
  rows[2] = {"read5  (the members search_mesh_2d reads)",
             timeit([&] { k_read5_soa<<<g, 256>>>(icap, d_mask, s, out); }),
             timeit([&] { k_read5_rec<<<g, 256>>>(icap, d_mask, L, rec, out); }), 0.f};
  const float unpack = timeit([&] { k_unpack<<<g, 256>>>(icap, d_mask, L, rec, s); });
  printf("%-46s %9s %9s %9s   (us per launch)\n", "", "soa", "rec", "rec_t");
  float d_rec = 0, d_rect = 0;
  for (auto& r : rows) {
    printf("%-46s %9.1f %9.1f %9.1f\n", r.name, r.soa, r.rec, r.rec_t);
    d_rec += r.rec - r.soa;
    d_rect += (r.rec_t > 0 ? r.rec_t : r.rec) - r.soa;
  }
  printf("records -> SoA pass (what record mode would save): %.1f us here, 292 us in the library (k_move_unpack<4>)\n", unpack);
  printf("extra time of the three lambdas in record mode: lane = row %+.1f us, lane = column %+.1f us\n", d_rec, d_rect);
  return 0;
}
