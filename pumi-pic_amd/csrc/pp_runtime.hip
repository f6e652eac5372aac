// pp_runtime.hip -- device selection, stream, memory and event helpers of the C-ABI.
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <unordered_map>
#include "pp_internal.hpp"

namespace pp {
// roctx ranges (the reference's Kokkos::Profiling::pushRegion / popRegion, e.g. adjacency.tpp:480,613).
// Resolved at run time and only when PP_ROCTX=1: rocprofv3 --marker-trace reads the ranges of
// librocprofiler-sdk-roctx (older tools: libroctx64).
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static bool roctx_ready() {
  static int state = -1;
  if (state >= 0) return state == 1;
  state = 0;
  const char* on = getenv("PP_ROCTX");
  if (!on || atoi(on) == 0) return false;
  const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4",
                         "libroctx64.so"};
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
    if (g_roctx_push && g_roctx_pop) {
      state = 1;
      return true;
    }
  }
  return false;
}
void range_push(const char* name) {
  if (roctx_ready()) (void)g_roctx_push(name);
}
void range_pop() {
  if (roctx_ready()) (void)g_roctx_pop();
}

static thread_local std::string g_err;
static hipStream_t g_stream = nullptr;
static bool g_init = false;

unsigned long long next_version() {
  static unsigned long long v = 0;
  return ++v;
}
void set_error(const std::string& msg) { g_err = msg; }
hipStream_t stream() { return g_stream; }
bool initialised() { return g_init; }

// ---- the pool (see pp_malloc below)
__global__ void k_fill64(unsigned long long* __restrict__ p, unsigned long long v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
namespace {
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;     // handed out: pointer -> block size
  std::multimap<size_t, void*> idle;          // cached: block size -> pointer
  size_t live_bytes = 0, idle_bytes = 0, limit = 0;
  long long hits = 0, misses = 0;
};
Pool& pool() {
  static Pool* p = new Pool();  // (never destroyed: frees may arrive from static destructors of the caller)
  return *p;
}
size_t round_block(size_t n) {
  if (n <= 4096) return 4096;
  if (n <= ((size_t)1 << 20)) {  // next power of two
    size_t r = 4096;
    while (r < n) r <<= 1;
    return r;
  }
  return (n + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);  // 2 MiB granules
}
void trim_locked(Pool& P, size_t keep) {
  while (P.idle_bytes > keep && !P.idle.empty()) {
    auto it = std::prev(P.idle.end());  // largest first
    (void)hipFree(it->second);
    P.idle_bytes -= it->first;
    P.idle.erase(it);
  }
}
}  // namespace
void* pool_alloc(size_t bytes) {
  Pool& P = pool();
  const size_t want = round_block(bytes);
  std::lock_guard<std::mutex> lk(P.mu);
  auto it = P.idle.lower_bound(want);
  if (it != P.idle.end() && it->first <= want + want / 8) {
    void* p = it->second;
    const size_t sz = it->first;
    P.idle.erase(it);
    P.idle_bytes -= sz;
    P.live[p] = sz;
    P.live_bytes += sz;
    ++P.hits;
    return p;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess && !P.idle.empty()) {  // out of memory with blocks cached: give them back, try again
    (void)hipGetLastError();
    (void)hipStreamSynchronize(g_stream);
    trim_locked(P, 0);
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) {
    set_error(std::string("pp_malloc: hipMalloc of ") + std::to_string(want) + " bytes: " + hipGetErrorString(e));
    return nullptr;
  }
  ++P.misses;
  P.live[p] = want;
  P.live_bytes += want;
  return p;
}
int pool_free(void* dev) {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  auto it = P.live.find(dev);
  if (it == P.live.end()) {  // not from pp_malloc (a pointer the caller got from hipMalloc): the runtime's free
    PP_HIP_CHECK(hipFree(dev));
    return PP_OK;
  }
  const size_t sz = it->second;
  P.live.erase(it);
  P.live_bytes -= sz;
  if (!P.limit) {
    // Default cap of the cache: 1/64 of the device memory, at least 1 GiB, at most 4 GiB (4 GiB on a 288-GB MI355X:
    // a hundred per-step arrays of a 10 M-particle driver).  Round 5 kept up to an eighth (36 GB), which other
    // allocators in the process (torch, RCCL, the caller's hipMalloc) could not reclaim.  PP_POOL_LIMIT_MB in the
    // environment or pp_pool_set_limit change it.
    size_t fr = 0, tot = 0;
    const size_t gib = (size_t)1 << 30;
    P.limit = (hipMemGetInfo(&fr, &tot) == hipSuccess) ? std::min(std::max(tot / 64, gib), 4 * gib) : gib;
    if (const char* e = getenv("PP_POOL_LIMIT_MB")) P.limit = (size_t)std::max(0ll, atoll(e)) << 20;
  }
  // (a pp_malloc block that somebody released with hipFree leaves its entry behind; should the runtime hand the
  //  address out again to ANOTHER allocator and that block come back through pp_free, the recorded size is not the
  //  block's: ask the runtime what it is before caching it -- round-5 advisor)
  {
    void* base = nullptr;
    size_t real = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &real, (hipDeviceptr_t)dev) != hipSuccess || base != dev || real < sz) {
      (void)hipGetLastError();
      PP_HIP_CHECK(hipFree(dev));
      return PP_OK;
    }
  }
  if (sz > P.limit || P.limit == 0) {
    PP_HIP_CHECK(hipFree(dev));
    return PP_OK;
  }
  P.idle.emplace(sz, dev);
  P.idle_bytes += sz;
  if (P.idle_bytes > P.limit) {  // (blocks in the cache may still be read by queued kernels: drain first)
    (void)hipStreamSynchronize(g_stream);
    trim_locked(P, P.limit / 2);
  }
  return PP_OK;
}
int pool_trim() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  if (!P.idle.empty()) PP_HIP_CHECK(hipStreamSynchronize(g_stream));
  trim_locked(P, 0);
  return PP_OK;
}
void pool_set_limit(size_t bytes) {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  P.limit = bytes ? bytes : 1;  // (0 would mean "not chosen yet": one byte caches nothing)
  if (P.idle_bytes > P.limit) {
    (void)hipStreamSynchronize(g_stream);
    trim_locked(P, P.limit / 2);
  }
}
void pool_stats(size_t* live_bytes, size_t* cached_bytes, long long* hits, long long* misses) {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  if (live_bytes) *live_bytes = P.live_bytes;
  if (cached_bytes) *cached_bytes = P.idle_bytes;
  if (hits) *hits = P.hits;
  if (misses) *misses = P.misses;
}
}  // namespace pp

extern "C" {

const char* pp_last_error(void) { return pp::g_err.c_str(); }
const char* pp_version(void) { return "pumipic_hip 0.1 (gfx950)"; }

int pp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pp_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    pp::set_error("pp_init: no HIP device visible -- libpumipic_hip has no CPU fallback");
    return PP_EHIP;
  }
  PP_REQUIRE(device >= 0 && device < n, "pp_init: device index out of range");
  PP_HIP_CHECK(hipSetDevice(device));
  if (!pp::g_stream) PP_HIP_CHECK(hipStreamCreateWithFlags(&pp::g_stream, hipStreamNonBlocking));
  pp::g_init = true;
  return PP_OK;
}

void* pp_stream(void) { return (void*)pp::g_stream; }

// the HIP runtime's sticky "last error" of this thread, without clearing it (0 = none).  A launch that
// failed unnoticed would sit there and be picked up by the next library that checks (RCCL does after
// its own launches); tests/conftest.py asserts after every GPU test that nothing is left behind.
int pp_peek_hip_error(const char** msg_out) {
  const hipError_t e = hipPeekAtLastError();
  if (msg_out) *msg_out = e == hipSuccess ? "" : hipGetErrorString(e);
  return (int)e;
}

int pp_sync(void) {
  PP_HIP_CHECK(hipStreamSynchronize(pp::g_stream));
  return PP_OK;
}

// ---- device memory pool behind pp_malloc / pp_free.
// The reference's drivers allocate per step (a fresh elem_ids array per search, new_elems / new_procs per
// migration: test/pseudoXGCm.cpp:142-146, src/pumipic_ptcl_ops.hpp:56-60) and Kokkos serves those from its own
// pools; hipMalloc / hipFree cost 0.1-1 ms apiece at these sizes and hipFree drains the device.  A freed block
// goes to a size-keyed free list and is handed out again to the next request it fits (within 1/8 of its size).
// Safe without events because everything the library and the mirror headers enqueue runs on ONE stream
// (pp_stream()): whatever still reads a freed block was enqueued before whatever the next owner enqueues.
// Cached bytes are bounded (an eighth of the device memory, at least 1 GiB); an allocation failure empties the
// cache and retries.
void* pp_malloc(size_t bytes) {
  return pp::pool_alloc(bytes ? bytes : 1);
}
int pp_free(void* dev) {
  if (!dev) return PP_OK;
  pp::gyro_map_invalidate(dev, 1);
  return pp::pool_free(dev);
}
int pp_pool_trim(void) { return pp::pool_trim(); }
int pp_pool_set_limit(size_t bytes) {
  pp::pool_set_limit(bytes);
  return PP_OK;
}
int pp_pool_stats(size_t* live_bytes, size_t* cached_bytes, long long* hits, long long* misses) {
  pp::pool_stats(live_bytes, cached_bytes, hits, misses);
  return PP_OK;
}
// fill `count` items of `pattern_bytes` (1, 2, 4 or 8) bytes each with the pattern at pattern_host, on the
// library stream: Kokkos::View / Omega_h::Write<T>(n, value) without a host array and a copy
int pp_fill(void* dev, const void* pattern_host, int pattern_bytes, size_t count) {
  PP_REQUIRE(pattern_host && (pattern_bytes == 1 || pattern_bytes == 2 || pattern_bytes == 4 || pattern_bytes == 8),
             "pp_fill: pattern of 1, 2, 4 or 8 bytes");
  if (!count) return PP_OK;
  PP_REQUIRE(dev, "pp_fill: null device pointer");
  pp::gyro_map_invalidate(dev, count * (size_t)pattern_bytes);
  unsigned long long v = 0;
  memcpy(&v, pattern_host, (size_t)pattern_bytes);
  bool same = true;  // every byte equal (0, -1, ...): the runtime's byte fill
  for (int i = 1; i < pattern_bytes; ++i) same = same && ((v >> (8 * i)) & 0xff) == (v & 0xff);
  if (same) {
    PP_HIP_CHECK(hipMemsetAsync(dev, (int)(v & 0xff), count * (size_t)pattern_bytes, pp::g_stream));
  } else if (pattern_bytes == 2) {
    PP_HIP_CHECK(hipMemsetD16Async((hipDeviceptr_t)dev, (unsigned short)v, count, pp::g_stream));
  } else if (pattern_bytes == 4) {
    PP_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)dev, (int)(unsigned)v, count, pp::g_stream));
  } else {
    pp::k_fill64<<<(unsigned)std::min<size_t>((count + 255) / 256, 4096), 256, 0, pp::g_stream>>>(
        (unsigned long long*)dev, v, count);
    PP_LAUNCH_CHECK();
  }
  return PP_OK;
}
int pp_memcpy_h2d(void* dev, const void* host, size_t bytes) {
  if (!bytes) return PP_OK;
  pp::gyro_map_invalidate(dev, bytes);
  PP_HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, pp::g_stream));
  PP_HIP_CHECK(hipStreamSynchronize(pp::g_stream));
  return PP_OK;
}
int pp_memcpy_d2h(void* host, const void* dev, size_t bytes) {
  if (!bytes) return PP_OK;
  PP_HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, pp::g_stream));
  PP_HIP_CHECK(hipStreamSynchronize(pp::g_stream));
  return PP_OK;
}
int pp_memset(void* dev, int value, size_t bytes) {
  if (!bytes) return PP_OK;
  pp::gyro_map_invalidate(dev, bytes);
  PP_HIP_CHECK(hipMemsetAsync(dev, value, bytes, pp::g_stream));
  return PP_OK;
}

void* pp_event_create(void) {
  hipEvent_t ev = nullptr;
  PP_HIP_CHECK_NULL(hipEventCreate(&ev));
  return (void*)ev;
}
int pp_event_record(void* ev) {
  PP_HIP_CHECK(hipEventRecord((hipEvent_t)ev, pp::g_stream));
  return PP_OK;
}
float pp_event_elapsed_ms(void* start, void* stop) {
  float ms = -1.f;
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
  return ms;
}
int pp_event_destroy(void* ev) {
  if (ev) PP_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
  return PP_OK;
}

int pp_range_push(const char* name) {
  pp::range_push(name ? name : "");
  return PP_OK;
}
int pp_range_pop(void) {
  pp::range_pop();
  return PP_OK;
}

}  // extern "C"
