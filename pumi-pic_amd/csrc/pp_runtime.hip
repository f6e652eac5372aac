// pp_runtime.hip -- device selection, stream, memory and event helpers of the C-ABI.
#include <dlfcn.h>
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

void* pp_malloc(size_t bytes) {
  void* p = nullptr;
  PP_HIP_CHECK_NULL(hipMalloc(&p, bytes ? bytes : 1));
  return p;
}
int pp_free(void* dev) {
  if (dev) pp::gyro_map_invalidate(dev, 1);
  if (dev) PP_HIP_CHECK(hipFree(dev));
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
