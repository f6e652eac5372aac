// compat/Kokkos_Core.hpp -- NOT Kokkos.  The reference's drivers are user code written against Kokkos names
// (test/pseudoXGCm.cpp, test/ellipticalPush.hpp, test/gyroScatter.hpp, performance_tests/ps_combo160.cpp,
// particle_structs/test/Distribute.{h,cpp}); north_star asks that they compile and run unchanged on this library.
// This header gives exactly the names those files spell, each a few lines of HIP on the library's stream
// (pp_stream): a 1-D View is pumipic::View, parallel_for / parallel_reduce / parallel_scan are plain gfx950 kernels
// (wave64 shuffles, one LDS stage), the math functions are the device libm.  No Kokkos source, no backend dispatch.
#pragma once
#include <hip/hip_runtime.h>
#include <cassert>
#include <chrono>
#include <cmath>
#include <string>
#include <thread>
#include <typeinfo>
#include "../pumipic_mpi.hpp"
#include "../pumipic_adjacency.hpp"

#define KOKKOS_LAMBDA [=] __host__ __device__
#define KOKKOS_INLINE_FUNCTION __host__ __device__ inline
#define KOKKOS_FUNCTION __host__ __device__

namespace Kokkos {
typedef ::pumipic::DeviceSpace DefaultExecutionSpace;
typedef ::pumipic::HostSpace DefaultHostExecutionSpace;
typedef ::pumipic::HostSpace HostSpace;
typedef ::pumipic::HostSpace Serial;
typedef ::pumipic::DeviceSpace HIPSpace;
typedef ::pumipic::DeviceSpace HIP;

// Kokkos::View<T*[, Space]>: the device array of the mirror
namespace detail {
template <class T>
struct ViewOf;
template <class T>
struct ViewOf<T*> {
  typedef ::pumipic::View<T> type;
};
}  // namespace detail
template <class T, class... Props>
using View = typename detail::ViewOf<T>::type;

// Kokkos::TeamPolicy<Space>(league, team): only the team size matters (the chunk height C of a Sell-C-sigma)
template <class... Props>
using TeamPolicy = ::pumipic::TeamPolicy;

// one process per GPU: the launcher's LOCAL_RANK picks the device (PP_DEVICE overrides)
inline void initialize() {
  const int device = getenv("PP_DEVICE") ? atoi(getenv("PP_DEVICE")) : (getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0);
  ::pumipic::pp_check(pp_init(device), "Kokkos::initialize");
}
inline void initialize(int&, char**) { initialize(); }
inline void finalize() {}
inline void fence() { ::pumipic::fence(); }
inline void fence(const std::string&) { ::pumipic::fence(); }
typedef ::pumipic::Timer Timer;

namespace Profiling {  // roctx ranges when the library was started with PP_ROCTX=1
inline void pushRegion(const std::string& name) { (void)pp_range_push(name.c_str()); }
inline void popRegion() { (void)pp_range_pop(); }
}  // namespace Profiling

// ---- math (Kokkos_MathematicalFunctions): device libm / host libm
#define PP_KK_MATH1(fn)                                         \
  __host__ __device__ inline double fn(double x) { return ::fn(x); } \
  __host__ __device__ inline float fn(float x) { return ::fn##f(x); } \
  template <class I, class = typename std::enable_if<std::is_integral<I>::value>::type> \
  __host__ __device__ inline double fn(I x) { return ::fn((double)x); }
PP_KK_MATH1(sqrt)
PP_KK_MATH1(sin)
PP_KK_MATH1(cos)
PP_KK_MATH1(tan)
PP_KK_MATH1(exp)
PP_KK_MATH1(log)
PP_KK_MATH1(fabs)
PP_KK_MATH1(floor)
PP_KK_MATH1(ceil)
PP_KK_MATH1(round)
#undef PP_KK_MATH1
template <class A, class B>
__host__ __device__ inline auto pow(A a, B b) -> decltype(::pow((double)a, (double)b)) {
  return ::pow((double)a, (double)b);
}
template <class A, class B>
__host__ __device__ inline double atan2(A a, B b) {
  return ::atan2((double)a, (double)b);
}
template <class T>
__host__ __device__ inline T abs(T a) { return a < 0 ? -a : a; }
template <class T>
__host__ __device__ inline T min(T a, T b) { return b < a ? b : a; }
template <class T>
__host__ __device__ inline T max(T a, T b) { return a < b ? b : a; }

// ---- atomics
template <class T, class U>
__host__ __device__ inline T atomic_fetch_add(T* p, U v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, (T)v);
#else
  const T old = *p;
  *p = old + (T)v;
  return old;
#endif
}
template <class T, class U>
__host__ __device__ inline void atomic_add(T* p, U v) { (void)atomic_fetch_add(p, v); }
template <class T>
__host__ __device__ inline void atomic_increment(T* p) { (void)atomic_fetch_add(p, (T)1); }
template <class T>
__host__ __device__ inline void atomic_inc(T* p) { (void)atomic_fetch_add(p, (T)1); }
template <class T>
__host__ __device__ inline void atomic_dec(T* p) { (void)atomic_fetch_add(p, (T)-1); }
template <class T>
__host__ __device__ inline T atomic_fetch_max(T* p, T v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicMax(p, v);
#else
  const T old = *p;
  if (v > old) *p = v;
  return old;
#endif
}
template <class T>
__host__ __device__ inline T atomic_fetch_min(T* p, T v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicMin(p, v);
#else
  const T old = *p;
  if (v < old) *p = v;
  return old;
#endif
}

// ---- deep_copy / mirrors between Views (device <-> device on the library stream), View <- scalar
template <class T>
inline void deep_copy(::pumipic::View<T> dst, ::pumipic::View<T> src) {
  if (dst.size() && hipMemcpyAsync(dst.data(), src.data(), dst.size() * sizeof(T), hipMemcpyDeviceToDevice,
                                   (hipStream_t)pp_stream()) != hipSuccess)
    ::pumipic::pp_check(PP_EHIP, "deep_copy");
}
template <class T>
inline void deep_copy(::pumipic::View<T> dst, const T& value) {
  if (dst.size()) ::pumipic::pp_check(pp_fill(dst.data(), &value, (int)sizeof(T), dst.size()), "deep_copy");
}

// ---- parallel_for over [0, n)
namespace detail {
template <class F>
__global__ void for_kernel(long long n, F f) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f((int)i);
}
// block-wide inclusive scan support: wave64 shuffles + one LDS stage (256 threads = 4 waves)
template <class T>
__device__ inline T wave_incl_scan(T v) {
  const int lane = threadIdx.x & 63;
  for (int d = 1; d < 64; d <<= 1) {
    const T o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
template <class T>
__device__ inline T wave_sum(T v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}
constexpr int kBlock = 256, kItems = 8;  // one block owns kBlock * kItems consecutive indices
// pass 1 of reduce / scan: each block folds its tile with the functor's "not final" form
template <class T, class F>
__global__ void reduce_tiles(long long n, F f, T* __restrict__ tile_sum) {
  __shared__ T s_w[kBlock / 64];
  const long long base = (long long)blockIdx.x * (kBlock * kItems) + (long long)threadIdx.x * kItems;
  T acc = T();
  for (int k = 0; k < kItems; ++k)
    if (base + k < n) f((int)(base + k), acc);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T t = T();
    for (int w = 0; w < kBlock / 64; ++w) t += s_w[w];
    tile_sum[blockIdx.x] = t;
  }
}
template <class T, class F>
__global__ void scan_tile_sums(long long n, F f, T* __restrict__ tile_sum) {
  __shared__ T s_w[kBlock / 64];
  const long long base = (long long)blockIdx.x * (kBlock * kItems) + (long long)threadIdx.x * kItems;
  T acc = T();
  for (int k = 0; k < kItems; ++k)
    if (base + k < n) f((int)(base + k), acc, false);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T t = T();
    for (int w = 0; w < kBlock / 64; ++w) t += s_w[w];
    tile_sum[blockIdx.x] = t;
  }
}
// one block: exclusive scan of the tile sums in place; the grand total lands in tile_sum[ntiles]
template <class T>
__global__ void scan_of_tiles(int ntiles, T* __restrict__ tile_sum) {
  __shared__ T s_w[kBlock / 64];
  __shared__ T s_carry;
  if (threadIdx.x == 0) s_carry = T();
  __syncthreads();
  for (int b = 0; b < ntiles; b += kBlock) {
    const int i = b + (int)threadIdx.x;
    const T v = i < ntiles ? tile_sum[i] : T();
    T inc = wave_incl_scan(v);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    T off = s_carry;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += s_w[w];
    if (i < ntiles) tile_sum[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == kBlock - 1) s_carry = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[ntiles] = s_carry;
}
// pass 2 of scan: every index is visited in order inside its thread with the running prefix and final = true
template <class T, class F>
__global__ void scan_final(long long n, F f, const T* __restrict__ tile_off) {
  __shared__ T s_w[kBlock / 64];
  const long long base = (long long)blockIdx.x * (kBlock * kItems) + (long long)threadIdx.x * kItems;
  T mine = T();
  for (int k = 0; k < kItems; ++k)
    if (base + k < n) f((int)(base + k), mine, false);
  const T inc = wave_incl_scan(mine);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
  __syncthreads();
  T run = tile_off[blockIdx.x] + inc - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += s_w[w];
  for (int k = 0; k < kItems; ++k)
    if (base + k < n) f((int)(base + k), run, true);
}
// the value type of a reduce / scan functor: its second parameter, by reference
template <class F>
struct arg2 : arg2<decltype(&F::operator())> {};
template <class C, class R, class I, class T, class... Rest>
struct arg2<R (C::*)(I, T&, Rest...) const> {
  typedef T type;
};
inline unsigned tiles_of(long long n) { return (unsigned)((n + kBlock * kItems - 1) / (kBlock * kItems)); }
}  // namespace detail

template <class F>
inline void parallel_for(long long n, const F& f) {
  if (n <= 0) return;
  hipLaunchKernelGGL(detail::for_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)pp_stream(), n, f);
}
template <class F>
inline void parallel_for(const std::string&, long long n, const F& f) { parallel_for(n, f); }

// parallel_reduce(name, n, f(i, T& update), T& result): sum
template <class F, class T>
inline void parallel_reduce(const std::string&, long long n, const F& f, T& result) {
  result = T();
  if (n <= 0) return;
  const unsigned nt = detail::tiles_of(n);
  ::pumipic::View<T> tiles = ::pumipic::View<T>::uninitialized((size_t)nt + 1);
  hipLaunchKernelGGL((detail::reduce_tiles<T, F>), dim3(nt), dim3(detail::kBlock), 0, (hipStream_t)pp_stream(), n, f,
                     tiles.data());
  hipLaunchKernelGGL((detail::scan_of_tiles<T>), dim3(1), dim3(detail::kBlock), 0, (hipStream_t)pp_stream(), (int)nt,
                     tiles.data());
  ::pumipic::pp_check(pp_memcpy_d2h(&result, tiles.data() + nt, sizeof(T)), "parallel_reduce");
}
template <class F, class T>
inline void parallel_reduce(long long n, const F& f, T& result) { parallel_reduce(std::string(), n, f, result); }

// parallel_scan(name, n, f(i, T& update, bool final)): exclusive prefix handed to the final pass, index order
template <class F>
inline void parallel_scan(const std::string&, long long n, const F& f) {
  typedef typename detail::arg2<F>::type T;
  if (n <= 0) return;
  const unsigned nt = detail::tiles_of(n);
  ::pumipic::View<T> tiles = ::pumipic::View<T>::uninitialized((size_t)nt + 1);
  hipLaunchKernelGGL((detail::scan_tile_sums<T, F>), dim3(nt), dim3(detail::kBlock), 0, (hipStream_t)pp_stream(), n,
                     f, tiles.data());
  hipLaunchKernelGGL((detail::scan_of_tiles<T>), dim3(1), dim3(detail::kBlock), 0, (hipStream_t)pp_stream(), (int)nt,
                     tiles.data());
  hipLaunchKernelGGL((detail::scan_final<T, F>), dim3(nt), dim3(detail::kBlock), 0, (hipStream_t)pp_stream(), n, f,
                     tiles.data());
  ::pumipic::fence();  // (tiles is released when this returns)
}
template <class F>
inline void parallel_scan(long long n, const F& f) { parallel_scan(std::string(), n, f); }
template <class F, class T>
inline void parallel_scan(const std::string& name, long long n, const F& f, T& total) {
  parallel_scan(name, n, f);
  parallel_reduce(name, n, [=] __host__ __device__(const int i, T& u) { f(i, u, false); }, total);
}
}  // namespace Kokkos
#define PP_KOKKOS_CORE_DONE
#ifdef PP_ADJACENCY_BODY_DONE
#include "Omega_h_mesh.hpp"  // (see the end of pumipic_adjacency.hpp)
#endif
