// compat/Kokkos_Random.hpp -- NOT Kokkos (see compat/Kokkos_Core.hpp).  Random_XorShift64_Pool as the reference's
// drivers use it (performance_tests/ps_combo160.cpp:190-221, particle_structs/test/Distribute.h:37-47,
// Distribute.cpp:75-130): pool(seed); gen = pool.get_state(); gen.drand(max) / gen.urand(max) / ...;
// pool.free_state(gen).  Marsaglia's xorshift64* (Vigna 2016: x ^= x>>12; x ^= x<<25; x ^= x>>27; x *
// 0x2545F4914F6CDD1D), one state per thread slot seeded through splitmix64 of (seed, slot).  Streams are
// deterministic for a given launch shape; they are not Kokkos's streams (which differ per backend anyway).
#pragma once
#include "Kokkos_Core.hpp"

namespace Kokkos {
struct Random_XorShift64 {
  unsigned long long state_;
  int slot_;
  __host__ __device__ unsigned long long urand64() {
    state_ ^= state_ >> 12;
    state_ ^= state_ << 25;
    state_ ^= state_ >> 27;
    return state_ * 2685821657736338717ULL;
  }
  __host__ __device__ unsigned urand() { return (unsigned)(urand64() >> 32); }
  __host__ __device__ unsigned urand(unsigned range) { return range ? urand() % range : 0u; }  // [0, range)
  __host__ __device__ unsigned urand(unsigned start, unsigned end) { return start + urand(end - start); }
  __host__ __device__ int rand() { return (int)(urand() >> 1); }
  __host__ __device__ int rand(int range) { return range > 0 ? rand() % range : 0; }
  __host__ __device__ int rand(int start, int end) { return start + rand(end - start); }
  __host__ __device__ double drand() { return (double)(urand64() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
  __host__ __device__ double drand(double range) { return drand() * range; }
  __host__ __device__ double drand(double start, double end) { return start + drand() * (end - start); }
  __host__ __device__ float frand() { return (float)(urand64() >> 40) * (1.0f / 16777216.0f); }
  __host__ __device__ float frand(float range) { return frand() * range; }
  __host__ __device__ float frand(float start, float end) { return start + frand() * (end - start); }
  __host__ __device__ double normal() {  // Marsaglia's polar method
    double S = 2.0, U = 0.0;
    while (S >= 1.0 || S == 0.0) {
      U = 2.0 * drand() - 1.0;
      const double V = 2.0 * drand() - 1.0;
      S = U * U + V * V;
    }
    return U * ::sqrt(-2.0 * ::log(S) / S);
  }
  __host__ __device__ double normal(double mean, double std_dev = 1.0) { return mean + normal() * std_dev; }
};

template <class Space = DefaultExecutionSpace>
class Random_XorShift64_Pool {
 public:
  typedef Random_XorShift64 generator_type;
  Random_XorShift64_Pool() : n_(0), s_(nullptr) {}
  explicit Random_XorShift64_Pool(unsigned long long seed) { init(seed, kSlots); }
  void init(unsigned long long seed, int nslots) {
    n_ = nslots;
    std::vector<unsigned long long> h((size_t)n_);
    for (int i = 0; i < n_; ++i) {  // splitmix64
      unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i + 1);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
      z ^= z >> 31;
      h[(size_t)i] = z ? z : 0x2545F4914F6CDD1DULL;  // (the all-zero state is the one fixed point)
    }
    states_ = ::pumipic::View<unsigned long long>::uninitialized((size_t)n_);
    states_.from_host(h.data());
    s_ = states_.data();
  }
  // a thread's slot is its global index modulo the pool size; 2^20 slots exceed the 524 288 threads a MI355X
  // keeps resident (256 CUs x 2048), so two live threads do not share a slot
  __host__ __device__ Random_XorShift64 get_state() const {
#if defined(__HIP_DEVICE_COMPILE__)
    const int slot = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) % n_);
#else
    const int slot = 0;
#endif
    return get_state(slot);
  }
  __host__ __device__ Random_XorShift64 get_state(int slot) const {
    Random_XorShift64 g;
    g.slot_ = slot;
#if defined(__HIP_DEVICE_COMPILE__)
    g.state_ = s_[slot];
#else
    g.state_ = 0x2545F4914F6CDD1DULL + (unsigned long long)slot;
#endif
    return g;
  }
  __host__ __device__ void free_state(const Random_XorShift64& g) const {
#if defined(__HIP_DEVICE_COMPILE__)
    s_[g.slot_] = g.state_;
#else
    (void)g;
#endif
  }

 private:
  static constexpr int kSlots = 1 << 20;
  int n_;
  unsigned long long* s_;
  ::pumipic::View<unsigned long long> states_;
};
}  // namespace Kokkos
