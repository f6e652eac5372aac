// ppMemUsage.hpp -- support/ppMemUsage.hpp:26-35: free / total bytes of the device the library runs on.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "particle_structs.hpp"
typedef pumipic::DeviceSpace DeviceSpace;  // (support/ppMemUsage.hpp:4-10: Kokkos::HIPSpace under PP_USE_HIP)
static inline void getMemUsage(size_t* free, size_t* total) {
  if (hipMemGetInfo(free, total) != hipSuccess) *free = *total = 0;
}
// gpuMemcpy / gpuFree (support/ppMemUsage.hpp:37-62): a functor's device copy
template <typename FunctionType>
FunctionType* gpuMemcpy(FunctionType& fn) {
  FunctionType* fn_d = nullptr;
  if (hipMalloc((void**)&fn_d, sizeof(FunctionType)) != hipSuccess ||
      hipMemcpy(fn_d, &fn, sizeof(FunctionType), hipMemcpyHostToDevice) != hipSuccess)
    return nullptr;
  return fn_d;
}
template <typename FunctionType>
void gpuFree(FunctionType& fn_d) { (void)hipFree(fn_d); }
