// ppMemUsage.hpp -- support/ppMemUsage.hpp:26-35: free / total bytes of the device the library runs on.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
static inline void getMemUsage(size_t* free, size_t* total) {
  if (hipMemGetInfo(free, total) != hipSuccess) *free = *total = 0;
}
