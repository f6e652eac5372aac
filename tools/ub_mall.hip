// Does a buffer written by one kernel come back faster than HBM when the next kernel reads it?  (MI355X: 256 MB of
// Infinity Cache in front of HBM.)  hipcc --offload-arch=gfx950 -O2 tools/ub_mall.hip -o ub_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_write(uint4* p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(v, v + 1, v + 2, (unsigned)i);
}
__global__ void k_read(const uint4* p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 q = p[i];
    acc += q.x ^ q.y ^ q.z ^ q.w;
  }
  if (acc == 0x12345678u) *out = acc;
}
int main(int argc, char** argv) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  unsigned* out;
  hipMalloc(&out, 4);
  uint4 *buf, *big;
  const size_t big_bytes = (size_t)2 << 30;
  hipMalloc(&big, big_bytes);
  hipMemset(big, 1, big_bytes);
  for (int mb : {32, 64, 128, 184, 256, 384, 512}) {
    const size_t bytes = (size_t)mb << 20, n = bytes / 16;
    hipMalloc(&buf, bytes);
    float warm = 0, cold = 0;
    for (int rep = 0; rep < 5; ++rep) {
      float ms;
      // warm: read straight after the write
      k_write<<<2048, 256>>>(buf, n, rep);
      hipEventRecord(a);
      k_read<<<2048, 256>>>(buf, n, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      if (rep) warm += ms;
      // cold: 2 GB of other data read in between
      k_write<<<2048, 256>>>(buf, n, rep);
      k_read<<<2048, 256>>>(big, big_bytes / 16, out);
      hipEventRecord(a);
      k_read<<<2048, 256>>>(buf, n, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      if (rep) cold += ms;
    }
    printf("%4d MB: read after write %7.1f us (%5.2f TB/s)   read after 2 GB of other reads %7.1f us (%5.2f TB/s)\n", mb,
           warm / 4 * 1e3, bytes / (warm / 4 * 1e-3) / 1e12, cold / 4 * 1e3, bytes / (cold / 4 * 1e-3) / 1e12);
    hipFree(buf);
  }
  return 0;
}
