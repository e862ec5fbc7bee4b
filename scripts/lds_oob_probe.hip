// Does a ds_read_b128 beyond the workgroup's LDS allocation return zeros on gfx950?  (probe for a zero-row-free padding select)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) unsigned char* lptr_t;
__global__ void probe(unsigned* out, unsigned lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (unsigned i = threadIdx.x; i < lds_bytes / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0xA5A50000u + i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(lptr_t)smem;
  const unsigned addrs[6] = {base + lds_bytes - 16, base + lds_bytes, base + lds_bytes + 4096, 0x3fff0u, 0xffff0u, 0x7ffffff0u};
  for (int k = 0; k < 6; ++k) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addrs[k]) : "memory");
    if (threadIdx.x == 0) { out[4 * k] = v.x; out[4 * k + 1] = v.y; out[4 * k + 2] = v.z; out[4 * k + 3] = v.w; }
  }
}
int main() {
  unsigned* d; hipMalloc(&d, 96 * 4);
  for (unsigned lds : {65536u, 163840u}) {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), lds, 0, d, lds);
    unsigned h[24]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lds=%u err=%s\n", lds, hipGetErrorString(hipDeviceSynchronize()));
    const char* names[6] = {"last16", "end+0", "end+4096", "0x3fff0", "0xffff0", "0x7ffffff0"};
    for (int k = 0; k < 6; ++k) printf("  %-10s %08x %08x %08x %08x\n", names[k], h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]);
  }
  return 0;
}
