// How long does the legacy K = 16 form v_mfma_f32_16x16x16_f16 take on gfx950 next to the K = 32 form (16 cycles)?  If it ran
// in 8 cycles, conv4n_kernel's pair loop could issue three K = 16 products instead of two K = 32 ones (one of them half zeros).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a8, b8;
  f16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.001f * (threadIdx.x + i)); b8[i] = (_Float16)(0.002f * (threadIdx.x * 3 + i)); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, cyc, iters);
  }
  hipDeviceSynchronize();
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  // one wave per SIMD (256 threads = 4 waves per block, 4 blocks per CU compete: 1024 blocks / 256 CUs) -> report per-instruction clock64 ticks
  printf("16x16x32 f16: %.2f ticks per MFMA per wave;  16x16x16 f16: %.2f ticks per MFMA per wave (clock64 ticks, same units)\n",
         (double)h[0] / (iters * 8.0), (double)h[1] / (iters * 8.0));
  return 0;
}
