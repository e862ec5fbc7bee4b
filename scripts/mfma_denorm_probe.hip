// Does v_mfma_f32_32x32x16_f16 keep subnormal f16 inputs, or flush them to zero?  (probe for the f16 hi+lo pair mode:
// the lo parts of small activations are f16 subnormals.)  Also the f16 conversion of a subnormal result.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(float* out, float a_val, float b_val) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  a[0] = (_Float16)a_val;        // every lane: A[row][k0] = a_val, B[k0][col] = b_val for the lane's first k
  b[0] = (_Float16)b_val;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
  float* d; hipMalloc(&d, 16);
  const float cases[4][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /*2^-20: f16 subnormal*/, 1024.0f},
                             {3.0517578125e-05f /*2^-15: subnormal*/, 1.0f}, {6.103515625e-05f /*2^-14: min normal*/, 1.0f}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
    float h[3]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a=%.6e (as f16 %.6e) b=%.6e -> acc[0]=%.6e  expected(2 lanes' k share row 0? see note) %.6e per k-term\n", c[0], h[1], c[1], h[0], (double)h[1] * h[2]);
  }
  return 0;
}
