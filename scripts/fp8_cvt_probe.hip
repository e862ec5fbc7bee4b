#include <hip/hip_runtime.h>
__global__ void k(const float* in, unsigned* out) {
  float a = in[4*threadIdx.x], b = in[4*threadIdx.x+1], c = in[4*threadIdx.x+2], d = in[4*threadIdx.x+3];
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  out[threadIdx.x] = (unsigned)r;
}
int main() {
  const int n = 64*4; float h[n]; unsigned o[64];
  float vals[] = {0.f, 1.f, -1.f, 448.f, 500.f, -1000.f, 0.0019531f, 0.001f, 0.017f, 0.0156f, 17.f, 18.f, 19.f, 0.3f, 240.f, 250.f, 3.75f, 3.5f, 3.25f, 1e-4f, 464.f, 0.0009765625f};
  for (int i = 0; i < n; ++i) h[i] = vals[i % (sizeof(vals)/4)] * ((i / 22) % 2 ? 1.03f : 1.0f);
  float* di; unsigned* dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) for (int b = 0; b < 4; ++b) printf("%.9g %u\n", h[4*i+b], (o[i] >> (8*b)) & 0xff);
  return 0;
}
