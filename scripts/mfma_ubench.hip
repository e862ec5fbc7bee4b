// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 in the conv kernel's register pattern
// (8 independent accumulators, A reused 4x, B reused 2x), 1 or 2 waves per SIMD, optional LDS fragment reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512) void k(const uint4* in, float* out, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  uint4* l = reinterpret_cast<uint4*>(smem);
  for (int i = tid; i < 8192; i += blockDim.x) l[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc[2][4];
  for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 4; ++ni) for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  uint4 a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = in[lane + i * 64];
  for (int i = 0; i < 4; ++i) b[i] = in[lane + (2 + i) * 64];
  long long t0 = __builtin_readcyclecounter();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3 && (it & 7) == 0) {   // 32 KB weight-tile DMA per 8 K-steps: 4 x 1 KB pieces per wave (8 waves)
      const unsigned char* src = reinterpret_cast<const unsigned char*>(in) + (((it >> 3) & 3) * 32768);
      for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src + piece * 1024 + lane * 16), "s"(lds0 + 65536 + piece * 1024) : "memory");
      }
    }
    if ((MODE == 4 || MODE == 5) && (it & 3) == 0) {   // 1x1-layer pattern: 80 KB (A 64 + B 16) per 4 K-steps, L2-resident source
      const unsigned char* src = reinterpret_cast<const unsigned char*>(in) + (((it >> 2) & 1) * 65536) + ((blockIdx.x & 7) * 16384);
      for (int q = 0; q < 10; ++q) {
        const int piece = wave * 10 + q;
        if (MODE == 4) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src + (piece & 63) * 1024 + lane * 16), "s"(lds0 + 49152 + piece * 1024) : "memory");
        } else {
          const uint4 v = *reinterpret_cast<const uint4*>(src + (piece & 63) * 1024 + lane * 16);
          *reinterpret_cast<uint4*>(smem + 49152 + piece * 1024 + lane * 16) = v;
        }
      }
    }
    if (MODE >= 1) {   // fragment reads from LDS each K-step, like the conv kernel (prefetch distance 1 handled by compiler)
      const int base = ((it * 6) & 127) * 64 + lane;
      for (int i = 0; i < 2; ++i) a[i] = l[base + i * 64];
      for (int i = 0; i < 4; ++i) b[i] = l[base + (2 + i) * 64];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
    if ((MODE == 2 || MODE == 3) && (it & 7) == 7) {
      if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (MODE >= 4 && (it & 3) == 3) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 4; ++ni) for (int r = 0; r < 16; ++r) s += acc[mi][ni][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// The same wave tile (64 rows x 128 columns) built from v_mfma_f32_16x16x32_bf16: 4 x 8 tiles of 16x16, one iteration = 32 k =
// 32 MFMAs, 4 A + 8 B fragment reads -- the instruction a 16-row padding-skip granularity would need.
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(512) void k16(const uint4* in, float* out, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  uint4* l = reinterpret_cast<uint4*>(smem);
  for (int i = tid; i < 8192; i += blockDim.x) l[i] = in[i & 1023];
  __syncthreads();
  f32x4 acc[4][8];
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 8; ++ni) for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
  uint4 a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = in[lane + i * 64];
  for (int i = 0; i < 8; ++i) b[i] = in[lane + (4 + i) * 64];
  long long t0 = __builtin_readcyclecounter();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3 && (it & 3) == 0) {   // 32 KB weight-tile DMA per 4 iterations (= 8 K-steps of 16)
      const unsigned char* src = reinterpret_cast<const unsigned char*>(in) + (((it >> 2) & 3) * 32768);
      for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src + piece * 1024 + lane * 16), "s"(lds0 + 65536 + piece * 1024) : "memory");
      }
    }
    if (MODE >= 1) {
      const int base = ((it * 12) & 127) * 64 + lane;
      for (int i = 0; i < 4; ++i) a[i] = l[base + i * 64];
      for (int i = 0; i < 8; ++i) b[i] = l[base + (4 + i) * 64];
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
    if ((MODE == 2 || MODE == 3) && (it & 3) == 3) {
      if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 8; ++ni) for (int r = 0; r < 4; ++r) s += acc[mi][ni][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  uint4* in; float* out; long long* cyc;
  hipMalloc(&in, 1 << 20); hipMemset(in, 0x3c, 1 << 20);
  const char* names[6] = {"register operands", "LDS frag reads", "+ barrier / 8 K-steps", "+ 32 KB LDS-DMA / 8 K-steps",
                          "1x1 pattern: 80 KB LDS-DMA + barrier / 4 K-steps", "1x1 pattern: 80 KB via VGPR + ds_write / 4 K-steps"};
  hipMalloc(&out, 256 * 512 * 4 * 8); hipMalloc(&cyc, 8);
  const int iters = 4000;
  for (int mode = 0; mode < 6; ++mode)
    for (int threads : {256, 512}) {
      if (mode >= 3 && threads != 512) continue;
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 131072, 0, in, out, iters, cyc);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 131072, 0, in, out, iters, cyc);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 131072, 0, in, out, iters, cyc);
        else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 131072, 0, in, out, iters, cyc);
        else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 131072, 0, in, out, iters, cyc);
        else hipLaunchKernelGGL(k<5>, dim3(256), dim3(threads), 131072, 0, in, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double mf = (double)iters * 8;                      // MFMAs per wave
        const double waves_per_simd = threads / 256.0;
        if (rep) printf("mode %d (%s) waves/SIMD %.0f: %.1f s_memtime ticks per MFMA per wave, %.1f TFLOP/s chip (%.3f ms)\n", mode,
               names[mode], waves_per_simd, c / mf,
               256.0 * (threads / 64) * mf * 32768 / (ms * 1e-3) / 1e12, ms);
      }
    }
  for (int mode = 0; mode < 4; ++mode)
    for (int threads : {256, 512}) {
      if (mode >= 3 && threads != 512) continue;
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int it16 = iters / 2;
        if (mode == 0) hipLaunchKernelGGL(k16<0>, dim3(256), dim3(threads), 131072, 0, in, out, it16, cyc);
        else if (mode == 1) hipLaunchKernelGGL(k16<1>, dim3(256), dim3(threads), 131072, 0, in, out, it16, cyc);
        else if (mode == 2) hipLaunchKernelGGL(k16<2>, dim3(256), dim3(threads), 131072, 0, in, out, it16, cyc);
        else hipLaunchKernelGGL(k16<3>, dim3(256), dim3(threads), 131072, 0, in, out, it16, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)it16 * 32;                      // 16x16x32 MFMAs per wave (16384 FLOP each)
        if (rep) printf("16x16x32 mode %d (%s) waves/SIMD %.0f: %.1f TFLOP/s chip (%.3f ms)\n", mode, names[mode], threads / 256.0,
               256.0 * (threads / 64) * mf * 16384 / (ms * 1e-3) / 1e12, ms);
      }
    }
  return 0;
}
