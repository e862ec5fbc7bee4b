// Round 6, step 1 of the "FP8 cross terms" question: what does the matrix pipe sustain on the instruction MIX
// an expert tap layer would issue if the two cross terms lo*W_hi + hi*W_lo of the three-product scheme went through ONE
// block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 = two taps x [lo8 | hi8] of 16 channels) into the SAME fp32 accumulators
// as the f16 hi*W_hi products?  Registers only (no LDS traffic), one 512-thread workgroup per CU like conv8n_kernel, a wave owns
// 4 tiles x 2 column tiles = 8 accumulators; operands with the mantissa entropy of real data (random f16 / random e4m3 bytes).
// Per PAIR of taps and (tile, column tile):
//   X3        3 + 3 v_mfma_f32_32x32x16_f16                          (today's pair K loop, conv8n.hip tile_mma)
//   X8        2 f16 + 1 scaled fp8 K=64                              (ideal pairing: 2.5 fp8 instructions per row of 5 taps)
//   X8r       5 f16 + 3 fp8 per ROW of 5 taps                        (the last tap of a row pairs with nothing)
//   X6        2 f16 + 1 scaled fp6 (e2m3) K=64                       (for information: FP6 runs at the FP4 rate on gfx950)
//   F16 / FP8 / FP6 alone: the pipe's own rate per format on such data
// Printed: ms, instructions / s, and the speed-up of the useful multiply rate over X3 (the gate of step 1: >= 1.35x).
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_x8_ubench.hip -o scripts/mfma_x8_ubench      run: scripts/mfma_x8_ubench [0|1]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void mma16(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <int FMT>   // 0: e4m3, 2: e2m3 (fp6; the upper 2 of the 8 operand registers are ignored)
__device__ __forceinline__ void mma8(f32x16& acc, const i32x8& a, const i32x8& b, int sa, int sb) {
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, FMT, FMT, 0, sa, 0, sb);
}

enum { X3 = 0, X8 = 1, X8R = 2, X6 = 3, F16 = 4, FP8 = 5, FP6 = 6 };

template <int V>
__global__ __launch_bounds__(512) void k(const unsigned char* in, float* out, int rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  f32x16 acc[4][2];
  for (int j = 0; j < 4; ++j) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;
  uint4 ah[4], al[4], bh[2], bl[2];
  i32x8 a8[4], b8[2];
  const uint4* src = reinterpret_cast<const uint4*>(in) + (size_t)tid * 32;
  for (int j = 0; j < 4; ++j) { ah[j] = src[j]; al[j] = src[4 + j]; }
  for (int n = 0; n < 2; ++n) { bh[n] = src[8 + n]; bl[n] = src[10 + n]; }
  const i32x8* src8 = reinterpret_cast<const i32x8*>(in + (1 << 19)) + (size_t)tid * 8;
  for (int j = 0; j < 4; ++j) a8[j] = src8[j];
  for (int n = 0; n < 2; ++n) b8[n] = src8[4 + n];
  int sa = 0x7f - 3, sb = 0x7f - 8;            // E8M0 block scales 2^-3, 2^-8 (per-layer constants in the real kernel)
  asm volatile("" : "+v"(sa), "+v"(sb));
  if (smem[tid] == 77) sa = 1;                 // keeps the LDS allocation (one workgroup per CU) alive
  for (int g = 0; g < rows; ++g) {             // one row of 5 taps per trip (X8: two rows = 5 pairs per two trips -> 10 taps, 5 fp8)
#pragma unroll
    for (int u = 0; u < 5; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (V == X3) {
          mma16(acc[j][0], ah[j], bh[0]); mma16(acc[j][1], ah[j], bh[1]);
          mma16(acc[j][0], al[j], bh[0]); mma16(acc[j][1], al[j], bh[1]);
          mma16(acc[j][0], ah[j], bl[0]); mma16(acc[j][1], ah[j], bl[1]);
        } else if (V == F16) {
          mma16(acc[j][0], ah[j], bh[0]); mma16(acc[j][1], ah[j], bh[1]);
        } else if (V == FP8 || V == FP6) {
          mma8<V == FP8 ? 0 : 2>(acc[j][0], a8[j], b8[0], sa, sb); mma8<V == FP8 ? 0 : 2>(acc[j][1], a8[j], b8[1], sa, sb);
        } else {
          mma16(acc[j][0], ah[j], bh[0]); mma16(acc[j][1], ah[j], bh[1]);
          const bool cross = (V == X8R) ? (u == 1 || u == 3 || u == 4) : (((g * 5 + u) & 1) == 1);
          if (cross) { mma8<V == X6 ? 2 : 0>(acc[j][0], a8[j], b8[0], sa, sb); mma8<V == X6 ? 2 : 0>(acc[j][1], a8[j], b8[1], sa, sb); }
        }
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[j][n][r];
  out[blockIdx.x * 512 + tid] = s;
}

static double t_x3 = 0;
template <int V>
void run(const char* name, const unsigned char* in, float* out, double n16, double n8) {   // instructions per tap and (tile, column tile)
  const int rows = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), dim3(1024), dim3(512), 163840, 0, in, out, rows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double slots = 1024.0 * 8 * rows * 5 * 8;          // (wave, tap, tile, column tile)
  const double tf16 = slots * n16 * 32768 / (best * 1e-3) / 1e12, tf8 = slots * n8 * 131072 / (best * 1e-3) / 1e12;
  if (V == X3) t_x3 = best;
  printf("%-6s %8.3f ms   f16 %7.1f TFLOP/s + f8f6f4 %7.1f TFLOP/s issued", name, best, tf16, tf8);
  if (V <= X6) printf("   useful multiply rate vs X3: %.3fx", t_x3 / best);
  printf("\n");
}

int main(int argc, char** argv) {
  const int data_mode = argc > 1 ? atoi(argv[1]) : 0;      // 0: random, 1: zeros
  unsigned char* in; float* out;
  hipMalloc(&in, 1 << 20); hipMalloc(&out, 1024 * 512 * 4);
  unsigned char* h = (unsigned char*)calloc(1 << 20, 1);
  srand(1);
  if (data_mode == 0) {
    uint16_t* h16 = (uint16_t*)h;
    for (int i = 0; i < (1 << 18); ++i) h16[i] = (uint16_t)(((rand() & 1) << 15) | ((11 + rand() % 4) << 10) | (rand() & 1023));
    for (int i = (1 << 19); i < (1 << 20); ++i) {          // random e4m3 (as fp6: random e2m3 sextets), no NaN encodings
      h[i] = (unsigned char)(((rand() & 1) << 7) | ((4 + rand() % 6) << 3) | (rand() & 7));   // |x| in [2^-3, 2^3): sums stay finite
    }
  }
  printf("data: %s\n", data_mode == 0 ? "random f16 / random e4m3 bytes" : "zeros");
  hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice);
  run<X3>("X3", in, out, 3, 0);
  run<X8>("X8", in, out, 1, 0.5);
  run<X8R>("X8r", in, out, 1, 0.6);
  run<X6>("X6", in, out, 1, 0.5);
  run<F16>("F16", in, out, 1, 0);
  run<FP8>("FP8", in, out, 0, 1);
  run<FP6>("FP6", in, out, 0, 1);
  return 0;
}
