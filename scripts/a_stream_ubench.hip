// Micro-benchmark for the 1x1x1-layer input path: how fast can a workgroup pull its 512-row x 128-B
// input chunks out of a channels-last activation tensor that lives in HBM (row stride = Cin bytes),
// while issuing the layer's MFMAs?
//   mode 0  LDS-DMA (global_load_lds_dwordx4) into a double-buffered LDS tile + barrier per chunk
//           (what conv_igemm_kernel<KPIPE> does), fragments read back with ds_read_b128;
//   mode 1  fragments loaded straight into VGPRs (global_load_dwordx4, lane = (row, k-half)),
//           next chunk prefetched into a second register set, no LDS and no barrier for A;
//   mode 2  like 1 with a prefetch distance of two chunks.
// Every mode issues 32 MFMAs per wave per chunk (2 x 4 tiles x 4 K-steps), B fragments from LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) unsigned char* lptr_t;

__device__ __forceinline__ void glds16(const unsigned char* src, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void mma8(f32x16 (&acc)[2][4], const uint4 (&a)[2], const uint4 (&b)[4]) {
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni], 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* in, float* out, int tiles_per_wg, int row_bytes, int n_chunks, int reread) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  unsigned char* Bs = smem + 131072;   // 16 KiB static "weight tile"
  for (int i = tid; i < 1024; i += 512) reinterpret_cast<uint4*>(Bs)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  f32x16 acc[2][4];
  for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 4; ++ni) for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int khalf = lane >> 5;
  auto load_b = [&](int kk, uint4 (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      b[ni] = *reinterpret_cast<const uint4*>(Bs + (ni * 32 + (lane & 31)) * 128 + (((kk * 2 + khalf) ^ (((lane & 31) >> 1) & 7)) << 4));
  };
  for (int t = 0; t < tiles_per_wg; ++t) {
    // like the conv kernel: the 8 blocks of a dispatch group sit on the 8 XCDs, each XCD re-reads its tile `reread` times
    const size_t tile = ((size_t)((blockIdx.x >> 3) / reread) * 8 + (blockIdx.x & 7)) * tiles_per_wg + t;
    const unsigned char* base = in + tile * 512 * (size_t)row_bytes;
    if (MODE == 0) {
      const unsigned char* src[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = (wave * 8 + j) * 8 + (lane >> 3);
        src[j] = base + (size_t)row * row_bytes + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
      }
      auto stage = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(src[j] + c * 128, lds0 + buf * 65536 + (wave * 8 + j) * 1024);
      };
      stage(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int c = 0; c < n_chunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < n_chunks) stage(c + 1, cur ^ 1);
        const unsigned char* A = smem + cur * 65536;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          uint4 a[2], b[4];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const int row = wave * 64 + mi * 32 + (lane & 31);
            a[mi] = *reinterpret_cast<const uint4*>(A + row * 128 + (((kk * 2 + khalf) ^ ((row >> 1) & 7)) << 4));
          }
          load_b(kk, b);
          mma8(acc, a, b);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    } else {
      constexpr int D = (MODE == 1) ? 1 : 2;      // prefetch distance in chunks
      const unsigned char* rowp[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) rowp[mi] = base + (size_t)(wave * 64 + mi * 32 + (lane & 31)) * row_bytes + khalf * 16;
      uint4 a[D + 1][4][2];
      auto fetch = [&](int c, uint4 (&dst)[4][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) dst[kk][mi] = *reinterpret_cast<const uint4*>(rowp[mi] + c * 128 + kk * 32);
      };
#pragma unroll
      for (int d = 0; d < D; ++d) if (d < n_chunks) fetch(d, a[d]);
      // n_chunks is a multiple of D+1 in this benchmark so the register ring can be indexed statically
      for (int c0 = 0; c0 < n_chunks; c0 += D + 1) {
#pragma unroll
        for (int u = 0; u <= D; ++u) {
          const int c = c0 + u;
          if (c + D < n_chunks) fetch(c + D, a[(u + D) % (D + 1)]);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            uint4 b[4];
            load_b(kk, b);
            mma8(acc, a[u][kk], b);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
  float s = 0;
  for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 4; ++ni) for (int r = 0; r < 16; ++r) s += acc[mi][ni][r];
  out[blockIdx.x * 512 + tid] = s;
}

int main(int argc, char** argv) {
  const int row_bytes = argc > 1 ? atoi(argv[1]) : 768;       // Cin = 384 bf16
  const int n_chunks = row_bytes / 128;
  const int reread = argc > 2 ? atoi(argv[2]) : 1;            // n_tiles of the layer: how many workgroups read each input tile
  const int tiles_per_wg = 1, wgs = 8192 * reread;             // 8192 tiles x 512 rows x 768 B = 3.2 GB: HBM-resident
  unsigned char* in; float* out;
  const size_t bytes = (size_t)(wgs / reread) * tiles_per_wg * 512 * row_bytes;
  hipMalloc(&in, bytes); hipMemset(in, 0x3c, bytes);
  hipMalloc(&out, (size_t)wgs * 512 * 4);
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 1 && n_chunks % 2) continue;
    if (mode == 2 && n_chunks % 3) continue;
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(512), 147456, 0, in, out, tiles_per_wg, row_bytes, n_chunks, reread);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(512), 147456, 0, in, out, tiles_per_wg, row_bytes, n_chunks, reread);
      else hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(512), 147456, 0, in, out, tiles_per_wg, row_bytes, n_chunks, reread);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
      const double flops = (double)wgs * tiles_per_wg * n_chunks * 8 * 32 * 32768.0;
      if (rep) printf("mode %d row_bytes %d reread %d: %.3f ms  %.2f TB/s HBM input, %.1f GB/s staged per CU, %.0f TFLOP/s\n", mode, row_bytes,
                      reread, ms, bytes / (ms * 1e-3) / 1e12, bytes * (double)reread / (ms * 1e-3) / 1e9 / 256, flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
