// Implicit-GEMM conv3d / fully-connected kernel on the gfx950 matrix cores.
//
// Replaces tf.nn.conv3d + bias_add + inference batch-norm + ReLU
// (utils/tf_util.py:298-311, 491-494) and tf.matmul + bias (+BN, ReLU)
// (utils/tf_util.py:340-351).  BN is folded into weights/bias on the host
// (model.hip), so the epilogue is bias + optional ReLU.
//
// Decomposition (one 512-thread workgroup = 8 wave64):
//   M tile  = 512 GEMM rows = whole points (1 point at 8^3, 8 at 4^3, 64 at 2^3, 512 for FC),
//             so every tap of every output voxel finds its input row inside the tile:
//             the K-chunk of the input is staged into LDS ONCE and re-read for all k^3 taps
//             (the halo never goes back to HBM/L2);
//   N tile  = TN (64 or 128) output channels;
//   K loop  = input-channel chunks (128 bytes per row: 64 x bf16/f16 or 32 x f32) outer,
//             taps inner; per (chunk, tap) a TN x 128 B weight tile is streamed from L2
//             into a double-buffered LDS slot while the previous tap's MFMAs run.
//   wave w  = rows [64w, 64w+64) x all TN columns: 2 x (TN/32) tiles of 32x32, fp32 accumulate.
// LDS rows are 128 B with the 16-B slot index XOR-swizzled by (row>>1)&7, which makes the
// ds_read_b128 fragment loads of 32 consecutive rows conflict-free (guide: T2).
// Zero padding is a per-lane predicate on the A-fragment load, never a memory halo.
//
// 16-bit mode uses v_mfma_f32_32x32x16_{bf16,f16}; the f32 parity mode uses
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) on the same LDS image: a lane's 16 bytes are
// 8 consecutive k (16-bit) or 4 consecutive k (f32) and A and B use the same k order.
#include "kernels.h"

namespace nesti {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int DT> __device__ __forceinline__ void mma(f32x16& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma<NESTI_BF16>(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<NESTI_F16>(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<NESTI_F32>(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

constexpr int kThreads = 512;
constexpr int kABytes = kTileM * kRowBytes;   // 64 KiB

template <int DT, int TN>
__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + kABytes;
  constexpr int NI = TN / 32;
  constexpr int kBTile = TN * kRowBytes;
  constexpr int kBVec = kBTile / 16 / kThreads;   // uint4 per thread per weight tile
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m_tile = blockIdx.x % p.m_tiles;
  const int n_tile = blockIdx.x / p.m_tiles;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int log2S = p.log2S, log2V = 3 * log2S;
  const int S = 1 << log2S, V = 1 << log2V;
  const long long total_rows = (long long)npts << log2V;
  const long long r0 = (long long)m_tile * kTileM;
  if (r0 >= total_rows) return;

  // ---- A staging: thread -> (slot, rows (tid>>3) + 64 j) --------------------------------
  const int a_slot = tid & 7;
  const int a_row0 = tid >> 3;
  long long a_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long long gr = r0 + a_row0 + 64 * j;
    if (gr < total_rows) {
      long long pt = gr >> log2V;
      const long long vox = gr & (V - 1);
      if (p.point_index) pt = p.point_index[pt];
      a_off[j] = (((pt << log2V) + vox) * p.in_cstride + p.in_coff) * kEsz + a_slot * 16;
    } else {
      a_off[j] = -1;
    }
  }
  const int a_dst = a_row0 * kRowBytes + ((a_slot ^ ((a_row0 >> 1) & 7)) << 4);   // + j*64*128

  // ---- per-lane fragment coordinates ----------------------------------------------------
  int rz[2], ry[2], rx[2], rrow[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    rrow[mi] = wave * 64 + mi * 32 + (lane & 31);
    const int vox = rrow[mi] & (V - 1);
    rz[mi] = vox >> (2 * log2S);
    ry[mi] = (vox >> log2S) & (S - 1);
    rx[mi] = vox & (S - 1);
  }
  const int khalf = lane >> 5;
  const int b_row = (lane & 31) * kRowBytes;
  const int b_sw = ((lane & 31) >> 1) & 7;

  f32x16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
  const unsigned char* w_tile =
      reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)n_tile * p.n_chunks * p.n_taps * kBTile;

  for (int c = 0; c < p.n_chunks; ++c) {
    // stage A(c) and B(c, tap 0)
    uint4 av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      av[j] = make_uint4(0, 0, 0, 0);
      if (a_off[j] >= 0) av[j] = *reinterpret_cast<const uint4*>(in_b + a_off[j] + (long long)c * kRowBytes);
    }
    uint4 bpre[kBVec];
    {
      const unsigned char* src = w_tile + (size_t)c * p.n_taps * kBTile;
#pragma unroll
      for (int q = 0; q < kBVec; ++q) bpre[q] = *reinterpret_cast<const uint4*>(src + (tid + q * kThreads) * 16);
    }
    __syncthreads();   // every wave is done with the previous chunk's A and B tiles
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4*>(As + a_dst + j * 64 * kRowBytes) = av[j];
#pragma unroll
    for (int q = 0; q < kBVec; ++q) *reinterpret_cast<uint4*>(Bs + (tid + q * kThreads) * 16) = bpre[q];
    __syncthreads();

    for (int t = 0; t < p.n_taps; ++t) {
      const bool more = (t + 1 < p.n_taps);
      if (more) {
        const unsigned char* src = w_tile + ((size_t)c * p.n_taps + t + 1) * kBTile;
#pragma unroll
        for (int q = 0; q < kBVec; ++q) bpre[q] = *reinterpret_cast<const uint4*>(src + (tid + q * kThreads) * 16);
      }
      const unsigned char* Bcur = Bs + (t & 1) * kBTile;
      const int dz = p.tap[t][0], dy = p.tap[t][1], dx = p.tap[t][2];
      const int shift = dz * (1 << (2 * log2S)) + dy * S + dx;
      bool ok[2];
      int a_base[2], a_sw[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        ok[mi] = ((unsigned)(rz[mi] + dz) < (unsigned)S) & ((unsigned)(ry[mi] + dy) < (unsigned)S) &
                 ((unsigned)(rx[mi] + dx) < (unsigned)S);
        const int srow = rrow[mi] + shift;
        a_base[mi] = srow * kRowBytes;
        a_sw[mi] = (srow >> 1) & 7;
      }
      if (__ballot(ok[0] | ok[1]) != 0ull) {   // whole-wave zero taps (z-plane halo): skip the MFMAs
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int slot = kk * 2 + khalf;
          uint4 a[2], b[NI];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            a[mi] = make_uint4(0, 0, 0, 0);
            if (ok[mi]) a[mi] = *reinterpret_cast<const uint4*>(As + a_base[mi] + ((slot ^ a_sw[mi]) << 4));
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            b[ni] = *reinterpret_cast<const uint4*>(Bcur + ni * 32 * kRowBytes + b_row + ((slot ^ b_sw) << 4));
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) mma<DT>(acc[mi][ni], a[mi], b[ni]);
        }
      }
      if (more) {
        unsigned char* Bnext = Bs + ((t + 1) & 1) * kBTile;
#pragma unroll
        for (int q = 0; q < kBVec; ++q) *reinterpret_cast<uint4*>(Bnext + (tid + q * kThreads) * 16) = bpre[q];
      }
      __syncthreads();
    }
  }

  // ---- epilogue: bias + ReLU, transpose through a wave-private LDS scratch, 16-B stores ---
  // (the barrier that ended the last tap guarantees nobody still reads As)
  unsigned char* scratch = smem + wave * 8192;
  const int out_esz = p.out_f32 ? 4 : kEsz;
  const int seg = 64 * out_esz;                 // bytes of one 64-channel row segment
  const int lanes_per_row = seg >> 4;           // 8 or 16
  const int rows_per_iter = 64 / lanes_per_row; // 8 or 4
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int nh = 0; nh < TN / 64; ++nh) {
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int ni = nh * 2 + n2;
        const int col = n2 * 32 + (lane & 31);
        const float bias = p.bias[n_tile * TN + ni * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;
          float v = acc[mi][ni][r] + bias;
          if (p.relu) v = fmaxf(v, 0.f);
          if (out_esz == 4) reinterpret_cast<float*>(scratch)[row * 64 + col] = v;
          else reinterpret_cast<uint16_t*>(scratch)[row * 64 + col] = Elem<DT == NESTI_F32 ? NESTI_BF16 : DT>::from_f32(v);
        }
      }
      for (int it = 0; it < 32 / rows_per_iter; ++it) {
        const int row = it * rows_per_iter + lane / lanes_per_row;
        const int cpos = lane % lanes_per_row;
        const uint4 v = *reinterpret_cast<const uint4*>(scratch + row * seg + cpos * 16);
        const long long gr = r0 + wave * 64 + mi * 32 + row;
        if (gr < total_rows) {
          *reinterpret_cast<uint4*>(out_b + (gr * p.out_cstride + p.out_coff + n_tile * TN + nh * 64) * out_esz + cpos * 16) = v;
        }
      }
    }
  }
}

template <int DT, int TN>
int launch_one(const ConvParams& p, int n_tiles, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds = kABytes + 2 * TN * kRowBytes;
  if (!attr_set) {
    NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<DT, TN>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)(p.m_tiles * n_tiles)), block(kThreads);
  hipLaunchKernelGGL((conv_igemm_kernel<DT, TN>), grid, block, lds, stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_conv(const ConvParams& p, int dtype, int TN, int n_tiles, hipStream_t stream) {
  if (p.m_tiles <= 0 || n_tiles <= 0) return 0;
  if (TN == 128) {
    if (dtype == NESTI_BF16) return launch_one<NESTI_BF16, 128>(p, n_tiles, stream);
    if (dtype == NESTI_F16) return launch_one<NESTI_F16, 128>(p, n_tiles, stream);
    if (dtype == NESTI_F32) return launch_one<NESTI_F32, 128>(p, n_tiles, stream);
  } else if (TN == 64) {
    if (dtype == NESTI_BF16) return launch_one<NESTI_BF16, 64>(p, n_tiles, stream);
    if (dtype == NESTI_F16) return launch_one<NESTI_F16, 64>(p, n_tiles, stream);
    if (dtype == NESTI_F32) return launch_one<NESTI_F32, 64>(p, n_tiles, stream);
  }
  NESTI_FAIL("launch_conv: unsupported dtype / tile");
}

}  // namespace nesti
