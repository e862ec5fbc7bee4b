// k^3-tap conv3d layers with 128-column tiles: the "free-running" implicit-GEMM kernel.
//
// Same arithmetic and the same ConvParams as conv_igemm_kernel (conv.hip); it replaces
// tf.nn.conv3d + bias_add + inference batch-norm + ReLU (+ the 2^3 max-pool that follows the block)
// (utils/tf_util.py:298-311, 424-428, 491-494) for the layers that dominate the network: the 3^3 / 5^3 taps at
// 8^3 and the 2^3 / 4^3 (3^3 / 5^3) taps at 4^3 with N tile 128.
//
// What is different from conv_igemm_kernel:
//   * a wave owns 32 output columns x 256 rows (8 MFMA tiles of 32x32) instead of 128 columns x 64 rows.  Its
//     weight fragment (32 columns x 64 k = 4 KiB per tap) is PRIVATE: it streams from L2 straight into VGPRs
//     (weights are packed fragment-major on the host, one contiguous 1-KiB wave load per K-step), one tap ahead.
//     No weight tile in LDS, no LDS-DMA per tap, and therefore NO workgroup barrier per tap: the waves run freely
//     through the k^3 taps of a channel chunk and only meet once per chunk, when the (double-buffered) input chunk
//     in LDS is swapped.
//   * waves c and c+4 share a SIMD and are the two row halves of column group c: every matrix pipe sees all 16
//     row tiles, so skipping the tiles that a padding tap pushes entirely outside the volume shortens every pipe's
//     work equally.  Tiles are shaped for that: (8x,2y,2z) blocks at 8^3 when k >= 4, x-lines of all 8
//     points at 4^3, plain row runs otherwise (p.remap, kernels.h).
//   * the epilogue always goes through the fp32 LDS tile (bias + ReLU, then full-resolution store and/or the fused
//     2^3 max-pool).
#include <type_traits>

#include "kernels.h"
#include "mma.h"

namespace nesti {
namespace {

constexpr int kThreads = 512;
constexpr int kABytes = kTileM * kRowBytes;   // 64 KiB per input-chunk buffer
constexpr int kPoolStride = 272;              // bytes per row of the fp32 [512][64] epilogue tile (+16 B pad)
constexpr int kZeroOff = 2 * kABytes;         // all-zero 128-B row
constexpr int kTapTile = 128 * kRowBytes;     // bytes of one (chunk, tap) weight tile

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// 16 B per lane from saddr base + voff + KK KiB straight into registers, NOT tracked by hipcc's waitcnt insertion
// (see conv_taps_kernel); the tied operand keeps the fragment in one register across the tap loop.
template <int KK>
__device__ __forceinline__ void load_frag_untracked(u32x4& dst, unsigned voff, const unsigned char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(base), "n"(KK * 1024) : "memory");
}
// at most 3 younger loads outstanding => this fragment has landed
__device__ __forceinline__ void wait_frag(u32x4& frag) { asm volatile("s_waitcnt vmcnt(3)" : "+v"(frag) : : "memory"); }

template <int DT>
__global__ __launch_bounds__(kThreads) void conv_taps_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = wave & 3, mh = wave >> 2;   // column group (32 columns), row half (8 tiles)

  const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
  const int n_tile = grp % p.n_tiles;
  const int m_tile = (grp / p.n_tiles) * 8 + xcd;
  if (m_tile >= p.m_tiles) return;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int log2S = p.log2S, log2V = 3 * log2S;
  const int V = 1 << log2V;
  const int S = p.s_real ? p.s_real : (1 << log2S);   // bounds only (rows are laid out in the 2^log2S index space)
  const long long total_rows = (long long)npts << log2V;
  const long long r0 = (long long)m_tile * kTileM;
  if (r0 >= total_rows) return;
  const int remap = p.remap;

  // ---- tile geometry: row(t, l) = base(t) + off(l); voxel = (zt + zl, yt + yl, xl) ------------------------------
  const int l32 = lane & 31, khalf = lane >> 5;
  int off_l, zl, yl, xl, key_hi = 0, key_mask = 7;
  if (remap && log2S == 3) {            // (8x, 2y, 2z) blocks, t = 4 zp + yp
    zl = l32 >> 4; yl = (l32 >> 3) & 1; xl = l32 & 7;
    off_l = zl * 64 + yl * 8 + xl;
  } else if (remap) {                   // 4^3: the x-line (y, z) of all 8 points, t = 4 z + y
    zl = 0; yl = 0; xl = l32 & 3;
    off_l = (l32 >> 2) * 64 + xl;
    key_mask = 1; key_hi = ((l32 >> 2) & 3) << 1;     // swizzle key = ((row >> 1) & 1) | ((point & 3) << 1)
  } else {                              // rows [32t, 32t+32)
    const int lv = l32 & (V - 1), Sm = (1 << log2S) - 1;
    zl = lv >> (2 * log2S); yl = (lv >> log2S) & Sm; xl = lv & Sm;
    off_l = l32;
  }
  int base_t[8], zt[8], yt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = 8 * mh + i;
    if (remap && log2S == 3) { base_t[i] = 128 * (t >> 2) + 16 * (t & 3); zt[i] = 2 * (t >> 2); yt[i] = 2 * (t & 3); }
    else if (remap) { base_t[i] = 16 * (t >> 2) + 4 * (t & 3); zt[i] = t >> 2; yt[i] = t & 3; }
    else { const int tv = (32 * t) & (V - 1); base_t[i] = 32 * t; zt[i] = tv >> (2 * log2S); yt[i] = (tv >> log2S) & ((1 << log2S) - 1); }
  }

  // ---- input chunk staging (as conv_igemm_kernel: wave w, piece j covers LDS rows (8w+j)*8 .. +8) ----------------
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  auto stage_a = [&](int c, int buf) __attribute__((always_inline)) {   // once per chunk: addresses are recomputed, not kept
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row_l = (wave * 8 + j) * 8 + (lane >> 3);
      const int key = ((row_l >> 1) & key_mask) | (key_mask == 1 ? ((row_l >> 6) & 3) << 1 : 0);
      const int slot = (lane & 7) ^ key;                       // inverse swizzle on the SOURCE
      const long long gr = r0 + row_l;
      if (gr < total_rows) {
        long long pt = gr >> log2V;
        const long long vox = gr & (V - 1);
        if (p.point_index) pt = p.point_index[pt];
        const long long off = (((pt << log2V) + vox) * p.in_cstride + p.in_coff) * kEsz + slot * 16 + (long long)c * kRowBytes;
        glds16(in_b + off, lds0 + buf * kABytes + (wave * 8 + j) * 1024);
      }
    }
  };

  // This wave's weight fragments of the current tap; b[kk] is refilled for the next tap right after its last use.
  // The loads are issued through inline asm so that hipcc does not track them: behind the wave-uniform branches of
  // the tile-skipping code it would otherwise park an s_waitcnt vmcnt(0) in front of every MFMA.  Four refills are in
  // flight at any time, issued in K-step order, so "at most 3 outstanding" (wait_b) means b[kk] has landed; the
  // untracked LDS-DMA of the input chunk only ever makes that wait longer, never shorter.
  u32x4 b[4] = {}, bn[4] = {};   // current tap's fragments / the next tap's, in flight during the whole tap
  const unsigned b_voff = (unsigned)(ng * 4096 + lane * 16);
  const unsigned char* w_tile0 = reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)n_tile * p.n_chunks * p.n_taps * kTapTile;
  auto load_b = [&](int q, auto KK) __attribute__((always_inline)) {   // q = chunk * n_taps + tap
    constexpr int kk = decltype(KK)::value;
    load_frag_untracked<kk>(bn[kk], b_voff, w_tile0 + (size_t)q * kTapTile);
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // one tap of one chunk: all live tiles x 4 K-steps.  A fragments come from LDS (zero row for padding lanes).
  auto do_tap = [&](const int dz, const int dy, const int dx, const unsigned char* Abuf, const int q_next)
      __attribute__((always_inline)) {
    const int shift = (dz << (2 * log2S)) + (dy << log2S) + dx;
    const int r_l = off_l + shift;
    const bool okx = (unsigned)(xl + dx) < (unsigned)S;
    const int keyk = ((((r_l >> 1) & key_mask) | key_hi) ^ khalf) << 4;   // (key ^ khalf) * 16
    const int ra_l = r_l * kRowBytes;
    int addr[8];
    unsigned live = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = okx & ((unsigned)(zl + zt[i] + dz) < (unsigned)S) & ((unsigned)(yl + yt[i] + dy) < (unsigned)S);
      addr[i] = ok ? ra_l + base_t[i] * kRowBytes : kZeroOff;
      live |= (__ballot(ok) != 0ull) ? (1u << i) : 0u;
    }
    live = __builtin_amdgcn_readfirstlane(live);
    // the next tap's weight fragments stream in behind this tap's MFMAs
    if (q_next >= 0) {
      load_b(q_next, std::integral_constant<int, 0>{});
      load_b(q_next, std::integral_constant<int, 1>{});
      load_b(q_next, std::integral_constant<int, 2>{});
      load_b(q_next, std::integral_constant<int, 3>{});
    }
    int o[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o[kk] = (kk << 5) ^ keyk;   // ((2 kk) ^ key ^ khalf) * 16: swizzled slot per K-step
    uint4 bk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bk[kk] = __builtin_bit_cast(uint4, b[kk]);
    // tile-outer: a live tile is four MFMAs on ONE accumulator (back-to-back accumulation is the fast path of the
    // matrix pipe); the next live tile's four fragments are read from LDS before them.  A dead tile costs two scalar
    // branches and nothing else.
    uint4 a[2][4];
    auto load_a = [&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      if (live & (1u << i)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[i & 1][kk] = *reinterpret_cast<const uint4*>(Abuf + addr[i] + o[kk]);
      }
    };
    auto tile = [&](auto I) __attribute__((always_inline)) {
      constexpr int i = decltype(I)::value;
      if constexpr (i < 7) load_a(std::integral_constant<int, i + 1>{});
      if (live & (1u << i)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) mma<DT>(acc[i], a[i & 1][kk], bk[kk]);
      }
    };
    load_a(std::integral_constant<int, 0>{});
    tile(std::integral_constant<int, 0>{});
    tile(std::integral_constant<int, 1>{});
    tile(std::integral_constant<int, 2>{});
    tile(std::integral_constant<int, 3>{});
    tile(std::integral_constant<int, 4>{});
    tile(std::integral_constant<int, 5>{});
    tile(std::integral_constant<int, 6>{});
    tile(std::integral_constant<int, 7>{});
    // hand the prefetched fragments over (issued a whole tap ago)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]) : : "memory");
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b[kk] = bn[kk];
  };

  // ---- main loop: chunks outer, taps inner, two taps per iteration (weight registers ping-pong) -----------------
  if (tid < 32) reinterpret_cast<uint32_t*>(smem + kZeroOff)[tid] = 0u;
  const int n_taps = p.n_taps, Q = p.n_chunks * n_taps;
  stage_a(0, 0);
  load_b(0, std::integral_constant<int, 0>{});
  load_b(0, std::integral_constant<int, 1>{});
  load_b(0, std::integral_constant<int, 2>{});
  load_b(0, std::integral_constant<int, 3>{});
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]) : : "memory");
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) b[kk] = bn[kk];
  __syncthreads();
  if (p.n_chunks > 1) stage_a(1, 1);
  // taps are enumerated dz-major over the clipped range [lo, hi]^3 (model.hip: pack_layer), so the offsets are
  // counted instead of being fetched from the kernel arguments per tap
  const int d_lo = p.tap[0][0], d_hi = p.tap[n_taps - 1][0];
  int c = 0, t = 0, dz = d_lo, dy = d_lo, dx = d_lo;
#pragma unroll 1
  for (int q = 0; q < Q; ++q) {
    do_tap(dz, dy, dx, smem + (c & 1) * kABytes, q + 1 < Q ? q + 1 : -1);
    if (++dx > d_hi) { dx = d_lo; if (++dy > d_hi) { dy = d_lo; ++dz; } }
    if (++t == n_taps) {
      t = 0; ++c; dz = d_lo;
      wait_vm0();          // the next chunk's input (and the prefetched fragments) have landed
      __syncthreads();     // ... for every wave, and every wave is done reading the previous buffer
      if (c + 1 < p.n_chunks) stage_a(c + 1, (c + 1) & 1);
    }
  }
  __syncthreads();   // nobody still reads A when the epilogue tile overwrites it

  // ---- epilogue: bias + ReLU into the fp32 LDS tile (64 columns at a time), then the cooperative stores ----------
  const int out_esz = p.out_f32 ? 4 : kEsz;
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
  unsigned char* mp_b = reinterpret_cast<unsigned char*>(p.mp_out);
  const int out_col0 = p.out_coff + n_tile * 128;
  const float bv = p.bias[n_tile * 128 + ng * 32 + l32];
  const float act_floor = p.relu ? 0.f : -INFINITY;
  const int Vo = V >> 3, So = 1 << (log2S - 1), log2So = log2S - 1;
  auto cvt_store = [&](unsigned char* dst, const float4& v) __attribute__((always_inline)) {
    if (out_esz == 4) {
      *reinterpret_cast<float4*>(dst) = v;
    } else {
      using E = Elem<DT == NESTI_F32 ? NESTI_BF16 : DT>;
      *reinterpret_cast<uint2*>(dst) = make_uint2(E::pack2(v.x, v.y), E::pack2(v.z, v.w));
    }
  };
#pragma unroll 1
  for (int nh = 0; nh < 2; ++nh) {
    if ((ng >> 1) == nh) {
      const int col = (ng & 1) * 32 + l32;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int l = (r & 3) + 8 * (r >> 2) + 4 * khalf;
          int row;
          if (remap && log2S == 3) row = base_t[i] + (l >> 4) * 64 + ((l >> 3) & 1) * 8 + (l & 7);
          else if (remap) row = base_t[i] + (l >> 2) * 64 + (l & 3);
          else row = base_t[i] + l;
          *reinterpret_cast<float*>(smem + row * kPoolStride + col * 4) = fmaxf(acc[i][r] + bv, act_floor);
        }
    }
    __syncthreads();
    if (p.mp_mode != 1) {   // full-resolution rows, 16 consecutive lanes = one 64-channel row segment
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int item = it * kThreads + tid;
        const int row = item >> 4, cg = item & 15;
        const long long gr = r0 + row;
        if (gr < total_rows)
          cvt_store(out_b + (gr * p.out_cstride + out_col0 + nh * 64 + cg * 4) * out_esz,
                    *reinterpret_cast<const float4*>(smem + row * kPoolStride + cg * 16));
      }
    }
    if (p.mp_mode != 0) {   // fused 2^3 stride-2 max-pool of the activated values (utils/tf_util.py:424-428)
#pragma unroll
      for (int it = 0; it < 2; ++it) {   // 64 pooled rows x 16 channel groups
        const int item = it * kThreads + tid;
        const int orow = item >> 4, cg = item & 15;
        const int pt_l = orow >> (log2V - 3), cell = orow & (Vo - 1);
        const int cz = cell >> (2 * log2So), cy = (cell >> log2So) & (So - 1), cx = cell & (So - 1);
        const int base = (pt_l << log2V) + ((((2 * cz) << log2S) + 2 * cy) << log2S) + 2 * cx;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int row = base + ((a >> 2) << (2 * log2S)) + (((a >> 1) & 1) << log2S) + (a & 1);
          const float4 v = *reinterpret_cast<const float4*>(smem + row * kPoolStride + cg * 16);
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        const long long go = (r0 >> 3) + orow;
        if (go < (total_rows >> 3))
          cvt_store(mp_b + (go * p.mp_cstride + out_col0 + nh * 64 + cg * 4) * out_esz, m);
      }
    }
    __syncthreads();
  }
}

constexpr size_t kLdsBytes = (size_t)kTileM * kPoolStride + 16 > (size_t)2 * kABytes + kRowBytes
                                 ? (size_t)kTileM * kPoolStride + 16 : (size_t)2 * kABytes + kRowBytes;

template <int DT>
int launch_taps_dt(const ConvParams& p, hipStream_t stream) {
  static bool attr_set = false;
  static_assert(kLdsBytes <= 163840, "LDS budget");
  if (!attr_set) {
    NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_taps_kernel<DT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
    attr_set = true;
  }
  const int groups = (p.m_tiles + 7) / 8;
  dim3 grid((unsigned)(groups * 8 * p.n_tiles)), block(kThreads);
  hipLaunchKernelGGL((conv_taps_kernel<DT>), grid, block, kLdsBytes, stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_conv_taps(const ConvParams& p, int dtype, hipStream_t stream) {
  if (p.m_tiles <= 0 || p.n_tiles <= 0) return 0;
  if (p.n_taps < 2 || p.pool_k > 1 || p.split_tile != p.n_tiles) NESTI_FAIL("launch_conv_taps: plain multi-tap layers only");
  if (p.log2S < 1) NESTI_FAIL("launch_conv_taps: needs a volume");
  if (p.remap && p.log2S != 2 && p.log2S != 3) NESTI_FAIL("launch_conv_taps: remapped tiles exist at 4^3 and 8^3 only");
  if (p.mp_mode != 0 && !p.mp_out) NESTI_FAIL("launch_conv_taps: fused max-pool needs an output");
  if (dtype == NESTI_BF16) return launch_taps_dt<NESTI_BF16>(p, stream);
  if (dtype == NESTI_F16) return launch_taps_dt<NESTI_F16>(p, stream);
  if (dtype == NESTI_F32) return launch_taps_dt<NESTI_F32>(p, stream);
  NESTI_FAIL("launch_conv_taps: unsupported dtype");
}

}  // namespace nesti
