// k^3-tap conv3d on the 8^3 volume (k = 3, 5): the dominant layers of the network (the 5^3 taps at 8^3 are ~2/3 of all
// multiply-accumulates of a top-1 forward pass).
//
// Same arithmetic as conv_igemm_kernel (conv.hip): tf.nn.conv3d 'SAME' + bias_add + inference batch-norm (folded on the host)
// + ReLU (utils/tf_util.py:298-311, 491-494), optionally followed by the block's 2^3 / 2 max-pool (utils/tf_util.py:424-428,
// e.g. models/experts_n_est.py:198) fused into the epilogue.
//
// Why another kernel.  conv_igemm_kernel's time follows the number of MFMAs it issues and the weight bytes it streams
// (profiles/r01_mfma_ubench.txt).  With one point per workgroup a 32-row MFMA tile spans several y or z values, so a padding
// tap can only skip it when ALL of them leave the volume (issued / nominal 0.81 for 5^3, against 0.61 useful), and every
// workgroup re-streams every tap's weight tile for its one point.  Here
//   * an MFMA tile is the x-line (y, z) of FOUR points (32 rows = 4 points x 8 x): a tap (dz, dy, dx) skips the tile exactly
//     when y + dy or z + dz leaves the volume, at single-voxel granularity on both axes (issued / nominal 0.72 for 5^3, 0.84
//     for 3^3), and the skip is a scalar test -- all 32 rows of a tile share (y, z);
//   * the K chunk is 64 bytes per row (32 x 16-bit or 16 x f32 channels), so the 4 points' chunk stays resident in LDS for
//     all k^3 taps and a tap's weight tile serves 4 points instead of one: a quarter of the weight stream per output;
//   * the only per-lane padding test left is x + dx: an out-of-range lane reads an LDS address beyond the allocation, which
//     returns zeros on gfx950 (scripts/lds_oob_probe.hip; checked at model creation, conv8_selftest), so no zero rows are kept;
//   * a wave owns FOUR x-line tiles x TWO 32-column tiles, so that every A fragment read from LDS feeds two MFMAs (plain
//     modes) or three + three (pair modes): 0.75 / 0.5 KB of LDS reads per MFMA (round 3: 1.125 / 0.75 with eight tiles x one
//     column tile per wave, 2-5 % slower same-box; these kernels sit within ~10 % of the matrix pipe's own rate on such data,
//     DESIGN.md 4.3).
//
// Workgroup = 4 points x ONE z half of the output volume (z in [4h, 4h + 4)) x 64 output channels.  It stages the
// NZ = 4 + k/2 source planes that half can reach (6 of 8 for k = 5: 96 KiB instead of 128) and 4 KiB of weights per tap.
// Tile (y, z) of the half lives with wave (y + z) & 7 -- every wave owns one tile of every z plane and one of every y plane
// (a Latin square), so whatever planes a tap kills, every wave, hence every SIMD's matrix pipe, loses the same number of
// tiles -- and the four tiles a wave reads for tap (dz, dy) are consecutive slots of wave (w + dy + dz) & 7's run; source
// plane z' sits at slot NZ * ((y' + z') & 7) + z' - zlo.  A fragment address is one per-lane base (x + dx shift, swizzle)
// plus a wave-uniform base plus a compile-time j * 2048.  Waves w and w + 4 (one SIMD) together hold all eight y lines of
// every z plane of the half.
//
// Pipeline.  The weight tiles of one (dz, dy) row of taps stream L2 -> LDS by LDS-DMA two rows ahead into 3 slots, one barrier
// per row (k = 5); pairs of rows one pair ahead into 2 slots, one barrier per pair (k = 3).  Within a wave the A fragments of
// tap t + 1 are read into the registers tile j's MFMAs of tap t have just consumed (a full tap of lookahead).
//
// LDS: [0, 64 KiB) weight slots (3 x 20 KiB rows for k = 5, 2 x 24 KiB row pairs for k = 3), [64 KiB, 64 + 16 NZ KiB) the
// input chunk; rows are 64 B with the 16-B slot XOR-swizzled by the point index (input) / (row >> 2) & 3 (weights), applied on
// the DMA source address, which makes every ds_read_b128 lane group conflict-free.  The epilogue reuses the LDS as an fp32
// staging tile.
//
// X3 (pair modes, model.hip: PackedLayer::x3n): the K chunk is 16 channels -- an LDS row holds [hi k0..15 | lo k0..15] of the
// activation pair (staged from the two planes of the [hi 64 | lo 64] row groups) and a weight row [W_hi k0..15 | W_lo k0..15]
// -- and a tap multiplies hi * W_hi + lo * W_hi + hi * W_lo from ONE set of fragment reads.
//
// X8 (round 6, NESTI_F16X8 / NESTI_F16X8C, the experts' tap layers picked by nesti_model::x8_mask): the two CROSS terms of the pair
// scheme carry 2^-11 of the result, so they go through ONE block-scaled FP8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3, same fp32
// accumulators) instead of two f16 MFMAs each: an LDS row is [hi k0..15 f16 | lo8 k0..15 | hi8 k0..15] (hi from the pair buffer's hi
// plane, the two e4m3 planes from the side buffer the producing 1x1x1 layer writes: conv.hip, ConvParams::aux8_out), a weight row
// [W_hi f16 | W_hi8 | W_lo8], and the K = 64 of one FP8 instruction is TWO taps x [lo8 | hi8]: lanes 0-31 (K block 0) read tap u's
// rows, lanes 32-63 (K block 1) tap u + 1's -- a per-lane address like the x shift itself.  Per tap and (tile, column tile): one f16
// MFMA (hi * W_hi, exact products) and, every second tap (and at the odd tap that ends a row, its second K block reading zeros), one
// FP8 MFMA: lo8 * W_hi8 + hi8 * W_lo8 with the E8M0 block scales 2^-sa, 2^-sb undoing the power-of-two pre-scales.  Measured before it
// was built (profiles/r06_fp8_cross_step0.txt, r06_x8_ubench.txt): the residual is 28x below single-product f16 (max 1 - cos 1.1e-6
// over the bench cloud with both 5^3 layers of every expert in this form) at 1.47x the pair loop's multiply rate.
//
// What did NOT pay (profiles/r02_conv8_experiments.txt): seven re-schedulings of this loop -- waves out of
// phase, counted vmcnt, hand-counted lgkmcnt with inline-asm reads, adjacent MFMA pairs, K-step-major order, weights straight
// from L2 into registers without a row barrier, barrier-free rows through LDS counters -- all within +-3 % or slower.  The
// kernel runs against the chip's power / clock limit: its time follows the MFMAs issued and the operand bits they toggle
// (zero data: +9 %), which is what the tile layout above reduces.
#include <string.h>

#include <mutex>

#include <type_traits>

#include "kernels.h"
#include "mma.h"

namespace nesti {
namespace {

constexpr int kThreadsN = 512;
constexpr int kPtsN = 4;
constexpr int kTileN = 2048;                 // 32 rows x 64 B
constexpr int kBTileN = 2 * kTileN;          // one tap's weights: two 32-column tiles
constexpr int kAOffN = 65536;
constexpr unsigned kOobN = 0x40000u;         // beyond any LDS allocation: ds_read returns 0
constexpr int kEpiStrideN = 144;             // bytes per row of the fp32 [1024][32] epilogue tile (+16 B pad)

typedef unsigned u32x4n_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4n_t* lds_u32x4n_ptr;
__device__ __forceinline__ uint4 lds128n(unsigned addr) {
  const u32x4n_t v = *(lds_u32x4n_ptr)(size_t)addr;
  return make_uint4(v.x, v.y, v.z, v.w);
}

// X6: the second 16 bytes of an FP6 block hold two data dwords, the scale byte's dword and padding.  Read as 8 + 4 bytes the six data
// dwords of an operand are a 16-byte and an 8-byte result, which hipcc allocates as ONE six-register tuple; read as 16 + 16 it assembles
// the tuple with two v_mov per operand in front of every instruction (a forced eight-register tuple spills instead: measured, -3.5 %)
typedef unsigned u32x2n_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2n_t* lds_u32x2n_ptr;
typedef const __attribute__((address_space(3))) unsigned* lds_u32n_ptr;
template <bool X6>
__device__ __forceinline__ uint4 lds_hi16n(unsigned addr) {
  if constexpr (X6) {
    const u32x2n_t d = *(lds_u32x2n_ptr)(size_t)addr;
    const unsigned sc = *(lds_u32n_ptr)(size_t)(addr + 8u);
    return make_uint4(d.x, d.y, sc, 0u);
  } else {
    return lds128n(addr);
  }
}

template <int K> constexpr int lds_bytes_n() { return kAOffN + 8 * (4 + (K - 1) / 2) * kTileN; }

typedef int i32x8n_t __attribute__((ext_vector_type(8)));

template <int DT, int K, int MODE>     // MODE 0: plain, 1: the pair K loop (X3), 2: f16 hi * W_hi + FP8 cross terms (X8), 3: the same with FP6 blocks (X6)
__device__ __forceinline__ void conv8n_tile(const ConvParams& p, const unsigned bid, const int tid_in) {
  constexpr bool X3 = MODE == 1, X8 = MODE >= 2, X6 = MODE == 3;
  static_assert(!X8 || DT == NESTI_F16, "the FP8 cross-term loop is an f16 pair-mode variant");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;
  constexpr int LO = (K - 1) / 2;
  constexpr int NG = K * K;                  // (dz, dy) rows of taps per chunk
  constexpr int NZ = 4 + LO;                 // staged source planes
  constexpr int R = (K == 3) ? 2 : 1;        // rows of taps per weight slot
  constexpr int NS = (K == 3) ? 2 : 3;
  constexpr int AHEAD = NS - 1;
  constexpr int kSlot = R * K * kBTileN;
  constexpr int NSR = (NG + R - 1) / R;
  static_assert(NS * kSlot <= kAOffN, "weight slots must fit below the input chunk");
  static_assert((K & 1) == 1, "the weight-fragment ping-pong assumes an odd number of taps per row");

  const int tid = tid_in, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware block -> tile map: the 2 halves x n_tiles column pairs of a group of 4 points stay on one XCD's L2
  const int xcd = bid & 7, grp = bid >> 3;
  const int per_m = 2 * p.n_tiles;                       // p.n_tiles = 64-column pairs
  const int sub = grp % per_m;
  const int n_pair = sub >> 1, half = sub & 1;
  const int m_tile = (grp / per_m) * 8 + xcd;
  if (m_tile >= p.m_tiles) return;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int p0 = m_tile * kPtsN;
  if (p0 >= npts) return;
  const int np_here = min(kPtsN, npts - p0);
  const int z0 = 4 * half;                               // first output plane of this half
  const int zlo = half ? 8 - NZ : 0;                     // first staged source plane

  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in) +
                              ((size_t)p0 * 512 * p.in_cstride + p.in_coff) * kEsz;
  const unsigned char* w_tile = reinterpret_cast<const unsigned char*>(p.wpk) +
                                (size_t)n_pair * p.n_chunks * (K * K * K) * kBTileN;

  // ---- A staging: wave w fills the NZ slots of its run; piece h of a tile = rows 16h .. 16h+15 (row = 8 pt + x) ------
  const int st_row = lane >> 2;
  unsigned a_voff[2];
  bool a_ok[2];
  bool a_is8[2];                                        // X8: this lane's 16-B slot comes from the side buffer (slot 2: its lo8
  const unsigned char* aux_b = nullptr;                  // plane, slot 3: its hi8 plane) instead of the pair buffer's hi plane (slots 0, 1)
  if constexpr (X8) aux_b = reinterpret_cast<const unsigned char*>(p.aux8_in) + (size_t)p0 * 512 * p.aux8_stride;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pt = 2 * h + (st_row >> 3), x = st_row & 7;
    const int kslot = (lane & 3) ^ pt;                   // inverse swizzle on the SOURCE (LDS-DMA writes lane-linear)
    a_voff[h] = (unsigned)((pt * 512 + x) * p.in_cstride * kEsz + (X3 ? (kslot & 1) * 16 + (kslot >> 1) * (2 * kSplitGroup) : kslot * 16));
    a_is8[h] = X8 && kslot >= 2;
    if (a_is8[h]) a_voff[h] = (unsigned)((pt * 512 + x) * p.aux8_stride + (kslot - 2) * kSplitGroup);
    a_ok[h] = pt < np_here;
  }
  auto stage_a = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int jz = 0; jz < NZ; ++jz) {
      const int zp = zlo + jz, yp = (wave - zp) & 7;     // source tile (y', z') of this wave's run: (y' + z') & 7 == wave
      if constexpr (X8) {
        // chunk c = 16 channels: the hi plane of the pair row group [hi 64 | lo 64] (32 B at (c & 3) * 32) and 16 B of each e4m3
        // plane of the side row group [lo8 64 | hi8 64]
        const size_t off16 = (size_t)((zp * 64 + yp * 8) * p.in_cstride) * kEsz + (size_t)(c >> 2) * (2 * kPairPlanes * kSplitGroup) + (size_t)(c & 3) * 32;
        const size_t off8 = (size_t)(zp * 64 + yp * 8) * p.aux8_stride + (size_t)(c >> 2) * (2 * kSplitGroup) + (size_t)(c & 3) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (a_ok[h]) glds16((a_is8[h] ? aux_b + off8 : in_b + off16) + a_voff[h], lds0 + kAOffN + (wave * NZ + jz) * kTileN + h * 1024);
      } else {
      const unsigned char* src = in_b + (size_t)((zp * 64 + yp * 8) * p.in_cstride) * kEsz +
                                 (X3 ? (size_t)(c >> 2) * (2 * kPairPlanes * kSplitGroup) + (size_t)(c & 3) * 32
                                     : p.in_pair ? (size_t)(c >> 1) * (2 * kPairPlanes * kSplitGroup) + (size_t)(c & 1) * 64 : (size_t)c * 64);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (a_ok[h]) glds16(src + a_voff[h], lds0 + kAOffN + (wave * NZ + jz) * kTileN + h * 1024);
      }
    }
  };
  auto stage_b = [&](int c, int sr, int slot) __attribute__((always_inline)) {   // rows sr * R .. of chunk c
    const unsigned char* src = w_tile + ((size_t)c * (K * K * K) + (size_t)sr * R * K) * kBTileN;
    const int pieces = 4 * K * min(R, NG - sr * R);
    for (int pid = wave; pid < pieces; pid += 8)
      glds16(src + pid * 1024 + lane * 16, lds0 + slot * kSlot + pid * 1024);
  };

  // ---- per-lane fragment coordinates ---------------------------------------------------------------------------
  const int l31 = lane & 31, khalf = lane >> 5;
  const int pt = l31 >> 3, rx = l31 & 7;
  const int a_sw = (khalf ^ pt) & 3, b_sw = (khalf ^ (l31 >> 2)) & 3;
  const unsigned a_lane = lds0 + kAOffN + (unsigned)((pt * 8 + rx) * 64 + (a_sw << 4));
  const unsigned b_lane = lds0 + (unsigned)(l31 * 64 + (b_sw << 4));
  const unsigned a_d1 = (a_sw & 2) ? (unsigned)-32 : 32u, b_d1 = (b_sw & 2) ? (unsigned)-32 : 32u;

  // liveness of this wave's tiles: tile j is (y = (wave - z0 - j) & 7, z = z0 + j); 4 bits per dy / per dz
  unsigned ymask_pack = 0u, zmask_pack = 0u;
#pragma unroll
  for (int d = 0; d < K; ++d) {
    unsigned my = 0, mz = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = (wave - z0 - j) & 7, z = z0 + j;
      if ((unsigned)(y + d - LO) < 8u) my |= 1u << j;
      if ((unsigned)(z + d - LO) < 8u) mz |= 1u << j;
    }
    ymask_pack |= my << (4 * d);
    zmask_pack |= mz << (4 * d);
  }
  // Per-tap bookkeeping is kept off the critical path (a SIMD's two waves run the same phase between the row barriers, so
  // every scalar / vector instruction a wave spends between two MFMA groups is matrix-pipe idle time): the (dz, dy) loops
  // carry their masks, source-run offsets and weight-slot offsets incrementally (no divisions), the x-shifted per-lane read
  // addresses are K precomputed VGPRs.
  unsigned pa[K];                                        // per-lane A read address for x shift d - LO, or out of range
#pragma unroll
  for (int d = 0; d < K; ++d) pa[d] = ((unsigned)(rx + d - LO) < 8u) ? a_lane + (unsigned)((d - LO) * 64) : kOobN;
  // X8: one FP8 instruction takes a PAIR of taps -- K block khalf (lanes 0-31 / 32-63) is the pair's tap khalf, and a lane's 32
  // operand bytes are slots 2, 3 ([lo8 | hi8] / [W_hi8 | W_lo8]) of ITS tap's row.  The taps of a chunk are paired as one flat
  // sequence (K is odd: rows alternate between (0 1)(2 3)(4 | next row's 0) and (1 2)(3 4)), so a chunk issues K^3 / 2 FP8
  // instructions per tile, not K^2 (K + 1) / 2; the pair that straddles two rows takes each half's liveness from its own row.
  unsigned dslot_a = 0u, a8_d1 = 0u, b8k = 0u, b8_d1 = 0u;
  if constexpr (X8) {
    const int bkey = (l31 >> 2) & 3;
    dslot_a = (unsigned)((((2 ^ pt) & 3) - a_sw) * 16);          // from a tap's f16 fragment address (slot khalf) to its slot 2
    a8_d1 = (unsigned)((((3 ^ pt) & 3) - ((2 ^ pt) & 3)) * 16);
    b8k = lds0 + (unsigned)(l31 * 64 + (((2 ^ bkey) & 3) << 4) + khalf * kBTileN);   // slot 2 of this column's row in tap khalf of a weight row
    b8_d1 = (unsigned)((((3 ^ bkey) & 3) - ((2 ^ bkey) & 3)) * 16);
  }
  auto mask_of = [&](int dzi, int dyi) __attribute__((always_inline)) -> unsigned {
    return (zmask_pack >> (4 * dzi)) & (ymask_pack >> (4 * dyi)) & 0xfu;
  };
  // LDS byte offset of the source run of tap row (dz, dy): slots of wave (w + dy + dz) & 7, starting at plane z0 + dz
  auto base_of = [&](int dzi, int dyi) __attribute__((always_inline)) -> int {
    return (((wave + dyi + dzi - 2 * LO) & 7) * NZ + (z0 - zlo + dzi - LO)) * kTileN;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

  // b[u & 1] holds tap u's weight fragments [column tile][K-step, or hi / lo]; K is odd, so the last tap of a row leaves the
  // next row's first set in b[1]: it is moved to b[0] once per row
  uint4 a[4][2], b[2][2][2];
  auto load_b = [&](uint4 (&dst)[2][2], unsigned src) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      dst[n][0] = lds128n(src + n * kTileN);
      dst[n][1] = lds128n(src + n * kTileN + b_d1);
    }
  };
  auto tile_mma = [&](int j, const uint4 (&bc)[2][2]) __attribute__((always_inline)) {
    if (X3) {   // hi * W_hi, lo * W_hi, hi * W_lo for both column tiles (a[j][0] = hi, a[j][1] = lo; bc[n][0] = W_hi, bc[n][1] = W_lo)
      mma<DT>(acc[j][0], a[j][0], bc[0][0]);
      mma<DT>(acc[j][1], a[j][0], bc[1][0]);
      mma<DT>(acc[j][0], a[j][1], bc[0][0]);
      mma<DT>(acc[j][1], a[j][1], bc[1][0]);
      mma<DT>(acc[j][0], a[j][0], bc[0][1]);
      mma<DT>(acc[j][1], a[j][0], bc[1][1]);
    } else {
      mma<DT>(acc[j][0], a[j][0], bc[0][0]);
      mma<DT>(acc[j][1], a[j][0], bc[1][0]);
      mma<DT>(acc[j][0], a[j][1], bc[0][1]);
      mma<DT>(acc[j][1], a[j][1], bc[1][1]);
#ifdef CONV8N_DOUBLE_MMA   // timing-only experiment (wrong results): twice the MFMAs per (chunk, tap) step on the same fragment reads
      mma<DT>(acc[j][0], a[j][0], bc[0][1]);
      mma<DT>(acc[j][1], a[j][0], bc[1][1]);
      mma<DT>(acc[j][0], a[j][1], bc[0][0]);
      mma<DT>(acc[j][1], a[j][1], bc[1][0]);
#endif
    }
  };
  // one (dz, dy) row of K taps
  auto row = [&](unsigned mask_g, unsigned mask_n, int base_g, int base_n, unsigned boff_g, unsigned boff_n,
                 bool pair_end) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const bool last_u = (u == K - 1);
      unsigned m_mm = mask_g, m_rd = last_u ? mask_n : mask_g;   // tiles multiplied now / tiles whose next-tap fragments are read
      asm volatile("" : "+s"(m_mm), "+s"(m_rd));
      const unsigned nb0 = pa[last_u ? 0 : u + 1] + (unsigned)(last_u ? base_n : base_g);
      const unsigned nb1 = nb0 + a_d1;
      const unsigned bsrc = b_lane + (last_u ? boff_n : boff_g + (unsigned)((u + 1) * kBTileN));
      // Fragment reads are UNCONDITIONAL (a dead tile reads its slot anyway; only its MFMAs are skipped), so the number of
      // reads in flight is static and the compiler places an exact s_waitcnt lgkmcnt(N) in front of each tile's MFMAs
      // instead of one full drain per tap.
      if (R > 1 && last_u && pair_end) {
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wave's reads of the slot are done
        // shared slots (k = 3): confirm the next pair's weights before this tap prefetches their first fragments, and
        // free the current slot for the fill the next row starts
        wait_vm0();
        __builtin_amdgcn_s_barrier();
      }
      if (u == 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n) { b[0][n][0] = b[1][n][0]; b[0][n][1] = b[1][n][1]; }
      }
      uint4(&bc)[2][2] = b[u & 1];
      uint4(&bn)[2][2] = b[(u + 1) & 1];
      load_b(bn, bsrc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (__builtin_expect((m_mm & (1u << j)) != 0, 1)) tile_mma(j, bc);
        a[j][0] = lds128n(nb0 + j * kTileN);
        a[j][1] = lds128n(nb1 + j * kTileN);
      }
    }
    if (R == 1) {
      wait_vm0();                                        // the next row's weight DMA (issued two rows ago) has landed
      __builtin_amdgcn_s_barrier();
    }
  };


  // ---- X8: f16 hi * W_hi per tap, FP8 cross terms per tap pair ---------------------------------------------------------------
  uint4 a8[2][2], b8[2][2];                              // a8: TWO rolling tile buffers (tiles j and j + 2 of a cross step share one) x
                                                         // [slot 2 (lo8), slot 3 (hi8)]; b8: [column tile][slot 2 (W_hi8), slot 3 (W_lo8)]
  int sc_a = p.x8_scale_a, sc_b = p.x8_scale_b;          // X8: E8M0 block scales (2^-sa, 2^-sb), one per operand and layer
  if constexpr (!X6) asm volatile("" : "+v"(sc_a), "+v"(sc_b));   // the scale operands are VGPRs (hipcc fails to copy them out of SGPRs itself)
  auto mma8 = [&](f32x16& c, const uint4 (&av)[2], const uint4 (&bv)[2]) __attribute__((always_inline)) {
    const i32x8n_t a_ = {(int)av[0].x, (int)av[0].y, (int)av[0].z, (int)av[0].w, (int)av[1].x, (int)av[1].y, (int)av[1].z, (int)av[1].w};
    const i32x8n_t b_ = {(int)bv[0].x, (int)bv[0].y, (int)bv[0].z, (int)bv[0].w, (int)bv[1].x, (int)bv[1].y, (int)bv[1].z, (int)bv[1].w};
    if constexpr (X6) {    // e2m3: 32 six-bit elements in dwords 0-5, the block's own E8M0 scale in byte 0 of dword 6 (the instruction ignores
                           // dwords 6-7 as data: scripts/fp6_probe.hip); an out-of-range or dead half reads zeros, scale 2^-127 included
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_, b_, c, 2, 2, 0, (int)av[1].z, 0, (int)bv[1].z);
    } else {
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_, b_, c, 0, 0, 0, sc_a, 0, sc_b);
    }
  };
  // A cross step (the tap that ends a pair) walks the four tiles with two FP8 fragment buffers: tiles 0 / 1 arrive prefetched, tile
  // j + 2's fragments of the SAME pair are read into tile j's buffer right behind tile j's MFMAs (tile j + 1's four MFMAs cover the
  // LDS latency), and behind tiles 2 / 3 the buffers take tiles 0 / 1 of the NEXT pair.
  // Row type T = row parity within the chunk.  T = 0: pairs (0 1) (2 3) ... end at the odd taps, and the last tap pairs with tap 0
  // of the NEXT row (K block 1 reads that row's source run and weights; a tile dead in one of the two rows reads zeros for that
  // half, and the instruction is issued when it is live in either).  T = 1: tap 0 is already done, pairs (1 2) (3 4) ... end at the
  // even taps.  The last row of a chunk is of type 0 (K^2 is odd) and has no next row: mask_n = 0 makes its second half zeros.
  auto a8_same = [&](int d0, unsigned base) __attribute__((always_inline)) -> unsigned {     // pair (d0 | d0 + 1) of one row
    return (khalf ? pa[d0 + 1] : pa[d0]) + dslot_a + base;
  };
  auto row8 = [&](const int T, unsigned mask_g, unsigned mask_n, int base_g, int base_n, unsigned boff_g, unsigned boff_n, bool more,
                  bool pair_end) __attribute__((always_inline)) {       // T is a literal at both call sites: it folds after inlining
    // the pair that straddles this row and the next (T = 0 only): per-lane source, weight and liveness selections
    unsigned selA = 0u, selB = 0u, lv = 0u;
    if (T == 0) {
      selA = (khalf ? pa[0] + (unsigned)base_n : pa[K - 1] + (unsigned)base_g) + dslot_a;
      selB = khalf ? (more ? b8k - (unsigned)kBTileN + boff_n : kOobN) : b8k + (unsigned)((K - 1) * kBTileN) + boff_g;
      lv = khalf ? mask_n : mask_g;
    }
    auto a8_x = [&](int j) __attribute__((always_inline)) -> unsigned { return ((lv >> j) & 1u) ? selA : kOobN; };
    const unsigned m_mm = (unsigned)__builtin_amdgcn_readfirstlane((int)mask_g);   // scalars: the tile skips are s_cbranch, not exec masks
    const unsigned m_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(mask_g | mask_n));
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const bool last_u = (u == K - 1);
      const bool cross = T == 0 ? ((u & 1) || last_u) : (u >= 2 && !(u & 1));   // this tap ends a pair: its FP8 MFMAs are issued here
      const bool xrow = T == 0 && last_u;                                        // ... the pair that straddles the rows
      const bool next_x = T == 0 && u == K - 2;                                  // the pair AFTER this one straddles the rows
      const unsigned nb0 = pa[last_u ? 0 : u + 1] + (unsigned)(last_u ? base_n : base_g);
      const unsigned bsrc = b_lane + (last_u ? boff_n : boff_g + (unsigned)((u + 1) * kBTileN));
      if (R > 1 && last_u && pair_end) {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        wait_vm0();
        __builtin_amdgcn_s_barrier();
      }
      // this pair's first tap d0 (same-row pairs) and the next pair's: (u + 1 | u + 2) in this row, the straddling pair, or the
      // next row's first pair -- (1 | 2) after a type-0 row, (0 | 1) after a type-1 row
      const int d0 = u - 1;
      unsigned cb8 = 0u, nb8 = 0u, bsrc8 = 0u;
      if (cross) {
        if (!xrow) cb8 = a8_same(d0, (unsigned)base_g);
        if (!next_x) {
          const int nd0 = last_u ? (T == 0 ? 1 : 0) : u + 1;
          nb8 = a8_same(nd0, (unsigned)(last_u ? base_n : base_g));
          bsrc8 = b8k + (unsigned)(nd0 * kBTileN) + (last_u ? boff_n : boff_g);
        } else {
          bsrc8 = selB;
        }
      }
      // f16 hi * W_hi, column tile by column tile: a W_hi fragment is free for the next tap's after four MFMAs, an A fragment after
      // its second one -- every operand is single-buffered
      // three copies of the tile mask, each opaque to the compiler: otherwise it merges the repeated bit tests into vector compares
      // (v_cndmask + v_cmp + s_andn2 per branch instead of s_bitcmp + s_cbranch)
      unsigned m0 = m_mm, m1 = m_mm, m2 = xrow ? m_x : m_mm;
      asm volatile("" : "+s"(m0), "+s"(m1), "+s"(m2));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (__builtin_expect((m0 & (1u << j)) != 0, 1)) mma<DT>(acc[j][0], a[j][0], b[0][0][0]);
      b[0][0][0] = lds128n(bsrc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (__builtin_expect((m1 & (1u << j)) != 0, 1)) mma<DT>(acc[j][1], a[j][0], b[0][1][0]);
        a[j][0] = lds128n(nb0 + j * kTileN);
      }
      b[0][1][0] = lds128n(bsrc + kTileN);
      if (cross) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (__builtin_expect((m2 & (1u << j)) != 0, 1)) {
            mma8(acc[j][0], a8[j & 1], b8[0]);
            mma8(acc[j][1], a8[j & 1], b8[1]);
          }
          // tiles 2 / 3 of THIS pair behind tiles 0 / 1, tiles 0 / 1 of the NEXT pair behind tiles 2 / 3
          const unsigned src8 = j < 2 ? (xrow ? a8_x(j + 2) : cb8) + (j + 2) * kTileN : (next_x ? a8_x(j - 2) : nb8) + (j - 2) * kTileN;
          a8[j & 1][0] = lds128n(src8);
          a8[j & 1][1] = lds_hi16n<X6>(src8 + a8_d1);
        }
      }
      if (cross) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          b8[n][0] = lds128n(bsrc8 + n * kTileN);
          b8[n][1] = lds_hi16n<X6>(bsrc8 + b8_d1 + n * kTileN);
        }
      }
    }
    if (R == 1) {
      wait_vm0();
      __builtin_amdgcn_s_barrier();
    }
  };

  for (int c = 0; c < p.n_chunks; ++c) {
    __syncthreads();                       // every wave is done with the previous chunk
    stage_a(c);
#pragma unroll
    for (int sr = 0; sr < AHEAD; ++sr) stage_b(c, sr, sr);
    wait_vm0();
    __syncthreads();
    {                                      // prologue: fragments of tap 0 (row 0, dx = -LO); every row starts by moving b[1] to b[0]
      const unsigned m0 = mask_of(0, 0);
      const unsigned nb0 = pa[0] + (unsigned)base_of(0, 0);
      const unsigned nb1 = nb0 + a_d1;
      if constexpr (X8) {
        const unsigned nb8 = a8_same(0, (unsigned)base_of(0, 0));      // row 0 is of type 0: its first pair is (0 | 1)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          b[0][n][0] = lds128n(b_lane + n * kTileN);
          b8[n][0] = lds128n(b8k + n * kTileN);
          b8[n][1] = lds_hi16n<X6>(b8k + b8_d1 + n * kTileN);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j][0] = lds128n(nb0 + j * kTileN);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          a8[j][0] = lds128n(nb8 + j * kTileN);
          a8[j][1] = lds_hi16n<X6>(nb8 + a8_d1 + j * kTileN);
        }
      } else {
      load_b(b[1], b_lane);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (m0 & (1u << j)) {
          a[j][0] = lds128n(nb0 + j * kTileN);
          a[j][1] = lds128n(nb1 + j * kTileN);
        }
      }
    }
    // row g = (dzi, dyi); its weights sit at LDS offset boff; the fill that keeps AHEAD slot fills in flight goes to `fslot`
    int dzi = 0, dyi = 0, fill = AHEAD, fslot = AHEAD % NS, g = 0;
    unsigned boff = 0u;
    auto do_row = [&](const int T) __attribute__((always_inline)) {        // T: the row's parity (X8: its pairing type), a literal at every call
      if (g % R == 0 && fill < NSR) {      // g % R: R is 1 or 2
        stage_b(c, fill, fslot);
        ++fill;
        fslot = (fslot + 1 == NS) ? 0 : fslot + 1;
      }
      const bool more = g + 1 < NG;
      int dzn = dzi, dyn = dyi + 1;
      if (dyn == K) { dyn = 0; ++dzn; }
      // next row's weights: the second half of this slot (k = 3, first row of a pair) or the start of the next slot
      unsigned boff_n = (R > 1 && (g & 1) == 0) ? boff + (unsigned)(K * kBTileN) : (boff - (boff % kSlot)) + kSlot;
      if (boff_n >= (unsigned)(NS * kSlot)) boff_n = 0u;
      const unsigned mask_g = mask_of(dzi, dyi), mask_n = more ? mask_of(dzn, dyn) : 0u;
      const int base_g = base_of(dzi, dyi), base_n = more ? base_of(dzn, dyn) : 0;
      const bool pair_end = (g % R == R - 1) || !more;
      if constexpr (X8) row8(T, mask_g, mask_n, base_g, base_n, boff, boff_n, more, pair_end);
      else row(mask_g, mask_n, base_g, base_n, boff, boff_n, pair_end);
      dzi = dzn; dyi = dyn; boff = boff_n; ++g;
    };
    if constexpr (X8) {
      // rows in pairs (type 0, type 1) as straight-line code, the odd last row behind the loop: a run-time branch between the two
      // row bodies inside the loop makes hipcc spill hundreds of registers (the accumulators pass through its merge point)
      for (int gp = 0; gp < NG / 2; ++gp) {
        do_row(0);
        do_row(1);
      }
      do_row(0);
    } else {
      for (int gg = 0; gg < NG; ++gg) do_row(0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __syncthreads();                                       // nobody still reads A / B: the LDS becomes the epilogue tile

  // ---- epilogue: bias + ReLU in fp32 through an LDS tile [4 pts x 256 voxels of the half][32], one pass per column tile --
  const int out_esz = p.out_f32 ? 4 : kEsz;
  const float act_floor = p.relu ? 0.f : -INFINITY;
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
  unsigned char* mp_b = reinterpret_cast<unsigned char*>(p.mp_out);
  auto cvt_store8 = [&](unsigned char* base, long long row_elems, int col, const float4& f0, const float4& f1) __attribute__((always_inline)) {
    if (out_esz == 4) {
      float4* dst = reinterpret_cast<float4*>(base + (row_elems + col) * 4);
      dst[0] = f0;
      dst[1] = f1;
    } else {
      using E = Elem<DT == NESTI_F32 ? NESTI_BF16 : DT>;
      store_act8<E>(base, row_elems, col, f0, f1, p.split);
    }
  };
  auto epi_pass = [&](auto NN) __attribute__((always_inline)) {
    constexpr int n = decltype(NN)::value;
    const int n_tile = 2 * n_pair + n;
    const float bv = p.bias[n_tile * 32 + l31];
    const int out_col0 = p.out_coff + n_tile * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                        // z = z0 + j, y = (wave - z0 - j) & 7
      const int y = (wave - z0 - j) & 7;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * khalf;      // MFMA row = 8 pt + x
        const int row = ((((m >> 3) * 4 + j) * 8 + y) << 3) + (m & 7);
        *reinterpret_cast<float*>(smem + row * kEpiStrideN + l31 * 4) = fmaxf(fmaf(acc[j][n][r], p.acc_scale, bv), act_floor);
      }
    }
    __syncthreads();
    if (p.mp_mode != 1) {                                // full-resolution rows: 4 lanes x 8 channels = one 32-channel row segment
#pragma unroll 2
      for (int it = 0; it < 8; ++it) {
        const int item = it * kThreadsN + tid;
        const int row = item >> 2, seg = item & 3;
        const int ptl = row >> 8, vox = z0 * 64 + (row & 255);
        if (ptl < np_here) {
          const float4 f0 = *reinterpret_cast<const float4*>(smem + row * kEpiStrideN + seg * 32);
          const float4 f1 = *reinterpret_cast<const float4*>(smem + row * kEpiStrideN + seg * 32 + 16);
          cvt_store8(out_b, ((long long)(p0 + ptl) * 512 + vox) * p.out_cstride, out_col0 + seg * 8, f0, f1);
        }
      }
    }
    if (p.mp_mode != 0) {                                // fused 2^3 / 2 max-pool of the activated values: 4 pts x 2 x 4 x 4 cells
      const int cell = tid >> 2, seg = tid & 3;          // 128 cells x 4 segments of 8 channels
      const int ptl = cell >> 5, cz = (cell >> 4) & 1, cy = (cell >> 2) & 3, cx = cell & 3;
      float4 m0 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), m1 = m0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = ((ptl * 4 + 2 * cz + (q >> 2)) * 8 + 2 * cy + ((q >> 1) & 1)) * 8 + 2 * cx + (q & 1);
        const float4 f0 = *reinterpret_cast<const float4*>(smem + row * kEpiStrideN + seg * 32);
        const float4 f1 = *reinterpret_cast<const float4*>(smem + row * kEpiStrideN + seg * 32 + 16);
        m0.x = fmaxf(m0.x, f0.x); m0.y = fmaxf(m0.y, f0.y); m0.z = fmaxf(m0.z, f0.z); m0.w = fmaxf(m0.w, f0.w);
        m1.x = fmaxf(m1.x, f1.x); m1.y = fmaxf(m1.y, f1.y); m1.z = fmaxf(m1.z, f1.z); m1.w = fmaxf(m1.w, f1.w);
      }
      if (ptl < np_here) {
        const int ovox = ((2 * half + cz) * 4 + cy) * 4 + cx;
        cvt_store8(mp_b, ((long long)(p0 + ptl) * 64 + ovox) * p.mp_cstride, out_col0 + seg * 8, m0, m1);
      }
    }
    __syncthreads();
  };
  epi_pass(std::integral_constant<int, 0>{});
  epi_pass(std::integral_constant<int, 1>{});
}

template <int DT, int K, int MODE, bool WALK>
__global__ __launch_bounds__(kThreadsN) void conv8n_kernel(const ConvParams p) {
  if constexpr (!WALK) {
    conv8n_tile<DT, K, MODE>(p, blockIdx.x, threadIdx.x);
  } else {
    // walking launch (kernels.h: ConvParams::walk), a kernel of its own so that the one-tile-per-workgroup kernel keeps its register
    // allocation: only the tiles below the live row count; the thread index is laundered per trip, otherwise hipcc hoists every
    // per-lane address out of the tile loop and spills
    unsigned n_blocks;
    {
      int npts = p.npoints;
      if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
      const unsigned m_live = (unsigned)((npts + kPtsN - 1) / kPtsN);
        n_blocks = (m_live + 7) / 8 * 8 * 2u * (unsigned)p.n_tiles;
    }
    for (unsigned bid = blockIdx.x; bid < n_blocks; bid += gridDim.x) {
      if (bid != blockIdx.x) __syncthreads();    // the previous tile's epilogue is done with the LDS
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      conv8n_tile<DT, K, MODE>(p, bid, tid);
    }
  }
}

template <int DT, int K, int MODE, bool WALK>
int launch_conv8n_one_w(const ConvParams& p, hipStream_t stream) {
  constexpr int kMaxDevices = 64;
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  NESTI_CHECK_HIP(hipGetDevice(&dev));
  constexpr int lds = lds_bytes_n<K>();
  static_assert(lds <= 163840 && 1024 * kEpiStrideN <= lds, "LDS budget");
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev]) {
    NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8n_kernel<DT, K, MODE, WALK>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev] = true;
  }
  const int groups = (p.m_tiles + 7) / 8;
  const unsigned n_blocks = (unsigned)(groups * 8 * 2 * p.n_tiles);
  dim3 grid(WALK ? std::min(n_blocks, p.walk > 1 ? (unsigned)p.walk : kWalkGrid) : n_blocks), block(kThreadsN);
  hipLaunchKernelGGL((conv8n_kernel<DT, K, MODE, WALK>), grid, block, lds, stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

// p.walk picks the walking instantiation (a separate kernel: kernels.h, ConvParams::walk)
template <int DT, int K, int MODE>
int launch_conv8n_one(const ConvParams& p, hipStream_t stream) {
  return p.walk ? launch_conv8n_one_w<DT, K, MODE, true>(p, stream) : launch_conv8n_one_w<DT, K, MODE, false>(p, stream);
}

template <int DT>
int launch_conv8n_dt(const ConvParams& p, int k, hipStream_t stream) {
  if constexpr (DT == NESTI_F16) {
    if (p.x8) {
      if (!p.aux8_in || p.aux8_stride <= 0 || !p.split) NESTI_FAIL("launch_conv8n: the FP8 cross-term loop needs the side buffer of e4m3 planes and pair outputs");
      if (p.x8_fmt == 6) {
        if (k == 5) return launch_conv8n_one<DT, 5, 3>(p, stream);
        if (k == 3) return launch_conv8n_one<DT, 3, 3>(p, stream);
      }
      if (k == 5) return launch_conv8n_one<DT, 5, 2>(p, stream);
      if (k == 3) return launch_conv8n_one<DT, 3, 2>(p, stream);
    }
  }
  if (p.x8) NESTI_FAIL("launch_conv8n: the FP8 cross-term loop is an f16 pair-mode variant");
  if constexpr (DT != NESTI_F32) {
    if (p.x3native && k == 5) return launch_conv8n_one<DT, 5, 1>(p, stream);
    if (p.x3native && k == 3) return launch_conv8n_one<DT, 3, 1>(p, stream);
  }
  if (p.x3native) NESTI_FAIL("launch_conv8n: the pair K loop is for the 16-bit kernels");
  if (k == 5) return launch_conv8n_one<DT, 5, 0>(p, stream);
  if (k == 3) return launch_conv8n_one<DT, 3, 0>(p, stream);
  NESTI_FAIL("launch_conv8n: kernel size must be 3 or 5");
}

// reads 16 B at kOobN from a workgroup that allocated `lds` bytes of dynamic LDS filled with a non-zero pattern
__global__ void lds_oob_probe_kernel(unsigned* out, unsigned lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (unsigned i = threadIdx.x; i < lds / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0xA5A50000u + i;
  __syncthreads();
  const uint4 v = lds128n(kOobN + 16u * threadIdx.x);
  const uint4 w = lds128n((unsigned)(size_t)(lptr_t)smem + 16u * threadIdx.x);      // in range: must read the pattern back
  atomicOr(&out[0], v.x | v.y | v.z | v.w);
  atomicOr(&out[1], w.x == 0xA5A50000u + 4u * threadIdx.x ? 0u : 1u);
}

}  // namespace

int conv8_selftest() {
  constexpr int kMaxDevices = 64;
  static int state[kMaxDevices] = {};          // 0: not run, 1: passed, 2: failed
  static std::mutex mu;                        // models may be created from several host threads
  std::lock_guard<std::mutex> lk(mu);
  int dev = 0;
  NESTI_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) NESTI_FAIL("conv8_selftest: device index out of range");
  if (state[dev] == 1) return 0;
  if (state[dev] == 0) {
    hipDeviceProp_t prop;
    NESTI_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      state[dev] = 2;
    } else {
      unsigned* d = nullptr;
      unsigned h[2] = {1u, 1u};
      NESTI_CHECK_HIP(hipMalloc((void**)&d, 8));
      NESTI_CHECK_HIP(hipMemset(d, 0, 8));
      for (unsigned lds : {65536u, 147456u, 163840u}) {
        NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_oob_probe_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(lds_oob_probe_kernel, dim3(1), dim3(256), lds, 0, d, lds);
      }
      NESTI_CHECK_HIP(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
      (void)hipFree(d);
      state[dev] = (h[0] == 0u && h[1] == 0u) ? 1 : 2;
    }
  }
  if (state[dev] == 2)
    NESTI_FAIL("this device is not a gfx950 whose out-of-range LDS reads return zero: the 8^3 tap kernels (conv8n.hip) "
               "take their x padding from such reads and would compute wrong convolutions here");
  return 0;
}

// p.m_tiles = groups of 4 points, p.n_tiles = 64-column tile PAIRS, p.n_chunks = 64-byte K chunks, weights packed
// [pair][chunk][tap][2 x 32 rows][64 B]
int launch_conv8n(const ConvParams& p, int dtype, int k, hipStream_t stream) {
  if (p.m_tiles <= 0 || p.n_tiles <= 0) return 0;
  if (p.log2S != 3 || p.s_real) NESTI_FAIL("launch_conv8n: the 8^3 volume only");
  if (p.n_taps != k * k * k) NESTI_FAIL("launch_conv8n: all k^3 taps must be present");
  if (p.point_index) NESTI_FAIL("launch_conv8n: no input gather (k^3 layers never read the routed MuPS tensor)");
  if (p.pool_k > 1 || p.split_tile != p.n_tiles) NESTI_FAIL("launch_conv8n: no fused avg-pool / merged layers");
  if (p.mp_mode == 2) NESTI_FAIL("launch_conv8n: max-pool mode 2 is conv1's (a 1x1x1 layer)");
  if (p.mp_mode != 0 && !p.mp_out) NESTI_FAIL("launch_conv8n: fused max-pool needs an output");
  if (dtype == NESTI_BF16) return launch_conv8n_dt<NESTI_BF16>(p, k, stream);
  if (dtype == NESTI_F16) return launch_conv8n_dt<NESTI_F16>(p, k, stream);
  if (dtype == NESTI_F32) return launch_conv8n_dt<NESTI_F32>(p, k, stream);
  NESTI_FAIL("launch_conv8n: unsupported dtype");
}

}  // namespace nesti
