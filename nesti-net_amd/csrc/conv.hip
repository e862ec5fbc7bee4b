// Implicit-GEMM conv3d / fully-connected kernel on the gfx950 matrix cores.
//
// Replaces tf.nn.conv3d + bias_add + inference batch-norm + ReLU
// (utils/tf_util.py:298-311, 491-494), tf.matmul + bias (+BN, ReLU)
// (utils/tf_util.py:340-351) and -- fused into the epilogue -- the k^3 stride-1 SAME
// tf.nn.avg_pool3d of the inception pool branch (utils/tf_util.py:450-454,
// models/experts_n_est.py:307-310).  BN is folded into weights/bias on the host (model.hip).
//
// Decomposition (one 512-thread workgroup = 8 wave64):
//   M tile  = 512 GEMM rows = whole points (1 point at 8^3, 8 at 4^3, 64 at 2^3, 512 for FC),
//             so every tap of every output voxel finds its input row inside the tile:
//             a K-chunk of the input is staged into LDS ONCE and re-read for all k^3 taps
//             (the halo never goes back to HBM/L2);
//   N tile  = TN (64 or 128) output channels;
//   K loop  = input-channel chunks (128 bytes per row: 64 x bf16/f16 or 32 x f32) outer,
//             taps inner; per (chunk, tap) a TN x 128 B weight tile streams from L2 straight
//             into one of 4 LDS slots (global_load_lds, no VGPR round trip) while the current
//             taps' MFMAs run; two taps per barrier;
//   wave w  = rows [64w, 64w+64) x all TN columns: 2 x (TN/32) tiles of 32x32, fp32 accumulate.
// LDS rows are 128 B with the 16-B slot index XOR-swizzled by (row>>1)&7 (applied on the global
// SOURCE address, since global_load_lds writes lane-linear), which makes the ds_read_b128
// fragment loads of 32 consecutive rows conflict-free.  Zero padding is an address select: a
// lane whose tap leaves the volume reads a dedicated all-zero LDS row.
//
// 1x1x1 / FC layers (one tap) use the KPIPE variant: A and B chunks are double-buffered and
// chunk c+1 streams in while chunk c multiplies.  An inception's conv1 and conv4 share their
// input, so they run as ONE launch (column tiles >= split_tile belong to conv4); because
// avg-pool and a 1x1 conv commute, conv4's tiles average the fp32 accumulators over the k^3
// window (divisor = taps inside the volume) in the epilogue instead of pooling the input first.
//
// 16-bit mode uses v_mfma_f32_32x32x16_{bf16,f16}; the f32 parity mode uses
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) on the same LDS image: a lane's 16 bytes are
// 8 consecutive k (16-bit) or 4 consecutive k (f32) and A and B use the same k order.
#include <type_traits>

#include "kernels.h"
#include "mma.h"

namespace nesti {
namespace {

constexpr int kThreads = 512;
constexpr int kABytes = kTileM * kRowBytes;   // 64 KiB
constexpr int kPoolStride = 272;              // bytes per row of the fp32 [512][64] pooling tile (+16 B pad)

// Row (within the workgroup's 512) that lane-row l of MFMA tile (wave, mi) owns.
// remap == 0: tile t = 2*wave + mi holds rows [32t, 32t+32).
// remap == 1: a padding tap skips whole tiles, and a SIMD's matrix pipe serves waves c and c+4, so the tiles are
//   shaped to be skippable and dealt to the SIMDs as a Latin square (SIMD c gets one tile of every z slab and one
//   of every y slab): every tap then leaves all four pipes (nearly) the same number of live tiles.
//   8^3: tile = x 0..7 x y pair yp x z pair s, s = 2*(wave>>2) + mi, yp = (c - s) & 3, c = wave & 3;
//   4^3: tile = the x-line (y, z) of all 8 points of the workgroup, z = s, y = (c - s) & 3 -- both y and z skip at
//        single-voxel granularity (issued/nominal 0.5625 for the 4^3 taps; 0.66 with (4x,2y) half-planes of 4 points).
//   2^3 (remap == 2, round 5): tile = ONE voxel of 32 of the workgroup's 64 points, so a padding tap kills whole tiles in all three
//        axes (issued / nominal 0.42 for k = 2, 0.30 for the 27 kept taps of k = 4, instead of 1).  The chunk sits in LDS in
//        (voxel block, point) order -- block b = rows [64 b, 64 b + 64) holds voxel vox2(b) of the 64 points -- so a tile is 32
//        CONSECUTIVE LDS rows (natural swizzle, conflict-free) and wave w owns block w; vox2 puts complementary voxels v and 7 - v
//        on the two waves of a SIMD: whatever axes a tap shifts along, each SIMD keeps (nearly) the same number of live tiles.
//        tile_row still returns the row's offset in GLOBAL order (what the epilogues store by).
__device__ __forceinline__ int vox2(int b) { return b < 4 ? b : 11 - b; }      // block <-> voxel, its own inverse
__device__ __forceinline__ int tile_row(int remap, int log2S, int wave, int mi, int l) {
  if (!remap) return wave * 64 + mi * 32 + l;
  if (remap == 2) return ((mi * 32 + l) << 3) + vox2(wave);
  const int c = wave & 3, s = 2 * (wave >> 2) + mi;
  const int q = (c - s) & 3;
  if (log2S == 3) return ((2 * s + (l >> 4)) << 6) + ((2 * q + ((l >> 3) & 1)) << 3) + (l & 7);
  return ((l >> 2) << 6) + (s << 4) + (q << 2) + (l & 3);
}
// 16-B slot swizzle key of an LDS input row: the rows one ds_read_b128 lane group touches must differ in
// (row & 1, key).  Natural and 8^3 tiles: four runs of 4 consecutive rows that differ in row bits 2..3; remapped 4^3
// tiles: runs of 4 rows from different points -- the points of one lane group differ in their low two bits (row bits 6..7).
// Branch-free: key = ((row >> 1) & m_lo) | (((row >> 6) & m_hi) << 1) with (m_lo, m_hi) = (7, 0) or (1, 3).
struct SwzKey {
  int m_lo, m_hi;
  __device__ __forceinline__ SwzKey(int remap, int log2S) {
    const bool pts = remap && log2S == 2;
    m_lo = pts ? 1 : 7;
    m_hi = pts ? 3 : 0;
  }
  __device__ __forceinline__ int operator()(int row) const { return ((row >> 1) & m_lo) | (((row >> 6) & m_hi) << 1); }
};

// X3 (the pair modes NESTI_F16X3 / NESTI_BF16X3): an activation row holds, per 64-channel group, the planes [hi 64 | lo 64]; a K
// chunk is 32 channels -- the LDS row [hi k0..31 | lo k0..31], staged from the two planes, the weight row [W_hi | W_lo] --
// and one set of fragment reads feeds three MFMAs (lo * W_hi, hi * W_hi, hi * W_lo): 1.5x the MFMAs per LDS byte, per
// staged byte and per barrier of the plain kernel.
// one (M tile, N tile) of the layer; `bid` is the position in the XCD-aware block order (blockIdx.x of a one-tile-per-workgroup launch)
// X2 (kernels.h: ConvParams::x2; KPIPE, 16-bit, not X3): plain activations x pair-packed weights.  The A side is the plain loop's:
// 64-channel super-chunks, 128-byte rows (whole cache lines), two buffers.  The B side is the pair packing's: 32-channel chunks, rows
// [W_hi k0..31 | W_lo k0..31], two buffers.  A 32-channel step h of super-chunk C reads the A fragments of slots 4 h .. 4 h + 3 and per
// 16-channel K-step one A fragment set feeds hi * W_hi and hi * W_lo.
template <int DT, int TN, bool KPIPE, bool X3, bool X2 = false>
__device__ __forceinline__ void conv_igemm_tile(const ConvParams& p, const unsigned bid, const int tid_in) {
  static_assert(!X2 || (KPIPE && !X3 && DT != NESTI_F32), "X2 is a 16-bit one-tap variant");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NI = TN / 32;
  constexpr int kBTile = TN * kRowBytes;
  constexpr int kBVec = TN / 64;                  // 1-KiB pieces of a weight tile per wave
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;
  constexpr int kNA = KPIPE ? 2 : 1;              // A buffers
  constexpr int kARow = kRowBytes;                // bytes of one A row in LDS (X2 too: the A side is staged in whole 128-byte lines --
                                                  // 64-byte rows fetched every line twice and made the loop 68 % slower than plain)
  constexpr int kABuf = kTileM * kARow;           // one A buffer
  constexpr int kAPieces = kABuf / (8 * 1024);    // 1-KiB staging pieces per wave and chunk
  unsigned char* As = smem;
  unsigned char* Bs = smem + kNA * kABuf;
  constexpr int kNB = KPIPE ? 2 : 4;              // weight-tile slots (general variant: 2 groups of 2 taps)
  constexpr int kZeroOff = kNA * kABuf + kNB * kBTile;   // all-zero 128-B row (general variant only)

  const int tid = tid_in, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware block -> tile map: the 8 blocks of a dispatch group land on the 8 XCDs; each XCD
  // keeps its M tile and walks the N tiles, so the staged input stays in that XCD's L2.
  const int xcd = bid & 7, grp = bid >> 3;
  const int n_tile = grp % p.n_tiles;
  const int m_tile = (grp / p.n_tiles) * 8 + xcd;
  if (m_tile >= p.m_tiles) return;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int remap = KPIPE ? 0 : p.remap;
  const SwzKey swz_key(remap, p.log2S);
  const int log2S = p.log2S, log2V = 3 * log2S;
  const int S = 1 << log2S, V = 1 << log2V;
  const unsigned Sb = p.s_real ? (unsigned)p.s_real : (unsigned)S;   // real volume edge (3 inside a 4^3 index space)
  const long long total_rows = (long long)npts << log2V;
  const long long r0 = (long long)m_tile * kTileM;
  if (r0 >= total_rows) return;

  // ---- A staging: wave w, piece j covers LDS rows (w*8+j)*8 .. +8; lane -> (row, slot') ----------
  long long a_off[kAPieces];
#pragma unroll
  for (int j = 0; j < kAPieces; ++j) {
    const int row_l = (wave * 8 + j) * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ swz_key(row_l);   // inverse swizzle on the SOURCE
    // remap == 2: LDS row (block b, point pt) <- global row 8 pt + vox2(b)
    const long long gr = r0 + (remap == 2 ? ((row_l & 63) << 3) + vox2(row_l >> 6) : row_l);
    if (gr < total_rows || X2) {                   // X2 counts its LDS-DMA instructions (s_waitcnt vmcnt(N)): a row beyond the live rows
      const long long grs = gr < total_rows ? gr : r0;   // stages the tile's first row instead (its outputs are never stored)
      long long pt = grs >> log2V;
      const long long vox = grs & (V - 1);
      if (p.point_index) pt = p.point_index[pt];
      a_off[j] = (((pt << log2V) + vox) * p.in_cstride + p.in_coff) * kEsz +
                 (X3 ? (slot & 3) * 16 + (slot >> 2) * (2 * kSplitGroup) : slot * 16);
    } else {
      a_off[j] = -1;
    }
  }
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
  const unsigned char* w_tile =
      reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)n_tile * p.n_chunks * p.n_taps * kBTile;

  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;   // LDS byte address of the dynamic segment
  auto stage_a = [&](int c, int a_buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < kAPieces; ++j)
      if (a_off[j] >= 0)
        glds16(in_b + a_off[j] + (X3 ? (long long)(c >> 1) * (2 * kPairPlanes * kSplitGroup) + (c & 1) * 64
                                     : (long long)c * p.in_chunk_bytes),      // X2: c counts 64-channel SUPER-chunks here
               lds0 + a_buf * kABuf + (wave * kAPieces + j) * 1024);
  };
  auto stage_b = [&](int c, int t, int b_buf) __attribute__((always_inline)) {
    const unsigned char* src = w_tile + ((size_t)c * p.n_taps + t) * kBTile;
#pragma unroll
    for (int q = 0; q < kBVec; ++q) {
      const int piece = wave * kBVec + q;
      glds16(src + piece * 1024 + lane * 16, lds0 + kNA * kABuf + b_buf * kBTile + piece * 1024);
    }
  };

  // ---- per-lane fragment coordinates ----------------------------------------------------
  int rz[2], ry[2], rx[2], rrow[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    // the LDS row the lane's fragment row lives in (remap == 2: LDS order differs from global order, see tile_row)
    rrow[mi] = remap == 2 ? wave * 64 + mi * 32 + (lane & 31) : tile_row(remap, log2S, wave, mi, lane & 31);
    const int vox = remap == 2 ? vox2(wave) : rrow[mi] & (V - 1);
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

  // one (chunk, tap) step: 4 K-steps of 2 x NI MFMAs; the fragments of K-step kk+1 are read from LDS
  // before the MFMAs of K-step kk are issued, so the LDS latency hides behind the matrix pipe.
  // live0 / live1 (wave-uniform): which of the wave's two M tiles take part in this tap; a dead tile's MFMAs are
  // branched over (its fragment reads hit the zero row).
  auto compute = [&](const bool live0, const bool live1, const unsigned char* Acur, const unsigned char* Bcur,
                     const int (&a_addr)[2], const int (&a_sw)[2]) __attribute__((always_inline)) {
    if constexpr (X3) {
      // fragment sets of one 16-channel K-step t: hi / lo of the two M tiles, W_hi / W_lo of the NI N tiles.  The three
      // products run lo*W_hi, hi*W_hi, hi*W_lo, and each set is reloaded for step t + 1 as soon as its last product has
      // been issued, so every read has at least 2 x NI MFMAs to land (no extra registers: 4 + 2 NI fragments).
      uint4 ah[2], al[2], bh[NI], bl[NI];
      auto ld_a = [&](int t, int lo, uint4 (&f)[2]) __attribute__((always_inline)) {
        const int slot = lo * 4 + t * 2 + khalf;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) f[mi] = *reinterpret_cast<const uint4*>(Acur + a_addr[mi] + ((slot ^ a_sw[mi]) << 4));
      };
      auto ld_b = [&](int t, int lo, uint4 (&f)[NI]) __attribute__((always_inline)) {
        const int slot = lo * 4 + t * 2 + khalf;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          f[ni] = *reinterpret_cast<const uint4*>(Bcur + ni * 32 * kRowBytes + b_row + ((slot ^ b_sw) << 4));
      };
      auto mm = [&](const uint4 (&af)[2], const uint4 (&bf)[NI]) __attribute__((always_inline)) {
        if (KPIPE || __builtin_expect(live0, 1)) {      // the live path falls through: no taken branch per MFMA group
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma<DT>(acc[0][ni], af[0], bf[ni]);
        }
        if (KPIPE || __builtin_expect(live1, 1)) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma<DT>(acc[1][ni], af[1], bf[ni]);
        }
      };
      ld_a(0, 1, al);
      ld_b(0, 0, bh);
      ld_a(0, 0, ah);
      ld_b(0, 1, bl);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        mm(al, bh);
        if (t < 1) ld_a(t + 1, 1, al);
        __builtin_amdgcn_sched_barrier(0);
        mm(ah, bh);
        if (t < 1) ld_b(t + 1, 0, bh);
        __builtin_amdgcn_sched_barrier(0);
        mm(ah, bl);
        if (t < 1) {
          ld_a(t + 1, 0, ah);
          ld_b(t + 1, 1, bl);
        }
      }
      return;
    }
    if constexpr (X2) {
      // per 16-channel K-step t: one A fragment set (the lane's 16 B of the 64-byte row), W_hi and W_lo of the NI column tiles
      uint4 a0[2], a1[2], bh[NI], bl[NI];
      const int xh = live0 ? 0 : 1;                // (the live0 argument carries the 32-channel half of the A super-chunk)
      auto ld_a = [&](int t, uint4 (&f)[2]) __attribute__((always_inline)) {
        const int slot = xh * 4 + t * 2 + khalf;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) f[mi] = *reinterpret_cast<const uint4*>(Acur + a_addr[mi] + ((slot ^ a_sw[mi]) << 4));
      };
      auto ld_b = [&](int t, int lo, uint4 (&f)[NI]) __attribute__((always_inline)) {
        const int slot = lo * 4 + t * 2 + khalf;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          f[ni] = *reinterpret_cast<const uint4*>(Bcur + ni * 32 * kRowBytes + b_row + ((slot ^ b_sw) << 4));
      };
      auto mm = [&](const uint4 (&af)[2], const uint4 (&bf)[NI]) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) mma<DT>(acc[mi][ni], af[mi], bf[ni]);
      };
      ld_a(0, a0);
      ld_b(0, 0, bh);
      ld_b(0, 1, bl);
      ld_a(1, a1);
      __builtin_amdgcn_sched_barrier(0);
      mm(a0, bh);
      ld_b(1, 0, bh);
      __builtin_amdgcn_sched_barrier(0);
      mm(a0, bl);
      ld_b(1, 1, bl);
      __builtin_amdgcn_sched_barrier(0);
      mm(a1, bh);
      mm(a1, bl);
      return;
    }
    uint4 a[2][2], b[2][NI];
    auto load_frags = [&](int kk, uint4 (&af)[2], uint4 (&bf)[NI]) __attribute__((always_inline)) {
      const int slot = kk * 2 + khalf;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        af[mi] = *reinterpret_cast<const uint4*>(Acur + a_addr[mi] + ((slot ^ a_sw[mi]) << 4));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        bf[ni] = *reinterpret_cast<const uint4*>(Bcur + ni * 32 * kRowBytes + b_row + ((slot ^ b_sw) << 4));
    };
    load_frags(0, a[0], b[0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) load_frags(kk + 1, a[(kk + 1) & 1], b[(kk + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this K-step's MFMAs (hipcc sinks it otherwise)
      if (KPIPE || __builtin_expect(live0, 1)) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma<DT>(acc[0][ni], a[kk & 1][0], b[kk & 1][ni]);
      }
      if (KPIPE || __builtin_expect(live1, 1)) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mma<DT>(acc[1][ni], a[kk & 1][1], b[kk & 1][ni]);
      }
    }
  };

  if constexpr (KPIPE) {
    // ---- one tap: software pipeline over input-channel chunks --------------------------------
    int a_addr[2], a_sw[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) { a_addr[mi] = rrow[mi] * kARow; a_sw[mi] = swz_key(rrow[mi]); }
    if constexpr (X2) {
      // p.n_chunks counts the 32-channel weight chunks c = 2 C + h.  One barrier per chunk, at its top: it publishes B(c) (and, for
      // h == 0, the A super-chunk C) and frees the buffers chunk c - 1 was multiplied from; right after it B(c + 1) goes to the other B
      // buffer and, for h == 0, A(C + 1) to the other A buffer -- so the A side is two 32-channel steps ahead, the B side one.
      // LDS-DMA completes in order and every wave issues the same number of instructions per stage (rows beyond the live rows stage
      // the tile's first row), so at the top of an h == 1 chunk "B(c) has landed" is s_waitcnt vmcnt(kAPieces): only the A super-chunk
      // issued after it may still be in flight; at the top of an h == 0 chunk everything issued must have landed.
      stage_a(0, 0);
      stage_b(0, 0, 0);
      for (int c = 0; c < p.n_chunks; ++c) {
        const int h = c & 1, C = c >> 1;
        if (h == 0 || c + 1 >= p.n_chunks) wait_vm0();       // (the last chunk's predecessor issued no A behind its B)
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAPieces) : "memory");
        __syncthreads();
#ifndef IGEMM_X2_NOFILL    // timing-only experiment (wrong results): the X2 K loop without its LDS fill
        if (c + 1 < p.n_chunks) stage_b(c + 1, 0, (c + 1) & 1);
        if (h == 0 && c + 2 < p.n_chunks) stage_a(C + 1, (C + 1) & 1);
#endif
        compute(h == 0, true, As + (C & 1) * kABuf, Bs + (c & 1) * kBTile, a_addr, a_sw);
      }
      __syncthreads();                               // the epilogue reuses the LDS
    } else {
    stage_a(0, 0);
    stage_b(0, 0, 0);
    wait_vm0();
    __syncthreads();
#ifdef IGEMM_NO_MAIN      // timing-only experiment (wrong results): the epilogue alone
    for (int c = 0; c < 0; ++c) {
#else
    for (int c = 0; c < p.n_chunks; ++c) {
#endif
      const int cur = c & 1;
      if (c + 1 < p.n_chunks) {
        stage_a(c + 1, cur ^ 1);
        stage_b(c + 1, 0, cur ^ 1);
      }
      compute(true, true, As + cur * kABuf, Bs + cur * kBTile, a_addr, a_sw);
      wait_vm0();
      __syncthreads();
    }
    }
  } else {
    // ---- k^3 taps: A chunk resident, weight tiles double-buffered -----------------------------
    if (tid < 32) reinterpret_cast<uint32_t*>(smem + kZeroOff)[tid] = 0u;
    for (int c = 0; c < p.n_chunks; ++c) {
      __syncthreads();   // every wave is done with the previous chunk's tiles
      stage_a(c, 0);
      stage_b(c, 0, 0);
      if (p.n_taps > 1) stage_b(c, 1, 1);
      wait_vm0();
      __syncthreads();
      // two taps per barrier: group g = step & 1 holds the weight tiles of taps 2*step and 2*step+1
      const int n_steps = (p.n_taps + 1) >> 1;
      for (int st = 0; st < n_steps; ++st) {
        const int grp_cur = st & 1;
        if (st + 1 < n_steps) {
          const int t2 = 2 * (st + 1);
          stage_b(c, t2, (grp_cur ^ 1) * 2);
          if (t2 + 1 < p.n_taps) stage_b(c, t2 + 1, (grp_cur ^ 1) * 2 + 1);
        }
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * st + u;
          if (t >= p.n_taps) break;
          const int dz = p.tap[t][0], dy = p.tap[t][1], dx = p.tap[t][2];
          const int shift = dz * (1 << (2 * log2S)) + dy * S + dx;
          int a_addr[2], a_sw[2];
          bool live[2];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const bool ok = ((unsigned)(rz[mi] + dz) < Sb) & ((unsigned)(ry[mi] + dy) < Sb) & ((unsigned)(rx[mi] + dx) < Sb);
            // remap == 2: the source voxel's block instead of a linear row shift (ok is wave-uniform there)
            const int srow = remap == 2 ? rrow[mi] + ((vox2((vox2(wave) + shift) & 7) - wave) << 6) : rrow[mi] + shift;
            a_addr[mi] = ok ? srow * kRowBytes : kZeroOff;   // padding tap -> the zero row
            a_sw[mi] = ok ? swz_key(srow) : 0;
            live[mi] = __ballot(ok) != 0ull;                 // an all-padding tile issues no MFMAs
          }
          if (live[0] | live[1]) compute(live[0], live[1], As, Bs + (grp_cur * 2 + u) * kBTile, a_addr, a_sw);
        }
        wait_vm0();
        __syncthreads();
      }
    }
  }

#ifdef IGEMM_EMPTY        // timing-only experiment: workgroup dispatch + the first chunk's staging, nothing else
  if (KPIPE) return;
#endif
  // ---- epilogue (the barrier that ended the last step guarantees nobody still reads A/B) -------
#ifdef IGEMM_NO_EPILOGUE  // timing-only experiment (wrong results): the main loop alone (the accumulators stay live through a
  if (KPIPE) {            // store that never happens)
    float sink = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += acc[mi][ni][r];
    if (sink == 1.2345e-31f) reinterpret_cast<float*>(p.out)[0] = sink;
    return;
  }
#endif
  const int out_esz = p.out_f32 ? 4 : kEsz;
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
#ifdef IGEMM_NO_STORE     // timing-only experiment: the epilogue's LDS / VALU work without its global stores (a row bound the compiler
  const long long total_rows_all = total_rows;   // cannot see through: every `gr < total_rows` below fails at run time)
#define total_rows (p.npoints < 0 ? total_rows_all : 0ll)
#endif
  const bool second = n_tile >= p.split_tile;            // conv4 half of a merged conv1|conv4 launch
  const int n_local = second ? n_tile - p.split_tile : n_tile;
  const int out_col0 = (second ? p.out_coff2 : p.out_coff) + n_local * TN;
  const float* bias = p.bias + n_tile * TN;
  const float act_floor = p.relu ? 0.f : -INFINITY;      // ReLU as one v_max
  // kernels.h: ConvParams::aux8_out -- power-of-two pre-scales of the e4m3 planes (exact multiplications)
  const float aux_mul_lo = __builtin_ldexpf(1.f, p.x8_sa), aux_mul_hi = __builtin_ldexpf(1.f, p.x8_sc);

  if (KPIPE && second && p.pool_k > 1) {
    // avg_pool3d(k, SAME, stride 1) of the pre-activation: per 64-column half the fp32 accumulators go
    // through an LDS tile [512][64 (+4 pad)]; a thread then owns whole x-lines of one 4-channel group, loads
    // each of the K^2 neighbouring lines once (an out-of-volume line reads a zero row) and forms all S
    // windowed sums from registers.  S and K are compile-time so every load is independent.
    unsigned char* const zero16 = smem + kTileM * kPoolStride;
    if (tid < 4) reinterpret_cast<uint32_t*>(zero16)[tid] = 0u;
    // SR = real volume edge (== S except for the 3^3 grid embedded in 4^3: dead rows neither contribute nor count)
    auto pool_lines = [&](auto SS, auto SRR, auto KK, int nh) __attribute__((always_inline)) {
      constexpr int S_ = decltype(SS)::value, SR_ = decltype(SRR)::value, K_ = decltype(KK)::value;
      constexpr int lo = (K_ - 1) / 2;
      constexpr int log2S_ = (S_ == 8) ? 3 : (S_ == 4) ? 2 : 1;
      // item -> (x-line, 4-channel group): a wave's four 16-lane groups are the four lines (y0, z0) of one 2 x 2 (y, z)
      // cell, so that the 2^3 / 2 max-pool that may follow (mp_mode2) is an x-pair max in registers plus two
      // cross-lane maxima (lane ^ 16, lane ^ 32) -- no second pass through LDS
      constexpr int H_ = S_ / 2, log2H_ = log2S_ - 1;
#pragma unroll 1
      for (int it = 0; it < 16 / S_; ++it) {
        const int cg = lane & 15, y0 = (lane >> 4) & 1, z0 = lane >> 5;
        const int cell = it * 8 + wave;
        const int cy = cell & (H_ - 1), cz = (cell >> log2H_) & (H_ - 1), cpt = cell >> (2 * log2H_);
        const int y = 2 * cy + y0, z = 2 * cz + z0;
        const int line = (((cpt << log2S_) + z) << log2S_) + y;
        const int row0 = line << log2S_;
        // separable box sum: add the K^2 neighbouring x-lines first, then one x window over the column sums
        float4 colsum[S_];
#pragma unroll
        for (int x = 0; x < S_; ++x) colsum[x] = make_float4(0.f, 0.f, 0.f, 0.f);
        int nz = 0, ny = 0;
#pragma unroll
        for (int a = 0; a < K_; ++a) nz += ((unsigned)(z + a - lo) < (unsigned)SR_) ? 1 : 0;
#pragma unroll
        for (int b = 0; b < K_; ++b) ny += ((unsigned)(y + b - lo) < (unsigned)SR_) ? 1 : 0;
        // the axis whose neighbours sit in one lane's accumulators was already summed in registers (pool_half):
        // S = 8 -> y, S = 4 -> z; only the other one is walked here
        constexpr int a_lo = (S_ == 4) ? lo : 0, a_hi = (S_ == 4) ? lo + 1 : K_;
        constexpr int b_lo = (S_ == 8) ? lo : 0, b_hi = (S_ == 8) ? lo + 1 : K_;
#pragma unroll 1   // keep at most K lines x S loads in flight: full unrolling spills
        for (int a = a_lo; a < a_hi; ++a) {
#pragma unroll
          for (int b = b_lo; b < b_hi; ++b) {
            const bool ok = ((unsigned)(z + a - lo) < (unsigned)SR_) & ((unsigned)(y + b - lo) < (unsigned)SR_);
            const int nrow0 = row0 + (((a - lo) * S_ + (b - lo)) << log2S_);
            const unsigned char* base = ok ? smem + nrow0 * kPoolStride + cg * 16 : zero16;
            const int stride = ok ? kPoolStride : 0;
            float4 v[S_];
#pragma unroll
            for (int x = 0; x < S_; ++x) v[x] = *reinterpret_cast<const float4*>(base + x * stride);
#pragma unroll
            for (int x = 0; x < S_; ++x) { colsum[x].x += v[x].x; colsum[x].y += v[x].y; colsum[x].z += v[x].z; colsum[x].w += v[x].w; }
          }
        }
        float4 sum[S_];
#pragma unroll
        for (int x = 0; x < S_; ++x) {
          sum[x] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int c = 0; c < K_; ++c) {
            const int xx = x + c - lo;
            if (xx >= 0 && xx < SR_) { sum[x].x += colsum[xx].x; sum[x].y += colsum[xx].y; sum[x].z += colsum[xx].z; sum[x].w += colsum[xx].w; }
          }
        }
        const float4 bb = *reinterpret_cast<const float4*>(bias + nh * 64 + cg * 4);
        const int col = out_col0 + nh * 64 + cg * 4;
        float4 o[S_];
#pragma unroll
        for (int x = 0; x < S_; ++x) {
          int nx = 0;
#pragma unroll
          for (int c = 0; c < K_; ++c) nx += (x + c - lo >= 0 && x + c - lo < SR_) ? 1 : 0;
          const float inv = p.acc_scale / (float)max(1, nz * ny * nx);   // taps inside the volume (utils/tf_util.py:450-454); 0 only on dead rows
          o[x] = make_float4(fmaxf(sum[x].x * inv + bb.x, act_floor), fmaxf(sum[x].y * inv + bb.y, act_floor),
                             fmaxf(sum[x].z * inv + bb.z, act_floor), fmaxf(sum[x].w * inv + bb.w, act_floor));
        }
        if (!p.mp_mode2) {
#pragma unroll
          for (int x = 0; x < S_; ++x) {
            const long long gr = r0 + row0 + x;
            if (gr < total_rows) {
              if (out_esz == 4) {
                *reinterpret_cast<float4*>(out_b + (gr * p.out_cstride + col) * 4) = o[x];
              } else {
                using E = Elem<DT == NESTI_F32 ? NESTI_BF16 : DT>;
                store_act4<E>(out_b, gr * p.out_cstride, col, o[x].x, o[x].y, o[x].z, o[x].w, p.split);
              }
            }
          }
        } else {
          // tf.nn.max_pool3d 2^3 / 2 (utils/tf_util.py:424-428) of the activated values: x pairs here, then the cell's other
          // three lines from lanes ^ 16 (y) and ^ 32 (z); the cell's first 16 lanes store the pooled row segments
          unsigned char* mp_b2 = reinterpret_cast<unsigned char*>(p.mp_out);
#pragma unroll
          for (int xc = 0; xc < H_; ++xc) {
            float4 m = make_float4(fmaxf(o[2 * xc].x, o[2 * xc + 1].x), fmaxf(o[2 * xc].y, o[2 * xc + 1].y),
                                   fmaxf(o[2 * xc].z, o[2 * xc + 1].z), fmaxf(o[2 * xc].w, o[2 * xc + 1].w));
#pragma unroll
            for (int d = 16; d <= 32; d <<= 1) {
              m.x = fmaxf(m.x, __shfl_xor(m.x, d, 64)); m.y = fmaxf(m.y, __shfl_xor(m.y, d, 64));
              m.z = fmaxf(m.z, __shfl_xor(m.z, d, 64)); m.w = fmaxf(m.w, __shfl_xor(m.w, d, 64));
            }
            const long long go = (r0 >> 3) + (((((long long)cpt << log2H_) + cz) << log2H_) + cy) * H_ + xc;
            if (lane < 16 && go < (total_rows >> 3)) {
              if (out_esz == 4) {
                *reinterpret_cast<float4*>(mp_b2 + (go * p.mp_cstride + col) * 4) = m;
              } else {
                using E = Elem<DT == NESTI_F32 ? NESTI_BF16 : DT>;
                store_act4<E>(mp_b2, go * p.mp_cstride, col, m.x, m.y, m.z, m.w, p.split);
              }
            }
          }
        }
      }
    };
    // one call per 64-column half with a compile-time index: runtime-indexed register arrays go to scratch.
    // A lane's 32 accumulators of one column cover row bits {0,1} (r&3), {3,4} (r>>2) and 5 (mi): at 8^3 that is
    // the whole y axis, at 4^3 the whole z axis.  That axis of the box sum is taken here on registers (K-1 adds
    // per value, no LDS), which divides the LDS line reads of pool_lines by K.
    auto pool_half = [&](auto NH, auto SS, auto SRR, auto KK) __attribute__((always_inline)) {
      constexpr int nh = decltype(NH)::value, S_ = decltype(SS)::value, SR_ = decltype(SRR)::value, K_ = decltype(KK)::value;
      constexpr int lo = (K_ - 1) / 2;
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int ni = nh * 2 + n2;
        const int col = n2 * 32 + (lane & 31);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = 0.f;
            if constexpr (S_ == 8) {            // y = (r>>2) + 4*mi
              const int y = (r >> 2) + 4 * mi;
#pragma unroll
              for (int b = 0; b < K_; ++b) {
                const int yy = y + b - lo;
                if (yy >= 0 && yy < 8) v += acc[yy >> 2][ni][((yy & 3) << 2) | (r & 3)];
              }
            } else if constexpr (S_ == 4) {     // z = (r>>3) + 2*mi
              const int z = (r >> 3) + 2 * mi;
#pragma unroll
              for (int a = 0; a < K_; ++a) {
                const int zz = z + a - lo;
                if (zz >= 0 && zz < SR_) v += acc[zz >> 1][ni][((zz & 1) << 3) | (r & 7)];
              }
            } else {
              v = acc[mi][ni][r];
            }
            const int row = wave * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            *reinterpret_cast<float*>(smem + row * kPoolStride + col * 4) = v;
          }
      }
      __syncthreads();
      pool_lines(SS, SRR, KK, nh);
      __syncthreads();
    };
    auto pool_tile = [&](auto SS, auto SRR, auto KK) __attribute__((always_inline)) {
      pool_half(std::integral_constant<int, 0>{}, SS, SRR, KK);
      if constexpr (TN == 128) pool_half(std::integral_constant<int, 1>{}, SS, SRR, KK);
    };
    using std::integral_constant;
    if (log2S == 3 && p.pool_k == 3) pool_tile(integral_constant<int, 8>{}, integral_constant<int, 8>{}, integral_constant<int, 3>{});
    else if (log2S == 2 && p.pool_k == 2 && Sb == 3) pool_tile(integral_constant<int, 4>{}, integral_constant<int, 3>{}, integral_constant<int, 2>{});
    else if (log2S == 2 && p.pool_k == 2) pool_tile(integral_constant<int, 4>{}, integral_constant<int, 4>{}, integral_constant<int, 2>{});
    else if (log2S == 2 && p.pool_k == 3) pool_tile(integral_constant<int, 4>{}, integral_constant<int, 4>{}, integral_constant<int, 3>{});
    else if (log2S == 1 && p.pool_k == 2) pool_tile(integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{});
    return;
  }

  if (p.mp_mode != 0 && !second) {
    // Fused 2^3 stride-2 max-pool (and optionally the full-resolution store): per 64-column half the ACTIVATED
    // values go through the fp32 LDS tile [512][64 (+4 pad)]; a thread then reduces whole 2x2x2 cells of one
    // 4-channel group.  max(relu(x + b)) is taken on the final values, exactly like pooling the stored tensor.
    unsigned char* mp_b = reinterpret_cast<unsigned char*>(p.mp_out);
    const int Vo = V >> 3, So = S >> 1, log2So = log2S - 1;
    auto cvt_store = [&](unsigned char* base, long long row_elems, int col, const float4& v) __attribute__((always_inline)) {
      if (out_esz == 4) {
        *reinterpret_cast<float4*>(base + (row_elems + col) * 4) = v;
      } else {
        using E = Elem<DT == NESTI_F32 ? NESTI_BF16 : DT>;
        store_act4<E>(base, row_elems, col, v.x, v.y, v.z, v.w, p.split);
      }
    };
    auto mp_half = [&](auto NH) __attribute__((always_inline)) {
      constexpr int nh = decltype(NH)::value;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int ni = nh * 2 + n2;
          const int col = n2 * 32 + (lane & 31);
          const float bv = bias[ni * 32 + (lane & 31)];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = tile_row(remap, log2S, wave, mi, (r & 3) + 8 * (r >> 2) + 4 * khalf);
            *reinterpret_cast<float*>(smem + row * kPoolStride + col * 4) = fmaxf(fmaf(acc[mi][ni][r], p.acc_scale, bv), act_floor);
          }
        }
      __syncthreads();
      if (p.mp_mode == 2) {   // full-resolution rows, 16 consecutive lanes = one 64-channel row segment
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int item = it * kThreads + tid;
          const int row = item >> 4, cg = item & 15;
          const long long gr = r0 + row;
          if (gr < total_rows) {
            const float4 v4 = *reinterpret_cast<const float4*>(smem + row * kPoolStride + cg * 16);
            cvt_store(out_b, gr * p.out_cstride, out_col0 + nh * 64 + cg * 4, v4);
            if (p.aux8_out && p.x8_fmt != 6)   // the e4m3 planes of conv1's outputs for the FP8 cross terms of the block's tap layers (conv8n.hip X8)
              store_aux8_4(reinterpret_cast<unsigned char*>(p.aux8_out) + gr * p.aux8_stride, n_local * TN + nh * 64 + cg * 4,
                           v4.x, v4.y, v4.z, v4.w, aux_mul_lo, aux_mul_hi);
          }
        }
        if (p.aux8_out && p.x8_fmt == 6) {     // the FP6 form: one thread = one row's 16-channel chunk (its block scale needs all 16)
#pragma unroll 2
          for (int it = 0; it < 512 * 4 / kThreads; ++it) {
            const int item = it * kThreads + tid;
            const int row = item >> 2, c16 = item & 3;
            const long long gr = r0 + row;
            if (gr < total_rows) {
              const unsigned char* src = smem + row * kPoolStride + c16 * 64;
              store_aux6_16(reinterpret_cast<unsigned char*>(p.aux8_out) + gr * p.aux8_stride, n_local * TN + nh * 64 + c16 * 16,
                            *reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 16),
                            *reinterpret_cast<const float4*>(src + 32), *reinterpret_cast<const float4*>(src + 48));
            }
          }
        }
      }
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
          cvt_store(mp_b, go * p.mp_cstride, out_col0 + nh * 64 + cg * 4, m);
      }
      __syncthreads();
    };
    mp_half(std::integral_constant<int, 0>{});
    if constexpr (TN == 128) mp_half(std::integral_constant<int, 1>{});
    return;
  }

  // plain epilogue: bias + ReLU in fp32, transpose 32 x 64 blocks through a wave-private fp32 LDS scratch, then
  // convert on the way out (one lane = 16 output bytes) -- no per-element branches, no sub-dword LDS writes
  unsigned char* scratch = smem + wave * (32 * kPoolStride);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int nh = 0; nh < TN / 64; ++nh) {
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int ni = nh * 2 + n2;
        const int col = n2 * 32 + (lane & 31);
        const float bv = bias[ni * 32 + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;
          *reinterpret_cast<float*>(scratch + row * kPoolStride + col * 4) = fmaxf(fmaf(acc[mi][ni][r], p.acc_scale, bv), act_floor);
        }
      }
      if (out_esz == 4) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {                     // 16 lanes x 16 B = one 64-float row segment
          const int row = it * 4 + (lane >> 4), cpos = lane & 15;
          const uint4 v = *reinterpret_cast<const uint4*>(scratch + row * kPoolStride + cpos * 16);
          const long long gr = r0 + tile_row(remap, log2S, wave, mi, row);
          if (gr < total_rows)
            *reinterpret_cast<uint4*>(out_b + (gr * p.out_cstride + out_col0 + nh * 64) * 4 + cpos * 16) = v;
        }
      } else {
        using E = Elem<DT == NESTI_F32 ? NESTI_BF16 : DT>;
#pragma unroll
        for (int it = 0; it < 4; ++it) {                     // 8 lanes x 16 B = one 64-element row segment
          const int row = it * 8 + (lane >> 3), cpos = lane & 7;
          const float4 f0 = *reinterpret_cast<const float4*>(scratch + row * kPoolStride + cpos * 32);
          const float4 f1 = *reinterpret_cast<const float4*>(scratch + row * kPoolStride + cpos * 32 + 16);
          const long long gr = r0 + tile_row(remap, log2S, wave, mi, row);
          if (gr < total_rows) {
            store_act8<E>(out_b, gr * p.out_cstride, out_col0 + nh * 64 + cpos * 8, f0, f1, p.split);
            if (p.aux8_out && !second && p.x8_fmt != 6)
              store_aux8_8(reinterpret_cast<unsigned char*>(p.aux8_out) + gr * p.aux8_stride, n_local * TN + nh * 64 + cpos * 8, f0, f1,
                           aux_mul_lo, aux_mul_hi);
          }
        }
        if (p.aux8_out && !second && p.x8_fmt == 6) {      // the FP6 form: 4 lanes x 16 channels = one 64-channel row segment
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int row = it * 16 + (lane >> 2), c16 = lane & 3;
            const unsigned char* src = scratch + row * kPoolStride + c16 * 64;
            const long long gr = r0 + tile_row(remap, log2S, wave, mi, row);
            if (gr < total_rows)
              store_aux6_16(reinterpret_cast<unsigned char*>(p.aux8_out) + gr * p.aux8_stride, n_local * TN + nh * 64 + c16 * 16,
                            *reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 16),
                            *reinterpret_cast<const float4*>(src + 32), *reinterpret_cast<const float4*>(src + 48));
          }
        }
      }
    }
  }
}

#ifdef IGEMM_NO_STORE
#undef total_rows
#endif

template <int DT, int TN, bool KPIPE, bool X3, bool WALK, bool X2 = false>
__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(const ConvParams p) {
  if constexpr (!WALK) {
    conv_igemm_tile<DT, TN, KPIPE, X3, X2>(p, blockIdx.x, threadIdx.x);
  } else {
    // walking launch (kernels.h: ConvParams::walk), a kernel of its own so that the one-tile-per-workgroup kernel keeps its register
    // allocation: only the tiles below the live row count; the thread index is laundered per trip, otherwise hipcc hoists every
    // per-lane address out of the tile loop and spills
    unsigned n_blocks;
    {
      int npts = p.npoints;
      if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
      const long long rows = (long long)npts << (3 * p.log2S);
      const unsigned m_live = (unsigned)((rows + kTileM - 1) / kTileM);
        n_blocks = (m_live + 7) / 8 * 8 * (unsigned)p.n_tiles;
    }
    for (unsigned bid = blockIdx.x; bid < n_blocks; bid += gridDim.x) {
      if (bid != blockIdx.x) __syncthreads();    // the previous tile's epilogue is done with the LDS
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      conv_igemm_tile<DT, TN, KPIPE, X3, X2>(p, bid, tid);
    }
  }
}

template <int TN, bool KPIPE>
constexpr size_t lds_bytes() {
  constexpr size_t kPoolTile = (size_t)kTileM * kPoolStride + 16;         // fp32 pooling tile of the epilogues
  constexpr size_t loop = KPIPE ? (size_t)2 * kABytes + 2 * TN * kRowBytes         // 160 KiB at TN = 128
                                : (size_t)kABytes + 4 * TN * kRowBytes + kRowBytes; // 4 weight slots + zero row
  return loop > kPoolTile ? loop : kPoolTile;
}

template <int DT, int TN, bool KPIPE, bool X3, bool WALK, bool X2 = false>
int launch_one_w(const ConvParams& p, hipStream_t stream) {
  // the dynamic-LDS opt-in is a per-device function attribute: one flag per device, not per process
  constexpr int kMaxDevices = 64;
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  NESTI_CHECK_HIP(hipGetDevice(&dev));
  constexpr size_t lds = lds_bytes<TN, KPIPE>();
  static_assert(lds <= 163840, "LDS budget");
  static_assert(!KPIPE || lds >= (size_t)kTileM * kPoolStride + 16, "pooling tile + zero slot must fit");
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev]) {
    NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<DT, TN, KPIPE, X3, WALK, X2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev] = true;
  }
  const int groups = (p.m_tiles + 7) / 8;
  const unsigned n_blocks = (unsigned)(groups * 8 * p.n_tiles);
  dim3 grid(WALK ? std::min(n_blocks, p.walk > 1 ? (unsigned)p.walk : kWalkGrid) : n_blocks), block(kThreads);
  hipLaunchKernelGGL((conv_igemm_kernel<DT, TN, KPIPE, X3, WALK, X2>), grid, block, lds, stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

// p.walk picks the walking instantiation (a separate kernel: kernels.h, ConvParams::walk)
template <int DT, int TN, bool KPIPE, bool X3>
int launch_one(const ConvParams& p, hipStream_t stream) {
  return p.walk ? launch_one_w<DT, TN, KPIPE, X3, true>(p, stream) : launch_one_w<DT, TN, KPIPE, X3, false>(p, stream);
}

template <int DT>
int launch_dt(const ConvParams& p, int TN, hipStream_t stream) {
  const bool kpipe = (p.n_taps == 1);
  if constexpr (DT != NESTI_F32) {
    if (p.x2) {          // plain activations x pair-packed weights (kernels.h: ConvParams::x2): one-tap layers only; never a walking launch
      if (!kpipe || p.x3native || p.walk) NESTI_FAIL("launch_conv: x2 is the filter pass's 1x1x1 / FC variant");
      if (TN == 128) return launch_one_w<DT, 128, true, false, false, true>(p, stream);
      if (TN == 64) return launch_one_w<DT, 64, true, false, false, true>(p, stream);
      NESTI_FAIL("launch_conv: unsupported N tile");
    }
    if (p.x3native) {
      if (TN == 128) return kpipe ? launch_one<DT, 128, true, true>(p, stream) : launch_one<DT, 128, false, true>(p, stream);
      if (TN == 64) return kpipe ? launch_one<DT, 64, true, true>(p, stream) : launch_one<DT, 64, false, true>(p, stream);
      NESTI_FAIL("launch_conv: unsupported N tile");
    }
  }
  if (p.x3native || p.x2) NESTI_FAIL("launch_conv: the pair / exact-weight K loops are for the 16-bit kernels");
  if (TN == 128) return kpipe ? launch_one<DT, 128, true, false>(p, stream) : launch_one<DT, 128, false, false>(p, stream);
  if (TN == 64) return kpipe ? launch_one<DT, 64, true, false>(p, stream) : launch_one<DT, 64, false, false>(p, stream);
  NESTI_FAIL("launch_conv: unsupported N tile");
}

}  // namespace

int launch_conv(const ConvParams& p, int dtype, int TN, hipStream_t stream) {
  if (p.m_tiles <= 0 || p.n_tiles <= 0) return 0;
  if (p.pool_k > 1 && p.n_taps != 1) NESTI_FAIL("launch_conv: fused pooling needs a 1x1x1 layer");
  if (p.mp_mode != 0 && (p.log2S < 1 || !p.mp_out)) NESTI_FAIL("launch_conv: fused max-pool needs a volume >= 2^3 and an output");
  if (p.s_real && !(p.s_real == 3 && p.log2S == 2)) NESTI_FAIL("launch_conv: s_real is the 3^3 grid inside a 4^3 index space only");
  if (p.s_real && p.pool_k == 3) NESTI_FAIL("launch_conv: fused 3^3 avg-pool is not built for the embedded 3^3 volume");
  if (p.s_real && (p.mp_mode != 0 || p.mp_mode2 != 0)) NESTI_FAIL("launch_conv: the fused 2^3 max-pool does not apply to the 3^3 volume");
  if (p.mp_mode2 != 0 && (p.pool_k <= 1 || !p.mp_out || p.log2S < 1)) NESTI_FAIL("launch_conv: mp_mode2 belongs to a fused avg-pool half with a pooled output");
  if (p.pool_k > 1 && !((p.log2S == 3 && p.pool_k == 3) || (p.log2S == 2 && (p.pool_k == 2 || p.pool_k == 3)) ||
                        (p.log2S == 1 && p.pool_k == 2)))
    NESTI_FAIL("launch_conv: fused pooling supports (S,k) in {(8,3),(4,3),(4,2),(2,2)}");
  if (dtype == NESTI_BF16) return launch_dt<NESTI_BF16>(p, TN, stream);
  if (dtype == NESTI_F16) return launch_dt<NESTI_F16>(p, TN, stream);
  if (dtype == NESTI_F32) return launch_dt<NESTI_F32>(p, TN, stream);
  NESTI_FAIL("launch_conv: unsupported dtype");
}

}  // namespace nesti
