// k^3-tap conv3d on the 8^3 volume (k = 3, 5), FOUR points per workgroup: the dominant layers of the network
// (the 5^3 taps at 8^3 are ~2/3 of all multiply-accumulates of a top-1 forward pass).
//
// Same arithmetic as conv_igemm_kernel (conv.hip): tf.nn.conv3d 'SAME' + bias_add + inference batch-norm (folded on
// the host) + ReLU (utils/tf_util.py:298-311, 491-494), optionally followed by the block's 2^3 / 2 max-pool
// (utils/tf_util.py:424-428, e.g. models/experts_n_est.py:198) fused into the epilogue.
//
// Why another kernel.  conv_igemm_kernel's time follows the number of MFMAs it issues and the weight bytes it streams
// (profiles/r01_mfma_ubench.txt).  With one point per workgroup a 32-row MFMA tile spans several y or z values, so
// a padding tap can only skip it when ALL of them leave the volume (issued / nominal 0.81 for 5^3, against 0.61
// useful), and every workgroup re-streams every tap's weight tile for its one point.  Here
//   * the M tile is 4 points x 512 voxels = 2048 rows, and an MFMA tile is the x-line (y, z) of the 4 points
//     (32 rows = 4 points x 8 x): a tap (dz, dy, dx) skips the tile exactly when y + dy or z + dz leaves the volume,
//     at single-voxel granularity on both axes (issued / nominal 0.72 for 5^3, 0.84 for 3^3), and the skip is a
//     scalar test -- all 32 rows of a tile share (y, z);
//   * the N tile is 32 output channels, the K chunk 64 bytes per row (32 x 16-bit or 16 x f32 channels), so the
//     4 points' chunk (128 KiB) stays resident in LDS for all k^3 taps and a tap's weight tile is 2 KiB for 4
//     points instead of 16 KiB for one: a quarter of the weight stream per output;
//   * the only per-lane padding test left is x + dx: an out-of-range lane reads an LDS address beyond the
//     allocation, which returns zeros on gfx950 (scripts/lds_oob_probe.hip), so no zero rows are kept.
//
// Tile -> wave map.  Tile (y, z) lives in LDS slot q = 8 * ((y + z) & 7) + z and belongs to wave (y + z) & 7: every
// wave owns exactly one tile of every z plane and one of every y plane (a Latin square), so whatever planes a tap
// kills, every wave -- hence every SIMD's matrix pipe -- loses the same number of tiles.  The tiles a wave reads for
// tap (dz, dy) are the 8 consecutive slots of wave (w + dy + dz) & 7 shifted by dz, so fragment addresses are one
// per-lane base (x + dx shift, swizzle) plus a wave-uniform base plus a compile-time j * 2048.
//
// Pipeline.  Weight tiles of one (dz, dy) row of taps (k tiles, 2 KiB each) stream L2 -> LDS by LDS-DMA two rows
// ahead into 3 slots, one barrier per row (k = 5); pairs of rows one pair ahead into 2 slots, one barrier per pair (k = 3).  Within a wave, the A fragments of tap t + 1 are read into the registers
// tile j's MFMAs of tap t have just consumed (a full tap of lookahead, 64 VGPRs), one s_waitcnt lgkmcnt(0) per tap.
//
// LDS: [0, 32 KiB) weight slots (3 x 10 KiB for k = 5, 2 x 12 KiB for k = 3), [32, 160 KiB) the input chunk; rows are 64 B with the 16-B slot
// XOR-swizzled by the point index (input) / (row >> 2) & 3 (weights), applied on the DMA source address, which makes
// every ds_read_b128 lane group conflict-free.  The epilogue reuses the whole 160 KiB as an fp32 staging tile.
//
// What did NOT pay (profiles/r02_conv8_experiments.txt): seven re-schedulings of this loop -- waves out of
// phase, counted vmcnt, hand-counted lgkmcnt with inline-asm reads, adjacent MFMA pairs, K-step-major order, weights
// straight from L2 into registers without a row barrier, barrier-free rows through LDS counters -- all within +-3 %
// or slower.  The kernel runs against the chip's power / clock limit: its time follows the MFMAs issued and the operand
// bits they toggle (zero data: +9 %), which is what the tile layout above reduces.
#include <type_traits>

#include "kernels.h"
#include "mma.h"

namespace nesti {
namespace {

constexpr int kThreads8 = 512;
constexpr int kPts = 4;                      // points per workgroup
constexpr int kTileBytes = 2048;             // 32 rows x 64 B
constexpr int kAOff = 32768;                 // input chunk at [32 KiB, 160 KiB)
constexpr int kLds8 = 163840;
constexpr unsigned kOob = 0x40000u;          // beyond any LDS allocation: ds_read returns 0
constexpr int kEpiStride = 144;              // bytes per row of the fp32 [1024][32] epilogue tile (+16 B pad)

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4_t* lds_u32x4_ptr;
// ds_read_b128 at an LDS byte address (a 32-bit value, so that an out-of-range address can be formed at all)
__device__ __forceinline__ uint4 lds128(unsigned addr) {
  const u32x4_t v = *(lds_u32x4_ptr)(size_t)addr;
  return make_uint4(v.x, v.y, v.z, v.w);
}

// X3 (pair modes, model.hip: PackedLayer::x3n): the K chunk is 16 channels -- an LDS row holds [hi k0..15 | lo k0..15] of the
// activation pair (staged from the two planes of the [hi 64 | lo 64] row groups) and a weight row [W_hi k0..15 | W_lo k0..15]
// -- and a tap multiplies hi * W_hi + lo * W_hi + hi * W_lo from ONE set of fragment reads: three MFMAs per 2 + 2/8
// ds_read_b128 instead of two (round 3, same box: -7 % on these layers against running the pair as three planes [hi | lo | hi]
// with weights [W_hi ; W_hi ; W_lo] through the plain kernel; the time of these kernels follows LDS bytes per MFMA).
template <int DT, int K, bool X3>
__global__ __launch_bounds__(kThreads8) void conv8_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;
  constexpr int LO = (K - 1) / 2;
  constexpr int NG = K * K;                  // (dz, dy) rows of taps per chunk
  // Weight slots: a slot holds R rows of taps and is recycled behind a workgroup barrier.  k = 5: one row (10 KiB) per
  // slot, three slots, streamed two rows ahead.  k = 3: a row is only 3 taps, so two rows (12 KiB) share a slot and a
  // barrier -- 5 barriers per chunk instead of 9 -- with two slots, streamed one pair ahead.
  constexpr int R = (K == 3) ? 2 : 1;
  constexpr int NS = (K == 3) ? 2 : 3;
  constexpr int AHEAD = NS - 1;
  constexpr int kSlot = R * K * kTileBytes;
  constexpr int NSR = (NG + R - 1) / R;      // slot fills per chunk
  static_assert(NS * kSlot <= kAOff, "weight slots must fit below the input chunk");
  static_assert((NG - K) % R == 0 && (NG - 1) % R == 0, "the early staging points must follow a barrier");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware block -> tile map (as conv_igemm_kernel): an M tile's N tiles stay on one XCD's L2
  const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
  const int n_tile = grp % p.n_tiles;
  const int m_tile = (grp / p.n_tiles) * 8 + xcd;
  if (m_tile >= p.m_tiles) return;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int p0 = m_tile * kPts;
  if (p0 >= npts) return;
  const int np_here = min(kPts, npts - p0);

  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in) +
                              ((size_t)p0 * 512 * p.in_cstride + p.in_coff) * kEsz;
  const unsigned char* w_tile = reinterpret_cast<const unsigned char*>(p.wpk) +
                                (size_t)n_tile * p.n_chunks * (K * K * K) * kTileBytes;

  // ---- A staging: wave w fills its own 8 tile slots; piece h of a tile = rows 16h .. 16h+15 (row = 8 pt + x) -----
  const int st_row = lane >> 2;                          // 0..15 within the piece
  unsigned a_voff[2];
  bool a_ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pt = 2 * h + (st_row >> 3), x = st_row & 7;
    const int kslot = (lane & 3) ^ pt;                   // inverse swizzle on the SOURCE (LDS-DMA writes lane-linear)
    a_voff[h] = (unsigned)((pt * 512 + x) * p.in_cstride * kEsz + (X3 ? (kslot & 1) * 16 + (kslot >> 1) * (2 * kSplitGroup) : kslot * 16));
    a_ok[h] = pt < np_here;
  }
  // tiles (bit j) of this wave whose slots are staged; split so that part of the next chunk can be prefetched while the
  // current chunk's last taps -- which no longer read those slots -- still run
  auto stage_a = [&](int c, unsigned tiles) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!(tiles & (1u << j))) continue;
      const int y = (wave - j) & 7;                      // tile (y, z = j) of this wave
      // X3: chunk c = channels [16 c, 16 c + 16) of 64-channel group c >> 2, whose two planes sit 128 B apart in a 256-B run
      const unsigned char* src = in_b + (size_t)((j * 64 + y * 8) * p.in_cstride) * kEsz +
                                 (X3 ? (size_t)(c >> 2) * (2 * kPairPlanes * kSplitGroup) + (size_t)(c & 3) * 32 : (size_t)c * 64);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (a_ok[h]) glds16(src + a_voff[h], lds0 + kAOff + (wave * 8 + j) * kTileBytes + h * 1024);
    }
  };
  auto stage_b = [&](int c, int sr, int slot) __attribute__((always_inline)) {   // rows sr * R .. of chunk c
    const unsigned char* src = w_tile + ((size_t)c * (K * K * K) + (size_t)sr * R * K) * kTileBytes;
    const int pieces = 2 * K * min(R, NG - sr * R);
    for (int pid = wave; pid < pieces; pid += 8)
      glds16(src + pid * 1024 + lane * 16, lds0 + slot * kSlot + pid * 1024);
  };
  auto b_slot = [&](int g) __attribute__((always_inline)) -> unsigned {   // LDS byte offset of row g's weight tiles
    return (unsigned)(((g / R) % NS) * kSlot + (g % R) * K * kTileBytes);
  };

  // ---- per-lane fragment coordinates ---------------------------------------------------------------------------
  const int l31 = lane & 31, khalf = lane >> 5;
  const int pt = l31 >> 3, rx = l31 & 7;
  // 16-B slot of K-step kk within the 64-B row: (2 kk + khalf) ^ key; the kk = 1 address is the kk = 0 address -+ 32
  const int a_sw = (khalf ^ pt) & 3, b_sw = (khalf ^ (l31 >> 2)) & 3;
  const unsigned a_lane = lds0 + kAOff + (unsigned)((pt * 8 + rx) * 64 + (a_sw << 4));
  const unsigned b_lane = lds0 + (unsigned)(l31 * 64 + (b_sw << 4));
  const unsigned a_d1 = (a_sw & 2) ? (unsigned)-32 : 32u, b_d1 = (b_sw & 2) ? (unsigned)-32 : 32u;

  // y-liveness of this wave's tiles per dy, packed 8 bits per dy (tile j has y = (wave - j) & 7)
  unsigned long long ymask_pack = 0ull;
#pragma unroll
  for (int d = 0; d < K; ++d) {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int y = (wave - j) & 7;
      if ((unsigned)(y + d - LO) < 8u) m |= 1u << j;
    }
    ymask_pack |= (unsigned long long)m << (8 * d);
  }
  // tiles (wave-uniform) that tap row g = (dz, dy) touches, and the LDS byte offset of the source tile run
  auto row_mask = [&](int g) __attribute__((always_inline)) -> unsigned {
    const int dz = g / K - LO, dy = g % K - LO;
    const unsigned zm = dz >= 0 ? (0xffu >> dz) : ((0xffu << (-dz)) & 0xffu);
    (void)dy;
    return zm & (unsigned)((ymask_pack >> (8 * (g % K))) & 0xffull);
  };
  auto row_base = [&](int g) __attribute__((always_inline)) -> int {
    const int dz = g / K - LO, dy = g % K - LO;
    return ((((wave + dy + dz) & 7) * 8) + dz) * kTileBytes;
  };

  // Taps run dz-major, dy next.  The last dz slab (dz = +LO) reads source planes z' >= LO only, its last row (dy = +LO)
  // source lines y' >= LO only: this wave's slots with z < LO are free from the start of that slab, those with y < LO
  // from the start of that row, and the next chunk is staged into them early.  The rest waits for the chunk boundary.
  unsigned pre_slab = 0, pre_row = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int y = (wave - j) & 7;
    if (j < LO) pre_slab |= 1u << j;
    else if (y < LO) pre_row |= 1u << j;
  }
  if (!(p.remap & 1)) pre_slab = pre_row = 0;          // A/B switch (NESTI_CONV8_FLAGS bit 0)
  const unsigned pre_none = 0xffu & ~(pre_slab | pre_row);

  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // b[u & 1] holds tap u's weight fragments, b[(u + 1) & 1] receives the next tap's; K is odd, so the last tap of a row
  // leaves the next row's first set in b[1]: it is moved to b[0] once per row
  uint4 a[8][2], b[2][2];

  for (int c = 0; c < p.n_chunks; ++c) {
    __syncthreads();                       // every wave is done with the previous chunk
    stage_a(c, c == 0 ? 0xffu : pre_none);
#pragma unroll
    for (int sr = 0; sr < AHEAD; ++sr) stage_b(c, sr, sr);
    wait_vm0();
    __syncthreads();
    // prologue: fragments of tap 0 (row 0, dx = -LO)
    {
      const unsigned m0 = row_mask(0);
      const bool okx = (unsigned)(rx - LO) < 8u;
      const unsigned nb0 = okx ? a_lane + (unsigned)(row_base(0) - LO * 64) : kOob;
      const unsigned nb1 = nb0 + a_d1;
      b[1][0] = lds128(b_lane);                         // tap 0 of every row starts by moving b[1] to b[0]
      b[1][1] = lds128(b_lane + b_d1);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (m0 & (1u << j)) {
          a[j][0] = lds128(nb0 + j * kTileBytes);
          a[j][1] = lds128(nb1 + j * kTileBytes);
        }
    }
    for (int g = 0; g < NG; ++g) {
      if (g % R == 0 && g / R + AHEAD < NSR) stage_b(c, g / R + AHEAD, (g / R + AHEAD) % NS);
      if (c + 1 < p.n_chunks) {            // early staging of the next chunk into slots this chunk no longer reads
        if (g == NG - K) stage_a(c + 1, pre_slab);
        if (g == NG - 1) stage_a(c + 1, pre_row);
      }
      const unsigned mask_g = row_mask(g);
      const unsigned mask_n = (g + 1 < NG) ? row_mask(g + 1) : 0u;
      const int base_g = row_base(g);
      const int base_n = (g + 1 < NG) ? row_base(g + 1) : 0;
      const unsigned bslot_g = b_slot(g);
      const unsigned bslot_n = b_slot(g + 1);
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const bool last_u = (u == K - 1);
        // the tile tests are re-derived from the integer masks in every tap (opaque to CSE): a condition kept alive across
        // taps is materialised as a lane mask and costs VALU work per branch
        unsigned m_mm = mask_g, m_rd = last_u ? mask_n : mask_g;   // tiles multiplied now / tiles whose next-tap fragments are read
        asm volatile("" : "+s"(m_mm), "+s"(m_rd));
        const int dxn = last_u ? -LO : u + 1 - LO;
        const bool okx = (unsigned)(rx + dxn) < 8u;
        const unsigned nb0 = okx ? a_lane + (unsigned)((last_u ? base_n : base_g) + dxn * 64) : kOob;
        const unsigned nb1 = nb0 + a_d1;
        const unsigned bsrc = b_lane + (last_u ? bslot_n : bslot_g + (unsigned)((u + 1) * kTileBytes));
        // ---- top of tap: everything read during the previous tap has landed -------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0), vmcnt / expcnt untouched
        __builtin_amdgcn_sched_barrier(0);
        if (R > 1 && last_u && (g % R == R - 1 || g == NG - 1)) {
          // Shared slots (k = 3): the next pair's weights were streamed ONE fill ahead, so they must be confirmed before
          // this tap prefetches their first fragments; and every wave's last reads of the current slot (issued during the
          // previous tap) have returned at this point, so the barrier also frees it for the fill the next row starts.
          wait_vm0();
          __builtin_amdgcn_s_barrier();
        }
        if (u == 0) {                                    // K is odd: the previous row (or the prologue) left tap 0's set in b[1]
          b[0][0] = b[1][0];
          b[0][1] = b[1][1];
        }
        uint4(&bc)[2] = b[u & 1];
        uint4(&bn)[2] = b[(u + 1) & 1];
        if (m_rd) {                                      // there is a next tap in this chunk
          bn[0] = lds128(bsrc);
          bn[1] = lds128(bsrc + b_d1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (!last_u) {
            if (__builtin_expect((m_mm & (1u << j)) != 0, 1)) {
              mma<DT>(acc[j], a[j][0], bc[0]);
              mma<DT>(acc[j], a[j][1], X3 ? bc[0] : bc[1]);
              if (X3) mma<DT>(acc[j], a[j][0], bc[1]);
              a[j][0] = lds128(nb0 + j * kTileBytes);
              a[j][1] = lds128(nb1 + j * kTileBytes);
            }
          } else {
            if (__builtin_expect((m_mm & (1u << j)) != 0, 1)) {
              mma<DT>(acc[j], a[j][0], bc[0]);
              mma<DT>(acc[j], a[j][1], X3 ? bc[0] : bc[1]);
              if (X3) mma<DT>(acc[j], a[j][0], bc[1]);
            }
            if (__builtin_expect((m_rd & (1u << j)) != 0, 1)) {
              a[j][0] = lds128(nb0 + j * kTileBytes);
              a[j][1] = lds128(nb1 + j * kTileBytes);
            }
          }
        }
      }
      static_assert((K & 1) == 1, "the weight-fragment ping-pong assumes an odd number of taps per row");
      if (R == 1) {
        wait_vm0();                                      // the next row's weight DMA (issued two rows ago) has landed
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __syncthreads();                                       // nobody still reads A / B: the LDS becomes the epilogue tile

  // ---- epilogue: bias + ReLU in fp32 through an LDS tile, two passes over z (0..3, 4..7) ---------------------------
  const int out_esz = p.out_f32 ? 4 : kEsz;
  const float bv = p.bias[n_tile * 32 + l31];
  const float act_floor = p.relu ? 0.f : -INFINITY;
  const int out_col0 = p.out_coff + n_tile * 32;
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
  auto epi_pass = [&](auto HH) __attribute__((always_inline)) {
    constexpr int hh = decltype(HH)::value;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * hh + jj;                         // z = j, y = (wave - j) & 7
      const int y = (wave - j) & 7;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * khalf;      // MFMA row = 8 pt + x
        const int row = ((((m >> 3) * 4 + jj) * 8 + y) << 3) + (m & 7);
        *reinterpret_cast<float*>(smem + row * kEpiStride + l31 * 4) = fmaxf(fmaf(acc[j][r], p.acc_scale, bv), act_floor);
      }
    }
    __syncthreads();
    if (p.mp_mode != 1) {                                // full-resolution rows: 4 lanes x 8 channels = one 32-channel row segment
#pragma unroll 2
      for (int it = 0; it < 8; ++it) {
        const int item = it * kThreads8 + tid;
        const int row = item >> 2, seg = item & 3;
        const int ptl = row >> 8, vox = (4 * hh) * 64 + (row & 255);
        if (ptl < np_here) {
          const float4 f0 = *reinterpret_cast<const float4*>(smem + row * kEpiStride + seg * 32);
          const float4 f1 = *reinterpret_cast<const float4*>(smem + row * kEpiStride + seg * 32 + 16);
          cvt_store8(out_b, ((long long)(p0 + ptl) * 512 + vox) * p.out_cstride, out_col0 + seg * 8, f0, f1);
        }
      }
    }
    if (p.mp_mode != 0) {                                // fused 2^3 / 2 max-pool of the activated values: 4 pts x 2 x 4 x 4 cells
      const int item = tid;                              // 128 cells x 4 segments of 8 channels
      const int cell = item >> 2, seg = item & 3;
      const int ptl = cell >> 5, cz = (cell >> 4) & 1, cy = (cell >> 2) & 3, cx = cell & 3;
      float4 m0 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), m1 = m0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = ((ptl * 4 + 2 * cz + (q >> 2)) * 8 + 2 * cy + ((q >> 1) & 1)) * 8 + 2 * cx + (q & 1);
        const float4 f0 = *reinterpret_cast<const float4*>(smem + row * kEpiStride + seg * 32);
        const float4 f1 = *reinterpret_cast<const float4*>(smem + row * kEpiStride + seg * 32 + 16);
        m0.x = fmaxf(m0.x, f0.x); m0.y = fmaxf(m0.y, f0.y); m0.z = fmaxf(m0.z, f0.z); m0.w = fmaxf(m0.w, f0.w);
        m1.x = fmaxf(m1.x, f1.x); m1.y = fmaxf(m1.y, f1.y); m1.z = fmaxf(m1.z, f1.z); m1.w = fmaxf(m1.w, f1.w);
      }
      if (ptl < np_here) {
        const int ovox = ((2 * hh + cz) * 4 + cy) * 4 + cx;
        cvt_store8(mp_b, ((long long)(p0 + ptl) * 64 + ovox) * p.mp_cstride, out_col0 + seg * 8, m0, m1);
      }
    }
    __syncthreads();
  };
  epi_pass(std::integral_constant<int, 0>{});
  epi_pass(std::integral_constant<int, 1>{});
}

template <int DT, int K, bool X3>
int launch_conv8_one(const ConvParams& p, hipStream_t stream) {
  constexpr int kMaxDevices = 64;
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  NESTI_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev]) {
    NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8_kernel<DT, K, X3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLds8));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev] = true;
  }
  const int groups = (p.m_tiles + 7) / 8;
  dim3 grid((unsigned)(groups * 8 * p.n_tiles)), block(kThreads8);
  hipLaunchKernelGGL((conv8_kernel<DT, K, X3>), grid, block, kLds8, stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int DT>
int launch_conv8_dt(const ConvParams& p, int k, hipStream_t stream) {
  if constexpr (DT != NESTI_F32) {
    if (p.x3native && k == 5) return launch_conv8_one<DT, 5, true>(p, stream);
    if (p.x3native && k == 3) return launch_conv8_one<DT, 3, true>(p, stream);
  }
  if (p.x3native) NESTI_FAIL("launch_conv8: the pair-native K loop is for the 16-bit kernels");
  if (k == 5) return launch_conv8_one<DT, 5, false>(p, stream);
  if (k == 3) return launch_conv8_one<DT, 3, false>(p, stream);
  NESTI_FAIL("launch_conv8: kernel size must be 3 or 5");
}

}  // namespace

// p.m_tiles = groups of 4 points, p.n_tiles = 32-column tiles, p.n_chunks = 64-byte K chunks, p.n_taps = k^3
int launch_conv8(const ConvParams& p, int dtype, int k, hipStream_t stream) {
  if (p.m_tiles <= 0 || p.n_tiles <= 0) return 0;
  if (p.log2S != 3 || p.s_real) NESTI_FAIL("launch_conv8: the 8^3 volume only");
  if (p.n_taps != k * k * k) NESTI_FAIL("launch_conv8: all k^3 taps must be present");
  if (p.point_index) NESTI_FAIL("launch_conv8: no input gather (k^3 layers never read the routed MuPS tensor)");
  if (p.pool_k > 1 || p.split_tile != p.n_tiles) NESTI_FAIL("launch_conv8: no fused avg-pool / merged layers");
  if (p.mp_mode == 2) NESTI_FAIL("launch_conv8: max-pool mode 2 is conv1's (a 1x1x1 layer)");
  if (p.mp_mode != 0 && !p.mp_out) NESTI_FAIL("launch_conv8: fused max-pool needs an output");
  if (dtype == NESTI_BF16) return launch_conv8_dt<NESTI_BF16>(p, k, stream);
  if (dtype == NESTI_F16) return launch_conv8_dt<NESTI_F16>(p, k, stream);
  if (dtype == NESTI_F32) return launch_conv8_dt<NESTI_F32>(p, k, stream);
  NESTI_FAIL("launch_conv8: unsupported dtype");
}

}  // namespace nesti
