// k^3-tap conv3d on the 4^3 volume (k = 2 .. 5): conv2 / conv3 of the inception blocks after the first max-pool
// (models/experts_n_est.py:198-205, 265-272; the 4^3 blocks of the ablation towers, models/ss_norm_est.py:60-75).
//
// Same arithmetic as conv_igemm_kernel (conv.hip): tf.nn.conv3d 'SAME' + bias_add + inference batch-norm (folded on the host)
// + ReLU (utils/tf_util.py:298-311, 491-494), optionally followed by the block's 2^3 / 2 max-pool (utils/tf_util.py:424-428)
// fused into the epilogue.
//
// Why another kernel.  At 4^3 a tap (dz, dy, dx) of a 4-wide kernel lands inside the volume for only 42 % of the output voxels
// (k = 4: ((3 + 4 + 3 + 2) / 16)^3), and conv_igemm_kernel -- whose best 32-row tile there is the x-line (y, z) of 8 points -- can
// skip a tile only when y + dy or z + dz leaves the volume: it issues 56 % of the nominal MFMAs, a third more than useful,
// behind a barrier every two taps.  Here
//   * an MFMA tile is ONE voxel of SIXTEEN points (v_mfma_f32_16x16x32: 16 rows x 16 columns x 32 channels), so padding is a
//     whole-tile property in all three axes and the kernel issues exactly the useful MFMAs (issued / nominal 0.42 for k = 4,
//     0.67 for k = 2);
//   * a wave owns TWO x-lines (y, z) of the volume -- 8 voxel tiles x 4 column tiles, 128 accumulator registers -- and works
//     through the taps one (dz, dy) ROW at a time: the four source tiles of line (y + dy, z + dz) are read from LDS once per row
//     and serve all k taps of the row (tile x uses source x + dx: the x padding is resolved at compile time, no test, no
//     wasted MFMA), a dead line (y + dy or z + dz outside) is skipped by one scalar branch per tap;
//   * workgroup = 16 points x 64 voxels x 64 output columns; the K chunk is 64 bytes per row (32 x 16-bit or 16 x f32 channels),
//     so the 16 points' chunk (64 KiB) stays resident in LDS for all k^3 taps; per row a wave reads 8 A + 4 k B fragments for up
//     to 24 k MFMAs;
//   * the layer is ONE stream of n_chunks * k^2 tap rows: the weight tiles of a row stream L2 -> LDS by LDS-DMA into two slots
//     (the fill of row g + 2 is issued at the row's only barrier, at the top of row g's last tap), and for k <= 4 the NEXT chunk's
//     input streams into a second input buffer during the first rows of the current chunk, confirmed by counted s_waitcnt
//     vmcnt -- no chunk boundary is waited for (timing-only bound for hiding the input staging: -6.4 / -6.9 %; got -4 / -5 %).
//
// Tile -> wave map.  Line (y, z) lives in LDS slots 4 L .. 4 L + 3 (x = 0 .. 3) with L = 4 ((y + z) & 3) + z: class
// c = (y + z) & 3 holds one line of every y and of every z -- a Latin square -- so a tap row kills (nearly) the same number of
// lines in every class.  Waves c and c + 4 (one SIMD) own class c, wave (c, h) its lines z = 2 h and 2 h + 1, and the source
// lines of tap row (dz, dy) are the slots 4 ((c + dy + dz) & 3) + 2 h + dz (+ 1): one wave-uniform base per row plus compile-time
// offsets.
//
// LDS: two weight slots of k taps x 4 KiB, then one (k = 5) or two (k <= 4: 32 + 128 KiB for k = 4) input buffers of 64 KiB; rows
// are 64 B with the 16-B slot XOR-swizzled by {0, 2, 3, 1}[(row >> 2) & 3], applied on the DMA source address, which makes every
// ds_read_b128 lane group of a 16-row fragment read conflict-free.  The epilogue reuses the LDS as an fp32 staging tile (two
// passes of 32 columns).
//
// X3 (pair modes, model.hip: PackedLayer::x3n): the K chunk is 16 channels -- an LDS row holds [hi k0..15 | lo k0..15] and a
// weight row [W_hi k0..15 | W_lo k0..15]; the K = 32 MFMA multiplies [hi | lo] x [W_hi ; W_hi] (= hi W_hi + lo W_hi) and
// [hi | lo] x [W_lo ; 0] (= hi W_lo; the zero half comes from an out-of-range LDS address, which reads as zeros on gfx950).
#include <type_traits>

#include "kernels.h"
#include "mma.h"

namespace nesti {
namespace {

constexpr int kThreads4 = 512;
constexpr int kPts4 = 16;                    // points per workgroup
constexpr int kTile4 = 1024;                 // 16 rows x 64 B
constexpr int kBTap4 = 4 * kTile4;           // one tap's weights: 64 columns x 64 B
constexpr int kABuf4 = 64 * kTile4;          // one input chunk: 64 voxel tiles = 64 KiB
constexpr int kEpiStride4 = 144;             // bytes per row of the fp32 [1024][32] epilogue tile (+16 B pad)
constexpr int kEpi4 = 1024 * kEpiStride4;    // 144 KiB
// LDS of the instantiation for kernel edge K: two weight slots of one tap row each (K taps x 4 KiB), then the input chunk --
// double-buffered where that fits 160 KiB (K <= 4: 32 + 128 KiB for K = 4), single otherwise (K = 5: 40 + 64 KiB)
template <int K> constexpr bool dba4() { return 2 * K * kBTap4 + 2 * kABuf4 <= 163840; }
template <int K> constexpr int lds4() {
  constexpr int loop = 2 * K * kBTap4 + (dba4<K>() ? 2 : 1) * kABuf4;
  return loop > kEpi4 ? loop : kEpi4;
}
constexpr unsigned kOob4 = 0x40000u;         // beyond any LDS allocation: ds_read returns 0

typedef unsigned u32x4q_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4q_t* lds_u32x4q_ptr;
__device__ __forceinline__ uint4 lds128q(unsigned addr) {
  const u32x4q_t v = *(lds_u32x4q_ptr)(size_t)addr;
  return make_uint4(v.x, v.y, v.z, v.w);
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int DT> __device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<NESTI_BF16>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<NESTI_F16>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<NESTI_F32>(f32x4& acc, const uint4& a, const uint4& b) {
  // exact fp32: lane (row, kb) holds channels 4 kb .. 4 kb + 3 of the 16-channel chunk; A and B use the same order
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// 16-B slot swizzle key of row r of a 16-row tile (see the header): with it the four row quads a ds_read_b128 lane group
// touches land in four different 16-B columns of the 64-B row
__device__ __forceinline__ int swz4(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }

// K = kernel edge: k^3 taps, one weight slot / one barrier per (dz, dy) row of K taps
template <int DT, int K, bool X3>
__device__ __forceinline__ void conv4n_tile(const ConvParams& p, const unsigned bid, const int tid_in) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;
  constexpr int NB = X3 ? 2 : 1;             // weight fragment sets per column tile (pair modes: [W_hi ; W_hi] and [W_lo ; 0])
  constexpr int LO = (K - 1) / 2;
  constexpr int NR = K * K;                  // tap rows per chunk
  // a row's source tiles are read one row ahead into a second register set where the registers allow it (hipcc spills the
  // other instantiations; without it the reads' latency is exposed once per row, behind the row barrier)
  constexpr bool PREF = !X3 && K == 4;
  static_assert(K >= 2 && K <= 5, "kernel edge");
  constexpr bool DBA = dba4<K>();            // the next chunk's input streams into a second buffer while this one is multiplied
  constexpr int kSlotB = K * kBTap4;         // one weight slot = the K taps of one (dz, dy) row
  constexpr int kAOff = 2 * kSlotB;
  constexpr int APR = NR >= 6 ? 4 : 2;       // rows over which the next chunk's 8 staging instructions per wave are spread
  constexpr int APP = 8 / APR;               // ... and how many of them a wave issues per row
  static_assert(lds4<K>() <= 163840, "LDS budget");

  const int tid = tid_in, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3, h = wave >> 2;    // Latin class and z half of this wave's two lines

  // XCD-aware block -> tile map: the column tiles of a group of 16 points stay on one XCD's L2
  const int xcd = bid & 7, grp = bid >> 3;
  const int n_tile = grp % p.n_tiles;
  const int m_tile = (grp / p.n_tiles) * 8 + xcd;
  if (m_tile >= p.m_tiles) return;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int p0 = m_tile * kPts4;
  if (p0 >= npts) return;
  const int np_here = min(kPts4, npts - p0);

  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in) + ((size_t)p0 * 64 * p.in_cstride + p.in_coff) * kEsz;
  const unsigned char* w_tile = reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)n_tile * p.n_chunks * (K * K * K) * kBTap4;

  // ---- A staging: wave w fills slots 8 w .. 8 w + 7, one LDS-DMA instruction (64 lanes x 16 B = 16 rows x 64 B) per tile ----
  const int st_row = lane >> 2;                                    // point within the group
  const int st_slot = (lane & 3) ^ swz4(st_row);                   // inverse swizzle on the SOURCE (LDS-DMA writes lane-linear)
  const bool st_ok = st_row < np_here;
  const size_t st_off = (size_t)st_row * 64 * p.in_cstride * kEsz +
                        (X3 ? (size_t)((st_slot & 1) * 16 + (st_slot >> 1) * (2 * kSplitGroup)) : (size_t)st_slot * 16);
  // tiles q0 .. q0 + nq - 1 of this wave's eight, chunk c, into input buffer `buf`
  auto stage_a = [&](int c, int buf, int q0, int nq) __attribute__((always_inline)) {
    const size_t coff = X3 ? (size_t)(c >> 2) * (2 * kPairPlanes * kSplitGroup) + (size_t)(c & 3) * 32
                           : p.in_pair ? (size_t)(c >> 1) * (2 * kPairPlanes * kSplitGroup) + (size_t)(c & 1) * 64 : (size_t)c * 64;
#pragma unroll
    for (int qq = 0; qq < nq; ++qq) {
      const int s = wave * 8 + q0 + qq;                            // slot -> voxel: x = s & 3, line L = s >> 2 = 4 class + z
      const int x = s & 3, z = (s >> 2) & 3, y = ((s >> 4) - z) & 3;
      const int v = 16 * z + 4 * y + x;
      if (st_ok) glds16(in_b + (size_t)v * p.in_cstride * kEsz + coff + st_off, lds0 + kAOff + buf * kABuf4 + s * kTile4);
    }
  };
  // tap row g of the layer's stream of rows (g = chunk * K^2 + row: the packed weights are contiguous in g) into weight slot g & 1
  auto stage_b = [&](int g) __attribute__((always_inline)) {
    const unsigned char* src = w_tile + (size_t)g * kSlotB;
    for (int pid = wave; pid < 4 * K; pid += 8) glds16(src + pid * 1024 + lane * 16, lds0 + (g & 1) * kSlotB + pid * 1024);
  };

  // ---- per-lane fragment coordinates ------------------------------------------------------------------------------
  const int r16 = lane & 15, kb = lane >> 4;
  const unsigned frag = (unsigned)(r16 * 64 + ((kb ^ swz4(r16)) << 4));
  const unsigned a_lane = lds0 + kAOff + frag;
  // weight fragments: plain = the lane's own K block; pair modes: set 0 reads W_hi for both halves of K ([W_hi ; W_hi]),
  // set 1 reads W_lo for the hi half and zeros (an out-of-range address) for the lo half
  unsigned b_lane[NB];
  if (X3) {
    b_lane[0] = lds0 + (unsigned)(r16 * 64 + (((kb & 1) ^ swz4(r16)) << 4));
    b_lane[NB - 1] = kb < 2 ? lds0 + (unsigned)(r16 * 64 + (((2 + (kb & 1)) ^ swz4(r16)) << 4)) : kOob4;
  } else {
    b_lane[0] = lds0 + frag;
  }

  // liveness of this wave's two lines j: z_j = 2 h + j, y_j = (cw - z_j) & 3; per shift d in [-2, 2] two bits per axis
  unsigned ymk = 0u, zmk = 0u;
#pragma unroll
  for (int d = 0; d < 5; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int z = 2 * h + j, y = (cw - z) & 3;
      if ((unsigned)(y + d - 2) < 4u) ymk |= 1u << (2 * d + j);
      if ((unsigned)(z + d - 2) < 4u) zmk |= 1u << (2 * d + j);
    }
  // tap row (iz, iy) -> (live-line bits, per-lane LDS address of the row's first source tile: line 0, x' = 0)
  auto row_info = [&](int iz, int iy, int buf, unsigned& live, unsigned& abase) __attribute__((always_inline)) {
    const int dz = iz - LO, dy = iy - LO;
    live = (zmk >> (2 * (dz + 2))) & (ymk >> (2 * (dy + 2))) & 3u;
    const int tile0 = 4 * (4 * ((cw + dy + dz) & 3) + 2 * h + dz);            // may be negative: only dead lines go out of range
    abase = a_lane + (unsigned)(buf * kABuf4 + tile0 * kTile4);
  };

  f32x4 acc[2][4][4];                        // [line][x][column tile]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][x][n][r] = 0.f;

  uint4 a[2][4], an[2][4], b[2][4][NB];
  auto load_a = [&](uint4 (&dst)[2][4], unsigned base) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int x = 0; x < 4; ++x) dst[j][x] = lds128q(base + (4 * j + x) * kTile4);
  };
  auto load_b = [&](uint4 (&dst)[4][NB], unsigned off) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int s = 0; s < NB; ++s) dst[n][s] = lds128q(b_lane[s] + off + n * kTile4);
  };

  // The layer is ONE stream of G = n_chunks * K^2 tap rows.  Row g multiplies out of weight slot g & 1 and input buffer
  // (chunk & 1); per row there is one barrier, at the top of its last tap: by then every wave holds that tap's weight fragments
  // in registers, so the row's slot is free and the fill of row g + 2 is issued into it, and the fill of row g + 1 (issued a row
  // earlier) is confirmed before the last tap prefetches row g + 1's first fragments.  The next chunk's input is staged into the
  // other buffer during the first APR rows of a chunk, APP LDS-DMA instructions per wave and row, and confirmed a row later by
  // the same counted s_waitcnt (LDS-DMA completes in order): no chunk boundary is ever waited for.
  const int G = p.n_chunks * NR;
  stage_a(0, 0, 0, 8);
  stage_b(0);
  stage_b(1);
  wait_vm0();
  __syncthreads();
  unsigned live_c, abase_c;
  row_info(0, 0, 0, live_c, abase_c);
  load_b(b[K & 1], 0u);                      // tap u of a row uses b[u & 1]; an odd row leaves the next row's first set in b[1]
  if (PREF) load_a(an, abase_c);
  int c = 0, row = 0, iz = 0, iy = 0;        // chunk, row within the chunk and its (dz, dy) index
  for (int g = 0; g < G; ++g) {
    // the row after this one: next (dz, dy) of this chunk, or row 0 of the next chunk in the other input buffer
    int cn = c, rown = row + 1, izn = iz, iyn = iy + 1;
    if (iyn == K) { iyn = 0; ++izn; }
    if (rown == NR) { rown = 0; izn = 0; iyn = 0; ++cn; }
    unsigned live_n = 0u, abase_n = abase_c;
    if (g + 1 < G) row_info(izn, iyn, DBA ? (cn & 1) : 0, live_n, abase_n);
    const bool stage_now = DBA && row < APR && c + 1 < p.n_chunks;
    if (stage_now) stage_a(c + 1, (c + 1) & 1, row * APP, APP);
    if (!PREF) {
      load_a(a, abase_c);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int x = 0; x < 4; ++x) a[j][x] = an[j][x];
      load_a(an, abase_n);                   // (the last row re-reads its own tiles: harmless)
    }
    const unsigned lv = live_c;
    const unsigned slot_off = (unsigned)((g & 1) * kSlotB), nslot_off = (unsigned)(((g + 1) & 1) * kSlotB);
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const bool last = (u == K - 1);
      if ((K & 1) && u == 0) {               // moved to b[0] once per row
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int s = 0; s < NB; ++s) b[0][n][s] = b[1][n][s];
      }
      uint4(&bc)[4][NB] = b[u & 1];
      uint4(&bn)[4][NB] = b[(u + 1) & 1];
      if (last) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the row's weight slot (and of its input tiles) are done
        if (stage_now) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APP) : "memory");   // row g + 1's weights have landed; the APP
        else wait_vm0();                                                            // input pieces issued this row may still fly
        __builtin_amdgcn_s_barrier();
        if (g + 2 < G) stage_b(g + 2);
        if (!DBA && rown == 0 && g + 1 < G) {
          // single input buffer (K = 5): the chunk is dead once every wave has passed this barrier (its tiles were read at the
          // row start); the next one is staged under this last tap's MFMAs and confirmed before the next row reads it
          stage_a(cn, 0, 0, 8);
        }
      }
      // fragment reads are unconditional (a dead line's reads land anywhere, its MFMAs are skipped): the number of reads in
      // flight is static, so the compiler places exact s_waitcnt lgkmcnt(N)
      load_b(bn, last ? nslot_off : slot_off + (unsigned)((u + 1) * kBTap4));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (__builtin_expect((lv & (1u << j)) != 0, 1)) {
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int xs = x + u - LO;       // source voxel x' = x + dx: compile-time padding test
            if (xs >= 0 && xs < 4) {
#pragma unroll
              for (int s = 0; s < NB; ++s)
#pragma unroll
                for (int n = 0; n < 4; ++n) mma16<DT>(acc[j][x][n], a[j][xs], bc[n][s]);
            }
          }
        }
      }
    }
    if (!DBA && rown == 0 && g + 1 < G) {    // K = 5: the freshly staged chunk (and the fill issued with it) before anyone reads it
      wait_vm0();
      __builtin_amdgcn_s_barrier();
    }
    live_c = live_n;
    abase_c = abase_n;
    c = cn; row = rown; iz = izn; iy = iyn;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __syncthreads();                           // nobody still reads A / B: the LDS becomes the epilogue tile

  // ---- epilogue: bias + ReLU in fp32 through an LDS tile [16 pts x 64 voxels][32], one pass per 32 columns ------------------
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
  auto epi_pass = [&](auto PP) __attribute__((always_inline)) {
    constexpr int pass = decltype(PP)::value;
    const int out_col0 = p.out_coff + n_tile * 64 + pass * 32;
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      const int n = 2 * pass + n2;
      const float bv = p.bias[n_tile * 64 + n * 16 + r16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int z = 2 * h + j, y = (cw - z) & 3;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int v = 16 * z + 4 * y + x;
#pragma unroll
          for (int r = 0; r < 4; ++r) {        // D: column = lane & 15, row (point) = 4 (lane >> 4) + r
            const int row = (4 * kb + r) * 64 + v;
            *reinterpret_cast<float*>(smem + row * kEpiStride4 + (n2 * 16 + r16) * 4) = fmaxf(fmaf(acc[j][x][n][r], p.acc_scale, bv), act_floor);
          }
        }
      }
    }
    __syncthreads();
    if (p.mp_mode != 1) {                    // full-resolution rows: 4 lanes x 8 channels = one 32-channel row segment
#pragma unroll 2
      for (int it = 0; it < 8; ++it) {
        const int item = it * kThreads4 + tid;
        const int row = item >> 2, seg = item & 3;
        const int ptl = row >> 6, vox = row & 63;
        if (ptl < np_here) {
          const float4 f0 = *reinterpret_cast<const float4*>(smem + row * kEpiStride4 + seg * 32);
          const float4 f1 = *reinterpret_cast<const float4*>(smem + row * kEpiStride4 + seg * 32 + 16);
          cvt_store8(out_b, ((long long)(p0 + ptl) * 64 + vox) * p.out_cstride, out_col0 + seg * 8, f0, f1);
        }
      }
    }
    if (p.mp_mode != 0) {                    // fused 2^3 / 2 max-pool of the activated values: 16 pts x 8 cells x 4 segments
      const int cell = tid >> 2, seg = tid & 3;
      const int ptl = cell >> 3, cz = (cell >> 2) & 1, cy = (cell >> 1) & 1, cx = cell & 1;
      float4 m0 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), m1 = m0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = ptl * 64 + (2 * cz + (q >> 2)) * 16 + (2 * cy + ((q >> 1) & 1)) * 4 + 2 * cx + (q & 1);
        const float4 f0 = *reinterpret_cast<const float4*>(smem + row * kEpiStride4 + seg * 32);
        const float4 f1 = *reinterpret_cast<const float4*>(smem + row * kEpiStride4 + seg * 32 + 16);
        m0.x = fmaxf(m0.x, f0.x); m0.y = fmaxf(m0.y, f0.y); m0.z = fmaxf(m0.z, f0.z); m0.w = fmaxf(m0.w, f0.w);
        m1.x = fmaxf(m1.x, f1.x); m1.y = fmaxf(m1.y, f1.y); m1.z = fmaxf(m1.z, f1.z); m1.w = fmaxf(m1.w, f1.w);
      }
      if (ptl < np_here)
        cvt_store8(mp_b, ((long long)(p0 + ptl) * 8 + (cz * 4 + cy * 2 + cx)) * p.mp_cstride, out_col0 + seg * 8, m0, m1);
    }
    __syncthreads();
  };
  epi_pass(std::integral_constant<int, 0>{});
  epi_pass(std::integral_constant<int, 1>{});
}

template <int DT, int K, bool X3, bool WALK>
__global__ __launch_bounds__(kThreads4) void conv4n_kernel(const ConvParams p) {
  if constexpr (!WALK) {
    conv4n_tile<DT, K, X3>(p, blockIdx.x, threadIdx.x);
  } else {
    // walking launch (kernels.h: ConvParams::walk), a kernel of its own so that the one-tile-per-workgroup kernel keeps its register
    // allocation: only the tiles below the live row count; the thread index is laundered per trip, otherwise hipcc hoists every
    // per-lane address out of the tile loop and spills
    unsigned n_blocks;
    {
      int npts = p.npoints;
      if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
      const unsigned m_live = (unsigned)((npts + kPts4 - 1) / kPts4);
        n_blocks = (m_live + 7) / 8 * 8 * (unsigned)p.n_tiles;
    }
    for (unsigned bid = blockIdx.x; bid < n_blocks; bid += gridDim.x) {
      if (bid != blockIdx.x) __syncthreads();    // the previous tile's epilogue is done with the LDS
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      conv4n_tile<DT, K, X3>(p, bid, tid);
    }
  }
}

template <int DT, int K, bool X3, bool WALK>
int launch_conv4n_one_w(const ConvParams& p, hipStream_t stream) {
  constexpr int kMaxDevices = 64;
  static bool attr_set[kMaxDevices] = {};
  int dev = 0;
  NESTI_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !attr_set[dev]) {
    NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv4n_kernel<DT, K, X3, WALK>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds4<K>()));
    if (dev >= 0 && dev < kMaxDevices) attr_set[dev] = true;
  }
  const int groups = (p.m_tiles + 7) / 8;
  const unsigned n_blocks = (unsigned)(groups * 8 * p.n_tiles);
  dim3 grid(WALK ? std::min(n_blocks, p.walk > 1 ? (unsigned)p.walk : kWalkGrid) : n_blocks), block(kThreads4);
  hipLaunchKernelGGL((conv4n_kernel<DT, K, X3, WALK>), grid, block, lds4<K>(), stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

// p.walk picks the walking instantiation (a separate kernel: kernels.h, ConvParams::walk)
template <int DT, int K, bool X3>
int launch_conv4n_one(const ConvParams& p, hipStream_t stream) {
  return p.walk ? launch_conv4n_one_w<DT, K, X3, true>(p, stream) : launch_conv4n_one_w<DT, K, X3, false>(p, stream);
}

template <int DT>
int launch_conv4n_dt(const ConvParams& p, int k, hipStream_t stream) {
  if constexpr (DT != NESTI_F32) {
    if (p.x3native) {
      // pair modes: k = 2 and 4 (experts_n_est); the odd kernels of the ablation towers stay on conv_igemm_kernel there (their
      // pair-loop instantiations of this kernel need more than 256 registers), model.hip: use_conv4
      if (k == 2) return launch_conv4n_one<DT, 2, true>(p, stream);
      if (k == 4) return launch_conv4n_one<DT, 4, true>(p, stream);
      NESTI_FAIL("launch_conv4n: the pair K loop is built for k = 2 and k = 4");
    }
  }
  if (p.x3native) NESTI_FAIL("launch_conv4n: the pair K loop is for the 16-bit kernels");
  if (k == 2) return launch_conv4n_one<DT, 2, false>(p, stream);
  if (k == 3) return launch_conv4n_one<DT, 3, false>(p, stream);
  if (k == 4) return launch_conv4n_one<DT, 4, false>(p, stream);
  if (k == 5) return launch_conv4n_one<DT, 5, false>(p, stream);
  NESTI_FAIL("launch_conv4n: kernel size must be 2 .. 5");
}

}  // namespace

// p.m_tiles = groups of 16 points, p.n_tiles = 64-column tiles, p.n_chunks = 64-byte K chunks, weights packed
// [n tile][chunk][tap][64 rows][64 B] (model.hip: pack_layer, kind 3)
int launch_conv4n(const ConvParams& p, int dtype, int k, hipStream_t stream) {
  if (p.m_tiles <= 0 || p.n_tiles <= 0) return 0;
  if (p.log2S != 2 || p.s_real) NESTI_FAIL("launch_conv4n: the 4^3 volume only");
  if (k < 2 || k > 5 || p.n_taps != k * k * k || p.tap_k != k) NESTI_FAIL("launch_conv4n: all k^3 taps (k = 2 .. 5) must be present");
  for (int t = 0; t < p.n_taps; ++t) {       // ... in x-fastest order around lo = (k - 1) / 2: the kernel derives them from counters
    const int lo = (k - 1) / 2;
    if (p.tap[t][0] != t / (k * k) - lo || p.tap[t][1] != (t / k) % k - lo || p.tap[t][2] != t % k - lo)
      NESTI_FAIL("launch_conv4n: taps must be in (dz, dy, dx) order, x fastest");
  }
  if (p.point_index) NESTI_FAIL("launch_conv4n: no input gather (k^3 layers never read the routed MuPS tensor)");
  if (p.pool_k > 1 || p.split_tile != p.n_tiles) NESTI_FAIL("launch_conv4n: no fused avg-pool / merged layers");
  if (p.mp_mode == 2) NESTI_FAIL("launch_conv4n: max-pool mode 2 is conv1's (a 1x1x1 layer)");
  if (p.mp_mode != 0 && !p.mp_out) NESTI_FAIL("launch_conv4n: fused max-pool needs an output");
  if (dtype == NESTI_BF16) return launch_conv4n_dt<NESTI_BF16>(p, k, stream);
  if (dtype == NESTI_F16) return launch_conv4n_dt<NESTI_F16>(p, k, stream);
  if (dtype == NESTI_F32) return launch_conv4n_dt<NESTI_F32>(p, k, stream);
  NESTI_FAIL("launch_conv4n: unsupported dtype");
}

}  // namespace nesti
