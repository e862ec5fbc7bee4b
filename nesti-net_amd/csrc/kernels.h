// Kernel parameter blocks and launchers shared between the .hip files and the
// host-side graph executor (model.hip).
#pragma once
#include <algorithm>

#include "common.h"

namespace nesti {

constexpr int kTileM = 512;       // GEMM rows (voxels x points) per workgroup
constexpr unsigned kWalkGrid = 1024;   // workgroups of a walking launch (ConvParams::walk): a multiple of 8, so a workgroup's tiles stay
                                       // on its XCD (tile & 7 == blockIdx.x & 7); four per CU, enough to run a non-empty round
constexpr int kRowBytes = 128;    // bytes of one K-chunk row in LDS (64 x 16-bit or 32 x f32)
constexpr int kMaxTaps = 125;     // 5^3

static inline int chunk_elems(int dt) { return kRowBytes / (int)dtype_size(dt); }  // KC

// One conv3d / fully-connected layer as an implicit GEMM:
//   out[r, n] = act( sum_{tap, c} in[shift(r, tap), c] * W[tap, c, n] + bias[n] )
// rows r = point * V + voxel (V = S^3, channels-last); a tap that leaves the S^3 volume
// contributes zero (TF 'SAME', utils/tf_util.py:298-300).
struct ConvParams {
  const void* in;      // [points * V, in_cstride] elements of the model dtype
  void* out;           // [points * V, out_cstride] (model dtype, or f32 if out_f32)
  const void* wpk;     // packed weights [n_tiles][n_chunks][n_taps][TN][128 B] (pre-swizzled)
  const float* bias;   // [n_tiles * TN] BN-folded bias
  const int32_t* npoints_ptr;   // optional device-side point count (top-1 routing); NULL -> npoints
  const int32_t* point_index;   // optional gather of INPUT points (routing); NULL -> identity
  int npoints;         // capacity (grid is sized for this)
  int in_cstride, in_coff;
  int in_chunk_bytes;  // plain (non-pair) K loop of conv_igemm_kernel: byte distance between consecutive 128-byte K chunks of an
                       // input row -- 128, or 256 when a plain-f16 layer reads the hi planes of a pair-layout tensor (model.hip)
  int in_pair;         // plain K loop of conv8n_kernel / conv4n_kernel: 1 = the input tensor has the pair layout and only its hi planes
                       // are read (64-byte chunk c = 32 channels sits at byte (c >> 1) * 256 + (c & 1) * 64 of a row); with
                       // `split` the outputs are written as pairs again -- a single-product layer inside a pair-mode tower
  int out_cstride, out_coff;
  int n_chunks, n_taps;
  int tap_k;           // kernel edge k when all k^3 taps are present in x-fastest order (conv4n_kernel derives the taps from counters)
  int log2S;           // S in {1,2,4,8}: the index space rows are laid out in
  int s_real;          // 0, or the real volume edge when it is smaller than S (3^3 Gaussian grid embedded in 4^3:
                       // voxels with a coordinate >= s_real are dead rows -- never read as neighbours, contents ignored)
  int relu, out_f32;
  int m_tiles, n_tiles;
  // merged inception conv1|conv4 launch: column tiles >= split_tile are conv4's; they write at
  // out_coff2 and average the pre-activation over the pool_k^3 SAME window (pool_k == 1: plain).
  int split_tile;      // == n_tiles for an ordinary layer
  int out_coff2;
  int pool_k;
  // fused tf.nn.max_pool3d 2^3 stride 2 (utils/tf_util.py:424-428) for the FIRST tile group: 1 = write only the pooled
  // tensor, 2 = write the full-resolution tensor and the pooled one
  void* mp_out;        // [points * V/8, mp_cstride], same channel offsets as `out`
  int mp_cstride;
  int mp_mode;
  int mp_mode2;        // 1: the conv4 half (fused avg-pool epilogue) writes ONLY its 2^3 / 2 max-pooled tensor, into mp_out at
                       // out_coff2 -- the block is followed by max_pool3d and nobody reads conv4 at full resolution
  // k^3-tap layers: 1 = the 16 32-row MFMA tiles of a workgroup are (8x,2y,2z) blocks (8^3) / x-lines of the
  // 8 points (4^3), dealt to the 4 SIMDs as a Latin square so that the tiles a padding tap skips are spread evenly
  // over the matrix pipes (conv.hip: tile_row).  0 = tile t holds rows [32t, 32t+32).  2 (2^3 volumes) = a tile is ONE voxel of
  // 32 points and the chunk sits in LDS in (voxel, point) order: padding skips whole tiles in all three axes.
  int remap;
  // A launch that is probably EMPTY (a later round of a routing / flag list walk, a widening pass of the two-stage gate: the row
  // count sits in device memory) is made with a small fixed grid whose workgroups WALK the tiles (tile = blockIdx.x, += gridDim.x,
  // bounded by the live row count): an empty launch then costs a few hundred workgroups instead of one per tile of the capacity
  // (a full-capacity grid of early-exiting workgroups costs ~1.1 ns each: 0.11 ms for 100k).  0 = one tile per workgroup; 1 = kWalkGrid
  // workgroups; > 1 = that many (a multiple of 8): the conditioning guard's towers see a few dozen rows and use 64.
  int walk;
  // NESTI_BF16X3 / NESTI_F16X3 (common.h): in_cstride / in_coff / out_cstride / mp_cstride and n_chunks are PHYSICAL (two planes per
  // 64-channel group); out_coff / out_coff2 stay logical and every 16-bit store goes through split_col + two planes.
  int split;
  int x2;              // 1x1x1 / FC layers of the NESTI_F16X3C filter pass (conv_igemm_kernel, KPIPE): PLAIN 16-bit activations times the
                       // pair-packed weights [W_hi | W_lo] -- hi * W_hi + hi * W_lo, i.e. the layer multiplies by its exact weights.  K chunks
                       // of 32 channels: 64-byte A rows, 128-byte B rows; n_chunks / acc_scale / wpk are the pair packing's.  These layers
                       // are fill-bound, so the second product is nearly free, and it removes the weight-rounding half of the filter's error
  int x3native;        // pair modes: the kernels' pair K loop (conv.hip / conv8n.hip: X3) -- K chunks [hi | lo] x [W_hi | W_lo], three MFMAs
                       // per fragment set; the packed weights follow (model.hip: PackedLayer::x3n)
  // FP8 cross terms (NESTI_F16X8 / NESTI_F16X8C; conv8n.hip X8).  Producer side (a 1x1x1 layer, conv.hip): aux8_out != NULL makes the
  // FIRST tile group (conv1) also write the e4m3 planes of its activated outputs v = hi + lo into the side buffer -- per row and
  // 64-channel group [lo8 64 B | hi8 64 B] with lo8 = e4m3(lo 2^x8_sa), hi8 = e4m3(v 2^x8_sc), saturated at +-448; aux8_stride = bytes
  // per row.  Consumer side (conv8n_kernel): x8 = 1, aux8_in = that buffer, x8_scale_a / x8_scale_b = the E8M0 codes (127 - sa,
  // 127 - sb) of the block scales that undo the pre-scales of the activation and weight planes
  const void* aux8_in;
  void* aux8_out;
  int aux8_stride;
  int x8, x8_sa, x8_sc, x8_scale_a, x8_scale_b;
  // x8_fmt == 6: the block-scaled FP6 (e2m3) form of the same cross terms (conv8n.hip X6).  The side buffer keeps its geometry; the 32
  // bytes of a row's 16-channel chunk (16 where lo8 went, 16 where hi8 went) now hold 32 six-bit elements -- slot 2i = e2m3(lo_i 2^11 / s),
  // slot 2i + 1 = e2m3(hi_i / s) -- then the block's own scale s = 2^(E - 2), E = exponent of the chunk's largest |hi|, as an E8M0 byte
  // (byte 24), zeros after it; |lo_i 2^11| <= |hi_i| element by element, so one scale serves both halves.  The weight rows follow the
  // same pattern (model.hip: pack_layer_x6, the 2^-11 folded into their scale byte); x8_sa / x8_sc / x8_scale_* are not used.
  int x8_fmt;
  float acc_scale;     // the accumulators are multiplied by this before the bias (1, or 2^-s when the layer's packed weights
                       // carry a 2^s scale: NESTI_F16X3 keeps the weight pairs in f16's normal range that way)
  int8_t tap[kMaxTaps][4];   // dz, dy, dx, -
};

int launch_conv(const ConvParams& p, int dtype, int TN, hipStream_t stream);
// k^3 taps (k = 3, 5) on the 8^3 volume, four points per workgroup (conv8.hip): p.m_tiles = groups of 4 points,
// p.n_tiles = 32-column tiles, p.n_chunks = 64-byte K chunks, weights packed [n tile][chunk][tap][32][64 B]
int launch_conv8(const ConvParams& p, int dtype, int k, hipStream_t stream);
// the same layers with A-fragment reuse (conv8n.hip): a workgroup = 4 points x one z half x 64 columns; p.n_tiles = 64-column
// pairs, weights packed [pair][chunk][tap][2 x 32 rows][64 B]
int launch_conv8n(const ConvParams& p, int dtype, int k, hipStream_t stream);
// k^3 taps (k = 2 .. 5) on the 4^3 volume (conv4n.hip): a workgroup = 16 points x 64 voxels x 64 columns, an MFMA tile = one voxel of
// 16 points; p.m_tiles = groups of 16 points, p.n_tiles = 64-column tiles, p.n_chunks = 64-byte K chunks, weights packed
// [n tile][chunk][tap][64 rows][64 B]
int launch_conv4n(const ConvParams& p, int dtype, int k, hipStream_t stream);
// conv8_kernel / conv8n_kernel read the x padding from an LDS address beyond the workgroup's allocation and rely on the
// hardware returning zeros there (gfx950 does: scripts/lds_oob_probe.hip).  Checked once per device, at model creation:
// the device must be gfx950 and a probe kernel must read zeros; otherwise the model is refused (returns 1 with a message).
int conv8_selftest();

struct PoolParams {
  const void* in;
  void* out;
  const int32_t* npoints_ptr;
  int npoints;
  int in_cstride, in_coff, out_cstride, out_coff;
  int C;               // channels to process (multiple of 8)
  int log2S;           // input S
  int split;           // pair modes: cstrides physical, in_coff / out_coff / C logical (split_col), values = hi + lo
};
int launch_maxpool2(const PoolParams& p, int dtype, hipStream_t stream);
// tf.nn.max_pool3d [3,3,3] stride 2 SAME on a 3^3 volume (models/experts_n_est.py:238): input rows in the 4^3-embedded
// layout (log2S = 2), output 2^3 (8 rows per point); output cell o covers input {o, o+1} per axis.
int launch_maxpool3s2(const PoolParams& p, int dtype, hipStream_t stream);

// softmax over the first E logits of each row + first-index arg-max
// (models/experts_n_est.py:177, test_n_est_w_experts.py:150); optional routing lists.
int launch_gate_finish(const float* logits, int lstride, int B, int E, float* probs,
                       int32_t* expert, int32_t* counts /*[E] or NULL*/,
                       int32_t* lists /*[E][B] or NULL*/, hipStream_t stream);
// NESTI_F16X3C, the two-stage gate (pool.hip): keep [B, NESTI_MAX_EXPERTS] f32 (the f16 pass's logits), flag_list [B],
// cstat = the model's device counters, fcounts = a 512-byte block of the call's workspace laid out as int32 words:
//   [0] rows flagged by the filter pass, [kRoundCountsOff + r] of them in recheck round r (cap rows per round),
//   [kTauEffOff] the call's threshold tau_eff as a float (raised to the upper end of every band a widening pass has covered),
//   [kWidenUpperOff] the upper end of the current widening pass's band (float, snapshotted when the pass starts),
//   [kWidenCountOff] rows flagged by the current widening pass, [kWidenRoundsOff + r] of them in its tower round r
constexpr int kMaxCascadeRounds = 24;
constexpr int kRoundCountsOff = 8, kTauEffOff = 40, kWidenUpperOff = 41, kWidenCountOff = 48, kWidenRoundsOff = 56;
static_assert(kRoundCountsOff + kMaxCascadeRounds <= kTauEffOff && kWidenRoundsOff + kMaxCascadeRounds <= 128, "fcounts layout");
int launch_gate_flag(const float* logits, int lstride, int B, int E, float tau, float widen, float* probs, int32_t* expert,
                     float* keep, int32_t* fcounts, int32_t* flag_list, int cap, int n_rounds, unsigned long long* cstat,
                     hipStream_t stream);
// after the recheck rounds, one widening pass: snapshot upper = widen * max_margin_err, flag list (storage reused) of the rows
// in [tau_eff, upper), its round counts, then tau_eff = max(tau_eff, upper)
int launch_gate_widen(const float* keep, int B, int E, float widen, int32_t* fcounts, int32_t* flag_list, int cap, int n_rounds,
                      unsigned long long* cstat, hipStream_t stream);
int launch_gate_recheck(const float* logits, int lstride, const int32_t* flag_list, const int32_t* count_ptr, int cap, int E,
                        const float* keep, float* probs, int32_t* expert, unsigned long long* cstat, hipStream_t stream);
// max_margin_err of the model's counters <-> a device float (multi-GPU: every rank filters with the largest error any rank
// has measured, nesti_model_gate_error_export / _import)
int launch_gate_error_export(const unsigned long long* cstat, float* dst, hipStream_t stream);
int launch_gate_error_import(unsigned long long* cstat, const float* src, int n, hipStream_t stream);
// out[i * n_rounds + r] = clamp(counts[i] - r * cap, 0, cap): the rows of list i that round r of a `cap`-row tower covers
int launch_round_counts(const int32_t* counts, int n_lists, int cap, int n_rounds, int32_t* out, hipStream_t stream);
// ms_sw_n_est's switch (models/ms_sw_n_est.py:80-82): noise = logits[b*lstride]; expert = noise < threshold ? 0 : 1;
// probs[b] = noise (one column); optional routing lists over the 2 towers.
int launch_switch_finish(const float* logits, int lstride, int B, float threshold, float* probs,
                         int32_t* expert, int32_t* counts /*[2] or NULL*/, int32_t* lists /*[2][B] or NULL*/,
                         hipStream_t stream);
// the conditioning guard of the FP8 cross-term experts (pool.hip): band of |n| for this pass, flag expert e's rows inside it, replace
// them by their f16x3 re-evaluation and measure |dn|
int launch_x8_guard_begin(int pass, float thr, float scale, int B, unsigned long long* gstat, float* slot, hipStream_t stream);
int launch_x8_guard_flag(const int32_t* list, const int32_t* count_ptr, int count_cap, const float* normals, const float* slot,
                         int32_t* glist, int32_t* gcount, int cap, unsigned long long* gstat, hipStream_t stream);
int launch_x8_guard_fix(const float* src, int sstride, const int32_t* glist, const int32_t* gcount, int cap, float* normals,
                        unsigned long long* gstat, hipStream_t stream);
// build routing lists from a caller-supplied expert assignment
int launch_route(const int32_t* expert, int B, int E, int32_t* counts, int32_t* lists,
                 hipStream_t stream);
// out[index[j]*3 + c] = src[j*sstride + c] for j < *count (or count_cap when count_ptr NULL)
int launch_scatter3(const float* src, int sstride, const int32_t* index, const int32_t* count_ptr,
                    int count_cap, float* out, hipStream_t stream);

}  // namespace nesti
