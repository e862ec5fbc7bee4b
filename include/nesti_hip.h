/*
 * nesti_hip.h -- C-ABI of libnesti_hip.so: the MI355X (gfx950) implementation of
 * Nesti-Net's per-point inference hot path.
 *
 * The reference (sitzikbs/Nesti-Net, Python 2.7 + TF 1.12) has no FFI or operator
 * registry; its seams are Python call sites.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every `*_dev` pointer is a caller-owned DEVICE pointer (e.g. torch tensor
 *     .data_ptr()); the library never allocates outputs or frees inputs;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     launches are asynchronous on it and no call synchronises the device
 *     unless stated;
 *   - return value 0 = ok, non-zero = error; nesti_last_error() returns a
 *     thread-local message for the last failing call on this thread;
 *   - model handles are immutable after creation (except the NESTI_F16X3C gate margin, which must not be changed while
 *     forward calls are in flight, and its device-side counters, which forward calls update atomically): concurrent
 *     forward calls on different streams are safe provided they use different workspaces.
 */
#ifndef NESTI_HIP_H
#define NESTI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NESTI_MAX_SCALES 4
#define NESTI_MAX_EXPERTS 8
#define NESTI_MUPS_CH 20 /* channels per scale: utils/tf_util.py:711-720 */

/* element types for activations / weights.  NESTI_BF16X3 / NESTI_F16X3 are MODEL dtypes only (nesti_model_create):
 * activations and weights are kept as a 16-bit (hi, lo) pair, v ~ hi + lo, and every multiply is the three 16-bit MFMA
 * products hi*hi + lo*hi + hi*lo accumulated in fp32 -- a third of the 16-bit rate instead of the fp32 MFMA rate.
 * bf16 pairs hold 2^-17 relative; f16 pairs hold max(2^-23 relative, 3e-8 absolute) with the weights of each layer scaled
 * by a power of two into f16's normal range (undone in the epilogue): the mode that keeps test_n_est_w_experts.py's outputs
 * within the 1e-5 cosine / arg-max tolerance with two orders of magnitude to spare. */
enum { NESTI_F32 = 0, NESTI_BF16 = 1, NESTI_F16 = 2, NESTI_BF16X3 = 3, NESTI_F16X3 = 4, NESTI_F16X3C = 5, NESTI_F16X8 = 6, NESTI_F16X8C = 7 };
/* NESTI_F16X3C ("cascade", MODEL dtype only, gated models): everything that reaches the outputs is computed as in
 * NESTI_F16X3 -- the experts, and the gating net for every query whose decision could depend on it -- but the gating net
 * first runs in plain f16 as a FILTER (plain f16 activations and tap layers; its 1x1x1 / FC layers, which are fill-bound, multiply
 * by the exact pair-packed weights -- two products -- which halves the filter's error variance for a third more time in those
 * layers): a query whose top-2 logit margin in that pass is at least the gate margin tau keeps its arg-max (a flip would need a
 * filter error of tau), every other query is decided again by the f16x3 gating net.
 * expert_out / normals_out are then those of NESTI_F16X3 as long as the f16 gate's error on a logit difference stays below
 * the margin, which every call re-measures on the queries it decides twice and widens by itself when the measured error
 * comes within a factor NESTI_GATE_WIDEN of it (nesti_model_cascade_stats); probs_out carries the filter pass's probabilities
 * (within ~0.013 of NESTI_F16X3's) for the queries that were not re-decided. */

/* NESTI_F16X8 / NESTI_F16X8C (round 6; MODEL dtypes only, experts_n_est on the 8^3 grid): NESTI_F16X3 / NESTI_F16X3C with the two
 * CROSS terms of the pair scheme -- lo * W_hi + hi * W_lo, 2^-11 of a layer's result -- of the EXPERT towers' tap layers at 8^3
 * (models/experts_n_est.py:258-262: conv2 (3^3) and conv3 (5^3) of inception1 / inception2, 86 % of an expert's multiply-accumulates)
 * computed by one block-scaled MFMA of a narrow format (K = 64, fp32 accumulate into the same accumulator: FP6 e2m3 x e2m3 with one scale
 * per 16-channel block by default, or FP8 e4m3 x e4m3 with one scale per layer -- nesti_model_set_x8_format) instead of four f16 MFMAs;
 * hi * W_hi stays an exact f16 product.  The gating net -- filter AND recheck -- is untouched, so expert_out is NESTI_F16X3C's bit for
 * bit.  The normals differ from NESTI_F16X3's by a residual 28x below single-product f16 (1 - cos p50 6e-10, p99 2e-8), and the
 * queries on which that residual could matter -- expert outputs of very small norm -- are evaluated again in f16x3 proper by the
 * conditioning guard (nesti_model_set_x8_guard below): measured over 2.38 M queries of 32 clouds, max 1 - cos 5.8e-7 against f16x3,
 * none above 2.5e-6, 0.21 % of the outputs re-evaluated (profiles/r06_stream32_check.json), against the 1e-5 tolerance.  The e4m3
 * planes' power-of-two pre-scales come from the producing layer's folded batch-norm (|beta| + 8 |gamma|: a data-free bound on its
 * activations; larger values saturate at the format's 448 and lose only their own cross terms).
 * nesti_model_set_x8_layers picks the layers: bit 0 / 1 = inception1 conv2 (3^3) / conv3 (5^3), bit 2 / 3 = inception2 conv2 / conv3;
 * the default is 0b1111 (all four), 0b1010 keeps to the 5^3 layers, 0 is NESTI_F16X3 proper.  Must not be changed while forward calls
 * are in flight (declared below: nesti_model_set_x8_layers). */

/* which graph nesti_model_create builds */
enum {
  NESTI_ARCH_EXPERTS = 0, /* models/experts_n_est.py:40-108  (MoE, the hot path)    */
  NESTI_ARCH_SINGLE = 1,  /* models/ss_norm_est.py:35-92     (BASELINE config 0)    */
  NESTI_ARCH_MULTI = 2,   /* models/ms_norm_est.py:45-140    (multi-scale ablation) */
  NESTI_ARCH_SWITCH = 3   /* models/ms_sw_n_est.py:41-89     (noise-switched two-scale ablation):
                           * 2 scales; gate = noise_est_net on scale 1, tower 0 = 'small' (scale 0),
                           * tower 1 = 'large' (scale 1); n_experts / expert_scale_* are ignored      */
};

/* Hyper-parameters the reference reads from parameters.p / gmm.p
 * (test_n_est_w_experts.py:46-54, :201). */
typedef struct {
  int arch;                                /* NESTI_ARCH_*                                      */
  int n_scales;                            /* len(patch_radius)                                 */
  int points_per_scale;                    /* num_point (P)                                     */
  int grid_n;                              /* Gaussians per axis: 8, or 3 (experts_n_est only)  */
  double variance;                         /* gmm covariance (0.0156)                           */
  int n_experts;                           /* E                                                 */
  int expert_scale_lo[NESTI_MAX_EXPERTS];  /* min(expert_dict[i])   models/experts_n_est.py:100 */
  int expert_scale_cnt[NESTI_MAX_EXPERTS]; /* len(expert_dict[i])   models/experts_n_est.py:101 */
} nesti_config_t;

/* One named float32 host tensor in TF variable layout
 * (conv [kd,kh,kw,Cin,Cout] utils/tf_util.py:289; fc [In,Out] utils/tf_util.py:334). */
typedef struct {
  const char* name;
  const float* data; /* host pointer; may be NULL in nesti_model_describe output */
  int ndim;
  int64_t dims[5];
} nesti_tensor_t;

typedef struct nesti_model nesti_model_t;

const char* nesti_last_error(void);
const char* nesti_version(void);

/* Fill cfg with the published Nesti-Net configuration (3 radii, 8^3 Gaussians, variance 0.0156, 7 experts with the
 * expert_dict of train_n_est_w_experts.py:62) -- not the argparse defaults of that script (3^3 Gaussians). */
void nesti_default_config(nesti_config_t* cfg);

/* utils/utils.py:70-95 get_3d_grid_gmm: host arrays w[n^3], mu[n^3*3], sigma[n^3*3]
 * (sigma = sqrt(covariances_), as fed at test_n_est_w_experts.py:146). */
int nesti_gmm_grid(int n, double variance, float* w, float* mu, float* sigma);

/* utils/tf_util.py:655-753 get_3dmfv_n_est + models/experts_n_est.py:66-76 (MuPS
 * assembly) with the exact TF placeholder contract (models/experts_n_est.py:26-35):
 *   points_dev [B, S*P, 3] f32, n_eff_dev [B, S] int32
 *   out_dev    [B, R, R, R, out_cstride] of out_dtype; channel 20*s+c holds
 *              scale s, statistic c; channels >= 20*S are written as zero.
 * Rows whose n_eff is 0 (the zero-padded tail of the reference's last batch,
 * test_n_est_w_experts.py:134-140) are written as zeros instead of NaN.
 * out_dtype NESTI_BF16X3 / NESTI_F16X3: out_cstride is a multiple of 128 16-bit elements and channel c is stored as the planes
 * hi at 128*(c/64) + c%64 and lo 64 elements further (value = hi + lo). */
int nesti_mups_forward(const nesti_config_t* cfg, const float* points_dev,
                       const int32_t* n_eff_dev, int B, void* out_dev, int out_dtype,
                       int out_cstride, void* stream);

/* utils/pcpnet_dataset.py:286-343 __getitem__ (center='point', use_pca=False,
 * point_tuple=1) for M query points of one cloud, on the GPU:
 *   cloud_dev [N,3] f32; query_idx_dev [M] int32 (NULL = points query_row0..query_row0+M-1,
 *   the 'full' sampler utils/pcpnet_dataset.py:41-55); r_abs[S] = bbdiag*rad as double
 *   (utils/pcpnet_dataset.py:282); query_row0 = patch row of the first query within
 *   its shape (so the subsample below does not depend on how rows are batched).
 * Ball membership is the fp64 test scipy's cKDTree applies (:304).  When a ball
 * holds more than P points the P kept are those with the smallest
 * (hash(seed, query_row0+i, scale, index), index) keys -- a uniform P-subset like :320-321
 * but reproducible; see DESIGN.md.  Outputs (any may be NULL):
 *   points_out_dev [M,S*P,3] f32, n_eff_out_dev [M,S] int32,
 *   nbr_idx_out_dev [M,S*P] int32 (-1 padded), n_ball_out_dev [M,S] int32 (uncapped).
 * grid_ws_dev / grid_ws_bytes: scratch from nesti_patches_workspace_bytes(N). */
size_t nesti_patches_workspace_bytes(int N);
/* Step 1 (once per cloud; replaces the cKDTree build, utils/pcpnet_dataset.py:37): bounding
 * box, cell counts, scan and cell-ordered copy of the cloud into grid_ws_dev. */
int nesti_patches_grid(const nesti_config_t* cfg, const float* cloud_dev, int N,
                       const double* r_abs, void* grid_ws_dev, size_t grid_ws_bytes,
                       void* stream);
/* Step 2 (per batch of queries; replaces __getitem__, utils/pcpnet_dataset.py:286-343). */
int nesti_patches_query(const nesti_config_t* cfg, const float* cloud_dev, int N,
                        const int32_t* query_idx_dev, int M, const double* r_abs,
                        uint64_t seed, int query_row0, float* points_out_dev,
                        int32_t* n_eff_out_dev, int32_t* nbr_idx_out_dev,
                        int32_t* n_ball_out_dev, const void* grid_ws_dev,
                        size_t grid_ws_bytes, void* stream);
/* Steps 1 + 2 in one call. */
int nesti_patches_build(const nesti_config_t* cfg, const float* cloud_dev, int N,
                        const int32_t* query_idx_dev, int M, const double* r_abs,
                        uint64_t seed, int query_row0, float* points_out_dev,
                        int32_t* n_eff_out_dev, int32_t* nbr_idx_out_dev,
                        int32_t* n_ball_out_dev, void* grid_ws_dev, size_t grid_ws_bytes,
                        void* stream);

/* ---- the reference's own subsample ORDER on the GPU (utils/pcpnet_dataset.py:304, 320-321) ----------------------------------
 * When a ball holds n > P points the reference keeps ball[rng.choice(n, P, replace=False)] of cKDTree's traversal-ordered ball.
 * query_ball_point returns a ball in ascending position in tree.indices, so the order is a sort key and the random stream only
 * needs the ball sizes:
 *   nesti_patches_count      n_ball_out_dev[M, S] = ball sizes of the queries (the count pass; nothing else is written);
 *   nesti_refstream_picks    (host, below) replays the shared RandomState over those sizes in visiting order -> pick table;
 *   nesti_patches_query_ref  the patch tensors exactly as PointcloudPatchDataset.__getitem__ builds them: per query the points
 *                            inside the largest ball are sorted in LDS by tree_rank_dev[i] (= position of point i in
 *                            tree.indices; tree_order_dev = tree.indices itself, both int32 [N], built on the host with
 *                            scipy.spatial.cKDTree(pts, 10) like utils/pcpnet_dataset.py:37), each scale's ball is taken in
 *                            that order, n <= P: as it is, rows beyond n zero; n > P: ball[picks] with the P uint16 picks at
 *                            picks_dev[pick_offsets_dev[q * S + s]] (offset -1: the ball holds <= P points).
 * A ball of more than nesti_patches_ref_max_ball() points does not fit the LDS sort: its n_eff comes back as -1 and its rows
 * zero -- callers hold the counts and must refuse such a shape before the launch (provider.CloudPatches does). */
int nesti_patches_count(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                        const double* r_abs, int query_row0, int32_t* n_ball_out_dev, const void* grid_ws_dev,
                        size_t grid_ws_bytes, void* stream);
int nesti_patches_ref_max_ball(void);
int nesti_patches_query_ref(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                            const double* r_abs, int query_row0, const int32_t* tree_rank_dev, const int32_t* tree_order_dev,
                            const uint16_t* picks_dev, const int64_t* pick_offsets_dev, float* points_out_dev,
                            int32_t* n_eff_out_dev, int32_t* nbr_idx_out_dev, const void* grid_ws_dev, size_t grid_ws_bytes,
                            void* stream);

/* Enumerate the variables the graph for cfg expects (names follow the reference's
 * scopes, models/experts_n_est.py:155-314).  Call with infos=NULL to get the count. */
int nesti_model_describe(const nesti_config_t* cfg, int* n_tensors, nesti_tensor_t* infos,
                         int max_infos);

/* tf.train.Saver().restore equivalent (test_n_est_w_experts.py:98-105): takes the
 * float32 variables, folds the inference-mode batch norm (utils/tf_util.py:491-494)
 * into weights+bias, repacks for the MFMA kernels in `dtype` (any of the four element types) and uploads.
 * This call allocates device memory owned by the handle and synchronises. */
int nesti_model_create(const nesti_config_t* cfg, const nesti_tensor_t* tensors,
                       int n_tensors, int dtype, nesti_model_t** out);
void nesti_model_destroy(nesti_model_t* m);

/* ---- NESTI_F16X3C: the gate margin and the running statistics of the two-stage gate ---------------------------------
 * tau is in units of the gating net's last-layer outputs (the values softmax sees, models/experts_n_est.py:174-177).
 * The margin protects itself: a forward call filters with
 *     tau_eff = max(tau, NESTI_GATE_WIDEN x max_margin_err measured since the last reset)
 * and, after its own recheck rounds, re-decides the rows whose f16 margin lies between that threshold and NESTI_GATE_WIDEN x
 * the (possibly larger) error it has just measured -- up to NESTI_GATE_WIDEN_PASSES further, normally empty passes on the
 * device (each takes a snapshot of the largest error when it starts and covers the band up to NESTI_GATE_WIDEN x that, so an
 * error first seen inside a widening pass is covered by the next pass of the same call), no host synchronisation.  A row
 * keeps the f16 arg-max only while its margin is at least NESTI_GATE_WIDEN x the largest error the f16 gate has shown on any
 * row decided twice up to the start of the call's last widening pass.  The guarantee is per call: calls in flight on other
 * streams share the counters, and what they measure after this call's last snapshot protects this stream from its next
 * call on.
 * nesti_model_cascade_stats synchronises `stream`, copies the counters accumulated by every forward call since the last
 * reset and optionally resets them:
 *   queries        rows that went through the gate,
 *   rechecked      rows decided by the f16x3 gate (f16 top-2 margin below tau_eff, or caught by a widening round),
 *   changed        rechecked rows whose arg-max differs between the two gates,
 *   max_margin_err largest |(l_a - l_k)_f16 - (l_a - l_k)_f16x3| over the rechecked rows and all experts k, a = the f16
 *                  arg-max: the f16 gate's error on exactly the quantity tau guards,
 *   sum_sq_pair_err / pairs: the same errors squared and summed over all (rechecked row, k != a) pairs, and their count:
 *                  sqrt(sum / pairs) is the standard deviation sigma of the f16 pass's error on one logit difference (the
 *                  errors are rounding noise: zero-mean, independent of the margin); tau is chosen as a multiple of it,
 *   widened        rows re-decided by a widening pass, widen_events: widening passes that were not empty,
 *   tau_eff        the threshold the next forward call starts from. */
#define NESTI_GATE_WIDEN 1.5f
#define NESTI_GATE_WIDEN_PASSES 3
typedef struct {
  uint64_t queries, rechecked, changed;
  float max_margin_err;
  float tau;
  double sum_sq_pair_err;
  uint64_t pairs;
  uint64_t widened, widen_events;
  float tau_eff;
} nesti_cascade_stats_t;
int nesti_model_set_gate_margin(nesti_model_t* m, float tau);
int nesti_model_cascade_stats(const nesti_model_t* m, nesti_cascade_stats_t* out, int reset, void* stream);
/* Multi-GPU (no reference counterpart): every rank should filter with the largest error ANY rank has
 * measured.  _export writes this model's max_margin_err to dst_dev[0]; _import raises it to the largest finite value of
 * src_dev[0..n).  Both are one tiny kernel on `stream`, no host synchronisation: the value rides in a spare row of the
 * per-step all-gather (nesti-net_amd/dist.py) instead of a collective of its own. */
int nesti_model_gate_error_export(const nesti_model_t* m, float* dst_dev, void* stream);
int nesti_model_gate_error_import(nesti_model_t* m, const float* src_dev, int n, void* stream);

/* EXPERIMENT (pair-mode experts_n_est models, 8^3 grid, created after nesti_experiment_mix_enable(1); no reference counterpart): which of the experts' k^3 tap layers run ONE
 * 16-bit product (hi * W_hi, reading only the hi planes of their pair-layout input, writing pairs again) instead of three.
 * Bits: 0 / 1 = inception1 conv2 (3^3) / conv3 (5^3), 2 / 3 = inception2 conv2 / conv3, 4 / 5 = inception4 conv2 (2^3) / conv3 (4^3).
 * 0 (the default) is NESTI_F16X3 proper.  Any other value does NOT hold the 1e-5 cosine tolerance on every query
 * (profiles/r05_expert_mix.txt); it exists to measure that.  Must not be changed while forward calls are in flight. */
int nesti_experiment_mix_enable(int on); /* process-wide, BEFORE nesti_model_create: pack the extra single-product copies (default off) */
int nesti_model_set_expert_mix(nesti_model_t* m, int mask);
/* The same switch for the gating net of a NESTI_F16X3 / NESTI_BF16X3 model (non-cascade nesti_gate_forward / nesti_forward):
 * on != 0 runs ALL its k^3 tap layers at 8^3 / 4^3 single-product, the 1x1x1 / FC layers stay three-product ("medium" gate);
 * on == 2 additionally rounds every layer's output to 16 bits (lo plane = 0): the numerics of a plain-f16 gate whose 1x1x1 / FC
 * layers multiply by the exact weights (hi * W_hi + hi * W_lo) -- the "exact-weight filter" of profiles/r05_gate_medium.txt; mode 2
 * exists only in measurement builds (EXTRA_CXXFLAGS=-DNESTI_EXPERIMENT_XW), the product library refuses it. */
int nesti_model_set_gate_mix(nesti_model_t* m, int on);
/* NESTI_F16X8 / NESTI_F16X8C models: which expert tap layers at 8^3 take their cross terms through FP8 (see the dtype's comment above;
 * default 0b1111, 0 = NESTI_F16X3 proper). */
int nesti_model_set_x8_layers(nesti_model_t* m, int mask);
/* ... and in which format: 8 = OCP e4m3 with one power-of-two scale per layer and operand (above), 6 = OCP e2m3 (FP6) with one scale per
 * 16-channel block of a row, taken from the block's largest |hi| and stored beside the elements (the instruction runs FP6 at twice the
 * FP8 rate; |lo 2^11| <= |hi| element by element, so one scale serves a block's lo and hi halves; the residual is ~1.2x e4m3's and
 * the conditioning guard below bounds its effect exactly as it does for e4m3).  The model holds both packings; must not be changed
 * while forward calls are in flight; recalibrate the guard after a change. */
int nesti_model_set_x8_format(nesti_model_t* m, int bits);
/* The CONDITIONING GUARD of those models (top-1 routed calls: nesti_forward, nesti_estimate_normals[_multi], nesti_experts_forward
 * with an expert assignment).  The FP8 residual moves an expert's raw output n by |dn| -- ~5e-5, at most 2e-4 on 100 000 queries, and
 * independent of |n| -- and 1 - cos against the three-product result is (|dn| / |n|)^2 / 2: only outputs of very small norm can be
 * tilted by more than the bar.  Every query whose |n| is below
 *     thr_eff = max(thr, NESTI_X8_GUARD_WIDEN x largest |dn| measured so far / sqrt(2 x NESTI_X8_GUARD_BAR))
 * is evaluated again by its expert in f16x3 proper and that result replaces the FP8 one (a fraction of a per cent of the queries);
 * the rows decided twice measure |dn|, so the threshold follows the measurement like the two-stage gate's margin does, and a call
 * whose own measurement opens a wider band re-evaluates the rows in between in NESTI_X8_GUARD_WIDEN_PASSES further passes.  An
 * un-re-evaluated query therefore differs from f16x3 by 1 - cos <= NESTI_X8_GUARD_BAR / NESTI_X8_GUARD_WIDEN^2 as long as |dn| stays
 * below the largest value measured.  nesti_model_set_x8_guard: thr >= 0 (default NESTI_X8_GUARD_DEFAULT; calibrate it like the gate
 * margin: calibrate.calibrate_x8_guard), thr < 0 switches the guard off, +inf re-evaluates everything (calibration).  Must not be
 * changed while forward calls are in flight.  Because thr_eff follows what the model has measured so far, WHICH rows are re-evaluated
 * (hence the last bits of a few normals near the threshold, never the expert index) depends on the order and partition of the batches
 * a model has seen since its counters were last reset (nesti_model_x8_guard_stats with reset = 1). */
#define NESTI_X8_GUARD_BAR 2.5e-6f
#define NESTI_X8_GUARD_WIDEN 1.5f
#ifndef NESTI_X8_GUARD_WIDEN_PASSES
#define NESTI_X8_GUARD_WIDEN_PASSES 1
#endif
#define NESTI_X8_GUARD_DEFAULT 0.25f
typedef struct {
  uint64_t queries;    /* routed queries seen since the last reset                                      */
  uint64_t rechecked;  /* ... of them re-evaluated in f16x3                                              */
  uint64_t dropped;    /* flagged rows NOT re-evaluated because an expert's guard list was full (0 unless the threshold is absurd) */
  float max_dn;        /* largest |n_x8 - n_f16x3| measured on the re-evaluated rows                     */
  float thr, thr_eff;  /* the configured threshold and the one the next call starts from                 */
} nesti_x8_guard_stats_t;
int nesti_model_set_x8_guard(nesti_model_t* m, float thr);
int nesti_model_x8_guard_stats(const nesti_model_t* m, nesti_x8_guard_stats_t* out, int reset, void* stream);

/* Workspace of ONE tower for `batch` queries, from the configuration alone (no device needed): tower = -1 the gating
 * net, 0..E-1 an expert.  dtype as nesti_model_create (NESTI_F16X3C: the gate figure is the f16 filter's). */
size_t nesti_tower_workspace_bytes(const nesti_config_t* cfg, int dtype, int tower, int batch);

/* Scratch size for forward calls of up to max_batch points. */
size_t nesti_workspace_bytes(const nesti_model_t* m, int max_batch);
int nesti_model_mups_cstride(const nesti_model_t* m); /* channel stride of the internal MuPS tensor, in elements
                                                       * (NESTI_BF16X3 / NESTI_F16X3: 2 x the padded channel count) */
int nesti_model_mups_rows(const nesti_model_t* m);    /* rows per point of the internal MuPS tensor: 512 (8^3 grid)
                                                       * or 64 (3^3 grid: row 16i+4j+k of a 4^3 index space, rows with
                                                       * a coordinate of 3 are zero) */
/* MuPS (utils/tf_util.py:655-753 + models/experts_n_est.py:66-76) of a batch in the layout and dtype
 * nesti_gate_forward / nesti_experts_forward read: mups_out_dev is [B, nesti_model_mups_rows, nesti_model_mups_cstride]. */
int nesti_model_mups(const nesti_model_t* m, const float* points_dev, const int32_t* n_eff_dev, int B,
                     void* mups_out_dev, void* stream);

/* scale_manager_net + arg-max (models/experts_n_est.py:155-179,
 * test_n_est_w_experts.py:150): mups_dev is [B,R^3,cstride] in the model dtype.
 * probs_out_dev [B,E] f32 (row-major, i.e. the transpose done at :151),
 * expert_out_dev [B] int32 (first index on ties, like np.argmax).
 * NESTI_ARCH_SWITCH: probs_out_dev is [B,1] = noise_est (models/ms_sw_n_est.py:75) and
 * expert_out_dev[b] = noise_est < 0.015 ? 0 (small) : 1 (large) (:80-82). */
int nesti_gate_forward(const nesti_model_t* m, const void* mups_dev, int B, void* ws_dev,
                       size_t ws_bytes, float* probs_out_dev, int32_t* expert_out_dev,
                       void* stream);

/* normal_est_net for every expert (models/experts_n_est.py:99-105, 243-291).
 * expert_dev == NULL : evaluate all experts, normals_out_dev is [E,B,3] (what the
 *                      reference's sess.run returns, test_n_est_w_experts.py:148);
 * expert_dev != NULL : top-1 routing -- only expert_dev[b] is evaluated for point b
 *                      and normals_out_dev is [B,3] (== n_est[e*,b,:], :152). */
int nesti_experts_forward(const nesti_model_t* m, const void* mups_dev, const int32_t* expert_dev,
                          int B, void* ws_dev, size_t ws_bytes, float* normals_out_dev,
                          void* stream);

/* One sess.run([n_pred, experts_prob]) + arg-max/select
 * (test_n_est_w_experts.py:142-152) with top-1 routing:
 *   points_dev [B,S*P,3] f32, n_eff_dev [B,S] int32 ->
 *   normals_out_dev [B,3] f32, expert_out_dev [B] int32, probs_out_dev [B,E] f32.
 * For NESTI_ARCH_SINGLE / NESTI_ARCH_MULTI only normals are produced (expert/probs may be NULL).
 * For NESTI_ARCH_SWITCH (one sess.run of test_n_est_w_switching.py:136) probs_out_dev is [B,1] =
 * noise_est and expert_out_dev the tower chosen by the 0.015 threshold. */
int nesti_forward(const nesti_model_t* m, const float* points_dev, const int32_t* n_eff_dev,
                  int B, void* ws_dev, size_t ws_bytes, float* normals_out_dev,
                  int32_t* expert_out_dev, float* probs_out_dev, void* stream);

/* The body of the reference's predict loop for ONE shape in one call (test_n_est_w_experts.py:129-152 with
 * utils/pcpnet_dataset.py:286-343 behind the data loader): search grid (when build_grid != 0; replaces the cKDTree of
 * utils/pcpnet_dataset.py:37) + ball query + MuPS + gating + routed expert for patch rows
 * [query_row0, query_row0 + M) of the cloud (query_idx_dev / r_abs / seed / grid_ws_dev as in nesti_patches_query),
 * `batch` queries at a time.  The patch tensors only ever live in ws_dev
 * (nesti_estimate_workspace_bytes(m, batch) bytes).  Outputs as nesti_forward: [M,3], [M], [M,E]. */
size_t nesti_estimate_workspace_bytes(const nesti_model_t* m, int batch);
/* The same figure from the configuration alone (no model, no device): what a caller needs to size batches against free memory
 * before anything is created (nesti-net_amd/cli.py: fit_batch).  dtype as nesti_model_create. */
size_t nesti_estimate_workspace_bytes_for_config(const nesti_config_t* cfg, int dtype, int batch);
int nesti_estimate_normals(const nesti_model_t* m, const float* cloud_dev, int N,
                           const int32_t* query_idx_dev, int M, const double* r_abs, uint64_t seed,
                           int query_row0, int batch, int build_grid, void* grid_ws_dev,
                           size_t grid_ws_bytes, void* ws_dev, size_t ws_bytes,
                           float* normals_out_dev, int32_t* expert_out_dev, float* probs_out_dev,
                           void* stream);

/* Several shapes in flight (BASELINE config 4; also every rank of a multi-GPU job, which holds a block of rows of
 * every shape): the queries of all items are processed as ONE stream of `batch`-sized batches, so small shapes / small
 * shards share the gate and expert launches instead of each paying for its own partially filled rounds.  Item i
 * contributes n_queries rows (as nesti_estimate_normals would for that shape, search grid already built); outputs are the
 * items' rows concatenated in order: [sum n_queries, 3], [sum], [sum, E].  8^3 Gaussian grid. */
typedef struct {
  const float* cloud_dev;        /* [n_points, 3] f32 */
  int n_points;
  const int32_t* query_idx_dev;  /* [n_queries] or NULL = rows query_row0 .. query_row0 + n_queries - 1 */
  int n_queries;
  double r_abs[NESTI_MAX_SCALES];
  uint64_t seed;
  int query_row0;
  const void* grid_ws_dev;       /* from nesti_patches_grid on this shape */
  size_t grid_ws_bytes;
} nesti_shape_queries_t;
int nesti_estimate_normals_multi(const nesti_model_t* m, const nesti_shape_queries_t* items, int n_items, int batch,
                                 void* ws_dev, size_t ws_bytes, float* normals_out_dev, int32_t* expert_out_dev,
                                 float* probs_out_dev, void* stream);

/* ---- the reference's own subsample stream, replayed on the host (refreplay.cpp) ----------
 * utils/pcpnet_dataset.py:237-240, 320-321: ONE numpy RandomState(seed) shared by every patch and scale of every shape, one
 * rng.choice(n, P, replace=False) per ball that holds n > P points, in visiting order (patch-major, scale-minor).  The stream
 * object holds the MT19937 state between calls.  nesti_refstream_picks walks `n_balls` ball sizes in visiting order and writes,
 * for every over-full ball, the P positions choice() returns (positions in cKDTree's traversal-ordered ball, uint16: balls of
 * more than 65535 points are refused) at picks_out[offsets_out[b] .. + P); offsets_out[b] = -1 for a ball with n <= P (it draws
 * nothing).  A call that fails leaves the stream untouched.  Host only; bit-identical to numpy (tests/test_refreplay.py). */
typedef struct nesti_refstream nesti_refstream_t;
nesti_refstream_t* nesti_refstream_create(uint32_t seed);
void nesti_refstream_destroy(nesti_refstream_t* s);
int nesti_refstream_picks(nesti_refstream_t* s, const int32_t* sizes, int64_t n_balls, int P, uint16_t* picks_out,
                          int64_t picks_capacity, int64_t* offsets_out, int64_t* n_over_out);

/* ---- text I/O of the file seam (host only) ------------------------------------------------
 * Reading stays np.loadtxt (utils/pcpnet_dataset.py:250): numpy 2's parser is faster than a strtod loop, and the
 * .npy cache the reference writes next to the file makes it a one-off.  The writers are native: np.savetxt's
 * '%.18e' formatting dominates the wall time of a shape once the compute takes a second. */
/* np.savetxt(path, a) with the default '%.18e' format (test_n_est_w_experts.py:182-183, 187-188). */
int nesti_write_text_f32(const char* path, const float* data, int64_t rows, int cols);
/* np.savetxt(path, a.astype(int), fmt='%i') (test_n_est_w_experts.py:185-186). */
int nesti_write_text_i32(const char* path, const int32_t* data, int64_t rows);

/* CRC-32C of n host bytes continuing from `crc` (0 to start): the checksum of TensorFlow's tensor bundle, verified on
 * restore (test_n_est_w_experts.py:98-105 -> tf_ckpt.read_bundle).  Stored masked: ((crc >> 15) | (crc << 17)) + 0xa282ead8. */
uint32_t nesti_crc32c(const void* data, size_t n, uint32_t crc);

/* The OCP FP6 e2m3 code (sign bit 5, exponent bits 4-3 with bias 1, mantissa bits 2-0; grid 0, 0.125 .. 7.5) of value * inv_scale, rounded to
 * nearest even and saturating -- the host-side encoder of the FP6 weight packing (nesti_model_set_x8_format), exposed so that it can be
 * checked against the instruction's own decoding without a device (tests/test_abi.py; scripts/fp6_probe.hip holds the device side). */
int nesti_f32_to_e2m3(float value, float inv_scale);

/* ---- measurement support (bench.py's roofline leg; no reference counterpart) ------------
 * nesti_profile_enable(1) makes every kernel launch of the forward path record a pair of
 * hipEvents on its stream; nesti_profile_read() synchronises on them and returns, per
 * (phase, category), the summed kernel time in ms and the number of launches since enable:
 * arrays of NESTI_PROF_PHASES * NESTI_PROF_CATEGORIES entries, index = phase * NESTI_PROF_CATEGORIES + category.
 * Categories are kernels (the four conv categories are the layer classes of DESIGN.md 4.3 / 4.4); phases say which part
 * of the forward pass launched them (NESTI_F16X3C: the gate's f16 filter pass counts as GATE, its f16x3 pass as RECHECK).
 * Not thread-safe; leave it off outside measurements. */
enum { NESTI_PROF_CONV8_K5 = 0,   /* conv8n_kernel, 5^3 taps at 8^3                                 */
       NESTI_PROF_CONV8_K3 = 1,   /* conv8n_kernel, 3^3 taps at 8^3                                 */
       NESTI_PROF_TAPS = 2,       /* conv4n_kernel: k^3 taps at 4^3; conv_igemm_kernel: taps at 2^3
                                   * and on the 3^3 grid embedded in 4^3                            */
       NESTI_PROF_ONE_BY_ONE = 3, /* conv_igemm_kernel, 1x1x1 layers (+ fused avg-pool) and FC      */
       NESTI_PROF_MUPS = 4, NESTI_PROF_POOL = 5, NESTI_PROF_PATCHES = 6, NESTI_PROF_CATEGORIES = 7 };
enum { NESTI_PHASE_INPUT = 0,     /* search grid, ball query, MuPS                                  */
       NESTI_PHASE_GATE = 1, NESTI_PHASE_RECHECK = 2, NESTI_PHASE_EXPERTS = 3,
       NESTI_PHASE_GUARD = 4,     /* the conditioning guard's f16x3 re-evaluations (NESTI_F16X8 / NESTI_F16X8C); its first pass overlaps
                                   * the experts on an auxiliary stream, so its launch durations are not wall time                  */
       NESTI_PROF_PHASES = 5 };
int nesti_profile_enable(int on);
int nesti_profile_read(double* ms /*[NESTI_PROF_PHASES * NESTI_PROF_CATEGORIES]*/,
                       long long* launches /*[NESTI_PROF_PHASES * NESTI_PROF_CATEGORIES]*/);
/* Multiply-accumulates per point of one tower (tower = -1: gating net, 0..E-1: expert), over the layers of conv
 * category `kind` (NESTI_PROF_CONV8_K5 .. NESTI_PROF_ONE_BY_ONE; -1: all of them):
 *   nominal = dense conv as TensorFlow executes it (zero-padding taps included),
 *   useful  = taps that land inside the volume only (the algorithmic figure),
 *   issued  = what the MFMA kernels issue for ONE 16-bit product: channel padding included, padding taps included except
 *             the MFMA tiles the kernels skip (x-lines whose y + dy or z + dz leaves the volume, at 8^3 and 4^3); the
 *             pair modes issue three such products per multiply. */
int nesti_model_macs(const nesti_model_t* m, int tower, int kind, double* nominal, double* useful,
                     double* issued);

#ifdef __cplusplus
}
#endif
#endif /* NESTI_HIP_H */
