// Host side of libnesti_hip.so: C-ABI entry points, the Nesti-Net graph
// (models/experts_n_est.py:40-314) expressed as a list of kernel launches, batch-norm folding
// and weight repacking for the MFMA kernels, and the workspace planner.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <string>
#include <vector>

#include "kernels.h"

#ifndef NESTI_GUARD_WALK_GRID     // measurement builds may override it (scripts/ab_guard.sh); 1 = the default walking grid (kWalkGrid)
#define NESTI_GUARD_WALK_GRID 64
#endif
#ifndef NESTI_GUARD_LANES         // auxiliary streams of the conditioning guard, one per caller stream (measurement builds: 1)
#define NESTI_GUARD_LANES 2
#endif
#ifndef NESTI_GATE_WIDEN_WALK_GRID    // the two-stage gate's widening passes hold at most a few hundred rows (normally none)
#define NESTI_GATE_WIDEN_WALK_GRID 1
#endif
#ifndef NESTI_GUARD_WIDEN_WALK_GRID   // the guard's widening pass is normally EMPTY: its launches only have to be dispatched and retire
#define NESTI_GUARD_WIDEN_WALK_GRID 8
#endif

namespace nesti {

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_error;
// nesti_experiment_mix_enable: models created while this is set also pack their tap layers for the single-product experiments
// (nesti_model_set_expert_mix / _gate_mix); off by default, so product models neither pay the packing time nor hold the copies
static bool g_experiment_mix = false;
void set_error(const std::string& msg) { g_error = msg; }

// ------------------------------------------------------------------------------------------
// optional per-category kernel timing with hipEvents on the launch stream (bench.py roofline leg)
// ------------------------------------------------------------------------------------------
constexpr int kProfSlots = NESTI_PROF_PHASES * NESTI_PROF_CATEGORIES;   // slot = phase * NESTI_PROF_CATEGORIES + category
struct ProfState {
  bool on = false;
  int phase = NESTI_PHASE_INPUT;     // set by the forward path (prof_phase); launches are booked under it
  std::vector<hipEvent_t> pool;      // recycled events
  std::vector<std::pair<hipEvent_t, hipEvent_t>> spans[kProfSlots];
  double ms[kProfSlots] = {0};
  long long launches[kProfSlots] = {0};
};
static ProfState g_prof;
static std::mutex g_prof_mu;   // forward calls may come from several host threads (one stream each)

static hipEvent_t prof_event() {
  if (!g_prof.pool.empty()) { hipEvent_t e = g_prof.pool.back(); g_prof.pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
// A stream that is being captured into a hipGraph records nothing: events captured into a graph cannot be
// synchronised on or timed afterwards.
static bool prof_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cs != hipStreamCaptureStatusNone;
}
void prof_phase(int phase) { if (g_prof.on) g_prof.phase = phase; }
// prof_begin returns a token for prof_end (-1: nothing recorded): slot * 2^20 + index of the span within the slot
int prof_begin(int cat, hipStream_t st) {
  if (!g_prof.on || prof_capturing(st)) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  // the input kernels are launched between forward passes, whatever phase the last one ended in
  const int phase = (cat == NESTI_PROF_MUPS || cat == NESTI_PROF_PATCHES) ? NESTI_PHASE_INPUT : g_prof.phase;
  const int slot = phase * NESTI_PROF_CATEGORIES + cat;
  if (g_prof.spans[slot].size() >= (1u << 20)) return -1;
  hipEvent_t a = prof_event(), b = prof_event();
  (void)hipEventRecord(a, st);
  g_prof.spans[slot].push_back({a, b});
  return (slot << 20) | ((int)g_prof.spans[slot].size() - 1);
}
void prof_end(int, int token, hipStream_t st) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const int slot = token >> 20, idx = token & ((1 << 20) - 1);
  if (slot < kProfSlots && idx < (int)g_prof.spans[slot].size()) (void)hipEventRecord(g_prof.spans[slot][idx].second, st);
}
static void prof_collect() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int c = 0; c < kProfSlots; ++c) {
    for (auto& sp : g_prof.spans[c]) {
      (void)hipEventSynchronize(sp.second);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, sp.first, sp.second) == hipSuccess) g_prof.ms[c] += ms;
      g_prof.launches[c] += 1;
      g_prof.pool.push_back(sp.first);
      g_prof.pool.push_back(sp.second);
    }
    g_prof.spans[c].clear();
  }
}

uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
uint16_t host_f32_to_f16(float f) {
  _Float16 h = (_Float16)f;   // clang host: IEEE RNE conversion
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}

float host_f16_to_f32(uint16_t b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}

namespace {

constexpr int kPad = 64;   // channel segments are zero-padded to multiples of this
inline int pad_to(int c, int a) { return (c + a - 1) / a * a; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// ------------------------------------------------------------------------------------------
// graph description
// ------------------------------------------------------------------------------------------
struct LayerDesc {
  std::string scope;
  std::string scope2;        // non-empty: a second 1x1 layer on the same input, fused into this launch
  int pool_k = 1;            // avg-pool window applied to scope2's pre-activation (conv4 of an inception)
  bool is_fc = false;
  int k = 1;                 // kernel size (1 for fc)
  int log2S = 0;             // spatial index space the layer runs at
  int s_real = 0;            // 3: the 3^3 Gaussian grid embedded in the 4^3 index space (0: the volume is 2^log2S)
  int cin = 0, cout = 0;     // real channel counts (TF variable shapes); scope2 has the same shape
  std::vector<int> in_pos;   // real input channel -> position inside the padded input slice
  int Cin_p = 0, Cout_p = 0;
  bool bn = true, relu = true;
};

struct BufSpec { int log2S; int C; bool f32; bool aux8 = false; };   // aux8: side buffer of e4m3 planes, 2 bytes per channel (conv8n.hip X8)

struct Op {
  enum Kind { CONV, MAX, MAX3 } kind;   // MAX3: max_pool3d [3,3,3] stride 2 SAME, 3^3 (embedded) -> 2^3
  int in_buf = 0, in_coff = 0, out_buf = 0, out_coff = 0, out_coff2 = 0;
  int in_cstride = 0;          // 0: the input buffer's channel count; else a flattened view (FC on S^3 x C)
  int mp_buf = -1, mp_mode = 0; // fused 2^3 max-pool of the first tile group into this buffer (1: pooled only, 2: both)
  int mp_mode2 = 0;             // 1: the conv4 half writes only its pooled tensor too (kernels.h: ConvParams::mp_mode2)
  int layer = -1;
  int C = 0, k = 0, log2S = 0;
  bool out_f32 = false;
  // FP8 cross terms (expert towers of NESTI_F16X8 / NESTI_F16X8C models): a block's conv1 (aux_out_buf >= 0) can also write the e4m3
  // planes of its outputs; the block's tap layers (aux_in_buf >= 0, x8_bit = their bit of nesti_model::x8_mask) read them.
  // x8_bits (producer) = the bits of the tap layers that read its planes; aux_layer (consumer) = the producer's layer index
  int aux_out_buf = -1, aux_in_buf = -1, x8_bit = -1, x8_bits = 0, aux_layer = -1;
};

struct Tower {
  std::vector<BufSpec> bufs;   // bufs[0] = MuPS X0 (external)
  std::vector<Op> ops;
  int out_buf = -1;            // f32 [NB, 64]
  int n_out = 0;               // real outputs (E or 3)
};

struct ChanMap { std::vector<int> pos; int C = 0; };   // real channel -> padded position, padded width

struct Graph {
  nesti_config_t cfg;
  int gate_x0_log2S() const { return cfg.grid_n == 3 ? 2 : 3; }   // index space of the MuPS rows of one point
  int mups_cstride = 64;
  bool x8 = false;           // NESTI_F16X8 / NESTI_F16X8C: the expert towers carry side buffers for the FP8 cross terms
  std::vector<LayerDesc> layers;
  Tower gate;
  std::vector<Tower> experts;
};

struct Builder {
  Graph& g;
  bool x8 = false;           // expert towers of an NESTI_F16X8 / NESTI_F16X8C model: side buffers for the 8^3 blocks' tap layers
  int x8_block = 0;          // ... and which 8^3 inception block of the tower is being built (0, 1)
  int s_real = 0;            // stamped on the layers built while it is set (conv_net_3g)
  explicit Builder(Graph& gg) : g(gg) {}

  int add_layer(const LayerDesc& d) { g.layers.push_back(d); return (int)g.layers.size() - 1; }

  int conv(Tower& T, const std::string& scope, int k, int log2S, int in_buf, int in_coff, const ChanMap& in,
           int cout, int out_buf, int out_coff, bool bn = true, bool relu = true, bool fc = false, bool out_f32 = false,
           const std::string& scope2 = "", int out_coff2 = 0, int pool_k = 1) {
    LayerDesc d;
    d.scope = scope; d.scope2 = scope2; d.pool_k = pool_k; d.is_fc = fc; d.k = k; d.log2S = log2S;
    d.s_real = fc ? 0 : s_real;
    d.cin = (int)in.pos.size(); d.cout = cout; d.in_pos = in.pos; d.Cin_p = in.C;
    d.Cout_p = pad_to(cout, kPad) * (scope2.empty() ? 1 : 2); d.bn = bn; d.relu = relu;
    Op op; op.kind = Op::CONV; op.in_buf = in_buf; op.in_coff = in_coff; op.out_buf = out_buf; op.out_coff = out_coff;
    op.out_coff2 = out_coff2;
    op.layer = add_layer(d); op.log2S = log2S; op.out_f32 = out_f32;
    T.ops.push_back(op);
    return d.Cout_p;
  }

  // models/experts_n_est.py:294-314.  conv1 and conv4 read the same tensor (avg_pool3d commutes with the
  // 1x1x1 convolution), so they are one launch; conv4's columns are averaged in the kernel epilogue.
  // then_maxpool: the block is followed by tf_util.max_pool3d 2^3/2 (e.g. models/experts_n_est.py:198).  conv1
  // stores full resolution (conv2/conv3 read it) AND its pooled tensor, conv2/conv3 store only the pooled tensor
  // (nobody reads them at full resolution); conv4's columns come out of the avg-pool epilogue at full resolution
  // and are max-pooled by the small standalone kernel, restricted to their channel range.  Returns the buffer the
  // next block reads.
  int inception(Tower& T, const std::string& scope, int in_buf, const ChanMap& in, int F, int k0, int k1, int log2S,
                ChanMap* out_map, bool then_maxpool = false) {
    const int H = F / 2;   // int(n_filters/2)  :299
    const int Fp = pad_to(F, kPad), Hp = pad_to(H, kPad);
    const int C = Fp + Hp + Hp + Fp;
    T.bufs.push_back({log2S, C, false});
    const int ob = (int)T.bufs.size() - 1;
    int pb = -1;
    if (then_maxpool) {
      T.bufs.push_back({log2S - 1, C, false});
      pb = (int)T.bufs.size() - 1;
    }
    ChanMap c1; c1.C = Fp; for (int i = 0; i < F; ++i) c1.pos.push_back(i);
    conv(T, scope + "_conv1", 1, log2S, in_buf, 0, in, F, ob, 0, true, true, false, false,
         scope + "_conv4", Fp + Hp + Hp, k0);
    // conv4 behind a max-pool: when its avg-pool runs in the epilogue (k0 > 1) the 2^3 max is taken there as well and
    // its full-resolution columns are never written; with k0 == 1 (plain columns) the small standalone kernel pools them
    const bool fuse4 = then_maxpool && k0 > 1 && !s_real;
    if (then_maxpool) { T.ops.back().mp_buf = pb; T.ops.back().mp_mode = 2; T.ops.back().mp_mode2 = fuse4 ? 1 : 0; }
    const size_t conv1_op = T.ops.size() - 1;
    int ab = -1;
    if (x8 && log2S == 3 && !s_real && x8_block < 2 && (k0 == 3 || k0 == 5) && (k1 == 3 || k1 == 5)) {
      BufSpec ax{log2S, Fp, false};
      ax.aux8 = true;
      T.bufs.push_back(ax);
      ab = (int)T.bufs.size() - 1;
      T.ops[conv1_op].aux_out_buf = ab;
      T.ops[conv1_op].x8_bits = 3 << (2 * x8_block);
    }
    conv(T, scope + "_conv2", k0, log2S, ob, 0, c1, H, ob, Fp);
    if (then_maxpool) { T.ops.back().mp_buf = pb; T.ops.back().mp_mode = 1; }
    if (ab >= 0) { T.ops.back().aux_in_buf = ab; T.ops.back().x8_bit = 2 * x8_block; T.ops.back().aux_layer = T.ops[conv1_op].layer; }
    conv(T, scope + "_conv3", k1, log2S, ob, 0, c1, H, ob, Fp + Hp);
    if (then_maxpool) { T.ops.back().mp_buf = pb; T.ops.back().mp_mode = 1; }
    if (ab >= 0) { T.ops.back().aux_in_buf = ab; T.ops.back().x8_bit = 2 * x8_block + 1; T.ops.back().aux_layer = T.ops[conv1_op].layer; ++x8_block; }
    out_map->pos.clear();
    for (int i = 0; i < F; ++i) out_map->pos.push_back(i);
    for (int i = 0; i < H; ++i) out_map->pos.push_back(Fp + i);
    for (int i = 0; i < H; ++i) out_map->pos.push_back(Fp + Hp + i);
    for (int i = 0; i < F; ++i) out_map->pos.push_back(Fp + Hp + Hp + i);
    out_map->C = C;
    if (then_maxpool) {
      if (!fuse4) {
        Op op; op.kind = Op::MAX; op.in_buf = ob; op.out_buf = pb; op.in_coff = op.out_coff = Fp + Hp + Hp;
        op.C = Fp; op.log2S = log2S;
        T.ops.push_back(op);
      }
      return pb;
    }
    return ob;
  }

  // fully connected stack on a [NB,1,C] feature (utils/tf_util.py:314-351)
  int fc_stack(Tower& T, int in_buf, ChanMap m, const std::vector<std::string>& scopes, const std::vector<int>& widths,
               bool last_relu, int first_in_cstride = 0) {
    int buf = in_buf;
    for (size_t i = 0; i < scopes.size(); ++i) {
      const bool last = (i + 1 == scopes.size());
      const int Cp = pad_to(widths[i], kPad);
      T.bufs.push_back({0, Cp, last});
      const int ob = (int)T.bufs.size() - 1;
      conv(T, scopes[i], 1, 0, buf, 0, m, widths[i], ob, 0, /*bn=*/!last, /*relu=*/last ? last_relu : true, /*fc=*/true,
           /*out_f32=*/last);
      if (i == 0) T.ops.back().in_cstride = first_in_cstride;
      m.pos.clear(); m.C = Cp;
      for (int c = 0; c < widths[i]; ++c) m.pos.push_back(c);
      buf = ob;
    }
    return buf;
  }

  void init_tower(Tower& T) {
    T.bufs.clear(); T.ops.clear();
    T.bufs.push_back({g.cfg.grid_n == 3 ? 2 : 3, g.mups_cstride, false});   // 0: X0 (3^3 grid: rows in a 4^3 index space)
  }

  // conv_net_3g (models/experts_n_est.py:217-240): four inception blocks on the 3^3 grid (kernel sizes [2,3], [2,3],
  // [1,2], [1,2]; k0 = 1 makes conv2 a 1x1x1 layer and the avg-pool of the conv4 branch the identity), then
  // max_pool3d [3,3,3] stride 2 SAME -> 2^3 x 1536, flattened voxel-major.  The 27 voxels live in a 4^3 index space
  // (kernels.h: ConvParams::s_real).  Returns the pooled buffer; *flat describes it as one FC input row.
  int conv_net_3g(Tower& T, const std::string& s, ChanMap m, ChanMap* flat) {
    s_real = 3;
    int b = inception(T, "inception1" + s, 0, m, 128, 2, 3, 2, &m);
    b = inception(T, "inception2" + s, b, m, 256, 2, 3, 2, &m);
    b = inception(T, "inception3" + s, b, m, 256, 1, 2, 2, &m);
    b = inception(T, "inception4" + s, b, m, 512, 1, 2, 2, &m);
    s_real = 0;
    T.bufs.push_back({1, m.C, false});
    const int pb = (int)T.bufs.size() - 1;
    Op op; op.kind = Op::MAX3; op.in_buf = b; op.out_buf = pb; op.in_coff = op.out_coff = 0; op.C = m.C; op.log2S = 2;
    T.ops.push_back(op);                                                    // maxpool5  :238
    flat->pos.clear(); flat->C = 8 * m.C;
    for (int v = 0; v < 8; ++v)
      for (size_t c = 0; c < m.pos.size(); ++c) flat->pos.push_back(v * m.C + m.pos[c]);
    return pb;
  }

  // scale_manager_net + conv_net_8g (models/experts_n_est.py:155-215)
  void build_gate() {
    Tower& T = g.gate;
    init_tower(T);
    const int S = g.cfg.n_scales;
    ChanMap m; m.C = g.mups_cstride;
    for (int c = 0; c < 20 * S; ++c) m.pos.push_back(c);
    const std::string s = "gating_conv";
    if (g.cfg.grid_n == 3) {   // models/experts_n_est.py:162-163
      ChanMap flat;
      const int pb = conv_net_3g(T, s, m, &flat);
      T.out_buf = fc_stack(T, pb, flat, {"fc1noise", "fc2noise", "fc3noise", "fc4noise"}, {1024, 256, 128, g.cfg.n_experts},
                           /*last_relu=*/true, flat.C);
      T.n_out = g.cfg.n_experts;
      return;
    }
    int b = inception(T, "inception1" + s, 0, m, 128, 3, 5, 3, &m);
    b = inception(T, "inception2" + s, b, m, 256, 3, 5, 3, &m);
    b = inception(T, "inception3" + s, b, m, 256, 3, 5, 3, &m, true);    // + maxpool4  :198
    b = inception(T, "inception5" + s, b, m, 512, 2, 4, 2, &m);
    b = inception(T, "inception6" + s, b, m, 512, 2, 4, 2, &m, true);    // + maxpool7  :206
    b = inception(T, "inception8" + s, b, m, 512, 1, 2, 1, &m, true);    // + maxpool9  :211
    T.out_buf = fc_stack(T, b, m, {"fc1noise", "fc2noise", "fc3noise", "fc4noise"}, {1024, 256, 128, g.cfg.n_experts},
                         /*last_relu=*/true);   // relu on fc4: models/experts_n_est.py:174
    T.n_out = g.cfg.n_experts;
  }

  // The "ss" tower shared by the ablation models: inception x3 @8^3 [3,5] -> maxpool -> inception x2 @4^3
  // [3,k1_small] -> maxpool -> flatten 2^3 x 1536 voxel-major (tf.reshape) -> fc 1024/256/128/n_out.  Used by
  // ss_norm_est.get_model (models/ss_norm_est.py:35-92), ms_norm_est.get_model (models/ms_norm_est.py:45-140) and
  // the three towers of ms_sw_n_est (models/ms_sw_n_est.py:139-215).  Dropout is the identity at inference.  fc1
  // is an FC over a flattened view of the pooled buffer.
  void build_ss_tower(Tower& T, int scale_lo, int scale_cnt, const std::function<std::string(int)>& name,
                      const std::string& fc_suffix, int n_out, bool last_relu, int k1_small) {
    init_tower(T);
    ChanMap m; m.C = g.mups_cstride;
    for (int c = 0; c < 20 * scale_cnt; ++c) m.pos.push_back(20 * scale_lo + c);
    int b = inception(T, name(1), 0, m, 128, 3, 5, 3, &m);
    b = inception(T, name(2), b, m, 256, 3, 5, 3, &m);
    b = inception(T, name(3), b, m, 256, 3, 5, 3, &m, true);
    b = inception(T, name(5), b, m, 512, 3, k1_small, 2, &m);
    b = inception(T, name(6), b, m, 512, 3, k1_small, 2, &m, true);
    ChanMap flat; flat.C = 8 * m.C;
    for (int v = 0; v < 8; ++v)
      for (size_t c = 0; c < m.pos.size(); ++c) flat.pos.push_back(v * m.C + m.pos[c]);
    T.out_buf = fc_stack(T, b, flat, {"fc1" + fc_suffix, "fc2" + fc_suffix, "fc3" + fc_suffix, "fc4" + fc_suffix},
                         {1024, 256, 128, n_out}, last_relu, flat.C);
    T.n_out = n_out;
  }

  // ss_norm_est (one scale, 4^3 kernels [3,5], scopes 'inception<L>') / ms_norm_est (S scales concatenated on
  // channels, 4^3 kernels [3,4], scopes 'inception_s<S-1>_l_<L>')
  void build_single() {
    const bool multi = g.cfg.arch == NESTI_ARCH_MULTI;
    const int S = g.cfg.n_scales;
    auto name = [=](int layer) {
      return multi ? "inception_s" + std::to_string(S - 1) + "_l_" + std::to_string(layer) : "inception" + std::to_string(layer);
    };
    build_ss_tower(g.experts[0], 0, S, name, "", 3, /*last_relu=*/false, multi ? 4 : 5);
  }

  // ms_sw_n_est.get_model (models/ms_sw_n_est.py:41-89): noise_est_net on the LARGE scale (scale 1) with a ReLU on
  // its single output (:172), normal_est_net 'small' on scale 0 and 'large' on scale 1 (:77-78); the driver keeps
  // n_est_small where noise_est < 0.015 (:80-82).  Here: gate = the noise tower, expert 0 = small, expert 1 = large.
  void build_switch() {
    auto scoped = [](const std::string& sfx) {
      return [sfx](int layer) { return "inception" + std::to_string(layer) + sfx; };
    };
    build_ss_tower(g.gate, 1, 1, scoped("noise"), "noise", 1, /*last_relu=*/true, 5);
    build_ss_tower(g.experts[0], 0, 1, scoped("small"), "small", 3, /*last_relu=*/false, 5);
    build_ss_tower(g.experts[1], 1, 1, scoped("large"), "large", 3, /*last_relu=*/false, 5);
  }

  // normal_est_net, 8^3 branch (models/experts_n_est.py:243-291)
  void build_expert(int i) {
    Tower& T = g.experts[i];
    init_tower(T);
    const int lo = g.cfg.expert_scale_lo[i], cnt = g.cfg.expert_scale_cnt[i];
    ChanMap m; m.C = g.mups_cstride;
    for (int c = 0; c < 20 * cnt; ++c) m.pos.push_back(20 * lo + c);   // MuPS[..., start:end]  :100-102
    const std::string s = "Expert_" + std::to_string(i);
    if (g.cfg.grid_n == 3) {   // models/experts_n_est.py:275-276: the 3^3 branch ignores `divider`
      ChanMap flat;
      const int pb = conv_net_3g(T, s + "_expert_conv", m, &flat);
      T.out_buf = fc_stack(T, pb, flat, {"fc1" + s, "fc2" + s, "fc3" + s, "fc4" + s}, {512, 128, 64, 3}, /*last_relu=*/false, flat.C);
      T.n_out = 3;
      return;
    }
    const int F1 = 128 / cnt;   // np.round(128 / divider) under Python-2 integer division  :254
    x8 = g.x8; x8_block = 0;
    int b = inception(T, "inception1" + s, 0, m, F1, 3, 5, 3, &m);
    b = inception(T, "inception2" + s, b, m, 256, 3, 5, 3, &m, true);    // + maxpool3  :261
    x8 = false;
    b = inception(T, "inception4" + s, b, m, 256, 2, 4, 2, &m, true);    // + maxpool5  :266
    b = inception(T, "inception6" + s, b, m, 512, 2, 4, 1, &m, true);    // + maxpool7  :271
    T.out_buf = fc_stack(T, b, m, {"fc1" + s, "fc2" + s, "fc3" + s, "fc4" + s}, {512, 128, 64, 3}, /*last_relu=*/false);
    T.n_out = 3;
  }
};

int build_graph(const nesti_config_t* cfg, Graph* g, bool x8 = false) {
  if (cfg->arch != NESTI_ARCH_EXPERTS && cfg->arch != NESTI_ARCH_SINGLE && cfg->arch != NESTI_ARCH_MULTI &&
      cfg->arch != NESTI_ARCH_SWITCH)
    NESTI_FAIL("unknown arch");
  if (cfg->arch == NESTI_ARCH_SWITCH && cfg->n_scales != 2)
    NESTI_FAIL("NESTI_ARCH_SWITCH (ms_sw_n_est) takes exactly two scales (models/ms_sw_n_est.py:50)");
  if (cfg->arch == NESTI_ARCH_SINGLE && cfg->n_scales != 1) NESTI_FAIL("NESTI_ARCH_SINGLE (ss_norm_est) takes exactly one scale");
  if (cfg->grid_n != 8 && !(cfg->grid_n == 3 && cfg->arch == NESTI_ARCH_EXPERTS))
    NESTI_FAIL("the Gaussian grid must be 8^3 (any model) or 3^3 (experts_n_est only: the ablation models are 8^3-only, "
               "models/ms_sw_n_est.py:183)");
  if (cfg->n_scales < 1 || cfg->n_scales > NESTI_MAX_SCALES) NESTI_FAIL("bad n_scales");
  if (cfg->n_experts < 1 || cfg->n_experts > NESTI_MAX_EXPERTS) NESTI_FAIL("bad n_experts");
  for (int i = 0; i < cfg->n_experts && cfg->arch == NESTI_ARCH_EXPERTS; ++i) {
    if (cfg->expert_scale_cnt[i] < 1 || cfg->expert_scale_lo[i] < 0 ||
        cfg->expert_scale_lo[i] + cfg->expert_scale_cnt[i] > cfg->n_scales)
      NESTI_FAIL("expert scale range outside [0, n_scales)");
  }
  g->cfg = *cfg;
  g->x8 = x8 && cfg->arch == NESTI_ARCH_EXPERTS && cfg->grid_n == 8;
  g->mups_cstride = pad_to(20 * cfg->n_scales, kPad);
  g->layers.clear();
  Builder b(*g);
  g->gate = Tower();
  if (cfg->arch == NESTI_ARCH_SINGLE || cfg->arch == NESTI_ARCH_MULTI) {
    g->cfg.n_experts = 1;
    g->experts.assign(1, Tower());
    b.build_single();
    return 0;
  }
  if (cfg->arch == NESTI_ARCH_SWITCH) {
    g->cfg.n_experts = 2;
    g->cfg.expert_scale_lo[0] = 0; g->cfg.expert_scale_lo[1] = 1;
    g->cfg.expert_scale_cnt[0] = g->cfg.expert_scale_cnt[1] = 1;
    g->experts.assign(2, Tower());
    b.build_switch();
    return 0;
  }
  b.build_gate();
  g->experts.assign(cfg->n_experts, Tower());
  for (int i = 0; i < cfg->n_experts; ++i) b.build_expert(i);
  return 0;
}

// ------------------------------------------------------------------------------------------
// device-resident packed layers
// ------------------------------------------------------------------------------------------
struct PackedLayer {
  void* wpk = nullptr;
  float* bias = nullptr;
  int TN = 64, n_tiles = 0, split_tile = 0, n_chunks = 0, n_taps = 0;
  int kind = 0;              // 0: conv_igemm_kernel (conv.hip), 2: conv8n_kernel (conv8n.hip: 4 points x a z half x 64 columns
                             // per workgroup), 3: conv4n_kernel (conv4n.hip: 16 points x 64 voxels x 64 columns)
  bool x3n = false;          // pair modes: K chunks [hi | lo] / [W_hi | W_lo], three MFMAs per fragment set
  int mix_bit = -1;          // packed_mix entries: which bit of nesti_model::expert_mix switches this layer to the single-product loop
  float acc_scale = 1.0f;    // 2^-s when the packed weights carry a 2^s scale (NESTI_F16X3)
  bool x8 = false;           // packed_x8 entries: rows [W_hi f16 16 ch | W_hi8 16 ch | W_lo8 16 ch] (conv8n.hip X8)
  int x8_sb = 0;             // ... W_hi8 = e4m3(W_hi 2^sb), W_lo8 = e4m3(W_lo 2^(sb + 11))
  int x8_sc = 0;             // producer layers (an 8^3 block's conv1): hi8 = e4m3(v 2^sc), lo8 = e4m3(lo 2^(sc + 11))
  int8_t tap[kMaxTaps][4];
};

}  // namespace
}  // namespace nesti

struct nesti_model {
  nesti::Graph graph;
  int dtype = NESTI_BF16;                      // compute dtype of everything that reaches the outputs (NESTI_F16X3C -> NESTI_F16X3)
  std::vector<nesti::PackedLayer> packed;
  // NESTI_F16X3C: the gating net's layers once more in plain f16 (indexed like `packed`, other entries empty), the gate
  // margin and the device counters of the two-stage gate (include/nesti_hip.h: nesti_cascade_stats_t)
  bool cascade = false;
  std::vector<nesti::PackedLayer> packed_fast;
  // pair modes, EXPERIMENT (nesti_model_set_expert_mix): the experts' k^3 tap layers at 8^3 / 4^3 once more
  // in plain f16 -- a layer whose bit is set in expert_mix reads the hi planes of its pair-layout input, multiplies ONE product
  // and writes its outputs as pairs again
  std::vector<nesti::PackedLayer> packed_mix;
  int expert_mix = 0;
  // NESTI_F16X8 / NESTI_F16X8C: the experts' tap layers at 8^3 once more in the FP8 cross-term packing; x8_mask picks which of them run
  // it (include/nesti_hip.h: nesti_model_set_x8_layers)
  std::vector<nesti::PackedLayer> packed_x8;
  int x8_mask = 0;
  // the same layers once more in the block-scaled FP6 form of the cross terms, and which of the two forms forward calls use (8 or 6;
  // include/nesti_hip.h: nesti_model_set_x8_format)
  std::vector<nesti::PackedLayer> packed_x6;
  int x8_fmt = 6;
  // ... and their conditioning guard (pool.hip: x8_guard_*): outputs with |n| below max(x8_guard_thr, NESTI_X8_GUARD_WIDEN x largest
  // measured |dn| / theta) are re-evaluated in f16x3 proper; gstat = the device counters (include/nesti_hip.h: nesti_x8_guard_stats_t)
  float x8_guard_thr = NESTI_X8_GUARD_DEFAULT;
  int x8_guard_walk = NESTI_GUARD_WALK_GRID;   // workgroups of the guard towers' walking launches (kernels.h: ConvParams::walk)
  unsigned long long* gstat = nullptr;
  // a guard tower sees a handful of rows, so it is latency-bound (one workgroup walks a layer's whole K loop: ~3 ms per tower): expert
  // e's guard runs on ONE auxiliary stream while the caller's stream goes on with expert e + 1 (events in both directions).  One
  // stream, not E: with more streams than hardware queues (4 by default) the event waits of one stream block the kernels of another
  // that shares its queue -- measured: E side streams cost the two-stream mode 3 %
  // kGuardLanes such sets, keyed by the caller's stream (two library batches in flight on two streams -- the way the command line runs --
  // get an auxiliary stream each: 2 + 2 = the 4 hardware queues; a third caller stream shares lane 0)
  static constexpr int kGuardLanes = NESTI_GUARD_LANES;
  struct GuardLane {
    hipStream_t gstream = nullptr;
    hipEvent_t gev_done[NESTI_MAX_EXPERTS] = {}, gev_join = nullptr;
    hipStream_t owner = nullptr;
    bool owned = false;
    std::mutex gmu;
  };
  mutable GuardLane glane[kGuardLanes];
  mutable std::mutex glane_mu;
  GuardLane* guard_lane(hipStream_t caller) const {
    std::lock_guard<std::mutex> lk(glane_mu);
    for (int i = 0; i < kGuardLanes; ++i) if (glane[i].owned && glane[i].owner == caller) return &glane[i];
    for (int i = 0; i < kGuardLanes; ++i) if (!glane[i].owned) { glane[i].owned = true; glane[i].owner = caller; return &glane[i]; }
    return &glane[0];
  }
  int gate_mix = 0;          // EXPERIMENT (nesti_model_set_gate_mix): the f16x3 gating passes run their tap layers single-product
  float tau = 0.25f;
  unsigned long long* cstat = nullptr;
  ~nesti_model() {
    for (auto* v : {&packed, &packed_fast, &packed_mix, &packed_x8, &packed_x6})
      for (auto& p : *v) {
        if (p.wpk) (void)hipFree(p.wpk);
        if (p.bias) (void)hipFree(p.bias);
      }
    if (cstat) (void)hipFree(cstat);
    if (gstat) (void)hipFree(gstat);
    for (auto& gl : glane) {
      if (gl.gstream) (void)hipStreamDestroy(gl.gstream);
      for (auto& ev : gl.gev_done) if (ev) (void)hipEventDestroy(ev);
      if (gl.gev_join) (void)hipEventDestroy(gl.gev_join);
    }
  }
};

namespace nesti {
namespace {

struct TensorTable {
  std::map<std::string, const nesti_tensor_t*> by_name;
  const nesti_tensor_t* get(const std::string& n) const {
    auto it = by_name.find(n);
    return it == by_name.end() ? nullptr : it->second;
  }
};

bool shape_is(const nesti_tensor_t* t, std::initializer_list<int64_t> dims) {
  if (!t || !t->data || t->ndim != (int)dims.size()) return false;
  int i = 0;
  for (int64_t d : dims) if (t->dims[i++] != d) return false;
  return true;
}

// Expected variables of one layer, in TF naming (utils/tf_util.py:289-302, 332-342, 473-479).
void layer_tensors(const LayerDesc& d, std::vector<std::pair<std::string, std::vector<int64_t>>>* out) {
  for (const std::string* sc : {&d.scope, &d.scope2}) {
    if (sc->empty()) continue;
    if (d.is_fc) out->push_back({*sc + "/weights", {d.cin, d.cout}});
    else out->push_back({*sc + "/weights", {d.k, d.k, d.k, d.cin, d.cout}});
    out->push_back({*sc + "/biases", {d.cout}});
    if (d.bn) {
      for (const char* n : {"beta", "gamma", "mean", "var"}) out->push_back({*sc + "/bn/" + n, {d.cout}});
    }
  }
}

// BN-folded weights/bias of one TF layer (utils/tf_util.py:298-311, 491-494)
struct Folded {
  const float* w = nullptr;        // [taps][cin][cout]
  std::vector<float> scale, bias;  // per real output channel
};

int fold_layer(const LayerDesc& d, const std::string& scope, const TensorTable& tt, Folded* f) {
  const nesti_tensor_t* w = tt.get(scope + "/weights");
  const nesti_tensor_t* b = tt.get(scope + "/biases");
  const bool wok = d.is_fc ? shape_is(w, {d.cin, d.cout}) : shape_is(w, {d.k, d.k, d.k, d.cin, d.cout});
  if (!wok) NESTI_FAIL("missing or mis-shaped tensor " + scope + "/weights");
  if (!shape_is(b, {d.cout})) NESTI_FAIL("missing or mis-shaped tensor " + scope + "/biases");
  f->w = w->data;
  f->scale.assign(d.cout, 1.0f);
  f->bias.resize(d.cout);
  for (int n = 0; n < d.cout; ++n) f->bias[n] = b->data[n];
  if (d.bn) {
    const nesti_tensor_t* beta = tt.get(scope + "/bn/beta");
    const nesti_tensor_t* gamma = tt.get(scope + "/bn/gamma");
    const nesti_tensor_t* mean = tt.get(scope + "/bn/mean");
    const nesti_tensor_t* var = tt.get(scope + "/bn/var");
    if (!shape_is(beta, {d.cout}) || !shape_is(gamma, {d.cout}) || !shape_is(mean, {d.cout}) || !shape_is(var, {d.cout}))
      NESTI_FAIL("missing or mis-shaped batch-norm tensors under " + scope + "/bn/");
    for (int n = 0; n < d.cout; ++n) {
      // tf.nn.batch_normalization(x, mean, var, beta, gamma, 1e-3)  (utils/tf_util.py:494)
      const double inv = (double)gamma->data[n] / sqrt((double)var->data[n] + 1e-3);
      f->scale[n] = (float)inv;
      f->bias[n] = (float)(((double)b->data[n] - (double)mean->data[n]) * inv + (double)beta->data[n]);
    }
  }
  return 0;
}

// Which layers run on conv8n_kernel (conv8n.hip): the k^3 taps (k = 3, 5) on the 8^3 volume; everything else is
// conv_igemm_kernel's (conv.hip)
bool use_conv8(const LayerDesc& d) {
  if (d.is_fc || d.log2S != 3 || d.s_real || !d.scope2.empty()) return false;
  return d.k == 5 || d.k == 3;
}
// ... and on conv4n_kernel (conv4n.hip): the k^3 taps (k = 2 .. 5) on the 4^3 volume (not the 3^3 grid embedded in 4^3)
bool use_conv4(const LayerDesc& d, int dtype) {
  if (d.is_fc || d.log2S != 2 || d.s_real || !d.scope2.empty() || d.k < 2 || d.k > 5) return false;
  return act_planes(dtype) == 1 || !(d.k & 1);   // pair modes: the even kernels only (conv4n.hip: launch_conv4n_dt)
}

// Error attribution in the pair modes (scripts/exp_attribution.py), ONLY in builds made with -DNESTI_ATTRIBUTION (the product
// library has no such switch): NESTI_X3_PLAIN = "regex,regex,..." -- a layer whose scope matches drops the hi * W_lo product
// (its W_lo weights are packed as zeros: the layer then sees its weights rounded to 16 bits).  Round 3's full sweep
// (profiles/r03_attribution_sweep.txt) also switched off lo * W_hi per layer; that needed the three-plane layout
// [hi | lo | hi] x [W_hi ; W_hi ; W_lo] of commit 6f246d7 and is not available in the two-plane layout.
#ifdef NESTI_ATTRIBUTION
int x3_drop_mask(const LayerDesc& d) {
  const char* e = getenv("NESTI_X3_PLAIN");
  if (!e || !*e) return 0;
  std::string spec(e);
  size_t pos = 0;
  while (pos <= spec.size()) {
    size_t end = spec.find(',', pos);
    if (end == std::string::npos) end = spec.size();
    const std::string item = spec.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    try {
      const std::regex re(item);
      if (std::regex_search(d.scope, re) || (!d.scope2.empty() && std::regex_search(d.scope2, re))) {
        fprintf(stderr, "libnesti_hip (attribution build): layer %s packed WITHOUT its W_lo plane\n", d.scope.c_str());
        return 2;
      }
    } catch (const std::regex_error&) {
      fprintf(stderr, "libnesti_hip (attribution build): bad regex '%s' in NESTI_X3_PLAIN\n", item.c_str());
    }
  }
  return 0;
}
#else
inline int x3_drop_mask(const LayerDesc&) { return 0; }
#endif

// OCP e4m3 (1-4-3, bias 7, no infinities, largest finite 448) of a float, round to nearest even, saturating; subnormals kept
uint8_t host_f32_to_e4m3(float f) {
  const uint8_t sign = std::signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (!(a == a)) return 0x7f;
  a = std::min(a, 448.f);
  if (a == 0.f) return sign;
  int e;
  (void)frexpf(a, &e);                       // a = m 2^e, m in [0.5, 1)
  e = std::max(e - 1, -6);                   // exponent of the leading bit, not below the smallest normal's
  const float quantum = ldexpf(1.f, e - 3);
  float q = nearbyintf(a / quantum);         // default rounding mode: to nearest, ties to even
  float v = q * quantum;
  if (v == 0.f) return sign;
  if (v < ldexpf(1.f, -6)) return (uint8_t)(sign | (uint8_t)nearbyintf(v / ldexpf(1.f, -9)));   // subnormal: exponent field 0
  int e2;
  (void)frexpf(v, &e2);
  e2 -= 1;
  const int mant = (int)nearbyintf(v / ldexpf(1.f, e2 - 3)) - 8;
  return (uint8_t)(sign | ((e2 + 7) << 3) | mant);
}

// The FP8 cross-term packing of one k^3 tap layer at 8^3 (conv8n.hip X8; NESTI_F16X8 / NESTI_F16X8C): per (column pair, 16-channel
// chunk, tap) 64 rows x 64 B = [W_hi f16 k0..15 | W_hi8 k0..15 | W_lo8 k0..15] of the SAME scaled weights the pair packing holds
// (W 2^e, |W| 2^e < 2^14): W_hi8 = e4m3(W_hi 2^sb), W_lo8 = e4m3((W - W_hi) 2^(sb + 11)) with sb = -6 (both below 256).
// e2m3 (1-2-3, bias 1, largest finite 7.5) code of a non-negative magnitude already divided by its block scale: round to nearest even,
// saturating.  The code is monotone in the value and the grid is piecewise uniform: [0, 1) step 1/8 (subnormals), [1, 2) 1/8, [2, 4) 1/4,
// [4, 7.5] 1/2; a value that rounds up to the next binade's first point gets that point's code.
uint8_t host_mag_to_e2m3(float a) {
  if (!(a == a)) return 31;
  int code;
  if (a < 2.f) code = (int)nearbyintf(a * 8.f);                    // 0 .. 16 (16 = 2.0)
  else if (a < 4.f) code = 16 + (int)nearbyintf((a - 2.f) * 4.f);    // .. 24 (= 4.0)
  else code = 24 + (int)nearbyintf((std::min(a, 8.f) - 4.f) * 2.f);
  return (uint8_t)std::min(code, 31);
}
inline uint8_t host_f32_to_e2m3(float f, float inv_scale) {
  return (uint8_t)(host_mag_to_e2m3(fabsf(f) * inv_scale) | (std::signbit(f) ? 32 : 0));
}

// fmt == 6: the block-scaled FP6 form of the same rows (conv8n.hip X6; kernels.h: ConvParams::x8_fmt): the 32 bytes that hold
// [W_hi8 | W_lo8] hold instead 32 e2m3 elements -- slot 2i = W_hi[i] / s, slot 2i + 1 = W_lo[i] 2^11 / s of the chunk's 16 input channels
// (the order the producer's conversion instruction writes [lo | hi] activations in, so that slot products are lo W_hi and hi W_lo) -- and
// in byte 24 the E8M0 code of s 2^-11 (s = 2^(E - 2), E = exponent of the chunk's largest |W_hi|; the 2^-11 undoes BOTH 2^11 pre-scales,
// the activations' and the weights', since every product carries exactly one of them).
int pack_layer_x8(const LayerDesc& d, const TensorTable& tt, PackedLayer* pl, int fmt = 8) {
  if (!use_conv8(d) || !d.scope2.empty()) NESTI_FAIL("internal: pack_layer_x8 is for the k^3 tap layers at 8^3");
  Folded f;
  if (fold_layer(d, d.scope, tt, &f)) return 1;
  const int lo = (d.k - 1) / 2;
  pl->n_taps = 0;
  std::vector<int> tap_widx;
  for (int a = 0; a < d.k; ++a)
    for (int bb = 0; bb < d.k; ++bb)
      for (int c = 0; c < d.k; ++c) {
        pl->tap[pl->n_taps][0] = (int8_t)(a - lo); pl->tap[pl->n_taps][1] = (int8_t)(bb - lo);
        pl->tap[pl->n_taps][2] = (int8_t)(c - lo); pl->tap[pl->n_taps][3] = 0;
        tap_widx.push_back((a * d.k + bb) * d.k + c);
        ++pl->n_taps;
      }
  pl->kind = 2; pl->x3n = false; pl->x8 = true; pl->TN = 64;
  if (d.Cout_p % 64 || d.Cin_p % kSplitGroup) NESTI_FAIL("internal: pack_layer_x8 needs 64-aligned channel counts");
  pl->n_tiles = d.Cout_p / 64; pl->split_tile = pl->n_tiles;
  constexpr int chunk_ch = 16;
  pl->n_chunks = d.Cin_p / chunk_ch;
  float wmax = 0.f;
  for (size_t t = 0; t < tap_widx.size(); ++t) {
    const float* wt = f.w + (size_t)tap_widx[t] * d.cin * d.cout;
    for (int c = 0; c < d.cin; ++c)
      for (int n = 0; n < d.cout; ++n) wmax = std::max(wmax, fabsf(wt[(size_t)c * d.cout + n] * f.scale[n]));
  }
  int e = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    (void)frexpf(wmax, &e);
    e = std::min(24, std::max(-8, 14 - e));      // as pack_layer: wmax 2^e in [2^13, 2^14)
  }
  const float wmul = ldexpf(1.0f, e);
  pl->acc_scale = ldexpf(1.0f, -e);
  pl->x8_sb = -6;
  const float mul_hi8 = ldexpf(1.f, pl->x8_sb), mul_lo8 = ldexpf(1.f, pl->x8_sb + 11);
  std::vector<int> inv(d.Cin_p, -1);
  for (int c = 0; c < d.cin; ++c) inv[d.in_pos[c]] = c;
  const size_t tile_bytes = (size_t)64 * 64;
  const size_t total = (size_t)pl->n_tiles * pl->n_chunks * pl->n_taps * tile_bytes;
  std::vector<unsigned char> host(total, 0);
  for (int nt = 0; nt < pl->n_tiles; ++nt)
    for (int ch = 0; ch < pl->n_chunks; ++ch)
      for (int t = 0; t < pl->n_taps; ++t) {
        unsigned char* tile = host.data() + (((size_t)nt * pl->n_chunks + ch) * pl->n_taps + t) * tile_bytes;
        const float* wt = f.w + (size_t)tap_widx[t] * d.cin * d.cout;
        float whi[16][64] = {}, wlo[16][64] = {};          // fmt 6: the chunk's pair split, per column
        for (int kc = 0; kc < chunk_ch; ++kc) {
          const int cr = inv[ch * chunk_ch + kc];
          if (cr < 0) continue;
          const float* wrow = wt + (size_t)cr * d.cout;
          for (int nl = 0; nl < 64; ++nl) {
            const int n = nt * 64 + nl;
            if (n >= d.cout) break;
            const float v = wrow[n] * f.scale[n] * wmul;
            const uint16_t h = host_f32_to_f16(v);
            const float hf = host_f16_to_f32(h);
            const int key = (nl >> 2) & 3;               // conv8n_kernel's weight-row swizzle
            unsigned char* row = tile + (size_t)nl * 64;
            memcpy(row + (((kc >> 3) ^ key) << 4) + (kc & 7) * 2, &h, 2);
            if (fmt == 6) { whi[kc][nl] = hf; wlo[kc][nl] = (v - hf) * 2048.f; continue; }
            row[((2 ^ key) << 4) + kc] = host_f32_to_e4m3(hf * mul_hi8);
            row[((3 ^ key) << 4) + kc] = host_f32_to_e4m3((v - hf) * mul_lo8);
          }
        }
        if (fmt == 6)
          for (int nl = 0; nl < 64; ++nl) {
            float amax = 0.f;
            for (int kc = 0; kc < chunk_ch; ++kc) amax = std::max(amax, fabsf(whi[kc][nl]));
            if (!(amax > 0.f) || !std::isfinite(amax)) continue;          // an all-zero (padding) block: codes 0, scale byte 0
            int e;
            (void)frexpf(amax, &e);                                        // amax = m 2^e, m in [0.5, 1): leading exponent e - 1
            const int sexp = e - 1 - 2;                                    // s = 2^sexp: the largest element lands in [4, 8)
            const float inv_s = ldexpf(1.f, -sexp);
            unsigned char blk[32] = {};
            for (int kc = 0; kc < chunk_ch; ++kc) {
              const unsigned c2[2] = {host_f32_to_e2m3(whi[kc][nl], inv_s), host_f32_to_e2m3(wlo[kc][nl], inv_s)};
              for (int j = 0; j < 2; ++j) {
                const int pos = 6 * (2 * kc + j);
                const unsigned w = c2[j] << (pos & 7);
                blk[pos >> 3] |= (unsigned char)w;
                blk[(pos >> 3) + 1] |= (unsigned char)(w >> 8);
              }
            }
            const int sbyte = sexp - 11 + 127;
            if (sbyte < 1 || sbyte > 254) NESTI_FAIL("internal: FP6 weight block scale out of the E8M0 range");
            blk[24] = (unsigned char)sbyte;
            const int key = (nl >> 2) & 3;
            unsigned char* row = tile + (size_t)nl * 64;
            memcpy(row + ((2 ^ key) << 4), blk, 16);
            memcpy(row + ((3 ^ key) << 4), blk + 16, 16);
          }
      }
  std::vector<float> bias_p((size_t)pl->n_tiles * 64, 0.f);
  for (int n = 0; n < d.cout; ++n) bias_p[n] = f.bias[n];
  NESTI_CHECK_HIP(hipMalloc(&pl->wpk, total));
  NESTI_CHECK_HIP(hipMemcpy(pl->wpk, host.data(), total, hipMemcpyHostToDevice));
  NESTI_CHECK_HIP(hipMalloc((void**)&pl->bias, bias_p.size() * sizeof(float)));
  NESTI_CHECK_HIP(hipMemcpy(pl->bias, bias_p.data(), bias_p.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

// Power-of-two pre-scale of the e4m3 activation planes a block's conv1 writes for the FP8 cross terms: hi8 = e4m3(v 2^sc) must stay
// below the format's 448.  A data-free bound from the layer's own batch-norm: after tf.nn.batch_normalization the pre-activation of
// channel n is beta_n + gamma_n z with z ~ N(0, 1) on the data the statistics were taken from, so |v| <= |beta_n| + 8 |gamma_n| but for
// 8-sigma events; 2^sc brings that bound into (128, 256].  A larger value saturates and loses only its own cross terms.
int x8_activation_exponent(const LayerDesc& d, const TensorTable& tt) {
  float amax = 16.f;
  const nesti_tensor_t* beta = tt.get(d.scope + "/bn/beta");
  const nesti_tensor_t* gamma = tt.get(d.scope + "/bn/gamma");
  if (d.bn && beta && gamma && beta->data && gamma->data) {
    amax = 0.f;
    for (int n = 0; n < d.cout; ++n) amax = std::max(amax, fabsf(beta->data[n]) + 8.f * fabsf(gamma->data[n]));
  }
  if (!(amax > 0.f) || !std::isfinite(amax)) amax = 16.f;
  int e;
  (void)frexpf(amax, &e);                        // amax = m 2^e, m in [0.5, 1): amax <= 2^e
  return std::min(20, std::max(-8, 8 - e));
}

int pack_layer(const LayerDesc& d, const TensorTable& tt, int dtype, PackedLayer* pl) {
  const int n_parts = d.scope2.empty() ? 1 : 2;
  Folded parts[2];
  if (fold_layer(d, d.scope, tt, &parts[0])) return 1;
  if (n_parts == 2 && fold_layer(d, d.scope2, tt, &parts[1])) return 1;
  const int part_p = d.Cout_p / n_parts;   // padded width of one part
  const int S = d.s_real ? d.s_real : (1 << d.log2S);
  const int lo = (d.k - 1) / 2;   // TF SAME, stride 1
  pl->n_taps = 0;
  std::vector<int> tap_widx;
  for (int a = 0; a < d.k; ++a)
    for (int bb = 0; bb < d.k; ++bb)
      for (int c = 0; c < d.k; ++c) {
        const int dz = a - lo, dy = bb - lo, dx = c - lo;
        if (abs(dz) >= S || abs(dy) >= S || abs(dx) >= S) continue;   // never lands inside the volume
        pl->tap[pl->n_taps][0] = (int8_t)dz; pl->tap[pl->n_taps][1] = (int8_t)dy;
        pl->tap[pl->n_taps][2] = (int8_t)dx; pl->tap[pl->n_taps][3] = 0;
        tap_widx.push_back((a * d.k + bb) * d.k + c);
        ++pl->n_taps;
      }
  const size_t esz = dtype_size(dtype);
  // NESTI_BF16X3 / NESTI_F16X3 (common.h): a K chunk of a packed weight row is [W_hi | W_lo] for half as many channels as the
  // plain chunk holds (the kernels' pair K loop multiplies hi*W_hi + lo*W_hi + hi*W_lo from it: conv.hip / conv8n.hip, X3)
  const int planes = act_planes(dtype);
  const int drop = planes > 1 ? x3_drop_mask(d) : 0;
  pl->kind = use_conv8(d) ? 2 : use_conv4(d, dtype) ? 3 : 0;
  if (pl->kind >= 2 && d.Cout_p % 64) NESTI_FAIL("internal: conv8n_kernel / conv4n_kernel need 64-column tiles");
  pl->x3n = planes > 1;
  const int K_phys = d.Cin_p * planes;
  const int row_bytes = pl->kind >= 1 ? 64 : kRowBytes;   // bytes of one K chunk of one row
  const int KC = row_bytes / (int)esz;
  const int chunk_ch = KC / planes;                       // input channels per K chunk
  pl->TN = pl->kind >= 2 ? 64 : (part_p % 128 == 0) ? 128 : 64;   // a tile never straddles the two parts
  pl->n_tiles = d.Cout_p / pl->TN;
  pl->split_tile = part_p / pl->TN * (n_parts == 2 ? 1 : n_parts);
  if (n_parts == 1) pl->split_tile = pl->n_tiles;
  pl->n_chunks = K_phys / KC;
  if (K_phys % KC || d.Cin_p % kSplitGroup) NESTI_FAIL("internal: Cin_p not a multiple of the K chunk");
  if (n_parts == 2 && pl->n_taps != 1) NESTI_FAIL("internal: fused layers must be 1x1x1");
  // NESTI_F16X3: one power-of-two scale per layer brings the largest folded weight to ~2^14, so that the lo halves of the
  // weight pairs (2^-12 of a weight) are normal f16 numbers; the epilogue multiplies the accumulators by 2^-s (exact)
  float wmul = 1.0f;
  pl->acc_scale = 1.0f;
  if (dtype == NESTI_F16X3) {
    float wmax = 0.f;
    for (int part = 0; part < n_parts; ++part) {
      const Folded& f = parts[part];
      for (size_t t = 0; t < tap_widx.size(); ++t) {
        const float* wt = f.w + (size_t)tap_widx[t] * d.cin * d.cout;
        for (int c = 0; c < d.cin; ++c)
          for (int n = 0; n < d.cout; ++n) wmax = std::max(wmax, fabsf(wt[(size_t)c * d.cout + n] * f.scale[n]));
      }
    }
    int e = 0;
    if (wmax > 0.f && std::isfinite(wmax)) {
      (void)frexpf(wmax, &e);                  // wmax = m 2^e, m in [0.5, 1)
      e = std::min(24, std::max(-8, 14 - e));  // wmax 2^s in [2^13, 2^14)
    }
    wmul = ldexpf(1.0f, e);
    pl->acc_scale = ldexpf(1.0f, -e);
  }
  std::vector<int> inv(d.Cin_p, -1);
  for (int c = 0; c < d.cin; ++c) inv[d.in_pos[c]] = c;
  const size_t tile_bytes = (size_t)pl->TN * row_bytes;
  const size_t total = (size_t)pl->n_tiles * pl->n_chunks * pl->n_taps * tile_bytes;
  std::vector<unsigned char> host(total, 0);
  const int per_slot = 16 / (int)esz;
  for (int nt = 0; nt < pl->n_tiles; ++nt) {
    const int part = (nt * pl->TN) / part_p;
    const int n_base = nt * pl->TN - part * part_p;    // first real channel of this tile within its part
    const Folded& f = parts[part];
    for (int ch = 0; ch < pl->n_chunks; ++ch)
      for (int t = 0; t < pl->n_taps; ++t) {
        unsigned char* tile = host.data() + (((size_t)nt * pl->n_chunks + ch) * pl->n_taps + t) * tile_bytes;
        const float* wt = f.w + (size_t)tap_widx[t] * d.cin * d.cout;
        for (int kc = 0; kc < KC; ++kc) {
          // which half of the row this K position is (0: W_hi, 1: W_lo) and the padded input channel it multiplies
          const int plane = kc / chunk_ch;
          const int cr = inv[ch * chunk_ch + kc % chunk_ch];
          if (cr < 0) continue;
          if (plane == 1 && (drop & 2)) continue;
          const float* wrow = wt + (size_t)cr * d.cout;
          const int slot = kc / per_slot, within = kc % per_slot;
          for (int nl = 0; nl < pl->TN; ++nl) {
            const int n = n_base + nl;
            if (n >= d.cout) break;
            const float v = wrow[n] * f.scale[n] * wmul;
            // the kernels' LDS image: row nl, 16-B slot XOR-swizzled (conv8n_kernel: 64-B rows, key (row >> 2) & 3;
            // conv4n_kernel: 64-B rows, key {0, 2, 3, 1}[(row >> 2) & 3])
            const int key64 = pl->kind == 3 ? (0x78 >> (2 * ((nl >> 2) & 3))) & 3 : (nl >> 2) & 3;
            unsigned char* dst = pl->kind >= 1
                ? tile + (size_t)nl * 64 + ((slot ^ key64) << 4) + within * esz
                : tile + (size_t)nl * kRowBytes + ((slot ^ ((nl >> 1) & 7)) << 4) + within * esz;
            if (dtype == NESTI_F32) memcpy(dst, &v, 4);
            else if (planes > 1) {
              const bool b16 = dtype == NESTI_BF16X3;
              uint16_t h = b16 ? host_f32_to_bf16(v) : host_f32_to_f16(v);
              if (plane == 1) {                              // W_lo = rne(W - W_hi)
                float hf;
                if (b16) { const uint32_t hb = (uint32_t)h << 16; memcpy(&hf, &hb, 4); } else hf = host_f16_to_f32(h);
                h = b16 ? host_f32_to_bf16(v - hf) : host_f32_to_f16(v - hf);
              }
              memcpy(dst, &h, 2);
            } else {
              const uint16_t h = (dtype == NESTI_BF16) ? host_f32_to_bf16(v) : host_f32_to_f16(v);
              memcpy(dst, &h, 2);
            }
          }
        }
      }
  }
  std::vector<float> bias_p((size_t)pl->n_tiles * pl->TN, 0.f);
  for (int part = 0; part < n_parts; ++part)
    for (int n = 0; n < d.cout; ++n) bias_p[(size_t)part * part_p + n] = parts[part].bias[n];
  NESTI_CHECK_HIP(hipMalloc(&pl->wpk, total));
  NESTI_CHECK_HIP(hipMemcpy(pl->wpk, host.data(), total, hipMemcpyHostToDevice));
  NESTI_CHECK_HIP(hipMalloc((void**)&pl->bias, bias_p.size() * sizeof(float)));
  NESTI_CHECK_HIP(hipMemcpy(pl->bias, bias_p.data(), bias_p.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

// ------------------------------------------------------------------------------------------
// workspace planning and tower execution
// ------------------------------------------------------------------------------------------
size_t buf_bytes(const BufSpec& b, int NB, int dtype) {
  const size_t e = b.aux8 ? 2 : b.f32 ? 4 : dtype_size(dtype) * act_planes(dtype);
  return align_up(((size_t)NB << (3 * b.log2S)) * b.C * e, 256);
}
// Workspace placement of a tower's buffers: a buffer lives from the first launch that writes it to the last launch that
// reads it (the tower's output until the end); buffers whose lifetimes do not overlap share memory (first fit), which
// brings the gating tower from 2.5 to 1.6 MB per query in 16-bit and lets one library batch cover a 100k-point cloud.
struct Placement {
  std::vector<size_t> off;   // per buffer (index 0 = the external MuPS tensor: unused)
  size_t total = 0;
};
Placement place_tower(const Tower& T, int NB, int dtype) {
  const int n = (int)T.bufs.size(), n_ops = (int)T.ops.size();
  std::vector<int> first(n, 1 << 30), last(n, -1);
  for (int k = 0; k < n_ops; ++k) {
    const Op& op = T.ops[k];
    for (int b : {op.out_buf, op.mp_buf, op.aux_out_buf})
      if (b >= 1) { first[b] = std::min(first[b], k); last[b] = std::max(last[b], k); }
    for (int b : {op.in_buf, op.aux_in_buf})
      if (b >= 1) last[b] = std::max(last[b], k);
  }
  if (T.out_buf >= 1) last[T.out_buf] = n_ops;
  std::vector<int> order;
  for (int i = 1; i < n; ++i) if (last[i] >= 0) order.push_back(i);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return first[a] != first[b] ? first[a] < first[b] : a < b; });
  struct Block { size_t off, size; int until; };
  std::vector<Block> live, freeb;
  Placement P;
  P.off.assign(n, 0);
  for (int i : order) {
    // release what is no longer read, coalescing neighbours
    for (size_t j = 0; j < live.size();) {
      if (live[j].until < first[i]) { freeb.push_back(live[j]); live.erase(live.begin() + j); } else ++j;
    }
    std::sort(freeb.begin(), freeb.end(), [](const Block& a, const Block& b) { return a.off < b.off; });
    for (size_t j = 0; j + 1 < freeb.size();) {
      if (freeb[j].off + freeb[j].size == freeb[j + 1].off) { freeb[j].size += freeb[j + 1].size; freeb.erase(freeb.begin() + j + 1); } else ++j;
    }
    const size_t need = buf_bytes(T.bufs[i], NB, dtype);
    size_t at = (size_t)-1;
    for (size_t j = 0; j < freeb.size(); ++j) {
      if (freeb[j].size >= need) {
        at = freeb[j].off;
        freeb[j].off += need; freeb[j].size -= need;
        if (freeb[j].size == 0) freeb.erase(freeb.begin() + j);
        break;
      }
    }
    if (at == (size_t)-1) {
      // grow at the top; a free block that ends at the top is extended instead of wasted
      if (!freeb.empty() && freeb.back().off + freeb.back().size == P.total) {
        at = freeb.back().off;
        P.total = at + need;
        freeb.pop_back();
      } else {
        at = P.total;
        P.total += need;
      }
    }
    P.off[i] = at;
    live.push_back({at, need, last[i]});
  }
  return P;
}
size_t tower_bytes(const Tower& T, int NB, int dtype) { return place_tower(T, NB, dtype).total; }

// Which k^3 layers use the Latin-square tile layout (kernels.h: ConvParams::remap): those where a 32-row tile can
// fall entirely on padding -- 5^3 taps at 8^3 (|d| = 2 clears a y/z pair) and every multi-tap layer at 4^3.  3^3 at
// 8^3 only ever clears single planes, which no 32-row tile shape can balance over four SIMDs.
int conv_remap(int k, int log2S, int n_taps) {
  if (n_taps <= 1) return 0;
  if (log2S == 1) return 2;              // 2^3: single-voxel tiles of 32 points (conv.hip: remap == 2)
  if (log2S != 2 && log2S != 3) return 0;
  return (log2S == 2 || k >= 4) ? 1 : 0;
}

// which NESTI_PROF_* conv category a layer's launch is booked under
int conv_category(const LayerDesc& d, const PackedLayer& pl) {
  if (pl.kind == 2) return d.k == 5 ? NESTI_PROF_CONV8_K5 : NESTI_PROF_CONV8_K3;
  return pl.n_taps > 1 ? NESTI_PROF_TAPS : NESTI_PROF_ONE_BY_ONE;
}

constexpr int kGateMixBit = 8;   // packed_mix entries of the gating net: RunCtx::mix bit 8 switches all of them

struct RunCtx {
  const nesti_model* m;
  int NB;                        // capacity (points)
  const int32_t* npoints_ptr;    // device-side live count or NULL
  const int32_t* point_index;    // gather for reads of bufs 0/1, or NULL
  hipStream_t stream;
  bool fast = false;             // NESTI_F16X3C filter pass: this tower runs in plain f16 on rc.m->packed_fast while the
                                 // MuPS tensor it reads keeps the model's pair layout (only the hi plane is read)
  int mix = 0;                   // expert towers in a pair mode: the layers whose packed_mix bit is set here run single-product
  bool zero_lo = false;          // experiment (gate_mix == 2): every layer writes its outputs rounded to 16 bits (lo plane = 0)
  int walk = 0;                  // this pass over a device-side list is probably empty (a later round, a widening pass): its conv
                                 // launches use small walking grids (kernels.h: ConvParams::walk)
  int x8 = 0;                    // expert towers of an NESTI_F16X8 / NESTI_F16X8C model: nesti_model::x8_mask (which tap layers run the
                                 // FP8 cross-term loop; their block's conv1 then also writes the e4m3 planes)
};

int run_tower(const RunCtx& rc, const Tower& T, const void* X0, unsigned char* ws, size_t ws_bytes, float** out) {
  const int dtype = rc.fast ? NESTI_F16 : rc.m->dtype;
  const int x0_planes = act_planes(rc.m->dtype);
  std::vector<unsigned char*> ptr(T.bufs.size(), nullptr);
  ptr[0] = (unsigned char*)X0;
  const Placement P = place_tower(T, rc.NB, dtype);
  for (size_t i = 1; i < T.bufs.size(); ++i) ptr[i] = ws + P.off[i];
  if (P.total > ws_bytes) NESTI_FAIL("workspace too small for this batch");
  for (const Op& op : T.ops) {
    const bool ext_in = op.in_buf < 1;
    if (op.kind == Op::CONV) {
      const LayerDesc& d = rc.m->graph.layers[op.layer];
      const bool mixl = !rc.fast && rc.mix && op.layer < (int)rc.m->packed_mix.size() && rc.m->packed_mix[op.layer].wpk &&
                        ((rc.mix >> rc.m->packed_mix[op.layer].mix_bit) & 1);
      // NESTI_F16X3C filter pass: its one-tap layers (1x1x1 conv1|conv4, FC) multiply the plain-f16 activations by the model's own
      // PAIR-packed weights (conv_igemm_kernel's X2 loop: hi * W_hi + hi * W_lo).  Those layers are fill-bound, so the second product
      // costs ~20 % more weight-tile fill and no matrix-pipe time that shows, and it removes the weight-rounding part of their error:
      // the filter's sigma on a logit difference drops from 0.021 to 0.012 (profiles/r05_gate_medium.txt), the threshold with it
      const bool x2l = rc.fast && rc.m->packed_fast[op.layer].wpk == nullptr;
      const bool x8l = !rc.fast && !mixl && op.aux_in_buf >= 0 && op.x8_bit >= 0 && ((rc.x8 >> op.x8_bit) & 1) &&
                       op.layer < (int)rc.m->packed_x8.size() && rc.m->packed_x8[op.layer].wpk;
      const bool x6 = rc.m->x8_fmt == 6;
      const PackedLayer& pl = x2l ? rc.m->packed[op.layer] : rc.fast ? rc.m->packed_fast[op.layer]
                              : mixl ? rc.m->packed_mix[op.layer] : x8l ? (x6 ? rc.m->packed_x6 : rc.m->packed_x8)[op.layer] : rc.m->packed[op.layer];
      ConvParams p;
      memset(&p, 0, sizeof(p));
      p.in_pair = mixl ? 1 : 0;
      p.x2 = x2l ? 1 : 0;
      p.in = ptr[op.in_buf]; p.out = ptr[op.out_buf]; p.wpk = pl.wpk; p.bias = pl.bias;
      p.npoints_ptr = rc.npoints_ptr; p.point_index = ext_in ? rc.point_index : nullptr;
      p.npoints = rc.NB;
      // pair modes: strides and the input offset are physical (a 64-aligned logical offset x 3), output column
      // offsets stay logical (kernels.h: ConvParams::split); an fp32 output buffer is an ordinary one
      const int planes = act_planes(dtype);
      p.split = planes > 1 ? (rc.zero_lo ? 2 : 1) : 0;
      const int in_planes = ext_in ? x0_planes : planes;
      p.in_cstride = (op.in_cstride ? op.in_cstride : T.bufs[op.in_buf].C) * in_planes; p.in_coff = op.in_coff * in_planes;
      // distance between consecutive K chunks of a PLAIN kernel's input row: 128 B, except in the NESTI_F16X3C filter pass,
      // whose plain-f16 first layer reads the hi plane of each 64-channel group [hi | lo] of the pair-layout MuPS tensor
      p.in_chunk_bytes = kRowBytes * (planes == 1 ? in_planes : 1);
      p.out_cstride = T.bufs[op.out_buf].C * (op.out_f32 ? 1 : planes); p.out_coff = op.out_coff;
      p.n_chunks = pl.n_chunks; p.n_taps = pl.n_taps; p.tap_k = d.k; p.log2S = d.log2S; p.s_real = d.s_real;
      p.relu = d.relu ? 1 : 0; p.out_f32 = op.out_f32 ? 1 : 0; p.acc_scale = pl.acc_scale; p.x3native = (pl.x3n && !x2l) ? 1 : 0;
      const long long rows = (long long)rc.NB << (3 * d.log2S);
      p.m_tiles = pl.kind == 3 ? (rc.NB + 15) / 16 : pl.kind == 2 ? (rc.NB + 3) / 4 : (int)((rows + kTileM - 1) / kTileM);
      p.n_tiles = pl.n_tiles; p.split_tile = pl.split_tile; p.out_coff2 = op.out_coff2; p.pool_k = d.pool_k;
      if (op.mp_buf >= 0) { p.mp_out = ptr[op.mp_buf]; p.mp_cstride = T.bufs[op.mp_buf].C * planes; p.mp_mode = op.mp_mode; p.mp_mode2 = op.mp_mode2; }
      if (x8l) {                                   // consumer: the FP8 cross-term loop on the planes the block's conv1 wrote
        const int sc = rc.m->packed[op.aux_layer].x8_sc;
        p.x8 = 1; p.aux8_in = ptr[op.aux_in_buf]; p.aux8_stride = T.bufs[op.aux_in_buf].C * 2;
        p.x8_scale_a = 127 - (sc + 11); p.x8_scale_b = 127 - pl.x8_sb;
        p.x8_fmt = x6 ? 6 : 8;
      }
      if (!rc.fast && op.aux_out_buf >= 0 && (rc.x8 & op.x8_bits) && !rc.m->packed_x8.empty()) {   // producer
        p.aux8_out = ptr[op.aux_out_buf]; p.aux8_stride = T.bufs[op.aux_out_buf].C * 2;
        p.x8_sc = rc.m->packed[op.layer].x8_sc; p.x8_sa = p.x8_sc + 11;
        p.x8_fmt = x6 ? 6 : 8;
      }
      memcpy(p.tap, pl.tap, sizeof(p.tap));
      p.remap = conv_remap(d.k, d.log2S, pl.n_taps);
      p.walk = rc.walk;
      const int cat = conv_category(d, pl);
      const int tok = prof_begin(cat, rc.stream);
      // (a mixed layer is a plain f16 / bf16 kernel inside a pair-mode tower: kernel_dtype is the same element type either way)
      const int rcv = pl.kind == 2   ? launch_conv8n(p, kernel_dtype(dtype), d.k, rc.stream)
                      : pl.kind == 3 ? launch_conv4n(p, kernel_dtype(dtype), d.k, rc.stream)
                                     : launch_conv(p, kernel_dtype(dtype), pl.TN, rc.stream);
      prof_end(cat, tok, rc.stream);
      if (rcv) return 1;
    } else {
      PoolParams p;
      memset(&p, 0, sizeof(p));
      p.in = ptr[op.in_buf]; p.out = ptr[op.out_buf];
      p.npoints_ptr = rc.npoints_ptr;
      p.npoints = rc.NB;
      const int planes = act_planes(dtype);
      p.split = planes > 1 ? 1 : 0;
      p.in_cstride = T.bufs[op.in_buf].C * planes; p.in_coff = op.in_coff;
      p.out_cstride = T.bufs[op.out_buf].C * planes; p.out_coff = op.out_coff;
      p.C = op.C; p.log2S = op.log2S;
      const int tok = prof_begin(NESTI_PROF_POOL, rc.stream);
      const int rcp = op.kind == Op::MAX3 ? launch_maxpool3s2(p, kernel_dtype(dtype), rc.stream)
                                          : launch_maxpool2(p, kernel_dtype(dtype), rc.stream);
      prof_end(NESTI_PROF_POOL, tok, rc.stream);
      if (rcp) return 1;
    }
  }
  *out = reinterpret_cast<float*>(ptr[T.out_buf]);
  return 0;
}

// channel stride of the MuPS rows the towers read, in elements (pair modes: two planes per 64-channel group)
int mups_stride(const nesti_model* m) { return m->graph.mups_cstride * act_planes(m->dtype); }

// NESTI_F16X3C: the f16x3 gate re-decides the flagged rows `cap` at a time (its workspace is 3x the filter's per row, and
// only a fraction of a batch is flagged): small batches in one round, large ones in quarters
int cascade_cap(int NB) { return NB <= 4096 ? NB : (int)align_up((size_t)(NB + 3) / 4, 256); }
int cascade_rounds(int NB) { return (NB + cascade_cap(NB) - 1) / cascade_cap(NB); }

// A routed expert sees about 1 / E of a batch, so its tower is sized for a quarter of a large batch and run in up to four
// rounds over its routing list (rounds beyond the list's length launch empty grids: ~0.2 % of a 100k batch); the workspace of a
// batch is then set by the gating net alone and a whole 100k-point cloud is one library batch in every mode but f16x3 / f32.
int expert_cap(int NB) { return NB <= 8192 ? NB : (int)align_up((size_t)(NB + 3) / 4, 256); }
// rows of ONE expert the conditioning guard can re-evaluate per pass (a fraction of a per cent of a batch are flagged at all)
int guard_cap(int NB) { return std::min(expert_cap(NB), 2048); }

size_t max_tower_bytes(const nesti_model* m, int NB) {
  size_t t = m->cascade ? std::max(tower_bytes(m->graph.gate, NB, NESTI_F16), tower_bytes(m->graph.gate, cascade_cap(NB), m->dtype))
                        : tower_bytes(m->graph.gate, NB, m->dtype);
  for (const Tower& e : m->graph.experts) t = std::max(t, tower_bytes(e, expert_cap(NB), m->dtype));
  return t;
}

struct WsLayout {
  size_t x0, probs, expert, counts, lists, ecounts, glist, keep, flags, fcounts, tower, total;
};
WsLayout ws_layout(const nesti_model* m, int NB) {
  WsLayout L;
  const size_t act = align_up(((size_t)NB << (3 * m->graph.gate_x0_log2S())) * mups_stride(m) * dtype_size(m->dtype), 256);
  size_t o = 0;
  L.x0 = o; o += act;
  L.probs = o; o += align_up((size_t)NB * NESTI_MAX_EXPERTS * 4, 256);
  L.expert = o; o += align_up((size_t)NB * 4, 256);
  L.counts = o; o += 256;
  L.lists = o; o += align_up((size_t)NB * NESTI_MAX_EXPERTS * 4, 256);
  L.ecounts = o; o += 1024;          // [E][rounds] rows of each expert round; words 128-135 / 140-141: the conditioning guard's list lengths and |n| band
  L.glist = o;
  if (m->graph.x8) o += align_up((size_t)NESTI_MAX_EXPERTS * guard_cap(NB) * 4, 256);   // the conditioning guard's row lists, one per expert
  L.keep = L.flags = L.fcounts = o;
  if (m->cascade) {   // the f16 gate's logits, the flag list, [flag count | per-round counts]
    L.keep = o; o += align_up((size_t)NB * NESTI_MAX_EXPERTS * 4, 256);
    L.flags = o; o += align_up((size_t)NB * 4, 256);
    L.fcounts = o; o += 512;          // kernels.h: the two-stage gate's per-call counters
  }
  L.tower = o; o += max_tower_bytes(m, NB);
  L.total = o;
  return L;
}

// NESTI_F16X3C (include/nesti_hip.h): the gating net in plain f16 over the batch, then in f16x3 over the rows whose f16
// top-2 margin is below tau (gathered through the flag list like a routed expert gathers its rows), `cap` rows per round;
// the routing lists are built from the final arg-max.  ws = the forward workspace (ws_layout), capacity NB >= B.
int gate_cascade(const nesti_model* m, const void* X0, int B, unsigned char* ws, const WsLayout& L, int NB, float* probs,
                 int32_t* expert, int32_t* counts, int32_t* lists, hipStream_t stream) {
  const int E = m->graph.cfg.n_experts;
  unsigned char* tower_ws = ws + L.tower;
  const size_t tower_bytes_ = L.total - L.tower;
  float* keep = (float*)(ws + L.keep);
  int32_t* flag_list = (int32_t*)(ws + L.flags);
  int32_t* fcounts = (int32_t*)(ws + L.fcounts);           // kernels.h: kRoundCountsOff, kTauEffOff, kWidenCountOff, kWidenRoundsOff
  const int lstride = m->graph.gate.bufs[m->graph.gate.out_buf].C;
  float* logits = nullptr;
  RunCtx fast{m, B, nullptr, nullptr, stream, /*fast=*/true};
  prof_phase(NESTI_PHASE_GATE);
  if (run_tower(fast, m->graph.gate, X0, tower_ws, tower_bytes_, &logits)) return 1;
  prof_phase(NESTI_PHASE_RECHECK);
  const int cap = std::min(cascade_cap(NB), B), rounds = (B + cap - 1) / cap;
  if (launch_gate_flag(logits, lstride, B, E, m->tau, NESTI_GATE_WIDEN, probs, expert, keep, fcounts, flag_list, cap, rounds,
                       m->cstat, stream))
    return 1;
  // pass 0: the rows below the call's threshold; passes 1 .. NESTI_GATE_WIDEN_PASSES: widening passes -- the band between the
  // threshold reached so far and NESTI_GATE_WIDEN x the largest error measured up to the start of the pass, so an error first
  // seen inside a widening pass is covered by the next one of the SAME call (normally every pass is empty: its launches find a
  // zero row count on the device and return; no host synchronisation, so the whole call stays graph-capturable)
  for (int pass = 0; pass <= NESTI_GATE_WIDEN_PASSES; ++pass) {
    if (pass >= 1 && launch_gate_widen(keep, B, E, NESTI_GATE_WIDEN, fcounts, flag_list, cap, rounds, m->cstat, stream)) return 1;
    const int32_t* round_counts = fcounts + (pass == 0 ? kRoundCountsOff : kWidenRoundsOff);
    for (int r = 0; r < rounds; ++r) {
      RunCtx exact{m, cap, round_counts + r, flag_list + (size_t)r * cap, stream};
      exact.walk = pass >= 1 ? NESTI_GATE_WIDEN_WALK_GRID : r >= 1 ? 1 : 0;   // round 0 of pass 0 holds the flagged rows; everything after it is normally empty
      if (run_tower(exact, m->graph.gate, X0, tower_ws, tower_bytes_, &logits)) return 1;
      if (launch_gate_recheck(logits, lstride, flag_list + (size_t)r * cap, round_counts + r, cap, E, keep, probs, expert,
                              m->cstat, stream))
        return 1;
    }
  }
  if (counts) return launch_route(expert, B, E, counts, lists, stream);
  return 0;
}

int gate_impl(const nesti_model* m, const void* X0, int B, unsigned char* tower_ws, size_t tower_bytes_,
              float* probs, int32_t* expert, int32_t* counts, int32_t* lists, hipStream_t stream) {
  RunCtx rc{m, B, nullptr, nullptr, stream, false, m->gate_mix ? (1 << kGateMixBit) : 0};
  rc.zero_lo = m->gate_mix == 2;
  float* logits = nullptr;
  prof_phase(NESTI_PHASE_GATE);
  if (run_tower(rc, m->graph.gate, X0, tower_ws, tower_bytes_, &logits)) return 1;
  const int lstride = m->graph.gate.bufs[m->graph.gate.out_buf].C;
  if (m->graph.cfg.arch == NESTI_ARCH_SWITCH)   // noise_est < 0.015 -> small, else large (models/ms_sw_n_est.py:80-82)
    return launch_switch_finish(logits, lstride, B, 0.015f, probs, expert, counts, lists, stream);
  return launch_gate_finish(logits, lstride, B, m->graph.cfg.n_experts, probs, expert, counts, lists, stream);
}

// NB = the batch capacity the workspace was laid out for (ws_layout); ecounts = its per-(expert, round) counter block; glist = the
// conditioning guard's row list (x8 models, top-1 routing)
constexpr int kGuardCountOff = 128, kGuardSlotOff = 140;      // int32 words of the ecounts block: [E] list lengths, the |n| band
int experts_impl(const nesti_model* m, const void* X0, int B, int NB, unsigned char* tower_ws, size_t tower_bytes_,
                 const int32_t* counts, const int32_t* lists, int32_t* ecounts, int32_t* glist, float* normals, hipStream_t stream) {
  const int E = m->graph.cfg.n_experts;
  const int cap = std::min(expert_cap(NB), B), rounds = (B + cap - 1) / cap;
  prof_phase(NESTI_PHASE_EXPERTS);
  if (counts && launch_round_counts(counts, E, cap, rounds, ecounts, stream)) return 1;
  const size_t x0_row = ((size_t)1 << (3 * m->graph.gate_x0_log2S())) * mups_stride(m) * dtype_size(m->dtype);   // one query's MuPS rows
  // the conditioning guard of the FP8 cross-term layers (pool.hip): once expert e's rows are written, those whose |n| falls inside the
  // pass's band go through the SAME tower in f16x3 proper, which replaces them and measures |dn|; a second pass covers the band a
  // larger measurement of THIS call may have opened (normally empty).  A guard tower sees a handful of rows, so it is latency-bound
  // (~3 ms: one workgroup walks a layer's whole K loop): in the first pass expert e's guard runs on the model's auxiliary stream, in its
  // own slice of the tower workspace, while the caller's stream goes on with expert e + 1
  const bool guard = counts && glist && m->x8_mask && m->gstat && m->x8_guard_thr >= 0.f;
  const int gcap = std::min(guard_cap(NB), B);
  float* gslot = guard ? reinterpret_cast<float*>(ecounts + kGuardSlotOff) : nullptr;
  int32_t* gcount = guard ? ecounts + kGuardCountOff : nullptr;                 // [E]
  const float gscale = NESTI_X8_GUARD_WIDEN / sqrtf(2.f * NESTI_X8_GUARD_BAR);
  size_t main_bytes = 0, guard_bytes = 0;
  for (int e = 0; e < E && guard; ++e) {
    main_bytes = std::max(main_bytes, align_up(tower_bytes(m->graph.experts[e], cap, m->dtype), 256));
    guard_bytes = std::max(guard_bytes, tower_bytes(m->graph.experts[e], gcap, m->dtype));
  }
  // (a stream that is being captured into a hipGraph keeps everything on itself: the guard then runs on the caller's stream, one
  // tower after the other, like it does when the two workspace slices do not fit)
  nesti_model::GuardLane* lane = guard ? m->guard_lane(stream) : nullptr;
  const bool side = guard && lane->gstream && main_bytes + guard_bytes <= tower_bytes_ && !prof_capturing(stream);
  std::unique_lock<std::mutex> glk;
  if (side) glk = std::unique_lock<std::mutex>(lane->gmu);   // the lane's auxiliary stream and events: one call enqueues on them at a time
  auto guard_expert = [&](int e, hipStream_t st, unsigned char* arena, size_t arena_bytes, int walk_grid) -> int {
    const Tower& T = m->graph.experts[e];
    int32_t* gl = glist + (size_t)e * gcap;
    if (launch_x8_guard_flag(lists + (size_t)e * B, counts + e, B, normals, gslot, gl, gcount + e, gcap, m->gstat, st)) return 1;
    float* out = nullptr;
    RunCtx rc{m, gcap, gcount + e, gl, st, false, m->expert_mix};
    rc.walk = walk_grid;                       // x8 = 0: the three-product loop everywhere; a small walking grid (a few dozen rows)
    if (run_tower(rc, T, X0, arena, arena_bytes, &out)) return 1;
    return launch_x8_guard_fix(out, T.bufs[T.out_buf].C, gl, gcount + e, gcap, normals, m->gstat, st);
  };
  if (guard && launch_x8_guard_begin(0, m->x8_guard_thr, gscale, B, m->gstat, gslot, stream)) return 1;
  for (int e = 0; e < E; ++e) {
    const Tower& T = m->graph.experts[e];
    const int ostride = T.bufs[T.out_buf].C;
    for (int r = 0; r < rounds; ++r) {
      float* out = nullptr;
      if (counts) {   // top-1 routing: only the points whose arg-max is e (test_n_est_w_experts.py:150-152), `cap` of them per round
        const int32_t* list = lists + (size_t)e * B + (size_t)r * cap;
        const int32_t* cnt = ecounts + e * rounds + r;
        RunCtx rc{m, cap, cnt, list, stream, false, m->expert_mix};
        rc.x8 = m->x8_mask;
        rc.walk = r >= 1;                      // an expert sees ~1 / E of a batch: rounds after the first are normally empty
        if (run_tower(rc, T, X0, tower_ws, tower_bytes_, &out)) return 1;
        if (launch_scatter3(out, ostride, list, cnt, cap, normals, stream)) return 1;
      } else {        // reference behaviour: every expert on every point -> [E,B,3], rows [r * cap, ...) of the batch per round
        const int take = std::min(cap, B - r * cap);
        RunCtx rc{m, take, nullptr, nullptr, stream, false, m->expert_mix};
        rc.x8 = m->x8_mask;
        if (run_tower(rc, T, (const unsigned char*)X0 + (size_t)r * cap * x0_row, tower_ws, tower_bytes_, &out)) return 1;
        if (launch_scatter3(out, ostride, nullptr, nullptr, take, normals + ((size_t)e * B + (size_t)r * cap) * 3, stream)) return 1;
      }
    }
    if (guard) {
      prof_phase(NESTI_PHASE_GUARD);
      if (side) {
        NESTI_CHECK_HIP(hipEventRecord(lane->gev_done[e], stream));
        NESTI_CHECK_HIP(hipStreamWaitEvent(lane->gstream, lane->gev_done[e], 0));
        if (guard_expert(e, lane->gstream, tower_ws + main_bytes, tower_bytes_ - main_bytes, m->x8_guard_walk)) return 1;
      } else if (guard_expert(e, stream, tower_ws, tower_bytes_, m->x8_guard_walk)) {
        return 1;
      }
      prof_phase(NESTI_PHASE_EXPERTS);
    }
  }
  if (guard) {
    if (side) {
      NESTI_CHECK_HIP(hipEventRecord(lane->gev_join, lane->gstream));
      NESTI_CHECK_HIP(hipStreamWaitEvent(stream, lane->gev_join, 0));
      glk.unlock();
    }
    prof_phase(NESTI_PHASE_GUARD);
    for (int pass = 1; pass <= NESTI_X8_GUARD_WIDEN_PASSES; ++pass) {
      if (launch_x8_guard_begin(pass, m->x8_guard_thr, gscale, B, m->gstat, gslot, stream)) return 1;
      for (int e = 0; e < E; ++e)
        if (guard_expert(e, stream, tower_ws, tower_bytes_, NESTI_GUARD_WIDEN_WALK_GRID)) return 1;
    }
  }
  return 0;
}

}  // namespace
}  // namespace nesti

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace nesti;

extern "C" {

const char* nesti_last_error(void) { return g_error.c_str(); }
const char* nesti_version(void) {
  // _lib.py refuses a library whose version string contains "TIMING-EXPERIMENTS" (common.h); measurement builds are named too
  return "nesti-hip 0.6 (gfx950)"
#ifdef NESTI_TIMING_EXPERIMENTS
         " TIMING-EXPERIMENTS build: WRONG RESULTS by construction"
#endif
#ifdef NESTI_ATTRIBUTION
         " [measurement build: NESTI_ATTRIBUTION]"
#endif
#ifdef NESTI_EXPERIMENT_XW
         " [measurement build: NESTI_EXPERIMENT_XW]"
#endif
      ;
}

void nesti_default_config(nesti_config_t* cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->arch = NESTI_ARCH_EXPERTS;
  cfg->n_scales = 3;                 // --patch_radius 0.01 0.03 0.05 (the published setting; the script default is 0.005 0.01 0.03)
  cfg->points_per_scale = 512;       // --num_point
  cfg->grid_n = 8;                   // --n_gaussians 8
  cfg->variance = 0.0156;            // --gmm_variance
  cfg->n_experts = 7;
  const int lo[7] = {0, 0, 1, 1, 2, 2, 0}, cnt[7] = {1, 1, 1, 1, 1, 1, 3};   // expert_dict  :62
  for (int i = 0; i < 7; ++i) { cfg->expert_scale_lo[i] = lo[i]; cfg->expert_scale_cnt[i] = cnt[i]; }
}

int nesti_gmm_grid(int n, double variance, float* w, float* mu, float* sigma) {
  if (n < 1 || !w || !mu || !sigma) NESTI_FAIL("nesti_gmm_grid: bad arguments");
  // np.mgrid[step-1 : 1-step : n j] per axis, reshape [3,-1].T  => x slowest (utils/utils.py:81-87)
  const double step = 1.0 / n;
  const double a0 = step - 1.0, a1 = 1.0 - step;
  const int G = n * n * n;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      for (int k = 0; k < n; ++k) {
        const int g = (i * n + j) * n + k;
        const int idx[3] = {i, j, k};
        for (int c = 0; c < 3; ++c) {
          const double v = (n == 1) ? a0 : a0 + (a1 - a0) * idx[c] / (double)(n - 1);
          mu[g * 3 + c] = (float)v;
          sigma[g * 3 + c] = (float)sqrt(variance);   // np.sqrt(gmm.covariances_)  test_n_est_w_experts.py:146
        }
        w[g] = (float)(1.0 / G);                       // utils/utils.py:89
      }
  return 0;
}

int nesti_mups_forward(const nesti_config_t* cfg, const float* points_dev, const int32_t* n_eff_dev, int B,
                       void* out_dev, int out_dtype, int out_cstride, void* stream) {
  if (B <= 0) return 0;   // empty batch: nothing to do
  if (!cfg || !points_dev || !n_eff_dev || !out_dev) NESTI_FAIL("nesti_mups_forward: null argument");
  const int tok = prof_begin(NESTI_PROF_MUPS, (hipStream_t)stream);
  const int rc = launch_mups(cfg, points_dev, n_eff_dev, B, out_dev, out_dtype, out_cstride, /*embed4=*/0, (hipStream_t)stream);
  prof_end(NESTI_PROF_MUPS, tok, (hipStream_t)stream);
  return rc;
}

int nesti_model_describe(const nesti_config_t* cfg, int* n_tensors, nesti_tensor_t* infos, int max_infos) {
  if (!cfg || !n_tensors) NESTI_FAIL("nesti_model_describe: null argument");
  Graph g;
  if (build_graph(cfg, &g)) return 1;
  static thread_local std::vector<std::string> names;
  std::vector<std::pair<std::string, std::vector<int64_t>>> all;
  for (const LayerDesc& d : g.layers) layer_tensors(d, &all);
  *n_tensors = (int)all.size();
  if (!infos) return 0;
  if (max_infos < (int)all.size()) NESTI_FAIL("nesti_model_describe: infos array too small");
  names.clear();
  names.reserve(all.size());
  for (size_t i = 0; i < all.size(); ++i) {
    names.push_back(all[i].first);
    infos[i].name = names.back().c_str();
    infos[i].data = nullptr;
    infos[i].ndim = (int)all[i].second.size();
    for (int d = 0; d < 5; ++d) infos[i].dims[d] = d < infos[i].ndim ? all[i].second[d] : 0;
  }
  return 0;
}

int nesti_model_create(const nesti_config_t* cfg, const nesti_tensor_t* tensors, int n_tensors, int dtype,
                       nesti_model_t** out) {
  if (!cfg || !tensors || !out) NESTI_FAIL("nesti_model_create: null argument");
  if (dtype != NESTI_F32 && dtype != NESTI_BF16 && dtype != NESTI_F16 && dtype != NESTI_BF16X3 && dtype != NESTI_F16X3 &&
      dtype != NESTI_F16X3C && dtype != NESTI_F16X8 && dtype != NESTI_F16X8C)
    NESTI_FAIL("nesti_model_create: bad dtype");
  if (dtype_cascade(dtype) && cfg->arch != NESTI_ARCH_EXPERTS)
    NESTI_FAIL("nesti_model_create: NESTI_F16X3C / NESTI_F16X8C is the two-stage gate of experts_n_est; use NESTI_F16X3 for the other models");
  if (dtype_x8(dtype) && (cfg->arch != NESTI_ARCH_EXPERTS || cfg->grid_n != 8))
    NESTI_FAIL("nesti_model_create: NESTI_F16X8 / NESTI_F16X8C (FP8 cross terms in the expert towers) is for experts_n_est on the 8^3 grid");
  std::unique_ptr<nesti_model> m(new nesti_model());
  m->cascade = dtype_cascade(dtype);
  const bool x8 = dtype_x8(dtype);
  dtype = main_dtype(dtype);
  m->dtype = dtype;
  if (build_graph(cfg, &m->graph, x8)) return 1;
  TensorTable tt;
  for (int i = 0; i < n_tensors; ++i) if (tensors[i].name) tt.by_name[tensors[i].name] = &tensors[i];
  m->packed.resize(m->graph.layers.size());
  for (size_t i = 0; i < m->graph.layers.size(); ++i)
    if (pack_layer(m->graph.layers[i], tt, dtype, &m->packed[i])) return 1;
  // the 8^3 tap kernel's x padding relies on out-of-range LDS reads returning zero: checked once per device, and only for
  // models that have such layers
  bool any_conv8 = false;
  for (const PackedLayer& pl : m->packed) any_conv8 = any_conv8 || pl.kind == 2;
  if (any_conv8 && conv8_selftest()) return 1;
  if (g_experiment_mix && act_planes(dtype) > 1 && cfg->arch == NESTI_ARCH_EXPERTS && cfg->grid_n == 8) {
    m->packed_mix.resize(m->graph.layers.size());
    const int plain = kernel_dtype(dtype);
    // the gating net's tap layers (conv8n_kernel / conv4n_kernel): one switch for all of them, bit kGateMixBit
    for (const Op& op : m->graph.gate.ops) {
      if (op.kind != Op::CONV) continue;
      const LayerDesc& d = m->graph.layers[op.layer];
      if (!use_conv8(d) && !(use_conv4(d, plain) && m->packed[op.layer].kind == 3)) continue;
      if (pack_layer(d, tt, plain, &m->packed_mix[op.layer])) return 1;
      m->packed_mix[op.layer].mix_bit = kGateMixBit;
    }
    for (const Tower& T : m->graph.experts)
      for (const Op& op : T.ops) {
        if (op.kind != Op::CONV) continue;
        const LayerDesc& d = m->graph.layers[op.layer];
        if (!use_conv8(d) && !(use_conv4(d, plain) && m->packed[op.layer].kind == 3)) continue;
        // "inception<B>Expert_<i>_conv<2|3>": blocks 1, 2 (8^3) and 4 (4^3); conv2 is the smaller kernel of the block
        const int blk = d.scope.size() > 9 ? d.scope[9] - '0' : 0;
        const int idx = blk == 1 ? 0 : blk == 2 ? 1 : blk == 4 ? 2 : -1;
        if (idx < 0) continue;
        if (pack_layer(d, tt, plain, &m->packed_mix[op.layer])) return 1;
        m->packed_mix[op.layer].mix_bit = 2 * idx + (d.scope.back() == '3' ? 1 : 0);
      }
  }
  if (x8) {
    // the experts' tap layers at 8^3 once more in the FP8 cross-term packing, and the pre-scale of the planes their block's conv1 writes
    m->packed_x8.resize(m->graph.layers.size());
    m->packed_x6.resize(m->graph.layers.size());
    for (const Tower& T : m->graph.experts)
      for (const Op& op : T.ops) {
        if (op.kind != Op::CONV) continue;
        if (op.aux_out_buf >= 0) m->packed[op.layer].x8_sc = x8_activation_exponent(m->graph.layers[op.layer], tt);
        if (op.aux_in_buf >= 0 && pack_layer_x8(m->graph.layers[op.layer], tt, &m->packed_x8[op.layer])) return 1;
        if (op.aux_in_buf >= 0 && pack_layer_x8(m->graph.layers[op.layer], tt, &m->packed_x6[op.layer], 6)) return 1;
      }
    m->x8_mask = 0xF;          // all four tap layers at 8^3 (include/nesti_hip.h: nesti_model_set_x8_layers)
    NESTI_CHECK_HIP(hipMalloc((void**)&m->gstat, 64));
    NESTI_CHECK_HIP(hipMemset(m->gstat, 0, 64));
    for (auto& gl : m->glane) {
      NESTI_CHECK_HIP(hipStreamCreateWithFlags(&gl.gstream, hipStreamNonBlocking));
      NESTI_CHECK_HIP(hipEventCreateWithFlags(&gl.gev_join, hipEventDisableTiming));
      for (int e = 0; e < cfg->n_experts; ++e) NESTI_CHECK_HIP(hipEventCreateWithFlags(&gl.gev_done[e], hipEventDisableTiming));
    }
  }
  if (m->cascade) {
    m->packed_fast.resize(m->graph.layers.size());
    for (const Op& op : m->graph.gate.ops) {
      if (op.kind != Op::CONV) continue;
      // the tap layers in plain f16; the one-tap layers (conv_igemm_kernel, kind 0, a single tap) keep an EMPTY entry: the filter pass
      // runs them on the model's pair-packed weights (run_tower: x2l)
      if (m->packed[op.layer].kind == 0 && m->packed[op.layer].n_taps == 1) continue;
      if (pack_layer(m->graph.layers[op.layer], tt, NESTI_F16, &m->packed_fast[op.layer])) return 1;
    }
    NESTI_CHECK_HIP(hipMalloc((void**)&m->cstat, 64));
    NESTI_CHECK_HIP(hipMemset(m->cstat, 0, 64));
  }
  NESTI_CHECK_HIP(hipDeviceSynchronize());
  *out = m.release();
  return 0;
}

void nesti_model_destroy(nesti_model_t* m) { delete m; }

int nesti_model_set_gate_margin(nesti_model_t* m, float tau) {
  if (!m || !m->cascade) NESTI_FAIL("nesti_model_set_gate_margin: not a NESTI_F16X3C model");
  if (!(tau >= 0.f)) NESTI_FAIL("nesti_model_set_gate_margin: tau must be >= 0");
  m->tau = tau;
  return 0;
}

int nesti_experiment_mix_enable(int on) {
  g_experiment_mix = on != 0;
  return 0;
}

int nesti_model_set_expert_mix(nesti_model_t* m, int mask) {
  if (!m) NESTI_FAIL("nesti_model_set_expert_mix: null model");
  if (mask && m->packed_mix.empty()) NESTI_FAIL("nesti_model_set_expert_mix: pair-mode experts_n_est models (8^3 grid) created after nesti_experiment_mix_enable(1) only");
  if (mask < 0 || mask >= (1 << 6)) NESTI_FAIL("nesti_model_set_expert_mix: mask has six bits");
  m->expert_mix = mask;
  return 0;
}

int nesti_model_set_x8_layers(nesti_model_t* m, int mask) {
  if (!m) NESTI_FAIL("nesti_model_set_x8_layers: null model");
  if (m->packed_x8.empty()) NESTI_FAIL("nesti_model_set_x8_layers: not an NESTI_F16X8 / NESTI_F16X8C model");
  if (mask < 0 || mask > 0xF) NESTI_FAIL("nesti_model_set_x8_layers: mask has four bits (inception1 conv2 / conv3, inception2 conv2 / conv3)");
  m->x8_mask = mask;
  return 0;
}

int nesti_f32_to_e2m3(float value, float inv_scale) { return (int)host_f32_to_e2m3(value, inv_scale); }

int nesti_model_set_x8_format(nesti_model_t* m, int bits) {
  if (!m) NESTI_FAIL("nesti_model_set_x8_format: null model");
  if (m->packed_x8.empty()) NESTI_FAIL("nesti_model_set_x8_format: not an NESTI_F16X8 / NESTI_F16X8C model");
  if (bits != 8 && bits != 6) NESTI_FAIL("nesti_model_set_x8_format: 8 (e4m3, one scale per layer) or 6 (e2m3, one scale per 16-channel block)");
  m->x8_fmt = bits;
  return 0;
}

int nesti_model_set_x8_guard(nesti_model_t* m, float thr) {
  if (!m || !m->gstat) NESTI_FAIL("nesti_model_set_x8_guard: not an NESTI_F16X8 / NESTI_F16X8C model");
  if (thr != thr) NESTI_FAIL("nesti_model_set_x8_guard: threshold is NaN");
  m->x8_guard_thr = thr;
  return 0;
}

int nesti_model_x8_guard_stats(const nesti_model_t* m, nesti_x8_guard_stats_t* out, int reset, void* stream) {
  if (!m || !m->gstat || !out) NESTI_FAIL("nesti_model_x8_guard_stats: not an NESTI_F16X8 / NESTI_F16X8C model / null argument");
  unsigned long long h[8];
  NESTI_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  NESTI_CHECK_HIP(hipMemcpy(h, m->gstat, sizeof(h), hipMemcpyDeviceToHost));
  if (reset) NESTI_CHECK_HIP(hipMemset(m->gstat, 0, 64));
  out->queries = h[0]; out->rechecked = h[1]; out->dropped = h[3];
  const uint32_t bits = (uint32_t)h[2];
  memcpy(&out->max_dn, &bits, 4);
  out->thr = m->x8_guard_thr;
  out->thr_eff = m->x8_guard_thr < 0.f ? m->x8_guard_thr
                                       : std::max(m->x8_guard_thr, NESTI_X8_GUARD_WIDEN * out->max_dn / sqrtf(2.f * NESTI_X8_GUARD_BAR));
  return 0;
}

int nesti_model_set_gate_mix(nesti_model_t* m, int on) {
  if (!m) NESTI_FAIL("nesti_model_set_gate_mix: null model");
  if (on && m->packed_mix.empty()) NESTI_FAIL("nesti_model_set_gate_mix: pair-mode experts_n_est models (8^3 grid) created after nesti_experiment_mix_enable(1) only");
#ifndef NESTI_EXPERIMENT_XW
  if (on == 2) NESTI_FAIL("nesti_model_set_gate_mix: mode 2 (the exact-weight filter emulation of profiles/r05_gate_medium.txt) needs a "
                          "library built with EXTRA_CXXFLAGS=-DNESTI_EXPERIMENT_XW");
#endif
  m->gate_mix = on == 2 ? 2 : on ? 1 : 0;   // 2: additionally every layer's OUTPUT is rounded to 16 bits (lo = 0): the numerics of a
                                            // plain-f16 gate whose 1x1x1 / FC layers multiply by the exact weights (hi * W_hi + hi * W_lo)
  return 0;
}

int nesti_model_cascade_stats(const nesti_model_t* m, nesti_cascade_stats_t* out, int reset, void* stream) {
  if (!m || !m->cascade || !out) NESTI_FAIL("nesti_model_cascade_stats: not a NESTI_F16X3C model");
  unsigned long long h[8];
  NESTI_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  NESTI_CHECK_HIP(hipMemcpy(h, m->cstat, sizeof(h), hipMemcpyDeviceToHost));
  if (reset) NESTI_CHECK_HIP(hipMemset(m->cstat, 0, 64));
  out->queries = h[0]; out->rechecked = h[1]; out->changed = h[2];
  const uint32_t bits = (uint32_t)h[3];
  memcpy(&out->max_margin_err, &bits, 4);
  out->tau = m->tau;
  memcpy(&out->sum_sq_pair_err, &h[4], 8);
  out->pairs = h[5];
  out->widened = h[6];
  out->widen_events = h[7];
  out->tau_eff = std::max(m->tau, NESTI_GATE_WIDEN * out->max_margin_err);
  return 0;
}

int nesti_model_gate_error_export(const nesti_model_t* m, float* dst_dev, void* stream) {
  if (!m || !m->cascade || !dst_dev) NESTI_FAIL("nesti_model_gate_error_export: not a NESTI_F16X3C model / null argument");
  return launch_gate_error_export(m->cstat, dst_dev, (hipStream_t)stream);
}

int nesti_model_gate_error_import(nesti_model_t* m, const float* src_dev, int n, void* stream) {
  if (!m || !m->cascade || (n > 0 && !src_dev)) NESTI_FAIL("nesti_model_gate_error_import: not a NESTI_F16X3C model / null argument");
  return launch_gate_error_import(m->cstat, src_dev, n, (hipStream_t)stream);
}

size_t nesti_tower_workspace_bytes(const nesti_config_t* cfg, int dtype, int tower, int batch) {
  if (!cfg || batch <= 0) return 0;
  Graph g;
  if (build_graph(cfg, &g, dtype_x8(dtype))) return 0;
  if (tower < -1 || tower >= (int)g.experts.size()) return 0;
  const int dt = tower < 0 && dtype_cascade(dtype) ? NESTI_F16 : main_dtype(dtype);
  return tower_bytes(tower < 0 ? g.gate : g.experts[tower], batch, dt);
}

size_t nesti_workspace_bytes(const nesti_model_t* m, int max_batch) {
  if (!m || max_batch <= 0) return 0;
  return ws_layout(m, max_batch).total;
}

int nesti_model_mups_cstride(const nesti_model_t* m) { return m ? nesti::mups_stride(m) : 0; }
int nesti_model_mups_rows(const nesti_model_t* m) { return m ? 1 << (3 * m->graph.gate_x0_log2S()) : 0; }

int nesti_model_mups(const nesti_model_t* m, const float* points_dev, const int32_t* n_eff_dev, int B, void* mups_out_dev,
                     void* stream) {
  if (B <= 0) return 0;
  if (!m || !points_dev || !n_eff_dev || !mups_out_dev) NESTI_FAIL("nesti_model_mups: null argument");
  const int tok = prof_begin(NESTI_PROF_MUPS, (hipStream_t)stream);
  const int rc = launch_mups(&m->graph.cfg, points_dev, n_eff_dev, B, mups_out_dev, m->dtype, mups_stride(m),
                             /*embed4=*/m->graph.cfg.grid_n == 3, (hipStream_t)stream);
  prof_end(NESTI_PROF_MUPS, tok, (hipStream_t)stream);
  return rc;
}

int nesti_gate_forward(const nesti_model_t* m, const void* mups_dev, int B, void* ws_dev, size_t ws_bytes,
                       float* probs_out_dev, int32_t* expert_out_dev, void* stream) {
  if (B <= 0) return 0;   // empty batch: nothing to do
  if (!m || !mups_dev || !ws_dev) NESTI_FAIL("nesti_gate_forward: null argument");
  if (m->graph.cfg.arch != NESTI_ARCH_EXPERTS && m->graph.cfg.arch != NESTI_ARCH_SWITCH)
    NESTI_FAIL("nesti_gate_forward: this model has no gating net");
  if (B <= 0) return 0;
  const WsLayout L = ws_layout(m, B);
  if (L.total > ws_bytes) NESTI_FAIL("nesti_gate_forward: workspace too small (see nesti_workspace_bytes)");
  unsigned char* ws = (unsigned char*)ws_dev;
  hipStream_t st = (hipStream_t)stream;
  if (m->cascade)
    return gate_cascade(m, mups_dev, B, ws, L, B, probs_out_dev, expert_out_dev ? expert_out_dev : (int32_t*)(ws + L.expert),
                        nullptr, nullptr, st);
  return gate_impl(m, mups_dev, B, ws + L.tower, L.total - L.tower, probs_out_dev, expert_out_dev, nullptr,
                   nullptr, st);
}

int nesti_experts_forward(const nesti_model_t* m, const void* mups_dev, const int32_t* expert_dev, int B, void* ws_dev,
                          size_t ws_bytes, float* normals_out_dev, void* stream) {
  if (B <= 0) return 0;   // empty batch: nothing to do
  if (!m || !mups_dev || !ws_dev || !normals_out_dev) NESTI_FAIL("nesti_experts_forward: null argument");
  if (B <= 0) return 0;
  const WsLayout L = ws_layout(m, B);
  if (L.total > ws_bytes) NESTI_FAIL("nesti_experts_forward: workspace too small (see nesti_workspace_bytes)");
  unsigned char* ws = (unsigned char*)ws_dev;
  hipStream_t st = (hipStream_t)stream;
  int32_t* counts = nullptr;
  int32_t* lists = nullptr;
  if (expert_dev) {
    counts = (int32_t*)(ws + L.counts);
    lists = (int32_t*)(ws + L.lists);
    if (launch_route(expert_dev, B, m->graph.cfg.n_experts, counts, lists, st)) return 1;
  }
  return experts_impl(m, mups_dev, B, B, ws + L.tower, L.total - L.tower, counts, lists, (int32_t*)(ws + L.ecounts),
                      m->graph.x8 ? (int32_t*)(ws + L.glist) : nullptr, normals_out_dev, st);
}

// gate -> routing -> experts on a MuPS tensor X0 that already sits in the workspace
static int forward_tail(const nesti_model_t* m, const void* X0, int B, int NB, unsigned char* ws, const WsLayout& L,
                        float* normals_out_dev, int32_t* expert_out_dev, float* probs_out_dev, hipStream_t st) {
  if (m->graph.cfg.arch == NESTI_ARCH_SINGLE || m->graph.cfg.arch == NESTI_ARCH_MULTI)   // single-tower ablations: the tower's output IS n_pred (test_n_est.py:136-141)
    return experts_impl(m, X0, B, NB, ws + L.tower, L.total - L.tower, nullptr, nullptr, nullptr, nullptr, normals_out_dev, st);
  float* probs = probs_out_dev ? probs_out_dev : (float*)(ws + L.probs);
  int32_t* expert = expert_out_dev ? expert_out_dev : (int32_t*)(ws + L.expert);
  int32_t* counts = (int32_t*)(ws + L.counts);
  int32_t* lists = (int32_t*)(ws + L.lists);
  if (m->cascade ? gate_cascade(m, X0, B, ws, L, NB, probs, expert, counts, lists, st)
                 : gate_impl(m, X0, B, ws + L.tower, L.total - L.tower, probs, expert, counts, lists, st))
    return 1;
  return experts_impl(m, X0, B, NB, ws + L.tower, L.total - L.tower, counts, lists, (int32_t*)(ws + L.ecounts),
                      m->graph.x8 ? (int32_t*)(ws + L.glist) : nullptr, normals_out_dev, st);
}

int nesti_forward(const nesti_model_t* m, const float* points_dev, const int32_t* n_eff_dev, int B, void* ws_dev,
                  size_t ws_bytes, float* normals_out_dev, int32_t* expert_out_dev, float* probs_out_dev, void* stream) {
  if (B <= 0) return 0;   // empty batch: nothing to do
  if (!m || !points_dev || !n_eff_dev || !ws_dev || !normals_out_dev) NESTI_FAIL("nesti_forward: null argument");
  const WsLayout L = ws_layout(m, B);
  if (L.total > ws_bytes) NESTI_FAIL("nesti_forward: workspace too small (see nesti_workspace_bytes)");
  unsigned char* ws = (unsigned char*)ws_dev;
  hipStream_t st = (hipStream_t)stream;
  void* X0 = ws + L.x0;
  const int tok = prof_begin(NESTI_PROF_MUPS, st);
  const int rcm = launch_mups(&m->graph.cfg, points_dev, n_eff_dev, B, X0, m->dtype, mups_stride(m),
                              /*embed4=*/m->graph.cfg.grid_n == 3, st);
  prof_end(NESTI_PROF_MUPS, tok, st);
  if (rcm) return 1;
  return forward_tail(m, X0, B, B, ws, L, normals_out_dev, expert_out_dev, probs_out_dev, st);
}

// ---- fused end-to-end entry: search grid + ball query + MuPS + gate + routed experts, batch by batch ---------------
// 8^3 Gaussian grid: patches_mups_kernel goes from the cloud to the MuPS tensor in one kernel (the patch tensors are never
// written); 3^3 grid: patches_kernel + mups3_kernel through a staging buffer in the workspace.
static size_t est_points_bytes(const nesti_model* m, int batch) {
  if (m->graph.cfg.grid_n == 8) return 0;
  return align_up((size_t)batch * m->graph.cfg.n_scales * m->graph.cfg.points_per_scale * 3 * sizeof(float), 256);
}
static size_t est_neff_bytes(const nesti_model* m, int batch) {
  return align_up((size_t)batch * m->graph.cfg.n_scales * sizeof(int32_t), 256);
}

size_t nesti_estimate_workspace_bytes(const nesti_model_t* m, int batch) {
  if (!m || batch <= 0) return 0;
  return est_points_bytes(m, batch) + est_neff_bytes(m, batch) + ws_layout(m, batch).total;
}

size_t nesti_estimate_workspace_bytes_for_config(const nesti_config_t* cfg, int dtype, int batch) {
  if (!cfg || batch <= 0) return 0;
  // the layout functions only look at the graph and the mode flags: a model shell without weights sizes exactly like the real one
  nesti_model shell;
  shell.cascade = dtype_cascade(dtype);
  shell.dtype = main_dtype(dtype);
  if (build_graph(cfg, &shell.graph, dtype_x8(dtype))) return 0;
  return nesti_estimate_workspace_bytes(&shell, batch);
}

int nesti_estimate_normals(const nesti_model_t* m, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                           const double* r_abs, uint64_t seed, int query_row0, int batch, int build_grid,
                           void* grid_ws_dev, size_t grid_ws_bytes, void* ws_dev, size_t ws_bytes,
                           float* normals_out_dev, int32_t* expert_out_dev, float* probs_out_dev, void* stream) {
  if (M <= 0) return 0;   // no queries: nothing to do
  if (!m || !cloud_dev || !r_abs || !grid_ws_dev || !ws_dev || !normals_out_dev)
    NESTI_FAIL("nesti_estimate_normals: null argument");
  if (N <= 0) NESTI_FAIL("nesti_estimate_normals: empty cloud");
  if (batch <= 0) NESTI_FAIL("nesti_estimate_normals: batch must be positive");
  if (query_row0 < 0 || (!query_idx_dev && (long long)query_row0 + M > (long long)N))
    NESTI_FAIL("nesti_estimate_normals: query rows [query_row0, query_row0 + M) exceed the cloud (N points)");
  if (nesti_estimate_workspace_bytes(m, batch) > ws_bytes)
    NESTI_FAIL("nesti_estimate_normals: workspace too small (see nesti_estimate_workspace_bytes)");
  if (grid_ws_bytes < nesti_patches_workspace_bytes(N)) NESTI_FAIL("nesti_estimate_normals: grid workspace too small");
  const nesti_config_t* cfg = &m->graph.cfg;
  for (int s = 0; s < cfg->n_scales; ++s)
    if (!(r_abs[s] > 0.0)) NESTI_FAIL("nesti_estimate_normals: radii must be positive");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)ws_dev;
  float* points = (float*)ws;
  int32_t* n_eff = (int32_t*)(ws + est_points_bytes(m, batch));
  unsigned char* fwd_ws = ws + est_points_bytes(m, batch) + est_neff_bytes(m, batch);
  const WsLayout L = ws_layout(m, batch);
  if (build_grid && nesti_patches_grid(cfg, cloud_dev, N, r_abs, grid_ws_dev, grid_ws_bytes, stream)) return 1;
  const int E = m->graph.cfg.arch == NESTI_ARCH_SWITCH ? 1 : m->graph.cfg.n_experts;   // columns of probs_out
  const bool fused = cfg->grid_n == 8;
  for (int done = 0; done < M; done += batch) {
    const int take = std::min(batch, M - done);
    const int32_t* qidx = query_idx_dev ? query_idx_dev + done : nullptr;
    float* n_out = normals_out_dev + (size_t)done * 3;
    int32_t* e_out = expert_out_dev ? expert_out_dev + done : nullptr;
    float* p_out = probs_out_dev ? probs_out_dev + (size_t)done * E : nullptr;
    if (fused) {
      const int tok = prof_begin(NESTI_PROF_MUPS, st);
      const int rcf = launch_patches_mups(cfg, cloud_dev, N, qidx, take, r_abs, seed, query_row0 + done, grid_ws_dev,
                                          fwd_ws + L.x0, m->dtype, mups_stride(m), n_eff, st);
      prof_end(NESTI_PROF_MUPS, tok, st);
      if (rcf) return 1;
      if (forward_tail(m, fwd_ws + L.x0, take, batch, fwd_ws, L, n_out, e_out, p_out, st)) return 1;
    } else {
      if (nesti_patches_query(cfg, cloud_dev, N, qidx, take, r_abs, seed, query_row0 + done, points, n_eff, nullptr, nullptr,
                              grid_ws_dev, grid_ws_bytes, stream))
        return 1;
      if (nesti_forward(m, points, n_eff, take, fwd_ws, L.total, n_out, e_out, p_out, stream)) return 1;
    }
  }
  return 0;
}

int nesti_estimate_normals_multi(const nesti_model_t* m, const nesti_shape_queries_t* items, int n_items, int batch,
                                 void* ws_dev, size_t ws_bytes, float* normals_out_dev, int32_t* expert_out_dev,
                                 float* probs_out_dev, void* stream) {
  {
    long long any = 0;
    for (int i = 0; items && i < n_items; ++i) any += items[i].n_queries > 0 ? items[i].n_queries : 0;
    if (any == 0) return 0;   // no queries: nothing to do
  }
  if (!m || !items || !ws_dev || !normals_out_dev) NESTI_FAIL("nesti_estimate_normals_multi: null argument");
  if (batch <= 0) NESTI_FAIL("nesti_estimate_normals_multi: batch must be positive");
  const nesti_config_t* cfg = &m->graph.cfg;
  if (cfg->grid_n != 8) NESTI_FAIL("nesti_estimate_normals_multi: the 8^3 Gaussian grid only (use nesti_estimate_normals per shape)");
  if (nesti_estimate_workspace_bytes(m, batch) > ws_bytes)
    NESTI_FAIL("nesti_estimate_normals_multi: workspace too small (see nesti_estimate_workspace_bytes)");
  long long total = 0;
  for (int i = 0; i < n_items; ++i) {
    const nesti_shape_queries_t& it = items[i];
    if (it.n_queries < 0 || (it.n_queries > 0 && (!it.cloud_dev || !it.grid_ws_dev || it.n_points <= 0)))
      NESTI_FAIL("nesti_estimate_normals_multi: bad item");
    if (it.query_row0 < 0 || (!it.query_idx_dev && (long long)it.query_row0 + it.n_queries > (long long)it.n_points))
      NESTI_FAIL("nesti_estimate_normals_multi: query rows of an item exceed its cloud");
    if (it.n_queries > 0 && it.grid_ws_bytes < nesti_patches_workspace_bytes(it.n_points))
      NESTI_FAIL("nesti_estimate_normals_multi: grid workspace of an item too small");
    for (int s = 0; s < cfg->n_scales; ++s)
      if (it.n_queries > 0 && !(it.r_abs[s] > 0.0)) NESTI_FAIL("nesti_estimate_normals_multi: radii must be positive");
    total += it.n_queries;
  }
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* fwd_ws = (unsigned char*)ws_dev + est_points_bytes(m, batch) + est_neff_bytes(m, batch);
  const WsLayout L = ws_layout(m, batch);
  unsigned char* X0 = fwd_ws + L.x0;
  const size_t row_bytes = (size_t)nesti_model_mups_rows(m) * mups_stride(m) * dtype_size(m->dtype);   // one query's MuPS
  const int E = m->graph.cfg.arch == NESTI_ARCH_SWITCH ? 1 : m->graph.cfg.n_experts;
  long long done = 0;           // rows emitted so far
  int item = 0, item_done = 0;  // cursor into the items
  while (done < total) {
    int fill = 0;
    const int tok = prof_begin(NESTI_PROF_MUPS, st);
    while (fill < batch && item < n_items) {
      const nesti_shape_queries_t& it = items[item];
      const int take = std::min(batch - fill, it.n_queries - item_done);
      if (take > 0 &&
          launch_patches_mups(cfg, it.cloud_dev, it.n_points, it.query_idx_dev ? it.query_idx_dev + item_done : nullptr, take,
                              it.r_abs, it.seed, it.query_row0 + item_done, it.grid_ws_dev, X0 + (size_t)fill * row_bytes,
                              m->dtype, mups_stride(m), nullptr, st)) {
        prof_end(NESTI_PROF_MUPS, tok, st);      // close the timing span on the error path too
        return 1;
      }
      fill += take;
      item_done += take;
      if (item_done >= it.n_queries) { ++item; item_done = 0; }
    }
    prof_end(NESTI_PROF_MUPS, tok, st);
    if (forward_tail(m, X0, fill, batch, fwd_ws, L, normals_out_dev + (size_t)done * 3, expert_out_dev ? expert_out_dev + done : nullptr,
                     probs_out_dev ? probs_out_dev + (size_t)done * E : nullptr, st))
      return 1;
    done += fill;
  }
  return 0;
}

int nesti_profile_enable(int on) {
  prof_collect();
  for (int c = 0; c < kProfSlots; ++c) { g_prof.ms[c] = 0; g_prof.launches[c] = 0; }
  g_prof.on = on != 0;
  g_prof.phase = NESTI_PHASE_INPUT;
  return 0;
}

int nesti_profile_read(double* ms, long long* launches) {
  prof_collect();   // synchronises on the recorded events
  for (int c = 0; c < kProfSlots; ++c) {
    if (ms) ms[c] = g_prof.ms[c];
    if (launches) launches[c] = g_prof.launches[c];
  }
  return 0;
}

int nesti_model_macs(const nesti_model_t* m, int tower, int kind, double* nominal, double* useful, double* issued) {
  if (!m) NESTI_FAIL("nesti_model_macs: null model");
  if (kind < -1 || kind > NESTI_PROF_ONE_BY_ONE) NESTI_FAIL("nesti_model_macs: kind must be -1 or a conv category");
  const int E = m->graph.cfg.n_experts;
  if (tower < -1 || tower >= E) NESTI_FAIL("nesti_model_macs: tower must be -1 (gate) or an expert index");
  const Tower& T = tower < 0 ? m->graph.gate : m->graph.experts[tower];
  double nom = 0, use = 0, iss = 0;
  for (const Op& op : T.ops) {
    if (op.kind != Op::CONV) continue;
    const LayerDesc& d = m->graph.layers[op.layer];
    const PackedLayer& pl = m->packed[op.layer];
    if (kind >= 0 && conv_category(d, pl) != kind) continue;
    const int S = d.s_real ? d.s_real : (1 << d.log2S), V = S * S * S, lo = (d.k - 1) / 2;
    long long valid = 0;   // sum over output voxels of the taps that land inside the volume
    for (int z = 0; z < S; ++z) for (int y = 0; y < S; ++y) for (int x = 0; x < S; ++x)
      for (int a = 0; a < d.k; ++a) for (int b = 0; b < d.k; ++b) for (int c = 0; c < d.k; ++c) {
        const int zz = z + a - lo, yy = y + b - lo, xx = x + c - lo;
        if (zz >= 0 && zz < S && yy >= 0 && yy < S && xx >= 0 && xx < S) ++valid;
      }
    const int parts = d.scope2.empty() ? 1 : 2;
    nom += (double)parts * V * d.k * d.k * d.k * d.cin * d.cout;
    use += (double)parts * valid * d.cin * d.cout;
    // MFMA tiles the kernels issue: conv8n_kernel (8^3) and the remapped conv_igemm_kernel layout at 4^3 hold one x-line
    // (y, z) per 32-row tile and skip it when y + dy or z + dz leaves the volume; elsewhere every kept tap is issued in full
    // (conv4n_kernel's tile is a single voxel: it issues exactly the taps that land inside the volume)
    double tap_sum = pl.n_taps;
    const int Si = 1 << d.log2S;
    const bool voxel_tiles = pl.kind == 3 || (pl.kind == 0 && !d.s_real && conv_remap(d.k, d.log2S, pl.n_taps) == 2);
    if (pl.n_taps > 1 && (pl.kind >= 1 || voxel_tiles || (d.log2S == 2 && conv_remap(d.k, d.log2S, pl.n_taps)))) {
      tap_sum = 0;
      for (int t = 0; t < pl.n_taps; ++t)
        tap_sum += (double)std::max(0, S - abs(pl.tap[t][0])) * std::max(0, S - abs(pl.tap[t][1])) / ((double)Si * Si) *
                   (voxel_tiles ? (double)std::max(0, S - abs(pl.tap[t][2])) / Si : 1.0);
    }
    iss += (double)(1 << (3 * d.log2S)) * tap_sum * d.Cin_p * d.Cout_p;
  }
  if (nominal) *nominal = nom;
  if (useful) *useful = use;
  if (issued) *issued = iss;
  return 0;
}

}  // extern "C"
