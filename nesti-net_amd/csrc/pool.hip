// Memory-bound helper kernels of the CNN towers: 2^3 max pooling on channels-last activations (the
// average pool and most max pools are fused into conv epilogues, conv.hip), the gating softmax +
// arg-max + routing, and the final scatter.
#include "kernels.h"

namespace nesti {
namespace {

constexpr int kThreads = 256;

template <int DT> struct Vec16;   // 16-byte vector of elements <-> floats
template <> struct Vec16<NESTI_F32> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <int DT> struct Vec16_16 {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = Elem<DT>::to_f32((uint16_t)(w[i] & 0xffffu));
      f[2 * i + 1] = Elem<DT>::to_f32((uint16_t)(w[i] >> 16));
    }
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)Elem<DT>::from_f32(f[2 * i]) | ((uint32_t)Elem<DT>::from_f32(f[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <> struct Vec16<NESTI_BF16> : Vec16_16<NESTI_BF16> {};
template <> struct Vec16<NESTI_F16> : Vec16_16<NESTI_F16> {};

// Pair-mode activations (NESTI_BF16X3 / NESTI_F16X3, common.h): eight logical channels = the hi vector at split_col(col)
// and the lo vector one 64-element plane further; the value is hi + lo.  Stores emit both planes.
template <int DT>
__device__ __forceinline__ void load_vec(const unsigned char* base, long long row_elems, int coff, int cv, int split, float* f) {
  using V = Vec16<DT>;
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;
  if constexpr (DT != NESTI_F32) {
    if (split) {
      const unsigned char* s0 = base + (row_elems + split_col(coff + cv * 8)) * 2;
      float h[8], l[8];
      V::unpack(*reinterpret_cast<const uint4*>(s0), h);
      V::unpack(*reinterpret_cast<const uint4*>(s0 + 2 * kSplitGroup), l);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = h[e] + l[e];
      return;
    }
  }
  V::unpack(*reinterpret_cast<const uint4*>(base + (row_elems + coff) * kEsz + cv * 16), f);
}
template <int DT>
__device__ __forceinline__ void store_vec(unsigned char* base, long long row_elems, int coff, int cv, int split, const float* m) {
  using V = Vec16<DT>;
  constexpr int kEsz = (DT == NESTI_F32) ? 4 : 2;
  if constexpr (DT != NESTI_F32) {
    if (split) {
      store_act8<Elem<DT>>(base, row_elems, coff + cv * 8, make_float4(m[0], m[1], m[2], m[3]), make_float4(m[4], m[5], m[6], m[7]), 1);
      return;
    }
  }
  *reinterpret_cast<uint4*>(base + (row_elems + coff) * kEsz + cv * 16) = V::pack(m);
}

// tf.nn.max_pool3d 2^3 stride 2 SAME on an even volume (utils/tf_util.py:424-428)
template <int DT>
__global__ __launch_bounds__(kThreads) void maxpool2_kernel(const PoolParams p) {
  using V = Vec16<DT>;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int log2S = p.log2S, log2V = 3 * log2S;
  const int log2So = log2S - 1, So = 1 << log2So, log2Vo = 3 * log2So;
  const int vecs = p.C / V::N;
  const long long total = ((long long)npts << log2Vo) * vecs;
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int cv = (int)(i % vecs);
    const long long orow = i / vecs;
    const long long pt = orow >> log2Vo;
    const int vox = (int)(orow & ((1 << log2Vo) - 1));
    const int z = vox >> (2 * log2So), y = (vox >> log2So) & (So - 1), x = vox & (So - 1);
    float m[V::N];
#pragma unroll
    for (int e = 0; e < V::N; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int zz = 2 * z + (a >> 2), yy = 2 * y + ((a >> 1) & 1), xx = 2 * x + (a & 1);
      const long long srow = (pt << log2V) + (((zz << log2S) + yy) << log2S) + xx;
      float f[V::N];
      load_vec<DT>(in_b, srow * p.in_cstride, p.in_coff, cv, p.split, f);
#pragma unroll
      for (int e = 0; e < V::N; ++e) m[e] = fmaxf(m[e], f[e]);
    }
    store_vec<DT>(out_b, orow * p.out_cstride, p.out_coff, cv, p.split, m);
  }
}

// tf.nn.max_pool3d(k=[3,3,3], stride 2, SAME) on a 3^3 volume (conv_net_3g, models/experts_n_est.py:238): TF pads one
// voxel in front, so output cell o reads input {2o-1, 2o, 2o+1} clipped to [0,2] = {o, o+1} per axis.  Input rows live
// in the 4^3 index space (row 16 z + 4 y + x), output is a dense 2^3.
template <int DT>
__global__ __launch_bounds__(kThreads) void maxpool3s2_kernel(const PoolParams p) {
  using V = Vec16<DT>;
  int npts = p.npoints;
  if (p.npoints_ptr) npts = min(npts, *p.npoints_ptr);
  const int nv = p.C / V::N;
  const long long total = (long long)npts * 8 * nv;
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % nv);
    const long long orow = idx / nv;
    const long long pt = orow >> 3;
    const int cell = (int)(orow & 7);
    const int z = cell >> 2, y = (cell >> 1) & 1, x = cell & 1;
    float m[V::N];
#pragma unroll
    for (int e = 0; e < V::N; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int zz = z + (a >> 2), yy = y + ((a >> 1) & 1), xx = x + (a & 1);
      const long long srow = pt * 64 + zz * 16 + yy * 4 + xx;
      float f[V::N];
      load_vec<DT>(in_b, srow * p.in_cstride, p.in_coff, cv, p.split, f);
#pragma unroll
      for (int e = 0; e < V::N; ++e) m[e] = fmaxf(m[e], f[e]);
    }
    store_vec<DT>(out_b, orow * p.out_cstride, p.out_coff, cv, p.split, m);
  }
}

// softmax (models/experts_n_est.py:177) + np.argmax first-index tie-break
// (test_n_est_w_experts.py:150) + optional routing lists for top-1 execution.
__global__ void gate_finish_kernel(const float* __restrict__ logits, int lstride, int B, int E,
                                   float* __restrict__ probs, int32_t* __restrict__ expert,
                                   int32_t* __restrict__ counts, int32_t* __restrict__ lists) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float l[NESTI_MAX_EXPERTS], mx = -INFINITY;
  for (int e = 0; e < E; ++e) { l[e] = logits[(size_t)b * lstride + e]; mx = fmaxf(mx, l[e]); }
  float sum = 0.f;
  for (int e = 0; e < E; ++e) { l[e] = expf(l[e] - mx); sum += l[e]; }
  int best = 0;
  float pb = -1.f;
  for (int e = 0; e < E; ++e) {
    const float pr = l[e] / sum;
    if (probs) probs[(size_t)b * E + e] = pr;
    if (pr > pb) { pb = pr; best = e; }
  }
  if (expert) expert[b] = best;
  if (counts) {
    const int pos = atomicAdd(&counts[best], 1);
    lists[(size_t)best * B + pos] = b;
  }
}

// ---- NESTI_F16X3C: the two-stage gate (include/nesti_hip.h) -----------------------------------------------------------
// cstat: device counters of the model: [0] queries, [1] rechecked (rows decided by the f16x3 gate, widening round included),
// [2] changed, [3] bits of max_margin_err (a float >= 0 orders like its bit pattern), [4] a double: sum of the squared pair
// errors, [5] the number of pairs in that sum, [6] rows rechecked by a widening round, [7] forward calls whose widening round
// was not empty.  The stats are accumulated with atomics in flag-list order, which is not deterministic: the double sum may
// differ in its last bits from run to run (it only feeds sigma, a diagnostic and one input of the calibration).
//
// The margin REACTS to what the gate measures (it does not just report it): the threshold of a forward call is
//   tau_eff = max(tau, NESTI_GATE_WIDEN * max_margin_err so far),
// snapshotted into the call's workspace by gate_begin_kernel (forward calls on other streams share the counters and may raise
// them while this one runs), and after the call's own recheck rounds up to NESTI_GATE_WIDEN_PASSES widening passes flag the band
// [tau_eff, NESTI_GATE_WIDEN * max_margin_err) again, each against a snapshot taken when it starts, so that an error the call
// has just measured -- in its recheck rounds or in an earlier widening pass -- is covered for the rows of the same call.
// Unrechecked rows therefore keep a margin of at least NESTI_GATE_WIDEN x the largest error seen on any row decided twice up to
// the start of the call's last widening pass.  The guarantee is per call: two calls in flight on two streams share the
// counters, and an error one of them measures after the other's last snapshot protects the other only from its next call on.
__global__ void gate_begin_kernel(int32_t* __restrict__ fcounts, const unsigned long long* __restrict__ cstat, float tau,
                                  float widen) {
  if (threadIdx.x != 0) return;
  fcounts[0] = 0;                                    // flag count of the filter pass
  fcounts[kWidenCountOff] = 0;                       // ... of the widening round
  const float m = __uint_as_float((unsigned)cstat[3]);
  reinterpret_cast<float*>(fcounts)[kTauEffOff] = fmaxf(tau, widen * m);
}

__device__ __forceinline__ float top2_margin(const float* l, int E, bool* has_nan) {
  float mx = -INFINITY, second = -INFINITY;
  bool nan = false;
  for (int e = 0; e < E; ++e) {
    nan |= !(l[e] == l[e]);
    if (l[e] > mx) { second = mx; mx = l[e]; } else second = fmaxf(second, l[e]);
  }
  *has_nan = nan;
  return mx - second;
}

// Stage 1 (on the plain-f16 gate's logits): softmax + first-index arg-max like gate_finish_kernel; the logits are kept for
// stage 2's error measurement and every row whose top-2 logit margin is below tau_eff -- or that holds a NaN logit (fmaxf and
// the > comparisons above skip NaNs, so such a row could otherwise keep a large finite margin) -- is appended to flag_list.
__global__ void gate_flag_kernel(const float* __restrict__ logits, int lstride, int B, int E,
                                 float* __restrict__ probs, int32_t* __restrict__ expert, float* __restrict__ keep,
                                 int32_t* __restrict__ fcounts, int32_t* __restrict__ flag_list,
                                 unsigned long long* __restrict__ cstat) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float tau = reinterpret_cast<const float*>(fcounts)[kTauEffOff];
  float l[NESTI_MAX_EXPERTS], mx = -INFINITY;
  for (int e = 0; e < E; ++e) {
    l[e] = logits[(size_t)b * lstride + e];
    keep[(size_t)b * NESTI_MAX_EXPERTS + e] = l[e];
    mx = fmaxf(mx, l[e]);
  }
  bool has_nan;
  const float margin = top2_margin(l, E, &has_nan);
  const bool flag = has_nan || !(margin >= tau);   // a NaN margin (inf - inf) is rechecked too
  float sum = 0.f;
  for (int e = 0; e < E; ++e) { l[e] = expf(l[e] - mx); sum += l[e]; }
  int best = 0;
  float pb = -1.f;
  for (int e = 0; e < E; ++e) {
    const float pr = l[e] / sum;
    if (probs) probs[(size_t)b * E + e] = pr;
    if (pr > pb) { pb = pr; best = e; }
  }
  expert[b] = best;
  if (flag) flag_list[atomicAdd(&fcounts[0], 1)] = b;
  // one counter update per wave
  const unsigned long long fm = __ballot(flag), am = __ballot(true);
  if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)am) - 1u) {
    atomicAdd(&cstat[0], (unsigned long long)__popcll(am));
    if (fm) atomicAdd(&cstat[1], (unsigned long long)__popcll(fm));
  }
}

// One widening pass (model.hip: gate_cascade runs up to NESTI_GATE_WIDEN_PASSES of them after the recheck rounds, each normally
// empty).  gate_widen_begin_kernel snapshots the pass's upper bound = widen x the largest error measured so far -- by this call's
// own recheck rounds or by calls on other streams -- into the call's workspace, so that every thread of the pass flags against
// the same band; gate_widen_kernel appends the rows whose f16 margin lies in [tau_eff, upper) to the flag list (the previous
// list is dead by then, its storage is reused); gate_widen_end_kernel writes the tower-round counts and raises the call's tau_eff
// to `upper`, so the next pass starts where this one ended: an error first measured INSIDE a widening pass is covered by the
// following pass of the same call.
__global__ void gate_widen_begin_kernel(int32_t* __restrict__ fcounts, const unsigned long long* __restrict__ cstat, float widen) {
  if (threadIdx.x != 0) return;
  fcounts[kWidenCountOff] = 0;
  reinterpret_cast<float*>(fcounts)[kWidenUpperOff] = widen * __uint_as_float((unsigned)cstat[3]);
}
__global__ void gate_widen_kernel(const float* __restrict__ keep, int B, int E, int32_t* __restrict__ fcounts,
                                  int32_t* __restrict__ flag_list, unsigned long long* __restrict__ cstat) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float lower = reinterpret_cast<const float*>(fcounts)[kTauEffOff];
  const float upper = reinterpret_cast<const float*>(fcounts)[kWidenUpperOff];
  if (!(upper > lower)) return;
  float l[NESTI_MAX_EXPERTS];
  for (int e = 0; e < E; ++e) l[e] = keep[(size_t)b * NESTI_MAX_EXPERTS + e];
  bool has_nan;
  const float margin = top2_margin(l, E, &has_nan);
  if (has_nan || !(margin >= lower && margin < upper)) return;    // NaN rows went through the first list
  const int pos = atomicAdd(&fcounts[kWidenCountOff], 1);
  flag_list[pos] = b;
  atomicAdd(&cstat[1], 1ull);
  atomicAdd(&cstat[6], 1ull);
  if (pos == 0) atomicAdd(&cstat[7], 1ull);
}
__global__ void gate_widen_end_kernel(int32_t* __restrict__ fcounts, int cap, int n_rounds) {
  const int t = threadIdx.x;
  if (t < n_rounds) fcounts[kWidenRoundsOff + t] = max(0, min(cap, fcounts[kWidenCountOff] - t * cap));
  if (t == 0) {
    float* f = reinterpret_cast<float*>(fcounts);
    if (f[kWidenUpperOff] > f[kTauEffOff]) f[kTauEffOff] = f[kWidenUpperOff];
  }
}

// multi-GPU: the largest error of the f16 gate as a device float, and the other ranks' values folded back in (dist.py)
__global__ void gate_error_export_kernel(const unsigned long long* __restrict__ cstat, float* __restrict__ dst) {
  if (threadIdx.x == 0) dst[0] = __uint_as_float((unsigned)cstat[3]);
}
__global__ void gate_error_import_kernel(unsigned long long* __restrict__ cstat, const float* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = src[i];
  if (v > 0.f && v < INFINITY) atomicMax(&cstat[3], (unsigned long long)__float_as_uint(v));   // NaN compares false
}

// rows [r * cap, r * cap + cap) of list i are one round of a tower that runs `cap` rows at a time:
// out[i * n_rounds + r] = how many of them exist (the flag list of the two-stage gate: one list; the routing lists: E)
__global__ void round_counts_kernel(const int32_t* __restrict__ counts, int n_lists, int cap, int n_rounds, int32_t* __restrict__ out) {
  const int t = threadIdx.x;
  if (t < n_lists * n_rounds) out[t] = max(0, min(cap, counts[t / n_rounds] - (t % n_rounds) * cap));
}

// Stage 2 (on the f16x3 gate's logits of one round, compact rows j <-> query flag_list[j]): the final probabilities and
// arg-max of those queries, and the f16 gate's error on the logit differences against its own arg-max.
__global__ void gate_recheck_kernel(const float* __restrict__ logits, int lstride, const int32_t* __restrict__ flag_list,
                                    const int32_t* __restrict__ count_ptr, int cap, int E, const float* __restrict__ keep,
                                    float* __restrict__ probs, int32_t* __restrict__ expert,
                                    unsigned long long* __restrict__ cstat) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= min(cap, *count_ptr)) return;
  const int b = flag_list[j];
  float l[NESTI_MAX_EXPERTS], mx = -INFINITY;
  for (int e = 0; e < E; ++e) { l[e] = logits[(size_t)j * lstride + e]; mx = fmaxf(mx, l[e]); }
  const int a = expert[b];                       // the f16 gate's arg-max
  float err = 0.f, sq = 0.f;
  for (int e = 0; e < E; ++e) {
    const float d16 = keep[(size_t)b * NESTI_MAX_EXPERTS + a] - keep[(size_t)b * NESTI_MAX_EXPERTS + e];
    const float pe = fabsf(d16 - (l[a] - l[e]));      // the f16 pass's error on the logit difference (a, e); 0 for e == a
    err = fmaxf(err, pe);
    sq += pe * pe;
  }
  float sum = 0.f;
  for (int e = 0; e < E; ++e) { l[e] = expf(l[e] - mx); sum += l[e]; }
  int best = 0;
  float pb = -1.f;
  for (int e = 0; e < E; ++e) {
    const float pr = l[e] / sum;
    if (probs) probs[(size_t)b * E + e] = pr;
    if (pr > pb) { pb = pr; best = e; }
  }
  expert[b] = best;
  if (best != a) atomicAdd(&cstat[2], 1ull);
  if (err == err) {
    atomicMax(&cstat[3], (unsigned long long)__float_as_uint(err));
    atomicAdd(reinterpret_cast<double*>(&cstat[4]), (double)sq);
    atomicAdd(&cstat[5], (unsigned long long)(E - 1));
  }
}

// tf.where(noise_est < 0.015, n_est_small, n_est_large) as a routing decision (models/ms_sw_n_est.py:80-82)
__global__ void switch_finish_kernel(const float* __restrict__ logits, int lstride, int B, float threshold,
                                     float* __restrict__ probs, int32_t* __restrict__ expert,
                                     int32_t* __restrict__ counts, int32_t* __restrict__ lists) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float noise = logits[(size_t)b * lstride];
  const int pick = (noise < threshold) ? 0 : 1;   // a NaN compares false -> large, like tf.where
  if (probs) probs[b] = noise;
  if (expert) expert[b] = pick;
  if (counts) {
    const int pos = atomicAdd(&counts[pick], 1);
    lists[(size_t)pick * B + pos] = b;
  }
}

__global__ void route_kernel(const int32_t* __restrict__ expert, int B, int E,
                             int32_t* __restrict__ counts, int32_t* __restrict__ lists) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int e = expert[b];
  if (e < 0 || e >= E) return;
  const int pos = atomicAdd(&counts[e], 1);
  lists[(size_t)e * B + pos] = b;
}

__global__ void scatter3_kernel(const float* __restrict__ src, int sstride, const int32_t* __restrict__ index,
                                const int32_t* __restrict__ count_ptr, int count_cap, float* __restrict__ out) {
  int n = count_cap;
  if (count_ptr) n = min(n, *count_ptr);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long long d = index ? index[j] : j;
#pragma unroll
  for (int c = 0; c < 3; ++c) out[d * 3 + c] = src[(size_t)j * sstride + c];
}

// Zeroes the routing counters.  A kernel, not hipMemsetAsync: inside a captured hipGraph the memset node of this
// 4*E-byte region did not take effect on replay (ROCm 7.2, gfx950) -- the counters then accumulated from replay to
// replay and the expert towers read stale list entries.
__global__ void zero_counts_kernel(int32_t* __restrict__ counts, int n) {
  if ((int)threadIdx.x < n) counts[threadIdx.x] = 0;
}

int grid_for(long long work) {
  long long g = (work + kThreads - 1) / kThreads;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

int launch_maxpool2(const PoolParams& p, int dtype, hipStream_t stream) {
  if (p.npoints <= 0) return 0;
  if (p.log2S < 1) NESTI_FAIL("maxpool2: input volume must be >= 2^3");
  const int n = (dtype == NESTI_F32) ? 4 : 8;
  if (p.C % n) NESTI_FAIL("maxpool2: C must be a multiple of the 16-byte vector");
  const long long work = ((long long)p.npoints << (3 * (p.log2S - 1))) * (p.C / n);
  dim3 grid(grid_for(work)), block(kThreads);
  if (dtype == NESTI_F32) hipLaunchKernelGGL(maxpool2_kernel<NESTI_F32>, grid, block, 0, stream, p);
  else if (dtype == NESTI_BF16) hipLaunchKernelGGL(maxpool2_kernel<NESTI_BF16>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(maxpool2_kernel<NESTI_F16>, grid, block, 0, stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_maxpool3s2(const PoolParams& p, int dtype, hipStream_t stream) {
  if (p.npoints <= 0) return 0;
  const int n = (dtype == NESTI_F32) ? 4 : 8;
  if (p.C % n) NESTI_FAIL("maxpool3s2: C must be a multiple of the 16-byte vector");
  const long long work = (long long)p.npoints * 8 * (p.C / n);
  dim3 grid(grid_for(work)), block(kThreads);
  if (dtype == NESTI_F32) hipLaunchKernelGGL(maxpool3s2_kernel<NESTI_F32>, grid, block, 0, stream, p);
  else if (dtype == NESTI_BF16) hipLaunchKernelGGL(maxpool3s2_kernel<NESTI_BF16>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(maxpool3s2_kernel<NESTI_F16>, grid, block, 0, stream, p);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_gate_finish(const float* logits, int lstride, int B, int E, float* probs, int32_t* expert,
                       int32_t* counts, int32_t* lists, hipStream_t stream) {
  if (B <= 0) return 0;
  if (E > NESTI_MAX_EXPERTS) NESTI_FAIL("gate_finish: too many experts");
  if (counts) hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, stream, counts, E);
  hipLaunchKernelGGL(gate_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, logits, lstride, B, E,
                     probs, expert, counts, lists);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_gate_flag(const float* logits, int lstride, int B, int E, float tau, float widen, float* probs, int32_t* expert,
                     float* keep, int32_t* fcounts, int32_t* flag_list, int cap, int n_rounds, unsigned long long* cstat,
                     hipStream_t stream) {
  if (B <= 0) return 0;
  if (E > NESTI_MAX_EXPERTS || n_rounds > kMaxCascadeRounds) NESTI_FAIL("gate_flag: too many experts / rounds");
  hipLaunchKernelGGL(gate_begin_kernel, dim3(1), dim3(64), 0, stream, fcounts, cstat, tau, widen);
  hipLaunchKernelGGL(gate_flag_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, logits, lstride, B, E, probs,
                     expert, keep, fcounts, flag_list, cstat);
  hipLaunchKernelGGL(round_counts_kernel, dim3(1), dim3(64), 0, stream, fcounts, 1, cap, n_rounds, fcounts + kRoundCountsOff);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_gate_widen(const float* keep, int B, int E, float widen, int32_t* fcounts, int32_t* flag_list, int cap, int n_rounds,
                      unsigned long long* cstat, hipStream_t stream) {
  if (B <= 0) return 0;
  if (n_rounds > kMaxCascadeRounds) NESTI_FAIL("gate_widen: too many rounds");
  hipLaunchKernelGGL(gate_widen_begin_kernel, dim3(1), dim3(64), 0, stream, fcounts, cstat, widen);
  hipLaunchKernelGGL(gate_widen_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, keep, B, E, fcounts, flag_list, cstat);
  hipLaunchKernelGGL(gate_widen_end_kernel, dim3(1), dim3(64), 0, stream, fcounts, cap, n_rounds);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_gate_error_export(const unsigned long long* cstat, float* dst, hipStream_t stream) {
  hipLaunchKernelGGL(gate_error_export_kernel, dim3(1), dim3(64), 0, stream, cstat, dst);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_gate_error_import(unsigned long long* cstat, const float* src, int n, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gate_error_import_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cstat, src, n);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_round_counts(const int32_t* counts, int n_lists, int cap, int n_rounds, int32_t* out, hipStream_t stream) {
  if (n_lists * n_rounds > 256) NESTI_FAIL("round_counts: too many lists x rounds");
  hipLaunchKernelGGL(round_counts_kernel, dim3(1), dim3(256), 0, stream, counts, n_lists, cap, n_rounds, out);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_gate_recheck(const float* logits, int lstride, const int32_t* flag_list, const int32_t* count_ptr, int cap, int E,
                        const float* keep, float* probs, int32_t* expert, unsigned long long* cstat, hipStream_t stream) {
  if (cap <= 0) return 0;
  hipLaunchKernelGGL(gate_recheck_kernel, dim3((cap + 255) / 256), dim3(256), 0, stream, logits, lstride, flag_list,
                     count_ptr, cap, E, keep, probs, expert, cstat);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_switch_finish(const float* logits, int lstride, int B, float threshold, float* probs, int32_t* expert,
                         int32_t* counts, int32_t* lists, hipStream_t stream) {
  if (B <= 0) return 0;
  if (counts) hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, stream, counts, 2);
  hipLaunchKernelGGL(switch_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, logits, lstride, B, threshold,
                     probs, expert, counts, lists);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- the conditioning guard of the FP8 cross-term experts (NESTI_F16X8 / NESTI_F16X8C; model.hip: experts_impl) ---------------------
// The FP8 residual moves an expert's output n by |dn| ~ 5e-5 (<= 2e-4 on 100 000 queries), independent of |n|; 1 - cos against the
// three-product result is (|dn| / |n|)^2 / 2, so only outputs of very small norm can be tilted by more than the bar.  A query whose
// |n| is below thr = widen x (largest |dn| measured so far) / theta, theta = sqrt(2 x 2.5e-6), is therefore evaluated AGAIN by its
// expert in f16x3 proper and the result replaces the X8 one; the rows decided twice measure |dn| (it does not depend on |n|), and the
// threshold follows the measurement like the two-stage gate's margin does.
//   gstat (device, per model): [0] queries, [1] rows re-evaluated, [2] largest |dn| (float bits), [3] rows dropped because a list was full
//   slot  (device, per call):  [0] lower, [1] upper end of the current pass's |n| band (floats)
__global__ void x8_guard_begin_kernel(int pass, float thr, float scale, int B, unsigned long long* __restrict__ gstat, float* __restrict__ slot) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float hi = fmaxf(thr, scale * __uint_as_float((unsigned)gstat[2]));      // scale = widen / theta
  if (pass == 0) {
    slot[0] = 0.f;
    slot[1] = hi;
    atomicAdd(&gstat[0], (unsigned long long)B);
  } else {
    slot[0] = slot[1];
    slot[1] = fmaxf(hi, slot[1]);
  }
}
// expert e's routing list -> the rows whose |n| lies in the band, compacted into glist (at most cap of them)
__global__ void x8_guard_flag_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ count_ptr, int count_cap,
                                     const float* __restrict__ normals, const float* __restrict__ slot, int32_t* __restrict__ glist,
                                     int32_t* __restrict__ gcount, int cap, unsigned long long* __restrict__ gstat) {
  const int n = min(count_cap, *count_ptr);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int i = list[j];
  const float x = normals[(size_t)i * 3], y = normals[(size_t)i * 3 + 1], z = normals[(size_t)i * 3 + 2];
  const float r = sqrtf(x * x + y * y + z * z);
  if (!(r >= slot[0] && r < slot[1]) && r == r) return;                            // a NaN output is re-evaluated too
  const int pos = atomicAdd(gcount, 1);
  if (pos < cap) glist[pos] = i;
  else atomicAdd(&gstat[3], 1ull);
}
// the f16x3 results of the flagged rows replace the X8 ones; |dn| is measured on the way
__global__ void x8_guard_fix_kernel(const float* __restrict__ src, int sstride, const int32_t* __restrict__ glist,
                                    const int32_t* __restrict__ gcount, int cap, float* __restrict__ normals,
                                    unsigned long long* __restrict__ gstat) {
  const int n = min(cap, *gcount);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0 && n > 0) atomicAdd(&gstat[1], (unsigned long long)n);
  if (j >= n) return;
  const long long i = glist[j];
  float d2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float nw = src[(size_t)j * sstride + c], od = normals[i * 3 + c];
    d2 += (nw - od) * (nw - od);
    normals[i * 3 + c] = nw;
  }
  const float dn = sqrtf(d2);
  if (dn > 0.f && dn < INFINITY) atomicMax(&gstat[2], (unsigned long long)__float_as_uint(dn));
}

int launch_x8_guard_begin(int pass, float thr, float scale, int B, unsigned long long* gstat, float* slot, hipStream_t stream) {
  hipLaunchKernelGGL(x8_guard_begin_kernel, dim3(1), dim3(64), 0, stream, pass, thr, scale, B, gstat, slot);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_x8_guard_flag(const int32_t* list, const int32_t* count_ptr, int count_cap, const float* normals, const float* slot,
                         int32_t* glist, int32_t* gcount, int cap, unsigned long long* gstat, hipStream_t stream) {
  if (count_cap <= 0) return 0;
  hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, stream, gcount, 1);
  hipLaunchKernelGGL(x8_guard_flag_kernel, dim3((count_cap + 255) / 256), dim3(256), 0, stream, list, count_ptr, count_cap, normals,
                     slot, glist, gcount, cap, gstat);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_x8_guard_fix(const float* src, int sstride, const int32_t* glist, const int32_t* gcount, int cap, float* normals,
                        unsigned long long* gstat, hipStream_t stream) {
  if (cap <= 0) return 0;
  // a walking-size grid: the list is normally a few dozen rows; a full one is walked by the same threads
  hipLaunchKernelGGL(x8_guard_fix_kernel, dim3((cap + 255) / 256), dim3(256), 0, stream, src, sstride, glist, gcount, cap, normals, gstat);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_route(const int32_t* expert, int B, int E, int32_t* counts, int32_t* lists, hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, stream, counts, E);
  hipLaunchKernelGGL(route_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, expert, B, E, counts, lists);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_scatter3(const float* src, int sstride, const int32_t* index, const int32_t* count_ptr,
                    int count_cap, float* out, hipStream_t stream) {
  if (count_cap <= 0) return 0;
  hipLaunchKernelGGL(scatter3_kernel, dim3((count_cap + 255) / 256), dim3(256), 0, stream, src, sstride,
                     index, count_ptr, count_cap, out);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace nesti
