// MuPS (multi-scale point statistics) kernel for gfx950.
//
// Computes what utils/tf_util.py:655-753 (get_3dmfv_n_est) + models/experts_n_est.py:66-76
// compute, but never materialises the [B,P,G,3] broadcast tensors the TF graph
// builds (utils/tf_util.py:671-678).  On the reference's uniform product grid with
// isotropic sigma the posterior factorises per axis, Q_ijk = qx_i * qy_j * qz_k
// (SURVEY.md §8(a')), so one patch point costs 24 exp instead of 512.
//
// Mapping: one 256-thread workgroup per query point; thread t owns the two
// Gaussians (i0, j, k) and (i0+4, j, k) with k = t&7, j = (t>>3)&7, i0 = t>>6
// (flat Gaussian index 64*i + 8*j + k, x slowest -- utils/utils.py:84-87) and
// keeps their 2x20 running max/min/sum statistics in registers.  Patch points
// are processed in chunks of 64: the per-axis terms (q, d, d^2-1) of a chunk are
// produced by 192 threads into LDS, then every thread sweeps the chunk with
// broadcast ds_read_b128s.  The kernel is fp32-VALU bound.
#include "common.h"
#include "patches_dev.h"

namespace nesti {

namespace {

constexpr int kR = 8;          // Gaussians per axis
constexpr int kG = kR * kR * kR;
constexpr int kChunk = 64;     // patch points staged per sweep
constexpr int kThreads = 256;

typedef float f2 __attribute__((ext_vector_type(2)));   // the thread's two Gaussians: packed-f32 VALU (v_pk_*)

struct Stats {
  f2 sq, mq;                    // sum Q, max Q
  f2 mu_max[3], mu_min[3], mu_sum[3];
  f2 sg_max[3], sg_min[3], sg_sum[3];
  __device__ __forceinline__ void init() {
    sq = f2{0.f, 0.f};
    mq = f2{-INFINITY, -INFINITY};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mu_max[c] = f2{-INFINITY, -INFINITY}; mu_min[c] = f2{INFINITY, INFINITY}; mu_sum[c] = f2{0.f, 0.f};
      sg_max[c] = f2{-INFINITY, -INFINITY}; sg_min[c] = f2{INFINITY, INFINITY}; sg_sum[c] = f2{0.f, 0.f};
    }
  }
  // Q = posterior; d[c] = (x_c - mu_c)/sigma; e[c] = d[c]^2 - 1
  __device__ __forceinline__ void update(f2 Q, const f2 (&d)[3], const f2 (&e)[3]) {
    sq += Q;
    mq = __builtin_elementwise_max(mq, Q);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f2 m = Q * d[c];                         // utils/tf_util.py:713
      mu_max[c] = __builtin_elementwise_max(mu_max[c], m);
      mu_min[c] = __builtin_elementwise_min(mu_min[c], m);
      mu_sum[c] += m;
      const f2 v = Q * e[c];                         // utils/tf_util.py:717
      sg_max[c] = __builtin_elementwise_max(sg_max[c], v);
      sg_min[c] = __builtin_elementwise_min(sg_min[c], v);
      sg_sum[c] += v;
    }
  }
  // two patch rows at once: the 26 running max/min become 3-input v_max3_f32 / v_min3_f32, halving their count
  __device__ __forceinline__ void update2(f2 Qa, const f2 (&da)[3], const f2 (&ea)[3], f2 Qb, const f2 (&db)[3],
                                          const f2 (&eb)[3]) {
    sq += Qa;
    sq += Qb;
    mq = __builtin_elementwise_max(__builtin_elementwise_max(mq, Qa), Qb);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f2 ma = Qa * da[c], mb = Qb * db[c];
      mu_max[c] = __builtin_elementwise_max(__builtin_elementwise_max(mu_max[c], ma), mb);
      mu_min[c] = __builtin_elementwise_min(__builtin_elementwise_min(mu_min[c], ma), mb);
      mu_sum[c] += ma;
      mu_sum[c] += mb;
      const f2 va = Qa * ea[c], vb = Qb * eb[c];
      sg_max[c] = __builtin_elementwise_max(__builtin_elementwise_max(sg_max[c], va), vb);
      sg_min[c] = __builtin_elementwise_min(__builtin_elementwise_min(sg_min[c], va), vb);
      sg_sum[c] += va;
      sg_sum[c] += vb;
    }
  }
};

// raw statistics of ONE of the two Gaussians
struct Stats1 {
  float sq, mq;
  float mu_max[3], mu_min[3], mu_sum[3];
  float sg_max[3], sg_min[3], sg_sum[3];
};
__device__ __forceinline__ Stats1 pick(const Stats& s, int which) {
  Stats1 o;
  o.sq = s.sq[which]; o.mq = s.mq[which];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o.mu_max[c] = s.mu_max[c][which]; o.mu_min[c] = s.mu_min[c][which]; o.mu_sum[c] = s.mu_sum[c][which];
    o.sg_max[c] = s.sg_max[c][which]; o.sg_min[c] = s.sg_min[c][which]; o.sg_sum[c] = s.sg_sum[c][which];
  }
  return o;
}

__device__ __forceinline__ float signed_sqrt(float v) {   // utils/tf_util.py:732-735, alpha = 0.5
  return copysignf(sqrtf(fabsf(v)), v);
}

// the 20 channels of one scale of one Gaussian row; row_off = row * channel stride (physical), col0 = 20 s (logical)
template <int DT>
__device__ __forceinline__ void store20(void* out, size_t row_off, int col0, const float (&v)[20]) {
  using E = Elem<DT>;
  if constexpr (DT == NESTI_F32) {
    float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row_off + col0);
#pragma unroll
    for (int q = 0; q < 5; ++q) p[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  } else if constexpr (is_x3<DT>) {
#pragma unroll
    for (int q = 0; q < 5; ++q)
      store_act4<E>(reinterpret_cast<unsigned char*>(out), (long long)row_off, col0 + 4 * q, v[4 * q], v[4 * q + 1], v[4 * q + 2],
                    v[4 * q + 3], 1);
  } else {
    uint2* p = reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + row_off + col0);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      uint32_t lo = (uint32_t)E::from_f32(v[4 * q]) | ((uint32_t)E::from_f32(v[4 * q + 1]) << 16);
      uint32_t hi = (uint32_t)E::from_f32(v[4 * q + 2]) | ((uint32_t)E::from_f32(v[4 * q + 3]) << 16);
      p[q] = make_uint2(lo, hi);
    }
  }
}
// zeros in the padding channels of one row: logical columns [pad0, width) -- for the pair modes in both planes
// of every 64-channel group (cstride is the physical stride, 2 x the logical width)
template <int DT>
__device__ __forceinline__ void zero_pad_row(void* out, size_t row_off, int pad0, int cstride) {
  using E = Elem<DT>;
  typename E::T* o = reinterpret_cast<typename E::T*>(out) + row_off;
  if constexpr (is_x3<DT>) {
    for (int c = 0; c < cstride; ++c) {
      const int logical = (c / (kPairPlanes * kSplitGroup)) * kSplitGroup + (c & (kSplitGroup - 1));
      if (logical >= pad0) o[c] = 0;
    }
  } else {
    for (int c = pad0; c < cstride; ++c) o[c] = E::from_f32(0.f);
  }
}

// Turn raw statistics into the 20 channels of one Gaussian, before L2 normalisation.
// Channel order: utils/tf_util.py:710-719,744-747.
__device__ __forceinline__ void finish(const Stats1& s, int nrows, bool has_masked, float m_f,
                                       float w, float (&v)[20]) {
  const float rsw = 1.0f / sqrtf(w);           // 1/sqrt(w)        :709,714
  const float rs2w = 1.0f / sqrtf(2.0f * w);   // 1/sqrt(2w)       :718
  // Masked rows (index > n_eff) contribute exactly 0 to every max/min/sum (:697,:702).
  float pmax = (s.mq - w);
  if (has_masked) pmax = fmaxf(pmax, 0.f);
  v[0] = pmax * rsw;
  v[1] = (s.sq - (float)nrows * w) * rsw;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float a = s.mu_max[c], b = s.mu_min[c], d = s.sg_max[c], e = s.sg_min[c];
    if (has_masked) { a = fmaxf(a, 0.f); b = fminf(b, 0.f); d = fmaxf(d, 0.f); e = fminf(e, 0.f); }
    v[2 + c] = a * rsw;
    v[5 + c] = b * rsw;
    v[8 + c] = s.mu_sum[c] * rsw;
    v[11 + c] = d * rs2w;
    v[14 + c] = e * rs2w;
    v[17 + c] = s.sg_sum[c] * rs2w;
  }
#pragma unroll
  for (int c = 0; c < 20; ++c) v[c] = signed_sqrt(v[c] / m_f);   // :727-735
}

// LDS state of one query's MuPS sweep.  Per staged point: x axis as 4 pairs (i, i+4) laid out for packed math:
// {q_i, q_i+4, d_i, d_i+4} and {e_i, e_i+4, -, -}; y and z axes as {q, d, d^2-1, -} per grid index
struct MupsShared {
  float4 stage_x[kChunk][2 * (kR / 2)];
  float4 stage_yz[kChunk][2 * kR];
  float red[kThreads / 64][20];
  float norm2[20];
};

// One scale of one query: the 20 statistics of the 512 Gaussians of patch rows 0 .. min(m + 1, P) - 1, written to
// channels [20 s, 20 s + 20) of the query's 512 output rows.  coord(row, axis) returns the patch coordinate (rows >= m
// are the reference's zero padding); m = n_eff of this scale.
template <int DT, class Coord>
__device__ __forceinline__ void mups_one_scale(MupsShared& sh, Coord coord, int m, int P, int b, int s, void* __restrict__ out,
                                               int cstride, float sigma, float w, int t) {
  const int k = t & 7, j = (t >> 3) & 7, i0 = t >> 6;
  const int lane = t & 63, wave = t >> 6;
  const int g0 = 64 * i0 + 8 * j + k;
  const int g1 = g0 + 256;
  const size_t o0 = ((size_t)b * kG + g0) * cstride;
  const size_t o1 = ((size_t)b * kG + g1) * cstride;
  if (m <= 0) {   // zero-padded batch tail (test_n_est_w_experts.py:134-140): skip, do not divide by 0
    float z[20];
#pragma unroll
    for (int c = 0; c < 20; ++c) z[c] = 0.f;
    store20<DT>(out, o0, 20 * s, z);
    store20<DT>(out, o1, 20 * s, z);
    return;
  }
  // rows 0..m are unmasked: `mask = r > n_eff` (utils/tf_util.py:693), so row m -- normally the
  // first zero-padding row -- is counted as a point.
  const int nrows = min(m + 1, P);
  const bool has_masked = nrows < P;

  Stats acc;
  acc.init();

  for (int c0 = 0; c0 < nrows; c0 += kChunk) {
    __syncthreads();
    if (t < 3 * kChunk) {
      const int axis = t >> 6, nl = t & 63;
      if (c0 + nl < nrows) {
        const float x = coord(c0 + nl, axis);
        float d[kR], e[kR], sum = 0.f;
#pragma unroll
        for (int i = 0; i < kR; ++i) {
          const float mu = -0.875f + 0.25f * (float)i;      // utils/utils.py:81-87 (exact in fp32)
          d[i] = (x - mu) / sigma;                           // utils/tf_util.py:687
          e[i] = expf(-0.5f * d[i] * d[i]);
          sum += e[i];
        }
        if (axis == 0) {
#pragma unroll
          for (int i = 0; i < kR / 2; ++i) {
            sh.stage_x[nl][2 * i] = make_float4(e[i] / sum, e[i + 4] / sum, d[i], d[i + 4]);
            sh.stage_x[nl][2 * i + 1] = make_float4(d[i] * d[i] - 1.0f, d[i + 4] * d[i + 4] - 1.0f, 0.f, 0.f);
          }
        } else {
#pragma unroll
          for (int i = 0; i < kR; ++i)
            sh.stage_yz[nl][(axis - 1) * kR + i] = make_float4(e[i] / sum, d[i], d[i] * d[i] - 1.0f, 0.f);
        }
      }
    }
    __syncthreads();
    const int cnt = min(kChunk, nrows - c0);
    auto row_terms = [&](int nl, f2* Q, f2 (&d)[3], f2 (&e)[3]) __attribute__((always_inline)) {
      const float4 XA = sh.stage_x[nl][2 * i0];
      const float4 XB = sh.stage_x[nl][2 * i0 + 1];
      const float4 Y = sh.stage_yz[nl][j];
      const float4 Z = sh.stage_yz[nl][kR + k];
      asm volatile("" ::"v"(Y.w), "v"(Z.w));   // keep these ds_read_b128: a 12-byte ds_read_b96 costs 8 LDS cycles, not 4
      const float qyz = Y.x * Z.x;
      *Q = f2{XA.x, XA.y} * qyz;
      d[0] = f2{XA.z, XA.w}; d[1] = f2{Y.y, Y.y}; d[2] = f2{Z.y, Z.y};
      e[0] = f2{XB.x, XB.y}; e[1] = f2{Y.z, Y.z}; e[2] = f2{Z.z, Z.z};
    };
    int nl = 0;
    for (; nl + 1 < cnt; nl += 2) {
      f2 Qa, Qb, da[3], ea[3], db[3], eb[3];
      row_terms(nl, &Qa, da, ea);
      row_terms(nl + 1, &Qb, db, eb);
      acc.update2(Qa, da, ea, Qb, db, eb);
    }
    if (nl < cnt) {
      f2 Q, d[3], e[3];
      row_terms(nl, &Q, d, e);
      acc.update(Q, d, e);
    }
  }

  float v0[20], v1[20];
  const float m_f = (float)m;                    // utils/tf_util.py:722
  finish(pick(acc, 0), nrows, has_masked, m_f, w, v0);
  finish(pick(acc, 1), nrows, has_masked, m_f, w, v1);

  // L2 normalisation over the 512 Gaussians, per channel (utils/tf_util.py:738-740)
  float part[20];
#pragma unroll
  for (int c = 0; c < 20; ++c) {
    float pp = v0[c] * v0[c] + v1[c] * v1[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pp += __shfl_xor(pp, off, 64);
    part[c] = pp;
  }
  __syncthreads();   // previous scale's norm2 readers are done
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 20; ++c) sh.red[wave][c] = part[c];
  }
  __syncthreads();
  if (t < 20) sh.norm2[t] = sh.red[0][t] + sh.red[1][t] + sh.red[2][t] + sh.red[3][t];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 20; ++c) {
    const float inv = 1.0f / sqrtf(fmaxf(sh.norm2[c], 1e-12f));
    v0[c] *= inv;
    v1[c] *= inv;
  }
  store20<DT>(out, o0, 20 * s, v0);
  store20<DT>(out, o1, 20 * s, v1);
}

// zero the padding channels [20*S, cstride) of this thread's two output rows so downstream GEMMs can read whole rows
template <int DT>
__device__ __forceinline__ void mups_zero_pad(void* __restrict__ out, int b, int S, int cstride, int t) {
  const int pad0 = 20 * S;
  if (pad0 < cstride / (is_x3<DT> ? kPairPlanes : 1)) {
    const int k = t & 7, j = (t >> 3) & 7, i0 = t >> 6;
    const int g0 = 64 * i0 + 8 * j + k, g1 = g0 + 256;
    zero_pad_row<DT>(out, ((size_t)b * kG + g0) * cstride, pad0, cstride);
    zero_pad_row<DT>(out, ((size_t)b * kG + g1) * cstride, pad0, cstride);
  }
}

template <int DT>
__global__ __launch_bounds__(kThreads) void mups_kernel(const float* __restrict__ points,
                                                        const int32_t* __restrict__ n_eff, int B,
                                                        int S, int P, void* __restrict__ out,
                                                        int cstride, float sigma, float w) {
  __shared__ MupsShared sh;
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  for (int s = 0; s < S; ++s) {
    const int m = n_eff[(size_t)b * S + s];
    const float* pts = points + ((size_t)b * S + s) * (size_t)P * 3;
    mups_one_scale<DT>(sh, [&](int row, int axis) { return pts[(size_t)row * 3 + axis]; }, m, P, b, s, out, cstride, sigma, w, t);
  }
  mups_zero_pad<DT>(out, b, S, cstride, t);
}

// The product path: ball query + subsample (patches_dev.h) feeding the MuPS sweep directly -- the [B, S*P, 3] patch
// tensor of the reference's data loader (utils/pcpnet_dataset.py:330-343 -> models/experts_n_est.py:26-35) is never
// written: the selected neighbour indices stay in LDS and a patch coordinate is formed, with the same IEEE subtraction
// and division, when the sweep stages it.  Bit-identical to patches_kernel + mups_kernel.
template <int DT>
__global__ __launch_bounds__(kThreads) void patches_mups_kernel(const PatchParams p, void* __restrict__ out, int cstride,
                                                                float sigma, float w) {
  __shared__ PatchShared psh;
  __shared__ MupsShared sh;
  const int q = blockIdx.x;
  const int t = threadIdx.x;
  float cf[3];
  patch_query_setup(p, psh, q, t, cf);
  for (int s = 0; s < p.S; ++s) {
    const int n_ball = psh.s_count[s];
    const int m = patch_select_scale(p, psh, q, t, s, cf);
    const float rad = p.rad_f[s];
    mups_one_scale<DT>(sh, [&](int row, int axis) { return row < m ? patch_coord(p, psh.sel[row], axis, cf[axis], rad) : 0.f; },
                       m, p.P, q, s, out, cstride, sigma, w, t);
    if (t == 0) {
      if (p.n_eff_out) p.n_eff_out[(size_t)q * p.S + s] = m;
      if (p.n_ball_out) p.n_ball_out[(size_t)q * p.S + s] = n_ball;
    }
  }
  mups_zero_pad<DT>(out, q, p.S, cstride, t);
}

// ---- 3^3 Gaussian grid (27 Gaussians, --num_gaussians 3: the reference's training default, train_n_est_w_experts.py:55)
// The same arithmetic on a small grid: one 64-thread workgroup per query, thread g < 27 owns Gaussian
// g = 9 i + 3 j + k; 64 patch points at a time are expanded into per-axis (q, d, d^2-1) terms in LDS.  Output rows:
// dense [B, 27, cstride] or, for the convolution towers, embedded in a 4^3 index space [B, 64, cstride] with row
// 16 i + 4 j + k (rows with a coordinate of 3 are written as zeros).
template <int DT>
__global__ __launch_bounds__(64) void mups3_kernel(const float* __restrict__ points, const int32_t* __restrict__ n_eff,
                                                   int B, int S, int P, void* __restrict__ out, int cstride,
                                                   float sigma, float w, float mu0, float mu1, float mu2, int embed) {
  __shared__ float4 stage[64][9];     // [point][3 * axis + grid index] = {q, d, d^2 - 1, -}
  __shared__ float norm2[20];
  const int b = blockIdx.x, t = threadIdx.x;
  const bool own = t < 27;
  const int gi = t / 9, gj = (t / 3) % 3, gk = t % 3;
  const int rows_pp = embed ? 64 : 27;
  const int row = embed ? 16 * gi + 4 * gj + gk : t;
  const float mus[3] = {mu0, mu1, mu2};
  for (int s = 0; s < S; ++s) {
    const int m = n_eff[(size_t)b * S + s];
    const size_t o0 = ((size_t)b * rows_pp + row) * cstride;
    float v[20];
    if (m <= 0) {     // zero-padded batch tail: skip, do not divide by 0
#pragma unroll
      for (int c = 0; c < 20; ++c) v[c] = 0.f;
      if (own) store20<DT>(out, o0, 20 * s, v);
      continue;
    }
    const int nrows = min(m + 1, P);               // mask = r > n_eff (utils/tf_util.py:693)
    const bool has_masked = nrows < P;
    const float* pts = points + ((size_t)b * S + s) * (size_t)P * 3;
    Stats1 st;
    st.sq = 0.f; st.mq = -INFINITY;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      st.mu_max[c] = -INFINITY; st.mu_min[c] = INFINITY; st.mu_sum[c] = 0.f;
      st.sg_max[c] = -INFINITY; st.sg_min[c] = INFINITY; st.sg_sum[c] = 0.f;
    }
    for (int c0 = 0; c0 < nrows; c0 += 64) {
      __syncthreads();
      if (c0 + t < nrows) {
#pragma unroll
        for (int axis = 0; axis < 3; ++axis) {
          const float x = pts[(size_t)(c0 + t) * 3 + axis];
          float d[3], e[3], sum = 0.f;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            d[i] = (x - mus[i]) / sigma;                       // utils/tf_util.py:687
            e[i] = expf(-0.5f * d[i] * d[i]);
            sum += e[i];
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) stage[t][3 * axis + i] = make_float4(e[i] / sum, d[i], d[i] * d[i] - 1.0f, 0.f);
        }
      }
      __syncthreads();
      const int cnt = min(64, nrows - c0);
      if (own) {
        for (int nl = 0; nl < cnt; ++nl) {
          const float4 X = stage[nl][gi], Y = stage[nl][3 + gj], Z = stage[nl][6 + gk];
          const float Q = X.x * (Y.x * Z.x);
          const float d[3] = {X.y, Y.y, Z.y}, e[3] = {X.z, Y.z, Z.z};
          st.sq += Q;
          st.mq = fmaxf(st.mq, Q);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float mm = Q * d[c], vv = Q * e[c];
            st.mu_max[c] = fmaxf(st.mu_max[c], mm); st.mu_min[c] = fminf(st.mu_min[c], mm); st.mu_sum[c] += mm;
            st.sg_max[c] = fmaxf(st.sg_max[c], vv); st.sg_min[c] = fminf(st.sg_min[c], vv); st.sg_sum[c] += vv;
          }
        }
      }
    }
    if (own) finish(st, nrows, has_masked, (float)m, w, v);
    else {
#pragma unroll
      for (int c = 0; c < 20; ++c) v[c] = 0.f;
    }
    // L2 normalisation over the 27 Gaussians, per channel (utils/tf_util.py:738-740)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 20; ++c) {
      float p2 = v[c] * v[c];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p2 += __shfl_xor(p2, off, 64);
      if (t == 0) norm2[c] = p2;
    }
    __syncthreads();
    if (own) {
#pragma unroll
      for (int c = 0; c < 20; ++c) v[c] *= 1.0f / sqrtf(fmaxf(norm2[c], 1e-12f));
      store20<DT>(out, o0, 20 * s, v);
    }
  }
  // padding channels [20 S, cstride) of the owned rows, and (embedded layout) the whole dead rows: zeros
  for (int r = t; r < rows_pp; r += 64) {
    const bool dead = embed && (((r >> 4) & 3) == 3 || ((r >> 2) & 3) == 3 || (r & 3) == 3);
    zero_pad_row<DT>(out, ((size_t)b * rows_pp + r) * cstride, dead ? 0 : 20 * S, cstride);
  }
}

}  // namespace

int launch_mups(const nesti_config_t* cfg, const float* points, const int32_t* n_eff, int B,
                void* out, int out_dtype, int out_cstride, int embed4, hipStream_t stream) {
  if (cfg->grid_n != kR && cfg->grid_n != 3) NESTI_FAIL("nesti_mups_forward: the Gaussian grid must be 8^3 or 3^3");
  if (cfg->n_scales < 1 || cfg->n_scales > NESTI_MAX_SCALES) NESTI_FAIL("nesti_mups_forward: bad n_scales");
  const bool x3 = act_planes(out_dtype) > 1;   // two planes per 64-channel group: out_cstride is the physical stride
  if (x3 && out_cstride % (kPairPlanes * kSplitGroup)) NESTI_FAIL("nesti_mups_forward: a pair-mode row is a whole number of 128-element groups");
  if (out_cstride / (x3 ? kPairPlanes : 1) < 20 * cfg->n_scales || (out_cstride % 4) != 0)
    NESTI_FAIL("nesti_mups_forward: out_cstride must be >= 20*S and a multiple of 4");
  if (out_dtype != NESTI_F32 && out_dtype != NESTI_BF16 && out_dtype != NESTI_F16 && !x3) NESTI_FAIL("nesti_mups_forward: unknown out_dtype");
  if (B <= 0) return 0;
  const float sigma = (float)sqrt(cfg->variance);     // np.sqrt in f64, then the f32 placeholder: test_n_est_w_experts.py:146
  const int S = cfg->n_scales, P = cfg->points_per_scale;
  if (cfg->grid_n == 3) {
    const float w = (float)(1.0 / 27.0);
    // np.mgrid[1/3 - 1 : 1 - 1/3 : 3j] in f64, fed through the f32 placeholder (utils/utils.py:81-87)
    const double a0 = 1.0 / 3 - 1.0, a1 = 1.0 - 1.0 / 3;
    const float mu0 = (float)a0, mu1 = (float)(a0 + (a1 - a0) * 1 / 2.0), mu2 = (float)(a0 + (a1 - a0) * 2 / 2.0);
    dim3 grid(B), block(64);
    if (out_dtype == NESTI_F32)
      hipLaunchKernelGGL(mups3_kernel<NESTI_F32>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w, mu0, mu1, mu2, embed4);
    else if (out_dtype == NESTI_BF16)
      hipLaunchKernelGGL(mups3_kernel<NESTI_BF16>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w, mu0, mu1, mu2, embed4);
    else if (out_dtype == NESTI_BF16X3)
      hipLaunchKernelGGL(mups3_kernel<NESTI_BF16X3>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w, mu0, mu1, mu2, embed4);
    else if (out_dtype == NESTI_F16X3)
      hipLaunchKernelGGL(mups3_kernel<NESTI_F16X3>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w, mu0, mu1, mu2, embed4);
    else
      hipLaunchKernelGGL(mups3_kernel<NESTI_F16>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w, mu0, mu1, mu2, embed4);
    NESTI_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const float w = 1.0f / (float)kG;                  // utils/utils.py:89
  dim3 grid(B), block(kThreads);
  if (out_dtype == NESTI_F32)
    hipLaunchKernelGGL(mups_kernel<NESTI_F32>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w);
  else if (out_dtype == NESTI_BF16)
    hipLaunchKernelGGL(mups_kernel<NESTI_BF16>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w);
  else if (out_dtype == NESTI_BF16X3)
    hipLaunchKernelGGL(mups_kernel<NESTI_BF16X3>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w);
  else if (out_dtype == NESTI_F16X3)
    hipLaunchKernelGGL(mups_kernel<NESTI_F16X3>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w);
  else
    hipLaunchKernelGGL(mups_kernel<NESTI_F16>, grid, block, 0, stream, points, n_eff, B, S, P, out, out_cstride, sigma, w);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_patches_mups(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                        const double* r_abs, uint64_t seed, int query_row0, const void* grid_ws_dev, void* out, int out_dtype,
                        int out_cstride, int32_t* n_eff_out_dev, hipStream_t stream) {
  if (cfg->grid_n != kR) NESTI_FAIL("launch_patches_mups: the fused kernel serves the 8^3 Gaussian grid");
  if (cfg->points_per_scale < 1 || 2 * cfg->points_per_scale > kListCap) NESTI_FAIL("launch_patches_mups: points_per_scale must be in [1, 512]");
  const bool x3 = act_planes(out_dtype) > 1;
  if (out_cstride / (x3 ? kPairPlanes : 1) < 20 * cfg->n_scales || (out_cstride % (x3 ? kPairPlanes * kSplitGroup : 4)) != 0)
    NESTI_FAIL("launch_patches_mups: bad channel stride");
  if (M <= 0) return 0;
  PatchParams p;
  patch_params_fill(&p, cfg, cloud_dev, N, query_idx_dev, M, r_abs, seed, query_row0, grid_ws_dev);
  p.n_eff_out = n_eff_out_dev;
  const float sigma = (float)sqrt(cfg->variance);
  const float w = 1.0f / (float)kG;
  dim3 grid(M), block(kThreads);
  if (out_dtype == NESTI_F32) hipLaunchKernelGGL(patches_mups_kernel<NESTI_F32>, grid, block, 0, stream, p, out, out_cstride, sigma, w);
  else if (out_dtype == NESTI_BF16) hipLaunchKernelGGL(patches_mups_kernel<NESTI_BF16>, grid, block, 0, stream, p, out, out_cstride, sigma, w);
  else if (out_dtype == NESTI_F16) hipLaunchKernelGGL(patches_mups_kernel<NESTI_F16>, grid, block, 0, stream, p, out, out_cstride, sigma, w);
  else if (out_dtype == NESTI_BF16X3) hipLaunchKernelGGL(patches_mups_kernel<NESTI_BF16X3>, grid, block, 0, stream, p, out, out_cstride, sigma, w);
  else if (out_dtype == NESTI_F16X3) hipLaunchKernelGGL(patches_mups_kernel<NESTI_F16X3>, grid, block, 0, stream, p, out, out_cstride, sigma, w);
  else NESTI_FAIL("launch_patches_mups: unknown out_dtype");
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace nesti
