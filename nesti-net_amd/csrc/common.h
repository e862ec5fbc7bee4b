// Shared helpers for libnesti_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/nesti_hip.h"

// Timing-only switches (conv.hip: IGEMM_X2_NOFILL / IGEMM_NO_MAIN / IGEMM_EMPTY / IGEMM_NO_EPILOGUE / IGEMM_NO_STORE, conv8n.hip:
// CONV8N_DOUBLE_MMA) build a library that computes WRONG results on purpose (they time one phase of a kernel).  They exist only
// behind ONE guard: a build that sets any of them without -DNESTI_TIMING_EXPERIMENTS does not compile, and a build with the guard
// says so in nesti_version(), which nesti-net_amd/_lib.py refuses to load unless NESTI_ALLOW_TIMING_BUILD=1 is set.
#if !defined(NESTI_TIMING_EXPERIMENTS) && (defined(IGEMM_X2_NOFILL) || defined(IGEMM_NO_MAIN) || defined(IGEMM_EMPTY) || \
                                          defined(IGEMM_NO_EPILOGUE) || defined(IGEMM_NO_STORE) || defined(CONV8N_DOUBLE_MMA))
#error "timing-only switches need -DNESTI_TIMING_EXPERIMENTS (the resulting library computes wrong results and says so in nesti_version())"
#endif

namespace nesti {

void set_error(const std::string& msg);

#define NESTI_CHECK_HIP(expr)                                                         \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      ::nesti::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));          \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

#define NESTI_FAIL(msg)           \
  do {                            \
    ::nesti::set_error(msg);      \
    return 1;                     \
  } while (0)

static inline size_t dtype_size(int dt) { return dt == NESTI_F32 ? 4 : 2; }
// NESTI_BF16X3: the kernels are the bf16 ones with the pair K loop (conv.hip / conv8n.hip: X3); an activation row holds, per
// group of 64 channels, the two 64-element planes [hi | lo] (128 elements), a packed weight row [W_hi | W_lo] per K chunk,
// and one set of fragment reads feeds hi*W_hi + lo*W_hi + hi*W_lo.  Writers emit the planes (split_col / split_pack2 below).
// NESTI_F16X3: the same with f16 pairs and the f16 kernels.
// NESTI_F16X3C is NESTI_F16X3 everywhere except in the gating net's first pass (model.hip: gate_cascade)
static inline int main_dtype(int dt) { return (dt == NESTI_F16X3C || dt == NESTI_F16X8 || dt == NESTI_F16X8C) ? NESTI_F16X3 : dt; }
static inline bool dtype_cascade(int dt) { return dt == NESTI_F16X3C || dt == NESTI_F16X8C; }
static inline bool dtype_x8(int dt) { return dt == NESTI_F16X8 || dt == NESTI_F16X8C; }
static inline int kernel_dtype(int dt) { return dt == NESTI_BF16X3 ? NESTI_BF16 : dt == NESTI_F16X3 ? NESTI_F16 : dt; }
constexpr int kPairPlanes = 2;    // hi, lo
static inline int act_planes(int dt) { return (dt == NESTI_BF16X3 || dt == NESTI_F16X3) ? kPairPlanes : 1; }
constexpr int kSplitGroup = 64;
__host__ __device__ __forceinline__ int split_col(int col) { return (col >> 6) * (kPairPlanes * kSplitGroup) + (col & (kSplitGroup - 1)); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- element conversion (device) ------------------------------------------
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, quiet NaN): no branches per element.
typedef __attribute__((ext_vector_type(2))) float nesti_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 nesti_bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 nesti_f16x2;
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  return *reinterpret_cast<uint16_t*>(&h);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) {
  _Float16 h = *reinterpret_cast<_Float16*>(&b);
  return (float)h;
}

template <int DT> struct Elem;
template <> struct Elem<NESTI_F32> {
  using T = float;
  static __device__ __forceinline__ T from_f32(float f) { return f; }
  static __device__ __forceinline__ float to_f32(T v) { return v; }
};
template <> struct Elem<NESTI_BF16> {
  using T = uint16_t;
  static __device__ __forceinline__ T from_f32(float f) { return f32_to_bf16_bits(f); }
  static __device__ __forceinline__ float to_f32(T v) { return bf16_bits_to_f32(v); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // two elements in one dword, lo at the lower address
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((nesti_f32x2){lo, hi}, nesti_bf16x2));
  }
};
template <> struct Elem<NESTI_F16> {
  using T = uint16_t;
  static __device__ __forceinline__ T from_f32(float f) { return f32_to_f16_bits(f); }
  static __device__ __forceinline__ float to_f32(T v) { return f16_bits_to_f32(v); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((nesti_f32x2){lo, hi}, nesti_f16x2));
  }
};

// (hi, lo) pair of two values in the 16-bit type E: hi = rne(v), lo = rne(v - hi) (v - hi is exact in fp32)
template <class E>
__device__ __forceinline__ void split_pack2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = E::pack2(a, b);
  lo = E::pack2(a - E::to_f32((uint16_t)(hi & 0xffffu)), b - E::to_f32((uint16_t)(hi >> 16)));
}

// Four / eight consecutive 16-bit channels of one activation row, starting at LOGICAL column col (a multiple of 4 / 8,
// so a vector never straddles a 64-channel group); row_elems = row * physical channel stride.
template <class E>
__device__ __forceinline__ void store_act4(unsigned char* base, long long row_elems, int col, float a, float b, float c, float d, int split) {
  if (!split) {
    *reinterpret_cast<uint2*>(base + (row_elems + col) * 2) = make_uint2(E::pack2(a, b), E::pack2(c, d));
    return;
  }
  uint32_t h0, l0, h1, l1;
  split_pack2<E>(a, b, h0, l0);
  split_pack2<E>(c, d, h1, l1);
#ifdef NESTI_EXPERIMENT_XW             // measurement builds only (model.hip: gate_mix == 2): the value rounded to 16 bits, lo = 0
  if (split == 2) l0 = l1 = 0u;
#endif
  unsigned char* d0 = base + (row_elems + split_col(col)) * 2;
  *reinterpret_cast<uint2*>(d0) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(d0 + 2 * kSplitGroup) = make_uint2(l0, l1);
}
template <class E>
__device__ __forceinline__ void store_act8(unsigned char* base, long long row_elems, int col, const float4& f0, const float4& f1, int split) {
  if (!split) {
    *reinterpret_cast<uint4*>(base + (row_elems + col) * 2) =
        make_uint4(E::pack2(f0.x, f0.y), E::pack2(f0.z, f0.w), E::pack2(f1.x, f1.y), E::pack2(f1.z, f1.w));
    return;
  }
  uint4 h, l;
  split_pack2<E>(f0.x, f0.y, h.x, l.x);
  split_pack2<E>(f0.z, f0.w, h.y, l.y);
  split_pack2<E>(f1.x, f1.y, h.z, l.z);
  split_pack2<E>(f1.z, f1.w, h.w, l.w);
#ifdef NESTI_EXPERIMENT_XW
  if (split == 2) l = make_uint4(0u, 0u, 0u, 0u);
#endif
  unsigned char* d0 = base + (row_elems + split_col(col)) * 2;
  *reinterpret_cast<uint4*>(d0) = h;
  *reinterpret_cast<uint4*>(d0 + 2 * kSplitGroup) = l;
}

// ---- e4m3 planes of the FP8 cross terms (kernels.h: ConvParams::aux8_out; conv8n.hip X8) ----------------------------------------
// Four values -> four OCP e4m3 bytes (v_cvt_pk_fp8_f32: round to nearest even, subnormals kept), saturated at +-448 (the
// conversion itself would produce NaN above the format's range).
__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
// The side-buffer row of one activation row: per 64-channel group [lo8 64 B | hi8 64 B]; `col` = logical column (multiple of 4),
// v = the activated fp32 values, hi = f16(v): lo8 = e4m3((v - hi) mul_lo), hi8 = e4m3(v mul_hi), mul_* = the layer's power-of-two pre-scales.
__device__ __forceinline__ void store_aux8_4(unsigned char* aux_row, int col, float a, float b, float c, float d, float mul_lo, float mul_hi) {
  const float ha = f16_bits_to_f32(f32_to_f16_bits(a)), hb = f16_bits_to_f32(f32_to_f16_bits(b));
  const float hc = f16_bits_to_f32(f32_to_f16_bits(c)), hd = f16_bits_to_f32(f32_to_f16_bits(d));
  unsigned char* dst = aux_row + (col >> 6) * (2 * kSplitGroup) + (col & (kSplitGroup - 1));
  *reinterpret_cast<uint32_t*>(dst) = pack4_e4m3((a - ha) * mul_lo, (b - hb) * mul_lo, (c - hc) * mul_lo, (d - hd) * mul_lo);
  *reinterpret_cast<uint32_t*>(dst + kSplitGroup) = pack4_e4m3(a * mul_hi, b * mul_hi, c * mul_hi, d * mul_hi);
}
__device__ __forceinline__ void store_aux8_8(unsigned char* aux_row, int col, const float4& f0, const float4& f1, float mul_lo, float mul_hi) {
  store_aux8_4(aux_row, col, f0.x, f0.y, f0.z, f0.w, mul_lo, mul_hi);
  store_aux8_4(aux_row, col + 4, f1.x, f1.y, f1.z, f1.w, mul_lo, mul_hi);
}

// The FP6 form of the same side-buffer chunk (kernels.h: ConvParams::x8_fmt == 6): 16 consecutive channels of one row -> 32 e2m3 elements
// under ONE power-of-two scale + that scale as an E8M0 byte.  v_cvt_scalef32_2xpk16_fp6_f32 writes src0[i] to slot 2i and src1[i] to slot
// 2i + 1, divides by `scale`, rounds to nearest even and saturates at +-7.5 (scripts/fp6_probe.hip).  `col` = logical column, multiple of 16.
typedef float f32x16c_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x6c_t __attribute__((ext_vector_type(6)));
__device__ __forceinline__ void store_aux6_16(unsigned char* aux_row, int col, const float4& f0, const float4& f1, const float4& f2, const float4& f3) {
  const float v[16] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w, f3.x, f3.y, f3.z, f3.w};
  f32x16c_t lo, hi;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float h = f16_bits_to_f32(f32_to_f16_bits(v[i]));
    hi[i] = h;
    lo[i] = (v[i] - h) * 2048.f;
    amax = fmaxf(amax, fabsf(h));
  }
  int sb = (int)(__float_as_uint(amax) >> 23) - 2;       // the largest element lands in [4, 8) (7.5 < x < 8 saturates)
  sb = sb < 1 ? 1 : sb;                                   // an all-zero chunk: any scale
  const u32x6c_t r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo, hi, __uint_as_float((unsigned)sb << 23));
  unsigned char* dst = aux_row + (col >> 6) * (2 * kSplitGroup) + (col & (kSplitGroup - 1));
  *reinterpret_cast<uint4*>(dst) = make_uint4(r[0], r[1], r[2], r[3]);
  *reinterpret_cast<uint4*>(dst + kSplitGroup) = make_uint4(r[4], r[5], (unsigned)sb, 0u);
}

// the pair modes as store types (mups.hip): 16-bit elements, two planes
template <> struct Elem<NESTI_BF16X3> : Elem<NESTI_BF16> {};
template <> struct Elem<NESTI_F16X3> : Elem<NESTI_F16> {};
template <int DT> constexpr bool is_x3 = (DT == NESTI_BF16X3 || DT == NESTI_F16X3);

// host-side conversions used by the weight repacker
uint16_t host_f32_to_bf16(float f);
uint16_t host_f32_to_f16(float f);
float host_f16_to_f32(uint16_t h);

// ---- optional kernel timing (model.hip) -----------------------------------------
void prof_phase(int phase);                                 // NESTI_PHASE_*: what the following launches are booked under
int prof_begin(int category, hipStream_t st);              // returns a token for prof_end (-1: nothing recorded)
void prof_end(int category, int token, hipStream_t st);

// ---- launchers implemented in the .hip files -------------------------------
// embed4: 3^3 grid only -- write rows in a 4^3 index space (64 rows per point, dead rows zero) for the conv towers
int launch_mups(const nesti_config_t* cfg, const float* points, const int32_t* n_eff, int B,
                void* out, int out_dtype, int out_cstride, int embed4, hipStream_t stream);

// ball query + subsample + MuPS in one kernel (mups.hip: patches_mups_kernel); 8^3 grid.  n_eff_out_dev may be NULL.
int launch_patches_mups(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                        const double* r_abs, uint64_t seed, int query_row0, const void* grid_ws_dev, void* out, int out_dtype,
                        int out_cstride, int32_t* n_eff_out_dev, hipStream_t stream);

}  // namespace nesti
