// MFMA / LDS-DMA primitives shared by the conv kernels (conv.hip, conv8n.hip).  gfx950 only.
#pragma once
#include "kernels.h"

namespace nesti {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int DT> __device__ __forceinline__ void mma(f32x16& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma<NESTI_BF16>(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<NESTI_F16>(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<NESTI_F32>(f32x16& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

typedef __attribute__((address_space(3))) unsigned char* lptr_t;

// 16 B per lane, global -> LDS, asynchronous LDS-DMA.  Issued through inline asm so that hipcc does not
// track it: with the builtin the compiler parks an s_waitcnt vmcnt(0) in front of the next ds_read and the
// weight-tile latency is exposed on every tap.  Completion is waited for by hand (wait_vm0) before the
// barrier that publishes the tile.  lds_dst is a wave-uniform LDS byte address; lane i lands at lds_dst + 16 i.
__device__ __forceinline__ void glds16(const unsigned char* src, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace
}  // namespace nesti
