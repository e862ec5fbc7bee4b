// Probe (gfx950): what v_cvt_scalef32_2xpk16_fp6_f32 writes, and how v_mfma_scale_f32_32x32x64_f8f6f4 reads FP6 (e2m3) operands.
//   hipcc --offload-arch=gfx950 -O2 scripts/fp6_probe.hip -o scripts/fp6_probe.bin && ./scripts/fp6_probe.bin
// Questions:
//   1. element order of the conversion's 32 outputs (src0[i], src1[i] -> which 6-bit slots), rounding, saturation, meaning of `scale`;
//   2. MFMA with cbsz = blgp = 2: element j of a lane's K block at bits [6j, 6j + 6) of v[0:5]; lanes 32-63 = K block 1; per-lane E8M0
//      scale taken from byte `opsel` of the scale VGPR; dwords 6-7 of the operand ignored (so a scale byte may live there).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));

__global__ void cvt_probe(const float* a, const float* b, float scale, unsigned* out) {
  v16f x, y;
  for (int i = 0; i < 16; ++i) { x[i] = a[i]; y[i] = b[i]; }
  v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x, y, scale);
  if (threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}

// A codes [32 rows][64 k], B codes [64 k][32 cols]; scale bytes per (row, kblock) / (col, kblock)
__global__ void mfma_probe(const unsigned char* A, const unsigned char* B, const unsigned char* sA, const unsigned char* sB, int garbage,
                           int opsel_case, float* C) {
  const int lane = threadIdx.x, r = lane & 31, kb = lane >> 5;
  unsigned long long bits[3] = {0, 0, 0}, bitsb[3] = {0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const unsigned long long ca = A[r * 64 + kb * 32 + j] & 63, cb = B[(kb * 32 + j) * 32 + r] & 63;
    const int pos = 6 * j;
    bits[pos >> 6] |= ca << (pos & 63);
    if ((pos & 63) > 58) bits[(pos >> 6) + 1] |= ca >> (64 - (pos & 63));
    bitsb[pos >> 6] |= cb << (pos & 63);
    if ((pos & 63) > 58) bitsb[(pos >> 6) + 1] |= cb >> (64 - (pos & 63));
  }
  v8i a, b;
  for (int i = 0; i < 3; ++i) {
    a[2 * i] = (int)(unsigned)bits[i]; a[2 * i + 1] = (int)(unsigned)(bits[i] >> 32);
    b[2 * i] = (int)(unsigned)bitsb[i]; b[2 * i + 1] = (int)(unsigned)(bitsb[i] >> 32);
  }
  // dwords 6-7: the scale byte in byte 0 of dword 6 (what a kernel reading 32-byte rows would hold there) + garbage elsewhere
  const int sa = sA[r * 2 + kb], sb = sB[r * 2 + kb];
  a[6] = sa | (garbage & 0xffffff00); a[7] = garbage;
  b[6] = sb | (garbage & 0xffffff00); b[7] = garbage;
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  if (opsel_case == 0)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, a[6], 0, b[6]);
  else {   // scale in byte 2 of a separate VGPR
    const int va = (sa << 16) | 0x7f00007f, vb = (sb << 16) | 0x7f00007f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 2, va, 2, vb);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * kb;
    C[row * 32 + r] = c[i];
  }
}

static float dec(int code) {
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}

int main() {
  // ---- 1. the conversion ----
  float ha[16], hb[16];
  const float probe_a[16] = {0.f, 0.125f, 0.1875f, 0.3125f, 1.f, 1.0625f, 1.1875f, 2.f, 3.75f, 3.9f, 7.5f, 7.8f, 100.f, -1.f, -0.06f, -7.75f};
  const float probe_b[16] = {0.5f, 0.5625f, 0.4375f, 1.5f, 2.5f, 5.f, 6.f, 7.f, 7.25f, -2.25f, -2.125f, -2.375f, 0.0624f, 0.0626f, 1e-9f, -3.f};
  memcpy(ha, probe_a, sizeof ha); memcpy(hb, probe_b, sizeof hb);
  float *da, *db; unsigned* dout;
  hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 24);
  for (float scale : {1.f, 4.f, 0.25f}) {
    for (int i = 0; i < 16; ++i) { ha[i] = probe_a[i] * scale; hb[i] = probe_b[i] * scale; }
    hipMemcpy(da, ha, 64, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64, hipMemcpyHostToDevice);
    cvt_probe<<<1, 64>>>(da, db, scale, dout);
    unsigned o[6]; hipMemcpy(o, dout, 24, hipMemcpyDeviceToHost);
    unsigned long long w[3] = {o[0] | ((unsigned long long)o[1] << 32), o[2] | ((unsigned long long)o[3] << 32), o[4] | ((unsigned long long)o[5] << 32)};
    printf("cvt scale %g (inputs = probe x scale):\n  slot: value   (src0 = a[i], src1 = b[i])\n", scale);
    for (int j = 0; j < 32; ++j) {
      const int pos = 6 * j;
      unsigned long long c = w[pos >> 6] >> (pos & 63);
      if ((pos & 63) > 58) c |= w[(pos >> 6) + 1] << (64 - (pos & 63));
      printf("  %2d: %6.3f%s", j, dec((int)(c & 63)), (j & 3) == 3 ? "\n" : "");
    }
    printf("  a: "); for (int i = 0; i < 16; ++i) printf("%g ", probe_a[i]); printf("\n  b: "); for (int i = 0; i < 16; ++i) printf("%g ", probe_b[i]); printf("\n");
  }
  // ---- 2. the MFMA ----
  std::vector<unsigned char> A(32 * 64), B(64 * 32), sA(64), sB(64);
  for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) A[r * 64 + k] = (unsigned char)((r * 7 + k * 3 + (k >> 5) * 11) % 64);
  for (int k = 0; k < 64; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = (unsigned char)((k * 5 + n * 13 + 1) % 64);
  for (int r = 0; r < 32; ++r) for (int kb = 0; kb < 2; ++kb) { sA[r * 2 + kb] = (unsigned char)(127 + (r % 3) - kb); sB[r * 2 + kb] = (unsigned char)(124 + (r % 2) + 2 * kb); }
  unsigned char *dA, *dB, *dsA, *dsB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsA, 64); hipMalloc(&dsB, 64); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dsA, sA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsB, sB.data(), 64, hipMemcpyHostToDevice);
  for (int cs = 0; cs < 2; ++cs)
    for (int garbage : {0, (int)0xdeadbeef}) {
      mfma_probe<<<1, 64>>>(dA, dB, dsA, dsB, garbage, cs, dC);
      std::vector<float> C(1024); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      int bad = 0; double worst = 0;
      for (int r = 0; r < 32; ++r) for (int n = 0; n < 32; ++n) {
        double ref = 0;
        for (int kb = 0; kb < 2; ++kb) {
          double s = 0;
          for (int j = 0; j < 32; ++j) s += (double)dec(A[r * 64 + kb * 32 + j] & 63) * dec(B[(kb * 32 + j) * 32 + n] & 63);
          ref += s * exp2((double)sA[r * 2 + kb] - 127) * exp2((double)sB[n * 2 + kb] - 127);
        }
        const double d = fabs(ref - C[r * 32 + n]);
        if (d > 1e-6 * (1 + fabs(ref))) ++bad;
        if (d > worst) worst = d;
      }
      printf("mfma fp6: scale source %s, dwords 6-7 %s: %d of 1024 outputs differ from the model (worst |diff| %.3g)  C[0][0] %.6g C[5][7] %.6g\n",
             cs ? "byte 2 of a separate VGPR" : "byte 0 of operand dword 6", garbage ? "garbage" : "clean", bad, worst, C[0], C[5 * 32 + 7]);
    }
  return 0;
}
