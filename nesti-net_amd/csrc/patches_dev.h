// Device-side pieces of the multi-scale patch extraction (utils/pcpnet_dataset.py:286-343) shared by patches_kernel
// (patches.hip: materialises the patch tensors, the parity entry point) and patches_mups_kernel (mups.hip: feeds the
// selected neighbours straight into the MuPS sweep, the product path).  See patches.hip for the data structure.
#pragma once
#include <string.h>

#include "kernels.h"

namespace nesti {
namespace {

constexpr int kMaxDim = 128;
constexpr int kMaxCells = kMaxDim * kMaxDim * kMaxDim;
constexpr int kPatchThreads = 256;
constexpr int kListCap = 1024;   // candidates kept for the final rank sort (>= 2P for P = 512)

struct GridHeader {
  double minv[3];
  double inv_cell;
  int dims[3];
  int ncells;
};

struct WsLayout {
  size_t header, bbox, count, start, cursor, sorted, total;
};
inline WsLayout patch_ws_layout(int N) {
  WsLayout L;
  size_t o = 0;
  L.header = o; o += 256;
  L.bbox = o; o += 256;
  L.count = o; o += align_up((size_t)(kMaxCells + 1) * 4, 256);
  L.start = o; o += align_up((size_t)(kMaxCells + 1) * 4, 256);
  L.cursor = o; o += align_up((size_t)(kMaxCells + 1) * 4, 256);
  L.sorted = o; o += align_up((size_t)N * 16, 256);
  L.total = o;
  return L;
}

__device__ __forceinline__ void cell_coords(const GridHeader& h, float x, float y, float z, int* ix, int* iy, int* iz) {
  *ix = min(h.dims[0] - 1, max(0, (int)floor(((double)x - h.minv[0]) * h.inv_cell)));
  *iy = min(h.dims[1] - 1, max(0, (int)floor(((double)y - h.minv[1]) * h.inv_cell)));
  *iz = min(h.dims[2] - 1, max(0, (int)floor(((double)z - h.minv[2]) * h.inv_cell)));
}
__device__ __forceinline__ int cell_flat(const GridHeader& h, int ix, int iy, int iz) {
  return (iz * h.dims[1] + iy) * h.dims[0] + ix;   // x fastest: a row of cells is one contiguous span
}

// splitmix64 finaliser over (seed, query, scale, point): the documented subsample key
__device__ __forceinline__ unsigned subsample_hash(unsigned long long seed, unsigned q, unsigned s, unsigned idx) {
  unsigned long long z = seed ^ ((unsigned long long)q << 34) ^ ((unsigned long long)s << 32) ^ (unsigned long long)idx;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (unsigned)(z >> 32);
}

struct PatchParams {
  const float* cloud;
  const float4* sorted;
  const int* start;
  const GridHeader* header;
  const int32_t* query_idx;
  int M, N, S, P, row0;
  unsigned long long seed;
  double r2[NESTI_MAX_SCALES];     // r*r, like cKDTree's upper_bound for p = 2
  float rad_f[NESTI_MAX_SCALES];   // (float)r : torch divides the f32 patch by the scalar in f32
  float* points_out;
  int32_t* n_eff_out;
  int32_t* nbr_out;
  int32_t* n_ball_out;
};


// Shared-memory state of one query (one 256-thread workgroup)
struct PatchShared {
  int span_beg[9], span_end[9];
  int s_count[NESTI_MAX_SCALES];
  int s_cnt;
  unsigned long long keys[kListCap];
  int sel[kListCap];
};

// The query point, the 3 x 3 cell block as nine contiguous x-spans of the cell-ordered copy, and the ball sizes of
// every scale (pass A).  Ends with a barrier: sh.s_count[] is valid on return.
__device__ __forceinline__ void patch_query_setup(const PatchParams& p, PatchShared& sh, int q, int t, float (&cf)[3]) {
  int qi = p.query_idx ? p.query_idx[q] : p.row0 + q;         // 'full' sampler: patch row == point index
  qi = min(max(qi, 0), p.N - 1);
  const GridHeader h = *p.header;
  cf[0] = p.cloud[(size_t)qi * 3]; cf[1] = p.cloud[(size_t)qi * 3 + 1]; cf[2] = p.cloud[(size_t)qi * 3 + 2];
  const double cx = cf[0], cy = cf[1], cz = cf[2];
  if (t < 9) {
    int ix, iy, iz;
    cell_coords(h, cf[0], cf[1], cf[2], &ix, &iy, &iz);
    const int zz = iz + t / 3 - 1, yy = iy + t % 3 - 1;
    int b = 0, e = 0;
    if (zz >= 0 && zz < h.dims[2] && yy >= 0 && yy < h.dims[1]) {
      const int x0 = max(ix - 1, 0), x1 = min(ix + 1, h.dims[0] - 1);
      b = p.start[cell_flat(h, x0, yy, zz)];
      e = p.start[cell_flat(h, x1, yy, zz) + 1];
    }
    sh.span_beg[t] = b;
    sh.span_end[t] = e;
  }
  if (t < NESTI_MAX_SCALES) sh.s_count[t] = 0;
  __syncthreads();
  // ---- pass A: ball sizes ------------------------------------------------------------------
  int local[NESTI_MAX_SCALES] = {0, 0, 0, 0};
  for (int sp = 0; sp < 9; ++sp) {
    for (int i = sh.span_beg[sp] + t; i < sh.span_end[sp]; i += kPatchThreads) {
      const float4 c = p.sorted[i];
      const double dx = (double)c.x - cx, dy = (double)c.y - cy, dz = (double)c.z - cz;
      double d2 = __dmul_rn(dx, dx);
      d2 = __dadd_rn(d2, __dmul_rn(dy, dy));
      d2 = __dadd_rn(d2, __dmul_rn(dz, dz));
#pragma unroll
      for (int s = 0; s < NESTI_MAX_SCALES; ++s)
        if (s < p.S && d2 <= p.r2[s]) ++local[s];
    }
  }
#pragma unroll
  for (int s = 0; s < NESTI_MAX_SCALES; ++s)
    if (s < p.S && local[s]) atomicAdd(&sh.s_count[s], local[s]);
  __syncthreads();
}

// Scale s: the n_eff = min(ball, P) neighbours with the smallest (hash, index) keys, in key order, into sh.sel[]
// (pass B + rank sort).  Ends with a barrier: sh.sel[0 .. n_eff) is valid on return.  Returns n_eff.
__device__ __forceinline__ int patch_select_scale(const PatchParams& p, PatchShared& sh, int q, int t, int s, const float (&cf)[3]) {
  const double cx = cf[0], cy = cf[1], cz = cf[2];
  const int n_ball = sh.s_count[s];
  const int n_eff = min(n_ball, p.P);   // utils/pcpnet_dataset.py:310
  // ---- pass B: collect the hits whose key is <= T; T is bisected until P <= kept <= cap --
  unsigned lo = 0u, hi = 0xffffffffu, T = 0xffffffffu;
  if (n_ball > p.P) T = (unsigned)fmin(4294967295.0, 4294967296.0 * 1.25 * (double)p.P / (double)n_ball);
  int kept = 0;
  for (int iter = 0; iter < 40; ++iter) {
    __syncthreads();
    if (t == 0) sh.s_cnt = 0;
    __syncthreads();
    for (int sp = 0; sp < 9; ++sp) {
      for (int i = sh.span_beg[sp] + t; i < sh.span_end[sp]; i += kPatchThreads) {
        const float4 c = p.sorted[i];
        const double dx = (double)c.x - cx, dy = (double)c.y - cy, dz = (double)c.z - cz;
        double d2 = __dmul_rn(dx, dx);
        d2 = __dadd_rn(d2, __dmul_rn(dy, dy));
        d2 = __dadd_rn(d2, __dmul_rn(dz, dz));
        if (d2 <= p.r2[s]) {
          const unsigned idx = (unsigned)__float_as_int(c.w);
          const unsigned hsh = subsample_hash(p.seed, (unsigned)(p.row0 + q), (unsigned)s, idx);
          if (hsh <= T) {
            const int pos = atomicAdd(&sh.s_cnt, 1);
            if (pos < kListCap) sh.keys[pos] = ((unsigned long long)hsh << 32) | idx;
          }
        }
      }
    }
    __syncthreads();
    kept = sh.s_cnt;
    if (kept >= n_eff && kept <= kListCap) break;
    if (kept < n_eff) lo = T + 1u; else hi = T - 1u;
    T = lo + (hi - lo) / 2u;
  }
  kept = min(kept, kListCap);
  // ---- rank sort: position = number of smaller keys; keep the first n_eff -----------------
  for (int e = t; e < kept; e += kPatchThreads) {
    const unsigned long long my = sh.keys[e];
    int rank = 0;
    for (int j = 0; j < kept; ++j) rank += (sh.keys[j] < my) ? 1 : 0;
    if (rank < n_eff) sh.sel[rank] = (int)(unsigned)(my & 0xffffffffull);
  }
  __syncthreads();
  return n_eff;
}

// coordinate `axis` of neighbour idx relative to the query, scaled: (pts[idx] - pts[center]) / rad in f32 with IEEE
// subtraction and division (utils/pcpnet_dataset.py:330-343)
__device__ __forceinline__ float patch_coord(const PatchParams& p, int idx, int axis, float c, float rad) {
  return __fdiv_rn(__fsub_rn(p.cloud[(size_t)idx * 3 + axis], c), rad);
}

// host side: fill the kernel parameter block from the C-ABI arguments (validated by the caller)
inline void patch_params_fill(PatchParams* p, const nesti_config_t* cfg, const float* cloud_dev, int N,
                              const int32_t* query_idx_dev, int M, const double* r_abs, uint64_t seed, int query_row0,
                              const void* grid_ws_dev) {
  const WsLayout L = patch_ws_layout(N);
  const unsigned char* ws = (const unsigned char*)grid_ws_dev;
  memset(p, 0, sizeof(*p));
  p->cloud = cloud_dev;
  p->sorted = (const float4*)(ws + L.sorted);
  p->start = (const int*)(ws + L.start);
  p->header = (const GridHeader*)(ws + L.header);
  p->query_idx = query_idx_dev;
  p->M = M; p->N = N; p->S = cfg->n_scales; p->P = cfg->points_per_scale; p->seed = seed; p->row0 = query_row0;
  for (int s = 0; s < cfg->n_scales; ++s) {
    p->r2[s] = r_abs[s] * r_abs[s];
    p->rad_f[s] = (float)r_abs[s];
  }
}

}  // namespace
}  // namespace nesti
