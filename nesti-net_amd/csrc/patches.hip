// Multi-scale patch extraction on the GPU: the job utils/pcpnet_dataset.py:286-343
// (__getitem__ with center='point', use_pca=False, point_tuple=1) does per query with
// scipy.spatial.cKDTree.query_ball_point + RandomState.choice.
//
// Data structure: a uniform grid with cell edge >= the largest radius, built per cloud
// (bbox -> counts -> scan -> fill; points re-ordered into cell order as float4 {x,y,z,index}
// so a query streams contiguous memory).  One 256-thread workgroup per query visits the
// 3x3x3 cell block as 9 contiguous x-spans, tests every candidate once against all scales in
// fp64 exactly as cKDTree does (d2 = dx*dx, += dy*dy, += dz*dz, no FMA; d2 <= r*r), and keeps
// per scale the P hits with the smallest (hash, index) keys, in key order.
#include <string.h>

#include <algorithm>

#include "kernels.h"

namespace nesti {
namespace {

constexpr int kMaxDim = 128;
constexpr int kMaxCells = kMaxDim * kMaxDim * kMaxDim;
constexpr int kThreads = 256;
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
WsLayout ws_layout(int N) {
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

__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void bbox_init_kernel(unsigned* bb) {
  if (threadIdx.x < 3) bb[threadIdx.x] = 0xffffffffu;
  else if (threadIdx.x < 6) bb[threadIdx.x] = 0u;
}

__global__ void bbox_kernel(const float* __restrict__ cloud, int N, unsigned* bb) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = cloud[(size_t)i * 3 + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], off, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      atomicMin(&bb[c], f2ord(mn[c]));
      atomicMax(&bb[3 + c], f2ord(mx[c]));
    }
  }
}

__global__ void header_kernel(const unsigned* bb, double cell_min, GridHeader* h) {
  double ext = 0.0;
  for (int c = 0; c < 3; ++c) {
    h->minv[c] = (double)ord2f(bb[c]);
    ext = fmax(ext, (double)ord2f(bb[3 + c]) - (double)ord2f(bb[c]));
  }
  double cell = fmax(cell_min, ext / (double)(kMaxDim - 1));
  if (!(cell > 0.0)) cell = 1.0;
  h->inv_cell = 1.0 / cell;
  int n = 1;
  for (int c = 0; c < 3; ++c) {
    int d = (int)floor(((double)ord2f(bb[3 + c]) - h->minv[c]) * h->inv_cell) + 1;
    d = max(1, min(kMaxDim, d));
    h->dims[c] = d;
    n *= d;
  }
  h->ncells = n;
}

__device__ __forceinline__ void cell_coords(const GridHeader& h, float x, float y, float z, int* ix, int* iy, int* iz) {
  *ix = min(h.dims[0] - 1, max(0, (int)floor(((double)x - h.minv[0]) * h.inv_cell)));
  *iy = min(h.dims[1] - 1, max(0, (int)floor(((double)y - h.minv[1]) * h.inv_cell)));
  *iz = min(h.dims[2] - 1, max(0, (int)floor(((double)z - h.minv[2]) * h.inv_cell)));
}
__device__ __forceinline__ int cell_flat(const GridHeader& h, int ix, int iy, int iz) {
  return (iz * h.dims[1] + iy) * h.dims[0] + ix;   // x fastest: a row of cells is one contiguous span
}

__global__ void count_kernel(const float* __restrict__ cloud, int N, const GridHeader* hp, int* count) {
  const GridHeader h = *hp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    int ix, iy, iz;
    cell_coords(h, cloud[(size_t)i * 3], cloud[(size_t)i * 3 + 1], cloud[(size_t)i * 3 + 2], &ix, &iy, &iz);
    atomicAdd(&count[cell_flat(h, ix, iy, iz)], 1);
  }
}

// exclusive scan of count[0..ncells) into start[0..ncells]; one 1024-thread block
__global__ void scan_kernel(const int* __restrict__ count, int* __restrict__ start, int* __restrict__ cursor,
                            const GridHeader* hp) {
  __shared__ int part[1024];
  const int n = hp->ncells;
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b = min(n, t * per), e = min(n, b + per);
  int s = 0;
  for (int i = b; i < e; ++i) s += count[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = b; i < e; ++i) {
    start[i] = run;
    cursor[i] = run;
    run += count[i];
  }
  if (t == 1023) start[n] = part[1023];
}

__global__ void fill_kernel(const float* __restrict__ cloud, int N, const GridHeader* hp, int* cursor,
                            float4* __restrict__ sorted) {
  const GridHeader h = *hp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const float x = cloud[(size_t)i * 3], y = cloud[(size_t)i * 3 + 1], z = cloud[(size_t)i * 3 + 2];
    int ix, iy, iz;
    cell_coords(h, x, y, z, &ix, &iy, &iz);
    const int pos = atomicAdd(&cursor[cell_flat(h, ix, iy, iz)], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float(i));
  }
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

__global__ __launch_bounds__(kThreads) void patches_kernel(const PatchParams p) {
  __shared__ int span_beg[9], span_end[9];
  __shared__ int s_count[NESTI_MAX_SCALES];
  __shared__ int s_cnt;
  __shared__ unsigned long long keys[kListCap];
  __shared__ int sel[kListCap];

  const int q = blockIdx.x;
  const int t = threadIdx.x;
  int qi = p.query_idx ? p.query_idx[q] : p.row0 + q;         // 'full' sampler: patch row == point index
  qi = min(max(qi, 0), p.N - 1);
  const GridHeader h = *p.header;
  const float cxf = p.cloud[(size_t)qi * 3], cyf = p.cloud[(size_t)qi * 3 + 1], czf = p.cloud[(size_t)qi * 3 + 2];
  const double cx = cxf, cy = cyf, cz = czf;

  if (t < 9) {
    int ix, iy, iz;
    cell_coords(h, cxf, cyf, czf, &ix, &iy, &iz);
    const int zz = iz + t / 3 - 1, yy = iy + t % 3 - 1;
    int b = 0, e = 0;
    if (zz >= 0 && zz < h.dims[2] && yy >= 0 && yy < h.dims[1]) {
      const int x0 = max(ix - 1, 0), x1 = min(ix + 1, h.dims[0] - 1);
      b = p.start[cell_flat(h, x0, yy, zz)];
      e = p.start[cell_flat(h, x1, yy, zz) + 1];
    }
    span_beg[t] = b;
    span_end[t] = e;
  }
  if (t < NESTI_MAX_SCALES) s_count[t] = 0;
  __syncthreads();

  // ---- pass A: ball sizes ------------------------------------------------------------------
  int local[NESTI_MAX_SCALES] = {0, 0, 0, 0};
  for (int sp = 0; sp < 9; ++sp) {
    for (int i = span_beg[sp] + t; i < span_end[sp]; i += kThreads) {
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
    if (s < p.S && local[s]) atomicAdd(&s_count[s], local[s]);
  __syncthreads();

  for (int s = 0; s < p.S; ++s) {
    const int n_ball = s_count[s];
    const int n_eff = min(n_ball, p.P);   // utils/pcpnet_dataset.py:310
    // ---- pass B: collect the hits whose key is <= T; T is bisected until P <= kept <= cap --
    unsigned lo = 0u, hi = 0xffffffffu, T = 0xffffffffu;
    if (n_ball > p.P) T = (unsigned)fmin(4294967295.0, 4294967296.0 * 1.25 * (double)p.P / (double)n_ball);
    int kept = 0;
    for (int iter = 0; iter < 40; ++iter) {
      __syncthreads();
      if (t == 0) s_cnt = 0;
      __syncthreads();
      for (int sp = 0; sp < 9; ++sp) {
        for (int i = span_beg[sp] + t; i < span_end[sp]; i += kThreads) {
          const float4 c = p.sorted[i];
          const double dx = (double)c.x - cx, dy = (double)c.y - cy, dz = (double)c.z - cz;
          double d2 = __dmul_rn(dx, dx);
          d2 = __dadd_rn(d2, __dmul_rn(dy, dy));
          d2 = __dadd_rn(d2, __dmul_rn(dz, dz));
          if (d2 <= p.r2[s]) {
            const unsigned idx = (unsigned)__float_as_int(c.w);
            const unsigned hsh = subsample_hash(p.seed, (unsigned)(p.row0 + q), (unsigned)s, idx);
            if (hsh <= T) {
              const int pos = atomicAdd(&s_cnt, 1);
              if (pos < kListCap) keys[pos] = ((unsigned long long)hsh << 32) | idx;
            }
          }
        }
      }
      __syncthreads();
      kept = s_cnt;
      if (kept >= n_eff && kept <= kListCap) break;
      if (kept < n_eff) lo = T + 1u; else hi = T - 1u;
      T = lo + (hi - lo) / 2u;
    }
    kept = min(kept, kListCap);
    // ---- rank sort: position = number of smaller keys; keep the first n_eff -----------------
    for (int e = t; e < kept; e += kThreads) {
      const unsigned long long my = keys[e];
      int rank = 0;
      for (int j = 0; j < kept; ++j) rank += (keys[j] < my) ? 1 : 0;
      if (rank < n_eff) sel[rank] = (int)(unsigned)(my & 0xffffffffull);
    }
    __syncthreads();
    const float rad = p.rad_f[s];
    for (int r = t; r < p.P; r += kThreads) {
      const size_t row = ((size_t)q * p.S + s) * p.P + r;
      float ox = 0.f, oy = 0.f, oz = 0.f;
      int idx = -1;
      if (r < n_eff) {
        idx = sel[r];
        // (pts[idx] - pts[center]) / rad in f32  (utils/pcpnet_dataset.py:330-343)
        ox = __fdiv_rn(__fsub_rn(p.cloud[(size_t)idx * 3], cxf), rad);
        oy = __fdiv_rn(__fsub_rn(p.cloud[(size_t)idx * 3 + 1], cyf), rad);
        oz = __fdiv_rn(__fsub_rn(p.cloud[(size_t)idx * 3 + 2], czf), rad);
      }
      if (p.points_out) {
        p.points_out[row * 3] = ox;
        p.points_out[row * 3 + 1] = oy;
        p.points_out[row * 3 + 2] = oz;
      }
      if (p.nbr_out) p.nbr_out[row] = idx;
    }
    if (t == 0) {
      if (p.n_eff_out) p.n_eff_out[(size_t)q * p.S + s] = n_eff;
      if (p.n_ball_out) p.n_ball_out[(size_t)q * p.S + s] = n_ball;
    }
  }
}

}  // namespace
}  // namespace nesti

using namespace nesti;

extern "C" {

size_t nesti_patches_workspace_bytes(int N) { return N > 0 ? ws_layout(N).total : 0; }

int nesti_patches_grid(const nesti_config_t* cfg, const float* cloud_dev, int N, const double* r_abs,
                       void* grid_ws_dev, size_t grid_ws_bytes, void* stream) {
  if (!cfg || !cloud_dev || !r_abs || !grid_ws_dev) NESTI_FAIL("nesti_patches_grid: null argument");
  if (N <= 0) NESTI_FAIL("nesti_patches_grid: empty cloud");
  if (cfg->n_scales < 1 || cfg->n_scales > NESTI_MAX_SCALES) NESTI_FAIL("nesti_patches_grid: bad n_scales");
  const WsLayout L = ws_layout(N);
  if (grid_ws_bytes < L.total) NESTI_FAIL("nesti_patches_grid: grid workspace too small");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)grid_ws_dev;
  GridHeader* header = (GridHeader*)(ws + L.header);
  unsigned* bb = (unsigned*)(ws + L.bbox);
  int* count = (int*)(ws + L.count);
  int* start = (int*)(ws + L.start);
  int* cursor = (int*)(ws + L.cursor);
  float4* sorted = (float4*)(ws + L.sorted);
  double rmax = 0.0;
  for (int s = 0; s < cfg->n_scales; ++s) {
    if (!(r_abs[s] > 0.0)) NESTI_FAIL("nesti_patches_grid: radii must be positive");
    rmax = fmax(rmax, r_abs[s]);
  }
  const int gb = std::min(1024, (N + 255) / 256);
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, st, bb);
  hipLaunchKernelGGL(bbox_kernel, dim3(gb), dim3(256), 0, st, cloud_dev, N, bb);
  hipLaunchKernelGGL(header_kernel, dim3(1), dim3(1), 0, st, bb, rmax * 1.0001, header);
  NESTI_CHECK_HIP(hipMemsetAsync(count, 0, (size_t)(kMaxCells + 1) * 4, st));
  hipLaunchKernelGGL(count_kernel, dim3(gb), dim3(256), 0, st, cloud_dev, N, header, count);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, count, start, cursor, header);
  hipLaunchKernelGGL(fill_kernel, dim3(gb), dim3(256), 0, st, cloud_dev, N, header, cursor, sorted);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int nesti_patches_query(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                        const double* r_abs, uint64_t seed, int query_row0, float* points_out_dev,
                        int32_t* n_eff_out_dev, int32_t* nbr_idx_out_dev, int32_t* n_ball_out_dev,
                        const void* grid_ws_dev, size_t grid_ws_bytes, void* stream) {
  if (!cfg || !cloud_dev || !r_abs || !grid_ws_dev) NESTI_FAIL("nesti_patches_query: null argument");
  if (N <= 0) NESTI_FAIL("nesti_patches_query: empty cloud");
  if (cfg->n_scales < 1 || cfg->n_scales > NESTI_MAX_SCALES) NESTI_FAIL("nesti_patches_query: bad n_scales");
  if (cfg->points_per_scale < 1 || 2 * cfg->points_per_scale > kListCap)
    NESTI_FAIL("nesti_patches_query: points_per_scale must be in [1, 512]");
  const WsLayout L = ws_layout(N);
  if (grid_ws_bytes < L.total) NESTI_FAIL("nesti_patches_query: grid workspace too small");
  if (M <= 0) return 0;
  // 'full' sampler (query_idx NULL): patch row == point index, so the row range must lie inside the cloud.  With a
  // query list the indices live on the device; the host mirror (provider.CloudPatches) validates them once at upload
  // and the kernel clamps defensively (a bad index then yields a wrong patch, never an out-of-bounds read).
  if (query_row0 < 0) NESTI_FAIL("nesti_patches_query: query_row0 must be >= 0");
  if (!query_idx_dev && (long long)query_row0 + M > (long long)N)
    NESTI_FAIL("nesti_patches_query: query rows [query_row0, query_row0 + M) exceed the cloud (N points)");
  const unsigned char* ws = (const unsigned char*)grid_ws_dev;
  PatchParams p;
  memset(&p, 0, sizeof(p));
  p.cloud = cloud_dev;
  p.sorted = (const float4*)(ws + L.sorted);
  p.start = (const int*)(ws + L.start);
  p.header = (const GridHeader*)(ws + L.header);
  p.query_idx = query_idx_dev;
  p.M = M; p.N = N; p.S = cfg->n_scales; p.P = cfg->points_per_scale; p.seed = seed; p.row0 = query_row0;
  for (int s = 0; s < cfg->n_scales; ++s) {
    if (!(r_abs[s] > 0.0)) NESTI_FAIL("nesti_patches_query: radii must be positive");
    p.r2[s] = r_abs[s] * r_abs[s];
    p.rad_f[s] = (float)r_abs[s];
  }
  p.points_out = points_out_dev; p.n_eff_out = n_eff_out_dev; p.nbr_out = nbr_idx_out_dev; p.n_ball_out = n_ball_out_dev;
  const int tok = prof_begin(NESTI_PROF_PATCHES, (hipStream_t)stream);
  hipLaunchKernelGGL(patches_kernel, dim3(M), dim3(kThreads), 0, (hipStream_t)stream, p);
  prof_end(NESTI_PROF_PATCHES, tok, (hipStream_t)stream);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int nesti_patches_build(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                        const double* r_abs, uint64_t seed, int query_row0, float* points_out_dev,
                        int32_t* n_eff_out_dev, int32_t* nbr_idx_out_dev, int32_t* n_ball_out_dev, void* grid_ws_dev,
                        size_t grid_ws_bytes, void* stream) {
  if (nesti_patches_grid(cfg, cloud_dev, N, r_abs, grid_ws_dev, grid_ws_bytes, stream)) return 1;
  return nesti_patches_query(cfg, cloud_dev, N, query_idx_dev, M, r_abs, seed, query_row0, points_out_dev,
                             n_eff_out_dev, nbr_idx_out_dev, n_ball_out_dev, grid_ws_dev, grid_ws_bytes, stream);
}

}  // extern "C"
