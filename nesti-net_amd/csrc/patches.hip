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
#include <string>

#include "kernels.h"
#include "patches_dev.h"

namespace nesti {
namespace {

constexpr int kThreads = kPatchThreads;

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

__global__ __launch_bounds__(kPatchThreads) void patches_kernel(const PatchParams p) {
  __shared__ PatchShared sh;
  const int q = blockIdx.x;
  const int t = threadIdx.x;
  float cf[3];
  patch_query_setup(p, sh, q, t, cf);
  for (int s = 0; s < p.S; ++s) {
    const int n_ball = sh.s_count[s];
    const int n_eff = patch_select_scale(p, sh, q, t, s, cf);
    const float rad = p.rad_f[s];
    for (int r = t; r < p.P; r += kPatchThreads) {
      const size_t row = ((size_t)q * p.S + s) * p.P + r;
      float ox = 0.f, oy = 0.f, oz = 0.f;
      int idx = -1;
      if (r < n_eff) {
        idx = sh.sel[r];
        ox = patch_coord(p, idx, 0, cf[0], rad);
        oy = patch_coord(p, idx, 1, cf[1], rad);
        oz = patch_coord(p, idx, 2, cf[2], rad);
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


// ---- the reference's own subsample order (utils/pcpnet_dataset.py:304, 320-321) on the GPU ---------------------------------
// scipy's cKDTree.query_ball_point returns a ball in ascending position in tree.indices (it visits `lesser` before `greater`
// and a leaf is a contiguous slice of tree.indices; tests/test_refreplay.py), so "traversal order" is a SORT KEY: the host builds
// the tree once per shape and uploads rank[i] = position of point i in tree.indices and order = tree.indices.  The random stream
// only needs the ball SIZES (patches_count_kernel), is replayed natively on the host (refreplay.cpp) and comes back as a pick
// table.  patches_ref_kernel then collects the ball of the largest radius once (key = rank << 4 | scales-containing-it mask),
// sorts it in LDS (bitonic), and per scale compacts the members of that scale's ball in rank order and applies
//     n <= P: the ball itself in traversal order, rows beyond n zero (:310-311, :298);   n > P: ball[picks]  (:320-321).
constexpr int kRefCap = 16384;          // largest ball (of the largest radius) the LDS sort holds; the host checks the counts first
constexpr int kRefThreads = 256;

struct RefParams {
  PatchParams p;
  const int32_t* rank;                  // [N] position of point i in cKDTree's tree.indices
  const int32_t* order;                 // [N] tree.indices
  const uint16_t* picks;                // pick table of this call (refreplay.cpp)
  const long long* pick_off;            // [M * S] offset of (query, scale)'s P picks, -1: the ball holds <= P points
};

__global__ __launch_bounds__(kPatchThreads) void patches_count_kernel(const PatchParams p) {
  __shared__ PatchShared sh;
  float cf[3];
  patch_query_setup(p, sh, blockIdx.x, threadIdx.x, cf);
  if (threadIdx.x < p.S) p.n_ball_out[(size_t)blockIdx.x * p.S + threadIdx.x] = sh.s_count[threadIdx.x];
}

__global__ __launch_bounds__(kRefThreads) void patches_ref_kernel(const RefParams rp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ref[];
  unsigned* keys = reinterpret_cast<unsigned*>(smem_ref);               // [kRefCap] sort keys
  unsigned* list = keys + kRefCap;                                      // [kRefCap] one scale's ball: ranks in traversal order
  __shared__ int span_beg[9], span_end[9], s_cnt, s_part[kRefThreads];
  const PatchParams& p = rp.p;
  const int q = blockIdx.x, t = threadIdx.x;
  int qi = p.query_idx ? p.query_idx[q] : p.row0 + q;
  qi = min(max(qi, 0), p.N - 1);
  const GridHeader h = *p.header;
  float cf[3] = {p.cloud[(size_t)qi * 3], p.cloud[(size_t)qi * 3 + 1], p.cloud[(size_t)qi * 3 + 2]};
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
    span_beg[t] = b;
    span_end[t] = e;
  }
  if (t == 0) s_cnt = 0;
  __syncthreads();
  // ---- collect every point inside at least one ball: key = rank << 4 | mask of the scales whose ball holds it ---------------
  for (int sp = 0; sp < 9; ++sp) {
    for (int i = span_beg[sp] + t; i < span_end[sp]; i += kRefThreads) {
      const float4 c = p.sorted[i];
      const double dx = (double)c.x - cx, dy = (double)c.y - cy, dz = (double)c.z - cz;
      double d2 = __dmul_rn(dx, dx);
      d2 = __dadd_rn(d2, __dmul_rn(dy, dy));
      d2 = __dadd_rn(d2, __dmul_rn(dz, dz));
      unsigned mask = 0u;
#pragma unroll
      for (int s = 0; s < NESTI_MAX_SCALES; ++s)
        if (s < p.S && d2 <= p.r2[s]) mask |= 1u << s;
      if (mask) {
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < kRefCap) keys[pos] = ((unsigned)rp.rank[__float_as_int(c.w)] << 4) | mask;
      }
    }
  }
  __syncthreads();
  const int n_all = s_cnt;
  if (n_all > kRefCap) {                 // refused on the host before the launch (nesti_patches_query_ref's caller holds the counts)
    for (int s = 0; s < p.S; ++s) {
      for (int r = t; r < p.P; r += kRefThreads) {
        const size_t row = ((size_t)q * p.S + s) * p.P + r;
        if (p.points_out) { p.points_out[row * 3] = 0.f; p.points_out[row * 3 + 1] = 0.f; p.points_out[row * 3 + 2] = 0.f; }
        if (p.nbr_out) p.nbr_out[row] = -1;
      }
      if (t == 0 && p.n_eff_out) p.n_eff_out[(size_t)q * p.S + s] = -1;
    }
    return;
  }
  int n2 = 1;
  while (n2 < n_all) n2 <<= 1;
  for (int i = n_all + t; i < n2; i += kRefThreads) keys[i] = 0xffffffffu;
  __syncthreads();
  // ---- bitonic sort, ascending ---------------------------------------------------------------------------------------------
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < (n2 >> 1); i += kRefThreads) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
        const unsigned a = keys[lo], b = keys[hi];
        const bool up = (lo & k) == 0;
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  // ---- per scale: compact the members of its ball (rank order), then gather ------------------------------------------------
  const int per = (n_all + kRefThreads - 1) / kRefThreads;
  const int c0 = min(n_all, t * per), c1 = min(n_all, c0 + per);
  for (int s = 0; s < p.S; ++s) {
    int mine = 0;
    for (int i = c0; i < c1; ++i) mine += (keys[i] >> s) & 1u;
    s_part[t] = mine;
    __syncthreads();
    for (int off = 1; off < kRefThreads; off <<= 1) {
      const int v = (t >= off) ? s_part[t - off] : 0;
      __syncthreads();
      s_part[t] += v;
      __syncthreads();
    }
    const int n_ball = s_part[kRefThreads - 1];
    int w = s_part[t] - mine;
    for (int i = c0; i < c1; ++i)
      if ((keys[i] >> s) & 1u) list[w++] = keys[i] >> 4;
    __syncthreads();
    const int n_eff = min(n_ball, p.P);                                   // utils/pcpnet_dataset.py:310
    const long long po = rp.pick_off ? rp.pick_off[(size_t)q * p.S + s] : -1;
    const float rad = p.rad_f[s];
    for (int r = t; r < p.P; r += kRefThreads) {
      const size_t row = ((size_t)q * p.S + s) * p.P + r;
      float ox = 0.f, oy = 0.f, oz = 0.f;
      int idx = -1;
      if (r < n_eff) {
        int pos = r;
        if (n_ball > p.P) pos = (po >= 0) ? min((int)rp.picks[po + r], n_ball - 1) : r;   // a missing pick row cannot happen
        idx = rp.order[list[pos]];                                                        // when the host replayed these counts
        ox = patch_coord(p, idx, 0, cf[0], rad);
        oy = patch_coord(p, idx, 1, cf[1], rad);
        oz = patch_coord(p, idx, 2, cf[2], rad);
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
    __syncthreads();                      // `list` and s_part are reused by the next scale
  }
}

}  // namespace
}  // namespace nesti

using namespace nesti;

extern "C" {

size_t nesti_patches_workspace_bytes(int N) { return N > 0 ? patch_ws_layout(N).total : 0; }

int nesti_patches_grid(const nesti_config_t* cfg, const float* cloud_dev, int N, const double* r_abs,
                       void* grid_ws_dev, size_t grid_ws_bytes, void* stream) {
  if (!cfg || !cloud_dev || !r_abs || !grid_ws_dev) NESTI_FAIL("nesti_patches_grid: null argument");
  if (N <= 0) NESTI_FAIL("nesti_patches_grid: empty cloud");
  if (cfg->n_scales < 1 || cfg->n_scales > NESTI_MAX_SCALES) NESTI_FAIL("nesti_patches_grid: bad n_scales");
  const WsLayout L = patch_ws_layout(N);
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
  const WsLayout L = patch_ws_layout(N);
  if (grid_ws_bytes < L.total) NESTI_FAIL("nesti_patches_query: grid workspace too small");
  if (M <= 0) return 0;
  // 'full' sampler (query_idx NULL): patch row == point index, so the row range must lie inside the cloud.  With a
  // query list the indices live on the device; the host mirror (provider.CloudPatches) validates them once at upload
  // and the kernel clamps defensively (a bad index then yields a wrong patch, never an out-of-bounds read).
  if (query_row0 < 0) NESTI_FAIL("nesti_patches_query: query_row0 must be >= 0");
  if (!query_idx_dev && (long long)query_row0 + M > (long long)N)
    NESTI_FAIL("nesti_patches_query: query rows [query_row0, query_row0 + M) exceed the cloud (N points)");
  for (int s = 0; s < cfg->n_scales; ++s)
    if (!(r_abs[s] > 0.0)) NESTI_FAIL("nesti_patches_query: radii must be positive");
  PatchParams p;
  patch_params_fill(&p, cfg, cloud_dev, N, query_idx_dev, M, r_abs, seed, query_row0, grid_ws_dev);
  p.points_out = points_out_dev; p.n_eff_out = n_eff_out_dev; p.nbr_out = nbr_idx_out_dev; p.n_ball_out = n_ball_out_dev;
  const int tok = prof_begin(NESTI_PROF_PATCHES, (hipStream_t)stream);
  hipLaunchKernelGGL(patches_kernel, dim3(M), dim3(kThreads), 0, (hipStream_t)stream, p);
  prof_end(NESTI_PROF_PATCHES, tok, (hipStream_t)stream);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

static int check_query_args(const char* who, const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev,
                            int M, const double* r_abs, int query_row0, const void* grid_ws_dev, size_t grid_ws_bytes) {
  const std::string w(who);
  if (!cfg || !cloud_dev || !r_abs || !grid_ws_dev) NESTI_FAIL(w + ": null argument");
  if (N <= 0) NESTI_FAIL(w + ": empty cloud");
  if (cfg->n_scales < 1 || cfg->n_scales > NESTI_MAX_SCALES) NESTI_FAIL(w + ": bad n_scales");
  if (cfg->points_per_scale < 1 || 2 * cfg->points_per_scale > kListCap) NESTI_FAIL(w + ": points_per_scale must be in [1, 512]");
  if (grid_ws_bytes < patch_ws_layout(N).total) NESTI_FAIL(w + ": grid workspace too small");
  if (query_row0 < 0) NESTI_FAIL(w + ": query_row0 must be >= 0");
  if (M > 0 && !query_idx_dev && (long long)query_row0 + M > (long long)N)
    NESTI_FAIL(w + ": query rows [query_row0, query_row0 + M) exceed the cloud (N points)");
  for (int s = 0; s < cfg->n_scales; ++s)
    if (!(r_abs[s] > 0.0)) NESTI_FAIL(w + ": radii must be positive");
  return 0;
}

int nesti_patches_count(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                        const double* r_abs, int query_row0, int32_t* n_ball_out_dev, const void* grid_ws_dev,
                        size_t grid_ws_bytes, void* stream) {
  if (check_query_args("nesti_patches_count", cfg, cloud_dev, N, query_idx_dev, M, r_abs, query_row0, grid_ws_dev, grid_ws_bytes)) return 1;
  if (M <= 0) return 0;
  if (!n_ball_out_dev) NESTI_FAIL("nesti_patches_count: null output");
  PatchParams p;
  patch_params_fill(&p, cfg, cloud_dev, N, query_idx_dev, M, r_abs, 0, query_row0, grid_ws_dev);
  p.n_ball_out = n_ball_out_dev;
  const int tok = prof_begin(NESTI_PROF_PATCHES, (hipStream_t)stream);
  hipLaunchKernelGGL(patches_count_kernel, dim3(M), dim3(kThreads), 0, (hipStream_t)stream, p);
  prof_end(NESTI_PROF_PATCHES, tok, (hipStream_t)stream);
  NESTI_CHECK_HIP(hipGetLastError());
  return 0;
}

int nesti_patches_ref_max_ball(void) { return kRefCap; }

int nesti_patches_query_ref(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                            const double* r_abs, int query_row0, const int32_t* tree_rank_dev, const int32_t* tree_order_dev,
                            const uint16_t* picks_dev, const int64_t* pick_offsets_dev, float* points_out_dev,
                            int32_t* n_eff_out_dev, int32_t* nbr_idx_out_dev, const void* grid_ws_dev, size_t grid_ws_bytes,
                            void* stream) {
  if (check_query_args("nesti_patches_query_ref", cfg, cloud_dev, N, query_idx_dev, M, r_abs, query_row0, grid_ws_dev, grid_ws_bytes)) return 1;
  if (M <= 0) return 0;
  if (!tree_rank_dev || !tree_order_dev || !pick_offsets_dev) NESTI_FAIL("nesti_patches_query_ref: null tree order / pick offsets");
  RefParams rp;
  patch_params_fill(&rp.p, cfg, cloud_dev, N, query_idx_dev, M, r_abs, 0, query_row0, grid_ws_dev);
  rp.p.points_out = points_out_dev; rp.p.n_eff_out = n_eff_out_dev; rp.p.nbr_out = nbr_idx_out_dev;
  rp.rank = tree_rank_dev; rp.order = tree_order_dev; rp.picks = picks_dev; rp.pick_off = (const long long*)pick_offsets_dev;
  constexpr int lds = 2 * kRefCap * (int)sizeof(unsigned);
  static bool attr_set[64] = {};
  int dev = 0;
  NESTI_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    NESTI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&patches_ref_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int tok = prof_begin(NESTI_PROF_PATCHES, (hipStream_t)stream);
  hipLaunchKernelGGL(patches_ref_kernel, dim3(M), dim3(kRefThreads), lds, (hipStream_t)stream, rp);
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
