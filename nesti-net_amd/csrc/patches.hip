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

int nesti_patches_build(const nesti_config_t* cfg, const float* cloud_dev, int N, const int32_t* query_idx_dev, int M,
                        const double* r_abs, uint64_t seed, int query_row0, float* points_out_dev,
                        int32_t* n_eff_out_dev, int32_t* nbr_idx_out_dev, int32_t* n_ball_out_dev, void* grid_ws_dev,
                        size_t grid_ws_bytes, void* stream) {
  if (nesti_patches_grid(cfg, cloud_dev, N, r_abs, grid_ws_dev, grid_ws_bytes, stream)) return 1;
  return nesti_patches_query(cfg, cloud_dev, N, query_idx_dev, M, r_abs, seed, query_row0, points_out_dev,
                             n_eff_out_dev, nbr_idx_out_dev, n_ball_out_dev, grid_ws_dev, grid_ws_bytes, stream);
}

}  // extern "C"
