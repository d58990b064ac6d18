// LiDAR front end: deterministic hard voxelization, dynamic voxelization, mean VFE and the
// sparse -> dense scatter.
//
// Contract (mmdet3d 0.18.1 Voxelization, deterministic hard path; reached from
// models/detectors/unibev_detector.py:163-167):
//   c_j = floor((p_j - min_j) / size_j) in f32 (subtract, IEEE divide, floor), point dropped unless
//   0 <= c_j < grid_j on all three axes; voxels are numbered in order of FIRST APPEARANCE in the
//   input; each keeps its first `max_points` points in input order; voxels whose number would reach
//   `max_voxels` are dropped; coors are stored (z, y, x).
//
// The CUDA op the reference reaches does this with an O(N^2) duplicate scan and a single-thread
// serial kernel.  Here it is five short data-parallel passes with the same output:
//   1. coords -> 32-bit linear voxel key; insert (key -> min point index) into an open-addressing
//      hash table with 64-bit atomicMin (the minimum point index of a voxel IS its first appearance);
//   2. head flags: point i is a head iff it is its voxel's minimum index;
//   3. exclusive scan of the head flags over the input order = voxel number by first appearance;
//   4. every point pushes its index through its voxel's T-slot "keep the T smallest" chain
//      (slot t: old = atomicMin(slot, x); x = max(old, x)) — order-independent, so deterministic;
//   5. gather: voxel v copies points slots[v][0..T) (ascending = input order) and counts them.
#include "ubv_common.h"

namespace ubv {

constexpr uint32_t kBadKey = 0xFFFFFFFFu;
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kSlotEmpty = 0x7f7f7f7f;     // memset(0x7f) pattern, larger than any point index

struct VoxGeom {
  float min[3], size[3];
  int grid[3];
};

__device__ __forceinline__ uint32_t hash_u32(uint32_t k) {
  k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
  return k;
}

// Voxel coordinate of one point; false when outside the grid (or NaN).
__device__ __forceinline__ bool point_coord(const float* __restrict__ p, const VoxGeom& g,
                                            int (&c)[3]) {
#pragma clang fp contract(off)
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float f = floorf((p[j] - g.min[j]) / g.size[j]);
    const bool in = (f >= 0.0f) && (f < (float)g.grid[j]);     // false for NaN
    c[j] = in ? (int)f : -1;
    ok = ok && in;
  }
  return ok;
}

__device__ __forceinline__ void vox_key_insert_body(const int bid, 
    const float* __restrict__ points, int N, int F, VoxGeom g, uint32_t* __restrict__ keys,
    unsigned long long* __restrict__ table, uint32_t mask) {
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c[3];
  const bool ok = point_coord(points + (long)i * F, g, c);
  const uint32_t key = ok ? (uint32_t)((c[2] * g.grid[1] + c[1]) * g.grid[0] + c[0]) : kBadKey;
  keys[i] = key;
  if (!ok) return;
  const unsigned long long packed = ((unsigned long long)key << 32) | (uint32_t)i;
  uint32_t h = hash_u32(key) & mask;
  for (;;) {
    unsigned long long cur = table[h];
    if (cur == kEmpty) {
      cur = atomicCAS(&table[h], kEmpty, packed);
      if (cur == kEmpty) return;
    }
    if ((uint32_t)(cur >> 32) == key) { atomicMin(&table[h], packed); return; }
    h = (h + 1) & mask;
  }
}

// head[i] = 1 iff point i is the first point of its voxel; first[i] = first point of i's voxel.
__device__ __forceinline__ void vox_head_body(const int bid, 
    const uint32_t* __restrict__ keys, int N, const unsigned long long* __restrict__ table,
    uint32_t mask, int* __restrict__ first, int* __restrict__ head) {
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t key = keys[i];
  int f = -1;
  if (key != kBadKey) {
    uint32_t h = hash_u32(key) & mask;
    for (;;) {
      const unsigned long long cur = table[h];
      if ((uint32_t)(cur >> 32) == key) { f = (int)(uint32_t)cur; break; }
      h = (h + 1) & mask;
    }
  }
  first[i] = f;
  head[i] = (f == i) ? 1 : 0;
}

// ---- exclusive scan over int32 (3 kernels: per-block, block sums, add) ------------------------------
constexpr int kScanBlock = 1024;

__device__ __forceinline__ void scan_block_body(const int bid, const int* __restrict__ in,
                                                         int* __restrict__ out,
                                                         int* __restrict__ sums, int N) {
  __shared__ int wave_tot[4];
  const int base = bid * kScanBlock + threadIdx.x * 4;
  int v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[k] = (base + k < N) ? in[base + k] : 0; s += v[k]; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wv; ++w) off += wave_tot[w];
  int run = off + incl - s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < N) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) sums[bid] = off + incl;
}

__device__ __forceinline__ void scan_sums_body(int* __restrict__ sums, int nblocks, int* __restrict__ total,
                                 int clamp) {
  // one wave, sequential over chunks of 64 block sums
  const int lane = threadIdx.x;
  int carry = 0;
  for (int base = 0; base < nblocks; base += 64) {
    const int v = (base + lane < nblocks) ? sums[base + lane] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (base + lane < nblocks) sums[base + lane] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) *total = carry < clamp ? carry : clamp;
}

// Adds the block offsets and, per point, pushes it into its voxel's slot chain.
__device__ __forceinline__ void vox_assign_body(const int bid, 
    const uint32_t* __restrict__ keys, const int* __restrict__ first,
    const int* __restrict__ scan, const int* __restrict__ sums, int N, VoxGeom g, int max_points,
    int max_voxels, int* __restrict__ slots, int32_t* __restrict__ coors) {
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int f = first[i];
  if (f < 0) return;
  const int v = scan[f] + sums[f / kScanBlock];
  if (v >= max_voxels) return;
  if (f == i) {
    const uint32_t key = keys[i];
    const int cx = (int)(key % (uint32_t)g.grid[0]);
    const int cy = (int)((key / (uint32_t)g.grid[0]) % (uint32_t)g.grid[1]);
    const int cz = (int)(key / ((uint32_t)g.grid[0] * (uint32_t)g.grid[1]));
    coors[3 * (long)v] = cz; coors[3 * (long)v + 1] = cy; coors[3 * (long)v + 2] = cx;
  }
  int x = i;
  int* s = slots + (long)v * max_points;
  for (int t = 0; t < max_points; ++t) {
    const int old = atomicMin(&s[t], x);
    x = old > x ? old : x;
    if (x == kSlotEmpty) break;
  }
}

__device__ __forceinline__ void vox_gather_body(const int bid, 
    const float* __restrict__ points, int F, const int* __restrict__ slots,
    const int* __restrict__ voxel_num, int max_points, float* __restrict__ voxels,
    int32_t* __restrict__ num_points, int max_voxels) {
  // one thread per (voxel, slot, feature)
  const long t = (long)bid * blockDim.x + threadIdx.x;
  const long per_v = (long)max_points * F;
  if (t >= (long)max_voxels * per_v) return;
  const int v = (int)(t / per_v);
  const int r = (int)(t - (long)v * per_v);
  const int slot = r / F, f = r - slot * F;
  const bool live = v < *voxel_num;
  const int idx = live ? slots[(long)v * max_points + slot] : kSlotEmpty;
  voxels[t] = (idx != kSlotEmpty) ? points[(long)idx * F + f] : 0.0f;
  if (r == 0) {
    int n = 0;
    if (live)
      for (int k = 0; k < max_points; ++k) n += slots[(long)v * max_points + k] != kSlotEmpty;
    num_points[v] = n;
  }
}


// ---- kernels: one sample (blockIdx.x walks the sample) and a batch (blockIdx.y = sample; every per-sample array at
// a fixed stride, point counts and cloud pointers in VoxBatch) — a batch is ONE launch chain instead of one per cloud
constexpr int kVoxMaxBatch = 16;
struct VoxBatch {
  const float* points[kVoxMaxBatch];
  int n[kVoxMaxBatch];
  long ws_stride;                 // bytes of workspace per sample
  long key_off, first_off, head_off, scan_off, sums_off, table_off, slots_off;   // offsets inside a sample's workspace
  uint32_t mask;                  // table capacity - 1 (sized for the largest cloud)
};

__global__ __launch_bounds__(256) void vox_key_insert_kernel(const float* __restrict__ points, int N, int F, VoxGeom g,
                                                             uint32_t* __restrict__ keys,
                                                             unsigned long long* __restrict__ table, uint32_t mask) {
  vox_key_insert_body(blockIdx.x, points, N, F, g, keys, table, mask);
}
__global__ __launch_bounds__(256) void vox_head_kernel(const uint32_t* __restrict__ keys, int N,
                                                       const unsigned long long* __restrict__ table, uint32_t mask,
                                                       int* __restrict__ first, int* __restrict__ head) {
  vox_head_body(blockIdx.x, keys, N, table, mask, first, head);
}
__global__ __launch_bounds__(256) void scan_block_kernel(const int* __restrict__ in, int* __restrict__ out,
                                                         int* __restrict__ sums, int N) {
  scan_block_body(blockIdx.x, in, out, sums, N);
}
__global__ void scan_sums_kernel(int* __restrict__ sums, int nblocks, int* __restrict__ total, int clamp) {
  scan_sums_body(sums, nblocks, total, clamp);
}
__global__ __launch_bounds__(256) void vox_assign_kernel(const uint32_t* __restrict__ keys, const int* __restrict__ first,
                                                         const int* __restrict__ scan, const int* __restrict__ sums, int N,
                                                         VoxGeom g, int max_points, int max_voxels, int* __restrict__ slots,
                                                         int32_t* __restrict__ coors) {
  vox_assign_body(blockIdx.x, keys, first, scan, sums, N, g, max_points, max_voxels, slots, coors);
}
__global__ __launch_bounds__(256) void vox_gather_kernel(const float* __restrict__ points, int F, const int* __restrict__ slots,
                                                         const int* __restrict__ voxel_num, int max_points,
                                                         float* __restrict__ voxels, int32_t* __restrict__ num_points,
                                                         int max_voxels) {
  vox_gather_body(blockIdx.x, points, F, slots, voxel_num, max_points, voxels, num_points, max_voxels);
}

#define UBV_VOX_WS(type, off) ((type*)(ws + (long)blockIdx.y * vb.ws_stride + vb.off))
__global__ __launch_bounds__(256) void vox_key_insert_batch_kernel(VoxBatch vb, char* ws, int F, VoxGeom g) {
  vox_key_insert_body(blockIdx.x, vb.points[blockIdx.y], vb.n[blockIdx.y], F, g, UBV_VOX_WS(uint32_t, key_off),
                      UBV_VOX_WS(unsigned long long, table_off), vb.mask);
}
__global__ __launch_bounds__(256) void vox_head_batch_kernel(VoxBatch vb, char* ws) {
  vox_head_body(blockIdx.x, UBV_VOX_WS(uint32_t, key_off), vb.n[blockIdx.y], UBV_VOX_WS(unsigned long long, table_off),
                vb.mask, UBV_VOX_WS(int, first_off), UBV_VOX_WS(int, head_off));
}
__global__ __launch_bounds__(256) void scan_block_batch_kernel(VoxBatch vb, char* ws) {
  scan_block_body(blockIdx.x, UBV_VOX_WS(int, head_off), UBV_VOX_WS(int, scan_off), UBV_VOX_WS(int, sums_off),
                  vb.n[blockIdx.y]);
}
__global__ void scan_sums_batch_kernel(VoxBatch vb, char* ws, int* __restrict__ voxel_num, int clamp) {
  const int n = vb.n[blockIdx.y];
  scan_sums_body(UBV_VOX_WS(int, sums_off), (n + kScanBlock - 1) / kScanBlock, voxel_num + blockIdx.y, clamp);
}
__global__ __launch_bounds__(256) void vox_assign_batch_kernel(VoxBatch vb, char* ws, VoxGeom g, int max_points,
                                                               int max_voxels, int32_t* __restrict__ coors) {
  vox_assign_body(blockIdx.x, UBV_VOX_WS(uint32_t, key_off), UBV_VOX_WS(int, first_off), UBV_VOX_WS(int, scan_off),
                  UBV_VOX_WS(int, sums_off), vb.n[blockIdx.y], g, max_points, max_voxels, UBV_VOX_WS(int, slots_off),
                  coors + (long)blockIdx.y * max_voxels * 3);
}
__global__ __launch_bounds__(256) void vox_gather_batch_kernel(VoxBatch vb, char* ws, int F, const int* __restrict__ voxel_num,
                                                               int max_points, float* __restrict__ voxels,
                                                               int32_t* __restrict__ num_points, int max_voxels) {
  vox_gather_body(blockIdx.x, vb.points[blockIdx.y], F, UBV_VOX_WS(int, slots_off), voxel_num + blockIdx.y, max_points,
                  voxels + (long)blockIdx.y * max_voxels * max_points * F, num_points + (long)blockIdx.y * max_voxels,
                  max_voxels);
}
#undef UBV_VOX_WS

__global__ __launch_bounds__(256) void dynamic_voxelize_kernel(const float* __restrict__ points,
                                                               int N, int F, VoxGeom g,
                                                               int32_t* __restrict__ coors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c[3];
  const bool ok = point_coord(points + (long)i * F, g, c);
  coors[3 * (long)i] = ok ? c[2] : -1;
  coors[3 * (long)i + 1] = ok ? c[1] : -1;
  coors[3 * (long)i + 2] = ok ? c[0] : -1;
}

__global__ __launch_bounds__(256) void voxel_mean_kernel(const float* __restrict__ voxels,
                                                         const int32_t* __restrict__ num_points,
                                                         const int32_t* __restrict__ voxel_num,
                                                         float* __restrict__ mean, int max_voxels,
                                                         int T, int F) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)max_voxels * F) return;
  const int v = (int)(t / F), f = (int)(t - (long)v * F);
  if (voxel_num != nullptr && v >= *voxel_num) return;
  float s = 0.0f;
  for (int k = 0; k < T; ++k) s += voxels[((long)v * T + k) * F + f];
  const int np_ = num_points[v];
  mean[t] = np_ > 0 ? s / (float)np_ : 0.0f;      // (rows past the voxel count of a padded batch hold no points)
}

__global__ __launch_bounds__(256) void sparse_to_dense_kernel(
    const float* __restrict__ feats, const int32_t* __restrict__ coors,
    const int32_t* __restrict__ m_dev, int m, float* __restrict__ dense, int B, int C, int D, int Hs,
    int Ws) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int M = m_dev ? *m_dev : m;
  if (t >= (long)M * C) return;
  const int i = (int)(t / C), c = (int)(t - (long)i * C);
  const int b = coors[4 * (long)i], z = coors[4 * (long)i + 1], y = coors[4 * (long)i + 2],
            x = coors[4 * (long)i + 3];
  if (b < 0 || b >= B || z < 0 || z >= D || y < 0 || y >= Hs || x < 0 || x >= Ws) return;
  dense[((((long)b * C + c) * D + z) * Hs + y) * Ws + x] = feats[t];
}

static bool make_geom(const float* vs, const float* rg, VoxGeom& g) {
  for (int j = 0; j < 3; ++j) {
    g.min[j] = rg[j];
    g.size[j] = vs[j];
    // mmdet3d: grid = round((max - min) / size)
    g.grid[j] = (int)__builtin_roundf((rg[3 + j] - rg[j]) / vs[j]);
    if (!(vs[j] > 0.0f) || g.grid[j] <= 0) return false;
  }
  return (double)g.grid[0] * g.grid[1] * g.grid[2] < 4294967295.0;
}

static uint32_t table_capacity(int N) {
  uint32_t cap = 1024;
  while (cap < 2u * (uint32_t)(N > 0 ? N : 1)) cap <<= 1;
  return cap;
}

struct VoxWs {
  size_t keys, first, head, scan, sums, table, slots, total;
};
static VoxWs vox_layout(int N, int max_points, int max_voxels) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  VoxWs w;
  size_t o = 0;
  const size_t n = (size_t)(N > 0 ? N : 1);
  w.keys = o; o += up(n * 4);
  w.first = o; o += up(n * 4);
  w.head = o; o += up(n * 4);
  w.scan = o; o += up(n * 4);
  w.sums = o; o += up(((n + kScanBlock - 1) / kScanBlock) * 4);
  w.table = o; o += up((size_t)table_capacity(N) * 8);
  w.slots = o; o += up((size_t)max_voxels * max_points * 4);
  w.total = o;
  return w;
}

}  // namespace ubv

extern "C" int64_t ubv_hard_voxelize_workspace(int N, int max_points, int max_voxels) {
  if (N < 0 || max_points <= 0 || max_voxels <= 0) return -1;
  return (int64_t)ubv::vox_layout(N, max_points, max_voxels).total;
}

extern "C" int ubv_hard_voxelize(const float* points, float* voxels, int32_t* coors,
                                 int32_t* num_points, int32_t* voxel_num, void* workspace,
                                 int64_t workspace_bytes, int N, int F,
                                 const float* voxel_size_host, const float* range_host,
                                 int max_points, int max_voxels, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(voxels && coors && num_points && voxel_num && workspace && voxel_size_host &&
                    range_host, "hard_voxelize: null pointer");
  UBV_CHECK_ARG(N >= 0 && F >= 3 && max_points > 0 && max_voxels > 0,
                "hard_voxelize: bad dimension (N=%d F=%d T=%d M=%d)", N, F, max_points, max_voxels);
  UBV_CHECK_ARG(N == 0 || points != nullptr, "hard_voxelize: null points");
  VoxGeom g;
  UBV_CHECK_ARG(make_geom(voxel_size_host, range_host, g), "hard_voxelize: bad voxel grid");
  const VoxWs w = vox_layout(N, max_points, max_voxels);
  UBV_CHECK_ARG(workspace_bytes >= (int64_t)w.total, "hard_voxelize: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)w.total);
  hipStream_t st = as_stream(stream);
  char* ws = (char*)workspace;
  uint32_t* keys = (uint32_t*)(ws + w.keys);
  int* first = (int*)(ws + w.first);
  int* head = (int*)(ws + w.head);
  int* scan = (int*)(ws + w.scan);
  int* sums = (int*)(ws + w.sums);
  unsigned long long* table = (unsigned long long*)(ws + w.table);
  int* slots = (int*)(ws + w.slots);
  const uint32_t cap = table_capacity(N);
  if (hipMemsetAsync(table, 0xFF, (size_t)cap * 8, st) != hipSuccess ||
      hipMemsetAsync(slots, 0x7f, (size_t)max_voxels * max_points * 4, st) != hipSuccess) {
    set_error("hard_voxelize: memset failed");
    return UBV_ERR_LAUNCH;
  }
  const int nb = (N + 255) / 256;
  const int sb = (N + kScanBlock - 1) / kScanBlock;
  if (N > 0) {
    hipLaunchKernelGGL(vox_key_insert_kernel, dim3(nb), dim3(256), 0, st, points, N, F, g, keys,
                       table, cap - 1);
    hipLaunchKernelGGL(vox_head_kernel, dim3(nb), dim3(256), 0, st, keys, N, table, cap - 1, first,
                       head);
    hipLaunchKernelGGL(scan_block_kernel, dim3(sb), dim3(256), 0, st, head, scan, sums, N);
  }
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(64), 0, st, sums, N > 0 ? sb : 0, voxel_num,
                     max_voxels);
  if (N > 0)
    hipLaunchKernelGGL(vox_assign_kernel, dim3(nb), dim3(256), 0, st, keys, first, scan, sums, N, g,
                       max_points, max_voxels, slots, coors);
  const long nt = (long)max_voxels * max_points * F;
  hipLaunchKernelGGL(vox_gather_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, points,
                     F, slots, voxel_num, max_points, voxels, num_points, max_voxels);
  UBV_CHECK_LAUNCH("hard_voxelize");
  return UBV_OK;
}


extern "C" int64_t ubv_hard_voxelize_batch_workspace(int B, int n_max, int max_points, int max_voxels) {
  if (B <= 0 || B > ubv::kVoxMaxBatch || n_max < 0 || max_points <= 0 || max_voxels <= 0) return -1;
  return (int64_t)B * (int64_t)ubv::vox_layout(n_max, max_points, max_voxels).total;
}

extern "C" int ubv_hard_voxelize_batch(const float* const* points_host, const int* n_host, int B, float* voxels,
                                       int32_t* coors, int32_t* num_points, int32_t* voxel_num, void* workspace,
                                       int64_t workspace_bytes, int F, const float* voxel_size_host,
                                       const float* range_host, int max_points, int max_voxels, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(points_host && n_host && voxels && coors && num_points && voxel_num && workspace && voxel_size_host &&
                    range_host, "hard_voxelize_batch: null pointer");
  UBV_CHECK_ARG(B > 0 && B <= kVoxMaxBatch, "hard_voxelize_batch: 1 .. %d clouds per call (got %d)", kVoxMaxBatch, B);
  UBV_CHECK_ARG(F >= 3 && max_points > 0 && max_voxels > 0, "hard_voxelize_batch: bad dimension (F=%d T=%d M=%d)", F,
                max_points, max_voxels);
  VoxGeom g;
  UBV_CHECK_ARG(make_geom(voxel_size_host, range_host, g), "hard_voxelize_batch: bad voxel grid");
  int n_max = 0;
  VoxBatch vb{};
  for (int b = 0; b < B; ++b) {
    UBV_CHECK_ARG(n_host[b] >= 0 && (n_host[b] == 0 || points_host[b] != nullptr), "hard_voxelize_batch: cloud %d", b);
    vb.points[b] = points_host[b];
    vb.n[b] = n_host[b];
    n_max = n_host[b] > n_max ? n_host[b] : n_max;
  }
  const VoxWs w = vox_layout(n_max, max_points, max_voxels);
  UBV_CHECK_ARG(workspace_bytes >= (int64_t)B * (int64_t)w.total, "hard_voxelize_batch: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)B * (long long)w.total);
  vb.ws_stride = (long)w.total;
  vb.key_off = (long)w.keys; vb.first_off = (long)w.first; vb.head_off = (long)w.head; vb.scan_off = (long)w.scan;
  vb.sums_off = (long)w.sums; vb.table_off = (long)w.table; vb.slots_off = (long)w.slots;
  const uint32_t cap = table_capacity(n_max);
  vb.mask = cap - 1;
  hipStream_t st = as_stream(stream);
  char* ws = (char*)workspace;
  // tables (0xFF) and slot chains (0x7f7f7f7f) of every sample: their regions sit at fixed offsets of each stride
  for (int b = 0; b < B; ++b) {
    if (hipMemsetAsync(ws + (size_t)b * w.total + w.table, 0xFF, (size_t)cap * 8, st) != hipSuccess ||
        hipMemsetAsync(ws + (size_t)b * w.total + w.slots, 0x7f, (size_t)max_voxels * max_points * 4, st) != hipSuccess) {
      set_error("hard_voxelize_batch: memset failed");
      return UBV_ERR_LAUNCH;
    }
  }
  const int nb = (n_max + 255) / 256, sb = (n_max + kScanBlock - 1) / kScanBlock;
  if (n_max > 0) {
    hipLaunchKernelGGL(vox_key_insert_batch_kernel, dim3(nb, B), dim3(256), 0, st, vb, ws, F, g);
    hipLaunchKernelGGL(vox_head_batch_kernel, dim3(nb, B), dim3(256), 0, st, vb, ws);
    hipLaunchKernelGGL(scan_block_batch_kernel, dim3(sb, B), dim3(256), 0, st, vb, ws);
  }
  hipLaunchKernelGGL(scan_sums_batch_kernel, dim3(1, B), dim3(64), 0, st, vb, ws, voxel_num, max_voxels);
  if (n_max > 0)
    hipLaunchKernelGGL(vox_assign_batch_kernel, dim3(nb, B), dim3(256), 0, st, vb, ws, g, max_points, max_voxels, coors);
  const long nt = (long)max_voxels * max_points * F;
  hipLaunchKernelGGL(vox_gather_batch_kernel, dim3((unsigned)((nt + 255) / 256), B), dim3(256), 0, st, vb, ws, F, voxel_num,
                     max_points, voxels, num_points, max_voxels);
  UBV_CHECK_LAUNCH("hard_voxelize_batch");
  return UBV_OK;
}

extern "C" int ubv_dynamic_voxelize(const float* points, int32_t* coors, int N, int F,
                                    const float* voxel_size_host, const float* range_host,
                                    void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(coors && voxel_size_host && range_host, "dynamic_voxelize: null pointer");
  UBV_CHECK_ARG(N >= 0 && F >= 3, "dynamic_voxelize: bad dimension");
  VoxGeom g;
  UBV_CHECK_ARG(make_geom(voxel_size_host, range_host, g), "dynamic_voxelize: bad voxel grid");
  if (N == 0) return UBV_OK;
  UBV_CHECK_ARG(points != nullptr, "dynamic_voxelize: null points");
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream),
                     points, N, F, g, coors);
  UBV_CHECK_LAUNCH("dynamic_voxelize");
  return UBV_OK;
}

extern "C" int ubv_voxel_mean(const float* voxels, const int32_t* num_points,
                              const int32_t* voxel_num, float* mean, int max_voxels, int max_points,
                              int F, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(voxels && num_points && mean, "voxel_mean: null pointer");
  UBV_CHECK_ARG(max_voxels >= 0 && max_points > 0 && F > 0, "voxel_mean: bad dimension");
  if (max_voxels == 0) return UBV_OK;
  const long n = (long)max_voxels * F;
  hipLaunchKernelGGL(voxel_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     as_stream(stream), voxels, num_points, voxel_num, mean, max_voxels, max_points, F);
  UBV_CHECK_LAUNCH("voxel_mean");
  return UBV_OK;
}

extern "C" int ubv_sparse_to_dense(const float* feats, const int32_t* coors, const int32_t* m_dev,
                                   int m, float* dense, int B, int C, int D, int Hs, int Ws,
                                   void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(feats && coors && dense, "sparse_to_dense: null pointer");
  UBV_CHECK_ARG(m >= 0 && B > 0 && C > 0 && D > 0 && Hs > 0 && Ws > 0, "sparse_to_dense: bad dimension");
  if (m == 0) return UBV_OK;
  const long n = (long)m * C;
  hipLaunchKernelGGL(sparse_to_dense_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     as_stream(stream), feats, coors, m_dev, m, dense, B, C, D, Hs, Ws);
  UBV_CHECK_LAUNCH("sparse_to_dense");
  return UBV_OK;
}
