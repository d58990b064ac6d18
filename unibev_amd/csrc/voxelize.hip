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

constexpr int kScanBlock = 1024;        // points per block of the head-flag scan

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
    int32_t* __restrict__ num_points, int32_t* __restrict__ coors, float* __restrict__ mean, int max_voxels) {
  // one thread per (voxel, slot): the point's F features (round 1: one thread per feature — 5x the threads, each
  // re-reading the slot)
  const long t = (long)bid * blockDim.x + threadIdx.x;
  if (t >= (long)max_voxels * max_points) return;
  const int v = (int)(t / max_points);
  const int slot = (int)(t - (long)v * max_points);
  const bool live = v < *voxel_num;
  const int idx = live ? slots[t] : kSlotEmpty;
  float* out = voxels + t * F;
  if (idx != kSlotEmpty) {
    const float* p = points + (long)idx * F;
    for (int f = 0; f < F; ++f) out[f] = p[f];
  } else {
    for (int f = 0; f < F; ++f) out[f] = 0.0f;
  }
  if (slot == 0) {
    int n = 0;
    if (live)
      for (int k = 0; k < max_points; ++k) n += slots[(long)v * max_points + k] != kSlotEmpty;
    else
      coors[3 * (long)v] = coors[3 * (long)v + 1] = coors[3 * (long)v + 2] = 0;      // (rows past the count: zeros)
    num_points[v] = n;
    if (mean != nullptr) {
      // HardSimpleVFE on the way: sum of the stored points in slot order (the valid slots come first, ascending) / n —
      // the additions ubv_voxel_mean makes, the zero padding left out
      for (int f = 0; f < F; ++f) {
        float sum = 0.0f;
        for (int k = 0; k < n; ++k) sum += points[(long)slots[(long)v * max_points + k] * F + f];
        mean[(long)v * F + f] = n > 0 ? sum / (float)n : 0.0f;
      }
    }
  }
}


// ---- kernels: one sample (blockIdx.x walks the sample) and a batch (blockIdx.y = sample; every per-sample array at
// a fixed stride, point counts and cloud pointers in VoxBatch) — a batch is ONE launch chain instead of one per cloud
constexpr int kVoxMaxBatch = 16;
struct VoxBatch {
  const float* points[kVoxMaxBatch];
  int n[kVoxMaxBatch];
  long ws_stride;                 // bytes of workspace per sample
  long key_off, first_off, scan_off, sums_off, table_off, slots_off;   // offsets inside a sample's workspace
  uint32_t mask;                  // table capacity - 1 (sized for the largest cloud)
};

#define UBV_VOX_WS(type, off) ((type*)(ws + (long)blockIdx.y * vb.ws_stride + vb.off))
// The chain is FIVE launches for any batch (round 3: 2 fills per cloud + 8 kernels; a 30 000-point cloud is ~20 us of
// kernel work, so the chain's time was its launch gaps):
//   fill (tables 0xFF.., slot chains 0x7f7f7f7f, every sample) | keys + insert | head flags + per-1024 scan (one
//   block owns its 1024 points: no pass between them) | block-sum scan in LDS by every block + slot chains +
//   coordinates + voxel count | gather.
// (One cooperative kernel with grid-wide barriers between the phases measured SLOWER — 113 us against 68 for the
//  chain at 120 k voxels: each barrier is an agent-scope release / acquire across the 8 XCDs' L2s, ~12 us.)
constexpr int kVoxMaxScanBlocks = 1024;                    // block sums scanned in LDS: clouds up to 1 M points

__global__ __launch_bounds__(256) void vox_fill_batch_kernel(VoxBatch vb, char* ws, long table_bytes, long slot_bytes) {
  uint4* t = UBV_VOX_WS(uint4, table_off);
  const uint4 ff = make_uint4(~0u, ~0u, ~0u, ~0u);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < table_bytes / 16; i += (long)gridDim.x * 256) t[i] = ff;
  uint4* sl = UBV_VOX_WS(uint4, slots_off);
  const uint4 e = make_uint4((uint32_t)kSlotEmpty, (uint32_t)kSlotEmpty, (uint32_t)kSlotEmpty, (uint32_t)kSlotEmpty);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < slot_bytes / 16; i += (long)gridDim.x * 256) sl[i] = e;
}
__global__ __launch_bounds__(256) void vox_key_insert_batch_kernel(VoxBatch vb, char* ws, int F, VoxGeom g) {
  vox_key_insert_body(blockIdx.x, vb.points[blockIdx.y], vb.n[blockIdx.y], F, g, UBV_VOX_WS(uint32_t, key_off),
                      UBV_VOX_WS(unsigned long long, table_off), vb.mask);
}
// blockIdx.x = one 1024-point scan block.  A thread looks up the first points of ITS four consecutive points (the four
// probes in flight together), keeps the head flags in registers and scans them: the flags never go through memory.
__global__ __launch_bounds__(256) void vox_head_scan_batch_kernel(VoxBatch vb, char* ws) {
  __shared__ int wave_tot[4];
  const int n = vb.n[blockIdx.y];
  const uint32_t* __restrict__ keys = UBV_VOX_WS(uint32_t, key_off);
  const unsigned long long* __restrict__ table = UBV_VOX_WS(unsigned long long, table_off);
  int* __restrict__ first = UBV_VOX_WS(int, first_off);
  int* __restrict__ scan = UBV_VOX_WS(int, scan_off);
  const int base = blockIdx.x * kScanBlock + threadIdx.x * 4;
  uint32_t key[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) key[k] = base + k < n ? keys[base + k] : kBadKey;
  int v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int f = -1;
    if (key[k] != kBadKey) {
      uint32_t h = hash_u32(key[k]) & vb.mask;
      for (;;) {
        const unsigned long long cur = table[h];
        if ((uint32_t)(cur >> 32) == key[k]) { f = (int)(uint32_t)cur; break; }
        h = (h + 1) & vb.mask;
      }
    }
    if (base + k < n) first[base + k] = f;
    v[k] = (f == base + k && f >= 0) ? 1 : 0;
    s += v[k];
  }
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
    if (base + k < n) scan[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) UBV_VOX_WS(int, sums_off)[blockIdx.x] = off + incl;
}
__global__ __launch_bounds__(256) void vox_assign_batch_kernel(VoxBatch vb, char* ws, VoxGeom g, int max_points,
                                                               int max_voxels, int32_t* __restrict__ coors,
                                                               int32_t* __restrict__ voxel_num) {
  // every block scans the block sums itself (at most kVoxMaxScanBlocks of them): no pass of its own
  __shared__ int s_pre[kVoxMaxScanBlocks];
  const int n = vb.n[blockIdx.y], sb = (n + kScanBlock - 1) / kScanBlock;
  const int* sums = UBV_VOX_WS(int, sums_off);
  for (int i = threadIdx.x; i < sb; i += 256) s_pre[i] = sums[i];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < sb; base += 64) {
      const int v = (base + lane < sb) ? s_pre[base + lane] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      if (base + lane < sb) s_pre[base + lane] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0 && blockIdx.x == 0) voxel_num[blockIdx.y] = carry < max_voxels ? carry : max_voxels;
  }
  __syncthreads();
  vox_assign_body(blockIdx.x, UBV_VOX_WS(uint32_t, key_off), UBV_VOX_WS(int, first_off), UBV_VOX_WS(int, scan_off), s_pre, n,
                  g, max_points, max_voxels, UBV_VOX_WS(int, slots_off), coors + (long)blockIdx.y * max_voxels * 3);
}
__global__ __launch_bounds__(256) void vox_gather_batch_kernel(VoxBatch vb, char* ws, int F, const int* __restrict__ voxel_num,
                                                               int max_points, float* __restrict__ voxels,
                                                               int32_t* __restrict__ num_points, int32_t* __restrict__ coors,
                                                               float* __restrict__ mean, int max_voxels) {
  vox_gather_body(blockIdx.x, vb.points[blockIdx.y], F, UBV_VOX_WS(int, slots_off), voxel_num + blockIdx.y, max_points,
                  voxels + (long)blockIdx.y * max_voxels * max_points * F, num_points + (long)blockIdx.y * max_voxels,
                  coors + (long)blockIdx.y * max_voxels * 3,
                  mean != nullptr ? mean + (long)blockIdx.y * max_voxels * F : nullptr, max_voxels);
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
  if (voxel_num != nullptr && v >= *voxel_num) { mean[t] = 0.0f; return; }
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
  size_t keys, first, scan, sums, table, slots, total;
};
static VoxWs vox_layout(int N, int max_points, int max_voxels) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  VoxWs w;
  size_t o = 0;
  const size_t n = (size_t)(N > 0 ? N : 1);
  w.keys = o; o += up(n * 4);
  w.first = o; o += up(n * 4);
  w.scan = o; o += up(n * 4);
  w.sums = o; o += up(((n + kScanBlock - 1) / kScanBlock) * 4);
  w.table = o; o += up((size_t)table_capacity(N) * 8);
  w.slots = o; o += up((size_t)max_voxels * max_points * 4);
  w.total = o;
  return w;
}


// the five launches over B samples (vb filled by the caller)
static int vox_run(VoxBatch& vb, int B, int n_max, char* ws, const VoxWs& w, int F, const VoxGeom& g, int max_points,
                   int max_voxels, float* voxels, int32_t* coors, int32_t* num_points, int32_t* voxel_num, float* mean,
                   hipStream_t st) {
  vb.ws_stride = (long)w.total;
  vb.key_off = (long)w.keys; vb.first_off = (long)w.first; vb.scan_off = (long)w.scan;
  vb.sums_off = (long)w.sums; vb.table_off = (long)w.table; vb.slots_off = (long)w.slots;
  vb.mask = table_capacity(n_max) - 1;
  const int nb = (n_max + 255) / 256, sb = (n_max + kScanBlock - 1) / kScanBlock;
  if (sb > kVoxMaxScanBlocks) return UBV_ERR_UNSUPPORTED;
  const long table_bytes = (long)(w.slots - w.table), slot_bytes = (long)(w.total - w.slots);
  const long fill16 = (table_bytes + slot_bytes) / 16;
  const unsigned fb = (unsigned)((fill16 / 4 + 255) / 256);                  // ~4 stores of 16 bytes per thread
  hipLaunchKernelGGL(vox_fill_batch_kernel, dim3(fb < 1 ? 1 : (fb > 1024 ? 1024 : fb), B), dim3(256), 0, st, vb, ws,
                     table_bytes, slot_bytes);
  if (n_max > 0) {
    hipLaunchKernelGGL(vox_key_insert_batch_kernel, dim3(nb, B), dim3(256), 0, st, vb, ws, F, g);
    hipLaunchKernelGGL(vox_head_scan_batch_kernel, dim3(sb, B), dim3(256), 0, st, vb, ws);
  }
  // (no points: one block per sample still writes voxel_num = 0)
  hipLaunchKernelGGL(vox_assign_batch_kernel, dim3(nb > 0 ? nb : 1, B), dim3(256), 0, st, vb, ws, g, max_points, max_voxels,
                     coors, voxel_num);
  const long nt = (long)max_voxels * max_points;
  hipLaunchKernelGGL(vox_gather_batch_kernel, dim3((unsigned)((nt + 255) / 256), B), dim3(256), 0, st, vb, ws, F, voxel_num,
                     max_points, voxels, num_points, coors, mean, max_voxels);
  return UBV_OK;
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
  VoxBatch vb{};
  vb.points[0] = points;
  vb.n[0] = N;
  if (vox_run(vb, 1, N, (char*)workspace, w, F, g, max_points, max_voxels, voxels, coors, num_points, voxel_num, nullptr,
              as_stream(stream)) != UBV_OK) {
    set_error("hard_voxelize: clouds of more than %d points are not supported", kVoxMaxScanBlocks * kScanBlock);
    return UBV_ERR_UNSUPPORTED;
  }
  UBV_CHECK_LAUNCH("hard_voxelize");
  return UBV_OK;
}

extern "C" int64_t ubv_hard_voxelize_batch_workspace(int B, int n_max, int max_points, int max_voxels) {
  if (B <= 0 || B > ubv::kVoxMaxBatch || n_max < 0 || max_points <= 0 || max_voxels <= 0) return -1;
  return (int64_t)B * (int64_t)ubv::vox_layout(n_max, max_points, max_voxels).total;
}

static int hard_voxelize_batch_impl(const float* const* points_host, const int* n_host, int B, float* voxels,
                                    int32_t* coors, int32_t* num_points, int32_t* voxel_num, float* mean, void* workspace,
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
  if (vox_run(vb, B, n_max, (char*)workspace, w, F, g, max_points, max_voxels, voxels, coors, num_points, voxel_num, mean,
              as_stream(stream)) != UBV_OK) {
    set_error("hard_voxelize_batch: clouds of more than %d points are not supported", kVoxMaxScanBlocks * kScanBlock);
    return UBV_ERR_UNSUPPORTED;
  }
  UBV_CHECK_LAUNCH("hard_voxelize_batch");
  return UBV_OK;
}

extern "C" int ubv_hard_voxelize_batch(const float* const* points_host, const int* n_host, int B, float* voxels,
                                       int32_t* coors, int32_t* num_points, int32_t* voxel_num, void* workspace,
                                       int64_t workspace_bytes, int F, const float* voxel_size_host,
                                       const float* range_host, int max_points, int max_voxels, void* stream) {
  return hard_voxelize_batch_impl(points_host, n_host, B, voxels, coors, num_points, voxel_num, nullptr, workspace,
                                  workspace_bytes, F, voxel_size_host, range_host, max_points, max_voxels, stream);
}

extern "C" int ubv_hard_voxelize_batch_vfe(const float* const* points_host, const int* n_host, int B, float* voxels,
                                           int32_t* coors, int32_t* num_points, int32_t* voxel_num, float* mean,
                                           void* workspace, int64_t workspace_bytes, int F, const float* voxel_size_host,
                                           const float* range_host, int max_points, int max_voxels, void* stream) {
  if (mean == nullptr) { ubv::set_error("hard_voxelize_batch_vfe: null mean"); return UBV_ERR_INVALID; }
  return hard_voxelize_batch_impl(points_host, n_host, B, voxels, coors, num_points, voxel_num, mean, workspace,
                                  workspace_bytes, F, voxel_size_host, range_host, max_points, max_voxels, stream);
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
