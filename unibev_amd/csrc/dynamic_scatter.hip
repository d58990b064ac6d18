// dynamic_point_to_voxel_forward ("DynamicScatter"): reduce the features of all points that share a
// voxel coordinate — the operator level's third voxel op ([ext] mmdet3d
// 0.18.1 `dynamic_point_to_voxel_forward(feats, coors, reduce_type)`; no shipped config reaches it,
// BASELINE.json's north_star names it).
//
// Contract restated from the published op:
//   * a point with any negative coordinate is dropped (its map entry is -1);
//   * voxels are the UNIQUE coordinate rows in ascending lexicographic order (torch.unique_dim,
//     sorted) — voxel_coors [M, D];
//   * point2voxel_map [N] gives each point's voxel, voxel_points_count [M] the points per voxel;
//   * voxel_feats [M, C] = sum | mean (= sum / count) | max over the voxel's points.
// The published CUDA kernel accumulates with atomics in arrival order; here every voxel is summed
// by one thread per channel in INPUT order (a stable sort keeps the points of a voxel in input
// order), so the result is deterministic and equals the sequential oracle bit for bit.
//
// Passes: 64-bit key per point (coordinates packed most-significant first, invalid = all ones) ->
// stable radix sort of (key, point index) [rocPRIM through hipCUB: a library primitive, not a hot
// kernel] -> head flags + exclusive scan = voxel id -> segment reduce.  Nothing is read back: the
// voxel count stays on the device, outputs are full-capacity (N rows).
#include <hipcub/hipcub.hpp>

#include "ubv_common.h"

namespace ubv {

constexpr uint64_t kBadKey = ~0ull;

__global__ __launch_bounds__(256) void ds_key_kernel(const int32_t* __restrict__ coors, int N, int D, int bits,
                                                     uint64_t* __restrict__ keys, int32_t* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  uint64_t k = 0;
  bool bad = false;
  for (int d = 0; d < D; ++d) {
    const int32_t c = coors[(long)i * D + d];
    bad |= c < 0 || (uint64_t)c >= (1ull << bits);
    k = (k << bits) | (uint64_t)(uint32_t)c;
  }
  keys[i] = bad ? kBadKey : k;
  idx[i] = i;
}

// head[i] = 1 where a new valid voxel starts in the sorted order
__global__ __launch_bounds__(256) void ds_head_kernel(const uint64_t* __restrict__ keys, int N,
                                                      int32_t* __restrict__ head) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const uint64_t k = keys[i];
  head[i] = (k != kBadKey && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// vid = inclusive scan of head - 1; writes the map, the segment starts, the voxel coordinates and M
__global__ __launch_bounds__(256) void ds_assign_kernel(const uint64_t* __restrict__ keys,
                                                        const int32_t* __restrict__ idx,
                                                        const int32_t* __restrict__ head,
                                                        const int32_t* __restrict__ scan, int N, int D,
                                                        const int32_t* __restrict__ coors,
                                                        int32_t* __restrict__ map, int32_t* __restrict__ seg_start,
                                                        int32_t* __restrict__ out_coors, int32_t* __restrict__ m_dev) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const bool valid = keys[i] != kBadKey;
  const int vid = scan[i] - 1;                         // scan is inclusive
  const int p = idx[i];
  map[p] = valid ? vid : -1;
  if (head[i]) {
    seg_start[vid] = i;
    for (int d = 0; d < D; ++d) out_coors[(long)vid * D + d] = coors[(long)p * D + d];
  }
  // the number of valid points = position of the first invalid key; M = scan at the last valid one
  if (i == N - 1) {
    m_dev[0] = scan[i];
    m_dev[1] = 0;
  }
}

// number of valid (sorted-first) points, needed to close the last segment
__global__ __launch_bounds__(256) void ds_nvalid_kernel(const uint64_t* __restrict__ keys, int N,
                                                        int32_t* __restrict__ m_dev) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const bool valid = keys[i] != kBadKey;
  if (valid && (i == N - 1 || keys[i + 1] == kBadKey)) m_dev[1] = i + 1;
}

// one thread per (voxel, channel): sequential reduction over the voxel's points in input order
__global__ __launch_bounds__(256) void ds_reduce_kernel(const float* __restrict__ feats, int C,
                                                        const int32_t* __restrict__ idx,
                                                        const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ m_dev, int reduce, int N,
                                                        float* __restrict__ out, int32_t* __restrict__ count) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int M = m_dev[0];
  const long v = t / C;
  const int c = (int)(t - v * C);
  if (v >= M) return;
  const int s = seg_start[v];
  const int e = (v + 1 < M) ? seg_start[v + 1] : m_dev[1];
  float acc = feats[(long)idx[s] * C + c];
  for (int i = s + 1; i < e; ++i) {
    const float x = feats[(long)idx[i] * C + c];
    acc = (reduce == 2) ? fmaxf(acc, x) : acc + x;
  }
  if (reduce == 1) acc = acc / (float)(e - s);
  out[v * C + c] = acc;
  if (c == 0) count[v] = e - s;
}

struct DsWs { size_t keys, keys2, idx, idx2, head, scan, seg, tmp, total; size_t tmp_bytes; };
static DsWs ds_ws(int N) {
  DsWs w;
  size_t sort_tmp = 0, scan_tmp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                           (const int32_t*)nullptr, (int32_t*)nullptr, N);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_tmp, (const int32_t*)nullptr, (int32_t*)nullptr, N);
  w.tmp_bytes = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t o = 0;
  w.keys = o; o += al((size_t)N * 8);
  w.keys2 = o; o += al((size_t)N * 8);
  w.idx = o; o += al((size_t)N * 4);
  w.idx2 = o; o += al((size_t)N * 4);
  w.head = o; o += al((size_t)N * 4);
  w.scan = o; o += al((size_t)N * 4);
  w.seg = o; o += al((size_t)N * 4);
  w.tmp = o; o += al(w.tmp_bytes);
  w.total = o;
  return w;
}

}  // namespace ubv

extern "C" int64_t ubv_dynamic_scatter_workspace(int N) {
  return N > 0 ? (int64_t)ubv::ds_ws(N).total : 0;
}

extern "C" int ubv_dynamic_point_to_voxel_forward(const float* feats, const int32_t* coors, int N, int C,
                                                  int D, int reduce_type, float* voxel_feats,
                                                  int32_t* voxel_coors, int32_t* point2voxel_map,
                                                  int32_t* voxel_points_count, int32_t* voxel_num,
                                                  void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(N >= 0 && C > 0 && D >= 1 && D <= 4, "dynamic_scatter: need N >= 0, C > 0, 1 <= D <= 4");
  UBV_CHECK_ARG(reduce_type >= 0 && reduce_type <= 2, "dynamic_scatter: reduce_type %d (0 sum, 1 mean, 2 max)",
                reduce_type);
  UBV_CHECK_ARG(voxel_num != nullptr, "dynamic_scatter: voxel_num is required");
  hipStream_t st = as_stream(stream);
  if (N == 0) {
    if (hipMemsetAsync(voxel_num, 0, 2 * sizeof(int32_t), st) != hipSuccess) {
      set_error("dynamic_scatter: memset failed");
      return UBV_ERR_LAUNCH;
    }
    return UBV_OK;
  }
  UBV_CHECK_ARG(feats && coors && voxel_feats && voxel_coors && point2voxel_map && voxel_points_count,
                "dynamic_scatter: null pointer");
  const DsWs w = ds_ws(N);
  UBV_CHECK_ARG(workspace != nullptr && workspace_bytes >= (int64_t)w.total,
                "dynamic_scatter: workspace of %lld bytes needed, got %lld", (long long)w.total,
                (long long)workspace_bytes);
  char* ws = (char*)workspace;
  uint64_t* keys = (uint64_t*)(ws + w.keys);
  uint64_t* keys2 = (uint64_t*)(ws + w.keys2);
  int32_t* idx = (int32_t*)(ws + w.idx);
  int32_t* idx2 = (int32_t*)(ws + w.idx2);
  int32_t* head = (int32_t*)(ws + w.head);
  int32_t* scan = (int32_t*)(ws + w.scan);
  int32_t* seg = (int32_t*)(ws + w.seg);
  const int bits = 63 / D > 21 ? 21 : 63 / D;            // 21 bits per coordinate for D = 3, 15 for D = 4
  const dim3 grid((N + 255) / 256), blk(256);
  hipLaunchKernelGGL(ds_key_kernel, grid, blk, 0, st, coors, N, D, bits, keys, idx);
  size_t tmp = w.tmp_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(ws + w.tmp, tmp, keys, keys2, idx, idx2, N, 0, 64, st) != hipSuccess) {
    set_error("dynamic_scatter: radix sort failed");
    return UBV_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(ds_head_kernel, grid, blk, 0, st, keys2, N, head);
  tmp = w.tmp_bytes;
  if (hipcub::DeviceScan::InclusiveSum(ws + w.tmp, tmp, head, scan, N, st) != hipSuccess) {
    set_error("dynamic_scatter: scan failed");
    return UBV_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(ds_assign_kernel, grid, blk, 0, st, keys2, idx2, head, scan, N, D, coors, point2voxel_map,
                     seg, voxel_coors, voxel_num);
  hipLaunchKernelGGL(ds_nvalid_kernel, grid, blk, 0, st, keys2, N, voxel_num);
  const long threads = (long)N * C;                       // capacity: at most N voxels
  hipLaunchKernelGGL(ds_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), blk, 0, st, feats, C, idx2, seg,
                     voxel_num, reduce_type, N, voxel_feats, voxel_points_count);
  UBV_CHECK_LAUNCH("dynamic_point_to_voxel_forward");
  return UBV_OK;
}
