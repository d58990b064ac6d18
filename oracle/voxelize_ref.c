/* ORACLE — test infrastructure only; never linked into the product library.
 *
 * Sequential CPU restatement of the LiDAR front end the reference reaches through mmdet3d 0.18.1
 * (un-vendored; pinned in /root/reference/docs/installation.md:6-9):
 *   - Voxelization / hard_voxelize, deterministic, called at
 *     projects/UniBEV/unibev_plugin/models/detectors/unibev_detector.py:163-167 with the config
 *     projects/UniBEV/configs/unibev/unibev_nus_LC_cnw_256_modality_dropout.py:186-190;
 *   - HardSimpleVFE (unibev_detector.py:117, config :191-193);
 *   - SparseConvTensor.dense() (tail of SparseEncoder, config :194-208);
 *   - dynamic_voxelize.
 * The algorithm is the published contract restated in SURVEY.md section 8(a) row a19-V / Appendix A:
 * iterate points in input order; c_j = floor((p_j - min_j) / size_j) in float32; reject outside
 * the grid; look up / create the voxel (skip the point if a new voxel would exceed max_voxels);
 * append the point if the voxel holds fewer than max_points.  coors are stored (z, y, x).
 * Parity status: unpinned by reference tests (the reference has none and the op's source is not
 * vendored); pinned instead by the hand-made known-answer clouds in tests/test_oracle_voxel.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int coord_of(const float* p, const float* vs, const float* rg, const int* grid, int* c) {
  for (int j = 0; j < 3; ++j) {
    volatile float d = p[j] - rg[j];          /* round the subtraction to f32 before dividing */
    volatile float q = d / vs[j];
    float f = floorf(q);
    if (!(f >= 0.0f && f < (float)grid[j])) return 0;   /* also rejects NaN */
    c[j] = (int)f;
  }
  return 1;
}

static void grid_of(const float* vs, const float* rg, int* grid) {
  for (int j = 0; j < 3; ++j) grid[j] = (int)roundf((rg[3 + j] - rg[j]) / vs[j]);
}

/* returns the number of voxels; voxels must be zero-initialised by the caller */
int oracle_hard_voxelize(const float* points, int N, int F, const float* voxel_size,
                         const float* coors_range, int max_points, int max_voxels, float* voxels,
                         int32_t* coors, int32_t* num_points_per_voxel) {
  int grid[3];
  grid_of(voxel_size, coors_range, grid);
  const size_t cells = (size_t)grid[0] * grid[1] * grid[2];
  int32_t* cell_to_voxel = (int32_t*)malloc(cells * sizeof(int32_t));
  memset(cell_to_voxel, 0xFF, cells * sizeof(int32_t));     /* -1 */
  int voxel_num = 0;
  for (int i = 0; i < N; ++i) {
    int c[3];
    if (!coord_of(points + (size_t)i * F, voxel_size, coors_range, grid, c)) continue;
    const size_t cell = ((size_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
    int v = cell_to_voxel[cell];
    if (v == -1) {
      if (voxel_num >= max_voxels) continue;
      v = voxel_num++;
      cell_to_voxel[cell] = v;
      coors[3 * v] = c[2]; coors[3 * v + 1] = c[1]; coors[3 * v + 2] = c[0];
    }
    const int n = num_points_per_voxel[v];
    if (n < max_points) {
      memcpy(voxels + ((size_t)v * max_points + n) * F, points + (size_t)i * F, F * sizeof(float));
      num_points_per_voxel[v] = n + 1;
    }
  }
  free(cell_to_voxel);
  return voxel_num;
}

void oracle_dynamic_voxelize(const float* points, int N, int F, const float* voxel_size,
                             const float* coors_range, int32_t* coors) {
  int grid[3];
  grid_of(voxel_size, coors_range, grid);
  for (int i = 0; i < N; ++i) {
    int c[3];
    if (coord_of(points + (size_t)i * F, voxel_size, coors_range, grid, c)) {
      coors[3 * i] = c[2]; coors[3 * i + 1] = c[1]; coors[3 * i + 2] = c[0];
    } else {
      coors[3 * i] = coors[3 * i + 1] = coors[3 * i + 2] = -1;
    }
  }
}

/* HardSimpleVFE: voxels[:, :, :F].sum(1) / num_points */
void oracle_voxel_mean(const float* voxels, const int32_t* num_points, int M, int T, int F,
                       float* mean) {
  for (int v = 0; v < M; ++v)
    for (int f = 0; f < F; ++f) {
      volatile float s = 0.0f;
      for (int k = 0; k < T; ++k) s = s + voxels[((size_t)v * T + k) * F + f];
      mean[(size_t)v * F + f] = s / (float)num_points[v];
    }
}

/* SparseConvTensor.dense(): dense must be zero-initialised */
void oracle_sparse_to_dense(const float* feats, const int32_t* coors, int M, int B, int C, int D,
                            int H, int W, float* dense) {
  for (int i = 0; i < M; ++i) {
    const int b = coors[4 * i], z = coors[4 * i + 1], y = coors[4 * i + 2], x = coors[4 * i + 3];
    for (int c = 0; c < C; ++c)
      dense[((((size_t)b * C + c) * D + z) * H + y) * W + x] = feats[(size_t)i * C + c];
  }
}
