/* ORACLE — test infrastructure only.
 *
 * Sequential restatement of [ext] mmdet3d 0.18.1 `dynamic_point_to_voxel_forward(feats, coors,
 * reduce_type)` (the "DynamicScatter" op named by BASELINE.json's north_star; SURVEY.md section 8(b)
 * operator level).  The op is not vendored under /root/reference and the reference holds no test
 * for it: PARITY UNPINNED — this file follows the published contract and is itself checked against
 * numpy's np.unique (the torch.unique_dim the published op is built on) in tests/test_oracle_voxel.py:
 *   - a point with any negative coordinate is dropped, its map entry is -1;
 *   - voxels = unique coordinate rows, ascending lexicographic order;
 *   - feats reduced per voxel by sum (0), mean = sum / count (1) or max (2);
 * sums run over a voxel's points in INPUT order (the published CUDA kernel adds atomically in
 * arrival order: any order is an instance of it; input order is the deterministic one).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t key[4]; int idx; } ds_item;
static int g_D;
static int ds_cmp(const void* a, const void* b) {
  const ds_item* x = (const ds_item*)a;
  const ds_item* y = (const ds_item*)b;
  for (int d = 0; d < g_D; ++d) {
    if (x->key[d] < y->key[d]) return -1;
    if (x->key[d] > y->key[d]) return 1;
  }
  return (x->idx > y->idx) - (x->idx < y->idx);          /* input order within a voxel */
}

int oracle_dynamic_scatter(const float* feats, const int32_t* coors, int N, int C, int D, int reduce,
                           float* out_feats, int32_t* out_coors, int32_t* map, int32_t* count) {
  ds_item* items = (ds_item*)malloc(sizeof(ds_item) * (size_t)(N > 0 ? N : 1));
  int nv = 0;
  for (int i = 0; i < N; ++i) {
    int bad = 0;
    for (int d = 0; d < D; ++d) bad |= coors[(size_t)i * D + d] < 0;
    map[i] = -1;
    if (bad) continue;
    for (int d = 0; d < D; ++d) items[nv].key[d] = coors[(size_t)i * D + d];
    items[nv].idx = i;
    ++nv;
  }
  g_D = D;
  qsort(items, (size_t)nv, sizeof(ds_item), ds_cmp);
  int M = 0;
  for (int i = 0; i < nv; ++i) {
    int newv = (i == 0);
    if (!newv)
      for (int d = 0; d < D; ++d) newv |= items[i].key[d] != items[i - 1].key[d];
    const float* f = feats + (size_t)items[i].idx * C;
    if (newv) {
      for (int d = 0; d < D; ++d) out_coors[(size_t)M * D + d] = (int32_t)items[i].key[d];
      for (int c = 0; c < C; ++c) out_feats[(size_t)M * C + c] = f[c];
      count[M] = 1;
      ++M;
    } else {
      float* o = out_feats + (size_t)(M - 1) * C;
      for (int c = 0; c < C; ++c) {
        if (reduce == 2) o[c] = (f[c] > o[c]) ? f[c] : o[c];     /* fmaxf on non-NaN data */
        else o[c] = o[c] + f[c];
      }
      count[M - 1] += 1;
    }
    map[items[i].idx] = M - 1;
  }
  if (reduce == 1)
    for (int v = 0; v < M; ++v)
      for (int c = 0; c < C; ++c) out_feats[(size_t)v * C + c] = out_feats[(size_t)v * C + c] / (float)count[v];
  free(items);
  return M;
}
