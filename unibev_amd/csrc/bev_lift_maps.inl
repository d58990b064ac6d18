// MAPS backward plan — grad_value for LARGE per-camera maps (25x45 tokens per camera in the reference's
// cat-128 config; anything the single-band CAMERA plans do not hold).  Included by bev_lift.hip inside
// namespace ubv, after the GRID owner-tile kernel whose TileAcc / tile_own it reuses.
//
// The row-band CAMERA kernel re-walks a camera's visible queries once per band and keeps the points
// that fall into it: 7 bands at 25x45, and 85 % of the projected pillars land in two of them
// (1049 us per launch at bs = 2).  Here the points are binned by owner tile exactly as in the GRID
// plan, with two differences the camera geometry forces:
//   * the buckets are an exact CSR (count pass, scan, fill pass — lift_bin_kernel MODE 1 / 2): the
//     load per 8x8 tile varies 100:1, no fixed capacity fits;
//   * a bucket is cut into WORK ITEMS of kItemRecs records, one wave each, so the horizon tiles do not
//     serialise 13 000 records behind one wave; an item of a multi-item bucket writes a partial tile
//     (slab), maps_reduce_kernel adds a bucket's slabs in item order — no float atomics.  (Records enter a
//     bucket in arrival order, as on the GRID plan: grad_value repeats to the order of its f32 sums.)
// Every bucket has at least one item, so every pixel of grad_value is stored exactly once.
constexpr int kItemRecs = 1024;

// One block: exclusive scans of the bucket counts and of the items per bucket (chunks of 1024 with a
// carry), the item -> bucket table, the totals.
__global__ __launch_bounds__(1024) void maps_scan_kernel(const LiftArgs a, int n) {
  __shared__ int wtot[2][16];
  __shared__ int carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 2) carry[tid] = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int c = i < n ? a.bin_cnt[i] : 0;
    const int it = i < n ? max(1, (c + kItemRecs - 1) / kItemRecs) : 0;
    int sc = c, si = it;                                 // inclusive scans inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int tc = __shfl_up(sc, d, 64), ti = __shfl_up(si, d, 64);
      if (lane >= d) { sc += tc; si += ti; }
    }
    if (lane == 63) { wtot[0][wv] = sc; wtot[1][wv] = si; }
    __syncthreads();
    int oc = carry[0], oi = carry[1];
    for (int w = 0; w < wv; ++w) { oc += wtot[0][w]; oi += wtot[1][w]; }
    const int start = oc + sc - c, first = oi + si - it;
    if (i < n) {
      a.bin_start[i] = start;
      a.item_first[i] = first;
      for (int k = 0; k < it; ++k)
        if (first + k < a.max_items) a.item_bucket[first + k] = i;
    }
    __syncthreads();
    if (tid == 1023) { carry[0] = oc + sc; carry[1] = oi + si; }
    __syncthreads();
  }
  if (tid == 0) {
    a.bin_start[n] = carry[0];
    a.item_first[n] = carry[1];
    *a.n_items = min(carry[1], a.max_items);
  }
}

// One wave per work item (4 per block): the records [start + k * kItemRecs, + kItemRecs) of its bucket.
template <typename T, int DH, int P, int RB>
__global__ __launch_bounds__(256, 3) void lift_bwd_value_items_kernel(const LiftArgs a, const TileArgs t) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_all[];
  using L = TileLds<T, DH, RB>;
  const int item = blockIdx.x * 4 + wave_in_block();
  if (item >= *a.n_items) return;
  const int lane = threadIdx.x & 63;
  const int tiles = t.tiles_x * t.tiles_y;
  const int bk = a.item_bucket[item];
  const int k = item - a.item_first[bk], nit = a.item_first[bk + 1] - a.item_first[bk];
  TileGeom g;
  {
    const int tile = bk % tiles, mh = bk / tiles;
    g.h = mh % a.H;
    const int map = mh / a.H;
    g.cam = map % a.Nc; g.b = map / a.Nc; g.ck = 0;
    g.x0 = (tile % t.tiles_x) * 8; g.y0 = (tile / t.tiles_x) * 8;
    g.tw = min(8, a.fw - g.x0); g.th = min(8, a.fh - g.y0);
    g.npx = 64;
  }
  uint16_t* __restrict__ lds = lds_all + wave_in_block() * L::kWords;
  TileAcc<T, DH, RB> ta;
  ta.init(lds, lane);
  const long row = (long)a.H * DH;
  const T* __restrict__ gout = (const T*)a.gout;
  const int s0 = a.bin_start[bk] + k * kItemRecs;
  const int n = min(a.bin_start[bk + 1] - s0, kItemRecs);
  const float4* __restrict__ bp = a.bins + s0;
  const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 nrec = (lane < n) ? bp[lane] : zero4;
  for (int e0 = 0; e0 < n; e0 += 64) {
    const float4 rec = nrec;
    const bool valid = e0 + lane < n;
    if (e0 + 64 < n) nrec = (e0 + 64 + lane < n) ? bp[e0 + 64 + lane] : zero4;
    const int q = valid ? __float_as_int(rec.w) : 0;
    int lp[4];
    float cwt[4];
    const Footprint f = footprint_px(rec.x, rec.y, a.fh, a.fw);
    const bool any = tile_own(f, rec.z, valid, g, 8, lp, cwt);
    ta.add(lp, cwt, any, gout + (long)g.b * a.Nq * row + g.h * DH, (unsigned)q * (unsigned)row, lane);
  }
  if (ta.fill > 0) ta.flush(lane);
  // ---- D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = lane & 31;
  if (col >= DH) return;
  if (nit > 1) {                                         // partial tile [64 pixels][DH]
    float* __restrict__ sl = a.slab + (long)item * 64 * DH;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        sl[px * DH + col] = ta.acc[rb][r];
      }
    return;
  }
  const long mbase = ((long)g.b * a.Nc + g.cam) * a.fh * a.fw * row + g.h * DH;
  float* __restrict__ gv = a.gvalue + mbase;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int lx = px & 7, ly = px >> 3;
      if (lx < g.tw && ly < g.th) {
        const long o = ((long)(g.y0 + ly) * a.fw + (g.x0 + lx)) * row + col;
        const float v = ta.acc[rb][r];
        if (sizeof(T) == 2 && a.gvalue_lp != nullptr) ((T*)a.gvalue_lp)[mbase + o] = elem<T>::from_float(v);
        else gv[o] = v;
      }
    }
}

// One wave per multi-item bucket: its slabs summed in item order -> grad_value.
template <typename T, int DH>
__global__ __launch_bounds__(256) void maps_reduce_kernel(const LiftArgs a, int tiles_x, int tiles, int n) {
  const int bk = blockIdx.x * 4 + wave_in_block();
  if (bk >= n) return;
  const int first = a.item_first[bk], nit = min(a.item_first[bk + 1], a.max_items) - first;
  if (nit <= 1) return;
  const int lane = threadIdx.x & 63;
  const int tile = bk % tiles, mh = bk / tiles;
  const int h = mh % a.H, map = mh / a.H;
  const int x0 = (tile % tiles_x) * 8, y0 = (tile / tiles_x) * 8;
  const long row = (long)a.H * DH;
  const long mbase = (long)map * a.fh * a.fw * row + h * DH;
  const float* __restrict__ sl = a.slab + (long)first * 64 * DH;
  for (int e = lane * 4; e < 64 * DH; e += 256) {
    float4 s = *reinterpret_cast<const float4*>(sl + e);
    for (int k = 1; k < nit; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(sl + (long)k * 64 * DH + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int px = e / DH, c = e % DH;
    const int lx = px & 7, ly = px >> 3;
    if (x0 + lx < a.fw && y0 + ly < a.fh) {
      const long o = ((long)(y0 + ly) * a.fw + (x0 + lx)) * row + c;
      if (sizeof(T) == 2 && a.gvalue_lp != nullptr) {
        const float v[4] = {s.x, s.y, s.z, s.w};
        vec_io<T, 4>::store((T*)a.gvalue_lp + mbase + o, v);
      } else {
        *reinterpret_cast<float4*>(a.gvalue + mbase + o) = s;
      }
    }
  }
}
