// GridMask on the device (the image-side augmentation of the detector).
//
// Reference: models/utils/grid_mask.py:70-123 builds the (1.5 h x 1.5 w) stripe mask with numpy loops on the host,
// pushes it through PIL, crops it, uploads it and multiplies ([n*c, h, w] x [h, w]) — per training step.  The mask
// is a closed form of five integers (period d, stripe length l, the two phases, the crop offsets), so the
// multiplication evaluates it in registers: nothing is built, uploaded or read.
#include "ubv_common.h"

namespace ubv {

struct GridMaskArgs { int h, w, d, l, st_h, st_w, off_y, off_x, nh, nw, use_h, use_w, mode; };

__device__ __forceinline__ bool gm_stripe(int p, int st, int d, int l, int n) {
  const int r = p - st;                       // stripes start at d * i + st, i < n, and are l wide
  if (r < 0) return false;
  const int i = r / d;
  return i < n && r - i * d < l;
}

template <typename T>
__global__ __launch_bounds__(256) void grid_mask_kernel(const T* __restrict__ x, T* __restrict__ y, long planes,
                                                        const GridMaskArgs g) {
  const long total = planes * g.h * g.w;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int px = (int)(i % g.w), py = (int)((i / g.w) % g.h);
    const bool hit = (g.use_h && gm_stripe(py + g.off_y, g.st_h, g.d, g.l, g.nh)) ||
                     (g.use_w && gm_stripe(px + g.off_x, g.st_w, g.d, g.l, g.nw));
    const bool keep = g.mode == 1 ? hit : !hit;      // mode 1 inverts the mask: only the stripes survive
    y[i] = keep ? x[i] : elem<T>::from_float(0.0f);
  }
}

}  // namespace ubv

extern "C" int ubv_grid_mask(const void* x, void* y, int64_t planes, int h, int w, int d, int l, int st_h, int st_w,
                             int use_h, int use_w, int mode, int dtype, void* stream) {
  using namespace ubv;
  UBV_CHECK_ARG(planes >= 0 && h > 0 && w > 0 && (planes == 0 || (x && y)), "grid_mask: bad arguments");
  UBV_CHECK_ARG(d >= 2 && l >= 1 && l < d && st_h >= 0 && st_h < d && st_w >= 0 && st_w < d,
                "grid_mask: need d >= 2, 1 <= l < d, 0 <= st < d (got d=%d l=%d st_h=%d st_w=%d)", d, l, st_h, st_w);
  UBV_CHECK_ARG(dtype >= 0 && dtype <= 2, "grid_mask: unknown dtype %d", dtype);
  if (planes == 0) return UBV_OK;
  const int hh = (int)(1.5 * h), ww = (int)(1.5 * w);
  const GridMaskArgs g{h, w, d, l, st_h, st_w, (hh - h) / 2, (ww - w) / 2, hh / d, ww / d, use_h, use_w, mode};
  const long total = (long)planes * h * w;
  const long blocks = (total + 255) / 256;
  const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case UBV_F32:
      hipLaunchKernelGGL(grid_mask_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, (long)planes, g);
      break;
    case UBV_F16:
      hipLaunchKernelGGL(grid_mask_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, (long)planes, g);
      break;
    default:
      hipLaunchKernelGGL(grid_mask_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (long)planes,
                         g);
      break;
  }
  UBV_CHECK_LAUNCH("grid_mask");
  return UBV_OK;
}
