// Pillar reference points -> camera pixels + visibility, one pass, no host syncs.
// Reference: ImgEncoder.point_sampling, models/modules/encoder_unibev_detr_img.py:112-187, plus the
// per-camera visibility / count bookkeeping of spatial_cross_attention_img.py:141-153, 209-211.
//
// Arithmetic follows the reference's fp32 op order (each torch op rounds once): de-normalise with a
// multiply then an add (:127-132), 4x4 @ 4x1 product, mask z > 1e-5 (:157), divide by
// max(z, 1e-5) (:163-164), divide by the image size of sample 0 (:166-167, quirk q5), strict
// inequalities (:174-177).  Contraction into FMAs is disabled so the rounding points stay put.
#include "ubv_common.h"

namespace ubv {

__global__ __launch_bounds__(256) void point_sampling_kernel(
    const float* __restrict__ l2i, const float* __restrict__ xs, const float* __restrict__ ys,
    const float* __restrict__ zs, float sx, float ox, float sy, float oy, float sz, float oz,
    float img_h, float img_w, float* __restrict__ ref_cam, uint8_t* __restrict__ mask,
    uint8_t* __restrict__ vis0, float* __restrict__ count, int B, int Nc, int bev_h, int bev_w,
    int D) {
#pragma clang fp contract(off)
  const int Nq = bev_h * bev_w;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * Nq) return;
  const int b = (int)(t / Nq), q = (int)(t - (long)b * Nq);
  const int qy = q / bev_w, qx = q - qy * bev_w;
  const float X = xs[qx] * sx + ox;
  const float Y = ys[qy] * sy + oy;
  int seen = 0;
  for (int cam = 0; cam < Nc; ++cam) {
    const float* m = l2i + ((long)b * Nc + cam) * 16;
    bool any = false;
    for (int d = 0; d < D; ++d) {
      const float Z = zs[d] * sz + oz;
      const float cx = ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
      const float cy = ((m[4] * X + m[5] * Y) + m[6] * Z) + m[7];
      const float cz = ((m[8] * X + m[9] * Y) + m[10] * Z) + m[11];
      const float eps = 1e-5f;
      bool ok = cz > eps;
      const float den = fmaxf(cz, eps);
      const float u = (cx / den) / img_w;
      const float v = (cy / den) / img_h;
      ok = ok && (v > 0.0f) && (v < 1.0f) && (u < 1.0f) && (u > 0.0f);
      const long o = ((((long)cam * B + b) * Nq + q) * D + d);
      ref_cam[2 * o] = u;
      ref_cam[2 * o + 1] = v;
      mask[o] = ok ? 1 : 0;
      any = any || ok;
    }
    if (b == 0) vis0[(long)cam * Nq + q] = any ? 1 : 0;
    seen += any ? 1 : 0;
  }
  count[t] = (float)(seen < 1 ? 1 : seen);
}

}  // namespace ubv

extern "C" int ubv_point_sampling(const float* lidar2img, const float* xs, const float* ys,
                                  const float* zs, const float* pc_range_host, float img_h,
                                  float img_w, float* ref_cam, uint8_t* bev_mask, uint8_t* vis0,
                                  float* count, int B, int Nc, int bev_h, int bev_w, int D,
                                  void* stream) {
  UBV_CHECK_ARG(lidar2img && xs && ys && zs && pc_range_host && ref_cam && bev_mask && vis0 && count,
                "point_sampling: null pointer");
  UBV_CHECK_ARG(B > 0 && Nc > 0 && bev_h > 0 && bev_w > 0 && D > 0,
                "point_sampling: non-positive dimension");
  const float* r = pc_range_host;
  // (pc_range[3] - pc_range[0]) is evaluated in Python (double) and rounded to f32 when it meets
  // the f32 tensor; same here.
  const float sx = (float)((double)r[3] - (double)r[0]), sy = (float)((double)r[4] - (double)r[1]),
              sz = (float)((double)r[5] - (double)r[2]);
  const long n = (long)B * bev_h * bev_w;
  hipLaunchKernelGGL(ubv::point_sampling_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ubv::as_stream(stream), lidar2img, xs, ys, zs, sx, r[0], sy, r[1], sz, r[2],
                     img_h, img_w, ref_cam, bev_mask, vis0, count, B, Nc, bev_h, bev_w, D);
  UBV_CHECK_LAUNCH("point_sampling");
  return UBV_OK;
}
