/* ORACLE — test infrastructure only; never linked into the product library.
 *
 * Plain-C loop restatement of the multi-scale deformable attention sampling op the reference
 * reaches through mmcv-full 1.3.17 `_ext.ms_deform_attn_forward/backward` (un-vendored; call
 * sites projects/UniBEV/unibev_plugin/models/modules/spatial_cross_attention_img.py:432-438,
 * spatial_cross_attention_pts.py:439-445, decoder.py:324-330).  Semantics:
 * pixel = loc * size - 0.5, bilinear, zero padding per corner; backward is the exact derivative.
 * Double precision accumulation so it can arbitrate between f32 implementations.
 * Parity: checked against the reference-recorded golden vectors (tests/golden/msda.npz, produced
 * with torch grid_sample == mmcv's published CPU definition) in tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

typedef struct { int idx[4]; double w[4]; double m[4]; double lx, ly; } fp_t;

static void footprint(double lx_, double ly_, int H, int W, fp_t* f) {
  const double x = lx_ * W - 0.5, y = ly_ * H - 0.5;
  const int inside = (y > -1.0) && (x > -1.0) && (y < H) && (x < W);
  const double xf = floor(x), yf = floor(y);
  const int x0 = inside ? (int)xf : 0, y0 = inside ? (int)yf : 0;
  f->lx = inside ? x - xf : 0.0;
  f->ly = inside ? y - yf : 0.0;
  const int xs[2] = {x0, x0 + 1}, ys[2] = {y0, y0 + 1};
  for (int k = 0; k < 4; ++k) {
    const int xx = xs[k & 1], yy = ys[k >> 1];
    const int ok = inside && xx >= 0 && xx <= W - 1 && yy >= 0 && yy <= H - 1;
    f->m[k] = ok ? 1.0 : 0.0;
    f->idx[k] = ok ? yy * W + xx : 0;
    const double wx = (k & 1) ? f->lx : 1.0 - f->lx, wy = (k >> 1) ? f->ly : 1.0 - f->ly;
    f->w[k] = ok ? wx * wy : 0.0;
  }
}

void oracle_msda_forward(const float* value, const int64_t* ss, const int64_t* ls,
                         const float* loc, const float* aw, float* out, int B, int S, int H, int Dh,
                         int L, int Nq, int P) {
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Nq; ++q)
      for (int h = 0; h < H; ++h)
        for (int c = 0; c < Dh; ++c) {
          double acc = 0.0;
          for (int l = 0; l < L; ++l)
            for (int p = 0; p < P; ++p) {
              const size_t pi = ((((size_t)b * Nq + q) * H + h) * L + l) * P + p;
              fp_t f;
              footprint(loc[2 * pi], loc[2 * pi + 1], (int)ss[2 * l], (int)ss[2 * l + 1], &f);
              double s = 0.0;
              for (int k = 0; k < 4; ++k)
                s += f.w[k] * value[(((size_t)b * S + ls[l] + f.idx[k]) * H + h) * Dh + c];
              acc += aw[pi] * s;
            }
          out[(((size_t)b * Nq + q) * H + h) * Dh + c] = (float)acc;
        }
}

/* gvalue (double, zero-initialised by the caller), gloc, gaw */
void oracle_msda_backward(const float* value, const int64_t* ss, const int64_t* ls,
                          const float* loc, const float* aw, const float* gout, double* gvalue,
                          float* gloc, float* gaw, int B, int S, int H, int Dh, int L, int Nq,
                          int P) {
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Nq; ++q)
      for (int h = 0; h < H; ++h)
        for (int l = 0; l < L; ++l)
          for (int p = 0; p < P; ++p) {
            const size_t pi = ((((size_t)b * Nq + q) * H + h) * L + l) * P + p;
            const int Hh = (int)ss[2 * l], Ww = (int)ss[2 * l + 1];
            fp_t f;
            footprint(loc[2 * pi], loc[2 * pi + 1], Hh, Ww, &f);
            double dot[4] = {0, 0, 0, 0};
            for (int c = 0; c < Dh; ++c) {
              const double g = gout[(((size_t)b * Nq + q) * H + h) * Dh + c];
              for (int k = 0; k < 4; ++k) {
                const size_t o = (((size_t)b * S + ls[l] + f.idx[k]) * H + h) * Dh + c;
                dot[k] += g * value[o] * f.m[k];
                gvalue[o] += aw[pi] * f.w[k] * g;
              }
            }
            const double hx = 1.0 - f.lx, hy = 1.0 - f.ly;
            gaw[pi] = (float)(hy * hx * dot[0] + hy * f.lx * dot[1] + f.ly * hx * dot[2] +
                              f.ly * f.lx * dot[3]);
            gloc[2 * pi] = (float)(aw[pi] * ((dot[1] - dot[0]) * hy + (dot[3] - dot[2]) * f.ly) * Ww);
            gloc[2 * pi + 1] = (float)(aw[pi] * ((dot[2] - dot[0]) * hx + (dot[3] - dot[1]) * f.lx) * Hh);
          }
}
