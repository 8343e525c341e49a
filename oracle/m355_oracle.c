/*
 * m355_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the arithmetic the reference's hot path executes
 * (efirdc/Segmentation-Pipeline is pure Python: every op below is the stock
 * torch.nn op named in the comment, called at the cited reference file:line,
 * paths relative to the reference repo root).  Same signatures as
 * include/m355seg.h with an `m355o_` prefix; `workspace`/`stream` are ignored.
 * Sums accumulate in double and round once to fp32, so the oracle is at least
 * as accurate as either torch-CPU or the HIP kernels it checks.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library -- and only as the checker.  The product (segmentation-pipeline_amd/)
 * never imports, links or calls it.
 *
 * Pinning: tests/test_oracle_cpu.py checks every function here against torch-CPU
 * ops and against golden vectors generated from the real reference modules
 * (tools/gen_golden.py -> tests/golden/).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/m355seg.h"

#define IDX5(n, c, z, y, x, C, D, H, W) \
  (((((int64_t)(n) * (C) + (c)) * (D) + (z)) * (H) + (y)) * (int64_t)(W) + (x))

static int64_t dense_or(int64_t s, int64_t d) { return s ? s : d; }
static int conv_out(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }
static int convt_out(int in, const m355_conv3d_desc* d) {
  return (in - 1) * d->stride - 2 * d->pad + d->k + d->out_pad;
}

int m355o_version(void) { return M355_ABI_VERSION; }

/* round-to-nearest-even fp32 -> bf16 -> fp32 (what v_cvt_pk_bf16_f32 does to an operand) */
static float bf16r(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return f; /* NaN */
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}
/* round-to-nearest-even to IEEE binary16 and back (subnormals and overflow to inf included) */
static float f16r(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
  if (a >= 0x7F800000u) return f;                 /* inf / nan */
  if (a >= 0x477FF000u) {                         /* >= 65520 rounds to inf */
    const uint32_t inf = sign | 0x7F800000u;
    float r;
    memcpy(&r, &inf, 4);
    return r;
  }
  if (a < 0x38800000u) {                          /* below 2^-14: multiples of 2^-24 */
    const float q = nearbyintf(fabsf(f) * 16777216.0f) / 16777216.0f;
    return sign ? -q : q;
  }
  uint32_t r = a + 0x00000FFFu + ((a >> 13) & 1u); /* keep 10 mantissa bits */
  r &= 0xFFFFE000u;
  r |= sign;
  float out;
  memcpy(&out, &r, 4);
  return out;
}
/* test hook: the operand rounding of a compute mode (tests/test_oracle_cpu.py checks it against torch's casts) */
float m355o_round_operand(float v, int32_t compute) {
  return compute == M355_COMPUTE_BF16 ? bf16r(v) : compute == M355_COMPUTE_F16 ? f16r(v) : v;
}
#define OPND(d, v) ((d)->compute == M355_COMPUTE_BF16 ? (double)bf16r(v) : (d)->compute == M355_COMPUTE_F16 ? (double)f16r(v) : (double)(v))

/* ------------------------------------------------------------------ conv3d
 * nn.Conv3d: models/components.py:36,42,51; models/modular_unet.py:83,99;
 * F.conv3d in BlurConv3d: components.py:119.  `add` = residual sum, components.py:67-68. */
int m355o_conv3d_fwd(const m355_conv3d_desc* d, const float* x, const float* w, const float* bias,
                     const float* add, float* y, void* ws, size_t wsb, void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int N = d->N, Ci = d->Cin, Co = d->Cout, D = d->D, H = d->H, W = d->W, k = d->k,
            s = d->stride, p = d->pad;
  const int OD = conv_out(D, k, s, p), OH = conv_out(H, k, s, p), OW = conv_out(W, k, s, p);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)Ci * D * H * W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)Co * OD * OH * OW);
  const int k3 = k * k * k;
#pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < Co; ++o)
      for (int oz = 0; oz < OD; ++oz)
        for (int oy = 0; oy < OH; ++oy)
          for (int ox = 0; ox < OW; ++ox) {
            double acc = bias ? (double)bias[o] : 0.0;
            for (int c = 0; c < Ci; ++c) {
              const float* xc = x + n * xbs + (int64_t)c * D * H * W;
              const float* wc = w + ((int64_t)o * Ci + c) * k3;
              for (int dz = 0; dz < k; ++dz) {
                const int iz = oz * s + dz - p;
                if (iz < 0 || iz >= D) continue;
                for (int dy = 0; dy < k; ++dy) {
                  const int iy = oy * s + dy - p;
                  if (iy < 0 || iy >= H) continue;
                  for (int dx = 0; dx < k; ++dx) {
                    const int ix = ox * s + dx - p;
                    if (ix < 0 || ix >= W) continue;
                    acc += OPND(d, wc[(dz * k + dy) * k + dx]) * OPND(d, xc[((int64_t)iz * H + iy) * W + ix]);
                  }
                }
              }
            }
            const int64_t yi = n * ybs + (((int64_t)o * OD + oz) * OH + oy) * OW + ox;
            if (add) acc += (double)add[yi];
            y[yi] = (float)acc;
          }
  return 0;
}

/* autograd of the op above w.r.t. its input */
int m355o_conv3d_bwd_data(const m355_conv3d_desc* d, const float* dy, const float* w, float* dx,
                          void* ws, size_t wsb, void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int N = d->N, Ci = d->Cin, Co = d->Cout, D = d->D, H = d->H, W = d->W, k = d->k,
            s = d->stride, p = d->pad;
  const int OD = conv_out(D, k, s, p), OH = conv_out(H, k, s, p), OW = conv_out(W, k, s, p);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)Ci * D * H * W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)Co * OD * OH * OW);
  const int k3 = k * k * k;
#pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < Ci; ++c)
      for (int iz = 0; iz < D; ++iz)
        for (int iy = 0; iy < H; ++iy)
          for (int ix = 0; ix < W; ++ix) {
            double acc = 0.0;
            for (int o = 0; o < Co; ++o) {
              const float* dyo = dy + n * ybs + (int64_t)o * OD * OH * OW;
              const float* wc = w + ((int64_t)o * Ci + c) * k3;
              for (int dz = 0; dz < k; ++dz) {
                const int tz = iz + p - dz;
                if (tz < 0 || tz % s || tz / s >= OD) continue;
                for (int dyy = 0; dyy < k; ++dyy) {
                  const int ty = iy + p - dyy;
                  if (ty < 0 || ty % s || ty / s >= OH) continue;
                  for (int dxx = 0; dxx < k; ++dxx) {
                    const int tx = ix + p - dxx;
                    if (tx < 0 || tx % s || tx / s >= OW) continue;
                    acc += OPND(d, wc[(dz * k + dyy) * k + dxx]) *
                           OPND(d, dyo[((int64_t)(tz / s) * OH + ty / s) * OW + tx / s]);
                  }
                }
              }
            }
            dx[n * xbs + (((int64_t)c * D + iz) * H + iy) * W + ix] = (float)acc;
          }
  return 0;
}

/* autograd w.r.t. weight and bias */
int m355o_conv3d_bwd_weight(const m355_conv3d_desc* d, const float* x, const float* dy, float* dw,
                            float* dbias, void* ws, size_t wsb, void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int N = d->N, Ci = d->Cin, Co = d->Cout, D = d->D, H = d->H, W = d->W, k = d->k,
            s = d->stride, p = d->pad;
  const int OD = conv_out(D, k, s, p), OH = conv_out(H, k, s, p), OW = conv_out(W, k, s, p);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)Ci * D * H * W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)Co * OD * OH * OW);
  const int k3 = k * k * k;
#pragma omp parallel for collapse(2) schedule(static)
  for (int o = 0; o < Co; ++o)
    for (int c = 0; c < Ci; ++c)
      for (int tap = 0; tap < k3; ++tap) {
        const int dz = tap / (k * k), dyy = (tap / k) % k, dxx = tap % k;
        double acc = 0.0;
        for (int n = 0; n < N; ++n) {
          const float* xc = x + n * xbs + (int64_t)c * D * H * W;
          const float* dyo = dy + n * ybs + (int64_t)o * OD * OH * OW;
          for (int oz = 0; oz < OD; ++oz) {
            const int iz = oz * s + dz - p;
            if (iz < 0 || iz >= D) continue;
            for (int oy = 0; oy < OH; ++oy) {
              const int iy = oy * s + dyy - p;
              if (iy < 0 || iy >= H) continue;
              for (int ox = 0; ox < OW; ++ox) {
                const int ix = ox * s + dxx - p;
                if (ix < 0 || ix >= W) continue;
                acc += OPND(d, dyo[((int64_t)oz * OH + oy) * OW + ox]) * OPND(d, xc[((int64_t)iz * H + iy) * W + ix]);
              }
            }
          }
        }
        dw[((int64_t)o * Ci + c) * k3 + tap] = (float)acc;
      }
  if (dbias) {
    for (int o = 0; o < Co; ++o) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n) {
        const float* dyo = dy + n * ybs + (int64_t)o * OD * OH * OW;
        for (int64_t i = 0; i < (int64_t)OD * OH * OW; ++i) acc += dyo[i];
      }
      dbias[o] = (float)acc;
    }
  }
  return 0;
}

/* ---------------------------------------------------------- conv-transpose3d
 * nn.ConvTranspose3d via the upsample_class hook: models/modular_unet.py:20-21,72-81,96;
 * F.conv_transpose3d in BlurConvTranspose3d: models/components.py:152. */
int m355o_conv_transpose3d_fwd(const m355_conv3d_desc* d, const float* x, const float* w,
                               const float* bias, float* y, void* ws, size_t wsb, void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int N = d->N, Ci = d->Cin, Co = d->Cout, D = d->D, H = d->H, W = d->W, k = d->k,
            s = d->stride, p = d->pad;
  const int OD = convt_out(D, d), OH = convt_out(H, d), OW = convt_out(W, d);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)Ci * D * H * W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)Co * OD * OH * OW);
  const int k3 = k * k * k;
#pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < Co; ++o)
      for (int oz = 0; oz < OD; ++oz)
        for (int oy = 0; oy < OH; ++oy)
          for (int ox = 0; ox < OW; ++ox) {
            double acc = bias ? (double)bias[o] : 0.0;
            for (int c = 0; c < Ci; ++c) {
              const float* xc = x + n * xbs + (int64_t)c * D * H * W;
              const float* wc = w + ((int64_t)c * Co + o) * k3;
              for (int dz = 0; dz < k; ++dz) {
                const int tz = oz + p - dz;
                if (tz < 0 || tz % s || tz / s >= D) continue;
                for (int dyy = 0; dyy < k; ++dyy) {
                  const int ty = oy + p - dyy;
                  if (ty < 0 || ty % s || ty / s >= H) continue;
                  for (int dxx = 0; dxx < k; ++dxx) {
                    const int tx = ox + p - dxx;
                    if (tx < 0 || tx % s || tx / s >= W) continue;
                    acc += (double)xc[((int64_t)(tz / s) * H + ty / s) * W + tx / s] *
                           (double)wc[(dz * k + dyy) * k + dxx];
                  }
                }
              }
            }
            y[n * ybs + (((int64_t)o * OD + oz) * OH + oy) * OW + ox] = (float)acc;
          }
  return 0;
}

int m355o_conv_transpose3d_bwd_data(const m355_conv3d_desc* d, const float* dy, const float* w,
                                    float* dx, void* ws, size_t wsb, void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int N = d->N, Ci = d->Cin, Co = d->Cout, D = d->D, H = d->H, W = d->W, k = d->k,
            s = d->stride, p = d->pad;
  const int OD = convt_out(D, d), OH = convt_out(H, d), OW = convt_out(W, d);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)Ci * D * H * W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)Co * OD * OH * OW);
  const int k3 = k * k * k;
#pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < Ci; ++c)
      for (int iz = 0; iz < D; ++iz)
        for (int iy = 0; iy < H; ++iy)
          for (int ix = 0; ix < W; ++ix) {
            double acc = 0.0;
            for (int o = 0; o < Co; ++o) {
              const float* dyo = dy + n * ybs + (int64_t)o * OD * OH * OW;
              const float* wc = w + ((int64_t)c * Co + o) * k3;
              for (int dz = 0; dz < k; ++dz) {
                const int oz = iz * s + dz - p;
                if (oz < 0 || oz >= OD) continue;
                for (int dyy = 0; dyy < k; ++dyy) {
                  const int oy = iy * s + dyy - p;
                  if (oy < 0 || oy >= OH) continue;
                  for (int dxx = 0; dxx < k; ++dxx) {
                    const int ox = ix * s + dxx - p;
                    if (ox < 0 || ox >= OW) continue;
                    acc += (double)dyo[((int64_t)oz * OH + oy) * OW + ox] *
                           (double)wc[(dz * k + dyy) * k + dxx];
                  }
                }
              }
            }
            dx[n * xbs + (((int64_t)c * D + iz) * H + iy) * W + ix] = (float)acc;
          }
  return 0;
}

int m355o_conv_transpose3d_bwd_weight(const m355_conv3d_desc* d, const float* x, const float* dy,
                                      float* dw, float* dbias, void* ws, size_t wsb,
                                      void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int N = d->N, Ci = d->Cin, Co = d->Cout, D = d->D, H = d->H, W = d->W, k = d->k,
            s = d->stride, p = d->pad;
  const int OD = convt_out(D, d), OH = convt_out(H, d), OW = convt_out(W, d);
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)Ci * D * H * W);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)Co * OD * OH * OW);
  const int k3 = k * k * k;
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < Ci; ++c)
    for (int o = 0; o < Co; ++o)
      for (int tap = 0; tap < k3; ++tap) {
        const int dz = tap / (k * k), dyy = (tap / k) % k, dxx = tap % k;
        double acc = 0.0;
        for (int n = 0; n < N; ++n) {
          const float* xc = x + n * xbs + (int64_t)c * D * H * W;
          const float* dyo = dy + n * ybs + (int64_t)o * OD * OH * OW;
          for (int iz = 0; iz < D; ++iz) {
            const int oz = iz * s + dz - p;
            if (oz < 0 || oz >= OD) continue;
            for (int iy = 0; iy < H; ++iy) {
              const int oy = iy * s + dyy - p;
              if (oy < 0 || oy >= OH) continue;
              for (int ix = 0; ix < W; ++ix) {
                const int ox = ix * s + dxx - p;
                if (ox < 0 || ox >= OW) continue;
                acc += (double)xc[((int64_t)iz * H + iy) * W + ix] *
                       (double)dyo[((int64_t)oz * OH + oy) * OW + ox];
              }
            }
          }
        }
        dw[((int64_t)c * Co + o) * k3 + tap] = (float)acc;
      }
  if (dbias) {
    for (int o = 0; o < Co; ++o) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n) {
        const float* dyo = dy + n * ybs + (int64_t)o * OD * OH * OW;
        for (int64_t i = 0; i < (int64_t)OD * OH * OW; ++i) acc += dyo[i];
      }
      dbias[o] = (float)acc;
    }
  }
  return 0;
}

/* --------------------------------------------------- normalisation (+ act)
 * normalization_class / activation_class of Block3d: models/components.py:24-26,52-55
 * (nn.BatchNorm3d default; nn.GroupNorm via functools.partial; nn.ReLU default). */
int64_t m355o_norm_num_stats(const m355_norm_desc* d) {
  return d->groups == 0 ? d->C : (int64_t)d->N * d->groups;
}

static float act_f(float v, int act, float slope) {
  if (act == M355_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == M355_ACT_LEAKY_RELU) return v > 0.f ? v : v * slope;
  return v;
}
static float act_g(float pre, int act, float slope) {
  if (act == M355_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
  if (act == M355_ACT_LEAKY_RELU) return pre > 0.f ? 1.f : slope;
  return 1.f;
}
static int64_t stat_of(const m355_norm_desc* d, int n, int c) {
  return d->groups == 0 ? c : (int64_t)n * d->groups + c / (d->C / d->groups);
}

int m355o_norm_stats(const m355_norm_desc* d, const float* x, float* mean, float* rstd,
                     float* running_mean, float* running_var, float momentum, void* ws,
                     size_t wsb, void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ns = m355o_norm_num_stats(d);
  double* s1 = (double*)calloc((size_t)ns, sizeof(double));
  double* s2 = (double*)calloc((size_t)ns, sizeof(double));
  for (int n = 0; n < d->N; ++n)
    for (int c = 0; c < d->C; ++c) {
      const float* p = x + n * xbs + (int64_t)c * d->S;
      double a = 0.0, b = 0.0;
      for (int64_t i = 0; i < d->S; ++i) { a += p[i]; b += (double)p[i] * p[i]; }
      const int64_t s = stat_of(d, n, c);
      s1[s] += a; s2[s] += b;
    }
  const double count = d->groups == 0 ? (double)d->N * d->S : (double)(d->C / d->groups) * d->S;
  for (int64_t s = 0; s < ns; ++s) {
    const double m = s1[s] / count;
    double var = s2[s] / count - m * m;
    if (var < 0) var = 0;
    mean[s] = (float)m;
    rstd[s] = (float)(1.0 / sqrt(var + (double)d->eps));
    if (running_mean) running_mean[s] = (1.f - momentum) * running_mean[s] + momentum * (float)m;
    if (running_var) {
      const double unb = count > 1 ? var * count / (count - 1) : var;
      running_var[s] = (1.f - momentum) * running_var[s] + momentum * (float)unb;
    }
  }
  free(s1); free(s2);
  return 0;
}

/* Fused conv + statistics (Block3d conv -> norm, reference models/components.py:51-53): the oracle
 * keeps ONE partial slot per (sample, channel) -- the exact (sum, sum of squares) of y in double,
 * rounded to float -- which is all the contract promises about the partials. */
int64_t m355o_conv3d_stats_slots(const m355_conv3d_desc* d) {
  return (d && d->k == 3 && d->stride == 1 && d->pad == 1) ? 1 : 0;
}
int m355o_conv3d_fwd_stats(const m355_conv3d_desc* d, const float* x, const float* w, const float* bias,
                           const float* add, float* y, float* part, void* ws, size_t wsb, void* stream) {
  const int rc = m355o_conv3d_fwd(d, x, w, bias, add, y, ws, wsb, stream);
  if (rc) return rc;
  const int64_t S = (int64_t)d->D * d->H * d->W;
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->Cout * S);
  for (int n = 0; n < d->N; ++n)
    for (int o = 0; o < d->Cout; ++o) {
      const float* p = y + n * ybs + (int64_t)o * S;
      double a = 0.0, b = 0.0;
      for (int64_t i = 0; i < S; ++i) { a += p[i]; b += (double)p[i] * p[i]; }
      part[((int64_t)n * d->Cout + o) * 2 + 0] = (float)a;
      part[((int64_t)n * d->Cout + o) * 2 + 1] = (float)b;
    }
  return 0;
}
int m355o_norm_stats_from_partials(const m355_norm_desc* d, const float* part, int64_t slots, float* mean,
                                   float* rstd, float* running_mean, float* running_var, float momentum,
                                   void* ws, size_t wsb, void* stream) {
  (void)stream; (void)ws; (void)wsb;
  const int64_t ns = m355o_norm_num_stats(d);
  double* s1 = (double*)calloc((size_t)ns, sizeof(double));
  double* s2 = (double*)calloc((size_t)ns, sizeof(double));
  for (int n = 0; n < d->N; ++n)
    for (int64_t q = 0; q < slots; ++q)
      for (int c = 0; c < d->C; ++c) {
        const float* v = part + (((int64_t)n * slots + q) * d->C + c) * 2;
        const int64_t s = stat_of(d, n, c);
        s1[s] += v[0]; s2[s] += v[1];
      }
  const double count = d->groups == 0 ? (double)d->N * d->S : (double)(d->C / d->groups) * d->S;
  for (int64_t s = 0; s < ns; ++s) {
    const double m = s1[s] / count;
    double var = s2[s] / count - m * m;
    if (var < 0) var = 0;
    mean[s] = (float)m;
    rstd[s] = (float)(1.0 / sqrt(var + (double)d->eps));
    if (running_mean) running_mean[s] = (1.f - momentum) * running_mean[s] + momentum * (float)m;
    if (running_var) {
      const double unb = count > 1 ? var * count / (count - 1) : var;
      running_var[s] = (1.f - momentum) * running_var[s] + momentum * (float)unb;
    }
  }
  free(s1); free(s2);
  return 0;
}

int m355o_norm_stats_from_running(const m355_norm_desc* d, const float* rm, const float* rv,
                                  float* mean, float* rstd, void* stream) {
  (void)stream;
  for (int c = 0; c < d->C; ++c) { mean[c] = rm[c]; rstd[c] = 1.f / sqrtf(rv[c] + d->eps); }
  return 0;
}

int m355o_norm_act_fwd(const m355_norm_desc* d, const float* x, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, const float* add,
                       float* y, void* stream) {
  (void)stream;
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->C * d->S);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < d->N; ++n)
    for (int c = 0; c < d->C; ++c) {
      const int64_t s = stat_of(d, n, c);
      const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
      const float* xp = x + n * xbs + (int64_t)c * d->S;
      float* yp = y + n * ybs + (int64_t)c * d->S;
      const int64_t abs_ = dense_or(d->add_batch_stride, (int64_t)d->C * d->S);
      const float* ap = add ? add + n * abs_ + (int64_t)c * d->S : NULL;
      for (int64_t i = 0; i < d->S; ++i) {
        const float pre = (xp[i] - mean[s]) * rstd[s] * g + b;
        yp[i] = act_f(pre, d->act, d->act_slope) + (ap ? ap[i] : 0.f);
      }
    }
  return 0;
}

int m355o_norm_act_bwd(const m355_norm_desc* d, const float* x, const float* dy, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, float* dx,
                       float* dgamma, float* dbeta, int training, void* ws, size_t wsb,
                       void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const int64_t xbs = dense_or(d->x_batch_stride, (int64_t)d->C * d->S);
  const int64_t ybs = dense_or(d->y_batch_stride, (int64_t)d->C * d->S);
  const int64_t ns = m355o_norm_num_stats(d);
  double* m1 = (double*)calloc((size_t)ns, sizeof(double));
  double* m2 = (double*)calloc((size_t)ns, sizeof(double));
  double* dg = (double*)calloc((size_t)d->C, sizeof(double));
  double* db = (double*)calloc((size_t)d->C, sizeof(double));
  for (int n = 0; n < d->N; ++n)
    for (int c = 0; c < d->C; ++c) {
      const int64_t s = stat_of(d, n, c);
      const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
      const float* xp = x + n * xbs + (int64_t)c * d->S;
      const float* dp = dy + n * ybs + (int64_t)c * d->S;
      double a = 0.0, bb = 0.0;
      for (int64_t i = 0; i < d->S; ++i) {
        const float xh = (xp[i] - mean[s]) * rstd[s];
        const float gg = dp[i] * act_g(xh * g + b, d->act, d->act_slope);
        a += gg; bb += (double)gg * xh;
      }
      db[c] += a; dg[c] += bb;
      m1[s] += (double)g * a; m2[s] += (double)g * bb;
    }
  const double count = d->groups == 0 ? (double)d->N * d->S : (double)(d->C / d->groups) * d->S;
  for (int n = 0; n < d->N; ++n)
    for (int c = 0; c < d->C; ++c) {
      const int64_t s = stat_of(d, n, c);
      const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
      const float* xp = x + n * xbs + (int64_t)c * d->S;
      const float* dp = dy + n * ybs + (int64_t)c * d->S;
      float* op = dx + n * xbs + (int64_t)c * d->S;
      const double a1 = training ? m1[s] / count : 0.0, a2 = training ? m2[s] / count : 0.0;
      for (int64_t i = 0; i < d->S; ++i) {
        const float xh = (xp[i] - mean[s]) * rstd[s];
        const double gg = (double)dp[i] * act_g(xh * g + b, d->act, d->act_slope) * g;
        op[i] = (float)(rstd[s] * (gg - a1 - xh * a2));
      }
    }
  for (int c = 0; c < d->C; ++c) {
    if (dgamma) dgamma[c] = (float)dg[c];
    if (dbeta) dbeta[c] = (float)db[c];
  }
  free(m1); free(m2); free(dg); free(db);
  return 0;
}

/* --------------------------------------------------------------- pooling
 * nn.AvgPool3d(2, 2, count_include_pad=False): models/modular_unet.py:22,41,64,92 */
int m355o_avgpool3d_2x_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H,
                           int32_t W, int64_t xbs_, int64_t ybs_, void* stream) {
  (void)stream;
  const int OD = D / 2, OH = H / 2, OW = W / 2;
  const int64_t xbs = dense_or(xbs_, (int64_t)C * D * H * W), ybs = dense_or(ybs_, (int64_t)C * OD * OH * OW);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int z = 0; z < OD; ++z)
        for (int yy = 0; yy < OH; ++yy)
          for (int xx = 0; xx < OW; ++xx) {
            double acc = 0.0;
            for (int a = 0; a < 2; ++a)
              for (int b = 0; b < 2; ++b)
                for (int e = 0; e < 2; ++e)
                  acc += x[n * xbs + (((int64_t)c * D + 2 * z + a) * H + 2 * yy + b) * W + 2 * xx + e];
            y[n * ybs + (((int64_t)c * OD + z) * OH + yy) * OW + xx] = (float)(acc / 8.0);
          }
  return 0;
}

int m355o_avgpool3d_2x_bwd(const float* dy, float* dx, int32_t N, int32_t C, int32_t D, int32_t H,
                           int32_t W, int64_t dybs_, int64_t dxbs_, void* stream) {
  (void)stream;
  const int OD = D / 2, OH = H / 2, OW = W / 2;
  const int64_t dxbs = dense_or(dxbs_, (int64_t)C * D * H * W), dybs = dense_or(dybs_, (int64_t)C * OD * OH * OW);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int z = 0; z < D; ++z)
        for (int yy = 0; yy < H; ++yy)
          for (int xx = 0; xx < W; ++xx)
            dx[n * dxbs + (((int64_t)c * D + z) * H + yy) * W + xx] =
                dy[n * dybs + (((int64_t)c * OD + z / 2) * OH + yy / 2) * OW + xx / 2] * 0.125f;
  return 0;
}

/* ------------------------------------------------------------- upsampling
 * nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True):
 * models/modular_unet.py:20,39,80,96.  Index/weight rule of ATen UpSample.h. */
static void lin(int o, int in, int out, int* i0, int* i1, float* l0, float* l1) {
  if (in == out) { *i0 = *i1 = o; *l0 = 1.f; *l1 = 0.f; return; }
  const float ratio = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = ratio * (float)o;
  int a = (int)floorf(src);
  if (a > in - 1) a = in - 1;
  float l = src - (float)a;
  if (l < 0.f) l = 0.f;
  if (l > 1.f) l = 1.f;
  *i0 = a; *i1 = a + (a < in - 1 ? 1 : 0); *l1 = l; *l0 = 1.f - l;
}

int m355o_upsample_trilinear2x_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t D,
                                   int32_t H, int32_t W, int64_t xbs_, int64_t ybs_, void* stream) {
  (void)stream;
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const int64_t xbs = dense_or(xbs_, (int64_t)C * D * H * W), ybs = dense_or(ybs_, (int64_t)C * OD * OH * OW);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      const float* xp = x + n * xbs + (int64_t)c * D * H * W;
      for (int oz = 0; oz < OD; ++oz)
        for (int oy = 0; oy < OH; ++oy)
          for (int ox = 0; ox < OW; ++ox) {
            int z0, z1, y0, y1, x0, x1; float lz0, lz1, ly0, ly1, lx0, lx1;
            lin(oz, D, OD, &z0, &z1, &lz0, &lz1);
            lin(oy, H, OH, &y0, &y1, &ly0, &ly1);
            lin(ox, W, OW, &x0, &x1, &lx0, &lx1);
#define V(z, yy, xx) (double)xp[((int64_t)(z) * H + (yy)) * W + (xx)]
            const double v = lz0 * (ly0 * (lx0 * V(z0, y0, x0) + lx1 * V(z0, y0, x1)) +
                                    ly1 * (lx0 * V(z0, y1, x0) + lx1 * V(z0, y1, x1))) +
                             lz1 * (ly0 * (lx0 * V(z1, y0, x0) + lx1 * V(z1, y0, x1)) +
                                    ly1 * (lx0 * V(z1, y1, x0) + lx1 * V(z1, y1, x1)));
#undef V
            y[n * ybs + (((int64_t)c * OD + oz) * OH + oy) * OW + ox] = (float)v;
          }
    }
  return 0;
}

/* scatter form (the textbook transpose of the forward), double accumulators */
int m355o_upsample_trilinear2x_bwd(const float* dy, float* dx, int32_t N, int32_t C, int32_t D,
                                   int32_t H, int32_t W, int64_t dybs_, int64_t dxbs_,
                                   void* stream) {
  (void)stream;
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const int64_t dxbs = dense_or(dxbs_, (int64_t)C * D * H * W), dybs = dense_or(dybs_, (int64_t)C * OD * OH * OW);
  const int64_t S = (int64_t)D * H * W;
  double* acc = (double*)malloc((size_t)S * sizeof(double));
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      memset(acc, 0, (size_t)S * sizeof(double));
      const float* dp = dy + n * dybs + (int64_t)c * OD * OH * OW;
      for (int oz = 0; oz < OD; ++oz)
        for (int oy = 0; oy < OH; ++oy)
          for (int ox = 0; ox < OW; ++ox) {
            int z[2], yv[2], xv[2]; float lz[2], ly[2], lx[2];
            lin(oz, D, OD, &z[0], &z[1], &lz[0], &lz[1]);
            lin(oy, H, OH, &yv[0], &yv[1], &ly[0], &ly[1]);
            lin(ox, W, OW, &xv[0], &xv[1], &lx[0], &lx[1]);
            const double g = dp[((int64_t)oz * OH + oy) * OW + ox];
            for (int a = 0; a < 2; ++a)
              for (int b = 0; b < 2; ++b)
                for (int e = 0; e < 2; ++e)
                  acc[((int64_t)z[a] * H + yv[b]) * W + xv[e]] += g * lz[a] * ly[b] * lx[e];
          }
      float* op = dx + n * dxbs + (int64_t)c * S;
      for (int64_t i = 0; i < S; ++i) op[i] = (float)acc[i];
    }
  free(acc);
  return 0;
}

/* ---------------------------------------------------------------- softmax
 * nn.Softmax(dim=1): models/modular_unet.py:26,46,84,100; StochasticMatrix: components.py:170-185 */
int m355o_softmax_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t inner, int64_t S,
                      float diag_bias, void* stream) {
  (void)stream;
  const int64_t cs = (int64_t)inner * S;
  for (int n = 0; n < N; ++n)
    for (int in = 0; in < inner; ++in)
      for (int64_t s = 0; s < S; ++s) {
        const int64_t base = ((int64_t)n * C * inner + in) * S + s;
        double mx = -INFINITY;
        for (int c = 0; c < C; ++c) {
          double v = x[base + c * cs];
          if (inner > 1 && c == in) v = (float)(x[base + c * cs] + diag_bias);
          if (v > mx) mx = v;
        }
        double sum = 0.0;
        for (int c = 0; c < C; ++c) {
          double v = x[base + c * cs];
          if (inner > 1 && c == in) v = (float)(x[base + c * cs] + diag_bias);
          sum += exp(v - mx);
        }
        for (int c = 0; c < C; ++c) {
          double v = x[base + c * cs];
          if (inner > 1 && c == in) v = (float)(x[base + c * cs] + diag_bias);
          y[base + c * cs] = (float)(exp(v - mx) / sum);
        }
      }
  return 0;
}

int m355o_softmax_bwd(const float* y, const float* dy, float* dx, int32_t N, int32_t C,
                      int32_t inner, int64_t S, void* stream) {
  (void)stream;
  const int64_t cs = (int64_t)inner * S;
  for (int n = 0; n < N; ++n)
    for (int in = 0; in < inner; ++in)
      for (int64_t s = 0; s < S; ++s) {
        const int64_t base = ((int64_t)n * C * inner + in) * S + s;
        double dot = 0.0;
        for (int c = 0; c < C; ++c) dot += (double)y[base + c * cs] * dy[base + c * cs];
        for (int c = 0; c < C; ++c)
          dx[base + c * cs] = (float)((double)y[base + c * cs] * ((double)dy[base + c * cs] - dot));
      }
  return 0;
}

/* ------------------------------------------------------- hybrid dice loss
 * HybridLogisticDiceLoss.forward: criterions/hybrid_logistic_dice_loss.py:13-43.
 * eps = 1e-8 (:15); overlap/total (:17-21); dice_coeffs (:22); prediction_safe (:25, the
 * python scalars 1e-8 and 1+1e-8 are rounded to fp32 by torch, so 1+eps == 1.0f);
 * logistic = mean(t*log(p_safe)) (:27) * class weights (:28-31); losses (:33-37). */
size_t m355o_hybrid_loss_workspace(int32_t N, int32_t C, int64_t S) { (void)N; (void)C; (void)S; return 0; }

int m355o_hybrid_loss_fwd(const float* p, const float* t, int32_t N, int32_t C, int64_t S,
                          float dice_weight, const float* cw, int32_t square_dice, float* out3,
                          float* sums, void* ws, size_t wsb, void* stream) {
  (void)ws; (void)wsb; (void)stream;
  const float eps = 1e-8f, denom = (float)(1.0 + 1e-8);
  double dice_acc = 0.0, log_acc = 0.0;
  for (int nc = 0; nc < N * C; ++nc) {
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    const float* pp = p + (int64_t)nc * S; const float* tp = t + (int64_t)nc * S;
    for (int64_t i = 0; i < S; ++i) {
      v0 += (double)pp[i] * tp[i];
      v1 += square_dice ? (double)pp[i] * pp[i] : pp[i];
      v2 += square_dice ? (double)tp[i] * tp[i] : tp[i];
      const float ps = (pp[i] + eps) / denom;
      v3 += (double)tp[i] * log((double)ps);
    }
    sums[nc * 4 + 0] = (float)v0; sums[nc * 4 + 1] = (float)v1;
    sums[nc * 4 + 2] = (float)v2; sums[nc * 4 + 3] = (float)v3;
    const float dice = 2.f * (float)v0 / (((float)v1 + (float)v2) + 1e-8f);
    dice_acc += 1.0 - dice;
    float logistic = (float)(v3 / (double)S);
    if (cw) logistic *= cw[nc % C];
    log_acc += -logistic;
  }
  const float dl = (float)(dice_acc / (N * C)), ll = (float)(log_acc / (N * C));
  out3[0] = (1.f - dice_weight) * ll + dice_weight * dl;
  out3[1] = dl; out3[2] = ll;
  return 0;
}

int m355o_hybrid_loss_bwd(const float* p, const float* t, const float* sums, const float* dloss,
                          int32_t N, int32_t C, int64_t S, float dice_weight, const float* cw,
                          int32_t square_dice, float* dp, void* stream) {
  (void)stream;
  const double g = dloss[0];
  for (int nc = 0; nc < N * C; ++nc) {
    const double O = sums[nc * 4], T = (double)sums[nc * 4 + 1] + sums[nc * 4 + 2] + 1e-8;
    const double w = cw ? cw[nc % C] : 1.0;
    const double inv = 1.0 / (N * C);
    for (int64_t i = 0; i < S; ++i) {
      const double pv = p[(int64_t)nc * S + i], tv = t[(int64_t)nc * S + i];
      const double dlog = -(1.0 - dice_weight) * w * inv / (double)S * tv / (double)(float)(pv + 1e-8f);
      const double dTdp = square_dice ? 2.0 * pv : 1.0;
      const double ddice = -dice_weight * inv * (2.0 * tv / T - 2.0 * O * dTdp / (T * T));
      dp[(int64_t)nc * S + i] = (float)(g * (dlog + ddice));
    }
  }
  return 0;
}

/* ------------------------------------------------------------ elementwise */
int m355o_copy_channels(const float* src, float* dst, int32_t N, int32_t C, int64_t S, int64_t sbs_,
                        int64_t dbs_, void* stream) {
  (void)stream;
  const int64_t CS = (int64_t)C * S, sbs = dense_or(sbs_, CS), dbs = dense_or(dbs_, CS);
  for (int n = 0; n < N; ++n) memcpy(dst + n * dbs, src + n * sbs, (size_t)CS * sizeof(float));
  return 0;
}
int m355o_channel_scale(const float* x, const float* scale, float* y, int32_t N, int32_t C,
                        int64_t S, void* stream) {
  (void)stream;
  for (int64_t nc = 0; nc < (int64_t)N * C; ++nc)
    for (int64_t i = 0; i < S; ++i) y[nc * S + i] = x[nc * S + i] * scale[nc];
  return 0;
}
int m355o_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  (void)stream;
  for (int64_t i = 0; i < n; ++i) y[i] = a[i] + b[i];
  return 0;
}

/* space-to-depth / depth-to-space by 2: pure index permutations used to run the reference's strided
 * Blur convolutions (models/components.py:91-154) as stride-1 3x3x3 convolutions */
static int s2d_o(const float* x, float* y, int N, int C, int D, int H, int W, int64_t xbs_, int64_t ybs_, int to_depth) {
  const int64_t dense = (int64_t)C * D * H * W;
  const int64_t fbs = dense_or(to_depth ? xbs_ : ybs_, dense), pbs = dense_or(to_depth ? ybs_ : xbs_, dense);
  const int HD = D / 2, HH = H / 2, HW2 = W / 2;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int z = 0; z < D; ++z)
        for (int yy = 0; yy < H; ++yy)
          for (int xx = 0; xx < W; ++xx) {
            const int p = (z & 1) * 4 + (yy & 1) * 2 + (xx & 1);
            const int64_t f = n * fbs + (((int64_t)c * D + z) * H + yy) * W + xx;
            const int64_t q = n * pbs + ((((int64_t)c * 8 + p) * HD + z / 2) * HH + yy / 2) * HW2 + xx / 2;
            if (to_depth) y[q] = x[f]; else y[f] = x[q];
          }
  return 0;
}
int m355o_space_to_depth2(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                          int64_t xbs, int64_t ybs, void* stream) { (void)stream; return s2d_o(x, y, N, C, D, H, W, xbs, ybs, 1); }
int m355o_depth_to_space2(const float* x, float* y, int32_t N, int32_t C, int32_t D, int32_t H, int32_t W,
                          int64_t xbs, int64_t ybs, void* stream) { (void)stream; return s2d_o(x, y, N, C, D, H, W, xbs, ybs, 0); }

/* Blur-convolution weight transform (reference models/components.py:112-119, 145-152; see
 * include/m355seg.h): standardise (unbiased std, eps 1e-5) -> depthwise 2x2x2 box blur with padding 1 ->
 * gather into the sparse 3x3x3 filter of the space-to-depth formulation.  All in double. */
static int blur_d(int par, int tap, int transposed) { return transposed ? 3 - 2 * tap + par : 2 * tap - 1 + par; }
static int64_t blur_o(int a, int b, int p, int t, int A, int B, int transposed) {
  return transposed ? (((int64_t)b * 8 + p) * A + a) * 27 + t : (((int64_t)a * B + b) * 8 + p) * 27 + t;
}
int m355o_blur_weight_fwd(const float* w, const float* scale, float* wexp, float* mean_std, int32_t A, int32_t B,
                          int32_t standardize, int32_t transposed, void* stream) {
  (void)stream;
  const int n = B * 27;
  for (int a = 0; a < A; ++a) {
    const float* wa = w + (int64_t)a * n;
    double m = 0.0, inv = 1.0;
    if (standardize) {
      double s1 = 0.0, s2 = 0.0;
      for (int i = 0; i < n; ++i) { s1 += wa[i]; }
      m = s1 / n;
      for (int i = 0; i < n; ++i) { s2 += (wa[i] - m) * (wa[i] - m); }
      const double sd = n > 1 ? sqrt(s2 / (n - 1)) : 0.0;
      inv = 1.0 / (sd + 1e-5);
      mean_std[a * 2 + 0] = (float)m;
      mean_std[a * 2 + 1] = (float)sd;
    }
    for (int b = 0; b < B; ++b)
      for (int p = 0; p < 8; ++p)
        for (int t = 0; t < 27; ++t) {
          const int dz = blur_d(p >> 2, t / 9, transposed), dy = blur_d((p >> 1) & 1, (t / 3) % 3, transposed),
                    dx = blur_d(p & 1, t % 3, transposed);
          double v = 0.0;
          if (dz >= 0 && dz <= 3 && dy >= 0 && dy <= 3 && dx >= 0 && dx <= 3) {
            for (int q = 0; q < 8; ++q) {
              const int z = dz + (q >> 2) - 1, y = dy + ((q >> 1) & 1) - 1, x = dx + (q & 1) - 1;
              if (z >= 0 && z <= 2 && y >= 0 && y <= 2 && x >= 0 && x <= 2)
                v += ((double)wa[b * 27 + (z * 3 + y) * 3 + x] - m) * inv;
            }
            v *= scale[b];
          }
          wexp[blur_o(a, b, p, t, A, B, transposed)] = (float)v;
        }
  }
  return 0;
}
int m355o_blur_weight_bwd(const float* dwexp, const float* w, const float* scale, const float* mean_std, float* dw,
                          int32_t A, int32_t B, int32_t standardize, int32_t transposed, void* stream) {
  (void)stream;
  const int n = B * 27;
  double* g = (double*)malloc(sizeof(double) * (size_t)n);
  for (int a = 0; a < A; ++a) {
    const float* wa = w + (int64_t)a * n;
    for (int i = 0; i < n; ++i) g[i] = 0.0;
    /* adjoint of the gather + blur: every valid (p, t) sends its gradient to the 8 filter taps it summed */
    for (int b = 0; b < B; ++b)
      for (int p = 0; p < 8; ++p)
        for (int t = 0; t < 27; ++t) {
          const int dz = blur_d(p >> 2, t / 9, transposed), dy = blur_d((p >> 1) & 1, (t / 3) % 3, transposed),
                    dx = blur_d(p & 1, t % 3, transposed);
          if (dz < 0 || dz > 3 || dy < 0 || dy > 3 || dx < 0 || dx > 3) continue;
          const double go = (double)dwexp[blur_o(a, b, p, t, A, B, transposed)] * scale[b];
          for (int q = 0; q < 8; ++q) {
            const int z = dz + (q >> 2) - 1, y = dy + ((q >> 1) & 1) - 1, x = dx + (q & 1) - 1;
            if (z >= 0 && z <= 2 && y >= 0 && y <= 2 && x >= 0 && x <= 2) g[b * 27 + (z * 3 + y) * 3 + x] += go;
          }
        }
    if (!standardize) {
      for (int i = 0; i < n; ++i) dw[(int64_t)a * n + i] = (float)g[i];
      continue;
    }
    const double m = mean_std[a * 2 + 0], sd = mean_std[a * 2 + 1], se = sd + 1e-5;
    double sg = 0.0, sgw = 0.0;
    for (int i = 0; i < n; ++i) { sg += g[i]; sgw += g[i] * (wa[i] - m); }
    const double c1 = sg / n, c2 = (n > 1 && sd > 0.0) ? sgw / (se * se * (n - 1) * sd) : 0.0;
    for (int i = 0; i < n; ++i) dw[(int64_t)a * n + i] = (float)((g[i] - c1) / se - (wa[i] - m) * c2);
  }
  free(g);
  return 0;
}

/* WSConv3d weight standardisation, models/components.py:81-88 (torch.std is unbiased). */
int m355o_weight_standardize_fwd(const float* w, float* wn, float* mean_std, int32_t A, int32_t n, void* stream) {
  (void)stream;
  for (int a = 0; a < A; ++a) {
    const float* wa = w + (int64_t)a * n;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; ++i) s1 += wa[i];
    const double m = s1 / n;
    for (int i = 0; i < n; ++i) s2 += (wa[i] - m) * (wa[i] - m);
    const double sd = n > 1 ? sqrt(s2 / (n - 1)) : 0.0;
    mean_std[a * 2 + 0] = (float)m;
    mean_std[a * 2 + 1] = (float)sd;
    for (int i = 0; i < n; ++i) wn[(int64_t)a * n + i] = (float)((wa[i] - m) / (sd + 1e-5));
  }
  return 0;
}
int m355o_weight_standardize_bwd(const float* dwn, const float* w, const float* mean_std, float* dw, int32_t A,
                                 int32_t n, void* stream) {
  (void)stream;
  for (int a = 0; a < A; ++a) {
    const float* wa = w + (int64_t)a * n;
    const float* g = dwn + (int64_t)a * n;
    const double m = mean_std[a * 2 + 0], sd = mean_std[a * 2 + 1], se = sd + 1e-5;
    double sg = 0.0, sgw = 0.0;
    for (int i = 0; i < n; ++i) { sg += g[i]; sgw += g[i] * (wa[i] - m); }
    const double c1 = sg / n, c2 = (n > 1 && sd > 0.0) ? sgw / (se * se * (n - 1) * sd) : 0.0;
    for (int i = 0; i < n; ++i) dw[(int64_t)a * n + i] = (float)((g[i] - c1) / se - (wa[i] - m) * c2);
  }
  return 0;
}

/* ------------------------------------------------- sliding-window patches
 * PatchPredict: prediction.py:132-143.  The arithmetic lives in torchio 0.18.45
 * (GridSampler / GridAggregator overlap_mode='average'), absent from the reference tree:
 * restated from its documented behaviour -- output += patch, count += 1, output / count. */
int m355o_patch_gather(const float* vol, const int32_t* loc, float* patches, int32_t P, int32_t C,
                       int32_t V0, int32_t V1, int32_t V2, int32_t ps0, int32_t ps1, int32_t ps2,
                       void* stream) {
  (void)stream; (void)V0;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < ps0; ++i)
        for (int j = 0; j < ps1; ++j)
          for (int k = 0; k < ps2; ++k)
            patches[((((int64_t)p * C + c) * ps0 + i) * ps1 + j) * ps2 + k] =
                vol[(((int64_t)c * V0 + loc[p * 3] + i) * V1 + loc[p * 3 + 1] + j) * V2 + loc[p * 3 + 2] + k];
  return 0;
}
int m355o_patch_accumulate(const float* patches, const int32_t* loc, float* accum, float* count,
                           int32_t P, int32_t C, int32_t V0, int32_t V1, int32_t V2, int32_t ps0,
                           int32_t ps1, int32_t ps2, void* stream) {
  (void)stream;
  const int64_t V = (int64_t)V0 * V1 * V2;
  for (int p = 0; p < P; ++p)
    for (int i = 0; i < ps0; ++i)
      for (int j = 0; j < ps1; ++j)
        for (int k = 0; k < ps2; ++k) {
          const int64_t v = (((int64_t)loc[p * 3] + i) * V1 + loc[p * 3 + 1] + j) * V2 + loc[p * 3 + 2] + k;
          for (int c = 0; c < C; ++c)
            accum[c * V + v] += patches[((((int64_t)p * C + c) * ps0 + i) * ps1 + j) * ps2 + k];
          count[v] += 1.f;
        }
  return 0;
}
int m355o_patch_finalize(const float* accum, const float* count, float* out, int32_t C, int64_t V,
                         void* stream) {
  (void)stream;
  for (int c = 0; c < C; ++c)
    for (int64_t v = 0; v < V; ++v) out[c * V + v] = accum[c * V + v] / count[v];
  return 0;
}

/* ----------------------------------------------------- evaluation counts
 * argmax: transforms/custom_label_transforms.py:267; TP/FP/FN/TN:
 * evaluators/segmentation_evaluator.py:69-78. */
int m355o_argmax_confusion(const float* prob, const int32_t* target, int32_t* argmax_out,
                           int64_t* counts, int32_t N, int32_t C, int64_t S, void* stream) {
  (void)stream;
  memset(counts, 0, (size_t)N * C * 4 * sizeof(int64_t));
  for (int n = 0; n < N; ++n)
    for (int64_t s = 0; s < S; ++s) {
      int best = 0; float bv = prob[(int64_t)n * C * S + s];
      for (int c = 1; c < C; ++c) {
        const float v = prob[((int64_t)n * C + c) * S + s];
        if (v > bv) { bv = v; best = c; }
      }
      if (argmax_out) argmax_out[(int64_t)n * S + s] = best;
      const int tg = target[(int64_t)n * S + s];
      for (int c = 0; c < C; ++c) {
        const int pc = best == c, tc = tg == c;
        counts[((int64_t)n * C + c) * 4 + (pc && tc ? 0 : pc ? 1 : tc ? 2 : 3)] += 1;
      }
    }
  return 0;
}

/* ------------------------------------------------- test-time ensembles
 * models/ensemble.py:16-35 (apply_strategy: stack -> mean, or argmax -> mode -> one_hot) and :50-103 (members run
 * on x.permute(perm).flip(f) and are mapped back with .flip(f).permute(inverse)).  perm3[j] = canonical axis of
 * member axis j, flip bit j = member axis j reversed. */
static int64_t member_offset(const int32_t* size, const int32_t* perm, int flip, int i0, int i1, int i2) {
  const int idx[3] = {i0, i1, i2};
  int msize[3], a[3];
  for (int j = 0; j < 3; ++j) {
    msize[j] = size[perm[j]];
    a[j] = ((flip >> j) & 1) ? msize[j] - 1 - idx[perm[j]] : idx[perm[j]];
  }
  return ((int64_t)a[0] * msize[1] + a[1]) * msize[2] + a[2];
}
int m355o_flip_permute(const float* x, float* member, int32_t N, int32_t C, const int32_t* size3, const int32_t* perm3,
                       int32_t flip_mask, void* stream) {
  (void)stream;
  const int64_t S = (int64_t)size3[0] * size3[1] * size3[2];
  for (int64_t nc = 0; nc < (int64_t)N * C; ++nc)
    for (int i0 = 0; i0 < size3[0]; ++i0)
      for (int i1 = 0; i1 < size3[1]; ++i1)
        for (int i2 = 0; i2 < size3[2]; ++i2)
          member[nc * S + member_offset(size3, perm3, flip_mask, i0, i1, i2)] =
              x[nc * S + ((int64_t)i0 * size3[1] + i1) * size3[2] + i2];
  return 0;
}
int m355o_ensemble_accumulate(const float* pred, float* acc, int32_t* votes, int32_t N, int32_t C, const int32_t* size3,
                              const int32_t* perm3, int32_t flip_mask, int32_t mode, int32_t first, void* stream) {
  (void)stream;
  const int64_t S = (int64_t)size3[0] * size3[1] * size3[2];
  for (int n = 0; n < N; ++n)
    for (int i0 = 0; i0 < size3[0]; ++i0)
      for (int i1 = 0; i1 < size3[1]; ++i1)
        for (int i2 = 0; i2 < size3[2]; ++i2) {
          const int64_t v = ((int64_t)i0 * size3[1] + i1) * size3[2] + i2;
          const float* p = pred + (int64_t)n * C * S + member_offset(size3, perm3, flip_mask, i0, i1, i2);
          if (mode == 0) {
            for (int c = 0; c < C; ++c) {
              float* a = acc + ((int64_t)n * C + c) * S + v;
              *a = first ? p[(int64_t)c * S] : *a + p[(int64_t)c * S];
            }
          } else {
            int best = 0;
            for (int c = 1; c < C; ++c)
              if (p[(int64_t)c * S] > p[(int64_t)best * S]) best = c;
            for (int c = 0; c < C; ++c) {
              int32_t* q = votes + ((int64_t)n * C + c) * S + v;
              if (first) *q = c == best ? 1 : 0;
              else if (c == best) *q += 1;
            }
          }
        }
  return 0;
}
int m355o_ensemble_finalize(const float* acc, const int32_t* votes, float* mean_out, int64_t* onehot_out, int32_t N,
                            int32_t C, int64_t S, int32_t members, int32_t mode, void* stream) {
  (void)stream;
  if (mode == 0) {
    for (int64_t i = 0; i < (int64_t)N * C * S; ++i) mean_out[i] = acc[i] * (1.0f / (float)members);
    return 0;
  }
  for (int n = 0; n < N; ++n)
    for (int64_t v = 0; v < S; ++v) {
      int best = 0;
      for (int c = 1; c < C; ++c)
        if (votes[((int64_t)n * C + c) * S + v] > votes[((int64_t)n * C + best) * S + v]) best = c;
      for (int c = 0; c < C; ++c) onehot_out[((int64_t)n * C + c) * S + v] = c == best ? 1 : 0;
    }
  return 0;
}

/* gradient of a tensor that feeds both AvgPool3d(2,2) and a skip connection (models/modular_unet.py:90-92) */
int m355o_avgpool3d_2x_bwd_add(const float* dy, const float* add, float* dx, int32_t N, int32_t C, int32_t D, int32_t H,
                               int32_t W, int64_t dybs, int64_t abs_, int64_t dxbs, void* stream) {
  (void)stream;
  const int OD = D / 2, OH = H / 2, OW = W / 2;
  const int64_t S = (int64_t)D * H * W, OS = (int64_t)OD * OH * OW;
  if (!dybs) dybs = (int64_t)C * OS;
  if (!abs_) abs_ = (int64_t)C * S;
  if (!dxbs) dxbs = (int64_t)C * S;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int z = 0; z < D; ++z)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const int64_t sp = (int64_t)c * S + ((int64_t)z * H + y) * W + x;
            dx[n * dxbs + sp] = add[n * abs_ + sp] +
                                dy[n * dybs + (int64_t)c * OS + ((int64_t)(z / 2) * OH + y / 2) * OW + x / 2] * 0.125f;
          }
  return 0;
}

/* GridSampler(padding_mode) / GridAggregator crop (torchio 0.18.45, behind prediction.py:114,132): numpy.pad index
 * rules for 'constant' (0), 'edge' (1), 'reflect' (2), 'symmetric' (3), 'wrap' (4). */
static int pad_map_o(int p, int V, int mode) {
  if (p >= 0 && p < V) return p;
  if (mode == 1) return p < 0 ? 0 : V - 1;
  if (mode == 2) {
    if (V == 1) return 0;
    const int period = 2 * V - 2;
    int q = p % period;
    if (q < 0) q += period;
    return q < V ? q : period - q;
  }
  if (mode == 3) {
    const int period = 2 * V;
    int q = p % period;
    if (q < 0) q += period;
    return q < V ? q : period - 1 - q;
  }
  if (mode == 4) {
    int q = p % V;
    return q < 0 ? q + V : q;
  }
  return -1;
}
int m355o_patch_gather_padded(const float* vol, const int32_t* loc, float* patches, int32_t P, int32_t C, int32_t V0,
                              int32_t V1, int32_t V2, int32_t ps0, int32_t ps1, int32_t ps2, int32_t b0, int32_t b1,
                              int32_t b2, int32_t mode, float value, void* stream) {
  (void)stream;
  for (int p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < ps0; ++i)
        for (int j = 0; j < ps1; ++j)
          for (int k = 0; k < ps2; ++k) {
            const int s0 = pad_map_o(loc[p * 3] + i - b0, V0, mode), s1 = pad_map_o(loc[p * 3 + 1] + j - b1, V1, mode),
                      s2 = pad_map_o(loc[p * 3 + 2] + k - b2, V2, mode);
            patches[((((int64_t)p * C + c) * ps0 + i) * ps1 + j) * ps2 + k] =
                (s0 < 0 || s1 < 0 || s2 < 0) ? value : vol[(((int64_t)c * V0 + s0) * V1 + s1) * V2 + s2];
          }
  return 0;
}
int m355o_patch_finalize_crop(const float* accum, const float* count, float* out, int32_t C, int32_t P0, int32_t P1,
                              int32_t P2, int32_t b0, int32_t b1, int32_t b2, void* stream) {
  (void)stream;
  const int V0 = P0 - 2 * b0, V1 = P1 - 2 * b1, V2 = P2 - 2 * b2;
  const int64_t PV = (int64_t)P0 * P1 * P2;
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < V0; ++i)
      for (int j = 0; j < V1; ++j)
        for (int k = 0; k < V2; ++k) {
          const int64_t pv = ((int64_t)(i + b0) * P1 + (j + b1)) * P2 + (k + b2);
          out[(((int64_t)c * V0 + i) * V1 + j) * V2 + k] = accum[(int64_t)c * PV + pv] / count[pv];
        }
  return 0;
}


/* m355_patch_aggregate_grid: GridAggregator('average') over a whole tile grid (prediction.py:124-152), tiles summed per
 * voxel in tile order in fp32 (the order of repeated m355o_patch_accumulate calls), then divided by their count */
int m355o_patch_aggregate_grid(const float* tiles, const int32_t* starts, int32_t n0, int32_t n1, int32_t n2, float* out,
                               int32_t C, int32_t V0, int32_t V1, int32_t V2, int32_t ps0, int32_t ps1, int32_t ps2,
                               int32_t b0, int32_t b1, int32_t b2, void* stream) {
  (void)stream;
  const int64_t V = (int64_t)V0 * V1 * V2, PS = (int64_t)ps0 * ps1 * ps2;
  const int32_t *s0 = starts, *s1 = starts + n0, *s2 = starts + n0 + n1;
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < V0; ++i)
      for (int j = 0; j < V1; ++j)
        for (int k = 0; k < V2; ++k) {
          float sum = 0.f;
          int cnt = 0;
          for (int a = 0; a < n0; ++a) {
            const int di = i + b0 - s0[a];
            if (di < 0 || di >= ps0) continue;
            for (int b = 0; b < n1; ++b) {
              const int dj = j + b1 - s1[b];
              if (dj < 0 || dj >= ps1) continue;
              for (int q = 0; q < n2; ++q) {
                const int dk = k + b2 - s2[q];
                if (dk < 0 || dk >= ps2) continue;
                const int64_t p = ((int64_t)a * n1 + b) * n2 + q;
                const float t = tiles[(p * C + c) * PS + ((int64_t)di * ps1 + dj) * ps2 + dk];
                sum = cnt == 0 ? t : sum + t;
                ++cnt;
              }
            }
          }
          out[(int64_t)c * V + ((int64_t)i * V1 + j) * V2 + k] = sum / (float)cnt;
        }
  return M355_OK;
}
