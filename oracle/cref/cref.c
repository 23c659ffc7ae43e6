/* oracle/cref/cref.c - C restatement of the reference's CPU ops, loop for loop (`cpu_ref`).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Nothing in the product loads this file's library; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, through oracle/cref.py.  PARITY PIN STATUS: unpinned by the
 * reference (it ships no vectors and cannot be compiled here); this file is a SECOND, independent statement of the same
 * algorithm - the reference's own loop nests and accumulation order in fp32, `#pragma omp parallel for` where the reference
 * calls `parallelize` - that tests/test_cref_cpu.py holds against the numpy oracle (oracle/ops.py, im2col + BLAS) and against
 * the committed fixtures in tests/golden.
 *
 * Layouts are the reference's: activations CHW (helpers/utils.mojo:805-811), token tensors (T, D), weights OIHW /
 * (out, in).  All arrays are dense float32.  Citations are file:line under /root/reference.
 *
 * Where the reference's literal body is undefined or wrong the restatement follows SURVEY.md Appendix A "build implements"
 * (softmax over the key axis with the maximum subtracted, per-token LayerNorm, standard head split), exactly as
 * oracle/ops.py does; each such place says so.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

API int cref_version(void) { return 1; }

API int cref_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* the bench picks the thread count from the host's CPU quota (oracle/cref.py threads_available) */
API void cref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---- elementwise ------------------------------------------------------------------------------------------------- */

/* SiLU.forward helpers/utils.mojo:1892-1902: x / (1 + exp(-x)) */
API void cref_silu(const float* x, float* y, long n) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) y[i] = x[i] / (1.0f + expf(-x[i]));
}

/* Gelu.forward helpers/utils.mojo:1908-1919: 0.5 (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) x (:1914) */
API void cref_gelu_tanh(const float* x, float* y, long n) {
  const float c = sqrtf((float)(2.0 / M_PI));
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) {
    const float v = x[i];
    const float cdf = 0.5f * (1.0f + tanhf(c * (v + 0.044715f * v * v * v)));
    y[i] = v * cdf;
  }
}

/* CLIP's activation, intended form x sigmoid(1.702 x) (clip.mojo:49-50; App.A D15) */
API void cref_quick_gelu(const float* x, float* y, long n) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) y[i] = x[i] / (1.0f + expf(-x[i] * 1.702f));
}

/* get_time_embedding helpers/utils.mojo:353-370 -> 320 values: f_i = 10000^(-i/160) (App.A D9), [cos(t f), sin(t f)] */
API void cref_time_embedding(float t, float* out) {
  for (int i = 0; i < 160; i++) {
    const float f = (float)pow(10000.0, -(double)i / 160.0);
    const float x = f * t;
    out[i] = cosf(x);
    out[160 + i] = sinf(x);
  }
}

/* ---- pad / conv / upsample ---------------------------------------------------------------------------------------- */

/* Matrix.pad helpers/utils.mojo:1383-1413: zeros (top, bottom), (left, right) around every channel */
API void cref_pad(const float* x, int C, int H, int W, int pt, int pb, int pl, int pr, float* y) {
  const int Hp = H + pt + pb, Wp = W + pl + pr;
  memset(y, 0, sizeof(float) * (size_t)C * Hp * Wp);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; c++)
    for (int h = 0; h < H; h++)
      memcpy(y + ((size_t)c * Hp + h + pt) * Wp + pl, x + ((size_t)c * H + h) * W, sizeof(float) * W);
}

/* Conv2D.forward helpers/utils.mojo:1738-1811.
 * pad (:1744-1747) ; Ho = floor((H - k)/sy) + 1 (:1752-1758) ; parallelize over output channels (:1809) ; per output pixel
 * (tile_2d walks y then x in steps of the stride, :405-409): for every input channel the k x k products are summed
 * (`multiply(...).sum()`, :1777) and that sum is added to the running fp32 total (:1779) ; + bias[o] (:1782).  Only the first
 * I = in_channels input channels are read (:1771; App.A D11) - the caller passes exactly those.
 * The x loop is innermost here so that one row of output pixels advances together; every pixel still sees the reference's
 * order of additions (taps inside a channel, channels in order). */
API void cref_conv2d(const float* x, int I, int H, int W, const float* w, const float* bias, int O, int k, int pt, int pb,
                     int pl, int pr, int sy, int sx, float* y) {
  const int Hp = H + pt + pb, Wp = W + pl + pr;
  const float* xp = x;
  float* tmp = NULL;
  if (pt || pb || pl || pr) {
    tmp = (float*)malloc(sizeof(float) * (size_t)I * Hp * Wp);
    cref_pad(x, I, H, W, pt, pb, pl, pr, tmp);
    xp = tmp;
  }
  const int Ho = (Hp - k) / sy + 1, Wo = (Wp - k) / sx + 1;
#pragma omp parallel
  {
    float* total = (float*)malloc(sizeof(float) * (size_t)Wo * 2);
    float* part = total + Wo;
#pragma omp for schedule(static) collapse(2)
    for (int o = 0; o < O; o++) {
      for (int yo = 0; yo < Ho; yo++) {
        const float* wk = w + (size_t)o * I * k * k;
        for (int xo = 0; xo < Wo; xo++) total[xo] = 0.0f;
        for (int ic = 0; ic < I; ic++) {
          for (int xo = 0; xo < Wo; xo++) part[xo] = 0.0f;
          for (int ky = 0; ky < k; ky++) {
            const float* row = xp + ((size_t)ic * Hp + (size_t)yo * sy + ky) * Wp;
            for (int kx = 0; kx < k; kx++) {
              const float wv = wk[((size_t)ic * k + ky) * k + kx];
              if (sx == 1) {
                for (int xo = 0; xo < Wo; xo++) part[xo] += row[xo + kx] * wv;
              } else {
                for (int xo = 0; xo < Wo; xo++) part[xo] += row[(size_t)xo * sx + kx] * wv;
              }
            }
          }
          for (int xo = 0; xo < Wo; xo++) total[xo] += part[xo];
        }
        const float bv = bias ? bias[o] : 0.0f;
        float* out = y + ((size_t)o * Ho + yo) * Wo;
        for (int xo = 0; xo < Wo; xo++) out[xo] = total[xo] + bv;
      }
    }
    free(total);
  }
  free(tmp);
}

/* Upsample.forward helpers/utils.mojo:1979-2010, build semantics (App.A D1): nearest neighbour x2 in H and W */
API void cref_upsample_nearest2x(const float* x, int C, int H, int W, float* y) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; c++)
    for (int h = 0; h < 2 * H; h++)
      for (int v = 0; v < 2 * W; v++) y[((size_t)c * 2 * H + h) * 2 * W + v] = x[((size_t)c * H + h / 2) * W + v / 2];
}

/* ---- norms --------------------------------------------------------------------------------------------------------- */

/* mean / std of one contiguous block, helpers/utils.mojo:1365-1380: mean = sum / n ; std = sqrt(sum((x - mean)^2) / n)
 * (population, despite the comment at :1370).  The reference adds fp32 SIMD partial sums in an unspecified order; the
 * restatement accumulates in double so that the checker does not depend on an order the reference does not define. */
static void mean_std(const float* x, long n, float* mean, float* std) {
  double s = 0.0;
  for (long i = 0; i < n; i++) s += (double)x[i];
  const float mu = (float)(s / (double)n);
  double q = 0.0;
  for (long i = 0; i < n; i++) {
    const double d = (double)x[i] - (double)mu;
    q += d * d;
  }
  *mean = mu;
  *std = (float)sqrt(q / (double)n);
}

/* GroupNorm.forward helpers/utils.mojo:1845-1885: per group (parallelize :1883) mean and std of its channels x H x W,
 * y = (x - mean) / (std + eps) * gamma with gamma = 1 (:1833) and eps added to the std (:1871-1873).  Only the first C
 * channels are normalised (:1847; App.A D11) - the caller passes exactly those. */
API void cref_group_norm(const float* x, int C, long HW, int G, float eps, float* y) {
  const long n = (long)(C / G) * HW;
#pragma omp parallel for schedule(static)
  for (int g = 0; g < G; g++) {
    float mean, std;
    mean_std(x + (size_t)g * n, n, &mean, &std);
    const float* xi = x + (size_t)g * n;
    float* yo = y + (size_t)g * n;
    for (long i = 0; i < n; i++) yo[i] = (xi[i] - mean) / (std + eps) * 1.0f;
  }
}

/* LayerNorm helpers/utils.mojo:2052-2061 on tokens (T, C), build semantics (App.A D8): the GroupNorm formula per token */
API void cref_layer_norm(const float* x, long T, int C, float eps, float* y) {
#pragma omp parallel for schedule(static)
  for (long t = 0; t < T; t++) {
    float mean, std;
    mean_std(x + (size_t)t * C, C, &mean, &std);
    for (int c = 0; c < C; c++) y[(size_t)t * C + c] = (x[(size_t)t * C + c] - mean) / (std + eps);
  }
}

/* ---- matmul / linear / softmax / attention ------------------------------------------------------------------------- */

/* Matrix.matmul helpers/utils.mojo:1549-1569: out zeroed (:1555), then per channel c and row m (both parallelize):
 * for k: for n: out[c, m, n] += a[c, m, k] * b[c, k, n]  - the k loop outside the n loop, fp32.
 * b_shared: the same (K, N) matrix for every channel (how Linear applies its weight). */
API void cref_matmul(const float* a, const float* b, float* out, int Bc, long M, long K, long N, int b_shared) {
#pragma omp parallel for schedule(static) collapse(2)
  for (int c = 0; c < Bc; c++)
    for (long m = 0; m < M; m++) {
      const float* ar = a + ((size_t)c * M + m) * K;
      const float* bm = b + (b_shared ? 0 : (size_t)c * K * N);
      float* o = out + ((size_t)c * M + m) * N;
      for (long n = 0; n < N; n++) o[n] = 0.0f;
      for (long k = 0; k < K; k++) {
        const float av = ar[k];
        const float* br = bm + (size_t)k * N;
        for (long n = 0; n < N; n++) o[n] += av * br[n];
      }
    }
}

static void transpose2d(const float* x, long R, long Cc, float* y) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < R; r++)
    for (long c = 0; c < Cc; c++) y[(size_t)c * R + r] = x[(size_t)r * Cc + c];
}

/* Linear.forward helpers/utils.mojo:1954-1976: x.matmul(weight.transpose(1, 2)) (:1959), then + bias per output column
 * (:1961-1974; broadcast over rows, App.A D2).  weight (out, in) (:1943). */
API void cref_linear(const float* x, const float* w, const float* bias, float* y, long M, long K, long N) {
  float* wt = (float*)malloc(sizeof(float) * (size_t)K * N);
  transpose2d(w, N, K, wt);
  cref_matmul(x, wt, y, 1, M, K, N, 1);
  free(wt);
  if (bias) {
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; m++)
      for (long n = 0; n < N; n++) y[(size_t)m * N + n] += bias[n];
  }
}

/* Softmax helpers/utils.mojo:411-448 as the attention calls it (dim = 2, helpers/attention.mojo:59,112): exp, then divide by
 * the sum.  Build semantics (App.A D6): over the key (last) axis, with the row maximum subtracted first (the literal code
 * normalises columns and subtracts nothing).  s is (rows, n), in place. */
API void cref_softmax_rows(float* s, long rows, long n) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; r++) {
    float* p = s + (size_t)r * n;
    float mx = p[0];
    for (long j = 1; j < n; j++) mx = p[j] > mx ? p[j] : mx;
    float sum = 0.0f;
    for (long j = 0; j < n; j++) {
      p[j] = expf(p[j] - mx);
      sum += p[j];
    }
    for (long j = 0; j < n; j++) p[j] = p[j] / sum;
  }
}

/* The attention core of Self_Attention.forward helpers/attention.mojo:30-62 and Cross_Attention.forward :105-115:
 * heads split off the feature axis (standard view + transpose, App.A D5), weight = q k^T (:46), optional mask j > i
 * (:48-55, App.A D7), / sqrt(d_head) (:57-58), Softmax(dim = 2) (:59), weight v (:60), heads merged back (:61-62).
 * q (Tq, D), k / v (Tk, D) -> o (Tq, D). */
API void cref_attention_core(const float* q, const float* k, const float* v, float* o, long Tq, long Tk, int D, int H,
                             int causal) {
  const int dh = D / H;
  float* qh = (float*)malloc(sizeof(float) * (size_t)H * Tq * dh);
  float* kt = (float*)malloc(sizeof(float) * (size_t)H * dh * Tk);
  float* vh = (float*)malloc(sizeof(float) * (size_t)H * Tk * dh);
  float* s = (float*)malloc(sizeof(float) * (size_t)H * Tq * Tk);
  float* oh = (float*)malloc(sizeof(float) * (size_t)H * Tq * dh);
#pragma omp parallel for schedule(static)
  for (int h = 0; h < H; h++) {
    for (long t = 0; t < Tq; t++)
      for (int d = 0; d < dh; d++) qh[((size_t)h * Tq + t) * dh + d] = q[(size_t)t * D + h * dh + d];
    for (long t = 0; t < Tk; t++)
      for (int d = 0; d < dh; d++) {
        kt[((size_t)h * dh + d) * Tk + t] = k[(size_t)t * D + h * dh + d];  /* k.transpose(1, 2) */
        vh[((size_t)h * Tk + t) * dh + d] = v[(size_t)t * D + h * dh + d];
      }
  }
  cref_matmul(qh, kt, s, H, Tq, dh, Tk, 0);
  const float scale = sqrtf((float)dh);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)H * Tq; i++) {
    const long row = i % Tq;
    float* p = s + (size_t)i * Tk;
    for (long j = 0; j < Tk; j++) {
      if (causal && j > row) p[j] = -INFINITY;
      p[j] = p[j] / scale;
    }
  }
  cref_softmax_rows(s, (long)H * Tq, Tk);
  cref_matmul(s, vh, oh, H, Tq, Tk, dh, 0);
#pragma omp parallel for schedule(static)
  for (long t = 0; t < Tq; t++)
    for (int h = 0; h < H; h++)
      for (int d = 0; d < dh; d++) o[(size_t)t * D + h * dh + d] = oh[((size_t)h * Tq + t) * dh + d];
  free(qh); free(kt); free(vh); free(s); free(oh);
}
