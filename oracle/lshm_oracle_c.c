/* Plain-C restatement of the per-op math of the LSHM hot path.
 *
 * TEST INFRASTRUCTURE ONLY (checker for tests/, never linked into the product).
 * Independent of PyTorch: straight loops, double accumulation, written from the
 * formulas in SURVEY.md Appendix A, each function citing the reference call site
 * (paths relative to the upstream repo).  tests/test_oracle_c.py checks that it
 * agrees with oracle/lshm_oracle.py (which is pinned to reference-generated
 * golden vectors), so the two restatements cross-validate each other.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -o oracle/_build/liblshm_oracle_c.so oracle/lshm_oracle_c.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double elu_d(double v) { return v > 0 ? v : expm1(v); }

/* F.elu(conv2d(x, w, b, stride=2, padding=1)), kernel 4x4        src/lofar_models.py:31-41,73-78 */
void oc_conv2d_k4s2p1(const float* x, const float* w, const float* b, float* y, int B, int Cin, int Cout,
                      int H, int W, int act) {
  const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < B; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox) {
          double acc = b ? b[co] : 0.0;
          for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 4; ++ky) {
              const int iy = 2 * oy - 1 + ky;
              if (iy < 0 || iy >= H) continue;
              for (int kx = 0; kx < 4; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                acc += (double)w[((co * Cin + ci) * 4 + ky) * 4 + kx] * x[((size_t)(n * Cin + ci) * H + iy) * W + ix];
              }
            }
          y[((size_t)(n * Cout + co) * Ho + oy) * Wo + ox] = (float)(act ? elu_d(acc) : acc);
        }
}

/* conv_transpose2d(x, w, b, stride=2, padding=1), w (Cin,Cout,4,4)  src/lofar_models.py:52-57,93-98 */
void oc_tconv2d_k4s2p1(const float* x, const float* w, const float* b, float* y, int B, int Cin, int Cout,
                       int H, int W, int act) {
  const int Ho = 2 * H, Wo = 2 * W;
  const size_t n_out = (size_t)B * Cout * Ho * Wo;
  double* acc = (double*)calloc(n_out, sizeof(double));
  /* scatter form: every input pixel adds a 4x4 stamp at (2iy-1+ky, 2ix-1+kx) */
  for (int n = 0; n < B; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int iy = 0; iy < H; ++iy)
        for (int ix = 0; ix < W; ++ix) {
          const double v = x[((size_t)(n * Cin + ci) * H + iy) * W + ix];
          for (int co = 0; co < Cout; ++co)
            for (int ky = 0; ky < 4; ++ky) {
              const int oy = 2 * iy - 1 + ky;
              if (oy < 0 || oy >= Ho) continue;
              for (int kx = 0; kx < 4; ++kx) {
                const int ox = 2 * ix - 1 + kx;
                if (ox < 0 || ox >= Wo) continue;
                acc[((size_t)(n * Cout + co) * Ho + oy) * Wo + ox] += v * w[((ci * Cout + co) * 4 + ky) * 4 + kx];
              }
            }
        }
  for (size_t i = 0; i < n_out; ++i) {
    const int co = (int)((i / ((size_t)Ho * Wo)) % Cout);
    const double v = acc[i] + (b ? b[co] : 0.0);
    y[i] = (float)(act ? elu_d(v) : v);
  }
  free(acc);
}

/* conv1d(x, w, b, stride=4, padding=1), kernel 4                   src/lofar_models.py:115-125,158-163 */
void oc_conv1d_k4s4p1(const float* x, const float* w, const float* b, float* y, int B, int Cin, int Cout,
                      int L, int act) {
  const int Lo = (L - 2) / 4 + 1;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < B; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int j = 0; j < Lo; ++j) {
        double acc = b ? b[co] : 0.0;
        for (int ci = 0; ci < Cin; ++ci)
          for (int k = 0; k < 4; ++k) {
            const int pos = 4 * j - 1 + k;
            if (pos < 0 || pos >= L) continue;
            acc += (double)w[(co * Cin + ci) * 4 + k] * x[(size_t)(n * Cin + ci) * L + pos];
          }
        y[(size_t)(n * Cout + co) * Lo + j] = (float)(act ? elu_d(acc) : acc);
      }
}

/* conv_transpose1d(x, w, b, stride=4, padding=0), w (Cin,Cout,4)   src/lofar_models.py:137-142,178-183 */
void oc_tconv1d_k4s4p0(const float* x, const float* w, const float* b, float* y, int B, int Cin, int Cout,
                       int L, int act) {
  const int Lo = 4 * L;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < B; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int i = 0; i < L; ++i)
        for (int k = 0; k < 4; ++k) {
          double acc = b ? b[co] : 0.0;
          for (int ci = 0; ci < Cin; ++ci)
            acc += (double)x[(size_t)(n * Cin + ci) * L + i] * w[(ci * Cout + co) * 4 + k];
          y[(size_t)(n * Cout + co) * Lo + 4 * i + k] = (float)(act ? elu_d(acc) : acc);
        }
}

/* F.linear (+ optional ELU)                                         src/lofar_models.py:80-83,89-91 */
void oc_linear(const float* x, const float* w, const float* b, float* y, int B, int K, int N, int act) {
  for (int n = 0; n < B; ++n)
    for (int o = 0; o < N; ++o) {
      double acc = b ? b[o] : 0.0;
      for (int k = 0; k < K; ++k) acc += (double)x[(size_t)n * K + k] * w[(size_t)o * K + k];
      y[(size_t)n * N + o] = (float)(act ? elu_d(acc) : acc);
    }
}

/* kron(scales, uv) -> cat(sin, cos)                                 src/lofar_models.py:60-62 */
void oc_uv_harmonics(const float* uv, const float* scales, int H, int B, float* out) {
  for (int n = 0; n < B; ++n)
    for (int h = 0; h < H; ++h)
      for (int c = 0; c < 2; ++c) {
        const float a = scales[h] * uv[2 * n + c]; /* product formed in fp32, as torch.kron does */
        out[(size_t)n * 4 * H + 2 * h + c] = (float)sin((double)a);
        out[(size_t)n * 4 * H + 2 * H + 2 * h + c] = (float)cos((double)a);
      }
}

/* Kmeans.forward and its closed-form gradient                       src/lofar_models.py:199-209, SURVEY A.1 */
double oc_khm(const float* X, const float* M, int B, int K, int D, double p, double eps, double* dX, double* dM) {
  const double c = 1.0 / ((double)B * K * D);
  double loss = 0.0;
  double* s = (double*)malloc(sizeof(double) * K);
  double* W = (double*)malloc(sizeof(double) * K);
  if (dM) memset(dM, 0, sizeof(double) * K * D);
  for (int i = 0; i < B; ++i) {
    double e = 0.0;
    for (int k = 0; k < K; ++k) {
      double a = 0.0;
      for (int d = 0; d < D; ++d) {
        const double t = (double)X[(size_t)i * D + d] - M[(size_t)k * D + d];
        a += t * t;
      }
      s[k] = a;
      e += 1.0 / (pow(a, p / 2.0) + eps);
    }
    loss += K / (e + eps);
    for (int k = 0; k < K; ++k) {
      const double g = pow(s[k], p / 2.0) + eps;
      const double pm1 = (s[k] > 0 || p > 2.0) ? pow(s[k], p / 2.0 - 1.0) : 1.0;
      W[k] = c * K / ((e + eps) * (e + eps)) * p * pm1 / (g * g);
    }
    for (int d = 0; d < D; ++d) {
      double gx = 0.0;
      for (int k = 0; k < K; ++k) {
        const double t = (double)X[(size_t)i * D + d] - M[(size_t)k * D + d];
        gx += W[k] * t;
        if (dM) dM[(size_t)k * D + d] -= W[k] * t;
      }
      if (dX) dX[(size_t)i * D + d] = gx;
    }
  }
  free(s);
  free(W);
  return c * loss;
}

/* Kmeans.cluster_similarity                                          src/lofar_models.py:214-229 */
double oc_cluster_similarity(const float* M, int K, int D, double eps) {
  double loss = 0.0;
  for (int i = 0; i < K; ++i) {
    double nii = 0.0;
    for (int d = 0; d < D; ++d) nii += (double)M[(size_t)i * D + d] * M[(size_t)i * D + d];
    const double ni = sqrt(nii);
    const double den = exp(nii / (ni * ni + eps));
    double num = 0.0;
    for (int j = 0; j < K; ++j) {
      if (j == i) continue;
      double dot = 0.0, njj = 0.0;
      for (int d = 0; d < D; ++d) {
        dot += (double)M[(size_t)i * D + d] * M[(size_t)j * D + d];
        njj += (double)M[(size_t)j * D + d] * M[(size_t)j * D + d];
      }
      num += exp(dot / (ni * sqrt(njj) + eps));
    }
    loss += num / (den + eps);
  }
  return loss / ((double)K * D);
}

/* augmented_loss(mu, bpb, batch_size)                                src/kharmonic_lofar.py:97-110 */
double oc_augmented_loss(const float* Z, int rows, int D, int bpb, int batch_size) {
  double loss = 0.0;
  for (int g = 0; g < batch_size; ++g) {
    double prod = 0.0;
    for (int i = 0; i < bpb; ++i) {
      const int ri = g * bpb + i;
      if (ri >= rows) break;
      double ni = 0.0;
      for (int d = 0; d < D; ++d) ni += (double)Z[(size_t)ri * D + d] * Z[(size_t)ri * D + d];
      ni = sqrt(ni) + 1e-6;
      for (int j = i + 1; j < bpb; ++j) {
        const int rj = g * bpb + j;
        if (rj >= rows) break;
        double nj = 0.0, dot = 0.0;
        for (int d = 0; d < D; ++d) {
          nj += (double)Z[(size_t)rj * D + d] * Z[(size_t)rj * D + d];
          dot += (double)Z[(size_t)ri * D + d] * Z[(size_t)rj * D + d];
        }
        nj = sqrt(nj) + 1e-6;
        prod += exp(-dot / (ni * nj));
      }
    }
    loss += prod / bpb;
  }
  return loss / ((double)batch_size * bpb);
}

/* fftn(dims 2,3, ortho) -> roll by N/2 -> cat(Re, Im) -> clamp: direct O(N^3) DFT (separable)
 * Demo.ipynb:169-175, src/lofar_tools.py:24-30.  x (planes, N, N) -> re, im (planes, N, N) */
void oc_fft2_features(const float* x, float* re, float* im, int planes, int N, double clampv) {
  double* cr = (double*)malloc(sizeof(double) * N);
  double* sr = (double*)malloc(sizeof(double) * N);
  for (int k = 0; k < N; ++k) {
    cr[k] = cos(2.0 * M_PI * k / N);
    sr[k] = -sin(2.0 * M_PI * k / N);
  }
#pragma omp parallel for
  for (int pl = 0; pl < planes; ++pl) {
    double* tr = (double*)malloc(sizeof(double) * N * N);
    double* ti = (double*)malloc(sizeof(double) * N * N);
    const float* src = x + (size_t)pl * N * N;
    for (int r = 0; r < N; ++r) /* rows */
      for (int v = 0; v < N; ++v) {
        double ar = 0, ai = 0;
        for (int c = 0; c < N; ++c) {
          const int ph = (int)(((long)v * c) % N);
          ar += src[r * N + c] * cr[ph];
          ai += src[r * N + c] * sr[ph];
        }
        tr[r * N + v] = ar;
        ti[r * N + v] = ai;
      }
    for (int u = 0; u < N; ++u) /* columns, then shift + scale + clamp */
      for (int v = 0; v < N; ++v) {
        double ar = 0, ai = 0;
        for (int r = 0; r < N; ++r) {
          const int ph = (int)(((long)u * r) % N);
          ar += tr[r * N + v] * cr[ph] - ti[r * N + v] * sr[ph];
          ai += tr[r * N + v] * sr[ph] + ti[r * N + v] * cr[ph];
        }
        const int ou = (u + N / 2) % N, ov = (v + N / 2) % N;
        double a = ar / N, b2 = ai / N;
        a = a > clampv ? clampv : (a < -clampv ? -clampv : a);
        b2 = b2 > clampv ? clampv : (b2 < -clampv ? -clampv : b2);
        re[(size_t)pl * N * N + ou * N + ov] = (float)a;
        im[(size_t)pl * N * N + ou * N + ov] = (float)b2;
      }
    free(tr);
    free(ti);
  }
  free(cr);
  free(sr);
}
