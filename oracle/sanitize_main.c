/* TEST INFRASTRUCTURE ONLY: every function of the C restatement (lshm_oracle_c.c) on small, exactly-sized heap buffers under
 * AddressSanitizer + UndefinedBehaviorSanitizer (`make sanitize`, tests/test_oracle_c.py::test_c_oracle_is_clean_under_sanitizers).
 * The restatement is what the parity tests trust; an out-of-bounds read in it would be a silent error in the checker.
 * (GPU-side sanitizers are not available on the pool; the HIP library has no CPU build.) */
#include <stdio.h>

#include "lshm_oracle_c.c"

static float* rnd(size_t n, unsigned* seed) {
  float* p = (float*)malloc(sizeof(float) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    *seed = *seed * 1664525u + 1013904223u;
    p[i] = (float)((*seed >> 8) & 0xffff) / 65536.f - 0.5f;
  }
  return p;
}
static int finite_all(const float* p, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!(fabs((double)p[i]) < 1e30)) return 0;
  return 1;
}
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "sanitize_main: check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
  unsigned seed = 12345u;
  const int B = 3;
  { /* 2-D conv / transposed conv, k4 s2 p1 */
    const int Ci = 4, Co = 6, H = 8, W = 16;
    float *x = rnd((size_t)B * Ci * H * W, &seed), *w = rnd((size_t)Co * Ci * 16, &seed), *b = rnd(Co, &seed);
    float* y = (float*)malloc(sizeof(float) * B * Co * (H / 2) * (W / 2));
    oc_conv2d_k4s2p1(x, w, b, y, B, Ci, Co, H, W, 1);
    CHECK(finite_all(y, (size_t)B * Co * (H / 2) * (W / 2)));
    float *wt = rnd((size_t)Ci * Co * 16, &seed);
    float* yt = (float*)malloc(sizeof(float) * B * Co * 2 * H * 2 * W);
    oc_tconv2d_k4s2p1(x, wt, b, yt, B, Ci, Co, H, W, 0);
    CHECK(finite_all(yt, (size_t)B * Co * 4 * H * W));
    free(x); free(w); free(b); free(y); free(wt); free(yt);
  }
  { /* 1-D conv k4 s4 p1 / transposed conv k4 s4 p0 */
    const int Ci = 4, Co = 5, L = 64;
    float *x = rnd((size_t)B * Ci * L, &seed), *w = rnd((size_t)Co * Ci * 4, &seed), *b = rnd(Co, &seed);
    float* y = (float*)malloc(sizeof(float) * B * Co * ((L - 2) / 4 + 1));
    oc_conv1d_k4s4p1(x, w, b, y, B, Ci, Co, L, 1);
    CHECK(finite_all(y, (size_t)B * Co * ((L - 2) / 4 + 1)));
    float *wt = rnd((size_t)Ci * Co * 4, &seed);
    float* yt = (float*)malloc(sizeof(float) * B * Co * 4 * L);
    oc_tconv1d_k4s4p0(x, wt, b, yt, B, Ci, Co, L, 1);
    CHECK(finite_all(yt, (size_t)B * Co * 4 * L));
    free(x); free(w); free(b); free(y); free(wt); free(yt);
  }
  { /* dense layer, harmonic features */
    const int K = 20, N = 7, H = 4;
    float *x = rnd((size_t)B * K, &seed), *w = rnd((size_t)N * K, &seed), *b = rnd(N, &seed);
    float* y = (float*)malloc(sizeof(float) * B * N);
    oc_linear(x, w, b, y, B, K, N, 1);
    oc_linear(x, w, NULL, y, B, K, N, 0);
    CHECK(finite_all(y, (size_t)B * N));
    float *uv = rnd(2 * B, &seed), *sc = rnd(H, &seed);
    float* out = (float*)malloc(sizeof(float) * B * 4 * H);
    oc_uv_harmonics(uv, sc, H, B, out);
    CHECK(finite_all(out, (size_t)B * 4 * H));
    free(x); free(w); free(b); free(y); free(uv); free(sc); free(out);
  }
  { /* K-harmonic means (with and without gradients), centroid similarity, augmented loss */
    const int Bk = 9, K = 5, D = 12, bpb = 3;
    float *X = rnd((size_t)Bk * D, &seed), *M = rnd((size_t)K * D, &seed);
    double* dX = (double*)malloc(sizeof(double) * Bk * D);
    double* dM = (double*)malloc(sizeof(double) * K * D);
    const double l0 = oc_khm(X, M, Bk, K, D, 3.5, 1e-9, dX, dM), l1 = oc_khm(X, M, Bk, K, D, 3.5, 1e-9, NULL, NULL);
    CHECK(l0 == l1 && l0 > 0.0 && l0 < 1e30);
    const double cs = oc_cluster_similarity(M, K, D, 1e-9);
    CHECK(fabs(cs) < 1e30);
    const double al = oc_augmented_loss(X, Bk, D, bpb, Bk / bpb);
    CHECK(al >= 0.0 && al < 1e30);
    free(X); free(M); free(dX); free(dM);
  }
  { /* FFT feature op */
    const int planes = 2, N = 8;
    float* x = rnd((size_t)planes * N * N, &seed);
    float* re = (float*)malloc(sizeof(float) * planes * N * N);
    float* im = (float*)malloc(sizeof(float) * planes * N * N);
    oc_fft2_features(x, re, im, planes, N, 1e3);
    CHECK(finite_all(re, (size_t)planes * N * N) && finite_all(im, (size_t)planes * N * N));
    free(x); free(re); free(im);
  }
  printf("sanitize_main: ok\n");
  return 0;
}
