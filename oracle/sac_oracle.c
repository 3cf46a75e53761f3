/* TEST INFRASTRUCTURE ONLY -- see sac_oracle.h.
 *
 * Plain-C restatement of the sac (autch/sac v0.0.4) optimizer + stereo predictor cascade.
 * Flat arrays, one function per reference class; every function cites the reference
 * file:line it follows (paths relative to /root/reference). Arithmetic keeps the reference's
 * operation order; build with -ffp-contract=off and no FMA-enabling -march (oracle/Makefile).
 */
#include "sac_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NMAX 32 /* max OLS order (reference uses S+D = 24) */

/* ------------------------------------------------------------------ rsqrtss model */
/* src/pred/nlms.h:34-57: movss / rsqrtss / movss, result used without Newton step. */
static const float *g_tab = 0;
static int g_bits = 0;

float oracle_rsqrt_native(float x) {
  float r;
  __asm__ __volatile__("movss %1, %%xmm0\n"
                       "rsqrtss %%xmm0, %%xmm1\n"
                       "movss %%xmm1, %0\n"
                       : "=m"(r)
                       : "m"(x)
                       : "xmm0", "xmm1");
  return r;
}

static float rsqrt_table(const float *tab, int bits, float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  uint32_t e = (u >> 23) & 0xff, mant = u & 0x7fffff;
  float r;
  if (u >> 31) { /* negative (or -0): not reachable from LMSADA2 (arg >= 0.001) */
    return oracle_rsqrt_native(x);
  }
  if (e == 255) {
    if (mant == 0) return 0.0f; /* +inf -> +0 */
    u |= 0x400000;              /* quiet the NaN */
    memcpy(&r, &u, 4);
    return r;
  }
  if (e == 0) return oracle_rsqrt_native(x); /* zero / denormal: not reachable */
  uint32_t par = e & 1;
  uint32_t idx = (par << bits) | (bits ? (mant >> (23 - bits)) : 0);
  uint32_t tb;
  memcpy(&tb, &tab[idx], 4);
  /* canonical exponent field: 127 for parity 1, 128 for parity 0; e - ec = 2q */
  int ec = par ? 127 : 128;
  int q = ((int)e - ec) / 2; /* exact: e-ec is even */
  tb = (uint32_t)((int32_t)tb - (q << 23));
  memcpy(&r, &tb, 4);
  return r;
}

void oracle_set_rsqrt_table(const float *tab, int bits) { /* the table is copied: the caller's buffer may go away */
  static float copy[1 << 14];
  if (tab && bits >= 0 && bits <= 13) {
    memcpy(copy, tab, sizeof(float) * ((size_t)2 << bits));
    g_tab = copy;
  } else g_tab = 0;
  g_bits = bits;
}

float oracle_rsqrt(float x) { return g_tab ? rsqrt_table(g_tab, g_bits, x) : oracle_rsqrt_native(x); }

int oracle_probe_rsqrt(float *tab, int cap) {
  const uint32_t NM = 1u << 23;
  uint32_t *out = (uint32_t *)malloc(2 * (size_t)NM * 4);
  if (!out) return -1;
  for (uint32_t par = 0; par < 2; par++) {
    uint32_t e = par ? 127 : 128;
    for (uint32_t m = 0; m < NM; m++) {
      uint32_t u = (e << 23) | m;
      float x, r;
      memcpy(&x, &u, 4);
      r = oracle_rsqrt_native(x);
      memcpy(&out[par * NM + m], &r, 4);
    }
  }
  int bits = -1;
  for (int k = 0; k <= 23; k++) {
    uint32_t mask = ~((1u << (23 - k)) - 1) & (NM - 1);
    int ok = 1;
    for (uint32_t par = 0; par < 2 && ok; par++)
      for (uint32_t m = 0; m < NM; m++)
        if (out[par * NM + m] != out[par * NM + (m & mask)]) { ok = 0; break; }
    if (ok) { bits = k; break; }
  }
  if (bits < 0 || (2 << bits) > cap) { free(out); return -1; }
  for (uint32_t par = 0; par < 2; par++)
    for (uint32_t i = 0; i < (1u << bits); i++)
      memcpy(&tab[(par << bits) | i], &out[par * NM + (i << (23 - bits))], 4);
  free(out);
  /* exponent-scale invariance: every normal exponent x every table cell x 3 mantissa fills */
  for (uint32_t e = 1; e < 255; e++)
    for (uint32_t i = 0; i < (1u << bits); i++)
      for (int f = 0; f < 3; f++) {
        uint32_t low = (1u << (23 - bits)) - 1;
        uint32_t m = (i << (23 - bits)) | (f == 0 ? 0 : f == 1 ? low : (low >> 1));
        uint32_t u = (e << 23) | m;
        float x, a, b;
        memcpy(&x, &u, 4);
        a = oracle_rsqrt_native(x);
        b = rsqrt_table(tab, bits, x);
        if (memcmp(&a, &b, 4)) return -1;
      }
  return bits;
}

/* ------------------------------------------------------------------ RunExp (src/utils.h:8-18) */
static inline double runexp(double alpha, double sum, double val) { return alpha * sum + (1. - alpha) * val; }

/* ------------------------------------------------------------------ LMSADA2 (src/pred/nlms.h:96-139) */
typedef struct {
  int S, D;
  double mu1, mu2, pred;
  double *w, *eg, *hist;
} lms_t;

static void lms_init(lms_t *l, int S, int D, double mu1, double mu2) {
  l->S = S; l->D = D; l->mu1 = mu1; l->mu2 = mu2; l->pred = 0.;
  l->w = (double *)calloc(S + D, sizeof(double));
  l->eg = (double *)calloc(S + D, sizeof(double));
  l->hist = (double *)calloc(S + D, sizeof(double));
}
static void lms_free(lms_t *l) { free(l->w); free(l->eg); free(l->hist); }

/* nlms.h:107-113 */
static double lms_predict(lms_t *l, double val1) {
  const int S = l->S, D = l->D;
  for (int i = (S + D) - 1; i > S; i--) l->hist[i] = l->hist[i - 1];
  l->hist[S] = val1;
  double pred = 0.;
  for (int i = 0; i < S + D; i++) pred += l->w[i] * l->hist[i];
  l->pred = pred;
  return pred;
}
/* nlms.h:114-131 */
static void lms_update(lms_t *l, double val) {
  const int S = l->S, D = l->D;
  const double err = val - l->pred;
  const double rho = 0.95;
  double mu = l->mu1;
  for (int i = 0; i < S; i++) {
    const double grad = err * l->hist[i];
    l->eg[i] = rho * l->eg[i] + (1.0 - rho) * (grad * grad);
    l->w[i] += (mu * grad * oracle_rsqrt((float)(l->eg[i] + 0.001)));
  }
  mu = l->mu2;
  for (int i = S; i < S + D; i++) {
    const double grad = err * l->hist[i];
    l->eg[i] = rho * l->eg[i] + (1.0 - rho) * (grad * grad);
    l->w[i] += (mu * grad * oracle_rsqrt((float)(l->eg[i] + 0.001)));
  }
  for (int i = S - 1; i > 0; i--) l->hist[i] = l->hist[i - 1];
  l->hist[0] = val;
}

/* ------------------------------------------------------------------ AutoCov (src/math/cov.h:13-51) */
typedef struct {
  int n;
  double alpha;
  double c[NMAX][NMAX], b[NMAX], hist[NMAX];
} cov_t;

static void cov_init(cov_t *a, double alpha, int n) { /* cov.h:15-18 */
  memset(a, 0, sizeof(*a));
  a->n = n; a->alpha = alpha;
  for (int i = 0; i < n; i++) a->c[i][i] = 1.0;
}
static void cov_update_cov(cov_t *a) { /* cov.h:19-31 (WLS) */
  for (int j = 0; j < a->n; j++)
    for (int i = 0; i <= j; i++) a->c[j][i] = a->alpha * a->c[j][i] + (a->hist[i] * a->hist[j]);
}
static void cov_update_b(cov_t *a, double val) { /* cov.h:32-43 */
  for (int i = 0; i < a->n; i++) a->b[i] = a->alpha * a->b[i] + (val * a->hist[i]);
}
static void cov_update_hist(cov_t *a, double val, int n0, int n1) { /* cov.h:44-47 */
  for (int i = n1; i > n0; i--) a->hist[i] = a->hist[i - 1];
  a->hist[n0] = val;
}

/* ------------------------------------------------------------------ Cholesky (src/math/cholesky.cpp:16-77) */
typedef struct {
  int n;
  double regp;
  double m[NMAX][NMAX], sol[NMAX][NMAX], y[NMAX];
} chol_t;

static void chol_init(chol_t *c, int n) { memset(c, 0, sizeof(*c)); c->n = n; }

static int chol_factor(chol_t *c, double src[NMAX][NMAX]) { /* cholesky.cpp:16-35 */
  const int n = c->n;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) c->m[i][j] = src[i][j];
  for (int i = 0; i < n; i++) c->m[i][i] += c->regp;
  double sum;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < i; j++) {
      sum = c->m[i][j];
      for (int k = 0; k < j; k++) sum -= (c->m[i][k] * c->m[j][k]);
      c->m[i][j] = sum / c->m[j][j];
    }
    sum = c->m[i][i];
    for (int k = 0; k < i; k++) sum -= (c->m[i][k] * c->m[i][k]);
    if (sum > 0.) c->m[i][i] = sqrt(sum);
    else return 1;
  }
  return 0;
}
static void chol_calc_y(chol_t *c, const double *b) { /* cholesky.cpp:37-46 */
  for (int i = 0; i < c->n; i++) {
    double sum = b[i];
    for (int j = 0; j < i; j++) sum -= (c->m[i][j] * c->y[j]);
    c->y[i] = sum / c->m[i][i];
  }
}
static void chol_solve_x(chol_t *c, const double *b, double *x) { /* cholesky.cpp:48-60 */
  chol_calc_y(c, b);
  for (int i = c->n - 1; i >= 0; i--) {
    double sum = c->y[i];
    for (int j = i + 1; j < c->n; j++) sum -= (c->m[j][i] * x[j]);
    x[i] = sum / c->m[i][i];
  }
}
static void chol_solve_all(chol_t *c, const double *b) { /* cholesky.cpp:63-77 */
  chol_calc_y(c, b);
  for (int o = 0; o < c->n; o++)
    for (int i = o; i >= 0; i--) {
      double sum = c->y[i];
      for (int j = i + 1; j <= o; j++) sum -= (c->m[j][i] * c->sol[o][j]);
      c->sol[o][i] = sum / c->m[i][i];
    }
}

int oracle_cholesky_kat(void) { /* cholesky.cpp:3-14, EPS = 1e-14 (global.h:11) */
  static double A[NMAX][NMAX];
  const double a[4][4] = {{9, 3, -6, 12}, {3, 26, -7, -11}, {-6, -7, 9, 7}, {12, -11, 7, 65}};
  const double b[4] = {72, 34, 22, 326}, want[4] = {2, 4, 3, 5};
  double x[4], d = 0.;
  chol_t c;
  chol_init(&c, 4);
  memset(A, 0, sizeof(A));
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) A[i][j] = a[i][j];
  chol_factor(&c, A);
  chol_solve_x(&c, b, x);
  for (int i = 0; i < 4; i++) d += fabs(want[i] - x[i]);
  return d < 1E-14;
}

/* Factor + Solve of one matrix (stage-level driver): snap = packed lower triangle (row by row) + b[n]; fast = 0: all-orders
 * Solve(b) into sol[o(o+1)/2 + i] with `regp` on the diagonal; fast = 1: Solve(b, x) into sol[0..n). Returns 1 = indefinit. */
int oracle_chol_solve(int n, double regp, int fast, const double *snap, double *sol) {
  static double A[NMAX][NMAX];
  chol_t c;
  if (n > NMAX) return -1;
  chol_init(&c, n);
  c.regp = regp;
  memset(A, 0, sizeof(A));
  for (int r = 0, k = 0; r < n; r++)
    for (int q = 0; q <= r; q++, k++) { A[r][q] = snap[k]; A[q][r] = snap[k]; }
  const double *b = snap + n * (n + 1) / 2;
  if (chol_factor(&c, A)) return 1;
  if (fast) chol_solve_x(&c, b, sol);
  else {
    chol_solve_all(&c, b);
    for (int o = 0; o < n; o++)
      for (int i = 0; i <= o; i++) sol[o * (o + 1) / 2 + i] = c.sol[o][i];
  }
  return 0;
}

/* ------------------------------------------------------------------ BlendErr (src/pred/blend.h:7-53) */
typedef struct {
  int n;
  double alpha;
  double serr[NMAX + 1], w[NMAX], p[NMAX], pm;
} blend_t;

static void blend_init(blend_t *b, int n, double alpha) { memset(b, 0, sizeof(*b)); b->n = n; b->alpha = alpha; }
static double blend_predict(blend_t *b, const double *pred) { /* blend.h:13-29 (argmin loop is dead) */
  for (int i = 0; i < b->n; i++) b->p[i] = pred[i];
  b->pm = 0.;
  for (int i = 0; i < b->n; i++) b->pm += b->w[i] * b->p[i];
  return b->pm;
}
static void blend_update(blend_t *b, int32_t val) { /* blend.h:30-46 */
  for (int i = 0; i < b->n; i++) {
    double e = (val - b->p[i]);
    b->serr[i] = runexp(b->alpha, b->serr[i], e * e);
  }
  b->serr[b->n] = runexp(b->alpha, b->serr[b->n], (val - b->pm) * (val - b->pm));
  double wsum = 0;
  for (int i = 0; i < b->n; i++) {
    double e = b->serr[i];
    double t = 10000. / (e + 0.0001);
    b->w[i] = t * t;
    wsum += b->w[i];
  }
  if (wsum > 0)
    for (int i = 0; i < b->n; i++) b->w[i] /= wsum;
}

/* ------------------------------------------------------------------ LPC3 (src/pred/lpc.h:172-228) */
typedef struct {
  cov_t cov;
  chol_t chol;
  blend_t blend;
  double ssum[NMAX];
  int S, D, k, nk;
} lpc3_t;

static void lpc3_init(lpc3_t *l, double alpha, int S, int D, int k, double regp) { /* lpc.h:181-185 */
  cov_init(&l->cov, alpha, S + D);
  chol_init(&l->chol, S + D);
  blend_init(&l->blend, S + D, 0.98);
  memset(l->ssum, 0, sizeof(l->ssum));
  l->S = S; l->D = D; l->k = k; l->nk = 0;
  l->chol.regp = regp;
}
static double lpc3_predict(lpc3_t *l, double val1) { /* lpc.h:186-201 */
  const int n = l->S + l->D;
  if (l->D) cov_update_hist(&l->cov, val1, l->S, n - 1);
  for (int order = 0; order < n; order++) {
    double sum = 0.;
    for (int i = 0; i <= order; i++) sum += l->chol.sol[order][i] * l->cov.hist[i];
    l->ssum[order] = sum;
  }
  return blend_predict(&l->blend, l->ssum);
}
static void lpc3_update(lpc3_t *l, double val0) { /* lpc.h:202-220 */
  blend_update(&l->blend, (int32_t)val0);
  cov_update_cov(&l->cov);
  cov_update_b(&l->cov, val0);
  cov_update_hist(&l->cov, val0, 0, l->S - 1);
  l->nk++;
  if (l->nk >= l->k) {
    l->nk = 0;
    if (chol_factor(&l->chol, l->cov.c)) { /* "matrix indefinit": keep previous sol */
    } else chol_solve_all(&l->chol, l->cov.b);
  }
}

/* ------------------------------------------------------------------ LPC2 (src/pred/lpc.h:254-293) */
typedef struct {
  cov_t cov;
  chol_t chol;
  double sol[NMAX], pred;
  int S, D, k, nk;
} lpc2_t;

static void lpc2_init(lpc2_t *l, double alpha, int S, int D, int k, double regp) {
  cov_init(&l->cov, alpha, S + D);
  chol_init(&l->chol, S + D);
  memset(l->sol, 0, sizeof(l->sol));
  l->S = S; l->D = D; l->k = k; l->nk = 0; l->pred = 0.;
  l->chol.regp = regp;
}
static double lpc2_predict(lpc2_t *l, double val1) { /* lpc.h:261-267 */
  const int n = l->S + l->D;
  if (l->D) cov_update_hist(&l->cov, val1, l->S, n - 1);
  l->pred = 0.;
  for (int i = 0; i < n; i++) l->pred += l->sol[i] * l->cov.hist[i];
  return l->pred;
}
static void lpc2_update(lpc2_t *l, double val) { /* lpc.h:268-284 */
  cov_update_cov(&l->cov);
  cov_update_b(&l->cov, val);
  cov_update_hist(&l->cov, val, 0, l->S - 1);
  l->nk++;
  if (l->nk >= l->k) {
    l->nk = 0;
    if (chol_factor(&l->chol, l->cov.c)) {
    } else chol_solve_x(&l->chol, l->cov.b, l->sol);
  }
}

/* ------------------------------------------------------------------ RLS (src/pred/rls.cpp:61-108) */
typedef struct {
  int n;
  double alpha, p;
  double hist[8], w[8], k[8], P[8][8];
} rls_t;

static void rls_init(rls_t *r, int n, double alpha) { /* rls.cpp:61-65 */
  memset(r, 0, sizeof(*r));
  r->n = n; r->alpha = alpha;
  for (int i = 0; i < n; i++) r->P[i][i] = 1;
}
static double rls_predict(rls_t *r, const double *pred) { /* rls.cpp:67-78 */
  for (int i = 0; i < r->n; i++) r->hist[i] = pred[i];
  r->p = 0.;
  for (int i = 0; i < r->n; i++) r->p += r->hist[i] * r->w[i];
  return r->p;
}
static void rls_update(rls_t *r, double val) { /* rls.cpp:80-108 */
  const int n = r->n;
  double vt1[8], m1[8][8], m2[8][8];
  for (int i = 0; i < n; i++) { /* MatrixMulVec(P,hist), rls.cpp:10-16 */
    double sum = 0.;
    for (int j = 0; j < n; j++) sum += r->P[i][j] * r->hist[j];
    vt1[i] = sum;
  }
  double inner = 0.; /* VecMulVecInner(hist,vt1), rls.cpp:3-8 */
  for (int i = 0; i < n; i++) inner += r->hist[i] * vt1[i];
  const double sc = 1. / (r->alpha + inner);
  for (int i = 0; i < n; i++) r->k[i] = sc * vt1[i]; /* ScalarMulVec, rls.cpp:30-36 */
  for (int j = 0; j < n; j++) /* VecMulVecOuter(k,hist), rls.cpp:38-45 */
    for (int i = 0; i < n; i++) m1[j][i] = r->k[j] * r->hist[i];
  for (int i = 0; i < n; i++) /* MatrixMulMatrix(m1,P), rls.cpp:47-59 */
    for (int j = 0; j < n; j++) {
      double sum = 0.;
      for (int k = 0; k < n; k++) sum += m1[i][k] * r->P[k][j];
      m2[i][j] = sum;
    }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) r->P[i][j] = 1. / r->alpha * (r->P[i][j] - m2[i][j]); /* rls.cpp:99 */
  for (int i = 0; i < n; i++) r->w[i] = r->w[i] + (val - r->p) * r->k[i]; /* rls.cpp:107 */
}

/* ------------------------------------------------------------------ profiles (src/codec.cpp:8-30) */
int oracle_base_profile(int profile, double *out) {
  static const double fast[1][3] = {{0.99, 0.999, 0.996}};
  static const double normal[6][3] = {{0.99, 0.9999, 0.998}, {0.000001, 0.001, 0.0004}, {0.000001, 0.001, 0.0001},
                                      {0.00001, 0.01, 0.003}, {0.0001, 0.1, 0.01},       {0.99, 0.9999, 0.998}};
  static const double high[8][3] = {{0.98, 0.9999, 0.997},       {0.0000001, 0.0001, 0.00001}, {0.0000001, 0.0001, 0.000005},
                                    {0.000001, 0.001, 0.0005},   {0.000001, 0.001, 0.0001},    {0.0001, 0.01, 0.003},
                                    {0.001, 0.1, 0.01},          {0.99, 0.9999, 0.998}};
  const double *src;
  int n;
  if (profile == 0) { src = &fast[0][0]; n = 1; }
  else if (profile == 1) { src = &normal[0][0]; n = 6; }
  else if (profile == 2 || profile == 3) { src = &high[0][0]; n = 8; } /* 3: StereoHigh with LPC3(alpha,24,8,8,500) -- no reference counterpart */
  else return 0;
  for (int i = 0; i < 3 * n; i++) out[i] = src[i];
  return n;
}

/* ------------------------------------------------------------------ stereo cascades (src/profile.cpp) */
#define MAXST 4 /* LMS stages */
typedef struct {
  int type, nlms;
  /* fast */
  lpc2_t lpc2[2];
  double x[2];
  /* normal / high */
  lpc3_t lpc[2];
  lms_t lms[MAXST][2];
  rls_t mix[2];
  double val[2][MAXST + 1], p[MAXST + 1], pv[MAXST];
} stereo_t;

static stereo_t *stereo_new(int type, const double *c) {
  stereo_t *s = (stereo_t *)calloc(1, sizeof(stereo_t));
  s->type = type;
  if (type == 0) { /* StereoFast, profile.cpp:3-8 */
    for (int ch = 0; ch < 2; ch++) lpc2_init(&s->lpc2[ch], c[0], 16, 8, 32, 0.);
    s->x[0] = s->x[1] = 0.;
  } else if (type == 1) { /* StereoNormal, profile.cpp:46-55 */
    s->nlms = 3;
    for (int ch = 0; ch < 2; ch++) {
      lpc3_init(&s->lpc[ch], c[0], 16, 8, 8, 500);
      lms_init(&s->lms[0][ch], 256, 128, c[1], c[2]);
      lms_init(&s->lms[1][ch], 32, 16, c[3], c[3]);
      lms_init(&s->lms[2][ch], 8, 4, c[4], c[4]);
      rls_init(&s->mix[ch], 3, c[5]);
    }
  } else { /* StereoHigh, profile.cpp:113-124; type 3: the same cascade with a 32-tap OLS stage (S = 24, D = 8), a shape the
              reference's LPC3 constructor accepts (lpc.h:181) but no shipped profile uses: parity unpinned at cascade level */
    s->nlms = 4;
    for (int ch = 0; ch < 2; ch++) {
      lpc3_init(&s->lpc[ch], c[0], type == 3 ? 24 : 16, 8, 8, 500);
      lms_init(&s->lms[0][ch], 1280, 640, c[1], c[2]);
      lms_init(&s->lms[1][ch], 256, 128, c[3], c[4]);
      lms_init(&s->lms[2][ch], 32, 16, c[5], c[5]);
      lms_init(&s->lms[3][ch], 8, 4, c[6], c[6]);
      rls_init(&s->mix[ch], 4, c[7]);
    }
  }
  return s;
}
static void stereo_free(stereo_t *s) {
  if (s->type != 0)
    for (int k = 0; k < s->nlms; k++)
      for (int ch = 0; ch < 2; ch++) lms_free(&s->lms[k][ch]);
  free(s);
}

/* StereoNormal::Predict(ch0,ch1) profile.cpp:57-68 / StereoHigh::Predict(ch0,ch1) :126-139 */
static double cascade_predict(stereo_t *s, int ch0, int ch1) {
  s->p[0] = lpc3_predict(&s->lpc[ch0], s->val[ch1][0]);
  for (int k = 0; k < s->nlms; k++) s->p[k + 1] = lms_predict(&s->lms[k][ch0], s->val[ch1][k + 1]);
  s->pv[0] = s->p[1];
  for (int k = 1; k < s->nlms; k++) s->pv[k] = s->pv[k - 1] + s->p[k + 1];
  double p0 = s->p[0];
  double p1 = rls_predict(&s->mix[ch0], s->pv);
  return p0 + p1;
}
/* CalcError profile.cpp:70-76 / :141-148 */
static void cascade_calc_error(stereo_t *s, int ch, int32_t v) {
  s->val[ch][0] = v;
  for (int k = 1; k <= s->nlms; k++) s->val[ch][k] = s->val[ch][k - 1] - s->p[k - 1];
}
/* Update profile.cpp:102-111 / :176-186 */
static void cascade_update(stereo_t *s) {
  for (int ch = 0; ch < 2; ch++) {
    lpc3_update(&s->lpc[ch], s->val[ch][0]);
    for (int k = 0; k < s->nlms; k++) lms_update(&s->lms[k][ch], s->val[ch][k + 1]);
    rls_update(&s->mix[ch], s->val[ch][1]);
  }
}

/* Predict(val0,val1,&err0,&err1) + Update(): profile.cpp:22-32 (fast), :78-87, :150-159 */
static void stereo_step(stereo_t *s, int32_t v0, int32_t v1, int32_t *e0, int32_t *e1) {
  if (s->type == 0) {
    double pa = lpc2_predict(&s->lpc2[0], s->x[1]); /* ch0=0,ch1=1 */
    lpc2_update(&s->lpc2[0], v0);
    s->x[0] = v0;
    double pb = lpc2_predict(&s->lpc2[1], s->x[0]); /* swapped */
    lpc2_update(&s->lpc2[1], v1);
    s->x[1] = v1;
    *e0 = (int32_t)(v0 - round(pa));
    *e1 = (int32_t)(v1 - round(pb));
  } else {
    int32_t p0 = (int32_t)round(cascade_predict(s, 0, 1));
    *e0 = v0 - p0;
    cascade_calc_error(s, 0, v0);
    int32_t p1 = (int32_t)round(cascade_predict(s, 1, 0));
    *e1 = v1 - p1;
    cascade_calc_error(s, 1, v1);
    cascade_update(s);
  }
}

/* ------------------------------------------------------------------ MathUtils (src/utils.h:42-61) */
static int ilog2(int val) { int nbits = 0; while (val >>= 1) nbits++; return nbits; }
static int32_t s2u(int32_t val) {
  if (val < 0) val = 2 * (-val);
  else if (val > 0) val = (2 * val) - 1;
  return val;
}

/* ------------------------------------------------------------------ Remap (src/coder/map.cpp:89-152) */
#define RSCALE (1 << 15)
typedef struct { unsigned char usedl[RSCALE + 1], usedh[RSCALE + 1]; } remap_t;

static void remap_analyse(remap_t *r, const int32_t *src, int n) { /* map.cpp:94-130 */
  memset(r, 0, sizeof(*r));
  for (int i = 0; i < n; i++) {
    int val = src[i];
    if (val > 0) { if (val <= RSCALE) r->usedh[val] = 1; }
    else if (val < 0) { val = -val; if (val <= RSCALE) r->usedl[val] = 1; }
  }
}
static int remap_is_used(const remap_t *r, int val) { /* map.cpp:132-139 */
  if (val > RSCALE) return 0;
  if (val < -RSCALE) return 0;
  if (val > 0) return r->usedh[val];
  if (val < 0) return r->usedl[-val];
  return 1;
}
/* Remap::Map, map.cpp:141-152, line by line. `err = -err` and `pred + sgn * i` are signed overflow when a diverged filter's
 * prediction is far outside the sample range (err == INT32_MIN; a walk that runs over the end of int32): undefined in the
 * reference, "parity unpinned" there. This file is compiled with -fwrapv (oracle/Makefile), which is what the reference's
 * build does in practice: -INT32_MIN stays negative (the loop does not run, 0 comes back), the walk continues at the other
 * end of int32. */
static int32_t remap_map_literal(const remap_t *r, int32_t pred, int32_t err) {
  int sgn = 1;
  if (err == 0) return 0;
  if (err < 0) { err = -err; sgn = -1; }
  int merr = 0;
  for (int i = 1; i <= err; i++)
    if (remap_is_used(r, pred + (sgn * i))) merr++;
  return sgn * merr;
}
/* The same count without the idle steps: isUsed() is 0 outside [-RSCALE, RSCALE] (map.cpp:133-134), so only the i whose
 * (wrapped) pred + sgn * i lies in the table are visited -- a residual of 3e8 is 3e8 loop steps in the literal form, a
 * wrapped walk 2^31. Two runs of i: before the walk leaves int32, and after it has come back at the other end.
 * tests/test_oracle_golden.py::test_remap_map_forms compares the two forms. */
static int32_t remap_map(const remap_t *r, int32_t pred, int32_t err) {
  if (err == 0 || err == INT32_MIN) return 0;
  const int sgn = err < 0 ? -1 : 1;
  const int64_t n = err < 0 ? -(int64_t)err : (int64_t)err, p = pred;
  const int64_t nowrap = sgn > 0 ? (int64_t)INT32_MAX - p : p - (int64_t)INT32_MIN; /* steps before the walk leaves int32 */
  int merr = 0;
  for (int part = 0; part < 2; part++) {
    const int64_t off = part ? ((int64_t)1 << 32) : 0; /* value visited at step i: p + sgn * i - sgn * off */
    int64_t ia = sgn > 0 ? -RSCALE - p + off : p - RSCALE + off, ib = sgn > 0 ? RSCALE - p + off : p + RSCALE + off;
    const int64_t lo = part ? nowrap + 1 : 1, hi = part ? n : (n < nowrap ? n : nowrap);
    if (ia < lo) ia = lo;
    if (ib > hi) ib = hi;
    for (int64_t i = ia; i <= ib; i++)
      if (remap_is_used(r, (int)(p + sgn * i - sgn * off))) merr++;
  }
  return sgn * merr;
}

void oracle_remap_map(const int32_t *src, int n, const int32_t *pred, const int32_t *err, int m, int32_t *out) {
  remap_t *r = (remap_t *)malloc(sizeof(remap_t));
  remap_analyse(r, src, n);
  for (int i = 0; i < m; i++) out[i] = remap_map(r, pred[i], err[i]);
  free(r);
}
void oracle_remap_map_literal(const int32_t *src, int n, const int32_t *pred, const int32_t *err, int m, int32_t *out) {
  remap_t *r = (remap_t *)malloc(sizeof(remap_t));
  remap_analyse(r, src, n);
  for (int i = 0; i < m; i++) out[i] = remap_map_literal(r, pred[i], err[i]);
  free(r);
}

/* ------------------------------------------------------------------ PredictStereoFrame (src/codec.cpp:154-214) */
int oracle_predict_stereo_frame(int profile, const double *coefs, const int32_t *s0, const int32_t *s1,
                                int total, int from, int n, int32_t *e0, int32_t *e1, int32_t *u0,
                                int32_t *u1, int32_t *m0, int32_t *m1, int *maxbpn) {
  double base[8 * 3], c[8];
  int nc = oracle_base_profile(profile, base);
  if (!nc) return 1;
  for (int i = 0; i < nc; i++) c[i] = coefs ? coefs[i] : base[3 * i + 2];
  stereo_t *sp = stereo_new(profile, c);
  remap_t *map0 = 0, *map1 = 0;
  const int want_map = (m0 || m1 || maxbpn);
  if (want_map) { /* FrameCoder::Predict, codec.cpp:448-452 */
    map0 = (remap_t *)malloc(sizeof(remap_t));
    map1 = (remap_t *)malloc(sizeof(remap_t));
    remap_analyse(map0, s0, total);
    remap_analyse(map1, s1, total);
  }
  int32_t emax0 = 0, emax1 = 0, emax0m = 0, emax1m = 0;
  for (int i = 0; i < n; i++) {
    int32_t d0, d1;
    stereo_step(sp, s0[from + i], s1[from + i], &d0, &d1);
    if (e0) e0[i] = d0;
    if (e1) e1[i] = d1;
    int32_t a = s2u(d0), b = s2u(d1);
    if (u0) u0[i] = a;
    if (u1) u1[i] = b;
    if (a > emax0) emax0 = a;
    if (b > emax1) emax1 = b;
    if (want_map) {
      int32_t am = s2u(remap_map(map0, s0[from + i] - d0, d0));
      int32_t bm = s2u(remap_map(map1, s1[from + i] - d1, d1));
      if (m0) m0[i] = am;
      if (m1) m1[i] = bm;
      if (am > emax0m) emax0m = am;
      if (bm > emax1m) emax1m = bm;
    }
  }
  if (maxbpn) {
    maxbpn[0] = ilog2(emax0); maxbpn[1] = ilog2(emax1);
    maxbpn[2] = ilog2(emax0m); maxbpn[3] = ilog2(emax1m);
  }
  stereo_free(sp);
  free(map0); free(map1);
  return 0;
}

/* ------------------------------------------------------------------ UnpredictStereoFrame (src/codec.cpp:216-235) */
static int32_t remap_unmap(const remap_t *r, int32_t pred, int32_t merr) { /* map.cpp:154-168 */
  int sgn = 1;
  if (merr == 0) return 0;
  if (merr < 0) { merr = -merr; sgn = -1; }
  long long err = 1;
  int terr = 0;
  while (1) {
    const long long v = (long long)pred + sgn * err;
    /* past the far edge of the table nothing is used any more: the reference would loop forever (no encoder writes such a
       stream); stop with a sentinel */
    if (sgn > 0 ? v > RSCALE : v < -RSCALE) return sgn * (4 * RSCALE + 1);
    /* before the near edge isUsed() is false as well: the reference walks there one step at a time; same result, skipped */
    if (sgn > 0 ? v < -RSCALE : v > RSCALE) { err = sgn > 0 ? (-(long long)RSCALE - pred) : ((long long)pred - RSCALE); continue; }
    if (remap_is_used(r, (int)v)) terr++;
    if (terr == merr) break;
    err++;
  }
  return (int32_t)(sgn * err);
}
static int32_t clamp_i32(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : v > hi ? hi : v; } /* global.h:16 */

/* Unpredict(err0,err1,&val0,&val1) + Update(): profile.cpp:34-43 (fast: no unmap, no clamp),
 * :89-100 (normal: unmap), :161-174 (high: clamp to [minval,maxval], then unmap) */
static void stereo_unstep(stereo_t *s, int32_t e0, int32_t e1, const remap_t *map0, const remap_t *map1,
                          const int32_t *minval, const int32_t *maxval, int32_t *v0, int32_t *v1) {
  if (s->type == 0) {
    double pa = lpc2_predict(&s->lpc2[0], s->x[1]);
    *v0 = (int32_t)(e0 + round(pa));
    lpc2_update(&s->lpc2[0], *v0);
    s->x[0] = *v0;
    double pb = lpc2_predict(&s->lpc2[1], s->x[0]);
    *v1 = (int32_t)(e1 + round(pb));
    lpc2_update(&s->lpc2[1], *v1);
    s->x[1] = *v1;
  } else {
    int32_t p0 = (int32_t)round(cascade_predict(s, 0, 1));
    if (s->type >= 2) p0 = clamp_i32(p0, minval[0], maxval[0]);
    if (map0) e0 = remap_unmap(map0, p0, e0);
    *v0 = (int32_t)((uint32_t)e0 + (uint32_t)p0); /* garbage streams: the reference's int add overflows (UB); its build wraps */
    cascade_calc_error(s, 0, *v0);
    int32_t p1 = (int32_t)round(cascade_predict(s, 1, 0));
    if (s->type >= 2) p1 = clamp_i32(p1, minval[1], maxval[1]);
    if (map1) e1 = remap_unmap(map1, p1, e1);
    *v1 = (int32_t)((uint32_t)e1 + (uint32_t)p1);
    cascade_calc_error(s, 1, *v1);
    cascade_update(s);
  }
}

/* used0/used1: NULL = channel not mapped (enc_mapped false), else 2*(RSCALE+1) bytes: usedl[0..RSCALE] then usedh[0..RSCALE]
 * (map.h: Remap::usedl/usedh as the MapEncoder delivers them). minmax = {minval0, maxval0, minval1, maxval1} (high profile). */
int oracle_unpredict_stereo_frame(int profile, const double *coefs, const int32_t *e0, const int32_t *e1, int n,
                                  const unsigned char *used0, const unsigned char *used1, const int32_t *minmax,
                                  int32_t *v0, int32_t *v1) {
  double base[8 * 3], c[8];
  int nc = oracle_base_profile(profile, base);
  if (!nc) return 1;
  for (int i = 0; i < nc; i++) c[i] = coefs ? coefs[i] : base[3 * i + 2];
  stereo_t *sp = stereo_new(profile, c);
  remap_t *map0 = 0, *map1 = 0;
  if (used0) { map0 = (remap_t *)malloc(sizeof(remap_t)); memcpy(map0->usedl, used0, RSCALE + 1); memcpy(map0->usedh, used0 + RSCALE + 1, RSCALE + 1); }
  if (used1) { map1 = (remap_t *)malloc(sizeof(remap_t)); memcpy(map1->usedl, used1, RSCALE + 1); memcpy(map1->usedh, used1 + RSCALE + 1, RSCALE + 1); }
  int32_t mn[2] = {minmax ? minmax[0] : INT32_MIN, minmax ? minmax[2] : INT32_MIN};
  int32_t mx[2] = {minmax ? minmax[1] : INT32_MAX, minmax ? minmax[3] : INT32_MAX};
  for (int i = 0; i < n; i++) stereo_unstep(sp, e0[i], e1[i], map0, map1, mn, mx, &v0[i], &v1[i]);
  stereo_free(sp);
  free(map0); free(map1);
  return 0;
}

/* Remap::Analyse as the decoder's tables: 2*(RSCALE+1) bytes, usedl then usedh (map.cpp:111-130) */
void oracle_remap_tables(const int32_t *src, int n, unsigned char *used) {
  remap_t *r = (remap_t *)malloc(sizeof(remap_t));
  remap_analyse(r, src, n);
  memcpy(used, r->usedl, RSCALE + 1);
  memcpy(used + RSCALE + 1, r->usedh, RSCALE + 1);
  free(r);
}

/* ------------------------------------------------------------------ cost functions (src/codec.cpp:302-339) */
double oracle_cost_meanrms(const int32_t *buf, int n) { /* codec.cpp:302-312 */
  if (n) {
    double pw = 0.0;
    for (int i = 0; i < n; i++) pw += abs(buf[i]);
    return pw / (double)n;
  }
  return 0.;
}
double oracle_cost_golomb(const int32_t *buf, int n, double *state) { /* codec.cpp:315-339 */
  double nbits = 0, mean = *state;
  if (n) {
    for (int i = 0; i < n; i++) {
      int32_t val = s2u(buf[i]);
      int m = (int)mean; if (m < 1) m = 1;
      int q = val / m;
      nbits += (q + 1);
      if (m > 1) {
        int b = ceil(log(m) / log(2));
        nbits += b;
      }
      mean = runexp(0.98, mean, val);
    }
    *state = mean;
    return nbits / (double)n;
  }
  return 0;
}

/* ------------------------------------------------------------------ Optimize / GetCost (src/codec.cpp:341-408) */
int oracle_encode_mono_normal(const int32_t *s2u, int numsamples, int maxbpn, unsigned char *out, int cap); /* sac_coder.c */
/* cost_fn: 0 CostGolomb (what the reference's Optimize uses, codec.cpp:364), 1 CostMeanRMS (codec.cpp:302-312), 2 bits per
 * sample of the bit-plane coder's payload (no reference counterpart) */
int oracle_optimize_cf(int profile, const double *prof_in, int ncoef_loop, const int32_t *s0, const int32_t *s1,
                       int numsamples, int cost_fn, double *best, double *trace, int max_rows);
int oracle_optimize(int profile, const double *prof_in, int ncoef_loop, const int32_t *s0, const int32_t *s1,
                    int numsamples, double *best, double *trace, int max_rows) {
  return oracle_optimize_cf(profile, prof_in, ncoef_loop, s0, s1, numsamples, 0, best, trace, max_rows);
}
static double cost_bitplane(const int32_t *u, int n) {
  if (!n) return 0.;
  int32_t mx = 0;
  for (int i = 0; i < n; i++) if (u[i] > mx) mx = u[i];
  return 8. * (double)oracle_encode_mono_normal(u, n, ilog2(mx), 0, 0) / (double)n;
}
int oracle_optimize_cf(int profile, const double *prof_in, int ncoef_loop, const int32_t *s0, const int32_t *s1,
                       int numsamples, int cost_fn, double *best, double *trace, int max_rows) {
  double prof[8 * 3], c[8];
  int nc = oracle_base_profile(profile, prof);
  if (!nc) return -1;
  if (prof_in) memcpy(prof, prof_in, sizeof(double) * 3 * nc);
  if (ncoef_loop > nc) return -1;
  const int n = numsamples / 8;           /* codec.cpp:356-357 */
  const int start = (numsamples - n) / 2; /* codec.cpp:359 */
  int32_t *e0 = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1)), *e1 = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
  int32_t *u0 = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1)), *u1 = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
  double mean = 0.; /* one CostGolomb instance, never reset (codec.cpp:364) */
  int rows = 0;
#define GETCOST(coef, x, dst)                                                                     \
  do {                                                                                            \
    for (int q = 0; q < nc; q++) c[q] = prof[3 * q + 2];                                          \
    c[coef] = (x);                                                                                \
    oracle_predict_stereo_frame(profile, c, s0, s1, numsamples, start, n, e0, e1, u0, u1, 0, 0, 0); \
    double c1, c2;                                                                                \
    if (cost_fn == 1) { c1 = oracle_cost_meanrms(e0, n); c2 = oracle_cost_meanrms(e1, n); }      \
    else if (cost_fn == 2) { c1 = cost_bitplane(u0, n); c2 = cost_bitplane(u1, n); }             \
    else { c1 = oracle_cost_golomb(e0, n, &mean); c2 = oracle_cost_golomb(e1, n, &mean); }        \
    dst = (c1 + c2) / 2.;                                                                         \
  } while (0)
  double best_cost;
  GETCOST(0, prof[2], best_cost);
  if (trace && rows < max_rows) { trace[3 * rows] = -1; trace[3 * rows + 1] = prof[2]; trace[3 * rows + 2] = best_cost; }
  rows++;
  for (int i = 0; i < ncoef_loop; i++) {
    double best_x = prof[3 * i + 2];
    double x = prof[3 * i];
    double xinc = (prof[3 * i + 1] - prof[3 * i]) / 10.;
    while (x <= prof[3 * i + 1]) {
      double cost;
      GETCOST(i, x, cost);
      if (trace && rows < max_rows) { trace[3 * rows] = i; trace[3 * rows + 1] = x; trace[3 * rows + 2] = cost; }
      rows++;
      if (cost < best_cost) { best_x = x; best_cost = cost; }
      x += xinc;
    }
    prof[3 * i + 2] = best_x;
  }
#undef GETCOST
  for (int i = 0; i < nc; i++) best[i] = prof[3 * i + 2];
  free(e0); free(e1); free(u0); free(u1);
  return rows;
}

/* ------------------------------------------------------------------ stage-level drivers */
void oracle_lms_pair(int S, int D, double mu1, double mu2, const double *x0, const double *x1, int n,
                     double *p0, double *p1) {
  lms_t l[2];
  lms_init(&l[0], S, D, mu1, mu2);
  lms_init(&l[1], S, D, mu1, mu2);
  double prev1 = 0.;
  for (int t = 0; t < n; t++) {
    p0[t] = lms_predict(&l[0], prev1);
    p1[t] = lms_predict(&l[1], x0[t]);
    lms_update(&l[0], x0[t]);
    lms_update(&l[1], x1[t]);
    prev1 = x1[t];
  }
  lms_free(&l[0]); lms_free(&l[1]);
}

void oracle_lpc3_pair(double alpha, int S, int D, const int32_t *s0, const int32_t *s1, int n, double *p0, double *p1) {
  lpc3_t *l = (lpc3_t *)malloc(2 * sizeof(lpc3_t));
  lpc3_init(&l[0], alpha, S, D, 8, 500);
  lpc3_init(&l[1], alpha, S, D, 8, 500);
  double prev1 = 0.;
  for (int t = 0; t < n; t++) {
    p0[t] = lpc3_predict(&l[0], prev1);
    p1[t] = lpc3_predict(&l[1], (double)s0[t]);
    lpc3_update(&l[0], (double)s0[t]);
    lpc3_update(&l[1], (double)s1[t]);
    prev1 = (double)s1[t];
  }
  free(l);
}

void oracle_rls_run(int n, double alpha, const double *pv, const double *val, int len, double *p) {
  rls_t r;
  rls_init(&r, n, alpha);
  for (int t = 0; t < len; t++) {
    p[t] = rls_predict(&r, &pv[t * n]);
    rls_update(&r, val[t]);
  }
}

void oracle_cascade_trace(int profile, const double *coefs, const int32_t *s0, const int32_t *s1, int n,
                          double *p_out, double *v_out, int32_t *e0, int32_t *e1) {
  double base[8 * 3], c[8];
  int nc = oracle_base_profile(profile, base);
  if (profile < 1 || profile > 3) return;
  for (int i = 0; i < nc; i++) c[i] = coefs ? coefs[i] : base[3 * i + 2];
  stereo_t *sp = stereo_new(profile, c);
  const int nst = sp->nlms + 1;
  for (int t = 0; t < n; t++) {
    int32_t q0 = (int32_t)round(cascade_predict(sp, 0, 1));
    for (int k = 0; k < nst; k++) p_out[(t * 2 + 0) * nst + k] = sp->p[k];
    cascade_calc_error(sp, 0, s0[t]);
    int32_t q1 = (int32_t)round(cascade_predict(sp, 1, 0));
    for (int k = 0; k < nst; k++) p_out[(t * 2 + 1) * nst + k] = sp->p[k];
    cascade_calc_error(sp, 1, s1[t]);
    for (int ch = 0; ch < 2; ch++)
      for (int k = 0; k < nst; k++) v_out[(t * 2 + ch) * nst + k] = sp->val[ch][k];
    if (e0) e0[t] = s0[t] - q0;
    if (e1) e1[t] = s1[t] - q1;
    cascade_update(sp);
  }
  stereo_free(sp);
}
