/* TEST INFRASTRUCTURE: drives the oracle's entry points (oracle/sac_oracle.c, oracle/sac_coder.c) on small inputs -- every
 * profile, ragged windows, mapped channels, the coder on plain and mapped residuals, degenerate inputs -- in a build with
 * -fsanitize=address,undefined (oracle/Makefile: `make sanitize`). Exit status 0 and no sanitizer report = pass.
 * (The encode -> decode round trips are checked here as well, so the run is not just "did not crash".) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/sac_oracle.h"
int oracle_unpredict_stereo_frame(int profile, const double *coefs, const int32_t *e0, const int32_t *e1, int n,
                                  const unsigned char *used0, const unsigned char *used1, const int32_t *minmax, int32_t *v0, int32_t *v1);
void oracle_remap_tables(const int32_t *src, int n, unsigned char *used);
int oracle_encode_mono_normal(const int32_t *s2u, int numsamples, int maxbpn, unsigned char *out, int cap);
int oracle_encode_mono_mapped(const int32_t *s2u_map, int numsamples, int maxbpn_map, const unsigned char *used, unsigned char *out, int cap);

static uint32_t rng = 12345;
static int32_t rnd(int amp) { rng = rng * 1664525u + 1013904223u; return (int32_t)((rng >> 8) % (2u * (unsigned)amp + 1u)) - amp; }

int main(void) {
  enum { N = 3000 };
  static int32_t s0[N], s1[N], e0[N], e1[N], u0[N], u1[N], m0[N], m1[N], v0[N], v1[N];
  static unsigned char used[2][2 * 32769], out[16 * N + 70000];
  int acc = 0;
  for (int i = 0; i < N; i++) { acc += (i / 50) & 1 ? 211 : -197; s0[i] = acc + rnd(300); s1[i] = (acc * 3) / 4 + rnd(200); }
  int fails = 0;
  if (!oracle_cholesky_kat()) { printf("cholesky KAT failed\n"); fails++; }
  for (int prof = 0; prof <= 3; prof++) {
    const int lens[4] = {N, 1, 0, 257};
    for (int li = 0; li < 4; li++) {
      const int n = lens[li];
      int mb[4] = {0, 0, 0, 0};
      if (oracle_predict_stereo_frame(prof, NULL, s0, s1, n, 0, n, e0, e1, u0, u1, m0, m1, mb)) { printf("predict rc\n"); fails++; }
      if (oracle_unpredict_stereo_frame(prof, NULL, e0, e1, n, NULL, NULL, NULL, v0, v1)) { printf("unpredict rc\n"); fails++; }
      if (memcmp(v0, s0, 4 * (size_t)n) || memcmp(v1, s1, 4 * (size_t)n)) { printf("round trip failed: profile %d n %d\n", prof, n); fails++; }
      if (n && prof < 3) {
        const int k1 = oracle_encode_mono_normal(u0, n, mb[0], out, (int)sizeof(out));
        oracle_remap_tables(s1, n, used[1]);
        const int k2 = oracle_encode_mono_mapped(m1, n, mb[3], used[1], out, (int)sizeof(out));
        if (k1 <= 0 || k2 <= 0) { printf("coder size\n"); fails++; }
      }
    }
    /* a window inside the frame, custom coefficients */
    double base[24], cf[8];
    const int nc = oracle_base_profile(prof, base);
    for (int i = 0; i < nc; i++) cf[i] = base[3 * i + 2];
    if (oracle_predict_stereo_frame(prof, cf, s0, s1, N, 700, 900, e0, e1, u0, u1, NULL, NULL, NULL)) fails++;
  }
  { /* Optimize on a short frame, every cost function; CostGolomb state carried */
    double best[8], trace[3 * 128];
    int oracle_optimize_cf(int, const double *, int, const int32_t *, const int32_t *, int, int, double *, double *, int);
    for (int cfn = 0; cfn < 3; cfn++)
      if (oracle_optimize_cf(2, NULL, 8, s0, s1, 1600, cfn, best, trace, 128) != 84) { printf("optimize rows\n"); fails++; }
    if (oracle_optimize(1, NULL, 6, s0, s1, 1600, best, trace, 128) <= 0) fails++;
    double st = 0.;
    (void)oracle_cost_golomb(e0, 900, &st);
    (void)oracle_cost_golomb(e1, 0, &st);
    (void)oracle_cost_meanrms(e0, 0);
  }
  { /* garbage residuals through the decoder with mapped channels and a narrow clamp: must terminate, stay in bounds */
    for (int i = 0; i < N; i++) { e0[i] = rnd(40000); e1[i] = rnd(300); }
    oracle_remap_tables(s0, N, used[0]);
    const int32_t mm[4] = {-100, 100, -5, 5};
    for (int prof = 1; prof <= 3; prof++) oracle_unpredict_stereo_frame(prof, NULL, e0, e1, 600, used[0], used[1], mm, v0, v1);
  }
  printf(fails ? "san_oracle: %d failure(s)\n" : "san_oracle ok\n", fails);
  return fails != 0;
}
