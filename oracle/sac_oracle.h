/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the sac optimizer / predictor
 * cascade hot path. Never linked into, imported by, or called from the product path
 * (sac_amd/, libsacopt.so). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / the timed CPU baseline.
 *
 * Pinned: bit-exact against the real reference compiled in oracle/_ref (see
 * tests/test_oracle_vs_ref.py) and against tests/golden/ fixtures generated from it.
 */
#ifndef SAC_ORACLE_H
#define SAC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* x86 rsqrtss model (reference: src/pred/nlms.h:34-57). tab==NULL selects the native
 * instruction of the CPU the oracle runs on; otherwise tab holds 2^(bits+1) floats indexed
 * by (exponent-field parity << bits) | (top `bits` mantissa bits), for exponent fields
 * 127 (parity 1, x in [1,2)) and 128 (parity 0, x in [2,4)). */
void  oracle_set_rsqrt_table(const float *tab, int bits);
float oracle_rsqrt(float x);
float oracle_rsqrt_native(float x);
/* Probe the host CPU: fills tab (capacity cap floats), returns bits (>=0) or -1 if the
 * model (parity x top-k mantissa bits, exponent-scale invariant) does not hold / cap too small. */
int   oracle_probe_rsqrt(float *tab, int cap);

int    oracle_base_profile(int profile, double *out /* ncoef x {vmin,vmax,vdef} */);
int    oracle_cholesky_kat(void);
int    oracle_chol_solve(int n, double regp, int fast, const double *snap, double *sol);

/* FrameCoder::PredictStereoFrame (src/codec.cpp:154-214). s0/s1 hold the whole frame
 * (`total` samples, used for Remap::Analyse as FrameCoder::Predict does, codec.cpp:448-452);
 * the window [from, from+n) is predicted from a fresh predictor state.
 * Any output pointer may be NULL. maxbpn = {maxbpn0,maxbpn1,maxbpn_map0,maxbpn_map1}. */
int oracle_predict_stereo_frame(int profile, const double *coefs, const int32_t *s0,
                                const int32_t *s1, int total, int from, int n, int32_t *e0,
                                int32_t *e1, int32_t *u0, int32_t *u1, int32_t *m0, int32_t *m1,
                                int *maxbpn);

/* CostGolomb::Calc / CostMeanRMS::Calc (src/codec.cpp:302-339). */
double oracle_cost_golomb(const int32_t *buf, int n, double *state_inout);
double oracle_cost_meanrms(const int32_t *buf, int n);

/* FrameCoder::Optimize (src/codec.cpp:354-408) incl. GetCost (:341-352). prof = ncoef x
 * {vmin,vmax,vdef} (NULL -> base profile); ncoef_loop = number of coefs searched (8 in the
 * reference). trace rows {coef(-1 = baseline), x, cost}; returns rows. */
int oracle_optimize(int profile, const double *prof, int ncoef_loop, const int32_t *s0,
                    const int32_t *s1, int numsamples, double *best, double *trace, int max_rows);

/* Stage-level entry points used to pin individual kernels. */
void oracle_lms_pair(int S, int D, double mu1, double mu2, const double *x0, const double *x1,
                     int n, double *p0, double *p1);
void oracle_lpc3_pair(double alpha, int S, int D, const int32_t *s0, const int32_t *s1, int n, double *p0,
                      double *p1);
void oracle_rls_run(int n, double alpha, const double *pv, const double *val, int len, double *p);
void oracle_cascade_trace(int profile, const double *coefs, const int32_t *s0, const int32_t *s1,
                          int n, double *p_out, double *v_out, int32_t *e0, int32_t *e1);
/* FrameCoder::UnpredictStereoFrame (src/codec.cpp:216-235) with Stereo{Fast,Normal,High}::Unpredict
 * (src/profile.cpp:34-43, 89-100, 161-174) and Remap::Unmap (src/coder/map.cpp:154-168).
 * used0/used1: NULL = not mapped, else usedl[0..32768] then usedh[0..32768] as bytes; minmax: {min0,max0,min1,max1} or NULL. */
int oracle_unpredict_stereo_frame(int profile, const double *coefs, const int32_t *e0, const int32_t *e1, int n,
                                  const unsigned char *used0, const unsigned char *used1, const int32_t *minmax,
                                  int32_t *v0, int32_t *v1);
void oracle_remap_tables(const int32_t *src, int n, unsigned char *used);
void oracle_remap_map(const int32_t *src, int n, const int32_t *pred, const int32_t *err, int m,
                      int32_t *out);

#ifdef __cplusplus
}
#endif
/* sac_coder.c: the host-side bit-plane coder, restated; payload sizes / bytes of one channel */
int oracle_encode_mono_normal(const int32_t *s2u, int numsamples, int maxbpn, unsigned char *out, int cap);
int oracle_encode_mono_mapped(const int32_t *s2u_map, int numsamples, int maxbpn_map, const unsigned char *used, unsigned char *out, int cap);

int oracle_optimize_cf(int profile, const double *prof_in, int ncoef_loop, const int32_t *s0, const int32_t *s1,
                       int numsamples, int cost_fn, double *best, double *trace, int max_rows);

#endif

