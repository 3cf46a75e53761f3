// Job descriptors shared by the kernels (kernels.hip) and the host orchestration (sacopt.cpp).
// One descriptor = one (frame, candidate) chain for one stage kernel; arrays of descriptors are
// uploaded once per launch and indexed by blockIdx.x.
#ifndef SACOPT_JOBS_H
#define SACOPT_JOBS_H
#include <stdint.h>

#define LMS_CHUNK 512     // samples staged into the LDS history ring at a time
#ifndef RSQ_LDS_MAX
#define RSQ_LDS_MAX 8192  // rsqrt table entries kept in LDS (bits <= 12: Intel needs 10, AMD Zen 12)
#endif
// Table entries as 16-bit words: the floats are in [0.5, 2) with the low mantissa byte clear (top byte 0x3F,
// low byte 0), so bits 8..23 carry everything. sacopt_create checks that every entry fits.
#define RSQ_PACKABLE(u) ((((u) >> 24) == 0x3Fu) && (((u) & 0xFFu) == 0u))
#define RSQ_PACK(u) ((uint16_t)(((u) >> 8) & 0xFFFFu))
#define RSQ_UNPACK(v) (0x3F000000u | ((uint32_t)(v) << 8))

struct LmsJob {
  const double *x[2];  // stage inputs val[ch][k]      (n doubles each)
  double *p[2];        // stage predictions p[k]
  double *xn[2];       // next stage inputs val[ch][k+1] = x - p (may be NULL)
  double mu1, mu2;
  int32_t n;           // samples to process in this launch: absolute times [t0, t0+n)
  int32_t t0;          // chunk start (multiple of LMS_CHUNK); arrays are indexed by absolute time
  double *state;       // w/eg carried between chunks ([2][R][NT] doubles); NULL = single launch
};

struct LpcJob {
  const int32_t *s[2];  // window samples (already offset by `from`)
  double *p[2];         // LPC3: p[0]
  double *x1[2];        // LPC3: val[ch][1] = v - p[0]
  int32_t *e[2];        // LPC2 (fast profile): error[ch]
  int32_t *u[2];        // LPC2: s2u_error[ch]
  int32_t *umax;        // LPC2: [2] max s2u (must be zeroed before the launch)
  double alpha;
  int32_t n, nsnap;     // this launch covers times [t0, n); nsnap = n / k  (k = 8 for LPC3, 32 for LPC2)
  int32_t t0;           // chunk start (multiple of 512); 0 = from the beginning
  double *cstate[2];    // [324] covariance + b carried between chunks (NULL = single launch)
  double *estate[2];    // [24]  BlendErr running errors carried between chunks
  // scratch, per channel (released after the stage has run)
  double *snap[2];      // [nsnap][324] covariance (packed lower triangle) + b after every k-th Update
  double *sol[2];       // [nsnap][300] all-orders solutions (LPC2: [nsnap][24])
  int32_t *fail[2];     // [nsnap] 1 = "matrix indefinit", previous solution stays in force
  double *ss[2];        // [n][24] all-orders predictions   (LPC3)
  double *serr[2];      // [n][24] BlendErr running errors  (LPC3)
};

struct MixJob {
  const int32_t *s[2];
  const double *p0[2];
  const double *pk[4][2];  // LMS stage predictions p[1..NS]
  const double *x1[2];     // val[ch][1] (RLS update target, profile.cpp:184)
  int32_t *e[2];           // error[ch]           (may be NULL)
  int32_t *u[2];           // s2u_error[ch]       (may be NULL)
  int32_t *umax;           // [2] max s2u          (may be NULL)
  double alpha;
  int32_t n, t0;           // absolute times [t0, t0+n)
  double *state;           // RLS P,w (+max) carried between chunks ([2][MIX_STATE] doubles); NULL = single launch
  double *pout[2];         // RLS predictions p[t] (stage-level test entry sacopt_rls_run only; NULL in the cascades)
  int32_t direct;          // 1: pk[k] already are the mixer inputs pv[k] (sacopt_rls_run); 0: pv = cumulative sums of pk (profile.cpp:134)
};
#define MIX_STATE 24

struct CostJob {
  const int32_t *const *bufs;  // nbuf residual buffers in CostGolomb call order (cand0 ch0, cand0 ch1, ...)
  double *costs;               // nbuf/2 outputs: (Calc(ch0)+Calc(ch1))/2
  double *state;               // carried running mean (in/out), may be NULL
  int32_t n, nbuf;  // nbuf < 0: single-buffer mode, costs[0] = Calc(bufs[0])
  double *scr;      // buffer-parallel form: [3*nbuf] guessed ends F, actual ends E, per-buffer costs (else NULL)
  int32_t *ok;      // buffer-parallel form: [1] 1 = verified, the sequential kernel has nothing to do (else NULL)
};

// Remap (src/coder/map.cpp:89-152) for one (frame, channel).
#define MAP_SCALE 32768
#define MAP_TAB (2 * MAP_SCALE + 1)  // values -scale..scale
struct MapJob {
  const int32_t *s;   // whole frame (numsamples) -- Remap::Analyse input (codec.cpp:448-452)
  const int32_t *sw;  // s + from: the predicted window
  const int32_t *e;   // error[ch][0..n)
  int32_t *m;         // out: s2u_error_map[ch][0..n) (may be NULL: only the max is wanted)
  int32_t *tab;       // [MAP_TAB] scratch, zeroed before k_map_mark; after k_map_scan: tab[v+scale] = #used values in [-scale, v]
  int32_t *mmax;      // [1] max of m (zeroed before the launch)
  int32_t numsamples, n;
};

// UnpredictStereoFrame (src/codec.cpp:216-235) for one frame.
#define UN_MAPW 2052  // 32-bit words per table of the decoder's used-value map (65537 bits = 2049 words, padded)
struct UnJob {
  const int32_t *e[2];     // residuals error[ch][0..n) as the entropy decoder delivers them (mapped where map[ch] != NULL)
  int32_t *v[2];           // out: samples
  const uint32_t *map[2];  // NULL = channel not mapped (enc_mapped false); else [2*UN_MAPW]: bitmap of used values
                           // (bit v+32768, value 0 always set) then the number of set bits in the words before each word
  int32_t n;
  int32_t minval[2], maxval[2];  // StereoHigh::Unpredict clamps its prediction to these (profile.cpp:163,170)
  double coef[8];
};

#endif
