/* libsacopt -- MI355X (gfx950) implementation of sac's encoder-side parameter optimizer and
 * stereo predictor cascade, behind a plain C ABI.
 *
 * The reference (autch/sac v0.0.4) has no plugin/FFI layer: the seam is a set of private
 * C++ members of FrameCoder. Each entry point below replaces one of them
 * (reference file:line relative to the reference tree) and keeps its argument meaning.
 * All functions return 0 on success, non-zero on error (sacopt_last_error() has the text);
 * nothing throws across the ABI. Sample / residual pointers may be host or device
 * pointers (detected with hipPointerGetAttributes); host buffers are staged through
 * internal device buffers. One context per GPU (or several: each has its own streams, arena and error text). A context
 * is not thread-safe; different contexts are independent and may be driven from different threads at the same time
 * (sacopt_last_error(NULL) is per thread). The one thing contexts of a GPU share is the static scratch of
 * sacopt_cost_golomb_noalloc, which refuses overlapping calls.
 *
 * There is NO CPU fallback: every entry point fails if no gfx950 device is usable.
 */
#ifndef SACOPT_H
#define SACOPT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sacopt_ctx sacopt_ctx;

#define SACOPT_PROFILE_FAST   0 /* StereoFast   src/profile.cpp:3-43   (1 coef)  */
#define SACOPT_PROFILE_NORMAL 1 /* StereoNormal src/profile.cpp:46-111 (6 coefs) */
#define SACOPT_PROFILE_HIGH   2 /* StereoHigh   src/profile.cpp:113-186 (8 coefs) */
/* StereoHigh with a 32-tap OLS stage, LPC3(alpha, S = 24, D = 8, k = 8, regp = 500): BASELINE configs[4] ("32-tap OLS + long
 * LMS cascade"). Same 8 coefficients and tables as HIGH. The reference's LPC3 takes S and D as constructor arguments
 * (src/pred/lpc.h:181-185) but no profile it ships uses 24 + 8 (src/profile.cpp:116), so the stage is pinned against the
 * reference class and the cascade -- encoder and decoder side -- against the oracle only ("parity unpinned"). */
#define SACOPT_PROFILE_HIGH32 3
#define SACOPT_MAX_COEFS 8

/* One frame of one stereo pair (FrameCoder::samples[0..1], src/codec.h:29). */
typedef struct {
  const int32_t *s0;  /* channel 0 samples, numsamples entries (host or device) */
  const int32_t *s1;  /* channel 1 samples */
  int32_t numsamples; /* FrameCoder::numsamples (src/codec.h:48) */
} sacopt_frame;

/* Residual outputs of PredictStereoFrame for one frame; any pointer may be NULL.
 * (FrameCoder::error / s2u_error / s2u_error_map and framestats[].maxbpn{,_map},
 *  src/codec.cpp:165-211.) Host or device pointers. When map0 or map1 is non-NULL the library
 *  runs Remap::Analyse over the whole frame (as FrameCoder::Predict does, codec.cpp:448-452)
 *  and Remap::Map per sample (src/coder/map.cpp:111-152) on the device; otherwise
 *  maxbpn[2..3] are reported as -1. */
typedef struct {
  int32_t *err0, *err1;     /* error[ch][0..n)          */
  int32_t *s2u0, *s2u1;     /* s2u_error[ch][0..n)      */
  int32_t *map0, *map1;     /* s2u_error_map[ch][0..n)  */
  int32_t maxbpn[4];        /* out: maxbpn0, maxbpn1, maxbpn_map0, maxbpn_map1 (host memory, in the struct) */
} sacopt_residuals;

/* ---- context -------------------------------------------------------------------------- */
/* The LMS stages use the x86 `rsqrtss` approximation (src/pred/nlms.h:34-57), whose value is
 * CPU-specific. rsqrt_tab/rsqrt_bits give the table to emulate: 2^(bits+1) floats indexed by
 * (exponent-field parity << bits) | top `bits` mantissa bits (see sacopt_probe_rsqrt).
 * rsqrt_tab == NULL: probe the CPU this process runs on (what the reference would use here).
 * Entries must be rsqrtss-like: in [0.5, 2) with at most 16 significant bits (Intel and AMD produce 12);
 * the table is kept as 16-bit words on the device and sacopt_create fails for anything else. */
int sacopt_create(int device, const float *rsqrt_tab, int rsqrt_bits, sacopt_ctx **out);
void sacopt_destroy(sacopt_ctx *ctx);
const char *sacopt_last_error(const sacopt_ctx *ctx); /* ctx may be NULL: error of a failed create */

/* Host-only: probe this CPU's rsqrtss. tab must hold `cap` floats; returns bits >= 0, or -1 if
 * the table model does not hold for this CPU or cap < 2^(bits+1). */
int sacopt_probe_rsqrt(float *tab, int cap);

/* SacProfile tables of FrameCoder::FrameCoder (src/codec.cpp:8-30): writes ncoef x
 * {vmin, vmax, vdef}; returns ncoef (0 for an unknown profile). */
int sacopt_base_profile(int profile, double *out);

/* ---- FrameCoder::PredictStereoFrame(profile, 0, 1, from, n)   src/codec.cpp:154-214 ------ */
/* For every frame f: fresh predictor state, predict samples [from[f], from[f]+n[f]) with
 * coefs[f*ncoef .. ) and write the residuals. from/n == NULL means the whole frame.
 * Remap tables are built over the whole frame as FrameCoder::Predict does (codec.cpp:448-452)
 * when map outputs are requested. */
int sacopt_predict_stereo_frames(sacopt_ctx *ctx, int profile, int nframes, const sacopt_frame *frames,
                                 const int32_t *from, const int32_t *n, const double *coefs,
                                 sacopt_residuals *out);

/* ---- CostFunction (src/codec.h:10-15): what GetCost / Optimize evaluate a candidate's residuals with ----------------------
 * GOLOMB   CostGolomb  (src/codec.cpp:315-339): the one FrameCoder::Optimize instantiates (:364); running mean carried from call to call
 * MEANRMS  CostMeanRMS (src/codec.cpp:302-312): the reference's other implementation (mean |e|), instantiated nowhere
 * BITPLANE no counterpart in the reference: 8 x bytes / samples of the payload the host's bit-plane coder
 *          would write for the buffer (sacopt_coded_sizes below; plain variant, a fresh coder per buffer as Encode() has)
 * The choice holds for the context until changed; default GOLOMB. */
#define SACOPT_COST_GOLOMB 0
#define SACOPT_COST_MEANRMS 1
#define SACOPT_COST_BITPLANE 2
int sacopt_set_cost_function(sacopt_ctx *ctx, int which);

/* ---- FrameCoder::GetCost  src/codec.cpp:341-352  (+ CostGolomb::Calc :315-339) ------------- */
/* Evaluates, for every frame, `ncand` candidate coefficient vectors (cand_coefs[(f*ncand+c)*ncoef..])
 * over the window [from[f], from[f]+n[f]) and returns costs[f*ncand+c] =
 * (Calc(err0)+Calc(err1))/2. The CostGolomb running mean is carried from candidate to candidate
 * in index order exactly as one CostGolomb instance would (codec.cpp:364); cost_state[f]
 * (in/out, may be NULL = start from 0) is that running mean. */
int sacopt_get_costs(sacopt_ctx *ctx, int profile, int nframes, const sacopt_frame *frames,
                     const int32_t *from, const int32_t *n, int ncand, const double *cand_coefs,
                     double *costs, double *cost_state);

/* The candidates of the LAST sacopt_get_costs call on this context, costed again -- CostGolomb::Calc over the same
 * residuals in the same order (src/codec.cpp:349-350) -- from another running-mean start state cost_state[f] (in/out;
 * NULL = 0). The cascade is not re-run and the residuals do not leave the device. Any other call on the context discards
 * the residuals (then this fails). Used by the multi-GPU candidate split, where a rank's slice of the
 * grid is first costed from a guessed state and then from the state the previous slice really ended with. */
int sacopt_recost_last(sacopt_ctx *ctx, double *costs, double *cost_state);

/* ---- FrameCoder::Optimize  src/codec.cpp:354-408 --------------------------------------------- */
/* Coordinate grid search over the first `ncoef_loop` coefficients (the reference hard-codes 8,
 * codec.cpp:389; pass the profile's coefficient count for fast/normal). prof = ncoef x
 * {vmin,vmax,vdef} shared by all frames (NULL = base profile). Outputs (host memory):
 *   best[f*ncoef + i]   the selected vdef values,
 *   trace[(f*max_rows + r)*3 ..] optional rows {coef (-1 = baseline call), x, cost} in call order,
 *   rows[f]             optional number of GetCost calls made (84 for the high profile). */
int sacopt_optimize_frames(sacopt_ctx *ctx, int profile, int nframes, const sacopt_frame *frames,
                           const double *prof, int ncoef_loop, double *best, double *trace,
                           int max_rows, int32_t *rows);

/* ---- FrameCoder::Predict  src/codec.cpp:442-461, with the Optimize call of :455-456 enabled ---- */
/* Per frame: Optimize(optprofile) (if ncoef_loop > 0; selected values -> best[f*ncoef..]) and
 * PredictStereoFrame(baseprofile, 0, 1, 0, numsamples) -> out[f]. The stream is coded with the
 * base profile exactly as the reference does (:459), so the two are independent and run
 * concurrently on the GPU. prof = ncoef x {vmin,vmax,vdef} (NULL = base profile). */
int sacopt_encode_frames(sacopt_ctx *ctx, int profile, int nframes, const sacopt_frame *frames,
                         const double *prof, int ncoef_loop, double *best, sacopt_residuals *out);

/* ---- FrameCoder::Predict with the optimiser's result USED  src/codec.cpp:442-461 ---- */
/* What :455-459 would be with `PredictStereoFrame(optprofile, ...)`: per frame Optimize(optprofile) over the first
 * ncoef_loop (>= 1) coefficients -> best[f*ncoef..], then PredictStereoFrame with those values -> out[f]. Same results as
 * sacopt_optimize_frames followed by sacopt_predict_stereo_frames(coefs = best), in one call: the optimiser fixes the
 * coefficients in order (:389-406), so each part of the full-frame pass starts as soon as its own parameters are final
 * instead of after the last round. The stream such frames belong to carries per-frame parameters (sac_amd/host/sacfile.h,
 * SAC_PROFILE_PARAMS); it is not the reference's default stream. */
int sacopt_encode_frames_selected(sacopt_ctx *ctx, int profile, int nframes, const sacopt_frame *frames,
                                  const double *prof, int ncoef_loop, double *best, sacopt_residuals *out);

/* ---- FrameCoder::UnpredictStereoFrame(profile, 0, 1, numsamples)   src/codec.cpp:216-235 ------ */
/* Decoder-side cascade: Stereo{Fast,Normal,High}::Unpredict + Update (src/profile.cpp:34-43,
 * 89-111, 161-186) with Remap::Unmap (src/coder/map.cpp:154-168) for channels coded mapped. One frame of one
 * stereo pair as FrameCoder::Decode leaves it (src/codec.cpp:510-528): */
typedef struct {
  const int32_t *e0, *e1;  /* error[ch][0..numsamples): residuals, mapped where used{0,1} != NULL (host or device)   */
  int32_t *v0, *v1;        /* out: samples[ch][0..numsamples) (host or device)                                      */
  int32_t numsamples;
  const uint8_t *used0, *used1; /* NULL = framestats[ch].enc_mapped false; else the channel's Remap tables as the
                                   MapEncoder decoded them: usedl[0..32768] then usedh[0..32768], one byte each (HOST) */
  int32_t minval[2], maxval[2]; /* framestats[ch].minval/maxval: StereoHigh::Unpredict clamps its prediction to
                                   them (src/profile.cpp:163,170; the encoder does not clamp). Other profiles ignore them. */
} sacopt_dec_frame;
/* coefs: nframes x ncoef predictor parameters (NULL = base profile, what DecodeFile uses, src/codec.cpp:463-478).
 * Fresh predictor state per frame; frames are decoded concurrently, one workgroup each. A mapped residual that asks
 * for more used values than the table holds in its direction (no encoder writes one; the reference would not terminate)
 * decodes as a step of +-131073; every other case, predictions far outside the table included, gives the reference's value. */
int sacopt_unpredict_stereo_frames(sacopt_ctx *ctx, int profile, int nframes, const sacopt_dec_frame *frames,
                                   const double *coefs);

/* ---- CostGolomb::Calc / CostMeanRMS::Calc on caller-supplied residuals (src/codec.cpp:302-339) - */
int sacopt_cost_golomb(sacopt_ctx *ctx, const int32_t *buf, int n, double *state_inout, double *cost);
int sacopt_cost_meanrms(sacopt_ctx *ctx, const int32_t *buf, int n, double *cost);

/* ---- size of FrameCoder::EncodeMonoFrame_Normal's payload (src/codec.cpp:237-247): BitplaneCoder::Encode (src/coder/vle.cpp:110-138)
 * over RangeCoderSH (src/coder/range.cpp:52-90) -- SURVEY.md 8f.4 ------------------------------------------------------------
 * sizes[b] = the number of bytes the host's bit-plane coder writes for the s2u residuals s2u[b][0..n[b]) coded with maxbpn[b]
 * planes + 1 (framestats[ch].maxbpn, src/codec.cpp:207-208), exactly. Every buffer is a coder of its own and all of a call's
 * coders run at once (a workgroup each, up to 8192; a lane each beyond), so the call pays off for many buffers (candidates x
 * channels x frames); the bytes themselves: sacopt_encode_payloads below. s2u: host array of nbuf
 * host-or-device pointers; n, maxbpn, sizes: host arrays. The coder's logistic tables (src/model/domain.h:12-23) are
 * tabulated with the libm of the host that created the context, as the reference's static initialiser does.
 * maxbpn[b] in 0..30. A value may have bits above plane maxbpn[b] (s2u of a residual beyond 2^30 wraps negative and the
 * reference's signed maximum skips it): as in the reference, planes maxbpn[b]..0 of its two's-complement bits are coded. */
int sacopt_coded_sizes(sacopt_ctx *ctx, int nbuf, const int32_t *const *s2u, const int32_t *n, const int32_t *maxbpn, int32_t *sizes);

/* ---- the payload BYTES: FrameCoder::EncodeMonoFrame_Normal (src/codec.cpp:237-247) and EncodeMonoFrame_Mapped (:249-265) ----
 * What the host's coder writes for one channel of one frame, on the device: BitplaneCoder::Encode (src/coder/vle.cpp:110-138)
 * over RangeCoderSH (src/coder/range.cpp:52-90), for a mapped channel behind MapEncoder::Encode of its Remap tables
 * (src/coder/map.cpp:3-72) through the same range coder. Buffer b: s2u[b][0..n[b]) with maxbpn[b] (0..30) as for
 * sacopt_coded_sizes; used == NULL or used[b] == NULL: the plain variant (feed s2u_error, maxbpn); else used[b] = usedl[0..32768]
 * then usedh[0..32768], one byte each (Remap's tables, as sacopt_dec_frame takes them): the mapped variant (feed
 * s2u_error_map, maxbpn_map). out[b]: cap[b] bytes (host or device); sizes[b] = the payload's length -- bytes beyond cap[b] are
 * dropped, so sizes[b] > cap[b] says the buffer was too small (n[b] * 4 + 8 KiB + 64 always fits the plain variant's planes and the
 * map). FrameCoder::EncodeMonoFrame keeps the mapped payload unless the plain one is strictly smaller (codec.cpp:268-280): call with
 * both variants of a channel and compare sizes. At most 8192 buffers per call; a frame's buffers are independent coders, so
 * the call is as long as its longest buffer (a 176 400-sample channel: ~0.9 s) whatever their number, up to two per CU. */
int sacopt_encode_payloads(sacopt_ctx *ctx, int nbuf, const int32_t *const *s2u, const int32_t *n, const int32_t *maxbpn,
                           const uint8_t *const *used, uint8_t *const *out, const int32_t *cap, int32_t *sizes);

/* ---- CostGolomb::Calc without any allocation ------------------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE. Same arguments, result and state handling as sacopt_cost_golomb, for callers that
 * cannot allocate: callable while allocation is forbidden -- for example while ANOTHER stream of the process is being
 * captured into a HIP graph, in tight serving loops, under harnesses that forbid hipMalloc. The call itself is blocking
 * (it synchronises the library's private stream and copies its two result doubles to the caller) and is NOT capturable.
 * The staging copy of a host `buf` and the kernels' partial results (the buffer is costed in parallel segments whose
 * guessed start states are verified, see kernels.hip) live in a file-scope device array of 4 MiB + 1.6 KiB that every GPU
 * the library runs on holds until the process ends; the call makes no hipMalloc / hipFree / hipHostMalloc, creates no
 * event and touches neither the context's arena nor its staging memory. Limits: n <= SACOPT_NOALLOC_MAX_SAMPLES (larger n
 * is an error, as is a NULL ctx); all calls on one GPU -- from any context -- share that GPU's array: a call that finds it
 * in use by another context or thread returns an error ("already running on this GPU") instead of racing. */
#define SACOPT_NOALLOC_MAX_SAMPLES (1 << 20)
int sacopt_cost_golomb_noalloc(sacopt_ctx *ctx, const int32_t *buf, int n, double *state_inout, double *cost);

/* ---- stage-level entry points (used by the parity tests to pin single kernels) --------------- */
/* Stereo pair of LMSADA2(S,D,mu1,mu2) (src/pred/nlms.h:96-139) driven as the cascades drive them.
 * Supported (S,D): (1280,640) (256,128) (32,16) (8,4). x/p are host or device doubles. */
int sacopt_lms_pair(sacopt_ctx *ctx, int S, int D, double mu1, double mu2, const double *x0,
                    const double *x1, int n, double *p0, double *p1);
/* Stereo pair of LPC3(alpha,S,D,8,500) (src/pred/lpc.h:172-228). Built (S,D): (16,8) as shipped, (24,8). */
int sacopt_lpc3_pair(sacopt_ctx *ctx, double alpha, int S, int D, const int32_t *s0, const int32_t *s1, int n,
                     double *p0, double *p1);
/* RLS(nst,alpha) mixer (src/pred/rls.cpp:61-108): p[t] = Predict(pv[t][0..nst)); Update(val[t]). Runs on the cascades' own
 * mixer kernel (k_mix), chunked with carried state; pv / val / p are host arrays, pv row-major [len][nst]. */
int sacopt_rls_run(sacopt_ctx *ctx, int nst, double alpha, const double *pv, const double *val,
                   int len, double *p);
/* Cholesky::Factor + Solve (src/math/cholesky.cpp:16-77) of one covariance snapshot through the LPC stages' kernel:
 * snap = packed lower triangle of the n x n matrix (row by row, n(n+1)/2 doubles) followed by b[n]. fast = 0: LPC3's form
 * (regp 500 added to the diagonal, solutions for every leading order: sol[o(o+1)/2 + i], n = 24 or 32); fast = 1: LPC2's
 * form (regp 0, one full-order solve: sol[0..n), n = 24). *fail = 1: "matrix indefinit" (lpc.h:216-218), sol untouched. */
int sacopt_chol_solve(sacopt_ctx *ctx, int n, int fast, const double *snap, double *sol, int *fail);
/* out[v] = ((0. + x[v][0]) + x[v][1]) + ... : the strict left-to-right FP64 sum of src/pred/nlms.h:110-111 on the device
 * helpers the LMS kernels use (matrix pipe as an exact adder; nterms 1920: speculative exact summation; 384 / 48 / 12: one
 * MFMA chain). x: count vectors of nterms doubles, host or device. */
int sacopt_serial_sum(sacopt_ctx *ctx, int nterms, const double *x, int count, double *out);
/* Stage outputs of one cascade run over a whole frame (normal / high profiles; coefs NULL = base profile), as the stage
 * kernels hand them on: p_out[s][ch][t] = stage s prediction (s = 0: LPC3, then the LMS stages; the reference's p[s],
 * src/profile.cpp:128-132), v_out[s][ch][t] = val[ch][s+1](t), the next stage's input (profile.cpp:141-148). Host arrays
 * of (1 + number of LMS stages) x 2 x numsamples doubles each. */
int sacopt_cascade_trace(sacopt_ctx *ctx, int profile, const sacopt_frame *frame, const double *coefs, double *p_out, double *v_out);

/* Timing of the last call, measured with HIP events on the library's own streams: total device
 * ms, and for the dominant kernel (high profile: k_lms1c, the per-channel LMSADA2(1280,640)
 * kernel, in calls of up to 20 frames, the two-channel kernel of the same stage in larger ones; other profiles: their
 * first LMS stage): summed launch durations in ms, launch count
 * and stereo samples processed by those launches. No counterpart in the reference (measurement
 * only). */
int sacopt_last_timing(const sacopt_ctx *ctx, double *total_ms, double *lms_ms, int *lms_launches,
                       long long *lms_units);

#ifdef __cplusplus
}
#endif
#endif
