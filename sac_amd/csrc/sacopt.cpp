// libsacopt host side: C ABI (include/sacopt.h) + stage scheduling on one HIP stream.
//
// Scheduling model: the cascade is feed-forward between stages, so a candidate
// evaluation is a chain of stage kernels (LPC3 -> LMS.. -> mixer) and a stage's output depends
// only on the coefficients of the stages up to it. Stage outputs are memoised per frame on that
// coefficient prefix: the coordinate search of FrameCoder::Optimize (src/codec.cpp:354-408)
// changes one coefficient per round, so every round re-runs only the stages downstream of it
// (bit-identical to the reference, which recomputes everything: src/codec.cpp:341-352).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <unordered_map>
#include <functional>
#include <string>
#include <vector>

#include "../../include/sacopt.h"
#include "jobs.h"

extern "C" {
int sacopt_lms_state_doubles(int S, int D);
int sacopt_launch_lms(int S, int D, const LmsJob *jobs, int njobs, const uint16_t *tab, int bits, hipStream_t st, int limit, int ct0, int clen);
int sacopt_lms_variant(int S, int D, int njobs, int bits, int limit);
// (1280,640) stage: k_lms1c gives a chain two CUs (4.85k cycles per sample against ~8.6k for both channels on one CU). Full-frame
// pass: taken for up to 48 frames per call -- measured on the encode call, same run, k_lms1c against k_lms2: 24 frames 592 / 665 ms,
// 30: 676 / 705, 40: 805 / 840, 50: 966 / 972, 60: 1105 / 1076, 75: 1334 / 1284 (beyond ~50 frames the pass shares the chip with the
// optimizer's rounds and the call is throughput-bound: the two-channel kernel is cheaper per CU). Optimizer windows: up to 20 chains
// per launch (up to 96 chains on k_lms1c changed nothing for encode calls of 2..15 frames: the full-frame pass bounds them).
#ifndef LEAD_SPLIT_MIN
#define LEAD_SPLIT_MIN 256  // 34 / 40 / 50 frames (340-550 chains per round), same run: 738 -> 653, 793 -> 733, 925 -> 821 ms
#endif
#ifndef LEAD_SPLIT_DEFSET
#define LEAD_SPLIT_DEFSET 1
#endif
#ifndef LEAD_SPLIT_WAYS
#define LEAD_SPLIT_WAYS 3  // 75 frames, same run: 1 sequence 1212 ms, 2: 1150, 3: 1132, 4: 1137
#endif
#ifndef LEAD_SPLIT_ANY
#define LEAD_SPLIT_ANY 0  // 1: also rounds led by another LMS stage
#endif
#ifndef ROUND_CHUNKS_BIG
#define ROUND_CHUNKS_BIG 16  // chunks per optimizer window when a round's lead launch has more workgroups than the chip has CUs
#endif
#ifndef L1C_FRAMES
#define L1C_FRAMES 48
#endif
#ifndef L1C_WINDOW
#define L1C_WINDOW 20
#endif
int sacopt_launch_lpc3(const LpcJob *jobs, int njobs, int max_n, int ols_n, hipStream_t st);
int sacopt_launch_lpc2(const LpcJob *jobs, int njobs, int max_n, hipStream_t st);
int sacopt_launch_mix(int ns, const MixJob *jobs, int njobs, hipStream_t st, int ct0, int clen);
int sacopt_launch_cost(const CostJob *jobs, int njobs, const int *pow2_bits, hipStream_t st);
int sacopt_launch_cost_par(const CostJob *jobs, int njobs, int nbuf, const int *pow2_bits, hipStream_t st);
int sacopt_launch_meanrms(const int32_t *e, int n, double *out, hipStream_t st);
int sacopt_launch_chol(const LpcJob *job, int n, int fast, hipStream_t st);
int sacopt_launch_sumkat(int nterms, const double *x, int count, double *out, hipStream_t st);
int sacopt_launch_map(const MapJob *jobs, int njobs, int max_numsamples, int max_n, hipStream_t st);
int sacopt_launch_unpredict(int type, const UnJob *jobs, int njobs, const uint16_t *tab, int bits, hipStream_t st);
struct BpJob { const int32_t *buf; int32_t n, maxbpn; };
struct RcState { unsigned long long lowc; uint32_t range, ffnum, cache, pos; };
struct BpEmit { uint8_t *out; const uint8_t *used; RcState *rc; int32_t cap, pad; };
static_assert(sizeof(RcState) == 24 && sizeof(BpEmit) == 32 && sizeof(BpJob) == 16, "layouts shared with bitplane.h");
extern "C" int sacopt_launch_payloads(const BpJob *jobs, const BpEmit *emit, int nbuf, const int32_t *which, int nmapped,
                                      const int16_t *fwd, const uint16_t *inv, const uint16_t *sse_init, int tscale, int xscale, int32_t *sizes,
                                      hipStream_t st);
int sacopt_launch_bpsize(const BpJob *jobs, int nbuf, int maxn, int maxplanes, int32_t *T, uint8_t *msb, uint32_t *sse, uint8_t *lb,
                         const int16_t *fwd, const uint16_t *inv, const uint16_t *sse_init, int tscale, int xscale, int32_t *sizes, hipStream_t st);
int sacopt_launch_meanrms_batch(const int32_t *const *bufs, const int32_t *ns, int nbuf, double *out, hipStream_t st);
int sacopt_launch_gather_i32(const int32_t *const *ptrs, int n, int32_t *out, hipStream_t st);
int sacopt_noalloc_scratch(int device, void **ptr, size_t *stage_bytes, size_t *ctl_doubles, int *max_n);
int sacopt_launch_cost_noalloc(const int32_t *e, int n, double *ctl, const int *pow2_bits, hipStream_t st);
}

#ifdef SACOPT_HOSTPROF  // timing builds: host time per section to stderr (tools/ only)
#include <chrono>
static double hp_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HP_MARK(var) const double var = hp_now()
#define HP_PRINT(...) fprintf(stderr, __VA_ARGS__)
#else
#define HP_MARK(var)
#define HP_PRINT(...)
#endif
static thread_local std::string g_create_error;  // error text of a failed sacopt_create on THIS thread
// sacopt_cost_golomb_noalloc: every context on a GPU shares that GPU's static scratch. A call that finds the scratch in use
// (another context / thread on the same GPU) returns an error instead of racing on it.
static std::atomic_flag g_na_busy[64] = {};

struct sacopt_ctx {
  int device = 0;
  hipStream_t st = nullptr;
  hipStream_t sA = nullptr, sB = nullptr;  // final-pass pipeline: dominant LMS stage / everything else
  hipStream_t sC = nullptr;                // follower stages of the optimizer rounds
  hipStream_t sE = nullptr;                // optimizer followers behind the first one (may be NULL: all on sC)
  hipStream_t split[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // alternative {st, sB, sC, sD, sE} (see sacopt_create)
  hipStream_t big[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};    // {st, sB, sC, sD, sE} on every CU: calls with more chains than CUs
  hipStream_t up = nullptr;                // stream of upload() (NULL: st); the full-frame pipeline stages its job tables on sB
  hipStream_t sAall = nullptr;             // lead stage of the full-frame pass when it has more workgroups than sA's reserved CUs
  hipStream_t sD = nullptr;                // final-pass pipeline: stages behind the second LMS stage (may be NULL)
  hipStream_t sL2[2] = {nullptr, nullptr};  // further lead streams of the optimizer's rounds in large batches (every CU; see eval_chains)
  // (Every CU-masked stream is a hardware queue of its own, and the chip has a fixed number of them: with these 19 everything
  // measured here is fine (tools/queue_cliff.sh: a process may hold 22), with 23 -- the same lead sequences for the default stream set -- EVERY call was 1.5x slower,
  // 15 frames 399 -> 600 ms: the queues are time-sliced once they outnumber the hardware's. Do not add streams; reuse.)
  // Events are created once and reused from call to call (a 15-frame encode call records several hundred): two pools, with
  // and without timing, handed out in order and taken back as a whole by begin_call once every stream is idle.
  std::vector<hipEvent_t> ev_timed, ev_plain;
  size_t ev_timed_used = 0, ev_plain_used = 0;
  uint16_t *d_tab = nullptr;  // rsqrtss table, RSQ_PACK'ed
  int bits = 0;
  int *d_pow2 = nullptr;
  std::string err;
  // bump arena (device)
  char *arena = nullptr;
  size_t cap = 0, used = 0;
  // pinned host staging for job tables and other small uploads: hipMemcpyAsync reads it after the enqueueing function has
  // returned, so it lives in the context until the next call has synchronised every stream (begin_call)
  struct PinBlock { char *p; size_t cap, used; };
  std::vector<PinBlock> pins;
  // timing of the last call
  std::vector<std::pair<hipEvent_t, hipEvent_t>> lms_ev;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double total_ms = 0, lms_ms = 0;
  int lms_launches = 0;
  long long lms_units = 0;  // stereo samples processed by the timed launches
  int timed_variant = 1;    // which (1280,640) kernel's launches are timed: 1 = k_lms1c (calls of <= 20 frames), 0 = the two-channel kernel
  // LogDomain tables of the host's bit-plane coder (src/model/domain.h:6-57), tabulated with THIS host's libm at create
  int16_t *d_bp_fwd = nullptr;
  uint16_t *d_bp_inv = nullptr, *d_bp_sse = nullptr;
  int bp_tscale = 0, bp_xscale = 0;
  // residual buffers of the last sacopt_get_costs call (still in the arena), for sacopt_recost_last
  struct CostBufs {
    std::vector<const int32_t *> e;     // error[ch]     [frame][cand][channel]
    std::vector<const int32_t *> u;     // s2u_error[ch] (same order)
    std::vector<const int32_t *> umax;  // address of max s2u_error[ch] (same order)
    std::vector<int> n;                 // window length per frame
  };
  struct Kept {
    bool valid = false;
    int nframes = 0, ncand = 0;
    CostBufs b;
    size_t mark = 0;                    // arena position behind the kept data
  } kept;
  int cost_fn = SACOPT_COST_GOLOMB;     // CostFunction the optimizer minimises (sacopt_set_cost_function)
  // sacopt_cost_golomb_noalloc: device address of this GPU's file-scope scratch array (kernels.hip), taken once at create
  unsigned char *na_scratch = nullptr;
  size_t na_stage_bytes = 0, na_ctl_doubles = 0;
  int na_max_n = 0;
};

static hipEvent_t take_event(sacopt_ctx *c, bool timed);
#define HIPCK(ctx, call)                                                                          \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                             \
      return 1;                                                                                   \
    }                                                                                             \
  } while (0)

// An event from the context's pool (nullptr: creation failed, c->err set). Valid until the next begin_call.
static hipEvent_t take_event(sacopt_ctx *c, bool timed) {
  std::vector<hipEvent_t> &pool = timed ? c->ev_timed : c->ev_plain;
  size_t &used = timed ? c->ev_timed_used : c->ev_plain_used;
  if (used == pool.size()) {
    hipEvent_t e = nullptr;
    const hipError_t rc = timed ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) { c->err = std::string("hipEventCreate: ") + hipGetErrorString(rc); return nullptr; }
    pool.push_back(e);
  }
  return pool[used++];
}
#define TAKE_EVENT(ctx, var, timed)                 \
  do {                                              \
    (var) = take_event((ctx), (timed));             \
    if (!(var)) return 1;                           \
  } while (0)

// ------------------------------------------------------------------------------------------ rsqrtss
static float rsqrt_native(float x) {  // src/pred/nlms.h:34-57
  float r;
  __asm__ __volatile__("movss %1, %%xmm0\n rsqrtss %%xmm0, %%xmm1\n movss %%xmm1, %0\n" : "=m"(r) : "m"(x) : "xmm0", "xmm1");
  return r;
}
static float rsqrt_from_table(const float *tab, int bits, float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint32_t e = (u >> 23) & 0xff, mant = u & 0x7fffff, par = e & 1;
  const uint32_t idx = (par << bits) | (bits ? (mant >> (23 - bits)) : 0);
  uint32_t tb;
  memcpy(&tb, &tab[idx], 4);
  const int q = ((int)e - (par ? 127 : 128)) / 2;
  tb = (uint32_t)((int32_t)tb - (q << 23));
  float r;
  memcpy(&r, &tb, 4);
  return r;
}
extern "C" int sacopt_probe_rsqrt(float *tab, int cap) {
  const uint32_t NM = 1u << 23;
  std::vector<uint32_t> out(2 * (size_t)NM);
  for (uint32_t par = 0; par < 2; par++) {
    const uint32_t e = par ? 127 : 128;
    for (uint32_t m = 0; m < NM; m++) {
      const uint32_t u = (e << 23) | m;
      float x, r;
      memcpy(&x, &u, 4);
      r = rsqrt_native(x);
      memcpy(&out[par * NM + m], &r, 4);
    }
  }
  int bits = -1;
  for (int k = 0; k <= 23 && bits < 0; k++) {
    const uint32_t mask = ~((1u << (23 - k)) - 1) & (NM - 1);
    bool ok = true;
    for (uint32_t par = 0; par < 2 && ok; par++)
      for (uint32_t m = 0; m < NM; m++)
        if (out[par * NM + m] != out[par * NM + (m & mask)]) { ok = false; break; }
    if (ok) bits = k;
  }
  if (bits < 0 || (2 << bits) > cap) return -1;
  for (uint32_t par = 0; par < 2; par++)
    for (uint32_t i = 0; i < (1u << bits); i++) memcpy(&tab[(par << bits) | i], &out[par * NM + (i << (23 - bits))], 4);
  for (uint32_t e = 1; e < 255; e++)  // exponent-scale invariance
    for (uint32_t i = 0; i < (1u << bits); i++)
      for (int f = 0; f < 3; f++) {
        const uint32_t low = (1u << (23 - bits)) - 1;
        const uint32_t m = (i << (23 - bits)) | (f == 0 ? 0 : f == 1 ? low : (low >> 1));
        const uint32_t u = (e << 23) | m;
        float x, a, b;
        memcpy(&x, &u, 4);
        a = rsqrt_native(x);
        b = rsqrt_from_table(tab, bits, x);
        if (memcmp(&a, &b, 4)) return -1;
      }
  return bits;
}

// ------------------------------------------------------------------------------------------ profiles
struct StageDesc { int S, D, c1, c2; };
struct ProfDesc {
  int type, ncoef, nlms;
  int ols_n;       // taps of the OLS stage (S + D): 24 for the reference's profiles, 32 for SACOPT_PROFILE_HIGH32
  StageDesc lms[4];
  int cmix;        // coefficient index of the RLS alpha
  int nprefix[6];  // coefs that determine the output of stage s (0 = LPC, 1..nlms = LMS, nlms+1 = mixer)
  int stage_of_coef[8];
};
static bool get_prof(int type, ProfDesc *p) {
  memset(p, 0, sizeof(*p));
  p->type = type;
  p->ols_n = (type == SACOPT_PROFILE_HIGH32) ? 32 : 24;
  if (type == SACOPT_PROFILE_HIGH || type == SACOPT_PROFILE_HIGH32) {  // StereoHigh ctor, src/profile.cpp:113-124 (HIGH32: LPC3(alpha, 24, 8, 8, 500))
    p->ncoef = 8; p->nlms = 4;
    p->lms[0] = {1280, 640, 1, 2}; p->lms[1] = {256, 128, 3, 4}; p->lms[2] = {32, 16, 5, 5}; p->lms[3] = {8, 4, 6, 6};
    p->cmix = 7;
    const int np[6] = {1, 3, 5, 6, 7, 8};
    memcpy(p->nprefix, np, sizeof(np));
    const int sc[8] = {0, 1, 1, 2, 2, 3, 4, 5};
    memcpy(p->stage_of_coef, sc, sizeof(sc));
  } else if (type == SACOPT_PROFILE_NORMAL) {  // StereoNormal ctor, src/profile.cpp:46-55
    p->ncoef = 6; p->nlms = 3;
    p->lms[0] = {256, 128, 1, 2}; p->lms[1] = {32, 16, 3, 3}; p->lms[2] = {8, 4, 4, 4};
    p->cmix = 5;
    const int np[6] = {1, 3, 4, 5, 6, 0};
    memcpy(p->nprefix, np, sizeof(np));
    const int sc[8] = {0, 1, 1, 2, 3, 4, 0, 0};
    memcpy(p->stage_of_coef, sc, sizeof(sc));
  } else if (type == SACOPT_PROFILE_FAST) {  // StereoFast ctor, src/profile.cpp:3-8
    p->ncoef = 1; p->nlms = 0; p->cmix = -1;
    p->nprefix[0] = 1;
  } else return false;
  return true;
}

extern "C" int sacopt_base_profile(int profile, double *out) {  // src/codec.cpp:8-30
  static const double fast[1][3] = {{0.99, 0.999, 0.996}};
  static const double normal[6][3] = {{0.99, 0.9999, 0.998}, {0.000001, 0.001, 0.0004}, {0.000001, 0.001, 0.0001},
                                      {0.00001, 0.01, 0.003}, {0.0001, 0.1, 0.01},       {0.99, 0.9999, 0.998}};
  static const double high[8][3] = {{0.98, 0.9999, 0.997},     {0.0000001, 0.0001, 0.00001}, {0.0000001, 0.0001, 0.000005},
                                    {0.000001, 0.001, 0.0005}, {0.000001, 0.001, 0.0001},    {0.0001, 0.01, 0.003},
                                    {0.001, 0.1, 0.01},        {0.99, 0.9999, 0.998}};
  const double *src;
  int n;
  if (profile == 0) { src = &fast[0][0]; n = 1; }
  else if (profile == 1) { src = &normal[0][0]; n = 6; }
  else if (profile == 2 || profile == SACOPT_PROFILE_HIGH32) { src = &high[0][0]; n = 8; }
  else return 0;
  for (int i = 0; i < 3 * n; i++) out[i] = src[i];
  return n;
}

// ------------------------------------------------------------------------------------------ context
extern "C" const char *sacopt_last_error(const sacopt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int sacopt_create(int device, const float *rsqrt_tab, int rsqrt_bits, sacopt_ctx **out) {
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (libsacopt has no CPU fallback)"; return 1; }
  if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return 1; }
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return 1; }
  if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) { g_create_error = std::string("device is ") + pr.gcnArchName + ", libsacopt is built for gfx950 only"; return 1; }
  std::vector<float> tab;
  int bits = rsqrt_bits;
  if (!rsqrt_tab) {
    tab.resize(RSQ_LDS_MAX);
    bits = sacopt_probe_rsqrt(tab.data(), RSQ_LDS_MAX);
    if (bits < 0) { g_create_error = "host rsqrtss does not fit the parity x top-k-mantissa table model (k <= 12)"; return 1; }
  } else {
    if (bits < 0 || (2 << bits) > RSQ_LDS_MAX) { g_create_error = "rsqrt_bits out of range (0..12)"; return 1; }
    tab.assign(rsqrt_tab, rsqrt_tab + (2 << bits));
  }
  sacopt_ctx *c = new sacopt_ctx;
  c->device = device;
  c->bits = bits;
#ifdef SACOPT_ABLATE  // tools/ builds only: kernel-variant switches for A/B timing (kernels.hip). The product library reads no such variable.
  if (const char *dbg = getenv("SACOPT_DBG_ALL")) c->bits |= atoi(dbg) << 8;
#endif
  // sacopt_destroy copes with a half-built context (every member starts null): streams, tables and events created before
  // the failing step are released with it
  auto fail = [&](const char *what) { g_create_error = what; sacopt_destroy(c); return 1; };
  if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
  // Streams and hardware queues. HIP multiplexes same-priority streams onto a small pool of in-order hardware
  // queues (3 usable here: a 4th normal-priority stream shared a queue with the pre-enqueued full-frame
  // pipeline and serialised behind it; priorities have their own pools, but a second high-priority stream
  // shares the first one's queue and a low-priority stream never got scheduled). A stream created with an
  // explicit CU mask gets a queue of its own, and the mask is useful in itself: the lead stage of the
  // full-frame pass -- 2 workgroups per frame, each a CU-filling serial chain that bounds the whole call --
  // gets 32 CUs to itself, everything else runs on the others (sharing CUs with the follower stages cost
  // that kernel 12 %). SACOPT_CUMASK=0 falls back to unmasked streams (lead/follower split by priority).
  bool masked = false;
  {
    hipDeviceProp_t prop;
    const char *env = getenv("SACOPT_CUMASK");
    if (!(env && atoi(env) == 0) && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount >= 128) {
      const int words = (prop.multiProcessorCount + 31) / 32;
      std::vector<uint32_t> ma(words, 0u), mr(words, 0xffffffffu);
      ma[0] = 0xffffffffu; mr[0] = 0u;
      if (prop.multiProcessorCount % 32) mr[words - 1] = (1u << (prop.multiProcessorCount % 32)) - 1u;
      masked = hipExtStreamCreateWithCUMask(&c->sA, words, ma.data()) == hipSuccess &&
               hipExtStreamCreateWithCUMask(&c->st, words, mr.data()) == hipSuccess &&
               hipExtStreamCreateWithCUMask(&c->sB, words, mr.data()) == hipSuccess &&
               hipExtStreamCreateWithCUMask(&c->sC, words, mr.data()) == hipSuccess &&
               hipExtStreamCreateWithCUMask(&c->sD, words, mr.data()) == hipSuccess &&
               hipExtStreamCreateWithCUMask(&c->sE, words, mr.data()) == hipSuccess;
      // Second set for small batches of sacopt_encode_frames (split[]: st, sB, sC, sD, sE): the full-frame pass's follower
      // streams on 32 CUs of their own, the optimizer on the remaining 192. An optimizer round's lead launch is as slow
      // as its slowest workgroup, and the workgroups that shared a CU with a full-frame follower (busy all the time) were:
      // encode 554 -> 537 ms for 15 frames. With more lead workgroups than 192 CUs (or a chip that is full anyway) the
      // shared set is better (75 frames: 6.6 against 6.1 M samples/s).
      // (Follower set of 16 or 48 CUs instead of 32: 486 / 541 against 428 ms for 15 frames -- masks that are not whole 32-bit words
      // cut across the XCDs unevenly.)
      if (masked && prop.multiProcessorCount >= 256) {
        std::vector<uint32_t> mo(mr), mf(mr);
        for (int cu = 32; cu < prop.multiProcessorCount; cu++) {
          if (cu < 64) mo[cu >> 5] &= ~(1u << (cu & 31));
          else mf[cu >> 5] &= ~(1u << (cu & 31));
        }
        const uint32_t *mk[5] = {mo.data(), mf.data(), mo.data(), mf.data(), mo.data()};
        bool ok = true;
        for (int i = 0; i < 5 && ok; i++) ok = hipExtStreamCreateWithCUMask(&c->split[i], words, mk[i]) == hipSuccess;
        if (!ok) {
          (void)hipGetLastError();
          for (hipStream_t &p : c->split) if (p) { (void)hipStreamDestroy(p); p = nullptr; }
        }
      }
      if (masked) {
        std::vector<uint32_t> mall(mr);
        mall[0] = 0xffffffffu;
        masked = hipExtStreamCreateWithCUMask(&c->sAall, words, mall.data()) == hipSuccess;
        // Third set for large batches (big[]: st, sB, sC, sD, sE on EVERY CU): with more lead workgroups than CUs nothing is
        // latency-bound any more, the full-frame pass's lead stage runs on sAall, and 32 CUs kept free for it would idle once it
        // has finished (75 frames: 12 % of the chip from the optimizer's third round on)
#ifdef SACOPT_ABLATE
        const bool no_big = getenv("SACOPT_NOBIG") != nullptr;  // timing: seven hardware queues fewer
#else
        const bool no_big = false;
#endif
        if (masked && !no_big) {
          bool ok = true;
          // (Measured and not adopted: the optimizer's follower streams on 1 / 2 / 3 XCDs of their own, its lead stream on the others, so
          // that two lead workgroups share every CU there -- 75 frames 2050 / 1480 / 1445 against 1280 ms: the lead launches do get
          // 1.46x as much out of a CU, but the followers need more than a quarter of the chip.)
          // (big[3] on big[1]'s stream -- one hardware queue less, see sL2 -- was measured: 40 frames 733 -> 851 ms, 75 frames 1127 -> 1194)
          for (int i = 0; i < 5 && ok; i++) ok = hipExtStreamCreateWithCUMask(&c->big[i], words, mall.data()) == hipSuccess;
          for (int i = 0; i < 2 && ok; i++)
            if (hipExtStreamCreateWithCUMask(&c->sL2[i], words, mall.data()) != hipSuccess) { (void)hipGetLastError(); c->sL2[i] = nullptr; break; }
          if (!ok) {
            (void)hipGetLastError();
            for (hipStream_t &p : c->big) if (p) { (void)hipStreamDestroy(p); p = nullptr; }
          }
        }
      }
      if (!masked) {
        (void)hipGetLastError();
        hipStream_t *all[12] = {&c->sA, &c->st, &c->sB, &c->sC, &c->sD, &c->sAall, &c->sE, &c->big[0], &c->big[1], &c->big[2], &c->big[3], &c->big[4]};
        for (hipStream_t *p : all) if (*p) { (void)hipStreamDestroy(*p); *p = nullptr; }
      }
    }
  }
  if (!masked) {
    if (hipStreamCreate(&c->st) != hipSuccess || hipStreamCreate(&c->sA) != hipSuccess || hipStreamCreate(&c->sB) != hipSuccess)
      return fail("hipStreamCreate failed");
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&c->sC, hipStreamDefault, hi) != hipSuccess) return fail("hipStreamCreateWithPriority failed");
    c->sD = nullptr;  // full-frame followers all on sB
  }
  std::vector<uint16_t> packed(2 << bits);
  for (int i = 0; i < (2 << bits); i++) {
    uint32_t u;
    memcpy(&u, &tab[i], 4);
    if (!RSQ_PACKABLE(u)) return fail("rsqrtss table entry outside [0.5,2) or with more than 16 significant bits: not an rsqrtss table");
    packed[i] = RSQ_PACK(u);
  }
  if (hipMalloc(&c->d_tab, sizeof(uint16_t) * (2 << bits)) != hipSuccess) return fail("hipMalloc(table) failed");
  if (hipMemcpy(c->d_tab, packed.data(), sizeof(uint16_t) * (2 << bits), hipMemcpyHostToDevice) != hipSuccess) return fail("table upload failed");
  int pw[32];
  for (int k = 0; k < 31; k++) pw[k] = (int)ceil(log((double)(1 << k)) / log(2));  // codec.cpp:329 with THIS host's libm
  pw[31] = 31;
  if (hipMalloc(&c->d_pow2, sizeof(pw)) != hipSuccess) return fail("hipMalloc failed");
  if (hipMemcpy(c->d_pow2, pw, sizeof(pw), hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
  {  // stretch / squash tables of the bit-plane coder's logistic mixing (domain.h:12-23) and the SSE stage's initial map (sse.h:94-105)
    std::vector<int16_t> fwd(1 << 15);
    std::vector<uint16_t> inv(1 << 12), sse(33);
    const int PS = 1 << 15;
    for (int i = 0; i < PS; i++) fwd[i] = (int16_t)floor(log((i + 0.5) / (PS - i - 0.5)) * 256. + 0.5);
    for (int i = -2048; i <= 2047; i++) inv[i + 2048] = (uint16_t)floor((double)PS / (1.0 + exp(-(double)i / 256.)));
    c->bp_tscale = fwd[PS - 1];
    c->bp_xscale = (2 * c->bp_tscale) / 31;
    if (c->bp_xscale == 0) c->bp_xscale = 1;
    for (int i = 0; i <= 32; i++) {
      const int x = i * c->bp_xscale - c->bp_tscale;
      sse[i] = x < -2048 ? 0 : (x > 2047 ? PS - 1 : inv[x + 2048]);
    }
    if (hipMalloc(&c->d_bp_fwd, fwd.size() * 2) != hipSuccess || hipMalloc(&c->d_bp_inv, inv.size() * 2) != hipSuccess ||
        hipMalloc(&c->d_bp_sse, 33 * 2) != hipSuccess) return fail("hipMalloc failed");
    if (hipMemcpy(c->d_bp_fwd, fwd.data(), fwd.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_bp_inv, inv.data(), inv.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_bp_sse, sse.data(), 33 * 2, hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return fail("hipEventCreate failed");
  {  // this device's copy of the static scratch (hipGetSymbolAddress with the device current; the array's name is no address)
    void *p = nullptr;
    if (sacopt_noalloc_scratch(device, &p, &c->na_stage_bytes, &c->na_ctl_doubles, &c->na_max_n)) return fail("hipGetSymbolAddress(scratch) failed");
    c->na_scratch = static_cast<unsigned char *>(p);
  }
  *out = c;
  return 0;
}

extern "C" void sacopt_destroy(sacopt_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (hipStream_t q : {c->st, c->sA, c->sAall, c->sB, c->sC, c->sD, c->sE, c->split[0], c->split[1], c->split[2], c->split[3], c->split[4],
                       c->big[0], c->big[1], c->big[2], c->big[3], c->big[4], c->sL2[0], c->sL2[1]})
    if (q) (void)hipStreamSynchronize(q);
  for (hipEvent_t e : c->ev_timed) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_plain) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->arena) (void)hipFree(c->arena);
  for (auto &b : c->pins) (void)hipHostFree(b.p);
  if (c->d_tab) (void)hipFree(c->d_tab);
  if (c->d_pow2) (void)hipFree(c->d_pow2);
  if (c->d_bp_fwd) (void)hipFree(c->d_bp_fwd);
  if (c->d_bp_inv) (void)hipFree(c->d_bp_inv);
  if (c->d_bp_sse) (void)hipFree(c->d_bp_sse);
  if (c->sA) (void)hipStreamDestroy(c->sA);
  if (c->sB) (void)hipStreamDestroy(c->sB);
  if (c->sC) (void)hipStreamDestroy(c->sC);
  if (c->sD) (void)hipStreamDestroy(c->sD);
  if (c->sAall) (void)hipStreamDestroy(c->sAall);
  if (c->sE) (void)hipStreamDestroy(c->sE);
  for (hipStream_t p : c->split) if (p) (void)hipStreamDestroy(p);
  for (hipStream_t p : c->big) if (p) (void)hipStreamDestroy(p);
  for (hipStream_t p : c->sL2) if (p) (void)hipStreamDestroy(p);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

extern "C" int sacopt_set_cost_function(sacopt_ctx *c, int which) {
  if (!c) return 1;
  if (which != SACOPT_COST_GOLOMB && which != SACOPT_COST_MEANRMS && which != SACOPT_COST_BITPLANE) { c->err = "unknown cost function"; return 1; }
  c->cost_fn = which;
  return 0;
}

extern "C" int sacopt_last_timing(const sacopt_ctx *c, double *total_ms, double *lms_ms, int *lms_launches,
                                  long long *lms_units) {
  if (!c) return 1;
  if (lms_units) *lms_units = c->lms_units;
  if (total_ms) *total_ms = c->total_ms;
  if (lms_ms) *lms_ms = c->lms_ms;
  if (lms_launches) *lms_launches = c->lms_launches;
  return 0;
}

// ------------------------------------------------------------------------------------------ helpers
static int arena_reserve(sacopt_ctx *c, size_t bytes) {
  c->used = 0;
  if (bytes <= c->cap) return 0;
  if (c->arena) { HIPCK(c, hipStreamSynchronize(c->st)); HIPCK(c, hipFree(c->arena)); c->arena = nullptr; c->cap = 0; }
  HIPCK(c, hipMalloc(&c->arena, bytes));
  c->cap = bytes;
  return 0;
}
template <class T>
static T *arena_alloc(sacopt_ctx *c, size_t count) {
  const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  if (c->used + bytes > c->cap) return nullptr;
  T *p = reinterpret_cast<T *>(c->arena + c->used);
  c->used += bytes;
  return p;
}
static bool is_device_ptr(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice;
}
// Pinned staging memory that stays valid until the next call's begin_call (see sacopt_ctx::pins).
static void *pin_alloc(sacopt_ctx *c, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (!c->pins.empty()) {
    auto &b = c->pins.back();
    if (b.used + bytes <= b.cap) { void *p = b.p + b.used; b.used += bytes; return p; }
  }
  size_t cap = c->pins.empty() ? ((size_t)4 << 20) : 2 * c->pins.back().cap;
  if (cap < bytes) cap = bytes;
  char *p = nullptr;
  if (hipHostMalloc((void **)&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  c->pins.push_back({p, cap, bytes});
  return p;
}
// host bytes -> device, asynchronously on `s`, through pinned staging (the source may go away as soon as this returns)
static int upload_bytes(sacopt_ctx *c, void *dst_dev, const void *src, size_t bytes, hipStream_t s) {
  if (!bytes) return 0;
  void *pin = pin_alloc(c, bytes);
  if (!pin) { c->err = "hipHostMalloc (staging) failed"; return 1; }
  memcpy(pin, src, bytes);
  HIPCK(c, hipMemcpyAsync(dst_dev, pin, bytes, hipMemcpyHostToDevice, s));
  return 0;
}
template <class T>
static int upload(sacopt_ctx *c, const std::vector<T> &h, T **d) {
  *d = arena_alloc<T>(c, h.size() ? h.size() : 1);
  if (!*d) { c->err = "device arena exhausted"; return 1; }
  return upload_bytes(c, *d, h.data(), h.size() * sizeof(T), c->up ? c->up : c->st);
}
template <class T>
static int to_dev(sacopt_ctx *c, const T *src, size_t count, const T **out) {
  if (is_device_ptr(src)) { *out = src; return 0; }
  T *p = arena_alloc<T>(c, count);
  if (!p) { c->err = "device arena exhausted"; return 1; }
  HIPCK(c, hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, c->st));
  *out = p;
  return 0;
}
static int begin_call(sacopt_ctx *c) {
  HIPCK(c, hipSetDevice(c->device));
  c->err.clear();
  // a call that failed half-way may have left work in flight on the pipeline streams; the arena is about to be reused
  for (hipStream_t s : {c->st, c->sA, c->sAall, c->sB, c->sC, c->sD, c->sE, c->split[0], c->split[1], c->split[2], c->split[3], c->split[4],
                       c->big[0], c->big[1], c->big[2], c->big[3], c->big[4], c->sL2[0], c->sL2[1]})
    if (s) HIPCK(c, hipStreamSynchronize(s));
  // nothing is in flight any more: the staging memory of the previous call can be reused (several blocks -> one of their size)
  if (c->pins.size() > 1) {
    size_t tot = 0;
    for (auto &b : c->pins) { tot += b.cap; (void)hipHostFree(b.p); }
    c->pins.clear();
    char *p = nullptr;
    if (hipHostMalloc((void **)&p, tot, hipHostMallocDefault) == hipSuccess) c->pins.push_back({p, tot, 0});
    else (void)hipGetLastError();
  } else if (!c->pins.empty()) c->pins[0].used = 0;
  c->lms_ev.clear();  // every stream is idle: all pooled events are free again
  c->ev_timed_used = c->ev_plain_used = 0;
  c->total_ms = c->lms_ms = 0;
  c->lms_launches = 0;
  c->lms_units = 0;
  c->kept.valid = false;  // the arena is about to be reused
  c->timed_variant = 1;
  return 0;
}
static int end_timing(sacopt_ctx *c) {
  HIPCK(c, hipEventRecord(c->ev1, c->st));
  HIPCK(c, hipStreamSynchronize(c->st));
  float ms = 0;
  HIPCK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->total_ms = ms;
  for (auto &p : c->lms_ev) {
    HIPCK(c, hipEventElapsedTime(&ms, p.first, p.second));
    c->lms_ms += ms;
  }
  c->lms_launches = (int)c->lms_ev.size();
  return 0;
}

// Swaps {st, sB, sC, sD, sE} for another stream set (sacopt_create: split[] / big[]) for the duration of a call.
struct StreamSetGuard {
  sacopt_ctx *c;
  hipStream_t keep[5];
  bool on;
  StreamSetGuard(sacopt_ctx *ctx, hipStream_t *set) : c(ctx), keep{ctx->st, ctx->sB, ctx->sC, ctx->sD, ctx->sE}, on(set && set[0]) {
    if (on) { c->st = set[0]; c->sB = set[1]; c->sC = set[2]; c->sD = set[3]; c->sE = set[4]; }
  }
  ~StreamSetGuard() { if (on) { c->st = keep[0]; c->sB = keep[1]; c->sC = keep[2]; c->sD = keep[3]; c->sE = keep[4]; } }
};
// Calls whose full-frame pass has more lead workgroups than the 32 CUs reserved for it run that stage on sAall anyway: from
// there on nothing is gained by keeping the other streams off those CUs (75 frames: 1350 -> 1291 ms; but 24 / 30 frames, whose
// full-frame lead stage still sits on the reserved CUs and bounds the call: 667 -> 722, 710 -> 787 ms with full masks).
// Window evaluations without a full-frame pass (Optimize, GetCost) take the full masks as soon as a round has more chains than CUs.
#ifndef BIG_FRAMES
#define BIG_FRAMES 32
#endif

// A frame window staged on the device
struct DevFrame {
  const int32_t *s[2];  // device, whole frame
  int numsamples, from, n;
};

// Outputs of one stage for one memo key
struct StageBuf {
  double *p[2] = {nullptr, nullptr};
  double *xn[2] = {nullptr, nullptr};
  int32_t *e[2] = {nullptr, nullptr};
  int32_t *u[2] = {nullptr, nullptr};
  int32_t *umax = nullptr;
};
struct Chain {
  int frame;
  double coefs[SACOPT_MAX_COEFS];
  StageBuf sb[6];
};
// Memo key: (frame, stage, the coefficient prefix that determines the stage's output, as bits: -0. and 0. are different keys,
// which only costs a recomputation). Hashed: the optimizer of a 75-frame call looks up ~5 000 keys per round in a table that
// grows to ~38 000 entries, and the host time between two rounds is GPU idle time.
struct MemoKey {
  int32_t frame, stage;
  uint64_t c[SACOPT_MAX_COEFS];
  bool operator==(const MemoKey &o) const { return frame == o.frame && stage == o.stage && memcmp(c, o.c, sizeof(c)) == 0; }
};
struct MemoHash {
  size_t operator()(const MemoKey &k) const {
    uint64_t h = 1469598103934665603ull ^ ((uint64_t)(uint32_t)k.frame << 32 | (uint32_t)k.stage);
    for (int i = 0; i < SACOPT_MAX_COEFS; i++) { h ^= k.c[i]; h *= 1099511628211ull; h ^= h >> 29; }
    return (size_t)h;
  }
};
typedef std::unordered_map<MemoKey, StageBuf, MemoHash> Memo;

static MemoKey memo_key(int frame, int stage, const double *coefs, int ncoef) {
  MemoKey k;
  k.frame = frame; k.stage = stage;
  for (int i = 0; i < SACOPT_MAX_COEFS; i++) k.c[i] = 0;
  memcpy(k.c, coefs, sizeof(double) * ncoef);
  return k;
}

// bytes of arena one chain needs for all its stages
static size_t chain_bytes(const ProfDesc &pd, int n) {
  const size_t nd = ((size_t)n * 8 + 255) & ~(size_t)255, ni = ((size_t)n * 4 + 255) & ~(size_t)255;
  return (size_t)(pd.nlms + 1) * 4 * nd + 4 * ni + 256 + 80 * 1024 + 4 * 5 * 512;  // + carried stage state + job tables
}

// LPC scratch (covariance snapshots, solutions, all-orders sums, running errors) for one chain
static size_t lpc_scratch_bytes(const ProfDesc &pd, int n) {
  const int type = pd.type, N = pd.ols_n, tri = N * (N + 1) / 2;
  const int ks = (type == SACOPT_PROFILE_FAST) ? 32 : 8;
  const size_t nsnap = (size_t)n / ks;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t per_ch = al(nsnap * (tri + N) * 8) + al(nsnap * (type == SACOPT_PROFILE_FAST ? N : tri) * 8) + al(nsnap * 4 + 4);
  if (type != SACOPT_PROFILE_FAST) per_ch += 2 * al((size_t)n * N * 8);
  return 2 * per_ch + 1024;
}
static int lpc_scratch_alloc(sacopt_ctx *c, const ProfDesc &pd, LpcJob &j) {
  const int type = pd.type, N = pd.ols_n, tri = N * (N + 1) / 2;
  const int ks = (type == SACOPT_PROFILE_FAST) ? 32 : 8;
  j.nsnap = j.n / ks;
  for (int k = 0; k < 2; k++) {
    j.snap[k] = arena_alloc<double>(c, (size_t)j.nsnap * (tri + N));
    j.sol[k] = arena_alloc<double>(c, (size_t)j.nsnap * (type == SACOPT_PROFILE_FAST ? N : tri));
    j.fail[k] = arena_alloc<int32_t>(c, (size_t)j.nsnap + 1);
    if (!j.snap[k] || !j.sol[k] || !j.fail[k]) { c->err = "device arena exhausted (LPC scratch)"; return 1; }
    if (type != SACOPT_PROFILE_FAST) {
      j.ss[k] = arena_alloc<double>(c, (size_t)j.n * N);
      j.serr[k] = arena_alloc<double>(c, (size_t)j.n * N);
      if (!j.ss[k] || !j.serr[k]) { c->err = "device arena exhausted (LPC scratch)"; return 1; }
    }
  }
  return 0;
}

// Chunk of the full-frame pipeline (final_pass_enqueue): short enough that the follower stages of the last
// chunk are a small tail, long enough that kernel prologues (table + history reload) stay negligible.
#define PIPE_CHUNK (16 * LMS_CHUNK) /* 8192 samples */
// job tables + carried state of the pipelined pass for one chain of n samples
static size_t pipe_bytes(int n) {
  const size_t nchunk = (size_t)n / PIPE_CHUNK + 20;  // + the short first chunk and the halving tail
  return nchunk * (4 * (sizeof(LmsJob) + 256) + sizeof(MixJob) + 256 + sizeof(LpcJob) + 256) + 16 * 1024 * 8 + 16384;
}

// Job tables collected on the host and sent up in ONE copy (a copy per table cost ~70 us each: 150 of them put 10 ms in
// front of the pipeline's first kernel).
struct Stager {
  std::vector<char> host;
  size_t add(const void *p, size_t bytes) {
    const size_t off = (host.size() + 255) & ~(size_t)255;
    host.resize(off + bytes);
    if (bytes) memcpy(host.data() + off, p, bytes);
    return off;
  }
};

// Run all chains through the cascade, re-using memoised stage outputs. `want_u`: also write s2u.
static int eval_chains(sacopt_ctx *c, const ProfDesc &pd, const std::vector<DevFrame> &fr, std::vector<Chain> &chains,
                       Memo &memo) {
  const int nstage = (pd.type == SACOPT_PROFILE_FAST) ? 1 : pd.nlms + 2;
  std::vector<std::vector<int>> todo_s(nstage);
  HP_MARK(hp0);
  for (int s = 0; s < nstage; s++) {
    std::vector<int> todo;  // chain indices that own a new buffer at this stage
    for (size_t ci = 0; ci < chains.size(); ci++) {
      Chain &ch = chains[ci];
      const MemoKey key = memo_key(ch.frame, s, ch.coefs, pd.nprefix[s]);
      auto it = memo.find(key);
      if (it != memo.end()) { ch.sb[s] = it->second; continue; }
      const int n = fr[ch.frame].n;
      StageBuf b;
      const bool last = (s == nstage - 1);
      if (!last) {
        for (int k = 0; k < 2; k++) {
          b.p[k] = arena_alloc<double>(c, n);
          b.xn[k] = arena_alloc<double>(c, n);
          if (!b.p[k] || !b.xn[k]) { c->err = "device arena exhausted"; return 1; }
        }
      } else {
        for (int k = 0; k < 2; k++) {
          b.e[k] = arena_alloc<int32_t>(c, n);
          b.u[k] = arena_alloc<int32_t>(c, n);
          if (!b.e[k] || !b.u[k]) { c->err = "device arena exhausted"; return 1; }
        }
        b.umax = arena_alloc<int32_t>(c, 2);
        if (!b.umax) { c->err = "device arena exhausted"; return 1; }
      }
      ch.sb[s] = b;
      memo[key] = b;
      todo.push_back((int)ci);
    }
    if (todo.empty()) continue;
    if (pd.type == SACOPT_PROFILE_FAST) {
      std::vector<LpcJob> jobs;
      for (int ci : todo) {
        const Chain &ch = chains[ci];
        const DevFrame &f = fr[ch.frame];
        LpcJob j;
        memset(&j, 0, sizeof(j));
        for (int k = 0; k < 2; k++) { j.s[k] = f.s[k] + f.from; j.e[k] = ch.sb[0].e[k]; j.u[k] = ch.sb[0].u[k]; }
        j.umax = ch.sb[0].umax;
        j.alpha = ch.coefs[0];
        j.n = f.n;
        jobs.push_back(j);
      }
      LpcJob *dj = arena_alloc<LpcJob>(c, jobs.size());
      if (!dj) { c->err = "device arena exhausted"; return 1; }
      // (the scratch is NOT handed back to the arena: memory reused "in stream order" is only safe for consumers on this
      // stream, and the full-frame pipeline, enqueued next on other streams, would have overwritten it -- a fault seen once)
      int max_n = 0;
      for (LpcJob &j : jobs) {
        if (lpc_scratch_alloc(c, pd, j)) return 1;
        max_n = j.n > max_n ? j.n : max_n;
        HIPCK(c, hipMemsetAsync(j.umax, 0, 8, c->st));
      }
      if (upload_bytes(c, dj, jobs.data(), jobs.size() * sizeof(LpcJob), c->st)) return 1;
      if (sacopt_launch_lpc2(dj, (int)jobs.size(), max_n, c->st)) { c->err = "LPC2 launch failed"; return 1; }
    } else if (s == 0) {
      std::vector<LpcJob> jobs;
      for (int ci : todo) {
        const Chain &ch = chains[ci];
        const DevFrame &f = fr[ch.frame];
        LpcJob j;
        memset(&j, 0, sizeof(j));
        for (int k = 0; k < 2; k++) { j.s[k] = f.s[k] + f.from; j.p[k] = ch.sb[0].p[k]; j.x1[k] = ch.sb[0].xn[k]; }
        j.alpha = ch.coefs[0];
        j.n = f.n;
        jobs.push_back(j);
      }
      LpcJob *dj = arena_alloc<LpcJob>(c, jobs.size());
      if (!dj) { c->err = "device arena exhausted"; return 1; }
      int max_n = 0;
      for (LpcJob &j : jobs) {
        if (lpc_scratch_alloc(c, pd, j)) return 1;
        max_n = j.n > max_n ? j.n : max_n;
      }
      if (upload_bytes(c, dj, jobs.data(), jobs.size() * sizeof(LpcJob), c->st)) return 1;
      if (sacopt_launch_lpc3(dj, (int)jobs.size(), max_n, pd.ols_n, c->st)) { c->err = "LPC3 launch failed"; return 1; }
    } else {
      todo_s[s] = todo;  // LMS stages and the mixer are launched below, pipelined over chunks
    }
  }
  if (pd.type == SACOPT_PROFILE_FAST) return 0;
  int lead = 0;  // first stage that has work: it runs on c->st, the stages behind it on c->sC
  for (int s = 1; s < nstage && !lead; s++)
    if (!todo_s[s].empty()) lead = s;
  if (!lead) return 0;
  int max_n = 0;
  for (int ci : todo_s[nstage - 1]) max_n = fr[chains[ci].frame].n > max_n ? fr[chains[ci].frame].n : max_n;
  // windows long enough are cut into ROUND_CHUNKS chunks so that the stages overlap (stage state is carried);
  // the round ends one follower-chunk after the lead stage's last chunk
  // (measured, one gpurun call: 8 / 16 / 24 chunks: 15 frames 448 / 433 / 432 ms, 75 frames 1670 / 1690 / 1735 ms; 3 chunks: 75 frames
  // 1660 ms but 30 frames 807 against 753 ms)
  const int ROUND_CHUNKS = (todo_s[lead].size() > 256) ? ROUND_CHUNKS_BIG : 16;
  const int chunk = (max_n >= 8 * LMS_CHUNK) ? ((max_n + ROUND_CHUNKS * LMS_CHUNK - 1) / (ROUND_CHUNKS * LMS_CHUNK)) * LMS_CHUNK : (max_n > 0 ? max_n : 1);
  const int nchunk = max_n > 0 ? (max_n + chunk - 1) / chunk : 1;
  const bool carry = nchunk > 1;
  std::vector<std::vector<double *>> state(nstage);
  if (carry)
    for (int s = lead; s < nstage; s++) {
      const size_t nd = (s <= pd.nlms) ? (size_t)sacopt_lms_state_doubles(pd.lms[s - 1].S, pd.lms[s - 1].D) : 2 * MIX_STATE;
      for (size_t q = 0; q < todo_s[s].size(); q++) {
        double *p = arena_alloc<double>(c, nd);
        if (!p) { c->err = "device arena exhausted"; return 1; }
        state[s].push_back(p);
      }
    }
  hipEvent_t ev_join = nullptr;
  // ---- ONE job table per stage, describing every chain's whole window; a chunk launch gets its span [t0, t0 + chunk) as kernel
  // arguments (jobs.h: LMS_SPAN) and clips it to each chain's window. (A table per stage AND chunk was 5.4 MB per round of a 75-frame
  // call: 3.5 ms of host time between two rounds -- GPU idle time -- to build and send it.)
  Stager stg;
  std::vector<size_t> o_jobs(nstage, 0);
  std::vector<int> n_jobs(nstage, 0);
  std::vector<long long> units_all(nchunk, 0);
  for (int s = lead; s < nstage; s++) {
    if (s <= pd.nlms) {
      const StageDesc &sd = pd.lms[s - 1];
      std::vector<LmsJob> jobs;
      for (size_t q = 0; q < todo_s[s].size(); q++) {
        const Chain &ch = chains[todo_s[s][q]];
        const int n = fr[ch.frame].n;
        LmsJob j;
        for (int h = 0; h < 2; h++) { j.x[h] = ch.sb[s - 1].xn[h]; j.p[h] = ch.sb[s].p[h]; j.xn[h] = ch.sb[s].xn[h]; }
        j.mu1 = ch.coefs[sd.c1];
        j.mu2 = ch.coefs[sd.c2];
        j.t0 = 0;
        j.n = n;
        j.state = carry ? state[s][q] : nullptr;
        jobs.push_back(j);
        if (s == 1)
          for (int k = 0; k < nchunk; k++) { const int r = n - k * chunk; units_all[k] += r < 0 ? 0 : (r < chunk ? r : chunk); }
      }
      n_jobs[s] = (int)jobs.size();
      o_jobs[s] = stg.add(jobs.data(), jobs.size() * sizeof(LmsJob));
    } else {
      std::vector<MixJob> jobs;
      for (size_t q = 0; q < todo_s[s].size(); q++) {
        const Chain &ch = chains[todo_s[s][q]];
        const DevFrame &f = fr[ch.frame];
        MixJob j;
        memset(&j, 0, sizeof(j));
        for (int h = 0; h < 2; h++) {
          j.s[h] = f.s[h] + f.from;
          j.p0[h] = ch.sb[0].p[h];
          j.x1[h] = ch.sb[0].xn[h];
          for (int q2 = 0; q2 < pd.nlms; q2++) j.pk[q2][h] = ch.sb[q2 + 1].p[h];
          j.e[h] = ch.sb[s].e[h];
          j.u[h] = ch.sb[s].u[h];
        }
        j.umax = ch.sb[s].umax;
        j.alpha = ch.coefs[pd.cmix];
        j.t0 = 0;
        j.n = f.n;
        j.state = carry ? state[s][q] : nullptr;
        jobs.push_back(j);
      }
      n_jobs[s] = (int)jobs.size();
      o_jobs[s] = stg.add(jobs.data(), jobs.size() * sizeof(MixJob));
    }
  }
  char *d_tables = arena_alloc<char>(c, stg.host.size() + 256);
  if (!d_tables) { c->err = "device arena exhausted"; return 1; }
  HP_MARK(hp1);
  if (upload_bytes(c, d_tables, stg.host.data(), stg.host.size(), c->st)) return 1;
  HP_MARK(hp2);
  std::vector<void *> d_jobs(nstage, nullptr);
  for (int s = lead; s < nstage; s++) d_jobs[s] = d_tables + o_jobs[s];
  // Large batches: the lead stage's chains in two halves on two streams (LEAD_SPLIT_MIN chains or more: more workgroups than the chip
  // holds at once). A chunk launch ends with its slowest workgroup; with one sequence of launches the chip runs half empty at the end
  // of every chunk, with two that drift apart one's tail is the other's head.
  int nseq = 1;  // launch sequences of the lead stage
  // streams for the further sequences: sL2 (every CU) with the large-batch set; with the default set (17..32 frames) two idle streams
  // of the small-batch set, whose 192 CUs are a subset of the default set's 224 (no new hardware queue: see sL2)
  hipStream_t xl[2] = {nullptr, nullptr};
  if (c->big[0] && c->st == c->big[0]) { xl[0] = c->sL2[0]; xl[1] = c->sL2[1]; }
  else if (LEAD_SPLIT_DEFSET && c->split[0] && c->st != c->split[0]) { xl[0] = c->split[0]; xl[1] = c->split[2]; }
  if (LEAD_SPLIT_MIN > 0 && carry && lead <= pd.nlms && (LEAD_SPLIT_ANY || pd.lms[lead - 1].S == 1280) && n_jobs[lead] >= LEAD_SPLIT_MIN)
    while (nseq < LEAD_SPLIT_WAYS && nseq - 1 < 2 && xl[nseq - 1]) nseq++;
  const bool two = nseq > 1;
  if (two) {
    hipEvent_t e0;
    TAKE_EVENT(c, e0, false);
    HIPCK(c, hipEventRecord(e0, c->st));  // tables are up, the round before is over
    for (int q = 1; q < nseq; q++) HIPCK(c, hipStreamWaitEvent(xl[q - 1], e0, 0));
  }
  for (int k = 0; k < nchunk; k++) {
    const long long units = units_all[k];
    const int ct0 = k * chunk, clen = carry ? chunk : 0;  // (a single launch covers the jobs' own spans)
    // ---- launches: lead stage on c->st, the first follower on c->sC, the others on c->sE, each behind an event: three
    // pipeline stages per chunk (one follower stream ran (32,16), (8,4) and the mixer of a chunk back to back, as long as
    // an lms2 lead chunk: the followers fell behind and left a tail of three chunks after the lead's last one)
    for (int s = lead; s < nstage; s++) {
      hipStream_t st = (s == lead) ? c->st : ((s == lead + 1 || !c->sE) ? c->sC : c->sE);
      if (n_jobs[s]) {
        if (s <= pd.nlms) {
          const StageDesc &sd = pd.lms[s - 1];
          // the dominant kernel (k_lms1c for the high profile, the first LMS stage otherwise) is timed for the roofline
          const bool dominant = (s == 1) && (sd.S != 1280 || sacopt_lms_variant(sd.S, sd.D, n_jobs[s], c->bits, L1C_WINDOW) == c->timed_variant);
          hipEvent_t a = nullptr, b = nullptr;
          if (dominant) {
            TAKE_EVENT(c, a, true);
            TAKE_EVENT(c, b, true);
            HIPCK(c, hipEventRecord(a, st));
          }
          const int nA = (two && s == lead) ? n_jobs[s] / nseq : n_jobs[s];
          if (sacopt_launch_lms(sd.S, sd.D, (const LmsJob *)d_jobs[s], nA, c->d_tab, c->bits, st, L1C_WINDOW, ct0, clen)) { c->err = "k_lms launch failed"; return 1; }
          if (dominant) {
            HIPCK(c, hipEventRecord(b, st));
            c->lms_ev.push_back({a, b});
            c->lms_units += units * nA / n_jobs[s];
          }
          if (nA < n_jobs[s]) {  // the other parts on the further lead streams: the launch sequences drift apart, one's tail fills with another's head
            for (int q = 1; q < nseq; q++) {
              const int j0 = nA * q, j1 = q == nseq - 1 ? n_jobs[s] : nA * (q + 1);
              if (sacopt_launch_lms(sd.S, sd.D, (const LmsJob *)d_jobs[s] + j0, j1 - j0, c->d_tab, c->bits, xl[q - 1], L1C_WINDOW, ct0, clen)) { c->err = "k_lms launch failed"; return 1; }
              hipEvent_t e2;
              TAKE_EVENT(c, e2, false);
              HIPCK(c, hipEventRecord(e2, xl[q - 1]));
              HIPCK(c, hipStreamWaitEvent(lead < nstage - 1 ? c->sC : c->st, e2, 0));
            }
          }
        } else {
          if (sacopt_launch_mix(pd.nlms, (const MixJob *)d_jobs[s], n_jobs[s], st, ct0, clen)) { c->err = "k_mix launch failed"; return 1; }
        }
      }
      if (s == lead && lead < nstage - 1) {
        hipEvent_t e;
        TAKE_EVENT(c, e, false);
        HIPCK(c, hipEventRecord(e, c->st));
        HIPCK(c, hipStreamWaitEvent(c->sC, e, 0));
      }
      if (s == lead + 1 && c->sE && s < nstage - 1) {
        hipEvent_t e;
        TAKE_EVENT(c, e, false);
        HIPCK(c, hipEventRecord(e, c->sC));
        HIPCK(c, hipStreamWaitEvent(c->sE, e, 0));
      }
    }
  }
  if (lead < nstage - 1) {  // join: c->st continues (cost kernel, copies) after the follower stages
    hipStream_t lasts = (c->sE && lead + 1 < nstage - 1) ? c->sE : c->sC;
    TAKE_EVENT(c, ev_join, false);
    HIPCK(c, hipEventRecord(ev_join, lasts));
    HIPCK(c, hipStreamWaitEvent(c->st, ev_join, 0));
  }
  HP_MARK(hp3);
  HP_PRINT("eval_chains: %zu chains, build %.2f ms, upload %.2f ms (%zu bytes), launches %.2f ms\n", chains.size(), hp1 - hp0, hp2 - hp1, stg.host.size(), hp3 - hp2);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Full-frame pass, pipelined. A frame is one long time-serial chain per stage; stages are
// feed-forward, so the chain is cut into chunks of PIPE_CHUNK samples and stage s works on chunk
// k while stage s-1 works on chunk k+1 (stage state -- LMS weights/energies, RLS P/w -- is
// carried through small state buffers, histories are re-read from the stage inputs). The
// dominant LMS stage runs on stream sA, all other stages on sB; the critical path is the
// dominant stage alone. Everything is enqueued without host synchronisation, so a caller can
// overlap it with work on c->st (the optimizer rounds). Results are identical to eval_chains.
// `done` receives an event recorded when the last stage has finished.
// `staged` (optional): event behind which the input samples are on the device. With it the pass does not touch c->st at
// all -- its job tables go up on c->sB -- so it can be enqueued AFTER work that already sits on c->st (the optimizer's
// first round) without waiting for it.
static int final_pass_enqueue(sacopt_ctx *c, const ProfDesc &pd, const std::vector<DevFrame> &fr, std::vector<Chain> &chains,
                              hipEvent_t *done, hipEvent_t staged = nullptr) {
  struct UpGuard { sacopt_ctx *c; ~UpGuard() { c->up = nullptr; } } up_guard{c};
  if (staged) {
    HIPCK(c, hipStreamWaitEvent(c->sB, staged, 0));
    c->up = c->sB;
  }
  const int nstage = pd.nlms + 2, last = pd.nlms + 1;
  const int nch = (int)chains.size();
  int max_n = 0;
  for (Chain &ch : chains) {
    const int n = fr[ch.frame].n;
    max_n = n > max_n ? n : max_n;
    for (int s = 0; s < nstage; s++) {
      StageBuf b;
      if (s != last) {
        for (int k = 0; k < 2; k++) {
          b.p[k] = arena_alloc<double>(c, n);
          b.xn[k] = arena_alloc<double>(c, n);
          if (!b.p[k] || !b.xn[k]) { c->err = "device arena exhausted"; return 1; }
        }
      } else {
        for (int k = 0; k < 2; k++) {
          b.e[k] = arena_alloc<int32_t>(c, n);
          b.u[k] = arena_alloc<int32_t>(c, n);
          if (!b.e[k] || !b.u[k]) { c->err = "device arena exhausted"; return 1; }
        }
        b.umax = arena_alloc<int32_t>(c, 2);
        if (!b.umax) { c->err = "device arena exhausted"; return 1; }
      }
      ch.sb[s] = b;
    }
  }
  // Chunk edges (multiples of LMS_CHUNK): a short first chunk -- its LPC stage is the only work in front of the lead
  // stage -- then PIPE_CHUNK samples each, and a tail that halves down to 2 * LMS_CHUNK: what remains after the lead stage's
  // last launch is the follower stages of the LAST chunk only.
  std::vector<int> edges = {0};
  {
    int t = 0;
    if (max_n > 4 * PIPE_CHUNK) { t = 2 * LMS_CHUNK; edges.push_back(t); }
    while (max_n - t > PIPE_CHUNK + PIPE_CHUNK / 2) { t += PIPE_CHUNK; edges.push_back(t); }
    while (max_n - t > 2 * LMS_CHUNK) {
      const int rem = max_n - t;
      t += ((rem / 2 + LMS_CHUNK - 1) / LMS_CHUNK) * LMS_CHUNK;
      edges.push_back(t);
    }
    if (t < max_n || edges.size() == 1) edges.push_back(max_n);
  }
  const int nchunk = (int)edges.size() - 1;
  auto chunk_end = [&](int k, int n) { return n < edges[k + 1] ? n : edges[k + 1]; };
  // ---- all job tables, built and uploaded up front
  std::vector<LpcJob> lpc(nch);
  for (int i = 0; i < nch; i++) {
    const Chain &ch = chains[i];
    const DevFrame &f = fr[ch.frame];
    LpcJob &j = lpc[i];
    memset(&j, 0, sizeof(j));
    for (int k = 0; k < 2; k++) { j.s[k] = f.s[k] + f.from; j.p[k] = ch.sb[0].p[k]; j.x1[k] = ch.sb[0].xn[k]; }
    j.alpha = ch.coefs[0];
    j.n = f.n;
    if (lpc_scratch_alloc(c, pd, j)) return 1;  // stays allocated for the whole call
    for (int k = 0; k < 2; k++) {  // the LPC stage is chunked too: covariance / BlendErr state carried
      j.cstate[k] = arena_alloc<double>(c, pd.ols_n * (pd.ols_n + 1) / 2 + pd.ols_n);
      j.estate[k] = arena_alloc<double>(c, pd.ols_n);
      if (!j.cstate[k] || !j.estate[k]) { c->err = "device arena exhausted"; return 1; }
    }
  }
  Stager stg;
  std::vector<LpcJob *> d_lpc(nchunk, nullptr);
  std::vector<size_t> o_lpc(nchunk, 0), o_mix(nchunk, 0);
  std::vector<std::vector<size_t>> o_lms(pd.nlms, std::vector<size_t>(nchunk, 0));
  std::vector<int> n_lpc(nchunk, 0), len_lpc(nchunk, 0);
  for (int k = 0; k < nchunk; k++) {
    const int t0 = edges[k];
    std::vector<LpcJob> jobs;
    for (int i = 0; i < nch; i++) {
      const int n = fr[chains[i].frame].n;
      if (t0 >= n) continue;
      LpcJob j = lpc[i];
      j.t0 = t0;
      j.n = chunk_end(k, n);
      j.nsnap = j.n / 8;
      len_lpc[k] = (j.n - t0 > len_lpc[k]) ? j.n - t0 : len_lpc[k];
      jobs.push_back(j);
    }
    n_lpc[k] = (int)jobs.size();
    o_lpc[k] = stg.add(jobs.data(), jobs.size() * sizeof(LpcJob));
  }
  std::vector<std::vector<LmsJob *>> d_lms(pd.nlms, std::vector<LmsJob *>(nchunk, nullptr));
  std::vector<std::vector<int>> n_lms(pd.nlms, std::vector<int>(nchunk, 0));
  std::vector<MixJob *> d_mix(nchunk, nullptr);
  std::vector<int> n_mix(nchunk, 0);
  std::vector<long long> units(nchunk, 0);
  std::vector<std::vector<double *>> lms_state(pd.nlms, std::vector<double *>(nch, nullptr));
  std::vector<double *> mix_state(nch, nullptr);
  for (int i = 0; i < nch; i++) {
    for (int q = 0; q < pd.nlms; q++) {
      lms_state[q][i] = arena_alloc<double>(c, sacopt_lms_state_doubles(pd.lms[q].S, pd.lms[q].D));
      if (!lms_state[q][i]) { c->err = "device arena exhausted"; return 1; }
    }
    mix_state[i] = arena_alloc<double>(c, 2 * MIX_STATE);
    if (!mix_state[i]) { c->err = "device arena exhausted"; return 1; }
  }
  for (int k = 0; k < nchunk; k++) {
    const int t0 = edges[k];
    for (int q = 0; q < pd.nlms; q++) {
      std::vector<LmsJob> jobs;
      for (int i = 0; i < nch; i++) {
        const Chain &ch = chains[i];
        const int n = fr[ch.frame].n;
        if (t0 >= n) continue;
        LmsJob j;
        for (int h = 0; h < 2; h++) { j.x[h] = ch.sb[q].xn[h]; j.p[h] = ch.sb[q + 1].p[h]; j.xn[h] = ch.sb[q + 1].xn[h]; }
        j.mu1 = ch.coefs[pd.lms[q].c1];
        j.mu2 = ch.coefs[pd.lms[q].c2];
        j.t0 = t0;
        j.n = chunk_end(k, n) - t0;
        j.state = lms_state[q][i];
        jobs.push_back(j);
        if (q == 0) units[k] += j.n;
      }
      n_lms[q][k] = (int)jobs.size();
      o_lms[q][k] = stg.add(jobs.data(), jobs.size() * sizeof(LmsJob));
    }
    std::vector<MixJob> jobs;
    for (int i = 0; i < nch; i++) {
      const Chain &ch = chains[i];
      const DevFrame &f = fr[ch.frame];
      if (t0 >= f.n && !(k == 0 && f.n == 0)) continue;
      MixJob j;
      memset(&j, 0, sizeof(j));
      for (int h = 0; h < 2; h++) {
        j.s[h] = f.s[h] + f.from;
        j.p0[h] = ch.sb[0].p[h];
        j.x1[h] = ch.sb[0].xn[h];
        for (int q = 0; q < pd.nlms; q++) j.pk[q][h] = ch.sb[q + 1].p[h];
        j.e[h] = ch.sb[last].e[h];
        j.u[h] = ch.sb[last].u[h];
      }
      j.umax = ch.sb[last].umax;
      j.alpha = ch.coefs[pd.cmix];
      j.t0 = t0;
      j.n = chunk_end(k, f.n) - t0;
      j.state = mix_state[i];
      jobs.push_back(j);
    }
    n_mix[k] = (int)jobs.size();
    o_mix[k] = stg.add(jobs.data(), jobs.size() * sizeof(MixJob));
  }
  {
    char *dev = arena_alloc<char>(c, stg.host.size() + 256);
    if (!dev) { c->err = "device arena exhausted"; return 1; }
    if (upload_bytes(c, dev, stg.host.data(), stg.host.size(), c->up ? c->up : c->st)) return 1;
    for (int k = 0; k < nchunk; k++) {
      d_lpc[k] = reinterpret_cast<LpcJob *>(dev + o_lpc[k]);
      d_mix[k] = reinterpret_cast<MixJob *>(dev + o_mix[k]);
      for (int q = 0; q < pd.nlms; q++) d_lms[q][k] = reinterpret_cast<LmsJob *>(dev + o_lms[q][k]);
    }
  }
  // ---- enqueue
  auto new_event = [&](hipEvent_t *e) -> int {
    TAKE_EVENT(c, *e, false);
    return 0;
  };
  // sA owns 32 CUs: right for a handful of frames (2 workgroups each), wrong for a large batch
  const int lead_wgs = nch * (sacopt_lms_variant(pd.lms[0].S, pd.lms[0].D, nch, c->bits, L1C_FRAMES) ? 2 : 1);
  hipStream_t sLead = (c->sAall && lead_wgs > 32) ? c->sAall : c->sA;
  hipEvent_t ev_in;
  if (new_event(&ev_in)) return 1;
  HIPCK(c, hipEventRecord(ev_in, staged ? c->sB : c->st));  // inputs + job tables are on the device behind this
  c->up = nullptr;
  if (!staged) HIPCK(c, hipStreamWaitEvent(c->sB, ev_in, 0));
  HIPCK(c, hipStreamWaitEvent(sLead, ev_in, 0));
  // The LPC stage runs on sB two chunks ahead of the lead stage (it is ~10x cheaper), so only its first
  // chunk is ever in front of the lead stage
  std::vector<hipEvent_t> ev_lpc(nchunk, nullptr);
  auto lpc_chunk = [&](int k) -> int {
    if (k >= nchunk) return 0;
    if (sacopt_launch_lpc3(d_lpc[k], n_lpc[k], len_lpc[k], pd.ols_n, c->sB)) { c->err = "LPC3 launch failed"; return 1; }
    if (new_event(&ev_lpc[k])) return 1;
    HIPCK(c, hipEventRecord(ev_lpc[k], c->sB));
    return 0;
  };
  if (lpc_chunk(0) || lpc_chunk(1)) return 1;
  hipStream_t last_stream = c->sB;
  for (int k = 0; k < nchunk; k++) {
    HIPCK(c, hipStreamWaitEvent(sLead, ev_lpc[k], 0));
    hipEvent_t a = nullptr, b = nullptr, e1;
    const bool dominant = pd.lms[0].S != 1280 || sacopt_lms_variant(pd.lms[0].S, pd.lms[0].D, n_lms[0][k], c->bits, L1C_FRAMES) == c->timed_variant;
    if (dominant) {
      TAKE_EVENT(c, a, true);
      TAKE_EVENT(c, b, true);
      HIPCK(c, hipEventRecord(a, sLead));
    }
    if (sacopt_launch_lms(pd.lms[0].S, pd.lms[0].D, d_lms[0][k], n_lms[0][k], c->d_tab, c->bits, sLead, L1C_FRAMES, 0, 0)) { c->err = "k_lms launch failed"; return 1; }
    if (dominant) {
      HIPCK(c, hipEventRecord(b, sLead));
      c->lms_ev.push_back({a, b});
      c->lms_units += units[k];
    }
    if (new_event(&e1)) return 1;
    HIPCK(c, hipEventRecord(e1, sLead));
    HIPCK(c, hipStreamWaitEvent(c->sB, e1, 0));
    // second LMS stage on sB, everything behind it on sD: three pipeline stages per chunk, so that the
    // followers of chunk k together take no longer than the lead stage of chunk k+1
    hipStream_t fs = c->sB;
    for (int q = 1; q < pd.nlms; q++) {
      if (sacopt_launch_lms(pd.lms[q].S, pd.lms[q].D, d_lms[q][k], n_lms[q][k], c->d_tab, c->bits, fs, L1C_FRAMES, 0, 0)) { c->err = "k_lms launch failed"; return 1; }
      if (q == 1 && c->sD) {
        hipEvent_t e2;
        if (new_event(&e2)) return 1;
        HIPCK(c, hipEventRecord(e2, c->sB));
        HIPCK(c, hipStreamWaitEvent(c->sD, e2, 0));
        fs = c->sD;
        if (lpc_chunk(k + 2)) return 1;
      }
    }
    if (!c->sD && lpc_chunk(k + 2)) return 1;
    if (sacopt_launch_mix(pd.nlms, d_mix[k], n_mix[k], fs, 0, 0)) { c->err = "k_mix launch failed"; return 1; }
    last_stream = fs;
  }
  if (new_event(done)) return 1;
  HIPCK(c, hipEventRecord(*done, last_stream));  // the mixer of the last chunk ran there
  return 0;
}

// ---------------------------------------------------------------------------------------------
// The same pipeline enqueued in FOUR parts, for sacopt_encode_frames_selected: the frame is predicted with the parameters the
// optimizer selects, and the optimizer fixes them in coefficient order (codec.cpp:389-406) -- the OLS stage's after round 0,
// the lead stage's after round 2, ... -- so every part of the full-frame pass starts as soon as ITS parameters are final instead
// of after the last round. Parts: 0 = OLS stage (c->sB), 1 = first LMS stage (sLead), 2 = second LMS stage (c->sB), 3 = the
// stages behind it and the mixer (c->sD), chunk k of a part behind an event of chunk k of the part before. Same chunk edges,
// job contents and kernels as final_pass_enqueue: identical results.
struct FinalPipe {
  sacopt_ctx *c = nullptr;
  const ProfDesc *pd = nullptr;
  const std::vector<DevFrame> *fr = nullptr;
  std::vector<Chain> *chains = nullptr;
  std::vector<int> edges;
  int nchunk = 0, nch = 0;
  std::vector<LpcJob> lpc;
  std::vector<std::vector<double *>> lms_state;
  std::vector<double *> mix_state;
  std::vector<hipEvent_t> ev_lpc, ev_l0, ev_l1;
  hipStream_t sLead = nullptr;
  hipEvent_t staged = nullptr, done = nullptr;
  bool queued[4] = {false, false, false, false};

  int chunk_end(int k, int n) const { return n < edges[k + 1] ? n : edges[k + 1]; }
  // the last coefficient (among the first ncoef_loop, the ones the optimizer decides) that part g reads
  int needs(int g, int ncoef_loop) const {
    int m = -1;
    auto use = [&](int ci) { if (ci < ncoef_loop && ci > m) m = ci; };
    use(0);
    if (g >= 1) { use(pd->lms[0].c1); use(pd->lms[0].c2); }
    if (g >= 2 && pd->nlms > 1) { use(pd->lms[1].c1); use(pd->lms[1].c2); }
    if (g >= 3) { for (int q = 2; q < pd->nlms; q++) { use(pd->lms[q].c1); use(pd->lms[q].c2); } use(pd->cmix); }
    return m;
  }
  int prepare(sacopt_ctx *ctx, const ProfDesc &p, const std::vector<DevFrame> &f, std::vector<Chain> &ch, hipEvent_t staged_ev) {
    c = ctx; pd = &p; fr = &f; chains = &ch; staged = staged_ev;
    const int nstage = p.nlms + 2, last = p.nlms + 1;
    nch = (int)ch.size();
    int max_n = 0;
    for (Chain &x : ch) {
      const int n = f[x.frame].n;
      max_n = n > max_n ? n : max_n;
      for (int s = 0; s < nstage; s++) {
        StageBuf b;
        if (s != last) {
          for (int k = 0; k < 2; k++) {
            b.p[k] = arena_alloc<double>(c, n);
            b.xn[k] = arena_alloc<double>(c, n);
            if (!b.p[k] || !b.xn[k]) { c->err = "device arena exhausted"; return 1; }
          }
        } else {
          for (int k = 0; k < 2; k++) {
            b.e[k] = arena_alloc<int32_t>(c, n);
            b.u[k] = arena_alloc<int32_t>(c, n);
            if (!b.e[k] || !b.u[k]) { c->err = "device arena exhausted"; return 1; }
          }
          b.umax = arena_alloc<int32_t>(c, 2);
          if (!b.umax) { c->err = "device arena exhausted"; return 1; }
        }
        x.sb[s] = b;
      }
    }
    edges = {0};  // (as final_pass_enqueue)
    {
      int t = 0;
      if (max_n > 4 * PIPE_CHUNK) { t = 2 * LMS_CHUNK; edges.push_back(t); }
      while (max_n - t > PIPE_CHUNK + PIPE_CHUNK / 2) { t += PIPE_CHUNK; edges.push_back(t); }
      while (max_n - t > 2 * LMS_CHUNK) {
        const int rem = max_n - t;
        t += ((rem / 2 + LMS_CHUNK - 1) / LMS_CHUNK) * LMS_CHUNK;
        edges.push_back(t);
      }
      if (t < max_n || edges.size() == 1) edges.push_back(max_n);
    }
    nchunk = (int)edges.size() - 1;
    lpc.resize(nch);
    for (int i = 0; i < nch; i++) {
      const Chain &x = ch[i];
      const DevFrame &df = f[x.frame];
      LpcJob &j = lpc[i];
      memset(&j, 0, sizeof(j));
      for (int k = 0; k < 2; k++) { j.s[k] = df.s[k] + df.from; j.p[k] = x.sb[0].p[k]; j.x1[k] = x.sb[0].xn[k]; }
      j.n = df.n;
      if (lpc_scratch_alloc(c, p, j)) return 1;
      for (int k = 0; k < 2; k++) {
        j.cstate[k] = arena_alloc<double>(c, p.ols_n * (p.ols_n + 1) / 2 + p.ols_n);
        j.estate[k] = arena_alloc<double>(c, p.ols_n);
        if (!j.cstate[k] || !j.estate[k]) { c->err = "device arena exhausted"; return 1; }
      }
    }
    lms_state.assign(p.nlms, std::vector<double *>(nch, nullptr));
    mix_state.assign(nch, nullptr);
    for (int i = 0; i < nch; i++) {
      for (int q = 0; q < p.nlms; q++) {
        lms_state[q][i] = arena_alloc<double>(c, sacopt_lms_state_doubles(p.lms[q].S, p.lms[q].D));
        if (!lms_state[q][i]) { c->err = "device arena exhausted"; return 1; }
      }
      mix_state[i] = arena_alloc<double>(c, 2 * MIX_STATE);
      if (!mix_state[i]) { c->err = "device arena exhausted"; return 1; }
    }
    ev_lpc.assign(nchunk, nullptr); ev_l0.assign(nchunk, nullptr); ev_l1.assign(nchunk, nullptr);
    const int lead_wgs = nch * (sacopt_lms_variant(p.lms[0].S, p.lms[0].D, nch, c->bits, L1C_FRAMES) ? 2 : 1);
    sLead = (c->sAall && lead_wgs > 32) ? c->sAall : c->sA;
    return 0;
  }
  // job table of LMS stage q, chunk k (the chains still running in that chunk)
  void lms_jobs(int q, int k, std::vector<LmsJob> &jobs, long long *units) const {
    const int t0 = edges[k];
    for (int i = 0; i < nch; i++) {
      const Chain &x = (*chains)[i];
      const int n = (*fr)[x.frame].n;
      if (t0 >= n) continue;
      LmsJob j;
      for (int h = 0; h < 2; h++) { j.x[h] = x.sb[q].xn[h]; j.p[h] = x.sb[q + 1].p[h]; j.xn[h] = x.sb[q + 1].xn[h]; }
      j.mu1 = x.coefs[pd->lms[q].c1];
      j.mu2 = x.coefs[pd->lms[q].c2];
      j.t0 = t0;
      j.n = chunk_end(k, n) - t0;
      j.state = lms_state[q][i];
      jobs.push_back(j);
      if (units) *units += j.n;
    }
  }
  // Enqueue part g with the coefficients the chains hold NOW. Parts must be enqueued in order.
  int enqueue(int g) {
    if (queued[g]) return 0;
    queued[g] = true;
    const int last = pd->nlms + 1;
    hipStream_t fs = c->sD ? c->sD : c->sB;
    hipStream_t st = g == 0 ? c->sB : (g == 1 ? sLead : (g == 2 ? c->sB : fs));
    Stager stg;
    std::vector<size_t> off;  // per chunk (and stage, part 3) in launch order
    std::vector<int> cnt;
    std::vector<long long> units(nchunk, 0);
    std::vector<int> len_lpc(nchunk, 0);
    for (int k = 0; k < nchunk; k++) {
      const int t0 = edges[k];
      if (g == 0) {
        std::vector<LpcJob> jobs;
        for (int i = 0; i < nch; i++) {
          const int n = (*fr)[(*chains)[i].frame].n;
          if (t0 >= n) continue;
          LpcJob j = lpc[i];
          j.alpha = (*chains)[i].coefs[0];
          j.t0 = t0;
          j.n = chunk_end(k, n);
          j.nsnap = j.n / 8;
          len_lpc[k] = (j.n - t0 > len_lpc[k]) ? j.n - t0 : len_lpc[k];
          jobs.push_back(j);
        }
        cnt.push_back((int)jobs.size());
        off.push_back(stg.add(jobs.data(), jobs.size() * sizeof(LpcJob)));
      } else if (g == 1 || g == 2) {
        std::vector<LmsJob> jobs;
        lms_jobs(g - 1, k, jobs, g == 1 ? &units[k] : nullptr);
        cnt.push_back((int)jobs.size());
        off.push_back(stg.add(jobs.data(), jobs.size() * sizeof(LmsJob)));
      } else {
        for (int q = 2; q < pd->nlms; q++) {
          std::vector<LmsJob> jobs;
          lms_jobs(q, k, jobs, nullptr);
          cnt.push_back((int)jobs.size());
          off.push_back(stg.add(jobs.data(), jobs.size() * sizeof(LmsJob)));
        }
        std::vector<MixJob> jobs;
        for (int i = 0; i < nch; i++) {
          const Chain &x = (*chains)[i];
          const DevFrame &f = (*fr)[x.frame];
          if (t0 >= f.n && !(k == 0 && f.n == 0)) continue;
          MixJob j;
          memset(&j, 0, sizeof(j));
          for (int h = 0; h < 2; h++) {
            j.s[h] = f.s[h] + f.from;
            j.p0[h] = x.sb[0].p[h];
            j.x1[h] = x.sb[0].xn[h];
            for (int q = 0; q < pd->nlms; q++) j.pk[q][h] = x.sb[q + 1].p[h];
            j.e[h] = x.sb[last].e[h];
            j.u[h] = x.sb[last].u[h];
          }
          j.umax = x.sb[last].umax;
          j.alpha = x.coefs[pd->cmix];
          j.t0 = t0;
          j.n = chunk_end(k, f.n) - t0;
          j.state = mix_state[i];
          jobs.push_back(j);
        }
        cnt.push_back((int)jobs.size());
        off.push_back(stg.add(jobs.data(), jobs.size() * sizeof(MixJob)));
      }
    }
    char *dev = arena_alloc<char>(c, stg.host.size() + 256);
    if (!dev) { c->err = "device arena exhausted"; return 1; }
    if (g == 0 || g == 1) HIPCK(c, hipStreamWaitEvent(st, staged, 0));  // the frames are on the device behind this
    if (upload_bytes(c, dev, stg.host.data(), stg.host.size(), st)) return 1;
    size_t idx = 0;
    for (int k = 0; k < nchunk; k++) {
      if (g == 0) {
        if (sacopt_launch_lpc3(reinterpret_cast<LpcJob *>(dev + off[idx]), cnt[idx], len_lpc[k], pd->ols_n, st)) { c->err = "LPC3 launch failed"; return 1; }
        idx++;
        TAKE_EVENT(c, ev_lpc[k], false);
        HIPCK(c, hipEventRecord(ev_lpc[k], st));
      } else if (g == 1) {
        HIPCK(c, hipStreamWaitEvent(st, ev_lpc[k], 0));
        hipEvent_t a = nullptr, b = nullptr;
        const bool dominant = pd->lms[0].S != 1280 || sacopt_lms_variant(pd->lms[0].S, pd->lms[0].D, cnt[idx], c->bits, L1C_FRAMES) == c->timed_variant;
        if (dominant) {
          TAKE_EVENT(c, a, true);
          TAKE_EVENT(c, b, true);
          HIPCK(c, hipEventRecord(a, st));
        }
        if (sacopt_launch_lms(pd->lms[0].S, pd->lms[0].D, reinterpret_cast<LmsJob *>(dev + off[idx]), cnt[idx], c->d_tab, c->bits, st, L1C_FRAMES, 0, 0)) { c->err = "k_lms launch failed"; return 1; }
        idx++;
        if (dominant) {
          HIPCK(c, hipEventRecord(b, st));
          c->lms_ev.push_back({a, b});
          c->lms_units += units[k];
        }
        TAKE_EVENT(c, ev_l0[k], false);
        HIPCK(c, hipEventRecord(ev_l0[k], st));
      } else if (g == 2) {
        HIPCK(c, hipStreamWaitEvent(st, ev_l0[k], 0));
        if (sacopt_launch_lms(pd->lms[1].S, pd->lms[1].D, reinterpret_cast<LmsJob *>(dev + off[idx]), cnt[idx], c->d_tab, c->bits, st, L1C_FRAMES, 0, 0)) { c->err = "k_lms launch failed"; return 1; }
        idx++;
        TAKE_EVENT(c, ev_l1[k], false);
        HIPCK(c, hipEventRecord(ev_l1[k], st));
      } else {
        HIPCK(c, hipStreamWaitEvent(st, ev_l1[k], 0));
        for (int q = 2; q < pd->nlms; q++) {
          if (sacopt_launch_lms(pd->lms[q].S, pd->lms[q].D, reinterpret_cast<LmsJob *>(dev + off[idx]), cnt[idx], c->d_tab, c->bits, st, L1C_FRAMES, 0, 0)) { c->err = "k_lms launch failed"; return 1; }
          idx++;
        }
        if (sacopt_launch_mix(pd->nlms, reinterpret_cast<MixJob *>(dev + off[idx]), cnt[idx], st, 0, 0)) { c->err = "k_mix launch failed"; return 1; }
        idx++;
      }
    }
    if (g == 3) {
      TAKE_EVENT(c, done, false);
      HIPCK(c, hipEventRecord(done, st));
    }
    return 0;
  }
};

// Full-frame (or any window) pass for all chains; long windows go through the pipeline.
static bool final_pass_pipelined(const ProfDesc &pd, const std::vector<DevFrame> &fr, const std::vector<Chain> &chains) {
  int max_n = 0;
  for (const Chain &ch : chains) max_n = fr[ch.frame].n > max_n ? fr[ch.frame].n : max_n;
  return !(pd.type == SACOPT_PROFILE_FAST || max_n <= 2 * PIPE_CHUNK);
}
static int run_final_pass(sacopt_ctx *c, const ProfDesc &pd, const std::vector<DevFrame> &fr, std::vector<Chain> &chains,
                          hipEvent_t *done, hipEvent_t staged = nullptr) {
  *done = nullptr;
  if (!final_pass_pipelined(pd, fr, chains)) {
    Memo memo;
    return eval_chains(c, pd, fr, chains, memo);
  }
  return final_pass_enqueue(c, pd, fr, chains, done, staged);
}

// Caller-supplied frames and windows, checked BEFORE anything is sized from them (64-bit arithmetic: from + n may overflow int)
static int check_frames(sacopt_ctx *c, int nframes, const sacopt_frame *frames, const int32_t *from, const int32_t *n) {
  for (int f = 0; f < nframes; f++) {
    const long long ns = frames[f].numsamples, fr = from ? from[f] : 0, nn = n ? n[f] : ns - fr;
    if (ns < 0) { c->err = "negative numsamples"; return 1; }
    if (fr < 0 || nn < 0 || fr + nn > ns) { c->err = "window outside frame"; return 1; }
    if (ns && (!frames[f].s0 || !frames[f].s1)) { c->err = "null sample pointer"; return 1; }
  }
  return 0;
}

// Stage input frames: device pointers are used in place, host pointers are copied into the arena.
static int stage_frames(sacopt_ctx *c, int nframes, const sacopt_frame *frames, const int32_t *from, const int32_t *n,
                        std::vector<DevFrame> &out) {
  out.resize(nframes);
  for (int f = 0; f < nframes; f++) {
    DevFrame &d = out[f];
    d.numsamples = frames[f].numsamples;
    d.from = from ? from[f] : 0;
    d.n = n ? n[f] : d.numsamples - d.from;
    if (d.numsamples < 0 || d.from < 0 || d.n < 0 || d.from + d.n > d.numsamples) { c->err = "window outside frame"; return 1; }
    const int32_t *src[2] = {frames[f].s0, frames[f].s1};
    for (int k = 0; k < 2; k++) {
      if (d.numsamples == 0) { d.s[k] = arena_alloc<int32_t>(c, 1); continue; }
      if (!src[k]) { c->err = "null sample pointer"; return 1; }
      if (is_device_ptr(src[k])) d.s[k] = src[k];
      else {
        int32_t *p = arena_alloc<int32_t>(c, d.numsamples);
        if (!p) { c->err = "device arena exhausted"; return 1; }
        HIPCK(c, hipMemcpyAsync(p, src[k], (size_t)d.numsamples * 4, hipMemcpyHostToDevice, c->st));
        d.s[k] = p;
      }
    }
  }
  return 0;
}

static int copy_out(sacopt_ctx *c, void *dst, const void *src_dev, size_t bytes) {
  if (!dst || !bytes) return 0;
  HIPCK(c, hipMemcpyAsync(dst, src_dev, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->st));
  return 0;
}

static int ilog2(int v) { int nb = 0; while (v >>= 1) nb++; return nb; }  // MathUtils::iLog2 (utils.h:42-46)

// Residual outputs of a finished full pass (all work joined into c->st): error / s2u_error copies,
// maxbpn, and -- when the caller asks for s2u_error_map -- Remap::Analyse over the whole frame
// (codec.cpp:448-452) + Remap::Map per sample (codec.cpp:175-204) on the device. Ends the timed region.
static size_t map_bytes(int n) { return 2 * ((((size_t)MAP_TAB * 4 + 255) & ~(size_t)255) + (((size_t)n * 4 + 255) & ~(size_t)255) + 256 + sizeof(MapJob)) + 1024; }
static int emit_outputs(sacopt_ctx *c, const ProfDesc &pd, const std::vector<DevFrame> &fr, const std::vector<Chain> &chains,
                        sacopt_residuals *out) {
  const int nframes = (int)fr.size();
  const int last = (pd.type == SACOPT_PROFILE_FAST) ? 0 : pd.nlms + 1;
  std::vector<int32_t> um(2 * (size_t)nframes), mm(2 * (size_t)nframes, 0);
  std::vector<MapJob> mj;
  std::vector<int> mj_frame;
  int max_ns = 0, max_n = 0;
  for (int f = 0; f < nframes; f++) {
    if (!out[f].map0 && !out[f].map1) continue;
    const StageBuf &b = chains[f].sb[last];
    for (int k = 0; k < 2; k++) {
      MapJob j;
      j.s = fr[f].s[k];
      j.sw = fr[f].s[k] + fr[f].from;
      j.e = b.e[k];
      int32_t *dst = k ? out[f].map1 : out[f].map0;
      j.m = (dst && is_device_ptr(dst)) ? dst : (dst ? arena_alloc<int32_t>(c, fr[f].n > 0 ? fr[f].n : 1) : nullptr);
      j.tab = arena_alloc<int32_t>(c, MAP_TAB);
      j.mmax = arena_alloc<int32_t>(c, 1);
      if ((dst && !j.m) || !j.tab || !j.mmax) { c->err = "device arena exhausted"; return 1; }
      HIPCK(c, hipMemsetAsync(j.tab, 0, (size_t)MAP_TAB * 4, c->st));
      HIPCK(c, hipMemsetAsync(j.mmax, 0, 4, c->st));
      j.numsamples = fr[f].numsamples;
      j.n = fr[f].n;
      mj.push_back(j);
      mj_frame.push_back(f);
    }
    max_ns = fr[f].numsamples > max_ns ? fr[f].numsamples : max_ns;
    max_n = fr[f].n > max_n ? fr[f].n : max_n;
  }
  if (!mj.empty()) {
    MapJob *dj;
    if (upload(c, mj, &dj)) return 1;
    if (sacopt_launch_map(dj, (int)mj.size(), max_ns, max_n, c->st)) { c->err = "Remap launch failed"; return 1; }
  }
  for (size_t i = 0; i < mj.size(); i++) {
    const int f = mj_frame[i], k = (int)(i & 1);
    int32_t *dst = k ? out[f].map1 : out[f].map0;
    if (dst && dst != mj[i].m && copy_out(c, dst, mj[i].m, (size_t)fr[f].n * 4)) return 1;
    HIPCK(c, hipMemcpyAsync(&mm[2 * f + k], mj[i].mmax, 4, hipMemcpyDeviceToHost, c->st));
  }
  for (int f = 0; f < nframes; f++) {
    const StageBuf &b = chains[f].sb[last];
    const size_t bytes = (size_t)fr[f].n * 4;
    if (copy_out(c, out[f].err0, b.e[0], bytes) || copy_out(c, out[f].err1, b.e[1], bytes) ||
        copy_out(c, out[f].s2u0, b.u[0], bytes) || copy_out(c, out[f].s2u1, b.u[1], bytes))
      return 1;
    HIPCK(c, hipMemcpyAsync(&um[2 * f], b.umax, 8, hipMemcpyDeviceToHost, c->st));
  }
  if (end_timing(c)) return 1;
  for (int f = 0; f < nframes; f++) {
    out[f].maxbpn[0] = ilog2(um[2 * f]);  // codec.cpp:207-208
    out[f].maxbpn[1] = ilog2(um[2 * f + 1]);
    const bool maps = out[f].map0 || out[f].map1;
    out[f].maxbpn[2] = maps ? ilog2(mm[2 * f]) : -1;  // codec.cpp:210-211 (only when map outputs are requested)
    out[f].maxbpn[3] = maps ? ilog2(mm[2 * f + 1]) : -1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ PredictStereoFrame
extern "C" int sacopt_predict_stereo_frames(sacopt_ctx *c, int profile, int nframes, const sacopt_frame *frames,
                                            const int32_t *from, const int32_t *n, const double *coefs,
                                            sacopt_residuals *out) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  ProfDesc pd;
  if (!get_prof(profile, &pd)) { c->err = "unknown profile"; return 1; }
  if (nframes < 0 || (nframes && (!frames || !out))) { c->err = "bad arguments"; return 1; }
  if (check_frames(c, nframes, frames, from, n)) return 1;
  c->timed_variant = sacopt_lms_variant(1280, 640, nframes, c->bits, L1C_FRAMES);
  double base[3 * SACOPT_MAX_COEFS];
  sacopt_base_profile(profile, base);
  size_t need = 1 << 20;
  for (int f = 0; f < nframes; f++) {
    const int nn = n ? n[f] : frames[f].numsamples - (from ? from[f] : 0);
    need += chain_bytes(pd, nn > 0 ? nn : 0) + 2 * (((size_t)frames[f].numsamples * 4 + 255) & ~(size_t)255) + 4096 +
            lpc_scratch_bytes(pd, nn > 0 ? nn : 0) + sizeof(LpcJob) + 256 + pipe_bytes(nn > 0 ? nn : 0) +
            ((out[f].map0 || out[f].map1) ? map_bytes(nn > 0 ? nn : 0) : 0);
  }
  if (arena_reserve(c, need)) return 1;
  StreamSetGuard set_guard(c, nframes > BIG_FRAMES ? c->big : nullptr);
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<DevFrame> fr;
  if (stage_frames(c, nframes, frames, from, n, fr)) return 1;
  std::vector<Chain> chains(nframes);
  for (int f = 0; f < nframes; f++) {
    chains[f].frame = f;
    for (int i = 0; i < pd.ncoef; i++) chains[f].coefs[i] = coefs ? coefs[(size_t)f * pd.ncoef + i] : base[3 * i + 2];
  }
  hipEvent_t done = nullptr;
  if (run_final_pass(c, pd, fr, chains, &done)) return 1;
  if (done) HIPCK(c, hipStreamWaitEvent(c->st, done, 0));
  return emit_outputs(c, pd, fr, chains, out);
}

// ------------------------------------------------------------------------------------------ GetCost / Optimize
// Cost of `ncand` candidates per frame, in call order: (Calc(error[0]) + Calc(error[1])) / 2 (GetCost, codec.cpp:349-351) with the
// context's CostFunction. CostGolomb: running mean carried from call to call (d_state[f]); the other two carry nothing.
static int ilog2(int v);
static int cost_round_bufs(sacopt_ctx *c, int nframes, int ncand, const sacopt_ctx::CostBufs &cb, double *d_state, std::vector<double> &costs_host) {
  const size_t nb = 2 * (size_t)nframes * ncand;
  costs_host.assign((size_t)nframes * ncand, 0.);
  if (c->cost_fn == SACOPT_COST_GOLOMB) {
    const int32_t **d_bufs;
    if (upload(c, cb.e, &d_bufs)) return 1;
    double *d_costs = arena_alloc<double>(c, (size_t)nframes * ncand);
    if (!d_costs) { c->err = "device arena exhausted"; return 1; }
    std::vector<CostJob> jobs(nframes);
    for (int f = 0; f < nframes; f++) {
      jobs[f].bufs = d_bufs + 2 * (size_t)f * ncand;
      jobs[f].costs = d_costs + (size_t)f * ncand;
      jobs[f].state = d_state + f;
      jobs[f].n = cb.n[f];
      jobs[f].nbuf = 2 * ncand;
      jobs[f].scr = arena_alloc<double>(c, 6 * (size_t)ncand);
      jobs[f].ok = arena_alloc<int32_t>(c, 1);
      if (!jobs[f].scr || !jobs[f].ok) { c->err = "device arena exhausted"; return 1; }
    }
    CostJob *dj;
    if (upload(c, jobs, &dj)) return 1;
    if (sacopt_launch_cost_par(dj, nframes, 2 * ncand, c->d_pow2, c->st)) { c->err = "k_cost launch failed"; return 1; }
    HIPCK(c, hipMemcpyAsync(costs_host.data(), d_costs, costs_host.size() * 8, hipMemcpyDeviceToHost, c->st));
    HIPCK(c, hipStreamSynchronize(c->st));
    return 0;
  }
  std::vector<int32_t> ns(nb);
  for (size_t i = 0; i < nb; i++) ns[i] = cb.n[i / (2 * (size_t)ncand)];
  std::vector<double> per(nb, 0.);
  if (c->cost_fn == SACOPT_COST_MEANRMS) {  // CostMeanRMS::Calc, codec.cpp:302-312
    const int32_t **d_bufs;
    int32_t *d_ns;
    if (upload(c, cb.e, &d_bufs) || upload(c, ns, &d_ns)) return 1;
    double *d_out = arena_alloc<double>(c, nb);
    if (!d_out) { c->err = "device arena exhausted"; return 1; }
    if (sacopt_launch_meanrms_batch(d_bufs, d_ns, (int)nb, d_out, c->st)) { c->err = "k_meanrms launch failed"; return 1; }
    HIPCK(c, hipMemcpyAsync(per.data(), d_out, nb * 8, hipMemcpyDeviceToHost, c->st));
    HIPCK(c, hipStreamSynchronize(c->st));
  } else {  // bits per sample the host's bit-plane coder spends on the buffer (a fresh coder per buffer, as Encode() has)
    const int32_t **d_um;
    if (upload(c, cb.umax, &d_um)) return 1;
    int32_t *d_mx = arena_alloc<int32_t>(c, nb);
    if (!d_mx) { c->err = "device arena exhausted"; return 1; }
    if (sacopt_launch_gather_i32(d_um, (int)nb, d_mx, c->st)) { c->err = "gather launch failed"; return 1; }
    std::vector<int32_t> mx(nb);
    HIPCK(c, hipMemcpyAsync(mx.data(), d_mx, nb * 4, hipMemcpyDeviceToHost, c->st));
    HIPCK(c, hipStreamSynchronize(c->st));
    std::vector<BpJob> jobs(nb);
    int maxn = 0, maxplanes = 1;
    for (size_t i = 0; i < nb; i++) {
      jobs[i].buf = cb.u[i]; jobs[i].n = ns[i]; jobs[i].maxbpn = ilog2(mx[i]);  // framestats[ch].maxbpn, codec.cpp:207-208
      maxn = ns[i] > maxn ? ns[i] : maxn;
      maxplanes = jobs[i].maxbpn + 1 > maxplanes ? jobs[i].maxbpn + 1 : maxplanes;
    }
    const size_t nblk = (nb + 63) / 64, nctx = (size_t)maxplanes * 32;
    BpJob *dj;
    if (upload(c, jobs, &dj)) return 1;
    int32_t *T = arena_alloc<int32_t>(c, nblk * (size_t)(maxn ? maxn : 1) * 64);
    uint8_t *msb = arena_alloc<uint8_t>(c, nblk * ((size_t)maxn + 16) * 64);
    uint32_t *sse = arena_alloc<uint32_t>(c, nblk * nctx * 66 * 64);
    uint8_t *lb = arena_alloc<uint8_t>(c, nblk * nctx * 64);
    int32_t *dsz = arena_alloc<int32_t>(c, nb);
    if (!T || !msb || !sse || !lb || !dsz) { c->err = "device arena exhausted (bit-plane cost)"; return 1; }
    if (sacopt_launch_bpsize(dj, (int)nb, maxn, maxplanes, T, msb, sse, lb, c->d_bp_fwd, c->d_bp_inv, c->d_bp_sse, c->bp_tscale, c->bp_xscale, dsz, c->st)) { c->err = "k_bpsize launch failed"; return 1; }
    std::vector<int32_t> sz(nb);
    HIPCK(c, hipMemcpyAsync(sz.data(), dsz, nb * 4, hipMemcpyDeviceToHost, c->st));
    HIPCK(c, hipStreamSynchronize(c->st));
    for (size_t i = 0; i < nb; i++) per[i] = ns[i] ? 8. * (double)sz[i] / (double)ns[i] : 0.;
  }
  for (size_t k = 0; k < costs_host.size(); k++) costs_host[k] = (per[2 * k] + per[2 * k + 1]) / 2.;
  return 0;
}
// arena bytes the bit-plane cost of one round needs beyond the chains themselves (nb buffers of up to n samples, 32 planes at most)
static size_t bitplane_cost_bytes(size_t nb, int n, int planes) {
  const size_t nblk = (nb + 63) / 64;
  return nblk * ((size_t)n * 256 + ((size_t)n + 16) * 64 + (size_t)planes * 32 * (66 * 256 + 64)) + nb * 64 + 65536;
}
static void chain_bufs(const ProfDesc &pd, const std::vector<DevFrame> &fr, const std::vector<Chain> &chains, int ncand, sacopt_ctx::CostBufs &cb) {
  const int nframes = (int)fr.size();
  const int last = (pd.type == SACOPT_PROFILE_FAST) ? 0 : pd.nlms + 1;
  const size_t nb = 2 * (size_t)nframes * ncand;
  cb.e.resize(nb); cb.u.resize(nb); cb.umax.resize(nb);
  cb.n.resize(nframes);
  for (int f = 0; f < nframes; f++) {
    cb.n[f] = fr[f].n;
    for (int k = 0; k < ncand; k++) {
      const Chain &ch = chains[(size_t)f * ncand + k];
      for (int h = 0; h < 2; h++) {
        const size_t i = 2 * ((size_t)f * ncand + k) + h;
        cb.e[i] = ch.sb[last].e[h]; cb.u[i] = ch.sb[last].u[h]; cb.umax[i] = ch.sb[last].umax + h;
      }
    }
  }
}
static int cost_round(sacopt_ctx *c, const ProfDesc &pd, const std::vector<DevFrame> &fr, const std::vector<Chain> &chains,
                      int ncand, double *d_state, std::vector<double> &costs_host) {
  sacopt_ctx::CostBufs cb;
  chain_bufs(pd, fr, chains, ncand, cb);
  return cost_round_bufs(c, (int)fr.size(), ncand, cb, d_state, costs_host);
}

extern "C" int sacopt_get_costs(sacopt_ctx *c, int profile, int nframes, const sacopt_frame *frames, const int32_t *from,
                                const int32_t *n, int ncand, const double *cand_coefs, double *costs, double *cost_state) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  ProfDesc pd;
  if (!get_prof(profile, &pd)) { c->err = "unknown profile"; return 1; }
  if (nframes < 0 || ncand < 0 || (nframes && ncand && (!frames || !cand_coefs || !costs))) { c->err = "bad arguments"; return 1; }
  if (!nframes || !ncand) return 0;
  if (check_frames(c, nframes, frames, from, n)) return 1;
  size_t need = 1 << 20;
  for (int f = 0; f < nframes; f++) {
    const int nn = n ? n[f] : frames[f].numsamples - (from ? from[f] : 0);
    need += (size_t)ncand * chain_bytes(pd, nn > 0 ? nn : 0) + 2 * (((size_t)frames[f].numsamples * 4 + 255) & ~(size_t)255) + 4096 +
            (size_t)ncand * (64 + lpc_scratch_bytes(pd, nn > 0 ? nn : 0) + sizeof(LpcJob) + 256) +
            2 * ((size_t)ncand * (16 + 8 + 48) + sizeof(CostJob) + 2048);  // cost tables, twice: one more set for sacopt_recost_last
  }
  if (c->cost_fn == SACOPT_COST_BITPLANE) {
    int mxn = 0;
    for (int f = 0; f < nframes; f++) { const int nn = n ? n[f] : frames[f].numsamples - (from ? from[f] : 0); mxn = nn > mxn ? nn : mxn; }
    need += 2 * bitplane_cost_bytes(2 * (size_t)nframes * ncand, mxn, 32);
  }
  if (arena_reserve(c, need)) return 1;
  StreamSetGuard set_guard(c, (long long)nframes * ncand > 256 ? c->big : nullptr);
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<DevFrame> fr;
  if (stage_frames(c, nframes, frames, from, n, fr)) return 1;
  std::vector<Chain> chains((size_t)nframes * ncand);
  for (int f = 0; f < nframes; f++)
    for (int k = 0; k < ncand; k++) {
      Chain &ch = chains[(size_t)f * ncand + k];
      ch.frame = f;
      memcpy(ch.coefs, cand_coefs + ((size_t)f * ncand + k) * pd.ncoef, sizeof(double) * pd.ncoef);
    }
  Memo memo;
  if (eval_chains(c, pd, fr, chains, memo)) return 1;
  std::vector<double> st(nframes, 0.);
  if (cost_state) st.assign(cost_state, cost_state + nframes);
  double *d_state;
  if (upload(c, st, &d_state)) return 1;
  std::vector<double> ch_costs;
  if (cost_round(c, pd, fr, chains, ncand, d_state, ch_costs)) return 1;
  memcpy(costs, ch_costs.data(), ch_costs.size() * 8);
  if (cost_state) HIPCK(c, hipMemcpyAsync(cost_state, d_state, (size_t)nframes * 8, hipMemcpyDeviceToHost, c->st));
  if (end_timing(c)) return 1;
  // the residuals stay where they are until the next call: sacopt_recost_last can cost them again from another state
  chain_bufs(pd, fr, chains, ncand, c->kept.b);
  c->kept.nframes = nframes; c->kept.ncand = ncand; c->kept.mark = c->used; c->kept.valid = true;
  return 0;
}

// The candidates of the last sacopt_get_costs call, costed again from another CostGolomb start state: the cascade is not
// re-run, the residuals never left the device. (Multi-GPU candidate split, SURVEY.md 8e: a rank first costs its slice from a
// guessed state and later from the state its predecessor really ended with.)
extern "C" int sacopt_recost_last(sacopt_ctx *c, double *costs, double *cost_state) {
  if (!c) return 1;
  HIPCK(c, hipSetDevice(c->device));
  c->err.clear();
  if (!c->kept.valid) { c->err = "sacopt_recost_last: no sacopt_get_costs call to re-cost (any other call on the context discards its residuals)"; return 1; }
  if (!costs) { c->err = "bad arguments"; return 1; }
  HIPCK(c, hipStreamSynchronize(c->st));
  c->used = c->kept.mark;  // scratch of an earlier recost is reused
  const int nframes = c->kept.nframes;
  std::vector<double> st(nframes, 0.);
  if (cost_state) st.assign(cost_state, cost_state + nframes);
  double *d_state;
  if (upload(c, st, &d_state)) return 1;
  std::vector<double> ch_costs;
  if (cost_round_bufs(c, nframes, c->kept.ncand, c->kept.b, d_state, ch_costs)) return 1;
  memcpy(costs, ch_costs.data(), ch_costs.size() * 8);
  if (cost_state) {
    HIPCK(c, hipMemcpyAsync(cost_state, d_state, (size_t)nframes * 8, hipMemcpyDeviceToHost, c->st));
    HIPCK(c, hipStreamSynchronize(c->st));
  }
  return 0;
}

// candidate grids, generated exactly as the reference loop does (codec.cpp:392-404)
static int build_grid(sacopt_ctx *c, const double *prof, int ncoef_loop, std::vector<std::vector<double>> &grid, size_t *total_cand) {
  grid.assign(ncoef_loop, std::vector<double>());
  *total_cand = 1;
  for (int i = 0; i < ncoef_loop; i++) {
    double x = prof[3 * i];
    const double xinc = (prof[3 * i + 1] - prof[3 * i]) / 10.;
    while (x <= prof[3 * i + 1]) {
      grid[i].push_back(x);
      x += xinc;
      if (grid[i].size() > 64) { c->err = "degenerate coefficient range (grid does not terminate)"; return 1; }
    }
    *total_cand += grid[i].size();
  }
  return 0;
}
// arena bytes Optimize needs for one frame with an n-sample window
static size_t optimize_bytes(const ProfDesc &pd, int ncoef_loop, const std::vector<std::vector<double>> &grid, size_t total_cand, int wn) {
  size_t lpc_jobs = 1;  // most LPC jobs of this frame alive in one round
  for (int i = 0; i < ncoef_loop; i++)
    if (pd.nprefix[0] > i && grid[i].size() > lpc_jobs) lpc_jobs = grid[i].size();
  lpc_jobs += 1;  // the baseline chain rides along with the first round
  return total_cand * (chain_bytes(pd, wn) + 8192 + sizeof(LpcJob)) + 32768 + lpc_jobs * (lpc_scratch_bytes(pd, wn) + 256);
}

// The search itself, on frames already staged (fr[f].from/n = optimisation window), all on c->st.
// `after_first` (optional) runs once, right after the first round's kernels are enqueued and before the host waits for them.
static int optimize_core(sacopt_ctx *c, const ProfDesc &pd, const std::vector<DevFrame> &fr, const double *prof, int ncoef_loop,
                         const std::vector<std::vector<double>> &grid, double *best, double *trace, int max_rows, int32_t *rows_out,
                         const std::function<int()> &after_first = nullptr,
                         const std::function<int(int, const std::vector<std::vector<double>> &)> &after_round = nullptr) {
  const int nframes = (int)fr.size();
  std::vector<double> st(nframes, 0.);  // one CostGolomb instance per Optimize call (codec.cpp:364)
  double *d_state;
  if (upload(c, st, &d_state)) return 1;
  std::vector<std::vector<double>> cur(nframes, std::vector<double>(pd.ncoef));
  for (int f = 0; f < nframes; f++)
    for (int i = 0; i < pd.ncoef; i++) cur[f][i] = prof[3 * i + 2];
  std::vector<double> best_cost(nframes);
  std::vector<int> rows(nframes, 0);
  Memo memo;
  memo.reserve((size_t)nframes * 96 * 6);  // every candidate's stage outputs stay memoised for the whole search
  auto add_row = [&](int f, double coef, double x, double cost) {
    if (trace && rows[f] < max_rows) {
      double *r = trace + ((size_t)f * max_rows + rows[f]) * 3;
      r[0] = coef; r[1] = x; r[2] = cost;
    }
    rows[f]++;
  };
  // The baseline call GetCost(profile, &CostFunc, 0, coefs[0].vdef, ...) (codec.cpp:386) does not depend on
  // anything the search decides, so it is evaluated together with the first round's candidates
  // (first in CostGolomb call order, exactly where the reference makes it).
  bool base_done = false;
  for (int i = 0; i <= ncoef_loop; i++) {
    const int nc = i < ncoef_loop ? (int)grid[i].size() : 0;
    if (i < ncoef_loop && !nc) continue;
    if (i == ncoef_loop && base_done) break;  // i == ncoef_loop: no round at all -> baseline alone
    const int nb = base_done ? 0 : 1, tot = nc + nb;
    std::vector<Chain> chains((size_t)nframes * tot);
    for (int f = 0; f < nframes; f++)
      for (int k = 0; k < tot; k++) {
        Chain &ch = chains[(size_t)f * tot + k];
        ch.frame = f;
        memcpy(ch.coefs, cur[f].data(), 8 * pd.ncoef);
        if (k >= nb) ch.coefs[i] = grid[i][k - nb];
      }
    HP_MARK(hpa);
    if (eval_chains(c, pd, fr, chains, memo)) return 1;
    if (!base_done && after_first && after_first()) return 1;
    std::vector<double> costs;
    HP_MARK(hpb);
    if (cost_round(c, pd, fr, chains, tot, d_state, costs)) return 1;
    HP_MARK(hpc);
    HP_PRINT("round %d: eval_chains (+ full-frame enqueue) %.2f ms, cost_round incl. wait %.2f ms\n", i, hpb - hpa, hpc - hpb);
    for (int f = 0; f < nframes; f++) {
      if (nb) { best_cost[f] = costs[(size_t)f * tot]; add_row(f, -1, cur[f][0], best_cost[f]); }
      if (!nc) continue;
      double best_x = cur[f][i];
      for (int k = 0; k < nc; k++) {
        const double cost = costs[(size_t)f * tot + nb + k];
        add_row(f, i, grid[i][k], cost);
        if (cost < best_cost[f]) { best_x = grid[i][k]; best_cost[f] = cost; }  // strict <: earliest wins ties (codec.cpp:399)
      }
      cur[f][i] = best_x;  // later coefficients see earlier winners (codec.cpp:406)
    }
    base_done = true;
    if (after_round && nc && after_round(i, cur)) return 1;  // coefficients 0..i are final from here on
  }
  for (int f = 0; f < nframes; f++) {
    memcpy(best + (size_t)f * pd.ncoef, cur[f].data(), 8 * pd.ncoef);
    if (rows_out) rows_out[f] = rows[f];
  }
  return 0;
}

extern "C" int sacopt_optimize_frames(sacopt_ctx *c, int profile, int nframes, const sacopt_frame *frames, const double *prof_in,
                                      int ncoef_loop, double *best, double *trace, int max_rows, int32_t *rows_out) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  ProfDesc pd;
  if (!get_prof(profile, &pd)) { c->err = "unknown profile"; return 1; }
  if (nframes < 0 || (nframes && (!frames || !best))) { c->err = "bad arguments"; return 1; }
  if (ncoef_loop < 0 || ncoef_loop > pd.ncoef) { c->err = "ncoef_loop exceeds the profile's coefficient count (the reference would index past coefs[], codec.cpp:389)"; return 1; }
  if (!nframes) return 0;
  if (check_frames(c, nframes, frames, nullptr, nullptr)) return 1;
  double prof[3 * SACOPT_MAX_COEFS];
  sacopt_base_profile(profile, prof);
  if (prof_in) memcpy(prof, prof_in, sizeof(double) * 3 * pd.ncoef);
  std::vector<std::vector<double>> grid;
  size_t total_cand;
  if (build_grid(c, prof, ncoef_loop, grid, &total_cand)) return 1;
  std::vector<int32_t> from(nframes), wn(nframes);  // windows (codec.cpp:356-359)
  size_t need = 1 << 20;
  for (int f = 0; f < nframes; f++) {
    wn[f] = frames[f].numsamples / 8;
    from[f] = (frames[f].numsamples - wn[f]) / 2;
    need += optimize_bytes(pd, ncoef_loop, grid, total_cand, wn[f]) + 2 * (((size_t)frames[f].numsamples * 4 + 255) & ~(size_t)255);
  }
  if (c->cost_fn == SACOPT_COST_BITPLANE) {  // scratch of the bit-plane cost, once per round (the arena only grows within a call)
    int mxn = 0;
    for (int f = 0; f < nframes; f++) mxn = wn[f] > mxn ? wn[f] : mxn;
    need += (size_t)(ncoef_loop + 1) * bitplane_cost_bytes(2 * (size_t)nframes * 13, mxn, 32)  /* <= 12 candidates + the baseline per round */;
  }
  if (arena_reserve(c, need)) return 1;
  StreamSetGuard set_guard(c, (long long)nframes * 11 > 256 ? c->big : nullptr);
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<DevFrame> fr;
  if (stage_frames(c, nframes, frames, from.data(), wn.data(), fr)) return 1;
  if (optimize_core(c, pd, fr, prof, ncoef_loop, grid, best, trace, max_rows, rows_out)) return 1;
  return end_timing(c);
}

// ------------------------------------------------------------------------------------------ FrameCoder::Predict
// FrameCoder::Predict (src/codec.cpp:442-461) with the optimizer call of :455-456 enabled:
// Optimize(optprofile) on the centre window and PredictStereoFrame(baseprofile,0,1,0,numsamples)
// are independent ("ref+opt": the frame is coded with the base profile), so the full-frame pass
// is enqueued on the pipeline streams first and runs underneath the optimizer rounds.
extern "C" int sacopt_encode_frames(sacopt_ctx *c, int profile, int nframes, const sacopt_frame *frames, const double *prof_in,
                                    int ncoef_loop, double *best, sacopt_residuals *out) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  ProfDesc pd;
  if (!get_prof(profile, &pd)) { c->err = "unknown profile"; return 1; }
  if (nframes < 0 || (nframes && (!frames || !out || (ncoef_loop > 0 && !best)))) { c->err = "bad arguments"; return 1; }
  if (ncoef_loop < 0 || ncoef_loop > pd.ncoef) { c->err = "ncoef_loop exceeds the profile's coefficient count (the reference would index past coefs[], codec.cpp:389)"; return 1; }
  if (!nframes) return 0;
  if (check_frames(c, nframes, frames, nullptr, nullptr)) return 1;
  double prof[3 * SACOPT_MAX_COEFS];
  sacopt_base_profile(profile, prof);
  if (prof_in) memcpy(prof, prof_in, sizeof(double) * 3 * pd.ncoef);
  c->timed_variant = sacopt_lms_variant(1280, 640, nframes, c->bits, L1C_FRAMES);
  std::vector<std::vector<double>> grid;
  size_t total_cand = 0;
  if (ncoef_loop > 0 && build_grid(c, prof, ncoef_loop, grid, &total_cand)) return 1;
  size_t need = 1 << 20;
  for (int f = 0; f < nframes; f++) {
    const int ns = frames[f].numsamples;
    need += chain_bytes(pd, ns) + 2 * (((size_t)ns * 4 + 255) & ~(size_t)255) + 4096 + lpc_scratch_bytes(pd, ns) + sizeof(LpcJob) + 256 + pipe_bytes(ns) +
            ((out[f].map0 || out[f].map1) ? map_bytes(ns) : 0);
    if (ncoef_loop > 0) need += optimize_bytes(pd, ncoef_loop, grid, total_cand, ns / 8);
  }
  if (ncoef_loop > 0 && c->cost_fn == SACOPT_COST_BITPLANE) {
    int mxn = 0;
    for (int f = 0; f < nframes; f++) mxn = frames[f].numsamples / 8 > mxn ? frames[f].numsamples / 8 : mxn;
    need += (size_t)(ncoef_loop + 1) * bitplane_cost_bytes(2 * (size_t)nframes * 13, mxn, 32)  /* <= 12 candidates + the baseline per round */;
  }
  if (arena_reserve(c, need)) return 1;
  // small batches: the stream set that keeps the full-frame followers off the optimizer's CUs; large ones: every stream on
  // every CU (see sacopt_create)
  bool use_split = c->split[0] && ncoef_loop > 0 && nframes <= 16 && pd.nlms == 4;
#ifdef SACOPT_ABLATE
  if (getenv("SACOPT_NOSPLIT")) use_split = false;
#endif
  StreamSetGuard set_guard(c, use_split ? c->split : (nframes > BIG_FRAMES ? c->big : nullptr));
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<DevFrame> fr;
  if (stage_frames(c, nframes, frames, nullptr, nullptr, fr)) return 1;
  // full-frame pass with the base profile (codec.cpp:459)
  std::vector<Chain> chains(nframes);
  for (int f = 0; f < nframes; f++) {
    chains[f].frame = f;
    for (int i = 0; i < pd.ncoef; i++) chains[f].coefs[i] = prof[3 * i + 2];
  }
  hipEvent_t done = nullptr;
  // The optimizer is the longer of the two and its first round needs ~20 ms of LPC work before its lead stage starts,
  // the full-frame pipeline ~15 ms of host time to enqueue: the optimizer's first round goes first, the pipeline is
  // enqueued while it runs (it stays off c->st: see final_pass_enqueue).
  const bool opt_first = ncoef_loop > 0 && final_pass_pipelined(pd, fr, chains);
  hipEvent_t staged = nullptr;
  if (opt_first) {
    TAKE_EVENT(c, staged, false);
    HIPCK(c, hipEventRecord(staged, c->st));
  } else if (run_final_pass(c, pd, fr, chains, &done)) return 1;
  if (ncoef_loop > 0) {  // Optimize on the centre window (codec.cpp:356-359)
    std::vector<DevFrame> win = fr;
    for (DevFrame &w : win) { w.n = w.numsamples / 8; w.from = (w.numsamples - w.n) / 2; }
    bool ran = false;
    auto enqueue_final = [&]() -> int { ran = true; return run_final_pass(c, pd, fr, chains, &done, staged); };
    if (optimize_core(c, pd, win, prof, ncoef_loop, grid, best, nullptr, 0, nullptr,
                      opt_first ? std::function<int()>(enqueue_final) : std::function<int()>())) return 1;
    if (opt_first && !ran && run_final_pass(c, pd, fr, chains, &done, staged)) return 1;  // no round ran (empty grids)
  }
  if (done) HIPCK(c, hipStreamWaitEvent(c->st, done, 0));
  return emit_outputs(c, pd, fr, chains, out);
}

// SURVEY 8f.2 as ONE call: Optimize(optprofile), then PredictStereoFrame(optprofile, 0, 1, 0, numsamples) -- the frame is coded
// with the selected parameters, so the full-frame pass depends on the optimizer. It still does not have to wait for the last
// round: the parts of the pass start as soon as their parameters are final (FinalPipe).
extern "C" int sacopt_encode_frames_selected(sacopt_ctx *c, int profile, int nframes, const sacopt_frame *frames, const double *prof_in,
                                             int ncoef_loop, double *best, sacopt_residuals *out) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  ProfDesc pd;
  if (!get_prof(profile, &pd)) { c->err = "unknown profile"; return 1; }
  if (nframes < 0 || (nframes && (!frames || !out || !best))) { c->err = "bad arguments"; return 1; }
  if (ncoef_loop <= 0 || ncoef_loop > pd.ncoef) { c->err = "ncoef_loop must be 1..the profile's coefficient count (the reference would index past coefs[], codec.cpp:389)"; return 1; }
  if (!nframes) return 0;
  if (check_frames(c, nframes, frames, nullptr, nullptr)) return 1;
  double prof[3 * SACOPT_MAX_COEFS];
  sacopt_base_profile(profile, prof);
  if (prof_in) memcpy(prof, prof_in, sizeof(double) * 3 * pd.ncoef);
  c->timed_variant = sacopt_lms_variant(1280, 640, nframes, c->bits, L1C_FRAMES);
  std::vector<std::vector<double>> grid;
  size_t total_cand = 0;
  if (build_grid(c, prof, ncoef_loop, grid, &total_cand)) return 1;
  size_t need = 1 << 20;
  for (int f = 0; f < nframes; f++) {
    const int ns = frames[f].numsamples;
    need += chain_bytes(pd, ns) + 2 * (((size_t)ns * 4 + 255) & ~(size_t)255) + 4096 + lpc_scratch_bytes(pd, ns) + sizeof(LpcJob) + 256 + pipe_bytes(ns) +
            ((out[f].map0 || out[f].map1) ? map_bytes(ns) : 0);
    need += optimize_bytes(pd, ncoef_loop, grid, total_cand, ns / 8);
  }
  if (c->cost_fn == SACOPT_COST_BITPLANE) {
    int mxn = 0;
    for (int f = 0; f < nframes; f++) mxn = frames[f].numsamples / 8 > mxn ? frames[f].numsamples / 8 : mxn;
    need += (size_t)(ncoef_loop + 1) * bitplane_cost_bytes(2 * (size_t)nframes * 13, mxn, 32);
  }
  if (arena_reserve(c, need)) return 1;
  const bool use_split = c->split[0] && nframes <= 16 && pd.nlms == 4;
  StreamSetGuard set_guard(c, use_split ? c->split : (nframes > BIG_FRAMES ? c->big : nullptr));
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<DevFrame> fr;
  if (stage_frames(c, nframes, frames, nullptr, nullptr, fr)) return 1;
  std::vector<Chain> chains(nframes);
  for (int f = 0; f < nframes; f++) {
    chains[f].frame = f;
    for (int i = 0; i < pd.ncoef; i++) chains[f].coefs[i] = prof[3 * i + 2];
  }
  std::vector<DevFrame> win = fr;
  for (DevFrame &w : win) { w.n = w.numsamples / 8; w.from = (w.numsamples - w.n) / 2; }
  hipEvent_t done = nullptr;
  if (!final_pass_pipelined(pd, fr, chains)) {  // short frames / the fast profile: nothing to overlap
    if (optimize_core(c, pd, win, prof, ncoef_loop, grid, best, nullptr, 0, nullptr)) return 1;
    for (int f = 0; f < nframes; f++) memcpy(chains[f].coefs, best + (size_t)f * pd.ncoef, 8 * pd.ncoef);
    if (run_final_pass(c, pd, fr, chains, &done)) return 1;
  } else {
    hipEvent_t staged;
    TAKE_EVENT(c, staged, false);
    HIPCK(c, hipEventRecord(staged, c->st));
    FinalPipe fp;
    if (fp.prepare(c, pd, fr, chains, staged)) return 1;
    auto after_round = [&](int i, const std::vector<std::vector<double>> &cur) -> int {
      for (int g = 0; g < 4; g++)
        if (!fp.queued[g] && fp.needs(g, ncoef_loop) <= i) {
          for (int f = 0; f < nframes; f++) memcpy(chains[f].coefs, cur[f].data(), 8 * pd.ncoef);
          if (fp.enqueue(g)) return 1;
        } else if (!fp.queued[g]) break;  // in order
      return 0;
    };
    if (optimize_core(c, pd, win, prof, ncoef_loop, grid, best, nullptr, 0, nullptr, nullptr, after_round)) return 1;
    for (int f = 0; f < nframes; f++) memcpy(chains[f].coefs, best + (size_t)f * pd.ncoef, 8 * pd.ncoef);
    for (int g = 0; g < 4; g++) if (fp.enqueue(g)) return 1;  // whatever no round released (rounds without candidates)
    done = fp.done;
  }
  if (done) HIPCK(c, hipStreamWaitEvent(c->st, done, 0));
  return emit_outputs(c, pd, fr, chains, out);
}

// ------------------------------------------------------------------------------------------ UnpredictStereoFrame
// Decoder's used-value tables (Remap::usedl/usedh, src/coder/map.h) -> bitmap over the values -32768..32768 (bit v+32768;
// value 0 counts as used, map.cpp:138) followed by the number of set bits in the words before each word.
static void build_unmap_table(const uint8_t *used, uint32_t *out) {
  memset(out, 0, 2 * UN_MAPW * sizeof(uint32_t));
  const uint8_t *usedl = used, *usedh = used + 32769;
  auto set = [&](int v) { const unsigned x = (unsigned)(v + 32768); out[x >> 5] |= 1u << (x & 31); };
  set(0);
  for (int v = 1; v <= 32768; v++) {
    if (usedh[v]) set(v);
    if (usedl[v]) set(-v);
  }
  uint32_t acc = 0;
  for (int w = 0; w < UN_MAPW; w++) { out[UN_MAPW + w] = acc; acc += (uint32_t)__builtin_popcount(out[w]); }
}

extern "C" int sacopt_unpredict_stereo_frames(sacopt_ctx *c, int profile, int nframes, const sacopt_dec_frame *frames,
                                              const double *coefs) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  ProfDesc pd;
  if (!get_prof(profile, &pd)) { c->err = "unknown profile"; return 1; }
  if (nframes < 0 || (nframes && !frames)) { c->err = "bad arguments"; return 1; }
  if (!nframes) return 0;
  double base[3 * SACOPT_MAX_COEFS];
  sacopt_base_profile(profile, base);
  size_t need = 1 << 20;
  for (int f = 0; f < nframes; f++) {
    const sacopt_dec_frame &d = frames[f];
    if (d.numsamples < 0 || (d.numsamples && (!d.e0 || !d.e1 || !d.v0 || !d.v1))) { c->err = "bad frame"; return 1; }
    need += 4 * (((size_t)d.numsamples * 4 + 255) & ~(size_t)255) + 2 * (2 * UN_MAPW * 4 + 256) + sizeof(UnJob) + 1024;
  }
  if (arena_reserve(c, need)) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<UnJob> jobs(nframes);
  std::vector<std::vector<uint32_t>> tabs;  // pageable staging for the async uploads: alive until the stream is synchronised
  tabs.reserve(2 * (size_t)nframes);
  for (int f = 0; f < nframes; f++) {
    const sacopt_dec_frame &d = frames[f];
    UnJob &j = jobs[f];
    memset(&j, 0, sizeof(j));
    const int n = d.numsamples;
    const int32_t *src[2] = {d.e0, d.e1};
    int32_t *dst[2] = {d.v0, d.v1};
    const uint8_t *used[2] = {d.used0, d.used1};
    for (int k = 0; k < 2; k++) {
      if (n && to_dev(c, src[k], n, &j.e[k])) return 1;
      j.v[k] = (n && is_device_ptr(dst[k])) ? dst[k] : arena_alloc<int32_t>(c, n ? n : 1);
      if (!j.v[k]) { c->err = "device arena exhausted"; return 1; }
      if (used[k] && profile != SACOPT_PROFILE_FAST) {  // StereoFast::Unpredict never unmaps (profile.cpp:34-43)
        tabs.emplace_back(2 * UN_MAPW);
        build_unmap_table(used[k], tabs.back().data());
        uint32_t *dm = arena_alloc<uint32_t>(c, 2 * UN_MAPW);
        if (!dm) { c->err = "device arena exhausted"; return 1; }
        if (upload_bytes(c, dm, tabs.back().data(), 2 * UN_MAPW * 4, c->st)) return 1;
        j.map[k] = dm;
      }
      j.minval[k] = d.minval[k];
      j.maxval[k] = d.maxval[k];
    }
    j.n = n;
    for (int i = 0; i < pd.ncoef; i++) j.coef[i] = coefs ? coefs[(size_t)f * pd.ncoef + i] : base[3 * i + 2];
  }
  UnJob *dj;
  if (upload(c, jobs, &dj)) return 1;
  hipEvent_t a, b;
  TAKE_EVENT(c, a, true);
  TAKE_EVENT(c, b, true);
  HIPCK(c, hipEventRecord(a, c->st));
  // one workgroup per frame, a CU each: launch on the stream that may use every CU (c->st leaves out the CUs reserved for
  // the encoder's lead stage), ordered behind the uploads and in front of the copies by events
  hipStream_t ks = c->sAall ? c->sAall : c->st;
  if (ks != c->st) HIPCK(c, hipStreamWaitEvent(ks, a, 0));
  if (sacopt_launch_unpredict(profile, dj, nframes, c->d_tab, c->bits, ks)) { c->err = "k_unpredict launch failed"; return 1; }
  HIPCK(c, hipEventRecord(b, ks));
  if (ks != c->st) HIPCK(c, hipStreamWaitEvent(c->st, b, 0));
  c->lms_ev.push_back({a, b});
  for (int f = 0; f < nframes; f++) {
    const size_t bytes = (size_t)frames[f].numsamples * 4;
    c->lms_units += frames[f].numsamples;
    if (bytes && jobs[f].v[0] != frames[f].v0 && copy_out(c, frames[f].v0, jobs[f].v[0], bytes)) return 1;
    if (bytes && jobs[f].v[1] != frames[f].v1 && copy_out(c, frames[f].v1, jobs[f].v[1], bytes)) return 1;
  }
  return end_timing(c);
}

// ------------------------------------------------------------------------------------------ cost on caller residuals
extern "C" int sacopt_cost_golomb(sacopt_ctx *c, const int32_t *buf, int n, double *state_inout, double *cost) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (n < 0 || !cost || (n && !buf)) { c->err = "bad arguments"; return 1; }
  if (n == 0) { *cost = 0.; return 0; }  // codec.cpp:335: empty buffer -> 0, state untouched
  if (arena_reserve(c, (size_t)n * 4 + (1 << 16))) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  const int32_t *d = buf;
  if (!is_device_ptr(buf)) {
    int32_t *p = arena_alloc<int32_t>(c, n);
    HIPCK(c, hipMemcpyAsync(p, buf, (size_t)n * 4, hipMemcpyHostToDevice, c->st));
    d = p;
  }
  // single-buffer mode of k_cost (nbuf < 0): cost = Calc(buf), state carried
  std::vector<const int32_t *> bufs = {d, d};
  const int32_t **d_bufs;
  if (upload(c, bufs, &d_bufs)) return 1;
  std::vector<double> stv = {state_inout ? *state_inout : 0.};
  double *d_state;
  if (upload(c, stv, &d_state)) return 1;
  double *d_cost = arena_alloc<double>(c, 2);
  CostJob j;
  j.bufs = d_bufs; j.costs = d_cost; j.state = d_state; j.n = n; j.nbuf = -1; j.scr = nullptr; j.ok = nullptr;  // nbuf < 0: single-buffer mode
  std::vector<CostJob> jobs = {j};
  CostJob *dj;
  if (upload(c, jobs, &dj)) return 1;
  if (sacopt_launch_cost(dj, 1, c->d_pow2, c->st)) { c->err = "k_cost launch failed"; return 1; }
  HIPCK(c, hipMemcpyAsync(cost, d_cost, 8, hipMemcpyDeviceToHost, c->st));
  if (state_inout) HIPCK(c, hipMemcpyAsync(state_inout, d_state, 8, hipMemcpyDeviceToHost, c->st));
  return end_timing(c);
}

// CostGolomb::Calc that allocates nothing (include/sacopt.h): staging + partial results live in the per-GPU static scratch.
// No hipMalloc/hipFree/hipHostMalloc, no event creation, no arena, no std::vector: pointer checks, at most three copies,
// one memset and three kernel launches on c->st, then a stream synchronise.
extern "C" int sacopt_cost_golomb_noalloc(sacopt_ctx *c, const int32_t *buf, int n, double *state_inout, double *cost) {
  if (!c) return 1;
  HIPCK(c, hipSetDevice(c->device));
  c->err.clear();
  if (n < 0 || !cost || (n && !buf)) { c->err = "bad arguments"; return 1; }
  if (n > c->na_max_n) { c->err = "sacopt_cost_golomb_noalloc: n exceeds the static scratch (at most 1048576 samples per call)"; return 1; }
  if (n == 0) { *cost = 0.; return 0; }  // codec.cpp:335: empty buffer -> 0, state untouched
  if (c->device < 0 || c->device >= 64) { c->err = "sacopt_cost_golomb_noalloc: device index beyond the scratch table"; return 1; }
  if (g_na_busy[c->device].test_and_set(std::memory_order_acquire)) { c->err = "sacopt_cost_golomb_noalloc already running on this GPU (all contexts of a GPU share its static scratch: serialise the calls)"; return 1; }
  struct Release { int d; ~Release() { g_na_busy[d].clear(std::memory_order_release); } } release{c->device};
  HIPCK(c, hipStreamSynchronize(c->st));  // an earlier call that failed half-way may have left work on the stream
  double *ctl = reinterpret_cast<double *>(c->na_scratch + c->na_stage_bytes);
  const int32_t *d = buf;
  if (!is_device_ptr(buf)) {
    HIPCK(c, hipMemcpyAsync(c->na_scratch, buf, (size_t)n * 4, hipMemcpyHostToDevice, c->st));
    d = reinterpret_cast<const int32_t *>(c->na_scratch);
  }
  // the scratch keeps the last call's contents: the control block starts from zero, then the caller's state goes in
  HIPCK(c, hipMemsetAsync(ctl, 0, c->na_ctl_doubles * 8, c->st));
  const double st0 = state_inout ? *state_inout : 0.;
  HIPCK(c, hipMemcpyAsync(ctl, &st0, 8, hipMemcpyHostToDevice, c->st));
  if (sacopt_launch_cost_noalloc(d, n, ctl, c->d_pow2, c->st)) { c->err = "k_cseg launch failed"; return 1; }
  double res[2];
  HIPCK(c, hipMemcpyAsync(res, ctl, 16, hipMemcpyDeviceToHost, c->st));
  HIPCK(c, hipStreamSynchronize(c->st));
  if (state_inout) *state_inout = res[0];
  *cost = res[1];
  return 0;
}

// ------------------------------------------------------------------------------------------ bit-plane coder size
extern "C" int sacopt_coded_sizes(sacopt_ctx *c, int nbuf, const int32_t *const *s2u, const int32_t *n, const int32_t *maxbpn, int32_t *sizes) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (nbuf < 0 || (nbuf && (!s2u || !n || !maxbpn || !sizes))) { c->err = "bad arguments"; return 1; }
  if (!nbuf) return 0;
  int maxn = 0, maxplanes = 1;
  size_t host_bytes = 0;
  for (int b = 0; b < nbuf; b++) {
    if (n[b] < 0 || (n[b] && !s2u[b])) { c->err = "bad buffer"; return 1; }
    if (maxbpn[b] < 0 || maxbpn[b] > 30) { c->err = "maxbpn out of range (0..30)"; return 1; }
    maxn = n[b] > maxn ? n[b] : maxn;
    maxplanes = maxbpn[b] + 1 > maxplanes ? maxbpn[b] + 1 : maxplanes;
    host_bytes += ((size_t)n[b] * 4 + 255) & ~(size_t)255;
  }
  const size_t nblk = (nbuf + 63) / 64, nctx = (size_t)maxplanes * 32;
  const size_t need = host_bytes + nblk * ((size_t)maxn * 256 + ((size_t)maxn + 16) * 64 + nctx * 66 * 256 + nctx * 64) + (size_t)nbuf * (sizeof(BpJob) + 8) + (1 << 16);
  if (arena_reserve(c, need)) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<BpJob> jobs(nbuf);
  for (int b = 0; b < nbuf; b++) {
    jobs[b].n = n[b]; jobs[b].maxbpn = maxbpn[b]; jobs[b].buf = nullptr;
    if (n[b] && to_dev(c, s2u[b], (size_t)n[b], &jobs[b].buf)) return 1;
  }
  BpJob *dj;
  if (upload(c, jobs, &dj)) return 1;
  int32_t *T = arena_alloc<int32_t>(c, nblk * (size_t)(maxn ? maxn : 1) * 64);
  uint8_t *msb = arena_alloc<uint8_t>(c, nblk * ((size_t)maxn + 16) * 64);
  uint32_t *sse = arena_alloc<uint32_t>(c, nblk * nctx * 66 * 64);
  uint8_t *lb = arena_alloc<uint8_t>(c, nblk * nctx * 64);
  int32_t *dsz = arena_alloc<int32_t>(c, nbuf);
  if (!T || !msb || !sse || !lb || !dsz) { c->err = "device arena exhausted"; return 1; }
  hipEvent_t a, b2;
  TAKE_EVENT(c, a, true);
  TAKE_EVENT(c, b2, true);
  HIPCK(c, hipEventRecord(a, c->st));
  if (sacopt_launch_bpsize(dj, nbuf, maxn, maxplanes, T, msb, sse, lb, c->d_bp_fwd, c->d_bp_inv, c->d_bp_sse, c->bp_tscale, c->bp_xscale, dsz, c->st)) { c->err = "k_bpsize launch failed"; return 1; }
  HIPCK(c, hipEventRecord(b2, c->st));
  c->lms_ev.push_back({a, b2});
  HIPCK(c, hipMemcpyAsync(sizes, dsz, (size_t)nbuf * 4, hipMemcpyDeviceToHost, c->st));
  return end_timing(c);
}

// ------------------------------------------------------------------------------------------ the payload bytes themselves
// FrameCoder::EncodeMonoFrame_Normal / _Mapped (codec.cpp:237-265) on the device: include/sacopt.h
extern "C" int sacopt_encode_payloads(sacopt_ctx *c, int nbuf, const int32_t *const *s2u, const int32_t *n, const int32_t *maxbpn,
                                      const uint8_t *const *used, uint8_t *const *out, const int32_t *cap, int32_t *sizes) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (nbuf < 0 || (nbuf && (!s2u || !n || !maxbpn || !out || !cap || !sizes))) { c->err = "bad arguments"; return 1; }
  if (!nbuf) return 0;
  if (nbuf > 8192) { c->err = "more than 8192 buffers in one call"; return 1; }
  size_t in_bytes = 0, out_bytes = 0;
  int nmapped = 0;
  for (int b = 0; b < nbuf; b++) {
    if (n[b] < 0 || (n[b] && !s2u[b])) { c->err = "bad buffer"; return 1; }
    if (maxbpn[b] < 0 || maxbpn[b] > 30) { c->err = "maxbpn out of range (0..30)"; return 1; }
    if (cap[b] < 0 || (cap[b] && !out[b])) { c->err = "bad output buffer"; return 1; }
    in_bytes += ((size_t)n[b] * 4 + 255) & ~(size_t)255;
    out_bytes += ((size_t)cap[b] + 255) & ~(size_t)255;
    if (used && used[b]) { nmapped++; in_bytes += (2 * 32769 + 255) & ~(size_t)255; }
  }
  const size_t need = in_bytes + out_bytes + (size_t)nbuf * (sizeof(BpJob) + sizeof(BpEmit) + sizeof(RcState) + 8 + 256) + (1 << 16);
  if (arena_reserve(c, need)) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<BpJob> jobs(nbuf);
  std::vector<BpEmit> emit(nbuf);
  std::vector<int32_t> which;
  std::vector<uint8_t *> dout(nbuf, nullptr);
  RcState *rcs = arena_alloc<RcState>(c, nbuf);
  if (!rcs) { c->err = "device arena exhausted"; return 1; }
  for (int b = 0; b < nbuf; b++) {
    jobs[b].n = n[b]; jobs[b].maxbpn = maxbpn[b]; jobs[b].buf = nullptr;
    if (n[b] && to_dev(c, s2u[b], (size_t)n[b], &jobs[b].buf)) return 1;
    emit[b].used = nullptr; emit[b].rc = rcs + b; emit[b].cap = cap[b]; emit[b].pad = 0;
    if (used && used[b]) {
      if (to_dev(c, used[b], (size_t)2 * 32769, &emit[b].used)) return 1;
      which.push_back(b);
    }
    dout[b] = (cap[b] && is_device_ptr(out[b])) ? out[b] : arena_alloc<uint8_t>(c, cap[b] ? cap[b] : 1);
    if (!dout[b]) { c->err = "device arena exhausted"; return 1; }
    emit[b].out = dout[b];
  }
  if (which.empty()) which.push_back(0);
  BpJob *dj;
  BpEmit *de;
  int32_t *dw;
  if (upload(c, jobs, &dj) || upload(c, emit, &de) || upload(c, which, &dw)) return 1;
  int32_t *dsz = arena_alloc<int32_t>(c, nbuf);
  if (!dsz) { c->err = "device arena exhausted"; return 1; }
  if (sacopt_launch_payloads(dj, de, nbuf, dw, nmapped, c->d_bp_fwd, c->d_bp_inv, c->d_bp_sse, c->bp_tscale, c->bp_xscale, dsz, c->st)) { c->err = "payload kernels: launch failed"; return 1; }
  HIPCK(c, hipMemcpyAsync(sizes, dsz, (size_t)nbuf * 4, hipMemcpyDeviceToHost, c->st));
  HIPCK(c, hipStreamSynchronize(c->st));
  for (int b = 0; b < nbuf; b++) {  // the bytes that were written (a size beyond cap[b] tells the caller its buffer was too small)
    const size_t got = (size_t)(sizes[b] < cap[b] ? sizes[b] : cap[b]);
    if (got && dout[b] != out[b]) HIPCK(c, hipMemcpyAsync(out[b], dout[b], got, hipMemcpyDeviceToHost, c->st));
  }
  return end_timing(c);
}

extern "C" int sacopt_cost_meanrms(sacopt_ctx *c, const int32_t *buf, int n, double *cost) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (n < 0 || !cost || (n && !buf)) { c->err = "bad arguments"; return 1; }
  if (n == 0) { *cost = 0.; return 0; }
  if (arena_reserve(c, (size_t)n * 4 + (1 << 16))) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  const int32_t *d = buf;
  if (!is_device_ptr(buf)) {
    int32_t *p = arena_alloc<int32_t>(c, n);
    HIPCK(c, hipMemcpyAsync(p, buf, (size_t)n * 4, hipMemcpyHostToDevice, c->st));
    d = p;
  }
  double *d_cost = arena_alloc<double>(c, 1);
  if (sacopt_launch_meanrms(d, n, d_cost, c->st)) { c->err = "k_meanrms launch failed"; return 1; }
  HIPCK(c, hipMemcpyAsync(cost, d_cost, 8, hipMemcpyDeviceToHost, c->st));
  return end_timing(c);
}

// ------------------------------------------------------------------------------------------ stage-level entry points

extern "C" int sacopt_lms_pair(sacopt_ctx *c, int S, int D, double mu1, double mu2, const double *x0, const double *x1, int n,
                               double *p0, double *p1) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (n < 0 || (n && (!x0 || !x1 || !p0 || !p1))) { c->err = "bad arguments"; return 1; }
  if (!n) return 0;
  if (arena_reserve(c, (size_t)n * 8 * 4 + (1 << 16))) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  LmsJob j;
  memset(&j, 0, sizeof(j));
  if (to_dev(c, x0, n, &j.x[0]) || to_dev(c, x1, n, &j.x[1])) return 1;
  j.p[0] = arena_alloc<double>(c, n);
  j.p[1] = arena_alloc<double>(c, n);
  j.mu1 = mu1; j.mu2 = mu2; j.n = n; j.t0 = 0; j.state = nullptr;
  std::vector<LmsJob> jobs = {j};
  int sw = 0;
#ifdef SACOPT_ABLATE
  if (const char *dbg = getenv("SACOPT_DBG")) sw = atoi(dbg) << 8;
  // timing builds: the same job SACOPT_REPL times in one launch (every copy writes the same values to the same outputs), to time a
  // chain -- and read its phase clocks -- while the chip is full
  if (const char *rp = getenv("SACOPT_REPL")) jobs.assign(atoi(rp) > 0 ? atoi(rp) : 1, j);
#endif
  LmsJob *dj;
  if (upload(c, jobs, &dj)) return 1;
  hipEvent_t a, b;
  TAKE_EVENT(c, a, true);
  TAKE_EVENT(c, b, true);
  HIPCK(c, hipEventRecord(a, c->st));
  if (sacopt_launch_lms(S, D, dj, (int)jobs.size(), c->d_tab, c->bits | sw, c->st, L1C_FRAMES, 0, 0)) { c->err = "unsupported (S,D) or launch failure"; return 1; }
  HIPCK(c, hipEventRecord(b, c->st));
  c->lms_ev.push_back({a, b});
  if (copy_out(c, p0, j.p[0], (size_t)n * 8) || copy_out(c, p1, j.p[1], (size_t)n * 8)) return 1;
  return end_timing(c);
}

extern "C" int sacopt_lpc3_pair(sacopt_ctx *c, double alpha, int S, int D, const int32_t *s0, const int32_t *s1, int n, double *p0, double *p1) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (n < 0 || (n && (!s0 || !s1 || !p0 || !p1))) { c->err = "bad arguments"; return 1; }
  ProfDesc pd;
  if (!((S == 16 || S == 24) && D == 8) || !get_prof(S == 16 ? SACOPT_PROFILE_HIGH : SACOPT_PROFILE_HIGH32, &pd)) { c->err = "unsupported (S,D): (16,8) and (24,8) are built"; return 1; }
  if (!n) return 0;
  if (arena_reserve(c, (size_t)n * 8 * 6 + lpc_scratch_bytes(pd, n) + (1 << 16))) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  LpcJob j;
  memset(&j, 0, sizeof(j));
  if (to_dev(c, s0, n, &j.s[0]) || to_dev(c, s1, n, &j.s[1])) return 1;
  for (int k = 0; k < 2; k++) { j.p[k] = arena_alloc<double>(c, n); j.x1[k] = arena_alloc<double>(c, n); }
  j.alpha = alpha; j.n = n;
  if (lpc_scratch_alloc(c, pd, j)) return 1;
  std::vector<LpcJob> jobs = {j};
  LpcJob *dj;
  if (upload(c, jobs, &dj)) return 1;
  if (sacopt_launch_lpc3(dj, 1, n, pd.ols_n, c->st)) { c->err = "LPC3 launch failed"; return 1; }
  if (copy_out(c, p0, j.p[0], (size_t)n * 8) || copy_out(c, p1, j.p[1], (size_t)n * 8)) return 1;
  return end_timing(c);
}

// RLS(nst, alpha) driven through the PRODUCT's mixer kernel k_mix<nst> (lane-per-P[i][j] layout, DPP row sums), in chunks of
// 700 samples with the carried state handed from launch to launch exactly as the cascades' pipelines do.
extern "C" int sacopt_rls_run(sacopt_ctx *c, int nst, double alpha, const double *pv, const double *val, int len, double *p) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (len < 0 || (len && (!pv || !val || !p))) { c->err = "bad arguments"; return 1; }
  if (nst != 3 && nst != 4) { c->err = "RLS order must be 3 or 4"; return 1; }
  if (!len) return 0;
  if (arena_reserve(c, (size_t)len * 8 * (nst + 6) + (size_t)len * 4 + (1 << 16) + (size_t)(len / 700 + 2) * (sizeof(MixJob) + 256))) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  // column k of pv -> its own array (k_mix reads one array per stage)
  std::vector<double> col((size_t)len);
  MixJob j;
  memset(&j, 0, sizeof(j));
  for (int k = 0; k < nst; k++) {
    std::vector<double> hv((size_t)len);
    if (is_device_ptr(pv)) { c->err = "sacopt_rls_run takes host arrays"; return 1; }
    for (int t = 0; t < len; t++) hv[t] = pv[(size_t)t * nst + k];
    double *d;
    if (upload(c, hv, &d)) return 1;
    j.pk[k][0] = j.pk[k][1] = d;
  }
  const double *dval;
  if (to_dev(c, val, len, &dval)) return 1;
  int32_t *zs = arena_alloc<int32_t>(c, len);
  double *zp = arena_alloc<double>(c, len), *dp = arena_alloc<double>(c, len), *dp1 = arena_alloc<double>(c, len);
  double *state = arena_alloc<double>(c, 2 * MIX_STATE);
  if (!zs || !zp || !dp || !dp1 || !state) { c->err = "device arena exhausted"; return 1; }
  HIPCK(c, hipMemsetAsync(zs, 0, (size_t)len * 4, c->st));
  HIPCK(c, hipMemsetAsync(zp, 0, (size_t)len * 8, c->st));
  for (int h = 0; h < 2; h++) { j.s[h] = zs; j.p0[h] = zp; j.x1[h] = dval; }
  j.pout[0] = dp; j.pout[1] = dp1;
  j.direct = 1;
  j.alpha = alpha;
  j.state = state;
  constexpr int CHUNK = 700;
  for (int t0 = 0; t0 < len; t0 += CHUNK) {
    j.t0 = t0;
    j.n = len - t0 < CHUNK ? len - t0 : CHUNK;
    std::vector<MixJob> jobs = {j};
    MixJob *dj;
    if (upload(c, jobs, &dj)) return 1;
    if (sacopt_launch_mix(nst, dj, 1, c->st, 0, 0)) { c->err = "k_mix launch failed"; return 1; }
  }
  if (copy_out(c, p, dp, (size_t)len * 8)) return 1;
  return end_timing(c);
}

// One covariance snapshot through k_chol (Cholesky::Factor + Solve, src/math/cholesky.cpp:16-77), as the LPC stages run it.
extern "C" int sacopt_chol_solve(sacopt_ctx *c, int n, int fast, const double *snap, double *sol, int *fail) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (!snap || !sol || !fail) { c->err = "bad arguments"; return 1; }
  if (!((n == 24) || (n == 32 && !fast))) { c->err = "built orders: 24 (LPC3 all-orders and LPC2 full-order solve), 32 (LPC3)"; return 1; }
  const int tri = n * (n + 1) / 2, nsol = fast ? n : tri;
  if (arena_reserve(c, 1 << 16)) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  LpcJob j;
  memset(&j, 0, sizeof(j));
  std::vector<double> hs(snap, snap + tri + n);
  double *dsnap;
  if (upload(c, hs, &dsnap)) return 1;
  double *dsol = arena_alloc<double>(c, nsol);
  int32_t *dfail = arena_alloc<int32_t>(c, 2);
  HIPCK(c, hipMemsetAsync(dsol, 0, (size_t)nsol * 8, c->st));
  j.snap[0] = dsnap; j.sol[0] = dsol; j.fail[0] = dfail; j.nsnap = 1; j.t0 = 0;
  std::vector<LpcJob> jobs = {j};
  LpcJob *dj;
  if (upload(c, jobs, &dj)) return 1;
  if (sacopt_launch_chol(dj, n, fast, c->st)) { c->err = "k_chol launch failed"; return 1; }
  int32_t hf = 0;
  HIPCK(c, hipMemcpyAsync(sol, dsol, (size_t)nsol * 8, hipMemcpyDeviceToHost, c->st));
  HIPCK(c, hipMemcpyAsync(&hf, dfail, 4, hipMemcpyDeviceToHost, c->st));
  if (end_timing(c)) return 1;
  *fail = hf;
  return 0;
}

// Strict left-to-right FP64 sums on the device helpers the LMS kernels use (the matrix pipe as an exact adder).
extern "C" int sacopt_serial_sum(sacopt_ctx *c, int nterms, const double *x, int count, double *out) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  if (count < 0 || (count && (!x || !out))) { c->err = "bad arguments"; return 1; }
  if (nterms != 1920 && nterms != 384 && nterms != 48 && nterms != 12) { c->err = "built chain lengths: 1920, 384, 48, 12"; return 1; }
  if (!count) return 0;
  if (arena_reserve(c, (size_t)count * (nterms + 1) * 8 + (1 << 16))) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  const double *dx;
  if (to_dev(c, x, (size_t)count * nterms, &dx)) return 1;
  double *dout = arena_alloc<double>(c, count);
  if (!dout) { c->err = "device arena exhausted"; return 1; }
  if (sacopt_launch_sumkat(nterms, dx, count, dout, c->st)) { c->err = "k_sumkat launch failed"; return 1; }
  if (copy_out(c, out, dout, (size_t)count * 8)) return 1;
  return end_timing(c);
}

// Per-stage trace of one cascade run (profile.cpp:126-148): stage predictions p[k] and stage inputs val[ch][k+1] as the stage
// kernels left them in HBM. p_out / v_out: [nstage][2][n] doubles, nstage = 1 + number of LMS stages (LPC3 first).
extern "C" int sacopt_cascade_trace(sacopt_ctx *c, int profile, const sacopt_frame *frame, const double *coefs, double *p_out, double *v_out) {
  if (!c) return 1;
  if (begin_call(c)) return 1;
  ProfDesc pd;
  if (!get_prof(profile, &pd) || pd.type == SACOPT_PROFILE_FAST) { c->err = "cascade trace: normal / high profiles"; return 1; }
  if (!frame || !p_out || !v_out) { c->err = "bad arguments"; return 1; }
  if (check_frames(c, 1, frame, nullptr, nullptr)) return 1;
  const int n = frame->numsamples;
  if (!n) return 0;
  double base[3 * SACOPT_MAX_COEFS];
  sacopt_base_profile(profile, base);
  if (arena_reserve(c, (1 << 20) + chain_bytes(pd, n) + 2 * (((size_t)n * 4 + 255) & ~(size_t)255) + 4096 + lpc_scratch_bytes(pd, n) + sizeof(LpcJob) + 256)) return 1;
  HIPCK(c, hipEventRecord(c->ev0, c->st));
  std::vector<DevFrame> fr;
  if (stage_frames(c, 1, frame, nullptr, nullptr, fr)) return 1;
  std::vector<Chain> chains(1);
  chains[0].frame = 0;
  for (int i = 0; i < pd.ncoef; i++) chains[0].coefs[i] = coefs ? coefs[i] : base[3 * i + 2];
  Memo memo;
  if (eval_chains(c, pd, fr, chains, memo)) return 1;
  for (int s = 0; s <= pd.nlms; s++)
    for (int h = 0; h < 2; h++) {
      if (copy_out(c, p_out + ((size_t)s * 2 + h) * n, chains[0].sb[s].p[h], (size_t)n * 8)) return 1;
      if (copy_out(c, v_out + ((size_t)s * 2 + h) * n, chains[0].sb[s].xn[h], (size_t)n * 8)) return 1;
    }
  return end_timing(c);
}
