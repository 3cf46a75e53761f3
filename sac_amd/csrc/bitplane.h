// the size of a channel's payload as the host's bit-plane coder would write it -- BitplaneCoder::Encode
// (src/coder/vle.cpp:3-138) on RangeCoderSH (src/coder/range.cpp:52-90) with its context models (src/model/counter.h:20-65,
// mixer.h:56-101, sse.h:88-125, domain.h:6-57), i.e. FrameCoder::EncodeMonoFrame_Normal (src/codec.cpp:237-247) without the
// bytes. Included by kernels.hip.
//
// Two forms: k_bppipe (end of this file: a five-stage pipeline per coder, what sacopt_launch_bpsize takes for up to BP_PIPE_MAX
// buffers) and k_bpsize (a lane or a wave per coder; the first correct version and the A/B partner).
// k_bpsize: the coder is one adaptive chain per buffer: every binary decision updates the counters, mixers and the SSE stage the next
// one reads, bit plane after bit plane. Nothing inside a buffer is parallel, so the parallelism is ACROSS buffers -- the
// candidates x channels x frames of an optimizer round: one lane per buffer, a wave = 64 coders in lockstep. All arithmetic
// is the reference's integer arithmetic; the only floating point of the original, the stretch / squash tables of LogDomain
// (log / exp of the host libm), is tabulated by the host at sacopt_create and read from LDS.
//   LDS (per wave):  stretch table 32768 x int16, squash table 4096 x uint16, 1/(n+3) table, and per lane the 322 bit
//                    counters and 42 mixer weights ([element][lane]: bank = lane, conflict-free whatever the contexts are)
//                    and the SSE stage's last-bit flags (a bit per context)
//   HBM (per wave):  residuals transposed to [sample][lane] (coalesced), msb[sample][lane], and the SSE maps
//                    [context][2][33][lane] -- (maxbpn+1) x 32 contexts of 264 bytes per lane do not fit the LDS.
#define BP_PBITS 15
#define BP_PSCALE (1 << BP_PBITS)
#define BP_PSCALEm (BP_PSCALE - 1)
#define BP_WBITS 16
#define BP_DBITS 12
#define BP_WRANGE (1 << 19)
#define BP_SSE_N 32
#define BP_NC1 64    // bpn1[1 + msb], bpn1[1 + 32 + lctx]: indices < 49
#define BP_NC2 258   // bpn2[0], bpn2[1 + ctx1]: indices < 257
#define BP_NW 42     // lmixref[8][3], lmixsig[8][2], finalmix[2]  (mixctx = state & 7: 8 of the 32 mixers are ever used)
#define BP_DIVN 1024 // div_tbl entries reachable with the coder's counter limit (1000)
#ifndef BP_PIPE
#define BP_PIPE 1        // stage-pipeline form k_bppipe for passes of up to BP_PIPE_MAX buffers (A/B switch)
#endif
#ifndef BP_PIPE_MAX
#define BP_PIPE_MAX 8192
#endif
#ifndef BP_WPC_MAX
#define BP_WPC_MAX 2048  // ... and up to this many the wave-per-coder form (8 coders per workgroup: one workgroup per CU, 256 CUs)
#endif
#ifndef BP_LAT_MAX
#define BP_LAT_MAX 4096  // passes of up to this many buffers use the latency form of k_bpsize (8 coders per wave)
#endif

struct BpJob {
  const int32_t *buf;  // s2u residuals (non-negative), n entries
  int32_t n, maxbpn;
};
// ---- the payload BYTES (sacopt_encode_payloads): the range coder's state where one coder hands over to the next
// (FrameCoder::EncodeMonoFrame_Mapped, codec.cpp:249-265: MapEncoder and BitplaneCoder write through ONE RangeCoderSH)
struct RcState {
  unsigned long long lowc;
  uint32_t range, ffnum, cache, pos;
};
struct BpEmit {
  uint8_t *out;         // payload bytes (device), cap of them; bytes beyond cap are dropped, the size still counts them
  const uint8_t *used;  // null: plain variant; else usedl[0..32768] then usedh[0..32768] (Remap's tables): mapped variant
  RcState *rc;          // mapped: where k_mapenc leaves the coder's state for k_bppipe<true>
  int32_t cap, pad;
};
static_assert(sizeof(RcState) == 24 && sizeof(BpEmit) == 32 && sizeof(BpJob) == 16, "layouts shared with sacopt.cpp");

__device__ __forceinline__ int bp_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int bp_idiv32(int val, int s) { return val < 0 ? -(((-val) + (1 << (s - 1))) >> s) : (val + (1 << (s - 1))) >> s; }
__device__ __forceinline__ long long bp_idiv64(long long val, int s) { return val < 0 ? -(((-val) + (1 << (s - 1))) >> s) : (val + (1 << (s - 1))) >> s; }

// residuals of 64 buffers -> T[(block * maxn + s) * 64 + lane]
__global__ __launch_bounds__(256) void k_bp_transpose(const BpJob *__restrict__ jobs, int nbuf, int maxn, int32_t *__restrict__ T) {
  const int lane = threadIdx.x & 63, blk = blockIdx.y, b = blk * 64 + lane;
  const int s0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (b >= nbuf) {
    for (int s = s0; s < s0 + 64 && s < maxn; s++) T[((size_t)blk * maxn + s) * 64 + lane] = 0;
    return;
  }
  const BpJob j = jobs[b];
  for (int s = s0; s < s0 + 64 && s < maxn; s++) T[((size_t)blk * maxn + s) * 64 + lane] = s < j.n ? j.buf[s] : 0;
}

// tables: fwd[32768] int16 | inv[4096] uint16 | sse init map[33] uint16 | tscale, xscale (2 x int32 at the end), as the host laid them out
struct BpTables {
  const int16_t *fwd;
  const uint16_t *inv;
  const uint16_t *sse_init;
  int tscale, xscale;
};
// LPW = coders (lanes) per wave. (Since k_bppipe only LPW = 64 is reachable in the product library -- passes of more than
// BP_PIPE_MAX buffers; the two latency forms are the A/B partners of timing builds with -DBP_PIPE=0.)
//   64  throughput form: every lane a coder; the SSE maps ((maxbpn+1) x 32 contexts x 264 bytes per coder) live in HBM.
//   8   latency form, for passes with few buffers (an optimizer round: candidates x channels x frames): the chip has more CUs than
//       such a pass has waves, so what counts is the time per decision of ONE coder. Eight coders per wave leave room for the SSE
//       maps in LDS -- a plane's contexts ((bpn << 5) + msb, vle.cpp:98) are touched in that plane only, so ONE plane's maps
//       (32 contexts x 2 x 33 bins) are enough and are simply re-initialised per plane: no HBM access in the decision chain at
//       all (the dependent L2 round trip per decision was ~1 us of 2.9) -- and residuals / msb values are fetched 24 samples
//       ahead through a small LDS ring (the msb value of sample s+4 used to be a global load in front of every decision).
//   WPC (with LPW = 8): a WAVE per coder instead of a lane -- 8 waves per workgroup, every lane of a wave computes the same coder
//       (LDS reads broadcast, stores coincide). A wave then runs only the code path its coder takes (lanes that hold different
//       coders execute PredictRef AND PredictSig, masked) and the eight coders' dependent chains interleave on the four SIMDs.
template <int LPW, bool WPC = false>
__global__ __launch_bounds__(WPC ? 64 * LPW : 64) void k_bpsize(const BpJob *__restrict__ jobs, int nbuf, int maxn, int maxplanes, const int32_t *__restrict__ T,
                                               uint8_t *__restrict__ msb_g, uint32_t *__restrict__ sse_g, uint8_t *__restrict__ lb_g,
                                               BpTables tb, int32_t *__restrict__ sizes) {
  constexpr bool LAT = LPW < 64;
  constexpr int SSE_PLANE = 32 * 2 * 33;  // entries of one plane's SSE maps per coder
  __shared__ int16_t fwd[BP_PSCALE];
  __shared__ uint16_t inv[1 << BP_DBITS];
  __shared__ uint16_t divt[BP_DIVN];
  __shared__ uint16_t c1[BP_NC1][LPW], c2[BP_NC2][LPW];
  __shared__ int wts[BP_NW][LPW];
  __shared__ uint32_t lbm[LAT ? 1 : 32][LPW];  // SSENL::lb of context (bpn << 5) + msb: bit msb of word [bpn][lane] (LAT: a register per plane)
  __shared__ uint32_t ssel[LAT ? SSE_PLANE : 1][LPW];  // LAT: the current plane's SSE maps, [(msb * 2 + lb) * 33 + q][lane]: p1 | counter << 16
  __shared__ int32_t rT[LAT ? 32 : 1][LPW];            // LAT: prefetch rings, slot = sample index & 31
  __shared__ uint8_t rM[LAT ? 32 : 1][LPW];
  constexpr int NTH = WPC ? 64 * LPW : 64;  // threads of the workgroup
  const int lane = threadIdx.x, blk = blockIdx.x;
  const int col = WPC ? (int)(threadIdx.x >> 6) : lane;  // this thread's coder within the workgroup
  const int b = blk * LPW + col;
  const bool act = col < LPW;
  const int ln = act ? col : 0;  // (idle lanes of the latency form address column 0 and never store)
  const bool wr = !WPC || (threadIdx.x & 63) == 0;  // the one thread of a coder that stores to HBM
  for (int i = lane; i < BP_PSCALE; i += NTH) fwd[i] = tb.fwd[i];
  for (int i = lane; i < (1 << BP_DBITS); i += NTH) inv[i] = tb.inv[i];
  for (int i = lane; i < BP_DIVN; i += NTH) divt[i] = (uint16_t)(BP_PSCALE / (i + 3));  // counter.h:38-50
  for (int i = lane; i < BP_NC1 * LPW; i += NTH) (&c1[0][0])[i] = BP_PSCALE >> 1;
  for (int i = lane; i < BP_NC2 * LPW; i += NTH) (&c2[0][0])[i] = BP_PSCALE >> 1;
  for (int i = lane; i < BP_NW * LPW; i += NTH) (&wts[0][0])[i] = 0;  // NMixLogistic::Init(0)
  if (!LAT) for (int i = 0; i < 32; i++) lbm[i][ln] = 0;
  const bool live = act && b < nbuf;
  const int n = live ? jobs[b].n : 0, mbp = live ? jobs[b].maxbpn : -1;
  // scratch is laid out per 64 buffers whatever LPW is: buffer b -> block b / 64, column b % 64
  const int32_t *Tb = T + (size_t)(b >> 6) * maxn * 64 + (b & 63);
  uint8_t *msb = msb_g + (size_t)(b >> 6) * (maxn + 16) * 64 + (b & 63);       // [s + 4][lane]: zero entries on either side
  const int nctx = maxplanes * 32;
  uint32_t *sse = sse_g + (size_t)blk * nctx * 66 * 64 + lane;         // (LPW = 64) [(ctx * 2 + lb) * 33 + q][lane]: p1 | counter << 16
  (void)lb_g;
  if (WPC) { if (live) for (int s = (int)(threadIdx.x & 63); s < maxn + 16; s += 64) msb[(size_t)s * 64] = 0; }
  else if (live || !LAT) for (int s = 0; s < maxn + 16; s++) msb[(size_t)s * 64] = 0;
  if (!LAT) for (int e = 0; e < nctx * 66; e++) sse[(size_t)e * 64] = tb.sse_init[e % 33];  // SSENL ctor, sse.h:94-105
  __syncthreads();
  const int tscale = tb.tscale, xscale = tb.xscale;
  // x / xscale for 0 <= x < 2^23 by a multiplication: q = mulhi(x, ceil(2^32 / xscale)) is the quotient or one more
  const uint32_t xmagic = (uint32_t)((0x100000000ull + (uint32_t)xscale - 1) / (uint32_t)xscale);
  auto divx = [&](int x) -> int { int q = (int)__umulhi((uint32_t)x, xmagic); q -= (q * xscale > x) ? 1 : 0; return q; };

  auto dinv = [&](int x) -> int { return x < -(1 << (BP_DBITS - 1)) ? 0 : (x > (1 << (BP_DBITS - 1)) - 1 ? BP_PSCALE - 1 : (int)inv[x + (1 << (BP_DBITS - 1))]); };
  auto lc16 = [&](uint16_t *c, int bit, int L) {  // LinearCounter16::update(bit, L), counter.h:30-35
    const int p = *c, err = (bit << BP_PBITS) - p;
    *c = (uint16_t)bp_clamp(p + bp_idiv32(L * err, BP_PBITS), 1, BP_PSCALEm);
  };
  auto lcl = [&](uint32_t *e, int bit, int limit) {  // LinearCounterLimit::update, counter.h:57-65
    uint32_t v = *e;
    int p1 = (int)(v & 0xffffu), cnt = (int)(v >> 16);
    if (cnt < limit) cnt++;
    const int dp = bit ? ((BP_PSCALE - p1) * (int)divt[cnt]) >> BP_PBITS : -((p1 * (int)divt[cnt]) >> BP_PBITS);
    p1 = bp_clamp(p1 + dp, 1, BP_PSCALEm);
    *e = (uint32_t)p1 | ((uint32_t)cnt << 16);
  };
  // RangeCoderSH (range.cpp:52-90), counting the bytes it would put
  uint32_t range = 0xFFFFFFFFu, ffnum = 0, cache = 0;
  unsigned long long lowc = 0;
  int pos = 0;
  auto shift_low = [&]() {
    const uint32_t carry = (uint32_t)(lowc >> 32), low = (uint32_t)lowc;
    if (low < 0xFF000000u || carry) { pos += 1 + (int)ffnum; ffnum = 0; cache = low >> 24; }
    else ffnum++;
    lowc = (unsigned long long)(low << 8);
  };
  (void)cache;

  int topbpn = mbp;  // the workgroup walks the planes of its deepest buffer; a coder joins at its own maxbpn
  int topn = n;
  if constexpr (WPC) {  // (the plane loop holds workgroup barriers: its bounds are the same in every wave)
    for (int k = 0; k < LPW; k++)
      if (blk * LPW + k < nbuf) { topbpn = max(topbpn, jobs[blk * LPW + k].maxbpn); topn = max(topn, jobs[blk * LPW + k].n); }
  } else {
    for (int o = 32; o; o >>= 1) topbpn = max(topbpn, __shfl_xor(topbpn, o));
    for (int o = 32; o; o >>= 1) topn = max(topn, __shfl_xor(topn, o));
  }
  // LAT: entry idx of the residuals / of the msb array (which has maxn + 16 rows), 0 beyond the ends
  auto gT = [&](int idx) -> int32_t { return (live && idx < topn) ? Tb[(size_t)idx * 64] : 0; };
  auto gM = [&](int idx) -> uint8_t { return (live && idx < maxn + 16) ? msb[(size_t)idx * 64] : (uint8_t)0; };
  for (int bpn = topbpn; bpn >= 0; bpn--) {
    const bool plane_on = live && bpn <= mbp;
    uint32_t state = 0, lbreg = 0;
    if constexpr (LAT) {  // this plane's SSE maps start from the constructor's values (sse.h:94-105); the rings from sample 0
      __syncthreads();
      for (int e = lane; e < SSE_PLANE * LPW; e += NTH) (&ssel[0][0])[e] = tb.sse_init[(e / LPW) % 33];
      if (act)
        for (int k = 0; k < 24; k++) { rT[k][ln] = gT(k); rM[k][ln] = gM(k); }
      __syncthreads();
    }
    // sliding windows: residuals s-2..s+2, msb s-4..s+4 (the entries beyond the buffer's ends read as 0: vle.cpp:19-28, :36-39)
    int a_m2 = 0, a_m1 = 0, a_0 = 0, a_p1 = 0, a_p2 = 0;
    int m[9];
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = 0;
    if constexpr (LAT) {
      a_0 = rT[0][ln]; a_p1 = rT[1][ln]; a_p2 = rT[2][ln];
#pragma unroll
      for (int k = 4; k < 9; k++) m[k] = rM[k][ln];
    } else {
      if (topn > 0) { a_0 = Tb[0]; a_p1 = topn > 1 ? Tb[64] : 0; a_p2 = topn > 2 ? Tb[128] : 0; }
#pragma unroll
      for (int k = 4; k < 9; k++) m[k] = msb[(size_t)(k) * 64];  // m[4] = msb[s = 0] ... m[8] = msb[s = 4]
    }
    for (int s0 = 0; s0 < topn; s0 += (LAT ? 8 : topn)) {
    int32_t pT[LAT ? 8 : 1];
    uint8_t pM[LAT ? 8 : 1];
    if constexpr (LAT) {  // entries s0 + 24 .. s0 + 31 of both arrays: issued here, in the rings eight decisions later
#pragma unroll
      for (int k = 0; k < 8; k++) { pT[k] = gT(s0 + 24 + k); pM[k] = gM(s0 + 24 + k); }
    }
    const int send = LAT ? min(s0 + 8, topn) : topn;
    for (int s = s0; s < send; s++) {
      if (plane_on && s < n) {
        const int bit = (a_0 >> bpn) & 1;
        const int mixctx = (int)(state & 7);
        const int mymsb = m[4];
        const bool sig = mymsb != 0;
        uint16_t *pc0 = &c2[0][ln], *pc1, *pc2 = nullptr;
        int *w;
        int nin;
        if (sig) {  // PredictRef, vle.cpp:32-60
          const int lval = a_m1, lval2 = a_m2;
          const int nval = (s < n - 1) ? a_p1 : 0, nval2 = (s < n - 2) ? a_p2 : 0;
          const int tval = a_0 >> (bpn + 1), tval1 = tval << 1;
          int lctx = 0;
          if ((lval >> bpn) > tval1) lctx += 1;
          if ((nval >> (bpn + 1)) > tval) lctx += 2;
          if ((lval2 >> bpn) > tval1) lctx += 4;
          if ((nval2 >> (bpn + 1)) > tval) lctx += 8;
          pc1 = &c1[1 + mymsb][ln];
          pc2 = &c1[1 + 32 + lctx][ln];
          w = &wts[mixctx * 3][ln];
          nin = 3;
        } else {  // PredictSig, vle.cpp:71-92
          int ctx1 = 0;
          if (m[3]) ctx1 += 1;
          if (m[5] && s + 1 <= n - 1) ctx1 += 2;
          if (m[2]) ctx1 += 4;
          if (m[6] && s + 2 <= n - 1) ctx1 += 8;
          if (m[1]) ctx1 += 16;
          if (m[7] && s + 3 <= n - 1) ctx1 += 32;
          if (m[0]) ctx1 += 64;
          if (m[8] && s + 4 <= n - 1) ctx1 += 128;
          pc1 = &c2[1 + ctx1][ln];
          w = &wts[24 + mixctx * 2][ln];
          nin = 2;
        }
        (void)nin;
        // NMixLogistic::Predict, mixer.h:73-84
        const int x0 = fwd[*pc0], x1 = fwd[*pc1], x2 = sig ? (int)fwd[*pc2] : 0;
        long long sum = (long long)(w[0] * x0) + (long long)(w[LPW] * x1);
        if (sig) sum += (long long)(w[2 * LPW] * x2);
        const int pd = bp_clamp(dinv((int)bp_idiv64(sum, BP_WBITS)), 1, BP_PSCALEm);
        // SSENL::Predict, sse.h:106-118  (context (bpn << 5) + msb, vle.cpp:98)
        const int ctx = (bpn << 5) + mymsb;
        const int lb = LAT ? (int)((lbreg >> mymsb) & 1u) : (int)((lbm[LAT ? 0 : bpn][ln] >> mymsb) & 1u);
        int pq = (int)fwd[pd] + tscale;
        pq = pq < 0 ? 0 : (pq > 2 * tscale ? 2 * tscale : pq);
        const int q = divx(pq), pmod = pq - q * xscale;
        uint32_t *e0 = LAT ? &ssel[(mymsb * 2 + lb) * 33 + q][ln] : &sse[(size_t)((ctx * 2 + lb) * 33 + q) * 64];
        uint32_t *e1 = e0 + (LAT ? LPW : 64);
        const int pl = (int)(*e0 & 0xffffu), ph = (int)(*e1 & 0xffffu);
        const int ps = bp_clamp(divx(pl * (xscale - pmod) + ph * pmod), 1, BP_PSCALEm);
        // finalmix[0].Predict({sse, p1}), vle.cpp:100-102
        int *wf = &wts[40][ln];
        const int y0 = fwd[ps], y1 = fwd[pd];
        const long long fs = (long long)(wf[0] * y0) + (long long)(wf[LPW] * y1);
        const int pf = bp_clamp(dinv((int)bp_idiv64(fs, BP_WBITS)), 1, BP_PSCALEm);
        // RangeCoderSH::EncodeBitOne, range.cpp:65-70
        const uint32_t rnew = (uint32_t)(((unsigned long long)range * ((uint32_t)(BP_PSCALE - pf) << (32 - BP_PBITS))) >> 32);
        if (bit) { range -= rnew; lowc += rnew; } else range = rnew;
        while (range < 0x01000000u) { range <<= 8; shift_low(); }
        // UpdateRef / UpdateSig (vle.cpp:62-69, :94-101): counters at rate 350, mixer at 800
        lc16(pc0, bit, 350);
        lc16(pc1, bit, 350);
        if (sig) lc16(pc2, bit, 350);
        {
          const int err = (bit << BP_PBITS) - pd;
          w[0] = bp_clamp(w[0] + bp_idiv32(bp_idiv32(x0 * err, BP_DBITS) * 800, BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
          w[LPW] = bp_clamp(w[LPW] + bp_idiv32(bp_idiv32(x1 * err, BP_DBITS) * 800, BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
          if (sig) w[2 * LPW] = bp_clamp(w[2 * LPW] + bp_idiv32(bp_idiv32(x2 * err, BP_DBITS) * 800, BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        }
        state = (state << 1) + (sig ? 0u : 1u);
        // UpdateSSE (vle.cpp:104-108): both bins at limit 1000, lb = bit; final mixer at 300
        lcl(e0, bit, 1000);
        lcl(e1, bit, 1000);
        if constexpr (LAT) lbreg = (lbreg & ~(1u << mymsb)) | ((uint32_t)bit << mymsb);
        else lbm[LAT ? 0 : bpn][ln] = (lbm[LAT ? 0 : bpn][ln] & ~(1u << mymsb)) | ((uint32_t)bit << mymsb);
        {
          const int err = (bit << BP_PBITS) - pf;
          wf[0] = bp_clamp(wf[0] + bp_idiv32(bp_idiv32(y0 * err, BP_DBITS) * 300, BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
          wf[LPW] = bp_clamp(wf[LPW] + bp_idiv32(bp_idiv32(y1 * err, BP_DBITS) * 300, BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        }
        if (!sig && bit) { m[4] = bpn; if (wr) msb[(size_t)(s + 4) * 64] = (uint8_t)bpn; }  // vle.cpp:132
      }
      // slide the windows
      a_m2 = a_m1; a_m1 = a_0; a_0 = a_p1; a_p1 = a_p2;
#pragma unroll
      for (int k = 0; k < 8; k++) m[k] = m[k + 1];
      if constexpr (LAT) {
        a_p2 = rT[(s + 3) & 31][ln];
        m[8] = rM[(s + 9) & 31][ln];
      } else {
        a_p2 = (s + 3 < topn) ? Tb[(size_t)(s + 3) * 64] : 0;
        m[8] = msb[(size_t)(s + 9) * 64];  // msb[s + 1 + 4], stored at [s + 5 + 4]; the array has 4 zero entries past maxn
      }
    }
    if constexpr (LAT) {
      if (act) {
#pragma unroll
        for (int k = 0; k < 8; k++) { rT[(s0 + 24 + k) & 31][ln] = pT[k]; rM[(s0 + 24 + k) & 31][ln] = pM[k]; }
      }
    }
    }
  }
  if (live) {
    for (int i = 0; i < 5; i++) shift_low();  // RangeCoderSH::Stop, range.cpp:60-63
    if (wr) sizes[b] = pos;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Stage-pipeline form (round 3). On the ENCODER side every context and every coded bit is a function of the residuals alone,
// so the coder's adaptive parts only feed FORWARD: counters -> mixer -> SSE -> final mixer -> range coder; nothing a later
// part computes is read by an earlier one. The one chain of ~450 dependent instructions per decision therefore splits into
// five shorter ones that run side by side, one wave each, a block of 64 decisions apart:
//   S0  contexts of 64 decisions at once (a lane per sample: msb values are highest-set-bit tests against the plane number)
//   S1  the three bit counters of a decision (lanes 0..2), predictions to LDS
//   S2  NMixLogistic of the decision's context: stretch, dot product, squash, weight update
//   S3  SSENL of context (bpn << 5) + msb: one plane's maps in LDS, re-initialised per plane (a plane's contexts are its own)
//   S4  final mixer + RangeCoderSH byte count
// Iteration `it` of the workgroup loop: stage i works on block it - i, one workgroup barrier per iteration; the hand-over
// buffers are rings of 8 blocks in LDS. The time per decision is the slowest stage's chain (~300 cycles) instead of the sum.
// One or two coders per workgroup (NCOD below); the 64 KB stretch table is read from global memory (BPP_FWD_LDS).
// Same integer arithmetic, same order of updates per counter / weight / bin: sizes identical to k_bpsize.
// branch-free forms of the rounding shifts and clamps (the values of the serial loops are wave-uniform: the compiler keeps them in
// scalar registers and turned every `x < 0 ? .. : ..` into a scalar branch, ~25 cycles each, twenty per decision)
// idiv_signed (global.h): val < 0 ? -((-val + h) >> s) : (val + h) >> s with h = 2^(s-1). For val < 0 that is
// ceil((val - h) / 2^s) = (val + h - 1) >> s (arithmetic shift), so both cases are (val + h + (val >> 31)) >> s.
__device__ __forceinline__ int bpp_idiv32(int val, int sh) { return (val + (1 << (sh - 1)) + (val >> 31)) >> sh; }
__device__ __forceinline__ long long bpp_idiv64(long long val, int sh) { return (val + (1ll << (sh - 1)) + (val >> 63)) >> sh; }
__device__ __forceinline__ int bpp_clamp(int v, int lo, int hi) { return max(lo, min(v, hi)); }
#ifndef BPP_ALLSTORE
#define BPP_ALLSTORE 1  // stores of wave-uniform values by every lane (no exec-mask branch) instead of lane 0 only
#endif
#if BPP_ALLSTORE
#define BPP_ONE
#else
#define BPP_ONE if (lane == 0)
#endif
#define BPP_B 64
#define BPP_RING 8
// coders per workgroup (NCOD): 1 (5 waves, 28 KB of LDS: five workgroups fit a CU) for calls of up to 1280 buffers -- a coder
// that has its CU to itself needs 570 cycles per decision against 670 when two share one (60 full-frame buffers 830 against
// 980 ms), and two workgroups of one coder are never slower than one workgroup of two (an optimizer round's 360 buffers 87
// against 105 ms, 720 buffers 118 against 158 ms); 2 (10 waves, 45 KB) beyond.
#ifndef BPP_ONE_MAX
#define BPP_ONE_MAX 1280
#endif
#define BPP_STAGES 5
#define BPP_SSE (32 * 2 * 33)
// EMIT: also write the bytes (stage S4: RangeCoderSH::ShiftLow's cache / carry / 0xFF-run logic, range.cpp:80-90, one byte store
// by lane 0 per output byte), starting from the state k_mapenc left when the buffer is coded with its map in front.
template <bool EMIT, int NCOD>
__global__ __launch_bounds__(NCOD * BPP_STAGES * 64) void k_bppipe(const BpJob *__restrict__ jobs, int nbuf, BpTables tb, int32_t *__restrict__ sizes, const BpEmit *__restrict__ emit) {
  constexpr int NT = NCOD * BPP_STAGES * 64;
#ifndef BPP_FWD_LDS
#define BPP_FWD_LDS 0  // 1: the stretch table (64 KB) in LDS as well. Every read of it is in a stage's vector prologue (a lane per decision,
#endif                 // 64 at once), off the serial chains: from global memory (L2) the stages are as fast and the workgroup needs 45 KB less
  __shared__ int16_t fwd_l[BPP_FWD_LDS ? BP_PSCALE : 1];
  const int16_t *const fwd = BPP_FWD_LDS ? fwd_l : tb.fwd;
  __shared__ uint16_t inv[(1 << BP_DBITS) + 2];  // [0] = 0 and [4097] = PSCALE - 1: LogDomain::Inv beyond its table (domain.h:33-38)
  __shared__ uint16_t divt[BP_DIVN];
  __shared__ uint16_t sinit[BP_SSE_N + 1];
  __shared__ uint32_t rec[NCOD][BPP_RING][BPP_B];
  __shared__ uint16_t pcs[NCOD][BPP_RING][4][BPP_B];  // (row 3: where the lanes without a counter store)
  __shared__ uint16_t pds[NCOD][BPP_RING][BPP_B], pss[NCOD][BPP_RING][BPP_B];
  __shared__ uint16_t c1[NCOD][BP_NC1], c2[NCOD][BP_NC2];
  __shared__ int wts[NCOD][48];  // lmixref[8][3], lmixsig[8][2 + one weight that stays 0 (its input is 0)]
  __shared__ uint32_t ssel[NCOD][BPP_SSE];  // [(msb * 2 + lb) * 33 + q]: p1 | counter << 16
  __shared__ int32_t tile[NCOD][BPP_B + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cod = wave / BPP_STAGES, stage = wave - cod * BPP_STAGES;
  const int b = blockIdx.x * NCOD + cod;
  const bool live = b < nbuf;
  if (BPP_FWD_LDS) for (int i = tid; i < BP_PSCALE; i += NT) fwd_l[i] = tb.fwd[i];
  for (int i = tid; i < (1 << BP_DBITS); i += NT) inv[i + 1] = tb.inv[i];
  if (tid == 0) { inv[0] = 0; inv[(1 << BP_DBITS) + 1] = BP_PSCALE - 1; }
  for (int i = tid; i < BP_DIVN; i += NT) divt[i] = (uint16_t)(BP_PSCALE / (i + 3));  // counter.h:38-50
  if (tid <= BP_SSE_N) sinit[tid] = tb.sse_init[tid];
  for (int i = tid; i < NCOD * BP_NC1; i += NT) (&c1[0][0])[i] = BP_PSCALE >> 1;
  for (int i = tid; i < NCOD * BP_NC2; i += NT) (&c2[0][0])[i] = BP_PSCALE >> 1;
  for (int i = tid; i < NCOD * 48; i += NT) (&wts[0][0])[i] = 0;  // NMixLogistic::Init(0)
  const int n = live ? jobs[b].n : 0, mbp = live ? jobs[b].maxbpn : -1;
  const int32_t *buf = live ? jobs[b].buf : nullptr;
  const int nb = (n + BPP_B - 1) / BPP_B;
  const int nblk = (n > 0 && mbp >= 0) ? (mbp + 1) * nb : 0;
  int nblk_max = 0;  // (the loop holds workgroup barriers: its bound is the same in every wave)
  for (int k = 0; k < NCOD; k++) {
    const int bb = blockIdx.x * NCOD + k;
    if (bb < nbuf) {
      const int nn = jobs[bb].n, mm = jobs[bb].maxbpn;
      const int q = (nn > 0 && mm >= 0) ? (mm + 1) * ((nn + BPP_B - 1) / BPP_B) : 0;
      nblk_max = q > nblk_max ? q : nblk_max;
    }
  }
  const int tscale = tb.tscale, xscale = tb.xscale;
  const uint32_t xmagic = (uint32_t)((0x100000000ull + (uint32_t)xscale - 1) / (uint32_t)xscale);
  auto divx = [&](int x) -> int { int q = (int)__umulhi((uint32_t)x, xmagic); q -= (q * xscale > x) ? 1 : 0; return q; };
  auto dinv = [&](int x) -> int { return (int)inv[bpp_clamp(x, -(1 << (BP_DBITS - 1)) - 1, 1 << (BP_DBITS - 1)) + (1 << (BP_DBITS - 1)) + 1]; };
  // S2 / S3 / S4 state (registers of the stage's wave)
  int curmix = -1, cw0 = 0, cw1 = 0, cw2 = 0;  // S2: the mixer used last and its weights
  uint32_t lbreg = 0;
  int wf0 = 0, wf1 = 0;
  uint32_t range = 0xFFFFFFFFu, ffnum = 0, cache = 0;
  unsigned long long lowc = 0;
  int pos = 0;
  uint8_t *outp = nullptr;
  int cap = 0;
  if (EMIT && live) {
    const BpEmit em = emit[b];
    outp = em.out; cap = em.cap;
    if (em.used) { const RcState r = *em.rc; range = r.range; ffnum = r.ffnum; cache = r.cache; lowc = r.lowc; pos = (int)r.pos; }
  }
  auto put = [&](uint32_t v) {  // BufIO::PutByte
    if (lane == 0 && pos < cap) outp[pos] = (uint8_t)v;
    pos++;
  };
  auto shift_low = [&]() {  // RangeCoderSH::ShiftLow (range.cpp:80-90); !EMIT: counting the bytes it would put
    const uint32_t carry = (uint32_t)(lowc >> 32), low = (uint32_t)lowc;
    if (low < 0xFF000000u || carry) {
      if constexpr (EMIT) {
        put(cache + carry);
        for (; ffnum != 0; ffnum--) put(carry - 1u);
        cache = low >> 24;
      } else { pos += 1 + (int)ffnum; ffnum = 0; }
    } else ffnum++;
    lowc = (unsigned long long)(low << 8);
  };
  __syncthreads();
#ifdef BPP_PROF  // busy cycles of every stage wave of workgroup 0 (timing builds)
  long long busy = 0;
#endif
  for (int it = 0; it < nblk_max + BPP_STAGES - 1; it++) {
    const int k = it - stage;
#ifdef BPP_PROF
    const long long tb0 = clock64();
#endif
    if (k >= 0 && k < nblk) {
      const int pl = k / nb, bpn = mbp - pl, s0 = (k - pl * nb) * BPP_B, slot = k & (BPP_RING - 1);
      const int jn = min(BPP_B, n - s0);  // decisions of this block
      if (stage == 0) {  // ---- contexts (vle.cpp:17-28, 32-60, 71-92), a lane per sample
        for (int i = lane; i < BPP_B + 8; i += 64) { const int idx = s0 - 4 + i; tile[cod][i] = (idx >= 0 && idx < n) ? buf[idx] : 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int a[9];
#pragma unroll
        for (int q = 0; q < 9; q++) a[q] = tile[cod][lane + q];  // a[4] = abuf[s]
        __builtin_amdgcn_wave_barrier();
        const int s = s0 + lane;
        // msb[] of a sample (vle.cpp:132: the plane of its first set bit, once that plane's decision has been coded; 0 before --
        // and for good when that plane is 0): `before` its decision in this plane, and `after`. "First set bit" = of the CODED
        // planes: a value may have bits above maxbpn (s2u of a residual beyond 2^30 wraps negative and MaxBPN's signed maximum
        // skips it, codec.cpp:178-211); the reference never looks at those planes for msb[], but its context comparisons below
        // shift the whole signed value.
        const int vmask = (int)((2u << mbp) - 1u);
        auto hb = [&](int x) -> int { x &= vmask; return x ? 31 - __clz(x) : -1; };
        auto before = [&](int x) -> int { const int h = hb(x); return h > bpn ? h : 0; };
        auto after = [&](int x) -> int { const int h = hb(x); return h >= bpn ? h : 0; };
        const int mymsb = before(a[4]);
        const bool sig = mymsb != 0;
        const int bit = (a[4] >> bpn) & 1;
        int state = 0;  // (state << 1) + !significant per coded decision of this plane (vle.cpp:128): the last three
#pragma unroll
        for (int q = 1; q <= 3; q++) if (s - q >= 0 && before(a[4 - q]) == 0) state |= 1 << (q - 1);
        int idx1, idx2 = 0;
        if (sig) {
          const int tval = a[4] >> (bpn + 1), tval1 = tval << 1;
          int lctx = 0;
          if ((a[3] >> bpn) > tval1) lctx += 1;
          if ((a[5] >> (bpn + 1)) > tval) lctx += 2;
          if ((a[2] >> bpn) > tval1) lctx += 4;
          if ((a[6] >> (bpn + 1)) > tval) lctx += 8;
          idx1 = 1 + mymsb;
          idx2 = 1 + 32 + lctx;
        } else {
          int ctx1 = 0;
          if (after(a[3])) ctx1 += 1;
          if (before(a[5])) ctx1 += 2;
          if (after(a[2])) ctx1 += 4;
          if (before(a[6])) ctx1 += 8;
          if (after(a[1])) ctx1 += 16;
          if (before(a[7])) ctx1 += 32;
          if (after(a[0])) ctx1 += 64;
          if (before(a[8])) ctx1 += 128;
          idx1 = 1 + ctx1;
        }
        rec[cod][slot][lane] = (uint32_t)bit | ((uint32_t)sig << 1) | ((uint32_t)state << 2) | ((uint32_t)mymsb << 5) | ((uint32_t)idx1 << 10) |
                               ((uint32_t)idx2 << 19) | ((s < n) ? 0x80000000u : 0u);
      } else if (stage == 1) {  // ---- LinearCounter16 x 3 (counter.h:30-35, rate 350: vle.cpp:62-69, 94-101), lane = counter slot
        // (stages 1-4: what a decision needs and no earlier decision changes is fetched for the whole block at once, a lane per
        // decision; the serial loop takes decision j's values out of lane j)
        const uint32_t rv = rec[cod][slot][lane];
        for (int j = 0; j < jn; j++) {
          const uint32_t r = __builtin_amdgcn_readlane(rv, j);
          const int bit = r & 1, sig = (r >> 1) & 1, idx1 = (r >> 10) & 511, idx2 = (r >> 19) & 63;
          // lane 0: bpn2[0]; lane 1: bpn1[1 + msb] / bpn2[1 + ctx1]; lane 2: bpn1[33 + lctx], or an unused counter when the
          // decision has two inputs (its prediction is not read then); the other lanes: that unused counter
          const int i1 = sig ? idx1 : 0, i2 = sig ? 0 : idx1;  // index into c1 (bpn1) / c2 (bpn2)
          uint16_t *addr = lane == 0 ? &c2[cod][0] : (lane == 1 ? (sig ? &c1[cod][i1] : &c2[cod][i2]) : &c1[cod][(lane == 2 && sig) ? idx2 : 63]);
          const int p = *addr;
          const int err = (bit << BP_PBITS) - p;
          *addr = (uint16_t)bpp_clamp(p + bpp_idiv32(__mul24(350, err), BP_PBITS), 1, BP_PSCALEm);
          pcs[cod][slot][min(lane, 3)][j] = (uint16_t)p;
        }
      } else if (stage == 2) {  // ---- NMixLogistic (mixer.h:73-93, rate 800)
        const uint32_t rv = rec[cod][slot][lane];
        const int xv0 = fwd[pcs[cod][slot][0][lane] & BP_PSCALEm], xv1 = fwd[pcs[cod][slot][1][lane] & BP_PSCALEm],
                  xv2 = ((rv >> 1) & 1) ? (int)fwd[pcs[cod][slot][2][lane] & BP_PSCALEm] : 0;
        uint16_t pdv = 0;
        for (int j = 0; j < jn; j++) {
          const uint32_t r = __builtin_amdgcn_readlane(rv, j);
          const int bit = r & 1, sig = (r >> 1) & 1, mixctx = (r >> 2) & 7;
          const int x0 = __builtin_amdgcn_readlane(xv0, j), x1 = __builtin_amdgcn_readlane(xv1, j), x2 = __builtin_amdgcn_readlane(xv2, j);
          const int mix = (mixctx + 8 * (1 - sig)) * 3;
          int *w = &wts[cod][mix];
          if (mix != curmix) { cw0 = w[0]; cw1 = w[1]; cw2 = w[2]; curmix = mix; }  // (written through below: LDS always holds every mixer's weights)
          const int w0 = cw0, w1 = cw1, w2 = cw2;
          const long long sum = (long long)(w0) * (x0) + (long long)(w1) * (x1) + (long long)(w2) * (x2);
          const int pd = bpp_clamp(dinv((int)bpp_idiv64(sum, BP_WBITS)), 1, BP_PSCALEm);
          const int err = (bit << BP_PBITS) - pd;
          const int n0 = bpp_clamp(w0 + bpp_idiv32(__mul24(bpp_idiv32((x0) * (err), BP_DBITS), 800), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
          const int n1 = bpp_clamp(w1 + bpp_idiv32(__mul24(bpp_idiv32((x1) * (err), BP_DBITS), 800), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
          const int n2 = bpp_clamp(w2 + bpp_idiv32(__mul24(bpp_idiv32((x2) * (err), BP_DBITS), 800), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
          cw0 = n0; cw1 = n1; cw2 = n2;
          BPP_ONE { w[0] = n0; w[1] = n1; w[2] = n2; }
          if (lane == j) pdv = (uint16_t)pd;
        }
        pds[cod][slot][lane] = pdv;
      } else if (stage == 3) {  // ---- SSENL<32> (sse.h:94-124, limit 1000) of context (bpn << 5) + msb (vle.cpp:96-108)
        if (s0 == 0) {  // a new plane: its maps start from the constructor's values
          for (int e = lane; e < BPP_SSE; e += 64) ssel[cod][e] = sinit[e % 33];
          lbreg = 0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        const uint32_t rv = rec[cod][slot][lane];
        int qv;
        {
          int pq = (int)fwd[pds[cod][slot][lane] & BP_PSCALEm] + tscale;
          pq = pq < 0 ? 0 : (pq > 2 * tscale ? 2 * tscale : pq);
          const int q = divx(pq);
          qv = q | ((pq - q * xscale) << 8);
        }
        uint16_t psv = 0;
        for (int j = 0; j < jn; j++) {
          const uint32_t r = __builtin_amdgcn_readlane(rv, j);
          const int bit = r & 1, mymsb = (r >> 5) & 31;
          const int qq = __builtin_amdgcn_readlane(qv, j), q = qq & 255, pmod = qq >> 8;
          const int lb = (int)((lbreg >> mymsb) & 1u);
          uint32_t *e0 = &ssel[cod][(mymsb * 2 + lb) * 33 + q];
          const uint32_t v0 = e0[0], v1 = e0[1];
          const int pl0 = (int)(v0 & 0xffffu), ph0 = (int)(v1 & 0xffffu);
          const int ps = bpp_clamp(divx((pl0) * (xscale - pmod) + (ph0) * (pmod)), 1, BP_PSCALEm);
          auto lcl = [&](uint32_t v) -> uint32_t {  // LinearCounterLimit::update, counter.h:57-65
            int p1 = (int)(v & 0xffffu);
            const int cnt = min((int)(v >> 16) + 1, 1000);
            const int dv = (int)divt[cnt];  // (computed from a float reciprocal instead: 765 against 523 cycles for this stage)
            const int nb1 = bit - 1;  // 0 / -1
            const int d = (p1 + bit * (BP_PSCALE - 2 * p1)) * (dv) >> BP_PBITS;  // bit ? (PSCALE - p1) * dv : p1 * dv
            p1 = bpp_clamp(p1 + ((d ^ nb1) - nb1), 1, BP_PSCALEm);
            return (uint32_t)p1 | ((uint32_t)cnt << 16);
          };
          const uint32_t u0 = lcl(v0), u1 = lcl(v1);
          BPP_ONE { e0[0] = u0; e0[1] = u1; }
          if (lane == j) psv = (uint16_t)ps;
          lbreg = (lbreg & ~(1u << mymsb)) | ((uint32_t)bit << mymsb);
        }
        pss[cod][slot][lane] = psv;
      } else {  // ---- finalmix[0].Predict({sse, p1}) (vle.cpp:100-102, rate 300) + RangeCoderSH::EncodeBitOne (range.cpp:65-70)
        const uint32_t rv = rec[cod][slot][lane];
        const int yv0 = fwd[pss[cod][slot][lane] & BP_PSCALEm], yv1 = fwd[pds[cod][slot][lane] & BP_PSCALEm];
        for (int j = 0; j < jn; j++) {
          const uint32_t r = __builtin_amdgcn_readlane(rv, j);
          const int bit = r & 1;
          const int y0 = __builtin_amdgcn_readlane(yv0, j), y1 = __builtin_amdgcn_readlane(yv1, j);
          const long long fs = (long long)(wf0) * (y0) + (long long)(wf1) * (y1);
          const int pf = bpp_clamp(dinv((int)bpp_idiv64(fs, BP_WBITS)), 1, BP_PSCALEm);
          const uint32_t rnew = (uint32_t)(((unsigned long long)range * ((uint32_t)(BP_PSCALE - pf) << (32 - BP_PBITS))) >> 32);
          { const uint32_t mb = 0u - (uint32_t)bit; lowc += (unsigned long long)(rnew & mb); range = (range & mb) - (rnew & mb) + (rnew & ~mb); }
          while (range < 0x01000000u) { range <<= 8; shift_low(); }
          const int err = (bit << BP_PBITS) - pf;
          wf0 = bpp_clamp(wf0 + bpp_idiv32(__mul24(bpp_idiv32((y0) * (err), BP_DBITS), 300), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
          wf1 = bpp_clamp(wf1 + bpp_idiv32(__mul24(bpp_idiv32((y1) * (err), BP_DBITS), 300), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        }
      }
    }
#ifdef BPP_PROF
    busy += clock64() - tb0;
#endif
    __syncthreads();
  }
#ifdef BPP_PROF
  if (blockIdx.x == 0 && lane == 0 && nblk > 0) printf("k_bppipe coder %d stage %d: %lld busy cycles per decision (%d blocks)\n", cod, stage, busy / ((long long)nblk * BPP_B), nblk);
#endif
  if (live && stage == BPP_STAGES - 1) {
    for (int i = 0; i < 5; i++) shift_low();  // RangeCoderSH::Stop, range.cpp:60-63
    if (lane == 0) sizes[b] = pos;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// MapEncoder::Encode (src/coder/map.cpp:3-72, map.h:10-36): the used-value tables of a mapped channel, 2 x 32768 binary
// decisions through four counters, a four-input mixer, one SSE stage, the final mixer and the range coder -- the front part of
// FrameCoder::EncodeMonoFrame_Mapped's payload (codec.cpp:249-265). One wave per buffer; every value of the chain is
// wave-uniform (the compiler keeps it in scalar registers), the tables of 64 values at a time are fetched a lane each.
// 65 536 decisions of ~1k cycles: ~30 ms, all mapped buffers of a call side by side; the bit planes follow from the state
// this kernel leaves in emit[b].rc (k_bppipe<true>).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_mapenc(const BpEmit *__restrict__ emit, const int32_t *__restrict__ which, BpTables tb) {
  __shared__ int16_t fwd[BP_PSCALE];
  __shared__ uint16_t inv[(1 << BP_DBITS) + 2];
  __shared__ uint16_t divt[BP_DIVN];
  __shared__ uint32_t smap[2][BP_SSE_N + 2];  // SSENL's two maps: p1 | counter << 16
  __shared__ uint16_t cnt[24];
  __shared__ int wm[8][4];  // mixl[4], mixh[4]
  const int lane = threadIdx.x;
  const BpEmit em = emit[which[blockIdx.x]];
  for (int i = lane; i < BP_PSCALE; i += 64) fwd[i] = tb.fwd[i];
  for (int i = lane; i < (1 << BP_DBITS); i += 64) inv[i + 1] = tb.inv[i];
  if (lane == 0) { inv[0] = 0; inv[(1 << BP_DBITS) + 1] = BP_PSCALE - 1; }
  for (int i = lane; i < BP_DIVN; i += 64) divt[i] = (uint16_t)(BP_PSCALE / (i + 3));
  if (lane <= BP_SSE_N) smap[0][lane] = smap[1][lane] = tb.sse_init[lane];
  if (lane < 24) cnt[lane] = BP_PSCALE >> 1;
  if (lane < 32) (&wm[0][0])[lane] = 0;
  __syncthreads();
  const int tscale = tb.tscale, xscale = tb.xscale;
  const uint32_t xmagic = (uint32_t)((0x100000000ull + (uint32_t)xscale - 1) / (uint32_t)xscale);
  auto divx = [&](int x) -> int { int q = (int)__umulhi((uint32_t)x, xmagic); q -= (q * xscale > x) ? 1 : 0; return q; };
  auto dinv = [&](int x) -> int { return (int)inv[bpp_clamp(x, -(1 << (BP_DBITS - 1)) - 1, 1 << (BP_DBITS - 1)) + (1 << (BP_DBITS - 1)) + 1]; };
  uint32_t range = 0xFFFFFFFFu, ffnum = 0, cache = 0;
  unsigned long long lowc = 0;
  int pos = 0;
  uint8_t *outp = em.out;
  const int cap = em.cap;
  auto put = [&](uint32_t v) {
    if (lane == 0 && pos < cap) outp[pos] = (uint8_t)v;
    pos++;
  };
  auto shift_low = [&]() {  // range.cpp:80-90
    const uint32_t carry = (uint32_t)(lowc >> 32), low = (uint32_t)lowc;
    if (low < 0xFF000000u || carry) {
      put(cache + carry);
      for (; ffnum != 0; ffnum--) put(carry - 1u);
      cache = low >> 24;
    } else ffnum++;
    lowc = (unsigned long long)(low << 8);
  };
  const uint8_t *ul = em.used, *uh = em.used + 32769;
  int l1 = ul[0], h1 = uh[0], l2 = 0, h2 = 0;  // ul[i-1], uh[i-1], ul[i-2], uh[i-2]
  int wf0 = 0, wf1 = 0, lb = 0;
  for (int i0 = 1; i0 <= 1 << 15; i0 += 64) {
    const int ulv = ul[i0 + lane], uhv = uh[i0 + lane];
    for (int j = 0; j < 64; j++) {
      const int cl = __builtin_amdgcn_readlane(ulv, j), ch = __builtin_amdgcn_readlane(uhv, j);
#pragma unroll
      for (int hi = 0; hi < 2; hi++) {
        // PredictLow (map.cpp:8-20) / PredictHigh (:22-34)
        const int ctx1 = hi ? h1 : l1, ctx2 = hi ? cl : h1, ctx3 = hi ? h2 : l2, bit = hi ? ch : cl, cb = hi ? 12 : 0;
        const int a0 = cb + ctx1, a1 = cb + 2 + ctx2, a2 = cb + 4 + (ctx1 << 1) + ctx3, a3 = cb + 8 + (ctx1 << 1) + ctx2;
        int *w = wm[(hi ? 4 : 0) + ctx1 + (ctx3 << 1)];
        const int p0 = cnt[a0], p1 = cnt[a1], p2 = cnt[a2], p3 = cnt[a3];
        const int x0 = fwd[p0], x1 = fwd[p1], x2 = fwd[p2], x3 = fwd[p3];
        const int w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        const long long sum = (long long)(w0 * x0) + (long long)(w1 * x1) + (long long)(w2 * x2) + (long long)(w3 * x3);  // mixer.h:73-84
        const int pm = bpp_clamp(dinv((int)bpp_idiv64(sum, BP_WBITS)), 1, BP_PSCALEm);
        // PredictSSE (map.cpp:45-49): SSENL<32> (sse.h:106-118), then finalmix
        int pq = (int)fwd[pm] + tscale;
        pq = pq < 0 ? 0 : (pq > 2 * tscale ? 2 * tscale : pq);
        const int q = divx(pq), pmod = pq - q * xscale;
        const uint32_t v0 = smap[lb][q], v1 = smap[lb][q + 1];
        const int ps = bpp_clamp(divx((int)(v0 & 0xffffu) * (xscale - pmod) + (int)(v1 & 0xffffu) * pmod), 1, BP_PSCALEm);
        const int y0 = fwd[ps], y1 = fwd[pm];
        const long long fs = (long long)(wf0 * y0) + (long long)(wf1 * y1);
        const int pf = bpp_clamp(dinv((int)bpp_idiv64(fs, BP_WBITS)), 1, BP_PSCALEm);
        // RangeCoderSH::EncodeBitOne (range.cpp:65-70)
        const uint32_t rnew = (uint32_t)(((unsigned long long)range * ((uint32_t)(BP_PSCALE - pf) << (32 - BP_PBITS))) >> 32);
        if (bit) { range -= rnew; lowc += rnew; } else range = rnew;
        while (range < 0x01000000u) { range <<= 8; shift_low(); }
        // Update (map.cpp:36-43): counters rate 500, mixer rate 1000; UpdateSSE (:51-55): limit 300, finalmix rate 500
        auto lc = [&](int a, int p) { cnt[a] = (uint16_t)bpp_clamp(p + bpp_idiv32(__mul24(500, (bit << BP_PBITS) - p), BP_PBITS), 1, BP_PSCALEm); };
        lc(a0, p0); lc(a1, p1); lc(a2, p2); lc(a3, p3);
        const int err = (bit << BP_PBITS) - pm;
        w[0] = bpp_clamp(w0 + bpp_idiv32(__mul24(bpp_idiv32(x0 * err, BP_DBITS), 1000), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        w[1] = bpp_clamp(w1 + bpp_idiv32(__mul24(bpp_idiv32(x1 * err, BP_DBITS), 1000), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        w[2] = bpp_clamp(w2 + bpp_idiv32(__mul24(bpp_idiv32(x2 * err, BP_DBITS), 1000), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        w[3] = bpp_clamp(w3 + bpp_idiv32(__mul24(bpp_idiv32(x3 * err, BP_DBITS), 1000), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        auto lcl = [&](uint32_t v) -> uint32_t {  // LinearCounterLimit::update, counter.h:57-65
          int pp = (int)(v & 0xffffu);
          const int c2 = min((int)(v >> 16) + 1, 300);
          const int dv = (int)divt[c2];
          const int dp = bit ? ((BP_PSCALE - pp) * dv) >> BP_PBITS : -((pp * dv) >> BP_PBITS);
          pp = bpp_clamp(pp + dp, 1, BP_PSCALEm);
          return (uint32_t)pp | ((uint32_t)c2 << 16);
        };
        smap[lb][q] = lcl(v0); smap[lb][q + 1] = lcl(v1);
        lb = bit;
        const int ef = (bit << BP_PBITS) - pf;
        wf0 = bpp_clamp(wf0 + bpp_idiv32(__mul24(bpp_idiv32(y0 * ef, BP_DBITS), 500), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
        wf1 = bpp_clamp(wf1 + bpp_idiv32(__mul24(bpp_idiv32(y1 * ef, BP_DBITS), 500), BP_DBITS), -BP_WRANGE, BP_WRANGE - 1);
      }
      l2 = l1; h2 = h1; l1 = cl; h1 = ch;
    }
  }
  if (lane == 0) {
    RcState r;
    r.lowc = lowc; r.range = range; r.ffnum = ffnum; r.cache = cache; r.pos = (uint32_t)pos;
    *em.rc = r;
  }
}
