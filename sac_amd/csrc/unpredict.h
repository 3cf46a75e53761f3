// Decoder-side cascade: FrameCoder::UnpredictStereoFrame (src/codec.cpp:216-235) with
// Stereo{Fast,Normal,High}::Unpredict + Update (src/profile.cpp:34-43, 89-111, 161-186) and Remap::Unmap
// (src/coder/map.cpp:154-168). Included by kernels.hip (uses its device helpers).
//
// Unlike the encoder, nothing here is feed-forward between stages: the sample v(t) is only known once
// EVERY stage has predicted, and every stage's next input depends on v(t). So one workgroup decodes one
// frame with all stages in lockstep, two channel phases per sample (channel 1 predicts from v0(t)):
//   [A] products of all LMS stages of the channel            (all threads)          | barrier
//   [B] exact serial sums, side by side on different waves:  (1280,640): spec_exact_sum on waves 0-3,
//       (256,128): one MFMA chain on wave 5, (32,16)+(8,4): wave 7, LPC3 all-orders + BlendErr predict
//       (or LPC2) on wave 6                                                          | barrier
//   [C] wave 7: RLS predict, round, clamp (high), Unmap, v = e + p, CalcError -> next stage inputs | barrier
// then [D] Update of everything for both channels (all threads), and every k-th sample the Cholesky
// factorisation + solve (waves 0/1 = channel 0/1). Every scalar is produced by the reference's operation
// sequence (same helpers as the encoder kernels), so v is bit-identical to the reference decoder's.
#define UN_NT 1024
#define UN_IO 256  // residuals / samples staged through LDS, this many per channel at a time

// position of the k-th (1-based) set bit of a 32-bit word that has at least k
__device__ __forceinline__ int kth_set_bit(uint32_t x, int k) {
  int pos = 0;
#pragma unroll
  for (int s = 16; s; s >>= 1) {
    const int c = __popc(x & ((1u << s) - 1u));
    if (k > c) { k -= c; x >>= s; pos += s; }
  }
  return pos;
}
// Remap::Unmap(pred, merr) (map.cpp:154-168) for a whole wave (arguments and result uniform). mp: LDS, UN_MAPW bitmap
// words over the values -32768..32768 (bit v+32768; value 0 always set, map.cpp:138) then UN_MAPW counts of set bits
// in the words before each word (entry 2049 = total). The reference walks err = 1, 2, ... until `merr` used values
// have been passed: that is a select of the (rank + merr)-th set bit, at whatever distance it lies (a garbage prediction far
// outside the table still terminates in the reference, with the true distance). A stream that asks for more used values
// than exist in that direction makes the reference loop forever; only then the result is +-(4*32768 + 1), as in the oracle.
__device__ __forceinline__ int unmap_wave(const uint32_t *mp, int pred, int merr, int lane) {
  if (merr == 0) return 0;
  const uint32_t *bw = mp, *rk = mp + UN_MAPW;
  const int sgn = merr < 0 ? -1 : 1, need = merr < 0 ? (int)(0u - (uint32_t)merr) : merr;  // (-INT32_MIN stays INT32_MIN, as in the reference's build)
  const int total = (int)rk[2049];
  const long long ip = (long long)pred + 32768;  // position of pred
  auto rank_le = [&](long long x) -> int {      // set bits at positions <= x
    if (x < 0) return 0;
    if (x > 65536) return total;
    const int w = (int)(x >> 5), b = (int)(x & 31);
    return (int)rk[w] + __popc(bw[w] & (0xffffffffu >> (31 - b)));
  };
  long long kg;
  int w0;
  if (sgn > 0) { kg = (long long)rank_le(ip) + need; w0 = ip + 1 < 0 ? 0 : (ip + 1 > 65536 ? 2048 : (int)((ip + 1) >> 5)); }
  else { kg = (long long)rank_le(ip - 1) - need + 1; w0 = ip - 1 < 0 ? 0 : (ip - 1 > 65536 ? 2048 : (int)((ip - 1) >> 5)); }
  if (kg < 1 || kg > total) return sgn * (4 * 32768 + 1);  // (need = INT32_MIN, from merr = INT32_MIN, ends here too: the oracle's walk never counts down to it)
  const int kglob = (int)kg;
  // word holding the kglob-th set bit: probe 64 words from w0 in the walking direction, else bisect the counts
  int w = w0 + sgn * lane;
  const bool in = w >= 0 && w <= 2048;
  const bool hit = in && (int)rk[w] < kglob && kglob <= (int)rk[w + 1];
  const unsigned long long m = __ballot(hit);
  if (m) w = w0 + sgn * (int)__builtin_ctzll(m);
  else {
    int lo = 0, hi = 2048;  // invariant: rk[lo] < kglob <= rk[hi + 1]
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int)rk[mid + 1] >= kglob) hi = mid; else lo = mid + 1;
    }
    w = lo;
  }
  w = __builtin_amdgcn_readfirstlane(w);
  const long long x = 32ll * w + kth_set_bit(bw[w], kglob - (int)rk[w]);
  // d = distance walked. x is in 0..65536 but pred is any int32 (a garbage stream reaches INT32_MIN through cvt_i32_x86), so
  // d can reach 2^31 + 32768: beyond int32. The reference's `pred + sgn * err` is signed overflow (undefined behaviour) in
  // exactly those cases, so there is nothing to pin ("parity unpinned" there): the step wraps modulo 2^32, which is what the
  // oracle's (int32_t) cast and the reference's compiled code (two's-complement add) do in practice
  // (tests/test_gpu_unpredict.py::test_unpredict_unmap_out_of_table decodes garbage streams whose predictions leave the table).
  const long long d = sgn > 0 ? x - ip : ip - x;
  return (int)(unsigned)(unsigned long long)(sgn * d);
}

// Out of line: the factorisation holds a matrix row and a solution row per lane (96 VGPRs). Inlined, the allocator
// spills the per-sample state of every wave around it; as a call, only waves 0/1 pay, every k-th sample.
template <bool FAST, int N>
__device__ __noinline__ void chol_wave_call(const double *snap, double *solo, int lane) { chol_wave<FAST, N>(snap, solo, lane); }

template <int TYPE>  // 0 fast, 1 normal, 2 high, 3 high with the 32-tap OLS stage LPC3(alpha, 24, 8, 8, 500) (SACOPT_PROFILE_HIGH32)
__global__ __launch_bounds__(UN_NT) void k_unpredict(const UnJob *__restrict__ jobs, const uint16_t *__restrict__ gtab, int bits) {
  constexpr bool BIG = TYPE >= 2, LMS = TYPE >= 1;
  // the OLS stage: OL_N taps of which OL_S are the channel's own history (lpc.h:181-185 takes both)
  constexpr int OL_N = TYPE == 3 ? 32 : 24, OL_S = TYPE == 3 ? 24 : 16, OL_TRI = LPC_TRI_(OL_N), OL_SNAP = LPC_SNAP_(OL_N);
  constexpr int OL_ROW = OL_N + 1;                            // row stride of the product table (odd: operand reads on different banks)
  constexpr int NCE = (2 * OL_SNAP + UN_NT - 1) / UN_NT;       // AutoCov elements per thread (1; 2 for the 32-tap stage: 1120 elements)
  constexpr bool MAPG = TYPE == 3;  // the Unmap tables stay in HBM: with the larger OLS tables the workgroup's LDS has no room for 33 KB more
  constexpr int NS = BIG ? 4 : 3;  // LMS stages (TYPE >= 1): [ (1280,640) ] (256,128) (32,16) (8,4)
  constexpr int KS = TYPE == 0 ? 32 : 8;
  constexpr int SEG = L1_SEG, SEGP = L1_SEGP, NSEG = 15, SB = 1280, MB = 2047;
  constexpr int kM = BIG ? 1 : 0, kS = kM + 1, kT = kM + 2;  // stage numbers of the three smaller LMS stages
  __shared__ double ringB[2][BIG ? 2048 : 1];  // stage inputs val[c][k+1](t) at [t & mask]
  __shared__ double ringM[2][512], ringS[2][64], ringT[2][16];
  __shared__ double prodsB[BIG ? 16 * SEGP : 1];
  __shared__ double prodsM[384], prodsS[48], prodsT[16];  // (the MFMA chain prefetches a few operands past prodsM)
  __shared__ int sJ[16][16];
  __shared__ double sA[16], sX1;
  __shared__ uint16_t tab[LMS ? RSQ_LDS_MAX : 1];
  __shared__ double vh[2][64];  // decoded samples as doubles, [t & 63]
  __shared__ double snapL[2][OL_SNAP], solL[2][OL_TRI];
  __shared__ double pst[8];      // p[0..NS] of the channel being decoded (profile.cpp:128-132)
  __shared__ double errL[2][4];  // val[c][k+1] - p[k+1]: LMSADA2::Update's err (nlms.h:116) per stage
  __shared__ uint32_t mapL[2][MAPG ? 1 : 2 * UN_MAPW];
  __shared__ int32_t ebuf[2][UN_IO], vbuf[2][UN_IO];
  // state only one wave touches lives in LDS, [slot][lane], not in registers (a 1024-thread workgroup has 128 per lane):
  // BlendErr of wave 6 (slots c: ssum, 2+c: serr, 4+c: w, 6+c: pm), RLS of wave 7 (0: P[i][j], 1..4: w, 5+4c+k: hist,
  // 13+c: p, 15+c: val[c][1])
  __shared__ double blS[8][64], rlS[17][64];
  // weights and gradient energies of the three smaller LMS stages ([w ch0, w ch1, eg ch0, eg ch1][tap]); only the
  // (1280,640) taps stay in registers
  __shared__ double lmM[4][LMS ? 384 : 1], lmS[4][64];
  // LPC3::Predict on the matrix pipe: products sol[o][i]*hist[i] as a 24 x 24 table (row stride 25: the 16 chain lanes of
  // an operand read on different banks; entries i > o stay +0, which leaves a sum that started at +0 unchanged),
  // the products w[o]*ssum[o] of BlendErr::Predict, and (order << 8 | i) of every packed solution entry
  __shared__ double prodL[LMS ? OL_N * OL_ROW : 1], prodP[OL_N];
  __shared__ uint16_t triL[LMS ? OL_TRI : 1];

  const UnJob &job = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = job.n;  // wave: an SGPR, roles branch on it
  bits &= 0xff;
  const bool mapped0 = LMS && job.map[0], mapped1 = LMS && job.map[1];
  if (LMS) for (int i = tid; i < (2 << bits); i += UN_NT) tab[i] = gtab[i];
  if (BIG) for (int i = tid; i < 2 * 2048; i += UN_NT) (&ringB[0][0])[i] = 0.;
  if (BIG) for (int i = tid; i < 16 * SEGP; i += UN_NT) prodsB[i] = 0.;
  for (int i = tid; i < 2 * 512; i += UN_NT) (&ringM[0][0])[i] = 0.;
  if (tid < 128) (&ringS[0][0])[tid] = 0.;
  if (tid < 32) (&ringT[0][0])[tid] = 0.;
  if (tid < 16) prodsT[tid] = 0.;
  if (tid < 128) (&vh[0][0])[tid] = 0.;
  for (int i = tid; i < 2 * OL_TRI; i += UN_NT) (&solL[0][0])[i] = 0.;
  if (!MAPG && mapped0) for (int i = tid; i < 2 * UN_MAPW; i += UN_NT) mapL[0][i] = job.map[0][i];
  if (!MAPG && mapped1) for (int i = tid; i < 2 * UN_MAPW; i += UN_NT) mapL[1][i] = job.map[1][i];

  // ---- per-thread state
  // (1280,640): wave s owns segment s (taps 128 s + 64 r + lane), for both channels
  double wB[2][2], gB[2][2];  // (the history values are read again from the rings in [D]: registers are short)
  // offB[r] (UN_FRESH): own: lag 1+i; other: j = i - 1280 (lag j + lagx of the channel), tap i = 128 wave + 64 r + lane
  const bool vB = BIG && wave < NSEG;
  const bool ownB = wave < SB / SEG;  // segments 0..9 hold the channel's own history, 10..14 the other channel's
  const double muB = ownB ? job.coef[1] : job.coef[2];
#pragma unroll
  for (int r = 0; r < 2; r++) wB[0][r] = wB[1][r] = gB[0][r] = gB[1][r] = 0.;
  // (256,128): threads 0..383; (32,16): wave 15 lanes 0..47; (8,4): wave 15 lanes 48..59
  const bool ownM = wave < 4;
  const int cM = BIG ? 3 : 1;  // coefficient numbers (codec.cpp:8-30)
  const double muM = ownM ? job.coef[cM] : job.coef[cM + 1], muS = job.coef[cM + 2], muT = job.coef[cM + 3];
  if (LMS) for (int i = tid; i < 4 * 384; i += UN_NT) (&lmM[0][0])[i] = 0.;
  if (tid < 4 * 64) (&lmS[0][0])[tid] = 0.;  // (32,16) or (8,4) tap of wave 15
  if (LMS) {
    for (int i = tid; i < OL_N * OL_ROW; i += UN_NT) prodL[i] = 0.;
    for (int e = tid; e < OL_TRI; e += UN_NT) {
      int o = 0;
      while ((o + 1) * (o + 2) / 2 <= e) o++;
      triL[e] = (uint16_t)((o << 8) | (e - o * (o + 1) / 2));
    }
  }
  // AutoCov elements (cov.h:13-51): element e = tid + UN_NT * u of 2 * OL_SNAP (648: one per thread; 1120: up to two), channel e / OL_SNAP
  int cpack[NCE];  // ci | cj << 8 | channel << 16 | isc << 17 | valid << 18: one register per element across the loop
  double cv[NCE];
#pragma unroll
  for (int u = 0; u < NCE; u++) {
    const int el = tid + UN_NT * u;
    const int cch = el / OL_SNAP, ce = el - cch * OL_SNAP;
    const bool vC = el < 2 * OL_SNAP, isc = ce < OL_TRI;
    int cj = 0, ci = 0;
    if (isc) { while ((cj + 1) * (cj + 2) / 2 <= ce) cj++; ci = ce - cj * (cj + 1) / 2; }
    else ci = ce - OL_TRI;
    cv[u] = (vC && isc && ci == cj) ? 1.0 : 0.0;  // cov.h:17
    cpack[u] = ci | (cj << 8) | ((cch & 1) << 16) | ((int)isc << 17) | ((int)vC << 18);
  }
  const double alpha = job.coef[0];
  // hist[i] of LPC channel ch at time t (cov.h:44-47 as indices): i < OL_S own v(t-1-i), else other v(t - lagb - (i-OL_S))
  auto lpc_h = [&](int ch, int i, int t) -> double {
    const int idx = (i < OL_S) ? (t - 1 - i) : (t - (ch == 0 ? 1 : 0) - (i - OL_S));
    return vh[(i < OL_S) ? ch : 1 - ch][idx & 63];
  };
  // BlendErr (blend.h:7-53) of both channels: wave 6, lane o = order o, lane 24 = the blended prediction's error
  for (int i = tid; i < 8 * 64; i += UN_NT) (&blS[0][0])[i] = 0.;
  for (int i = tid; i < 17 * 64; i += UN_NT) (&rlS[0][0])[i] = 0.;
  // RLS mixer (rls.cpp:61-108): wave 7, unit u = lane >> 4 = channel (units 2, 3 idle), lane (i, j) owns P[i][j]
  __syncthreads();
  if (wave == 7) {
    const int ri = (lane >> 2) & 3, rj = lane & 3;
    rlS[0][lane] = (ri < NS && rj < NS && ri == rj) ? 1. : 0.;  // P = I (rls.h)
  }
  const double ralpha = LMS ? job.coef[BIG ? 7 : 5] : 1.;
  const double ria = 1. / ralpha;
  // mk (UN_FRESH) = lane >> 4: MFMA operand lane, term 4s + mk at step s
  const int32_t mnv[2] = {job.minval[0], job.minval[1]}, mxv[2] = {job.maxval[0], job.maxval[1]};
  long long rounds = 0;
  __syncthreads();
  // Values derived from the lane number are recomputed where a phase needs them, from a thread index the compiler cannot
  // see through: hoisted out of the sample loop they (and the addresses built on them) filled the 128 registers a lane of
  // a 1024-thread workgroup has, and every reload from scratch memory sat in front of a barrier the other waves wait at.
#define UN_FRESH                                                                                                   \
  int tid = threadIdx.x;                                                                                           \
  asm volatile("" : "+v"(tid));                                                                                    \
  const int lane = tid & 63;                                                                                       \
  const int offB[2] = {ownB ? 1 + SEG * wave + lane : SEG * wave + lane - SB,                                      \
                       ownB ? 1 + SEG * wave + 64 + lane : SEG * wave + 64 + lane - SB};                           \
  const bool vM = LMS && wave < 6, vS = LMS && wave == 15 && lane < 48, vT = LMS && wave == 15 && lane >= 48 && lane < 60; \
  const int iT = lane - 48;                                                                                        \
  const bool ownS = lane < 32, ownT = iT < 8;                                                                      \
  const int offM = ownM ? 1 + tid : tid - 256, offS = ownS ? 1 + lane : lane - 32, offT = ownT ? 1 + iT : iT - 8;  \
  const int mk = lane >> 4;                                                                                        \
  const int ri = (lane >> 2) & 3, rj = lane & 3, ru = lane >> 4, base16 = lane & ~15;                              \
  (void)offB; (void)vM; (void)vS; (void)vT; (void)offM; (void)offS; (void)offT; (void)mk; (void)ri; (void)rj; (void)ru; (void)base16;
#ifdef UN_PROF  // phase clocks of wave 0 (a participant of every barrier), written over the first samples of channel 0
  long long pfA = 0, pfB = 0, pfC = 0, pfD = 0, pfK = 0, pfT = clock64();
  long long pwB = 0, pwD = 0;  // every wave: its own busy time in [B] and [D] (until it reaches the closing barrier)
#define UN_TICK(acc) { const long long now_ = clock64(); acc += now_ - pfT; pfT = now_; }
#define UN_BUSY(acc) { acc += clock64() - pfT; }
#else
#define UN_TICK(acc)
#define UN_BUSY(acc)
#endif

  for (int t = 0; t < n; t++) {
    if ((t & (UN_IO - 1)) == 0) {
      if (t > 0) {
        __syncthreads();
        for (int k = tid; k < 2 * UN_IO; k += UN_NT) (k < UN_IO ? job.v[0] : job.v[1])[t - UN_IO + (k & (UN_IO - 1))] = (&vbuf[0][0])[k];
      }
      for (int k = tid; k < 2 * UN_IO; k += UN_NT) {
        const int tau = t + (k & (UN_IO - 1));
        (&ebuf[0][0])[k] = tau < n ? (k < UN_IO ? job.e[0] : job.e[1])[tau] : 0;
      }
      __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int lagx = c == 0 ? 1 : 0;  // channel 0 sees the other channel up to t-1, channel 1 up to t (profile.cpp:128-132)
      // ---- [A] products w[i]*hist[i] (nlms.h:111)
      if constexpr (LMS) {
        UN_FRESH
        if constexpr (BIG) {
          double part = 0.;
#pragma unroll
          for (int r = 0; r < 2; r++) {
            const double h = ownB ? ringB[c][(t - offB[r]) & MB] : ringB[1 - c][(t - lagx - offB[r]) & MB];
            const double pr = wB[c][r] * h;
            if (vB) prodsB[wave * SEGP + 64 * r + lane] = pr;
            part += pr;
          }
          // approximate segment sum (tree order: a guess only)
          part = row_prefix16(part);
          {
            int lo = __double2loint(part), hi = __double2hiint(part);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
            part += __hiloint2double(hi, lo);
            lo = __double2loint(part); hi = __double2hiint(part);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
            hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false);
            part += __hiloint2double(hi, lo);
          }
          if (lane == 63 && vB) sA[wave] = part;
        }
        if (vM) {
          const double h = ownM ? ringM[c][(t - offM) & 511] : ringM[1 - c][(t - lagx - offM) & 511];
          prodsM[tid] = lmM[c][tid] * h;
        }
        if (vS) {
          const double h = ownS ? ringS[c][(t - offS) & 63] : ringS[1 - c][(t - lagx - offS) & 63];
          prodsS[lane] = lmS[c][lane] * h;
        }
        if (vT) {
          const double h = ownT ? ringT[c][(t - offT) & 15] : ringT[1 - c][(t - lagx - offT) & 15];
          prodsT[iT] = lmS[c][lane] * h;
        }
        __syncthreads();
      }
      UN_TICK(pfA)
      // ---- [B] the strict left-to-right sums (nlms.h:110-111, lpc.h:190-199, blend.h:15-17). Waves 4..6 pass through
      // spec_exact_sum's barrier as bystanders first and sum while wave 0 walks and verifies (the (256,128) chain is as
      // long as walk + verification; in front of the barrier it would delay the walk)
      {
      UN_FRESH
      if constexpr (BIG) {
        const double res = spec_exact_sum<NSEG, SEG / 4, SEGP>((uint32_t)(uintptr_t)&prodsB[0], sJ, &sX1, sA, wave < 4 ? wave : -1,
                                                               wave == 0, 0, rounds, lane);
        if (wave == 0 && lane == SPEC_RESULT_LANE(NSEG)) pst[1] = res;
      }
      if (LMS && wave == 5) {  // SIMD 1 (waves go round the four SIMDs): wave 0's verification pass has SIMD 0's matrix pipe to itself
        const double acc = mfma_chain_lds<6>((uint32_t)(uintptr_t)&prodsM[mk]);
        if (lane == 0) pst[kM + 1] = acc;
      }
      if (LMS && wave == 7) {
        double a3[12], a4[3], acc3 = 0., acc4 = 0.;
#pragma unroll
        for (int u = 0; u < 12; u++) a3[u] = prodsS[4 * u + mk];
#pragma unroll
        for (int u = 0; u < 3; u++) a4[u] = prodsT[4 * u + mk];
#pragma unroll
        for (int u = 0; u < 12; u++) acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a3[u], 1.0, acc3, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 3; u++) acc4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[u], 1.0, acc4, 0, 0, 0);
        if (lane == 0) { pst[kS + 1] = acc3; pst[kT + 1] = acc4; }
      }
      if (wave == 6) {
        if constexpr (LMS) {  // LPC3::Predict: all-orders sums (lpc.h:190-199), then BlendErr::Predict (blend.h:13-17)
          // A lane that keeps order `lane` to itself executes 24 dependent multiply-adds and 24 more for the blend at the
          // single-wave FP64 issue rate (4k cycles: it bounded the phase). Instead: the 300 products, 5 per lane, into the
          // table; the 24 sums are MFMA adder chains (B == 1: ((C + a0) + a1) + a2) + a3 per step, IEEE rounding after
          // each add, terms in the reference's order i = 0, 1, ...), 16 orders per instruction, 6 steps.
#pragma unroll
          for (int q = 0; q < (OL_TRI + 63) / 64; q++) {
            const int e = lane + 64 * q;
            if (e < OL_TRI) {
              const int oi = triL[e], i = oi & 255;
              prodL[(oi >> 8) * OL_ROW + i] = solL[c][e] * lpc_h(c, i, t);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          constexpr int NQ = OL_N / 4;  // MFMA steps of an order's sum
          const int r0 = (lane & 15) * OL_ROW + mk;                                                       // orders 0..15
          const int r1 = (16 + (lane & 15) < OL_N ? 16 + (lane & 15) : OL_N - 1) * OL_ROW + mk;         // orders 16..OL_N-1 (rest: copies)
          double a0[NQ], a1[NQ], acc0 = 0., acc1 = 0.;
#pragma unroll
          for (int u = 0; u < NQ; u++) { a0[u] = prodL[r0 + 4 * u]; a1[u] = prodL[r1 + 4 * u]; }
#pragma unroll
          for (int u = 0; u < NQ; u++) {
            acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[u], 1.0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[u], 1.0, acc1, 0, 0, 0);
          }
          const int jD = 4 * ((lane >> 2) & 3) + (lane >> 4);  // chain of this D-layout lane
          if ((lane & 3) == 0) {
            blS[c][jD] = acc0;
            if (16 + jD < OL_N) blS[c][16 + jD] = acc1;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (lane < OL_N) prodP[lane] = blS[4 + c][lane] * blS[c][lane];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          double ap[NQ], pm = 0.;
#pragma unroll
          for (int u = 0; u < NQ; u++) ap[u] = prodP[4 * u + mk];
#pragma unroll
          for (int u = 0; u < NQ; u++) pm = __builtin_amdgcn_mfma_f64_4x4x4f64(ap[u], 1.0, pm, 0, 0, 0);
          if (lane == 0) { pst[0] = pm; blS[6 + c][OL_N] = pm; }  // slot OL_N: the error of the blend itself (blend.h:38)
        } else {  // LPC2::Predict (lpc.h:261-267)
          double p = 0.;
#pragma unroll
          for (int i = 0; i < OL_N; i++) p += solL[c][i] * lpc_h(c, i, t);
          if (lane == 0) pst[0] = p;
        }
      }
      }
      UN_BUSY(pwB)
      __syncthreads();
      UN_TICK(pfB)
      // ---- [C] mix, round, unmap, reconstruct, CalcError (profile.cpp:134-148, 161-174)
      if (wave == 7) {
        UN_FRESH
        const int32_t e_in = ebuf[c][t & (UN_IO - 1)];
        int32_t v;
        double val[NS + 1];
        if constexpr (LMS) {
          double h[NS];
          h[0] = pst[1];
#pragma unroll
          for (int k = 1; k < NS; k++) h[k] = h[k - 1] + pst[k + 1];  // pv (profile.cpp:134)
          double p = 0.;
#pragma unroll
          for (int k = 0; k < NS; k++) p += h[k] * rlS[1 + k][lane];  // RLS::Predict (rls.cpp:67-72), this lane's unit
          const double pc = bcast(p, 16 * c);
          int32_t q = cvt_i32_x86(round(pst[0] + pc));
          if (BIG) q = q < mnv[c] ? mnv[c] : (q > mxv[c] ? mxv[c] : q);  // profile.cpp:163,170
          int32_t e = e_in;
          if (c == 0 ? mapped0 : mapped1) e = unmap_wave(MAPG ? job.map[c] : mapL[c], q, e, lane);
          v = (int32_t)((uint32_t)e + (uint32_t)q);  // wraps like the reference build
          val[0] = (double)v;
#pragma unroll
          for (int k = 1; k <= NS; k++) val[k] = val[k - 1] - pst[k - 1];  // CalcError (profile.cpp:141-148)
#pragma unroll
          for (int k = 0; k < NS; k++) rlS[5 + 4 * c + k][lane] = h[k];
          rlS[13 + c][lane] = pc; rlS[15 + c][lane] = val[1];
          if (lane == 0) {
            if (BIG) ringB[c][t & MB] = val[1];
            ringM[c][t & 511] = val[kM + 1];
            ringS[c][t & 63] = val[kS + 1];
            ringT[c][t & 15] = val[kT + 1];
#pragma unroll
            for (int k = 0; k < NS; k++) errL[c][k] = val[k + 1] - pst[k + 1];
          }
        } else {
          v = cvt_i32_x86((double)e_in + round(pst[0]));  // profile.cpp:37,41
        }
        if (lane == 0) { vh[c][t & 63] = (double)v; vbuf[c][t & (UN_IO - 1)] = v; }
      }
      __syncthreads();
      UN_TICK(pfC)
    }
    // ---- [D] Update (profile.cpp:102-111, 176-186)
    {
    UN_FRESH
    if constexpr (LMS) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int lagx = c == 0 ? 1 : 0;
        if (vB) {
          const double err = errL[c][0];
#pragma unroll
          for (int r = 0; r < 2; r++) {
            const double h = ownB ? ringB[c][(t - offB[r]) & MB] : ringB[1 - c][(t - lagx - offB[r]) & MB];
            const double grad = err * h;
            gB[c][r] = 0.95 * gB[c][r] + (1.0 - 0.95) * (grad * grad);
            wB[c][r] += (muB * grad * rsq_emul2(gB[c][r] + 0.001, tab, bits));
          }
        }
        if (vM) {
          const double h = ownM ? ringM[c][(t - offM) & 511] : ringM[1 - c][(t - lagx - offM) & 511];
          const double grad = errL[c][kM] * h;
          const double g = 0.95 * lmM[2 + c][tid] + (1.0 - 0.95) * (grad * grad);
          lmM[2 + c][tid] = g;
          lmM[c][tid] = lmM[c][tid] + (muM * grad * rsq_emul2(g + 0.001, tab, bits));
        }
        if (vS || vT) {
          const double h = vS ? (ownS ? ringS[c][(t - offS) & 63] : ringS[1 - c][(t - lagx - offS) & 63])
                              : (ownT ? ringT[c][(t - offT) & 15] : ringT[1 - c][(t - lagx - offT) & 15]);
          const double grad = (vS ? errL[c][kS] : errL[c][kT]) * h;
          const double g = 0.95 * lmS[2 + c][lane] + (1.0 - 0.95) * (grad * grad);
          lmS[2 + c][lane] = g;
          lmS[c][lane] = lmS[c][lane] + ((vS ? muS : muT) * grad * rsq_emul2(g + 0.001, tab, bits));
        }
      }
      if (wave == 6 || wave == 8) {  // BlendErr::Update((int32_t)val0) (blend.h:30-46): channel 0 on wave 6, channel 1 on wave 8
        const int c = wave == 6 ? 0 : 1;
        {
          const double val = vh[c][t & 63];
          const double e = val - (lane == OL_N ? blS[6 + c][lane] : blS[c][lane]);
          const double se = 0.98 * blS[2 + c][lane] + (1. - 0.98) * (e * e);
          blS[2 + c][lane] = se;
          const double tt = 10000. / (se + 0.0001);
          double wv = tt * tt, wsum = 0;
#pragma unroll
          for (int i = 0; i < OL_N; i++) wsum += bcast(wv, i);
          if (wsum > 0) wv /= wsum;
          blS[4 + c][lane] = wv;
        }
      }
      if (wave == 7) {  // RLS::Update(val[ch][1]) (rls.cpp:80-108), unit = channel
        double h[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) h[k] = rlS[5 + 4 * (ru & 1) + k][lane];
        const double p = rlS[13 + (ru & 1)][lane], cx1 = rlS[15 + (ru & 1)][lane];
        double rP = rlS[0][lane];
        double pc[NS];  // column j of P (before the update): P[k][j]
#pragma unroll
        for (int k = 0; k < NS; k++) pc[k] = __shfl(rP, base16 + 4 * k + rj);
        const double pr = rP * pick<NS>(h, rj);  // P[i][j]*hist[j]
        double vt = 0.;                          // vt1[i] = sum_j P[i][j]*hist[j]
        vt += quad_bcast<0>(pr);
        vt += quad_bcast<1>(pr);
        vt += quad_bcast<2>(pr);
        if (NS > 3) vt += quad_bcast<3>(pr);
        double v1[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) v1[k] = __shfl(vt, base16 + 4 * k);
        double inner = 0.;
#pragma unroll
        for (int k = 0; k < NS; k++) inner += h[k] * v1[k];
        const double sc = 1. / (ralpha + inner);
        double kk[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) kk[k] = sc * v1[k];
        const double ki = pick<NS>(kk, ri);
        double m2 = 0.;  // m2[i][j] = sum_k (k[i]*hist[k])*P[k][j]
#pragma unroll
        for (int k = 0; k < NS; k++) m2 += (ki * h[k]) * pc[k];
        rlS[0][lane] = ria * (rP - m2);
        const double dv = cx1 - p;
#pragma unroll
        for (int k = 0; k < NS; k++) rlS[1 + k][lane] = rlS[1 + k][lane] + dv * kk[k];
      }
    }
#pragma unroll
    for (int u = 0; u < NCE; u++) {
      const int ci = cpack[u] & 255, cj = (cpack[u] >> 8) & 255, cch = (cpack[u] >> 16) & 1;
      const bool isc = (cpack[u] >> 17) & 1, vC = (cpack[u] >> 18) & 1;
      if (vC) {  // AutoCov::UpdateCov / UpdateB (cov.h:19-43)
        const double hi = lpc_h(cch, ci, t);
        const double hj = isc ? lpc_h(cch, cj, t) : vh[cch][t & 63];
        if (isc) cv[u] = alpha * cv[u] + (hi * hj);
        else cv[u] = alpha * cv[u] + (hj * hi);
      }
    }
    UN_BUSY(pwD)
    UN_TICK(pfD)
    if (((t + 1) % KS) == 0) {  // lpc.h:213-219 / :277-283
#pragma unroll
      for (int u = 0; u < NCE; u++)
        if ((cpack[u] >> 18) & 1) (&snapL[0][0])[tid + UN_NT * u] = cv[u];  // (element e of channel e / OL_SNAP: the two snapshots are contiguous)
      __syncthreads();
      if (wave < 2) chol_wave_call<TYPE == 0, OL_N>(snapL[wave], solL[wave], lane);  // "matrix indefinit": the previous solution stays
      __syncthreads();
      UN_TICK(pfK)
    }
    }
  }
  __syncthreads();
#ifdef UN_PROF
  if (tid == 0 && n >= 256) { vbuf[0][0] = (int32_t)(pfA >> 10); vbuf[0][1] = (int32_t)(pfB >> 10); vbuf[0][2] = (int32_t)(pfC >> 10);
    vbuf[0][3] = (int32_t)(pfD >> 10); vbuf[0][4] = (int32_t)(pfK >> 10); }
  if (lane == 0 && n >= 256) { vbuf[1][wave] = (int32_t)(pwB >> 10); vbuf[1][16 + wave] = (int32_t)(pwD >> 10); }
  __syncthreads();
#endif
  if (n > 0) {
    const int tb = (n - 1) & ~(UN_IO - 1);
    for (int k = tid; k < 2 * UN_IO; k += UN_NT) {
      const int tau = tb + (k & (UN_IO - 1));
      if (tau < n) (k < UN_IO ? job.v[0] : job.v[1])[tau] = (&vbuf[0][0])[k];
    }
  }
}

extern "C" int sacopt_launch_unpredict(int type, const UnJob *jobs, int njobs, const uint16_t *tab, int bits, hipStream_t st) {
  if (njobs <= 0) return 0;
  if (type == 0) hipLaunchKernelGGL(k_unpredict<0>, dim3(njobs), dim3(UN_NT), 0, st, jobs, tab, bits);
  else if (type == 1) hipLaunchKernelGGL(k_unpredict<1>, dim3(njobs), dim3(UN_NT), 0, st, jobs, tab, bits);
  else if (type == 2) hipLaunchKernelGGL(k_unpredict<2>, dim3(njobs), dim3(UN_NT), 0, st, jobs, tab, bits);
  else if (type == 3) hipLaunchKernelGGL(k_unpredict<3>, dim3(njobs), dim3(UN_NT), 0, st, jobs, tab, bits);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
