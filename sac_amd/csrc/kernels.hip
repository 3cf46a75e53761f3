// gfx950 (MI355X) kernels for sac's predictor cascade + optimizer cost.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (contraction OFF: the reference is
// plain SSE2 mul/add). FP64 everywhere, reference operation order.
//
// Decomposition: the cascade of src/profile.cpp:113-186 is feed-forward between
// stages (no stage reads a later stage's state), so every stage is its own time-serial kernel
// over a window, one workgroup per (frame, candidate) chain:
//     k_lpc3  -> k_lms<1280,640> -> k_lms<256,128> -> k_lms<32,16> -> k_lms<8,4> -> k_mix -> k_cost
//
// Exact serial sums: the reference's dot products are strict left-to-right FP64 chains
// (src/pred/nlms.h:110-111). v_mfma_f64_4x4x4 with B == 1.0 computes, per output row,
//     D = (((C + a0) + a1) + a2) + a3      with IEEE rounding after every add
// (an FMA chain whose multiplier is exactly 1; verified on hardware, tools/ubench.hip, and by
// the parity tests), at 5.8 cycles/term against 14 for a dependent v_add_f64 -- so the long
// chains run on the matrix pipe as *adders*, not as a GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jobs.h"

// ------------------------------------------------------------------------------------------
// x86 rsqrtss emulation (src/pred/nlms.h:34-57): f64 -> f32 (RNE), table on exponent parity x
// top `bits` mantissa bits, exponent rescale, f32 -> f64.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double rsq_emul(double a, const float *tab, int bits) {
  const float f = (float)a;
  const uint32_t u = __float_as_uint(f);
  const uint32_t e = (u >> 23) & 0xffu, mant = u & 0x7fffffu;
  const uint32_t par = e & 1u;
  const uint32_t idx = (par << bits) | (mant >> (23 - bits));
  uint32_t tb = __float_as_uint(tab[idx]);
  const int q = ((int)e - (par ? 127 : 128)) >> 1;  // e - ec is even: exact
  tb -= (uint32_t)(q << 23);
  if (e == 255u) tb = mant ? (u | 0x400000u) : 0u;  // NaN stays NaN, +inf -> +0
  return (double)__uint_as_float(tb);
}

__device__ __forceinline__ double bcast(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------
// LMSADA2 stereo pair (src/pred/nlms.h:96-139), driven as StereoHigh/StereoNormal drive it
// (src/profile.cpp:126-148,176-186):
//   pred0(t) = lms[0].Predict(x1(t-1));  pred1(t) = lms[1].Predict(x0(t));  lms[c].Update(xc(t))
// One workgroup per chain. Taps are spread over the threads (w, eg in registers), the history
// is a ring of stage inputs in LDS (shifts of nlms.h:109,130 become index arithmetic), products
// go to LDS in the operand layout of the f64 MFMA, wave 0 sums them on the matrix pipe.
// ------------------------------------------------------------------------------------------
template <int S, int D, int NT, int R, int RL>
__global__ __launch_bounds__(NT) void k_lms(const LmsJob *__restrict__ jobs, const float *__restrict__ gtab,
                                            int bits) {
  constexpr int T = S + D, T4 = T / 4, PSTR = T4 + 1, M = RL - 1, CH = LMS_CHUNK;
  static_assert(T % 4 == 0, "tap count must be a multiple of 4");
  static_assert(RL >= CH + (S > D ? S : D) + 1, "ring too short");
  static_assert(NT * R >= 2 * T, "not enough tap slots");
  constexpr int U = (T4 % 8 == 0) ? 8 : (T4 % 4 == 0) ? 4 : (T4 % 3 == 0) ? 3 : 1;
  __shared__ double ring[2][RL];
  __shared__ double prods[2][4][PSTR];
  __shared__ double predv[2];
  __shared__ float tab[RSQ_LDS_MAX];

  const LmsJob job = jobs[blockIdx.x];
  const int tid = threadIdx.x, n = job.n;
  for (int i = tid; i < (2 << bits); i += NT) tab[i] = gtab[i];
  for (int i = tid; i < 2 * RL; i += NT) (&ring[0][0])[i] = 0.;
  __syncthreads();

  // per-slot constants
  double w[R], eg[R], mu[R];
  int hoff[R], hsel[R], paddr[R], cs[R];
  bool valid[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int g = r * NT + tid;
    valid[r] = g < 2 * T;
    const int c = (g >= T) ? 1 : 0, i = g - c * T;
    cs[r] = c;
    int src, lag;
    if (i < S) { src = c; lag = 1 + i; mu[r] = job.mu1; }
    else { const int j = i - S; src = 1 - c; lag = (c == 0 ? 1 : 0) + j; mu[r] = job.mu2; }
    hsel[r] = src;
    hoff[r] = (-lag) & M;
    paddr[r] = (c * 4 + (i & 3)) * PSTR + (i >> 2);
    w[r] = 0.; eg[r] = 0.;
    if (!valid[r]) { hsel[r] = 0; hoff[r] = 0; paddr[r] = 0; }
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int mk = lane >> 4, mc = lane & 15;
  const bool mact = mc < 2;
  const double *mp = &prods[mact ? mc : 0][mk][0];

  for (int t = 0; t < n; t++) {
    if ((t & (CH - 1)) == 0) {
      for (int idx = tid; idx < 2 * CH; idx += NT) {
        const int c = idx / CH, k = idx - c * CH, tau = t + k;
        const double *xs = c ? job.x[1] : job.x[0];
        ring[c][tau & M] = (tau < n) ? xs[tau] : 0.;
      }
      __syncthreads();
    }
    // ---- products w[i]*hist[i] (nlms.h:111)
    double h[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      h[r] = ring[hsel[r]][(t + hoff[r]) & M];
      if (valid[r]) (&prods[0][0][0])[paddr[r]] = w[r] * h[r];
    }
    __syncthreads();
    // ---- pred = 0.; for i: pred += w[i]*hist[i]  -- strict order, on the matrix pipe
    if (wave == 0) {
      double acc = 0.;
#pragma unroll 1
      for (int s0 = 0; s0 < T4; s0 += U) {
        double a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = mact ? mp[s0 + u] : 0.;
#pragma unroll
        for (int u = 0; u < U; u++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], 1.0, acc, 0, 0, 0);
      }
      // chain c = 4*blk + i lives in D lanes 16*i + 4*blk + j: chain 0 -> lane 0, chain 1 -> lane 16
      if (lane == 0 || lane == 16) {
        const int c = lane >> 4;
        predv[c] = acc;
        const double xc = ring[c][t & M];
        double *po = c ? job.p[1] : job.p[0], *xo = c ? job.xn[1] : job.xn[0];
        po[t] = acc;
        if (xo) xo[t] = xc - acc;  // CalcError (profile.cpp:141-148)
      }
    }
    __syncthreads();
    // ---- Update (nlms.h:114-129)
    const double e0 = ring[0][t & M] - predv[0];
    const double e1 = ring[1][t & M] - predv[1];
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (valid[r]) {
        const double err = cs[r] ? e1 : e0;
        const double grad = err * h[r];
        eg[r] = 0.95 * eg[r] + (1.0 - 0.95) * (grad * grad);
        w[r] += (mu[r] * grad * rsq_emul(eg[r] + 0.001, tab, bits));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// LPC3(alpha, S=16, D=8, k=8, regp=500) (src/pred/lpc.h:172-228) incl. AutoCov (math/cov.h),
// Cholesky all-orders solve (math/cholesky.cpp:16-77) and BlendErr (pred/blend.h:7-53).
// The stage only ever sees input samples (no feedback from later stages), so both channels
// run independently: one wave per (chain, channel); lane o owns order o / matrix row o.
//   ch0: Predict(v1(t-1)) ; ch1: Predict(v0(t))       (profile.cpp:128, :150-158)
// ------------------------------------------------------------------------------------------
#define LPC_N 24
#define LPC_S 16
#define LPC_RL 256
#define LPC_CH 128
__device__ __forceinline__ int32_t s2u(int32_t v) { return v < 0 ? 2 * (-v) : (v > 0 ? 2 * v - 1 : 0); }

// FAST == false: LPC3 (k=8, regp=500, all orders + BlendErr) writing p[0] and val[ch][1].
// FAST == true : LPC2 (src/pred/lpc.h:254-293; k=32, regp=0, full order only) as StereoFast
//                uses it (src/profile.cpp:22-32), writing the residual directly.
template <bool FAST>
__global__ __launch_bounds__(64) void k_lpc(const LpcJob *__restrict__ jobs) {
  constexpr int N = LPC_N, S = LPC_S, RL = LPC_RL, M = RL - 1, CH = LPC_CH;
  constexpr int KSOLVE = FAST ? 32 : 8;
  constexpr double REGP = FAST ? 0. : 500.;
  __shared__ double ring[2][RL];
  __shared__ double cm[N + 1][N];
  const LpcJob job = jobs[blockIdx.x];
  const int ch = blockIdx.y, oc = 1 - ch, lagb = (ch == 0) ? 1 : 0;
  const int lane = threadIdx.x, n = job.n;
  const double alpha = job.alpha;
  const int32_t *s0 = job.s[0], *s1 = job.s[1];
  double *pout = ch ? job.p[1] : job.p[0], *xout = ch ? job.x1[1] : job.x1[0];
  int32_t *eout = ch ? job.e[1] : job.e[0], *uout = ch ? job.u[1] : job.u[0];
  for (int i = lane; i < 2 * RL; i += 64) (&ring[0][0])[i] = 0.;
  for (int i = lane; i < (N + 1) * N; i += 64) (&cm[0][0])[i] = 0.;

  // covariance slots: element e = lane + 64 r of the packed lower triangle -> (j, i<=j)
  double cv[5];
  int cj[5], ci[5], ai[5], aj[5];  // (ring << 16) | offset of hist[i], hist[j] relative to t
  bool cval[5];
  auto haddr = [&](int i) -> int {
    if (i < S) return ((-(1 + i)) & M) | (ch << 16);
    return ((-(lagb + (i - S))) & M) | (oc << 16);
  };
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int e = lane + 64 * r;
    cval[r] = e < N * (N + 1) / 2;
    int j = 0;
    while ((j + 1) * (j + 2) / 2 <= e) j++;
    const int i = e - j * (j + 1) / 2;
    cj[r] = cval[r] ? j : 0; ci[r] = cval[r] ? i : 0;
    cv[r] = (cval[r] && i == j) ? 1.0 : 0.0;  // cov.h:17
    ai[r] = haddr(ci[r]); aj[r] = haddr(cj[r]);
  }
  const int ab = haddr(lane < N ? lane : 0);
  auto hist_at = [&](int a, int t) -> double { return ring[a >> 16][(t + (a & 0xffff)) & M]; };

  double bv = 0., serr = 0., bw = 0.;
  double sol[N];
#pragma unroll
  for (int i = 0; i < N; i++) sol[i] = 0.;
  int nk = 0;
  int32_t umax = 0;

  for (int t = 0; t < n; t++) {
    if ((t & (CH - 1)) == 0) {
      __syncthreads();
      for (int idx = lane; idx < 2 * CH; idx += 64) {
        const int c = idx / CH, k = idx - c * CH, tau = t + k;
        const int32_t *ss = c ? s1 : s0;
        ring[c][tau & M] = (tau < n) ? (double)ss[tau] : 0.;
      }
      __syncthreads();
    }
    const double v = ring[ch][t & M];
    double ss = 0.;
    if constexpr (!FAST) {
      // ---- LPC3::Predict (lpc.h:186-201): all orders, then BlendErr::Predict (blend.h:13-29)
#pragma unroll
      for (int i = 0; i < N; i++) {
        const double hi = (i < S) ? ring[ch][(t - 1 - i) & M] : ring[oc][(t - lagb - (i - S)) & M];
        if (i <= lane) ss += sol[i] * hi;
      }
      const double q = bw * ss;
      double pm = 0.;
#pragma unroll
      for (int i = 0; i < N; i++) pm += bcast(q, i);
      if (lane == 0) {
        pout[t] = pm;
        xout[t] = v - pm;  // val[ch][1] = val[ch][0] - p[0] (profile.cpp:144)
      }
      // BlendErr::Update (blend.h:30-46); serr[n] is dead state and not kept
      const double e = (v - ss);
      serr = 0.98 * serr + (1. - 0.98) * (e * e);
      const double tt = 10000. / (serr + 0.0001);
      double wv = tt * tt;
      double wsum = 0;
#pragma unroll
      for (int i = 0; i < N; i++) wsum += bcast(wv, i);
      if (wsum > 0) wv /= wsum;
      bw = (lane < N) ? wv : 0.;
    } else {
      // ---- LPC2::Predict (lpc.h:261-267); sol is wave-uniform
#pragma unroll
      for (int i = 0; i < N; i++) {
        const double hi = (i < S) ? ring[ch][(t - 1 - i) & M] : ring[oc][(t - lagb - (i - S)) & M];
        ss += sol[i] * hi;
      }
      if (lane == 0) {  // err = val - round(p) (profile.cpp:30-31)
        const int32_t e = (int32_t)(v - round(ss));
        const int32_t ue = s2u(e);
        if (eout) eout[t] = e;
        if (uout) uout[t] = ue;
        umax = ue > umax ? ue : umax;
      }
    }
#pragma unroll
    for (int r = 0; r < 5; r++)  // AutoCov::UpdateCov (cov.h:19-31)
      if (cval[r]) cv[r] = alpha * cv[r] + (hist_at(ai[r], t) * hist_at(aj[r], t));
    if (lane < N) bv = alpha * bv + (v * hist_at(ab, t));  // UpdateB (cov.h:32-43)

    if (++nk >= KSOLVE) {
      nk = 0;
      // gather row `lane` of the covariance (row 24 = b) through LDS
#pragma unroll
      for (int r = 0; r < 5; r++)
        if (cval[r]) cm[cj[r]][ci[r]] = cv[r];
      if (lane < N) cm[N][lane] = bv;
      __syncthreads();
      double m[N];
      const int row = lane <= N ? lane : 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
        m[k] = cm[row][k];
        if (k == lane) m[k] += REGP;  // regp (cholesky.cpp:20)
      }
      // Cholesky::Factor (cholesky.cpp:16-35), column by column; lane 24 carries CalcY (:37-46)
      bool fail = false;
#pragma unroll
      for (int j = 0; j < N; j++) {
        if (!fail) {
          double sum = m[j];
#pragma unroll
          for (int k = 0; k < j; k++) sum -= (m[k] * bcast(m[k], j));
          const double dj = bcast(sum, j);
          if (dj > 0.) {
            const double diag = sqrt(dj);
            m[j] = (lane == j) ? diag : sum / diag;
          } else fail = true;  // "matrix indefinit": keep the previous solution (lpc.h:216-218)
        }
      }
      if (!fail) {
        if constexpr (!FAST) {  // Cholesky::Solve(b) for every leading order (cholesky.cpp:63-77)
#pragma unroll
          for (int i = N - 1; i >= 0; i--) {
            double sum = bcast(m[i], N);
#pragma unroll
            for (int j = i + 1; j < N; j++) {
              const double mji = bcast(m[i], j);
              if (j <= lane) sum -= (mji * sol[j]);
            }
            const double d = bcast(m[i], i);
            if (i <= lane) sol[i] = sum / d;
          }
        } else {  // Cholesky::Solve(b, x) (cholesky.cpp:48-60), computed uniformly by every lane
#pragma unroll
          for (int i = N - 1; i >= 0; i--) {
            double sum = bcast(m[i], N);
#pragma unroll
            for (int j = i + 1; j < N; j++) sum -= (bcast(m[i], j) * sol[j]);
            sol[i] = sum / bcast(m[i], i);
          }
        }
      }
    }
  }
  if constexpr (FAST)
    if (lane == 0 && job.umax) job.umax[ch] = umax;
}

// ------------------------------------------------------------------------------------------
// Stage mixer + residual: RLS(NS, alpha) (src/pred/rls.cpp:61-108) over the cumulative stage sums
// (profile.cpp:134), prediction rounding and residual (profile.cpp:152-157), S2U and max
// (codec.cpp:188-211). The two channels' mixers are independent: one lane per (chain, channel).
// ------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(64) void k_mix(const MixJob *__restrict__ jobs, int njobs) {
  const int gid = blockIdx.x * 64 + threadIdx.x;
  if (gid >= 2 * njobs) return;
  const MixJob &job = jobs[gid >> 1];
  const int ch = gid & 1, n = job.n;
  const double alpha = job.alpha;
  const int32_t *s = job.s[ch];
  const double *p0 = job.p0[ch], *x1 = job.x1[ch];
  const double *pk[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) pk[k] = job.pk[k][ch];
  int32_t *eo = job.e[ch], *uo = job.u[ch];
  double w[NS], P[NS][NS];
#pragma unroll
  for (int i = 0; i < NS; i++) {
    w[i] = 0.;
#pragma unroll
    for (int j = 0; j < NS; j++) P[i][j] = (i == j) ? 1. : 0.;
  }
  int32_t umax = 0;
  for (int t = 0; t < n; t++) {
    double hist[NS];
    hist[0] = pk[0][t];
#pragma unroll
    for (int k = 1; k < NS; k++) hist[k] = hist[k - 1] + pk[k][t];  // pv (profile.cpp:134)
    double p = 0.;
#pragma unroll
    for (int i = 0; i < NS; i++) p += hist[i] * w[i];  // RLS::Predict (rls.cpp:67-72)
    const int32_t v = s[t];
    const int32_t q = (int32_t)round(p0[t] + p);
    const int32_t e = v - q;
    if (eo) eo[t] = e;
    const int32_t ue = s2u(e);
    if (uo) uo[t] = ue;
    umax = ue > umax ? ue : umax;
    // RLS::Update(val[ch][1]) (rls.cpp:80-108)
    double vt1[NS], kk[NS], m2[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
      double sum = 0.;
#pragma unroll
      for (int j = 0; j < NS; j++) sum += P[i][j] * hist[j];
      vt1[i] = sum;
    }
    double inner = 0.;
#pragma unroll
    for (int i = 0; i < NS; i++) inner += hist[i] * vt1[i];
    const double sc = 1. / (alpha + inner);
#pragma unroll
    for (int i = 0; i < NS; i++) kk[i] = sc * vt1[i];
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
      for (int j = 0; j < NS; j++) {
        double sum = 0.;
#pragma unroll
        for (int k = 0; k < NS; k++) sum += (kk[i] * hist[k]) * P[k][j];
        m2[i][j] = sum;
      }
    const double ia = 1. / alpha;
#pragma unroll
    for (int i = 0; i < NS; i++)
#pragma unroll
      for (int j = 0; j < NS; j++) P[i][j] = ia * (P[i][j] - m2[i][j]);
    const double dv = x1[t] - p;
#pragma unroll
    for (int i = 0; i < NS; i++) w[i] = w[i] + dv * kk[i];
  }
  if (job.umax) job.umax[ch] = umax;
}

// ------------------------------------------------------------------------------------------
// CostGolomb::Calc (src/codec.cpp:315-339) over a list of residual buffers in call order, the
// running mean carried from buffer to buffer exactly as one CostGolomb instance does (:364).
// One wave per frame: lane 0 walks the RunExp recurrence (the only true serial dependency),
// all lanes do the divisions / bit counts. nbits is a sum of small integers, exact in FP64 in
// any order, so it is accumulated in int64.
// ------------------------------------------------------------------------------------------
#define COST_BLK 512
__global__ __launch_bounds__(64) void k_cost(const CostJob *__restrict__ jobs, const int *__restrict__ pow2_bits) {
  __shared__ double cval[COST_BLK];
  __shared__ int32_t uval[COST_BLK];
  __shared__ int32_t mint[COST_BLK];
  __shared__ long long red[64];
  const CostJob job = jobs[blockIdx.x];
  const int lane = threadIdx.x, n = job.n;
  double mean = job.state ? *job.state : 0.;
  const bool single = job.nbuf < 0;
  const int nbuf = single ? 1 : job.nbuf;
  for (int b = 0; b < nbuf; b += 2) {
    double c[2] = {0., 0.};
    for (int h = 0; h < (single ? 1 : 2); h++) {
      const int32_t *e = job.bufs[b + h];
      long long nb = 0;
      for (int b0 = 0; b0 < n; b0 += COST_BLK) {
        const int cnt = min(COST_BLK, n - b0);
        for (int k = lane; k < cnt; k += 64) {
          const int32_t val = s2u(e[b0 + k]);
          uval[k] = val;
          cval[k] = (1. - 0.98) * val;  // RunExp::Update (utils.h:13-15)
        }
        __syncthreads();
        if (lane == 0) {
          for (int k = 0; k < cnt; k++) {
            const int mi = (int)mean;
            mint[k] = mi > 1 ? mi : 1;
            mean = 0.98 * mean + cval[k];
          }
        }
        __syncthreads();
        for (int k = lane; k < cnt; k += 64) {
          const int m = mint[k];
          nb += uval[k] / m + 1;
          if (m > 1) {  // ceil(log(m)/log(2)) with the host libm's value at exact powers of two
            const int fl = 31 - __clz(m);
            nb += ((m & (m - 1)) == 0) ? pow2_bits[fl] : fl + 1;
          }
        }
        __syncthreads();
      }
      red[lane] = nb;
      __syncthreads();
      if (lane == 0) {
        long long tot = 0;
        for (int i = 0; i < 64; i++) tot += red[i];
        c[h] = n ? (double)tot / (double)n : 0.;
      }
      __syncthreads();
    }
    if (lane == 0) job.costs[b >> 1] = single ? c[0] : (c[0] + c[1]) / 2.;  // GetCost (codec.cpp:349-351)
  }
  if (lane == 0 && job.state) *job.state = mean;
}

// single buffer variant used by sacopt_cost_golomb / meanrms
__global__ void k_meanrms(const int32_t *e, int n, double *out) {
  // CostMeanRMS::Calc (codec.cpp:302-312): sum of |e| as integers (exact in FP64), / n
  __shared__ long long red[256];
  long long s = 0;
  for (int i = threadIdx.x; i < n; i += 256) { const int v = e[i]; s += v < 0 ? -(long long)v : v; }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long tot = 0;
    for (int i = 0; i < 256; i++) tot += red[i];
    *out = n ? (double)tot / (double)n : 0.;
  }
}

// RLS stand-alone driver (sacopt_rls_run): p[t] = Predict(pv[t]); Update(val[t]).
template <int NS>
__global__ void k_rls(const double *pv, const double *val, int len, double alpha, double *pout) {
  if (threadIdx.x || blockIdx.x) return;
  double w[NS], P[NS][NS];
  for (int i = 0; i < NS; i++) { w[i] = 0.; for (int j = 0; j < NS; j++) P[i][j] = (i == j) ? 1. : 0.; }
  for (int t = 0; t < len; t++) {
    double hist[NS];
    for (int k = 0; k < NS; k++) hist[k] = pv[t * NS + k];
    double p = 0.;
    for (int i = 0; i < NS; i++) p += hist[i] * w[i];
    pout[t] = p;
    double vt1[NS], kk[NS], m2[NS][NS];
    for (int i = 0; i < NS; i++) { double sum = 0.; for (int j = 0; j < NS; j++) sum += P[i][j] * hist[j]; vt1[i] = sum; }
    double inner = 0.;
    for (int i = 0; i < NS; i++) inner += hist[i] * vt1[i];
    const double sc = 1. / (alpha + inner);
    for (int i = 0; i < NS; i++) kk[i] = sc * vt1[i];
    for (int i = 0; i < NS; i++)
      for (int j = 0; j < NS; j++) { double sum = 0.; for (int k = 0; k < NS; k++) sum += (kk[i] * hist[k]) * P[k][j]; m2[i][j] = sum; }
    const double ia = 1. / alpha;
    for (int i = 0; i < NS; i++) for (int j = 0; j < NS; j++) P[i][j] = ia * (P[i][j] - m2[i][j]);
    const double dv = val[t] - p;
    for (int i = 0; i < NS; i++) w[i] = w[i] + dv * kk[i];
  }
}

// ------------------------------------------------------------------------------------------
// host-callable launchers (sacopt.cpp is compiled by the host compiler pass of the same TU set)
// ------------------------------------------------------------------------------------------
extern "C" int sacopt_launch_lms(int S, int D, const LmsJob *jobs, int njobs, const float *tab, int bits,
                                 hipStream_t st) {
  if (njobs <= 0) return 0;
  if (S == 1280 && D == 640) hipLaunchKernelGGL((k_lms<1280, 640, 1024, 4, 2048>), dim3(njobs), dim3(1024), 0, st, jobs, tab, bits);
  else if (S == 256 && D == 128) hipLaunchKernelGGL((k_lms<256, 128, 256, 3, 1024>), dim3(njobs), dim3(256), 0, st, jobs, tab, bits);
  else if (S == 32 && D == 16) hipLaunchKernelGGL((k_lms<32, 16, 64, 2, 1024>), dim3(njobs), dim3(64), 0, st, jobs, tab, bits);
  else if (S == 8 && D == 4) hipLaunchKernelGGL((k_lms<8, 4, 64, 1, 1024>), dim3(njobs), dim3(64), 0, st, jobs, tab, bits);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_lpc3(const LpcJob *jobs, int njobs, hipStream_t st) {
  if (njobs <= 0) return 0;
  hipLaunchKernelGGL(k_lpc<false>, dim3(njobs, 2), dim3(64), 0, st, jobs);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_lpc2(const LpcJob *jobs, int njobs, hipStream_t st) {
  if (njobs <= 0) return 0;
  hipLaunchKernelGGL(k_lpc<true>, dim3(njobs, 2), dim3(64), 0, st, jobs);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_mix(int ns, const MixJob *jobs, int njobs, hipStream_t st) {
  if (njobs <= 0) return 0;
  const int blocks = (2 * njobs + 63) / 64;
  if (ns == 4) hipLaunchKernelGGL((k_mix<4>), dim3(blocks), dim3(64), 0, st, jobs, njobs);
  else if (ns == 3) hipLaunchKernelGGL((k_mix<3>), dim3(blocks), dim3(64), 0, st, jobs, njobs);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_cost(const CostJob *jobs, int njobs, const int *pow2_bits, hipStream_t st) {
  if (njobs <= 0) return 0;
  hipLaunchKernelGGL(k_cost, dim3(njobs), dim3(64), 0, st, jobs, pow2_bits);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_meanrms(const int32_t *e, int n, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_meanrms, dim3(1), dim3(256), 0, st, e, n, out);
  return hipGetLastError() != hipSuccess;
}
extern "C" int sacopt_launch_rls(int ns, const double *pv, const double *val, int len, double alpha, double *p,
                                 hipStream_t st) {
  if (ns == 4) hipLaunchKernelGGL((k_rls<4>), dim3(1), dim3(64), 0, st, pv, val, len, alpha, p);
  else if (ns == 3) hipLaunchKernelGGL((k_rls<3>), dim3(1), dim3(64), 0, st, pv, val, len, alpha, p);
  else return 1;
  return hipGetLastError() != hipSuccess;
}
