// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// Harness around the *real* reference (autch/sac v0.0.4), compiled from the sources where
// they lie under /root/reference/src (see oracle/Makefile; nothing is copied into this repo).
// It exists so that the CPU restatement in oracle/sac_oracle.c -- and through it the HIP path --
// can be pinned against the reference's own private hot-path entry points:
//   FrameCoder::PredictStereoFrame  (src/codec.cpp:154-214)
//   FrameCoder::GetCost             (src/codec.cpp:341-352)
//   FrameCoder::Optimize            (src/codec.cpp:354-408)
//   CostGolomb::Calc                (src/codec.cpp:315-339)
//   StereoHigh/StereoNormal         (src/profile.cpp:46-186)
//   LMSADA2 / LPC3 / LPC2 / RLS     (src/pred/nlms.h:96, src/pred/lpc.h:172,254, src/pred/rls.cpp)
//
// The reference keeps these private and defines CostGolomb inside codec.cpp, so this TU
// includes codec.cpp itself after `#define private public` (layout is unchanged by that).
// Only available in the build container (the reference does not travel to the GPU box as
// source; the built oracle/_ref/libsacref.so does).
#include <vector>
#include <string>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <ctime>
#include <iostream>
#include <sstream>
#include <algorithm>
#include <fstream>
#include <iomanip>
#include <chrono>
#include <cstring>

#define private public
#define protected public
#include "codec.cpp"
#undef private
#undef protected

namespace {
struct CoutMute {
  std::streambuf *old;
  std::ostringstream sink;
  CoutMute() { old = std::cout.rdbuf(sink.rdbuf()); }
  ~CoutMute() { std::cout.rdbuf(old); }
};

void load_frame(FrameCoder &fc, const int32_t *s0, const int32_t *s1, int total) {
  for (int i = 0; i < total; i++) { fc.samples[0][i] = s0[i]; fc.samples[1][i] = s1[i]; }
  fc.SetNumSamples(total);
  // FrameCoder::Predict (codec.cpp:448-452): remap tables are built before prediction
  for (int ch = 0; ch < 2; ch++) {
    fc.framestats[ch].mymap.Reset();
    fc.framestats[ch].mymap.Analyse(&(fc.samples[ch][0]), total);
  }
}
void set_coefs(SacProfile &p, const double *coefs, int ncoef) {
  if (!coefs) return;
  for (int i = 0; i < ncoef && i < (int)p.coefs.size(); i++) p.coefs[i].vdef = coefs[i];
}
}  // namespace

extern "C" {

float ref_rsqrt(float x) { return rsqrt(x); }

int ref_cholesky_test() { Cholesky c(4); return c.Test() ? 1 : 0; }

// The reference's Cholesky class on one matrix: snap = packed lower triangle + b[n] (layout of oracle_chol_solve).
int ref_chol_solve(int n, double regp, int fast, const double *snap, double *sol) {
  Cholesky c(n);
  c.regp = regp;
  Matrix A(n);
  for (int r = 0, k = 0; r < n; r++)
    for (int q = 0; q <= r; q++, k++) { A[r][q] = snap[k]; A[q][r] = snap[k]; }
  std::vector<double> b(snap + n * (n + 1) / 2, snap + n * (n + 1) / 2 + n);
  if (c.Factor(A)) return 1;
  if (fast) { std::vector<double> x(n); c.Solve(b, x); for (int i = 0; i < n; i++) sol[i] = x[i]; }
  else {
    c.Solve(b);
    for (int o = 0; o < n; o++)
      for (int i = 0; i <= o; i++) sol[o * (o + 1) / 2 + i] = c.sol[o][i];
  }
  return 0;
}

// Base profile table (codec.cpp:8-30): writes ncoef x {vmin,vmax,vdef}, returns ncoef.
int ref_base_profile(int profile, double *out) {
  FrameCoder fc(2, 16, profile);
  int n = (int)fc.baseprofile.coefs.size();
  for (int i = 0; i < n; i++) {
    out[3 * i] = fc.baseprofile.coefs[i].vmin;
    out[3 * i + 1] = fc.baseprofile.coefs[i].vmax;
    out[3 * i + 2] = fc.baseprofile.coefs[i].vdef;
  }
  return n;
}

// PredictStereoFrame(profile, 0, 1, from, n) on a frame holding `total` samples.
// coefs==NULL -> base profile. maxbpn[4] = {maxbpn0, maxbpn1, maxbpn_map0, maxbpn_map1}.
void ref_predict_stereo_frame(int profile, const double *coefs, int ncoef, const int32_t *s0,
                              const int32_t *s1, int total, int from, int n, int32_t *e0,
                              int32_t *e1, int32_t *u0, int32_t *u1, int32_t *m0, int32_t *m1,
                              int *maxbpn) {
  CoutMute mute;
  FrameCoder fc(2, total, profile);
  load_frame(fc, s0, s1, total);
  SacProfile p = fc.baseprofile;
  set_coefs(p, coefs, ncoef);
  fc.PredictStereoFrame(p, 0, 1, from, n);
  for (int i = 0; i < n; i++) {
    if (e0) e0[i] = fc.error[0][i];
    if (e1) e1[i] = fc.error[1][i];
    if (u0) u0[i] = fc.s2u_error[0][i];
    if (u1) u1[i] = fc.s2u_error[1][i];
    if (m0) m0[i] = fc.s2u_error_map[0][i];
    if (m1) m1[i] = fc.s2u_error_map[1][i];
  }
  if (maxbpn) {
    maxbpn[0] = fc.framestats[0].maxbpn;
    maxbpn[1] = fc.framestats[1].maxbpn;
    maxbpn[2] = fc.framestats[0].maxbpn_map;
    maxbpn[3] = fc.framestats[1].maxbpn_map;
  }
}

// UnpredictStereoFrame(profile, 0, 1, n) (codec.cpp:216-235) on residuals error[ch][0..n).
// used0/used1: NULL = framestats[ch].enc_mapped false, else Remap::usedl[0..scale] then usedh[0..scale] (bytes);
// minmax = {minval0, maxval0, minval1, maxval1} = framestats[ch].minval/maxval (read by StereoHigh::Unpredict only;
// the reference's decoder never sets them, the caller does).
void ref_unpredict_stereo_frame(int profile, const double *coefs, int ncoef, const int32_t *e0, const int32_t *e1, int n,
                                const unsigned char *used0, const unsigned char *used1, const int32_t *minmax,
                                int32_t *v0, int32_t *v1) {
  CoutMute mute;
  FrameCoder fc(2, n, profile);
  fc.SetNumSamples(n);
  const unsigned char *used[2] = {used0, used1};
  for (int ch = 0; ch < 2; ch++) {
    auto &fs = fc.framestats[ch];
    fs.enc_mapped = used[ch] != nullptr;
    fs.mymap.Reset();
    if (used[ch]) {
      const int sc = fs.mymap.scale;
      for (int i = 0; i <= sc; i++) { fs.mymap.usedl[i] = used[ch][i] != 0; fs.mymap.usedh[i] = used[ch][sc + 1 + i] != 0; }
    }
    fs.minval = minmax ? minmax[2 * ch] : INT32_MIN;
    fs.maxval = minmax ? minmax[2 * ch + 1] : INT32_MAX;
  }
  for (int i = 0; i < n; i++) { fc.error[0][i] = e0[i]; fc.error[1][i] = e1[i]; }
  SacProfile p = fc.baseprofile;
  set_coefs(p, coefs, ncoef);
  fc.UnpredictStereoFrame(p, 0, 1, n);
  for (int i = 0; i < n; i++) { v0[i] = fc.samples[0][i]; v1[i] = fc.samples[1][i]; }
}

// CostGolomb::Calc with the carried running mean exposed (state_inout).
double ref_cost_golomb(const int32_t *buf, int n, double *state_inout) {
  CostGolomb c;
  c.mean_err.sum = *state_inout;
  double r = c.Calc(buf, n);
  *state_inout = c.mean_err.sum;
  return r;
}

double ref_cost_meanrms(const int32_t *buf, int n) {
  CostMeanRMS c;
  return c.Calc(buf, n);
}

// The real FrameCoder::Optimize on the base profile ("ref+opt": codec.cpp:455-456 un-commented).
// best[ncoef] receives the selected vdef values.
int ref_optimize(int profile, const int32_t *s0, const int32_t *s1, int numsamples, double *best) {
  CoutMute mute;
  FrameCoder fc(2, numsamples, profile);
  load_frame(fc, s0, s1, numsamples);
  SacProfile opt = fc.baseprofile;
  if (opt.coefs.size() < 8) return -1;  // Optimize indexes coefs[0..7] unconditionally (codec.cpp:389)
  fc.Optimize(opt);
  for (size_t i = 0; i < opt.coefs.size(); i++) best[i] = opt.coefs[i].vdef;
  return (int)opt.coefs.size();
}

// Same schedule as Optimize (codec.cpp:354-408) driven through the real GetCost with one
// CostGolomb instance, recording every (coef, x, cost) triple. ncoef_loop lets non-high
// profiles run the documented deviation (loop over coefs.size() instead of 8).
// trace: rows of 3 doubles {coef(-1 for the baseline call), x, cost}; returns number of rows.
static int optimize_trace_cf(CostFunction &CostFunc, int profile, const int32_t *s0, const int32_t *s1, int numsamples,
                             int ncoef_loop, double *best, double *trace, int max_rows);
int ref_optimize_trace(int profile, const int32_t *s0, const int32_t *s1, int numsamples,
                       int ncoef_loop, double *best, double *trace, int max_rows) {
  CostGolomb cf;
  return optimize_trace_cf(cf, profile, s0, s1, numsamples, ncoef_loop, best, trace, max_rows);
}
// ... with the reference's OTHER CostFunction, CostMeanRMS (codec.cpp:302-312; instantiated nowhere upstream)
int ref_optimize_trace_meanrms(int profile, const int32_t *s0, const int32_t *s1, int numsamples,
                               int ncoef_loop, double *best, double *trace, int max_rows) {
  CostMeanRMS cf;
  return optimize_trace_cf(cf, profile, s0, s1, numsamples, ncoef_loop, best, trace, max_rows);
}
static int optimize_trace_cf(CostFunction &CostFunc, int profile, const int32_t *s0, const int32_t *s1, int numsamples,
                             int ncoef_loop, double *best, double *trace, int max_rows) {
  CoutMute mute;
  FrameCoder fc(2, numsamples, profile);
  load_frame(fc, s0, s1, numsamples);
  SacProfile prof = fc.baseprofile;
  int samples_to_optimize = numsamples / 8;
  int start = (numsamples - samples_to_optimize) / 2;
  int rows = 0;
  double best_cost = fc.GetCost(prof, &CostFunc, 0, prof.coefs[0].vdef, start, samples_to_optimize);
  if (rows < max_rows) { trace[3 * rows] = -1; trace[3 * rows + 1] = prof.coefs[0].vdef; trace[3 * rows + 2] = best_cost; rows++; }
  for (int i = 0; i < ncoef_loop; i++) {
    double best_x = prof.coefs[i].vdef;
    double x = prof.coefs[i].vmin;
    double xinc = (prof.coefs[i].vmax - prof.coefs[i].vmin) / 10.;
    while (x <= prof.coefs[i].vmax) {
      double cost = fc.GetCost(prof, &CostFunc, i, x, start, samples_to_optimize);
      if (rows < max_rows) { trace[3 * rows] = i; trace[3 * rows + 1] = x; trace[3 * rows + 2] = cost; rows++; }
      if (cost < best_cost) { best_x = x; best_cost = cost; }
      x += xinc;
    }
    prof.coefs[i].vdef = best_x;
  }
  for (size_t i = 0; i < prof.coefs.size(); i++) best[i] = prof.coefs[i].vdef;
  return rows;
}

// Per-stage trace of StereoHigh (profile.cpp:113-186) / StereoNormal (:46-111).
// p_out[t][ch][k] (k<nst) = stage predictions, v_out[t][ch][k] = stage inputs after CalcError.
// nst = 5 (high) or 4 (normal). Steps exactly as Predict(int32..)+Update() do.
void ref_cascade_trace(int profile, const double *coefs, int ncoef, const int32_t *s0,
                       const int32_t *s1, int n, double *p_out, double *v_out, int32_t *e0,
                       int32_t *e1) {
  CoutMute mute;
  FrameCoder fc(2, 16, profile);
  SacProfile prof = fc.baseprofile;
  set_coefs(prof, coefs, ncoef);
  if (profile == 2) {
    StereoHigh sp(prof, fc.framestats);
    for (int t = 0; t < n; t++) {
      int32_t q0 = (int32_t)round(sp.Predict(0, 1));
      for (int k = 0; k < 5; k++) p_out[(t * 2 + 0) * 5 + k] = sp.p[k];
      sp.CalcError(0, s0[t]);
      int32_t q1 = (int32_t)round(sp.Predict(1, 0));
      for (int k = 0; k < 5; k++) p_out[(t * 2 + 1) * 5 + k] = sp.p[k];
      sp.CalcError(1, s1[t]);
      for (int ch = 0; ch < 2; ch++)
        for (int k = 0; k < 5; k++) v_out[(t * 2 + ch) * 5 + k] = sp.val[ch][k];
      if (e0) e0[t] = s0[t] - q0;
      if (e1) e1[t] = s1[t] - q1;
      sp.Update();
    }
  } else if (profile == 1) {
    StereoNormal sp(prof, fc.framestats);
    for (int t = 0; t < n; t++) {
      int32_t q0 = (int32_t)round(sp.Predict(0, 1));
      for (int k = 0; k < 4; k++) p_out[(t * 2 + 0) * 4 + k] = sp.p[k];
      sp.CalcError(0, s0[t]);
      int32_t q1 = (int32_t)round(sp.Predict(1, 0));
      for (int k = 0; k < 4; k++) p_out[(t * 2 + 1) * 4 + k] = sp.p[k];
      sp.CalcError(1, s1[t]);
      for (int ch = 0; ch < 2; ch++)
        for (int k = 0; k < 4; k++) v_out[(t * 2 + ch) * 4 + k] = sp.val[ch][k];
      if (e0) e0[t] = s0[t] - q0;
      if (e1) e1[t] = s1[t] - q1;
      sp.Update();
    }
  }
}

// One stereo pair of LMSADA2(S,D,mu1,mu2) stages driven as the cascades drive them:
//   p0 = lms[0].Predict(x1[t-1]) ; p1 = lms[1].Predict(x0[t]) ; lms[c].Update(xc[t]).
void ref_lms_pair(int S, int D, double mu1, double mu2, const double *x0, const double *x1, int n,
                  double *p0, double *p1) {
  std::vector<LMSADA2> lms(2, LMSADA2(S, D, mu1, mu2));
  double prev1 = 0.;
  for (int t = 0; t < n; t++) {
    p0[t] = lms[0].Predict(prev1);
    p1[t] = lms[1].Predict(x0[t]);
    lms[0].Update(x0[t]);
    lms[1].Update(x1[t]);
    prev1 = x1[t];
  }
}

// One stereo pair of LPC3(alpha,S,D,8,500) stages (lpc.h:181: S, D are constructor arguments) driven as StereoHigh/Normal drive them.
void ref_lpc3_pair(double alpha, int S, int D, const int32_t *s0, const int32_t *s1, int n, double *p0, double *p1) {
  CoutMute mute;
  std::vector<LPC3> lpc(2, LPC3(alpha, S, D, 8, 500));
  double prev1 = 0.;
  for (int t = 0; t < n; t++) {
    p0[t] = lpc[0].Predict(prev1);
    p1[t] = lpc[1].Predict((double)s0[t]);
    lpc[0].Update((double)s0[t]);
    lpc[1].Update((double)s1[t]);
    prev1 = (double)s1[t];
  }
}

// RLS(n,alpha) mixer: p[t] = Predict(pv[t][0..n)); Update(val[t]).
void ref_rls_run(int n, double alpha, const double *pv, const double *val, int len, double *p) {
  RLS r(n, alpha);
  std::vector<double> v(n);
  for (int t = 0; t < len; t++) {
    for (int i = 0; i < n; i++) v[i] = pv[t * n + i];
    p[t] = r.Predict(v);
    r.Update(val[t]);
  }
}

// Remap::Map (coder/map.cpp:141-152) after Analyse over src[0..n).
void ref_remap_map(const int32_t *src, int n, const int32_t *pred, const int32_t *err, int m, int32_t *out) {
  Remap r;
  r.Reset();
  std::vector<int32_t> tmp(src, src + n);
  r.Analyse(tmp.data(), n);
  for (int i = 0; i < m; i++) out[i] = r.Map(pred[i], err[i]);
}

}  // extern "C"

namespace {
// Frame record exactly as FrameCoder::WriteEncoded lays it out (codec.cpp:480-503): u32 numsamples, then per
// channel u32 blocksize, u16 flag (bit 9 = mapped, low bits = maxbpn), payload. Little-endian.
int pack_frame(FrameCoder &fc, uint8_t *out, int cap) {
  std::vector<uint8_t> v;
  auto p32 = [&](uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((x >> (8 * i)) & 0xff); };
  auto p16 = [&](uint32_t x) { for (int i = 0; i < 2; i++) v.push_back((x >> (8 * i)) & 0xff); };
  p32(fc.numsamples);
  for (int ch = 0; ch < 2; ch++) {
    const uint32_t blocksize = fc.encoded[ch].GetBufPos();
    p32(blocksize);
    uint16_t flag = 0;
    if (fc.framestats[ch].enc_mapped) { flag |= (1 << 9); flag |= fc.framestats[ch].maxbpn_map; }
    else flag |= fc.framestats[ch].maxbpn;
    p16(flag);
    v.insert(v.end(), fc.encoded[ch].GetBuf().begin(), fc.encoded[ch].GetBuf().begin() + blocksize);
  }
  if ((int)v.size() > cap) return -(int)v.size();
  memcpy(out, v.data(), v.size());
  return (int)v.size();
}
}  // namespace

extern "C" {
// The reference's own frame: FrameCoder::Predict() + Encode() (codec.cpp:442-472) -> frame record bytes.
int ref_encode_frame(int profile, const int32_t *s0, const int32_t *s1, int total, uint8_t *out, int cap) {
  CoutMute mute;
  FrameCoder fc(2, total, profile);
  for (int i = 0; i < total; i++) { fc.samples[0][i] = s0[i]; fc.samples[1][i] = s1[i]; }
  fc.SetNumSamples(total);
  fc.Predict();
  fc.Encode();
  return pack_frame(fc, out, cap);
}
// The reference's bit-plane coder fed with residuals computed elsewhere (the HIP path): what a host that
// calls libsacopt for FrameCoder::Predict and keeps Encode()/WriteEncoded() produces.
int ref_encode_residuals(int profile, const int32_t *s0, const int32_t *s1, int total, const int32_t *u0,
                         const int32_t *u1, const int32_t *m0, const int32_t *m1, const int *maxbpn,
                         uint8_t *out, int cap) {
  CoutMute mute;
  FrameCoder fc(2, total, profile);
  load_frame(fc, s0, s1, total);  // samples + Remap tables for MapEncoder (codec.cpp:448-452, :258)
  for (int i = 0; i < total; i++) {
    fc.s2u_error[0][i] = u0[i]; fc.s2u_error[1][i] = u1[i];
    fc.s2u_error_map[0][i] = m0[i]; fc.s2u_error_map[1][i] = m1[i];
  }
  fc.framestats[0].maxbpn = maxbpn[0]; fc.framestats[1].maxbpn = maxbpn[1];
  fc.framestats[0].maxbpn_map = maxbpn[2]; fc.framestats[1].maxbpn_map = maxbpn[3];
  fc.Encode();
  return pack_frame(fc, out, cap);
}

// Encode() alone, channel by channel: the BitplaneCoder payloads a host-side container writer (sac_amd/host/sacfile.h)
// packs. out = payload0 | payload1; info = {len0, enc_mapped0, flagbits0, len1, enc_mapped1, flagbits1} with flagbits =
// maxbpn_map when mapped, else maxbpn (codec.cpp:492-497).
int ref_encode_channels(int profile, const int32_t *s0, const int32_t *s1, int total, const int32_t *u0, const int32_t *u1,
                        const int32_t *m0, const int32_t *m1, const int *maxbpn, uint8_t *out, int cap, int *info) {
  CoutMute mute;
  FrameCoder fc(2, total, profile);
  load_frame(fc, s0, s1, total);
  for (int i = 0; i < total; i++) {
    fc.s2u_error[0][i] = u0[i]; fc.s2u_error[1][i] = u1[i];
    fc.s2u_error_map[0][i] = m0[i]; fc.s2u_error_map[1][i] = m1[i];
  }
  fc.framestats[0].maxbpn = maxbpn[0]; fc.framestats[1].maxbpn = maxbpn[1];
  fc.framestats[0].maxbpn_map = maxbpn[2]; fc.framestats[1].maxbpn_map = maxbpn[3];
  fc.Encode();
  int pos = 0;
  for (int ch = 0; ch < 2; ch++) {
    const int len = (int)fc.encoded[ch].GetBufPos();
    if (pos + len > cap) return -1;
    memcpy(out + pos, fc.encoded[ch].GetBuf().data(), len);
    pos += len;
    info[3 * ch] = len;
    info[3 * ch + 1] = fc.framestats[ch].enc_mapped ? 1 : 0;
    info[3 * ch + 2] = fc.framestats[ch].enc_mapped ? fc.framestats[ch].maxbpn_map : fc.framestats[ch].maxbpn;
  }
  return pos;
}
// Decode() alone (codec.cpp:282-300, :474-478): payloads as ReadEncoded leaves them -> error[ch][0..n) and, for a mapped
// channel, the Remap tables the MapEncoder decoded (usedl[0..32768] then usedh[0..32768], one byte each).
void ref_decode_channels(int profile, int n, const uint8_t *p0, int len0, int mapped0, int bpn0, const uint8_t *p1, int len1,
                         int mapped1, int bpn1, int32_t *e0, int32_t *e1, unsigned char *used0, unsigned char *used1) {
  CoutMute mute;
  FrameCoder fc(2, n, profile);
  fc.SetNumSamples(n);
  const uint8_t *p[2] = {p0, p1};
  const int len[2] = {len0, len1}, mapped[2] = {mapped0, mapped1}, bpn[2] = {bpn0, bpn1};
  unsigned char *used[2] = {used0, used1};
  for (int ch = 0; ch < 2; ch++) {
    fc.framestats[ch].enc_mapped = mapped[ch] != 0;
    fc.framestats[ch].maxbpn = bpn[ch];  // ReadEncoded: maxbpn = flag & 0xff for both kinds (codec.cpp:519)
    std::vector<uint8_t> &b = fc.encoded[ch].GetBuf();
    b.assign(p[ch], p[ch] + len[ch]);
    b.resize(len[ch] + 64, 0);
  }
  fc.Decode();
  for (int i = 0; i < n; i++) { e0[i] = fc.error[0][i]; e1[i] = fc.error[1][i]; }
  for (int ch = 0; ch < 2; ch++)
    if (mapped[ch] && used[ch]) {
      const int sc = fc.framestats[ch].mymap.scale;
      for (int i = 0; i <= sc; i++) { used[ch][i] = fc.framestats[ch].mymap.usedl[i]; used[ch][sc + 1 + i] = fc.framestats[ch].mymap.usedh[i]; }
    }
}

// Timing aid for bench.py's cpu_baseline leg: FrameCoder::Encode() alone (codec.cpp:469-472: per channel both
// BitplaneCoder variants, smaller kept) on residuals computed elsewhere. Returns the two payload sizes summed.
int ref_encode_only(int profile, const int32_t *s0, const int32_t *s1, int total, const int32_t *u0, const int32_t *u1,
                    const int32_t *m0, const int32_t *m1, const int *maxbpn) {
  CoutMute mute;
  FrameCoder fc(2, total, profile);
  load_frame(fc, s0, s1, total);
  for (int i = 0; i < total; i++) {
    fc.s2u_error[0][i] = u0[i]; fc.s2u_error[1][i] = u1[i];
    fc.s2u_error_map[0][i] = m0[i]; fc.s2u_error_map[1][i] = m1[i];
  }
  fc.framestats[0].maxbpn = maxbpn[0]; fc.framestats[1].maxbpn = maxbpn[1];
  fc.framestats[0].maxbpn_map = maxbpn[2]; fc.framestats[1].maxbpn_map = maxbpn[3];
  fc.Encode();
  return (int)(fc.encoded[0].GetBufPos() + fc.encoded[1].GetBufPos());
}
}
