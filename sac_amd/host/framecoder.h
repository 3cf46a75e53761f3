// Host-side mirror of the reference's FrameCoder members that lie on the optimizer/predictor
// path (src/codec.h:18-50), implemented on top of the C ABI (include/sacopt.h). Same names,
// argument meaning and side effects as the reference:
//   SetNumSamples / GetNumSamples          src/codec.h:21-22
//   Predict()                              src/codec.cpp:442-461  (with :455-456 un-commented when optimize=true)
//   PredictStereoFrame(profile,0,1,from,n) src/codec.cpp:154-214
//   GetCost(profile,coef,testval,start,n)  src/codec.cpp:341-352  (CostGolomb state carried, :364)
//   Optimize(profile)                      src/codec.cpp:354-408
//   Unpredict() / UnpredictStereoFrame(profile,0,1,n)  src/codec.cpp:463-478, :216-235 (decoder side; UnpredictBatch
//                                          for a batch of frames). framestats[ch].enc_mapped / mymap / minval / maxval
//                                          (src/codec.h:49) appear as enc_mapped[ch] / used[ch] / minval[ch] / maxval[ch].
// Buffers are the reference's public members (samples / error / s2u_error / s2u_error_map, src/codec.h:29-31);
// maxbpn / maxbpn_map stand for framestats[ch].maxbpn{,_map} (src/codec.h:49, src/codec.cpp:207-211).
// PredictBatch is the batched form a throughput-minded Codec::EncodeFile would call. With apply = true the
// frame is predicted with the parameters Optimize selected (the reference codes with baseprofile, :459) --
// such frames go into a stream whose header carries SAC_PROFILE_PARAMS (sacfile.h): ToRecord / FromRecord move the
// parameters between a FrameCoder and a FrameRecord.
// Header-only; link with libsacopt.so. A FrameCoder is not thread-safe (like the reference); different contexts may be driven
// from different threads (sharder.h).
#ifndef SACOPT_FRAMECODER_H
#define SACOPT_FRAMECODER_H
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sacopt.h"
#include "sacfile.h"

namespace sacopt {

struct SacProfile {  // src/profile.h:10-47
  struct coef { double vmin, vmax, vdef; };
  int type = 0;
  std::vector<coef> coefs;
  double Get(int num) const { return (num >= 0 && num < (int)coefs.size()) ? coefs[num].vdef : 0.; }
};

class FrameCoder {
 public:
  FrameCoder(sacopt_ctx *ctx, int numch, int framesz, int profile) : ctx_(ctx), numchannels(numch), framesize(framesz) {
    if (numch != 2) throw std::invalid_argument("only stereo frames (src/codec.cpp:457)");
    double tab[3 * SACOPT_MAX_COEFS];
    const int n = sacopt_base_profile(profile, tab);  // src/codec.cpp:8-30
    if (!n) throw std::invalid_argument("unknown profile");
    baseprofile.type = profile;
    baseprofile.coefs.resize(n);
    for (int i = 0; i < n; i++) baseprofile.coefs[i] = {tab[3 * i], tab[3 * i + 1], tab[3 * i + 2]};
    optprofile = baseprofile;
    for (auto *v : {&samples, &error, &s2u_error, &s2u_error_map}) { v->resize(2); for (auto &c : *v) c.resize(framesz); }
    maxbpn[0] = maxbpn[1] = maxbpn_map[0] = maxbpn_map[1] = 0;
  }
  void SetNumSamples(int nsamples) { numsamples = nsamples; }
  int GetNumSamples() const { return numsamples; }

  // FrameCoder::Predict with the optimizer call of codec.cpp:455-456 enabled on request.
  // One library call: the optimizer rounds and the full-frame pass overlap on the GPU.
  void Predict(bool optimize = false) {
    FrameCoder *self = this;
    PredictBatch(&self, 1, optimize);
  }

  // Predict() for a batch of frames (same profile type) in one call; frames are independent (codec.cpp:156-159).
  static void PredictBatch(FrameCoder *const *fc, int nframes, bool optimize, bool apply = false) {
    if (nframes > 0) PredictBatchOn(fc[0]->ctx_, fc, nframes, optimize, apply);
  }
  // ... on a given context (the multi-GPU sharder of sharder.h hands each context its own range of frames)
  static void PredictBatchOn(sacopt_ctx *ctx, FrameCoder *const *fc, int nframes, bool optimize, bool apply = false) {
    if (nframes <= 0) return;
    if (apply) { PredictBatchApply(ctx, fc, nframes); return; }
    FrameCoder &f0 = *fc[0];
    const int ncoef = (int)f0.baseprofile.coefs.size();
    std::vector<sacopt_frame> fr(nframes);
    std::vector<sacopt_residuals> out(nframes);
    std::vector<double> prof(3 * ncoef), best((size_t)nframes * ncoef);
    for (int i = 0; i < ncoef; i++) { prof[3 * i] = f0.baseprofile.coefs[i].vmin; prof[3 * i + 1] = f0.baseprofile.coefs[i].vmax; prof[3 * i + 2] = f0.baseprofile.coefs[i].vdef; }
    for (int b = 0; b < nframes; b++) {
      FrameCoder &f = *fc[b];
      if (f.baseprofile.type != f0.baseprofile.type) throw std::invalid_argument("mixed profiles in one batch");
      fr[b] = {f.samples[0].data(), f.samples[1].data(), f.numsamples};
      out[b] = sacopt_residuals();
      out[b].err0 = f.error[0].data(); out[b].err1 = f.error[1].data();
      out[b].s2u0 = f.s2u_error[0].data(); out[b].s2u1 = f.s2u_error[1].data();
      out[b].map0 = f.s2u_error_map[0].data(); out[b].map1 = f.s2u_error_map[1].data();
    }
    check_on(ctx, sacopt_encode_frames(ctx, f0.baseprofile.type, nframes, fr.data(), prof.data(), optimize ? ncoef : 0, best.data(), out.data()));
    for (int b = 0; b < nframes; b++) {
      FrameCoder &f = *fc[b];
      f.maxbpn[0] = out[b].maxbpn[0]; f.maxbpn[1] = out[b].maxbpn[1];
      f.maxbpn_map[0] = out[b].maxbpn[2]; f.maxbpn_map[1] = out[b].maxbpn[3];
      if (optimize) {
        f.optprofile = f.baseprofile;
        for (int i = 0; i < ncoef; i++) f.optprofile.coefs[i].vdef = best[(size_t)b * ncoef + i];
      }
    }
  }

  // Optimize, then PredictStereoFrame(optprofile, 0, 1, 0, numsamples): one library call (the parts of the full-frame pass start as
  // soon as the optimizer has fixed their parameters)
  static void PredictBatchApply(sacopt_ctx *ctx, FrameCoder *const *fc, int nframes) {
    FrameCoder &f0 = *fc[0];
    const int ncoef = (int)f0.baseprofile.coefs.size();
    std::vector<sacopt_frame> fr(nframes);
    std::vector<sacopt_residuals> out(nframes);
    std::vector<double> prof(3 * ncoef), best((size_t)nframes * ncoef);
    for (int i = 0; i < ncoef; i++) { prof[3 * i] = f0.baseprofile.coefs[i].vmin; prof[3 * i + 1] = f0.baseprofile.coefs[i].vmax; prof[3 * i + 2] = f0.baseprofile.coefs[i].vdef; }
    for (int b = 0; b < nframes; b++) {
      FrameCoder &f = *fc[b];
      if (f.baseprofile.type != f0.baseprofile.type) throw std::invalid_argument("mixed profiles in one batch");
      fr[b] = {f.samples[0].data(), f.samples[1].data(), f.numsamples};
      out[b] = sacopt_residuals();
      out[b].err0 = f.error[0].data(); out[b].err1 = f.error[1].data();
      out[b].s2u0 = f.s2u_error[0].data(); out[b].s2u1 = f.s2u_error[1].data();
      out[b].map0 = f.s2u_error_map[0].data(); out[b].map1 = f.s2u_error_map[1].data();
    }
    check_on(ctx, sacopt_encode_frames_selected(ctx, f0.baseprofile.type, nframes, fr.data(), prof.data(), ncoef, best.data(), out.data()));
    for (int b = 0; b < nframes; b++) {
      FrameCoder &f = *fc[b];
      f.maxbpn[0] = out[b].maxbpn[0]; f.maxbpn[1] = out[b].maxbpn[1];
      f.maxbpn_map[0] = out[b].maxbpn[2]; f.maxbpn_map[1] = out[b].maxbpn[3];
      f.optprofile = f.baseprofile;
      for (int i = 0; i < ncoef; i++) f.optprofile.coefs[i].vdef = best[(size_t)b * ncoef + i];
    }
  }
  // FrameCoder::Encode (codec.cpp:469-472) of a batch ON THE DEVICE: both payload variants of both channels of every frame in one
  // sacopt_encode_payloads call (EncodeMonoFrame_Normal / _Mapped, codec.cpp:237-265), the mapped one kept unless the plain one is
  // strictly smaller (codec.cpp:268-280). Fills recs[b].ch[c] (payload, enc_mapped, maxbpn) and enc_mapped[c]; needs the
  // residuals of PredictBatch*. The host's own coder is the alternative: this pays for large batches.
  static void EncodeBatchOn(sacopt_ctx *ctx, FrameCoder *const *fc, int nframes, FrameRecord *recs) {
    if (nframes <= 0) return;
    const int nbuf = 4 * nframes;
    std::vector<const int32_t *> src(nbuf);
    std::vector<int32_t> n(nbuf), mb(nbuf), cap(nbuf), sizes(nbuf);
    std::vector<const uint8_t *> used(nbuf, nullptr);
    std::vector<std::vector<uint8_t>> tabs((size_t)2 * nframes), outs(nbuf);
    std::vector<uint8_t *> out(nbuf);
    for (int b = 0; b < nframes; b++) {
      FrameCoder &f = *fc[b];
      for (int ch = 0; ch < 2; ch++) {
        const int k = 4 * b + 2 * ch;
        tabs[(size_t)2 * b + ch] = AnalyseUsed(f.samples[ch].data(), f.numsamples);
        src[k] = f.s2u_error[ch].data(); mb[k] = f.maxbpn[ch];
        src[k + 1] = f.s2u_error_map[ch].data(); mb[k + 1] = f.maxbpn_map[ch]; used[k + 1] = tabs[(size_t)2 * b + ch].data();
        for (int v = 0; v < 2; v++) {
          n[k + v] = f.numsamples;
          cap[k + v] = 4 * f.numsamples + 8192 + 64;
          outs[k + v].resize((size_t)cap[k + v]);
          out[k + v] = outs[k + v].data();
        }
      }
    }
    check_on(ctx, sacopt_encode_payloads(ctx, nbuf, src.data(), n.data(), mb.data(), used.data(), out.data(), cap.data(), sizes.data()));
    for (int k = 0; k < nbuf; k++)  // incompressible data can need more than 4 bytes per sample: that buffer again, with the size the call reported
      if (sizes[k] > cap[k]) {
        cap[k] = sizes[k];
        outs[k].resize((size_t)cap[k]);
        out[k] = outs[k].data();
        check_on(ctx, sacopt_encode_payloads(ctx, 1, &src[k], &n[k], &mb[k], &used[k], &out[k], &cap[k], &sizes[k]));
      }
    for (int b = 0; b < nframes; b++) {
      FrameCoder &f = *fc[b];
      recs[b].numsamples = (uint32_t)f.numsamples;
      for (int ch = 0; ch < 2; ch++) {
        const int k = 4 * b + 2 * ch;
        const bool mapped = !(sizes[k] < sizes[k + 1]);
        const int pick = k + (mapped ? 1 : 0);
        f.enc_mapped[ch] = mapped;
        recs[b].ch[ch].enc_mapped = mapped;
        recs[b].ch[ch].maxbpn = mapped ? f.maxbpn_map[ch] : f.maxbpn[ch];
        recs[b].ch[ch].payload.assign(outs[pick].begin(), outs[pick].begin() + sizes[pick]);
      }
    }
  }
  // Per-frame stream fields of the parameter extension (sacfile.h): the frame's optprofile values; minval / maxval are
  // written as the full int32 range, which makes StereoHigh::Unpredict's clamp (profile.cpp:163,170) the identity -- the
  // encoder does not clamp (profile.cpp:152,156), so anything narrower would not round-trip.
  void ToRecord(FrameRecord &rec) const {
    rec.numsamples = (uint32_t)numsamples;
    rec.ncoef = (int)optprofile.coefs.size();
    for (int i = 0; i < rec.ncoef; i++) rec.coefs[i] = optprofile.coefs[i].vdef;
    for (int c = 0; c < 2; c++) { rec.minval[c] = INT32_MIN; rec.maxval[c] = INT32_MAX; }
  }
  void FromRecord(const FrameRecord &rec) {
    if (rec.ncoef != (int)baseprofile.coefs.size()) throw std::invalid_argument("parameter count of the record does not fit the profile");
    numsamples = (int)rec.numsamples;
    optprofile = baseprofile;
    for (int i = 0; i < rec.ncoef; i++) optprofile.coefs[i].vdef = rec.coefs[i];
    for (int c = 0; c < 2; c++) { minval[c] = rec.minval[c]; maxval[c] = rec.maxval[c]; enc_mapped[c] = rec.ch[c].enc_mapped; }
  }

  void PredictStereoFrame(const SacProfile &profile, int ch0, int ch1, int from, int n) {
    if (ch0 != 0 || ch1 != 1) throw std::invalid_argument("channel order");
    sacopt_frame fr = {samples[0].data(), samples[1].data(), numsamples};
    sacopt_residuals out = {};
    out.err0 = error[0].data(); out.err1 = error[1].data();
    out.s2u0 = s2u_error[0].data(); out.s2u1 = s2u_error[1].data();
    out.map0 = s2u_error_map[0].data(); out.map1 = s2u_error_map[1].data();
    std::vector<double> c(profile.coefs.size());
    for (size_t i = 0; i < c.size(); i++) c[i] = profile.coefs[i].vdef;
    const int32_t f = from, nn = n;
    check(sacopt_predict_stereo_frames(ctx_, profile.type, 1, &fr, &f, &nn, c.data(), &out));
    maxbpn[0] = out.maxbpn[0]; maxbpn[1] = out.maxbpn[1];
    maxbpn_map[0] = out.maxbpn[2]; maxbpn_map[1] = out.maxbpn[3];
  }

  double GetCost(SacProfile &profile, int coef, double testval, int start_sample, int samples_to_optimize) {
    sacopt_frame fr = {samples[0].data(), samples[1].data(), numsamples};
    std::vector<double> c(profile.coefs.size());
    for (size_t i = 0; i < c.size(); i++) c[i] = profile.coefs[i].vdef;
    c[coef] = testval;
    const int32_t f = start_sample, nn = samples_to_optimize;
    double cost = 0.;
    check(sacopt_get_costs(ctx_, profile.type, 1, &fr, &f, &nn, 1, c.data(), &cost, &cost_state));
    return cost;
  }
  void ResetCostState() { cost_state = 0.; }  // a fresh CostGolomb instance (codec.cpp:364)

  void Optimize(SacProfile &profile) {
    sacopt_frame fr = {samples[0].data(), samples[1].data(), numsamples};
    std::vector<double> prof(3 * profile.coefs.size()), best(profile.coefs.size());
    for (size_t i = 0; i < profile.coefs.size(); i++) { prof[3 * i] = profile.coefs[i].vmin; prof[3 * i + 1] = profile.coefs[i].vmax; prof[3 * i + 2] = profile.coefs[i].vdef; }
    check(sacopt_optimize_frames(ctx_, profile.type, 1, &fr, prof.data(), (int)profile.coefs.size(), best.data(), nullptr, 0, nullptr));
    for (size_t i = 0; i < best.size(); i++) profile.coefs[i].vdef = best[i];
  }

  // Remap::Analyse (src/coder/map.cpp:111-130) as the byte tables the MapEncoder codes (usedl[0..32768], usedh[0..32768])
  static std::vector<uint8_t> AnalyseUsed(const int32_t *src, int n) {
    std::vector<uint8_t> u(2 * 32769, 0);
    for (int i = 0; i < n; i++) {
      const int64_t v = src[i];
      if (v > 0 && v <= 32768) u[32769 + v] = 1;
      else if (v < 0 && -v <= 32768) u[-v] = 1;
    }
    return u;
  }

  // FrameCoder::Unpredict (codec.cpp:463-478): error[ch] (mapped where enc_mapped[ch]) -> samples[ch], base profile
  void Unpredict() {
    FrameCoder *self = this;
    UnpredictBatch(&self, 1);
  }
  void UnpredictStereoFrame(const SacProfile &profile, int ch0, int ch1, int n) {
    if (ch0 != 0 || ch1 != 1) throw std::invalid_argument("channel order");
    sacopt_dec_frame d = dec_frame(n);
    std::vector<double> c(profile.coefs.size());
    for (size_t i = 0; i < c.size(); i++) c[i] = profile.coefs[i].vdef;
    check(sacopt_unpredict_stereo_frames(ctx_, profile.type, 1, &d, c.data()));
  }
  // with_optprofile: decode with each frame's optprofile (streams that carry optimised parameters)
  static void UnpredictBatch(FrameCoder *const *fc, int nframes, bool with_optprofile = false) {
    if (nframes > 0) UnpredictBatchOn(fc[0]->ctx_, fc, nframes, with_optprofile);
  }
  static void UnpredictBatchOn(sacopt_ctx *ctx, FrameCoder *const *fc, int nframes, bool with_optprofile = false) {
    if (nframes <= 0) return;
    FrameCoder &f0 = *fc[0];
    const int ncoef = (int)f0.baseprofile.coefs.size();
    std::vector<sacopt_dec_frame> d(nframes);
    std::vector<double> c((size_t)nframes * ncoef);
    for (int b = 0; b < nframes; b++) {
      FrameCoder &f = *fc[b];
      if (f.baseprofile.type != f0.baseprofile.type) throw std::invalid_argument("mixed profiles in one batch");
      d[b] = f.dec_frame(f.numsamples);
      const SacProfile &p = with_optprofile ? f.optprofile : f.baseprofile;
      for (int i = 0; i < ncoef; i++) c[(size_t)b * ncoef + i] = p.coefs[i].vdef;
    }
    check_on(ctx, sacopt_unpredict_stereo_frames(ctx, f0.baseprofile.type, nframes, d.data(), c.data()));
  }

  std::vector<std::vector<int32_t>> samples, error, s2u_error, s2u_error_map;
  int maxbpn[2], maxbpn_map[2];
  bool enc_mapped[2] = {false, false};
  std::vector<uint8_t> used[2];
  int32_t minval[2] = {INT32_MIN, INT32_MIN}, maxval[2] = {INT32_MAX, INT32_MAX};
  SacProfile baseprofile, optprofile;

 private:
  sacopt_dec_frame dec_frame(int n) {
    sacopt_dec_frame d = {};
    d.e0 = error[0].data(); d.e1 = error[1].data(); d.v0 = samples[0].data(); d.v1 = samples[1].data(); d.numsamples = n;
    d.used0 = enc_mapped[0] ? used[0].data() : nullptr;
    d.used1 = enc_mapped[1] ? used[1].data() : nullptr;
    for (int k = 0; k < 2; k++) {
      if (enc_mapped[k] && used[k].size() != 2 * 32769) throw std::invalid_argument("enc_mapped without Remap tables");
      d.minval[k] = minval[k]; d.maxval[k] = maxval[k];
    }
    return d;
  }
  void check(int rc) { check_on(ctx_, rc); }
  static void check_on(sacopt_ctx *ctx, int rc) { if (rc) throw std::runtime_error(std::string("sacopt: ") + sacopt_last_error(ctx)); }
  sacopt_ctx *ctx_;
  int numchannels, framesize, numsamples = 0;
  double cost_state = 0.;
};

}  // namespace sacopt
#endif
