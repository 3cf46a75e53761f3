// Host-side mirror of the reference's FrameCoder members that lie on the optimizer/predictor
// path (src/codec.h:18-50), implemented on top of the C ABI (include/sacopt.h). Same names,
// argument meaning and side effects as the reference:
//   SetNumSamples / GetNumSamples          src/codec.h:21-22
//   Predict()                              src/codec.cpp:442-461  (with :455-456 un-commented when optimize=true)
//   PredictStereoFrame(profile,0,1,from,n) src/codec.cpp:154-214
//   GetCost(profile,coef,testval,start,n)  src/codec.cpp:341-352  (CostGolomb state carried, :364)
//   Optimize(profile)                      src/codec.cpp:354-408
// Buffers are the reference's public members (samples / error / s2u_error, src/codec.h:29-31).
// Header-only; link with libsacopt.so. Not thread-safe (like the reference).
#ifndef SACOPT_FRAMECODER_H
#define SACOPT_FRAMECODER_H
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sacopt.h"

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
    for (auto *v : {&samples, &error, &s2u_error}) { v->resize(2); for (auto &c : *v) c.resize(framesz); }
    maxbpn[0] = maxbpn[1] = 0;
  }
  void SetNumSamples(int nsamples) { numsamples = nsamples; }
  int GetNumSamples() const { return numsamples; }

  // FrameCoder::Predict with the optimizer call of codec.cpp:455-456 enabled on request.
  void Predict(bool optimize = false) {
    if (optimize) { optprofile = baseprofile; Optimize(optprofile); }
    PredictStereoFrame(baseprofile, 0, 1, 0, numsamples);  // "ref+opt": the stream uses baseprofile (codec.cpp:459)
  }

  void PredictStereoFrame(const SacProfile &profile, int ch0, int ch1, int from, int n) {
    if (ch0 != 0 || ch1 != 1) throw std::invalid_argument("channel order");
    sacopt_frame fr = {samples[0].data(), samples[1].data(), numsamples};
    sacopt_residuals out = {};
    out.err0 = error[0].data(); out.err1 = error[1].data();
    out.s2u0 = s2u_error[0].data(); out.s2u1 = s2u_error[1].data();
    std::vector<double> c(profile.coefs.size());
    for (size_t i = 0; i < c.size(); i++) c[i] = profile.coefs[i].vdef;
    const int32_t f = from, nn = n;
    check(sacopt_predict_stereo_frames(ctx_, profile.type, 1, &fr, &f, &nn, c.data(), &out));
    maxbpn[0] = out.maxbpn[0]; maxbpn[1] = out.maxbpn[1];
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

  std::vector<std::vector<int32_t>> samples, error, s2u_error;
  int maxbpn[2];
  SacProfile baseprofile, optprofile;

 private:
  void check(int rc) { if (rc) throw std::runtime_error(std::string("sacopt: ") + sacopt_last_error(ctx_)); }
  sacopt_ctx *ctx_;
  int numchannels, framesize, numsamples = 0;
  double cost_state = 0.;
};

}  // namespace sacopt
#endif
