// TEST INFRASTRUCTURE ONLY (it links the reference's objects, which exist in the build container only).
//
// The drop-in demonstration of INTEGRATION.md as a program: the reference's OWN sac binary -- its main(), WAV reader, .sac
// writer, bit-plane / range coder, all compiled unmodified from /root/reference/src -- with exactly two member functions
// replaced by calls into libsacopt through the C ABI (include/sacopt.h):
//   FrameCoder::Predict()    src/codec.cpp:442-461   -> sacopt_encode_frames   (with the Optimize call of :455-456 enabled
//                                                       when SAC_GPU_OPTIMIZE is set; the stream stays the base profile's)
//   FrameCoder::Unpredict()  src/codec.cpp:463-467   -> sacopt_unpredict_stereo_frames
// oracle/Makefile weakens the reference's definitions of the two in a COPY of its codec.o (objcopy --weaken-symbol), so that
// these strong definitions win at link time; nothing in the reference's sources is edited.
// `oracle/_ref/sac_gpu --encode --normal in.wav out.sac` must write the file `oracle/_ref/sac` writes
// (tests/test_gpu_stream.py::test_reference_program_with_gpu_predictor).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#define private public
#define protected public
#include "codec.h"
#undef private
#undef protected

#include "../include/sacopt.h"

static sacopt_ctx *gpu() {
  static sacopt_ctx *ctx = nullptr;
  if (!ctx && sacopt_create(0, nullptr, 0, &ctx)) {  // NULL table: this host's rsqrtss, what the reference itself would execute
    std::cerr << "sacopt_create: " << sacopt_last_error(nullptr) << std::endl;
    exit(2);
  }
  return ctx;
}

void FrameCoder::Predict() {
  for (int ch = 0; ch < numchannels; ch++) {  // the MapEncoder needs the host tables (codec.cpp:448-452, :258)
    framestats[ch].mymap.Reset();
    framestats[ch].mymap.Analyse(&(samples[ch][0]), numsamples);
  }
  if (numchannels != 2) return;
  sacopt_frame fr = {&samples[0][0], &samples[1][0], numsamples};
  sacopt_residuals out = {};
  out.err0 = &error[0][0]; out.err1 = &error[1][0];
  out.s2u0 = &s2u_error[0][0]; out.s2u1 = &s2u_error[1][0];
  out.map0 = &s2u_error_map[0][0]; out.map1 = &s2u_error_map[1][0];
  std::vector<double> prof, best(baseprofile.coefs.size());
  for (auto &k : baseprofile.coefs) { prof.push_back(k.vmin); prof.push_back(k.vmax); prof.push_back(k.vdef); }
  const bool optimize = getenv("SAC_GPU_OPTIMIZE") != nullptr && baseprofile.coefs.size() >= 1;
  if (sacopt_encode_frames(gpu(), baseprofile.type, 1, &fr, prof.data(), optimize ? (int)baseprofile.coefs.size() : 0, best.data(), &out)) {
    std::cerr << "sacopt_encode_frames: " << sacopt_last_error(gpu()) << std::endl;
    exit(2);
  }
  if (optimize) for (double b : best) std::cout << b << std::endl;  // what Optimize prints (codec.cpp:405)
  framestats[0].maxbpn = out.maxbpn[0]; framestats[1].maxbpn = out.maxbpn[1];          // codec.cpp:207-208
  framestats[0].maxbpn_map = out.maxbpn[2]; framestats[1].maxbpn_map = out.maxbpn[3];  // codec.cpp:210-211
}

void FrameCoder::Unpredict() {
  if (numchannels != 2) return;
  sacopt_dec_frame d = {};
  d.e0 = &error[0][0]; d.e1 = &error[1][0]; d.v0 = &samples[0][0]; d.v1 = &samples[1][0]; d.numsamples = numsamples;
  std::vector<uint8_t> u[2];
  for (int k = 0; k < 2; k++) {
    if (framestats[k].enc_mapped) {  // usedl[0..32768] then usedh[0..32768]
      Remap &m = framestats[k].mymap;
      u[k].assign(m.usedl.begin(), m.usedl.end());
      u[k].insert(u[k].end(), m.usedh.begin(), m.usedh.end());
    }
    d.minval[k] = framestats[k].minval; d.maxval[k] = framestats[k].maxval;  // profile.cpp:163,170 (never set by the reference's decoder)
  }
  d.used0 = u[0].empty() ? nullptr : u[0].data();
  d.used1 = u[1].empty() ? nullptr : u[1].data();
  if (sacopt_unpredict_stereo_frames(gpu(), baseprofile.type, 1, &d, nullptr)) {
    std::cerr << "sacopt_unpredict_stereo_frames: " << sacopt_last_error(gpu()) << std::endl;
    exit(2);
  }
}
