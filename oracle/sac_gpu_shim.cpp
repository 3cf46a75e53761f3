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
//
// The batched binding of INTEGRATION.md as well: with SAC_GPU_BATCH=B (> 1) a third member is replaced,
//   Codec::EncodeFile()      src/codec.cpp:530-556   -> the same loop over batches of B frames: ONE sacopt_encode_frames call per
//                                                       batch; the frames' Encode() -- the reference's own bit-plane / range
//                                                       coder, both variants per channel (codec.cpp:268-280, 469-472) -- runs on
//                                                       SAC_GPU_CODER_THREADS host threads WHILE the GPU works on the next batch
//                                                       (frames are independent: codec.cpp:156-159, 240-242); WriteEncoded in
//                                                       frame order. The file must still be the reference's, byte for byte.
// It prints one line "sac_gpu_timing {...}" to stderr: wall time of the loop, of the GPU calls and of the coder.
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
#include <memory>
#include <string>
#include <thread>
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

// ---- Codec::EncodeFile (src/codec.cpp:530-556), batched and pipelined when SAC_GPU_BATCH > 1 -----------------------------
static void predict_batch(std::vector<std::unique_ptr<FrameCoder>> &fc, int nb, bool optimize) {
  if (!nb) return;
  std::vector<sacopt_frame> fr(nb);
  std::vector<sacopt_residuals> out(nb);
  const SacProfile &bp = fc[0]->baseprofile;
  std::vector<double> prof, best((size_t)nb * bp.coefs.size());
  for (auto &k : bp.coefs) { prof.push_back(k.vmin); prof.push_back(k.vmax); prof.push_back(k.vdef); }
  for (int b = 0; b < nb; b++) {
    FrameCoder &f = *fc[b];
    for (int ch = 0; ch < f.numchannels; ch++) {  // the MapEncoder needs the host tables (codec.cpp:448-452, :258)
      f.framestats[ch].mymap.Reset();
      f.framestats[ch].mymap.Analyse(&(f.samples[ch][0]), f.numsamples);
    }
    fr[b] = {&f.samples[0][0], &f.samples[1][0], f.numsamples};
    out[b] = sacopt_residuals();
    out[b].err0 = &f.error[0][0]; out[b].err1 = &f.error[1][0];
    out[b].s2u0 = &f.s2u_error[0][0]; out[b].s2u1 = &f.s2u_error[1][0];
    out[b].map0 = &f.s2u_error_map[0][0]; out[b].map1 = &f.s2u_error_map[1][0];
  }
  if (sacopt_encode_frames(gpu(), bp.type, nb, fr.data(), prof.data(), optimize ? (int)bp.coefs.size() : 0, best.data(), out.data())) {
    std::cerr << "sacopt_encode_frames: " << sacopt_last_error(gpu()) << std::endl;
    exit(2);
  }
  for (int b = 0; b < nb; b++) {
    FrameCoder &f = *fc[b];
    f.framestats[0].maxbpn = out[b].maxbpn[0]; f.framestats[1].maxbpn = out[b].maxbpn[1];          // codec.cpp:207-208
    f.framestats[0].maxbpn_map = out[b].maxbpn[2]; f.framestats[1].maxbpn_map = out[b].maxbpn[3];  // codec.cpp:210-211
  }
}

// SAC_GPU_DEVICE_CODER=1: FrameCoder::Encode (codec.cpp:469-472) of a batch on the device as well -- both payload variants of
// every channel in ONE sacopt_encode_payloads call, the smaller kept as EncodeMonoFrame does (codec.cpp:268-280). No host coder
// threads then: the host only reads the WAV and writes the records.
static void device_encode_batch(std::vector<std::unique_ptr<FrameCoder>> &fc, int nb) {
  if (!nb) return;
  const int nbuf = 4 * nb;
  std::vector<const int32_t *> src(nbuf);
  std::vector<int32_t> n(nbuf), mb(nbuf), cap(nbuf), sizes(nbuf);
  std::vector<const uint8_t *> used(nbuf, nullptr);
  std::vector<std::vector<uint8_t>> tabs((size_t)2 * nb, std::vector<uint8_t>(2 * 32769)), outs(nbuf);
  std::vector<uint8_t *> out(nbuf);
  for (int b = 0; b < nb; b++) {
    FrameCoder &f = *fc[b];
    for (int ch = 0; ch < 2; ch++) {
      const int k = 4 * b + 2 * ch;
      std::vector<uint8_t> &t = tabs[(size_t)2 * b + ch];
      for (int i = 0; i <= 32768; i++) { t[i] = f.framestats[ch].mymap.usedl[i]; t[32769 + i] = f.framestats[ch].mymap.usedh[i]; }
      src[k] = &f.s2u_error[ch][0]; mb[k] = f.framestats[ch].maxbpn;
      src[k + 1] = &f.s2u_error_map[ch][0]; mb[k + 1] = f.framestats[ch].maxbpn_map; used[k + 1] = t.data();
      for (int v = 0; v < 2; v++) {
        n[k + v] = f.numsamples;
        cap[k + v] = 4 * f.numsamples + 8192 + 64;
        outs[k + v].resize((size_t)cap[k + v]);
        out[k + v] = outs[k + v].data();
      }
    }
  }
  if (sacopt_encode_payloads(gpu(), nbuf, src.data(), n.data(), mb.data(), used.data(), out.data(), cap.data(), sizes.data())) {
    std::cerr << "sacopt_encode_payloads: " << sacopt_last_error(gpu()) << std::endl;
    exit(2);
  }
  for (int k = 0; k < nbuf; k++)  // incompressible data can need more than 4 bytes per sample: that buffer again, with the size the call reported
    if (sizes[k] > cap[k]) {
      cap[k] = sizes[k];
      outs[k].resize((size_t)cap[k]);
      out[k] = outs[k].data();
      if (sacopt_encode_payloads(gpu(), 1, &src[k], &n[k], &mb[k], &used[k], &out[k], &cap[k], &sizes[k])) {
        std::cerr << "sacopt_encode_payloads: " << sacopt_last_error(gpu()) << std::endl;
        exit(2);
      }
    }
  for (int b = 0; b < nb; b++) {
    FrameCoder &f = *fc[b];
    for (int ch = 0; ch < 2; ch++) {
      const int k = 4 * b + 2 * ch;
      const bool mapped = !(sizes[k] < sizes[k + 1]);  // codec.cpp:272-279
      const int pick = k + (mapped ? 1 : 0);
      f.framestats[ch].enc_mapped = mapped;
      f.encoded[ch].Reset();
      for (int i = 0; i < sizes[pick]; i++) f.encoded[ch].PutByte(outs[pick][(size_t)i]);
    }
  }
}

void Codec::EncodeFile(Wav &myWav, Sac &mySac, int profile) {
  const int numchannels = myWav.getNumChannels();
  framesize = 4 * myWav.getSampleRate();
  const int B = std::max(1, getenv("SAC_GPU_BATCH") ? atoi(getenv("SAC_GPU_BATCH")) : 1);
  const int T = std::max(1, getenv("SAC_GPU_CODER_THREADS") ? atoi(getenv("SAC_GPU_CODER_THREADS")) : 16);
  mySac.SetProfile(profile);
  mySac.WriteHeader(myWav);
  myWav.InitFileBuf(framesize);
  int samplescoded = 0, samplestocode = myWav.getNumSamples();
  if (B == 1 || numchannels != 2) {  // the reference's loop as it stands (its Predict is the one above)
    FrameCoder myFrame(numchannels, framesize, profile);
    while (samplestocode > 0) {
      int samplesread = myWav.ReadSamples(myFrame.samples, framesize);
      myFrame.SetNumSamples(samplesread);
      myFrame.Predict();
      myFrame.Encode();
      myFrame.WriteEncoded(mySac);
      samplescoded += samplesread;
      PrintProgress(samplescoded, myWav.getNumSamples());
      samplestocode -= samplesread;
    }
    std::cout << std::endl;
    return;
  }
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  gpu();  // context creation (rsqrtss probe, streams) is not part of the loop's time
  const bool optimize = getenv("SAC_GPU_OPTIMIZE") != nullptr;
  const bool device_coder = getenv("SAC_GPU_DEVICE_CODER") != nullptr;
  std::vector<std::unique_ptr<FrameCoder>> set[2];
  for (auto &s : set) for (int b = 0; b < B; b++) s.emplace_back(new FrameCoder(numchannels, framesize, profile));
  int nb[2] = {0, 0};
  std::vector<std::thread> coders;           // coding the batch in set[cur ^ 1] while the GPU works on set[cur]
  std::vector<double> coder_busy((size_t)T, 0.);
  double gpu_s = 0., coder_crit_s = 0.;
  int batches = 0;
  const auto t_begin = clk::now();
  auto start_coding = [&](int which) {
    coders.clear();
    for (int t = 0; t < T && t < nb[which]; t++)
      coders.emplace_back([&, t, which] {
        const auto a = clk::now();
        for (int b = t; b < nb[which]; b += T) set[which][b]->Encode();  // codec.cpp:469-472, unchanged
        coder_busy[(size_t)t] = secs(a, clk::now());
      });
  };
  auto finish_coding = [&](int which) {  // join, then the records in frame order
    for (auto &th : coders) th.join();
    double mx = 0.;
    for (size_t t = 0; t < coders.size(); t++) mx = std::max(mx, coder_busy[t]);
    coder_crit_s += mx;
    coders.clear();
    for (int b = 0; b < nb[which]; b++) {
      set[which][b]->WriteEncoded(mySac);
      samplescoded += set[which][b]->GetNumSamples();
    }
    PrintProgress(samplescoded, myWav.getNumSamples());
  };
  int cur = 0;
  bool pending = false;  // a batch is being coded in set[cur ^ 1]
  while (samplestocode > 0) {
    nb[cur] = 0;
    while (nb[cur] < B && samplestocode > 0) {
      FrameCoder &f = *set[cur][nb[cur]];
      const int samplesread = myWav.ReadSamples(f.samples, framesize);
      f.SetNumSamples(samplesread);
      samplestocode -= samplesread;
      nb[cur]++;
    }
    const auto a = clk::now();
    predict_batch(set[cur], nb[cur], optimize);  // GPU (blocking); the coder threads of the previous batch run meanwhile
    if (device_coder) {  // ... or no host coder at all: the payloads come from the device too, the records go out at once
      device_encode_batch(set[cur], nb[cur]);
      gpu_s += secs(a, clk::now());
      batches++;
      for (int b = 0; b < nb[cur]; b++) {
        set[cur][b]->WriteEncoded(mySac);
        samplescoded += set[cur][b]->GetNumSamples();
      }
      PrintProgress(samplescoded, myWav.getNumSamples());
      continue;
    }
    gpu_s += secs(a, clk::now());
    batches++;
    if (pending) finish_coding(cur ^ 1);
    start_coding(cur);
    pending = true;
    cur ^= 1;
  }
  if (pending) finish_coding(cur ^ 1);
  std::cout << std::endl;
  const double total = secs(t_begin, clk::now());
  std::cerr << "sac_gpu_timing {\"samples\": " << myWav.getNumSamples() << ", \"batch_frames\": " << B << ", \"batches\": " << batches
            << ", \"coder_threads\": " << (device_coder ? 0 : T) << ", \"device_coder\": " << (device_coder ? 1 : 0) << ", \"optimize\": " << (optimize ? 1 : 0) << ", \"loop_s\": " << total << ", \"gpu_calls_s\": " << gpu_s
            << ", \"coder_critical_path_s\": " << coder_crit_s << "}" << std::endl;
}
