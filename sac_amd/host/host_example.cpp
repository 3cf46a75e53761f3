// Host-side example / test driver: encodes synthetic frames through sacopt::FrameCoder (the C++ mirror of the
// reference's FrameCoder on this path) and prints, per frame, the selected parameters and checksums of the
// residual arrays. tests/test_gpu_parity.py::test_cpp_host_framecoder compares the output with the ctypes path.
//   usage: host_example <profile> <nframes> <numsamples> <optimize 0|1>
#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "framecoder.h"

// integer-deterministic stereo test signal (same generator family as sac_amd/synth.py is NOT needed here:
// the test reads the samples back through the checksum of the inputs as well)
static void fill(std::vector<int32_t> &l, std::vector<int32_t> &r, int n, uint32_t seed) {
  uint32_t s = seed * 2654435761u + 12345u;
  int32_t a = 0, b = 0;
  for (int i = 0; i < n; i++) {
    s = s * 1664525u + 1013904223u;
    const int32_t noise = (int32_t)((s >> 16) & 0x3ff) - 512;
    a += ((i / 64) & 1) ? 37 : -37;   // slow triangle
    b = (a * 3) / 4 + noise / 4;
    l[i] = a * 8 + noise;
    r[i] = b * 8 - noise / 2;
  }
}
static uint64_t fnv(const int32_t *p, int n) {
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < n; i++) { h ^= (uint32_t)p[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s profile nframes numsamples optimize\n", argv[0]); return 2; }
  const int profile = atoi(argv[1]), nframes = atoi(argv[2]), n = atoi(argv[3]), optimize = atoi(argv[4]);
  sacopt_ctx *ctx = nullptr;
  if (sacopt_create(0, nullptr, 0, &ctx)) { fprintf(stderr, "sacopt_create: %s\n", sacopt_last_error(nullptr)); return 1; }
  try {
    std::vector<std::unique_ptr<sacopt::FrameCoder>> fc;
    std::vector<sacopt::FrameCoder *> ptr;
    for (int b = 0; b < nframes; b++) {
      fc.emplace_back(new sacopt::FrameCoder(ctx, 2, n, profile));
      fill(fc[b]->samples[0], fc[b]->samples[1], n, 100 + b);
      fc[b]->SetNumSamples(n);
      ptr.push_back(fc[b].get());
    }
    sacopt::FrameCoder::PredictBatch(ptr.data(), nframes, optimize != 0);
    // decoder round trip (FrameCoder::Unpredict, codec.cpp:463-478): channel 0 from its plain residuals, channel 1 as a
    // mapped channel (U2S of s2u_error_map, utils.h:62-66, with the Remap tables of the frame); samples are overwritten
    std::vector<std::vector<int32_t>> keep0(nframes), keep1(nframes), keepe1(nframes);
    for (int b = 0; b < nframes; b++) {
      sacopt::FrameCoder &f = *fc[b];
      keep0[b] = f.samples[0]; keep1[b] = f.samples[1]; keepe1[b] = f.error[1];
      f.enc_mapped[1] = true;
      f.used[1] = sacopt::FrameCoder::AnalyseUsed(f.samples[1].data(), n);
      for (int i = 0; i < n; i++) { const int32_t u = f.s2u_error_map[1][i]; f.error[1][i] = (u & 1) ? ((u + 1) >> 1) : -(u >> 1); }
      std::fill(f.samples[0].begin(), f.samples[0].end(), 0);
      std::fill(f.samples[1].begin(), f.samples[1].end(), 0);
    }
    sacopt::FrameCoder::UnpredictBatch(ptr.data(), nframes);
    std::vector<int> dec_ok(nframes);
    for (int b = 0; b < nframes; b++) {
      sacopt::FrameCoder &f = *fc[b];
      dec_ok[b] = f.samples[0] == keep0[b] && f.samples[1] == keep1[b];
      f.samples[0] = keep0[b]; f.samples[1] = keep1[b]; f.error[1] = keepe1[b];
    }
    for (int b = 0; b < nframes; b++) {
      sacopt::FrameCoder &f = *fc[b];
      printf("frame %d in %016" PRIx64 " %016" PRIx64 " err %016" PRIx64 " %016" PRIx64 " s2u %016" PRIx64 " %016" PRIx64
             " map %016" PRIx64 " %016" PRIx64 " maxbpn %d %d %d %d best",
             b, fnv(f.samples[0].data(), n), fnv(f.samples[1].data(), n), fnv(f.error[0].data(), n), fnv(f.error[1].data(), n),
             fnv(f.s2u_error[0].data(), n), fnv(f.s2u_error[1].data(), n), fnv(f.s2u_error_map[0].data(), n),
             fnv(f.s2u_error_map[1].data(), n), f.maxbpn[0], f.maxbpn[1], f.maxbpn_map[0], f.maxbpn_map[1]);
      for (auto &c : f.optprofile.coefs) printf(" %.17g", c.vdef);
      printf(" decode %s\n", dec_ok[b] ? "ok" : "MISMATCH");
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    sacopt_destroy(ctx);
    return 1;
  }
  sacopt_destroy(ctx);
  return 0;
}
