// Host-side example / test driver: encodes synthetic frames through sacopt::FrameCoder (the C++ mirror of the
// reference's FrameCoder on this path) and prints, per frame, the selected parameters and checksums of the
// residual arrays. tests/test_gpu_parity.py::test_cpp_host_framecoder compares the output with the ctypes path.
//   usage: host_example <profile> <nframes> <numsamples> <optimize 0|1|2> [nctx]   (2: also the "apply" round trip -- frames coded
//          with the parameters Optimize selected travel through the parameter-extension stream records of sacfile.h and decode back;
//          nctx > 1: that many contexts, on the node's GPUs round robin, driven by sharder.h's EncodeFrames / DecodeFrames -- one host
//          thread and one contiguous range of frames per context)
#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "framecoder.h"
#include "sharder.h"

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
  const int nctx = argc > 5 ? std::max(1, atoi(argv[5])) : 1;
  int ndev = 1;
  if (const char *e = getenv("HOST_EXAMPLE_GPUS")) ndev = std::max(1, atoi(e));  // GPUs to spread the contexts over (default: all on GPU 0)
  std::vector<sacopt_ctx *> ctxs((size_t)nctx, nullptr);
  for (int i = 0; i < nctx; i++)
    if (sacopt_create(i % ndev, nullptr, 0, &ctxs[i])) {
      fprintf(stderr, "sacopt_create: %s\n", sacopt_last_error(nullptr));
      for (sacopt_ctx *c : ctxs) sacopt_destroy(c);
      return 1;
    }
  sacopt_ctx *ctx = ctxs[0];
  auto destroy_all = [&] { for (sacopt_ctx *c : ctxs) sacopt_destroy(c); };
  try {
    std::vector<std::unique_ptr<sacopt::FrameCoder>> fc;
    std::vector<sacopt::FrameCoder *> ptr;
    for (int b = 0; b < nframes; b++) {
      fc.emplace_back(new sacopt::FrameCoder(ctx, 2, n, profile));
      fill(fc[b]->samples[0], fc[b]->samples[1], n, 100 + b);
      fc[b]->SetNumSamples(n);
      ptr.push_back(fc[b].get());
    }
    if (nctx > 1) sacopt::EncodeFrames(ctxs.data(), nctx, ptr.data(), nframes, optimize != 0);
    else sacopt::FrameCoder::PredictBatch(ptr.data(), nframes, optimize != 0);
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
    if (nctx > 1) sacopt::DecodeFrames(ctxs.data(), nctx, ptr.data(), nframes);
    else sacopt::FrameCoder::UnpredictBatch(ptr.data(), nframes);
    std::vector<int> dec_ok(nframes);
    for (int b = 0; b < nframes; b++) {
      sacopt::FrameCoder &f = *fc[b];
      dec_ok[b] = f.samples[0] == keep0[b] && f.samples[1] == keep1[b];
      f.samples[0] = keep0[b]; f.samples[1] = keep1[b]; f.error[1] = keepe1[b];
    }
    int apply_ok = -1;
    if (optimize == 2) {  // predict with optprofile, carry the parameters in the stream, decode with them
      std::vector<std::unique_ptr<sacopt::FrameCoder>> ac, dc;
      std::vector<sacopt::FrameCoder *> ap, dp;
      for (int b = 0; b < nframes; b++) {
        ac.emplace_back(new sacopt::FrameCoder(ctx, 2, n, profile));
        ac[b]->samples[0] = fc[b]->samples[0]; ac[b]->samples[1] = fc[b]->samples[1];
        ac[b]->SetNumSamples(n);
        ap.push_back(ac[b].get());
      }
      if (nctx > 1) sacopt::EncodeFrames(ctxs.data(), nctx, ap.data(), nframes, true, true);
      else sacopt::FrameCoder::PredictBatch(ap.data(), nframes, true, true);
      // stream: header with the flag, one record per frame; the payload stands in for the host's entropy coder (raw residuals)
      std::vector<uint8_t> stream;
      sacopt::SacHeader h;
      h.numsamples = (uint32_t)(nframes * n); h.profile = (uint32_t)profile | sacopt::SAC_PROFILE_PARAMS;
      h.metadata = sacopt::CanonicalWavMetadata(2, 44100, 16, h.numsamples);
      sacopt::WriteHeader(stream, h);
      for (int b = 0; b < nframes; b++) {
        sacopt::FrameRecord rec;
        ac[b]->ToRecord(rec);
        for (int c = 0; c < 2; c++) {
          const uint8_t *p = reinterpret_cast<const uint8_t *>(ac[b]->error[c].data());
          rec.ch[c].payload.assign(p, p + 4 * (size_t)n);
          rec.ch[c].maxbpn = ac[b]->maxbpn[c];
        }
        sacopt::WriteEncoded(stream, rec, true);
      }
      sacopt::SacHeader h2;
      size_t pos = sacopt::ReadHeader(stream.data(), stream.size(), h2);
      apply_ok = pos && h2.has_params() && (int)h2.profile_type() == profile;
      for (int b = 0; b < nframes && apply_ok; b++) {
        sacopt::FrameRecord rec;
        const size_t used = sacopt::ReadEncoded(stream.data() + pos, stream.size() - pos, rec, true);
        if (!used) { apply_ok = 0; break; }
        pos += used;
        dc.emplace_back(new sacopt::FrameCoder(ctx, 2, n, profile));
        dc[b]->FromRecord(rec);
        for (int c = 0; c < 2; c++) memcpy(dc[b]->error[c].data(), rec.ch[c].payload.data(), 4 * (size_t)n);
        dp.push_back(dc[b].get());
      }
      if (apply_ok) {
        if (nctx > 1) sacopt::DecodeFrames(ctxs.data(), nctx, dp.data(), nframes, true);
        else sacopt::FrameCoder::UnpredictBatch(dp.data(), nframes, true);
        for (int b = 0; b < nframes; b++)
          apply_ok = apply_ok && dc[b]->samples[0] == fc[b]->samples[0] && dc[b]->samples[1] == fc[b]->samples[1] && pos == stream.size();
      }
    }
    // FrameCoder::Encode on the device (EncodeBatchOn): the frame records of the stream the reference would write
    std::vector<sacopt::FrameRecord> recs((size_t)nframes);
    sacopt::FrameCoder::EncodeBatchOn(ctx, ptr.data(), nframes, recs.data());
    for (int b = 0; b < nframes; b++) {
      sacopt::FrameCoder &f = *fc[b];
      printf("frame %d in %016" PRIx64 " %016" PRIx64 " err %016" PRIx64 " %016" PRIx64 " s2u %016" PRIx64 " %016" PRIx64
             " map %016" PRIx64 " %016" PRIx64 " maxbpn %d %d %d %d best",
             b, fnv(f.samples[0].data(), n), fnv(f.samples[1].data(), n), fnv(f.error[0].data(), n), fnv(f.error[1].data(), n),
             fnv(f.s2u_error[0].data(), n), fnv(f.s2u_error[1].data(), n), fnv(f.s2u_error_map[0].data(), n),
             fnv(f.s2u_error_map[1].data(), n), f.maxbpn[0], f.maxbpn[1], f.maxbpn_map[0], f.maxbpn_map[1]);
      for (auto &c : f.optprofile.coefs) printf(" %.17g", c.vdef);
      printf(" decode %s", dec_ok[b] ? "ok" : "MISMATCH");
      if (apply_ok >= 0) printf(" apply %s", apply_ok ? "ok" : "MISMATCH");
      std::vector<uint8_t> rec;
      sacopt::WriteEncoded(rec, recs[(size_t)b], false);
      uint64_t h = 1469598103934665603ull;
      for (uint8_t v : rec) h = (h ^ v) * 1099511628211ull;
      printf(" record %zu %016" PRIx64 " mapped %d %d", rec.size(), h, (int)recs[(size_t)b].ch[0].enc_mapped, (int)recs[(size_t)b].ch[1].enc_mapped);
      printf("\n");
    }
  } catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    destroy_all();
    return 1;
  }
  destroy_all();
  return 0;
}
