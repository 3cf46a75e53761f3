// Frames of a file over several GPUs: the Codec::EncodeFile / DecodeFile loops (src/codec.cpp:530-583) with the frames cut
// into contiguous ranges, one range per context (one context per GPU), one host thread per context. Frames are independent
// (fresh predictor and coder per frame, src/codec.cpp:156-159,240-242), so no range needs anything from another: there is no
// collective. Results land in the FrameCoders, i.e. in frame order, whatever order the ranges finish in; the caller then runs the
// unchanged host coder (Encode / WriteEncoded) over them in order.
// Header-only, on top of framecoder.h. libsacopt: a context is not thread-safe, different contexts are independent.
#ifndef SACOPT_SHARDER_H
#define SACOPT_SHARDER_H
#include <exception>
#include <thread>
#include <vector>

#include "framecoder.h"

namespace sacopt {

// [lo, hi) of shard `rank` of `world`: contiguous, balanced, the first nframes % world shards one frame longer
// (the same cut as sac_amd/dist.py: frame_shard)
inline void FrameShard(int nframes, int world, int rank, int *lo, int *hi) {
  const int base = nframes / world, extra = nframes % world;
  *lo = rank * base + (rank < extra ? rank : extra);
  *hi = *lo + base + (rank < extra ? 1 : 0);
}

namespace detail {
template <class F>
inline void for_each_shard(int nctx, int nframes, F &&work) {
  if (nctx <= 0) throw std::invalid_argument("no context");
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)nctx);
  for (int r = 0; r < nctx; r++) {
    int lo, hi;
    FrameShard(nframes, nctx, r, &lo, &hi);
    if (hi <= lo) continue;
    th.emplace_back([&, r, lo, hi] {
      try { work(r, lo, hi); } catch (...) { err[(size_t)r] = std::current_exception(); }
    });
  }
  for (auto &t : th) t.join();
  for (auto &e : err) if (e) std::rethrow_exception(e);  // the first failed shard's error, after every thread has ended
}
}  // namespace detail

// FrameCoder::Predict (with Optimize when `optimize`) for all frames, sharded over ctxs[0..nctx). max_batch > 0 cuts a
// shard's range into library calls of at most that many frames (0: one call per shard).
inline void EncodeFrames(sacopt_ctx *const *ctxs, int nctx, FrameCoder *const *fc, int nframes, bool optimize, bool apply = false, int max_batch = 0) {
  detail::for_each_shard(nctx, nframes, [&](int r, int lo, int hi) {
    const int step = max_batch > 0 ? max_batch : hi - lo;
    for (int b = lo; b < hi; b += step) FrameCoder::PredictBatchOn(ctxs[r], fc + b, (hi - b < step) ? hi - b : step, optimize, apply);
  });
}

// FrameCoder::Unpredict for all frames, sharded the same way
inline void DecodeFrames(sacopt_ctx *const *ctxs, int nctx, FrameCoder *const *fc, int nframes, bool with_optprofile = false, int max_batch = 0) {
  detail::for_each_shard(nctx, nframes, [&](int r, int lo, int hi) {
    const int step = max_batch > 0 ? max_batch : hi - lo;
    for (int b = lo; b < hi; b += step) FrameCoder::UnpredictBatchOn(ctxs[r], fc + b, (hi - b < step) ? hi - b : step, with_optprofile);
  });
}

}  // namespace sacopt
#endif
