// TEST INFRASTRUCTURE: sac_amd/host/sacfile.h (through the C entry points of sachost.cpp) fed truncated and garbage input in a
// build with -fsanitize=address,undefined (oracle/Makefile: `make sanitize`): every prefix of a valid stream, and random byte
// flips at every position class. A reader may reject (0) or accept; it must never read outside [p, p + len) -- the inputs are
// exact-size heap copies, so an overread is an AddressSanitizer report -- and never report more bytes than it was given.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sac_amd/host/sacfile.h"
extern "C" {
int sachost_read_header(const uint8_t *p, int len, uint32_t *fields);
int sachost_read_frame(const uint8_t *p, int len, int with_params, uint32_t *numsamples, int32_t *chan, int32_t *ncoef, double *coefs, int32_t *minmax);
}
using namespace sacopt;

static int failures = 0;
static void check(bool ok, const char *what) { if (!ok) { printf("FAIL: %s\n", what); failures++; } }

static void feed(const std::vector<uint8_t> &bytes, size_t hdr_len, bool params) {
  std::vector<uint8_t> *exact = new std::vector<uint8_t>(bytes);  // heap copy of exactly len bytes
  exact->shrink_to_fit();
  const uint8_t *p = exact->data();
  const size_t len = exact->size();
  SacHeader h;
  const size_t used = ReadHeader(p, len, h);
  check(used <= len, "ReadHeader consumed more than it was given");
  uint32_t fields[6];
  check((size_t)sachost_read_header(p, (int)len, fields) == used, "C entry point disagrees with ReadHeader");
  if (len >= hdr_len) {
    FrameRecord f;
    size_t pos = hdr_len;
    for (int k = 0; k < 4 && pos <= len; k++) {
      const size_t u = ReadEncoded(p + pos, len - pos, f, params);
      check(u <= len - pos, "ReadEncoded consumed more than it was given");
      uint32_t ns; int32_t chan[8], nc, mm[4]; double cf[8];
      check((size_t)sachost_read_frame(p + pos, (int)(len - pos), params, &ns, chan, &nc, cf, mm) == u, "C entry point disagrees with ReadEncoded");
      if (!u) break;
      check(f.ncoef >= 0 && f.ncoef <= SAC_MAX_COEFS, "ncoef out of range accepted");
      pos += u;
    }
  }
  delete exact;
}

int main() {
  uint32_t rng = 777;
  auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  for (int params = 0; params < 2; params++) {
    std::vector<uint8_t> s;
    SacHeader h;
    h.numsamples = 20000; h.profile = 2 | (params ? SAC_PROFILE_PARAMS : 0);
    h.metadata = CanonicalWavMetadata(2, 2000, 16, h.numsamples);
    WriteHeader(s, h);
    const size_t hdr_len = s.size();
    for (int fr = 0; fr < 3; fr++) {
      FrameRecord f;
      f.numsamples = fr < 2 ? 8000 : 4000;
      f.ncoef = 8;
      for (int i = 0; i < 8; i++) f.coefs[i] = 0.001 * (i + 1);
      for (int c = 0; c < 2; c++) { f.minval[c] = INT32_MIN; f.maxval[c] = INT32_MAX; f.ch[c].maxbpn = 11 + c; f.ch[c].enc_mapped = c == 1; f.ch[c].payload.resize(37 + 11 * fr + c); for (auto &b : f.ch[c].payload) b = (uint8_t)rnd(); }
      WriteEncoded(s, f, params != 0);
    }
    feed(s, hdr_len, params != 0);                                     // the valid stream
    { SacHeader h2; check(ReadHeader(s.data(), s.size(), h2) == hdr_len && h2.has_params() == (params != 0), "valid header rejected"); }
    for (size_t cut = 0; cut < s.size(); cut++) feed(std::vector<uint8_t>(s.begin(), s.begin() + cut), hdr_len, params != 0);  // every prefix
    for (int it = 0; it < 4000; it++) {                               // byte flips, also in length fields; random tails
      std::vector<uint8_t> g(s);
      const int nflip = 1 + (int)(rnd() % 4);
      for (int k = 0; k < nflip; k++) g[rnd() % g.size()] = (uint8_t)rnd();
      if (it & 1) g.resize(rnd() % (g.size() + 1));
      feed(g, hdr_len, params != 0);
    }
    for (int it = 0; it < 500; it++) {                                // pure garbage of every small size, with and without the magic
      std::vector<uint8_t> g(rnd() % 80);
      for (auto &b : g) b = (uint8_t)rnd();
      if ((it & 3) == 0 && g.size() >= 4) memcpy(g.data(), "SAC2", 4);
      feed(g, 0, params != 0);
    }
  }
  { uint32_t fields[6]; uint8_t b[4] = {0}; check(sachost_read_header(b, -5, fields) == 0, "negative length accepted"); }
  printf(failures ? "san_sacfile: %d failure(s)\n" : "san_sacfile ok\n", failures);
  return failures != 0;
}
