// C entry points over sac_amd/host/sacfile.h (the host-side .sac container writer/reader), so that the Python tests can
// drive the same C++ code a C++ host links header-only. Built into sac_amd/libsachost.so; pure host code, no GPU.
#include <cstring>

#include "sacfile.h"

using namespace sacopt;

static int emit(const std::vector<uint8_t> &v, uint8_t *out, int cap) {
  if ((int)v.size() > cap) return -(int)v.size();
  if (!v.empty()) memcpy(out, v.data(), v.size());
  return (int)v.size();
}

extern "C" {
int sachost_canonical_wav_metadata(uint32_t numch, uint32_t rate, uint32_t bits, uint32_t numsamples, uint8_t *out, int cap) {
  return emit(CanonicalWavMetadata(numch, rate, bits, numsamples), out, cap);
}
int sachost_write_header(uint32_t numch, uint32_t rate, uint32_t bits, uint32_t numsamples, uint32_t profile, const uint8_t *meta,
                         int metalen, uint8_t *out, int cap) {
  SacHeader h;
  h.numchannels = numch; h.samplerate = rate; h.bitspersample = bits; h.numsamples = numsamples; h.profile = profile;
  if (metalen > 0) h.metadata.assign(meta, meta + metalen);
  std::vector<uint8_t> v;
  WriteHeader(v, h);
  return emit(v, out, cap);
}
// fields: numch, rate, bits, numsamples, profile, metadata length; the metadata starts at byte 24. Returns bytes consumed (0 = error).
int sachost_read_header(const uint8_t *p, int len, uint32_t *fields) {
  SacHeader h;
  if (len < 0 || !p) return 0;
  const size_t used = ReadHeader(p, (size_t)len, h);
  if (!used) return 0;
  fields[0] = h.numchannels; fields[1] = h.samplerate; fields[2] = h.bitspersample; fields[3] = h.numsamples;
  fields[4] = h.profile; fields[5] = (uint32_t)h.metadata.size();
  return (int)used;
}
// chan = {mapped0, maxbpn0, mapped1, maxbpn1}; minmax = {min0, max0, min1, max1}
int sachost_write_frame(uint32_t numsamples, const uint8_t *p0, int len0, const uint8_t *p1, int len1, const int32_t *chan,
                        int with_params, int ncoef, const double *coefs, const int32_t *minmax, uint8_t *out, int cap) {
  FrameRecord f;
  f.numsamples = numsamples;
  f.ch[0].payload.assign(p0, p0 + len0); f.ch[1].payload.assign(p1, p1 + len1);
  for (int c = 0; c < 2; c++) { f.ch[c].enc_mapped = chan[2 * c] != 0; f.ch[c].maxbpn = chan[2 * c + 1]; }
  if (with_params) {
    if (ncoef < 0 || ncoef > SAC_MAX_COEFS) return 0;
    f.ncoef = ncoef;
    for (int i = 0; i < ncoef; i++) f.coefs[i] = coefs[i];
    for (int c = 0; c < 2; c++) { f.minval[c] = minmax[2 * c]; f.maxval[c] = minmax[2 * c + 1]; }
  }
  std::vector<uint8_t> v;
  WriteEncoded(v, f, with_params != 0);
  return emit(v, out, cap);
}
// chan = {payload offset, len, mapped, maxbpn} x 2 (offsets relative to p). Returns bytes consumed (0 = error).
int sachost_read_frame(const uint8_t *p, int len, int with_params, uint32_t *numsamples, int32_t *chan, int32_t *ncoef,
                       double *coefs, int32_t *minmax) {
  FrameRecord f;
  if (len < 0 || !p) return 0;
  const size_t used = ReadEncoded(p, (size_t)len, f, with_params != 0);
  if (!used) return 0;
  *numsamples = f.numsamples;
  size_t pos = 4 + (with_params ? 1 + 8 * (size_t)f.ncoef + 16 : 0);
  for (int c = 0; c < 2; c++) {
    pos += 6;
    chan[4 * c] = (int32_t)pos; chan[4 * c + 1] = (int32_t)f.ch[c].payload.size();
    chan[4 * c + 2] = f.ch[c].enc_mapped ? 1 : 0; chan[4 * c + 3] = f.ch[c].maxbpn;
    pos += f.ch[c].payload.size();
  }
  *ncoef = f.ncoef;
  for (int i = 0; i < f.ncoef; i++) coefs[i] = f.coefs[i];
  for (int c = 0; c < 2; c++) { minmax[2 * c] = f.minval[c]; minmax[2 * c + 1] = f.maxval[c]; }
  return (int)used;
}
}
