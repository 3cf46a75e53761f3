// Host-side .sac container: header and frame records, byte for byte as the reference writes and reads them
//   Sac::WriteHeader / ReadHeader            src/file/sac.cpp:3-23, :32-48   ("SAC2", 24 bytes + packed WAV chunks)
//   Chunks::PackMetaData                     src/file/wav.cpp:16-28
//   FrameCoder::WriteEncoded / ReadEncoded   src/codec.cpp:480-503, :505-522
// plus the OPT-IN extension of SURVEY.md §8f.2 that carries the optimiser's per-frame parameters:
//   header: bit 8 of the u32 `profile` field (SAC_PROFILE_PARAMS) -- the reference writes 0..2 there (main.cpp:74-76,
//           sac.cpp:17) and its decoder would take a flagged value for an unknown profile, so flagged streams are NOT
//           reference-decodable; unflagged streams are exactly the reference's;
//   frame:  after u32 numsamples: u8 ncoef, ncoef x f64 (little-endian IEEE) = the frame's SacProfile vdef values, then
//           per channel i32 minval, i32 maxval (what StereoHigh::Unpredict clamps to, src/profile.cpp:163,170 -- the
//           reference never transmits them, which is why its --high streams do not round-trip).
// The entropy-coded payloads are opaque here: they come from the reference's BitplaneCoder (src/coder/vle.cpp), which stays
// on the host unchanged. Header-only, no dependency on the GPU library.
#ifndef SACOPT_SACFILE_H
#define SACOPT_SACFILE_H
#include <cstdint>
#include <cstring>
#include <vector>

namespace sacopt {

constexpr uint32_t SAC_PROFILE_PARAMS = 1u << 8;
constexpr int SAC_MAX_COEFS = 8;

inline void put16LH(std::vector<uint8_t> &o, uint32_t v) { o.push_back(v & 0xff); o.push_back((v >> 8) & 0xff); }  // utils.h:132-151
inline void put32LH(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((v >> (8 * i)) & 0xff); }
inline void put64LH(std::vector<uint8_t> &o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((v >> (8 * i)) & 0xff); }
inline uint32_t get16LH(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8; }
inline uint32_t get32LH(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline uint64_t get64LH(const uint8_t *p) { return (uint64_t)get32LH(p) | (uint64_t)get32LH(p + 4) << 32; }

struct SacHeader {
  uint32_t numchannels = 2, samplerate = 44100, bitspersample = 16, numsamples = 0;
  uint32_t profile = 0;           // 0 fast, 1 normal, 2 high; | SAC_PROFILE_PARAMS for the extension
  std::vector<uint8_t> metadata;  // the WAV's chunks as Chunks::PackMetaData packs them
  bool has_params() const { return (profile & SAC_PROFILE_PARAMS) != 0; }
  uint32_t profile_type() const { return profile & 0xff; }
};

// Metadata of a canonical PCM WAV (RIFF, 16-byte fmt, data last), as Wav::ReadHeader collects it (wav.cpp:88-140) and
// PackMetaData lays it out: {id, csize, 'WAVE'} {id, 16, fmt body} {id, csize} -- 44 bytes, i.e. the WAV header itself.
inline std::vector<uint8_t> CanonicalWavMetadata(uint32_t numchannels, uint32_t samplerate, uint32_t bits, uint32_t numsamples) {
  std::vector<uint8_t> m;
  const uint32_t blockalign = numchannels * (bits / 8), datasize = numsamples * blockalign;
  put32LH(m, 0x46464952); put32LH(m, 36 + datasize); put32LH(m, 0x45564157);                    // "RIFF" size "WAVE"
  put32LH(m, 0x20746d66); put32LH(m, 16);                                                        // "fmt "
  put16LH(m, 1); put16LH(m, numchannels); put32LH(m, samplerate); put32LH(m, samplerate * blockalign); put16LH(m, blockalign); put16LH(m, bits);
  put32LH(m, 0x61746164); put32LH(m, datasize);                                                  // "data"
  return m;
}

inline void WriteHeader(std::vector<uint8_t> &o, const SacHeader &h) {  // sac.cpp:3-23
  o.push_back('S'); o.push_back('A'); o.push_back('C'); o.push_back('2');
  put16LH(o, h.numchannels); put32LH(o, h.samplerate); put16LH(o, h.bitspersample); put32LH(o, h.numsamples);
  put32LH(o, h.profile); put32LH(o, (uint32_t)h.metadata.size());
  o.insert(o.end(), h.metadata.begin(), h.metadata.end());
}
// returns bytes consumed, 0 on error (not a SAC2 stream / truncated)
inline size_t ReadHeader(const uint8_t *p, size_t len, SacHeader &h) {  // sac.cpp:32-48
  if (len < 24 || memcmp(p, "SAC2", 4) != 0) return 0;
  h.numchannels = get16LH(p + 4); h.samplerate = get32LH(p + 6); h.bitspersample = get16LH(p + 10);
  h.numsamples = get32LH(p + 12); h.profile = get32LH(p + 16);
  const uint32_t ms = get32LH(p + 20);
  if (len < 24 + (size_t)ms) return 0;
  h.metadata.assign(p + 24, p + 24 + ms);
  return 24 + (size_t)ms;
}

struct ChannelBlock {
  std::vector<uint8_t> payload;  // encoded[ch] buffer, blocksize bytes
  bool enc_mapped = false;       // flag bit 9
  int maxbpn = 0;                // low bits of the flag: maxbpn_map when mapped, else maxbpn (codec.cpp:492-497)
};
struct FrameRecord {
  uint32_t numsamples = 0;
  ChannelBlock ch[2];
  // extension (streams whose header carries SAC_PROFILE_PARAMS)
  int ncoef = 0;
  double coefs[SAC_MAX_COEFS] = {0};
  int32_t minval[2] = {0, 0}, maxval[2] = {0, 0};
};

inline void WriteEncoded(std::vector<uint8_t> &o, const FrameRecord &f, bool with_params) {  // codec.cpp:480-503
  put32LH(o, f.numsamples);
  if (with_params) {
    o.push_back((uint8_t)f.ncoef);
    for (int i = 0; i < f.ncoef; i++) { uint64_t u; memcpy(&u, &f.coefs[i], 8); put64LH(o, u); }
    for (int c = 0; c < 2; c++) { put32LH(o, (uint32_t)f.minval[c]); put32LH(o, (uint32_t)f.maxval[c]); }
  }
  for (int c = 0; c < 2; c++) {
    put32LH(o, (uint32_t)f.ch[c].payload.size());
    uint32_t flag = 0;
    if (f.ch[c].enc_mapped) flag |= 1u << 9;
    flag |= (uint32_t)f.ch[c].maxbpn;
    put16LH(o, flag);
    o.insert(o.end(), f.ch[c].payload.begin(), f.ch[c].payload.end());
  }
}
// returns bytes consumed, 0 on error
inline size_t ReadEncoded(const uint8_t *p, size_t len, FrameRecord &f, bool with_params) {  // codec.cpp:505-522
  size_t pos = 0;
  if (len < 4) return 0;
  f.numsamples = get32LH(p); pos = 4;
  f.ncoef = 0;
  if (with_params) {
    if (len < pos + 1) return 0;
    f.ncoef = p[pos++];
    if (f.ncoef > SAC_MAX_COEFS || len < pos + 8 * (size_t)f.ncoef + 16) return 0;
    for (int i = 0; i < f.ncoef; i++) { const uint64_t u = get64LH(p + pos); memcpy(&f.coefs[i], &u, 8); pos += 8; }
    for (int c = 0; c < 2; c++) { f.minval[c] = (int32_t)get32LH(p + pos); f.maxval[c] = (int32_t)get32LH(p + pos + 4); pos += 8; }
  }
  for (int c = 0; c < 2; c++) {
    if (len < pos + 6) return 0;
    const uint32_t bs = get32LH(p + pos), flag = get16LH(p + pos + 4);
    pos += 6;
    if (len < pos + bs) return 0;
    f.ch[c].enc_mapped = (flag >> 9) != 0;
    f.ch[c].maxbpn = (int)(flag & 0xff);
    f.ch[c].payload.assign(p + pos, p + pos + bs);
    pos += bs;
  }
  return pos;
}

}  // namespace sacopt
#endif
