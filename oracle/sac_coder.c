/* TEST / ANALYSIS INFRASTRUCTURE ONLY -- never linked into or called by the product path.
 *
 * Plain-C restatement of sac's bit-plane context-mixing coder (autch/sac v0.0.4), the host-side step that follows the
 * optimizer / predictor path: SURVEY.md 8f.4 ("bit-plane coder cost"). It exists to say what a frame COSTS in bytes for a
 * given parameter set without the reference (tools/coded_bytes.py), and as the checker of the device kernels that compute
 * the coded size (the optimizer's other objective) and the payload bytes (sac_amd/csrc/bitplane.h). Every function cites the reference file:line it follows
 * (paths relative to /root/reference). Pinned byte for byte against the reference's own Encode() in
 * tests/test_oracle_vs_ref.py::test_bitplane_coder_bytes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PBITS 15 /* src/model/model.h:4-11 */
#define PSCALE (1 << PBITS)
#define PSCALEh (PSCALE >> 1)
#define PSCALEm (PSCALE - 1)
#define WBITS 16

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; } /* src/global.h:16 */
static inline int idiv_signed32(int val, int s) { return val < 0 ? -(((-val) + (1 << (s - 1))) >> s) : (val + (1 << (s - 1))) >> s; }
static inline int64_t idiv_signed64(int64_t val, int64_t s) { return val < 0 ? -(((-val) + (1 << (s - 1))) >> s) : (val + (1 << (s - 1))) >> s; }

/* ------------------------------------------------------------------ LogDomain (src/model/domain.h:6-57) */
static int g_fwd[PSCALE], g_inv[1 << 12], g_dom_max, g_dom_ready = 0;
#define DOM_SCALE 256
#define DOM_DBITS 12
#define DOM_DMIN (-(1 << (DOM_DBITS - 1)))
#define DOM_DMAX ((1 << (DOM_DBITS - 1)) - 1)
static void dom_init(void) {
  if (g_dom_ready) return;
  for (int i = 0; i < PSCALE; i++) g_fwd[i] = (int)floor(log((i + 0.5) / (PSCALE - i - 0.5)) * (double)DOM_SCALE + 0.5); /* :14 */
  g_dom_max = g_fwd[PSCALE - 1];
  for (int i = DOM_DMIN; i <= DOM_DMAX; i++) g_inv[i - DOM_DMIN] = (int)floor((double)PSCALE / (1.0 + exp(-(double)i / (double)DOM_SCALE))); /* :21-22 */
  g_dom_ready = 1;
}
static inline int dom_fwd(int p) { return g_fwd[p]; }
static inline int dom_inv(int x) { return x < DOM_DMIN ? 0 : (x > DOM_DMAX ? PSCALE - 1 : g_inv[x - DOM_DMIN]); } /* :33-38 */

/* ------------------------------------------------------------------ counters (src/model/counter.h:6-65) */
typedef struct { uint16_t p1; } lc16_t; /* LinearCounter16 */
static inline void lc16_update(lc16_t *c, int bit, int L) { /* counter.h:30-35 */
  const int err = (bit << PBITS) - c->p1;
  const int px = (int)c->p1 + idiv_signed32(L * err, PBITS);
  c->p1 = (uint16_t)clampi(px, 1, PSCALEm);
}
typedef struct { uint16_t p1, counter; } lcl_t; /* LinearCounterLimit */
static int g_div[PSCALE]; /* div_tbl, counter.h:38-50 */
static inline void lcl_update(lcl_t *c, int bit, int limit) { /* counter.h:57-65 */
  if (c->counter < limit) c->counter++;
  const int dp = bit ? ((PSCALE - c->p1) * g_div[c->counter]) >> PBITS : -((c->p1 * g_div[c->counter]) >> PBITS);
  c->p1 = (uint16_t)clampi(c->p1 + dp, 1, PSCALEm);
}

/* ------------------------------------------------------------------ NMixLogistic (src/model/mixer.h:56-101) */
#define WRANGE (1 << 19)
typedef struct { int n; int16_t x[4]; int w[4]; int16_t pd; } nmix_t;
static void nmix_init(nmix_t *m, int n) { memset(m, 0, sizeof(*m)); m->n = n; } /* weights start at 0 (:67,70-72) */
static int nmix_predict(nmix_t *m, const int *p) { /* :73-84 */
  int64_t sum = 0;
  for (int i = 0; i < m->n; i++) {
    m->x[i] = (int16_t)dom_fwd(p[i]);
    sum += (int64_t)(m->w[i] * m->x[i]);
  }
  sum = idiv_signed64(sum, WBITS);
  m->pd = (int16_t)clampi(dom_inv((int)sum), 1, PSCALEm);
  return m->pd;
}
static void nmix_update(nmix_t *m, int bit, int rate) { /* :85-93 */
  const int err = (bit << PBITS) - m->pd;
  for (int i = 0; i < m->n; i++) {
    const int de = idiv_signed32(m->x[i] * err, DOM_DBITS);
    m->w[i] = clampi(m->w[i] + idiv_signed32(de * rate, DOM_DBITS), -WRANGE, WRANGE - 1);
  }
}

/* ------------------------------------------------------------------ SSENL<32> (src/model/sse.h:88-125) */
#define SSE_N 32
typedef struct { int tscale, xscale, lb; uint16_t p_quant; lcl_t map[2][SSE_N + 1]; } ssenl_t;
static void ssenl_init(ssenl_t *s) { /* :94-105 */
  s->tscale = g_dom_max;
  s->xscale = (2 * s->tscale) / (SSE_N - 1);
  if (s->xscale == 0) s->xscale = 1;
  for (int i = 0; i <= SSE_N; i++) {
    const int x = dom_inv(i * s->xscale - s->tscale);
    s->map[0][i].p1 = s->map[1][i].p1 = (uint16_t)x;
    s->map[0][i].counter = s->map[1][i].counter = 0;
  }
  s->lb = 0;
  s->p_quant = 0;
}
static int ssenl_predict(ssenl_t *s, int p1) { /* :106-118 */
  int pq = dom_fwd(p1) + s->tscale;
  pq = pq < 0 ? 0 : pq;
  pq = pq > 2 * s->tscale ? 2 * s->tscale : pq;
  s->p_quant = (uint16_t)(pq / s->xscale);
  const int p_mod = pq - (s->p_quant * s->xscale);
  const int pl = s->map[s->lb][s->p_quant].p1, ph = s->map[s->lb][s->p_quant + 1].p1;
  const int px = (pl * (s->xscale - p_mod) + ph * p_mod) / s->xscale;
  return clampi(px, 1, PSCALEm);
}
static void ssenl_update(ssenl_t *s, int bit, int rate) { /* :119-124 */
  lcl_update(&s->map[s->lb][s->p_quant], bit, rate);
  lcl_update(&s->map[s->lb][s->p_quant + 1], bit, rate);
  s->lb = bit;
}

/* ------------------------------------------------------------------ RangeCoderSH, encoder side (src/coder/range.cpp:58-90) */
typedef struct { uint32_t range, ffnum, cache; uint64_t lowc; uint8_t *out; size_t pos, cap; } rc_t;
static void rc_put(rc_t *r, int v) { if (r->out && r->pos < r->cap) r->out[r->pos] = (uint8_t)v; r->pos++; } /* BufIO::PutByte */
static void rc_init(rc_t *r, uint8_t *out, size_t cap) { r->range = 0xFFFFFFFFu; r->lowc = 0; r->ffnum = r->cache = 0; r->out = out; r->cap = cap; r->pos = 0; }
static void rc_shift_low(rc_t *r) { /* :80-90 */
  const uint32_t carry = (uint32_t)(r->lowc >> 32), low = (uint32_t)r->lowc;
  if (low < 0xFF000000u || carry) {
    rc_put(r, (int)(r->cache + carry));
    for (; r->ffnum != 0; r->ffnum--) rc_put(r, (int)(carry - 1));
    r->cache = low >> 24;
  } else r->ffnum++;
  r->lowc = (uint64_t)(low << 8);
}
static void rc_encode_bit(rc_t *r, uint32_t p1, int bit) { /* :65-70, SCALE_RANGE range.h:20 */
  const uint32_t rnew = (uint32_t)(((uint64_t)r->range * ((uint32_t)(PSCALE - p1) << (32 - PBITS))) >> 32);
  if (bit) { r->range -= rnew; r->lowc += rnew; } else r->range = rnew;
  while (r->range < 0x01000000u) { r->range <<= 8; rc_shift_low(r); }
}
static void rc_stop(rc_t *r) { for (int i = 0; i < 5; i++) rc_shift_low(r); } /* :60-63 */

/* ------------------------------------------------------------------ MapEncoder::Encode (src/coder/map.cpp:3-72, map.h:10-36) */
typedef struct { lc16_t cnt[24]; nmix_t mixl[4], mixh[4], finalmix; ssenl_t sse; } mapenc_t;
static void mapenc_encode(rc_t *rc, const uint8_t *ul, const uint8_t *uh) {
  mapenc_t *m = (mapenc_t *)malloc(sizeof(mapenc_t));
  for (int i = 0; i < 24; i++) m->cnt[i].p1 = PSCALEh;
  for (int i = 0; i < 4; i++) { nmix_init(&m->mixl[i], 4); nmix_init(&m->mixh[i], 4); }
  nmix_init(&m->finalmix, 2);
  ssenl_init(&m->sse);
  for (int i = 1; i <= 1 << 15; i++)
    for (int hi = 0; hi < 2; hi++) {
      lc16_t *pc[4];
      nmix_t *mix;
      int bit;
      if (!hi) { /* PredictLow, map.cpp:8-20 */
        const int ctx1 = ul[i - 1], ctx2 = uh[i - 1], ctx3 = i > 1 ? ul[i - 2] : 0;
        pc[0] = &m->cnt[ctx1]; pc[1] = &m->cnt[2 + ctx2]; pc[2] = &m->cnt[4 + (ctx1 << 1) + ctx3]; pc[3] = &m->cnt[8 + (ctx1 << 1) + ctx2];
        mix = &m->mixl[ctx1 + (ctx3 << 1)];
        bit = ul[i];
      } else { /* PredictHigh, :22-34 */
        const int ctx1 = uh[i - 1], ctx2 = ul[i], ctx3 = i > 1 ? uh[i - 2] : 0;
        pc[0] = &m->cnt[12 + ctx1]; pc[1] = &m->cnt[12 + 2 + ctx2]; pc[2] = &m->cnt[12 + 4 + (ctx1 << 1) + ctx3]; pc[3] = &m->cnt[12 + 8 + (ctx1 << 1) + ctx2];
        mix = &m->mixh[ctx1 + (ctx3 << 1)];
        bit = uh[i];
      }
      int p[4] = {pc[0]->p1, pc[1]->p1, pc[2]->p1, pc[3]->p1};
      const int pm = nmix_predict(mix, p);
      int vp[2] = {ssenl_predict(&m->sse, pm), pm}; /* PredictSSE, :45-49 */
      rc_encode_bit(rc, (uint32_t)nmix_predict(&m->finalmix, vp), bit);
      for (int k = 0; k < 4; k++) lc16_update(pc[k], bit, 500); /* Update, :36-43 */
      nmix_update(mix, bit, 1000);
      ssenl_update(&m->sse, bit, 300); /* UpdateSSE, :51-55 */
      nmix_update(&m->finalmix, bit, 500);
    }
  free(m);
}

/* ------------------------------------------------------------------ BitplaneCoder::Encode (src/coder/vle.cpp:3-138, vle.h:13-47) */
typedef struct {
  lc16_t bpn1[1 << 10], bpn2[1 << 10];
  nmix_t lmixref[32], lmixsig[32], finalmix;
  ssenl_t sse[2048];
} bpc_t;
static void bitplane_encode(rc_t *rc, const int32_t *abuf, int numsamples, int maxbpn) {
  bpc_t *b = (bpc_t *)malloc(sizeof(bpc_t));
  int *msb = (int *)calloc((size_t)numsamples + 1, sizeof(int));
  for (int i = 0; i < 1 << 10; i++) b->bpn1[i].p1 = b->bpn2[i].p1 = PSCALEh;
  for (int i = 0; i < 32; i++) { nmix_init(&b->lmixref[i], 3); nmix_init(&b->lmixsig[i], 2); }
  nmix_init(&b->finalmix, 2); /* only finalmix[0] is ever used (vle.cpp:102) */
  for (int i = 0; i < 2048; i++) ssenl_init(&b->sse[i]);
  for (int bpn = maxbpn; bpn >= 0; bpn--) {
    uint32_t state = 0;
    for (int s = 0; s < numsamples; s++) {
      const int bit = (abuf[s] >> bpn) & 1;
      const int mixctx = (int)(state & 7);
      lc16_t *pc[3];
      int npc, p1, significant = msb[s] != 0; /* sigst[0] (vle.cpp:19, :122) */
      nmix_t *mix;
      if (significant) { /* PredictRef, vle.cpp:32-60 */
        const int val = abuf[s];
        const int lval = s > 0 ? abuf[s - 1] : 0, lval2 = s > 1 ? abuf[s - 2] : 0;
        const int nval = s < numsamples - 1 ? abuf[s + 1] : 0, nval2 = s < numsamples - 2 ? abuf[s + 2] : 0;
        int lctx = 0;
        const int tval = val >> (bpn + 1), tval1 = tval << 1;
        if ((lval >> bpn) > tval1) lctx += 1 << 0;
        if ((nval >> (bpn + 1)) > tval) lctx += 1 << 1;
        if ((lval2 >> bpn) > tval1) lctx += 1 << 2;
        if ((nval2 >> (bpn + 1)) > tval) lctx += 1 << 3;
        pc[0] = &b->bpn2[0]; pc[1] = &b->bpn1[1 + msb[s]]; pc[2] = &b->bpn1[1 + 32 + lctx];
        npc = 3;
        mix = &b->lmixref[mixctx];
      } else { /* PredictSig, vle.cpp:71-92, with GetSigState :17-28 */
        int ctx1 = 0;
        for (int k = 1; k <= 4; k++) {
          if (s - k >= 0 && msb[s - k]) ctx1 += 1 << (2 * k - 2);           /* sigst[1], [3], [5], [7] */
          if (s + k <= numsamples - 1 && msb[s + k]) ctx1 += 1 << (2 * k - 1); /* sigst[2], [4], [6], [8] */
        }
        pc[0] = &b->bpn2[0]; pc[1] = &b->bpn2[1 + ctx1];
        npc = 2;
        mix = &b->lmixsig[mixctx];
      }
      int vp[3] = {pc[0]->p1, pc[1]->p1, npc == 3 ? pc[2]->p1 : 0};
      p1 = nmix_predict(mix, vp);
      ssenl_t *psse = &b->sse[(bpn << 5) + msb[s]]; /* PredictSSE, vle.cpp:96-102 */
      int vq[2] = {ssenl_predict(psse, p1), p1};
      rc_encode_bit(rc, (uint32_t)nmix_predict(&b->finalmix, vq), bit);
      for (int k = 0; k < npc; k++) lc16_update(pc[k], bit, 350); /* UpdateRef / UpdateSig, vle.cpp:62-69, 94-101 */
      nmix_update(mix, bit, 800);
      state = (state << 1) + (significant ? 0 : 1);
      ssenl_update(psse, bit, 1000); /* UpdateSSE, vle.cpp:104-108 */
      nmix_update(&b->finalmix, bit, 300);
      if (!significant && bit) msb[s] = bpn;
    }
  }
  free(msb);
  free(b);
}

static void coder_init(void) {
  dom_init();
  for (int i = 0; i < PSCALE; i++) g_div[i] = PSCALE / (i + 3);
}

/* FrameCoder::EncodeMonoFrame_Normal (src/codec.cpp:237-247): s2u residuals -> payload. Returns the payload size; bytes go to
 * out[0..cap) when out != NULL (the size is exact either way). */
int oracle_encode_mono_normal(const int32_t *s2u, int numsamples, int maxbpn, uint8_t *out, int cap) {
  coder_init();
  rc_t rc;
  rc_init(&rc, out, (size_t)(cap > 0 ? cap : 0));
  bitplane_encode(&rc, s2u, numsamples, maxbpn);
  rc_stop(&rc);
  return (int)rc.pos;
}
/* FrameCoder::EncodeMonoFrame_Mapped (src/codec.cpp:249-265): the used-value tables (usedl[0..32768] then usedh[0..32768], one
 * byte each) coded by MapEncoder, then the mapped s2u residuals. */
int oracle_encode_mono_mapped(const int32_t *s2u_map, int numsamples, int maxbpn_map, const uint8_t *used, uint8_t *out, int cap) {
  coder_init();
  rc_t rc;
  rc_init(&rc, out, (size_t)(cap > 0 ? cap : 0));
  mapenc_encode(&rc, used, used + 32769);
  bitplane_encode(&rc, s2u_map, numsamples, maxbpn_map);
  rc_stop(&rc);
  return (int)rc.pos;
}
