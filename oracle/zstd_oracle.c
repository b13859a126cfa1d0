/*
 * oracle/zstd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see zstd_oracle.h).
 *
 * Plain-C restatement of the zstd frame decoder (RFC 8878 / SURVEY.md Appendix A),
 * organised in the phases K0..K7 of SURVEY.md section 2 so that every GPU phase has a
 * CPU twin whose intermediates (literal buffer, sequence triples) can be diffed.
 *
 *   K0 frame/block headers  -> parse_frame_header(), decode_frame() block loop   (A.1, A.2)
 *   K1 Huffman table build   -> huf_read_table()                                 (A.4)
 *   K2 Huffman literals      -> huf_decode_stream(), decode_literals()           (A.4)
 *   K3 FSE table build       -> fse_read_ncount(), fse_build()                   (A.3)
 *   K4 FSE sequence decode   -> decode_sequences()                               (A.5)
 *   K5 sequence execute      -> execute_sequences()                              (A.5)
 *   K6 raw / RLE blocks      -> decode_frame()                                   (A.2)
 *   K7 XXH64 content check   -> ozs_xxh64()                                      (A.6)
 *
 * Behaviour choices that the spec leaves open follow libzstd 1.5.6 (the version the
 * reference pins, Cargo.lock:2371-2396) as used by copy_decode (reference
 * src/main.rs:463-467): every concatenated frame is decoded, skippable frames are
 * skipped, the checksum is verified, any malformed input is an error.
 */
#include "zstd_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ERR(c) do { return (c); } while (0)
#define CHECK(cond, c) do { if (!(cond)) return (c); } while (0)

static inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static inline uint32_t rd24(const uint8_t* p) { return rd16(p) | ((uint32_t)p[2] << 16); }
static inline uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }
static inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static inline int highbit(uint32_t v) { return 31 - __builtin_clz(v); }

/* ------------------------------------------------------------------ K7: XXH64 (A.6) */
#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
static inline uint64_t xmerge(uint64_t h, uint64_t v) { v = xround(0, v); h ^= v; return h * XP1 + XP4; }

uint64_t ozs_xxh64(const uint8_t* p, size_t n, uint64_t seed) {
    const uint8_t* end = p + n;
    uint64_t h;
    if (n >= 32) {
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t* lim = end - 32;
        do {
            v1 = xround(v1, rd64(p)); v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16)); v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)n;
    while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl64(h, 27) * XP1 + XP4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * XP1; h = rotl64(h, 23) * XP2 + XP3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p) * XP5; h = rotl64(h, 11) * XP1; p++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------ bit readers */
/* bits [bitpos, bitpos+n) of the little-endian integer p[0..nbytes); indices < 0 read as 0.
 * n <= 56. */
static inline uint64_t bits_at(const uint8_t* p, size_t nbytes, int64_t bitpos, int n) {
    if (n == 0) return 0;
    if (bitpos < 0) {
        int64_t neg = -bitpos;
        if (neg >= n) return 0;
        return bits_at(p, nbytes, 0, n - (int)neg) << neg;
    }
    size_t byte = (size_t)(bitpos >> 3);
    int sh = (int)(bitpos & 7);
    uint64_t v = 0;
    if (byte < nbytes) {
        size_t avail = nbytes - byte;
        if (avail >= 8) v = rd64(p + byte);
        else memcpy(&v, p + byte, avail); /* host is little-endian */
    }
    v >>= sh;
    return v & ((1ULL << n) - 1);
}

/* backward bitstream (Appendix A preamble): pos = data bits still unread */
typedef struct { const uint8_t* p; size_t n; int64_t pos; } bbr;
static int bbr_init(bbr* b, const uint8_t* p, size_t n) {
    if (n == 0) return OZS_E_CORRUPT;
    if (p[n - 1] == 0) return OZS_E_CORRUPT; /* marker bit missing */
    b->p = p; b->n = n;
    b->pos = (int64_t)(n - 1) * 8 + highbit(p[n - 1]);
    return 0;
}
static inline uint32_t bbr_read(bbr* b, int n) { b->pos -= n; return (uint32_t)bits_at(b->p, b->n, b->pos, n); }
static inline uint32_t bbr_peek(const bbr* b, int n) { return (uint32_t)bits_at(b->p, b->n, b->pos - n, n); }

/* ------------------------------------------------------------------ K3: FSE tables (A.3) */
typedef struct { uint16_t base; uint8_t sym; uint8_t nb; } fse_ent;
typedef struct { fse_ent e[512]; int log; } fse_tab;

/* Reads a normalized-count header (forward bitstream).  Returns bytes consumed (>0) or <0. */
static int fse_read_ncount(const uint8_t* src, size_t n, int max_log, int max_sym,
                           int16_t* norm, int* nsym_out, int* log_out) {
    CHECK(n >= 1, OZS_E_CORRUPT);
    int64_t bit = 0;
    int64_t limit = (int64_t)n * 8;
    int al = 5 + (int)bits_at(src, n, bit, 4); bit += 4;
    CHECK(al <= max_log, OZS_E_CORRUPT);
    int remaining = 1 << al;
    int sym = 0;
    while (remaining > 0 && sym <= max_sym) {
        int nbits = highbit((uint32_t)(remaining + 1)) + 1;
        CHECK(bit < limit, OZS_E_CORRUPT);
        int val = (int)bits_at(src, n, bit, nbits); bit += nbits; /* bits past the end read 0 */
        int lower = (1 << (nbits - 1)) - 1;
        int thr = (1 << nbits) - 1 - (remaining + 1);
        if ((val & lower) < thr) { bit -= 1; val &= lower; }
        else if (val > lower) val -= thr;
        int p = val - 1;
        remaining -= (p < 0) ? 1 : p;
        CHECK(remaining >= 0, OZS_E_CORRUPT);
        norm[sym++] = (int16_t)p;
        if (p == 0) {
            for (;;) {
                CHECK(bit < limit, OZS_E_CORRUPT);
                int r = (int)bits_at(src, n, bit, 2); bit += 2;
                for (int i = 0; i < r; i++) { CHECK(sym <= max_sym, OZS_E_CORRUPT); norm[sym++] = 0; }
                if (r != 3) break;
            }
        }
    }
    CHECK(remaining == 0, OZS_E_CORRUPT);
    CHECK(sym <= max_sym + 1, OZS_E_CORRUPT);
    CHECK(bit <= limit, OZS_E_CORRUPT);
    *nsym_out = sym; *log_out = al;
    return (int)((bit + 7) >> 3);
}

static int fse_build(fse_tab* t, const int16_t* norm, int nsym, int log) {
    int size = 1 << log, high = size;
    uint16_t next[256];
    CHECK(log <= 9 && nsym <= 256, OZS_E_CORRUPT);
    for (int s = 0; s < nsym; s++)
        if (norm[s] == -1) { high--; t->e[high].sym = (uint8_t)s; next[s] = 1; }
    int step = (size >> 1) + (size >> 3) + 3, pos = 0, mask = size - 1;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] <= 0) continue;
        next[s] = (uint16_t)norm[s];
        for (int i = 0; i < norm[s]; i++) {
            t->e[pos].sym = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    CHECK(pos == 0, OZS_E_CORRUPT);
    for (int i = 0; i < size; i++) {
        int s = t->e[i].sym;
        uint32_t d = next[s]++;
        int nb = log - highbit(d);
        t->e[i].nb = (uint8_t)nb;
        t->e[i].base = (uint16_t)((d << nb) - (uint32_t)size);
    }
    t->log = log;
    return 0;
}
static void fse_rle(fse_tab* t, int sym) { t->e[0].sym = (uint8_t)sym; t->e[0].nb = 0; t->e[0].base = 0; t->log = 0; }

/* predefined distributions and code tables (A.5) */
static const int16_t LL_DEF[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const int16_t ML_DEF[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
static const int16_t OF_DEF[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const uint32_t LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
static const uint8_t LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const uint32_t ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
static const uint8_t ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};

/* ------------------------------------------------------------------ K1: Huffman table (A.4) */
typedef struct { uint8_t sym[4096]; uint8_t len[4096]; int log; } huf_tab; /* up to HUF_TABLELOG_MAX = 12 bits */

/* Decodes the FSE-compressed weight stream (two interleaved states). */
static int huf_fse_weights(const uint8_t* src, size_t n, uint8_t* w, int* nw) {
    int16_t norm[256]; int nsym, log;
    int hdr = fse_read_ncount(src, n, 6, 255, norm, &nsym, &log);
    CHECK(hdr > 0, OZS_E_CORRUPT);
    fse_tab* t = (fse_tab*)malloc(sizeof(fse_tab));
    if (!t) return OZS_E_CORRUPT;
    int rc = fse_build(t, norm, nsym, log);
    if (rc) { free(t); return rc; }
    bbr b;
    if ((size_t)hdr >= n || bbr_init(&b, src + hdr, n - hdr)) { free(t); return OZS_E_CORRUPT; }
    uint32_t s1 = bbr_read(&b, log), s2 = bbr_read(&b, log);
    int k = 0; rc = OZS_E_CORRUPT;
    for (;;) {
        if (k > 253) break;
        w[k++] = t->e[s1].sym; s1 = t->e[s1].base + bbr_read(&b, t->e[s1].nb);
        if (b.pos < 0) { w[k++] = t->e[s2].sym; rc = 0; break; }
        if (k > 253) break;
        w[k++] = t->e[s2].sym; s2 = t->e[s2].base + bbr_read(&b, t->e[s2].nb);
        if (b.pos < 0) { w[k++] = t->e[s1].sym; rc = 0; break; }
    }
    free(t);
    *nw = k;
    return rc;
}

/* Reads a Huffman tree description; returns bytes consumed (>0) or <0. */
static int huf_read_table(huf_tab* t, const uint8_t* src, size_t n) {
    uint8_t w[256]; int nw = 0; int used;
    CHECK(n >= 1, OZS_E_CORRUPT);
    int hb = src[0];
    if (hb >= 128) {
        nw = hb - 127;
        int bytes = (nw + 1) / 2;
        CHECK((size_t)(1 + bytes) <= n, OZS_E_CORRUPT);
        for (int i = 0; i < nw; i++) {
            uint8_t b = src[1 + i / 2];
            w[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        used = 1 + bytes;
    } else {
        CHECK(hb >= 1 && (size_t)(1 + hb) <= n, OZS_E_CORRUPT);
        int rc = huf_fse_weights(src + 1, (size_t)hb, w, &nw);
        if (rc) return rc;
        used = 1 + hb;
    }
    uint32_t total = 0; int rank[16] = {0};
    for (int i = 0; i < nw; i++) {
        CHECK(w[i] <= 12, OZS_E_CORRUPT);
        rank[w[i]]++;
        if (w[i]) total += 1u << (w[i] - 1);
    }
    CHECK(total != 0, OZS_E_CORRUPT);
    int maxbits = highbit(total) + 1;
    CHECK(maxbits <= 12, OZS_E_CORRUPT); /* libzstd's limit (HUF_TABLELOG_MAX): a tree of depth 12 is accepted by 1.4.8 and 1.5.7 alike
                                            (tests/golden: hand_huf12_*), although the format's text says 11 and no encoder emits one */
    uint32_t left = (1u << maxbits) - total;
    CHECK((left & (left - 1)) == 0, OZS_E_CORRUPT);
    int wl = highbit(left) + 1;
    w[nw++] = (uint8_t)wl; rank[wl]++;
    CHECK(rank[1] >= 2 && (rank[1] & 1) == 0, OZS_E_CORRUPT); /* libzstd HUF_readStats */
    /* canonical fill: weight 1 (longest codes) first, symbols ascending within a weight */
    uint32_t start[16]; uint32_t pos = 0;
    for (int r = 1; r <= maxbits; r++) { start[r] = pos; pos += (uint32_t)rank[r] << (r - 1); }
    CHECK(pos == (1u << maxbits), OZS_E_CORRUPT);
    for (int s = 0; s < nw; s++) {
        int r = w[s]; if (!r) continue;
        uint32_t cnt = 1u << (r - 1);
        for (uint32_t i = 0; i < cnt; i++) { t->sym[start[r] + i] = (uint8_t)s; t->len[start[r] + i] = (uint8_t)(maxbits + 1 - r); }
        start[r] += cnt;
    }
    t->log = maxbits;
    return used;
}

/* ------------------------------------------------------------------ K2: Huffman literals (A.4) */
static int g_lit_inexact; /* the last verdict was "a Huffman literal stream was not consumed exactly" */
int ozs_last_verdict_lit_inexact(void) { return g_lit_inexact; }
/* Which rule decides about a literal stream that is not consumed exactly.
 *   1 (default): the reference's pin, libzstd 1.5.x (Cargo.lock:2371-2396; >= 1.5.4).  Four-stream sections go through its "fast"
 *      decoding loops whenever they are eligible (huf_fast_eligible below).  Those loops decode exactly `regen` symbols and never look
 *      at where a stream's read point ends up:
 *        - an UNDER-consumed stream is accepted, its leftover bits are ignored (g_lit_lenient);
 *        - a last byte of zero -- no end mark -- is eight data bits (HUF_initFastDStream);
 *        - a stream that RUNS OUT reads on into the bytes in front of it -- the previous stream's, then the jump table's: the loops'
 *          lower bound is the section's first byte, not the stream's (g_lit_through).  libzstd refuses such a stream only if its read
 *          pointer stands more than 8 bytes below the stream when the five-symbols-a-round loop ends (about the last ten symbols of
 *          stream 1 are decoded behind that test), which depends on which of its two decoders (X1 / X2) the size heuristic picked;
 *          here the stream is accepted whenever the section's bytes suffice.  tests/test_oracle.py counts both sides by name;
 *        - a stream that needs bits from below the section's first byte is rejected (g_lit_over): libzstd decodes on from a bit
 *          container it no longer refills.
 *      One-stream sections, ineligible four-stream sections and trees of depth 12 take libzstd's checked loops: exact consumption or
 *      corruption_detected, as RFC 8878 4.2.2 says.
 *   0: RFC 8878 / libzstd 1.4.x: every stream must be consumed exactly. */
static int g_huf_rule = 1;
void ozs_set_huf_rule(int rule) { g_huf_rule = rule; }
static int g_lit_lenient; /* the last decode ACCEPTED a literal stream that was not consumed exactly (rule 1) */
int ozs_last_verdict_lit_lenient(void) { return g_lit_lenient; }
static int g_lit_through; /* the last decode ACCEPTED a literal stream that ran out and read on into the bytes in front of it (rule 1) */
int ozs_last_verdict_lit_through(void) { return g_lit_through; }
static int g_lit_over;    /* the last verdict was "a stream of a fast-loop section needed bits from below the section's first byte" (rule 1) */
int ozs_last_verdict_lit_over(void) { return g_lit_over; }
static int huf_decode_stream(const huf_tab* t, const uint8_t* src, size_t n, uint8_t* out, size_t nout, int fast, size_t below) {
    bbr b;
    if (fast) { /* libzstd 1.5 HUF_initFastDStream: no end mark is no error, the byte is data */
        if (n == 0) return OZS_E_CORRUPT;
        b.p = src; b.n = n; b.pos = src[n - 1] ? (int64_t)(n - 1) * 8 + highbit(src[n - 1]) : (int64_t)n * 8;
        b.p = src - below; b.n = n + below; b.pos += 8 * (int64_t)below; /* the loops' lower bound is the section's first byte */
    } else {
        int rc = bbr_init(&b, src, n);
        if (rc) return rc;
    }
    for (size_t i = 0; i < nout; i++) {
        uint32_t idx = bbr_peek(&b, t->log);
        out[i] = t->sym[idx];
        b.pos -= t->len[idx];
    }
    if (fast) {
        if (b.pos < 0) { g_lit_inexact = 1; g_lit_over = 1; return OZS_E_CORRUPT; }
        if (b.pos > 8 * (int64_t)below) g_lit_lenient = 1;
        else if (b.pos < 8 * (int64_t)below) g_lit_through = 1;
        return 0;
    }
    if (b.pos != 0) { g_lit_inexact = 1; return OZS_E_CORRUPT; } /* must end exactly at bit 0 (RFC 8878 4.2.2; libzstd's checked loops: BIT_endOfDStream) */
    return 0;
}
/* libzstd 1.5 HUF_DecompressFastArgs_init: the fast loops run when the table is indexed by 11 bits (every tree but one of depth 12: shallower
 * trees are rescaled to 11), every stream has at least 8 bytes, and the fourth stream's share starts inside the output (3 * ceil(regen / 4) < regen:
 * all sizes but 6 and 9).  (64-bit little-endian hosts with BMI2, i.e. any x86-64 machine of the last ten years: the reference's platform.) */
static int huf_fast_eligible(const huf_tab* t, size_t l1, size_t l2, size_t l3, size_t l4, uint32_t regen) {
    uint32_t seg = (regen + 3) / 4;
    return g_huf_rule == 1 && t->log <= 11 && l1 >= 8 && l2 >= 8 && l3 >= 8 && l4 >= 8 && 3 * seg < regen;
}

/* ------------------------------------------------------------------ frame context */
typedef struct {
    huf_tab huf; int huf_valid;
    fse_tab ll, of, ml; int fse_valid;
    uint32_t rep[3];
    const uint8_t* dict_content; size_t dict_len;
    uint32_t dict_id;
} dctx;

typedef struct {
    uint8_t* lit;       /* literal buffer, OZS_BLOCK_MAX */
    ozs_seq* seq;       /* OZS_MAX_SEQ */
} scratch;

static int decode_literals(dctx* d, scratch* sc, const uint8_t* src, size_t n, uint32_t block_max,
                           size_t* consumed, uint32_t* nlit, ozs_block_info* bi) {
    CHECK(n >= 1, OZS_E_CORRUPT);
    int type = src[0] & 3, sf = (src[0] >> 2) & 3;
    uint32_t regen, comp = 0; size_t hs; int streams = 0;
    if (type < 2) {
        if (sf == 0 || sf == 2) { hs = 1; regen = src[0] >> 3; }
        else if (sf == 1) { CHECK(n >= 2, OZS_E_CORRUPT); hs = 2; regen = (src[0] >> 4) + ((uint32_t)src[1] << 4); }
        else { CHECK(n >= 3, OZS_E_CORRUPT); hs = 3; regen = (src[0] >> 4) + ((uint32_t)src[1] << 4) + ((uint32_t)src[2] << 12); }
        CHECK(regen <= block_max, OZS_E_CORRUPT);
        if (type == 0) { CHECK(hs + regen <= n, OZS_E_CORRUPT); memcpy(sc->lit, src + hs, regen); *consumed = hs + regen; }
        else { CHECK(hs + 1 <= n, OZS_E_CORRUPT); memset(sc->lit, src[hs], regen); *consumed = hs + 1; }
    } else {
        CHECK(n >= 3, OZS_E_CORRUPT);
        if (type == 3) CHECK(d->huf_valid, OZS_E_DICT); /* libzstd: dictionary_corrupted (ZSTD_decodeLiteralsBlock, litEntropy == 0), before the sizes */
        if (sf == 0 || sf == 1) { hs = 3; uint32_t v = rd24(src); regen = (v >> 4) & 0x3FF; comp = v >> 14; streams = sf ? 4 : 1; }
        else if (sf == 2) { CHECK(n >= 4, OZS_E_CORRUPT); hs = 4; uint32_t v = rd32(src); regen = (v >> 4) & 0x3FFF; comp = v >> 18; streams = 4; }
        else { CHECK(n >= 5, OZS_E_CORRUPT); hs = 5; uint64_t v = (uint64_t)rd32(src) | ((uint64_t)src[4] << 32); regen = (uint32_t)(v >> 4) & 0x3FFFF; comp = (uint32_t)(v >> 22); streams = 4; }
        CHECK(regen <= block_max, OZS_E_CORRUPT);
        CHECK(regen > 0, OZS_E_CORRUPT);
        if (streams == 4) CHECK(regen >= 6, OZS_E_CORRUPT); /* libzstd MIN_LITERALS_FOR_4_STREAMS */
        CHECK(hs + comp <= n, OZS_E_CORRUPT);
        const uint8_t* p = src + hs; size_t rem = comp;
        if (type == 2) {
            int used = huf_read_table(&d->huf, p, rem);
            CHECK(used > 0, OZS_E_CORRUPT);
            d->huf_valid = 1; p += used; rem -= (size_t)used;
        } else {
            CHECK(d->huf_valid, OZS_E_DICT);
        }
        if (streams == 1) {
            int rc = huf_decode_stream(&d->huf, p, rem, sc->lit, regen, 0, 0);
            if (rc) return rc;
        } else {
            CHECK(rem >= 10, OZS_E_CORRUPT);
            size_t l1 = rd16(p), l2 = rd16(p + 2), l3 = rd16(p + 4);
            CHECK(6 + l1 + l2 + l3 <= rem, OZS_E_CORRUPT);
            size_t l4 = rem - 6 - l1 - l2 - l3;
            uint32_t seg = (regen + 3) / 4;
            CHECK(3 * seg <= regen, OZS_E_CORRUPT);
            const uint8_t* s = p + 6; int rc;
            const int fast = huf_fast_eligible(&d->huf, l1, l2, l3, l4, regen);
            if ((rc = huf_decode_stream(&d->huf, s, l1, sc->lit, seg, fast, 6))) return rc;
            if ((rc = huf_decode_stream(&d->huf, s + l1, l2, sc->lit + seg, seg, fast, 6 + l1))) return rc;
            if ((rc = huf_decode_stream(&d->huf, s + l1 + l2, l3, sc->lit + 2 * seg, seg, fast, 6 + l1 + l2))) return rc;
            if ((rc = huf_decode_stream(&d->huf, s + l1 + l2 + l3, l4, sc->lit + 3 * seg, regen - 3 * seg, fast, 6 + l1 + l2 + l3))) return rc;
        }
        *consumed = hs + comp;
    }
    *nlit = regen;
    if (bi) { bi->lit_type = (uint32_t)type; bi->lit_streams = (uint32_t)streams; bi->huf_max_bits = (type >= 2) ? (uint32_t)d->huf.log : 0; }
    return 0;
}

/* builds one of the three sequence tables according to its mode; returns bytes consumed or <0 */
static int seq_table(fse_tab* t, int mode, const uint8_t* p, size_t n, int max_log, int max_sym,
                     const int16_t* def, int def_n, int def_log, int repeat_ok) {
    int nsym, log;
    switch (mode) {
    case 0: { int rc = fse_build(t, def, def_n, def_log); return rc ? rc : 0; }
    case 1: CHECK(n >= 1, OZS_E_CORRUPT); CHECK(p[0] <= max_sym, OZS_E_CORRUPT); fse_rle(t, p[0]); return 1;
    case 2: {
        int16_t norm[256];
        int used = fse_read_ncount(p, n, max_log, max_sym, norm, &nsym, &log);
        CHECK(used > 0, OZS_E_CORRUPT);
        int rc = fse_build(t, norm, nsym, log);
        return rc ? rc : used;
    }
    default: CHECK(repeat_ok, OZS_E_CORRUPT); return 0;
    }
}

/* K4: sequences section -> resolved (ll, ml, off) triples */
/* libzstd decodes and executes sequence after sequence (ZSTD_decompressSequences_body): an execution error of sequence i is
 * reported before anything the bitstream does wrong behind it, and that the bitstream was not consumed exactly is found last.
 * So this function gives no verdict on the bitstream: *valid_out = the sequences decoded before the first field that reaches
 * below the stream's start (from there on libzstd decodes whatever its bit container holds: not a property of the format),
 * *stream_bad = the stream was over-read or not consumed exactly.  execute_sequences() turns that into the verdict. */
static int g_unpinned; /* the last verdict came from "the bitstream was over-read before the block's sequences were executed" */
int ozs_last_verdict_unpinned(void) { return g_unpinned; }
static int g_inexact; /* the last verdict was "the sequence bitstream was not consumed exactly" (every sequence executed): libzstd older than 1.5.4 does not look */
int ozs_last_verdict_inexact(void) { return g_inexact; }
static int decode_sequences(dctx* d, scratch* sc, const uint8_t* src, size_t n, uint32_t* nseq_out, uint32_t* valid_out, int* stream_bad, ozs_block_info* bi) {
    CHECK(n >= 1, OZS_E_CORRUPT);
    const uint8_t* p = src; const uint8_t* end = src + n;
    uint32_t nseq = *p++;
    if (nseq > 0x7F) {
        if (nseq == 0xFF) { CHECK(p + 2 <= end, OZS_E_CORRUPT); nseq = rd16(p) + 0x7F00; p += 2; }
        else { CHECK(p + 1 <= end, OZS_E_CORRUPT); nseq = ((nseq - 0x80) << 8) + *p++; }
    }
    *nseq_out = nseq; *valid_out = nseq; *stream_bad = 0;
    if (nseq == 0) { CHECK(p == end, OZS_E_CORRUPT); return 0; }
    CHECK(nseq <= OZS_MAX_SEQ, OZS_E_CORRUPT);
    CHECK(p + 1 <= end, OZS_E_CORRUPT);
    int modes = *p++;
    CHECK((modes & 3) == 0, OZS_E_CORRUPT);
    int llm = modes >> 6, ofm = (modes >> 4) & 3, mlm = (modes >> 2) & 3;
    if (bi) { bi->ll_mode = (uint32_t)llm; bi->of_mode = (uint32_t)ofm; bi->ml_mode = (uint32_t)mlm; }
    int used;
    used = seq_table(&d->ll, llm, p, (size_t)(end - p), 9, 35, LL_DEF, 36, 6, d->fse_valid); CHECK(used >= 0, OZS_E_CORRUPT); p += used;
    used = seq_table(&d->of, ofm, p, (size_t)(end - p), 8, 31, OF_DEF, 29, 5, d->fse_valid); CHECK(used >= 0, OZS_E_CORRUPT); p += used;
    used = seq_table(&d->ml, mlm, p, (size_t)(end - p), 9, 52, ML_DEF, 53, 6, d->fse_valid); CHECK(used >= 0, OZS_E_CORRUPT); p += used;
    d->fse_valid = 1; /* libzstd sets fseEntropy once a block with sequences is decoded */
#ifdef OZS_SEQ_HOOK /* experiments under tests/native include this file and look at the tables and the bitstream here */
    OZS_SEQ_HOOK(&d->ll, &d->of, &d->ml, p, (size_t)(end - p), nseq);
#endif
    bbr b; CHECK(bbr_init(&b, p, (size_t)(end - p)) == 0, OZS_E_CORRUPT);
    uint32_t sll = bbr_read(&b, d->ll.log), sof = bbr_read(&b, d->of.log), sml = bbr_read(&b, d->ml.log);
    if (b.pos < 0) { *valid_out = 0; *stream_bad = 1; return 0; } /* (shorter than the three initial states) */
    uint32_t rep0 = d->rep[0], rep1 = d->rep[1], rep2 = d->rep[2];
    for (uint32_t i = 0; i < nseq; i++) {
        fse_ent el = d->ll.e[sll], eo = d->of.e[sof], em = d->ml.e[sml];
        CHECK(el.sym <= 35 && em.sym <= 52 && eo.sym <= 31, OZS_E_CORRUPT);
        uint32_t ofv = (1u << eo.sym) + bbr_read(&b, eo.sym);
        uint32_t ml = ML_BASE[em.sym] + bbr_read(&b, ML_BITS[em.sym]);
        uint32_t ll = LL_BASE[el.sym] + bbr_read(&b, LL_BITS[el.sym]);
        uint32_t off;
        if (ofv > 3) { off = ofv - 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
        else {
            uint32_t idx = ofv - 1 + (ll == 0);
            if (idx == 0) off = rep0;
            else if (idx == 1) { off = rep1; rep1 = rep0; rep0 = off; }
            else if (idx == 2) { off = rep2; rep2 = rep1; rep1 = rep0; rep0 = off; }
            else { off = rep0 - 1; if (off == 0) off = 0xFFFFFFFFu; /* libzstd 1.5: "0 is not valid: force offset to -1 => corruption detected at execSequence" */
                   rep2 = rep1; rep1 = rep0; rep0 = off; }
        }
        sc->seq[i].ll = ll; sc->seq[i].ml = ml; sc->seq[i].off = off;
        if (i + 1 < nseq) {
            sll = el.base + bbr_read(&b, el.nb);
            sml = em.base + bbr_read(&b, em.nb);
            sof = eo.base + bbr_read(&b, eo.nb);
        }
        if (b.pos < 0) { *valid_out = i; *stream_bad = 1; return 0; } /* (sequence i itself took bits from below the start) */
    }
    if (b.pos != 0) *stream_bad = 1;
    d->rep[0] = rep0; d->rep[1] = rep1; d->rep[2] = rep2;
    return 0;
}

/* K5: execute.  frame_start..op is this frame's output so far; the dictionary content sits
 * logically just before frame_start. */
static int execute_sequences(const dctx* d, const scratch* sc, uint32_t nlit, uint32_t nseq, uint32_t valid, int stream_bad,
                             uint8_t* frame_start, uint8_t** opp, uint8_t* oend) {
    uint8_t* op = *opp; const uint8_t* lit = sc->lit; uint32_t lpos = 0;
    uint8_t* block_start = op;
    g_unpinned = 0;
    /* The bitstream ran out inside the block: libzstd goes on decoding whatever its bit container holds and rejects the block in the
     * end, with the class its garbage leads to -- not a property of the format.  Rejected here at once (the GPU walker stops there too). */
    if (valid < nseq) { g_unpinned = 1; return OZS_E_CORRUPT; }
    for (uint32_t i = 0; i < valid; i++) {
        uint32_t ll = sc->seq[i].ll, ml = sc->seq[i].ml, off = sc->seq[i].off;
        /* the checks of ZSTD_execSequenceEnd, in its order: room in the destination, literals left, then the offset */
        CHECK((size_t)(oend - op) >= (size_t)ll + ml, OZS_E_DSTSIZE);
        CHECK(ll <= nlit - lpos, OZS_E_CORRUPT);
        CHECK((size_t)(op - block_start) + ll + ml <= OZS_BLOCK_MAX, OZS_E_CORRUPT);
        memcpy(op, lit + lpos, ll); op += ll; lpos += ll;
        size_t have = (size_t)(op - frame_start);
        if (off > have) {
            size_t back = off - have; /* reaches into the dictionary */
            CHECK(back <= d->dict_len, OZS_E_CORRUPT);
            const uint8_t* dp = d->dict_content + d->dict_len - back;
            while (ml && back) { *op++ = *dp++; ml--; back--; }
        }
        for (uint32_t k = 0; k < ml; k++) { op[k] = op[(ptrdiff_t)k - (ptrdiff_t)off]; }
        op += ml;
    }
    if (stream_bad) { g_inexact = 1; return OZS_E_CORRUPT; }    /* not consumed exactly: checked behind the loop (libzstd >= 1.5.4) */
    uint32_t rest = nlit - lpos;
    CHECK((size_t)(oend - op) >= rest, OZS_E_DSTSIZE);
    CHECK((size_t)(op - block_start) + rest <= OZS_BLOCK_MAX, OZS_E_CORRUPT);
    memcpy(op, lit + lpos, rest); op += rest;
    *opp = op;
    return 0;
}

/* ------------------------------------------------------------------ K0: frame header (A.1) */
typedef struct {
    uint64_t window, fcs; int has_fcs, has_checksum, single; uint32_t dict_id; size_t hsize;
} frame_hdr;

static int parse_frame_header(const uint8_t* p, size_t n, frame_hdr* h) {
    CHECK(n >= 5, OZS_E_TRUNCATED);
    int fhd = p[4];
    int fcsf = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    CHECK((fhd & 0x08) == 0, OZS_E_UNSUPPORTED); /* reserved bit */
    static const int did_sz[4] = {0, 1, 2, 4};
    static const int fcs_sz[4] = {0, 2, 4, 8};
    size_t hs = 5 + (single ? 0 : 1) + (size_t)did_sz[did] + (size_t)(fcsf ? fcs_sz[fcsf] : (single ? 1 : 0));
    CHECK(n >= hs, OZS_E_TRUNCATED);
    const uint8_t* q = p + 5;
    uint64_t window = 0;
    if (!single) { int b = *q++; int wl = 10 + (b >> 3); window = (1ULL << wl) + ((1ULL << wl) >> 3) * (uint64_t)(b & 7); }
    uint32_t dict_id = 0;
    if (did == 1) { dict_id = q[0]; q += 1; } else if (did == 2) { dict_id = rd16(q); q += 2; } else if (did == 3) { dict_id = rd32(q); q += 4; }
    h->has_fcs = 1;
    if (fcsf == 0) { if (single) h->fcs = *q++; else { h->fcs = 0; h->has_fcs = 0; } }
    else if (fcsf == 1) { h->fcs = (uint64_t)rd16(q) + 256; q += 2; }
    else if (fcsf == 2) { h->fcs = rd32(q); q += 4; }
    else { h->fcs = rd64(q); q += 8; }
    if (single) window = h->fcs;
    h->window = window; h->single = single; h->has_checksum = (fhd >> 2) & 1; h->dict_id = dict_id; h->hsize = hs;
    return 0;
}

/* dictionary (A.7) */
typedef struct { dctx init; int formatted; } dict_state;

static int load_dict(dict_state* ds, const uint8_t* dict, size_t n) {
    memset(&ds->init, 0, sizeof(ds->init));
    ds->init.rep[0] = 1; ds->init.rep[1] = 4; ds->init.rep[2] = 8;
    ds->formatted = 0;
    if (!dict || n == 0) return 0;
    if (n < 8 || rd32(dict) != 0xEC30A437u) { ds->init.dict_content = dict; ds->init.dict_len = n; return 0; }
    ds->formatted = 1;
    ds->init.dict_id = rd32(dict + 4);
    const uint8_t* p = dict + 8; const uint8_t* end = dict + n;
    int used = huf_read_table(&ds->init.huf, p, (size_t)(end - p)); CHECK(used > 0, OZS_E_DICT); p += used;
    int16_t norm[256]; int nsym, log;
    used = fse_read_ncount(p, (size_t)(end - p), 8, 31, norm, &nsym, &log); CHECK(used > 0, OZS_E_DICT); p += used;
    CHECK(fse_build(&ds->init.of, norm, nsym, log) == 0, OZS_E_DICT);
    used = fse_read_ncount(p, (size_t)(end - p), 9, 52, norm, &nsym, &log); CHECK(used > 0, OZS_E_DICT); p += used;
    CHECK(fse_build(&ds->init.ml, norm, nsym, log) == 0, OZS_E_DICT);
    used = fse_read_ncount(p, (size_t)(end - p), 9, 35, norm, &nsym, &log); CHECK(used > 0, OZS_E_DICT); p += used;
    CHECK(fse_build(&ds->init.ll, norm, nsym, log) == 0, OZS_E_DICT);
    CHECK(p + 12 <= end, OZS_E_DICT);
    size_t content = (size_t)(end - (p + 12));
    for (int i = 0; i < 3; i++) { uint32_t r = rd32(p + 4 * i); CHECK(r != 0 && r <= content, OZS_E_DICT); ds->init.rep[i] = r; }
    p += 12;
    ds->init.dict_content = p; ds->init.dict_len = content;
    ds->init.huf_valid = 1; ds->init.fse_valid = 1;
    return 0;
}

static void trace_block(ozs_trace* tr, const ozs_block_info* bi) {
    if (!tr) return;
    if (tr->blocks && tr->n < tr->cap) tr->blocks[tr->n] = *bi;
    tr->n++;
}

/* one zstd frame starting at src (magic already verified); returns 0 or <0 */
static int decode_frame(const uint8_t* src, size_t n, size_t* consumed, uint8_t* dst, size_t cap, size_t* produced,
                        const dict_state* ds, scratch* sc, ozs_trace* tr) {
    frame_hdr h; int rc = parse_frame_header(src, n, &h);
    if (rc) return rc;
    /* copy_decode is a streaming decoder: libzstd's default ZSTD_d_windowLogMax (27) applies */
    if (h.window > (1ULL << 27) + 1) return OZS_E_UNSUPPORTED;
    /* libzstd: a frame that names a dictionary fails unless exactly that dictionary is loaded */
    if (h.dict_id && h.dict_id != (ds->formatted ? ds->init.dict_id : 0u)) return OZS_E_DICT;
    dctx* d = (dctx*)malloc(sizeof(dctx));
    if (!d) return OZS_E_CORRUPT;
    *d = ds->init;
    uint32_t block_max = (uint32_t)(h.window < OZS_BLOCK_MAX ? h.window : OZS_BLOCK_MAX);
    const uint8_t* ip = src + h.hsize; const uint8_t* iend = src + n;
    uint8_t* op = dst; uint8_t* oend = dst + cap;
    rc = 0;
    for (;;) {
        if (iend - ip < 3) { rc = OZS_E_TRUNCATED; break; }
        uint32_t bh = rd24(ip); ip += 3;
        int last = bh & 1, type = (bh >> 1) & 3; uint32_t bsize = bh >> 3;
        ozs_block_info bi; memset(&bi, 0, sizeof(bi)); bi.block_type = (uint32_t)type;
        if (type == 3) { rc = OZS_E_CORRUPT; break; }
        if (bsize > block_max) { rc = OZS_E_CORRUPT; break; }
        if (type == 0) {
            if ((size_t)(iend - ip) < bsize) { rc = OZS_E_TRUNCATED; break; }
            if ((size_t)(oend - op) < bsize) { rc = OZS_E_DSTSIZE; break; }
            memcpy(op, ip, bsize); op += bsize; ip += bsize; bi.regen = bsize;
        } else if (type == 1) {
            if (iend - ip < 1) { rc = OZS_E_TRUNCATED; break; }
            if ((size_t)(oend - op) < bsize) { rc = OZS_E_DSTSIZE; break; }
            memset(op, *ip, bsize); op += bsize; ip += 1; bi.regen = bsize;
        } else {
            if ((size_t)(iend - ip) < bsize) { rc = OZS_E_TRUNCATED; break; }
            if (bsize < 2) { rc = OZS_E_CORRUPT; break; }
            size_t lit_used; uint32_t nlit = 0, nseq = 0;
            rc = decode_literals(d, sc, ip, bsize, block_max, &lit_used, &nlit, &bi);
            if (rc) break;
            if (lit_used >= bsize) { rc = OZS_E_CORRUPT; break; } /* sequences section needs >= 1 byte */
            uint32_t valid = 0; int stream_bad = 0;
            rc = decode_sequences(d, sc, ip + lit_used, bsize - lit_used, &nseq, &valid, &stream_bad, &bi);
            if (rc) break;
            uint8_t* before = op;
            rc = execute_sequences(d, sc, nlit, nseq, valid, stream_bad, dst, &op, oend);
            if (rc) break;
            bi.n_lit = nlit; bi.n_seq = nseq; bi.regen = (uint32_t)(op - before);
            if (tr) {
                bi.lit_hash = ozs_xxh64(sc->lit, nlit, 0);
                bi.seq_hash = ozs_xxh64((const uint8_t*)sc->seq, (size_t)nseq * sizeof(ozs_seq), 0);
                if (tr->lit_dump) { size_t c = nlit < tr->lit_cap ? nlit : tr->lit_cap; memcpy(tr->lit_dump, sc->lit, c); tr->lit_n = nlit; }
                if (tr->seq_dump) { size_t c = nseq < tr->seq_cap ? nseq : tr->seq_cap; memcpy(tr->seq_dump, sc->seq, c * sizeof(ozs_seq)); tr->seq_n = nseq; }
            }
            ip += bsize;
        }
        trace_block(tr, &bi);
        if (last) break;
    }
    free(d);
    if (rc) return rc;
    size_t out = (size_t)(op - dst);
    if (h.has_fcs && out != h.fcs) return OZS_E_CORRUPT;
    if (h.has_checksum) {
        if (iend - ip < 4) return OZS_E_TRUNCATED;
        uint32_t want = rd32(ip); ip += 4;
        if ((uint32_t)ozs_xxh64(dst, out, 0) != want) return OZS_E_CHECKSUM;
    }
    *consumed = (size_t)(ip - src); *produced = out;
    return 0;
}

int ozs_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len,
               const uint8_t* dict, size_t dict_len, ozs_trace* trace) {
    dict_state* ds = (dict_state*)malloc(sizeof(dict_state));
    scratch sc; sc.lit = (uint8_t*)malloc(OZS_BLOCK_MAX + 32); sc.seq = (ozs_seq*)malloc(sizeof(ozs_seq) * (OZS_MAX_SEQ + 1));
    int rc = OZS_E_CORRUPT;
    size_t pos = 0, out = 0;
    g_unpinned = 0; g_inexact = 0; g_lit_inexact = 0; g_lit_lenient = 0; g_lit_through = 0; g_lit_over = 0; /* (per call: a decode that fails before any sequence is executed must not inherit the previous call's flags) */
    if (trace) trace->n = 0;
    if (!ds || !sc.lit || !sc.seq) goto done;
    rc = load_dict(ds, dict, dict_len);
    if (rc) goto done;
    while (pos < n) {
        if (n - pos < 4) { rc = OZS_E_TRUNCATED; goto done; }
        uint32_t magic = rd32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (n - pos < 8) { rc = OZS_E_TRUNCATED; goto done; }
            uint64_t sz = rd32(src + pos + 4);
            if ((uint64_t)(n - pos - 8) < sz) { rc = OZS_E_TRUNCATED; goto done; }
            pos += 8 + (size_t)sz;
            continue;
        }
        if (magic != 0xFD2FB528u) { rc = OZS_E_BADMAGIC; goto done; }
        size_t used = 0, made = 0;
        rc = decode_frame(src + pos, n - pos, &used, dst + out, cap - out, &made, ds, &sc, trace);
        if (rc) goto done;
        pos += used; out += made;
    }
    rc = 0;
done:
    if (out_len) *out_len = out;
    free(ds); free(sc.lit); free(sc.seq);
    return rc;
}

uint64_t ozs_content_size(const uint8_t* src, size_t n) {
    size_t pos = 0; uint64_t total = 0; int unknown = 0;
    while (pos < n) {
        if (n - pos < 4) return UINT64_MAX - 1;
        uint32_t magic = rd32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (n - pos < 8) return UINT64_MAX - 1;
            uint64_t sz = rd32(src + pos + 4);
            if ((uint64_t)(n - pos - 8) < sz) return UINT64_MAX - 1;
            pos += 8 + (size_t)sz; continue;
        }
        if (magic != 0xFD2FB528u) return UINT64_MAX - 1;
        frame_hdr h; if (parse_frame_header(src + pos, n - pos, &h)) return UINT64_MAX - 1;
        if (!h.has_fcs) unknown = 1; else total += h.fcs;
        size_t p = pos + h.hsize;
        for (;;) { /* walk the block chain to the frame's end */
            if (n - p < 3) return UINT64_MAX - 1;
            uint32_t bh = rd24(src + p); p += 3;
            int type = (bh >> 1) & 3; uint32_t bs = bh >> 3;
            if (type == 3) return UINT64_MAX - 1;
            size_t adv = (type == 1) ? 1 : bs;
            if (n - p < adv) return UINT64_MAX - 1;
            p += adv;
            if (bh & 1) break;
        }
        if (h.has_checksum) { if (n - p < 4) return UINT64_MAX - 1; p += 4; }
        pos = p;
    }
    return unknown ? UINT64_MAX : total;
}

const char* ozs_strerror(int code) {
    switch (code) {
    case OZS_OK: return "ok";
    case OZS_E_CORRUPT: return "corrupt input";
    case OZS_E_TRUNCATED: return "truncated input";
    case OZS_E_CHECKSUM: return "content checksum mismatch";
    case OZS_E_DSTSIZE: return "destination too small";
    case OZS_E_UNSUPPORTED: return "unsupported frame parameter";
    case OZS_E_BADMAGIC: return "unknown frame magic";
    case OZS_E_DICT: return "dictionary missing, wrong or corrupt";
    default: return "unknown error";
    }
}
