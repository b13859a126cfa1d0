/*
 * corpus/corpus_gen.c -- deterministic synthetic corpora for tests and bench.py.
 *
 * Not part of the decode path: it produces the INPUTS (what the reference's writer would have
 * left in data_dir).  Files are compressed with the libzstd already on the machine (dlopen),
 * with the reference writer's settings -- level, pledged source size (=> Frame_Content_Size),
 * content checksum on (reference src/main.rs:781-791).  Compression stays on the CPU in the
 * reference too; only decode is rebuilt for the GPU.
 *
 * PRNG: splitmix64, seed = config_id * 1000003 + file_index.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t x; } rng;
static inline uint64_t rnext(rng* r) {
    uint64_t z = (r->x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint32_t rbelow(rng* r, uint32_t n) { return (uint32_t)(rnext(r) % n); }

static const char* WORDS[24] = {
    "alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet", "kilo", "lima",
    "mike", "november", "oscar", "papa", "quebec", "romeo", "sierra", "tango", "uniform", "victor", "whiskey", "xray"};

typedef struct { uint8_t* p; size_t n, cap; } sink;
static inline void put(sink* s, const char* t, size_t len) {
    size_t room = s->cap - s->n;
    if (len > room) len = room;
    memcpy(s->p + s->n, t, len); s->n += len;
}
static inline void puts_(sink* s, const char* t) { put(s, t, strlen(t)); }
static void putu(sink* s, uint64_t v) { char b[24]; int n = snprintf(b, sizeof b, "%llu", (unsigned long long)v); put(s, b, (size_t)n); }

/* kind 0: JSON lines */
static void gen_json(rng* r, sink* s, uint64_t first_id) {
    uint64_t id = first_id;
    while (s->n < s->cap) {
        char b[64];
        puts_(s, "{\"id\":"); putu(s, id++);
        puts_(s, ",\"uuid\":\"");
        snprintf(b, sizeof b, "%016llx%016llx", (unsigned long long)rnext(r), (unsigned long long)rnext(r)); put(s, b, 32);
        puts_(s, "\",\"ts\":"); putu(s, 1700000000ULL + rnext(r) % 10000000ULL);
        puts_(s, ",\"user\":{\"name\":\""); puts_(s, WORDS[rbelow(r, 24)]); puts_(s, "_"); puts_(s, WORDS[rbelow(r, 24)]);
        puts_(s, "\",\"age\":"); putu(s, 18 + rbelow(r, 72));
        uint32_t sc = rbelow(r, 1000000);
        int n = snprintf(b, sizeof b, ",\"score\":%u.%03u", sc / 1000, sc % 1000); put(s, b, (size_t)n);
        puts_(s, ",\"active\":"); puts_(s, (rnext(r) & 1) ? "true" : "false");
        puts_(s, "},\"tags\":[");
        uint32_t nt = 1 + rbelow(r, 5);
        for (uint32_t i = 0; i < nt; i++) { if (i) puts_(s, ","); puts_(s, "\""); puts_(s, WORDS[rbelow(r, 24)]); puts_(s, "\""); }
        puts_(s, "],\"msg\":\"");
        uint32_t nw = 3 + rbelow(r, 17);
        for (uint32_t i = 0; i < nw; i++) { if (i) puts_(s, " "); puts_(s, WORDS[rbelow(r, 24)]); }
        puts_(s, "\"}\n");
    }
}

/* kind 1: word-skewed English-like text */
static void gen_text(rng* r, sink* s) {
    static const char* SYL[32] = {"an", "ber", "con", "de", "er", "fi", "gra", "ha", "in", "jo", "ki", "lo", "men", "no", "or", "pre",
                                  "qu", "re", "st", "ti", "un", "ver", "wi", "ex", "ly", "za", "th", "ing", "ed", "al", "ou", "ch"};
    char vocab[512][16];
    for (int i = 0; i < 512; i++) {
        int ns = 1 + (int)rbelow(r, 3); vocab[i][0] = 0;
        for (int k = 0; k < ns; k++) strcat(vocab[i], SYL[rbelow(r, 32)]);
    }
    int col = 0;
    while (s->n < s->cap) {
        /* zipf-ish: index = 512 * u^3 */
        double u = (double)(rnext(r) >> 11) / 9007199254740992.0;
        int idx = (int)(512.0 * u * u * u);
        const char* w = vocab[idx & 511];
        puts_(s, w); col += (int)strlen(w) + 1;
        uint32_t q = rbelow(r, 40);
        if (q == 0) puts_(s, ". "); else if (q == 1) puts_(s, ", "); else puts_(s, " ");
        if (col > 72) { puts_(s, "\n"); col = 0; }
    }
}

/* kind 2: XML-like markup */
static void gen_markup(rng* r, sink* s) {
    uint64_t id = rnext(r) % 100000;
    puts_(s, "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<records>\n");
    while (s->n < s->cap) {
        puts_(s, "  <record id=\""); putu(s, id++); puts_(s, "\" kind=\""); puts_(s, WORDS[rbelow(r, 6)]); puts_(s, "\">\n");
        uint32_t nf = 2 + rbelow(r, 5);
        for (uint32_t i = 0; i < nf; i++) {
            const char* tag = WORDS[6 + rbelow(r, 8)];
            puts_(s, "    <"); puts_(s, tag); puts_(s, ">");
            if (rnext(r) & 1) putu(s, rnext(r) % 100000); else { puts_(s, WORDS[rbelow(r, 24)]); puts_(s, " "); puts_(s, WORDS[rbelow(r, 24)]); }
            puts_(s, "</"); puts_(s, tag); puts_(s, ">\n");
        }
        puts_(s, "  </record>\n");
    }
}

/* kind 3: little-endian int32 columns with small deltas */
static void gen_int32(rng* r, sink* s) {
    int32_t v[4] = {1000, -500000, 77, 1 << 20};
    while (s->n < s->cap) {
        for (int c = 0; c < 4; c++) {
            v[c] += (int32_t)rbelow(r, 17) - 8 + (c == 1 ? 3 : 0);
            uint8_t b[4] = {(uint8_t)v[c], (uint8_t)(v[c] >> 8), (uint8_t)(v[c] >> 16), (uint8_t)(v[c] >> 24)};
            put(s, (const char*)b, 4);
        }
    }
}

/* kind 4: DNA-like 4-symbol text with occasional repeats */
static void gen_dna(rng* r, sink* s) {
    static const char A[4] = {'A', 'C', 'G', 'T'};
    while (s->n < s->cap) {
        if (s->n > 4096 && rbelow(r, 64) == 0) { /* copy an earlier stretch */
            size_t len = 16 + rbelow(r, 200), back = 1 + rbelow(r, 4000);
            for (size_t i = 0; i < len && s->n < s->cap; i++) { s->p[s->n] = s->p[s->n - back]; s->n++; }
        } else {
            uint64_t x = rnext(r);
            for (int i = 0; i < 32 && s->n < s->cap; i++) { s->p[s->n++] = (uint8_t)A[(x >> (2 * i)) & 3]; }
        }
        if ((s->n % 61) == 60 && s->n < s->cap) s->p[s->n++] = '\n';
    }
}

/* kind 5: x-ray-like noisy 12-bit samples in 16-bit LE words */
static void gen_xray(rng* r, sink* s) {
    uint32_t base = 2048;
    while (s->n < s->cap) {
        base = (base + rbelow(r, 33) - 16) & 0xFFF;
        uint32_t v = (base + rbelow(r, 64)) & 0xFFF;
        uint8_t b[2] = {(uint8_t)v, (uint8_t)(v >> 8)};
        put(s, (const char*)b, 2);
    }
}

/* kind 6: incompressible */
static void gen_random(rng* r, sink* s) {
    while (s->n < s->cap) { uint64_t x = rnext(r); put(s, (const char*)&x, 8); }
}

/* kind 7: long repeats / zeros (RLE blocks, giant overlapping matches) */
static void gen_repeats(rng* r, sink* s) {
    while (s->n < s->cap) {
        uint32_t mode = rbelow(r, 4);
        size_t len = 64 + rbelow(r, 60000);
        if (mode == 0) { size_t room = s->cap - s->n; if (len > room) len = room; memset(s->p + s->n, 0, len); s->n += len; }
        else if (mode == 1) { uint8_t c = (uint8_t)rnext(r); size_t room = s->cap - s->n; if (len > room) len = room; memset(s->p + s->n, c, len); s->n += len; }
        else {
            char pat[40]; uint32_t pl = 2 + rbelow(r, 30);
            for (uint32_t i = 0; i < pl; i++) pat[i] = (char)('a' + rbelow(r, 26));
            for (size_t i = 0; i < len && s->n < s->cap; i++) s->p[s->n++] = (uint8_t)pat[i % pl];
        }
    }
}

/* Fills out[0..n) for `kind`, deterministically from (cfg_id, index). */
void corpus_gen(int kind, uint64_t cfg_id, uint64_t index, uint8_t* out, size_t n) {
    rng r = {cfg_id * 1000003ULL + index};
    sink s = {out, 0, n};
    switch (kind) {
    case 0: gen_json(&r, &s, index * 1000ULL); break;
    case 1: gen_text(&r, &s); break;
    case 2: gen_markup(&r, &s); break;
    case 3: gen_int32(&r, &s); break;
    case 4: gen_dna(&r, &s); break;
    case 5: gen_xray(&r, &s); break;
    case 6: gen_random(&r, &s); break;
    case 7: gen_repeats(&r, &s); break;
    default: memset(out, 0, n); break;
    }
}

/* ------------------------------------------------------------------ libzstd (compress side only) */
static void* H;
static unsigned (*p_isError)(size_t);
static size_t (*p_compressBound)(size_t);
static void* (*p_createCCtx)(void);
static size_t (*p_freeCCtx)(void*);
static size_t (*p_CCtx_setParameter)(void*, int, int);
static size_t (*p_compress2)(void*, void*, size_t, const void*, size_t);
static const char* (*p_versionString)(void);
static size_t (*p_CCtx_loadDictionary)(void*, const void*, size_t); /* optional (config 5) */
static size_t (*p_trainFromBuffer)(void*, size_t, const void*, const size_t*, unsigned); /* optional */
#define SYM(v, name) do { *(void**)(&v) = dlsym(H, name); if (!v) return -2; } while (0)

int corpus_open_zstd(const char* path) {
    if (H) return 0;
    const char* cands[] = {path, "libzstd.so.1", "/lib/x86_64-linux-gnu/libzstd.so.1", "/usr/lib/x86_64-linux-gnu/libzstd.so.1",
                           "/opt/conda/lib/libzstd.so.1", "libzstd.so"};
    for (unsigned i = 0; i < sizeof(cands) / sizeof(cands[0]) && !H; i++)
        if (cands[i] && cands[i][0]) H = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
    if (!H) return -1;
    SYM(p_isError, "ZSTD_isError"); SYM(p_compressBound, "ZSTD_compressBound"); SYM(p_createCCtx, "ZSTD_createCCtx");
    SYM(p_freeCCtx, "ZSTD_freeCCtx"); SYM(p_CCtx_setParameter, "ZSTD_CCtx_setParameter"); SYM(p_compress2, "ZSTD_compress2");
    SYM(p_versionString, "ZSTD_versionString");
    *(void**)(&p_CCtx_loadDictionary) = dlsym(H, "ZSTD_CCtx_loadDictionary");
    *(void**)(&p_trainFromBuffer) = dlsym(H, "ZDICT_trainFromBuffer");
    return 0;
}
const char* corpus_zstd_version(void) { return H ? p_versionString() : ""; }
size_t corpus_bound(size_t n) { return H ? p_compressBound(n) : 0; }

typedef struct {
    int kind_base, kind_mod; uint64_t cfg_id; uint64_t first_index; uint64_t stride; uint32_t nfiles;
    const uint64_t* raw_offs; const uint64_t* raw_sizes; uint8_t* raw;
    uint8_t* comp; const uint64_t* comp_offs; uint64_t* comp_sizes;
    int level, checksum; volatile uint32_t* next; int fail;
    const uint8_t* dict; size_t dict_len; /* config 5: every file is compressed with this dictionary (or NULL) */
} build_arg;

static void* build_worker(void* v) {
    build_arg* a = (build_arg*)v;
    void* c = p_createCCtx();
    if (a->dict && a->dict_len) {
        p_CCtx_setParameter(c, 100, a->level); /* the dictionary is digested for this level; it stays loaded across compress2 calls */
        if (!p_CCtx_loadDictionary || p_isError(p_CCtx_loadDictionary(c, a->dict, a->dict_len))) { a->fail = 1; p_freeCCtx(c); return NULL; }
    }
    for (;;) {
        uint32_t i = __sync_fetch_and_add(a->next, 1);
        if (i >= a->nfiles) break;
        uint64_t idx = a->first_index + (uint64_t)i * a->stride;
        int kind = a->kind_base + (a->kind_mod ? (int)(idx % (uint64_t)a->kind_mod) : 0);
        uint8_t* raw = a->raw + a->raw_offs[i];
        if (a->kind_base >= 0) corpus_gen(kind, a->cfg_id, idx, raw, (size_t)a->raw_sizes[i]); /* (< 0: the caller has filled `raw` -- real files) */
        p_CCtx_setParameter(c, 100, a->level);
        p_CCtx_setParameter(c, 201, a->checksum);
        size_t cap = p_compressBound((size_t)a->raw_sizes[i]);
        size_t r = p_compress2(c, a->comp + a->comp_offs[i], cap, raw, (size_t)a->raw_sizes[i]);
        if (p_isError(r)) { a->fail = 1; break; }
        a->comp_sizes[i] = r;
    }
    p_freeCCtx(c);
    return NULL;
}

/* Generates and compresses files first_index + i*stride, i in [0, nfiles) (stride = the number of
 * GPUs when files are dealt round-robin: rank r takes first_index = r).  File i is of kind
 * kind_base + (index % kind_mod) (kind_mod 0 => always kind_base).  raw/comp are caller
 * buffers laid out by raw_offs / comp_offs (comp slots must hold corpus_bound(raw_size)).
 * Each file is ONE frame written like the reference writer: level, checksum, pledged size. */
int corpus_build_dict(int kind_base, int kind_mod, uint64_t cfg_id, uint64_t first_index, uint64_t stride, uint32_t nfiles,
                      const uint64_t* raw_offs, const uint64_t* raw_sizes, uint8_t* raw,
                      uint8_t* comp, const uint64_t* comp_offs, uint64_t* comp_sizes,
                      int level, int checksum, int nthreads, const uint8_t* dict, size_t dict_len);
int corpus_build(int kind_base, int kind_mod, uint64_t cfg_id, uint64_t first_index, uint64_t stride, uint32_t nfiles,
                 const uint64_t* raw_offs, const uint64_t* raw_sizes, uint8_t* raw,
                 uint8_t* comp, const uint64_t* comp_offs, uint64_t* comp_sizes,
                 int level, int checksum, int nthreads) {
    return corpus_build_dict(kind_base, kind_mod, cfg_id, first_index, stride, nfiles, raw_offs, raw_sizes, raw, comp, comp_offs, comp_sizes,
                             level, checksum, nthreads, NULL, 0);
}

/* ZDICT_trainFromBuffer over `n` samples laid out back to back; returns the dictionary size or < 0. */
long corpus_train_dict(uint8_t* dict, size_t cap, const uint8_t* samples, const size_t* sizes, unsigned n) {
    if (!H || !p_trainFromBuffer) return -1;
    size_t r = p_trainFromBuffer(dict, cap, samples, sizes, n);
    return p_isError(r) ? -2 : (long)r;
}

int corpus_build_dict(int kind_base, int kind_mod, uint64_t cfg_id, uint64_t first_index, uint64_t stride, uint32_t nfiles,
                      const uint64_t* raw_offs, const uint64_t* raw_sizes, uint8_t* raw,
                      uint8_t* comp, const uint64_t* comp_offs, uint64_t* comp_sizes,
                      int level, int checksum, int nthreads, const uint8_t* dict, size_t dict_len) {
    if (!H) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    pthread_t th[64]; build_arg args[64]; volatile uint32_t next = 0;
    for (int t = 0; t < nthreads; t++) {
        build_arg a = {kind_base, kind_mod, cfg_id, first_index, stride ? stride : 1, nfiles, raw_offs, raw_sizes, raw, comp, comp_offs, comp_sizes,
                       level, checksum, &next, 0, dict, dict_len};
        args[t] = a;
        pthread_create(&th[t], NULL, build_worker, &args[t]);
    }
    int fail = 0;
    for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); fail |= args[t].fail; }
    return fail ? -3 : 0;
}
