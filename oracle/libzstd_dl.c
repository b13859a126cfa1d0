/*
 * oracle/libzstd_dl.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * dlopen() binding to the libzstd shared object already on the machine, with
 * hand-declared prototypes of the stable ABI (no headers needed).  libzstd is the
 * un-vendored dependency that does the reference's arithmetic (reference Cargo.toml:34,
 * Cargo.lock:2371-2396; call sites src/main.rs:463 decode, :781-791 encode), so it
 *   (1) pins the C restatement in zstd_oracle.c (tests + tests/golden/make_golden.py), and
 *   (2) is the "reference" CPU baseline bench.py times on the GPU box's host cores:
 *       zref_time_stream8k()  = B1, the shape copy_decode + io::copy produce (8 KiB chunks)
 *       zref_time_oneshot_mt() = B2, one ZSTD_decompressDCtx per file on a thread pool.
 * Decompressed output is fully determined by the format, so any libzstd version pins
 * decode parity; the version actually loaded is reported by zref_version().
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { const void* src; size_t size; size_t pos; } z_in;
typedef struct { void* dst; size_t size; size_t pos; } z_out;

static void* H;
static const char* (*p_versionString)(void);
static unsigned (*p_isError)(size_t);
static const char* (*p_getErrorName)(size_t);
static size_t (*p_compressBound)(size_t);
static size_t (*p_compress)(void*, size_t, const void*, size_t, int);
static size_t (*p_decompress)(void*, size_t, const void*, size_t);
static void* (*p_createCCtx)(void);
static size_t (*p_freeCCtx)(void*);
static size_t (*p_CCtx_setParameter)(void*, int, int);
static size_t (*p_CCtx_loadDictionary)(void*, const void*, size_t);
static size_t (*p_compress2)(void*, void*, size_t, const void*, size_t);
static void* (*p_createDCtx)(void);
static size_t (*p_freeDCtx)(void*);
static size_t (*p_decompressDCtx)(void*, void*, size_t, const void*, size_t);
static size_t (*p_decompress_usingDict)(void*, void*, size_t, const void*, size_t, const void*, size_t);
static size_t (*p_decompressStream)(void*, z_out*, z_in*);
static size_t (*p_initDStream)(void*);
static size_t (*p_compressStream2)(void*, z_out*, z_in*, int);
static size_t (*p_trainFromBuffer)(void*, size_t, const void*, const size_t*, unsigned);
static unsigned long long (*p_getFrameContentSize)(const void*, size_t);
static void* (*p_createDDict)(const void*, size_t);
static size_t (*p_freeDDict)(void*);
static size_t (*p_decompress_usingDDict)(void*, void*, size_t, const void*, size_t, const void*);

#define SYM(v, name) do { *(void**)(&v) = dlsym(H, name); if (!v) return -2; } while (0)

int zref_open(const char* path) {
    if (H) return 0;
    const char* cands[] = {path, "libzstd.so.1", "/lib/x86_64-linux-gnu/libzstd.so.1",
                           "/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1", "libzstd.so"};
    for (unsigned i = 0; i < sizeof(cands) / sizeof(cands[0]) && !H; i++)
        if (cands[i] && cands[i][0]) H = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
    if (!H) return -1;
    SYM(p_versionString, "ZSTD_versionString"); SYM(p_isError, "ZSTD_isError"); SYM(p_getErrorName, "ZSTD_getErrorName");
    SYM(p_compressBound, "ZSTD_compressBound"); SYM(p_compress, "ZSTD_compress"); SYM(p_decompress, "ZSTD_decompress");
    SYM(p_createCCtx, "ZSTD_createCCtx"); SYM(p_freeCCtx, "ZSTD_freeCCtx"); SYM(p_CCtx_setParameter, "ZSTD_CCtx_setParameter");
    SYM(p_CCtx_loadDictionary, "ZSTD_CCtx_loadDictionary"); SYM(p_compress2, "ZSTD_compress2");
    SYM(p_createDCtx, "ZSTD_createDCtx"); SYM(p_freeDCtx, "ZSTD_freeDCtx"); SYM(p_decompressDCtx, "ZSTD_decompressDCtx");
    SYM(p_decompress_usingDict, "ZSTD_decompress_usingDict"); SYM(p_decompressStream, "ZSTD_decompressStream");
    SYM(p_initDStream, "ZSTD_initDStream"); SYM(p_compressStream2, "ZSTD_compressStream2");
    SYM(p_trainFromBuffer, "ZDICT_trainFromBuffer"); SYM(p_getFrameContentSize, "ZSTD_getFrameContentSize");
    SYM(p_createDDict, "ZSTD_createDDict"); SYM(p_freeDDict, "ZSTD_freeDDict"); SYM(p_decompress_usingDDict, "ZSTD_decompress_usingDDict");
    return 0;
}

const char* zref_version(void) { return H ? p_versionString() : ""; }
size_t zref_bound(size_t n) { return p_compressBound(n); }
const char* zref_error_name(long code) { return p_getErrorName((size_t)code); }

/* zstd::bulk::compress(data, level) -- what the reference's tests feed (tests/convert.rs:15-43) */
long zref_compress_simple(void* dst, size_t cap, const void* src, size_t n, int level) {
    size_t r = p_compress(dst, cap, src, n, level);
    return p_isError(r) ? -1 : (long)r;
}

/* The reference writer's settings (src/main.rs:781-791): level, pledged size (=> FCS), checksum.
 * flags: bit0 checksum, bit1 omit content size (ZSTD_c_contentSizeFlag=200 -> 0).
 * window_log > 0 sets ZSTD_c_windowLog (101).  dict may be NULL. */
long zref_compress(void* dst, size_t cap, const void* src, size_t n, int level, int flags, int window_log,
                   const void* dict, size_t dict_len) {
    void* c = p_createCCtx();
    if (!c) return -1;
    p_CCtx_setParameter(c, 100, level);
    p_CCtx_setParameter(c, 201, flags & 1);
    if (flags & 2) p_CCtx_setParameter(c, 200, 0);
    if (window_log > 0) p_CCtx_setParameter(c, 101, window_log);
    if (dict && dict_len) p_CCtx_loadDictionary(c, dict, dict_len);
    size_t r = p_compress2(c, dst, cap, src, n);
    p_freeCCtx(c);
    return p_isError(r) ? -1 : (long)r;
}

/* streaming compress without a pledged size => frame without FCS, input fed in `chunk` pieces
 * with ZSTD_e_flush(1) between them when flush_each != 0 (=> several blocks). */
long zref_compress_stream(void* dst, size_t cap, const void* src, size_t n, int level, int checksum, size_t chunk, int flush_each) {
    void* c = p_createCCtx();
    if (!c) return -1;
    p_CCtx_setParameter(c, 100, level);
    p_CCtx_setParameter(c, 201, checksum);
    z_out o = {dst, cap, 0};
    size_t pos = 0; long rc = 0;
    if (chunk == 0) chunk = n ? n : 1;
    while (pos < n) {
        size_t take = n - pos < chunk ? n - pos : chunk;
        z_in in = {(const char*)src + pos, take, 0};
        while (in.pos < in.size) {
            size_t r = p_compressStream2(c, &o, &in, 0);
            if (p_isError(r)) { rc = -1; goto out; }
        }
        if (flush_each) { z_in e = {NULL, 0, 0}; size_t r; do { r = p_compressStream2(c, &o, &e, 1); if (p_isError(r)) { rc = -1; goto out; } } while (r); }
        pos += take;
    }
    { z_in e = {NULL, 0, 0}; size_t r; do { r = p_compressStream2(c, &o, &e, 2); if (p_isError(r)) { rc = -1; goto out; } } while (r); }
    rc = (long)o.pos;
out:
    p_freeCCtx(c);
    return rc;
}

long zref_decompress(void* dst, size_t cap, const void* src, size_t n) {
    size_t r = p_decompress(dst, cap, src, n);
    return p_isError(r) ? -(long)(0 - r) : (long)r; /* negative libzstd error code */
}

long zref_decompress_dict(void* dst, size_t cap, const void* src, size_t n, const void* dict, size_t dict_len) {
    void* d = p_createDCtx();
    size_t r = p_decompress_usingDict(d, dst, cap, src, n, dict, dict_len);
    p_freeDCtx(d);
    return p_isError(r) ? -(long)(0 - r) : (long)r;
}

/* copy_decode shape: a streaming decoder drained through an 8 KiB buffer until EOF
 * (reference src/main.rs:463-467).  Returns bytes produced or a negative value. */
static long stream8k(void* ds, uint8_t* dst, size_t cap, const void* src, size_t n) {
    uint8_t buf[8192];
    p_initDStream(ds);
    z_in in = {src, n, 0};
    size_t total = 0, last = 0;
    while (in.pos < in.size || last != 0) {
        z_out o = {buf, sizeof(buf), 0};
        size_t before = in.pos;
        size_t r = p_decompressStream(ds, &o, &in);
        if (p_isError(r)) return -(long)(0 - r);
        if (total + o.pos > cap) return -70; /* dstSize_tooSmall */
        memcpy(dst + total, buf, o.pos);
        total += o.pos; last = r;
        if (in.pos == in.size && o.pos < sizeof(buf)) {
            if (r != 0) return -72; /* srcSize_wrong: EOF inside a frame */
            break;
        }
        if (in.pos == before && o.pos == 0) return -1; /* no progress */
    }
    return (long)total;
}

long zref_decompress_stream8k(void* dst, size_t cap, const void* src, size_t n) {
    void* ds = p_createDCtx();
    long r = stream8k(ds, (uint8_t*)dst, cap, src, n);
    p_freeDCtx(ds);
    return r;
}

unsigned long long zref_frame_content_size(const void* src, size_t n) { return p_getFrameContentSize(src, n); }

long zref_train_dict(void* dict, size_t cap, const void* samples, const size_t* sizes, unsigned nsamples) {
    size_t r = p_trainFromBuffer(dict, cap, samples, sizes, nsamples);
    return p_isError(r) ? -1 : (long)r;
}

/* ------------------------------------------------------------------ timed baselines */
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* B1: one thread, streaming, 8 KiB chunks; files [0, nfiles).  Returns seconds; *bytes = output bytes. */
double zref_time_stream8k(const uint8_t* blob, const uint64_t* offs, const uint64_t* sizes, unsigned nfiles,
                          uint8_t* scratch, size_t scratch_cap, uint64_t* bytes) {
    void* ds = p_createDCtx();
    uint64_t total = 0;
    double t0 = now_s();
    for (unsigned i = 0; i < nfiles; i++) {
        long r = stream8k(ds, scratch, scratch_cap, blob + offs[i], (size_t)sizes[i]);
        if (r < 0) { total = 0; break; }
        total += (uint64_t)r;
    }
    double t1 = now_s();
    p_freeDCtx(ds);
    *bytes = total;
    return t1 - t0;
}

typedef struct {
    const uint8_t* blob; const uint64_t* offs; const uint64_t* sizes; unsigned nfiles, ntasks;
    uint8_t* out; const uint64_t* out_offs; const uint64_t* out_caps;
    volatile unsigned* next; uint64_t bytes; int fail;
    const void* ddict; /* config 5: a digested dictionary shared by all workers (NULL: none) */
} mt_arg;

static void* mt_worker(void* v) {
    mt_arg* a = (mt_arg*)v;
    void* d = p_createDCtx();
    /* pass 0 decodes into the shared out[] (each file exactly once: used for verification); later passes
     * decode into a private buffer -- libzstd uses dst as its window, so two threads must never share one */
    uint64_t maxcap = 0;
    for (unsigned i = 0; i < a->nfiles; i++) if (a->out_caps[i] > maxcap) maxcap = a->out_caps[i];
    uint8_t* priv = (uint8_t*)malloc((size_t)maxcap + 64);
    for (;;) {
        unsigned t = __sync_fetch_and_add(a->next, 1);
        if (t >= a->ntasks) break;
        unsigned i = t % a->nfiles; /* pass number = t / nfiles */
        uint8_t* dst = t < a->nfiles ? a->out + a->out_offs[i] : priv;
        size_t r = a->ddict ? p_decompress_usingDDict(d, dst, (size_t)a->out_caps[i], a->blob + a->offs[i], (size_t)a->sizes[i], a->ddict)
                            : p_decompressDCtx(d, dst, (size_t)a->out_caps[i], a->blob + a->offs[i], (size_t)a->sizes[i]);
        if (p_isError(r)) { a->fail = 1; break; }
        a->bytes += r;
    }
    free(priv);
    p_freeDCtx(d);
    return NULL;
}

/* B2: nthreads workers created ONCE, `passes` passes over the files (one file per task, reused DCtx per
 * thread, one-shot decode into out[]).  Returns seconds; *bytes = decompressed bytes of all passes. */
static double time_oneshot_mt(const uint8_t* blob, const uint64_t* offs, const uint64_t* sizes, unsigned nfiles,
                              uint8_t* out, const uint64_t* out_offs, const uint64_t* out_caps, unsigned nthreads, unsigned passes,
                              uint64_t* bytes, const void* ddict) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 512) nthreads = 512;
    if (passes < 1) passes = 1;
    pthread_t th[512]; static mt_arg args[512];
    volatile unsigned next = 0;
    double t0 = now_s();
    for (unsigned t = 0; t < nthreads; t++) {
        mt_arg a = {blob, offs, sizes, nfiles, nfiles * passes, out, out_offs, out_caps, &next, 0, 0, ddict};
        args[t] = a;
        pthread_create(&th[t], NULL, mt_worker, &args[t]);
    }
    uint64_t total = 0; int fail = 0;
    for (unsigned t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); total += args[t].bytes; fail |= args[t].fail; }
    double t1 = now_s();
    *bytes = fail ? 0 : total;
    return t1 - t0;
}
double zref_time_oneshot_mt(const uint8_t* blob, const uint64_t* offs, const uint64_t* sizes, unsigned nfiles,
                            uint8_t* out, const uint64_t* out_offs, const uint64_t* out_caps, unsigned nthreads, unsigned passes,
                            uint64_t* bytes) {
    return time_oneshot_mt(blob, offs, sizes, nfiles, out, out_offs, out_caps, nthreads, passes, bytes, NULL);
}
/* the same with one dictionary for every file (config 5), digested once outside the timed region (ZSTD_createDDict) */
double zref_time_oneshot_mt_dict(const uint8_t* blob, const uint64_t* offs, const uint64_t* sizes, unsigned nfiles,
                                 uint8_t* out, const uint64_t* out_offs, const uint64_t* out_caps, unsigned nthreads, unsigned passes,
                                 uint64_t* bytes, const void* dict, size_t dict_len) {
    void* dd = p_createDDict(dict, dict_len);
    if (!dd) { *bytes = 0; return 0.0; }
    double t = time_oneshot_mt(blob, offs, sizes, nfiles, out, out_offs, out_caps, nthreads, passes, bytes, dd);
    p_freeDDict(dd);
    return t;
}
