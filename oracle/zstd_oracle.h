/*
 * oracle/zstd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the zstd frame decoder that the reference reaches
 * through `zstd::stream::copy_decode` (reference src/main.rs:463-467; the arithmetic
 * lives in the un-vendored dependency libzstd 1.5.6 = zstd-sys 2.0.13+zstd.1.5.6,
 * reference Cargo.lock:2371-2396).  Restated from the published format (RFC 8878)
 * as summarised in SURVEY.md Appendix A.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product (libmzd.so) never links or calls it.
 *
 * Parity pin: checked against (a) the frames the reference's own tests hold
 * (tests/convert.rs:15-43 `bulk::compress(b"...",0)`, tests/cmdline.rs:19-29 payloads
 * written through the reference writer settings src/main.rs:781-791), and
 * (b) outputs of the system libzstd run in the build container
 * (tests/golden/make_golden.py, committed with the vectors it made).
 */
#ifndef ZSTD_ORACLE_H
#define ZSTD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error classes -- same numbering as include/mzd.h */
#define OZS_OK 0
#define OZS_E_CORRUPT (-1)
#define OZS_E_TRUNCATED (-2)
#define OZS_E_CHECKSUM (-3)
#define OZS_E_DSTSIZE (-4)
#define OZS_E_UNSUPPORTED (-5)
#define OZS_E_BADMAGIC (-7)
#define OZS_E_DICT (-8)

#define OZS_BLOCK_MAX (128u * 1024u)
#define OZS_MAX_SEQ 43691u

typedef struct {
    uint32_t ll, ml, off; /* off = resolved back-distance (after repeat-offset logic) */
} ozs_seq;

/* Per-block intermediates (the "CPU twin" of each GPU phase). */
typedef struct {
    uint32_t block_type;  /* 0 raw 1 rle 2 compressed */
    uint32_t lit_type;    /* 0 raw 1 rle 2 huf 3 treeless (compressed blocks only) */
    uint32_t lit_streams; /* 1 or 4 for huf/treeless, 0 otherwise */
    uint32_t huf_max_bits;
    uint32_t n_lit;
    uint32_t n_seq;
    uint32_t ll_mode, of_mode, ml_mode; /* 0 predef 1 rle 2 fse 3 repeat */
    uint32_t regen;       /* bytes this block appended */
    uint64_t lit_hash;    /* XXH64(seed 0) over the literal buffer */
    uint64_t seq_hash;    /* XXH64 over the ozs_seq array */
} ozs_block_info;

typedef struct {
    ozs_block_info* blocks; /* caller array or NULL */
    size_t cap;             /* capacity of blocks[] */
    size_t n;               /* filled */
    /* optional raw dumps of the LAST compressed block decoded */
    uint8_t* lit_dump; size_t lit_cap; size_t lit_n;
    ozs_seq* seq_dump; size_t seq_cap; size_t seq_n;
} ozs_trace;

/* Whole-file decode: every concatenated frame, skippable frames skipped, checksum
 * verified when present (semantics of copy_decode: reference src/main.rs:463).
 * dict may be NULL.  Returns OZS_OK or a negative class; *out_len = bytes produced. */
int ozs_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len,
               const uint8_t* dict, size_t dict_len, ozs_trace* trace);

/* Sum of Frame_Content_Size over all frames; UINT64_MAX if any frame omits it;
 * UINT64_MAX-1 on a malformed header. */
uint64_t ozs_content_size(const uint8_t* src, size_t n);

uint64_t ozs_xxh64(const uint8_t* p, size_t n, uint64_t seed);

const char* ozs_strerror(int code);

/* 1 when the last ozs_decode() rejected its input because a sequence bitstream ran out before the block's sequences were
 * executed: libzstd rejects such input too, but with whatever class the garbage it then decodes leads to -- not a property of
 * the format, so tests that pin error CLASSES against ZSTD_getErrorCode skip those inputs.  Not thread-safe (test hook). */
int ozs_last_verdict_unpinned(void);
/* 1 when the last ozs_decode() rejected its input because a block's sequence bitstream was not consumed exactly (all of its sequences
 * executed without an error): libzstd older than 1.5.4 does not check that and reports whatever comes next. */
int ozs_last_verdict_inexact(void);
/* 1 when the last ozs_decode() rejected its input because a Huffman literal stream was not consumed exactly (RFC 8878 4.2.2 calls
 * that corrupt, and so does libzstd 1.4; libzstd 1.5 decodes on and leaves the garbage to the content checksum, if there is one). */
int ozs_last_verdict_lit_inexact(void);
/* The rule for literal streams that are not consumed exactly: 1 (default) libzstd 1.5.x's, the reference's pin -- the fast loops of
 * eligible four-stream sections decode their symbols and ignore what is left; 0: RFC 8878 / libzstd 1.4.x, always exact.  Test hook. */
void ozs_set_huf_rule(int rule);
/* 1 when the last ozs_decode() ACCEPTED a literal stream that was not consumed exactly (rule 1): libzstd 1.4.x refuses that input. */
int ozs_last_verdict_lit_lenient(void);
/* 1 when the last ozs_decode() ACCEPTED a literal stream that ran out and read on into the bytes in front of it (rule 1): libzstd 1.5.x
 * does the same unless the overrun is deep when its loop ends (zstd_oracle.c). */
int ozs_last_verdict_lit_through(void);
/* 1 when the last ozs_decode() rejected its input because a stream of a fast-loop section needed bits from below the section's first
 * byte (rule 1): libzstd 1.5.x decodes on from a bit container it no longer refills. */
int ozs_last_verdict_lit_over(void);

#ifdef __cplusplus
}
#endif
#endif
