// mzd_kernels.hip -- the zstd frame decoder as HIP kernels for gfx950 (MI355X / CDNA4).
//
// Replaces the arithmetic behind `zstd::stream::copy_decode` (reference src/main.rs:463-467;
// libzstd 1.5.6 via zstd-sys, reference Cargo.lock:2371-2396), written from the format
// (RFC 8878) for 64-lane wavefronts.  Not a port of libzstd.
//
// Mapping: one workgroup (4 wavefronts, ~34 KB of LDS, 4 workgroups per CU) decodes one block at a time in a
// persistent grid.  Two drivers share the block pipeline:
//   mzd_decode_kernel_files   a workgroup owns a file and walks its frames and blocks in order (launches in which
//                             no file can have more than one block);
//   mzd_decode_kernel_tasks   one block per task: the blocks of a frame run on different workgroups, the state
//                             between them (tables, repeat offsets, output position, checksum) is handed over in
//                             task order.
// Per compressed block, one role per wavefront, connected by LDS flags and unbounded HBM queues:
//   K0  headers                 lane 0, from LDS copies; the sequence header inside wavefront 0's role   (A.1, A.2)
//   K1  Huffman tree            lane 0 decodes the weights, the wavefront validates, ranks and fills      (A.4)
//   K2  Huffman literals        64 lanes per stream by self-synchronising sub-stream decode (memoized entry
//                               offsets) + DPP scan for the output offsets; streams handed out by a queue
//   K3  FSE tables x3           one wavefront, ballot-rank symbol spread                                  (A.3)
//   K4a FSE state walk          wavefront 0: the serial chain, tables + bitstream ring in LDS, 16-byte records (A.5)
//   K4b plan                    wavefront 3: fields, symbolic repeat offsets (DPP scan), positions, validation
//   K5  sequence execute        wavefront 1: runs of <= 64 sequences staged in LDS, prefetched HBM sources,
//                               LDS->LDS matches in rounds, 16-byte coalesced flushes                     (A.5)
//   K6  raw / RLE blocks        256 lanes, coalesced
//   K7  XXH64                   wavefront 2 behind the copier, groups of 8 stripes                         (A.6)
// Everything is integer/byte work bound by latency and instruction issue, so there is no MFMA here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <type_traits>

#include <rocprim/warp/warp_scan.hpp>

#include "../../include/mzd.h"
#include "mzd_device.h"
#include "mzd_tables.h"

#ifndef MZD_PRIO_WALK
#define MZD_PRIO_WALK 3
#endif
#ifndef MZD_PRIO_COPY
#define MZD_PRIO_COPY 2
#endif
#ifndef MZD_PRIO_PLAN
#define MZD_PRIO_PLAN 1
#endif

namespace mzd {

#if defined(MZD_STAMPS) || defined(MZD_TFIN)
#define TFIN(k) do { if (lane == 0) S.tfin[k] = __builtin_readcyclecounter() - S.tstart; } while (0)
#define TSTART() do { if (tid == 0) { S.tstart = __builtin_readcyclecounter(); S.tfin[10] = S.tstart - S.ttask; } } while (0)
#define TTASK() do { if (tid == 0) S.ttask = __builtin_readcyclecounter(); } while (0)
#define TTASK_END() do { if (tid == 0) S.tfin[11] = __builtin_readcyclecounter() - S.ttask; } while (0)
#define TCOUNT(k, v) do { if (lane == 0) atomicAdd((unsigned long long*)&S.tfin[k], (unsigned long long)(v)); } while (0)
#define TFIN_FLUSH() do { if (tid == 0 && a.debug) { for (int k_ = 0; k_ < 12; k_++) a.debug[a.wg0 + blockIdx.x].tfin[k_] = S.tfin[k_]; } } while (0)
#else
#define TFIN(k)
#define TCOUNT(k, v)
#define TTASK()
#define TTASK_END()
#define TSTART()
#define TFIN_FLUSH()
#endif

// Diagnostic build only: per-phase cycle sums of the workgroup (lane 0), never in the product .so.
#ifdef MZD_STAMPS
#define STAMP_DECL uint64_t st_prev = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k) do { uint64_t t_ = __builtin_readcyclecounter(); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#define STAMP_FLUSH() do { if (tid == 0 && a.debug) { for (int k_ = 0; k_ < 8; k_++) { a.debug[a.wg0 + blockIdx.x].stamp[k_] = st_acc[k_]; a.debug[a.wg0 + blockIdx.x].cstamp[k_] = S.cdiag[k_]; } a.debug[a.wg0 + blockIdx.x].stamp[2] = S.c.diag_slow; } } while (0)
#define CSTAMP_DECL uint64_t cs_prev = __builtin_readcyclecounter()
#define CSTAMP(k) do { uint64_t t_ = __builtin_readcyclecounter(); if (lane == 0) S.cdiag[k] += t_ - cs_prev; cs_prev = t_; } while (0)
#if 0
#endif
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH()
#define CSTAMP_DECL
#define CSTAMP(k)
#endif


// ------------------------------------------------------------------------------------ LDS
constexpr int kRingBytes = 8192; // sequence-bitstream ring: 8 chunks of 1 KiB (+16 mirrored bytes)
constexpr int kChunk = 1024;
constexpr int kRingChunks = kRingBytes / kChunk;

struct Ctl {
    uint64_t pos;        // next unread input byte of the file
    uint64_t out;        // bytes produced for this file
    uint64_t frame_out0; // `out` at the start of the current frame
    uint64_t fcs;
    uint64_t lit_off;    // file offset of the raw literals / first Huffman stream
    uint64_t seq_off;    // file offset of the sequence bitstream
    int32_t err;
    uint32_t action;     // 0 frame, 1 skip, 2 done
    uint32_t job;
    uint32_t has_fcs, has_cksum, block_max;
    uint32_t btype, bsize, last;
    uint32_t lit_type, nlit, streams, huf_log, huf_valid, huf_nw;
    uint32_t s_off[4], s_len[4], s_out[4], s_n[4];
    uint32_t lit_is_raw;
    uint32_t huf_tree_off, huf_tree_len;           // Huffman tree description inside the block
    uint32_t huf_ready, huf_fill, lit_done, walk_prog; // intra-workgroup flags of the block pipeline
    uint32_t next_stream, streams_done, streams_mask; // Huffman streams are handed out to whichever wavefront is free; mask: bit k = stream k decoded
    uint32_t tables_ready, plan_prog, copy_prog, plan_lit_used; // walker -> planner -> copier
    uint32_t plan_too_long; // 1: the plan ends with a chunk that cannot be executed (literals run out / output passes 128 KiB); 2: only the literals after
                            // the last sequence pass 128 KiB.  No error yet: the copier, which reports in stream order, gives the verdict
    uint32_t seq_parsed;                           // the sequence header is parsed: nseq, seq_off, seq_len, modes are final
    uint32_t exec_done;                            // the copying wavefront has finished the block
    uint64_t exec_pos;                             // output bytes complete and visible (published by the executor)
    uint32_t diag_slow;                            // diagnostic build: walker iterations that needed a lower window
    uint32_t nseq, mode[3], al[3], nsym[3], fse_valid, seq_len;
    uint32_t rep[3];
    uint32_t rep_op[4];                             // the block's repeat-offset transform (start slots -> end slots): s, v0, v1, v2
    uint32_t dict_content_len;
    const uint8_t* dict_content;
    // the task (one block of one file) and what its predecessor published
    uint32_t lds_dict_fse, lds_dict_huf;            // driver 1: dictionary (handle) whose FSE / Huffman tables sit unmodified in LDS, or 0
    uint32_t t_valid, task, in_frame, with_dict;
    uint32_t pred_ready;                            // the predecessor's state is in pred_* (LDS flag of the block pipeline)
    int32_t pred_err;
    uint32_t pred_rep[3];
    uint32_t tables_published;
    uint64_t pred_out, pred_frame_out0, pred_xstripes, pred_xxh[4];
};

struct __attribute__((aligned(16))) Shared {
    uint8_t ring[kRingBytes + 16]; // first: at LDS offset 0 the walker's window address needs no base add
    // FSE decode entries, 8 bytes: low dword = byte offset of the next state's entry before the
    // fresh bits are added (8 * nextStateBase); high dword = nbBits | (extra+nbBits) << 8 | symbol << 16 | extra << 24
    uint64_t ll[512];
    uint64_t ml[512];
    uint64_t of[256];
    uint8_t stage[3 * (2048 + 16)]; // K5 staging: the run being assembled and the two before it (kStage each)
    uint8_t hseg2[2048 + 64];       // Huffman stream segment of wavefront 2 (it still decodes while the copier already uses `stage`)
    uint32_t ll_base[36], ml_base[53]; // code -> base value (copied once from constant memory)
#ifdef MZD_STAMPS
    uint64_t cdiag[8];
#endif
#if defined(MZD_STAMPS) || defined(MZD_TFIN)
    uint64_t ttask, tstart, tfin[12]; // block start; finish of walker / copier / hasher / planner; literals ready; tables ready
#endif
    uint16_t huf[2048]; // sym | len << 8
    int16_t norm[3][64];
    uint16_t next[3][64];
    alignas(16) int16_t wnorm[256]; // FSE table of the Huffman weights.  wnorm + wtab + weights (1 KiB, contiguous) double as the
                                    // copier's literal scratch (kLitScratch): the copying wavefront is the one that decodes the weights, earlier
    uint32_t wtab[64];  // sym | nb << 8 | base << 16
    uint8_t weights[256];
    Ctl c;
    // driver 1, files of one block: while the copier and the hasher finish file A, the idle walking wavefront takes the
    // next file and parses its headers into `c2` (header bytes staged in a free part of the ring)
    Ctl c2;
    uint32_t pre_job, pre_valid; // the job taken ahead (kNoJob: none) and whether c2 holds its parsed first block
    // driver 1: the small fields of the dictionary the workgroup used last (config 5: every file names the same one --
    // reading them from HBM again for each file costs a round trip per dependent load)
    struct { uint32_t id, formatted, al[3], huf_log, rep[3], content_len; const uint8_t* content; } dcache;
    struct { const uint8_t* src; uint64_t n; uint8_t* dst; uint64_t cap; uint32_t dict; } pj; // the job table entry of pre_job (read once, by pre_parse_next)
};

// The workgroup's LDS image.  File scope, so that every device function addresses it with DS
// instructions and immediate offsets (a `Shared&` parameter would be a flat pointer).
__shared__ Shared S;

__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld24(const uint8_t* p) { return ld16(p) | ((uint32_t)p[2] << 16); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return ld16(p) | (ld16(p + 2) << 16); }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { return (uint64_t)ld32(p) | ((uint64_t)ld32(p + 4) << 32); }
__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ldu64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ int hibit(uint32_t v) { return 31 - __builtin_clz(v); }

__device__ __forceinline__ void wg_fence() {
    // make this wave's global stores visible to later loads of the same workgroup (same CU, same L1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// bits [bitpos, bitpos+n) of the little-endian integer p[0..nbytes); indices < 0 and >= 8*nbytes
// read as 0.  n <= 32.  Lane-0 parsing helper (the input is readable MZD_SRC_PADDING past its end).
__device__ __noinline__ uint32_t bits_at(const uint8_t* p, uint32_t nbytes, int32_t bitpos, int n) {
    if (n == 0) return 0;
    if (bitpos < 0) {
        int neg = -bitpos;
        if (neg >= n) return 0;
        return bits_at(p, nbytes, 0, n - neg) << neg;
    }
    uint32_t byte = (uint32_t)bitpos >> 3;
    if (byte >= nbytes) return 0;
    uint64_t v = ldu64(p + byte);
    uint32_t avail = nbytes - byte;
    if (avail < 8) v &= (1ull << (avail * 8)) - 1;
    v >>= (bitpos & 7);
    return (uint32_t)(v & ((1ull << n) - 1));
}

// Intra-workgroup flags in LDS (the block pipeline): relaxed atomics + workgroup fences.  Every spin
// also ends when an error is posted, and is bounded.
__device__ __forceinline__ uint32_t flag_load(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void flag_store(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
// first error wins: a wavefront that merely gave up because another one failed must not overwrite the cause
__device__ __forceinline__ void post_err(int32_t* err, int rc) {
    if (rc) { int32_t expected = 0; __atomic_compare_exchange_n(err, &expected, rc, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); }
}
// (a wait that runs out is a failure of the launch -- a co-tenant starved the workgroup, a role died -- not of the input:
//  it posts MZD_E_DEVICE, and whatever the waiting role reports afterwards loses to it)
__device__ __forceinline__ bool spin_ge(const uint32_t* p, uint32_t want, int32_t* err) {
    for (uint32_t it = 0; it < (1u << 24); it++) {
        if (flag_load(p) >= want) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return true; }
        if (__atomic_load_n(err, __ATOMIC_RELAXED)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    post_err(err, MZD_E_DEVICE);
    return false;
}

// ------------------------------------------------------------------------------------ K3
// Normalized-count header (A.3), forward bitstream.  Lane 0.  Returns bytes used or < 0.
// LD(byte) returns the 8 bytes at `byte` of the header (readable past its end); the variants differ only in
// where the header lives: HBM (dictionary, Huffman weights) or the LDS staging area (sequence headers).
template <class LD>
__device__ __forceinline__ int read_ncount_t(LD ld, uint32_t n, int max_log, int max_sym, int16_t* norm, uint32_t* nsym_out, uint32_t* log_out) {
    if (n < 1) return MZD_E_CORRUPT;
    const int32_t limit = (int32_t)(n > 4096 ? 4096 : n) * 8;
    // bits [bit, bit+nb) of the header, zero past its end; nb <= 16.  The header is read upwards a few bits at a
    // time: a 64-bit register window, refilled every ~6 symbols (a lone lane pays ~60 cycles per LDS/HBM read).
    uint64_t win = 0; int32_t wbase = 0, wtop = 0; // window = header bits [wbase, wtop)
    auto take = [&](int32_t bit, int nb) -> int {
        if (bit < wbase || bit + nb > wtop) {
            const uint32_t byte = (uint32_t)bit >> 3;
            wbase = (int32_t)(byte * 8); wtop = wbase + 64;
            win = 0;
            if (byte < n) {
                win = ld(byte);
                const uint32_t avail = n - byte;
                if (avail < 8) win &= (1ull << (avail * 8)) - 1;
            }
        }
        return (int)((win >> (bit - wbase)) & ((1u << nb) - 1));
    };
    int32_t bit = 0;
    int al = 5 + take(bit, 4);
    bit += 4;
    if (al > max_log) return MZD_E_CORRUPT;
    int remaining = 1 << al, sym = 0;
    while (remaining > 0 && sym <= max_sym) {
        int nb = hibit((uint32_t)(remaining + 1)) + 1;
        if (bit >= limit) return MZD_E_CORRUPT;
        int val = take(bit, nb);
        bit += nb;
        int lower = (1 << (nb - 1)) - 1;
        int thr = (1 << nb) - 1 - (remaining + 1);
        if ((val & lower) < thr) { bit -= 1; val &= lower; }
        else if (val > lower) val -= thr;
        int p = val - 1;
        remaining -= (p < 0) ? 1 : p;
        if (remaining < 0) return MZD_E_CORRUPT;
        norm[sym++] = (int16_t)p;
        if (p == 0) {
            for (;;) {
                if (bit >= limit) return MZD_E_CORRUPT;
                int r = take(bit, 2);
                bit += 2;
                for (int i = 0; i < r; i++) { if (sym > max_sym) return MZD_E_CORRUPT; norm[sym++] = 0; }
                if (r != 3) break;
            }
        }
    }
    if (remaining != 0 || sym > max_sym + 1 || bit > limit) return MZD_E_CORRUPT;
    *nsym_out = (uint32_t)sym;
    *log_out = (uint32_t)al;
    return (bit + 7) >> 3;
}
struct HbmBytes { const uint8_t* p; __device__ __forceinline__ uint64_t operator()(uint32_t o) const { return ldu64(p + o); } };
struct StageBytes { // offset into S.stage
    uint32_t base;
    __device__ __forceinline__ uint64_t operator()(uint32_t o) const { uint64_t v; __builtin_memcpy(&v, &S.stage[base + o], 8); return v; }
};
__device__ __noinline__ int read_ncount(const uint8_t* src, uint32_t n, int max_log, int max_sym, int16_t* norm, uint32_t* nsym_out, uint32_t* log_out) {
    return read_ncount_t(HbmBytes{src}, n, max_log, max_sym, norm, nsym_out, log_out);
}
__device__ __noinline__ int read_ncount_staged(uint32_t stage_off, uint32_t n, int max_log, int max_sym, int16_t* norm, uint32_t* nsym_out, uint32_t* log_out) {
    return read_ncount_t(StageBytes{stage_off}, n, max_log, max_sym, norm, nsym_out, log_out);
}

// number of extra bits of a code: kind 0 LL, 1 OF, 2 ML
__device__ __forceinline__ uint32_t code_extra(uint32_t s, int kind) { return kind == 0 ? LL_BITS[s] : (kind == 1 ? s : ML_BITS[s]); }
__device__ __forceinline__ uint64_t pack_entry(uint32_t nbase, uint32_t nb, uint32_t s, int kind) {
    uint32_t extra = code_extra(s, kind);
    uint32_t hi = nb | ((extra + nb) << 8) | (s << 16) | (extra << 24);
    return (uint64_t)(nbase * 8u) | ((uint64_t)hi << 32);
}

// Table build (A.3) by one lane.  kind 0 LL, 1 OF, 2 ML.
__device__ __noinline__ int build_seq_table(uint64_t* tab, const int16_t* norm, uint16_t* next, uint32_t nsym, uint32_t log, int kind) {
    uint32_t size = 1u << log, high = size;
    for (uint32_t s = 0; s < nsym; s++)
        if (norm[s] == -1) { high--; tab[high] = s; next[s] = 1; }
    uint32_t step = (size >> 1) + (size >> 3) + 3, pos = 0, mask = size - 1;
    for (uint32_t s = 0; s < nsym; s++) {
        int c = norm[s];
        if (c <= 0) continue;
        next[s] = (uint16_t)c;
        for (int i = 0; i < c; i++) {
            tab[pos] = s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return MZD_E_CORRUPT;
    for (uint32_t i = 0; i < size; i++) {
        uint32_t s = (uint32_t)tab[i];
        uint32_t d = next[s]++;
        uint32_t nb = log - (uint32_t)hibit(d);
        uint32_t nbase = (d << nb) - size;
        tab[i] = pack_entry(nbase, nb, s, kind);
    }
    return 0;
}

__device__ void rle_seq_table(uint64_t* tab, uint32_t s, int kind) { tab[0] = pack_entry(0, 0, s, kind); }

// ------------------------------------------------------------------------------------ K1
// Huffman tree description (A.4) -> S.weights[0..nw), S.c.huf_log.  Lane 0.  Returns bytes used or < 0.
template <class LD>
__device__ __forceinline__ int read_huf_weights_t(LD ld, uint32_t n, uint16_t* next) { // `next`: 512 bytes of scratch for the weights' FSE table build
    if (n < 1) return MZD_E_CORRUPT;
    auto byte_at = [&](uint32_t o) -> uint32_t { return (uint32_t)(ld(o) & 0xFF); };
    // bits [bitpos, bitpos+nb) of the little-endian integer made of bytes [base, base+len); indices < 0 read as 0; nb <= 16
    auto take = [&](uint32_t base, uint32_t len, int32_t bitpos, int nb) -> uint32_t {
        if (nb == 0) return 0u;
        int32_t neg = 0;
        if (bitpos < 0) { neg = -bitpos; if (neg >= nb) return 0u; nb -= neg; bitpos = 0; }
        uint32_t byte = (uint32_t)bitpos >> 3;
        if (byte >= len) return 0u;
        uint64_t v = ld(base + byte);
        uint32_t avail = len - byte;
        if (avail < 8) v &= (1ull << (avail * 8)) - 1;
        return (uint32_t)((v >> (bitpos & 7)) & ((1u << nb) - 1)) << neg;
    };
    uint32_t hb = byte_at(0), nw = 0;
    int used;
    uint8_t* w = S.weights;
    if (hb >= 128) {
        nw = hb - 127;
        uint32_t bytes = (nw + 1) / 2;
        if (1 + bytes > n) return MZD_E_CORRUPT;
        for (uint32_t i = 0; i < nw; i++) {
            uint32_t b = byte_at(1 + i / 2);
            w[i] = (uint8_t)((i & 1) ? (b & 15) : (b >> 4));
        }
        used = 1 + (int)bytes;
    } else {
        if (hb < 1 || 1 + hb > n) return MZD_E_CORRUPT;
        uint32_t nsym, log;
        struct Shift { LD ld; uint32_t o; __device__ __forceinline__ uint64_t operator()(uint32_t k) const { return ld(o + k); } };
        int hdr = read_ncount_t(Shift{ld, 1}, hb, 6, 255, S.wnorm, &nsym, &log);
        if (hdr <= 0) return MZD_E_CORRUPT;
        // tiny FSE table (<= 64 entries) built in place
        uint32_t size = 1u << log, high = size;
        for (uint32_t s = 0; s < nsym; s++)
            if (S.wnorm[s] == -1) { high--; S.wtab[high] = s; next[s] = 1; }
        uint32_t step = (size >> 1) + (size >> 3) + 3, pos = 0, mask = size - 1;
        for (uint32_t s = 0; s < nsym; s++) {
            int c = S.wnorm[s];
            if (c <= 0) continue;
            next[s] = (uint16_t)c;
            for (int i = 0; i < c; i++) {
                S.wtab[pos] = s;
                do { pos = (pos + step) & mask; } while (pos >= high);
            }
        }
        if (pos != 0) return MZD_E_CORRUPT;
        for (uint32_t i = 0; i < size; i++) {
            uint32_t s = S.wtab[i], d = next[s]++;
            uint32_t nb = log - (uint32_t)hibit(d);
            S.wtab[i] = s | (nb << 8) | (((d << nb) - size) << 16);
        }
        if ((uint32_t)hdr >= hb) return MZD_E_CORRUPT;
        const uint32_t bs = 1 + (uint32_t)hdr; // offset of the weight bitstream
        const uint32_t bl = hb - (uint32_t)hdr;
        const uint32_t lastb = byte_at(bs + bl - 1);
        if (lastb == 0) return MZD_E_CORRUPT;
        int32_t bpos = (int32_t)(bl - 1) * 8 + hibit(lastb);
        // the stream is read downwards a few bits at a time: a 64-bit register window, refilled every ~10 symbols
        uint64_t win = 0; int32_t wbase = 0, wtop = 0; // window = stream bits [wbase, wtop)
        auto bits = [&](int32_t bp, int nb) -> uint32_t { // stream bits [bp, bp + nb), nb <= 6; bits below 0 read as 0
            if (nb == 0) return 0u;
            if (bp < 0) return take(bs, bl, bp, nb);
            if (bp < wbase || bp + nb > wtop) {
                int32_t lo = bp + 16 - 64; if (lo < 0) lo = 0; // top of the window >= bp + 9 > bp + nb
                wbase = lo & ~7; wtop = wbase + 64;
                const uint32_t byte = (uint32_t)wbase >> 3; // < bl
                win = ld(bs + byte);
                const uint32_t avail = bl - byte;
                if (avail < 8) win &= (1ull << (avail * 8)) - 1;
            }
            return (uint32_t)(win >> (bp - wbase)) & ((1u << nb) - 1);
        };
        bpos -= (int32_t)log; uint32_t s1 = bits(bpos, (int)log);
        bpos -= (int32_t)log; uint32_t s2 = bits(bpos, (int)log);
        int ok = 0;
        // Two interleaved states.  While a pair of weights cannot exhaust the stream (<= 6 bits each), both entries and the
        // 8 stream bytes below the read point are read together: two weights per LDS round trip, no branch on their values.
        while (bpos >= 12 && nw <= 252) {
            const uint32_t e1 = S.wtab[s1], e2 = S.wtab[s2];
            int32_t bi = (bpos - 56) >> 3;
            bi = bi < 0 ? 0 : bi;
            const uint64_t W = ld(bs + (uint32_t)bi); // stream bits [8 bi, 8 bi + 64): the read point lies 12..63 bits up
            const uint32_t nb1 = (e1 >> 8) & 0xFF, nb2 = (e2 >> 8) & 0xFF;
            const uint32_t h = (uint32_t)bpos - (uint32_t)bi * 8;
            const uint32_t both = (uint32_t)(W >> (h - nb1 - nb2)); // state 1's fresh bits above state 2's
            const uint16_t two = (uint16_t)((e1 & 0xFF) | ((e2 & 0xFF) << 8));
            __builtin_memcpy(w + nw, &two, 2); // (nw is even here)
            nw += 2;
            s1 = (e1 >> 16) + ((both >> nb2) & ((1u << nb1) - 1));
            s2 = (e2 >> 16) + (both & ((1u << nb2) - 1));
            bpos -= (int32_t)(nb1 + nb2);
        }
        for (;;) { // the tail, a weight at a time; ends when the stream is over-read
            if (nw > 253) break;
            uint32_t e = S.wtab[s1];
            w[nw++] = (uint8_t)e; int nb = (e >> 8) & 0xFF; bpos -= nb; s1 = (e >> 16) + bits(bpos, nb);
            if (bpos < 0) { w[nw++] = (uint8_t)S.wtab[s2]; ok = 1; break; }
            if (nw > 253) break;
            e = S.wtab[s2];
            w[nw++] = (uint8_t)e; nb = (e >> 8) & 0xFF; bpos -= nb; s2 = (e >> 16) + bits(bpos, nb);
            if (bpos < 0) { w[nw++] = (uint8_t)S.wtab[s1]; ok = 1; break; }
        }
        if (!ok) return MZD_E_CORRUPT;
        used = 1 + (int)hb;
    }
    S.c.huf_nw = nw; // the implied last weight, the validation and the table come from finish_huf_table_wave
    return used;
}
// (scratch: the copier's staging buffer is idle until the literals exist; [256, 512) holds the sequence header, [1024, 1161) the tree)
__device__ __noinline__ int read_huf_weights(const uint8_t* src, uint32_t n) { return read_huf_weights_t(HbmBytes{src}, n, (uint16_t*)(void*)(S.stage + 1536)); }               // dictionary (HBM)
struct RingBytes { // offset into S.ring
    uint32_t base;
    __device__ __forceinline__ uint64_t operator()(uint32_t o) const { uint64_t v; __builtin_memcpy(&v, &S.ring[base + o], 8); return v; }
};
__device__ __noinline__ int read_ncount_ring(uint32_t ring_off, uint32_t n, int max_log, int max_sym, int16_t* norm, uint32_t* nsym_out, uint32_t* log_out) {
    return read_ncount_t(RingBytes{ring_off}, n, max_log, max_sym, norm, nsym_out, log_out);
}
__device__ __noinline__ int read_huf_weights_staged(uint32_t stage_off, uint32_t n) { return read_huf_weights_t(StageBytes{stage_off}, n, (uint16_t*)(void*)(S.stage + 1536)); } // a block's tree, staged in LDS


// ------------------------------------------------------------------------------------ K2
// wave-wide inclusive scans on the DPP path (row_shr / row_bcast: no LDS traffic, no ds_bpermute latency)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    (void)lane;
    using WS = rocprim::warp_scan<uint32_t, 64>;
    WS::storage_type* st = nullptr; // the DPP implementation keeps no state in LDS
    uint32_t r;
    WS().inclusive_scan(v, r, *st, rocprim::plus<uint32_t>());
    return r;
}

// Huffman decode table from the explicit weights S.weights[0 .. huf_nw) (A.4), by one wavefront; lane l owns
// symbols l, l+64, l+128, l+192.  Validates the weights, derives the implied last one, and fills the canonical
// table: weight 1 (longest codes) first, equal weights in symbol order -- positions come from ballots, not from
// per-symbol counting loops.  Sets S.c.huf_log.  Returns 0 or MZD_E_CORRUPT.
__device__ __noinline__ int finish_huf_table_wave(int lane) {
    uint32_t nw = (uint32_t)__builtin_amdgcn_readfirstlane(S.c.huf_nw);
    if (nw < 1 || nw > 255) return MZD_E_CORRUPT;
    uint32_t w[4], tot = 0;
    bool bad = false;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t sym = (uint32_t)g * 64 + (uint32_t)lane;
        w[g] = sym < nw ? S.weights[sym] : 0u;
        if (w[g] > 12) { bad = true; w[g] = 0; }
        tot += w[g] ? 1u << (w[g] - 1) : 0u;
    }
    if (__any(bad)) return MZD_E_CORRUPT;
    const uint32_t total = __builtin_amdgcn_readlane(wave_incl_scan(tot, lane), 63);
    if (total == 0) return MZD_E_CORRUPT;
    const uint32_t maxbits = (uint32_t)hibit(total) + 1;
    if (maxbits > 11) return MZD_E_CORRUPT;
    const uint32_t left = (1u << maxbits) - total;
    if (left & (left - 1)) return MZD_E_CORRUPT;
    const uint32_t wl = (uint32_t)hibit(left) + 1;
#pragma unroll
    for (int g = 0; g < 4; g++) if ((uint32_t)g * 64 + (uint32_t)lane == nw) w[g] = wl; // the implied last symbol
    nw++;
    const uint64_t below = (1ull << lane) - 1;
    uint32_t at[4] = {0, 0, 0, 0}, p2 = 0;
    for (uint32_t r = 1; r <= maxbits; r++) {
        uint32_t cnt_r = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint64_t m = __ballot(w[g] == r);
            if (w[g] == r) at[g] = p2 + ((cnt_r + (uint32_t)__builtin_popcountll(m & below)) << (r - 1));
            cnt_r += (uint32_t)__builtin_popcountll(m);
        }
        if (r == 1 && (cnt_r < 2 || (cnt_r & 1))) return MZD_E_CORRUPT;
        p2 += cnt_r << (r - 1);
    }
    if (p2 != (1u << maxbits)) return MZD_E_CORRUPT; // also catches weights above maxbits
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t sym = (uint32_t)g * 64 + (uint32_t)lane;
        const uint32_t cnt = w[g] ? 1u << (w[g] - 1) : 0u;
        const uint32_t e = sym | ((maxbits + 1 - w[g]) << 8);
        if (cnt == 1) S.huf[at[g]] = (uint16_t)e;
        else if (cnt && cnt < 64) { // aligned to cnt (>= 2): pairs
            uint32_t* q = reinterpret_cast<uint32_t*>(&S.huf[at[g]]);
            for (uint32_t i = 0; i < cnt / 2; i++) q[i] = e | (e << 16);
        }
        uint64_t big = __ballot(cnt >= 64); // few symbols own most of the table: all 64 lanes fill those together
        while (big) {
            const int src = __builtin_ctzll(big);
            const uint32_t a0 = __builtin_amdgcn_readlane(at[g], src), c0 = __builtin_amdgcn_readlane(cnt, src), e0 = __builtin_amdgcn_readlane(e, src);
            uint32_t* q = reinterpret_cast<uint32_t*>(&S.huf[a0]);
            for (uint32_t i = (uint32_t)lane; i < c0 / 2; i += 64) q[i] = e0 | (e0 << 16);
            big &= big - 1;
        }
    }
    if (lane == 0) { S.c.huf_nw = nw; S.c.huf_log = maxbits; }
    return 0;
}

// One Huffman stream decoded by the 64 lanes of a wavefront (A.4), ~2 KiB of stream at a time:
//   * the segment is staged in LDS with coalesced 16-byte loads (`seg`, 2 KiB + 64 bytes, private to the wavefront);
//   * lane k starts at bit k*C of the segment (a guess for k > 0); lanes then re-start from their
//     predecessor's exit position until the chain is consistent.  Text-like codes self-synchronise within a
//     few symbols, so that takes a round or two.  Near-flat codes (noisy samples, already-compressed bytes)
//     do not: there the truth travels one lane per round -- but a lane can only ever be entered at one of
//     L bit offsets (a code word straddles its lower boundary by < L bits), so every lane keeps the
//     results of the offsets it has already walked (12 bits each) and a round normally costs a shuffle and a
//     lookup, not a walk;
//   * a DPP scan of the symbol counts gives the output offsets, and a last pass writes.
// A walk reads the stream through a 64-bit window loaded once per five symbols (5 * 11 bits <= 57).
constexpr int32_t kSegBits = 64 * 248; // 31 bytes per lane: lane windows fall into different LDS banks

#ifndef MZD_HUF_MINC
#define MZD_HUF_MINC 16
#endif
__device__ __noinline__ int huf_stream_wave(const uint8_t* sp, uint32_t sl, uint8_t* out, uint32_t nsym, uint32_t L, uint8_t* seg, int lane) {
    if (sl == 0) return MZD_E_CORRUPT;
    uint32_t last = sp[sl - 1];
    if (last == 0) return MZD_E_CORRUPT;
    const int32_t nbits = (int32_t)((sl - 1) * 8 + (uint32_t)hibit(last));
    const uint32_t mask = (1u << L) - 1;
    const uint16_t* const tab = S.huf;
    int32_t pos = 0;        // bits consumed so far (wave-uniform, exact)
    uint32_t done = 0;      // symbols written so far
    while (pos < nbits) {
        const int32_t s0 = pos, s1 = pos + kSegBits < nbits ? pos + kSegBits : nbits;
        // stage stream bytes [blo - 16, bhi): everything the segment can touch (16 bits of slack below it) behind a
        // 16-byte prefix, so that a window may start up to 8 bytes below the lowest byte needed; bytes below the
        // stream start read as zero (bits below bit 0 of a backward stream are zero)
        const int32_t lowbit = nbits - s1 - 16;
        const uint32_t blo = lowbit > 0 ? ((uint32_t)lowbit >> 3) & ~15u : 0u;
        const uint32_t bhi = (uint32_t)((nbits - s0) + 7) >> 3; // <= sl
        for (uint32_t o = (uint32_t)lane * 16; o < bhi - blo + 16; o += 1024) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (blo + o >= 16) __builtin_memcpy(&v, sp + (blo + o - 16), 16); // may over-read <= 15 bytes past the stream (input padding)
            *reinterpret_cast<uint4*>(seg + o) = v;
        }
        const int32_t seg_bias = 16 - (int32_t)blo; // stream byte j lives at seg[j + seg_bias] (the index is formed first: a pointer below `seg` would be out of bounds)
        int32_t C = (s1 - s0 + 63) / 64;
        if (C < MZD_HUF_MINC) C = MZD_HUF_MINC;
        int32_t q0 = s0 + lane * C, q1 = q0 + C;
        if (q0 > s1) q0 = s1;
        if (q1 > s1) q1 = s1;
        if (lane == 63) q1 = s1;
        const int32_t lim = nbits - q1;
        // decode from stream position `from` until the lane's upper boundary; returns the exit position
        auto walk = [&](int32_t from, uint32_t& cnt, uint8_t* dst) -> int32_t {
            int32_t rem = nbits - from; // bits below the read point
            uint32_t c = 0;
            uint64_t acc = 0; // the write pass packs 8 symbols per HBM store (byte stores cost a sector write each)
            while (rem > lim) {
                const int32_t bi = (rem - 57) >> 3; // window = stream bytes [bi, bi + 8): the 57..64 bits below the read point
                uint64_t W;
                __builtin_memcpy(&W, seg + (uint32_t)(bi + seg_bias), 8);
                int32_t h = rem - bi * 8;           // read point inside the window
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const bool act = rem > lim;
                    const uint32_t e = tab[(uint32_t)(W >> (uint32_t)(h - (int32_t)L)) & mask];
                    uint32_t l = e >> 8;
                    l = l ? l : 1u;
                    if (dst && act) {
                        acc |= (uint64_t)(e & 0xFF) << ((c & 7) * 8);
                        if ((c & 7) == 7) { __builtin_memcpy(dst + (c & ~7u), &acc, 8); acc = 0; }
                    }
                    l = act ? l : 0u;
                    c += act ? 1u : 0u;
                    h -= (int32_t)l;
                    rem -= (int32_t)l;
                }
            }
            if (dst) for (uint32_t k = c & ~7u; k < c; k++) { dst[k] = (uint8_t)acc; acc >>= 8; }
            cnt = c;
            return nbits - rem;
        };
        // results of the entry offsets already walked: 12 bits per offset j = start - q0 (0..10):
        // (exit - q1 + 1) | count << 4; 0 = not walked yet
        uint64_t m0 = 0, m1 = 0; uint32_t m2 = 0;
        auto memo_get = [&](uint32_t j) -> uint32_t {
            const uint64_t w = j < 5 ? m0 : (j < 10 ? m1 : (uint64_t)m2);
            const uint32_t sh = 12 * (j < 5 ? j : (j < 10 ? j - 5 : 0u));
            return j <= 10 ? (uint32_t)(w >> sh) & 0xFFFu : 0u;
        };
        auto memo_put = [&](uint32_t j, uint32_t e) {
            if (j < 5) m0 |= (uint64_t)e << (12 * j);
            else if (j < 10) m1 |= (uint64_t)e << (12 * (j - 5));
            else if (j == 10) m2 = e;
        };
        int32_t start = q0;
        uint32_t cnt = 0;
        int32_t exitp = walk(start, cnt, nullptr);
        if (cnt < 256 && (uint32_t)(exitp - q1) < 15) memo_put(0, (uint32_t)(exitp - q1 + 1) | (cnt << 4));
        for (int round = 0; round < 64; round++) {
            int32_t pe = __shfl_up(exitp, 1);
            int32_t ns = lane == 0 ? s0 : pe;
            bool changed = ns != start;
            if (!__any(changed)) break;
            bool need = false;
            uint32_t j = 0;
            if (changed) {
                start = ns;
                j = (uint32_t)(start - q0); // < L for a lane that is entered from below; anything else is simply walked
                const uint32_t e = memo_get(j);
                if (e) { exitp = q1 + (int32_t)(e & 15) - 1; cnt = e >> 4; }
                else need = true;
            }
            if (__any(need)) {
                if (need) {
                    exitp = walk(start, cnt, nullptr);
                    if (cnt < 256 && (uint32_t)(exitp - q1) < 15) memo_put(j, (uint32_t)(exitp - q1 + 1) | (cnt << 4));
                }
#ifdef MZD_STAMPS
                if (lane == 0) atomicAdd(&S.c.diag_slow, 1u); // diagnostic: synchronisation rounds that had to walk
#endif
            }
        }
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const int32_t endp = __builtin_amdgcn_readlane(exitp, 63);
        if (done + total > nsym || endp > nbits) return MZD_E_CORRUPT; // never write past this stream's share of the literals
        uint32_t dummy;
        walk(start, dummy, out + done + (incl - cnt));
        done += total;
        pos = endp;
    }
    if (done != nsym) return MZD_E_CORRUPT; // pos == nbits here: the stream was consumed exactly
    return 0;
}

// ------------------------------------------------------------------------------------ copies
// 64 lanes copy n bytes; regions do not overlap.
__device__ __noinline__ void wave_copy(uint8_t* d, const uint8_t* s, uint32_t n, int lane) {
    // head: bring d to 16-B alignment
    uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > n) head = n;
    if ((uint32_t)lane < head) d[lane] = s[lane];
    d += head; s += head; n -= head;
    uint32_t nv = n >> 4;
    uint32_t i = (uint32_t)lane;
    for (; i + 192 < nv; i += 256) { // four 16-byte loads in flight per lane (a lone wavefront is latency-bound)
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, s + (size_t)i * 16, 16);
        __builtin_memcpy(&v1, s + (size_t)(i + 64) * 16, 16);
        __builtin_memcpy(&v2, s + (size_t)(i + 128) * 16, 16);
        __builtin_memcpy(&v3, s + (size_t)(i + 192) * 16, 16);
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v0;
        *reinterpret_cast<uint4*>(d + (size_t)(i + 64) * 16) = v1;
        *reinterpret_cast<uint4*>(d + (size_t)(i + 128) * 16) = v2;
        *reinterpret_cast<uint4*>(d + (size_t)(i + 192) * 16) = v3;
    }
    for (; i < nv; i += 64) {
        uint4 v;
        __builtin_memcpy(&v, s + (size_t)i * 16, 16);
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v;
    }
    uint32_t tail = n & 15;
    if ((uint32_t)lane < tail) d[(size_t)nv * 16 + lane] = s[(size_t)nv * 16 + lane];
}

// n threads-of-a-workgroup version (raw blocks, RLE fills)
__device__ __noinline__ void wg_copy(uint8_t* d, const uint8_t* s, uint32_t n, int tid) {
    uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > n) head = n;
    if ((uint32_t)tid < head) d[tid] = s[tid];
    d += head; s += head; n -= head;
    uint32_t nv = n >> 4;
    for (uint32_t i = tid; i < nv; i += kWG) {
        uint4 v;
        __builtin_memcpy(&v, s + (size_t)i * 16, 16);
        *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v;
    }
    uint32_t tail = n & 15;
    if ((uint32_t)tid < tail) d[(size_t)nv * 16 + tid] = s[(size_t)nv * 16 + tid];
}

__device__ __noinline__ void wg_fill(uint8_t* d, uint32_t byte, uint32_t n, int tid) {
    uint32_t head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > n) head = n;
    if ((uint32_t)tid < head) d[tid] = (uint8_t)byte;
    d += head; n -= head;
    uint32_t w = byte * 0x01010101u;
    uint4 v = make_uint4(w, w, w, w);
    uint32_t nv = n >> 4;
    for (uint32_t i = tid; i < nv; i += kWG) *reinterpret_cast<uint4*>(d + (size_t)i * 16) = v;
    uint32_t tail = n & 15;
    if ((uint32_t)tid < tail) d[(size_t)nv * 16 + tid] = (uint8_t)byte;
}

// 64 lanes replicate the `off` bytes before d over d[0..n)  (a match whose source overlaps its
// destination: byte k = pattern[k mod off]).  Long ones (zero pages, sparse files: one match can be a
// whole block) are not done 64 bytes at a time: once at least 4 KiB of the pattern exist, byte k equals byte
// k - P for any multiple P of off, so the rest is plain 16-byte-per-lane copying from one period back, a period at a
// time (each period is complete -- and its stores have landed -- before the next one reads it).
__device__ __noinline__ void wave_pattern(uint8_t* d, uint32_t off, uint32_t n, int lane) {
    const uint8_t* pat = d - off;
    uint32_t period = off, done = 0;
    if (off < 4096) {
        period = ((4096 + off - 1) / off) * off;
        const uint32_t head = n < period ? n : period;
        uint32_t idx = (uint32_t)lane % off;
        const uint32_t step = 64u % off;
        for (uint32_t k = lane; k < head; k += 64) {
            d[k] = pat[idx];
            idx += step;
            if (idx >= off) idx -= off;
        }
        done = head;
        wg_fence();
    }
    while (done < n) {
        const uint32_t chunk = n - done < period ? n - done : period;
        wave_copy(d + done, d + done - period, chunk, lane);
        done += chunk;
        wg_fence();
    }
}

// ------------------------------------------------------------------------------------ K3 (wave-parallel)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, int lane) {
    (void)lane;
    using WS = rocprim::warp_scan<uint32_t, 64>;
    WS::storage_type* st = nullptr;
    uint32_t r;
    WS().inclusive_scan(v, r, *st, rocprim::maximum<uint32_t>());
    return r;
}

// FSE decode table (A.3) built by the 64 lanes of one wavefront; same result as build_seq_table.
//   A (lane = symbol)  counts -> low-probability symbols at the top, slot ranges by a scan
//   B (lane = 8 slots) slot -> symbol by a max-scan over range-start marks
//   C (lane = step j)  position (j*step)&mask, ranked among the positions below `high` by ballot
//   D (lane = symbol)  state numbering in table order: every symbol walks the table once
// tmp: 2 KiB of LDS scratch (tabsym[512], mark[512], per-symbol masks / counters / extra-bit counts).
__device__ __noinline__ void build_seq_table_wave(uint64_t* tab, const int16_t* norm, uint32_t nsym, uint32_t log, int kind, uint8_t* tmp, int lane) {
    uint8_t* const tabsym = tmp;
    uint8_t* const mark = tmp + 512;
    const uint32_t size = 1u << log, mask = size - 1;
    for (uint32_t k = lane; k < size; k += 64) mark[k] = 0;
    // A
    const int c = (uint32_t)lane < nsym ? norm[lane] : 0;
    const uint32_t is_low = c == -1 ? 1u : 0u, p = c > 0 ? (uint32_t)c : 0u;
    const uint32_t low_incl = wave_incl_scan(is_low, lane), p_incl = wave_incl_scan(p, lane);
    const uint32_t n_low = __builtin_amdgcn_readlane(low_incl, 63);
    const uint32_t high = size - n_low;
    if (is_low) tabsym[size - low_incl] = (uint8_t)lane; // first low symbol -> size-1, next -> size-2, ...
    if (p) mark[p_incl - p] = (uint8_t)lane;
    // B: slotSym[k] = max mark at or before k (symbols ascend with k; symbol 0's mark is 0 like "no mark")
    {
        const uint32_t k0 = (uint32_t)lane * 8;
        uint32_t m[8], run = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) { uint32_t v = k0 + t < size ? mark[k0 + t] : 0; run = v > run ? v : run; m[t] = run; }
        uint32_t incl = wave_incl_max(run, lane);
        uint32_t prev = __shfl_up(incl, 1);
        if (lane == 0) prev = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) if (k0 + t < size) mark[k0 + t] = (uint8_t)(m[t] > prev ? m[t] : prev);
    }
    // C
    {
        const uint32_t step = (size >> 1) + (size >> 3) + 3;
        uint32_t running = 0;
        for (uint32_t j0 = 0; j0 < size; j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            const uint32_t pj = (j * step) & mask;
            const bool v = j < size && pj < high;
            const uint64_t bal = __ballot(v);
            const uint32_t k = running + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1));
            if (v) tabsym[pj] = mark[k];
            running += (uint32_t)__builtin_popcountll(bal);
        }
    }
    // D (lane = table position, 64 ascending positions per step): the state number of a position is the
    // symbol's count + the number of lower positions holding the same symbol.  Inside a step that rank
    // comes from a per-symbol lane mask built with LDS atomic ORs; across steps from a per-symbol counter.
    {
        uint64_t* const smask = reinterpret_cast<uint64_t*>(tmp + 1024); // [64]
        uint32_t* const scnt = reinterpret_cast<uint32_t*>(tmp + 1536);  // [64]
        uint32_t* const sext = reinterpret_cast<uint32_t*>(tmp + 1792);  // [64] extra bits of each code
        smask[lane] = 0;
        scnt[lane] = c == -1 ? 1u : (c > 0 ? (uint32_t)c : 0u);
        sext[lane] = (uint32_t)lane < nsym ? code_extra((uint32_t)lane, kind) : 0;
        for (uint32_t i0 = 0; i0 < size; i0 += 64) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool act = i < size;
            const uint32_t sy = act ? tabsym[i] : 63;
            if (act) __atomic_fetch_or(&smask[sy], 1ull << lane, __ATOMIC_RELAXED);
            const uint64_t m = __atomic_load_n(&smask[sy], __ATOMIC_RELAXED);
            const uint32_t basec = scnt[sy];
            if (act) {
                const uint32_t d = basec + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
                const uint32_t nb = log - (uint32_t)hibit(d);
                const uint32_t extra = sext[sy];
                const uint32_t hi = nb | ((extra + nb) << 8) | (sy << 16) | (extra << 24);
                tab[i] = (uint64_t)(((d << nb) - size) * 8u) | ((uint64_t)hi << 32);
                if ((uint32_t)lane == 63u - (uint32_t)__builtin_clzll(m)) { // the symbol's highest position in this step
                    scnt[sy] = basec + (uint32_t)__builtin_popcountll(m);
                    __atomic_store_n(&smask[sy], 0ull, __ATOMIC_RELAXED);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------ K4
// FSE sequence decode (A.5), split in two:
//
//  (a) walk_sequences_wave -- the part that is serial by construction.  One bitstream carries three
//      interleaved tANS states; state i+1 depends on the bits state i consumed, so a
//      single wavefront walks it.  On a lone wavefront every instruction costs 6-7.5 cycles of issue (tools/micro/asm_micro.hip),
//      so the loop does only what the chain needs: three table reads + one 8-byte bitstream
//      window (all LDS, issued together), the bit budget of the sequence, the three state updates.
//      Per sequence it records {three state offsets, bit position} (16 bytes, its registers as they stand) and nothing else.
//  (b) field conversion -- everything that is NOT a chain: extra bits, base values.  One lane per
//      sequence, straight from the records of (a); done by the planning wavefront (plan_wave),
//      64 sequences at a time, while the walker is already further down the stream.
//  Repeat-offset resolution (a chain again, but a cheap one) happens in plan_wave.
//
// The bitstream is read backwards through an 8 KiB LDS ring, filled 1 KiB at a time with one 16-byte
// load per lane (coalesced).  Ring coordinates ("g-offsets") are stream byte index + bias,
// bias = 16 + (sp & 15): chunk boundaries are 16-B aligned in HBM and everything below the first
// stream byte reads as zero (bits below bit 0 of a backward stream are zero).  The first 16 bytes
// are mirrored behind the ring so that an unaligned 8-byte read never has to wrap.
struct SeqStream {
    const uint8_t* gbase; // HBM address of g-offset 0 (16-B aligned; may lie before the buffer, never dereferenced there)
    uint32_t bias;        // g-offset of stream byte 0
    uint32_t gend;        // g-offset one past the last stream byte
    int32_t lowest;       // lowest chunk resident in the ring
};

__device__ __forceinline__ void ring_load_chunk(const SeqStream& st, int32_t chunk, int lane) {
    uint32_t o = (uint32_t)chunk * kChunk + (uint32_t)lane * 16; // g-offset of this lane's piece
    uint4 v = make_uint4(0, 0, 0, 0);
    if (o + 16 > st.bias && o < st.gend) {
        v = *reinterpret_cast<const uint4*>(st.gbase + o);
        if (o < st.bias) { // zero the bytes in front of the stream
            uint32_t z = st.bias - o; // 1..15
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t lo = (uint32_t)k * 4;
                if (z >= lo + 4) w[k] = 0;
                else if (z > lo) w[k] &= ~0u << ((z - lo) * 8);
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    uint32_t slot = (uint32_t)chunk & (kRingChunks - 1);
    *reinterpret_cast<uint4*>(&S.ring[slot * kChunk + (uint32_t)lane * 16]) = v;
    if (slot == 0 && lane == 0) *reinterpret_cast<uint4*>(&S.ring[kRingBytes]) = v; // mirror
}

// the 8 ring bytes that end at g-offset e (exclusive), as a little-endian u64
__device__ __forceinline__ uint64_t ring_read64(uint32_t e) {
    uint64_t v;
    __builtin_memcpy(&v, &S.ring[(e - 8) & (kRingBytes - 1)], 8);
    return v;
}

// walk record (uint4): LL, ML, OF state offsets, g-bit position - 32 -- the walker's state as it stands
//   (state offsets are byte offsets into the tables: 8 * state)

constexpr uint32_t kWalkFin = 0x80000000u;
constexpr uint32_t kNoJob = 0xFFFFFFFFu;
constexpr uint32_t kDoneJob = 0xFFFFFFFEu; // the queue is empty
#ifndef MZD_PRE_PRIO
#define MZD_PRE_PRIO 2
#endif
constexpr uint32_t kPreStage = 2304;        // S.ring[2304 .. 3072): between the Huffman segments of the two helper wavefronts
constexpr uint32_t kInRing = 0x80000000u;   // parse_seq_header: the staged header lies in S.ring, not in S.stage

// The hot form of the chain, hand-scheduled: runs of kWalkGroup steps until n steps are done, or a group met a
// sequence wider than its window (slack < 0: the group is void, the caller takes it again carefully from {sx, sy}, the
// packed state at the group's start), or the read head comes within one group of the lowest resident ring chunk
// (Gm < thresh: the caller refills).  A lone wavefront issues in order, 6.25 (4-byte encodings) to 7.5 cycles (8-byte) an instruction, and the four LDS reads
// return through a 64 B/clk path (32 clks): a step is the 12 chain instructions + the reads' round trip, ~137 cycles
// (tools/micro/asm_micro.hip); whatever the chain does not need -- the record store, packing the next record, the
// slack bookkeeping, publishing progress -- sits behind the reads, in the shadow of their latency.  Same arithmetic
// as the careful C++ step.  Progress (records visible to the planner: all but the newest kWalkLag stores have landed) is
// published once per group.  Registers: v[48:55] the three entries and the window, v[64:71] temporaries, v[80:81]
// the packed record (all caller-saved in the AMDGPU calling convention).  Table and ring addresses are immediates:
// S must start at LDS address 0 (checked by the caller, which otherwise keeps the C++ form).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kWalkGroup = 8;
constexpr uint32_t kWalkLag = 32;
#define MZD_SDWA_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
// v[84:87] = {LL, ML, OF state offsets, read head - 32}: the record of a step is its state, stored as it stands
#define MZD_WALK_STEP(SH, RECOFF, TAIL) \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_add3_u32 v64, v49, v51, v53\n"                 /* nbBits sums | total bits << 8 | ... */ \
    "v_add_u32_e32 v65, v49, v51\n"                   /* bit offset of the LL field: nbO + nbM (low 5 bits) */ \
    "v_sub_u32_sdwa " SH ", %[av], v64" MZD_SDWA_B1   /* window bits below what this sequence consumes */ \
    "v_sub_u32_sdwa v87, v87, v64" MZD_SDWA_B1 \
    "v_lshrrev_b64 v[66:67], " SH ", v[54:55]\n" \
    "v_lshrrev_b32_e32 v71, 3, v87\n" \
    "v_bfe_u32 v64, v66, 0, v49\n" \
    "v_bfe_u32 v69, v66, v49, v51\n" \
    "v_bfe_u32 v70, v66, v65, v53\n" \
    "v_lshl_add_u32 v86, v64, 3, v48\n" \
    "v_lshl_add_u32 v85, v69, 3, v50\n" \
    "v_lshl_add_u32 v84, v70, 3, v52\n" \
    "ds_read_b64 v[48:49], v86 offset:%[oO]\n" \
    "ds_read_b64 v[50:51], v85 offset:%[oM]\n" \
    "ds_read_b64 v[52:53], v84 offset:%[oL]\n" \
    "v_and_b32_e32 v71, 0x1ffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    "global_store_dwordx4 %[woff], v[84:87], %[base] offset:" RECOFF "\n" /* the NEXT step's record: the state as it is now */ \
    "v_and_or_b32 %[av], v87, 31, 32\n" \
    TAIL
#define MZD_WALK_SLACK "v_min3_i32 %[slack], %[slack], %[sa], %[sb]\n"
__device__ __forceinline__ void walk_run_asm(uint32_t& vL, uint32_t& vM, uint32_t& vO, uint32_t& Gm, uint32_t& woff, int32_t& slack, uint32_t& n,
                                             int32_t pv, uint4& start, int32_t thresh, uint32_t prog_lds,
                                             __attribute__((address_space(1))) uint8_t* gwalk) {
    static_assert(kRingBytes - 4 == 0x1ffc && offsetof(Shared, ring) == 0, "the window address mask / the ring's place are spelled out in MZD_WALK_STEP");
    static_assert(kWalkGroup == 8 && kWalkLag == 32, "spelled out below");
    uint32_t av, sa, sb;
    asm volatile(
        "v_mov_b32_e32 v84, %[vL]\n v_mov_b32_e32 v85, %[vM]\n v_mov_b32_e32 v86, %[vO]\n v_mov_b32_e32 v87, %[Gm]\n"
        "v_lshrrev_b32_e32 v71, 3, v87\n"
        "ds_read_b64 v[48:49], v86 offset:%[oO]\n"
        "ds_read_b64 v[50:51], v85 offset:%[oM]\n"
        "ds_read_b64 v[52:53], v84 offset:%[oL]\n"
        "v_and_b32_e32 v71, 0x1ffc, v71\n"
        "ds_read2_b32 v[54:55], v71 offset1:1\n"
        "global_store_dwordx4 %[woff], v[84:87], %[base]\n" // the first step's record
        "v_and_or_b32 %[av], v87, 31, 32\n"
        "1:\n"
        "v_mov_b32_e32 %[s0], v84\n v_mov_b32_e32 %[s1], v85\n v_mov_b32_e32 %[s2], v86\n v_mov_b32_e32 %[s3], v87\n" // the group's starting state (the reads are in flight)
        MZD_WALK_STEP("%[sa]", "16", "")
        MZD_WALK_STEP("%[sb]", "32", MZD_WALK_SLACK "s_waitcnt vmcnt(32)\n v_max_i32_e32 v69, 0, %[pv]\n ds_write_b32 %[prog], v69\n v_add_u32_e32 %[pv], 8, %[pv]\n") // publish
        MZD_WALK_STEP("%[sa]", "48", "")
        MZD_WALK_STEP("%[sb]", "64", MZD_WALK_SLACK)
        MZD_WALK_STEP("%[sa]", "80", "")
        MZD_WALK_STEP("%[sb]", "96", MZD_WALK_SLACK)
        MZD_WALK_STEP("%[sa]", "112", "")
        MZD_WALK_STEP("%[sb]", "128", MZD_WALK_SLACK "v_add_u32_e32 %[woff], 128, %[woff]\n v_subrev_u32_e32 v69, %[thresh], v87\n v_min_i32_e32 v69, v69, %[slack]\n v_cmp_gt_i32_e32 vcc, 0, v69\n")
        "s_sub_u32 %[n], %[n], 8\n"
        "s_cbranch_vccnz 2f\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "2:\n"
        "s_waitcnt lgkmcnt(0)\n" // (the reads issued by the last step: nothing may be in flight into v[48:55] past this block)
        "v_mov_b32_e32 %[vL], v84\n v_mov_b32_e32 %[vM], v85\n v_mov_b32_e32 %[vO], v86\n v_mov_b32_e32 %[Gm], v87\n"
        : [vL] "+v"(vL), [vM] "+v"(vM), [vO] "+v"(vO), [Gm] "+v"(Gm), [woff] "+v"(woff), [slack] "+v"(slack), [av] "=&v"(av), [n] "+s"(n),
          [pv] "+v"(pv), [s0] "=&v"(start.x), [s1] "=&v"(start.y), [s2] "=&v"(start.z), [s3] "=&v"(start.w), [sa] "=&v"(sa), [sb] "=&v"(sb)
        : [base] "s"(gwalk), [thresh] "s"(thresh), [prog] "v"(prog_lds),
          [oL] "n"(offsetof(Shared, ll)), [oM] "n"(offsetof(Shared, ml)), [oO] "n"(offsetof(Shared, of))
        : "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v84", "v85", "v86", "v87", "vcc", "scc", "memory");
}

__device__ __noinline__ int walk_sequences_wave(const uint8_t* sp, uint32_t sl, uint32_t nseq_in, uint4* walk, uint32_t* prog, int lane) {
    const uint32_t nseq = __builtin_amdgcn_readfirstlane(nseq_in);
    // a global (not flat) pointer: flat stores would also count on lgkmcnt, i.e. sit in the LDS waits below
    __attribute__((address_space(1))) uint8_t* gwalk;
    {
        uint64_t wp = (uint64_t)(uintptr_t)walk;
        wp = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wp) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wp >> 32)) << 32); // the builtin returns int: no sign extension
        gwalk = (__attribute__((address_space(1))) uint8_t*)wp;
    }
    uint32_t woff = 0; // byte offset of the next record (a VGPR next to a scalar base: cheapest store form)
    asm volatile("" : "+v"(woff));
    if (sl == 0) return MZD_E_CORRUPT;
    uint32_t last = sp[sl - 1];
    if (last == 0) return MZD_E_CORRUPT;
    SeqStream st;
    uint32_t skew = (uint32_t)((uintptr_t)sp & 15);
    st.bias = 16 + skew;
    st.gbase = sp - st.bias;
    st.gend = sl + st.bias;
    const uint32_t Gzero = st.bias * 8; // read head at stream bit 0
    uint32_t G = (sl - 1) * 8 + (uint32_t)hibit(last) + Gzero; // g-bits below the read head
    int32_t top = (int32_t)((st.gend - 1) / kChunk);
    st.lowest = top;
    ring_load_chunk(st, top, lane);
    if (top >= 1) { ring_load_chunk(st, top - 1, lane); st.lowest = top - 1; }

    const uint32_t alL = S.c.al[0], alO = S.c.al[1], alM = S.c.al[2];
    uint32_t vL, vO, vM; // state byte offsets
    {
        uint32_t e = (G + 7) >> 3;
        uint64_t B = ring_read64(e) << (e * 8 - G);
        uint32_t n = alL + alO + alM;
        if (G - Gzero < n) return MZD_E_CORRUPT;
        vL = alL ? (uint32_t)(B >> (64 - alL)) : 0; B <<= alL;
        vO = alO ? (uint32_t)(B >> (64 - alO)) : 0; B <<= alO;
        vM = alM ? (uint32_t)(B >> (64 - alM)) : 0;
        G -= n;
        vL *= 8; vO *= 8; vM *= 8;
    }
    const uint8_t* const tL = reinterpret_cast<const uint8_t*>(S.ll);
    const uint8_t* const tM = reinterpret_cast<const uint8_t*>(S.ml);
    const uint8_t* const tO = reinterpret_cast<const uint8_t*>(S.of);
    uint32_t i = 0;
    const uint32_t nupd = nseq - 1; // sequences followed by a state update
    const bool lds_at_zero = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)S.ring == 0; // (walk_run_asm spells LDS addresses out)
    const uint32_t prog_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)prog;
    uint32_t Gm = G - 32; // the loop carries the read head minus 32 (saves an add per sequence)
    // One careful step of the chain: the window moves down a dword at a time until the sequence fits (long extra-bit
    // fields: about one sequence in hundreds).  The hot form (walk_run_asm) has no such branch -- a branch on freshly
    // loaded LDS data costs ~35 cycles per sequence on a lone wavefront -- it only notes that a group met such a sequence.
    auto careful_step = [&]() {
        uint64_t eL, eM, eO;
        __builtin_memcpy(&eL, tL + vL, 8);
        __builtin_memcpy(&eM, tM + vM, 8);
        __builtin_memcpy(&eO, tO + vO, 8);
        // window: the 8 ring bytes at the 4-byte aligned address whose 64 bits end above the read head
        // (two aligned dwords; an unaligned 8-byte LDS read costs ~40 cycles more)
        const uint32_t u = Gm; // read head - 32
        uint32_t ra = (u >> 3) & (kRingBytes - 4);
        uint64_t X;
        __builtin_memcpy(&X, &S.ring[ra], 8);
        *(__attribute__((address_space(1))) u32x4*)(gwalk + woff) = u32x4{vL, vM, vO, Gm};
        woff += 16;
        const uint32_t hL = (uint32_t)(eL >> 32), hM = (uint32_t)(eM >> 32), hO = (uint32_t)(eO >> 32);
        const uint32_t total = ((hL + hM + hO) >> 8) & 0xFF;
        uint32_t av = (u & 31) | 32; // bits of the window below the read head: 32..63
        while (__builtin_amdgcn_ballot_w64(total > av) != 0) {
            ra = (ra - 4) & (kRingBytes - 4);
            __builtin_memcpy(&X, &S.ring[ra], 8);
            av += 32;
        }
        // fresh state bits sit at the bottom of what this sequence consumes: OF lowest, then ML, then LL
        // (at most 26 bits together: one 64-bit shift, then 32-bit field extracts)
        const uint32_t Y = (uint32_t)(X >> ((av - total) & 63));
        const uint32_t bO = __builtin_amdgcn_ubfe(Y, 0, hO);           // width = nbBits, the low bits of the entry
        const uint32_t bM = __builtin_amdgcn_ubfe(Y, hO, hM);          // offset nbO (low 5 bits)
        const uint32_t bL = __builtin_amdgcn_ubfe(Y, hO + hM, hL);     // offset nbO + nbM
        vO = (uint32_t)eO + (bO << 3);
        vM = (uint32_t)eM + (bM << 3);
        vL = (uint32_t)eL + (bL << 3);
        Gm -= total;
    };
    constexpr int32_t kLook = (int32_t)(kWalkGroup * 12 + 24) * 8; // bits a group can consume (<= 89 a sequence) + the window above the head
    while (i < nupd) {
        // keep the ring one group ahead of the read head
        while (st.lowest > 0 && (int32_t)Gm < st.lowest * (int32_t)(kChunk * 8) + kLook) {
            st.lowest--;
            ring_load_chunk(st, st.lowest, lane);
        }
        const uint32_t left = nupd - i;
        if (lds_at_zero && left >= kWalkGroup) {
            uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(left & ~(kWalkGroup - 1));
            const uint32_t n0 = n;
            // (the whole stream resident: the run still ends at the first group that read past the stream's start -- a corrupt
            //  stream: published records must never carry a position outside the stream, the planner addresses HBM with them;
            //  records younger than kWalkLag are not published, so stopping at the group's end is early enough)
            const int32_t thresh = __builtin_amdgcn_readfirstlane(st.lowest > 0 ? st.lowest * (int32_t)(kChunk * 8) + kLook : (int32_t)Gzero - 32);
            int32_t slack = 64; // minimum over a group of (window bits - bits needed)
            uint4 start;
            walk_run_asm(vL, vM, vO, Gm, woff, slack, n, (int32_t)i - (int32_t)kWalkLag, start, thresh, prog_lds, gwalk);
            i += n0 - n;
            if (__builtin_amdgcn_ballot_w64(slack < 0) != 0) { // the last group is void: once more from its start, carefully
                i -= kWalkGroup; woff -= 16 * kWalkGroup;
                vL = start.x; vM = start.y; vO = start.z; Gm = start.w;
                for (uint32_t k = 0; k < kWalkGroup; k++) careful_step();
                i += kWalkGroup;
            }
        } else {
            const uint32_t stop = left < kWalkGroup ? nupd : i + kWalkGroup;
            for (; i < stop; i++) careful_step();
        }
        if ((int32_t)(Gm + 32 - Gzero) < 0) return MZD_E_CORRUPT; // over-read
    }
    G = Gm + 32;
    // last sequence: extra bits only
    {
        uint64_t eL, eM, eO;
        __builtin_memcpy(&eL, tL + vL, 8);
        __builtin_memcpy(&eM, tM + vM, 8);
        __builtin_memcpy(&eO, tO + vO, 8);
        *(__attribute__((address_space(1))) u32x4*)(gwalk + woff) = u32x4{vL, vM, vO, G - 32};
        uint32_t extra = (uint32_t)(eL >> 56) + (uint32_t)(eM >> 56) + (uint32_t)(eO >> 56);
        if (G - Gzero != extra) return MZD_E_CORRUPT; // the bitstream must be consumed exactly
    }
    return 0; // the caller publishes nseq | kWalkFin after a release fence
}

// n bits (n <= 32) whose top is g-bit `top` (exclusive), read from HBM
__device__ __forceinline__ uint32_t stream_bits(const uint8_t* gbase, uint32_t top, uint32_t n) {
    uint32_t lo = top - n;
    uint64_t v = ldu64(gbase + (lo >> 3)) >> (lo & 7);
    return n ? (uint32_t)v & (uint32_t)((1ull << n) - 1) : 0u;
}

// ------------------------------------------------------------------------------------ K5
// Sequence execution (A.5) by one wavefront, 64 sequences per step (lane = sequence).
//   1. repeat offsets: the rule of A.5 is a chain over the sequences; it is resolved with a
//      wave scan over "symbolic" register-file transforms (each of the three slots is either a
//      constant or an input slot plus a delta), so 64 sequences cost log2(64) shuffle rounds;
//   2. scans of ll and ll+ml give every lane its literal source and its output position;
//   3. runs of short sequences are assembled in an LDS staging buffer (kStage bytes): literals and
//      matches whose source lies before the run come from HBM with 8-byte accesses, matches
//      whose source is inside the run are resolved LDS->LDS in rounds (a match is ready when its
//      source lies below the output of the first unfinished sequence), then the run is flushed
//      to HBM with coalesced 16-byte stores;
//   4. long literal runs / matches bypass the staging buffer and are copied by all 64 lanes
//      (overlapping matches replicate their pattern).
constexpr uint32_t kStage = 2048;
constexpr uint32_t kShort = 64; // longest literal run / match that goes through the staging buffer

typedef __attribute__((address_space(3))) uint8_t* lds_p;

struct RepOp { uint32_t s; int32_t v0, v1, v2; }; // s: 2 bits per slot (0..2 input slot, 3 constant)
__device__ __forceinline__ uint32_t rep_src(uint32_t s, int j) { return (s >> (2 * j)) & 3; }
// result = g applied after f.  All selects work on values pinned in registers: left to itself the
// compiler turns "pick one of three struct fields" into an indexed load from a stack copy of the
// struct, i.e. three dependent scratch-memory round trips per scan step.
__device__ __forceinline__ int32_t sel3(uint32_t k, int32_t a0, int32_t a1, int32_t a2) {
    int32_t r = k == 1 ? a1 : a2;
    return k == 0 ? a0 : r;
}
__device__ __forceinline__ RepOp rep_compose(RepOp g, RepOp f) {
    asm volatile("" : "+v"(f.s), "+v"(f.v0), "+v"(f.v1), "+v"(f.v2));
    asm volatile("" : "+v"(g.s), "+v"(g.v0), "+v"(g.v1), "+v"(g.v2));
    RepOp r;
    const uint32_t g0 = g.s & 3, g1 = (g.s >> 2) & 3, g2 = (g.s >> 4) & 3;
    const uint32_t s0 = g0 == 3 ? 3u : (f.s >> (2 * g0)) & 3;
    const uint32_t s1 = g1 == 3 ? 3u : (f.s >> (2 * g1)) & 3;
    const uint32_t s2 = g2 == 3 ? 3u : (f.s >> (2 * g2)) & 3;
    r.s = s0 | (s1 << 2) | (s2 << 4);
    r.v0 = g.v0 + (g0 == 3 ? 0 : sel3(g0, f.v0, f.v1, f.v2));
    r.v1 = g.v1 + (g1 == 3 ? 0 : sel3(g1, f.v0, f.v1, f.v2));
    r.v2 = g.v2 + (g2 == 3 ? 0 : sel3(g2, f.v0, f.v1, f.v2));
    return r;
}
__device__ __forceinline__ uint32_t rep_eval(RepOp f, int j, uint32_t r0, uint32_t r1, uint32_t r2) {
    asm volatile("" : "+v"(f.s), "+v"(f.v0), "+v"(f.v1), "+v"(f.v2));
    const uint32_t src = rep_src(f.s, j);
    const int32_t v = j == 0 ? f.v0 : (j == 1 ? f.v1 : f.v2);
    const uint32_t in = (uint32_t)sel3(src, (int32_t)r0, (int32_t)r1, (int32_t)r2);
    return (src == 3 ? 0u : in) + (uint32_t)v;
}

// Per-lane copies of n (<= 64) bytes, 8 bytes at a time plus one (over-reading) 8-byte tail word stored
// as exact 4/2/1 pieces.  On a SIMD machine every step costs issue slots whether or not a lane takes
// part, so the chunk loops stop at the longest copy in the wavefront (wave-uniform `__any` exits:
// typical matches are 4..24 bytes, typical literal runs 0..8).  Loads and stores are separate halves so
// that a run's HBM loads can be issued a whole pipeline step before they are needed.  All sources may be
// read up to 7 bytes past their end (LDS: always in bounds; literals and frame bytes: padded buffers).
typedef const __attribute__((address_space(1))) uint8_t* gcptr;
struct GlobalLd {
    const uint8_t* p;
    __device__ __forceinline__ uint64_t u64(uint32_t o) const { uint64_t v; __builtin_memcpy(&v, (gcptr)(p + o), 8); return v; }
};
struct LdsLd {
    const uint8_t* p;
    __device__ __forceinline__ uint64_t u64(uint32_t o) const { uint64_t v; __builtin_memcpy(&v, p + o, 8); return v; }
};
struct LdsSt {
    uint8_t* p;
    __device__ __forceinline__ void u64(uint32_t o, uint64_t v) const { __builtin_memcpy(p + o, &v, 8); }
    __device__ __forceinline__ void u32(uint32_t o, uint32_t v) const { __builtin_memcpy(p + o, &v, 4); }
    __device__ __forceinline__ void u16(uint32_t o, uint32_t v) const { uint16_t w = (uint16_t)v; __builtin_memcpy(p + o, &w, 2); }
    __device__ __forceinline__ void u8(uint32_t o, uint32_t v) const { p[o] = (uint8_t)v; }
};
template <int NQ> struct CopyRegs { uint64_t v[NQ]; uint64_t tl; }; // NQ full 8-byte chunks + the tail word
template <int NQ, class LD>
__device__ __forceinline__ void regs_load(uint32_t n, LD ld, CopyRegs<NQ>& r) { // n <= 8 * NQ + 7; n = 0 on idle lanes
    const uint32_t q = n >> 3;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)NQ; j++) {
        if (!__any(j < q)) break;
        if (j < q) r.v[j] = ld.u64(j * 8);
    }
    if (n & 7) r.tl = ld.u64(q * 8);
}
template <int NQ, class ST>
__device__ __forceinline__ void regs_store(uint32_t n, ST st, const CopyRegs<NQ>& r) {
    const uint32_t q = n >> 3, t = q * 8;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)NQ; j++) {
        if (!__any(j < q)) break;
        if (j < q) st.u64(j * 8, r.v[j]);
    }
    if (n & 4) st.u32(t, (uint32_t)r.tl);
    if (n & 2) st.u16(t + (n & 4), (uint32_t)(r.tl >> ((n & 4) * 8)));
    if (n & 1) st.u8(t + (n & 6), (uint32_t)(r.tl >> ((n & 6) * 8)));
}
template <class LD, class ST>
__device__ __forceinline__ void copy_short(uint32_t n, LD ld, ST st) { // n <= 64 (n == 64: eight chunks, no tail)
    CopyRegs<8> r;
    regs_load<8>(n, ld, r);
    regs_store<8>(n, st, r);
}

constexpr uint32_t kPlanFin = 0x80000000u;
constexpr int kPlanBlockTooLong = -64; // plan_wave: the block's output passes 128 KiB (internal: becomes Ctl::plan_too_long)

struct PlanCtx { // what the planning wavefront needs
    const uint4* walk;       // state-walk records of the block (HBM scratch)
    const uint8_t* seq_sp;   // the block's sequence bitstream
    const uint32_t* prog;    // walker progress (LDS)
    uint32_t nlit;
    uint32_t rep_known;      // the repeat offsets at the start of the block are known (first block of a frame)
    uint32_t rep[3];
};

// Offsets in the plan: a plain value, or -- when the block starts before its predecessor has finished, so that
// the repeat offsets at its start are still unknown -- a reference to one of the three start slots plus a delta.
// The copier resolves those (it runs after the predecessor).  0 is never a valid offset.
constexpr uint32_t kOffTag = 0x80000000u;
constexpr int32_t kOffBias = 1 << 28;
__device__ __forceinline__ uint32_t off_symbolic(uint32_t slot, int32_t delta) { return kOffTag | (slot << 29) | ((uint32_t)(delta + kOffBias) & 0x1FFFFFFFu); }

// K4(b) + the bookkeeping half of K5, by one wavefront, 64 sequences per step (lane = sequence):
// field conversion from the walk records, repeat offsets, positions, what can be validated without knowing
// where the block's output starts (the copier checks capacity and offsets).  The result goes to the plan array in
// HBM: per sequence {ll, ml, off, output offset inside the chunk}.  The block's total repeat-offset transform
// (start slots -> end slots) is left in S.c.rep_op.  Returns 0 or an error.
__device__ __noinline__ int plan_wave(uint4* seqs, uint32_t nseq_in, const PlanCtx& cx, int lane) {
    const uint32_t nseq = __builtin_amdgcn_readfirstlane(nseq_in);
    uint32_t opos = 0; // output produced so far, relative to the block start
    uint32_t lpos = 0;
    // R: block start -> before the current chunk.  Known start offsets make it a constant map (every offset then
    // comes out as a plain value); unknown ones the identity.
    RepOp R;
    if (cx.rep_known) { R.s = 3 | (3 << 2) | (3 << 4); R.v0 = (int32_t)cx.rep[0]; R.v1 = (int32_t)cx.rep[1]; R.v2 = (int32_t)cx.rep[2]; }
    else { R.s = 0 | (1 << 2) | (2 << 4); R.v0 = 0; R.v1 = 0; R.v2 = 0; }
    // The walk records and the extra bits live in HBM (the walker may be arbitrarily far ahead, e.g. while
    // the literals are still being decoded).  Their latency is taken off this wavefront's critical path
    // by a two-stage software pipeline: while chunk k is planned, the records of chunk k+2 and the bit
    // windows of chunk k+1 are in flight.
    const uint32_t bias = 16 + (uint32_t)((uintptr_t)cx.seq_sp & 15);
    const uint8_t* const gbase = cx.seq_sp - bias;
    auto wait_walker = [&](uint32_t need) -> bool { // true when sequences [0, need) are recorded
        if (need > nseq) need = nseq;
        uint32_t pg = 0, it = 0;
        for (; it < (1u << 24); it++) {
            pg = flag_load(cx.prog);
            if ((pg & ~kWalkFin) >= need || (pg & kWalkFin)) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (it == (1u << 24)) post_err(&S.c.err, MZD_E_DEVICE); // (a wait that ran out: see spin_ge)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return (pg & ~kWalkFin) >= need;
    };
    struct Win { uint32_t hL, hM, hO, G; uint64_t bO, bM, bL; }; // entry words + raw 8-byte windows of one sequence
    auto load_rec = [&](uint32_t idx) -> uint4 { return idx < nseq ? cx.walk[idx] : make_uint4(0, 0, 0, 0); };
    auto issue_bits = [&](uint4 w, bool live, Win& o) {
        const uint32_t vL = w.x, vM = w.y, vO = w.z;
        o.G = w.w + 32; // records carry the read head - 32
        o.hL = (uint32_t)(S.ll[vL >> 3] >> 32); o.hM = (uint32_t)(S.ml[vM >> 3] >> 32); o.hO = (uint32_t)(S.of[vO >> 3] >> 32);
        o.bO = 0; o.bM = 0; o.bL = 0;
        if (live) {
            const uint32_t xM = o.hM >> 24, xO = o.hO >> 24, xL = o.hL >> 24;
            const uint32_t tO = o.G - xO, tM = tO - xM, tL = tM - xL; // bottoms of the three fields
            o.bO = ldu64(gbase + (tO >> 3)); o.bM = ldu64(gbase + (tM >> 3)); o.bL = ldu64(gbase + (tL >> 3));
        }
    };
    if (!wait_walker(128)) return MZD_E_CORRUPT;
    uint4 recA = load_rec((uint32_t)lane), recB = load_rec(64 + (uint32_t)lane); // chunks 0 and 1
    Win win;
    issue_bits(recA, (uint32_t)lane < nseq, win);
    uint32_t chunk = 0;
    for (uint32_t base = 0; base < nseq; base += 64, chunk++) {
        const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
        const uint32_t i = base + (uint32_t)lane;
        const bool valid = (uint32_t)lane < cnt;
        // everything this wavefront stored an iteration ago has landed: chunk k-1 of the plan is public
        wg_fence();
        if (lane == 0) flag_store(&S.c.plan_prog, chunk);
        // stage 1: records of chunk k+2, bit windows of chunk k+1 (recB arrived an iteration ago)
        if (!wait_walker(base + 192)) return MZD_E_CORRUPT; // the walker failed (it posted the error) or never got there
        const uint4 recC = load_rec(base + 128 + (uint32_t)lane);
        Win next;
        issue_bits(recB, base + 64 + (uint32_t)lane < nseq, next);
        // stage 2: fields of chunk k from the windows issued an iteration ago
        uint32_t ll = 0, ml = 0, ofv = 4;
        if (valid) {
            const uint32_t cL = (win.hL >> 16) & 0xFF, cM = (win.hM >> 16) & 0xFF, cO = (win.hO >> 16) & 0xFF;
            const uint32_t xL = win.hL >> 24, xM = win.hM >> 24, xO = win.hO >> 24;
            const uint32_t tO = win.G - xO, tM = tO - xM, tL = tM - xL;
            const uint32_t vO = xO ? (uint32_t)(win.bO >> (tO & 7)) & (uint32_t)((1ull << xO) - 1) : 0u;
            const uint32_t vM = xM ? (uint32_t)(win.bM >> (tM & 7)) & (uint32_t)((1ull << xM) - 1) : 0u;
            const uint32_t vL = xL ? (uint32_t)(win.bL >> (tL & 7)) & (uint32_t)((1ull << xL) - 1) : 0u;
            ofv = (1u << cO) + vO;
            ml = S.ml_base[cM] + vM;
            ll = S.ll_base[cL] + vL;
        }
        win = next; recB = recC;
        // ---- repeat offsets
        uint32_t off;
        {
            RepOp op;
            uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u);
            if (!valid || (ofv <= 3 && idx == 0)) { op.s = 0 | (1 << 2) | (2 << 4); op.v0 = 0; op.v1 = 0; op.v2 = 0; }
            else if (ofv > 3) { op.s = 3 | (0 << 2) | (1 << 4); op.v0 = (int32_t)(ofv - 3); op.v1 = 0; op.v2 = 0; }
            else if (idx == 1) { op.s = 1 | (0 << 2) | (2 << 4); op.v0 = 0; op.v1 = 0; op.v2 = 0; }
            else if (idx == 2) { op.s = 2 | (0 << 2) | (1 << 4); op.v0 = 0; op.v1 = 0; op.v2 = 0; }
            else { op.s = 0 | (0 << 2) | (1 << 4); op.v0 = -1; op.v1 = 0; op.v2 = 0; }
            RepOp acc; // inclusive scan: acc = op_lane o ... o op_0 (DPP path)
            {
                using WR = rocprim::warp_scan<RepOp, 64>;
                WR::storage_type* st = nullptr;
                WR().inclusive_scan(op, acc, *st, [](const RepOp& earlier, const RepOp& later) { return rep_compose(later, earlier); });
            }
            RepOp before; // exclusive
            before.s = __shfl_up(acc.s, 1); before.v0 = __shfl_up(acc.v0, 1); before.v1 = __shfl_up(acc.v1, 1); before.v2 = __shfl_up(acc.v2, 1);
            if (lane == 0) { before.s = 0 | (1 << 2) | (2 << 4); before.v0 = 0; before.v1 = 0; before.v2 = 0; }
            const RepOp T = rep_compose(before, R); // block start -> just before this sequence
            if (ofv > 3) off = ofv - 3;
            else {
                const uint32_t slot = idx == 1 ? 1u : (idx == 2 ? 2u : 0u); // idx 0 and 3 read slot 0
                const uint32_t src = (T.s >> (2 * slot)) & 3;
                const int32_t v = sel3(slot, T.v0, T.v1, T.v2) - (idx == 3 ? 1 : 0);
                if (src == 3) off = v > 0 ? (uint32_t)v : 0u; // 0: invalid, the copier rejects it
                else off = off_symbolic(src, v);
            }
            // chunk end -> R of the next chunk
            RepOp last;
            last.s = __builtin_amdgcn_readlane(acc.s, 63); last.v0 = __builtin_amdgcn_readlane(acc.v0, 63);
            last.v1 = __builtin_amdgcn_readlane(acc.v1, 63); last.v2 = __builtin_amdgcn_readlane(acc.v2, 63);
            R = rep_compose(last, R);
        }
        // ---- positions and validation
        const uint32_t tot = ll + ml;
        const uint32_t incl_t = wave_incl_scan(tot, lane), incl_l = wave_incl_scan(ll, lane);
        const uint32_t chunk_tot = __builtin_amdgcn_readlane(incl_t, 63), chunk_lit = __builtin_amdgcn_readlane(incl_l, 63);
        const uint32_t ex_t = incl_t - tot; // this sequence's output offset inside the 64-chunk
        // the plan of this sequence: {ll, ml, offset, output offset inside the chunk} -> HBM (unbounded, so the
        // planner never waits for the copier, which may still be decoding literals); also what mzd_debug_last_block shows
        if (valid) seqs[i] = make_uint4(ll, ml, off, ex_t);
        if (chunk_lit > cx.nlit - lpos || opos + chunk_tot > kBlockMax) {
            // the literals run out, or the block's output passes 128 KiB, inside this chunk: it is still published -- the copier
            // finds the first offending sequence in stream order -- and it is the plan's last (the mark is set first)
            if (lane == 0) S.c.plan_too_long = 1;
            wg_fence();
            if (lane == 0) flag_store(&S.c.plan_prog, chunk + 1);
            return kPlanBlockTooLong;
        }
        opos += chunk_tot;
        lpos += chunk_lit;
    }
    const uint32_t rest = cx.nlit - lpos;
    wg_fence();
    if (lane == 0) {
        S.c.rep_op[0] = R.s; S.c.rep_op[1] = (uint32_t)R.v0; S.c.rep_op[2] = (uint32_t)R.v1; S.c.rep_op[3] = (uint32_t)R.v2;
        S.c.plan_lit_used = lpos;
        if (opos + rest > kBlockMax) S.c.plan_too_long = 2; // only the literals after the last sequence pass the limit: every chunk is published
        flag_store(&S.c.plan_prog, chunk);
    }
    return opos + rest > kBlockMax ? kPlanBlockTooLong : 0;
}

struct CopyCtx {
    const uint4* plan;       // the block's plan (HBM): {ll, ml, off, output offset inside the chunk} per sequence
    uint8_t* dst;            // the file's output buffer
    uint64_t frame_start;    // offset of the current frame's first byte in dst
    const uint8_t* dict;     // dictionary content (logically just before frame_start) or null
    uint32_t dict_len;
    const uint8_t* lit;      // literal buffer of the block
    uint32_t nlit;
    uint64_t cap;            // capacity of dst
    uint32_t lit_streams;    // Huffman streams the literals arrive in (0: all literals are there from the start)
    uint32_t rep[3];         // the repeat offsets at the start of the block (the plan may refer to them)
    uint4* plan_wb;          // debug view only: resolved offsets are written back to the plan (else null)
};

// The copying half of K5, by one wavefront.  It publishes the finished output position in S.c.exec_pos
// for the hashing wavefront.
//
// Unit of work: a RUN = consecutive short sequences (<= kShort literal bytes and match bytes each) whose
// output fits one LDS staging buffer (kStage bytes); long sequences are copied straight to HBM by all
// 64 lanes.  A run is assembled in LDS and flushed with coalesced 16-byte stores.  Where a match's
// source lives, relative to the run being assembled:
//     inside the run ............ resolved LDS -> LDS in rounds (ready when the source lies below the
//                                 output of the first unfinished sequence)
//     in the previous two runs .. their staging buffers are still in LDS (three buffers rotate), so
//                                 it never matters whether their flushes have landed
//     older ..................... HBM.  Every flush first waits for the flush before it, hence all
//                                 output older than the previous two runs has landed.
// The HBM reads of a run (its literals and its old matches) are issued one run AHEAD (software
// pipeline: prepare(run k+1), then finish(run k)), so their latency hides behind the LDS work.
// Literal runs of 65..~2000 bytes inside a staged run: one after the other, all 64 lanes copy 16 bytes each from the
// literal buffer (HBM) into the staging buffer (LDS; not 16-byte aligned in general: two 8-byte stores per lane).
// Out of line: its registers must not count against the copier's main loop.
__device__ __noinline__ void medium_literals(const uint8_t* lit, uint8_t* sb, uint32_t ll, uint32_t my_lit, uint32_t rel_out, int lane) {
    uint64_t med = __ballot(ll > kShort);
    while (med) {
        const int sl = __builtin_ctzll(med);
        const uint32_t n = __builtin_amdgcn_readlane(ll, sl), lp = __builtin_amdgcn_readlane(my_lit, sl), ro = __builtin_amdgcn_readlane(rel_out, sl);
        const uint8_t* const src_ = lit + lp;
        lds_p const dst_ = (lds_p)(sb + ro);
        for (uint32_t k = (uint32_t)lane * 16; k + 16 <= n; k += 1024) {
            uint64_t v0, v1;
            __builtin_memcpy(&v0, (gcptr)(src_ + k), 8);
            __builtin_memcpy(&v1, (gcptr)(src_ + k + 8), 8);
            __builtin_memcpy(dst_ + k, &v0, 8);
            __builtin_memcpy(dst_ + k + 8, &v1, 8);
        }
        const uint32_t t0 = n & ~15u;
        if (t0 + (uint32_t)lane < n) dst_[t0 + lane] = *(gcptr)(src_ + t0 + lane);
        med &= med - 1;
    }
}

struct RunRegs { // one lane's share of a prepared run (kept small: two of these are live in the copier's loop)
    uint32_t ll, ml, rel_out;       // ll = ml = 0 on lanes outside the run
    int32_t rel_src;                // match source relative to the run start (the offset is rel_out + ll - rel_src)
    uint32_t meta;                  // bits 0-2 kind: 0 none, 1 LDS (this run or the two before it), 4 HBM (prefetched), 5 HBM (> 31 bytes, loaded at finish)
                                    // bit 3: kind 1 byte by byte (overlapping match, or a source that straddles buffers); bits 4..: kind 1, plain: byte offset of the source in S.stage
    int32_t ready_at;               // kind 1: run-relative output position that must be complete first
    uint32_t my_lit;
    __device__ __forceinline__ uint32_t kind() const { return meta & 7; }
    __device__ __forceinline__ bool bytewise() const { return (meta & 8) != 0; }
    __device__ __forceinline__ uint32_t src_lds() const { return meta >> 4; }
};
constexpr uint32_t kLitScratch = 1024;
struct RunInfo { // wave-uniform
    uint64_t run_pos; uint32_t T, buf; bool bigl;
    uint32_t lit0;   // the run's literals: one contiguous piece of the literal buffer starting here ...
    bool lit_pre;    // ... of at most kLitScratch bytes: prefetched by a coalesced load (16 bytes per lane) and dealt out through LDS
    bool v1, v2; uint32_t T1, T2, buf1, buf2; // the two runs before it
};

// Errors of the execute stage are reported the way the reference finds them: it decodes ALL sequences of a block (and its
// literals) before it executes any, and then takes the sequences in order, each checked against the destination's end,
// then the 128 KiB block limit, then its offset.  Both functions run once, after the copier's loop (cold code).
// A verdict of the copying wavefront waits until the walker and the literal decoders have theirs (a corrupt bitstream
// wins: it is posted first) ...
__device__ __noinline__ int exec_verdict(int rc, uint32_t nseq, uint32_t lit_streams) {
    uint32_t it = 0;
    for (; it < (1u << 24); it++) {
        const bool walked = !nseq || (flag_load(&S.c.walk_prog) & kWalkFin) != 0;
        const bool lits = !lit_streams || __atomic_load_n(&S.c.streams_done, __ATOMIC_RELAXED) >= lit_streams;
        if ((walked && lits) || __atomic_load_n(&S.c.err, __ATOMIC_RELAXED)) break;
        __builtin_amdgcn_s_sleep(4);
    }
    if (it == (1u << 24)) post_err(&S.c.err, MZD_E_DEVICE);
    return rc;
}
// ... and inside the chunk that cannot be executed (sequences base .. base+63 of the plan, read again here) the earliest
// offending sequence decides.  room / blk_room: bytes left in the destination / under the block limit at the chunk's
// start; hist: output of the frame + dictionary bytes before the chunk; rep: the block's starting repeat offsets.
__device__ __noinline__ int chunk_verdict(const uint4* plan, uint32_t base, int lane, uint64_t room, uint32_t blk_room, uint64_t hist, uint32_t lit_room,
                                          uint32_t rep0, uint32_t rep1, uint32_t rep2, uint32_t nseq, uint32_t lit_streams) {
    const bool valid = base + (uint32_t)lane < nseq;
    const uint4 pe = valid ? plan[base + (uint32_t)lane] : make_uint4(0, 0, 0, 0);
    uint32_t off = pe.z;
    if (off & kOffTag) off = (uint32_t)sel3((off >> 29) & 3, (int32_t)rep0, (int32_t)rep1, (int32_t)rep2) + (off & 0x1FFFFFFFu) - (uint32_t)kOffBias; // (as in copy_wave)
    const uint32_t ll = pe.x, ml = pe.y, ex_t = pe.w, incl_t = ex_t + ll + ml;
    // per sequence the reference checks: literals left (lit_room: literals not yet used at the chunk's start), destination's
    // end, block limit, offset -- "destination too small" only if nothing before it in that order is wrong
    const uint64_t nolit = __ballot(valid && wave_incl_scan(ll, lane) > lit_room);
    const uint64_t over = __ballot(valid && incl_t > room);
    const uint64_t bad = __ballot(valid && (incl_t > blk_room || off == 0 || off > hist + ex_t + ll));
    const int fl = nolit ? __builtin_ctzll(nolit) : 64, fo = over ? __builtin_ctzll(over) : 64, fb = bad ? __builtin_ctzll(bad) : 64;
    // (none of the three: the plan ended here without a sequence of this chunk being at fault, which cannot happen; corrupt)
    return exec_verdict(fo < fl && fo <= fb ? MZD_E_DSTSIZE : MZD_E_CORRUPT, nseq, lit_streams);
}

__device__ __noinline__ int copy_wave(uint32_t nseq_in, const CopyCtx& cx, uint64_t* opos_io, int lane) {
    const uint32_t nseq = __builtin_amdgcn_readfirstlane(nseq_in);
    // the context lives in the caller's frame (scratch memory): what the loops use is read once, into scalar
    // registers (wave-uniform; the vector registers are all taken); the rare paths read the rest where they need it
    auto u32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto u64 = [&](uint64_t v) -> uint64_t { return (uint64_t)u32((uint32_t)v) | ((uint64_t)u32((uint32_t)(v >> 32)) << 32); };
    uint8_t* const dst = (uint8_t*)(uintptr_t)u64((uint64_t)(uintptr_t)cx.dst);
    const uint8_t* const lit = (const uint8_t*)(uintptr_t)u64((uint64_t)(uintptr_t)cx.lit);
    const uint64_t cap = u64(cx.cap), frame_start = u64(cx.frame_start);
    const uint32_t dict_len = u32(cx.dict_len), nlit_all = u32(cx.nlit), lit_streams = u32(cx.lit_streams);
    const uint4* const plan = (const uint4*)(uintptr_t)u64((uint64_t)(uintptr_t)cx.plan);
    const uint8_t* const dict_end = (const uint8_t*)(uintptr_t)u64((uint64_t)(uintptr_t)cx.dict + cx.dict_len); // one past the dictionary content (or null)
    // where an old match's bytes are: in the output, or -- before the frame start -- in the dictionary
    auto match_src = [&](int32_t rel_src, uint64_t run_pos) -> const uint8_t* {
        const int64_t at = (int64_t)run_pos + rel_src - (int64_t)frame_start; // relative to the frame start
        return at >= 0 ? dst + frame_start + at : dict_end + at;
    };
    uint64_t opos = *opos_io;
    uint32_t lpos = 0;
    CSTAMP_DECL;
    auto wait_plan = [&](uint32_t nchunks_needed) -> bool { // true when that many chunks are planned
        uint32_t pg = 0, it = 0;
        for (; it < (1u << 24); it++) {
            pg = flag_load(&S.c.plan_prog);
            if ((pg & ~kPlanFin) >= nchunks_needed || (pg & kPlanFin)) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (it == (1u << 24)) post_err(&S.c.err, MZD_E_DEVICE);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return (pg & ~kPlanFin) >= nchunks_needed;
    };
    // literals become available stream by stream (in order: stream k fills [s_out[k], s_out[k] + s_n[k]))
    uint32_t lit_avail = lit_streams ? 0u : nlit_all;
    auto wait_lits = [&](uint32_t need) -> bool {
        if (need <= lit_avail) return true;
        if (need > nlit_all) return false; // more literals than the block has (the caller tells the two failures apart)
        for (uint32_t it = 0; it < (1u << 24); it++) {
            const uint32_t m = flag_load(&S.c.streams_mask);
            const uint32_t k = (uint32_t)__builtin_ctz(~m); // first stream not decoded yet
            lit_avail = k >= lit_streams ? nlit_all : S.c.s_out[k];
            if (need <= lit_avail) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return true; }
            if (__atomic_load_n(&S.c.err, __ATOMIC_RELAXED)) return false;
            __builtin_amdgcn_s_sleep(4);
        }
        post_err(&S.c.err, MZD_E_DEVICE);
        return false;
    };
    constexpr uint32_t kBufStride = kStage + 16;
    auto stagebuf = [&](uint32_t k) -> uint8_t* { return S.stage + k * kBufStride; };
    const uint32_t nchunks = (nseq + 63) / 64;

    // history: the two staged runs before the one being prepared (h1 most recent)
    bool v1 = false, v2 = false;
    uint32_t T1 = 0, T2 = 0, runno = 0;
    RunRegs R;  RunInfo RI;  bool haveR = false; // the prepared, unfinished run
    R.ll = R.ml = 0; R.meta = 0;
    // The prefetched HBM bytes of a run (its literals and its old match bytes, <= 31 each per lane).  One set is
    // enough: the loop stores run k's bytes to LDS (finish_regs), THEN issues run k+1's loads into the same
    // registers, and only then does the long part of run k (rounds, flush), which hides the loads' latency.
    // (The compiler waits with vmcnt(0) wherever the number of loads in flight depends on control flow, so
    // nothing else may be outstanding at the point where the registers are consumed.)
    uint4 pfC = make_uint4(0, 0, 0, 0); // literals: the run's whole piece of the literal buffer, 16 bytes per lane (runs with more than
                                        // kLitScratch literal bytes read theirs straight from HBM when the run is finished)
    static_assert(offsetof(Shared, wtab) == offsetof(Shared, wnorm) + 512 && offsetof(Shared, weights) == offsetof(Shared, wnorm) + 768 && offsetof(Shared, wnorm) % 16 == 0, "the literal scratch");
    uint8_t* const lscr = reinterpret_cast<uint8_t*>(S.wnorm);
    CopyRegs<3> pfO; // old match bytes: <= 31 per lane

    // finishing a prepared run, part 1: the prefetched bytes (literals, old matches) go to the staging buffer
    auto finish_regs = [&](RunRegs& r, const RunInfo& ri) {
        uint8_t* const sb = stagebuf(ri.buf);
        CSTAMP(2);
        if (ri.lit_pre) { // the prefetched piece goes to the scratch as it is; every lane then takes its own literals out of it
            *reinterpret_cast<uint4*>(lscr + (uint32_t)lane * 16) = pfC;
            copy_short(r.ll <= kShort ? r.ll : 0u, LdsLd{lscr + (r.my_lit - ri.lit0)}, LdsSt{sb + r.rel_out});
        } else // more literal bytes than the scratch holds: up to 64 bytes per lane straight from HBM
            copy_short(r.ll <= kShort ? r.ll : 0u, GlobalLd{lit + r.my_lit}, LdsSt{sb + r.rel_out});
        if (ri.bigl) medium_literals(lit, sb, r.ll, r.my_lit, r.rel_out, lane); // literal runs of 65..~2000 bytes (noisy data), one after the other, by all 64 lanes
        regs_store<3>(r.kind() == 4 ? r.ml : 0u, LdsSt{sb + r.rel_out + r.ll}, pfO);
        CSTAMP(3);
    };
    // part 2: LDS -> LDS copies in rounds, flush
    auto finish_rest = [&](RunRegs& r, const RunInfo& ri) {
        uint8_t* const sb = stagebuf(ri.buf);
        const uint8_t* const b1 = stagebuf(ri.buf1);
        const uint8_t* const b2 = stagebuf(ri.buf2);
        const uint32_t rel_m = r.rel_out + r.ll;
        if (__any(r.kind() == 5)) copy_short(r.kind() == 5 ? r.ml : 0u, GlobalLd{match_src(r.rel_src, ri.run_pos)}, LdsSt{sb + rel_m});
        CSTAMP(4);
        // everything whose source is in LDS, in rounds: a copy may start once the output below `ready_at` is complete,
        // and the output is complete up to the match of the first sequence that is still pending
        bool pending = r.kind() == 1;
        uint64_t pm = __ballot(pending);
        while (pm) {
            const int first = __builtin_ctzll(pm);
            const int32_t hwm = (int32_t)__builtin_amdgcn_readlane(rel_m, first);
            const bool ready = pending && r.ready_at <= hwm;
            const bool fast = ready && !r.bytewise();
            copy_short(fast ? r.ml : 0u, LdsLd{S.stage + r.src_lds()}, LdsSt{sb + rel_m});
            if (__any(ready && r.bytewise())) {
                if (ready && r.bytewise()) {
                    const uint32_t off_ = rel_m - (uint32_t)r.rel_src;
                    uint32_t idx = 0;
                    for (uint32_t k = 0; k < r.ml; k++) {
                        const int32_t p = r.rel_src + (int32_t)idx;
                        const int32_t d = -p;
                        uint8_t bv; // typed loads: hipcc 7.2 miscompiles a load through a pointer selected between HBM and LDS
                        if (p >= 0) bv = *(const __attribute__((address_space(3))) uint8_t*)(sb + p);
                        else if (ri.v1 && d <= (int32_t)ri.T1) bv = *(const __attribute__((address_space(3))) uint8_t*)(b1 + ((int32_t)ri.T1 - d));
                        else if (ri.v1 && ri.v2 && d <= (int32_t)(ri.T1 + ri.T2)) bv = *(const __attribute__((address_space(3))) uint8_t*)(b2 + ((int32_t)(ri.T1 + ri.T2) - d));
                        else bv = *(const __attribute__((address_space(1))) uint8_t*)(dst + ri.run_pos + p);
                        sb[rel_m + k] = bv;
                        idx++;
                        if (idx == off_) idx = 0;
                    }
                }
            }
            pending = pending && !ready;
            pm = __ballot(pending);
        }
        // flush: LDS -> HBM, 16 bytes per lane.  First wait for the previous flush (and whatever else is in flight).
        CSTAMP(5);
        wg_fence();
        CSTAMP(6);
        if (lane == 0) __atomic_store_n(&S.c.exec_pos, ri.run_pos, __ATOMIC_RELAXED); // everything before this run has landed
        uint8_t* g = dst + ri.run_pos;
        for (uint32_t k = (uint32_t)lane * 16; k + 16 <= ri.T; k += 1024) {
            uint4 v = *reinterpret_cast<const uint4*>(sb + k);
            __builtin_memcpy(g + k, &v, 16);
        }
        const uint32_t tail0 = ri.T & ~15u; // the last partial 16 bytes: one byte per lane
        if (tail0 + (uint32_t)lane < ri.T) g[tail0 + lane] = sb[tail0 + lane];
        CSTAMP(7);
    };

    uint4 pe_next = make_uint4(0, 0, 0, 0);
    if (nseq) {
        if (!wait_plan(1)) return MZD_E_CORRUPT; // the planner failed and posted the error
        if ((uint32_t)lane < nseq) pe_next = plan[lane];
    }
    uint32_t chunk = 0;
    uint32_t blk_room = kBlockMax; // bytes left under the block limit
    uint32_t tbase = 0xFFFFFFFFu; // the chunk that cannot be executed (see chunk_verdict)
    for (uint32_t base = 0; base < nseq; base += 64, chunk++) {
        const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
        const uint4 pe = pe_next; // loaded an iteration ago
        CSTAMP(1);
        bool cut = false; // the plan ends with this chunk (the block's output passes 128 KiB in it)
        if (chunk + 1 < nchunks) { // prefetch the next chunk's plan
            if (wait_plan(chunk + 2)) {
                CSTAMP(0);
                const uint32_t j = base + 64 + (uint32_t)lane;
                pe_next = j < nseq ? plan[j] : make_uint4(0, 0, 0, 0);
            } else if (__atomic_load_n(&S.c.plan_too_long, __ATOMIC_RELAXED) == 1) cut = true;
            else return MZD_E_CORRUPT;
        }
        const bool valid = (uint32_t)lane < cnt;
        const uint32_t ll = valid ? pe.x : 0, ml = valid ? pe.y : 0, ex_t = pe.w;
        uint32_t off = pe.z;
        if (off & kOffTag) { // an offset left symbolic by the planner: start slot + delta
            const uint32_t slot = (off >> 29) & 3;
            off = (uint32_t)sel3(slot, (int32_t)cx.rep[0], (int32_t)cx.rep[1], (int32_t)cx.rep[2]) + (off & 0x1FFFFFFFu) - (uint32_t)kOffBias;
            if (cx.plan_wb && valid) cx.plan_wb[base + (uint32_t)lane].z = off;
        }
        const uint32_t incl_t = ex_t + ll + ml;
        const uint32_t chunk_tot = __builtin_amdgcn_readlane(incl_t, cnt - 1);
        if (chunk_tot > cap - opos || chunk_tot > blk_room || cut ||
            __any(valid && (off == 0 || off > (opos + ex_t + ll - frame_start) + dict_len))) { // (beyond the window's history)
            tbase = base;
            break;
        }
        blk_room -= chunk_tot;
        const uint32_t incl_l = wave_incl_scan(ll, lane);
        const uint32_t my_lit = lpos + (incl_l - ll);
        lpos += __builtin_amdgcn_readlane(incl_l, 63);
        if (__builtin_expect(!wait_lits(lpos), 0)) {
            if (lpos <= nlit_all) return MZD_E_CORRUPT; // a literal stream failed (the error is posted)
            lpos -= __builtin_amdgcn_readlane(incl_l, 63); // the literals run out inside this chunk (the plan's last: see plan_wave)
            tbase = base;
            break;
        }
        const uint64_t mdst = opos + ex_t + ll; // absolute match destination
        // a match that starts before the frame reads the dictionary (config 5: most matches of a small record do).  When
        // its whole source lies there it is an ordinary old match with another base address; one that runs from the
        // dictionary into the output takes the long path.
        const bool in_dict = valid && off > mdst - frame_start;
        const bool dict_whole = in_dict && off - (mdst - frame_start) >= ml;
        // literal runs of up to ~2 KiB stay inside a run (the wavefront copies them into the staging buffer together);
        // only longer ones, long matches and matches that leave the dictionary go the direct way
        const bool islong = valid && (ll > kStage - kShort || ml > kShort || (in_dict && !dict_whole));
        const uint64_t longmask = __ballot(islong);

        uint32_t a = 0;
        while (a < cnt) {
            const uint32_t base_t = __builtin_amdgcn_readlane(ex_t, a);
            const uint64_t run_pos = opos + base_t; // absolute output position of lane a's literals
            if ((longmask >> a) & 1) { // a long sequence: drain the pipeline, then all 64 lanes copy it straight to HBM
                if (haveR) { finish_regs(R, RI); finish_rest(R, RI); haveR = false; }
                const uint32_t l = __builtin_amdgcn_readlane(ll, a), m = __builtin_amdgcn_readlane(ml, a);
                const uint32_t o = __builtin_amdgcn_readlane(off, a), lp = __builtin_amdgcn_readlane(my_lit, a);
                wave_copy(dst + run_pos, lit + lp, l, lane); // literals do not depend on earlier output: no fence in front
                wg_fence();                                   // everything so far (flushes and these literals) has landed
                if (lane == 0) __atomic_store_n(&S.c.exec_pos, run_pos + l, __ATOMIC_RELAXED);
                uint8_t* d = dst + run_pos + l;
                const uint64_t have = run_pos + l - frame_start;
                if (o > have) { // starts inside the dictionary: owner lane, sequential semantics
                    if ((uint32_t)lane == a) {
                        uint64_t back = o - have;
                        const uint8_t* dp = cx.dict + dict_len - back;
                        uint32_t k = 0;
                        for (; k < m && k < back; k++) d[k] = dp[k];
                        for (; k < m; k++) d[k] = dst[frame_start + (k - back)];
                    }
                } else if (o >= m) wave_copy(d, d - o, m, lane);
                else wave_pattern(d, o, m, lane);
                if (m) wg_fence();
                v1 = v2 = false; // nothing older is in LDS any more; all of it has landed in HBM
                a++;
                continue;
            }
            // ---- prepare run [a, b): classify, issue its HBM loads
            const uint64_t stop = __ballot(valid && (uint32_t)lane > a && (islong || incl_t - base_t > kStage));
            const uint32_t b = stop ? (uint32_t)__builtin_ctzll(stop) : cnt;
            RunRegs N; RunInfo NI;
            NI.run_pos = run_pos;
            NI.T = __builtin_amdgcn_readlane(incl_t, b - 1) - base_t;
            NI.buf = runno % 3; NI.buf1 = (runno + 2) % 3; NI.buf2 = (runno + 1) % 3;
            NI.v1 = v1; NI.v2 = v2; NI.T1 = T1; NI.T2 = T2;
            const bool act = (uint32_t)lane >= a && (uint32_t)lane < b;
            N.ll = act ? ll : 0; N.ml = act ? ml : 0; N.rel_out = ex_t - base_t; N.my_lit = my_lit;
            const uint32_t rel_m = N.rel_out + N.ll;
            N.rel_src = (int32_t)rel_m - (int32_t)off; // off < 2^31 (validated against the window by the planner)
            uint32_t kind = 0, src_lds = 0; bool bytewise = false;
            // a copy from LDS may start once the output below source start + min(ml, off) is complete
            // (never positive for sources that lie entirely in the two previous runs)
            N.ready_at = N.rel_src + (int32_t)(N.ml < off ? N.ml : off);
            if (N.ml) {
                const bool plain = off >= N.ml;
                const int32_t pd = -N.rel_src;          // distance of the source start before the run start
                const int32_t pe_ = pd - (int32_t)N.ml; // distance of the source end before the run start (>= 0: entirely older)
                const int32_t lim1 = v1 ? (int32_t)T1 : 0, lim2 = lim1 + ((v1 && v2) ? (int32_t)T2 : 0);
                kind = 1;
                if (!plain) bytewise = true;                                                               // overlapping: replicate byte by byte
                else if (dict_whole) kind = N.ml > 31 ? 5 : 4;                                             // in the dictionary: HBM, like older output
                else if (N.rel_src >= 0) src_lds = NI.buf * kBufStride + (uint32_t)N.rel_src;              // inside this run
                else if (pe_ < 0) bytewise = true;                                                         // straddles the run start
                else if (v1 && pd <= lim1) src_lds = NI.buf1 * kBufStride + (uint32_t)(lim1 - pd);         // inside the previous run
                else if (v1 && v2 && pe_ >= lim1 && pd <= lim2) src_lds = NI.buf2 * kBufStride + (uint32_t)(lim2 - pd); // inside the run before it
                else if (pe_ >= lim2 && run_pos - (uint64_t)pe_ + 8 <= cap) kind = N.ml > 31 ? 5 : 4;   // older: HBM (may over-read 7 bytes)
                else bytewise = true;                                                                       // straddles two buffers / ends at the buffer end
            }
            N.meta = kind | (bytewise ? 8u : 0u) | (src_lds << 4);
            NI.bigl = __any(N.ll > kShort);
            NI.lit0 = __builtin_amdgcn_readlane(my_lit, a);
            const uint32_t lit_bytes = __builtin_amdgcn_readlane(my_lit + ll, b - 1) - NI.lit0; // (lanes a .. b-1 are valid: their literals are consecutive)
            NI.lit_pre = lit_bytes <= kLitScratch;
            // ---- the previous run's prefetched bytes leave the registers; this run's loads take their place and
            //      stay in flight during the long part of the previous run
            if (haveR) finish_regs(R, RI);
            if (NI.lit_pre && (uint32_t)lane * 16 < lit_bytes) __builtin_memcpy(&pfC, (gcptr)(lit + NI.lit0 + (uint32_t)lane * 16), 16); // (may read up to 15 bytes past the piece: padded buffers)
            regs_load<3>(N.kind() == 4 ? N.ml : 0u, GlobalLd{match_src(N.rel_src, run_pos)}, pfO);
            if (haveR) finish_rest(R, RI);
            R = N; RI = NI; haveR = true;
            v2 = v1; T2 = T1; v1 = true; T1 = NI.T; runno++;
            a = b;
        }
        opos += chunk_tot;
    }
    if (tbase != 0xFFFFFFFFu) return chunk_verdict(plan, tbase, lane, cap - opos, blk_room, (opos - frame_start) + dict_len, nlit_all - lpos, cx.rep[0], cx.rep[1], cx.rep[2], nseq, lit_streams);
    if (haveR) { finish_regs(R, RI); finish_rest(R, RI); }
    // the literals after the last sequence: the planner has validated them once it is finished
    if (nseq) {
        uint32_t it = 0;
        for (; it < (1u << 24); it++) {
            if (flag_load(&S.c.plan_prog) & kPlanFin) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (it == (1u << 24)) post_err(&S.c.err, MZD_E_DEVICE);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (__atomic_load_n(&S.c.err, __ATOMIC_RELAXED)) return MZD_E_CORRUPT;
        if (__atomic_load_n(&S.c.plan_too_long, __ATOMIC_RELAXED) == 2) // only the literals after the last sequence pass the block limit: the destination's end comes first
            return exec_verdict(cap - *opos_io <= kBlockMax ? MZD_E_DSTSIZE : MZD_E_CORRUPT, nseq, lit_streams);
        if (lpos != __atomic_load_n(&S.c.plan_lit_used, __ATOMIC_RELAXED)) return MZD_E_CORRUPT;
        if (nlit_all - lpos > cap - opos) return MZD_E_DSTSIZE;
    } else {
        if (nlit_all > kBlockMax) return MZD_E_CORRUPT;
        if (nlit_all > cap - opos) return MZD_E_DSTSIZE; // a block without sequences has no planner to check this
    }
    const uint32_t rest = nlit_all - lpos;
    if (!wait_lits(nlit_all)) return MZD_E_CORRUPT;
    if (lit + lpos != dst + opos) wave_copy(dst + opos, lit + lpos, rest, lane); // (literal-only block decoded in place: nothing to move)
    opos += rest;
    wg_fence();
    if (lane == 0) __atomic_store_n(&S.c.exec_pos, opos, __ATOMIC_RELAXED);
    *opos_io = opos;
    return 0;
}

// ------------------------------------------------------------------------------------ K7
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t h, uint64_t v) { v = xround(0, v); h ^= v; return h * XP1 + XP4; }

// XXH64(seed 0) by one wavefront, incrementally: lanes 0..3 carry the four accumulators `v`;
// `stripes` counts the 32-byte stripes already absorbed.  The hashing wavefront follows the
// executing one through the frame (xxh_advance up to the published output position) and closes
// the digest at the frame end (xxh_finish).
__device__ __forceinline__ uint64_t xxh_init(int lane) {
    const int l = lane & 3; // every group of four lanes carries the same four accumulators
    return l == 0 ? XP1 + XP2 : (l == 1 ? XP2 : (l == 2 ? 0ull : 0ull - XP1));
}
// rotl by 31 as two funnel shifts ({lo,hi} >> 1 and {hi,lo} >> 1)
__device__ __forceinline__ uint64_t rotl64_31(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    return (uint64_t)__builtin_amdgcn_alignbit(lo, hi, 1) | ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 1) << 32);
}
__device__ __forceinline__ uint64_t xround31(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64_31(acc); return acc * XP1; }
// acc + (`t` of the lane 4*J further up in the row of 16): DPP row_shl on the addend, folded into the two halves of the
// 64-bit add (lanes past the row's end add 0; `t` was written several instructions earlier -- the chain step in between
// -- which covers the two wait states a DPP read needs after a VALU write)
template <int J>
__device__ __forceinline__ uint64_t add_row_up(uint64_t acc, uint64_t t) {
    if (J == 0) return acc + t;
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    const uint32_t tlo = (uint32_t)t, thi = (uint32_t)(t >> 32);
    if (J == 1) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    if (J == 2) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    if (J == 3) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
template <int J>
__device__ __forceinline__ uint64_t xchain31(uint64_t acc, uint64_t t) { acc = add_row_up<J>(acc, t); acc = rotl64_31(acc); return acc * XP1; }
__device__ __noinline__ void xxh_advance(uint64_t& v, uint64_t& stripes, uint64_t upto, const uint8_t* p, int lane) {
    if (upto <= stripes) return;
#ifdef MZD_EXP_NOHASH
    stripes = upto; return;
#endif
    { // all 64 lanes run (four copies of a row of 16): no divergent region around the loop
        // Per stripe and accumulator: acc = rotl31(acc + in * P2) * P1.  The product in * P2 is not part of the serial
        // chain, so a row of 16 lanes computes it for FOUR stripes at once (lane 4 s + a: stripe s, accumulator a --
        // 128 contiguous bytes); the chain itself runs in the row's lanes 0..3, which pick the products of stripes
        // 1..3 out of the lanes above them (DPP row shifts folded into the adds).  That is one 64-bit multiply per
        // stripe on the chain instead of two (integer multiplies are quarter rate, and this wavefront shares its SIMD
        // with another file's walker).  Groups of 32 stripes with no per-stripe bounds checks, the next group's loads
        // in flight while the current one is absorbed, two register sets used alternately (no hand-over copies).
        gcptr q = (gcptr)(p + (lane & 15) * 8 + stripes * 32);
        uint64_t n = upto - stripes;
        uint64_t acc = v;
        uint64_t A[8], B[8];
        auto load8 = [&](uint64_t (&r)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) __builtin_memcpy(&r[k], q + k * 128, 8);
            q += 1024; n -= 32;
        };
        auto absorb4 = [&](uint64_t in) { // four stripes
            const uint64_t t = in * XP2;
            acc = xchain31<0>(acc, t);
            acc = xchain31<1>(acc, t);
            acc = xchain31<2>(acc, t);
            acc = xchain31<3>(acc, t);
        };
        auto absorb8 = [&](const uint64_t (&r)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) absorb4(r[k]);
        };
        if (n >= 32) {
            load8(A);
            for (;;) {
                if (n < 32) { absorb8(A); break; }
                load8(B);
                absorb8(A);
                if (n < 32) { absorb8(B); break; }
                load8(A);
                absorb8(B);
            }
        }
        // fewer than 32 stripes left: whole groups of four, then (only at the end of a frame: the follower advances in
        // groups of 8) the last one to three stripes.
        const uint32_t g4 = (uint32_t)(n >> 2), rest = (uint32_t)(n & 3); // (wave-uniform)
#pragma unroll
        for (int k = 0; k < 8; k++) if ((uint32_t)k < g4 || ((uint32_t)k == g4 && (uint32_t)((lane & 15) >> 2) < rest)) __builtin_memcpy(&A[k], q + k * 128, 8); // (never past stripe `upto`)
#pragma unroll
        for (int k = 0; k < 8; k++) if ((uint32_t)k < g4) absorb4(A[k]);
        if (rest) {
            uint64_t in = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) if ((uint32_t)k == g4) in = A[k];
            const uint64_t t = in * XP2;
            acc = xchain31<0>(acc, t);
            if (rest > 1) acc = xchain31<1>(acc, t);
            if (rest > 2) acc = xchain31<2>(acc, t);
        }
        v = acc;
    }
    stripes = upto;
}
__device__ __noinline__ uint64_t xxh_finish(uint64_t v, const uint8_t* p, uint64_t n, int lane) {
    uint64_t h;
    if (n >= 32) {
        uint64_t v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = XP5;
    }
    h += n;
    const uint8_t* q = p + (n / 32) * 32;
    const uint8_t* end = p + n;
    while (q + 8 <= end) { h ^= xround(0, ld64(q)); h = rotl64(h, 27) * XP1 + XP4; q += 8; }
    if (q + 4 <= end) { h ^= (uint64_t)ld32(q) * XP1; h = rotl64(h, 23) * XP2 + XP3; q += 4; }
    while (q < end) { h ^= (uint64_t)(*q) * XP5; h = rotl64(h, 11) * XP1; q++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

// ------------------------------------------------------------------------------------ K0 + block driver
__device__ __noinline__ void parse_frame_or_skip(Ctl& c, const uint8_t* src, uint64_t n, const DevDict* dicts, uint32_t ndicts, uint32_t job_dict) {
    uint64_t pos = c.pos;
    if (pos >= n) { c.action = 2; return; }
    if (n - pos < 4) { c.err = MZD_E_TRUNCATED; return; }
    const uint8_t* p = src + pos;
    uint32_t magic = ld32(p);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
        if (n - pos < 8) { c.err = MZD_E_TRUNCATED; return; }
        uint64_t sz = ld32(p + 4);
        if (n - pos - 8 < sz) { c.err = MZD_E_TRUNCATED; return; }
        c.pos = pos + 8 + sz;
        c.action = 1;
        return;
    }
    if (magic != 0xFD2FB528u) { c.err = MZD_E_BADMAGIC; return; }
    if (n - pos < 5) { c.err = MZD_E_TRUNCATED; return; }
    uint32_t fhd = p[4];
    uint32_t fcsf = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    if (fhd & 0x08) { c.err = MZD_E_UNSUPPORTED; return; }
    uint32_t did_sz = did == 3 ? 4 : did, fcs_sz = fcsf == 0 ? single : (1u << fcsf);
    uint64_t hs = 5 + (single ? 0 : 1) + did_sz + fcs_sz;
    if (n - pos < hs) { c.err = MZD_E_TRUNCATED; return; }
    const uint8_t* q = p + 5;
    uint64_t window = 0;
    if (!single) { uint32_t b = *q++; uint32_t wl = 10 + (b >> 3); window = (1ull << wl) + ((1ull << wl) >> 3) * (b & 7); }
    uint32_t dict_id = 0;
    if (did == 1) { dict_id = q[0]; q += 1; } else if (did == 2) { dict_id = ld16(q); q += 2; } else if (did == 3) { dict_id = ld32(q); q += 4; }
    c.has_fcs = 1;
    if (fcsf == 0) { if (single) c.fcs = *q++; else { c.fcs = 0; c.has_fcs = 0; } }
    else if (fcsf == 1) { c.fcs = (uint64_t)ld16(q) + 256; }
    else if (fcsf == 2) { c.fcs = ld32(q); }
    else { c.fcs = ld64(q); }
    if (single) window = c.fcs;
    if (window > (1ull << 27) + 1) { c.err = MZD_E_UNSUPPORTED; return; } // copy_decode is a streaming decoder (windowLogMax 27)
    c.block_max = (uint32_t)(window < kBlockMax ? window : kBlockMax);
    c.has_cksum = (fhd >> 2) & 1;
    c.pos = pos + hs;
    c.frame_out0 = c.out;
    c.rep[0] = 1; c.rep[1] = 4; c.rep[2] = 8;
    c.huf_valid = 0; c.fse_valid = 0;
    c.dict_content = nullptr; c.dict_content_len = 0;
    c.action = 0;
    // dictionary
    const DevDict* dd = (job_dict >= 1 && job_dict <= ndicts) ? &dicts[job_dict - 1] : nullptr;
    // libzstd: a frame that names a dictionary fails unless exactly that dictionary is loaded
    if (dict_id && dict_id != (dd && dd->formatted ? dd->dict_id : 0u)) { c.err = MZD_E_DICT; return; }
    if (dd) c.action = 3; // frame with dictionary: tables are copied in by the workgroup
}

__device__ __noinline__ void parse_block_header(Ctl& c, const uint8_t* src, uint64_t n) {
    if (n - c.pos < 3) { c.err = MZD_E_TRUNCATED; return; }
    uint32_t bh = ld24(src + c.pos);
    c.pos += 3;
    c.last = bh & 1; c.btype = (bh >> 1) & 3; c.bsize = bh >> 3;
    if (c.btype == 3 || c.bsize > c.block_max) { c.err = MZD_E_CORRUPT; return; }
    uint64_t need = c.btype == 1 ? 1 : c.bsize;
    if (n - c.pos < need) { c.err = MZD_E_TRUNCATED; return; }
    if (c.btype == 2 && c.bsize < 2) { c.err = MZD_E_CORRUPT; return; }
}

// literals section header (+ Huffman weights).  Lane 0.
__device__ __noinline__ void parse_literals(Ctl& c, const uint8_t* b, uint32_t n) {
    uint32_t type = b[0] & 3, sf = (b[0] >> 2) & 3;
    uint32_t regen, comp = 0, hs, streams = 0;
    c.lit_type = type;
    c.lit_is_raw = 0;
    if (type < 2) {
        if (sf == 0 || sf == 2) { hs = 1; regen = b[0] >> 3; }
        else if (sf == 1) { if (n < 2) { c.err = MZD_E_CORRUPT; return; } hs = 2; regen = (b[0] >> 4) + ((uint32_t)b[1] << 4); }
        else { if (n < 3) { c.err = MZD_E_CORRUPT; return; } hs = 3; regen = (b[0] >> 4) + ((uint32_t)b[1] << 4) + ((uint32_t)b[2] << 12); }
        if (regen > c.block_max) { c.err = MZD_E_CORRUPT; return; }
        uint32_t body = type == 0 ? regen : 1;
        if (hs + body > n) { c.err = MZD_E_CORRUPT; return; }
        c.nlit = regen; c.streams = 0;
        c.lit_off = c.pos + hs;
        c.lit_is_raw = type == 0;
        c.seq_off = c.pos + hs + body;
        c.seq_len = n - hs - body;
        return;
    }
    if (n < 3) { c.err = MZD_E_CORRUPT; return; }
    if (sf == 0 || sf == 1) { hs = 3; uint32_t v = ld24(b); regen = (v >> 4) & 0x3FF; comp = v >> 14; streams = sf ? 4 : 1; }
    else if (sf == 2) { if (n < 4) { c.err = MZD_E_CORRUPT; return; } hs = 4; uint32_t v = ld32(b); regen = (v >> 4) & 0x3FFF; comp = v >> 18; streams = 4; }
    else { if (n < 5) { c.err = MZD_E_CORRUPT; return; } hs = 5; uint64_t v = (uint64_t)ld32(b) | ((uint64_t)b[4] << 32); regen = (uint32_t)(v >> 4) & 0x3FFFF; comp = (uint32_t)(v >> 22); streams = 4; }
    if (regen > c.block_max || regen == 0 || (streams == 4 && regen < 6) || hs + comp > n) { c.err = MZD_E_CORRUPT; return; }
    const uint8_t* p = b + hs;
    uint32_t rem = comp;
    if (type == 2) { // the tree is decoded later by another wavefront; here only its extent
        if (rem < 1) { c.err = MZD_E_CORRUPT; return; }
        uint32_t hb = p[0];
        uint32_t tl = hb >= 128 ? 1 + ((hb - 127) + 1) / 2 : 1 + hb;
        if (tl > rem) { c.err = MZD_E_CORRUPT; return; }
        c.huf_tree_off = (uint32_t)(p - b); c.huf_tree_len = tl;
        p += tl; rem -= tl;
    } else if (!c.huf_valid) { c.err = MZD_E_CORRUPT; return; }
    uint32_t base = (uint32_t)(p - b); // offset of the streams inside the block
    if (streams == 1) {
        c.s_off[0] = base; c.s_len[0] = rem; c.s_out[0] = 0; c.s_n[0] = regen;
    } else {
        if (rem < 10) { c.err = MZD_E_CORRUPT; return; }
        uint32_t l1 = ld16(p), l2 = ld16(p + 2), l3 = ld16(p + 4);
        if (6 + l1 + l2 + l3 > rem) { c.err = MZD_E_CORRUPT; return; }
        uint32_t l4 = rem - 6 - l1 - l2 - l3;
        uint32_t seg = (regen + 3) / 4;
        if (3 * seg > regen) { c.err = MZD_E_CORRUPT; return; }
        c.s_off[0] = base + 6; c.s_off[1] = c.s_off[0] + l1; c.s_off[2] = c.s_off[1] + l2; c.s_off[3] = c.s_off[2] + l3;
        c.s_len[0] = l1; c.s_len[1] = l2; c.s_len[2] = l3; c.s_len[3] = l4;
        c.s_out[0] = 0; c.s_out[1] = seg; c.s_out[2] = 2 * seg; c.s_out[3] = 3 * seg;
        c.s_n[0] = c.s_n[1] = c.s_n[2] = seg; c.s_n[3] = regen - 3 * seg;
    }
    c.nlit = regen; c.streams = streams;
    c.seq_off = c.pos + hs + comp;
    c.seq_len = n - hs - comp;
}

// sequences section header: nbSeq, modes, table descriptions.  Lane 0 of the walking wavefront, while other
// wavefronts already work on the literals (errors are posted first-wins).
// `stage_off`: where `b` lies inside S.stage (the normalized-count reader addresses the staging area by offset)
__device__ __noinline__ void parse_seq_header(Ctl& c, const uint8_t* b, uint32_t n, uint32_t stage_off) {
    if (n < 1) { post_err(&c.err, MZD_E_CORRUPT); return; }
    const uint8_t* p = b;
    const uint8_t* end = b + n;
    uint32_t nseq = *p++;
    if (nseq > 0x7F) {
        if (nseq == 0xFF) { if (p + 2 > end) { post_err(&c.err, MZD_E_CORRUPT); return; } nseq = ld16(p) + 0x7F00; p += 2; }
        else { if (p + 1 > end) { post_err(&c.err, MZD_E_CORRUPT); return; } nseq = ((nseq - 0x80) << 8) + *p++; }
    }
    c.nseq = nseq;
    if (nseq == 0) { if (p != end) post_err(&c.err, MZD_E_CORRUPT); return; }
    if (nseq > kMaxSeq - 1 || p + 1 > end) { post_err(&c.err, MZD_E_CORRUPT); return; }
    uint32_t modes = *p++;
    if (modes & 3) { post_err(&c.err, MZD_E_CORRUPT); return; }
    c.mode[0] = modes >> 6; c.mode[1] = (modes >> 4) & 3; c.mode[2] = (modes >> 2) & 3;
    const int max_log[3] = {9, 8, 9}, max_sym[3] = {35, 31, 52};
    for (int t = 0; t < 3; t++) {
        uint32_t m = c.mode[t];
        if (m == 1) {
            if (p + 1 > end || *p > max_sym[t]) { post_err(&c.err, MZD_E_CORRUPT); return; }
            c.nsym[t] = *p++; // the symbol itself
        } else if (m == 2) {
            // the header was staged at S.stage + 256 by the caller
            const uint32_t at = (stage_off & ~kInRing) + (uint32_t)(p - b);
            int used = (stage_off & kInRing) ? read_ncount_ring(at, (uint32_t)(end - p), max_log[t], max_sym[t], S.norm[t], &c.nsym[t], &c.al[t])
                                             : read_ncount_staged(at, (uint32_t)(end - p), max_log[t], max_sym[t], S.norm[t], &c.nsym[t], &c.al[t]);
            if (used <= 0) { post_err(&c.err, MZD_E_CORRUPT); return; }
            p += used;
        } else if (m == 3) {
            if (!c.fse_valid) { post_err(&c.err, MZD_E_CORRUPT); return; }
        }
    }
    c.seq_off += (uint64_t)(p - b);
    c.seq_len = (uint32_t)(end - p);
}

// The three sequence tables of a block, built one after the other by ONE wavefront.
__device__ __noinline__ void build_tables_wave(int lane) {
    Ctl& c = S.c;
    for (int t = 0; t < 3; t++) {
        uint64_t* tab = t == 0 ? S.ll : (t == 1 ? S.of : S.ml);
        const uint32_t m = c.mode[t];
        if (m == 0) {
            const int16_t* def = t == 0 ? LL_DEF : (t == 1 ? OF_DEF : ML_DEF);
            const uint32_t n = t == 0 ? 36 : (t == 1 ? 29 : 53), lg = t == 1 ? 5 : 6;
            if ((uint32_t)lane < n) S.norm[t][lane] = def[lane];
            build_seq_table_wave(tab, S.norm[t], n, lg, t, S.ring, lane);
            if (lane == 0) c.al[t] = lg;
        } else if (m == 1) {
            if (lane == 0) { rle_seq_table(tab, c.nsym[t], t); c.al[t] = 0; }
        } else if (m == 2) {
            build_seq_table_wave(tab, S.norm[t], c.nsym[t], c.al[t], t, S.ring, lane);
        }
    }
}

// Control words live in LDS and are written by lane 0 (or one lane per wavefront).  Every
// decision the workgroup takes on them is read through WG_SNAPSHOT: barrier, every lane copies
// the words it needs into registers, barrier -- so no lane can still be reading a word when the
// next step rewrites it, and all 256 lanes always take the same branch.
#define WG_SNAPSHOT(...) do { __syncthreads(); __VA_ARGS__; __syncthreads(); } while (0)


// The launch's queue: tickets are job indices, or -- behind the small-file kernel -- indices into the launch's job list
// (the host's part, then what that kernel handed on: KernelArgs::job_list).
__device__ __forceinline__ uint32_t queue_len(const KernelArgs& a) { return a.job_list ? a.nlist_fixed + __atomic_load_n(&a.counter[4], __ATOMIC_RELAXED) : a.njobs; }
__device__ __forceinline__ uint32_t queue_job(const KernelArgs& a, uint32_t ticket) { return a.job_list ? a.job_list[ticket] : ticket; }
__device__ __forceinline__ uint32_t take_job(const KernelArgs& a) { // one lane
    const uint32_t t = atomicAdd(&a.counter[0], 1u);
    return t < queue_len(a) ? queue_job(a, t) : kDoneJob;
}

// Driver 1, by the walking wavefront once its own work on a file's last block is done: take the next file and parse
// the headers of its first block (frame header, block header, literals header, sequence header with its three
// normalized-count descriptions: ~60 K cycles of serial parsing) into S.c2, so that the workgroup finds them ready
// when the copier and the hasher are through with the current file.  Only the plain case is prepared (one frame start,
// a compressed first block, no error); anything else leaves pre_valid = 0 and the file is parsed the normal way.
__device__ __noinline__ void pre_parse_next(const KernelArgs& a, int lane) {
    Ctl& c2 = S.c2;
    uint32_t j2 = 0;
    if (lane == 0) { j2 = take_job(a); S.pre_job = j2; S.pre_valid = 0; }
    j2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)j2);
    if (j2 >= a.njobs) return;
    const uint8_t* const src = a.jobs[j2].src;
    const uint64_t n = a.jobs[j2].src_len;
    const uint32_t job_dict = a.jobs[j2].dict;
    if (lane == 0) { S.pj.src = src; S.pj.n = n; S.pj.dst = a.jobs[j2].dst; S.pj.cap = a.jobs[j2].dst_cap; S.pj.dict = job_dict; }
    if (lane == 0) {
        c2.pos = 0; c2.out = 0; c2.err = 0; c2.action = 0; c2.btype = 0; c2.diag_slow = 0;
        if (job_dict > a.ndicts) c2.err = MZD_E_DICT;
        else parse_frame_or_skip(c2, src, n, a.dicts, a.ndicts, job_dict);
        if (!c2.err && (c2.action == 0 || c2.action == 3)) {
            if (c2.action == 3 && a.dicts[job_dict - 1].formatted) { c2.huf_valid = 1; c2.fse_valid = 1; } // (the tables themselves are loaded by the workgroup)
            parse_block_header(c2, src, n);
        } else if (!c2.err) c2.err = MZD_E_PARAM; // skippable frame / end of file: not prepared
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (c2.err || c2.btype != 2) return; // (wave-uniform: every lane reads the same words)
    const uint64_t pos0 = c2.pos;
    const uint32_t bsize = c2.bsize;
    uint8_t* const ps = S.ring + kPreStage;
    for (uint32_t k = (uint32_t)lane; k < bsize && k < 256; k += 64) ps[k] = src[pos0 + k];
    if (lane == 0) parse_literals(c2, ps, bsize);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (c2.err) return;
    const uint64_t seq_off = c2.seq_off;
    const uint32_t seq_len = c2.seq_len;
    for (uint32_t k = (uint32_t)lane; k < seq_len && k < 256; k += 64) ps[256 + k] = src[seq_off + k];
    if (lane == 0) parse_seq_header(c2, ps + 256, seq_len, kInRing | (kPreStage + 256));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (c2.err) return;
    // (Building the Huffman table ahead as well was tried and measured slower: the ~65 K cycles of weight decoding then
    //  queue behind this wavefront's own walk instead of running beside it on the copying wavefront.)
    if (lane == 0) S.pre_valid = 1;
}

// ---- driver 1: one workgroup decodes a whole file, block after block.  Used when no file of the launch can have more
// than one block (every output capacity <= 128 KiB): nothing is forked, nothing is published, the file's state
// stays in registers and LDS.
__global__ __launch_bounds__(kWG, 4) void mzd_decode_kernel_files(KernelArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t slot = a.wg0 + blockIdx.x; // this workgroup's place in the scratch arrays
    uint8_t* const lit_buf = a.lit_scratch + (size_t)slot * kLitStride;
    uint4* const seqs = a.seq_scratch + (size_t)slot * kSeqStride;
    uint4* const walk = a.walk_scratch + (size_t)slot * kSeqStride;
    Ctl& c = S.c;
    if (tid < 36) S.ll_base[tid] = LL_BASE[tid];
    if (tid < 53) S.ml_base[tid] = ML_BASE[tid];
    if (tid == 0) { c.lds_dict_fse = 0; c.lds_dict_huf = 0; S.pre_job = kNoJob; S.pre_valid = 0; S.dcache.id = 0; }

    for (;;) {
        TTASK();
        if (tid == 0) { // the file the walking wavefront took ahead (pre_parse_next), or the next one of the queue
            if (S.pre_job != kNoJob) { c.job = S.pre_job; c.t_valid = S.pre_valid | 2u; S.pre_job = kNoJob; S.pre_valid = 0; } // (bit 1: S.pj holds the job's table entry)
            else { c.job = take_job(a); c.t_valid = 0; }
        }
        uint32_t j, tv, lf = 0, lh = 0;
        WG_SNAPSHOT(j = c.job; tv = c.t_valid; lf = c.lds_dict_fse; lh = c.lds_dict_huf);
        bool pre = (tv & 1) != 0; // the first block's headers are already parsed (in S.c2)
        if (j >= a.njobs) break;
        const bool stashed = (tv & 2) != 0;
        const uint8_t* const src = stashed ? S.pj.src : a.jobs[j].src;
        const uint64_t n = stashed ? S.pj.n : a.jobs[j].src_len;
        uint8_t* const dst = stashed ? S.pj.dst : a.jobs[j].dst;
        const uint64_t cap = stashed ? S.pj.cap : a.jobs[j].dst_cap;
        const uint32_t job_dict = stashed ? S.pj.dict : a.jobs[j].dict;
        if (pre) { // the prepared control block replaces the current one: by all threads, a dword each (what must survive was read above)
            static_assert(sizeof(Ctl) % 4 == 0, "copied by dwords");
            for (uint32_t k = (uint32_t)tid; k < sizeof(Ctl) / 4; k += kWG) reinterpret_cast<uint32_t*>(&c)[k] = reinterpret_cast<const uint32_t*>(&S.c2)[k];
            __syncthreads();
        }
        if (tid == 0) {
            if (pre) { c.lds_dict_fse = lf; c.lds_dict_huf = lh; c.job = j; }
            else { c.pos = 0; c.out = 0; c.err = 0; c.action = 0; }
            c.diag_slow = 0;
#ifdef MZD_STAMPS
            for (int k_ = 0; k_ < 8; k_++) S.cdiag[k_] = 0;
#endif
            if (j == 0) a.counter[1] = a.wg0 + blockIdx.x;
            if (job_dict > a.ndicts) c.err = MZD_E_DICT;
        }
        int err = 0;
        uint32_t action = 0;
        uint64_t xv = 0, xstripes = 0; // K7 state of the hashing wavefront (wave 2)
        STAMP_DECL;

        // ---------------- frames (K0)
        while (true) {
            const bool frame_pre = pre; // (only the first frame of the file can have been prepared)
            if (tid == 0 && !c.err && !frame_pre) parse_frame_or_skip(c, src, n, a.dicts, a.ndicts, job_dict);
            WG_SNAPSHOT(err = c.err; action = c.action);
            if (err || action == 2) break;
            if (action == 1) continue; // skippable frame
            xv = xxh_init(lane); xstripes = 0;
            const bool hashing = c.has_cksum != 0; // stable for the whole frame
            if (action == 3) { // dictionary: entropy tables, repeat offsets and content
                const DevDict* dd = &a.dicts[job_dict - 1];
                if (tid == 0 && S.dcache.id != job_dict) {
                    S.dcache.formatted = dd->formatted; S.dcache.huf_log = dd->huf_log; S.dcache.content_len = dd->content_len; S.dcache.content = dd->content;
                    for (int t_ = 0; t_ < 3; t_++) { S.dcache.al[t_] = dd->al[t_]; S.dcache.rep[t_] = dd->rep[t_]; }
                    S.dcache.id = job_dict;
                }
                __syncthreads();
                if (S.dcache.formatted) {
                    // Config 5 (many small frames, one dictionary): a workgroup keeps the dictionary's tables resident in LDS
                    // from file to file -- such frames use them as they are (repeat-mode tables, treeless literals), so the
                    // 14 KB copy happens once per workgroup, not once per file.  Any block that rebuilds a table clears the mark.
                    const bool have_fse = c.lds_dict_fse == job_dict, have_huf = c.lds_dict_huf == job_dict;
                    __syncthreads();
                    if (!have_fse) {
                        for (int i = tid; i < 512; i += kWG) { S.ll[i] = dd->ll[i]; S.ml[i] = dd->ml[i]; }
                        for (int i = tid; i < 256; i += kWG) S.of[i] = dd->of[i];
                    }
                    if (!have_huf) for (int i = tid; i < 2048; i += kWG) S.huf[i] = dd->huf[i];
                    if (tid == 0) {
                        c.lds_dict_fse = job_dict; c.lds_dict_huf = job_dict;
                        // (a prepared first block has its sequence header parsed already: only repeat-mode tables take the dictionary's log)
                        for (int t_ = 0; t_ < 3; t_++) if (!frame_pre || c.mode[t_] == 3) c.al[t_] = S.dcache.al[t_];
                        c.huf_log = S.dcache.huf_log; c.huf_valid = 1; c.fse_valid = 1;
                        c.rep[0] = S.dcache.rep[0]; c.rep[1] = S.dcache.rep[1]; c.rep[2] = S.dcache.rep[2];
                    }
                }
                if (tid == 0) { c.dict_content = S.dcache.content; c.dict_content_len = S.dcache.content_len; }
            }
            // ---------------- blocks
            uint32_t last = 0;
            while (true) {
                const bool block_pre = pre;
                pre = false;
                if (tid == 0 && !block_pre) parse_block_header(c, src, n);
                uint32_t btype = 0, bsize = 0;
                uint64_t out0 = 0, pos0 = 0;
                WG_SNAPSHOT(err = c.err; btype = c.btype; bsize = c.bsize; last = c.last; out0 = c.out; pos0 = c.pos);
                if (err) break;
                if (btype == 0) { // K6 raw
                    if (bsize > cap - out0) { if (tid == 0) c.err = MZD_E_DSTSIZE; }
                    else {
                        wg_copy(dst + out0, src + pos0, bsize, tid);
                        if (tid == 0) { c.out = out0 + bsize; c.pos = pos0 + bsize; }
                    }
                } else if (btype == 1) { // K6 RLE
                    if (bsize > cap - out0) { if (tid == 0) c.err = MZD_E_DSTSIZE; }
                    else {
                        wg_fill(dst + out0, src[pos0], bsize, tid);
                        if (tid == 0) { c.out = out0 + bsize; c.pos = pos0 + 1; }
                    }
                } else {
                    const uint8_t* const blk = src + pos0;
                    STAMP(0);
                    // K0/K1/K3 headers: where everything is; nothing is decoded yet.  The two header regions
                    // (<= 256 bytes each: literals header + tree extent + jump table; sequence count, modes and
                    // the three normalized-count headers) are staged in LDS first, so that lane 0's byte-wise
                    // parsing does not pay an HBM round trip per byte.
                    TSTART();
                    if (!block_pre) {
                        for (uint32_t k = tid; k < bsize && k < 256; k += kWG) S.stage[k] = blk[k];
                        __syncthreads();
                    }
                    if (tid == 0) {
                        c.huf_ready = 0; c.huf_fill = 0; c.lit_done = 0; c.walk_prog = 0; c.exec_done = 0; c.exec_pos = out0;
                        c.next_stream = 0; c.streams_done = 0; c.streams_mask = 0;
                        c.tables_ready = 0; c.plan_prog = 0; c.copy_prog = 0; c.plan_lit_used = 0; c.plan_too_long = 0; c.seq_parsed = 0;
                        c.rep_op[0] = 0 | (1 << 2) | (2 << 4); c.rep_op[1] = 0; c.rep_op[2] = 0; c.rep_op[3] = 0;
                        if (!block_pre) parse_literals(c, S.stage, bsize);
                    }
                    uint32_t lit_type = 0, nlit = 0, streams = 0, nseq = 0, seq_len = 0;
                    uint64_t lit_off = 0, seq_off = 0;
                    WG_SNAPSHOT(err = c.err; lit_type = c.lit_type; nlit = c.nlit; streams = c.streams; lit_off = c.lit_off;
                                seq_off = c.seq_off; seq_len = c.seq_len);
                    if (err) break;
                    if (!block_pre) {
                        for (uint32_t k = tid; k < seq_len && k < 256; k += kWG) S.stage[256 + k] = src[seq_off + k];
                        __syncthreads();
                    }
                    STAMP(1);
                    // The sequence header (three normalized-count descriptions: a serial bit parse) is read by lane 0 of
                    // the walking wavefront INSIDE the pipeline, so the literal side (Huffman tree, streams) starts at once.
                    auto get_seq = [&]() -> bool { // nseq / seq_off / seq_len once the header is parsed; false: the block failed
                        if (!spin_ge(&c.seq_parsed, 1, &c.err) || __atomic_load_n(&c.err, __ATOMIC_RELAXED)) return false;
                        nseq = c.nseq; seq_off = c.seq_off; seq_len = c.seq_len;
                        return true;
                    };
                    const uint8_t* const lit = lit_type == 0 ? src + lit_off : lit_buf;
                    // K2 worker: take Huffman streams from the block's queue until none is left
                    // 2 KiB of LDS per decoding wavefront, borrowed from buffers that are idle while literals decode: the
                    // copier's staging buffers (waves 1, 2); the walker's ring (waves 0, 3: they decode after the walk)
                    uint8_t* const hseg = wave == 1 ? S.stage + 2064 : (wave == 2 ? S.hseg2 : (wave == 0 ? S.ring : S.ring + 4096));
                    // A block without sequences IS its literals: the Huffman streams are then decoded straight into the
                    // output (no literal buffer, no copy), provided they fit -- decided once the sequence header is parsed.
                    auto lit_in_place = [&]() -> bool { return lit_type >= 2 && nseq == 0 && nlit <= cap - out0; };
                    auto huf_streams = [&](uint32_t max_take) {
                        const uint32_t hl = c.huf_log;
                        uint8_t* const lbase = lit_in_place() ? dst + out0 : lit_buf;
                        for (uint32_t took = 0; took < max_take; took++) {
                            // every lane takes part (lanes != 0 add 0): no divergent region around the returning atomic
                            uint32_t st = __atomic_fetch_add(&c.next_stream, lane == 0 ? 1u : 0u, __ATOMIC_RELAXED);
                            st = (uint32_t)__builtin_amdgcn_readfirstlane(st);
                            if (st >= streams || st >= 4) break;
                            int r = 0;
                            if (!__atomic_load_n(&c.err, __ATOMIC_RELAXED))
                                r = huf_stream_wave(blk + c.s_off[st], c.s_len[st], lbase + c.s_out[st], c.s_n[st], hl, hseg, lane);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) { post_err(&c.err, r); __atomic_fetch_or(&c.streams_mask, 1u << st, __ATOMIC_RELAXED); __atomic_fetch_add(&c.streams_done, 1u, __ATOMIC_RELAXED); }
                        }
                    };
                    // wavefronts 0 and 3 decode literals too once their own role is over (at once in a block without sequences)
                    auto huf_helper = [&]() {
                        if (lit_type < 2) return;
                        if (lit_type == 2 && !spin_ge(&c.huf_fill, 2, &c.err)) return;
                        if (!get_seq()) return;
                        huf_streams(4);
                    };
                    // ---- the block pipeline, one role per wavefront:
                    //   wave 0  K3 tables, K4a serial state walk
                    //   wave 1  K1/K2 literals (streams 0,1), then the copying half of K5
                    //   wave 2  K2 literals (streams 2,3), then K7 hashing behind the copier
                    //   wave 3  K4b field conversion + repeat offsets + positions (the plan), behind the walker
                    if (wave == 0) {
                        __builtin_amdgcn_s_setprio(MZD_PRIO_WALK); // header parse, tables and walk are one serial chain: the block's critical path
                        if (lane == 0 && !block_pre) parse_seq_header(c, S.stage + 256, seq_len, 256);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) flag_store(&c.seq_parsed, 1);
                        TFIN(6);
                        if (get_seq() && nseq) {
                            if (lane == 0 && (c.mode[0] != 3 || c.mode[1] != 3 || c.mode[2] != 3)) c.lds_dict_fse = 0; // no longer the dictionary's
                            build_tables_wave(lane);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) flag_store(&c.tables_ready, 1);
                            STAMP(4);
                            TFIN(5);
                            __builtin_amdgcn_s_setprio(MZD_PRIO_WALK); // the chain is the critical path: win issue arbitration on this SIMD
                            int rc = walk_sequences_wave(src + seq_off, seq_len, nseq, walk, &c.walk_prog, lane);
                            __builtin_amdgcn_s_setprio(0);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) {
                                post_err(&c.err, rc);
                                c.fse_valid = 1;
                                flag_store(&c.walk_prog, rc ? kWalkFin : (nseq | kWalkFin)); // a failed walk publishes nothing
                            }
                            STAMP(5);
                            TFIN(0);
                        }
                        __builtin_amdgcn_s_setprio(0);
                        huf_helper();
                        if (last && pos0 + bsize + (hashing ? 4u : 0u) == n) { __builtin_amdgcn_s_setprio(MZD_PRE_PRIO); pre_parse_next(a, lane); __builtin_amdgcn_s_setprio(0); } // this block closes the file: the next file's headers, meanwhile
                    } else if (wave == 3) {
                        if (get_seq() && nseq) {
                            int rc = MZD_E_CORRUPT;
                            if (spin_ge(&c.tables_ready, 1, &c.err)) {
                                PlanCtx px{walk, src + seq_off, &c.walk_prog, nlit, 1u, {c.rep[0], c.rep[1], c.rep[2]}};
                                __builtin_amdgcn_s_setprio(MZD_PRIO_PLAN);
                                rc = plan_wave(seqs, nseq, px, lane);
                                __builtin_amdgcn_s_setprio(0);
                                if (rc == kPlanBlockTooLong) rc = 0; // (not an error yet: Ctl::plan_too_long, copy_wave)
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) {
                                post_err(&c.err, rc);
                                flag_store(&c.plan_prog, flag_load(&c.plan_prog) | kPlanFin);
                            }
                            TFIN(3);
                        }
                        huf_helper(); // the ring (its staging area) is free: the walker has finished before the planner does
                    } else {
                        int rc = 0;
                        // the literals gate the copier (the tail of the block): the copying wavefront's tree + first stream run
                        // at the copier's priority, the remaining streams just below
                        if (wave == 1) __builtin_amdgcn_s_setprio(MZD_PRIO_COPY); else __builtin_amdgcn_s_setprio(MZD_PRIO_PLAN);
                        if (lit_type == 2) { // K1: the Huffman tree (from an LDS copy of its description), by wavefront 1
                            if (wave == 1) { // weights: serial (lane 0); table: the whole wavefront
                                const uint32_t tl = c.huf_tree_len; // <= 129 bytes
                                for (uint32_t k = (uint32_t)lane; k < tl + 8; k += 64) S.stage[1024 + k] = k < tl ? blk[c.huf_tree_off + k] : 0;
                                int used = 1;
                                if (lane == 0) { c.lds_dict_huf = 0; used = read_huf_weights_staged(1024, c.huf_tree_len); } // (the table is no longer a dictionary's)
                                used = __builtin_amdgcn_readfirstlane(used);
                                TFIN(7);
                                int hr = used <= 0 ? MZD_E_CORRUPT : finish_huf_table_wave(lane);
                                if (lane == 0) { if (hr) post_err(&c.err, hr); else c.huf_valid = 1; }
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                if (lane == 0) flag_store(&c.huf_fill, 2);
                                TFIN(8);
                            }
                            spin_ge(&c.huf_fill, 2, &c.err);
                        }
                        const bool failed = __atomic_load_n(&c.err, __ATOMIC_RELAXED) != 0;
                        if (lit_type == 1) { // RLE literals
                            uint32_t w = (uint32_t)src[lit_off] * 0x01010101u;
                            for (uint32_t k = (uint32_t)(tid - 64) * 16; k < nlit; k += 128 * 16)
                                *reinterpret_cast<uint4*>(lit_buf + k) = make_uint4(w, w, w, w); // lit_buf has slack past nlit
                        } else if (lit_type >= 2 && !failed && get_seq()) { // K2: the copying wavefront decodes one stream and then
                            huf_streams(wave == 1 && !lit_in_place() ? 1u : 4u); // copies behind the literals; wavefront 2 (and idle ones) drain the queue
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            post_err(&c.err, rc);
                            __atomic_fetch_add(&c.lit_done, 1u, __ATOMIC_RELAXED);
                        }
                        __builtin_amdgcn_s_setprio(0);
                        STAMP(3);
                        if (wave == 1) TFIN(4);
                        if (wave == 1) { // the copying half of K5
                            uint64_t opos = out0;
                            rc = MZD_E_CORRUPT;
                            if (get_seq() && (lit_type != 1 || spin_ge(&c.lit_done, 2, &c.err))) { // RLE literals: both halves filled
                                CopyCtx cx{seqs, dst, c.frame_out0, c.dict_content, c.dict_content_len, lit_in_place() ? dst + out0 : lit, nlit, cap, lit_type >= 2 ? streams : 0u, {c.rep[0], c.rep[1], c.rep[2]}, nullptr};
                                TFIN(9);
                                __builtin_amdgcn_s_setprio(MZD_PRIO_COPY); // second on the critical path, behind the walker
                                rc = copy_wave(nseq, cx, &opos, lane);
                                __builtin_amdgcn_s_setprio(0);
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) {
                                post_err(&c.err, rc);
                                flag_store(&c.exec_done, 1);
                                c.out = opos; c.pos = pos0 + bsize;
                                if (a.debug) {
                                    DebugSlot& ds = a.debug[a.wg0 + blockIdx.x];
                                    ds.n_lit = nlit; ds.n_seq = nseq; ds.lit_is_raw = lit_type == 0 || lit_in_place(); ds.lit_raw_ptr = (uint64_t)(uintptr_t)(lit_in_place() ? dst + out0 : lit);
                                }
                            }
                            STAMP(6);
                            TFIN(1);
                        } else if (hashing) { // wave 2, K7: hash behind the copier while it works
                            const uint8_t* fp = dst + c.frame_out0;
                            for (uint32_t it = 0; it < (1u << 24); it++) {
                                const uint32_t fin = flag_load(&c.exec_done);
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                const uint64_t pos = __atomic_load_n(&c.exec_pos, __ATOMIC_RELAXED);
                                uint64_t upto = (pos - c.frame_out0) / 32;
                                if (!fin) upto = upto >= xstripes + 64 ? xstripes + ((upto - xstripes) & ~7ull) : xstripes; // >= 2 KiB at a time, whole groups of 8 stripes
                                if (upto > xstripes) xxh_advance(xv, xstripes, upto, fp, lane);
                                else if (fin || __atomic_load_n(&c.err, __ATOMIC_RELAXED)) break;
                                else __builtin_amdgcn_s_sleep(8);
                                if (fin) break;
                            }
                            TFIN(2);
                        }
                    }
                }
                if (btype == 2) { // the block's repeat-offset transform (the planner leaves it symbolic) -> the offsets after it
                    __syncthreads();
                    if (tid == 0) {
                        RepOp Rf; Rf.s = c.rep_op[0]; Rf.v0 = (int32_t)c.rep_op[1]; Rf.v1 = (int32_t)c.rep_op[2]; Rf.v2 = (int32_t)c.rep_op[3];
                        const uint32_t a0 = c.rep[0], a1 = c.rep[1], a2 = c.rep[2];
                        c.rep[0] = rep_eval(Rf, 0, a0, a1, a2); c.rep[1] = rep_eval(Rf, 1, a0, a1, a2); c.rep[2] = rep_eval(Rf, 2, a0, a1, a2);
                    }
                }
                WG_SNAPSHOT(err = c.err);
                STAMP(6);
                if (err || last) break;
            }
            if (err) break;
            // ---------------- frame trailer: content size and checksum (K7)
            if (tid == 0) {
                uint64_t made = c.out - c.frame_out0;
                if (c.has_fcs && made != c.fcs) c.err = MZD_E_CORRUPT;
                else if (c.has_cksum && n - c.pos < 4) c.err = MZD_E_TRUNCATED;
            }
            uint32_t has_ck = 0;
            uint64_t fout0 = 0, out_now = 0, pos_now = 0;
            WG_SNAPSHOT(err = c.err; has_ck = c.has_cksum; fout0 = c.frame_out0; out_now = c.out; pos_now = c.pos);
            if (err) break;
            if (has_ck) {
                if (wave == 2) {
                    const uint32_t stored = ld32(src + pos_now); // (issued before the digest is closed: its round trip overlaps)
                    xxh_advance(xv, xstripes, (out_now - fout0) / 32, dst + fout0, lane);
                    uint64_t h = xxh_finish(xv, dst + fout0, out_now - fout0, lane);
                    if (lane == 0) {
#ifndef MZD_EXP_NOHASH
                        if ((uint32_t)h != stored) c.err = MZD_E_CHECKSUM;
#endif
                        c.pos = pos_now + 4;
                    }
                    STAMP(7);
                }
                WG_SNAPSHOT(err = c.err);
                if (err) break;
            }
        }
        if (tid == 0) { a.jobs[j].out_len = c.out; a.jobs[j].status = c.err; }
        STAMP_FLUSH();
        TTASK_END();
        TFIN_FLUSH();
        __syncthreads();
    }
}


// ---- driver 2: block tasks.  inter-workgroup hand-over (agent scope): a task publishes, its successor on another CU acquires
__device__ __forceinline__ uint32_t g_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void g_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void g_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Agent-scope fences write back / invalidate the XCD's whole L2 (buffer_wbl2 / buffer_inv): they are kept for the one thing
// that needs them -- the output bytes a successor on another XCD reads -- and everything small (task records, per-file
// state, table areas) travels through agent-scope atomic loads and stores, which are coherent by themselves.
// `g_settle` orders such stores before the flag that publishes them.
template <class T> __device__ __forceinline__ T g_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void g_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_settle() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
// wait until *p >= want (bounded: a launch that lost a task must end, not hang); one lane calls this
__device__ __noinline__ bool g_wait_ge(const uint32_t* p, uint32_t want) {
    for (uint32_t it = 0; it < (1u << 23); it++) {
        if (g_load(p) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}
// lane 0: what the predecessor of task t left behind -> S.c.pred_*.  false: the launch is broken (timeout).
__device__ __noinline__ bool load_pred(const FileState* fs, uint32_t t) {
    Ctl& c = S.c;
    if (t == 0) {
        c.pred_err = 0; c.pred_out = 0; c.pred_frame_out0 = 0; c.pred_xstripes = 0;
        c.pred_rep[0] = 1; c.pred_rep[1] = 4; c.pred_rep[2] = 8;
        for (int k = 0; k < 4; k++) c.pred_xxh[k] = 0;
    } else {
        if (!g_wait_ge(&fs->copied, t)) { c.pred_err = MZD_E_DEVICE; c.pred_out = 0; c.pred_frame_out0 = 0; c.pred_xstripes = 0; return false; }
        c.pred_err = g_ld(&fs->err); c.pred_out = g_ld(&fs->out); c.pred_frame_out0 = g_ld(&fs->frame_out0); c.pred_xstripes = g_ld(&fs->xstripes);
        c.pred_rep[0] = g_ld(&fs->rep[0]); c.pred_rep[1] = g_ld(&fs->rep[1]); c.pred_rep[2] = g_ld(&fs->rep[2]);
        for (int k = 0; k < 4; k++) c.pred_xxh[k] = g_ld(&fs->xxh[k]);
    }
    return true;
}

__global__ __launch_bounds__(kWG, 4) void mzd_decode_kernel_tasks(KernelArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t slot = a.wg0 + blockIdx.x; // this workgroup's place in the scratch arrays
    uint8_t* const lit_buf = a.lit_scratch + (size_t)slot * kLitStride;
    uint4* const seqs = a.seq_scratch + (size_t)slot * kSeqStride;
    uint4* const walk = a.walk_scratch + (size_t)slot * kSeqStride;
    Ctl& c = S.c;
    if (tid < 36) S.ll_base[tid] = LL_BASE[tid];
    if (tid < 53) S.ml_base[tid] = ML_BASE[tid];

    for (;;) {
        // ---------------- take a task: tickets below njobs are the first blocks of the files, the others the pushed
        // continuations in push order (a ticket may have to wait for its record; it gives up once every file is finished)
        if (tid == 0) {
            c.t_valid = 0;
            const uint32_t ticket = atomicAdd(&a.counter[0], 1u);
            const uint32_t nq = queue_len(a);
            if (ticket < nq) { c.job = queue_job(a, ticket); c.task = 0; c.pos = 0; c.in_frame = 0; c.with_dict = 0; c.t_valid = 1; }
            else {
                const uint32_t m = ticket - nq;
                const ContRecord* r = &a.ring[m % a.ring_cap];
                const uint64_t want = ((uint64_t)a.epoch << 32) | (uint64_t)(m + 1);
                for (uint32_t it = 0; it < (1u << 23); it++) {
                    bool got = __hip_atomic_load(&r->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
                    if (!got && g_load(&a.counter[3]) >= nq) { // every file is finished: nothing is pushed any more
                        got = __hip_atomic_load(&r->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
                        if (!got) break;
                    }
                    if (got) {
                        c.job = g_ld(&r->job); c.task = g_ld(&r->task); c.pos = g_ld(&r->pos); c.in_frame = g_ld(&r->in_frame); c.with_dict = g_ld(&r->with_dict);
                        c.has_fcs = g_ld(&r->has_fcs); c.has_cksum = g_ld(&r->has_cksum); c.block_max = g_ld(&r->block_max); c.fcs = g_ld(&r->fcs);
                        c.t_valid = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
        }
        uint32_t t_valid = 0, j = 0, t = 0, in_frame = 0;
        WG_SNAPSHOT(t_valid = c.t_valid; j = c.job; t = c.task; in_frame = c.in_frame);
        if (!t_valid) break;
        TTASK();
        const uint8_t* const src = a.jobs[j].src;
        const uint64_t n = a.jobs[j].src_len;
        uint8_t* const dst = a.jobs[j].dst;
        const uint64_t cap = a.jobs[j].dst_cap;
        const uint32_t job_dict = a.jobs[j].dict;
        FileState* const fs = &a.fstate[j];
        TableArea* const ta = &a.tables[j];
        if (tid == 0) {
            c.out = 0; c.err = 0; c.action = 0; c.diag_slow = 0; c.pred_ready = 0; c.tables_published = 0; c.last = 0;
#ifdef MZD_STAMPS
            for (int k_ = 0; k_ < 8; k_++) S.cdiag[k_] = 0;
#endif
            if (job_dict > a.ndicts) c.err = MZD_E_DICT;
            if (in_frame) { // the frame's context travels with the task; its dictionary content is looked up again
                c.dict_content = nullptr; c.dict_content_len = 0;
                if (c.with_dict && job_dict >= 1 && job_dict <= a.ndicts) { c.dict_content = a.dicts[job_dict - 1].content; c.dict_content_len = a.dicts[job_dict - 1].content_len; }
                c.huf_valid = 1; c.fse_valid = 1; // provisional: what is inherited is checked when it is fetched
            }
        }
        int err = 0;
        uint32_t action = 0;
        uint64_t xv = 0, xstripes = 0; // K7 state of the hashing wavefront (wave 2)
        STAMP_DECL;

        // ---------------- frame header (K0): a task that does not continue a frame starts at one (or at the file's end)
        bool frame_first = false;
        if (!in_frame) {
            for (;;) {
                if (tid == 0 && !c.err) parse_frame_or_skip(c, src, n, a.dicts, a.ndicts, job_dict);
                WG_SNAPSHOT(err = c.err; action = c.action);
                if (err || action != 1) break; // 1: a skippable frame was skipped, look again
            }
            frame_first = !err && action != 2;
            if (frame_first && action == 3) { // dictionary: entropy tables, repeat offsets and content
                const DevDict* dd = &a.dicts[job_dict - 1];
                if (dd->formatted) {
                    for (int i = tid; i < 512; i += kWG) { S.ll[i] = dd->ll[i]; S.ml[i] = dd->ml[i]; }
                    for (int i = tid; i < 256; i += kWG) S.of[i] = dd->of[i];
                    for (int i = tid; i < 2048; i += kWG) S.huf[i] = dd->huf[i];
                    if (tid == 0) {
                        c.al[0] = dd->al[0]; c.al[1] = dd->al[1]; c.al[2] = dd->al[2];
                        c.huf_log = dd->huf_log; c.huf_valid = 1; c.fse_valid = 1;
                        c.rep[0] = dd->rep[0]; c.rep[1] = dd->rep[1]; c.rep[2] = dd->rep[2];
                    }
                }
                if (tid == 0) { c.dict_content = dd->content; c.dict_content_len = dd->content_len; c.with_dict = 1; }
            }
        }
        const bool hashing = c.has_cksum != 0; // (garbage without a frame; unused then)

        // ---------------- block header, and the successor is pushed before anything is decoded
        bool have_block = !err && (in_frame || frame_first);
        uint32_t btype = 0, bsize = 0, last = 0;
        uint64_t pos0 = 0;
        if (have_block) {
            if (tid == 0) parse_block_header(c, src, n);
            WG_SNAPSHOT(err = c.err; btype = c.btype; bsize = c.bsize; last = c.last; pos0 = c.pos);
            if (err) have_block = false;
        }
        bool is_final = true; // no successor: this task closes the file
        if (have_block) {
            const uint64_t body_end = pos0 + (btype == 1 ? 1u : bsize);
            uint64_t next_pos = body_end;
            bool push = !last;
            if (last) { // the next frame, if any, starts behind the optional checksum
                const bool ck_ok = !hashing || n - body_end >= 4; // a truncated checksum is reported by this task
                next_pos = body_end + (hashing ? 4 : 0);
                push = ck_ok && next_pos < n;
            }
            if (push) {
                is_final = false;
                if (tid == 0) {
                    const uint32_t m = atomicAdd(&a.counter[2], 1u);
                    ContRecord* r = &a.ring[m % a.ring_cap];
                    g_st(&r->job, j); g_st(&r->task, t + 1); g_st(&r->pos, next_pos); g_st(&r->in_frame, last ? 0u : 1u); g_st(&r->with_dict, c.with_dict);
                    g_st(&r->has_fcs, c.has_fcs); g_st(&r->has_cksum, c.has_cksum); g_st(&r->block_max, c.block_max); g_st(&r->fcs, c.fcs);
                    g_settle();
                    __hip_atomic_store(&r->seq, ((uint64_t)a.epoch << 32) | (uint64_t)(m + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }

        // ---------------- the block
        uint64_t out0 = 0, frame_start = 0, out_end = 0;
        bool pred_loaded = false; // (workgroup-uniform) c.pred_* is filled in
        if (have_block && btype < 2) { // K6 raw / RLE: nothing to decode ahead; wait for the predecessor, then copy / fill
            if (tid == 0) { load_pred(fs, t); }
            int perr = 0;
            WG_SNAPSHOT(perr = c.pred_err; out0 = c.pred_out; frame_start = frame_first ? c.pred_out : c.pred_frame_out0);
            if (t && wave == 2) g_acquire(); // the hash reads what the predecessor wrote
            pred_loaded = true;
            out_end = out0;
            if (!perr) {
                if (bsize > cap - out0) { if (tid == 0) c.err = MZD_E_DSTSIZE; }
                else {
                    if (btype == 0) wg_copy(dst + out0, src + pos0, bsize, tid);
                    else wg_fill(dst + out0, src[pos0], bsize, tid);
                    out_end = out0 + bsize;
                }
            }
            if (tid == 0) { c.out = out_end; c.pos = pos0 + (btype == 1 ? 1u : bsize); }
            __syncthreads();
            if (wave == 2 && hashing && !perr) { // this block's stripes (K7 state travels from task to task)
                xv = frame_first ? xxh_init(lane) : c.pred_xxh[lane & 3];
                xstripes = frame_first ? 0 : c.pred_xstripes;
                xxh_advance(xv, xstripes, (out_end - frame_start) / 32, dst + frame_start, lane);
            }
        } else if (have_block) {
            const uint8_t* const blk = src + pos0;
            STAMP(0);
            // K0/K1/K3 headers: where everything is; nothing is decoded yet.  The two header regions
            // (<= 256 bytes each: literals header + tree extent + jump table; sequence count, modes and
            // the three normalized-count headers) are staged in LDS first, so that lane 0's byte-wise
            // parsing does not pay an HBM round trip per byte.
            TSTART();
            for (uint32_t k = tid; k < bsize && k < 256; k += kWG) S.stage[k] = blk[k];
            __syncthreads();
            if (tid == 0) {
                c.huf_ready = 0; c.huf_fill = 0; c.lit_done = 0; c.walk_prog = 0; c.exec_done = 0; c.exec_pos = 0;
                c.next_stream = 0; c.streams_done = 0; c.streams_mask = 0;
                c.tables_ready = 0; c.plan_prog = 0; c.copy_prog = 0; c.plan_lit_used = 0; c.plan_too_long = 0; c.seq_parsed = 0;
                c.rep_op[0] = 0 | (1 << 2) | (2 << 4); c.rep_op[1] = 0; c.rep_op[2] = 0; c.rep_op[3] = 0; // identity: a block without sequences
                parse_literals(c, S.stage, bsize);
            }
            uint32_t lit_type = 0, nlit = 0, streams = 0, nseq = 0, seq_len = 0;
            uint64_t lit_off = 0, seq_off = 0;
            WG_SNAPSHOT(err = c.err; lit_type = c.lit_type; nlit = c.nlit; streams = c.streams; lit_off = c.lit_off;
                        seq_off = c.seq_off; seq_len = c.seq_len);
            if (!err) {
                for (uint32_t k = tid; k < seq_len && k < 256; k += kWG) S.stage[256 + k] = src[seq_off + k];
                __syncthreads();
                STAMP(1);
                // The sequence header (three normalized-count descriptions: a serial bit parse) is read by lane 0 of
                // the walking wavefront INSIDE the pipeline, so the literal side (Huffman tree, streams) starts at once.
                auto get_seq = [&]() -> bool { // nseq / seq_off / seq_len once the header is parsed; false: the block failed
                    if (!spin_ge(&c.seq_parsed, 1, &c.err) || __atomic_load_n(&c.err, __ATOMIC_RELAXED)) return false;
                    nseq = c.nseq; seq_off = c.seq_off; seq_len = c.seq_len;
                    return true;
                };
                const uint8_t* const lit = lit_type == 0 ? src + lit_off : lit_buf;
                // K2 worker: take Huffman streams from the block's queue until none is left
                // 2 KiB of LDS per decoding wavefront, borrowed from buffers that are idle while literals decode: the
                // copier's staging buffers (wave 1); the walker's ring (waves 0, 3: they decode after the walk)
                uint8_t* const hseg = wave == 1 ? S.stage + 2064 : (wave == 2 ? S.hseg2 : (wave == 0 ? S.ring : S.ring + 4096));
                // A block without sequences IS its literals: the Huffman streams are then decoded straight into the
                // output (no literal buffer, no copy), provided they fit and the output position is already known
                // (first task of a file) -- decided once the sequence header is parsed.
                auto lit_in_place = [&]() -> bool { return lit_type >= 2 && nseq == 0 && t == 0 && nlit <= cap; };
                auto huf_streams = [&](uint32_t max_take) {
                    const uint32_t hl = c.huf_log;
                    uint8_t* const lbase = lit_in_place() ? dst : lit_buf;
                    for (uint32_t took = 0; took < max_take; took++) {
                        // every lane takes part (lanes != 0 add 0): no divergent region around the returning atomic
                        uint32_t st = __atomic_fetch_add(&c.next_stream, lane == 0 ? 1u : 0u, __ATOMIC_RELAXED);
                        st = (uint32_t)__builtin_amdgcn_readfirstlane(st);
                        if (st >= streams || st >= 4) break;
                        int r = 0;
                        if (!__atomic_load_n(&c.err, __ATOMIC_RELAXED))
                            r = huf_stream_wave(blk + c.s_off[st], c.s_len[st], lbase + c.s_out[st], c.s_n[st], hl, hseg, lane);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) { post_err(&c.err, r); __atomic_fetch_or(&c.streams_mask, 1u << st, __ATOMIC_RELAXED); __atomic_fetch_add(&c.streams_done, 1u, __ATOMIC_RELAXED); }
                    }
                };
                // wavefronts 0 and 3 decode literals too once their own role is over (at once in a block without sequences)
                auto huf_helper = [&]() {
                    if (lit_type < 2) return;
                    if (!spin_ge(&c.huf_fill, 2, &c.err)) return;
                    if (!get_seq()) return;
                    huf_streams(4);
                };
                // the file's tables at version t (what the predecessor left): one lane waits, the wavefront copies
                auto wait_tables = [&]() -> bool {
                    int ok = 1;
                    if (lane == 0) ok = g_wait_ge(&fs->tables_ver, t) ? 1 : 0;
                    ok = __builtin_amdgcn_readfirstlane(ok);
                    return ok != 0;
                };
                // ---- the block pipeline, one role per wavefront:
                //   wave 0  K3 tables, K4a serial state walk
                //   wave 1  K1/K2 literals (first stream), then the copying half of K5 (after the predecessor)
                //   wave 2  K2 literals (other streams), then K7 hashing behind the copier
                //   wave 3  publishes the tables for the successor, K4b field conversion + repeat offsets + positions (the plan)
                if (wave == 0) {
                    __builtin_amdgcn_s_setprio(MZD_PRIO_WALK); // header parse, tables and walk are one serial chain: the block's critical path
                    if (lane == 0) parse_seq_header(c, S.stage + 256, seq_len, 256);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) flag_store(&c.seq_parsed, 1);
                    TFIN(6);
                    if (get_seq() && nseq) {
                        int rc = 0;
                        const bool inherit = !frame_first && (c.mode[0] == 3 || c.mode[1] == 3 || c.mode[2] == 3);
                        if (inherit) { // repeat mode: the table the previous block used (another workgroup built it)
                            if (!wait_tables()) rc = MZD_E_DEVICE;
                            else if (!g_ld(&fs->fse_valid)) rc = MZD_E_CORRUPT;
                            else {
                                if (c.mode[0] == 3) { for (int i = lane; i < 512; i += 64) S.ll[i] = g_ld(&ta->ll[i]); if (lane == 0) c.al[0] = g_ld(&fs->al[0]); }
                                if (c.mode[1] == 3) { for (int i = lane; i < 256; i += 64) S.of[i] = g_ld(&ta->of[i]); if (lane == 0) c.al[1] = g_ld(&fs->al[1]); }
                                if (c.mode[2] == 3) { for (int i = lane; i < 512; i += 64) S.ml[i] = g_ld(&ta->ml[i]); if (lane == 0) c.al[2] = g_ld(&fs->al[2]); }
                            }
                        }
                        if (!rc) build_tables_wave(lane);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) { post_err(&c.err, rc); flag_store(&c.tables_ready, 1); }
                        STAMP(4);
                        TFIN(5);
                        if (!rc) {
                            __builtin_amdgcn_s_setprio(MZD_PRIO_WALK); // the chain is the critical path: win issue arbitration on this SIMD
                            rc = walk_sequences_wave(src + seq_off, seq_len, nseq, walk, &c.walk_prog, lane);
                            __builtin_amdgcn_s_setprio(0);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            post_err(&c.err, rc);
                            flag_store(&c.walk_prog, rc ? kWalkFin : (nseq | kWalkFin)); // a failed walk publishes nothing
                        }
                        STAMP(5);
                        TFIN(0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    huf_helper();
                } else if (wave == 3) {
                    const bool seq_ok = get_seq();
                    // ---- the successor's inheritance: once this block's tables are final (and whatever it inherits itself has
                    // been read), the kinds it rebuilt go to the file's table area and the version moves on
                    if (seq_ok && !is_final) { // (a file's last task has nobody to publish for)
                        bool ok = true;
                        if (nseq) ok = spin_ge(&c.tables_ready, 1, &c.err);
                        if (ok && lit_type >= 2) ok = spin_ge(&c.huf_fill, 2, &c.err);
                        if (ok && !__atomic_load_n(&c.err, __ATOMIC_RELAXED) && wait_tables()) {
                            if (!last) {
                                const bool fse_new = nseq != 0 || (frame_first && c.fse_valid), huf_new = lit_type == 2 || (frame_first && c.huf_valid);
                                if (fse_new) {
                                    for (int i = lane; i < 512; i += 64) { g_st(&ta->ll[i], S.ll[i]); g_st(&ta->ml[i], S.ml[i]); }
                                    for (int i = lane; i < 256; i += 64) g_st(&ta->of[i], S.of[i]);
                                }
                                if (huf_new) for (int i = lane; i < 1024; i += 64) g_st(&reinterpret_cast<uint32_t*>(ta->huf)[i], reinterpret_cast<const uint32_t*>(S.huf)[i]);
                                if (lane == 0) {
                                    if (fse_new) { g_st(&fs->al[0], c.al[0]); g_st(&fs->al[1], c.al[1]); g_st(&fs->al[2], c.al[2]); g_st(&fs->fse_valid, 1u); }
                                    else if (frame_first) g_st(&fs->fse_valid, 0u);
                                    if (huf_new) { g_st(&fs->huf_log, c.huf_log); g_st(&fs->huf_valid, 1u); }
                                    else if (frame_first) g_st(&fs->huf_valid, 0u);
                                }
                            }
                            g_settle();
                            if (lane == 0) { g_store(&fs->tables_ver, t + 1); c.tables_published = 1; }
                        }
                    }
                    if (seq_ok && nseq) {
                        int rc = MZD_E_CORRUPT;
                        if (spin_ge(&c.tables_ready, 1, &c.err)) {
                            PlanCtx px{walk, src + seq_off, &c.walk_prog, nlit, frame_first ? 1u : 0u, {c.rep[0], c.rep[1], c.rep[2]}};
                            __builtin_amdgcn_s_setprio(MZD_PRIO_PLAN);
                            rc = plan_wave(seqs, nseq, px, lane);
                            __builtin_amdgcn_s_setprio(0);
                            if (rc == kPlanBlockTooLong) rc = 0; // (not an error yet: Ctl::plan_too_long, copy_wave)
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            post_err(&c.err, rc);
                            flag_store(&c.plan_prog, flag_load(&c.plan_prog) | kPlanFin);
                        }
                        TFIN(3);
                    }
                    huf_helper(); // the ring (its staging area) is free: the walker has finished before the planner does
                } else {
                    int rc = 0;
                    // the literals gate the copier (the tail of the block): the copying wavefront's tree + first stream run
                    // at the copier's priority, the remaining streams just below
                    if (wave == 1) __builtin_amdgcn_s_setprio(MZD_PRIO_COPY); else __builtin_amdgcn_s_setprio(MZD_PRIO_PLAN);
                    if (lit_type >= 2) { // K1: the Huffman table, by wavefront 1: built from this block's tree or inherited
                        if (wave == 1) {
                            int hr = 0;
                            if (lit_type == 2) {
                                const uint32_t tl = c.huf_tree_len; // <= 129 bytes
                                for (uint32_t k = (uint32_t)lane; k < tl + 8; k += 64) S.stage[1024 + k] = k < tl ? blk[c.huf_tree_off + k] : 0;
                                int used = 1;
                                if (lane == 0) used = read_huf_weights_staged(1024, c.huf_tree_len);
                                used = __builtin_amdgcn_readfirstlane(used);
                                TFIN(7);
                                hr = used <= 0 ? MZD_E_CORRUPT : finish_huf_table_wave(lane);
                            } else if (!frame_first) { // treeless: the table of the previous compressed-literals block
                                if (!wait_tables()) hr = MZD_E_DEVICE;
                                else if (!g_ld(&fs->huf_valid)) hr = MZD_E_CORRUPT;
                                else {
                                    for (int i = lane; i < 1024; i += 64) reinterpret_cast<uint32_t*>(S.huf)[i] = g_ld(&reinterpret_cast<const uint32_t*>(ta->huf)[i]);
                                    if (lane == 0) c.huf_log = g_ld(&fs->huf_log);
                                }
                            }
                            if (lane == 0 && hr) post_err(&c.err, hr);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) flag_store(&c.huf_fill, 2);
                            TFIN(8);
                        }
                        spin_ge(&c.huf_fill, 2, &c.err);
                    }
                    const bool failed = __atomic_load_n(&c.err, __ATOMIC_RELAXED) != 0;
                    if (lit_type == 1) { // RLE literals
                        uint32_t w = (uint32_t)src[lit_off] * 0x01010101u;
                        for (uint32_t k = (uint32_t)(tid - 64) * 16; k < nlit; k += 128 * 16)
                            *reinterpret_cast<uint4*>(lit_buf + k) = make_uint4(w, w, w, w); // lit_buf has slack past nlit
                    } else if (lit_type >= 2 && !failed && get_seq()) { // K2: the copying wavefront decodes one stream and then
                        huf_streams(wave == 1 && !lit_in_place() ? 1u : 4u); // copies behind the literals; wavefront 2 (and idle ones) drain the queue
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) {
                        post_err(&c.err, rc);
                        __atomic_fetch_add(&c.lit_done, 1u, __ATOMIC_RELAXED);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    STAMP(3);
                    if (wave == 1) TFIN(4);
                    if (wave == 1) { // the copying half of K5: it needs the predecessor's output, position and repeat offsets
                        if (lane == 0) { load_pred(fs, t); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); flag_store(&c.pred_ready, 1); }
                        spin_ge(&c.pred_ready, 1, &c.err);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        if (t) g_acquire(); // the predecessor's output (another XCD's L2 may have held it)
                        const uint64_t o0 = c.pred_out;
                        uint64_t opos = o0;
                        rc = 0;
                        if (c.pred_err) rc = 0; // the file has already failed: nothing to copy (the error travels on)
                        else if (!get_seq() || (lit_type == 1 && !spin_ge(&c.lit_done, 2, &c.err))) rc = MZD_E_CORRUPT;
                        else {
                            const uint64_t fstart = frame_first ? o0 : c.pred_frame_out0;
                            CopyCtx cx{seqs, dst, fstart, c.dict_content, c.dict_content_len, lit_in_place() ? dst : lit, nlit, cap, lit_type >= 2 ? streams : 0u,
                                       {frame_first ? c.rep[0] : c.pred_rep[0], frame_first ? c.rep[1] : c.pred_rep[1], frame_first ? c.rep[2] : c.pred_rep[2]},
                                       a.debug ? seqs : nullptr};
                            TFIN(9);
                            __builtin_amdgcn_s_setprio(MZD_PRIO_COPY); // second on the critical path, behind the walker
                            rc = copy_wave(nseq, cx, &opos, lane);
                            __builtin_amdgcn_s_setprio(0);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            post_err(&c.err, rc);
                            c.out = opos; c.pos = pos0 + bsize;
                            flag_store(&c.exec_done, 1);
                            if (a.debug) {
                                DebugSlot& ds = a.debug[a.wg0 + blockIdx.x];
                                ds.n_lit = nlit; ds.n_seq = nseq; ds.lit_is_raw = lit_type == 0 || lit_in_place(); ds.lit_raw_ptr = (uint64_t)(uintptr_t)(lit_in_place() ? dst : lit);
                                if (j == 0) atomicMax(&a.counter[1], (t << 12) | (a.wg0 + blockIdx.x)); // the slot that ran the last compressed block of job 0
                            }
                        }
                        STAMP(6);
                        TFIN(1);
                    } else if (hashing) { // wave 2, K7: hash behind the copier while it works (state from the predecessor)
                        if (spin_ge(&c.pred_ready, 1, &c.err)) {
                            if (t) g_acquire();
                            if (!c.pred_err) {
                                const uint64_t fstart = frame_first ? c.pred_out : c.pred_frame_out0;
                                xv = frame_first ? xxh_init(lane) : c.pred_xxh[lane & 3];
                                xstripes = frame_first ? 0 : c.pred_xstripes;
                                const uint8_t* fp = dst + fstart;
                                for (uint32_t it = 0; it < (1u << 24); it++) {
                                    const uint32_t fin = flag_load(&c.exec_done);
                                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                    const uint64_t pos = fin ? c.out : __atomic_load_n(&c.exec_pos, __ATOMIC_RELAXED);
                                    uint64_t upto = pos > fstart ? (pos - fstart) / 32 : 0;
                                    if (!fin) upto = upto >= xstripes + 64 ? xstripes + ((upto - xstripes) & ~7ull) : xstripes; // >= 2 KiB at a time, whole groups of 8 stripes
                                    if (upto > xstripes) xxh_advance(xv, xstripes, upto, fp, lane);
                                    else if (fin || __atomic_load_n(&c.err, __ATOMIC_RELAXED)) break;
                                    else __builtin_amdgcn_s_sleep(8);
                                    if (fin) break;
                                }
                            }
                        }
                        TFIN(2);
                    }
                }
            }
            WG_SNAPSHOT(err = c.err);
            STAMP(6);
            pred_loaded = c.pred_ready != 0;
        }

        // ---------------- completion, in task order: frame trailer (K7), then the state for the successor
        __syncthreads();
        if (tid == 0 && !pred_loaded) load_pred(fs, t);
        int perr = 0;
        uint64_t pred_out = 0, fstart = 0, out_now = 0, pos_now = 0;
        uint32_t has_ck = 0;
        WG_SNAPSHOT(perr = c.pred_err; pred_out = c.pred_out; fstart = frame_first ? c.pred_out : c.pred_frame_out0; err = c.err; out_now = c.out; pos_now = c.pos; has_ck = c.has_cksum; last = c.last);
        if (!have_block) out_now = pred_out; // a task without a block (end of file, or a header error) produces nothing
        int final_err = perr ? perr : err;
        if (!final_err && have_block && last) { // frame trailer: content size and checksum
            if (tid == 0) {
                if (c.has_fcs && out_now - fstart != c.fcs) c.err = MZD_E_CORRUPT;
                else if (has_ck && n - pos_now < 4) c.err = MZD_E_TRUNCATED;
            }
            WG_SNAPSHOT(err = c.err);
            if (!err && has_ck) {
                if (wave == 2) {
                    xxh_advance(xv, xstripes, (out_now - fstart) / 32, dst + fstart, lane);
                    uint64_t h = xxh_finish(xv, dst + fstart, out_now - fstart, lane);
                    if (lane == 0) {
#ifndef MZD_EXP_NOHASH
                        if ((uint32_t)h != ld32(src + pos_now)) c.err = MZD_E_CHECKSUM;
#endif
                    }
                    STAMP(7);
                }
                WG_SNAPSHOT(err = c.err);
            }
            final_err = err;
        }
        // the state for the successor (or the file's result), published in task order
        if (!is_final) {
            if (wave == 2) { // K7 state lives in this wavefront's registers
                if (lane < 4) g_st(&fs->xxh[lane], xv);
                if (lane == 0) g_st(&fs->xstripes, xstripes);
            }
            if (!c.tables_published) { // raw/RLE block, early error: the tables are unchanged, the version still moves on
                if (tid == 0) c.t_valid = g_wait_ge(&fs->tables_ver, t) ? 1u : 0u;
                uint32_t ver_ok = 0, hv = 0, fv = 0;
                WG_SNAPSHOT(ver_ok = c.t_valid; hv = c.huf_valid; fv = c.fse_valid);
                if (ver_ok) {
                    if (frame_first && !final_err) { // the frame starts here: what its successors inherit is the dictionary's tables, or nothing
                        if (fv) {
                            for (int i = tid; i < 512; i += kWG) { g_st(&ta->ll[i], S.ll[i]); g_st(&ta->ml[i], S.ml[i]); }
                            for (int i = tid; i < 256; i += kWG) g_st(&ta->of[i], S.of[i]);
                        }
                        if (hv) for (int i = tid; i < 1024; i += kWG) g_st(&reinterpret_cast<uint32_t*>(ta->huf)[i], reinterpret_cast<const uint32_t*>(S.huf)[i]);
                        if (tid == 0) {
                            g_st(&fs->fse_valid, fv); g_st(&fs->huf_valid, hv); g_st(&fs->huf_log, c.huf_log);
                            g_st(&fs->al[0], c.al[0]); g_st(&fs->al[1], c.al[1]); g_st(&fs->al[2], c.al[2]);
                        }
                    }
                    g_settle();
                    __syncthreads();
                    if (tid == 0) g_store(&fs->tables_ver, t + 1);
                }
            }
            if (tid == 0) {
                const uint32_t ri0 = frame_first ? c.rep[0] : c.pred_rep[0], ri1 = frame_first ? c.rep[1] : c.pred_rep[1], ri2 = frame_first ? c.rep[2] : c.pred_rep[2];
                RepOp Rf;
                if (have_block && btype == 2) { Rf.s = c.rep_op[0]; Rf.v0 = (int32_t)c.rep_op[1]; Rf.v1 = (int32_t)c.rep_op[2]; Rf.v2 = (int32_t)c.rep_op[3]; }
                else { Rf.s = 0 | (1 << 2) | (2 << 4); Rf.v0 = 0; Rf.v1 = 0; Rf.v2 = 0; }
                g_st(&fs->rep[0], rep_eval(Rf, 0, ri0, ri1, ri2)); g_st(&fs->rep[1], rep_eval(Rf, 1, ri0, ri1, ri2)); g_st(&fs->rep[2], rep_eval(Rf, 2, ri0, ri1, ri2));
                g_st(&fs->err, (int32_t)final_err);
                g_st(&fs->out, final_err ? pred_out : out_now);
                g_st(&fs->frame_out0, fstart);
            }
            // this task's output bytes must be in memory before the successor (possibly on another XCD) is let go:
            // every wavefront that stored output has waited for its stores (its own fences); one agent-scope release
            // writes the XCD's L2 back
            g_settle();
            __syncthreads();
            if (tid == 0) { g_release(); g_store(&fs->copied, t + 1); }
        } else if (tid == 0) { // the file is finished: its result, and one file less to wait for
            a.jobs[j].out_len = final_err ? pred_out : out_now;
            a.jobs[j].status = final_err;
            atomicAdd(&a.counter[3], 1u);
        }
        TTASK_END();
        STAMP_FLUSH();
        TFIN_FLUSH();
        __syncthreads();
    }
}

// Dictionary (A.7) -> DevDict: the entropy tables in the exact LDS layout, built once on the
// device with the same routines the decoder uses.  One workgroup.
__global__ __launch_bounds__(kWG) void mzd_dict_kernel(const uint8_t* dict, uint32_t n, DevDict* out, int32_t* status) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Ctl& c = S.c;
    __shared__ uint32_t pos_after_huf, pos_after_tables;
    if (tid == 0) {
        c.err = 0; c.action = 0;
        if (n < 8 || ld32(dict) != 0xEC30A437u) c.action = 1; // raw content
        else {
            int used = read_huf_weights(dict + 8, n - 8);
            if (used <= 0) c.err = MZD_E_DICT;
            pos_after_huf = 8 + (uint32_t)(used > 0 ? used : 0);
        }
    }
    __syncthreads();
    if (c.action == 1) {
        if (tid == 0) {
            out->formatted = 0; out->dict_id = 0; out->content = dict; out->content_len = n;
            out->rep[0] = 1; out->rep[1] = 4; out->rep[2] = 8;
            *status = MZD_OK;
        }
        return;
    }
    if (c.err) { if (tid == 0) *status = c.err; return; }
    if (wave == 0) { int hr = finish_huf_table_wave(lane); if (hr && lane == 0) c.err = MZD_E_DICT; }
    __syncthreads();
    if (c.err) { if (tid == 0) *status = c.err; return; }
    if (tid == 0) {
        const uint8_t* p = dict + pos_after_huf;
        const uint8_t* end = dict + n;
        const int order[3] = {1, 2, 0}; // OF, ML, LL (A.7)
        const int max_log[3] = {9, 8, 9}, max_sym[3] = {35, 31, 52};
        for (int k = 0; k < 3 && !c.err; k++) {
            int t = order[k];
            int used = read_ncount(p, (uint32_t)(end - p), max_log[t], max_sym[t], S.norm[t], &c.nsym[t], &c.al[t]);
            if (used <= 0) { c.err = MZD_E_DICT; break; }
            p += used;
            c.mode[t] = 2;
        }
        if (!c.err && (end - p) < 12) c.err = MZD_E_DICT;
        pos_after_tables = (uint32_t)(p - dict);
    }
    __syncthreads();
    if (c.err) { if (tid == 0) *status = MZD_E_DICT; return; }
    if (wave == 0) build_tables_wave(lane);
    __syncthreads();
    for (int i = tid; i < 512; i += kWG) { out->ll[i] = S.ll[i]; out->ml[i] = S.ml[i]; }
    for (int i = tid; i < 256; i += kWG) out->of[i] = S.of[i];
    for (int i = tid; i < 2048; i += kWG) out->huf[i] = S.huf[i];
    if (tid == 0) {
        const uint8_t* p = dict + pos_after_tables;
        uint32_t content = n - pos_after_tables - 12;
        int ok = 1;
        for (int i = 0; i < 3; i++) { uint32_t r = ld32(p + 4 * i); if (r == 0 || r > content) ok = 0; out->rep[i] = r; }
        out->al[0] = c.al[0]; out->al[1] = c.al[1]; out->al[2] = c.al[2];
        out->huf_log = c.huf_log;
        out->dict_id = ld32(dict + 4);
        out->formatted = 1;
        out->content = p + 12;
        out->content_len = content;
        *status = ok ? MZD_OK : MZD_E_DICT;
    }
}

void launch_dict_kernel(const uint8_t* dict, uint32_t n, DevDict* out, int32_t* status, void* stream) {
    hipLaunchKernelGGL(mzd_dict_kernel, dim3(1), dim3(kWG), 0, (hipStream_t)stream, dict, n, out, status);
}

void* decode_kernel_ptr(int tasks) { return tasks ? (void*)mzd_decode_kernel_tasks : (void*)mzd_decode_kernel_files; }

void launch_decode(const KernelArgs& a, uint32_t grid, void* stream) {
    if (a.use_tasks) hipLaunchKernelGGL(mzd_decode_kernel_tasks, dim3(grid), dim3(kWG), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mzd_decode_kernel_files, dim3(grid), dim3(kWG), 0, (hipStream_t)stream, a);
}

int kernel_lds_bytes() { return (int)sizeof(Shared); }

} // namespace mzd
