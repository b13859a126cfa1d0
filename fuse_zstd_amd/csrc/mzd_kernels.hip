// mzd_kernels.hip -- the zstd frame decoder as HIP kernels for gfx950 (MI355X / CDNA4).
//
// Replaces the arithmetic behind `zstd::stream::copy_decode` (reference src/main.rs:463-467;
// libzstd 1.5.6 via zstd-sys, reference Cargo.lock:2371-2396), written from the format
// (RFC 8878) for 64-lane wavefronts.  Not a port of libzstd.
//
// Mapping: one workgroup (4 wavefronts) owns one file; a persistent grid pulls files from a
// work queue.  Per compressed block:
//   K0  headers                 lane 0                                         (A.1, A.2)
//   K1  Huffman weights/table   lane 0 decodes the weights, 256 lanes fill     (A.4)
//   K2  Huffman literals        one wavefront per stream, 64 lanes per stream by
//                               self-synchronising sub-stream decode + ballot/scan offsets
//   K3  FSE tables x3           three wavefronts, one table each               (A.3)
//   K4  FSE sequence decode     one wavefront, wave-uniform; tables in LDS; the backward
//                               bitstream streamed through an LDS ring by coalesced
//                               16-B/lane loads                                 (A.5)
//   K5  sequence execute        one wavefront, 64 sequences per step: scan for positions,
//                               literal copies, multi-round match resolution    (A.5)
//   K6  raw / RLE blocks        256 lanes, coalesced
//   K7  XXH64                   4 lanes (one per accumulator) + lane 0 tail     (A.6)
// Everything is integer/byte work bound by latency and HBM, so there is no MFMA here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mzd.h"
#include "mzd_device.h"

namespace mzd {

// ------------------------------------------------------------------------------------ LDS
constexpr int kRingBytes = 4096; // sequence-bitstream ring: 4 chunks of 1 KiB
constexpr int kRingDw = kRingBytes / 4;
constexpr int kChunk = 1024;

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
    uint32_t nseq, mode[3], al[3], nsym[3], fse_valid, seq_len;
    uint32_t rep[3];
    uint32_t dict_content_len;
    const uint8_t* dict_content;
};

struct __attribute__((aligned(16))) Shared {
    uint64_t ll[512];   // FSE entries: [31:0] base value, [47:32] next-state base, [55:48] nbBits, [63:56] extra bits
    uint64_t ml[512];
    uint64_t of[256];
    uint32_t ring[kRingDw];
    uint16_t huf[2048]; // sym | len << 8
    int16_t norm[3][64];
    uint16_t next[3][64];
    int16_t wnorm[256]; // FSE table of the Huffman weights
    uint32_t wtab[64];  // sym | nb << 8 | base << 16
    uint8_t weights[256];
    uint32_t rank_start[16];
    Ctl c;
};

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

__device__ const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const int16_t LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ const int16_t ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__device__ const int16_t OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

// ------------------------------------------------------------------------------------ K3
// Normalized-count header (A.3), forward bitstream.  Lane 0.  Returns bytes used or < 0.
__device__ __noinline__ int read_ncount(const uint8_t* src, uint32_t n, int max_log, int max_sym, int16_t* norm, uint32_t* nsym_out, uint32_t* log_out) {
    if (n < 1) return MZD_E_CORRUPT;
    int32_t bit = 0, limit = (int32_t)(n > 4096 ? 4096 : n) * 8;
    int al = 5 + (int)bits_at(src, n, bit, 4);
    bit += 4;
    if (al > max_log) return MZD_E_CORRUPT;
    int remaining = 1 << al, sym = 0;
    while (remaining > 0 && sym <= max_sym) {
        int nb = hibit((uint32_t)(remaining + 1)) + 1;
        if (bit >= limit) return MZD_E_CORRUPT;
        int val = (int)bits_at(src, n, bit, nb);
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
                int r = (int)bits_at(src, n, bit, 2);
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

// Table build (A.3) by one lane.  kind 0 LL, 1 OF, 2 ML selects the code -> (base, extra) map.
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
        uint32_t base, extra;
        if (kind == 0) { base = LL_BASE[s]; extra = LL_BITS[s]; }
        else if (kind == 1) { base = 1u << s; extra = s; }
        else { base = ML_BASE[s]; extra = ML_BITS[s]; }
        tab[i] = (uint64_t)base | ((uint64_t)nbase << 32) | ((uint64_t)nb << 48) | ((uint64_t)extra << 56);
    }
    return 0;
}

__device__ void rle_seq_table(uint64_t* tab, uint32_t s, int kind) {
    uint32_t base, extra;
    if (kind == 0) { base = LL_BASE[s]; extra = LL_BITS[s]; }
    else if (kind == 1) { base = 1u << s; extra = s; }
    else { base = ML_BASE[s]; extra = ML_BITS[s]; }
    tab[0] = (uint64_t)base | ((uint64_t)extra << 56);
}

// ------------------------------------------------------------------------------------ K1
// Huffman tree description (A.4) -> S.weights[0..nw), S.c.huf_log.  Lane 0.  Returns bytes used or < 0.
__device__ __noinline__ int read_huf_weights(Shared& S, const uint8_t* src, uint32_t n) {
    if (n < 1) return MZD_E_CORRUPT;
    uint32_t hb = src[0], nw = 0;
    int used;
    uint8_t* w = S.weights;
    if (hb >= 128) {
        nw = hb - 127;
        uint32_t bytes = (nw + 1) / 2;
        if (1 + bytes > n) return MZD_E_CORRUPT;
        for (uint32_t i = 0; i < nw; i++) {
            uint32_t b = src[1 + i / 2];
            w[i] = (uint8_t)((i & 1) ? (b & 15) : (b >> 4));
        }
        used = 1 + (int)bytes;
    } else {
        if (hb < 1 || 1 + hb > n) return MZD_E_CORRUPT;
        const uint8_t* p = src + 1;
        uint32_t nsym, log;
        int hdr = read_ncount(p, hb, 6, 255, S.wnorm, &nsym, &log);
        if (hdr <= 0) return MZD_E_CORRUPT;
        // tiny FSE table (<= 64 entries) built in place
        uint32_t size = 1u << log, high = size;
        uint16_t* next = (uint16_t*)S.ring; // the ring is idle during literal decoding
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
        const uint8_t* bs = p + hdr;
        uint32_t bl = hb - (uint32_t)hdr;
        if (bs[bl - 1] == 0) return MZD_E_CORRUPT;
        int32_t bpos = (int32_t)(bl - 1) * 8 + hibit(bs[bl - 1]);
        bpos -= (int32_t)log; uint32_t s1 = bits_at(bs, bl, bpos, (int)log);
        bpos -= (int32_t)log; uint32_t s2 = bits_at(bs, bl, bpos, (int)log);
        int ok = 0;
        for (;;) { // two interleaved states; ends when the stream is over-read
            if (nw > 253) break;
            uint32_t e = S.wtab[s1];
            w[nw++] = (uint8_t)e; int nb = (e >> 8) & 0xFF; bpos -= nb; s1 = (e >> 16) + bits_at(bs, bl, bpos, nb);
            if (bpos < 0) { w[nw++] = (uint8_t)S.wtab[s2]; ok = 1; break; }
            if (nw > 253) break;
            e = S.wtab[s2];
            w[nw++] = (uint8_t)e; nb = (e >> 8) & 0xFF; bpos -= nb; s2 = (e >> 16) + bits_at(bs, bl, bpos, nb);
            if (bpos < 0) { w[nw++] = (uint8_t)S.wtab[s1]; ok = 1; break; }
        }
        if (!ok) return MZD_E_CORRUPT;
        used = 1 + (int)hb;
    }
    uint32_t total = 0, rank[13];
    for (int r = 0; r < 13; r++) rank[r] = 0;
    for (uint32_t i = 0; i < nw; i++) {
        uint32_t x = w[i];
        if (x > 12) return MZD_E_CORRUPT;
        rank[x]++;
        if (x) total += 1u << (x - 1);
    }
    if (total == 0) return MZD_E_CORRUPT;
    uint32_t maxbits = (uint32_t)hibit(total) + 1;
    if (maxbits > 11) return MZD_E_CORRUPT;
    uint32_t left = (1u << maxbits) - total;
    if (left & (left - 1)) return MZD_E_CORRUPT;
    uint32_t wl = (uint32_t)hibit(left) + 1;
    w[nw++] = (uint8_t)wl;
    rank[wl]++;
    if (rank[1] < 2 || (rank[1] & 1)) return MZD_E_CORRUPT;
    uint32_t p2 = 0;
    for (uint32_t r = 1; r <= maxbits; r++) { S.rank_start[r] = p2; p2 += rank[r] << (r - 1); }
    if (p2 != (1u << maxbits)) return MZD_E_CORRUPT;
    for (uint32_t i = nw; i < 256; i++) w[i] = 0;
    S.c.huf_nw = nw;
    S.c.huf_log = maxbits;
    return used;
}

// Canonical table fill by all 256 lanes: lane s owns symbol s.
__device__ __noinline__ void fill_huf_table(Shared& S, int tid) {
    uint32_t wt = S.weights[tid];
    if (wt) {
        uint32_t before = 0;
        for (int s = 0; s < tid; s++) before += (S.weights[s] == wt);
        uint32_t cnt = 1u << (wt - 1);
        uint32_t at = S.rank_start[wt] + before * cnt;
        uint16_t e = (uint16_t)((uint32_t)tid | ((S.c.huf_log + 1 - wt) << 8));
        for (uint32_t i = 0; i < cnt; i++) S.huf[at + i] = e;
    }
}

// ------------------------------------------------------------------------------------ K2
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// One Huffman stream decoded by the 64 lanes of a wavefront (A.4).
// Lane k starts at bit k*C of the stream (a guess for k > 0); lanes then re-start from
// their predecessor's exit position until the chain is consistent (Huffman codes
// self-synchronise, so this takes a couple of rounds), a scan of the symbol counts gives
// the output offsets, and a last pass writes.  Returns 0 or an error.
__device__ __noinline__ int huf_stream_wave(const uint8_t* sp, uint32_t sl, uint8_t* out, uint32_t nsym, const uint16_t* tab, uint32_t L, int lane) {
    if (sl == 0) return MZD_E_CORRUPT;
    uint32_t last = sp[sl - 1];
    if (last == 0) return MZD_E_CORRUPT;
    const uint32_t nbits = (sl - 1) * 8 + (uint32_t)hibit(last);
    const uint32_t mask = (1u << L) - 1;
    uint32_t C = (nbits + 63) / 64;
    if (C < 32) C = 32;
    uint32_t q0 = (uint32_t)lane * C, q1 = q0 + C;
    if (q0 > nbits) q0 = nbits;
    if (q1 > nbits) q1 = nbits;
    if (lane == 63) q1 = nbits;

    auto peek = [&](uint32_t pos) -> uint32_t {
        uint32_t rem = nbits - pos;
        if (rem >= L) { uint32_t lo = rem - L; return (ldu32(sp + (lo >> 3)) >> (lo & 7)) & mask; }
        return ((ldu32(sp) & ((1u << rem) - 1)) << (L - rem)) & mask;
    };
    auto span = [&](uint32_t from, uint32_t& cnt) -> uint32_t {
        uint32_t pos = from, c = 0;
        while (pos < q1) { uint32_t l = tab[peek(pos)] >> 8; pos += l ? l : 1u; c++; }
        cnt = c;
        return pos;
    };
    uint32_t start = q0, cnt = 0;
    uint32_t exitp = span(start, cnt);
    for (int round = 0; round < 64; round++) {
        uint32_t pe = __shfl_up(exitp, 1);
        uint32_t ns = lane == 0 ? 0u : pe;
        bool changed = ns != start;
        if (!__any(changed)) break;
        if (changed) { start = ns; exitp = span(start, cnt); }
    }
    uint32_t incl = wave_incl_scan(cnt, lane);
    uint32_t total = __shfl(incl, 63), endp = __shfl(exitp, 63);
    if (total != nsym || endp != nbits) return MZD_E_CORRUPT;
    uint32_t o = incl - cnt, pos = start;
    while (pos < q1) {
        uint32_t e = tab[peek(pos)];
        out[o++] = (uint8_t)e;
        pos += (e >> 8) ? (e >> 8) : 1u;
    }
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
    for (uint32_t i = lane; i < nv; i += 64) {
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
// destination: byte k = pattern[k mod off])
__device__ __noinline__ void wave_pattern(uint8_t* d, uint32_t off, uint32_t n, int lane) {
    const uint8_t* pat = d - off;
    uint32_t idx = (uint32_t)lane % off;
    uint32_t step = 64u % off;
    for (uint32_t k = lane; k < n; k += 64) {
        d[k] = pat[idx];
        idx += step;
        if (idx >= off) idx -= off;
    }
}

// ------------------------------------------------------------------------------------ K4
// The sequence bitstream is read backwards through a 4 KiB LDS ring filled 1 KiB at a time with
// one 16-byte load per lane.  Ring coordinates ("g-offsets") are stream byte index + bias,
// bias = 16 + (sp & 15), so that chunk boundaries are 16-B aligned in HBM and everything below
// the first stream byte reads as zero (bits below bit 0 of a backward stream are zero).
struct SeqStream {
    const uint8_t* gbase; // HBM address of g-offset 0 (16-B aligned; may lie before the buffer, never dereferenced there)
    uint32_t bias;        // g-offset of stream byte 0
    uint32_t gend;        // g-offset one past the last stream byte
    int32_t lowest;       // lowest chunk resident in the ring
};

__device__ void ring_load_chunk(Shared& S, const SeqStream& st, int32_t chunk, int lane) {
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
    *reinterpret_cast<uint4*>(&S.ring[((uint32_t)(chunk & 3) * kChunk + (uint32_t)lane * 16) >> 2]) = v;
}

// 8 bytes ending at g-offset e (exclusive), as a little-endian u64.
__device__ __forceinline__ uint64_t ring_read64(const Shared& S, uint32_t e) {
    uint32_t a = (e - 8) & (kRingBytes - 1);
    uint32_t i = a >> 2, sh = (a & 3) * 8;
    uint32_t d0 = S.ring[i], d1 = S.ring[(i + 1) & (kRingDw - 1)], d2 = S.ring[(i + 2) & (kRingDw - 1)];
    uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);
    return ((uint64_t)hi << 32) | lo;
}

// FSE sequence decode (A.5) by one wavefront, wave-uniform (every lane computes the same values;
// lane 0 stores).  Writes nseq resolved triples {ll, ml, off, 0}.  Returns 0 or an error.
__device__ __noinline__ int decode_sequences_wave(Shared& S, const uint8_t* sp, uint32_t sl, uint32_t nseq, uint4* seqs, int lane) {
    if (sl == 0) return MZD_E_CORRUPT;
    uint32_t last = sp[sl - 1];
    if (last == 0) return MZD_E_CORRUPT;
    SeqStream st;
    uint32_t skew = (uint32_t)((uintptr_t)sp & 15);
    st.bias = 16 + skew;
    st.gbase = sp - st.bias;
    st.gend = sl + st.bias;
    // G = number of g-bits below the read head
    uint64_t G = (uint64_t)(sl - 1) * 8 + (uint32_t)hibit(last) + (uint64_t)st.bias * 8;
    const uint64_t Gzero = (uint64_t)st.bias * 8; // read head at stream bit 0
    int32_t top = (int32_t)((st.gend - 1) / kChunk);
    st.lowest = top;
    ring_load_chunk(S, st, top, lane);
    for (int k = 1; k <= 2 && top - k >= 0; k++) { ring_load_chunk(S, st, top - k, lane); st.lowest = top - k; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const uint32_t alL = S.c.al[0], alO = S.c.al[1], alM = S.c.al[2];
    // initial states
    uint32_t sL, sO, sM;
    {
        uint32_t e = (uint32_t)((G + 7) >> 3);
        uint64_t B = ring_read64(S, e) << (e * 8 - G);
        uint32_t n = alL + alO + alM;
        if (G - Gzero < n) return MZD_E_CORRUPT;
        sL = alL ? (uint32_t)(B >> (64 - alL)) : 0; B <<= alL;
        sO = alO ? (uint32_t)(B >> (64 - alO)) : 0; B <<= alO;
        sM = alM ? (uint32_t)(B >> (64 - alM)) : 0;
        G -= n;
    }
    uint32_t rep0 = S.c.rep[0], rep1 = S.c.rep[1], rep2 = S.c.rep[2];
    int err = 0;
    for (uint32_t i = 0; i < nseq; i++) {
        // keep the ring ahead of the read head (uniform branch)
        uint32_t e = (uint32_t)((G + 7) >> 3);
        if (st.lowest > 0 && (int32_t)e - 160 < st.lowest * kChunk) {
            st.lowest--;
            ring_load_chunk(S, st, st.lowest, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        uint64_t eL = S.ll[sL], eO = S.of[sO], eM = S.ml[sM];
        uint64_t X = ring_read64(S, e);
        uint32_t s0 = (uint32_t)(e * 8 - G);
        uint32_t xO = (uint32_t)(eO >> 56), xM = (uint32_t)(eM >> 56), xL = (uint32_t)(eL >> 56);
        uint32_t nL = (uint32_t)(eL >> 48) & 0xFF, nM = (uint32_t)(eM >> 48) & 0xFF, nO = (uint32_t)(eO >> 48) & 0xFF;
        const bool lastseq = (i + 1 == nseq);
        if (lastseq) { nL = nM = nO = 0; }
        uint32_t total = xO + xM + xL + nL + nM + nO;
        if (G - Gzero < total) { err = MZD_E_CORRUPT; break; }
        uint32_t vO, vM, vL, uL, uM, uO;
        if (s0 + total <= 64) { // one window covers the whole sequence (the common case)
            uint64_t B = X << s0;
            vO = xO ? (uint32_t)(B >> (64 - xO)) : 0; B <<= xO;
            vM = xM ? (uint32_t)(B >> (64 - xM)) : 0; B <<= xM;
            vL = xL ? (uint32_t)(B >> (64 - xL)) : 0; B <<= xL;
            uL = nL ? (uint32_t)(B >> (64 - nL)) : 0; B <<= nL;
            uM = nM ? (uint32_t)(B >> (64 - nM)) : 0; B <<= nM;
            uO = nO ? (uint32_t)(B >> (64 - nO)) : 0;
            G -= total;
        } else { // long extra-bit fields: re-window between the three groups
            uint64_t B = X << s0;
            vO = xO ? (uint32_t)(B >> (64 - xO)) : 0;
            G -= xO;
            uint32_t e2 = (uint32_t)((G + 7) >> 3);
            B = ring_read64(S, e2) << (e2 * 8 - G);
            vM = xM ? (uint32_t)(B >> (64 - xM)) : 0; B <<= xM;
            vL = xL ? (uint32_t)(B >> (64 - xL)) : 0;
            G -= xM + xL;
            e2 = (uint32_t)((G + 7) >> 3);
            B = ring_read64(S, e2) << (e2 * 8 - G);
            uL = nL ? (uint32_t)(B >> (64 - nL)) : 0; B <<= nL;
            uM = nM ? (uint32_t)(B >> (64 - nM)) : 0; B <<= nM;
            uO = nO ? (uint32_t)(B >> (64 - nO)) : 0;
            G -= nL + nM + nO;
        }
        // next states first: they start the next iteration's table reads
        sL = ((uint32_t)(eL >> 32) & 0xFFFF) + uL;
        sM = ((uint32_t)(eM >> 32) & 0xFFFF) + uM;
        sO = ((uint32_t)(eO >> 32) & 0xFFFF) + uO;
        uint32_t ll = (uint32_t)eL + vL, ml = (uint32_t)eM + vM, ofv = (uint32_t)eO + vO;
        uint32_t off;
        if (ofv > 3) { off = ofv - 3; rep2 = rep1; rep1 = rep0; rep0 = off; }
        else {
            uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u);
            if (idx == 0) off = rep0;
            else if (idx == 1) { off = rep1; rep1 = rep0; rep0 = off; }
            else if (idx == 2) { off = rep2; rep2 = rep1; rep1 = rep0; rep0 = off; }
            else { off = rep0 - 1; if (off == 0) { err = MZD_E_CORRUPT; break; } rep2 = rep1; rep1 = rep0; rep0 = off; }
        }
        if (lane == 0) seqs[i] = make_uint4(ll, ml, off, 0);
    }
    if (err) return err;
    if (G != Gzero) return MZD_E_CORRUPT; // the bitstream must be consumed exactly
    if (lane == 0) { S.c.rep[0] = rep0; S.c.rep[1] = rep1; S.c.rep[2] = rep2; }
    wg_fence(); // lane 0's stores of the triples -> the loads of all 64 lanes in execute_wave
    return 0;
}

// ------------------------------------------------------------------------------------ K5
// Sequence execution (A.5) by one wavefront, 64 sequences per step.
//   1. scans of ll and ll+ml give every lane its literal source and output position;
//   2. literal runs are copied (long ones by the whole wave);
//   3. matches are resolved in rounds: a match is ready when its source lies below the
//      output of the first unfinished sequence; ready short matches are copied one per
//      lane, long ones by the whole wave; overlapping matches replicate their pattern.
// dst/frame_start/opos are absolute; returns 0 or an error; *opos_io advances.
constexpr uint32_t kLongCopy = 48;

__device__ __noinline__ int execute_wave(const uint4* seqs, uint32_t nseq, const uint8_t* lit, uint32_t nlit, uint8_t* dst,
                            uint64_t frame_start, uint64_t* opos_io, uint64_t cap, const uint8_t* dict, uint32_t dict_len, int lane) {
    uint64_t opos = *opos_io;
    const uint64_t block_start = opos;
    uint32_t lpos = 0;
    for (uint32_t base = 0; base < nseq; base += 64) {
        uint32_t i = base + (uint32_t)lane;
        bool valid = i < nseq;
        uint4 s = valid ? seqs[i] : make_uint4(0, 0, 0, 0);
        uint32_t ll = s.x, ml = s.y, off = s.z;
        uint32_t incl_t = wave_incl_scan(ll + ml, lane), incl_l = wave_incl_scan(ll, lane);
        uint32_t chunk_tot = __shfl(incl_t, 63), chunk_lit = __shfl(incl_l, 63);
        if (chunk_lit > nlit - lpos) return MZD_E_CORRUPT;
        if ((opos - block_start) + chunk_tot > kBlockMax) return MZD_E_CORRUPT;
        if (chunk_tot > cap - opos) return MZD_E_DSTSIZE;
        uint32_t rel_out = incl_t - (ll + ml);      // relative to opos
        uint32_t my_lit = lpos + (incl_l - ll);
        uint32_t rel_m = rel_out + ll;               // match destination, relative
        uint64_t mdst = opos + rel_m;
        uint64_t avail = (mdst - frame_start) + dict_len;
        bool bad = valid && ml && (off == 0 || off > avail);
        if (__any(bad)) return MZD_E_CORRUPT;
        // ---- literals
        uint64_t longl = __ballot(ll > kLongCopy);
        while (longl) {
            int src_lane = __builtin_ctzll(longl);
            longl &= longl - 1;
            uint32_t l = __shfl(ll, src_lane), ro = __shfl(rel_out, src_lane), lp = __shfl(my_lit, src_lane);
            wave_copy(dst + opos + ro, lit + lp, l, lane);
        }
        if (ll <= kLongCopy) {
            uint8_t* d = dst + opos + rel_out;
            const uint8_t* sp = lit + my_lit;
            for (uint32_t k = 0; k < ll; k++) d[k] = sp[k];
        }
        wg_fence();
        // ---- matches
        bool pending = valid && ml > 0;
        // matches that start inside the dictionary: handled by the owning lane, byte by byte
        if (pending && off > mdst - frame_start) {
            uint64_t back = off - (mdst - frame_start);
            const uint8_t* dp = dict + dict_len - back;
            uint8_t* d = dst + mdst;
            uint32_t k = 0;
            for (; k < ml && k < back; k++) d[k] = dp[k];
            // remainder continues from the start of the frame's own output
            // (source index k - back relative to frame_start), sequential semantics
            for (; k < ml; k++) d[k] = dst[frame_start + (k - back)];
            pending = false;
        }
        int64_t rel_src = (int64_t)rel_m - (int64_t)off; // may be far negative: earlier chunks
        uint32_t span = ml < off ? ml : off;             // bytes of source actually distinct
        uint64_t pm = __ballot(pending);
        while (pm) {
            int first = __builtin_ctzll(pm);
            int64_t hwm = (int64_t)__shfl(rel_m, first);
            bool ready = pending && (rel_src + (int64_t)span <= hwm);
            uint64_t rlong = __ballot(ready && ml > kLongCopy);
            while (rlong) {
                int sl_ = __builtin_ctzll(rlong);
                rlong &= rlong - 1;
                uint32_t m = __shfl(ml, sl_), o = __shfl(off, sl_), rm = __shfl(rel_m, sl_);
                uint8_t* d = dst + opos + rm;
                if (o >= m) wave_copy(d, d - o, m, lane);
                else wave_pattern(d, o, m, lane);
            }
            if (ready && ml <= kLongCopy) {
                uint8_t* d = dst + mdst;
                const uint8_t* sp = d - off;
                uint32_t idx = 0;
                for (uint32_t k = 0; k < ml; k++) {
                    d[k] = sp[idx];
                    idx++;
                    if (idx == off) idx = 0;
                }
            }
            pending = pending && !ready;
            wg_fence();
            pm = __ballot(pending);
        }
        opos += chunk_tot;
        lpos += chunk_lit;
    }
    uint32_t rest = nlit - lpos;
    if (rest > cap - opos) return MZD_E_DSTSIZE;
    if ((opos - block_start) + rest > kBlockMax) return MZD_E_CORRUPT;
    wave_copy(dst + opos, lit + lpos, rest, lane);
    opos += rest;
    wg_fence();
    *opos_io = opos;
    return 0;
}

// ------------------------------------------------------------------------------------ K7
constexpr uint64_t XP1 = 0x9E3779B185EBCA87ull, XP2 = 0xC2B2AE3D27D4EB4Full, XP3 = 0x165667B19E3779F9ull,
                   XP4 = 0x85EBCA77C2B2AE63ull, XP5 = 0x27D4EB2F165667C5ull;
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t h, uint64_t v) { v = xround(0, v); h ^= v; return h * XP1 + XP4; }

// XXH64(seed 0) of p[0..n) by one wavefront: lanes 0..3 carry the four accumulators.
__device__ __noinline__ uint64_t xxh64_wave(const uint8_t* p, uint64_t n, int lane) {
    uint64_t h;
    uint64_t done = 0;
    if (n >= 32) {
        uint64_t v = 0;
        if (lane == 0) v = XP1 + XP2; else if (lane == 1) v = XP2; else if (lane == 2) v = 0; else if (lane == 3) v = 0 - XP1;
        uint64_t stripes = n / 32;
        if (lane < 4) {
            // the accumulator chain is serial; keep 8 stripes of loads in flight ahead of it
            const uint8_t* q = p + lane * 8;
            uint64_t cur[8], nxt[8];
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = (uint64_t)k < stripes ? ldu64(q + (uint64_t)k * 32) : 0;
            for (uint64_t s = 0; s < stripes; s += 8) {
#pragma unroll
                for (int k = 0; k < 8; k++) nxt[k] = s + 8 + k < stripes ? ldu64(q + (s + 8 + k) * 32) : 0;
#pragma unroll
                for (int k = 0; k < 8; k++) if (s + k < stripes) v = xround(v, cur[k]);
#pragma unroll
                for (int k = 0; k < 8; k++) cur[k] = nxt[k];
            }
        }
        done = stripes * 32;
        uint64_t v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = XP5;
    }
    h += n;
    const uint8_t* q = p + done;
    const uint8_t* end = p + n;
    while (q + 8 <= end) { h ^= xround(0, ld64(q)); h = rotl64(h, 27) * XP1 + XP4; q += 8; }
    if (q + 4 <= end) { h ^= (uint64_t)ld32(q) * XP1; h = rotl64(h, 23) * XP2 + XP3; q += 4; }
    while (q < end) { h ^= (uint64_t)(*q) * XP5; h = rotl64(h, 11) * XP1; q++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    return h;
}

// ------------------------------------------------------------------------------------ K0 + block driver
__device__ __noinline__ void parse_frame_or_skip(Shared& S, const uint8_t* src, uint64_t n, const DevDict* dicts, uint32_t ndicts, uint32_t job_dict) {
    Ctl& c = S.c;
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

__device__ __noinline__ void parse_block_header(Shared& S, const uint8_t* src, uint64_t n) {
    Ctl& c = S.c;
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
__device__ __noinline__ void parse_literals(Shared& S, const uint8_t* b, uint32_t n) {
    Ctl& c = S.c;
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
    if (type == 2) {
        int used = read_huf_weights(S, p, rem);
        if (used <= 0) { c.err = MZD_E_CORRUPT; return; }
        p += used; rem -= (uint32_t)used;
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

// sequences section header: nbSeq, modes, table descriptions.  Lane 0.
__device__ __noinline__ void parse_seq_header(Shared& S, const uint8_t* b, uint32_t n) {
    Ctl& c = S.c;
    if (n < 1) { c.err = MZD_E_CORRUPT; return; }
    const uint8_t* p = b;
    const uint8_t* end = b + n;
    uint32_t nseq = *p++;
    if (nseq > 0x7F) {
        if (nseq == 0xFF) { if (p + 2 > end) { c.err = MZD_E_CORRUPT; return; } nseq = ld16(p) + 0x7F00; p += 2; }
        else { if (p + 1 > end) { c.err = MZD_E_CORRUPT; return; } nseq = ((nseq - 0x80) << 8) + *p++; }
    }
    c.nseq = nseq;
    if (nseq == 0) { if (p != end) c.err = MZD_E_CORRUPT; return; }
    if (nseq > kMaxSeq - 1 || p + 1 > end) { c.err = MZD_E_CORRUPT; return; }
    uint32_t modes = *p++;
    if (modes & 3) { c.err = MZD_E_CORRUPT; return; }
    c.mode[0] = modes >> 6; c.mode[1] = (modes >> 4) & 3; c.mode[2] = (modes >> 2) & 3;
    const int max_log[3] = {9, 8, 9}, max_sym[3] = {35, 31, 52};
    for (int t = 0; t < 3; t++) {
        uint32_t m = c.mode[t];
        if (m == 1) {
            if (p + 1 > end || *p > max_sym[t]) { c.err = MZD_E_CORRUPT; return; }
            c.nsym[t] = *p++; // the symbol itself
        } else if (m == 2) {
            int used = read_ncount(p, (uint32_t)(end - p), max_log[t], max_sym[t], S.norm[t], &c.nsym[t], &c.al[t]);
            if (used <= 0) { c.err = MZD_E_CORRUPT; return; }
            p += used;
        } else if (m == 3) {
            if (!c.fse_valid) { c.err = MZD_E_CORRUPT; return; }
        }
    }
    c.seq_off += (uint64_t)(p - b);
    c.seq_len = (uint32_t)(end - p);
}

__device__ __noinline__ void build_tables_wave(Shared& S, int wave, int lane) {
    // wave t builds table t (0 LL, 1 OF, 2 ML); lane 0 of each does the work
    if (wave > 2 || lane != 0) return;
    Ctl& c = S.c;
    int t = wave;
    uint64_t* tab = t == 0 ? S.ll : (t == 1 ? S.of : S.ml);
    uint32_t m = c.mode[t];
    int rc = 0;
    if (m == 0) {
        if (t == 0) { for (int i = 0; i < 36; i++) S.norm[0][i] = LL_DEF[i]; rc = build_seq_table(tab, S.norm[0], S.next[0], 36, 6, 0); c.al[0] = 6; }
        else if (t == 1) { for (int i = 0; i < 29; i++) S.norm[1][i] = OF_DEF[i]; rc = build_seq_table(tab, S.norm[1], S.next[1], 29, 5, 1); c.al[1] = 5; }
        else { for (int i = 0; i < 53; i++) S.norm[2][i] = ML_DEF[i]; rc = build_seq_table(tab, S.norm[2], S.next[2], 53, 6, 2); c.al[2] = 6; }
    } else if (m == 1) {
        rle_seq_table(tab, c.nsym[t], t);
        c.al[t] = 0;
    } else if (m == 2) {
        rc = build_seq_table(tab, S.norm[t], S.next[t], c.nsym[t], c.al[t], t);
    }
    if (rc) c.err = rc;
}

// Control words live in LDS and are written by lane 0 (or one lane per wavefront).  Every
// decision the workgroup takes on them is read through WG_SNAPSHOT: barrier, every lane copies
// the words it needs into registers, barrier -- so no lane can still be reading a word when the
// next step rewrites it, and all 256 lanes always take the same branch.
#define WG_SNAPSHOT(...) do { __syncthreads(); __VA_ARGS__; __syncthreads(); } while (0)

// Diagnostic build only: per-phase cycle sums of the workgroup (lane 0), never in the product .so.
#ifdef MZD_STAMPS
#define STAMP_DECL uint64_t st_prev = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k) do { uint64_t t_ = __builtin_readcyclecounter(); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#define STAMP_FLUSH() do { if (tid == 0 && a.debug) for (int k_ = 0; k_ < 8; k_++) a.debug[blockIdx.x].stamp[k_] = st_acc[k_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH()
#endif

__global__ __launch_bounds__(kWG, 4) void mzd_decode_kernel(KernelArgs a) {
    __shared__ Shared S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* const lit_buf = a.lit_scratch + (size_t)blockIdx.x * kLitStride;
    uint4* const seqs = a.seq_scratch + (size_t)blockIdx.x * kSeqStride;
    Ctl& c = S.c;

    for (;;) {
        if (tid == 0) c.job = atomicAdd(a.counter, 1u);
        uint32_t j;
        WG_SNAPSHOT(j = c.job);
        if (j >= a.njobs) break;
        const uint8_t* const src = a.jobs[j].src;
        const uint64_t n = a.jobs[j].src_len;
        uint8_t* const dst = a.jobs[j].dst;
        const uint64_t cap = a.jobs[j].dst_cap;
        const uint32_t job_dict = a.jobs[j].dict;
        if (tid == 0) {
            c.pos = 0; c.out = 0; c.err = 0; c.action = 0;
            if (j == 0 && a.job_slot0) *a.job_slot0 = blockIdx.x;
            if (job_dict > a.ndicts) c.err = MZD_E_DICT;
        }
        int err = 0;
        uint32_t action = 0;
        STAMP_DECL;

        // ---------------- frames (K0)
        while (true) {
            if (tid == 0 && !c.err) parse_frame_or_skip(S, src, n, a.dicts, a.ndicts, job_dict);
            WG_SNAPSHOT(err = c.err; action = c.action);
            if (err || action == 2) break;
            if (action == 1) continue; // skippable frame
            if (action == 3) { // dictionary: entropy tables, repeat offsets and content
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
                if (tid == 0) { c.dict_content = dd->content; c.dict_content_len = dd->content_len; }
            }

            // ---------------- blocks
            uint32_t last = 0;
            while (true) {
                if (tid == 0) parse_block_header(S, src, n);
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
                    if (tid == 0) parse_literals(S, blk, bsize); // K1 (weights)
                    uint32_t lit_type = 0, nlit = 0, streams = 0, huf_log = 0;
                    uint64_t lit_off = 0;
                    WG_SNAPSHOT(err = c.err; lit_type = c.lit_type; nlit = c.nlit; streams = c.streams; huf_log = c.huf_log; lit_off = c.lit_off);
                    if (err) break;
                    STAMP(1);
                    if (lit_type == 2) { // K1 (table)
                        fill_huf_table(S, tid);
                        if (tid == 0) c.huf_valid = 1;
                        __syncthreads();
                    }
                    STAMP(2);
                    // K2: literals; lane 0 then parses the sequences header (K3)
                    const uint8_t* lit = lit_buf;
                    if (lit_type == 0) lit = src + lit_off;
                    else if (lit_type == 1) wg_fill(lit_buf, src[lit_off], nlit, tid);
                    else if ((uint32_t)wave < streams) {
                        int rc = huf_stream_wave(blk + c.s_off[wave], c.s_len[wave], lit_buf + c.s_out[wave], c.s_n[wave], S.huf, huf_log, lane);
                        if (rc && lane == 0) c.err = rc;
                    }
                    if (tid == 0) parse_seq_header(S, src + c.seq_off, c.seq_len);
                    uint32_t nseq = 0, seq_len = 0;
                    uint64_t seq_off = 0;
                    WG_SNAPSHOT(err = c.err; nseq = c.nseq; seq_off = c.seq_off; seq_len = c.seq_len);
                    if (err) break;
                    STAMP(3);
                    if (nseq) { // K3 tables
                        build_tables_wave(S, wave, lane);
                        WG_SNAPSHOT(err = c.err);
                        if (err) break;
                    }
                    STAMP(4);
                    if (wave == 0) { // K4 + K5 on one wavefront
                        int rc = 0;
                        uint64_t opos = out0;
                        if (nseq) rc = decode_sequences_wave(S, src + seq_off, seq_len, nseq, seqs, lane);
                        STAMP(5);
                        if (!rc) rc = execute_wave(seqs, nseq, lit, nlit, dst, c.frame_out0, &opos, cap, c.dict_content, c.dict_content_len, lane);
                        if (lane == 0) {
                            if (rc) c.err = rc;
                            if (nseq) c.fse_valid = 1;
                            c.out = opos; c.pos = pos0 + bsize;
                            STAMP(6);
                            if (a.debug) {
                                DebugSlot& ds = a.debug[blockIdx.x];
                                ds.n_lit = nlit; ds.n_seq = nseq; ds.lit_is_raw = lit_type == 0; ds.lit_raw_ptr = (uint64_t)(uintptr_t)lit;
                            }
                        }
                    }
                }
                WG_SNAPSHOT(err = c.err);
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
                if (wave == 0) {
                    uint64_t h = xxh64_wave(dst + fout0, out_now - fout0, lane);
                    if (lane == 0) {
                        if ((uint32_t)h != ld32(src + pos_now)) c.err = MZD_E_CHECKSUM;
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
        __syncthreads();
    }
}

// Dictionary (A.7) -> DevDict: the entropy tables in the exact LDS layout, built once on the
// device with the same routines the decoder uses.  One workgroup.
__global__ __launch_bounds__(kWG) void mzd_dict_kernel(const uint8_t* dict, uint32_t n, DevDict* out, int32_t* status) {
    __shared__ Shared S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Ctl& c = S.c;
    __shared__ uint32_t pos_after_huf, pos_after_tables;
    if (tid == 0) {
        c.err = 0; c.action = 0;
        if (n < 8 || ld32(dict) != 0xEC30A437u) c.action = 1; // raw content
        else {
            int used = read_huf_weights(S, dict + 8, n - 8);
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
    fill_huf_table(S, tid);
    __syncthreads();
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
    build_tables_wave(S, wave, lane);
    __syncthreads();
    if (c.err) { if (tid == 0) *status = MZD_E_DICT; return; }
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

void* decode_kernel_ptr() { return (void*)mzd_decode_kernel; }

void launch_decode(const KernelArgs& a, uint32_t grid, void* stream) {
    hipLaunchKernelGGL(mzd_decode_kernel, dim3(grid), dim3(kWG), 0, (hipStream_t)stream, a);
}

int kernel_lds_bytes() { return (int)sizeof(Shared); }

} // namespace mzd
