// mzd_k_tables.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
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
// the table's place in S: an entry's low word is the ADDRESS of the next state's base entry (the walker reads with it as it stands)
__device__ __forceinline__ uint32_t table_lds(int kind) { return lds_base() + (kind == 0 ? kLdsLL : (kind == 1 ? kLdsOF : kLdsML)); }
__device__ __forceinline__ uint64_t pack_entry(uint32_t nbase, uint32_t nb, uint32_t s, int kind) {
    uint32_t extra = code_extra(s, kind);
    uint32_t hi = nb | ((extra + nb) << 8) | (s << 16) | (extra << 24);
    return (uint64_t)(table_lds(kind) + nbase * 8u) | ((uint64_t)hi << 32);
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
    if (maxbits > 12) return MZD_E_CORRUPT; // (libzstd's limit, HUF_TABLELOG_MAX; the format's text says 11)
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
    // A tree of depth 12 fills the 2 048 entries as pairs of neighbouring codes (mzd_device.h: kHufEntries): every position and count
    // below is halved, and two codes of length 12 that share an entry put the odd one's symbol behind the table.  (A class starts at a
    // multiple of its symbols' share: the classes behind it are multiples of it and so is their total, a power of two.)
    const uint32_t sh = maxbits == 12 ? 1u : 0u;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t sym = (uint32_t)g * 64 + (uint32_t)lane;
        const uint32_t cnt1 = w[g] ? 1u << (w[g] - 1) : 0u;
        const uint32_t cnt = cnt1 >> sh, a2 = at[g] >> sh;
        const uint32_t e = sym | ((maxbits + 1 - w[g]) << 8);
        if (sh && cnt1 == 1) { if (at[g] & 1) reinterpret_cast<uint8_t*>(&S.huf[2048])[a2] = (uint8_t)sym; else S.huf[a2] = (uint16_t)e; }
        else if (cnt == 1) S.huf[a2] = (uint16_t)e;
        else if (cnt && cnt < 64) { // aligned to cnt (>= 2): pairs
            uint32_t* q = reinterpret_cast<uint32_t*>(&S.huf[a2]);
            for (uint32_t i = 0; i < cnt / 2; i++) q[i] = e | (e << 16);
        }
        uint64_t big = __ballot(cnt >= 64); // few symbols own most of the table: all 64 lanes fill those together
        while (big) {
            const int src = __builtin_ctzll(big);
            const uint32_t a0 = __builtin_amdgcn_readlane(a2, src), c0 = __builtin_amdgcn_readlane(cnt, src), e0 = __builtin_amdgcn_readlane(e, src);
            uint32_t* q = reinterpret_cast<uint32_t*>(&S.huf[a0]);
            for (uint32_t i = (uint32_t)lane; i < c0 / 2; i += 64) q[i] = e0 | (e0 << 16);
            big &= big - 1;
        }
    }
    if (lane == 0) { S.c.huf_nw = nw; S.c.huf_log = maxbits; }
    return 0;
}

