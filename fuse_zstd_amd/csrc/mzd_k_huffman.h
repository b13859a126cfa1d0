// mzd_k_huffman.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
// ------------------------------------------------------------------------------------ K2
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
constexpr int32_t kSegBits = 64 * 248; // 31 bytes per lane: lane windows fall into different LDS banks (the segment buffer: 2 KiB + 64)
constexpr int32_t kSegBitsHalf = 64 * 120; // ... 15 bytes per lane: a segment buffer of 1 KiB + 64 (wavefront 2 with five workgroups per CU)

#ifndef MZD_HUF_MINC
#define MZD_HUF_MINC 16
#endif
// BIG: a tree of depth 12 (L == 12), which libzstd accepts and no encoder emits: the table is indexed by the top 11 of the next 12
// bits and two codes of length 12 share an entry (mzd_device.h: kHufEntries); four symbols per window instead of five (4 * 12 <= 57).
//
// `below` decides what becomes of a stream that is not consumed exactly (the rule of the reference's pin, libzstd 1.5.x -- restated in
// oracle/zstd_oracle.c: huf_decode_stream, huf_fast_eligible):
//   kHufStrict   libzstd's checked loops (one stream; a four-stream section that is not eligible for its fast loops): the stream needs its end
//                mark and must be consumed exactly (RFC 8878 4.2.2);
//   otherwise    libzstd's fast loops: they decode `nsym` symbols and never look at where the read point ends up -- what is left of the stream is
//                ignored, a last byte of zero is eight data bits, and a stream that runs out reads on into the bytes in front of it, down to the
//                section's first byte (the jump table's), `below` bytes under the stream; needing more than that is corrupt.
constexpr uint32_t kHufStrict = 0xFFFFFFFFu;
template <bool BIG>
__device__ __noinline__ int huf_stream_wave_t(const uint8_t* sp, uint32_t sl, uint8_t* out, uint32_t nsym, uint32_t L, uint8_t* seg, int32_t seg_bits, int lane, uint32_t below) {
    if (sl == 0) return MZD_E_CORRUPT;
    uint32_t last = sp[sl - 1];
    const bool fast = !BIG && below != kHufStrict;
    if (last == 0 && !fast) return MZD_E_CORRUPT;
    const int32_t own_bits = last ? (int32_t)((sl - 1) * 8 + (uint32_t)hibit(last)) : (int32_t)(sl * 8); // the stream's own bits
    if (fast) { sp -= below; sl += below; } // (what may be read: the section from its first byte)
    const int32_t nbits = own_bits + (fast ? (int32_t)(below * 8) : 0);
    const uint32_t mask = (1u << L) - 1;
    const uint16_t* const tab = S.huf;
    const uint32_t lseg = lds_offset_of(seg); // (the staging segment is in LDS: DS instructions -- through the generic pointer every window load is a flat_load)
    int32_t pos = 0;        // bits consumed so far (wave-uniform, exact)
    uint32_t done = 0;      // symbols written so far
    while (pos < nbits && !(fast && done >= nsym)) {
        const int32_t send = pos < own_bits ? own_bits : nbits; // (segments end at the stream's own first bit: a valid stream decodes nothing below it)
        const int32_t s0 = pos, s1 = pos + seg_bits < send ? pos + seg_bits : send;
        // stage stream bytes [blo - 16, bhi): everything the segment can touch (16 bits of slack below it) behind a
        // 16-byte prefix, so that a window may start up to 8 bytes below the lowest byte needed; bytes below the
        // stream start read as zero (bits below bit 0 of a backward stream are zero)
        const int32_t lowbit = nbits - s1 - 16;
        const uint32_t blo = lowbit > 0 ? ((uint32_t)lowbit >> 3) & ~15u : 0u;
        const uint32_t bhi = (uint32_t)((nbits - s0) + 7) >> 3; // <= sl
        for (uint32_t o = (uint32_t)lane * 16; o < bhi - blo + 16; o += 1024) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (blo + o >= 16) __builtin_memcpy(&v, sp + (blo + o - 16), 16); // may over-read <= 15 bytes past the stream (input padding)
            lds_store_u128(lseg + o, v);
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
        auto walk = [&](int32_t from, uint32_t& cnt, uint8_t* dst, uint32_t maxw) -> int32_t { // (maxw: symbols the lane may write, ~0u = all)
            int32_t rem = nbits - from; // bits below the read point
            uint32_t c = 0;
            if (!BIG && maxw == ~0u) {
                // Whole windows of five symbols without the per-symbol boundary test, as long as the window STARTS above the lane's
                // boundary: the last one may pass it -- it is then taken back and left to the checked loop below, which reads the same
                // window.  (Four instruction slots per symbol instead of eleven; the decoder is bound by VALU issue on noisy data.)
                int32_t rem_p = rem;
                uint32_t it = 0;
                for (; it < 52 && rem > lim; it++) { // (a window consumes >= 5 bits of <= 248; the bound only guards a table with a void entry)
                    rem_p = rem;
                    const int32_t bi = (rem - 57) >> 3;
                    const uint64_t W = lds_load_u64(lseg + (uint32_t)(bi + seg_bias));
                    uint32_t t = (uint32_t)(rem - bi * 8) - L - 1; // shift of the next code's window, less one: the index comes out doubled (a byte offset)
                    uint32_t e[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const uint32_t ix2 = (uint32_t)(W >> t) & (mask << 1);
                        e[k] = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(tab) + ix2);
                        t -= e[k] >> 8;
                    }
                    rem = bi * 8 + (int32_t)(t + L + 1);
                    if (dst && rem > lim) { // five symbols, all of them the lane's own: a dword and a byte, wherever they fall (the sectors are completed in L2)
                        const uint32_t four = (e[0] & 0xFF) | (e[1] & 0xFF) << 8 | (e[2] & 0xFF) << 16 | e[3] << 24;
                        __builtin_memcpy(dst + c, &four, 4);
                        dst[c + 4] = (uint8_t)e[4];
                    }
                    c += 5;
                }
                if (it && rem <= lim) { rem = rem_p; c -= 5; } // (the last window reached the boundary: again, symbol by symbol)
            }
            while (rem > lim) {
                const int32_t bi = (rem - 57) >> 3; // window = stream bytes [bi, bi + 8): the 57..64 bits below the read point
                uint64_t W;
                W = lds_load_u64(lseg + (uint32_t)(bi + seg_bias));
                int32_t h = rem - bi * 8;           // read point inside the window
#pragma unroll
                for (int k = 0; k < (BIG ? 4 : 5); k++) {
                    const bool act = rem > lim;
                    const uint32_t v = (uint32_t)(W >> (uint32_t)(h - (int32_t)L)) & mask;
                    uint32_t e = tab[BIG ? v >> 1 : v];
                    uint32_t l = e >> 8;
                    if (BIG && l == 12 && (v & 1)) e = reinterpret_cast<const uint8_t*>(&S.huf[2048])[v >> 1]; // the odd one of two codes of length 12
                    l = l ? l : 1u;
                    if (dst && act && c < maxw) dst[c] = (uint8_t)e; // (one window per walk, but for a tree of depth 12)
                    l = act ? l : 0u;
                    c += act ? 1u : 0u;
                    h -= (int32_t)l;
                    rem -= (int32_t)l;
                }
            }
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
        int32_t exitp = walk(start, cnt, nullptr, ~0u);
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
                    exitp = walk(start, cnt, nullptr, ~0u);
                    if (cnt < 256 && (uint32_t)(exitp - q1) < 15) memo_put(j, (uint32_t)(exitp - q1 + 1) | (cnt << 4));
                }
#if defined(MZD_STAMPS) && !defined(MZD_EXP_ROUNDS)
                if (lane == 0) atomicAdd(&S.c.diag_slow, 1u); // diagnostic: synchronisation rounds that had to walk
#endif
            }
        }
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const int32_t endp = __builtin_amdgcn_readlane(exitp, 63);
        uint32_t dummy;
        if (fast && done + total > nsym) { // the stream's symbols end inside this segment: what is left is ignored.  Never write past the stream's share
            const uint32_t first = done + (incl - cnt); // of the literals: the lane that holds symbol nsym writes its first ones only (symbol by symbol)
            if (first < nsym) walk(start, dummy, out + first, first + cnt <= nsym ? cnt : nsym - first);
            return 0;
        }
        if (done + total > nsym || endp > nbits) return MZD_E_CORRUPT; // never write past this stream's share of the literals; a code that passes the lowest readable bit
        walk(start, dummy, out + done + (incl - cnt), ~0u);
        done += total;
        pos = endp;
    }
    if (done != nsym) return MZD_E_CORRUPT; // (checked loops: pos == nbits here, the stream was consumed exactly; fast loops: the section's bytes ran out)
    return 0;
}
__device__ __forceinline__ int huf_stream_wave(const uint8_t* sp, uint32_t sl, uint8_t* out, uint32_t nsym, uint32_t L, uint8_t* seg, int32_t seg_bits, int lane, uint32_t below) {
    return L == 12 ? huf_stream_wave_t<true>(sp, sl, out, nsym, L, seg, seg_bits, lane, kHufStrict) : huf_stream_wave_t<false>(sp, sl, out, nsym, L, seg, seg_bits, lane, below);
}
