// mzd_l_tables.h -- the small-file kernel's table builders (included by mzd_lds.hip behind its LDS accessors): FSE decode tables
// (A.3) by all the lanes of a file.  gfx950 only.
//
// Round 3 built a table on ONE lane (two loops over table positions with an LDS round trip each: 59 K cycles for a group's twelve
// tables, 16 K for its four Huffman-weight tables, of a 527 K-cycle group).  Here the file's LPF lanes share the positions:
//
//   step 1, lane = symbol:   positive counts are scanned into rank starts (DPP), every symbol with states marks the rank where its
//                            run starts; a "less than one" symbol takes its position at the top and sets the bit of the spread's
//                            visit that lands there in the SKIP mask (the spread visits position (k * step) mod size at visit k,
//                            step odd: visit k(p) = p * step^-1 mod size, no search);
//   step 2, lane = 8 ranks:  prefix maximum over the marks: R[j] = the symbol of rank j;
//   step 3, lane = position: rank j = k(p) - (skipped visits before k(p): one popcount per 64 visits), symbol = R[j] -> S[p];
//   step 4, lane = position: the numbering, LPF consecutive positions per round in ascending rounds: a position's state number is its
//                            symbol's counter (norm[], bumped once per round by the symbol's first lane) + the number of lower lanes
//                            of the round with the same symbol (a 16-bit lane mask per symbol, OR-ed together in LDS);
//   step 5:                  the "less than one" entries (their places overlap S).
// Work memory: none beyond the table's own place and the file's share of the wavefront's dump area (which only the execution
// uses): R = the table's first half (dwords), the skip mask (size / 8 bytes) behind it, S = its top eighth (bytes); the lane masks
// (53 symbols x LPF bits, packed) = 8 * LPF bytes at `lmo`.  Entries as everywhere (fse_entry).
#pragma once

// step^-1 mod size, step = size/2 + size/8 + 3, for table logs 5..9
DI uint32_t fse_step_inv(uint32_t log) { return log == 5 ? 7u : (log == 6 ? 3u : (log == 7 ? 91u : (log == 8 ? 11u : 363u))); }

// KIND 0 LL, 1 OF, 2 ML, 3 Huffman weights (no extra bits).  act: this file builds a table (uniform in the file); the lanes of
// files that do not run along with stores aimed at `dumpq` (8 bytes of their own).  Returns false when the counts are not a
// distribution over the table (uniform in the file).  5 <= log <= 9, nsym <= 64.
template <int KIND, int LPF>
DI bool build_fse_file(bool act, uint32_t tab, uint32_t norm_off, uint32_t nsym, uint32_t log, uint32_t lmo, uint32_t sub) {
    if (!act) return true;
    const uint32_t size = 1u << log, mask = size - 1;
    const uint32_t inv = fse_step_inv(log);
    const uint32_t skipo = tab + 4 * size; // (size / 8 bytes of the 3 * size between R and S)
    auto extra_of = [&](uint32_t s) -> uint32_t { return KIND == 0 ? L32(kShLL + 4 * s) >> 24 : (KIND == 1 ? s : (KIND == 2 ? L32(kShML + 4 * s) >> 24 : 0u)); };
    // ---- clear: the marks (dwords [0, size)), the skip mask, the lane masks
    if (act) {
        for (uint32_t o = 16 * sub; o < 4 * size; o += 16 * LPF) lds_sv16(tab + o, V16{0, 0});
        for (uint32_t o = 8 * sub; o < (size > 64 ? size >> 3 : 8u); o += 8 * LPF) L64(skipo + o) = 0;
        L64(lmo + 8 * sub) = 0;
    }
    wsync();
    // ---- step 1
    uint32_t carry = 0; // positive counts | "less than one" symbols << 16, over the symbols below the round
    for (uint32_t s0 = 0; s0 < nsym; s0 += LPF) {
        const uint32_t s = s0 + sub;
        const int32_t c = s < nsym ? (int32_t)L16s(norm_off + 2 * s) : 0;
        const uint32_t v = c > 0 ? (uint32_t)c : (c == -1 ? 0x10000u : 0u);
        uint32_t inc = v;
        inc += seg_shr<1, LPF>(inc, sub); inc += seg_shr<2, LPF>(inc, sub);
        if (LPF > 4) inc += seg_shr<4, LPF>(inc, sub);
        if (LPF > 8) inc += seg_shr<8, LPF>(inc, sub);
        const uint32_t before = carry + inc - v;
        carry += bcast<LPF - 1, LPF>(inc);
        if (c > 0 && (before & 0xFFFF) < size) L32(tab + 4 * (before & 0xFFFF)) = s;
        if (c == -1) { // position size - 1 - (its index among the "less than one" symbols): taken out of the spread
            const uint32_t q = (mask - (before >> 16)) & mask;
            const uint32_t k = (q * inv) & mask;
            lds_or32(skipo + 4 * (k >> 5), 1u << (k & 31));
        }
    }
    const uint32_t nlow = carry >> 16, high = size - nlow;
    const bool good = !act || ((carry & 0xFFFF) == high && nlow <= size);
    wsync();
    // ---- step 2: R = prefix maximum of the marks, 8 ranks per lane and round
    {
        uint32_t cmax = 0;
        for (uint32_t r0 = 0; r0 < size; r0 += 8 * LPF) {
            const uint32_t at = tab + 4 * (r0 + 8 * sub);
            const bool in = r0 + 8 * sub < size;
            V16 a = {0, 0}, b = {0, 0};
            if (in) { a = lds_v16(at); b = lds_v16(at + 16); }
            uint32_t m0 = (uint32_t)a.a, m1 = (uint32_t)(a.a >> 32), m2 = (uint32_t)a.b, m3 = (uint32_t)(a.b >> 32);
            uint32_t m4 = (uint32_t)b.a, m5 = (uint32_t)(b.a >> 32), m6 = (uint32_t)b.b, m7 = (uint32_t)(b.b >> 32);
            m1 = max(m1, m0); m2 = max(m2, m1); m3 = max(m3, m2); m4 = max(m4, m3); m5 = max(m5, m4); m6 = max(m6, m5); m7 = max(m7, m6);
            uint32_t sc = m7; // inclusive maximum over the file's lanes
            sc = max(sc, seg_shr<1, LPF>(sc, sub)); sc = max(sc, seg_shr<2, LPF>(sc, sub));
            if (LPF > 4) sc = max(sc, seg_shr<4, LPF>(sc, sub));
            if (LPF > 8) sc = max(sc, seg_shr<8, LPF>(sc, sub));
            const uint32_t below = max(cmax, seg_shr<1, LPF>(sc, sub));
            cmax = max(cmax, bcast<LPF - 1, LPF>(sc));
            if (in) {
                m0 = max(m0, below); m1 = max(m1, below); m2 = max(m2, below); m3 = max(m3, below);
                m4 = max(m4, below); m5 = max(m5, below); m6 = max(m6, below); m7 = max(m7, below);
                lds_sv16(at, V16{(uint64_t)m0 | ((uint64_t)m1 << 32), (uint64_t)m2 | ((uint64_t)m3 << 32)});
                lds_sv16(at + 16, V16{(uint64_t)m4 | ((uint64_t)m5 << 32), (uint64_t)m6 | ((uint64_t)m7 << 32)});
            }
        }
    }
    wsync();
    // ---- step 3: the symbol of every position below `high`
    {
        const uint32_t nw64 = size > 64 ? size >> 6 : 1u;
        for (uint32_t p0 = 0; p0 < high; p0 += LPF) {
            const uint32_t p = p0 + sub;
            const uint32_t k = (p * inv) & mask;
            uint32_t skipped = 0;
            if (nlow) {
                for (uint32_t w = 0; w < nw64; w++) {
                    const uint64_t bw = L64(skipo + 8 * w);
                    const int32_t rel = (int32_t)k - (int32_t)(64 * w);
                    const uint64_t m = rel <= 0 ? 0ull : (rel >= 64 ? ~0ull : ((1ull << rel) - 1));
                    skipped += (uint32_t)__builtin_popcountll(bw & m);
                }
            }
            if (p < high) {
                const uint32_t j = (k - skipped) & mask;
                L8(tab + 7 * size + p) = (uint8_t)L32(tab + 4 * j);
            }
        }
    }
    wsync();
    // ---- step 4: the numbering
    for (uint32_t p0 = 0; p0 < high; p0 += LPF) {
        const uint32_t p = p0 + sub;
        const bool on = p < high;
        const uint32_t s = on ? L8(tab + 7 * size + p) : 0u;
        const uint32_t mo = lmo + 4 * ((s * LPF) >> 5), sh = (s * LPF) & 31;
        if (on) lds_or32(mo, (1u << sub) << sh);
        wsync();
        if (on) {
            const uint32_t lm = (L32(mo) >> sh) & ((1u << LPF) - 1);
            const uint32_t old = L16(norm_off + 2 * s);
            const uint32_t lower = (uint32_t)__builtin_popcount(lm & ((1u << sub) - 1));
            const uint32_t d = old + lower;
            const uint32_t nb = log - (uint32_t)hibit32(d | 1u);
            const uint64_t e = fse_entry(tab, ((d << nb) - size) & mask, nb, s, extra_of(s));
            asm volatile("" ::: "memory");
            L64(tab + 8 * p) = e;
            if (lower == 0) L16(norm_off + 2 * s) = (uint16_t)(old + (uint32_t)__builtin_popcount(lm));
            lds_xor32(mo, (1u << sub) << sh); // (the mask is clean again for the next round)
        }
        wsync();
    }
    // ---- step 5: "less than one": a single state at the top, numbered 1 -> nbBits = log, next-state base 0
    if (nlow) {
        uint32_t lc = 0;
        for (uint32_t s0 = 0; s0 < nsym; s0 += LPF) {
            const uint32_t s = s0 + sub;
            const bool low = s < nsym && L16s(norm_off + 2 * s) == -1;
            uint32_t inc = low ? 1u : 0u;
            inc += seg_shr<1, LPF>(inc, sub); inc += seg_shr<2, LPF>(inc, sub);
            if (LPF > 4) inc += seg_shr<4, LPF>(inc, sub);
            if (LPF > 8) inc += seg_shr<8, LPF>(inc, sub);
            if (low) L64(tab + 8 * ((mask - (lc + inc - 1)) & mask)) = fse_entry(tab, 0, log, s, extra_of(s));
            lc += bcast<LPF - 1, LPF>(inc);
        }
    }
    wsync();
    return good;
}
