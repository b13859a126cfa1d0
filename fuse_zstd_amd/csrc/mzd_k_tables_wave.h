// mzd_k_tables_wave.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
// ------------------------------------------------------------------------------------ K3 (wave-parallel)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, int lane) {
    (void)lane;
    return wave_incl_scan_op(v, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
}

// FSE decode table (A.3) built by the 64 lanes of one wavefront; same result as build_seq_table.
//   A (lane = symbol)  counts -> low-probability symbols at the top, slot ranges by a scan
//   B (lane = 8 slots) slot -> symbol by a max-scan over range-start marks
//   C (lane = step j)  position (j*step)&mask, ranked among the positions below `high` by ballot
//   D (lane = symbol)  state numbering in table order: every symbol walks the table once
// tmp: 2 KiB of LDS scratch (tabsym[512], mark[512], per-symbol masks / counters / extra-bit counts).
__device__ __noinline__ void build_seq_table_wave(uint64_t* tab, const int16_t* norm, uint32_t nsym, uint32_t log, int kind, uint8_t* tmp, int lane) {
    MZD_IN_LDS(tab); MZD_IN_LDS(norm); MZD_IN_LDS(tmp);
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
                tab[i] = (uint64_t)(table_lds(kind) + ((d << nb) - size) * 8u) | ((uint64_t)hi << 32);
                if ((uint32_t)lane == 63u - (uint32_t)__builtin_clzll(m)) { // the symbol's highest position in this step
                    scnt[sy] = basec + (uint32_t)__builtin_popcountll(m);
                    __atomic_store_n(&smask[sy], 0ull, __ATOMIC_RELAXED);
                }
            }
        }
    }
}

