// mzd_k_resolve.h -- part of the block-task driver of mzd_kernels.hip.  Included there, inside namespace mzd, behind the block
// pipeline; not a translation unit of its own.
#pragma once
// ------------------------------------------------------------------------------------ K5, resolved ahead of the predecessor
// A multi-block frame is entropy-decoded block-parallel but executed in order, and sequence execution is a chain of
// dependent copies (record after record in JSON): ~0.45 ms per 128 KiB block for the copying wavefront, the whole
// critical path of a big file on an otherwise idle machine.  When a launch has few tasks for the machine
// (KernelArgs::resolve) the chain is taken out of the in-order stage:
//   1. as soon as a block's plan is complete -- long before its predecessor has finished -- the workgroup writes a BYTE MAP
//      of the block: for every output byte its source, which is a literal (index into the block's literals), a byte OLDER
//      than the block (distance before the block's start), or an earlier byte of the block itself (position);
//   2. pointer jumping (map[p] = map[map[p]] while that is a position in the block) removes the third kind in
//      log2(longest chain) rounds of 256 independent lanes -- no data is touched, only indices;
//   3. in task order, the block is then a GATHER: every byte comes from the literals or from output that is complete.
// The repeat offsets a block starts with travel on a chain of their own (FileState::rep_ver), handed on as soon as a
// task's plan is complete, so that step 1 never meets a symbolic offset.
// Anything unusual -- an error, a dictionary reference, a block that does not fit, rounds that do not converge --
// falls back to the copying wavefront (copy_wave), which reports errors in the reference's order: nothing has been
// written when the decision is taken.
constexpr uint32_t kResLit = 0x80000000u;   // map entry: literal, bits 0-29 index into the block's literals
constexpr uint32_t kResPrev = 0x40000000u;  // map entry: output older than the block, bits 0-29 = distance before the block start - 1
constexpr uint32_t kResIdx = 0x3FFFFFFFu;
constexpr uint32_t kResRounds = 24;         // > log2(128 Ki) + slack (reads race with writes of the same round, which only helps)

// Step 1, lane = sequence, a chunk of 64 sequences per call.  cbase[k] = {output, literals} before chunk k (plan_wave).
// Returns false -- and writes nothing -- when the chunk holds an offset that is still symbolic and `rep_known` is not set.
// bad: an offset of 0, or a sequence that lies outside the block, was seen (the copier must give the verdict); maxprev: the largest distance a match reaches before the
// block start.
__device__ __forceinline__ bool resolve_build_chunk(uint32_t* map, const uint4* plan, const uint4* cbase, uint32_t chunk, uint32_t nseq, bool rep_known,
                                                    uint32_t rep0, uint32_t rep1, uint32_t rep2, uint4* plan_wb, int lane, uint32_t& bad, uint32_t& maxprev) {
    const uint32_t i = chunk * 64 + (uint32_t)lane;
    bool valid = i < nseq;
    const PlanEnt pe = valid ? plan_of(plan)[i] : make_uint2(0, 1);
    uint32_t ll, ml, off;
    plan_expand(plan_of(plan), i, pe, ll, ml, off);
    if (__any(valid && (off & kOffTag) != 0)) { // start slot + delta (plan_wave)
        if (!rep_known) return false;
        if (off & kOffTag) {
            off = (uint32_t)sel3((off >> 29) & 3, (int32_t)rep0, (int32_t)rep1, (int32_t)rep2) + (off & 0x1FFFFFFFu) - (uint32_t)kOffBias;
            if (plan_wb && valid) plan_of(plan_wb)[i].y = off;
        }
    }
    const uint4 cb = cbase[chunk];
    const uint32_t ex_t = wave_incl_scan(ll + ml, lane) - ll - ml; // (the sequence's output offset inside the chunk)
    const uint32_t p_l = cb.x + ex_t, p_m = p_l + ll;
    const uint32_t li = cb.y + (wave_incl_scan(ll, lane) - ll);
    // A plan made of garbage passes 128 KiB in the chunk the planner marks (Ctl::plan_too_long) -- which is public all the same, and
    // may be seen here before the mark: a sequence that does not lie inside the block writes nothing (the map's slot ends there,
    // the next workgroup's begins), and the copier gives the verdict.  (No overflow: ll, ml < 2^18, 64 of them in a chunk.)
    const bool valid_in = valid;
    valid = valid && p_m + ml <= kBlockMax;
    if (valid_in && (!valid || off == 0 || off >= (1u << 30))) bad = 1;
    if (valid && ml && off > p_m) { const uint32_t d = off - p_m; maxprev = d > maxprev ? d : maxprev; }
    // short pieces: every lane its own, four entries per store (16 bytes, 4-byte aligned), as many steps as the longest needs
    typedef __attribute__((address_space(1))) uint32_t* gmap;
    auto ent = [&](uint32_t pos) -> uint32_t { const int32_t q = (int32_t)pos - (int32_t)off; return q >= 0 ? (uint32_t)q : (kResPrev | (uint32_t)(-q - 1)); };
    {
        const uint32_t n = (valid && ll <= 64) ? ll : 0;
        gmap const m = (gmap)(map + p_l);
        for (uint32_t k = 0; __any(k < n); k += 4) {
            const uint32_t e = kResLit | (li + k);
            if (k + 4 <= n) { const uint4 v = make_uint4(e, e + 1, e + 2, e + 3); __builtin_memcpy(m + k, &v, 16); }
            else if (k < n) { m[k] = e; if (k + 1 < n) m[k + 1] = e + 1; if (k + 2 < n) m[k + 2] = e + 2; }
        }
    }
    {
        const uint32_t n = (valid && ml <= 64) ? ml : 0;
        gmap const m = (gmap)(map + p_m);
        for (uint32_t k = 0; __any(k < n); k += 4) {
            const uint32_t e0 = ent(p_m + k), e1 = ent(p_m + k + 1), e2 = ent(p_m + k + 2), e3 = ent(p_m + k + 3);
            if (k + 4 <= n) { const uint4 v = make_uint4(e0, e1, e2, e3); __builtin_memcpy(m + k, &v, 16); }
            else if (k < n) { m[k] = e0; if (k + 1 < n) m[k + 1] = e1; if (k + 2 < n) m[k + 2] = e2; }
        }
    }
    // long pieces: one after the other, by the whole wavefront
    uint64_t lm = __ballot(valid && ll > 64);
    while (lm) {
        const int sl = __builtin_ctzll(lm);
        const uint32_t n = __builtin_amdgcn_readlane(ll, sl), p = __builtin_amdgcn_readlane(p_l, sl), l0 = __builtin_amdgcn_readlane(li, sl);
        for (uint32_t k = (uint32_t)lane; k < n; k += 64) map[p + k] = kResLit | (l0 + k);
        lm &= lm - 1;
    }
    uint64_t mm = __ballot(valid && ml > 64);
    while (mm) {
        const int sl = __builtin_ctzll(mm);
        const uint32_t n = __builtin_amdgcn_readlane(ml, sl), p = __builtin_amdgcn_readlane(p_m, sl), o = __builtin_amdgcn_readlane(off, sl);
        for (uint32_t k = (uint32_t)lane; k < n; k += 64) { const int32_t q = (int32_t)(p + k) - (int32_t)o; map[p + k] = q >= 0 ? (uint32_t)q : (kResPrev | (uint32_t)(-q - 1)); }
        mm &= mm - 1;
    }
    return true;
}
__device__ __forceinline__ void resolve_build_post(uint32_t bad, uint32_t maxprev, int lane) { // a wavefront's findings -> S.res[0], S.res[1]
    if (__any(bad != 0) && lane == 0) __atomic_fetch_or(&S.res[0], 1u, __ATOMIC_RELAXED);
    for (int sh = 32; sh; sh >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)maxprev, sh); maxprev = o > maxprev ? o : maxprev; }
    if (lane == 0) __atomic_fetch_max(&S.res[1], maxprev, __ATOMIC_RELAXED);
}

// Step 1 BEHIND THE PLANNER, by the two wavefronts that have nothing to do while the block's chain is walked (the copying
// and the hashing one; which = 0 / 1): chunk after chunk as the plan is published.  A chunk with a symbolic offset (the
// block's first sequences, typically) is noted in S.res_sym and left to resolve_build_rest.  Ends when every chunk is
// built, or the planner has finished without publishing the chunk (it failed or cut the plan: the caller falls back).
__device__ __noinline__ void resolve_build_follow(uint32_t* map, const uint4* plan, const uint4* cbase, uint32_t nseq_in, uint32_t which, int lane) {
    const uint32_t nseq = __builtin_amdgcn_readfirstlane(nseq_in);
    const uint32_t nchunks = (nseq + 63) / 64;
    uint32_t bad = 0, maxprev = 0;
    for (uint32_t chunk = which; chunk < nchunks; chunk += 2) {
        bool have = false;
        for (uint32_t it = 0; it < (1u << 24); it++) { // chunk `chunk` is public once the planner has started on chunk + 1 (or has finished)
            const uint32_t pg = flag_load(&S.c.plan_prog);
            if ((pg & ~kPlanFin) > chunk) { have = true; break; }
            if (pg & kPlanFin) { have = (pg & ~kPlanFin) > chunk && !__atomic_load_n(&S.c.err, __ATOMIC_RELAXED) && !__atomic_load_n(&S.c.plan_too_long, __ATOMIC_RELAXED) && !__atomic_load_n(&S.c.walk_inexact, __ATOMIC_RELAXED); break; }
            if (__atomic_load_n(&S.c.err, __ATOMIC_RELAXED)) break;
            __builtin_amdgcn_s_sleep(8);
            if (it == (1u << 24) - 1 && lane == 0) DEVSITE(13);
        }
        if (!have) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (!resolve_build_chunk(map, plan, cbase, chunk, nseq, false, 0, 0, 0, nullptr, lane, bad, maxprev) && lane == 0) {
            const uint32_t at = __atomic_fetch_add(&S.res_nsym, 1u, __ATOMIC_RELAXED);
            if (at < kResSymMax) S.res_sym[at] = chunk;
        }
    }
    resolve_build_post(bad, maxprev, lane);
}
// ... and what it left, once the repeat offsets the block starts with are known (all four wavefronts, after a barrier): the
// noted chunks (all chunks if there were more than the list holds), the literals behind the last sequence, the padding.
__device__ __noinline__ void resolve_build_rest(uint32_t* map, const uint4* plan, const uint4* cbase, uint32_t nseq_in, uint32_t out_seqs, uint32_t lit_used,
                                                uint32_t nlit, uint32_t rep0, uint32_t rep1, uint32_t rep2, uint4* plan_wb, int lane, int wave) {
    const uint32_t nseq = __builtin_amdgcn_readfirstlane(nseq_in);
    const uint32_t nchunks = (nseq + 63) / 64;
    const uint32_t nsym = S.res_nsym;
    uint32_t bad = 0, maxprev = 0;
    if (nsym > kResSymMax) { for (uint32_t chunk = (uint32_t)wave; chunk < nchunks; chunk += 4) resolve_build_chunk(map, plan, cbase, chunk, nseq, true, rep0, rep1, rep2, plan_wb, lane, bad, maxprev); }
    else for (uint32_t k = (uint32_t)wave; k < nsym; k += 4) resolve_build_chunk(map, plan, cbase, S.res_sym[k], nseq, true, rep0, rep1, rep2, plan_wb, lane, bad, maxprev);
    if (wave == 0) { // the literals behind the last sequence, and the padding up to a whole 16 bytes of entries
        const uint32_t rest = nlit - lit_used;
        for (uint32_t k = (uint32_t)lane; k < rest; k += 64) map[out_seqs + k] = kResLit | (lit_used + k);
        if (lane < 4) map[out_seqs + rest + (uint32_t)lane] = kResLit;
    }
    resolve_build_post(bad, maxprev, lane);
}

// Step 2, all 256 threads (workgroup barriers inside), TILE BY TILE THROUGH LDS.  A round over the whole map in HBM/L2 is bound
// by the CU's address path (~3 scattered dwords per clock: 45 K cycles per round, ~10 rounds); but entries only ever refer to
// LOWER positions, so the block can be taken in ascending tiles: entries that refer below the tile take the (final) value of
// their target with one global load, the rest is pointer jumping inside the tile -- in LDS, where a scattered read costs a
// few cycles.  The tile lives in the front of the workgroup's LDS image (sequence ring, FSE tables, staging buffers: all
// dead once the block's walk, plan and literals are complete).  true: no entry refers to the block any more.
constexpr uint32_t kResTile = MZD_WGS_PER_CU >= 5 ? 5120 : 6144; // entries (20 / 24 KiB: what lies in front of Shared::ll_base)
__device__ __noinline__ bool resolve_jump_tiled(uint32_t* map, uint32_t B, int tid) {
    static_assert(offsetof(Shared, ring) == 0 && offsetof(Shared, ll_base) >= kResTile * 4, "the tile overlays ring, ll, ml, of, stage, hseg2");
    uint32_t* const tile = reinterpret_cast<uint32_t*>(&S);
    const uint32_t nall = (B + 3) & ~3u; // (the padding entries behind B are literals)
    bool good = true;
    constexpr uint32_t kPer = kResTile / (kWG * 4); // vectors per thread and tile
    auto load_tile = [&](uint32_t a0, uint4 (&r)[kPer]) { // this thread's vectors of the tile at a0 (literals past the map's end)
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) { const uint32_t i = a0 + (u * kWG + (uint32_t)tid) * 4; r[u] = i < nall ? *reinterpret_cast<const uint4*>(map + i) : make_uint4(kResLit, kResLit, kResLit, kResLit); }
    };
    uint4 v[kPer], nx[kPer];
    load_tile(0, nx);
    for (uint32_t a0 = 0; a0 < nall; a0 += kResTile) {
        const uint32_t n = nall - a0 < kResTile ? nall - a0 : kResTile; // a multiple of 4
        // into LDS; what refers below the tile is final there: one hop (all loads of a thread in flight together; the NEXT tile's
        // vectors -- nobody writes them before their turn -- are requested before this tile's work and arrive during it)
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) v[u] = nx[u];
        load_tile(a0 + kResTile, nx);
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) { // (positions are < 2^17: flagged entries are never below a0)
            const uint32_t wx = v[u].x < a0 ? map[v[u].x] : v[u].x, wy = v[u].y < a0 ? map[v[u].y] : v[u].y;
            const uint32_t wz = v[u].z < a0 ? map[v[u].z] : v[u].z, ww = v[u].w < a0 ? map[v[u].w] : v[u].w;
            v[u] = make_uint4(wx, wy, wz, ww);
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) { const uint32_t i = (u * kWG + (uint32_t)tid) * 4; if (i < n) *reinterpret_cast<uint4*>(tile + i) = v[u]; }
        __syncthreads();
        // pointer jumping inside the tile.  A thread keeps its entries in registers and hops each one along its chain until it
        // is final: every read lands on a lower position or on a final value, so this ends after at most as many steps as the
        // chain is long, and in about log2 of that since the other threads' entries move on at the same time (they publish every
        // step; a stale read is still an ancestor).  No barrier, no flag.
        uint32_t step = 0;
        for (;; step++) {
            bool mine = false;
#pragma unroll
            for (uint32_t u = 0; u < kPer; u++) mine |= (v[u].x < kResPrev) | (v[u].y < kResPrev) | (v[u].z < kResPrev) | (v[u].w < kResPrev);
            if (!__any(mine) || step >= 4 * kResRounds) break;
#pragma unroll
            for (uint32_t u = 0; u < kPer; u++) { // (entries of vectors past the tile's end are literals)
                const uint32_t wx = v[u].x < kResPrev ? tile[v[u].x - a0] : v[u].x, wy = v[u].y < kResPrev ? tile[v[u].y - a0] : v[u].y;
                const uint32_t wz = v[u].z < kResPrev ? tile[v[u].z - a0] : v[u].z, ww = v[u].w < kResPrev ? tile[v[u].w - a0] : v[u].w;
                v[u] = make_uint4(wx, wy, wz, ww);
            }
#pragma unroll
            for (uint32_t u = 0; u < kPer; u++) { const uint32_t i = (u * kWG + (uint32_t)tid) * 4; if (i < n) *reinterpret_cast<uint4*>(tile + i) = v[u]; }
        }
        good = good && step < 4 * kResRounds;
#pragma unroll
        for (uint32_t u = 0; u < kPer; u++) { const uint32_t i = (u * kWG + (uint32_t)tid) * 4; if (i < n) *reinterpret_cast<uint4*>(map + a0 + i) = v[u]; }
        wg_fence(); // the next tile reads these entries
        __syncthreads(); // (and every wavefront is done with this tile's LDS image)
    }
    // `good` is a wavefront's own finding (its entries' chains): the caller branches on the answer around workgroup barriers, so it
    // must be the same in all four.  (A consistent map always settles; one that does not has been written to from outside.)
    if (!good && (tid & 63) == 0) DEVSITE(15);
    // (an AND over the workgroup through a word of the image: __syncthreads_and keeps a static LDS object of its own, and the image must
    //  be the kernel's only one -- it starts at LDS address 0)
    if (tid == 0) S.and_word = 1u;
    __syncthreads();
    if (!good && (tid & 63) == 0) __atomic_store_n(&S.and_word, 0u, __ATOMIC_RELAXED);
    __syncthreads();
    const bool all_good = flag_load_u(&S.and_word) != 0;
    __syncthreads();
    return all_good;
}

// Step 2, all 256 threads (workgroup barriers inside).  true: no entry refers to the block any more.
__device__ __noinline__ bool resolve_jump(uint32_t* map, uint32_t B, int tid) {
    const uint32_t n4 = (B + 3) / 4;
    uint4* const m4 = reinterpret_cast<uint4*>(map);
    for (uint32_t round = 0; round < kResRounds; round++) {
        if (tid == 0) S.res[2] = 0;
        __syncthreads();
        uint32_t open = 0;
        // four vectors of four entries per step: their loads, then their (up to 16) dependent loads, are in flight together (more
        // changes nothing: a round is bound by the CU's address path, ~3 scattered dwords per clock)
        constexpr uint32_t U = 4;
        for (uint32_t q0 = (uint32_t)tid; q0 < n4; q0 += U * kWG) {
            uint4 v[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) { const uint32_t q = q0 + u * kWG; v[u] = q < n4 ? m4[q] : make_uint4(kResLit, kResLit, kResLit, kResLit); }
            uint4 w[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                w[u].x = v[u].x < kResPrev ? map[v[u].x] : v[u].x; w[u].y = v[u].y < kResPrev ? map[v[u].y] : v[u].y;
                w[u].z = v[u].z < kResPrev ? map[v[u].z] : v[u].z; w[u].w = v[u].w < kResPrev ? map[v[u].w] : v[u].w;
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const bool any = (v[u].x < kResPrev) | (v[u].y < kResPrev) | (v[u].z < kResPrev) | (v[u].w < kResPrev);
                if (any) { // (vectors past the end hold no such entry)
                    m4[q0 + u * kWG] = w[u];
                    open |= (uint32_t)(w[u].x < kResPrev) | (uint32_t)(w[u].y < kResPrev) | (uint32_t)(w[u].z < kResPrev) | (uint32_t)(w[u].w < kResPrev);
                }
            }
        }
        if (open) __atomic_store_n(&S.res[2], 1u, __ATOMIC_RELAXED);
        wg_fence();
        __syncthreads();
        const bool done = __atomic_load_n(&S.res[2], __ATOMIC_RELAXED) == 0;
        __syncthreads();
        if (done) return true;
    }
    return false;
}

// Step 3, all 256 threads, in task order: out[out0 + p] = literal or older output.  The caller has checked that the block
// fits and that no match reaches before the frame.
__device__ __noinline__ void resolve_gather(const uint32_t* map, uint32_t B, const uint8_t* lit, uint8_t* dst, uint64_t out0, int tid) {
    const uint4* const m4 = reinterpret_cast<const uint4*>(map);
    const uint8_t* const hist = dst + out0 - 1; // hist[-d]: d bytes before the block's last older byte
    uint8_t* const o = dst + out0;
    const uint32_t n4 = (B + 3) / 4;
    auto byte_of = [&](uint32_t e) -> uint32_t {
        const uint32_t ix = e & kResIdx;
        return (e & kResLit) ? (uint32_t)lit[ix] : (uint32_t)*(hist - ix);
    };
    for (uint32_t q0 = (uint32_t)tid; q0 < n4; q0 += 4 * kWG) { // four dwords per step: 16 byte loads in flight together
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t q = q0 + (uint32_t)u * kWG; v[u] = q < n4 ? m4[q] : make_uint4(kResLit, kResLit, kResLit, kResLit); }
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = byte_of(v[u].x) | (byte_of(v[u].y) << 8) | (byte_of(v[u].z) << 16) | (byte_of(v[u].w) << 24);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = (q0 + (uint32_t)u * kWG) * 4;
            if (p + 4 <= B) __builtin_memcpy(o + p, &w[u], 4);
            else for (uint32_t k = 0; p + k < B; k++) o[p + k] = (uint8_t)(w[u] >> (8 * k));
        }
    }
}

// Step 3 of a frame with a checksum: the XXH64 chain (46 cycles per 32-byte stripe, strictly serial through the frame) is the
// longest thing left in the in-order stage, so it runs BESIDE the gather: wavefronts 0, 1 and 3 gather in ascending steps
// of kResStep bytes and publish their progress (S.res_prog), wavefront 2 hashes behind the slowest of them.
constexpr uint32_t kResU = 8;                          // dwords per lane and step (4 and 16 measured: no better)
constexpr uint32_t kResStepDw = 3 * 64 * kResU;        // dwords per step of the three gathering wavefronts
constexpr uint32_t kResStep = kResStepDw * 4;          // bytes
__device__ __noinline__ void resolve_gather3(const uint32_t* map, uint32_t B, const uint8_t* lit, uint8_t* dst, uint64_t out0, int gw, int lane) {
    const uint4* const m4 = reinterpret_cast<const uint4*>(map);
    const uint8_t* const hist = dst + out0 - 1;
    uint8_t* const o = dst + out0;
    const uint32_t n4 = (B + 3) / 4;
    auto byte_of = [&](uint32_t e) -> uint32_t {
        const uint32_t ix = e & kResIdx;
        return (e & kResLit) ? (uint32_t)lit[ix] : (uint32_t)*(hist - ix);
    };
    uint32_t step = 0;
    const uint32_t qbase = (uint32_t)gw * 64 + (uint32_t)lane;
    auto load_step = [&](uint32_t q0, uint4 (&r)[kResU]) {
#pragma unroll
        for (uint32_t u = 0; u < kResU; u++) { const uint32_t q = q0 + u * 192; r[u] = q < n4 ? m4[q] : make_uint4(kResLit, kResLit, kResLit, kResLit); }
    };
    uint4 v[kResU], nx[kResU];
    load_step(qbase, nx);
    for (uint32_t q0 = qbase; q0 - qbase < n4; q0 += kResStepDw, step++) {
#pragma unroll
        for (uint32_t u = 0; u < kResU; u++) v[u] = nx[u];
        load_step(q0 + kResStepDw, nx); // (the next step's entries arrive while this step's bytes are fetched)
        uint32_t w[kResU];
#pragma unroll
        for (uint32_t u = 0; u < kResU; u++) w[u] = byte_of(v[u].x) | (byte_of(v[u].y) << 8) | (byte_of(v[u].z) << 16) | (byte_of(v[u].w) << 24);
#pragma unroll
        for (uint32_t u = 0; u < kResU; u++) {
            const uint32_t p = (q0 + u * 192) * 4;
            if (p + 4 <= B) __builtin_memcpy(o + p, &w[u], 4);
            else for (uint32_t k = 0; p + k < B; k++) o[p + k] = (uint8_t)(w[u] >> (8 * k));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // this step's bytes have landed
        if (lane == 0) flag_store(&S.res_prog[gw], step + 1);
    }
}
// wavefront 2 beside resolve_gather3: hashes the block's bytes as the steps complete.  fp: the frame's first byte;
// rel0 = out0 - frame start.  false: a wait ran out (the launch is broken).
__device__ __noinline__ bool resolve_hash_behind(uint64_t& xv, uint64_t& xstripes, const uint8_t* fp, uint64_t rel0, uint32_t B, int lane) {
    const uint32_t nsteps = ((B + 3) / 4 + kResStepDw - 1) / kResStepDw;
    for (uint32_t it = 0; it < (1u << 24); it++) {
        const uint32_t p0 = flag_load(&S.res_prog[0]), p1 = flag_load(&S.res_prog[1]), p2 = flag_load(&S.res_prog[2]);
        const uint32_t p = p0 < p1 ? (p0 < p2 ? p0 : p2) : (p1 < p2 ? p1 : p2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const bool fin = p >= nsteps;
        const uint64_t ready = fin ? (uint64_t)B : (uint64_t)p * kResStep;
        uint64_t upto = (rel0 + ready) / 32;
        if (!fin) upto = upto >= xstripes + 64 ? xstripes + ((upto - xstripes) & ~7ull) : xstripes; // >= 2 KiB at a time, whole groups of 8 stripes
        if (upto > xstripes) xxh_advance(xv, xstripes, upto, fp, lane);
        else if (!fin) __builtin_amdgcn_s_sleep(8);
        if (fin) return true;
    }
    return false;
}
