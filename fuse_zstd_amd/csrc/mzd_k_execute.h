// mzd_k_execute.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
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
// (kStage, the bytes of a staged run: mzd_k_common.h)
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

// The same chain as a scan over REFERENCES (plan_wave's common case): a transform is three bytes -- new slot j holds old slot b (b = 0..2)
// or the offset value lane k brought (0x40 | k); byte 3 is 3 so that the identity is v_perm's identity selector.
constexpr uint32_t kRefId = 0x03020100u;
__device__ __forceinline__ uint32_t ref_compose(uint32_t later, uint32_t earlier) {
    const uint32_t p = __builtin_amdgcn_perm(0u, earlier, later); // later's bytes select among earlier's ...
    const uint32_t mask = ((later & 0x00404040u) >> 6) * 0xFFu;   // ... except where they are lanes
    return (later & mask) | (p & ~mask);
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

// MZD_W3: the planning wavefront is also the hashing one -- in its waits for the walker it takes the checksum (and the host mirror) as far as
// the copier's published position allows (mzd_k_pipeline.h: follow_step)
struct FollowHook { uint64_t* xv; uint64_t* xstripes; uint64_t* mirrored; const uint8_t* frame; uint64_t fstart; uint8_t* dst; uint8_t* dst2; bool hashing; };
__device__ __noinline__ bool follow_step(const FollowHook& h, int lane);
struct PlanCtx { // what the planning wavefront needs
    const uint4* walk;       // state-walk records of the block (HBM scratch)
    const uint8_t* seq_sp;   // the block's sequence bitstream
    const uint32_t* prog;    // walker progress (LDS)
    uint32_t nlit;
    uint32_t rep_known;      // the repeat offsets at the start of the block are known (first block of a frame)
    uint32_t rep[3];
    uint4* chunk_base;       // [k] = {output, literals} of the block before chunk k (for mzd_k_resolve.h): the walk records' array, whose
                             // entry k the planner has consumed by the time it plans chunk k
    uint32_t seq_len;        // bytes of the sequence bitstream: no field is fetched from outside it
    struct FollowHook* hook; // MZD_W3: what this wavefront does while it waits for the walker (the checksum, the host mirror), or null
};

// Offsets in the plan: a plain value, or -- when the block starts before its predecessor has finished, so that
// the repeat offsets at its start are still unknown -- a reference to one of the three start slots plus a delta.
// The copier resolves those (it runs after the predecessor).  0 is never a valid offset.
constexpr uint32_t kOffTag = 0x80000000u;
constexpr int32_t kOffBias = 1 << 28;
__device__ __forceinline__ uint32_t off_symbolic(uint32_t slot, int32_t delta) { return kOffTag | (slot << 29) | ((uint32_t)(delta + kOffBias) & 0x1FFFFFFFu); }

// The plan in HBM, 8 bytes a sequence (round 6; 16 before: the plan was 150 KB of a 128 KiB JSON block's traffic, written and read):
// {ll | ml << 16, offset}.  A length of 0xFFFF or more is spelled 0xFFFF and stands in full at the same index of the array's second half
// ({ll, ml}; one block in thousands holds such a sequence).  A sequence's output offset inside its chunk of 64 is not stored: its readers scan.
typedef uint2 PlanEnt;
__device__ __forceinline__ PlanEnt* plan_of(uint4* seqs) { return reinterpret_cast<PlanEnt*>(seqs); }             // (the workgroup's kSeqStride x 16 bytes)
__device__ __forceinline__ const PlanEnt* plan_of(const uint4* seqs) { return reinterpret_cast<const PlanEnt*>(seqs); }
__device__ __forceinline__ void plan_store(PlanEnt* plan, uint32_t i, uint32_t ll, uint32_t ml, uint32_t off) {
    typedef __attribute__((address_space(1))) uint8_t* gptr_;
    const PlanEnt e = make_uint2((ll < 0xFFFFu ? ll : 0xFFFFu) | (ml < 0xFFFFu ? ml : 0xFFFFu) << 16, off);
    __builtin_memcpy((gptr_)(uintptr_t)(plan + i), &e, 8);
    if (ll >= 0xFFFFu || ml >= 0xFFFFu) { const uint2 w = make_uint2(ll, ml); __builtin_memcpy((gptr_)(uintptr_t)(plan + kSeqStride + i), &w, 8); }
}
// e: entry i as loaded (zero for a lane without a sequence)
__device__ __forceinline__ void plan_expand(const PlanEnt* plan, uint32_t i, PlanEnt e, uint32_t& ll, uint32_t& ml, uint32_t& off) {
    typedef const __attribute__((address_space(1))) uint8_t* gcptr_;
    ll = e.x & 0xFFFFu; ml = e.x >> 16; off = e.y;
    const bool big = ll == 0xFFFFu || ml == 0xFFFFu;
    if (__any(big)) { // (wave-uniform, rare)
        if (big) { uint2 w; __builtin_memcpy(&w, (gcptr_)(uintptr_t)(plan + kSeqStride + i), 8); ll = w.x; ml = w.y; }
    }
}

// K4(b) + the bookkeeping half of K5, by one wavefront, 64 sequences per step (lane = sequence):
// field conversion from the walk records, repeat offsets, positions, what can be validated without knowing
// where the block's output starts (the copier checks capacity and offsets).  The result goes to the plan array in
// HBM (plan_store).  The block's total repeat-offset transform
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
    // (the context lives in the caller's frame: what the loop uses is read once, into scalar registers; HBM is addressed through GLOBAL
    //  pointers -- a flat access also counts on the LDS counter, so every wait for a table entry would wait for the records and bit
    //  windows that are in flight for the chunks ahead)
    auto u32_ = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto u64_ = [&](uint64_t v) -> uint64_t { return (uint64_t)u32_((uint32_t)v) | ((uint64_t)u32_((uint32_t)(v >> 32)) << 32); };
    typedef __attribute__((address_space(1))) uint8_t* gptr;
    const uint8_t* const seq_sp = (const uint8_t*)(uintptr_t)u64_((uint64_t)(uintptr_t)cx.seq_sp);
    const uint32_t cx_nlit = u32_(cx.nlit);
    uint4* const chunk_base = (uint4*)(uintptr_t)u64_((uint64_t)(uintptr_t)cx.chunk_base);
    const uint32_t bias = 16 + (uint32_t)((uintptr_t)seq_sp & 15);
    const uint8_t* const gbase = seq_sp - bias;
    const uint32_t g_hi = (bias + u32_(cx.seq_len)) * 8; // positions (bits from gbase) lie in [bias * 8, g_hi]: whatever the records say, nothing outside is fetched
    auto wait_walker = [&](uint32_t need) -> bool { // true when sequences [0, need) are recorded
        if (need > nseq) need = nseq;
        uint32_t pg = 0, it = 0;
        for (; it < (1u << 24); it++) {
            pg = flag_load_u(&S.c.walk_prog); // (= cx.prog, spelled as the LDS word it is: through the generic pointer it was a flat load)
            if ((pg & ~kWalkFin) >= need || (pg & kWalkFin)) break;
            if (!(cx.hook && follow_step(*cx.hook, lane))) __builtin_amdgcn_s_sleep(4);
        }
        if (it == (1u << 24)) { DEVSITE(2); post_err(&S.c.err, MZD_E_DEVICE); } // (a wait that ran out: see spin_ge)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return (pg & ~kWalkFin) >= need;
    };
    struct Win { uint32_t hL, hM, hO, G; uint64_t bO, bM, bL; }; // entry words + raw 8-byte windows of one sequence
    // records are 8 bytes (walk_record): three 16-bit state addresses (absolute in LDS) and 16 bits nobody reads
    const uint2* const recs = reinterpret_cast<const uint2*>((uintptr_t)u64_((uint64_t)(uintptr_t)cx.walk));
    auto load_rec = [&](uint32_t idx) -> uint2 { uint2 r = make_uint2(0, 0); if (idx < nseq) __builtin_memcpy(&r, (gcptr)(recs + idx), 8); return r; };
    // The records carry the three states only.  Where a sequence's fields lie follows from the states themselves: a sequence consumes
    // extra + nbBits of each of its three entries (byte 1 of their high words; < 128 bits together), so a chunk's read heads are its
    // first one minus an exclusive scan of those sums -- a DPP scan a chunk here instead of a select and a fourth lane's store per
    // sequence on the walking wavefront, whose instruction slots are the block's critical path.  Returns the chunk's total.
    auto issue_bits = [&](uint2 w2, uint32_t first_full, bool live, Win& o) -> uint32_t {
        const uint32_t vL = w2.x & 0xFFFFu, vM = w2.x >> 16, vO = w2.y & 0xFFFFu;
        o.hL = (uint32_t)(lds_entry(vL) >> 32); o.hM = (uint32_t)(lds_entry(vM) >> 32); o.hO = (uint32_t)(lds_entry(vO) >> 32); // (records hold state addresses)
        const uint32_t tot_ = live ? ((o.hL + o.hM + o.hO) >> 8) & 0xFFu : 0u;
        const uint32_t incl_ = wave_incl_scan(tot_, lane);
        o.G = first_full - (incl_ - tot_) + 32; // (first_full: the chunk's first read head - 32)
        const uint32_t chunk_bits = __builtin_amdgcn_readlane(incl_, 63);
        o.bO = 0; o.bM = 0; o.bL = 0;
#ifdef MZD_EXP_PLANDIAG
        if (live && !(o.G <= g_hi && o.G >= bias * 8) && atomicCAS(&g_plandiag[0], 0u, 1u) == 0u) {
            g_plandiag[1] = S.c.job; g_plandiag[2] = (uint32_t)lane; g_plandiag[3] = o.G; g_plandiag[4] = g_hi; g_plandiag[5] = bias * 8; g_plandiag[6] = first_full;
            g_plandiag[7] = nseq; g_plandiag[8] = w2.x; g_plandiag[9] = w2.y; g_plandiag[10] = grp_index(); g_plandiag[11] = vgrid(); g_plandiag[12] = flag_load(&S.c.walk_prog); g_plandiag[13] = flag_load(&S.c.walk_g0);
            g_plandiag[14] = o.hL; g_plandiag[15] = o.hO;
        }
#endif
        // (the walker stops at the first group that over-reads a corrupt stream and never publishes it; this bound is the second line: a position
        //  that left the stream would be a wild HBM read here -- the fields then read as zero and the block fails in the copier's or the walker's verdict)
        if (live && o.G <= g_hi && o.G >= bias * 8) { // (a field reaches at most 48 bits below its top: the 8 bytes in front of the stream are the block's own)
            const uint32_t xM = o.hM >> 24, xO = o.hO >> 24, xL = o.hL >> 24;
            const uint32_t tO = o.G - xO, tM = tO - xM, tL = tM - xL; // bottoms of the three fields
            const gcptr gb = (gcptr)gbase;
            __builtin_memcpy(&o.bO, gb + (tO >> 3), 8); __builtin_memcpy(&o.bM, gb + (tM >> 3), 8); __builtin_memcpy(&o.bL, gb + (tL >> 3), 8);
        }
        return chunk_bits;
    };
    if (!wait_walker(128)) return MZD_E_CORRUPT;
    uint2 recA = load_rec((uint32_t)lane), recB = load_rec(64 + (uint32_t)lane); // chunks 0 and 1
    uint32_t gfirst = flag_load_u(&S.c.walk_g0); // full position of the current chunk's first record ...
    Win win;
    uint32_t bits_cur = issue_bits(recA, gfirst, (uint32_t)lane < nseq, win); // (bits the current chunk consumes)
    uint32_t chunk = 0;
    for (uint32_t base = 0; base < nseq; base += 64, chunk++) {
        const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
        const uint32_t i = base + (uint32_t)lane;
        const bool valid = (uint32_t)lane < cnt;
        if (lane == 0) { const uint4 cb_ = make_uint4(opos, lpos, 0, 0); __builtin_memcpy((gptr)(uintptr_t)(chunk_base + chunk), &cb_, 16); }
        // everything this wavefront stored an iteration ago has landed: chunk k-1 of the plan is public
        wg_fence();
        if (lane == 0) flag_store(&S.c.plan_prog, chunk);
        // stage 1: records of chunk k+2, bit windows of chunk k+1 (recB arrived an iteration ago)
        if (!wait_walker(base + 192)) return MZD_E_CORRUPT; // the walker failed (it posted the error) or never got there
        const uint2 recC = load_rec(base + 128 + (uint32_t)lane);
        Win next;
        const uint32_t gnext = gfirst - bits_cur;
        bits_cur = issue_bits(recB, gnext, base + 64 + (uint32_t)lane < nseq, next);
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
        win = next; recA = recB; recB = recC; gfirst = gnext;
        // ---- repeat offsets
        uint32_t off;
        const uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u); // (meaningful when ofv <= 3)
        if (!__any(valid && ofv <= 3 && idx == 3)) {
            // The common chunk (no "repeat 0 minus one", the only rule that makes a new VALUE out of the state): the scan runs over
            // REFERENCES, not values.  The state before a sequence is three bytes, each the chunk's start slot 0..2 or "the offset lane k
            // brought" (0x40 | k); a sequence permutes / pushes that triple, and composing two of them is one v_perm_b32 + a byte merge
            // (ref_compose) instead of the ~45 instructions of the value-carrying rep_compose (this scan was half of the planner's VALU work).
            const uint32_t pushv = ofv - 3;
            uint32_t T = kRefId;
            if (valid) {
                if (ofv > 3) T = 0x03010040u | (uint32_t)lane;
                else if (idx == 1) T = 0x03020001u;
                else if (idx == 2) T = 0x03010002u;
            }
            const uint32_t P = wave_incl_scan_op(T, kRefId, [](uint32_t earlier, uint32_t later) { return ref_compose(later, earlier); });
            uint32_t Ex = __shfl_up(P, 1);
            if (lane == 0) Ex = kRefId;
            // the chunk's start slots as the plan spells them: a plain value, or "block start slot + delta" (uniform)
            const uint32_t r_s = __builtin_amdgcn_readfirstlane(R.s);
            const int32_t r_v0 = __builtin_amdgcn_readfirstlane(R.v0), r_v1 = __builtin_amdgcn_readfirstlane(R.v1), r_v2 = __builtin_amdgcn_readfirstlane(R.v2);
            auto start_word = [&](uint32_t j, int32_t v) -> uint32_t {
                const uint32_t src = (r_s >> (2 * j)) & 3;
                return src == 3 ? (v > 0 ? (uint32_t)v : 0u) : off_symbolic(src, v);
            };
            const uint32_t w0 = start_word(0, r_v0), w1 = start_word(1, r_v1), w2 = start_word(2, r_v2);
            const uint32_t slot = idx == 1 ? 1u : (idx == 2 ? 2u : 0u);
            const uint32_t ref = (Ex >> (8 * slot)) & 0xFF;
            const uint32_t from_lane = (uint32_t)__shfl((int)pushv, (int)(ref & 63));
            const uint32_t from_start = (uint32_t)sel3(ref & 3, (int32_t)w0, (int32_t)w1, (int32_t)w2);
            off = ofv > 3 ? pushv : ((ref & 0x40) ? from_lane : from_start);
            // chunk end -> R of the next chunk (uniform)
            const uint32_t PL = __builtin_amdgcn_readlane(P, 63);
            uint32_t ns = 0;
            int32_t nv[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const uint32_t b = (PL >> (8 * j)) & 0xFF;
                if (b & 0x40) { ns |= 3u << (2 * j); nv[j] = (int32_t)__builtin_amdgcn_readlane(pushv, b & 63); }
                else { ns |= ((r_s >> (2 * (b & 3))) & 3) << (2 * j); nv[j] = sel3(b & 3, r_v0, r_v1, r_v2); }
            }
            R.s = ns; R.v0 = nv[0]; R.v1 = nv[1]; R.v2 = nv[2];
        } else {
            RepOp op; // (a chunk that holds a "repeat 0 minus one": the value-carrying scan)
            if (!valid || (ofv <= 3 && idx == 0)) { op.s = 0 | (1 << 2) | (2 << 4); op.v0 = 0; op.v1 = 0; op.v2 = 0; }
            else if (ofv > 3) { op.s = 3 | (0 << 2) | (1 << 4); op.v0 = (int32_t)(ofv - 3); op.v1 = 0; op.v2 = 0; }
            else if (idx == 1) { op.s = 1 | (0 << 2) | (2 << 4); op.v0 = 0; op.v1 = 0; op.v2 = 0; }
            else if (idx == 2) { op.s = 2 | (0 << 2) | (1 << 4); op.v0 = 0; op.v1 = 0; op.v2 = 0; }
            else { op.s = 0 | (0 << 2) | (1 << 4); op.v0 = -1; op.v1 = 0; op.v2 = 0; }
            RepOp acc = op; // inclusive scan: acc = op_lane o ... o op_0 (the DPP steps of wave_incl_scan_op, four words at a time)
            {
                constexpr uint32_t kIdS = 0u | (1u << 2) | (2u << 4); // the identity: every slot is itself
                auto step = [&](auto ctrl_c, auto rows_c) {
                    constexpr int CTRL = decltype(ctrl_c)::value, ROWS = decltype(rows_c)::value;
                    RepOp e;
                    e.s = dpp_take<CTRL, ROWS>(kIdS, acc.s);
                    e.v0 = (int32_t)dpp_take<CTRL, ROWS>(0u, (uint32_t)acc.v0); e.v1 = (int32_t)dpp_take<CTRL, ROWS>(0u, (uint32_t)acc.v1); e.v2 = (int32_t)dpp_take<CTRL, ROWS>(0u, (uint32_t)acc.v2);
                    acc = rep_compose(acc, e);
                };
                step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xF>{});
                step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xF>{});
                step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xF>{});
                step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xF>{});
                step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});
                step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});
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
        // the plan of this sequence -> HBM (unbounded, so the planner never waits for the copier, which may still be decoding
        // literals); also what mzd_debug_last_block shows
        if (valid) plan_store(plan_of(seqs), i, ll, ml, off);
        if (chunk_lit > cx_nlit - lpos || opos + chunk_tot > kBlockMax) {
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
    const uint32_t rest = cx_nlit - lpos;
    wg_fence();
    if (lane == 0) {
        S.c.rep_op[0] = R.s; S.c.rep_op[1] = (uint32_t)R.v0; S.c.rep_op[2] = (uint32_t)R.v1; S.c.rep_op[3] = (uint32_t)R.v2;
        S.c.plan_lit_used = lpos; S.c.plan_out = opos;
        if (opos + rest > kBlockMax) S.c.plan_too_long = 2; // only the literals after the last sequence pass the limit: every chunk is published
        flag_store(&S.c.plan_prog, chunk);
    }
    return opos + rest > kBlockMax ? kPlanBlockTooLong : 0;
}

struct CopyCtx {
    const uint4* plan;       // the block's plan (HBM; plan_of, plan_expand)
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
    if (it == (1u << 24)) { DEVSITE(3); post_err(&S.c.err, MZD_E_DEVICE); }
    return rc;
}
// ... and inside the chunk that cannot be executed (sequences base .. base+63 of the plan, read again here) the earliest
// offending sequence decides.  room / blk_room: bytes left in the destination / under the block limit at the chunk's
// start; hist: output of the frame + dictionary bytes before the chunk; rep: the block's starting repeat offsets.
__device__ __noinline__ int chunk_verdict(const PlanEnt* plan, uint32_t base, int lane, uint64_t room, uint32_t blk_room, uint64_t hist, uint32_t lit_room,
                                          uint32_t rep0, uint32_t rep1, uint32_t rep2, uint32_t nseq, uint32_t lit_streams) {
    const bool valid = base + (uint32_t)lane < nseq;
    const PlanEnt pe = valid ? plan[base + (uint32_t)lane] : make_uint2(0, 0);
    uint32_t ll, ml, off;
    plan_expand(plan, base + (uint32_t)lane, pe, ll, ml, off);
    if (off & kOffTag) off = (uint32_t)sel3((off >> 29) & 3, (int32_t)rep0, (int32_t)rep1, (int32_t)rep2) + (off & 0x1FFFFFFFu) - (uint32_t)kOffBias; // (as in copy_wave)
    const uint32_t incl_t = wave_incl_scan(ll + ml, lane), ex_t = incl_t - ll - ml;
    // per sequence the reference checks (ZSTD_execSequenceEnd): the destination's end, literals left (lit_room: literals not yet used at
    // the chunk's start), then -- ours -- the block limit, then the offset: "destination too small" when the earliest offending sequence
    // has that wrong, whatever else is wrong with it
    const uint64_t nolit = __ballot(valid && wave_incl_scan(ll, lane) > lit_room);
    const uint64_t over = __ballot(valid && incl_t > room);
    const uint64_t bad = __ballot(valid && (incl_t > blk_room || off == 0 || off > hist + ex_t + ll));
    const int fl = nolit ? __builtin_ctzll(nolit) : 64, fo = over ? __builtin_ctzll(over) : 64, fb = bad ? __builtin_ctzll(bad) : 64;
    // (none of the three: the plan ended here without a sequence of this chunk being at fault, which cannot happen; corrupt)
    return exec_verdict(fo <= fl && fo <= fb ? MZD_E_DSTSIZE : MZD_E_CORRUPT, nseq, lit_streams);
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
    const PlanEnt* const plan = plan_of((const uint4*)(uintptr_t)u64((uint64_t)(uintptr_t)cx.plan));
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
            pg = flag_load_u(&S.c.plan_prog);
            if ((pg & ~kPlanFin) >= nchunks_needed || (pg & kPlanFin)) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (it == (1u << 24)) { DEVSITE(4); post_err(&S.c.err, MZD_E_DEVICE); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return (pg & ~kPlanFin) >= nchunks_needed;
    };
    // literals become available stream by stream (in order: stream k fills [s_out[k], s_out[k] + s_n[k]))
    uint32_t lit_avail = lit_streams ? 0u : nlit_all;
    auto wait_lits = [&](uint32_t need) -> bool {
        if (need <= lit_avail) return true;
        if (need > nlit_all) return false; // more literals than the block has (the caller tells the two failures apart)
        for (uint32_t it = 0; it < (1u << 24); it++) {
            const uint32_t m = flag_load_u(&S.c.streams_mask);
            const uint32_t k = (uint32_t)__builtin_ctz(~m); // first stream not decoded yet
            lit_avail = k >= lit_streams ? nlit_all : u32(S.c.s_out[k]);
            if (need <= lit_avail) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return true; }
            if (flag_load_u((const uint32_t*)&S.c.err)) return false;
            __builtin_amdgcn_s_sleep(4);
        }
        DEVSITE(5);
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
#if defined(MZD_STAMPS) && defined(MZD_EXP_ROUNDS)
            if (lane == 0) S.c.diag_slow += 1; // (diagnostic: LDS -> LDS rounds of the block)
#endif
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
        __attribute__((address_space(1))) uint8_t* const g = (__attribute__((address_space(1))) uint8_t*)(dst + ri.run_pos); // (global stores)
        for (uint32_t k = (uint32_t)lane * 16; k + 16 <= ri.T; k += 1024) {
            uint4 v = *reinterpret_cast<const uint4*>(sb + k);
            __builtin_memcpy(g + k, &v, 16);
        }
        const uint32_t tail0 = ri.T & ~15u; // the last partial 16 bytes: one byte per lane
        if (tail0 + (uint32_t)lane < ri.T) g[tail0 + lane] = sb[tail0 + lane];
        CSTAMP(7);
    };

    PlanEnt pe_next = make_uint2(0, 0);
    if (nseq) {
        if (!wait_plan(1)) return MZD_E_CORRUPT; // the planner failed and posted the error
        if ((uint32_t)lane < nseq) __builtin_memcpy(&pe_next, (gcptr)(uintptr_t)(plan + lane), 8); // (global, not flat: see plan_wave)
    }
    uint32_t chunk = 0;
    uint32_t blk_room = kBlockMax; // bytes left under the block limit
    uint32_t tbase = 0xFFFFFFFFu; // the chunk that cannot be executed (see chunk_verdict)
    for (uint32_t base = 0; base < nseq; base += 64, chunk++) {
        const uint32_t cnt = nseq - base < 64 ? nseq - base : 64;
        const PlanEnt pe = pe_next; // loaded an iteration ago
        if (lane == 0) flag_store(&S.c.copy_prog, chunk); // (how far this wavefront is: the walker yields when it is far ahead, mzd_k_walk.h)
        CSTAMP(1);
        bool cut = false; // the plan ends with this chunk (the block's output passes 128 KiB in it)
        if (chunk + 1 < nchunks) { // prefetch the next chunk's plan
            if (wait_plan(chunk + 2)) {
                CSTAMP(0);
                const uint32_t j = base + 64 + (uint32_t)lane;
                pe_next = make_uint2(0, 0);
                if (j < nseq) __builtin_memcpy(&pe_next, (gcptr)(uintptr_t)(plan + j), 8);
            } else if (flag_load_u(&S.c.plan_too_long) == 1) cut = true;
            else return MZD_E_CORRUPT;
        }
        const bool valid = (uint32_t)lane < cnt;
        uint32_t ll, ml, off; // (a lane without a sequence: the entry is zero)
        plan_expand(plan, base + (uint32_t)lane, pe, ll, ml, off);
        if (off & kOffTag) { // an offset left symbolic by the planner: start slot + delta
            const uint32_t slot = (off >> 29) & 3;
            off = (uint32_t)sel3(slot, (int32_t)cx.rep[0], (int32_t)cx.rep[1], (int32_t)cx.rep[2]) + (off & 0x1FFFFFFFu) - (uint32_t)kOffBias;
            if (cx.plan_wb && valid) plan_of(cx.plan_wb)[base + (uint32_t)lane].y = off;
        }
        // the sequence's output offset inside the chunk, and its literals' place among the chunk's: ONE scan over both sums where every
        // length of the chunk is short (64 x 127 < 2^16: the lower sum never carries into the upper), else two
        uint32_t incl_t, incl_l;
        if (!__any((ll | ml) > 127u)) { const uint32_t P = wave_incl_scan(ll | (ll + ml) << 16, lane); incl_l = P & 0xFFFFu; incl_t = P >> 16; }
        else { incl_t = wave_incl_scan(ll + ml, lane); incl_l = wave_incl_scan(ll, lane); }
        const uint32_t ex_t = incl_t - ll - ml;
        const uint32_t chunk_tot = __builtin_amdgcn_readlane(incl_t, cnt - 1);
        if (chunk_tot > cap - opos || chunk_tot > blk_room || cut ||
            __any(valid && (off == 0 || off > (opos + ex_t + ll - frame_start) + dict_len))) { // (beyond the window's history)
            tbase = base;
            break;
        }
        blk_room -= chunk_tot;
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
            if (flag_load_u(&S.c.plan_prog) & kPlanFin) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (it == (1u << 24)) { DEVSITE(6); post_err(&S.c.err, MZD_E_DEVICE); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (flag_load_u((const uint32_t*)&S.c.err)) return MZD_E_CORRUPT;
        if (flag_load_u(&S.c.walk_inexact)) return MZD_E_CORRUPT; // every sequence executed, the bitstream not consumed exactly (the planner has finished, so the walker has)
        if (flag_load_u(&S.c.plan_too_long) == 2) // only the literals after the last sequence pass the block limit: the destination's end comes first
            return exec_verdict(cap - *opos_io <= kBlockMax ? MZD_E_DSTSIZE : MZD_E_CORRUPT, nseq, lit_streams);
        if (lpos != flag_load_u(&S.c.plan_lit_used)) return MZD_E_CORRUPT;
        if (nlit_all - lpos > cap - opos) return MZD_E_DSTSIZE;
    } else {
        if (nlit_all > kBlockMax) return MZD_E_CORRUPT;
        if (nlit_all > cap - opos) return MZD_E_DSTSIZE; // a block without sequences has no planner to check this
    }
    const uint32_t rest = nlit_all - lpos;
    if (lit + lpos == dst + opos && lit_streams > 1) {
        // a literal-only block decoded in place: nothing to move, but the hashing wavefront follows exec_pos -- it is told of every
        // stream's share of the output as the streams before it complete (the hash of 128 KiB takes as long as a stream's decode)
        for (uint32_t k = 1; k < lit_streams; k++) {
            const uint32_t upto = u32(S.c.s_out[k]);
            if (!wait_lits(upto)) return MZD_E_CORRUPT;
            wg_fence();
            if (lane == 0) __atomic_store_n(&S.c.exec_pos, opos + upto, __ATOMIC_RELAXED);
        }
    }
    if (!wait_lits(nlit_all)) return MZD_E_CORRUPT;
    if (lit + lpos != dst + opos) wave_copy(dst + opos, lit + lpos, rest, lane); // (literal-only block decoded in place: nothing to move)
    opos += rest;
    wg_fence();
    if (lane == 0) __atomic_store_n(&S.c.exec_pos, opos, __ATOMIC_RELAXED);
    *opos_io = opos;
    return 0;
}

