// mzd_k_walk.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
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

__device__ __forceinline__ uint4 ring_fetch_chunk(const SeqStream& st, int32_t chunk, int lane) { // this lane's 16 bytes of the chunk, from HBM
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
    return v;
}
__device__ __forceinline__ void ring_put_chunk(int32_t chunk, const uint4& v, int lane) {
    uint32_t slot = (uint32_t)chunk & (kRingChunks - 1);
    *reinterpret_cast<uint4*>(&S.ring[slot * kChunk + (uint32_t)lane * 16]) = v;
    if (slot == 0 && lane == 0) *reinterpret_cast<uint4*>(&S.ring[kRingBytes]) = v; // mirror
}
__device__ __forceinline__ void ring_load_chunk(const SeqStream& st, int32_t chunk, int lane) { ring_put_chunk(chunk, ring_fetch_chunk(st, chunk, lane), lane); }
// chunks hi, hi - 1, ..., hi - cnt + 1 (cnt <= 6): the loads are in flight together
__device__ __forceinline__ void ring_load_chunks(const SeqStream& st, int32_t hi, int32_t cnt, int lane) {
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) if (k < cnt) v[k] = ring_fetch_chunk(st, hi - k, lane);
#pragma unroll
    for (int k = 0; k < 6; k++) if (k < cnt) ring_put_chunk(hi - k, v[k], lane);
}

// the 8 ring bytes that end at g-offset e (exclusive), as a little-endian u64
__device__ __forceinline__ uint64_t ring_read64(uint32_t e) {
    uint64_t v;
    __builtin_memcpy(&v, &S.ring[(e - 8) & (kRingBytes - 1)], 8);
    return v;
}


constexpr uint32_t kWalkFin = 0x80000000u;
constexpr int kWalkInexact = 1; // walk_sequences_wave: every sequence walked, the bitstream not consumed exactly (Ctl::walk_inexact)
constexpr uint32_t kNoJob = 0xFFFFFFFFu;
constexpr uint32_t kDoneJob = 0xFFFFFFFEu; // the queue is empty
#ifndef MZD_PRE_PRIO
#define MZD_PRE_PRIO 2
#endif
constexpr uint32_t kPreStage = 2304;        // S.ring[2304 .. 3072): between the Huffman segments of the two helper wavefronts
constexpr uint32_t kInRing = 0x80000000u;   // parse_seq_header: the staged header lies in S.ring, not in S.stage

// The hot form of the chain, hand-scheduled: runs of kWalkGroup steps until n steps are done, or a group met a
// sequence wider than its window (slack < 0: the group is void, the caller takes it again carefully from the state saved
// at the group's start), or the read head comes within one group of the lowest resident ring chunk (Gm < thresh: the caller refills).
//
// What a step costs on a lone wavefront is its instruction SLOTS: in-order issue, 6.25 (4-byte encodings) to 7.5 cycles (8-byte)
// each whether the instructions depend on each other or not -- more when the other wavefronts of the SIMD are busy -- plus one LDS
// round trip (~50 cycles).  So the three states live in three LANES of a quad (lane 0 LL, 1 ML, 2 OF; lane 3 walks a dummy entry
// that consumes nothing and leads to itself; the 16 quads of the wavefront do the same work -- a sparse wavefront issues slower):
// ONE table read, ONE field extract and ONE address add per step instead of three, the bit counts of the other two states come
// through DPP quad permutes of the entry's high word (all of them read the word as loaded -- a DPP read of a register written by
// the VALU instruction before it would need wait states).  In front of the table read: 4 DPP adds/moves, the shift amount, the
// 64-bit shift, the field, the address = 8 VALU slots (the one-lane form: 12 + three reads).  Everything the chain does not need --
// the read head, the next window's address and read, the record store, publishing progress -- sits behind the read, in the shadow
// of its latency; the window is waited for separately, right before the shift (it was issued last and is needed later).
// Same arithmetic as the careful C++ step.  The record of a step is the state as it stands: lane k of the quad stores 16 bits, its state's
// address (LL, ML, OF; the fourth lane's are nothing anybody reads).  Progress (records visible to the planner: all but the newest kWalkLag stores have
// landed) is published once per group.  Registers: v[48:49] the lane's entry, v[54:55] the window, v[64:71] temporaries, v84 the
// lane's state address, v87 the read head - 32.  Entries hold ABSOLUTE LDS addresses (pack_entry); the ring's address is a per-lane operand.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// walk record, 8 bytes: LL, ML, OF state addresses (LDS: < 2^16) and 16 spare bits -- the planner, which knows where the walk started
// (Ctl::walk_g0), works a chunk's positions out of the states themselves (plan_wave)
__device__ __forceinline__ uint64_t walk_record(uint32_t vL, uint32_t vM, uint32_t vO, uint32_t gm) { return (uint64_t)(vL | (vM << 16)) | ((uint64_t)(vO | (gm << 16)) << 32); }
constexpr uint32_t kWalkGroup = 8;
#ifndef MZD_WALK_RUN
#define MZD_WALK_RUN 384
#endif
constexpr uint32_t kWalkRun = MZD_WALK_RUN; // steps of a run of the hot loop at most
constexpr uint32_t kWalkLag = 32;
#ifndef MZD_WALK_YIELD
#define MZD_WALK_YIELD 448
#endif
constexpr uint32_t kWalkYield = MZD_WALK_YIELD;
#define MZD_SDWA_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define MZD_DPP_ALL " row_mask:0xf bank_mask:0xf\n"
// the record: lane k of a quad stores 16 bits -- the three state addresses (the fourth lane's are nothing anybody reads: the planner works the
// positions out of the states, plan_wave)
#define MZD_WALK_REC(RECOFF) "global_store_short %[woff], v84, %[base] offset:" RECOFF "\n"
#define MZD_WALK_STEP(SH, RECOFF, TAIL) \
    "s_waitcnt lgkmcnt(1)\n"                                             /* the entry is there (the window may still be on its way) */ \
    "v_add_u32_dpp v64, v49, v49 quad_perm:[1,0,3,2]" MZD_DPP_ALL        /* pair sums of the high words */ \
    "v_mov_b32_dpp v65, v49 quad_perm:[1,2,3,3]" MZD_DPP_ALL             /* the high word one lane up (lane 2: the dummy's, 0) */ \
    "v_add_u32_dpp v65, v49, v65 quad_perm:[2,3,3,3]" MZD_DPP_ALL        /* + two lanes up: bit offset of the own field (low 5 bits): nbO + nbM, nbO, 0 */ \
    "v_add_u32_dpp v64, v64, v64 quad_perm:[2,3,0,1]" MZD_DPP_ALL        /* all four: nbBits sums | total bits << 8 | ...  (v64: written two slots ago) */ \
    "v_sub_u32_sdwa " SH ", %[av], v64" MZD_SDWA_B1                      /* window bits below what this sequence consumes */ \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_lshrrev_b64 v[66:67], " SH ", v[54:55]\n" \
    "v_bfe_u32 v69, v66, v65, v49\n"                                     /* the lane's fresh state bits: width = nbBits, the low bits of the entry */ \
    "v_lshl_add_u32 v84, v69, 3, v48\n" \
    "ds_read_b64 v[48:49], v84\n" \
    "v_sub_u32_sdwa v87, v87, v64" MZD_SDWA_B1                           /* (behind the read from here on) the read head */ \
    "v_bfe_u32 v71, v87, 5, 11\n"                                        /* the window's dword in the ring ... */ \
    "v_lshl_add_u32 v71, v71, 2, %[ringb]\n"                             /* ... of the lane's file */ \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    MZD_WALK_REC(RECOFF)                                                 /* the NEXT step's record: the state as it is now, 16 bits a field */ \
    "v_and_or_b32 %[av], v87, 31, 32\n" \
    TAIL
#define MZD_WALK_SLACK "v_min3_i32 %[slack], %[slack], %[sa], %[sb]\n"
// Everything a lane needs is a per-lane operand, so the quads of the wavefront may walk DIFFERENT files (two groups in a workgroup: walk_run):
// A: the lane's state address (lane & 3: LL, ML, OF, the dummy), absolute in LDS; Gm: its file's read head - 32; woff: byte offset of the lane's
// 16 bits of the next record from `gwalk`; pv: its file's progress value; prog_lds / ringb / thresh: its file's progress word, ring and lower bound.
// n: steps to run (wave-uniform, a multiple of 8); the run ends early for every quad when any quad's group is void or reaches its bound.
__device__ __forceinline__ void walk_run_asm(uint32_t& A, uint32_t& Gm, uint32_t& woff, int32_t& slack, uint32_t& n,
                                             int32_t pv, uint32_t& startA, uint32_t& startG, int32_t thresh, uint32_t prog_lds, uint32_t ringb,
                                             uint64_t lanes, __attribute__((address_space(1))) uint8_t* gwalk) {
    static_assert(kRingBytes == 4 << 11 && offsetof(Shared, ring) == 0, "the window's address is the ring's + 4 * bits [5, 16) of the read head - 32");
    static_assert(kWalkGroup == 8 && kWalkLag == 32 && kWalkLag >= kWalkGroup + 2, "spelled out below");
    uint32_t av, sa, sb;
    uint64_t saved_exec;
    asm volatile(
        "v_mov_b32_e32 v84, %[A]\n v_mov_b32_e32 v87, %[Gm]\n"
        "s_and_saveexec_b64 %[ex], %[lanes]\n" // (the incoming mask is kept and put back: the loop runs on `lanes` of the lanes that were active)
        "v_bfe_u32 v71, v87, 5, 11\n"
        "ds_read_b64 v[48:49], v84\n"
        "v_lshl_add_u32 v71, v71, 2, %[ringb]\n"
        "ds_read2_b32 v[54:55], v71 offset1:1\n"
        MZD_WALK_REC("0") // the first step's record
        "v_and_or_b32 %[av], v87, 31, 32\n"
        "1:\n"
        "v_mov_b32_e32 %[s0], v84\n v_mov_b32_e32 %[s3], v87\n" // the group's starting state (the reads are in flight)
        MZD_WALK_STEP("%[sa]", "8", "")
        MZD_WALK_STEP("%[sb]", "16", MZD_WALK_SLACK "s_waitcnt vmcnt(32)\n v_max_i32_e32 v69, 0, %[pv]\n ds_write_b32 %[prog], v69\n v_add_u32_e32 %[pv], 8, %[pv]\n") // publish
        MZD_WALK_STEP("%[sa]", "24", "")
        MZD_WALK_STEP("%[sb]", "32", MZD_WALK_SLACK)
        MZD_WALK_STEP("%[sa]", "40", "")
        MZD_WALK_STEP("%[sb]", "48", MZD_WALK_SLACK)
        MZD_WALK_STEP("%[sa]", "56", "")
        MZD_WALK_STEP("%[sb]", "64", MZD_WALK_SLACK "v_add_u32_e32 %[woff], 64, %[woff]\n v_sub_u32_e32 v69, v87, %[thresh]\n v_min_i32_e32 v69, v69, %[slack]\n v_cmp_gt_i32_e32 vcc, 0, v69\n")
        "s_sub_u32 %[n], %[n], 8\n"
        "s_cbranch_vccnz 2f\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "2:\n"
        "s_waitcnt lgkmcnt(0)\n" // (the reads issued by the last step: nothing may be in flight into v[48:55] past this block)
        "s_mov_b64 exec, %[ex]\n"
        "v_mov_b32_e32 %[A], v84\n v_mov_b32_e32 %[Gm], v87\n"
        : [A] "+v"(A), [Gm] "+v"(Gm), [woff] "+v"(woff), [slack] "+v"(slack), [av] "=&v"(av), [n] "+s"(n),
          [pv] "+v"(pv), [s0] "=&v"(startA), [s3] "=&v"(startG), [sa] "=&v"(sa), [sb] "=&v"(sb), [ex] "=&s"(saved_exec)
        : [base] "s"(gwalk), [thresh] "v"(thresh), [prog] "v"(prog_lds), [lanes] "s"(lanes), [ringb] "v"(ringb)
        : "v48", "v49", "v54", "v55", "v64", "v65", "v66", "v67", "v69", "v70", "v71", "v84", "v87", "vcc", "scc", "memory");
}

// ---- two groups in a workgroup: one wavefront runs both groups' chains (round 6; the MZD_PAIRS build).
// A walk is 140 K of a 128 KiB JSON block's 305 K VALU instructions, issued for four of 64 lanes.  With two files in a workgroup the two
// walking wavefronts share ONE instruction stream: group 1's POSTS where its chain stands (WalkShare) and sleeps; group 0's takes the
// request and runs both chains in one loop -- lanes 0..3 its own file, lanes 4..7 the partner's: every operand of the loop is a per-lane
// register.  Group 0's wavefront HOLDS the partner's chain from run to run (it writes the chain's state back into the request, a store per
// lane, and takes it from there at its next run) until the partner's file needs something only its own wavefront can do -- a ring
// refill, a void group's careful steps, the end of its walk: then the results go back (state 2) and group 1's wavefront wakes, does that,
// and posts again.  Group 0's own such work stalls the partner's chain for its duration (a refill is six KiB of bitstream now; a void
// group about seven times a block).  Nobody waits long for a partner: group 1's walks alone when group 0's is not in a walk
// (WalkShare::active), group 0's waits a bounded while for a post that is due (the partner is refilling) and else walks alone, a short run.
#ifdef MZD_EXP_PLANDIAG
#define WSTATP(k, v) do { if (lane == 0) atomicAdd(&g_plandiag[k], (uint32_t)(v)); } while (0)
#else
#define WSTATP(k, v)
#endif
struct WalkRun { uint32_t vL, vM, vO, Gm, done; bool voided, broken; uint32_t sL, sM, sO, sG; }; // (all wave-uniform; broken: a bounded wait ran out)
#if MZD_PAIRS
constexpr uint32_t kWalkMeet = 24;   // polls (~150 cycles each) group 0's wavefront waits for a post that is due
constexpr uint32_t kWalkSolo = 64;   // steps of its run alone when the post did not come
// group 0's wavefront, when it leaves its walk: a chain it still holds goes back to its owner
__device__ __forceinline__ void walk_release_partner(int lane) {
    if (grp_index() != 0) return;
    WalkShare& other = S_of(1u).wk;
    if (flag_load_u(&other.state) != 3u) return;
    if (lane == 0) {
        other.rA[0] = other.A[0]; other.rA[1] = other.A[1]; other.rA[2] = other.A[2]; other.rGm = other.Gm; other.voided = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        flag_store(&other.state, 2u);
    }
}
#endif

// One run of the hot loop for the calling wavefront's file, from (vL, vM, vO, Gm) for at most n steps (a multiple of 8): alone, together with
// the partner group's, or by the partner (which may go on to n_all steps: all that is left of the walk).  woff: byte offset of the file's next
// record in ITS record array; pv: its progress value;
// yield: the file's copier is far behind its walk (the walker then gives way to the copiers on its SIMD: walk_sequences_wave).
__device__ __forceinline__ WalkRun walk_run(uint32_t vL, uint32_t vM, uint32_t vO, uint32_t Gm, uint32_t woff, uint32_t n, uint32_t n_all, int32_t thresh, int32_t pv,
                                            uint32_t prog_lds, bool yield, __attribute__((address_space(1))) uint8_t* gwalk, int lane) {
    const uint32_t g = grp_index();
    const uint32_t q = (uint32_t)lane & 3;
    WalkRun r;
    r.broken = false;
    bool joint = false;
#if MZD_PAIRS
    {
        WalkShare& mine = S_of(g).wk;
        WalkShare& other = S_of(g ^ 1u).wk;
        if (g == 1) {
            if (flag_load_u(&other.active)) { // post, sleep
                if (lane == 0) {
                    mine.A[0] = vL; mine.A[1] = vM; mine.A[2] = vO; mine.Gm = Gm; mine.woff = woff; mine.n = n_all; mine.thresh = (uint32_t)thresh; mine.pv = (uint32_t)pv; mine.prog_lds = prog_lds;
                    mine.done = 0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    flag_store(&mine.state, 1u);
                }
                bool alone = false;
                uint32_t st = 0;
#ifdef MZD_EXP_PLANDIAG
                const uint64_t t0_ = __builtin_readcyclecounter();
#endif
                for (uint32_t it = 0; it < (1u << 20); it++) {
                    st = flag_load_u(&mine.state);
                    if (st == 2u) break;
                    if (st == 1u && !flag_load_u(&other.active)) { // the partner has left its walk: take the request back, unless it was taken in this very moment
                        uint32_t got = 1u;
                        if (lane == 0) { uint32_t exp = 1u; got = __atomic_compare_exchange_n(&mine.state, &exp, 0u, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED) ? 1u : 0u; }
                        if (__builtin_amdgcn_readfirstlane((int)got)) { alone = true; break; }
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                if (!alone && st != 2u) { // (a wait that ran out is a failure of the launch, never a walk on garbage)
                    DEVSITE(30);
                    r.broken = true; r.vL = vL; r.vM = vM; r.vO = vO; r.Gm = Gm; r.done = 0; r.voided = false; r.sL = vL; r.sM = vM; r.sO = vO; r.sG = Gm;
                    return r;
                }
#ifdef MZD_EXP_PLANDIAG
                WSTATP(6, (__builtin_readcyclecounter() - t0_) >> 10); WSTATP(alone ? 4 : 11, 1);
#endif
                if (!alone) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    r.vL = flag_load_u(&mine.rA[0]); r.vM = flag_load_u(&mine.rA[1]); r.vO = flag_load_u(&mine.rA[2]); r.Gm = flag_load_u(&mine.rGm);
                    r.done = flag_load_u(&mine.done); r.voided = flag_load_u(&mine.voided) != 0;
                    r.sL = flag_load_u(&mine.sA[0]); r.sM = flag_load_u(&mine.sA[1]); r.sO = flag_load_u(&mine.sA[2]); r.sG = flag_load_u(&mine.sG);
                    if (lane == 0) flag_store(&mine.state, 0u);
                    return r;
                }
            }
        } else {
            uint32_t st = flag_load_u(&other.state);
            if (st == 3u) joint = true; // a chain this wavefront holds
            else if (flag_load_u(&other.active)) { // a post is due: the partner is refilling its ring or redoing a group
#ifdef MZD_EXP_PLANDIAG
                const uint64_t t0_ = __builtin_readcyclecounter();
#endif
                for (uint32_t it = 0; it < kWalkMeet && st != 1u && flag_load_u(&other.active); it++) { __builtin_amdgcn_s_sleep(1); st = flag_load_u(&other.state); }
                if (st == 1u) {
                    uint32_t got = 0;
                    if (lane == 0) { uint32_t exp = 1u; got = __atomic_compare_exchange_n(&other.state, &exp, 3u, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED) ? 1u : 0u; }
                    joint = __builtin_amdgcn_readfirstlane((int)got) != 0;
                }
#ifdef MZD_EXP_PLANDIAG
                WSTATP(5, (__builtin_readcyclecounter() - t0_) >> 10);
#endif
                if (!joint && n > kWalkSolo) n = kWalkSolo; // (it will come: meet soon)
                if (!joint) WSTATP(2, 1);
            }
        }
    }
#endif
    // the lane's file: this wavefront's, or -- a joint run, rows 1 and 3 of the wavefront's four rows of 16 lanes -- the partner's.  (A quad in
    // each row: two quads in ONE row of an otherwise idle wavefront issue 3x slower, tools/micro/exec_micro.hip -- lanes 0..7 were 17-30 % behind.)
    uint32_t A = q == 0 ? vL : (q == 1 ? vM : (q == 2 ? vO : g * (uint32_t)sizeof(Shared) + kLdsWalkDummy));
    uint32_t gm = Gm, wl = woff + 2 * q, ringb = g * (uint32_t)sizeof(Shared), prog = prog_lds;
    int32_t th = thresh, pvl = pv;
    uint32_t n_run = n;
#if MZD_PAIRS
    const bool pl = joint && (((uint32_t)lane >> 4) & 1u) != 0;
    uint32_t on = 0;
    if (joint) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        WalkShare& other = S_of(1u).wk;
        on = flag_load_u(&other.n);
        if (on < n_run) n_run = on;
        // (a joint run gives way only when BOTH files' copiers are far behind their walks; the partner's, as this wavefront sees it now)
        yield = yield && (uint32_t)((int32_t)flag_load_u(&other.pv) + (int32_t)kWalkLag) > kWalkYield + 64 * flag_load_u(&S_of(1u).c.copy_prog);
        if (pl) {
            A = q < 3 ? other.A[q] : (uint32_t)sizeof(Shared) + kLdsWalkDummy;
            gm = other.Gm; th = (int32_t)other.thresh; pvl = (int32_t)other.pv; prog = other.prog_lds; ringb = (uint32_t)sizeof(Shared);
            wl = other.woff + 2 * q + (uint32_t)(kSeqStride * sizeof(uint4)); // (the groups' record arrays are neighbours in one allocation: slot, slot + 1)
        }
    }
#endif
    if (yield) MZD_SETPRIO(MZD_PRIO_WALK_YIELD); else MZD_SETPRIO(MZD_PRIO_WALK);
    // The lanes that run the loop: all 64 (16 quads doing the same work) when the wavefront has its SIMD to itself -- few active lanes
    // issue VALU work 2-4x slower there (tools/micro/exec_micro.hip) --, one quad a file when the launch fills the machine: with four
    // wavefronts on the SIMD that penalty is gone, and the walkers' table reads no longer take a quarter of the CU's LDS bandwidth from
    // the copiers (64 lanes x 8 bytes per read, twice a step, four walkers per CU).  Measured: cfg2 +2.8 %, one file alone -2.4 % -> by launch size.
    const bool full_machine = vgrid() > 512;
    const uint32_t lanes_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(full_machine ? (joint ? 0x000F000Fu : 0xFu) : ~0u));
    const uint32_t lanes_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(full_machine ? 0u : ~0u));
    const uint64_t lanes = (uint64_t)lanes_lo | ((uint64_t)lanes_hi << 32);
    int32_t slack = 64; // minimum over a group of (window bits - bits needed)
    uint32_t startA, startG;
    const uint32_t n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_run);
    uint32_t nn = n0;
#ifdef MZD_EXP_PLANDIAG
    const uint64_t ta_ = __builtin_readcyclecounter();
#endif
    walk_run_asm(A, gm, wl, slack, nn, pvl, startA, startG, th, prog, ringb, lanes, gwalk);
    const uint32_t done = n0 - nn;
#ifdef MZD_EXP_PLANDIAG
    WSTATP(joint ? 14 : 15, (__builtin_readcyclecounter() - ta_) >> 6);
#endif
    if (joint) { WSTATP(1, 1); WSTATP(7, done); } else { WSTATP(3, 1); WSTATP(8, done); }
    const uint64_t voidm = __builtin_amdgcn_ballot_w64(slack < 0);
#if MZD_PAIRS
    if (joint) {
        WalkShare& other = S_of(1u).wk;
        // does the partner's chain need its own wavefront?  a void group, the ring's lower bound, the end of what it asked for
        const bool p_void = (voidm & 0xFFFF0000FFFF0000ull) != 0;
        const bool p_low = __builtin_amdgcn_readlane((int)(gm - (uint32_t)th), 16) < 0;
        const bool p_need = p_void || p_low || done >= on;
        if (lane >= 16 && lane < 19) { other.rA[lane - 16] = A; other.sA[lane - 16] = startA; other.A[lane - 16] = A; }
        if (lane == 16) {
            other.rGm = gm; other.sG = startG; other.voided = p_void ? 1u : 0u;
            other.Gm = gm; other.woff = other.woff + 8 * done; other.n = on - done; other.pv = other.pv + done; other.done = other.done + done;
        }
        if (p_need) {
            WSTATP(9, 1); if (p_void) WSTATP(12, 1); if (p_low) WSTATP(13, 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 16) flag_store(&other.state, 2u);
        }
    }
#endif
    r.vL = __builtin_amdgcn_readlane(A, 0); r.vM = __builtin_amdgcn_readlane(A, 1); r.vO = __builtin_amdgcn_readlane(A, 2);
    r.Gm = (uint32_t)__builtin_amdgcn_readfirstlane((int)gm); // (the loop may have run on quad 0 only)
    r.done = done; r.voided = (voidm & (joint ? 0x0000FFFF0000FFFFull : ~0ull)) != 0;
    r.sL = __builtin_amdgcn_readlane(startA, 0); r.sM = __builtin_amdgcn_readlane(startA, 1); r.sO = __builtin_amdgcn_readlane(startA, 2); r.sG = (uint32_t)__builtin_amdgcn_readfirstlane((int)startG);
    return r;
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
    const uint32_t img = lds_base(); // entries and states are ABSOLUTE LDS addresses (pack_entry)
    uint32_t vL, vO, vM;
    {
        uint32_t e = (G + 7) >> 3;
        uint64_t B = ring_read64(e) << (e * 8 - G);
        uint32_t n = alL + alO + alM;
        if (G - Gzero < n) return MZD_E_CORRUPT;
        vL = alL ? (uint32_t)(B >> (64 - alL)) : 0; B <<= alL;
        vO = alO ? (uint32_t)(B >> (64 - alO)) : 0; B <<= alO;
        vM = alM ? (uint32_t)(B >> (64 - alM)) : 0;
        G -= n;
        vL = vL * 8 + img + kLdsLL; vO = vO * 8 + img + kLdsOF; vM = vM * 8 + img + kLdsML;
    }
    uint32_t i = 0;
    const uint32_t nupd = nseq - 1; // sequences followed by a state update
    const uint32_t prog_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)prog;
    uint32_t Gm = G - 32; // the loop carries the read head minus 32 (saves an add per sequence)
    if (lane == 0) S.c.walk_g0 = Gm; // (the first record's full position: the planner reads it behind the first published progress)
    // One careful step of the chain: the window moves down a dword at a time until the sequence fits (long extra-bit
    // fields: about one sequence in hundreds).  The hot form (walk_run_asm) has no such branch -- a branch on freshly
    // loaded LDS data costs ~35 cycles per sequence on a lone wavefront -- it only notes that a group met such a sequence.
    auto careful_step = [&]() -> bool { // true: the sequence was wider than the first window
        const uint64_t eL = lds_entry(vL), eM = lds_entry(vM), eO = lds_entry(vO);
        // window: the 8 ring bytes at the 4-byte aligned address whose 64 bits end above the read head
        // (two aligned dwords; an unaligned 8-byte LDS read costs ~40 cycles more)
        const uint32_t u = Gm; // read head - 32
        uint32_t ra = (u >> 3) & (kRingBytes - 4);
        uint64_t X;
        __builtin_memcpy(&X, &S.ring[ra], 8);
        *(__attribute__((address_space(1))) uint64_t*)(gwalk + woff) = walk_record(vL, vM, vO, 0);
        woff += 8;
        const uint32_t hL = (uint32_t)(eL >> 32), hM = (uint32_t)(eM >> 32), hO = (uint32_t)(eO >> 32);
        const uint32_t total = ((hL + hM + hO) >> 8) & 0xFF;
        uint32_t av = (u & 31) | 32; // bits of the window below the read head: 32..63
        const bool wide = __builtin_amdgcn_ballot_w64(total > av) != 0;
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
        return wide;
    };
    constexpr int32_t kLook = (int32_t)(kWalkGroup * 12 + 24) * 8; // bits a group can consume (<= 89 a sequence) + the window above the head
#if defined(MZD_STAMPS) && defined(MZD_EXP_WALKSTAT)
    uint64_t ws_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt_ = __builtin_readcyclecounter(), wu_;
#define WSTAT(k, cnt) do { wu_ = __builtin_readcyclecounter(); ws_[k] += wu_ - wt_; wt_ = wu_; ws_[(k) + 1] += (cnt); } while (0)
#else
#define WSTAT(k, cnt)
#endif
    WSTATP(10, 1);
#if MZD_PAIRS
    if (lane == 0) flag_store(&S.wk.active, 1u); // (two groups in a workgroup: this wavefront will come to the meeting point -- walk_run)
#endif
    int rc_walk = 0;
    while (i < nupd) {
        WSTAT(6, 0);
        // Where the copier cannot keep up (files of many short matches: its LDS rounds under a full machine), a walker further ahead only
        // takes issue slots from the copiers it shares its SIMD with: it yields while its own file's copier is more than kWalkYield
        // sequences behind (checked each time the assembly run returns: per KiB of bitstream).  Measured (thresholds 320 .. 2048, side by side): 400 .. 512 is best -- cfg2 +3.5 %, cfg2x8 +4 %, cfg3 +1 %; files
        // whose copier keeps up (cfg3's sequence-heavy classes) keep the walker in front.
        const bool yield = i > kWalkYield + 64 * flag_load_u(&S.c.copy_prog); // (takes effect in walk_run: a joint run gives way only when both files would)
        // keep the ring ahead of the read head.  Round 6: every refill brings all the chunks the ring has room for (the chunks above the read
        // head are dead), not one: a run of the hot loop ends when the head nears the lowest resident chunk, and with two files a run ends for
        // both when either needs a refill -- 21 refills a 128 KiB JSON block were 21 HBM round trips in front of both chains.
        if (st.lowest > 0 && (int32_t)Gm < st.lowest * (int32_t)(kChunk * 8) + kLook) {
            const int32_t head_chunk = (int32_t)((Gm + 32 + 64) / (kChunk * 8)); // the highest chunk a window may still touch
            int32_t want = head_chunk + 1 - (int32_t)kRingChunks + 1;          // the lowest chunk that fits the ring beside it
            if (want < 0) want = 0;
            if (want > st.lowest - 1) want = st.lowest - 1;
            while (st.lowest > want) { // (six at a time, their loads in flight together)
                const int32_t cnt = st.lowest - want < 6 ? st.lowest - want : 6;
                ring_load_chunks(st, st.lowest - 1, cnt, lane);
                st.lowest -= cnt;
                WSTAT(0, cnt);
            }
        }
        const uint32_t left = nupd - i;
        if (left >= kWalkGroup) {
            // (a run is at most kWalkRun steps: the yield above is looked at per run -- when a run ended per KiB of bitstream, at every ring refill, that
            //  was every ~400 sequences -- and a partner group that starts walking is met at the run's end)
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((left < kWalkRun ? left : kWalkRun) & ~(kWalkGroup - 1));
            // (the whole stream resident: the run still ends at the first group that read past the stream's start -- a corrupt
            //  stream: published records must never carry a position outside the stream, the planner addresses HBM with them;
            //  records younger than kWalkLag are not published, so stopping at the group's end is early enough)
            const int32_t thresh = __builtin_amdgcn_readfirstlane(st.lowest > 0 ? st.lowest * (int32_t)(kChunk * 8) + kLook : (int32_t)Gzero - 32);
            const WalkRun r = walk_run(vL, vM, vO, Gm, woff, n, (uint32_t)__builtin_amdgcn_readfirstlane(left & ~(kWalkGroup - 1)), thresh, (int32_t)i - (int32_t)kWalkLag, prog_lds, yield, gwalk, lane);
            if (r.broken) { rc_walk = MZD_E_DEVICE; break; }
            i += r.done;
            woff += 8 * r.done;
            vL = r.vL; vM = r.vM; vO = r.vO; Gm = r.Gm;
            WSTAT(2, 1);
            if (r.voided) { // the last group is void: once more from its start, carefully
                i -= kWalkGroup; woff -= 8 * kWalkGroup;
                vL = r.sL; vM = r.sM; vO = r.sO; Gm = r.sG;
                // ... up to and including the first sequence that needed the wider window (what follows goes back to the hot form)
                for (uint32_t k = 0; k < kWalkGroup; k++) { i++; if (careful_step()) break; }
                WSTAT(4, 1);
            }
        } else {
            for (; i < nupd; i++) careful_step();
        }
        if ((int32_t)(Gm + 32 - Gzero) < 0) { rc_walk = MZD_E_CORRUPT; break; } // over-read
    }
#if MZD_PAIRS
    walk_release_partner(lane);
    if (lane == 0) flag_store(&S.wk.active, 0u);
#endif
    if (rc_walk) return rc_walk;
#if defined(MZD_STAMPS) && defined(MZD_EXP_WALKSTAT)
    if (lane == 0) for (int k_ = 0; k_ < 8; k_++) S.cdiag[k_] = ws_[k_];
#endif
    MZD_SETPRIO(MZD_PRIO_WALK);
    G = Gm + 32;
    // last sequence: extra bits only
    {
        const uint64_t eL = lds_entry(vL), eM = lds_entry(vM), eO = lds_entry(vO);
        *(__attribute__((address_space(1))) uint64_t*)(gwalk + woff) = walk_record(vL, vM, vO, 0);
        uint32_t extra = (uint32_t)(eL >> 56) + (uint32_t)(eM >> 56) + (uint32_t)(eO >> 56);
        if (G - Gzero < extra) return MZD_E_CORRUPT;  // the last sequence's fields reach below the stream's start: over-read, as above
        if (G - Gzero != extra) return kWalkInexact; // the bitstream must be consumed exactly: found last (Ctl::walk_inexact)
    }
    return 0; // the caller publishes nseq | kWalkFin after a release fence
}

// n bits (n <= 32) whose top is g-bit `top` (exclusive), read from HBM
__device__ __forceinline__ uint32_t stream_bits(const uint8_t* gbase, uint32_t top, uint32_t n) {
    uint32_t lo = top - n;
    uint64_t v = ldu64(gbase + (lo >> 3)) >> (lo & 7);
    return n ? (uint32_t)v & (uint32_t)((1ull << n) - 1) : 0u;
}

