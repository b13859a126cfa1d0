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
// Same arithmetic as the careful C++ step.  The record of a step is the state as it stands: lane k of quad 0 stores dword k
// {LL, ML, OF state address, read head - 32}.  Progress (records visible to the planner: all but the newest kWalkLag stores have
// landed) is published once per group.  Registers: v[48:49] the lane's entry, v[54:55] the window, v[64:71] temporaries, v84 the
// lane's state address, v87 the read head - 32.  Entries hold ABSOLUTE LDS addresses (pack_entry) and the ring's address is an
// immediate: S must start at LDS address 0 (checked by the caller, which otherwise keeps the C++ form).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// walk record, 8 bytes: LL, ML, OF state addresses (LDS: < 2^16) and the low 16 bits of (read head - 32) -- a sequence consumes < 128 bits, so the
// planner, which knows where the walk started (Ctl::walk_g0), unwraps the positions chunk by chunk (plan_wave)
__device__ __forceinline__ uint64_t walk_record(uint32_t vL, uint32_t vM, uint32_t vO, uint32_t gm) { return (uint64_t)(vL | (vM << 16)) | ((uint64_t)(vO | (gm << 16)) << 32); }
constexpr uint32_t kWalkGroup = 8;
constexpr uint32_t kWalkLag = 32;
#ifndef MZD_WALK_YIELD
#define MZD_WALK_YIELD 448
#endif
constexpr uint32_t kWalkYield = MZD_WALK_YIELD;
#define MZD_SDWA_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define MZD_DPP_ALL " row_mask:0xf bank_mask:0xf\n"
// the record: lane k of quad 0 stores 16 bits -- the three state addresses; the fourth lane's are the read head's (MZD_REC_NOPOS = 0)
// or nothing anybody reads (1: the planner works the positions out of the states, plan_wave: a select less per step)
#ifndef MZD_REC_NOPOS
#define MZD_REC_NOPOS 1
#endif
#if MZD_REC_NOPOS
#define MZD_WALK_REC(RECOFF) "global_store_short %[woff], v84, %[base] offset:" RECOFF "\n"
#else
#define MZD_WALK_REC(RECOFF) "v_cndmask_b32_e64 v70, v84, v87, %[l3]\n" "global_store_short %[woff], v70, %[base] offset:" RECOFF "\n"
#endif
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
    "v_lshrrev_b32_e32 v71, 3, v87\n" \
    "v_and_b32_e32 v71, 0x1ffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    MZD_WALK_REC(RECOFF)                                                 /* the NEXT step's record: the state as it is now, 16 bits a field */ \
    "v_and_or_b32 %[av], v87, 31, 32\n" \
    TAIL
#define MZD_WALK_SLACK "v_min3_i32 %[slack], %[slack], %[sa], %[sb]\n"
// A: the lane's state address (lane & 3: LL, ML, OF, the dummy); woff: byte offset of the lane's dword of the next record
__device__ __forceinline__ void walk_run_asm(uint32_t& A, uint32_t& Gm, uint32_t& woff, int32_t& slack, uint32_t& n,
                                             int32_t pv, uint32_t& startA, uint32_t& startG, int32_t thresh, uint32_t prog_lds,
                                             __attribute__((address_space(1))) uint8_t* gwalk) {
    static_assert(kRingBytes - 4 == 0x1ffc && offsetof(Shared, ring) == 0, "the window address mask / the ring's place are spelled out in MZD_WALK_STEP");
    static_assert(kWalkGroup == 8 && kWalkLag == 32 && kWalkLag >= kWalkGroup + 2, "spelled out below");
    uint32_t av, sa, sb;
    uint64_t saved_exec;
    const uint64_t l3 = 0x8888888888888888ull; // lane 3 of every quad: its record dword is the read head
// The lanes that run the loop: all 64 (16 quads doing the same work) when the wavefront has its SIMD to itself -- few active lanes
    // issue VALU work 2-4x slower there (tools/micro/exec_micro.hip) --, quad 0 alone when the launch fills the machine: with four
    // wavefronts on the SIMD that penalty is gone, and the walkers' table reads no longer take a quarter of the CU's LDS bandwidth from
    // the copiers (64 lanes x 8 bytes per read, twice a step, four walkers per CU).  Measured: cfg2 +2.8 %, one file alone -2.4 % -> by launch size.
    const uint64_t lanes = gridDim.x > 512 ? 0xFull : ~0ull;
    asm volatile(
        "v_mov_b32_e32 v84, %[A]\n v_mov_b32_e32 v87, %[Gm]\n"
        "s_and_saveexec_b64 %[ex], %[lanes]\n" // (the incoming mask is kept and put back: the loop runs on `lanes` of the lanes that were active)
        "v_lshrrev_b32_e32 v71, 3, v87\n"
        "ds_read_b64 v[48:49], v84\n"
        "v_and_b32_e32 v71, 0x1ffc, v71\n"
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
        MZD_WALK_STEP("%[sb]", "64", MZD_WALK_SLACK "v_add_u32_e32 %[woff], 64, %[woff]\n v_subrev_u32_e32 v69, %[thresh], v87\n v_min_i32_e32 v69, v69, %[slack]\n v_cmp_gt_i32_e32 vcc, 0, v69\n")
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
        : [base] "s"(gwalk), [thresh] "s"(thresh), [prog] "v"(prog_lds), [l3] "s"(l3), [lanes] "s"(lanes)
        : "v48", "v49", "v54", "v55", "v64", "v65", "v66", "v67", "v69", "v70", "v71", "v84", "v87", "vcc", "scc", "memory");
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
        vL = vL * 8 + kLdsLL; vO = vO * 8 + kLdsOF; vM = vM * 8 + kLdsML; // state addresses (entries hold addresses: pack_entry)
    }
    const uint8_t* const tL = reinterpret_cast<const uint8_t*>(&S); // (one base: the states are offsets into S)
    const uint8_t* const tM = tL;
    const uint8_t* const tO = tL;
    uint32_t i = 0;
    const uint32_t nupd = nseq - 1; // sequences followed by a state update
    const bool lds_at_zero = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)S.ring == 0; // (walk_run_asm spells LDS addresses out)
    const uint32_t prog_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)prog;
    uint32_t Gm = G - 32; // the loop carries the read head minus 32 (saves an add per sequence)
    if (lane == 0) S.c.walk_g0 = Gm; // (the first record's full position: the planner reads it behind the first published progress)
    // One careful step of the chain: the window moves down a dword at a time until the sequence fits (long extra-bit
    // fields: about one sequence in hundreds).  The hot form (walk_run_asm) has no such branch -- a branch on freshly
    // loaded LDS data costs ~35 cycles per sequence on a lone wavefront -- it only notes that a group met such a sequence.
    auto careful_step = [&]() -> bool { // true: the sequence was wider than the first window
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
        *(__attribute__((address_space(1))) uint64_t*)(gwalk + woff) = walk_record(vL, vM, vO, Gm);
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
    while (i < nupd) {
        WSTAT(6, 0);
        // Where the copier cannot keep up (files of many short matches: its LDS rounds under a full machine), a walker further ahead only
        // takes issue slots from the copiers it shares its SIMD with: it yields while its own file's copier is more than kWalkYield
        // sequences behind (checked each time the assembly run returns: per KiB of bitstream).  Measured (thresholds 320 .. 2048, side by side): 400 .. 512 is best -- cfg2 +3.5 %, cfg2x8 +4 %, cfg3 +1 %; files
        // whose copier keeps up (cfg3's sequence-heavy classes) keep the walker in front.
        if (i > kWalkYield + 64 * flag_load_u(&S.c.copy_prog)) MZD_SETPRIO(MZD_PRIO_WALK_YIELD); else MZD_SETPRIO(MZD_PRIO_WALK);
        // keep the ring one group ahead of the read head
        while (st.lowest > 0 && (int32_t)Gm < st.lowest * (int32_t)(kChunk * 8) + kLook) {
            st.lowest--;
            ring_load_chunk(st, st.lowest, lane);
            WSTAT(0, 1);
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
            const uint32_t q = (uint32_t)lane & 3;
            uint32_t A = q == 0 ? vL : (q == 1 ? vM : (q == 2 ? vO : kLdsWalkDummy));
            uint32_t wl = woff + 2 * q, startA, startG;
            walk_run_asm(A, Gm, wl, slack, n, (int32_t)i - (int32_t)kWalkLag, startA, startG, thresh, prog_lds, gwalk);
            i += n0 - n;
            woff += 8 * (n0 - n);
            Gm = (uint32_t)__builtin_amdgcn_readfirstlane((int)Gm); // (the loop may have run on quad 0 only)
            vL = __builtin_amdgcn_readlane(A, 0); vM = __builtin_amdgcn_readlane(A, 1); vO = __builtin_amdgcn_readlane(A, 2);
            WSTAT(2, 1);
            if (__builtin_amdgcn_ballot_w64(slack < 0) != 0) { // the last group is void: once more from its start, carefully
                i -= kWalkGroup; woff -= 8 * kWalkGroup;
                vL = __builtin_amdgcn_readlane(startA, 0); vM = __builtin_amdgcn_readlane(startA, 1); vO = __builtin_amdgcn_readlane(startA, 2); Gm = (uint32_t)__builtin_amdgcn_readfirstlane((int)startG);
                // ... up to and including the first sequence that needed the wider window (what follows goes back to the hot form)
                for (uint32_t k = 0; k < kWalkGroup; k++) { i++; if (careful_step()) break; }
                WSTAT(4, 1);
            }
        } else {
            const uint32_t stop = left < kWalkGroup ? nupd : i + kWalkGroup;
            for (; i < stop; i++) careful_step();
        }
        if ((int32_t)(Gm + 32 - Gzero) < 0) return MZD_E_CORRUPT; // over-read
    }
#if defined(MZD_STAMPS) && defined(MZD_EXP_WALKSTAT)
    if (lane == 0) for (int k_ = 0; k_ < 8; k_++) S.cdiag[k_] = ws_[k_];
#endif
    MZD_SETPRIO(MZD_PRIO_WALK);
    G = Gm + 32;
    // last sequence: extra bits only
    {
        uint64_t eL, eM, eO;
        __builtin_memcpy(&eL, tL + vL, 8);
        __builtin_memcpy(&eM, tM + vM, 8);
        __builtin_memcpy(&eO, tO + vO, 8);
        *(__attribute__((address_space(1))) uint64_t*)(gwalk + woff) = walk_record(vL, vM, vO, G - 32);
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

