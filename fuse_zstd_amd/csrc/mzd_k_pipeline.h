// mzd_k_pipeline.h -- ONE compressed block through the workgroup's pipeline: the role dispatch both drivers share.
// Included by mzd_kernels.hip inside namespace mzd, behind the phase headers; not a translation unit of its own.
//
//   wave 0  K0 sequence header, K3 tables, K4a serial state walk            (mzd_k_tables*.h, mzd_k_walk.h)
//   wave 1  K1 Huffman tree / K2 first stream, then the copying half of K5  (mzd_k_huffman.h, mzd_k_execute.h)
//   wave 2  K2 remaining streams, then K7 hashing behind the copier         (mzd_k_xxh64.h)
//   wave 3  K4b plan: fields, repeat offsets, positions                     (mzd_k_execute.h)
//
// The drivers differ only in where a block's predecessor state comes from, and that is what TASKS selects:
//   TASKS = false (mzd_decode_kernel_files): the workgroup decodes its file's blocks in order -- tables, repeat offsets,
//           output position and checksum state are simply still in LDS / registers; the walking wavefront may already have
//           parsed this block's headers while the previous file was finishing (`block_pre`), and parses the next file's when
//           it is done with a file's last block.
//   TASKS = true  (mzd_decode_kernel_tasks): the block is a task of its own; inherited tables are fetched from the file's
//           table area at version t, rebuilt ones published at t + 1; the copier and the hasher wait for the predecessor
//           task (FileState::copied == t) for position, repeat offsets and XXH64 state; offsets in the plan stay symbolic.
#pragma once
#ifndef MZD_MIRROR_MIN
#define MZD_MIRROR_MIN 2048 // bytes of finished output before the hashing wavefront turns to the host mirror
#endif

// (BlockArgs: mzd_k_common.h -- the workgroup's copy lives in the LDS image)

// What the roles of one block share (all values wave-uniform).  The roles are separate functions -- each wavefront calls
// exactly one, so each gets a register allocation of its own instead of one allocation for all four roles' live values.
// (At most 64 bytes: the roles take it by value, and a bigger struct travels through the private segment -- stores by every lane at
//  every role call, loads in the callee.  What can be derived is: the launch's arguments, the thread index, the wavefront's LDS
//  segment, the place of a literal-only block.)
template <bool TASKS> struct BlockRun {
    const BlockArgs& b;
    int lane, wave;
    uint32_t lit_type, nlit, streams, nseq, seq_len;
    uint64_t lit_off, seq_off;
    const uint8_t* lit;   // the block's literals: inside the input (raw) or the workgroup's literal buffer
    __device__ __forceinline__ uint8_t* hseg() const { return wave == 1 ? S.stage + 2064 : (wave == 2 ? S.hseg2 : (wave == 0 ? S.ring : S.ring + 4096)); } // this wavefront's 2 KiB Huffman segment in LDS
    __device__ __forceinline__ uint8_t* place() const { return TASKS ? b.dst : b.dst + b.out0; } // where a literal-only block is decoded in place

    // nseq / seq_off / seq_len once the sequence header is parsed (by the walking wavefront); false: the block failed
    __device__ __forceinline__ bool get_seq() {
        Ctl& c = S.c;
        if (!spin_ge(&c.seq_parsed, 1, &c.err) || __atomic_load_n(&c.err, __ATOMIC_RELAXED)) return false;
        nseq = c.nseq; seq_off = c.seq_off; seq_len = c.seq_len;
        return true;
    }
    // A block without sequences IS its literals: the Huffman streams are then decoded straight into the output (no literal
    // buffer, no copy), provided they fit and the output position is known (a task: only the file's first)
    __device__ __forceinline__ bool lit_in_place() const {
        if (TASKS) return lit_type >= 2 && nseq == 0 && b.t == 0 && nlit <= b.cap;
        return lit_type >= 2 && nseq == 0 && nlit <= b.cap - b.out0;
    }
    // K2 worker: take Huffman streams from the block's queue until none is left
    __device__ __forceinline__ void huf_streams(uint32_t max_take) {
        Ctl& c = S.c;
        const uint32_t hl = c.huf_log;
        uint8_t* const lbase = lit_in_place() ? place() : b.lit_buf;
        // libzstd 1.5's fast loops take a four-stream section when its table is indexed by 11 bits, every stream has >= 8 bytes and the fourth
        // stream's share starts inside the output; they do not insist on exact consumption (mzd_k_huffman.h: kHufStrict)
        const bool fast_loops = streams == 4 && hl <= 11 && c.s_len[0] >= 8 && c.s_len[1] >= 8 && c.s_len[2] >= 8 && c.s_len[3] >= 8 && c.s_n[3] != 0;
        for (uint32_t took = 0; took < max_take; took++) {
            // every lane takes part (lanes != 0 add 0): no divergent region around the returning atomic
            uint32_t st = __atomic_fetch_add(&c.next_stream, lane == 0 ? 1u : 0u, __ATOMIC_RELAXED);
            st = (uint32_t)__builtin_amdgcn_readfirstlane(st);
            if (st >= streams || st >= 4) break;
#ifdef MZD_EXP_STREAMSTAMP // (diagnostic: when which wavefront takes stream st, and when it is done -- slots 0..3 and 4, 5, 6, 9)
            if (lane == 0) { S.sst[st] = ((__builtin_readcyclecounter() - S.tstart) << 2) | (uint64_t)wave; if (took == 0) S.swv[4 * wave + 3] = __builtin_readcyclecounter() - S.tstart; }
#endif
            int r = 0;
            if (!__atomic_load_n(&c.err, __ATOMIC_RELAXED))
                r = huf_stream_wave(b.blk + c.s_off[st], c.s_len[st], lbase + c.s_out[st], c.s_n[st], hl, hseg(), (wave == 2 && kSeg2Bytes < 2048) ? kSegBitsHalf : kSegBits, lane,
                                    fast_loops ? c.s_off[st] - (c.s_off[0] - 6) : kHufStrict);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) { post_err(&c.err, r); __atomic_fetch_or(&c.streams_mask, 1u << st, __ATOMIC_RELAXED); __atomic_fetch_add(&c.streams_done, 1u, __ATOMIC_RELAXED); }
#ifdef MZD_EXP_STREAMSTAMP
            if (lane == 0) S.sst[4 + st] = __builtin_readcyclecounter() - S.tstart;
#endif
        }
    }
    // wavefronts 0 and 3 decode literals too once their own role is over (at once in a block without sequences)
    __device__ __forceinline__ void huf_helper() {
        Ctl& c = S.c;
        if (lit_type < 2) return;
#ifdef MZD_EXP_STREAMSTAMP
        if (lane == 0) S.swv[4 * wave + 2] = __builtin_readcyclecounter() - S.tstart;
#endif
        if ((TASKS || lit_type == 2) && !spin_ge(&c.huf_fill, 2, &c.err)) return; // (a task fetches even an inherited table)
        if (!get_seq()) return;
        huf_streams(4);
    }
    // TASKS: the file's tables at version t (what the predecessor left): one lane waits, the wavefront copies
    __device__ __forceinline__ bool wait_tables() const {
        int ok = 1;
        if (lane == 0) ok = g_wait_ge(&b.fs->tables_ver, b.t, 3) ? 1 : 0;
        ok = __builtin_amdgcn_readfirstlane(ok);
        return ok != 0;
    }
};

// ---- wave 0: K0 sequence header, K3 tables, K4a serial state walk
template <bool TASKS>
__device__ __noinline__ void role_walk(BlockRun<TASKS> r) {
    MZD_IN_LDS(&r.b);
    Ctl& c = S.c;
    const BlockArgs& b = r.b;
    const int lane = r.lane;
#ifdef MZD_EXP_STREAMSTAMP
    if (lane == 0) S.swv[0] = __builtin_readcyclecounter() - S.tstart;
#endif
    MZD_SETPRIO(MZD_PRIO_WALK); // header parse, tables and walk are one serial chain: the block's critical path
    if (lane == 0 && !b.block_pre) parse_seq_header(c, S.stage + 256, r.seq_len, 256);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) flag_store(&c.seq_parsed, 1);
    TFIN(6);
#ifdef MZD_EXP_STREAMSTAMP
    const bool gs_ = r.get_seq();
    if (lane == 0) S.swv[1] = __builtin_readcyclecounter() - S.tstart;
    if (gs_ && r.nseq) {
#else
    if (r.get_seq() && r.nseq) {
#endif
        int rc = 0;
        if (TASKS) {
            const bool inherit = !b.frame_first && (c.mode[0] == 3 || c.mode[1] == 3 || c.mode[2] == 3);
            if (inherit) { // repeat mode: the table the previous block used (another workgroup built it)
                FileState* const fs = b.fs; TableArea* const ta = b.ta;
                if (!r.wait_tables()) { DEVSITE(7); rc = MZD_E_DEVICE; }
                else if (!g_ld(&fs->fse_valid)) rc = MZD_E_CORRUPT;
                else {
                    if (c.mode[0] == 3) { for (int i = lane; i < 512; i += 64) S.ll[i] = g_ld(&ta->ll[i]); if (lane == 0) c.al[0] = g_ld(&fs->al[0]); }
                    if (c.mode[1] == 3) { for (int i = lane; i < 256; i += 64) S.of[i] = g_ld(&ta->of[i]); if (lane == 0) c.al[1] = g_ld(&fs->al[1]); }
                    if (c.mode[2] == 3) { for (int i = lane; i < 512; i += 64) S.ml[i] = g_ld(&ta->ml[i]); if (lane == 0) c.al[2] = g_ld(&fs->al[2]); }
                }
            }
        } else if (lane == 0 && (c.mode[0] != 3 || c.mode[1] != 3 || c.mode[2] != 3)) c.lds_dict_fse = 0; // no longer the dictionary's
        if (!rc) build_tables_wave(lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) { post_err(&c.err, rc); flag_store(&c.tables_ready, 1); }
        TFIN(5);
        if (!rc) {
            MZD_SETPRIO(MZD_PRIO_WALK); // the chain is the critical path: win issue arbitration on this SIMD
            rc = walk_sequences_wave(b.src + r.seq_off, r.seq_len, r.nseq, b.walk, &c.walk_prog, lane);
            MZD_SETPRIO(0);
        }
        if (rc == kWalkInexact) { rc = 0; if (lane == 0) c.walk_inexact = 1; } // (every record is there: the planner and the copier go on; the copier gives the verdict)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
            post_err(&c.err, rc);
            if (!TASKS) c.fse_valid = 1;
            flag_store(&c.walk_prog, rc ? kWalkFin : (r.nseq | kWalkFin)); // a failed walk publishes nothing
        }
        TFIN(0);
    }
    MZD_SETPRIO(MZD_PRIO_HELP);
    r.huf_helper();
    MZD_SETPRIO(0);
    if (!TASKS && b.last && b.pos0 + b.bsize + (b.hashing ? 4u : 0u) == b.n) { // this block closes the file: the next file's headers, meanwhile
        MZD_SETPRIO(MZD_PRE_PRIO); pre_parse_next(*r.b.args, lane); MZD_SETPRIO(0);
    }
}

// ---- wave 3: (a task: publishes the tables for its successor,) K4b plan: fields, repeat offsets, positions
template <bool TASKS>
__device__ __noinline__ void role_plan(BlockRun<TASKS> r, FollowHook* hook = nullptr) {
    MZD_IN_LDS(&r.b);
    Ctl& c = S.c;
    const BlockArgs& b = r.b;
    const int lane = r.lane;
#ifdef MZD_EXP_STREAMSTAMP
    if (lane == 0) S.swv[12] = __builtin_readcyclecounter() - S.tstart;
#endif
    const bool seq_ok = r.get_seq();
#ifdef MZD_EXP_STREAMSTAMP
    if (lane == 0) S.swv[13] = __builtin_readcyclecounter() - S.tstart;
#endif
    if (TASKS && seq_ok && !b.is_final) {
        // the successor's inheritance: once this block's tables are final (and whatever it inherits itself has been read),
        // the kinds it rebuilt go to the file's table area and the version moves on (a file's last task has nobody to publish for)
        FileState* const fs = b.fs; TableArea* const ta = b.ta;
        bool ok = true;
        if (r.nseq) ok = spin_ge(&c.tables_ready, 1, &c.err);
        if (ok && r.lit_type >= 2) ok = spin_ge(&c.huf_fill, 2, &c.err);
        if (ok && !__atomic_load_n(&c.err, __ATOMIC_RELAXED) && r.wait_tables()) {
            if (!b.last) {
                const bool fse_new = r.nseq != 0 || (b.frame_first && c.fse_valid), huf_new = r.lit_type == 2 || (b.frame_first && c.huf_valid);
                if (fse_new) {
                    for (int i = lane; i < 512; i += 64) { g_st(&ta->ll[i], S.ll[i]); g_st(&ta->ml[i], S.ml[i]); }
                    for (int i = lane; i < 256; i += 64) g_st(&ta->of[i], S.of[i]);
                }
                if (huf_new) for (int i = lane; i < (int)kHufWords; i += 64) g_st(&reinterpret_cast<uint32_t*>(ta->huf)[i], reinterpret_cast<const uint32_t*>(S.huf)[i]);
                if (lane == 0) {
                    if (fse_new) { g_st(&fs->al[0], c.al[0]); g_st(&fs->al[1], c.al[1]); g_st(&fs->al[2], c.al[2]); g_st(&fs->fse_valid, 1u); }
                    else if (b.frame_first) g_st(&fs->fse_valid, 0u);
                    if (huf_new) { g_st(&fs->huf_log, c.huf_log); g_st(&fs->huf_valid, 1u); }
                    else if (b.frame_first) g_st(&fs->huf_valid, 0u);
                }
            }
            g_settle();
            if (lane == 0) { g_publish(&fs->tables_ver, b.t + 1); c.tables_published = 1; }
        }
    }
    if (seq_ok && r.nseq) {
        int rc = MZD_E_CORRUPT;
        if (spin_ge(&c.tables_ready, 1, &c.err)) {
            // (a task plans before its predecessor has finished: the repeat offsets at its start are unknown unless it opens the frame)
            PlanCtx px{b.walk, b.src + r.seq_off, &c.walk_prog, r.nlit, (!TASKS || b.frame_first) ? 1u : 0u, {c.rep[0], c.rep[1], c.rep[2]}, b.walk, r.seq_len, hook};
            MZD_SETPRIO(MZD_PRIO_PLAN);
            rc = plan_wave(b.seqs, r.nseq, px, lane);
            MZD_SETPRIO(0);
            if (rc == kPlanBlockTooLong) rc = 0; // (not an error yet: Ctl::plan_too_long, copy_wave)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
            post_err(&c.err, rc);
            flag_store(&c.plan_prog, flag_load(&c.plan_prog) | kPlanFin);
        }
        TFIN(3);
    }
    if (TASKS && r.b.args->resolve) { // resolving launches: the repeat offsets go on to the successor as soon as this block's transform is known
        if (lane == 0) S.res[3] = rep_hop(b.fs, b.t, b.frame_first, seq_ok && r.nseq != 0 && !__atomic_load_n(&c.err, __ATOMIC_RELAXED), S.res_rep) ? 1u : 0u;
    }
    MZD_SETPRIO(MZD_PRIO_HELP);
    r.huf_helper(); // the ring (its staging area) is free: the walker has finished before the planner does
    MZD_SETPRIO(0);
}

// ---- waves 1 and 2: K1 Huffman table (wave 1), K2 literal streams, then the copying half of K5 (wave 1) / K7 behind it (wave 2)
// phase: 3 both halves; 1 the literals only (a resolving task: mzd_k_resolve.h executes the block); 2 copy / hash only (its fallback)
template <bool TASKS>
__device__ __noinline__ void role_literals_then_copy_or_hash(BlockRun<TASKS> r, uint64_t& xv, uint64_t& xstripes, uint64_t& mirrored, int phase) {
    MZD_IN_LDS(&r.b);
    Ctl& c = S.c;
    const BlockArgs& b = r.b;
    const KernelArgs& a = *r.b.args;
    const int lane = r.lane, wave = r.wave, tid = r.wave * 64 + r.lane;
    const uint32_t lit_type = r.lit_type, nlit = r.nlit, streams = r.streams, t = b.t;
    uint8_t* const dst = b.dst;
    const bool frame_first = b.frame_first;
    int rc = 0;
    if (phase & 1) {
    // the literals gate the copier (the tail of the block): the copying wavefront's tree + first stream run at the
    // copier's priority, the remaining streams just below
    if (wave == 1) MZD_SETPRIO(MZD_PRIO_COPY); else MZD_SETPRIO(MZD_PRIO_PLAN);
    if (lit_type == 2 || (TASKS && lit_type == 3)) { // K1: the Huffman table, by wavefront 1: built from this block's tree, or (a task) inherited
        if (wave == 1) {
            int hr = 0;
            if (lit_type == 2) { // weights: serial (lane 0, from an LDS copy of the description); table: the whole wavefront
                const uint32_t tl = c.huf_tree_len; // <= 129 bytes
                for (uint32_t k = (uint32_t)lane; k < tl + 8; k += 64) S.stage[1024 + k] = k < tl ? b.blk[c.huf_tree_off + k] : 0;
                int used = 1;
                if (lane == 0) { if (!TASKS) c.lds_dict_huf = 0; used = read_huf_weights_staged(1024, c.huf_tree_len); } // (the table is no longer a dictionary's)
                used = __builtin_amdgcn_readfirstlane(used);
                TFIN(7);
                hr = used <= 0 ? MZD_E_CORRUPT : finish_huf_table_wave(lane);
                if (!TASKS && lane == 0 && !hr) c.huf_valid = 1;
            } else if (!frame_first) { // treeless: the table of the previous compressed-literals block
                FileState* const fs = b.fs; TableArea* const ta = b.ta;
                if (!r.wait_tables()) { DEVSITE(8); hr = MZD_E_DEVICE; }
                else if (!g_ld(&fs->huf_valid)) hr = MZD_E_DICT; // (treeless literals without a tree: libzstd's dictionary_corrupted)
                else {
                    for (int i = lane; i < (int)kHufWords; i += 64) reinterpret_cast<uint32_t*>(S.huf)[i] = g_ld(&reinterpret_cast<const uint32_t*>(ta->huf)[i]);
                    if (lane == 0) c.huf_log = g_ld(&fs->huf_log);
                }
            }
            // (treeless literals without a tree are found before anything else in the block: the class outranks what the walking
            //  wavefront may have posted about the sequence section meanwhile)
            if (lane == 0 && hr) { if (hr == MZD_E_DICT) __atomic_store_n(&c.err, (int32_t)MZD_E_DICT, __ATOMIC_RELAXED); else post_err(&c.err, hr); }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag_store(&c.huf_fill, 2);
            TFIN(8);
        }
#ifdef MZD_EXP_STREAMSTAMP
        if (lane == 0 && wave == 2) S.swv[4 * wave + 0] = __builtin_readcyclecounter() - S.tstart;
#endif
        spin_ge(&c.huf_fill, 2, &c.err);
#ifdef MZD_EXP_STREAMSTAMP
        if (lane == 0) S.swv[4 * wave + 1] = __builtin_readcyclecounter() - S.tstart;
#endif
    }
    const bool failed = __atomic_load_n(&c.err, __ATOMIC_RELAXED) != 0;
    if (lit_type == 1) { // RLE literals
        uint32_t w = (uint32_t)b.src[r.lit_off] * 0x01010101u;
        for (uint32_t k = (uint32_t)(tid - 64) * 16; k < nlit; k += 128 * 16)
            *reinterpret_cast<uint4*>(b.lit_buf + k) = make_uint4(w, w, w, w); // lit_buf has slack past nlit
    } else if (lit_type >= 2 && !failed && r.get_seq()) { // K2: the copying wavefront decodes one stream and then
        // (MZD_W3: the other literal wavefront plans behind its streams, so the copying one takes two of the four)
        r.huf_streams(wave == 1 && !r.lit_in_place() && (phase & 2) ? (MZD_W3 ? 2u : 1u) : 4u); // copies behind the literals; wavefront 2 (and idle ones) drain the queue
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
        post_err(&c.err, rc);
        __atomic_fetch_add(&c.lit_done, 1u, __ATOMIC_RELAXED);
    }
    MZD_SETPRIO(0);
    if (wave == 1) TFIN(4);
    }
#if !MZD_W3
    if (phase == 1) { // a resolving task: these two wavefronts build the block's byte map behind the planner (mzd_k_resolve.h)
        if (r.get_seq() && r.nseq) resolve_build_follow(a.resolve_map + (size_t)(a.wg0 + vblock()) * kResMapStride, b.seqs, b.walk, r.nseq, (uint32_t)(wave - 1), lane);
        return;
    }
#endif
    if (!(phase & 2)) return;
    if (wave == 1) { // the copying half of K5
        uint64_t opos = b.out0;
        rc = MZD_E_CORRUPT;
        bool run = true;
        uint64_t fstart = c.frame_out0;
        uint32_t r0 = c.rep[0], r1 = c.rep[1], r2 = c.rep[2];
        if (TASKS) { // it needs the predecessor's output, position and repeat offsets
            if (lane == 0) { load_pred(b.fs, t); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); flag_store(&c.pred_ready, 1); }
            spin_ge(&c.pred_ready, 1, &c.err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (t) g_acquire(); // the predecessor's output (another XCD's L2 may have held it)
            opos = c.pred_out;
            fstart = frame_first ? opos : c.pred_frame_out0;
            if (!frame_first) { r0 = c.pred_rep[0]; r1 = c.pred_rep[1]; r2 = c.pred_rep[2]; }
            if (c.pred_err) { rc = 0; run = false; } // the file has already failed: nothing to copy (the error travels on)
        }
        if (run && r.get_seq() && (lit_type != 1 || spin_ge(&c.lit_done, 2, &c.err))) { // RLE literals: both halves filled
            CopyCtx cx{b.seqs, dst, fstart, c.dict_content, c.dict_content_len, r.lit_in_place() ? r.place() : r.lit, nlit, b.cap, lit_type >= 2 ? streams : 0u,
                       {r0, r1, r2}, (TASKS && a.debug) ? b.seqs : nullptr};
            TFIN(9);
            MZD_SETPRIO(MZD_PRIO_COPY); // second on the critical path, behind the walker
            rc = copy_wave(r.nseq, cx, &opos, lane);
            MZD_SETPRIO(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
            post_err(&c.err, rc);
            c.out = opos; c.pos = b.pos0 + b.bsize;
            flag_store(&c.exec_done, 1);
            if (a.debug) {
                DebugSlot& ds = a.debug[a.wg0 + vblock()];
                ds.n_lit = nlit; ds.n_seq = r.nseq; ds.lit_is_raw = lit_type == 0 || r.lit_in_place(); ds.lit_raw_ptr = (uint64_t)(uintptr_t)(r.lit_in_place() ? r.place() : r.lit);
                if (TASKS && b.job == 0) atomicMax(&a.counter[1], (t << 12) | (a.wg0 + vblock())); // the slot that ran the last compressed block of job 0
            }
        }
        TFIN(1);
    } else if (b.hashing || b.dst2) { // wave 2: K7 hash and / or the host mirror, behind the copier while it works
        const bool hashing = b.hashing, mirror = b.dst2 != nullptr;
        bool run = true;
        uint64_t fstart = c.frame_out0;
        if (TASKS) { // state from the predecessor
            run = spin_ge(&c.pred_ready, 1, &c.err);
            if (run) {
                if (t) g_acquire();
                if (c.pred_err) run = false;
                else {
                    fstart = frame_first ? c.pred_out : c.pred_frame_out0;
                    xv = frame_first ? xxh_init(lane) : c.pred_xxh[lane & 3];
                    xstripes = frame_first ? 0 : c.pred_xstripes;
                    mirrored = c.pred_out; // a task mirrors its own block
                }
            }
        }
        if (run) {
            const uint8_t* fp = dst + fstart;
            for (uint32_t it = 0; it < (1u << 24); it++) {
                const uint32_t fin = flag_load(&c.exec_done);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint64_t pos = (TASKS && fin) ? c.out : __atomic_load_n(&c.exec_pos, __ATOMIC_RELAXED);
                bool did = false;
                if (hashing) {
                    uint64_t upto = pos > fstart ? (pos - fstart) / 32 : 0;
                    if (!fin) upto = upto >= xstripes + 64 ? xstripes + ((upto - xstripes) & ~7ull) : xstripes; // >= 2 KiB at a time, whole groups of 8 stripes
                    if (upto > xstripes) { xxh_advance(xv, xstripes, upto, fp, lane); did = true; }
                }
                if (mirror && !(TASKS && fin)) { // (a task leaves the rest to its completion: its successor must not wait for PCIe)
                    const uint64_t to = fin ? pos : (pos >= mirrored + MZD_MIRROR_MIN ? mirrored + ((pos - mirrored) & ~1023ull) : mirrored); // >= MZD_MIRROR_MIN bytes at a time, whole KiB
                    if (to > mirrored) { mirror_wave(dst, b.dst2, mirrored, to, lane); mirrored = to; did = true; }
                }
                if (fin) break;
                if (!did) { if (__atomic_load_n(&c.err, __ATOMIC_RELAXED)) break; __builtin_amdgcn_s_sleep(8); }
                if (it == (1u << 24) - 1 && lane == 0) DEVSITE(12);
            }
        }
        TFIN(2);
    }
}

// MZD_W3: a piece of the following wavefront's second job, done in the planner's waits: the checksum up to the copier's published position,
// at most 64 stripes (2 KiB) a call, and the host mirror.  true: something was done.
__device__ __noinline__ bool follow_step(const FollowHook& h, int lane) {
    Ctl& c = S.c;
    if (flag_load_u(&c.exec_done)) return false; // (the rest is the hashing loop's, behind the plan)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint64_t pos = __atomic_load_n(&c.exec_pos, __ATOMIC_RELAXED);
    bool did = false;
    if (h.hashing) {
        uint64_t upto = pos > h.fstart ? (pos - h.fstart) / 32 : 0;
        const uint64_t xs = *h.xstripes;
        upto = upto >= xs + 64 ? xs + 64 : xs; // whole groups of 8 stripes, 2 KiB at a time: the walker's next records are looked at in between
        if (upto > xs) { xxh_advance(*h.xv, *h.xstripes, upto, h.frame, lane); did = true; }
    }
    if (h.dst2) {
        const uint64_t m = *h.mirrored;
        const uint64_t to = pos >= m + MZD_MIRROR_MIN ? m + ((pos - m) & ~1023ull) : m;
        if (to > m) { mirror_wave(h.dst, h.dst2, m, to < m + 4096 ? to : m + 4096, lane); *h.mirrored = to < m + 4096 ? to : m + 4096; did = true; }
    }
    return did;
}

// One compressed block.  Returns false when its headers already failed (c.err is set): nothing was started.
template <bool TASKS>
__device__ __forceinline__ bool compressed_block(const KernelArgs& a, const BlockArgs& b, uint64_t& xv, uint64_t& xstripes, uint64_t& mirrored, int tid, int lane, int wave,
                                                 bool defer_copy = false) {
    Ctl& c = S.c;
    int err = 0;
    // K0/K1/K3 headers: where everything is; nothing is decoded yet.  The two header regions (<= 256 bytes each: literals
    // header + tree extent + jump table; sequence count, modes and the three normalized-count headers) are staged in LDS
    // first, so that lane 0's byte-wise parsing does not pay an HBM round trip per byte.
    TSTART();
    if (!b.block_pre) {
        for (uint32_t k = tid; k < b.bsize && k < 256; k += kWG) S.stage[k] = b.blk[k];
        grp_sync();
    }
    if (tid == 0) {
        c.huf_ready = 0; c.huf_fill = 0; c.lit_done = 0; c.walk_prog = 0; c.walk_inexact = 0; c.exec_done = 0; c.exec_pos = TASKS ? 0 : b.out0;
        c.next_stream = 0; c.streams_done = 0; c.streams_mask = 0;
        c.tables_ready = 0; c.plan_prog = 0; c.copy_prog = 0; c.plan_lit_used = 0; c.plan_too_long = 0; c.seq_parsed = 0;
        S.res[0] = 0; S.res[1] = 0; S.res_nsym = 0;
        c.rep_op[0] = 0 | (1 << 2) | (2 << 4); c.rep_op[1] = 0; c.rep_op[2] = 0; c.rep_op[3] = 0; // identity: a block without sequences
        if (!b.block_pre) parse_literals(c, S.stage, b.bsize);
    }
    uint32_t lit_type = 0, nlit = 0, streams = 0, seq_len = 0;
    uint64_t lit_off = 0, seq_off = 0;
    WG_SNAPSHOT(err = c.err; lit_type = c.lit_type; nlit = c.nlit; streams = c.streams; lit_off = c.lit_off;
                seq_off = c.seq_off; seq_len = c.seq_len);
    if (err) {
        if (TASKS && !b.frame_first && lit_type == 3 && b.bsize >= 3 && err != MZD_E_DICT) {
            // A block that continues a frame parses its headers before it knows what the frame has inherited (Ctl::huf_valid is
            // provisional): treeless literals without a tree are libzstd's first finding in the section, before its sizes
            if (tid == 0 && g_wait_ge(&b.fs->tables_ver, b.t) && !g_ld(&b.fs->huf_valid)) c.err = MZD_E_DICT;
            grp_sync();
        }
        return false;
    }
    if (!b.block_pre) {
        for (uint32_t k = tid; k < seq_len && k < 256; k += kWG) S.stage[256 + k] = b.src[seq_off + k];
        grp_sync();
    }
    // The sequence header (three normalized-count descriptions: a serial bit parse) is read by lane 0 of the walking
    // wavefront INSIDE the pipeline, so the literal side (Huffman tree, streams) starts at once.
    // 2 KiB of LDS per decoding wavefront, borrowed from buffers that are idle while literals decode: the copier's staging
    // buffers (wave 1); the walker's ring (waves 0, 3: they decode after the walk)
    BlockRun<TASKS> r{b, lane, wave, lit_type, nlit, streams, 0u, seq_len, lit_off, seq_off,
                      lit_type == 0 ? b.src + lit_off : b.lit_buf};
    static_assert(sizeof(BlockRun<TASKS>) <= 64, "passed in registers");
#if MZD_W3
    // three wavefronts: the walking one, the copying one, and one that decodes its share of the literals, PLANS behind the walker -- hashing
    // and mirroring in the planner's waits (follow_step) -- and finishes the checksum behind the copier
    if (wave == 0) role_walk<TASKS>(r);
    else if (wave == 1) role_literals_then_copy_or_hash<TASKS>(r, xv, xstripes, mirrored, 3);
    else {
        role_literals_then_copy_or_hash<TASKS>(r, xv, xstripes, mirrored, 1);
        FollowHook hk{&xv, &xstripes, &mirrored, b.dst + c.frame_out0, c.frame_out0, b.dst, b.dst2, b.hashing};
        role_plan<TASKS>(r, (b.hashing || b.dst2) ? &hk : nullptr);
        role_literals_then_copy_or_hash<TASKS>(r, xv, xstripes, mirrored, 2);
    }
#else
    if (wave == 0) role_walk<TASKS>(r);
    else if (wave == 3) role_plan<TASKS>(r);
    else role_literals_then_copy_or_hash<TASKS>(r, xv, xstripes, mirrored, defer_copy ? 1 : 3);
#endif
    return true;
}

#if !MZD_W3
// The deferred second half (a task that could not be resolved after all): wavefront 1 copies, wavefront 2 hashes / mirrors.
__device__ __forceinline__ void compressed_block_copy(const KernelArgs& a, const BlockArgs& b, uint64_t& xv, uint64_t& xstripes, uint64_t& mirrored, int tid, int lane, int wave) {
    Ctl& c = S.c;
    if (wave != 1 && wave != 2) return;
    BlockRun<true> r{b, lane, wave, c.lit_type, c.nlit, c.streams, 0u, c.seq_len, c.lit_off, c.seq_off,
                     c.lit_type == 0 ? b.src + c.lit_off : b.lit_buf};
    role_literals_then_copy_or_hash<true>(r, xv, xstripes, mirrored, 2);
}
#endif
