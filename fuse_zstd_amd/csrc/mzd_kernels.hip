// mzd_kernels.hip -- the zstd frame decoder as HIP kernels for gfx950 (MI355X / CDNA4).
//
// Replaces the arithmetic behind `zstd::stream::copy_decode` (reference src/main.rs:463-467;
// libzstd 1.5.6 via zstd-sys, reference Cargo.lock:2371-2396), written from the format
// (RFC 8878) for 64-lane wavefronts.  Not a port of libzstd.
//
// Mapping: one workgroup (4 wavefronts, ~34 KB of LDS, 4 workgroups per CU) decodes one block at a time in a
// persistent grid.  Two drivers share the block pipeline:
//   mzd_decode_kernel_files   a workgroup owns a file and walks its frames and blocks in order (launches in which
//                             no file can have more than one block);
//   mzd_decode_kernel_tasks   one block per task: the blocks of a frame run on different workgroups, the state
//                             between them (tables, repeat offsets, output position, checksum) is handed over in
//                             task order.
// Per compressed block, one role per wavefront, connected by LDS flags and unbounded HBM queues:
//   K0  headers                 lane 0, from LDS copies; the sequence header inside wavefront 0's role   (A.1, A.2)
//   K1  Huffman tree            lane 0 decodes the weights, the wavefront validates, ranks and fills      (A.4)
//   K2  Huffman literals        64 lanes per stream by self-synchronising sub-stream decode (memoized entry
//                               offsets) + DPP scan for the output offsets; streams handed out by a queue
//   K3  FSE tables x3           one wavefront, ballot-rank symbol spread                                  (A.3)
//   K4a FSE state walk          wavefront 0: the serial chain, tables + bitstream ring in LDS, 16-byte records (A.5)
//   K4b plan                    wavefront 3: fields, symbolic repeat offsets (DPP scan), positions, validation
//   K5  sequence execute        wavefront 1: runs of <= 64 sequences staged in LDS, prefetched HBM sources,
//                               LDS->LDS matches in rounds, 16-byte coalesced flushes                     (A.5)
//   K6  raw / RLE blocks        256 lanes, coalesced
//   K7  XXH64                   wavefront 2 behind the copier, groups of 8 stripes                         (A.6)
// Everything is integer/byte work bound by latency and instruction issue, so there is no MFMA here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <type_traits>

#include <type_traits>

#include "../../include/mzd.h"
#include "mzd_device.h"
#include "mzd_tables.h"

#ifndef MZD_PRIO_WALK
#define MZD_PRIO_WALK 3
#endif
#ifndef MZD_PRIO_WALK_YIELD
#define MZD_PRIO_WALK_YIELD 1 // (a walker far ahead of its copier: mzd_k_walk.h)
#endif
#ifndef MZD_PRIO_COPY
#define MZD_PRIO_COPY 2
#endif
#ifndef MZD_PRIO_PLAN
#define MZD_PRIO_PLAN 1
#endif
#ifndef MZD_LB_WAVES
#if MZD_W3
#define MZD_LB_WAVES 4 // five workgroups of three wavefronts a CU: 15 wavefronts, at most four a SIMD -- 128 registers
#else
#define MZD_LB_WAVES MZD_WGS_PER_CU
#endif
#endif
#ifdef MZD_EXP_NOPRIO // (experiment: no s_setprio instruction at all)
#define MZD_SETPRIO(x) ((void)0)
#else
#define MZD_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#ifndef MZD_PRIO_HELP
#define MZD_PRIO_HELP 0 // wavefronts 0 and 3 decoding literal streams once their own role is over (BlockRun::huf_helper)
#endif

namespace mzd {

#if defined(MZD_STAMPS) || defined(MZD_TFIN)
#define TFIN(k) do { if (lane == 0) S.tfin[k] = __builtin_readcyclecounter() - S.tstart; } while (0)
#define TSTART() do { if (tid == 0) { S.tstart = __builtin_readcyclecounter(); S.tfin[10] = S.tstart - S.ttask; } } while (0)
#ifdef MZD_TFIN_ABS
#define TTASK() do { if (tid == 0) { S.ttask = __builtin_readcyclecounter(); S.tabs = (uint64_t)wall_clock64(); } } while (0)
#else
#define TTASK() do { if (tid == 0) S.ttask = __builtin_readcyclecounter(); } while (0)
#endif
#ifdef MZD_TFIN_ABS // (experiment: absolute clock values -- the spread of workgroup starts and ends over a launch)
#define TTASK_END() do { if (tid == 0) { S.tfin[11] = (uint64_t)wall_clock64(); S.tfin[10] = S.tabs; } } while (0) /* 100 MHz, the same on every CU */
#else
#define TTASK_END() do { if (tid == 0) S.tfin[11] = __builtin_readcyclecounter() - S.ttask; } while (0)
#endif
#define TCOUNT(k, v) do { if (lane == 0) atomicAdd((unsigned long long*)&S.tfin[k], (unsigned long long)(v)); } while (0)
#ifdef MZD_EXP_STREAMSTAMP
#define TFIN_FLUSH() do { if (tid == 0 && a.debug) { for (int k_ = 0; k_ < 12; k_++) a.debug[a.wg0 + vblock()].tfin[k_] = S.tfin[k_]; \
    for (int k_ = 0; k_ < 4; k_++) a.debug[a.wg0 + vblock()].tfin[k_] = S.sst[k_]; \
    a.debug[a.wg0 + vblock()].tfin[4] = (S.swv[0] << 32) | (S.swv[1] & 0xFFFFFFFFull); a.debug[a.wg0 + vblock()].tfin[5] = (S.swv[2] << 32) | (S.swv[3] & 0xFFFFFFFFull); \
    a.debug[a.wg0 + vblock()].tfin[6] = (S.swv[12] << 32) | (S.swv[13] & 0xFFFFFFFFull); a.debug[a.wg0 + vblock()].tfin[7] = (S.swv[14] << 32) | (S.swv[15] & 0xFFFFFFFFull); } } while (0)
#else
#define TFIN_FLUSH() do { if (tid == 0 && a.debug) { for (int k_ = 0; k_ < 12; k_++) a.debug[a.wg0 + vblock()].tfin[k_] = S.tfin[k_]; } } while (0)
#endif
#else
#define TFIN(k)
#define TCOUNT(k, v)
#define TTASK()
#define TTASK_END()
#define TSTART()
#define TFIN_FLUSH()
#endif

// Diagnostic build only: per-phase cycle sums of the workgroup (lane 0), never in the product .so.
#ifdef MZD_STAMPS
#define STAMP_DECL uint64_t st_prev = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k) do { uint64_t t_ = __builtin_readcyclecounter(); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#define STAMP_FLUSH() do { if (tid == 0 && a.debug) { for (int k_ = 0; k_ < 8; k_++) { a.debug[a.wg0 + vblock()].stamp[k_] = st_acc[k_]; a.debug[a.wg0 + vblock()].cstamp[k_] = S.cdiag[k_]; } a.debug[a.wg0 + vblock()].stamp[2] = S.c.diag_slow; } } while (0)
#define CSTAMP_DECL uint64_t cs_prev = __builtin_readcyclecounter()
#ifdef MZD_EXP_WALKSTAT // (the copier's slots show the walker's statistics instead: tools/stamps.py prints them raw)
#define CSTAMP(k) do { } while (0)
#else
#define CSTAMP(k) do { uint64_t t_ = __builtin_readcyclecounter(); if (lane == 0) S.cdiag[k] += t_ - cs_prev; cs_prev = t_; } while (0)
#endif
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH()
#define CSTAMP_DECL
#ifdef MZD_MARKS // (reading the ISA: where the copier's phases begin)
#define CSTAMP(k) asm volatile("; ======== CSTAMP " #k)
#else
#define CSTAMP(k)
#endif
#endif



// ---- the phases, one header each (dependency order)
#include "mzd_k_common.h"
#include "mzd_k_tables.h"
#include "mzd_k_huffman.h"
#include "mzd_k_bytes.h"
#include "mzd_k_tables_wave.h"
#include "mzd_k_walk.h"
#include "mzd_k_execute.h"
#include "mzd_k_xxh64.h"
#include "mzd_k_headers.h"

// ---- inter-workgroup hand-over of the block-task driver (agent scope): a task publishes, its successor on another CU acquires
__device__ __forceinline__ uint32_t g_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifdef MZD_EXP_NOFENCE // (experiment builds only: what the cross-XCD fences cost; results are not valid)
__device__ __forceinline__ void g_acquire() {}
__device__ __forceinline__ void g_release() {}
#else
__device__ __forceinline__ void g_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void g_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
#endif
__device__ __forceinline__ void g_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Agent-scope fences write back / invalidate the XCD's whole L2 (buffer_wbl2 / buffer_inv): they are kept for the one thing
// that needs them -- the output bytes a successor on another XCD reads -- and everything small (task records, per-file
// state, table areas) travels through agent-scope atomic loads and stores, which are coherent by themselves.
// `g_settle` orders such stores before the flag that publishes them.
template <class T> __device__ __forceinline__ T g_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void g_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void g_settle() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
// ... and that is NOT enough across XCDs: with the data stores merely waited for (vmcnt) a successor on another XCD saw the flag
// before the data about once in a hundred runs of a 586-block file (the checksum chain's state: right bytes, wrong digest -- found
// when this round's faster task start changed the timing; `tools/stress_handover.py`).  Every flag that publishes such data is
// therefore stored behind an agent-scope RELEASE (L2 write-back + wait), like the one that publishes the output bytes.
__device__ __forceinline__ void g_publish(uint32_t* flag, uint32_t v) { g_release(); g_store(flag, v); }
// wait until *p >= want (bounded: a launch that lost a task must end, not hang); one lane calls this
__device__ __noinline__ bool g_wait_ge(const uint32_t* p, uint32_t want, uint32_t site = 0) {
    for (uint32_t it = 0; it < (1u << 23); it++) {
        if (g_load(p) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    DEVSITE(20 + site);
    (void)site;
    return false;
}
// lane 0: what the predecessor of task t left behind -> S.c.pred_*.  false: the launch is broken (timeout).
// load_pred_copy: position, repeat offsets, first error (FileState::copied); load_pred_hash: the checksum chain
// (FileState::hashed), and a checksum failure posted on it, which outranks whatever the bytes' chain says (it is earlier in
// the stream); load_pred: both.
__device__ __noinline__ bool load_pred_copy(const FileState* fs, uint32_t t) {
    Ctl& c = S.c;
    if (t == 0) {
        c.pred_err = 0; c.pred_out = 0; c.pred_frame_out0 = 0;
        c.pred_rep[0] = 1; c.pred_rep[1] = 4; c.pred_rep[2] = 8;
    } else {
        if (!g_wait_ge(&fs->copied, t)) { DEVSITE(9); c.pred_err = MZD_E_DEVICE; c.pred_out = 0; c.pred_frame_out0 = 0; return false; }
        c.pred_err = g_ld(&fs->err); c.pred_out = g_ld(&fs->out); c.pred_frame_out0 = g_ld(&fs->frame_out0);
        c.pred_rep[0] = g_ld(&fs->rep[0]); c.pred_rep[1] = g_ld(&fs->rep[1]); c.pred_rep[2] = g_ld(&fs->rep[2]);
    }
    return true;
}
__device__ __noinline__ bool load_pred_hash(const FileState* fs, uint32_t t) {
    Ctl& c = S.c;
    if (t == 0) {
        c.pred_xstripes = 0;
        for (int k = 0; k < 4; k++) c.pred_xxh[k] = 0;
    } else {
        if (!g_wait_ge(&fs->hashed, t)) { DEVSITE(10); c.pred_err = MZD_E_DEVICE; c.pred_xstripes = 0; return false; }
        c.pred_xstripes = g_ld(&fs->xstripes);
        for (int k = 0; k < 4; k++) c.pred_xxh[k] = g_ld(&fs->xxh[k]);
        const int32_t he = g_ld(&fs->herr);
        if (he) { c.pred_err = he; c.pred_out = g_ld(&fs->herr_out); }
    }
    return true;
}
__device__ __noinline__ bool load_pred(const FileState* fs, uint32_t t) {
    const bool a_ = load_pred_copy(fs, t);
    return a_ && load_pred_hash(fs, t);
}

// One lane.  The repeat-offset chain: wait for the predecessor's, apply this task's transform (identity unless it planned
// sequences), publish.  `rin` receives the offsets this task starts with.  false: the launch is broken (timeout).
__device__ __noinline__ bool rep_hop(FileState* fs, uint32_t t, bool frame_first, bool planned, uint32_t* rin) {
    Ctl& c = S.c;
    const bool ok = g_wait_ge(&fs->rep_ver, t, 1);
    uint32_t r0 = c.rep[0], r1 = c.rep[1], r2 = c.rep[2]; // a frame starts with its own (1, 4, 8 or the dictionary's)
    if (!frame_first) { r0 = g_ld(&fs->rep_e[0]); r1 = g_ld(&fs->rep_e[1]); r2 = g_ld(&fs->rep_e[2]); }
    rin[0] = r0; rin[1] = r1; rin[2] = r2;
    RepOp Rf;
    if (planned) { Rf.s = c.rep_op[0]; Rf.v0 = (int32_t)c.rep_op[1]; Rf.v1 = (int32_t)c.rep_op[2]; Rf.v2 = (int32_t)c.rep_op[3]; }
    else { Rf.s = 0 | (1 << 2) | (2 << 4); Rf.v0 = 0; Rf.v1 = 0; Rf.v2 = 0; }
    g_st(&fs->rep_e[0], rep_eval(Rf, 0, r0, r1, r2)); g_st(&fs->rep_e[1], rep_eval(Rf, 1, r0, r1, r2)); g_st(&fs->rep_e[2], rep_eval(Rf, 2, r0, r1, r2));
    g_settle();
    g_publish(&fs->rep_ver, t + 1);
    return ok;
}


#if !MZD_W3
#include "mzd_k_resolve.h"
#endif
#include "mzd_k_pipeline.h"

// The launch's last workgroup to finish zeroes the counter block of the lane's NEXT launch (KernelArgs::counter_next): no
// memset between launches.  counter[6] counts the groups that are done.
__device__ __forceinline__ void clean_next_counters(const KernelArgs& a, int tid) {
    if (tid == 0 && a.counter_next && atomicAdd(&a.counter[6], 1u) == vgrid() - 1)
        for (uint32_t k = 0; k < kCounterWords; k++) a.counter_next[k] = 0;
}

// ---- driver 1: one workgroup decodes a whole file, block after block.  Used when no file of the launch can have more
// than one block (every output capacity <= 128 KiB): nothing is forked, nothing is published, the file's state
// stays in registers and LDS.
#if MZD_PAIRS
#define MZD_FILES_KERNEL mzd_decode_kernel_pairs
#elif MZD_W3
#define MZD_FILES_KERNEL mzd_decode_kernel_files3
#else
#define MZD_FILES_KERNEL mzd_decode_kernel_files
#endif
__global__ __launch_bounds__(kWG * kGroupsMax, MZD_LB_WAVES) void MZD_FILES_KERNEL(KernelArgs) {
    const KernelArgs& a = launch_args();
    // Inside the group (mzd_k_common.h).  `wave` is the wavefront's ROLE.  The hardware puts a workgroup's wavefronts on the four SIMDs in
    // rotation: with one group a workgroup every SIMD holds one wavefront of each role; with two groups wavefronts w and w + 4 share a SIMD, and
    // the same role in both would put a CU's four copying wavefronts on ONE SIMD.  So a group's roles are rotated by its index, and by two more in
    // every other workgroup: the four groups of a CU then bring each SIMD one wavefront of each role.  (Measured with and without: no difference
    // in this kernel -- its loss is elsewhere, mzd_host.cpp: enqueue -- the rotation is kept as the placement that is right by construction.)
#ifndef MZD_ROT_BIT
#define MZD_ROT_BIT 8
#endif
    const int rot_ = grp_count() > 1 ? (int)(grp_index() + 2u * ((blockIdx.x >> MZD_ROT_BIT) & 1u)) : 0;
    const int lane = threadIdx.x & 63, wave = (int)(((threadIdx.x >> 6) + rot_) & 3), tid = wave * 64 + lane;
    // what the groups of a workgroup share must be in place before either of them moves: the barrier counters and the walkers' rendezvous
#if MZD_PAIRS
    if (tid == 0) { S.bar = 0; S.wk.active = 0; S.wk.state = 0; }
    __syncthreads(); // (the workgroup's only hardware barrier: every wavefront of both groups is still here)
#endif
    // A group's first ticket is its own index: one past the queue's end has nothing to do -- the launch behind the small-file
    // kernel when that kernel handed nothing on.  Leaving at once keeps most of the kernel's private-segment stores out of HBM (1 200
    // bytes per lane: the roles' register spills, and a copy per lane of the launch's arguments, whose address the roles take): an idle
    // launch of 256 workgroups wrote 22.5 MB (profiles/r02_cfg4_pmc.json), now 10 (the per-lane argument copy, made on entry).  A body
    // in a function of its own, called behind this test, brings that to 7 MB but costs the headline 3 % (measured): not taken.
    if (vblock() >= queue_len(a)) { clean_next_counters(a, tid); return; }
    const uint32_t slot = a.wg0 + vblock(); // this group's place in the scratch arrays
    uint8_t* const lit_buf = a.lit_scratch + (size_t)slot * kLitStride;
    uint4* const seqs = a.seq_scratch + (size_t)slot * kSeqStride;
    uint4* const walk = a.walk_scratch + (size_t)slot * kSeqStride;
    Ctl& c = S.c;
    if (tid < 36) S.ll_base[tid] = LL_BASE[tid];
    if (tid < 53) S.ml_base[tid] = ML_BASE[tid];
    if (tid == 0) S.walk_dummy = lds_base() + kLdsWalkDummy; // {own address, no bits}: what the walker's fourth lane follows
    if (tid == 0) { c.lds_dict_fse = 0; c.lds_dict_huf = 0; S.pre_job = kNoJob; S.pre_valid = 0; S.dcache.id = 0; S.took_first = 0; }

    for (;;) {
        TTASK();
        if (tid == 0) { // the file the walking wavefront took ahead (pre_parse_next), or the next one of the queue
            if (S.pre_job != kNoJob) { c.job = S.pre_job; c.t_valid = S.pre_valid | 2u; S.pre_job = kNoJob; S.pre_valid = 0; } // (bit 1: S.pj holds the job's table entry)
            else { c.job = take_job(a); c.t_valid = 0; }
        }
        uint32_t j, tv, lf = 0, lh = 0;
        WG_SNAPSHOT(j = c.job; tv = c.t_valid; lf = c.lds_dict_fse; lh = c.lds_dict_huf);
        bool pre = (tv & 1) != 0; // the first block's headers are already parsed (in S.c2)
        if (j >= a.njobs) break;
        const bool stashed = (tv & 2) != 0;
        const uint8_t* const src = stashed ? S.pj.src : a.jobs[j].src;
        const uint64_t n = stashed ? S.pj.n : a.jobs[j].src_len;
        uint8_t* const dst = stashed ? S.pj.dst : a.jobs[j].dst;
        const uint64_t cap = stashed ? S.pj.cap : a.jobs[j].dst_cap;
        const uint32_t job_dict = stashed ? S.pj.dict : a.jobs[j].dict;
        uint8_t* const dst2 = a.jobs[j].dst2; // (host mirror of the output, or null)
        uint64_t mirrored = 0;                // bytes of the file already mirrored (wave 2's)
        if (pre) { // the prepared control block replaces the current one: by all threads, a dword each (what must survive was read above)
            static_assert(sizeof(Ctl) % 4 == 0, "copied by dwords");
            for (uint32_t k = (uint32_t)tid; k < sizeof(Ctl) / 4; k += kWG) reinterpret_cast<uint32_t*>(&c)[k] = reinterpret_cast<const uint32_t*>(&S.c2)[k];
            grp_sync();
        }
        if (tid == 0) {
            if (pre) { c.lds_dict_fse = lf; c.lds_dict_huf = lh; c.job = j; }
            else { c.pos = 0; c.out = 0; c.err = 0; c.action = 0; }
            c.diag_slow = 0;
#ifdef MZD_STAMPS
            for (int k_ = 0; k_ < 8; k_++) S.cdiag[k_] = 0;
#endif
            if (j == 0) a.counter[1] = a.wg0 + vblock();
            if (job_dict > a.ndicts) c.err = MZD_E_DICT;
        }
        int err = 0;
        uint32_t action = 0;
        uint64_t xv = 0, xstripes = 0; // K7 state of the hashing wavefront (wave 2)
        STAMP_DECL;

        // ---------------- frames (K0)
        while (true) {
            const bool frame_pre = pre; // (only the first frame of the file can have been prepared)
            if (tid == 0 && !c.err && !frame_pre) parse_frame_or_skip(c, src, n, a.dicts, a.ndicts, job_dict);
            WG_SNAPSHOT(err = c.err; action = c.action);
            if (err || action == 2) break;
            if (action == 1) continue; // skippable frame
            xv = xxh_init(lane); xstripes = 0;
            const bool hashing = c.has_cksum != 0; // stable for the whole frame
            if (action == 3) { // dictionary: entropy tables, repeat offsets and content
                const DevDict* dd = &a.dicts[job_dict - 1];
                if (tid == 0 && S.dcache.id != job_dict) {
                    S.dcache.formatted = dd->formatted; S.dcache.huf_log = dd->huf_log; S.dcache.content_len = dd->content_len; S.dcache.content = dd->content;
                    for (int t_ = 0; t_ < 3; t_++) { S.dcache.al[t_] = dd->al[t_]; S.dcache.rep[t_] = dd->rep[t_]; }
                    S.dcache.id = job_dict;
                }
                grp_sync();
                if (S.dcache.formatted) {
                    // Config 5 (many small frames, one dictionary): a workgroup keeps the dictionary's tables resident in LDS
                    // from file to file -- such frames use them as they are (repeat-mode tables, treeless literals), so the
                    // 14 KB copy happens once per workgroup, not once per file.  Any block that rebuilds a table clears the mark.
                    const bool have_fse = c.lds_dict_fse == job_dict, have_huf = c.lds_dict_huf == job_dict;
                    grp_sync();
                    if (!have_fse) { // (an entry's low word is an LDS address in the image of a workgroup's first group: mzd_device.h)
                        const uint64_t rb = lds_base();
                        for (int i = tid; i < 512; i += kWG) { S.ll[i] = dd->ll[i] + rb; S.ml[i] = dd->ml[i] + rb; }
                        for (int i = tid; i < 256; i += kWG) S.of[i] = dd->of[i] + rb;
                    }
                    if (!have_huf) for (int i = tid; i < (int)kHufEntries; i += kWG) S.huf[i] = dd->huf[i];
                    if (tid == 0) {
                        c.lds_dict_fse = job_dict; c.lds_dict_huf = job_dict;
                        // (a prepared first block has its sequence header parsed already: only repeat-mode tables take the dictionary's log)
                        for (int t_ = 0; t_ < 3; t_++) if (!frame_pre || c.mode[t_] == 3) c.al[t_] = S.dcache.al[t_];
                        c.huf_log = S.dcache.huf_log; c.huf_valid = 1; c.fse_valid = 1;
                        c.rep[0] = S.dcache.rep[0]; c.rep[1] = S.dcache.rep[1]; c.rep[2] = S.dcache.rep[2];
                    }
                }
                if (tid == 0) { c.dict_content = S.dcache.content; c.dict_content_len = S.dcache.content_len; }
            }
            // ---------------- blocks
            uint32_t last = 0;
            while (true) {
                const bool block_pre = pre;
                pre = false;
                if (tid == 0 && !block_pre) parse_block_header(c, src, n);
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
                    // (the block's arguments live in LDS: the roles take them by reference, and on the stack they would be a copy per lane)
                    if (tid == 0) { S.ba = BlockArgs{src, n, dst, cap, dst2, src + pos0, bsize, pos0, out0, lit_buf, seqs, walk, last, hashing, block_pre, 0u, false, true, nullptr, nullptr, j, &a}; }
                    grp_sync();
                    if (!compressed_block<false>(a, S.ba, xv, xstripes, mirrored, tid, lane, wave)) { WG_SNAPSHOT(err = c.err); break; } // (an error of the block's headers: the class it posted stands)
                }
                if (btype == 2) { // the block's repeat-offset transform (the planner leaves it symbolic) -> the offsets after it
                    grp_sync();
                    if (tid == 0) {
                        RepOp Rf; Rf.s = c.rep_op[0]; Rf.v0 = (int32_t)c.rep_op[1]; Rf.v1 = (int32_t)c.rep_op[2]; Rf.v2 = (int32_t)c.rep_op[3];
                        const uint32_t a0 = c.rep[0], a1 = c.rep[1], a2 = c.rep[2];
                        c.rep[0] = rep_eval(Rf, 0, a0, a1, a2); c.rep[1] = rep_eval(Rf, 1, a0, a1, a2); c.rep[2] = rep_eval(Rf, 2, a0, a1, a2);
                    }
                }
                WG_SNAPSHOT(err = c.err);
                STAMP(6);
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
                if (wave == 2) {
                    const uint32_t stored = ld32(src + pos_now); // (issued before the digest is closed: its round trip overlaps)
                    xxh_advance(xv, xstripes, (out_now - fout0) / 32, dst + fout0, lane);
                    uint64_t h = xxh_finish(xv, dst + fout0, out_now - fout0, lane);
                    if (lane == 0) {
#ifndef MZD_EXP_NOHASH
                        if ((uint32_t)h != stored) c.err = MZD_E_CHECKSUM;
#endif
                        c.pos = pos_now + 4;
                    }
                    STAMP(7);
                }
                WG_SNAPSHOT(err = c.err);
                if (err) break;
            }
        }
        if (dst2) { // what the mirror has not seen yet: raw / RLE blocks, the last bytes of the file
            uint64_t out_final = 0;
            WG_SNAPSHOT(err = c.err; out_final = c.out);
            if (wave == 2 && !err && out_final > mirrored) mirror_wave(dst, dst2, mirrored, out_final, lane);
        }
        if (tid == 0) { a.jobs[j].out_len = c.out; a.jobs[j].status = c.err; }
        STAMP_FLUSH();
        TTASK_END();
        TFIN_FLUSH();
        grp_sync();
    }
    clean_next_counters(a, tid);
}

#if !MZD_PAIRS && !MZD_W3
// ---- driver 2: block tasks (the hand-over helpers are above, in front of the shared block pipeline)
#ifdef MZD_EXP_DEVSITE // (experiment: the first segment of a task that took more than 50 ms, with its job and task)
#define DEVSLOW_DECL uint64_t ds_t0_ = wall_clock64()
#define DEVSLOW(k) do { const uint64_t n_ = wall_clock64(); if (tid == 0 && n_ - ds_t0_ > 5000000ull) DEVSITE(((200u + (k)) << 16) | ((j & 0xFF) << 8) | (t & 0xFF)); ds_t0_ = n_; } while (0)
#else
#define DEVSLOW_DECL ((void)0)
#define DEVSLOW(k) ((void)0)
#endif
__global__ __launch_bounds__(kWG, MZD_LB_WAVES) void mzd_decode_kernel_tasks(KernelArgs) {
    const KernelArgs& a = launch_args();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t slot = a.wg0 + vblock(); // this workgroup's place in the scratch arrays
    uint8_t* const lit_buf = a.lit_scratch + (size_t)slot * kLitStride;
    uint4* const seqs = a.seq_scratch + (size_t)slot * kSeqStride;
    uint4* const walk = a.walk_scratch + (size_t)slot * kSeqStride;
    Ctl& c = S.c;
    if (tid < 36) S.ll_base[tid] = LL_BASE[tid];
    if (tid < 53) S.ml_base[tid] = ML_BASE[tid];
    if (tid == 0) S.walk_dummy = lds_base() + kLdsWalkDummy; // {own address, no bits}: what the walker's fourth lane follows
    if (tid == 0) S.took_first = 0;

    for (;;) {
        // ---------------- take a task: tickets below njobs are the first blocks of the files, the others the pushed
        // continuations in push order (a ticket may have to wait for its record; it gives up once every file is finished)
        if (tid == 0) {
            c.t_valid = 0;
            const uint32_t ticket = take_ticket(a);
            const uint32_t nq = queue_len(a);
            if (ticket < nq && queue_job(a, ticket) >= a.njobs) atomicAdd(&a.counter[3], 1u); // (a list entry that names no job of the launch: counted as finished, never decoded)
            else if (ticket < nq) { c.job = queue_job(a, ticket); c.task = 0; c.pos = 0; c.in_frame = 0; c.with_dict = 0; c.t_valid = 1; }
            else {
                const uint32_t m = ticket - nq;
                const ContRecord* r = &a.ring[m % a.ring_cap];
                const uint64_t want = ((uint64_t)a.epoch << 32) | (uint64_t)(m + 1);
                for (uint32_t it = 0; it < (1u << 23); it++) {
                    bool got = __hip_atomic_load(&r->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
                    if (!got && g_load(&a.counter[3]) >= nq) { // every file is finished: nothing is pushed any more
                        got = __hip_atomic_load(&r->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
                        if (!got) break;
                    }
#ifdef MZD_EXP_DEVSITE
                    if (it == (1u << 23) - 1) DEVSITE((100u << 16) | (m & 0xFFFF));
#endif
                    if (got) {
                        c.job = g_ld(&r->job); c.task = g_ld(&r->task); c.pos = g_ld(&r->pos); c.in_frame = g_ld(&r->in_frame); c.with_dict = g_ld(&r->with_dict);
                        c.has_fcs = g_ld(&r->has_fcs); c.has_cksum = g_ld(&r->has_cksum); c.block_max = g_ld(&r->block_max); c.fcs = g_ld(&r->fcs);
                        c.t_valid = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
        }
        uint32_t t_valid = 0, j = 0, t = 0, in_frame = 0;
        WG_SNAPSHOT(t_valid = c.t_valid; j = c.job; t = c.task; in_frame = c.in_frame);
        if (!t_valid) break;
        TTASK();
        DEVSLOW_DECL;
        const uint8_t* const src = a.jobs[j].src;
        const uint64_t n = a.jobs[j].src_len;
        uint8_t* const dst = a.jobs[j].dst;
        const uint64_t cap = a.jobs[j].dst_cap;
        const uint32_t job_dict = a.jobs[j].dict;
        uint8_t* const dst2 = a.jobs[j].dst2; // (host mirror of the output, or null)
        uint64_t mirrored = 0;                // wave 2: how far this task's block has been mirrored
        bool have_mirrored = false;           // ... `mirrored` is valid (a compressed block whose copier ran)
        bool rep_hopped = false;              // resolving launches: this task has passed the repeat-offset chain on
        bool early_done = false;              // ... and has published both of its hand-overs already (bytes early, checksum state behind)
        FileState* const fs = &a.fstate[j];
        TableArea* const ta = &a.tables[j];
        if (tid == 0) {
            c.out = 0; c.err = 0; c.action = 0; c.diag_slow = 0; c.pred_ready = 0; c.tables_published = 0; c.last = 0;
#ifdef MZD_STAMPS
            for (int k_ = 0; k_ < 8; k_++) S.cdiag[k_] = 0;
#endif
            if (job_dict > a.ndicts) c.err = MZD_E_DICT;
            if (in_frame) { // the frame's context travels with the task; its dictionary content is looked up again
                c.dict_content = nullptr; c.dict_content_len = 0;
                if (c.with_dict && job_dict >= 1 && job_dict <= a.ndicts) { c.dict_content = a.dicts[job_dict - 1].content; c.dict_content_len = a.dicts[job_dict - 1].content_len; }
                c.huf_valid = 1; c.fse_valid = 1; // provisional: what is inherited is checked when it is fetched
            }
        }
        int err = 0;
        uint32_t action = 0;
        uint64_t xv = 0, xstripes = 0; // K7 state of the hashing wavefront (wave 2)
        STAMP_DECL;

        // ---------------- frame header (K0): a task that does not continue a frame starts at one (or at the file's end)
        bool frame_first = false;
        if (!in_frame) {
            for (;;) {
                if (tid == 0 && !c.err) parse_frame_or_skip(c, src, n, a.dicts, a.ndicts, job_dict);
                WG_SNAPSHOT(err = c.err; action = c.action);
                if (err || action != 1) break; // 1: a skippable frame was skipped, look again
            }
            frame_first = !err && action != 2;
            if (frame_first && action == 3) { // dictionary: entropy tables, repeat offsets and content
                const DevDict* dd = &a.dicts[job_dict - 1];
                if (dd->formatted) {
                    for (int i = tid; i < 512; i += kWG) { S.ll[i] = dd->ll[i]; S.ml[i] = dd->ml[i]; }
                    for (int i = tid; i < 256; i += kWG) S.of[i] = dd->of[i];
                    for (int i = tid; i < (int)kHufEntries; i += kWG) S.huf[i] = dd->huf[i];
                    if (tid == 0) {
                        c.al[0] = dd->al[0]; c.al[1] = dd->al[1]; c.al[2] = dd->al[2];
                        c.huf_log = dd->huf_log; c.huf_valid = 1; c.fse_valid = 1;
                        c.rep[0] = dd->rep[0]; c.rep[1] = dd->rep[1]; c.rep[2] = dd->rep[2];
                    }
                }
                if (tid == 0) { c.dict_content = dd->content; c.dict_content_len = dd->content_len; c.with_dict = 1; }
            }
        }
        const bool hashing = c.has_cksum != 0; // (garbage without a frame; unused then)

        // ---------------- block header, and the successor is pushed before anything is decoded
        bool have_block = !err && (in_frame || frame_first);
        uint32_t btype = 0, bsize = 0, last = 0;
        uint64_t pos0 = 0;
        bool pred_done = false; // the predecessor task had finished when this one read its block header
        if (have_block) {
            if (tid == 0) { parse_block_header(c, src, n); S.res[3] = (t != 0 && g_load(&fs->copied) >= t) ? 1u : 0u; } // (is the predecessor done already?)
            WG_SNAPSHOT(err = c.err; btype = c.btype; bsize = c.bsize; last = c.last; pos0 = c.pos; pred_done = S.res[3] != 0);
            if (err) have_block = false;
        }
        bool is_final = true; // no successor: this task closes the file
        if (have_block) {
            const uint64_t body_end = pos0 + (btype == 1 ? 1u : bsize);
            uint64_t next_pos = body_end;
            bool push = !last;
            if (last) { // the next frame, if any, starts behind the optional checksum
                const bool ck_ok = !hashing || n - body_end >= 4; // a truncated checksum is reported by this task
                next_pos = body_end + (hashing ? 4 : 0);
                push = ck_ok && next_pos < n;
            }
            if (push) {
                is_final = false;
                if (tid == 0) {
                    const uint32_t m = atomicAdd(&a.counter[2], 1u);
                    ContRecord* r = &a.ring[m % a.ring_cap];
                    g_st(&r->job, j); g_st(&r->task, t + 1); g_st(&r->pos, next_pos); g_st(&r->in_frame, last ? 0u : 1u); g_st(&r->with_dict, c.with_dict);
                    g_st(&r->has_fcs, c.has_fcs); g_st(&r->has_cksum, c.has_cksum); g_st(&r->block_max, c.block_max); g_st(&r->fcs, c.fcs);
                    g_settle();
                    g_release(); // (see g_publish)
                    __hip_atomic_store(&r->seq, ((uint64_t)a.epoch << 32) | (uint64_t)(m + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }

        DEVSLOW(1);
        if (a.resolve && !(have_block && btype == 2)) { // resolving launches: a task that plans nothing hands the repeat offsets on at once
            if (tid == 0) rep_hop(fs, t, frame_first, false, S.res_rep);
            rep_hopped = true;
        }

        // ---------------- the block
        uint64_t out0 = 0, frame_start = 0, out_end = 0;
        bool pred_loaded = false; // (workgroup-uniform) c.pred_* is filled in
        if (have_block && btype < 2) { // K6 raw / RLE: nothing to decode ahead; wait for the predecessor, then copy / fill
            if (tid == 0) { load_pred(fs, t); }
            int perr = 0;
            WG_SNAPSHOT(perr = c.pred_err; out0 = c.pred_out; frame_start = frame_first ? c.pred_out : c.pred_frame_out0);
            if (t && wave == 2) g_acquire(); // the hash reads what the predecessor wrote
            pred_loaded = true;
            out_end = out0;
            if (!perr) {
                if (bsize > cap - out0) { if (tid == 0) c.err = MZD_E_DSTSIZE; }
                else {
                    if (btype == 0) wg_copy(dst + out0, src + pos0, bsize, tid);
                    else wg_fill(dst + out0, src[pos0], bsize, tid);
                    out_end = out0 + bsize;
                }
            }
            if (tid == 0) { c.out = out_end; c.pos = pos0 + (btype == 1 ? 1u : bsize); }
            __syncthreads();
            if (wave == 2 && hashing && !perr) { // this block's stripes (K7 state travels from task to task)
                xv = frame_first ? xxh_init(lane) : c.pred_xxh[lane & 3];
                xstripes = frame_first ? 0 : c.pred_xstripes;
                xxh_advance(xv, xstripes, (out_end - frame_start) / 32, dst + frame_start, lane);
            }
        } else if (have_block) {
            // (launch-wide: every task then keeps the repeat-offset chain going.  A file's first block has no predecessor to
            //  wait for: its copier and hasher follow its walker as they always do, and the file's checksum chain starts early)
            // resolve == 2 (a launch with many tasks per workgroup slot): only a task whose predecessor is NOT done yet resolves ahead --
            // one that can copy at once is better off with the streaming copier, which runs beside the walk.
            // resolve == 3 (what lies between the two): every other task of a file -- the odd ones -- resolves ahead: the chain of in-order copies
            // is half as long, and so is the work the byte maps cost.
            const bool resolving = a.resolve != 0 && t != 0 && !(a.resolve == 2 && pred_done) && !(a.resolve == 3 && !(t & 1));
            if (tid == 0) { S.ba = BlockArgs{src, n, dst, cap, dst2, src + pos0, bsize, pos0, 0, lit_buf, seqs, walk, last, hashing, false, t, frame_first, is_final, fs, ta, j, &a}; }
            __syncthreads();
            const BlockArgs& ba = S.ba; // (in LDS: see driver 1)
            const bool started = compressed_block<true>(a, ba, xv, xstripes, mirrored, tid, lane, wave, resolving);
            bool resolved = false;
            __syncthreads();
            DEVSLOW(2);
            if (resolving && started) {
                __syncthreads(); // walk, plan and literals are complete (every role has returned); nothing of the block has been written yet
                TFIN(7); // (diagnostic builds: the resolve timeline reuses the literal-side slots 7, 8, 4 and the copier's 9, 1, 2)
                // (the planning wavefront has passed the repeat-offset chain on: S.res_rep, S.res[3]; the copying and the hashing wavefront
                //  have built the map behind the planner, but for the chunks with symbolic offsets: S.res_sym)
                uint32_t nseq = 0, nlit = 0, pout = 0, lused = 0, too_long = 0, lit_type = 0, r0 = 0, r1 = 0, r2 = 0, hop_ok = 0, inexact = 0;
                uint64_t lit_off = 0;
                WG_SNAPSHOT(err = c.err; nseq = c.nseq; nlit = c.nlit; pout = c.plan_out; lused = c.plan_lit_used; too_long = c.plan_too_long; lit_type = c.lit_type;
                            lit_off = c.lit_off; r0 = S.res_rep[0]; r1 = S.res_rep[1]; r2 = S.res_rep[2]; hop_ok = S.res[3]; inexact = c.walk_inexact);
                uint32_t* const map = a.resolve_map + (size_t)slot * kResMapStride;
                const uint32_t B = pout + (nlit - lused);
                bool ok = !err && !inexact && hop_ok && nseq != 0 && !too_long && lused <= nlit && B <= kBlockMax; // (anything wrong: the copying wavefront gives the verdict, in the reference's order)
                if (ok) {
                    resolve_build_rest(map, seqs, walk, nseq, pout, lused, nlit, r0, r1, r2, a.debug ? seqs : nullptr, lane, wave);
                    wg_fence();
                    uint32_t bad = 0;
                    WG_SNAPSHOT(bad = S.res[0]);
                    TFIN(8);
                    ok = !bad && resolve_jump_tiled(map, B, tid);
                    TFIN(4);
                }
                if (ok) { // in task order from here
                    // A block of a frame with a checksum, not the file's last task, hands its BYTES over as soon as they are gathered
                    // and its checksum state when the hash is done (FileState::hashed): the successor's gather does not wait for it.
                    uint32_t tpub = 0;
                    WG_SNAPSHOT(tpub = c.tables_published);
                    const bool early = hashing && !is_final && tpub != 0 && a.resolve == 1; // (launches with many tasks per slot are not chain-bound)
                    if (tid == 0) { if (early) load_pred_copy(fs, t); else load_pred(fs, t); }
                    int perr = 0;
                    uint64_t fstart0 = 0;
                    uint32_t reach = 0;
                    WG_SNAPSHOT(perr = c.pred_err; out0 = c.pred_out; fstart0 = frame_first ? c.pred_out : c.pred_frame_out0; reach = S.res[1]);
                    pred_loaded = true;
                    TFIN(9);
                    if (!perr && B <= cap - out0 && reach <= out0 - fstart0) {
                        // the predecessors' output (another XCD's L2 may have held it).  (Agent-scope byte loads instead of the fence --
                        // to keep the map and the literals in this XCD's L2 -- were measured: 2.6x slower, every byte a memory request.)
                        if (t) g_acquire();
                        const uint8_t* const lits = lit_type == 0 ? src + lit_off : lit_buf;
                        if (hashing) { // K7 beside the gather: wavefront 2 hashes behind the other three (the state travels from task to task)
                            if (tid < 3) S.res_prog[tid] = 0;
                            __syncthreads();
                            const uint64_t out_end = out0 + B;
                            // (the frame's trailer checks that need no hash: content size, room for the checksum)
                            const int err_c = !last ? 0 : ((c.has_fcs && out_end - fstart0 != c.fcs) ? MZD_E_CORRUPT : (n - (pos0 + bsize) < 4 ? MZD_E_TRUNCATED : 0));
                            if (wave == 2) {
                                bool chain_ok = true;
                                if (early) { // the checksum chain has its own hand-over
                                    int ok_ = 1;
                                    if (lane == 0) ok_ = load_pred_hash(fs, t) ? 1 : 0;
                                    chain_ok = __builtin_amdgcn_readfirstlane(ok_) != 0;
                                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                }
                                const bool dead = early && (!chain_ok || __atomic_load_n(&c.pred_err, __ATOMIC_RELAXED) != 0); // a checksum failed earlier in the file
                                xv = frame_first ? xxh_init(lane) : c.pred_xxh[lane & 3];
                                xstripes = frame_first ? 0 : c.pred_xstripes;
                                if (!dead && !resolve_hash_behind(xv, xstripes, dst + fstart0, out0 - fstart0, B, lane) && lane == 0) { DEVSITE(11); post_err(&c.err, MZD_E_DEVICE); }
                                if (early) {
                                    int herr_now = 0;
                                    if (!dead && last && !err_c) { // close the digest
                                        xxh_advance(xv, xstripes, (out_end - fstart0) / 32, dst + fstart0, lane);
                                        const uint64_t h = xxh_finish(xv, dst + fstart0, out_end - fstart0, lane);
#ifndef MZD_EXP_NOHASH
                                        if ((uint32_t)h != ld32(src + pos0 + bsize)) herr_now = MZD_E_CHECKSUM;
#endif
                                    }
                                    if (lane < 4) g_st(&fs->xxh[lane], xv);
                                    if (lane == 0) {
                                        g_st(&fs->xstripes, xstripes);
                                        if (herr_now) { g_st(&fs->herr_out, out0); g_st(&fs->herr, (int32_t)herr_now); }
                                    }
                                    g_settle();
                                    if (lane == 0) g_publish(&fs->hashed, t + 1);
                                }
                                TFIN(2);
                            } else {
                                resolve_gather3(map, B, lits, dst, out0, wave == 3 ? 2 : wave, lane);
                                if (wave == 0) TFIN(1);
                                if (early && wave == 0) { // the bytes' hand-over: when the other two gathering wavefronts are through as well
                                    const uint32_t nsteps = ((B + 3) / 4 + kResStepDw - 1) / kResStepDw;
                                    if (lane == 0) {
                                        for (uint32_t it = 0; it < (1u << 24); it++) {
                                            if (flag_load(&S.res_prog[1]) >= nsteps && flag_load(&S.res_prog[2]) >= nsteps) break;
                                            __builtin_amdgcn_s_sleep(4);
                                            if (it == (1u << 24) - 1) DEVSITE(14);
                                        }
                                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                        RepOp Rf; Rf.s = c.rep_op[0]; Rf.v0 = (int32_t)c.rep_op[1]; Rf.v1 = (int32_t)c.rep_op[2]; Rf.v2 = (int32_t)c.rep_op[3];
                                        const uint32_t ri0 = frame_first ? c.rep[0] : S.res_rep[0], ri1 = frame_first ? c.rep[1] : S.res_rep[1], ri2 = frame_first ? c.rep[2] : S.res_rep[2];
                                        g_st(&fs->rep[0], rep_eval(Rf, 0, ri0, ri1, ri2)); g_st(&fs->rep[1], rep_eval(Rf, 1, ri0, ri1, ri2)); g_st(&fs->rep[2], rep_eval(Rf, 2, ri0, ri1, ri2));
                                        g_st(&fs->err, (int32_t)err_c);
                                        g_st(&fs->out, err_c ? out0 : out_end);
                                        g_st(&fs->frame_out0, fstart0);
                                        g_settle();
                                        g_release(); // (every gathering wavefront has waited for its stores: resolve_gather3)
                                        g_store(&fs->copied, t + 1);
                                    }
                                }
                            }
                            if (early) { early_done = true; if (tid == 0 && err_c) c.err = err_c; }
                        } else {
                            resolve_gather(map, B, lits, dst, out0, tid);
                            wg_fence();
                            TFIN(1);
                        }
                        __syncthreads();
                        if (tid == 0) { c.out = out0 + B; c.pos = pos0 + bsize; }
                        if (a.debug && tid == 0) {
                            DebugSlot& ds = a.debug[a.wg0 + vblock()];
                            ds.n_lit = nlit; ds.n_seq = nseq; ds.lit_is_raw = lit_type == 0; ds.lit_raw_ptr = (uint64_t)(uintptr_t)(lit_type == 0 ? src + lit_off : lit_buf);
                            if (j == 0) atomicMax(&a.counter[1], (t << 12) | (a.wg0 + blockIdx.x));
                        }
                        resolved = true;
                    } else if (perr) { // the file has already failed: nothing to execute (the error travels on)
                        if (early && tid == 0) load_pred_hash(fs, t); // (this task publishes both chains together, below: the checksum chain's turn must have come)
                        if (tid == 0) { c.out = out0; c.pos = pos0 + bsize; }
                        resolved = true;
                    } else if (early) { // (falls back to the copying wavefront, which starts from the complete predecessor state)
                        if (tid == 0) load_pred_hash(fs, t);
                        __syncthreads();
                    }
                }
            }
            if (started) rep_hopped = true; // (role_plan)
            DEVSLOW(3);
            if (resolving && started && !resolved) { compressed_block_copy(a, ba, xv, xstripes, mirrored, tid, lane, wave); have_mirrored = true; __syncthreads(); DEVSLOW(4); }
            else if (!resolving) have_mirrored = true;
            WG_SNAPSHOT(err = c.err);
            STAMP(6);
            if (!resolved) pred_loaded = c.pred_ready != 0;
        }
        if (a.resolve && !rep_hopped) { // (a compressed block whose headers failed)
            if (tid == 0) rep_hop(fs, t, frame_first, false, S.res_rep);
            rep_hopped = true;
        }

        // ---------------- completion, in task order: frame trailer (K7), then the state for the successor
        __syncthreads();
        DEVSLOW(5);
        if (tid == 0 && !pred_loaded) load_pred(fs, t);
        int perr = 0;
        uint64_t pred_out = 0, fstart = 0, out_now = 0, pos_now = 0;
        uint32_t has_ck = 0;
        WG_SNAPSHOT(perr = c.pred_err; pred_out = c.pred_out; fstart = frame_first ? c.pred_out : c.pred_frame_out0; err = c.err; out_now = c.out; pos_now = c.pos; has_ck = c.has_cksum; last = c.last);
        if (!have_block) out_now = pred_out; // a task without a block (end of file, or a header error) produces nothing
        int final_err = perr ? perr : err;
        const bool mirror_rest = dst2 && !final_err && out_now > pred_out; // this task's bytes the host mirror has not seen yet (all of a raw / RLE block)
        const uint64_t mirror_from = (have_mirrored && mirrored >= pred_out && mirrored <= out_now) ? mirrored : pred_out;
        if (!final_err && have_block && last && !early_done) { // frame trailer: content size and checksum
            if (tid == 0) {
                if (c.has_fcs && out_now - fstart != c.fcs) c.err = MZD_E_CORRUPT;
                else if (has_ck && n - pos_now < 4) c.err = MZD_E_TRUNCATED;
            }
            WG_SNAPSHOT(err = c.err);
            if (!err && has_ck) {
                if (wave == 2) {
                    xxh_advance(xv, xstripes, (out_now - fstart) / 32, dst + fstart, lane);
                    uint64_t h = xxh_finish(xv, dst + fstart, out_now - fstart, lane);
                    if (lane == 0) {
#ifndef MZD_EXP_NOHASH
                        if ((uint32_t)h != ld32(src + pos_now)) c.err = MZD_E_CHECKSUM;
#endif
                    }
                    STAMP(7);
                }
                WG_SNAPSHOT(err = c.err);
            }
            final_err = err;
        }
        // the state for the successor (or the file's result), published in task order
        if (early_done) {} // (done: mzd_k_resolve.h path above)
        else if (!is_final) {
            if (wave == 2) { // K7 state lives in this wavefront's registers
                if (lane < 4) g_st(&fs->xxh[lane], xv);
                if (lane == 0) g_st(&fs->xstripes, xstripes);
            }
            if (!c.tables_published) { // raw/RLE block, early error: the tables are unchanged, the version still moves on
                if (tid == 0) c.t_valid = g_wait_ge(&fs->tables_ver, t, 2) ? 1u : 0u;
                uint32_t ver_ok = 0, hv = 0, fv = 0;
                WG_SNAPSHOT(ver_ok = c.t_valid; hv = c.huf_valid; fv = c.fse_valid);
                if (ver_ok) {
                    if (frame_first && !final_err) { // the frame starts here: what its successors inherit is the dictionary's tables, or nothing
                        if (fv) {
                            for (int i = tid; i < 512; i += kWG) { g_st(&ta->ll[i], S.ll[i]); g_st(&ta->ml[i], S.ml[i]); }
                            for (int i = tid; i < 256; i += kWG) g_st(&ta->of[i], S.of[i]);
                        }
                        if (hv) for (int i = tid; i < (int)kHufWords; i += kWG) g_st(&reinterpret_cast<uint32_t*>(ta->huf)[i], reinterpret_cast<const uint32_t*>(S.huf)[i]);
                        if (tid == 0) {
                            g_st(&fs->fse_valid, fv); g_st(&fs->huf_valid, hv); g_st(&fs->huf_log, c.huf_log);
                            g_st(&fs->al[0], c.al[0]); g_st(&fs->al[1], c.al[1]); g_st(&fs->al[2], c.al[2]);
                        }
                    }
                    g_settle();
                    __syncthreads();
                    if (tid == 0) g_publish(&fs->tables_ver, t + 1);
                }
            }
            if (tid == 0) {
                const uint32_t ri0 = frame_first ? c.rep[0] : c.pred_rep[0], ri1 = frame_first ? c.rep[1] : c.pred_rep[1], ri2 = frame_first ? c.rep[2] : c.pred_rep[2];
                RepOp Rf;
                if (have_block && btype == 2) { Rf.s = c.rep_op[0]; Rf.v0 = (int32_t)c.rep_op[1]; Rf.v1 = (int32_t)c.rep_op[2]; Rf.v2 = (int32_t)c.rep_op[3]; }
                else { Rf.s = 0 | (1 << 2) | (2 << 4); Rf.v0 = 0; Rf.v1 = 0; Rf.v2 = 0; }
                g_st(&fs->rep[0], rep_eval(Rf, 0, ri0, ri1, ri2)); g_st(&fs->rep[1], rep_eval(Rf, 1, ri0, ri1, ri2)); g_st(&fs->rep[2], rep_eval(Rf, 2, ri0, ri1, ri2));
                g_st(&fs->err, (int32_t)final_err);
                g_st(&fs->out, final_err ? pred_out : out_now);
                g_st(&fs->frame_out0, fstart);
            }
            // this task's output bytes must be in memory before the successor (possibly on another XCD) is let go:
            // every wavefront that stored output has waited for its stores (its own fences); one agent-scope release
            // writes the XCD's L2 back
            g_settle();
            __syncthreads();
            if (tid == 0) { g_release(); g_store(&fs->hashed, t + 1); g_store(&fs->copied, t + 1); }
        } else if (tid == 0) { // the file is finished: its result, and one file less to wait for
            a.jobs[j].out_len = final_err ? pred_out : out_now;
            a.jobs[j].status = final_err;
            atomicAdd(&a.counter[3], 1u);
        }
        // the host mirror of this task's block: behind the hand-over -- the successor copies on, this wavefront feeds PCIe
        if (mirror_rest && wave == 2 && out_now > mirror_from) mirror_wave(dst, dst2, mirror_from, out_now, lane);
        DEVSLOW(6);
        TTASK_END();
        STAMP_FLUSH();
        TFIN_FLUSH();
        __syncthreads();
    }
    clean_next_counters(a, tid);
}

// Dictionary (A.7) -> DevDict: the entropy tables in the exact LDS layout, built once on the
// device with the same routines the decoder uses.  One workgroup.
__global__ __launch_bounds__(kWG) void mzd_dict_kernel(const uint8_t* dict, uint32_t n, DevDict* out, int32_t* status) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Ctl& c = S.c;
    uint32_t& pos_after_huf = S.res[0]; uint32_t& pos_after_tables = S.res[1]; // (no static LDS object: the image starts at LDS address 0, and the entries built here name its addresses)
    if (tid == 0) {
        c.err = 0; c.action = 0;
        if (n < 8 || ld32(dict) != 0xEC30A437u) c.action = 1; // raw content
        else {
            int used = read_huf_weights(dict + 8, n - 8);
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
    if (wave == 0) { int hr = finish_huf_table_wave(lane); if (hr && lane == 0) c.err = MZD_E_DICT; }
    __syncthreads();
    if (c.err) { if (tid == 0) *status = c.err; return; }
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
    if (wave == 0) build_tables_wave(lane);
    __syncthreads();
    for (int i = tid; i < 512; i += kWG) { out->ll[i] = S.ll[i]; out->ml[i] = S.ml[i]; }
    for (int i = tid; i < 256; i += kWG) out->of[i] = S.of[i];
    for (int i = tid; i < (int)kHufEntries; i += kWG) out->huf[i] = S.huf[i];
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

void* decode_kernel_ptr(int tasks) { return tasks ? (void*)mzd_decode_kernel_tasks : (void*)mzd_decode_kernel_files; }

#ifdef MZD_EXP_DEVSITE
void devsite_take(uint32_t* out3) { uint32_t z[4] = {0, 0, 0, 0}; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out3, HIP_SYMBOL(g_devsite), 12); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_devsite), z, 16); }
#endif
// grid: GROUPS (mzd_k_common.h); groups_per_wg: 1, or 2 for driver 1 (two files a workgroup, one walking wavefront for both)
#endif // !MZD_PAIRS && !MZD_W3

#ifdef MZD_EXP_PLANDIAG
#if MZD_PAIRS
#define MZD_PLANDIAG_TAKE plandiag_take_pairs
#elif MZD_W3
#define MZD_PLANDIAG_TAKE plandiag_take_w3
#else
#define MZD_PLANDIAG_TAKE plandiag_take
#endif
void MZD_PLANDIAG_TAKE(uint32_t* out16) { uint32_t z[16] = {0}; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_plandiag), 64); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_plandiag), z, 64); }
#endif
#if MZD_W3
void launch_decode_w3(const KernelArgs& a, uint32_t grid, void* stream) {
    hipLaunchKernelGGL(mzd_decode_kernel_files3, dim3(grid), dim3(kWG), 0, (hipStream_t)stream, a);
}
int w3_workgroups_per_cu() {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)mzd_decode_kernel_files3, kWG, 0) != hipSuccess) return 0;
    return per_cu > 5 ? 5 : per_cu; // (LDS is allocated in 1 280-byte steps: five images of 32 000 bytes fit a CU, whatever the API says of six)
}
#elif MZD_PAIRS
// grid: GROUPS of four wavefronts (an even number: the host rounds); two a workgroup
void launch_decode_pairs(const KernelArgs& a, uint32_t grid, void* stream) {
    hipLaunchKernelGGL(mzd_decode_kernel_pairs, dim3((grid + 1) / 2), dim3(2 * kWG), 2 * sizeof(Shared), (hipStream_t)stream, a);
}
// once per device: two images are more dynamic LDS than a kernel may ask for by default
int pairs_prepare_device() {
    return hipFuncSetAttribute((const void*)mzd_decode_kernel_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGroupsMax * sizeof(Shared))) == hipSuccess ? 0 : MZD_E_DEVICE;
}
#else
void launch_decode(const KernelArgs& a, uint32_t grid, void* stream) {
    if (a.use_tasks) hipLaunchKernelGGL(mzd_decode_kernel_tasks, dim3(grid), dim3(kWG), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mzd_decode_kernel_files, dim3(grid), dim3(kWG), 0, (hipStream_t)stream, a);
}
int kernel_lds_bytes() { return (int)sizeof(Shared); }
#endif

} // namespace mzd

