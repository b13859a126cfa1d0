// mzd_k_common.h -- part of the block pipeline of mzd_kernels.hip (see the map at the top of that file).  Included there, inside
// namespace mzd, in dependency order; not a translation unit of its own.
#pragma once
// ------------------------------------------------------------------------------------ LDS
constexpr int kRingBytes = 8192; // sequence-bitstream ring: 8 chunks of 1 KiB (+16 mirrored bytes)
constexpr int kChunk = 1024;
constexpr int kRingChunks = kRingBytes / kChunk;

struct Ctl {
    uint64_t pos;        // next unread input byte of the file
    uint64_t out;        // bytes produced for this file
    uint64_t frame_out0; // `out` at the start of the current frame
    uint64_t fcs;
    uint64_t lit_off;    // file offset of the raw literals / first Huffman stream
    uint64_t seq_off;    // file offset of the sequence bitstream
    int32_t err;
    uint32_t action;     // 0 frame, 1 skip, 2 done
    uint32_t job;
    uint32_t has_fcs, has_cksum, block_max;
    uint32_t btype, bsize, last;
    uint32_t lit_type, nlit, streams, huf_log, huf_valid, huf_nw;
    uint32_t s_off[4], s_len[4], s_out[4], s_n[4];
    uint32_t lit_is_raw;
    uint32_t huf_tree_off, huf_tree_len;           // Huffman tree description inside the block
    uint32_t huf_ready, huf_fill, lit_done, walk_prog; // intra-workgroup flags of the block pipeline
    uint32_t walk_inexact; // the walk ended without having consumed the sequence bitstream exactly: an error, but one that is found BEHIND the
                           // block's sequences -- the reference executes sequence after sequence and checks the stream's end last, so an
                           // execution error (destination too small, literals, offset) of any sequence is reported first (libzstd >= 1.5.4)
    uint32_t next_stream, streams_done, streams_mask; // Huffman streams are handed out to whichever wavefront is free; mask: bit k = stream k decoded
    uint32_t tables_ready, plan_prog, copy_prog, plan_lit_used; // walker -> planner -> copier
    uint32_t walk_g0;                               // (read head - 32) of the block's first walk record: records carry 16 bits of it
    uint32_t plan_out;      // output bytes of the block's sequences (the literals behind the last one not counted), once the plan is complete
    uint32_t plan_too_long; // 1: the plan ends with a chunk that cannot be executed (literals run out / output passes 128 KiB); 2: only the literals after
                            // the last sequence pass 128 KiB.  No error yet: the copier, which reports in stream order, gives the verdict
    uint32_t seq_parsed;                           // the sequence header is parsed: nseq, seq_off, seq_len, modes are final
    uint32_t exec_done;                            // the copying wavefront has finished the block
    uint64_t exec_pos;                             // output bytes complete and visible (published by the executor)
    uint32_t diag_slow;                            // diagnostic build: walker iterations that needed a lower window
    uint32_t nseq, mode[3], al[3], nsym[3], fse_valid, seq_len;
    uint32_t rep[3];
    uint32_t rep_op[4];                             // the block's repeat-offset transform (start slots -> end slots): s, v0, v1, v2
    uint32_t dict_content_len;
    const uint8_t* dict_content;
    // the task (one block of one file) and what its predecessor published
    uint32_t lds_dict_fse, lds_dict_huf;            // driver 1: dictionary (handle) whose FSE / Huffman tables sit unmodified in LDS, or 0
    uint32_t t_valid, task, in_frame, with_dict;
    uint32_t pred_ready;                            // the predecessor's state is in pred_* (LDS flag of the block pipeline)
    int32_t pred_err;
    uint32_t pred_rep[3];
    uint32_t tables_published;
    uint64_t pred_out, pred_frame_out0, pred_xstripes, pred_xxh[4];
};

// One block of one file, as the drivers hand it to the block pipeline (mzd_k_pipeline.h).  The roles take it by reference: the workgroup's
// copy is S.ba (on the stack it would be a copy per lane in the private segment).
struct BlockArgs {
    const uint8_t* src; uint64_t n;       // the file
    uint8_t* dst; uint64_t cap;
    uint8_t* dst2;                        // mirror of the output in pinned host memory, or null
    const uint8_t* blk; uint32_t bsize;   // the block's content
    uint64_t pos0;                        // its offset in the file
    uint64_t out0;                        // TASKS = false: output position at the block's start
    uint8_t* lit_buf; uint4* seqs; uint4* walk; // the workgroup's HBM scratch
    uint32_t last;
    bool hashing;
    bool block_pre;                       // TASKS = false: headers already parsed (pre_parse_next)
    // TASKS = true: the task
    uint32_t t; bool frame_first, is_final;
    FileState* fs; TableArea* ta;
    uint32_t job;
    const KernelArgs* args;               // the launch's arguments (the roles reach them through the block: launch_args)
};

#ifndef MZD_PAIRS
#define MZD_PAIRS 0
#endif
// MZD_W3 (round 6, the third compile of this source): driver 1 with workgroups of THREE wavefronts, five to a CU -- the walking, the copying
// and a FOLLOWING wavefront that plans behind the walker and hashes in the planner's waits (mzd_k_pipeline.h) -- where the launch fills
// the machine more than once: 15 wavefronts a CU keep the 128 registers the copier needs (20 would leave 96: profiles/r05_wg5_ab.txt), and
// the image goes on the diet below.
#ifndef MZD_W3
#define MZD_W3 0
#endif
#if MZD_W3 && !defined(MZD_WGS_PER_CU)
#define MZD_WGS_PER_CU 5
#endif
// Residency: MZD_WGS_PER_CU workgroups share a CU -- 512 / that many registers a lane, and an LDS image of at most 160 KiB / that
// many, in the 1 280-byte steps LDS is allocated in (tools/micro/lds_granule_micro.hip).  Five: 96 registers, 32 000 bytes.
#ifndef MZD_WGS_PER_CU
#define MZD_WGS_PER_CU 4
#endif
#if defined(MZD_TFIN) || defined(MZD_STAMPS)
constexpr uint32_t kStage = MZD_WGS_PER_CU >= 5 ? 1472 : 2048; // (the diagnostic builds' stamps take LDS too)
#else
constexpr uint32_t kStage = MZD_WGS_PER_CU >= 5 ? 1536 : 2048;
#endif
static_assert(true, ""); // K5: bytes of a staged run (mzd_k_execute.h)
constexpr uint32_t kSeg2Bytes = MZD_WGS_PER_CU >= 5 ? 1024 : 2048; // Huffman stream segment of wavefront 2 (the others' are 2 KiB: mzd_k_huffman.h)
constexpr uint32_t kResSymMax = 30;
// Two groups in a workgroup: the walking wavefronts meet before every run of their hot loop; group 1's posts what its run starts from and
// sleeps, group 0's runs both chains in one loop -- a quad of lanes each -- and writes back where group 1's ended (mzd_k_walk.h: walk_run).
struct WalkShare {
    uint32_t active;   // this group's walking wavefront is inside a walk (it will come to the meeting point again)
    uint32_t state;    // the request: 0 none, 1 posted, 3 taken by the partner, 2 results are in
    uint32_t A[3], Gm, woff, n, thresh, pv, prog_lds, yield;               // request: state addresses (LL, ML, OF), read head - 32, record offset, steps, lower bound, progress
    uint32_t rA[3], rGm, done, voided, sA[3], sG;                   // results: the same after the run, steps done, the last group void, the state at its start
};
struct __attribute__((aligned(16))) Shared {
    uint8_t ring[kRingBytes + 16]; // first: at LDS offset 0 the walker's window address needs no base add
    // FSE decode entries, 8 bytes: low dword = LDS address (offset into S) of the next state's entry before the
    // fresh bits are added (table + 8 * nextStateBase); high dword = nbBits | (extra+nbBits) << 8 | symbol << 16 | extra << 24
    uint64_t ll[512];
    uint64_t ml[512];
    uint64_t of[256];
    uint8_t stage[3 * (kStage + 16)]; // K5 staging: the run being assembled and the two before it (kStage each); >= 2064 + 2112: the copying wavefront's Huffman segment lies at 2064
    uint8_t hseg2[kSeg2Bytes + 64]; // Huffman stream segment of wavefront 2 (it still decodes while the copier already uses `stage`)
    uint32_t ll_base[36], ml_base[53]; // code -> base value (copied once from constant memory)
#ifdef MZD_STAMPS
    uint64_t cdiag[8];
#endif
#if defined(MZD_STAMPS) || defined(MZD_TFIN)
    uint64_t ttask, tstart, tabs, tfin[12]; // block start; finish of walker / copier / hasher / planner; literals ready; tables ready
#endif
#ifdef MZD_EXP_STREAMSTAMP
    uint64_t sst[8]; // (diagnostic: take time << 2 | wavefront of streams 0..3, then their finish times)
    uint64_t swv[16]; // (per wavefront: role entry, past the wait for the Huffman table, in front of the stream queue, first stream taken)
#endif
    uint16_t huf[kHufEntries]; // sym | len << 8 (mzd_device.h: kHufEntries)
    int16_t norm[3][64];
    uint16_t next[3][64];
    alignas(16) int16_t wnorm[256]; // FSE table of the Huffman weights.  wnorm + wtab + weights (1 KiB, contiguous) double as the
                                    // copier's literal scratch (kLitScratch): the copying wavefront is the one that decodes the weights, earlier
    uint32_t wtab[64];  // sym | nb << 8 | base << 16
    uint8_t weights[256];
    BlockArgs ba;        // the block being decoded (written by thread 0 before the pipeline starts)
    uint64_t walk_dummy; // {its own address, 0}: the entry the walker's fourth lane follows (mzd_k_walk.h); outside what mzd_k_resolve.h overlays
    Ctl c;
    // driver 1, files of one block: while the copier and the hasher finish file A, the idle walking wavefront takes the
    // next file and parses its headers into `c2` (header bytes staged in a free part of the ring)
    Ctl c2;
    uint32_t res[4];             // mzd_k_resolve.h: bad offset seen, farthest reach before the block, a round left entries open; [3] spare
    uint32_t res_rep[3];         // ... the repeat offsets the task starts with (rep_hop)
    uint32_t res_prog[3];        // ... steps the three gathering wavefronts have completed (resolve_gather3)
    uint32_t res_nsym, res_sym[kResSymMax]; // ... chunks left out by the build that follows the planner: they hold offsets still symbolic
    uint32_t took_first;         // this workgroup has used its first ticket (take_ticket)
    uint32_t and_word;           // mzd_k_resolve.h: a workgroup-wide AND
#if MZD_PAIRS
    uint32_t bar;                // grp_sync: arrivals at the group's barriers (workgroups of two groups: s_barrier would join both)
    WalkShare wk;                // the walking wavefronts' rendezvous (mzd_k_walk.h)
#endif
    uint32_t pre_job, pre_valid; // the job taken ahead (kNoJob: none) and whether c2 holds its parsed first block
    // driver 1: the small fields of the dictionary the workgroup used last (config 5: every file names the same one --
    // reading them from HBM again for each file costs a round trip per dependent load)
    struct { uint32_t id, formatted, al[3], huf_log, rep[3], content_len; const uint8_t* content; } dcache;
    struct { const uint8_t* src; uint64_t n; uint8_t* dst; uint64_t cap; uint32_t dict; } pj; // the job table entry of pre_job (read once, by pre_parse_next)
#ifdef MZD_EXP_PADLDS // (experiment: three workgroups per CU instead of four -- how throughput follows residency)
    uint8_t pad_[MZD_EXP_PADLDS];
#endif
};

// A GROUP is four wavefronts with an LDS image of their own: what decodes one file (driver 1) or one block (driver 2).  A hardware workgroup
// holds ONE group (256 threads: the block-task driver, single-round launches of driver 1) or TWO (512 threads, round 6: the two groups' walking
// wavefronts run their hot loop in ONE wavefront, a quad of lanes each -- mzd_k_walk.h: the chain's instructions are issued once for both files).
// The groups of a workgroup share nothing else: each has its queue tickets, its scratch slot, its barriers (grp_sync).
//
// The image lives in the dynamic LDS segment (the kernels have no static LDS object, so it starts at LDS address 0): group g's image is
// lds_img[g].  `S` is the calling wavefront's image; with two groups its address is not a compile-time constant (a base register per
// function: measured at 1.3-2 % of a launch, profiles/r06_rtbase_ab.txt).
#ifndef MZD_EXP_PADLDS
static_assert(sizeof(Shared) <= (128 / MZD_WGS_PER_CU) * 1280, "MZD_WGS_PER_CU groups per CU");
static_assert(sizeof(Shared) % 16 == 0, "images are 16-byte aligned");
static_assert(sizeof(((Shared*)nullptr)->stage) >= 2064 + 2048 + 64, "the copying wavefront's Huffman segment");
#endif
// The source is compiled TWICE (Makefile): MZD_PAIRS = 0, the kernels of one group a workgroup -- the image is a file-scope object at LDS address
// 0, every place in it a compile-time constant, nothing of the above costs anything --, and MZD_PAIRS = 1, driver 1 alone as
// mzd_decode_kernel_pairs with two groups a workgroup, the images in the dynamic LDS segment.
#if MZD_PAIRS
extern __shared__ __attribute__((aligned(16))) uint8_t lds_img[];
constexpr uint32_t kGroupsMax = 2;
__device__ __forceinline__ uint32_t grp_index() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)); }
__device__ __forceinline__ uint32_t grp_count() { return kGroupsMax; }
__device__ __forceinline__ Shared& S_of(uint32_t g) { return *reinterpret_cast<Shared*>(lds_img + g * (uint32_t)sizeof(Shared)); }
#define S (S_of(grp_index()))
#else
__shared__ Shared S_one;
constexpr uint32_t kGroupsMax = 1;
__device__ __forceinline__ uint32_t grp_index() { return 0u; }
__device__ __forceinline__ uint32_t grp_count() { return 1u; }
__device__ __forceinline__ Shared& S_of(uint32_t) { return S_one; }
#define S S_one
#endif
__device__ __forceinline__ uint32_t vblock() { return blockIdx.x * grp_count() + grp_index(); } // the group's index in the launch: its first ticket, its scratch slot
__device__ __forceinline__ uint32_t vgrid() { return gridDim.x * grp_count(); }
// places inside an image (add the image's LDS address: lds_base())
constexpr uint32_t kLdsLL = (uint32_t)offsetof(Shared, ll), kLdsML = (uint32_t)offsetof(Shared, ml), kLdsOF = (uint32_t)offsetof(Shared, of);
constexpr uint32_t kLdsWalkDummy = (uint32_t)offsetof(Shared, walk_dummy);
static_assert(kLdsLL == kBlkLdsLL && kLdsML == kBlkLdsML && kLdsOF == kBlkLdsOF, "mzd_device.h names the tables' places (dictionary images in HBM carry them)");
static_assert(kGroupsMax * sizeof(Shared) - sizeof(Shared) + kLdsOF + 2048 <= 65536, "walk records hold 16-bit state addresses");
typedef __attribute__((address_space(3))) uint8_t lds_byte;
__device__ __forceinline__ uint32_t lds_base() { return grp_index() * (uint32_t)sizeof(Shared); } // LDS address of the calling wavefront's image
// A pointer parameter that always names a place in LDS: says so, so that the function's accesses are DS instructions, not flat ones (with the
// image at a compile-time address the compiler saw that by itself -- every call passed the same constant)
#define MZD_IN_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)(p)))
// an FSE entry by its ABSOLUTE LDS address (what entries, walk records and the walker's registers hold)
__device__ __forceinline__ uint64_t lds_entry(uint32_t state_addr) { uint64_t v; __builtin_memcpy(&v, (const lds_byte*)(uintptr_t)state_addr, 8); return v; }

// The launch's arguments are read where the runtime put them (the kernel-argument segment: constant memory), never through the
// by-value parameter: the roles take them by reference, and a reference to the parameter makes the compiler keep a copy of it per
// LANE in the private segment -- 116 bytes x 256 lanes of stores per workgroup before anything else happens.
// (In the KERNEL function only: inside a called function the intrinsic did not give the kernel's segment here -- the roles reach the arguments
//  through BlockArgs::args.)
__device__ __forceinline__ const KernelArgs& launch_args() {
    return *reinterpret_cast<const KernelArgs*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
}
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld24(const uint8_t* p) { return ld16(p) | ((uint32_t)p[2] << 16); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return ld16(p) | (ld16(p + 2) << 16); }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { return (uint64_t)ld32(p) | ((uint64_t)ld32(p + 4) << 32); }
__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ldu64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ int hibit(uint32_t v) { return 31 - __builtin_clz(v); }

__device__ __forceinline__ void wg_fence() {
    // make this wave's global stores visible to later loads of the same workgroup (same CU, same L1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// bits [bitpos, bitpos+n) of the little-endian integer p[0..nbytes); indices < 0 and >= 8*nbytes
// read as 0.  n <= 32.  Lane-0 parsing helper (the input is readable MZD_SRC_PADDING past its end).
__device__ __noinline__ uint32_t bits_at(const uint8_t* p, uint32_t nbytes, int32_t bitpos, int n) {
    if (n == 0) return 0;
    if (bitpos < 0) {
        int neg = -bitpos;
        if (neg >= n) return 0;
        return bits_at(p, nbytes, 0, n - neg) << neg;
    }
    uint32_t byte = (uint32_t)bitpos >> 3;
    if (byte >= nbytes) return 0;
    uint64_t v = ldu64(p + byte);
    uint32_t avail = nbytes - byte;
    if (avail < 8) v &= (1ull << (avail * 8)) - 1;
    v >>= (bitpos & 7);
    return (uint32_t)(v & ((1ull << n) - 1));
}

#ifdef MZD_EXP_PLANDIAG // (experiment: the first live plan lane whose position left the stream)
static __device__ uint32_t g_plandiag[16];
#endif
// Intra-workgroup flags in LDS (the block pipeline): relaxed atomics + workgroup fences.  Every spin
// also ends when an error is posted, and is bounded.
__device__ __forceinline__ uint32_t flag_load(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
// The same for a wavefront that branches on the word: all lanes read one address, but the compiler cannot know that -- a branch on the
// loaded value counts as divergent, and every wave-uniform boolean alive across it is then merged through exec-mask arithmetic
// (three scalar instructions each, at every join).  Reading it through readfirstlane makes value and branch uniform.
__device__ __forceinline__ uint32_t flag_load_u(const uint32_t* p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(p, __ATOMIC_RELAXED)); }
// LDS by offset: an address-space-qualified access is a DS instruction (a generic pointer into LDS makes FLAT ones, which
// take the vector memory path as well)
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) { return (uint32_t)(uintptr_t)(const lds_byte*)p; }
__device__ __forceinline__ uint64_t lds_load_u64(uint32_t off) { uint64_t v; __builtin_memcpy(&v, (const lds_byte*)(uintptr_t)off, 8); return v; } // any alignment
__device__ __forceinline__ void lds_store_u128(uint32_t off, const uint4& v) { __builtin_memcpy((lds_byte*)__builtin_assume_aligned((lds_byte*)(uintptr_t)off, 16), &v, 16); } // 16-byte aligned
__device__ __forceinline__ void flag_store(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
// first error wins: a wavefront that merely gave up because another one failed must not overwrite the cause
__device__ __forceinline__ void post_err(int32_t* err, int rc) {
    if (rc) { int32_t expected = 0; __atomic_compare_exchange_n(err, &expected, rc, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); }
}
// (a wait that runs out is a failure of the launch -- a co-tenant starved the workgroup, a role died -- not of the input:
//  it posts MZD_E_DEVICE, and whatever the waiting role reports afterwards loses to it)
#ifdef MZD_EXP_DEVSITE // (experiment: which bounded wait ran out -- the highest site id seen, in counter word 7)
__device__ uint32_t g_devsite[4]; // first, max, count (read and cleared by devsite_take, mzd_kernels.hip)
#define DEVSITE(id) do { atomicCAS(&g_devsite[0], 0u, (uint32_t)(id)); atomicMax(&g_devsite[1], (uint32_t)(id)); atomicAdd(&g_devsite[2], 1u); } while (0)
#else
#define DEVSITE(id) ((void)0)
#endif
__device__ __forceinline__ bool spin_ge(const uint32_t* p, uint32_t want, int32_t* err) {
    for (uint32_t it = 0; it < (1u << 24); it++) {
        if (flag_load(p) >= want) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return true; }
        if (__atomic_load_n(err, __ATOMIC_RELAXED)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    DEVSITE(1 + (uint32_t)(((const uint8_t*)p - (const uint8_t*)&S) & 0xFFF) * 16);
    post_err(err, MZD_E_DEVICE);
    return false;
}


// The group's barrier.  One group in the workgroup: the hardware's.  Two: s_barrier would join all eight wavefronts, whose groups work on
// different files -- the four wavefronts of a group count their arrivals in their image instead (arrival k of barrier b is 4 b + k; everybody
// leaves when 4 b + 4 are in).  Same memory semantics as __syncthreads: the wavefront's stores are complete before, its loads start after.
#if MZD_PAIRS
__device__ __noinline__ void grp_sync_counted() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t* const bar = &S.bar;
    uint32_t old = __atomic_fetch_add(bar, (threadIdx.x & 63) == 0 ? 1u : 0u, __ATOMIC_RELAXED); // (every lane takes part: no divergent region around the atomic)
    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
    const uint32_t target = (old | 3u) + 1u;
    if (old + 1u != target) {
        uint32_t it = 0;
        for (; it < (1u << 22) && (int32_t)(flag_load_u(bar) - target) < 0; it++) __builtin_amdgcn_s_sleep(4); // (a waiting wavefront must not eat issue slots: the hardware's barrier costs none)
#ifdef MZD_EXP_PLANDIAG
        if (it == (1u << 22) && (threadIdx.x & 63) == 0 && atomicCAS(&g_plandiag[0], 0u, 4u) == 0u) { g_plandiag[1] = S.c.job; g_plandiag[2] = old; g_plandiag[3] = flag_load(bar); g_plandiag[4] = threadIdx.x; g_plandiag[6] = blockIdx.x; g_plandiag[7] = vgrid(); }
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void grp_sync() { grp_sync_counted(); }
#else
__device__ __forceinline__ void grp_sync() { __syncthreads(); }
#endif

// wave-wide inclusive scans, hand-written on the DPP path (no LDS traffic, no ds_bpermute latency): Hillis-Steele inside a row
// of 16 lanes (row_shr 1, 2, 4, 8: a lane whose source lies outside its row keeps the identity), then a row's total into the row
// above it (row_bcast:15 into rows 1 and 3) and the lower half's total into the upper half (row_bcast:31 into rows 2 and 3).
// op(earlier, later); `ident` is op's identity.
template <int CTRL, int ROWMASK> __device__ __forceinline__ uint32_t dpp_take(uint32_t ident, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROWMASK, 0xF, false);
}
template <class Op> __device__ __forceinline__ uint32_t wave_incl_scan_op(uint32_t v, uint32_t ident, Op op) {
    v = op(dpp_take<0x111, 0xF>(ident, v), v);
    v = op(dpp_take<0x112, 0xF>(ident, v), v);
    v = op(dpp_take<0x114, 0xF>(ident, v), v);
    v = op(dpp_take<0x118, 0xF>(ident, v), v);
    v = op(dpp_take<0x142, 0xA>(ident, v), v);
    v = op(dpp_take<0x143, 0xC>(ident, v), v);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    (void)lane;
    return wave_incl_scan_op(v, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}

