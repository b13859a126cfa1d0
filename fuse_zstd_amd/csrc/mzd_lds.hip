// mzd_lds.hip -- the small-file kernel: a file's whole decode inside LDS.
//
// Same arithmetic as mzd_kernels.hip (the frame decoder behind `zstd::stream::copy_decode`, reference
// src/main.rs:463-467; format: RFC 8878 / SURVEY.md Appendix A), for the corpus the reference's own benchmark reads
// (benchmarks/parallel-files.fio:3-7: thousands of files of a few KiB).  A wavefront takes a GROUP of G files and gives
// each LPF = 64 / G lanes.  A file's LDS slot is used twice:
//
//     entropy phase:   [ ring | tables | compressed input ]        execute phase:   [ output window ]
//
//   * the compressed file is copied into the slot once, 16 bytes per lane (coalesced); every parser and bit reader
//     works on LDS bytes;
//   * Huffman literals go to a scratch in HBM (L2-resident: written and read by the same wavefront);
//   * the FSE state walk does only what the chain needs (three table reads, one bitstream window, the three state
//     updates: mzd_k_walk.h's step, with per-lane tables) and records its state per sequence in the ring; field
//     extraction is done by LPF lanes at once, a lane per sequence, from those records; the sequences -- 8 bytes each:
//     literal length, match length, offset value -- follow the literals into the scratch;
//   * then the slot becomes the file's OUTPUT WINDOW (what the entropy phase kept there is dead; residency is set by
//     max(entropy image, window), not by their sum): LPF sequences at a time, a lane per sequence -- positions by scans,
//     repeat offsets by a scan over references, every lane copies its own literals, matches are resolved LDS -> LDS in
//     rounds (never an HBM round trip);
//   * XXH64 reads the window; the finished file leaves LDS with whole-wavefront 16-byte stores (to the destination and,
//     when the caller's buffer is pinned host memory, to its mirror: DevJob::dst2).
// Chains that are serial by construction (Huffman weights, normalized counts, the state walk) run on one lane per file,
// the files of a group in lockstep.  Only the plain case is decoded here: ONE frame of ONE block without an error of any
// kind.  Anything else -- several frames or blocks, skippable frames, tables that do not fit the slot, every malformed
// input -- is handed on untouched (nothing has left LDS by then): the job index is appended to the launch's redo list and
// the general driver behind this kernel decodes it, so error classes and their order stay those of the block pipeline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <type_traits>

#include "../../include/mzd.h"
#include "mzd_device.h"
#include "mzd_tables.h"

namespace mzd {
namespace lw {

#define DI __device__ __forceinline__

typedef const __attribute__((address_space(1))) uint8_t* gcp;
typedef __attribute__((address_space(1))) uint8_t* gp;
struct V16 { uint64_t a, b; }; // 16 bytes, any alignment
DI V16 gv16(const uint8_t* p) { V16 v; __builtin_memcpy(&v, (gcp)p, 16); return v; }
DI void gsv16(uint8_t* p, const V16& v) { __builtin_memcpy((gp)p, &v, 16); }
DI uint32_t gu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, (gcp)p, 4); return v; }
DI uint32_t gu8(const uint8_t* p) { return *(gcp)p; }
DI uint64_t gu64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, (gcp)p, 8); return v; }
DI void gs32(uint8_t* p, uint32_t v) { __builtin_memcpy((gp)p, &v, 4); }
DI void gs64(uint8_t* p, uint64_t v) { __builtin_memcpy((gp)p, &v, 8); }
DI void gs8(uint8_t* p, uint32_t v) { *(gp)p = (uint8_t)v; }
DI int hibit32(uint32_t v) { return 31 - __builtin_clz(v); }
DI uint32_t bfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); } // (offset and width: low 5 bits)

// ------------------------------------------------------------------------------------ LDS image (dynamic)
// Offsets are LDS ADDRESSES: the dynamic segment is the kernel's only LDS object, so it starts at 0 (checked at kernel entry; a
// launch where it does not hands every file on).  Spelled through the array's symbol, every access with a computed address
// cost one more instruction (`v_add_u32 v, lds, v` with lds = 0, resolved only at link time).
extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
#define MZD_LDS_AS __attribute__((address_space(3)))
template <class T> DI MZD_LDS_AS T* lptr(uint32_t off) { return (MZD_LDS_AS T*)(uintptr_t)off; }
DI MZD_LDS_AS uint32_t& L32(uint32_t off) { return *lptr<uint32_t>(off); }
DI MZD_LDS_AS uint16_t& L16(uint32_t off) { return *lptr<uint16_t>(off); }
DI MZD_LDS_AS int16_t& L16s(uint32_t off) { return *lptr<int16_t>(off); }
DI MZD_LDS_AS uint8_t& L8(uint32_t off) { return *lptr<uint8_t>(off); }
DI MZD_LDS_AS uint64_t& L64(uint32_t off) { return *lptr<uint64_t>(off); } // 8-byte aligned
DI uint64_t lds_u64(uint32_t off) { uint64_t v; __builtin_memcpy(&v, lptr<uint8_t>(off), 8); return v; } // any alignment
DI uint32_t lds_u32(uint32_t off) { uint32_t v; __builtin_memcpy(&v, lptr<uint8_t>(off), 4); return v; }
DI V16 lds_u128(uint32_t off) { V16 v; __builtin_memcpy(&v, lptr<uint8_t>(off), 16); return v; } // any alignment: ONE ds_read_b128 (two of them for 32 bytes ran faster than the ds_read2_b64 pairs the compiler makes of four 8-byte reads)
DI V16 lds_v16(uint32_t off) { V16 v; __builtin_memcpy(&v, (MZD_LDS_AS uint8_t*)__builtin_assume_aligned(lptr<uint8_t>(off), 16), 16); return v; } // 16-byte aligned
DI void lds_sv16(uint32_t off, const V16& v) { __builtin_memcpy((MZD_LDS_AS uint8_t*)__builtin_assume_aligned(lptr<uint8_t>(off), 16), &v, 16); }

DI void lds_s64(uint32_t off, uint64_t v) { __builtin_memcpy(lptr<uint8_t>(off), &v, 8); } // any alignment
DI void lds_s32(uint32_t off, uint32_t v) { __builtin_memcpy(lptr<uint8_t>(off), &v, 4); }
DI void lds_s16(uint32_t off, uint32_t v) { const uint16_t w = (uint16_t)v; __builtin_memcpy(lptr<uint8_t>(off), &w, 2); }
DI void lds_xor32(uint32_t off, uint32_t v) { __hip_atomic_fetch_xor(lptr<uint32_t>(off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DI void lds_add32(uint32_t off, uint32_t v) { __hip_atomic_fetch_add(lptr<uint32_t>(off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DI void lds_or32(uint32_t off, uint32_t v) { __hip_atomic_fetch_or(lptr<uint32_t>(off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The lanes of a wavefront talk through LDS without a barrier: a wavefront's LDS operations execute in issue order, so a
// read issued after another lane's write sees it.  What has to be kept is the ORDER OF ISSUE: no memory operation moves
// across this point.
DI void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("" ::: "memory");
}

// shared by the wavefront: code tables; the dictionary's image (DICT)
constexpr uint32_t kShLL = 0;     // uint32 [36]: baseline | extra bits << 24
constexpr uint32_t kShML = 144;   // uint32 [53]
constexpr uint32_t kShWalkDummy = 360; // 8 bytes: {its own address, 0}: the entry the walk's fourth lane follows
constexpr uint32_t kShDump = 384;  // 8 bytes per lane: where the stores of idle lanes go (select-style code: no branch around a store)
constexpr uint32_t kShBytes = 896;
constexpr uint32_t kDLL = 0, kDML = 4096, kDOF = 8192, kDHuf = 10240, kDictImg = 14336; // FSE entries of 8 bytes, Huffman entries of 2
constexpr uint32_t kAux = 256;    // per file: the normalized counts of the three sequence tables, later the walk records
// scratch of the Huffman weights: their FSE table [64 x 8] | its counts
constexpr uint32_t kWTab = 0, kWNorm = 512, kWStage = 576; // (the weights' bitstream behind 16 zero bytes: 576 .. 720) // in the table area, which the Huffman table takes over once the weights are decoded; the weights themselves: the ring

// ---- the file's LPF lanes (LPF = 16, 8 or 4: inside one DPP row of 16 lanes; LPF = 32: two rows)
// the value of the lane N below, 0 for the first N lanes of the file (LPF = 32: of each of its rows -- seg_scan_add carries over)
template <int N, int LPF> DI uint32_t seg_shr(uint32_t v, uint32_t sub) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, false); // row_shr:N
    return (LPF >= 16 || sub >= (uint32_t)N) ? t : 0u;
}
// inclusive prefix sum over the file's lanes
template <int LPF> DI uint32_t seg_scan_add(uint32_t v, uint32_t sub) {
    v += seg_shr<1, LPF>(v, sub); v += seg_shr<2, LPF>(v, sub);
    if (LPF > 4) v += seg_shr<4, LPF>(v, sub);
    if (LPF > 8) v += seg_shr<8, LPF>(v, sub);
    if (LPF > 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3: the total of the file's lower row
    return v;
}
// the value of the file's lane K, in all its lanes
template <int K, int LPF> DI uint32_t bcast(uint32_t v) {
    if (LPF == 32) { // (two rows: through scalar registers)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v, K), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 32 + K);
        uint32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(0xFFFFFFFF00000000ull)); return r; // (one scalar operand per instruction: the values go through vector registers)
    }
    if (LPF == 16) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 + (K & 15), 0xF, 0xF, false); // row_newbcast:K
    if (LPF == 8) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x150 + (K & 7), 0xF, 0x3, false);
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x150 + 8 + (K & 7), 0xF, 0xC, false);
    }
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, (K & 3) * 0x55, 0xF, 0xF, false); // quad_perm: [K, K, K, K]
}
// a file's share of a ballot
template <int LPF> DI uint32_t file_bits(uint64_t ballot, uint32_t f) { return (uint32_t)(ballot >> (f * LPF)) & (uint32_t)((1ull << LPF) - 1); }
// Lane masks and selects that stay selects: left to itself the compiler turns a chain of `c == k ? a : b` into exec-mask
// branch regions (two or three scalar instructions and a branch each; on a lone wavefront every one of them costs an issue slot)
typedef uint64_t lmask;
DI lmask m_eq(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 32); }
DI lmask m_ne(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 33); }
DI lmask m_gt(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 34); }
DI lmask m_ge(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 35); }
DI lmask m_lt(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 36); }
DI lmask m_le(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 37); }
DI uint32_t sel(lmask m, uint32_t t, uint32_t f) { uint32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m)); return r; }
// ---- repeat offsets (A.5) as a scan.  The state before a sequence is three REFERENCES: to one of the three offsets the step
// started with (0, 1, 2) or to the offset value a sequence of the step brought along (flag | its lane in the file; the flag is
// 0x10, 0x20 with 32 lanes a file).  A sequence is a transform of that triple -- a new offset: (own, s0, s1); "repeat 0": identity;
// 1: (s1, s0, s2); 2: (s2, s0, s1) -- kept as three bytes, and composing two of them is one byte permute: the later one's bytes
// select among the earlier one's, except where they are constants.  ("repeat 0 minus one" makes a new VALUE out of a reference:
// steps that hold one take the serial form.)
constexpr uint32_t kRepId = 0x03020100u;
template <int LPF> struct RepRef { static constexpr uint32_t flag = LPF > 16 ? 0x20u : 0x10u, sh = LPF > 16 ? 5u : 4u; };
template <int LPF> DI uint32_t rep_compose(uint32_t later, uint32_t earlier) {
    const uint32_t p = __builtin_amdgcn_perm(0u, earlier, later);
    const uint32_t mask = ((later & (RepRef<LPF>::flag * 0x00010101u)) >> RepRef<LPF>::sh) * 0xFFu;
    return (later & mask) | (p & ~mask);
}
template <int N, int LPF> DI uint32_t rep_shr(uint32_t v, uint32_t sub) { // the transform of the lane N below, the identity for the first N lanes of the file (LPF = 32: of each row)
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)kRepId, (int)v, 0x110 + N, 0xF, 0xF, false);
    return (LPF >= 16 || sub >= (uint32_t)N) ? t : kRepId;
}
// inclusive scan of the transforms over the file's lanes
template <int LPF> DI uint32_t rep_scan(uint32_t P, uint32_t sub) {
    P = rep_compose<LPF>(P, rep_shr<1, LPF>(P, sub));
    P = rep_compose<LPF>(P, rep_shr<2, LPF>(P, sub));
    if (LPF > 4) P = rep_compose<LPF>(P, rep_shr<4, LPF>(P, sub));
    if (LPF > 8) P = rep_compose<LPF>(P, rep_shr<8, LPF>(P, sub));
    if (LPF > 16) P = rep_compose<LPF>(P, (uint32_t)__builtin_amdgcn_update_dpp((int)kRepId, (int)P, 0x142, 0xA, 0xF, false)); // (row_bcast:15: the lower row's composition)
    return P;
}
// the scan's value one lane down: the state before this lane's sequence (the identity for the file's first lane)
template <int LPF> DI uint32_t rep_before(uint32_t P, uint32_t sub) {
    if (LPF <= 16) return rep_shr<1, LPF>(P, sub);
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)kRepId, (int)P, 0x138, 0xF, 0xF, false); // wave_shr:1
    return sub >= 1u ? t : kRepId;
}

// n (<= 31) bytes to LDS offset `at` in exact pieces of 16 / 8 / 4 / 2 / 1, in two parts, so that a wavefront none of whose lanes holds 16 bytes or more skips the upper read and the first two stores:
// store_piece16 takes the 16-byte piece (n & 16) from lo and moves hi down; store_exact15 stores n & 15 bytes held in d[0..3]
// (Round 5: the stores are EXEC-MASKED -- `if (n & 8) store` -- where round 4 aimed the idle lanes' stores at a dump area to save the
//  exec-mask bookkeeping: an LDS instruction is served in groups of 16 lanes (a file's lanes are one group) and a group without an
//  active lane costs the LDS pipeline nothing, while a store to the dump costs it as much as a real one -- and the CU's LDS pipeline,
//  shared by all its wavefronts, is what the execution is short of: cfg4 -3.4 %, cfg4x4 -7.8 %, cfg5 -4 %.)
DI void store_piece16(uint32_t& at, uint32_t n, uint32_t& d0, uint32_t& d1, uint32_t& d2, uint32_t& d3, uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3) {
    if (n & 16) { lds_s64(at, (uint64_t)d0 | ((uint64_t)d1 << 32)); lds_s64(at + 8, (uint64_t)d2 | ((uint64_t)d3 << 32)); d0 = h0; d1 = h1; d2 = h2; d3 = h3; at += 16; }
}
DI void store_exact15(uint32_t at, uint32_t n, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3) {
    if (n & 8) { lds_s64(at, (uint64_t)d0 | ((uint64_t)d1 << 32)); d0 = d2; d1 = d3; at += 8; }
    if (n & 4) { lds_s32(at, d0); d0 = d1; at += 4; }
    if (n & 2) { lds_s16(at, d0); d0 >>= 16; at += 2; }
    if (n & 1) L8(at) = (uint8_t)d0;
}
// the minimum over the file's lanes, in all of them: DPP swaps inside quads, half rows and rows (no trip through the LDS crossbar
// but for the second row of 32 lanes)
template <int LPF> DI uint32_t seg_min(uint32_t x, uint32_t lane) { // (the DPP operand folded into the minimum: the compiler makes a move and a minimum of each stage)
    if (LPF >= 16) asm("s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                       " v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
    else if (LPF == 8) asm("s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                           " v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x));
    else asm("s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x));
    if (LPF >= 32) x = min(x, (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ 16u) * 4u), (int)x));
    return x;
}
// n bytes from LDS offset src to dst <= src by the file's LPF lanes, four bytes a lane and round (ascending rounds, every round reads
// before it writes: a round's stores end below the next round's reads); the run's last 1..3 bytes in exact pieces
template <int LPF> DI void copy_run_lanes(uint32_t dst, uint32_t src, uint32_t n, uint32_t sub) {
    for (uint32_t q = 4 * sub; q < n + 4 * sub; q += 4 * LPF) { // (the same number of rounds for all the file's lanes)
        const uint32_t left = q < n ? n - q : 0u;
        uint32_t v = 0;
        if (left) v = lds_u32(src + q);
        asm volatile("" ::: "memory");
        if (left >= 4) lds_s32(dst + q, v);
        else { if (left & 2) lds_s16(dst + q, v); if (left & 1) L8(dst + q + (left & 2)) = (uint8_t)(v >> (8 * (left & 2))); }
        asm volatile("" ::: "memory");
    }
}
template <int N, class F> DI void static_for(F&& f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// FSE decode entry, 8 bytes (mzd_k_tables.h: pack_entry): low word = LDS address of the entry of next-state base (so that
// the new state's address is low + 8 * bits); high word = nbBits | (extra + nbBits) << 8 | symbol << 16 | extra << 24
DI uint64_t fse_entry(uint32_t tab_off, uint32_t nbase, uint32_t nb, uint32_t sym, uint32_t extra) {
    return (uint64_t)(tab_off + nbase * 8u) | ((uint64_t)(nb | ((extra + nb) << 8) | (sym << 16) | (extra << 24)) << 32);
}

#include "mzd_l_tables.h"

// ------------------------------------------------------------------------------------ bit readers over LDS bytes
struct LBack { // backward (the Huffman weights' stream: <= 128 bytes), zero below the start
    uint32_t base;
    int32_t h;
    uint64_t cur;
    int32_t avail;
    DI bool init(uint32_t off, uint32_t sl) {
        base = off; cur = 0; avail = 0; h = 0;
        if (sl == 0) return false;
        const uint32_t last = L8(off + sl - 1);
        if (last == 0) return false;
        h = (int32_t)((sl - 1) * 8) + hibit32(last);
        return true;
    }
    DI void refill() {
        if (h <= 0) { cur = 0; avail = 64; return; }
        const int32_t b = (h - 1) >> 3;
        uint64_t W = b >= 7 ? lds_u64(base + (uint32_t)(b - 7)) : lds_u64(base) << (8 * (7 - b)); // (bytes before the stream are not its own)
        const int32_t sh = 8 * (b + 1) - h;
        cur = W << sh;
        avail = 64 - sh;
        if (h < avail) cur &= ~0ull << (64 - h);
    }
    DI uint32_t read(uint32_t n) { // n <= 32
        if ((int32_t)n > avail) refill();
        const uint32_t v = n ? (uint32_t)(cur >> (64 - n)) : 0u;
        cur <<= n; avail -= (int32_t)n; h -= (int32_t)n;
        return v;
    }
};

// Normalized counts (A.3) from LDS bytes (>= 16 readable bytes behind them) -> int16 norm[] in LDS at `norm_off` (up to three
// entries past the last symbol are written too: zeros).  Returns bytes used or 0 (give up).  sym_cap: symbols the caller has room
// for (<= max_sym + 1).  One lane per file, the files of a group in lockstep, so a round is ONE straight piece of code whatever it
// reads -- a count, a count of zero with the zero-run field behind it, or a zero-run field that continues a run (round 3's form
// branched per case: four files in four different cases ran all of them, 90 instructions a symbol).  The description is read
// upwards through a 64-bit register window; the window a round uses was requested by the round before it (a round consumes at
// most 12 bits), so no round waits for LDS.  Bits past the description's end need no masking and no check per field: every field
// is at least one bit wide, so a field that touches them leaves `bit` past the limit, which is checked once at the end; a
// count that overshoots leaves `remaining` negative, which ends the loop and fails the same way.
DI uint32_t read_ncount_lane(uint32_t src_off, uint32_t n, int max_log, int max_sym, int sym_cap, uint32_t norm_off, uint32_t& nsym_out, uint32_t& log_out) {
    if (n < 1) return 0;
    const int32_t limit = (int32_t)(n > 4096 ? 4096 : n) * 8;
    uint64_t wc = lds_u64(src_off); // description bits [bc, bc + 64)
    const int al = 5 + (int)((uint32_t)wc & 15);
    if (al > max_log) return 0;
    const int32_t lim = max_sym + 1 < sym_cap ? max_sym + 1 : sym_cap;
    int32_t bit = 4, bc = 0, remaining = 1 << al, sym = 0;
    uint32_t rep = 0; // the next field continues a zero run
    for (;;) {
        const bool go = (sym <= lim) & ((rep != 0) | ((remaining > 0) & (sym < lim))); // (no short circuits: each one was an exec-mask region of its own, twenty scalar instructions a round)
        if (!go) break;
        const int32_t bn = bit & ~7;
        const uint64_t wn = lds_u64(src_off + ((uint32_t)bn >> 3));
        const uint32_t x = (uint32_t)(wc >> (bit - bc));
        const uint32_t rp1 = (uint32_t)remaining + 1;              // (>= 1)
        const uint32_t nb = 32u - (uint32_t)__builtin_clz(rp1);
        const uint32_t half = 1u << (nb - 1);
        const uint32_t vlow = x & (half - 1), vfull = x & (2 * half - 1);
        const uint32_t thr = 2 * half - 1 - rp1;
        const bool small = vlow < thr;
        const uint32_t v2 = small ? vlow : vfull - (vfull >= half ? thr : 0u);
        const int32_t pr = (int32_t)v2 - 1;
        const uint32_t adv = nb - (small ? 1u : 0u);
        const bool zero = rep == 0 && pr == 0;
        const uint32_t r = (rep ? x : x >> adv) & 3;              // the zero-run field this round consumes, if it does
        lds_s64(norm_off + 2 * (uint32_t)sym, rep ? 0ull : (uint64_t)(uint16_t)pr);
        bit += rep ? 2 : (int32_t)adv + (zero ? 2 : 0);
        remaining -= rep ? 0 : (pr < 0 ? 1 : pr);
        sym += rep ? (int32_t)r : 1 + (zero ? (int32_t)r : 0);
        rep = ((((rep != 0) | zero) & (r == 3)) ? 1u : 0u);
        wc = wn; bc = bn;
    }
    if (remaining != 0 || sym > lim || bit > limit) return 0;
    nsym_out = (uint32_t)sym;
    log_out = (uint32_t)al;
    return (uint32_t)((bit + 7) >> 3);
}

// ------------------------------------------------------------------------------------ XXH64 pieces (A.6), over LDS bytes
DI uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
DI uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
DI uint64_t xmerge(uint64_t hh, uint64_t v) { v = xround(0, v); hh ^= v; return hh * XP1 + XP4; }
// The stripe loop with the product in * P2 off the chain (mzd_k_xxh64.h's form, for a file's lanes): lane 4 s + a of the file holds
// the product for stripe s of a group, accumulator a; the chain runs in the file's first four lanes and picks the products of
// stripes 1.. out of the lanes above (DPP row shifts folded into the 64-bit add): one 64-bit multiply per stripe on the chain, not two.
DI uint64_t rotl64_31(uint64_t x) { const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32); return (uint64_t)__builtin_amdgcn_alignbit(lo, hi, 1) | ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 1) << 32); }
template <int J> DI uint64_t add_row_up(uint64_t acc, uint64_t t) { // acc + (t of the lane 4 J further up in the row; lanes past the row's end add 0)
    if (J == 0) return acc + t;
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    const uint32_t tlo = (uint32_t)t, thi = (uint32_t)(t >> 32);
    if (J == 1) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    if (J == 2) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    if (J == 3) asm("v_add_co_u32_dpp %0, vcc, %2, %0 row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32_dpp %1, vcc, %3, %1, vcc row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo), "+v"(hi) : "v"(tlo), "v"(thi) : "vcc");
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
template <int J> DI uint64_t xchain31(uint64_t acc, uint64_t t) { acc = add_row_up<J>(acc, t); acc = rotl64_31(acc); return acc * XP1; }
DI uint64_t xxh_tail(uint64_t hh, uint32_t q, uint32_t end) {
    while (q + 8 <= end) { hh ^= xround(0, lds_u64(q)); hh = rotl64(hh, 27) * XP1 + XP4; q += 8; }
    if (q + 4 <= end) { hh ^= (uint64_t)lds_u32(q) * XP1; hh = rotl64(hh, 23) * XP2 + XP3; q += 4; }
    while (q < end) { hh ^= (uint64_t)L8(q) * XP5; hh = rotl64(hh, 11) * XP1; q++; }
    hh ^= hh >> 33; hh *= XP2; hh ^= hh >> 29; hh *= XP3; hh ^= hh >> 32;
    return hh;
}

// Diagnostic build only (-DMZD_SMALL_STAMPS): cycle counter of workgroup 0 at every phase boundary of its first group.
// -DMZD_SMALL_STAMPS_LIGHT: the same without the cycle ACCUMULATORS inside the walk and the execution (their registers put the 8 / 4
// kernel past 256 and halve its residency: a cfg4 launch then takes two rounds) -- the per-workgroup clock values alone, in the product's shape.
#ifdef MZD_SMALL_STAMPS_LIGHT
#define MZD_SMALL_STAMPS
#define MZD_SS_NOACC
#endif
#ifdef MZD_SMALL_STAMPS
#define SSTAMP(k) do { if (a.stamps && w0 && wv == 0 && lane == 0 && first_group) { if (blockIdx.x == 0) a.stamps[k] = __builtin_readcyclecounter(); if ((k) < 12 && blockIdx.x < 3072) a.stamps[2048 + 16 * blockIdx.x + 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define SSTAMP(k)
#endif

// A match the in-order step's one-byte-per-lane form does not cover: longer than LPF, overlapping its own output, or
// starting in the dictionary content (logically just before the frame).  The file's lanes, a byte each per round.
template <int LPF, bool DICT>
__device__ __noinline__ void rare_match(uint32_t outo, uint32_t mp, uint32_t off, uint32_t m, uint32_t sub, const uint8_t* dict_end, uint32_t dict_len) {
    if (off - 1 >= mp + dict_len) return; // (a wrong offset: the file is handed on)
    if (DICT && off > mp) {
        const uint32_t back = off - mp;
        const uint32_t n1 = m < back ? m : back;
        for (uint32_t q = sub; q < n1; q += LPF) L8(outo + mp + q) = (uint8_t)gu8(dict_end - back + q);
        mp += n1; m -= n1;
        if (!m) return;
    }
    asm volatile("" ::: "memory");
    if (off >= LPF || off >= m) { // a round never reads what it writes
        for (uint32_t q = sub; q < m; q += LPF) { const uint32_t v = L8(outo + mp - off + q); asm volatile("" ::: "memory"); L8(outo + mp + q) = (uint8_t)v; asm volatile("" ::: "memory"); }
    } else { // period < LPF: every lane repeats one byte of the pattern
        uint32_t rr = sub; // sub mod off
        if (LPF > 16 && rr >= 16 * off) rr -= 16 * off;
        if (rr >= 8 * off) rr -= 8 * off;
        if (rr >= 4 * off) rr -= 4 * off;
        if (rr >= 2 * off) rr -= 2 * off;
        if (rr >= off) rr -= off;
        uint32_t per = off; // the largest multiple of the period <= LPF
        while (per + off <= LPF) per += off;
        const uint32_t v = L8(outo + mp - off + rr);
        asm volatile("" ::: "memory");
        if (sub < per) for (uint32_t q = sub; q < m; q += per) L8(outo + mp + q) = (uint8_t)v;
    }
    asm volatile("" ::: "memory");
}

// The state walk's hot form, hand-scheduled (mzd_k_walk.h's step with per-lane tables: an entry's low word is the LDS address of its
// next-state base, so the new state's address is low + 8 * bits; bit positions are LDS bit addresses).  One call walks N sequences and
// records each one's state {LL, ML, OF entry addresses, read head - 32} in the ring (16 bytes a sequence) before stepping over it.
// The hot form of the walk (mzd_k_walk.h's, for per-file tables): the file's three states live in three LANES of its first quad
// (lane 0 LL, 1 ML, 2 OF; lane 3 follows a dummy entry that consumes nothing and leads to itself), so a step is ONE table read, one
// field extract and one address add; the other states' bit counts come through DPP quad permutes of the entry's high word as it was
// loaded.  On a lone wavefront a step costs its instruction slots in front of the table read plus that read's round trip: 8 VALU
// slots here against 12 + three reads with one lane per file.  Behind the read, in the shadow of its latency: the read head, the next
// window's read, the record (lane k stores dword k of {LL, ML, OF state address, read head - 32}).  A step whose sequence is wider
// than the window (32..63 bits below the head) leaves `slack` negative: the caller takes the whole call again with the C++ form
// from the state it saved.  LDS addresses are spelled as they are: the dynamic LDS segment must start at 0 (checked).
#define MZD_SDWA_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define MZD_DPP_ALL " row_mask:0xf bank_mask:0xf\n"
#define MZD_LW_SHADOW(RECOFF) \
    "v_lshrrev_b32_e32 v71, 3, v87\n" \
    "v_and_b32_e32 v71, 0x3fffc, v71\n" \
    "ds_read2_b32 v[54:55], v71 offset1:1\n" \
    "v_cndmask_b32_e64 v70, v84, v87, %[l3]\n" \
    "ds_write_b32 %[ring], v70 offset:" RECOFF "\n"   /* the NEXT step's record: the state as it is now */ \
    "v_and_or_b32 %[av], v87, 31, 32\n"
#define MZD_LW_STEP(SH) \
    "s_waitcnt lgkmcnt(2)\n"                                            /* the entry is there (behind it: the window, the record) */ \
    "v_add_u32_dpp v64, v49, v49 quad_perm:[1,0,3,2]" MZD_DPP_ALL       /* pair sums of the high words */ \
    "v_mov_b32_dpp v65, v49 quad_perm:[1,2,3,3]" MZD_DPP_ALL            /* the high word one lane up (lane 2: the dummy's, 0) */ \
    "v_add_u32_dpp v65, v49, v65 quad_perm:[2,3,3,3]" MZD_DPP_ALL       /* + two lanes up: bit offset of the own field: nbO + nbM, nbO, 0 */ \
    "v_add_u32_dpp v64, v64, v64 quad_perm:[2,3,0,1]" MZD_DPP_ALL       /* all four: nbBits sums | total bits << 8 (v64: written two slots ago) */ \
    "v_sub_u32_sdwa " SH ", %[av], v64" MZD_SDWA_B1                     /* window bits below what this sequence consumes */ \
    "s_waitcnt lgkmcnt(1)\n"                                            /* the window */ \
    "v_lshrrev_b64 v[66:67], " SH ", v[54:55]\n" \
    "v_bfe_u32 v69, v66, v65, v49\n"                                    /* the lane's fresh state bits */ \
    "v_lshl_add_u32 v84, v69, 3, v48\n" \
    "ds_read_b64 v[48:49], v84\n" \
    "v_sub_u32_sdwa v87, v87, v64" MZD_SDWA_B1                          /* (behind the read from here on) the read head */
#define MZD_LW_PAIR(R1, R2) MZD_LW_STEP("%[sa]") MZD_LW_SHADOW(R1) MZD_LW_STEP("%[sb]") MZD_LW_SHADOW(R2) "v_min3_i32 %[slack], %[slack], %[sa], %[sb]\n"
#define MZD_LW_LAST(R1) MZD_LW_STEP("%[sa]") MZD_LW_SHADOW(R1) MZD_LW_STEP("%[sb]") "v_min3_i32 %[slack], %[slack], %[sa], %[sb]\n"
// N steps.  A: the lane's state address (lane & 3: LL, ML, OF, the dummy); ring: the file's record ring + 4 * (lane & 3).
// Runs on the first quad of every file that is walking (the caller's exec mask); everything stays inside the quad.
template <int N>
DI void walk_asm(uint32_t& A, uint32_t& Gm, int32_t& slack, uint32_t ring) {
    uint32_t av, sa, sb;
    const uint64_t l3 = 0x8888888888888888ull; // lane 3 of a quad: its record dword is the read head
#define MZD_LW_HEAD \
        "v_mov_b32_e32 v84, %[A]\n v_mov_b32_e32 v87, %[Gm]\n" \
        "ds_read_b64 v[48:49], v84\n" \
        MZD_LW_SHADOW("0")
#define MZD_LW_TAIL \
        "s_waitcnt lgkmcnt(0)\n v_mov_b32_e32 %[A], v84\n v_mov_b32_e32 %[Gm], v87\n"
#define MZD_LW_OPS \
        : [A] "+v"(A), [Gm] "+v"(Gm), [slack] "+v"(slack), [av] "=&v"(av), [sa] "=&v"(sa), [sb] "=&v"(sb) \
        : [ring] "v"(ring), [l3] "s"(l3) \
        : "v48", "v49", "v54", "v55", "v64", "v65", "v66", "v67", "v69", "v70", "v71", "v84", "v87", "memory"
    if constexpr (N == 16)
        asm volatile(MZD_LW_HEAD MZD_LW_PAIR("16", "32") MZD_LW_PAIR("48", "64") MZD_LW_PAIR("80", "96") MZD_LW_PAIR("112", "128")
                     MZD_LW_PAIR("144", "160") MZD_LW_PAIR("176", "192") MZD_LW_PAIR("208", "224") MZD_LW_LAST("240") MZD_LW_TAIL MZD_LW_OPS);
    else if constexpr (N == 8)
        asm volatile(MZD_LW_HEAD MZD_LW_PAIR("16", "32") MZD_LW_PAIR("48", "64") MZD_LW_PAIR("80", "96") MZD_LW_LAST("112") MZD_LW_TAIL MZD_LW_OPS);
    else
        asm volatile(MZD_LW_HEAD MZD_LW_PAIR("16", "32") MZD_LW_LAST("48") MZD_LW_TAIL MZD_LW_OPS);
}

struct DictInfo { // the dictionary whose image sits in LDS (wave-uniform)
    uint32_t handle, formatted, dict_id, content_len, huf_log;
    uint32_t al[3], rep[3];
    const uint8_t* content;
};

// ------------------------------------------------------------------------------------ the kernel
// One wavefront per workgroup; a persistent loop over groups of G files.
// XG: files EXECUTED at a time (XG divides G).  The entropy phases take all G files of a group through in lockstep (64 / G lanes a
// file: their serial chains cost the same for eight files as for four), then the group's files are executed XG at a time with
// 64 / XG lanes each.  XG < G is for launches that would otherwise need two rounds of groups: the entropy images of G = 8 files of
// 4 KiB (3.9 KB each) fit where only four windows (4.1 KB each) do, so five wavefronts per CU hold 40 files -- 10 240 on the
// device -- instead of 32.  What an execution pass needs to know about a file crosses over in a 32-byte record in LDS.
// NW = 2: a HELPER wavefront beside the one that decodes.  The sequences section's header -- nbSeq, modes, three normalized-count
// descriptions: a serial parse on one lane per file, 48 K of a group's 600 K cycles -- depends on nothing the Huffman phases
// produce, so the helper parses it (on another SIMD) while the first wavefront decodes weights, table and literal streams, and
// hands the result over in LDS; then it sleeps at the group's last barrier.  Three workgroup barriers a group; the helper
// learns the group (and the launch's end) from a control word.  Ten wavefronts a CU instead of five: 168 registers.
// ND > 1 (dictionary launches that name ONE dictionary): ND decoding wavefronts in a workgroup, each with groups, slots, dump area and
// records of its own -- nothing crosses between them, no barrier -- that SHARE the dictionary's table image (14 KB) and the code
// tables: one image a CU instead of one a wavefront is what lets a fifth wavefront's slots fit (cfg5: 32 -> 40 files a CU).  Every
// wavefront writes the image itself before it reads it (the same bytes from all of them).
template <int G, bool DICT, int XG, int NW = 1, int ND = 1>
// Wavefronts a SIMD the kernels are built for -- what lds_waves_by_registers (below) tells the host: three for the plain G = 4 kernel
// (168 registers), two for the other kernels without a dictionary image and for the dictionary kernels of five or more wavefronts a
// workgroup (256), one for the other dictionary kernels (265).  Stated, not left to the compiler: the 8 / 4 kernel sits at exactly 256,
// and one register more would halve a launch's residency without a word (a diagnostic build did: 257, a cfg4 launch in two rounds).
#define MZD_LDS_MINWAVES(G, DICT, NW, ND) (((G == 4 || NW > 1) && !DICT) ? 3 : ((ND > 4 || !DICT) ? 2 : 1))
__global__ __launch_bounds__(64 * NW * ND, MZD_LDS_MINWAVES(G, DICT, NW, ND)) void mzd_lds_kernel(LdsArgs a) {
    constexpr uint32_t LPF = 64 / G; // lanes per file in the entropy phases
    constexpr uint32_t XLPF = 64 / XG, NX = G / XG; // lanes per file = sequences per plan step in the execution; passes
    static_assert(LPF >= 4, "four Huffman streams");
    static_assert(XLPF >= 4 && XLPF <= 32 && G % XG == 0, "four XXH64 accumulators; a file's lanes inside one DPP row, or two (the scans carry over)");
    static_assert(NW == 1 || (NW == 2 && !DICT), "the helper wavefront: files without a dictionary");
    static_assert(ND == 1 || (DICT && NW == 1), "several decoding wavefronts around one dictionary image");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = NW * ND > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
    const bool w0 = NW == 1 || wv == 0; // the decoding wavefront (wave-uniform)
    const uint32_t wslot = blockIdx.x * ND + (ND > 1 ? wv : 0u), nwslots = gridDim.x * ND; // this decoding wavefront among the launch's
    const uint32_t f = lane / LPF, sub = lane % LPF;
    const bool leader = sub == 0;
    // the files' records behind the wavefront's tables; (NW = 2) 64 bytes a file between the two wavefronts: {flags, n, cap, dict} from
    // the decoding one, the parsed sequence header back from the helper
    constexpr uint32_t kShRec = kShBytes, kShSeqRec = kShBytes + 32 * G, kShAll = kShSeqRec + (NW > 1 ? 64 * G : 0);
    constexpr uint32_t kShCtl = kShWalkDummy + 8; // (NW = 2) the group the decoding wavefront is on; >= the number of groups: the launch is over
    auto wgsync = [&]() { if constexpr (NW > 1) __syncthreads(); };
    const uint32_t dict_off = kShAll;
    const uint32_t ent = a.tab_bytes + kAux + a.comp_bytes;                // the entropy phase's image of a file ...
    const uint32_t stride = G != XG ? ent : (ent > a.out_bytes ? ent : a.out_bytes); // ... and (XG == G) its output window share the slot
    // the wavefront's own part of the image: dump area, records, slots (wavefront 0's lie around the shared dictionary image as ever;
    // the others' behind its slots)
    const uint32_t wpriv = (ND > 1 && wv) ? kShAll + kDictImg + G * stride + (wv - 1) * (512 + 32 * G + G * stride) : 0u;
    const uint32_t shDump = wpriv ? wpriv : kShDump, shRec = wpriv ? wpriv + 512 : kShRec;
    const uint32_t slots0 = wpriv ? wpriv + 512 + 32 * G : kShAll + (DICT ? kDictImg : 0u);
    // the file's slot: [ counts, later walk records | tables | compressed input ].  The tables sit right in front of the input: the
    // sequence tables are built when the literals are decoded, so they may grow over the input's dead front -- everything up to
    // the sequences section -- and a launch whose slots leave less than three full tables' room still keeps its files
    const uint32_t ringo = slots0 + f * stride;
    const uint32_t tabo = ringo + kAux, cmp = tabo + a.tab_bytes;
    // the file's share of the scratch in HBM: literals, then the sequences -- 4 bytes each: literal length (7 bits) | match length - 3
    // (6) | offset value (19); a sequence that does not fit (a literal run of 127 bytes or more, a match of 66 or more) says so in
    // its literal-length field and has its full 8-byte record in a second array at the same index, touched by those sequences only
    // (round 4 wrote 8 bytes a sequence: 27 of the 108 MB a launch of 10 000 files moved)
    uint8_t* const lit_g = a.scratch + (size_t)(wslot * G + f) * ((size_t)a.lit_stride + 12u * (size_t)a.seq_cap);
    uint8_t* const seq_g = lit_g + a.lit_stride;
    uint8_t* const seq8_g = seq_g + 4u * (size_t)a.seq_cap; // (seq_cap is even: 8-byte aligned)

    if (w0) {
        if (lane < 36) L32(kShLL + 4 * lane) = LL_BASE[lane] | ((uint32_t)LL_BITS[lane] << 24);
        if (lane < 53) L32(kShML + 4 * lane) = ML_BASE[lane] | ((uint32_t)ML_BITS[lane] << 24);
        if (lane == 0) L64(kShWalkDummy) = (uint64_t)kShWalkDummy;
    }
    wsync();
    DictInfo di;
    di.handle = 0; di.formatted = 0; di.dict_id = 0; di.content_len = 0; di.huf_log = 0; di.content = nullptr;
    for (int t = 0; t < 3; t++) { di.al[t] = 0; di.rep[t] = 0; }

    const bool lds_at_zero = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds == 0; // (offsets are used as LDS addresses)
    // A launch without a general driver behind it (LdsArgs::counter_next): the last wavefront to leave publishes how many files were
    // handed on -- the host decodes them when it collects the launch -- and zeroes the counter block of the lane's next launch.
    auto leave = [&]() {
        if (a.counter_next && w0 && lane == 0 && atomicAdd(&a.counter[6], 1u) == nwslots - 1) { // (a wavefront's additions to word 4 have returned by now)
            *a.handed_on = atomicAdd(&a.counter[4], 0u);
            for (uint32_t k = 0; k < kCounterWords; k++) a.counter_next[k] = 0;
        }
    };
    if (!lds_at_zero) { // cannot be (this kernel has no other LDS object); if it ever is, the general driver decodes everything
        if (w0) for (uint32_t i = wslot * 64 + lane; i < a.n; i += nwslots * 64) a.redo_list[atomicAdd(&a.counter[4], 1u)] = a.list[i];
        leave();
        return;
    }
    const uint32_t ngroups = (a.n + G - 1) / G;
    bool first_group = true;
    (void)first_group;
    // The way to a group's input is four dependent trips to HBM (ticket -> list entry -> job entry -> bytes).  They are made for the
    // NEXT group while this one is decoded, one trip per phase: by the time a group starts, its first kPF x 16 bytes per lane sit in registers.
    constexpr int kPF = 6;
    struct JobRegs { bool have, fits; uint32_t job, n, cap, jdict; const uint8_t* src; uint8_t* dst; uint8_t* dst2; };
    auto ticket = [&]() -> uint32_t { uint32_t t = 0; if (lane == 0) t = atomicAdd(&a.counter[5], 1u); return (uint32_t)__builtin_amdgcn_readfirstlane((int)t); };
    auto list_entry = [&](uint32_t gg) -> uint32_t { const uint32_t fi = gg * G + f; return (gg < ngroups && fi < a.n) ? a.list[fi] : 0xFFFFFFFFu; };
    auto job_entry = [&](uint32_t jb) -> JobRegs {
        JobRegs J; J.have = jb != 0xFFFFFFFFu; J.fits = false; J.job = jb; J.n = 0; J.cap = 0; J.jdict = 0; J.src = nullptr; J.dst = nullptr; J.dst2 = nullptr;
        if (J.have) {
            const DevJob& dj = a.jobs[jb];
            J.src = dj.src; J.dst = dj.dst; J.dst2 = dj.dst2; J.jdict = dj.dict;
            J.fits = dj.src_len + 16 <= a.comp_bytes && dj.dst_cap + 16 <= a.out_bytes; // (the host sized the slots for the launch's largest file)
            J.n = J.fits ? (uint32_t)dj.src_len : 0u; J.cap = J.fits ? (uint32_t)dj.dst_cap : 0u;
        }
        return J;
    };
    auto prefetch = [&](const JobRegs& J, V16 (&pf)[kPF]) { // (inputs are readable MZD_SRC_PADDING bytes past their end)
#pragma unroll
        for (int k = 0; k < kPF; k++) { const uint32_t o = 16 * (sub + LPF * (uint32_t)k); pf[k] = o < J.n ? gv16(J.src + o) : V16{0, 0}; }
    };
#ifdef MZD_SMALL_STAMPS
    // every workgroup: [real-time clock (100 MHz) at entry, at exit, HW_ID | XCC_ID << 32, groups taken] from stamp 2048 on
    const uint64_t wg_t0_ = __builtin_amdgcn_s_memrealtime();
    uint32_t wg_groups_ = 0, wg_rounds_ = 0, wg_steps_ = 0;
    uint64_t wg_te_ = 0;
#endif
    uint32_t g = 0xFFFFFFFFu;
    JobRegs J = job_entry(0xFFFFFFFFu);
    V16 pf[kPF];
    if (w0) { g = ticket(); J = job_entry(list_entry(g)); prefetch(J, pf); }
    for (;;) {
        SSTAMP(0);
        if constexpr (NW > 1) { // the helper learns the group from the decoding wavefront (first barrier of the group)
            if (!w0) { __syncthreads(); g = (uint32_t)__builtin_amdgcn_readfirstlane((int)L32(kShCtl)); }
            else if (g >= ngroups) { if (lane == 0) L32(kShCtl) = g; __syncthreads(); }
        }
        if (g >= ngroups) break;
        // (what the trips for the next group leave behind outlives the entropy phases' scope)
        const bool early = g + nwslots < ngroups;
        uint32_t g_next = 0xFFFFFFFFu;
        JobRegs Jn;
        V16 pfn[kPF];
      { // ---- the entropy phases: G files, LPF lanes each

        // =============================== the group's files: the compressed bytes -> LDS
        const uint32_t fidx = g * G + f;
        bool have = J.have, fits = J.fits;
        uint32_t n = J.n, cap = J.cap, jdict = J.jdict;
        const uint32_t job = J.job;
        const uint8_t* const src = J.src; uint8_t* const dst = J.dst; uint8_t* const dst2 = J.dst2;
        if (w0 && fits) { // 16 bytes per lane: what came ahead, then the rest, four loads in flight
#pragma unroll
            for (int k = 0; k < kPF; k++) { const uint32_t o = 16 * (sub + LPF * (uint32_t)k); if (o < n) lds_sv16(cmp + o, pf[k]); }
            uint32_t k = (sub + LPF * kPF) * 16;
            for (; k + 3 * LPF * 16 < n; k += 4 * LPF * 16) {
                const V16 v0 = gv16(src + k), v1 = gv16(src + k + LPF * 16), v2 = gv16(src + k + 2 * LPF * 16), v3 = gv16(src + k + 3 * LPF * 16);
                lds_sv16(cmp + k, v0); lds_sv16(cmp + k + LPF * 16, v1); lds_sv16(cmp + k + 2 * LPF * 16, v2); lds_sv16(cmp + k + 3 * LPF * 16, v3);
            }
            for (; k < n; k += LPF * 16) lds_sv16(cmp + k, gv16(src + k));
        }
        // (trip 1 for the next group -- only while at least a grid's worth of groups is left behind this one: a wavefront that books its
        //  next group early takes it from one that would have been free sooner, which matters when a launch has about a group per wavefront)
        if (w0 && early) g_next = ticket();
        if constexpr (NW > 1) { // the group's input is in LDS: the helper may start (what it needs to know of a file goes along)
            const uint32_t xo = kShSeqRec + 64 * f;
            if (w0) {
                if (leader) lds_sv16(xo, V16{(uint64_t)((have ? 1u : 0u) | (fits ? 2u : 0u)) | ((uint64_t)n << 32), (uint64_t)cap | ((uint64_t)jdict << 32)});
                if (lane == 0) L32(kShCtl) = g;
                __syncthreads();
            } else {
                const V16 x = lds_v16(xo);
                have = ((uint32_t)x.a & 1u) != 0; fits = ((uint32_t)x.a & 2u) != 0; n = (uint32_t)(x.a >> 32); cap = (uint32_t)x.b; jdict = (uint32_t)(x.b >> 32);
            }
        }
        // the group's dictionary: the first one named (the host sorts the list by dictionary)
        if (DICT) {
            const uint64_t named = __ballot(have && jdict != 0 && jdict <= a.ndicts);
            if (named) {
                const uint32_t want = (uint32_t)__builtin_amdgcn_readlane((int)jdict, __builtin_ctzll(named));
                if (want != di.handle) {
                    const DevDict* dd = &a.dicts[want - 1];
                    di.handle = want; di.formatted = dd->formatted; di.dict_id = dd->dict_id; di.content_len = dd->content_len;
                    di.huf_log = dd->huf_log; di.content = dd->content;
                    for (int t = 0; t < 3; t++) { di.al[t] = dd->al[t]; di.rep[t] = dd->rep[t]; }
                    if (di.formatted) { // (entries as the block pipeline keeps them: next-state offsets relative to the table -> LDS addresses)
                        for (uint32_t i = lane; i < 512; i += 64) { L64(dict_off + kDLL + 8 * i) = dd->ll[i] + (uint64_t)(int64_t)((int32_t)(dict_off + kDLL) - (int32_t)kBlkLdsLL); L64(dict_off + kDML + 8 * i) = dd->ml[i] + (uint64_t)(int64_t)((int32_t)(dict_off + kDML) - (int32_t)kBlkLdsML); } // (rebased: mzd_device.h)
                        for (uint32_t i = lane; i < 256; i += 64) L64(dict_off + kDOF + 8 * i) = dd->of[i] + (uint64_t)(int64_t)((int32_t)(dict_off + kDOF) - (int32_t)kBlkLdsOF);
                        for (uint32_t i = lane; i < 1024; i += 64) L32(dict_off + kDHuf + 4 * i) = __builtin_amdgcn_perm(0u, reinterpret_cast<const uint32_t*>(dd->huf)[i], 0x02030001u); // (this kernel's entries: length | symbol << 8, the bytes of the block pipeline's swapped)
                    }
                }
            }
        }
        wsync();
        SSTAMP(1);

        // =============================== headers: frame, block, literals section (every lane of the file, from LDS)
        uint32_t why = 0; // (diagnostic build: where the file left the fast path)
        (void)why;
        bool ok = have && fits; // still on the fast path
        bool done = false;      // finished without a block to decode (empty file, raw / RLE block)
        uint32_t has_fcs = 0, has_ck = 0, fcs = 0, btype = 0, bsize = 0, b0 = 0;
        uint32_t lit_type = 0, nlit = 0, streams = 0, lit_off = 0, tree_off = 0, tree_len = 0;
        uint32_t s_len0 = 0, s_len1 = 0, s_len2 = 0, s_len3 = 0, s_base = 0;
        uint32_t seq_off = 0, seq_len = 0;
        const bool with_d = DICT && ok && jdict != 0;
        if (ok && n == 0) { done = true; } // no frame at all: nothing to decode
        else if (ok) {
            ok = false;
            do {
                if (jdict > a.ndicts || (jdict && (!DICT || jdict != di.handle))) break;
                if (n < 9) break;
                if (lds_u32(cmp) != 0xFD2FB528u) break;
                const uint32_t fhd = L8(cmp + 4);
                const uint32_t fcsf = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
                if (fhd & 8) break;
                const uint32_t did_sz = did == 3 ? 4u : did, fcs_sz = fcsf == 0 ? single : (1u << fcsf);
                const uint32_t hs = 5 + (single ? 0u : 1u) + did_sz + fcs_sz;
                if (n < hs + 3) break;
                uint32_t q = 5;
                uint64_t window = 0;
                if (!single) { const uint32_t b = L8(cmp + q); q++; const uint32_t wl = 10 + (b >> 3); window = (1ull << wl) + ((1ull << wl) >> 3) * (b & 7); }
                uint32_t frame_dict = 0;
                if (did_sz) { frame_dict = lds_u32(cmp + q) & (did_sz == 4 ? 0xFFFFFFFFu : ((1u << (8 * did_sz)) - 1)); q += did_sz; }
                has_fcs = 1;
                uint64_t fcs64 = 0;
                if (fcsf == 0) { if (single) fcs64 = L8(cmp + q); else has_fcs = 0; }
                else if (fcsf == 1) fcs64 = (lds_u32(cmp + q) & 0xFFFF) + 256;
                else if (fcsf == 2) fcs64 = lds_u32(cmp + q);
                else fcs64 = lds_u64(cmp + q);
                if (single) window = fcs64;
                if (window > (1ull << 27) + 1) break;
                const uint32_t block_max = (uint32_t)(window < kBlockMax ? window : kBlockMax);
                has_ck = (fhd >> 2) & 1;
                if (frame_dict && frame_dict != (with_d && di.formatted ? di.dict_id : 0u)) break;
                if (has_fcs && fcs64 > cap) break;
                fcs = (uint32_t)fcs64;
                const uint32_t bh = lds_u32(cmp + hs) & 0xFFFFFF;
                const uint32_t last = bh & 1;
                btype = (bh >> 1) & 3; bsize = bh >> 3;
                if (!last || btype == 3 || bsize > block_max) break;
                const uint32_t body = btype == 1 ? 1u : bsize;
                if ((uint64_t)hs + 3 + body + (has_ck ? 4u : 0u) != n) break; // exactly one frame of one block, nothing behind it
                b0 = hs + 3;
                if (btype < 2) {
                    if (bsize > cap || (has_fcs && fcs != bsize)) break;
                    done = true; ok = true; // (the content is bsize bytes)
                    break;
                }
                if (bsize < 2) break;
                // ---- literals section header (A.4)
                const uint64_t lb = lds_u64(cmp + b0);
                const uint32_t c0 = (uint32_t)lb & 0xFF, c1 = (uint32_t)(lb >> 8) & 0xFF, c2 = (uint32_t)(lb >> 16) & 0xFF;
                const uint32_t sf = (c0 >> 2) & 3;
                lit_type = c0 & 3;
                uint32_t hl, regen, comp = 0;
                if (lit_type < 2) {
                    if (sf == 0 || sf == 2) { hl = 1; regen = c0 >> 3; }
                    else if (sf == 1) { hl = 2; regen = (c0 >> 4) + (c1 << 4); }
                    else { if (bsize < 3) break; hl = 3; regen = (c0 >> 4) + (c1 << 4) + (c2 << 12); }
                    if (regen > block_max || regen > cap) break;
                    const uint32_t lbody = lit_type == 0 ? regen : 1u;
                    if (hl + lbody >= bsize) break; // (the sequences section needs at least one byte)
                    nlit = regen; streams = 0;
                    lit_off = b0 + hl;
                    seq_off = b0 + hl + lbody; seq_len = bsize - hl - lbody;
                } else {
                    if (bsize < 3) break;
                    if (sf < 2) { hl = 3; const uint32_t v = (uint32_t)lb & 0xFFFFFF; regen = (v >> 4) & 0x3FF; comp = v >> 14; streams = sf ? 4 : 1; }
                    else if (sf == 2) { if (bsize < 4) break; hl = 4; const uint32_t v = (uint32_t)lb; regen = (v >> 4) & 0x3FFF; comp = v >> 18; streams = 4; }
                    else { if (bsize < 5) break; hl = 5; const uint64_t v = lb & 0xFFFFFFFFFFull; regen = (uint32_t)(v >> 4) & 0x3FFFF; comp = (uint32_t)(v >> 22); streams = 4; }
                    if (regen > block_max || regen > cap || regen == 0 || (streams == 4 && regen < 6) || hl + comp >= bsize) break;
                    uint32_t p_off = b0 + hl, rem = comp;
                    if (lit_type == 2) {
                        if (a.tab_bytes < 1024) break; // no room for a private Huffman table (and the weights' scratch) in this launch's slots
                        if (rem < 1) break;
                        const uint32_t hb = L8(cmp + p_off);
                        const uint32_t tl = hb >= 128 ? 1 + ((hb - 127) + 1) / 2 : 1 + hb;
                        if (tl > rem || (hb < 128 && hb < 1)) break;
                        tree_off = p_off; tree_len = tl;
                        p_off += tl; rem -= tl;
                    } else if (!(with_d && di.formatted) || di.huf_log > 11) break; // treeless without a table to reuse (a dictionary's tree of depth 12 -- libzstd takes one, no encoder makes one -- lives in the general path's pair table)
                    if (streams == 1) { s_base = p_off; s_len0 = rem; if (rem == 0) break; }
                    else {
                        if (rem < 10) break;
                        const uint64_t jt = lds_u64(cmp + p_off);
                        const uint32_t l1 = (uint32_t)jt & 0xFFFF, l2 = (uint32_t)(jt >> 16) & 0xFFFF, l3 = (uint32_t)(jt >> 32) & 0xFFFF;
                        if (6 + l1 + l2 + l3 > rem) break;
                        const uint32_t l4 = rem - 6 - l1 - l2 - l3;
                        const uint32_t seg = (regen + 3) / 4;
                        if (3 * seg > regen || !l1 || !l2 || !l3 || !l4) break;
                        s_base = p_off + 6; s_len0 = l1; s_len1 = l2; s_len2 = l3; s_len3 = l4;
                    }
                    nlit = regen;
                    seq_off = b0 + hl + comp; seq_len = bsize - hl - comp;
                }
                ok = true;
            } while (false);
        }
        (void)tree_len;
        if (have && !ok) why = 1;
        bool live = ok && !done; // a compressed block to decode
        const uint32_t stored_ck = (ok && has_ck && n >= 4) ? lds_u32(cmp + n - 4) : 0u; // (read now: the windows will lie over the input)
        const uint32_t rle_byte = (ok && done && n != 0 && btype == 1) ? L8(cmp + b0) : 0u; // (an RLE block's byte; a raw block's bytes are taken from the input in HBM when the file's pass comes)
        const uint32_t job_next = (w0 && early) ? list_entry(g_next) : 0xFFFFFFFFu; // (trip 2)
        SSTAMP(2);

        // =============================== Huffman weights and decode table (one lane per file)
        uint32_t huf_log = di.huf_log, huf_off = dict_off + kDHuf; // treeless: the dictionary's table
        if (w0 && live && lit_type == 2) {
            uint32_t good = 0, maxbits_l = 0, nw_l = 0;
            // (the weights: in the ring -- with a helper wavefront, which is writing the sequence tables' counts there meanwhile, in the table
            //  area's last 256 bytes: a table of more than 9 bits then does not fit beside them, and its file is handed on)
            const uint32_t wts = NW > 1 ? tabo + a.tab_bytes - 256 : ringo, wtab = tabo + kWTab, wnorm = tabo + kWNorm;
            const uint32_t tp = cmp + tree_off; // the tree description
            const uint32_t hb = L8(tp);
            // FSE-coded weights: their normalized counts (the file's first lane), then their decode table (all its lanes)
            uint32_t w_nsym = 0, w_log = 0, w_hdr = 0;
            if (leader && hb < 128) {
                w_hdr = read_ncount_lane(tp + 1, hb, 6, 255, 16, wnorm, w_nsym, w_log);
                if (w_hdr >= hb) w_hdr = 0;
            }
            {
                const int ld = (int)(f * LPF);
                w_hdr = (uint32_t)__shfl((int)w_hdr, ld); w_nsym = (uint32_t)__shfl((int)w_nsym, ld); w_log = (uint32_t)__shfl((int)w_log, ld);
            }
            wsync();
            SSTAMP(13);
            const bool w_tab = build_fse_file<3, (int)LPF>(w_hdr != 0, wtab, wnorm, w_nsym, w_log, shDump + 8 * LPF * f, sub);
            // the weights' bitstream (<= 127 bytes, read backwards) is staged behind 16 zero bytes: fields that reach below the stream's
            // start read zeros there (A.4) without a mask in the loop
            const uint32_t wstage = tabo + kWStage;
            if (w_hdr != 0) {
                if (sub < 2) L64(wstage + 8 * sub) = 0;
                for (uint32_t o = 8 * sub; o < 128; o += 8 * LPF) lds_s64(wstage + 16 + o, lds_u64(tp + 1 + w_hdr + o));
            }
            wsync();
            SSTAMP(14);
            if (leader) {
                uint32_t nw = 0;
                do {
                    if (hb >= 128) { // direct: 4 bits per weight, high nibble first
                        nw = hb - 127;
                        for (uint32_t i = 0; i < nw; i++) {
                            const uint32_t by = L8(tp + 1 + i / 2);
                            L8(wts + i) = (uint8_t)((i & 1) ? (by & 15) : (by >> 4));
                        }
                    } else {
                        if (w_hdr == 0 || !w_tab) break;
                        // Two interleaved states over the backward bitstream; the stream's over-read ends it (A.4).  A round decodes one weight
                        // from each state: both entries are fetched together, both fields come out of one shift of the window, both weights
                        // go out in one store.  Bit positions are LDS bit addresses (G: the read head, bits below it are unread); the window a
                        // round uses was requested by the round before it (a round consumes <= 12 bits of the 56 a window holds below its head).
                        const uint32_t log = w_log, sl = hb - w_hdr;
                        const uint32_t lastb = L8(wstage + 16 + sl - 1);
                        if (lastb == 0) break;
                        const uint32_t Gw0 = 8 * (wstage + 16);
                        uint32_t Gw = Gw0 + (sl - 1) * 8 + (uint32_t)hibit32(lastb);
                        uint32_t bc = 8 * ((Gw >> 3) - 7);
                        uint64_t wc = lds_u64(bc >> 3);
                        uint32_t s1, s2;
                        {
                            const uint32_t y = (uint32_t)(wc >> (Gw - 2 * log - bc)); // (a stream shorter than the two states: zeros from the pad, and the first round ends it)
                            s2 = wtab + 8 * bfe(y, 0, log); s1 = wtab + 8 * bfe(y, log, log);
                            Gw -= 2 * log;
                        }
                        // While neither state can run the stream out in a round (a state takes at most `log` bits): rounds without the
                        // over-read bookkeeping, which is half of a round's instruction slots (a lone wavefront pays every slot in full).
                        while (Gw >= Gw0 + 2 * log && nw <= 250) {
                            const uint64_t e1 = L64(s1), e2 = L64(s2);
                            const uint32_t bn = 8 * ((Gw >> 3) - 7);
                            const uint64_t wn = lds_u64(bn >> 3);
                            const uint32_t h1 = (uint32_t)(e1 >> 32), h2 = (uint32_t)(e2 >> 32);
                            const uint32_t n1 = h1 & 0xFF, n2 = h2 & 0xFF;
                            const uint32_t low = Gw - n1 - n2;
                            const uint32_t y = (uint32_t)(wc >> (low - bc));
                            L16(wts + nw) = (uint16_t)(((h1 >> 16) & 0xFF) | ((h2 >> 8) & 0xFF00));
                            s1 = (uint32_t)e1 + 8 * bfe(y, n2, n1);
                            s2 = (uint32_t)e2 + 8 * bfe(y, 0, n2);
                            Gw = low; nw += 2;
                            wc = wn; bc = bn;
                        }
                        // (ONE exit from the loop, sorted out behind it: with an exit per case the exec-mask bookkeeping was longer than the round)
                        bool over1, over2;
                        do {
                            const uint64_t e1 = L64(s1), e2 = L64(s2);
                            const uint32_t bn = 8 * ((Gw >> 3) - 7);
                            const uint64_t wn = lds_u64(bn >> 3);
                            const uint32_t h1 = (uint32_t)(e1 >> 32), h2 = (uint32_t)(e2 >> 32);
                            const uint32_t n1 = h1 & 0xFF, n2 = h2 & 0xFF;
                            const uint32_t low = Gw - n1 - n2;
                            const uint32_t y = (uint32_t)(wc >> (low - bc));
                            L16(wts + nw) = (uint16_t)(((h1 >> 16) & 0xFF) | ((h2 >> 8) & 0xFF00));
                            s1 = (uint32_t)e1 + 8 * bfe(y, n2, n1);
                            s2 = (uint32_t)e2 + 8 * bfe(y, 0, n2);
                            over1 = (int32_t)(Gw - n1 - Gw0) < 0; over2 = (int32_t)(low - Gw0) < 0;
                            Gw = low; nw += 2;
                            wc = wn; bc = bn;
                        } while (!(over1 | over2) && nw <= 252);
                        const bool fin = over1 | over2;
                        if (!over1 && over2) { L8(wts + nw) = (uint8_t)(L32(s1 + 4) >> 16); nw++; }
                        if (!fin) break;
                    }
                    if (nw < 1 || nw > 255) break;
                    for (uint32_t i = nw; i < (nw & ~7u) + 8; i++) L8(wts + i) = 0; // (the class lanes below read the weights eight at a time: zeros behind the last one)
                    nw_l = nw;
                    good = 1;
                } while (false);
            }
            SSTAMP(15);
            good = (uint32_t)__shfl((int)good, (int)(f * LPF));
            nw_l = (uint32_t)__shfl((int)nw_l, (int)(f * LPF));
            wsync();
            // ---- validation, implied last weight, canonical table (A.4).  Lane = symbol, LPF consecutive symbols per round:
            //   pass 1: a histogram of the weights (LDS adds: two 16-bit counters per word); the weight classes 1..12 then sit on the file's
            //           lanes (class v on lane (v - 1) % LPF): total, table depth, the implied last weight, the classes' start positions
            //           (weight 1 = longest codes first) -- which take the histogram's place as the classes' write heads;
            //   pass 2: a symbol's entries start at its class's write head + (lower lanes of the round with the same weight) x its span
            //           -- a lane mask per class, OR-ed together in LDS as in mzd_l_tables.h's numbering -- and the class's first lane
            //           moves the head on.  Round 3 had a class lane walk all the weights for its symbols, one LDS round trip and a
            //           divergent loop per symbol: 20 K cycles a group.
            // Work memory: the file's share of the dump area (24 + 12 * LPF / 8 bytes of its 8 * LPF).
            if (good) {
                constexpr uint32_t NCL = (12 + LPF - 1) / LPF;
                const uint32_t hist = shDump + 8 * LPF * f, hmask = hist + 24;
                constexpr uint32_t kHistBytes = (24 + (12 * LPF + 7) / 8 + 3) & ~3u; // twelve 16-bit counters, twelve lane masks of LPF bits
                static_assert(kHistBytes <= 8 * LPF, "the histogram and the lane masks stay inside the file's share of the dump area");
                for (uint32_t o = 4 * sub; o < kHistBytes; o += 4 * LPF) L32(hist + o) = 0;
                wsync();
                uint32_t over = 0;
                for (uint32_t s0 = 0; s0 < nw_l; s0 += LPF) {
                    const uint32_t s = s0 + sub;
                    const uint32_t w = s < nw_l ? L8(wts + s) : 0u;
                    over |= w > 12 ? 1u : 0u;
                    if (w - 1 < 12) lds_add32(hist + 4 * ((w - 1) >> 1), 1u << (16 * ((w - 1) & 1)));
                }
                over = __ballot(over != 0) >> (f * LPF) & ((1ull << LPF) - 1) ? 1u : 0u; // a weight above 12 anywhere in the file
                wsync();
                uint32_t cnt[NCL];
#pragma unroll
                for (uint32_t j = 0; j < NCL; j++) { const uint32_t v = sub + 1 + j * LPF; cnt[j] = v <= 12 ? (L32(hist + 4 * ((v - 1) >> 1)) >> (16 * ((v - 1) & 1))) & 0xFFFF : 0u; }
                uint32_t part = 0;
#pragma unroll
                for (uint32_t j = 0; j < NCL; j++) { const uint32_t v = sub + 1 + j * LPF; part += v <= 12 ? cnt[j] << (v - 1) : 0u; }
                uint32_t tot = part; // sum over the file's lanes
                tot += seg_shr<1, LPF>(tot, sub); tot += seg_shr<2, LPF>(tot, sub);
                if (LPF > 4) tot += seg_shr<4, LPF>(tot, sub);
                if (LPF > 8) tot += seg_shr<8, LPF>(tot, sub);
                const uint32_t total = bcast<LPF - 1, LPF>(tot);
                bool g2 = over == 0 && total != 0;
                const uint32_t maxbits = g2 ? (uint32_t)hibit32(total) + 1 : 1u;
                g2 = g2 && maxbits <= 11 && (2u << maxbits) + (NW > 1 ? 256u : 0u) <= a.tab_bytes; // (a table that does not fit the slot, a tree of depth 12: the general path takes the file)
                const uint32_t left = (1u << maxbits) - total;
                g2 = g2 && (left & (left - 1)) == 0;
                const uint32_t wl = (uint32_t)hibit32(left | 1u) + 1; // the implied last weight, of symbol nw
                if (g2 && leader) L8(wts + nw_l) = (uint8_t)wl;
#pragma unroll
                for (uint32_t j = 0; j < NCL; j++) if (sub + 1 + j * LPF == wl) cnt[j]++;
                const uint32_t r1 = bcast<0, LPF>(cnt[0]); // symbols of weight 1: an even number, at least two
                g2 = g2 && r1 >= 2 && (r1 & 1) == 0;
                // start positions: exclusive scan of the class sizes, class after class
                {
                    uint32_t carry = 0;
#pragma unroll
                    for (uint32_t j = 0; j < NCL; j++) {
                        const uint32_t v = sub + 1 + j * LPF;
                        const uint32_t sz = v <= 12 ? cnt[j] << (v - 1) : 0u;
                        uint32_t inc = sz;
                        inc += seg_shr<1, LPF>(inc, sub); inc += seg_shr<2, LPF>(inc, sub);
                        if (LPF > 4) inc += seg_shr<4, LPF>(inc, sub);
                        if (LPF > 8) inc += seg_shr<8, LPF>(inc, sub);
                        if (v <= 12) L16(hist + 2 * (v - 1)) = (uint16_t)(carry + inc - sz); // the class's write head (the histogram has been read)
                        carry += bcast<LPF - 1, LPF>(inc);
                    }
                    g2 = g2 && carry == (1u << maxbits); // (also catches weights above maxbits)
                }
                SSTAMP(16);
                wsync();
                if (g2) {
                    for (uint32_t s0 = 0; s0 <= nw_l; s0 += LPF) {
                        const uint32_t s = s0 + sub;
                        const uint32_t w = s <= nw_l ? L8(wts + s) : 0u;
                        const bool on = w != 0;
                        const uint32_t c = on ? w - 1 : 0u; // (span = 1 << c entries)
                        const uint32_t mo = hmask + 4 * ((c * LPF) >> 5), sh = (c * LPF) & 31;
                        if (on) lds_or32(mo, (1u << sub) << sh);
                        wsync();
                        if (on) {
                            const uint32_t lm = (L32(mo) >> sh) & ((1u << LPF) - 1);
                            const uint32_t lower = (uint32_t)__builtin_popcount(lm & ((1u << sub) - 1));
                            const uint32_t head = L16(hist + 2 * c);
                            const uint32_t at = tabo + 2 * (head + (lower << c));
                            const uint32_t e = (s << 8) | (maxbits - c), e2 = e | (e << 16); // (an entry: code length | symbol << 8 -- the length is the window's shift count as it is loaded)
                            asm volatile("" ::: "memory");
                            if (c == 0) L16(at) = (uint16_t)e;
                            else if (c == 1) L32(at) = e2;
                            else if (c == 2) L64(at) = (uint64_t)e2 | ((uint64_t)e2 << 32);
                            else { const V16 q = {(uint64_t)e2 | ((uint64_t)e2 << 32), (uint64_t)e2 | ((uint64_t)e2 << 32)}; for (uint32_t o = 0; o < (2u << c); o += 16) lds_sv16(at + o, q); }
                            if (lower == 0) L16(hist + 2 * c) = (uint16_t)(head + ((uint32_t)__builtin_popcount(lm) << c));
                            lds_xor32(mo, (1u << sub) << sh); // (the mask is clean again for the next round)
                        }
                        wsync();
                    }
                    maxbits_l = maxbits;
                }
                good = g2 ? 1u : 0u;
            }
            { const uint64_t gm = __ballot(good != 0); good = ((gm >> (f * LPF)) & ((1ull << LPF) - 1)) == ((1ull << LPF) - 1) ? 1u : 0u; } // (uniform in the file)
            huf_log = maxbits_l; huf_off = tabo;
            if (!good) { ok = false; live = false; why = 2; }
        }
        wsync();
        Jn = job_entry(job_next); // (trip 3)
        SSTAMP(3);

        // =============================== Huffman streams -> the literal scratch (lane = (file, stream))
        {
            uint32_t lit_bad = 0;
            if (w0 && live && lit_type >= 2 && sub < streams) {
                const uint32_t st = sub;
                const uint32_t seg = (nlit + 3) / 4;
                const uint32_t sl = streams == 1 ? s_len0 : (st == 0 ? s_len0 : (st == 1 ? s_len1 : (st == 2 ? s_len2 : s_len3)));
                const uint32_t rel = streams == 1 ? 0u : (st == 0 ? 0u : (st == 1 ? s_len0 : (st == 2 ? s_len0 + s_len1 : s_len0 + s_len1 + s_len2)));
                const uint32_t lbase = cmp + s_base + rel;
                const uint32_t nsym = streams == 1 ? nlit : (st < 3 ? seg : nlit - 3 * seg);
                uint8_t* const out = lit_g + (streams == 1 ? 0u : st * seg);
                const uint32_t Lg = huf_log, tab = huf_off;
                bool good = sl != 0;
                const uint32_t last = good ? L8(lbase + sl - 1) : 1u;
                good = good && last != 0;
                if (good) {
                    int32_t h = (int32_t)((sl - 1) * 8) + hibit32(last); // unread bits
                    // the 57..64 bits below the read head, MSB-aligned; bits below the stream's start read as zero (A.4: the last symbols
                    // may peek past it).  Every stream is preceded by >= 8 bytes of its file.
                    auto window = [&](int32_t hh) -> uint64_t {
                        int32_t b = (hh - 1) >> 3;
                        b = b < -1 ? -1 : b;
                        uint64_t w = lds_u64(lbase + (uint32_t)(b + 9) - 16);
                        w <<= (uint32_t)(8 * (b + 1) - hh) & 63;
                        const uint64_t keep = hh >= 64 ? ~0ull : (hh <= 0 ? 0ull : ~0ull << (64 - hh));
                        return w & keep;
                    };
                    uint32_t k = 0;
                    const uint32_t shH = 32 - Lg; // (Lg <= 11: a code's table index comes out of the window's high word)
                    auto look = [&](uint64_t c) -> uint32_t { return L16(tab + 2 * ((uint32_t)(c >> 32) >> shH)); };
                    // Four symbols (<= 44 bits) per window, one 4-byte store.  While at least 64 unread bits are left the window is a plain
                    // unaligned read shifted into place; the entries' low bytes (the lengths) add up without carrying into the symbols.
                    for (; k + 4 <= nsym && h >= 64; k += 4) {
                        const uint32_t b = (uint32_t)(h - 1) >> 3;
                        uint64_t cur = lds_u64(lbase + b - 7) << ((8 * (b + 1) - (uint32_t)h) & 63);
                        const uint32_t e0 = look(cur); cur <<= (e0 & 63);
                        const uint32_t e1 = look(cur); cur <<= (e1 & 63);
                        const uint32_t e2 = look(cur); cur <<= (e2 & 63);
                        const uint32_t e3 = look(cur);
                        h -= (int32_t)((e0 + e1 + e2 + e3) & 0xFF);
                        gs32(out + k, __builtin_amdgcn_perm(e1, e0, 0x0c0c0501u) | (__builtin_amdgcn_perm(e3, e2, 0x0c0c0501u) << 16));
                    }
                    uint64_t cur = window(h);
                    for (; k + 4 <= nsym; k += 4) { // the stream's last bytes: the window is masked below the stream's start
                        const uint32_t e0 = look(cur); cur <<= (e0 & 63);
                        const uint32_t e1 = look(cur); cur <<= (e1 & 63);
                        const uint32_t e2 = look(cur); cur <<= (e2 & 63);
                        const uint32_t e3 = look(cur);
                        h -= (int32_t)((e0 + e1 + e2 + e3) & 0xFF);
                        gs32(out + k, (e0 >> 8) | (e1 & 0xFF00) | ((e2 & 0xFF00) << 8) | ((e3 & 0xFF00) << 16));
                        cur = window(h);
                    }
                    for (; k < nsym; k++) {
                        const uint32_t e0 = look(cur);
                        gs8(out + k, e0 >> 8);
                        cur <<= (e0 & 63);
                        h -= (int32_t)(e0 & 0xFF);
                    }
                    good = h == 0; // consumed exactly
                }
                if (!good) lit_bad = 1;
            }
            const uint64_t badm = __ballot(lit_bad != 0); // a failed stream condemns its file
            if ((badm >> (f * LPF)) & ((1ull << LPF) - 1)) { ok = false; live = false; why = 3; }
            // RLE literals: the scratch is filled with the byte
            if (w0 && live && lit_type == 1) { const uint32_t v = L8(cmp + lit_off) * 0x01010101u; for (uint32_t k = 4 * sub; k < nlit; k += 4 * LPF) gs32(lit_g + k, v); } // (slack past nlit)
        }
        wsync();
        prefetch(Jn, pfn); // (trip 4: in flight from here to the next group's start)
        SSTAMP(4);

        // =============================== sequences section header (one lane per file): nbSeq, modes, normalized counts
        uint32_t nseq = 0, bs_off = 0, bs_len = 0;
        uint32_t tabL = 0, tabO = 0, tabM = 0, alL = 0, alO = 0, alM = 0; // LDS offset of each table, its log
        uint32_t modes3 = 0, rle_syms = 0, nsyms = 0;
        uint32_t sq_good = 0;
        if (live && (NW == 1 || !w0)) { // (with a helper wavefront: its work, beside the Huffman phases above)
            uint32_t& good = sq_good;
            if (leader) {
                do {
                    const uint32_t sp = cmp + seq_off;
                    const uint64_t w = lds_u64(sp);
                    uint32_t p = 1;
                    nseq = (uint32_t)w & 0xFF;
                    if (nseq > 0x7F) {
                        if (nseq == 0xFF) { if (p + 2 > seq_len) break; nseq = ((uint32_t)(w >> 8) & 0xFFFF) + 0x7F00; p = 3; }
                        else { if (p + 1 > seq_len) break; nseq = ((nseq - 0x80) << 8) + ((uint32_t)(w >> 8) & 0xFF); p = 2; }
                    }
                    if (nseq == 0) { good = p == seq_len; break; }
                    if (nseq > cap / 3 + 1 || p + 1 > seq_len) break; // (every match is >= 3 bytes)
                    const uint32_t modes = (uint32_t)(w >> (8 * p)) & 0xFF;
                    p++;
                    if (modes & 3) break;
                    bool tbad = false;
                    uint32_t used_entries = 0;
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        const uint32_t m = (modes >> (6 - 2 * t)) & 3;
                        const int max_log = t == 1 ? 8 : 9, max_sym = t == 0 ? 35 : (t == 1 ? 31 : 52);
                        const uint32_t noff = ringo + (t == 0 ? 0u : (t == 1 ? 72u : 136u));
                        uint32_t tab = 0, al = 0, rs = 0, ns = 0;
                        if (m == 0) { // predefined distribution: built like a described one
                            ns = t == 0 ? 36u : (t == 1 ? 29u : 53u); al = t == 1 ? 5u : 6u;
                            for (uint32_t s = 0; s < ns; s++) L16s(noff + 2 * s) = t == 0 ? LL_DEF[s] : (t == 1 ? OF_DEF[s] : ML_DEF[s]);
                            tab = tabo + 8 * used_entries; used_entries += 1u << al;
                        } else if (m == 1) {
                            if (p + 1 > seq_len) { tbad = true; break; }
                            rs = L8(sp + p); p++;
                            if (rs > (uint32_t)max_sym) { tbad = true; break; }
                            tab = tabo + 8 * used_entries; used_entries += 2; al = 0; // (two entries: every table starts 16-byte aligned)
                        } else if (m == 2) {
                            if (p >= seq_len) { tbad = true; break; }
                            const uint32_t used = read_ncount_lane(sp + p, seq_len - p, max_log, max_sym, max_sym + 1, noff, ns, al);
                            if (used == 0) { tbad = true; break; }
                            p += used;
                            tab = tabo + 8 * used_entries; used_entries += 1u << al;
                        } else {
                            if (!(with_d && di.formatted)) { tbad = true; break; }
                            tab = dict_off + (t == 0 ? kDLL : (t == 1 ? kDOF : kDML)); al = t == 0 ? di.al[0] : (t == 1 ? di.al[1] : di.al[2]);
                        }
                        if (t == 0) { tabL = tab; alL = al; } else if (t == 1) { tabO = tab; alO = al; } else { tabM = tab; alM = al; }
                        modes3 |= m << (2 * t); rle_syms |= rs << (8 * t); nsyms |= ns << (8 * t);
                    }
                    if (tbad || used_entries * 8 > ((a.tab_bytes + seq_off) & ~15u)) break; // (the table area and the input in front of the sequences section: dead by now)
                    if (p >= seq_len) break; // the bitstream needs at least one byte
                    bs_off = seq_off + p; bs_len = seq_len - p;
                    good = 1;
                } while (false);
            }
            if constexpr (NW == 1) {
                const int ld = (int)(f * LPF);
                good = (uint32_t)__shfl((int)good, ld);
                nseq = (uint32_t)__shfl((int)nseq, ld); bs_off = (uint32_t)__shfl((int)bs_off, ld); bs_len = (uint32_t)__shfl((int)bs_len, ld);
                tabL = (uint32_t)__shfl((int)tabL, ld); tabO = (uint32_t)__shfl((int)tabO, ld); tabM = (uint32_t)__shfl((int)tabM, ld);
                alL = (uint32_t)__shfl((int)alL, ld); alO = (uint32_t)__shfl((int)alO, ld); alM = (uint32_t)__shfl((int)alM, ld);
                modes3 = (uint32_t)__shfl((int)modes3, ld); rle_syms = (uint32_t)__shfl((int)rle_syms, ld); nsyms = (uint32_t)__shfl((int)nsyms, ld);
                if (!good) { ok = false; live = false; nseq = 0; why = 4; }
            }
        }
        if constexpr (NW > 1) { // the parsed header crosses over in LDS (second barrier of the group); every lane of the file reads it
            const uint32_t xo = kShSeqRec + 64 * f + 16;
            if (!w0) {
                if (leader) {
                    lds_sv16(xo, V16{(uint64_t)sq_good | ((uint64_t)nseq << 32), (uint64_t)bs_off | ((uint64_t)bs_len << 32)});
                    lds_sv16(xo + 16, V16{(uint64_t)tabL | ((uint64_t)tabO << 32), (uint64_t)tabM | ((uint64_t)(alL | (alO << 8) | (alM << 16)) << 32)});
                    lds_sv16(xo + 32, V16{(uint64_t)modes3 | ((uint64_t)rle_syms << 32), (uint64_t)nsyms});
                }
                __syncthreads();
            } else {
                __syncthreads();
                if (live) {
                    const V16 x0 = lds_v16(xo), x1 = lds_v16(xo + 16), x2 = lds_v16(xo + 32);
                    sq_good = (uint32_t)x0.a; nseq = (uint32_t)(x0.a >> 32); bs_off = (uint32_t)x0.b; bs_len = (uint32_t)(x0.b >> 32);
                    tabL = (uint32_t)x1.a; tabO = (uint32_t)(x1.a >> 32); tabM = (uint32_t)x1.b;
                    const uint32_t als = (uint32_t)(x1.b >> 32); alL = als & 0xFF; alO = (als >> 8) & 0xFF; alM = (als >> 16) & 0xFF;
                    modes3 = (uint32_t)x2.a; rle_syms = (uint32_t)(x2.a >> 32); nsyms = (uint32_t)x2.b;
                    if (!sq_good) { ok = false; live = false; nseq = 0; why = 4; }
                }
            }
        }
        wsync();
        SSTAMP(5);
      if (w0) { // (the helper is through with the group: it waits at the group's last barrier)

        // =============================== FSE decode tables (every table by all the lanes of its file: mzd_l_tables.h)
        {
            const bool tables = live && nseq != 0;
            const uint32_t mL = modes3 & 3, mO = (modes3 >> 2) & 3, mM = (modes3 >> 4) & 3;
            if (tables && sub < 3) { // a table of one symbol (RLE mode)
                const int t = (int)sub;
                const uint32_t m = (modes3 >> (2 * t)) & 3;
                if (m == 1) {
                    const uint32_t tab = t == 0 ? tabL : (t == 1 ? tabO : tabM);
                    const uint32_t s = (rle_syms >> (8 * t)) & 0xFF;
                    const uint32_t extra = t == 0 ? L32(kShLL + 4 * s) >> 24 : (t == 1 ? s : L32(kShML + 4 * s) >> 24);
                    L64(tab) = fse_entry(tab, 0, 0, s, extra);
                }
            }
            const uint32_t work = shDump + 8 * LPF * f; // (the lane masks: the file's share of the dump area, idle until the execution)
            bool tg = build_fse_file<0, (int)LPF>(tables && (mL == 0 || mL == 2), tabL, ringo, nsyms & 0xFF, alL, work, sub);
            tg &= build_fse_file<1, (int)LPF>(tables && (mO == 0 || mO == 2), tabO, ringo + 72, (nsyms >> 8) & 0xFF, alO, work, sub);
            tg &= build_fse_file<2, (int)LPF>(tables && (mM == 0 || mM == 2), tabM, ringo + 136, (nsyms >> 16) & 0xFF, alM, work, sub);
            if (!tg) { ok = false; live = false; nseq = 0; why = 5; }
        }
        wsync();
        SSTAMP(6);

        // =============================== FSE state walk -> field extraction -> the sequence scratch, LPF sequences at a time
        // Bit positions are LDS bit addresses (8 * byte offset + bit): G = the read head, bits below it are unread.
        // Per step of LPF sequences: the walk, one lane per file, branch-free -- the state as it stands is the record of a
        // sequence --, then lane = sequence: the fields from the records, 8 bytes per sequence to the scratch (coalesced).
        bool bad = false;
        uint32_t nrun = 0;
        {
            const uint32_t G0 = 8 * (cmp + bs_off); // the stream's bit 0
            uint32_t Gh = G0, aL = 0, aM = 0, aO = 0;
            if (live && nseq) {
                const uint32_t lastb = L8(cmp + bs_off + bs_len - 1);
                bad = lastb == 0;
                Gh = G0 + (bs_len - 1) * 8 + (uint32_t)hibit32(lastb | 1u);
                const uint32_t need = alL + alO + alM; // <= 26
                bad |= Gh - G0 < need;
                const uint32_t e = (Gh + 7) >> 3;
                const uint64_t X = lds_u64(e - 8);
                const uint32_t Y = (uint32_t)(X >> ((Gh - 8 * e + 64 - need) & 63)); // the three initial states: LL, OF, ML from the top
                aM = tabM + 8 * bfe(Y, 0, alM); aO = tabO + 8 * bfe(Y, alM, alO); aL = tabL + 8 * bfe(Y, alM + alO, alL);
                Gh -= need;
            }
            nrun = (live && !bad) ? nseq : 0u;
#if defined(MZD_SMALL_STAMPS) && !defined(MZD_SS_NOACC)
            uint64_t tw_ = 0, tp_ = 0, t0_ = __builtin_readcyclecounter(), t1_ = 0;
#define GSTAMP(acc) do { t1_ = __builtin_readcyclecounter(); acc += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define GSTAMP(acc)
#endif
            // (WS sequences per call of the walk, at least 16 whatever LPF is -- the ring holds 16 records --: a call's fixed cost, the
            //  hand-over of the state into the quad and back and the checks around it, was as long as four of its steps)
            constexpr uint32_t WS = LPF < 16 ? 16u : LPF;
            for (uint32_t c0 = 0; __any(c0 < nrun && !bad); c0 += WS) {
                const bool act = c0 < nrun && !bad;
                // ---- the walk
                uint32_t wbad = 0;
                bool done_asm = false;
                if (lds_at_zero) { // the hot form, on the first quad of every file whose step is a full one (every sequence followed by a state update)
                    if (act && sub < 4 && c0 + WS < nrun) {
                        auto quad = [](uint32_t v, auto ctrl_c) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl_c)::value, 0xF, 0xF, false); };
                        const std::integral_constant<int, 0x00> q0{}; const std::integral_constant<int, 0x55> q1{}; const std::integral_constant<int, 0xAA> q2{};
                        const uint32_t bL = quad(aL, q0), bM = quad(aM, q0), bO = quad(aO, q0), bG = quad(Gh, q0); // the leader's state
                        uint32_t A = sel(m_eq(sub, 0), bL, sel(m_eq(sub, 1), bM, sel(m_eq(sub, 2), bO, kShWalkDummy)));
                        uint32_t xG = bG - 32;
                        int32_t slack = 64;
                        walk_asm<(int)WS>(A, xG, slack, ringo + 4 * sub);
                        const uint32_t nM = quad(A, q1), nO = quad(A, q2);
                        if (leader && slack >= 0) { aL = A; aM = nM; aO = nO; Gh = xG + 32; done_asm = true; } // (else: a sequence wider than 32..63 bits met the window's edge)
                    }
                }
                if (act && leader) {
                    auto record = [&](uint32_t k) { lds_sv16(ringo + 16 * k, V16{(uint64_t)aL | ((uint64_t)aM << 32), (uint64_t)aO | ((uint64_t)(Gh - 32) << 32)}); };
                    auto step = [&]() {
                        const uint32_t e = (Gh + 7) >> 3;
                        const uint64_t X = lds_u64(e - 8); // the 57..64 bits below the read head
                        const uint64_t EL = L64(aL), EM = L64(aM), EO = L64(aO);
                        const uint32_t hL = (uint32_t)(EL >> 32), hM = (uint32_t)(EM >> 32), hO = (uint32_t)(EO >> 32);
                        const uint32_t tot = ((hL + hM + hO) >> 8) & 0xFF; // every bit this sequence consumes
                        const int32_t s = (int32_t)(Gh - 8 * e + 64) - (int32_t)tot;
                        wbad |= (uint32_t)s >> 31; // wider than the window: handed on
                        const uint32_t Y = (uint32_t)(X >> (s & 63));
                        // fresh state bits sit at the bottom of what the sequence consumes: OF lowest, then ML, then LL
                        aO = (uint32_t)EO + 8 * bfe(Y, 0, hO);
                        aM = (uint32_t)EM + 8 * bfe(Y, hO, hM);
                        aL = (uint32_t)EL + 8 * bfe(Y, hO + hM, hL);
                        Gh -= tot;
                    };
                    if (c0 + WS < nrun) { // every sequence of the step is followed by a state update
                        if (!done_asm) {
                            for (uint32_t k = 0; k < WS; k++) { record(k); step(); }
                        }
                    } else { // the file's last step
                        const uint32_t cnt = nrun - c0;
                        for (uint32_t k = 0; k < cnt; k++) { record(k); if (k + 1 < cnt) step(); }
                    }
                    wbad |= (uint32_t)(Gh - G0) >> 31; // over-read (bounded: <= WS * 89 bits below the stream, inside the slot)
                }
                wsync();
                GSTAMP(tw_);
                // ---- lane = sequence c0 + e0 + sub
                uint32_t pbad = wbad;
#pragma unroll 1
                for (uint32_t e0 = 0; e0 < WS; e0 += LPF)
                if (act && c0 + e0 + sub < nrun) {
                    const V16 r16 = lds_v16(ringo + 16 * (e0 + sub));
                    const struct { uint32_t x, y, z, w; } r = {(uint32_t)r16.a, (uint32_t)(r16.a >> 32), (uint32_t)r16.b, (uint32_t)(r16.b >> 32)};
                    const uint32_t hL = L32(r.x + 4), hM = L32(r.y + 4), hO = L32(r.z + 4);
                    const uint32_t xL = hL >> 24, xM = hM >> 24, xO = hO >> 24;
                    const uint32_t cL = (hL >> 16) & 0xFF, cM = (hM >> 16) & 0xFF, cO = (hO >> 16) & 0xFF;
                    const uint32_t xs = xL + xM + xO;
                    const uint32_t tL = r.w + 32 - xs; // bottom of the LL field (the record holds the read head - 32)
                    if ((int32_t)(tL - G0) < 0 || xs + (tL & 7) > 64) pbad = 1;
                    uint64_t W = lds_u64(tL >> 3) >> (tL & 7);
                    const uint32_t vL = (uint32_t)W & (uint32_t)((1ull << xL) - 1); W >>= xL;
                    const uint32_t vM = (uint32_t)W & (uint32_t)((1ull << xM) - 1); W >>= xM;
                    const uint32_t vO = (uint32_t)(W & ((1ull << xO) - 1));
                    const uint32_t ofv = (1u << cO) + vO;
                    const uint32_t ml = (L32(kShML + 4 * cM) & 0xFFFFFF) + vM;
                    const uint32_t ll = (L32(kShLL + 4 * cL) & 0xFFFFFF) + vL;
                    if (c0 + e0 + sub + 1 == nrun && tL != G0) pbad = 1; // the bitstream must be consumed exactly
                    if (((ll | ml) >> 14) | (cO > 18)) pbad = 1;     // (cannot be right for a window of <= 8 KiB and a dictionary of <= 128 KiB; keeps the packed fields in range)
                    const bool small = ll < 127 && ml - 3 < 63; // (every match length is >= 3)
                    gs32(seq_g + 4 * (c0 + e0 + sub), (small ? ll | ((ml - 3) << 7) : 127u) | (ofv << 13));
                    if (!small) gs64(seq8_g + 8 * (c0 + e0 + sub), (uint64_t)(ll | (ml << 14)) | ((uint64_t)ofv << 32));
                }
                {
                    const uint64_t pm = __ballot(pbad != 0);
                    if ((pm >> (f * LPF)) & ((1ull << LPF) - 1)) bad = true;
                }
                wsync();
                GSTAMP(tp_);
            }
#if defined(MZD_SMALL_STAMPS) && !defined(MZD_SS_NOACC)
            if (a.stamps && w0 && wv == 0 && blockIdx.x == 0 && lane == 0 && first_group) { a.stamps[10] = tw_; a.stamps[11] = tp_; }
#endif
            if (bad) { ok = false; live = false; nrun = 0; why = 6; }
        }
        // what this wavefront stored to the scratch is read back by its other lanes: the stores have to have left (same CU: same L1)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#ifdef MZD_EXP_AGENT_ACQ
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
        // =============================== what the execution needs of a file: its record
        if (leader) {
            const uint32_t fl = (have ? 1u : 0u) | (ok ? 2u : 0u) | (live ? 4u : 0u) | (done ? 8u : 0u) | (btype << 4) | (lit_type << 6) | (has_fcs << 8) | (has_ck << 9) |
                                (with_d ? 1u << 10 : 0u) | (n != 0 ? 1u << 11 : 0u) | (why << 16);
            const uint32_t ro = shRec + 32 * f;
            lds_sv16(ro, V16{(uint64_t)fl | ((uint64_t)job << 32), (uint64_t)nlit | ((uint64_t)nrun << 32)});
            lds_sv16(ro + 16, V16{(uint64_t)fcs | ((uint64_t)stored_ck << 32), (uint64_t)((bsize & 0xFFFF) | (b0 << 16)) | ((uint64_t)((lit_off & 0xFFFF) | (rle_byte << 16)) << 32)});
        }
        wsync();
        SSTAMP(7);
      } // (w0)
      }
#ifdef MZD_SMALL_STAMPS
      if (first_group) wg_te_ = __builtin_amdgcn_s_memrealtime(); // (the entropy phases end)
#endif
      if (w0)
      for (uint32_t pass = 0; pass < NX; pass++) {
        // =============================== an execution pass: XG of the group's files, XLPF lanes each (the names of the entropy phases, for this pass's files)
        constexpr uint32_t LPF = XLPF;
        const uint32_t f = lane / LPF, sub = lane % LPF;
        const bool leader = sub == 0;
        const uint32_t gfile = pass * XG + f; // the file's place in its group
        const uint32_t fidx = g * G + gfile;
        (void)fidx;
        const V16 r0_ = lds_v16(shRec + 32 * gfile), r1_ = lds_v16(shRec + 32 * gfile + 16);
        const uint32_t rfl = (uint32_t)r0_.a, job = (uint32_t)(r0_.a >> 32), nlit = (uint32_t)r0_.b;
        uint32_t nrun = (uint32_t)(r0_.b >> 32);
        const uint32_t fcs = (uint32_t)r1_.a, stored_ck = (uint32_t)(r1_.a >> 32);
        const uint32_t bsize = (uint32_t)r1_.b & 0xFFFF, b0 = ((uint32_t)r1_.b >> 16) & 0xFFFF, lit_off = (uint32_t)(r1_.b >> 32) & 0xFFFF, rle_byte = (uint32_t)(r1_.b >> 48);
        const bool have = (rfl & 1) != 0, done = (rfl & 8) != 0, with_d = DICT && (rfl & (1u << 10)) != 0;
        bool ok = (rfl & 2) != 0, live = (rfl & 4) != 0;
        const uint32_t btype = (rfl >> 4) & 3, lit_type = (rfl >> 6) & 3, has_fcs = (rfl >> 8) & 1, has_ck = (rfl >> 9) & 1, n = (rfl >> 11) & 1; // (n: the file is not empty)
        uint32_t why = rfl >> 16;
        (void)why;
        const uint8_t* src = nullptr; uint8_t* dst = nullptr; uint8_t* dst2 = nullptr; uint32_t cap = 0;
        if (have) { const DevJob& dj = a.jobs[job]; src = dj.src; dst = dj.dst; dst2 = dj.dst2; cap = (uint32_t)dj.dst_cap; } // (L2: the entropy phases read the entry)
        const uint32_t outo = slots0 + f * (G != XG ? a.out_bytes : stride); // the file's output window
        uint8_t* const lit_g = a.scratch + (size_t)(wslot * G + gfile) * ((size_t)a.lit_stride + 12u * (size_t)a.seq_cap);
        uint8_t* const seq_g = lit_g + a.lit_stride;
        uint8_t* const seq8_g = seq_g + 4u * (size_t)a.seq_cap;
        const uint8_t* const lit_p = lit_type == 0 ? src + lit_off : lit_g; // the literals: raw where the input has them (HBM), else the scratch
        uint32_t out_len = (ok && done && n != 0) ? bsize : 0u; // the decoded file: out_len bytes at LDS offset res_off
        const uint32_t res_off = outo;
        // raw / RLE blocks: the window takes the block's bytes from the input (HBM: its image in LDS may lie under another file's window) / is filled with the byte
        if (ok && done && n != 0) {
            if (btype == 0) { for (uint32_t k = 16 * sub; k < bsize; k += 16 * LPF) { const V16 v = gv16(src + b0 + k); lds_sv16(outo + k, v); } } // (inputs are readable 16 bytes past their end; the window has 16 bytes of slack)
            else { const uint32_t v = rle_byte * 0x01010101u; for (uint32_t k = 4 * sub; k < bsize; k += 4 * LPF) L32(outo + k) = v; }
        }
        wsync();

        // =============================== execution: the slot is the file's output window now, LPF sequences at a time, lane = sequence.
        // Software pipeline over the steps: the sequences of step c + 2 and the literals of step c + 1 are in flight (HBM scratch, L2)
        // while step c is executed.
        //   A(c): fields, positions by scans over the file's lanes, what can be checked without the offsets, the literal requests;
        //   B(c): repeat offsets (A.5) by a scan over references; the literals and the matches that lie wholly in the dictionary depend
        //         on nothing: every lane stores its own; the matches inside the window in rounds, LDS -> LDS.
        if (live) {
            const uint32_t dict_len = with_d ? di.content_len : 0u;
            const uint8_t* const dict_end = with_d ? di.content + di.content_len : nullptr;
            uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
            if (with_d && di.formatted) { rep0 = di.rep[0]; rep1 = di.rep[1]; rep2 = di.rep[2]; }
            const uint32_t dump = shDump + 8 * lane; // where the stores of lanes that have nothing to store go
            bool xbad = false;
            uint32_t lpos = 0, opos = 0; // literals consumed / output produced before the step that A() looks at
            // the literals: from the scratch (raw ones: from the input) into the TAIL of the window (cap - nlit ..), 16 bytes per lane.
            // The write head of the execution never passes the literal read head -- what is still to be written is at least the
            // literals still to be read -- so literals and output share the window.
            const uint32_t lit_base = outo + cap - nlit;
            // (four trips in flight: one after the other, each waited for, was five round trips to L2 in front of every pass)
            for (uint32_t q = 16 * sub; q < nlit; q += 4 * 16 * LPF) { // (<= 15 bytes past cap: the window's slack)
                constexpr uint32_t S = 16 * LPF;
                const bool h1 = q + S < nlit, h2 = q + 2 * S < nlit, h3 = q + 3 * S < nlit;
                const V16 v0 = gv16(lit_p + q), v1 = h1 ? gv16(lit_p + q + S) : V16{0, 0}, v2 = h2 ? gv16(lit_p + q + 2 * S) : V16{0, 0}, v3 = h3 ? gv16(lit_p + q + 3 * S) : V16{0, 0};
                lds_s64(lit_base + q, v0.a); lds_s64(lit_base + q + 8, v0.b);
                if (h1) { lds_s64(lit_base + q + S, v1.a); lds_s64(lit_base + q + S + 8, v1.b); }
                if (h2) { lds_s64(lit_base + q + 2 * S, v2.a); lds_s64(lit_base + q + 2 * S + 8, v2.b); }
                if (h3) { lds_s64(lit_base + q + 3 * S, v3.a); lds_s64(lit_base + q + 3 * S + 8, v3.b); }
            }
            wsync();
            // a step's records: the 4-byte ones are requested three steps ahead; a step ahead of their use (they have arrived) the full
            // records of the sequences that say so are requested from the second array; spelled as round 4's 8-byte record for what follows
            auto load_rec4 = [&](uint32_t c0) -> uint32_t { return c0 + sub < nrun ? gu32(seq_g + 4 * (c0 + sub)) : 127u; }; // (past the end: "see the second array", where load_rec8 puts the empty sequence)
            auto load_rec8 = [&](uint32_t c0, uint32_t r4) -> uint64_t { // (wave-uniform skip: most steps hold no such sequence)
                const bool big = c0 + sub < nrun && (r4 & 127u) == 127u;
                if (!__ballot(big)) return 0ull;
                return big ? gu64(seq8_g + 8 * (c0 + sub)) : 0ull;
            };
            auto widen = [&](uint32_t r4, uint64_t r8) -> uint64_t { // (selects, no branches)
                const lmask big = m_eq(r4 & 127u, 127u);
                const uint32_t lo = (r4 & 127u) | ((((r4 >> 7) & 63u) + 3u) << 14), hi = r4 >> 13;
                return (uint64_t)sel(big, (uint32_t)r8, lo) | ((uint64_t)sel(big, (uint32_t)(r8 >> 32), hi) << 32);
            };
            struct StepA { uint32_t ll, ml, w0, lp, op, chunk_l, chunk_t; };
            auto stage_a = [&](uint64_t rec, StepA& A) {
                uint32_t ll = (uint32_t)rec & 0x3FFF, ml = ((uint32_t)rec >> 14) & 0x3FFF;
                const uint32_t ofv = (uint32_t)(rec >> 32);
                const uint32_t il = seg_scan_add<LPF>(ll, sub), it = seg_scan_add<LPF>(ll + ml, sub);
                A.lp = lpos + il - ll; A.op = opos + it - ll - ml; // this sequence's literals / its output
                const uint64_t pm = __ballot((A.lp + ll > nlit) | (A.op + ll + ml > cap)); // literals left, room in the destination (A.5)
                if (file_bits<LPF>(pm, f)) { xbad = true; ll = 0; ml = 0; if (!why) why = 7; }
                if (xbad) { ll = 0; ml = 0; } // (nothing more of this file is executed)
                A.ll = ll; A.ml = ml;
                A.w0 = (ll | ml) == 0 ? 0u : (ofv > 3 ? 4u | ((ofv - 3) << 3) : ofv - 1 + (ll == 0 ? 1u : 0u)); // repeat code (0..3; 4 = a new offset) | (offset value - 3) << 3
                A.chunk_l = bcast<LPF - 1, LPF>(il); A.chunk_t = bcast<LPF - 1, LPF>(it);
                lpos += A.chunk_l; opos += A.chunk_t;
            };
#if defined(MZD_SMALL_STAMPS) && !defined(MZD_SS_NOACC)
            uint64_t xa_ = 0, xr_ = 0, xl_ = 0, xm_ = 0, xn_ = 0, xf_ = 0, xc_ = 0, xq_ = 0, xrare_ = 0, x0_ = __builtin_readcyclecounter(), x1_ = 0;
#define XSTAMP(acc) do { x1_ = __builtin_readcyclecounter(); acc += x1_ - x0_; x0_ = x1_; } while (0)
#else
#define XSTAMP(acc)
#endif
            StepA cur;
            uint32_t q1 = load_rec4(LPF), q2 = load_rec4(2 * LPF); // the 4-byte records of steps c + 1 and c + 2 ...
            uint64_t o1;                                            // ... and what step c + 1 has in the second array
            { const uint32_t r4 = load_rec4(0); const uint64_t r8 = load_rec8(0, r4); o1 = load_rec8(LPF, q1); stage_a(widen(r4, r8), cur); } // (two trips: the three 4-byte records, then what they point at)
            for (uint32_t c0 = 0; c0 < nrun; c0 += LPF) {
                // (what is consumed first, what is requested behind it: the compiler waits for EVERYTHING in flight where the number of loads
                //  depends on control flow, so nothing younger than a step may be outstanding when a record is used)
                const uint64_t recB = widen(q1, o1);
                asm volatile("" ::: "memory");
                const uint64_t o2 = load_rec8(c0 + 2 * LPF, q2); // (q2 was requested a step ago: it decides here, before the next request goes out)
                asm volatile("" ::: "memory");
                const uint32_t q3 = load_rec4(c0 + 3 * LPF);
                StepA nxt;
#ifdef MZD_SMALL_STAMPS
                wg_steps_++;
#endif
                XSTAMP(xm_);
                stage_a(recB, nxt);
                q1 = q2; q2 = q3; o1 = o2;
                XSTAMP(xa_);
                // ---- B(c)
                uint32_t ll = cur.ll, ml = cur.ml;
                const uint32_t w0 = cur.w0, lp = cur.lp, op = cur.op;
                // repeat offsets: every lane its sequence's offset
                const uint32_t c = w0 & 7, pushv = w0 >> 3;
                uint32_t off;
                if (__ballot(c == 3) == 0) { // the scan over references
                    const uint32_t T = sel(m_eq(c, 4), (0x03010000u | RepRef<LPF>::flag) | sub, sel(m_eq(c, 0), kRepId, sel(m_eq(c, 1), 0x03020001u, 0x03010002u)));
                    uint32_t P = T;
                    P = rep_scan<LPF>(P, sub);
                    const uint32_t Ex = rep_before<LPF>(P, sub); // the state before this sequence
                    const uint32_t PL = bcast<LPF - 1, LPF>(P);  // ... and behind the step's last
                    // (the four trips through the crossbar are requested together: one wait)
                    auto fetch = [&](uint32_t ref) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & ~(LPF - 1)) | (ref & (LPF - 1))) * 4), (int)pushv); };
                    auto pick = [&](uint32_t ref, uint32_t vc) -> uint32_t {
                        const uint32_t vi = sel(m_eq(ref, 0), rep0, sel(m_eq(ref, 1), rep1, rep2));
                        return sel(m_ne(ref & RepRef<LPF>::flag, 0), vc, vi);
                    };
                    const uint32_t rm = (Ex >> (8 * (c & 3))) & 0xFF, r0 = PL & 0xFF, r1 = (PL >> 8) & 0xFF, r2 = (PL >> 16) & 0xFF;
                    const uint32_t vm = fetch(rm), v0 = fetch(r0), v1 = fetch(r1), v2 = fetch(r2);
                    off = sel(m_eq(c, 4), pushv, pick(rm, vm));
                    const uint32_t n0 = pick(r0, v0), n1 = pick(r1, v1), n2 = pick(r2, v2);
                    rep0 = n0; rep1 = n1; rep2 = n2;
                } else { // in order
                    off = 0;
                    static_for<LPF>([&](auto kc) {
                        constexpr int K = decltype(kc)::value;
                        const uint32_t a0 = bcast<K, LPF>(w0);
                        const uint32_t ck = a0 & 7, pk = a0 >> 3;
                        const lmask e0 = m_eq(ck, 0), e1 = m_eq(ck, 1), e2 = m_eq(ck, 2), e3 = m_eq(ck, 3);
                        const uint32_t X = sel(e2, rep2, sel(e3, rep0 - 1, pk));
                        const uint32_t o = sel(e0, rep0, sel(e1, rep1, X));
                        rep2 = sel(e0 | e1, rep2, rep1); rep1 = sel(e0, rep1, rep0); rep0 = o;
                        off = sel(m_eq(sub, K), o, off);
                    });
                }
                XSTAMP(xr_);
                const uint32_t mp = op + ll; // where the match goes
                {   // offset within the history, and not zero ("rep0 - 1")
                    const uint64_t om = __ballot(ml != 0 && off - 1 >= mp + dict_len);
                    if (file_bits<LPF>(om, f)) { xbad = true; ll = 0; ml = 0; if (!why) why = 8; }
                }
                // the literals (<= 31 bytes per lane in exact pieces; longer runs by the file's lanes, a byte each per round) and the
                // matches that lie wholly in the dictionary (requested first).  The window's tail is the literals' source: what a sequence
                // writes ends at or below its own literals' end, so the order is -- every short run is read; the long runs are copied,
                // ascending (none of them reaches the source of a later run); the short runs are written.
                {
                    bool dfull = false;
                    V16 D0 = {0, 0}, D1 = {0, 0};
                    if (DICT) {
                        dfull = ml != 0 && ml < 32 && off > mp && off - mp >= ml;
#ifdef MZD_EXP_NODICTFETCH // (experiment: what the trips to the dictionary content in L2 cost -- the bytes stored are wrong)
                        const uint8_t* dp = seq_g;
#else
                        const uint8_t* dp = dfull ? dict_end - (off - mp) : seq_g;
#endif // (dictionary buffers are readable 32 bytes past their end; a lane without such a match reads anything readable)
                        D0 = gv16(dp); D1 = gv16(dp + 16);
                    }
                    const uint32_t sa = lit_base + lp;
                    const uint32_t ls = ll < 32 ? ll : 0u; // (what this lane stores itself)
                    const bool lit_hi = __ballot(ls >= 16) != 0; // (wave-uniform: half the steps have no run of 16..31 literals)
                    V16 AB_ = {0, 0}, CD_ = {0, 0};
                    if (ls) AB_ = lds_u128(sa); // (two thirds of the sequences bring no literals: their lanes stay out of the read)
                    if (ls >= 16) CD_ = lds_u128(sa + 16);
                    asm volatile("" ::: "memory");
                    // (this loop runs per file: the lanes of files that are through are not here)
                    uint32_t big = file_bits<LPF>(__ballot(ll >= 32), f);
                    while (big) {
                        const uint32_t bl4 = ((lane & ~(LPF - 1)) + (uint32_t)__builtin_ctz(big)) * 4;
                        big &= big - 1;
                        const uint32_t n2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bl4, (int)ll), o2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bl4, (int)op), s2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bl4, (int)sa);
                        copy_run_lanes<LPF>(outo + o2, s2, n2, sub); // (destination at or below the source)
                    }
                    {
                        uint32_t at = outo + op, d0 = (uint32_t)AB_.a, d1 = (uint32_t)(AB_.a >> 32), d2 = (uint32_t)AB_.b, d3 = (uint32_t)(AB_.b >> 32);
                        if (lit_hi) store_piece16(at, ls, d0, d1, d2, d3, (uint32_t)CD_.a, (uint32_t)(CD_.a >> 32), (uint32_t)CD_.b, (uint32_t)(CD_.b >> 32));
                        store_exact15(at, ls, d0, d1, d2, d3);
                    }
                    if (DICT) {
                        {
                            const uint32_t dn = dfull ? ml : 0u;
                            uint32_t at = outo + mp, d0 = (uint32_t)D0.a, d1 = (uint32_t)(D0.a >> 32), d2 = (uint32_t)D0.b, d3 = (uint32_t)(D0.b >> 32);
                            store_piece16(at, dn, d0, d1, d2, d3, (uint32_t)D1.a, (uint32_t)(D1.a >> 32), (uint32_t)D1.b, (uint32_t)(D1.b >> 32));
                            store_exact15(at, dn, d0, d1, d2, d3);
                        }
                        if (dfull) ml = 0; // done
                    }
                }
                wsync();
                XSTAMP(xl_);
                // the matches inside the window, in rounds: a lane copies its own match (<= 31 bytes, not overlapping itself) once
                // everything below its source's end is final, i.e. once that end is at or below the match of the file's first sequence still
                // waiting; a first sequence of any other kind -- longer, overlapping, starting in the dictionary -- is executed by the
                // file's lanes together (its source is complete by then).  The first waiting sequence never waits, so every round ends one.
                // A waiting sequence's place in its file's order travels with what a copy by the file's lanes TOGETHER needs: mp << 18 |
                // m << 13 | off (a simple match: off <= mp < 2^13 -- the capacity is at most 8 KiB and a match has a byte --, m < 32; any other kind: the m field 0).  The minimum over the waiting lanes
                // (DPP, no trip through LDS) is the file's first waiting sequence.  Rounds come in two forms (the choice is wave-uniform):
                //   wide: every lane whose source is complete copies its own match (two reads, six exact-piece stores; the upper 16 bytes
                //         only when some lane has them) -- the first round of a step finishes a third of the lanes, later ones three;
                //   together: once no file has more than kTogether sequences waiting, each round copies the first waiting match of every
                //         file, 32 / LPF bytes per lane (byte reads, byte stores): a third of a wide round's instructions and LDS work.
                {
                    constexpr uint32_t kTogether = 3;
                    const uint32_t m = ml;
                    const bool simple = (m < 32) & (off >= m) & (off <= mp);
                    bool pending = m != 0;
                    const uint32_t send = mp - off + m; // end of the source
                    const uint32_t key0 = (mp << 18) | (simple ? (m << 13) | off : 0u);
                    const bool rare_any = __ballot(pending & !simple) != 0; // (wave-uniform: most steps hold simple matches only)
                    bool wide = true, first_round = true; // (wave-uniform; a step's first round is a wide one)
                    for (;;) {
                        const uint64_t pm = __ballot(pending);
                        if (!pm) break;
#ifdef MZD_SMALL_STAMPS
                        wg_rounds_++;
#ifndef MZD_SS_NOACC
                        xn_++;
#endif
#endif
                        if (wide && !first_round) wide = __ballot((uint32_t)__builtin_popcount(file_bits<LPF>(pm, f)) > kTogether) != 0; // (the counts only fall)
                        first_round = false;
                        const uint32_t K = seg_min<LPF>(pending ? key0 : 0xFFFFFFFFu, lane);
                        const bool is_first = pending & (key0 == K);
#ifdef MZD_SMALL_STAMPS
                        XSTAMP(xf_);
#endif
                        if (wide) {
                            const uint32_t F = K >> 18;
                            const bool ready = pending & simple & (send <= F);
                            const uint32_t n = ready ? m : 0u;
                            const uint32_t ra = outo + mp - off;
                            const bool hi = __ballot(n >= 16) != 0;
                            V16 AB_ = {0, 0}, CD_ = {0, 0};
                            if (ready) AB_ = lds_u128(ra); // (only the lanes that copy take part in the reads)
                            if (n >= 16) CD_ = lds_u128(ra + 16);
                            asm volatile("" ::: "memory");
                            uint32_t at = outo + mp, d0 = (uint32_t)AB_.a, d1 = (uint32_t)(AB_.a >> 32), d2 = (uint32_t)AB_.b, d3 = (uint32_t)(AB_.b >> 32);
                            if (hi) store_piece16(at, n, d0, d1, d2, d3, (uint32_t)CD_.a, (uint32_t)(CD_.a >> 32), (uint32_t)CD_.b, (uint32_t)(CD_.b >> 32));
                            store_exact15(at, n, d0, d1, d2, d3);
                            asm volatile("" ::: "memory");
                            pending = pending & !ready;
                        } else { // the first waiting match of every file, by the file's lanes: bytes sub, sub + LPF, ... (its source ends at or below its own start: complete)
                            const uint32_t km = K == 0xFFFFFFFFu ? 0u : (K >> 13) & 31u; // (0: nothing waits in this file, or its first is not of the simple kind)
                            const uint32_t kd = outo + (K >> 18), ks = kd - (K & 0x1FFFu);
                            constexpr int NB = LPF >= 32 ? 1 : 32 / (int)LPF; // bytes per lane: sub, sub + LPF, ... (< 32)
                            uint32_t vb[NB];
#pragma unroll
                            for (int j = 0; j < NB; j++) { vb[j] = 0; if (sub + (uint32_t)j * LPF < km) vb[j] = L8(ks + sub + (uint32_t)j * LPF); }
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int j = 0; j < NB; j++) if (sub + (uint32_t)j * LPF < km) L8(kd + sub + (uint32_t)j * LPF) = (uint8_t)vb[j];
                            asm volatile("" ::: "memory");
                            pending = pending & !(is_first & simple);
                        }
#ifdef MZD_SMALL_STAMPS
                        XSTAMP(xc_);
#endif
                        const bool fc = pending & !simple & is_first; // the first one waiting, and not of the simple kind: longer, overlapping itself, starting in the dictionary
                        const uint64_t cm = rare_any ? __ballot(fc) : 0ull;
                        if (cm) {
                            const uint32_t seg = file_bits<LPF>(cm, f);
                            const uint32_t fl4 = ((lane & ~(LPF - 1)) + (seg ? (uint32_t)__builtin_ctz(seg) : 0u)) * 4;
                            const uint32_t fmp = (uint32_t)__builtin_amdgcn_ds_bpermute((int)fl4, (int)mp), foff = (uint32_t)__builtin_amdgcn_ds_bpermute((int)fl4, (int)off), fm = (uint32_t)__builtin_amdgcn_ds_bpermute((int)fl4, (int)m);
                            if (seg) rare_match<LPF, DICT>(outo, fmp, foff, fm, sub, dict_end, dict_len);
                            pending = pending & !fc;
#if defined(MZD_SMALL_STAMPS) && !defined(MZD_SS_NOACC)
                            xrare_++;
#endif
                        }
                        asm volatile("" ::: "memory");
#ifdef MZD_SMALL_STAMPS
                        XSTAMP(xq_);
#endif
                    }
                }
                wsync();
                cur = nxt;
            }
#if defined(MZD_SMALL_STAMPS) && !defined(MZD_SS_NOACC)
            if (a.stamps && w0 && wv == 0 && blockIdx.x == 0 && lane == 0 && first_group) { a.stamps[18] = xa_; a.stamps[19] = xr_; a.stamps[20] = xl_; a.stamps[21] = xm_; a.stamps[22] = xn_; a.stamps[23] = xf_; a.stamps[24] = xc_; a.stamps[25] = xq_; a.stamps[26] = xrare_; }
#endif
            // the literals behind the last sequence
            bool good = !xbad;
            if (good) {
                const uint32_t lend = lpos, oend = opos; // (A() ran one step past the end: zero sequences, nothing added)
                const uint32_t rest = nlit - lend;
                good = rest <= cap - oend;
                if (good) {
                    copy_run_lanes<LPF>(outo + oend, lit_base + lend, rest, sub); // (destination at or below the source)
                    out_len = oend + rest;
                    good = !(has_fcs && out_len != fcs);
                }
            }
            if (!good) { ok = false; live = false; if (!why) why = 9; }
        }
        wsync();
        SSTAMP(17);

        // =============================== XXH64 over the window (lane = (file, accumulator))
        {
            uint32_t ck_bad = 0;
            const bool hashing = ok && has_ck && n != 0;
            const uint32_t nstripes = hashing ? out_len / 32 : 0u;
            uint64_t v = (sub & 3) == 0 ? XP1 + XP2 : ((sub & 3) == 1 ? XP2 : ((sub & 3) == 2 ? 0ull : 0ull - XP1));
            if constexpr (LPF >= 8) { // SPG stripes per group, a lane per (stripe, accumulator); the chain in the file's first four lanes
                constexpr uint32_t SPG = LPF >= 16 ? 4 : 2;
                if (sub < 4 * SPG) { // (32 lanes a file: its first row)
                    uint32_t q = res_off + 8 * sub;
                    const uint32_t ng = nstripes / SPG, rest = nstripes % SPG;
                    auto absorb = [&](uint64_t in) {
                        const uint64_t t = in * XP2;
                        v = xchain31<0>(v, t); v = xchain31<1>(v, t);
                        if (SPG == 4) { v = xchain31<2>(v, t); v = xchain31<3>(v, t); }
                    };
                    uint32_t g = 0;
                    for (; g + 2 <= ng; g += 2) { // (two groups' reads in flight)
                        const uint64_t i0 = lds_u64(q), i1 = lds_u64(q + 32 * SPG);
                        q += 64 * SPG;
                        absorb(i0); absorb(i1);
                    }
                    if (g < ng) { absorb(lds_u64(q)); q += 32 * SPG; }
                    if (rest) { // the last one to three stripes (lanes of stripes past the end hold anything: not used)
                        const uint64_t t = lds_u64(q) * XP2;
                        v = xchain31<0>(v, t);
                        if (SPG == 4) { if (rest > 1) v = xchain31<1>(v, t); if (rest > 2) v = xchain31<2>(v, t); }
                    }
                }
            } else if (sub < 4) {
                uint32_t q = res_off + 8 * sub;
                uint32_t s = 0;
                for (; s + 4 <= nstripes; s += 4) {
                    const uint64_t i0 = lds_u64(q), i1 = lds_u64(q + 32), i2 = lds_u64(q + 64), i3 = lds_u64(q + 96);
                    v = xround(v, i0); v = xround(v, i1); v = xround(v, i2); v = xround(v, i3);
                    q += 128;
                }
                for (; s < nstripes; s++) { v = xround(v, lds_u64(q)); q += 32; }
            }
            if (sub < 4) {
                const uint32_t acc_i = sub;
                const int base = (int)(lane & ~3u);
                const uint64_t v1 = __shfl(v, base), v2 = __shfl(v, base + 1), v3 = __shfl(v, base + 2), v4 = __shfl(v, base + 3);
                if (hashing && acc_i == 0) {
                    uint64_t hh;
                    if (out_len >= 32) {
                        hh = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
                        hh = xmerge(hh, v1); hh = xmerge(hh, v2); hh = xmerge(hh, v3); hh = xmerge(hh, v4);
                    } else hh = XP5;
                    hh += out_len;
                    hh = xxh_tail(hh, res_off + (out_len / 32) * 32, res_off + out_len);
#ifndef MZD_EXP_NOCK
#ifndef MZD_EXP_NODICTFETCH
                    if ((uint32_t)hh != stored_ck) ck_bad = 1;
#endif
#endif
                }
            }
            const uint64_t badm = __ballot(ck_bad != 0);
            if (file_bits<LPF>(badm, f)) { ok = false; if (!why) why = 10; }
        }
        SSTAMP(8);

        // =============================== the finished files leave LDS: all 64 lanes per file, 16 bytes per lane, to the destination
        // and to its mirror in the caller's pinned memory (DevJob::dst2)
        {
            uint64_t m = __ballot(leader && have && ok && out_len != 0);
            while (m) {
                const int fl_ = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)out_len, fl_);
                const uint32_t from = (uint32_t)__builtin_amdgcn_readlane((int)res_off, fl_);
                const uint64_t d1 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)dst, fl_) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uintptr_t)dst >> 32), fl_) << 32);
                const uint64_t d2 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)dst2, fl_) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uintptr_t)dst2 >> 32), fl_) << 32);
                uint8_t* const to = (uint8_t*)(uintptr_t)d1;
                uint8_t* const to2 = (uint8_t*)(uintptr_t)d2;
                const bool aligned = (from & 15) == 0; // (a raw block lies where the input put it)
                for (uint32_t o = lane * 16; o + 16 <= len; o += 1024) {
                    V16 v;
                    if (aligned) v = lds_v16(from + o); else { v.a = lds_u64(from + o); v.b = lds_u64(from + o + 8); }
                    gsv16(to + o, v);
                    if (to2) gsv16(to2 + o, v);
                }
                const uint32_t tail = len & ~15u;
                if (tail + lane < len) { const uint32_t v = L8(from + tail + lane); gs8(to + tail + lane, v); if (to2) gs8(to2 + tail + lane, v); }
            }
        }

        // =============================== results: done here, or handed to the general driver
        if (have && leader) {
            if (ok) { a.jobs[job].out_len = out_len; a.jobs[job].status = MZD_OK; }
            else {
                const uint32_t k = atomicAdd(&a.counter[4], 1u); a.redo_list[k] = job;
#ifdef MZD_SMALL_STAMPS
                if (a.stamps) { atomicAdd((unsigned long long*)&a.stamps[32 + (why & 15)], 1ull); a.stamps[48 + (why & 15)] = ((uint64_t)fidx << 32) | (uint64_t)(lane | (g << 8));
                    const unsigned long long kk = atomicAdd((unsigned long long*)&a.stamps[64], 1ull); if (kk < 960) a.stamps[65 + kk] = ((uint64_t)why << 32) | fidx; }
#endif
            }
        }
      } // pass
        SSTAMP(9);
#ifdef MZD_SMALL_STAMPS
        wg_groups_++;
#endif
        first_group = false;
        wgsync(); // (the group's last barrier: the slots are rewritten by the next group)
        if (w0) {
            if (!early) { g_next = ticket(); Jn = job_entry(list_entry(g_next)); prefetch(Jn, pfn); } // (wave-uniform)
            g = g_next; J = Jn;
#pragma unroll
            for (int k = 0; k < kPF; k++) pf[k] = pfn[k];
        }
        wsync(); // the slots are rewritten by the next group
    }
    leave();
#ifdef MZD_SMALL_STAMPS
    if (a.stamps && w0 && wv == 0 && lane == 0 && blockIdx.x < 3072) {
        uint64_t* w = a.stamps + 2048 + 16 * blockIdx.x;
        w[0] = wg_t0_; w[1] = __builtin_amdgcn_s_memrealtime();
        w[2] = (uint64_t)(uint32_t)__builtin_amdgcn_s_getreg(63492) | ((uint64_t)(uint32_t)__builtin_amdgcn_s_getreg(63508) << 32); w[3] = (uint64_t)wg_groups_ | ((uint64_t)wg_rounds_ << 16) | ((uint64_t)wg_steps_ << 32) | ((uint64_t)(uint32_t)(wg_te_ - wg_t0_) << 48);
    }
#endif
}

} // namespace lw

// wavefronts a CU holds by the kernels' register budgets: three per SIMD for the plain G = 4 kernel (168 registers: __launch_bounds__), two for the others
uint32_t lds_waves_by_registers(int g, int xg, int with_dict, int nw, int nd) { (void)xg; return ((g == 4 || nw > 1) && !with_dict) ? 12u : ((with_dict && nd <= 4) ? 4u : 8u); } // (a workgroup with a helper wavefront counts twice; the dictionary kernels take 267 registers -- one wavefront a SIMD -- unless built for five or more around an image: 256)
uint32_t lds_kernel_bytes_per_file(uint32_t tab_bytes, uint32_t comp_bytes) { return tab_bytes + lw::kAux + comp_bytes; } // a file's entropy image
uint32_t lds_kernel_bytes(int g, int xg, int with_dict, uint32_t tab_bytes, uint32_t comp_bytes, uint32_t out_bytes, int nw, int nd) {
    const uint32_t ent = tab_bytes + lw::kAux + comp_bytes;
    const uint32_t files = g != xg ? std::max<uint32_t>((uint32_t)g * ent, (uint32_t)xg * out_bytes) : (uint32_t)g * std::max(ent, out_bytes);
    const uint32_t one = lw::kShBytes + 32u * (uint32_t)g + (nw > 1 ? 64u * (uint32_t)g : 0u) + (with_dict ? lw::kDictImg : 0u) + files; // (a helper wavefront: 64 bytes a file between the two)
    return one + (nd > 1 ? (uint32_t)(nd - 1) * (512u + 32u * (uint32_t)g + files) : 0u); // (further decoding wavefronts around the one dictionary image: dump area, records, slots)
}
// what a slot has left for tables beside its input when the output window, not the input, sets its size (multiple of 16, at most 4 KiB)
uint32_t lds_spare_table_bytes(uint32_t comp_bytes, uint32_t out_bytes) {
    const uint32_t used = lw::kAux + comp_bytes;
    return out_bytes > used ? std::min<uint32_t>((out_bytes - used) & ~15u, 4096u) : 0u; // (the ring behind the tables holds 16-byte records)
}
size_t lds_scratch_per_file(uint32_t lit_stride, uint32_t seq_cap) { return (size_t)lit_stride + 12u * (size_t)seq_cap; } // (4-byte records + the sparse array of full ones)

// every instantiation the host can ask for: (files per wavefront, with a dictionary image, files executed at a time, wavefronts per workgroup)
#define MZD_LDS_VARIANTS(X) X(4, false, 4, 1, 1) X(8, false, 8, 1, 1) X(16, false, 16, 1, 1) X(8, false, 4, 1, 1) X(8, false, 4, 2, 1) X(4, false, 2, 1, 1) X(4, true, 4, 1, 1) X(8, true, 8, 1, 1) X(16, true, 16, 1, 1) X(8, true, 8, 1, 5) X(8, true, 8, 1, 8)
// The kernels ask for up to 160 KiB of dynamic LDS (the default limit is 64 KiB): the attribute belongs to the CURRENT device's
// function object, so it is raised once per device, from init_device (mzd_host.cpp), for every instantiation.
int lds_prepare_device() {
#define X(GG, DD, XX, WW, NN) if (hipFuncSetAttribute((const void*)lw::mzd_lds_kernel<GG, DD, XX, WW, NN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return MZD_E_DEVICE; }
    MZD_LDS_VARIANTS(X)
#undef X
    return MZD_OK;
}
int launch_lds(const LdsArgs& a, uint32_t grid, int g, int xg, int with_dict, int nw, int nd, void* stream) {
    const uint32_t bytes = lds_kernel_bytes(g, xg, with_dict, a.tab_bytes, a.comp_bytes, a.out_bytes, nw, nd);
    hipStream_t s = (hipStream_t)stream;
#define X(GG, DD, XX, WW, NN) if (g == GG && xg == XX && (with_dict != 0) == DD && nw == WW && nd == NN) { hipLaunchKernelGGL((lw::mzd_lds_kernel<GG, DD, XX, WW, NN>), dim3(grid), dim3(64 * WW * NN), bytes, s, a); return MZD_OK; }
    MZD_LDS_VARIANTS(X)
#undef X
    return MZD_E_PARAM;
}

} // namespace mzd
