// mzd_small.hip -- the small-file kernel: one LANE per file for everything that is a serial chain.
//
// Same arithmetic as mzd_kernels.hip (the frame decoder behind `zstd::stream::copy_decode`, reference
// src/main.rs:463-467; format: RFC 8878 / SURVEY.md Appendix A), other mapping.  A file of a few KiB is a handful of
// short chains -- Huffman weights, three normalized-count headers, the FSE state walk, the sequence execution, XXH64 --
// and a wavefront that walks ONE of them uses one lane in 64 (2.9 wave-instructions per output byte).
// Here a wavefront takes a GROUP of G files (BASELINE config 4: 10 000 x 4 KiB, G = 16, private tables in LDS;
// config 5: 50 000 records with one dictionary, G = 64, the dictionary's tables shared in LDS) and every phase runs on
// all of them at once:
//     A  frame / block / literals headers                 lane = file
//     B  Huffman weights (FSE-coded or direct)            lane = file
//     C  Huffman decode table                             lane = file           -> the file's LDS slot
//     D  Huffman streams                                  lane = (file, stream) -> literal scratch (HBM, L2-resident)
//     E  sequence header, normalized counts               lane = file
//     F  FSE decode tables                                lane = (file, table)  -> the file's LDS slot (Huffman table is dead)
//     G  state walk + repeat offsets + execute            lane = file, four sequences per step: HBM requests, then the walk of
//                                                         the next four (LDS only), then the stores
//     H  XXH64                                            lane = (file, accumulator)
// Header bytes, Huffman streams and sequence bitstreams are staged in LDS before the loops that read them: a loop that
// loads from HBM waits a round trip per load (s_waitcnt vmcnt(0) also waits for every store in flight), and a scattered
// vector-memory instruction costs a wavefront ~200 cycles of issue -- measured, tools/small_stamps.py.
// Only the plain case is decoded here: ONE frame holding ONE block, no error of any kind.  Anything else -- several
// frames or blocks, skippable frames, tables that do not fit the slot, every malformed input -- is handed, untouched,
// to the general drivers (the job index is appended to the launch's job list; mzd_host.cpp runs them right behind
// this kernel), so error classes and their order stay those of the block pipeline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <type_traits>

#include "../../include/mzd.h"
#include "mzd_device.h"
#include "mzd_tables.h"

namespace mzd {

#define DI __device__ __forceinline__

typedef const __attribute__((address_space(1))) uint8_t* gcp;
typedef __attribute__((address_space(1))) uint8_t* gp;

DI uint64_t gu64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, (gcp)p, 8); return v; }
DI uint32_t gu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, (gcp)p, 4); return v; }
DI uint32_t gu8(const uint8_t* p) { return *(gcp)p; }
DI void gs64(uint8_t* p, uint64_t v) { __builtin_memcpy((gp)p, &v, 8); }
DI void gs32(uint8_t* p, uint32_t v) { __builtin_memcpy((gp)p, &v, 4); }
DI void gs8(uint8_t* p, uint32_t v) { *(gp)p = (uint8_t)v; }
DI int hibit32(uint32_t v) { return 31 - __builtin_clz(v); }
struct V16 { uint64_t a, b; }; // 16 bytes, any alignment (global_load / global_store_dwordx4)
DI V16 gv16(const uint8_t* p) { V16 v; __builtin_memcpy(&v, (gcp)p, 16); return v; }
DI void gsv16(uint8_t* p, const V16& v) { __builtin_memcpy((gp)p, &v, 16); }

// ------------------------------------------------------------------------------------ LDS image (dynamic)
extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

// FSE decode entry, 4 bytes: next-state base (10) | nbBits (4) << 10 | symbol (8) << 14 | extra bits of the code (6) << 22
DI uint32_t fse_entry(uint32_t nbase, uint32_t nb, uint32_t sym, uint32_t extra) { return nbase | (nb << 10) | (sym << 14) | (extra << 22); }

constexpr uint32_t kOffPredef = 0;     // uint32 [160]: LL (64) | OF (32) | ML (64), predefined distributions
constexpr uint32_t kOffLLCode = 640;   // uint32 [36]: baseline | extra bits << 24
constexpr uint32_t kOffMLCode = 784;   // uint32 [53]
constexpr uint32_t kOffFiles = 1024;   // per-file areas
constexpr uint32_t kAux = 256;         // per file: Huffman weights, then the normalized counts of the three sequence tables
constexpr uint32_t kCtxBytes = 160;    // per file: FileLds (<= 128 bytes) + 32 bytes of scratch (rank counters)
constexpr uint32_t kCtxScratch = 128;
constexpr uint32_t kTreeStage = 1280;  // G = 16: the Huffman tree description is staged at main + kTreeStage (192 bytes), behind the weights' FSE table and counts
constexpr uint32_t kSeqStage = 192;    // staged bytes of a sequences section header
constexpr uint32_t kDictBytes = (512 + 512 + 256) * 4 + 2048 * 2;

struct FileLds { // what lanes other than the file's own need to know
    uint64_t src, dst, dst2;
    uint32_t live, streams, huf_off, huf_log; // huf_off: byte offset of the Huffman table in LDS
    uint32_t hs_lds;                          // LDS byte offset of the staged Huffman streams, or 0
    uint32_t s_off[4], s_len[4];
    uint32_t nlit, out_len, has_ck, n, nseq;
    uint8_t mode[4], al[4], nsym[4], rle[4];  // per sequence table (LL, OF, ML)
    uint32_t tab[3];                          // dword index of the table in LDS
};
static_assert(sizeof(FileLds) <= kCtxScratch, "FileLds");

template <int G> struct Lay {
    static constexpr uint32_t kMain = G == 16 ? 2048u : 16u; // Huffman table (<= 2^10 entries), later the FSE tables (<= 512 entries)
    static constexpr uint32_t kAuxB = G == 16 ? kAux : 0u;
    static constexpr uint32_t kStride = kMain + kAuxB + kCtxBytes;
    static constexpr uint32_t kSeqStageB = G == 16 ? kSeqStage : 256u; // staged bytes of a sequences section header (G = 16: inside the slot); G = 64: the file's stage, later its sequence bitstream
    static constexpr uint32_t kDict = kOffFiles + G * kStride;
    DI static uint32_t main_off(uint32_t f) { return kOffFiles + f * kStride; }
    DI static uint32_t aux_off(uint32_t f) { return main_off(f) + kMain; }
    DI static uint32_t ctx_off(uint32_t f) { return main_off(f) + kMain + kAuxB; }
};
DI uint32_t& L32(uint32_t off) { return *reinterpret_cast<uint32_t*>(lds + off); }
DI uint16_t& L16(uint32_t off) { return *reinterpret_cast<uint16_t*>(lds + off); }
DI int16_t& L16s(uint32_t off) { return *reinterpret_cast<int16_t*>(lds + off); }
DI uint8_t& L8(uint32_t off) { return lds[off]; }
template <int G> DI FileLds& fl(uint32_t f) { return *reinterpret_cast<FileLds*>(lds + Lay<G>::ctx_off(f)); }

DI void wave_sync() { // LDS + global writes of this wavefront visible to its other lanes (one wavefront per workgroup)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ------------------------------------------------------------------------------------ bit readers (per lane, HBM/L2)
// Bit readers over header bytes staged in LDS (`base` byte offset, `n` bytes staged, >= 8 readable bytes behind them).
DI uint64_t lds_u64(uint32_t off) { uint64_t v; __builtin_memcpy(&v, lds + off, 8); return v; }
struct LBack { // backward (the Huffman weights' stream: <= 128 bytes), zero below the start
    uint32_t base;
    int32_t h;
    uint64_t cur;
    int32_t avail;
    DI bool init(uint32_t off, uint32_t sl) {
        base = off; cur = 0; avail = 0; h = 0;
        if (sl == 0) return false;
        const uint32_t last = lds[off + sl - 1];
        if (last == 0) return false;
        h = (int32_t)((sl - 1) * 8) + hibit32(last);
        return true;
    }
    DI void refill() {
        if (h <= 0) { cur = 0; avail = 64; return; }
        const int32_t b = (h - 1) >> 3;
        uint64_t W = b >= 7 ? lds_u64(base + (uint32_t)(b - 7)) : lds_u64(base) << (8 * (7 - b)); // (the staged bytes before the stream are not its own)
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

// Normalized counts (A.3) from header bytes staged in LDS (>= 8 readable bytes behind them) -> int16 norm[] in LDS at byte
// offset `norm_off`.  Returns bytes used or 0 (give up).  sym_cap: symbols the caller has room for (<= max_sym + 1).
// Every field (<= 10 bits) is one unaligned 4-byte LDS read at its bit position: no window to maintain.  Bits past the
// description's end need no masking: a read that touches them either leaves `bit` past the limit (rejected), or it is the
// short form of a value whose dropped bit was the only one outside, which does not enter the value.
DI uint32_t read_ncount_lane(uint32_t src_off, uint32_t n, int max_log, int max_sym, int sym_cap, uint32_t norm_off, uint32_t& nsym_out, uint32_t& log_out) {
    if (n < 1) return 0;
    const int32_t limit = (int32_t)(n > 4096 ? 4096 : n) * 8;
    auto bits = [&](int32_t bit) -> uint32_t { uint32_t v; __builtin_memcpy(&v, lds + src_off + ((uint32_t)bit >> 3), 4); return v >> (bit & 7); }; // >= 25 bits
    int32_t bit = 4;
    const int al = 5 + (int)(bits(0) & 15);
    if (al > max_log) return 0;
    int remaining = 1 << al, sym = 0;
    bool bad = false;
    while (remaining > 0 && sym <= max_sym && !bad) {
        const int nb = hibit32((uint32_t)(remaining + 1)) + 1;
        bad |= bit >= limit;
        const int val = (int)(bits(bit) & ((1u << nb) - 1));
        const int lower = (1 << (nb - 1)) - 1;
        const int thr = (1 << nb) - 1 - (remaining + 1);
        const bool small = (val & lower) < thr;
        const int v2 = small ? (val & lower) : (val > lower ? val - thr : val);
        bit += small ? nb - 1 : nb;
        const int pr = v2 - 1;
        remaining -= (pr < 0) ? 1 : pr;
        bad |= (remaining < 0) | (sym >= sym_cap);
        if (bad) break;
        L16s(norm_off + 2 * (uint32_t)sym) = (int16_t)pr;
        sym++;
        if (pr == 0) { // runs of zero-probability symbols: 2 bits each, 3 = "and more"
            for (;;) {
                if (bit >= limit) { bad = true; break; }
                const int r = (int)(bits(bit) & 3);
                bit += 2;
                if (sym + r > max_sym + 1 || sym + r > sym_cap) { bad = true; break; }
                for (int i = 0; i < r; i++) L16s(norm_off + 2 * (uint32_t)(sym + i)) = 0;
                sym += r;
                if (r != 3) break;
            }
        }
    }
    if (bad || remaining != 0 || sym > max_sym + 1 || bit > limit) return 0;
    nsym_out = (uint32_t)sym;
    log_out = (uint32_t)al;
    return (uint32_t)((bit + 7) >> 3);
}

// FSE decode table (A.3) by ONE lane: `tab_off` byte offset of the uint32 table in LDS, norm (LDS) is turned into the
// per-symbol state counters on the way.  kind 0 LL, 1 OF, 2 ML, 3 Huffman weights (no extra bits).
DI bool build_fse_lane(uint32_t tab_off, uint32_t norm_off, uint32_t nsym, uint32_t log, int kind) {
    const uint32_t size = 1u << log, mask = size - 1;
    uint32_t high = size;
    for (uint32_t s = 0; s < nsym; s++)
        if (L16s(norm_off + 2 * s) == -1) { high--; L32(tab_off + 4 * high) = s; }
    const uint32_t step = (size >> 1) + (size >> 3) + 3;
    uint32_t pos = 0;
    for (uint32_t s = 0; s < nsym; s++) {
        const int c = L16s(norm_off + 2 * s);
        for (int i = 0; i < c; i++) {
            L32(tab_off + 4 * pos) = s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return false;
    for (uint32_t s = 0; s < nsym; s++)
        if (L16s(norm_off + 2 * s) == -1) L16s(norm_off + 2 * s) = 1;
    for (uint32_t i = 0; i < size; i++) {
        const uint32_t s = L32(tab_off + 4 * i);
        const uint32_t d = (uint32_t)(uint16_t)L16s(norm_off + 2 * s);
        L16s(norm_off + 2 * s) = (int16_t)(d + 1);
        const uint32_t nb = log - (uint32_t)hibit32(d);
        const uint32_t extra = kind == 0 ? L32(kOffLLCode + 4 * s) >> 24 : (kind == 1 ? s : (kind == 2 ? L32(kOffMLCode + 4 * s) >> 24 : 0u));
        L32(tab_off + 4 * i) = fse_entry((d << nb) - size, nb, s, extra);
    }
    return true;
}

// ------------------------------------------------------------------------------------ per-lane copies (HBM/L2)
// n bytes forward, byte-sequential semantics for dst - src >= 8 (or disjoint regions).  Wide stores may clobber up to 7
// bytes past the copy (they are rewritten by what follows, or lie between out_len and dst_cap); never past `dlim`.
DI void copy_lane(uint8_t* d, const uint8_t* s, uint32_t n, const uint8_t* dlim) {
    if (d + n + 8 <= dlim) {
        for (uint32_t k = 0; k < n; k += 8) gs64(d + k, gu64(s + k));
    } else {
        for (uint32_t k = 0; k < n; k++) gs8(d + k, gu8(s + k));
    }
}
// a match: n bytes from d - off, overlap allowed (A.5 "byte-sequentially")
DI void match_lane(uint8_t* d, uint32_t off, uint32_t n, const uint8_t* dlim) {
    if (off >= 32 && d + n + 32 <= dlim) { // (two 16-byte pieces in flight)
        const uint8_t* s = d - off;
        for (uint32_t k = 0; k < n; k += 32) {
            const V16 a = gv16(s + k), b = gv16(s + k + 16);
            gsv16(d + k, a);
            if (k + 16 < n) gsv16(d + k + 16, b);
        }
    } else if (d + n + 8 <= dlim) {
        if (off >= 8) {
            const uint8_t* s = d - off;
            if (off >= 16) { // two chunks in flight
                for (uint32_t k = 0; k < n; k += 16) {
                    const uint64_t a = gu64(s + k), b = gu64(s + k + 8);
                    gs64(d + k, a);
                    if (k + 8 < n) gs64(d + k + 8, b);
                }
            } else {
                for (uint32_t k = 0; k < n; k += 8) gs64(d + k, gu64(s + k));
            }
        } else { // period < 8: replicate the pattern to 8 bytes, advance by the largest multiple of the period
            uint64_t m = gu64(d - off);
            uint32_t sh = off * 8;
            m &= (1ull << sh) - 1;
            while (sh < 64) { m |= m << sh; sh *= 2; }
            const uint32_t stride = (8 / off) * off;
            for (uint32_t k = 0; k < n; k += stride) gs64(d + k, m);
        }
    } else {
        for (uint32_t k = 0; k < n; k++) gs8(d + k, gu8(d + k - off));
    }
}

// ------------------------------------------------------------------------------------ XXH64 pieces (A.6)
DI uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
DI uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
DI uint64_t xmerge(uint64_t hh, uint64_t v) { v = xround(0, v); hh ^= v; return hh * XP1 + XP4; }
DI uint64_t xxh_tail(uint64_t hh, const uint8_t* q, const uint8_t* end) {
    while (q + 8 <= end) { hh ^= xround(0, gu64(q)); hh = rotl64(hh, 27) * XP1 + XP4; q += 8; }
    if (q + 4 <= end) { hh ^= (uint64_t)gu32(q) * XP1; hh = rotl64(hh, 23) * XP2 + XP3; q += 4; }
    while (q < end) { hh ^= (uint64_t)gu8(q) * XP5; hh = rotl64(hh, 11) * XP1; q++; }
    hh ^= hh >> 33; hh *= XP2; hh ^= hh >> 29; hh *= XP3; hh ^= hh >> 32;
    return hh;
}

// Diagnostic build only (-DMZD_SMALL_STAMPS): cycle counter of workgroup 0 at every phase boundary of its first group.
#ifdef MZD_SMALL_STAMPS
#define SSTAMP(k) do { if (a.stamps && blockIdx.x == 0 && lane == 0 && first_group) a.stamps[k] = __builtin_readcyclecounter(); } while (0)
#else
#define SSTAMP(k)
#endif

// ------------------------------------------------------------------------------------ the kernel
struct DictInfo { // the dictionary whose tables sit in LDS (wave-uniform)
    uint32_t handle, formatted, dict_id, content_len, huf_log;
    uint32_t al[3], rep[3];
    const uint8_t* content;
};

// 8 bytes starting at byte i of the first 32 bytes of a file (little endian; zero past byte 31)
struct Hdr32 { uint64_t w[4]; };
DI uint64_t hdr_at(const Hdr32& h, uint32_t i) {
    const uint32_t k = i >> 3, sh = (i & 7) * 8;
    const uint64_t a = k == 0 ? h.w[0] : (k == 1 ? h.w[1] : (k == 2 ? h.w[2] : (k == 3 ? h.w[3] : 0ull)));
    const uint64_t b = k == 0 ? h.w[1] : (k == 1 ? h.w[2] : (k == 2 ? h.w[3] : 0ull));
    return sh ? (a >> sh) | (b << (64 - sh)) : a;
}

template <int G>
__global__ __launch_bounds__(64) void mzd_small_kernel(SmallArgs a) {
    using LY = Lay<G>;
    constexpr int LPF = 64 / G; // lanes per file in the (file, part) phases: 4 or 1
    const int lane = threadIdx.x;
    const uint32_t dict_off = LY::kDict;
    const uint32_t stage64_off = dict_off + (a.with_dict ? kDictBytes : 0u); // G = 64: the staged sequence headers, behind the dictionary's tables
    (void)stage64_off;

    // ---- once per wavefront: code tables, predefined tables
    if (lane < 36) L32(kOffLLCode + 4 * lane) = LL_BASE[lane] | ((uint32_t)LL_BITS[lane] << 24);
    if (lane < 53) L32(kOffMLCode + 4 * lane) = ML_BASE[lane] | ((uint32_t)ML_BITS[lane] << 24);
    wave_sync();
    {
        // the three predefined distributions, built by lanes 0..2 with the per-lane builder (norm scratch: the dictionary
        // area when there is one, else the first file slots -- both idle here)
        const uint32_t scratch = a.with_dict ? dict_off : kOffFiles;
        if (lane < 3) {
            const uint32_t noff = scratch + (uint32_t)lane * 128;
            const uint32_t nsym = lane == 0 ? 36u : (lane == 1 ? 29u : 53u);
            for (uint32_t s = 0; s < nsym; s++) L16s(noff + 2 * s) = lane == 0 ? LL_DEF[s] : (lane == 1 ? OF_DEF[s] : ML_DEF[s]);
            const uint32_t toff = kOffPredef + (lane == 0 ? 0u : (lane == 1 ? 256u : 384u));
            build_fse_lane(toff, noff, nsym, lane == 1 ? 5u : 6u, lane);
        }
        wave_sync();
    }
    DictInfo di;
    di.handle = 0; di.formatted = 0; di.dict_id = 0; di.content_len = 0; di.huf_log = 0; di.content = nullptr;
    for (int t = 0; t < 3; t++) { di.al[t] = 0; di.rep[t] = 0; }

    uint8_t* const lit_base = a.lit_scratch + (size_t)blockIdx.x * G * a.lit_stride;
    const uint32_t ngroups = (a.nsmall + G - 1) / G;

    bool first_group = true;
    (void)first_group;
    for (;;) {
        uint32_t g = 0;
        SSTAMP(0);
        if (lane == 0) g = atomicAdd(&a.counter[5], 1u);
        g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        if (g >= ngroups) break;

        // =============================== phase A: headers (lane = file)
        const uint32_t fidx = g * G + (uint32_t)lane;
        const bool have = lane < G && fidx < a.nsmall;
        uint32_t job = 0;
        const uint8_t* src = nullptr; uint8_t* dst = nullptr;
        uint32_t n = 0, cap = 0, jdict = 0;
        uint8_t* dst2 = nullptr; // host mirror of the output, or null
        if (have) {
            job = a.small_list[fidx];
            const DevJob& dj = a.jobs[job];
            src = dj.src; dst = dj.dst; n = (uint32_t)dj.src_len; cap = (uint32_t)dj.dst_cap; jdict = dj.dict; dst2 = dj.dst2;
        }
        // the group's dictionary: the first one named (the host sorts the list by dictionary)
        if (a.with_dict) {
            const uint64_t named = __ballot(have && jdict != 0 && jdict <= a.ndicts);
            if (named) {
                const uint32_t want = (uint32_t)__builtin_amdgcn_readlane((int)jdict, __builtin_ctzll(named));
                if (want != di.handle) {
                    const DevDict* dd = &a.dicts[want - 1];
                    di.handle = want; di.formatted = dd->formatted; di.dict_id = dd->dict_id; di.content_len = dd->content_len;
                    di.huf_log = dd->huf_log; di.content = dd->content;
                    for (int t = 0; t < 3; t++) { di.al[t] = dd->al[t]; di.rep[t] = dd->rep[t]; }
                    if (di.formatted) {
                        auto conv = [](uint64_t e) -> uint32_t { const uint32_t lo = (uint32_t)e, hi = (uint32_t)(e >> 32); return fse_entry(lo >> 3, hi & 0xFF, (hi >> 16) & 0xFF, hi >> 24); };
                        for (int i = lane; i < 512; i += 64) { L32(dict_off + 4 * i) = conv(dd->ll[i]); L32(dict_off + 2048 + 4 * i) = conv(dd->ml[i]); }
                        for (int i = lane; i < 256; i += 64) L32(dict_off + 4096 + 4 * i) = conv(dd->of[i]);
                        for (int i = lane; i < 1024; i += 64) L32(dict_off + 5120 + 4 * i) = reinterpret_cast<const uint32_t*>(dd->huf)[i];
                    }
                    wave_sync();
                }
            }
        }
        bool ok = have;       // still on the fast path
        bool done = false;    // finished without a block to decode (empty file, raw / RLE block)
        uint32_t out_len = 0;
        uint32_t has_fcs = 0, has_ck = 0, fcs = 0, btype = 0, bsize = 0, b0 = 0;
        uint32_t lit_type = 0, nlit = 0, streams = 0, lit_off = 0, tree_off = 0, tree_len = 0;
        uint32_t s_len0 = 0, s_len1 = 0, s_len2 = 0, s_len3 = 0, s_base = 0;
        uint32_t seq_off = 0, seq_len = 0;
        const bool with_d = ok && jdict != 0;
        if (ok && n == 0) { done = true; }  // no frame at all: nothing to decode
        else if (ok) {
            ok = false;
            do {
                if (jdict > a.ndicts || (jdict && jdict != di.handle)) break;
                if (n < 9 || n > kSmallSrcMax) break;
                Hdr32 h;
#pragma unroll
                for (int k = 0; k < 4; k++) h.w[k] = (uint32_t)(8 * k + 8) <= n + MZD_SRC_PADDING ? gu64(src + 8 * k) : 0ull;
                if ((uint32_t)h.w[0] != 0xFD2FB528u) break;
                const uint32_t fhd = (uint32_t)(h.w[0] >> 32) & 0xFF;
                const uint32_t fcsf = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
                if (fhd & 8) break;
                const uint32_t did_sz = did == 3 ? 4u : did, fcs_sz = fcsf == 0 ? single : (1u << fcsf);
                const uint32_t hs = 5 + (single ? 0u : 1u) + did_sz + fcs_sz;
                if (n < hs + 3) break;
                uint32_t q = 5;
                uint64_t window = 0;
                if (!single) { const uint32_t b = (uint32_t)hdr_at(h, q) & 0xFF; q++; const uint32_t wl = 10 + (b >> 3); window = (1ull << wl) + ((1ull << wl) >> 3) * (b & 7); }
                uint32_t frame_dict = 0;
                if (did_sz) { frame_dict = (uint32_t)(hdr_at(h, q) & (did_sz == 4 ? 0xFFFFFFFFull : ((1ull << (8 * did_sz)) - 1))); q += did_sz; }
                has_fcs = 1;
                uint64_t fcs64 = 0;
                if (fcsf == 0) { if (single) fcs64 = hdr_at(h, q) & 0xFF; else has_fcs = 0; }
                else if (fcsf == 1) fcs64 = (hdr_at(h, q) & 0xFFFF) + 256;
                else if (fcsf == 2) fcs64 = hdr_at(h, q) & 0xFFFFFFFFull;
                else fcs64 = hdr_at(h, q);
                if (single) window = fcs64;
                if (window > (1ull << 27) + 1) break;
                const uint32_t block_max = (uint32_t)(window < kBlockMax ? window : kBlockMax);
                has_ck = (fhd >> 2) & 1;
                if (frame_dict && frame_dict != (with_d && di.formatted ? di.dict_id : 0u)) break;
                if (has_fcs && fcs64 > cap) break;
                fcs = (uint32_t)fcs64;
                const uint32_t bh = (uint32_t)hdr_at(h, hs) & 0xFFFFFF;
                const uint32_t last = bh & 1;
                btype = (bh >> 1) & 3; bsize = bh >> 3;
                if (!last || btype == 3 || bsize > block_max) break;
                const uint32_t body = btype == 1 ? 1u : bsize;
                if ((uint64_t)hs + 3 + body + (has_ck ? 4u : 0u) != n) break; // exactly one frame of one block, nothing behind it
                b0 = hs + 3;
                if (btype < 2) {
                    if (bsize > cap || (has_fcs && fcs != bsize)) break;
                    out_len = bsize; done = true; ok = true;
                    break;
                }
                if (bsize < 2) break;
                // ---- literals section header (A.4)
                const uint64_t lb = hdr_at(h, b0);
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
                        if (G != 16) break; // no room for a private Huffman table in this layout
                        if (rem < 1) break;
                        const uint32_t hb = (uint32_t)hdr_at(h, p_off) & 0xFF;
                        const uint32_t tl = hb >= 128 ? 1 + ((hb - 127) + 1) / 2 : 1 + hb;
                        if (tl > rem || (hb < 128 && hb < 1)) break;
                        tree_off = p_off; tree_len = tl;
                        p_off += tl; rem -= tl;
                    } else if (!(with_d && di.formatted)) break; // treeless without a table to reuse
                    if (streams == 1) { s_base = p_off; s_len0 = rem; if (rem == 0) break; }
                    else {
                        if (rem < 10) break;
                        const uint64_t jt = gu64(src + p_off);
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
        // publish what other lanes need
        if (lane < G) {
            FileLds& F = fl<G>((uint32_t)lane);
            F.src = (uint64_t)(uintptr_t)src; F.dst = (uint64_t)(uintptr_t)dst; F.dst2 = (uint64_t)(uintptr_t)dst2; F.n = n;
            F.live = (ok && !done) ? 1u : 0u;
            F.streams = (ok && !done && lit_type >= 2) ? streams : 0u;
            F.nlit = nlit;
            F.s_off[0] = s_base; F.s_off[1] = s_base + s_len0; F.s_off[2] = s_base + s_len0 + s_len1; F.s_off[3] = s_base + s_len0 + s_len1 + s_len2;
            F.s_len[0] = s_len0; F.s_len[1] = s_len1; F.s_len[2] = s_len2; F.s_len[3] = s_len3;
            F.has_ck = has_ck; F.out_len = out_len; F.nseq = 0;
        }
        // raw / RLE blocks: all 64 lanes copy / fill, one file after the other
        {
            uint64_t plain = __ballot(ok && done && n != 0 && bsize != 0);
            while (plain) {
                const int fl_ = __builtin_ctzll(plain);
                plain &= plain - 1;
                const uint32_t bs = (uint32_t)__builtin_amdgcn_readlane((int)bsize, fl_), bt = (uint32_t)__builtin_amdgcn_readlane((int)btype, fl_), bo = (uint32_t)__builtin_amdgcn_readlane((int)b0, fl_);
                const uint64_t sp64 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)src, fl_) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uintptr_t)src >> 32), fl_) << 32);
                const uint64_t dp64 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)dst, fl_) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uintptr_t)dst >> 32), fl_) << 32);
                const uint8_t* s_ = (const uint8_t*)(uintptr_t)sp64 + bo;
                uint8_t* d_ = (uint8_t*)(uintptr_t)dp64;
                if (bt == 0) for (uint32_t k = (uint32_t)lane; k < bs; k += 64) gs8(d_ + k, gu8(s_ + k));
                else { const uint32_t v = gu8(s_); for (uint32_t k = (uint32_t)lane; k < bs; k += 64) gs8(d_ + k, v); }
            }
        }
        bool live = ok && !done; // a compressed block to decode
        SSTAMP(1);

        // =============================== staging: the Huffman tree descriptions -> LDS (lane = (file, quarter))
        // (a lone lane parsing bits out of HBM pays a round trip per window, and every lane of the group at another moment)
        if (G == 16) {
            const uint32_t f = (uint32_t)lane >> 2, q = (uint32_t)lane & 3;
            const uint64_t tl = __ballot(live && lit_type == 2); // files with a tree
            const uint32_t toff = (uint32_t)__shfl((int)tree_off, (int)f);
            if ((tl >> f) & 1) {
                const FileLds& F = fl<G>(f);
                const uint8_t* sp = (const uint8_t*)(uintptr_t)F.src;
                const uint32_t dst_off = LY::main_off(f) + kTreeStage + q * 48;
#pragma unroll
                for (int k = 0; k < 6; k++) { // <= 129 bytes of tree; nothing is read past the input + its padding
                    const uint32_t at = toff + q * 48 + 8 * k;
                    const uint64_t v = (q * 48 + 8 * k < 144 && at + 8 <= F.n + MZD_SRC_PADDING) ? gu64(sp + at) : 0ull;
                    __builtin_memcpy(lds + dst_off + 8 * k, &v, 8);
                }
            }
            wave_sync();
        }

        // =============================== phase B + C: Huffman weights and table (lane = file; G = 16 only)
        uint32_t huf_log = di.huf_log, huf_off = dict_off + 5120; // treeless: the dictionary's table
        if (G == 16) {
            const uint32_t mo = LY::main_off((uint32_t)lane), ao = LY::aux_off((uint32_t)lane), so = LY::ctx_off((uint32_t)lane) + kCtxScratch;
            if (live && lit_type == 2) {
                bool good = false;
                uint32_t nw = 0;
                do {
                    const uint32_t tp = mo + kTreeStage; // the staged tree description
                    const uint32_t hb = L8(tp);
                    if (hb >= 128) { // direct: 4 bits per weight, high nibble first
                        nw = hb - 127;
                        for (uint32_t i = 0; i < nw; i++) {
                            const uint32_t by = L8(tp + 1 + i / 2);
                            L8(ao + i) = (uint8_t)((i & 1) ? (by & 15) : (by >> 4));
                        }
                    } else {
                        uint32_t nsym = 0, log = 0;
                        const uint32_t hdr = read_ncount_lane(tp + 1, hb, 6, 255, 16, mo + 1024, nsym, log);
                        if (hdr == 0 || hdr >= hb) break;
                        if (!build_fse_lane(mo, mo + 1024, nsym, log, 3)) break;
                        LBack rd;
                        if (!rd.init(tp + 1 + hdr, hb - hdr)) break;
                        rd.refill();
                        uint32_t s1 = rd.read(log), s2 = rd.read(log);
                        bool fin = false;
                        for (;;) { // two interleaved states; the stream's over-read ends it (A.4)
                            if (nw > 253) break;
                            uint32_t e = L32(mo + 4 * s1);
                            L8(ao + nw) = (uint8_t)(e >> 14); nw++;
                            s1 = (e & 0x3FF) + rd.read((e >> 10) & 15);
                            if (rd.h < 0) { L8(ao + nw) = (uint8_t)(L32(mo + 4 * s2) >> 14); nw++; fin = true; break; }
                            if (nw > 253) break;
                            e = L32(mo + 4 * s2);
                            L8(ao + nw) = (uint8_t)(e >> 14); nw++;
                            s2 = (e & 0x3FF) + rd.read((e >> 10) & 15);
                            if (rd.h < 0) { L8(ao + nw) = (uint8_t)(L32(mo + 4 * s1) >> 14); nw++; fin = true; break; }
                        }
                        if (!fin) break;
                    }
                    // ---- validation, implied last weight, canonical table (A.4)
                    if (nw < 1 || nw > 255) break;
                    for (uint32_t r = 0; r < 16; r++) L16(so + 2 * r) = 0; // rank counters
                    uint32_t total = 0;
                    bool wbad = false;
                    for (uint32_t i = 0; i < nw; i++) {
                        const uint32_t w = L8(ao + i);
                        if (w > 12) { wbad = true; break; }
                        L16(so + 2 * w) = (uint16_t)(L16(so + 2 * w) + 1);
                        total += w ? 1u << (w - 1) : 0u;
                    }
                    if (wbad || total == 0) break;
                    const uint32_t maxbits = (uint32_t)hibit32(total) + 1;
                    if (maxbits > 10) break; // 11: valid, but the table would not fit the slot (the general path takes it)
                    const uint32_t left = (1u << maxbits) - total;
                    if (left & (left - 1)) break;
                    const uint32_t wl = (uint32_t)hibit32(left) + 1;
                    L8(ao + nw) = (uint8_t)wl; nw++;
                    L16(so + 2 * wl) = (uint16_t)(L16(so + 2 * wl) + 1);
                    const uint32_t r1 = L16(so + 2);
                    if (r1 < 2 || (r1 & 1)) break;
                    uint32_t pos = 0; // rank counters -> start positions (weight 1 = longest codes first)
                    for (uint32_t r = 1; r <= maxbits; r++) { const uint32_t c = L16(so + 2 * r); L16(so + 2 * r) = (uint16_t)pos; pos += c << (r - 1); }
                    if (pos != (1u << maxbits)) break; // also catches weights above maxbits
                    for (uint32_t s = 0; s < nw; s++) { // (the staged tree and the weights' FSE table are dead: the table takes their place)
                        const uint32_t w = L8(ao + s);
                        if (!w) continue;
                        const uint32_t cnt = 1u << (w - 1), at = L16(so + 2 * w);
                        L16(so + 2 * w) = (uint16_t)(at + cnt);
                        const uint32_t e = s | ((maxbits + 1 - w) << 8);
                        if (cnt == 1) L16(mo + 2 * at) = (uint16_t)e;
                        else for (uint32_t i = 0; i < cnt; i += 2) L32(mo + 2 * (at + i)) = e | (e << 16);
                    }
                    huf_log = maxbits; huf_off = mo;
                    good = true;
                } while (false);
                if (!good) { ok = false; live = false; }
            }
        }
        if (lane < G) {
            FileLds& F = fl<G>((uint32_t)lane);
            F.huf_off = huf_off; F.huf_log = huf_log;
            if (!live) { F.live = 0; F.streams = 0; }
        }
        wave_sync();
        SSTAMP(2);

        // =============================== staging: the Huffman streams -> LDS, where they fit (lane = (file, part))
        // A hot loop must not wait for HBM: the compiler guards every use of a loaded value with s_waitcnt vmcnt(0), which
        // also waits for every store in flight.  So the streams of a file (one contiguous piece of its block) are copied
        // into LDS first -- G = 16: behind the file's Huffman table in its slot; G = 64: into the file's stage.
        uint32_t hs_lds = 0; // LDS byte offset of the first stream's first byte, or 0 (read from HBM)
        if (live && lit_type >= 2) {
            const uint32_t tbytes = (G == 16 && lit_type == 2) ? (2u << huf_log) : 0u; // (treeless: the dictionary's table, shared)
            const uint32_t room0 = G == 16 ? LY::main_off((uint32_t)lane) + tbytes : stage64_off + (uint32_t)lane * LY::kSeqStageB;
            const uint32_t room1 = G == 16 ? LY::main_off((uint32_t)lane) + LY::kMain : room0 + LY::kSeqStageB;
            const uint32_t total = s_len0 + s_len1 + s_len2 + s_len3;
            if (room0 + 16 + total + 8 <= room1) hs_lds = room0 + 16;
        }
        {
            const uint32_t f = (uint32_t)lane / LPF, q = (uint32_t)lane % LPF;
            const uint32_t to = (uint32_t)__shfl((int)hs_lds, (int)f), from = (uint32_t)__shfl((int)s_base, (int)f);
            const uint32_t len = (uint32_t)__shfl((int)(s_len0 + s_len1 + s_len2 + s_len3), (int)f);
            if (to) {
                const uint8_t* sp = (const uint8_t*)(uintptr_t)fl<G>(f).src + from;
                if (q == 0) { const uint64_t z = 0; __builtin_memcpy(lds + to - 16, &z, 8); __builtin_memcpy(lds + to - 8, &z, 8); }
                uint32_t k = q * 8;
                for (; k + 3 * LPF * 8 < len; k += 4 * LPF * 8) { // four loads in flight
                    const uint64_t v0 = gu64(sp + k), v1 = gu64(sp + k + LPF * 8), v2 = gu64(sp + k + 2 * LPF * 8), v3 = gu64(sp + k + 3 * LPF * 8);
                    __builtin_memcpy(lds + to + k, &v0, 8); __builtin_memcpy(lds + to + k + LPF * 8, &v1, 8);
                    __builtin_memcpy(lds + to + k + 2 * LPF * 8, &v2, 8); __builtin_memcpy(lds + to + k + 3 * LPF * 8, &v3, 8);
                }
                for (; k < len; k += LPF * 8) { const uint64_t v = gu64(sp + k); __builtin_memcpy(lds + to + k, &v, 8); } // (reads <= 7 bytes past the streams: input padding; writes stay below room1)
            }
        }
        if (lane < G) fl<G>((uint32_t)lane).hs_lds = hs_lds;
        wave_sync();

        // =============================== phase D: Huffman streams -> literal scratch (lane = (file, stream))
        uint32_t lit_bad = 0; // per lane: a stream of file `lane / LPF` failed
        {
            // One stream by one lane.  The unread bits sit MSB-aligned in `cur` (`av` of them are real window bits); the window
            // is re-read from the staged bytes (or from HBM) when fewer than 22 are left; two symbols (<= 22 bits) per step.
            auto one_stream = [&](uint32_t f, uint32_t st, auto staged_c) -> bool {
                constexpr bool STAGED = decltype(staged_c)::value;
                const FileLds& F = fl<G>(f);
                const uint32_t rel = F.s_off[st] - F.s_off[0];
                const uint8_t* sp = (const uint8_t*)(uintptr_t)F.src + F.s_off[st];
                const uint32_t lbase = F.hs_lds + rel;
                const bool in_lds = F.hs_lds != 0;
                const uint32_t sl = F.s_len[st], seg = (F.nlit + 3) / 4;
                const uint32_t nsym = F.streams == 1 ? F.nlit : (st < 3 ? seg : F.nlit - 3 * seg);
                uint8_t* out = lit_base + (size_t)f * a.lit_stride + (F.streams == 1 ? 0u : st * seg);
                const uint32_t L = F.huf_log, tab = F.huf_off;
                if (sl == 0) return false;
                const uint32_t last = (STAGED || in_lds) ? L8(lbase + sl - 1) : gu8(sp + sl - 1);
                if (last == 0) return false;
                int32_t h = (int32_t)((sl - 1) * 8) + hibit32(last); // unread bits
                // the 64 bits below the read head, MSB-aligned; bits below the stream's start read as zero (A.4: the last symbols
                // may peek past it).  Every stream is preceded by >= 8 readable bytes (header bytes of its file / the zero pad).
                auto window = [&](int32_t hh, int32_t& av) -> uint64_t {
                    int32_t b = (hh - 1) >> 3;
                    b = b < -1 ? -1 : b;
                    uint64_t w;
                    if (STAGED || in_lds) w = lds_u64(lbase + (uint32_t)(b + 9) - 16);
                    else w = gu64(sp + (b - 7));
                    const uint32_t sh = (uint32_t)(8 * (b + 1) - hh) & 63;
                    w <<= sh;
                    av = 64 - (int32_t)sh;
                    const uint64_t keep = hh >= 64 ? ~0ull : (hh <= 0 ? 0ull : ~0ull << (64 - hh));
                    return w & keep;
                };
                int32_t av;
                uint64_t cur = window(h, av);
                uint32_t k = 0, acc = 0;
                const uint32_t shL = 64 - L;
                for (; k + 2 <= nsym; k += 2) {
                    const uint32_t e0 = L16(tab + 2 * (uint32_t)(cur >> shL));
                    cur <<= (e0 >> 8);
                    const uint32_t e1 = L16(tab + 2 * (uint32_t)(cur >> shL));
                    cur <<= (e1 >> 8);
                    const int32_t used = (int32_t)((e0 >> 8) + (e1 >> 8));
                    av -= used; h -= used;
                    acc |= ((e0 & 0xFF) | ((e1 & 0xFF) << 8)) << (8 * (k & 2));
                    if (k & 2) { gs32(out + k - 2, acc); acc = 0; }
                    if (av < 22) cur = window(h, av);
                }
                // the tail: what is left of a group of four, and an odd last symbol
                if (k & 2) { gs8(out + k - 2, acc & 0xFF); gs8(out + k - 1, (acc >> 8) & 0xFF); }
                if (k < nsym) {
                    const uint32_t e0 = L16(tab + 2 * (uint32_t)(cur >> shL));
                    gs8(out + k, e0 & 0xFF);
                    h -= (int32_t)(e0 >> 8);
                }
                return h == 0; // consumed exactly
            };
            // (wave-uniform choice: the staged form has no HBM load in its loop at all)
            const bool all_staged = __ballot(live && lit_type >= 2 && hs_lds == 0) == 0;
            if (LPF == 4) {
                const uint32_t f = (uint32_t)lane >> 2, st = (uint32_t)lane & 3;
                if (st < fl<G>(f).streams) {
                    const bool r = all_staged ? one_stream(f, st, std::true_type{}) : one_stream(f, st, std::false_type{});
                    if (!r) lit_bad = 1;
                }
            } else {
                const uint32_t f = (uint32_t)lane;
                const uint32_t ns = fl<G>(f).streams;
                for (uint32_t st = 0; st < ns; st++) {
                    const bool r = all_staged ? one_stream(f, st, std::true_type{}) : one_stream(f, st, std::false_type{});
                    if (!r) { lit_bad = 1; break; }
                }
            }
        }
        if (LPF == 4) { // a failed stream condemns its file
            const uint64_t badm = __ballot(lit_bad != 0);
            if (lane < G && ((badm >> (4 * lane)) & 0xF)) { ok = false; live = false; }
        } else if (lit_bad) { ok = false; live = false; }
        // RLE literals: the scratch is filled with the byte (lane = file)
        if (live && lit_type == 1) {
            const uint32_t v = gu8(src + lit_off) * 0x01010101u;
            uint8_t* out = lit_base + (size_t)lane * a.lit_stride;
            for (uint32_t k = 0; k < nlit; k += 4) gs32(out + k, v); // (slack past nlit)
        }
        SSTAMP(3);

        // =============================== staging: the sequences section headers -> LDS (the Huffman tables are dead)
        wave_sync();
        {
            const uint32_t f = (uint32_t)lane / LPF, q = (uint32_t)lane % LPF;
            const uint64_t lv = __ballot(live);
            const uint32_t soff = (uint32_t)__shfl((int)seq_off, (int)f), slen = (uint32_t)__shfl((int)seq_len, (int)f);
            if ((lv >> f) & 1) {
                // kSeqStageB bytes per file (nbSeq, modes and three normalized-count descriptions take <= ~150; G = 64 needs
                // only nbSeq, modes and the RLE symbols: 16), 8-byte pieces;
                // nothing is read past the section's end + input padding
                const uint8_t* sp = (const uint8_t*)(uintptr_t)fl<G>(f).src + soff;
                const uint32_t dst_off = (G == 16 ? LY::main_off(f) : stage64_off + f * LY::kSeqStageB);
                for (uint32_t k = q * 8; k < (G == 16 ? kSeqStage : 16u); k += LPF * 8) { const uint64_t v = k < slen + 8 ? gu64(sp + k) : 0ull; __builtin_memcpy(lds + dst_off + k, &v, 8); }
            }
        }
        wave_sync();

        // =============================== phase E: sequences section header (lane = file)
        uint32_t nseq = 0, bs_off = 0, bs_len = 0;
        uint32_t tabL = 0, tabO = 0, tabM = 0, alL = 0, alO = 0, alM = 0;
        uint32_t modes3 = 0, rle_syms = 0, nsyms = 0, als = 0, tab_bytes = 0;
        if (live) {
            bool good = false;
            do {
                const uint32_t sp = (G == 16 ? LY::main_off((uint32_t)lane) : stage64_off + (uint32_t)lane * LY::kSeqStageB); // the staged header
                const uint32_t stage_n = G == 16 ? kSeqStage : 16u;
                const uint32_t staged = seq_len < stage_n - 8 ? seq_len : stage_n - 8;
                const uint64_t w = lds_u64(sp);
                uint32_t p = 1;
                nseq = (uint32_t)w & 0xFF;
                if (nseq > 0x7F) {
                    if (nseq == 0xFF) { if (p + 2 > seq_len) break; nseq = ((uint32_t)(w >> 8) & 0xFFFF) + 0x7F00; p = 3; }
                    else { if (p + 1 > seq_len) break; nseq = ((nseq - 0x80) << 8) + ((uint32_t)(w >> 8) & 0xFF); p = 2; }
                }
                if (nseq == 0) { good = p == seq_len; break; }
                if (nseq > kMaxSeq - 1 || p + 1 > seq_len) break;
                const uint32_t modes = (uint32_t)(w >> (8 * p)) & 0xFF;
                p++;
                if (modes & 3) break;
                const uint32_t ao = LY::aux_off((uint32_t)lane);
                bool tbad = false;
                uint32_t used_entries = 0;
                const uint32_t main_dw = LY::main_off((uint32_t)lane) / 4;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const uint32_t m = (modes >> (6 - 2 * t)) & 3;
                    const int max_log = t == 1 ? 8 : 9, max_sym = t == 0 ? 35 : (t == 1 ? 31 : 52);
                    uint32_t tab = 0, al = 0, rs = 0, ns = 0;
                    if (m == 0) { tab = kOffPredef / 4 + (t == 0 ? 0u : (t == 1 ? 64u : 96u)); al = t == 1 ? 5u : 6u; }
                    else if (m == 1) {
                        if (p + 1 > seq_len || p >= staged) { tbad = true; break; }
                        rs = L8(sp + p); p++;
                        if (rs > (uint32_t)max_sym) { tbad = true; break; }
                        tab = main_dw + used_entries; used_entries += 1; al = 0;
                    } else if (m == 2) {
                        if (G != 16 || p >= staged) { tbad = true; break; } // (G = 64: no room for private tables in this layout)
                        const uint32_t noff = ao + (t == 0 ? 0u : (t == 1 ? 72u : 136u));
                        const uint32_t avail = seq_len - p < staged - p ? seq_len - p : staged - p; // a description that runs past the staged bytes fails here: handed on
                        const uint32_t used = read_ncount_lane(sp + p, avail, max_log, max_sym, max_sym + 1, noff, ns, al);
                        if (used == 0) { tbad = true; break; }
                        p += used;
                        tab = main_dw + used_entries; used_entries += 1u << al;
                    } else {
                        if (!(with_d && di.formatted)) { tbad = true; break; }
                        tab = dict_off / 4 + (t == 0 ? 0u : (t == 1 ? 1024u : 512u)); al = t == 0 ? di.al[0] : (t == 1 ? di.al[1] : di.al[2]);
                    }
                    if (t == 0) { tabL = tab; alL = al; } else if (t == 1) { tabO = tab; alO = al; } else { tabM = tab; alM = al; }
                    modes3 |= m << (2 * t); rle_syms |= rs << (8 * t); nsyms |= ns << (8 * t); als |= al << (8 * t);
                }
                if (tbad || used_entries * 4 > LY::kMain) break;
                if (p >= seq_len) break; // the bitstream needs at least one byte
                bs_off = seq_off + p; bs_len = seq_len - p;
                tab_bytes = used_entries * 4;
                good = true;
            } while (false);
            if (!good) { ok = false; live = false; }
        }
        if (lane < G) {
            FileLds& F = fl<G>((uint32_t)lane);
            F.live = live ? 1u : 0u; F.nseq = live ? nseq : 0u;
            for (int t = 0; t < 3; t++) { F.mode[t] = (uint8_t)((modes3 >> (2 * t)) & 3); F.al[t] = (uint8_t)(als >> (8 * t)); F.nsym[t] = (uint8_t)(nsyms >> (8 * t)); F.rle[t] = (uint8_t)(rle_syms >> (8 * t)); }
            F.tab[0] = tabL; F.tab[1] = tabO; F.tab[2] = tabM;
        }
        wave_sync();
        SSTAMP(4);

        // =============================== phase F: FSE decode tables (lane = (file, table); the staged header is dead)
        {
            uint32_t tb_bad = 0;
            auto one_table = [&](uint32_t f, int t) -> bool {
                const FileLds& F = fl<G>(f);
                if (!F.live || !F.nseq) return true;
                const uint32_t m = F.mode[t];
                if (m == 1) {
                    const uint32_t s = F.rle[t];
                    const uint32_t extra = t == 0 ? L32(kOffLLCode + 4 * s) >> 24 : (t == 1 ? s : L32(kOffMLCode + 4 * s) >> 24);
                    L32(4 * F.tab[t]) = fse_entry(0, 0, s, extra);
                } else if (m == 2) {
                    const uint32_t noff = LY::aux_off(f) + (t == 0 ? 0u : (t == 1 ? 72u : 136u));
                    return build_fse_lane(4 * F.tab[t], noff, F.nsym[t], F.al[t], t);
                }
                return true;
            };
            if (LPF == 4) {
                const uint32_t f = (uint32_t)lane >> 2;
                const int t = lane & 3;
                if (t < 3 && !one_table(f, t)) tb_bad = 1;
                const uint64_t badm = __ballot(tb_bad != 0);
                if (lane < G && ((badm >> (4 * lane)) & 0xF)) { ok = false; live = false; }
            } else {
                for (int t = 0; t < 3; t++) if (!one_table((uint32_t)lane, t)) tb_bad = 1;
                if (tb_bad) { ok = false; live = false; }
            }
        }
        wave_sync();
        SSTAMP(5);

        // =============================== staging: the sequence bitstreams -> LDS, where they fit (lane = (file, part))
        // G = 16: behind the file's tables in its slot; G = 64: in the file's 256-byte stage.  16 zero bytes in front: a window
        // that reaches below the stream's start reads zeros.  A stream that does not fit is read from HBM (a round trip per sequence).
        uint32_t bs_lds = 0; // LDS byte offset of stream byte 0, or 0
        if (live && nseq) {
            const uint32_t room0 = G == 16 ? LY::main_off((uint32_t)lane) + ((tab_bytes + 7) & ~7u) : stage64_off + (uint32_t)lane * LY::kSeqStageB;
            const uint32_t room1 = G == 16 ? LY::main_off((uint32_t)lane) + LY::kMain : room0 + LY::kSeqStageB;
            if (room0 + 16 + bs_len + 8 <= room1) bs_lds = room0 + 16;
        }
        {
            const uint32_t f = (uint32_t)lane / LPF, q = (uint32_t)lane % LPF;
            const uint32_t to = (uint32_t)__shfl((int)bs_lds, (int)f), from = (uint32_t)__shfl((int)bs_off, (int)f), len = (uint32_t)__shfl((int)bs_len, (int)f);
            if (to) {
                const uint8_t* sp = (const uint8_t*)(uintptr_t)fl<G>(f).src + from;
                if (q == 0) { const uint64_t z = 0; __builtin_memcpy(lds + to - 16, &z, 8); __builtin_memcpy(lds + to - 8, &z, 8); }
                uint32_t k = q * 8;
                for (; k + 3 * LPF * 8 < len; k += 4 * LPF * 8) { // four loads in flight
                    const uint64_t v0 = gu64(sp + k), v1 = gu64(sp + k + LPF * 8), v2 = gu64(sp + k + 2 * LPF * 8), v3 = gu64(sp + k + 3 * LPF * 8);
                    __builtin_memcpy(lds + to + k, &v0, 8); __builtin_memcpy(lds + to + k + LPF * 8, &v1, 8);
                    __builtin_memcpy(lds + to + k + 2 * LPF * 8, &v2, 8); __builtin_memcpy(lds + to + k + 3 * LPF * 8, &v3, 8);
                }
                for (; k < len; k += LPF * 8) { const uint64_t v = gu64(sp + k); __builtin_memcpy(lds + to + k, &v, 8); } // (reads <= 7 bytes past the stream: input padding; writes stay below room1)
            }
        }
        wave_sync();

        // =============================== phase G: FSE state walk + repeat offsets + execute (lane = file), four sequences per step
        // Software pipeline.  Per step: (1) the HBM requests of the step's four sequences, in straight-line code without a
        // branch around any load -- 16 literal bytes per sequence (up to 48 where the run is longer) and 16 / 32 bytes of every
        // match whose source is older than the step's output or lies in the dictionary; (2) the WALK of the next four
        // sequences -- three table reads and one window of the bitstream per sequence, all LDS, no branch on its data --
        // which takes longer than the requests' round trip; (3) the step's stores, in order (whole 16-byte pieces: what
        // spills past a piece is rewritten by the piece behind it).  A match that reads what its own step wrote, literal runs
        // over 48 and matches over 32 bytes, and the last steps before the destination's end take the sequential path.
        auto walk_execute = [&](auto staged_c) {
            constexpr bool STAGED = decltype(staged_c)::value; // every bitstream of the group sits in LDS: no HBM load in the walk
            const uint8_t* const sp = src + bs_off;
            const bool in_lds = bs_lds != 0;
            auto window = [&](int32_t hh) -> uint64_t { // the 8 stream bytes that end with the byte holding bit hh - 1
                int32_t b = (hh - 1) >> 3;
                b = b < -1 ? -1 : b; // (an over-read stream: flagged below; the read stays inside the zero pad / the file's header bytes)
                uint64_t w;
                if (STAGED || in_lds) w = lds_u64(bs_lds + (uint32_t)(b + 9) - 16);
                else w = gu64(sp + (b - 7));
                return w;
            };
            bool bad = false;
            int32_t h = 0;
            uint32_t sL = 0, sO = 0, sM = 0;
            uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
            uint64_t W = 0;
            if (nseq) {
                const uint32_t lastb = (STAGED || in_lds) ? L8(bs_lds + bs_len - 1) : gu8(sp + bs_len - 1);
                bad = lastb == 0;
                h = (int32_t)((bs_len - 1) * 8) + hibit32(lastb | 1u);
                const uint32_t need = alL + alO + alM; // <= 26
                bad |= h < (int32_t)need;
                const int32_t b = (h - 1) >> 3;
                uint64_t cur = window(h) << ((8 * (b + 1) - h) & 63);
                sL = alL ? (uint32_t)(cur >> (64 - alL)) : 0u; cur <<= alL;
                sO = alO ? (uint32_t)(cur >> (64 - alO)) : 0u; cur <<= alO;
                sM = alM ? (uint32_t)(cur >> (64 - alM)) : 0u;
                h -= (int32_t)need;
                if (with_d && di.formatted) { rep0 = di.rep[0]; rep1 = di.rep[1]; rep2 = di.rep[2]; }
                W = window(h);
            }
            struct Seq4 { uint32_t ll[4], ml[4], off[4]; };
            auto walk4 = [&](uint32_t i0, Seq4& q) { // sequences i0 .. i0 + 3 (zero past the last one)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    q.ll[k] = 0; q.ml[k] = 0; q.off[k] = 0;
                    const uint32_t i = i0 + (uint32_t)k;
                    if (i < nseq) {
                        int32_t b = (h - 1) >> 3;
                        b = b < -1 ? -1 : b;
                        const uint32_t sh = (uint32_t)(8 * (b + 1) - h) & 63;
                        const uint64_t cur = W << sh;
                        const uint32_t avail = 64 - sh; // >= 57 while the stream lasts
                        const uint32_t eL = L32(4 * (tabL + sL)), eO = L32(4 * (tabO + sO)), eM = L32(4 * (tabM + sM));
                        const uint32_t xL = eL >> 22, xO = eO >> 22, xM = eM >> 22;
                        const uint32_t nbL = (eL >> 10) & 15, nbO = (eO >> 10) & 15, nbM = (eM >> 10) & 15;
                        const bool lastq = i + 1 == nseq;
                        const uint32_t tot_x = xL + xM + xO, tot_s = lastq ? 0u : nbL + nbM + nbO, tot = tot_x + tot_s;
                        const int32_t hn = h - (int32_t)tot;
                        const uint64_t Wn = window(hn); // the next sequence's window: in flight from here on
                        uint32_t vO, vM, vL, bL, bM, bO;
                        if (__builtin_expect(tot <= avail, 1)) { // from the read head down: OF, ML, LL extra bits; LL, ML, OF state bits
                            const uint64_t Y = tot ? cur >> (64 - tot) : 0ull;
                            const uint32_t y = (uint32_t)Y;
                            bO = y & ((1u << nbO) - 1);
                            bM = (y >> nbO) & ((1u << nbM) - 1);
                            bL = (y >> (nbO + nbM)) & ((1u << nbL) - 1);
                            const uint64_t Y2 = Y >> tot_s; // (tot_s <= 26)
                            vL = (uint32_t)Y2 & ((1u << xL) - 1);
                            vM = (uint32_t)(Y2 >> xL) & ((1u << xM) - 1);
                            vO = (uint32_t)((Y2 >> (xL + xM)) & ((1ull << xO) - 1));
                        } else { // more than a window's worth of extra bits (about one sequence in hundreds): three looks
                            vO = xO ? (uint32_t)(cur >> (64 - xO)) : 0u;
                            const int32_t h2 = h - (int32_t)xO;
                            const int32_t b2 = (h2 - 1) >> 3;
                            uint64_t c2 = window(h2) << ((uint32_t)(8 * ((b2 < -1 ? -1 : b2) + 1) - h2) & 63);
                            vM = xM ? (uint32_t)(c2 >> (64 - xM)) : 0u; c2 <<= xM;
                            vL = xL ? (uint32_t)(c2 >> (64 - xL)) : 0u;
                            const int32_t h3 = h2 - (int32_t)(xM + xL);
                            const int32_t b3 = (h3 - 1) >> 3;
                            uint64_t c3 = window(h3) << ((uint32_t)(8 * ((b3 < -1 ? -1 : b3) + 1) - h3) & 63);
                            bL = nbL ? (uint32_t)(c3 >> (64 - nbL)) : 0u; c3 <<= nbL;
                            bM = nbM ? (uint32_t)(c3 >> (64 - nbM)) : 0u; c3 <<= nbM;
                            bO = nbO ? (uint32_t)(c3 >> (64 - nbO)) : 0u;
                        }
                        bad |= hn < 0;
                        const uint32_t ofv = (1u << ((eO >> 14) & 0xFF)) + vO;
                        const uint32_t ml = (L32(kOffMLCode + 4 * ((eM >> 14) & 0xFF)) & 0xFFFFFF) + vM;
                        const uint32_t ll = (L32(kOffLLCode + 4 * ((eL >> 14) & 0xFF)) & 0xFFFFFF) + vL;
                        sL = lastq ? sL : (eL & 0x3FF) + bL; sM = lastq ? sM : (eM & 0x3FF) + bM; sO = lastq ? sO : (eO & 0x3FF) + bO;
                        // ---- repeat offsets (A.5), as selects
                        const bool is_rep = ofv <= 3;
                        const uint32_t idx = ofv - 1 + (ll == 0 ? 1u : 0u); // meaningful when is_rep
                        const uint32_t pick = idx == 0 ? rep0 : (idx == 1 ? rep1 : (idx == 2 ? rep2 : rep0 - 1));
                        const uint32_t off = is_rep ? pick : ofv - 3;
                        bad |= off == 0; // (only "rep0 - 1" can be zero)
                        const bool keep = is_rep & (idx == 0), swap1 = is_rep & (idx == 1);
                        const uint32_t n2 = (keep | swap1) ? rep2 : rep1, n1 = keep ? rep1 : rep0, n0 = keep ? rep0 : off;
                        rep2 = n2; rep1 = n1; rep0 = n0;
                        h = hn; W = Wn;
                        // (a sequence that cannot fit the capacity, and whatever a failed walk decodes, is stopped by the step's
                        //  validation: lengths are clamped so that the sums there cannot wrap)
                        q.ll[k] = ll > 0xFFFFFF ? 0xFFFFFFu : ll; q.ml[k] = ml > 0xFFFFFF ? 0xFFFFFFu : ml; q.off[k] = off;
                    }
                }
            };
            const uint8_t* lit = lit_type == 0 ? src + lit_off : lit_base + (size_t)lane * a.lit_stride;
            const uint8_t* const dlim = dst + cap;
            const uint32_t dict_len = with_d ? di.content_len : 0u;
            const uint8_t* const dict_end = with_d ? di.content + di.content_len : nullptr;
            uint32_t lpos = 0, opos = 0;
            bool sbad = false;
            Seq4 cur4, nxt4;
            walk4(0, cur4);
#ifdef MZD_SMALL_STAMPS
            uint64_t tq = 0, tw = 0, ts = 0, t0_ = 0, t1_ = 0;
#define GSTAMP(acc) do { t1_ = __builtin_readcyclecounter(); acc += t1_ - t0_; t0_ = t1_; } while (0)
            t0_ = __builtin_readcyclecounter();
#else
#define GSTAMP(acc)
#endif
            for (uint32_t i = 0; i < nseq; i += 4) {
                uint32_t ll[4], ml[4], off[4], lp[4], op[4];
                uint32_t lp_run = lpos, op_run = opos;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    off[k] = cur4.off[k]; ll[k] = cur4.ll[k]; ml[k] = cur4.ml[k];
                    lp[k] = lp_run; op[k] = op_run;
                    lp_run += ll[k]; op_run += ll[k] + ml[k];
                }
                // validation in stream order: literals left, room in the destination, offset within the history (A.5)
                bool vbad = bad;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    vbad |= (lp[k] + ll[k] > nlit) | (op[k] + ll[k] + ml[k] > cap) | (off[k] > op[k] + ll[k] + dict_len);
                if (vbad) { sbad = true; break; }
                const bool roomy = op_run + 64 <= cap; // whole-width stores spill up to 47 bytes past a piece
                V16 Lw[4][3] = {}, Mw[4][2] = {};
                bool lin[4], pre[4];
                // ---- (1) requests (every address is readable whatever the sequence is: no branch around the first piece)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lin[k] = roomy & (ll[k] <= 48) & (ll[k] != 0);
                    Lw[k][0] = gv16(lit + lp[k]); // (literal buffers and inputs are readable 16 bytes past their end)
                    // (a memory instruction costs the wavefront ~100+ cycles of L1-miss traffic whatever its lanes do: the pieces that
                    //  few sequences need sit behind a branch -- ~5 % of the literal runs are longer than 16 bytes, ~15 % of the matches)
                    if (lin[k] & (ll[k] > 16)) { Lw[k][1] = gv16(lit + lp[k] + 16); Lw[k][2] = gv16(lit + lp[k] + (ll[k] > 32 ? 32u : 16u)); }
                    const uint32_t mpos = op[k] + ll[k]; // output position of the match
                    const bool in_dict = off[k] > mpos;
                    const bool whole_dict = in_dict & (off[k] - mpos >= ml[k]);
                    const bool old = !in_dict & (off[k] >= (mpos - opos) + ml[k]); // the source ends before this step's output begins
                    pre[k] = roomy & (ml[k] <= 32) & (old | whole_dict) & (ml[k] != 0);
                    const uint8_t* mp = whole_dict ? dict_end - (off[k] - mpos) : dst + mpos - off[k];
                    mp = pre[k] ? mp : lit; // (anything readable)
                    Mw[k][0] = gv16(mp); // (dictionary buffers are padded; output reads end below the step's start + 32 <= cap)
                    if (pre[k] & (ml[k] > 16)) Mw[k][1] = gv16(mp + 16);
                }
                GSTAMP(tq);
                // ---- (2) the next step's sequences, in the shadow of the requests
                walk4(i + 4, nxt4);
                GSTAMP(tw);
                // ---- (3) stores, in order
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint8_t* d = dst + op[k];
                    if (lin[k]) { gsv16(d, Lw[k][0]); if (ll[k] > 16) { gsv16(d + 16, Lw[k][1]); gsv16(d + 32, Lw[k][2]); } } // (16 or 48 bytes: a run of 17..32 rewrites [32, 48) with what the pieces behind it bring)
                    else if (ll[k]) copy_lane(d, lit + lp[k], ll[k], dlim);
                    d += ll[k];
                    if (pre[k]) { gsv16(d, Mw[k][0]); if (ml[k] > 16) gsv16(d + 16, Mw[k][1]); }
                    else if (ml[k]) {
                        uint32_t m = ml[k];
                        const uint32_t mpos = op[k] + ll[k];
                        if (off[k] > mpos) { // starts in the dictionary content (logically just before the frame)
                            const uint32_t back = off[k] - mpos;
                            const uint32_t n1 = m < back ? m : back;
                            copy_lane(d, dict_end - back, n1, dlim);
                            d += n1; m -= n1;
                        }
                        if (m) match_lane(d, off[k], m, dlim);
                    }
                }
                lpos = lp_run; opos = op_run;
                cur4 = nxt4;
                GSTAMP(ts);
            }
#ifdef MZD_SMALL_STAMPS
            if (a.stamps && blockIdx.x == 0 && lane == 0 && first_group) { a.stamps[9] = tq; a.stamps[10] = tw; a.stamps[11] = ts; }
#endif
            bad |= sbad | (nseq != 0 && h != 0); // the bitstream must be consumed exactly
            bool good = !bad;
            if (good) {
                const uint32_t rest = nlit - lpos;
                good = rest <= cap - opos;
                if (good) {
                    copy_lane(dst + opos, lit + lpos, rest, dlim);
                    opos += rest;
                    good = !(has_fcs && opos != fcs);
                    out_len = opos;
                }
            }
            if (!good) { ok = false; live = false; }
        };
        {
            const bool all_staged = __ballot(live && nseq != 0 && bs_lds == 0) == 0; // (wave-uniform)
            if (live) { if (all_staged) walk_execute(std::true_type{}); else walk_execute(std::false_type{}); }
        }
        SSTAMP(6);
        if (lane < G) {
            FileLds& F = fl<G>((uint32_t)lane);
            F.out_len = out_len;
            F.live = (ok && has_ck && n != 0) ? 1u : 0u; // to be hashed
        }
        wave_sync();
        SSTAMP(7);

        // =============================== phase H: XXH64 (lane = (file, accumulator))
        {
            uint32_t ck_bad = 0;
            if (LPF == 4) {
                const uint32_t f = (uint32_t)lane >> 2, acc_i = (uint32_t)lane & 3;
                const FileLds& F = fl<G>(f);
                const bool hashing = F.live != 0;
                const uint8_t* p = (const uint8_t*)(uintptr_t)F.dst;
                const uint32_t len = F.out_len, nstripes = hashing ? len / 32 : 0u;
                uint64_t v = acc_i == 0 ? XP1 + XP2 : (acc_i == 1 ? XP2 : (acc_i == 2 ? 0ull : 0ull - XP1));
                const uint8_t* q = p + 8 * acc_i;
                uint32_t s = 0;
                for (; s + 4 <= nstripes; s += 4) { // four loads in flight
                    const uint64_t i0 = gu64(q), i1 = gu64(q + 32), i2 = gu64(q + 64), i3 = gu64(q + 96);
                    v = xround(v, i0); v = xround(v, i1); v = xround(v, i2); v = xround(v, i3);
                    q += 128;
                }
                for (; s < nstripes; s++) { v = xround(v, gu64(q)); q += 32; }
                const int base = lane & ~3;
                const uint64_t v1 = __shfl(v, base), v2 = __shfl(v, base + 1), v3 = __shfl(v, base + 2), v4 = __shfl(v, base + 3);
                if (hashing && acc_i == 0) {
                    uint64_t hh;
                    if (len >= 32) {
                        hh = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
                        hh = xmerge(hh, v1); hh = xmerge(hh, v2); hh = xmerge(hh, v3); hh = xmerge(hh, v4);
                    } else hh = XP5;
                    hh += len;
                    hh = xxh_tail(hh, p + (len / 32) * 32, p + len);
                    const uint32_t stored = gu32((const uint8_t*)(uintptr_t)F.src + F.n - 4);
                    if ((uint32_t)hh != stored) ck_bad = 1;
                }
                const uint64_t badm = __ballot(ck_bad != 0);
                if (lane < G && ((badm >> (4 * lane)) & 0xF)) ok = false;
            } else {
                if (ok && has_ck && n != 0) {
                    const uint8_t* p = dst;
                    const uint32_t len = out_len, nstripes = len / 32;
                    uint64_t v1 = XP1 + XP2, v2 = XP2, v3 = 0, v4 = 0ull - XP1;
                    const uint8_t* q = p;
                    for (uint32_t s = 0; s < nstripes; s++) {
                        const uint64_t i0 = gu64(q), i1 = gu64(q + 8), i2 = gu64(q + 16), i3 = gu64(q + 24);
                        v1 = xround(v1, i0); v2 = xround(v2, i1); v3 = xround(v3, i2); v4 = xround(v4, i3);
                        q += 32;
                    }
                    uint64_t hh;
                    if (len >= 32) {
                        hh = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
                        hh = xmerge(hh, v1); hh = xmerge(hh, v2); hh = xmerge(hh, v3); hh = xmerge(hh, v4);
                    } else hh = XP5;
                    hh += len;
                    hh = xxh_tail(hh, q, p + len);
                    if ((uint32_t)hh != gu32(src + n - 4)) ok = false;
                }
            }
        }

        // =============================== the host mirror (DevJob::dst2): the decoded files to the caller's pinned memory, all 64
        // lanes per file, 16 bytes per lane (the bytes were written by other lanes of this wavefront: the sync above)
        {
            uint64_t m = __ballot(have && ok && dst2 != nullptr && out_len != 0);
            if (m) wave_sync();
            while (m) {
                const int fl_ = __builtin_ctzll(m);
                m &= m - 1;
                const FileLds& F = fl<G>((uint32_t)fl_);
                const uint8_t* from = (const uint8_t*)(uintptr_t)F.dst;
                uint8_t* to = (uint8_t*)(uintptr_t)F.dst2;
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)out_len, fl_);
                for (uint32_t o = (uint32_t)lane * 16; o + 16 <= len; o += 1024) gsv16(to + o, gv16(from + o));
                const uint32_t tail = len & ~15u;
                if (tail + (uint32_t)lane < len) gs8(to + tail + lane, gu8(from + tail + lane));
            }
        }

        // =============================== results: done here, or handed to the general drivers
        if (have) {
            if (ok) { a.jobs[job].out_len = out_len; a.jobs[job].status = MZD_OK; }
            else { const uint32_t k = atomicAdd(&a.counter[4], 1u); a.redo_list[k] = job; }
        }
        SSTAMP(8);
        first_group = false;
        wave_sync(); // the LDS slots are rewritten by the next group
    }
}

uint32_t small_lds_bytes(int g, int with_dict) {
    return (g == 64 ? Lay<64>::kDict + 64 * Lay<64>::kSeqStageB : Lay<16>::kDict) + (with_dict ? kDictBytes : 0u);
}

void launch_small(const SmallArgs& a, uint32_t grid, int g, uint32_t lds_at_least, void* stream) {
    uint32_t bytes = small_lds_bytes(g, a.with_dict != 0);
    if (bytes < lds_at_least) bytes = lds_at_least; // (mzd_host.cpp: fewer workgroups per CU, spread over all of them)
    if (g == 64) hipLaunchKernelGGL(mzd_small_kernel<64>, dim3(grid), dim3(64), bytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mzd_small_kernel<16>, dim3(grid), dim3(64), bytes, (hipStream_t)stream, a);
}

} // namespace mzd
