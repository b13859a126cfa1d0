// mzd_host.cpp -- host runtime behind include/mzd.h: device/scratch management, job tables, kernel dispatch
// (small-file kernel, then the general drivers), the chunked PCIe pipeline of the host-pointer entry points,
// round-robin sharding of files over GPUs, and the C++ mirror of the reference's file-handle table (reference
// src/file.rs:10-135) with open/read/release (reference src/main.rs:451-513, 595-599).
//
// All decoding happens in mzd_kernels.hip / mzd_lds.hip; nothing here decodes (there is no CPU fallback:
// without a usable GPU every decode entry point returns MZD_E_DEVICE).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "../../include/mzd.h"
#include "mzd_device.h"

namespace mzd {
void launch_dict_kernel(const uint8_t* dict, uint32_t n, DevDict* out, int32_t* status, void* stream);
#ifdef MZD_EXP_DEVSITE
void devsite_take(uint32_t* out3);
#endif
#ifdef MZD_EXP_PLANDIAG
void plandiag_take(uint32_t* out16);
void plandiag_take_pairs(uint32_t* out16);
#endif
void* decode_kernel_ptr(int tasks);
}

namespace {

using namespace mzd;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { last_hip_error = e_; return MZD_E_DEVICE; } } while (0)
thread_local hipError_t last_hip_error = hipSuccess;

constexpr size_t kAlign = 256;
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kSlots = 4;                       // launches of the host path that may be in flight on one device
constexpr size_t kSmallLitBytes = 512u << 20;   // literal scratch of the small-file kernel: every resident file's literals (shared by the slots)
constexpr uint32_t kMaxDicts = 64;
constexpr size_t kMaxChunks = 16;               // ... and at most this many chunks per call
constexpr size_t kChunkBytes = 24u << 20;       // host path: input + output bytes per pipeline chunk

std::atomic<int> g_small_g{0}, g_small_xg{0}, g_small_nw{0}, g_trace_t2{0}, g_keep_behind{0}, g_resolve{0}, g_pairs{0}, g_small_nd{0}, g_t2_mode{0}; // mzd_debug_host_path 4 / 5 / 7: the small-file kernel's files per wavefront / executed at a time; the host path's timing trace
std::atomic<unsigned> g_small_grid{0};               // mzd_debug_host_path 6: its grid (0: as many wavefronts as the device holds)
constexpr uint32_t kLdsPerCu = 160u * 1024u, kLdsGranule = 1280u; // (a workgroup's LDS is allocated in steps of 320 dwords: tools/micro/lds_granule_micro.hip -- five workgroups of 32 000 bytes share a CU, five of 32 640 do not, and the occupancy API says they do)
std::atomic<int> g_force_driver{0}; // mzd_debug_set_driver: 0 automatic, 1 / 2 that general driver only (no small-file kernel), 3 automatic with the
                                    // small-file kernel for every eligible file however few,
                                    // 4 / 5 block tasks with / without blocks resolved ahead (mzd_k_resolve.h) whatever the launch's size

// What one launch runs on: a stream, a counter block, a range of workgroup slots of the scratch arrays, a share of the
// small-file kernel's literal scratch, the per-launch state of the block-task driver.  The host path keeps up to kSlots of
// them in flight; the device-pointer entry points use all of the device at once (`whole`).
struct Lane {
    hipStream_t stream = nullptr;
    uint32_t* counter = nullptr;  // kCounterWords: the block of the most recent launch (one of `cnt`)
    uint32_t* cnt[2] = {nullptr, nullptr}; // the lane's two blocks alternate: a launch's last workgroup cleans the other one (mzd_kernels.hip)
    unsigned flip = 0;
    uint32_t wg0 = 0, nwg = 0;
    uint32_t wg0_3 = 0, nwg3 = 0;   // ... its share when driver 1 runs workgroups of three wavefronts, five to a CU (mzd_decode_kernel_files3)
    uint8_t* small_lit = nullptr;
    size_t small_lit_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    FileState* fstate = nullptr;   // per file of a launch: what a block task hands to its successor
    TableArea* tables = nullptr;
    ContRecord* ring = nullptr;
    size_t task_cap = 0;
    uint32_t epoch = 0;
    std::mutex submit_mu;          // a launch is a sequence of stream operations: one thread at a time per lane
};

struct Staging { // device (and pinned host) buffers of one host-path call
    uint8_t *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    size_t d_in_cap = 0, d_out_cap = 0, h_in_cap = 0, h_out_cap = 0;
    DevJob* d_jobs = nullptr; DevJob* h_jobs = nullptr; // h_jobs pinned
    uint32_t* d_lists = nullptr; uint32_t* h_lists = nullptr; // 2 entries per job: small list | job list
    size_t jobs_cap = 0;
    hipEvent_t ev[kMaxChunks][4] = {}; // per chunk of a call: inputs there, kernels start / end (timed), results there -- created once
    bool busy = false;
};

struct Device {
    int hip_id = 0;
    int index = 0;
    uint32_t max_wg = 0, cus = 0;
    uint32_t max_wg3 = 0;           // workgroup slots of the three-wavefront build of driver 1 (>= max_wg: the scratch arrays are sized for it)
    uint8_t* lit_scratch = nullptr;
    uint4* seq_scratch = nullptr;
    uint4* walk_scratch = nullptr;
    uint8_t* small_lit = nullptr;
    size_t small_lit_total = 0;
    uint32_t last_lds_lit_stride = 0, last_lds_seq_cap = 0; // geometry of the small-file kernel's scratch in the most recent whole-device launch (mzd_debug_small_scratch)
    uint32_t* resolve_map = nullptr; // kResMapStride words per workgroup slot (mzd_k_resolve.h); null: blocks are never resolved ahead (mzd_config::resolve_ahead)
    DebugSlot* debug = nullptr;
    uint32_t* counters = nullptr; // 2 x (kSlots + 1) blocks of kCounterWords
    Lane lane[kSlots];
    Lane whole;                   // the whole device (its stream is lane 0's: a whole-device launch runs when no lane is busy)
    hipStream_t copy_in = nullptr, copy_out = nullptr; // the host path's transfers: one stream per direction, shared by all calls
    Staging staging[2];
    DevDict* d_dicts = nullptr;
    uint32_t ndicts = 0;          // highest handle in use
    bool dict_used[kMaxDicts] = {};
    void* dict_bufs[kMaxDicts] = {};
    float last_ms = 0.f;
    char last_kernels[96] = "";           // what the most recent whole-device launch ran, dominant kernel first (mzd_last_kernel_name); under names_mu
    std::mutex names_mu;
    void set_kernels(const char* a, const char* b = nullptr) {
        std::lock_guard<std::mutex> lk(names_mu);
        if (b) snprintf(last_kernels, sizeof last_kernels, "%s+%s", a, b); else snprintf(last_kernels, sizeof last_kernels, "%s", a);
    }
    uint32_t* job0_counter = nullptr;      // counter block of the launch that decoded job 0 of the most recent call (mzd_debug_last_block)
    uint32_t job0_snap[kCounterWords] = {}; // ... as it stood when a launch of small files alone was collected (the launch of what it handed on
    bool job0_snap_valid = false;          //     cleans that block): what mzd_debug_counters reports then
    std::atomic<bool> whole_used{false};   // a whole-device launch may still be running on a caller's stream: lanes wait for its end event
    std::mutex t2_submit_mu; // (mzd_debug_host_path 13)
    // resources: lanes and stagings are handed out under `mu`
    std::mutex mu;
    std::condition_variable cv;
    int lane_inflight[kSlots] = {}; // chunks submitted and not yet retired, per lane
    unsigned next_lane = 0;
    bool whole_busy = false;
    int whole_waiting = 0;
    std::mutex dict_mu;
};

std::mutex g_mu;
std::vector<std::shared_ptr<Device>> g_dev;

void free_lane(Lane& l, bool own_stream) {
    hipFree(l.fstate); hipFree(l.tables); hipFree(l.ring);
    if (l.ev0) hipEventDestroy(l.ev0);
    if (l.ev1) hipEventDestroy(l.ev1);
    if (own_stream && l.stream) hipStreamDestroy(l.stream);
}

void free_device(Device& d) {
    hipSetDevice(d.hip_id);
    hipDeviceSynchronize();
    hipFree(d.lit_scratch); hipFree(d.seq_scratch); hipFree(d.walk_scratch); hipFree(d.small_lit); hipFree(d.resolve_map); hipFree(d.debug); hipFree(d.counters); hipFree(d.d_dicts);
    for (void* p : d.dict_bufs) hipFree(p);
    for (auto& s : d.staging) {
        hipFree(s.d_in); hipFree(s.d_out); hipFree(s.d_jobs); hipFree(s.d_lists);
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        if (s.h_jobs) hipHostFree(s.h_jobs);
        if (s.h_lists) hipHostFree(s.h_lists);
        for (auto& c : s.ev) for (hipEvent_t e : c) if (e) hipEventDestroy(e);
    }
    for (auto& l : d.lane) free_lane(l, true);
    free_lane(d.whole, false);
    if (d.copy_in) hipStreamDestroy(d.copy_in);
    if (d.copy_out) hipStreamDestroy(d.copy_out);
}

struct InitCfg { uint32_t max_workgroups = 0; size_t small_scratch_bytes = 0; bool resolve_ahead = true; };

int init_device(Device& d, int hip_id, int index, const InitCfg& cfg) {
    d.hip_id = hip_id; d.index = index;
    HIPCHK(hipSetDevice(hip_id));
    int cus = 0, per_cu = 0, per_cu2 = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, hip_id));
    { int prc = pairs_prepare_device(); if (prc) return prc; } // (two LDS images are more dynamic LDS than a kernel may ask for by default)
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)decode_kernel_ptr(0), kWG, 0));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, (const void*)decode_kernel_ptr(1), kWG, 0));
    { int lrc = lds_prepare_device(); if (lrc) return lrc; } // (the small-file kernels' dynamic-LDS limit: per device)
    per_cu = std::min(per_cu, per_cu2); // the block-task driver needs its whole grid resident
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    d.cus = (uint32_t)cus;
    d.max_wg = (uint32_t)(cus * per_cu);
    if (cfg.max_workgroups) d.max_wg = std::max<uint32_t>(4u * kSlots, std::min<uint32_t>(d.max_wg, cfg.max_workgroups)) / kSlots * kSlots;
    d.small_lit_total = std::max<size_t>(16u << 20, cfg.small_scratch_bytes ? cfg.small_scratch_bytes : kSmallLitBytes) / (kSlots * kAlign) * (kSlots * kAlign);
    d.max_wg3 = std::max<uint32_t>(d.max_wg, (uint32_t)(cus * std::max(1, w3_workgroups_per_cu())) / kSlots * kSlots);
    if (cfg.max_workgroups) d.max_wg3 = d.max_wg; // (a caller that bounds the device memory gets no extra slots)
    HIPCHK(hipMalloc(&d.lit_scratch, (size_t)d.max_wg3 * kLitStride));
    HIPCHK(hipMalloc(&d.seq_scratch, (size_t)d.max_wg3 * kSeqStride * sizeof(uint4)));
    HIPCHK(hipMalloc(&d.walk_scratch, (size_t)d.max_wg3 * kSeqStride * sizeof(uint4)));
    HIPCHK(hipMalloc(&d.small_lit, d.small_lit_total));
    if (cfg.resolve_ahead) HIPCHK(hipMalloc(&d.resolve_map, (size_t)d.max_wg * kResMapStride * sizeof(uint32_t)));
    const size_t debug_bytes = std::max<size_t>((size_t)d.max_wg3 * sizeof(DebugSlot), (2048 + 16 * 3072) * sizeof(uint64_t)); // (diagnostic builds of the small-file kernel keep 1 032 stamps there: mzd_debug_small_stamps)
    HIPCHK(hipMalloc(&d.debug, debug_bytes));
    HIPCHK(hipMemset(d.debug, 0, debug_bytes));
    HIPCHK(hipMalloc(&d.counters, 2 * (kSlots + 1) * kCounterWords * sizeof(uint32_t)));
    HIPCHK(hipMemset(d.counters, 0, 2 * (kSlots + 1) * kCounterWords * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&d.d_dicts, sizeof(DevDict) * kMaxDicts));
    HIPCHK(hipMemset(d.d_dicts, 0, sizeof(DevDict) * kMaxDicts));
    const uint32_t share = d.max_wg / kSlots;
    for (int k = 0; k < kSlots; k++) {
        Lane& l = d.lane[k];
        HIPCHK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        l.cnt[0] = d.counters + (size_t)(2 * k) * kCounterWords; l.cnt[1] = l.cnt[0] + kCounterWords; l.counter = l.cnt[0];
        l.wg0 = (uint32_t)k * share; l.nwg = share;
        l.nwg3 = d.max_wg3 / kSlots; l.wg0_3 = (uint32_t)k * l.nwg3;
        l.small_lit = d.small_lit + (size_t)k * (d.small_lit_total / kSlots); l.small_lit_bytes = d.small_lit_total / kSlots;
        HIPCHK(hipEventCreate(&l.ev0));
        HIPCHK(hipEventCreate(&l.ev1));
    }
    HIPCHK(hipStreamCreateWithFlags(&d.copy_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&d.copy_out, hipStreamNonBlocking));
    Lane& w = d.whole;
    w.stream = d.lane[0].stream;
    w.cnt[0] = d.counters + (size_t)(2 * kSlots) * kCounterWords; w.cnt[1] = w.cnt[0] + kCounterWords; w.counter = w.cnt[0];
    w.wg0 = 0; w.nwg = d.max_wg; w.wg0_3 = 0; w.nwg3 = d.max_wg3; w.small_lit = d.small_lit; w.small_lit_bytes = d.small_lit_total;
    HIPCHK(hipEventCreate(&w.ev0));
    HIPCHK(hipEventCreate(&w.ev1));
    return MZD_OK;
}

// ---- lanes and stagings are handed out under Device::mu ----------------------------------------------------------
// A lane is a stream plus its share of the scratch: launches queued on it run one after the other, so any number of chunks --
// of any number of calls -- may be in flight on it.  What is counted is only whether the device is idle (take_whole).
// holding: the caller has chunks in flight already; it must not wait behind a whole-device request that waits for them.
int lane_begin(Device& d, bool holding) {
    std::unique_lock<std::mutex> lk(d.mu);
    d.cv.wait(lk, [&] { return !d.whole_busy && (holding || !d.whole_waiting); });
    const int k = (int)(d.next_lane++ % kSlots);
    d.lane_inflight[k]++;
    return k;
}
void lane_end(Device& d, int k) {
    { std::lock_guard<std::mutex> lk(d.mu); d.lane_inflight[k]--; }
    d.cv.notify_all();
}
void take_whole(Device& d) {
    std::unique_lock<std::mutex> lk(d.mu);
    d.whole_waiting++;
    d.cv.wait(lk, [&] { if (d.whole_busy) return false; for (int c : d.lane_inflight) if (c) return false; return true; });
    d.whole_waiting--;
    d.whole_busy = true;
}
void give_whole(Device& d) {
    { std::lock_guard<std::mutex> lk(d.mu); d.whole_busy = false; }
    d.cv.notify_all();
}
struct WholeGuard {
    Device& d;
    explicit WholeGuard(Device& dd) : d(dd) { take_whole(d); }
    ~WholeGuard() { give_whole(d); }
};
Staging* take_staging(Device& d) {
    std::unique_lock<std::mutex> lk(d.mu);
    for (;;) {
        for (auto& s : d.staging) if (!s.busy) { s.busy = true; return &s; }
        d.cv.wait(lk);
    }
}
void give_staging(Device& d, Staging* s) {
    { std::lock_guard<std::mutex> lk(d.mu); s->busy = false; }
    d.cv.notify_all();
}

// per-file task state of a lane for launches of up to n files (grow-only)
int ensure_task_state(Lane& l, size_t n) {
    if (n <= l.task_cap) return MZD_OK;
    size_t cap = std::max<size_t>(n + n / 4, 1024);
    hipFree(l.fstate); hipFree(l.tables); hipFree(l.ring);
    l.fstate = nullptr; l.tables = nullptr; l.ring = nullptr; l.task_cap = 0;
    HIPCHK(hipMalloc(&l.fstate, cap * sizeof(FileState)));
    HIPCHK(hipMalloc(&l.tables, cap * sizeof(TableArea)));
    HIPCHK(hipMalloc(&l.ring, (cap + 1) * sizeof(ContRecord)));
    HIPCHK(hipMemset(l.ring, 0, (cap + 1) * sizeof(ContRecord)));
    l.task_cap = cap;
    return MZD_OK;
}

// ---- what a launch decodes with which kernel --------------------------------------------------------------------
// Small files (one frame of one block in the plain case; capacity <= kSmallCap) go to the small-file kernel, one lane per
// file; the others -- and whatever that kernel hands on -- to a general driver: block tasks when some file can have more
// than one block, else a workgroup per file.
struct Plan {
    uint32_t njobs = 0, nsmall = 0, nbig = 0;
    bool with_dict = false, multi = false;
    uint32_t lit_stride = 0;   // literal scratch per small file: largest capacity + 64
    int lds_g = 4, lds_xg = 4; // files per wavefront of the small-file kernel (mzd_lds.hip), and how many of them it executes at a time
    int lds_nw = 1;            // its wavefronts per workgroup (2: a helper wavefront parses the sequence headers beside the Huffman phases)
    int lds_nd = 1;            // decoding wavefronts per workgroup around ONE dictionary image (launches that name a single dictionary)
    uint32_t lds_tab = 0, lds_comp = 0, lds_out = 0; // its slot geometry (LdsArgs)
    uint32_t big_tasks = 0;    // workgroups worth launching for the files that are not small
    uint64_t blocks = 0;       // block tasks of those files, estimated from their capacities
    uint32_t nmulti = 0;       // ... how many of them can have more than one block
    bool lpt = false;          // the general driver takes its files through the big list, largest first (make_plan)
};
// lists: [0, njobs) small list (job indices sorted by dictionary), [njobs, 2 njobs) job list of the general driver
Plan make_plan(const DevJob* jobs, size_t njobs, uint32_t* lists, uint32_t max_wg, uint32_t cus) {
    Plan p;
    p.njobs = (uint32_t)njobs;
    const int force = g_force_driver.load(std::memory_order_relaxed);
    uint32_t* small = lists;
    uint32_t* big = lists + njobs;
    bool all_dict = true, one_dict = true;
    uint32_t first_dict = 0;
    uint64_t tasks = 0;
    size_t maxcap = 0, maxbig = 0, maxbigsrc = 0, minbigsrc = ~(size_t)0;
    // The small-file kernel pays off from about two thousand small files on: its launch lasts as long as one group of files
    // (~0.25 ms for 4 KiB files) however few they are, and runs before the general driver, while fewer files fill the general
    // driver's idle workgroup slots for less (profiles/r03_small_policy.txt: 4 KiB files, general driver / small-file kernel:
    // 1 024 files 0.15 / 0.23 ms, 2 048 0.26 / 0.25, 4 096 0.46 / 0.29, 10 000 1.04 / 0.57; 512-byte files cross at 1 024)
    size_t eligible = 0, maxsrc = 0;
    for (size_t i = 0; i < njobs; i++) eligible += jobs[i].dst_cap <= kSmallCap && jobs[i].src_len <= kSmallSrcMax;
    const bool small_ok = force == 3 || (force == 0 && eligible >= 2ull * max_wg); // (3: whenever a file is eligible -- tests)
    for (size_t i = 0; i < njobs; i++) {
        const DevJob& j = jobs[i];
        const bool is_small = small_ok && j.dst_cap <= kSmallCap && j.src_len <= kSmallSrcMax;
        if (is_small) {
            small[p.nsmall++] = (uint32_t)i;
            if (j.dict) { p.with_dict = true; if (!first_dict) first_dict = j.dict; else if (j.dict != first_dict) one_dict = false; } else all_dict = false;
            maxcap = std::max<size_t>(maxcap, j.dst_cap);
            maxsrc = std::max<size_t>(maxsrc, j.src_len);
        } else {
            big[p.nbig++] = (uint32_t)i;
            if (j.dst_cap > kBlockMax) { p.multi = true; p.nmulti++; }
            maxbig = std::max<size_t>(maxbig, j.dst_cap);
            maxbigsrc = std::max<size_t>(maxbigsrc, j.src_len); minbigsrc = std::min<size_t>(minbigsrc, j.src_len);
            p.blocks += 1 + j.dst_cap / kBlockMax;
            if (tasks < max_wg) tasks += 1 + j.src_len / 2048;
        }
    }
    if (p.nsmall) {
        p.lit_stride = (uint32_t)align_up(maxcap + 64, 64);
        // The files of a group run in lockstep, so a group takes as long as its largest file: the list is sorted by size (a
        // counting sort: capacity in steps of 32 bytes), and by dictionary first -- a group shares one dictionary's tables.
        {
            constexpr uint32_t kBuckets = kSmallCap / 32 + 1;
            static thread_local std::vector<uint32_t> cnt, tmp;
            const uint32_t nd = p.with_dict ? kMaxDicts + 2 : 1;
            cnt.assign((size_t)nd * kBuckets + 1, 0);
            auto key = [&](uint32_t i) -> uint32_t {
                const uint32_t dk = p.with_dict ? std::min<uint32_t>(jobs[i].dict, kMaxDicts + 1) : 0u;
                return dk * kBuckets + (uint32_t)(jobs[i].dst_cap / 32);
            };
            for (uint32_t k = 0; k < p.nsmall; k++) cnt[key(small[k]) + 1]++;
            for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
            tmp.resize(p.nsmall);
            for (uint32_t k = 0; k < p.nsmall; k++) tmp[cnt[key(small[k])]++] = small[k];
            std::copy(tmp.begin(), tmp.end(), small);
        }
        // the LDS kernel's slots: sized for the launch's largest file.  Files that all name a dictionary bring no tables of their
        // own in the plain case (a file that does is handed on); else 2 KiB hold a 10-bit Huffman table, later 256 FSE entries
        // (three tables of <= 512 sequences), 4 KiB twice that
        p.lds_comp = (uint32_t)align_up(maxsrc + 16, 16);
        p.lds_out = (uint32_t)align_up(maxcap + 16, 16);
        // (all files name a dictionary: no table area of their own is paid for, but what the window leaves free beside the input is
        //  one -- the few records that bring a predefined, RLE or described table then stay in this kernel: cfg5 handed 8 of 50 000
        //  on, and the general driver's launch behind this kernel lasts 0.16 ms however few files it gets)
        p.lds_tab = (p.with_dict && all_dict) ? lds_spare_table_bytes(p.lds_comp, p.lds_out) : (maxcap <= 5120 ? 2048u : 4096u);
        {   // files per wavefront: residency is set by LDS (and by registers: two wavefronts per SIMD), whatever G is; fewer files
            // per wavefront spread a launch's tail better and cost nothing but lane efficiency in the serial phases, which the
            // SIMDs have to spare (measured, cfg4: G = 4 0.52 ms, G = 8 0.64 ms)
            p.lds_g = (p.with_dict && all_dict) ? 8 : 4; // (one dictionary image per wavefront: worth more files per image; cfg5: G = 4 1.03 ms, 8 0.94, 16 1.25)
            p.lds_xg = p.lds_g;
            // A launch of a little more than one round of groups (10 000 files of 4 KiB: 8 192 are resident at G = 4 -- a CU's LDS holds
            // 39 windows of 4.1 KB and 40 would be needed) takes ONE round when the entropy phases run on eight files per wavefront
            // and only four are executed at a time: eight entropy images fit where four windows do (mzd_lds.hip, XG).  The table area
            // is cut to what lets W wavefronts of eight files share a CU (files whose tables need more are handed on); not below
            // 1 792 bytes (a 9-bit Huffman table, 224 FSE entries).
            auto waves_per_cu = [&](int g, int xg, uint32_t tab, int nw = 1) -> uint32_t { // (workgroups per CU)
                const uint32_t lds = (uint32_t)align_up(lds_kernel_bytes(g, xg, p.with_dict, tab, p.lds_comp, p.lds_out, nw), kLdsGranule);
                return lds > kLdsPerCu ? 0u : std::min<uint32_t>(lds_waves_by_registers(g, xg, p.with_dict, nw) / (uint32_t)nw, kLdsPerCu / lds);
            };
            auto one_round = [&](int g, int xg, uint32_t& tab, int nw = 1) -> bool { // does the launch fit ONE round of groups of g files executed xg at a time?
                if (p.with_dict || cus == 0) return false;
                const uint32_t w = (uint32_t)((p.nsmall + (uint64_t)g * cus - 1) / ((uint64_t)g * cus)); // workgroups per CU
                if (w < 1 || w > lds_waves_by_registers(g, xg, 0, nw) / (uint32_t)nw) return false;
                const uint32_t budget = kLdsPerCu / w / kLdsGranule * kLdsGranule;
                const uint32_t fixed = lds_kernel_bytes(g, xg, 0, 0, 0, 0, nw) - (uint32_t)g * lds_kernel_bytes_per_file(0, 0); // the wavefront's tables and the files' records
                if (budget < fixed + (uint32_t)xg * p.lds_out) return false;
                const uint32_t per_file = (budget - fixed) / (uint32_t)g;            // tables + (counts, work, input)
                const uint32_t rest = lds_kernel_bytes_per_file(0, p.lds_comp);      // (counts, work, input)
                if (per_file < rest + 1792u) return false;
                tab = std::min<uint32_t>(p.lds_tab, (per_file - rest) & ~15u);
                return waves_per_cu(g, xg, tab, nw) >= w;
            };
            uint32_t split_tab = 0;
            if (!p.with_dict && cus) {
                // What the shapes cost, measured (tools/small_shapes.py, profiles/r05_small_shapes.txt; JSON files, kernel ms at 10 000 files):
                //   4 KiB: 4/4 0.390 (two rounds), 8/4 0.290, 4/2 0.315, 8/8 0.454 | 2 KiB: 4/4 0.202, 8/8 0.185 | 512 B: 4/4 0.141, 8/8 0.110
                //   8 KiB: 4/4 0.755 (three rounds of 16 files a CU), 4/2 0.697 (two of 28)
                // A CU's wavefronts share its LDS pipeline and, beyond two a SIMD, its issue slots: a launch that fits one round either
                // way is faster on FEWER wavefronts of more files (8/8: the serial phases cost the same for eight files as for four).
                const uint32_t w44 = waves_per_cu(4, 4, p.lds_tab);
                const uint64_t cap44 = (uint64_t)cus * w44 * 4;
                const uint32_t need44 = (uint32_t)((p.nsmall + 4ull * cus - 1) / (4ull * cus)); // wavefronts per CU that hold the launch at 4 / 4
                if (p.nsmall > cap44) { // more than one round at 4 / 4
                    // (8 / 4 first.  4 / 2 -- ten wavefronts of four files per CU, each file executed by 32 lanes -- was built and measured in
                    //  round 5: its groups are shorter (a group's median 281 us against 8 / 4's 254 on one workgroup alone) but ten wavefronts
                    //  share a CU's LDS pipeline and two SIMDs hold three of them: the slowest wavefront ends at 330 us, 8 / 4's at 297
                    //  (tools/lds_wg.py, profiles/r05_lds_wg_*.txt).  It stays for launches 8 / 4 cannot hold in one round.)
                    // (8 / 4 with a HELPER wavefront -- the sequence headers, 48 K of a group's 600 K cycles, parsed on a second wavefront beside the
                    //  Huffman phases -- was built and measured in round 5 and is no faster: 0.290-0.293 ms against 0.2895.  Ten wavefronts a
                    //  CU put two or three on every SIMD, a workgroup's two go to SIMDs 0/2 or 1/3, so decoding wavefronts share SIMDs with each
                    //  other where five one-wavefront workgroups have three SIMDs to themselves -- and both phases are bound by instruction issue;
                    //  168 registers instead of 256 cost 2 % by themselves.  It stays behind mzd_debug_host_path 9 = 2.)
                    if (g_small_nw.load(std::memory_order_relaxed) == 2 && one_round(8, 4, split_tab, 2)) { p.lds_g = 8; p.lds_xg = 4; p.lds_nw = 2; p.lds_tab = split_tab; }
                    else if (one_round(8, 4, split_tab)) { p.lds_g = 8; p.lds_xg = 4; p.lds_tab = split_tab; }
                    else if (one_round(4, 2, split_tab)) { p.lds_g = 4; p.lds_xg = 2; p.lds_tab = split_tab; }
                    else if (const uint64_t cap42 = (uint64_t)cus * waves_per_cu(4, 2, p.lds_tab) * 4; cap42 && 5 * ((p.nsmall + cap42 - 1) / cap42) <= 4 * ((p.nsmall + cap44 - 1) / cap44)) { p.lds_g = 4; p.lds_xg = 2; } // big windows: two windows a wavefront at a time let a fifth wavefront in -- a fifth fewer rounds or better (10 000 x 8 KiB: two rounds instead of three, 0.677 against 0.766 ms)
                    else if (maxcap <= 768 && waves_per_cu(8, 8, p.lds_tab) * 8 > w44 * 4) { p.lds_g = 8; p.lds_xg = 8; } // many rounds of tiny files: more of them resident, half the wavefronts (512 B x 40 000: 0.265 against 0.304 ms; from 1 KiB on 4 / 4 is ahead)
                } else if (need44 > 8 && one_round(8, 8, split_tab) && split_tab == p.lds_tab) { p.lds_g = 8; p.lds_xg = 8; } // one round either way: five wavefronts of eight rather than ten of four
            }
            if (p.with_dict && all_dict && one_dict && cus && g_small_nw.load(std::memory_order_relaxed) != 1) {
                // one dictionary in the launch: several decoding wavefronts share its table image (14 KB), one image a workgroup instead of
                // one a wavefront -- cfg5: five wavefronts of eight records a CU where four fitted (mzd_lds.hip, ND)
                auto files_per_cu = [&](int nd) -> uint32_t {
                    const uint32_t lds = (uint32_t)align_up(lds_kernel_bytes(8, 8, 1, p.lds_tab, p.lds_comp, p.lds_out, 1, nd), kLdsGranule);
                    return lds > kLdsPerCu ? 0u : std::min<uint32_t>(lds_waves_by_registers(8, 8, 1, 1, nd) / (uint32_t)nd, kLdsPerCu / lds) * (uint32_t)nd * 8u;
                };
                uint32_t best = files_per_cu(1);
                for (int nd : {5, 8}) { const uint32_t fc = files_per_cu(nd); if (fc > best) { best = fc; p.lds_nd = nd; } }
                if (const int fnd = g_small_nd.load(std::memory_order_relaxed); (fnd == 1 || fnd == 5 || fnd == 8) && (fnd == 1 || files_per_cu(fnd))) { p.lds_nd = fnd; p.lds_g = 8; p.lds_xg = 8; } // (mzd_debug_host_path 12: the tests' way to these shapes)
            }
            const int dbg_g = g_small_g.load(std::memory_order_relaxed), dbg_xg = g_small_xg.load(std::memory_order_relaxed); // (mzd_debug_host_path 4 / 5)
            if (dbg_g == 4 || dbg_g == 8 || dbg_g == 16) {
                p.lds_nd = 1;
                p.lds_g = dbg_g; p.lds_xg = (dbg_xg == 4 && dbg_g == 8 && !p.with_dict) ? 4 : ((dbg_xg == 2 && dbg_g == 4 && !p.with_dict) ? 2 : dbg_g);
                p.lds_tab = (p.with_dict && all_dict) ? lds_spare_table_bytes(p.lds_comp, p.lds_out) : (maxcap <= 5120 ? 2048u : 4096u);
                p.lds_nw = (g_small_nw.load(std::memory_order_relaxed) == 2 && p.lds_g == 8 && p.lds_xg == 4) ? 2 : 1;
                if (p.lds_g != p.lds_xg && one_round(p.lds_g, p.lds_xg, split_tab, p.lds_nw)) p.lds_tab = split_tab;
            }
            while (p.lds_g > 4 && lds_kernel_bytes(p.lds_g, p.lds_xg, p.with_dict, p.lds_tab, p.lds_comp, p.lds_out, p.lds_nw, p.lds_nd) > kLdsPerCu) { p.lds_g /= 2; p.lds_xg = p.lds_g; p.lds_nw = 1; p.lds_nd = 1; }
        }
    }
    // Multi-block files in a launch that fills the machine many times over: block tasks keep a workgroup slot waiting while a file's
    // blocks are copied one after the other (the blocks of a 1 MiB file hold eight slots for the time of eight copies), so such a launch
    // gives every file to ONE workgroup (driver 1 walks a file's blocks in order: tables and state stay in LDS, nothing is handed over)
    // and hands the files out largest first, which keeps the launch's tail short.  Only when no file's chain is longer than a slot's
    // fair share of the launch (else the longest file would be the launch: one 64 MiB file is 500 blocks).  cfg4lu (10 000 files,
    // 4 KiB .. 1 MiB): 12.05 -> 9.96 ms; the same through driver 1 in the caller's order: 13.35 ms (tools/lpt_order.py).
    // (Single-block files in such a launch -- at least two per slot -- are ordered too, by their COMPRESSED size: a block's time is
    //  its sequence count, which the input's size follows more closely than the output's.  A mix of seven data classes, 8 000 x 128 KiB
    //  (cfg3's mix at eight times its size): 148 -> 164 GiB/s.)
    const bool lpt_multi = p.multi && force == 0 && (uint64_t)(1 + maxbig / kBlockMax) * max_wg <= p.blocks;
    const bool lpt_single = !p.multi && force == 0 && p.nbig >= 2ull * max_wg && maxbigsrc > minbigsrc + minbigsrc / 2; // (files of like size: nothing to gain, cfg2x8 -1 %)
    if (lpt_multi || lpt_single) {
        p.multi = false; p.lpt = true;
        constexpr uint32_t kSizeBuckets = 4096; // size in steps of 4 KiB (single-block launches: input size in steps of 64 bytes), the rest in the last
        static thread_local std::vector<uint32_t> cnt, tmp;
        cnt.assign(kSizeBuckets + 1, 0);
        auto key = [&](uint32_t i) -> uint32_t { return kSizeBuckets - 1 - (uint32_t)std::min<size_t>(lpt_multi ? jobs[i].dst_cap >> 12 : jobs[i].src_len >> 6, kSizeBuckets - 1); };
        for (uint32_t k = 0; k < p.nbig; k++) cnt[key(big[k]) + 1]++;
        for (size_t k = 1; k < cnt.size(); k++) cnt[k] += cnt[k - 1];
        tmp.resize(p.nbig);
        for (uint32_t k = 0; k < p.nbig; k++) tmp[cnt[key(big[k])]++] = big[k];
        std::copy(tmp.begin(), tmp.end(), big);
    }
    if (force == 1) p.multi = false;
    if (force == 2 || force == 4 || force == 5) p.multi = true;
    p.big_tasks = (uint32_t)std::min<uint64_t>(tasks, max_wg);
    return p;
}

// enqueue one launch on lane `l`: [small-file kernel] + general driver, timed by the lane's events
// handed_on (device word, or null): a launch of small files alone ends with the small-file kernel -- the general driver's launch
// behind it, 19 us of a 10 000-file launch that hands nothing on, is left out; the kernel's last wavefront stores the number of
// files it handed on there and the caller decodes those when it collects (redo_handed_on)
int enqueue(Device& d, Lane& l, hipStream_t s, DevJob* d_jobs, const Plan& p, const uint32_t* d_lists, hipEvent_t ev0, hipEvent_t ev1, uint32_t* handed_on = nullptr) {
    const uint32_t njobs = p.njobs;
    KernelArgs ka;
    l.counter = l.cnt[l.flip]; l.flip ^= 1u; // (clean: the lane's previous launch zeroed it; this one zeroes l.cnt[l.flip])
    ka.jobs = d_jobs; ka.njobs = njobs; ka.counter = l.counter; ka.counter_next = l.cnt[l.flip];
    ka.lit_scratch = d.lit_scratch; ka.seq_scratch = d.seq_scratch; ka.walk_scratch = d.walk_scratch;
    ka.dicts = d.d_dicts; ka.ndicts = d.ndicts; ka.debug = d.debug; ka.job_slot0 = l.counter + 1;
    ka.job_list = nullptr; ka.nlist_fixed = 0; ka.wg0 = l.wg0;
    const bool use_tasks = p.multi;
    if (use_tasks) {
        int trc = ensure_task_state(l, njobs);
        if (trc) return trc;
        l.epoch++; // ring records of earlier launches never match
        HIPCHK(hipMemsetAsync(l.fstate, 0, (size_t)njobs * sizeof(FileState), s));
    }
    ka.fstate = l.fstate; ka.tables = l.tables; ka.ring = l.ring; ka.ring_cap = (uint32_t)l.task_cap + 1; ka.epoch = l.epoch;
    ka.use_tasks = use_tasks ? 1u : 0u;
    const bool solo = handed_on && p.nsmall && p.nbig == 0;
    // few tasks for the lane's workgroups: the in-order copy stage is the critical path -- resolve blocks ahead (mzd_k_resolve.h)
    const int force = g_force_driver.load(std::memory_order_relaxed);
    ka.resolve_map = d.resolve_map;
    // 1: every task after a file's first resolves ahead -- launches of few multi-block files (chains however many blocks they have: one
    // 64 MiB file is 500 blocks) or of up to two block tasks per workgroup slot; 0: in order -- half as many multi-block files as slots or
    // more (3 / 8 of them): the files themselves are the parallelism, and a block that runs its copier beside its walk costs no byte map; 2: only tasks
    // whose predecessor is still running when they start -- what lies between.  Measured (tools/big_resolve.py, profiles/r05_big_resolve.txt;
    // 1 MiB JSON files, kernel ms, all ahead / behind a running predecessor / in order): 100 files 1.61 / 1.69 / 4.26, 200: 2.54 / 2.52 /
    // 4.41, 400: 4.72 / 4.56 / 4.58, 800: 9.02 / 7.84 / 5.39 (round 4 resolved every task ahead up to eight tasks a slot: 800 files 9.03).
    // Between the two, 5 / 16 .. 15 / 32 of the slots in multi-block files (320 .. 480 of 1 024), every OTHER task resolves ahead (3): the chain of in-order
    // copies is half as long and so is the byte maps' work -- 350 files 3.99 -> 3.81 ms, 400: 4.49 -> 4.12, 450: 4.64 -> 4.45; 300: 3.49 / 3.53, 500: 4.71 / 4.78
    // (profiles/r05_big_resolve.txt).
    ka.resolve = !use_tasks || force == 5 || !d.resolve_map ? 0u : ((force == 4 || p.blocks <= 2ull * l.nwg || p.nmulti <= l.nwg / 4) ? 1u : (p.nmulti >= l.nwg * 15 / 32 ? 0u : (p.nmulti >= l.nwg * 5 / 16 ? 3u : 2u)));
    if (const int fr = g_resolve.load(std::memory_order_relaxed); fr && use_tasks && d.resolve_map) ka.resolve = (uint32_t)(fr - 1); // (mzd_debug_host_path 10: 1 in order, 2 every task ahead, 3 only behind a running predecessor, 4 every other task)
    // Driver 1 with workgroups of THREE wavefronts, five to a CU (mzd_decode_kernel_files3, the MZD_W3 build: the walking, the copying and a
    // wavefront that plans and, in the planner's waits, hashes): five chains of sequences a CU where four ran, at the copier's 128 registers.
    // Built and measured in round 6 (tools/pairs_probe.py, profiles/r06_w3_probe.txt): byte-exact, and no faster -- a file takes 35 % longer
    // beside four others than beside three (4 000 x 128 KiB JSON: 2.85 against 2.64 ms; 8 000: 5.10 against 5.14), so a CU's throughput is
    // what it was: with four files a CU the SIMDs' VALU pipes are 79 % busy (profiles/r05_pmc_cfg2x8_util.txt) and a fifth file finds no
    // room there.  Never chosen by the library; mzd_debug_host_path 11 = 3 runs it (the parity suite does, as driver "1w").
    const int fp = g_pairs.load(std::memory_order_relaxed); // (mzd_debug_host_path 11: 1 four wavefronts a file, 2 two files a workgroup, 3 three wavefronts a file)
    const uint32_t queued = p.nsmall ? p.nbig + std::min<uint32_t>(p.nsmall, 256u) : njobs;
    const bool w3 = !use_tasks && fp == 3 && l.nwg3 > l.nwg;
    const char* const files_kernel = w3 ? "mzd_decode_kernel_files3" : "mzd_decode_kernel_files";
    if (ev0) HIPCHK(hipEventRecord(ev0, s)); // (null: an untimed launch -- mzd_batch_launch_ex)
    uint32_t grid;
    if (p.nsmall) {
        LdsArgs la;
        la.jobs = d_jobs; la.list = d_lists; la.n = p.nsmall; la.counter = l.counter;
        la.redo_list = const_cast<uint32_t*>(d_lists) + njobs + p.nbig;
        la.dicts = d.d_dicts; la.ndicts = d.ndicts;
        la.tab_bytes = p.lds_tab; la.comp_bytes = p.lds_comp; la.out_bytes = p.lds_out;
        la.stamps = reinterpret_cast<uint64_t*>(d.debug); // (diagnostic builds: the first debug slot's first bytes; unused otherwise)
        la.counter_next = solo ? l.cnt[l.flip] : nullptr; la.handed_on = solo ? handed_on : nullptr;
        const uint32_t ngroups = (p.nsmall + (uint32_t)p.lds_g - 1) / (uint32_t)p.lds_g;
        const uint32_t lds = (uint32_t)align_up(lds_kernel_bytes(p.lds_g, p.lds_xg, p.with_dict, p.lds_tab, p.lds_comp, p.lds_out, p.lds_nw, p.lds_nd), kLdsGranule);
        // one wavefront per workgroup; as many as the whole device holds, also for a launch on one of the host path's lanes: this
        // kernel uses none of the per-workgroup scratch the lanes divide, and a chunk of small files that gets a quarter of the wave
        // slots takes four rounds of groups where one would do (cfg4 host -> host: 1.5 -> ms)
        const uint32_t resident = d.cus * std::max<uint32_t>(1u, std::min<uint32_t>(lds_waves_by_registers(p.lds_g, p.lds_xg, p.with_dict, p.lds_nw, p.lds_nd) / (uint32_t)(p.lds_nw * p.lds_nd), kLdsPerCu / lds)); // (workgroups)
        la.lit_stride = p.lit_stride; la.seq_cap = ((p.lit_stride - 64) / 3 + 3) & ~1u; // (even: the array of full records behind the 4-byte ones is 8-byte aligned)
        la.scratch = l.small_lit;
        if (&l == &d.whole) { d.last_lds_lit_stride = la.lit_stride; d.last_lds_seq_cap = la.seq_cap; }
        const size_t per_wave = (size_t)p.lds_g * lds_scratch_per_file(la.lit_stride, la.seq_cap);
        const uint32_t by_scratch = (uint32_t)std::max<size_t>(1, l.small_lit_bytes / per_wave / (size_t)p.lds_nd);
        const uint32_t ngroups_wg = (ngroups + (uint32_t)p.lds_nd - 1) / (uint32_t)p.lds_nd; // (a workgroup's decoding wavefronts take a group each)
        const uint32_t dbg_grid = g_small_grid.load(std::memory_order_relaxed); // (mzd_debug_host_path 6: experiments with fewer resident wavefronts)
        { int lrc = launch_lds(la, std::min(ngroups_wg, std::min(dbg_grid ? dbg_grid : resident, by_scratch)), p.lds_g, p.lds_xg, p.with_dict ? 1 : 0, p.lds_nw, p.lds_nd, s); if (lrc) return lrc; }
        if (&l == &d.whole) {
            char small[48];
            if (p.lds_nw > 1 || p.lds_nd > 1) snprintf(small, sizeof small, "mzd_lds_kernel<%d,%s,%d,%d,%d>", p.lds_g, p.with_dict ? "true" : "false", p.lds_xg, p.lds_nw, p.lds_nd);
            else snprintf(small, sizeof small, "mzd_lds_kernel<%d,%s,%d>", p.lds_g, p.with_dict ? "true" : "false", p.lds_xg);
            const char* big = use_tasks ? "mzd_decode_kernel_tasks" : files_kernel;
            if (solo) d.set_kernels(small);
            else if (p.nbig > p.nsmall) d.set_kernels(big, small);
            else d.set_kernels(small, big); // (the general driver's launch behind it takes what was handed on: usually nothing)
        }
        HIPCHK(hipGetLastError());
        if (solo) { HIPCHK(hipEventRecord(ev1, s)); return MZD_OK; }
        ka.job_list = d_lists + njobs; ka.nlist_fixed = p.nbig;
        grid = std::max<uint32_t>(p.big_tasks, std::min<uint32_t>(p.nbig + std::min<uint32_t>(p.nsmall, 256u), l.nwg));
    } else {
        grid = use_tasks ? std::max<uint32_t>(p.big_tasks, std::min<uint32_t>(njobs, l.nwg)) : njobs;
        if (&l == &d.whole) d.set_kernels(use_tasks ? "mzd_decode_kernel_tasks" : files_kernel);
        if (p.lpt) { ka.job_list = d_lists + njobs; ka.nlist_fixed = p.nbig; } // (largest first; nothing is appended: counter word 4 stays 0)
    }
    grid = std::max<uint32_t>(1u, std::min<uint32_t>(grid, l.nwg));
    // Driver 1 with TWO files a workgroup (mzd_kernels.hip compiled with MZD_PAIRS, mzd_k_walk.h: one wavefront runs both files' sequence
    // chains, the other walking wavefront sleeps): built in round 6 on the premise that the block pipeline is bound by VALU issue, and it
    // is not -- the shared walk takes 11 % of a launch's VALU instructions away (SQ_INSTS_VALU 310 K -> 276 K a 128 KiB JSON file) and the
    // launch is 15-30 % SLOWER (4 000 files: 2.63 -> 3.41 ms): a joint step costs 160 cycles against 151, every ring refill and void group of
    // one file stalls the other's chain, and wavefronts wait on memory and LDS half of their time, on issue an eighth
    // (profiles/r06_pairs_walkstat.txt, r06_pairs_pmc.txt).  Never chosen by the library; mzd_debug_host_path 11 = 2 runs it (the parity
    // suite does, as driver "1p").
    int groups = 1;
    if (!use_tasks && fp == 2) {
        groups = 2;
        grid = std::max<uint32_t>(2u, std::min<uint32_t>((grid + 1u) & ~1u, l.nwg & ~1u)); // (whole workgroups: the odd group would lie outside the lane's scratch slots)
    }
    if (w3) { ka.wg0 = l.wg0_3; grid = std::max<uint32_t>(1u, std::min<uint32_t>(queued, l.nwg3)); }
    if (groups == 2) { launch_decode_pairs(ka, grid, s); if (&l == &d.whole) d.set_kernels("mzd_decode_kernel_pairs"); }
    else if (w3) launch_decode_w3(ka, grid, s);
    else launch_decode(ka, grid, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev1, s));
    return MZD_OK;
}

// What a launch without a general driver behind it handed on (`count` job indices at d_lists + njobs): one launch of driver 1.
int redo_handed_on(Device& d, Lane& l, hipStream_t s, DevJob* d_jobs, uint32_t njobs, const uint32_t* d_lists, uint32_t count) {
    KernelArgs ka;
    l.counter = l.cnt[l.flip]; l.flip ^= 1u;
    ka.jobs = d_jobs; ka.njobs = njobs; ka.counter = l.counter; ka.counter_next = l.cnt[l.flip];
    ka.lit_scratch = d.lit_scratch; ka.seq_scratch = d.seq_scratch; ka.walk_scratch = d.walk_scratch;
    ka.dicts = d.d_dicts; ka.ndicts = d.ndicts; ka.debug = d.debug; ka.job_slot0 = l.counter + 1;
    ka.job_list = d_lists + njobs; ka.nlist_fixed = count; ka.wg0 = l.wg0; // (the list is complete: nothing is appended, word 4 stays 0)
    ka.fstate = l.fstate; ka.tables = l.tables; ka.ring = l.ring; ka.ring_cap = (uint32_t)l.task_cap + 1; ka.epoch = l.epoch;
    ka.use_tasks = 0u; ka.resolve_map = d.resolve_map; ka.resolve = 0u;
    launch_decode(ka, std::max<uint32_t>(1u, std::min<uint32_t>(count, l.nwg)), s);
    HIPCHK(hipGetLastError());
    return MZD_OK;
}

void fill_devjob(DevJob& j, const void* src, size_t src_len, void* dst, size_t cap, uint32_t dict, const Device& d, void* dst2 = nullptr) {
    j.src = (const uint8_t*)src; j.src_len = src_len; j.dst = (uint8_t*)dst; j.dst_cap = cap; j.dst2 = (uint8_t*)dst2;
    j.out_len = 0; j.status = MZD_E_DEVICE;
    j.dict = (dict >= 1 && dict <= kMaxDicts && d.dict_used[dict - 1]) ? dict : (dict ? 0xFFFFFFFFu : 0u); // an unloaded handle: MZD_E_DICT
}

int ensure_staging_jobs(Staging& st, size_t n) {
    if (n <= st.jobs_cap) return MZD_OK;
    size_t cap = std::max<size_t>(n + n / 4, 1024);
    hipFree(st.d_jobs); hipFree(st.d_lists);
    if (st.h_jobs) hipHostFree(st.h_jobs);
    if (st.h_lists) hipHostFree(st.h_lists);
    st.d_jobs = nullptr; st.h_jobs = nullptr; st.d_lists = nullptr; st.h_lists = nullptr; st.jobs_cap = 0;
    HIPCHK(hipMalloc(&st.d_jobs, cap * sizeof(DevJob)));
    HIPCHK(hipHostMalloc(&st.h_jobs, cap * sizeof(DevJob), hipHostMallocDefault));
    HIPCHK(hipMalloc(&st.d_lists, (cap * 2 + 4) * sizeof(uint32_t))); // (+ the word a launch of small files alone stores its hand-on count in)
    HIPCHK(hipHostMalloc(&st.h_lists, cap * 2 * sizeof(uint32_t), hipHostMallocDefault));
    st.jobs_cap = cap;
    return MZD_OK;
}

// jobs carry DEVICE pointers; the whole device, one launch
int run_device_jobs(Device& d, mzd_job* jobs, size_t njobs, hipStream_t s) {
    HIPCHK(hipSetDevice(d.hip_id));
    if (njobs == 0) return MZD_OK;
    if (njobs > 0x7FFFFFF0u) return MZD_E_PARAM;
    Staging* st = take_staging(d); // (same order as the host path: staging first, then execution resources)
    struct Give { Device& d; Staging* s; ~Give() { give_staging(d, s); } } give{d, st};
    WholeGuard g(d);
    int rc = ensure_staging_jobs(*st, njobs);
    if (rc) return rc;
    for (size_t i = 0; i < njobs; i++) fill_devjob(st->h_jobs[i], jobs[i].src, jobs[i].src_len, jobs[i].dst, jobs[i].dst_cap, jobs[i].dict_id, d);
    const Plan p = make_plan(st->h_jobs, njobs, st->h_lists, d.max_wg, d.cus);
    if (!s) s = d.whole.stream;
    HIPCHK(hipMemcpyAsync(st->d_jobs, st->h_jobs, njobs * sizeof(DevJob), hipMemcpyHostToDevice, s));
    if (p.nsmall || p.lpt) HIPCHK(hipMemcpyAsync(st->d_lists, st->h_lists, njobs * 2 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    const bool solo = p.nsmall && p.nbig == 0 && !g_keep_behind.load(std::memory_order_relaxed);
    uint32_t* const d_handed = st->d_lists + 2 * st->jobs_cap;
    rc = enqueue(d, d.whole, s, st->d_jobs, p, st->d_lists, d.whole.ev0, d.whole.ev1, solo ? d_handed : nullptr);
    d.job0_counter = d.whole.counter;
    if (rc) { hipStreamSynchronize(s); return rc; }
    uint32_t handed = 0;
    d.job0_snap_valid = false;
    if (solo) {
        HIPCHK(hipMemcpyAsync(&handed, d_handed, sizeof handed, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(d.job0_snap, d.job0_counter, sizeof d.job0_snap, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipMemcpyAsync(st->h_jobs, st->d_jobs, njobs * sizeof(DevJob), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    d.job0_snap_valid = solo;
    HIPCHK(hipEventElapsedTime(&d.last_ms, d.whole.ev0, d.whole.ev1));
    if (solo && handed) { // (files that were not plain: the general driver, now)
        if (handed > njobs) return MZD_E_DEVICE;
        HIPCHK(hipEventRecord(d.whole.ev0, s));
        rc = redo_handed_on(d, d.whole, s, st->d_jobs, (uint32_t)njobs, st->d_lists, handed);
        if (rc) { hipStreamSynchronize(s); return rc; }
        HIPCHK(hipEventRecord(d.whole.ev1, s));
        HIPCHK(hipMemcpyAsync(st->h_jobs, st->d_jobs, njobs * sizeof(DevJob), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        float more = 0.f;
        HIPCHK(hipEventElapsedTime(&more, d.whole.ev0, d.whole.ev1));
        d.last_ms += more;
    }
    for (size_t i = 0; i < njobs; i++) { jobs[i].out_len = (size_t)st->h_jobs[i].out_len; jobs[i].status = st->h_jobs[i].status; jobs[i].device = d.index; }
    return MZD_OK;
}

// ---- the host path ------------------------------------------------------------------------------------------------
// The staging copies (user buffers <-> pinned memory) are memory-bound host work: split over a few threads (one thread
// moves ~10 GB/s; the PCIe link ~50).  fn(k) handles piece k of [0, n).
std::atomic<unsigned> g_copy_threads{8};        // host threads of the staging copies (mzd_debug_host_path 3)
// A small pool of persistent workers (spawning threads per copy cost more than the copies of a 24 MB chunk and made the
// host path's time jump from call to call).  One loop at a time; a caller that finds the pool busy -- another call's copy --
// runs its loop alone.  Pieces are handed out by a counter; the caller works too.
class CopyPool {
public:
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_work_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run(size_t n, unsigned want, const std::function<void(size_t)>& fn) {
        std::unique_lock<std::mutex> own(run_mu_, std::try_to_lock);
        if (!own.owns_lock() || want <= 1) { for (size_t k = 0; k < n; k++) fn(k); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (th_.size() + 1 < want) th_.emplace_back([this] { worker(); });
            fn_ = &fn; n_ = n; next_.store(0); active_ = 0; helpers_ = std::min<size_t>(want - 1, th_.size()); gen_++;
        }
        cv_work_.notify_all();
        for (size_t k; (k = next_.fetch_add(1)) < n;) fn(k);
        std::unique_lock<std::mutex> lk(mu_);
        helpers_ = 0; // (workers that have not started on this loop stay out)
        cv_done_.wait(lk, [this] { return active_ == 0; });
        fn_ = nullptr;
    }
private:
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || (gen_ != seen && helpers_ > 0); });
            if (stop_) return;
            seen = gen_; helpers_--; active_++;
            const std::function<void(size_t)>* fn = fn_;
            const size_t n = n_;
            lk.unlock();
            for (size_t k; (k = next_.fetch_add(1)) < n;) (*fn)(k);
            lk.lock();
            if (--active_ == 0) cv_done_.notify_all();
        }
    }
    std::mutex run_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> th_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, helpers_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};
CopyPool g_pool;
template <class F>
void parallel_for(size_t n, size_t total_bytes, F fn) {
    unsigned want = total_bytes < (4u << 20) ? 1u : std::min<unsigned>(g_copy_threads.load(std::memory_order_relaxed), std::max(1u, std::thread::hardware_concurrency() / 2));
    if (want <= 1 || n < 2 * want) { for (size_t k = 0; k < n; k++) fn(k); return; }
    g_pool.run(n, want, std::function<void(size_t)>(fn));
}
// one big copy, split by bytes
void parallel_memcpy(uint8_t* d, const uint8_t* s, size_t n) {
    if (n < (4u << 20)) { memcpy(d, s, n); return; }
    const size_t piece = 2u << 20, np = (n + piece - 1) / piece;
    parallel_for(np, n, [=](size_t k) { const size_t o = k * piece; memcpy(d + o, s + o, std::min(piece, n - o)); });
}

// what mzd_host_alloc handed out: [base, base + size), looked up per job (hipPointerGetAttributes costs microseconds)
std::atomic<int> g_direct_chunks{kSlots};
std::mutex g_pin_mu;
std::map<uintptr_t, size_t> g_pinned;
bool in_pinned_registry(const void* p, size_t len) {
    if (len == 0) return true;
    if (!p) return false;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = g_pinned.upper_bound((uintptr_t)p);
    if (it == g_pinned.begin()) return false;
    --it;
    return (uintptr_t)p + len <= it->first + it->second;
}

bool is_pinned_host(const void* p) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// A RUN = consecutive jobs whose buffers lie one behind the other in the caller's memory (inputs: ascending with gaps below
// a page, so that the bytes between them are readable; outputs: exactly adjacent, so that nothing but output bytes is
// ever written).  A run is mirrored 1:1 in the device staging buffer and crosses the link as ONE copy -- straight from /
// into the caller's memory when that is pinned (mzd_host_alloc), else through the pinned staging image.
struct Layout {
    std::vector<size_t> off;       // device offset of job k's buffer
    std::vector<uint32_t> run_of;  // run of job k
    std::vector<size_t> run_dev, run_len; // device offset and byte length of each run
    std::vector<const uint8_t*> run_host;
    size_t total = 0;
};
Layout make_layout(const mzd_job* jobs, const std::vector<size_t>& idx, bool input, size_t tail) {
    Layout L;
    const size_t n = idx.size();
    L.off.resize(n); L.run_of.resize(n);
    const uint8_t* run_host = nullptr; size_t run_dev = 0, run_len = 0;
    bool open = false;
    auto close = [&]() {
        L.run_dev.push_back(run_dev); L.run_len.push_back(run_len); L.run_host.push_back(run_host);
        L.total = align_up(run_dev + run_len + tail, kAlign);
    };
    for (size_t k = 0; k < n; k++) {
        const mzd_job& j = jobs[idx[k]];
        const uint8_t* p = input ? j.src : j.dst;
        const size_t len = input ? j.src_len : j.dst_cap;
        bool joins = false;
        if (open && p && run_host && p >= run_host + run_len) {
            const size_t gap = (size_t)(p - (run_host + run_len));
            joins = input ? gap < 4096 : gap == 0;
        }
        if (!joins) {
            if (open) close();
            run_host = p; run_dev = L.total + (p ? ((uintptr_t)p & 15) : 0); run_len = 0; open = true;
        }
        const size_t rel = p && run_host ? (size_t)(p - run_host) : 0;
        L.off[k] = run_dev + rel;
        L.run_of[k] = (uint32_t)L.run_dev.size();
        run_len = rel + len;
    }
    if (open) close();
    return L;
}

// HOST-pointer jobs `idx` on one device, as a pipeline of chunks: while chunk k decodes, chunk k+1 crosses the link one way
// and chunk k-1 the other (kSlots launches in flight, each on its own stream and its own share of the scratch).
int run_host_jobs(Device& d, mzd_job* jobs, const std::vector<size_t>& idx) {
    const bool trace = g_trace_t2.load(std::memory_order_relaxed) != 0; // diagnostic (mzd_debug_host_path 7): where a call's wall time goes, to stderr
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) { if (trace) fprintf(stderr, "[mzd t2] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count()); };
    HIPCHK(hipSetDevice(d.hip_id));
    const size_t n = idx.size();
    if (n == 0) return MZD_OK;
    if (n > 0x7FFFFFF0u) return MZD_E_PARAM;
    Staging* st = take_staging(d);
    struct Give { Device& d; Staging* s; ~Give() { give_staging(d, s); } } give{d, st};
    const Layout Lin = make_layout(jobs, idx, true, MZD_SRC_PADDING), Lout = make_layout(jobs, idx, false, 16);
    // straight from / into the caller's memory?  Only when it is pinned and the runs are few (a copy call costs microseconds)
    const auto few_runs = [&](const Layout& L) { return L.run_dev.size() <= std::max<size_t>(8, n / 64); };
    auto pinned = [&](const Layout& L, bool input) {
        bool reg = true; // every buffer inside an mzd_host_alloc allocation?  (one lock; consecutive buffers mostly share an allocation)
        {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            uintptr_t lo = 1, hi = 0; // the allocation the previous buffer was found in
            for (size_t k = 0; k < n && reg; k++) {
                const uintptr_t p = (uintptr_t)(input ? (const void*)jobs[idx[k]].src : (const void*)jobs[idx[k]].dst);
                const size_t len = input ? jobs[idx[k]].src_len : jobs[idx[k]].dst_cap;
                if (len == 0) continue;
                if (p >= lo && p + len <= hi) continue;
                auto it = g_pinned.upper_bound(p);
                if (!p || it == g_pinned.begin()) { reg = false; break; }
                --it;
                lo = it->first; hi = it->first + it->second;
                reg = p + len <= hi;
            }
        }
        if (reg) return true;
        if (!few_runs(L)) return false; // memory pinned by other means: asked of the runtime, run by run
        for (size_t r = 0; r < L.run_dev.size(); r++)
            if (L.run_len[r] && !(L.run_host[r] && is_pinned_host(L.run_host[r]) && is_pinned_host(L.run_host[r] + L.run_len[r] - 1))) return false;
        return true;
    };
    mark("layouts");
    // inputs cross the link straight from the caller's memory when that is pinned and the runs are few (a copy call costs
    // microseconds); outputs are written into pinned caller memory by the kernels themselves, wherever the buffers lie
    const bool in_direct = few_runs(Lin) && pinned(Lin, true), out_direct = pinned(Lout, false);
    mark("pinned lookups");
    auto grow_dev = [&](uint8_t*& p, size_t& cap, size_t want) -> int {
        if (want <= cap) return MZD_OK;
        hipFree(p); p = nullptr; cap = 0;
        const size_t c = align_up(want + want / 4, 1 << 20);
        HIPCHK(hipMalloc(&p, c));
        cap = c;
        return MZD_OK;
    };
    auto grow_host = [&](uint8_t*& p, size_t& cap, size_t want) -> int {
        if (want <= cap) return MZD_OK;
        if (p) hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t c = align_up(want + want / 4, 1 << 20);
        HIPCHK(hipHostMalloc(&p, c, hipHostMallocDefault));
        cap = c;
        return MZD_OK;
    };
    int rc;
    if ((rc = grow_dev(st->d_in, st->d_in_cap, Lin.total))) return rc;
    if ((rc = grow_dev(st->d_out, st->d_out_cap, Lout.total))) return rc;
    if (!in_direct && (rc = grow_host(st->h_in, st->h_in_cap, Lin.total))) return rc;
    if (!out_direct && (rc = grow_host(st->h_out, st->h_out_cap, Lout.total))) return rc;
    if ((rc = ensure_staging_jobs(*st, n))) return rc;
    for (size_t k = 0; k < n; k++) {
        const mzd_job& j = jobs[idx[k]];
        // out_direct: the caller's buffers are pinned -- the kernels mirror every file into them while they decode (DevJob::dst2),
        // so the way back over PCIe overlaps the decode instead of following it
        fill_devjob(st->h_jobs[k], st->d_in + Lin.off[k], j.src_len, st->d_out + Lout.off[k], j.dst_cap, j.dict_id, d, out_direct ? j.dst : nullptr);
    }
    mark("job table filled");
    // chunks: at least kSlots when the batch is worth splitting, ~kChunkBytes each, cut at job boundaries
    size_t bytes_total = 0;
    for (size_t k = 0; k < n; k++) bytes_total += jobs[idx[k]].src_len + jobs[idx[k]].dst_cap;
    // (a chunk's launch lasts at least as long as its longest file's block chain: many small chunks of a big batch would add those up)
    size_t nchunks = bytes_total < (8u << 20) ? 1 : std::max<size_t>(kSlots, std::min<size_t>(kMaxChunks, (bytes_total + kChunkBytes - 1) / kChunkBytes));
    if (out_direct) nchunks = std::min<size_t>(nchunks, (size_t)g_direct_chunks.load(std::memory_order_relaxed)); // (nothing to copy back: chunks only let the first kernels start before the last inputs arrive)
    nchunks = std::min(nchunks, n);
    std::vector<size_t> cut{0};
    {
        size_t acc = 0, target = (bytes_total + nchunks - 1) / nchunks;
        for (size_t k = 0; k < n; k++) {
            acc += jobs[idx[k]].src_len + jobs[idx[k]].dst_cap;
            if ((acc >= target && cut.size() < nchunks) || k + 1 == n) { cut.push_back(k + 1); acc = 0; }
        }
    }
    nchunks = cut.size() - 1;
    // the copies of a job range: one per run piece
    auto for_run_pieces = [&](const Layout& L, size_t c0, size_t c1, bool input, const std::function<void(size_t dev_off, const uint8_t* host, size_t len)>& fn) {
        size_t k = c0;
        while (k < c1) {
            const uint32_t r = L.run_of[k];
            size_t e = k;
            while (e + 1 < c1 && L.run_of[e + 1] == r) e++;
            const mzd_job& jl = jobs[idx[e]];
            const size_t lo = L.off[k], hi = L.off[e] + (input ? jl.src_len : jl.dst_cap);
            const mzd_job& jf = jobs[idx[k]];
            fn(lo, input ? jf.src : jf.dst, hi - lo);
            k = e + 1;
        }
    };
    // Per chunk: its inputs on the copy-in stream, its kernels on a lane (round-robin), its outputs on the copy-out stream,
    // tied together by events.
    // Everything is submitted without waiting; then the chunks are retired in order.
    struct Chunk { int lane = -1; hipEvent_t in = nullptr, k0 = nullptr, k1 = nullptr, done = nullptr; };
    std::vector<Chunk> ch(nchunks);
    int result = MZD_OK;
    size_t submitted = 0;
    // what each chunk's launch runs, and the whole job table (+ lists) in one copy ahead of the inputs
    std::vector<Plan> plans(nchunks);
    bool any_small = false;
    for (size_t c = 0; c < nchunks; c++) {
        plans[c] = make_plan(st->h_jobs + cut[c], cut[c + 1] - cut[c], st->h_lists + 2 * cut[c], d.lane[0].nwg, d.cus);
        any_small = any_small || plans[c].nsmall != 0 || plans[c].lpt;
    }
    mark("plans");
    // One chunk with more block tasks than a lane has workgroup slots (a single big file): the whole device instead of a
    // quarter of it (the call then waits until no other launch is in flight, like a call on device pointers).
    constexpr int kWholeLane = -2;
    const bool use_whole = nchunks == 1 && plans[0].blocks > d.lane[0].nwg;
    if (use_whole) { plans[0] = make_plan(st->h_jobs, n, st->h_lists, d.max_wg, d.cus); any_small = plans[0].nsmall != 0 || plans[0].lpt; }
    {
        hipError_t e = hipMemcpyAsync(st->d_jobs, st->h_jobs, n * sizeof(DevJob), hipMemcpyHostToDevice, d.copy_in);
        if (e == hipSuccess && any_small) e = hipMemcpyAsync(st->d_lists, st->h_lists, n * 2 * sizeof(uint32_t), hipMemcpyHostToDevice, d.copy_in);
        if (e != hipSuccess) result = MZD_E_DEVICE;
    }
    const int t2_mode = g_t2_mode.load(std::memory_order_relaxed);
    std::unique_lock<std::mutex> submit_all(d.t2_submit_mu, std::defer_lock);
    if (t2_mode & 1) submit_all.lock();
    for (size_t c = 0; c < nchunks && result == MZD_OK; c++) {
        const size_t c0 = cut[c], c1 = cut[c + 1];
        Chunk& k = ch[c];
        hipError_t e = hipSuccess;
        if (!st->ev[c][0]) { // (the staging's events are made once: creating and destroying 16 per call cost ~0.1 ms)
            e = hipEventCreateWithFlags(&st->ev[c][0], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreate(&st->ev[c][1]);
            if (e == hipSuccess) e = hipEventCreate(&st->ev[c][2]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&st->ev[c][3], hipEventDisableTiming);
        }
        if (e != hipSuccess) { result = MZD_E_DEVICE; break; }
        k.in = st->ev[c][0]; k.k0 = st->ev[c][1]; k.k1 = st->ev[c][2]; k.done = st->ev[c][3];
        int ln = kWholeLane;
        if (use_whole) take_whole(d); else ln = lane_begin(d, submitted != 0);
        k.lane = ln;
        submitted = c + 1;
        Lane& l = use_whole ? d.whole : d.lane[ln];
        const Plan& p = plans[c];
        int erc = MZD_OK;
        // inputs
        if (e == hipSuccess && erc == MZD_OK && !in_direct) {
            size_t bytes = 0;
            for (size_t q = c0; q < c1; q++) bytes += jobs[idx[q]].src_len;
            if (few_runs(Lin)) { // few long runs: big copies split by bytes
                for_run_pieces(Lin, c0, c1, true, [&](size_t dev_off, const uint8_t* host, size_t len) { if (len && host) parallel_memcpy(st->h_in + dev_off, host, len); });
            } else {
                parallel_for(c1 - c0, bytes, [&](size_t r) {
                    const mzd_job& j = jobs[idx[c0 + r]];
                    if (j.src_len && j.src) memcpy(st->h_in + Lin.off[c0 + r], j.src, j.src_len);
                });
            }
            const size_t lo = Lin.off[c0], hi = Lin.off[c1 - 1] + jobs[idx[c1 - 1]].src_len;
            if (hi > lo) e = hipMemcpyAsync(st->d_in + lo, st->h_in + lo, hi - lo, hipMemcpyHostToDevice, d.copy_in);
        } else if (e == hipSuccess && erc == MZD_OK) {
            for_run_pieces(Lin, c0, c1, true, [&](size_t dev_off, const uint8_t* host, size_t len) {
                if (len && e == hipSuccess) e = hipMemcpyAsync(st->d_in + dev_off, host, len, hipMemcpyHostToDevice, d.copy_in);
            });
        }
        if (e == hipSuccess) e = hipEventRecord(k.in, d.copy_in);
        if (e == hipSuccess) { // the kernels
            std::lock_guard<std::mutex> sub(l.submit_mu);
            e = hipStreamWaitEvent(l.stream, k.in, 0);
            if (e == hipSuccess && d.whole_used.load(std::memory_order_relaxed)) e = hipStreamWaitEvent(l.stream, d.whole.ev1, 0); // a device-path launch may still run on a caller's stream
            if (e == hipSuccess) erc = enqueue(d, l, l.stream, st->d_jobs + c0, p, st->d_lists + 2 * c0, k.k0, k.k1);
            if (c == 0) { d.job0_counter = l.counter; d.job0_snap_valid = false; }
        }
        // results and outputs
        if (e == hipSuccess && erc == MZD_OK) e = hipStreamWaitEvent(d.copy_out, k.k1, 0);
        if (e == hipSuccess && erc == MZD_OK) e = hipMemcpyAsync(st->h_jobs + c0, st->d_jobs + c0, (c1 - c0) * sizeof(DevJob), hipMemcpyDeviceToHost, d.copy_out);
        if (e == hipSuccess && erc == MZD_OK) {
            if (!out_direct) {
                const size_t lo = Lout.off[c0], hi = Lout.off[c1 - 1] + jobs[idx[c1 - 1]].dst_cap;
                if (hi > lo) e = hipMemcpyAsync(st->h_out + lo, st->d_out + lo, hi - lo, hipMemcpyDeviceToHost, d.copy_out);
            } // (else: the kernels have written the caller's buffers themselves)
        }
        if (e == hipSuccess && erc == MZD_OK) e = hipEventRecord(k.done, d.copy_out);
        if (e != hipSuccess || erc != MZD_OK) result = erc != MZD_OK ? erc : MZD_E_DEVICE;
    }
    if (result != MZD_OK) { // something could not be submitted: let what is in flight finish, report the failure
        hipStreamSynchronize(d.copy_in);
        for (size_t c = 0; c < submitted; c++) hipStreamSynchronize(ch[c].lane == kWholeLane ? d.whole.stream : d.lane[ch[c].lane].stream);
        hipStreamSynchronize(d.copy_out);
    }
    if (submit_all.owns_lock()) submit_all.unlock();
    mark("all chunks submitted");
    float ms_sum = 0.f;
    for (size_t c = 0; c < nchunks; c++) { // retire in order: wait, hand bytes and results to the caller
        Chunk& k = ch[c];
        if (c < submitted && result == MZD_OK) {
            if (t2_mode & 2) {
                hipError_t q;
                while ((q = hipEventQuery(k.done)) == hipErrorNotReady) std::this_thread::yield();
                if (q != hipSuccess) result = MZD_E_DEVICE;
            } else if (hipEventSynchronize(k.done) != hipSuccess) result = MZD_E_DEVICE;
            float ms = 0.f;
            if (result == MZD_OK && hipEventElapsedTime(&ms, k.k0, k.k1) == hipSuccess) ms_sum += ms;
        }
        if (k.lane >= 0) lane_end(d, k.lane);
        else if (k.lane == kWholeLane) give_whole(d);
        if (c >= submitted || result != MZD_OK) continue;
        const size_t c0 = cut[c], c1 = cut[c + 1];
        size_t bytes = 0;
        for (size_t q = c0; q < c1; q++) bytes += std::min<size_t>(st->h_jobs[q].out_len, jobs[idx[q]].dst_cap);
        parallel_for(c1 - c0, out_direct ? 0 : bytes, [&](size_t r) {
            const size_t q = c0 + r;
            mzd_job& j = jobs[idx[q]];
            j.status = st->h_jobs[q].status;
            j.out_len = (size_t)st->h_jobs[q].out_len;
            j.device = d.index;
            const size_t ncopy = std::min<size_t>(j.out_len, j.dst_cap);
            if (!out_direct && ncopy && j.dst) memcpy(j.dst, st->h_out + Lout.off[q], ncopy);
        });
    }
    mark("retired");
    if (result == MZD_OK) d.last_ms = ms_sum;
    return result;
}

std::shared_ptr<Device> get_device(int i) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (i < 0 || (size_t)i >= g_dev.size()) return nullptr;
    return g_dev[(size_t)i];
}

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

void drop_devices(std::vector<std::shared_ptr<Device>>& devs) { // waits until nobody uses a device, then frees it
    for (auto& d : devs) {
        take_whole(*d);
        { std::unique_lock<std::mutex> lk(d->mu); d->cv.wait(lk, [&] { for (auto& s : d->staging) if (s.busy) return false; return true; }); }
        free_device(*d);
    }
    devs.clear();
}

} // namespace

struct mzd_batch {
    int device;
    DevJob* d_jobs;
    DevJob* h_jobs;
    uint32_t* d_lists; // 2 * njobs entries + the word a launch of small files alone stores its hand-on count in
    size_t njobs;
    Plan plan;
    std::shared_ptr<Device> dev;
    bool solo = false; // the last launch ended with the small-file kernel: what it handed on is decoded by collect
    bool timed = true; // the last launch recorded its start event (mzd_last_kernel_ms)
};

extern "C" {

int mzd_init_ex(const mzd_config* cfg) {
    if (!cfg || cfg->struct_size < offsetof(mzd_config, max_workgroups)) return MZD_E_PARAM;
    auto has = [&](size_t off, size_t sz) { return cfg->struct_size >= off + sz; };
    InitCfg ic;
    if (has(offsetof(mzd_config, max_workgroups), sizeof(uint32_t))) ic.max_workgroups = cfg->max_workgroups;
    if (has(offsetof(mzd_config, small_scratch_bytes), sizeof(size_t))) ic.small_scratch_bytes = cfg->small_scratch_bytes;
    if (has(offsetof(mzd_config, resolve_ahead), sizeof(int))) ic.resolve_ahead = cfg->resolve_ahead != 0;
    // (nothing here touches the environment: the number of hardware queues the HIP runtime maps streams onto, GPU_MAX_HW_QUEUES,
    //  is the application's to set before its first HIP call -- include/mzd.h)
    std::vector<std::shared_ptr<Device>> old;
    { std::lock_guard<std::mutex> lk(g_mu); old.swap(g_dev); }
    drop_devices(old); // (calls in flight on the old devices finish first)
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MZD_E_DEVICE;
    std::vector<int> ids;
    if (!cfg->device_ids || cfg->n_devices <= 0) ids.push_back(0);
    else ids.assign(cfg->device_ids, cfg->device_ids + cfg->n_devices);
    std::vector<std::shared_ptr<Device>> fresh;
    for (int id : ids) {
        if (id < 0 || id >= count) { for (auto& d : fresh) free_device(*d); return MZD_E_PARAM; }
        auto d = std::make_shared<Device>();
        int rc = init_device(*d, id, (int)fresh.size(), ic);
        if (rc) { free_device(*d); for (auto& e : fresh) free_device(*e); return rc; }
        fresh.push_back(std::move(d));
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev = std::move(fresh);
    return MZD_OK;
}

int mzd_init(const int* device_ids, int n) {
    mzd_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device_ids = device_ids; cfg.n_devices = n; cfg.resolve_ahead = 1;
    return mzd_init_ex(&cfg);
}

void mzd_shutdown(void) {
    std::vector<std::shared_ptr<Device>> old;
    { std::lock_guard<std::mutex> lk(g_mu); old.swap(g_dev); }
    drop_devices(old);
}

int mzd_device_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_dev.size();
}

// Pinned host memory every initialised device can copy from / into directly (no staging copy on the host path).
// Diagnostic knob of the host path.  what 2: at most `value` chunks per call when the outputs are mirrored by the kernels.
int mzd_debug_host_path(int device, int what, int value) {
    (void)device;
    if (what == 2) { g_direct_chunks.store(value < 1 ? 1 : value); return MZD_OK; }
    if (what == 3) { g_copy_threads.store(value < 1 ? 1u : (unsigned)value); return MZD_OK; }
    if (what == 4) { g_small_g.store(value); return MZD_OK; }
    if (what == 5) { g_small_xg.store(value); return MZD_OK; }
    if (what == 6) { g_small_grid.store(value < 0 ? 0u : (unsigned)value); return MZD_OK; }
    if (what == 7) { g_trace_t2.store(value); return MZD_OK; }
    if (what == 8) { g_keep_behind.store(value); return MZD_OK; }
    if (what == 9) { g_small_nw.store(value); return MZD_OK; }
    if (what == 13) { g_t2_mode.store(value); return MZD_OK; } // host path experiments: bit 0 one call at a time inside the submission loop, bit 1 retire by polling hipEventQuery
    if (what == 12) { g_small_nd.store(value); return MZD_OK; } // the dictionary kernels' decoding wavefronts around one table image (launches that name ONE dictionary): 0 the library's choice, 1 / 5 / 8
    if (what == 11) { g_pairs.store(value); return MZD_OK; } // driver 1's workgroups: 0 the library's choice, 1 four wavefronts a file, 2 two files a workgroup (one walking wavefront for both), 3 three wavefronts a file, five workgroups a CU
    if (what == 10) { g_resolve.store(value); return MZD_OK; } // (the small-file kernel's wavefronts per workgroup: 0 the library's choice, 1 never a helper wavefront, 2 with the 8 / 4 shape always) // (the general driver's launch behind a launch of small files alone stays: A/B)
    return MZD_E_PARAM;
}
void* mzd_host_alloc(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n ? n : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    { std::lock_guard<std::mutex> lk(g_pin_mu); g_pinned[(uintptr_t)p] = n ? n : 1; }
    return p;
}
void mzd_host_free(void* p) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(g_pin_mu); g_pinned.erase((uintptr_t)p); }
    hipHostFree(p);
}

// Frame header walk (RFC 8878 3.1.1): no entropy decoding, so it stays on the host.
// bound: instead of "unknown" for a frame without a content size, what its blocks can regenerate at most -- a raw or RLE block
// its stated size, a compressed block min(128 KiB, the frame's window)
static uint64_t content_walk(const uint8_t* src, size_t n, bool bound) {
    size_t pos = 0;
    uint64_t total = 0;
    bool unknown = false;
    while (pos < n) {
        if (n - pos < 4) return MZD_CONTENTSIZE_ERROR;
        uint32_t magic = rd32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (n - pos < 8) return MZD_CONTENTSIZE_ERROR;
            uint64_t sz = rd32(src + pos + 4);
            if (n - pos - 8 < sz) return MZD_CONTENTSIZE_ERROR;
            pos += 8 + (size_t)sz;
            continue;
        }
        if (magic != 0xFD2FB528u || n - pos < 5) return MZD_CONTENTSIZE_ERROR;
        uint32_t fhd = src[pos + 4];
        uint32_t fcsf = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
        if (fhd & 8) return MZD_CONTENTSIZE_ERROR;
        size_t hs = 5 + (single ? 0 : 1) + (did == 3 ? 4 : did) + (fcsf == 0 ? single : (1u << fcsf));
        if (n - pos < hs) return MZD_CONTENTSIZE_ERROR;
        const uint8_t* q = src + pos + hs - (fcsf == 0 ? single : (1u << fcsf));
        bool has_fcs = true;
        uint64_t fcs = 0;
        if (fcsf == 0) { if (single) fcs = *q; else has_fcs = false; }
        else if (fcsf == 1) fcs = (uint64_t)rd16(q) + 256;
        else if (fcsf == 2) fcs = rd32(q);
        else fcs = rd64(q);
        uint64_t window = fcs;
        if (!single) { const uint32_t wb = src[pos + 5]; const uint32_t wl = 10 + (wb >> 3); window = (1ull << wl) + ((1ull << wl) >> 3) * (wb & 7); }
        const uint64_t block_max = std::min<uint64_t>(window, kBlockMax);
        uint64_t by_blocks = 0;
        size_t p = pos + hs;
        for (;;) { // block chain
            if (n - p < 3) return MZD_CONTENTSIZE_ERROR;
            uint32_t bh = rd16(src + p) | ((uint32_t)src[p + 2] << 16);
            p += 3;
            uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
            if (type == 3) return MZD_CONTENTSIZE_ERROR;
            size_t adv = type == 1 ? 1 : bs;
            if (n - p < adv) return MZD_CONTENTSIZE_ERROR;
            by_blocks += type == 2 ? block_max : bs;
            p += adv;
            if (bh & 1) break;
        }
        if (fhd & 4) { if (n - p < 4) return MZD_CONTENTSIZE_ERROR; p += 4; }
        if (has_fcs) total += fcs; else if (bound) total += by_blocks; else unknown = true;
        pos = p;
    }
    return unknown ? MZD_CONTENTSIZE_UNKNOWN : total;
}
uint64_t mzd_content_size(const uint8_t* src, size_t n) { return content_walk(src, n, false); }
uint64_t mzd_content_bound(const uint8_t* src, size_t n) { return content_walk(src, n, true); }

int mzd_decode_batch(mzd_job* jobs, size_t njobs) {
    std::vector<std::shared_ptr<Device>> devs;
    { std::lock_guard<std::mutex> lk(g_mu); devs = g_dev; }
    const size_t ndev = devs.size();
    if (ndev == 0) return MZD_E_DEVICE;
    if (!jobs && njobs) return MZD_E_PARAM;
    for (size_t i = 0; i < njobs; i++) { jobs[i].status = MZD_E_DEVICE; jobs[i].out_len = 0; jobs[i].device = -1; }
    std::vector<std::vector<size_t>> shard(ndev);
    for (size_t i = 0; i < njobs; i++) shard[i % ndev].push_back(i); // file i -> GPU i mod N
    std::vector<int> rcs(ndev, MZD_OK);
    if (ndev == 1) {
        rcs[0] = run_host_jobs(*devs[0], jobs, shard[0]);
    } else {
        std::vector<std::thread> th;
        for (size_t d = 0; d < ndev; d++)
            th.emplace_back([&, d] { rcs[d] = run_host_jobs(*devs[d], jobs, shard[d]); });
        for (auto& t : th) t.join();
    }
    for (int rc : rcs) if (rc) return rc;
    // a destination that was too small: out_len says what would have sufficed (host buffers: the headers can be walked here)
    for (size_t i = 0; i < njobs; i++)
        if (jobs[i].status == MZD_E_DSTSIZE && jobs[i].src) {
            const uint64_t need = mzd_content_bound(jobs[i].src, jobs[i].src_len);
            jobs[i].out_len = need < MZD_CONTENTSIZE_ERROR && need > jobs[i].dst_cap ? (size_t)need : 0;
        }
    return MZD_OK;
}

int mzd_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len) {
    mzd_job j;
    memset(&j, 0, sizeof(j));
    j.src = src; j.src_len = n; j.dst = dst; j.dst_cap = cap;
    int rc = mzd_decode_batch(&j, 1);
    if (out_len) *out_len = j.out_len;
    return rc ? rc : j.status;
}

int mzd_decode_batch_device(int device, mzd_job* jobs, size_t njobs, void* stream) {
    auto d = get_device(device);
    if (!d) return MZD_E_DEVICE;
    if (!jobs && njobs) return MZD_E_PARAM;
    return run_device_jobs(*d, jobs, njobs, (hipStream_t)stream);
}

int mzd_batch_prepare(int device, const mzd_job* jobs, size_t njobs, mzd_batch** out) {
    auto d = get_device(device);
    if (!d) return MZD_E_DEVICE;
    if (!jobs || !out || njobs == 0 || njobs > 0x7FFFFFF0u) return MZD_E_PARAM;
    HIPCHK(hipSetDevice(d->hip_id));
    auto* b = new mzd_batch{device, nullptr, nullptr, nullptr, njobs, Plan{}, d, false, true};
    std::vector<uint32_t> lists(njobs * 2);
    if (hipMalloc(&b->d_jobs, njobs * sizeof(DevJob)) != hipSuccess || hipMalloc(&b->d_lists, (njobs * 2 + 4) * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc(&b->h_jobs, njobs * sizeof(DevJob), hipHostMallocDefault) != hipSuccess) {
        hipFree(b->d_jobs); hipFree(b->d_lists); delete b; return MZD_E_DEVICE;
    }
    for (size_t i = 0; i < njobs; i++) fill_devjob(b->h_jobs[i], jobs[i].src, jobs[i].src_len, jobs[i].dst, jobs[i].dst_cap, jobs[i].dict_id, *d);
    b->plan = make_plan(b->h_jobs, njobs, lists.data(), d->max_wg, d->cus);
    if (hipMemcpy(b->d_jobs, b->h_jobs, njobs * sizeof(DevJob), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b->d_lists, lists.data(), njobs * 2 * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(b->d_lists + njobs * 2, 0, 4 * sizeof(uint32_t)) != hipSuccess) { // (the word a launch of small files alone leaves its hand-on count in: never read uninitialised)
        hipFree(b->d_jobs); hipFree(b->d_lists); hipHostFree(b->h_jobs); delete b; return MZD_E_DEVICE;
    }
    *out = b;
    return MZD_OK;
}

int mzd_batch_launch(mzd_batch* b, void* stream) { return mzd_batch_launch_ex(b, stream, 0u); }

int mzd_batch_launch_ex(mzd_batch* b, void* stream, unsigned flags) {
    if (!b) return MZD_E_PARAM;
    Device& d = *b->dev;
    HIPCHK(hipSetDevice(d.hip_id));
    WholeGuard g(d); // (launches of one batch follow each other on `stream`; host-path launches wait for their end event)
    d.whole_used.store(true, std::memory_order_relaxed);
    const bool solo = b->plan.nsmall && b->plan.nbig == 0 && !g_keep_behind.load(std::memory_order_relaxed);
    // (untimed: the start event is left out -- one packet less between two launches of a measurement loop; the end event stays, it is
    //  what host-path launches wait for when a device-path launch may still be running on a caller's stream)
    const int rc = enqueue(d, d.whole, stream ? (hipStream_t)stream : d.whole.stream, b->d_jobs, b->plan, b->d_lists, (flags & MZD_LAUNCH_UNTIMED) ? nullptr : d.whole.ev0, d.whole.ev1, solo ? b->d_lists + 2 * b->njobs : nullptr);
    b->solo = solo && rc == MZD_OK; // (a launch that failed left no count behind: collect must not go looking for one)
    b->timed = (flags & MZD_LAUNCH_UNTIMED) == 0;
    d.job0_counter = d.whole.counter; d.job0_snap_valid = false;
    return rc;
}

int mzd_batch_collect(mzd_batch* b, mzd_job* jobs, void* stream) {
    if (!b) return MZD_E_PARAM;
    Device& d = *b->dev;
    HIPCHK(hipSetDevice(d.hip_id));
    hipStream_t s = stream ? (hipStream_t)stream : d.whole.stream;
    uint32_t handed = 0;
    d.job0_snap_valid = false;
    if (b->solo) {
        HIPCHK(hipMemcpyAsync(&handed, b->d_lists + 2 * b->njobs, sizeof handed, hipMemcpyDeviceToHost, s));
        if (d.job0_counter) HIPCHK(hipMemcpyAsync(d.job0_snap, d.job0_counter, sizeof d.job0_snap, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipMemcpyAsync(b->h_jobs, b->d_jobs, b->njobs * sizeof(DevJob), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (b->solo) { d.job0_snap[4] = handed; d.job0_snap_valid = true; } // (word 4 is the batch's own record: another batch's launch may have cleaned the block since)
    if (b->timed) HIPCHK(hipEventElapsedTime(&d.last_ms, d.whole.ev0, d.whole.ev1)); // (an untimed last launch: the figure of the last timed one stands)
    if (b->solo && handed) { // (the launch ended with the small-file kernel: the files it handed on take the general driver now)
        if (handed > b->njobs) return MZD_E_DEVICE;
        WholeGuard g(d);
        HIPCHK(hipEventRecord(d.whole.ev0, s));
        const int rc = redo_handed_on(d, d.whole, s, b->d_jobs, (uint32_t)b->njobs, b->d_lists, handed);
        if (rc) { hipStreamSynchronize(s); return rc; }
        HIPCHK(hipEventRecord(d.whole.ev1, s));
        HIPCHK(hipMemcpyAsync(b->h_jobs, b->d_jobs, b->njobs * sizeof(DevJob), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        float redo_ms = 0; // (what was handed on is part of the batch's decode: its launch counts, as in run_device_jobs)
        HIPCHK(hipEventElapsedTime(&redo_ms, d.whole.ev0, d.whole.ev1));
        d.last_ms += redo_ms;
    }
    if (jobs)
        for (size_t i = 0; i < b->njobs; i++) { jobs[i].out_len = (size_t)b->h_jobs[i].out_len; jobs[i].status = b->h_jobs[i].status; jobs[i].device = d.index; }
    return MZD_OK;
}

void mzd_batch_free(mzd_batch* b) {
    if (!b) return;
    hipSetDevice(b->dev->hip_id);
    hipFree(b->d_jobs);
    hipFree(b->d_lists);
    hipHostFree(b->h_jobs);
    delete b;
}

int mzd_load_dict(const uint8_t* dict, size_t n, uint32_t* dict_id) {
    std::vector<std::shared_ptr<Device>> devs;
    { std::lock_guard<std::mutex> lk(g_mu); devs = g_dev; }
    if (devs.empty()) return MZD_E_DEVICE;
    if (!dict || n == 0 || n > 0x7FFFFFFFu || !dict_id) return MZD_E_PARAM;
    // the same handle on every device: the lowest one free everywhere
    uint32_t slot = kMaxDicts;
    for (uint32_t k = 0; k < kMaxDicts && slot == kMaxDicts; k++) {
        bool free_all = true;
        for (auto& d : devs) { std::lock_guard<std::mutex> lk(d->dict_mu); if (d->dict_used[k]) free_all = false; }
        if (free_all) slot = k;
    }
    if (slot == kMaxDicts) return MZD_E_PARAM;
    for (auto& dp : devs) {
        Device& d = *dp;
        WholeGuard g(d);
        std::lock_guard<std::mutex> lk(d.dict_mu);
        HIPCHK(hipSetDevice(d.hip_id));
        uint8_t* buf = nullptr;
        int32_t* st = nullptr;
        HIPCHK(hipMalloc(&buf, align_up(n + 64, kAlign))); // (readable 32 bytes past the content: mzd_lds.hip fetches dictionary matches in two 16-byte pieces)
        hipFree(d.dict_bufs[slot]);
        d.dict_bufs[slot] = buf;
        HIPCHK(hipMemset(buf, 0, align_up(n + 64, kAlign)));
        HIPCHK(hipMemcpy(buf, dict, n, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&st, 64));
        HIPCHK(hipMemset(st, 0xFF, 64));
        launch_dict_kernel(buf, (uint32_t)n, d.d_dicts + slot, st, d.whole.stream);
        HIPCHK(hipGetLastError());
        int32_t status = MZD_E_DEVICE;
        HIPCHK(hipMemcpyAsync(&status, st, 4, hipMemcpyDeviceToHost, d.whole.stream));
        HIPCHK(hipStreamSynchronize(d.whole.stream));
        hipFree(st);
        if (status != MZD_OK) return status;
        d.dict_used[slot] = true;
        d.ndicts = std::max(d.ndicts, slot + 1);
    }
    *dict_id = slot + 1;
    return MZD_OK;
}

int mzd_unload_dict(uint32_t dict_id) {
    std::vector<std::shared_ptr<Device>> devs;
    { std::lock_guard<std::mutex> lk(g_mu); devs = g_dev; }
    if (devs.empty()) return MZD_E_DEVICE;
    if (dict_id < 1 || dict_id > kMaxDicts) return MZD_E_PARAM;
    int rc = MZD_E_PARAM;
    for (auto& dp : devs) {
        Device& d = *dp;
        WholeGuard g(d); // no launch in flight reads the dictionary
        std::lock_guard<std::mutex> lk(d.dict_mu);
        if (!d.dict_used[dict_id - 1]) continue;
        hipSetDevice(d.hip_id);
        d.dict_used[dict_id - 1] = false;
        hipFree(d.dict_bufs[dict_id - 1]);
        d.dict_bufs[dict_id - 1] = nullptr;
        rc = MZD_OK;
    }
    return rc;
}

int mzd_debug_set_driver(int driver) {
    if (driver < 0 || driver > 5) return MZD_E_PARAM;
    g_force_driver.store(driver, std::memory_order_relaxed);
    return MZD_OK;
}

// Diagnostic: the counter block of the launch that decoded job 0 of the most recent call (mzd_device.h: tickets, pushes,
// files finished, jobs the small-file kernel handed on, its group tickets).
int mzd_debug_counters(int device, uint32_t* out8) {
    auto dp = get_device(device);
    if (!dp || !out8) return MZD_E_PARAM;
    Device* d = dp.get();
    WholeGuard g(*d);
    HIPCHK(hipSetDevice(d->hip_id));
    if (!d->job0_counter) return MZD_E_PARAM;
    if (d->job0_snap_valid) memcpy(out8, d->job0_snap, sizeof d->job0_snap);
    else HIPCHK(hipMemcpy(out8, d->job0_counter, kCounterWords * sizeof(uint32_t), hipMemcpyDeviceToHost));
#ifdef MZD_EXP_DEVSITE
    mzd::devsite_take(out8 + 5); // (words 5, 6, 7: first, max, count)
#endif
#ifdef MZD_EXP_PLANDIAG
    for (int tu_ = 0; tu_ < 2; tu_++) { uint32_t pd[16]; if (tu_) mzd::plandiag_take_pairs(pd); else mzd::plandiag_take(pd); if (pd[10]) { fprintf(stderr, "WALKSTAT tu %d: walks %u joint runs %u (steps %u) solo-after-wait %u solo runs %u (steps %u) slave alone %u slave served %u releases %u (void %u low %u) master wait Kcyc %u slave wait Kcyc %u; asm cycles/step joint %.1f solo %.1f\n", tu_, pd[10], pd[1], pd[7], pd[2], pd[3], pd[8], pd[4], pd[11], pd[9], pd[12], pd[13], pd[5], pd[6], pd[7] ? 64.0 * pd[14] / pd[7] : 0.0, pd[8] ? 64.0 * pd[15] / pd[8] : 0.0); } }
#endif
    return MZD_OK;
}

int mzd_debug_last_block(int device, uint8_t* lit, size_t lit_cap, size_t* n_lit, uint32_t* seq4, size_t seq_cap, size_t* n_seq) {
    auto dp = get_device(device);
    if (!dp) return MZD_E_DEVICE;
    Device* d = dp.get();
    WholeGuard g(*d);
    HIPCHK(hipSetDevice(d->hip_id));
    uint32_t slot = 0;
    if (!d->job0_counter) return MZD_E_PARAM;
    HIPCHK(hipMemcpy(&slot, d->job0_counter + 1, 4, hipMemcpyDeviceToHost));
    slot &= 0xFFFu; // task << 12 | slot
    if (slot >= d->max_wg3) return MZD_E_PARAM;
    DebugSlot ds;
    HIPCHK(hipMemcpy(&ds, d->debug + slot, sizeof(ds), hipMemcpyDeviceToHost));
    if (n_lit) *n_lit = ds.n_lit;
    if (n_seq) *n_seq = ds.n_seq;
    if (lit && ds.n_lit) {
        const void* srcp = ds.lit_is_raw ? (const void*)(uintptr_t)ds.lit_raw_ptr : (const void*)(d->lit_scratch + (size_t)slot * kLitStride);
        HIPCHK(hipMemcpy(lit, srcp, std::min<size_t>(lit_cap, ds.n_lit), hipMemcpyDeviceToHost));
    }
    if (seq4 && ds.n_seq) { // the plan: {ll | ml << 16, off} a sequence; a length of 0xFFFF or more in full in the array's second half (mzd_k_execute.h: plan_store)
        const size_t ns = std::min<size_t>(seq_cap, std::min<size_t>(ds.n_seq, kSeqStride));
        std::vector<uint32_t> nar(2 * ns), wide(2 * ns);
        const uint8_t* const base = reinterpret_cast<const uint8_t*>(d->seq_scratch + (size_t)slot * kSeqStride);
        HIPCHK(hipMemcpy(nar.data(), base, ns * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(wide.data(), base + (size_t)kSeqStride * 8, ns * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ns; i++) {
            uint32_t ll = nar[2 * i] & 0xFFFFu, ml = nar[2 * i] >> 16;
            if (ll == 0xFFFFu || ml == 0xFFFFu) { ll = wide[2 * i]; ml = wide[2 * i + 1]; }
            seq4[4 * i] = ll; seq4[4 * i + 1] = ml; seq4[4 * i + 2] = nar[2 * i + 1]; seq4[4 * i + 3] = 0;
        }
    }
    return MZD_OK;
}

// Diagnostic (libmzd_diag.so, built with -DMZD_STAMPS): per-phase cycle sums of the workgroup that ran job 0.
int mzd_debug_stamps(int device, uint64_t* out8) {
    auto dp = get_device(device);
    if (!dp || !out8) return MZD_E_PARAM;
    Device* d = dp.get();
    WholeGuard g(*d);
    HIPCHK(hipSetDevice(d->hip_id));
    uint32_t slot = 0;
    if (!d->job0_counter) return MZD_E_PARAM;
    HIPCHK(hipMemcpy(&slot, d->job0_counter + 1, 4, hipMemcpyDeviceToHost));
    slot &= 0xFFFu; // task << 12 | slot
    if (slot >= d->max_wg3) return MZD_E_PARAM;
    DebugSlot ds;
    HIPCHK(hipMemcpy(&ds, d->debug + slot, sizeof(ds), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out8[i] = ds.stamp[i];
    for (int i = 0; i < 8; i++) out8[8 + i] = ds.cstamp[i];
    for (int i = 0; i < 6; i++) out8[16 + i] = ds.tfin[i];
    return MZD_OK;
}

// Diagnostic: what the small-file kernel's entropy phase left in its scratch for resident file slot `slot` (the f-th file of the group
// that workgroup b decoded last: slot = b * G + f) in the most recent call on device pointers: its literals (lit_n bytes) and its
// sequences, 8 bytes each -- literal length | match length << 14 | offset VALUE << 32 (before repeat-offset resolution, A.5).
int mzd_debug_small_scratch(int device, uint32_t slot, uint8_t* lit, size_t lit_n, uint64_t* seq, size_t seq_n) {
    auto dp = get_device(device);
    if (!dp) return MZD_E_PARAM;
    WholeGuard g(*dp);
    HIPCHK(hipSetDevice(dp->hip_id));
    if (!dp->last_lds_lit_stride) return MZD_E_PARAM;
    const size_t per = (size_t)dp->last_lds_lit_stride + 12u * (size_t)dp->last_lds_seq_cap;
    if ((slot + 1) * per > dp->small_lit_total || lit_n > dp->last_lds_lit_stride || seq_n > dp->last_lds_seq_cap) return MZD_E_PARAM;
    if (lit && lit_n) HIPCHK(hipMemcpy(lit, dp->small_lit + slot * per, lit_n, hipMemcpyDeviceToHost));
    if (seq && seq_n) { // the kernel's 4-byte records, widened (the full records of the sequences that do not fit: the second array)
        std::vector<uint32_t> r4(seq_n);
        std::vector<uint64_t> r8(seq_n);
        const uint8_t* const base = dp->small_lit + slot * per + dp->last_lds_lit_stride;
        HIPCHK(hipMemcpy(r4.data(), base, 4 * seq_n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(r8.data(), base + 4u * (size_t)dp->last_lds_seq_cap, 8 * seq_n, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < seq_n; i++)
            seq[i] = (r4[i] & 127u) == 127u ? r8[i] : ((uint64_t)((r4[i] & 127u) | ((((r4[i] >> 7) & 63u) + 3u) << 14)) | ((uint64_t)(r4[i] >> 13) << 32));
    }
    return MZD_OK;
}

// Diagnostic (a build with -DMZD_SMALL_STAMPS): the 8 phase stamps of the small-file kernel's workgroup 0.
int mzd_debug_small_wg_stamps(int device, uint64_t* out, int n) { // (diagnostic builds: per-workgroup entry / exit clocks of the small-file kernel, 16 values each)
    auto dp = get_device(device);
    if (!dp || !out || n < 0 || n > 3072) return MZD_E_PARAM;
    HIPCHK(hipSetDevice(dp->hip_id));
    HIPCHK(hipMemcpy(out, (uint64_t*)dp->debug + 2048, (size_t)n * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MZD_OK;
}
int mzd_debug_small_stamps(int device, uint64_t* out8) { // (64 values; 32..47: files that left the fast path, by reason)
    auto dp = get_device(device);
    if (!dp || !out8) return MZD_E_PARAM;
    WholeGuard g(*dp);
    HIPCHK(hipSetDevice(dp->hip_id));
    HIPCHK(hipMemcpy(out8, dp->debug, 1032 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset((uint8_t*)dp->debug + 32 * sizeof(uint64_t), 0, 1000 * sizeof(uint64_t)));
    return MZD_OK;
}

// Diagnostic: tfin[12] of every workgroup slot (n_slots * 12 values); returns the slot count.
int mzd_debug_tfin_all(int device, uint64_t* out, int max_slots) {
    auto dp = get_device(device);
    if (!dp || !out) return MZD_E_PARAM;
    Device* d = dp.get();
    WholeGuard g(*d);
    HIPCHK(hipSetDevice(d->hip_id));
    int n = std::min<int>((int)d->max_wg3, max_slots);
    std::vector<DebugSlot> all((size_t)n);
    HIPCHK(hipMemcpy(all.data(), d->debug, sizeof(DebugSlot) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) for (int k = 0; k < 12; k++) out[(size_t)i * 12 + k] = all[(size_t)i].tfin[k];
    return n;
}

int mzd_last_kernel_ms(int device, float* ms) {
    auto d = get_device(device);
    if (!d || !ms) return MZD_E_PARAM;
    *ms = d->last_ms;
    return MZD_OK;
}

const char* mzd_last_kernel_name(int device) { // (a copy of the calling thread's own: a launch on another thread, or mzd_shutdown, cannot change it under the reader)
    static thread_local char name[96];
    auto d = get_device(device);
    if (!d) return "";
    std::lock_guard<std::mutex> lk(d->names_mu);
    memcpy(name, d->last_kernels, sizeof name);
    return name;
}

const char* mzd_strerror(int code) {
    switch (code) {
    case MZD_OK: return "ok";
    case MZD_E_CORRUPT: return "corrupt input";
    case MZD_E_TRUNCATED: return "truncated input";
    case MZD_E_CHECKSUM: return "content checksum mismatch";
    case MZD_E_DSTSIZE: return "destination too small";
    case MZD_E_UNSUPPORTED: return "unsupported frame parameter";
    case MZD_E_DEVICE: return "no usable GPU / HIP error / not initialised";
    case MZD_E_BADMAGIC: return "unknown frame magic";
    case MZD_E_DICT: return "dictionary missing, wrong or corrupt";
    case MZD_E_PARAM: return "bad argument";
    default: return "unknown error";
    }
}

const char* mzd_version(void) { return "mzd 0.2 (gfx950)"; }

} // extern "C"

// ---------------------------------------------------------------------------------------
// Host mirror of OpenedFiles (reference src/file.rs) + open/read/release wrappers.
// ---------------------------------------------------------------------------------------
// Lazy / seekable files: what the host's header walk finds in a file whose frames all carry a content size
struct LazyBlock { size_t hdr; uint32_t type, size; bool last; };   // a block header at input offset `hdr`
struct LazyFrame {
    size_t in_off, in_len;        // the frame's bytes in the file
    size_t hdr_len;               // its header
    uint64_t out_off, out_len;    // where its content lies in the decoded file
    uint32_t window_log;          // a Window_Descriptor exponent that covers the frame's window (prefix decodes)
    std::vector<LazyBlock> blocks;
    uint64_t valid = 0;           // decoded bytes of this frame present in `bytes` (a prefix)
};
struct DecodedFile { // plays the role of the anonymous tempfile (reference src/main.rs:462)
    std::vector<uint8_t> bytes;
    bool lazy = false;
    std::vector<uint8_t> zst;     // lazy: the compressed file (kept: reads decode from it)
    std::vector<LazyFrame> frames;
};

struct FileHandler { // reference src/file.rs:20-28
    int32_t flags;
    bool needs_sync;
    std::shared_ptr<DecodedFile> file; // try_clone() of the tempfile == another reference to the same bytes
    bool has_refs;
    uint64_t inode;
};

struct mzd_fs {
    std::map<uint64_t, std::set<uint64_t>> inode_map; // mount_point_inode_mapping, src/file.rs:12
    std::map<uint64_t, FileHandler> handlers;         // src/file.rs:13
    uint64_t decodes = 0;
    uint64_t decoded_bytes = 0;   // bytes the GPU produced for this table so far
    std::mutex mu;

    bool new_fh(uint64_t* out) { // smallest free handle number, src/file.rs:38-45
        uint64_t i = 0;
        for (auto& kv : handlers) { if (kv.first != i) break; i++; }
        *out = i;
        return true;
    }
};

extern "C" {

mzd_fs* mzd_fs_new(void) { return new mzd_fs(); }
void mzd_fs_free(mzd_fs* fs) { delete fs; }

// What n bytes of frames can regenerate at most: every block regenerates <= 128 KiB and costs >= 4 bytes (header + an RLE byte).
// Content sizes come from untrusted headers and size allocations: one above this bound is refused before anything is allocated.
static uint64_t max_regenerated(size_t n) { return ((uint64_t)n / 4 + 1) * kBlockMax; }

int64_t mzd_fs_open(mzd_fs* fs, uint64_t ino, int32_t flags, const uint8_t* zst, size_t zst_len, uint64_t* real_size) {
    if (!fs) return -EINVAL;
    std::lock_guard<std::mutex> lk(fs->mu);
    // "Already opened by some other process": duplicate, no decode (src/main.rs:453-459, src/file.rs:67-102)
    auto it = fs->inode_map.find(ino);
    if (it != fs->inode_map.end() && !it->second.empty()) {
        const FileHandler& h = fs->handlers.at(*it->second.begin());
        uint64_t fh;
        fs->new_fh(&fh);
        fs->handlers[fh] = FileHandler{flags, false, h.file, true, ino};
        it->second.insert(fh);
        if (real_size) *real_size = h.file->bytes.size();
        return (int64_t)fh;
    }
    // copy_decode(source, target).map_err(|_| EFAULT)   (src/main.rs:463-467)
    auto file = std::make_shared<DecodedFile>();
    uint64_t want = mzd_content_size(zst, zst_len);
    if (want == MZD_CONTENTSIZE_ERROR) return -EFAULT;
    if (want != MZD_CONTENTSIZE_UNKNOWN && want > max_regenerated(zst_len)) return -EFAULT; // (a header that promises more than its blocks can hold)
    // frames without a content size: a guess first (eight times the input, at least 1 MiB); if that is too small, the decode
    // reports what the block headers allow at most (mzd_content_bound) and the second attempt cannot fail for size
    size_t cap = want == MZD_CONTENTSIZE_UNKNOWN ? (size_t)std::min<uint64_t>(std::max<size_t>(zst_len * 8, 1 << 20), mzd_content_bound(zst, zst_len)) : (size_t)want;
    for (int attempt = 0; attempt < 2; attempt++) {
        try { file->bytes.resize(cap); } catch (const std::exception&) { return -ENOMEM; }
        size_t out_len = 0;
        int rc = mzd_decode(zst, zst_len, file->bytes.data(), cap, &out_len);
        fs->decodes++;
        if (rc == MZD_OK) { file->bytes.resize(out_len); fs->decoded_bytes += out_len; break; }
        if (rc == MZD_E_DSTSIZE && want == MZD_CONTENTSIZE_UNKNOWN && attempt == 0 && out_len > cap) { cap = out_len; continue; }
        return -EFAULT;
    }
    uint64_t fh;
    if (!fs->new_fh(&fh)) return -EBUSY; // src/main.rs:490
    fs->handlers[fh] = FileHandler{flags, false, file, true, ino};
    fs->inode_map[ino].insert(fh);
    if (real_size) *real_size = file->bytes.size(); // -> user.real_size xattr (src/main.rs:473-482)
    return (int64_t)fh;
}

// ---- N4: lazy open.  Nothing is decoded at open; the host walks frame and block headers (no entropy decoding) and every
// read decodes what it needs: the frames that cover the range -- frames are independent -- and, inside a frame, the blocks up
// to the range's end (a block needs its predecessors' window, tables and repeat offsets, so a frame is decoded from its
// start: as a shorter frame made of its first blocks).  What has been decoded stays (reference read_wrapper only ever
// slices: src/main.rs:495-513).  Files with a frame that carries no content size are decoded eagerly, as by mzd_fs_open.
static bool lazy_index(const uint8_t* src, size_t n, std::vector<LazyFrame>& frames, uint64_t* total) {
    size_t pos = 0;
    uint64_t out = 0;
    while (pos < n) {
        if (n - pos < 4) return false;
        const uint32_t magic = rd32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (n - pos < 8) return false;
            const uint64_t sz = rd32(src + pos + 4);
            if (n - pos - 8 < sz) return false;
            pos += 8 + (size_t)sz;
            continue;
        }
        if (magic != 0xFD2FB528u || n - pos < 5) return false;
        const uint32_t fhd = src[pos + 4];
        const uint32_t fcsf = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
        if ((fhd & 8) || did) return false; // (dictionary frames: decoded eagerly, with the caller's dictionary)
        const size_t fcs_sz = fcsf == 0 ? single : (1u << fcsf);
        const size_t hs = 5 + (single ? 0 : 1) + fcs_sz;
        if (n - pos < hs || fcs_sz == 0) return false; // no content size: not seekable
        const uint8_t* q = src + pos + hs - fcs_sz;
        uint64_t fcs = fcsf == 0 ? *q : (fcsf == 1 ? (uint64_t)rd16(q) + 256 : (fcsf == 2 ? rd32(q) : rd64(q)));
        uint64_t window = fcs;
        if (!single) { const uint32_t b = src[pos + 5]; const uint32_t wl = 10 + (b >> 3); window = (1ull << wl) + ((1ull << wl) >> 3) * (b & 7); }
        if (window > (1ull << 27)) return false;
        LazyFrame f;
        f.in_off = pos; f.hdr_len = hs; f.out_off = out; f.out_len = fcs;
        f.window_log = 10;
        while ((1ull << f.window_log) < window) f.window_log++;
        size_t p = pos + hs;
        for (;;) {
            if (n - p < 3) return false;
            const uint32_t bh = rd16(src + p) | ((uint32_t)src[p + 2] << 16);
            LazyBlock b{p, (bh >> 1) & 3, bh >> 3, (bh & 1) != 0};
            if (b.type == 3) return false;
            const size_t adv = b.type == 1 ? 1 : b.size;
            if (n - p - 3 < adv) return false;
            f.blocks.push_back(b);
            p += 3 + adv;
            if (b.last) break;
        }
        if (fhd & 4) { if (n - p < 4) return false; p += 4; }
        f.in_len = p - pos;
        out += fcs;
        pos = p;
        frames.push_back(std::move(f));
    }
    *total = out;
    return true;
}

int64_t mzd_fs_open_lazy(mzd_fs* fs, uint64_t ino, int32_t flags, const uint8_t* zst, size_t zst_len, uint64_t* real_size) {
    if (!fs) return -EINVAL;
    {
        std::lock_guard<std::mutex> lk(fs->mu);
        auto it = fs->inode_map.find(ino);
        if (it != fs->inode_map.end() && !it->second.empty()) { // another handle of the inode: share its file (src/file.rs:67-102)
            const FileHandler& h = fs->handlers.at(*it->second.begin());
            uint64_t fh;
            fs->new_fh(&fh);
            fs->handlers[fh] = FileHandler{flags, false, h.file, true, ino};
            it->second.insert(fh);
            if (real_size) *real_size = h.file->bytes.size();
            return (int64_t)fh;
        }
    }
    auto file = std::make_shared<DecodedFile>();
    uint64_t total = 0;
    if (!lazy_index(zst, zst_len, file->frames, &total) || total > (1ull << 40)) return mzd_fs_open(fs, ino, flags, zst, zst_len, real_size);
    if (total > max_regenerated(zst_len)) return -EFAULT; // (headers that promise more than their blocks can hold)
    if (mzd_device_count() == 0) return -EFAULT; // (no GPU: the first read could not decode either; fail at open like the eager path)
    file->lazy = true;
    try {
        file->zst.assign(zst, zst + zst_len);
        file->bytes.resize((size_t)total);
    } catch (const std::exception&) { return -ENOMEM; }
    std::lock_guard<std::mutex> lk(fs->mu);
    uint64_t fh;
    if (!fs->new_fh(&fh)) return -EBUSY;
    fs->handlers[fh] = FileHandler{flags, false, file, true, ino};
    fs->inode_map[ino].insert(fh);
    if (real_size) *real_size = total; // every frame states its content size: this is what user.real_size gets (src/main.rs:473-482)
    return (int64_t)fh;
}

// decode what [lo, hi) of a lazy file needs; false: the file is corrupt (the reference would have failed at open with EFAULT)
// The first `nblocks` blocks of frame `fr` of the file `zst`, as a frame of their own: a header without content size / checksum
// (window descriptor only), the blocks' bytes as they are, the last kept block marked Last_Block.
static void lazy_synth(const std::vector<uint8_t>& zst, const LazyFrame& fr, size_t nblocks, std::vector<uint8_t>& synth) {
    const LazyBlock& lb = fr.blocks[nblocks - 1];
    const size_t body_end = lb.hdr + 3 + (lb.type == 1 ? 1 : lb.size);
    synth.reserve(6 + (body_end - (fr.in_off + fr.hdr_len)) + MZD_SRC_PADDING);
    const uint8_t head[6] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, (uint8_t)((fr.window_log - 10) << 3)};
    synth.assign(head, head + 6);
    synth.insert(synth.end(), zst.begin() + (ptrdiff_t)(fr.in_off + fr.hdr_len), zst.begin() + (ptrdiff_t)body_end);
    synth[6 + (lb.hdr - (fr.in_off + fr.hdr_len))] |= 1; // Last_Block
}

static bool lazy_fill(mzd_fs* fs, DecodedFile& f, uint64_t lo, uint64_t hi) {
    struct Want { LazyFrame* fr; uint64_t upto; size_t nblocks; std::vector<uint8_t> synth; };
    for (int round = 0; round < 40; round++) {
        std::vector<Want> want;
        for (auto& fr : f.frames) {
            if (fr.out_off + fr.out_len <= lo || fr.out_off >= hi) continue;
            const uint64_t need = std::min<uint64_t>(hi, fr.out_off + fr.out_len) - fr.out_off; // bytes of the frame wanted, from its start
            if (fr.valid >= need) continue;
            // blocks to decode: a block regenerates at most 128 KiB; if that guess falls short the next round takes more
            size_t nb = std::min<size_t>(fr.blocks.size(), (size_t)((need + kBlockMax - 1) / kBlockMax) + (size_t)round * (1u + (size_t)round));
            if (nb == 0) nb = 1;
            if (round >= 6) nb = fr.blocks.size(); // (a frame of many tiny blocks: stop guessing)
            want.push_back(Want{&fr, need, nb, {}});
        }
        if (want.empty()) return true;
        std::vector<mzd_job> jobs(want.size());
        for (size_t k = 0; k < want.size(); k++) {
            Want& w = want[k];
            LazyFrame& fr = *w.fr;
            mzd_job& j = jobs[k];
            memset(&j, 0, sizeof(j));
            j.dst = f.bytes.data() + fr.out_off;
            if (w.nblocks >= fr.blocks.size()) { // the whole frame, as it is (content size and checksum verified)
                j.src = f.zst.data() + fr.in_off; j.src_len = fr.in_len; j.dst_cap = (size_t)fr.out_len;
            } else { // its first blocks as a frame of their own: a header without content size / checksum, the last kept block marked last
                lazy_synth(f.zst, fr, w.nblocks, w.synth);
                j.src = w.synth.data(); j.src_len = w.synth.size();
                j.dst_cap = (size_t)std::min<uint64_t>(fr.out_len, (uint64_t)w.nblocks * kBlockMax);
            }
        }
        if (mzd_decode_batch(jobs.data(), jobs.size()) != MZD_OK) return false;
        fs->decodes += jobs.size();
        for (size_t k = 0; k < want.size(); k++) {
            if (jobs[k].status != MZD_OK) return false;
            LazyFrame& fr = *want[k].fr;
            fs->decoded_bytes += jobs[k].out_len;
            fr.valid = std::max<uint64_t>(fr.valid, jobs[k].out_len);
            if (want[k].nblocks >= fr.blocks.size() && jobs[k].out_len != fr.out_len) return false;
        }
    }
    return false;
}

// Diagnostic / host-side test hook (no GPU needed): the lazy open's index of `zst` and, for frame `frame`, the synthetic frame
// made of its first `nblocks` blocks.  Returns the number of frames indexed (0: the file is not seekable -- it would be opened
// eagerly), < 0 on bad arguments; *total = the content size the index promises; the synthetic frame goes to synth (cap bytes;
// *synth_len = its length, also when it does not fit); *nblocks_of_frame = the blocks frame `frame` has.
int mzd_debug_lazy_plan(const uint8_t* zst, size_t n, uint32_t frame, uint32_t nblocks, uint8_t* synth, size_t cap, size_t* synth_len, uint64_t* total, uint32_t* nblocks_of_frame) {
    if (!zst && n) return MZD_E_PARAM;
    std::vector<LazyFrame> frames;
    uint64_t tot = 0;
    if (!lazy_index(zst, n, frames, &tot)) return 0;
    if (total) *total = tot;
    if (frame < frames.size()) {
        const LazyFrame& fr = frames[frame];
        if (nblocks_of_frame) *nblocks_of_frame = (uint32_t)fr.blocks.size();
        if (nblocks >= 1 && nblocks <= fr.blocks.size()) {
            std::vector<uint8_t> z(zst, zst + n), out;
            lazy_synth(z, fr, nblocks, out);
            if (synth_len) *synth_len = out.size();
            if (synth && out.size() <= cap) memcpy(synth, out.data(), out.size());
        }
    }
    return (int)std::min<size_t>(frames.size(), 0x7FFFFFFF);
}

int64_t mzd_fs_read(mzd_fs* fs, uint64_t fh, int64_t offset, uint32_t size, uint8_t* out) {
    if (!fs) return -EINVAL;
    std::lock_guard<std::mutex> lk(fs->mu);
    auto it = fs->handlers.find(fh);
    if (it == fs->handlers.end()) return -ENOENT; // src/main.rs:505
    DecodedFile& f = *it->second.file;
    const std::vector<uint8_t>& b = f.bytes;
    if (offset < 0) return -EINVAL;
    if ((uint64_t)offset >= b.size()) return 0; // read_at past EOF -> 0 bytes, then truncate (src/main.rs:506-511)
    size_t nread = std::min<size_t>(size, b.size() - (size_t)offset);
    if (f.lazy && nread && !lazy_fill(fs, f, (uint64_t)offset, (uint64_t)offset + nread)) return -EFAULT; // (what the reference reports at open)
    if (nread && out) memcpy(out, b.data() + offset, nread);
    return (int64_t)nread;
}

int mzd_fs_release(mzd_fs* fs, uint64_t fh) {
    if (!fs) return -EINVAL;
    std::lock_guard<std::mutex> lk(fs->mu);
    auto it = fs->handlers.find(fh); // OpenedFiles::close, src/file.rs:104-117
    if (it == fs->handlers.end()) return -EBADF;
    if (it->second.has_refs) {
        auto m = fs->inode_map.find(it->second.inode);
        if (m != fs->inode_map.end()) { m->second.erase(fh); if (m->second.empty()) fs->inode_map.erase(m); }
    }
    fs->handlers.erase(it);
    return 0;
}

uint64_t mzd_fs_decode_count(const mzd_fs* fs) { return fs ? fs->decodes : 0; }
uint64_t mzd_fs_decoded_bytes(const mzd_fs* fs) { return fs ? fs->decoded_bytes : 0; }

} // extern "C"
