// mzd_host.cpp -- host runtime behind include/mzd.h: device/scratch management, job tables,
// PCIe staging for the host-pointer entry points, round-robin sharding of files over GPUs,
// and the C++ mirror of the reference's file-handle table (reference src/file.rs:10-135) with
// open/read/release (reference src/main.rs:451-513, 595-599).
//
// All decoding happens in mzd_kernels.hip; nothing here decodes (there is no CPU fallback:
// without a usable GPU every decode entry point returns MZD_E_DEVICE).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstring>
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
void* decode_kernel_ptr(int tasks);
}

namespace {

using namespace mzd;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { last_hip_error = e_; return MZD_E_DEVICE; } } while (0)
thread_local hipError_t last_hip_error = hipSuccess;

constexpr size_t kAlign = 256;
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Device {
    int hip_id = 0;
    hipStream_t stream = nullptr;
    uint32_t max_wg = 0;
    uint8_t* lit_scratch = nullptr;
    uint4* seq_scratch = nullptr;
    uint4* walk_scratch = nullptr;
    DebugSlot* debug = nullptr;
    uint32_t* counter = nullptr; // [0] tickets, [1] task << 12 | slot of the last compressed block of job 0, [2] pushes, [3] files finished
    FileState* fstate = nullptr;  // per file of a launch: what a block task hands to its successor
    TableArea* tables = nullptr;
    ContRecord* ring = nullptr;   // pushed tasks
    size_t task_cap = 0;          // files the three arrays above are sized for
    uint32_t epoch = 0;
    DevJob* d_jobs = nullptr;
    size_t d_jobs_cap = 0;
    DevJob* h_jobs = nullptr; // pinned
    uint8_t *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    size_t in_cap = 0, out_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    DevDict* d_dicts = nullptr;
    uint32_t ndicts = 0;
    std::vector<void*> dict_bufs;
    std::mutex mu;
};

constexpr uint32_t kMaxDicts = 64;
std::mutex g_mu;
std::vector<std::unique_ptr<Device>> g_dev;

void free_device(Device& d) {
    hipSetDevice(d.hip_id);
    if (d.stream) hipStreamSynchronize(d.stream);
    hipFree(d.lit_scratch); hipFree(d.seq_scratch); hipFree(d.walk_scratch); hipFree(d.debug); hipFree(d.counter); hipFree(d.d_jobs); hipFree(d.fstate); hipFree(d.tables); hipFree(d.ring);
    hipFree(d.d_in); hipFree(d.d_out); hipFree(d.d_dicts);
    for (void* p : d.dict_bufs) hipFree(p);
    if (d.h_jobs) hipHostFree(d.h_jobs);
    if (d.h_in) hipHostFree(d.h_in);
    if (d.h_out) hipHostFree(d.h_out);
    if (d.ev0) hipEventDestroy(d.ev0);
    if (d.ev1) hipEventDestroy(d.ev1);
    if (d.stream) hipStreamDestroy(d.stream);
}

int init_device(Device& d, int hip_id) {
    d.hip_id = hip_id;
    HIPCHK(hipSetDevice(hip_id));
    HIPCHK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    int cus = 0, per_cu = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, hip_id));
    int per_cu2 = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)decode_kernel_ptr(0), kWG, 0));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, (const void*)decode_kernel_ptr(1), kWG, 0));
    per_cu = std::min(per_cu, per_cu2); // the block-task driver needs its whole grid resident
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    d.max_wg = (uint32_t)(cus * per_cu);
    HIPCHK(hipMalloc(&d.lit_scratch, (size_t)d.max_wg * kLitStride));
    HIPCHK(hipMalloc(&d.seq_scratch, (size_t)d.max_wg * kSeqStride * sizeof(uint4)));
    HIPCHK(hipMalloc(&d.walk_scratch, (size_t)d.max_wg * kSeqStride * sizeof(uint4)));
    HIPCHK(hipMalloc(&d.debug, (size_t)d.max_wg * sizeof(DebugSlot)));
    HIPCHK(hipMemset(d.debug, 0, (size_t)d.max_wg * sizeof(DebugSlot)));
    HIPCHK(hipMalloc(&d.counter, 64));
    HIPCHK(hipMemset(d.counter, 0, 64));
    HIPCHK(hipMalloc(&d.d_dicts, sizeof(DevDict) * kMaxDicts));
    HIPCHK(hipEventCreate(&d.ev0));
    HIPCHK(hipEventCreate(&d.ev1));
    return MZD_OK;
}

int ensure_jobs(Device& d, size_t n) {
    if (n <= d.d_jobs_cap) return MZD_OK;
    size_t cap = std::max<size_t>(n, 1024);
    hipFree(d.d_jobs);
    if (d.h_jobs) hipHostFree(d.h_jobs);
    d.d_jobs = nullptr; d.h_jobs = nullptr; d.d_jobs_cap = 0;
    HIPCHK(hipMalloc(&d.d_jobs, cap * sizeof(DevJob)));
    HIPCHK(hipHostMalloc(&d.h_jobs, cap * sizeof(DevJob), hipHostMallocDefault));
    d.d_jobs_cap = cap;
    return MZD_OK;
}

// per-file task state for launches of up to n files (grow-only)
int ensure_task_state(Device& d, size_t n) {
    if (n <= d.task_cap) return MZD_OK;
    size_t cap = std::max<size_t>(n + n / 4, 1024);
    hipFree(d.fstate); hipFree(d.tables); hipFree(d.ring);
    d.fstate = nullptr; d.tables = nullptr; d.ring = nullptr; d.task_cap = 0;
    HIPCHK(hipMalloc(&d.fstate, cap * sizeof(FileState)));
    HIPCHK(hipMalloc(&d.tables, cap * sizeof(TableArea)));
    HIPCHK(hipMalloc(&d.ring, (cap + 1) * sizeof(ContRecord)));
    HIPCHK(hipMemset(d.ring, 0, (cap + 1) * sizeof(ContRecord)));
    d.task_cap = cap;
    return MZD_OK;
}

KernelArgs make_args(Device& d, DevJob* jobs, uint32_t njobs) {
    KernelArgs a;
    a.jobs = jobs; a.njobs = njobs; a.counter = d.counter;
    a.lit_scratch = d.lit_scratch; a.seq_scratch = d.seq_scratch; a.walk_scratch = d.walk_scratch;
    a.dicts = d.d_dicts; a.ndicts = d.ndicts; a.debug = d.debug; a.job_slot0 = d.counter + 1;
    a.fstate = d.fstate; a.tables = d.tables; a.ring = d.ring; a.ring_cap = (uint32_t)d.task_cap + 1; a.epoch = d.epoch;
    return a;
}

// enqueue: reset queue head, time the kernel with events on the launch stream
// Workgroups worth launching: one per file, plus the block tasks big files will fork (a compressed 128 KiB block is
// rarely below 2 KiB), capped by what is resident at once.  Idle workgroups leave as soon as every file is finished.
// Returns 0 when no file of the launch can have more than one block (every capacity <= 128 KiB): then driver 1 runs,
// one workgroup per file.
uint32_t grid_for(const Device& d, const DevJob* h_jobs, size_t njobs) {
    bool multi = false;
    for (size_t i = 0; i < njobs && !multi; i++) multi = h_jobs[i].dst_cap > kBlockMax;
    if (!multi) return 0;
    uint64_t tasks = 0;
    for (size_t i = 0; i < njobs && tasks < d.max_wg; i++) tasks += 1 + h_jobs[i].src_len / 2048;
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(tasks, 1), d.max_wg);
}

int enqueue(Device& d, DevJob* d_jobs, uint32_t njobs, uint32_t grid, hipStream_t s) {
    bool use_tasks = grid != 0;
    if (const char* e = getenv("MZD_DRIVER")) use_tasks = atoi(e) == 2; // diagnostics: force driver 1 / 2
    if (use_tasks) {
        int trc = ensure_task_state(d, njobs);
        if (trc) return trc;
        d.epoch++; // ring records of earlier launches never match
        grid = std::max<uint32_t>(grid, std::min<uint32_t>(njobs, d.max_wg));
        HIPCHK(hipMemsetAsync(d.counter, 0, 16, s));
        HIPCHK(hipMemsetAsync(d.fstate, 0, (size_t)njobs * sizeof(FileState), s));
    } else {
        grid = std::min<uint32_t>(njobs, d.max_wg);
        HIPCHK(hipMemsetAsync(d.counter, 0, 16, s));
    }
    HIPCHK(hipEventRecord(d.ev0, s));
    KernelArgs ka = make_args(d, d_jobs, njobs);
    ka.use_tasks = use_tasks ? 1u : 0u;
    launch_decode(ka, grid, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d.ev1, s));
    return MZD_OK;
}

// jobs carry DEVICE pointers.  Caller holds d.mu.
int run_device_jobs(Device& d, mzd_job* jobs, size_t njobs, hipStream_t s) {
    HIPCHK(hipSetDevice(d.hip_id));
    if (njobs == 0) return MZD_OK;
    if (njobs > 0xFFFFFFF0u) return MZD_E_PARAM;
    int rc = ensure_jobs(d, njobs);
    if (rc) return rc;
    for (size_t i = 0; i < njobs; i++) {
        DevJob& j = d.h_jobs[i];
        j.src = jobs[i].src; j.src_len = jobs[i].src_len; j.dst = jobs[i].dst; j.dst_cap = jobs[i].dst_cap;
        j.out_len = 0; j.status = MZD_E_DEVICE; j.dict = jobs[i].dict_id;
    }
    HIPCHK(hipMemcpyAsync(d.d_jobs, d.h_jobs, njobs * sizeof(DevJob), hipMemcpyHostToDevice, s));
    rc = enqueue(d, d.d_jobs, (uint32_t)njobs, grid_for(d, d.h_jobs, njobs), s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d.h_jobs, d.d_jobs, njobs * sizeof(DevJob), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipEventElapsedTime(&d.last_ms, d.ev0, d.ev1));
    for (size_t i = 0; i < njobs; i++) { jobs[i].out_len = (size_t)d.h_jobs[i].out_len; jobs[i].status = d.h_jobs[i].status; }
    return MZD_OK;
}

int ensure_staging(Device& d, size_t in_bytes, size_t out_bytes) {
    if (in_bytes > d.in_cap) {
        size_t cap = align_up(in_bytes + in_bytes / 4, 1 << 20);
        hipFree(d.d_in); if (d.h_in) hipHostFree(d.h_in);
        d.d_in = nullptr; d.h_in = nullptr; d.in_cap = 0;
        HIPCHK(hipMalloc(&d.d_in, cap));
        HIPCHK(hipHostMalloc(&d.h_in, cap, hipHostMallocDefault));
        d.in_cap = cap;
    }
    if (out_bytes > d.out_cap) {
        size_t cap = align_up(out_bytes + out_bytes / 4, 1 << 20);
        hipFree(d.d_out); if (d.h_out) hipHostFree(d.h_out);
        d.d_out = nullptr; d.h_out = nullptr; d.out_cap = 0;
        HIPCHK(hipMalloc(&d.d_out, cap));
        HIPCHK(hipHostMalloc(&d.h_out, cap, hipHostMallocDefault));
        d.out_cap = cap;
    }
    return MZD_OK;
}

// The staging copies (user buffers <-> pinned memory) of a big batch are memory-bound host work: split over a few
// threads (one thread moves ~10 GB/s; the PCIe link ~50).  fn(k) handles job k of [0, n).
template <class F>
void parallel_jobs(size_t n, size_t total_bytes, F fn) {
    unsigned want = total_bytes < (8u << 20) ? 1u : std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
    if (want <= 1 || n < 2 * want) { for (size_t k = 0; k < n; k++) fn(k); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < want; t++)
        th.emplace_back([=]() { for (size_t k = n * t / want; k < n * (t + 1) / want; k++) fn(k); });
    for (auto& x : th) x.join();
}

// HOST-pointer jobs `idx` on one device: pinned staging -> H2D -> kernel -> D2H -> user buffers.
int run_host_jobs(Device& d, mzd_job* jobs, const std::vector<size_t>& idx) {
    std::lock_guard<std::mutex> lk(d.mu);
    HIPCHK(hipSetDevice(d.hip_id));
    if (idx.empty()) return MZD_OK;
    size_t in_total = 0, out_total = 0;
    std::vector<size_t> in_off(idx.size()), out_off(idx.size());
    for (size_t k = 0; k < idx.size(); k++) {
        const mzd_job& j = jobs[idx[k]];
        in_off[k] = in_total; in_total += align_up(j.src_len + MZD_SRC_PADDING, kAlign);
        out_off[k] = out_total; out_total += align_up(j.dst_cap + 16, kAlign);
    }
    int rc = ensure_staging(d, in_total, out_total);
    if (rc) return rc;
    rc = ensure_jobs(d, idx.size());
    if (rc) return rc;
    parallel_jobs(idx.size(), in_total, [&](size_t k) {
        const mzd_job& j = jobs[idx[k]];
        if (j.src_len) memcpy(d.h_in + in_off[k], j.src, j.src_len);
        memset(d.h_in + in_off[k] + j.src_len, 0, MZD_SRC_PADDING);
        DevJob& dj = d.h_jobs[k];
        dj.src = d.d_in + in_off[k]; dj.src_len = j.src_len; dj.dst = d.d_out + out_off[k]; dj.dst_cap = j.dst_cap;
        dj.out_len = 0; dj.status = MZD_E_DEVICE; dj.dict = j.dict_id;
    });
    hipStream_t s = d.stream;
    HIPCHK(hipMemcpyAsync(d.d_in, d.h_in, in_total, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d.d_jobs, d.h_jobs, idx.size() * sizeof(DevJob), hipMemcpyHostToDevice, s));
    rc = enqueue(d, d.d_jobs, (uint32_t)idx.size(), grid_for(d, d.h_jobs, idx.size()), s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(d.h_jobs, d.d_jobs, idx.size() * sizeof(DevJob), hipMemcpyDeviceToHost, s));
    // The output comes back in slices of jobs; while slice i+1 crosses the link, slice i is copied out to the callers' buffers.
    const size_t kSlice = 24u << 20;
    std::vector<size_t> cut{0};
    for (size_t k = 0, acc = 0; k < idx.size(); k++) {
        acc += (k + 1 < idx.size() ? out_off[k + 1] : out_total) - out_off[k];
        if (acc >= kSlice || k + 1 == idx.size()) { cut.push_back(k + 1); acc = 0; }
    }
    std::vector<hipEvent_t> evs(cut.size() - 1, nullptr);
    for (size_t i = 0; i + 1 < cut.size(); i++) {
        const size_t b0 = out_off[cut[i]], b1 = cut[i + 1] < idx.size() ? out_off[cut[i + 1]] : out_total;
        hipError_t e = hipMemcpyAsync(d.h_out + b0, d.d_out + b0, b1 - b0, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&evs[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(evs[i], s);
        if (e != hipSuccess) { hipStreamSynchronize(s); for (auto ev : evs) if (ev) hipEventDestroy(ev); return MZD_E_DEVICE; }
    }
    int result = MZD_OK;
    for (size_t i = 0; i + 1 < cut.size(); i++) {
        if (hipEventSynchronize(evs[i]) != hipSuccess) result = MZD_E_DEVICE; // the job table was copied before the first slice
        if (result == MZD_OK)
            parallel_jobs(cut[i + 1] - cut[i], (cut[i + 1] < idx.size() ? out_off[cut[i + 1]] : out_total) - out_off[cut[i]], [&](size_t r) {
                const size_t k = cut[i] + r;
                mzd_job& j = jobs[idx[k]];
                j.status = d.h_jobs[k].status;
                j.out_len = (size_t)d.h_jobs[k].out_len;
                size_t ncopy = std::min<size_t>(j.out_len, j.dst_cap);
                if (ncopy && j.dst) memcpy(j.dst, d.h_out + out_off[k], ncopy);
            });
    }
    hipStreamSynchronize(s);
    for (auto ev : evs) if (ev) hipEventDestroy(ev);
    if (result != MZD_OK) return result;
    HIPCHK(hipEventElapsedTime(&d.last_ms, d.ev0, d.ev1));
    return MZD_OK;
}

Device* get_device(int i) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (i < 0 || (size_t)i >= g_dev.size()) return nullptr;
    return g_dev[(size_t)i].get();
}

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

} // namespace

struct mzd_batch {
    int device;
    DevJob* d_jobs;
    DevJob* h_jobs;
    size_t njobs;
};

extern "C" {

int mzd_init(const int* device_ids, int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& d : g_dev) free_device(*d);
    g_dev.clear();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MZD_E_DEVICE;
    std::vector<int> ids;
    if (!device_ids || n <= 0) ids.push_back(0);
    else ids.assign(device_ids, device_ids + n);
    for (int id : ids) {
        if (id < 0 || id >= count) { for (auto& d : g_dev) free_device(*d); g_dev.clear(); return MZD_E_PARAM; }
        auto d = std::make_unique<Device>();
        int rc = init_device(*d, id);
        if (rc) { free_device(*d); for (auto& e : g_dev) free_device(*e); g_dev.clear(); return rc; }
        g_dev.push_back(std::move(d));
    }
    return MZD_OK;
}

void mzd_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& d : g_dev) free_device(*d);
    g_dev.clear();
}

int mzd_device_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_dev.size();
}

// Frame header walk (RFC 8878 3.1.1): no entropy decoding, so it stays on the host.
uint64_t mzd_content_size(const uint8_t* src, size_t n) {
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
        if (fcsf == 0) { if (single) total += *q; else unknown = true; }
        else if (fcsf == 1) total += (uint64_t)rd16(q) + 256;
        else if (fcsf == 2) total += rd32(q);
        else total += rd64(q);
        size_t p = pos + hs;
        for (;;) { // block chain
            if (n - p < 3) return MZD_CONTENTSIZE_ERROR;
            uint32_t bh = rd16(src + p) | ((uint32_t)src[p + 2] << 16);
            p += 3;
            uint32_t type = (bh >> 1) & 3, bs = bh >> 3;
            if (type == 3) return MZD_CONTENTSIZE_ERROR;
            size_t adv = type == 1 ? 1 : bs;
            if (n - p < adv) return MZD_CONTENTSIZE_ERROR;
            p += adv;
            if (bh & 1) break;
        }
        if (fhd & 4) { if (n - p < 4) return MZD_CONTENTSIZE_ERROR; p += 4; }
        pos = p;
    }
    return unknown ? MZD_CONTENTSIZE_UNKNOWN : total;
}

int mzd_decode_batch(mzd_job* jobs, size_t njobs) {
    size_t ndev;
    { std::lock_guard<std::mutex> lk(g_mu); ndev = g_dev.size(); }
    if (ndev == 0) return MZD_E_DEVICE;
    if (!jobs && njobs) return MZD_E_PARAM;
    for (size_t i = 0; i < njobs; i++) { jobs[i].status = MZD_E_DEVICE; jobs[i].out_len = 0; }
    std::vector<std::vector<size_t>> shard(ndev);
    for (size_t i = 0; i < njobs; i++) shard[i % ndev].push_back(i); // file i -> GPU i mod N
    std::vector<int> rcs(ndev, MZD_OK);
    if (ndev == 1) {
        rcs[0] = run_host_jobs(*get_device(0), jobs, shard[0]);
    } else {
        std::vector<std::thread> th;
        for (size_t d = 0; d < ndev; d++)
            th.emplace_back([&, d] { rcs[d] = run_host_jobs(*get_device((int)d), jobs, shard[d]); });
        for (auto& t : th) t.join();
    }
    for (int rc : rcs) if (rc) return rc;
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
    Device* d = get_device(device);
    if (!d) return MZD_E_DEVICE;
    if (!jobs && njobs) return MZD_E_PARAM;
    std::lock_guard<std::mutex> lk(d->mu);
    return run_device_jobs(*d, jobs, njobs, stream ? (hipStream_t)stream : d->stream);
}

int mzd_batch_prepare(int device, const mzd_job* jobs, size_t njobs, mzd_batch** out) {
    Device* d = get_device(device);
    if (!d) return MZD_E_DEVICE;
    if (!jobs || !out || njobs == 0 || njobs > 0xFFFFFFF0u) return MZD_E_PARAM;
    std::lock_guard<std::mutex> lk(d->mu);
    HIPCHK(hipSetDevice(d->hip_id));
    auto* b = new mzd_batch{device, nullptr, nullptr, njobs};
    if (hipMalloc(&b->d_jobs, njobs * sizeof(DevJob)) != hipSuccess || hipHostMalloc(&b->h_jobs, njobs * sizeof(DevJob), hipHostMallocDefault) != hipSuccess) {
        hipFree(b->d_jobs); delete b; return MZD_E_DEVICE;
    }
    for (size_t i = 0; i < njobs; i++) {
        DevJob& j = b->h_jobs[i];
        j.src = jobs[i].src; j.src_len = jobs[i].src_len; j.dst = jobs[i].dst; j.dst_cap = jobs[i].dst_cap;
        j.out_len = 0; j.status = MZD_E_DEVICE; j.dict = jobs[i].dict_id;
    }
    if (hipMemcpy(b->d_jobs, b->h_jobs, njobs * sizeof(DevJob), hipMemcpyHostToDevice) != hipSuccess) { hipFree(b->d_jobs); hipHostFree(b->h_jobs); delete b; return MZD_E_DEVICE; }
    *out = b;
    return MZD_OK;
}

int mzd_batch_launch(mzd_batch* b, void* stream) {
    if (!b) return MZD_E_PARAM;
    Device* d = get_device(b->device);
    if (!d) return MZD_E_DEVICE;
    std::lock_guard<std::mutex> lk(d->mu);
    HIPCHK(hipSetDevice(d->hip_id));
    return enqueue(*d, b->d_jobs, (uint32_t)b->njobs, grid_for(*d, b->h_jobs, b->njobs), stream ? (hipStream_t)stream : d->stream);
}

int mzd_batch_collect(mzd_batch* b, mzd_job* jobs, void* stream) {
    if (!b) return MZD_E_PARAM;
    Device* d = get_device(b->device);
    if (!d) return MZD_E_DEVICE;
    std::lock_guard<std::mutex> lk(d->mu);
    HIPCHK(hipSetDevice(d->hip_id));
    hipStream_t s = stream ? (hipStream_t)stream : d->stream;
    HIPCHK(hipMemcpyAsync(b->h_jobs, b->d_jobs, b->njobs * sizeof(DevJob), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipEventElapsedTime(&d->last_ms, d->ev0, d->ev1));
    if (jobs)
        for (size_t i = 0; i < b->njobs; i++) { jobs[i].out_len = (size_t)b->h_jobs[i].out_len; jobs[i].status = b->h_jobs[i].status; }
    return MZD_OK;
}

void mzd_batch_free(mzd_batch* b) {
    if (!b) return;
    Device* d = get_device(b->device);
    if (d) hipSetDevice(d->hip_id);
    hipFree(b->d_jobs);
    hipHostFree(b->h_jobs);
    delete b;
}

int mzd_load_dict(const uint8_t* dict, size_t n, uint32_t* dict_id) {
    size_t ndev;
    { std::lock_guard<std::mutex> lk(g_mu); ndev = g_dev.size(); }
    if (ndev == 0) return MZD_E_DEVICE;
    if (!dict || n == 0 || n > 0x7FFFFFFFu || !dict_id) return MZD_E_PARAM;
    uint32_t handle = 0;
    for (size_t k = 0; k < ndev; k++) {
        Device& d = *get_device((int)k);
        std::lock_guard<std::mutex> lk(d.mu);
        HIPCHK(hipSetDevice(d.hip_id));
        if (d.ndicts >= kMaxDicts) return MZD_E_PARAM;
        uint8_t* buf = nullptr;
        int32_t* st = nullptr;
        HIPCHK(hipMalloc(&buf, align_up(n + MZD_SRC_PADDING, kAlign)));
        d.dict_bufs.push_back(buf);
        HIPCHK(hipMemset(buf, 0, align_up(n + MZD_SRC_PADDING, kAlign)));
        HIPCHK(hipMemcpy(buf, dict, n, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&st, 64));
        HIPCHK(hipMemset(st, 0xFF, 64));
        launch_dict_kernel(buf, (uint32_t)n, d.d_dicts + d.ndicts, st, d.stream);
        HIPCHK(hipGetLastError());
        int32_t status = MZD_E_DEVICE;
        HIPCHK(hipMemcpyAsync(&status, st, 4, hipMemcpyDeviceToHost, d.stream));
        HIPCHK(hipStreamSynchronize(d.stream));
        hipFree(st);
        if (status != MZD_OK) return status;
        d.ndicts++;
        handle = d.ndicts;
    }
    *dict_id = handle;
    return MZD_OK;
}

int mzd_debug_last_block(int device, uint8_t* lit, size_t lit_cap, size_t* n_lit, uint32_t* seq4, size_t seq_cap, size_t* n_seq) {
    Device* d = get_device(device);
    if (!d) return MZD_E_DEVICE;
    std::lock_guard<std::mutex> lk(d->mu);
    HIPCHK(hipSetDevice(d->hip_id));
    uint32_t slot = 0;
    HIPCHK(hipMemcpy(&slot, d->counter + 1, 4, hipMemcpyDeviceToHost));
    slot &= 0xFFFu; // task << 12 | slot
    if (slot >= d->max_wg) return MZD_E_PARAM;
    DebugSlot ds;
    HIPCHK(hipMemcpy(&ds, d->debug + slot, sizeof(ds), hipMemcpyDeviceToHost));
    if (n_lit) *n_lit = ds.n_lit;
    if (n_seq) *n_seq = ds.n_seq;
    if (lit && ds.n_lit) {
        const void* srcp = ds.lit_is_raw ? (const void*)(uintptr_t)ds.lit_raw_ptr : (const void*)(d->lit_scratch + (size_t)slot * kLitStride);
        HIPCHK(hipMemcpy(lit, srcp, std::min<size_t>(lit_cap, ds.n_lit), hipMemcpyDeviceToHost));
    }
    if (seq4 && ds.n_seq)
        HIPCHK(hipMemcpy(seq4, d->seq_scratch + (size_t)slot * kSeqStride, std::min<size_t>(seq_cap, ds.n_seq) * 16, hipMemcpyDeviceToHost));
    return MZD_OK;
}

// Diagnostic (libmzd_diag.so, built with -DMZD_STAMPS): per-phase cycle sums of the workgroup that ran job 0.
int mzd_debug_stamps(int device, uint64_t* out8) {
    Device* d = get_device(device);
    if (!d || !out8) return MZD_E_PARAM;
    std::lock_guard<std::mutex> lk(d->mu);
    HIPCHK(hipSetDevice(d->hip_id));
    uint32_t slot = 0;
    HIPCHK(hipMemcpy(&slot, d->counter + 1, 4, hipMemcpyDeviceToHost));
    slot &= 0xFFFu; // task << 12 | slot
    if (slot >= d->max_wg) return MZD_E_PARAM;
    DebugSlot ds;
    HIPCHK(hipMemcpy(&ds, d->debug + slot, sizeof(ds), hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out8[i] = ds.stamp[i];
    for (int i = 0; i < 8; i++) out8[8 + i] = ds.cstamp[i];
    for (int i = 0; i < 6; i++) out8[16 + i] = ds.tfin[i];
    return MZD_OK;
}

// Diagnostic: tfin[12] of every workgroup slot (n_slots * 12 values); returns the slot count.
int mzd_debug_tfin_all(int device, uint64_t* out, int max_slots) {
    Device* d = get_device(device);
    if (!d || !out) return MZD_E_PARAM;
    std::lock_guard<std::mutex> lk(d->mu);
    HIPCHK(hipSetDevice(d->hip_id));
    int n = std::min<int>((int)d->max_wg, max_slots);
    std::vector<DebugSlot> all((size_t)n);
    HIPCHK(hipMemcpy(all.data(), d->debug, sizeof(DebugSlot) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) for (int k = 0; k < 12; k++) out[(size_t)i * 12 + k] = all[(size_t)i].tfin[k];
    return n;
}

int mzd_last_kernel_ms(int device, float* ms) {
    Device* d = get_device(device);
    if (!d || !ms) return MZD_E_PARAM;
    *ms = d->last_ms;
    return MZD_OK;
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

const char* mzd_version(void) { return "mzd 0.1 (gfx950)"; }

} // extern "C"

// ---------------------------------------------------------------------------------------
// Host mirror of OpenedFiles (reference src/file.rs) + open/read/release wrappers.
// ---------------------------------------------------------------------------------------
struct DecodedFile { // plays the role of the anonymous tempfile (reference src/main.rs:462)
    std::vector<uint8_t> bytes;
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
    size_t cap = want == MZD_CONTENTSIZE_UNKNOWN ? std::max<size_t>(zst_len * 8, 1 << 20) : (size_t)want;
    for (int attempt = 0; attempt < 8; attempt++) { // frames without a content size: grow until it fits
        file->bytes.resize(cap);
        size_t out_len = 0;
        int rc = mzd_decode(zst, zst_len, file->bytes.data(), cap, &out_len);
        fs->decodes++;
        if (rc == MZD_OK) { file->bytes.resize(out_len); break; }
        if (rc == MZD_E_DSTSIZE && want == MZD_CONTENTSIZE_UNKNOWN && attempt < 7) { cap *= 4; continue; }
        return -EFAULT;
    }
    uint64_t fh;
    if (!fs->new_fh(&fh)) return -EBUSY; // src/main.rs:490
    fs->handlers[fh] = FileHandler{flags, false, file, true, ino};
    fs->inode_map[ino].insert(fh);
    if (real_size) *real_size = file->bytes.size(); // -> user.real_size xattr (src/main.rs:473-482)
    return (int64_t)fh;
}

int64_t mzd_fs_read(mzd_fs* fs, uint64_t fh, int64_t offset, uint32_t size, uint8_t* out) {
    if (!fs) return -EINVAL;
    std::lock_guard<std::mutex> lk(fs->mu);
    auto it = fs->handlers.find(fh);
    if (it == fs->handlers.end()) return -ENOENT; // src/main.rs:505
    const std::vector<uint8_t>& b = it->second.file->bytes;
    if (offset < 0) return -EINVAL;
    if ((uint64_t)offset >= b.size()) return 0; // read_at past EOF -> 0 bytes, then truncate (src/main.rs:506-511)
    size_t nread = std::min<size_t>(size, b.size() - (size_t)offset);
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

} // extern "C"
