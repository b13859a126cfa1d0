// mzd_device.h -- structures shared by the HIP kernels (mzd_kernels.hip) and the host
// runtime (mzd_host.cpp).  gfx950 only.
#pragma once
#include <stdint.h>

namespace mzd {

constexpr int kWG = 256;                     // threads per workgroup = 4 wavefronts of 64
constexpr uint32_t kBlockMax = 128u * 1024u; // Block_Maximum_Size upper bound (RFC 8878 3.1.1.2.4)
constexpr uint32_t kMaxSeq = 43691u;         // a block regenerates <= 128 KiB and every match is >= 3 bytes
constexpr uint32_t kLitStride = kBlockMax + 64;
constexpr uint32_t kSeqStride = kMaxSeq + 21; // uint4 entries per workgroup (multiple of 64 keeps 16-B alignment)

// One file.  Device mirror of mzd_job (include/mzd.h).
struct DevJob {
    const uint8_t* src;
    uint64_t src_len;
    uint8_t* dst;
    uint64_t dst_cap;
    uint64_t out_len;
    int32_t status;
    uint32_t dict;
};

// A parsed dictionary resident in HBM (A.7): tables are stored exactly as the kernel keeps
// them in LDS so that a frame start is a straight copy.
struct DevDict {
    uint64_t ll[512];
    uint64_t ml[512];
    uint64_t of[256];
    uint16_t huf[2048];
    uint32_t al[3];
    uint32_t huf_log;
    uint32_t rep[3];
    uint32_t dict_id;
    uint32_t formatted;
    uint32_t content_len;
    const uint8_t* content;
};

struct DebugSlot { // what workgroup `slot` last decoded (mzd_debug_last_block)
    uint32_t n_lit;
    uint32_t n_seq;
    uint32_t lit_is_raw;
    uint32_t pad;
    uint64_t lit_raw_ptr;
    uint64_t stamp[8]; // diagnostic build only (-DMZD_STAMPS): cycles per phase, summed over blocks
    uint64_t cstamp[8]; // same, for the copying wavefront
    uint64_t tfin[12];  // (6..11: headers parsed, Huffman weights decoded, Huffman table filled, copier started, spare, spare)
                        // -DMZD_TFIN: cycles after block start when walker / copier / hasher / planner finished, literals were ready, tables were ready
};

struct KernelArgs {
    DevJob* jobs;
    uint32_t njobs;
    uint32_t* counter;    // work queue head
    uint8_t* lit_scratch; // kLitStride bytes per workgroup
    uint4* seq_scratch;   // kSeqStride uint4 per workgroup
    uint2* walk_scratch;  // kSeqStride uint2 per workgroup (state-walk records)
    const DevDict* dicts;
    uint32_t ndicts;
    DebugSlot* debug;     // gridDim.x entries
    uint32_t* job_slot0;  // slot that ran job 0
};

void launch_decode(const KernelArgs& a, uint32_t grid, void* stream);
int kernel_lds_bytes();

} // namespace mzd
