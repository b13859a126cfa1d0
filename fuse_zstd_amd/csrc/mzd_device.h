// mzd_device.h -- structures shared by the HIP kernels (mzd_kernels.hip) and the host
// runtime (mzd_host.cpp).  gfx950 only.
#pragma once
#include <stdint.h>

namespace mzd {

#if defined(MZD_W3) && MZD_W3
constexpr int kWG = 192;                     // the MZD_W3 build of driver 1: workgroups of three wavefronts (mzd_k_common.h)
#else
constexpr int kWG = 256;                     // threads per workgroup = 4 wavefronts of 64
#endif
constexpr uint32_t kBlockMax = 128u * 1024u; // Block_Maximum_Size upper bound (RFC 8878 3.1.1.2.4)
constexpr uint32_t kMaxSeq = 43691u;         // a block regenerates <= 128 KiB and every match is >= 3 bytes
constexpr uint32_t kLitStride = kBlockMax + 64;
constexpr uint32_t kResMapStride = kBlockMax + 64; // words per workgroup slot of the resolve map (mzd_k_resolve.h)
// A Huffman decode table: 2 048 entries (symbol | length << 8) indexed by the next 11 bits of the stream, or -- a tree of depth 12,
// which libzstd accepts (HUF_TABLELOG_MAX) although no encoder emits one -- indexed by the top 11 of the next 12 bits: an entry then
// stands for two neighbouring 12-bit codes; where those are two codes of length 12 (two symbols of weight 1: they come in pairs), the
// entry holds the even one's symbol and the odd one's sits in the 128 bytes behind the table (byte k for entry k).
constexpr uint32_t kHufEntries = 2048 + 64, kHufWords = kHufEntries / 2;
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
    uint8_t* dst2;   // optional mirror of the output in the caller's pinned HOST memory, written by the kernel as the file is
                     // decoded (the host path: the copy back over PCIe overlaps the decode instead of following it); null: none
};

// A parsed dictionary resident in HBM (A.7): tables are stored exactly as the block pipeline keeps
// them in LDS so that a frame start is a straight copy.  An FSE entry's low word is the LDS ADDRESS of the next state's base entry
// in that pipeline's image (mzd_k_common.h: Shared; asserted there): the small-file kernel, whose tables live elsewhere, rebases them.
constexpr uint32_t kBlkLdsLL = 8208, kBlkLdsML = 12304, kBlkLdsOF = 16400;
struct DevDict {
    uint64_t ll[512];
    uint64_t ml[512];
    uint64_t of[256];
    uint16_t huf[kHufEntries];
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

// ---- block tasks (N1: the blocks of one frame run on different workgroups) -------------------------------
// A workgroup decodes ONE block per task.  Task 0 of file j is handed out by the ticket counter; the task of
// the next block (or of the next frame) of the same file is pushed into a ring of continuation records by its
// predecessor as soon as that one has read its own block header, i.e. long before it has decoded anything.
// What a block needs from its predecessor travels through the file's FileState / TableArea, in task order:
//   * entropy tables (treeless literals, repeat-mode FSE tables): `tables_ver` = number of tasks whose tables
//     are in the area; a task reads what it inherits at version t, writes what it rebuilt, and publishes t + 1;
//   * output position, repeat offsets, XXH64 state, first error: published with `copied` = t + 1 when the
//     task's copy + hash are complete.  The copy stage of task t starts when copied == t.
struct FileState { // zeroed per launch
    uint32_t copied;
    uint32_t tables_ver;
    int32_t err;            // first error of the file, in task order
    uint32_t huf_valid, fse_valid, huf_log;
    uint32_t al[3];
    uint32_t rep[3];
    uint64_t out;           // bytes produced by the tasks completed so far
    uint64_t frame_out0;    // `out` at the start of the current frame
    uint64_t xstripes;      // XXH64: 32-byte stripes of the current frame absorbed so far
    uint64_t xxh[4];        // XXH64 accumulators
    // launches that resolve blocks ahead of their predecessors (KernelArgs::resolve): the repeat offsets travel on a chain
    // of their own, as soon as a task's plan is complete -- rep_e = the offsets after task rep_ver - 1
    uint32_t rep_ver;
    uint32_t rep_e[3];
    // The checksum chain (XXH64: strictly serial through a frame, 46 cycles per 32-byte stripe) is published separately from the
    // bytes: hashed = t + 1 means xstripes / xxh[] hold the state after task t.  Most tasks publish both together; a task that
    // resolved its block ahead hands `copied` over as soon as its bytes are gathered and `hashed` when its hash is done, so the
    // successor's gather does not wait for it.  A checksum failure found that way is posted here (first one wins): by then the
    // bytes' chain has moved on with err = 0.
    uint32_t hashed;
    int32_t herr;
    uint64_t herr_out;      // `out` before the task that found it
};
static_assert(sizeof(FileState) == 136, "FileState layout");

struct TableArea { // the entropy tables in force after the file's most recently published task (LDS layout)
    uint64_t ll[512];
    uint64_t ml[512];
    uint64_t of[256];
    uint16_t huf[kHufEntries];
};

struct ContRecord { // one pushed task: where it starts and the frame it belongs to
    uint64_t seq;           // launch epoch << 32 | (push index + 1) once the record is complete
    uint64_t pos;           // input offset of the block header (in_frame) or of the next frame header
    uint64_t fcs;
    uint32_t job, task;
    uint32_t in_frame;      // 0: starts at a frame header (or at the end of the file)
    uint32_t has_fcs, has_cksum, block_max, with_dict;
    uint32_t pad[3];
};
static_assert(sizeof(ContRecord) == 64, "ContRecord layout");

// counter words (uint32): 0 ticket, 1 slot that ran the last task of job 0, 2 pushes, 3 files finished,
// 4 jobs the small-file kernel handed on (appended to job_list behind its fixed part), 5 the small-file kernel's group ticket
constexpr uint32_t kCounterWords = 8;
struct KernelArgs {
    DevJob* jobs;
    uint32_t njobs;
    uint32_t* counter;    // see above; all zero when the launch starts ...
    uint32_t* counter_next; // ... because the launch before it on this lane cleaned it: the last workgroup to finish zeroes the block
                          // of the NEXT launch (two blocks per lane alternate; launches of a lane run one after the other)
    uint8_t* lit_scratch; // kLitStride bytes per workgroup
    uint4* seq_scratch;   // kSeqStride uint4 per workgroup
    uint4* walk_scratch;  // kSeqStride uint4 per workgroup (state-walk records: LL, ML, OF state offsets, read head - 32)
    const DevDict* dicts;
    uint32_t ndicts;
    DebugSlot* debug;     // gridDim.x entries
    uint32_t* job_slot0;  // = counter + 1
    FileState* fstate;    // njobs entries, zeroed per launch
    TableArea* tables;    // njobs entries
    ContRecord* ring;     // ring_cap entries
    uint32_t ring_cap;    // >= njobs + 1 (at most one unread continuation per file)
    uint32_t epoch;       // distinguishes this launch's ring records from stale ones
    uint32_t use_tasks;   // 1: driver 2 (block tasks), 0: driver 1 (a workgroup per file)
    // Launches behind the small-file kernel decode a LIST of jobs: job_list[0 .. nlist_fixed) chosen by the host (files that
    // are not small) followed by counter[4] entries appended on the device (small files that were not plain).  null: all jobs.
    const uint32_t* job_list;
    uint32_t nlist_fixed;
    uint32_t wg0;         // first workgroup slot of this launch in the scratch arrays (several launches may be in flight)
    // Block-task launches with few tasks for the machine (a single big file: the in-order copy stage is the whole critical
    // path): every block's sequences are resolved to a byte map -- each output byte's ultimate source, a literal or a byte
    // older than the block -- BEFORE its predecessor has finished (mzd_k_resolve.h), so that the in-order stage is a gather.
    uint32_t* resolve_map; // kBlockMax words per workgroup slot, or null
    uint32_t resolve;      // 1: use it
};

// ---- the small-file kernel (mzd_lds.hip) ---------------------------------------------------------------------------------
constexpr uint32_t kSmallCap = 8192;                   // eligible: dst_cap <= kSmallCap ...
constexpr uint32_t kSmallSrcMax = kSmallCap + 1024;    // ... and src_len <= kSmallSrcMax

// The whole file in LDS: G files per wavefront, 64 / G lanes per file; a file's
// slot = [tables | ring | compressed input | output window], sized by the host for the launch's largest file
struct LdsArgs {
    DevJob* jobs;
    const uint32_t* list;       // job indices, sorted by dictionary
    uint32_t n;
    uint32_t* counter;          // the launch's counter block (words 4 and 5)
    uint32_t* redo_list;        // = job_list + nlist_fixed
    const DevDict* dicts;
    uint32_t ndicts;
    uint32_t tab_bytes;         // per file: Huffman table, then the three FSE tables (0: files use the dictionary's tables only)
    uint32_t comp_bytes;        // per file: >= the largest src_len of the launch + 16, a multiple of 16
    uint32_t out_bytes;         // per file: >= the largest dst_cap of the launch + 16, a multiple of 16 (the window lies over the other three)
    uint8_t* scratch;           // HBM: per resident file (gridDim.x * G of them) lit_stride bytes of literals + 8 * seq_cap bytes of sequences
    uint32_t lit_stride;        // >= the largest dst_cap of the launch + 64
    uint32_t seq_cap;           // >= the largest dst_cap of the launch / 3 + 2 (every match is >= 3 bytes)
    uint64_t* stamps;           // diagnostic build (-DMZD_SMALL_STAMPS), else unused
    uint32_t* counter_next;     // null: a general driver's launch follows this one (it takes what is handed on and cleans the next
    uint32_t* handed_on;        // launch's counters).  Else: no launch follows; the last wavefront stores counter[4] here and cleans.
};
int launch_lds(const LdsArgs& a, uint32_t grid, int g, int xg, int with_dict, int nw, int nd, void* stream); // g files per wavefront, executed xg at a time; nw = 2: with a helper wavefront; nd: decoding wavefronts around one dictionary image
int lds_prepare_device();   // once per device, with that device current
uint32_t lds_kernel_bytes(int g, int xg, int with_dict, uint32_t tab_bytes, uint32_t comp_bytes, uint32_t out_bytes, int nw = 1, int nd = 1);
uint32_t lds_kernel_bytes_per_file(uint32_t tab_bytes, uint32_t comp_bytes);
uint32_t lds_waves_by_registers(int g, int xg, int with_dict, int nw = 1, int nd = 1);
size_t lds_scratch_per_file(uint32_t lit_stride, uint32_t seq_cap);
uint32_t lds_spare_table_bytes(uint32_t comp_bytes, uint32_t out_bytes);


void launch_decode(const KernelArgs& a, uint32_t grid, void* stream);
// driver 1 with two files a workgroup (mzd_kernels.hip compiled with MZD_PAIRS = 1: mzd_decode_kernel_pairs); grid counts GROUPS of four wavefronts
void launch_decode_pairs(const KernelArgs& a, uint32_t grid, void* stream);
int pairs_prepare_device();
// driver 1 with workgroups of three wavefronts, five to a CU (mzd_kernels.hip compiled with MZD_W3 = 1: mzd_decode_kernel_files3)
void launch_decode_w3(const KernelArgs& a, uint32_t grid, void* stream);
int w3_workgroups_per_cu();
int kernel_lds_bytes();

} // namespace mzd
