/*
 * include/mzd.h -- C ABI of the MI355X zstd frame decoder ("mzd").
 *
 * This is the drop-in boundary for the ONE codec call on fuse-zstd's read side:
 *
 *     zstd::stream::copy_decode(source_file, target_file).map_err(|_| libc::EFAULT)?;
 *                                            reference src/main.rs:463-467 (inside
 *                                            ZstdFS::open_wrapper, :451-493, reached from
 *                                            <ZstdFS as Filesystem>::open, :980-992)
 *
 * copy_decode decodes EVERY concatenated frame of the file, skips skippable frames,
 * verifies the content checksum when present and fails on any malformed input; the caller
 * maps every failure to EFAULT (:467).  `Filesystem::read` (:931-956 -> read_wrapper
 * :495-513) only slices the decoded bytes, so the decoded buffer is all it needs.
 *
 * Everything below is plain pointers and sizes (no torch / HIP types).  The decode itself
 * runs in hand-written HIP kernels on gfx950; there is NO CPU fallback: every decode entry
 * point returns MZD_E_DEVICE when no GPU / kernel image is usable.
 *
 * INTEGRATION.md shows the Rust `extern "C"` block + the 6-line patch to open_wrapper.
 */
#ifndef MZD_H
#define MZD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (0 = ok, negative = error class).  The open() caller maps any
 *      non-zero to EFAULT exactly as reference src/main.rs:467 does. ---- */
#define MZD_OK 0
#define MZD_E_CORRUPT (-1)     /* malformed frame / block / entropy section */
#define MZD_E_TRUNCATED (-2)   /* input ends inside a frame */
#define MZD_E_CHECKSUM (-3)    /* XXH64 content checksum mismatch */
#define MZD_E_DSTSIZE (-4)     /* destination capacity too small */
#define MZD_E_UNSUPPORTED (-5) /* reserved bit set, window > 128 MiB (streaming limit of copy_decode) */
#define MZD_E_DEVICE (-6)      /* no GPU, HIP error, library not initialised */
#define MZD_E_BADMAGIC (-7)    /* data that is neither a zstd nor a skippable frame */
#define MZD_E_DICT (-8)        /* frame needs a dictionary that is missing / different / corrupt */
#define MZD_E_PARAM (-9)       /* bad argument */

/* Device-resident inputs must be readable for MZD_SRC_PADDING bytes past src_len (the
 * kernels read the bitstreams with unaligned 8-byte loads).  Host-pointer entry points
 * stage the input themselves and need no padding from the caller. */
#define MZD_SRC_PADDING 16

#define MZD_CONTENTSIZE_UNKNOWN (UINT64_MAX)
#define MZD_CONTENTSIZE_ERROR (UINT64_MAX - 1)

/* One file (= what one open() decodes).  Replaces the (source, destination) pair of
 * copy_decode, reference src/main.rs:463-466. */
typedef struct mzd_job {
    const uint8_t* src; /* whole .zst file: all frames */
    size_t src_len;
    uint8_t* dst;       /* decoded bytes */
    size_t dst_cap;
    size_t out_len;     /* OUT: decoded length (what open_wrapper stores in user.real_size, :473-482); with status MZD_E_DSTSIZE on host
                           pointers: the capacity that would have sufficed (mzd_content_bound) */
    int32_t status;     /* OUT: MZD_OK or MZD_E_* */
    uint32_t dict_id;   /* 0 = none, else a handle from mzd_load_dict */
    int32_t device;     /* OUT: index (into mzd_init's list) of the GPU that decoded the job; -1 if none did */
} mzd_job;

/* Initialise `n` devices (HIP ordinals); ids == NULL / n == 0 -> device 0 only.
 * Allocates per-device scratch (literal + sequence buffers for every resident workgroup).
 * Calling it again re-initialises.  Returns MZD_OK or MZD_E_DEVICE. */
int mzd_init(const int* device_ids, int n);

/* The same with the per-device memory spelled out (mzd_init uses the defaults: about 2.3 GB per device).  Nothing here or in
 * mzd_init touches the process environment.  The host path keeps four kernel streams and two copy streams busy per device;
 * the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime starts: a
 * process that wants them all concurrent exports GPU_MAX_HW_QUEUES=8 itself before its first HIP call (bench.py and
 * INTEGRATION.md's daemon patch do; with the default the streams share queues -- correct, a few percent slower end to end).
 * Unknown trailing fields are ignored: set struct_size = sizeof(mzd_config). */
typedef struct mzd_config {
    size_t struct_size;
    const int* device_ids;        /* HIP ordinals; NULL / 0 -> device 0 only.  An ordinal may appear more than once:
                                     each entry is a device of its own to the library (tests of the N-device path on one card) */
    int n_devices;
    uint32_t max_workgroups;      /* resident workgroups the general drivers may use per device (0: all the device holds,
                                     1 024 on MI355X).  Their scratch is 1.6 MB each: literals, walk records, plan */
    size_t small_scratch_bytes;   /* scratch of the small-file kernel per device: literals + sequences of every resident
                                     small file (0: 512 MiB; at least 16 MiB) */
    int resolve_ahead;            /* 1 (default when struct_size does not reach it): byte maps for blocks resolved ahead of
                                     their predecessors (0.5 MB per resident workgroup); 0: none, blocks copy in order */
} mzd_config;
int mzd_init_ex(const mzd_config* cfg);
void mzd_shutdown(void);
int mzd_device_count(void); /* devices initialised by mzd_init (0 before) */

/* Pinned host memory that every initialised GPU copies from / into directly.  Optional: mzd_decode_batch takes any host
 * pointers; buffers from this allocator skip the staging copy on the host (the caller of reference src/main.rs:463 would
 * read the .zst into / decode into such buffers instead of a tempfile).  NULL when the allocation fails. */
void* mzd_host_alloc(size_t n);
void mzd_host_free(void* p);

/* Sum of Frame_Content_Size over all frames of a file (host-side header walk, no GPU):
 * the analogue of ZSTD_getFrameContentSize the caller uses to size `dst`.
 * MZD_CONTENTSIZE_UNKNOWN if some frame omits the field, MZD_CONTENTSIZE_ERROR if the
 * headers are malformed.  Frames written by the reference always carry it (src/main.rs:785-788). */
uint64_t mzd_content_size(const uint8_t* src, size_t n);
/* A capacity that is certain to hold the decoded file: frames that state their content size count with it (so the result is
 * mzd_content_size's whenever that is known), the others with what their block headers allow at most -- a raw or RLE block
 * its stated size, a compressed block min(128 KiB, window).  Host-side header walk, no GPU.  MZD_CONTENTSIZE_ERROR if the
 * headers are malformed.  SURVEY.md 8(b) "Ownership": a destination that was too small comes back as MZD_E_DSTSIZE with this
 * value in out_len (host-pointer entry points; 0 when the headers cannot be walked), so a file without content sizes costs at
 * most two decodes (mzd_fs_open: a guess, then this). */
uint64_t mzd_content_bound(const uint8_t* src, size_t n);

/* copy_decode on host buffers: stages src to the GPU, decodes, copies the result back.
 * Single call site shape of reference src/main.rs:463.  Thread-compatible with the
 * single-threaded fuser loop (reference DESIGN.md:5-7); also safe from several threads. */
int mzd_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len);

/* Many files per call, HOST pointers.  Jobs are dealt round-robin over the initialised
 * devices (job i -> device i mod N; no collective: files are independent).  Per-job
 * status/out_len/device are filled in.  Returns MZD_OK if the batch ran (inspect job status),
 * MZD_E_DEVICE / MZD_E_PARAM otherwise.  On each device the batch crosses PCIe as a pipeline of
 * chunks (copy in | decode | copy out on separate streams); several threads may call at once.
 * Bytes of dst between out_len and dst_cap are unspecified afterwards (never anything past dst_cap). */
int mzd_decode_batch(mzd_job* jobs, size_t njobs);

/* Many files per call, DEVICE pointers on `device` (index into mzd_init's list): src/dst
 * already in HBM, nothing crosses PCIe except the job table.  `stream` is a hipStream_t
 * (NULL = the library's own stream).  Blocks until the results are back. */
int mzd_decode_batch_device(int device, mzd_job* jobs, size_t njobs, void* stream);

/* Asynchronous form for measurement loops: the job table stays on the device.
 * prepare uploads the table once; launch enqueues one pass on `stream` and returns at
 * once; collect waits for the stream and copies status/out_len back into `jobs`.
 * A launch that holds nothing but small files (capacity <= 8 KiB) ends with the small-file kernel; the few files that kernel hands
 * on -- several frames or blocks, anything malformed -- are decoded by the general driver when the launch is collected (collect,
 * or the end of mzd_decode_batch_device): their outputs and statuses are complete when collect returns, not before. */
typedef struct mzd_batch mzd_batch;
int mzd_batch_prepare(int device, const mzd_job* jobs, size_t njobs, mzd_batch** out);
int mzd_batch_launch(mzd_batch* b, void* stream);
/* The same with flags.  MZD_LAUNCH_UNTIMED: the launch does not record its start event (mzd_last_kernel_ms keeps the figure of the last
 * timed launch) -- for loops that time many back-to-back launches themselves: an event is a packet between two kernels. */
#define MZD_LAUNCH_UNTIMED 1u
int mzd_batch_launch_ex(mzd_batch* b, void* stream, unsigned flags);
int mzd_batch_collect(mzd_batch* b, mzd_job* jobs, void* stream);
void mzd_batch_free(mzd_batch* b);

/* Dictionaries (ZSTD_dct_auto: formatted when it starts with 0xEC30A437, else raw
 * content).  Uploaded once to every initialised device.  The reference itself cannot open
 * dictionary frames (copy_decode has none); this serves BASELINE config 5. */
int mzd_load_dict(const uint8_t* dict, size_t n, uint32_t* dict_id);
/* Frees a dictionary on every device (a long-lived daemon cycles through more than the 64 that fit at once).  Jobs that
 * still name the handle fail with MZD_E_DICT.  MZD_E_PARAM for a handle that is not loaded. */
int mzd_unload_dict(uint32_t dict_id);

/* Test hook: literal buffer and sequence triples {ll, ml, off, 0} (u32 x 4) of the LAST
 * compressed block decoded by the workgroup that ran job 0 of the previous call on
 * `device`, for phase-by-phase comparison with the oracle's trace. */
int mzd_debug_last_block(int device, uint8_t* lit, size_t lit_cap, size_t* n_lit,
                         uint32_t* seq4, size_t seq_cap, size_t* n_seq);

/* Diagnostics (tests, tools/): which kernel a launch takes.  0 automatic (small files, when there are thousands of them:
 * the small-file kernel; 3: that kernel for every eligible file however few; then a
 * workgroup per file, or block tasks when a file can have several blocks), 1 / 2: that general driver only; 4 / 5: block
 * tasks, with / without resolving blocks ahead of their predecessors whatever the size of the launch. */
int mzd_debug_set_driver(int driver);
/* The 8 counter words of the launch that decoded job 0 of the most recent call: [0] queue tickets, [2] block tasks pushed,
 * [3] files finished by the block-task driver, [4] small files the small-file kernel handed on to the general driver,
 * [5] groups it took. */
int mzd_debug_counters(int device, uint32_t* out8);
/* Host-path diagnostics.  what 2: at most `value` chunks per call when the kernels mirror the outputs into pinned caller
 * memory (default 4).  what 3: `value` host threads for the staging copies of pageable buffers (default 8).  what 4 / 5: the small-file
 * kernel's files per wavefront / files executed at a time (0: the library's choice).  what 6: its grid.  what 7: a timing trace of
 * mzd_decode_batch on stderr.  what 8: a launch of small files alone keeps the general driver's launch behind it (A/B).  what 9: the small-file kernel's wavefronts
 * per workgroup (0: the library's choice, 1: never a helper wavefront, 2: with the 8 / 4 shape always).  what 10: how block tasks execute a
 * file's blocks (0: the library's choice, 1 in order, 2 every task resolved ahead, 3 only behind a running predecessor, 4 every other task).
 * what 11: driver 1's workgroups (0 / 1: one file each; 2: two files each, their sequence chains walked by ONE wavefront --
 * mzd_decode_kernel_pairs, built and measured in round 6, never chosen by the library: it is slower).  what 12: decoding wavefronts around ONE
 * dictionary table image in the small-file kernel, for launches whose files all name the same dictionary (0: the library's choice, 1 / 5 / 8).
 * what 11 = 3: driver 1 with workgroups of three wavefronts, five to a CU (mzd_decode_kernel_files3; measured, never chosen).  what 13: host-path
 * experiments of profiles/r06_t2_pairs.txt (bit 0: one call at a time inside the submission loop; bit 1: chunks retired by polling hipEventQuery). */
int mzd_debug_host_path(int device, int what, int value);
/* Diagnostic builds only (make diag / tfin): per-phase cycle sums of the workgroup that ran job 0; role finish times of
 * every workgroup slot.  In the product build they return zeros. */
int mzd_debug_stamps(int device, uint64_t* out22);
/* (a build with -DMZD_SMALL_STAMPS, `make sstamps`) the small-file kernel's phase stamps: 1 032 values, tools/lds_stamps.py */
int mzd_debug_small_stamps(int device, uint64_t* out);
/* (the same build) per workgroup of the small-file kernel's last launch, 16 values each: 100 MHz clock at entry, at exit,
 * HW_ID | XCC_ID << 32, groups | rounds | steps, then the clock at the phase boundaries of its first group; n <= 3 072: tools/lds_wg.py */
int mzd_debug_small_wg_stamps(int device, uint64_t* out, int n);
/* the small-file kernel's intermediates of resident file slot `slot` after a call on device pointers: literals, and sequences as
 * literal length | match length << 14 | offset value << 32 (before repeat-offset resolution) */
int mzd_debug_small_scratch(int device, uint32_t slot, uint8_t* lit, size_t lit_n, uint64_t* seq, size_t seq_n);
int mzd_debug_tfin_all(int device, uint64_t* out, int max_slots);
/* Host-side test hook (no GPU needed): the lazy open's index of `zst` and the synthetic frame of the first `nblocks` blocks of
 * frame `frame` (what a partial read decodes).  Returns the frames indexed, 0 when the file is not seekable. */
int mzd_debug_lazy_plan(const uint8_t* zst, size_t n, uint32_t frame, uint32_t nblocks, uint8_t* synth, size_t cap, size_t* synth_len, uint64_t* total, uint32_t* nblocks_of_frame);

/* Milliseconds the decode kernel of the last launch on `device` took (hipEvents on the
 * launch stream).  Valid after the launch has been collected. */
int mzd_last_kernel_ms(int device, float* ms);
/* The kernels the most recent launch on device pointers ran on `device`, the one that did most of the work first, '+' between
 * them: "mzd_lds_kernel<8,false,4>+mzd_decode_kernel_files", "mzd_decode_kernel_tasks", ... (the library's own record of what
 * make_plan chose; benchmarks quote it instead of guessing).  Valid until the next launch on that device. */
const char* mzd_last_kernel_name(int device);

const char* mzd_strerror(int code);
const char* mzd_version(void);

/* ------------------------------------------------------------------------------------
 * Host mirror of the caller's side of the path: the file-handle table that owns the
 * decoded bytes (reference src/file.rs:10-135 OpenedFiles/FileHandler) and the two
 * operations on it (open_wrapper src/main.rs:451-493, read_wrapper :495-513,
 * release_wrapper :595-599).  The decoded buffer plays the role of the tempfile.
 * Return values are the reference's: a file handle / byte count, or a NEGATIVE errno
 * (EFAULT for any decode failure :467, ENOENT for an unknown handle :505, EBUSY :490).
 * ------------------------------------------------------------------------------------ */
typedef struct mzd_fs mzd_fs;
mzd_fs* mzd_fs_new(void);
void mzd_fs_free(mzd_fs* fs);
/* open: `duplicate` an existing handle of `ino` without decoding (src/file.rs:67-102),
 * else decode `zst` (the bytes of <data_dir>/name.zst) on the GPU and insert (:47-65).
 * *real_size receives what goes into the user.real_size xattr (:473-482). */
int64_t mzd_fs_open(mzd_fs* fs, uint64_t ino, int32_t flags, const uint8_t* zst, size_t zst_len, uint64_t* real_size);
/* open without decoding (SURVEY.md 8f "lazy / seekable read"): the host walks frame and block headers; each read then decodes
 * only the frames that cover its range and, inside a frame, the blocks up to the range's end (kept for later reads).  Same
 * return values as mzd_fs_open; *real_size is the sum of the frames' content sizes.  A file with a frame that states no
 * content size is decoded eagerly (mzd_fs_open).  A corrupt file shows as -EFAULT at the first read that touches the damage
 * (the reference finds it at open: src/main.rs:467). */
int64_t mzd_fs_open_lazy(mzd_fs* fs, uint64_t ino, int32_t flags, const uint8_t* zst, size_t zst_len, uint64_t* real_size);
/* read: bytes [offset, offset+size) of the decoded file, short at EOF (:506-511). */
int64_t mzd_fs_read(mzd_fs* fs, uint64_t fh, int64_t offset, uint32_t size, uint8_t* out);
/* release: drops the handle; the decoded bytes go when the last handle of the inode goes. */
int mzd_fs_release(mzd_fs* fs, uint64_t fh);
/* number of GPU decodes performed so far (second opens of an inode must not add one). */
uint64_t mzd_fs_decode_count(const mzd_fs* fs);
/* bytes the GPU has produced for this table so far (a lazy file read in part decodes less than its size). */
uint64_t mzd_fs_decoded_bytes(const mzd_fs* fs);

#ifdef __cplusplus
}
#endif
#endif /* MZD_H */
