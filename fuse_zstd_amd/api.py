"""ctypes binding of include/mzd.h.  See the package docstring."""
import ctypes as C
import errno
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmzd.so")

OK, E_CORRUPT, E_TRUNCATED, E_CHECKSUM, E_DSTSIZE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
E_DEVICE, E_BADMAGIC, E_DICT, E_PARAM = -6, -7, -8, -9
SRC_PADDING = 16
CONTENTSIZE_UNKNOWN = 2**64 - 1
CONTENTSIZE_ERROR = 2**64 - 2

# every symbol include/mzd.h declares
EXPORTS = [
    "mzd_init", "mzd_init_ex", "mzd_shutdown", "mzd_device_count", "mzd_content_size", "mzd_content_bound", "mzd_decode", "mzd_decode_batch",
    "mzd_decode_batch_device", "mzd_batch_prepare", "mzd_batch_launch", "mzd_batch_launch_ex", "mzd_batch_collect", "mzd_batch_free",
    "mzd_load_dict", "mzd_unload_dict", "mzd_debug_last_block", "mzd_debug_set_driver", "mzd_debug_counters", "mzd_debug_host_path", "mzd_debug_stamps", "mzd_debug_small_stamps", "mzd_debug_small_wg_stamps", "mzd_debug_small_scratch", "mzd_debug_tfin_all", "mzd_debug_lazy_plan",
    "mzd_host_alloc", "mzd_host_free", "mzd_last_kernel_ms", "mzd_last_kernel_name", "mzd_strerror", "mzd_version",
    "mzd_fs_new", "mzd_fs_free", "mzd_fs_open", "mzd_fs_open_lazy", "mzd_fs_read", "mzd_fs_release", "mzd_fs_decode_count", "mzd_fs_decoded_bytes",
]


class MzdError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("%s: %s (%d)" % (what or "mzd", strerror(code), code))


class Job(C.Structure):  # mzd_job
    _fields_ = [("src", C.c_void_p), ("src_len", C.c_size_t), ("dst", C.c_void_p), ("dst_cap", C.c_size_t),
                ("out_len", C.c_size_t), ("status", C.c_int32), ("dict_id", C.c_uint32), ("device", C.c_int32)]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of libmzd.so, in-tree (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in ("mzd_kernels.hip", "mzd_lds.hip", "mzd_host.cpp", "mzd_device.h", "mzd_tables.h") + tuple(f for f in os.listdir(src_dir) if f.startswith(("mzd_k_", "mzd_l_")))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "mzd.h"))
    def stale():
        return force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale():
        # several ranks of one node may get here at once (bench.py under torch.distributed.run): one builds, the others wait and
        # find the library fresh
        import fcntl
        with open(os.path.join(src_dir, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.check_call(["make", "-C", src_dir, "-s", "-j5"])
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return _SO


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm bundles its own libamdhip64.so; two HIP runtimes in one process cannot both
    own the GPU (the second reports "No HIP GPUs").  Load torch's copy first (same soname), so
    libmzd.so binds to it and a later `import torch` finds it already there.  Without torch on
    the machine libmzd.so simply uses /opt/rocm's runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Loads libmzd.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise MzdError(E_DEVICE, "libmzd.so is missing: run fuse_zstd_amd.build() / make -C fuse_zstd_amd/csrc")
        _share_hip_runtime_with_torch()
        L = C.CDLL(_SO)
        L.mzd_init.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.mzd_content_size.restype = C.c_uint64
        L.mzd_content_size.argtypes = [C.c_char_p, C.c_size_t]
        if hasattr(L, "mzd_content_bound"):  # (tools/ load older builds of the library side by side)
            L.mzd_content_bound.restype = C.c_uint64
            L.mzd_content_bound.argtypes = [C.c_char_p, C.c_size_t]
        L.mzd_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mzd_decode_batch.argtypes = [C.POINTER(Job), C.c_size_t]
        L.mzd_decode_batch_device.argtypes = [C.c_int, C.POINTER(Job), C.c_size_t, C.c_void_p]
        L.mzd_batch_prepare.argtypes = [C.c_int, C.POINTER(Job), C.c_size_t, C.POINTER(C.c_void_p)]
        L.mzd_batch_launch.argtypes = [C.c_void_p, C.c_void_p]
        if hasattr(L, "mzd_batch_launch_ex"):  # (A/B runs load older builds of the library)
            L.mzd_batch_launch_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.mzd_batch_collect.argtypes = [C.c_void_p, C.POINTER(Job), C.c_void_p]
        L.mzd_batch_free.argtypes = [C.c_void_p]
        L.mzd_batch_free.restype = None
        L.mzd_load_dict.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.mzd_unload_dict.argtypes = [C.c_uint32]
        L.mzd_debug_set_driver.argtypes = [C.c_int]
        L.mzd_host_alloc.restype = C.c_void_p
        L.mzd_host_alloc.argtypes = [C.c_size_t]
        L.mzd_host_free.restype = None
        L.mzd_host_free.argtypes = [C.c_void_p]
        L.mzd_debug_last_block.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        L.mzd_last_kernel_ms.argtypes = [C.c_int, C.POINTER(C.c_float)]
        if hasattr(L, "mzd_last_kernel_name"):
            L.mzd_last_kernel_name.restype = C.c_char_p
            L.mzd_last_kernel_name.argtypes = [C.c_int]
        L.mzd_strerror.restype = C.c_char_p
        L.mzd_strerror.argtypes = [C.c_int]
        L.mzd_version.restype = C.c_char_p
        L.mzd_fs_new.restype = C.c_void_p
        L.mzd_fs_free.argtypes = [C.c_void_p]
        L.mzd_fs_free.restype = None
        L.mzd_fs_open.restype = C.c_int64
        L.mzd_fs_open.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.mzd_fs_open_lazy.restype = C.c_int64
        L.mzd_fs_open_lazy.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.mzd_fs_decoded_bytes.restype = C.c_uint64
        L.mzd_fs_decoded_bytes.argtypes = [C.c_void_p]
        L.mzd_fs_read.restype = C.c_int64
        L.mzd_fs_read.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_void_p]
        L.mzd_fs_release.argtypes = [C.c_void_p, C.c_uint64]
        L.mzd_fs_decode_count.restype = C.c_uint64
        L.mzd_fs_decode_count.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def strerror(code):
    try:
        return lib().mzd_strerror(code).decode()
    except MzdError:
        return "error %d" % code


class Config(C.Structure):  # mzd_config
    _fields_ = [("struct_size", C.c_size_t), ("device_ids", C.POINTER(C.c_int)), ("n_devices", C.c_int), ("max_workgroups", C.c_uint32),
                ("small_scratch_bytes", C.c_size_t), ("resolve_ahead", C.c_int)]


def init(device_ids=None, max_workgroups=0, small_scratch_bytes=0, resolve_ahead=True):
    """mzd_init; with any of the keyword arguments, mzd_init_ex (per-device memory spelled out).  An ordinal may be listed
    more than once: every entry is a device of its own to the library."""
    L = lib()
    arr = (C.c_int * len(device_ids))(*device_ids) if device_ids else None
    if max_workgroups or small_scratch_bytes or not resolve_ahead:
        cfg = Config(C.sizeof(Config), arr, len(device_ids) if device_ids else 0, max_workgroups, small_scratch_bytes, 1 if resolve_ahead else 0)
        L.mzd_init_ex.argtypes = [C.POINTER(Config)]
        rc = L.mzd_init_ex(C.byref(cfg))
    else:
        rc = L.mzd_init(arr, len(device_ids) if device_ids else 0)
    if rc != OK:
        raise MzdError(rc, "mzd_init")


def shutdown():
    lib().mzd_shutdown()


def device_count():
    return lib().mzd_device_count()


def content_size(src):
    src = bytes(src)
    return lib().mzd_content_size(src, len(src))


def content_bound(src):
    """A capacity certain to hold the decoded file (mzd_content_bound): the content size where the frames state it, else what
    their block headers allow at most."""
    src = bytes(src)
    return lib().mzd_content_bound(src, len(src))


def decode(src, cap=None, dict_id=0):
    """Whole-file decode of host bytes.  Returns (status, bytes)."""
    src = bytes(src)
    if cap is None:
        cs = content_size(src)
        cap = cs if cs < CONTENTSIZE_ERROR else max(64 * len(src), 1 << 20)
    res = decode_batch([src], [cap], [dict_id])
    return res[0]


def copy_decode(source, destination):
    """The reference's call shape: zstd::stream::copy_decode(source, destination)
    (src/main.rs:463-467).  `source` is a readable binary file object, `destination` a writable
    one.  Raises OSError(EFAULT) on any decode failure, like the `.map_err(|_| libc::EFAULT)`."""
    data = source.read()
    cs = content_size(data)
    if cs == CONTENTSIZE_ERROR:
        raise OSError(errno.EFAULT, "zstd decode failed")
    # frames without a content size: a guess first, then what the block headers allow at most (mzd_content_bound): two decodes at most
    bound = content_bound(data)
    cap = cs if cs != CONTENTSIZE_UNKNOWN else min(max(8 * len(data), 1 << 20), bound)
    rc, out = decode(data, cap)
    if rc == E_DSTSIZE and cs == CONTENTSIZE_UNKNOWN and bound < CONTENTSIZE_ERROR and bound > cap:
        rc, out = decode(data, bound)
    if rc != OK:
        raise OSError(errno.EFAULT, "zstd decode failed: %s" % strerror(rc))
    destination.write(out)


def decode_batch(srcs, caps, dict_ids=None):
    """Host-pointer batch (mzd_decode_batch): files are dealt round-robin over the initialised
    GPUs.  Returns [(status, bytes)]."""
    n = len(srcs)
    jobs = (Job * n)()
    keep = []
    for i in range(n):
        s = bytes(srcs[i])
        sb = C.create_string_buffer(s, len(s)) if len(s) else C.create_string_buffer(1)
        db = C.create_string_buffer(max(int(caps[i]), 1))
        keep.append((sb, db))
        jobs[i].src = C.cast(sb, C.c_void_p); jobs[i].src_len = len(s)
        jobs[i].dst = C.cast(db, C.c_void_p); jobs[i].dst_cap = int(caps[i])
        jobs[i].dict_id = dict_ids[i] if dict_ids else 0
    rc = lib().mzd_decode_batch(jobs, n)
    if rc != OK:
        raise MzdError(rc, "mzd_decode_batch")
    return [(jobs[i].status, keep[i][1].raw[:min(jobs[i].out_len, int(caps[i]))]) for i in range(n)]


def make_jobs(src_ptrs, src_lens, dst_ptrs, dst_caps, dict_ids=None):
    n = len(src_ptrs)
    jobs = (Job * n)()
    for i in range(n):
        jobs[i].src = int(src_ptrs[i]); jobs[i].src_len = int(src_lens[i])
        jobs[i].dst = int(dst_ptrs[i]); jobs[i].dst_cap = int(dst_caps[i])
        jobs[i].dict_id = int(dict_ids[i]) if dict_ids is not None else 0
    return jobs


def decode_batch_device(device, jobs, stream=None):
    """Device-pointer batch: src/dst already in HBM (src readable SRC_PADDING bytes past its end)."""
    rc = lib().mzd_decode_batch_device(device, jobs, len(jobs), stream)
    if rc != OK:
        raise MzdError(rc, "mzd_decode_batch_device")
    return [(j.status, j.out_len) for j in jobs]


class Batch:
    """prepare / launch / collect (measurement loops: the job table stays on the device)."""

    def __init__(self, device, jobs):
        self.jobs = jobs
        self.h = C.c_void_p()
        rc = lib().mzd_batch_prepare(device, jobs, len(jobs), C.byref(self.h))
        if rc != OK:
            raise MzdError(rc, "mzd_batch_prepare")

    def launch(self, stream=None, untimed=False):
        """untimed: MZD_LAUNCH_UNTIMED -- no start event in front of the launch (measurement loops time the whole loop themselves)."""
        if untimed and hasattr(lib(), "mzd_batch_launch_ex"):
            rc = lib().mzd_batch_launch_ex(self.h, stream, 1)
        else:
            rc = lib().mzd_batch_launch(self.h, stream)
        if rc != OK:
            raise MzdError(rc, "mzd_batch_launch")

    def collect(self, stream=None):
        rc = lib().mzd_batch_collect(self.h, self.jobs, stream)
        if rc != OK:
            raise MzdError(rc, "mzd_batch_collect")
        return [(j.status, j.out_len) for j in self.jobs]

    def free(self):
        if self.h:
            lib().mzd_batch_free(self.h)
            self.h = C.c_void_p()


def load_dict(data):
    data = bytes(data)
    did = C.c_uint32(0)
    rc = lib().mzd_load_dict(data, len(data), C.byref(did))
    if rc != OK:
        raise MzdError(rc, "mzd_load_dict")
    return did.value


def unload_dict(dict_id):
    rc = lib().mzd_unload_dict(dict_id)
    if rc != OK:
        raise MzdError(rc, "mzd_unload_dict")


def set_driver(driver):
    """Diagnostics: 0 automatic, 1 / 2 that general driver only (no small-file kernel), 4 / 5 block tasks with / without
    blocks resolved ahead of their predecessors."""
    rc = lib().mzd_debug_set_driver(int(driver))
    if rc != OK:
        raise MzdError(rc, "mzd_debug_set_driver")


def debug_counters(device=0):
    out = (C.c_uint32 * 8)()
    rc = lib().mzd_debug_counters(device, out)
    if rc != OK:
        raise MzdError(rc, "mzd_debug_counters")
    return list(out)


class HostBuffer:
    """Pinned host memory from mzd_host_alloc, viewed as a numpy uint8 array (`.a`)."""

    def __init__(self, n):
        import numpy as np
        self.n = int(n)
        self.ptr = lib().mzd_host_alloc(max(self.n, 1))
        if not self.ptr:
            raise MzdError(E_DEVICE, "mzd_host_alloc")
        self.a = np.ctypeslib.as_array((C.c_uint8 * max(self.n, 1)).from_address(self.ptr))[:self.n]

    def free(self):
        if self.ptr:
            self.a = None
            lib().mzd_host_free(self.ptr)
            self.ptr = None


def last_kernel_ms(device=0):
    ms = C.c_float(0)
    lib().mzd_last_kernel_ms(device, C.byref(ms))
    return ms.value


def last_kernel_name(device=0):
    """What the most recent device-pointer launch ran (dominant kernel first)."""
    return lib().mzd_last_kernel_name(device).decode()


def debug_last_block(device=0):
    """(literals bytes, [(ll, ml, off)]) of the last compressed block of job 0 of the last call."""
    lit = C.create_string_buffer(128 * 1024 + 64)
    seq = (C.c_uint32 * (4 * 43712))()
    nl, ns = C.c_size_t(0), C.c_size_t(0)
    rc = lib().mzd_debug_last_block(device, lit, len(lit), C.byref(nl), seq, 43712, C.byref(ns))
    if rc != OK:
        raise MzdError(rc, "mzd_debug_last_block")
    return lit.raw[:nl.value], [(seq[4 * i], seq[4 * i + 1], seq[4 * i + 2]) for i in range(ns.value)]


class ZstdFS:
    """Mirror of the read side of the reference's ZstdFS: decode-on-open, byte-range reads,
    handle sharing between opens of one inode (src/main.rs:451-513, src/file.rs:47-117).
    Errors are raised as OSError with the reference's errno."""

    def __init__(self):
        self._h = lib().mzd_fs_new()

    def open(self, ino, flags, zst_bytes, lazy=False):
        """-> (fh, real_size).  real_size is what open_wrapper stores in user.real_size (BE u64).
        lazy: nothing is decoded until a read needs it (mzd_fs_open_lazy)."""
        zst_bytes = bytes(zst_bytes)
        rs = C.c_uint64(0)
        fn = lib().mzd_fs_open_lazy if lazy else lib().mzd_fs_open
        fh = fn(self._h, ino, flags, zst_bytes, len(zst_bytes), C.byref(rs))
        if fh < 0:
            raise OSError(-fh, os.strerror(-fh))
        return fh, rs.value

    def read(self, fh, offset, size):
        buf = C.create_string_buffer(max(size, 1))
        n = lib().mzd_fs_read(self._h, fh, offset, size, buf)
        if n < 0:
            raise OSError(-n, os.strerror(-n))
        return buf.raw[:n]

    def release(self, fh):
        rc = lib().mzd_fs_release(self._h, fh)
        if rc < 0:
            raise OSError(-rc, os.strerror(-rc))

    @property
    def decode_count(self):
        return lib().mzd_fs_decode_count(self._h)

    @property
    def decoded_bytes(self):
        return lib().mzd_fs_decoded_bytes(self._h)

    def close(self):
        if self._h:
            lib().mzd_fs_free(self._h)
            self._h = None
