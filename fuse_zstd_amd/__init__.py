"""fuse_zstd_amd -- MI355X-native zstd frame decode behind fuse-zstd's open/read path.

Python-side view of the C ABI in include/mzd.h (libmzd.so: hand-written HIP kernels for gfx950 +
the C++ host runtime).  The decode entry points stand in for the reference's one codec call,
``zstd::stream::copy_decode`` (reference src/main.rs:463-467); ``ZstdFS`` mirrors the caller's
side (open_wrapper / read_wrapper / release_wrapper, src/main.rs:451-513, 595-599, and the
handle table src/file.rs).  There is no CPU fallback: without the built library or without a GPU
every call raises / returns MZD_E_DEVICE.
"""
from .api import (  # noqa: F401
    MzdError, Batch, ZstdFS, build, lib, init, shutdown, device_count, content_size, content_bound, copy_decode,
    decode, decode_batch, decode_batch_device, load_dict, unload_dict, set_driver, debug_counters, HostBuffer, last_kernel_ms, last_kernel_name, debug_last_block, strerror,
    OK, E_CORRUPT, E_TRUNCATED, E_CHECKSUM, E_DSTSIZE, E_UNSUPPORTED, E_DEVICE, E_BADMAGIC, E_DICT, E_PARAM,
    SRC_PADDING, CONTENTSIZE_UNKNOWN, CONTENTSIZE_ERROR, EXPORTS,
)
