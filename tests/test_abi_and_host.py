"""CPU suite: the C-ABI library loads and exports every symbol include/mzd.h declares; host-side
logic that needs no GPU (frame-header walk, error strings, argument checks, no-GPU behaviour)."""
import ctypes as C
import os
import re

import pytest

import fuse_zstd_amd as mzd
import oracle
from tests import golden_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VECS = golden_util.load_manifest()


@pytest.fixture(scope="module")
def lib():
    mzd.build()
    return mzd.lib()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "mzd.h")).read()
    declared = set(re.findall(r"\b(mzd_[a-z0-9_]+)\s*\(", header))
    declared -= {"mzd_job", "mzd_batch", "mzd_fs"}
    assert declared == set(mzd.EXPORTS), declared ^ set(mzd.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_header_cites_the_reference_boundary():
    header = open(os.path.join(ROOT, "include", "mzd.h")).read()
    for cite in ("src/main.rs:463-467", "src/main.rs:451-493", "src/file.rs", "src/main.rs:467"):
        assert cite in header or cite.replace("src/main.rs:451-493", ":451-493") in header


@pytest.mark.parametrize("v", [v for v in VECS if v.ok and v.dict is None], ids=lambda v: v.name)
def test_content_size_matches_oracle(lib, v):
    got = mzd.content_size(v.comp)
    want = oracle.content_size(v.comp)
    assert got == want
    assert got in (v.out_len, mzd.CONTENTSIZE_UNKNOWN)


def test_content_size_malformed(lib):
    assert mzd.content_size(b"\x01\x02\x03\x04\x05\x06") == mzd.CONTENTSIZE_ERROR
    assert mzd.content_size(b"\x28\xb5\x2f") == mzd.CONTENTSIZE_ERROR
    assert mzd.content_size(b"") == 0
    good = next(x for x in VECS if x.name == "json_4k").comp
    assert mzd.content_size(good[:-5]) == mzd.CONTENTSIZE_ERROR  # truncated inside the frame


@pytest.mark.parametrize("v", [v for v in VECS if v.ok and v.dict is None], ids=lambda v: v.name)
def test_content_bound_is_the_size_where_stated_and_never_below_it(lib, v):
    """mzd_content_bound: SURVEY.md 8(b) "Ownership" -- the capacity a too-small destination is told to retry with."""
    bound, size = mzd.content_bound(v.comp), mzd.content_size(v.comp)
    assert bound < mzd.CONTENTSIZE_ERROR and bound >= v.out_len
    if size != mzd.CONTENTSIZE_UNKNOWN:
        assert bound == size == v.out_len
    else:  # what the block headers allow at most: 128 KiB per compressed block, the stated size of raw / RLE blocks
        rc, out, blocks = oracle.decode(v.comp, cap=v.out_len, want_trace=True)
        assert rc == 0 and bound <= sum(b["regen"] if b["block_type"] < 2 else 131072 for b in blocks)


def test_content_bound_malformed(lib):
    assert mzd.content_bound(b"\x01\x02\x03\x04\x05\x06") == mzd.CONTENTSIZE_ERROR
    assert mzd.content_bound(b"") == 0
    good = next(x for x in VECS if x.name == "nofcs_stream_300k").comp
    assert mzd.content_bound(good[:-5]) == mzd.CONTENTSIZE_ERROR


def test_error_strings_and_codes(lib):
    for code in range(0, -10, -1):
        assert mzd.strerror(code) and mzd.strerror(code) != "unknown error"
    # same numbering as the oracle's classes (tests compare them directly)
    assert (mzd.E_CORRUPT, mzd.E_TRUNCATED, mzd.E_CHECKSUM, mzd.E_DSTSIZE, mzd.E_UNSUPPORTED, mzd.E_BADMAGIC, mzd.E_DICT) == \
           (oracle.E_CORRUPT, oracle.E_TRUNCATED, oracle.E_CHECKSUM, oracle.E_DSTSIZE, oracle.E_UNSUPPORTED, oracle.E_BADMAGIC, oracle.E_DICT)


def _has_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_gpu_fails_loudly_never_falls_back(lib):
    """Without a GPU the product path must refuse, not decode on the CPU."""
    with pytest.raises(mzd.MzdError) as e:
        mzd.init()
    assert e.value.code == mzd.E_DEVICE
    assert mzd.device_count() == 0
    good = next(x for x in VECS if x.name == "json_4k")
    buf = C.create_string_buffer(good.out_len)
    n = C.c_size_t(0)
    assert lib.mzd_decode(good.comp, len(good.comp), buf, good.out_len, C.byref(n)) == mzd.E_DEVICE
    fs = mzd.ZstdFS()
    with pytest.raises(OSError):  # open() maps every decode failure to EFAULT (reference src/main.rs:467)
        fs.open(1, 0, good.comp)
    with pytest.raises(OSError):  # the lazy open has nothing to decode yet, but promises reads it could never serve
        fs.open(2, 0, good.comp, lazy=True)
    fs.close()


def test_product_never_links_the_oracle():
    """The product library must not reference anything under oracle/ (it is test infrastructure)."""
    import subprocess
    so = os.path.join(ROOT, "fuse_zstd_amd", "libmzd.so")
    syms = subprocess.check_output(["nm", "-D", so]).decode()
    assert "ozs_" not in syms and "zref_" not in syms
    deps = subprocess.check_output(["ldd", so]).decode()
    assert "liboracle" not in deps and "libzstd" not in deps
    for f in ("mzd_host.cpp", "mzd_kernels.hip", "mzd_device.h"):
        text = open(os.path.join(ROOT, "fuse_zstd_amd", "csrc", f)).read()
        assert "oracle/" not in text and "zstd_oracle" not in text
    for f in ("api.py", "__init__.py"):
        text = open(os.path.join(ROOT, "fuse_zstd_amd", f)).read()
        assert "import oracle" not in text and "from oracle" not in text


def test_init_does_not_touch_the_environment():
    """The library leaves the process environment alone (round 2's mzd_init exported GPU_MAX_HW_QUEUES: a side effect on the host
    application that only worked if HIP had not started).  Checked in a fresh process through ctypes, without the Python wrapper:
    mzd_init and mzd_init_ex -- which fail with MZD_E_DEVICE here, there is no GPU -- leave os.environ as it was."""
    import subprocess, sys
    code = r"""
import ctypes as C, os, sys
before = dict(os.environ)
L = C.CDLL(sys.argv[1])
class Config(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("device_ids", C.POINTER(C.c_int)), ("n_devices", C.c_int), ("max_workgroups", C.c_uint32),
                ("small_scratch_bytes", C.c_size_t), ("resolve_ahead", C.c_int)]
rc1 = L.mzd_init(None, 0)
cfg = Config(C.sizeof(Config), None, 0, 128, 64 << 20, 0)
rc2 = L.mzd_init_ex(C.byref(cfg))
bad = Config(4, None, 0, 0, 0, 0)
rc3 = L.mzd_init_ex(C.byref(bad))
# the C runtime's view, not Python's cached os.environ
libc = C.CDLL(None); libc.getenv.restype = C.c_char_p
q = libc.getenv(b"GPU_MAX_HW_QUEUES")
print(rc1, rc2, rc3, q, dict(os.environ) == before)
"""
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code, mzd.api._SO], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    rc1, rc2, rc3, q, same = out.stdout.split()
    import torch
    if not torch.cuda.is_available():
        assert int(rc1) == mzd.E_DEVICE and int(rc2) == mzd.E_DEVICE
    assert int(rc3) == mzd.E_PARAM      # a struct too short to hold the device list
    assert q == "None" and same == "True"
