"""The product's HOST code under sanitizers (error boundary reference src/main.rs:467, src/errors.rs:4-10).
Everything here runs on the CPU -- there is no GPU sanitizer on this pool, and the host side is where untrusted bytes are parsed
by ordinary C++: mzd_content_size, the lazy open's index and its synthetic-frame builder, the open / read / release mirror, the
daemon's request parser and its batcher.
  * `make -C fuse_zstd_amd/csrc asan`: libmzd's sources + tests/native/host_san.cpp with AddressSanitizer + UBSan on the host side;
  * `make -C fuse_zstd_amd/csrc tsan`: the daemon with ThreadSanitizer.
"""
import ctypes as C
import errno
import os
import socket
import struct
import subprocess
import tempfile
import threading

import numpy as np
import pytest

import fuse_zstd_amd as mzd
import oracle
from tests import golden_util
from tests import test_fuse_daemon as fd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fuse_zstd_amd", "csrc")


def test_host_parsers_under_asan_and_ubsan():
    """Every golden vector (positives and negatives), truncations of each (all of them for vectors up to 600 bytes, 64 for longer
    ones) and hostile headers built by the driver -- content sizes of 2^60 and 2^64 - 1, a block size past the input, reserved
    bits, skippable frames whose size wraps, a million empty frames, a frame of 200 000 one-byte blocks -- through
    mzd_content_size, the lazy index + synthetic frames (each must parse again), mzd_fs_open / open_lazy / read / release and
    mzd_decode's entry checks, every input in a heap block of exactly its size."""
    subprocess.check_call(["make", "-C", CSRC, "-s", "asan"])
    files = sorted(os.path.join(ROOT, "tests", "golden", "frames", f) for f in os.listdir(os.path.join(ROOT, "tests", "golden", "frames")) if f.endswith(".zst"))
    assert len(files) > 100
    env = dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    out = subprocess.run([os.path.join(ROOT, "fuse_zstd_amd", "mzd_host_san")] + files, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "no sanitizer report" in out.stdout and "Sanitizer" not in out.stderr and "runtime error" not in out.stderr


def test_hostile_content_sizes_are_refused_before_anything_is_allocated():
    """A header may promise any content size; what n input bytes can regenerate is bounded (a block regenerates <= 128 KiB and
    costs >= 4 bytes).  mzd_fs_open / mzd_fs_open_lazy answer EFAULT for a promise beyond that -- no 2^60-byte vector::resize."""
    L = mzd.lib()
    L.mzd_fs_new.restype = C.c_void_p
    fs = L.mzd_fs_new()
    for cs in (1 << 60, (1 << 64) - 1, 1 << 41, 1 << 36):
        frame = b"\x28\xb5\x2f\xfd" + bytes([0xC0 | 0x20]) + struct.pack("<Q", cs) + b"\x29\x00\x00" + b"hello"
        rs = C.c_uint64(0)
        assert L.mzd_fs_open(C.c_void_p(fs), 5, 0, frame, len(frame), C.byref(rs)) == -errno.EFAULT
        assert L.mzd_fs_open_lazy(C.c_void_p(fs), 6, 0, frame, len(frame), C.byref(rs)) == -errno.EFAULT
    L.mzd_fs_free(C.c_void_p(fs))


def test_synthetic_prefix_frames_decode_to_prefixes():
    """The lazy read's synthetic frame (the first k blocks of a frame under a header without content size and checksum) is a
    valid frame whose output is the first k blocks' output: checked with the oracle on the multi-block golden vectors, for
    every k, through the host-side hook (no GPU)."""
    L = mzd.lib()
    L.mzd_debug_lazy_plan.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    checked = 0
    for v in golden_util.load_manifest():
        if not v.ok or v.dict is not None or v.name not in ("json_1m", "window_3m_l1", "nofcs_stream_300k", "multi_frame_skippable", "json_128k"):
            continue
        rc, want = oracle.decode(v.comp, cap=v.out_len)
        assert rc == 0
        total, nb, sl = C.c_uint64(0), C.c_uint32(0), C.c_size_t(0)
        buf = (C.c_uint8 * (len(v.comp) + 64))()
        nframes = L.mzd_debug_lazy_plan(v.comp, len(v.comp), 0, 1, buf, len(buf), C.byref(sl), C.byref(total), C.byref(nb))
        if v.name == "nofcs_stream_300k":
            assert nframes == 0  # no content size: not seekable, opened eagerly
            continue
        assert nframes >= 1 and total.value == v.out_len
        for k in range(1, nb.value + 1):
            L.mzd_debug_lazy_plan(v.comp, len(v.comp), 0, k, buf, len(buf), C.byref(sl), C.byref(total), C.byref(nb))
            rc, got = oracle.decode(bytes(buf[:sl.value]), cap=v.out_len)
            assert rc == 0 and got == want[:len(got)] and len(got) > 0, (v.name, k, rc)
            if k > 1:
                assert len(got) > prev
            prev = len(got)
            checked += 1
    assert checked >= 8


class _Raw(fd.FakeKernel):
    """The fake kernel with another daemon binary and pipelined (unanswered) sends."""
    def __init__(self, binary, data_dir, threads):
        self.k, d = socket.socketpair(socket.AF_UNIX, socket.SOCK_SEQPACKET)
        self.proc = subprocess.Popen([binary, "--data-dir", data_dir, "--fd", str(d.fileno()), "--threads", str(threads), "--batch-us", "500"],
                                     pass_fds=[d.fileno()], stderr=subprocess.PIPE, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1"))
        d.close()
        self.k.settimeout(120)
        self.unique = 0
        err, body = self.call(fd.INIT, 0, struct.pack("<IIII", 7, 31, 1 << 17, 0))
        assert err == 0


def test_daemon_batcher_and_request_parser_under_tsan():
    """Eight session threads of the daemon under ThreadSanitizer: a storm of pipelined LOOKUP / GETATTR / OPEN / READ / RELEASE /
    READDIR requests from the fake kernel -- opens of the same and of different inodes at once (here, without a GPU, every batch
    ends in EFAULT: the batcher's queue, its condition variables and the handle table are what runs) -- interleaved with
    malformed requests (short headers, lengths that lie, unknown opcodes, names without a terminator).  No data race, no lock-order
    inversion, every well-formed request answered, the daemon exits cleanly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("meant for the CPU build box (on a GPU box the gpu tests drive the daemon)")
    mzd.build()
    subprocess.check_call(["make", "-C", CSRC, "-s", "tsan"])
    binary = os.path.join(ROOT, "fuse_zstd_amd", "mzd_fused_tsan")
    probe = subprocess.run([binary], capture_output=True, text=True)
    if "unexpected memory mapping" in probe.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    d = tempfile.mkdtemp(prefix="mzd_tsan_")
    vecs = {v.name: v for v in golden_util.load_manifest()}
    names = ["json_4k", "json_128k", "ref_writer_01", "multi_frame_skippable", "bad_checksum", "proxy_text_128k"]
    for nme in names:
        with open(os.path.join(d, nme + ".zst"), "wb") as f:
            f.write(vecs[nme].comp)
    k = _Raw(binary, d, threads=8)
    inos = []
    for nme in names:
        err, ino, attr = k.lookup(1, nme)
        assert err == 0
        inos.append(ino)
    rng = np.random.RandomState(3)
    pending = {}
    answered = 0

    def drain(upto):
        nonlocal answered
        while len(pending) > upto:
            unique, err, body = k.recv()
            op = pending.pop(unique)
            answered += 1
            if op == fd.OPEN:
                assert err in (errno.EFAULT, 0), err
    for it in range(1500):
        r = rng.randint(0, 10)
        ino = inos[rng.randint(0, len(inos))]
        if r < 5:
            pending[k.send(fd.OPEN, ino, struct.pack("<II", os.O_RDONLY, 0))] = fd.OPEN
        elif r == 5:
            pending[k.send(fd.GETATTR, ino, struct.pack("<IIQ", 0, 0, 0))] = fd.GETATTR
        elif r == 6:
            pending[k.send(fd.LOOKUP, 1, names[rng.randint(0, len(names))].encode() + b"\0")] = fd.LOOKUP
        elif r == 7:
            body = struct.pack("<QQIIQII", 999, 0, 4096, 0, 0, 0, 0)
            pending[k.send(fd.READ, ino, body[:rng.choice([40, 24, 20, 7])])] = fd.READ  # an unknown handle: ENOENT; a body cut short: EINVAL
        elif r == 8:
            pending[k.send(fd.RELEASE, ino, struct.pack("<QIIQ", 999, 0, 0, 0))] = fd.RELEASE
        else:  # malformed: answered with an error or dropped, never a crash
            kind = rng.randint(0, 4)
            if kind == 0:
                k.k.send(b"\x01\x02\x03")                                                                     # shorter than a header: no answer
            elif kind == 1:
                k.unique += 1
                k.k.send(struct.pack("<IIQQIIII", 4000, fd.LOOKUP, k.unique, 1, 0, 0, 0, 0) + b"abc")       # a length that lies: what arrived counts
                pending[k.unique] = fd.LOOKUP
            elif kind == 2:
                pending[k.send(9999, ino, b"x" * 17)] = 9999                                               # unknown opcode: ENOSYS
            else:
                pending[k.send(fd.LOOKUP, 1, b"no-terminator")] = fd.LOOKUP
        if len(pending) > 48:
            drain(16)
    drain(0)  # every request that had a header is answered
    assert answered > 1000
    rc, log = k.close()
    assert "ThreadSanitizer" not in log, log[-6000:]
    assert rc == 0, log[-2000:]
