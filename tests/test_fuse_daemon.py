"""The read-side FUSE daemon (fuse_zstd_amd/csrc/mzd_fused.cpp) driven over the raw FUSE protocol.

No mount is needed (and none is possible on the GPU box: no /dev/fuse there): the test plays the kernel's part over
a SOCK_SEQPACKET socketpair handed to the daemon with --fd.  What is checked is what the reference's read side shows
(src/main.rs): `name.zst` appears as `name`, other regular files are hidden, st_size is user.real_size (0 before the
first open), open decodes the whole file on the GPU and read slices it, a second open shares the bytes, every decode
failure is EFAULT, everything that writes is refused.  CPU part: namespace + "no GPU fails loudly"; GPU part: bytes.
"""
import errno
import os
import shutil
import socket
import struct
import subprocess
import tempfile

import pytest

from tests import golden_util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(ROOT, "fuse_zstd_amd", "mzd_fused")

LOOKUP, FORGET, GETATTR, MKDIR, UNLINK, OPEN, READ, WRITE, STATFS, RELEASE, INIT, OPENDIR, READDIR, RELEASEDIR, DESTROY = 1, 2, 3, 9, 10, 14, 15, 16, 17, 18, 26, 27, 28, 29, 38
ATTR = struct.Struct("<QQQQQQIIIIIIIIII")  # fuse_attr, 88 bytes


def build_daemon():
    import fuse_zstd_amd
    fuse_zstd_amd.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "fuse_zstd_amd", "csrc"), "-s", "fused"])
    return BIN


class FakeKernel:
    def __init__(self, data_dir, threads=4, batch_us=2000, extra=()):
        build_daemon()
        self.k, d = socket.socketpair(socket.AF_UNIX, socket.SOCK_SEQPACKET)
        self.proc = subprocess.Popen([BIN, "--data-dir", data_dir, "--fd", str(d.fileno()), "--threads", str(threads), "--batch-us", str(batch_us)] + list(extra),
                                     pass_fds=[d.fileno()], stderr=subprocess.PIPE)
        d.close()
        self.k.settimeout(60)
        self.unique = 0
        err, body = self.call(INIT, 0, struct.pack("<IIII", 7, 31, 1 << 17, 0))
        assert err == 0 and struct.unpack_from("<II", body) == (7, 31)

    def send(self, opcode, nodeid, body=b""):
        self.unique += 1
        self.k.send(struct.pack("<IIQQIIII", 40 + len(body), opcode, self.unique, nodeid, 0, 0, 1234, 0) + body)
        return self.unique

    def recv(self):
        msg = self.k.recv(1 << 21)
        ln, err, unique = struct.unpack_from("<IiQ", msg)
        assert ln == len(msg)
        return unique, -err, msg[16:]

    def call(self, opcode, nodeid, body=b""):
        u = self.send(opcode, nodeid, body)
        unique, err, out = self.recv()
        assert unique == u
        return err, out

    def lookup(self, parent, name):
        err, out = self.call(LOOKUP, parent, name.encode() + b"\0")
        if err:
            return err, None, None
        nodeid = struct.unpack_from("<Q", out)[0]
        return 0, nodeid, ATTR.unpack_from(out, 40)

    def getattr(self, ino):
        err, out = self.call(GETATTR, ino, struct.pack("<IIQ", 0, 0, 0))
        return err, (ATTR.unpack_from(out, 16) if not err else None)

    def readdir(self, ino):
        err, _ = self.call(OPENDIR, ino, struct.pack("<II", 0, 0))
        if err:
            return err, None
        names, off = [], 0
        while True:
            err, out = self.call(READDIR, ino, struct.pack("<QQIIQII", 0, off, 4096, 0, 0, 0, 0))
            assert err == 0
            if not out:
                break
            p = 0
            while p < len(out):
                d_ino, d_off, namelen, typ = struct.unpack_from("<QQII", out, p)
                names.append((out[p + 24:p + 24 + namelen].decode(), typ, d_ino))
                off = d_off
                p += (24 + namelen + 7) & ~7
        self.call(RELEASEDIR, ino, struct.pack("<QIIQ", 0, 0, 0, 0))
        return 0, names

    def open(self, ino, flags=os.O_RDONLY):
        err, out = self.call(OPEN, ino, struct.pack("<II", flags, 0))
        return err, (struct.unpack_from("<Q", out)[0] if not err else None)

    def read(self, ino, fh, offset, size):
        return self.call(READ, ino, struct.pack("<QQIIQII", fh, offset, size, 0, 0, 0, 0))

    def release(self, ino, fh):
        return self.call(RELEASE, ino, struct.pack("<QIIQ", fh, 0, 0, 0))[0]

    def close(self):
        """DESTROY, then the daemon's exit code and what it wrote to stderr."""
        try:
            self.call(DESTROY, 0)
        finally:
            self.k.close()
        try:
            _, err = self.proc.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            self.proc.kill()
            raise
        return self.proc.returncode, err.decode(errors="replace")


@pytest.fixture()
def data_dir():
    d = tempfile.mkdtemp(prefix="mzd_fused_")
    vecs = {v.name: v for v in golden_util.load_manifest()}
    for name in ("json_4k", "json_128k", "json_1m", "ref_writer_01", "multi_frame_skippable"):
        with open(os.path.join(d, name + ".zst"), "wb") as f:
            f.write(vecs[name].comp)
    os.mkdir(os.path.join(d, "sub"))
    with open(os.path.join(d, "sub", "inner.zst"), "wb") as f:
        f.write(vecs["proxy_text_128k"].comp)
    with open(os.path.join(d, "plain.txt"), "wb") as f:  # not a .zst: hidden by the mount
        f.write(b"not compressed")
    with open(os.path.join(d, "broken.zst"), "wb") as f:
        f.write(vecs["bad_checksum"].comp)
    yield d, vecs
    shutil.rmtree(d, ignore_errors=True)


def test_namespace_and_refusals_without_decoding(data_dir):
    d, vecs = data_dir
    k = FakeKernel(d)
    err, names = k.readdir(1)
    assert err == 0
    shown = sorted(n for n, _, _ in names)
    assert shown == sorted(["json_4k", "json_128k", "json_1m", "ref_writer_01", "multi_frame_skippable", "sub", "broken"])  # plain.txt hidden, .zst stripped
    assert dict((n, t) for n, t, _ in names)["sub"] == 4 and dict((n, t) for n, t, _ in names)["json_4k"] == 8  # DT_DIR, DT_REG
    err, ino, attr = k.lookup(1, "json_4k")
    assert err == 0 and attr[0] == ino and (attr[9] & 0o170000) == 0o100000 and (attr[9] & 0o777) == 0o666
    assert ino == dict((n, i) for n, _, i in names)["json_4k"]  # readdir and lookup agree on the inode number
    assert k.lookup(1, "plain.txt")[0] == errno.ENOENT and k.lookup(1, "json_4k.zst")[0] == errno.ENOENT and k.lookup(1, "nope")[0] == errno.ENOENT
    err, sub, sattr = k.lookup(1, "sub")
    assert err == 0 and (sattr[9] & 0o170000) == 0o040000 and (sattr[9] & 0o777) == 0o777
    assert sorted(n for n, _, _ in k.readdir(sub)[1]) == ["inner"]
    assert k.lookup(sub, "inner")[0] == 0
    assert k.readdir(ino)[0] == errno.ENOTDIR
    err, a2 = k.getattr(ino)
    assert err == 0 and a2[0] == ino
    assert k.getattr(0xdeadbeef)[0] == errno.ENOENT
    # nothing that writes
    assert k.open(ino, os.O_WRONLY)[0] == errno.EROFS
    assert k.call(MKDIR, 1, struct.pack("<II", 0o755, 0) + b"x\0")[0] == errno.EROFS
    assert k.call(UNLINK, 1, b"json_4k\0")[0] == errno.EROFS
    assert k.call(WRITE, ino, struct.pack("<QQIIQII", 1, 0, 1, 0, 0, 0, 0) + b"x")[0] == errno.EROFS
    assert k.call(39, ino, b"")[0] == errno.ENOSYS  # FUSE_IOCTL: not implemented
    assert k.call(STATFS, 1)[0] == 0
    rc, log = k.close()
    assert rc == 0


def test_no_gpu_open_fails_loudly_never_decodes_on_the_cpu(data_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the decode path is covered by the gpu tests")
    d, vecs = data_dir
    k = FakeKernel(d)
    err, ino, attr = k.lookup(1, "json_4k")
    assert err == 0 and attr[1] == 0  # size unknown before the first successful open (reference: 0 without user.real_size)
    assert k.open(ino)[0] == errno.EFAULT
    assert k.getattr(ino)[1][1] == 0
    rc, log = k.close()
    assert rc == 0 and "no usable MI355X" in log and "0 files decoded" in log


@pytest.mark.gpu
def test_open_read_release_through_the_daemon(data_dir):
    d, vecs = data_dir
    k = FakeKernel(d)
    for name in ("json_4k", "json_128k", "json_1m", "ref_writer_01", "multi_frame_skippable"):
        want = vecs[name].expected()
        err, ino, attr = k.lookup(1, name)
        assert err == 0
        err, fh = k.open(ino)
        assert err == 0, name
        assert k.getattr(ino)[1][1] == len(want)  # st_size is published by the open
        got, off = b"", 0
        while True:  # sequential 128 KiB reads, as the kernel issues them
            err, chunk = k.read(ino, fh, off, 131072)
            assert err == 0
            if not chunk:
                break
            got += chunk; off += len(chunk)
        assert got == want, name
        assert k.read(ino, fh, max(len(want) - 10, 0), 100)[1] == want[max(len(want) - 10, 0):]  # short at the end
        assert k.read(ino, fh, len(want) + 5, 100) == (0, b"")
        err, fh2 = k.open(ino)  # second open of the same inode: shares the decoded bytes
        assert err == 0 and fh2 != fh and k.read(ino, fh2, 0, 64)[1] == want[:64]
        assert k.release(ino, fh) == 0
        assert k.read(ino, fh2, 1, 7)[1] == want[1:8]  # still valid: the other handle keeps the bytes
        assert k.release(ino, fh2) == 0
        assert k.read(ino, fh, 0, 1)[0] == errno.ENOENT
    err, sub, _ = k.lookup(1, "sub")
    err, ino, _ = k.lookup(sub, "inner")
    err, fh = k.open(ino)
    assert err == 0 and k.read(ino, fh, 0, 1 << 20)[1] == vecs["proxy_text_128k"].expected()
    err, bad, _ = k.lookup(1, "broken")
    assert k.open(bad)[0] == errno.EFAULT  # checksum mismatch: the reference maps every decode error to EFAULT
    rc, log = k.close()
    assert rc == 0 and "6 files decoded" in log, log  # five + inner: second opens decode nothing, the broken file is not counted


@pytest.mark.gpu
def test_concurrent_opens_are_decoded_in_batches():
    import corpus
    if not corpus.have_zstd():
        pytest.skip("no libzstd to build a corpus with")
    d = tempfile.mkdtemp(prefix="mzd_fused_many_")
    try:
        n = 96
        cp = corpus.build_corpus("json", 41, [20000 + 997 * i for i in range(n)])
        for i in range(n):
            with open(os.path.join(d, "f%03d.zst" % i), "wb") as f:
                f.write(cp.comp_file(i).tobytes())
        k = FakeKernel(d, threads=16, batch_us=20000)
        inos = [k.lookup(1, "f%03d" % i)[1] for i in range(n)]
        pend = {k.send(OPEN, ino, struct.pack("<II", os.O_RDONLY, 0)): i for i, ino in enumerate(inos)}  # all at once, like many readers
        fhs = {}
        for _ in range(n):
            unique, err, out = k.recv()
            assert err == 0
            fhs[pend.pop(unique)] = struct.unpack_from("<Q", out)[0]
        for i in range(n):
            assert k.read(inos[i], fhs[i], 0, 1 << 20)[1] == cp.raw_file(i).tobytes(), i
        rc, log = k.close()
        assert rc == 0
        files, batches, hits = [int(x) for x in log.split("mzd_fused:")[-1].replace(",", " ").split() if x.isdigit()]
        assert files == n and batches <= n // 4 and hits == 0, log  # sixteen session threads feed one batcher: many files per launch
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_real_mount_when_the_container_allows_it(data_dir):
    """Where /dev/fuse exists and mount(2) is permitted (the build container: root; not the GPU box) the daemon is
    mounted for real and the KERNEL is the client: names, sizes and refusals as the reference's mount shows them.
    Without a GPU an open fails with EFAULT ("Bad address") -- exactly what a decode failure looks like through the mount."""
    import time
    import torch
    if not os.path.exists("/dev/fuse") or os.geteuid() != 0:
        pytest.skip("no /dev/fuse or not root")
    d, vecs = data_dir
    build_daemon()
    mnt = tempfile.mkdtemp(prefix="mzd_mnt_")
    proc = subprocess.Popen([BIN, "--data-dir", d, "--mount", mnt, "--threads", "2"], stderr=subprocess.PIPE)
    try:
        for _ in range(50):
            if os.path.ismount(mnt) or proc.poll() is not None:
                break
            time.sleep(0.1)
        if not os.path.ismount(mnt):
            pytest.skip("mount(2) is not permitted here")
        assert sorted(os.listdir(mnt)) == sorted(["json_4k", "json_128k", "json_1m", "ref_writer_01", "multi_frame_skippable", "sub", "broken"])
        assert os.listdir(os.path.join(mnt, "sub")) == ["inner"]
        st = os.stat(os.path.join(mnt, "json_4k"))
        assert (st.st_mode & 0o777) == 0o666
        with pytest.raises(OSError) as ei:
            open(os.path.join(mnt, "nope"), "rb")
        assert ei.value.errno == errno.ENOENT
        with pytest.raises(OSError) as ei:
            open(os.path.join(mnt, "newfile"), "wb")
        assert ei.value.errno == errno.EROFS
        if torch.cuda.is_available():
            assert open(os.path.join(mnt, "json_128k"), "rb").read() == vecs["json_128k"].expected()
            assert os.stat(os.path.join(mnt, "json_128k")).st_size == len(vecs["json_128k"].expected())
        else:
            with pytest.raises(OSError) as ei:
                open(os.path.join(mnt, "json_4k"), "rb").read()
            assert ei.value.errno == errno.EFAULT
    finally:
        try:
            import ctypes
            ctypes.CDLL(None, use_errno=True).umount2(mnt.encode(), 2)  # MNT_DETACH
        except Exception:
            subprocess.call(["umount", "-l", mnt])
        try:
            proc.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()
        shutil.rmtree(mnt, ignore_errors=True)


@pytest.mark.gpu
def test_decode_ahead_serves_a_sequential_reader_from_few_launches():
    """One reader opening the files of a directory one after the other (tar, grep -r, one fio job): with --ahead N a miss
    decodes the next N files of the directory in the same launch, the following opens are served from that."""
    import corpus
    if not corpus.have_zstd():
        pytest.skip("no libzstd to build a corpus with")
    d = tempfile.mkdtemp(prefix="mzd_fused_seq_")
    try:
        n = 64
        cp = corpus.build_corpus("json", 43, [4096 + 131 * i for i in range(n)])
        for i in range(n):
            with open(os.path.join(d, "f%03d.zst" % i), "wb") as f:
                f.write(cp.comp_file(i).tobytes())
        k = FakeKernel(d, threads=2, batch_us=100, extra=["--ahead", "15"])
        for i in range(n):
            err, ino, _ = k.lookup(1, "f%03d" % i)
            err, fh = k.open(ino)
            assert err == 0
            assert k.read(ino, fh, 0, 1 << 20)[1] == cp.raw_file(i).tobytes(), i
            assert k.getattr(ino)[1][1] == int(cp.raw_sizes[i])
            assert k.release(ino, fh) == 0
        rc, log = k.close()
        files, batches, hits = [int(x) for x in log.split("mzd_fused:")[-1].replace(",", " ").split() if x.isdigit()]
        assert rc == 0 and files == n and batches == n // 16 and hits == n - n // 16, log
    finally:
        shutil.rmtree(d, ignore_errors=True)
