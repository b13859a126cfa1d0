"""corpus -- deterministic synthetic corpora for tests and bench.py.

Produces the INPUT side of the path: files as the reference's writer would have left them in
``data_dir`` (one zstd frame per file: level, content checksum, pledged size -- reference
src/main.rs:781-791), compressed by the libzstd already on the machine.  Nothing here decodes.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcorpus.so")

KINDS = {"json": 0, "text": 1, "markup": 2, "int32": 3, "dna": 4, "xray": 5, "random": 6, "repeats": 7}
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "corpus_gen.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, src, "-ldl", "-lpthread"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.corpus_gen.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
        L.corpus_gen.restype = None
        L.corpus_open_zstd.argtypes = [C.c_char_p]
        L.corpus_open_zstd.restype = C.c_int
        L.corpus_zstd_version.restype = C.c_char_p
        L.corpus_bound.argtypes = [C.c_size_t]
        L.corpus_bound.restype = C.c_size_t
        L.corpus_build.restype = C.c_int
        L.corpus_build.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.corpus_build_dict.restype = C.c_int
        L.corpus_build_dict.argtypes = L.corpus_build.argtypes + [C.c_void_p, C.c_size_t]
        L.corpus_train_dict.restype = C.c_long
        L.corpus_train_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint]
        _lib = L
    return _lib


def have_zstd():
    return lib().corpus_open_zstd(None) == 0


def zstd_version():
    return lib().corpus_zstd_version().decode() if have_zstd() else ""


def gen(kind, cfg_id, index, size):
    """Raw bytes of file `index` of config `cfg_id`."""
    k = KINDS[kind] if isinstance(kind, str) else int(kind)
    buf = np.empty(size, dtype=np.uint8)
    lib().corpus_gen(k, cfg_id, index, buf.ctypes.data, size)
    return buf.tobytes()


class Corpus:
    """raw: uint8 array of all files back to back; comp: packed compressed frames."""

    def __init__(self, raw, raw_offs, raw_sizes, comp, comp_offs, comp_sizes):
        self.raw, self.raw_offs, self.raw_sizes = raw, raw_offs, raw_sizes
        self.comp, self.comp_offs, self.comp_sizes = comp, comp_offs, comp_sizes

    @property
    def nfiles(self):
        return len(self.raw_sizes)

    def raw_file(self, i):
        o, n = int(self.raw_offs[i]), int(self.raw_sizes[i])
        return self.raw[o:o + n]

    def comp_file(self, i):
        o, n = int(self.comp_offs[i]), int(self.comp_sizes[i])
        return self.comp[o:o + n]


def build_corpus_from_chunks(chunks, level=3, checksum=True, nthreads=None, align=16):
    """Compress given byte strings (real data, e.g. 128 KiB pieces of the Silesia files: SURVEY.md 8d config 3), each as ONE frame written
    like the reference's writer.  Same Corpus layout as build_corpus."""
    return build_corpus(-1, 0, [len(c) for c in chunks], level=level, checksum=checksum, nthreads=nthreads, align=align, _raw_chunks=chunks)


def build_corpus(kind, cfg_id, sizes, first_index=0, stride=1, level=3, checksum=True, kind_mod=0, nthreads=None, align=16, dictionary=None, _raw_chunks=None):
    """Generate + compress len(sizes) files with indices first_index + i*stride.  kind_mod>0 cycles kinds kind..kind+kind_mod-1 by index
    (the Silesia-proxy mix).  Compressed frames are packed at `align`-byte boundaries."""
    L = lib()
    if L.corpus_open_zstd(None) != 0:
        raise RuntimeError("no libzstd shared object found: cannot compress a corpus on this machine")
    k = KINDS[kind] if isinstance(kind, str) else int(kind)
    sizes = np.asarray(sizes, dtype=np.uint64)
    n = len(sizes)
    raw_offs = np.zeros(n, dtype=np.uint64)
    if n > 1:
        raw_offs[1:] = np.cumsum((sizes[:-1] + np.uint64(15)) & ~np.uint64(15))
    raw_total = int(raw_offs[-1] + sizes[-1]) if n else 0
    raw = np.zeros(raw_total + 64, dtype=np.uint8)
    if _raw_chunks is not None:
        for o, c in zip(raw_offs, _raw_chunks):
            raw[int(o):int(o) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    bounds = np.array([L.corpus_bound(int(s)) for s in np.unique(sizes)], dtype=np.uint64)
    bmap = dict(zip([int(s) for s in np.unique(sizes)], [int(b) for b in bounds]))
    slot = np.array([bmap[int(s)] for s in sizes], dtype=np.uint64)
    tmp_offs = np.zeros(n, dtype=np.uint64)
    if n > 1:
        tmp_offs[1:] = np.cumsum(slot[:-1])
    tmp = np.empty(int(tmp_offs[-1] + slot[-1]) if n else 0, dtype=np.uint8)
    comp_sizes = np.zeros(n, dtype=np.uint64)
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 32)
    dbuf = np.frombuffer(bytes(dictionary), dtype=np.uint8) if dictionary else None
    rc = L.corpus_build_dict(k, kind_mod, cfg_id, first_index, stride, n, raw_offs.ctypes.data, sizes.ctypes.data, raw.ctypes.data,
                             tmp.ctypes.data, tmp_offs.ctypes.data, comp_sizes.ctypes.data, level, 1 if checksum else 0, nthreads,
                             dbuf.ctypes.data if dbuf is not None else None, len(dbuf) if dbuf is not None else 0)
    if rc != 0:
        raise RuntimeError("corpus_build failed: %d" % rc)
    a = np.uint64(align - 1)
    comp_offs = np.zeros(n, dtype=np.uint64)
    if n > 1:
        comp_offs[1:] = np.cumsum((comp_sizes[:-1] + a) & ~a)
    comp = np.zeros(int(comp_offs[-1] + comp_sizes[-1]) + 64 if n else 64, dtype=np.uint8)
    for i in range(n):
        o, t, c = int(comp_offs[i]), int(tmp_offs[i]), int(comp_sizes[i])
        comp[o:o + c] = tmp[t:t + c]
    return Corpus(raw, raw_offs, sizes, comp, comp_offs, comp_sizes)


def train_dict(kind, cfg_id, sizes, cap=112640, first_index=0):
    """ZDICT_trainFromBuffer on the files first_index .. of a config (the first 4 000 records, 110 KiB cap)."""
    L = lib()
    if L.corpus_open_zstd(None) != 0:
        raise RuntimeError("no libzstd shared object found")
    blob = b"".join(gen(kind, cfg_id, first_index + i, int(s)) for i, s in enumerate(sizes))
    szs = (C.c_size_t * len(sizes))(*[int(s) for s in sizes])
    buf = C.create_string_buffer(cap)
    r = L.corpus_train_dict(buf, cap, blob, szs, len(sizes))
    if r < 0:
        raise RuntimeError("ZDICT_trainFromBuffer failed (%d)" % r)
    return buf.raw[:r]
