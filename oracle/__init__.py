"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings to oracle/liboracle.so:
  * ``decode`` / ``content_size`` / ``xxh64`` -- the plain-C restatement of the zstd frame
    decoder (zstd_oracle.c) that stands in for ``zstd::stream::copy_decode``
    (reference src/main.rs:463-467; arithmetic in libzstd 1.5.6, Cargo.lock:2371-2396).
  * ``LibZstd`` -- dlopen binding to the libzstd already on the machine (libzstd_dl.c); it
    pins the restatement and is the "reference" CPU baseline of bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

OK, E_CORRUPT, E_TRUNCATED, E_CHECKSUM, E_DSTSIZE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
E_BADMAGIC, E_DICT = -7, -8
UNKNOWN_SIZE = 2**64 - 1
MALFORMED = 2**64 - 2


def build(force=False):
    """Compile liboracle.so with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("zstd_oracle.c", "libzstd_dl.c", "zstd_oracle.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


class BlockInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "block_type", "lit_type", "lit_streams", "huf_max_bits", "n_lit", "n_seq",
        "ll_mode", "of_mode", "ml_mode", "regen")] + [("lit_hash", C.c_uint64), ("seq_hash", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Seq(C.Structure):
    _fields_ = [("ll", C.c_uint32), ("ml", C.c_uint32), ("off", C.c_uint32)]


class Trace(C.Structure):
    _fields_ = [("blocks", C.POINTER(BlockInfo)), ("cap", C.c_size_t), ("n", C.c_size_t),
                ("lit_dump", C.POINTER(C.c_uint8)), ("lit_cap", C.c_size_t), ("lit_n", C.c_size_t),
                ("seq_dump", C.POINTER(Seq)), ("seq_cap", C.c_size_t), ("seq_n", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ozs_decode.restype = C.c_int
        L.ozs_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                 C.c_char_p, C.c_size_t, C.POINTER(Trace)]
        L.ozs_content_size.restype = C.c_uint64
        L.ozs_content_size.argtypes = [C.c_char_p, C.c_size_t]
        L.ozs_xxh64.restype = C.c_uint64
        L.ozs_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.ozs_last_verdict_unpinned.restype = C.c_int
        L.ozs_last_verdict_inexact.restype = C.c_int
        L.ozs_last_verdict_lit_inexact.restype = C.c_int
        L.ozs_last_verdict_lit_lenient.restype = C.c_int
        L.ozs_last_verdict_lit_over.restype = C.c_int
        L.ozs_last_verdict_lit_through.restype = C.c_int
        L.ozs_set_huf_rule.argtypes = [C.c_int]
        L.ozs_strerror.restype = C.c_char_p
        L.ozs_strerror.argtypes = [C.c_int]
        _lib = L
    return _lib


def decode(src, cap=None, dictionary=None, want_trace=False, dump=False):
    """Whole-file decode.  Returns (status, bytes) or (status, bytes, [block dicts])."""
    L = lib()
    src = bytes(src)
    if cap is None:
        cs = L.ozs_content_size(src, len(src))
        cap = cs if cs < MALFORMED else max(64 * len(src), 1 << 20)
    buf = C.create_string_buffer(max(int(cap), 1))
    out_len = C.c_size_t(0)
    tr = None
    keep = []
    if want_trace:
        tr = Trace()
        arr = (BlockInfo * 4096)()
        tr.blocks = C.cast(arr, C.POINTER(BlockInfo)); tr.cap = 4096; tr.n = 0
        keep.append(arr)
        if dump:
            lit = (C.c_uint8 * (128 * 1024))(); seq = (Seq * 43691)()
            tr.lit_dump = C.cast(lit, C.POINTER(C.c_uint8)); tr.lit_cap = len(lit)
            tr.seq_dump = C.cast(seq, C.POINTER(Seq)); tr.seq_cap = len(seq)
            keep += [lit, seq]
    d = bytes(dictionary) if dictionary else None
    rc = L.ozs_decode(src, len(src), buf, int(cap), C.byref(out_len), d, len(d) if d else 0,
                      C.byref(tr) if tr is not None else None)
    out = buf.raw[:out_len.value]
    if want_trace:
        blocks = [keep[0][i].as_dict() for i in range(min(tr.n, 4096))]
        if dump:
            blocks_extra = {"lit": bytes(keep[1][:tr.lit_n]), "seq": [(s.ll, s.ml, s.off) for s in keep[2][:tr.seq_n]]}
            return rc, out, blocks, blocks_extra
        return rc, out, blocks
    return rc, out


def last_verdict_unpinned():
    """The last decode() rejected its input because a sequence bitstream ran out inside a block: libzstd rejects it too, with the
    class its bit container's leftovers lead to (zstd_oracle.h)."""
    return bool(lib().ozs_last_verdict_unpinned())


def last_verdict_inexact():
    """True when the last decode() failed because a block's sequence bitstream was not consumed exactly (its sequences all executed):
    libzstd older than 1.5.4 does not look at that and reports what comes next."""
    return bool(lib().ozs_last_verdict_inexact())


def last_verdict_lit_inexact():
    """True when the last decode() failed because a Huffman literal stream was not consumed exactly: corrupt by RFC 8878 and for
    libzstd 1.4; libzstd 1.5 decodes on and leaves it to the content checksum."""
    return bool(lib().ozs_last_verdict_lit_inexact())


def last_verdict_lit_lenient():
    """True when the last decode() ACCEPTED a literal stream that was not consumed exactly, as the reference's libzstd 1.5.x does in its
    fast loops (libzstd 1.4.x refuses such input)."""
    return bool(lib().ozs_last_verdict_lit_lenient())


def last_verdict_lit_through():
    """True when the last decode() ACCEPTED a literal stream that ran out and read on into the bytes in front of it (libzstd 1.5.x's fast
    loops are bounded by the section's first byte, not the stream's)."""
    return bool(lib().ozs_last_verdict_lit_through())


def last_verdict_lit_over():
    """True when the last decode() failed because a literal stream of a fast-loop section needed bits from below the section's first byte."""
    return bool(lib().ozs_last_verdict_lit_over())


def set_huf_rule(rule):
    """1 (default): literal streams judged as the reference's libzstd 1.5.x does; 0: RFC 8878 / libzstd 1.4.x (always exact)."""
    lib().ozs_set_huf_rule(int(rule))


def content_size(src):
    src = bytes(src)
    return lib().ozs_content_size(src, len(src))


def xxh64(data, seed=0):
    data = bytes(data)
    return lib().ozs_xxh64(data, len(data), seed)


def strerror(code):
    return lib().ozs_strerror(code).decode()


class LibZstd:
    """The system libzstd through dlopen (libzstd_dl.c).  ``LibZstd.available()`` is False when
    no libzstd shared object can be found (then fixtures alone pin the oracle)."""
    _ok = None

    @classmethod
    def available(cls, path=None):
        if cls._ok is None:
            L = lib()
            L.zref_open.argtypes = [C.c_char_p]
            L.zref_open.restype = C.c_int
            cls._ok = (L.zref_open(path.encode() if path else None) == 0)
            if cls._ok:
                L.zref_version.restype = C.c_char_p
                L.zref_bound.restype = C.c_size_t; L.zref_bound.argtypes = [C.c_size_t]
                L.zref_compress_simple.restype = C.c_long
                L.zref_compress_simple.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
                L.zref_compress.restype = C.c_long
                L.zref_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                            C.c_char_p, C.c_size_t]
                L.zref_compress_stream.restype = C.c_long
                L.zref_compress_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                                   C.c_size_t, C.c_int]
                for f in (L.zref_decompress, L.zref_decompress_stream8k):
                    f.restype = C.c_long
                    f.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
                L.zref_decompress_dict.restype = C.c_long
                L.zref_decompress_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
                L.zref_train_dict.restype = C.c_long
                L.zref_train_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_uint]
                L.zref_frame_content_size.restype = C.c_ulonglong
                L.zref_frame_content_size.argtypes = [C.c_char_p, C.c_size_t]
                L.zref_time_stream8k.restype = C.c_double
                L.zref_time_stream8k.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t,
                                                 C.POINTER(C.c_uint64)]
                L.zref_time_oneshot_mt.restype = C.c_double
                L.zref_time_oneshot_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint64)]
                L.zref_time_oneshot_mt_dict.restype = C.c_double
                L.zref_time_oneshot_mt_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        return cls._ok

    @staticmethod
    def _opened():
        """lib() once libzstd is open behind the shim (its entry points are NULL before zref_open)."""
        if not LibZstd.available():
            raise RuntimeError("no libzstd shared object on this machine")
        return lib()

    @staticmethod
    def version():
        return lib().zref_version().decode()

    @staticmethod
    def compress_simple(data, level=0):
        """zstd::bulk::compress(data, level) (reference tests/convert.rs:15-43)."""
        data = bytes(data); L = LibZstd._opened()
        cap = L.zref_bound(len(data)); buf = C.create_string_buffer(cap)
        r = L.zref_compress_simple(buf, cap, data, len(data), level)
        assert r >= 0
        return buf.raw[:r]

    @staticmethod
    def compress(data, level=3, checksum=True, content_size=True, window_log=0, dictionary=None):
        """The reference writer's settings by default (src/main.rs:781-791)."""
        data = bytes(data); L = LibZstd._opened()
        cap = L.zref_bound(len(data)) + 64; buf = C.create_string_buffer(cap)
        flags = (1 if checksum else 0) | (0 if content_size else 2)
        d = bytes(dictionary) if dictionary else None
        r = L.zref_compress(buf, cap, data, len(data), level, flags, window_log, d, len(d) if d else 0)
        assert r >= 0
        return buf.raw[:r]

    @staticmethod
    def compress_stream(data, level=3, checksum=True, chunk=0, flush_each=False):
        data = bytes(data); L = LibZstd._opened()
        cap = L.zref_bound(len(data)) + 64 + 16 * (len(data) // max(chunk, 1) + 1 if chunk else 1)
        buf = C.create_string_buffer(cap)
        r = L.zref_compress_stream(buf, cap, data, len(data), level, 1 if checksum else 0, chunk, 1 if flush_each else 0)
        assert r >= 0
        return buf.raw[:r]

    @staticmethod
    def decompress(src, cap, stream8k=False, dictionary=None):
        """Returns bytes, or a negative int (libzstd error code) on failure."""
        src = bytes(src); L = LibZstd._opened()
        buf = C.create_string_buffer(max(int(cap), 1))
        if dictionary is not None:
            d = bytes(dictionary)
            r = L.zref_decompress_dict(buf, int(cap), src, len(src), d, len(d))
        elif stream8k:
            r = L.zref_decompress_stream8k(buf, int(cap), src, len(src))
        else:
            r = L.zref_decompress(buf, int(cap), src, len(src))
        return buf.raw[:r] if r >= 0 else int(r)

    @staticmethod
    def train_dict(samples, cap):
        L = LibZstd._opened()
        blob = b"".join(samples)
        sizes = (C.c_size_t * len(samples))(*[len(s) for s in samples])
        buf = C.create_string_buffer(cap)
        r = L.zref_train_dict(buf, cap, blob, sizes, len(samples))
        assert r >= 0, "ZDICT_trainFromBuffer failed"
        return buf.raw[:r]
