"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(include/mzd.h), against the oracle and the committed golden vectors -- byte-exact."""
import ctypes as C

import numpy as np
import pytest

import corpus
import fuse_zstd_amd as mzd
import oracle
from tests import golden_util

pytestmark = pytest.mark.gpu
VECS = golden_util.load_manifest()


@pytest.fixture(scope="module", autouse=True)
def gpu():
    mzd.build()
    mzd.init()  # raises MzdError(E_DEVICE) when there is no GPU: no silent fallback
    yield
    mzd.shutdown()


DRIVERS = ["auto", "1", "1w", "1p", "4", "5"]  # auto (mode 3): small files take the small-file kernel however few they are, the rest (and what it hands on) a general driver;
                                          # 1w: a workgroup per file, THREE wavefronts a workgroup and five workgroups a CU (mzd_debug_host_path 11 = 3: what the library
                                          # takes when a launch holds more files than the four-a-CU slots);
                                          # 1p: a workgroup per file with TWO files a workgroup -- one walking wavefront for both (mzd_debug_host_path 11 = 2);
                                          # 4 / 5: block tasks with / without blocks resolved ahead of their predecessors (mzd_k_resolve.h)


@pytest.fixture
def force_driver():
    """mzd_debug_set_driver for one test: 'auto' or one general driver alone ('1', '2', '4', '5'); 'auto:G:XG' also fixes the small-file
    kernel's shape (G files per wavefront through the entropy phases, XG of them executed at a time: mzd_debug_host_path 4 / 5)."""
    def set_(driver):
        parts = driver.split(":")
        if parts[0] in ("1p", "1w"):
            mzd.lib().mzd_debug_host_path(0, 11, 2 if parts[0] == "1p" else 3)
            parts[0] = "1"
        mzd.set_driver(3 if parts[0] == "auto" else int(parts[0]))
        if len(parts) >= 3:
            mzd.lib().mzd_debug_host_path(0, 4, int(parts[1]))
            mzd.lib().mzd_debug_host_path(0, 5, int(parts[2]))
        if len(parts) == 4:  # 'auto:8:4:2': with the helper wavefront (mzd_debug_host_path 9)
            mzd.lib().mzd_debug_host_path(0, 9, int(parts[3]))
    yield set_
    mzd.set_driver(0)
    mzd.lib().mzd_debug_host_path(0, 11, 0)
    mzd.lib().mzd_debug_host_path(0, 4, 0)
    mzd.lib().mzd_debug_host_path(0, 5, 0)
    mzd.lib().mzd_debug_host_path(0, 9, 0)


def test_golden_positive_batch():
    """Every positive vector without a dictionary, in ONE batch (one workgroup per file)."""
    vs = [v for v in VECS if v.ok and v.dict is None]
    res = mzd.decode_batch([v.comp for v in vs], [v.out_len for v in vs])
    bad = []
    for v, (st, out) in zip(vs, res):
        if st != 0 or out != v.expected():
            bad.append((v.name, st, len(out), v.out_len))
    assert not bad, bad


@pytest.mark.parametrize("v", [v for v in VECS if v.ok and v.dict is None], ids=lambda v: v.name)
def test_golden_positive_single(v):
    st, out = mzd.decode(v.comp, v.out_len)
    assert st == 0, mzd.strerror(st)
    assert out == v.expected()


@pytest.mark.parametrize("v", [v for v in VECS if not v.ok], ids=lambda v: v.name)
def test_golden_negative(v):
    st, _ = mzd.decode(v.comp, 1 << 22)
    assert st == v.oracle_class, (mzd.strerror(st), oracle.strerror(v.oracle_class))


@pytest.mark.parametrize("name", ["json_4k", "json_128k", "proxy_text_128k", "proxy_dna_300k", "hand_rle_lits_rle_tables",
                                  "hand_long_nbseq", "hand_direct_weights_4s", "json_1m", "zeros_128k", "rle_500k"])
def test_phase_intermediates_match_cpu_twin(name, force_driver):
    """Literal buffer (K2) and sequence triples (K4) of the last compressed block, as the block pipeline
    left them in its scratch, against the oracle's dump of the same block.  (The general drivers are forced; the small-file
    kernel's intermediates: test_small_file_kernel_intermediates_match_cpu_twin.)  zeros_128k / rle_500k: a match of 131 070 bytes --
    the plan's 8-byte entries spell a length of 0xFFFF or more in their second array (mzd_k_execute.h: plan_store)."""
    v = next(x for x in VECS if x.name == name)
    force_driver("2" if v.out_len > 131072 else "1")
    rc, out, blocks, dump = oracle.decode(v.comp, cap=v.out_len, want_trace=True, dump=True)
    assert rc == 0
    st, got = mzd.decode(v.comp, v.out_len)
    assert st == 0 and got == out
    lit, seq = mzd.debug_last_block(0)
    assert lit == dump["lit"]
    assert seq == dump["seq"]


@pytest.mark.parametrize("driver", ["1", "1w", "1p", "4", "5"])
def test_sequences_with_lengths_of_64_kib_and_more(driver, force_driver):
    """The block pipeline's plan holds a sequence in 8 bytes, 16 bits a length; a literal run or a match of 0xFFFF bytes or more stands in full
    in the plan's second array (mzd_k_execute.h: plan_store / plan_expand).  Frames whose one compressed block has a literal run of 100 000
    bytes in front of a match of 30 000 (raw literals: noise), the same with the lengths around the escape value, and the two in a
    multi-block file: bytes against the generator's, the sequence triples against the oracle's dump."""
    if not oracle.LibZstd.available():
        pytest.skip("no libzstd to build the frames with")
    rng = np.random.RandomState(5)
    noise = rng.randint(0, 256, size=140000, dtype=np.uint8).tobytes()
    raws = [noise[:100000] + noise[1000:31000] + b"abc" * 10,
            noise[:65535] + noise[100:65635] + b"xyz",      # ll = 65535 = the escape value itself, ml = 65535
            noise[:65534] + noise[7:65541] + noise[:9],      # one below: the 16-bit form
            noise[:120000] + noise[:5000] + noise[60000:126000] + noise[20000:90000] + b"tail"]  # two blocks and more: the block tasks' hand-overs
    comps = [oracle.LibZstd.compress(r, level=3, checksum=True) for r in raws]
    force_driver(driver)
    res = mzd.decode_batch(comps, [len(r) for r in raws])
    for k, (r, (st, out)) in enumerate(zip(raws, res)):
        assert st == 0 and out == r, (k, st)
    for k in (0, 1, 2):
        rc, out, blocks, dump = oracle.decode(comps[k], cap=len(raws[k]), want_trace=True, dump=True)
        assert rc == 0 and out == raws[k]
        st, got = mzd.decode(comps[k], len(raws[k]))
        assert st == 0 and got == raws[k]
        lit, seq = mzd.debug_last_block(0)
        assert seq == dump["seq"], k
        if k == 0:
            assert any(ll >= 0xFFFF for ll, _, _ in seq)


def test_dst_too_small_and_empty():
    v = next(x for x in VECS if x.name == "json_4k")
    st, _ = mzd.decode(v.comp, 100)
    assert st == mzd.E_DSTSIZE
    assert mzd.decode(b"", 16) == (0, b"")


def test_dst_too_small_reports_the_capacity_that_suffices():
    """SURVEY.md 8(b) "Ownership": MZD_E_DSTSIZE comes with the required size, also for frames without a content size -- a second
    decode with it succeeds; the open() mirror needs at most two decodes for such a file."""
    import ctypes as C
    L = mzd.lib()
    for name in ("nofcs_stream_300k", "json_128k", "multi_frame_skippable"):
        v = next(x for x in VECS if x.name == name)
        buf = C.create_string_buffer(max(v.out_len, 1))
        n = C.c_size_t(0)
        assert L.mzd_decode(v.comp, len(v.comp), buf, 100, C.byref(n)) == mzd.E_DSTSIZE
        assert n.value == mzd.content_bound(v.comp) >= v.out_len
        big = C.create_string_buffer(n.value)
        assert L.mzd_decode(v.comp, len(v.comp), big, n.value, C.byref(n)) == 0 and n.value == v.out_len
        rc, want = oracle.decode(v.comp, cap=v.out_len)
        assert rc == 0 and big.raw[:n.value] == want
    v = next(x for x in VECS if x.name == "nofcs_stream_300k")  # 300 KB from ~50 KB of input: the first guess (1 MiB) holds it
    fs = mzd.ZstdFS()
    fh, size = fs.open(7, 0, v.comp)
    assert size == v.out_len and fs.decode_count <= 2 and fs.read(fh, 0, v.out_len) == oracle.decode(v.comp, cap=v.out_len)[1]
    fs.release(fh)
    fs.close()


def test_dst_too_small_every_vector_matches_oracle():
    """Every positive vector (raw, RLE, literal-only, Huffman+sequences, multi-block, multi-frame) into a buffer that is
    one byte short / half size / one byte long, interleaved with full-size neighbours: same status class as the CPU
    twin, and the neighbours (whose device buffers sit right behind the short ones) still come out intact, i.e. nothing
    was written past a capacity."""
    vs = [v for v in VECS if v.ok and v.dict is None and v.out_len > 0]
    for cap_of in (lambda n: n - 1, lambda n: n // 2, lambda n: 1):
        for parity in (0, 1):
            caps = [max(cap_of(v.out_len), 0) if i % 2 == parity else v.out_len for i, v in enumerate(vs)]
            res = mzd.decode_batch([v.comp for v in vs], caps)
            for v, cap, (st, out) in zip(vs, caps, res):
                rc, _ = oracle.decode(v.comp, cap=cap)
                assert st == rc, (v.name, cap, st, rc)
                assert st != 0 or out == v.expected()[:cap], v.name


def test_dictionary_frames():
    vs = [v for v in VECS if v.ok and v.dict is not None]
    did = mzd.load_dict(vs[0].dict)
    res = mzd.decode_batch([v.comp for v in vs], [v.out_len for v in vs], [did] * len(vs))
    for v, (st, out) in zip(vs, res):
        assert st == 0 and out == v.expected(), v.name
    # the same frames without the dictionary must fail like libzstd (dictionary_wrong)
    st, _ = mzd.decode(vs[0].comp, vs[0].out_len)
    assert st == mzd.E_DICT
    rc, _ = oracle.decode(vs[0].comp, cap=vs[0].out_len)
    assert rc == oracle.E_DICT


needs_zstd = pytest.mark.skipif(not corpus.have_zstd(), reason="no libzstd shared object to compress a corpus with")


@needs_zstd
def test_small_file_kernel_intermediates_match_cpu_twin():
    """The small-file kernel (mzd_lds.hip) hands its entropy phase's results to its execute phase through a scratch in HBM: the
    literals (K1/K2) and, per sequence, literal length, match length and offset VALUE (K3/K4, before repeat-offset resolution).
    Both against the oracle's dump of the same block, for files of every data class decoded alone on device pointers (resident
    file slot 0); the offsets are resolved here with the rule of A.5 and compared with the oracle's resolved ones."""
    import torch
    L = mzd.lib()
    L.mzd_debug_small_scratch.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    dev = torch.device("cuda:0")
    checked = 0
    skipped = []
    mzd.set_driver(3)
    try:
        for kind, size, level in (("json", 4096, 3), ("json", 8000, 19), ("text", 3000, 3), ("markup", 2500, 1), ("int32", 4096, 3), ("dna", 6000, 3), ("repeats", 8192, 3), ("json", 700, 3),
                                  ("json", 2000, 1), ("markup", 4096, 3), ("text", 1500, 19), ("json", 5000, 3)):
            cp = corpus.build_corpus(kind, 31, [size], level=level)
            comp = cp.comp_file(0).tobytes()
            rc, want, blocks, dump = oracle.decode(comp, cap=size, want_trace=True, dump=True)
            assert rc == 0 and len(blocks) == 1
            b = blocks[0]
            if b["block_type"] != 2 or b["huf_max_bits"] > 10:
                continue  # (raw / RLE blocks have no intermediates; 11-bit Huffman tables are handed on to the general driver)
            src = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
            out = torch.zeros(size + 64, dtype=torch.uint8, device=dev)
            jobs = mzd.api.make_jobs([src.data_ptr()], [len(comp)], [out.data_ptr()], [size])
            res = mzd.decode_batch_device(0, jobs)
            assert res[0][0] == 0 and bytes(out.cpu().numpy()[:size]) == want
            if mzd.debug_counters(0)[4] != 0:
                skipped.append((kind, size, level))  # handed on to the general driver (tables that do not fit the slot): no scratch to look at
                continue
            nlit, nseq = b["n_lit"], b["n_seq"]
            lit = (C.c_uint8 * max(nlit, 1))()
            seq = (C.c_uint64 * max(nseq, 1))()
            assert L.mzd_debug_small_scratch(0, 0, lit, nlit if b["lit_type"] != 0 else 0, seq, nseq) == 0
            if b["lit_type"] != 0:  # (raw literals stay in the input: nothing is written to the scratch)
                assert bytes(lit[:nlit]) == dump["lit"], (kind, size)
            rep = [1, 4, 8]
            for k, (ll, ml, off) in enumerate(dump["seq"]):
                v = seq[k]
                gll, gml, ofv = v & 0x3FFF, (v >> 14) & 0x3FFF, v >> 32
                assert (gll, gml) == (ll, ml), (kind, size, k)
                if ofv > 3:
                    got = ofv - 3; rep = [got, rep[0], rep[1]]
                else:
                    idx = ofv - 1 + (1 if ll == 0 else 0)
                    if idx == 0: got = rep[0]
                    elif idx == 1: got = rep[1]; rep = [rep[1], rep[0], rep[2]]
                    elif idx == 2: got = rep[2]; rep = [rep[2], rep[0], rep[1]]
                    else: got = rep[0] - 1; rep = [got, rep[0], rep[1]]
                assert got == off, (kind, size, k, got, off)
            checked += 1
    finally:
        mzd.set_driver(0)
    assert checked >= 6, (checked, skipped)


def test_hand_over_between_block_tasks_holds_under_repetition():
    """Every positive vector in one batch, 250 times, under the library's own choice of driver.  `window_log10` is 586 blocks of
    1 KiB, each a task on another workgroup (often another XCD), resolved ahead of its predecessor, with the checksum chain's state
    travelling from task to task behind its own flag: a hand-over flag stored without an agent-scope release in front of it let a
    successor see the flag before the data about once in a hundred runs (right bytes, wrong digest; tools/stress_handover.py)."""
    vs = [v for v in VECS if v.ok and v.dict is None]
    comps, caps = [v.comp for v in vs], [v.out_len for v in vs]
    for rep in range(250):
        res = mzd.decode_batch(comps, caps)
        bad = [(v.name, st) for v, (st, out) in zip(vs, res) if st != 0 or out != v.expected()]
        assert not bad, (rep, bad)


@pytest.mark.parametrize("driver", DRIVERS)
def test_both_drivers_decode_every_vector(driver, force_driver):
    """The library has two kernel drivers: one workgroup per file (launches whose capacities are all <= 128 KiB) and
    block tasks (the blocks of a frame on different workgroups: tables, repeat offsets, output position and checksum
    state handed from task to task).  Forced through mzd_debug_set_driver (and left to the library: 'auto', where small files take the small-file kernel first), each must decode every positive vector (single- and
    multi-block, multi-frame, skippable, windows > 128 KiB) byte-exactly, in one batch and one by one, and report the
    oracle's error class on every negative vector."""
    force_driver(driver)
    vs = [v for v in VECS if v.ok and v.dict is None]
    res = mzd.decode_batch([v.comp for v in vs], [v.out_len for v in vs])
    bad = [(v.name, st) for v, (st, out) in zip(vs, res) if st != 0 or out != v.expected()]
    assert not bad, bad
    for v in vs:
        if len(v.comp) > (64 << 10) or v.out_len > (128 << 10): # the multi-block ones again, alone (few workgroups busy)
            st, out = mzd.decode(v.comp, v.out_len)
            assert st == 0 and out == v.expected(), v.name
    for v in [v for v in VECS if not v.ok]:
        st, _ = mzd.decode(v.comp, 1 << 22)
        assert st == v.oracle_class, (v.name, st, v.oracle_class)
    dv = [v for v in VECS if v.ok and v.dict is not None]
    if dv:
        did = mzd.load_dict(dv[0].dict)
        for v, (st, out) in zip(dv, mzd.decode_batch([v.comp for v in dv], [v.out_len for v in dv], [did] * len(dv))):
            assert st == 0 and out == v.expected(), v.name


@needs_zstd
@pytest.mark.parametrize("driver", DRIVERS + ["auto:8:4", "auto:8:4:2", "auto:4:2", "auto:8:8", "auto:16:16"])  # (the small-file kernel in every shape it is built in)
def test_every_single_byte_mutation_of_small_frames_matches_oracle(driver, force_driver):
    """The fuzz corpus of SURVEY.md row N3 on the device: EVERY byte of several small frames (Huffman + FSE blocks, a
    raw-literal block, an RLE-heavy one, levels 3 and 19) flipped three ways, every truncation, and every output capacity of one frame: ~14 000 cases in one launch.  For each
    mutant the status must be the oracle's and, where both accept, the bytes too -- a mutant must never hang, fault or
    write outside its output buffer (every job gets its own buffer: a stray write shows up as a neighbour's mismatch)."""
    force_driver(driver)
    Z = oracle.LibZstd
    frames = []
    for kind, seed, size, level in (("json", 12, 700, 3), ("json", 13, 2500, 19), ("text", 14, 2000, 3), ("repeats", 15, 3000, 3),
                                    ("int32", 16, 1200, 3), ("random", 17, 300, 3)):
        raw = corpus.gen(kind, seed, 1, size)
        frames.append((Z.compress(raw, level, True), size))
    cases = []
    for comp, size in frames:
        for pos in range(len(comp)):
            for flip in (0x01, 0x80, 0xFF):
                m = bytearray(comp)
                m[pos] ^= flip
                cases.append((bytes(m), size + 64))
        for cut in range(len(comp)):  # every truncation
            cases.append((comp[:cut], size + 64))
    comp0, size0 = frames[0]
    for cap in range(size0 + 2):      # every output capacity up to one more than needed
        cases.append((comp0, cap))
    res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
    bad = []
    for i, ((comp, cap), (st, out)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(comp, cap=cap)
        if st != rc or (st == 0 and out != want):
            bad.append((i, st, rc))
    assert not bad, (len(bad), bad[:10])


@needs_zstd
@pytest.mark.parametrize("driver", DRIVERS)
def test_multi_byte_mutations_match_oracle(driver, force_driver):
    """One to three mutated bytes per frame, eight data classes x three levels (tools/fuzz_more.py runs the same generator
    with more cases).  Several things are wrong at once in such frames, so the status depends on WHICH error is found
    first: the oracle decodes a block's sequences and literals before it executes any sequence, then takes the sequences
    in order (destination's end, 128 KiB block limit, offset).  The device pipeline reports in that order too -- e.g. a
    highly repetitive frame ("repeats": 128 KiB from 128 bytes) whose first sequence got a bad offset AND whose output
    would pass the destination is "corrupt", not "destination too small"."""
    force_driver(driver)
    rng = np.random.RandomState(8)
    cases = []
    for kind, seed, size in (("json", 41, 131072), ("text", 42, 100000), ("markup", 43, 60000), ("xray", 44, 131072), ("json", 45, 20000),
                             ("dna", 46, 50000), ("repeats", 47, 131072), ("json", 48, 4096)):
        for level in (1, 3, 19):
            cp = corpus.build_corpus(kind, seed, [size], level=level)
            good = cp.comp_file(0).tobytes()
            for _ in range(60):
                b = bytearray(good)
                for _ in range(int(rng.randint(1, 4))):
                    b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
                cases.append((bytes(b), size))
    res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
    bad = []
    for i, ((comp, cap), (st, out)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(comp, cap=cap)
        if st != rc or (st == 0 and out != want):
            bad.append((i, st, rc))
    assert not bad, (len(bad), bad[:10])


@needs_zstd
@pytest.mark.parametrize("driver", DRIVERS)
def test_literal_streams_not_consumed_exactly_follow_the_pinned_libzstd(driver, force_driver):
    """Row A7b (foreign files, reference README.md:20-26): frames WITHOUT a checksum whose Huffman literal streams no longer end at their first
    bit.  The reference's libzstd 1.5.x accepts most of them -- the fast loops of a four-stream section decode its symbols and never look at
    the streams' ends: leftover bits are ignored, a stream that runs out reads on into the bytes in front of it -- and returns bytes;
    RFC 8878 and libzstd 1.4 call them corrupt.  The oracle restates 1.5's rule and is pinned against 1.5's bytes on exactly these mutants
    (tests/test_oracle.py::test_error_classes_against_libzstd_1_5_when_loadable, same generator and seed); here every mutant's status and bytes
    on the device must be the oracle's, and the launch must contain accepted mutants of both kinds."""
    from tests.mutants import NOCHK_FRAMES, nochk_mutants
    force_driver(driver)
    rng = np.random.RandomState(5)
    cases = []
    for kind, size, level, extra in NOCHK_FRAMES:
        cases += nochk_mutants(kind, size, level, extra, 400 if kind == "xray" else 150, rng)
    res = mzd.decode_batch([m for _, m, _ in cases], [cap for _, _, cap in cases])
    bad = []
    lenient = through = 0
    for i, ((name, m, cap), (st, out)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(m, cap=cap)
        if rc == 0:
            lenient += oracle.last_verdict_lit_lenient()
            through += oracle.last_verdict_lit_through()
        if st != rc or (st == 0 and out != want):
            bad.append((i, name, st, rc))
    assert not bad, (len(bad), bad[:10])
    assert lenient > 100 and through > 100, (lenient, through)


@needs_zstd
def test_mutated_dictionary_frames_and_multi_frame_files_match_oracle():
    """Config-5-shaped records with mutated bytes, with the dictionary missing, truncated; files of frame + skippable frame +
    frame with mutated bytes (tools/fuzz_dict.py runs the same generator with more cases)."""
    rng = np.random.RandomState(3)
    Z = oracle.LibZstd
    assert Z.available()
    sizes = [int(x) for x in np.random.RandomState(55).randint(300, 3001, size=200)]
    d = corpus.train_dict("json", 5, sizes[:150], cap=40000)
    h = mzd.load_dict(d)
    cp = corpus.build_corpus("json", 5, sizes, dictionary=d)
    cases = []
    for i in range(0, 200, 2):
        good = cp.comp_file(i).tobytes()
        for _ in range(6):
            b = bytearray(good)
            for _ in range(int(rng.randint(1, 3))):
                b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
            cases.append((bytes(b), sizes[i], d, h))
        cases.append((good, sizes[i], None, 0))
        cases.append((good[:int(rng.randint(0, len(good)))], sizes[i], d, h))
    for k in range(60):
        a = Z.compress(corpus.gen("json", 60 + k, 1, 3000), 3, True)
        b2 = Z.compress(corpus.gen("text", 61 + k, 1, 2000), 3, bool(k & 1))
        skip = (0x184D2A50 + (k & 15)).to_bytes(4, "little") + (7).to_bytes(4, "little") + b"skipped"
        f = bytearray(a + skip + b2)
        if k % 3:
            for _ in range(int(rng.randint(1, 3))):
                f[int(rng.randint(0, len(f)))] ^= int(rng.randint(1, 256))
        cases.append((bytes(f), 5000 if k % 5 else 4000, None, 0))
    res = mzd.decode_batch([c[0] for c in cases], [c[1] for c in cases], [c[3] for c in cases])
    bad = []
    for i, ((comp, cap, dd, _), (st, out)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(comp, cap=cap, dictionary=dd)
        if st != rc or (st == 0 and out != want):
            bad.append((i, st, rc))
    assert not bad, (len(bad), bad[:10])


@needs_zstd
@pytest.mark.parametrize("nd", [1, 5, 8])
def test_mutated_records_of_one_dictionary_under_every_shape_of_the_dictionary_kernels(nd):
    """Config 5's kernels keep one dictionary table image for ND decoding wavefronts of a workgroup (ND = 5 or 8; the library picks them only
    for launches whose files ALL name the same dictionary, so the mixed batch above runs ND = 1).  Here every job names the dictionary: valid
    records, records with one to three mutated bytes, truncated ones -- what is corrupt is handed on to the general driver from inside a
    workgroup whose other wavefronts go on decoding around the same image -- under each forced ND (mzd_debug_host_path 12), status and bytes
    against the oracle."""
    rng = np.random.RandomState(17)
    # (eight wavefronts' slots beside one table image fit a CU only for small records: the library would not take ND = 8 for 3 000-byte ones)
    sizes = [int(x) for x in np.random.RandomState(56).randint(300, 1001 if nd == 8 else 3001, size=400)]
    d = corpus.train_dict("json", 6, sizes[:200], cap=40000)
    h = mzd.load_dict(d)
    cp = corpus.build_corpus("json", 6, sizes, dictionary=d)
    cases = []
    for i in range(400):
        good = cp.comp_file(i).tobytes()
        cases.append((good, sizes[i]))
        for _ in range(3):
            b = bytearray(good)
            for _ in range(int(rng.randint(1, 4))):
                b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
            cases.append((bytes(b), sizes[i]))
        cases.append((good[:int(rng.randint(0, len(good)))], sizes[i]))
        cases.append((good, max(0, sizes[i] - int(rng.randint(1, 40)))))  # (a destination that is too small)
    # device-resident (one launch on the whole device: the kernel's name is the library's record of what it ran)
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    offs = np.zeros(len(cases) + 1, dtype=np.int64); ooffs = np.zeros(len(cases) + 1, dtype=np.int64)
    for i, (c, cap) in enumerate(cases):
        offs[i + 1] = offs[i] + ((len(c) + 64 + 15) & ~15); ooffs[i + 1] = ooffs[i] + ((cap + 64 + 15) & ~15)
    img = np.zeros(int(offs[-1]) + 64, dtype=np.uint8)
    for i, (c, _) in enumerate(cases):
        img[int(offs[i]):int(offs[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    comp = torch.from_numpy(img).to(dev)
    out = torch.zeros(int(ooffs[-1]) + 64, dtype=torch.uint8, device=dev)
    jobs = mzd.api.make_jobs([comp.data_ptr() + int(o) for o in offs[:-1]], [len(c) for c, _ in cases], [out.data_ptr() + int(o) for o in ooffs[:-1]], [cap for _, cap in cases], [h] * len(cases))
    torch.cuda.synchronize()
    mzd.set_driver(3)
    mzd.lib().mzd_debug_host_path(0, 12, nd)
    try:
        res = mzd.decode_batch_device(0, jobs)
        name = mzd.last_kernel_name(0)
    finally:
        mzd.lib().mzd_debug_host_path(0, 12, 0)
        mzd.set_driver(0)
        mzd.unload_dict(h)
    if nd > 1:
        assert ",%d>" % nd in name.replace(" ", ""), name  # (mzd_lds_kernel<8,true,8,1,ND>)
    host = out.cpu().numpy()
    bad = []
    for i, ((c, cap), (st, n)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(c, cap=cap, dictionary=d)
        if st != rc or (st == 0 and host[int(ooffs[i]):int(ooffs[i]) + n].tobytes() != want):
            bad.append((i, st, rc))
    assert not bad, (nd, len(bad), bad[:10])


@needs_zstd
def test_random_mutations_of_128k_frames_match_oracle():
    """Single-block frames of the config-2 size (thousands of sequences: the state walk runs through many ring refills,
    the planner and the copier follow it through HBM queues), 400 random single-byte mutations each.  A corrupt
    sequence bitstream must end in the oracle's status without any stage acting on positions outside the stream."""
    rng = np.random.RandomState(91)
    cases = []
    for kind, seed in (("json", 31), ("text", 32), ("int32", 33), ("markup", 34)):
        cp = corpus.build_corpus(kind, seed, [131072])
        good = cp.comp_file(0).tobytes()
        for _ in range(400):
            b = bytearray(good)
            pos = int(rng.randint(0, len(b)))
            b[pos] ^= int(rng.randint(1, 256))
            cases.append((bytes(b), 131072))
    res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
    bad = []
    for i, ((comp, cap), (st, out)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(comp, cap=cap)
        if st != rc or (st == 0 and out != want):
            bad.append((i, st, rc))
    assert not bad, (len(bad), bad[:10])


@needs_zstd
@pytest.mark.parametrize("driver", DRIVERS)
def test_corrupted_multi_block_files_report_the_oracles_error(driver, force_driver):
    """Single-byte mutations in every block of multi-block frames (and truncations, and too-small outputs): the status
    must be the oracle's class -- with block tasks an error has to travel from the task that finds it to the task that
    closes the file, in stream order -- and a decode that still succeeds must produce the oracle's bytes."""
    force_driver(driver)
    rng = np.random.RandomState(77)
    cases = []
    for kind, size in (("json", 600000), ("text", 400000), ("xray", 300000), ("repeats", 500000)):
        cp = corpus.build_corpus(kind, 21, [size])
        good = cp.comp_file(0).tobytes()
        for _ in range(120):
            b = bytearray(good)
            pos = int(rng.randint(0, len(b)))
            b[pos] ^= int(rng.randint(1, 256))
            cases.append((bytes(b), size))
        for cut in (len(good) - 1, len(good) - 5, len(good) // 2, 40):
            cases.append((good[:cut], size))
        for cap in (size - 1, size // 2, 150000, 10):
            cases.append((good, cap))
    res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
    for i, ((comp, cap), (st, out)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(comp, cap=cap)
        assert st == rc, (i, cap, st, rc)
        assert st != 0 or out == want, i


@needs_zstd
@pytest.mark.parametrize("driver", DRIVERS)
def test_checksum_failures_inside_multi_frame_files_of_multi_block_frames(driver, force_driver):
    """Three multi-block frames in one file; the checksum of the first, second or third frame is wrong (or a byte of a block's
    content is).  With blocks resolved ahead the bytes' chain runs ahead of the checksum chain, so the failure is found when
    later frames have long been handed over: status and length must still be the oracle's (the first error in stream order)."""
    force_driver(driver)
    frames = [corpus.build_corpus(kind, 91 + k, [size]).comp_file(0).tobytes() for k, (kind, size) in enumerate((("json", 300000), ("text", 280000), ("json", 400000)))]
    total = 300000 + 280000 + 400000
    ends = np.cumsum([len(f) for f in frames])
    good = b"".join(frames)
    cases = [(good, total)]
    rng = np.random.RandomState(5)
    for k in range(3):
        b = bytearray(good); b[int(ends[k]) - 2] ^= 0x5A; cases.append((bytes(b), total))          # the stored checksum of frame k
        b = bytearray(good); pos = int(ends[k]) - 2000 - int(rng.randint(0, 50000)); b[pos] ^= 0x11; cases.append((bytes(b), total))  # content of frame k
    cases.append((good, 300000 + 280000 + 10))   # the destination ends inside the third frame
    res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
    for i, ((comp, cap), (st, out)) in enumerate(zip(cases, res)):
        rc, want = oracle.decode(comp, cap=cap)
        assert st == rc, (i, st, rc)
        assert st != 0 or out == want, i
    assert res[0][0] == 0 and res[1][0] == mzd.E_CHECKSUM


@needs_zstd
@pytest.mark.parametrize("driver", DRIVERS)
def test_multi_block_corpus_both_drivers(driver, force_driver):
    """Seeded files of 4 KiB .. 1 MiB (up to 8 blocks, treeless literals and repeat tables between them) in one batch."""
    force_driver(driver)
    sizes = [4096, 200000, 1 << 20, 131073, 70000, 1 << 20, 300000, 4096, 655360, 131072, 262144, 99999] * 3
    for kind in ("json", "text", "xray", "repeats"):
        cp = corpus.build_corpus(kind, 7, sizes)
        res = mzd.decode_batch([cp.comp_file(i).tobytes() for i in range(len(sizes))], sizes)
        for i, (st, out) in enumerate(res):
            assert st == 0 and out == cp.raw_file(i).tobytes(), (kind, i, sizes[i], st)




@needs_zstd
@pytest.mark.parametrize("kind,level", [("json", 3), ("json", 1), ("json", 19), ("text", 3), ("markup", 9), ("int32", 3),
                                        ("dna", 3), ("xray", 3), ("random", 3), ("repeats", 3)])
def test_seeded_corpus_vs_oracle(kind, level):
    """Seeded inputs at sizes the oracle finishes in seconds: GPU == oracle == original."""
    sizes = [0, 1, 17, 300, 4096, 20000, 65536, 131072, 131073, 262144, 400000][: (7 if level == 19 else 11)]
    cp = corpus.build_corpus(kind, 21, sizes, level=level)
    srcs = [cp.comp_file(i).tobytes() for i in range(cp.nfiles)]
    res = mzd.decode_batch(srcs, [int(s) for s in sizes])
    for i, (st, out) in enumerate(res):
        raw = cp.raw_file(i).tobytes()
        rc, ref = oracle.decode(srcs[i], cap=len(raw))
        assert rc == 0 and ref == raw
        assert st == 0 and out == raw, (kind, level, sizes[i], st)


@needs_zstd
@pytest.mark.parametrize("driver", DRIVERS)
def test_multi_block_frames_with_a_dictionary(driver, force_driver):
    """Frames of several blocks compressed WITH a dictionary: the first block starts from the dictionary's tables and
    repeat offsets, later blocks inherit tables from their predecessors (block tasks: through the file's table area)
    and matches may reach through earlier blocks into the dictionary content."""
    force_driver(driver)
    sizes = [150000, 400000, 131073, 3000, 262144, 700000]
    d = corpus.train_dict("json", 12, [2500] * 2000)
    cp = corpus.build_corpus("json", 12, sizes, dictionary=d, first_index=5000)
    h = mzd.load_dict(d)
    srcs = [cp.comp_file(i).tobytes() for i in range(len(sizes))]
    res = mzd.decode_batch(srcs, sizes, [h] * len(sizes))
    for i, (st, out) in enumerate(res):
        assert st == 0 and out == cp.raw_file(i).tobytes(), (i, sizes[i], st)
    rc, out = oracle.decode(srcs[1], cap=sizes[1], dictionary=d)
    assert rc == 0 and out == cp.raw_file(1).tobytes()
    st, _ = mzd.decode(srcs[1], sizes[1])  # the same frame without its dictionary
    assert st == oracle.decode(srcs[1], cap=sizes[1])[0] != 0


@needs_zstd
def test_config5_shape_shared_dictionary():
    """BASELINE config 5 in small: records of 300..3000 B compressed with one trained dictionary (treeless literals,
    repeat-mode tables: the dictionary's tables stay resident in LDS from file to file), interleaved with plain frames
    (which rebuild tables, so the residency marks must be dropped and restored) and with frames of a second dictionary."""
    rng = np.random.RandomState(5)
    sizes = [int(x) for x in rng.randint(300, 3001, size=3000)]
    d1 = corpus.train_dict("json", 5, sizes[:1500])
    d2 = corpus.train_dict("text", 6, [2000] * 600, cap=40000)
    c1 = corpus.build_corpus("json", 5, sizes, dictionary=d1)
    c2 = corpus.build_corpus("text", 6, sizes[:500], dictionary=d2)
    c0 = corpus.build_corpus("json", 9, sizes[:500])
    h1, h2 = mzd.load_dict(d1), mzd.load_dict(d2)
    srcs, caps, dids, want = [], [], [], []
    for i in range(len(sizes)):
        srcs.append(c1.comp_file(i).tobytes()); caps.append(sizes[i]); dids.append(h1); want.append(c1.raw_file(i).tobytes())
        if i % 6 == 0 and i // 6 < 500:
            k = i // 6
            srcs.append(c0.comp_file(k).tobytes()); caps.append(sizes[k]); dids.append(0); want.append(c0.raw_file(k).tobytes())
            srcs.append(c2.comp_file(k).tobytes()); caps.append(sizes[k]); dids.append(h2); want.append(c2.raw_file(k).tobytes())
    res = mzd.decode_batch(srcs, caps, dids)
    bad = [(i, st) for i, ((st, out), w) in enumerate(zip(res, want)) if st != 0 or out != w]
    assert not bad, bad[:10]
    rc, out = oracle.decode(srcs[0], cap=caps[0], dictionary=d1)
    assert rc == 0 and out == want[0]


@needs_zstd
def test_config2_shape_device_resident():
    """BASELINE config 2 shape, scaled down to 64 files: 128 KiB single-block JSON frames decoded
    from HBM to HBM through mzd_decode_batch_device; checked against the generator's bytes."""
    torch = pytest.importorskip("torch")
    nfiles = 64
    cp = corpus.build_corpus("json", 2, [131072] * nfiles, level=3)
    dev = torch.device("cuda:0")
    comp = torch.from_numpy(np.concatenate([cp.comp, np.zeros(64, np.uint8)])).to(dev)
    out = torch.zeros(nfiles * 131072, dtype=torch.uint8, device=dev)
    jobs = mzd.api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes,
                             [out.data_ptr() + i * 131072 for i in range(nfiles)], [131072] * nfiles)
    torch.cuda.synchronize()
    res = mzd.decode_batch_device(0, jobs)
    assert all(st == 0 and n == 131072 for st, n in res), res[:4]
    got = out.cpu().numpy()
    for i in range(nfiles):
        assert got[i * 131072:(i + 1) * 131072].tobytes() == cp.raw_file(i).tobytes(), i


def test_fs_open_read_release_mirror():
    """open_wrapper / read_wrapper / release semantics (reference src/main.rs:451-513, src/file.rs)."""
    v = next(x for x in VECS if x.name == "json_1m")
    want = v.expected()
    fs = mzd.ZstdFS()
    fh, size = fs.open(42, 0, v.comp)
    assert size == len(want) and fs.decode_count == 1
    fh2, size2 = fs.open(42, 0, v.comp)  # second open of the inode: duplicate, no decode
    assert fh2 != fh and size2 == size and fs.decode_count == 1
    got = b"".join(fs.read(fh, off, 131072) for off in range(0, len(want) + 131072, 131072))
    assert got == want
    assert fs.read(fh2, len(want) - 10, 4096) == want[-10:]  # short read at EOF
    fs.release(fh)
    assert fs.read(fh2, 0, 16) == want[:16]  # bytes stay valid while a handle remains
    with pytest.raises(OSError) as e:
        fs.read(fh, 0, 16)
    import errno
    assert e.value.errno == errno.ENOENT
    fs.release(fh2)
    bad = next(x for x in VECS if x.name == "bad_checksum")
    with pytest.raises(OSError) as e:
        fs.open(43, 0, bad.comp)
    assert e.value.errno == errno.EFAULT  # `.map_err(|_| libc::EFAULT)` src/main.rs:467
    fs.close()
