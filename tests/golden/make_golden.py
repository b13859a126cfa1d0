#!/usr/bin/env python3
"""Generates tests/golden/frames/* and tests/golden/manifest.json.

Run in the build container (needs a libzstd shared object; the one used is recorded in the
manifest).  Compressed bytes differ between libzstd versions, so the BYTES are committed; the
expected output is stored inline (small), or as (generator, seed, size) + XXH64 + length, and
every positive vector was decoded by libzstd itself before being written (one-shot AND the
8 KiB streaming shape of copy_decode, reference src/main.rs:463-467).

Groups
  ref_*      payloads of the reference's own tests, in both encodings they occur in:
             zstd::bulk::compress(x, 0) (tests/convert.rs:15-43) and the writer's settings
             level 3 + checksum + pledged size (src/main.rs:781-791; payload strings from
             tests/cmdline.rs:19-29,126,145,164 and tests/glitches.rs).
  json_* / proxy_*  synthetic corpora incl. the 1 MiB multi-block treeless chain.
  hand_*     hand-built frames for branches libzstd's encoder rarely/never emits
             (direct-weight Huffman header, RLE literals, RLE sequence tables, long nbSeq form).
  multi_* / nofcs_* / window_* / dict_*   container features.
  bad_*      negative set: expected = error class only.
"""
import base64
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import corpus  # noqa: E402
import oracle  # noqa: E402
from oracle import LibZstd as Z  # noqa: E402

FR = os.path.join(HERE, "frames")
os.makedirs(FR, exist_ok=True)
manifest = {"libzstd": None, "vectors": []}


def add(name, comp, expect_raw=None, gen=None, expect="ok", dictionary=None, note=""):
    """expect_raw: bytes (stored inline if small) ; gen: (kind,cfg,index,size) to regenerate."""
    with open(os.path.join(FR, name + ".zst"), "wb") as f:
        f.write(comp)
    e = {"name": name, "expect": expect, "note": note, "comp_len": len(comp)}
    if dictionary:
        e["dict"] = dictionary
    if expect == "ok":
        raw = expect_raw
        dct = open(os.path.join(FR, dictionary), "rb").read() if dictionary else None
        # libzstd is the pin: one-shot and the streaming shape must both give `raw`
        got = Z.decompress(comp, len(raw) + 64, dictionary=dct)
        assert got == raw, (name, "libzstd one-shot disagrees")
        if dct is None:
            got2 = Z.decompress(comp, len(raw) + 64, stream8k=True)
            assert got2 == raw, (name, "libzstd streaming disagrees")
        rc, out, blocks = oracle.decode(comp, cap=len(raw) + 64, dictionary=dct, want_trace=True)
        assert rc == 0 and out == raw, (name, "oracle disagrees", rc)
        e["out_len"] = len(raw)
        e["out_xxh64"] = "%016x" % oracle.xxh64(raw)
        if gen is not None:
            e["gen"] = list(gen)
        elif len(raw) <= 4096:
            e["out_b64"] = base64.b64encode(raw).decode()
        else:
            with open(os.path.join(FR, name + ".raw"), "wb") as f:
                f.write(raw)
            e["out_file"] = name + ".raw"
        e["blocks"] = blocks  # CPU-twin intermediates (literal / sequence hashes per block)
    else:
        dct = open(os.path.join(FR, dictionary), "rb").read() if dictionary else None
        got = Z.decompress(comp, 1 << 22, dictionary=dct)
        assert isinstance(got, int), (name, "libzstd accepted a negative vector")
        got2 = Z.decompress(comp, 1 << 22, stream8k=True) if dct is None else -1
        assert isinstance(got2, int), (name, "libzstd streaming accepted a negative vector")
        rc, _ = oracle.decode(comp, cap=1 << 22, dictionary=dct)
        assert rc != 0, (name, "oracle accepted a negative vector")
        e["oracle_class"] = rc
    manifest["vectors"].append(e)
    return e


# ------------------------------------------------------------------ hand-built frame helpers
def backward_stream(bits):
    """bits: string of '0'/'1' in the order the decoder reads them -> bytes (marker added)."""
    v = int("1" + bits, 2) if bits else 1
    return v.to_bytes((v.bit_length() + 7) // 8, "little")


def bits(v, n):
    return format(v, "0%db" % n) if n else ""


def block_header(last, btype, size):
    return ((size << 3) | (btype << 1) | last).to_bytes(3, "little")


def frame(blocks_bytes, content, checksum=True, single=True, window_byte=None):
    """Frame header with an 8-byte FCS (flag 3)."""
    fhd = (3 << 6) | ((1 if single else 0) << 5) | ((1 if checksum else 0) << 2)
    out = bytes([0x28, 0xB5, 0x2F, 0xFD, fhd])
    if not single:
        out += bytes([window_byte])
    out += len(content).to_bytes(8, "little") + blocks_bytes
    if checksum:
        out += (oracle.xxh64(content) & 0xFFFFFFFF).to_bytes(4, "little")
    return out


def lit_header_raw_rle(ltype, regen):
    if regen < 32:
        return bytes([(regen << 3) | ltype])
    if regen < 4096:
        return ((regen << 4) | (1 << 2) | ltype).to_bytes(2, "little")
    return ((regen << 4) | (3 << 2) | ltype).to_bytes(3, "little")


def nbseq_bytes(n, force_long=False):
    if n < 128 and not force_long:
        return bytes([n])
    if n < 0x7F00 and not force_long:
        return bytes([(n >> 8) + 128, n & 255])
    return bytes([255]) + (n - 0x7F00).to_bytes(2, "little")


LL_CODE16 = (16, 1)   # base 16, 1 extra bit
ML_CODE32 = (35, 1)   # base 35, 1 extra bit


def hand_rle_everything():
    """RLE literals + all three sequence tables in RLE mode."""
    # literals: 40 x 'Z' (RLE).  sequences: LL code 16 (ll=16|17), OF code 3 (offset_value 8..15), ML code 32 (35|36)
    seqs = [(16, 35, 12), (17, 36, 9)]  # (ll, ml, offset_value) ; 16+17 = 33 literals used, 7 trail
    lit = b"Z" * 40
    body = lit_header_raw_rle(1, 40) + b"Z"
    body += nbseq_bytes(len(seqs)) + bytes([(1 << 6) | (1 << 4) | (1 << 2)]) + bytes([16, 3, 32])
    b = ""
    for ll, ml, ofv in seqs:
        b += bits(ofv - 8, 3) + bits(ml - 35, 1) + bits(ll - 16, 1)
    body += backward_stream(b)
    # reference execution
    out = bytearray(); lp = 0
    for ll, ml, ofv in seqs:
        out += lit[lp:lp + ll]; lp += ll
        off = ofv - 3
        for _ in range(ml):
            out.append(out[-off])
    out += lit[lp:]
    return frame(block_header(1, 2, len(body)) + body, bytes(out)), bytes(out)


def hand_long_nbseq(n=0x7F00 + 100):
    """>= 0x7F00 sequences (3-byte nbSeq form): ll=0, ml=3, repeat offset 1 via RLE tables."""
    lit = b"ab"
    # first sequence: ll=2 would need another LL code; keep RLE tables: LL code 0 (ll=0) for all,
    # so the 2 literals trail; offsets must point into history: use a raw first block as history.
    hist = b"0123456789" * 4
    body = lit_header_raw_rle(0, len(lit)) + lit
    body += nbseq_bytes(n) + bytes([(1 << 6) | (1 << 4) | (1 << 2)]) + bytes([0, 4, 0])  # LL code 0, OF code 4, ML code 0 (ml=3)
    b = ""
    ofvs = []
    for i in range(n):
        ofv = 16 + (i * 7) % 16  # offset_value 16..31 -> offset 13..28
        ofvs.append(ofv)
        b += bits(ofv - 16, 4)
    body += backward_stream(b)
    out = bytearray(hist)
    for ofv in ofvs:
        off = ofv - 3
        for _ in range(3):
            out.append(out[-off])
    out += lit
    blocks = block_header(0, 0, len(hist)) + hist + block_header(1, 2, len(body)) + body
    return frame(blocks, bytes(out)), bytes(out)


def hand_direct_weights():
    """Huffman tree given as direct 4-bit weights, 1-stream and 4-stream literal sections, nbSeq=0."""
    # symbols 0..3 with weights w0=2,w1=1,w2=1,(w3 implicit=3) -> lengths 2,3,3,1
    code = {0: "01", 1: "000", 2: "001", 3: "1"}
    import random
    rnd = random.Random(7)
    out_frames = []
    for streams, nlit in ((1, 200), (4, 1021)):
        lit = bytes(rnd.choice([3, 3, 3, 3, 0, 0, 1, 2]) for _ in range(nlit))
        tree = bytes([127 + 3, (2 << 4) | 1, (1 << 4) | 0])  # 3 explicit weights 2,1,1
        if streams == 1:
            payload = backward_stream("".join(code[c] for c in lit))
            comp = tree + payload
            hdr = (2 | (0 << 2) | (nlit << 4) | (len(comp) << 14)).to_bytes(3, "little")
        else:
            seg = (nlit + 3) // 4
            parts = [lit[0:seg], lit[seg:2 * seg], lit[2 * seg:3 * seg], lit[3 * seg:]]
            ss = [backward_stream("".join(code[c] for c in p)) for p in parts]
            jump = b"".join(len(s).to_bytes(2, "little") for s in ss[:3])
            comp = tree + jump + b"".join(ss)
            hdr = (2 | (2 << 2) | (nlit << 4) | (len(comp) << 18)).to_bytes(4, "little")
        body = hdr + comp + nbseq_bytes(0)
        out_frames.append((frame(block_header(1, 2, len(body)) + body, lit), lit))
    return out_frames


def hand_huffman_depth(maxbits, streams, nlit, seed=5, treeless_second=0):
    """A Huffman tree of depth `maxbits` given as direct weights 1, 1, 2, 3, ..., maxbits (the sum of 2^(w-1) is 2^maxbits; the last
    weight is implied), literals drawn mostly from the short codes but with every symbol present, nbSeq = 0.  Depth 12 is libzstd's
    limit (HUF_TABLELOG_MAX: 1.4.8 and 1.5.7 decode these frames), depth 13 is refused.  treeless_second: a second block of that many
    literals whose literals section reuses the first block's tree (Treeless_Literals_Block)."""
    import random
    ws = [1, 1] + list(range(2, maxbits + 1))
    nsym = len(ws)
    order = sorted(range(nsym), key=lambda s: (ws[s], s))
    pos, code = 0, {}
    for s in order:  # canonical: weight 1 (the longest codes) first, symbols ascending within a weight
        w = ws[s]
        code[s] = format(pos >> (w - 1), "0%db" % (maxbits + 1 - w))
        pos += 1 << (w - 1)
    assert pos == 1 << maxbits
    rnd = random.Random(seed)
    pop = [nsym - 1] * 40 + [nsym - 2] * 20 + [nsym - 3] * 10 + list(range(nsym)) * 2
    ex = ws[:-1] + ([0] if (len(ws) - 1) % 2 else [])
    tree = bytes([127 + len(ws) - 1]) + bytes((ex[i] << 4) | ex[i + 1] for i in range(0, len(ex), 2))

    def section(lit, ltype, with_tree):
        n = len(lit)
        if streams == 1:
            comp = (tree if with_tree else b"") + backward_stream("".join(code[c] for c in lit))
            return (ltype | (0 << 2) | (n << 4) | (len(comp) << 14)).to_bytes(3, "little") + comp
        seg = (n + 3) // 4
        ss = [backward_stream("".join(code[c] for c in lit[k * seg:(k + 1) * seg if k < 3 else n])) for k in range(4)]
        comp = (tree if with_tree else b"") + b"".join(len(x).to_bytes(2, "little") for x in ss[:3]) + b"".join(ss)
        if n < 16384 and len(comp) < 16384:
            return (ltype | (2 << 2) | (n << 4) | (len(comp) << 18)).to_bytes(4, "little") + comp
        return (ltype | (3 << 2) | (n << 4) | (len(comp) << 22)).to_bytes(5, "little") + comp

    lit = bytes(rnd.choice(pop) for _ in range(nlit))
    body = section(lit, 2, True) + nbseq_bytes(0)
    if not treeless_second:
        return frame(block_header(1, 2, len(body)) + body, lit), lit
    lit2 = bytes(rnd.choice(pop) for _ in range(treeless_second))
    body2 = section(lit2, 3, False) + nbseq_bytes(0)
    return frame(block_header(0, 2, len(body)) + body + block_header(1, 2, len(body2)) + body2, lit + lit2), lit + lit2


def main():
    assert Z.available(), "make_golden.py needs a libzstd shared object"
    manifest["libzstd"] = Z.version()

    # ---------------------------------------------------------------- 1. the reference's own payloads
    ref_payloads = [
        b"", b"compressed data", b"overlap compressed", b"overlap plain",
        b"1st file in root", b"1st file in first", b"2nd file in first", b"1st file in second",
        b"2nd file in second", b"3rd file in second", b"1st file in third",
        b"new file content", b"truncated", b"truncated and appended",
        b"FIRST", b"SECOND", b"THIRD", b"BASIC", b"BASICAPPENDED", b"KEEP", b"IT", b"UNCONVERTED",
        b"ORIGINAL", b"OVERRIDE", b"TOO CLOSE", b"2 CLOSE", b"FH cache 1",
    ]
    for i, p in enumerate(ref_payloads):
        add("ref_bulk_%02d" % i, Z.compress_simple(p, 0), p, note="zstd::bulk::compress(%r, 0) (tests/convert.rs:15-43)" % p)
        add("ref_writer_%02d" % i, Z.compress(p, level=3, checksum=True), p,
            note="writer settings src/main.rs:781-791 on %r" % p)

    # ---------------------------------------------------------------- 2. JSON frames
    for nm, size, idx in (("json_4k", 4096, 1), ("json_128k", 131072, 2), ("json_1m", 1 << 20, 3)):
        raw = corpus.gen("json", 2, idx, size)
        add(nm, Z.compress(raw, 3, True), raw, gen=("json", 2, idx, size), note="level 3, checksum, FCS")
    raw = corpus.gen("json", 2, 4, 131072)
    for lvl in (1, 9, 19, -3):
        add("json_128k_l%s" % str(lvl).replace("-", "m"), Z.compress(raw, lvl, True), raw, gen=("json", 2, 4, 131072), note="level %d" % lvl)

    # ---------------------------------------------------------------- 3. Silesia-proxy classes (branch coverage)
    for k in ("text", "markup", "int32", "dna", "xray", "random", "repeats"):
        raw = corpus.gen(k, 3, 5, 131072)
        add("proxy_%s_128k" % k, Z.compress(raw, 3, True), raw, gen=(k, 3, 5, 131072), note="level 3")
    raw = corpus.gen("text", 3, 6, 20000)
    add("proxy_text_20k_l19", Z.compress(raw, 19, True), raw, gen=("text", 3, 6, 20000), note="level 19")
    raw = corpus.gen("dna", 3, 7, 300000)
    add("proxy_dna_300k", Z.compress(raw, 5, True), raw, gen=("dna", 3, 7, 300000), note="3 blocks")
    zeros = bytes(131072)
    add("zeros_128k", Z.compress(zeros, 3, True), zeros, gen=("zero", 0, 0, 131072), note="one sequence ml=131070 offset 1")
    big_rle = bytes([7]) * 500000
    add("rle_500k", Z.compress(big_rle, 3, True), big_rle, gen=("fill7", 0, 0, 500000), note="RLE blocks / giant overlap")

    # hand-built
    f, out = hand_rle_everything()
    add("hand_rle_lits_rle_tables", f, out, note="RLE literals; LL/OF/ML tables in RLE mode")
    f, out = hand_long_nbseq()
    add("hand_long_nbseq", f, out, note="nbSeq >= 0x7F00 (3-byte form), raw block as history")
    for (f, out), nm in zip(hand_direct_weights(), ("hand_direct_weights_1s", "hand_direct_weights_4s")):
        add(nm, f, out, note="Huffman tree as direct 4-bit weights")

    # a Huffman tree of depth 12: libzstd's own limit (accepted by 1.4.8 here and by 1.5.x: tests/test_oracle.py), one past the format text's 11
    for nm, (streams, nlit, tl) in (("hand_huf12_1s", (1, 300, 0)), ("hand_huf12_4s", (4, 1021, 0)), ("hand_huf12_4s_60k", (4, 60000, 0)),
                                    ("hand_huf12_treeless", (4, 5000, 3000)), ("hand_huf11_4s_60k", (4, 60000, 0))):
        f, out = hand_huffman_depth(11 if "huf11" in nm else 12, streams, nlit, treeless_second=tl)
        add(nm, f, out, note="Huffman tree of depth %d as direct weights%s" % (11 if "huf11" in nm else 12, "; second block treeless" if tl else ""))

    # ---------------------------------------------------------------- 4. container features
    a = corpus.gen("json", 4, 1, 5000); b = corpus.gen("text", 4, 2, 70000)
    skippable = bytes([0x5A, 0x2A, 0x4D, 0x18]) + (11).to_bytes(4, "little") + b"hello world"
    add("multi_frame_skippable", Z.compress(a, 3, True) + skippable + Z.compress(b, 3, False) + skippable, a + b,
        note="frame + skippable + frame + trailing skippable")
    raw = corpus.gen("json", 4, 3, 300000)
    add("nofcs_stream_300k", Z.compress_stream(raw, 3, True, chunk=50000, flush_each=True), raw, gen=("json", 4, 3, 300000),
        note="no FCS, window descriptor, flushed every 50 000 bytes (short blocks)")
    raw = corpus.gen("markup", 4, 4, 3 << 20)
    add("window_3m_l1", Z.compress(raw, 1, True), raw, gen=("markup", 4, 4, 3 << 20), note="3 MiB at level 1: 512 KiB window descriptor")
    raw = corpus.gen("json", 4, 5, 200000)
    add("nochecksum_200k", Z.compress(raw, 3, False), raw, gen=("json", 4, 5, 200000), note="no checksum")
    raw = corpus.gen("text", 4, 6, 600000)
    add("window_log10", Z.compress(raw, 3, True, window_log=10), raw, gen=("text", 4, 6, 600000), note="windowLog 10: 1 KiB blocks")

    # ---------------------------------------------------------------- 5. dictionary
    samples = [corpus.gen("json", 5, i, 300 + (i * 37) % 2700) for i in range(2000)]
    d = Z.train_dict(samples, 16 * 1024)
    with open(os.path.join(FR, "dict16k.bin"), "wb") as f:
        f.write(d)
    for i in range(8):
        raw = corpus.gen("json", 5, 5000 + i, 300 + (i * 331) % 2700)
        add("dict_%d" % i, Z.compress(raw, 3, True, dictionary=d), raw, gen=("json", 5, 5000 + i, len(raw)), dictionary="dict16k.bin")

    # ---------------------------------------------------------------- 6. negative set
    good = open(os.path.join(FR, "json_4k.zst"), "rb").read()
    for cut in (3, 5, 9, 12, 40, 400, len(good) - 5, len(good) - 1):
        add("bad_trunc_%d" % cut, good[:cut], expect="error")
    flipped = bytearray(good); flipped[-1] ^= 0x55
    add("bad_checksum", bytes(flipped), expect="error")
    rb = bytearray(good); rb[4] |= 0x08
    add("bad_reserved_bit", bytes(rb), expect="error")
    add("bad_magic", b"\x00\x01\x02\x03" + good[4:], expect="error")
    add("bad_trailing_garbage", good + b"\x01\x02\x03\x04\x05", expect="error")
    ob = bytearray(good); ob[200] ^= 0xFF
    add("bad_bitflip_200", bytes(ob), expect="error")
    small = Z.compress(b"x" * 10, 3, True)
    big_block = bytearray(small); hdr_at = 4 + 1 + 1  # magic, FHD, 1-byte FCS
    big_block[hdr_at:hdr_at + 3] = ((2000 << 3) | 1).to_bytes(3, "little")
    add("bad_oversize_block", bytes(big_block) + bytes(2000), expect="error")
    # offset beyond history: RLE tables, one sequence with offset 60 but only 2 bytes of output
    body = lit_header_raw_rle(0, 2) + b"ab" + nbseq_bytes(1) + bytes([(1 << 6) | (1 << 4) | (1 << 2)]) + bytes([2, 5, 0])
    body += backward_stream(bits(31, 5))
    add("bad_offset_beyond_history", frame(block_header(1, 2, len(body)) + body, b"ab" + b"x" * 3, checksum=False), expect="error")
    # treeless literals in a frame's SECOND block when no block before it carried a tree: libzstd's dictionary_corrupted, found before
    # the literals section's sizes and before the sequence section are looked at (a block task learns what the frame has inherited
    # only from its predecessor).  a: the rest of the block is sound; b: its sizes are impossible; c: its sequence section is garbage.
    b0 = lit_header_raw_rle(0, 6) + b"abcdef" + nbseq_bytes(0)
    for tag, lh, tail in (("a", (3 | (0 << 2) | (8 << 4) | (5 << 14)).to_bytes(3, "little"), bytes([0x11, 0x22, 0x33, 0x44, 0x81]) + nbseq_bytes(0)),
                          ("b", (3 | (0 << 2) | (8 << 4) | (900 << 14)).to_bytes(3, "little"), bytes([0x11, 0x22, 0x33, 0x44, 0x81]) + nbseq_bytes(0)),
                          ("c", (3 | (0 << 2) | (8 << 4) | (5 << 14)).to_bytes(3, "little"), bytes([0x11, 0x22, 0x33, 0x44, 0x81]) + nbseq_bytes(3) + bytes([0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x00]))):
        b1 = lh + tail
        add("bad_treeless_no_tree_block2_" + tag, frame(block_header(0, 2, len(b0)) + b0 + block_header(1, 2, len(b1)) + b1, b"abcdef" + b"x" * 58, checksum=False), expect="error")  # (a content size above every block's size)
    body = lit_header_raw_rle(0, 2) + b"ab" + nbseq_bytes(1) + bytes([(1 << 6) | (1 << 4) | (1 << 2) | 1]) + bytes([2, 5, 0]) + backward_stream(bits(0, 5))
    add("bad_seq_modes_reserved", frame(block_header(1, 2, len(body)) + body, b"ab" + b"x" * 3, checksum=False), expect="error")
    body = lit_header_raw_rle(0, 2) + b"ab" + nbseq_bytes(1) + bytes([(3 << 6)]) + backward_stream("0" * 11)
    add("bad_repeat_without_table", frame(block_header(1, 2, len(body)) + body, b"ab" + b"x" * 3, checksum=False), expect="error")

    f13, _ = hand_huffman_depth(13, 4, 1021)
    add("bad_huf13", f13, expect="error", note="Huffman tree of depth 13: past libzstd's HUF_TABLELOG_MAX")

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(FR, x)) for x in os.listdir(FR))
    print("wrote", len(manifest["vectors"]), "vectors,", tot, "bytes under", FR)


if __name__ == "__main__":
    main()
