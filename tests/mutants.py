"""Seeded mutants shared by the CPU pin against libzstd 1.5 (tests/test_oracle.py) and the GPU parity test of the same inputs
(tests/test_gpu_parity.py)."""
import corpus
import oracle

# Frames without a checksum (kind, size, level, extra mutated bytes per mutant): what libzstd 1.5 leaves to the checksum it ACCEPTS here.
NOCHK_FRAMES = (("xray", 60000, 3, 0), ("xray", 60000, 3, 6), ("json", 131072, 3, 0), ("text", 100000, 3, 2), ("text", 2000, 19, 0), ("int32", 131072, 3, 1), ("json", 700, 3, 0))


def nochk_mutants(kind, size, level, extra, count, rng):
    """`count` mutants of one checksum-less frame: a byte somewhere behind the frame header, and `extra` more within 2 000 bytes of it."""
    comp = bytearray(oracle.LibZstd.compress(corpus.gen(kind, 31, 1, size), level, False))
    out = []
    for _ in range(count):
        m = bytearray(comp)
        pos = int(rng.randint(6, len(m)))
        m[pos] ^= int(rng.choice([1, 0x80, 0xFF, int(rng.randint(1, 256))]))
        for _ in range(extra):
            m[min(len(m) - 1, pos + int(rng.randint(0, 2000)))] ^= int(rng.randint(1, 256))
        out.append(("nochk-%s-%d-l%d+%d" % (kind, size, level, extra), bytes(m), size))
    return out
