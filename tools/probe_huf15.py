"""What libzstd 1.5.x does with Huffman literal streams that are not consumed exactly, against the oracle's rule 1 (oracle/zstd_oracle.c).
Parent: builds checksum-less frames with the machine's libzstd and mutates bytes inside their literal sections; child (a process of its own:
two libzstd versions in one address space interpose each other) decodes every mutant with the 1.5.x that pillow's wheel ships and with the
oracle and prints the classes.  python tools/probe_huf15.py [n_mutants_per_frame]"""
import os, pickle, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import ctypes as C, glob, os, sys, sysconfig, pickle, collections
sys.path.insert(0, os.environ["MZD_ROOT"])
import oracle
L = None
roots = {sysconfig.get_paths().get("purelib", ""), sysconfig.get_paths().get("platlib", ""), "/usr/local/lib/python3.10/dist-packages"}
for r in roots:
    for p in sorted(glob.glob(r + "/pillow.libs/libzstd*.so*")):
        try:
            cand = C.CDLL(p); cand.ZSTD_versionString.restype = C.c_char_p
            if cand.ZSTD_versionString().decode().startswith("1.5."): L = cand
        except OSError: pass
if L is None: print("SKIP"); sys.exit(0)
L.ZSTD_decompressDCtx.restype = C.c_size_t
L.ZSTD_decompressDCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
L.ZSTD_isError.argtypes = [C.c_size_t]; L.ZSTD_getErrorCode.argtypes = [C.c_size_t]
L.ZSTD_createDCtx.restype = C.c_void_p
dctx = L.ZSTD_createDCtx()
cases = pickle.load(open(os.environ["MZD_CASES"], "rb"))
n = collections.Counter(); ex = collections.defaultdict(list)
for name, m, cap in cases:
    buf = C.create_string_buffer(cap + 1)
    r = L.ZSTD_decompressDCtx(dctx, buf, cap + 1, m, len(m))
    zerr = bool(L.ZSTD_isError(r)); zout = None if zerr else buf.raw[:r]
    rc, out = oracle.decode(m, cap=cap + 1)
    len_, over, inex, thr = oracle.last_verdict_lit_lenient(), oracle.last_verdict_lit_over(), oracle.last_verdict_lit_inexact(), oracle.last_verdict_lit_through()
    tag = (" (leftover bits ignored)" if len_ else "") + (" (read through)" if thr else "")
    if rc == 0 and not zerr: k = ("both accept, bytes equal" if out == zout else "BOTH ACCEPT, BYTES DIFFER") + tag
    elif rc == 0: k = "ORACLE ACCEPTS, 1.5 REJECTS" + tag
    elif not zerr: k = "1.5 accepts, oracle rejects: " + ("stream ran out (lit_over)" if over else ("inexact, checked loop" if inex else "OTHER rc=%d" % rc))
    else: k = "both reject" + (" (lit_over)" if over else (" (inexact, checked loop)" if inex else ""))
    n[k] += 1
    if len(ex[k]) < 3: ex[k].append(name)
print(L.ZSTD_versionString().decode())
for k in sorted(n): print("%6d  %s   e.g. %s" % (n[k], k, ex[k]))
'''

def main():
    import numpy as np
    import corpus, oracle
    Z = oracle.LibZstd
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.RandomState(7)
    cases = []
    for kind, size, level in (("xray", 60000, 3), ("json", 131072, 3), ("text", 100000, 3), ("int32", 131072, 3), ("json", 4096, 3), ("text", 2000, 19), ("dna", 50000, 3), ("json", 700, 3), ("xml", 20000, 1)):
        if kind not in corpus.KINDS: continue
        raw = corpus.gen(kind, 31, 1, size)
        comp = bytearray(Z.compress(raw, level, False))
        rc, out, blocks = oracle.decode(bytes(comp), cap=size, want_trace=True)
        assert rc == 0
        for k in range(per):
            m = bytearray(comp)
            pos = int(rng.randint(6, len(m)))
            m[pos] ^= int(rng.choice([1, 0x80, 0xFF, int(rng.randint(1, 256))]))
            for _ in range(int(os.environ.get("PROBE_EXTRA", "0"))):  # deeper corruption: more bytes of the same neighbourhood
                q = min(len(m) - 1, pos + int(rng.randint(0, 2000)))
                m[q] ^= int(rng.randint(1, 256))
            cases.append(("%s-%d-l%d@%d" % (kind, size, level, pos), bytes(m), size))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "cases.pkl")
        open(path, "wb").write(pickle.dumps(cases))
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MZD_ROOT=ROOT, MZD_CASES=path), capture_output=True, text=True)
        print(r.stdout, r.stderr[-3000:])

if __name__ == "__main__":
    main()
