"""Loader for tests/golden (made by tests/golden/make_golden.py).  Expected outputs are stored
inline, as a .raw file, or as (generator, cfg, index, size) to regenerate with corpus.gen."""
import base64
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
FRAMES = os.path.join(GOLD, "frames")


class Vector:
    def __init__(self, e):
        self.e = e
        self.name = e["name"]
        self.ok = e["expect"] == "ok"
        self.comp = open(os.path.join(FRAMES, self.name + ".zst"), "rb").read()
        self.dict = open(os.path.join(FRAMES, e["dict"]), "rb").read() if e.get("dict") else None
        self.blocks = e.get("blocks", [])
        self.out_len = e.get("out_len")
        self.out_xxh64 = int(e["out_xxh64"], 16) if self.ok else None
        self.oracle_class = e.get("oracle_class")

    def expected(self):
        e = self.e
        if "out_b64" in e:
            return base64.b64decode(e["out_b64"])
        if "out_file" in e:
            return open(os.path.join(FRAMES, e["out_file"]), "rb").read()
        kind, cfg, idx, size = e["gen"]
        if kind == "zero":
            return bytes(size)
        if kind == "fill7":
            return bytes([7]) * size
        import corpus
        return corpus.gen(kind, cfg, idx, size)

    def __repr__(self):
        return "Vector(%s)" % self.name


def load_manifest():
    m = json.load(open(os.path.join(GOLD, "manifest.json")))
    return [Vector(e) for e in m["vectors"]]
