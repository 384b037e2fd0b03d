"""Writes tests/golden/encoder_digests.json: SHA-256 and size of what the encoder algorithm of oracle/oracle_deflate.c
produces for the reference's two fixtures (gzip mode) at levels 0, 1, 3, 6 and for the Rle / HuffmanOnly / Fixed strategies.
The file pins the algorithm itself: a change of the match finder, the code construction or the block choice -- in the
oracle and, through the byte-for-byte GPU tests, in the kernels -- shows up as a digest change.
    python tools/make_encoder_digests.py"""
import hashlib
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

out = {}
for name in ("10x10y", "alice29.txt"):
    data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    for level, strategy in ((0, 0), (1, 0), (3, 0), (6, 0), (6, 2), (6, 3), (6, 4)):
        e = O.DeflateEncoder(O.MODE_GZIP, level, strategy)
        comp, ir, orr, st = e.encode(data, len(data) + 4096, O.OP_FINISH)
        assert st == O.ENC_FINISHED and zlib.decompress(comp, 31) == data
        out[f"{name}:gzip:level{level}:strategy{strategy}"] = {"size": len(comp), "sha256": hashlib.sha256(comp).hexdigest()}
with open(os.path.join(ROOT, "tests", "golden", "encoder_digests.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps({k: v["size"] for k, v in out.items()}, indent=1))
