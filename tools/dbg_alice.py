import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import compu_amd
g = os.path.join(ROOT, "tests", "golden")
for name in ("10x10y", "alice29.txt"):
    comp = open(os.path.join(g, name + ".compressed.gz"), "rb").read()
    data = open(os.path.join(g, name), "rb").read()
    dec = compu_amd.decoder_interface.zlib_hip(compu_amd.ZlibMode.Gzip)
    out = bytearray(len(data) + 64)
    r = dec.decode(comp, out)
    got = bytes(out[: len(out) - r.output_remain])
    print(name, r, len(got), len(data))
    if got != data:
        n = min(len(got), len(data))
        idx = next((i for i in range(n) if got[i] != data[i]), n)
        bad = sum(1 for i in range(n) if got[i] != data[i])
        print(" first mismatch at", idx, "mismatching bytes", bad)
        print(" want", data[max(0, idx - 40): idx + 40])
        print(" got ", got[max(0, idx - 40): idx + 40])
