"""Spill audit of the device code: per kernel the register counts the compiler reports and every spill
instruction by loop depth.  Runs here (no GPU): compiles each .hip to gfx950 assembly and reads it.

  python tools/spill_audit.py [--loops]  # table on stdout (markdown); --loops: where the parked scalar registers are re-read

A loop is a backward branch (s_cbranch_* / s_branch to a label defined above it); an instruction's depth is the number of such
[label, branch] intervals that contain it.  Spill instructions: scratch_load / scratch_store (lane registers to memory) and
v_writelane_b32 / v_readlane_b32 pairs the compiler makes to park scalar registers in lanes (a lane register counts as such a parking place when nothing but v_writelane writes it and nothing but v_readlane reads
it; the source's own readlane / writelane traffic -- rdlane(), the hand-written scalar loops -- is not a spill and is left out)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "compu_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def functions(asm):
    """name -> list of lines of the function body"""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L") and cur is None and ("@function" in asm[max(0, asm.find(line) - 200):asm.find(line)] or True):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.strip().startswith(".end_amdhsa_kernel") or re.match(r"^\s*\.size\s+" + re.escape(name or "") + r"\b", line) or line.startswith(".Lfunc_end"):
                out[name] = cur
                cur, name = None, None
                continue
            cur.append(line)
    return out


def audit(body):
    labels, insts = {}, []
    for line in body:
        t = line.strip()
        m = re.match(r"^(\.LBB\w+|\.Ltmp\w+|\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        insts.append(t)
    heads = {}
    for i, t in enumerate(insts):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            a = labels[m.group(1)]
            heads[a] = max(heads.get(a, a), i)  # one loop per header: from the label to its last backward branch
    loops = sorted(heads.items())
    # the compiler's own lane-parking registers: written by v_writelane only and read by v_readlane only
    wl_regs, other = set(), set()
    for t in insts:
        ops = re.findall(r"\bv(\d+)\b", t) + [str(k) for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t) for k in range(int(a), int(b) + 1)]
        if t.startswith("v_writelane_b32"):
            wl_regs.add(re.match(r"v_writelane_b32\s+v(\d+)", t).group(1))
        elif t.startswith("v_readlane_b32"):
            pass
        else:
            other.update(ops)
    park = wl_regs - other
    res = {}
    for i, t in enumerate(insts):
        kind = None
        if t.startswith(("scratch_load", "scratch_store")):
            kind = "scratch"
        elif t.startswith("v_writelane_b32") and re.match(r"v_writelane_b32\s+v(\d+)", t).group(1) in park:
            kind = "writelane"
        elif t.startswith("v_readlane_b32") and re.match(r"v_readlane_b32\s+\S+\s+v(\d+)", t) and re.match(r"v_readlane_b32\s+\S+\s+v(\d+)", t).group(1) in park:
            kind = "readlane"
        if kind:
            d = sum(1 for a, b in loops if a <= i <= b)
            res.setdefault(kind, {}).setdefault(d, 0)
            res[kind][d] += 1
    # per loop: parked-register traffic that belongs to the loop itself (not to a loop nested in it)
    def parked(t):
        m = re.match(r"v_readlane_b32\s+\S+\s+v(\d+)", t)
        if m and m.group(1) in park:
            return "r"
        m = re.match(r"v_writelane_b32\s+v(\d+)", t)
        return "w" if m and m.group(1) in park else None

    per_loop = []
    for a, b in loops:
        inner = [(c, d) for c, d in loops if a <= c and d <= b and (c, d) != (a, b)]
        own = [parked(t) for i, t in enumerate(insts[a:b + 1]) if not any(c <= a + i <= d for c, d in inner)]
        r, w = own.count("r"), own.count("w")
        if r or w:
            per_loop.append((b - a + 1, len(inner), r, w))
    audit.last_loops = per_loop
    return res, len(insts), len(loops)


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for f in sorted(os.listdir(SRC)):
            if f.endswith(".hip"):
                out = os.path.join(tmp, f[:-4] + ".s")
                procs.append((f, out, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, f)],
                                                       stderr=subprocess.DEVNULL)))
        for f, out, p in procs:
            if p.wait() != 0:
                print("compile failed:", f, file=sys.stderr)
                continue
            asm = open(out).read()
            meta = {}
            for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", asm, re.S):
                meta[m.group(2)] = dict(agpr=int(m.group(1)), sgpr=int(m.group(3)), sspill=int(m.group(4)), vgpr=int(m.group(5)), vspill=int(m.group(6)))
            lds = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+)", asm) for _ in ()}  # (kept simple: LDS is in DESIGN)
            for name, body in functions(asm).items():
                if name not in meta:
                    continue
                res, n, nl = audit(body)
                short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
                rows.append((f, short, meta[name], res, n, nl, list(audit.last_loops)))

    def fmt(d):
        return ", ".join(f"depth {k}: {v}" for k, v in sorted(d.items())) if d else "none"

    print("| file | kernel | VGPR (+AGPR) | SGPR | lane spills to scratch (vgpr_spill_count) | scalar registers parked in lanes (sgpr_spill_count) | `v_writelane` by loop depth | `v_readlane` by loop depth | scratch by loop depth | instructions / loops |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for f, short, m, res, n, nl, _ in rows:
        print(f"| `{f}` | `{short}` | {m['vgpr']} (+{m['agpr']}) | {m['sgpr']} | {m['vspill']} | {m['sspill']} | {fmt(res.get('writelane', {}))} | {fmt(res.get('readlane', {}))} | {fmt(res.get('scratch', {}))} | {n} / {nl} |")
    if "--loops" in sys.argv:
        print()
        print("Parked scalar registers by loop (loop length in instructions, loops nested in it, re-reads and writes that belong to the loop itself):")
        for f, short, m, res, n, nl, per_loop in rows:
            if per_loop:
                print(f"* `{short}`: " + "; ".join(f"{ln} instr / {ni} nested: {r} r + {w} w" for ln, ni, r, w in sorted(per_loop)))


if __name__ == "__main__":
    main()
