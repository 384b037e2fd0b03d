"""The C ABI: every symbol include/compu_hip.h declares is exported by the built library (no GPU needed),
Detection::detect on the host, and -- on the GPU box -- the C++ replay of the reference's tests."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "compu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chip_[a-z0-9_]+)\s*\(", text)) - {"chip_malloc_fn", "chip_free_fn"})


def test_library_exports_every_declared_symbol():
    import compu_amd

    lib = compu_amd.lib()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/compu_hip.h but not exported"
    assert lib.chip_version().startswith(b"compu-hip")
    # library-level argument checks need no device
    assert lib.chip_decode_batch(12345, 1, None, None, None, None, None, None, None, None, None, None) == -101
    # unknown option bits of the _ex entry point are refused before anything else is looked at
    assert lib.chip_decode_batch_ex(-15, 2, 1, None, None, None, None, None, None, None, None, None, None) == -101
    assert lib.chip_decode_batch_ex(-15, 0x80000001, 0, None, None, None, None, None, None, None, None, None, None) == -101
    assert lib.chip_decode_batch_ex(-15, 1, 0, None, None, None, None, None, None, None, None, None, None) == 0  # an empty batch is fine
    assert lib.chip_encode_bound(31, 65536) >= 65536 + 18


def _c_prototypes():
    """{name: number of parameters} of every function prototype in include/compu_hip.h."""
    text = open(os.path.join(ROOT, "include", "compu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(chip_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        if name in ("chip_malloc_fn", "chip_free_fn"):
            continue
        protos[name] = 0 if args in ("", "void") else args.count(",") + 1
    return protos


def _rust_externs():
    """{name: number of parameters} of every `pub fn` in the extern block of integration/src/hip_sys.rs."""
    text = open(os.path.join(ROOT, "integration", "src", "hip_sys.rs")).read()
    text = re.sub(r"//[^\n]*", " ", text)
    block = text[text.index('extern "C" {'):]
    ext = {}
    for m in re.finditer(r"pub fn (chip_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", block, flags=re.S):
        args = m.group(2).strip()
        # a parameter is `name: type`; commas inside `Option<fn(..)>` types do not occur in this file's extern block
        ext[m.group(1)] = 0 if not args else len([a for a in args.split(",") if ":" in a])
    return ext


def test_rust_bindings_cover_the_header():
    """The boundary a compu maintainer compiles against: every prototype of include/compu_hip.h has an `extern "C"` line in
    integration/src/hip_sys.rs with the same number of parameters, every extern exists in the header, and the library exports
    them all (a text check: this image has no Rust toolchain)."""
    import compu_amd

    protos, ext = _c_prototypes(), _rust_externs()
    assert len(protos) >= 30
    missing = sorted(set(protos) - set(ext))
    assert not missing, f"declared in include/compu_hip.h but not bound in hip_sys.rs: {missing}"
    stray = sorted(set(ext) - set(protos))
    assert not stray, f"bound in hip_sys.rs but not declared in include/compu_hip.h: {stray}"
    for name, n in protos.items():
        assert ext[name] == n, f"{name}: {n} parameters in the header, {ext[name]} in hip_sys.rs"
    lib = compu_amd.lib()
    for name in ext:
        assert hasattr(lib, name), f"{name} bound in hip_sys.rs but not exported by the library"
    # the hot path has safe faces next to the vtables
    dec = open(os.path.join(ROOT, "integration", "src", "decoder", "hip.rs")).read()
    enc = open(os.path.join(ROOT, "integration", "src", "encoder", "hip.rs")).read()
    for fn in ("decode_batch_host", "decode_batch_multi", "decode_batch_device", "detect_batch_device", "partition_units"):
        assert f"pub fn {fn}" in dec or f"pub unsafe fn {fn}" in dec, fn
    for fn in ("encode_batch_host", "encode_batch_device", "encode_bound"):
        assert f"pub fn {fn}" in enc or f"pub unsafe fn {fn}" in enc, fn


def test_encode_bound_covers_what_the_oracle_encoder_writes():
    """chip_encode_bound is the capacity that is always enough: worst cases of every level (incompressible input: stored
    blocks at level 0 / 1, a stored block per 65 472 tokens at the dynamic levels) through the oracle encoder."""
    import compu_amd
    from oracle import oracle as O

    lib = compu_amd.lib()
    rnd = os.urandom(400000)
    for mode in (O.MODE_DEFLATE, O.MODE_ZLIB, O.MODE_GZIP):
        for n in (0, 1, 63, 64, 65471, 65472, 65473, 65535, 65536, 131071, 200000, 400000):
            for level in (0, 1, 3, 6):
                e = O.DeflateEncoder(mode, level, 0)
                comp, ir, orr, st = e.encode(rnd[:n], n + 4096, O.OP_FINISH)
                assert st == O.ENC_FINISHED and len(comp) <= lib.chip_encode_bound(mode, n), (mode, n, level, len(comp))


def test_detection_table_matches_reference():
    """src/decoder/mod.rs:28-114 incl. the None cases (:97-104) and the 0x68 quirk (:80-82)"""
    import compu_amd
    from compu_amd import Detection

    assert Detection.detect(b"") is None and Detection.detect(b"\x1f") is None
    assert Detection.detect(b"\x1f\x8b") == Detection.Gzip
    assert Detection.detect(b"\x28\xb5\x2f") is None  # 3 bytes: not enough for the zstd magic
    assert Detection.detect(b"\x28\xb5\x2f\xfd") == Detection.Zstd
    assert Detection.detect(b"abcd") == Detection.Unknown and Detection.detect(b"ab") is None
    table = {0x08: (0x1D, 0x5B, 0x99, 0xD7), 0x18: (0x19, 0x57, 0x95, 0xD3), 0x28: (0x15, 0x53, 0x91, 0xCF), 0x38: (0x11, 0x4F, 0x8D, 0xCB),
             0x48: (0x0D, 0x4B, 0x89, 0xC7), 0x58: (0x09, 0x47, 0x85, 0xC3), 0x78: (0x01, 0x5E, 0x9C, 0xDA)}
    for cmf, flgs in table.items():
        for flg in flgs:
            assert Detection.detect(bytes([cmf, flg])) == Detection.Zlib
    for flg in (0x05, 0x43, 0x81, 0xDE):  # the 0x68 arm falls through in the reference
        assert Detection.detect(bytes([0x68, flg])) is None
        assert Detection.detect(bytes([0x68, flg, 0, 0])) == Detection.Unknown
    assert Detection.detect(b"\x78\x9d") is None  # right CMF, FLG not in the table


@pytest.mark.gpu
def test_cpp_replay_of_reference_tests():
    exe = os.path.join(ROOT, "tests", "cpp", "test_reference")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "compu_amd", "csrc", "build.sh")])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test result: ok" in out.stdout
