"""The load-bearing build rule, checked where it cannot pass vacuously: every HIP source is compiled to gfx950 DEVICE
ASSEMBLY with the library's own flags (physdock_amd/build.py compile_cmd) and the text is searched.

* no packed fp32 VALU instruction (v_pk_add / mul / fma_f32) anywhere: on MI355X such an op that reads a register pair a
  global load has just returned intermittently sees stale lanes 48-63 when a second kernel stream shares the CUs
  (NOTES.md "Packed fp32 VALU on freshly loaded registers"; reproducer tools/micro/pk_f32_hazard.hip);
* the check saw real device code: the matrix kernels contain the MFMA opcodes they are documented to use.
Runs on the build container (hipcc cross-compiles without a GPU)."""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

from physdock_amd import build

#: sources that must contain matrix instructions, with the opcode families DESIGN.md says they use
MFMA_EXPECTED = {
    "attn_split.hip": ("v_mfma_f32_32x32x16_bf16",),
    "gemm_split.hip": ("v_mfma_f32_32x32x16_bf16",),
    "attention.hip": ("v_mfma_f32_32x32x2_f32",),
    "gemm_stream.hip": ("v_mfma_f32_32x32x2_f32",),
    "gemm.hip": ("v_mfma_f32_32x32x2_f32",),
    "pool.hip": ("v_mfma_f32_32x32x16_bf16",),
    "gemm_f16.hip": ("v_mfma_f32_32x32x16_f16",),
    "attn_pipe.hip": ("v_mfma_f32_32x32x16_f16",),
}
PACKED = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")


@pytest.fixture(scope="module")
def device_asm():
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as d:
        def one(src):
            out = os.path.join(d, os.path.basename(src)[:-4] + ".s")
            r = subprocess.run(build.compile_cmd(src, out, mode=("-S", "--cuda-device-only")), capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            with open(out) as f:
                return os.path.basename(src), f.read()
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
            yield dict(ex.map(one, build.sources()))


def test_device_assembly_was_really_produced(device_asm):
    assert set(MFMA_EXPECTED) <= set(device_asm)
    for name, text in device_asm.items():
        assert ".amdgcn_target" in text and "gfx950" in text, name
    kernels = {name: text.count(".amdhsa_kernel") for name, text in device_asm.items()}
    assert all(kernels[n] >= 1 for n in device_asm if n != "api.hip"), kernels       # api.hip holds host entry points only
    assert sum(kernels.values()) >= 100, kernels
    for name, ops in MFMA_EXPECTED.items():
        for op in ops:
            assert device_asm[name].count(op) > 0, f"{name}: expected {op} in the device code"


def test_no_packed_f32_valu_anywhere(device_asm):
    hits = {name: len(PACKED.findall(text)) for name, text in device_asm.items()}
    assert sum(hits.values()) == 0, f"packed fp32 VALU instructions found: { {k: v for k, v in hits.items() if v} }"


def test_the_rule_is_not_vacuous():
    """without the flags hipcc DOES emit packed fp32 ops for this code base (so the test above checks something)"""
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    src = os.path.join(build.CSRC, "norm.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "norm.s")
        cmd = [c for c in build.compile_cmd(src, out, mode=("-S", "--cuda-device-only"))]
        for flag in build.NO_PACKED_F32:
            cmd.remove(flag)
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        with open(out) as f:
            assert len(PACKED.findall(f.read())) > 0
