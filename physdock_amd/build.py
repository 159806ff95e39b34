"""Build libphysdock_hip.so (hipcc, gfx950 only) in-tree next to this file."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libphysdock_hip.so")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return any(os.path.getmtime(d) > t for d in deps)


#: per-source extra flags.  attention.hip: the SLP vectoriser packs the softmax's f32 adds/muls into v_pk_*_f32,
#: which issue slower beside MFMAs on gfx950 (cdna guide: packed f32 VALU is an anti-lever next to MFMA).
#: gemm_split.hip: with SLP vectorisation hipcc (ROCm 7.2) packs the norm prologue into v_pk_add/mul/fma_f32 on a
#: register pair that a global_load_dwordx2 has just returned; on MI355X the kernel then intermittently stages wrong A
#: rows for lanes 48-63 of a wave (the last VALU pass) in the first staging after a tile change - seen only in the
#: multi-tile 32x64 / 64x64 wave layouts, gone with any perturbation of the schedule, and gone in every shape / 6 x
#: repetition of tools/diag_split3.py without the packed ops (NOTES.md).  Scalar f32 VALU is also what the CDNA guide
#: recommends beside MFMAs.  tests/test_gemm_split_gpu.py::test_split_multi_tile_stress guards it.
EXTRA_FLAGS = {"gemm_split.hip": ["-fno-slp-vectorize"], "attn_split.hip": ["-fno-slp-vectorize"]}


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
               "-Wno-unused-result", "-I", INCLUDE, "-I", CSRC, "-c", src, "-o", obj]
        if os.environ.get("PD_LAB"):          # lab build: in-kernel phase traces + getenv tuning overrides (never shipped)
            cmd[1:1] = ["-DPD_LAB=1"]
        if os.environ.get("PD_BK") and os.path.basename(src) == "gemm.hip":
            cmd[1:1] = ["-DPD_BK=" + os.environ["PD_BK"]]
        if os.environ.get("PD_STREAM_NOEMIT") and os.path.basename(src) == "gemm_stream.hip":
            cmd[1:1] = ["-DPD_STREAM_NOEMIT=1"]
        if os.environ.get("PD_STREAM_SAMETILE") and os.path.basename(src) == "gemm_stream.hip":
            cmd[1:1] = ["-DPD_STREAM_SAMETILE=1"]
        if os.environ.get("PD_GEMM_NOSTORE") and os.path.basename(src) == "gemm.hip":
            cmd[1:1] = ["-DPD_GEMM_NOSTORE=1"]
        if os.environ.get("PD_ABL") and os.path.basename(src) == "gemm_split.hip":      # lab: main-loop ablations (wrong results)
            cmd[1:1] = ["-DPD_ABL=" + os.environ["PD_ABL"]]
        for knob in ("PD_PB_UB", "PD_PB_TR", "PD_PB_TR4"):
            if os.environ.get(knob) and os.path.basename(src) == "pairbias.hip":
                cmd[1:1] = [f"-D{knob}=" + os.environ[knob]]
        if os.environ.get("PD_KSPLIT_MAX_BYTES") and os.path.basename(src) == "gemm_stream.hip":
            cmd[1:1] = ["-DPD_KSPLIT_MAX_BYTES=" + os.environ["PD_KSPLIT_MAX_BYTES"]]
        for knob in ("PD_SPLIT_MIN_TILES", "PD_SPLIT_MIN_TILES_SMALL"):
            if os.environ.get(knob) and os.path.basename(src) == "gemm_split.hip":
                cmd[1:1] = [f"-D{knob}=" + os.environ[knob]]
        if os.environ.get("PD_NO_EXTRA_FLAGS") != "1":
            cmd[1:1] = EXTRA_FLAGS.get(os.path.basename(src), [])
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
