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


#: No packed-fp32 VALU anywhere (v_pk_add / mul / fma_f32).  hipcc (ROCm 7.2) forms them both in the SLP vectoriser and when it
#: lowers explicit float2 / float4 arithmetic; on MI355X a packed op that reads a register pair a global_load has just returned
#: intermittently sees stale data in lanes 48-63 (the last pass of the wave) - whole output rows with (row % 8) in {6, 7} off
#: by O(1).  Timing decides: on an otherwise idle GPU the fp32 kernels never showed it (every parity test passed bit-
#: reproducibly), the first split GEMM did in multi-tile blocks (NOTES.md "staging hazard"), and with a second kernel stream on
#: the same CUs the generic GEMM with a norm prologue did in 1 of 2 launches (tools/concurrent_gemm_stress.py KIND=normproj:
#: 146 of 300 launches wrong; 0 of 300 with this flag set).  -fno-slp-vectorize alone leaves the ops that come from vector
#: types; disabling the target feature removes them all (the host pass prints "not a recognized feature", harmless).
#: Scalar f32 VALU is also what the CDNA guide recommends beside MFMAs.
#: tests/test_gemm_split_gpu.py::test_split_multi_tile_stress and tests/test_concurrent_streams_gpu.py guard it.
NO_PACKED_F32 = ["-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
#: attn_pipe.hip: fmaxf without the sNaN-quieting v_max_f32 x, x in front of every operand (no NaN can enter the running maximum:
#: scores are finite or -inf); infinities stay honoured (-inf masks, the initial maximum)
EXTRA_FLAGS = {"attn_pipe.hip": ["-fno-honor-nans"]}


def compile_cmd(src, out, mode=("-c",)):
    """the hipcc command line of one source file (shared by build() and the device-assembly check in
    tests/test_build_rules_cpu.py, which passes mode=("-S", "--cuda-device-only"))"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = os.path.basename(src)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
           "-Wno-unused-result", "-I", INCLUDE, "-I", CSRC, *mode, src, "-o", out]
    if os.environ.get("PD_NO_EXTRA_FLAGS") != "1":
        cmd[1:1] = NO_PACKED_F32
    if os.environ.get("PD_SPLIT2H_PLAIN"):   # lab: the two-part split without v_fma_mix (A/B of the instruction count)
        cmd[1:1] = ["-DPD_SPLIT2H_PLAIN=1"]
    if os.environ.get("PD_LAB"):          # lab build: in-kernel phase traces + getenv tuning overrides (never shipped)
        cmd[1:1] = ["-DPD_LAB=1"]
    if os.environ.get("PD_BK") and base == "gemm.hip":
        cmd[1:1] = ["-DPD_BK=" + os.environ["PD_BK"]]
    if os.environ.get("PD_STREAM_NOEMIT") and base == "gemm_stream.hip":
        cmd[1:1] = ["-DPD_STREAM_NOEMIT=1"]
    if os.environ.get("PD_STREAM_SAMETILE") and base == "gemm_stream.hip":
        cmd[1:1] = ["-DPD_STREAM_SAMETILE=1"]
    if os.environ.get("PD_GEMM_NOSTORE") and base == "gemm.hip":
        cmd[1:1] = ["-DPD_GEMM_NOSTORE=1"]
    if os.environ.get("PD_ABL") and base == "gemm_split.hip":      # lab: main-loop ablations (wrong results)
        cmd[1:1] = ["-DPD_ABL=" + os.environ["PD_ABL"]]
    for knob in ("PD_PB_UB", "PD_PB_TR", "PD_PB_TR4"):
        if os.environ.get(knob) and base == "pairbias.hip":
            cmd[1:1] = [f"-D{knob}=" + os.environ[knob]]
    if os.environ.get("PD_KSPLIT_MAX_BYTES") and base == "gemm_stream.hip":
        cmd[1:1] = ["-DPD_KSPLIT_MAX_BYTES=" + os.environ["PD_KSPLIT_MAX_BYTES"]]
    for knob in ("PD_F16_GLU_LDSW", "PD_F16_MIN_TILES", "PD_F16_MIN_TILES_SMALL", "PD_F16_MIN_TILES_SMALL_LONGK", "PD_F16_ABL", "PD_F16_T256", "PD_F16_T256_BPC", "PD_F16_ROWS_MIN_TILES", "PD_F16_ROWS_MIN_TILES64", "PD_F16_ROWS_NO_XPF", "PD_F16_ROWS_GIVEN_STATS", "PD_F16_WROWS_MIN_TILES", "PD_F16_WROWS_MIN_TILES_SPLIT", "PD_F16_WROWS_MAX_SPLIT", "PD_F16_WROWS_TINY", "PD_F16_WROWS_MIN_ITEMS", "PD_F16_WROWS_A2", "PD_F16_WCHUNK", "PD_F16_WROWS_GLU12", "PD_F16_WROWS_12", "PD_F16_WROWS_KS", "PD_F16_WROWS_PLAIN", "PD_F16_WROWS_PLAIN_MAX_TILES", "PD_F16_WROWS_ROUNDS"):
        if os.environ.get(knob) and base == "gemm_f16.hip":
            cmd[1:1] = [f"-D{knob}=" + os.environ[knob]]
    if os.environ.get("PD_ATTN_NOSPLIT_BLOCKS") and base == "attention.hip":     # lab: block count from which a launch is not key-split
        cmd[1:1] = ["-DPD_ATTN_NOSPLIT_BLOCKS=" + os.environ["PD_ATTN_NOSPLIT_BLOCKS"]]
    if os.environ.get("PD_ATTN_TAIL") and base == "attention.hip":     # lab: 0 = no key-split tail round
        cmd[1:1] = ["-DPD_ATTN_TAIL=" + os.environ["PD_ATTN_TAIL"]]
    if os.environ.get("PD_ATTN_MIN_WAVES") and base == "attention.hip":     # lab: query waves from which the split-operand kernels take a launch
        cmd[1:1] = ["-DPD_ATTN_MIN_WAVES=" + os.environ["PD_ATTN_MIN_WAVES"]]
    if os.environ.get("PD_ATTN_LAZY") and base == "attn_f16.hip":     # lab: lazy rescale of the attention accumulator (threshold in log2 units)
        cmd[1:1] = ["-DPD_ATTN_LAZY=" + os.environ["PD_ATTN_LAZY"]]
    if os.environ.get("PD_ATTN_ABL") and base == "attn_f16.hip":      # lab: VALU ablations of the fp16-parts attention (wrong results)
        cmd[1:1] = ["-DPD_ATTN_ABL=" + os.environ["PD_ATTN_ABL"]]
    if os.environ.get("PD_PIPE_LAZY") and base == "attn_pipe.hip":     # lab: threshold of the lazy running maximum (0: plain update)
        cmd[1:1] = ["-DPD_PIPE_LAZY=" + os.environ["PD_PIPE_LAZY"]]
    if os.environ.get("PD_PIPE_NW4_MAXNK") and base == "attn_pipe.hip":      # lab: key count up to which 128-query blocks of four waves run
        cmd[1:1] = ["-DPD_PIPE_NW4_MAXNK=" + os.environ["PD_PIPE_NW4_MAXNK"]]
    if os.environ.get("PD_PIPE_XCD") and base == "attn_pipe.hip":      # lab: XCD-aware block order
        cmd[1:1] = ["-DPD_PIPE_XCD=" + os.environ["PD_PIPE_XCD"]]
    if os.environ.get("PD_TRI_WLDS") and base == "tri_attn.hip":       # lab: 0 = the two-blocks-per-CU form (weights / low parts per wave from L2)
        cmd[1:1] = ["-DPD_TRI_WLDS=" + os.environ["PD_TRI_WLDS"]]
    if os.environ.get("PD_TRI_ZD") and base == "tri_attn.hip":         # lab: depth of the row-fragment ring of the in-block projection
        cmd[1:1] = ["-DPD_TRI_ZD=" + os.environ["PD_TRI_ZD"]]
    for knob in ("PD_TRI_TAIL_PF", "PD_TRI_TAIL_GRID0", "PD_TRI_TAIL_GRID1", "PD_TRI_TAIL_ABL", "PD_TRI_TAIL_EPI"):      # lab: tile prefetch / blocks per CU of the triangle tails
        if os.environ.get(knob) and base == "tri_tail.hip":
            cmd[1:1] = [f"-D{knob}=" + os.environ[knob]]
    if os.environ.get("PD_POOL_BPC") and base == "pool.hip":            # lab: register budget of the fused pool (blocks per CU)
        cmd[1:1] = ["-DPD_POOL_BPC=" + os.environ["PD_POOL_BPC"]]
    if os.environ.get("PD_TRI_ROWS2") and base == "tri_attn.hip":      # lab: two pair rows of one head per 16-wave block
        cmd[1:1] = ["-DPD_TRI_ROWS2=" + os.environ["PD_TRI_ROWS2"]]
    if os.environ.get("PD_TRI_XCD") and base == "tri_attn.hip":        # lab: 0 = the plain (row, head) block order
        cmd[1:1] = ["-DPD_TRI_XCD=" + os.environ["PD_TRI_XCD"]]
    if os.environ.get("PD_TRI_SKEW") and base == "tri_attn.hip":       # lab: start delay of the odd-head blocks (x 8 128 cycles)
        cmd[1:1] = ["-DPD_TRI_SKEW=" + os.environ["PD_TRI_SKEW"]]
    if os.environ.get("PD_TRI_ABL") and base == "tri_attn.hip":        # lab: timing ablations of the fused triangle attention (wrong results)
        cmd[1:1] = ["-DPD_TRI_ABL=" + os.environ["PD_TRI_ABL"]]
    if os.environ.get("PD_PIPE_RES") and base == "attn_pipe.hip":      # lab: 0 = no resident-K/V form for launches of <= 256 keys
        cmd[1:1] = ["-DPD_PIPE_RES=" + os.environ["PD_PIPE_RES"]]
    if os.environ.get("PD_PIPE_ABL") and base == "attn_pipe.hip":      # lab: timing ablations of the pipelined attention (wrong results)
        cmd[1:1] = ["-DPD_PIPE_ABL=" + os.environ["PD_PIPE_ABL"]]
    if os.environ.get("PD_TR_SILU") and base == "transition_f16.hip":     # lab: form of the SiLU in the fused transition (0 division, 1 / 2 reciprocal)
        cmd[1:1] = ["-DPD_TR_SILU=" + os.environ["PD_TR_SILU"]]
    if os.environ.get("PD_TRANSITION_MIN128") and base == "transition_f16.hip":     # lab: 128-row tiles from which the fused transition takes a launch
        cmd[1:1] = ["-DPD_TRANSITION_MIN128=" + os.environ["PD_TRANSITION_MIN128"]]
    if os.environ.get("PD_TRANSITION_BM") and base == "transition_f16.hip":     # lab: 128-row tiles, one block per CU
        cmd[1:1] = ["-DPD_TRANSITION_BM=" + os.environ["PD_TRANSITION_BM"]]
    for knob in ("PD_SPLIT_MIN_TILES", "PD_SPLIT_MIN_TILES_SMALL"):
        if os.environ.get(knob) and base == "gemm_split.hip":
            cmd[1:1] = [f"-D{knob}=" + os.environ[knob]]
    if os.environ.get("PD_NO_EXTRA_FLAGS") != "1":
        cmd[1:1] = EXTRA_FLAGS.get(base, [])
    return cmd


def build(force=False, verbose=True, only=None):
    """only: names of the sources to recompile (lab loops on the GPU box: the other objects of the last full build are reused)"""
    if not force and not only and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if only and os.path.basename(src) not in only and os.path.exists(obj):
            continue
        procs.append((src, subprocess.Popen(compile_cmd(src, obj), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
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
    build(force="--force" in sys.argv, only=[a for a in sys.argv[1:] if a.endswith(".hip")] or None)
