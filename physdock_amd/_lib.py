"""ctypes binding of libphysdock_hip.so (the C ABI of include/physdock_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a launcher returns an
error the call raises.  Tensors are torch device tensors used as typed device memory; all
launches go to the current torch HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libphysdock_hip.so")

ACT_NONE, ACT_SILU, ACT_SIGMOID, ACT_RELU = 0, 1, 2, 3
OUT_ROWMAJOR, OUT_TRANSPOSED, OUT_OPM, OUT_BIASFRAG = 0, 1, 2, 3
LOG2E = 1.4426950408889634

_fp = C.c_void_p


def _header_abi_version():
    """PD_ABI_VERSION of include/physdock_hip.h - the single source of the number every check compares against"""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "physdock_hip.h")
    return int(re.search(r"#define\s+PD_ABI_VERSION\s+(\d+)", open(hdr).read()).group(1))


ABI_VERSION = _header_abi_version()


def header_symbols():
    """names of every function include/physdock_hip.h declares"""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "physdock_hip.h")
    return sorted(set(re.findall(r"^\s*int\s+(pd_\w+)\s*\(", open(hdr).read(), flags=re.M)))


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", _fp), ("W", _fp), ("W3", _fp), ("A3", _fp), ("Y", _fp),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldw", C.c_int), ("ldy", C.c_int),
        ("batch", C.c_int),
        ("sA", C.c_longlong), ("sW", C.c_longlong), ("sY", C.c_longlong),
        ("a_kmajor", C.c_int), ("w_kmajor", C.c_int),
        ("stats", _fp), ("pro_w", _fp), ("pro_b", _fp),
        ("pro_rows_per_group", C.c_int), ("pro_gstride", C.c_int), ("pro_act", C.c_int),
        ("rowscale_acc", _fp), ("bias", _fp), ("sBias", C.c_longlong),
        ("hn_w", _fp), ("hn_cols", C.c_int), ("hn_split", C.c_int), ("hn_eps", C.c_float),
        ("act", C.c_int), ("glu", C.c_int),
        ("rowscale", _fp), ("maskadd", _fp), ("maskval", C.c_float),
        ("mul", _fp), ("ldmul", C.c_int), ("mul_rows_per_group", C.c_int), ("mul_gstride", C.c_int),
        ("out_scale", C.c_float),
        ("res", _fp), ("ldres", C.c_int), ("res_row_mod", C.c_int), ("sRes", C.c_longlong),
        ("out_mode", C.c_int), ("T1", C.c_int), ("T2", C.c_int), ("frag_transpose", C.c_int),
        ("vecA", C.c_int), ("vecW", C.c_int), ("vecY", C.c_int),
        ("ksplit_ws", _fp), ("ksplit_ws_bytes", C.c_longlong), ("ksplit", C.c_int),
        ("W2", _fp), ("w_inv", _fp), ("A2", _fp), ("a_amax", _fp),
        ("Y2", _fp), ("y2_amax", _fp), ("y2_col0", C.c_int), ("ldy2", C.c_int),      # ABI 6
        ("stats_inline", C.c_int), ("stats_eps", C.c_float),                          # ABI 7
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", _fp), ("K", _fp), ("V", _fp), ("O", _fp),
        ("nq", C.c_int), ("nk", C.c_int), ("nbatch", C.c_int), ("nheads", C.c_int),
        ("q_bs", C.c_longlong), ("q_ss", C.c_longlong), ("k_bs", C.c_longlong), ("k_ss", C.c_longlong),
        ("v_bs", C.c_longlong), ("v_ss", C.c_longlong), ("o_bs", C.c_longlong), ("o_ss", C.c_longlong),
        ("bias", _fp), ("scale", C.c_float), ("bias_nk", C.c_int), ("fp32_mfma", C.c_int),
        ("ws", _fp), ("ws_bytes", C.c_longlong), ("nsplit", C.c_int),
        ("f16x3", C.c_int), ("f16_q_amax", C.c_float), ("f16_k_amax", C.c_float), ("f16_v_amax", C.c_float), ("f16_amax", _fp), ("O2", _fp),
        ("K2", _fp), ("V2", _fp), ("kv2_bs", C.c_longlong), ("kv2_ss", C.c_longlong),      # ABI 6
        ("bias_prescale", C.c_float),                                                       # ABI 7
        ("o2_rows", C.c_longlong),                                                          # ABI 8
    ]


class TransitionArgs(C.Structure):
    """mirror of pd_transition_args"""
    _fields_ = [("x", _fp), ("M", C.c_int), ("C", C.c_int), ("hidden", C.c_int),
                ("shift", _fp), ("scale1p", _fp), ("gate", _fp), ("rows_per_group", C.c_int), ("gstride", C.c_int),
                ("eps", C.c_float), ("W13", _fp), ("w13_inv", _fp), ("W2", _fp), ("w2_inv", _fp), ("y_amax", _fp), ("h_amax", _fp),
                ("rms", C.c_int)]


class TriTailArgs(C.Structure):
    """mirror of pd_tri_tail_args"""
    _fields_ = [("z", _fp), ("o", _fp), ("M", C.c_int), ("C", C.c_int), ("Co", C.c_int), ("w_in", _fp), ("w_out", _fp),
                ("eps", C.c_float), ("Wg", _fp), ("wg_inv", _fp), ("bg", _fp), ("Wz", _fp), ("wz_inv", _fp), ("bz", _fp),
                ("zn_amax", _fp), ("on_amax", _fp), ("mode", C.c_int)]


class TriAttnArgs(C.Structure):
    """mirror of pd_tri_attn_args"""
    _fields_ = [("z2", _fp), ("W2", _fp), ("w_inv", _fp), ("bias", _fp), ("bias_prescale", C.c_float), ("bias_nk", C.c_int),
                ("o", _fp), ("T", C.c_int), ("Treal", C.c_int), ("C", C.c_int), ("nheads", C.c_int), ("transpose", C.c_int),
                ("zn_amax", C.c_float), ("qkv_amax", _fp), ("scale", C.c_float)]


class TriMulArgs(C.Structure):
    """mirror of pd_tri_mul_args"""
    _fields_ = [("q", _fp), ("k", _fp), ("o", _fp), ("T", C.c_int), ("Treal", C.c_int), ("nch", C.c_int), ("ch_stride", C.c_longlong),
                ("transpose", C.c_int), ("q_amax", _fp), ("k_amax", _fp)]


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m physdock_amd.build` "
                "(there is no CPU or PyTorch fallback for the sampler kernels)")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
        if _lib.pd_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libphysdock_hip.so ABI version {_lib.pd_abi_version()} != header {ABI_VERSION}: rebuild "
                               "(python -m physdock_amd.build --force)")
        if _lib.pd_gemm_args_size() != C.sizeof(GemmArgs) or _lib.pd_attn_args_size() != C.sizeof(AttnArgs):
            raise RuntimeError("ctypes mirrors of pd_gemm_args / pd_attn_args differ in size from the compiled structs")
        if _lib.pd_tri_attn_args_size() != C.sizeof(TriAttnArgs):
            raise RuntimeError("ctypes mirror of pd_tri_attn_args differs in size from the compiled struct")
    return _lib


#: every symbol include/physdock_hip.h declares (tests check the library exports all of them)
SYMBOLS = {}


def _declare(L):
    def sig(name, *argtypes):
        fn = getattr(L, name)
        fn.argtypes = list(argtypes)
        fn.restype = C.c_int
        SYMBOLS[name] = fn

    i, f, p, ll = C.c_int, C.c_float, C.c_void_p, C.c_longlong
    sig("pd_abi_version")
    sig("pd_gemm_args_size")
    sig("pd_attn_args_size")
    sig("pd_init")
    sig("pd_attention_occupancy")
    sig("pd_gemm", C.POINTER(GemmArgs), p)
    sig("pd_gemm_variant", C.POINTER(GemmArgs))
    sig("pd_rowstats", p, p, i, i, i, i, i, f, p)
    sig("pd_rownorm", p, p, p, p, p, i, i, i, f, i, p)
    sig("pd_norm_split", p, i, i, i, i, f, p, p, i, i, p, p)
    sig("pd_pair_bias", p, p, p, p, p, f, f, p, i, i, i, i, i, i, f, p)
    sig("pd_pair_bias_split", p, p, p, p, p, f, f, p, i, i, f, p, f, p)
    sig("pd_attention", C.POINTER(AttnArgs), p)
    sig("pd_attention_variant", C.POINTER(AttnArgs))
    sig("pd_attention_tail", C.POINTER(AttnArgs), C.POINTER(C.c_int))
    sig("pd_attention_bias_prescale_log2", f, f, f)
    sig("pd_graph_begin", p)
    sig("pd_graph_end", p, C.POINTER(C.c_void_p))
    sig("pd_graph_launch", p, p)
    sig("pd_graph_destroy", p)
    ull = C.c_ulonglong
    sig("pd_atom_pair_init", p, p, p, p, p, p, p, p, i, i, p)
    sig("pd_atom_pair_ffn", p, p, p, p, ll, i, i, p)
    sig("pd_pair_gather_add", p, p, p, i, i, i, p)
    sig("pd_pair_init_z", p, p, p, p, p, p, p, p, p, p, p, i, i, p)
    sig("pd_segment_pool", p, p, p, p, i, i, i, i, p)
    sig("pd_unpool_add", p, p, p, i, i, i, i, p)
    sig("pd_downscale_pool", p, p, p, p, p, p, p, i, i, i, i, i, i, p)
    sig("pd_gather_rows_add", p, p, p, i, i, p)
    sig("pd_axpby", p, p, f, p, p, f, ll, p)
    sig("pd_template_mask", p, p, p, p, i, i, p)
    sig("pd_template_feat", p, p, p, p, p, p, i, i, p)
    sig("pd_confidence_pair_init", p, p, p, p, p, p, p, i, i, p)
    sig("pd_pair_symmetrize", p, p, i, i, p)
    sig("pd_atom_dist_embed", p, p, p, p, i, i, p)
    sig("pd_target_feat", p, p, p, p, i, i, i, p)
    sig("pd_msa_feat", p, p, p, f, p, i, i, i, p)
    sig("pd_outer_mask", p, p, i, p)
    sig("pd_chain_contacts", p, p, p, p, i, p, f, p, i, p, p, p)
    sig("pd_pdb_format", p, p, p, p, p, i, i, i, p)
    sig("pd_augment", p, f, p, p, p, p, f, f, p, i, i, p, i, i, p)
    sig("pd_init_noise", p, p, i, f, i, i, p)
    sig("pd_precond", p, f, p, p, p, p, p, i, i, i, p)
    sig("pd_denoise", p, p, p, p, p, f, f, f, p, p, p, i, i, i, p)
    sig("pd_kabsch_align", p, p, p, ll, p, p, i, i, p)
    sig("pd_template_match", p, p, p, p, p, p, p, i, i, i, i, p)
    sig("pd_pose_dist", p, p, i, i, p)
    sig("pd_euler", p, p, p, p, f, f, f, p, i, i, p)
    sig("pd_pairwise_rmsd", p, p, p, p, p, i, i, i, p)
    sig("pd_timestep_embed", p, p, i, p)
    sig("pd_dit_bounds", p, i, i, i, i, i, p, p, p, p, p)
    sig("pd_norm_split2", p, i, i, i, i, f, p, p, i, i, p, p, p)
    sig("pd_transition_f16", C.POINTER(TransitionArgs), p)
    sig("pd_tri_tail", C.POINTER(TriTailArgs), p)
    sig("pd_tri_attention", C.POINTER(TriAttnArgs), p)
    sig("pd_tri_attn_args_size")
    sig("pd_tri_mul", C.POINTER(TriMulArgs), p)
    sig("pd_mmff_energy_grad", p, p, p, p, i, p)
    sig("pd_mmff_relax", p, p, p, p, p, ll, i, i, i, p)
    sig("pd_chirality", p, p, p, p, p, i, i, i, p)
    sig("pd_ligand_gather", p, p, p, i, i, i, p)
    sig("pd_ligand_scatter", p, p, p, p, i, i, i, p)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.float64, torch.int32, torch.int64, torch.uint8, torch.bfloat16, torch.float16), (t.device, t.dtype)
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


_inited = False


def init():
    global _inited
    L = lib()
    if not _inited:
        check(L.pd_init(), "pd_init")
        _inited = True
    return L
