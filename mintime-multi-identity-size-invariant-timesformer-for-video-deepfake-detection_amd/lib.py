"""ctypes binding of csrc/libmintime_hip.so (C ABI: include/mintime_hip.h).

There is deliberately NO fallback: `get()` raises if the shared library is missing or a symbol the
header declares is not exported, and every wrapper raises on a non-zero return code.
"""
import ctypes as C
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MT_LIB") or os.path.join(CSRC, "libmintime_hip.so")      # MT_LIB: a lab build of the library

_lib = None

f32p = C.c_void_p
i64 = C.c_int64


class RowMap(C.Structure):
    _fields_ = [("gin", C.c_int), ("gout", C.c_int), ("off", C.c_int)]


class GemmDesc(C.Structure):
    _fields_ = [("op", C.c_int), ("prologue", C.c_int), ("epilogue", C.c_int),
                ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("a_map", RowMap), ("b_map", RowMap), ("c_map", RowMap),
                ("bias", C.c_void_p), ("R", C.c_void_p), ("ldr", i64),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("gate", C.c_void_p), ("hw", C.c_int),
                ("C2", C.c_void_p), ("ldc2", i64), ("stats", C.c_void_p), ("stats_slots", C.c_int),
                ("n_half", C.c_int), ("split_k", C.c_int),
                ("A2", C.c_void_p), ("b_prologue", C.c_int), ("b_scale", C.c_void_p), ("b_shift", C.c_void_p),
                ("b_gate", C.c_void_p), ("b_hw", C.c_int),
                ("conv_H", C.c_int), ("conv_W", C.c_int), ("conv_C", C.c_int), ("conv_Ho", C.c_int), ("conv_Wo", C.c_int),
                ("conv_k", C.c_int), ("conv_stride", C.c_int), ("conv_pad", C.c_int), ("conv_act", C.c_int), ("conv_src_u8", C.c_int),
                ("col_sum", C.c_void_p), ("b_planes", C.c_void_p), ("b_plane_stride", C.c_int64),
                ("e_scale", C.c_void_p), ("e_shift", C.c_void_p), ("e_gate", C.c_void_p), ("e_dpool", C.c_void_p),
                ("e_mi", C.c_void_p), ("e_hw", C.c_int)]


class GemmPlanesDesc(C.Structure):
    _fields_ = [("op", C.c_int), ("epilogue", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("a_planes", C.c_void_p), ("b_planes", C.c_void_p), ("C", C.c_void_p), ("ldc", i64),
                ("bias", C.c_void_p), ("R", C.c_void_p), ("ldr", i64), ("C2", C.c_void_p), ("ldc2", i64),
                ("n_half", C.c_int), ("col_sum", C.c_void_p), ("c_planes", C.c_void_p), ("split_k", C.c_int),
                ("sk_workspace", C.c_void_p), ("sk_workspace_bytes", i64), ("stats", C.c_void_p), ("stats_slots", C.c_int)]


OP_NT, OP_NN, OP_TN = 0, 1, 2
PRO_NONE, PRO_BN_SWISH_GATE, PRO_BN_SWISH, PRO_AFFINE, PRO_BN_BWD, PRO_IM2COL = 0, 1, 2, 3, 4, 5
BPRO_NONE, BPRO_BN_SWISH_GATE, BPRO_IM2COL = 0, 1, 2
EPI_STORE, EPI_BIAS_RES, EPI_GEGLU, EPI_STATS, EPI_ATOMIC, EPI_GEGLU_BWD, EPI_ACCUM, EPI_SE_RED, EPI_ACT_BWD = 0, 1, 2, 3, 4, 5, 6, 7, 8

# name -> argtypes (restype is always int unless listed in _RESTYPES); mirrors include/mintime_hip.h
PROTOTYPES = {
    "mt_version": [],
    "mt_last_error": [],
    "mt_set_deterministic": [C.c_int],
    "mt_get_deterministic": [],
    "mt_det_release": [C.c_void_p],
    "mt_det_bn_sums": [f32p, f32p, f32p, i64, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "mt_gemm": [C.POINTER(GemmDesc), C.c_void_p],
    "mt_gemm_set_split": [C.c_int],
    "mt_gemm_get_split": [],
    "mt_split_planes": [f32p, C.c_void_p, C.c_int64, C.c_void_p],
    "mt_planes_elems": [C.c_int, C.c_int],
    "mt_split_planes_blk": [f32p, i64, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "mt_split_planes_blk_multi": [C.c_void_p, C.c_int, i64, C.c_void_p],
    "mt_bn_bwd_apply_planes": [f32p, f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "mt_gemm_planes": [C.POINTER(GemmPlanesDesc), C.c_void_p],
    "mt_gemm_planes_workspace_bytes": [],
    "mt_gemm_planes_set_persist": [C.c_int],
    "mt_mul_planes": [f32p, f32p, C.c_void_p, f32p, C.c_int, C.c_int, C.c_void_p],
    "mt_mul_add": [f32p, f32p, f32p, f32p, i64, C.c_void_p],
    "mt_geglu_bwd": [f32p, f32p, f32p, C.c_void_p, f32p, C.c_int, C.c_int, C.c_void_p],
    "mt_layernorm_fwd": [f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p],
    "mt_embed_fwd": [f32p, f32p, f32p, f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                     C.c_void_p],
    "mt_attn_fwd": [f32p, f32p, f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                    C.c_void_p, C.c_void_p],
    "mt_head_fwd": [f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p],
    "mt_stem_conv_fwd": [C.c_void_p, C.c_int, f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_stem_conv_fwd_valid": [C.c_void_p, C.c_int, f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_dwconv_fwd": [f32p, f32p, f32p, f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_int, C.c_void_p],
    "mt_dwconv_fwd_planes": [f32p] * 4 + [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p],
    "mt_dwconv_rc_supported": [C.c_int] * 5,
    "mt_dwconv_fwd_rc": [f32p, f32p, C.c_int, f32p, f32p, f32p, f32p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p],
    "mt_dwconv_bwd_rc": [f32p] * 6 + [C.c_int] + [f32p] * 4 + [C.c_void_p, C.c_int, f32p] + [C.c_int] * 7 + [C.c_void_p],
    "mt_bn_finalize": [C.c_void_p, C.c_int, C.c_double, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_float,
                       C.c_float, C.c_int, C.c_void_p],
    "mt_se_pool_parts": [C.c_int, C.c_int, C.c_int],
    "mt_se_pool_fwd": [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_se_gate_fwd": [f32p, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_bn_act_fwd": [f32p, f32p, f32p, f32p, f32p, i64, C.c_int, C.c_int, f32p, C.c_int, C.c_void_p],
    "mt_attn_aggregate": [f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p],
    "mt_build_clip_inputs": [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_void_p],
    "mt_layernorm_bwd": [f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p, C.c_void_p],
    "mt_layernorm_bwd_rows": [f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "mt_layernorm_bwd_cols": [f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_layernorm_bwd_rows_blocks": [C.c_int],
    "mt_layernorm_bwd_rows_sums": [f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_void_p, f32p, C.c_int, C.c_void_p],
    "mt_layernorm_bwd_cols_reduce": [f32p, C.c_int, C.c_int, f32p, f32p, f32p, C.c_void_p],
    "mt_colsum": [f32p, i64, RowMap, C.c_int, C.c_int, f32p, C.c_void_p],
    "mt_head_bwd": [f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                    C.c_void_p],
    "mt_embed_bwd": [f32p, f32p, f32p, f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_attn_bwd": [f32p, f32p, f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                    C.c_void_p, C.c_void_p],
    "mt_bn_act_bwd": [f32p] * 9 + [C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_bn_bwd_finalize": [C.c_void_p, C.c_int, C.c_double, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_void_p],
    "mt_se_bwd": [f32p] * 17 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_void_p],
    "mt_se_scratch_floats": [C.c_int, C.c_int, C.c_int],
    "mt_dwconv_bwd": [f32p] * 9 + [C.c_void_p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_int, f32p, f32p, C.c_void_p],
    "mt_dwconv_bwd_res2": [f32p] * 9 + [C.c_void_p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_int, f32p, f32p, C.c_void_p],
    "mt_conv_weight_pack": [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_conv_weight_unpack_grad": [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_maxpool_add_fwd": [f32p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_maxpool_bwd": [f32p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_maxpool_add_fwd_arg": [f32p] * 7 + [C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_maxpool_bwd_arg": [f32p, C.c_void_p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_maxpool_bn_bwd_apply_planes": [f32p, C.c_void_p, f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_bn_bwd_apply": [f32p, f32p, f32p, f32p, i64, C.c_int, C.c_void_p],
    "mt_conv1x1_rows_supported": [C.c_int, C.c_int, C.c_int],
    "mt_conv1x1_rows_instance": [C.c_int, C.c_int],
    "mt_conv1x1_rows": [f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p, C.c_void_p, C.c_int, C.c_int64,
                        C.c_int, C.c_int, C.c_void_p],
    "mt_bce_logits": [f32p, f32p, C.c_float, f32p, f32p, C.c_int, C.c_void_p],
    "mt_transpose_multi": [C.c_void_p, C.c_int, C.c_int64, C.c_void_p],
    "mt_sgd_multi": [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_void_p],
    "mt_adam_multi": [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, C.c_float,
                      C.c_int, C.c_void_p],
    "mt_conv1x1_wgrad_supported": [C.c_int, C.c_int],
    "mt_conv1x1_wgrad": [f32p] * 7 + [C.c_int, f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p],
    "mt_conv1x1_wgrad_wide_supported": [C.c_int, C.c_int],
    "mt_conv1x1_wgrad_wide": [f32p] * 7 + [C.c_int, f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p],
    "mt_conv1x1_bwd_fused_supported": [C.c_int, C.c_int],
    "mt_se_stage_fused_supported": [C.c_int, C.c_int, C.c_int],
    "mt_se_stage_fused": [f32p] * 7 + [C.c_int] + [f32p] * 5 + [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_conv1x1_bwd_fused": [f32p] * 7 + [C.c_int64, C.c_int, C.c_int, C.c_void_p],
    "mt_stem_conv_wgrad": [f32p] * 4 + [C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_stem_conv_wgrad_valid": [f32p] * 4 + [C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "mt_bn_act_fwd_planes": [f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_void_p, C.c_void_p],
    "mt_bn_swish_gate_planes": [f32p, f32p, f32p, f32p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "mt_plan_create": [C.POINTER(C.c_void_p)],
    "mt_plan_destroy": [C.c_void_p],
    "mt_plan_record_begin": [C.c_void_p],
    "mt_plan_record_end": [C.c_void_p],
    "mt_plan_size": [C.c_void_p],
    "mt_plan_tag": [C.c_int, C.c_double],
    "mt_plan_fork": [C.c_void_p, C.c_void_p],
    "mt_plan_run": [C.c_void_p, C.c_uint],
    "mt_plan_probe_read": [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "mt_memset_async": [C.c_void_p, C.c_int, i64, C.c_void_p],
    "mt_copy_async": [C.c_void_p, C.c_void_p, i64, C.c_void_p],
}
_RESTYPES = {"mt_last_error": C.c_char_p, "mt_planes_elems": C.c_int64, "mt_gemm_planes_workspace_bytes": C.c_int64}


class MintimeHipError(RuntimeError):
    pass


def build(verbose: bool = False):
    """Compile every HIP source for gfx950 into csrc/libmintime_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(16, os.cpu_count() or 4))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise MintimeHipError("building libmintime_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-8000:])
    if verbose:
        print(res.stdout[-2000:])
    global _lib
    _lib = None
    return LIB_PATH


# MT_VERSION of include/mintime_hip.h this binding was written against (tests/test_host_logic.py keeps the two equal; the package
# itself does not need the header at run time -- it may be copied or installed without the repository's include/ directory)
ABI_VERSION = 119


def header_version() -> int:
    """MT_VERSION as written in include/mintime_hip.h (repository checkouts only)."""
    import re
    with open(os.path.join(_HERE, "..", "include", "mintime_hip.h")) as f:
        return int(re.search(r"#define\s+MT_VERSION\s+(\d+)", f.read()).group(1))


def get():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MintimeHipError(
            f"{LIB_PATH} is missing: the MINTIME hot path has no CPU/eager fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MintimeHipError(f"libmintime_hip.so does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    v = lib.mt_version()
    if v != ABI_VERSION:
        raise MintimeHipError(f"libmintime_hip.so version {v} != binding version {ABI_VERSION}; rebuild it "
                              "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    _lib = lib
    return lib


def set_deterministic(on: bool) -> bool:
    """Deterministic mode (include/mintime_hip.h; reference train.py:110): no floating-point atomics anywhere in the training step,
    every reduction in a fixed order.  Also switched on by MT_DETERMINISTIC=1.  Returns the previous setting."""
    prev = bool(get().mt_get_deterministic())
    get().mt_set_deterministic(1 if on else 0)
    return prev


def deterministic() -> bool:
    return bool(get().mt_get_deterministic())


def set_gemm_split(on: bool) -> bool:
    """Select the matrix pipe of the prologue-free contractions (include/mintime_hip.h, mt_gemm_set_split): True = split-operand
    fp32 on the bf16 pipe (default), False = fp32 MFMA everywhere.  Returns the previous setting."""
    return bool(get().mt_gemm_set_split(1 if on else 0))


def split_planes(w):
    """The three exact bf16 pieces of an fp32 tensor (w == p[0] + p[1] + p[2]), shape [3, *w.shape], for mt_gemm's b_planes."""
    w = w.detach()
    if not w.is_contiguous() or w.numel() % 8:
        raise MintimeHipError("split_planes: contiguous tensor with numel % 8 == 0 expected")
    out = torch.empty((3,) + tuple(w.shape), dtype=torch.bfloat16, device=w.device)
    check(get().mt_split_planes(ptr(w), ptr(out), w.numel(), stream_ptr()), "mt_split_planes")
    return out


def planes_shape(rows, cols):
    """[3, Rp/32, Cp/16, 32, 16]: the blocked bf16 plane tensor of an fp32 [rows, cols] matrix (include/mintime_hip.h)."""
    return (3, (rows + 31) // 32, (cols + 15) // 16, 32, 16)


def planes_empty(rows, cols, device):
    """Uninitialised plane tensor; its producer kernel writes every element including the zero padding."""
    return torch.empty(planes_shape(rows, cols), dtype=torch.bfloat16, device=device)


def split_planes_blk(x, rows=None, cols=None, ld=None, out=None):
    """Plane tensor of an fp32 matrix (row-major, leading dimension ld): the converter for tensors no producer kernel emits."""
    if rows is None:
        rows, cols = x.shape[-2], x.shape[-1]
    if ld is None:
        ld = cols
    if out is None:
        out = planes_empty(rows, cols, x.device)
    check(get().mt_split_planes_blk(ptr(x), ld, rows, cols, ptr(out), stream_ptr()), "mt_split_planes_blk")
    return out


def planes_to_float(planes, rows, cols):
    """fp32 [rows, cols] = p0 + p1 + p2 of a plane tensor (tests / debugging only: plain torch ops)."""
    p = planes.float().sum(0)                       # exact: the three pieces of an fp32 value add back to it
    return p.permute(0, 2, 1, 3).reshape(p.shape[0] * 32, p.shape[1] * 16)[:rows, :cols]


_SK_WORKSPACES = {}


def streamk_workspace(device=None):
    """The stream-K scratch of the CURRENT stream (flags + one fp32 slab per persistent block): zero-filled once, lent to every
    mt_gemm_planes launch of that stream that asks for stream-K (include/mintime_hip.h)."""
    st = torch.cuda.current_stream(device)
    key = (st.device.index, st.cuda_stream)
    ws = _SK_WORKSPACES.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        n = get().mt_gemm_planes_workspace_bytes()
        ws = torch.zeros(n, dtype=torch.uint8, device=st.device)
        _SK_WORKSPACES[key] = ws
    return ws


def planes_fit(rows: int, cols: int) -> bool:
    """True when a rows x cols plane tensor can be a mt_gemm_planes operand: the loop reaches an operand's three bf16 planes by
    32-bit byte offsets from one base, so the padded tensor must stay under 4 GB (csrc/gemm_planes.hip raises otherwise)."""
    rp, cp = (rows + 31) // 32 * 32, (cols + 15) // 16 * 16
    return 6 * rp * cp <= 0xFFFFFFFF


def gemm_planes(op, a_planes, b_planes, M, N, K, Cout=None, ldc=0, epilogue=EPI_STORE, bias=None, R=None, ldr=0, C2=None, ldc2=0,
                n_half=0, col_sum=None, c_planes=None, split_k=0, streamk=None, stats=None, stats_slots=0):
    """streamk: True lends the stream's workspace (persistent grid sharing the (tile, k-step) list), False = one block per tile,
    None = MT_PLANES_STREAMK (default 0: on the TimeSformer's shapes one block per tile measured faster)."""
    d = GemmPlanesDesc()
    if streamk is None:
        streamk = os.environ.get("MT_PLANES_STREAMK", "0") != "0"
    if streamk and op != OP_TN:
        ws = streamk_workspace(a_planes.device)
        if ws is not None:
            d.sk_workspace, d.sk_workspace_bytes = ptr(ws), ws.numel()
    d.op, d.epilogue, d.M, d.N, d.K = op, epilogue, M, N, K
    d.a_planes, d.b_planes, d.C, d.ldc = ptr(a_planes), ptr(b_planes), ptr(Cout), ldc
    d.bias, d.R, d.ldr, d.C2, d.ldc2 = ptr(bias), ptr(R), ldr, ptr(C2), ldc2
    d.n_half, d.col_sum, d.c_planes, d.split_k = n_half, ptr(col_sum), ptr(c_planes), split_k
    d.stats, d.stats_slots = ptr(stats), stats_slots
    if recording() is not None:
        if op == OP_TN:
            tag_next(TAG_WGRAD, 2.0 * M * N * K)
        elif epilogue == EPI_GEGLU:
            tag_next(TAG_FF1, 2.0 * M * N * K)
    prof = PROFILE
    if prof is not None:
        for pr in prof:
            if "match_planes" in pr and pr["match_planes"](d):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(get().mt_gemm_planes(C.byref(d), stream_ptr()), "mt_gemm_planes")
                e1.record()
                pr["events"].append((e0, e1, 2.0 * M * N * K))
                return
    check(get().mt_gemm_planes(C.byref(d), stream_ptr()), "mt_gemm_planes")


def gemm_split_enabled() -> bool:
    return bool(get().mt_gemm_get_split())


def check(rc, what):
    if rc != 0:
        msg = get().mt_last_error()
        raise MintimeHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Refuses CPU tensors: the library only takes device memory.
    While this thread records a launch plan the tensor is pinned by the plan: every address a recorded call holds stays allocated
    (and is therefore never handed out again) for as long as the plan lives."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MintimeHipError("libmintime_hip takes device pointers only; got a CPU tensor (no CPU fallback exists)")
    rec = getattr(_REC, "plan", None)
    if rec is not None:
        rec.keep[id(t)] = t
    return C.c_void_p(t.data_ptr())


# ---- launch plans (include/mintime_hip.h "Launch plans"; csrc/plan.hip) ---------------------------------------------------------
_REC = threading.local()

TAG_WGRAD, TAG_FF1, TAG_DWCONV_DGRAD = 1, 2, 3      # probe tags of the bench legs (mt_plan_tag)


def recording():
    """The plan this thread is recording, or None."""
    return getattr(_REC, "plan", None)


class Plan:
    """One recorded phase: the C-side call list plus every tensor whose address it holds."""

    def __init__(self):
        h = C.c_void_p()
        check(get().mt_plan_create(C.byref(h)), "mt_plan_create")
        self.handle = h
        self.keep = {}
        self.ops = 0

    def __enter__(self):
        if recording() is not None:
            raise MintimeHipError("a launch plan is already being recorded on this thread")
        check(get().mt_plan_record_begin(self.handle), "mt_plan_record_begin")
        _REC.plan = self
        return self

    def __exit__(self, *exc):
        _REC.plan = None
        check(get().mt_plan_record_end(self.handle), "mt_plan_record_end")
        self.ops = get().mt_plan_size(self.handle)
        return False

    def pin(self, *tensors):
        for t in tensors:
            if t is not None:
                self.keep[id(t)] = t

    def run(self, probe_mask=0):
        check(get().mt_plan_run(self.handle, probe_mask), "mt_plan_run")

    def probe_read(self, tag):
        n, ms, work = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        check(get().mt_plan_probe_read(self.handle, tag, C.byref(n), C.byref(ms), C.byref(work)), "mt_plan_probe_read")
        return n.value, ms.value, work.value

    def __del__(self):
        try:
            if self.handle and _lib is not None:
                _lib.mt_plan_destroy(self.handle)
        except Exception:        # noqa: BLE001  (interpreter shutdown)
            pass


def tag_next(tag, work=0.0):
    """Tag the next recorded call (bench.py's probe legs time tagged calls with events inside mt_plan_run)."""
    if recording() is not None:
        get().mt_plan_tag(tag, float(work))


def zeros(shape, dtype, device):
    """torch.zeros as torch.empty + mt_memset_async: the fill is an entry point of the library, so a recorded phase re-issues it."""
    if torch.device(device).type != "cuda":
        return torch.zeros(shape, dtype=dtype, device=device)      # host-side buffer carving (the gloo tests of ddp.py): no launch
    t = torch.empty(shape, dtype=dtype, device=device)
    if t.numel():
        check(get().mt_memset_async(ptr(t), 0, t.numel() * t.element_size(), stream_ptr()), "mt_memset_async")
    return t


def zero_(t):
    if not t.is_contiguous():
        raise MintimeHipError("zero_: contiguous tensor expected")
    if t.numel():
        check(get().mt_memset_async(ptr(t), 0, t.numel() * t.element_size(), stream_ptr()), "mt_memset_async")
    return t


# Optional live kernel timing (bench.py's roofline legs): a list of probes {"match": fn(desc) -> bool, "events": [...]} for
# mt_gemm launches, and named probes {"name": str, "events": [...]} for other entry points (timed(name, fn, work)).
# Events are recorded on the stream the kernel is launched on (torch's current stream: inside SideStream.launch that is the
# side stream), so a duration is what the launch took IN the step, next to whatever the other stream was running.
PROFILE = None


def timed(name, fn, work=0.0):
    """Run fn() (one kernel launch); if a probe called `name` is active, bracket it with events and note `work` (bytes or flops)."""
    if name == "dwconv_dgrad":
        tag_next(TAG_DWCONV_DGRAD, work)
    prof = PROFILE
    if prof is not None:
        for pr in prof:
            if pr.get("name") == name:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                pr["events"].append((e0, e1, work))
                return
    fn()


def gemm(op, A, B, Cout, M, N, K, lda, ldb, ldc, prologue=PRO_NONE, epilogue=EPI_STORE, bias=None, R=None, ldr=0,
         scale=None, shift=None, gate=None, hw=1, C2=None, ldc2=0, stats=None, stats_slots=1, n_half=0, split_k=1,
         a_map=(0, 0, 0), b_map=(0, 0, 0), c_map=(0, 0, 0), A2=None, b_prologue=BPRO_NONE, b_scale=None, b_shift=None,
         b_gate=None, b_hw=1, conv=None, col_sum=None, b_planes=None, epi=None):
    d = GemmDesc()
    d.op, d.prologue, d.epilogue = op, prologue, epilogue
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(Cout)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.a_map, d.b_map, d.c_map = RowMap(*a_map), RowMap(*b_map), RowMap(*c_map)
    d.bias, d.R, d.ldr = ptr(bias), ptr(R), ldr
    d.scale, d.shift, d.gate, d.hw = ptr(scale), ptr(shift), ptr(gate), hw
    d.C2, d.ldc2, d.stats, d.stats_slots = ptr(C2), ldc2, ptr(stats), stats_slots
    d.n_half, d.split_k = n_half, split_k
    d.A2, d.b_prologue, d.b_scale, d.b_shift, d.b_gate, d.b_hw = ptr(A2), b_prologue, ptr(b_scale), ptr(b_shift), ptr(b_gate), b_hw
    d.col_sum = ptr(col_sum)
    if b_planes is not None:             # bf16 [3, N, K] from split_planes(B)
        d.b_planes, d.b_plane_stride = ptr(b_planes), b_planes[0].numel()
    if epi is not None:    # SE_RED / ACT_BWD: (e_scale, e_shift, e_gate, e_dpool, e_mi, e_hw)
        d.e_scale, d.e_shift, d.e_gate, d.e_dpool, d.e_mi, d.e_hw = (ptr(epi[0]), ptr(epi[1]), ptr(epi[2]), ptr(epi[3]),
                                                                      ptr(epi[4]), epi[5])
    if conv is not None:   # (H, W, C, Ho, Wo, k, stride, pad, act[, src_u8])
        (d.conv_H, d.conv_W, d.conv_C, d.conv_Ho, d.conv_Wo, d.conv_k, d.conv_stride, d.conv_pad, d.conv_act) = conv[:9]
        d.conv_src_u8 = conv[9] if len(conv) > 9 else 0
    if recording() is not None:
        if op == OP_TN and prologue == PRO_NONE and b_prologue == BPRO_NONE:
            tag_next(TAG_WGRAD, 2.0 * M * N * K)
        elif epilogue == EPI_GEGLU:
            tag_next(TAG_FF1, 2.0 * M * N * K)
    prof = PROFILE
    if prof is not None:
        for pr in prof:
            if "match" in pr and pr["match"](d):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(get().mt_gemm(C.byref(d), stream_ptr()), "mt_gemm")
                e1.record()
                pr["events"].append((e0, e1, 2.0 * M * N * K))
                return
    check(get().mt_gemm(C.byref(d), stream_ptr()), "mt_gemm")


def zero_grads(params, with_flat=False):
    """Zero-filled gradient tensors for `params` carved out of ONE flat buffer (one memset instead of hundreds);
    every view starts on a 16-byte boundary.  Entries for None params are None.  with_flat=True also returns the buffer
    (what the data-parallel reducer all-reduces in place)."""
    offs, total = [], 0
    for p in params:
        offs.append(total)
        if p is not None:
            total += (p.numel() + 3) // 4 * 4
    dev = next(p for p in params if p is not None).device
    flat = zeros(total, torch.float32, dev)
    views = [None if p is None else flat[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs)]
    return (views, flat) if with_flat else views


def grads_ready(model, params, flat):
    """End of a network's backward: every parameter gradient of `model` is complete and lives in `flat`.  A data-parallel reducer
    registered on the module (ddp.OverlappedGradReducer) starts its all-reduce here, under the rest of the backward pass."""
    hook = getattr(model, "_grads_ready_hook", None)
    if hook is not None:
        hook(params, flat)


def _low_priority_stream(device, cu_mask_words=None):
    """The weight-gradient stream at the LOWEST hardware queue priority: its kernels fill the CUs the critical path leaves idle
    instead of taking CUs from it (in-step trace: with equal priorities the main queue's LayerNorm-backward launches stretched
    2x next to the three-blocks-per-CU weight-gradient GEMMs).  torch only exposes normal / high, so the stream is created through
    the HIP runtime and wrapped.  MT_SIDE_PRIORITY=normal keeps torch's default."""
    if os.environ.get("MT_SIDE_PRIORITY", "low") != "low":
        return torch.cuda.Stream(device=device)
    if cu_mask_words:
        # experiment: confine the stream's kernels to a subset of the CUs (hipExtStreamCreateWithCUMask; 32 CUs per word)
        try:
            hip = C.CDLL("libamdhip64.so")
            idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
            arr = (C.c_uint32 * len(cu_mask_words))(*cu_mask_words)
            handle = C.c_void_p()
            with torch.cuda.device(idx):
                rc = hip.hipExtStreamCreateWithCUMask(C.byref(handle), C.c_uint32(len(cu_mask_words)), arr)
            if rc == 0:
                return torch.cuda.ExternalStream(handle.value, device=torch.device("cuda", idx))
        except (OSError, AttributeError):
            pass
    try:
        hip = C.CDLL("libamdhip64.so")
        least, greatest = C.c_int(0), C.c_int(0)
        idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            if hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) != 0 or least.value <= 0:
                return torch.cuda.Stream(device=device)
            handle = C.c_void_p()
            if hip.hipStreamCreateWithPriority(C.byref(handle), C.c_uint(1), least) != 0:      # 1 = hipStreamNonBlocking
                return torch.cuda.Stream(device=device)
        return torch.cuda.ExternalStream(handle.value, device=torch.device("cuda", idx))
    except (OSError, AttributeError):
        return torch.cuda.Stream(device=device)


class SideStream:
    """Second HIP stream for work that is independent of the critical path (weight-gradient GEMMs, bias sums): their
    blocks fill the CUs a skinny dgrad leaves idle (396 tiles on 256 CUs = 23 % idle slots).  Every launch is fenced by
    events both ways; tensors read on the side stream are pinned with record_stream so the caching allocator cannot
    hand their memory to a later main-stream allocation while the side kernel is still reading it."""

    _streams = {}

    def __init__(self, device, enabled=True, name="", hold=False):
        """hold: instead of record_stream, the tensors a side launch reads are kept alive by this object and dropped in HOST order
        at `release_point()`s, two points behind, after the main stream has been made to wait for the side stream's progress
        up to there.  record_stream hands a block back only once the allocator has seen the side event complete: with the host
        running steps ahead of the device (no per-step sync) every step then takes fresh memory for all such tensors, and a large
        configuration grows its pools until hipMalloc fails and the allocator frees everything and retries, every step (BASELINE
        config 5: 127 -> 253 GiB reserved in 12 steps, then 1.4 s / step instead of 0.2)."""
        self.enabled = enabled and os.environ.get("MT_SIDE_STREAM", "1") != "0"
        self.device = device
        self.hold = hold and os.environ.get("MT_SIDE_HOLD", "1") != "0"
        self.held, self.epochs = [], []
        if self.enabled:
            key = str(device) + name                 # name: a further low-priority stream (deferred weight gradients)
            if key not in SideStream._streams:
                mask = os.environ.get("MT_LATE_CU_MASK") if name else None      # "ffffffff,ffffffff,0,0,..." (hex words, 32 CUs each)
                SideStream._streams[key] = _low_priority_stream(device, [int(w, 16) for w in mask.split(",")] if mask else None)
            self.stream = SideStream._streams[key]
        self.pending = []

    def __del__(self):
        # hold mode on an error path: a backward that raised between a launch and wait() drops this object with tensors the side
        # stream may still be reading; make the main stream wait for the side stream BEFORE those references go back to its pool
        try:
            if self.enabled and self.hold and (self.held or self.epochs):
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
        except Exception:
            pass

    def launch(self, fn, reads=()):
        """Run fn() on the side stream once everything enqueued so far on the current stream is done."""
        if not self.enabled:
            fn()
            return None
        main = torch.cuda.current_stream(self.device)
        rec = recording()
        if rec is not None:
            # recording a launch plan: the dependency is an entry point of the library (recorded with the launches), and the
            # tensors the side launch reads are pinned by the plan instead of by the caching allocator's stream bookkeeping
            check(get().mt_plan_fork(C.c_void_p(main.cuda_stream), C.c_void_p(self.stream.cuda_stream)), "mt_plan_fork")
            rec.pin(*reads)
            with torch.cuda.stream(self.stream):
                fn()
            self.pending = [None]
            return None
        ready = torch.cuda.Event()
        ready.record(main)
        self.stream.wait_event(ready)
        if self.hold:
            self.held.append(reads)
        else:
            for t in reads:
                if t is not None:
                    t.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            fn()
        done = torch.cuda.Event()
        done.record(self.stream)
        self.pending.append(done)
        return done

    def release_point(self, depth=2):
        """hold mode: what the side launches since the previous point read may be freed once the main stream waits for the side
        stream's position NOW; that wait is issued `depth` points later (by then it costs nothing) and the tensors are dropped then."""
        if not (self.enabled and self.hold) or recording() is not None:
            return
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.epochs.append((ev, self.held))
        self.held = []
        while len(self.epochs) > depth:
            ev0, held0 = self.epochs.pop(0)
            torch.cuda.current_stream(self.device).wait_event(ev0)
            held0.clear()

    def wait(self, event=None):
        """Make the current (main) stream wait for one side launch, or for all of them."""
        if not self.enabled:
            return
        main = torch.cuda.current_stream(self.device)
        if event is not None:
            main.wait_event(event)
            return
        if self.pending:
            if self.pending[-1] is None:             # launches made while recording a plan
                check(get().mt_plan_fork(C.c_void_p(self.stream.cuda_stream), C.c_void_p(main.cuda_stream)), "mt_plan_fork")
            else:
                main.wait_event(self.pending[-1])    # the side stream is in order: its last launch implies all earlier ones
        self.pending = []
        self.held, self.epochs = [], []              # joined: everything the side stream read may go back to the main stream's pool
