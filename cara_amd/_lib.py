"""ctypes binding of libcara_hip.so (the C ABI declared in include/cara_hip.h).

There is no fallback: if the library cannot be loaded, or a call returns a non-zero status,
an exception is raised.  Tensors are handed over as raw device pointers; torch only owns the
memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CARA_LIB_PATH (diagnostic) points at another build of the same ABI for same-box A/B timing
LIB_PATH = os.environ.get("CARA_LIB_PATH") or os.path.join(_HERE, "libcara_hip.so")
# the same sources with IEEE-half MFMA operands (cara_amd/csrc/build.sh f16): `precision = "fp16"`
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.path.join(_HERE, "libcara_hip_f16.so")}

# every symbol include/cara_hip.h declares (tests check the library exports all of them)
SYMBOLS = (
    "cara_abi_version", "cara_build_arch", "cara_operand_type", "cara_gemm_bf16", "cara_gemm_tn_f32", "cara_pack_b_panels", "cara_gemm_scratch_bytes", "cara_skinny_xu", "cara_skinny_xu_r", "cara_tskinny_partial2_r", "cara_gemm_with_tskinny_r", "cara_gemm_rider_slab_format", "cara_gemm_epi_rider_chunks", "cara_gemm_dv_chunks", "cara_gemm_epi_rider_scratch_bytes", "cara_linear_fwd", "cara_linear_bwd",
    "cara_tskinny_scratch_bytes", "cara_tskinny_xtg", "cara_tskinny_partial", "cara_tskinny_partial2", "cara_tskinny_reduce", "cara_tskinny_reduce_many", "cara_gemm_with_tskinny", "cara_layernorm_fwd", "cara_layernorm_bwd", "cara_layernorm_fwd_xu", "cara_layernorm_bwd_xu", "cara_layernorm_fwd_ex", "cara_layernorm_bwd_ex",
    "cara_attention_fwd", "cara_attention_bwd", "cara_attention_cls_fwd", "cara_attention_cls_bwd", "cara_im2col_patches", "cara_assemble_tokens",
    "cara_cross_entropy", "cara_cross_entropy_ex", "cara_amp_update", "cara_allreduce_flat", "cara_head_forward", "cara_factor_grad_reduce_ex", "cara_f32_to_bf16", "cara_transpose_bf16", "cara_transpose_bf16_ld", "cara_pack_offsets",
    "cara_dense_delta_materialize", "cara_dense_delta_grad_scratch_bytes", "cara_dense_delta_grad", "cara_sum_slabs_f32", "cara_adamw_step",
    "cara_weight_dropout_hash", "cara_materialize_merge", "cara_dropout_grad_scratch_bytes", "cara_dropout_grad_contract", "cara_colsum_scratch_bytes", "cara_colsum_bf16", "cara_factor_prep", "cara_factor_grad_scratch_bytes", "cara_factor_grad_reduce", "cara_vit_workspace_bytes", "cara_vit_forward",
    "cara_vit_backward", "cara_head_backward", "cara_sizeof_struct", "cara_sizeof_gemm_args", "cara_profile_sites", "cara_profile_site_read", "cara_debug_tr_probe", "cara_debug_tr_frag",
)

EPI_BF16, EPI_F32, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_GELU_DG, EPI_MULH = range(7)


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int),
                ("A2", C.c_void_p), ("B2", C.c_void_p), ("Rp", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("bias", C.c_void_p), ("epi", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("C2", C.c_void_p), ("aux", C.c_void_p),
                ("rowscale", C.c_void_p), ("rows_per_sample", C.c_int),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
                ("batch", C.c_int), ("strideA", C.c_longlong), ("strideB", C.c_longlong), ("strideC", C.c_longlong),
                ("Ut", C.c_void_p), ("T_out", C.c_void_p), ("Tt_out", C.c_void_p), ("ldt", C.c_int),
                ("Bp", C.c_void_p), ("a_panels", C.c_int), ("c_panels", C.c_int), ("B3", C.c_void_p), ("Ut_rank", C.c_int),
                ("er_Tt", C.c_void_p), ("er_Gt", C.c_void_p), ("er_h", C.c_void_p), ("er_slabs_v", C.c_void_p), ("er_slabs_u", C.c_void_p),
                ("er_ldg", C.c_int), ("er_colsum", C.c_int), ("er_h_panels", C.c_int)]


class Geom(C.Structure):
    _fields_ = [("depth", C.c_int), ("dim", C.c_int), ("heads", C.c_int), ("rank", C.c_int),
                ("Rp", C.c_int), ("scale", C.c_float), ("cp_length", C.c_int)]


CP_FIELDS = ("A1", "A2", "A3", "A4", "P1", "P2", "P3", "R1", "R2", "bias1", "bias2", "bias3")


def cp_fields(cp_length: int = 4):
    """Names (without the CP_ prefix) of the CP tensors of a QKV tensorisation of order `cp_length`, in the order
    the engine passes them around: order 2 has neither A3 nor A4, order 3 no A4, order 5 an A5 (dim_experiment.py:264-297)."""
    if cp_length == 2:
        return tuple(n for n in CP_FIELDS if n not in ("A3", "A4"))
    if cp_length == 3:
        return tuple(n for n in CP_FIELDS if n != "A4")
    if cp_length == 5:
        return CP_FIELDS[:4] + ("A5",) + CP_FIELDS[4:]
    return CP_FIELDS


def cp_ptrs(fields, tensors):
    """cara_cp from tensors given in `fields` order (missing members stay NULL)."""
    return CpPtrs(**{n: ptr(t) for n, t in zip(fields, tensors)})


class CpPtrs(C.Structure):
    """the 12 tensors of the default (order-4) tensorisation in CP_FIELDS order, then A5 (order 5 only)"""
    _fields_ = [(n, C.c_void_p) for n in CP_FIELDS] + [("A5", C.c_void_p)]


class PackLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "Ut_qkv", "U_qkv", "Vs_qkv", "Vst_qkv", "Ut_proj", "U_proj", "Vs_proj", "Vst_proj",
        "Ut_fc1", "U_fc1", "Vs_fc1", "Vst_fc1", "Ut_fc2", "U_fc2", "Vs_fc2", "Vst_fc2",
        "bias_proj", "bias_fc1", "bias_fc2", "layer_stride", "total")]


class LayerGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dU_qkv", "dVs_qkv", "dU_proj", "dVs_proj", "dU_fc1", "dVs_fc1", "dU_fc2", "dVs_fc2",
        "dc_proj", "dc_fc1", "dc_fc2")]


class VitWeights(C.Structure):
    """Frozen backbone, device pointers.  [depth, ...] tensors are contiguous over layers."""
    _fields_ = [(n, C.c_void_p) for n in (
        "patch_w", "patch_b", "cls", "pos",
        "ln1_g", "ln1_b", "ln2_g", "ln2_b",
        "qkv_w", "qkv_wt", "qkv_b", "proj_w", "proj_wt", "proj_b",
        "fc1_w", "fc1_wt", "fc1_b", "fc2_w", "fc2_wt", "fc2_b",
        "norm_g", "norm_b",
        "qkv_wp", "qkv_wtp", "proj_wp", "proj_wtp", "fc1_wp", "fc1_wtp", "fc2_wp", "fc2_wtp")]


class VitShape(C.Structure):
    _fields_ = [("B", C.c_int), ("img", C.c_int), ("patch", C.c_int), ("chans", C.c_int),
                ("tokens", C.c_int), ("num_classes", C.c_int), ("eps", C.c_float),
                ("wd_exact", C.c_int), ("wd_p", C.c_float), ("wd_seed", C.c_uint), ("inference", C.c_int),
                ("loss_scale", C.c_void_p), ("found_inf", C.c_void_p)]


class TsReduce(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("slab_stride", C.c_size_t), ("D", C.c_void_p), ("colsum", C.c_void_p),
                ("batch", C.c_int), ("M", C.c_int), ("K1", C.c_int), ("Rp", C.c_int), ("Rc", C.c_int), ("wave_slabs", C.c_int)]


class Linear(C.Structure):   # cara_linear: one adapted linear per call (cara_linear_fwd / cara_linear_bwd)
    _fields_ = [("W", C.c_void_p), ("Wt", C.c_void_p), ("Ut", C.c_void_p), ("U", C.c_void_p), ("Vs", C.c_void_p), ("Vst", C.c_void_p),
                ("bias", C.c_void_p), ("in", C.c_int), ("out", C.c_int), ("Rp", C.c_int), ("rank", C.c_int)]


ADAMW_MAX_TENSORS, ADAMW_MAX_GROUPS = 32, 4


class AdamWTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_size_t), ("group", C.c_int)]


class AdamWArgs(C.Structure):
    _fields_ = [("t", AdamWTensor * ADAMW_MAX_TENSORS), ("ntensors", C.c_int), ("step", C.c_int),
                ("lr", C.c_float * ADAMW_MAX_GROUPS), ("weight_decay", C.c_float * ADAMW_MAX_GROUPS),
                ("one_minus_beta1", C.c_float), ("beta2", C.c_float), ("one_minus_beta2", C.c_float), ("eps", C.c_float),
                ("bias_correction1", C.c_float), ("bias_correction2_sqrt", C.c_float), ("skip_flag", C.c_void_p),
                ("dyn", C.c_void_p)]


# CARA_STRUCT_* of include/cara_hip.h -> the mirror above (lib() asserts that every size agrees with the library's)
STRUCT_MIRRORS = (GemmArgs, Geom, CpPtrs, PackLayout, LayerGrads, VitWeights, VitShape, TsReduce, Linear, AdamWArgs)


class CaraError(RuntimeError):
    pass


_libs = {}
_active = ["bf16"]   # the operand type the per-op wrappers below bind to (see `using`)


class using:
    """``with using("fp16"):`` -- the per-op wrappers of this module (gemm, skinny_xu, tskinny_xtg ...) and ``lib()`` without an
    argument bind to that operand build inside the block (the module-level Attention.forward / Mlp.forward path of an engine whose
    precision is "fp16").  A process-wide switch: not for concurrent use from several threads."""

    def __init__(self, operands: str):
        self.operands = operands

    def __enter__(self):
        self.prev = _active[0]
        _active[0] = self.operands
        return self

    def __exit__(self, *exc):
        _active[0] = self.prev
        return False


def act_dtype(operands=None):
    """torch dtype of the 16-bit activations / operands of a build"""
    return torch.float16 if (operands or _active[0]) == "fp16" else torch.bfloat16


def lib(operands: str = None) -> C.CDLL:
    """Load libcara_hip.so (operands = "bf16", the product) or libcara_hip_f16.so ("fp16": the same sources and ABI with IEEE-half
    MFMA operands) or raise.  Never falls back to another implementation.  No argument: the build `using` selected (default bf16)."""
    if operands is None:
        operands = _active[0]
    got = _libs.get(operands)
    if got is not None:
        return got
    if operands not in LIB_PATHS:
        raise CaraError(f"no library for operand type {operands!r}")
    path = LIB_PATHS[operands]
    if not os.path.exists(path):
        raise CaraError(
            f"{path} not found: build it with cara_amd/csrc/build.sh{' f16' if operands == 'fp16' else ''} (or __graft_entry__.build()). "
            "cara_amd has no CPU or eager fallback.")
    _lib = C.CDLL(path)
    _lib.cara_build_arch.restype = C.c_char_p
    _lib.cara_operand_type.restype = C.c_char_p
    if _lib.cara_operand_type() != operands.encode():
        raise CaraError(f"{path} was built for {_lib.cara_operand_type()!r} operands, not {operands!r}")
    _lib.cara_tskinny_scratch_bytes.restype = C.c_size_t
    _lib.cara_gemm_epi_rider_scratch_bytes.restype = C.c_size_t
    _lib.cara_factor_grad_scratch_bytes.restype = C.c_size_t
    if hasattr(_lib, "cara_vit_workspace_bytes"):
        _lib.cara_vit_workspace_bytes.restype = C.c_size_t
        _lib.cara_gemm_scratch_bytes.restype = C.c_size_t
        _lib.cara_weight_dropout_hash.restype = C.c_uint
        _lib.cara_dropout_grad_scratch_bytes.restype = C.c_size_t
        _lib.cara_colsum_scratch_bytes.restype = C.c_size_t
    _lib.cara_dense_delta_grad_scratch_bytes.restype = C.c_size_t
    _lib.cara_sizeof_struct.restype = C.c_size_t
    _lib.cara_sizeof_gemm_args.restype = C.c_size_t
    for which, mirror in enumerate(STRUCT_MIRRORS):   # a mirror that is short would make the library read past it
        want = int(_lib.cara_sizeof_struct(which))
        if want != C.sizeof(mirror):
            raise CaraError(f"{path}: sizeof({mirror.__name__}) is {C.sizeof(mirror)} here, {want} in the library: "
                            "cara_amd/_lib.py and include/cara_hip.h disagree (rebuild, or update the mirror)")
    _libs[operands] = _lib
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise CaraError(f"{what} failed with status {status} "
                        f"({ {1: 'CARA_E_ARG', 2: 'CARA_E_LAUNCH'}.get(status, '?') })")


def ptr(t) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise CaraError("cara_amd kernels need device tensors (no CPU path)")
    if not t.is_contiguous():
        raise CaraError("cara_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream(device=None) -> C.c_void_p:
    """Current stream of ``device`` (default: the current device).  Callers that own tensors pass their device and
    hold ``torch.cuda.device(device)`` around the library call: kernels launch on the CURRENT device."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ---- thin per-op wrappers (used by tests and by the module-level drop-in) ---------------------

def gemm(A, B, out, *, epi, bias=None, A2=None, B2=None, C2=None, aux=None, rowscale=None,
         rows_per_sample=0, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, scratch=None, Ut=None, T_out=None,
         Tt_out=None, Bp=None, a_panels=0, c_panels=0, B3=None, Ut_rank=0, epi_riders=None, dv=None):
    """epi_riders = (Tt, Gt, h, want_colsum[, h_panels]) with CARA_EPI_MULH: the launch's epilogue also leaves the partial sums of
    dVs = C^T T and dU = h^T G (cara_gemm_args::er_*); returns (out, slabs_v, slabs_u, chunks) then."""
    a = GemmArgs()
    a.Ut_rank = Ut_rank   # with Ut: the adapter's rank if known (<= 16 at Rp = 32: 16 of the 32 columns of T are computed)
    a.B3 = ptr(B3)   # optional second B operand: C = A B^T + A B3^T, accumulated in fp32
    a.Bp = ptr(Bp)   # optional K-panel-major image of B (pack_b_panels)
    a.a_panels, a.c_panels = a_panels, c_panels   # A given / C written as [K/32][P][32] panels
    if Ut is not None:   # adapter fully inside the GEMM: T = A Ut^T computed per tile (B2 = Vs must be given, A2 not)
        a.Ut, a.T_out, a.Tt_out = ptr(Ut), ptr(T_out), ptr(Tt_out)
        a.ldt = Tt_out.shape[1] if Tt_out is not None else 0
    if scratch is not None:   # uint8 tensor of gemm_scratch_bytes(): few-row products run as batched split-K
        a.scratch, a.scratch_bytes = ptr(scratch), scratch.numel() * scratch.element_size()
    a.M = M if M is not None else A.shape[0]
    a.K = K if K is not None else A.shape[1]
    a.N = N if N is not None else B.shape[0]
    a.A, a.lda = ptr(A), lda or A.shape[1]
    a.B, a.ldb = ptr(B), ldb or B.shape[1]
    a.A2, a.B2 = ptr(A2), ptr(B2)
    a.Rp = A2.shape[1] if A2 is not None else (B2.shape[1] if B2 is not None else 0)
    a.bias, a.epi = ptr(bias), epi
    a.C, a.ldc = ptr(out), ldc or out.shape[-1]
    a.C2, a.aux, a.rowscale = ptr(C2), ptr(aux), ptr(rowscale)
    a.rows_per_sample = rows_per_sample
    if epi_riders is not None:
        Tt, Gt, hh, want_cs = epi_riders[:4]
        a.er_Tt, a.er_Gt, a.er_h, a.er_ldg, a.er_colsum = ptr(Tt), ptr(Gt), ptr(hh), Tt.shape[1], 1 if want_cs else 0
        a.er_h_panels = epi_riders[4] if len(epi_riders) > 4 else 0
        chunks = int(lib().cara_gemm_epi_rider_chunks(C.byref(a)))
        if chunks <= 0:
            raise ValueError("this product cannot carry epilogue riders (cara_gemm_epi_rider_chunks)")
        nbytes = int(lib().cara_gemm_epi_rider_scratch_bytes(chunks, a.N))
        slabs_v = torch.zeros(nbytes, dtype=torch.uint8, device=out.device)
        slabs_u = torch.zeros(nbytes, dtype=torch.uint8, device=out.device)
        a.er_slabs_v, a.er_slabs_u = ptr(slabs_v), ptr(slabs_u)
        check(lib().cara_gemm_bf16(C.byref(a), stream()), "cara_gemm_bf16")
        return out, slabs_v, slabs_u, chunks
    if dv is not None:   # dv = (Tt, want_colsum) with CARA_EPI_BF16: dVs = A^T T (+ column sums of A) out of the launch's own A tiles
        Tt, want_cs = dv
        a.er_Tt, a.er_ldg, a.er_colsum = ptr(Tt), Tt.shape[1], 1 if want_cs else 0
        a.er_slabs_v = C.c_void_p(16)   # (any non-NULL value for the query)
        chunks = int(lib().cara_gemm_dv_chunks(C.byref(a), 0))
        if chunks <= 0:
            raise ValueError("this product cannot leave dVs from its A tiles (cara_gemm_dv_chunks)")
        slabs = torch.full((int(lib().cara_gemm_epi_rider_scratch_bytes(chunks, a.K)) // 4,), float("nan"), dtype=torch.float32, device=out.device)
        a.er_slabs_v = ptr(slabs)
        check(lib().cara_gemm_bf16(C.byref(a), stream()), "cara_gemm_bf16")
        return out, slabs, chunks
    check(lib().cara_gemm_bf16(C.byref(a), stream()), "cara_gemm_bf16")
    return out


def pack_b_panels(B):
    """bf16 [N, K] -> its K-panel-major image [K/32, N, 32] (cara_gemm_args::Bp)."""
    N, K = B.shape
    out = torch.empty(K // 32, N, 32, dtype=torch.bfloat16, device=B.device)
    check(lib().cara_pack_b_panels(ptr(B), B.stride(0), N, K, ptr(out), stream()), "cara_pack_b_panels")
    return out


def gemm_scratch_bytes() -> int:
    return int(lib().cara_gemm_scratch_bytes())


def skinny_xu(X, Ut, T, Tt=None, panels=False):
    """panels: X is the K-panel-major image [K/32, M, 32] of the [M, K] operand (ldx = -M)."""
    M, K = (X.shape[1], X.shape[0] * 32) if panels else X.shape
    Rp = Ut.shape[0]
    ldt = Tt.shape[1] if Tt is not None else 0
    check(lib().cara_skinny_xu(ptr(X), -M if panels else K, ptr(Ut), ptr(T), ptr(Tt), ldt, M, K, Rp, stream()), "cara_skinny_xu")
    return T


def tskinny_xtg(X, Gt, D, colsum=None, M=None, panels=False):
    """panels: X is the K-panel-major image [K1/32, M, 32] of the [M, K1] operand (ldx = -M)."""
    if panels:
        K1, M = X.shape[0] * 32, X.shape[1]
        nbytes = lib().cara_tskinny_scratch_bytes(M, K1, Gt.shape[0])
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
        check(lib().cara_tskinny_xtg(ptr(X), -M, ptr(Gt), Gt.shape[1], ptr(D), ptr(colsum), ptr(scratch),
                                     M, K1, Gt.shape[0], stream()), "cara_tskinny_xtg")
        return D
    M = M if M is not None else X.shape[0]
    K1 = X.shape[1]
    Rp = Gt.shape[0]
    nbytes = lib().cara_tskinny_scratch_bytes(M, K1, Rp)
    if nbytes == 0:
        raise CaraError("cara_tskinny_scratch_bytes: unsupported shape")
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
    check(lib().cara_tskinny_xtg(ptr(X), K1, ptr(Gt), Gt.shape[1], ptr(D), ptr(colsum), ptr(scratch),
                                 M, K1, Rp, stream()), "cara_tskinny_xtg")
    return D
