"""ctypes binding of ``libamt_hip.so`` (C ABI declared in ``include/amt_hip.h``).

There is deliberately no CPU fallback: if the shared library is missing, or a call returns a
non-zero status, this module raises.  Build the library with ``python -c "import __graft_entry__ as
g; g.build()"`` or ``video2music_amd/csrc/build.sh``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMT_LIB: an alternate build of the same library (tools/ab_build.sh: A/B kernel experiments on one box); default = the in-tree build
LIB_PATH = os.environ.get("AMT_LIB") or os.path.join(_HERE, "lib", "libamt_hip.so")


class AmtError(RuntimeError):
    pass


class AmtConfig(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("num_heads", C.c_int32), ("d_model", C.c_int32),
                ("dim_feedforward", C.c_int32), ("max_sequence_video", C.c_int32),
                ("max_sequence_chord", C.c_int32), ("total_vf_dim", C.c_int32), ("max_batch", C.c_int32)]


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float


class DecodeGemmArgs(C.Structure):           # amt_decode_gemm_args
    _fields_ = [("x", _P), ("ldx", _I), ("x2", _P), ("ldx2", _I), ("K1", _I), ("K", _I),
                ("w_low", _P), ("bias_low", _P), ("resid", _P), ("relu", _I), ("w_high", _P), ("bias_high", _P),
                ("n_low", _I), ("n_high", _I), ("pro", _I), ("fold_g", _P), ("fold_c", _P), ("ln_w", _P), ("ln_b", _P),
                ("y_low", _P), ("y_high", _P), ("scratch_low", _P), ("scratch_high", _P), ("B", _I), ("eps", _F)]


class V2StepArgs(C.Structure):               # amt_v2_step_args
    _fields_ = [("tab", _P), ("n_layers", _I), ("H", _I), ("E", _I), ("dff", _I), ("n_exp", _I), ("S", _I), ("max_seq", _I), ("B", _I),
                ("keys_dev", _P), ("state_dev", _P), ("logits_out", _P), ("ws", _P)]


class V2DecideArgs(C.Structure):             # amt_v2_decide_args
    _fields_ = [("tokens", _P), ("roots", _P), ("attrs", _P), ("T", _I), ("n_primer", _I), ("beam", _I), ("max_conseq_N", _I),
                ("max_conseq_chord", _I), ("temperature", _F), ("uniforms", _P), ("chord_embed", _I)]

# name -> argtypes (restype is int32 unless listed in _RESTYPES); mirrors include/amt_hip.h
SIGNATURES = {
    "amt_last_error": [],
    "amt_abi_version": [],
    "amt_create": [C.POINTER(AmtConfig), C.POINTER(_P)],
    "amt_destroy": [_P],
    "amt_load_weight": [_P, C.c_char_p, _P, _I, C.POINTER(C.c_int64)],
    "amt_finalize": [_P],
    "amt_encode": [_P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P],
    "amt_encode_resid": [_P, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P],
    "amt_set_option": [_P, C.c_char_p, _I],
    "amt_prefill": [_P, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "amt_generate_begin": [_P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P],
    "amt_generate_set_uniforms": [_P, _P, _P],
    "amt_generate_run": [_P, _I, _P, _P],
    "amt_generate_profile": [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P],
    "amt_generate_step_probs": [_P, _P, _P],
    "amt_generate_commit": [_P, _P, _P],
    "amt_generate_set_branch": [_P, _I],
    "amt_generate_end": [_P, _P, _P],
    "amt_generate": [_P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P],
    "amt_v2_decide_batch": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P],
    "amt_v2_step_decide_batch": [C.POINTER(V2StepArgs), C.POINTER(V2DecideArgs), _I, _P],
    "amt_decode_step_bytes": [_P, _I, _I, _I],
    "amt_linear_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "amt_layernorm_fwd": [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    "amt_rmsnorm_fwd": [_P, _P, _P, _I, _I, _F, _P],
    "amt_rmsnorm_resid_fwd": [_P, _P, _P, _P, _I, _I, _F, _P],
    "amt_diff_subln_fwd": [_P, _P, _P, _P, _I, _I, _F, _F, _F, _P],
    "amt_add_fwd": [_P, _P, _P, C.c_int64, _P],
    "amt_row_scale_add_fwd": [_P, _P, _P, _P, _I, _I, _P],
    "amt_rope_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "amt_rpr_attn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "amt_rpr_attn_nomask_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "amt_cross_attn_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "amt_attn_fwd": [_P, _P, _P, _P, C.POINTER(C.c_int64), _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "amt_concat_features_fwd": [_P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P],
    "amt_chord_embed_fwd": [_P] * 9 + [_I, _I, _I, _P],
    "amt_attn_decode_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "amt_decode_linear_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "amt_attn_decode_fold_fwd": [_P, _I] + [_P] * 10 + [_I, _I, _I, _I, _P, _I, _I, _I, _F, _F, _P],
    "amt_decode_gemm_ex_fwd": [C.POINTER(DecodeGemmArgs), _P],
    "amt_gqa_fwd": [_P] * 15 + [_I] * 7 + [_F, _P],
    "amt_gqa_rope_fwd": [_P] * 15 + [_I] * 7 + [_F, _P, _I, _I, _P],
    "amt_moe_scratch_floats": [_I, _I, _I, _I],
    "amt_moe_fwd": [_P] * 19 + [_I] * 4 + [_P],
    "amt_moe_topk_scratch_floats": [_I, _I, _I, _I, _I],
    "amt_moe_topk_fwd": [_P] * 19 + [_I] * 5 + [_P],
    "amt_moe_route_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "amt_glu_expert_fwd": [_P] * 9 + [_I, _I, _I, _P],
    "amt_moe_combine_fwd": [_P, _P, _P, _P, _P, _F, _P, _I, _I, _P],
    "amt_moe_ep_dispatch_plan_fwd": [_P, _I, _I, _P, _P, _P, _P, _P],
    "amt_gather_rows_fwd": [_P, _P, _P, _I, _I, _P],
    "amt_moe_ep_expert_scratch_floats": [_I, _I, _I, _I],
    "amt_moe_ep_expert_fwd": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "amt_dwconv1d_silu_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "amt_selective_scan_fwd": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "amt_concat2_fwd": [_P, _I, _P, _I, _P, _I, _I, _P],
    "amt_linear_ex_fwd": [_P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "amt_layernorm_post_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "amt_v2_step_ws_floats": [_I, _I, _I],
    "amt_pack_weight_fwd": [_P, _P, _I, _I, _P],
    "amt_v2_step": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "amt_rnn_seq_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "amt_v2_step_batch_ws_floats": [_I, _I, _I, _I],
    "amt_v2_last_step_launches": [],
    "amt_v2_step_batch": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
}
_RESTYPES = {"amt_last_error": C.c_char_p, "amt_decode_step_bytes": C.c_int64, "amt_moe_scratch_floats": C.c_int64, "amt_moe_topk_scratch_floats": C.c_int64,
             "amt_v2_step_ws_floats": C.c_int64, "amt_v2_step_batch_ws_floats": C.c_int64, "amt_moe_ep_expert_scratch_floats": C.c_int64}
_NO_STATUS = set(_RESTYPES) | {"amt_abi_version", "amt_v2_last_step_launches"}

ABI_VERSION = 3          # AMT_ABI_VERSION of include/amt_hip.h these prototypes were written against

_lib = None


def load():
    """Returns the loaded CDLL with typed prototypes; raises AmtError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled HIP runtime must be the one in the process before the extension is mapped, or the
    # extension binds to /opt/rocm's copy and the two runtimes do not see each other's device state
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AmtError(f"{LIB_PATH} is missing: the HIP extension is not built (run __graft_entry__.build()); "
                       "video2music_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.amt_abi_version.restype = C.c_int32
    have = lib.amt_abi_version()
    if have != ABI_VERSION:
        raise AmtError(f"{LIB_PATH} implements ABI version {have}, these bindings need {ABI_VERSION}: rebuild the library "
                       "(__graft_entry__.build())")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int32)
    _lib = lib
    return lib


def call(name, *args):
    """Calls an entry point and raises AmtError with the library's message on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name not in _NO_STATUS and rc != 0:
        msg = lib.amt_last_error()
        raise AmtError(f"{name} failed (status {rc}): {msg.decode() if msg else '?'}")
    return rc


def addr(t):
    """Device address of a (contiguous) tensor as a plain int / None: the form a ctypes Structure field of type c_void_p takes."""
    if t is None:
        return None
    assert t.is_contiguous(), "libamt_hip takes contiguous tensors"
    return t.data_ptr()


def ptr(t):
    """Raw device pointer of a (contiguous) torch tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "libamt_hip takes contiguous tensors"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    """The calling thread's current HIP stream as void* (torch.cuda.current_stream().cuda_stream, through the raw accessor: the
    Stream object costs ~7 us per call, which adds up in the host-bound phases -- a few hundred operator calls per generate)."""
    import torch
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:          # another torch build
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
