"""ctypes binding of libpg_hip.so (the C-ABI declared in include/pg_hip.h).

The product path has NO fallback: if the shared library is missing or a call returns a
non-zero status, a RuntimeError/ValueError is raised. (The CPU oracle under /oracle is test
infrastructure and is never imported from here.)
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PG_HIP_LIB=<path>: load an alternative build of the same C-ABI (experiment variants of build.py: PG_VARIANT / PG_ABLATE) — still a
# HIP library, still loaded or the import fails; the default is the production library
LIB_PATH = os.environ.get("PG_HIP_LIB") or os.path.join(_HERE, "lib", "libpg_hip.so")

ABI_VERSION = 1

c_f = ctypes.c_void_p  # device float* (passed as integer address)
c_i = ctypes.c_int
c_l = ctypes.c_long
c_z = ctypes.c_size_t
c_s = ctypes.c_void_p  # hipStream_t
c_ip = ctypes.POINTER(ctypes.c_int)  # host int array
c_flt = ctypes.c_float

# name -> (restype, argtypes); mirrors include/pg_hip.h one to one.
SIGNATURES = {
    "pg_abi_version": (c_i, []),
    "pg_last_error": (ctypes.c_char_p, []),
    "pg_conv2d_taps": (
        c_i,
        [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_i, c_f, c_i,
         c_s],
    ),
    "pg_conv2d_mfma": (
        c_i,
        [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_i, c_f, c_i,
         c_i, c_i, c_s],
    ),
    "pg_conv2d_mfma_ex": (
        c_i,
        [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_i, c_f, c_i,
         c_i, c_i, c_f, c_l, c_l, c_s],
    ),
    "pg_conv_dual_ok": (c_i, [c_i] * 4),
    "pg_conv2d_mfma_dual": (c_i, [c_f] * 4 + [c_i] * 5 + [c_f, c_i, c_f, c_s]),
    "pg_conv_gate_fusable": (c_i, [c_i, c_i, c_i, c_i, c_i, c_ip, c_ip]),
    "pg_conv2d_mfma_gate": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_i, c_i, c_f, c_f, c_s]),
    "pg_conv_mfma_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "pg_conv_frag_floats": (c_z, [c_i, c_i, c_i, c_i]),
    "pg_pack_conv_weight_frag": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_i, c_i, c_s]),
    "pg_pack_conv_weight_frag2": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_i, c_i, c_s]),
    "pg_pack_conv_weight": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_i, c_i, c_s]),
    "pg_packed_weight_floats": (c_z, [c_i, c_i, c_i]),
    "pg_conv_b_pad": (c_i, [c_i]),
    "pg_conv2d_wgrad": (
        c_i,
        [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ip, c_ip, c_ip,
         c_ip, c_i, c_f, c_z, c_s],
    ),
    "pg_conv2d_wgrad_workspace_floats": (c_z, [c_i, c_i, c_i]),
    "pg_mul_inplace": (c_i, [c_f, c_f, c_z, c_s]),
    "pg_nchw_layernorm_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_flt, c_s]),
    "pg_nchw_layernorm_bwd": (
        c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_z, c_s]),
    "pg_gpt_block_head_fwd": (c_i, [c_f] * 8 + [c_i, c_i, c_i, c_flt, c_s]),
    "pg_gpt_block_head_bwd": (c_i, [c_f] * 14 + [c_i, c_i, c_i, c_flt, c_f, c_z, c_s]),
    "pg_gpt_block_head_bwd_workspace_floats": (c_z, [c_i, c_i]),
    "pg_gpt_block_head_bwd_partial": (c_i, [c_f] * 8 + [c_i, c_i, c_i, c_flt, c_f, c_z, c_s]),
    "pg_gpt_blocks_reduce": (c_i, [c_i, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.POINTER(ctypes.c_void_p), c_i, c_i, c_i, c_s]),
    "pg_gpt_block_tail_fwd": (c_i, [c_f] * 11 + [c_i, c_i, c_i, c_i, c_flt, c_s]),
    "pg_gpt_block_tail_bwd": (c_i, [c_f] * 20 + [c_i, c_i, c_i, c_i, c_flt, c_f, c_z, c_s]),
    "pg_gpt_block_tail_bwd_workspace_floats": (c_z, [c_i, c_i]),
    "pg_gpt_block_tail_bwd_partial": (c_i, [c_f] * 12 + [c_i, c_i, c_i, c_i, c_flt, c_f, c_z, c_s]),
    "pg_gpt_block_head_bwd_with_tail": (c_i, [c_f] * 14 + [c_i, c_i, c_i, c_flt, c_f, c_z] + [c_f] * 9 + [c_s]),
    "pg_sample_embed": (c_i, [c_f] * 5 + [c_i] * 10 + [c_f, c_s]),
    "pg_attn_decode": (c_i, [c_f] * 4 + [c_i] * 8 + [c_f, c_s]),
    "pg_mlp_gelu_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_s]),
    "pg_mlp_gelu_bwd": (
        c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_z, c_s]),
    "pg_mlp_gelu_bwd_workspace_floats": (c_z, [c_i, c_i]),
    "pg_nchw_layernorm_bwd_res": (
        c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_z, c_s]),
    "pg_nchw_layernorm_bwd_workspace_floats": (c_z, [c_i, c_i, c_i]),
    "pg_causal_attn_fwd": (
        c_i,
        [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l, c_l, c_i, c_s],
    ),
    "pg_causal_attn_bwd": (
        c_i,
        [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l,
         c_l, c_l, c_l, c_l, c_l, c_i, c_s],
    ),
    "pg_causal_attn_bwd_dq": (
        c_i,
        [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l,
         c_l, c_l, c_l, c_l, c_l, c_i, c_s],
    ),
    "pg_causal_attn_bwd_dkv": (
        c_i,
        [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_l,
         c_l, c_l, c_l, c_l, c_l, c_i, c_s],
    ),
    "pg_image_positional_encoding": (c_i, [c_f, c_i, c_i, c_i, c_s]),
    "pg_act_fwd": (c_i, [c_f, c_f, c_z, c_i, c_s]),
    "pg_act_bwd": (c_i, [c_f, c_f, c_f, c_z, c_i, c_s]),
    "pg_gated_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_s]),
    "pg_gated_bwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_s]),
    "pg_gated_fwd_res": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_s]),
    "pg_act_bwd_from_out": (c_i, [c_f, c_f, c_f, c_f, c_z, c_i, c_s]),
    "pg_add": (c_i, [c_f, c_f, c_f, c_z, c_s]),
    "pg_fill": (c_i, [c_f, c_flt, c_z, c_s]),
    "pg_phase_merge4": (c_i, [c_f, ctypes.POINTER(ctypes.c_void_p), c_i, c_i, c_i, c_i, c_s]),
    "pg_phase_weights": (c_i, [c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_phase_weights_bwd": (c_i, [ctypes.POINTER(ctypes.c_void_p), c_f, c_i, c_i, c_i, c_i, c_s]),
    "pg_sum_rows": (c_i, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long), c_i, c_f, c_l, c_l, c_s]),
    "pg_avgpool2_bwd_res": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_add_bcast_fwd": (c_i, [c_f, c_f, c_f, c_i, c_z, c_s]),
    "pg_add_bcast_bwd": (c_i, [c_f, c_f, c_i, c_z, c_s]),
    "pg_vq_assign": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_s]),
    "pg_vq_ema_update": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, c_s]),
    "pg_vq_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_z, c_s]),
    "pg_mse_fwd": (c_i, [c_f, c_f, c_f, c_z, c_s]),
    "pg_mse_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_z, c_s]),
    "pg_bce_logits_fwd": (c_i, [c_f, c_f, c_f, c_i, c_z, c_s]),
    "pg_bce_logits_bwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_z, c_s]),
    "pg_avgpool2_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_avgpool2_bwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_upsample2_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_upsample2_bwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_phase_split2": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_s]),
    "pg_gauss_head_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_l, c_l, c_i, c_s]),
    "pg_gauss_head_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_l, c_l, c_i, c_s]),
    "pg_vec_mean_accum": (c_i, [c_f, c_i, c_f, c_s]),
    "pg_fill_scaled": (c_i, [c_f, c_flt, c_f, c_i, c_s]),
    "pg_sumsq_accum": (c_i, [c_f, c_z, c_f, c_s]),
    "pg_adam_prepare": (c_i, [c_f, c_s]),
    "pg_adam_step": (c_i, [c_f, c_f, c_f, c_f, c_z, c_f, c_flt, c_flt, c_flt, c_s]),
    "pg_attn_fused_bwd": (c_i, [c_i]),
    "pg_concat_elu_fwd": (c_i, [c_f, c_f, c_i, c_l, c_s]),
    "pg_concat_elu_bwd": (c_i, [c_f, c_f, c_f, c_i, c_l, c_s]),
    "pg_dmol_fwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_dmol_bwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_s]),
    "pg_copy_rows": (c_i, [c_f, c_f, c_l, c_l, c_l, c_l, c_i, c_s]),
    "pg_comm_unique_id": (c_i, [ctypes.c_char_p]),
    "pg_comm_init": (c_i, [c_i, c_i, ctypes.c_char_p]),
    "pg_comm_world": (c_i, []),
    "pg_comm_rccl_version": (c_i, []),
    "pg_allreduce_sum": (c_i, [c_f, c_z, c_i, c_s]),
    "pg_broadcast": (c_i, [c_f, c_z, c_i, c_i, c_s]),
    "pg_comm_destroy": (c_i, []),
}
COMM_ID_BYTES = 128
DTYPE_F32 = 0

_lib = None


def load():
    """Loads libpg_hip.so (once) and binds every symbol of the header. Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libpg_hip.so not found at {LIB_PATH}. Build it with "
            "`python pytorch-generative_amd/build.py` (or __graft_entry__.build()). "
            "There is no CPU/ATen fallback for the HIP operator path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.pg_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libpg_hip.so ABI version {got} != expected {ABI_VERSION}")
    if os.environ.get("PG_TRACE"):
        lib = _Traced(lib, os.environ["PG_TRACE"])
    _lib = lib
    return lib


class _Traced:
    """PG_TRACE=<path prefix>: one line per C-ABI call (name + scalar arguments, pointers in hex), flushed BEFORE the
    call — with AMD_SERIALIZE_KERNEL=3 the last line of <prefix>.<pid> names the launch a GPU fault came from."""

    def __init__(self, lib, prefix):
        self._lib = lib
        self._out = open(f"{prefix}.{os.getpid()}", "a", buffering=1)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("pg_") or name in ("pg_last_error", "pg_abi_version"):
            return fn
        out = self._out

        def call(*args):
            shown = []
            for a in args:
                if isinstance(a, int):
                    shown.append(hex(a) if a > 1 << 32 else str(a))
                elif isinstance(a, ctypes.Array):
                    shown.append(str(list(a)))
                else:
                    shown.append(repr(a))
            out.write(f"{name}({', '.join(shown)})\n")
            return fn(*args)

        setattr(self, name, call)
        return call


def check(rc, what):
    """Maps a C-ABI status to the Python exceptions of the reference's error convention."""
    if rc == 0:
        return
    msg = load().pg_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg} (status {rc})")
    if rc >= 1000:
        raise RuntimeError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: HIP error {rc}: {msg}")


def int_array(values):
    return (ctypes.c_int * len(values))(*values)
