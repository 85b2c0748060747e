"""torch.autograd bindings of the HIP kernels (the only arithmetic on the hot path).

Every function here launches kernels from libpg_hip.so on torch's *current* HIP stream with
raw device pointers (so a whole training step can be captured into a hipGraph). PyTorch is
used for memory (caching allocator), streams and the autograd tape only.

Weight gradients: if a parameter carries a `_pg_grad` tensor (a view into the trainer's flat
gradient buffer, zeroed once per step) the backward kernels accumulate straight into it with
fp32 atomics and autograd sees `None` for that parameter — no per-parameter AccumulateGrad
kernels, and the flat buffer is what RCCL all-reduces. Without `_pg_grad` the usual
`param.grad` protocol is followed.
"""

# One module per kernel family (round 6: this was a single 1 900-line file); the package re-exports every name, so that
# `from pytorch_generative_amd import ops; ops.conv2d_taps(...)` and the tests' / tools' uses of the helpers are unchanged.
from pytorch_generative_amd.ops._common import (  # noqa: F401
    ACT_NONE,
    ACT_RELU,
    ACT_ELU,
    ACT_GELU,
    ACT_ELU_OUT,
    GATE_TANH,
    GATE_IDENTITY,
    _ACT_IDS,
    _stream,
    RowDecode,
    _chk,
    _p,
    zeros,
    zeros_like,
    _sink,
    CONV_FMT_F32,
    CONV_FMT_B3,
    CONV_FMT_B3_GATE,
    FUSE_SKIP,
    _dense_per_image,
)
from pytorch_generative_amd.ops.elementwise import (  # noqa: F401
    _ConcatChannels,
    concat_channels,
    _Act,
    relu,
    elu,
    gelu,
    _Gated,
    gated_activation,
    _Add,
    add,
    _AddBcast,
    add_broadcast_batch,
    image_positional_encoding,
    mul_inplace_,
    _sum_into,
    _SumVectors,
    _Fanout,
    fanout,
    sum_vectors,
)
from pytorch_generative_amd.ops.conv import (  # noqa: F401
    ConvSpec,
    CONV_MFMA,
    _use_mfma,
    _pack_frag,
    _pack_frag_both,
    _pack,
    _ConvTaps,
    FUSE_PAIR,
    FUSE_LNSKIP,
    FUSE_QKV_EXTRA,
    _adjacent_view,
    conv_pair_views,
    _ConvPair,
    conv2d_pair,
    conv_mfma_ok,
    conv_two_residuals_ok,
    conv2d_taps,
    conv_gate_ok,
    conv_dual_ok,
    GradSlot,
    FUSE_DUAL,
    FUSE_GATE,
)
from pytorch_generative_amd.ops.gpt_block import (  # noqa: F401
    FUSE_MLP,
    mlp_gelu_supported,
    _MlpGelu,
    mlp_gelu,
    FUSE_BLOCK,
    DEFER_BLOCK_REDUCE,
    _grad_targets,
    new_block_chain,
    assert_no_pending_block_reductions,
    flush_block_reductions,
    _GPTBlockHead,
    _GPTBlockTail,
    gpt_block_supported,
    gpt_block_head,
    gpt_block_tail,
    _NCHWLayerNorm,
    nchw_layernorm,
    nchw_layernorm_skip,
)
from pytorch_generative_amd.ops.attention import (  # noqa: F401
    _CausalAttention,
    _CausalAttentionQKV,
    _MergeQKVWeight,
    merge_qkv_weight,
    set_deterministic,
    causal_attention_qkv,
    attention_dims_native,
    _pad16,
    causal_attention,
)
from pytorch_generative_amd.ops.losses import (  # noqa: F401
    _BCEWithLogitsSumMean,
    _DmolLossSumMean,
    dmol_loss_sum_mean,
    bce_with_logits_sum_mean,
    _ElboMean,
    elbo_terms,
)
from pytorch_generative_amd.ops.vae import (  # noqa: F401
    _AvgPool2,
    avg_pool2,
    _Upsample2,
    upsample2_nearest,
    _GaussHead,
    gaussian_head_unit,
    gaussian_head_pair,
    gaussian_head_prior,
    _PhaseSplit,
    _PhaseMerge,
    _PhaseWeights,
    _SplitInChannels,
    split_in_channels,
    phase_weights,
    phase_split,
    phase_merge,
    _PhaseMerge4,
    _PhaseSplit4,
    phase_split4,
    phase_merge4,
    _ConcatElu,
    concat_elu,
    _Resample2,
    subsample2,
    zero_insert2,
)
