"""The product path's route table: which of two equivalent implementations a module takes where it has a choice.

The SHIPPED value of every entry is its default below and nothing in the product changes it: the table is read-only
outside `override()`, a context manager for tests and measurement tools that want the other branch of one switch (an A/B
timing, the unfused route on which every module boundary of the reference exists and is compared with the golden
traces, a vendor-library cross-check). One table instead of a dozen mutable module globals: a test cannot leave a
switch flipped behind it (the manager restores on exit, also after an exception), and `routes.R` shows at a glance
what a frame runs.

    from simpb_amd.plugin import routes
    with routes.override(dense=False):      # tests/test_gpu_head.py "unfused"
        ...
"""
import contextlib

DEFAULTS = dict(
    # decoder dense layers as grouped segment-input GEMMs with host-folded weights (plugin/dense.py, csrc/gemm.hip).
    # False: one GEMM per nn.Linear plus the cat/add kernels around it -- every module boundary of the reference exists.
    dense=True,
    # grouped GEMM on the FP16 matrix cores with split operands, all four partial products (x = xh + xl / 2^11 to 22 bits,
    # fp32 accumulators; csrc/gemm.hip gemm_f16x3_kernel): against float64 its error is 0.45-0.6 x that of the exact-fp32
    # matrix-core kernel on every decoder shape (profiles/r04_gemm_split_error.txt: products of f16 pairs are exact in
    # fp32 and a 16-deep step rounds once), the whole GPU suite is green on it (profiles/r04_gputest_split.log: golden
    # streams, oracle, runners), decoder graph 2.06 -> 1.96 ms. Shipped since round 4; False: the v_mfma_f32_32x32x2_f32
    # kernel. (Rounds 1-3 left it off: an earlier three-term form moved one 2D query of the golden R50 stream across an
    # image border -- any re-rounding can; the four-term form does not on any fixture.)
    gemm_split_fp16=True,
    # attention core on the FP16 matrix cores with split operands (csrc/attention.hip attention_halfs_kernel): the
    # projections in front of it leave q / k / v as (hi, lo) half pairs in each element's own 32-bit word (csrc/gemm.hip
    # out_fmt, softmax scale folded into the query rows), S = three partial products, O = three, fp32 softmax in base 2.
    # Same bound against float64 as the exact kernel (tests/test_gpu_ops.py), whole GPU suite green on it
    # (profiles/r04_gputest_att.log); 900 x 900 x 8 heads 32 -> 20 us, decoder graph 1.95 -> 1.83 ms. Shipped since round 4.
    # False: the exact-fp32 v_mfma_f32_32x32x2_f32 kernel on fp32 operands.
    attention_split_fp16=True,
    # MLP chains: 4-row workgroups on the 4x4 matrix blocks with k4-packed weights (csrc/mlp_chain.hip). False +
    # chain_transposed False: the 16-row matrix-core kernel on the weights as stored; chain_transposed: the VALU kernel.
    chain_rows4=True,
    chain_transposed=False,
    # value_proj (group_attn.py:176) on the FP16 matrix cores with split operands. False: exact-fp32 kernel, 3.5x slower.
    split_value_proj=True,
    # decode_with2d's fixed-shape records by csrc/decode.hip (two launches). False: the PyTorch statement (~40 launches).
    fused_decode=True,
    # InstanceBank on its persistent state through csrc/bank.hip. False: the PyTorch statement of instance_bank.py.
    fused_bank=True,
    # ... its frame-end commit with one workgroup per stream for a batch of streams (they meet once inside the launch on an
    # arrival counter: csrc/bank.hip bank_cache_streams_kernel). False: one workgroup walks the streams (13 us each).
    bank_cache_per_stream=True,
    # fork the value branch of an attention operator onto a side stream (measured slower inside a replayed graph)
    parallel_branches=False,
    # own 1x1 / 3x3 convolutions and stem epilogue (csrc/conv1x1.hip, conv3x3.hip, bias_act.hip). False: vendor
    # convolutions (mmdet's statement of ResNet / FPN), the cross-check of tests/test_dense.py.
    conv1x1_kernel=True,
    conv3x3_kernel=True,
    stem_epilogue_kernel=True,
    # own 7x7 / stride-2 stem convolution (csrc/stem.hip). False: the vendor convolution.
    stem_kernel=True,
    # the FPN's four output convolutions as ONE launch (conv_staged_group_kernel): the three small levels' tiles fill the last
    # round of the large level's instead of three nearly empty launches of their own. False: one launch per level.
    fpn_grouped_out=True,
    # static-capacity allocation as one three-kernel entry point. False: the stepwise entry points.
    alloc_static_fused=True,
    # QueryGroupMultiScaleDeformableAttention without value_proj over the 89 760 camera tokens: sample the raw tokens per
    # (query, head), project the 8 x 256 sums with the folded W_out . W_value afterwards (csrc/msda_lin.hip; linearity).
    # False: value_proj over every token (with the backbone, runner.precompute_values) + the sampler on its output.
    msda_linear=True,
    # the `norm` operator in front of a refinement head runs inside the head's chain launch (leading LayerNorm stage of
    # csrc/mlp_chain.hip's 4-row kernel, which also writes the operator's output). False: a LayerNorm launch of its own.
    norm_in_refine=True,
    # DeformableFeatureAggregation: key points + projection + weight softmax inside the aggregation launch
    # (csrc/deform_agg_fused.hip). False: dfa_points + dfa_weights + the drop-in aggregation operator (three launches).
    fused_dfa=True,
    # ... reading the f16 copy of the camera tokens the FPN leaves beside the fp32 rows (same numbers, half the bytes).
    # Off: the launch is latency-bound at 900 anchors (26.5-27.4 us with f16 rows against 28.3-30.4 with fp32 rows,
    # profiles/r03_*), so the fp32 rows of the operator's own contract (ops/src/deformable_aggregation.cpp:22-28) stay.
    dfa_f16_tokens=False,
)


class _Routes:
    __slots__ = tuple(DEFAULTS) + ("_open",)

    def __init__(self):
        object.__setattr__(self, "_open", False)
        for k, v in DEFAULTS.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, name, value):
        if not self._open:
            raise AttributeError(f"routes.R.{name} is read-only: use `with routes.override({name}=...)` (tests / tools only)")
        if name not in DEFAULTS:
            raise AttributeError(f"no route named {name!r}")
        object.__setattr__(self, name, bool(value))

    def __repr__(self):
        return "Routes(" + ", ".join(f"{k}={getattr(self, k)}" for k in DEFAULTS) + ")"


R = _Routes()


@contextlib.contextmanager
def override(**switches):
    """Take the other branch of the named switches inside the block (tests and measurement tools only)."""
    unknown = [k for k in switches if k not in DEFAULTS]
    if unknown:
        raise KeyError(f"no such route(s): {unknown}")
    old = {k: getattr(R, k) for k in switches}
    object.__setattr__(R, "_open", True)
    try:
        for k, v in switches.items():
            setattr(R, k, v)
        object.__setattr__(R, "_open", False)
        yield R
    finally:
        object.__setattr__(R, "_open", True)
        for k, v in old.items():
            setattr(R, k, v)
        object.__setattr__(R, "_open", False)
