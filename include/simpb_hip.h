/* C-ABI of the MI355X (gfx950) kernels behind SimPB's hybrid 2D/3D decoder hot path.
 *
 * Plain pointers and sizes only: every pointer is a DEVICE pointer owned by the caller, the
 * callee never allocates or frees, every output buffer is fully overwritten, and `stream` is a
 * hipStream_t (NULL = the legacy default stream). Each function returns 0 on success or one of
 * the SIMPB_E* codes; nothing is thrown, nothing is printed.
 *
 * Citations are file:line under /root/reference/projects/mmdet3d_plugin/.
 */
#ifndef SIMPB_HIP_H
#define SIMPB_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SIMPB_OK 0
#define SIMPB_EINVAL 1   /* null pointer, non-positive size, or a layout the kernels cannot take */
#define SIMPB_ELAUNCH 2  /* hipGetLastError() after the launch was not hipSuccess */

#define SIMPB_RECORD3D_WIDTH 15 /* columns of the 3D detection record (simpb_decode3d_record) */

/* Library/ABI version (bumped when a signature changes). */
int simpb_abi_version(void);

/* Text of the HIP error behind the last SIMPB_ELAUNCH on the calling thread ("" if none). */
const char* simpb_last_error(void);

/* Optional per-launch timing for measurement (bench.py): while enabled, the sampler entry points
 * record a HIP event pair on their launch stream directly around the kernel launch.
 *   simpb_timing_enable(n)  allocate n event pairs (0 = disable and free); not thread-safe
 *   simpb_timing_read(id, ms_out, max_n)  waits for the recorded pairs of kernel `id`, writes their
 *                           elapsed milliseconds, returns how many (or -1)
 *   simpb_timing_reset()    forget the recorded pairs, keep the allocation */
#define SIMPB_KERNEL_DAF 1
#define SIMPB_KERNEL_MSDA 2
int simpb_timing_enable(int capacity);
int simpb_timing_read(int kernel_id, float* ms_out, int max_n);
void simpb_timing_reset(void);

/* Replaces `deformable_aggregation(...)` (ops/src/deformable_aggregation.cpp:4-19, launcher
 * ops/src/deformable_aggregation_cuda.cu:265-288, kernel :129-187); same argument order plus the
 * stream. Layouts (deformable_aggregation.cpp:22-28):
 *   mc_ms_feat        f32 [batch_size, num_feat, num_embeds]      channel innermost
 *   spatial_shape     i32 [num_cams, num_scale, 2] = (H, W)
 *   scale_start_index i32 [num_cams, num_scale]                  token offset inside num_feat
 *   sample_location   f32 [batch_size, num_anchors, num_pts, num_cams, 2] = (x/W_img, y/H_img)
 *   weights           f32 [batch_size, num_anchors, num_pts, num_cams, num_scale, num_groups]
 *   output            f32 [batch_size, num_anchors, num_embeds]  overwritten (no pre-zero needed)
 * num_embeds % num_groups must be 0. Results are deterministic (no atomics). */
int simpb_deformable_aggregation_forward(
    float* output, const float* mc_ms_feat, const int* spatial_shape, const int* scale_start_index,
    const float* sample_location, const float* weights,
    int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
    int num_anchors, int num_pts, int num_groups, void* stream);

/* DeformableFeatureAggregation.forward between its Linear layers in ONE launch (models/blocks.py:110-162): key points
 * (models/detection3d/blocks.py:181-222) -> project_points (blocks.py:198-213) -> softmax of _get_weights (:177-187) ->
 * the aggregation above. The workgroup that aggregates an anchor computes its num_pts x num_cams sampling locations and
 * its softmax weights on chip, so neither tensor exists in memory (simpb_dfa_points + simpb_dfa_weights +
 * simpb_deformable_aggregation_forward are the same arithmetic in three launches; tests compare the two).
 *   mc_ms_feat   f32 (feat_is_f16 = 0) or f16 (1) [batch_size, num_feat, num_embeds]; the f16 form is for tokens that are
 *                f16 numbers anyway (the fp16 backbone's output, simpb.py:63): same results, half the gather bytes
 *   anchor f32 [bs, A, 11]; learnable f32 [bs, A, num_learn * 3] (raw learnable_fc output); fix_scale f32 [num_fix, 3]
 *   projection_mat f32 [bs, cams, 4, 4]; image_wh f32 [bs, cams, 2]
 *   feat_logits f32 [bs, A, num_scale * num_pts * num_groups] = weights_fc(feature + anchor_embed)
 *   cam_logits  f32 [bs, cams, num_scale * num_pts * num_groups] = camera_embed . weights_fc.weight^T
 *   loc_out / weights_out: optional (NULL) copies of the two on-chip tensors in the layouts of the operator above
 * Supported layout: num_embeds <= 256, (num_embeds / num_groups) % 4 == 0, num_groups a power of two <= 64,
 * num_pts * num_cams <= 128, cams * num_scale * num_pts * num_groups <= 4096; anything else returns SIMPB_EINVAL. */
int simpb_dfa_fused_forward(
    float* output, const void* mc_ms_feat, int feat_is_f16, const int* spatial_shape, const int* scale_start_index,
    const float* anchor, const float* learnable, const float* fix_scale, const float* projection_mat, const float* image_wh,
    const float* feat_logits, const float* cam_logits, float* loc_out, float* weights_out, int batch_size, int num_cams,
    int num_feat, int num_embeds, int num_scale, int num_anchors, int num_fix, int num_learn, int num_groups, void* stream);

/* Replaces `deformable_aggregation_grad(...)` (ops/src/deformable_aggregation.cpp:64-84, launcher
 * ops/src/deformable_aggregation_cuda.cu:291-318, kernels :62-126,190-262), the backward of the
 * operator (ops/deformable_aggregation.py:39-75). Layouts as the forward; grad_output f32
 * [batch_size, num_anchors, num_embeds]. All three gradients are fully written (the callee clears
 * grad_mc_ms_feat itself; the caller need not pre-zero anything). grad_weights and
 * grad_sampling_location are deterministic; grad_mc_ms_feat is accumulated with float atomics.
 * Supported layout: num_embeds % 64 == 0, num_embeds <= 256, (num_embeds / num_groups) % 32 == 0,
 * num_pts * num_cams <= 128 (the shipped 256 / 8 / 13 x 6); anything else returns SIMPB_EINVAL. */
int simpb_deformable_aggregation_backward(
    float* grad_mc_ms_feat, float* grad_sampling_location, float* grad_weights, const float* mc_ms_feat,
    const int* spatial_shape, const int* scale_start_index, const float* sample_location, const float* weights,
    const float* grad_output, int batch_size, int num_cams, int num_feat, int num_embeds, int num_scale,
    int num_anchors, int num_pts, int num_groups, void* stream);

/* Replaces the per-camera loop over mmcv's `ms_deform_attn_forward` in
 * QueryGroupMultiScaleDeformableAttention.forward (models/group_attn.py:222-235): ONE launch for
 * all camera groups. Sampling rule = mmcv's CUDA op = grid_sample(bilinear, zeros,
 * align_corners=False).
 *   value          f32 [batch_size, num_cams, num_value, num_heads, channels]
 *   spatial_shapes i64 [num_levels, 2] = (H, W)      (group_attn.py passes torch.long)
 *   level_start    i64 [num_levels]
 *   sampling_loc   f32 [batch_size, num_query, num_heads, num_levels, num_points, 2] = (x, y) in [0,1]
 *   attn_weight    f32 [batch_size, num_query, num_heads, num_levels, num_points]
 *   query_cam      i32 [num_query]   camera group of each query slot (what query_groups encodes)
 *   output         f32 [batch_size, num_query, num_heads * channels]  overwritten */
int simpb_ms_deform_attn_grouped_forward(
    float* output, const float* value, const long long* spatial_shapes, const long long* level_start,
    const float* sampling_loc, const float* attn_weight, const int* query_cam,
    int batch_size, int num_cams, int num_value, int num_heads, int channels,
    int num_levels, int num_points, int num_query, void* stream);

/* The same operator WITHOUT value_proj over the camera tokens (models/group_attn.py:176-179): sampling and the value
 * projection are both linear, so for head h of a query
 *   sum_s a_s bilinear(W_h x + b_h)(s) = W_h . [sum_s a_s bilinear(x)(s)] + b_h [sum_s a_s (valid tap weights of s)]
 * (the reference zero-pads the PROJECTED map, hence the second bracket). This entry computes the brackets from the raw
 * tokens; the caller applies W_h (folded with output_proj) as one query-sized product over agg. Softmax of the attention
 * logits and reference point + offset / (W_l, H_l) (group_attn.py:181-201) are done here as well.
 *   agg        f32 [batch_size, num_query, ld_agg >= 8 * 256 + 128]: per query [head][256 channel sums] | 8 tap-weight
 *              sums | zeros up to 128; rows of capacity slots (query_cam < 0 or >= *m_live) are NOT written
 *   tokens     f16 (tokens_are_f16 = 1) or f32 [batch_size, num_cams, num_value, 256]: the camera tokens themselves
 *   raw        f32 [batch_size * num_query, ld_raw]: sampling_offsets (heads x levels x points x 2) | attention logits
 *   ref        f32 [batch_size * num_query, ld_ref >= 2]: reference point (x, y) in [0, 1]
 * Compiled for the shipped layout (8 heads x 32 channels, 4 levels, 4 points); anything else returns SIMPB_EINVAL and
 * the caller uses value_proj + simpb_ms_deform_attn_grouped_forward. */
int simpb_msda_linear_forward(float* agg, int ld_agg, const void* tokens, int tokens_are_f16,
                              const long long* spatial_shapes, const long long* level_start, const float* raw,
                              int ld_raw, const float* ref, int ld_ref, const int* query_cam, const int* m_live,
                              int batch_size, int num_cams, int num_value, int num_heads, int channels,
                              int num_levels, int num_points, int num_query, void* stream);

/* Backward of simpb_ms_deform_attn_grouped_forward (what mmcv's ms_deform_attn_backward does per camera
 * group behind models/group_attn.py:227-235). grad_output f32 [bs, num_query, heads*channels]; all three
 * gradients are fully written (grad_value is cleared by the callee and accumulated with float atomics;
 * the other two are deterministic). Supported layout: heads * channels == 256, channels/4 a power of two. */
int simpb_ms_deform_attn_grouped_backward(
    float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, const float* value,
    const long long* spatial_shapes, const long long* level_start, const float* sampling_loc, const float* attn_weight,
    const int* query_cam, const float* grad_output, int batch_size, int num_cams, int num_value, int num_heads,
    int channels, int num_levels, int num_points, int num_query, void* stream);

/* y[M, N] = x[M, K] . weight[N, K]^T + bias[N] (bias may be NULL), optional ReLU; exact fp32 on the
 * f32 matrix cores. Replaces nn.Linear where the reference runs it over every camera token:
 * value_proj in QueryGroupMultiScaleDeformableAttention.forward (models/group_attn.py:176), and the
 * other large-M Linear layers of the head. Row-major, K contiguous; K % 32 == 0; x and weight
 * 16-byte aligned. */
int simpb_linear_f32(float* y, const float* x, const float* weight, const float* bias, int M, int N, int K,
                     int relu, void* stream);

/* Same product on the FP16 matrix cores at fp32-grade accuracy: weight = weight_hi + weight_lo / 2048 with both parts
 * f16 [N, K] (weight_hi = half(weight), weight_lo = half((weight - weight_hi) * 2048), prepared once by the caller);
 * x is split the same way while it is staged; y = x_hi.W_hi^T + (x_hi.W_lo^T + x_lo.W_hi^T) / 2048 + bias in fp32
 * accumulators (the dropped x_lo.W_lo^T term is ~2^-22 relative). Requires |x|, |weight| < 65504. Used for value_proj
 * (models/group_attn.py:176), where the exact kernel above is compute-bound. K % 32 == 0; 16-byte aligned. */
int simpb_linear_f16x3(float* y, const float* x, const void* weight_hi, const void* weight_lo, const float* bias, int M,
                       int N, int K, void* stream);

/* The same product for an x that is already half precision (f16 [M, K]; e.g. the camera tokens as the fp16 backbone
 * produced them, simpb.py:63): its trailing part is zero, so y = x.W_hi^T + (x.W_lo^T) / 2048 + bias in two passes with the
 * error of the three-pass form, half the bytes of x and no splitting arithmetic. K % 64 == 0; 16-byte aligned. */
int simpb_linear_f16in_split(float* y, const void* x_f16, const void* weight_hi, const void* weight_lo, const float* bias,
                             int M, int N, int K, void* stream);

/* ResNet stem in one launch: conv 7x7 / stride 2 / padding 3 (3 -> 64, BN folded) + bias + ReLU + max_pool 3x3 / stride 2 /
 * padding 1 = mmdet ResNet.forward's maxpool(relu(bn1(conv1(x)))) (config :79-99 after tools/fuse_conv_bn.py:10-48).
 *   img_nhwc4      f16 [num_images, height, width, 4]  RGB + a zero channel (simpb_image_to_nhwc4_f16 makes it)
 *   weight_packed  f16 [28][2][64][4]: entry [ky * 4 + g][kb][n][c] = weight[n][c][ky][2 g + kb] (0 for c = 3 or tap 7)
 *   bias f16 [64];  out f16 [num_images, Hp, Wp, 64] (NHWC), Hp = ((height + 6 - 7) / 2 + 1 + 2 - 3) / 2 + 1
 * Conv results are rounded to f16 before bias / ReLU / maximum (fp32), i.e. what a stored f16 conv map + the fused
 * epilogue (simpb_bias_relu_maxpool_nhwc_f16) give. out_channels must be 64. */
int simpb_stem_conv7x7_pool_f16(void* out, const void* img_nhwc4, const void* weight_packed, const void* bias,
                                int num_images, int height, int width, int out_channels, void* stream);

/* f32 image with arbitrary element strides [num_images, channels <= 3, height, width] -> f16 [N, H, W, 4], missing
 * channels zero: the cast + channels_last copy in front of the stem. */
int simpb_image_to_nhwc4_f16(void* out, const float* img, long long stride_n, long long stride_c, long long stride_h,
                             long long stride_w, int num_images, int channels, int height, int width, void* stream);

/* Grouped small GEMM of the decoder: up to 4 independent problems per launch, each
 *   y[M, 0:N] (row stride ldy) = relu?( [x0 | x1 | ...][M, K] . w[N, K]^T (row stride ldw) + bias[N] )
 * where x is given as up to 4 column segments (pointer, row stride, width; widths sum to K). This is
 * how the reference's `torch.cat([feature, pos_embed], -1)` in front of every attention projection
 * (models/simpb_head.py:298-321), the `identity + out` branches followed by fc_after (:306-310) and
 * the identity_fc branch of AsymmetricFFN (models/blocks.py:384-393) become one product each, with
 * host-folded weights, instead of a concatenation, two vendor GEMMs and an add. Exact fp32
 * (v_mfma_f32_32x32x2_f32). Every segment width and K are multiples of 64; x, w 16-byte aligned,
 * row strides multiples of 4. m_live (device int, may be NULL): rows >= *m_live are capacity slots of the
 * static 2D query set and are written as zeros. Deterministic. */
#define SIMPB_GEMM_MAX_SEGS 4
#define SIMPB_GEMM_MAX_JOBS 4
#define SIMPB_GEMM_OUT_F32 0
#define SIMPB_GEMM_OUT_SPLIT_HALFS 1
typedef struct simpb_gemm_job {
  const float* x[SIMPB_GEMM_MAX_SEGS];
  int ldx[SIMPB_GEMM_MAX_SEGS];
  int kseg[SIMPB_GEMM_MAX_SEGS];
  int num_seg;
  int M, N, K;
  const float* w;
  const float* bias;
  float* y;
  const int* m_live;
  int ldw, ldy, relu;
  /* SIMPB_GEMM_OUT_F32 (0): y holds fp32 numbers. SIMPB_GEMM_OUT_SPLIT_HALFS (1): every element is written as the two halfs
   * the split-operand attention kernel multiplies (simpb_attention_split_halfs): low 16 bits half(v), high 16 bits
   * half((v - half(v)) * 2^11), in the element's own 32-bit word (ldy unchanged). Needs |v| < 65504. */
  int out_fmt;
  /* optional rank-1 term before the ReLU: rows with row_flag[row] != 0 also get bias2[col]. This is
   * the 257th input column of ReWeight.reduce (models/aggregation.py:19-21,71-72: Linear over
   * cat(query2d, is_center)) without materialising the concatenation. Both NULL = absent. */
  const int* row_flag;
  const float* bias2;
  /* optional pre-split weights, f16 [N, K] with dense rows: w_hi = half(w), w_lo = half((w - w_hi) * 2048). When every
   * job of a launch brings them (and its segment widths are multiples of 128) the products run on the FP16 matrix
   * cores in three passes at fp32-grade accuracy (see simpb_linear_f16x3); `w` stays the reference copy. Needs
   * |x|, |w| < 65504. Both NULL = exact fp32 path. */
  const void* w_hi;
  const void* w_lo;
} simpb_gemm_job;
typedef struct simpb_gemm_args {
  int num_jobs;
  int reserved;
  simpb_gemm_job job[SIMPB_GEMM_MAX_JOBS];
} simpb_gemm_args;
int simpb_gemm_f32(const simpb_gemm_args* args, void* stream);

/* out[M, 0:k0+k1] (row stride ldo) = LayerNorm([x0 | x1]) * gamma + beta over the concatenated width
 * (torch.nn.LayerNorm: eps 1e-5, biased variance); x1 may be NULL with k1 = 0. Width <= 512. The two
 * segments are the `cat` of residual_mode="cat" (models/blocks.py:158-161, group_attn.py:252-255) in front
 * of the FFN's pre_norm (blocks.py:385-386), and one segment is every `norm` op of the decoder.
 * m_live as in simpb_gemm_f32. out may alias x0 when k1 == 0. */
int simpb_layernorm_f32(float* out, int ldo, const float* x0, int ld0, int k0, const float* x1, int ld1, int k1,
                        const float* gamma, const float* beta, int num_rows, const int* m_live, void* stream);

/* Forward direction of feature_maps_format (ops/__init__.py:63-92) in one pass. Level l is
 * [num_images = bs*cams, H_l, W_l, channels] in memory (what a channels_last backbone emits), f16
 * (src_is_half) or f32; level_ptrs/level_hw are HOST arrays of num_levels device pointers / H_l*W_l.
 * col_feats f32 [bs, cams * sum_l H_l*W_l, channels], camera-major then level then row-major, as
 * ops/src/deformable_aggregation.cpp:22-28 expects. channels % 8 == 0. level_bias (HOST array of num_levels device
 * pointers, or NULL; entries may be NULL): per-channel bias of the same dtype added on the way (the last FPN
 * convolution then runs without its bias: mmdet FPN.fpn_convs, config :90-99); f16 sums are rounded to f16 first,
 * so the tokens are the numbers the separate bias pass produced. */
#define SIMPB_MAX_LEVELS 8
int simpb_format_tokens(float* col_feats, const void* const* level_ptrs, const void* const* level_bias,
                        const int* level_hw, int num_levels, int num_images, int channels, int src_is_half, void* stream);

/* Backbone convolution epilogue after conv-BN folding (tools/fuse_conv_bn.py:10-48): in place,
 * y f16 [num_pixels, channels] (NHWC) = relu?(y + bias[c] + residual?). bias f16 [channels]; residual f16
 * like y or NULL; channels % 8 == 0; all pointers 16-byte aligned. */
int simpb_bias_act_nhwc_f16(void* y, const void* bias, const void* residual, long long num_pixels, int channels,
                            int relu, void* stream);

/* Stem epilogue of the BN-folded ResNet (mmdet ResNet.forward: conv1 -> bn1 -> relu -> maxpool(3, stride 2, padding 1)):
 *   y[n, oy, ox, c] = relu( max over the window of x[n, 2*oy-1+dy, 2*ox-1+dx, c] + bias[c] ),  dy, dx in 0..2 (inside the map)
 * x f16 [num_images, in_h, in_w, channels] = the raw output of the 7x7 convolution (NHWC), y f16 [num_images, ho, wo, channels],
 * ho = (in_h - 1) / 2 + 1. Bias and ReLU commute with the maximum, so this equals the bias/ReLU pass followed by the pooling
 * kernel bit for bit, in one pass. channels % 8 == 0; 16-byte aligned. */
int simpb_bias_relu_maxpool_nhwc_f16(void* y, const void* x, const void* bias, int num_images, int in_h, int in_w,
                                     int channels, void* stream);

/* A whole BN-folded 1x1 convolution of the fp16 channels_last backbone in one launch:
 *   y[n, ho, wo, :] = relu?( x[n, ho*stride, wo*stride, :] . weight^T + bias (+ residual[n, ho, wo, :]) )
 * x f16 [num_images, in_h, in_w, in_channels] (NHWC), weight f16 [out_channels, in_channels], bias f16 [out_channels],
 * residual f16 like y or NULL (with residual_upsample2x: f16 [num_images, ho / 2, wo / 2, out_channels], read with
 * nearest-neighbour 2x upsampling = the FPN top-down sum lateral[i-1] += interpolate(lateral[i]) of mmdet FPN.forward;
 * ho, wo even), y f16 [num_images, ho, wo, out_channels] with ho = (in_h - 1) / stride + 1; input_bias f16
 * [in_channels] or NULL: x is first replaced by relu(x + input_bias) (rounded to f16), i.e. the epilogue of the
 * BN-folded 3x3 convolution that produced x (bottleneck conv2 -> conv3) without a pass of its own; fp32
 * accumulate. stride 1 or 2; in_channels % 64 == 0, out_channels % 8 == 0; 16-byte aligned. These are conv1 / conv3 /
 * downsample of every ResNet bottleneck and the FPN lateral convolutions (mmdet ResNet + FPN of
 * projects/configs/simpb_nus_r50_img_704x256.py:79-99 after tools/fuse_conv_bn.py:10-48). variant 0 = choose the kernel
 * from the shape; 1 = the round-1 kernel (single LDS stage; the only one that takes input_bias); 2 / 3 = 128 x 64 / 128 x 128
 * tiles of the staged pipeline of simpb_conv3x3_nhwc_f16 (same sums up to fp32 summation order). */
int simpb_conv1x1_nhwc_f16(void* y, const void* x, const void* weight, const void* bias, const void* residual,
                           int num_images, int in_h, int in_w, int in_channels, int out_channels, int stride, int relu,
                           int residual_upsample2x, const void* input_bias, int variant, void* stream);

/* A whole BN-folded 3x3 convolution (padding 1) of the fp16 channels_last backbone in one launch, as an implicit GEMM:
 *   out[n, ho, wo, :] = relu?( sum_{dy,dx} x[n, ho*stride + dy - 1, wo*stride + dx - 1, :] . weight[:, dy, dx, :]^T + bias )
 * x f16 [num_images, in_h, in_w, in_channels] (NHWC), weight f16 [out_channels, 3, 3, in_channels] (what PyTorch keeps for a
 * channels_last Conv2d weight), bias f16 [out_channels]; fp32 accumulate, one rounding to f16. Exactly one destination:
 *   y      f16 [num_images, ho, wo, out_channels], ho = (in_h - 1) / stride + 1, or
 *   tokens f32: the decoder's token buffer col_feats [bs, cams * tokens_per_cam, out_channels] of feature_maps_format
 *          (projects/mmdet3d_plugin/ops/__init__.py:63-92); image n is camera block n, this level's pixels start at row
 *          level_start of the block; the value written is the f16 result widened to f32 (what a separate format pass over
 *          the f16 map would write): the FPN's output convolutions produce the tokens themselves. tokens_f16 (f16, same
 *          layout, or NULL): the same rows without the widening, for simpb_linear_f16in_split (value_proj).
 * These are conv2 of every ResNet bottleneck and FPN.fpn_convs (mmdet ResNet + FPN of
 * projects/configs/simpb_nus_r50_img_704x256.py:79-99 after tools/fuse_conv_bn.py:10-48). stride 1 or 2;
 * in_channels % 64 == 0, out_channels % 8 == 0; 16-byte aligned. variant 0 = choose the tiling from the shape; 1-8 force one
 * (pixels x channels per workgroup: 1 = 128 x 64 and 2 = 256 x 64 with the activations read straight into the MFMA layout;
 * 3 = 32 x 64 and 4 = 64 x 64 with K split over the workgroup's waves; 5 = 128 x 64, 6 = 128 x 128, 7 = 96 x 64 and
 * 8 = 96 x 128 with both operands staged through LDS) -- all give the same sums up to fp32 summation order. */
int simpb_conv3x3_nhwc_f16(void* y, float* tokens, void* tokens_f16, int tokens_per_cam, int level_start, const void* x,
                           const void* weight, const void* bias, int num_images, int in_h, int in_w, int in_channels,
                           int out_channels, int stride, int relu, int variant, void* stream);

/* The same 3x3 / stride 1 convolution for up to four inputs of one channel count in ONE launch, each result written as its
 * level's token rows (f32 `tokens`, and f16 `tokens_f16` or NULL) exactly as simpb_conv3x3_nhwc_f16 does with `tokens`:
 * the four `fpn_convs[i].conv` of mmdet's FPN (projects/configs/simpb_nus_r50_img_704x256.py:92-99) behind
 * feature_maps_format (ops/__init__.py:63-92). x[j] f16 NHWC [num_images, in_h[j], in_w[j], Cin]; weight[j] f16
 * [Cout, 3, 3, Cin]; bias[j] f16 [Cout]; level_start[j] = the level's first row inside a camera's tokens_per_cam rows.
 * 96 x 128 staged tiles for every level (tiling 8 above): the small levels' tiles fill the last round of the large one. */
int simpb_conv3x3_group_tokens_f16(int num_levels, float* tokens, void* tokens_f16, int tokens_per_cam, const int* level_start,
                                   const void* const* x, const void* const* weight, const void* const* bias, int num_images,
                                   const int* in_h, const int* in_w, int in_channels, int out_channels, int relu, void* stream);

/* Attention core of torch.nn.MultiheadAttention (between in_proj and out_proj), exact fp32, flash
 * style, head_dim = 64: out[b,q,h*64+d] = sum_k softmax_k(scale * Q[b,q,h,:].K[b,k,h,:]) V[b,k,h,d].
 * q/k/v/out are [batch, N, heads*64] with row strides ldq/ldk/ldv/ldo (floats; batches are N*ld apart),
 * so slices of a fused projection buffer can be passed as they are. q and k 16-byte aligned.
 * Camera-grouped form (both tables non-NULL, num_key == num_query): slot q only attends slots of its
 * own group: query_cam i32 [N] (-1 = slot outside every group -> zeros), group_start i32 [cams+1];
 * this equals the reference's additive -inf block mask + nan_to_num (models/group_attn.py:104-131). */
int simpb_attention_f32(float* out, const float* q, const float* k, const float* v, const int* query_cam,
                        const int* group_start, int batch_size, int num_heads, int head_dim, int num_query,
                        int num_key, int ldq, int ldk, int ldv, int ldo, float scale, void* stream);

/* The same operator (same arguments, same layouts, fp32 in and out) on the FP16 matrix cores with split operands:
 * every fp32 operand is carried as two halfs (x = xh + xl / 2^11, 22 bits), partial products are exact in fp32
 * accumulators (csrc/attention.hip attention_f16s_kernel; fp32-grade: tests/test_gpu_ops.py holds it to the exact
 * kernel's bound against float64). What a frame runs since round 4 (plugin/routes.py attention_split_fp16); operands
 * must be inside the half-precision range (|x| < 65504: the decoder's projected queries / keys / values are O(10)). */
int simpb_attention_f32_split(float* out, const float* q, const float* k, const float* v, const int* query_cam,
                              const int* group_start, int batch_size, int num_heads, int head_dim, int num_query,
                              int num_key, int ldq, int ldk, int ldv, int ldo, float scale, void* stream);

/* ... and on operands the producing GEMM already left split (simpb_gemm_job.out_fmt = SIMPB_GEMM_OUT_SPLIT_HALFS): every
 * element of q / k / v is a 32-bit word holding half(x) in its low and half((x - half(x)) * 2^11) in its high 16 bits, at
 * the position (same strides ldq / ldk / ldv, counted in 32-bit words) the fp32 element would have. The softmax scale is
 * NOT applied here: the producer folds it into the query rows (1 / sqrt(64) is a power of two, so that is exact). */
int simpb_attention_split_halfs(float* out, const void* q, const void* k, const void* v, const int* query_cam,
                                const int* group_start, int batch_size, int num_heads, int head_dim, int num_query,
                                int num_key, int ldq, int ldk, int ldv, int ldo, void* stream);

/* Fused small-MLP chains: a whole `linear_relu_ln` stack (models/blocks.py:32-43) -- [Linear, ReLU]*,
 * LayerNorm, ..., optional last Linear and Scale -- in one launch; up to 8 independent chains over
 * the same rows share it (e.g. the pos/size/yaw/vel branches of SparseBox3DEncoder,
 * models/detection3d/blocks.py:57-74). The argument block is a plain host struct of device
 * pointers and sizes. Every width <= 256.
 *   op LINEAR:    w = weight, f32 [out_dim, in_dim] (or transposed, see weights_transposed); b = bias [out_dim] or NULL; relu 0/1
 *   op LAYERNORM: w = gamma, b = beta, f32 [in_dim]; eps 1e-5
 *   chain input:  IN_ROWS    x f32 rows of in_dim values, row stride ldx (floats); optional second
 *                            addend x2 (row stride ldx2): input = x + x2
 *                 IN_SINE2D  x f32 rows holding (x, y) in [0,1] at columns 0,1; the 256-d sine embedding
 *                            of models/utils.py:40-63 (cat(pos_y, pos_x)) is computed in the kernel
 *                 IN_ROWS_LN input = LayerNorm(x) * ln_w + ln_b (eps 1e-5) + x2: the decoder's `norm` operator in front of
 *                            a refinement head (models/simpb_head.py operation_order "norm", "refine*") inside the head's
 *                            launch; ln_out (row stride ld_ln_out, may be NULL) receives LayerNorm(x), the operator's
 *                            own output (rows >= *m_live as zeros). 4-row / 32-row kernels (weights_transposed == 2, 3) only.
 *   chain output: out rows of the last width, row stride ldo; out_scale (or NULL) multiplies column-wise
 *                 (mmcv Scale after the last Linear); then the optional `post` stage on v = out[row, t]:
 *     POST_REFINE3D  SparseBox3DRefinementModule.forward (models/detection3d/blocks.py:133-143):
 *                    v = (t >= div_col0 ? v / div[row / div_rows] : v) + res[row, t]   (res = anchor, ldres)
 *     POST_REFINE2D  SparseBox2DRefinementModule.forward (models/detection2d/blocks.py:122-125, 144):
 *                    v = sigmoid(v + (t < res_cols ? inverse_sigmoid(res[row, t]) : 0))   (res = anchor2d;
 *                    inverse_sigmoid of models/utils.py:4-8, eps 1e-5)
 *     POST_SIGMOID   v = sigmoid(v)   (ReWeight.alpha, models/aggregation.py:23-24) */
#define SIMPB_MLP_MAX_OPS 12
#define SIMPB_MLP_MAX_CHAINS 8
#define SIMPB_MLP_POST_NONE 0
#define SIMPB_MLP_POST_REFINE3D 1
#define SIMPB_MLP_POST_REFINE2D 2
#define SIMPB_MLP_POST_SIGMOID 3
#define SIMPB_MLP_LINEAR 0
#define SIMPB_MLP_LAYERNORM 1
#define SIMPB_MLP_IN_ROWS 0
#define SIMPB_MLP_IN_SINE2D 1
#define SIMPB_MLP_IN_ROWS_LN 2
typedef struct simpb_mlp_op {
  int type, in_dim, out_dim, relu;
  const float* w;
  const float* b;
} simpb_mlp_op;
typedef struct simpb_mlp_chain {
  const float* x;
  const float* x2;
  float* out;
  const float* out_scale;
  int ldx, ldx2, ldo, in_dim, in_mode, n_ops;
  int post, ldres, res_cols, div_rows, div_col0, reserved;
  const float* res;
  const float* div;
  simpb_mlp_op ops[SIMPB_MLP_MAX_OPS];
  const float* ln_w;   /* IN_ROWS_LN: gamma, beta f32 [in_dim] */
  const float* ln_b;
  float* ln_out;
  int ld_ln_out, reserved2;
} simpb_mlp_chain;
typedef struct simpb_mlp_args {
  int num_rows, num_chains;
  int weights_transposed;  /* 0: LINEAR.w is nn.Linear's [out_dim, in_dim] (16-row MFMA kernel); 1: [in_dim, out_dim] (VALU
                            * kernel); 2: k4-packed [in_dim / 4][out_dim][4] for layers with in_dim % 4 == 0, nn.Linear's
                            * layout for the others (4-row kernel on the 4x4 matrix blocks: 225 workgroups for 900 rows);
                            * 3: fragment-packed [ceil(out_dim / 32)][in_dim / 32][4][64][4] for layers with in_dim % 32 == 0
                            * -- element [t][c][q][32 h + r][e] = W[32 t + r][32 c + 16 h + 4 q + e], zeros past out_dim --
                            * nn.Linear's layout for the others (32-row kernel on the 32x32 matrix tiles: launches with
                            * thousands of rows, i.e. a batch of camera streams) */
  int reserved;
  simpb_mlp_chain chain[SIMPB_MLP_MAX_CHAINS];
  const int* m_live;       /* device int or NULL: rows >= *m_live are capacity slots of the static 2D query set; workgroups
                            * whose rows are all past it write zeros and leave (as in simpb_gemm_f32) */
} simpb_mlp_args;
int simpb_mlp_chain_forward(const simpb_mlp_args* args, void* stream);

/* InstanceBank (models/instance_bank.py) state updates for the 11-d box state. All tensors f32 unless noted;
 * A = instances per stream (<= 1024), T = cached instances, C = embed_dims (% 4 == 0).
 * simpb_bank_get (:83-113): anchor_out [bs, T, 11] = cached_anchor warped by T_temp2cur [bs, 4, 4] and advanced by
 *   its velocity over time_interval [bs] (anchor_projection with time_intervals = [-time_interval], :98-101);
 *   mask_out u8 [bs] = |time_interval| <= max_time_interval (:87); time_interval_out [bs] = time_interval where it is
 *   non-zero and valid, else default_time_interval (:108-113).
 * simpb_bank_update (:121-150): rows [0, T) of the outputs = cached rows, rows [T, A) = the A - T current instances
 *   with the largest max-class logit (cls [bs, A, num_classes]), for streams with mask != 0; other streams keep their
 *   current rows and get their instance_id (i64 [bs, A], may be NULL) reset to -1. index_scratch i32 [bs, A - T].
 * simpb_bank_cache (:152-196): cache() + get_instance_id() + update_instance_id(): confidence [bs, T] (in: last
 *   frame's, used when has_previous; out: this frame's), cached_feature [bs, T, C] / cached_anchor [bs, T, 11] out,
 *   instance_id i64 [bs, A] in/out (may be NULL), prev_id i64 [1] in/out, ids_out i64 [bs, A] = the ids
 *   get_instance_id returns; fresh ids are numbered over the flattened batch (:179-181); threshold applies to
 *   sigmoid(max class logit) when has_threshold. index_scratch i32 [bs, T]. hold i32 [num_hold] (may be NULL with
 *   num_hold = 0): overflow flags of this frame's 2D query sets (simpb_alloc_group_start); when any is non-zero the
 *   call writes NOTHING, so that the caller can re-run the frame with a larger capacity on the state the frame found. */
int simpb_bank_get(float* anchor_out, unsigned char* mask_out, float* time_interval_out, const float* cached_anchor,
                   const float* T_temp2cur, const float* time_interval, int batch_size, int num_temp,
                   float max_time_interval, float default_time_interval, void* stream);
int simpb_bank_update(float* feature_out, float* anchor_out, long long* instance_id, int* index_scratch,
                      const float* feature, const float* anchor, const float* cls, const float* cached_feature,
                      const float* cached_anchor, const unsigned char* mask, int batch_size, int num_anchors,
                      int num_classes, int num_temp, int embed_dims, void* stream);
/* simpb_bank_update in its two steps, for callers that can rank early: the ranking depends on the first decoder layer's
 * classification only, not on the bank, so a caller overlapping frames computes it beside the previous frame's decoder
 * (runner.SplitPipelinedRunner) and only the merge waits for the bank. The merge can also carry the anchor embeddings
 * (embed_out / embed [bs, A, E] / cached_embed [bs, T, E], all NULL = absent): embedding rows follow their anchor rows, which
 * replaces the encoder launch behind the update (models/simpb_head.py:621-622). hold (i32 [num_hold], may be NULL): when
 * any is set the instance_id reset of masked-out streams is skipped (the frame is going to be re-run). sticky (device i32,
 * may be NULL) chains the hold over frames decoded before the host has seen their predecessor's flags: simpb_bank_cache
 * writes "held back" (0 / 1) into it at the end of a frame, and a set word holds the NEXT frame's update and commit back
 * like one of its own flags. */
int simpb_bank_update_rank(int* index_scratch, const float* cls, int batch_size, int num_anchors, int num_classes,
                           int num_temp, void* stream);
int simpb_bank_update_merge(float* feature_out, float* anchor_out, float* embed_out, long long* instance_id, const int* index,
                            const float* feature, const float* anchor, const float* embed, const float* cached_feature,
                            const float* cached_anchor, const float* cached_embed, const unsigned char* mask, const int* hold,
                            int num_hold, const int* sticky, int batch_size, int num_anchors, int num_temp, int embed_dims,
                            int pos_embed_dims, void* stream);
int simpb_bank_cache(float* confidence, float* cached_feature, float* cached_anchor, long long* instance_id,
                     long long* prev_id, long long* ids_out, int* index_scratch, const float* feature, const float* anchor,
                     const float* cls, int batch_size, int num_anchors, int num_classes, int num_temp, int embed_dims,
                     int has_previous, float confidence_decay, int has_threshold, float threshold, const int* hold,
                     int num_hold, int* sticky, void* stream);
/* simpb_bank_cache with one workgroup per stream (a batch of streams: the serial form walks them, 13 us each). Same
 * arguments and results; sync_words: two persistent u32 words in device memory, zero before the first call and owned by
 * these calls from then on (an arrival counter and a launch count: the workgroups meet once between counting the fresh
 * instances of the streams in front of them and rewriting their own instance_id). batch_size <= 64; NULL sync_words or a
 * batch of one: the serial kernel. */
int simpb_bank_cache_streams(float* confidence, float* cached_feature, float* cached_anchor, long long* instance_id,
                             long long* prev_id, long long* ids_out, int* index_scratch, const float* feature, const float* anchor,
                             const float* cls, int batch_size, int num_anchors, int num_classes, int num_temp, int embed_dims,
                             int has_previous, float confidence_decay, int has_threshold, float threshold, const int* hold,
                             int num_hold, int* sticky, unsigned* sync_words, void* stream);

/* Fixed-shape detection records of SparseBox3DDecoder.decode_with2d (models/detection3d/decoder.py:124-252).
 * 3D (:133-167 with squeezed classes + decode_box :23-34), one workgroup per sample:
 *   score = max_c sigmoid(cls); the num_output best anchors; re-scored by sigmoid(quality[..., 0]) (quality
 *   may be NULL) and sorted again; rec3d f32 [bs, num_output, SIMPB_RECORD3D_WIDTH = 15] = x y z exp(w) exp(l) exp(h)
 *   atan2(sin, cos) vx vy vz | score | label | score before the re-score | the int64 instance id (or -1) as two
 *   32-bit lanes, low dword first, bit-cast into columns 13 and 14 (read them back through an int64 view: the
 *   reference keeps ids int64 end to end, decoder.py:247-251, and a float is exact only up to 2^24); rank_of_anchor i32 [bs, A] =
 *   rank of the anchor in that order, or -1. cls f32 [bs, A, C]; quality f32 [bs, A, 2]; box f32 [bs, A, 11];
 *   instance_id i64 [bs, A] or NULL. A <= 1024, num_output <= 512.
 * 2D (:168-175 + decode_box2d :36-51), one thread per slot: rec2d f32 [bs, N2, 8] = xyxy box in original image
 *   pixels | max_c sigmoid(cls2d) | label | rank of the slot's anchor (rank_of_anchor[q2a], or -1) | camera. */
int simpb_decode3d_record(float* rec3d, int* rank_of_anchor, const float* cls, const float* quality, const float* box,
                          const long long* instance_id, int batch_size, int num_anchors, int num_classes,
                          int num_output, void* stream);
int simpb_decode2d_record(float* rec2d, const float* cls2d, const float* box2d, const int* q2a, const int* query_cam,
                          const int* rank_of_anchor, int batch_size, int num_query2d, int num_classes, int num_anchors,
                          float crop_w, float crop_h, float crop_y0, float resize, void* stream);

/* The 2D record for a batch of INDEPENDENT streams (simpb_alloc_ragged below): cls2d [slots, C], box2d [slots, 4], q2a and
 * query_cam [slots] are the flat slot array, group_start i32 [bs * cams + 1] its group table, rank_of_anchor i32 [bs * A]
 * (simpb_decode3d_record's output, indexed by the flat anchor b * A + a that q2a holds). rec2d f32 [bs, rows_per_stream, 8]:
 * stream b's slots in its leading rows exactly as simpb_decode2d_record writes them for a batch of one (camera counted
 * within the stream), pad rows (rank -1, camera -1) behind them. (decoder.py:168-175 per sample) */
int simpb_decode2d_record_ragged(float* rec2d, const float* cls2d, const float* box2d, const int* q2a, const int* query_cam,
                                 const int* group_start, const int* rank_of_anchor, int batch_size, int rows_per_stream,
                                 int num_cams, int num_classes, int num_anchors, float crop_w, float crop_h, float crop_y0,
                                 float resize, void* stream);

/* Exchange form of a 2D record (simpb_amd/dist.py): out f32 [bs, rows_out, 8] = the rows of rec2d [bs, rows_in, 8] that
 * decode_with2d returns (rank >= 0 and camera >= 0: slots of the kept 3D boxes, decoder.py:176-251) in their order, pad rows
 * (zeros, rank -1, camera -1) behind them. num_output x num_cams rows always suffice (one slot per anchor and camera), so
 * the exchange shape does not depend on a runner's slot capacity. Rows past rows_out would be dropped (cannot happen at
 * that size). out_stride / in_stride: floats between consecutive streams (multiples of 4; the output may be a column range
 * of a wider send buffer). Both buffers 16-byte aligned. */
int simpb_record2d_compact(float* out, long long out_stride, const float* rec2d, long long in_stride, int batch_size, int rows_in,
                           int rows_out, void* stream);

/* Top-k of each score row, sorted descending (ties: lower index first): values f32 [bs, k], indices
 * i32 [bs, k] from scores f32 [bs, n], n <= 2048, k <= n. What `topk` of models/instance_bank.py:13-20
 * and the ranking of SparseBox3DDecoder.decode (models/detection3d/decoder.py:145-167) ask of torch.topk /
 * torch.sort; one launch, one workgroup per row. */
int simpb_topk_rows(float* values, int* indices, const float* scores, int batch_size, int n, int k, void* stream);

/* out[row] = sigmoid(dot(x[row, 0:k], w) + b[0]) : ReWeight.alpha (models/aggregation.py:23-24:
 * Linear(f_dim, 1) + Sigmoid) over rows of stride ldx. Rows >= *m_live (may be NULL) get 0. k % 4 == 0. */
int simpb_rowdot_sigmoid(float* out, const float* x, int ldx, const float* w, const float* b, int num_rows, int k,
                         const int* m_live, void* stream);

/* SparseBox3DKeyPointsGenerator.anchor_projection (models/detection3d/blocks.py:248-280) for one
 * transform: out[b, a] = anchor[b, a] with centre = R (centre - vel * time_interval[b]) + t (time_interval
 * may be NULL), velocity = R vel, size kept, and the yaw pair handled exactly as written there (:271-278:
 * the 2x2 block applied to [cos, sin], stored into the [sin, cos] slots in that order). This is what
 * InstanceBank.get (models/instance_bank.py:98-101) runs on the 600 cached anchors at the start of every
 * frame. anchor/out f32 [bs, A, 11]; T_src2dst f32 [bs, 4, 4] row-major; time_interval f32 [bs]. */
int simpb_anchor_projection(float* out, const float* anchor, const float* T_src2dst, const float* time_interval,
                            int batch_size, int num_anchors, void* stream);

/* Operands of simpb_ms_deform_attn_grouped_forward from the raw projections (models/group_attn.py:181-201):
 * raw rows (stride ldraw) = [sampling_offsets heads*L*P*2 | attention logits heads*L*P];
 * attn_weight = softmax over the L*P logits of each head; sampling_loc = ref[row, 0:2] + offset / (W_l, H_l).
 * ref rows have stride ldref (the 2-d reference points, simpb_head.py:523); spatial_shapes i64 [L, 2] = (H, W).
 * Rows >= *m_live (may be NULL) get zeros. Outputs in the layouts of simpb_ms_deform_attn_grouped_forward. */
int simpb_msda_prep(float* sampling_loc, float* attn_weight, const float* raw, int ldraw, const float* ref, int ldref,
                    const long long* spatial_shapes, int num_rows, int num_heads, int num_levels, int num_points,
                    const int* m_live, void* stream);

/* Sampling locations of the 3D deformable aggregation in its own layout: key points of
 * SparseBox3DKeyPointsGenerator.forward (models/detection3d/blocks.py:181-222: num_fix fixed scales x
 * exp(wlh) plus num_learn (sigmoid(learnable) - 0.5) x exp(wlh), rotated by yaw, + centre) projected by
 * DeformableFeatureAggregation.project_points (models/blocks.py:198-213: / clamp(depth, 1e-5), / image_wh).
 *   anchor f32 [bs, A, 11]; learnable f32 [bs, A, num_learn, 3] (output of learnable_fc);
 *   fix_scale f32 [num_fix, 3]; projection_mat f32 [bs, cams, 4, 4]; image_wh f32 [bs, cams, 2]
 *   loc f32 [bs, A, num_fix + num_learn, cams, 2]; key_points f32 [bs, A, P, 3] or NULL */
int simpb_dfa_points(float* loc, float* key_points, const float* anchor, const float* learnable,
                     const float* fix_scale, const float* projection_mat, const float* image_wh, int batch_size,
                     int num_anchors, int num_fix, int num_learn, int num_cams, void* stream);

/* Weights of the 3D deformable aggregation in its own layout (models/blocks.py:164-187 + :132-143):
 *   feat_logits f32 [bs, A, lvl*pts*groups] = weights_fc(feature + anchor_embed)
 *   cam_logits  f32 [bs, cams, lvl*pts*groups] = camera_embed @ weights_fc.weight^T (no bias)
 *   weights     f32 [bs, A, pts, cams, lvl, groups] = softmax over (cam, lvl, pts) per group of their sum */
int simpb_dfa_weights(float* weights, const float* feat_logits, const float* cam_logits, int batch_size,
                      int num_anchors, int num_cams, int num_levels, int num_pts, int num_groups, void* stream);

/* Adaptive query allocation, replaces DynamicQueryAllocation.projection_allocation
 * (models/allocation.py:27-144) in three steps; the caller reads `count` back between steps 2
 * and 3 to size the 2D query set (the reference does the same with .tolist() at :94).
 *
 * Step 1 (:30-83): project the 8 box corners (size = exp(wlh) clamped to limit_*) and the centre of
 * every anchor into every camera.
 *   anchor         f32 [batch_size, num_anchors, 11]   x,y,z,log w,log l,log h,sin,cos,vx,vy,vz
 *   projection_mat f32 [batch_size, num_cams, 4, 4]
 *   flag           u8  [batch_size, num_cams, num_anchors]   0 none, 1 corner-only, 2 centre valid
 *   sel_xy         f32 [batch_size, num_cams, num_anchors, 2] reference point in pixels
 *   depth          f32 [batch_size, num_cams, num_anchors]    centre depth (signed) */
int simpb_alloc_project(unsigned char* flag, float* sel_xy, float* depth, const float* anchor,
                        const float* projection_mat, int batch_size, int num_anchors, int num_cams,
                        float img_w, float img_h, float limit_w, float limit_l, float limit_h, void* stream);

/* Step 2 (:86-123): per (batch, cam) stable compaction in ascending anchor order.
 *   count i32 [batch_size, num_cams]; order i32 [batch_size, num_cams, num_anchors] (first count valid) */
int simpb_alloc_compact(int* count, int* order, const unsigned char* flag, int batch_size, int num_anchors,
                        int num_cams, void* stream);

/* Step 2b, optional (:91-99 without the host round trip): group_start i32 [num_cams + 1] =
 * prefix sums of the max-over-batch counts, clipped to `capacity`; overflow i32 [1] = 1 when the 2D
 * query set does not fit `capacity` (the caller must then redo the frame with a larger capacity). */
int simpb_alloc_group_start(int* group_start, int* overflow, const int* count, int batch_size, int num_cams,
                            int capacity, void* stream);

/* Step 3 (:103-142): fill the slot tables. group_start i32 [num_cams + 1] (device) are the
 * max-over-batch prefix sums (:91-99); slots past a sample's own count are pads (q2a = -1, zeros).
 *   ref_pts2d f32 [bs, num_query, 2] (divided by img_w, img_h); ref_depth2d f32 [bs, num_query, 1] = |depth|
 *   q2a i32 [bs, num_query] slot -> anchor; is_center i32 [bs, num_query];
 *   a2q i32 [bs, num_anchors, num_cams] (anchor, cam) -> slot or -1; query_cam i32 [num_query]
 *   (num_query may exceed group_start[num_cams]: those capacity slots get query_cam = -1 and pad values)
 * q2a/is_center are the index form of the reference's one-hot trans_matrix / center_matrix. */
int simpb_alloc_scatter(float* ref_pts2d, float* ref_depth2d, int* q2a, int* is_center, int* a2q, int* query_cam,
                        const int* group_start, const int* count, const int* order, const unsigned char* flag,
                        const float* sel_xy, const float* depth, int batch_size, int num_anchors, int num_cams,
                        int num_query, float img_w, float img_h, void* stream);

/* Steps 1-3 with a fixed capacity as three launches instead of five (a replayed frame pays ~4.7 us of dispatch per
 * kernel whatever it does, and these run three times per frame): the -1 fill of a2q rides in step 1 (same index space), the
 * group table of simpb_alloc_group_start (capacity = num_query) is derived by step 3's threads from the counts and
 * published by one of them. Every output of the four stepwise calls, same arithmetic, same tables; num_cams <= 8. (One workgroup walking
 * all steps was measured at 49 us against 24 us for the five launches: its steps hand over through global memory and
 * serialise the round trips.) */
int simpb_alloc_static(unsigned char* flag, float* sel_xy, float* depth, int* count, int* order, int* group_start,
                       int* overflow, float* ref_pts2d, float* ref_depth2d, int* q2a, int* is_center, int* a2q,
                       int* query_cam, const float* anchor, const float* projection_mat, int batch_size, int num_anchors,
                       int num_cams, int capacity, float img_w, float img_h, float limit_w, float limit_l, float limit_h,
                       void* stream);

/* The allocation for a batch of INDEPENDENT camera streams ("keep per-sample counts in the native path"):
 * DynamicQueryAllocation.projection_allocation (models/allocation.py:27-144) pads every camera group to the max over the
 * batch (:91-99), so a batch of streams with different headings carries ~3x the 2D slots a stream needs and attends the
 * pads as keys; here every stream keeps the set a batch of one gives it. Steps 1-2 as simpb_alloc_static; step 3 lays the
 * 2D set out as ONE flat slot array [batch_size * per_stream], stream-major then camera-major, live slots first:
 *   group_start i32 [bs * cams + 1] (group g = b * cams + cam), query_cam i32 [slots] = g (or -1: capacity slot),
 *   q2a i32 [slots] = b * num_anchors + a (or -1), is_center, ref_pts2d f32 [slots, 2], ref_depth2d f32 [slots],
 *   a2q i32 [bs, A, cams] = flat slot (or -1); overflow[0] = 1 when a stream needs more than per_stream slots (clipped).
 * The 2D operators then run as a batch of one over bs * cams camera groups (they take the group tables as they are),
 * and the 3D side sees [1, bs * A, .] views. batch_size * num_cams <= 96. */
int simpb_alloc_ragged(unsigned char* flag, float* sel_xy, float* depth, int* count, int* order, int* group_start,
                       int* overflow, float* ref_pts2d, float* ref_depth2d, int* q2a, int* is_center, int* a2q,
                       int* query_cam, const float* anchor, const float* projection_mat, int batch_size, int num_anchors,
                       int num_cams, int per_stream, float img_w, float img_h, float limit_w, float limit_l, float limit_h,
                       void* stream);

/* out[b, s, :] = src[b, q2a[b, s], :], zeros where q2a < 0: replaces
 * torch.matmul(ref_trans_matrix, instance_feature) (models/simpb_head.py:438). channels % 4 == 0. */
int simpb_gather_rows(float* out, const float* src, const int* q2a, int batch_size, int num_anchors,
                      int num_query, int channels, void* stream);

/* 2D -> 3D re-weighted mean, replaces ReWeight.forward's two dense matmuls
 * (models/aggregation.py:30-35) plus the adds at :88-89:
 *   out_q[b,a]   = q3d[b,a]   + sum_s alpha[b,s] q2d[b,s]   / clamp(sum_s alpha[b,s], 1e-5)
 *   out_pos[b,a] = pos3d[b,a] + sum_s alpha[b,s] pos2d[b,s] / clamp(...)      s over a2q[b,a,:] >= 0 */
int simpb_aggregate_2d_to_3d(float* out_q, float* out_pos, const float* q3d, const float* pos3d, const float* q2d,
                             const float* pos2d, const float* alpha, const int* a2q, int batch_size,
                             int num_anchors, int num_cams, int num_query, int channels, void* stream);

/* The same with ReWeight.alpha (models/aggregation.py:23-24) computed inside the launch instead of read from memory:
 * alpha[b,s] = sigmoid(hidden[b,s,:] . w_alpha + b_alpha) with hidden f32 [bs, num_query, hidden_dim] (row stride
 * ld_hidden) = ReLU(ReWeight.reduce(...)); every slot belongs to one anchor, so each alpha is computed once. hidden NULL:
 * alpha is read as above. num_cams <= 8; hidden_dim % 4 == 0; hidden, w_alpha 16-byte aligned. */
int simpb_aggregate_2d_to_3d_alpha(float* out_q, float* out_pos, const float* q3d, const float* pos3d, const float* q2d,
                                   const float* pos2d, const float* alpha, const int* a2q, const float* hidden,
                                   int ld_hidden, int hidden_dim, const float* w_alpha, const float* b_alpha,
                                   int batch_size, int num_anchors, int num_cams, int num_query, int channels,
                                   void* stream);

#ifdef __cplusplus
}
#endif
#endif
