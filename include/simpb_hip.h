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

/* Library/ABI version (bumped when a signature changes). */
int simpb_abi_version(void);

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

#ifdef __cplusplus
}
#endif
#endif
