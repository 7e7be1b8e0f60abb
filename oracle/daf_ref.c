/* ORACLE — test infrastructure only (never linked into the product).
 * Plain-C restatement of the reference's only native kernel, the forward of
 * deformable_aggregation (/root/reference/projects/mmdet3d_plugin/ops/src/deformable_aggregation_cuda.cu):
 *   bilinear_sampling            :13-59
 *   deformable_aggregation_kernel :129-187   (one CUDA thread per (b, a, p, cam, lvl, c); the
 *                                             atomicAdd at :183-186 becomes a sequential +=)
 * Index order, the (0,1) gate, `loc*size - 0.5` with a double literal, and the per-tap zero
 * padding follow the source line by line. Pinned by tests/test_oracle_golden.py against the
 * reference's own PyTorch fallback (interior points) and against the PyTorch restatement.
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libdaf_ref.so oracle/daf_ref.c -lm   (__graft_entry__.build) */
#include <math.h>

static float bilinear_sampling(const float* bottom_data, int height, int width, int num_embeds, float h_im,
                               float w_im, int base_ptr) {
  const int h_low = (int)floorf(h_im);
  const int w_low = (int)floorf(w_im);
  const int h_high = h_low + 1;
  const int w_high = w_low + 1;
  const float lh = h_im - h_low;
  const float lw = w_im - w_low;
  const float hh = 1 - lh, hw = 1 - lw;
  const int w_stride = num_embeds;
  const int h_stride = width * w_stride;
  const int h_low_ptr_offset = h_low * h_stride;
  const int h_high_ptr_offset = h_low_ptr_offset + h_stride;
  const int w_low_ptr_offset = w_low * w_stride;
  const int w_high_ptr_offset = w_low_ptr_offset + w_stride;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = bottom_data[h_low_ptr_offset + w_low_ptr_offset + base_ptr];
  if (h_low >= 0 && w_high <= width - 1) v2 = bottom_data[h_low_ptr_offset + w_high_ptr_offset + base_ptr];
  if (h_high <= height - 1 && w_low >= 0) v3 = bottom_data[h_high_ptr_offset + w_low_ptr_offset + base_ptr];
  if (h_high <= height - 1 && w_high <= width - 1) v4 = bottom_data[h_high_ptr_offset + w_high_ptr_offset + base_ptr];
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* output must be zero-initialised by the caller (at::zeros at deformable_aggregation.cpp:55). */
void daf_ref_forward(float* output, const float* mc_ms_feat, const int* spatial_shape, const int* scale_start_index,
                     const float* sample_location, const float* weights, int batch_size, int num_cams, int num_feat,
                     int num_embeds, int num_scale, int num_anchors, int num_pts, int num_groups) {
  const long long num_kernels =
      (long long)batch_size * num_pts * num_embeds * num_anchors * num_cams * num_scale;
  for (long long tid = 0; tid < num_kernels; ++tid) {
    long long idx = tid;
    const float weight = *(weights + idx / (num_embeds / num_groups));
    const int channel_index = (int)(idx % num_embeds);
    idx /= num_embeds;
    const int scale_index = (int)(idx % num_scale);
    idx /= num_scale;
    const int cam_index = (int)(idx % num_cams);
    idx /= num_cams;
    const int pts_index = (int)(idx % num_pts);
    idx /= num_pts;
    int anchor_index = (int)(idx % num_anchors);
    idx /= num_anchors;
    const int batch_index = (int)(idx % batch_size);
    anchor_index = batch_index * num_anchors + anchor_index;
    const int loc_offset = ((anchor_index * num_pts + pts_index) * num_cams + cam_index) << 1;
    const float loc_w = sample_location[loc_offset];
    if (loc_w <= 0 || loc_w >= 1) continue;
    const float loc_h = sample_location[loc_offset + 1];
    if (loc_h <= 0 || loc_h >= 1) continue;
    int cam_scale_index = cam_index * num_scale + scale_index;
    const int value_offset = (batch_index * num_feat + scale_start_index[cam_scale_index]) * num_embeds + channel_index;
    cam_scale_index = cam_scale_index << 1;
    const int h = spatial_shape[cam_scale_index];
    const int w = spatial_shape[cam_scale_index + 1];
    const float h_im = loc_h * h - 0.5;
    const float w_im = loc_w * w - 0.5;
    output[anchor_index * num_embeds + channel_index] +=
        bilinear_sampling(mc_ms_feat, h, w, num_embeds, h_im, w_im, value_offset) * weight;
  }
}
