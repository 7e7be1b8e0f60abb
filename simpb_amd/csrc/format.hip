// feature_maps_format (gfx950): FPN outputs -> the channel-last token buffer of the decoder.
//
// Replaces the forward direction of feature_maps_format
// (/root/reference/projects/mmdet3d_plugin/ops/__init__.py:63-92: reshape + cat + permute + flatten,
// i.e. three full copies of the 92 MB feature set per frame on top of the fp16->fp32 casts) with ONE
// pass: the backbone runs channels_last, so a level is already [bs*cams, H, W, C] in memory and a
// token row is C contiguous values; each row is converted (fp16 -> fp32 if needed) and written once
// at its place in col_feats [bs, cam-major / level / row-major tokens, C].
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);

namespace {

struct LevelArgs {
  const void* src[SIMPB_MAX_LEVELS];
  const void* bias[SIMPB_MAX_LEVELS];  // per-channel bias of the convolution that produced the level (same dtype), or null
  int hw[SIMPB_MAX_LEVELS];       // H*W of each level
  int start[SIMPB_MAX_LEVELS];    // token offset of each level inside one camera's block
};

// thread = 8 channels of one token; grid.y = level
template <typename T>
__global__ void format_tokens_kernel(float* __restrict__ col, LevelArgs lv, int images, int C, int tokens_per_cam) {
  const int lvl = blockIdx.y;
  const int hw = lv.hw[lvl];
  const int c8 = C / 8;
  const long long n = (long long)images * hw * c8;
  const T* __restrict__ src = static_cast<const T*>(lv.src[lvl]);
  const T* __restrict__ bias = static_cast<const T*>(lv.bias[lvl]);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8;
    const long long tok = i / c8;            // image * hw + pixel
    const int img = (int)(tok / hw), pix = (int)(tok - (long long)img * hw);
    const T* s = src + tok * C + c;
    // every load of the row (and of the bias) first, then one full wait: the previous row's stores of this grid-stride
    // loop are retired with it and no counted wait releases consumers while anything is in flight (store_fence.h)
    float v[8], bv[8];
    const T* bp = bias ? bias + c : src;   // a valid address either way; the values are only used when bias != null
    if constexpr (sizeof(T) == 2) {
      const uint4 raw = *reinterpret_cast<const uint4*>(s);
      const uint4 braw = *reinterpret_cast<const uint4*>(bp);
      simpb::loads_retired();
      const __half2* h2 = reinterpret_cast<const __half2*>(&raw);
      const __half2* b2 = reinterpret_cast<const __half2*>(&braw);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float2 f = __half22float2(h2[j]), g = __half22float2(b2[j]);
        v[2 * j] = f.x; v[2 * j + 1] = f.y;
        bv[2 * j] = g.x; bv[2 * j + 1] = g.y;
      }
    } else {
      const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
      const float4 ba = *reinterpret_cast<const float4*>(bp), bb = *reinterpret_cast<const float4*>(bp + 4);
      simpb::loads_retired();
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      bv[0] = ba.x; bv[1] = ba.y; bv[2] = ba.z; bv[3] = ba.w; bv[4] = bb.x; bv[5] = bb.y; bv[6] = bb.z; bv[7] = bb.w;
    }
    if (bias) {  // the last FPN convolution's bias rides along (it ran without one)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (sizeof(T) == 2) v[j] = __half2float(__float2half_rn(v[j] + bv[j]));  // rounded to fp16 as the
        else v[j] += bv[j];                                                              // separate bias pass did
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) simpb::pin(v[j]);
    float* d = col + ((long long)img * tokens_per_cam + lv.start[lvl] + pix) * C + c;
    *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

}  // namespace

extern "C" int simpb_format_tokens(float* col_feats, const void* const* level_ptrs, const void* const* level_bias,
                                   const int* level_hw, int num_levels, int num_images, int channels, int src_is_half,
                                   void* stream) {
  if (!col_feats || !level_ptrs || !level_hw || num_levels <= 0 || num_levels > SIMPB_MAX_LEVELS || num_images <= 0 ||
      channels <= 0 || channels % 8 != 0)
    return SIMPB_EINVAL;
  LevelArgs lv;
  int total = 0, max_hw = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (!level_ptrs[l] || level_hw[l] <= 0) return SIMPB_EINVAL;
    lv.src[l] = level_ptrs[l];
    lv.bias[l] = level_bias ? level_bias[l] : nullptr;
    lv.hw[l] = level_hw[l];
    lv.start[l] = total;
    total += level_hw[l];
    max_hw = level_hw[l] > max_hw ? level_hw[l] : max_hw;
  }
  (void)hipGetLastError();
  const long long n = (long long)num_images * max_hw * (channels / 8);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  dim3 grid(blocks, num_levels);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (src_is_half)
    hipLaunchKernelGGL(format_tokens_kernel<__half>, grid, dim3(256), 0, s, col_feats, lv, num_images, channels, total);
  else
    hipLaunchKernelGGL(format_tokens_kernel<float>, grid, dim3(256), 0, s, col_feats, lv, num_images, channels, total);
  return simpb_check_launch();
}
