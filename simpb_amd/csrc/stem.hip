// ResNet stem in one launch (gfx950): conv 7x7 / stride 2 / padding 3 (3 -> 64 channels, BN folded) + bias + ReLU +
// max_pool 3x3 / stride 2 / padding 1 -- mmdet ResNet.forward's `maxpool(relu(bn1(conv1(x))))`, the reference's
// backbone entry (projects/configs/simpb_nus_r50_img_704x256.py:79-99 after tools/fuse_conv_bn.py:10-48). It was the last
// vendor kernel of a frame (MIOpen picks its implementation at run time, and some of its solvers issue the double-K
// matrix instructions this code base must keep off the chip: DESIGN.md section 4); it also wrote the 34.6 MB
// pre-pooling map that only the pooling pass read back.
//
// Input: the image as f16 NHWC with FOUR channels per pixel (RGB + 0: `image_to_nhwc4` below, which replaces the cast /
// channels_last copies in front of the vendor kernel), so that a pixel is one aligned 8-byte operand. Implicit GEMM on
// v_mfma_f32_32x32x8_f16: M = conv pixels, N = 64, K = 7 rows x 4 pixel pairs x (2 pixels x 4 channels) = 224 (147 real
// taps: the 4th channel and the 8th tap of a row carry zero weights). A workgroup (4 waves) owns an 8 x 16 tile of the
// POOLED map = 18 x 34 conv pixels (halo included) = 20 blocks of 32 pixels, five per wave; the 41 x 74 input pixels
// under it are staged once in LDS (zero outside the image = the convolution's padding); weights (28 KB, packed by the
// host in fragment order) sit in LDS as well. Conv results are rounded to f16 (what the two-kernel route stored), then
// bias + ReLU + the 3x3 maximum in fp32, one rounding: bit-equal to conv -> bias_relu_maxpool.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);

namespace {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));

constexpr int kCout = 64;
constexpr int kTP_H = 8, kTP_W = 16;                 // pooled tile
constexpr int kTC_H = 2 * kTP_H + 2, kTC_W = 2 * kTP_W + 2;   // conv tile with halo: 18 x 34
constexpr int kConvPix = kTC_H * kTC_W;              // 612
constexpr int kBlocks = (kConvPix + 31) / 32;        // 20
constexpr int kIn_H = 2 * (kTC_H - 1) + 7, kIn_W = 2 * (kTC_W - 1) + 8;   // 41 x 74 input pixels (8th tap included)
constexpr int kSteps = 7 * 4;                        // k-steps of 8
constexpr int kThreads = 256;
constexpr int kConvLd = 32 + 8;                      // f16 row pitch of the staged conv tile: 32 channels per pass + bank spread

// fp32 NCHW (or any strides) image -> f16 [N, H, W, 4], channel 3 = 0
__global__ void image_to_nhwc4_kernel(h4* __restrict__ out, const float* __restrict__ img, long long sn, long long sc,
                                      long long sh, long long sw, int C, int H, int W, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const int y = (int)((i / W) % H);
  const long long n = i / ((long long)W * H);
  const float* p = img + n * sn + y * sh + x * sw;
  h4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
  v[0] = (_Float16)p[0];
  if (C > 1) v[1] = (_Float16)p[sc];
  if (C > 2) v[2] = (_Float16)p[2 * sc];
  out[i] = v;
}

__global__ __launch_bounds__(kThreads, 2) void stem_conv_pool_kernel(_Float16* __restrict__ out, const h4* __restrict__ img,
                                                                  const h4* __restrict__ wpack, const _Float16* __restrict__ bias,
                                                                  int H, int W, int Ho, int Wo, int Hp, int Wp) {
  // LDS: [input tile 41 x 74 x 8 B = 24 272 B | weights 28 x 2 x 64 x 8 B = 28 672 B], later reused for the conv tile, 32
  // channels at a time (612 x 40 halfs = 48 960 B): 53 KB per workgroup, three workgroups per CU
  constexpr int kInBytes = kIn_H * kIn_W * 8, kWBytes = kSteps * 2 * kCout * 8, kConvBytes = kConvPix * kConvLd * 2;
  constexpr int kLds = (kInBytes + kWBytes) > kConvBytes ? (kInBytes + kWBytes) : kConvBytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];
  h4* s_in = reinterpret_cast<h4*>(smem);
  h4* s_w = reinterpret_cast<h4*>(smem + kInBytes);
  _Float16* s_conv = reinterpret_cast<_Float16*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.z;
  const int py0 = blockIdx.y * kTP_H, px0 = blockIdx.x * kTP_W;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;      // conv pixel of tile position (0, 0)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;      // input pixel under it
  const h4* im = img + (size_t)n * H * W;

  // ---- stage the input pixels (zero outside the image) and the weights
  for (int i = tid; i < kIn_H * kIn_W; i += kThreads) {
    const int r = i / kIn_W, c = i - r * kIn_W;
    const int y = iy0 + r, x = ix0 + c;
    h4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    if (y >= 0 && y < H && x >= 0 && x < W) v = im[(size_t)y * W + x];
    s_in[i] = v;
  }
  for (int i = tid; i < kSteps * 2 * kCout; i += kThreads) s_w[i] = wpack[i];
  __syncthreads();

  // ---- implicit GEMM: this wave's five blocks of 32 conv pixels x 64 channels
  constexpr int kPer = kBlocks / 4;
  static_assert(kBlocks % 4 == 0, "blocks per wave");
  const int r32 = lane & 31, kb = lane >> 5;
  int a_off[kPer];   // LDS index of this lane's pixel (row r32 of the block) at tap (ky = 0, pair 0), + kb
#pragma unroll
  for (int b = 0; b < kPer; ++b) {
    const int m = min((wave * kPer + b) * 32 + r32, kConvPix - 1);   // (rows past the tile: repeated, never stored)
    const int r = m / kTC_W, c = m - r * kTC_W;
    a_off[b] = (2 * r) * kIn_W + 2 * c + kb;
  }
  f16acc acc[kPer][2];
#pragma unroll
  for (int b = 0; b < kPer; ++b)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[b][h][i] = 0.f;
#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int s = ky * 4 + g;
      const h4 b0 = s_w[(s * 2 + kb) * kCout + r32];
      const h4 b1 = s_w[(s * 2 + kb) * kCout + 32 + r32];
#pragma unroll
      for (int b = 0; b < kPer; ++b) {
        const h4 a = s_in[a_off[b] + ky * kIn_W + 2 * g];
        acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b0, acc[b][0], 0, 0, 0);
        acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b1, acc[b][1], 0, 0, 0);
      }
    }
  }
  __syncthreads();   // every wave is done with the input tile and the weights: the region becomes the conv tile

  // ---- per 32-channel half: conv tile to LDS, rounded to f16 like the stored map of the two-kernel route (C/D layout of the
  // 32x32 block: column (channel) = lane & 31, row (pixel) = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)), then bias + ReLU
  // + 3x3 / stride-2 maximum (csrc/bias_act.hip bias_relu_maxpool's arithmetic): thread -> (pooled pixel, 16 channels).
  // Conv pixels outside the conv map do not exist for the pooling (its padding): after the ReLU every existing value is
  // >= 0 and the centre of a window always exists, so they are skipped against a floor of 0.
  const int pp = tid >> 1, cq = (tid & 1) * 16;
  const int pr = pp / kTP_W, pc = pp - pr * kTP_W;
  const int py = py0 + pr, px = px0 + pc;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();   // the first half has been pooled
#pragma unroll
    for (int b = 0; b < kPer; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (wave * kPer + b) * 32 + (i & 3) + 8 * (i >> 2) + 4 * kb;
        if (m < kConvPix) s_conv[m * kConvLd + r32] = (_Float16)acc[b][h][i];
      }
    __syncthreads();
    if (py < Hp && px < Wp) {
      float best[16], bv[16];
#pragma unroll
      for (int c8 = 0; c8 < 2; ++c8) {
        const h8 t = *reinterpret_cast<const h8*>(bias + h * 32 + cq + 8 * c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { bv[8 * c8 + e] = (float)t[e]; best[8 * c8 + e] = 0.f; }
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int r = 2 * pr + dy, c = 2 * pc + dx;             // tile coordinates of conv pixel (2 py - 1 + dy, 2 px - 1 + dx)
          const int cy = cy0 + r, cx = cx0 + c;
          if (cy < 0 || cy >= Ho || cx < 0 || cx >= Wo) continue;
          const _Float16* src = s_conv + (r * kTC_W + c) * kConvLd + cq;
#pragma unroll
          for (int c8 = 0; c8 < 2; ++c8) {
            const h8 t = *reinterpret_cast<const h8*>(src + 8 * c8);
#pragma unroll
            for (int e = 0; e < 8; ++e) best[8 * c8 + e] = fmaxf(best[8 * c8 + e], (float)t[e] + bv[8 * c8 + e]);
          }
        }
      _Float16* o = out + (((size_t)n * Hp + py) * Wp + px) * kCout + h * 32 + cq;
#pragma unroll
      for (int c8 = 0; c8 < 2; ++c8) {
        h8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (_Float16)best[8 * c8 + e];
        *reinterpret_cast<h8*>(o + 8 * c8) = t;
      }
      simpb::stores_retired();   // store_fence.h: the second half's bias loads start with nothing of this one in flight
    }
  }
}

}  // namespace

extern "C" int simpb_image_to_nhwc4_f16(void* out, const float* img, long long stride_n, long long stride_c, long long stride_h,
                                        long long stride_w, int num_images, int channels, int height, int width, void* stream) {
  if (!out || !img || num_images <= 0 || channels <= 0 || channels > 3 || height <= 0 || width <= 0 ||
      (reinterpret_cast<size_t>(out) & 7))
    return SIMPB_EINVAL;
  const long long total = (long long)num_images * height * width;
  if (total > (1ll << 31) * 128) return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(image_to_nhwc4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<h4*>(out), img, stride_n, stride_c, stride_h, stride_w, channels, height, width, total);
  return simpb_check_launch();
}

extern "C" int simpb_stem_conv7x7_pool_f16(void* out, const void* img_nhwc4, const void* weight_packed, const void* bias,
                                           int num_images, int height, int width, int out_channels, void* stream) {
  if (!out || !img_nhwc4 || !weight_packed || !bias || num_images <= 0 || num_images > 65535 || height < 8 || width < 8 ||
      out_channels != kCout)
    return SIMPB_EINVAL;
  if ((reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(img_nhwc4) | reinterpret_cast<size_t>(weight_packed) |
       reinterpret_cast<size_t>(bias)) & 15)
    return SIMPB_EINVAL;
  const int ho = (height + 6 - 7) / 2 + 1, wo = (width + 6 - 7) / 2 + 1;   // conv 7x7, stride 2, padding 3
  const int hp = (ho + 2 - 3) / 2 + 1, wp = (wo + 2 - 3) / 2 + 1;         // max_pool 3x3, stride 2, padding 1
  if ((long long)num_images * height * width > (1ll << 31) - 1) return SIMPB_EINVAL;
  (void)hipGetLastError();
  dim3 grid((wp + kTP_W - 1) / kTP_W, (hp + kTP_H - 1) / kTP_H, num_images);
  hipLaunchKernelGGL(stem_conv_pool_kernel, grid, dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     static_cast<_Float16*>(out), static_cast<const h4*>(img_nhwc4), static_cast<const h4*>(weight_packed),
                     static_cast<const _Float16*>(bias), height, width, ho, wo, hp, wp);
  return simpb_check_launch();
}
