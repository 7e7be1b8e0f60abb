// Fixed-shape detection records of SparseBox3DDecoder.decode_with2d
// (/root/reference/projects/mmdet3d_plugin/models/detection3d/decoder.py:124-252) in two launches.
// In PyTorch the same work is ~40 launches of a few microseconds each at the end of every frame
// (sigmoid, max, top-k, gathers, re-score, sort, atan2/exp, cats, scatter, ...), all on the decoder's
// critical path because the next frame of the stream cannot start before this one has finished.
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);

namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ unsigned long long sort_key(float v, unsigned idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone map float -> uint
  return ((unsigned long long)u << 32) | (0xFFFFFFFFu - idx);
}
__device__ __forceinline__ unsigned key_index(unsigned long long k) { return 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull); }

// descending bitonic sort of CAP keys held in LDS by CAP/2 (or more) threads; `nthreads` threads call it
template <int CAP>
__device__ __forceinline__ void bitonic_desc(unsigned long long* key, int tid, int nthreads) {
  for (int size = 2; size <= CAP; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < CAP / 2; t += nthreads) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = key[lo], c = key[hi];
        if ((a < c) == desc) { key[lo] = c; key[hi] = a; }
      }
      __syncthreads();
    }
  }
}

constexpr int kCapA = 1024;  // anchors per row (900 shipped)
constexpr int kCapK = 512;   // kept boxes per row (300 shipped)

// decoder.py:133-167 (squeezed classes, quality re-score) + decode_box (:23-34), one workgroup per sample.
__global__ __launch_bounds__(512) void decode3d_kernel(float* __restrict__ rec3d, int* __restrict__ rank_of_anchor,
                                                       const float* __restrict__ cls, const float* __restrict__ quality,
                                                       const float* __restrict__ box, const long long* __restrict__ instance_id,
                                                       int A, int C, int K) {
  __shared__ unsigned long long key[kCapA];
  __shared__ unsigned long long key2[kCapK];
  __shared__ int label[kCapA];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int a = tid; a < kCapA; a += 512) {
    unsigned long long kv = 0ull;
    if (a < A) {
      const float* row = cls + ((size_t)b * A + a) * C;
      float m = row[0];
      int arg = 0;
      for (int c = 1; c < C; ++c)
        if (row[c] > m) { m = row[c]; arg = c; }
      label[a] = arg;
      kv = sort_key(sigmoidf(m), (unsigned)a);  // max of sigmoids = sigmoid of the max (monotone)
      rank_of_anchor[(size_t)b * A + a] = -1;
    }
    key[a] = kv;
  }
  __syncthreads();
  bitonic_desc<kCapA>(key, tid, 512);
  // re-score the K kept boxes by centerness and sort again (:154-167); ties keep the first ranking's order
  for (int r = tid; r < kCapK; r += 512) {
    unsigned long long kv = 0ull;
    if (r < K) {
      const unsigned a = key_index(key[r]);
      unsigned u = (unsigned)(key[r] >> 32);
      u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
      float s = __uint_as_float(u);
      if (quality) s *= sigmoidf(quality[((size_t)b * A + a) * 2]);
      kv = sort_key(s, (unsigned)r);
    }
    key2[r] = kv;
  }
  __syncthreads();
  bitonic_desc<kCapK>(key2, tid, 512);
  for (int r = tid; r < K; r += 512) {
    const unsigned first_rank = key_index(key2[r]);
    const unsigned a = key_index(key[first_rank]);
    unsigned u = (unsigned)(key2[r] >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    const float score = __uint_as_float(u);
    // the "score before the re-score" column is NOT re-ordered by the second sort in the reference
    // (decoder.py:157 clones it before :159-167 permute everything else): position r of the first ranking
    unsigned uo = (unsigned)(key[r] >> 32);
    uo = (uo & 0x80000000u) ? (uo & 0x7FFFFFFFu) : ~uo;
    const float* bx = box + ((size_t)b * A + a) * 11;
    float v[SIMPB_RECORD3D_WIDTH];
    v[0] = bx[0]; v[1] = bx[1]; v[2] = bx[2];
    v[3] = expf(bx[3]); v[4] = expf(bx[4]); v[5] = expf(bx[5]);
    v[6] = atan2f(bx[6], bx[7]);
    v[7] = bx[8]; v[8] = bx[9]; v[9] = bx[10];
    v[10] = score;
    v[11] = (float)label[a];
    v[12] = __uint_as_float(uo);
    // the int64 track id travels bit-exactly as two 32-bit lanes (a float holds integers only up to 2^24, which a
    // stream passes after ~56 k frames at 300 fresh ids per frame); read back with an int64 view of columns 13:15
    const long long id = instance_id ? instance_id[(size_t)b * A + a] : -1ll;
    v[13] = __uint_as_float((unsigned)((unsigned long long)id & 0xFFFFFFFFull));
    v[14] = __uint_as_float((unsigned)((unsigned long long)id >> 32));
#pragma unroll
    for (int k = 0; k < SIMPB_RECORD3D_WIDTH; ++k) simpb::pin(v[k]);
    simpb::loads_retired();  // store_fence.h
    float* o = rec3d + ((size_t)b * K + r) * SIMPB_RECORD3D_WIDTH;
#pragma unroll
    for (int k = 0; k < SIMPB_RECORD3D_WIDTH; ++k) o[k] = v[k];
    rank_of_anchor[(size_t)b * A + a] = r;
    simpb::stores_retired();  // a second trip (K > 512) starts with nothing of this one in flight
  }
}

// 2D half (:168-175, decode_box2d :36-51), one thread per 2D slot.
__global__ __launch_bounds__(256) void decode2d_kernel(float* __restrict__ rec2d, const float* __restrict__ cls2d,
                                                       const float* __restrict__ box2d, const int* __restrict__ q2a,
                                                       const int* __restrict__ query_cam, const int* __restrict__ rank_of_anchor,
                                                       int bs, int N2, int C, int A, float crop_w, float crop_h, float crop_y0,
                                                       float inv_resize) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs * N2) return;
  const int b = i / N2, slot = i - b * N2;
  const float* row = cls2d + (size_t)i * C;
  float m = row[0];
  int arg = 0;
  for (int c = 1; c < C; ++c)
    if (row[c] > m) { m = row[c]; arg = c; }
  const float* bx = box2d + (size_t)i * 4;
  const float cx = bx[0], cy = bx[1], w = bx[2], h = bx[3];
  float r[8];
  r[0] = fminf(fmaxf((cx - 0.5f * w) * crop_w, 0.f), crop_w) * inv_resize;
  r[1] = (fminf(fmaxf((cy - 0.5f * h) * crop_h, 0.f), crop_h) + crop_y0) * inv_resize;
  r[2] = fminf(fmaxf((cx + 0.5f * w) * crop_w, 0.f), crop_w) * inv_resize;
  r[3] = (fminf(fmaxf((cy + 0.5f * h) * crop_h, 0.f), crop_h) + crop_y0) * inv_resize;
  r[4] = sigmoidf(m);
  r[5] = (float)arg;
  const int a = q2a[i];
  r[6] = (a >= 0 && a < A) ? (float)rank_of_anchor[(size_t)b * A + a] : -1.f;
  r[7] = (float)query_cam[slot];
#pragma unroll
  for (int k = 0; k < 8; ++k) simpb::pin(r[k]);
  simpb::loads_retired();  // store_fence.h
  float* o = rec2d + (size_t)i * 8;
  *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
}

// The 2D half for a batch of independent streams (alloc.hip: alloc_scatter_ragged_kernel): the slot array is flat,
// stream-major, group g = b * cams + cam; stream b's slots are group_start[b * cams] .. group_start[(b + 1) * cams). Writes
// the same per-stream record a batch of one writes -- rec2d [bs, rows, 8], a stream's slots in its leading rows, camera
// counted within the stream -- and pad rows (rank -1, camera -1) behind them. One thread per (stream, row); q2a holds flat
// anchor indices b * A + a, which is also rank_of_anchor's index.
__global__ __launch_bounds__(256) void decode2d_ragged_kernel(float* __restrict__ rec2d, const float* __restrict__ cls2d,
                                                              const float* __restrict__ box2d, const int* __restrict__ q2a,
                                                              const int* __restrict__ query_cam,
                                                              const int* __restrict__ group_start,
                                                              const int* __restrict__ rank_of_anchor, int bs, int rows, int cams,
                                                              int C, int A, float crop_w, float crop_h, float crop_y0,
                                                              float inv_resize) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs * rows) return;
  const int b = i / rows, r_in = i - b * rows;
  const int lo = group_start[b * cams], hi = group_start[(b + 1) * cams];
  const bool live = lo + r_in < hi;
  const int j = live ? lo + r_in : 0;   // (slot 0 is readable whenever anything is: clamped loads, zero-weighted below)
  const float* row = cls2d + (size_t)j * C;
  float m = row[0];
  int arg = 0;
  for (int c = 1; c < C; ++c)
    if (row[c] > m) { m = row[c]; arg = c; }
  const float* bx = box2d + (size_t)j * 4;
  const float cx = bx[0], cy = bx[1], w = bx[2], h = bx[3];
  float r[8];
  r[0] = fminf(fmaxf((cx - 0.5f * w) * crop_w, 0.f), crop_w) * inv_resize;
  r[1] = (fminf(fmaxf((cy - 0.5f * h) * crop_h, 0.f), crop_h) + crop_y0) * inv_resize;
  r[2] = fminf(fmaxf((cx + 0.5f * w) * crop_w, 0.f), crop_w) * inv_resize;
  r[3] = (fminf(fmaxf((cy + 0.5f * h) * crop_h, 0.f), crop_h) + crop_y0) * inv_resize;
  r[4] = sigmoidf(m);
  r[5] = (float)arg;
  const int a = q2a[j];
  r[6] = (a >= 0 && a < bs * A) ? (float)rank_of_anchor[a] : -1.f;
  r[7] = (float)(query_cam[j] - b * cams);
  if (!live) {
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = 0.f;
    r[6] = r[7] = -1.f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) simpb::pin(r[k]);
  simpb::loads_retired();  // store_fence.h
  float* o = rec2d + (size_t)i * 8;
  *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
}

// Exchange form of the 2D record (simpb_amd/dist.py DetectionGather): only the rows decode_with2d returns -- slots whose
// anchor is among the kept 3D boxes (rank >= 0, decoder.py:176-251) -- in their slot order, pad rows (rank -1, camera -1)
// behind. At most num_output x num_cams rows exist (an anchor holds one slot per camera), whatever the runner's slot
// capacity is and however it grows, so every rank can size its exchange buffer without talking to the others. One
// workgroup per stream, stable ballot compaction.
__global__ __launch_bounds__(256) void record2d_compact_kernel(float* __restrict__ out, const float* __restrict__ rec2d,
                                                               int rows_in, int rows_out, long long in_stride,
                                                               long long out_stride) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* in = rec2d + (size_t)b * in_stride;   // (strides in floats: the output may be a column range of a wider buffer)
  float* o = out + (size_t)b * out_stride;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int r0 = 0; r0 < rows_in; r0 += 256) {
    const int r = r0 + tid;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = make_float4(0.f, 0.f, -1.f, -1.f);
    if (r < rows_in) {
      lo = *reinterpret_cast<const float4*>(in + (size_t)r * 8);
      hi = *reinterpret_cast<const float4*>(in + (size_t)r * 8 + 4);
    }
    simpb::loads_retired();
    const bool keep = r < rows_in && hi.z >= 0.f && hi.w >= 0.f;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    const int dst = off + __popcll(m & ((1ull << lane) - 1ull));
    if (keep && dst < rows_out) {
      *reinterpret_cast<float4*>(o + (size_t)dst * 8) = lo;
      *reinterpret_cast<float4*>(o + (size_t)dst * 8 + 4) = hi;
    }
    simpb::stores_retired();
    __syncthreads();
    if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  for (int r = min(s_base, rows_out) + tid; r < rows_out; r += 256) {
    *reinterpret_cast<float4*>(o + (size_t)r * 8) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(o + (size_t)r * 8 + 4) = make_float4(0.f, 0.f, -1.f, -1.f);
  }
}

}  // namespace

extern "C" int simpb_decode3d_record(float* rec3d, int* rank_of_anchor, const float* cls, const float* quality,
                                     const float* box, const long long* instance_id, int batch_size, int num_anchors,
                                     int num_classes, int num_output, void* stream) {
  if (!rec3d || !rank_of_anchor || !cls || !box || batch_size <= 0 || num_anchors <= 0 || num_anchors > kCapA ||
      num_classes <= 0 || num_output <= 0 || num_output > kCapK || num_output > num_anchors)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(decode3d_kernel, dim3(batch_size), dim3(512), 0, static_cast<hipStream_t>(stream), rec3d, rank_of_anchor,
                     cls, quality, box, instance_id, num_anchors, num_classes, num_output);
  return simpb_check_launch();
}

extern "C" int simpb_decode2d_record(float* rec2d, const float* cls2d, const float* box2d, const int* q2a,
                                     const int* query_cam, const int* rank_of_anchor, int batch_size, int num_query2d,
                                     int num_classes, int num_anchors, float crop_w, float crop_h, float crop_y0,
                                     float resize, void* stream) {
  if (!rec2d || !cls2d || !box2d || !q2a || !query_cam || !rank_of_anchor || batch_size <= 0 || num_query2d <= 0 ||
      num_classes <= 0 || num_anchors <= 0 || resize == 0.f)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  const int total = batch_size * num_query2d;
  hipLaunchKernelGGL(decode2d_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), rec2d, cls2d,
                     box2d, q2a, query_cam, rank_of_anchor, batch_size, num_query2d, num_classes, num_anchors, crop_w, crop_h,
                     crop_y0, 1.f / resize);
  return simpb_check_launch();
}

extern "C" int simpb_decode2d_record_ragged(float* rec2d, const float* cls2d, const float* box2d, const int* q2a,
                                            const int* query_cam, const int* group_start, const int* rank_of_anchor,
                                            int batch_size, int rows_per_stream, int num_cams, int num_classes, int num_anchors,
                                            float crop_w, float crop_h, float crop_y0, float resize, void* stream) {
  if (!rec2d || !cls2d || !box2d || !q2a || !query_cam || !group_start || !rank_of_anchor || batch_size <= 0 ||
      rows_per_stream <= 0 || num_cams <= 0 || num_classes <= 0 || num_anchors <= 0 || resize == 0.f ||
      (long long)batch_size * rows_per_stream > 0x7fffffffll / 8)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  const int total = batch_size * rows_per_stream;
  hipLaunchKernelGGL(decode2d_ragged_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), rec2d,
                     cls2d, box2d, q2a, query_cam, group_start, rank_of_anchor, batch_size, rows_per_stream, num_cams,
                     num_classes, num_anchors, crop_w, crop_h, crop_y0, 1.f / resize);
  return simpb_check_launch();
}

extern "C" int simpb_record2d_compact(float* out, long long out_stride, const float* rec2d, long long in_stride, int batch_size,
                                      int rows_in, int rows_out, void* stream) {
  if (!out || !rec2d || batch_size <= 0 || rows_in <= 0 || rows_out <= 0 || batch_size > 65535 ||
      ((reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(rec2d)) & 15) || out_stride < (long long)rows_out * 8 ||
      in_stride < (long long)rows_in * 8 || ((out_stride | in_stride) & 3))
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(record2d_compact_kernel, dim3(batch_size), dim3(256), 0, static_cast<hipStream_t>(stream), out, rec2d, rows_in,
                     rows_out, in_stride, out_stride);
  return simpb_check_launch();
}
