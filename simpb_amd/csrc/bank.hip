// InstanceBank (/root/reference/projects/mmdet3d_plugin/models/instance_bank.py) state updates as a
// handful of launches. The arithmetic is tiny (900 instances per stream); in PyTorch each of the three
// touch points of a frame is a run of 10-45 launches of 2-5 us each on the decoder's critical path:
//   get    (:79-119)  warp the cached anchors into the current frame, validity mask, time step
//   update (:121-150) after the first decoder layer: [cached 600 | best 300 current]
//   cache + get_instance_id + update_instance_id (:152-196) at the end of the frame
#include <hip/hip_runtime.h>
#include "../../include/simpb_hip.h"
#include "store_fence.h"

extern "C" int simpb_check_launch(void);

namespace {

constexpr int kCap = 1024;  // instances per stream (900 shipped)

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ unsigned long long sort_key(float v, unsigned idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (0xFFFFFFFFu - idx);
}
__device__ __forceinline__ unsigned key_index(unsigned long long k) { return 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull); }
__device__ __forceinline__ float key_value(unsigned long long k) {
  unsigned u = (unsigned)(k >> 32);
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(u);
}
// descending bitonic sort of kCap keys in LDS by 512 threads
__device__ __forceinline__ void bitonic_desc(unsigned long long* key, int tid) {
  for (int size = 2; size <= kCap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int lo = 2 * tid - (tid & (stride - 1)), hi = lo + stride;
      const bool desc = (lo & size) == 0;
      const unsigned long long a = key[lo], c = key[hi];
      if ((a < c) == desc) { key[lo] = c; key[hi] = a; }
      __syncthreads();
    }
  }
}
__device__ __forceinline__ float row_max(const float* row, int C) {
  float m = row[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
  return m;
}

// ---- get (:83-113): one thread per cached anchor + one per stream for the mask / time step
__global__ __launch_bounds__(256) void bank_get_kernel(float* __restrict__ out, unsigned char* __restrict__ mask,
                                                       float* __restrict__ dt_out, const float* __restrict__ anchor,
                                                       const float* __restrict__ T, const float* __restrict__ dt, int bs,
                                                       int n, float max_dt, float default_dt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < bs) {
    const float t = dt[i];
    simpb::loads_retired();
    const bool ok = fabsf(t) <= max_dt;  // :87
    mask[i] = ok ? 1 : 0;
    dt_out[i] = (t != 0.f && ok) ? t : default_dt;  // :108-113
    simpb::loads_retired();  // (a store fence as well: nothing in flight when the rows below are loaded)
  }
  if (i >= bs * n) return;
  const int b = i / n;
  float a[11], m[12];
  {
    const float* ap = anchor + (size_t)i * 11;
    const float* mp = T + (size_t)b * 16;
#pragma unroll
    for (int k = 0; k < 11; ++k) a[k] = ap[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = mp[k];
  }
  const float t = -dt[b];  // anchor_projection(..., time_intervals=[-time_interval]) (:98-101)
  simpb::loads_retired();  // every consumer behind a full vmcnt(0): no counted waits in this kernel
  const float vx = a[8], vy = a[9], vz = a[10];
  const float cx = a[0] - vx * t, cy = a[1] - vy * t, cz = a[2] - vz * t;
  const float s = a[6], c = a[7];  // yaw pair as written in detection3d/blocks.py:271-278
  float r[11];
  r[0] = m[0] * cx + m[1] * cy + m[2] * cz + m[3];
  r[1] = m[4] * cx + m[5] * cy + m[6] * cz + m[7];
  r[2] = m[8] * cx + m[9] * cy + m[10] * cz + m[11];
  r[3] = a[3]; r[4] = a[4]; r[5] = a[5];
  r[6] = m[0] * c + m[1] * s;
  r[7] = m[4] * c + m[5] * s;
  r[8] = m[0] * vx + m[1] * vy + m[2] * vz;
  r[9] = m[4] * vx + m[5] * vy + m[6] * vz;
  r[10] = m[8] * vx + m[9] * vy + m[10] * vz;
  // all eleven results in registers and every load retired before the first store (store_fence.h): scheduled freely,
  // the pass-through columns were stored among seven outstanding loads and this kernel faulted beside a busy queue
#pragma unroll
  for (int k = 0; k < 11; ++k) simpb::pin(r[k]);
  simpb::loads_retired();
  float* o = out + (size_t)i * 11;
#pragma unroll
  for (int k = 0; k < 11; ++k) o[k] = r[k];
}

// ---- update (:137-139): indices of the `fresh` best current instances by max class logit, one workgroup per stream
__global__ __launch_bounds__(512) void bank_update_rank_kernel(int* __restrict__ index, const float* __restrict__ cls, int A,
                                                               int C, int fresh) {
  __shared__ unsigned long long key[kCap];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int a = tid; a < kCap; a += 512) key[a] = a < A ? sort_key(row_max(cls + ((size_t)b * A + a) * C, C), (unsigned)a) : 0ull;
  __syncthreads();
  bitonic_desc(key, tid);
  for (int r = tid; r < fresh; r += 512) index[(size_t)b * fresh + r] = (int)key_index(key[r]);
}

// A frame whose 2D query set overflowed its static capacity is re-run by the caller with a larger one
// (simpb_amd/runner.py); `hold` are that frame's overflow flags (simpb_alloc_group_start): when any is set the
// frame-end commit below leaves the persistent state exactly as the frame found it. `sticky` (device word, may be NULL)
// chains the hold across frames that are decoded before the host has seen the previous frame's flags: the commit of a
// frame writes "I held back" into it, and the next frame's update / commit treat a set word like a flag of their own
// (that next frame computed on a state that is going to be rolled back, so it is re-run as well: runner.py).
__device__ __forceinline__ bool held(const int* hold, int num_hold, const int* sticky = nullptr) {
  bool h = sticky && *sticky != 0;
  for (int k = 0; k < num_hold; ++k) h |= hold[k] != 0;
  return h;
}

// ---- update (:140-149): rows [0, T) = cached, rows [T, A) = current[index], per stream under its mask; one
// wave per output row (feature C floats + anchor 11 floats [+ embedding E floats]); the ids of masked-out streams are reset
// (:147-149) -- unless a `hold` flag is set: then this frame (or the one decoded just before it) is going to be re-run from
// the state it found, and that state includes the ids.
// Embeddings (optional): the anchor embedding is a row-wise function of the anchor (SparseBox3DEncoder), so the embedding
// of the merged set is the same merge of the embeddings of its two sources -- cached_e = encoder(cached anchors as warped
// by bank_get), cur_e = encoder(current anchors) -- and the encoder launch behind the update (simpb_head.py:621-622) is a
// row gather here.
__global__ __launch_bounds__(64) void bank_merge_kernel(float* __restrict__ out_f, float* __restrict__ out_a,
                                                        long long* __restrict__ instance_id,
                                                        const float* __restrict__ cur_f, const float* __restrict__ cur_a,
                                                        const float* __restrict__ cached_f, const float* __restrict__ cached_a,
                                                        const int* __restrict__ index, const unsigned char* __restrict__ mask,
                                                        int A, int T, int C, float* __restrict__ out_e,
                                                        const float* __restrict__ cur_e, const float* __restrict__ cached_e, int E,
                                                        const int* __restrict__ hold, int num_hold,
                                                        const int* __restrict__ sticky) {
  const int r = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const bool use = mask[b] != 0;
  const float* sf;
  const float* sa;
  const float* se = nullptr;
  if (!use) {
    sf = cur_f + ((size_t)b * A + r) * C;
    sa = cur_a + ((size_t)b * A + r) * 11;
    if (out_e) se = cur_e + ((size_t)b * A + r) * E;
  } else if (r < T) {
    sf = cached_f + ((size_t)b * T + r) * C;
    sa = cached_a + ((size_t)b * T + r) * 11;
    if (out_e) se = cached_e + ((size_t)b * T + r) * E;
  } else {
    const int src = index[(size_t)b * (A - T) + (r - T)];
    sf = cur_f + ((size_t)b * A + src) * C;
    sa = cur_a + ((size_t)b * A + src) * 11;
    if (out_e) se = cur_e + ((size_t)b * A + src) * E;
  }
  float* of = out_f + ((size_t)b * A + r) * C;
  for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(of + c) = *reinterpret_cast<const float4*>(sf + c);
  if (out_e) {
    float* oe = out_e + ((size_t)b * A + r) * E;
    for (int c = lane * 4; c < E; c += 256) *reinterpret_cast<float4*>(oe + c) = *reinterpret_cast<const float4*>(se + c);
  }
  if (lane < 11) out_a[((size_t)b * A + r) * 11 + lane] = sa[lane];
  if (!use && instance_id && lane == 0 && !held(hold, num_hold, sticky)) instance_id[(size_t)b * A + r] = -1;
}

// ---- cache (:152-167) + get_instance_id (:169-184) + update_instance_id (:186-196): ONE workgroup walks the
// streams in order (fresh track ids are numbered over the flattened batch, :179-181).
__global__ __launch_bounds__(512) void bank_cache_kernel(float* __restrict__ conf, int* __restrict__ index,
                                                         long long* __restrict__ ids_out, long long* __restrict__ instance_id,
                                                         long long* __restrict__ prev_id, const float* __restrict__ cls,
                                                         int bs, int A, int C, int T, int has_prev, float decay,
                                                         int has_threshold, float threshold, const int* __restrict__ hold,
                                                         int num_hold, int* __restrict__ sticky) {
  __shared__ unsigned long long key[kCap];
  __shared__ long long ids[kCap];
  __shared__ int scan[kCap];
  __shared__ float fresh_score[kCap];
  const int tid = threadIdx.x;
  const bool hold_back = held(hold, num_hold, sticky);
  if (sticky) {   // what the frame decoded after this one has to respect (every thread has read the word before it changes)
    __syncthreads();
    if (tid == 0) *sticky = hold_back ? 1 : 0;
  }
  if (hold_back) return;  // workgroup-uniform: the frame is re-run by the caller, the state stays as it was
  long long next_id = *prev_id;
  for (int b = 0; b < bs; ++b) {
    // scores: sigmoid of the best class; tracked instances keep max(decayed previous, new) (:157-162)
    for (int a = tid; a < kCap; a += 512) {
      unsigned long long kv = 0ull;
      if (a < A) {
        const float s = sigmoidf(row_max(cls + ((size_t)b * A + a) * C, C));
        fresh_score[a] = s;
        float sc = s;
        if (has_prev && a < T) sc = fmaxf(conf[(size_t)b * T + a] * decay, s);
        kv = sort_key(sc, (unsigned)a);
      }
      key[a] = kv;
    }
    __syncthreads();
    bitonic_desc(key, tid);
    for (int r = tid; r < T; r += 512) {
      conf[(size_t)b * T + r] = key_value(key[r]);
      index[(size_t)b * T + r] = (int)key_index(key[r]);
    }
    // track ids (:172-183): instances without an id (and above the threshold) get consecutive fresh ones
    for (int a = tid; a < kCap; a += 512) {
      long long id = -1;
      int fresh = 0;
      if (a < A) {
        id = instance_id ? instance_id[(size_t)b * A + a] : -1;
        fresh = id < 0 && (!has_threshold || fresh_score[a] >= threshold);
      }
      ids[a] = id;
      scan[a] = fresh;
    }
    __syncthreads();
    // inclusive scan of the 1024 flags (Hillis-Steele, 10 steps, 2 elements per thread)
    for (int off = 1; off < kCap; off <<= 1) {
      const int a0 = tid, a1 = tid + 512;
      const int v0 = a0 >= off ? scan[a0 - off] : 0, v1 = scan[a1 - off];
      __syncthreads();
      scan[a0] += v0;
      scan[a1] += v1;
      __syncthreads();
    }
    const int total = scan[kCap - 1];
    for (int a = tid; a < A; a += 512) {
      const int inc = scan[a], prev = a ? scan[a - 1] : 0;
      if (inc != prev) ids[a] = next_id + (inc - 1);
    }
    __syncthreads();
    for (int a = tid; a < A; a += 512) {
      ids_out[(size_t)b * A + a] = ids[a];
      if (instance_id) instance_id[(size_t)b * A + a] = a < T ? ids[key_index(key[a])] : -1;  // :191-195
    }
    next_id += total;
    __syncthreads();
  }
  if (tid == 0) *prev_id = next_id;
}

// ---- the same commit with ONE WORKGROUP PER STREAM (a batch of streams: the serial walk above is 13 us per stream, 107 us at
// eight). Fresh track ids are numbered over the flattened batch (:179-181), so stream b's first fresh id is prev_id + the
// number of fresh instances of the streams in front of it: every workgroup counts those itself (ids < 0 and, with a
// threshold, score >= threshold: a few thousand loads), which needs the OTHER streams' instance_id as the frame found them
// -- and every stream rewrites its own instance_id below. So between "counted" and "rewrite" the workgroups meet once:
// an arrival counter in device memory (sync[0], never reset: launch e waits for (e + 1) * bs arrivals, e = sync[1], which
// workgroup 0 advances behind the meeting). All bs workgroups are resident at once (bs <= 64 workgroups of 512 threads on 256
// CUs), every one of them arrives before it waits, and the wait is bounded (a launch that could not be co-scheduled would
// fall through after ~2^22 polls instead of hanging). prev_id is read by everyone in front of the meeting and written by the
// last stream's workgroup behind it.
__global__ __launch_bounds__(512) void bank_cache_streams_kernel(float* __restrict__ conf, int* __restrict__ index,
                                                                 long long* __restrict__ ids_out,
                                                                 long long* __restrict__ instance_id,
                                                                 long long* __restrict__ prev_id, const float* __restrict__ cls,
                                                                 int bs, int A, int C, int T, int has_prev, float decay,
                                                                 int has_threshold, float threshold,
                                                                 const int* __restrict__ hold, int num_hold,
                                                                 int* __restrict__ sticky, unsigned* __restrict__ sync) {
  __shared__ unsigned long long key[kCap];
  __shared__ long long ids[kCap];
  __shared__ int scan[kCap];
  __shared__ float fresh_score[kCap];
  __shared__ int s_red[8];
  __shared__ int s_front;
  const int tid = threadIdx.x, b = blockIdx.x;
  const bool hold_back = held(hold, num_hold, sticky);
  if (sticky) {
    __syncthreads();
    if (tid == 0) *sticky = hold_back ? 1 : 0;   // (every workgroup writes the same verdict: a set word stays set, see held())
  }
  if (hold_back) return;  // uniform over the whole launch: nobody arrives, the counters stay where they are
  const long long first_id = *prev_id;
  const unsigned epoch = __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // fresh instances of the streams in front of this one
  int mine = 0;
  for (int i = tid; i < b * A; i += 512) {
    const long long id = instance_id ? instance_id[i] : -1;
    if (id < 0 && (!has_threshold || sigmoidf(row_max(cls + (size_t)i * C, C)) >= threshold)) ++mine;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m);
  if ((tid & 63) == 0) s_red[tid >> 6] = mine;
  // this stream's scores and ranking (as the serial kernel)
  for (int a = tid; a < kCap; a += 512) {
    unsigned long long kv = 0ull;
    if (a < A) {
      const float s = sigmoidf(row_max(cls + ((size_t)b * A + a) * C, C));
      fresh_score[a] = s;
      float sc = s;
      if (has_prev && a < T) sc = fmaxf(conf[(size_t)b * T + a] * decay, s);
      kv = sort_key(sc, (unsigned)a);
    }
    key[a] = kv;
  }
  __syncthreads();
  if (tid == 0) {
    int f = 0;
    for (int w = 0; w < 8; ++w) f += s_red[w];
    s_front = f;
  }
  bitonic_desc(key, tid);
  for (int a = tid; a < kCap; a += 512) {
    long long id = -1;
    int fresh = 0;
    if (a < A) {
      id = instance_id ? instance_id[(size_t)b * A + a] : -1;
      fresh = id < 0 && (!has_threshold || fresh_score[a] >= threshold);
    }
    ids[a] = id;
    scan[a] = fresh;
  }
  __syncthreads();
  for (int off = 1; off < kCap; off <<= 1) {
    const int a0 = tid, a1 = tid + 512;
    const int v0 = a0 >= off ? scan[a0 - off] : 0, v1 = scan[a1 - off];
    __syncthreads();
    scan[a0] += v0;
    scan[a1] += v1;
    __syncthreads();
  }
  const int total = scan[kCap - 1];
  const long long next_id = first_id + s_front;
  for (int a = tid; a < A; a += 512) {
    const int inc = scan[a], prev = a ? scan[a - 1] : 0;
    if (inc != prev) ids[a] = next_id + (inc - 1);
  }
  __syncthreads();
  // ---- the meeting: every stream has read what it needs of the others' instance_id and of prev_id
  if (tid == 0) {
    __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (epoch + 1u) * (unsigned)bs;
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if ((int)(__hip_atomic_load(&sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  for (int r = tid; r < T; r += 512) {
    conf[(size_t)b * T + r] = key_value(key[r]);
    index[(size_t)b * T + r] = (int)key_index(key[r]);
  }
  for (int a = tid; a < A; a += 512) {
    ids_out[(size_t)b * A + a] = ids[a];
    if (instance_id) instance_id[(size_t)b * A + a] = a < T ? ids[key_index(key[a])] : -1;  // :191-195
  }
  if (tid == 0) {
    if (b == bs - 1) *prev_id = next_id + total;
    if (b == 0) __hip_atomic_store(&sync[1], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- cache: kept rows of the feature / anchor tables into the persistent state, one wave per row
__global__ __launch_bounds__(64) void bank_gather_kernel(float* __restrict__ out_f, float* __restrict__ out_a,
                                                         const float* __restrict__ src_f, const float* __restrict__ src_a,
                                                         const int* __restrict__ index, int A, int T, int C,
                                                         const int* __restrict__ hold, int num_hold,
                                                         const int* __restrict__ sticky) {
  const int r = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  if (held(hold, num_hold, sticky)) return;   // (sticky: as bank_cache_kernel left it one launch earlier)
  const int src = index[(size_t)b * T + r];
  const float* sf = src_f + ((size_t)b * A + src) * C;
  float* of = out_f + ((size_t)b * T + r) * C;
  for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(of + c) = *reinterpret_cast<const float4*>(sf + c);
  if (lane < 11) out_a[((size_t)b * T + r) * 11 + lane] = src_a[((size_t)b * A + src) * 11 + lane];
}

}  // namespace

extern "C" int simpb_bank_get(float* anchor_out, unsigned char* mask_out, float* time_interval_out, const float* cached_anchor,
                              const float* T_temp2cur, const float* time_interval, int batch_size, int num_temp,
                              float max_time_interval, float default_time_interval, void* stream) {
  if (!anchor_out || !mask_out || !time_interval_out || !cached_anchor || !T_temp2cur || !time_interval || batch_size <= 0 ||
      num_temp <= 0)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  const int total = batch_size * num_temp;
  hipLaunchKernelGGL(bank_get_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), anchor_out,
                     mask_out, time_interval_out, cached_anchor, T_temp2cur, time_interval, batch_size, num_temp,
                     max_time_interval, default_time_interval);
  return simpb_check_launch();
}

extern "C" int simpb_bank_update_rank(int* index_scratch, const float* cls, int batch_size, int num_anchors, int num_classes,
                                      int num_temp, void* stream) {
  if (!index_scratch || !cls || batch_size <= 0 || num_anchors <= 0 || num_anchors > kCap || num_classes <= 0 || num_temp <= 0 ||
      num_temp >= num_anchors || batch_size > 65535)
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(bank_update_rank_kernel, dim3(batch_size), dim3(512), 0, static_cast<hipStream_t>(stream), index_scratch, cls,
                     num_anchors, num_classes, num_anchors - num_temp);
  return simpb_check_launch();
}

extern "C" int simpb_bank_update_merge(float* feature_out, float* anchor_out, float* embed_out, long long* instance_id,
                                       const int* index, const float* feature, const float* anchor, const float* embed,
                                       const float* cached_feature, const float* cached_anchor, const float* cached_embed,
                                       const unsigned char* mask, const int* hold, int num_hold, const int* sticky,
                                       int batch_size, int num_anchors, int num_temp, int embed_dims, int pos_embed_dims,
                                       void* stream) {
  if (!feature_out || !anchor_out || !index || !feature || !anchor || !cached_feature || !cached_anchor || !mask ||
      batch_size <= 0 || num_anchors <= 0 || num_anchors > kCap || num_temp <= 0 || num_temp >= num_anchors || embed_dims <= 0 ||
      (embed_dims & 3) || batch_size > 65535 || num_hold < 0 || (num_hold > 0 && !hold))
    return SIMPB_EINVAL;
  if (embed_out && (!embed || !cached_embed || pos_embed_dims <= 0 || (pos_embed_dims & 3))) return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(bank_merge_kernel, dim3(num_anchors, batch_size), dim3(64), 0, static_cast<hipStream_t>(stream), feature_out,
                     anchor_out, instance_id, feature, anchor, cached_feature, cached_anchor, index, mask, num_anchors, num_temp,
                     embed_dims, embed_out, embed, cached_embed, pos_embed_dims, hold, num_hold, sticky);
  return simpb_check_launch();
}

extern "C" int simpb_bank_update(float* feature_out, float* anchor_out, long long* instance_id, int* index_scratch,
                                 const float* feature, const float* anchor, const float* cls, const float* cached_feature,
                                 const float* cached_anchor, const unsigned char* mask, int batch_size, int num_anchors,
                                 int num_classes, int num_temp, int embed_dims, void* stream) {
  const int st = simpb_bank_update_rank(index_scratch, cls, batch_size, num_anchors, num_classes, num_temp, stream);
  if (st != SIMPB_OK) return st;
  return simpb_bank_update_merge(feature_out, anchor_out, nullptr, instance_id, index_scratch, feature, anchor, nullptr,
                                 cached_feature, cached_anchor, nullptr, mask, nullptr, 0, nullptr, batch_size, num_anchors,
                                 num_temp, embed_dims, 0, stream);
}

static int bank_cache_launch(float* confidence, float* cached_feature, float* cached_anchor, long long* instance_id,
                             long long* prev_id, long long* ids_out, int* index_scratch, const float* feature,
                             const float* anchor, const float* cls, int batch_size, int num_anchors, int num_classes,
                             int num_temp, int embed_dims, int has_previous, float confidence_decay, int has_threshold,
                             float threshold, const int* hold, int num_hold, int* sticky, unsigned* sync, void* stream) {
  if (!confidence || !cached_feature || !cached_anchor || !prev_id || !ids_out || !index_scratch || !feature || !anchor ||
      !cls || batch_size <= 0 || num_anchors <= 0 || num_anchors > kCap || num_classes <= 0 || num_temp <= 0 ||
      num_temp > num_anchors || embed_dims <= 0 || (embed_dims & 3) || batch_size > 65535 || num_hold < 0 ||
      (num_hold > 0 && !hold))
    return SIMPB_EINVAL;
  (void)hipGetLastError();
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (sync && batch_size > 1 && batch_size <= 64)
    hipLaunchKernelGGL(bank_cache_streams_kernel, dim3(batch_size), dim3(512), 0, s, confidence, index_scratch, ids_out, instance_id,
                       prev_id, cls, batch_size, num_anchors, num_classes, num_temp, has_previous, confidence_decay, has_threshold,
                       threshold, hold, num_hold, sticky, sync);
  else
    hipLaunchKernelGGL(bank_cache_kernel, dim3(1), dim3(512), 0, s, confidence, index_scratch, ids_out, instance_id, prev_id, cls,
                       batch_size, num_anchors, num_classes, num_temp, has_previous, confidence_decay, has_threshold, threshold, hold,
                       num_hold, sticky);
  hipLaunchKernelGGL(bank_gather_kernel, dim3(num_temp, batch_size), dim3(64), 0, s, cached_feature, cached_anchor, feature,
                     anchor, index_scratch, num_anchors, num_temp, embed_dims, hold, num_hold, sticky);
  return simpb_check_launch();
}

extern "C" int simpb_bank_cache(float* confidence, float* cached_feature, float* cached_anchor, long long* instance_id,
                                long long* prev_id, long long* ids_out, int* index_scratch, const float* feature,
                                const float* anchor, const float* cls, int batch_size, int num_anchors, int num_classes,
                                int num_temp, int embed_dims, int has_previous, float confidence_decay, int has_threshold,
                                float threshold, const int* hold, int num_hold, int* sticky, void* stream) {
  return bank_cache_launch(confidence, cached_feature, cached_anchor, instance_id, prev_id, ids_out, index_scratch, feature, anchor,
                           cls, batch_size, num_anchors, num_classes, num_temp, embed_dims, has_previous, confidence_decay,
                           has_threshold, threshold, hold, num_hold, sticky, nullptr, stream);
}

extern "C" int simpb_bank_cache_streams(float* confidence, float* cached_feature, float* cached_anchor, long long* instance_id,
                                        long long* prev_id, long long* ids_out, int* index_scratch, const float* feature,
                                        const float* anchor, const float* cls, int batch_size, int num_anchors, int num_classes,
                                        int num_temp, int embed_dims, int has_previous, float confidence_decay,
                                        int has_threshold, float threshold, const int* hold, int num_hold, int* sticky,
                                        unsigned* sync_words, void* stream) {
  return bank_cache_launch(confidence, cached_feature, cached_anchor, instance_id, prev_id, ids_out, index_scratch, feature, anchor,
                           cls, batch_size, num_anchors, num_classes, num_temp, embed_dims, has_previous, confidence_decay,
                           has_threshold, threshold, hold, num_hold, sticky, sync_words, stream);
}
