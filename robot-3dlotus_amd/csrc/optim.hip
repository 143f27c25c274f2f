// lotus-hip: the optimiser step of the 3D-LOTUS trainer as two multi-tensor launches:
//   * global gradient norm + clip coefficient  (torch.nn.utils.clip_grad_norm_, train_simple_policy.py:237-241)
//   * HF-style AdamW over every parameter tensor (genrobo3d/train/optim/adamw.py:53-112)
// The reference runs ~8 ATen kernels per tensor from a Python loop (421 tensors); here every tensor is cut into
// chunks of MT_CHUNK elements and one block updates one chunk, driven by device-resident pointer tables.
#include "common.h"

#define MT_CHUNK 4096  // elements per block: 256 threads x 4 float4

// partial[c] = sum of squares of chunk c (double accumulation, fixed tree -> deterministic)
__global__ __launch_bounds__(256) void mt_sqnorm_kernel(const float* const* __restrict__ g_ptrs, const long* __restrict__ numel,
                                                        const int* __restrict__ chunks, double* __restrict__ partial) {
  __shared__ double red[4];
  const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
  const float* g = g_ptrs[t];
  const long n = numel[t], beg = (long)c * MT_CHUNK;
  double s = 0.0;
  if (g) {
    const bool vec = (((uintptr_t)g) & 15) == 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long i = beg + (long)(it * 256 + threadIdx.x) * 4;
      if (vec && i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      } else {
        for (int e = 0; e < 4 && i + e < n; ++e) s += (double)g[i + e] * g[i + e];
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total L2 norm, out[1] = clip coefficient min(max_norm / (norm + 1e-6), 1)   (one block, fixed order)
__global__ __launch_bounds__(1024) void mt_norm_finish_kernel(const double* __restrict__ partial, int n, float max_norm,
                                                              float* __restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += partial[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += red[w];
    const float norm = (float)sqrt(tot);
    out[0] = norm;
    out[1] = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
  }
}

struct AdamP {
  float* const* p;
  const float* const* g;
  float* const* m;
  float* const* v;
  const long* numel;
  const float* step_size;  // [T]  lr * sqrt(1 - b2^t) / (1 - b1^t), evaluated in double by the caller (adamw.py:93-98)
  const float* decay;      // [T]  lr * weight_decay (0 = none)                                         (adamw.py:109-110)
  const int* chunks;       // [nchunks][2] tensor, chunk
  const float* clip;       // device scalar (mt_norm_finish_kernel out + 1) or null
  // optional bf16 shadows of the parameters (null table / null entries = none): the updated value is stored a second time,
  // rounded to bf16 — the weight operand of the bf16-storage products (BASELINE configs[4]: "bf16 weights with fp32
  // master weights") refreshed by the step that changes the master, one extra 2-byte store per element
  unsigned short* const* shadow;
  // optional usage mask (data-parallel training, parallel.GradReducer): used[used_idx ? used_idx[t] : t] == 0 means NO rank
  // produced a gradient for tensor t in this step -> the tensor is skipped exactly as a null gradient is (adamw.py:67-68).  The
  // flags are the MAX all-reduce of the ranks' per-parameter usage bits and never leave the device, so a parameter whose usage
  // flips is handled in the step it flips in, without a host synchronisation.
  const int* used;
  const int* used_idx;
  // omb = 1 - beta evaluated in DOUBLE by the host and rounded once, as the reference's `addcmul_(g, g, value=1.0 - beta2)`
  // does (adamw.py:86-87: a Python float); 1.f - beta2 in fp32 is off by 1e-6 relative (cancellation), which showed up as
  // 8 ulp in exp_avg_sq
  float beta1, beta2, omb1, omb2, eps;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float ss, float dec, float b1, float b2,
                                          float omb1, float omb2, float eps) {
  m = m * b1 + g * omb1;
  v = v * b2 + (g * g) * omb2;
  const float denom = sqrtf(v) + eps;
  p = p + (-ss) * (m / denom);
  if (dec > 0.f) p = p + (-dec) * p;
}

__global__ __launch_bounds__(256) void mt_adamw_kernel(AdamP a) {
  const int t = a.chunks[2 * blockIdx.x], c = a.chunks[2 * blockIdx.x + 1];
  const float* g = a.g[t];
  if (!g) return;  // parameter without a gradient this step (adamw.py:67-68)
  if (a.used && a.used[a.used_idx ? a.used_idx[t] : t] == 0) return;  // ... on any rank (device-side usage mask)
  float* p = a.p[t];
  float* m = a.m[t];
  float* v = a.v[t];
  const long n = a.numel[t], beg = (long)c * MT_CHUNK;
  const float ss = a.step_size[t], dec = a.decay[t], coef = a.clip ? a.clip[0] : 1.f;
  __bf16* sh = a.shadow ? reinterpret_cast<__bf16*>(a.shadow[t]) : nullptr;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v) | (((uintptr_t)sh) << 1)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long i = beg + (long)(it * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 3 < n) {
      float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      adam_elem(pv.x, gv.x * coef, mv.x, vv.x, ss, dec, a.beta1, a.beta2, a.omb1, a.omb2, a.eps);
      adam_elem(pv.y, gv.y * coef, mv.y, vv.y, ss, dec, a.beta1, a.beta2, a.omb1, a.omb2, a.eps);
      adam_elem(pv.z, gv.z * coef, mv.z, vv.z, ss, dec, a.beta1, a.beta2, a.omb1, a.omb2, a.eps);
      adam_elem(pv.w, gv.w * coef, mv.w, vv.w, ss, dec, a.beta1, a.beta2, a.omb1, a.omb2, a.eps);
      *reinterpret_cast<float4*>(p + i) = pv;
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
      if (sh) st4(sh + i, pv);
    } else {
      for (int e = 0; e < 4 && i + e < n; ++e) {
        adam_elem(p[i + e], g[i + e] * coef, m[i + e], v[i + e], ss, dec, a.beta1, a.beta2, a.omb1, a.omb2, a.eps);
        if (sh) sh[i + e] = (__bf16)p[i + e];
      }
    }
  }
}

extern "C" {

int lotus_mt_chunk(void) { return MT_CHUNK; }

// norm_out[0] = ||g||_2 over all tensors, norm_out[1] = clip coefficient for max_norm (1 if max_norm <= 0).
// g_ptrs [T] (null entries are skipped), numel [T], chunks [nchunks][2] and partial [nchunks] are device memory.
int lotus_grad_norm(const void* g_ptrs, const long* numel, const int* chunks, int nchunks, double* partial, float* norm_out,
                    float max_norm, void* stream) {
  LOTUS_CHECK_ARG(g_ptrs && numel && chunks && partial && norm_out && nchunks >= 0, "lotus_grad_norm: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (nchunks > 0)
    LOTUS_LAUNCH(mt_sqnorm_kernel, dim3(nchunks), dim3(256), 0, st, (const float* const*)g_ptrs, numel, chunks, partial);
  LOTUS_LAUNCH(mt_norm_finish_kernel, dim3(1), dim3(1024), 0, st, (const double*)partial, nchunks, max_norm, norm_out);
  LOTUS_LAUNCH_CHECK("lotus_grad_norm");
  return LOTUS_OK;
}

// One AdamW step for every tensor.  step_size / decay are per-tensor device arrays (see AdamP); clip_coef (device scalar,
// optional) scales the gradients first (clip_grad_norm_ folded into the update instead of rewriting the gradients).
// used / used_idx (device, optional): usage mask of the data-parallel step, see AdamP.
int lotus_adamw_step(const void* p_ptrs, const void* g_ptrs, const void* m_ptrs, const void* v_ptrs, const long* numel,
                     const float* step_size, const float* decay, const int* chunks, int nchunks, double beta1, double beta2,
                     double eps, const float* clip_coef, const void* shadow_ptrs, const int* used, const int* used_idx,
                     void* stream) {
  LOTUS_CHECK_ARG(p_ptrs && g_ptrs && m_ptrs && v_ptrs && numel && step_size && decay && chunks && nchunks >= 0,
                  "lotus_adamw_step: bad arguments");
  if (nchunks == 0) return LOTUS_OK;
  AdamP a;
  a.p = (float* const*)p_ptrs; a.g = (const float* const*)g_ptrs; a.m = (float* const*)m_ptrs; a.v = (float* const*)v_ptrs;
  a.numel = numel; a.step_size = step_size; a.decay = decay; a.chunks = chunks; a.clip = clip_coef;
  a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
  a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
  a.shadow = (unsigned short* const*)shadow_ptrs;
  a.used = used; a.used_idx = used_idx;
  LOTUS_LAUNCH(mt_adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a);
  LOTUS_LAUNCH_CHECK("lotus_adamw_step");
  return LOTUS_OK;
}

// dst[t][i] = bf16(src[t][i]) for every tensor of the tables (creation / refresh of weight shadows outside an optimiser step)
int lotus_shadow_cast(const void* src_ptrs, const void* dst_ptrs, const long* numel, const int* chunks, int nchunks, void* stream);

}  // extern "C"

__global__ __launch_bounds__(256) void mt_cast_bf16_kernel(const float* const* __restrict__ src, unsigned short* const* __restrict__ dst,
                                                           const long* __restrict__ numel, const int* __restrict__ chunks) {
  const int t = chunks[2 * blockIdx.x], c = chunks[2 * blockIdx.x + 1];
  const float* s = src[t];
  __bf16* d = reinterpret_cast<__bf16*>(dst[t]);
  const long n = numel[t], beg = (long)c * MT_CHUNK;
  const bool vec = ((((uintptr_t)s) | (((uintptr_t)d) << 1)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long i = beg + (long)(it * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 3 < n) st4(d + i, *reinterpret_cast<const float4*>(s + i));
    else
      for (int e = 0; e < 4 && i + e < n; ++e) d[i + e] = (__bf16)s[i + e];
  }
}

extern "C" int lotus_shadow_cast(const void* src_ptrs, const void* dst_ptrs, const long* numel, const int* chunks, int nchunks, void* stream) {
  LOTUS_CHECK_ARG(src_ptrs && dst_ptrs && numel && chunks && nchunks >= 0, "lotus_shadow_cast: bad arguments");
  if (nchunks == 0) return LOTUS_OK;
  LOTUS_LAUNCH(mt_cast_bf16_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const float* const*)src_ptrs,
               (unsigned short* const*)dst_ptrs, numel, chunks);
  LOTUS_LAUNCH_CHECK("lotus_shadow_cast");
  return LOTUS_OK;
}
