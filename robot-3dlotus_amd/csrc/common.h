// lotus-hip: shared device/host helpers (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// Every kernel of the library is launched through LOTUS_LAUNCH.  Normally that is hipLaunchKernelGGL; while a composite
// entry point (blocks.cpp) has armed `lotus_tls_stop_event`, the launch carries that event as its completion ("stop")
// event instead, so another stream can be ordered after THIS kernel without an event-record marker packet in the
// launching queue (a marker between two dependent kernels of the critical stream costs it 5-10 us; measured).
// (`__thread`, not `thread_local`: a C++ thread_local extern is reached through a weak init-function symbol, and a HIDDEN weak
// undefined symbol in position-independent code resolves to the load address instead of null — the call crashes)
extern __attribute__((visibility("hidden"))) __thread hipEvent_t lotus_tls_stop_event;  // library-internal: not an exported symbol
#ifdef LOTUS_EXP_SKIP_PROBE  // measurement builds only (tools/dbg/skip_probe.sh): LOTUS_EXP_SKIP=substr[,substr...] drops the launches
#include <stdlib.h>       // of the named kernels -> the step without that family = an upper bound of what speeding it up can give
static inline int lotus_exp_skip(const char* name) {
  const char* e = getenv("LOTUS_EXP_SKIP");
  if (!e) return 0;
  char buf[512]; strncpy(buf, e, 511); buf[511] = 0;
  for (char* t = strtok(buf, ","); t; t = strtok(nullptr, ",")) if (strstr(name, t)) return 1;
  return 0;
}
#define LOTUS_EXP_SKIP_CHECK(kernel) static int skip_ = -1; if (skip_ < 0) skip_ = lotus_exp_skip(#kernel); if (skip_) break;
#else
#define LOTUS_EXP_SKIP_CHECK(kernel)
#endif
#define LOTUS_LAUNCH(kernel, grid, block, lds, stream, ...)                                                              \
  do {                                                                                                                   \
    LOTUS_EXP_SKIP_CHECK(kernel)                                                                                         \
    if (lotus_tls_stop_event)                                                                                            \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, lotus_tls_stop_event, 0, __VA_ARGS__);            \
    else                                                                                                                 \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                                 \
  } while (0)
// An entry point with several launches hands an armed stop event to its LAST launch only (a completion event costs the
// launching queue ~4 us per kernel that carries it): hold it on entry, call last() right before the final launch.
struct StopEventOnLast {
  hipEvent_t ev;
  StopEventOnLast() : ev(lotus_tls_stop_event) { lotus_tls_stop_event = nullptr; }
  void last() { lotus_tls_stop_event = ev; }
  ~StopEventOnLast() { lotus_tls_stop_event = ev; }  // (the owner of the event disarms it)
};

// The small passes of the critical stream — LayerNorm / BatchNorm statistics and apply, the tap sum of the convolution, pooling
// and unpooling: vector-ALU and memory work, no MFMA — raise their waves' issue priority (s_setprio 3).  In the backward pass they
// share every SIMD with the weight-gradient stream's MFMA waves, and an fp32 MFMA leaves no issue slot to a vector-ALU
// instruction of another wave (tools/ubench/mfma_coissue.hip): at the default priority a 10 us normalisation pass takes 25-145 us
// in the step.  Measured (round 6, three alternating series, tools/dbg/ab_small_prio*.sh): +0.1 ... +0.5 % of the plain step,
// +0.7 % of the one-rank RCCL rehearsal (median of 15 fresh processes each).  The same priority on the MFMA kernels of the
// critical stream (dense, attention, tap convolution) measured 0 ... -0.3 %: they stay at the default.
#define LOTUS_T_PRIO() __builtin_amdgcn_s_setprio(3)

#define LOTUS_OK 0
#define LOTUS_E_ARG (-1)
#define LOTUS_E_LAUNCH (-2)
#define LOTUS_E_UNSUPPORTED (-3)
#define LOTUS_E_WORKSPACE (-4)

void lotus_set_error(const char* fmt, ...);

#define LOTUS_CHECK_ARG(cond, ...)          \
  do {                                      \
    if (!(cond)) {                          \
      lotus_set_error(__VA_ARGS__);         \
      return LOTUS_E_ARG;                   \
    }                                       \
  } while (0)

#define LOTUS_LAUNCH_CHECK(name)                                                   \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      lotus_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return LOTUS_E_LAUNCH;                                                       \
    }                                                                              \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activation storage type.  Every [rows][channels] ACTIVATION tensor in HBM (layer inputs / outputs, saved
// pre-activations, their gradients) is `act_t`; parameters, parameter gradients, statistics, logits of the losses'
// reductions and all accumulation stay fp32.  The library is compiled twice from the same sources: act_t = float (entry
// points lotus_*, the fp32 parity path) and, with -DLOTUS_ACT_BF16, act_t = bf16 (entry points lotus_b16_*: the
// "bf16 activations / fp32 master weights / fp32 accumulate" mode of BASELINE configs[4]).  Kernels never spell the
// width of an activation access: they go through ld4 / st4 (four consecutive channels: 16 bytes fp32, 8 bytes bf16,
// round-to-nearest-even on store) and ld1 / st1, which are overloaded on the pointer type, so the fp32 instantiation is
// token-for-token the code it was before.  All file-scope code of a translation unit lives in namespace LOTUS_NS so
// that the two variants of a kernel template never share a symbol.
#ifdef LOTUS_ACT_BF16
typedef __bf16 act_t;
#define LOTUS_NS lotus_b16
#define LOTUS_ACT_IS_BF16 1
#else
typedef float act_t;
#define LOTUS_NS lotus_f32
#define LOTUS_ACT_IS_BF16 0
#endif
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 lotus_bf16x2;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                     __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned lotus_pack_bf16(float a, float b) {
  const lotus_bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(lotus_pack_bf16(v.x, v.y), lotus_pack_bf16(v.z, v.w));
}
// group q of four consecutive elements
template <typename T> __device__ __forceinline__ float4 ld4q(const T* p, long q) { return ld4(p + 4 * q); }
template <typename T> __device__ __forceinline__ void st4q(T* p, long q, float4 v) { st4(p + 4 * q, v); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
// floats needed to hold n activation elements (host-side carving of flat float buffers)
static inline size_t lotus_act_floats(size_t n) { return LOTUS_ACT_IS_BF16 ? (n + 1) / 2 : n; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// activation codes shared by the C-ABI
#define LOTUS_ACT_NONE 0
#define LOTUS_ACT_GELU 1
#define LOTUS_ACT_LEAKY 2  // LeakyReLU(0.02), simple_policy_ptv3.py:42

// GELU (exact erf form, nn.GELU default) with ONE short branch-free evaluation of the normal distribution function.
// On gfx950 every vector-ALU instruction of an fp32-MFMA kernel is time the MFMA pipe stands still (DESIGN.md section 4,
// round 5), and libm's erff costs ~35 of them per element (two polynomial branches under exec masks + an exp with error
// compensation): GELU on the [65536, 512] hidden layer was as expensive as the product that feeds it.  Here
//   erfc(t) = 2^-q(t),  q = degree-8 minimax fit of -log2 erfc on [0, 4] weighted by erfc (|error of 2^-q| <= 1.1e-7,
//   tools/fit_erfc.py), erfc(t > 4) < 1.6e-8 by extrapolation, 0 from t = 8;   2 Phi(x) = erfc(-x / sqrt 2) = x >= 0 ? 2 - e : e
// costs 8 fma + v_exp_f32 + 5: |gelu - exact| <= 1.2e-7 max(1, |gelu|), |gelu' - exact| <= 1.3e-7 (float64 reference,
// 4 M points in [-12, 12]); negative arguments have no cancellation (erfc is formed directly).
__device__ __forceinline__ float lotus_two_phi(float x) {  // 2 Phi(x)
  // (beyond the fit range the polynomial keeps growing — q(5) = 40, q(6) = 65, q(8) = 270 — so 2^-q runs smoothly into 0: a
  //  clamp at 4 held erfc at 1.5e-8 and made gelu(x << 0) = 0.75e-8 x instead of 0; NaN inputs still give a finite cdf)
  const float t = fminf(fabsf(x) * 0.70710678118654752440f, 8.0f);
  float q = 4.435278970e-05f;
  q = fmaf(q, t, -4.369438975e-04f);
  q = fmaf(q, t, 1.460380852e-03f);
  q = fmaf(q, t, 8.251661202e-04f);
  q = fmaf(q, t, -2.830188721e-02f);
  q = fmaf(q, t, 1.485066414e-01f);
  q = fmaf(q, t, 9.184098244e-01f);
  q = fmaf(q, t, 1.627909303e+00f);
  q = fmaf(q, t, -2.171762503e-08f);
  const float e = __builtin_amdgcn_exp2f(-q);  // v_exp_f32: erfc(t)
  return x >= 0.f ? 2.0f - e : e;
}
__device__ __forceinline__ float gelu_f(float x) { return (0.5f * x) * lotus_two_phi(x); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2 / 2) / sqrt(2 pi)
  return fmaf(x, pdf, 0.5f * lotus_two_phi(x));
}
__device__ __forceinline__ float act_f(float x, int act) {
  if (act == LOTUS_ACT_GELU) return gelu_f(x);
  if (act == LOTUS_ACT_LEAKY) return x > 0.f ? x : 0.02f * x;
  return x;
}
__device__ __forceinline__ float act_grad_f(float pre, int act) {
  if (act == LOTUS_ACT_GELU) return gelu_grad_f(pre);
  if (act == LOTUS_ACT_LEAKY) return pre > 0.f ? 1.f : 0.02f;
  return 1.f;
}

// Counter-based dropout mask: keep iff hash(seed, idx) >= p * 2^32.  Stateless so the backward
// pass regenerates the identical mask from (seed, element index).
__device__ __forceinline__ uint32_t lotus_hash32(uint64_t seed, uint64_t idx) {
  // murmur3-style 32-bit finaliser over (index, seed): 3 multiplies, ~8 VALU ops (a 64-bit splitmix costs ~25)
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  x ^= (uint32_t)(idx >> 32) * 0x7FEB352Du + (uint32_t)(seed >> 32);
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// One 32-bit hash decides TWO consecutive elements (its halves against the upper 16 bits of the threshold: the drop
// probability is quantised to 2^-16): the hash is three integer multiplies (quarter rate) + five logic operations of
// vector ALU, which an fp32-MFMA kernel pays in MFMA time (round 5).
// Host side of the pair hash: the threshold and the scale of the probability the kernels actually apply, t16 / 65536 (at least
// 2^-16 for p > 0) — with 1 / (1 - p) of the unquantised p the expectation was off by up to 2^-16 / (1 - p), and p < 2^-16
// dropped nothing yet scaled.  The attention kernels compare one full 32-bit hash per element and keep 1 / (1 - p).
static inline void lotus_drop_setup(float p, unsigned* thresh, float* inv_keep) {
  if (!(p > 0.f)) { *thresh = 0; *inv_keep = 1.f; return; }
  unsigned t16 = (unsigned)((double)p * 65536.0);
  if (t16 == 0) t16 = 1;
  if (t16 > 65535) t16 = 65535;
  *thresh = t16 << 16;
  *inv_keep = (float)(1.0 / (1.0 - (double)t16 / 65536.0));
}
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
  const uint32_t hv = lotus_hash32(seed, idx >> 1);
  return ((idx & 1) ? (hv >> 16) : (hv & 0xffffu)) >= (thresh >> 16) ? inv_keep : 0.f;
}
// the scales of elements idx4 .. idx4 + 3, idx4 a multiple of 4: two hashes
__device__ __forceinline__ void dropout_scale4(uint64_t seed, uint64_t idx4, uint32_t thresh, float inv_keep, float (&m)[4]) {
  const uint32_t h0 = lotus_hash32(seed, idx4 >> 1), h1 = lotus_hash32(seed, (idx4 >> 1) + 1), t16 = thresh >> 16;
  m[0] = (h0 & 0xffffu) >= t16 ? inv_keep : 0.f;
  m[1] = (h0 >> 16) >= t16 ? inv_keep : 0.f;
  m[2] = (h1 & 0xffffu) >= t16 ? inv_keep : 0.f;
  m[3] = (h1 >> 16) >= t16 ? inv_keep : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
