// lotus-hip: what every dense-GEMM kernel family of gemm.hip shares — the launch parameter block, the fused epilogue of one
// float4 of the output and the hand-off of fused split-K partial tiles (gemm_kernel, gemm_dma_kernel).
#pragma once
#include "mma.h"

namespace LOTUS_NS {

struct GemmP {
  // element types are template parameters of the kernel (EA, EB, EC): activations are act_t, weights / weight gradients /
  // split-K partials are float (fwd, dgrad: act x float -> act; wgrad: act x act -> float)
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  long lda, ldb, ldc;
  const float* bias;      // [N]
  const act_t* residual;  // [M][ldc], added after the activation
  act_t* pre;             // [M][ldc], pre-activation (after bias) saved for backward
  const act_t* mulpre;    // [M][ldc], dgrad: multiply by act'(mulpre)
  int act;   // activation applied to the value
  int dact;  // derivative code used with mulpre
  int klen;          // K range per blockIdx.z (multiple of BK)
  long part_stride;  // C += z * part_stride when gridDim.z > 1
  float* bias_part;  // wgrad: column sums of dY, slice z at bias_part + z * bias_stride
  long bias_stride;
  int a_vec, b_vec;  // 16-byte vector loads allowed
  // output dropout (applied after act, before residual): keep iff hash >= thresh
  unsigned long long drop_seed;
  unsigned drop_thresh;
  float drop_inv_keep;
  int prec;  // operand precision of this call: 0 fp32 MFMA (exact), 1 bf16, 3 bf16x3 split
  // fused split-K (gridDim.z > 1 and cnt != null): every block stores its raw partial tile to part + z * part_stride,
  // the LAST block to arrive at a tile (per-tile counter) sums the nz partials in fixed z order and applies the epilogue
  // (deterministic: the order does not depend on which block is last); counters are left at zero
  float* part;
  unsigned* cnt;
  float* bias_out;   // wgrad, fused: final column sums (bias gradient)
  int accumulate;    // wgrad, fused: C / bias_out += result
  int c_float;       // fwd / dgrad: C is an fp32 split-K partial slab, not an activation tensor
  int b_act;         // fwd / dgrad, bf16-storage build: the weight matrix B is a bf16 SHADOW of the fp32 master (precision | 4)
  // tap-grouped products (sparse convolution of the deep levels as 27 gathered GEMMs in one launch, conv.hip): A rows are
  // gathered through a_rows; the M axis is 27 segments of tap_rows (a multiple of 64) rows, segment t holds tap_cnt[t] pairs
  // (row tiles past them leave at once) and multiplies the weight slice B + (mirror ? 26 - t : t) * b_tap_stride
  const int* a_rows;
  const int* tap_cnt;
  int tap_rows, b_tap_mirror;
  long b_tap_stride;
  int a_src_rows;  // rows of the gathered tensor A (bounds of the LDS-DMA descriptor; gemm_dma_tap_kernel)
  // LayerNorm backward fused into an input-gradient product whose block tile covers whole rows (gemm_dma_kernel, EPI 2): the
  // product is dy of the LayerNorm output; C receives dx = LN'(dy) (+ residual), ln_dz (optional) dx times the dropout mask of
  // drop_seed / drop_thresh, ln_part [row tiles][2][N] the column partials of dgamma / dbeta
  const act_t* ln_x;
  const float* ln_mean;
  const float* ln_rstd;
  const float* ln_gamma;
  float* ln_part;
  act_t* ln_dz;
};

// Partial tiles of a fused split-K product travel between blocks that may sit on different XCDs (one L2 each).  An
// agent-scope fence would write back / invalidate the whole L2 of the issuing XCD (measured: ~100 us per launch), so
// the partials themselves are moved with agent-scope relaxed atomics — write-through stores, L2-bypassing loads — and
// only workgroup-scope fences (s_waitcnt) order them against the arrival counter.
//
// HARDWARE CONTRACT (gfx942 / gfx950 only): this is the "sc1 stores AND sc1 loads on both sides" hand-off of
// MI355X_MICROARCH.md (Workgroup dispatch ... valid forms): a relaxed agent-scope atomic store lowers to `global_store ...
// sc1` (write-through: the bytes have left the XCD's L2 once vmcnt drains), a relaxed agent-scope atomic load to
// `global_load ... sc1` (served by memory, never by a stale L1 / remote-L2 line); an explicit `s_waitcnt vmcnt(0)` (inline
// asm: the compiler neither drops nor moves it) in every thread followed by `__syncthreads()` drains every partial store
// before lane 0 bumps the counter (tests/test_capi.py checks the disassembly for that wait).  It is NOT the HSA memory model's
// agent-scope release / acquire and is not portable to other targets or guaranteed against compiler changes.  Guards:
// the library is built for gfx950 only; LOTUS_SPLITK_FUSED=0 switches every split-K product to the two-launch path
// (partials, then a reduction kernel — ordinary kernel-boundary visibility); tests/test_gpu_ops.py compares the two paths
// and tests/test_gpu_fullsize_properties.py runs the fused one at the bench size against the two-launch result.
__device__ __forceinline__ void st_agent4(float* p, float4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  const unsigned long long lo = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
  const unsigned long long hi = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
  __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_agent4(const float* p) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<float*>(p));
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                     __uint_as_float((unsigned)(hi >> 32)));
}

// the full epilogue of one float4 of the output: bias, pre-activation copy, activation, act', dropout, residual
template <typename TC>
__device__ __forceinline__ void gemm_epilogue4(const GemmP& p, TC* __restrict__ C, long o, int col, float (&v)[4]) {
  if (p.bias) {
    const float4 bv = ld4(p.bias + col);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  if (p.pre) st4(p.pre + o, make_float4(v[0], v[1], v[2], v[3]));
  if (p.act != LOTUS_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.act);
  }
  if (p.mulpre) {
    const float4 m4 = ld4(p.mulpre + o);
    const float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= act_grad_f(mv[e], p.dact);
  }
  if (p.drop_thresh) {
    float dm[4];
    dropout_scale4(p.drop_seed, (unsigned long long)o, p.drop_thresh, p.drop_inv_keep, dm);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= dm[e];
  }
  if (p.residual) {
    const float4 r4 = ld4(p.residual + o);
    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
  }
  if (p.accumulate) {
    const float4 c4 = ld4(C + o);
    v[0] += c4.x; v[1] += c4.y; v[2] += c4.z; v[3] += c4.w;
  }
  st4(C + o, make_float4(v[0], v[1], v[2], v[3]));
}

// The tail of a fused split-K block (after its raw partial tile and bias partial have been stored): release the partial
// (agent scope: the other splits of the tile may run on another XCD / L2), count the arrival, and let the LAST block of the
// tile sum the nz partials in fixed z order and apply the epilogue.  All reads come after the acquire fence.  Shared by
// gemm_kernel and wgrad_stream_kernel.
template <bool SUM_A, int BM, int BN, typename EC>
__device__ __forceinline__ void splitk_fused_tail(const GemmP& p, int bx, int by, int m0, int n0, int tid) {
  __shared__ int s_last;
  // every write-through (sc1) partial store of this thread has left the XCD's L2 before the arrival is counted: the explicit
  // vmcnt(0) is what orders them (the workgroup-scope fence alone compiled to lgkmcnt(0) + s_barrier, with no
  // vmcnt wait between the sc1 stores and the counter atomic; inline asm is invisible to the pass that drops such waits)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int tile_id = by * (int)gridDim.x + bx;
  if (tid == 0) s_last = __hip_atomic_fetch_add(&p.cnt[tile_id], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.z - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int nz = gridDim.z;
  for (int i = tid; i < BM * (BN / 4); i += 256) {
    const int row = m0 + i / (BN / 4), col = n0 + (i % (BN / 4)) * 4;
    if (row >= p.M || col >= p.N) continue;
    const long o = (long)row * p.ldc + col;
    const float* q = p.part + o;
    float4 s4 = ld_agent4(q);
    int z = 1;
    for (; z + 3 < nz; z += 4) {  // four independent loads in flight, summed in z order
      const float4 v0 = ld_agent4(q + (long)z * p.part_stride);
      const float4 v1 = ld_agent4(q + (long)(z + 1) * p.part_stride);
      const float4 v2 = ld_agent4(q + (long)(z + 2) * p.part_stride);
      const float4 v3 = ld_agent4(q + (long)(z + 3) * p.part_stride);
      s4.x = (((s4.x + v0.x) + v1.x) + v2.x) + v3.x; s4.y = (((s4.y + v0.y) + v1.y) + v2.y) + v3.y;
      s4.z = (((s4.z + v0.z) + v1.z) + v2.z) + v3.z; s4.w = (((s4.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; z < nz; ++z) {
      const float4 v = ld_agent4(q + (long)z * p.part_stride);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    float v[4] = {s4.x, s4.y, s4.z, s4.w};
    gemm_epilogue4(p, static_cast<EC*>(p.C), o, col, v);
  }
  if (SUM_A && p.bias_out && p.bias_part && bx == 0) {
    for (int i = tid; i < BM; i += 256) {
      if (m0 + i >= p.M) continue;
      float sb = 0.f;
      for (int z = 0; z < nz; ++z)
        sb += __hip_atomic_load(p.bias_part + (long)z * p.bias_stride + m0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      p.bias_out[m0 + i] = p.accumulate ? p.bias_out[m0 + i] + sb : sb;
    }
  }
  if (tid == 0) p.cnt[tile_id] = 0;  // ready for the next launch on this stream
}

// gemm_dma.hip: the LDS-DMA kernels for tall products (returns LOTUS_GEMM_DMA_NA when a product is not theirs)
#define LOTUS_GEMM_DMA_NA (-100)
int launch_gemm_dma(GemmP& p, int layout, int nz, hipStream_t st);
int gemm_dma_wgrad_splits(int M, int N, int K);
int launch_gemm_dma_tap(GemmP& p, int layout, hipStream_t st);

}  // namespace LOTUS_NS
