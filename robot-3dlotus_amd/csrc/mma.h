// lotus-hip: fp32 MFMA block-tile machinery shared by the dense GEMMs and the sparse-conv
// gather-GEMMs.  v_mfma_f32_32x32x2_f32 is bit-equal to an fmaf chain (exact f32), which is what
// the 1e-4 logit parity bar needs; peak 157.3 TFLOP/s on MI355X.
//
// Block = 256 threads = 4 waves arranged 2 x 2; wave tile = (BM/2) x (BN/2) made of 32x32 MFMA
// tiles; K is consumed in slabs of BK = 16 staged through LDS.
//
// Operand fragment layout of mfma_f32_32x32x2f32 (MI355X guide §3):
//   A: lane l holds A[i = l & 31][k = l >> 5]      B: lane l holds B[k = l >> 5][j = l & 31]
//   C/D: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) for register r in [0, 16)
#pragma once
#include "common.h"

#define LOTUS_BK 16

// LDS images.  "KC" = the operand is k-contiguous in global memory (rows of K): image [R][BK+1]
// (odd stride -> conflict-free ds_read_b32 across 32 rows).  Otherwise the operand is contiguous
// along its non-reduction index: image [BK][R] (lanes read consecutive floats).
template <int R, bool KC, int BK = LOTUS_BK>
struct LdsTile {
  static constexpr int kStride = KC ? (BK + 1) : R;
  static constexpr int kFloats = KC ? R * (BK + 1) : BK * R;
  __device__ static __forceinline__ int idx(int r, int k) { return KC ? r * kStride + k : k * kStride + r; }
};

// Guarded 4-element global load of base[r * ld + c .. c + 3] (fp32 or bf16 storage).
template <typename T>
__device__ __forceinline__ float4 load4_guard(const T* __restrict__ base, long ld, int r, int c, int rmax,
                                              int cmax, bool vec_ok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < rmax && c < cmax) {
    const T* p = base + (long)r * ld + c;
    if (vec_ok && c + 3 < cmax) {
      v = ld4(p);
    } else {
      v.x = ld1(p);
      if (c + 1 < cmax) v.y = ld1(p + 1);
      if (c + 2 < cmax) v.z = ld1(p + 2);
      if (c + 3 < cmax) v.w = ld1(p + 3);
    }
  }
  return v;
}

// One BK slab of MFMAs for this wave.  acc[tm][tn] += A(32 rows) x B(32 cols).
template <int BM, int BN, bool A_KC, bool B_KC, bool SUM_A, int BK = LOTUS_BK>
__device__ __forceinline__ void mma_slab(const float* __restrict__ As, const float* __restrict__ Bs, int wr0,
                                         int wc0, f32x16 (&acc)[BM / 64][BN / 64], float (&asum)[BM / 64],
                                         unsigned tm_mask = 0xffffffffu) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int l31 = threadIdx.x & 31, h = (threadIdx.x >> 5) & 1;
#pragma unroll
  for (int kk = 0; kk < BK; kk += 2) {
    float a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a[tm] = As[LdsTile<BM, A_KC, BK>::idx(wr0 + tm * 32 + l31, kk + h)];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[LdsTile<BN, B_KC, BK>::idx(wc0 + tn * 32 + l31, kk + h)];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (SUM_A) asum[tm] += a[tm];
      if (tm_mask & (1u << tm)) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    }
  }
}

// Row / column of accumulator register r of tile (tm, tn) for this lane.
__device__ __forceinline__ int acc_row(int wr0, int tm, int r) {
  return wr0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * ((threadIdx.x >> 5) & 1);
}
__device__ __forceinline__ int acc_col(int wc0, int tn) { return wc0 + tn * 32 + (threadIdx.x & 31); }
