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

// Four consecutive operand elements AS LOADED.  The staging registers of the next slab must not be touched between the
// load and the LDS store behind the MFMAs: converting at the load site (what ld4 does for bf16 storage: two shifts and
// two ands per 8 bytes) makes the compiler wait for the load right there, and the global latency of every slab lands in
// front of its products (seen in the ISA of the bf16-storage build: s_waitcnt vmcnt directly behind the loads).
template <typename T> struct Raw4 { float4 v; };
template <> struct Raw4<__bf16> { uint2 v; };
__device__ __forceinline__ Raw4<float> ldraw(const float* p) { return {*reinterpret_cast<const float4*>(p)}; }
__device__ __forceinline__ Raw4<__bf16> ldraw(const __bf16* p) { return {*reinterpret_cast<const uint2*>(p)}; }
__device__ __forceinline__ float4 unraw(const Raw4<float>& r) { return r.v; }
__device__ __forceinline__ float4 unraw(const Raw4<__bf16>& r) {
  return make_float4(__builtin_bit_cast(float, r.v.x << 16), __builtin_bit_cast(float, r.v.x & 0xffff0000u),
                     __builtin_bit_cast(float, r.v.y << 16), __builtin_bit_cast(float, r.v.y & 0xffff0000u));
}
__device__ __forceinline__ Raw4<float> toraw(float4 v, const float*) { return {v}; }
__device__ __forceinline__ Raw4<__bf16> toraw(float4 v, const __bf16*) {  // (guarded loads of odd shapes: values are exact bf16)
  return {make_uint2(lotus_pack_bf16(v.x, v.y), lotus_pack_bf16(v.z, v.w))};
}
__device__ __forceinline__ Raw4<float> rsel(bool ok, Raw4<float> r) {
  return {make_float4(ok ? r.v.x : 0.f, ok ? r.v.y : 0.f, ok ? r.v.z : 0.f, ok ? r.v.w : 0.f)};
}
__device__ __forceinline__ Raw4<__bf16> rsel(bool ok, Raw4<__bf16> r) { return {make_uint2(ok ? r.v.x : 0u, ok ? r.v.y : 0u)}; }

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

// ---- bf16 operand path (PREC 1: plain bf16 operands, fp32 accumulate — the bf16 compute mode of BASELINE configs[4];
// PREC 3: "bf16x3" split, a = hi + lo with hi = bf16(a), lo = bf16(a - hi), products hi*hi + hi*lo + lo*hi accumulated
// in fp32: ~2^-17 relative error per product at 3/16 of the fp32-MFMA issue time).  v_mfma_f32_32x32x16_bf16: lane
// l holds 8 consecutive k (k = 8 * (l >> 5) + 0..7) of row / column l & 31; C/D layout as the fp32 32x32 form.
// LDS images are k-contiguous bf16 rows for BOTH global layouts: [R][BK] bf16, row stride BK * 2 + 16 bytes (the 16
// lanes of a ds_read_b128 group land on distinct 16-byte slots), PREC 3 keeps a second plane with the lo parts.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;

template <int R, int BK, int PREC>
struct BTile {
  static constexpr int kLdW = BK / 2 + 4;   // row stride in 32-bit words
  static constexpr int kPlaneW = R * kLdW;  // words per plane
  static constexpr int kWords = (PREC == 3 ? 2 : 1) * kPlaneW;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// (hi, lo) words of the value pair (a, b); lo = bf16(x - float(hi))
__device__ __forceinline__ void split_bf16(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf16(a, b);
  lo = pack_bf16(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}

// w[e] <- w[(e + rot) & 3] with value selects (a dynamically indexed register array would go through scratch)
__device__ __forceinline__ void rot4(unsigned (&w)[4], int rot) {
  const bool r1 = rot & 1, r2 = rot & 2;
  const unsigned a0 = r1 ? w[1] : w[0], a1 = r1 ? w[2] : w[1], a2 = r1 ? w[3] : w[2], a3 = r1 ? w[0] : w[3];
  w[0] = r2 ? a2 : a0; w[1] = r2 ? a3 : a1; w[2] = r2 ? a0 : a2; w[3] = r2 ? a1 : a3;
}

template <int BM, int BN, int BK, int PREC>
__device__ __forceinline__ void gemm_slab_bf16(const unsigned* __restrict__ Aw, const unsigned* __restrict__ Bw, int wr0,
                                               int wc0, f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64;
  using TA = BTile<BM, BK, PREC>;
  using TB = BTile<BN, BK, PREC>;
  const int l31 = threadIdx.x & 31, h = (threadIdx.x >> 5) & 1;
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    uint4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const unsigned* q = Aw + (wr0 + tm * 32 + l31) * TA::kLdW + ks * 8 + h * 4;
      ah[tm] = *reinterpret_cast<const uint4*>(q);
      if (PREC == 3) al[tm] = *reinterpret_cast<const uint4*>(q + TA::kPlaneW);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const unsigned* q = Bw + (wc0 + tn * 32 + l31) * TB::kLdW + ks * 8 + h * 4;
      bh[tn] = *reinterpret_cast<const uint4*>(q);
      if (PREC == 3) bl[tn] = *reinterpret_cast<const uint4*>(q + TB::kPlaneW);
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if (PREC == 3) {  // small terms first
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[tm]), __builtin_bit_cast(bf16x8, bh[tn]), acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[tm]), __builtin_bit_cast(bf16x8, bl[tn]), acc[tm][tn], 0, 0, 0);
        }
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[tm]), __builtin_bit_cast(bf16x8, bh[tn]), acc[tm][tn], 0, 0, 0);
      }
  }
}
