// lotus-hip: dense fp32 MFMA GEMMs with fused epilogues — the nn.Linear call sites of the
// 3D-LOTUS hot path (66 % of all MACs): qkv/proj/fc1/fc2/cpe.1/q/kv/down.proj/
// up.proj/heads, forward, dgrad and wgrad.
//
//   fwd   : Y[M,N]  = act(X[M,K] W[N,K]^T + b) (+ residual)      A k-contig, B k-contig
//   dgrad : dX[M,K] = (dY[M,N] W[N,K]) * act'(pre) (+ add)        A k-contig, B j-contig
//   wgrad : dW[N,K] = dY[M,N]^T X[M,K], db = colsum(dY)           A i-contig, B j-contig, split-K
//
// wgrad is deterministic: split-K partials go to a caller-provided workspace and are summed in a
// fixed order by a second kernel (no float atomics).
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace LOTUS_NS {


// FAST: operands are 16-byte aligned with ld % 4 == 0 and the contiguous extents are multiples of 4,
// so every global access is an unconditional float4 (rows / columns beyond the edge are clamped to a
// valid address; k beyond the split range is zeroed by a select).  !FAST keeps per-element guards
// (odd widths such as the 90- and 217-wide head layers).
// LDS images of the GEMM.  k-contiguous operands ("KC"): [R][BK + 4] — rows are 16-byte aligned, a lane's fragment run
// is read with ds_read_b128 (row stride 36/20/68 floats: the 16 lanes of a b128 group land on distinct 16-byte slots).
// Operands contiguous along their non-reduction index: [BK][R] (float4 row stores, ds_read_b32 fragment reads).
// Both operands consume k in the same permuted order: MFMA step s of lane-half h uses k = h * BK/2 + s, so the
// KC fragments are contiguous runs.
template <int R, bool KC, int BK>
struct GTile {
  static constexpr int kLd = KC ? BK + 4 : R;
  static constexpr int kFloats = KC ? R * (BK + 4) : BK * R;
};

template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool SUM_A>
__device__ __forceinline__ void gemm_slab(const float* __restrict__ As, const float* __restrict__ Bs, int wr0, int wc0,
                                          f32x16 (&acc)[BM / 64][BN / 64], float (&asum)[BM / 64]) {
  constexpr int TM = BM / 64, TN = BN / 64, KS = BK / 2;
  const int l31 = threadIdx.x & 31, h = (threadIdx.x >> 5) & 1;
  float a[TM][KS], b[TN][KS];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (A_KC) {
      const float* row = As + (wr0 + tm * 32 + l31) * GTile<BM, true, BK>::kLd + h * KS;
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) {
        const float4 v = ld4(row + 4 * q);
        a[tm][4 * q] = v.x; a[tm][4 * q + 1] = v.y; a[tm][4 * q + 2] = v.z; a[tm][4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) a[tm][s2] = As[(h * KS + s2) * BM + wr0 + tm * 32 + l31];
    }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    if (B_KC) {
      const float* row = Bs + (wc0 + tn * 32 + l31) * GTile<BN, true, BK>::kLd + h * KS;
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) {
        const float4 v = ld4(row + 4 * q);
        b[tn][4 * q] = v.x; b[tn][4 * q + 1] = v.y; b[tn][4 * q + 2] = v.z; b[tn][4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) b[tn][s2] = Bs[(h * KS + s2) * BN + wc0 + tn * 32 + l31];
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < KS; ++s2) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (SUM_A) asum[tm] += a[tm][s2];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s2], b[tn][s2], acc[tm][tn], 0, 0, 0);
    }
  }
}

// (the bf16 operand tiles — BTile, split_bf16, rot4, gemm_slab_bf16 — live in mma.h: the sparse-conv weight gradient shares them)

// (Raw4 / ldraw / unraw / toraw / rsel — operand elements as loaded — live in mma.h)
// value-wise select (a pointer select between the loaded vector and a zero constant goes through scratch)
__device__ __forceinline__ float4 zsel(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// RD = depth of the register staging ring: the global loads of slab i + RD are issued while slab i multiplies.  With bf16
// products a 32-deep slab is two MFMAs per wave (54 ns), so at RD = 1 a block is a CHAIN of K / 32 + 1 global-load round
// trips (7 us per 64 x 64 x 128 block, 14 us per launch at M = 65 536 = 2 rounds of resident blocks: what was measured);
// the ring keeps RD slabs in flight per block for 8 - 12 staging registers per extra slab.
template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool SUM_A, bool FAST, int PREC = 0, typename EA = float, typename EB = float,
          typename EC = float, int RD = 1, bool TAP = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  const EA* __restrict__ pA = static_cast<const EA*>(p.A);
  const EB* __restrict__ pB = static_cast<const EB*>(p.B);
  const int* __restrict__ a_rows = TAP ? p.a_rows : nullptr;  // (TAP: the tap-grouped convolution launches, their own kernel name in a trace)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A4 = BM * BK / 4 / 256, B4 = BN * BK / 4 / 256;  // float4 per thread
  static_assert(A4 >= 1 && B4 >= 1, "tile too small");
  static_assert(PREC == 0 || ((!SUM_A || !A_KC) && BK % 16 == 0 && (A_KC || A4 % 2 == 0) && (B_KC || B4 % 2 == 0)),
                "bf16 path: operands contiguous along their non-reduction index are staged as k pairs; the bias-gradient "
                "sums are taken from the fp32 staging registers of such an A operand");
  constexpr int AF = PREC ? BTile<BM, BK, PREC>::kWords : GTile<BM, A_KC, BK>::kFloats;
  constexpr int BF = PREC ? BTile<BN, BK, PREC>::kWords : GTile<BN, B_KC, BK>::kFloats;
  constexpr int WN = BN / 2, SLD = WN + 4;  // epilogue staging: per wave [32][WN + 4]
  constexpr int LDSF = (2 * AF + 2 * BF) > (4 * 32 * SLD) ? (2 * AF + 2 * BF) : (4 * 32 * SLD);
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  float* As = smem;
  float* Bs = smem + 2 * AF;

  const int tid = threadIdx.x, wave = tid >> 6;
  // XCD-aware tile order: workgroups go to the 8 XCDs round-robin by linear id, each XCD has its own L2.  Give every
  // XCD a contiguous run of the row-major tile list so that the column blocks sharing an A row tile (and the
  // neighbouring row tiles sharing B) hit the same L2 instead of fetching the operand once per XCD.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (gridDim.z > 1 && (gridDim.z & 7) == 0) {
    // split-K (weight gradients): the output tiles of ONE split read the same rows of both operands, so they must
    // share an L2.  Hardware id = (z * gy + y) * gx + x goes to XCD id % 8: give XCD c the splits z = c (mod 8) and walk
    // the tiles of a split on consecutive ids of that XCD.
    const int tiles = gridDim.x * gridDim.y;
    const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;           // slot-th block of this XCD
    bz = xcd + 8 * (slot / tiles);
    const int t = slot - (slot / tiles) * tiles;
    by = t / (int)gridDim.x;
    bx = t - by * (int)gridDim.x;
  } else {
    const int nbx = gridDim.x, total = nbx * gridDim.y;
    const int lin = by * nbx + bx, xcd = lin & 7, slot = lin >> 3;
    const int q = total >> 3, r = total & 7;
    const int t = xcd * q + min(xcd, r) + slot;
    by = t / nbx;
    bx = t - by * nbx;
  }
  const int m0 = by * BM, n0 = bx * BN;
  if constexpr (TAP) {  // (block-uniform) this row tile's tap: its pair count and its weight slice
    const int tap = m0 / p.tap_rows;
    if (m0 - tap * p.tap_rows >= p.tap_cnt[tap]) return;
    pB += (long)(p.b_tap_mirror ? 26 - tap : tap) * p.b_tap_stride;
  }
  auto arow = [&](int r) { r = min(r, p.M - 1); return TAP ? a_rows[r] : r; };
  const int kbeg = bz * p.klen;
  const int kend = min(p.K, kbeg + p.klen);
  const int wr0 = (wave >> 1) * (BM / 2), wc0 = (wave & 1) * (BN / 2);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float asum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) asum[i] = 0.f;

  float bsum[4] = {0.f, 0.f, 0.f, 0.f};  // bf16 path, SUM_A: this thread's partial row sums of A (exact fp32 inputs)
  Raw4<EA> rring_a[RD][A4];
  Raw4<EB> rring_b[RD][B4];
  auto gload = [&](Raw4<EA> (&rra)[A4], Raw4<EB> (&rrb)[B4], int k0) {
#pragma unroll
    for (int t = 0; t < A4; ++t) {
      const int f = tid + t * 256;
      if (A_KC) {
        const int row = f / (BK / 4), kq = f % (BK / 4);
        if (FAST) {
          const int k = k0 + kq * 4;
          rra[t] = rsel(k < kend, ldraw(pA + (long)arow(m0 + row) * p.lda + min(k, p.K - 4)));
        } else {
          rra[t] = toraw(load4_guard(pA, p.lda, m0 + row, k0 + kq * 4, p.M, kend, p.a_vec), pA);
        }
      } else {
        const int f2 = PREC ? tid + (t >> 1) * 256 : f;  // bf16 path: registers 2p, 2p+1 hold k, k+1 of one row quad
        const int kr = PREC ? (f2 / (BM / 4)) * 2 + (t & 1) : f / (BM / 4), iq = f2 % (BM / 4);
        if (FAST) {
          const int k = k0 + kr;
          rra[t] = rsel(k < kend, ldraw(pA + (long)min(k, p.K - 1) * p.lda + min(m0 + iq * 4, p.M - 4)));
        } else {
          rra[t] = toraw(load4_guard(pA, p.lda, k0 + kr, m0 + iq * 4, kend, p.M, p.a_vec), pA);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < B4; ++t) {
      const int f = tid + t * 256;
      if (B_KC) {
        const int row = f / (BK / 4), kq = f % (BK / 4);
        if (FAST) {
          const int k = k0 + kq * 4;
          rrb[t] = rsel(k < kend, ldraw(pB + (long)min(n0 + row, p.N - 1) * p.ldb + min(k, p.K - 4)));
        } else {
          rrb[t] = toraw(load4_guard(pB, p.ldb, n0 + row, k0 + kq * 4, p.N, kend, p.b_vec), pB);
        }
      } else {
        const int f2 = PREC ? tid + (t >> 1) * 256 : f;
        const int kr = PREC ? (f2 / (BN / 4)) * 2 + (t & 1) : f / (BN / 4), jq = f2 % (BN / 4);
        if (FAST) {
          const int k = k0 + kr;
          rrb[t] = rsel(k < kend, ldraw(pB + (long)min(k, p.K - 1) * p.ldb + min(n0 + jq * 4, p.N - 4)));
        } else {
          rrb[t] = toraw(load4_guard(pB, p.ldb, k0 + kr, n0 + jq * 4, kend, p.N, p.b_vec), pB);
        }
      }
    }
  };
  auto lstore = [&](const Raw4<EA> (&rra)[A4], const Raw4<EB> (&rrb)[B4], float* Ad, float* Bd) {
    float4 ra[A4], rb[B4];  // the staged elements as fp32, converted HERE (behind the products of the running slab)
#pragma unroll
    for (int t = 0; t < A4; ++t) ra[t] = unraw(rra[t]);
#pragma unroll
    for (int t = 0; t < B4; ++t) rb[t] = unraw(rrb[t]);
    if (PREC) {  // convert (and split) while staging: bf16 rows, k-contiguous for both layouts
      unsigned* Aw = reinterpret_cast<unsigned*>(Ad);
      unsigned* Bw = reinterpret_cast<unsigned*>(Bd);
      using TA = BTile<BM, BK, PREC ? PREC : 1>;
      using TB = BTile<BN, BK, PREC ? PREC : 1>;
      if (A_KC) {
#pragma unroll
        for (int t = 0; t < A4; ++t) {
          const int f = tid + t * 256, row = f / (BK / 4), kq = f % (BK / 4);
          unsigned h0, l0, h1, l1;
          split_bf16(ra[t].x, ra[t].y, h0, l0);
          split_bf16(ra[t].z, ra[t].w, h1, l1);
          *reinterpret_cast<uint2*>(&Aw[row * TA::kLdW + kq * 2]) = make_uint2(h0, h1);
          if (PREC == 3) *reinterpret_cast<uint2*>(&Aw[TA::kPlaneW + row * TA::kLdW + kq * 2]) = make_uint2(l0, l1);
        }
      } else {
#pragma unroll
        for (int t = 0; t + 1 < A4; t += 2) {
          const int f2 = tid + (t >> 1) * 256, kr2 = f2 / (BM / 4), iq = f2 % (BM / 4);
          const float va[4] = {ra[t].x, ra[t].y, ra[t].z, ra[t].w}, vb[4] = {ra[t + 1].x, ra[t + 1].y, ra[t + 1].z, ra[t + 1].w};
          unsigned hw[4], lw[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split_bf16(va[e], vb[e], hw[e], lw[e]);
          if (SUM_A && bx == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[e] += va[e] + vb[e];
          }
          // the four rows of a quad are written in a per-lane rotated order so that the 64 lanes of one store
          // instruction hit 64 (BM = 64) / 32 (BM = 128) distinct banks instead of 16 / 8
          const int rot = (iq >> 2) & 3;
          rot4(hw, rot);
          if (PREC == 3) rot4(lw, rot);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = iq * 4 + ((e + rot) & 3);
            Aw[row * TA::kLdW + kr2] = hw[e];
            if (PREC == 3) Aw[TA::kPlaneW + row * TA::kLdW + kr2] = lw[e];
          }
        }
      }
      if (B_KC) {
#pragma unroll
        for (int t = 0; t < B4; ++t) {
          const int f = tid + t * 256, row = f / (BK / 4), kq = f % (BK / 4);
          unsigned h0, l0, h1, l1;
          split_bf16(rb[t].x, rb[t].y, h0, l0);
          split_bf16(rb[t].z, rb[t].w, h1, l1);
          *reinterpret_cast<uint2*>(&Bw[row * TB::kLdW + kq * 2]) = make_uint2(h0, h1);
          if (PREC == 3) *reinterpret_cast<uint2*>(&Bw[TB::kPlaneW + row * TB::kLdW + kq * 2]) = make_uint2(l0, l1);
        }
      } else {
#pragma unroll
        for (int t = 0; t + 1 < B4; t += 2) {
          const int f2 = tid + (t >> 1) * 256, kr2 = f2 / (BN / 4), jq = f2 % (BN / 4);
          const float va[4] = {rb[t].x, rb[t].y, rb[t].z, rb[t].w}, vb[4] = {rb[t + 1].x, rb[t + 1].y, rb[t + 1].z, rb[t + 1].w};
          unsigned hw[4], lw[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split_bf16(va[e], vb[e], hw[e], lw[e]);
          const int rot = (jq >> 2) & 3;
          rot4(hw, rot);
          if (PREC == 3) rot4(lw, rot);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = jq * 4 + ((e + rot) & 3);
            Bw[row * TB::kLdW + kr2] = hw[e];
            if (PREC == 3) Bw[TB::kPlaneW + row * TB::kLdW + kr2] = lw[e];
          }
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < A4; ++t) {
      const int f = tid + t * 256;
      const float v[4] = {ra[t].x, ra[t].y, ra[t].z, ra[t].w};
      if (A_KC) {
        const int row = f / (BK / 4), kq = f % (BK / 4);
        st4(&Ad[row * GTile<BM, true, BK>::kLd + kq * 4], ra[t]);
      } else {
        const int kr = f / (BM / 4), iq = f % (BM / 4);
        st4(&Ad[kr * BM + iq * 4], ra[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < B4; ++t) {
      const int f = tid + t * 256;
      const float v[4] = {rb[t].x, rb[t].y, rb[t].z, rb[t].w};
      if (B_KC) {
        const int row = f / (BK / 4), kq = f % (BK / 4);
        st4(&Bd[row * GTile<BN, true, BK>::kLd + kq * 4], rb[t]);
      } else {
        const int kr = f / (BN / 4), jq = f % (BN / 4);
        st4(&Bd[kr * BN + jq * 4], rb[t]);
      }
    }
  };

  // Unguarded prefetch of a slab that lies fully inside [kbeg, kend): per-thread base pointers (rows / columns clamped
  // once) plus the slab offset — one 64-bit add per 16-byte load, no k clamp, no zero select.  Issued at the top of the
  // iteration and pinned there (sched_barrier), so the loads fly under the whole slab of MFMAs instead of being
  // sunk next to their consumers.
  const EA* pa[A4];
  const EB* pb[B4];
  if (FAST) {
#pragma unroll
    for (int t = 0; t < A4; ++t) {
      const int f = tid + t * 256;
      if (A_KC) pa[t] = pA + (long)arow(m0 + f / (BK / 4)) * p.lda + (f % (BK / 4)) * 4;
      else if (PREC) pa[t] = pA + (long)(((tid + (t >> 1) * 256) / (BM / 4)) * 2 + (t & 1)) * p.lda + min(m0 + ((tid + (t >> 1) * 256) % (BM / 4)) * 4, p.M - 4);
      else pa[t] = pA + (long)(f / (BM / 4)) * p.lda + min(m0 + (f % (BM / 4)) * 4, p.M - 4);
    }
#pragma unroll
    for (int t = 0; t < B4; ++t) {
      const int f = tid + t * 256;
      if (B_KC) pb[t] = pB + (long)min(n0 + f / (BK / 4), p.N - 1) * p.ldb + (f % (BK / 4)) * 4;
      else if (PREC) pb[t] = pB + (long)(((tid + (t >> 1) * 256) / (BN / 4)) * 2 + (t & 1)) * p.ldb + min(n0 + ((tid + (t >> 1) * 256) % (BN / 4)) * 4, p.N - 4);
      else pb[t] = pB + (long)(f / (BN / 4)) * p.ldb + min(n0 + (f % (BN / 4)) * 4, p.N - 4);
    }
  }
  auto gload_full = [&](Raw4<EA> (&rra)[A4], Raw4<EB> (&rrb)[B4], int k0) {
#pragma unroll
    for (int t = 0; t < A4; ++t) rra[t] = ldraw(pa[t] + (A_KC ? (long)k0 : (long)k0 * p.lda));
#pragma unroll
    for (int t = 0; t < B4; ++t) rrb[t] = ldraw(pb[t] + (B_KC ? (long)k0 : (long)k0 * p.ldb));
  };

  // double-buffered LDS, one barrier per slab; the next slab's global loads fly under the MFMAs.  (Measured and rejected:
  // a second staging register set with the loads issued TWO slabs ahead and both prologue slabs in flight at once —
  // 5.91 vs 5.81 ms over the forward / input-gradient launches of a step, stand-alone: with 4 blocks per CU the other
  // waves already cover the load latency; what these launches pay is a fixed ~6 us each, see DESIGN.md.)
  auto slab = [&](int cur) {
    // only the first column block reports the column sums of A (bias gradient): the others skip the VALU adds
    if (PREC) gemm_slab_bf16<BM, BN, BK, PREC ? PREC : 1>(reinterpret_cast<const unsigned*>(As + cur * AF), reinterpret_cast<const unsigned*>(Bs + cur * BF), wr0, wc0, acc);
    else if (SUM_A && bx == 0) gemm_slab<BM, BN, BK, A_KC, B_KC, true>(As + cur * AF, Bs + cur * BF, wr0, wc0, acc, asum);
    else gemm_slab<BM, BN, BK, A_KC, B_KC, false>(As + cur * AF, Bs + cur * BF, wr0, wc0, acc, asum);
  };
  auto simple_loop = [&]() {  // one slab of prefetch (the next slab's global loads fly under the MFMAs of the running one)
    gload(rring_a[0], rring_b[0], kbeg);
    lstore(rring_a[0], rring_b[0], As, Bs);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      if (FAST && k0 + 2 * BK <= kend) gload_full(rring_a[0], rring_b[0], k0 + BK);
      else gload(rring_a[0], rring_b[0], k0 + BK);  // tail / past-the-end prefetch: clamped, zeroed, never consumed past kend
      __builtin_amdgcn_sched_barrier(0);
      slab(cur);
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
      lstore(rring_a[0], rring_b[0], As + cur * AF, Bs + cur * BF);
      __syncthreads();
    }
  };
  if constexpr (RD == 1 || !FAST) {
    if (kbeg < kend) simple_loop();
  } else if (kend - kbeg < RD * BK) {
    if (kbeg < kend) simple_loop();
  } else {
    // Register ring: slot d holds slab d (mod RD); the slot whose slab has just gone to LDS is refilled with the slab RD
    // ahead.  The prologue and the steady-state loop contain NO branch around a load: the compiler then knows how many
    // younger loads are in flight at every LDS store and waits with a counted vmcnt instead of draining the queue (what
    // it does as soon as a load sits behind a block-uniform branch).  Only the last < RD slabs of a reduction that is not
    // a multiple of RD * BK are fetched conditionally.
#pragma unroll
    for (int d = 0; d < RD; ++d) gload_full(rring_a[d], rring_b[d], kbeg + d * BK);
    lstore(rring_a[0], rring_b[0], As, Bs);
    __syncthreads();
    int cur = 0, k0 = kbeg;
    for (; k0 + 2 * RD * BK <= kend; k0 += RD * BK) {
#pragma unroll
      for (int d = 0; d < RD; ++d) {
        gload_full(rring_a[d], rring_b[d], k0 + (d + RD) * BK);
        __builtin_amdgcn_sched_barrier(0);
        slab(cur);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
        lstore(rring_a[(d + 1) % RD], rring_b[(d + 1) % RD], As + cur * AF, Bs + cur * BF);
        __syncthreads();
      }
    }
    // drain: the RD slabs in the ring (all inside the range), then the < RD slabs fetched on the way
#pragma unroll
    for (int d = 0; d < 2 * RD - 1; ++d) {
      const int kk = k0 + d * BK;
      if (kk < kend) {
        if (d < RD && kk + RD * BK < kend) {
          if (kk + (RD + 1) * BK <= kend) gload_full(rring_a[d], rring_b[d], kk + RD * BK);
          else gload(rring_a[d], rring_b[d], kk + RD * BK);
        }
        __builtin_amdgcn_sched_barrier(0);
        slab(cur);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
        if (kk + BK < kend) lstore(rring_a[(d + 1) % RD], rring_b[(d + 1) % RD], As + cur * AF, Bs + cur * BF);
        __syncthreads();
      }
    }
  }

  // ---- epilogue
  const bool fused = FAST && p.cnt != nullptr && gridDim.z > 1;
  // (a fused split stores its raw partial into the fp32 `part` slabs; everything else goes to C)
  EC* __restrict__ C = static_cast<EC*>(p.C) + (long)bz * p.part_stride;
  float* __restrict__ Cp = fused ? p.part + (long)bz * p.part_stride : nullptr;
  if (FAST) {
    // Stage each wave's 32 x WN sub-tile through LDS and leave as float4 rows: 16 B per lane loads of
    // residual / pre-activation and 16 B stores (4 B-per-lane stores ran at ~1 TB/s, 4x below HBM).
    float* st = smem + wave * (32 * SLD);
    const int l31 = tid & 31, hh = (tid >> 5) & 1, lane = tid & 63;
    constexpr int LPR = WN / 4;         // lanes per row
    constexpr int RPI = 64 / LPR;       // rows per wave-instruction
    __syncthreads();                    // every wave is done with the operand images
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * hh) * SLD + tn * 32 + l31] = acc[tm][tn][r];
      __syncthreads();
      const int c4 = lane % LPR;
      const int col = n0 + wc0 + c4 * 4;
      if (col < p.N) {
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int rl = it * RPI + lane / LPR;
          const int row = m0 + wr0 + tm * 32 + rl;
          if (row < p.M) {
            const long o = (long)row * p.ldc + col;
            const float4 a4 = ld4(st + rl * SLD + c4 * 4);
            if (fused) {
              st_agent4(Cp + o, a4);  // raw partial of this split
            } else {
              float v[4] = {a4.x, a4.y, a4.z, a4.w};
              gemm_epilogue4(p, C, o, col, v);
            }
          }
        }
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + acc_col(wc0, tn);
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + acc_row(wr0, tm, r);
          if (row >= p.M) continue;
          const long o = (long)row * p.ldc + col;
          float v = acc[tm][tn][r] + bv;
          if (p.pre) st1(p.pre + o, v);
          v = act_f(v, p.act);
          if (p.mulpre) v *= act_grad_f(ld1(p.mulpre + o), p.dact);
          if (p.drop_thresh) v *= dropout_scale(p.drop_seed, (unsigned long long)o, p.drop_thresh, p.drop_inv_keep);
          if (p.residual) v += ld1(p.residual + o);
          st1(C + o, v);
        }
      }
  }
  if (SUM_A && PREC) {
    // every staged slab was summed exactly once (slabs past the split range are zero): reduce the partial row sums
    // of the 256 / (BM / 4) threads that staged the same row quad, in a fixed order
    __syncthreads();
    if (p.bias_part && bx == 0) {
      float4* red = reinterpret_cast<float4*>(smem);
      red[tid] = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
      __syncthreads();
      if (tid < BM) {
        constexpr int G = 256 / (BM / 4);
        float sacc = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 v = red[g * (BM / 4) + (tid >> 2)];
          sacc += (tid & 3) == 0 ? v.x : (tid & 3) == 1 ? v.y : (tid & 3) == 2 ? v.z : v.w;
        }
        if (m0 + tid < p.M) __hip_atomic_store(p.bias_part + (long)bz * p.bias_stride + m0 + tid, sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else if (SUM_A) {
    if (p.bias_part && bx == 0 && (wave & 1) == 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        float s = asum[tm] + __shfl_xor(asum[tm], 32, 64);
        const int i = m0 + wr0 + tm * 32 + (tid & 31);
        if ((tid & 32) == 0 && i < p.M) __hip_atomic_store(p.bias_part + (long)bz * p.bias_stride + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (FAST && fused) splitk_fused_tail<SUM_A, BM, BN, EC>(p, bx, by, m0, n0, tid);
}

// out[e] = sum_z part[z * stride + e].  Block = 16 float4 columns x 16 z-lanes: coalesced 256-byte row
// segments per z, 16 independent partial sums per column, fixed-order LDS tree -> deterministic.
template <typename T>
__global__ __launch_bounds__(256) void reduce_parts_kernel(const T* __restrict__ part, T* __restrict__ out, long n,
                                                           long stride, int nz, int accumulate) {
  // (round 6 measured double accumulation here: +0.27 ms per step on the weight-gradient queue and no change of any gradient
  // error at full size — the 7.6e-5 it was meant to shrink came from three LeakyReLU sign flips, DESIGN.md section 2: reverted)
  __shared__ float4 red[16][16];
  const int q = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const long e = ((long)blockIdx.x * 16 + q) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e + 3 < n) {
    for (int z = zl; z < nz; z += 16) {
      const float4 v = ld4(part + (long)z * stride + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  } else if (e < n) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = zl; z < nz; z += 16)
      for (int k = 0; k < 4 && e + k < n; ++k) t[k] += ld1(part + (long)z * stride + e + k);
    s = make_float4(t[0], t[1], t[2], t[3]);
  }
  red[zl][q] = s;
  __syncthreads();
  if (zl == 0 && e < n) {
    float4 t = red[0][q];
    for (int k = 1; k < 16; ++k) {
      t.x += red[k][q].x; t.y += red[k][q].y; t.z += red[k][q].z; t.w += red[k][q].w;
    }
    const float tv[4] = {t.x, t.y, t.z, t.w};
    for (int k = 0; k < 4 && e + k < n; ++k) st1(out + e + k, accumulate ? ld1(out + e + k) + tv[k] : tv[k]);
  }
}

int lotus_reduce_parts(const float* part, float* out, long n, long stride, int nz, int accumulate, hipStream_t st) {
  if (n <= 0) return LOTUS_OK;
  LOTUS_LAUNCH(reduce_parts_kernel<float>, dim3(cdiv(n, 64)), dim3(256), 0, st, part, out, n, stride, nz, accumulate);
  LOTUS_LAUNCH_CHECK("lotus_reduce_parts");
  return LOTUS_OK;
}

// out[e] = sum_z part[z * stride + e], z = 0 .. nz-1 in fixed order (e.g. the key-side partial slots of the
// cross-attention backward)
extern "C" int lotus_sum_slabs(const act_t* part, act_t* out, long n, long stride, int nz, void* stream) {
  LOTUS_CHECK_ARG(part && out && n >= 0 && nz >= 1, "lotus_sum_slabs: bad arguments");
  if (n <= 0) return LOTUS_OK;
  LOTUS_LAUNCH(reduce_parts_kernel<act_t>, dim3(cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream, part, out, n, stride, nz, 0);
  LOTUS_LAUNCH_CHECK("lotus_sum_slabs");
  return LOTUS_OK;
}

// out[r][c] (row stride out_ld) = sum_z part[z * stride + r * cols + c], z ascending: the key-side partial slots of one
// cross-attention backward summed straight into that block's column slice of the shared [L][sum 2C] gradient slab
__global__ __launch_bounds__(256) void sum_slabs_ld_kernel(const act_t* __restrict__ part, act_t* __restrict__ out, int rows, int cols,
                                                           long out_ld, long stride, int nz) {
  const int c4 = cols / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * c4) return;
  const int r = (int)(i / c4), q = (int)(i % c4);
  const act_t* src = part + (long)r * cols + 4 * q;
  float4 s = ld4(src);
  int z = 1;
  for (; z + 3 < nz; z += 4) {
    const float4 v0 = ld4(src + (long)z * stride), v1 = ld4(src + (long)(z + 1) * stride);
    const float4 v2 = ld4(src + (long)(z + 2) * stride), v3 = ld4(src + (long)(z + 3) * stride);
    s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x; s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
    s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z; s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
  }
  for (; z < nz; ++z) {
    const float4 v = ld4(src + (long)z * stride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  st4(out + (long)r * out_ld + 4 * q, s);
}

extern "C" int lotus_sum_slabs_ld(const act_t* part, act_t* out, int rows, int cols, long out_ld, long stride, int nz, void* stream) {
  LOTUS_CHECK_ARG(part && out && rows >= 0 && cols > 0 && cols % 4 == 0 && out_ld % 4 == 0 && stride % 4 == 0 && nz >= 1,
                  "lotus_sum_slabs_ld: bad arguments (cols, out_ld and stride must be multiples of 4)");
  if (rows == 0) return LOTUS_OK;
  LOTUS_LAUNCH(sum_slabs_ld_kernel, dim3(cdiv((long)rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, part, out, rows, cols,
               out_ld, stride, nz);
  LOTUS_LAUNCH_CHECK("lotus_sum_slabs_ld");
  return LOTUS_OK;
}

static inline int vec_ok(const void* p, long ld) { return (((uintptr_t)p) % 16 == 0) && (ld % 4 == 0); }  // (bf16 rows: 8-byte accesses, same rule)

#define GEMM_KALIGN 64  // split-K ranges are multiples of the largest slab depth

// Element types of a launch: A is always an activation; B is the weight matrix (fp32 master) except in the weight gradient
// (both operands are activations); C is an activation except for the weight gradient and for split-K partial slabs
// (p.c_float).  In the fp32 build every combination is <float, float, float>.
#define GEMM_GO_RD(BM_, BN_, BK_, PREC_, grid_, RD_)                                                                             \
  do {                                                                                                                           \
    using TB_ = std::conditional_t<SUM_A, act_t, float>;                                                                         \
    if constexpr (LOTUS_ACT_IS_BF16 && !SUM_A) {                                                                                 \
      if (p.b_act) { /* bf16 weight shadow: half the weight bytes per block, no conversion while staging */                      \
        if constexpr (PREC_ == 1 && FAST) {                                                                                      \
          if (p.c_float) LOTUS_LAUNCH((gemm_kernel<BM_, BN_, BK_, A_KC, B_KC, SUM_A, FAST, PREC_, act_t, act_t, float, RD_>), grid_, block, 0, st, p); \
          else LOTUS_LAUNCH((gemm_kernel<BM_, BN_, BK_, A_KC, B_KC, SUM_A, FAST, PREC_, act_t, act_t, act_t, RD_>), grid_, block, 0, st, p); \
        }                                                                                                                        \
      } else if (p.c_float) LOTUS_LAUNCH((gemm_kernel<BM_, BN_, BK_, A_KC, B_KC, SUM_A, FAST, PREC_, act_t, TB_, float, RD_>), grid_, block, 0, st, p); \
      else LOTUS_LAUNCH((gemm_kernel<BM_, BN_, BK_, A_KC, B_KC, SUM_A, FAST, PREC_, act_t, TB_, act_t, RD_>), grid_, block, 0, st, p);   \
    } else {                                                                                                                     \
      LOTUS_LAUNCH((gemm_kernel<BM_, BN_, BK_, A_KC, B_KC, SUM_A, FAST, PREC_, act_t, TB_, float, RD_>), grid_, block, 0, st, p);      \
    }                                                                                                                            \
  } while (0)
#define GEMM_GO(BM_, BN_, BK_, PREC_, grid_) GEMM_GO_RD(BM_, BN_, BK_, PREC_, grid_, 1)

// Tuning constants of gemm_kernel.  Every one of them was an environment switch while it was being measured (rounds 1-4;
// DESIGN.md section 4 has the A/B of each); the losers and their code are gone, these are the values that shipped:
//  * 64 x 64 block tiles everywhere: 128 x 128 / 128 x 64 tiles of THIS kernel win stand-alone from 4 blocks per CU but lost
//    inside the step at every batch size (820 vs 806 samples/s at 16 clouds) — the tall products now have gemm_dma_kernel;
//  * slab depth by grid size: <= 512 blocks 64 (nothing else on the CU hides the load latency), <= 2048 blocks 32, larger
//    grids 16 (occupancy); streamed weight-gradient operands 32; bf16 products always 32 (64 / 128: 1124 / 1020 vs 1134 samples/s);
//  * register staging ring: two slabs for exact-fp32 products on grids of <= 8192 blocks (+1.0 %, bit-identical; four lose to
//    their registers), four slabs for bf16 products over <= 8192 rows (+3 % on the PerAct preset; the tall layers lose: 7 -> 3
//    resident blocks per CU), none for weight gradients.
constexpr long kF32RingMaxBlocks = 8192;
constexpr int kBf16RingMaxRows = 8192;

template <bool A_KC, bool B_KC, bool SUM_A, bool FAST>
static int launch_gemm_t(GemmP& p, int nz, hipStream_t st) {
  const long blocks64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64);
  dim3 block(256);
  dim3 g64(cdiv(p.N, 64), cdiv(p.M, 64), nz);
  const int g_prec = p.prec;
  if (g_prec && SUM_A && FAST && !A_KC && !B_KC) {
    // weight gradients (split-K, 64x64 tiles): operands converted while staged; bias sums from the fp32 registers
    if constexpr (SUM_A && !A_KC && !B_KC && FAST) {
      if (g_prec == 1) {
        if constexpr (LOTUS_ACT_IS_BF16) {
          if (p.K <= kBf16RingMaxRows) GEMM_GO_RD(64, 64, 32, 1, g64, 4);
          else GEMM_GO(64, 64, 32, 1, g64);
        } else {
          GEMM_GO(64, 64, 32, 1, g64);
        }
      } else if constexpr (!LOTUS_ACT_IS_BF16) GEMM_GO(64, 64, 32, 3, g64);
    }
    LOTUS_LAUNCH_CHECK("lotus_gemm(bf16 wgrad)");
    return LOTUS_OK;
  }
  if (p.b_act && !(LOTUS_ACT_IS_BF16 && g_prec == 1 && !SUM_A && FAST && (A_KC || p.M % 4 == 0))) {
    lotus_set_error("lotus_linear: bf16 weight shadows (precision 5) need the bf16-storage build and 16-byte aligned operands whose "
                    "widths are multiples of 4");
    return LOTUS_E_UNSUPPORTED;
  }
  if (g_prec && !SUM_A && FAST && (A_KC || p.M % 4 == 0)) {
    // bf16 / bf16x3 operand path (forward and input-gradient products).  bf16 storage: one bf16 MFMA covers 16 k, so a
    // 32-deep slab is two MFMAs per wave between barriers and the block pays one global-load round trip per slab; the
    // staging ring keeps four slabs in flight where the grid is small enough to afford its registers.
    if constexpr (!SUM_A && FAST) {
      if (g_prec == 1) {
        if constexpr (LOTUS_ACT_IS_BF16) {
          if (p.M <= kBf16RingMaxRows) GEMM_GO_RD(64, 64, 32, 1, g64, 4);
          else GEMM_GO(64, 64, 32, 1, g64);
        } else GEMM_GO(64, 64, 32, 1, g64);
      } else if constexpr (!LOTUS_ACT_IS_BF16) {
        GEMM_GO(64, 64, 32, 3, g64);
      }
    }
    LOTUS_LAUNCH_CHECK("lotus_gemm(bf16)");
    return LOTUS_OK;
  }
  if constexpr (LOTUS_ACT_IS_BF16) {
    // bf16-storage build: the exact-fp32 product path only serves the shapes the vectorised bf16 path cannot take
    // (odd widths such as the 90- and 217-wide head layers)
    GEMM_GO(64, 64, 32, 0, g64);
  } else {
    // (tap-grouped products: about half of the row tiles leave at once — the grid the heuristics should see is the active one)
    const long blocks_eff = p.tap_rows ? blocks64 / 2 : blocks64;
    const int bk = SUM_A ? 32 : (blocks_eff * nz <= 512 ? 64 : (blocks_eff * nz <= 2048 ? 32 : 16));
    const int rd = SUM_A ? 1 : (blocks_eff * nz <= kF32RingMaxBlocks ? 2 : 1);
    if (p.tap_rows) {
      if constexpr (A_KC && !SUM_A && FAST && !LOTUS_ACT_IS_BF16) {
        if (bk == 64) LOTUS_LAUNCH((gemm_kernel<64, 64, 64, A_KC, B_KC, SUM_A, FAST, 0, act_t, float, float, 2, true>), g64, block, 0, st, p);
        else if (bk == 32) LOTUS_LAUNCH((gemm_kernel<64, 64, 32, A_KC, B_KC, SUM_A, FAST, 0, act_t, float, float, 2, true>), g64, block, 0, st, p);
        else LOTUS_LAUNCH((gemm_kernel<64, 64, 16, A_KC, B_KC, SUM_A, FAST, 0, act_t, float, float, 2, true>), g64, block, 0, st, p);
        LOTUS_LAUNCH_CHECK("lotus_gemm(tap-grouped)");
        return LOTUS_OK;
      } else {
        lotus_set_error("lotus_gemm: tap-grouped products need k-contiguous, 16-byte aligned fp32 rows");
        return LOTUS_E_UNSUPPORTED;
      }
    }
    if (rd == 2 && bk == 64) GEMM_GO_RD(64, 64, 64, 0, g64, 2);
    else if (rd == 2 && bk == 16) GEMM_GO_RD(64, 64, 16, 0, g64, 2);
    else if (rd == 2) GEMM_GO_RD(64, 64, 32, 0, g64, 2);
    else if (bk == 64) GEMM_GO(64, 64, 64, 0, g64);
    else if (bk == 32) GEMM_GO(64, 64, 32, 0, g64);
    else GEMM_GO(64, 64, 16, 0, g64);
  }
  LOTUS_LAUNCH_CHECK("lotus_gemm");
  return LOTUS_OK;
}

// FAST needs float4-safe extents on both operands: the contiguous dimension must be a multiple of 4
// (K for k-contiguous operands, M / N otherwise) and at least 4 wide
template <bool A_KC, bool B_KC>
static bool fast_ok(const GemmP& p) {
  const bool a_ok = p.a_vec && (A_KC ? (p.K % 4 == 0 && p.K >= 4) : (p.M % 4 == 0 && p.M >= 4));
  const bool b_ok = p.b_vec && (B_KC ? (p.K % 4 == 0 && p.K >= 4) : (p.N % 4 == 0 && p.N >= 4));
  const bool c_ok = p.N % 4 == 0 && p.ldc % 4 == 0 && p.part_stride % 4 == 0 && vec_ok(p.C, p.ldc) &&
                    (!p.bias || ((uintptr_t)p.bias) % 16 == 0) && (!p.residual || ((uintptr_t)p.residual) % 16 == 0) &&
                    (!p.pre || ((uintptr_t)p.pre) % 16 == 0) && (!p.mulpre || ((uintptr_t)p.mulpre) % 16 == 0) &&
                    (!p.part || ((uintptr_t)p.part) % 16 == 0);
  return a_ok && b_ok && c_ok;
}

template <bool A_KC, bool B_KC, bool SUM_A>
static int launch_gemm(GemmP& p, int nz, hipStream_t st) {
  const bool a_ok = p.a_vec && (A_KC ? (p.K % 4 == 0 && p.K >= 4) : (p.M % 4 == 0 && p.M >= 4));
  const bool b_ok = p.b_vec && (B_KC ? (p.K % 4 == 0 && p.K >= 4) : (p.N % 4 == 0 && p.N >= 4));
  const bool c_ok = p.N % 4 == 0 && p.ldc % 4 == 0 && p.part_stride % 4 == 0 && vec_ok(p.C, p.ldc) &&
                    (!p.bias || ((uintptr_t)p.bias) % 16 == 0) && (!p.residual || ((uintptr_t)p.residual) % 16 == 0) &&
                    (!p.pre || ((uintptr_t)p.pre) % 16 == 0) && (!p.mulpre || ((uintptr_t)p.mulpre) % 16 == 0);
  if (a_ok && b_ok && c_ok) {
    if constexpr (A_KC || SUM_A) {  // tall products: the LDS-DMA kernels (gemm_dma.hip) where they apply
      const int rc = launch_gemm_dma(p, A_KC ? (B_KC ? 0 : 1) : 2, nz, st);
      if (rc != LOTUS_GEMM_DMA_NA) return rc;
    }
    return launch_gemm_t<A_KC, B_KC, SUM_A, true>(p, nz, st);
  }
  return launch_gemm_t<A_KC, B_KC, SUM_A, false>(p, nz, st);
}

static void set_drop(GemmP& p, float drop_p, unsigned long long seed) {
  p.drop_seed = seed;
  lotus_drop_setup(drop_p, &p.drop_thresh, &p.drop_inv_keep);
}

// sum of split-K partials + the full epilogue (bias, pre, act, act', dropout, residual), float4 wide
__global__ void splitk_epilogue_kernel(GemmP p, const float* __restrict__ part, long stride, int nz) {
  const int n4 = p.N / 4;
  const long total4 = (long)p.M * n4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / n4), col = (int)(i % n4) * 4;
    const long o = (long)row * p.ldc + col;
    float4 s4 = ld4(part + o);
    for (int z = 1; z < nz; ++z) {
      const float4 v = ld4(part + (long)z * stride + o);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    float v[4] = {s4.x, s4.y, s4.z, s4.w};
    if (p.bias) {
      const float4 b = ld4(p.bias + col);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.pre) st4(p.pre + o, make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.act);
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
    st4(static_cast<act_t*>(p.C) + o, make_float4(v[0], v[1], v[2], v[3]));
  }
}

// Few output tiles but a long reduction (deep levels: M = 361..1450, K up to 3072): split K over
// blockIdx.z into a workspace and finish with splitk_epilogue_kernel.  Returns the split count.
static int fwd_splits(int M, int N, int K) {
  const long blocks = (long)cdiv(M, 64) * cdiv(N, 64);
  int nz = 1;
  // fewer than 2 blocks per CU and a long reduction: split K until ~2 blocks per CU, >= 384 deep each.  (Measured with
  // the reduction fused into the GEMM: ranges of 128-256 do NOT pay off — 1450x512x512 15 -> 19.5 us, 6077x256x256
  // 16.6 -> 19 us — the fixed latency of a block, not its MFMA time, bounds these sizes.)
  while (nz < 16 && blocks * nz < 512 && K / (nz * 2) >= 384) nz *= 2;
  return nz;
}

#define LOTUS_SPLITK_MAX_TILES 4096  // per-tile arrival counters of the fused split-K path (one unsigned each)

static bool splitk_fused_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("LOTUS_SPLITK_FUSED"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}

template <bool A_KC, bool B_KC>
static int run_gemm_splitk(GemmP& p, void* workspace, size_t workspace_bytes, unsigned* counters, hipStream_t st) {
  if (!splitk_fused_enabled()) counters = nullptr;
  const int nz = (p.N % 4 == 0 && p.ldc == p.N) ? fwd_splits(p.M, p.N, p.K) : 1;
  const size_t need = (size_t)nz * p.M * p.N * sizeof(float);
  if (nz == 1 || !workspace || workspace_bytes < need || ((uintptr_t)workspace) % 16) return launch_gemm<A_KC, B_KC, false>(p, 1, st);
  const int klen = cdiv(cdiv(p.K, nz), GEMM_KALIGN) * GEMM_KALIGN;
  const long tiles = (long)cdiv(p.M, 64) * cdiv(p.N, 64);
  if (counters && tiles <= LOTUS_SPLITK_MAX_TILES) {
    // one launch: partials + last-arrival reduction with the full epilogue
    GemmP q = p;
    q.part = (float*)workspace; q.part_stride = (long)p.M * p.N; q.cnt = counters; q.klen = klen;
    if (fast_ok<A_KC, B_KC>(q)) return launch_gemm<A_KC, B_KC, false>(q, nz, st);
  }
  GemmP q = p;
  q.C = workspace; q.c_float = 1; q.part_stride = (long)p.M * p.N;
  q.bias = nullptr; q.residual = nullptr; q.pre = nullptr; q.mulpre = nullptr; q.act = LOTUS_ACT_NONE; q.drop_thresh = 0;
  q.klen = klen;
  StopEventOnLast stop_ev;
  int rc = launch_gemm<A_KC, B_KC, false>(q, nz, st);
  if (rc) return rc;
  const long total4 = (long)p.M * p.N / 4;
  int g = cdiv(total4, 256);
  stop_ev.last();
  LOTUS_LAUNCH(splitk_epilogue_kernel, dim3(g > 2048 ? 2048 : g), dim3(256), 0, st, p, (const float*)workspace,
                     (long)p.M * p.N, nz);
  LOTUS_LAUNCH_CHECK("lotus_gemm(split-K epilogue)");
  return LOTUS_OK;
}

// Sparse convolution of a deep level as ONE launch of 27 gathered products (conv.hip, lotus_subm_conv tap path): mode 0
// part[27 * n64][cout] = x[tg_in[.]] W_t^T, mode 1 part[27 * n64][cin] = dy[tg_in[.]] W_(26-t).  w is the module's weight
// [cout][27][cin]: tap t is a [cout][cin] slice with row stride 27 * cin.  Exact fp32 products, fp32-storage build.
int lotus_conv_tap_gemm(int mode, const act_t* x, const float* w, float* part, const int* tg_in, const int* tg_cnt, int n64,
                        int src_rows, int cin, int cout, hipStream_t st) {
  if constexpr (LOTUS_ACT_IS_BF16) {
    return LOTUS_E_UNSUPPORTED;
  } else {
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.B = w; p.C = part;
    p.M = 27 * n64;
    p.N = mode == 0 ? cout : cin;
    p.K = mode == 0 ? cin : cout;
    p.lda = p.K; p.ldb = 27L * cin; p.ldc = p.N;
    p.act = LOTUS_ACT_NONE;
    p.klen = cdiv(p.K, GEMM_KALIGN) * GEMM_KALIGN;
    p.a_vec = vec_ok(x, p.K); p.b_vec = vec_ok(w, cin) && cin % 4 == 0; p.prec = 0;
    p.drop_inv_keep = 1.f;
    p.a_rows = tg_in; p.tap_cnt = tg_cnt; p.tap_rows = n64; p.b_tap_mirror = mode == 1; p.b_tap_stride = cin;
    p.a_src_rows = src_rows;
    {  // the LDS-DMA tiles where they apply (gemm_dma.h, gemm_dma_tap_kernel); else the 64 x 64 kernel below
      GemmP q = p;
      const int rc = launch_gemm_dma_tap(q, mode, st);
      if (rc != LOTUS_GEMM_DMA_NA) return rc;
    }
    if (!(fast_ok<true, true>(p))) {
      lotus_set_error("lotus_subm_conv(tap-grouped): rows, weights and the partial slab must be 16-byte aligned with widths that are multiples of 4");
      return LOTUS_E_UNSUPPORTED;
    }
    return mode == 0 ? launch_gemm<true, true, false>(p, 1, st) : launch_gemm<true, false, false>(p, 1, st);
  }
}

extern "C" {

// y = dropout(act(x w^T + bias)) + residual ; pre (optional) receives x w^T + bias.
// the zeroed per-stream counter buffer: one arrival counter per output tile of a fused split-K product, followed by the
// 64 counters of the fused BatchNorm statistics (norm.hip; lotus_bn_counters_offset() = their byte offset)
size_t lotus_splitk_counters_bytes(void) { return (size_t)(LOTUS_SPLITK_MAX_TILES + 64) * sizeof(unsigned); }
size_t lotus_bn_counters_offset(void) { return (size_t)LOTUS_SPLITK_MAX_TILES * sizeof(unsigned); }

size_t lotus_linear_workspace(int M, int N, int K) {
  const int a = fwd_splits(M, N, K), b = fwd_splits(M, K, N);
  const size_t wa = a > 1 ? (size_t)a * M * N * sizeof(float) : 0, wb = b > 1 ? (size_t)b * M * K * sizeof(float) : 0;
  return wa > wb ? wa : wb;
}

int lotus_linear_fwd(const act_t* x, const float* w, const float* bias, const act_t* residual, act_t* y,
                     act_t* pre, int M, int N, int K, int act, float drop_p, unsigned long long drop_seed,
                     int precision, void* workspace, size_t workspace_bytes, void* counters, void* stream) {
  LOTUS_CHECK_ARG(x && w && y && M >= 0 && N > 0 && K > 0, "lotus_linear_fwd: bad arguments");
  // precision 5 (bf16-storage build only) = precision 1 with `w` pointing at a bf16 shadow of the weights
  const int w_shadow = precision == 5;
  LOTUS_CHECK_ARG(!w_shadow || LOTUS_ACT_IS_BF16, "lotus_linear_fwd: precision 5 (bf16 weight shadow) exists in the bf16-storage build only");
  if (w_shadow) precision = 1;
  LOTUS_CHECK_ARG(precision == 0 || precision == 1 || precision == 3, "lotus_linear_fwd: precision must be 0, 1 or 3 (5 with bf16 weights)");
  LOTUS_CHECK_ARG(!(LOTUS_ACT_IS_BF16 && precision == 3), "lotus_linear_fwd: bf16x3 operands (precision 3) are not built for bf16 activation storage; use 1 (bf16) or 0");
  if (M == 0) return LOTUS_OK;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = w; p.C = y; p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldb = K; p.ldc = N;
  p.bias = bias; p.residual = residual; p.pre = pre; p.act = act;
  p.klen = cdiv(K, GEMM_KALIGN) * GEMM_KALIGN;
  p.a_vec = vec_ok(x, K); p.b_vec = vec_ok(w, K); p.prec = precision; p.b_act = w_shadow;
  set_drop(p, drop_p, drop_seed);
  return run_gemm_splitk<true, true>(p, workspace, workspace_bytes, (unsigned*)counters, (hipStream_t)stream);
}

// dx = (dy w) * act'(pre) * dropmask + add.   dy [M,N], w [N,K], dx/pre/add [M,K].
// `pre`/`act`/`drop_*` describe the layer that PRODUCED this layer's input (its pre-activation,
// activation and output dropout), so the chain rule through it is fused into this epilogue.
int lotus_linear_dgrad(const act_t* dy, const float* w, act_t* dx, const act_t* pre, const act_t* add, int M,
                       int N, int K, int act, float drop_p, unsigned long long drop_seed, int precision, void* workspace,
                       size_t workspace_bytes, void* counters, void* stream) {
  LOTUS_CHECK_ARG(dy && w && dx && M >= 0 && N > 0 && K > 0, "lotus_linear_dgrad: bad arguments");
  // precision 5 (bf16-storage build only) = precision 1 with `w` pointing at a bf16 shadow of the weights
  const int w_shadow = precision == 5;
  LOTUS_CHECK_ARG(!w_shadow || LOTUS_ACT_IS_BF16, "lotus_linear_dgrad: precision 5 (bf16 weight shadow) exists in the bf16-storage build only");
  if (w_shadow) precision = 1;
  LOTUS_CHECK_ARG(precision == 0 || precision == 1 || precision == 3, "lotus_linear_dgrad: precision must be 0, 1 or 3 (5 with bf16 weights)");
  LOTUS_CHECK_ARG(!(LOTUS_ACT_IS_BF16 && precision == 3), "lotus_linear_dgrad: bf16x3 operands (precision 3) are not built for bf16 activation storage; use 1 (bf16) or 0");
  if (M == 0) return LOTUS_OK;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.B = w; p.C = dx; p.M = M; p.N = K; p.K = N;
  p.lda = N; p.ldb = K; p.ldc = K;
  p.act = LOTUS_ACT_NONE;
  p.mulpre = pre; p.dact = act; p.residual = add;
  p.klen = cdiv(N, GEMM_KALIGN) * GEMM_KALIGN;
  p.a_vec = vec_ok(dy, N); p.b_vec = vec_ok(w, K); p.prec = precision; p.b_act = w_shadow;
  set_drop(p, drop_p, drop_seed);
  return run_gemm_splitk<true, false>(p, workspace, workspace_bytes, (unsigned*)counters, (hipStream_t)stream);
}

// The input gradient of a linear layer whose INPUT is a LayerNorm output, together with that LayerNorm's backward:
//   dn = dy w,   dx = LN'(dn; x, mean, rstd, gamma) + add,   dz = dx * dropout-mask(dz_p, dz_seed) (optional),
// column partials of dgamma / dbeta left in ln_workspace as [*nparts][2][K] for lotus_layernorm_bwd_params_n.
// Where the rows are many and one 128-wide tile covers them (K = 64 / 128: levels 0-1 of the backbone) the LayerNorm
// backward is the EPILOGUE of the product (gemm_dma_kernel, EPI 2): dn never reaches HBM, ln_bwd_kernel's four passes over
// [M, K] and its launch are gone.  Everywhere else: lotus_linear_dgrad into `dn`, then lotus_layernorm_bwd — same results to
// summation order.  (model.py:659-680: norm1 -> qkv, norm2 -> fc1; model_ca.py:114-124)
int lotus_layernorm_bwd(const act_t* dy, const act_t* x, const float* mean, const float* rstd, const float* gamma, const act_t* add,
                        act_t* dx, float* dgamma, float* dbeta, int M, int C, int accumulate, act_t* dz, float drop_p,
                        unsigned long long drop_seed, void* workspace, size_t workspace_bytes, void* stream);  // norm.hip
int lotus_layernorm_bwd_parts(int M, int C);                                                                  // norm.hip
int lotus_linear_dgrad_ln(const act_t* dy, const float* w, const act_t* x, const float* mean, const float* rstd, const float* gamma,
                          const act_t* add, act_t* dx, act_t* dn, act_t* dz, float dz_p, unsigned long long dz_seed, int M, int N,
                          int K, int precision, void* workspace, size_t workspace_bytes, void* counters, void* ln_workspace,
                          size_t ln_workspace_bytes, int* nparts, void* stream) {
  LOTUS_CHECK_ARG(dy && w && x && mean && rstd && gamma && dx && dn && nparts && ln_workspace && M >= 0 && N > 0 && K > 0,
                  "lotus_linear_dgrad_ln: bad arguments");
  LOTUS_CHECK_ARG(!dz || dz_p > 0.f, "lotus_linear_dgrad_ln: dz needs dz_p > 0");
  if (M == 0) { *nparts = 0; return LOTUS_OK; }
  if (precision == 0 && !LOTUS_ACT_IS_BF16 && (size_t)cdiv(M, 128) * 2 * K * sizeof(float) <= ln_workspace_bytes) {
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.A = dy; p.B = w; p.C = dx; p.M = M; p.N = K; p.K = N;
    p.lda = N; p.ldb = K; p.ldc = K;
    p.act = LOTUS_ACT_NONE; p.residual = add;
    p.klen = cdiv(N, GEMM_KALIGN) * GEMM_KALIGN;
    p.a_vec = vec_ok(dy, N); p.b_vec = vec_ok(w, K); p.prec = 0;
    set_drop(p, dz ? dz_p : 0.f, dz_seed);
    p.ln_x = x; p.ln_mean = mean; p.ln_rstd = rstd; p.ln_gamma = gamma; p.ln_part = (float*)ln_workspace; p.ln_dz = dz;
    if (p.a_vec && p.b_vec && vec_ok(dx, K) && vec_ok(x, K) && (!add || vec_ok(add, K))) {
      const int rc = launch_gemm_dma(p, 1, 1, (hipStream_t)stream);
      if (rc != LOTUS_GEMM_DMA_NA) { *nparts = cdiv(M, 128); return rc; }
    }
  }
  StopEventOnLast stop_ev;  // (an armed stop event belongs to the LAST launch: the LayerNorm backward)
  int rc = lotus_linear_dgrad(dy, w, dn, nullptr, nullptr, M, N, K, LOTUS_ACT_NONE, 0.f, 0, precision, workspace, workspace_bytes, counters, stream);
  if (rc) return rc;
  stop_ev.last();
  rc = lotus_layernorm_bwd(dn, x, mean, rstd, gamma, add, dx, nullptr, nullptr, M, K, 0, dz, dz ? dz_p : 0.f, dz_seed, ln_workspace,
                           ln_workspace_bytes, stream);
  *nparts = lotus_layernorm_bwd_parts(M, K);
  return rc;
}

static int wgrad_splits(int M, int N, int K) {
  if (const int dz = gemm_dma_wgrad_splits(M, N, K)) return dz;  // tall products on the LDS-DMA kernels: their own tiling
  int nz = 1;
  const long tiles = (long)cdiv(N, 64) * cdiv(K, 64);
  if (tiles >= 128 && M < 1024) return 1;  // deep levels: the second (reduce) launch costs more than it hides
  // both operands are streamed exactly once: keep ~4 blocks per CU in flight (Little's law), >= 128 rows each
  while (nz < 256 && tiles * nz < 1024 && M / (nz * 2) >= 128) nz *= 2;
  return nz;
}

size_t lotus_linear_wgrad_workspace(int M, int N, int K) {
  return (size_t)wgrad_splits(M, N, K) * ((size_t)N * K + N) * sizeof(float);
}

// dw (+)= dy^T x ; db (+)= colsum(dy).   dy [M,N], x [M,K], dw [N,K], db [N] (optional).
// When db == dw + N*K (one contiguous [N*K + N] gradient buffer) the split-K partials of both are
// summed by a single launch; with one split and accumulate == 0 the GEMM writes dw/db directly.
int lotus_linear_wgrad(const act_t* dy, const act_t* x, float* dw, float* db, int M, int N, int K,
                       int accumulate, int precision, void* workspace, size_t workspace_bytes, void* counters,
                       void* stream) {
  LOTUS_CHECK_ARG(dy && x && dw && M >= 0 && N > 0 && K > 0, "lotus_linear_wgrad: bad arguments");
  if (precision == 5) precision = 1;  // (the shadow flag of the forward / input-gradient products: no weights are read here)
  LOTUS_CHECK_ARG(precision == 0 || precision == 1 || precision == 3, "lotus_linear_wgrad: precision must be 0, 1 or 3");
  LOTUS_CHECK_ARG(!(LOTUS_ACT_IS_BF16 && precision == 3), "lotus_linear_wgrad: bf16x3 operands (precision 3) are not built for bf16 activation storage; use 1 (bf16) or 0");
  hipStream_t st = (hipStream_t)stream;
  const int nz = wgrad_splits(M, N, K);
  const size_t slab = (size_t)N * K + N;
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)nz * slab * sizeof(float),
                  "lotus_linear_wgrad: workspace too small (%zu < %zu)", workspace_bytes, (size_t)nz * slab * sizeof(float));
  const bool direct = nz == 1 && !accumulate;
  float* part = (float*)workspace;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.B = x; p.M = N; p.N = K; p.K = M;
  p.lda = N; p.ldb = K; p.ldc = K;
  p.klen = cdiv(cdiv(M > 0 ? M : 1, nz), GEMM_KALIGN) * GEMM_KALIGN;
  p.a_vec = vec_ok(dy, N); p.b_vec = vec_ok(x, K); p.prec = precision;
  set_drop(p, 0.f, 0);
  constexpr int fuse_max = 4;  // measured: nz 4 fused 46 -> 41 us, nz 16 fused 42 -> 51 us
  const long wtiles = (long)cdiv(N, 64) * cdiv(K, 64);
  if (!direct && counters && splitk_fused_enabled() && nz <= fuse_max && wtiles <= LOTUS_SPLITK_MAX_TILES) {
    // few splits: the last block of every output tile sums the partials (and the bias partials) itself
    GemmP q = p;
    q.C = dw; q.part = part; q.part_stride = (long)slab; q.cnt = (unsigned*)counters;
    q.bias_part = db ? part + (size_t)N * K : nullptr; q.bias_stride = (long)slab; q.bias_out = db; q.accumulate = accumulate;
    if (fast_ok<false, false>(q)) return launch_gemm<false, false, true>(q, nz, st);
  }
  if (direct) {
    p.C = dw; p.bias_part = db;
  } else {
    p.C = part; p.part_stride = (long)slab;
    p.bias_part = db ? part + (size_t)N * K : nullptr; p.bias_stride = (long)slab;
  }
  int rc = launch_gemm<false, false, true>(p, nz, st);
  if (rc || direct) return rc;
  const long n = (long)N * K;
  if (db && db == dw + n) return lotus_reduce_parts(part, dw, (long)slab, (long)slab, nz, accumulate, st);
  rc = lotus_reduce_parts(part, dw, n, (long)slab, nz, accumulate, st);
  if (!rc && db) rc = lotus_reduce_parts(part + n, db, (long)N, (long)slab, nz, accumulate, st);
  return rc;
}

}  // extern "C"

}  // namespace LOTUS_NS
