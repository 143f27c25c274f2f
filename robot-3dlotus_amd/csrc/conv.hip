// lotus-hip: submanifold sparse convolution (the spconv.SubMConv3d call sites of the reference:
// PointTransformerV3/model.py:615-622 (3^3 CPE of every Block) and :844-853 (5^3 stem); 29 % of
// all MACs) as gather-GEMMs on the fp32 MFMA path.
//
//   fwd   : y[p, :]  = sum_t W[:, t, :]   x[nbr[t][p], :] + b
//   dgrad : dx[q, :] = sum_t W[:, T-1-t, :]^T dy[nbr[t][q], :]      (mirrored taps)
//   wgrad : dW[:, t, :] = sum_p dy[p, :] (x) x[nbr[t][p], :]        (active pairs compacted per tap)
//
// The neighbour table is tap-major int32 nbr[T][N] (-1 = absent) built once per level by the
// front-end (front_end.hip) and shared by the encoder and decoder Blocks of that level.
// Output rows are processed in serialised (space-filling-curve) order so that a 128-row block is
// spatially coherent: taps with no active neighbour in the block are skipped for the whole block,
// and per-wave 32-row groups skip their MFMAs through a ballot mask.
#include "mma.h"

namespace LOTUS_NS {

int lotus_reduce_parts(const float* part, float* out, long n, long stride, int nz, int accumulate, hipStream_t st);
int lotus_conv_pairs_try(int mode, const act_t* x, const float* w, const float* w_t, const float* bias, const act_t* add,
                         act_t* y, const int* nbr, const int* rowidx, int n, int T, int cin, int cout, void* workspace,
                         size_t workspace_bytes, int prec, hipStream_t st, int* rc);
size_t lotus_conv_pairs_workspace(int n, int ND);
int lotus_conv_weight_transpose_impl(const float* w, float* wt, int cout, int T, int cin, int prec, hipStream_t st);
int lotus_conv_tap_gemm(int mode, const act_t* x, const float* w, float* part, const int* tg_in, const int* tg_cnt, int n64,
                        int src_rows, int cin, int cout, hipStream_t st);  // gemm.hip

// ---- tap-grouped path ----------------------------------------------------------------------------------------------------
// The convolution as 27 GATHERED dense products in one launch (rows through the tap plan of the front-end, one weight slice
// per row tile) into a partial slab [27 x n64][C], followed by a fixed-order gather-sum over the taps of every output row.
// Round 4 introduced it for the deep levels (few rows, wide layers), where the pair-compacted kernel streams the weights of
// its taps (3 x C x 128 x 4 bytes) through every 64-row tile: at level 3 of the bench batch (1450 rows, C = 512) 23x the
// 28 MB weight tensor per launch.  Round 5 found the opposite end to be just as bad for the pair kernel: at 4096 points per
// cloud a level-0 point has 7.2 active taps of 27 (level 1: 11.1; at 1024 points per cloud 2.6 / 6.6), so a (64-row tile, tap)
// holds ~17 (~4) pairs and its 32-pair MFMA groups are 50 % / 63 % (22 % / 46 %) full (tools/conv_bench.py) — the tap plan compacts the pairs of a tap over the WHOLE level instead,
// the groups are full, and the cost is the partial slab's round trip through HBM (pairs x C x 4 bytes each way).  With the
// products on the LDS-DMA tiles (gemm_dma_tap_kernel) the path is ~2x the pair kernel at every level of both bench shapes
// (tools/dbg/tap_conv_check.py) and worth +7 % of the step; it is now taken from 64 channels up (LOTUS_CONV_TAP_MINC) for
// every level whose slab [27 x n64][C] of fp32 stays below 2047 MB (LOTUS_CONV_TAP_SLAB_MB: the byte offsets of the DMA
// kernel are 31-bit; only the rows of active pairs are ever touched, but the caller's workspace has to span it).  Larger
// shapes, other widths and the bf16 operand modes stay on the pair-compacted kernel.
static int tap_min_width() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("LOTUS_CONV_TAP_MINC"); v = e ? atoi(e) : 64; }
  return v;
}
static size_t tap_slab_max() {
  static long v = -1;
  if (v < 0) { const char* e = getenv("LOTUS_CONV_TAP_SLAB_MB"); v = e ? atol(e) : 2047; }
  return (size_t)v << 20;
}
static size_t tap_part_bytes(int n, int ND);
static bool tap_shape_ok(int n, int cin, int cout) {
  return !LOTUS_ACT_IS_BF16 && n > 0 && cin >= tap_min_width() && cout >= tap_min_width() && cin % 64 == 0 && cout % 64 == 0 &&
         tap_part_bytes(n, cin > cout ? cin : cout) <= tap_slab_max();
}
static size_t tap_part_bytes(int n, int ND) { return (size_t)27 * ((n + 63) / 64 * 64) * ND * sizeof(float); }

// y[i] = bias + add[i] + sum over taps t = 0..26 of part[pos[t][i]]  (rows without the tap: pos = -1)
__global__ __launch_bounds__(256) void conv_tap_reduce_kernel(const float* __restrict__ part, const int* __restrict__ pos, int n, int ND,
                                                              const float* __restrict__ bias, const act_t* __restrict__ add,
                                                              act_t* __restrict__ y) {
  LOTUS_T_PRIO();
  const int n4 = ND / 4;
  const long total = (long)n * n4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int i = (int)(e / n4), c = (int)(e % n4) * 4;
    int q[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) q[t] = pos[(long)t * n + i];  // (all loads before the first dependent one)
    float4 acc = bias ? ld4(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      if (q[t] >= 0) {
        const float4 v = ld4(part + (long)q[t] * ND + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    if (add) {
      const float4 a = ld4(add + (long)i * ND + c);
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    st4(y + (long)i * ND + c, acc);
  }
}

struct ConvP {
  const act_t* x;   // [n][KD] gathered operand (features, or dy for dgrad)
  const float* w;   // [cout][T][cin]
  act_t* y;         // [n][ND]
  const float* bias;
  const act_t* add;  // [n][ND] optional addend
  const int* nbr;    // [T][n]
  const int* rowidx; // [n] processing order or null
  int n, T, cin, cout;
  int KD, ND;        // reduction / output channels
  int mirror;        // dgrad: use tap T-1-t of the weights
};

template <int BM, int BN, bool B_KC>
__global__ __launch_bounds__(256) void conv_kernel(ConvP p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A4 = BM * LOTUS_BK / 4 / 256, B4 = BN * LOTUS_BK / 4 / 256;
  __shared__ float As[LdsTile<BM, true>::kFloats];
  __shared__ float Bs[LdsTile<BN, B_KC>::kFloats];
  __shared__ int prow_s[BM];
  __shared__ int n_active_s;
  extern __shared__ int dyn_s[];  // nb_s[T][BM], tapmask[T], active[T]
  int* nb_s = dyn_s;
  int* tapmask_s = dyn_s + p.T * BM;
  int* active_s = tapmask_s + p.T;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int wr0 = (wave >> 1) * (BM / 2), wc0 = (wave & 1) * (BN / 2);

  for (int r = tid; r < BM; r += 256) {
    const int m = m0 + r;
    prow_s[r] = m < p.n ? (p.rowidx ? p.rowidx[m] : m) : -1;
  }
  __syncthreads();
  // neighbour slab + per-tap activity of each 32-row group (wave ballot)
  for (int t = wave; t < p.T; t += 4) {
    int mask = 0;
    for (int r0 = 0; r0 < BM; r0 += 64) {
      const int pr = prow_s[r0 + lane];
      const int nb = pr >= 0 ? p.nbr[(long)t * p.n + pr] : -1;
      nb_s[t * BM + r0 + lane] = nb;
      const unsigned long long b = __ballot(nb >= 0);
      if (b & 0xffffffffull) mask |= 1 << (r0 / 32);
      if (b >> 32) mask |= 1 << (r0 / 32 + 1);
    }
    if (lane == 0) tapmask_s[t] = mask;
  }
  __syncthreads();
  if (tid == 0) {
    int c = 0;
    for (int t = 0; t < p.T; ++t)
      if (tapmask_s[t]) active_s[c++] = t;
    n_active_s = c;
  }
  __syncthreads();
  const int n_active = n_active_s;
  const int kchunks = (p.KD + LOTUS_BK - 1) / LOTUS_BK;
  const int iters = n_active * kchunks;
  const long wld = (long)p.T * p.cin;
  const bool a_vec = (p.KD % 4 == 0) && (((uintptr_t)p.x) % 16 == 0);
  const bool b_vec = B_KC ? ((p.cin % 4 == 0) && (((uintptr_t)p.w) % 16 == 0))
                          : ((p.cin % 4 == 0) && (((uintptr_t)p.w) % 16 == 0));

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float asum_unused[TM];

  float4 ra[A4], rb[B4];
  auto gload = [&](int it) {
    const int t = active_s[it / kchunks];
    const int k0 = (it % kchunks) * LOTUS_BK;
#pragma unroll
    for (int q = 0; q < A4; ++q) {
      const int f = tid + q * 256;
      const int row = f / (LOTUS_BK / 4), kq = f % (LOTUS_BK / 4);
      const int src = nb_s[t * BM + row];
      ra[q] = src >= 0 ? load4_guard(p.x, p.KD, src, k0 + kq * 4, p.n, p.KD, a_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int tw = p.mirror ? (p.T - 1 - t) : t;
    const float* wb = p.w + (long)tw * p.cin;
#pragma unroll
    for (int q = 0; q < B4; ++q) {
      const int f = tid + q * 256;
      if (B_KC) {  // B(k, j) = w[j][tw][k]
        const int row = f / (LOTUS_BK / 4), kq = f % (LOTUS_BK / 4);
        rb[q] = load4_guard(wb, wld, n0 + row, k0 + kq * 4, p.ND, p.KD, b_vec);
      } else {  // B(k, j) = w[k][tw][j]
        const int kr = f / (BN / 4), jq = f % (BN / 4);
        rb[q] = load4_guard(wb, wld, k0 + kr, n0 + jq * 4, p.KD, p.ND, b_vec);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int q = 0; q < A4; ++q) {
      const int f = tid + q * 256;
      const int row = f / (LOTUS_BK / 4), kq = f % (LOTUS_BK / 4);
      const float v[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) As[LdsTile<BM, true>::idx(row, kq * 4 + e)] = v[e];
    }
#pragma unroll
    for (int q = 0; q < B4; ++q) {
      const int f = tid + q * 256;
      const float v[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
      if (B_KC) {
        const int row = f / (LOTUS_BK / 4), kq = f % (LOTUS_BK / 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) Bs[LdsTile<BN, true>::idx(row, kq * 4 + e)] = v[e];
      } else {
        const int kr = f / (BN / 4), jq = f % (BN / 4);
        st4(&Bs[kr * BN + jq * 4], rb[q]);
      }
    }
  };

  if (iters > 0) {
    gload(0);
    lstore();
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      const bool more = it + 1 < iters;
      if (more) gload(it + 1);
      const int t = active_s[it / kchunks];
      const unsigned tm_mask = ((unsigned)tapmask_s[t] >> ((wave >> 1) * TM)) & ((1u << TM) - 1);
      if (tm_mask) mma_slab<BM, BN, true, B_KC, false>(As, Bs, wr0, wc0, acc, asum_unused, tm_mask);
      __syncthreads();
      if (more) {
        lstore();
        __syncthreads();
      }
    }
  }

#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = n0 + acc_col(wc0, tn);
      if (col >= p.ND) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pr = prow_s[acc_row(wr0, tm, r)];
        if (pr < 0) continue;
        const long o = (long)pr * p.ND + col;
        float v = acc[tm][tn][r] + bv;
        if (p.add) v += p.add[o];
        p.y[o] = v;
      }
    }
}

// ------------------------------------------------------------------------------------ wgrad
struct ConvWgP {
  const act_t* dy;  // [n][cout]
  const act_t* x;   // [n][cin]
  const int* nbr;   // [T][n]
  float* part;      // [nsplit][cout][T][cin]
  float* bias_part; // slice z at bias_part + z * part_stride, or null
  long part_stride; // floats between split slabs
  int n, T, cin, cout, chunk;  // chunk = points per split (<= WG_MAX_PAIRS)
};
#define WG_MAX_PAIRS 2048

__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvWgP p) {
  constexpr int BM = 64, BN = 64;
  __shared__ float As[LOTUS_BK * BM];
  __shared__ float Bs[LOTUS_BK * BN];
  __shared__ int pair_p[WG_MAX_PAIRS + LOTUS_BK];
  __shared__ int pair_q[WG_MAX_PAIRS + LOTUS_BK];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_ci = (p.cin + BN - 1) / BN;
  const int co0 = (blockIdx.x / tiles_ci) * BM, ci0 = (blockIdx.x % tiles_ci) * BN;
  const int t = blockIdx.y;
  const int p0 = blockIdx.z * p.chunk, p1 = min(p.n, p0 + p.chunk);
  const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;

  // ordered compaction of the active (p, q = nbr[t][p]) pairs of this split: the neighbour ids of all its 256-row
  // windows are loaded up front (one memory latency for the split instead of one per window plus three barriers each)
  constexpr int NW = WG_MAX_PAIRS / 256;
  __shared__ int wcnt[NW][4];
  int total_r = 0;
  {
    int qv[NW];
    unsigned long long bv[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int pp = p0 + w * 256 + tid;
      qv[w] = pp < p1 ? p.nbr[(long)t * p.n + pp] : -1;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      bv[w] = __ballot(qv[w] >= 0);
      if (lane == 0) wcnt[w][wave] = __popcll(bv[w]);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int off = total_r;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < wave) off += wcnt[w][k];
        total_r += wcnt[w][k];
      }
      if (qv[w] >= 0) {
        const int rank = __popcll(bv[w] & ((1ull << lane) - 1ull));
        pair_p[off + rank] = p0 + w * 256 + tid;
        pair_q[off + rank] = qv[w];
      }
    }
  }
  __syncthreads();
  const int total_s = total_r;
  const int total = total_s;
  for (int i = total + tid; i < total + LOTUS_BK; i += 256) {
    pair_p[i] = -1;
    pair_q[i] = -1;
  }
  __syncthreads();

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  float asum[1] = {0.f};
  const bool a_vec = (p.cout % 4 == 0) && (((uintptr_t)p.dy) % 16 == 0);
  const bool b_vec = (p.cin % 4 == 0) && (((uintptr_t)p.x) % 16 == 0);
  const int kr = tid / 16, cq = tid % 16;  // one float4 per thread per operand: row kr, cols cq*4..

  float4 ra, rb;
  auto gload = [&](int k0) {
    const int pp = pair_p[k0 + kr], qq = pair_q[k0 + kr];
    ra = pp >= 0 ? load4_guard(p.dy, p.cout, pp, co0 + cq * 4, p.n, p.cout, a_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    rb = qq >= 0 ? load4_guard(p.x, p.cin, qq, ci0 + cq * 4, p.n, p.cin, b_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto lstore = [&]() {
    st4(&As[kr * BM + cq * 4], ra);
    st4(&Bs[kr * BN + cq * 4], rb);
  };
  if (total > 0) {
    gload(0);
    lstore();
    __syncthreads();
    for (int k0 = 0; k0 < total; k0 += LOTUS_BK) {
      const bool more = k0 + LOTUS_BK < total;
      if (more) gload(k0 + LOTUS_BK);
      mma_slab<BM, BN, false, false, true>(As, Bs, wr0, wc0, acc, asum);
      __syncthreads();
      if (more) {
        lstore();
        __syncthreads();
      }
    }
  }
  float* part = p.part + (long)blockIdx.z * p.part_stride;
  const int col = ci0 + acc_col(wc0, 0);
  if (col < p.cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = co0 + acc_row(wr0, 0, r);
      if (row < p.cout) part[((long)row * p.T + t) * p.cin + col] = acc[0][0][r];
    }
  }
  // bias gradient = column sums of dy over all points: the centre tap is active for every point
  if (p.bias_part && t == p.T / 2 && ci0 == 0 && (wave & 1) == 0) {
    const float s = asum[0] + __shfl_xor(asum[0], 32, 64);
    const int i = co0 + wr0 + (tid & 31);
    if ((tid & 32) == 0 && i < p.cout) p.bias_part[(long)blockIdx.z * p.part_stride + i] = s;
  }
}

// bf16 operand variants of the weight gradient (precision 1: bf16 operands; 3: bf16x3 split — the operand modes of the
// dense layers, gemm.hip): same decomposition and pair compaction, 32 pairs per slab; the gathered dy / x rows are
// converted (and split) while they are staged into k-contiguous bf16 images (mma.h BTile, pair index = k), the products
// are v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The bias gradient is summed from the fp32 staging registers.
__device__ __forceinline__ float4 zsel4(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

template <int PREC>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(ConvWgP p) {
  constexpr int BM = 64, BN = 64, BK = 32;
  using TA = BTile<BM, BK, PREC>;
  using TB = BTile<BN, BK, PREC>;
  __shared__ __attribute__((aligned(16))) unsigned Aw[TA::kWords];
  __shared__ __attribute__((aligned(16))) unsigned Bw[TB::kWords];
  __shared__ int pair_p[WG_MAX_PAIRS + BK];
  __shared__ int pair_q[WG_MAX_PAIRS + BK];
  static_assert(TA::kWords * 4 >= 256 * 16, "bias reduction reuses the A image");

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tiles_ci = (p.cin + BN - 1) / BN;
  const int co0 = (blockIdx.x / tiles_ci) * BM, ci0 = (blockIdx.x % tiles_ci) * BN;
  const int t = blockIdx.y;
  const int p0 = blockIdx.z * p.chunk, p1 = min(p.n, p0 + p.chunk);
  const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;

  constexpr int NW = WG_MAX_PAIRS / 256;
  __shared__ int wcnt[NW][4];
  int total = 0;
  {  // ordered compaction of the active pairs of this split (as conv_wgrad_kernel)
    int qv[NW];
    unsigned long long bv[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int pp = p0 + w * 256 + tid;
      qv[w] = pp < p1 ? p.nbr[(long)t * p.n + pp] : -1;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      bv[w] = __ballot(qv[w] >= 0);
      if (lane == 0) wcnt[w][wave] = __popcll(bv[w]);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int off = total;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < wave) off += wcnt[w][k];
        total += wcnt[w][k];
      }
      if (qv[w] >= 0) {
        const int rank = __popcll(bv[w] & ((1ull << lane) - 1ull));
        pair_p[off + rank] = p0 + w * 256 + tid;
        pair_q[off + rank] = qv[w];
      }
    }
  }
  __syncthreads();
  for (int i = total + tid; i < total + BK; i += 256) {
    pair_p[i] = -1;
    pair_q[i] = -1;
  }
  __syncthreads();

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const bool a_vec = (p.cout % 4 == 0) && (((uintptr_t)p.dy) % 16 == 0);
  const bool b_vec = (p.cin % 4 == 0) && (((uintptr_t)p.x) % 16 == 0);
  const int kr2 = tid / 16, iq = tid % 16;  // pairs 2 kr2, 2 kr2 + 1 of the slab; channels 4 iq .. 4 iq + 3 of the tile
  const bool want_bias = p.bias_part && t == p.T / 2 && ci0 == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // staging registers hold the rows AS LOADED, every load is issued unconditionally (padding pairs read row 0 and are
  // zeroed by a select when staged): a conversion or a branch at the load site makes the compiler wait for the load
  // there, in front of the products it should fly under (gemm.hip, conv_os_kernel)
  Raw4<act_t> rra[2], rrb[2];
  bool oka[2], okb[2];
  const bool a_fast = a_vec && co0 + iq * 4 + 3 < p.cout, b_fast = b_vec && ci0 + iq * 4 + 3 < p.cin;
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pp = pair_p[k0 + 2 * kr2 + e], qq = pair_q[k0 + 2 * kr2 + e];
      oka[e] = pp >= 0;
      okb[e] = qq >= 0;
      if (a_fast) rra[e] = ldraw(p.dy + (long)max(pp, 0) * p.cout + co0 + iq * 4);
      else rra[e] = toraw(load4_guard(p.dy, p.cout, max(pp, 0), co0 + iq * 4, p.n, p.cout, a_vec), p.dy);
      if (b_fast) rrb[e] = ldraw(p.x + (long)max(qq, 0) * p.cin + ci0 + iq * 4);
      else rrb[e] = toraw(load4_guard(p.x, p.cin, max(qq, 0), ci0 + iq * 4, p.n, p.cin, b_vec), p.x);
    }
  };
  auto stage = [&](unsigned* W, int plane, const float4& v0, const float4& v1, bool sum) __attribute__((always_inline)) {
    const float va[4] = {v0.x, v0.y, v0.z, v0.w}, vb[4] = {v1.x, v1.y, v1.z, v1.w};
    unsigned hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_bf16(va[e], vb[e], hw[e], lw[e]);
    if (sum) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bsum[e] += va[e] + vb[e];
    }
    const int rot = (iq >> 2) & 3;  // rotated row order: the 64 lanes of a store hit distinct banks (gemm.hip)
    rot4(hw, rot);
    if (PREC == 3) rot4(lw, rot);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = iq * 4 + ((e + rot) & 3);
      W[row * TA::kLdW + kr2] = hw[e];
      if (PREC == 3) W[plane + row * TA::kLdW + kr2] = lw[e];
    }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
    stage(Aw, TA::kPlaneW, zsel4(oka[0], unraw(rra[0])), zsel4(oka[1], unraw(rra[1])), want_bias);
    stage(Bw, TB::kPlaneW, zsel4(okb[0], unraw(rrb[0])), zsel4(okb[1], unraw(rrb[1])), false);
  };
  if (total > 0) {
    gload(0);
    lstore();
    __syncthreads();
    for (int k0 = 0; k0 < total; k0 += BK) {
      const bool more = k0 + BK < total;
      if (more) gload(k0 + BK);
      gemm_slab_bf16<BM, BN, BK, PREC>(Aw, Bw, wr0, wc0, acc);
      __syncthreads();
      if (more) {
        lstore();
        __syncthreads();
      }
    }
  }
  float* part = p.part + (long)blockIdx.z * p.part_stride;
  const int col = ci0 + acc_col(wc0, 0);
  if (col < p.cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = co0 + acc_row(wr0, 0, r);
      if (row < p.cout) part[((long)row * p.T + t) * p.cin + col] = acc[0][0][r];
    }
  }
  if (want_bias) {  // fixed-order sum of the 16 thread groups that staged the same channel quad
    float4* red = reinterpret_cast<float4*>(Aw);
    red[tid] = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
    __syncthreads();
    if (tid < BM) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float4 v = red[g * 16 + (tid >> 2)];
        s += (tid & 3) == 0 ? v.x : (tid & 3) == 1 ? v.y : (tid & 3) == 2 ? v.z : v.w;
      }
      if (co0 + tid < p.cout) p.bias_part[(long)blockIdx.z * p.part_stride + co0 + tid] = s;
    }
  }
}

static size_t conv_dyn_lds(int T, int BM) { return (size_t)(T * BM + 2 * T) * sizeof(int); }

extern "C" {

size_t lotus_subm_conv_workspace(int n, int cin, int cout) {
  const size_t a = lotus_conv_pairs_workspace(n, cout), b = lotus_conv_pairs_workspace(n, cin);
  const size_t c = tap_shape_ok(n, cin, cout) ? tap_part_bytes(n, cin > cout ? cin : cout) : 0;
  const size_t ab = a > b ? a : b;
  return ab > c ? ab : c;
}
// 1 when lotus_subm_conv takes the tap-grouped path for this shape if it is handed a tap plan (lotus_fe_tap_plan)
int lotus_conv_tap_eligible(int n, int cin, int cout) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("LOTUS_CONV_TAP"); on = (e && e[0] == '0') ? 0 : 1; }
  return on && tap_shape_ok(n, cin, cout) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Stem convolution (5^3 taps, cin = 6..8 -> cout = 64): 125 taps x 7 channels is far too thin for MFMA tiles
// (the dense gather-GEMM above spends > 80 % of its MFMAs on absent neighbours), so this one is a VALU kernel over
// the ACTIVE pairs only.  Block = 32 rows x 64 output channels; 8 lanes per row own 8 channels each.  Every row's
// active (tap, neighbour) list is compacted in tap order by prefix sums over its 8 lanes (deterministic).
#define STEM_ROWS 32
#define STEM_TP 128  // taps are scanned as 8 lanes x 16

}  // extern "C"

// wt[t][ci][c] = w[c][t][ci]
__global__ void conv_stem_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int T, int cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * T * cin) return;
  const int c = i % cout, rem = i / cout;  // rem = t * cin + ci
  wt[i] = w[(long)c * T * cin + rem];
}

// CIN: compile-time input width (0 = runtime p.cin): with it the 2 * CIN weight loads of a pair are issued together.
template <int CIN>
__global__ __launch_bounds__(256) void conv_smallcin_kernel(ConvP p) {
  __shared__ int list_nb[STEM_ROWS * STEM_TP];
  __shared__ unsigned char list_t[STEM_ROWS * STEM_TP];
  __shared__ int cnt_s[STEM_ROWS];
  const int tid = threadIdx.x, r = tid >> 3, j = tid & 7;
  const int m = blockIdx.y * STEM_ROWS + r, cb = blockIdx.x;
  const bool valid = m < p.n;
  {  // ordered compaction of this row's active taps
    int nbv[16];
    unsigned flags = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int t = 16 * j + k;
      nbv[k] = (valid && t < p.T) ? p.nbr[(long)t * p.n + m] : -1;
      flags |= (nbv[k] >= 0 ? 1u : 0u) << k;
    }
    const int mine = __popc(flags);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const int v = __shfl_up(incl, o, 8);
      if (j >= o) incl += v;
    }
    int pos = incl - mine;
    if (j == 7) cnt_s[r] = incl;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (flags & (1u << k)) {
        list_t[r * STEM_TP + pos] = (unsigned char)(16 * j + k);
        list_nb[r * STEM_TP + pos] = nbv[k];
        ++pos;
      }
  }
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = p.bias ? p.bias[cb * 64 + j * 8 + k] : 0.f;
  __syncthreads();
  const int len = cnt_s[r];
  // pair loop: the weights (pre-transposed [T][cin][cout], 224 KB for the stem) stay in L2 — a lane's 8 channels
  // are two 16-byte loads, the 8 lanes of a row read 256 contiguous bytes — so the kernel needs no weight staging,
  // no chunk barriers and only 20 KB of LDS: occupancy, not a software pipeline, hides the load latency.
  // Lane j fetches input channel j of the neighbour row; the row's 8 lanes share it through 8-wide shuffles.
  const float* wbase = p.w + cb * 64 + j * 8;
  const int cin = CIN ? CIN : p.cin;
  // Every global load of a pair used to sit behind the previous one (the neighbour's x, then one weight row per input
  // channel: ~8 serialised L2 latencies per pair, 23 pairs per row).  Now the next pair's tap / x are fetched while this
  // pair's weight rows — all issued at once — are in flight.  The FMA order per accumulator is unchanged.
  int t = 0;
  float xl = 0.f;
  if (len > 0) {
    t = list_t[r * STEM_TP];
    xl = j < cin ? p.x[(long)list_nb[r * STEM_TP] * cin + j] : 0.f;
  }
  for (int cur = 0; cur < len; ++cur) {
    const float* wr = wbase + (long)t * cin * p.cout;
    const float xc = xl;
    if (cur + 1 < len) {
      t = list_t[r * STEM_TP + cur + 1];
      xl = j < cin ? p.x[(long)list_nb[r * STEM_TP + cur + 1] * cin + j] : 0.f;
    }
    if (CIN) {
      float4 w0[CIN ? CIN : 1], w1[CIN ? CIN : 1];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        w0[ci] = ld4(wr + ci * p.cout);
        w1[ci] = ld4(wr + ci * p.cout + 4);
      }
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float xv = __shfl(xc, ci, 8);
        acc[0] = fmaf(xv, w0[ci].x, acc[0]); acc[1] = fmaf(xv, w0[ci].y, acc[1]); acc[2] = fmaf(xv, w0[ci].z, acc[2]); acc[3] = fmaf(xv, w0[ci].w, acc[3]);
        acc[4] = fmaf(xv, w1[ci].x, acc[4]); acc[5] = fmaf(xv, w1[ci].y, acc[5]); acc[6] = fmaf(xv, w1[ci].z, acc[6]); acc[7] = fmaf(xv, w1[ci].w, acc[7]);
      }
    } else {
      for (int ci = 0; ci < cin; ++ci) {
        const float4 w0 = ld4(wr + ci * p.cout);
        const float4 w1 = ld4(wr + ci * p.cout + 4);
        const float xv = __shfl(xc, ci, 8);
        acc[0] = fmaf(xv, w0.x, acc[0]); acc[1] = fmaf(xv, w0.y, acc[1]); acc[2] = fmaf(xv, w0.z, acc[2]); acc[3] = fmaf(xv, w0.w, acc[3]);
        acc[4] = fmaf(xv, w1.x, acc[4]); acc[5] = fmaf(xv, w1.y, acc[5]); acc[6] = fmaf(xv, w1.z, acc[6]); acc[7] = fmaf(xv, w1.w, acc[7]);
      }
    }
  }
  if (valid) {
    act_t* yo = p.y + (long)m * p.ND + cb * 64 + j * 8;
    if (p.add) {
      const act_t* ao = p.add + (long)m * p.ND + cb * 64 + j * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += ao[k];
    }
    st4(yo, make_float4(acc[0], acc[1], acc[2], acc[3]));
    st4(yo + 4, make_float4(acc[4], acc[5], acc[6], acc[7]));
  }
}

extern "C" {

// w_t [2][cout*T*cin]: MFMA-fragment-packed copies of w for the forward and the input-gradient direction
// (layout: conv_pairs.hip); cin and cout must be multiples of 32
int lotus_conv_weight_transpose(const float* w, float* w_t, int cout, int T, int cin, int precision, void* stream) {
  LOTUS_CHECK_ARG(w && w_t && cout > 0 && T > 0 && cin > 0, "lotus_conv_weight_transpose: bad arguments");
  LOTUS_CHECK_ARG(precision == 0 || precision == 1 || precision == 3, "lotus_conv_weight_transpose: precision must be 0, 1 or 3");
  return lotus_conv_weight_transpose_impl(w, w_t, cout, T, cin, precision, (hipStream_t)stream);
}

// mode 0: fwd  (x [n][cin]  -> y [n][cout]);  mode 1: dgrad (x = dy [n][cout] -> y = dx [n][cin]).
// w_t (optional) and workspace (optional) enable the pair-compacted tap-split fast path in both modes.
int lotus_subm_conv(int mode, const act_t* x, const float* w, const float* w_t, const float* bias, const act_t* add,
                    act_t* y, const int* nbr, const int* rowidx, int n, int T, int cin, int cout, int precision,
                    const int* tap_plan, void* workspace, size_t workspace_bytes, void* stream) {
  LOTUS_CHECK_ARG(x && w && y && nbr && n >= 0 && T > 0 && cin > 0 && cout > 0, "lotus_subm_conv: bad arguments");
  LOTUS_CHECK_ARG(precision == 0 || precision == 1 || precision == 3, "lotus_subm_conv: precision must be 0, 1 or 3");
  if (n == 0) return LOTUS_OK;
  if (tap_plan && T == 27 && precision == 0 && lotus_conv_tap_eligible(n, cin, cout)) {
    // A caller that hands over a tap plan for an eligible shape has skipped the packed weights (ops.conv_tap_active): falling
    // through to the generic kernel would be a silent 10x slowdown, so the preconditions of the path are hard errors.
    LOTUS_CHECK_ARG(workspace && ((uintptr_t)workspace) % 16 == 0 && workspace_bytes >= tap_part_bytes(n, mode == 0 ? cout : cin),
                    "lotus_subm_conv: the tap-grouped path needs a 16-byte aligned workspace of lotus_subm_conv_workspace(n, cin, cout) bytes");
    LOTUS_CHECK_ARG((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)bias) | ((uintptr_t)add)) % 16 == 0,
                    "lotus_subm_conv: the tap-grouped path needs 16-byte aligned operands");
    const int n64 = (n + 63) / 64 * 64, ND = mode == 0 ? cout : cin;
    const int rc = lotus_conv_tap_gemm(mode, x, w, (float*)workspace, tap_plan + 32, tap_plan, n64, n, cin, cout, (hipStream_t)stream);
    if (rc != LOTUS_OK) return rc;
    const long total4 = (long)n * ND / 4;
    const int g = (int)cdiv(total4, 256);
    LOTUS_LAUNCH(conv_tap_reduce_kernel, dim3(g > 4096 ? 4096 : g), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                 tap_plan + 32 + 27L * n64, n, ND, bias, add, y);
    LOTUS_LAUNCH_CHECK("lotus_subm_conv(tap-grouped)");
    return LOTUS_OK;
  }
  {
    int rc = 0;  // pair-compacted fast path (conv_pairs.hip) for the 3^3 CPE convolutions
    if (lotus_conv_pairs_try(mode, x, w, w_t, bias, add, y, nbr, rowidx, n, T, cin, cout, workspace, workspace_bytes,
                             precision, (hipStream_t)stream, &rc))
      return rc;
  }
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.y = y; p.bias = bias; p.add = add; p.nbr = nbr; p.rowidx = rowidx;
  p.n = n; p.T = T; p.cin = cin; p.cout = cout;
  p.KD = mode == 0 ? cin : cout;
  p.ND = mode == 0 ? cout : cin;
  p.mirror = mode == 1;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0 && cin <= 8 && cout % 64 == 0 && T <= STEM_TP && ((uintptr_t)y) % 16 == 0 && workspace &&
      ((uintptr_t)workspace) % 16 == 0 && workspace_bytes >= (size_t)T * cin * cout * sizeof(float)) {
    // thin-input stem: VALU kernel over the active pairs (rows are independent, so the processing order is moot);
    // the workspace receives the weights transposed to [T][cin][cout]
    LOTUS_LAUNCH(conv_stem_wt_kernel, dim3(cdiv((long)cout * T * cin, 256)), dim3(256), 0, st, w, (float*)workspace, cout, T,
                       cin);
    p.w = (const float*)workspace;
    const dim3 sg(cout / 64, cdiv(n, STEM_ROWS));
    if (cin == 7) LOTUS_LAUNCH((conv_smallcin_kernel<7>), sg, dim3(256), 0, st, p);
    else if (cin == 8) LOTUS_LAUNCH((conv_smallcin_kernel<8>), sg, dim3(256), 0, st, p);
    else if (cin == 6) LOTUS_LAUNCH((conv_smallcin_kernel<6>), sg, dim3(256), 0, st, p);
    else if (cin == 4) LOTUS_LAUNCH((conv_smallcin_kernel<4>), sg, dim3(256), 0, st, p);
    else LOTUS_LAUNCH((conv_smallcin_kernel<0>), sg, dim3(256), 0, st, p);
    LOTUS_LAUNCH_CHECK("lotus_subm_conv(stem)");
    return LOTUS_OK;
  }
  dim3 block(256);
  const size_t dyn = conv_dyn_lds(T, 128);
  LOTUS_CHECK_ARG(dyn + 20000 <= 160 * 1024, "lotus_subm_conv: %d taps do not fit LDS", T);
  if (p.ND <= 64) {
    dim3 grid(cdiv(p.ND, 64), cdiv(n, 128));
    if (mode == 0) {
      { static bool a1 = false; if (!a1) { (void)hipFuncSetAttribute((const void*)conv_kernel<128, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); a1 = true; } }
      LOTUS_LAUNCH((conv_kernel<128, 64, true>), grid, block, dyn, st, p);
    } else {
      { static bool a2 = false; if (!a2) { (void)hipFuncSetAttribute((const void*)conv_kernel<128, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); a2 = true; } }
      LOTUS_LAUNCH((conv_kernel<128, 64, false>), grid, block, dyn, st, p);
    }
  } else {
    dim3 grid(cdiv(p.ND, 128), cdiv(n, 128));
    if (mode == 0) {
      { static bool a3 = false; if (!a3) { (void)hipFuncSetAttribute((const void*)conv_kernel<128, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); a3 = true; } }
      LOTUS_LAUNCH((conv_kernel<128, 128, true>), grid, block, dyn, st, p);
    } else {
      { static bool a4 = false; if (!a4) { (void)hipFuncSetAttribute((const void*)conv_kernel<128, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); a4 = true; } }
      LOTUS_LAUNCH((conv_kernel<128, 128, false>), grid, block, dyn, st, p);
    }
  }
  LOTUS_LAUNCH_CHECK("lotus_subm_conv");
  return LOTUS_OK;
}

// Weight gradient of the thin-input stem (cin <= 8): dw[c][t][ci] += sum over the active pairs of tap t of
// dy[row][c] * x[nbr][ci].  Block = (64 output channels, tap t, row range z); the pairs of a 256-row window are
// compacted in row order (wave ballots), their input rows staged in LDS; wave g takes pairs g, g+4, ... and the four
// waves are summed in fixed order -> deterministic.  Same partial-slab layout as conv_wgrad_kernel.
static __global__ __launch_bounds__(256) void conv_smallcin_wgrad_kernel(ConvWgP p) {
  // All pairs of the block's row range (<= WG_MAX_PAIRS rows) are compacted first — the neighbour ids of every
  // 256-row window are loaded up front, so the windows cost one memory latency together instead of one each — then
  // the list is consumed in batches of SB pairs: gather their input rows into LDS, accumulate with 8 dy loads in
  // flight per lane.  (Per 256-row window: load ids, compact, gather, accumulate, each behind a barrier, left the
  // kernel at 213 us for 1.4 GFLOP; it runs alone at the very end of backward, so its time is step time.)  Pair order
  // = row order, wave g takes pairs g, g + 4, ... of the list and the four waves are summed in fixed order.
  constexpr int NW = WG_MAX_PAIRS / 256, SB = 512;
  __shared__ int prow_s[WG_MAX_PAIRS];
  __shared__ int pnb_s[WG_MAX_PAIRS];
  __shared__ __attribute__((aligned(16))) float xs[SB][8];
  __shared__ float red[4][64][8];
  __shared__ int wcnt[NW][4];
  const int tid = threadIdx.x, c = tid & 63, g = tid >> 6, lane = tid & 63;
  const int cb = blockIdx.x, t = blockIdx.y, z = blockIdx.z;
  const int rbeg = z * p.chunk, rend = min(p.n, rbeg + p.chunk);
  int total = 0;
  {
    int nbv[NW];
    unsigned long long mv[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int row = rbeg + w * 256 + tid;
      nbv[w] = row < rend ? p.nbr[(long)t * p.n + row] : -1;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      mv[w] = __ballot(nbv[w] >= 0);
      if (lane == 0) wcnt[w][g] = __popcll(mv[w]);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int base = total;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < g) base += wcnt[w][q];
        total += wcnt[w][q];
      }
      if (nbv[w] >= 0) {
        const int pos = base + __popcll(mv[w] & ((1ull << lane) - 1ull));
        prow_s[pos] = rbeg + w * 256 + tid;
        pnb_s[pos] = nbv[w];
      }
    }
  }
  __syncthreads();
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const act_t* dyc = p.dy + cb * 64 + c;
  for (int b0 = 0; b0 < total; b0 += SB) {
    const int nb = min(SB, total - b0);
#pragma unroll 4
    for (int i = tid; i < nb * 8; i += 256) {
      const int pi = i >> 3, ci = i & 7;
      xs[pi][ci] = ci < p.cin ? p.x[(long)pnb_s[b0 + pi] * p.cin + ci] : 0.f;
    }
    __syncthreads();
    int pi = g;
    for (; pi + 28 < nb; pi += 32) {  // 8 pairs of this wave per trip: their dy loads are issued together
      float dyv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) dyv[u] = dyc[(long)prow_s[b0 + pi + 4 * u] * p.cout];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 xa = ld4(&xs[pi + 4 * u][0]);
        const float4 xb = ld4(&xs[pi + 4 * u][4]);
        acc[0] = fmaf(dyv[u], xa.x, acc[0]); acc[1] = fmaf(dyv[u], xa.y, acc[1]); acc[2] = fmaf(dyv[u], xa.z, acc[2]); acc[3] = fmaf(dyv[u], xa.w, acc[3]);
        acc[4] = fmaf(dyv[u], xb.x, acc[4]); acc[5] = fmaf(dyv[u], xb.y, acc[5]); acc[6] = fmaf(dyv[u], xb.z, acc[6]); acc[7] = fmaf(dyv[u], xb.w, acc[7]);
      }
    }
    for (; pi < nb; pi += 4) {
      const float dyv = dyc[(long)prow_s[b0 + pi] * p.cout];
      const float4 xa = ld4(&xs[pi][0]);
      const float4 xb = ld4(&xs[pi][4]);
      acc[0] = fmaf(dyv, xa.x, acc[0]); acc[1] = fmaf(dyv, xa.y, acc[1]); acc[2] = fmaf(dyv, xa.z, acc[2]); acc[3] = fmaf(dyv, xa.w, acc[3]);
      acc[4] = fmaf(dyv, xb.x, acc[4]); acc[5] = fmaf(dyv, xb.y, acc[5]); acc[6] = fmaf(dyv, xb.z, acc[6]); acc[7] = fmaf(dyv, xb.w, acc[7]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[g][c][k] = acc[k];
  __syncthreads();
  if (g == 0) {
    float* out = p.part + (long)z * p.part_stride + ((long)(cb * 64 + c) * p.T + t) * p.cin;
    for (int k = 0; k < p.cin; ++k) out[k] = ((red[0][c][k] + red[1][c][k]) + red[2][c][k]) + red[3][c][k];
  }
}

// ---- duplicate voxels: see include/lotus_hip.h.  One thread per (sorted position, float4 column).
static __global__ void conv_dup_fold_kernel(const act_t* __restrict__ dy, const long long* __restrict__ code0,
                                     const int* __restrict__ order0, int n, int c4n, act_t* __restrict__ dyr) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(id / c4n), c4 = (int)(id % c4n);
  if (i >= n) return;
  const int row = order0[i];
  const long long key = code0[row];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i == 0 || code0[order0[i - 1]] != key) {  // representative: lowest index of its voxel (stable sort)
    acc = ld4q(dy, (long)row * c4n + c4);
    for (int j = i + 1; j < n; ++j) {
      const int rj = order0[j];
      if (code0[rj] != key) break;
      const float4 v = ld4q(dy, (long)rj * c4n + c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  st4q(dyr, (long)row * c4n + c4, acc);
}
static __global__ void conv_dup_mask_kernel(act_t* __restrict__ dx, const act_t* __restrict__ add, const int* __restrict__ rep,
                                     int n, int c4n) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(id / c4n), c4 = (int)(id % c4n);
  if (i >= n || rep[i] == i) return;
  st4q(dx, id, add ? ld4q(add, id) : make_float4(0.f, 0.f, 0.f, 0.f));
}

int lotus_conv_dup_fold(const act_t* dy, const long long* code0, const int* order0, int n, int C, act_t* dyr,
                        void* stream) {
  LOTUS_CHECK_ARG(dy && code0 && order0 && dyr && n >= 0 && C > 0 && C % 4 == 0, "lotus_conv_dup_fold: bad arguments");
  if (n == 0) return LOTUS_OK;
  const long total = (long)n * (C / 4);
  LOTUS_LAUNCH(conv_dup_fold_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, code0, order0, n,
                     C / 4, dyr);
  LOTUS_LAUNCH_CHECK("lotus_conv_dup_fold");
  return LOTUS_OK;
}
int lotus_conv_dup_mask(act_t* dx, const act_t* add, const int* rep, int n, int C, void* stream) {
  LOTUS_CHECK_ARG(dx && rep && n >= 0 && C > 0 && C % 4 == 0, "lotus_conv_dup_mask: bad arguments");
  if (n == 0) return LOTUS_OK;
  const long total = (long)n * (C / 4);
  LOTUS_LAUNCH(conv_dup_mask_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dx, add, rep, n, C / 4);
  LOTUS_LAUNCH_CHECK("lotus_conv_dup_mask");
  return LOTUS_OK;
}

// points per split of the weight-gradient kernels (LOTUS_CONV_WG_CHUNK, <= WG_MAX_PAIRS): a block's lifetime — the centre
// tap pairs EVERY point of its split — against the number of partial slabs
static int wg_chunk() {
  static int c = 0;
  // measured in the step (one box, +-0.1 %): 2048 / 1024 / 512 points -> 924.2 / 927.3 / 923.9 samples/s
  if (!c) { const char* e = getenv("LOTUS_CONV_WG_CHUNK"); c = e ? atoi(e) : 1024; if (c < 256 || c > WG_MAX_PAIRS || c % 256) c = 1024; }
  return c;
}

size_t lotus_subm_conv_wgrad_workspace(int n, int T, int cin, int cout) {
  const int nsplit = cdiv(n > 0 ? n : 1, wg_chunk());
  return (size_t)nsplit * ((size_t)cout * T * cin + cout) * sizeof(float);
}

// dw [cout][T][cin] (+)= sum_p dy[p] (x) x[nbr[t][p]] ;  db [cout] (+)= colsum(dy)
int lotus_subm_conv_wgrad(const act_t* dy, const act_t* x, float* dw, float* db, const int* nbr, int n, int T,
                          int cin, int cout, int accumulate, int precision, void* workspace, size_t workspace_bytes,
                          void* stream) {
  LOTUS_CHECK_ARG(dy && x && dw && nbr && n >= 0, "lotus_subm_conv_wgrad: bad arguments");
  LOTUS_CHECK_ARG(precision == 0 || precision == 1 || precision == 3, "lotus_subm_conv_wgrad: precision must be 0, 1 or 3");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = cdiv(n > 0 ? n : 1, wg_chunk());
  const size_t wsz = (size_t)cout * T * cin;
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)nsplit * (wsz + cout) * sizeof(float),
                  "lotus_subm_conv_wgrad: workspace too small");
  const size_t slab = wsz + cout;
  const bool direct = nsplit == 1 && !accumulate;
  ConvWgP p;
  p.dy = dy; p.x = x; p.nbr = nbr;
  p.n = n; p.T = T; p.cin = cin; p.cout = cout; p.chunk = wg_chunk();
  if (direct) {
    p.part = dw; p.bias_part = db; p.part_stride = 0;
  } else {
    p.part = (float*)workspace; p.part_stride = (long)slab;
    p.bias_part = db ? p.part + wsz : nullptr;
  }
  if (cin <= 8 && cout % 64 == 0 && !db) {  // thin-input stem: VALU kernel over the active pairs
    LOTUS_LAUNCH(conv_smallcin_wgrad_kernel, dim3(cout / 64, T, nsplit), dim3(256), 0, st, p);
  } else {
    dim3 grid(cdiv(cout, 64) * cdiv(cin, 64), T, nsplit);
    if (precision == 1) LOTUS_LAUNCH(conv_wgrad_bf16_kernel<1>, grid, dim3(256), 0, st, p);
    else if (precision == 3) LOTUS_LAUNCH(conv_wgrad_bf16_kernel<3>, grid, dim3(256), 0, st, p);
    else LOTUS_LAUNCH(conv_wgrad_kernel, grid, dim3(256), 0, st, p);
  }
  LOTUS_LAUNCH_CHECK("lotus_subm_conv_wgrad");
  if (direct) return LOTUS_OK;
  if (db && db == dw + wsz) return lotus_reduce_parts(p.part, dw, (long)slab, (long)slab, nsplit, accumulate, st);
  int rc = lotus_reduce_parts(p.part, dw, (long)wsz, (long)slab, nsplit, accumulate, st);
  if (!rc && db) rc = lotus_reduce_parts(p.part + wsz, db, cout, (long)slab, nsplit, accumulate, st);
  return rc;
}

}  // extern "C"

}  // namespace LOTUS_NS
