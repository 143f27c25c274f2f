// lotus-hip: pair-compacted submanifold convolution (fwd / dgrad) — the fast path of
// lotus_subm_conv for channel counts that are multiples of 32 (every Block CPE of the model).
//
// Only 7-13 of the 27 taps of a point are active (surface clouds), so an output-stationary
// gather-GEMM wastes 50-73 % of its MFMAs on zero rows.  Here each block owns a row tile of BM
// consecutive points in space-filling-curve order and, per tap, COMPACTS the active (output row,
// neighbour row) pairs with wave ballots; MFMA groups of 32 pairs are dense.  Each wave owns a
// 32-column slice of the output for its row tile, so the per-tap results can be accumulated into
// an LDS-resident output tile with plain ds_add (one owner per element, taps in fixed order ->
// deterministic, no global atomics).  Gathered rows are staged once per (tap, 32-channel chunk)
// in a double-buffered LDS image shared by the column-slice waves; weights stream from L2 straight
// into MFMA B fragments (k is permuted identically on both operands: lane-half h takes
// k = h*KC/2 + s, so every lane reads a contiguous run).
#include "mma.h"

struct ConvP2 {
  const float* x;    // [n][KD]
  const float* w;    // B operand: element (k, tap, j) at w[k * wld + tap * tapw + j]  (j contiguous)
  float* y;          // [n][ND]  (or partial slabs [nz][n][ND] when tap-split)
  const float* bias;
  const float* add;
  const int* nbr;    // [T][n]
  const int* rowidx; // [n] or null
  long wld;
  int tapw;
  int n, T, KD, ND, mirror;
  int tpz;           // taps per blockIdx.z
  long part_stride;  // floats between z slabs (0 = single slab, epilogue applies bias/add)
};

template <int NCS, int BM>
__global__ __launch_bounds__(256) void conv_pairs_kernel(ConvP2 p) {
  constexpr int NRT = 4 / NCS;          // row tiles per block
  constexpr int KC = NCS == 4 ? 32 : 16;  // reduction chunk staged per iteration
  constexpr int NW = 32 * NCS;          // output columns per block
  constexpr int TG = BM / 32;           // max pair groups per tap
  constexpr int GT = 64 * NCS;          // gather threads per row tile
  constexpr int G4 = BM * (KC / 4) / GT;  // float4 gathered per thread per iteration
  constexpr int MAXT = 27;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* out_s = smem;                               // [NRT][BM][NW]
  float* a_s = out_s + NRT * BM * NW;                // [NRT][2][BM][KC+1]
  int* src_s = (int*)(a_s + NRT * 2 * BM * (KC + 1));  // [NRT][MAXT][BM]
  int* cnt_s = src_s + NRT * MAXT * BM;              // [NRT][MAXT]
  int* act_s = cnt_s + NRT * MAXT;                   // [NRT][MAXT]
  int* nact_s = act_s + NRT * MAXT;                  // [NRT]
  int* prow_s = nact_s + 4;                          // [NRT][BM]
  unsigned char* row_s = (unsigned char*)(prow_s + NRT * BM);  // [NRT][MAXT][BM]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = tid & 31, hh = (tid >> 5) & 1;
  const int rt = wave / NCS, cs = wave % NCS;
  const int gt = tid - rt * GT;  // thread index inside the row tile's gather team
  const int n0 = blockIdx.x * NW;
  const int m0 = (blockIdx.y * NRT + rt) * BM;
  const int t_beg = blockIdx.z * p.tpz, t_end = min(p.T, t_beg + p.tpz);

  for (int i = tid; i < NRT * BM * NW; i += 256) out_s[i] = 0.f;
  for (int r = gt; r < BM; r += GT) {
    const int m = m0 + r;
    prow_s[rt * BM + r] = m < p.n ? (p.rowidx ? p.rowidx[m] : m) : -1;
  }
  __syncthreads();
  // per-tap ordered compaction of the active pairs of this row tile (wave ballot + popcount)
  for (int t = t_beg + cs; t < t_end; t += NCS) {
    int base = 0;
    for (int r0 = 0; r0 < BM; r0 += 64) {
      const int r = r0 + lane;
      const int pr = r < BM ? prow_s[rt * BM + r] : -1;
      const int nb = pr >= 0 ? p.nbr[(long)t * p.n + pr] : -1;
      const unsigned long long b = __ballot(nb >= 0);
      if (nb >= 0) {
        const int k = base + __popcll(b & ((1ull << lane) - 1ull));
        src_s[(rt * MAXT + t) * BM + k] = nb;
        row_s[(rt * MAXT + t) * BM + k] = (unsigned char)r;
      }
      base += __popcll(b);
    }
    if (lane == 0) cnt_s[rt * MAXT + t] = base;
  }
  __syncthreads();
  if (gt == 0) {
    int c = 0;
    for (int t = t_beg; t < t_end; ++t)
      if (cnt_s[rt * MAXT + t] > 0) act_s[rt * MAXT + c++] = t;
    nact_s[rt] = c;
  }
  __syncthreads();
  const int nkc = p.KD / KC;
  const int my_iters = nact_s[rt] * nkc;
  int max_iters = 0;
#pragma unroll
  for (int q = 0; q < NRT; ++q) max_iters = max(max_iters, nact_s[q] * nkc);

  f32x16 acc[TG];
#pragma unroll
  for (int g = 0; g < TG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  float4 ga[G4];
  float bcur[KC / 2], bnxt[KC / 2];
  float* my_a = a_s + rt * 2 * BM * (KC + 1);

  auto issue = [&](int it) {  // global loads of iteration `it` into registers (A rows + B fragment)
    const int t = act_s[rt * MAXT + it / nkc];
    const int k0 = (it % nkc) * KC;
    const int cnt = cnt_s[rt * MAXT + t];
#pragma unroll
    for (int q = 0; q < G4; ++q) {
      const int f = gt + q * GT;
      const int row = f / (KC / 4), kq = f % (KC / 4);
      ga[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < cnt) ga[q] = *reinterpret_cast<const float4*>(p.x + (long)src_s[(rt * MAXT + t) * BM + row] * p.KD + k0 + kq * 4);
    }
    const int tw = p.mirror ? (p.T - 1 - t) : t;
    // B(k, j): lanes j read consecutive floats (coalesced 128 B per half-wave and k)
    const float* wp = p.w + (long)(k0 + hh * (KC / 2)) * p.wld + (long)tw * p.tapw + n0 + cs * 32 + l31;
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) bnxt[s] = wp[(long)s * p.wld];
  };
  auto stage = [&](int buf) {  // gathered registers -> LDS image [BM][KC+1]
    float* dst = my_a + buf * BM * (KC + 1);
#pragma unroll
    for (int q = 0; q < G4; ++q) {
      const int f = gt + q * GT;
      const int row = f / (KC / 4), kq = f % (KC / 4);
      float* o = dst + row * (KC + 1) + kq * 4;
      o[0] = ga[q].x; o[1] = ga[q].y; o[2] = ga[q].z; o[3] = ga[q].w;
    }
  };

  if (my_iters > 0) {
    issue(0);
    stage(0);
  }
#pragma unroll
  for (int s = 0; s < KC / 2; ++s) bcur[s] = bnxt[s];
  __syncthreads();
  for (int it = 0; it < max_iters; ++it) {
    const bool live = it < my_iters;
    const bool more = it + 1 < my_iters;
    if (more) issue(it + 1);
    if (live) {
      const int t = act_s[rt * MAXT + it / nkc];
      const int cnt = cnt_s[rt * MAXT + t];
      const int ng = (cnt + 31) >> 5;
      const float* ab = my_a + (it & 1) * BM * (KC + 1);
#pragma unroll
      for (int s = 0; s < KC / 2; ++s) {
#pragma unroll
        for (int g = 0; g < TG; ++g)
          if (g < ng)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[(g * 32 + l31) * (KC + 1) + hh * (KC / 2) + s], bcur[s], acc[g], 0, 0, 0);
      }
      if ((it % nkc) == nkc - 1) {  // tap finished: fold into the LDS output tile (this wave owns its columns)
        float* ob = out_s + rt * BM * NW + cs * 32 + l31;
        const unsigned char* rows = row_s + (rt * MAXT + t) * BM;
#pragma unroll
        for (int g = 0; g < TG; ++g)
          if (g < ng) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int idx = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              if (idx < cnt) ob[(int)rows[idx] * NW] += acc[g][r];
              acc[g][r] = 0.f;
            }
          }
      }
    }
    if (more) {
      stage((it + 1) & 1);
#pragma unroll
      for (int s = 0; s < KC / 2; ++s) bcur[s] = bnxt[s];
    }
    __syncthreads();
  }
  // epilogue: coalesced row stores (final result, or this tap group's partial slab)
  const bool final_out = p.part_stride == 0;
  float* yo = p.y + (long)blockIdx.z * p.part_stride;
  for (int i = gt; i < BM * (NW / 4); i += GT) {
    const int r = i / (NW / 4), c4 = i % (NW / 4);
    const int pr = prow_s[rt * BM + r];
    if (pr < 0) continue;
    float4 v = *reinterpret_cast<const float4*>(out_s + (rt * BM + r) * NW + c4 * 4);
    const int col = n0 + c4 * 4;
    const long o = (long)pr * p.ND + col;
    if (final_out) {
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (p.add) {
        const float4 a = *reinterpret_cast<const float4*>(p.add + o);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    }
    *reinterpret_cast<float4*>(yo + o) = v;
  }
}

// y = sum_z part[z] + bias + add   (fixed order -> deterministic)
__global__ void conv_part_reduce_kernel(const float* __restrict__ part, long stride, int nz, const float* __restrict__ bias,
                                        const float* __restrict__ add, float* __restrict__ y, long total4, int nd4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(part)[i];
    for (int z = 1; z < nz; ++z) {
      const float4 v = reinterpret_cast<const float4*>(part + (long)z * stride)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      const float4 b = reinterpret_cast<const float4*>(bias)[i % nd4];
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (add) {
      const float4 a = reinterpret_cast<const float4*>(add)[i];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    reinterpret_cast<float4*>(y)[i] = s;
  }
}

// wt[ci][t][co] = w[co][t][ci]   (32x32 LDS tile transpose per tap)
__global__ __launch_bounds__(256) void conv_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int T,
                                                      int cin) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z, co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < cout && ci < cin) ? w[((long)co * T + t) * cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < cin && co < cout) wt[((long)ci * T + t) * cout + co] = tile[tx][r];
  }
}

template <int NCS, int BM>
static size_t pairs_smem() {
  constexpr int NRT = 4 / NCS, KC = NCS == 4 ? 32 : 16, NW = 32 * NCS, MAXT = 27;
  return (size_t)(NRT * BM * NW + NRT * 2 * BM * (KC + 1)) * 4 + (size_t)(NRT * MAXT * BM + 2 * NRT * MAXT + 4 + NRT * BM) * 4 +
         (size_t)NRT * MAXT * BM;
}

static int tap_splits(int n, int ND) {
  const long base = (long)cdiv(n, 128) * (ND <= 64 ? 1 : ND / 128);
  int nz = 1;
  while (nz < 9 && base * nz < 384) nz = nz == 1 ? 3 : 9;  // 27 taps -> 1, 3 or 9 groups
  return nz;
}

template <int NCS, int BM>
static int launch_pairs(ConvP2& p, int nz, hipStream_t st) {
  constexpr int NRT = 4 / NCS;
  const size_t sm = pairs_smem<NCS, BM>();
  (void)hipFuncSetAttribute((const void*)conv_pairs_kernel<NCS, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  dim3 grid(p.ND / (32 * NCS), cdiv(p.n, BM * NRT), nz);
  hipLaunchKernelGGL((conv_pairs_kernel<NCS, BM>), grid, dim3(256), sm, st, p);
  LOTUS_LAUNCH_CHECK("lotus_subm_conv(pairs)");
  return LOTUS_OK;
}

size_t lotus_conv_pairs_workspace(int n, int ND) {
  const int nz = tap_splits(n, ND);
  return nz > 1 ? (size_t)nz * n * ND * sizeof(float) : 0;
}

int lotus_conv_weight_transpose_impl(const float* w, float* wt, int cout, int T, int cin, hipStream_t st) {
  hipLaunchKernelGGL(conv_wt_kernel, dim3(cdiv(cin, 32), cdiv(cout, 32), T), dim3(256), 0, st, w, wt, cout, T, cin);
  LOTUS_LAUNCH_CHECK("lotus_conv_weight_transpose");
  return LOTUS_OK;
}

// returns 1 if the pair-compacted path handled the call, 0 if the shape is not eligible.
// mode 0 needs the transposed weights w_t [cin][T][cout]; mode 1 uses w [cout][T][cin] directly.
int lotus_conv_pairs_try(int mode, const float* x, const float* w, const float* w_t, const float* bias, const float* add,
                         float* y, const int* nbr, const int* rowidx, int n, int T, int cin, int cout, void* workspace,
                         size_t workspace_bytes, hipStream_t st, int* rc) {
  const int KD = mode == 0 ? cin : cout, ND = mode == 0 ? cout : cin;
  if (T != 27 || KD % 32 || ND % 64 || (ND > 64 && ND % 128)) return 0;
  if (mode == 0 && !w_t) return 0;
  if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)bias) | ((uintptr_t)add)) % 16) return 0;
  const int nz = tap_splits(n, ND);
  if (nz > 1 && (!workspace || workspace_bytes < (size_t)nz * n * ND * sizeof(float))) return 0;
  ConvP2 p;
  p.x = x; p.bias = bias; p.add = add; p.nbr = nbr; p.rowidx = rowidx;
  p.n = n; p.T = T; p.KD = KD; p.ND = ND; p.mirror = mode == 1;
  if (mode == 0) { p.w = w_t; p.wld = (long)T * cout; p.tapw = cout; }
  else           { p.w = w;   p.wld = (long)T * cin;  p.tapw = cin; }
  p.tpz = cdiv(T, nz);
  p.y = nz > 1 ? (float*)workspace : y;
  p.part_stride = nz > 1 ? (long)n * ND : 0;
  *rc = ND == 64 ? launch_pairs<2, 128>(p, nz, st) : launch_pairs<4, 128>(p, nz, st);
  if (*rc == 0 && nz > 1) {
    const long total4 = (long)n * ND / 4;
    int g = cdiv(total4, 256);
    hipLaunchKernelGGL(conv_part_reduce_kernel, dim3(g > 2048 ? 2048 : g), dim3(256), 0, st, (const float*)workspace,
                       (long)n * ND, nz, bias, add, y, total4, ND / 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { lotus_set_error("conv_part_reduce: %s", hipGetErrorString(e)); *rc = LOTUS_E_LAUNCH; }
  }
  return 1;
}
