// lotus-hip: LayerNorm and BatchNorm1d(+GELU) forward/backward — HBM-bound row/column
// reductions (nn.LayerNorm call sites PointTransformerV3/model.py:624-645, model_ca.py:114-124;
// nn.BatchNorm1d(eps=1e-3, momentum=0.01) model_ca.py:226 at the stem, every pooling and every
// unpooling branch).  One pass over the activations per kernel, float4 accesses, no atomics;
// column reductions go through per-block partials that a second tiny kernel sums in fixed order.
#include "common.h"
#include <stdlib.h>

namespace LOTUS_NS {

// --------------------------------------------------------------------------------- LayerNorm
// A row is owned by LPR lanes (16/32/64); lane holds float4 #(l + j * LPR), j < NV <= 4.
struct LnP {
  const act_t* x;
  const act_t* res;  // optional residual added AFTER the norm: y = LN(x) + res
  const float* gamma;
  const float* beta;
  act_t* y;
  float* mean;
  float* rstd;
  int M, C, LPR, NV;
  float eps;
};

__device__ __forceinline__ float group_sum(float v, int lpr) {
  for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(LnP p) {
  LOTUS_T_PRIO();
  const int rpb = 256 / p.LPR;
  const int row = blockIdx.x * rpb + threadIdx.x / p.LPR;
  const int l = threadIdx.x % p.LPR;
  const bool valid = row < p.M;
  const int c4 = p.C / 4;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int q = l + j * p.LPR;
    if (valid && j < p.NV && q < c4) {
      v[j] = ld4q(p.x + (long)row * p.C, q);
      s += v[j].x + v[j].y + v[j].z + v[j].w;
    }
  }
  const float mean = group_sum(s, p.LPR) / p.C;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = l + j * p.LPR;
    if (valid && j < p.NV && q < c4) {
      const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      ss += a * a + b * b + c * c + d * d;
    }
  }
  const float rstd = rsqrtf(group_sum(ss, p.LPR) / p.C + p.eps);
  if (!valid) return;
  if (l == 0) {
    if (p.mean) p.mean[row] = mean;
    if (p.rstd) p.rstd[row] = rstd;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = l + j * p.LPR;
    if (j < p.NV && q < c4) {
      const float4 g = ld4q(p.gamma, q);
      const float4 b = ld4q(p.beta, q);
      float4 o;
      o.x = (v[j].x - mean) * rstd * g.x + b.x;
      o.y = (v[j].y - mean) * rstd * g.y + b.y;
      o.z = (v[j].z - mean) * rstd * g.z + b.z;
      o.w = (v[j].w - mean) * rstd * g.w + b.w;
      if (p.res) {
        const float4 r = ld4q(p.res + (long)row * p.C, q);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      st4q(p.y + (long)row * p.C, q, o);
    }
  }
}

struct LnBwdP {
  const act_t* dy;
  const act_t* x;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const act_t* add;  // optional: dx += add
  act_t* dx;
  float* part;  // [gridDim.x][2][C]  (dgamma, dbeta partials)
  int M, C, LPR, NV;
  // optional second output: dz = dx * dropout mask of the layer that PRODUCED this block's input (its backward would
  // otherwise start with a stand-alone mask kernel over dx)
  act_t* dz;
  unsigned long long drop_seed;
  unsigned drop_thresh;
  float drop_inv_keep;
};

__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdP p) {
  LOTUS_T_PRIO();
  extern __shared__ float red[];  // [rpb][2][C]
  const int rpb = 256 / p.LPR;
  const int rslot = threadIdx.x / p.LPR, l = threadIdx.x % p.LPR;
  const int c4 = p.C / 4;
  float4 pg[4], pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) pg[j] = pb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int row0 = blockIdx.x * rpb; row0 < p.M; row0 += gridDim.x * rpb) {
    const int row = row0 + rslot;
    const bool valid = row < p.M;
    const float mean = valid ? p.mean[row] : 0.f, rstd = valid ? p.rstd[row] : 0.f;
    float4 xh[4], g[4], av[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = l + j * p.LPR;
      xh[j] = g[j] = av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && j < p.NV && q < c4) {
        const float4 xv = ld4q(p.x + (long)row * p.C, q);
        const float4 dv = ld4q(p.dy + (long)row * p.C, q);
        if (p.add) av[j] = ld4q(p.add + (long)row * p.C, q);  // in flight with x / dy
        const float4 gm = ld4q(p.gamma, q);
        xh[j] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        pg[j].x += dv.x * xh[j].x; pg[j].y += dv.y * xh[j].y; pg[j].z += dv.z * xh[j].z; pg[j].w += dv.w * xh[j].w;
        pb[j].x += dv.x; pb[j].y += dv.y; pb[j].z += dv.z; pb[j].w += dv.w;
        g[j] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
        s1 += g[j].x + g[j].y + g[j].z + g[j].w;
        s2 += g[j].x * xh[j].x + g[j].y * xh[j].y + g[j].z * xh[j].z + g[j].w * xh[j].w;
      }
    }
    s1 = group_sum(s1, p.LPR) / p.C;
    s2 = group_sum(s2, p.LPR) / p.C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = l + j * p.LPR;
      if (valid && j < p.NV && q < c4) {
        float4 o;
        o.x = rstd * (g[j].x - s1 - xh[j].x * s2);
        o.y = rstd * (g[j].y - s1 - xh[j].y * s2);
        o.z = rstd * (g[j].z - s1 - xh[j].z * s2);
        o.w = rstd * (g[j].w - s1 - xh[j].w * s2);
        o.x += av[j].x; o.y += av[j].y; o.z += av[j].z; o.w += av[j].w;
        st4q(p.dx + (long)row * p.C, q, o);
        if (p.dz) {
          const unsigned long long e = (unsigned long long)row * p.C + 4 * q;
          float dm[4];
          dropout_scale4(p.drop_seed, e, p.drop_thresh, p.drop_inv_keep, dm);
          const float4 z = make_float4(o.x * dm[0], o.y * dm[1], o.z * dm[2], o.w * dm[3]);
          st4q(p.dz + (long)row * p.C, q, z);
        }
      }
    }
  }
  // block reduction of the column partials
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = l + j * p.LPR;
    if (j < p.NV && q < c4) {
      st4q(red + (long)(rslot * 2 + 0) * p.C, q, pg[j]);
      st4q(red + (long)(rslot * 2 + 1) * p.C, q, pb[j]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * p.C; c += 256) {
    const int which = c / p.C, col = c % p.C;
    float s = 0.f;
    for (int r = 0; r < rpb; ++r) s += red[(long)(r * 2 + which) * p.C + col];
    p.part[((long)blockIdx.x * 2 + which) * p.C + col] = s;
  }
}

// out[which][c] (+)= sum_b part[b][which][c]; block = 32 columns x 32 row lanes, fixed-order tree
__global__ __launch_bounds__(1024) void colpart_reduce_kernel(const float* __restrict__ part, float* __restrict__ o0,
                                                              float* __restrict__ o1, int nb, int C, int accumulate) {
  __shared__ float red[32][33];
  const int cl = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < 2 * C) {
    int b = r;
    for (; b + 96 < nb; b += 128) {  // four independent loads in flight
      const float v0 = part[(long)b * 2 * C + c], v1 = part[(long)(b + 32) * 2 * C + c];
      const float v2 = part[(long)(b + 64) * 2 * C + c], v3 = part[(long)(b + 96) * 2 * C + c];
      s += (v0 + v1) + (v2 + v3);
    }
    for (; b < nb; b += 32) s += part[(long)b * 2 * C + c];
  }
  red[r][cl] = s;
  __syncthreads();
  if (r == 0 && c < 2 * C) {
    float t = 0.f;
    for (int k = 0; k < 32; ++k) t += red[k][cl];
    const int which = c / C, col = c % C;
    float* o = which ? o1 : o0;
    if (o) o[col] = accumulate ? o[col] + t : t;
  }
}

static int ln_geometry(int C, int* LPR, int* NV) {
  if (C % 4) return -1;
  const int c4 = C / 4;
  int lpr = 64;
  if (c4 <= 16) lpr = 16;
  else if (c4 <= 32) lpr = 32;
  const int nv = (c4 + lpr - 1) / lpr;
  if (nv > 4) return -1;
  *LPR = lpr;
  *NV = nv;
  return 0;
}

// backward geometry: as few lanes per row as 4 float4 per lane allow — every lane then has 8 independent 16-byte
// loads (x, dy) in flight per row instead of 2, which is what the HBM latency needs at 8-16 waves per CU
static int ln_geometry_bwd(int C, int* LPR, int* NV) {
  if (C % 4) return -1;
  const int c4 = C / 4;
  int lpr = 64;
  while (lpr > 4 && c4 <= (lpr / 2) * 4) lpr /= 2;
  const int nv = (c4 + lpr - 1) / lpr;
  if (nv > 4) return -1;
  *LPR = lpr;
  *NV = nv;
  return 0;
}

// --------------------------------------------------------------------------------- BatchNorm
#define BN_GS 16         // blocks per group of the fused (last-arrival) statistics reduction
#define BN_FUSED_MAX_GRID 256
#define BN_COUNTERS (1 + BN_FUSED_MAX_GRID / BN_GS)
// Column statistics in double: per-block partial (sum, sumsq) -> fixed-order reduction.
struct BnStatP {
  const act_t* x;     // [M][C]
  const act_t* dy;    // backward: dy (stats of dz, dz*xhat), else null
  const float* mean;  // backward
  const float* invstd;
  const float* gamma;
  const float* beta;
  double* part;  // [gridDim.x][2][C]
  int M, C, act;
  // fused reduction (round 4): with `cnt` the LAST block to arrive sums the per-block partials in block order (fixed order:
  // deterministic) into sums[2C + 1] = (.., .., M) and — forward, `mean` given — finishes the statistics: mean, 1/std, running
  // averages.  One launch instead of three (statistics, partial reduction, finalisation).  cnt is left at zero.
  unsigned* cnt;
  double* sums;
  float* out_mean;
  float* out_invstd;
  float* running_mean;
  float* running_var;
  float eps, momentum;
};

__global__ __launch_bounds__(256) void bn_stat_kernel(BnStatP p) {
  LOTUS_T_PRIO();
  extern __shared__ double dred[];  // [rslots][2][C]
  const int c4 = p.C / 4;
  const int tpr = c4 < 256 ? c4 : 256;  // threads per row
  const int rslots = 256 / tpr;
  const int rslot = threadIdx.x / tpr, l = threadIdx.x % tpr;
  const bool live = rslot < rslots;
  for (int q0 = 0; q0 < c4; q0 += tpr) {
    const int q = q0 + l;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (live && q < c4) {
      float4 mu, is, gm, bt;
      if (p.dy) {
        mu = ld4q(p.mean, q);
        is = ld4q(p.invstd, q);
        gm = ld4q(p.gamma, q);
        bt = ld4q(p.beta, q);
      }
      const int stride = gridDim.x * rslots;
      int row = blockIdx.x * rslots + rslot;
      if (!p.dy) {
        // forward statistics: four independent row loads in flight per thread (one load per iteration left the
        // kernel latency-bound at 0.24 TB/s); the accumulation order per thread is unchanged
        for (; row + 7 * stride < p.M; row += 8 * stride) {
          float4 xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) xv[u] = ld4q(p.x + (long)(row + u * stride) * p.C, q);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s0[e] += xs[e];
              s1[e] += (double)xs[e] * xs[e];
            }
          }
        }
      }
      if (p.dy) {  // backward statistics: two rows (four loads) in flight per thread
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
        const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
        for (; row + stride < p.M; row += 2 * stride) {
          float4 xv[2], dv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            xv[u] = ld4q(p.x + (long)(row + u * stride) * p.C, q);
            dv[u] = ld4q(p.dy + (long)(row + u * stride) * p.C, q);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, ds[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xh = (xs[e] - m4[e]) * i4[e];
              const float dz = ds[e] * act_grad_f(xh * g4[e] + b4[e], p.act);
              s0[e] += dz;
              s1[e] += (double)dz * xh;
            }
          }
        }
      }
      for (; row < p.M; row += stride) {
        const float4 xv = ld4q(p.x + (long)row * p.C, q);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        if (!p.dy) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0[e] += xs[e];
            s1[e] += (double)xs[e] * xs[e];
          }
        } else {
          const float4 dv = ld4q(p.dy + (long)row * p.C, q);
          const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
          const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
          const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (xs[e] - m4[e]) * i4[e];
            const float dz = ds[e] * act_grad_f(xh * g4[e] + b4[e], p.act);
            s0[e] += dz;
            s1[e] += (double)dz * xh;
          }
        }
      }
    }
    if (live && q < c4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dred[(long)(rslot * 2 + 0) * p.C + q * 4 + e] = s0[e];
        dred[(long)(rslot * 2 + 1) * p.C + q * 4 + e] = s1[e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * p.C; c += 256) {
    const int which = c / p.C, col = c % p.C;
    double s = 0;
    for (int r = 0; r < rslots; ++r) s += dred[(long)(r * 2 + which) * p.C + col];
    double* dst = p.part + ((long)blockIdx.x * 2 + which) * p.C + col;
    if (p.cnt) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by another CU below
    else *dst = s;
  }
  if (!p.cnt) return;
  // ---- two-level last-arrival reduction (hand-off contract: see gemm.hip st_agent4 — write-through stores, L1-bypassing
  // loads, workgroup fences around device-scope counters).  Blocks form groups of BN_GS consecutive ids: the last block of
  // a group to arrive sums the group's partial rows in row order into a group row, the last GROUP to finish sums the group
  // rows in group order — every sum has a fixed order, and no block ever re-reads more than BN_GS rows (one block alone
  // streams a hand-off at only ~65 GB/s: MI355X_MICROARCH.md, handoff-payload).  cnt[0] counts groups, cnt[1 + g] group g.
  __shared__ int s_last;
  const int ncol = 2 * p.C, nb = gridDim.x;
  const int group = blockIdx.x / BN_GS, ngroups = (nb + BN_GS - 1) / BN_GS;
  const int gsize = min(BN_GS, nb - group * BN_GS);
  double* gpart = p.part + (long)nb * ncol;  // [ngroups][2C]
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores above have been acknowledged (explicit, see gemm_common.h)
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(p.cnt + 1 + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)gsize - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int c = threadIdx.x; c < ncol; c += 256) {
    const double* src = p.part + (long)group * BN_GS * ncol + c;
    double v[BN_GS];
#pragma unroll
    for (int u = 0; u < BN_GS; ++u)  // all rows in flight at once (rows past the group re-read its first row, unused)
      v[u] = __hip_atomic_load(src + (long)(u < gsize ? u : 0) * ncol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double acc = 0;
#pragma unroll
    for (int u = 0; u < BN_GS; ++u) acc += u < gsize ? v[u] : 0.0;
    __hip_atomic_store(gpart + (long)group * ncol + c, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) p.cnt[1 + group] = 0;  // ready for the next launch on this stream
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores above have been acknowledged (explicit, see gemm_common.h)
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(p.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)ngroups - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int c = threadIdx.x; c < ncol; c += 256) {
    double acc = 0;
    int g0 = 0;
    for (; g0 + BN_GS <= ngroups; g0 += BN_GS) {
      double v[BN_GS];
#pragma unroll
      for (int u = 0; u < BN_GS; ++u) v[u] = __hip_atomic_load(gpart + (long)(g0 + u) * ncol + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < BN_GS; ++u) acc += v[u];
    }
    for (; g0 < ngroups; ++g0) acc += __hip_atomic_load(gpart + (long)g0 * ncol + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.sums[c] = acc;
    // backward with parameter-gradient outputs (lotus_batchnorm_bwd_stats_fused_params): the LOCAL sums are dbeta | dgamma
    if (p.dy && p.out_mean) (c < p.C ? p.out_mean : p.out_invstd)[c < p.C ? c : c - p.C] = (float)acc;
  }
  if (threadIdx.x == 0) {
    p.sums[ncol] = (double)p.M;
    *p.cnt = 0;
  }
  if (p.out_mean && !p.dy) {  // forward: finish the statistics here (what bn_finalize_kernel does)
    __syncthreads();  // sums[] written by this block's threads above
    const double count = (double)p.M;
    for (int c = threadIdx.x; c < p.C; c += 256) {
      const double m = p.sums[c] / count;
      double var = p.sums[p.C + c] / count - m * m;
      if (var < 0) var = 0;
      p.out_mean[c] = (float)m;
      p.out_invstd[c] = (float)(1.0 / sqrt(var + (double)p.eps));
      if (p.running_mean) {
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)m;
        p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
      }
    }
  }
}

__global__ __launch_bounds__(256) void bn_part_reduce_kernel(const double* __restrict__ part, double* __restrict__ sums,
                                                             int nb, int C, int M) {
  __shared__ double red[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
  if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * C] = (double)M;  // local row count travels with the sums
  double s = 0;
  if (c < 2 * C)
    for (int b = r; b < nb; b += 8) s += part[(long)b * 2 * C + c];
  red[r][threadIdx.x & 31] = s;
  __syncthreads();
  if (r == 0 && c < 2 * C) {
    double t = 0;
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    sums[c] = t;
  }
}

// sums = (sum x, sum x^2, count), already all-reduced across ranks when SyncBN.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int C, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double count = sums[2 * C];
  const double m = sums[c] / count;
  double var = sums[C + c] / count - m * m;
  if (var < 0) var = 0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float* mean,
                                     float* invstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  invstd[c] = 1.f / sqrtf(rv[c] + eps);
}

struct BnApplyP {
  const act_t* x;
  const act_t* dy;  // backward when non-null
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  const double* sums;  // backward (train): (sum dz, sum dz*xhat); null in eval mode
  act_t* y;            // forward output or dx
  float* dgamma;       // backward: written by block 0
  float* dbeta;
  long total4;  // M * C / 4
  int C, act, accumulate;
  // forward from (all-reduced) statistics sums (sum x, sum x^2, count) instead of mean / invstd (lotus_batchnorm_apply_sums: the
  // SyncBatchNorm forward without a finalisation launch): every thread derives the constants of its own column quad, the
  // threads of the first quad row also write mean / invstd (saved for backward) and update the running averages
  const double* fsums;
  float* out_mean;
  float* out_invstd;
  float* running_mean;
  float* running_var;
  float eps, momentum;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyP p) {
  LOTUS_T_PRIO();
  const int c4 = p.C / 4;
  // The launch makes gridDim.x * 256 a multiple of c4, so a thread keeps ONE column quad for its whole grid-stride
  // walk: the per-column constants (incl. the two fp64 divisions of the backward) are loaded / computed once, and the
  // walk keeps two row loads in flight.
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
  if (i0 < p.total4) {
    const int q = (int)(i0 % c4);
    float m4[4], i4[4];
    if (p.fsums) {  // the arithmetic of bn_finalize_kernel, per thread for its four columns
      const double count = p.fsums[2 * p.C];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = q * 4 + e;
        const double m = p.fsums[c] / count;
        double var = p.fsums[p.C + c] / count - m * m;
        if (var < 0) var = 0;
        m4[e] = (float)m;
        i4[e] = (float)(1.0 / sqrt(var + (double)p.eps));
        if (i0 < c4) {
          p.out_mean[c] = m4[e];
          p.out_invstd[c] = i4[e];
          if (p.running_mean) {
            const double unbiased = count > 1 ? var * count / (count - 1) : var;
            p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)m;
            p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
          }
        }
      }
    } else {
      const float4 mu = ld4q(p.mean, q);
      const float4 is = ld4q(p.invstd, q);
      m4[0] = mu.x; m4[1] = mu.y; m4[2] = mu.z; m4[3] = mu.w;
      i4[0] = is.x; i4[1] = is.y; i4[2] = is.z; i4[3] = is.w;
    }
    const float4 gm = ld4q(p.gamma, q);
    const float4 bt = ld4q(p.beta, q);
    const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
    float mdz[4] = {0.f, 0.f, 0.f, 0.f}, mdx[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.dy && p.sums) {
      const double count = p.sums[2 * p.C];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mdz[e] = (float)(p.sums[q * 4 + e] / count);
        mdx[e] = (float)(p.sums[p.C + q * 4 + e] / count);
      }
    }
    auto one = [&](const float4 xv, const float4 dv) {
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xs[e] - m4[e]) * i4[e];
        if (!p.dy) {
          o[e] = act_f(xh * g4[e] + b4[e], p.act);
        } else {
          const float dz = ds[e] * act_grad_f(xh * g4[e] + b4[e], p.act);
          o[e] = p.sums ? g4[e] * i4[e] * (dz - mdz[e] - xh * mdx[e]) : g4[e] * i4[e] * dz;
        }
      }
      return make_float4(o[0], o[1], o[2], o[3]);
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    long i = i0;
    for (; i + step < p.total4; i += 2 * step) {
      const float4 xa = ld4q(p.x, i), xb = ld4q(p.x, i + step);
      const float4 da = p.dy ? ld4q(p.dy, i) : z4;
      const float4 db = p.dy ? ld4q(p.dy, i + step) : z4;
      st4q(p.y, i, one(xa, da));
      st4q(p.y, i + step, one(xb, db));
    }
    if (i < p.total4)
      st4q(p.y, i, one(ld4q(p.x, i), p.dy ? ld4q(p.dy, i) : z4));
  }
  if (p.dy && p.dgamma && blockIdx.x == 0) {
    for (int c = threadIdx.x; c < p.C; c += 256) {
      const float dg = (float)p.sums[p.C + c], db = (float)p.sums[c];
      p.dgamma[c] = p.accumulate ? p.dgamma[c] + dg : dg;
      p.dbeta[c] = p.accumulate ? p.dbeta[c] + db : db;
    }
  }
}

// grid of bn_apply_kernel: ~2 float4 per thread, at most 4096 blocks, and gridDim.x * 256 a multiple of C / 4 (a thread
// then stays on one column quad)
static int bn_apply_grid(long total4, int C) {
  const int c4 = C / 4;
  int a = 256, b = c4;
  while (b) { const int t = a % b; a = b; b = t; }   // a = gcd(256, c4)
  const int unit = c4 / a;                             // blocks per column period
  long g = (total4 + 511) / 512;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  g = (g + unit - 1) / unit * unit;
  return (int)g;
}

#define BN_MAX_GRID 512
static int bn_grid(int M, int C) {
  const int c4 = C / 4;
  const int tpr = c4 < 256 ? c4 : 256;
  const int rslots = 256 / tpr;
  // 16 rows per thread.  Measured in the step: 16 / 4 / 2 rows -> 868 / 861 / 850 samples/s — more, smaller blocks raise the
  // stand-alone rate of this kernel but feed more partials to the fixed-order reduction behind it
  constexpr int rows = 16;
  int g = cdiv(M, rslots * rows);
  if (g > BN_MAX_GRID) g = BN_MAX_GRID;
  if (g < 1) g = 1;
  return g;
}
static size_t bn_dyn_lds(int C) {
  const int c4 = C / 4;
  const int tpr = c4 < 256 ? c4 : 256;
  size_t n = (size_t)(256 / tpr) * 2 * C;
  if (n < 256) n = 256;  // the fused reduction stages 256 partial sums
  return n * sizeof(double);
}
// grid of the fused statistics kernels: at most BN_FUSED_MAX_GRID blocks (16 groups of 16) — 256 blocks with eight row loads
// per thread in flight stream a level-0 tensor at HBM rate, and the partial slab + group rows fit the BN workspace
static int bn_grid_fused(int M, int C) {
  const int g = bn_grid(M, C);
  return g > BN_FUSED_MAX_GRID ? BN_FUSED_MAX_GRID : g;
}

extern "C" {

// y = LayerNorm(x; gamma, beta, eps) (+ res).  mean/rstd [M] saved for backward (optional).
int lotus_layernorm_fwd(const act_t* x, const act_t* res, const float* gamma, const float* beta, act_t* y,
                        float* mean, float* rstd, int M, int C, float eps, void* stream) {
  LnP p;
  p.x = x; p.res = res; p.gamma = gamma; p.beta = beta; p.y = y; p.mean = mean; p.rstd = rstd;
  p.M = M; p.C = C; p.eps = eps;
  LOTUS_CHECK_ARG(x && gamma && beta && y && M >= 0, "lotus_layernorm_fwd: bad arguments");
  LOTUS_CHECK_ARG(ln_geometry(C, &p.LPR, &p.NV) == 0, "lotus_layernorm_fwd: unsupported C=%d", C);
  if (M == 0) return LOTUS_OK;
  LOTUS_LAUNCH(ln_fwd_kernel, dim3(cdiv(M, 256 / p.LPR)), dim3(256), 0, (hipStream_t)stream, p);
  LOTUS_LAUNCH_CHECK("lotus_layernorm_fwd");
  return LOTUS_OK;
}

#define LN_BWD_MAX_GRID 1024
static int ln_bwd_grid(int M, int rpb) {
  int grid = cdiv(M > 0 ? M : 1, rpb * 2);  // >= 2 row groups per block, <= 4 blocks per CU
  return grid > LN_BWD_MAX_GRID ? LN_BWD_MAX_GRID : grid;
}
size_t lotus_layernorm_bwd_workspace(int M, int C) { return (size_t)LN_BWD_MAX_GRID * 2 * C * sizeof(float); }

// dx = LN'(dy) (+ add); dgamma/dbeta (+)= column sums.  With dgamma == NULL only dx is produced and the
// per-block column partials stay in `workspace` for lotus_layernorm_bwd_params (which a caller may run on
// another stream: the parameter gradients are off the critical path of backward).
int lotus_layernorm_bwd(const act_t* dy, const act_t* x, const float* mean, const float* rstd, const float* gamma,
                        const act_t* add, act_t* dx, float* dgamma, float* dbeta, int M, int C, int accumulate,
                        act_t* dz, float drop_p, unsigned long long drop_seed, void* workspace, size_t workspace_bytes,
                        void* stream) {
  LnBwdP p;
  p.dz = (dz && drop_p > 0.f) ? dz : nullptr;
  p.drop_seed = drop_seed;
  lotus_drop_setup(drop_p, &p.drop_thresh, &p.drop_inv_keep);
  LOTUS_CHECK_ARG(!dz || drop_p > 0.f, "lotus_layernorm_bwd: dz needs drop_p > 0");
  p.dy = dy; p.x = x; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.add = add; p.dx = dx;
  p.part = (float*)workspace; p.M = M; p.C = C;
  LOTUS_CHECK_ARG(dy && x && mean && rstd && gamma && dx && M >= 0, "lotus_layernorm_bwd: bad arguments");
  LOTUS_CHECK_ARG(ln_geometry_bwd(C, &p.LPR, &p.NV) == 0, "lotus_layernorm_bwd: unsupported C=%d", C);
  const int rpb = 256 / p.LPR;
  const int grid = ln_bwd_grid(M, rpb);
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)grid * 2 * C * sizeof(float),
                  "lotus_layernorm_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  LOTUS_LAUNCH(ln_bwd_kernel, dim3(grid), dim3(256), (size_t)rpb * 2 * C * sizeof(float), st, p);
  if (dgamma && dbeta)
    LOTUS_LAUNCH(colpart_reduce_kernel, dim3(cdiv(2 * C, 32)), dim3(1024), 0, st, p.part, dgamma, dbeta, grid, C,
                       accumulate);
  LOTUS_LAUNCH_CHECK("lotus_layernorm_bwd");
  return LOTUS_OK;
}

// dgamma/dbeta (+)= the column partials a preceding lotus_layernorm_bwd(dgamma = NULL) of the same (M, C) left in
// `workspace`; fixed summation order.
int lotus_layernorm_bwd_params(const void* workspace, int M, int C, float* dgamma, float* dbeta, int accumulate,
                               void* stream) {
  int lpr, nv;
  LOTUS_CHECK_ARG(workspace && dgamma && dbeta && M >= 0, "lotus_layernorm_bwd_params: bad arguments");
  LOTUS_CHECK_ARG(ln_geometry_bwd(C, &lpr, &nv) == 0, "lotus_layernorm_bwd_params: unsupported C=%d", C);
  const int grid = ln_bwd_grid(M, 256 / lpr);
  LOTUS_LAUNCH(colpart_reduce_kernel, dim3(cdiv(2 * C, 32)), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, dgamma, dbeta, grid, C, accumulate);
  LOTUS_LAUNCH_CHECK("lotus_layernorm_bwd_params");
  return LOTUS_OK;
}

// number of partial rows lotus_layernorm_bwd(dgamma = NULL) leaves in its workspace for (M, C)
int lotus_layernorm_bwd_parts(int M, int C) {
  int lpr, nv;
  if (ln_geometry_bwd(C, &lpr, &nv)) return 0;
  return ln_bwd_grid(M, 256 / lpr);
}
// lotus_layernorm_bwd_params for an explicit number of partial rows (what lotus_linear_dgrad_ln reports)
int lotus_layernorm_bwd_params_n(const void* workspace, int nparts, int C, float* dgamma, float* dbeta, int accumulate, void* stream) {
  LOTUS_CHECK_ARG(workspace && dgamma && dbeta && nparts >= 0 && C > 0, "lotus_layernorm_bwd_params_n: bad arguments");
  if (nparts == 0 && accumulate) return LOTUS_OK;
  LOTUS_LAUNCH(colpart_reduce_kernel, dim3(cdiv(2 * C, 32)), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, dgamma, dbeta,
               nparts, C, accumulate);
  LOTUS_LAUNCH_CHECK("lotus_layernorm_bwd_params_n");
  return LOTUS_OK;
}

size_t lotus_batchnorm_workspace(int M, int C) { return (size_t)BN_MAX_GRID * 2 * C * sizeof(double); }

// Forward statistics: sums[2*C+1] (double) = (sum x, sum x^2, M) over the M local rows.
int lotus_batchnorm_stats(const act_t* x, double* sums, int M, int C, void* workspace, size_t workspace_bytes,
                          void* stream) {
  LOTUS_CHECK_ARG(x && sums && C % 4 == 0 && M >= 0, "lotus_batchnorm_stats: bad arguments (C=%d)", C);
  const int grid = bn_grid(M, C);
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)grid * 2 * C * sizeof(double),
                  "lotus_batchnorm_stats: workspace too small");
  BnStatP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.part = (double*)workspace; p.M = M; p.C = C;
  hipStream_t st = (hipStream_t)stream;
  LOTUS_LAUNCH(bn_stat_kernel, dim3(grid), dim3(256), bn_dyn_lds(C), st, p);
  LOTUS_LAUNCH(bn_part_reduce_kernel, dim3(cdiv(2 * C, 32)), dim3(256), 0, st, p.part, sums, grid, C, M);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_stats");
  return LOTUS_OK;
}

// One-launch forward statistics (no SyncBatchNorm message in between): sums[2C + 1], mean, invstd and the running averages
// from a single pass over x; `counter` = one zeroed unsigned of the launching stream (left at zero).  Replaces
// lotus_batchnorm_stats + lotus_batchnorm_finalize (3 launches).
int lotus_batchnorm_stats_fused(const act_t* x, double* sums, float* mean, float* invstd, float* running_mean, float* running_var,
                                int M, int C, float eps, float momentum, void* workspace, size_t workspace_bytes, void* counter,
                                void* stream) {
  // (mean == invstd == null: the sums alone — the SyncBatchNorm forward, whose statistics are finished after the message)
  LOTUS_CHECK_ARG(x && sums && (!mean == !invstd) && counter && C % 4 == 0 && M > 0, "lotus_batchnorm_stats_fused: bad arguments (C=%d)", C);
  const int grid = bn_grid_fused(M, C);
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)(grid + BN_COUNTERS) * 2 * C * sizeof(double), "lotus_batchnorm_stats_fused: workspace too small");
  BnStatP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.part = (double*)workspace; p.M = M; p.C = C;
  p.cnt = (unsigned*)counter; p.sums = sums; p.out_mean = mean; p.out_invstd = invstd;
  p.running_mean = running_mean; p.running_var = running_var; p.eps = eps; p.momentum = momentum;
  LOTUS_LAUNCH(bn_stat_kernel, dim3(grid), dim3(256), bn_dyn_lds(C), (hipStream_t)stream, p);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_stats_fused");
  return LOTUS_OK;
}

// The same, and dbeta = sum dz, dgamma = sum dz * xhat of the LOCAL rows as fp32 vectors (what a SyncBatchNorm backward keeps
// before its sums are all-reduced: no conversion kernels between the statistics and the message).
int lotus_batchnorm_bwd_stats_fused_params(const act_t* dy, const act_t* x, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, double* sums, float* dgamma, float* dbeta, int M, int C, int act,
                                           void* workspace, size_t workspace_bytes, void* counter, void* stream);
// One-launch backward statistics: sums = (sum dz, sum dz * xhat, M) — lotus_batchnorm_bwd_stats without the second launch.
int lotus_batchnorm_bwd_stats_fused(const act_t* dy, const act_t* x, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, double* sums, int M, int C, int act, void* workspace, size_t workspace_bytes,
                                    void* counter, void* stream) {
  LOTUS_CHECK_ARG(dy && x && sums && counter && C % 4 == 0 && M > 0, "lotus_batchnorm_bwd_stats_fused: bad arguments");
  const int grid = bn_grid_fused(M, C);
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)(grid + BN_COUNTERS) * 2 * C * sizeof(double), "lotus_batchnorm_bwd_stats_fused: workspace too small");
  BnStatP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.dy = dy; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
  p.part = (double*)workspace; p.M = M; p.C = C; p.act = act;
  p.cnt = (unsigned*)counter; p.sums = sums;
  LOTUS_LAUNCH(bn_stat_kernel, dim3(grid), dim3(256), bn_dyn_lds(C), (hipStream_t)stream, p);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_bwd_stats_fused");
  return LOTUS_OK;
}

int lotus_batchnorm_bwd_stats_fused_params(const act_t* dy, const act_t* x, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, double* sums, float* dgamma, float* dbeta, int M, int C, int act,
                                           void* workspace, size_t workspace_bytes, void* counter, void* stream) {
  LOTUS_CHECK_ARG(dy && x && sums && dgamma && dbeta && counter && C % 4 == 0 && M > 0, "lotus_batchnorm_bwd_stats_fused_params: bad arguments");
  const int grid = bn_grid_fused(M, C);
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)(grid + BN_COUNTERS) * 2 * C * sizeof(double), "lotus_batchnorm_bwd_stats_fused_params: workspace too small");
  BnStatP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.dy = dy; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
  p.part = (double*)workspace; p.M = M; p.C = C; p.act = act;
  p.cnt = (unsigned*)counter; p.sums = sums; p.out_mean = dbeta; p.out_invstd = dgamma;
  LOTUS_LAUNCH(bn_stat_kernel, dim3(grid), dim3(256), bn_dyn_lds(C), (hipStream_t)stream, p);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_bwd_stats_fused_params");
  return LOTUS_OK;
}

// mean/invstd from (possibly all-reduced) sums[2*C+1]; updates running stats if given.
int lotus_batchnorm_finalize(const double* sums, float* mean, float* invstd, float* running_mean,
                             float* running_var, int C, float eps, float momentum, void* stream) {
  LOTUS_CHECK_ARG(sums && mean && invstd, "lotus_batchnorm_finalize: bad arguments");
  LOTUS_LAUNCH(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, mean,
                     invstd, running_mean, running_var, C, eps, momentum);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_finalize");
  return LOTUS_OK;
}

int lotus_batchnorm_eval_stats(const float* running_mean, const float* running_var, float* mean, float* invstd, int C,
                               float eps, void* stream) {
  LOTUS_LAUNCH(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean,
                     running_var, mean, invstd, C, eps);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_eval_stats");
  return LOTUS_OK;
}

// y = act((x - mean) * invstd * gamma + beta)
int lotus_batchnorm_apply(const act_t* x, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, act_t* y, int M, int C, int act, void* stream) {
  LOTUS_CHECK_ARG(x && y && C % 4 == 0, "lotus_batchnorm_apply: bad arguments");
  if (M == 0) return LOTUS_OK;
  BnApplyP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.y = y;
  p.total4 = (long)M * C / 4; p.C = C; p.act = act;
  const int grid = bn_apply_grid(p.total4, C);
  LOTUS_LAUNCH(bn_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_apply");
  return LOTUS_OK;
}

// The same from the statistics sums = (sum x, sum x^2, count) — already all-reduced when SyncBatchNorm is on — without the
// finalisation launch between the message and the apply pass: mean / invstd (saved for backward) and the running averages are
// written by the apply kernel itself.  M == 0 (an empty shard): the finalisation alone.
int lotus_batchnorm_apply_sums(const act_t* x, const double* sums, const float* gamma, const float* beta, act_t* y, float* mean,
                               float* invstd, float* running_mean, float* running_var, int M, int C, int act, float eps,
                               float momentum, void* stream) {
  LOTUS_CHECK_ARG(sums && mean && invstd && C % 4 == 0 && (M == 0 || (x && y)), "lotus_batchnorm_apply_sums: bad arguments");
  BnApplyP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.gamma = gamma; p.beta = beta; p.y = y;
  p.total4 = (long)M * C / 4; p.C = C; p.act = act;
  p.fsums = sums; p.out_mean = mean; p.out_invstd = invstd; p.running_mean = running_mean; p.running_var = running_var;
  p.eps = eps; p.momentum = momentum;
  const int grid = bn_apply_grid(p.total4, C);
  if (M == 0 || (long)grid * 256 < C / 4)  // (fewer threads than column quads cannot happen for M >= 1; kept as a guard)
    return lotus_batchnorm_finalize(sums, mean, invstd, running_mean, running_var, C, eps, momentum, stream) ||
           (M ? lotus_batchnorm_apply(x, mean, invstd, gamma, beta, y, M, C, act, stream) : LOTUS_OK);
  LOTUS_LAUNCH(bn_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_apply_sums");
  return LOTUS_OK;
}

// Backward statistics: sums = (sum dz, sum dz * xhat), dz = dy * act'(z).
int lotus_batchnorm_bwd_stats(const act_t* dy, const act_t* x, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, double* sums, int M, int C, int act,
                              void* workspace, size_t workspace_bytes, void* stream) {
  LOTUS_CHECK_ARG(dy && x && sums && C % 4 == 0, "lotus_batchnorm_bwd_stats: bad arguments");
  const int grid = bn_grid(M, C);
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= (size_t)grid * 2 * C * sizeof(double),
                  "lotus_batchnorm_bwd_stats: workspace too small");
  BnStatP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.dy = dy; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
  p.part = (double*)workspace; p.M = M; p.C = C; p.act = act;
  hipStream_t st = (hipStream_t)stream;
  LOTUS_LAUNCH(bn_stat_kernel, dim3(grid), dim3(256), bn_dyn_lds(C), st, p);
  LOTUS_LAUNCH(bn_part_reduce_kernel, dim3(cdiv(2 * C, 32)), dim3(256), 0, st, p.part, sums, grid, C, M);
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_bwd_stats");
  return LOTUS_OK;
}

// dx from dy; train = 1 uses batch statistics (sums[2*C+1] incl. the row count, all-reduced when SyncBN),
// train = 0 (eval) treats mean/invstd as constants.  dgamma/dbeta (+)= from sums.
int lotus_batchnorm_bwd_apply(const act_t* dy, const act_t* x, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const double* sums, act_t* dx, float* dgamma,
                              float* dbeta, int M, int C, int act, int train, int accumulate, void* stream) {
  LOTUS_CHECK_ARG(dy && x && dx && sums && C % 4 == 0, "lotus_batchnorm_bwd_apply: bad arguments");
  BnApplyP p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.dy = dy; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
  p.sums = sums; p.y = dx; p.dgamma = dgamma; p.dbeta = dbeta;
  p.total4 = (long)M * C / 4; p.C = C; p.act = act; p.accumulate = accumulate;
  const int grid = bn_apply_grid(p.total4, C);
  hipStream_t st = (hipStream_t)stream;
  if (!train) {
    // eval: dx = gamma * invstd * dz ; dgamma/dbeta still come from sums
    BnApplyP q = p;
    q.sums = nullptr;
    q.dgamma = nullptr;
    LOTUS_LAUNCH(bn_apply_kernel, dim3(grid), dim3(256), 0, st, q);
    BnApplyP g = p;
    g.total4 = 0;
    LOTUS_LAUNCH(bn_apply_kernel, dim3(1), dim3(256), 0, st, g);
  } else {
    LOTUS_LAUNCH(bn_apply_kernel, dim3(grid), dim3(256), 0, st, p);
  }
  LOTUS_LAUNCH_CHECK("lotus_batchnorm_bwd_apply");
  return LOTUS_OK;
}

}  // extern "C"

}  // namespace LOTUS_NS
