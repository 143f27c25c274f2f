// lotus-hip: grid pooling / unpooling, the gripper-action head reductions and the losses —
// HBM-bound segmented kernels (SerializedPooling model.py:760-766, SerializedUnpooling :817-828,
// ActionHead simple_policy_ptv3.py:113-157, compute_loss :308-373).  Segments are CSR ranges
// produced by the front-end, so every reduction is an ordered loop: no atomics, deterministic.
#include "common.h"

namespace LOTUS_NS {

int lotus_reduce_parts(const float* part, float* out, long n, long stride, int nz, int accumulate, hipStream_t st);  // gemm.hip

// ---------------------------------------------------------------- segment max over cluster members
// y[c][:] = max_{i in [seg[c], seg[c+1])} x[members[i]][:]   ; arg = winning parent row
__global__ void pool_max_fwd_kernel(const act_t* __restrict__ x, const int* __restrict__ members,
                                    const int* __restrict__ seg, int nc, int C, act_t* __restrict__ y,
                                    int* __restrict__ arg) {
  LOTUS_T_PRIO();
  const int c4 = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)nc * c4) return;
  const int c = (int)(gid / c4), q = (int)(gid % c4);
  const int a = seg[c], b = seg[c + 1];
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {-1, -1, -1, -1};
  for (int i = a; i < b; ++i) {
    const int p = members[i];
    const float4 v = ld4q(x + (long)p * C, q);
    const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (vs[e] > best[e]) {
        best[e] = vs[e];
        bi[e] = p;
      }
  }
  st4q(y + (long)c * C, q, make_float4(best[0], best[1], best[2], best[3]));
  reinterpret_cast<int4*>(arg + (long)c * C)[q] = make_int4(bi[0], bi[1], bi[2], bi[3]);
}

// dx[p][:] = (arg[cluster[p]][:] == p) ? dy[cluster[p]][:] : 0
__global__ void pool_max_bwd_kernel(const act_t* __restrict__ dy, const int* __restrict__ arg,
                                    const int* __restrict__ cluster, int n, int C, act_t* __restrict__ dx) {
  LOTUS_T_PRIO();
  const int c4 = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c4) return;
  const int p = (int)(gid / c4), q = (int)(gid % c4);
  const int c = cluster[p];
  const float4 g = ld4q(dy + (long)c * C, q);
  const int4 a = reinterpret_cast<const int4*>(arg + (long)c * C)[q];
  st4q(dx + (long)p * C, q, make_float4(a.x == p ? g.x : 0.f, a.y == p ? g.y : 0.f, a.z == p ? g.z : 0.f, a.w == p ? g.w : 0.f));
}

// x[p][:] = skip[p][:] + up[cluster[p]][:]
__global__ void unpool_fwd_kernel(const act_t* __restrict__ skip, const act_t* __restrict__ up,
                                  const int* __restrict__ cluster, int n, int C, act_t* __restrict__ x) {
  LOTUS_T_PRIO();
  const int c4 = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c4) return;
  const int p = (int)(gid / c4), q = (int)(gid % c4);
  const float4 s = ld4q(skip + (long)p * C, q);
  const float4 u = ld4q(up + (long)cluster[p] * C, q);
  st4q(x + (long)p * C, q, make_float4(s.x + u.x, s.y + u.y, s.z + u.z, s.w + u.w));
}

// dup[c][:] = sum over members of dx
__global__ void unpool_bwd_kernel(const act_t* __restrict__ dx, const int* __restrict__ members,
                                  const int* __restrict__ seg, int nc, int C, act_t* __restrict__ dup) {
  LOTUS_T_PRIO();
  const int c4 = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)nc * c4) return;
  const int c = (int)(gid / c4), q = (int)(gid % c4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = seg[c]; i < seg[c + 1]; ++i) {
    const float4 v = ld4q(dx + (long)members[i] * C, q);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  st4q(dup + (long)c * C, q, s);
}

// ---------------------------------------------------------------- per-cloud max over contiguous rows
// Two passes: (cloud, 128-column group, row split) blocks of 32 column quads x 32 row lanes keep the first-index maximum
// of their rows (16-byte loads), then one thread per (cloud, column) merges the CM_SPLITS partials in split order — the
// arg-max is the FIRST row attaining the maximum, as torch.max(x, 0) returns it.  (One block per (cloud, 32 columns)
// with 4-byte loads: 64 blocks, 46 us for 33 MB.)
#define CM_SPLITS 8
__global__ __launch_bounds__(1024) void cloud_max_part_kernel(const act_t* __restrict__ x, const int* __restrict__ off, int C,
                                                              float* __restrict__ pv, int* __restrict__ pi) {
  __shared__ float4 bv[32][33];
  __shared__ int4 bi[32][33];
  const int cq = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int b = blockIdx.x, q = blockIdx.y * 32 + cq, sp = blockIdx.z;  // q: float4 column index
  const int r0 = off[b], r1 = off[b + 1];
  const int chunk = (r1 - r0 + CM_SPLITS - 1) / CM_SPLITS;
  const int s0 = r0 + sp * chunk, s1 = min(r1, s0 + chunk);
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int idx[4] = {-1, -1, -1, -1};
  if (q * 4 < C)
    for (int r = s0 + ry; r < s1; r += 32) {
      const float4 v4 = ld4q(x + (long)r * C, q);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > best[e]) {
          best[e] = v[e];
          idx[e] = r;
        }
    }
  bv[ry][cq] = make_float4(best[0], best[1], best[2], best[3]);
  bi[ry][cq] = make_int4(idx[0], idx[1], idx[2], idx[3]);
  __syncthreads();
  if (ry == 0 && q * 4 < C) {
    for (int k = 1; k < 32; ++k) {
      const float4 v4 = bv[k][cq];
      const int4 i4 = bi[k][cq];
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
      const int ii[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > best[e] || (v[e] == best[e] && ii[e] >= 0 && (idx[e] < 0 || ii[e] < idx[e]))) {
          best[e] = v[e];
          idx[e] = ii[e];
        }
    }
    const long o = ((long)b * CM_SPLITS + sp) * C + q * 4;
    st4(pv + o, make_float4(best[0], best[1], best[2], best[3]));
    *reinterpret_cast<int4*>(pi + o) = make_int4(idx[0], idx[1], idx[2], idx[3]);
  }
}
__global__ void cloud_max_merge_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int B, int C,
                                       act_t* __restrict__ y, int* __restrict__ arg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  float best = -INFINITY;
  int idx = -1;
  for (int sp = 0; sp < CM_SPLITS; ++sp) {  // splits are ascending row ranges: strict > keeps the first maximum
    const float v = pv[((long)b * CM_SPLITS + sp) * C + c];
    const int ii = pi[((long)b * CM_SPLITS + sp) * C + c];
    if (ii >= 0 && (idx < 0 || v > best)) {
      best = v;
      idx = ii;
    }
  }
  y[i] = best;
  arg[i] = idx;
}

__global__ void cloud_max_bwd_kernel(const act_t* __restrict__ dy, const int* __restrict__ arg,
                                     const int* __restrict__ batch, int n, int C, const act_t* __restrict__ add,
                                     act_t* __restrict__ dx) {
  const int c4 = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c4) return;
  const int p = (int)(gid / c4), q = (int)(gid % c4);
  const int b = batch[p];
  const float4 g = ld4q(dy + (long)b * C, q);
  const int4 a = reinterpret_cast<const int4*>(arg + (long)b * C)[q];
  float4 o = make_float4(a.x == p ? g.x : 0.f, a.y == p ? g.y : 0.f, a.z == p ? g.z : 0.f, a.w == p ? g.w : 0.f);
  if (add) {
    const float4 v = ld4q(add + (long)p * C, q);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
  }
  st4q(dx + (long)p * C, q, o);
}

// ---------------------------------------------------------------- losses (simple_policy_ptv3.py:308-373)
// Position: per cloud b and axis c, soft-target cross entropy over all (point, bin) logits.
// xt[n][3*nb] logits (n, c, bin); tgt: cloud b at tgt_off = 3*nb*off[b], laid out [3][n_b*nb].
#define POS_CE_SPLITS 32
#define POS_CE_PART 8  // floats per slice partial: max, -, then (sum exp(x - max), sum t x, sum t) as doubles
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// slice s of cloud b, axis c: partial (max, sum exp(x - max), sum t x, sum t) of the heatmap cross-entropy.  The three sums are
// carried in DOUBLE up to the log-sum-exp: every probability of the (cloud, axis) is exp(x - lse), so an error of lse is a
// COMMON relative error of ~65 000 x 30 gradient entries that does not average out in the weight gradients behind them — with
// lse ~ 12 in fp32 that was 1e-6, and the head's first layer sat at 7.6e-5 of the float64 oracle (the fp32 reference: 1.7e-4).
__global__ __launch_bounds__(256) void pos_ce_part_kernel(const act_t* __restrict__ xt, const float* __restrict__ tgt,
                                                          const int* __restrict__ off, int nb,
                                                          float* __restrict__ part /*[B*3][SPLITS][POS_CE_PART]*/) {
  __shared__ float red[4];
  __shared__ double redd[4];
  const int s = blockIdx.x, bc = blockIdx.y, b = bc / 3, c = bc % 3;
  const int n0 = off[b], nn = off[b + 1] - n0;
  const int p0 = (int)((long)nn * s / POS_CE_SPLITS), p1 = (int)((long)nn * (s + 1) / POS_CE_SPLITS);
  const float* tb = tgt + (long)3 * nb * n0 + (long)c * nn * nb;
  const act_t* xb = xt + (long)n0 * (3 * nb) + c * nb;
  // lanes 0..31 of a half-wave walk the bins of one point; 8 points per block step
  const int j0 = threadIdx.x & 31, g = threadIdx.x >> 5;
  float m = -INFINITY;
  for (int p = p0 + g; p < p1; p += 8)
    for (int j = j0; j < nb; j += 32) m = fmaxf(m, xb[(long)p * (3 * nb) + j]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  double v[3] = {0.0, 0.0, 0.0};
  for (int p = p0 + g; p < p1; p += 8)
    for (int j = j0; j < nb; j += 32) {
      const float x = xb[(long)p * (3 * nb) + j], t = tb[(long)p * nb + j];
      v[0] += exp((double)x - (double)m);
      v[1] += (double)t * x;
      v[2] += t;
    }
  double tot[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double w = wave_sum_d(v[k]);
    if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = w;
    __syncthreads();
    tot[k] = ((redd[0] + redd[1]) + redd[2]) + redd[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* o = part + ((long)bc * POS_CE_SPLITS + s) * POS_CE_PART;
    o[0] = m; o[1] = 0.f;
    double* od = reinterpret_cast<double*>(o + 2);
    od[0] = tot[0]; od[1] = tot[1]; od[2] = tot[2];
  }
}

// fixed-order merge of the slices: stats[bc] = (loss, lse, tsum, lse - (float)lse): backward rebuilds the double lse from [1] + [3]
__global__ __launch_bounds__(64) void pos_ce_merge_kernel(const float* __restrict__ part, float* __restrict__ stats) {
  const int bc = blockIdx.x, s = threadIdx.x;
  float m = -INFINITY;
  double se = 0.0, stx = 0.0, st = 0.0;
  if (s < POS_CE_SPLITS) {
    const float* o = part + ((long)bc * POS_CE_SPLITS + s) * POS_CE_PART;
    const double* od = reinterpret_cast<const double*>(o + 2);
    m = o[0]; se = od[0]; stx = od[1]; st = od[2];
  }
  const float M = wave_max(m);
  se = wave_sum_d(m > -INFINITY ? se * exp((double)m - (double)M) : 0.0);
  stx = wave_sum_d(stx);
  st = wave_sum_d(st);
  if (s == 0) {
    const double lse = (double)M + log(se);
    float* o = stats + (long)bc * 4;
    o[0] = (float)(lse * st - stx);
    o[1] = (float)lse;
    o[2] = (float)st;
    o[3] = (float)(lse - (double)o[1]);
  }
}

// Rotation CE + openness BCE on ae[B][nrot*3 + 1]; gt[B][ga] with rot bins at 3..5 and open at ga-1.
// losses[4] = pos, rot, open, total.  dae = (d rot / d ae | d open / d ae) column-wise (rot logits and the
// open logit are disjoint columns), unscaled by the upstream gradient.
__global__ __launch_bounds__(256) void small_loss_kernel(const act_t* __restrict__ ae, const float* __restrict__ gt,
                                                         const float* __restrict__ pos_stats, int B, int nrot, int ga,
                                                         float pos_w, float rot_w, float* __restrict__ losses,
                                                         float* __restrict__ dae) {
  __shared__ float acc[3];
  const int W = nrot * 3 + 1;
  if (threadIdx.x < 3) acc[threadIdx.x] = 0.f;
  __syncthreads();
  float rot = 0.f, opn = 0.f, pos = 0.f;
  for (int i = threadIdx.x; i < B * 3; i += 256) {
    const int b = i / 3, a = i % 3;
    const act_t* row = ae + (long)b * W;
    float m = -INFINITY;
    for (int k = 0; k < nrot; ++k) m = fmaxf(m, row[k * 3 + a]);
    float se = 0.f;
    for (int k = 0; k < nrot; ++k) se += expf(row[k * 3 + a] - m);
    const float lse = m + logf(se);
    const int tk = (int)gt[(long)b * ga + 3 + a];
    rot += lse - row[tk * 3 + a];
    if (dae)
      for (int k = 0; k < nrot; ++k)
        dae[(long)b * W + k * 3 + a] = (expf(row[k * 3 + a] - lse) - (k == tk ? 1.f : 0.f)) / (B * 3);
    pos += pos_stats[(long)i * 4];
  }
  for (int b = threadIdx.x; b < B; b += 256) {
    const float x = ae[(long)b * W + W - 1], t = gt[(long)b * ga + ga - 1];
    opn += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    if (dae) dae[(long)b * W + W - 1] = (1.f / (1.f + expf(-x)) - t) / B;
  }
  __shared__ float red[3][4];
  float v3[3] = {pos, rot, opn};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float w = wave_sum(v3[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 3; ++k) acc[k] = ((red[k][0] + red[k][1]) + red[k][2]) + red[k][3];
    const float lp = acc[0] / (3.f * B), lr = acc[1] / (3.f * B), lo = acc[2] / B;
    losses[0] = lp; losses[1] = lr; losses[2] = lo; losses[3] = pos_w * lp + rot_w * lr + lo;
  }
}

// dxt[n][c][bin] = coef * (softmax * tsum - t) / (3B), coef = g[0] + pos_w * g[3] (device scalars)
__global__ void pos_ce_bwd_kernel(const act_t* __restrict__ xt, const float* __restrict__ tgt,
                                  const int* __restrict__ off, const int* __restrict__ batch,
                                  const float* __restrict__ stats, const float* __restrict__ gl, float pos_w, int B,
                                  int n, int nb, act_t* __restrict__ dxt) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)n * 3 * nb;
  if (gid >= total) return;
  const int p = (int)(gid / (3 * nb)), rem = (int)(gid % (3 * nb)), c = rem / nb, bin = rem % nb;
  const int b = batch[p];
  const int n0 = off[b], nn = off[b + 1] - n0;
  const float t = tgt[(long)3 * nb * n0 + (long)c * nn * nb + (long)(p - n0) * nb + bin];
  const float* s = stats + (long)(b * 3 + c) * 4;
  const float coef = (gl[0] + pos_w * gl[3]) / (3.f * B);
  dxt[gid] = (float)((double)coef * (exp((double)xt[gid] - ((double)s[1] + (double)s[3])) * (double)s[2] - (double)t));
}

// per-(cloud, axis) upstream gradients g[B*3] (trajectory heads weight each cloud by its step mask,
// motion_planner_ptv3.py:327-336): dxt = g[b*3+c] * (softmax * tsum - t)
__global__ void pos_ce_bwd_w_kernel(const act_t* __restrict__ xt, const float* __restrict__ tgt,
                                    const int* __restrict__ off, const int* __restrict__ batch,
                                    const float* __restrict__ stats, const float* __restrict__ g, int n, int nb,
                                    act_t* __restrict__ dxt) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)n * 3 * nb;
  if (gid >= total) return;
  const int p = (int)(gid / (3 * nb)), rem = (int)(gid % (3 * nb)), c = rem / nb, bin = rem % nb;
  const int b = batch[p];
  const int n0 = off[b], nn = off[b + 1] - n0;
  const float t = tgt[(long)3 * nb * n0 + (long)c * nn * nb + (long)(p - n0) * nb + bin];
  const float* s = stats + (long)(b * 3 + c) * 4;
  dxt[gid] = (float)((double)g[b * 3 + c] * (exp((double)xt[gid] - ((double)s[1] + (double)s[3])) * (double)s[2] - (double)t));
}

// dae_out = dae_saved * (upstream weight of its column): rot columns g[1] + rot_w * g[3], open column g[2] + g[3]
__global__ void ae_grad_kernel(const float* __restrict__ x, const float* __restrict__ gl, float rot_w, int W, long n,
                               act_t* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool open_col = (i % W) == W - 1;
  y[i] = x[i] * (open_col ? gl[2] + gl[3] : gl[1] + rot_w * gl[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// Trajectory losses of the motion planner on the [B, T]-sized tensors, motion_planner_ptv3.py:327-397 (heatmap_disc /
// euler_disc): one block.  Row r = b * T + t.  ae [B*T][W], W = nrot * 3 + 2 (rotation logits (bin, axis) at bin*3+axis,
// openness logit, stop logit); gt [B*T][ga] (rotation bins at 3..5, openness at ga-1); stop [B*T]; mask [B*T];
// ce [B*T][3] = heatmap cross entropy per (cloud, step, axis).
//   pos  = mean_b ( sum_t mask * sum_c ce / (3 * sum_t mask) )        rot = sum mask * CE_rot / sum mask / 3
//   open = sum mask * BCE(open) / sum mask                             stop likewise
// losses[5] = pos, rot, open, stop, total.  dae [B*T][W] and dce [B*T][3] receive the partial derivatives of the loss
// each column belongs to (the columns are disjoint), unscaled by the upstream gradient.
__global__ __launch_bounds__(256) void mp_loss_kernel(const act_t* __restrict__ ae, const float* __restrict__ gt,
                                                      const float* __restrict__ stop, const float* __restrict__ mask,
                                                      const float* __restrict__ ce, int B, int T, int nrot, int ga, float pos_w,
                                                      float rot_w, float* __restrict__ losses, float* __restrict__ dae,
                                                      float* __restrict__ dce) {
  extern __shared__ float msum_b[];  // [B] sum_t mask
  __shared__ float red[4][4];
  __shared__ float msum_s;
  const int W = nrot * 3 + 2, R = B * T, tid = threadIdx.x;
  for (int b = tid; b < B; b += 256) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += mask[b * T + t];
    msum_b[b] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += msum_b[b];
    msum_s = s;
  }
  __syncthreads();
  const float msum = msum_s;
  float pos = 0.f, rot = 0.f, opn = 0.f, stp = 0.f;
  for (int i = tid; i < R * 3; i += 256) {
    const int r = i / 3, a = i % 3, b = r / T;
    const float m = mask[r];
    const act_t* row = ae + (long)r * W;
    float mx = -INFINITY;
    for (int k = 0; k < nrot; ++k) mx = fmaxf(mx, row[k * 3 + a]);
    float se = 0.f;
    for (int k = 0; k < nrot; ++k) se += expf(row[k * 3 + a] - mx);
    const float lse = mx + logf(se);
    const int tk = (int)gt[(long)r * ga + 3 + a];
    rot += (lse - row[tk * 3 + a]) * m;
    const float sc = m / msum / 3.f;
    for (int k = 0; k < nrot; ++k) dae[(long)r * W + k * 3 + a] = (expf(row[k * 3 + a] - lse) - (k == tk ? 1.f : 0.f)) * sc;
    const float cpos = m / (3.f * msum_b[b] * B);
    pos += ce[i] * cpos;
    dce[i] = cpos;
  }
  for (int r = tid; r < R; r += 256) {
    const float m = mask[r], sc = m / msum;
    const float xo = ae[(long)r * W + W - 2], to = gt[(long)r * ga + ga - 1];
    opn += (fmaxf(xo, 0.f) - xo * to + log1pf(expf(-fabsf(xo)))) * m;
    dae[(long)r * W + W - 2] = (1.f / (1.f + expf(-xo)) - to) * sc;
    const float xs = ae[(long)r * W + W - 1], ts = stop[r];
    stp += (fmaxf(xs, 0.f) - xs * ts + log1pf(expf(-fabsf(xs)))) * m;
    dae[(long)r * W + W - 1] = (1.f / (1.f + expf(-xs)) - ts) * sc;
  }
  const float v4[4] = {pos, rot, opn, stp};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float w = wave_sum(v4[k]);
    if ((tid & 63) == 0) red[k][tid >> 6] = w;
  }
  __syncthreads();
  if (tid == 0) {
    float t4[4];
    for (int k = 0; k < 4; ++k) t4[k] = ((red[k][0] + red[k][1]) + red[k][2]) + red[k][3];
    const float lp = t4[0], lr = t4[1] / msum / 3.f, lo = t4[2] / msum, ls = t4[3] / msum;
    losses[0] = lp; losses[1] = lr; losses[2] = lo; losses[3] = ls; losses[4] = pos_w * lp + rot_w * lr + lo + ls;
  }
}
// upstream gradient g[5] (device) of the five losses -> d ae, d ce
__global__ void mp_loss_bwd_kernel(const float* __restrict__ dae, const float* __restrict__ dce, const float* __restrict__ g,
                                   float pos_w, float rot_w, int W, long nae, long nce, act_t* __restrict__ dae_out,
                                   float* __restrict__ dce_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nae) {
    const int col = (int)(i % W);
    const float sc = col == W - 1 ? g[3] + g[4] : col == W - 2 ? g[2] + g[4] : g[1] + rot_w * g[4];
    dae_out[i] = dae[i] * sc;
  }
  if (i < nce) dce_out[i] = dce[i] * (g[0] + pos_w * g[4]);
}

// elementwise helpers ---------------------------------------------------------------------------
__global__ void add_kernel(const act_t* __restrict__ a, const act_t* __restrict__ b, act_t* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 u = ld4q(a, i), v = ld4q(b, i);
    st4q(y, i, make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
  }
}
__global__ void dropout_mask_kernel(const act_t* __restrict__ x, act_t* __restrict__ y, long n,
                                    unsigned long long seed, unsigned thresh, float inv_keep) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = x[i] * dropout_scale(seed, (unsigned long long)i, thresh, inv_keep);
}

// DropPath (stochastic depth, timm.DropPath as the reference's Block uses it on the attention and MLP branches,
// PointTransformerV3/model.py:655-657,666,672): one Bernoulli draw per ROW (the tensor is [points][channels], so a "sample" is a
// point), kept rows scaled by 1 / (1 - p).  y = x + s_row * branch (x optional: the backward pass is the same map without it).
// Stateless like the dropout mask: keep iff hash(seed, row) >= p * 2^32.
__global__ void drop_path_kernel(const act_t* __restrict__ branch, const act_t* __restrict__ x, act_t* __restrict__ y,
                                 long total4, int c4n, unsigned long long seed, unsigned thresh, float inv_keep) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const float s = dropout_scale(seed, (unsigned long long)(i / c4n), thresh, inv_keep);
    float4 v = ld4q(branch, i);
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    if (x) {
      const float4 r = ld4q(x, i);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    st4q(y, i, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Trajectory head of the motion planner (genrobo3d/models/motion_planner_ptv3.py:88-97,113-114): the hidden layer of
// step t is dropout(LeakyReLU(base + bias_t)) where base = x W_x^T is shared by all steps and bias_t carries the step
// embedding.  Forward: one elementwise pass per step.  Backward: dpre_t = dh_t * act'(base + bias_t) * mask, accumulated
// over the steps into dbase, with per-block column sums (-> dbias_t) reduced in fixed order.
#define SA_ROWS 64  // rows per block of the backward pass
__global__ void step_act_fwd_kernel(const act_t* __restrict__ base, const float* __restrict__ bias, act_t* __restrict__ out,
                                    long total4, int c4n, int act, unsigned long long seed, unsigned thresh, float inv_keep) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const float4 b = ld4q(base, i);
    const float4 s = ld4q(bias, i % c4n);
    float v[4] = {b.x + s.x, b.y + s.y, b.z + s.z, b.w + s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = act_f(v[e], act);
      if (thresh) v[e] *= dropout_scale(seed, (unsigned long long)(4 * i + e), thresh, inv_keep);
    }
    st4q(out, i, make_float4(v[0], v[1], v[2], v[3]));
  }
}
// block = 256 threads = (256 / c4n) row lanes x c4n column quads; rows [blockIdx.x * SA_ROWS, +SA_ROWS)
__global__ __launch_bounds__(256) void step_act_bwd_kernel(const act_t* __restrict__ dh, const act_t* __restrict__ base,
                                                           const float* __restrict__ bias, act_t* __restrict__ dacc,
                                                           float* __restrict__ part, int M, int c4n, int act, int accumulate,
                                                           unsigned long long seed, unsigned thresh, float inv_keep) {
  extern __shared__ float4 red[];  // [row lanes][c4n]
  const int q = threadIdx.x % c4n, rl = threadIdx.x / c4n, lanes = 256 / c4n;
  const float4 s = ld4q(bias, q);
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  const int r1 = min(M, (int)(blockIdx.x + 1) * SA_ROWS);
  for (int row = blockIdx.x * SA_ROWS + rl; row < r1; row += lanes) {
    const long i = (long)row * c4n + q;
    const float4 b = ld4q(base, i);
    const float4 g = ld4q(dh, i);
    const float pre[4] = {b.x + s.x, b.y + s.y, b.z + s.z, b.w + s.w};
    float d[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[e] *= act_grad_f(pre[e], act);
      if (thresh) d[e] *= dropout_scale(seed, (unsigned long long)(4 * i + e), thresh, inv_keep);
    }
    cs.x += d[0]; cs.y += d[1]; cs.z += d[2]; cs.w += d[3];
    float4 o = make_float4(d[0], d[1], d[2], d[3]);
    if (accumulate) {
      const float4 a = ld4q(dacc, i);
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    st4q(dacc, i, o);
  }
  red[rl * c4n + q] = cs;
  __syncthreads();
  if (rl == 0) {
    float4 t = red[q];
    for (int k = 1; k < lanes; ++k) {
      const float4 v = red[k * c4n + q];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    st4q(part, (long)blockIdx.x * c4n + q, t);
  }
}
// ---------------------------------------------------------------------------------------------------------------
// Soft position targets and arg-max position decoding on the device (SURVEY.md 8f rank 2;
// genrobo3d/utils/action_position_utils.py:7-46 and :48-64 best='max').  Candidate coordinate of (point n, axis c,
// bin j): xyz[n][c] + (j - pos_bins) * pos_bin_size, evaluated in double exactly like the reference's numpy code.
#define PT_SPLITS 32

// candidate coordinate shift[j] + x with numpy's two roundings (no fma contraction)
__device__ __forceinline__ double cand_coord(int jrel, double bin, double x) {
#pragma clang fp contract(off)
  const double sft = (double)jrel * bin;
  return sft + x;
}

__device__ __forceinline__ double pt_weight(double dist, int kind) {  // kind 0 'plain', 1 'dist'
  if (kind == 0) return dist < 0.01 ? 1.0 : 0.0;
  return dist > 0.01 ? 0.0 : 1.0 / fmax(dist, 1e-4);
}

// first-index arg-min / arg-max pairs
__device__ __forceinline__ void pick_min(double& v, long& i, double v2, long i2) {
  if (v2 < v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
__device__ __forceinline__ void pick_max(float& v, long& i, float v2, long i2) {
  if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// slice s of (cloud b, axis c): (sum of weights, nearest candidate distance, its index n * nb + j)
__global__ __launch_bounds__(256) void pos_tgt_part_kernel(const float* __restrict__ pc, long ld, const int* __restrict__ off,
                                                           const float* __restrict__ gt, int ga,
                                                           const unsigned char* __restrict__ robot, int nb, double bin,
                                                           int kind, double* __restrict__ part) {
  __shared__ double rs[4], rv[4];
  __shared__ long ri[4];
  const int s = blockIdx.x, bc = blockIdx.y, b = bc / 3, c = bc % 3;
  const int n0 = off[b], nn = off[b + 1] - n0;
  const int p0 = (int)((long)nn * s / PT_SPLITS), p1 = (int)((long)nn * (s + 1) / PT_SPLITS);
  const double g = (double)gt[(long)b * ga + c];
  const int j0 = threadIdx.x & 31, grp = threadIdx.x >> 5, pb = nb / 2;
  double sum = 0.0, best = INFINITY;
  long bi = 0x7fffffffffffffffL;
  for (int p = p0 + grp; p < p1; p += 8) {
    const double x = (double)pc[(long)(n0 + p) * ld + c];
    const bool rob = robot && robot[n0 + p];
    for (int j = j0; j < nb; j += 32) {
      const double dist = fabs(g - cand_coord(j - pb, bin, x));
      if (!rob) sum += pt_weight(dist, kind);
      pick_min(best, bi, dist, (long)p * nb + j);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    const double v2 = __shfl_xor(best, o, 64);
    const long i2 = __shfl_xor(bi, o, 64);
    pick_min(best, bi, v2, i2);
  }
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = sum; rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    for (int w = 1; w < 4; ++w) pick_min(rv[0], ri[0], rv[w], ri[w]);
    double* o = part + ((long)bc * PT_SPLITS + s) * 3;
    o[0] = t; o[1] = rv[0]; o[2] = (double)ri[0];
  }
}

// stats[bc] = (total weight, index of the nearest candidate)   (slices in fixed order)
__global__ __launch_bounds__(64) void pos_tgt_merge_kernel(const double* __restrict__ part, double* __restrict__ stats) {
  const int bc = blockIdx.x;
  if (threadIdx.x) return;
  double sum = 0.0, best = INFINITY;
  long bi = 0x7fffffffffffffffL;
  for (int s = 0; s < PT_SPLITS; ++s) {
    const double* o = part + ((long)bc * PT_SPLITS + s) * 3;
    sum += o[0];
    pick_min(best, bi, o[1], (long)o[2]);
  }
  stats[bc * 2] = sum;
  stats[bc * 2 + 1] = (double)bi;
}

// tgt: per cloud [3][nn * nb] normalised weights (or the one-hot nearest candidate when an axis has no weight)
__global__ void pos_tgt_write_kernel(const float* __restrict__ pc, long ld, const int* __restrict__ off,
                                     const int* __restrict__ batch, const float* __restrict__ gt, int ga,
                                     const unsigned char* __restrict__ robot, int n, int nb, double bin, int kind,
                                     const double* __restrict__ stats, float* __restrict__ tgt) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * 3 * nb) return;
  const int j = (int)(gid % nb), c = (int)((gid / nb) % 3), p = (int)(gid / (3L * nb));
  const int b = batch[p], n0 = off[b], nn = off[b + 1] - n0, pl = p - n0;
  const double dist = fabs((double)gt[(long)b * ga + c] - cand_coord(j - nb / 2, bin, (double)pc[(long)p * ld + c]));
  const double sum = stats[(b * 3 + c) * 2];
  double w;
  if (sum == 0.0) w = ((long)pl * nb + j == (long)stats[(b * 3 + c) * 2 + 1]) ? 1.0 : 0.0;
  else w = (robot && robot[p]) ? 0.0 : pt_weight(dist, kind);
  const float out = kind == 0 ? (float)w / (float)(sum == 0.0 ? 1.0 : sum) : (float)(w / (sum == 0.0 ? 1.0 : sum));
  tgt[(long)3 * nb * n0 + (long)c * nn * nb + (long)pl * nb + j] = out;
}

// slice s of (cloud b, axis c): first arg-max of the position logits xt[n][c][j]
__global__ __launch_bounds__(256) void pos_argmax_part_kernel(const act_t* __restrict__ xt, const int* __restrict__ off, int nb,
                                                              double* __restrict__ part) {
  __shared__ float rv[4];
  __shared__ long ri[4];
  const int s = blockIdx.x, bc = blockIdx.y, b = bc / 3, c = bc % 3;
  const int n0 = off[b], nn = off[b + 1] - n0;
  const int p0 = (int)((long)nn * s / PT_SPLITS), p1 = (int)((long)nn * (s + 1) / PT_SPLITS);
  const int j0 = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float best = -INFINITY;
  long bi = 0x7fffffffffffffffL;
  for (int p = p0 + grp; p < p1; p += 8)
    for (int j = j0; j < nb; j += 32) pick_max(best, bi, xt[(long)(n0 + p) * (3 * nb) + c * nb + j], (long)p * nb + j);
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(best, o, 64);
    const long i2 = __shfl_xor(bi, o, 64);
    pick_max(best, bi, v2, i2);
  }
  if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) pick_max(rv[0], ri[0], rv[w], ri[w]);
    double* o = part + ((long)bc * PT_SPLITS + s) * 3;
    o[0] = (double)rv[0]; o[1] = (double)ri[0];
  }
}

__global__ __launch_bounds__(64) void pos_argmax_merge_kernel(const double* __restrict__ part, const float* __restrict__ pc,
                                                              long ld, const int* __restrict__ off, int nb, double bin,
                                                              double* __restrict__ best_pos) {
  const int bc = blockIdx.x, b = bc / 3, c = bc % 3;
  if (threadIdx.x) return;
  float best = -INFINITY;
  long bi = 0x7fffffffffffffffL;
  for (int s = 0; s < PT_SPLITS; ++s) {
    const double* o = part + ((long)bc * PT_SPLITS + s) * 3;
    pick_max(best, bi, (float)o[0], (long)o[1]);
  }
  if (off[b + 1] == off[b]) { best_pos[bc] = 0.0; return; }
  const long p = bi / nb;
  const int j = (int)(bi % nb);
  best_pos[bc] = cand_coord(j - nb / 2, bin, (double)pc[(long)(off[b] + p) * ld + c]);
}

extern "C" {

int lotus_pool_max_fwd(const act_t* x, const int* members, const int* seg, int nc, int C, act_t* y, int* arg,
                       void* stream) {
  LOTUS_CHECK_ARG(x && members && seg && y && arg && C % 4 == 0, "lotus_pool_max_fwd: bad arguments");
  if (nc == 0) return LOTUS_OK;
  LOTUS_LAUNCH(pool_max_fwd_kernel, dim3(cdiv((long)nc * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     members, seg, nc, C, y, arg);
  LOTUS_LAUNCH_CHECK("lotus_pool_max_fwd");
  return LOTUS_OK;
}
int lotus_pool_max_bwd(const act_t* dy, const int* arg, const int* cluster, int n, int C, act_t* dx, void* stream) {
  LOTUS_CHECK_ARG(dy && arg && cluster && dx && C % 4 == 0, "lotus_pool_max_bwd: bad arguments");
  if (n == 0) return LOTUS_OK;
  LOTUS_LAUNCH(pool_max_bwd_kernel, dim3(cdiv((long)n * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, arg,
                     cluster, n, C, dx);
  LOTUS_LAUNCH_CHECK("lotus_pool_max_bwd");
  return LOTUS_OK;
}
int lotus_unpool_fwd(const act_t* skip, const act_t* up, const int* cluster, int n, int C, act_t* x, void* stream) {
  LOTUS_CHECK_ARG(skip && up && cluster && x && C % 4 == 0, "lotus_unpool_fwd: bad arguments");
  if (n == 0) return LOTUS_OK;
  LOTUS_LAUNCH(unpool_fwd_kernel, dim3(cdiv((long)n * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, skip, up,
                     cluster, n, C, x);
  LOTUS_LAUNCH_CHECK("lotus_unpool_fwd");
  return LOTUS_OK;
}
int lotus_unpool_bwd(const act_t* dx, const int* members, const int* seg, int nc, int C, act_t* dup, void* stream) {
  LOTUS_CHECK_ARG(dx && members && seg && dup && C % 4 == 0, "lotus_unpool_bwd: bad arguments");
  if (nc == 0) return LOTUS_OK;
  LOTUS_LAUNCH(unpool_bwd_kernel, dim3(cdiv((long)nc * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, dx,
                     members, seg, nc, C, dup);
  LOTUS_LAUNCH_CHECK("lotus_unpool_bwd");
  return LOTUS_OK;
}
// per-cloud max over the contiguous row ranges [off[b], off[b+1])
size_t lotus_cloud_max_workspace(int B, int C) { return (size_t)B * CM_SPLITS * C * (sizeof(float) + sizeof(int)); }
int lotus_cloud_max_fwd(const act_t* x, const int* off, int B, int C, act_t* y, int* arg, void* workspace, size_t workspace_bytes,
                        void* stream) {
  LOTUS_CHECK_ARG(x && off && y && arg && B > 0 && C > 0 && C % 4 == 0 && ((uintptr_t)x) % 16 == 0,
                  "lotus_cloud_max_fwd: bad arguments (C must be a multiple of 4, x 16-byte aligned)");
  LOTUS_CHECK_ARG(workspace && ((uintptr_t)workspace) % 16 == 0 && workspace_bytes >= lotus_cloud_max_workspace(B, C),
                  "lotus_cloud_max_fwd: workspace too small");
  float* pv = (float*)workspace;
  int* pi = (int*)(pv + (size_t)B * CM_SPLITS * C);
  hipStream_t st = (hipStream_t)stream;
  LOTUS_LAUNCH(cloud_max_part_kernel, dim3(B, cdiv(C, 128), CM_SPLITS), dim3(1024), 0, st, x, off, C, pv, pi);
  LOTUS_LAUNCH(cloud_max_merge_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, st, (const float*)pv, (const int*)pi, B, C, y, arg);
  LOTUS_LAUNCH_CHECK("lotus_cloud_max_fwd");
  return LOTUS_OK;
}
int lotus_cloud_max_bwd(const act_t* dy, const int* arg, const int* batch, int n, int C, const act_t* add, act_t* dx,
                        void* stream) {
  LOTUS_CHECK_ARG(dy && arg && batch && dx && C % 4 == 0, "lotus_cloud_max_bwd: bad arguments");
  if (n == 0) return LOTUS_OK;
  LOTUS_LAUNCH(cloud_max_bwd_kernel, dim3(cdiv((long)n * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, arg,
                     batch, n, C, add, dx);
  LOTUS_LAUNCH_CHECK("lotus_cloud_max_bwd");
  return LOTUS_OK;
}

// floats the caller must provide as pos_stats: [B*3][4] statistics + the slice partials behind them
size_t lotus_loss_stats_floats(int B) { return (size_t)B * 3 * (4 + POS_CE_PART * POS_CE_SPLITS); }

// losses[4] = (pos, rot, open, total); pos_stats (lotus_loss_stats_floats(B) floats; the leading [B*3][4] are
// what backward reads) and dae [B][nrot*3+1] are saved for backward.
int lotus_loss_fwd(const act_t* xt, const act_t* ae, const float* tgt, const float* gt, const int* off, int B, int nb,
                   int nrot, int ga, float pos_w, float rot_w, float* losses, float* pos_stats, float* dae,
                   void* stream) {
  LOTUS_CHECK_ARG(xt && ae && tgt && gt && off && losses && pos_stats && B > 0, "lotus_loss_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  float* part = pos_stats + (size_t)B * 3 * 4;  // slice partials live behind the per-(cloud, axis) stats
  LOTUS_LAUNCH(pos_ce_part_kernel, dim3(POS_CE_SPLITS, B * 3), dim3(256), 0, st, xt, tgt, off, nb, part);
  LOTUS_LAUNCH(pos_ce_merge_kernel, dim3(B * 3), dim3(64), 0, st, (const float*)part, pos_stats);
  LOTUS_LAUNCH(small_loss_kernel, dim3(1), dim3(256), 0, st, ae, gt, pos_stats, B, nrot, ga, pos_w, rot_w, losses, dae);
  LOTUS_LAUNCH_CHECK("lotus_loss_fwd");
  return LOTUS_OK;
}
// dxt [n][3*nb] and dae_out [B][nrot*3+1] from the upstream gradient of the 4 losses (gl, device).
int lotus_loss_bwd(const act_t* xt, const float* tgt, const int* off, const int* batch, const float* pos_stats,
                   const float* dae_saved, const float* gl, float pos_w, float rot_w, int B, int n, int nb, int nrot,
                   act_t* dxt, act_t* dae_out, void* stream) {
  LOTUS_CHECK_ARG(xt && tgt && off && batch && pos_stats && dae_saved && gl && dxt && dae_out, "lotus_loss_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)n * 3 * nb;
  LOTUS_LAUNCH(pos_ce_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, xt, tgt, off, batch, pos_stats, gl,
                     pos_w, B, n, nb, dxt);
  const int W = nrot * 3 + 1;
  LOTUS_LAUNCH(ae_grad_kernel, dim3(cdiv((long)B * W, 256)), dim3(256), 0, st, dae_saved, gl, rot_w, W,
                     (long)B * W, dae_out);
  LOTUS_LAUNCH_CHECK("lotus_loss_bwd");
  return LOTUS_OK;
}

// Heatmap cross entropy alone (trajectory heads call it once per step): pos_stats[(b*3+c)*4] = CE of cloud b, axis c
int lotus_pos_ce_fwd(const act_t* xt, const float* tgt, const int* off, int B, int nb, float* pos_stats, void* stream) {
  LOTUS_CHECK_ARG(xt && tgt && off && pos_stats && B > 0 && nb > 0, "lotus_pos_ce_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  float* part = pos_stats + (size_t)B * 3 * 4;
  LOTUS_LAUNCH(pos_ce_part_kernel, dim3(POS_CE_SPLITS, B * 3), dim3(256), 0, st, xt, tgt, off, nb, part);
  LOTUS_LAUNCH(pos_ce_merge_kernel, dim3(B * 3), dim3(64), 0, st, (const float*)part, pos_stats);
  LOTUS_LAUNCH_CHECK("lotus_pos_ce_fwd");
  return LOTUS_OK;
}
int lotus_pos_ce_bwd(const act_t* xt, const float* tgt, const int* off, const int* batch, const float* pos_stats,
                     const float* g, int B, int n, int nb, act_t* dxt, void* stream) {
  LOTUS_CHECK_ARG(xt && tgt && off && batch && pos_stats && g && dxt && B > 0, "lotus_pos_ce_bwd: bad arguments");
  if (n == 0) return LOTUS_OK;
  const long total = (long)n * 3 * nb;
  LOTUS_LAUNCH(pos_ce_bwd_w_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, xt, tgt, off, batch,
                     pos_stats, g, n, nb, dxt);
  LOTUS_LAUNCH_CHECK("lotus_pos_ce_bwd");
  return LOTUS_OK;
}

// Trajectory losses of the motion planner ([B, T]-sized tensors), see mp_loss_kernel.
int lotus_mp_loss_fwd(const act_t* ae, const float* gt, const float* stop, const float* mask, const float* ce, int B, int T,
                      int nrot, int ga, float pos_w, float rot_w, float* losses, float* dae, float* dce, void* stream) {
  LOTUS_CHECK_ARG(ae && gt && stop && mask && ce && losses && dae && dce && B > 0 && T > 0 && nrot > 0 && ga >= 7 && B <= 8192,
                  "lotus_mp_loss_fwd: bad arguments");
  LOTUS_LAUNCH(mp_loss_kernel, dim3(1), dim3(256), (size_t)B * sizeof(float), (hipStream_t)stream, ae, gt, stop, mask, ce, B, T,
               nrot, ga, pos_w, rot_w, losses, dae, dce);
  LOTUS_LAUNCH_CHECK("lotus_mp_loss_fwd");
  return LOTUS_OK;
}
int lotus_mp_loss_bwd(const float* dae, const float* dce, const float* g, float pos_w, float rot_w, int B, int T, int nrot,
                      act_t* dae_out, float* dce_out, void* stream) {
  LOTUS_CHECK_ARG(dae && dce && g && dae_out && dce_out && B > 0 && T > 0 && nrot > 0, "lotus_mp_loss_bwd: bad arguments");
  const int W = nrot * 3 + 2;
  const long nae = (long)B * T * W, nce = (long)B * T * 3;
  LOTUS_LAUNCH(mp_loss_bwd_kernel, dim3(cdiv(nae, 256)), dim3(256), 0, (hipStream_t)stream, dae, dce, g, pos_w, rot_w, W, nae, nce,
               dae_out, dce_out);
  LOTUS_LAUNCH_CHECK("lotus_mp_loss_bwd");
  return LOTUS_OK;
}

int lotus_add(const act_t* a, const act_t* b, act_t* y, long n, void* stream) {
  LOTUS_CHECK_ARG(a && b && y && n % 4 == 0, "lotus_add: bad arguments");
  if (n == 0) return LOTUS_OK;
  int g = cdiv(n / 4, 256);
  LOTUS_LAUNCH(add_kernel, dim3(g > 4096 ? 4096 : g), dim3(256), 0, (hipStream_t)stream, a, b, y, n / 4);
  LOTUS_LAUNCH_CHECK("lotus_add");
  return LOTUS_OK;
}
// y = x * dropmask(seed, p) / (1 - p): forward dropout and its backward (same mask from the same seed)
int lotus_dropout(const act_t* x, act_t* y, long n, float p, unsigned long long seed, void* stream) {
  LOTUS_CHECK_ARG(x && y && p >= 0.f && p < 1.f, "lotus_dropout: bad arguments");
  if (n == 0) return LOTUS_OK;
  unsigned th; float inv;
  lotus_drop_setup(p, &th, &inv);
  int g = cdiv(n, 256);
  LOTUS_LAUNCH(dropout_mask_kernel, dim3(g > 4096 ? 4096 : g), dim3(256), 0, (hipStream_t)stream, x, y, n, seed, th, inv);
  LOTUS_LAUNCH_CHECK("lotus_dropout");
  return LOTUS_OK;
}

// y = x + droppath(branch) (x may be null): rows [M] of C channels, C % 4 == 0
int lotus_drop_path(const act_t* branch, const act_t* x, act_t* y, int M, int C, float p, unsigned long long seed, void* stream) {
  LOTUS_CHECK_ARG(branch && y && M >= 0 && C > 0 && C % 4 == 0 && p >= 0.f && p < 1.f, "lotus_drop_path: bad arguments");
  if (M == 0) return LOTUS_OK;
  unsigned th; float inv;
  lotus_drop_setup(p, &th, &inv);
  const long total4 = (long)M * (C / 4);
  int g = cdiv(total4, 256);
  LOTUS_LAUNCH(drop_path_kernel, dim3(g > 4096 ? 4096 : g), dim3(256), 0, (hipStream_t)stream, branch, x, y, total4, C / 4, seed, th, inv);
  LOTUS_LAUNCH_CHECK("lotus_drop_path");
  return LOTUS_OK;
}

static void drop_params(float p, unsigned* th, float* inv) { lotus_drop_setup(p, th, inv); }

int lotus_step_act_fwd(const act_t* base, const float* bias, act_t* out, int M, int C, int act, float drop_p,
                       unsigned long long drop_seed, void* stream) {
  LOTUS_CHECK_ARG(base && bias && out && M >= 0 && C > 0 && C % 4 == 0 && drop_p >= 0.f && drop_p < 1.f, "lotus_step_act_fwd: bad arguments");
  if (M == 0) return LOTUS_OK;
  unsigned th; float inv;
  drop_params(drop_p, &th, &inv);
  const long total4 = (long)M * C / 4;
  const int g = cdiv(total4, 256);
  LOTUS_LAUNCH(step_act_fwd_kernel, dim3(g > 8192 ? 8192 : g), dim3(256), 0, (hipStream_t)stream, base, bias, out, total4, C / 4,
                     act, drop_seed, th, inv);
  LOTUS_LAUNCH_CHECK("lotus_step_act_fwd");
  return LOTUS_OK;
}

size_t lotus_step_act_bwd_workspace(int M, int C) { return (size_t)cdiv(M > 0 ? M : 1, SA_ROWS) * C * sizeof(float); }

int lotus_step_act_bwd(const act_t* dh, const act_t* base, const float* bias, act_t* dbase, float* dbias, int M, int C, int act,
                       float drop_p, unsigned long long drop_seed, int accumulate, void* workspace, size_t workspace_bytes,
                       void* stream) {
  LOTUS_CHECK_ARG(dh && base && bias && dbase && dbias && M >= 0 && C > 0 && C % 4 == 0 && 256 % (C / 4) == 0 && drop_p >= 0.f && drop_p < 1.f,
                  "lotus_step_act_bwd: bad arguments (C / 4 must divide 256)");
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= lotus_step_act_bwd_workspace(M, C), "lotus_step_act_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) { (void)hipMemsetAsync(dbias, 0, C * sizeof(float), st); return LOTUS_OK; }
  unsigned th; float inv;
  drop_params(drop_p, &th, &inv);
  const int nb = cdiv(M, SA_ROWS), c4n = C / 4;
  LOTUS_LAUNCH(step_act_bwd_kernel, dim3(nb), dim3(256), 256 * sizeof(float4), st, dh, base, bias, dbase, (float*)workspace, M,
                     c4n, act, accumulate, drop_seed, th, inv);
  LOTUS_LAUNCH_CHECK("lotus_step_act_bwd");
  // column sums of the per-block partials: the parallel fixed-order reduction of gemm.hip (one thread per column over
  // all nb partials took 226 us per step head, 1.1 ms of a motion-planner step)
  return lotus_reduce_parts((const float*)workspace, dbias, (long)C, (long)C, nb, 0, st);
}

size_t lotus_pos_workspace(int B) { return (size_t)B * 3 * (PT_SPLITS * 3 + 2) * sizeof(double); }

// tgt (per cloud [3][nn * nb], the layout lotus_loss_fwd consumes) from the point coordinates (first 3 columns of pc)
// and the ground-truth positions gt[b][0..2].  kind 0 'plain', 1 'dist'; robot (optional) = 1 for points to exclude.
int lotus_pos_targets(const float* pc, long ld, const int* off, const int* batch, const float* gt, int ga,
                      const unsigned char* robot, int B, int n, int nb, double bin_size, int kind, float* tgt, void* workspace,
                      size_t workspace_bytes, void* stream) {
  LOTUS_CHECK_ARG(pc && off && batch && gt && tgt && B > 0 && nb > 0 && nb % 2 == 0 && (kind == 0 || kind == 1),
                  "lotus_pos_targets: bad arguments");
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= lotus_pos_workspace(B), "lotus_pos_targets: workspace too small");
  if (n == 0) return LOTUS_OK;
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace;
  double* stats = part + (size_t)B * 3 * PT_SPLITS * 3;
  LOTUS_LAUNCH(pos_tgt_part_kernel, dim3(PT_SPLITS, B * 3), dim3(256), 0, st, pc, ld, off, gt, ga, robot, nb, bin_size, kind, part);
  LOTUS_LAUNCH(pos_tgt_merge_kernel, dim3(B * 3), dim3(64), 0, st, (const double*)part, stats);
  LOTUS_LAUNCH(pos_tgt_write_kernel, dim3(cdiv((long)n * 3 * nb, 256)), dim3(256), 0, st, pc, ld, off, batch, gt, ga, robot, n,
                     nb, bin_size, kind, (const double*)stats, tgt);
  LOTUS_LAUNCH_CHECK("lotus_pos_targets");
  return LOTUS_OK;
}

// best_pos[b][c] (double) = coordinate of the first arg-max of the position logits xt [n][3 * nb] of cloud b, axis c
int lotus_pos_decode_max(const act_t* xt, const float* pc, long ld, const int* off, int B, int nb, double bin_size,
                         double* best_pos, void* workspace, size_t workspace_bytes, void* stream) {
  LOTUS_CHECK_ARG(xt && pc && off && best_pos && B > 0 && nb > 0 && nb % 2 == 0, "lotus_pos_decode_max: bad arguments");
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= lotus_pos_workspace(B), "lotus_pos_decode_max: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace;
  LOTUS_LAUNCH(pos_argmax_part_kernel, dim3(PT_SPLITS, B * 3), dim3(256), 0, st, xt, off, nb, part);
  LOTUS_LAUNCH(pos_argmax_merge_kernel, dim3(B * 3), dim3(64), 0, st, (const double*)part, pc, ld, off, nb, bin_size, best_pos);
  LOTUS_LAUNCH_CHECK("lotus_pos_decode_max");
  return LOTUS_OK;
}

}  // extern "C"

}  // namespace LOTUS_NS
