// lotus-hip: tile attention forward/backward in exact fp32 (MFMA 32x32x2) — one kernel pair
// serves both attention sites of the reference:
//   * SerializedAttention (PointTransformerV3/model.py:468-557, flash var-len path): patches of
//     <= 128 serialised points; q/k/v rows are gathered from qkv[N][3C] through order[pad]; the
//     result is scattered back through the owner positions (unpad[inverse]).
//   * CrossAttention (PointTransformerV3/model_ca.py:46-101): a 128-row tile of one cloud's
//     points attends to that cloud's <= 128 instruction tokens.
// q and k get a per-head LayerNorm(d, eps=1e-6) (qk_norm) inside the kernel.  The reference runs
// the softmax core in fp16 (model.py:544); here it is fp32 end-to-end, which is what the 1e-4
// logit parity against the fp32-ideal oracle requires (SURVEY.md Trap 2).
//
// LDS plan (floats): row images Q[128][33] K[128][33] V[128][33] (+ dO[128][33] in backward) — 51 / 68 KB.  There is NO
// score image: scores are computed transposed (S^T = K Q^T) so that a lane owns one query and its accumulator registers
// run over the keys; softmax, dropout and the P / dS operands of the following products live in registers (see the
// comments at the kernels).  Everything a (tile, head) needs stays on chip between the QK^T, softmax and PV stages.
#include "mma.h"
#include <stdlib.h>

namespace LOTUS_NS {

#define AT 128        // tile rows (queries) and max keys
#define ALD 33        // row stride of the [128][d<=32] images
#define SLD 129       // row stride of the score image

struct AttnP {
  const act_t* q; long q_ld; int q_off;
  const act_t* kv; long kv_ld; int k_off, v_off;
  const int* qidx;    // row gather for the q side (null = identity)
  const int* kidx;    // row gather for the k/v side (null = identity)
  const int* owner;   // per q position: write/use this row (null = all)
  const int* tiles;   // [ntiles][4] = q_start, q_len, k_start, k_len
  const int* blocks;  // bwd: [nblocks][6] = first_tile, n_tiles, tile_step, part_slot, k_start, k_len
  const float* qn_w; const float* qn_b; const float* kn_w; const float* kn_b;
  act_t* out; long out_ld;   // fwd output rows (indexed like q rows)
  float* lse;                // [npos][H]
  // backward
  const act_t* dout;         // gradient of out (same indexing as out)
  act_t* dq; long dq_ld; int dq_off;
  act_t* dkv; long dkv_ld; int dk_off, dv_off; long dkv_part_stride;
  float* ln_part;            // [nblocks * H][4][32]  dgamma_q, dbeta_q, dgamma_k, dbeta_k
  int atomic_out;            // 1: atomicAdd into dq/dkv (kept for generality; unused by the model)
  const int* kext;           // per k position: -1 = owner row, else row of dkv_extra (borrowed copy of a tail patch)
  act_t* dkv_extra;          // [n_extra][dkv_extra_ld] k | v gradients of the borrowed copies
  long dkv_extra_ld;
  int H, d;
  float scale, eps;
  // dropout on the attention probabilities (flash-attn dropout_p, model.py:547 / model_ca.py:64)
  unsigned long long drop_seed;
  unsigned drop_thresh;
  float drop_inv_keep;
};

// Attention-probability dropout: keep iff mix(seed, idx) >= thresh, idx = ((tile * H + h) * 128 + q) * 128 + key.  The
// index is split as (hi, lo): the hi / seed terms are hoisted by the callers (c2 = hi * 0x7FEB352D + seed_hi,
// s0 = seed_lo) and only the low word changes inside a tile.  One multiply-xorshift round after the affine step
// (8 VALU ops; the P-phase of these kernels is VALU-issue bound, the full murmur finaliser costs 11).
__device__ __forceinline__ bool keep_lo(unsigned lo, unsigned s0, unsigned c2, unsigned thresh) {
  unsigned x = lo * 0x9E3779B1u + s0;
  x ^= c2;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
  return x >= thresh;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// Load `len` rows (gathered through idx) of d floats at column offset coff into img[128][ALD];
// rows >= len and columns >= d are zero.  8 threads per row, float4 each.
__device__ __forceinline__ void load_rows(float* img, const act_t* base, long ld, int coff, const int* rows_s,
                                          int len, int d) {
  const int sub = threadIdx.x & 7;
  for (int r = threadIdx.x >> 3; r < AT; r += 32) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < len && sub * 4 < d) v = ld4(base + (long)rows_s[r] * ld + coff + sub * 4);
    float* o = img + r * ALD + sub * 4;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
}

// The same for N images at once with ALL global loads issued before the first LDS store: load_rows called N times
// runs N x 4 load -> s_waitcnt -> store round trips back to back (measured: ~13 serialised HBM latencies = most of a
// forward block's 23 us); here they are 4 N independent loads in flight.
struct RowSrc {
  float* img;
  const act_t* base;
  long ld;
  int coff;
  const int* rows_s;
  int len;
};
template <int N>
__device__ __forceinline__ void load_rows_batch(const RowSrc (&src)[N], int d) {
  const int sub = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  float4 v[N][4];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + 32 * j;
      v[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < src[i].len && sub * 4 < d)
        v[i][j] = ld4(src[i].base + (long)src[i].rows_s[r] * src[i].ld + src[i].coff + sub * 4);
    }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* o = src[i].img + (r0 + 32 * j) * ALD + sub * 4;
      o[0] = v[i][j].x; o[1] = v[i][j].y; o[2] = v[i][j].z; o[3] = v[i][j].w;
    }
}

// Per-row LayerNorm over d of img rows [0, len): img <- xhat (AFFINE=false) or xhat*g+b.
template <bool AFFINE>
__device__ __forceinline__ void ln_rows(float* img, float* rstd_s, int row, int len, int d, float eps,
                                        const float* g_s, const float* b_s) {
  if (row >= len) {
    if (row < AT) rstd_s[row] = 0.f;
    return;
  }
  float* x = img + row * ALD;
  float m = 0.f;
  for (int j = 0; j < d; ++j) m += x[j];
  m /= d;
  float v = 0.f;
  for (int j = 0; j < d; ++j) {
    const float c = x[j] - m;
    v += c * c;
  }
  const float rs = rsqrtf(v / d + eps);
  rstd_s[row] = rs;
  for (int j = 0; j < d; ++j) {
    const float xh = (x[j] - m) * rs;
    x[j] = AFFINE ? xh * g_s[j] + b_s[j] : xh;
  }
}

struct AttnSmem {
  float* Q; float* K; float* V; float* dO; float* S;
  float* qrstd; float* krstd; float* Dv; float* lse;
  float* gq; float* bq; float* gk; float* bk;
  int* qrow; int* krow; int* qown; int* kext;
};

__device__ __forceinline__ AttnSmem carve(float* base, bool bwd) {
  AttnSmem s;
  float* p = base;
  s.Q = p; p += AT * ALD;
  s.K = p; p += AT * ALD;
  s.V = p; p += AT * ALD;
  s.dO = p; if (bwd) p += AT * ALD;
  s.S = nullptr;  // scores live in registers (transposed-accumulator formulation)
  s.qrstd = p; p += AT;
  s.krstd = p; p += AT;
  s.Dv = p; if (bwd) p += AT;   // (backward only: the forward block must stay under 160 KB / 3)
  s.lse = p; if (bwd) p += AT;
  s.gq = p; p += 32; s.bq = p; p += 32; s.gk = p; p += 32; s.bk = p; p += 32;
  s.qrow = (int*)p; p += AT;
  s.krow = (int*)p; p += AT;
  s.qown = (int*)p; p += AT;
  s.kext = (int*)p; if (bwd) p += AT;
  return s;
}
static size_t attn_smem_bytes(bool bwd) {
  // forward: 3 row images + 2 + 3 small arrays = 53 760 B -> THREE blocks per CU (with the backward-only arrays it was
  // 55 296 B: 3 x 55 296 > 160 KB, i.e. two)
  return (size_t)((bwd ? 4 : 3) * AT * ALD + (bwd ? 4 : 2) * AT + 4 * 32 + (bwd ? 4 : 3) * AT) * sizeof(float);
}

__device__ __forceinline__ void load_affine(const AttnP& p, AttnSmem& s) {
  const int t = threadIdx.x;
  if (t < 32) {
    const bool in = t < p.d;
    s.gq[t] = in ? p.qn_w[t] : 0.f;
    s.bq[t] = in ? p.qn_b[t] : 0.f;
    s.gk[t] = in ? p.kn_w[t] : 0.f;
    s.bk[t] = in ? p.kn_b[t] : 0.f;
  }
}

// ------------------------------------------------------------------------------------ forward
// Scores are computed TRANSPOSED, S^T = K Q^T, so that in the MFMA accumulator layout a lane owns one query
// (col = lane & 31) and its registers run over the keys (row = (r & 3) + 8 (r >> 2) + 4 hh of key tile t):
//   * the softmax of a query is a reduction over the lane's own 64 registers plus one exchange with lane ^ 32;
//   * the probabilities are ALREADY the B fragments of O^T = V^T P^T (k = key: lane-half hh of step (t, r)
//     supplies exactly the key it holds in register r; the A fragment V[key][dcol] comes from the LDS image);
// the 128 x 128 score image never exists in LDS (3 row images = 51 KB -> three blocks per CU).
// ---- bf16 operand paths (PREC 1: bf16; PREC 3: bf16x3 split, see gemm.hip).  v_mfma_f32_32x32x16_bf16: a lane holds 8
// k-values (k = 8 * (lane >> 5) + 0..7) of its row / column; the C layout equals the fp32 32x32 form, so the softmax and
// the transposed-accumulator trick are unchanged.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 abf16x8;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 abf16x2;
__device__ __forceinline__ unsigned apack_bf16(float a, float b) {
  const abf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
template <int PREC>
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = apack_bf16(v[2 * q], v[2 * q + 1]);
    if (PREC == 3)
      l[q] = apack_bf16(v[2 * q] - __builtin_bit_cast(float, h[q] << 16), v[2 * q + 1] - __builtin_bit_cast(float, h[q] & 0xffff0000u));
    else
      l[q] = 0u;
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// acc += A * B with A = (ah, al), B = (bh, bl): lo*hi + hi*lo + hi*hi (PREC 3) or hi*hi (PREC 1)
template <int PREC>
__device__ __forceinline__ f32x16 mfma_split(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16 acc) {
  if (PREC == 3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, al), __builtin_bit_cast(abf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, ah), __builtin_bit_cast(abf16x8, bl), acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, ah), __builtin_bit_cast(abf16x8, bh), acc, 0, 0, 0);
}

template <int PREC>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  AttnSmem s = carve(smem, false);
  const int tid = threadIdx.x, wave = tid >> 6, l31 = tid & 31, hh = (tid >> 5) & 1;
  const int h = blockIdx.y, d = p.d;
  const int q_start = p.tiles[blockIdx.x * 4 + 0], q_len = p.tiles[blockIdx.x * 4 + 1];
  const int k_start = p.tiles[blockIdx.x * 4 + 2], k_len = p.tiles[blockIdx.x * 4 + 3];

  if (tid < AT) {
    s.qrow[tid] = tid < q_len ? (p.qidx ? p.qidx[q_start + tid] : q_start + tid) : -1;
    s.krow[tid] = tid < k_len ? (p.kidx ? p.kidx[k_start + tid] : k_start + tid) : -1;
    s.qown[tid] = tid < q_len ? (p.owner ? p.owner[q_start + tid] : 1) : 0;
  }
  load_affine(p, s);
  __syncthreads();
  {
    const RowSrc src[3] = {{s.Q, p.q, p.q_ld, p.q_off + h * d, s.qrow, q_len}, {s.K, p.kv, p.kv_ld, p.k_off + h * d, s.krow, k_len},
                           {s.V, p.kv, p.kv_ld, p.v_off + h * d, s.krow, k_len}};
    load_rows_batch<3>(src, d);
  }
  __syncthreads();
  if (tid < AT) ln_rows<true>(s.Q, s.qrstd, tid, q_len, d, p.eps, s.gq, s.bq);
  else ln_rows<true>(s.K, s.krstd, tid - AT, k_len, d, p.eps, s.gk, s.bk);
  __syncthreads();

  const int r0 = wave * 32;
  if (r0 >= q_len) return;
  const int ktiles = (k_len + 31) / 32;
  const int qi = r0 + l31;  // this lane's query
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = zero16();
  if constexpr (PREC != 0) {
    // S^T = K Q^T: lane = (key row | query column) l31, k-half hh holds d-columns 16 * q2 + 8 * hh + 0..7 (image columns
    // >= d are zero).  The query fragments are hoisted over the key tiles.
    uint4 qh[2], ql[2];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = s.Q[qi * ALD + q2 * 16 + hh * 8 + e];
      split8<PREC>(v, qh[q2], ql[q2]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < ktiles) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          if (q2 * 16 >= d) continue;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = s.K[(t * 32 + l31) * ALD + q2 * 16 + hh * 8 + e];
          uint4 kh, kl;
          split8<PREC>(v, kh, kl);
          acc[t] = mfma_split<PREC>(kh, kl, qh[q2], ql[q2], acc[t]);
        }
      }
  } else {
    for (int kk = 0; kk < d; kk += 2) {
      const float bq = s.Q[qi * ALD + kk + hh];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < ktiles) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.K[(t * 32 + l31) * ALD + kk + hh], bq, acc[t], 0, 0, 0);
    }
  }
  // softmax over this query's keys: registers of both lane halves
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float v = key < k_len ? acc[t][r] * p.scale : -INFINITY;
      acc[t][r] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = acc[t][r] > -INFINITY ? __expf(acc[t][r] - m) : 0.f;
      acc[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  if (p.drop_thresh) {
    // idx = row_base + key with row_base a multiple of 128: only the low word changes with the key and the
    // high-word / seed terms are per-lane constants (keep_lo)
    const unsigned long long rb = (((unsigned long long)blockIdx.x * p.H + h) * AT + qi) * AT;
    const unsigned lo0 = (unsigned)rb, c2 = (unsigned)(rb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
    const unsigned s0 = (unsigned)p.drop_seed;
    const float keep = inv * p.drop_inv_keep;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[t][r] *= keep_lo(lo0 + (unsigned)(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh), s0, c2, p.drop_thresh) ? keep : 0.f;
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= inv;
  }
  if (hh == 0 && qi < q_len && p.lse) p.lse[(long)(q_start + qi) * p.H + h] = m + logf(sum);
  // O^T[dcol][query] = sum_key V[key][dcol] P[query][key]
  f32x16 o = zero16();
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < ktiles) {
      if constexpr (PREC != 0) {
        // registers 8g .. 8g+7 of the probability tile are this lane's k-slots of MFMA g: keys 16g + (j & 3) + 8 (j >> 2)
        // + 4 hh — the V operand gathers exactly those rows
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float pv[8], vv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            pv[j] = acc[t][8 * g + j];
            vv[j] = s.V[(t * 32 + 16 * g + (j & 3) + 8 * (j >> 2) + 4 * hh) * ALD + l31];
          }
          uint4 ph, pl, vh, vl;
          split8<PREC>(pv, ph, pl);
          split8<PREC>(vv, vh, vl);
          o = mfma_split<PREC>(vh, vl, ph, pl, o);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(s.V[key * ALD + l31], acc[t][r], o, 0, 0, 0);
        }
      }
    }
  if (qi < q_len && s.qown[qi]) {
    act_t* orow = p.out + (long)s.qrow[qi] * p.out_ld + h * d;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int dc = 8 * q4 + 4 * hh;
      if (dc < d) st4(orow + dc, make_float4(o[4 * q4], o[4 * q4 + 1], o[4 * q4 + 2], o[4 * q4 + 3]));
    }
  }
}

// ------------------------------------------------------------------------------------ backward
__device__ __forceinline__ void store_rows(const float* img, act_t* base, long ld, int coff, const int* rows_s,
                                           const int* own_s, int len, int d, int atomic, const int* ext_s = nullptr,
                                           act_t* ext_base = nullptr, long ext_ld = 0, int ext_coff = 0) {
  const int sub = threadIdx.x & 7;
  for (int r = threadIdx.x >> 3; r < len; r += 32) {
    if (own_s && !own_s[r]) continue;
    if (sub * 4 >= d) continue;
    const float* v = img + r * ALD + sub * 4;
    act_t* o = base + (long)rows_s[r] * ld + coff + sub * 4;
    if (ext_s && ext_s[r] >= 0) o = ext_base + (long)ext_s[r] * ext_ld + ext_coff + sub * 4;  // borrowed copy
#if !LOTUS_ACT_IS_BF16  // (generality only: the model never asks for atomic accumulation)
    if (atomic) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(o + e, v[e]);
      continue;
    }
#endif
    st4(o, make_float4(v[0], v[1], v[2], v[3]));
  }
}

// LN backward of rows [0,len) in place: g (in: d wrt normalised+affine output) -> d wrt raw input.
__device__ __forceinline__ void ln_rows_bwd(float* g, const float* xh, const float* rstd_s, int row, int len, int d,
                                            const float* gam) {
  if (row >= len) return;
  float* gr = g + row * ALD;
  const float* xr = xh + row * ALD;
  float s1 = 0.f, s2 = 0.f;
  for (int j = 0; j < d; ++j) {
    const float t = gr[j] * gam[j];
    s1 += t;
    s2 += t * xr[j];
  }
  s1 /= d;
  s2 /= d;
  const float rs = rstd_s[row];
  for (int j = 0; j < d; ++j) gr[j] = rs * (gr[j] * gam[j] - s1 - xr[j] * s2);
}

template <int PREC>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  AttnSmem s = carve(smem, true);
  __shared__ float lnacc[4][32];
  __shared__ float colred[2][8][32];
  const int tid = threadIdx.x, wave = tid >> 6, l31 = tid & 31, hh = (tid >> 5) & 1;
  const int h = blockIdx.y, d = p.d;
  const int* bd = p.blocks + blockIdx.x * 6;
  const int first_tile = bd[0], n_tiles = bd[1], tile_step = bd[2], part_slot = bd[3];
  const int k_start = bd[4], k_len = bd[5];
  const int r0 = wave * 32;
  const int ktiles = (k_len + 31) / 32;

  if (tid < AT) {
    s.krow[tid] = tid < k_len ? (p.kidx ? p.kidx[k_start + tid] : k_start + tid) : -1;
    s.kext[tid] = (tid < k_len && p.kext) ? p.kext[k_start + tid] : -1;
  }
  if (tid < 4 * 32) lnacc[tid >> 5][tid & 31] = 0.f;
  load_affine(p, s);
  __syncthreads();
  {
    const RowSrc src[2] = {{s.K, p.kv, p.kv_ld, p.k_off + h * d, s.krow, k_len}, {s.V, p.kv, p.kv_ld, p.v_off + h * d, s.krow, k_len}};
    load_rows_batch<2>(src, d);
  }
  __syncthreads();
  if (tid >= AT) ln_rows<false>(s.K, s.krstd, tid - AT, k_len, d, p.eps, nullptr, nullptr);
  __syncthreads();

  f32x16 acc_dv = zero16(), acc_dk = zero16();
  const float gk_l = s.gk[l31], bk_l = s.bk[l31], gq_l = s.gq[l31], bq_l = s.bq[l31];

  for (int ti = 0; ti < n_tiles; ++ti) {
    const int tile = first_tile + ti * tile_step;
    const int q_start = p.tiles[tile * 4 + 0], q_len = p.tiles[tile * 4 + 1];
    if (tid < AT) {
      s.qrow[tid] = tid < q_len ? (p.qidx ? p.qidx[q_start + tid] : q_start + tid) : -1;
      s.qown[tid] = tid < q_len ? (p.owner ? p.owner[q_start + tid] : 1) : 0;
      s.lse[tid] = tid < q_len ? p.lse[(long)(q_start + tid) * p.H + h] : 0.f;
    }
    __syncthreads();
    {  // Q rows, dO rows (zero for non-owners) and D = rowsum(dO * O): twelve independent 16-byte loads per thread in flight
      const int sub = tid & 7, rr0 = tid >> 3;
      float4 qv[4], gv[4], ov[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rr0 + 32 * j;
        qv[j] = gv[j] = ov[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < q_len && sub * 4 < d) {
          qv[j] = ld4(p.q + (long)s.qrow[r] * p.q_ld + p.q_off + h * d + sub * 4);
          if (s.qown[r]) {
            const long o = (long)s.qrow[r] * p.out_ld + h * d + sub * 4;
            gv[j] = ld4(p.dout + o);
            ov[j] = ld4(p.out + o);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rr0 + 32 * j;
        float* oq = s.Q + r * ALD + sub * 4;
        oq[0] = qv[j].x; oq[1] = qv[j].y; oq[2] = qv[j].z; oq[3] = qv[j].w;
        float* o = s.dO + r * ALD + sub * 4;
        o[0] = gv[j].x; o[1] = gv[j].y; o[2] = gv[j].z; o[3] = gv[j].w;
        float dsum = gv[j].x * ov[j].x + gv[j].y * ov[j].y + gv[j].z * ov[j].z + gv[j].w * ov[j].w;
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        dsum += __shfl_xor(dsum, 4, 64);
        if (sub == 0) s.Dv[r] = dsum;
      }
    }
    __syncthreads();
    if (tid < AT) ln_rows<false>(s.Q, s.qrstd, tid, q_len, d, p.eps, nullptr, nullptr);
    __syncthreads();

    // No score image: P and dS are recomputed in registers in the two accumulator orientations that the
    // three output products need (each lane owns ONE key, resp. ONE query; its registers run over the other
    // index, which is exactly the B-fragment layout of the products that reduce over that index).
    //
    // ---- orientation A: lane <-> key kj of this wave's 32 keys, registers <-> queries of tile qt
    //        S[q][kj], dP[q][kj]  ->  dV^T[:, kj] += dO^T Pm ,  dKn^T[:, kj] += Qn^T dS
    // (with a single key tile — cross-attention to <= 32 instruction tokens — all four waves share it and split
    //  the query tiles instead; their partial dV / dK are summed in wave order at the end of the block)
    const bool share_k = ktiles == 1;
    const int rk = share_k ? 0 : r0;
    if (rk < k_len) {
      const int kj = rk + l31;
      // q_norm's affine is folded into the hoisted key fragment: (xq g + b) . kn = xq . (g kn) + b . kn, so the
      // A operand is the raw normalised row and the bias term is one per-key constant
      float kb[16], vb[16], c_a = 0.f;
      uint4 kbh[2], kbl[2], vbh[2], vbl[2];  // bf16 paths: the same fragments as whole MFMA operands (k = 16 q2 + 8 hh + j)
      if constexpr (PREC != 0) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          float k8[8], v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = 16 * q2 + 8 * hh + j;
            const float kn = s.K[kj * ALD + k] * s.gk[k] + s.bk[k];
            c_a += s.bq[k] * kn;
            k8[j] = kn * s.gq[k];
            v8[j] = s.V[kj * ALD + k];
          }
          split8<PREC>(k8, kbh[q2], kbl[q2]);
          split8<PREC>(v8, vbh[q2], vbl[q2]);
        }
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const int k = 2 * s2 + hh;
          const float kn = s.K[kj * ALD + k] * s.gk[k] + s.bk[k];
          c_a += s.bq[k] * kn;
          kb[s2] = kn * s.gq[k];
          vb[s2] = s.V[kj * ALD + k];
        }
      }
      c_a += __shfl_xor(c_a, 32, 64);
      // dropout index ((tile * H + h) * 128 + q) * 128 + key: the tile/head part is a multiple of 2^14, so the
      // (q, key) part only ever touches the low word
      const unsigned long long tb = ((unsigned long long)tile * p.H + h) * AT * AT;
      const unsigned lo_a = (unsigned)tb + kj, s0 = (unsigned)p.drop_seed;
      const unsigned c2 = (unsigned)(tb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
      const int qtiles = (q_len + 31) / 32;
      for (int qt = share_k ? wave : 0; qt < qtiles; qt += share_k ? 4 : 1) {
        f32x16 sa = zero16(), dpa = zero16();
        const float* qrow_p = s.Q + (qt * 32 + l31) * ALD + hh;
        const float* dorow_p = s.dO + (qt * 32 + l31) * ALD + hh;
        if constexpr (PREC != 0) {
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            if (q2 * 16 >= d) continue;
            float q8[8], d8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              q8[j] = s.Q[(qt * 32 + l31) * ALD + 16 * q2 + 8 * hh + j];
              d8[j] = s.dO[(qt * 32 + l31) * ALD + 16 * q2 + 8 * hh + j];
            }
            uint4 ah, al, bh, bl;
            split8<PREC>(q8, ah, al);
            split8<PREC>(d8, bh, bl);
            sa = mfma_split<PREC>(ah, al, kbh[q2], kbl[q2], sa);
            dpa = mfma_split<PREC>(bh, bl, vbh[q2], vbl[q2], dpa);
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) {
            sa = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow_p[2 * s2], kb[s2], sa, 0, 0, 0);
            dpa = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow_p[2 * s2], vb[s2], dpa, 0, 0, 0);
          }
        }
        float lse4[16], d4[16];  // per-query scalars of this lane's 16 rows: four aligned runs of four queries
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 a = ld4(s.lse + qt * 32 + 8 * q4 + 4 * hh);
          const float4 b = ld4(s.Dv + qt * 32 + 8 * q4 + 4 * hh);
          lse4[4 * q4] = a.x; lse4[4 * q4 + 1] = a.y; lse4[4 * q4 + 2] = a.z; lse4[4 * q4 + 3] = a.w;
          d4[4 * q4] = b.x; d4[4 * q4 + 1] = b.y; d4[4 * q4 + 2] = b.z; d4[4 * q4 + 3] = b.w;
        }
        const bool kok = kj < k_len;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = qq < q_len && kok;
          const float pv = ok ? __expf((sa[r] + c_a) * p.scale - lse4[r]) : 0.f;
          const bool keep = !p.drop_thresh || keep_lo(lo_a + qq * AT, s0, c2, p.drop_thresh);
          sa[r] = keep ? pv * p.drop_inv_keep : 0.f;                                                   // dropout(P)
          dpa[r] = p.scale * pv * ((keep ? dpa[r] * p.drop_inv_keep : 0.f) - d4[r]);                   // dS
        }
        if constexpr (PREC != 0) {
          // registers 8g .. 8g+7 are this lane's k-slots of MFMA g: queries 16g + (j & 3) + 8 (j >> 2) + 4 hh
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            float p8[8], s8[8], o8[8], q8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int qq = qt * 32 + 16 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
              p8[j] = sa[8 * g + j];
              s8[j] = dpa[8 * g + j];
              o8[j] = s.dO[qq * ALD + l31];
              q8[j] = s.Q[qq * ALD + l31] * gq_l + bq_l;
            }
            uint4 ph, pl, sh, sl, oh, ol, qh, ql;
            split8<PREC>(p8, ph, pl);
            split8<PREC>(s8, sh, sl);
            split8<PREC>(o8, oh, ol);
            split8<PREC>(q8, qh, ql);
            acc_dv = mfma_split<PREC>(oh, ol, ph, pl, acc_dv);
            acc_dk = mfma_split<PREC>(qh, ql, sh, sl, acc_dk);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            acc_dv = __builtin_amdgcn_mfma_f32_32x32x2f32(s.dO[qq * ALD + l31], sa[r], acc_dv, 0, 0, 0);
            acc_dk = __builtin_amdgcn_mfma_f32_32x32x2f32(s.Q[qq * ALD + l31] * gq_l + bq_l, dpa[r], acc_dk, 0, 0, 0);
          }
        }
      }
    }
    // ---- orientation B: lane <-> query qi of this wave's 32 queries, registers <-> keys of tile t
    //        S^T[k][qi], dP^T[k][qi]  ->  dQn^T[:, qi] += Kn^T dS^T
    f32x16 acc_dq = zero16();
    if (r0 < q_len) {
      const int qi = r0 + l31;
      float qb[16], dob[16], c_b = 0.f;  // k_norm's affine folded into the hoisted query fragment (as above)
      uint4 qbh[2], qbl[2], dobh[2], dobl[2];
      if constexpr (PREC != 0) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          float q8[8], d8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = 16 * q2 + 8 * hh + j;
            const float qn = s.Q[qi * ALD + k] * s.gq[k] + s.bq[k];
            c_b += s.bk[k] * qn;
            q8[j] = qn * s.gk[k];
            d8[j] = s.dO[qi * ALD + k];
          }
          split8<PREC>(q8, qbh[q2], qbl[q2]);
          split8<PREC>(d8, dobh[q2], dobl[q2]);
        }
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const int k = 2 * s2 + hh;
          const float qn = s.Q[qi * ALD + k] * s.gq[k] + s.bq[k];
          c_b += s.bk[k] * qn;
          qb[s2] = qn * s.gk[k];
          dob[s2] = s.dO[qi * ALD + k];
        }
      }
      c_b += __shfl_xor(c_b, 32, 64);
      const float lse_q = s.lse[qi], d_q = s.Dv[qi];
      const unsigned long long tb = ((unsigned long long)tile * p.H + h) * AT * AT;
      const unsigned lo_b = (unsigned)tb + qi * AT, s0 = (unsigned)p.drop_seed;
      const unsigned c2 = (unsigned)(tb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
      for (int t = 0; t < ktiles; ++t) {
        f32x16 sb = zero16(), dpb = zero16();
        const float* krow_p = s.K + (t * 32 + l31) * ALD + hh;
        const float* vrow_p = s.V + (t * 32 + l31) * ALD + hh;
        if constexpr (PREC != 0) {
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            if (q2 * 16 >= d) continue;
            float k8[8], v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              k8[j] = s.K[(t * 32 + l31) * ALD + 16 * q2 + 8 * hh + j];
              v8[j] = s.V[(t * 32 + l31) * ALD + 16 * q2 + 8 * hh + j];
            }
            uint4 ah, al, bh, bl;
            split8<PREC>(k8, ah, al);
            split8<PREC>(v8, bh, bl);
            sb = mfma_split<PREC>(ah, al, qbh[q2], qbl[q2], sb);
            dpb = mfma_split<PREC>(bh, bl, dobh[q2], dobl[q2], dpb);
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) {
            sb = __builtin_amdgcn_mfma_f32_32x32x2f32(krow_p[2 * s2], qb[s2], sb, 0, 0, 0);
            dpb = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow_p[2 * s2], dob[s2], dpb, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = key < k_len && qi < q_len;
          const float pv = ok ? __expf((sb[r] + c_b) * p.scale - lse_q) : 0.f;
          const bool keep = !p.drop_thresh || keep_lo(lo_b + key, s0, c2, p.drop_thresh);
          sb[r] = p.scale * pv * ((keep ? dpb[r] * p.drop_inv_keep : 0.f) - d_q);                      // dS^T
        }
        if constexpr (PREC != 0) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            float s8[8], k8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int key = t * 32 + 16 * g + (j & 3) + 8 * (j >> 2) + 4 * hh;
              s8[j] = sb[8 * g + j];
              k8[j] = s.K[key * ALD + l31] * gk_l + bk_l;
            }
            uint4 sh, sl, kh, kl;
            split8<PREC>(s8, sh, sl);
            split8<PREC>(k8, kh, kl);
            acc_dq = mfma_split<PREC>(kh, kl, sh, sl, acc_dq);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            acc_dq = __builtin_amdgcn_mfma_f32_32x32x2f32(s.K[key * ALD + l31] * gk_l + bk_l, sb[r], acc_dq, 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // dO image is free now: reuse it for dQn
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // accumulator is dQn^T: col = query (lane), row = head column
      const int dc = (r & 3) + 8 * (r >> 2) + 4 * hh;
      s.dO[(r0 + l31) * ALD + dc] = (r0 < q_len && dc < d) ? acc_dq[r] : 0.f;
    }
    __syncthreads();
    // q_norm affine gradients (column pass), then LN backward (row pass), then store
    {  // column sums over the tile's rows with all 256 threads: 8 row slices x 32 columns, then an 8-way tree
      const int j = tid & 31, part = tid >> 5;
      float ag = 0.f, ab = 0.f;
      if (j < d)
        for (int r = part * 16; r < min(q_len, part * 16 + 16); ++r) {
          const float g = s.dO[r * ALD + j];
          ag += g * s.Q[r * ALD + j];
          ab += g;
        }
      colred[0][part][j] = ag;
      colred[1][part][j] = ab;
    }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, j = tid & 31;  // 0: dgamma_q, 1: dbeta_q
      float a = 0.f;
      for (int k = 0; k < 8; ++k) a += colred[which][k][j];
      lnacc[which][j] += a;
    }
    if (tid < AT) ln_rows_bwd(s.dO, s.Q, s.qrstd, tid, q_len, d, s.gq);
    __syncthreads();
    store_rows(s.dO, p.dq, p.dq_ld, p.dq_off + h * d, s.qrow, s.qown, q_len, d, 0);  // one owner per row
    __syncthreads();
  }

  // ---- K / V gradients of this block
  // dV straight from registers -> V image (V no longer needed), dKn -> dO image
  if (ktiles == 1) {  // shared key tile: rows 0..31 get the four waves' partials in wave order, the rest is zero
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dc = (r & 3) + 8 * (r >> 2) + 4 * hh;
          float* pv = s.V + l31 * ALD + dc;
          float* pk = s.dO + l31 * ALD + dc;
          if (w == 0) {
            *pv = (dc < d) ? acc_dv[r] : 0.f;
            *pk = (dc < d) ? acc_dk[r] : 0.f;
          } else if (dc < d) {
            *pv += acc_dv[r];
            *pk += acc_dk[r];
          }
        }
      } else if (w == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dc = (r & 3) + 8 * (r >> 2) + 4 * hh;
          s.V[(r0 + l31) * ALD + dc] = 0.f;
          s.dO[(r0 + l31) * ALD + dc] = 0.f;
        }
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // accumulators are dV^T / dKn^T: col = key (lane), row = head column
      const int dc = (r & 3) + 8 * (r >> 2) + 4 * hh;
      s.V[(r0 + l31) * ALD + dc] = (dc < d) ? acc_dv[r] : 0.f;
      s.dO[(r0 + l31) * ALD + dc] = (dc < d) ? acc_dk[r] : 0.f;
    }
    __syncthreads();
  }
  {
    const int j = tid & 31, part = tid >> 5;
    float ag = 0.f, ab = 0.f;
    if (j < d)
      for (int r = part * 16; r < min(k_len, part * 16 + 16); ++r) {
        const float g = s.dO[r * ALD + j];
        ag += g * s.K[r * ALD + j];
        ab += g;
      }
    colred[0][part][j] = ag;
    colred[1][part][j] = ab;
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, j = tid & 31;  // 2: dgamma_k, 3: dbeta_k
    float a = 0.f;
    for (int k = 0; k < 8; ++k) a += colred[which][k][j];
    lnacc[2 + which][j] = a;
  }
  if (tid < AT) ln_rows_bwd(s.dO, s.K, s.krstd, tid, k_len, d, s.gk);
  __syncthreads();
  act_t* dkv = p.dkv + (long)part_slot * p.dkv_part_stride;
  store_rows(s.dO, dkv, p.dkv_ld, p.dk_off + h * d, s.krow, nullptr, k_len, d, p.atomic_out, p.dkv_extra ? s.kext : nullptr,
             p.dkv_extra, p.dkv_extra_ld, h * d);
  store_rows(s.V, dkv, p.dkv_ld, p.dv_off + h * d, s.krow, nullptr, k_len, d, p.atomic_out, p.dkv_extra ? s.kext : nullptr,
             p.dkv_extra, p.dkv_extra_ld, p.dv_off - p.dk_off + h * d);
  if (tid < 128) p.ln_part[((long)(blockIdx.x * p.H + h) * 4 + (tid >> 5)) * 32 + (tid & 31)] = lnacc[tid >> 5][tid & 31];
}

// ---- exact-fp32 backward, second formulation (round 5).  On gfx950 an fp32 MFMA and a vector-ALU instruction never overlap
// on a SIMD (tools/ubench/mfma_coissue.hip), so a block's time is MFMA cycles PLUS ~5 cycles per other vector instruction;
// attn_bwd_kernel<0> spent 11 vector instructions per MFMA and recomputed S / dP in a second orientation for dQ.  Here:
//   * S, P and dS exist ONCE, in the orientation lane <-> key (dV, dK reduce over the queries a lane's registers run over);
//   * dQ reduces over the keys, which sit on lanes: every wave drops its dS tile (32 keys x 32 queries) into an LDS exchange
//     image X[key][query] (aliased onto the V image: V lives in hoisted registers by then; 16-byte chunks XOR-swizzled so that
//     the ds_write_b128 of the producers and the ds_read_b32 of the consumers are conflict-free), and after one barrier all
//     four waves compute one 16 x 16 quadrant each of dQn^T[d][32 queries] over ALL keys with v_mfma_f32_16x16x4_f32
//     (same FLOP rate as the 32x32 form; no cross-wave sum, fixed order -> deterministic);
//   * the softmax scale, log2(e) and the q_norm bias term are folded into the hoisted key fragment and into the accumulator's
//     INITIAL value (P = exp2(acc - lse2[q]): two instructions), row / key bounds ride on +-inf instead of selects, and the
//     dropout hash advances by compile-time constants (one multiply per element instead of two).
// Per query tile and wave: 96 + 16 MFMA-equivalents and ~300 vector instructions (was 144 and ~750).
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xswz(int key) { return ((key & 1) << 2) | ((key >> 1) & 3); }

__global__ __launch_bounds__(256, 2) void attn_bwd2_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  AttnSmem s = carve(smem, true);
  __shared__ float lnacc[4][32];
  __shared__ float colred[2][8][32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = tid & 31, hh = (tid >> 5) & 1;
  const int h = blockIdx.y, d = p.d;
  const int* bd = p.blocks + blockIdx.x * 6;
  const int first_tile = bd[0], n_tiles = bd[1], tile_step = bd[2], part_slot = bd[3];
  const int k_start = bd[4], k_len = bd[5];
  const int r0 = wave * 32;
  const int ktiles = (k_len + 31) / 32;
  float* X = s.V;  // dS exchange image [128 keys][32 queries], valid between the hoist of V and the dV staging

  if (tid < AT) {
    s.krow[tid] = tid < k_len ? (p.kidx ? p.kidx[k_start + tid] : k_start + tid) : -1;
    s.kext[tid] = (tid < k_len && p.kext) ? p.kext[k_start + tid] : -1;
  }
  if (tid < 4 * 32) lnacc[tid >> 5][tid & 31] = 0.f;
  load_affine(p, s);
  __syncthreads();
  {
    const RowSrc src[2] = {{s.K, p.kv, p.kv_ld, p.k_off + h * d, s.krow, k_len}, {s.V, p.kv, p.kv_ld, p.v_off + h * d, s.krow, k_len}};
    load_rows_batch<2>(src, d);
  }
  __syncthreads();
  if (tid >= AT) ln_rows<false>(s.K, s.krstd, tid - AT, k_len, d, p.eps, nullptr, nullptr);
  __syncthreads();

  const float gq_l = s.gq[l31], bq_l = s.bq[l31];
  const float sl2 = p.scale * 1.4426950408889634f;
  // this wave's keys: hoisted fragments (k = 2 s2 + hh).  q_norm's affine and scale * log2(e) are folded into the key
  // fragment, (xq g + b) . kn = xq . (g kn) + b . kn; the bias term c_a starts the accumulator (-inf for absent keys)
  const bool has_keys = r0 < k_len;
  const int kj = r0 + l31;
  float kb[16], vb[16], c_a = 0.f;
  if (has_keys) {
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int k = 2 * s2 + hh;
      const float kn = s.K[kj * ALD + k] * s.gk[k] + s.bk[k];
      c_a += s.bq[k] * kn;
      kb[s2] = kn * s.gq[k] * sl2;
      vb[s2] = s.V[kj * ALD + k];
    }
    c_a += __shfl_xor(c_a, 32, 64);
    c_a = kj < k_len ? c_a * sl2 : -INFINITY;
  }
  f32x16 acc_dv = zero16(), acc_dk = zero16();
  // the dQ quadrant of this wave: head columns 16 dh + (lane & 15), queries 16 qh + (lane & 15) of the running query tile
  const int dh = wave & 1, qh = wave >> 1, l15 = lane & 15, kq = lane >> 4;
  const float gk_c = s.gk[16 * dh + l15], bk_c = s.bk[16 * dh + l15];
  __syncthreads();  // every wave holds its V fragment: the V image may become X

  for (int ti = 0; ti < n_tiles; ++ti) {
    const int tile = first_tile + ti * tile_step;
    const int q_start = p.tiles[tile * 4 + 0], q_len = p.tiles[tile * 4 + 1];
    if (tid < AT) {
      s.qrow[tid] = tid < q_len ? (p.qidx ? p.qidx[q_start + tid] : q_start + tid) : -1;
      s.qown[tid] = tid < q_len ? (p.owner ? p.owner[q_start + tid] : 1) : 0;
      s.lse[tid] = tid < q_len ? p.lse[(long)(q_start + tid) * p.H + h] * 1.4426950408889634f : INFINITY;  // log2 units
    }
    __syncthreads();
    {  // Q rows, dO rows (zero for non-owners) and D = rowsum(dO * O): twelve independent 16-byte loads per thread in flight
      const int sub = tid & 7, rr0 = tid >> 3;
      float4 qv[4], gv[4], ov[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rr0 + 32 * j;
        qv[j] = gv[j] = ov[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < q_len && sub * 4 < d) {
          qv[j] = ld4(p.q + (long)s.qrow[r] * p.q_ld + p.q_off + h * d + sub * 4);
          if (s.qown[r]) {
            const long o = (long)s.qrow[r] * p.out_ld + h * d + sub * 4;
            gv[j] = ld4(p.dout + o);
            ov[j] = ld4(p.out + o);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rr0 + 32 * j;
        float* oq = s.Q + r * ALD + sub * 4;
        oq[0] = qv[j].x; oq[1] = qv[j].y; oq[2] = qv[j].z; oq[3] = qv[j].w;
        float* o = s.dO + r * ALD + sub * 4;
        o[0] = gv[j].x; o[1] = gv[j].y; o[2] = gv[j].z; o[3] = gv[j].w;
        float dsum = gv[j].x * ov[j].x + gv[j].y * ov[j].y + gv[j].z * ov[j].z + gv[j].w * ov[j].w;
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        dsum += __shfl_xor(dsum, 4, 64);
        if (sub == 0) s.Dv[r] = dsum;
      }
    }
    __syncthreads();
    if (tid < AT) ln_rows<false>(s.Q, s.qrstd, tid, q_len, d, p.eps, nullptr, nullptr);
    __syncthreads();

    // dropout index ((tile * H + h) * 128 + q) * 128 + key: the (q, key) part only ever touches the low word (keep_lo);
    // x0 = lo * 0x9E3779B1 + s0 advances by a compile-time constant per register
    const unsigned long long tb = ((unsigned long long)tile * p.H + h) * AT * AT;
    const unsigned c2 = (unsigned)(tb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
    const unsigned x_lane = ((unsigned)tb + (unsigned)kj + (unsigned)(4 * hh * AT)) * 0x9E3779B1u + (unsigned)p.drop_seed;
    const int qtiles = (q_len + 31) / 32;
    for (int qt = 0; qt < qtiles; ++qt) {
      f32x16 sa, dpa;
      if (has_keys) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = c_a; dpa[r] = 0.f; }
        const float* qrow_p = s.Q + (qt * 32 + l31) * ALD + hh;
        const float* dorow_p = s.dO + (qt * 32 + l31) * ALD + hh;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          sa = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow_p[2 * s2], kb[s2], sa, 0, 0, 0);
          dpa = __builtin_amdgcn_mfma_f32_32x32x2f32(dorow_p[2 * s2], vb[s2], dpa, 0, 0, 0);
        }
        float lse4[16], d4[16];  // per-query scalars of this lane's 16 rows: four aligned runs of four queries
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 a = ld4(s.lse + qt * 32 + 8 * q4 + 4 * hh);
          const float4 b = ld4(s.Dv + qt * 32 + 8 * q4 + 4 * hh);
          lse4[4 * q4] = a.x; lse4[4 * q4 + 1] = a.y; lse4[4 * q4 + 2] = a.z; lse4[4 * q4 + 3] = a.w;
          d4[4 * q4] = b.x; d4[4 * q4 + 1] = b.y; d4[4 * q4 + 2] = b.z; d4[4 * q4 + 3] = b.w;
        }
        if (p.drop_thresh) {
          const unsigned xq = x_lane + (unsigned)(qt * 32 * AT) * 0x9E3779B1u;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sa[r] - lse4[r]);
            unsigned x = xq + (unsigned)(((r & 3) + 8 * (r >> 2)) * AT) * 0x9E3779B1u;
            x ^= c2;
            x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
            const float t = x >= p.drop_thresh ? p.drop_inv_keep : 0.f;
            sa[r] = e * t;                                       // dropout(P)
            dpa[r] = (e * p.scale) * (dpa[r] * t - d4[r]);       // dS
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sa[r] - lse4[r]);
            sa[r] = e;
            dpa[r] = (e * p.scale) * (dpa[r] - d4[r]);
          }
        }
      }
      __syncthreads();  // (A) the dQ products of the previous query tile are done with X
      if (has_keys) {
        float* xr = X + kj * 32;
        const int sw = xswz(kj);
#pragma unroll
        for (int g = 0; g < 4; ++g)  // registers 4g .. 4g+3 = queries 8g + 4hh + 0..3: one 16-byte chunk
          *reinterpret_cast<float4*>(xr + (((2 * g + hh) ^ sw) << 2)) = make_float4(dpa[4 * g], dpa[4 * g + 1], dpa[4 * g + 2], dpa[4 * g + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          acc_dv = __builtin_amdgcn_mfma_f32_32x32x2f32(s.dO[qq * ALD + l31], sa[r], acc_dv, 0, 0, 0);
          acc_dk = __builtin_amdgcn_mfma_f32_32x32x2f32(s.Q[qq * ALD + l31] * gq_l + bq_l, dpa[r], acc_dk, 0, 0, 0);
        }
      }
      __syncthreads();  // (B) X holds dS of every key for this query tile; nobody reads rows qt*32.. of the dO image any more
      {
        // dQn^T[dcol][q] = sum_key Kn[key][dcol] dS[q][key]: A[i = dcol][k = key], B[k = key][j = q], key = 4 s + (lane >> 4)
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        const int ql = 16 * qh + l15;
        const float* kcol = s.K + 16 * dh + l15;
        for (int t = 0; t < ktiles; ++t) {
#pragma unroll
          for (int s4 = 0; s4 < 8; ++s4) {
            const int key = 32 * t + 4 * s4 + kq;
            const float a = kcol[key * ALD] * gk_c + bk_c;
            const float b = X[key * 32 + ((((ql >> 2) ^ xswz(key)) << 2) | (ql & 3))];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
          }
        }
        float* o = s.dO + (qt * 32 + ql) * ALD + 16 * dh + 4 * kq;  // accumulator rows 4 (lane >> 4) + r, column lane & 15
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
      }
    }
    __syncthreads();
    // rows of query tiles that were not run (q_len <= 96) still hold dO: they are beyond q_len and never read below
    // q_norm affine gradients (column pass), then LN backward (row pass), then store
    {  // column sums over the tile's rows with all 256 threads: 8 row slices x 32 columns, then an 8-way tree
      const int j = tid & 31, part = tid >> 5;
      float ag = 0.f, ab = 0.f;
      if (j < d)
        for (int r = part * 16; r < min(q_len, part * 16 + 16); ++r) {
          const float g = s.dO[r * ALD + j];
          ag += g * s.Q[r * ALD + j];
          ab += g;
        }
      colred[0][part][j] = ag;
      colred[1][part][j] = ab;
    }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, j = tid & 31;  // 0: dgamma_q, 1: dbeta_q
      float a = 0.f;
      for (int k = 0; k < 8; ++k) a += colred[which][k][j];
      lnacc[which][j] += a;
    }
    if (tid < AT) ln_rows_bwd(s.dO, s.Q, s.qrstd, tid, q_len, d, s.gq);
    __syncthreads();
    store_rows(s.dO, p.dq, p.dq_ld, p.dq_off + h * d, s.qrow, s.qown, q_len, d, 0);  // one owner per row
    __syncthreads();
  }

  // ---- K / V gradients of this block: dV^T / dKn^T accumulators (col = key (lane), row = head column) -> V / dO images
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dc = (r & 3) + 8 * (r >> 2) + 4 * hh;
    s.V[(r0 + l31) * ALD + dc] = (dc < d) ? acc_dv[r] : 0.f;
    s.dO[(r0 + l31) * ALD + dc] = (dc < d) ? acc_dk[r] : 0.f;
  }
  __syncthreads();
  {
    const int j = tid & 31, part = tid >> 5;
    float ag = 0.f, ab = 0.f;
    if (j < d)
      for (int r = part * 16; r < min(k_len, part * 16 + 16); ++r) {
        const float g = s.dO[r * ALD + j];
        ag += g * s.K[r * ALD + j];
        ab += g;
      }
    colred[0][part][j] = ag;
    colred[1][part][j] = ab;
  }
  __syncthreads();
  if (tid < 64) {
    const int which = tid >> 5, j = tid & 31;  // 2: dgamma_k, 3: dbeta_k
    float a = 0.f;
    for (int k = 0; k < 8; ++k) a += colred[which][k][j];
    lnacc[2 + which][j] = a;
  }
  if (tid < AT) ln_rows_bwd(s.dO, s.K, s.krstd, tid, k_len, d, s.gk);
  __syncthreads();
  act_t* dkv = p.dkv + (long)part_slot * p.dkv_part_stride;
  store_rows(s.dO, dkv, p.dkv_ld, p.dk_off + h * d, s.krow, nullptr, k_len, d, p.atomic_out, p.dkv_extra ? s.kext : nullptr,
             p.dkv_extra, p.dkv_extra_ld, h * d);
  store_rows(s.V, dkv, p.dkv_ld, p.dv_off + h * d, s.krow, nullptr, k_len, d, p.atomic_out, p.dkv_extra ? s.kext : nullptr,
             p.dkv_extra, p.dkv_extra_ld, p.dv_off - p.dk_off + h * d);
  if (tid < 128) p.ln_part[((long)(blockIdx.x * p.H + h) * 4 + (tid >> 5)) * 32 + (tid & 31)] = lnacc[tid >> 5][tid & 31];
}

// out[which][j] (+)= sum_b part[b][which][j]   (which: dgamma_q, dbeta_q, dgamma_k, dbeta_k)
// one block per `which`; 32 row lanes x 32 columns, four loads in flight per lane, fixed-order tree -> deterministic
__global__ __launch_bounds__(1024) void attn_ln_reduce_kernel(const float* __restrict__ part, float* o0, float* o1,
                                                              float* o2, float* o3, int nb, int d, int accumulate) {
  __shared__ float red[32][33];
  const int which = blockIdx.x, j = threadIdx.x & 31, r = threadIdx.x >> 5;
  float s = 0.f;
  int b = r;
  for (; b + 96 < nb; b += 128) {
    const float v0 = part[((long)b * 4 + which) * 32 + j], v1 = part[((long)(b + 32) * 4 + which) * 32 + j];
    const float v2 = part[((long)(b + 64) * 4 + which) * 32 + j], v3 = part[((long)(b + 96) * 4 + which) * 32 + j];
    s += (v0 + v1) + (v2 + v3);
  }
  for (; b < nb; b += 32) s += part[((long)b * 4 + which) * 32 + j];
  red[r][j] = s;
  __syncthreads();
  if (r == 0 && j < d) {
    float t = 0.f;
    for (int k = 0; k < 32; ++k) t += red[k][j];
    float* o = which == 0 ? o0 : which == 1 ? o1 : which == 2 ? o2 : o3;
    o[j] = accumulate ? o[j] + t : t;
  }
}

static void set_attn_drop(AttnP& p, float drop_p, unsigned long long seed) {
  p.drop_seed = seed;
  p.drop_thresh = 0;
  p.drop_inv_keep = 1.f;
  if (drop_p > 0.f) {
    p.drop_thresh = (unsigned)(drop_p * 4294967296.0);
    if (p.drop_thresh == 0) p.drop_thresh = 1;
    p.drop_inv_keep = 1.f / (1.f - drop_p);
  }
}

// dqkv[point][k|v columns] += extra[e]  for every borrowed copy e (each point is borrowed at most once)
__global__ void attn_extra_fixup_kernel(const act_t* __restrict__ extra, long extra_ld, const int* __restrict__ ext_pos,
                                        const int* __restrict__ kidx, int n_extra, int w4, act_t* __restrict__ dkv,
                                        long dkv_ld, int dk_off) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n_extra * w4) return;
  const int e = (int)(gid / w4), c = (int)(gid % w4) * 4;
  const int point = kidx[ext_pos[e]];
  const float4 a = ld4(extra + (long)e * extra_ld + c);
  act_t* o = dkv + (long)point * dkv_ld + dk_off + c;
  float4 v = ld4(o);
  v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
  st4(o, v);
}

// ---------------------------------------------------------------------------------------------------------------
// Query-per-lane attention.  fp32 MFMA runs at the fp32 VECTOR rate on gfx950 (64 FLOP / clk / SIMD either way), so for the
// exact-fp32 mode the matrix cores buy nothing in QK^T and PV — but the 128 x 128 tile kernels above pay for them with row
// images, per-phase barriers, transposed-accumulator bookkeeping and, in backward, two recomputed products (MFMA
// utilisation 0.15 forward / 0.21 backward).  Here ONE LANE OWNS ONE QUERY: its q row, q_norm, scores, softmax and output
// stay in registers; the normalised keys and the values are LDS rows that every lane reads at the same address
// (broadcast); the forward pass has no barrier after the key set-up.  The backward pass keeps the per-query part in
// registers the same way (dS, dq, q_norm backward) and does the only reductions over queries — dV = (P mask)^T dO and
// T = dS^T qhat, from which d k and the q_norm parameter gradients follow (see xq_bwd_kernel) — with two MFMA chains per
// wave over wave-private LDS images, 32 keys at a time.
// Serves both call sites: the point <-> instruction cross attention (identity rows, 6-19 keys per cloud: one key chunk,
// several query tiles per block) and the patch self attention (gathered rows, 128 keys = four chunks, one tile per block;
// owner flags and the borrowed tail-patch copies as in the tile kernels).  Exact fp32 whatever `precision` says.
// Dropout masks use the same (tile, head, query, key) hash index as the tile kernels.

// rows j < n of K and V (D floats each, row ids from rows_s) -> [.][32] images, zero beyond n / D; 8 lanes per row
template <int D>
__device__ __forceinline__ void xq_load_keys(const AttnP& p, int h, const int* __restrict__ rows_s, int n, int npad,
                                             float* __restrict__ kraw, float* __restrict__ v_s) {
  for (int i = threadIdx.x; i < npad * 8; i += blockDim.x) {
    const int j = i >> 3, c4 = i & 7;
    float4 kv4 = make_float4(0.f, 0.f, 0.f, 0.f), vv4 = kv4;
    if (j < n && c4 * 4 < D) {
      const act_t* row = p.kv + (long)rows_s[j] * p.kv_ld + h * D + c4 * 4;
      kv4 = ld4(row + p.k_off);
      vv4 = ld4(row + p.v_off);
    }
    *reinterpret_cast<float4*>(kraw + j * 32 + c4 * 4) = kv4;
    *reinterpret_cast<float4*>(v_s + j * 32 + c4 * 4) = vv4;
  }
}

// LayerNorm(d, eps) of npad key rows in place, 8 lanes per row: khat -> kraw, rstd -> krstd (0 for rows >= n)
template <int D>
__device__ __forceinline__ void xq_norm_keys(float eps, int n, int npad, float* __restrict__ kraw, float* __restrict__ krstd) {
  for (int i = threadIdx.x; i < npad * 8; i += blockDim.x) {
    const int j = i >> 3, c4 = i & 7;
    const float4 v = *reinterpret_cast<const float4*>(kraw + j * 32 + c4 * 4);
    float sum = (c4 * 4 < D) ? (v.x + v.y) + (v.z + v.w) : 0.f;
    sum += __shfl_xor(sum, 1, 8); sum += __shfl_xor(sum, 2, 8); sum += __shfl_xor(sum, 4, 8);
    const float m = sum / D;
    const float cx = v.x - m, cy = v.y - m, cz = v.z - m, cw = v.w - m;
    float var = (c4 * 4 < D) ? (cx * cx + cy * cy) + (cz * cz + cw * cw) : 0.f;
    var += __shfl_xor(var, 1, 8); var += __shfl_xor(var, 2, 8); var += __shfl_xor(var, 4, 8);
    const float rs = (j < n) ? rsqrtf(var / D + eps) : 0.f;
    if (c4 * 4 < D) *reinterpret_cast<float4*>(kraw + j * 32 + c4 * 4) = make_float4(cx * rs, cy * rs, cz * rs, cw * rs);
    if (c4 == 0) krstd[j] = rs;
  }
}

template <int D>
__global__ __launch_bounds__(128) void xq_fwd_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) float kn_s[AT * 32], v_s[AT * 32], krstd_s[AT];
  __shared__ int krow_s[AT];
  const int tid = threadIdx.x, h = blockIdx.y;
  const int q_start = p.tiles[blockIdx.x * 4 + 0], q_len = p.tiles[blockIdx.x * 4 + 1];
  const int k_start = p.tiles[blockIdx.x * 4 + 2], k_len = p.tiles[blockIdx.x * 4 + 3];
  const int kpad = (k_len + 3) & ~3;  // keys are consumed four at a time
  if (tid < AT) krow_s[tid] = tid < k_len ? (p.kidx ? p.kidx[k_start + tid] : k_start + tid) : -1;
  __syncthreads();
  xq_load_keys<D>(p, h, krow_s, k_len, kpad, kn_s, v_s);
  __syncthreads();
  xq_norm_keys<D>(p.eps, k_len, kpad, kn_s, krstd_s);
  __syncthreads();
  for (int i = tid; i < kpad * 32; i += 128) {  // affine part of k_norm
    const int c = i & 31;
    if (c < D) kn_s[i] = kn_s[i] * p.kn_w[c] + p.kn_b[c];
  }
  __syncthreads();
  const int qi = tid;
  if (qi >= q_len) return;
  const int pos = q_start + qi;
  if (p.owner && !p.owner[pos]) return;  // borrowed copy of a tail patch: its output row is never used
  const long row = p.qidx ? p.qidx[pos] : pos;
  float q[D];
  {
    const act_t* qp = p.q + row * p.q_ld + p.q_off + h * D;
#pragma unroll
    for (int c4 = 0; c4 < D / 4; ++c4) {
      const float4 v = ld4(qp + c4 * 4);
      q[4 * c4] = v.x; q[4 * c4 + 1] = v.y; q[4 * c4 + 2] = v.z; q[4 * c4 + 3] = v.w;
    }
  }
  {  // q_norm (LayerNorm(d, eps) with affine), model.py:532 / model_ca.py:52; the softmax scale is folded in
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) m += q[c];
    m /= D;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const float t = q[c] - m;
      var += t * t;
    }
    const float rs = rsqrtf(var / D + p.eps);
#pragma unroll
    for (int c = 0; c < D; ++c) q[c] = (q[c] - m) * rs * p.qn_w[c] + p.qn_b[c];
  }
  const unsigned long long rb = (((unsigned long long)blockIdx.x * p.H + h) * AT + qi) * AT;
  const unsigned lo0 = (unsigned)rb, c2 = (unsigned)(rb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
  const unsigned s0 = (unsigned)p.drop_seed;
  float mx = -INFINITY, l = 0.f, o[D];
#pragma unroll
  for (int c = 0; c < D; ++c) o[c] = 0.f;
  for (int j0 = 0; j0 < k_len; j0 += 4) {  // streaming softmax, four keys per rescale
    float sc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float4* kr = reinterpret_cast<const float4*>(kn_s + (j0 + e) * 32);
      float a = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < D / 4; ++c4) {
        const float4 k4 = kr[c4];
        a = fmaf(q[4 * c4], k4.x, a); a = fmaf(q[4 * c4 + 1], k4.y, a);
        a = fmaf(q[4 * c4 + 2], k4.z, a); a = fmaf(q[4 * c4 + 3], k4.w, a);
      }
      sc[e] = (j0 + e < k_len) ? a * p.scale : -INFINITY;
    }
    const float mn = fmaxf(fmaxf(mx, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
    const float corr = __expf(mx - mn);
    mx = mn;
    l *= corr;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] *= corr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pj = __expf(sc[e] - mn);
      l += pj;
      float w = pj;
      if (p.drop_thresh) w = keep_lo(lo0 + (unsigned)(j0 + e), s0, c2, p.drop_thresh) ? pj * p.drop_inv_keep : 0.f;
      const float4* vr = reinterpret_cast<const float4*>(v_s + (j0 + e) * 32);
#pragma unroll
      for (int c4 = 0; c4 < D / 4; ++c4) {
        const float4 v4 = vr[c4];
        o[4 * c4] = fmaf(w, v4.x, o[4 * c4]); o[4 * c4 + 1] = fmaf(w, v4.y, o[4 * c4 + 1]);
        o[4 * c4 + 2] = fmaf(w, v4.z, o[4 * c4 + 2]); o[4 * c4 + 3] = fmaf(w, v4.w, o[4 * c4 + 3]);
      }
    }
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (p.lse) p.lse[(long)pos * p.H + h] = mx + logf(l);
  act_t* op = p.out + row * p.out_ld + h * D;
#pragma unroll
  for (int c4 = 0; c4 < D / 4; ++c4)
    st4(op + c4 * 4, make_float4(o[4 * c4] * inv, o[4 * c4 + 1] * inv, o[4 * c4 + 2] * inv, o[4 * c4 + 3] * inv));
}

// Backward.  With kn = khat gk + bk (k_norm output), qn = qhat gq + bq (q_norm output) and dS the score gradient:
//   s_ij = scale (qgk_i . khat_j + qb_i + kb_j),  qgk = qhat gq gk,  qb_i = (qhat_i gq) . bk,  kb_j = bq . kn_j
//   d qhat_i = gq (gk sum_j dS_ij khat_j + bk sum_j dS_ij)                               (registers, per query)
//   T = dS^T qhat  [keys x d],  cs_j = sum_i dS_ij                                       (MFMA / column sums over queries)
//   d kn_j = gq * T_j + bq cs_j,   d gq = sum_j kn_j * T_j,   d bq = sum_j kn_j cs_j
//   d V = (P * mask)^T dO                                                                (MFMA)
// Keys are taken 32 at a time (chunk kc); a block either has one chunk and several query tiles (cross attention: T / dV
// accumulate over the tiles) or one tile and several chunks (patch attention: the query state stays in registers across the
// chunks, every chunk is finished and stored on its own) — the host routes nothing else here.  Each wave reduces its own
// 64 rows of the LDS images, so there is no block barrier between the per-query phase and the products beyond the one
// that publishes the images.
template <int D>
__global__ __launch_bounds__(128) void xq_bwd_kernel(AttnP p) {
  constexpr int ILD = 33;
  __shared__ __attribute__((aligned(16))) float khat_s[32 * 32], v_s[32 * 32], krstd_s[32], kb_s[32], cs_s[2][32];
  __shared__ int krow_s[32], kext_s[32];
  __shared__ float ds_s[128 * ILD], pm_s[128 * ILD], qh_s[128 * ILD], do_s[128 * ILD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = tid & 31, hh = (tid >> 5) & 1, h = blockIdx.y;
  const int* bd = p.blocks + blockIdx.x * 6;
  const int first_tile = bd[0], n_tiles = bd[1], tile_step = bd[2], part_slot = bd[3], k_start = bd[4], k_len = bd[5];
  const int n_chunks = (k_len + 31) >> 5;
  // The per-query state (dx, dsum, ...) lives in registers across the key chunks of ONE tile; a block that walks several
  // tiles must therefore see a single chunk (<= 32 keys).  That is what the caller promises with k_max <= 32 (the
  // cross-attention's instruction tokens); a tile list that breaks the promise would give wrong d q / d k / d v silently
  //  — stop the kernel loudly instead.
  if (n_chunks > 1 && n_tiles > 1) __builtin_trap();
  for (int i = tid; i < 128 * ILD; i += 128) { qh_s[i] = 0.f; do_s[i] = 0.f; }  // columns >= D stay zero for the MFMA operands
  const unsigned s0 = (unsigned)p.drop_seed;
  const int qi = tid;
  // per-query state (lives across the key chunks when the block has a single tile)
  float qh[D], go[D], dx[D], qgk[D];  // qgk = qhat gq gk: s_ij = qgk_i . khat_j + qb_i + kb_j (both affines folded in)
  float Di = 0.f, lse = 0.f, rs = 0.f, qb = 0.f, dsum = 0.f;
  long qrow = 0;
  int tile = first_tile;
  bool act = false;  // this lane holds an owner query of the current tile
  float lnq[4] = {0.f, 0.f, 0.f, 0.f};  // tid < 32: d gq, d bq, d gk, d bk of channel tid, summed over the chunks
  act_t* dkv = p.dkv + (long)part_slot * p.dkv_part_stride;

  for (int kc = 0; kc < n_chunks; ++kc) {
    const int kl = min(32, k_len - kc * 32);
    __syncthreads();  // the previous chunk's epilogue is done with the key arrays and the images
    if (tid < 32) {
      const int pos = k_start + kc * 32 + tid;
      krow_s[tid] = tid < kl ? (p.kidx ? p.kidx[pos] : pos) : -1;
      kext_s[tid] = (tid < kl && p.kext && p.dkv_extra) ? p.kext[pos] : -1;
    }
    __syncthreads();
    xq_load_keys<D>(p, h, krow_s, kl, 32, khat_s, v_s);
    __syncthreads();
    xq_norm_keys<D>(p.eps, kl, 32, khat_s, krstd_s);
    __syncthreads();
    if (tid < 32) {  // kb_j = bq . kn_j
      float a = 0.f;
      for (int c = 0; c < D; ++c) a += (khat_s[tid * 32 + c] * p.kn_w[c] + p.kn_b[c]) * p.qn_b[c];
      kb_s[tid] = a;
    }
    __syncthreads();
    f32x16 accT = zero16(), accV = zero16();
    float cs_acc = 0.f;  // lanes 0..31 of each wave: column sums of dS over the wave's queries
    // The q / dO / O rows of tile ti + 1 are requested before the products of tile ti (a block runs one wave per SIMD:
    // nothing else would hide the round trip) and consumed at the top of the next iteration.
    Raw4<act_t> rq[D / 4], rg[D / 4], ro[D / 4];  // next tile's rows as loaded (converted when taken over: mma.h Raw4)
    bool nact = false;
    long nrow = 0;
    int npos = 0;
    auto fetch = [&](int ti_) {
      const int t_ = first_tile + ti_ * tile_step;
      const int q_start = p.tiles[t_ * 4 + 0], q_len = p.tiles[t_ * 4 + 1];
      npos = q_start + qi;
      nact = qi < q_len && (!p.owner || p.owner[npos]);
      if (nact) {
        nrow = p.qidx ? p.qidx[npos] : npos;
        const act_t* qp = p.q + nrow * p.q_ld + p.q_off + h * D;
        const act_t* gp = p.dout + nrow * p.out_ld + h * D;
        const act_t* op = p.out + nrow * p.out_ld + h * D;
#pragma unroll
        for (int c4 = 0; c4 < D / 4; ++c4) { rq[c4] = ldraw(qp + c4 * 4); rg[c4] = ldraw(gp + c4 * 4); ro[c4] = ldraw(op + c4 * 4); }
      }
    };
    if (kc == 0 || n_tiles > 1) fetch(0);
    for (int ti = 0; ti < n_tiles; ++ti) {
      float* dsr = ds_s + qi * ILD;
      float* pmr = pm_s + qi * ILD;
      if (kc == 0 || n_tiles > 1) {  // take over the query side of this tile
        tile = first_tile + ti * tile_step;
        const int pos = npos;
        act = nact;
        if (act) {
          qrow = nrow;
          Di = 0.f;
#pragma unroll
          for (int c4 = 0; c4 < D / 4; ++c4) {
            const float4 qv_ = unraw(rq[c4]), gv_ = unraw(rg[c4]), ov_ = unraw(ro[c4]);
            qh[4 * c4] = qv_.x; qh[4 * c4 + 1] = qv_.y; qh[4 * c4 + 2] = qv_.z; qh[4 * c4 + 3] = qv_.w;
            go[4 * c4] = gv_.x; go[4 * c4 + 1] = gv_.y; go[4 * c4 + 2] = gv_.z; go[4 * c4 + 3] = gv_.w;
            Di += (gv_.x * ov_.x + gv_.y * ov_.y) + (gv_.z * ov_.z + gv_.w * ov_.w);
          }
          float m = 0.f;
#pragma unroll
          for (int c = 0; c < D; ++c) m += qh[c];
          m /= D;
          float var = 0.f;
#pragma unroll
          for (int c = 0; c < D; ++c) {
            const float t = qh[c] - m;
            var += t * t;
          }
          rs = rsqrtf(var / D + p.eps);
          qb = 0.f;
          dsum = 0.f;
#pragma unroll
          for (int c = 0; c < D; ++c) {
            qh[c] = (qh[c] - m) * rs;
            dx[c] = 0.f;  // accumulates sum_j dS_ij khat_j[c]
            const float qg = qh[c] * p.qn_w[c];
            qgk[c] = qg * p.kn_w[c];
            qb = fmaf(qg, p.kn_b[c], qb);
          }
          lse = p.lse[(long)pos * p.H + h];
#pragma unroll
          for (int c = 0; c < D; ++c) { qh_s[qi * ILD + c] = qh[c]; do_s[qi * ILD + c] = go[c]; }
        } else {
#pragma unroll
          for (int c = 0; c < D; ++c) { qh_s[qi * ILD + c] = 0.f; do_s[qi * ILD + c] = 0.f; }
        }
      }
      if (act) {
        const unsigned long long rb = (((unsigned long long)tile * p.H + h) * AT + qi) * AT + kc * 32;
        const unsigned lo0 = (unsigned)rb, c2 = (unsigned)(rb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
        // two keys per iteration: eight independent dot-product chains and their LDS reads in flight together (a block
        // runs one wave per SIMD, so nothing else hides an LDS round trip or a dependent FMA chain)
        for (int j0 = 0; j0 < kl; j0 += 2) {
          float4 k4[2][D / 4];
          float sc[2], dp[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float4* kr = reinterpret_cast<const float4*>(khat_s + (j0 + e) * 32);
            const float4* vr = reinterpret_cast<const float4*>(v_s + (j0 + e) * 32);
            float a0 = kb_s[j0 + e] + qb, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < D / 4; ++c4) {
              const float4 kk = kr[c4], v4 = vr[c4];
              k4[e][c4] = kk;
              a0 = fmaf(qgk[4 * c4], kk.x, a0); a1 = fmaf(qgk[4 * c4 + 1], kk.y, a1);
              a0 = fmaf(qgk[4 * c4 + 2], kk.z, a0); a1 = fmaf(qgk[4 * c4 + 3], kk.w, a1);
              b0 = fmaf(go[4 * c4], v4.x, b0); b1 = fmaf(go[4 * c4 + 1], v4.y, b1);
              b0 = fmaf(go[4 * c4 + 2], v4.z, b0); b1 = fmaf(go[4 * c4 + 3], v4.w, b1);
            }
            sc[e] = a0 + a1;
            dp[e] = b0 + b1;
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int j = j0 + e;
            const bool in = j < kl;  // (rows kl .. 31 of the chunk are zero rows)
            const float pj = in ? __expf(sc[e] * p.scale - lse) : 0.f;
            const bool keep = !p.drop_thresh || keep_lo(lo0 + (unsigned)j, s0, c2, p.drop_thresh);
            const float pmj = keep ? pj * p.drop_inv_keep : 0.f;
            const float ds = p.scale * pj * ((keep ? dp[e] * p.drop_inv_keep : 0.f) - Di);
            dsr[j] = ds;
            pmr[j] = pmj;
            dsum += ds;
#pragma unroll
            for (int c4 = 0; c4 < D / 4; ++c4) {
              dx[4 * c4] = fmaf(ds, k4[e][c4].x, dx[4 * c4]); dx[4 * c4 + 1] = fmaf(ds, k4[e][c4].y, dx[4 * c4 + 1]);
              dx[4 * c4 + 2] = fmaf(ds, k4[e][c4].z, dx[4 * c4 + 2]); dx[4 * c4 + 3] = fmaf(ds, k4[e][c4].w, dx[4 * c4 + 3]);
            }
          }
        }
        for (int j = (kl + 1) & ~1; j < 32; ++j) { dsr[j] = 0.f; pmr[j] = 0.f; }
        if (kc == n_chunks - 1) {
          // d qhat[c] = gq[c] (gk[c] sum_j dS_ij khat_j[c] + bk[c] sum_j dS_ij), complete after the last chunk; then the
          // q_norm backward: dq = rs (dx - mean(dx) - qhat mean(dx qhat))
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int c = 0; c < D; ++c) {
            dx[c] = p.qn_w[c] * fmaf(p.kn_w[c], dx[c], p.kn_b[c] * dsum);
            a += dx[c];
            b = fmaf(dx[c], qh[c], b);
          }
          a /= D; b /= D;
          act_t* dqp = p.dq + qrow * p.dq_ld + p.dq_off + h * D;
#pragma unroll
          for (int c4 = 0; c4 < D / 4; ++c4)
            st4(dqp + c4 * 4, make_float4(rs * (dx[4 * c4] - a - qh[4 * c4] * b), rs * (dx[4 * c4 + 1] - a - qh[4 * c4 + 1] * b),
                                          rs * (dx[4 * c4 + 2] - a - qh[4 * c4 + 2] * b), rs * (dx[4 * c4 + 3] - a - qh[4 * c4 + 3] * b)));
        }
      } else {
        for (int j = 0; j < 32; ++j) { dsr[j] = 0.f; pmr[j] = 0.f; }
      }
      if (n_tiles > 1 && ti + 1 < n_tiles) fetch(ti + 1);
      __syncthreads();
      {  // this wave's 64 queries: T += dS^T qhat, dV += (P mask)^T dO; lane = (key | channel) l31, k-half hh = query parity
        const int base = wave * 64;
        if (lane < 32) {  // four independent partial sums: the loads of a dependent chain would be exposed one by one
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
          for (int i = 0; i < 64; i += 4) {
            a0 += ds_s[(base + i) * ILD + lane];
            a1 += ds_s[(base + i + 1) * ILD + lane];
            a2 += ds_s[(base + i + 2) * ILD + lane];
            a3 += ds_s[(base + i + 3) * ILD + lane];
          }
          cs_acc += (a0 + a1) + (a2 + a3);
        }
#pragma unroll 8
        for (int kk = 0; kk < 64; kk += 2) {
          const int r = (base + kk + hh) * ILD + l31;
          accT = __builtin_amdgcn_mfma_f32_32x32x2f32(ds_s[r], qh_s[r], accT, 0, 0, 0);
          accV = __builtin_amdgcn_mfma_f32_32x32x2f32(pm_s[r], do_s[r], accV, 0, 0, 0);
        }
      }
      __syncthreads();
    }
    // ---- this chunk's keys: combine the two waves, finish d k / d v, accumulate the LayerNorm parameter gradients.
    // Scratch: T / dV of the two waves in rows 0..63 of the dS / P images, d kn and the q_norm products in rows 64..127
    // (the query images qh_s / do_s must survive: a single-tile block re-uses them for the next chunk)
    float* T_s = ds_s;              // [2][32][ILD]
    float* V_s = pm_s;              // [2][32][ILD]
    float* dkn_s = ds_s + 64 * ILD;  // [32][ILD]
    float* gq_s = pm_s + 64 * ILD;   // [64][ILD]: kn * T, then kn * cs
    if (lane < 32) cs_s[wave][lane] = cs_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hh;
      T_s[(wave * 32 + key) * ILD + l31] = accT[r];
      V_s[(wave * 32 + key) * ILD + l31] = accV[r];
    }
    __syncthreads();
    for (int i = tid; i < 32 * 32; i += 128) {
      const int j = i >> 5, c = i & 31;
      const float T = T_s[j * ILD + c] + T_s[(32 + j) * ILD + c];
      const float cs = cs_s[0][j] + cs_s[1][j];
      const float kn = (c < D) ? khat_s[j * 32 + c] * p.kn_w[c] + p.kn_b[c] : 0.f;
      dkn_s[j * ILD + c] = (c < D && j < kl) ? p.qn_w[c] * T + p.qn_b[c] * cs : 0.f;
      gq_s[j * ILD + c] = (j < kl) ? kn * T : 0.f;
      gq_s[(32 + j) * ILD + c] = (j < kl) ? kn * cs : 0.f;
    }
    __syncthreads();
    if (tid < 32) {  // column sums over the keys
#pragma unroll 8
      for (int j = 0; j < 32; ++j) {
        lnq[0] += gq_s[j * ILD + tid];
        lnq[1] += gq_s[(32 + j) * ILD + tid];
        const float g = dkn_s[j * ILD + tid];
        lnq[2] += g * khat_s[j * 32 + tid];
        lnq[3] += g;
      }
    }
    for (int i = tid; i < 32 * 8; i += 128) {  // k_norm backward per key row (8 lanes per row) + the dV rows
      const int j = i >> 3, c4 = i & 7;
      float g[4], kh[4];
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = c4 * 4 + e;
        kh[e] = khat_s[j * 32 + c];
        g[e] = (c < D) ? dkn_s[j * ILD + c] * p.kn_w[c] : 0.f;
        a += g[e];
        b = fmaf(g[e], kh[e], b);
      }
      a += __shfl_xor(a, 1, 8); a += __shfl_xor(a, 2, 8); a += __shfl_xor(a, 4, 8);
      b += __shfl_xor(b, 1, 8); b += __shfl_xor(b, 2, 8); b += __shfl_xor(b, 4, 8);
      a /= D; b /= D;
      if (j < kl && c4 * 4 < D) {
        const float krs = krstd_s[j];
        const int c = c4 * 4;
        const float4 dk4 = make_float4(krs * (g[0] - a - kh[0] * b), krs * (g[1] - a - kh[1] * b), krs * (g[2] - a - kh[2] * b),
                                       krs * (g[3] - a - kh[3] * b));
        const float4 dv4 = make_float4(V_s[j * ILD + c] + V_s[(32 + j) * ILD + c], V_s[j * ILD + c + 1] + V_s[(32 + j) * ILD + c + 1],
                                       V_s[j * ILD + c + 2] + V_s[(32 + j) * ILD + c + 2], V_s[j * ILD + c + 3] + V_s[(32 + j) * ILD + c + 3]);
        if (kext_s[j] >= 0) {  // borrowed copy of a tail patch: its k | v gradient goes to the side buffer
          act_t* ep = p.dkv_extra + (long)kext_s[j] * p.dkv_extra_ld + h * D + c;
          st4(ep, dk4);
          st4(ep + (p.dv_off - p.dk_off), dv4);
        } else {
          act_t* kp = dkv + (long)krow_s[j] * p.dkv_ld + h * D + c;
          st4(kp + p.dk_off, dk4);
          st4(kp + p.dv_off, dv4);
        }
      }
    }
  }
  if (tid < 32) {
    float* lnp = p.ln_part + ((long)(blockIdx.x * p.H + h) * 4) * 32;
    lnp[tid] = lnq[0]; lnp[32 + tid] = lnq[1]; lnp[64 + tid] = lnq[2]; lnp[96 + tid] = lnq[3];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention backward, round 6: the key side of a block is ONE 32-wide tile (<= 32 instruction tokens of
// a cloud) that never leaves the registers, the queries stream through.  xq_bwd_kernel above keeps a whole query per lane (464
// registers, one wave per SIMD, 1 TB/s); here a query is shared by the two lanes (l31, h) of a wave — lane half h owns the 16
// channels c = (r & 3) + 8 (r >> 2) + 4 h, which is exactly the row set a 32 x 32 x 2 MFMA hands to that lane half — so that ALL five
// products are MFMA tiles whose operands are either persistent key-side registers or the lane's own query-side registers:
//   S^T[key][query]  = Khat * (qhat gq gk)^T + (kb + qb)        A = khat[key = l31][own channels]  (registers, loaded once)
//   dP^T[key][query] = V * dO^T                                 A = v[key = l31][own channels]     (registers, loaded once)
//   dx^T[c][query]   = Khat^T * dS^T                            A = khat[own keys][c = l31]        (registers), B = the lane's dS
//   T[key][c] += dS^T * qhat,  dV[key][c] += Pm^T * dO          over the queries: through wave-private LDS images
// The lane ends up with dx for exactly the channels it loaded, so q_norm backward is 16 in-lane terms + one lane^32 exchange and
// dq leaves as four 16-byte stores per lane.  ~215 registers: two waves per SIMD, two 4-wave blocks per CU.  Head width 32, one
// key chunk, no owner / borrowed rows (the point <-> instruction cross attention); everything else stays on xq_bwd_kernel.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void xq2_bwd_kernel(AttnP p) {
  constexpr int D = 32, XLD = 33, QLD = 36, ILD = 33;
  constexpr int WSZ = 2 * 32 * QLD + 2 * 32 * XLD;  // floats per wave: q image, dO image (row stride 36), dS^T and Pm^T (33)
  __shared__ __attribute__((aligned(16))) float khat_s[32 * 32], v_s[32 * 32], krstd_s[32], kb_s[32], cs_s[NW][32];
  __shared__ int krow_s[32];
  __shared__ __attribute__((aligned(16))) float g1_s[32], bkq_s[32];  // gq gk and gq bk per channel (re-read per use: registers are short)
  __shared__ __attribute__((aligned(16))) float img_s[NW * WSZ];
  const int tid = threadIdx.x, wave = tid >> 6, l31 = tid & 31, hh = (tid >> 5) & 1, h = blockIdx.y;
  const int* bd = p.blocks + blockIdx.x * 6;
  const int first_tile = bd[0], n_tiles = bd[1], tile_step = bd[2], part_slot = bd[3], k_start = bd[4], k_len = bd[5];
  if (k_len > 32) __builtin_trap();  // (the caller promised k_max <= 32)
  if (tid < 32) krow_s[tid] = tid < k_len ? (p.kidx ? p.kidx[k_start + tid] : k_start + tid) : -1;
  __syncthreads();
  xq_load_keys<D>(p, h, krow_s, k_len, 32, khat_s, v_s);
  __syncthreads();
  xq_norm_keys<D>(p.eps, k_len, 32, khat_s, krstd_s);
  __syncthreads();
  if (tid < 32) {  // kb_j = bq . kn_j
    float a = 0.f;
    for (int c = 0; c < D; ++c) a += (khat_s[tid * 32 + c] * p.kn_w[c] + p.kn_b[c]) * p.qn_b[c];
    kb_s[tid] = a;
    g1_s[tid] = p.qn_w[tid] * p.kn_w[tid];
    bkq_s[tid] = p.qn_w[tid] * p.kn_b[tid];
  }
  __syncthreads();
  // persistent key-side fragments of this lane: register r <-> channel (or key) (r & 3) + 8 (r >> 2) + 4 hh
  float KA[16], VA[16], KT[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
    KA[r] = khat_s[l31 * 32 + c];
    VA[r] = v_s[l31 * 32 + c];
    KT[r] = khat_s[c * 32 + l31];
  }
  auto chan4 = [&](const float* tab, int g) { return *reinterpret_cast<const float4*>(tab + 8 * g + 4 * hh); };  // this lane's channels
  f32x16 accT = zero16(), accV = zero16();
  float cs = 0.f;
  float* Qi = img_s + wave * WSZ;
  float* Gi = Qi + 32 * QLD;
  float* Xd = Gi + 32 * QLD;
  float* Xp = Xd + 32 * XLD;
  const unsigned s0 = (unsigned)p.drop_seed;
  const int qi = wave * 32 + l31;
  const int coff = h * D + 4 * hh;  // this lane's channels: coff + 8 g + e

  for (int ti = 0; ti < n_tiles; ++ti) {
    const int tile = first_tile + ti * tile_step;
    const int q_start = p.tiles[tile * 4 + 0], q_len = p.tiles[tile * 4 + 1];
    const int pos = q_start + qi;
    const bool act = qi < q_len;
    const long row = act ? (p.qidx ? (long)p.qidx[pos] : (long)pos) : 0;
    float qh[16], go[16];
    float Di = 0.f;
    {
      const act_t* qp = p.q + row * p.q_ld + p.q_off + coff;
      const act_t* gp = p.dout + row * p.out_ld + coff;
      const act_t* op = p.out + row * p.out_ld + coff;
      float4 qv[4], gv[4], ov[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        qv[g] = act ? ld4(qp + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        gv[g] = act ? ld4(gp + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        ov[g] = act ? ld4(op + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        qh[4 * g] = qv[g].x; qh[4 * g + 1] = qv[g].y; qh[4 * g + 2] = qv[g].z; qh[4 * g + 3] = qv[g].w;
        go[4 * g] = gv[g].x; go[4 * g + 1] = gv[g].y; go[4 * g + 2] = gv[g].z; go[4 * g + 3] = gv[g].w;
        Di += (gv[g].x * ov[g].x + gv[g].y * ov[g].y) + (gv[g].z * ov[g].z + gv[g].w * ov[g].w);
      }
    }
    Di += __shfl_xor(Di, 32, 64);
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) m += qh[r];
    m += __shfl_xor(m, 32, 64);
    m *= (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float t = qh[r] - m;
      var += t * t;
    }
    var += __shfl_xor(var, 32, 64);
    const float rs = rsqrtf(var * (1.f / D) + p.eps);
    float qb = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bk4 = chan4(bkq_s, g);
      const float bk[4] = {bk4.x, bk4.y, bk4.z, bk4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qh[4 * g + e] = (qh[4 * g + e] - m) * rs;
        qb = fmaf(qh[4 * g + e], bk[e], qb);
      }
    }
    qb += __shfl_xor(qb, 32, 64);
    const float lse = act ? p.lse[(long)pos * p.H + h] : INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // the query-side images of the products that reduce over the queries
      *reinterpret_cast<float4*>(Qi + l31 * QLD + 8 * g + 4 * hh) = make_float4(qh[4 * g], qh[4 * g + 1], qh[4 * g + 2], qh[4 * g + 3]);
      *reinterpret_cast<float4*>(Gi + l31 * QLD + 8 * g + 4 * hh) = make_float4(go[4 * g], go[4 * g + 1], go[4 * g + 2], go[4 * g + 3]);
    }
    // ---- S^T and dP^T for this lane's query and the 16 keys of its half
    f32x16 aS, aP = zero16();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 k4 = *reinterpret_cast<const float4*>(kb_s + 8 * g + 4 * hh);
      aS[4 * g] = k4.x + qb; aS[4 * g + 1] = k4.y + qb; aS[4 * g + 2] = k4.z + qb; aS[4 * g + 3] = k4.w + qb;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 g4 = chan4(g1_s, g);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * g + e;
        aS = __builtin_amdgcn_mfma_f32_32x32x2f32(KA[s], qh[s] * gg[e], aS, 0, 0, 0);
        aP = __builtin_amdgcn_mfma_f32_32x32x2f32(VA[s], go[s], aP, 0, 0, 0);
      }
    }
    const unsigned long long rb = (((unsigned long long)tile * p.H + h) * AT + qi) * AT;
    const unsigned lo0 = (unsigned)rb, c2 = (unsigned)(rb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
    float dsr[16];
    float dsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float pj = j < k_len ? __expf(aS[r] * p.scale - lse) : 0.f;
      const bool keep = !p.drop_thresh || keep_lo(lo0 + (unsigned)j, s0, c2, p.drop_thresh);
      const float pmj = keep ? pj * p.drop_inv_keep : 0.f;
      const float ds = p.scale * pj * ((keep ? aP[r] * p.drop_inv_keep : 0.f) - Di);
      dsr[r] = ds;
      dsum += ds;
      Xd[j * XLD + l31] = ds;
      Xp[j * XLD + l31] = pmj;
    }
    dsum += __shfl_xor(dsum, 32, 64);
    // ---- dx^T: this lane's channels of its query, then q_norm backward and the dq row
    f32x16 aX = zero16();
#pragma unroll
    for (int s = 0; s < 16; ++s) aX = __builtin_amdgcn_mfma_f32_32x32x2f32(KT[s], dsr[s], aX, 0, 0, 0);
    float a = 0.f, b = 0.f, dqh[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 g4 = chan4(g1_s, g), bk4 = chan4(bkq_s, g);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bk[4] = {bk4.x, bk4.y, bk4.z, bk4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        dqh[r] = fmaf(gg[e], aX[r], bk[e] * dsum);
        a += dqh[r];
        b = fmaf(dqh[r], qh[r], b);
      }
    }
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    a *= (1.f / D); b *= (1.f / D);
    if (act) {
      act_t* dqp = p.dq + row * p.dq_ld + p.dq_off + coff;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(dqp + 8 * g, make_float4(rs * (dqh[4 * g] - a - qh[4 * g] * b), rs * (dqh[4 * g + 1] - a - qh[4 * g + 1] * b),
                                     rs * (dqh[4 * g + 2] - a - qh[4 * g + 2] * b), rs * (dqh[4 * g + 3] - a - qh[4 * g + 3] * b)));
    }
    __syncthreads();  // the wave's images are complete (block barrier: the waves walk the tiles in step)
    // ---- T += dS^T qhat, dV += Pm^T dO over this wave's 32 queries; lane = (key | channel) l31, lane half = query half
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int qq = hh * 16 + s;
      const float xd = Xd[l31 * XLD + qq], xp = Xp[l31 * XLD + qq];
      cs += xd;
      accT = __builtin_amdgcn_mfma_f32_32x32x2f32(xd, Qi[qq * QLD + l31], accT, 0, 0, 0);
      accV = __builtin_amdgcn_mfma_f32_32x32x2f32(xp, Gi[qq * QLD + l31], accV, 0, 0, 0);
    }
    __syncthreads();  // before the next tile overwrites the images
  }
  // ---- the block's keys: sum the waves, finish d k / d v and the LayerNorm parameter partials (as xq_bwd_kernel)
  float* T_s = img_s;                       // [NW][32][ILD]
  float* V_s = T_s + NW * 32 * ILD;         // [NW][32][ILD]
  float* dkn_s = V_s + NW * 32 * ILD;       // [32][ILD]
  float* gq_s = dkn_s + 32 * ILD;           // [64][ILD]
  static_assert(2 * NW * 32 * ILD + 96 * ILD <= NW * WSZ, "epilogue scratch fits the tile images");
  cs += __shfl_xor(cs, 32, 64);
  if (hh == 0) cs_s[wave][l31] = cs;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = (r & 3) + 8 * (r >> 2) + 4 * hh;
    T_s[(wave * 32 + key) * ILD + l31] = accT[r];
    V_s[(wave * 32 + key) * ILD + l31] = accV[r];
  }
  __syncthreads();
  act_t* dkv = p.dkv + (long)part_slot * p.dkv_part_stride;
  for (int i = tid; i < 32 * 32; i += NW * 64) {
    const int j = i >> 5, c = i & 31;
    float T = 0.f, csj = 0.f, dv = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      T += T_s[(w * 32 + j) * ILD + c];
      dv += V_s[(w * 32 + j) * ILD + c];
      csj += cs_s[w][j];
    }
    const float kn = khat_s[j * 32 + c] * p.kn_w[c] + p.kn_b[c];
    dkn_s[j * ILD + c] = j < k_len ? p.qn_w[c] * T + p.qn_b[c] * csj : 0.f;
    gq_s[j * ILD + c] = j < k_len ? kn * T : 0.f;
    gq_s[(32 + j) * ILD + c] = j < k_len ? kn * csj : 0.f;
    V_s[j * ILD + c] = dv;  // (wave 0's slab now holds the sum: each (j, c) is read and written by this thread only)
  }
  __syncthreads();
  if (tid < 32) {  // column sums over the keys
    float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      l0 += gq_s[j * ILD + tid];
      l1 += gq_s[(32 + j) * ILD + tid];
      const float g = dkn_s[j * ILD + tid];
      l2 += g * khat_s[j * 32 + tid];
      l3 += g;
    }
    float* lnp = p.ln_part + ((long)(blockIdx.x * p.H + h) * 4) * 32;
    lnp[tid] = l0; lnp[32 + tid] = l1; lnp[64 + tid] = l2; lnp[96 + tid] = l3;
  }
  for (int i = tid; i < 32 * 8; i += NW * 64) {  // k_norm backward per key row (8 lanes per row) + the dV rows
    const int j = i >> 3, c4 = i & 7;
    float g[4], kh[4];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c4 * 4 + e;
      kh[e] = khat_s[j * 32 + c];
      g[e] = dkn_s[j * ILD + c] * p.kn_w[c];
      a += g[e];
      b = fmaf(g[e], kh[e], b);
    }
    a += __shfl_xor(a, 1, 8); a += __shfl_xor(a, 2, 8); a += __shfl_xor(a, 4, 8);
    b += __shfl_xor(b, 1, 8); b += __shfl_xor(b, 2, 8); b += __shfl_xor(b, 4, 8);
    a *= (1.f / D); b *= (1.f / D);
    if (j < k_len) {
      const float krs = krstd_s[j];
      const int c = c4 * 4;
      act_t* kp = dkv + (long)krow_s[j] * p.dkv_ld + h * D + c;
      st4(kp + p.dk_off, make_float4(krs * (g[0] - a - kh[0] * b), krs * (g[1] - a - kh[1] * b), krs * (g[2] - a - kh[2] * b),
                                     krs * (g[3] - a - kh[3] * b)));
      st4(kp + p.dv_off, make_float4(V_s[j * ILD + c], V_s[j * ILD + c + 1], V_s[j * ILD + c + 2], V_s[j * ILD + c + 3]));
    }
  }
}

// Cross-attention forward in the same layout (round 6): the scores of a query against the <= 32 keys are one MFMA tile (A = the
// normalised keys, scale folded in: registers), softmax is 16 in-lane values + one lane^32 exchange, O^T = V^T P^T a second tile
// whose B operand is the lane's own probabilities; no LDS beyond the key set-up, no barrier after it.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void xq2_fwd_kernel(AttnP p) {
  constexpr int D = 32;
  __shared__ __attribute__((aligned(16))) float kn_s[32 * 32], v_s[32 * 32], krstd_s[32], gq_s[32], bq_s[32];
  __shared__ int krow_s[32];
  const int tid = threadIdx.x, wave = tid >> 6, l31 = tid & 31, hh = (tid >> 5) & 1, h = blockIdx.y;
  const int q_start = p.tiles[blockIdx.x * 4 + 0], q_len = p.tiles[blockIdx.x * 4 + 1];
  const int k_start = p.tiles[blockIdx.x * 4 + 2], k_len = p.tiles[blockIdx.x * 4 + 3];
  if (k_len > 32) __builtin_trap();  // (the caller promised k_max <= 32)
  if (tid < 32) {
    krow_s[tid] = tid < k_len ? (p.kidx ? p.kidx[k_start + tid] : k_start + tid) : -1;
    gq_s[tid] = p.qn_w[tid];
    bq_s[tid] = p.qn_b[tid];
  }
  __syncthreads();
  xq_load_keys<D>(p, h, krow_s, k_len, 32, kn_s, v_s);
  __syncthreads();
  xq_norm_keys<D>(p.eps, k_len, 32, kn_s, krstd_s);
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += NW * 64) {  // affine part of k_norm, the softmax scale folded in
    const int c = i & 31;
    kn_s[i] = (kn_s[i] * p.kn_w[c] + p.kn_b[c]) * p.scale;
  }
  __syncthreads();
  float KA[16], VT[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
    KA[r] = kn_s[l31 * 32 + c];
    VT[r] = v_s[c * 32 + l31];
  }
  const int qi = wave * 32 + l31;
  const int pos = q_start + qi;
  const bool act = qi < q_len;
  const long row = act ? (p.qidx ? (long)p.qidx[pos] : (long)pos) : 0;
  const int coff = h * D + 4 * hh;
  float qn[16];
  {
    const act_t* qp = p.q + row * p.q_ld + p.q_off + coff;
    float4 qv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) qv[g] = act ? ld4(qp + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 4; ++g) { qn[4 * g] = qv[g].x; qn[4 * g + 1] = qv[g].y; qn[4 * g + 2] = qv[g].z; qn[4 * g + 3] = qv[g].w; }
  }
  float m = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) m += qn[r];
  m += __shfl_xor(m, 32, 64);
  m *= (1.f / D);
  float var = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float t = qn[r] - m;
    var += t * t;
  }
  var += __shfl_xor(var, 32, 64);
  const float rs = rsqrtf(var * (1.f / D) + p.eps);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 g4 = *reinterpret_cast<const float4*>(gq_s + 8 * g + 4 * hh), b4 = *reinterpret_cast<const float4*>(bq_s + 8 * g + 4 * hh);
    qn[4 * g] = fmaf((qn[4 * g] - m) * rs, g4.x, b4.x);
    qn[4 * g + 1] = fmaf((qn[4 * g + 1] - m) * rs, g4.y, b4.y);
    qn[4 * g + 2] = fmaf((qn[4 * g + 2] - m) * rs, g4.z, b4.z);
    qn[4 * g + 3] = fmaf((qn[4 * g + 3] - m) * rs, g4.w, b4.w);
  }
  f32x16 aS = zero16();
#pragma unroll
  for (int s = 0; s < 16; ++s) aS = __builtin_amdgcn_mfma_f32_32x32x2f32(KA[s], qn[s], aS, 0, 0, 0);
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * hh;
    aS[r] = j < k_len ? aS[r] : -INFINITY;
    mx = fmaxf(mx, aS[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const unsigned long long rb = (((unsigned long long)blockIdx.x * p.H + h) * AT + qi) * AT;
  const unsigned lo0 = (unsigned)rb, c2 = (unsigned)(rb >> 32) * 0x7FEB352Du + (unsigned)(p.drop_seed >> 32);
  const unsigned s0 = (unsigned)p.drop_seed;
  float l = 0.f, w[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * hh;
    const float pj = mx > -INFINITY ? __expf(aS[r] - mx) : 0.f;
    l += pj;
    w[r] = (!p.drop_thresh || keep_lo(lo0 + (unsigned)j, s0, c2, p.drop_thresh)) ? pj * p.drop_inv_keep : 0.f;
  }
  l += __shfl_xor(l, 32, 64);
  f32x16 aO = zero16();
#pragma unroll
  for (int s = 0; s < 16; ++s) aO = __builtin_amdgcn_mfma_f32_32x32x2f32(VT[s], w[s], aO, 0, 0, 0);
  if (!act) return;
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (p.lse && hh == 0) p.lse[(long)pos * p.H + h] = mx + logf(l);
  act_t* op = p.out + row * p.out_ld + coff;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    st4(op + 8 * g, make_float4(aO[4 * g] * inv, aO[4 * g + 1] * inv, aO[4 * g + 2] * inv, aO[4 * g + 3] * inv));
}

// Query-per-lane kernels: fp32 operand mode (or a caller-declared short key side, k_max <= 32: cross attention in every
// mode), head widths 32 / 24 / 16, no atomic accumulation; backward needs blocks that are "one key chunk, many tiles" or
// "one tile, many chunks" — k_max (the caller's upper bound of k_len, 0 = unknown: a patch, up to 128) tells which.
// LOTUS_XQ: 0 = tile kernels everywhere, 1 (default) = short key sides only, 2 = also the patch attention in fp32 mode,
// 3 = as 1 but the cross attention on the round-5 one-lane-per-query kernels (A/B of xq2_fwd / xq2_bwd_kernel).
// Measured stand-alone at the bench size (tools/attn_ab.py, us per launch, tile -> query per lane): cross attention forward
// 22 / 36 / 21 / 12.5 -> 15 / 24 / 13 / 11.6, backward 79 / 94 / 73 / 57 -> 66 / 73 / 52 / 40 (levels 0 (C 64), 0 (C 128), 1,
// 2); in the training step 0.95 -> 0.63 ms and +1.2 % throughput.  The 128-key patch attention LOSES on this path (forward
// 68 -> 143, backward 237 -> 396 at level 0, C 128): every lane re-reads each key / value row from LDS (one dword per FMA,
// ~3x what the LDS delivers beside the VALU), where the tile kernels feed 32 x 32 MFMA tiles from one fragment read — so
// mode 2 is a switch for experiments, not the default.
static int xq_mode() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("LOTUS_XQ"); on = e ? atoi(e) : 1; }
  return on;
}
static bool xq_ok(int k_max, int precision, int atomic_out, int d) {
  const bool geom = !atomic_out && (d == 32 || d == 24 || d == 16);
  const bool short_keys = k_max > 0 && k_max <= 32;
  return geom && ((short_keys && xq_mode() >= 1) || (!short_keys && precision == 0 && !LOTUS_ACT_IS_BF16 && xq_mode() == 2));
}

static int check_geom(int H, int d) { return (d % 4 == 0 && d <= 32 && d >= 4 && H > 0) ? 0 : -1; }

extern "C" {

// Forward.  tiles: int32 [ntiles][4] (device).  Rows: q row r lives at q + r*q_ld + q_off + h*d;
// k/v rows at kv + r*kv_ld + {k_off, v_off} + h*d; out rows are indexed like q rows.
int lotus_attention_fwd(const act_t* q, long q_ld, int q_off, const act_t* kv, long kv_ld, int k_off, int v_off,
                        const int* qidx, const int* kidx, const int* owner, const int* tiles, int ntiles,
                        const float* qn_w, const float* qn_b, const float* kn_w, const float* kn_b, act_t* out,
                        long out_ld, float* lse, int H, int d, float scale, float eps, float drop_p,
                        unsigned long long drop_seed, int precision, int k_max, void* stream) {
  LOTUS_CHECK_ARG(q && kv && tiles && out && check_geom(H, d) == 0, "lotus_attention_fwd: bad arguments (H=%d d=%d)", H, d);
  if (ntiles == 0) return LOTUS_OK;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.q = q; p.q_ld = q_ld; p.q_off = q_off; p.kv = kv; p.kv_ld = kv_ld; p.k_off = k_off; p.v_off = v_off;
  p.qidx = qidx; p.kidx = kidx; p.owner = owner; p.tiles = tiles;
  p.qn_w = qn_w; p.qn_b = qn_b; p.kn_w = kn_w; p.kn_b = kn_b;
  p.out = out; p.out_ld = out_ld; p.lse = lse; p.H = H; p.d = d; p.scale = scale; p.eps = eps;
  set_attn_drop(p, drop_p, drop_seed);
  if (xq_ok(k_max, precision, 0, d)) {  // one lane per query, keys as LDS broadcast rows
    if (d == 32 && k_max > 0 && k_max <= 32 && !owner && xq_mode() != 3)  // the cross attention proper: key tile in registers
      LOTUS_LAUNCH(xq2_fwd_kernel<4>, dim3(ntiles, H), dim3(256), 0, (hipStream_t)stream, p);
    else if (d == 32) LOTUS_LAUNCH(xq_fwd_kernel<32>, dim3(ntiles, H), dim3(128), 0, (hipStream_t)stream, p);
    else if (d == 24) LOTUS_LAUNCH(xq_fwd_kernel<24>, dim3(ntiles, H), dim3(128), 0, (hipStream_t)stream, p);
    else LOTUS_LAUNCH(xq_fwd_kernel<16>, dim3(ntiles, H), dim3(128), 0, (hipStream_t)stream, p);
    LOTUS_LAUNCH_CHECK("lotus_attention_fwd(query per lane)");
    return LOTUS_OK;
  }
  const size_t sm = attn_smem_bytes(false);
  const int prec = precision;
  if (prec == 3) {
    { static bool a1 = false; if (!a1) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); a1 = true; } }
    LOTUS_LAUNCH(attn_fwd_kernel<3>, dim3(ntiles, H), dim3(256), sm, (hipStream_t)stream, p);
  } else if (prec == 1) {
    { static bool a2 = false; if (!a2) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); a2 = true; } }
    LOTUS_LAUNCH(attn_fwd_kernel<1>, dim3(ntiles, H), dim3(256), sm, (hipStream_t)stream, p);
  } else {
    { static bool a3 = false; if (!a3) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); a3 = true; } }
    LOTUS_LAUNCH(attn_fwd_kernel<0>, dim3(ntiles, H), dim3(256), sm, (hipStream_t)stream, p);
  }
  LOTUS_LAUNCH_CHECK("lotus_attention_fwd");
  return LOTUS_OK;
}

size_t lotus_attention_bwd_workspace(int nblocks, int H) { return (size_t)nblocks * H * 4 * 32 * sizeof(float); }

// Backward.  blocks: int32 [nblocks][6] = first_tile, n_tiles, tile_step, part_slot, k_start, k_len.
// dq/dkv must be zero-initialised by the caller when atomic_out = 1.  With atomic_out = 0 every
// (part_slot, key row, head) is written exactly once (plain stores, deterministic).
int lotus_attention_bwd(const act_t* q, long q_ld, int q_off, const act_t* kv, long kv_ld, int k_off, int v_off,
                        const int* qidx, const int* kidx, const int* owner, const int* tiles, const int* blocks,
                        int nblocks, const float* qn_w, const float* qn_b, const float* kn_w, const float* kn_b,
                        const act_t* out, const act_t* dout, long out_ld, const float* lse, act_t* dq, long dq_ld,
                        int dq_off, act_t* dkv, long dkv_ld, int dk_off, int dv_off, long dkv_part_stride,
                        int atomic_out, const int* kext, const int* ext_pos, int n_extra, act_t* dkv_extra, float* dqn_w, float* dqn_b, float* dkn_w, float* dkn_b, int accumulate,
                        int H, int d, float scale, float eps, float drop_p, unsigned long long drop_seed,
                        int precision, int k_max, void* workspace, size_t workspace_bytes, void* stream) {
  LOTUS_CHECK_ARG(q && kv && tiles && blocks && out && dout && lse && dq && dkv && check_geom(H, d) == 0,
                  "lotus_attention_bwd: bad arguments");
  LOTUS_CHECK_ARG(workspace && workspace_bytes >= lotus_attention_bwd_workspace(nblocks, H),
                  "lotus_attention_bwd: workspace too small");
  if (nblocks == 0) return LOTUS_OK;
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.q = q; p.q_ld = q_ld; p.q_off = q_off; p.kv = kv; p.kv_ld = kv_ld; p.k_off = k_off; p.v_off = v_off;
  p.qidx = qidx; p.kidx = kidx; p.owner = owner; p.tiles = tiles; p.blocks = blocks;
  p.qn_w = qn_w; p.qn_b = qn_b; p.kn_w = kn_w; p.kn_b = kn_b;
  p.out = (act_t*)out; p.dout = dout; p.out_ld = out_ld; p.lse = (float*)lse;
  p.dq = dq; p.dq_ld = dq_ld; p.dq_off = dq_off;
  p.dkv = dkv; p.dkv_ld = dkv_ld; p.dk_off = dk_off; p.dv_off = dv_off; p.dkv_part_stride = dkv_part_stride;
  p.atomic_out = atomic_out; p.ln_part = (float*)workspace;
  p.kext = kext; p.dkv_extra = (kext && n_extra > 0) ? dkv_extra : nullptr; p.dkv_extra_ld = 2L * H * d;
  LOTUS_CHECK_ARG(!(kext && n_extra > 0) || (dkv_extra && ext_pos && dv_off - dk_off == H * d),
                  "lotus_attention_bwd: borrowed-row buffer needs dkv_extra, ext_pos and adjacent k|v column blocks");
  p.H = H; p.d = d; p.scale = scale; p.eps = eps;
  set_attn_drop(p, drop_p, drop_seed);
  hipStream_t st = (hipStream_t)stream;
  const size_t sm = attn_smem_bytes(true);
  const int prec = precision;
  StopEventOnLast stop_ev;
  if (xq_ok(k_max, precision, atomic_out, d)) {
    // the cross attention proper (head width 32, <= 32 keys, plain rows): keys in registers, queries streamed (xq2_bwd_kernel)
    if (d == 32 && k_max > 0 && k_max <= 32 && !owner && !p.dkv_extra && xq_mode() != 3)
      LOTUS_LAUNCH(xq2_bwd_kernel<4>, dim3(nblocks, H), dim3(256), 0, st, p);
    else if (d == 32) LOTUS_LAUNCH(xq_bwd_kernel<32>, dim3(nblocks, H), dim3(128), 0, st, p);
    else if (d == 24) LOTUS_LAUNCH(xq_bwd_kernel<24>, dim3(nblocks, H), dim3(128), 0, st, p);
    else LOTUS_LAUNCH(xq_bwd_kernel<16>, dim3(nblocks, H), dim3(128), 0, st, p);
  } else if (prec == 3) {
    { static bool a4 = false; if (!a4) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); a4 = true; } }
    LOTUS_LAUNCH(attn_bwd_kernel<3>, dim3(nblocks, H), dim3(256), sm, st, p);
  } else if (prec == 1) {
    { static bool a5 = false; if (!a5) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); a5 = true; } }
    LOTUS_LAUNCH(attn_bwd_kernel<1>, dim3(nblocks, H), dim3(256), sm, st, p);
  } else {
    { static bool a6 = false; if (!a6) { (void)hipFuncSetAttribute((const void*)attn_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); a6 = true; } }
    LOTUS_LAUNCH(attn_bwd2_kernel, dim3(nblocks, H), dim3(256), sm, st, p);
  }
  if (p.dkv_extra) {
    const int w4 = 2 * H * d / 4;
    LOTUS_LAUNCH(attn_extra_fixup_kernel, dim3(cdiv((long)n_extra * w4, 256)), dim3(256), 0, st, dkv_extra, 2L * H * d,
                       ext_pos, kidx, n_extra, w4, dkv, dkv_ld, dk_off);
  }
  stop_ev.last();
  LOTUS_LAUNCH(attn_ln_reduce_kernel, dim3(4), dim3(1024), 0, st, p.ln_part, dqn_w, dqn_b, dkn_w, dkn_b,
                     nblocks * H, d, accumulate);
  LOTUS_LAUNCH_CHECK("lotus_attention_bwd");
  return LOTUS_OK;
}

}  // extern "C"

}  // namespace LOTUS_NS
