// lotus-hip: composite entry points — one C call enqueues the whole forward or backward of a transformer sub-block.
//
// The Python host used to issue the 3-11 launches of a sub-block one C-ABI call at a time (allocation, argument
// conversion and bookkeeping per launch: ~8 us of interpreter time each, ~1000 launches per step).  These functions
// chain the SAME entry points in the SAME order on the same streams, so results are bit-identical to the per-launch
// path (tests/test_gpu_blocks.py); the host allocates three flat buffers per call (saved activations, gradients,
// temporaries) whose layouts are defined here.
//
// Weight gradients go to `side` (0 = same stream as everything else) after `lotus_streamlink_wait(link, main, side)`,
// exactly like ops._OnSide; join != 0 orders `main` after `side` at the end (ops "node" join mode).
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "../../include/lotus_hip.h"

extern __attribute__((visibility("hidden"))) __thread hipEvent_t lotus_tls_stop_event;  // common.h: LOTUS_LAUNCH
hipEvent_t lotus_link_next_event(unsigned long long link);  // lotus_capi.cpp
void lotus_set_error(const char* fmt, ...);

#define CHECK(x)        \
  do {                  \
    int rc_ = (x);      \
    if (rc_) return rc_; \
  } while (0)

static inline size_t al4(size_t n) { return (n + 3) & ~(size_t)3; }  // keep every slice 16-byte aligned

// Activation element type of this build (include/lotus_hip.h: float, or 16-bit bf16 storage in the lotus_b16_* twin).  The
// flat `saved` / `tmp` buffers are carved in bytes: activation slices take n * sizeof(act_t), statistics n * 4, each rounded
// up to 16 bytes; their sizes are still reported in floats (the host allocates fp32 words).  In the fp32 build the
// layout is what it always was.
typedef lotus_act_t act_t;
static inline size_t actf(size_t n) { return al4((n * sizeof(act_t) + 3) / 4); }  // floats holding n activations
struct Carve {
  char* p;
  explicit Carve(const void* base) : p((char*)base) {}
  act_t* act(size_t n) { act_t* r = (act_t*)p; p += actf(n) * 4; return r; }
  float* f32(size_t n) { float* r = (float*)p; p += al4(n) * 4; return r; }
};

// precision 5 = bf16 products with the LINEAR layers' weights given as bf16 shadows (bf16-storage build, gemm.hip); the
// attention and convolution entry points have no such operand and take the plain code
static inline int noshadow(int precision) { return precision == 5 ? 1 : precision; }

static inline int fork_side(unsigned long long link, void* main_s, void* side) {
  return side ? lotus_streamlink_wait(link, main_s, side) : 0;
}

// Fork bound to a producer: the kernels enqueued while a ForkAfter lives carry one of the link's events as their stop
// event (the last launch wins, like re-recording), and wait() orders `side` after it.  Unlike fork_side() this puts no
// event-record marker between the producer and the next kernel of the critical stream.
struct ForkAfter {
  hipEvent_t ev;
  void* side;
  ForkAfter(unsigned long long link, void* side_) : ev(side_ ? lotus_link_next_event(link) : nullptr), side(side_) {
    lotus_tls_stop_event = ev;
  }
  ~ForkAfter() { lotus_tls_stop_event = nullptr; }
  int wait() {
    lotus_tls_stop_event = nullptr;
    if (!side) return 0;
    hipError_t e = hipStreamWaitEvent((hipStream_t)side, ev, 0);
    if (e != hipSuccess) {
      lotus_set_error("lotus composite: hipStreamWaitEvent: %s", hipGetErrorString(e));
      return LOTUS_E_LAUNCH;
    }
    return 0;
  }
};
#define PRODUCE_THEN_FORK(call) \
  do {                          \
    ForkAfter fa_(link, side);  \
    CHECK(call);                \
    CHECK(fa_.wait());          \
  } while (0)

// The input gradient of the layer that reads a LayerNorm's output + that LayerNorm's backward (lotus_linear_dgrad_ln: ONE
// kernel on the many-row levels with C <= 128, else the product into `dn` and lotus_layernorm_bwd); the column partials of
// dgamma / dbeta are reduced on the weight-gradient stream when there is one.
static int dgrad_ln(const act_t* dyl, const float* w, act_t* dn, const act_t* x, const float* mean, const float* rstd, const float* g,
                    const act_t* add, act_t* dx, act_t* dz, float dz_p, unsigned long long dz_seed, float* dg, float* db, int M, int N,
                    int C, int precision, void* ws, size_t ws_bytes, void* counters, float* lnp, size_t lnp_bytes,
                    unsigned long long link, void* stream, void* side) {
  int nparts = 0;
  act_t* dzo = dz_p > 0.f ? dz : nullptr;
  if (side) {
    PRODUCE_THEN_FORK(lotus_linear_dgrad_ln(dyl, w, x, mean, rstd, g, add, dx, dn, dzo, dz_p, dz_seed, M, N, C, precision, ws, ws_bytes,
                                            counters, lnp, lnp_bytes, &nparts, stream));
    CHECK(lotus_layernorm_bwd_params_n(lnp, nparts, C, dg, db, 0, side));
  } else {
    CHECK(lotus_linear_dgrad_ln(dyl, w, x, mean, rstd, g, add, dx, dn, dzo, dz_p, dz_seed, M, N, C, precision, ws, ws_bytes, counters, lnp,
                                lnp_bytes, &nparts, stream));
    CHECK(lotus_layernorm_bwd_params_n(lnp, nparts, C, dg, db, 0, stream));
  }
  return LOTUS_OK;
}

extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// MLP sub-block: y = x + drop(fc2(drop(GELU(fc1(LN(x))))))   (PointTransformerV3/model.py:577-583, :669-673)
//   saved  [n M*C | hpre M*Hd | a M*Hd | mean M | rstd M]
//   grads  [dg C | db C | dw1 Hd*C | db1 Hd | dw2 C*Hd | db2 C]          (dw | db contiguous per layer)
//   tmp    [dz2 M*C | dh M*Hd | dn M*C | ln partials]
size_t lotus_ffn_saved_floats(int M, int C, int Hd) { return actf((size_t)M * C) + 2 * actf((size_t)M * Hd) + 2 * al4((size_t)M); }
size_t lotus_ffn_grads_floats(int C, int Hd) { return 2 * al4(C) + al4((size_t)Hd * C + Hd) + al4((size_t)C * Hd + C); }
size_t lotus_ffn_tmp_floats(int M, int C, int Hd) {
  return 2 * actf((size_t)M * C) + actf((size_t)M * Hd) + lotus_layernorm_bwd_workspace(M, C) / sizeof(float);
}
size_t lotus_ffn_ws_main_bytes(int M, int C, int Hd) {
  const size_t a = lotus_linear_workspace(M, Hd, C), b = lotus_linear_workspace(M, C, Hd);
  return a > b ? a : b;
}
size_t lotus_ffn_ws_side_bytes(int M, int C, int Hd) {
  const size_t a = lotus_linear_wgrad_workspace(M, Hd, C), b = lotus_linear_wgrad_workspace(M, C, Hd);
  return a > b ? a : b;
}

int lotus_ffn_fwd(const act_t* x, const float* g, const float* b, const float* w1, const float* b1, const float* w2,
                  const float* b2, act_t* y, float* saved, int M, int C, int Hd, float drop_p, unsigned long long seed1,
                  unsigned long long seed2, int precision, void* ws, size_t ws_bytes, void* counters, void* stream) {
  Carve sv(saved);
  act_t* n = sv.act((size_t)M * C);
  act_t* hpre = sv.act((size_t)M * Hd);
  act_t* a = sv.act((size_t)M * Hd);
  float* mean = sv.f32(M);
  float* rstd = sv.f32(M);
  const bool big = M > 8192;  // (the per-launch path only offers a split-K workspace to the small-M layers)
  CHECK(lotus_layernorm_fwd(x, nullptr, g, b, n, mean, rstd, M, C, 1e-5f, stream));
  CHECK(lotus_linear_fwd(n, w1, b1, nullptr, a, hpre, M, Hd, C, LOTUS_ACT_GELU, drop_p, seed1, precision, big ? nullptr : ws,
                         big ? 0 : ws_bytes, big ? nullptr : counters, stream));
  return lotus_linear_fwd(a, w2, b2, x, y, nullptr, M, C, Hd, LOTUS_ACT_NONE, drop_p, seed2, precision, big ? nullptr : ws,
                          big ? 0 : ws_bytes, big ? nullptr : counters, stream);
}

// dz_in (optional): dy already multiplied by the fc2 dropout mask (handed over by the next sub-block's backward).
// dz_out (optional, with dz_out_p > 0): dx times the dropout mask (dz_out_p, dz_out_seed) of the PREVIOUS sub-block.
int lotus_ffn_bwd(const act_t* dy, const act_t* dz_in, const act_t* x, const float* g, const float* w1, const float* w2,
                  const float* saved, act_t* dx, act_t* dz_out, float dz_out_p, unsigned long long dz_out_seed, float* grads,
                  float* tmp, int M, int C, int Hd, float drop_p, unsigned long long seed1, unsigned long long seed2,
                  int precision, void* ws_main, size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main,
                  void* counters_side, unsigned long long link, int join, void* stream, void* side) {
  Carve sv(saved);
  const act_t* n = sv.act((size_t)M * C);
  const act_t* hpre = sv.act((size_t)M * Hd);
  const act_t* a = sv.act((size_t)M * Hd);
  const float* mean = sv.f32(M);
  const float* rstd = sv.f32(M);
  float* dg = grads;
  float* db = dg + al4(C);
  float* dw1 = db + al4(C);
  float* db1 = dw1 + (size_t)Hd * C;
  float* dw2 = dw1 + al4((size_t)Hd * C + Hd);
  float* db2 = dw2 + (size_t)C * Hd;
  Carve tp(tmp);
  act_t* dz2 = tp.act((size_t)M * C);
  act_t* dh = tp.act((size_t)M * Hd);
  act_t* dn = tp.act((size_t)M * C);
  float* lnp = tp.f32(0);
  const size_t lnp_bytes = lotus_layernorm_bwd_workspace(M, C);
  void* sw = side ? side : stream;                                   // stream of the weight gradients
  void* wws = side ? ws_side : ws_main;                              // ... and their workspace / counters
  const size_t wws_bytes = side ? ws_side_bytes : ws_main_bytes;
  void* wcnt = side ? counters_side : counters_main;
  const bool big = M > 8192;
  // dz_in was written by the LayerNorm backward of the sub-block that ran just before this one, and the weight-gradient
  // stream is already ordered after that launch (its parameter-gradient reduction waited for it): no fork needed
  const act_t* dz = dz_in;
  if (!dz) {
    if (drop_p > 0.f) {
      PRODUCE_THEN_FORK(lotus_dropout(dy, dz2, (long)M * C, drop_p, seed2, stream));
      dz = dz2;
    } else {
      dz = dy;
      CHECK(fork_side(link, stream, side));
    }
  }
  CHECK(lotus_linear_wgrad(dz, a, dw2, db2, M, C, Hd, 0, precision, wws, wws_bytes, wcnt, sw));
  PRODUCE_THEN_FORK(lotus_linear_dgrad(dz, w2, dh, hpre, nullptr, M, C, Hd, LOTUS_ACT_GELU, drop_p, seed1, precision,
                                       big ? nullptr : ws_main, big ? 0 : ws_main_bytes, big ? nullptr : counters_main, stream));
  CHECK(lotus_linear_wgrad(dh, n, dw1, db1, M, Hd, C, 0, precision, wws, wws_bytes, wcnt, sw));
  CHECK(dgrad_ln(dh, w1, dn, x, mean, rstd, g, dy, dx, dz_out, dz_out_p, dz_out_seed, dg, db, M, Hd, C, precision, big ? nullptr : ws_main,
                 big ? 0 : ws_main_bytes, big ? nullptr : counters_main, lnp, lnp_bytes, link, stream, side));
  if (side && join) CHECK(lotus_streamlink_wait(link, side, stream));
  return LOTUS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Patch self-attention sub-block: y = x + drop(proj(PatchAttention(qkv(LN(x)))))   (model.py:468-557, :664-667)
//   saved [n M*C | qkv M*3C | att M*C | lse npad*H | mean M | rstd M]
//   grads [dg C | db C | dwqkv 3C*C + dbqkv 3C | gq d | bq d | gk d | bk d | dwp C*C + dbp C]
//   tmp   [dz M*C | datt M*C | dqkv M*3C | extra max(n_extra,1)*2C | dn M*C | ln partials]
size_t lotus_selfattn_saved_floats(int M, int C, int H, int npad) {
  return 2 * actf((size_t)M * C) + actf((size_t)M * 3 * C) + al4((size_t)npad * H) + 2 * al4((size_t)M);
}
size_t lotus_selfattn_grads_floats(int C, int H) {
  return 2 * al4(C) + al4((size_t)3 * C * C + 3 * C) + 4 * al4(C / H) + al4((size_t)C * C + C);
}
size_t lotus_selfattn_tmp_floats(int M, int C, int n_extra) {
  return 3 * actf((size_t)M * C) + actf((size_t)M * 3 * C) + actf((size_t)(n_extra > 1 ? n_extra : 1) * 2 * C) +
         lotus_layernorm_bwd_workspace(M, C) / sizeof(float);
}
size_t lotus_selfattn_ws_main_bytes(int M, int C, int H, int nblocks) {
  size_t a = lotus_linear_workspace(M, 3 * C, C), b = lotus_linear_workspace(M, C, C), c = lotus_attention_bwd_workspace(nblocks, H);
  if (b > a) a = b;
  return c > a ? c : a;
}
size_t lotus_selfattn_ws_side_bytes(int M, int C) {
  const size_t a = lotus_linear_wgrad_workspace(M, 3 * C, C), b = lotus_linear_wgrad_workspace(M, C, C);
  return a > b ? a : b;
}

int lotus_selfattn_fwd(const act_t* x, const float* g, const float* b, const float* wqkv, const float* bqkv, const float* qnw,
                       const float* qnb, const float* knw, const float* knb, const float* wp, const float* bp, act_t* y,
                       float* saved, const int* gidx, const int* owner, const int* tiles, int ntiles, int npad, int M, int C,
                       int H, float scale, float drop_p, unsigned long long seed, float attn_p, unsigned long long attn_seed,
                       int precision, void* ws, size_t ws_bytes, void* counters, void* stream) {
  const int d = C / H;
  Carve sv(saved);
  act_t* n = sv.act((size_t)M * C);
  act_t* qkv = sv.act((size_t)M * 3 * C);
  act_t* att = sv.act((size_t)M * C);
  float* lse = sv.f32((size_t)npad * H);
  float* mean = sv.f32(M);
  float* rstd = sv.f32(M);
  const bool big = M > 8192;
  CHECK(lotus_layernorm_fwd(x, nullptr, g, b, n, mean, rstd, M, C, 1e-5f, stream));
  CHECK(lotus_linear_fwd(n, wqkv, bqkv, nullptr, qkv, nullptr, M, 3 * C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws,
                         big ? 0 : ws_bytes, big ? nullptr : counters, stream));
  CHECK(lotus_attention_fwd(qkv, 3L * C, 0, qkv, 3L * C, C, 2 * C, gidx, gidx, owner, tiles, ntiles, qnw, qnb, knw, knb, att, (long)C, lse,
                            H, d, scale, 1e-6f, attn_p, attn_seed, noshadow(precision), 0, stream));
  return lotus_linear_fwd(att, wp, bp, x, y, nullptr, M, C, C, LOTUS_ACT_NONE, drop_p, seed, precision, big ? nullptr : ws,
                          big ? 0 : ws_bytes, big ? nullptr : counters, stream);
}

int lotus_selfattn_bwd(const act_t* dy, const act_t* dz_in, const act_t* x, const float* g, const float* wqkv, const float* qnw,
                       const float* qnb, const float* knw, const float* knb, const float* wp, const float* saved, act_t* dx,
                       float* grads, float* tmp, const int* gidx, const int* owner, const int* tiles, const int* blocks, int nblocks,
                       const int* kext, const int* ext_pos, int n_extra, int npad, int M, int C, int H, float scale, float drop_p,
                       unsigned long long seed, float attn_p, unsigned long long attn_seed, int precision, void* ws_main,
                       size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main, void* counters_side,
                       unsigned long long link, int join, void* stream, void* side) {
  const int d = C / H;
  Carve sv(saved);
  const act_t* n = sv.act((size_t)M * C);
  const act_t* qkv = sv.act((size_t)M * 3 * C);
  const act_t* att = sv.act((size_t)M * C);
  const float* lse = sv.f32((size_t)npad * H);
  const float* mean = sv.f32(M);
  const float* rstd = sv.f32(M);
  float* dg = grads;
  float* db = dg + al4(C);
  float* dwqkv = db + al4(C);
  float* dbqkv = dwqkv + (size_t)3 * C * C;
  float* gq = dwqkv + al4((size_t)3 * C * C + 3 * C);
  float* bq = gq + al4(d);
  float* gk = bq + al4(d);
  float* bk = gk + al4(d);
  float* dwp = bk + al4(d);
  float* dbp = dwp + (size_t)C * C;
  Carve tp(tmp);
  act_t* dzb = tp.act((size_t)M * C);
  act_t* datt = tp.act((size_t)M * C);
  act_t* dqkv = tp.act((size_t)M * 3 * C);
  act_t* extra = tp.act((size_t)(n_extra > 1 ? n_extra : 1) * 2 * C);
  act_t* dn = tp.act((size_t)M * C);
  float* lnp = tp.f32(0);
  const size_t lnp_bytes = lotus_layernorm_bwd_workspace(M, C);
  void* sw = side ? side : stream;
  void* wws = side ? ws_side : ws_main;
  const size_t wws_bytes = side ? ws_side_bytes : ws_main_bytes;
  void* wcnt = side ? counters_side : counters_main;
  const bool big = M > 8192;
  // dz_in was written by the LayerNorm backward of the sub-block that ran just before this one, and the weight-gradient
  // stream is already ordered after that launch (its parameter-gradient reduction waited for it): no fork needed
  const act_t* dz = dz_in;
  if (!dz) {
    if (drop_p > 0.f) {
      PRODUCE_THEN_FORK(lotus_dropout(dy, dzb, (long)M * C, drop_p, seed, stream));
      dz = dzb;
    } else {
      dz = dy;
      CHECK(fork_side(link, stream, side));
    }
  }
  CHECK(lotus_linear_wgrad(dz, att, dwp, dbp, M, C, C, 0, precision, wws, wws_bytes, wcnt, sw));
  CHECK(lotus_linear_dgrad(dz, wp, datt, nullptr, nullptr, M, C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws_main,
                           big ? 0 : ws_main_bytes, big ? nullptr : counters_main, stream));
  PRODUCE_THEN_FORK(lotus_attention_bwd(qkv, 3L * C, 0, qkv, 3L * C, C, 2 * C, gidx, gidx, owner, tiles, blocks, nblocks, qnw, qnb, knw, knb,
                                        att, datt, (long)C, lse, dqkv, 3L * C, 0, dqkv, 3L * C, C, 2 * C, 0, 0, kext, ext_pos, n_extra, extra,
                                        gq, bq, gk, bk, 0, H, d, scale, 1e-6f, attn_p, attn_seed, noshadow(precision), 0, ws_main, ws_main_bytes, stream));
  CHECK(lotus_linear_wgrad(dqkv, n, dwqkv, dbqkv, M, 3 * C, C, 0, precision, wws, wws_bytes, wcnt, sw));
  CHECK(dgrad_ln(dqkv, wqkv, dn, x, mean, rstd, g, dy, dx, nullptr, 0.f, 0, dg, db, M, 3 * C, C, precision, big ? nullptr : ws_main,
                 big ? 0 : ws_main_bytes, big ? nullptr : counters_main, lnp, lnp_bytes, link, stream, side));
  if (side && join) CHECK(lotus_streamlink_wait(link, side, stream));
  return LOTUS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention sub-block: y = x + drop(proj(CrossAttention(q(LN(x)), kv(context))))   (model_ca.py:46-101, :135-140)
//   saved [n M*C | q M*C | kv L*2C | att M*C | lse M*H | mean M | rstd M]
//   grads [dg C | db C | dwq C*C + dbq C | dwkv 2C*Cc + dbkv 2C | gq d | bq d | gk d | bk d | dwp C*C + dbp C]
//   tmp   [dz M*C | datt M*C | dq M*C | dkv_part G*L*2C | dkv L*2C | dn M*C | ln partials]
size_t lotus_crossattn_saved_floats(int M, int C, int H, int L) {
  return 3 * actf((size_t)M * C) + actf((size_t)L * 2 * C) + al4((size_t)M * H) + 2 * al4((size_t)M);
}
size_t lotus_crossattn_grads_floats(int C, int H, int Cc) {
  return 2 * al4(C) + 2 * al4((size_t)C * C + C) + al4((size_t)2 * C * Cc + 2 * C) + 4 * al4(C / H);
}
size_t lotus_crossattn_tmp_floats(int M, int C, int L, int G) {
  return 4 * actf((size_t)M * C) + actf((size_t)G * L * 2 * C) + actf((size_t)L * 2 * C) + lotus_layernorm_bwd_workspace(M, C) / sizeof(float);
}
size_t lotus_crossattn_ws_main_bytes(int M, int C, int H, int L, int Cc, int nblocks) {
  size_t a = lotus_linear_workspace(M, C, C), b = lotus_linear_workspace(L, 2 * C, Cc), c = lotus_attention_bwd_workspace(nblocks, H);
  if (b > a) a = b;
  return c > a ? c : a;
}
size_t lotus_crossattn_ws_side_bytes(int M, int C, int L, int Cc) {
  const size_t a = lotus_linear_wgrad_workspace(M, C, C), b = lotus_linear_wgrad_workspace(L, 2 * C, Cc);
  return a > b ? a : b;
}

int lotus_crossattn_fwd(const act_t* x, const act_t* context, const float* g, const float* b, const float* wq, const float* bq,
                        const float* wkv, const float* bkv, const float* qnw, const float* qnb, const float* knw, const float* knb,
                        const float* wp, const float* bp, act_t* y, float* saved, const int* tiles, int ntiles, int M, int C, int H,
                        int L, int Cc, float scale, float drop_p, unsigned long long seed, float attn_p, unsigned long long attn_seed,
                        int precision, int k_max, void* ws, size_t ws_bytes, void* counters, void* stream) {
  const int d = C / H;
  Carve sv(saved);
  act_t* n = sv.act((size_t)M * C);
  act_t* q = sv.act((size_t)M * C);
  act_t* kv = sv.act((size_t)L * 2 * C);
  act_t* att = sv.act((size_t)M * C);
  float* lse = sv.f32((size_t)M * H);
  float* mean = sv.f32(M);
  float* rstd = sv.f32(M);
  const bool big = M > 8192, bigL = L > 8192;
  CHECK(lotus_layernorm_fwd(x, nullptr, g, b, n, mean, rstd, M, C, 1e-5f, stream));
  CHECK(lotus_linear_fwd(n, wq, bq, nullptr, q, nullptr, M, C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws, big ? 0 : ws_bytes,
                         big ? nullptr : counters, stream));
  CHECK(lotus_linear_fwd(context, wkv, bkv, nullptr, kv, nullptr, L, 2 * C, Cc, LOTUS_ACT_NONE, 0.f, 0, precision, bigL ? nullptr : ws,
                         bigL ? 0 : ws_bytes, bigL ? nullptr : counters, stream));
  CHECK(lotus_attention_fwd(q, (long)C, 0, kv, 2L * C, 0, C, nullptr, nullptr, nullptr, tiles, ntiles, qnw, qnb, knw, knb, att, (long)C, lse, H,
                            d, scale, 1e-6f, attn_p, attn_seed, noshadow(precision), k_max, stream));
  return lotus_linear_fwd(att, wp, bp, x, y, nullptr, M, C, C, LOTUS_ACT_NONE, drop_p, seed, precision, big ? nullptr : ws,
                          big ? 0 : ws_bytes, big ? nullptr : counters, stream);
}

// dctx (optional): gradient of the context [L][Cc].  G = key-side partial slots of the attention backward.
int lotus_crossattn_bwd(const act_t* dy, const act_t* dz_in, const act_t* x, const act_t* context, const float* g, const float* wq,
                        const float* wkv, const float* qnw, const float* qnb, const float* knw, const float* knb, const float* wp,
                        const float* saved, act_t* dx, act_t* dctx, act_t* dz_out, float dz_out_p, unsigned long long dz_out_seed,
                        float* grads, float* tmp, const int* tiles, const int* blocks, int nblocks, int G, int M, int C, int H, int L,
                        int Cc, float scale, float drop_p, unsigned long long seed, float attn_p, unsigned long long attn_seed,
                        int precision, int k_max, void* ws_main, size_t ws_main_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main,
                        void* counters_side, unsigned long long link, int join, void* stream, void* side) {
  const int d = C / H;
  Carve sv(saved);
  const act_t* n = sv.act((size_t)M * C);
  const act_t* q = sv.act((size_t)M * C);
  const act_t* kv = sv.act((size_t)L * 2 * C);
  const act_t* att = sv.act((size_t)M * C);
  const float* lse = sv.f32((size_t)M * H);
  const float* mean = sv.f32(M);
  const float* rstd = sv.f32(M);
  float* dg = grads;
  float* db = dg + al4(C);
  float* dwq = db + al4(C);
  float* dbq = dwq + (size_t)C * C;
  float* dwkv = dwq + al4((size_t)C * C + C);
  float* dbkv = dwkv + (size_t)2 * C * Cc;
  float* gq = dwkv + al4((size_t)2 * C * Cc + 2 * C);
  float* bq_ = gq + al4(d);
  float* gk = bq_ + al4(d);
  float* bk_ = gk + al4(d);
  float* dwp = bk_ + al4(d);
  float* dbp = dwp + (size_t)C * C;
  Carve tp(tmp);
  act_t* dzb = tp.act((size_t)M * C);
  act_t* datt = tp.act((size_t)M * C);
  act_t* dq = tp.act((size_t)M * C);
  act_t* dkv_part = tp.act((size_t)G * L * 2 * C);
  act_t* dkv = tp.act((size_t)L * 2 * C);
  act_t* dn = tp.act((size_t)M * C);
  float* lnp = tp.f32(0);
  const size_t lnp_bytes = lotus_layernorm_bwd_workspace(M, C);
  void* sw = side ? side : stream;
  void* wws = side ? ws_side : ws_main;
  const size_t wws_bytes = side ? ws_side_bytes : ws_main_bytes;
  void* wcnt = side ? counters_side : counters_main;
  const bool big = M > 8192, bigL = L > 8192;
  // dz_in was written by the LayerNorm backward of the sub-block that ran just before this one, and the weight-gradient
  // stream is already ordered after that launch (its parameter-gradient reduction waited for it): no fork needed
  const act_t* dz = dz_in;
  if (!dz) {
    if (drop_p > 0.f) {
      PRODUCE_THEN_FORK(lotus_dropout(dy, dzb, (long)M * C, drop_p, seed, stream));
      dz = dzb;
    } else {
      dz = dy;
      CHECK(fork_side(link, stream, side));
    }
  }
  CHECK(lotus_linear_wgrad(dz, att, dwp, dbp, M, C, C, 0, precision, wws, wws_bytes, wcnt, sw));
  CHECK(lotus_linear_dgrad(dz, wp, datt, nullptr, nullptr, M, C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws_main,
                           big ? 0 : ws_main_bytes, big ? nullptr : counters_main, stream));
  const act_t* dkv_f = dkv_part;
  {
    ForkAfter fa(link, side);  // dq and d kv: the last launch in here carries the fork event
    if (G > 1) lotus_tls_stop_event = nullptr;
    CHECK(lotus_attention_bwd(q, (long)C, 0, kv, 2L * C, 0, C, nullptr, nullptr, nullptr, tiles, blocks, nblocks, qnw, qnb, knw, knb, att,
                              datt, (long)C, lse, dq, (long)C, 0, dkv_part, 2L * C, 0, C, (long)L * 2 * C, 0, nullptr, nullptr, 0, nullptr, gq,
                              bq_, gk, bk_, 0, H, d, scale, 1e-6f, attn_p, attn_seed, noshadow(precision), k_max, ws_main, ws_main_bytes, stream));
    if (G > 1) {  // fixed-order sum of the key-side partial slots
      lotus_tls_stop_event = fa.ev;
      CHECK(lotus_sum_slabs(dkv_part, dkv, (long)L * 2 * C, (long)L * 2 * C, G, stream));
      dkv_f = dkv;
    }
    CHECK(fa.wait());
  }
  CHECK(lotus_linear_wgrad(dkv_f, context, dwkv, dbkv, L, 2 * C, Cc, 0, precision, wws, wws_bytes, wcnt, sw));
  CHECK(lotus_linear_wgrad(dq, n, dwq, dbq, M, C, C, 0, precision, wws, wws_bytes, wcnt, sw));
  if (dctx)
    CHECK(lotus_linear_dgrad(dkv_f, wkv, dctx, nullptr, nullptr, L, 2 * C, Cc, LOTUS_ACT_NONE, 0.f, 0, precision, bigL ? nullptr : ws_main,
                             bigL ? 0 : ws_main_bytes, bigL ? nullptr : counters_main, stream));
  CHECK(dgrad_ln(dq, wq, dn, x, mean, rstd, g, dy, dx, dz_out, dz_out_p, dz_out_seed, dg, db, M, C, C, precision, big ? nullptr : ws_main,
                 big ? 0 : ws_main_bytes, big ? nullptr : counters_main, lnp, lnp_bytes, link, stream, side));
  if (side && join) CHECK(lotus_streamlink_wait(link, side, stream));
  return LOTUS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention sub-block with PRECOMPUTED keys / values.  The context is the same for every CABlock of a forward pass
// (model_ca.py:46-67: kv = Linear(256 -> 2C)(context)), so the model projects it once for all blocks — one product
// [L, 256] x [256, sum 2C] — and each block reads its column slice `kv` (row stride kv_ld) of that slab; backward writes
// d kv into the block's slice `dkv` (row stride dkv_ld) of the shared gradient slab, from which ONE input-gradient and ONE
// weight-gradient product follow (ops.KvAllFn).  Everything else is lotus_crossattn_fwd / _bwd.
//   saved [n M*C | q M*C | att M*C | lse M*H | mean M | rstd M]
//   grads [dg C | db C | dwq C*C + dbq C | gq d | bq d | gk d | bk d | dwp C*C + dbp C]
//   tmp   [dz M*C | datt M*C | dq M*C | dkv_part (G > 1 ? G : 0)*L*2C | dn M*C | ln partials]
size_t lotus_crossattn_kv_saved_floats(int M, int C, int H) { return 3 * actf((size_t)M * C) + al4((size_t)M * H) + 2 * al4((size_t)M); }
size_t lotus_crossattn_kv_grads_floats(int C, int H) { return 2 * al4(C) + 2 * al4((size_t)C * C + C) + 4 * al4(C / H); }
size_t lotus_crossattn_kv_tmp_floats(int M, int C, int L, int G) {
  return 4 * actf((size_t)M * C) + (G > 1 ? actf((size_t)G * L * 2 * C) : 0) + lotus_layernorm_bwd_workspace(M, C) / sizeof(float);
}
size_t lotus_crossattn_kv_ws_main_bytes(int M, int C, int H, int nblocks) {
  const size_t a = lotus_linear_workspace(M, C, C), c = lotus_attention_bwd_workspace(nblocks, H);
  return c > a ? c : a;
}
size_t lotus_crossattn_kv_ws_side_bytes(int M, int C) { return lotus_linear_wgrad_workspace(M, C, C); }

int lotus_crossattn_kv_fwd(const act_t* x, const act_t* kv, long kv_ld, const float* g, const float* b, const float* wq, const float* bq,
                           const float* qnw, const float* qnb, const float* knw, const float* knb, const float* wp, const float* bp,
                           act_t* y, float* saved, const int* tiles, int ntiles, int M, int C, int H, float scale, float drop_p,
                           unsigned long long seed, float attn_p, unsigned long long attn_seed, int precision, int k_max, void* ws,
                           size_t ws_bytes, void* counters, void* stream) {
  const int d = C / H;
  Carve sv(saved);
  act_t* n = sv.act((size_t)M * C);
  act_t* q = sv.act((size_t)M * C);
  act_t* att = sv.act((size_t)M * C);
  float* lse = sv.f32((size_t)M * H);
  float* mean = sv.f32(M);
  float* rstd = sv.f32(M);
  const bool big = M > 8192;
  CHECK(lotus_layernorm_fwd(x, nullptr, g, b, n, mean, rstd, M, C, 1e-5f, stream));
  CHECK(lotus_linear_fwd(n, wq, bq, nullptr, q, nullptr, M, C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws, big ? 0 : ws_bytes,
                         big ? nullptr : counters, stream));
  CHECK(lotus_attention_fwd(q, (long)C, 0, kv, kv_ld, 0, C, nullptr, nullptr, nullptr, tiles, ntiles, qnw, qnb, knw, knb, att, (long)C, lse, H,
                            d, scale, 1e-6f, attn_p, attn_seed, noshadow(precision), k_max, stream));
  return lotus_linear_fwd(att, wp, bp, x, y, nullptr, M, C, C, LOTUS_ACT_NONE, drop_p, seed, precision, big ? nullptr : ws,
                          big ? 0 : ws_bytes, big ? nullptr : counters, stream);
}

int lotus_crossattn_kv_bwd(const act_t* dy, const act_t* dz_in, const act_t* x, const act_t* kv, long kv_ld, const float* g,
                           const float* wq, const float* qnw, const float* qnb, const float* knw, const float* knb, const float* wp,
                           const float* saved, act_t* dx, act_t* dkv, long dkv_ld, act_t* dz_out, float dz_out_p,
                           unsigned long long dz_out_seed, float* grads, float* tmp, const int* tiles, const int* blocks, int nblocks,
                           int G, int M, int C, int H, int L, float scale, float drop_p, unsigned long long seed, float attn_p,
                           unsigned long long attn_seed, int precision, int k_max, void* ws_main, size_t ws_main_bytes, void* ws_side,
                           size_t ws_side_bytes, void* counters_main, void* counters_side, unsigned long long link, int join,
                           void* stream, void* side) {
  const int d = C / H;
  Carve sv(saved);
  const act_t* n = sv.act((size_t)M * C);
  const act_t* q = sv.act((size_t)M * C);
  const act_t* att = sv.act((size_t)M * C);
  const float* lse = sv.f32((size_t)M * H);
  const float* mean = sv.f32(M);
  const float* rstd = sv.f32(M);
  float* dg = grads;
  float* db = dg + al4(C);
  float* dwq = db + al4(C);
  float* dbq = dwq + (size_t)C * C;
  float* gq = dwq + al4((size_t)C * C + C);
  float* bq_ = gq + al4(d);
  float* gk = bq_ + al4(d);
  float* bk_ = gk + al4(d);
  float* dwp = bk_ + al4(d);
  float* dbp = dwp + (size_t)C * C;
  Carve tp(tmp);
  act_t* dzb = tp.act((size_t)M * C);
  act_t* datt = tp.act((size_t)M * C);
  act_t* dq = tp.act((size_t)M * C);
  act_t* dkv_part = G > 1 ? tp.act((size_t)G * L * 2 * C) : nullptr;
  act_t* dn = tp.act((size_t)M * C);
  float* lnp = tp.f32(0);
  const size_t lnp_bytes = lotus_layernorm_bwd_workspace(M, C);
  void* sw = side ? side : stream;
  void* wws = side ? ws_side : ws_main;
  const size_t wws_bytes = side ? ws_side_bytes : ws_main_bytes;
  void* wcnt = side ? counters_side : counters_main;
  const bool big = M > 8192;
  const act_t* dz = dz_in;
  if (!dz) {
    if (drop_p > 0.f) {
      PRODUCE_THEN_FORK(lotus_dropout(dy, dzb, (long)M * C, drop_p, seed, stream));
      dz = dzb;
    } else {
      dz = dy;
      CHECK(fork_side(link, stream, side));
    }
  }
  CHECK(lotus_linear_wgrad(dz, att, dwp, dbp, M, C, C, 0, precision, wws, wws_bytes, wcnt, sw));
  CHECK(lotus_linear_dgrad(dz, wp, datt, nullptr, nullptr, M, C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws_main,
                           big ? 0 : ws_main_bytes, big ? nullptr : counters_main, stream));
  // d q (-> the weight gradient of the q projection on the side stream) and d kv: with one key-side slot the attention
  // backward writes the block's slice of the shared gradient slab directly, else the slots are summed into it
  PRODUCE_THEN_FORK(lotus_attention_bwd(q, (long)C, 0, kv, kv_ld, 0, C, nullptr, nullptr, nullptr, tiles, blocks, nblocks, qnw, qnb, knw, knb,
                                        att, datt, (long)C, lse, dq, (long)C, 0, G > 1 ? dkv_part : dkv, G > 1 ? 2L * C : dkv_ld, 0, C,
                                        G > 1 ? (long)L * 2 * C : 0, 0, nullptr, nullptr, 0, nullptr, gq, bq_, gk, bk_, 0, H, d, scale,
                                        1e-6f, attn_p, attn_seed, noshadow(precision), k_max, ws_main, ws_main_bytes, stream));
  if (G > 1) CHECK(lotus_sum_slabs_ld(dkv_part, dkv, L, 2 * C, dkv_ld, (long)L * 2 * C, G, stream));
  CHECK(lotus_linear_wgrad(dq, n, dwq, dbq, M, C, C, 0, precision, wws, wws_bytes, wcnt, sw));
  CHECK(dgrad_ln(dq, wq, dn, x, mean, rstd, g, dy, dx, dz_out, dz_out_p, dz_out_seed, dg, db, M, C, C, precision, big ? nullptr : ws_main,
                 big ? 0 : ws_main_bytes, big ? nullptr : counters_main, lnp, lnp_bytes, link, stream, side));
  if (side && join) CHECK(lotus_streamlink_wait(link, side, stream));
  return LOTUS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Conditional positional encoding: y = x + LN(Linear(SubMConv3d_3(xs)))   (model.py:615-625, :660-662); xs == x in the
// encoder, the stale skip branch in the decoder (SURVEY.md Trap 3).
//   saved [c n*C | l n*C | mean n | rstd n]
//   grads [dg C | db C | dlw C*C + dlb C | dcw C*27*C + dcb C]
//   tmp   [dl n*C | dc n*C | dyr n*C | ln partials]
size_t lotus_cpe_saved_floats(int n, int C) { return 2 * actf((size_t)n * C) + 2 * al4((size_t)n); }
size_t lotus_cpe_grads_floats(int C) { return 2 * al4(C) + al4((size_t)C * C + C) + al4((size_t)C * 27 * C + C); }
size_t lotus_cpe_tmp_floats(int n, int C) { return 3 * actf((size_t)n * C) + lotus_layernorm_bwd_workspace(n, C) / sizeof(float); }
size_t lotus_cpe_ws_main_bytes(int n, int C) { return lotus_linear_workspace(n, C, C); }
size_t lotus_cpe_ws_conv_bytes(int n, int C) { return lotus_subm_conv_workspace(n, C, C); }
size_t lotus_cpe_ws_side_bytes(int n, int C) {
  const size_t a = lotus_linear_wgrad_workspace(n, C, C), b = lotus_subm_conv_wgrad_workspace(n, 27, C, C);
  return a > b ? a : b;
}

int lotus_cpe_fwd(const act_t* x, const act_t* xs, const float* cw, const float* cw_packed, const float* cb, const float* lw,
                  const float* lb, const float* g, const float* b, act_t* y, float* saved, const int* nbr27, const int* order0,
                  const int* tap_plan, int n, int C, int precision, void* ws, size_t ws_bytes, void* ws_conv, size_t ws_conv_bytes, void* counters, void* stream) {
  Carve sv(saved);
  act_t* c = sv.act((size_t)n * C);
  act_t* l = sv.act((size_t)n * C);
  float* mean = sv.f32(n);
  float* rstd = sv.f32(n);
  const bool big = n > 8192;
  CHECK(lotus_subm_conv(0, xs, cw, cw_packed, cb, nullptr, c, nbr27, order0, n, 27, C, C, noshadow(precision), tap_plan, ws_conv, ws_conv_bytes, stream));
  CHECK(lotus_linear_fwd(c, lw, lb, nullptr, l, nullptr, n, C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws, big ? 0 : ws_bytes,
                         big ? nullptr : counters, stream));
  return lotus_layernorm_fwd(l, x, g, b, y, mean, rstd, n, C, 1e-5f, stream);
}

// dx_conv = input gradient of the convolution (+ dy when add_dy: the encoder case, where it IS d x).  n_dup != 0: the
// level holds several points per voxel (code0 / order0 of the level drive the fold, nbr27[13] the mask).
int lotus_cpe_bwd(const act_t* dy, const act_t* xs, const float* cw, const float* cw_packed, const float* lw, const float* g,
                  const float* saved, act_t* dx_conv, int add_dy, float* grads, float* tmp, const int* nbr27, const int* order0,
                  const int* tap_plan, const long long* code0, int n_dup, int n, int C, int precision, void* ws_main, size_t ws_main_bytes, void* ws_conv,
                  size_t ws_conv_bytes, void* ws_side, size_t ws_side_bytes, void* counters_main, void* counters_side,
                  unsigned long long link, int join, void* stream, void* side) {
  Carve sv(saved);
  const act_t* c = sv.act((size_t)n * C);
  const act_t* l = sv.act((size_t)n * C);
  const float* mean = sv.f32(n);
  const float* rstd = sv.f32(n);
  float* dg = grads;
  float* db = dg + al4(C);
  float* dlw = db + al4(C);
  float* dlb = dlw + (size_t)C * C;
  float* dcw = dlw + al4((size_t)C * C + C);
  float* dcb = dcw + (size_t)C * 27 * C;
  Carve tp(tmp);
  act_t* dl = tp.act((size_t)n * C);
  act_t* dc = tp.act((size_t)n * C);
  act_t* dyr = tp.act((size_t)n * C);
  float* lnp = tp.f32(0);
  const size_t lnp_bytes = lotus_layernorm_bwd_workspace(n, C);
  void* sw = side ? side : stream;
  void* wws = side ? ws_side : ws_main;
  const size_t wws_bytes = side ? ws_side_bytes : ws_main_bytes;
  void* wcnt = side ? counters_side : counters_main;
  const bool big = n > 8192;
  if (side) {
    PRODUCE_THEN_FORK(lotus_layernorm_bwd(dy, l, mean, rstd, g, nullptr, dl, nullptr, nullptr, n, C, 0, nullptr, 0.f, 0, lnp, lnp_bytes, stream));
    CHECK(lotus_layernorm_bwd_params(lnp, n, C, dg, db, 0, side));
  } else {
    CHECK(lotus_layernorm_bwd(dy, l, mean, rstd, g, nullptr, dl, dg, db, n, C, 0, nullptr, 0.f, 0, lnp, lnp_bytes, stream));
  }
  CHECK(lotus_linear_wgrad(dl, c, dlw, dlb, n, C, C, 0, precision, wws, wws_bytes, wcnt, sw));
  PRODUCE_THEN_FORK(lotus_linear_dgrad(dl, lw, dc, nullptr, nullptr, n, C, C, LOTUS_ACT_NONE, 0.f, 0, precision, big ? nullptr : ws_main,
                                       big ? 0 : ws_main_bytes, big ? nullptr : counters_main, stream));
  CHECK(lotus_subm_conv_wgrad(dc, xs, dcw, dcb, nbr27, n, 27, C, C, 0, noshadow(precision), wws, wws_bytes, sw));
  const act_t* dsrc = dc;
  if (n_dup != 0) {
    CHECK(lotus_conv_dup_fold(dc, code0, order0, n, C, dyr, stream));
    dsrc = dyr;
  }
  CHECK(lotus_subm_conv(1, dsrc, cw, cw_packed, nullptr, add_dy ? dy : nullptr, dx_conv, nbr27, order0, n, 27, C, C, noshadow(precision), tap_plan, ws_conv,
                        ws_conv_bytes, stream));
  if (n_dup != 0) CHECK(lotus_conv_dup_mask(dx_conv, add_dy ? dy : nullptr, nbr27 + (size_t)13 * n, n, C, stream));
  if (side && join) CHECK(lotus_streamlink_wait(link, side, stream));
  return LOTUS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// One (Block, CABlock) pair per call (round 4): the five sub-blocks above chained on the host side of the C-ABI —
//   x1 = cpe(x, xs);  x2 = selfattn(x1);  x3 = ffn(x2);  x4 = crossattn_kv(x3, kv);  y = ffn(x4)
// (model_ca.py:270-310: Block i then CABlock i of a stage) — with the backward-pass hand-overs of the pre-masked
// gradients (ops.Handoff) wired inside.  Exactly the launches of the five composite calls in the same order on the same
// streams, so results are bit-identical to issuing them one by one (tests/test_gpu_round4.py); what it saves is host time:
// one Python -> C transition, one autograd node and one set of allocations per direction instead of five.
// Arguments travel as three host arrays (indices below): device pointers P, integers I, floating-point scalars F.
enum PairPtr {
  PP_X, PP_XS, PP_KV, PP_Y, PP_ACTS, PP_SAVED, PP_CW, PP_CWP, PP_CB, PP_LW, PP_LB, PP_G0, PP_B0,            // cpe
  PP_G1, PP_B1, PP_WQKV, PP_BQKV, PP_QNW, PP_QNB, PP_KNW, PP_KNB, PP_WP, PP_BP,                             // self-attention
  PP_G2, PP_B2, PP_W1, PP_B1F, PP_W2, PP_B2F,                                                               // mlp of the Block
  PP_G3, PP_B3, PP_WQ, PP_BQ, PP_CQNW, PP_CQNB, PP_CKNW, PP_CKNB, PP_CWP2, PP_CBP2,                         // cross-attention
  PP_G4, PP_B4, PP_W3, PP_B3F, PP_W4, PP_B4F,                                                               // mlp of the CABlock
  PP_NBR27, PP_ORDER0, PP_TAPPLAN, PP_CODE0, PP_GIDX, PP_OWNER, PP_STILES, PP_SBLOCKS, PP_KEXT, PP_EXTPOS, PP_CATILES, PP_CABLOCKS,
  PP_WS_MAIN, PP_WS_SIDE, PP_WS_CONV, PP_CNT_MAIN, PP_CNT_SIDE, PP_STREAM, PP_SIDE,
  PP_DY, PP_DX, PP_DXS, PP_DKV, PP_GRADS, PP_TMP,                                                           // backward only
  PP_COUNT
};
enum PairInt {
  PI_M, PI_C, PI_H, PI_HD, PI_NPAD, PI_NSTILES, PI_NEXTRA, PI_L, PI_NCATILES, PI_NCABLOCKS, PI_G, PI_KMAX, PI_NDUP, PI_SAME,
  PI_PREC, PI_KV_LD, PI_DKV_LD, PI_WS_MAIN, PI_WS_SIDE, PI_WS_CONV, PI_LINK, PI_SEED_SELF, PI_SEED_FFN1, PI_SEED_CROSS, PI_SEED_FFN2,
  PI_COUNT
};
enum PairFlt { PF_DROP, PF_ATTN, PF_SCALE, PF_COUNT };

static inline unsigned long long pair_mix(unsigned long long seed, unsigned long long k) {  // == ops.mix_seed (splitmix64 finaliser)
  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (k + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// Every sub-block's region of the flat saved / tmp buffers starts on a 256-byte boundary: the regions end with per-row
// statistics (M floats), and a following activation slab that is only 16-byte aligned makes every 128-byte row piece the
// attention kernels fetch straddle two cache lines (measured: -1.4 % on the whole step before this rounding).
static inline size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }
size_t lotus_pair_acts_floats(int M, int C) { return 4 * al64(actf((size_t)M * C)); }
size_t lotus_pair_saved_floats(int M, int C, int H, int Hd, int npad) {
  return al64(lotus_cpe_saved_floats(M, C)) + al64(lotus_selfattn_saved_floats(M, C, H, npad)) + 2 * al64(lotus_ffn_saved_floats(M, C, Hd)) +
         al64(lotus_crossattn_kv_saved_floats(M, C, H));
}
size_t lotus_pair_grads_floats(int C, int H, int Hd) {
  return lotus_cpe_grads_floats(C) + lotus_selfattn_grads_floats(C, H) + 2 * lotus_ffn_grads_floats(C, Hd) + lotus_crossattn_kv_grads_floats(C, H);
}
size_t lotus_pair_tmp_floats(int M, int C, int Hd, int n_extra, int L, int G) {
  return al64(lotus_cpe_tmp_floats(M, C)) + al64(lotus_selfattn_tmp_floats(M, C, n_extra)) + 2 * al64(lotus_ffn_tmp_floats(M, C, Hd)) +
         al64(lotus_crossattn_kv_tmp_floats(M, C, L, G)) + 4 * al64(actf((size_t)M * C));
}
static inline size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
size_t lotus_pair_ws_main_bytes(int M, int C, int H, int Hd, int nblocks_self, int nblocks_ca) {
  return max3(max3(lotus_cpe_ws_main_bytes(M, C), lotus_selfattn_ws_main_bytes(M, C, H, nblocks_self), lotus_ffn_ws_main_bytes(M, C, Hd)),
              lotus_crossattn_kv_ws_main_bytes(M, C, H, nblocks_ca), 0);
}
size_t lotus_pair_ws_side_bytes(int M, int C, int Hd) {
  return max3(max3(lotus_cpe_ws_side_bytes(M, C), lotus_selfattn_ws_side_bytes(M, C), lotus_ffn_ws_side_bytes(M, C, Hd)),
              lotus_crossattn_kv_ws_side_bytes(M, C), 0);
}
size_t lotus_pair_ws_conv_bytes(int M, int C) { return lotus_cpe_ws_conv_bytes(M, C); }
int lotus_pair_nptr(void) { return PP_COUNT; }
int lotus_pair_nint(void) { return PI_COUNT; }

#define PPTR(T, i) ((T)P[i])
int lotus_pair_fwd(const void* const* P, const long long* I, const double* F) {
  const int M = (int)I[PI_M], C = (int)I[PI_C], H = (int)I[PI_H], Hd = (int)I[PI_HD], npad = (int)I[PI_NPAD];
  const int prec = (int)I[PI_PREC];
  const float drop = (float)F[PF_DROP], attn_p = (float)F[PF_ATTN], scale = (float)F[PF_SCALE];
  void* ws = PPTR(void*, PP_WS_MAIN);
  const size_t ws_b = (size_t)I[PI_WS_MAIN];
  void* cnt = PPTR(void*, PP_CNT_MAIN);
  void* st = PPTR(void*, PP_STREAM);
  const size_t a = al64(actf((size_t)M * C));
  float* acts = PPTR(float*, PP_ACTS);
  act_t* x1 = (act_t*)acts;
  act_t* x2 = (act_t*)(acts + a);
  act_t* x3 = (act_t*)(acts + 2 * a);
  act_t* x4 = (act_t*)(acts + 3 * a);
  float* sv_cpe = PPTR(float*, PP_SAVED);
  float* sv_self = sv_cpe + al64(lotus_cpe_saved_floats(M, C));
  float* sv_ffn1 = sv_self + al64(lotus_selfattn_saved_floats(M, C, H, npad));
  float* sv_cross = sv_ffn1 + al64(lotus_ffn_saved_floats(M, C, Hd));
  float* sv_ffn2 = sv_cross + al64(lotus_crossattn_kv_saved_floats(M, C, H));
  const unsigned long long s_self = (unsigned long long)I[PI_SEED_SELF], s_f1 = (unsigned long long)I[PI_SEED_FFN1];
  const unsigned long long s_cross = (unsigned long long)I[PI_SEED_CROSS], s_f2 = (unsigned long long)I[PI_SEED_FFN2];
  CHECK(lotus_cpe_fwd(PPTR(const act_t*, PP_X), PPTR(const act_t*, PP_XS), PPTR(const float*, PP_CW), PPTR(const float*, PP_CWP),
                      PPTR(const float*, PP_CB), PPTR(const float*, PP_LW), PPTR(const float*, PP_LB), PPTR(const float*, PP_G0),
                      PPTR(const float*, PP_B0), x1, sv_cpe, PPTR(const int*, PP_NBR27), PPTR(const int*, PP_ORDER0), PPTR(const int*, PP_TAPPLAN), M, C, prec, ws, ws_b,
                      PPTR(void*, PP_WS_CONV), (size_t)I[PI_WS_CONV], cnt, st));
  CHECK(lotus_selfattn_fwd(x1, PPTR(const float*, PP_G1), PPTR(const float*, PP_B1), PPTR(const float*, PP_WQKV), PPTR(const float*, PP_BQKV),
                           PPTR(const float*, PP_QNW), PPTR(const float*, PP_QNB), PPTR(const float*, PP_KNW), PPTR(const float*, PP_KNB),
                           PPTR(const float*, PP_WP), PPTR(const float*, PP_BP), x2, sv_self, PPTR(const int*, PP_GIDX),
                           PPTR(const int*, PP_OWNER), PPTR(const int*, PP_STILES), (int)I[PI_NSTILES], npad, M, C, H, scale, drop, s_self,
                           attn_p, pair_mix(s_self, 1), prec, ws, ws_b, cnt, st));
  CHECK(lotus_ffn_fwd(x2, PPTR(const float*, PP_G2), PPTR(const float*, PP_B2), PPTR(const float*, PP_W1), PPTR(const float*, PP_B1F),
                      PPTR(const float*, PP_W2), PPTR(const float*, PP_B2F), x3, sv_ffn1, M, C, Hd, drop, s_f1, pair_mix(s_f1, 1), prec, ws,
                      ws_b, cnt, st));
  CHECK(lotus_crossattn_kv_fwd(x3, PPTR(const act_t*, PP_KV), (long)I[PI_KV_LD], PPTR(const float*, PP_G3), PPTR(const float*, PP_B3),
                               PPTR(const float*, PP_WQ), PPTR(const float*, PP_BQ), PPTR(const float*, PP_CQNW), PPTR(const float*, PP_CQNB),
                               PPTR(const float*, PP_CKNW), PPTR(const float*, PP_CKNB), PPTR(const float*, PP_CWP2),
                               PPTR(const float*, PP_CBP2), x4, sv_cross, PPTR(const int*, PP_CATILES), (int)I[PI_NCATILES], M, C, H, scale,
                               drop, s_cross, attn_p, pair_mix(s_cross, 1), prec, (int)I[PI_KMAX], ws, ws_b, cnt, st));
  return lotus_ffn_fwd(x4, PPTR(const float*, PP_G4), PPTR(const float*, PP_B4), PPTR(const float*, PP_W3), PPTR(const float*, PP_B3F),
                       PPTR(const float*, PP_W4), PPTR(const float*, PP_B4F), PPTR(act_t*, PP_Y), sv_ffn2, M, C, Hd, drop, s_f2,
                       pair_mix(s_f2, 1), prec, ws, ws_b, cnt, st);
}

// grads: [cpe | self | ffn1 | cross | ffn2] in the layouts of the five composites; dxs = input gradient of the convolution
// (added into dx when xs is x: PI_SAME).
int lotus_pair_bwd(const void* const* P, const long long* I, const double* F) {
  const int M = (int)I[PI_M], C = (int)I[PI_C], H = (int)I[PI_H], Hd = (int)I[PI_HD], npad = (int)I[PI_NPAD];
  const int n_extra = (int)I[PI_NEXTRA], L = (int)I[PI_L], G = (int)I[PI_G], prec = (int)I[PI_PREC];
  const float drop = (float)F[PF_DROP], attn_p = (float)F[PF_ATTN], scale = (float)F[PF_SCALE];
  void* wm = PPTR(void*, PP_WS_MAIN);
  void* wsd = PPTR(void*, PP_WS_SIDE);
  const size_t wm_b = (size_t)I[PI_WS_MAIN], wsd_b = (size_t)I[PI_WS_SIDE];
  void* cm = PPTR(void*, PP_CNT_MAIN);
  void* cs = PPTR(void*, PP_CNT_SIDE);
  void* st = PPTR(void*, PP_STREAM);
  void* side = PPTR(void*, PP_SIDE);
  const unsigned long long link = (unsigned long long)I[PI_LINK];
  const size_t a = al64(actf((size_t)M * C));
  float* acts = PPTR(float*, PP_ACTS);
  const act_t* x1 = (const act_t*)acts;
  const act_t* x2 = (const act_t*)(acts + a);
  const act_t* x3 = (const act_t*)(acts + 2 * a);
  const act_t* x4 = (const act_t*)(acts + 3 * a);
  const float* sv_cpe = PPTR(const float*, PP_SAVED);
  const float* sv_self = sv_cpe + al64(lotus_cpe_saved_floats(M, C));
  const float* sv_ffn1 = sv_self + al64(lotus_selfattn_saved_floats(M, C, H, npad));
  const float* sv_cross = sv_ffn1 + al64(lotus_ffn_saved_floats(M, C, Hd));
  const float* sv_ffn2 = sv_cross + al64(lotus_crossattn_kv_saved_floats(M, C, H));
  float* g_cpe = PPTR(float*, PP_GRADS);
  float* g_self = g_cpe + lotus_cpe_grads_floats(C);
  float* g_ffn1 = g_self + lotus_selfattn_grads_floats(C, H);
  float* g_cross = g_ffn1 + lotus_ffn_grads_floats(C, Hd);
  float* g_ffn2 = g_cross + lotus_crossattn_kv_grads_floats(C, H);
  float* t_cpe = PPTR(float*, PP_TMP);
  float* t_self = t_cpe + al64(lotus_cpe_tmp_floats(M, C));
  float* t_ffn1 = t_self + al64(lotus_selfattn_tmp_floats(M, C, n_extra));
  float* t_cross = t_ffn1 + al64(lotus_ffn_tmp_floats(M, C, Hd));
  float* t_ffn2 = t_cross + al64(lotus_crossattn_kv_tmp_floats(M, C, L, G));
  float* t_rest = t_ffn2 + al64(lotus_ffn_tmp_floats(M, C, Hd));
  act_t* d4 = (act_t*)t_rest;            // d x4, d x3, d x2 are temporaries; d x1 = the gradient the cpe receives
  act_t* d3 = (act_t*)(t_rest + a);
  act_t* d2 = (act_t*)(t_rest + 2 * a);
  // the pre-masked gradients handed from a sub-block's LayerNorm backward to its predecessor live in the predecessor's tmp
  // head (its `dz` slot is unused then): cross <- ffn2, ffn1 <- cross, self <- ffn1
  act_t* dz_cross = (act_t*)t_cross;     // first slice of the cross-attention tmp (dzb)
  act_t* dz_ffn1 = (act_t*)t_ffn1;       // first slice of the mlp tmp (dz2)
  act_t* dz_self = (act_t*)t_self;       // first slice of the self-attention tmp (dzb)
  const unsigned long long s_self = (unsigned long long)I[PI_SEED_SELF], s_f1 = (unsigned long long)I[PI_SEED_FFN1];
  const unsigned long long s_cross = (unsigned long long)I[PI_SEED_CROSS], s_f2 = (unsigned long long)I[PI_SEED_FFN2];
  const bool hand = drop > 0.f;
  // d x1 (what the cpe backward receives).  Encoder (xs is x): a temporary, the cpe backward writes conv-gradient + d x1
  // to PP_DX.  Decoder (xs = the stale skip branch): d x1 IS the gradient of x (the residual passes it through) and the
  // convolution's input gradient goes to PP_DXS.
  const int same = (int)I[PI_SAME];
  act_t* d1 = same ? (act_t*)(t_rest + 3 * a) : PPTR(act_t*, PP_DX);
  // mlp of the CABlock: dz_out masks d x4 with the cross-attention's projection dropout (drop, s_cross)
  CHECK(lotus_ffn_bwd(PPTR(const act_t*, PP_DY), nullptr, x4, PPTR(const float*, PP_G4), PPTR(const float*, PP_W3), PPTR(const float*, PP_W4),
                      sv_ffn2, d4, hand ? dz_cross : nullptr, hand ? drop : 0.f, s_cross, g_ffn2, t_ffn2, M, C, Hd, drop, s_f2, pair_mix(s_f2, 1),
                      prec, wm, wm_b, wsd, wsd_b, cm, cs, link, 0, st, side));
  // cross-attention: dz_out masks d x3 with the fc2 dropout of the Block's mlp (drop, mix(s_f1, 1))
  CHECK(lotus_crossattn_kv_bwd(d4, hand ? dz_cross : nullptr, x3, PPTR(const act_t*, PP_KV), (long)I[PI_KV_LD], PPTR(const float*, PP_G3),
                               PPTR(const float*, PP_WQ), PPTR(const float*, PP_CQNW), PPTR(const float*, PP_CQNB), PPTR(const float*, PP_CKNW),
                               PPTR(const float*, PP_CKNB), PPTR(const float*, PP_CWP2), sv_cross, d3, PPTR(act_t*, PP_DKV), (long)I[PI_DKV_LD],
                               hand ? dz_ffn1 : nullptr, hand ? drop : 0.f, pair_mix(s_f1, 1), g_cross, t_cross, PPTR(const int*, PP_CATILES),
                               PPTR(const int*, PP_CABLOCKS), (int)I[PI_NCABLOCKS], G, M, C, H, L, scale, drop, s_cross, attn_p,
                               pair_mix(s_cross, 1), prec, (int)I[PI_KMAX], wm, wm_b, wsd, wsd_b, cm, cs, link, 0, st, side));
  // mlp of the Block: dz_out masks d x2 with the self-attention's projection dropout (drop, s_self)
  CHECK(lotus_ffn_bwd(d3, hand ? dz_ffn1 : nullptr, x2, PPTR(const float*, PP_G2), PPTR(const float*, PP_W1), PPTR(const float*, PP_W2), sv_ffn1,
                      d2, hand ? dz_self : nullptr, hand ? drop : 0.f, s_self, g_ffn1, t_ffn1, M, C, Hd, drop, s_f1, pair_mix(s_f1, 1), prec, wm,
                      wm_b, wsd, wsd_b, cm, cs, link, 0, st, side));
  CHECK(lotus_selfattn_bwd(d2, hand ? dz_self : nullptr, x1, PPTR(const float*, PP_G1), PPTR(const float*, PP_WQKV), PPTR(const float*, PP_QNW),
                           PPTR(const float*, PP_QNB), PPTR(const float*, PP_KNW), PPTR(const float*, PP_KNB), PPTR(const float*, PP_WP), sv_self,
                           d1, g_self, t_self, PPTR(const int*, PP_GIDX), PPTR(const int*, PP_OWNER), PPTR(const int*, PP_STILES),
                           PPTR(const int*, PP_SBLOCKS), (int)I[PI_NSTILES], PPTR(const int*, PP_KEXT), PPTR(const int*, PP_EXTPOS), n_extra, npad,
                           M, C, H, scale, drop, s_self, attn_p, pair_mix(s_self, 1), prec, wm, wm_b, wsd, wsd_b, cm, cs, link, 0, st, side));
  // cpe: d x = d x1 (+ the convolution's input gradient when xs is x); separate xs -> its gradient goes to dxs
  return lotus_cpe_bwd(d1, PPTR(const act_t*, PP_XS), PPTR(const float*, PP_CW), PPTR(const float*, PP_CWP), PPTR(const float*, PP_LW),
                       PPTR(const float*, PP_G0), sv_cpe, same ? PPTR(act_t*, PP_DX) : PPTR(act_t*, PP_DXS), same, g_cpe, t_cpe,
                       PPTR(const int*, PP_NBR27), PPTR(const int*, PP_ORDER0), PPTR(const int*, PP_TAPPLAN), PPTR(const long long*, PP_CODE0), (int)I[PI_NDUP], M, C, prec, wm,
                       wm_b, PPTR(void*, PP_WS_CONV), (size_t)I[PI_WS_CONV], wsd, wsd_b, cm, cs, link, 0, st, side);
}
#undef PPTR

}  // extern "C"
