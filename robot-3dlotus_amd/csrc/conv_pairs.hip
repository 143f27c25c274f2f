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
#include <stdlib.h>
#include <type_traits>

namespace LOTUS_NS {

struct ConvP2 {
  const act_t* x;    // [n][KD]
  const float* w;    // packed B fragments: element (k, tap, j) at w[((tap * (KD / 4) + k / 4) * ND + j) * 4 + k % 4]
  act_t* y;          // [n][ND]
  float* ypart;      // fp32 partial slabs [nz][n][ND] when tap-split (part_stride != 0)
  const float* bias;
  const act_t* add;
  const int* nbr;    // [T][n]
  const int* rowidx; // [n] or null
  int n, T, KD, ND, mirror;
  int tpz;           // taps per blockIdx.z
  long part_stride;  // floats between z slabs (0 = single slab, epilogue applies bias/add)
};

// Kernel structure (v2, "resident rows"): the distinct neighbour rows touched by a 128-row tile
// (its own rows plus a halo, ~2-3x the tile) are hashed into an LDS table whose slot index IS the
// row's position in an LDS image xs[slot][KC+1].  Per reduction chunk the image is filled ONCE
// from HBM; every tap then reads its MFMA A fragments straight from the image through the
// per-pair slot index — no per-tap gather, no per-tap barrier.  (v1 re-gathered rows per
// (tap, chunk) and spent ~80 % of its time waiting for those loads.)  If a pathological tile
// overflows the table, its tap range is halved and processed in phases.
template <int NCS, int PREC = 0>
struct PairsCfg {
  // BM = 64 keeps the block at ~74 KB of LDS so that TWO blocks share a CU (8 waves): the hashing, image
  // fills and fold latencies of one block hide under the MFMAs of the other (BM = 128 / one block per CU
  // measured 43 % MFMA utilisation inside the tap loop)
  static constexpr int BM = 64, NRT = 4 / NCS, KC = NCS == 4 ? 32 : 16, NW = 32 * NCS, MAXT = 27;
  static constexpr int XW = KC;  // 32-bit words of one image row
  static constexpr int HT = 256;  // hash slots (= resident rows) per row tile
  // image row stride: 16-byte rows read with ds_read_b128 where the LDS budget allows (NCS = 2 sits exactly at
  // two blocks per CU with the odd stride)
  static constexpr bool AVEC = NCS == 4;
  static constexpr int XLD = AVEC ? XW + 4 : XW + 1;
  // output tile row stride: +4 floats so that the 16-byte fold accesses of lanes on consecutive rows land on
  // distinct 16-byte LDS slots
  static constexpr int OLD = NW + 4;
  static constexpr size_t bytes() {
    return (size_t)(NRT * (BM + 1) * OLD + NRT * HT * XLD) * 4 + (size_t)NRT * HT * 4 + (size_t)NRT * MAXT * BM +
           (size_t)NRT * MAXT * BM + (size_t)(NRT * MAXT + NRT * 64 + 16 + NRT * BM) * 4;
  }
};

// bf16 helpers (PREC 1: bf16 operands; PREC 3: bf16x3 split, see gemm.hip)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 cbf16x8;
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 cbf16x2;
__device__ __forceinline__ unsigned cpack_bf16(float a, float b) {
  const cbf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void csplit_bf16(float a, float b, unsigned& hi, unsigned& lo) {
  hi = cpack_bf16(a, b);
  lo = cpack_bf16(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}

// PREC != 0: the row image keeps, per row and chunk, KC/2 words of bf16 hi pairs followed by KC/2 words of lo pairs
// (same XLD as the fp32 image; PREC 1 fills the hi run only), the packed weights hold 8 words per (tap,
// 16-k block, k half, column): 4 hi + 4 lo (PREC 1: the 4 hi words only);
// a step issues KC/16 x (1 | 3) v_mfma_f32_32x32x16_bf16 instead of KC/2 fp32 MFMAs.
template <int NCS, int PREC>
__global__ __launch_bounds__(256, 2) void conv_pairs_kernel(ConvP2 p) {
  using Cfg = PairsCfg<NCS, PREC>;
  constexpr int BM = Cfg::BM, NRT = Cfg::NRT, KC = Cfg::KC, NW = Cfg::NW, MAXT = Cfg::MAXT, HT = Cfg::HT, XLD = Cfg::XLD;
  constexpr int OLD = Cfg::OLD;
  constexpr int TG = BM / 32;   // max pair groups per tap
  constexpr int GT = 64 * NCS;  // threads of one row tile's team
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* out_s = smem;                                   // [NRT][BM+1][OLD]  (row BM = dummy sink of padded pairs)
  float* xs_s = out_s + NRT * (BM + 1) * OLD;            // [NRT][HT][XLD]
  int* hkey_s = (int*)(xs_s + NRT * HT * XLD);      // [NRT][HT] global row id or -1
  int* cnt_s = hkey_s + NRT * HT;                        // [NRT][MAXT]
  int* grp_s = cnt_s + NRT * MAXT;                       // [NRT][64] pair groups: tap | offset << 8 | pairs << 16
  int* misc_s = grp_s + NRT * 64;                        // [16]: groups per rt (0..3), overflow (4), dummy row ids (8..15)
  int* prow_s = misc_s + 16;                             // [NRT][BM]
  unsigned char* slot_s = (unsigned char*)(prow_s + NRT * BM);    // [NRT][MAXT][BM] image slot of every pair (HT <= 256)
  unsigned char* row_s = (unsigned char*)(slot_s + NRT * MAXT * BM);  // [NRT][MAXT][BM]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = tid & 31, hh = (tid >> 5) & 1;
  const int rt = wave / NCS, cs = wave % NCS;
  const int gt = tid - rt * GT;
  // XCD-aware tile order: hardware block ids go to the 8 XCDs round-robin, each XCD has its own L2.  Consecutive row
  // tiles (consecutive along the space-filling curve) share most of their halo rows, so give every XCD a CONTIGUOUS run
  // of the (row tile, column block) list of this tap group instead of every 8th tile.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, total = gx * (int)gridDim.y;
    const int lin = by * gx + bx, xcd = lin & 7, slot = lin >> 3;
    const int q = total >> 3, r = total & 7;
    const int t = xcd * q + min(xcd, r) + slot;
    if (gx == 1) {
      by = t;
    } else {
      // several column blocks (C >= 256, the deep levels): a block's weight slices (taps x C x 128 columns) outweigh its
      // rows, so the run of an XCD walks the row tiles of ONE column block before moving to the next
      const int gy = gridDim.y;
      bx = t / gy;
      by = t - bx * gy;
    }
  }
  const int n0 = bx * NW;
  const int m0 = (by * NRT + rt) * BM;
  const int z_beg = blockIdx.z * p.tpz, z_end = min(p.T, z_beg + p.tpz);
  float* my_xs = xs_s + rt * HT * XLD;
  int* my_hkey = hkey_s + rt * HT;

  static_assert(HT * XLD >= MAXT * BM, "row image must be able to hold the temporary neighbour list");
  int* src_tmp = (int*)my_xs;  // [taps][BM] neighbour ids, aliased onto the row image (first phase only)
  for (int i = tid; i < NRT * (BM + 1) * OLD; i += 256) out_s[i] = 0.f;
  if (tid < 8) misc_s[8 + tid] = BM * 0x01010101;  // 32 row ids that all name the dummy sink row
  for (int r = gt; r < BM; r += GT) {
    const int m = m0 + r;
    prow_s[rt * BM + r] = m < p.n ? (p.rowidx ? p.rowidx[m] : m) : -1;
  }
  __syncthreads();
  // neighbour ids of the whole tile with independent (batched) loads -> LDS, then an ordered in-place
  // compaction of the active output rows of every tap (wave ballot + popcount)
  {
    // every load of the team is issued before the first result is stored: written as load -> store per iteration the
    // loop runs its MAXT * BM / GT (7 or 14) global loads one latency after the other at the head of every block
    const int total = (z_end - z_beg) * BM;
    constexpr int NL = (MAXT * BM + GT - 1) / GT;
    int vals[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = gt + j * GT;
      vals[j] = -1;
      if (i < total) {
        const int pr = prow_s[rt * BM + (i % BM)];
        if (pr >= 0) vals[j] = p.nbr[(long)(z_beg + i / BM) * p.n + pr];
      }
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = gt + j * GT;
      if (i < total) src_tmp[i] = vals[j];
    }
  }
  __syncthreads();
  for (int t = z_beg + cs; t < z_end; t += NCS) {
    int base = 0;
    for (int r0 = 0; r0 < BM; r0 += 64) {
      const int r = r0 + lane;
      const int nb = src_tmp[(t - z_beg) * BM + r];
      const unsigned long long b = __ballot(nb >= 0);
      if (nb >= 0) {
        const int k = base + __popcll(b & ((1ull << lane) - 1ull));
        row_s[(rt * MAXT + t) * BM + k] = (unsigned char)r;
        src_tmp[(t - z_beg) * BM + k] = nb;  // k <= r: in place; valid until the first image fill
      }
      base += __popcll(b);
    }
    if (lane == 0) cnt_s[rt * MAXT + t] = base;
    // pad the last group with the dummy sink row so that the fold needs no per-element predicate
    if (base + lane < ((base + 31) & ~31)) row_s[(rt * MAXT + t) * BM + base + lane] = (unsigned char)BM;
  }
  __syncthreads();

  const int nkc = p.KD / KC;
  // operand fragment registers: fp32 keeps one float per k; the bf16 paths keep whole 4-word MFMA operands (hi, lo per
  // 16-k block) so that no operand has to be re-assembled from scattered registers
  using Frag = typename std::conditional<PREC != 0, float4, float>::type;
  constexpr int NF = PREC ? KC / 8 : KC / 2;
  Frag bst[3][NF];
  static_assert(2 * MAXT <= 64, "the group list must fit the lanes of one VGPR");

  auto load_b = [&](Frag (&dst)[NF], int t, int kc) {  // weight fragment of (tap, chunk): this lane's KC/2 consecutive k
    // k-quads (4 consecutive k) are the 16-byte unit: quad (kc * KC + hh * KC/2) / 4 + q of column j sits at float4
    // index quad * ND + j, so a wave instruction reads two contiguous 512-byte runs; 32-bit index from the uniform
    // base -> scalar-base global loads with one VALU op of address math each
    const int tw = p.mirror ? (p.T - 1 - t) : t;
    const float4* wp4 = reinterpret_cast<const float4*>(p.w);
    if constexpr (PREC != 0) {  // per 16-k block: [hi x4 | lo x4] words of (k half hh, column j) -> float4 index 2 * (...)
#pragma unroll
      for (int q2 = 0; q2 < KC / 16; ++q2) {
        const unsigned tup = (unsigned)(((tw * (p.KD / 16) + kc * (KC / 16) + q2) * 2 + hh) * p.ND + n0 + cs * 32 + l31);
        const unsigned base = PREC == 3 ? tup * 2 : tup;  // PREC 1: hi words only (packed without the lo halves)
        dst[2 * q2] = wp4[base];
        if (PREC == 3) dst[2 * q2 + 1] = wp4[base + 1];
      }
    } else {
      const unsigned base = (unsigned)((tw * (p.KD / 4) + (kc * KC + hh * (KC / 2)) / 4) * p.ND + n0 + cs * 32 + l31);
#pragma unroll
      for (int q = 0; q < KC / 8; ++q) {
        const float4 v = wp4[base + (unsigned)(q * p.ND)];
        dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
      }
    }
  };

  int t0 = z_beg, span = z_end - z_beg;
  bool first_phase = true;
  while (t0 < z_end) {  // phases over tap ranges (one phase unless a tile overflows the row table)
    const int t1 = min(z_end, t0 + span);
    for (int i = gt; i < HT; i += GT) my_hkey[i] = -1;
    if (tid == 0) misc_s[4] = 0;
    __syncthreads();
    // hash every pair's neighbour row into the table; the table position is the LDS row slot.  One flat loop
    // over (tap, pair) so that the LDS round trips of different pairs overlap.
    for (int i = gt; i < (t1 - t0) * BM; i += GT) {
      const int t = t0 + i / BM, k = i % BM;
      if (k >= cnt_s[rt * MAXT + t]) continue;
      const int g = first_phase ? src_tmp[(t - z_beg) * BM + k]
                                : p.nbr[(long)t * p.n + prow_s[rt * BM + row_s[(rt * MAXT + t) * BM + k]]];
      unsigned pos = ((unsigned)g * 2654435761u >> 16) % HT;
      int probes = 0;
      while (true) {
        const int old = atomicCAS(&my_hkey[pos], -1, g);
        if (old == -1 || old == g) break;
        pos = pos + 1 == HT ? 0 : pos + 1;
        if (++probes >= HT) { misc_s[4] = 1; break; }
      }
      slot_s[(rt * MAXT + t) * BM + k] = (unsigned char)pos;
    }
    __syncthreads();
    if (misc_s[4]) {  // block-uniform: retry with half the taps (a single tap always fits: <= BM rows)
      __syncthreads();
      span = max(1, (t1 - t0) / 2);
      continue;
    }
    first_phase = false;  // the image fill below overwrites the temporary neighbour list
    if (gt == 0) {  // work list: dense groups of <= 32 pairs, taps in fixed order
      int c = 0;
      for (int t = t0; t < t1; ++t) {
        const int cnt = cnt_s[rt * MAXT + t];
        for (int o = 0; o < cnt; o += 32) grp_s[rt * 64 + c++] = t | (o << 8) | (min(32, cnt - o) << 16);
      }
      misc_s[rt] = c;
    }
    __syncthreads();
    const int ngr = __builtin_amdgcn_readfirstlane(misc_s[rt]);
    const int vgrp = grp_s[rt * 64 + min(lane, max(ngr - 1, 0))];  // lane j: j-th group (read as a scalar with v_readlane)

    // Resident image: chunk kc + 1 is fetched into registers while the groups of chunk kc run (one HBM/L2 read
    // per row and chunk, its latency under the MFMAs) and written to LDS between the two barriers ending the chunk.
    constexpr int FILL = HT * (KC / 4) / GT;
    static_assert(HT * (KC / 4) % GT == 0, "fill loop shape");
    float4 vv[FILL];
    auto issue_fill = [&](int kc) {
      int gs[FILL];
#pragma unroll
      for (int j = 0; j < FILL; ++j) gs[j] = my_hkey[(gt + j * GT) / (KC / 4)];
#pragma unroll
      for (int j = 0; j < FILL; ++j) {
        const int q = (gt + j * GT) % (KC / 4);
        vv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gs[j] >= 0) vv[j] = ld4(p.x + (long)gs[j] * p.KD + kc * KC + q * 4);
      }
    };
    auto store_fill = [&]() {
#pragma unroll
      for (int j = 0; j < FILL; ++j) {
        const int i = gt + j * GT;
        float* o = my_xs + (i / (KC / 4)) * XLD + (i % (KC / 4)) * 4;
        if (PREC) {  // words 2q, 2q+1 of the hi run (and of the lo run KC/2 words further) of this row
          unsigned* ow = reinterpret_cast<unsigned*>(my_xs) + (i / (KC / 4)) * XLD + (i % (KC / 4)) * 2;
          unsigned h0, l0, h1, l1;
          csplit_bf16(vv[j].x, vv[j].y, h0, l0);
          csplit_bf16(vv[j].z, vv[j].w, h1, l1);
          ow[0] = h0; ow[1] = h1;
          if (PREC == 3) { ow[KC / 2] = l0; ow[KC / 2 + 1] = l1; }
        } else if (Cfg::AVEC) {
          st4(o, vv[j]);
        } else {
          o[0] = vv[j].x; o[1] = vv[j].y; o[2] = vv[j].z; o[3] = vv[j].w;
        }
      }
    };
    issue_fill(0);
    store_fill();
    // fold base of this lane: its wave's 32-column slice, the 4-column run 4 * hh of every 8-column group
    float* ob = out_s + rt * (BM + 1) * OLD + cs * 32 + 4 * hh;
    constexpr int KS = KC / 2;  // MFMAs (k-steps) per group and chunk
    // image offset of this lane's A fragment run for group word gw: pair (offset + min(lane, pairs - 1)), k half hh
    auto a_base = [&](int gw) {
      const int t = gw & 255, off = (gw >> 8) & 255, cn = gw >> 16;
      return (int)slot_s[(rt * MAXT + t) * BM + off + min(l31, cn - 1)] * XLD + (PREC ? hh * 4 : hh * KS);
    };
    auto read_a = [&](Frag (&dst)[NF], int ab) {  // this lane's run of KS consecutive k of one image row
      if constexpr (PREC != 0) {  // per 16-k block q2: 4 hi words at 8 * q2 + 4 * hh, the lo words KC/2 further
#pragma unroll
        for (int q2 = 0; q2 < KC / 16; ++q2) {
          if (Cfg::AVEC) {
            dst[2 * q2] = ld4(my_xs + ab + q2 * 8);
            if (PREC == 3) dst[2 * q2 + 1] = ld4(my_xs + ab + KC / 2 + q2 * 8);
          } else {
            const float* r = my_xs + ab + q2 * 8;
            dst[2 * q2] = make_float4(r[0], r[1], r[2], r[3]);
            if (PREC == 3) dst[2 * q2 + 1] = make_float4(r[KC / 2], r[KC / 2 + 1], r[KC / 2 + 2], r[KC / 2 + 3]);
          }
        }
      } else if constexpr (Cfg::AVEC) {
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4) {
          const float4 v4 = ld4(my_xs + ab + s4 * 4);
          dst[s4 * 4] = v4.x; dst[s4 * 4 + 1] = v4.y; dst[s4 * 4 + 2] = v4.z; dst[s4 * 4 + 3] = v4.w;
        }
      } else {
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) dst[s2] = my_xs[ab + s2];
      }
    };
    // Flat (chunk, group) pipeline, one basic block per step, software-pipelined by hand (sched_barrier pins the
    // interleave) and continuous across chunk boundaries:
    //   weights  : fragments of step i + 2 are requested from L2 (three register stages)
    //   A        : fragments of step i + 1 are read from the image while step i multiplies (two stages; restarted
    //              after the image swap at a chunk boundary)
    //   fold     : the accumulators of step i - 1 are added into the LDS output tile under the MFMAs of step i
    //              (two accumulator sets); the wave owns its 32 columns and folds in fixed order
    // Group words live in the lanes of one VGPR and are read as scalars with v_readlane.
    const int nsteps = nkc * ngr;
    __syncthreads();
    if (ngr == 0) {  // nothing to multiply in this phase: keep the block's barrier sequence
      if (nkc > 1) issue_fill(1);
      for (int kc = 1; kc < nkc; ++kc) {
        __syncthreads();
        store_fill();
        if (kc + 1 < nkc) issue_fill(kc + 1);
        __syncthreads();
      }
    } else {
      f32x16 accs[2];
      Frag afr[2][NF];
      const unsigned char* prows = (const unsigned char*)(misc_s + 8);  // pending fold: none yet -> dummy sink
      int g_cur = 0, kc_cur = 0;  // position of the running step
      int g_b = 0, kc_b = 0;      // position of the weight prefetch (two steps ahead)
      auto adv = [&](int& g, int& kc) {
        if (++g == ngr) { g = 0; ++kc; }
      };
      load_b(bst[0], __builtin_amdgcn_readlane(vgrp, 0) & 255, 0);
      adv(g_b, kc_b);
      load_b(bst[1], __builtin_amdgcn_readlane(vgrp, g_b) & 255, min(kc_b, nkc - 1));
      adv(g_b, kc_b);
      if (nkc > 1) issue_fill(1);  // after the first weight loads: vmcnt retires in order
      read_a(afr[0], a_base(__builtin_amdgcn_readlane(vgrp, 0)));
      int abase_n = a_base(__builtin_amdgcn_readlane(vgrp, min(1, ngr - 1)));
      auto step = [&](const Frag (&bc)[NF], Frag (&bl)[NF], f32x16& ac, const f32x16& ap, Frag (&A)[NF], Frag (&An)[NF]) {
        if (g_cur == 0 && kc_cur > 0) {  // chunk boundary (block-uniform count: nkc - 1 per wave)
          __syncthreads();               // every wave is done with the old image
          store_fill();
          if (kc_cur + 1 < nkc) issue_fill(kc_cur + 1);
          __syncthreads();
          read_a(A, a_base(__builtin_amdgcn_readlane(vgrp, 0)));
        }
        const int gw = __builtin_amdgcn_readlane(vgrp, g_cur);
        const int gwb = __builtin_amdgcn_readlane(vgrp, g_b);
        load_b(bl, gwb & 255, min(kc_b, nkc - 1));
        const bool next_in_chunk = g_cur + 1 < ngr;
        const int gwn2 = __builtin_amdgcn_readlane(vgrp, g_cur + 2 < ngr ? g_cur + 2 : (g_cur + 2 - ngr) % ngr);
        // MFMA list of the step.  fp32: KS k-steps of 32x32x2.  bf16: per 16-k block (lo*hi, hi*lo,) hi*hi of
        // 32x32x16 (small terms first).  mm(i) issues entry i; the list is cut in four runs around the fold.
        constexpr int NM = PREC ? (KC / 16) * (PREC == 3 ? 3 : 1) : KS;
        constexpr int Q1 = PREC ? NM / 4 : KS / 8, Q2 = PREC ? NM / 2 : KS / 4, Q3 = PREC ? (3 * NM + 3) / 4 : 5 * KS / 8;
        auto mm = [&](int i) {
          if constexpr (PREC == 0) {
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[i], A[i], ac, 0, 0, 0);
          } else {
            const int q2 = PREC == 3 ? i / 3 : i, term = PREC == 3 ? i % 3 : 2;  // 0: w_lo x_hi, 1: w_hi x_lo, 2: w_hi x_hi
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8, bc[2 * q2 + (term == 0 ? 1 : 0)]),
                                                         __builtin_bit_cast(cbf16x8, A[2 * q2 + (term == 1 ? 1 : 0)]), ac, 0, 0, 0);
          }
        };
        // The product is computed TRANSPOSED (MFMA A operand = weight fragment, B operand = gathered rows):
        // C[channel][pair], so a lane owns ONE pair = one output row and its 16 registers are four runs of
        // four consecutive channels -> the fold is 4 x (ds_read_b128, 4 adds, ds_write_b128) through a single
        // row offset instead of 16 x (ds_read_b32, add, ds_write_b32) through 16 unpacked offsets.
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < Q1; ++s2) mm(s2);
        __builtin_amdgcn_sched_barrier(0);
        const int prid = prows[l31];  // output row of the pending group's pair l31 (dummy sink row for padding)
#pragma unroll
        for (int s2 = Q1; s2 < Q2; ++s2) mm(s2);
        __builtin_amdgcn_sched_barrier(0);
        float4* orow = reinterpret_cast<float4*>(ob + prid * OLD);
        float4 ov[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ov[q] = orow[2 * q];
        if (next_in_chunk) read_a(An, abase_n);
        const int ab2 = a_base(gwn2);
#pragma unroll
        for (int s2 = Q2; s2 < Q3; ++s2) mm(s2);
        __builtin_amdgcn_sched_barrier(0);
        // (LDS float atomics — ds_add_f32 — were measured 5x slower than read / add / write)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          orow[2 * q] = make_float4(ov[q].x + ap[4 * q], ov[q].y + ap[4 * q + 1], ov[q].z + ap[4 * q + 2], ov[q].w + ap[4 * q + 3]);
#pragma unroll
        for (int s2 = Q3; s2 < NM; ++s2) mm(s2);
        __builtin_amdgcn_sched_barrier(0);
        abase_n = ab2;
        prows = row_s + (rt * MAXT + (gw & 255)) * BM + ((gw >> 8) & 255);
        adv(g_cur, kc_cur);
        adv(g_b, kc_b);
      };
      for (int i = 0; i < nsteps; i += 6) {
        step(bst[0], bst[2], accs[0], accs[1], afr[0], afr[1]);
        if (i + 1 < nsteps) step(bst[1], bst[0], accs[1], accs[0], afr[1], afr[0]);
        if (i + 2 < nsteps) step(bst[2], bst[1], accs[0], accs[1], afr[0], afr[1]);
        if (i + 3 < nsteps) step(bst[0], bst[2], accs[1], accs[0], afr[1], afr[0]);
        if (i + 4 < nsteps) step(bst[1], bst[0], accs[0], accs[1], afr[0], afr[1]);
        if (i + 5 < nsteps) step(bst[2], bst[1], accs[1], accs[0], afr[1], afr[0]);
      }
      {  // fold of the last step (its accumulator set has the parity of nsteps - 1)
        float4* orow = reinterpret_cast<float4*>(ob + (int)prows[l31] * OLD);
        const bool odd = (nsteps - 1) & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = orow[2 * q];
          v.x += odd ? accs[1][4 * q] : accs[0][4 * q];
          v.y += odd ? accs[1][4 * q + 1] : accs[0][4 * q + 1];
          v.z += odd ? accs[1][4 * q + 2] : accs[0][4 * q + 2];
          v.w += odd ? accs[1][4 * q + 3] : accs[0][4 * q + 3];
          orow[2 * q] = v;
        }
      }
    }
    __syncthreads();
    t0 = t1;
  }

  // epilogue: coalesced row stores (final result, or this tap group's partial slab)
  const bool final_out = p.part_stride == 0;
  float* yo = p.ypart + (long)blockIdx.z * p.part_stride;
  for (int i = gt; i < BM * (NW / 4); i += GT) {
    const int r = i / (NW / 4), c4 = i % (NW / 4);
    const int pr = prow_s[rt * BM + r];
    if (pr < 0) continue;
    float4 v = ld4(out_s + (rt * (BM + 1) + r) * OLD + c4 * 4);
    const int col = n0 + c4 * 4;
    const long o = (long)pr * p.ND + col;
    if (final_out) {
      if (p.bias) {
        const float4 b = ld4(p.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (p.add) {
        const float4 a = ld4(p.add + o);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
    }
    if (final_out) st4(p.y + o, v);
    else st4(yo + o, v);
  }
}

// ------------------------------------------------------------------------------------ bf16 operands: output-stationary
// With v_mfma_f32_32x32x16_bf16 a product costs 1/16 of the fp32 form (3/16 as a bf16x3 split), and what bounds the
// pair-compacted kernel above is everything around the products: compaction, hashing rows into the LDS image, one LDS
// fold of the accumulators per 32-pair group (tools/conv_bench.py: bf16 launches only 1.5-2x faster than fp32 ones).  For
// the bf16 operand modes the zeros of an output-stationary formulation are cheaper than all of that:
//   * a wave owns 32 OUTPUT rows x NS*32 columns; its accumulators stay in registers over all taps and chunks (no fold,
//     no pair tables, no row image);
//   * per tap the lane of row r loads its neighbour row nbr[t][r] straight from global memory as MFMA B fragments — the
//     8 consecutive k of a bf16 MFMA operand are 16 contiguous bytes of a bf16 row (32 of an fp32 row, converted in
//     registers); an absent neighbour is a zero fragment (27-48 % of the fragments are real: 2-3.7x the products of
//     the compacted form, still several times below its cost);
//   * the packed weights of (tap, k chunk) — already in fragment order, conv_wpack_bf16_kernel — are staged through a
//     double-buffered LDS image shared by the block's four waves (128 rows per weight fetch), one barrier per stage;
//   * the product is transposed like above (weights = A operand): a lane owns one output row, the epilogue goes through
//     an LDS tile for coalesced row stores.
template <int PREC, int NS, int KCH>
struct OsCfg {
  // NB = row fragments (16 bytes per lane) of a stage: bf16 modes 8 k of a 16-k block, fp32 (PREC 0) 4 k of an 8-k block —
  // lane half h of a block takes k = 4 h + j for the four fp32 MFMAs j, which is the k-quad layout of the packed weights
  static constexpr int BR = 128, NW = 32 * NS, NB = PREC == 0 ? KCH / 8 : KCH / 16, MAXT = 27;
  static constexpr int U4T = PREC == 3 ? 2 : 1;                 // uint4 per packed (tap, 16-k block, k half, column) tuple
  static constexpr int STAGE_U4 = NB * 2 * NW * U4T;            // uint4 per weight stage (fp32: one float4 per k quad and column)
  static constexpr int OLD = NW + 4;                            // epilogue tile row stride (floats)
  // [2 weight stages | neighbour ids of the block's rows, all taps] during the tap loop; the epilogue tile reuses both
  static constexpr size_t loop_bytes() { return 2 * (size_t)STAGE_U4 * 16 + (size_t)MAXT * BR * sizeof(int); }
  static constexpr size_t tile_bytes() { return (size_t)BR * OLD * 4; }
  static constexpr size_t bytes() { return (loop_bytes() > tile_bytes() ? loop_bytes() : tile_bytes()) + BR * sizeof(int); }
};

template <int PREC, int NS, int KCH>
__global__ __launch_bounds__(256, 2) void conv_os_kernel(ConvP2 p) {
  static_assert(PREC == 0 || PREC == 1 || PREC == 3, "operand mode");
  static_assert(PREC != 0 || !LOTUS_ACT_IS_BF16, "the fp32 products read fp32 rows");
  using Cfg = OsCfg<PREC, NS, KCH>;
  constexpr int BR = Cfg::BR, NW = Cfg::NW, NB = Cfg::NB, U4T = Cfg::U4T, STAGE_U4 = Cfg::STAGE_U4, OLD = Cfg::OLD;
  constexpr int WPT = STAGE_U4 / 256;  // weight uint4 per thread and stage
  static_assert(STAGE_U4 % 256 == 0, "stage copy shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* wst = reinterpret_cast<uint4*>(smem);                                  // [2][STAGE_U4]
  int* nbr_s = reinterpret_cast<int*>(wst + 2 * STAGE_U4);                      // [taps of this block][BR]
  int* prow_s = reinterpret_cast<int*>(reinterpret_cast<char*>(smem) + (Cfg::bytes() - BR * sizeof(int)));

  const int tid = threadIdx.x, wave = tid >> 6, l31 = tid & 31, hh = (tid >> 5) & 1;
  // XCD-aware order as above: an XCD walks a contiguous run of row blocks of one column block
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy;
    const int lin = by * gx + bx, xcd = lin & 7, slot = lin >> 3;
    const int q = total >> 3, r = total & 7;
    const int t = xcd * q + min(xcd, r) + slot;
    bx = t / gy;
    by = t - bx * gy;
  }
  const int n0 = bx * NW, m0 = by * BR;
  const int z_beg = blockIdx.z * p.tpz, z_end = min(p.T, z_beg + p.tpz);
  const int nkc = p.KD / KCH, nst = (z_end - z_beg) * nkc;

  if (tid < BR) {
    const int m = m0 + tid;
    prow_s[tid] = m < p.n ? (p.rowidx ? p.rowidx[m] : m) : -1;
  }
  __syncthreads();
  {  // neighbour ids of the block's rows for all its taps, one batch of independent loads: inside the tap loop a row
     // fetch must not sit behind the id fetch it depends on (two memory latencies per stage otherwise)
    constexpr int NL = (Cfg::MAXT * BR + 255) / 256;
    const int total = (z_end - z_beg) * BR;
    int vals[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = tid + j * 256;
      vals[j] = -1;
      if (i < total) {
        const int prr = prow_s[i % BR];
        if (prr >= 0) vals[j] = p.nbr[(long)(z_beg + i / BR) * p.n + prr];
      }
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int i = tid + j * 256;
      if (i < total) nbr_s[i] = vals[j];
    }
  }
  const int lrow = wave * 32 + l31;  // this lane's output row within the block (both k halves of a row share it)

  f32x16 acc[NS];
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s2][r] = 0.f;

  // raw neighbour-row fragments of one stage: NB x 8 consecutive k of row q (this lane's k half).  bf16 storage: the 16
  // row bytes ARE the MFMA operand and two stages are kept in flight; fp32 storage: 32 bytes, converted in registers
  constexpr int RAWV = (LOTUS_ACT_IS_BF16 || PREC == 0) ? 1 : 2;  // 16-byte loads per fragment
  constexpr int KFR = PREC == 0 ? 4 : 8;                          // k per lane and fragment
  constexpr int DEPTH = LOTUS_ACT_IS_BF16 ? 2 : 1; // row stages in flight
  uint4 braw[DEPTH][NB][RAWV];
  auto load_rows = [&](uint4 (&dst)[NB][RAWV], int st) __attribute__((always_inline)) -> bool {  // rows of stage st (taps outer, chunks inner)
    const int tl = st / nkc, kc = st - tl * nkc;
    const int q = nbr_s[tl * BR + lrow];
    // every load is issued unconditionally (an absent neighbour reads row 0 and is zeroed by value selects): a branch
    // around each load makes the compiler wait for it before the next one — NB dependent round trips per stage
    const act_t* src = p.x + (long)max(q, 0) * p.KD + kc * KCH + hh * KFR;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int v = 0; v < RAWV; ++v) dst[kb][v] = reinterpret_cast<const uint4*>(src + kb * 2 * KFR)[v];
    return q >= 0;  // applied when the fragments are converted (selecting here would wait for the loads)
  };
  uint4 bhi[NB], blo[NB];
  auto convert_rows = [&](const uint4 (&src)[NB][RAWV], bool ok) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      if constexpr (LOTUS_ACT_IS_BF16 || PREC == 0) {  // the loaded bytes ARE the operand (bf16 rows / four fp32 k)
        bhi[kb] = make_uint4(ok ? src[kb][0].x : 0u, ok ? src[kb][0].y : 0u, ok ? src[kb][0].z : 0u, ok ? src[kb][0].w : 0u);
        blo[kb] = make_uint4(0u, 0u, 0u, 0u);
      } else {
        const float f[8] = {__builtin_bit_cast(float, src[kb][0].x), __builtin_bit_cast(float, src[kb][0].y),
                            __builtin_bit_cast(float, src[kb][0].z), __builtin_bit_cast(float, src[kb][0].w),
                            __builtin_bit_cast(float, src[kb][1].x), __builtin_bit_cast(float, src[kb][1].y),
                            __builtin_bit_cast(float, src[kb][1].z), __builtin_bit_cast(float, src[kb][1].w)};
        unsigned h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (PREC == 3) csplit_bf16(f[2 * e], f[2 * e + 1], h[e], l[e]);
          else { h[e] = cpack_bf16(f[2 * e], f[2 * e + 1]); l[e] = 0u; }
        }
        bhi[kb] = make_uint4(ok ? h[0] : 0u, ok ? h[1] : 0u, ok ? h[2] : 0u, ok ? h[3] : 0u);
        blo[kb] = make_uint4(ok ? l[0] : 0u, ok ? l[1] : 0u, ok ? l[2] : 0u, ok ? l[3] : 0u);
      }
    }
  };
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));  // native vectors: the array stays in registers
  u32x4_t wreg[WPT];
  const u32x4_t* wp4 = reinterpret_cast<const u32x4_t*>(p.w);
  auto load_w = [&](int st) __attribute__((always_inline)) {
    const int tl = st / nkc, kc = st - tl * nkc, t = z_beg + tl;
    const int tw = p.mirror ? (p.T - 1 - t) : t;
    constexpr int L = NW * U4T;  // contiguous uint4 of one (16-k block, k half) run of this column block
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int i = tid + j * 256, run = i / L, off = i - run * L;
      if constexpr (PREC == 0) {  // fp32 fragments: float4 (k quad, column) at (tap * KD/4 + quad) * ND + column
        wreg[j] = wp4[((long)tw * (p.KD / 4) + kc * (KCH / 4) + run) * p.ND + n0 + off];
      } else {
        const long tup = ((long)(tw * (p.KD / 16) + kc * NB + (run >> 1)) * 2 + (run & 1)) * p.ND + n0;
        wreg[j] = wp4[tup * U4T + off];
      }
    }
  };
  auto store_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < WPT; ++j) reinterpret_cast<u32x4_t*>(wst)[buf * STAGE_U4 + tid + j * 256] = wreg[j];
  };
  auto products = [&](int st) __attribute__((always_inline)) {
    const uint4* wb = wst + (st & 1) * STAGE_U4;
    const int tl = st / nkc;
    if (!__ballot(nbr_s[tl * BR + lrow] >= 0)) return;  // a tap none of the wave's rows has: nothing to add
    if constexpr (PREC == 0) {
      // exact fp32: four products (k = 8 kb + 4 hh + j on both operands) per 16-byte weight fragment; the fragment of
      // step i + 1 is requested before the MFMAs of step i (one LDS round trip per 256 MFMA cycles would otherwise be
      // exposed in front of every group of four)
      constexpr int N = NB * NS;
      uint4 a_cur = wb[(hh * NW) + l31];  // step 0: kb = 0, slice 0
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int kb = i / NS, s2 = i % NS;
        uint4 a_nxt = a_cur;
        if (i + 1 < N) a_nxt = wb[(((i + 1) / NS) * 2 + hh) * NW + ((i + 1) % NS) * 32 + l31];
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the read below the products it should fly under)
        acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a_cur.x), __builtin_bit_cast(float, bhi[kb].x), acc[s2], 0, 0, 0);
        acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a_cur.y), __builtin_bit_cast(float, bhi[kb].y), acc[s2], 0, 0, 0);
        acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a_cur.z), __builtin_bit_cast(float, bhi[kb].z), acc[s2], 0, 0, 0);
        acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a_cur.w), __builtin_bit_cast(float, bhi[kb].w), acc[s2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a_cur = a_nxt;
      }
      return;
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const uint4* a = wb + ((kb * 2 + hh) * NW + s2 * 32 + l31) * U4T;
        const uint4 ah = a[0];
        if constexpr (PREC == 3) {
          const uint4 al = a[1];
          acc[s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8, al), __builtin_bit_cast(cbf16x8, bhi[kb]), acc[s2], 0, 0, 0);
          acc[s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8, ah), __builtin_bit_cast(cbf16x8, blo[kb]), acc[s2], 0, 0, 0);
        }
        acc[s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8, ah), __builtin_bit_cast(cbf16x8, bhi[kb]), acc[s2], 0, 0, 0);
      }
    }
  };

  __syncthreads();  // nbr_s
  bool bok[DEPTH];
  load_w(0);
  bok[0] = load_rows(braw[0], 0);
  if (DEPTH == 2) bok[DEPTH - 1] = load_rows(braw[DEPTH - 1], min(1, nst - 1));
  store_w(0);
  convert_rows(braw[0], bok[0]);
  __syncthreads();
  // (every lambda of this kernel is always_inline: `stage` has several call sites, and an out-of-line closure keeps the
  // register arrays it captures — wreg, braw — in scratch memory: each weight load was waited for and spilled at once)
  // stage st: weights of st + 1 and rows of st + DEPTH go into registers, the products of st run from LDS / bhi, then the
  // rows of st + 1 become the operand registers and the weights of st + 1 the other LDS stage; one barrier per stage
  auto stage = [&](int st, uint4 (&fetch)[NB][RAWV], bool& fetch_ok, const uint4 (&next)[NB][RAWV], const bool& next_ok) __attribute__((always_inline)) {
    // every load of the stage is unconditional (the last stages re-fetch stage nst - 1): a load under a branch is a
    // "register or load" merge, for which the compiler waits right behind the load instead of where the value is used
    load_w(min(st + 1, nst - 1));
    fetch_ok = load_rows(fetch, min(st + DEPTH, nst - 1));
    products(st);
    store_w((st + 1) & 1);
    convert_rows(next, next_ok);
    __syncthreads();
  };
  if constexpr (DEPTH == 2) {
    for (int st = 0; st < nst; st += 2) {  // rows of st live in braw[0] (already converted), st + 1 in braw[1]
      stage(st, braw[0], bok[0], braw[1], bok[1]);
      if (st + 1 < nst) stage(st + 1, braw[1], bok[1], braw[0], bok[0]);
    }
  } else {
    for (int st = 0; st < nst; ++st) stage(st, braw[0], bok[0], braw[0], bok[0]);
  }

  // epilogue: accumulators (channel runs of one row per lane) -> LDS tile -> coalesced row stores
  float* out_s = smem;  // [BR][OLD], reuses the weight stages (every wave passed the loop's last barrier)
  {
    float* orow = out_s + (wave * 32 + l31) * OLD + 4 * hh;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        *reinterpret_cast<float4*>(orow + s2 * 32 + 8 * q4) =
            make_float4(acc[s2][4 * q4], acc[s2][4 * q4 + 1], acc[s2][4 * q4 + 2], acc[s2][4 * q4 + 3]);
  }
  __syncthreads();
  const bool final_out = p.part_stride == 0;
  float* yo = p.ypart + (long)blockIdx.z * p.part_stride;
  for (int i = tid; i < BR * (NW / 4); i += 256) {
    const int r = i / (NW / 4), c4 = i % (NW / 4);
    const int prr = prow_s[r];
    if (prr < 0) continue;
    float4 v = ld4(out_s + r * OLD + c4 * 4);
    const int col = n0 + c4 * 4;
    const long o = (long)prr * p.ND + col;
    if (final_out) {
      if (p.bias) {
        const float4 b = ld4(p.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (p.add) {
        const float4 a = ld4(p.add + o);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      st4(p.y + o, v);
    } else {
      st4(yo + o, v);
    }
  }
}

// y = sum_z part[z] + bias + add   (fixed order -> deterministic)
__global__ void conv_part_reduce_kernel(const float* __restrict__ part, long stride, int nz, const float* __restrict__ bias,
                                        const act_t* __restrict__ add, act_t* __restrict__ y, long total4, int nd4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    float4 s = ld4q(part, i);
    int z = 1;
    for (; z + 1 < nz; z += 2) {  // nz is 3 or 9: two independent loads in flight, summed in z order
      const float4 v0 = ld4q(part + (long)z * stride, i);
      const float4 v1 = ld4q(part + (long)(z + 1) * stride, i);
      s.x = (s.x + v0.x) + v1.x; s.y = (s.y + v0.y) + v1.y; s.z = (s.z + v0.z) + v1.z; s.w = (s.w + v0.w) + v1.w;
    }
    for (; z < nz; ++z) {
      const float4 v = ld4q(part + (long)z * stride, i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
      const float4 b = ld4q(bias, i % nd4);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (add) {
      const float4 a = ld4q(add, i);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    st4q(y, i, s);
  }
}

// Packed MFMA B fragments of a 3^3 convolution weight w [cout][T][cin], both directions in one buffer:
//   fwd   (dir 0): (k = cin index,  j = cout index)  value w[j][t][k]
//   dgrad (dir 1): (k = cout index, j = cin index)   value w[k][t][j]
// element (k, t, j) of direction dir lives at  dir * cout*T*cin + ((t * (KD / 4) + k / 4) * ND + j) * 4 + k % 4:
// quads of 4 consecutive k are the 16-byte unit a lane loads, and the 32 columns of a wave's slice are contiguous.
__global__ __launch_bounds__(256) void conv_wpack_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int T,
                                                         int cin) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z % T, dir = blockIdx.z / T;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long half = (long)cout * T * cin;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = w[((long)(co0 + r) * T + t) * cin + ci0 + tx];  // [co][ci]
  __syncthreads();
  // output run of a thread group: 32 columns j x one k-quad (128 floats, contiguous)
  const int jj = (threadIdx.x >> 2) & 31, k4 = threadIdx.x & 3, qsel = threadIdx.x >> 7;  // 2 quads per pass
  for (int q = qsel; q < 8; q += 2) {
    if (dir == 0) {  // k = ci, j = co
      const int k = ci0 + q * 4 + k4, j = co0 + jj;
      wp[(((long)t * (cin / 4) + k / 4) * cout + j) * 4 + k4] = tile[jj][q * 4 + k4];
    } else {         // k = co, j = ci
      const int k = co0 + q * 4 + k4, j = ci0 + jj;
      wp[half + (((long)t * (cout / 4) + k / 4) * cin + j) * 4 + k4] = tile[q * 4 + k4][jj];
    }
  }
}

// bf16 packing of the same weights (PREC 1 / 3): per direction, 8 words per (tap, 16-k block kb, k half h, column j)
// at ((((t * (KD/16) + kb) * 2 + h) * ND + j) * 8: words 0..3 = bf16 hi pairs of k = kb*16 + h*8 + (0,1)(2,3)(4,5)(6,7),
// words 4..7 = the lo pairs (w - float(hi)).  Same buffer size as the fp32 packing.  with_lo == 0 (plain bf16 operands):
// 4 words per tuple at (...) * 4, the dgrad half still starts at cout * T * cin words.
__global__ __launch_bounds__(256) void conv_wpack_bf16_kernel(const float* __restrict__ w, unsigned* __restrict__ wp, int cout,
                                                              int T, int cin, int with_lo) {
  const long half = (long)cout * T * cin;
  const long per_dir = half / 8;  // (t, kb, h, j) tuples per direction
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * per_dir; i += (long)gridDim.x * blockDim.x) {
    const int dir = i >= per_dir;
    long r = i - dir * per_dir;
    const int KD = dir == 0 ? cin : cout, ND = dir == 0 ? cout : cin;
    const int j = (int)(r % ND); r /= ND;
    const int h = (int)(r & 1); r >>= 1;
    const int kb = (int)(r % (KD / 16));
    const int t = (int)(r / (KD / 16));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kb * 16 + h * 8 + e;
      v[e] = dir == 0 ? w[((long)j * T + t) * cin + k] : w[((long)k * T + t) * cin + j];
    }
    unsigned hi[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) csplit_bf16(v[2 * q], v[2 * q + 1], hi[q], lo[q]);
    if (with_lo) {
      uint4* o = reinterpret_cast<uint4*>(wp + dir * half + (i - dir * per_dir) * 8);
      o[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      o[1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    } else {  // plain bf16 operands: 4 words per tuple, the direction halves stay where they are
      *reinterpret_cast<uint4*>(wp + dir * half + (i - dir * per_dir) * 4) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    }
  }
}

static int tap_splits(int n, int ND) {
  const long base = (long)cdiv(n, 64 * (ND <= 64 ? 2 : 1)) * (ND <= 64 ? 1 : ND / 128);
  int nz = 1;
  while (nz < 9 && base * nz < 768) nz = nz == 1 ? 3 : 9;  // 27 taps -> 1, 3 or 9 groups
  // (Measured and removed, tools/conv_bench.py: one tap per block — 27 groups — on the small grids: level 4 (361 rows, C 768)
  // 146 -> 138 us, level 3 (1450, C 512) 213 -> 247: a block's (tap, chunk) chain is NOT what bounds the deep levels — every
  // 64-row tile streams its taps' weights (3 x C x 128 x 4 bytes) through one CU, 23 (level 3) / 6 (level 4) times the weight
  // tensor per launch; those levels now run as tap-grouped dense products, conv.hip.)
  return nz;
}

template <int NCS, int PREC>
static int launch_pairs_p(ConvP2& p, int nz, hipStream_t st) {
  using Cfg = PairsCfg<NCS, PREC>;
  const size_t sm = Cfg::bytes();
  static bool attr_set = false;  // per instantiation; the attribute call costs host time on every launch otherwise
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_pairs_kernel<NCS, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr_set = true; }
  dim3 grid(p.ND / (32 * NCS), cdiv(p.n, Cfg::BM * Cfg::NRT), nz);
  LOTUS_LAUNCH((conv_pairs_kernel<NCS, PREC>), grid, dim3(256), sm, st, p);
  LOTUS_LAUNCH_CHECK("lotus_subm_conv(pairs)");
  return LOTUS_OK;
}
// the packed weights and the kernel must agree on the operand precision (the caller passes the same `precision` to
// lotus_conv_weight_transpose and lotus_subm_conv)
template <int NCS>
static int launch_pairs(ConvP2& p, int nz, int prec, hipStream_t st) {
  if (prec == 3) return launch_pairs_p<NCS, 3>(p, nz, st);
  if (prec == 1) return launch_pairs_p<NCS, 1>(p, nz, st);
  return launch_pairs_p<NCS, 0>(p, nz, st);
}

// output-stationary bf16 kernel: 128-row blocks; taps split 1 / 3 / 9 ways until the grid covers the CUs
static int os_splits(int n, int ND) {
  const long base = (long)cdiv(n, 128) * (ND == 64 ? 1 : ND / 128);
  int nz = 1;
  while (nz < 9 && base * nz < 256) nz = nz == 1 ? 3 : 9;
  return nz;
}
static int os_mode() {  // LOTUS_CONV_OS=0: the bf16 operand modes use the pair-compacted kernel as well
  static int on = -1;
  if (on < 0) { const char* e = getenv("LOTUS_CONV_OS"); on = e ? atoi(e) : 1; }
  return on;
}

template <int PREC, int NS, int KCH>
static int launch_os_t(ConvP2& p, int nz, hipStream_t st) {
  using Cfg = OsCfg<PREC, NS, KCH>;
  const size_t sm = Cfg::bytes();
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv_os_kernel<PREC, NS, KCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr_set = true; }
  dim3 grid(p.ND / Cfg::NW, cdiv(p.n, Cfg::BR), nz);
  LOTUS_LAUNCH((conv_os_kernel<PREC, NS, KCH>), grid, dim3(256), sm, st, p);
  LOTUS_LAUNCH_CHECK("lotus_subm_conv(output-stationary bf16)");
  return LOTUS_OK;
}
static int launch_os(ConvP2& p, int nz, int prec, hipStream_t st) {
  const bool wide = p.ND != 64;
  if (prec == 0) {
    if constexpr (!LOTUS_ACT_IS_BF16) return wide ? launch_os_t<0, 4, 64>(p, nz, st) : launch_os_t<0, 2, 64>(p, nz, st);
    else return LOTUS_E_UNSUPPORTED;
  }
  if (prec == 3) return wide ? launch_os_t<3, 4, 64>(p, nz, st) : launch_os_t<3, 2, 64>(p, nz, st);
  if (p.KD % 128 == 0) return wide ? launch_os_t<1, 4, 128>(p, nz, st) : launch_os_t<1, 2, 128>(p, nz, st);
  return wide ? launch_os_t<1, 4, 64>(p, nz, st) : launch_os_t<1, 2, 64>(p, nz, st);
}

size_t lotus_conv_pairs_workspace(int n, int ND) {
  const int a = tap_splits(n, ND), b = os_splits(n, ND), nz = a > b ? a : b;  // either kernel may serve the call
  return nz > 1 ? (size_t)nz * n * ND * sizeof(float) : 0;
}

int lotus_conv_weight_transpose_impl(const float* w, float* wp, int cout, int T, int cin, int prec, hipStream_t st) {
  if (cin % 32 || cout % 32) {
    lotus_set_error("lotus_conv_weight_transpose: cin and cout must be multiples of 32 (got %d, %d)", cin, cout);
    return LOTUS_E_UNSUPPORTED;
  }
  if (prec != 0) {
    const long tuples = 2L * cout * T * cin / 8;
    const int g = (int)((tuples + 255) / 256);
    LOTUS_LAUNCH(conv_wpack_bf16_kernel, dim3(g > 8192 ? 8192 : g), dim3(256), 0, st, w, (unsigned*)wp, cout, T, cin, prec == 3 ? 1 : 0);
  } else {
    LOTUS_LAUNCH(conv_wpack_kernel, dim3(cin / 32, cout / 32, 2 * T), dim3(256), 0, st, w, wp, cout, T, cin);
  }
  LOTUS_LAUNCH_CHECK("lotus_conv_weight_transpose");
  return LOTUS_OK;
}

// returns 1 if the pair-compacted path handled the call, 0 if the shape is not eligible.
// Both modes read the packed weights w_t of lotus_conv_weight_transpose (fwd half / dgrad half).
int lotus_conv_pairs_try(int mode, const act_t* x, const float* w, const float* w_t, const float* bias, const act_t* add,
                         act_t* y, const int* nbr, const int* rowidx, int n, int T, int cin, int cout, void* workspace,
                         size_t workspace_bytes, int prec, hipStream_t st, int* rc) {
  const int KD = mode == 0 ? cin : cout, ND = mode == 0 ? cout : cin;
  if (T != 27 || KD % 32 || ND % 64 || (ND > 64 && ND % 128)) return 0;
  if (!w_t || ((uintptr_t)w_t) % 16) return 0;
  if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)bias) | ((uintptr_t)add)) % 16) return 0;
  // exact fp32 products on the output-stationary kernel too (LOTUS_CONV_OS_F32 = 1 all eligible shapes | 2 the 64-wide
  // layers | 3 also small levels; fp32 storage only; OPT-IN): 2-3.7x the MFMAs of the compacted form but none of its folds,
  // tables and fills.  Measured (tools/conv_bench.py, us, pair-compacted -> output-stationary): level 0 C 64 189 -> 145,
  // level 1 C 64 111 -> 77, level 2 C 128 80 -> 69, but level 1 C 128 220 -> 231 (9 tap groups; 264 with 3), level 2
  // C 256 227 -> 226, level 4 135 -> 175; in the step 867 -> 872 samples/s with the 64-wide layers only, 839 with all —
  // the compacted kernel stays the fp32 default.  (Its different summation order is enough to flip max-pool arg-max ties
  // on the v1_init fixture: whole-gradient error 1.5e-3 against the 1e-4 bar the default path meets there — the re-routing
  // sensitivity tests/test_gpu_fullsize_oracle.py measures; per-op it is within 3e-6 of fp64 like the default.)
  static int os_f32 = -1;
  if (os_f32 < 0) { const char* e = getenv("LOTUS_CONV_OS_F32"); os_f32 = e ? atoi(e) : 0; }
  const bool f32_fit = os_f32 == 1 || (os_f32 >= 2 && ND == 64) || (os_f32 == 3 && n <= 8192 && ND <= 256);
  const bool os = KD % 64 == 0 && (prec != 0 ? os_mode() != 0 : (f32_fit && !LOTUS_ACT_IS_BF16));
  const int nz = os ? os_splits(n, ND) : tap_splits(n, ND);
  if (nz > 1 && (!workspace || workspace_bytes < (size_t)nz * n * ND * sizeof(float))) return 0;
  ConvP2 p;
  p.x = x; p.bias = bias; p.add = add; p.nbr = nbr; p.rowidx = rowidx;
  p.n = n; p.T = T; p.KD = KD; p.ND = ND; p.mirror = mode == 1;
  p.w = w_t + (mode == 0 ? 0 : (long)cout * T * cin);
  p.tpz = cdiv(T, nz);
  p.y = y; p.ypart = nz > 1 ? (float*)workspace : nullptr;
  p.part_stride = nz > 1 ? (long)n * ND : 0;
  if (os) *rc = launch_os(p, nz, prec, st);
  else *rc = ND == 64 ? launch_pairs<2>(p, nz, prec, st) : launch_pairs<4>(p, nz, prec, st);
  if (*rc == 0 && nz > 1) {
    const long total4 = (long)n * ND / 4;
    int g = cdiv(total4, 256);
    LOTUS_LAUNCH(conv_part_reduce_kernel, dim3(g > 2048 ? 2048 : g), dim3(256), 0, st, (const float*)workspace,
                       (long)n * ND, nz, bias, add, y, total4, ND / 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { lotus_set_error("conv_part_reduce: %s", hipGetErrorString(e)); *rc = LOTUS_E_LAUNCH; }
  }
  return 1;
}

}  // namespace LOTUS_NS
