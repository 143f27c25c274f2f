// lotus-hip: dense exact-fp32 GEMM with LDS-DMA operand staging and a VALU-free data path (round 5) — the nn.Linear call
// sites of the hot path (/root/reference/genrobo3d/models/PointTransformerV3/model.py:543-583, model_ca.py:46-101).
//
// The rule this kernel is built around was measured this round (tools/ubench/mfma_coissue.hip):
// on gfx950 an fp32 MFMA (`v_mfma_f32_32x32x2_f32`, 64 cycles) and ANY vector-ALU instruction — of the same wave or of
// another wave of the SIMD — never overlap: a co-resident wave's VALU instructions are starved while MFMAs are issued
// back to back, and each VALU instruction of the MFMA wave itself adds ~5 cycles.  Scalar-ALU, LDS and vector-memory
// instructions DO issue beside the MFMAs.  So the time of a SIMD is 64 x (MFMAs) + ~5 x (VALU instructions) of everything
// resident on it, and a dense layer is MFMA-bound only if its data path uses no VALU:
//  * operands go global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB per wave instruction): no staging
//    registers, no ds_write pass; the per-lane offset is loop invariant, the buffer descriptor lives in SGPRs and is
//    advanced with scalar instructions, out-of-range rows / reduction tails read as zeros through the descriptor's bounds
//    check, and NST - 1 slabs are in flight across `s_barrier` under counted `s_waitcnt vmcnt(N)` (the DMA statements are
//    inline asm: the compiler would otherwise drain them with vmcnt(0) in front of every ds_read of the running slab);
//  * the LDS image of a k-contiguous operand is lane-linear [rows][BK] (what LDS-DMA writes) with the 16-byte chunks of
//    a row XOR-swizzled on the SOURCE side, so a lane's fragment run is read with conflict-free ds_read_b128 at
//    loop-invariant addresses (+ immediate stage offsets);
//  * block tiles up to 128 x 128 with a 64 x 64 wave tile (2 x 2 accumulators of 32 x 32): 0.25 LDS reads per MFMA;
//  * the product is computed TRANSPOSED (MFMA "A" operand = the weight-side fragment, "B" operand = the activation-side
//    fragment): a lane owns ONE output row and four consecutive accumulator registers are four consecutive output
//    columns.  The accumulators are VGPRs (this translation unit is compiled with -mllvm -amdgpu-mfma-vgpr-form), so the
//    epilogue moves them with ds_write_b128 into a wave-private LDS tile, reads whole rows back and leaves with
//    `buffer_store_dwordx4`: one per-lane byte offset per 32 rows, the row advance in the scalar offset — no accumulator
//    copies, no 64-bit address arithmetic, 4 x 256 contiguous bytes per store instruction;
//  * the bias is the INITIAL value of the accumulators (loaded straight into them), not an addition behind the products.
// Per output element: the products in the k order of gemm_kernel with the same slab depth (MFMA step s of lane half h
// multiplies k = h * BK / 2 + s), on top of the bias instead of under it.
#pragma once
#include "gemm_common.h"

namespace LOTUS_NS {

typedef int dma_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dma_u32x4 __attribute__((ext_vector_type(4)));
typedef float dma_f32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0): an access at byte offset >= bytes returns zero / is dropped and touches no memory
__device__ __forceinline__ dma_i32x4 dma_rsrc(unsigned long long base, unsigned bytes) {
  dma_i32x4 r;
  r.x = (int)(unsigned)base;
  r.y = (int)(unsigned)(base >> 32);
  r.z = (int)bytes;
  r.w = 0x00020000;
  return r;
}

// one LDS-DMA wave instruction: lane l copies the 16 bytes at rsrc.base + voff(l) to LDS byte address lds + 16 l.
// M0 (the LDS destination base) is written in the statement that reads it and restored (compiler-reserved register).
__device__ __forceinline__ void dma16(dma_i32x4 rsrc, int voff, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds), "v"(voff), "s"(rsrc)
               : "memory");
}

template <int N> __device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N) : "memory"); }

// k-contiguous operand: rows of BK floats, 16-byte chunk c of row r stored at chunk position c ^ swz(r)
template <int BK>
__device__ __forceinline__ int kc_swz(int row) {
  return BK == 64 ? (row & 15) : BK == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

// four consecutive elements of the epilogue behind the bias: activation, act'(pre of the producing layer), dropout — the
// part of gemm_epilogue4 that is arithmetic (the loads / stores around it are buffer instructions at the call site)
__device__ __forceinline__ void dma_epi_math4(const GemmP& p, float (&v)[4], const dma_f32x4 mp, unsigned long long idx4) {
  if (p.act != LOTUS_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_f(v[e], p.act);
  }
  if (p.mulpre) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= act_grad_f(mp[e], p.dact);
  }
  if (p.drop_thresh) {
    float dm[4];
    dropout_scale4(p.drop_seed, idx4, p.drop_thresh, p.drop_inv_keep, dm);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= dm[e];
  }
}

// XKC / WKC: the activation-side (rows of C) / weight-side (columns of C) operand is contiguous along the reduction
// index (image [R][BK], swizzled); otherwise it is contiguous along its own output index (image [BK][R]).
//   fwd (1, 1), dgrad (1, 0), wgrad (0, 0)
// EPI: 0 = bias / saved pre-activation / residual only (no per-element arithmetic besides the residual add), 1 = everything,
//      2 = LayerNorm backward of the product (input gradients whose block tile covers whole rows: BN == N; GemmP::ln_*)
// ABL (kernel lab only): 1 skips the epilogue stores, 2 the DMA (operands are whatever the LDS holds), 3 both
// The block program: output tile (by, bx) of split bz (of nzgrid) of the product `p`.
// TAP: a row tile of the tap-grouped sparse convolution (GemmP::a_rows ...; conv.hip): the tile starts at row `tap_m0` of the
// partial slab, its activation rows are GATHERED through a_rows (the per-lane DMA offset is the gathered row: still loop
// invariant), rows at and past `tap_mb` do not exist (they read as zeros and are not stored), the weight-side operand is the
// slice of tap `tap_w`.
template <int BM, int BN, int BK, int NST, bool XKC, bool WKC, bool SUM_A, int EPI, int ABL = 0, bool TAP = false>
__device__ __forceinline__ void gemm_dma_block(const GemmP& p, int bx, int by, int bz, int nzgrid, int tap_m0 = 0, int tap_mb = 0,
                                               int tap_w = 0) {
  static_assert(!TAP || (XKC && !SUM_A && EPI == 0), "tap-grouped tiles: gathered k-contiguous rows, plain stores");
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32, KS = BK / 2;
  constexpr int A_FL = BM * BK, B_FL = BN * BK, ST_FL = A_FL + B_FL;  // floats per image / stage
  constexpr int TA = A_FL / 1024, TB = B_FL / 1024, D = TA + TB;       // DMA instructions per wave and slab
  static_assert(TA >= 1 && TB >= 1 && (TM == 1 || TM == 2) && (TN == 1 || TN == 2), "tile");
  static_assert(NST >= 2 && NST <= 4 && D * (NST - 2) < 64, "stages");
  constexpr int EP_FL = 4 * 32 * (WTN + 4) + (EPI == 2 ? 4 * 32 * 2 + 4 * 2 * WTN : 0);  // epilogue: one [32][WTN + 4] tile per wave (+ LN sums)
  __shared__ __attribute__((aligned(1024))) float smem[NST * ST_FL > EP_FL ? NST * ST_FL : EP_FL];

  const int tid = threadIdx.x, lane = tid & 63, l31 = tid & 31, h = (tid >> 5) & 1;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = TAP ? tap_m0 : by * BM, n0 = bx * BN;
  const int Mb = TAP ? tap_mb : p.M;  // row bound of this tile
  const int kbeg = bz * p.klen, kend = min(p.K, kbeg + p.klen);
  const int nslab = (kend - kbeg + BK - 1) / BK;
  const int lda = (int)p.lda, ldb = (int)p.ldb, ldc = (int)p.ldc;

  // ---- DMA plan: per-lane byte offsets (loop invariant) and the two descriptors at slab 0
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  int voa[TA], vob[TB];
  unsigned long long base_a, base_b;
  long rec_a, rec_b;        // bytes still inside the operand, seen from the descriptor base
  unsigned step_a, step_b;  // descriptor advance per slab
  if (XKC) {
    constexpr int CH = BK / 4, RPI = 64 / CH;
    const int row0 = lane / CH, slot = lane % CH;
#pragma unroll
    for (int t = 0; t < TA; ++t) {
      const int row = (wave * TA + t) * RPI + row0;
      if (TAP) {  // gathered row (a tile row past the bound: an offset no descriptor covers -> zeros)
        const int g = m0 + row < Mb ? p.a_rows[m0 + row] : -1;
        voa[t] = g >= 0 ? g * (lda * 4) + ((slot ^ kc_swz<BK>(row)) << 4) : 0x7ffffff0;
      } else {
        voa[t] = row * (lda * 4) + ((slot ^ kc_swz<BK>(row)) << 4);
      }
    }
    if (TAP) {  // the descriptor spans the whole gathered tensor
      base_a = (unsigned long long)(static_cast<const float*>(p.A) + kbeg);
      rec_a = ((long)(p.a_src_rows - 1) * p.lda + (kend - kbeg)) * 4;
    } else {
      base_a = (unsigned long long)(static_cast<const float*>(p.A) + (long)m0 * p.lda + kbeg);
      rec_a = ((long)(min(BM, p.M - m0) - 1) * p.lda + (kend - kbeg)) * 4;
    }
    step_a = BK * 4;
  } else {
    constexpr int C4 = BM / 4, RPI = 64 / C4;
#pragma unroll
    for (int t = 0; t < TA; ++t) voa[t] = (((wave * TA + t) * RPI + lane / C4) * lda + m0 + (lane % C4) * 4) * 4;
    base_a = (unsigned long long)(static_cast<const float*>(p.A) + (long)kbeg * p.lda);
    rec_a = (long)(kend - kbeg) * p.lda * 4;
    step_a = BK * (unsigned)lda * 4;
  }
  if (WKC) {
    constexpr int CH = BK / 4, RPI = 64 / CH;
    const int row0 = lane / CH, slot = lane % CH;
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      const int row = (wave * TB + t) * RPI + row0;
      vob[t] = row * (ldb * 4) + ((slot ^ kc_swz<BK>(row)) << 4);
    }
    base_b = (unsigned long long)(static_cast<const float*>(p.B) + (long)n0 * p.ldb + kbeg + (TAP ? (long)tap_w * p.b_tap_stride : 0L));
    rec_b = ((long)(min(BN, p.N - n0) - 1) * p.ldb + (kend - kbeg)) * 4;
    step_b = BK * 4;
  } else {
    constexpr int C4 = BN / 4, RPI = 64 / C4;
#pragma unroll
    for (int t = 0; t < TB; ++t) vob[t] = (((wave * TB + t) * RPI + lane / C4) * ldb + n0 + (lane % C4) * 4) * 4;
    base_b = (unsigned long long)(static_cast<const float*>(p.B) + (long)kbeg * p.ldb + (TAP ? (long)tap_w * p.b_tap_stride : 0L));
    rec_b = (long)(kend - kbeg) * p.ldb * 4;
    step_b = BK * (unsigned)ldb * 4;
  }
  auto issue = [&](int slab, int stage) {  // the DMA instructions of this wave for one slab (all wave-uniform scalars)
    const dma_i32x4 ra = dma_rsrc(base_a + (unsigned long long)slab * step_a, (unsigned)max(0L, rec_a - (long)slab * step_a));
    const dma_i32x4 rb = dma_rsrc(base_b + (unsigned long long)slab * step_b, (unsigned)max(0L, rec_b - (long)slab * step_b));
    const unsigned la = lds0 + (unsigned)(stage * ST_FL * 4) + (unsigned)(wave * TA) * 1024u;
    const unsigned lb = lds0 + (unsigned)((stage * ST_FL + A_FL) * 4) + (unsigned)(wave * TB) * 1024u;
#pragma unroll
    for (int t = 0; t < TA; ++t) dma16(ra, voa[t], la + t * 1024u);
#pragma unroll
    for (int t = 0; t < TB; ++t) dma16(rb, vob[t], lb + t * 1024u);
  };
  long long* dbg = (ABL & 8) ? reinterpret_cast<long long*>(p.bias_part) + (long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;  // lab: phase timestamps
  if ((ABL & 8) && tid == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[0] = (long long)wall_clock64(); dbg[1] = clock64(); dbg[6] = hwid; dbg[7] = xcc;
  }
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nslab && !(ABL & 2)) issue(s, s);

  // ---- accumulators, initialised with the bias of their columns: register r of tile (tm, tn) is column
  // n0 + wn WTN + tn 32 + (r & 3) + 8 (r >> 2) + 4 h of row m0 + wm WTM + tm 32 + l31
  const bool fused = p.cnt != nullptr && nzgrid > 1;
  const int colb = n0 + wn * WTN + 4 * h;
  f32x16 acc[TM][TN];
  if (p.bias && !fused && bz == 0) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = ld4(p.bias + min(colb + tn * 32 + 8 * g, p.N - 4));
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          acc[tm][tn][4 * g] = b4.x; acc[tm][tn][4 * g + 1] = b4.y; acc[tm][tn][4 * g + 2] = b4.z; acc[tm][tn][4 * g + 3] = b4.w;
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  float asum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) asum[i] = 0.f;
  const bool want_sum = SUM_A && p.bias_part && bx == 0 && wn == 0;  // (wave-uniform) column sums of the activation-side operand

  // fragment addresses (floats, inside a stage)
  int xoff[TM][XKC ? KS / 4 : 1], woff[TN][WKC ? KS / 4 : 1];
  if (XKC) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int row = wm * WTM + tm * 32 + l31;
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) xoff[tm][q] = row * BK + (((h * (KS / 4) + q) ^ kc_swz<BK>(row)) << 2);
    }
  } else {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) xoff[tm][0] = h * KS * BM + wm * WTM + tm * 32 + l31;
  }
  if (WKC) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int row = wn * WTN + tn * 32 + l31;
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) woff[tn][q] = A_FL + row * BK + (((h * (KS / 4) + q) ^ kc_swz<BK>(row)) << 2);
    }
  } else {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) woff[tn][0] = A_FL + h * KS * BN + wn * WTN + tn * 32 + l31;
  }

  auto compute = [&](const float* __restrict__ st, auto sum_tag) {
    constexpr bool SUM = decltype(sum_tag)::value;
    float xf[TM][KS], wf[TN][KS];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (XKC) {
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          const float4 v = ld4(st + xoff[tm][q]);
          xf[tm][4 * q] = v.x; xf[tm][4 * q + 1] = v.y; xf[tm][4 * q + 2] = v.z; xf[tm][4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[tm][s] = st[xoff[tm][0] + s * BM];
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      if (WKC) {
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          const float4 v = ld4(st + woff[tn][q]);
          wf[tn][4 * q] = v.x; wf[tn][4 * q + 1] = v.y; wf[tn][4 * q + 2] = v.z; wf[tn][4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[tn][s] = st[woff[tn][0] + s * BN];
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        if (SUM) asum[tm] += xf[tm][s];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[tn][s], xf[tm][s], acc[tm][tn], 0, 0, 0);
      }
  };

  // ---- slab pipeline: NST - 1 slabs in flight; one barrier per slab.  At the top of iteration i this wave's DMAs of slab
  // i have landed (counted vmcnt: the younger slabs stay in flight) and its fragment reads of slab i - 1 have returned
  // (lgkmcnt); behind the barrier that holds for every wave, so slab i may be read and the stage of slab i - 1 refilled.
  for (int i0 = 0; i0 < nslab; i0 += NST) {
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      const int i = i0 + s;
      if (i < nslab) {
        if (i + NST - 2 < nslab) dma_wait<D*(NST - 2)>();
        else dma_wait<0>();
        __builtin_amdgcn_s_barrier();
        if ((ABL & 8) && i == 0 && tid == 0) dbg[2] = clock64();
        if (i + NST - 1 < nslab && !(ABL & 2)) issue(i + NST - 1, (s + NST - 1) % NST);
        if (want_sum) compute(smem + s * ST_FL, std::true_type{});
        else compute(smem + s * ST_FL, std::false_type{});
      }
    }
  }

  if ((ABL & 8) && tid == 0) dbg[3] = clock64();
  // ---- epilogue without vector-ALU work in its data path.  The lane that owns output row l31 writes its 16-byte
  // accumulator quads into a wave-private LDS tile [32][WTN + 4] (ds_write_b128 straight from the accumulator VGPRs), the
  // wave reads the tile back as whole rows (WTN / 4 lanes per row) and leaves with buffer instructions that cover 4 x 256
  // (8 x 128) contiguous bytes each; the per-lane byte offset is computed once per 32 rows, the row advance is the scalar
  // offset of the instruction.  (Measured, tools/lab: storing row-per-lane — 32 rows x 32 bytes per instruction — straight from
  // the accumulators is 7-13 % slower on the store-heavy layers: the CU's one vector-memory path then holds back the operand
  // DMAs of the other resident blocks; non-temporal stores 2-3x slower.)
  constexpr int TLD = WTN + 4, LPR = WTN / 4, RPI = 64 / LPR, NIT = 32 / RPI;
  const unsigned cbytes = (unsigned)min((long)Mb * p.ldc * 4, 0xfffffffcL);
  const long zoff = (long)bz * p.part_stride;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
      fused ? (void*)(p.part + zoff) : (void*)(static_cast<float*>(p.C) + zoff), 0, (int)cbytes, 0x00020000);
  // (an absent tensor gets an empty descriptor: its loads return zeros, nothing below stores through it)
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(p.pre ? (void*)p.pre : p.C, 0, p.pre ? (int)cbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(p.residual ? (void*)p.residual : p.C, 0, p.residual ? (int)cbytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(p.mulpre ? (void*)p.mulpre : p.C, 0, p.mulpre ? (int)cbytes : 0, 0x00020000);
  dma_wait<0>();
  __builtin_amdgcn_s_barrier();  // every wave has read its last slab: the stages are free
  float* vt = smem + wave * (32 * TLD);
  const int c4 = lane % LPR, rsub = lane / LPR;
  const int colw = n0 + wn * WTN + c4 * 4;
  const bool full_rows = m0 + BM <= Mb;  // (block-uniform) the scalar row advance is not bounds-checked: whole tiles only
  const bool math = EPI == 1 && (p.act != LOTUS_ACT_NONE || p.mulpre || p.drop_thresh);
  if constexpr (EPI == 2) {
    // ---- LayerNorm backward of the product, in the row-contiguous layout (LPR lanes per row, 4 columns per lane): the two
    // row sums are reduced over the lanes of a row by butterflies and over the two column waves through LDS; the column
    // partials of dgamma / dbeta over the 128 rows of the tile go to ln_part[row tile][2][N] (fixed order throughout)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_x, 0, (int)cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(p.ln_dz ? (void*)p.ln_dz : p.C, 0, p.ln_dz ? (int)cbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmu = __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_mean, 0, p.M * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_rstd, 0, p.M * 4, 0x00020000);
    float* rowred = smem + 4 * 32 * TLD;        // [4 waves][32 rows][2]
    float* colred = rowred + 4 * 32 * 2;        // [4 waves][2][WTN]
    const float4 gam = ld4(p.ln_gamma + colw);
    const float inv_c = 1.f / (float)p.N;
    float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          st4(vt + l31 * TLD + tn * 32 + 8 * g + 4 * h, make_float4(acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1], acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]));
      const int row0 = m0 + wm * WTM + tm * 32 + rsub;
      const int vo = (row0 * ldc + colw) * 4;
      dma_f32x4 xv[NIT], ad[NIT], gv[NIT];
      float mu[NIT], rs[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {  // (VGPR row offsets: bounds-checked, the last row tile may be ragged)
        xv[it] = __builtin_bit_cast(dma_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + it * RPI * ldc * 4, 0, 0));
        ad[it] = __builtin_bit_cast(dma_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, vo + it * RPI * ldc * 4, 0, 0));
        mu[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmu, (row0 + it * RPI) * 4, 0, 0));
        rs[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (row0 + it * RPI) * 4, 0, 0));
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const float4 t4 = ld4(vt + (it * RPI + rsub) * TLD + c4 * 4);
        gv[it] = dma_f32x4{t4.x, t4.y, t4.z, t4.w};
      }
      dma_f32x4 xh[NIT], gy[NIT];
      float s1[NIT], s2[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const float nm = -mu[it] * rs[it];
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gmv = e == 0 ? gam.x : e == 1 ? gam.y : e == 2 ? gam.z : gam.w;
          xh[it][e] = fmaf(xv[it][e], rs[it], nm);
          gy[it][e] = gv[it][e] * gmv;
          a1 += gy[it][e];
          a2 = fmaf(gy[it][e], xh[it][e], a2);
          pg[e] = fmaf(gv[it][e], xh[it][e], pg[e]);
          pb[e] += gv[it][e];
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
        s1[it] = a1; s2[it] = a2;
        if (c4 == 0) *reinterpret_cast<float2*>(rowred + (wave * 32 + it * RPI + rsub) * 2) = make_float2(a1, a2);
      }
      __syncthreads();  // (no DMA is in flight: a plain barrier)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const float2 o2 = *reinterpret_cast<const float2*>(rowred + ((wave ^ 1) * 32 + it * RPI + rsub) * 2);
        const float m1 = (wn == 0 ? s1[it] + o2.x : o2.x + s1[it]) * inv_c, m2 = (wn == 0 ? s2[it] + o2.y : o2.y + s2[it]) * inv_c;
        dma_f32x4 dxv;
#pragma unroll
        for (int e = 0; e < 4; ++e) dxv[e] = fmaf(rs[it], gy[it][e] - m1 - xh[it][e] * m2, ad[it][e]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dma_u32x4, dxv), rc, vo + it * RPI * ldc * 4, 0, 0);
        if (p.ln_dz) {
          float dm[4];
          dropout_scale4(p.drop_seed, (unsigned long long)((long)(row0 + it * RPI) * ldc + colw), p.drop_thresh, p.drop_inv_keep, dm);
          const dma_f32x4 z = {dxv[0] * dm[0], dxv[1] * dm[1], dxv[2] * dm[2], dxv[3] * dm[3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dma_u32x4, z), rz, vo + it * RPI * ldc * 4, 0, 0);
        }
      }
      __syncthreads();  // rowred / vt are rewritten by the next pass
    }
    // column partials: over the row lanes of the wave (lanes with the same c4), then over the two row waves
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { pg[e] += __shfl_xor(pg[e], o, 64); pb[e] += __shfl_xor(pb[e], o, 64); }
    }
    if (rsub == 0) {
      st4(colred + (wave * 2 + 0) * WTN + c4 * 4, make_float4(pg[0], pg[1], pg[2], pg[3]));
      st4(colred + (wave * 2 + 1) * WTN + c4 * 4, make_float4(pb[0], pb[1], pb[2], pb[3]));
    }
    __syncthreads();
    if (tid < 2 * BN) {  // which = tid / BN, column = tid % BN: rows of wave (wm 0, wn) + rows of wave (wm 1, wn)
      const int which = tid / BN, col = tid % BN, cw = col / WTN, cc = col % WTN;
      const float sum = colred[((0 * 2 + cw) * 2 + which) * WTN + cc] + colred[((1 * 2 + cw) * 2 + which) * WTN + cc];
      if (n0 + col < p.N) p.ln_part[((long)by * 2 + which) * p.N + n0 + col] = sum;
    }
    return;
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    if (ABL & 1) {  // lab: keep the accumulators alive without storing them
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[tm][tn][r]));
      continue;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        st4(vt + l31 * TLD + tn * 32 + 8 * g + 4 * h, make_float4(acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1], acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]));
    const int row0 = m0 + wm * WTM + tm * 32 + rsub;
    const int vo = (row0 * ldc + colw) * 4;
    if (colw >= p.N) continue;
    auto ld = [&](__amdgpu_buffer_rsrc_t rs, int it) {
      return full_rows ? __builtin_bit_cast(dma_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, it * RPI * ldc * 4, 0))
                       : __builtin_bit_cast(dma_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + it * RPI * ldc * 4, 0, 0));
    };
    auto st = [&](__amdgpu_buffer_rsrc_t rs, int it, dma_f32x4 v, auto aux_tag) {
      constexpr int AUX = decltype(aux_tag)::value;
      if (full_rows) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dma_u32x4, v), rs, vo, it * RPI * ldc * 4, AUX);
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dma_u32x4, v), rs, vo + it * RPI * ldc * 4, 0, AUX);
    };
    dma_f32x4 v[NIT], r4[NIT], mp[NIT];
    if (p.residual) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) r4[it] = ld(rr, it);
    }
    if (math && p.mulpre) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) mp[it] = ld(rm, it);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const float4 t4 = ld4(vt + (it * RPI + rsub) * TLD + c4 * 4);
      v[it] = dma_f32x4{t4.x, t4.y, t4.z, t4.w};
    }
    if (fused) {  // raw partial of this split, write-through (sc1): read by the last block of the tile, possibly on another XCD
#pragma unroll
      for (int it = 0; it < NIT; ++it) st(rc, it, v[it], std::integral_constant<int, 16>{});
      continue;
    }
    if (p.pre) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) st(rp, it, v[it], std::integral_constant<int, 0>{});
    }
    if (math) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        float w4[4] = {v[it][0], v[it][1], v[it][2], v[it][3]};
        dma_epi_math4(p, w4, mp[it], (unsigned long long)((long)(row0 + it * RPI) * ldc + colw));
        v[it] = dma_f32x4{w4[0], w4[1], w4[2], w4[3]};
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (p.residual) v[it] += r4[it];
      st(rc, it, v[it], std::integral_constant<int, 0>{});
    }
  }
  if (ABL & 8) {
    if (tid == 0) dbg[4] = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { dbg[5] = clock64(); }
  }
  if (SUM_A) {
    if (want_sum) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const float s = asum[tm] + __shfl_xor(asum[tm], 32, 64);
        const int row = m0 + wm * WTM + tm * 32 + l31;
        if (h == 0 && row < p.M) __hip_atomic_store(p.bias_part + (long)bz * p.bias_stride + row, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (fused) splitk_fused_tail<SUM_A, BM, BN, float>(p, bx, by, m0, n0, tid);
}

template <int BM, int BN, int BK, int NST, bool XKC, bool WKC, bool SUM_A, int EPI, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_dma_kernel(GemmP p) {
  // XCD-aware tile order (as gemm_kernel): split-K launches give XCD c the splits z = c (mod 8); otherwise every XCD walks a
  // contiguous run of the row-major tile list, so the column blocks of one row tile share an L2
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (gridDim.z > 1 && (gridDim.z & 7) == 0) {
    const int tiles = gridDim.x * gridDim.y;
    const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
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
  gemm_dma_block<BM, BN, BK, NST, XKC, WKC, SUM_A, EPI, ABL>(p, bx, by, bz, (int)gridDim.z);
}

// The tap-grouped sparse convolution (conv.hip: 27 gathered products in one launch): the M axis is 27 segments of tap_rows rows,
// segment t holds tap_cnt[t] (activation row, output row) pairs, padded with valid rows to a multiple of 64.  Block `lin` of the
// XCD-ordered list takes the lin-th ACTIVE row tile (a scan over the 27 counters: scalar loads), so the tiles of one tap — one
// weight slice, neighbouring gathered rows — run on one XCD; blocks past the last active tile leave at once.
template <int BM, int BN, int BK, int NST, bool WKC>
__global__ __launch_bounds__(256) void gemm_dma_tap_kernel(GemmP p) {
  int nt[27], total_t = 0;  // (statically indexed, wave-uniform: scalar registers)
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    nt[t] = (p.tap_cnt[t] + BM - 1) / BM;
    total_t += nt[t];
  }
  const int nbx = gridDim.x, total = nbx * total_t;
  const int lin = blockIdx.y * nbx + blockIdx.x;
  if (lin >= total) return;
  const int xcd = lin & 7, slot = lin >> 3, q = total >> 3, r = total & 7;
  const int tl = xcd * q + min(xcd, r) + slot;
  int tile = tl / nbx;
  const int bx = tl - tile * nbx;
  int tap = 0, base = 0, c = 0;
#pragma unroll
  for (int t = 0; t < 26; ++t) {
    c += nt[t];
    if (tile >= c) { tap = t + 1; base = c; }
  }
  tile -= base;
  const int cnt = p.tap_cnt[tap], seg = tap * p.tap_rows;
  gemm_dma_block<BM, BN, BK, NST, true, WKC, false, 0, 0, true>(p, bx, 0, 0, 1, seg + tile * BM, seg + ((cnt + 63) & ~63),
                                                                 p.b_tap_mirror ? 26 - tap : tap);
}

// (Measured and removed in round 5: a grouped launch — one block program per (problem, tile) from a device table — for the
// DEFERRED weight gradients of the deep levels, recorded while the composite backward passes ran and flushed once per stage:
// stand-alone 631 -> 422 us (1450 rows), 206 -> 125 us (361 rows), 645 -> 499 us (6077 rows) for the stage's 21 / 10 / 21 products,
// -110 launches per step — and 946-949 against 943-947 samples/s in the step, inside the noise: the weight-gradient stream is not
// what the step waits for.  DESIGN.md section 4, round 5.)

}  // namespace LOTUS_NS
