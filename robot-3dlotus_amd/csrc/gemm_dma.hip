// lotus-hip: host side of the LDS-DMA dense kernels (gemm_dma.h) — which products they take and on which tile.  Its own
// translation unit because it is compiled with -mllvm -amdgpu-mfma-vgpr-form (accumulators in VGPRs: the epilogue stores
// straight from them); gemm.hip keeps the default register form for the kernels that were tuned with it.
#include "gemm_dma.h"
#include <stdlib.h>

namespace LOTUS_NS {

static int dma_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// layout: 0 forward (both operands k-contiguous), 1 input gradient (A k-contiguous, B column-contiguous), 2 weight gradient
// (both operands contiguous along their output index, column sums of A on request).  Returns LOTUS_GEMM_DMA_NA when the
// product is outside what these kernels take (the caller then runs gemm_kernel).
int launch_gemm_dma(GemmP& p, int layout, int nz, hipStream_t st) {
  const bool ln = p.ln_x != nullptr;  // LayerNorm backward of the product in the epilogue (layout 1, whole rows per tile)
  if constexpr (LOTUS_ACT_IS_BF16) {
    return LOTUS_GEMM_DMA_NA;  // (bf16 activation storage: gemm_kernel's bf16 paths)
  } else {
    // LOTUS_GEMM_DMA: 0 = off; LOTUS_GEMM_DMA_MINROWS: smallest row count (forward / input gradient: rows of the activation
    // operand; weight gradient: length of the reduction) these kernels are used from.  Measured (tools/lab/gemm_lab, MI355X):
    // they win from ~16 k rows; below that the grids are too small for 128-row tiles and gemm_kernel's deep-slab / split-K
    // forms are the better fit.
    static int on = -1, minrows = 0;
    if (on < 0) { on = dma_env("LOTUS_GEMM_DMA", 1); minrows = dma_env("LOTUS_GEMM_DMA_MINROWS", 16384); }
    if (!on || p.prec != 0 || p.tap_rows || p.b_act) return LOTUS_GEMM_DMA_NA;
    const int rows = layout == 2 ? p.K : p.M;
    if (rows < minrows) return LOTUS_GEMM_DMA_NA;
    const bool epi = !ln && (p.act != LOTUS_ACT_NONE || p.mulpre || p.drop_thresh);
    if (p.accumulate && !(p.cnt && nz > 1)) return LOTUS_GEMM_DMA_NA;
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    const long abytes = layout == 2 ? (long)p.K * p.lda * 4 : (long)p.M * p.lda * 4;
    const long bbytes = layout == 0 ? (long)p.N * p.ldb * 4 : (long)p.K * p.ldb * 4;
    if (!al16(p.A) || !al16(p.B) || !al16(p.C) || (p.lda & 3) || (p.ldb & 3) || (p.ldc & 3) || (p.N & 3) || (p.part_stride & 3) ||
        (p.bias && !al16(p.bias)) || (p.residual && !al16(p.residual)) || (p.pre && !al16(p.pre)) || (p.mulpre && !al16(p.mulpre)) ||
        (p.part && !al16(p.part)) || abytes >= (1L << 32) || bbytes >= (1L << 32) || (long)p.M * p.ldc * 4 >= (1L << 32))
      return LOTUS_GEMM_DMA_NA;
    // a k-contiguous operand is staged in whole slabs: the reduction (and every split of it) must be a multiple of the slab depth
    const bool wide = p.N > 64;
    const int bk = wide ? 16 : 32;
    // 128-row tiles need a grid that fills the GPU: measured in the step (bench.py --gemm-report), the level-1 products
    // (23 894 rows) with <= 128 output columns — 187 blocks — lose 10-20 % to gemm_kernel's 64 x 64 tiles, the 512-wide ones win
    static int minblocks = -1;
    if (minblocks < 0) minblocks = dma_env("LOTUS_GEMM_DMA_MINBLOCKS", 400);
    // (with the LayerNorm backward in the epilogue a whole kernel and four HBM passes go away: worth it on any grid that
    // covers half the CUs)
    if (layout != 2 && (long)cdiv(p.M, 128) * cdiv(p.N, wide ? 128 : 64) * nz < (ln ? 128 : minblocks)) return LOTUS_GEMM_DMA_NA;
    if (ln && (layout != 1 || nz != 1 || (p.N != 64 && p.N != 128) || p.ldc != p.N || p.act != LOTUS_ACT_NONE || p.mulpre || p.pre ||
               p.bias || !al16(p.ln_x) || !al16(p.ln_gamma) || (p.ln_dz && !al16(p.ln_dz)) || !p.ln_part || !p.ln_mean || !p.ln_rstd))
      return LOTUS_GEMM_DMA_NA;
    if (layout == 2 && gemm_dma_wgrad_splits(p.K, p.M, p.N) != nz) return LOTUS_GEMM_DMA_NA;  // (split plan of another kernel)
    const bool kc_any = layout != 2;
    if (kc_any && ((p.K % bk) || (nz > 1 && (p.klen % bk)))) return LOTUS_GEMM_DMA_NA;
    if (layout == 2 && ((p.M & 3) || (nz > 1 && (p.klen % 32)))) return LOTUS_GEMM_DMA_NA;
    dim3 block(256);
#define DMA_GO(BM, BN, BK, NST, XKC, WKC, SUMA)                                                                         \
  do {                                                                                                                  \
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), nz);                                                                        \
    if (epi) LOTUS_LAUNCH((gemm_dma_kernel<BM, BN, BK, NST, XKC, WKC, SUMA, 1>), grid, block, 0, st, p);                \
    else LOTUS_LAUNCH((gemm_dma_kernel<BM, BN, BK, NST, XKC, WKC, SUMA, 0>), grid, block, 0, st, p);                    \
  } while (0)
    // (Measured and removed: four stages = 64 KB = two blocks per CU, to leave LDS for the other queue's blocks: 941 / 944 / 944
    // against 944 / 946 / 947 samples/s with three stages.)
    if (layout == 0) {
      if (wide) DMA_GO(128, 128, 16, 3, true, true, false);
      else DMA_GO(128, 64, 32, 2, true, true, false);
    } else if (layout == 1 && ln) {
      dim3 grid(1, cdiv(p.M, 128), 1);
      if (wide) LOTUS_LAUNCH((gemm_dma_kernel<128, 128, 16, 3, true, false, false, 2>), grid, block, 0, st, p);
      else LOTUS_LAUNCH((gemm_dma_kernel<128, 64, 32, 2, true, false, false, 2>), grid, block, 0, st, p);
    } else if (layout == 1) {
      if (wide) DMA_GO(128, 128, 16, 3, true, false, false);
      else DMA_GO(128, 64, 32, 2, true, false, false);
    } else {  // weight gradient: rows of dW = p.M, columns = p.N, reduction over the activation rows
      const bool tall = p.M > 64;
      if (tall && wide) DMA_GO(128, 128, 16, 3, false, false, true);
      else if (tall) DMA_GO(128, 64, 32, 2, false, false, true);
      else if (wide) DMA_GO(64, 128, 32, 2, false, false, true);
      else DMA_GO(64, 64, 32, 3, false, false, true);
    }
#undef DMA_GO
    LOTUS_LAUNCH_CHECK("lotus_gemm(dma)");
    return LOTUS_OK;
  }
}

// The tap-grouped sparse convolution on the LDS-DMA tiles (p as lotus_conv_tap_gemm sets it up: a_rows / tap_cnt / tap_rows /
// b_tap_*, a_src_rows; layout 0 = forward, weights k-contiguous; 1 = input gradient).  The grid covers the worst case — every
// row of every tap active; the host does not know the pair counts — and its blocks past the last active tile leave after the
// scan of the 27 counters.
int launch_gemm_dma_tap(GemmP& p, int layout, hipStream_t st) {
  if constexpr (LOTUS_ACT_IS_BF16) {
    return LOTUS_GEMM_DMA_NA;
  } else {
    static int on = -1;
    if (on < 0) on = dma_env("LOTUS_GEMM_DMA", 1);
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    const bool wide = p.N > 64;
    const int bk = wide ? 16 : 32;
    if (!on || p.prec != 0 || !p.a_rows || !p.tap_cnt || p.tap_rows <= 0 || (p.tap_rows & 63) || p.a_src_rows <= 0 || !al16(p.A) ||
        !al16(p.B) || !al16(p.C) || (p.lda & 3) || (p.ldb & 3) || (p.ldc & 3) || (p.N & 3) || (p.K % bk) || (p.b_tap_stride & 3) ||
        (long)p.a_src_rows * p.lda * 4 >= 0x7ffffff0L || (long)p.K * p.ldb * 4 >= (1L << 32) || (long)p.N * p.ldb * 4 >= (1L << 32) ||
        (long)p.M * p.ldc * 4 >= (1L << 31) || p.bias || p.residual || p.pre || p.mulpre || p.act != LOTUS_ACT_NONE || p.drop_thresh)
      return LOTUS_GEMM_DMA_NA;
    p.klen = p.K;
    const dim3 grid(cdiv(p.N, wide ? 128 : 64), 27 * cdiv(p.tap_rows, 128)), block(256);
    if (layout == 0) {
      if (wide) LOTUS_LAUNCH((gemm_dma_tap_kernel<128, 128, 16, 3, true>), grid, block, 0, st, p);
      else LOTUS_LAUNCH((gemm_dma_tap_kernel<128, 64, 32, 2, true>), grid, block, 0, st, p);
    } else {
      if (wide) LOTUS_LAUNCH((gemm_dma_tap_kernel<128, 128, 16, 3, false>), grid, block, 0, st, p);
      else LOTUS_LAUNCH((gemm_dma_tap_kernel<128, 64, 32, 2, false>), grid, block, 0, st, p);
    }
    LOTUS_LAUNCH_CHECK("lotus_subm_conv(tap-grouped, dma)");
    return LOTUS_OK;
  }
}

// Split count of a weight gradient dW[N, K] over M activation rows when it runs on these kernels (0: it does not).  Known from
// the shape alone (the workspace query has no pointers): ~1 block per CU on 128 x 128 (64-wide where a side is 64) output
// tiles, at least 256 rows per split, a multiple of 8 (one XCD / L2 per split, see the kernel's block order).
int gemm_dma_wgrad_splits(int M, int N, int K) {
  if constexpr (LOTUS_ACT_IS_BF16) {
    return 0;
  } else {
    static int on = -1, minrows = 0, target = 0;
    if (on < 0) {
      on = dma_env("LOTUS_GEMM_DMA", 1);
      minrows = dma_env("LOTUS_GEMM_DMA_MINROWS", 16384);
      target = dma_env("LOTUS_GEMM_DMA_WGRAD_BLOCKS", 256);  // 0: weight gradients stay on gemm_kernel
    }
    if (!on || target <= 0 || M < minrows || (N & 3) || (K & 3)) return 0;
    const long tiles = (long)cdiv(N, N > 64 ? 128 : 64) * cdiv(K, K > 64 ? 128 : 64);
    long nz = (target + tiles - 1) / tiles;
    nz = nz >= 8 ? (nz / 8) * 8 : nz;
    if (nz < 1) nz = 1;
    // Measured (tools/dbg/dma_onoff.py): with fewer than ~256 rows per split a block is a cold prologue and a handful of
    // slabs — 23 894 x 128 x 128 (one 128 x 128 tile): 33 us here against 20.8 us on gemm_kernel's 64 x 64 tiles — those stay there
    return M / nz >= 256 ? (int)nz : 0;
  }
}

}  // namespace LOTUS_NS
