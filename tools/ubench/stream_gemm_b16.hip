// Micro-benchmark (diagnostic, not part of the library): Y[M][N] = X[M][K] W[N][K]^T with bf16 storage and bf16 MFMA, written as a
// STREAMING kernel — persistent blocks, the whole weight slice resident in LDS, X row tiles staged through LDS in full rows —
// to find out what the tall dense layers of the bf16 mode can reach (the library's 64 x 64 x 32 tile kernel: 0.35 of the byte roof).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/stream_gemm_b16.hip -o /tmp/sg && /tmp/sg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <string.h>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef unsigned short u16;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// LDS images: rows of K bf16 (K * 2 bytes, a multiple of 128), 16-byte chunks XOR-swizzled with the row so that the 16 lanes of a
// ds_read_b128 group (16 consecutive rows, same chunk) land on distinct 16-byte slots.
template <int K>
__device__ __forceinline__ int swz(int row, int chunk) {
  return row * (K / 8) + (chunk ^ (row & ((K / 8 < 16 ? K / 8 : 16) - 1)));  // uint4 index
}

// BM rows per tile, BN columns per block (the whole N when N <= 256), 256 threads: wave w owns columns [w * BN / 4, (w + 1) * BN / 4)
template <int K, int BN, int BM, int DEPTH, int DIRECT = 0>
__global__ __launch_bounds__(256) void stream_gemm(const u16* __restrict__ X, const u16* __restrict__ W, u16* __restrict__ Y, int M, int N) {
  constexpr int KC = K / 8;          // 16-byte chunks per row
  constexpr int WN = BN / 4;         // columns per wave
  constexpr int TN = WN / 32, TM = BM / 32;
  constexpr int XV = BM * KC / 256;  // uint4 per thread and X tile
  constexpr int WV = BN * KC / 256;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* Ws = smem;                  // [BN][KC]
  uint4* Xs = Ws + BN * KC;          // [BM][KC]
  uint4* Cs = Xs + BM * KC;          // [BM][BN / 8]  output tile (bf16), row-contiguous
  const int tid = threadIdx.x, wave = tid >> 6, l31 = tid & 31, hh = (tid >> 5) & 1;
  const int n0 = blockIdx.y * BN;
  const uint4* X4 = reinterpret_cast<const uint4*>(X);
  const uint4* W4 = reinterpret_cast<const uint4*>(W);
  // weights: once per block
#pragma unroll
  for (int j = 0; j < WV; ++j) {
    const int i = tid + j * 256, row = i / KC, c = i % KC;
    Ws[swz<K>(row, c)] = W4[(long)(n0 + row) * KC + c];
  }
  const int ntiles = (M + BM - 1) / BM;
  uint4 xr[DEPTH][XV];
  auto gload = [&](uint4 (&dst)[XV], int tile) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + j * 256, row = i / KC, c = i % KC;
      const int m = min(tile * BM + row, M - 1);
      dst[j] = X4[(long)m * KC + c];
    }
  };
  auto lstore = [&](const uint4 (&src)[XV]) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int i = tid + j * 256, row = i / KC, c = i % KC;
      Xs[swz<K>(row, c)] = src[j];
    }
  };
  int tile = blockIdx.x;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (tile + d * (int)gridDim.x < ntiles) gload(xr[d], tile + d * gridDim.x);
  __syncthreads();
  for (; tile < ntiles; tile += DEPTH * gridDim.x) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = tile + d * gridDim.x;
      if (t < ntiles) {
        lstore(xr[d]);
        __syncthreads();
        if (t + DEPTH * (int)gridDim.x < ntiles) gload(xr[d], t + DEPTH * gridDim.x);
        // products, transposed: D[n][m] = sum_k W[n][k] X[m][k]  (A operand = weights, B operand = rows): a lane owns one row m
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
        for (int s = 0; s < K / 16; ++s) {
          uint4 wf[TN], xf[TM];
#pragma unroll
          for (int b = 0; b < TN; ++b) wf[b] = Ws[swz<K>(wave * WN + b * 32 + l31, 2 * s + hh)];
#pragma unroll
          for (int a = 0; a < TM; ++a) xf[a] = Xs[swz<K>(a * 32 + l31, 2 * s + hh)];
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[b]), __builtin_bit_cast(bf16x8, xf[a]), acc[a][b], 0, 0, 0);
        }
        if constexpr (DIRECT) {
          // 8-byte pieces straight from the accumulators: lanes l and l + 32 write 16 contiguous bytes of one row
#pragma unroll
          for (int a = 0; a < TM; ++a) {
            const int m = t * BM + a * 32 + l31;
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int col = n0 + wave * WN + b * 32 + 8 * q + 4 * hh;
                const uint2 v = make_uint2(pack2(acc[a][b][4 * q], acc[a][b][4 * q + 1]), pack2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]));
                if (m < M) {
                  if (DIRECT == 2) { typedef unsigned u32x2 __attribute__((ext_vector_type(2))); u32x2 vv = {v.x, v.y}; __builtin_nontemporal_store(vv, reinterpret_cast<u32x2*>(Y + (long)m * N + col)); }
                  else *reinterpret_cast<uint2*>(Y + (long)m * N + col) = v;
                }
              }
          }
          __syncthreads();  // the X image is overwritten by the next tile
        } else {
        // lane (m = a * 32 + l31): registers 4q .. 4q+3 are columns n = b * 32 + 8 q + 4 hh + 0..3 of its row -> 8-byte pieces
        uint2* C2 = reinterpret_cast<uint2*>(Cs);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = a * 32 + l31, col = wave * WN + b * 32 + 8 * q + 4 * hh;
              C2[row * (BN / 4) + ((col / 4) ^ ((row & 7) * 2))] =
                  make_uint2(pack2(acc[a][b][4 * q], acc[a][b][4 * q + 1]), pack2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]));
            }
        __syncthreads();
        // coalesced row stores (16 bytes per lane); the XOR above permutes 16-byte slots within a row: undo it here
        constexpr int CV = BM * (BN / 8) / 256;
#pragma unroll
        for (int j = 0; j < CV; ++j) {
          const int i = tid + j * 256, row = i / (BN / 8), c = i % (BN / 8);
          const int m = t * BM + row;
          if (m < M) reinterpret_cast<uint4*>(Y)[(long)m * (N / 8) + n0 / 8 + c] = Cs[row * (BN / 8) + (c ^ (row & 7))];
        }
        }
      }
    }
  }
}

static float bf2f(u16 v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static u16 f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }

template <int K, int BN, int BM, int DEPTH, int DIRECT = 0>
static void run(int M, int N, int bpc, const char* tag) {
  std::vector<u16> hx((size_t)M * K), hw((size_t)N * K);
  for (auto& v : hx) v = f2bf((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hw) v = f2bf((rand() % 2001 - 1000) / 4000.f);
  u16 *x, *w, *y;
  (void)hipMalloc(&x, hx.size() * 2); (void)hipMalloc(&w, hw.size() * 2); (void)hipMalloc(&y, (size_t)M * N * 2);
  (void)hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  const size_t sm = (size_t)(BN * K / 8 + BM * K / 8 + (DIRECT ? 0 : BM * BN / 8)) * 16;
  (void)hipFuncSetAttribute((const void*)stream_gemm<K, BN, BM, DEPTH, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  const int ntiles = (M + BM - 1) / BM;
  dim3 grid(std::min(ntiles, 256 * bpc), N / BN);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_gemm<K, BN, BM, DEPTH, DIRECT>), grid, dim3(256), sm, 0, x, w, y, M, N);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_gemm<K, BN, BM, DEPTH, DIRECT>), grid, dim3(256), sm, 0, x, w, y, M, N);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20, bytes = 2.0 * ((double)M * K + (double)M * N) + 2.0 * N * K;
  std::vector<u16> hy((size_t)M * N);
  (void)hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int it = 0; it < 2000; ++it) {
    const int m = (int)((long)rand() * 7919 % M), n = rand() % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
    worst = std::max(worst, fabs(ref - bf2f(hy[(size_t)m * N + n])) / (fabs(ref) + 0.05));
  }
  printf("%-28s M=%7d N=%4d K=%4d  lds %3zu KB grid %4d x %d  %7.1f us  %.2f TB/s (%.2f of 6.3)  err %.1e %s\n", tag, M, N, K, sm / 1024,
         grid.x, grid.y, us, bytes / us / 1e6, bytes / us / 6.3e6, worst, worst < 2e-2 ? "ok" : "WRONG");
  (void)hipFree(x); (void)hipFree(w); (void)hipFree(y);
}

int main() {
  for (int M : {65536, 262144}) {
    run<128, 128, 64, 1>(M, 128, 1, "BM64 d1 1/CU");
    run<128, 128, 64, 1>(M, 128, 2, "BM64 d1 2/CU");
    run<128, 128, 64, 1, 1>(M, 128, 2, "BM64 d1 2/CU direct");
    run<128, 128, 64, 1, 1>(M, 128, 3, "BM64 d1 3/CU direct");
    run<128, 128, 64, 1, 2>(M, 128, 2, "BM64 d1 2/CU direct nt");
    run<128, 128, 32, 1>(M, 128, 2, "BM32 d1 2/CU");
    run<128, 128, 32, 1, 1>(M, 128, 3, "BM32 d1 3/CU direct");
    run<128, 128, 32, 2, 1>(M, 128, 3, "BM32 d2 3/CU direct");
    run<128, 256, 64, 1>(M, 512, 1, "N512 BN256 BM64 d1 1/CU");
    run<128, 256, 64, 1, 1>(M, 512, 2, "N512 BN256 BM64 d1 2/CU direct");
    run<128, 256, 32, 1, 1>(M, 512, 2, "N512 BN256 BM32 d1 2/CU direct");
    run<128, 256, 64, 1, 2>(M, 512, 2, "N512 BN256 BM64 d1 2/CU direct nt");
  }
  return 0;
}
