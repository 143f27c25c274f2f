// Micro-benchmark (round 5): how many instructions of a co-resident wave issue per fp32 MFMA of another wave on the same
// SIMD, by MFMA form (32x32x2 dependent chain / four independent accumulators / 16x16x4) and by co-instruction class.
// One block of 512 threads = 8 waves = 2 per SIMD; waves 0-3 run MFMAs, waves 4-7 run the co-instructions.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <int MF, int CO>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, const float* gsrc, int iters, int iters_co) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;
  for (int i = t; i < 16384; i += 512) lds[i] = i;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = l + i;
  float4 ld[8];
  for (int i = 0; i < 8; ++i) ld[i] = make_float4(0, 0, 0, 0);
  int sacc = iters;
  const long long t0 = clock64();
  if (wave < 4) {
    const float x = (float)l, y = 1.f + l;
    const bf16x8 bx = {(__bf16)1.f, (__bf16)2.f, (__bf16)x, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      if (MF == 0) {  // one accumulator: every MFMA waits for the previous one
#pragma unroll
        for (int s = 0; s < 16; ++s) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      } else if (MF == 1) {  // four independent accumulators
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
      } else if (MF == 2) {  // 16x16x4, four independent accumulators (32 per iteration = the flops of 16 32x32x2... /2)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c3, 0, 0, 0);
        }
      } else if (MF == 3) {  // bf16 32x32x16, four accumulators, 16 per iteration
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, bx, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, bx, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, bx, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, bx, a3, 0, 0, 0);
        }
      }
    }
  } else {
    for (int it = 0; it < iters_co; ++it) {
      if (CO == 1) {  // 64 independent-ish VALU fma
#pragma unroll
        for (int s = 0; s < 64; ++s) v[s & 15] = v[s & 15] * 1.0001f + 0.5f;
      } else if (CO == 2) {  // 64 ds_read_b128 (results consumed at the end of the iteration)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
          for (int s = 0; s < 8; ++s) ld[s] = *reinterpret_cast<const float4*>(&lds[((l + g * 8 + s) * 4 + (wave & 3) * 4096) & 16380]);
#pragma unroll
          for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(ld[s].x), "v"(ld[s].w));
        }
      } else if (CO == 3) {  // 64 SALU
#pragma unroll
        for (int s = 0; s < 64; ++s) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
      } else if (CO == 4) {  // 64 global loads (16 B per lane, L2 / L1 hits)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
          for (int s = 0; s < 8; ++s) ld[s] = *reinterpret_cast<const float4*>(gsrc + ((l + (g * 8 + s) * 64) & 4095) * 4);
#pragma unroll
          for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(ld[s].x), "v"(ld[s].w));
        }
      } else if (CO == 5) {  // 64 v_mov
#pragma unroll
        for (int s = 0; s < 64; ++s) asm volatile("v_mov_b32 %0, %1" : "=v"(v[s & 15]) : "v"(v[(s + 1) & 15]));
      }
    }
  }
  const long long t1 = clock64();
  float r = sacc;
  for (int i = 0; i < 16; ++i) r += a0[i] + a1[i] + a2[i] + a3[i] + v[i];
  for (int i = 0; i < 4; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
  out[t] = r;
  if (l == 0) cyc[wave] = t1 - t0;
}

// the MFMA wave itself carries NV VALU (NS SALU) instructions per MFMA (four accumulators, 16 MFMA per iteration); no co-wave
template <int NV, int NS>
__global__ __launch_bounds__(256) void k_own(float* out, long long* cyc, int iters) {
  const int t = threadIdx.x, l = t & 63;
  f32x16 a[4] = {{0}, {0}, {0}, {0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = l + i;
  int sacc = iters;
  const float x = (float)l, y = 1.f + l;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      a[s & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[s & 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(s * NV + q) & 7]) : "v"(y));
#pragma unroll
      for (int q = 0; q < NS; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
    }
  }
  const long long t1 = clock64();
  float r = sacc;
  for (int i = 0; i < 16; ++i) r += a[0][i] + a[1][i] + a[2][i] + a[3][i];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[t] = r;
  if (l == 0) cyc[t >> 6] = t1 - t0;
}
template <int NV, int NS>
static void run_own(float* out, long long* cyc) {
  hipLaunchKernelGGL((k_own<NV, NS>), dim3(1), dim3(256), 0, 0, out, cyc, 200);
  hipLaunchKernelGGL((k_own<NV, NS>), dim3(1), dim3(256), 0, 0, out, cyc, 200);
  (void)hipDeviceSynchronize();
  long long h[4];
  (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("own wave: %d VALU + %d SALU per MFMA: %.1f cycles per MFMA\n", NV, NS, (double)h[0] / (200 * 16));
}

template <int MF, int CO>
static void run(float* out, long long* cyc, const float* gsrc) {
  const char* mf[] = {"32x32x2 dependent (16/iter)", "32x32x2 x 4 acc (16/iter)", "16x16x4 x 4 acc (32/iter)", "bf16 32x32x16 x 4 acc (16/iter)"};
  const char* co[] = {"idle", "VALU fma", "ds_read_b128", "SALU add", "global_load x4", "v_mov"};
  printf("%-34s + %-15s", mf[MF], co[CO]);
  const int cfg[4][2] = {{200, 200}, {0, 200}, {200, 50}, {200, 800}};  // iterations of the MFMA waves / of the co-waves
  for (auto& c : cfg) {
    hipLaunchKernelGGL((k<MF, CO>), dim3(1), dim3(512), 0, 0, out, cyc, gsrc, c[0], c[1]);
    hipLaunchKernelGGL((k<MF, CO>), dim3(1), dim3(512), 0, 0, out, cyc, gsrc, c[0], c[1]);
    (void)hipDeviceSynchronize();
    long long h[8];
    (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf(" | %dx16 MFMA %6lld cyc, %dx64 co %6lld cyc", c[0], h[0], c[1], h[4]);
  }
  printf("\n");
}

int main() {
  float *out, *gsrc; long long* cyc;
  (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64); (void)hipMalloc(&gsrc, 65536 * 4);
  (void)hipMemset(gsrc, 0, 65536 * 4);
#define ROW(MF) run<MF, 0>(out, cyc, gsrc); run<MF, 1>(out, cyc, gsrc); run<MF, 2>(out, cyc, gsrc); run<MF, 3>(out, cyc, gsrc); run<MF, 4>(out, cyc, gsrc); run<MF, 5>(out, cyc, gsrc);
  run<1, 0>(out, cyc, gsrc); run<1, 1>(out, cyc, gsrc); run<1, 3>(out, cyc, gsrc); run<1, 5>(out, cyc, gsrc); run<1, 2>(out, cyc, gsrc);
  run<3, 1>(out, cyc, gsrc); run<3, 5>(out, cyc, gsrc);
  run_own<0, 0>(out, cyc); run_own<1, 0>(out, cyc); run_own<2, 0>(out, cyc); run_own<4, 0>(out, cyc); run_own<8, 0>(out, cyc); run_own<16, 0>(out, cyc);
  run_own<0, 2>(out, cyc); run_own<0, 8>(out, cyc); run_own<4, 8>(out, cyc);
  return 0;
}
