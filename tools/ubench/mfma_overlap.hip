// Micro-benchmark: does a wave's VALU / LDS work issue in the shadow of its own dependent MFMA chain?
// One wave per block, one block: cycles per iteration of [16 dependent v_mfma_f32_32x32x2_f32 + filler].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = i;
  __syncthreads();
  f32x16 acc = {0}, acc2 = {0};
  float a = l, b = 1.f, v[16];
  for (int i = 0; i < 16; ++i) v[i] = l + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      if (MODE == 1) {  // 4 independent VALU ops per MFMA
        v[(s * 4) & 15] += 1.f; v[(s * 4 + 1) & 15] += 1.f; v[(s * 4 + 2) & 15] += 1.f; v[(s * 4 + 3) & 15] += 1.f;
      }
      if (MODE == 2) {  // 1 LDS read + 1 LDS write per MFMA
        v[s] += lds[(l + s * 64 + it) & 4095];
        lds[(l + s * 64) & 4095] = v[(s + 8) & 15];
      }
      if (MODE == 3) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);  // second independent chain
      if (MODE == 4) {  // 8 VALU per MFMA
#pragma unroll
        for (int q = 0; q < 8; ++q) v[(s * 8 + q) & 15] += 1.f;
      }
      SB;
    }
  }
  long long t1 = clock64();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i] + acc2[i] + v[i];
  out[l] = r;
  if (l == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int nblk) {
  float* out; long long* cyc;
  hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
  const int iters = 200;
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(64), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s %7.1f clock64 ticks per 16-MFMA iteration\n", name, (double)h / iters);
}
int main() {
  run<0>("16 dependent MFMA", 1);
  run<1>("+4 VALU each", 1);
  run<4>("+8 VALU each", 1);
  run<2>("+1 ds_read +1 ds_write each", 1);
  run<3>("2 independent chains (32 MFMA)", 1);
  return 0;
}
