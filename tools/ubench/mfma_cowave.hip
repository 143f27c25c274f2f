// Micro-benchmark: do the MFMAs of one wave overlap with the VALU / LDS work of ANOTHER wave on the same SIMD?
// One block of 512 threads = 8 waves = 2 per SIMD.  Waves 0-3 run a dependent MFMA chain, waves 4-7 run `mode`.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int mode) {
  __shared__ float lds[8192];
  const int t = threadIdx.x, wave = t >> 6, l = t & 63;
  for (int i = t; i < 8192; i += 512) lds[i] = i;
  __syncthreads();
  f32x16 acc = {0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = l + i;
  const long long t0 = clock64();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((float)l, 1.f, acc, 0, 0, 0);
  } else if (mode == 1) {  // VALU only
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int s = 0; s < 64; ++s) v[s & 15] = v[s & 15] * 1.0001f + 0.5f;
  } else if (mode == 2) {  // LDS read-modify-write
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int a = (l + s * 64 + wave * 1024) & 8191;
        lds[a] = lds[a] + v[s];
      }
  } else if (mode == 3) {  // second MFMA chain
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, (float)l, acc, 0, 0, 0);
  }
  const long long t1 = clock64();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i] + v[i];
  out[t] = r;
  if (l == 0) cyc[wave] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 64);
  const char* names[] = {"idle", "VALU (64 fma / iter)", "LDS rmw (16 / iter)", "MFMA chain (16 / iter)"};
  for (int mode = 0; mode < 4; ++mode) {
    const int iters = 200;
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, iters, mode);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, iters, mode);
    (void)hipDeviceSynchronize();
    long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("co-wave %-24s MFMA wave: %7.1f cycles / 16 MFMA    co-wave: %7.1f cycles / iter\n", names[mode],
           (double)h[0] / iters, (double)h[4] / iters);
  }
  return 0;
}
