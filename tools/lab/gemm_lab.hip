// Kernel lab for the dense fp32 GEMM (round 5): times the LDS-DMA kernel variants of csrc/gemm_dma.h against the library's
// current entry points on the dense shapes of a training step, and checks every variant against a float64 product.
//   tools/lab/build.sh && tools/lab/gemm_lab tools/gemm_shapes.json [kind filter] [min M]
#include "../../robot-3dlotus_amd/csrc/gemm_dma.h"
#include <string>
#include <vector>
#include <algorithm>
#include <math.h>

using namespace lotus_f32;

extern "C" {
int lotus_linear_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, float* pre, int M, int N, int K,
                     int act, float drop_p, unsigned long long drop_seed, int precision, void* workspace, size_t workspace_bytes,
                     void* counters, void* stream);
int lotus_linear_dgrad(const float* dy, const float* w, float* dx, const float* pre, const float* add, int M, int N, int K, int act,
                       float drop_p, unsigned long long drop_seed, int precision, void* workspace, size_t workspace_bytes, void* counters,
                       void* stream);
int lotus_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, int accumulate, int precision,
                       void* workspace, size_t workspace_bytes, void* counters, void* stream);
size_t lotus_linear_wgrad_workspace(int M, int N, int K);
size_t lotus_linear_workspace(int M, int N, int K);
size_t lotus_splitk_counters_bytes(void);
const char* lotus_last_error(void);
}
void lotus_set_error(const char*, ...) {}
thread_local hipEvent_t lotus_tls_stop_event_lab = nullptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// C[i][j] = sum_k A(i,k) B(k,j) in float64
__global__ void ref_kernel(const float* A, long a_i, long a_k, const float* B, long b_k, long b_j, double* C, int M, int N, int K) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * N) return;
  const int i = (int)(idx / N), j = (int)(idx % N);
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)A[i * a_i + k * a_k] * (double)B[k * b_k + j * b_j];
  C[idx] = s;
}
__global__ void cmp_kernel(const float* C, const double* R, long n, double* out) {  // out[0] = max |diff|, out[1] = max |ref|
  __shared__ double sd[256], sr[256];
  double d = 0, r = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    d = fmax(d, fabs((double)C[i] - R[i])); r = fmax(r, fabs(R[i]));
  }
  sd[threadIdx.x] = d; sr[threadIdx.x] = r;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { sd[threadIdx.x] = fmax(sd[threadIdx.x], sd[threadIdx.x + o]); sr[threadIdx.x] = fmax(sr[threadIdx.x], sr[threadIdx.x + o]); } __syncthreads(); }
  if (threadIdx.x == 0) { atomicMax((unsigned long long*)&out[0], __double_as_longlong(sd[0])); atomicMax((unsigned long long*)&out[1], __double_as_longlong(sr[0])); }
}
__global__ void fill_kernel(float* p, long n, unsigned seed, float scale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned h = lotus_hash32(seed, (unsigned long long)i);
    p[i] = ((int)(h >> 8) - (1 << 23)) * (scale / (1 << 23));
  }
}
__global__ void sum_parts_kernel(const float* part, float* out, long n, long stride, int nz) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < nz; ++z) s += part[(long)z * stride + i];
  out[i] = s;
}

struct Variant { const char* name; int bm, bn, bk, nst; };
static const Variant kVariants[] = {
    {"128x128x16s2", 128, 128, 16, 2}, {"128x128x16s3", 128, 128, 16, 3}, {"128x128x32s2", 128, 128, 32, 2},
    {"128x64x16s3", 128, 64, 16, 3},   {"128x64x32s2", 128, 64, 32, 2},   {"128x64x32s3", 128, 64, 32, 3},
    {"64x128x16s3", 64, 128, 16, 3},   {"64x128x32s2", 64, 128, 32, 2},   {"64x128x32s3", 64, 128, 32, 3},
    {"64x64x32s2", 64, 64, 32, 2},     {"64x64x32s3", 64, 64, 32, 3},     {"64x64x16s3", 64, 64, 16, 3},
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

template <bool XKC, bool WKC, bool SUM_A>
static void launch_variant(int v, GemmP& p, int nz, hipStream_t st) {
  const Variant& V = kVariants[v];
  dim3 grid((p.N + V.bn - 1) / V.bn, (p.M + V.bm - 1) / V.bm, nz), block(256);
#define GO(BM, BN, BK, NST) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, BK, NST, XKC, WKC, SUM_A, 0>), grid, block, 0, st, p)
  switch (v) {
    case 0: GO(128, 128, 16, 2); break;
    case 1: GO(128, 128, 16, 3); break;
    case 2: GO(128, 128, 32, 2); break;
    case 3: GO(128, 64, 16, 3); break;
    case 4: GO(128, 64, 32, 2); break;
    case 5: GO(128, 64, 32, 3); break;
    case 6: GO(64, 128, 16, 3); break;
    case 7: GO(64, 128, 32, 2); break;
    case 8: GO(64, 128, 32, 3); break;
    case 9: GO(64, 64, 32, 2); break;
    case 10: GO(64, 64, 32, 3); break;
    case 11: GO(64, 64, 16, 3); break;
  }
#undef GO
}

template <typename F>
static float time_us(F fn, hipStream_t st, int reps = 3, int iters = 10) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipStreamSynchronize(st));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 1e3f / iters);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return best;
}

struct Shape { std::string kind; int M, N, K, count; };

template <int ABL>
static void launch_abl(int v, GemmP& p, hipStream_t st) {  // forward layout
  dim3 block(256);
  if (v == 0) hipLaunchKernelGGL((gemm_dma_kernel<128, 128, 16, 3, true, true, false, 0, ABL>), dim3((p.N + 127) / 128, (p.M + 127) / 128), block, 0, st, p);
  else if (v == 1) hipLaunchKernelGGL((gemm_dma_kernel<64, 128, 32, 2, true, true, false, 0, ABL>), dim3((p.N + 127) / 128, (p.M + 63) / 64), block, 0, st, p);
  else if (v == 2) hipLaunchKernelGGL((gemm_dma_kernel<64, 64, 32, 3, true, true, false, 0, ABL>), dim3((p.N + 63) / 64, (p.M + 63) / 64), block, 0, st, p);
  else hipLaunchKernelGGL((gemm_dma_kernel<128, 128, 32, 2, true, true, false, 0, ABL>), dim3((p.N + 127) / 128, (p.M + 127) / 128), block, 0, st, p);
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: gemm_lab shapes.json [kind] [minM] [check=1]\n"); return 1; }
  FILE* f = fopen(argv[1], "r");
  if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
  std::string txt; char buf[4096]; size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) txt.append(buf, n);
  fclose(f);
  std::vector<Shape> shapes;
  for (size_t pos = 0; (pos = txt.find("[\"", pos)) != std::string::npos; ++pos) {
    char kind[16]; int M, N, K, c;
    if (sscanf(txt.c_str() + pos, "[\"%15[a-z]\", %d, %d, %d, %d]", kind, &M, &N, &K, &c) == 5) shapes.push_back({kind, M, N, K, c});
  }
  const char* kfilter = argc > 2 ? argv[2] : "all";
  const int minM = argc > 3 ? atoi(argv[3]) : 0;
  const int check = argc > 4 ? atoi(argv[4]) : 1;
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t maxel = (size_t)65536 * 3072;  // (also 262144 x 768)
  float *x, *w, *dy, *out, *out2, *ws; double *ref, *cmp; void* counters;
  CK(hipMalloc(&x, maxel * 4)); CK(hipMalloc(&dy, maxel * 4)); CK(hipMalloc(&w, (size_t)3072 * 3072 * 4)); CK(hipMalloc(&out, maxel * 4));
  CK(hipMalloc(&out2, maxel * 4)); CK(hipMalloc(&ref, maxel * 8)); CK(hipMalloc(&cmp, 16));
  const size_t ws_bytes = (size_t)1 << 30;
  CK(hipMalloc(&ws, ws_bytes));
  CK(hipMalloc(&counters, lotus_splitk_counters_bytes())); CK(hipMemset(counters, 0, lotus_splitk_counters_bytes()));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, x, (long)maxel, 1u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, dy, (long)maxel, 2u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, w, (long)3072 * 3072, 3u, 0.05f);
  CK(hipStreamSynchronize(st));

  if (!strcmp(kfilter, "abl")) {  // where the time of a forward product goes: ablations of the store / DMA halves
    const int shp[][3] = {{65536, 128, 512}, {65536, 512, 128}, {65536, 128, 128}, {23894, 512, 128}, {6077, 1024, 256}, {262144, 512, 128}};
    const char* vn[4] = {"128x128x16s3", "64x128x32s2", "64x64x32s3", "128x128x32s2"};
    printf("# M N K | variant: full / no-store / no-dma / neither (us)\n");
    for (auto& sh : shp) {
      if ((size_t)sh[0] * std::max(sh[1], sh[2]) > maxel) continue;
      GemmP p; memset(&p, 0, sizeof(p));
      p.drop_inv_keep = 1.f; p.a_vec = p.b_vec = 1;
      p.A = x; p.B = w; p.C = out; p.M = sh[0]; p.N = sh[1]; p.K = sh[2]; p.lda = sh[2]; p.ldb = sh[2]; p.ldc = sh[1]; p.klen = sh[2];
      printf("%d %d %d |", sh[0], sh[1], sh[2]);
      for (int v = 0; v < 4; ++v) {
        const float t0 = time_us([&]() { launch_abl<0>(v, p, st); }, st), t1 = time_us([&]() { launch_abl<1>(v, p, st); }, st);
        const float t2 = time_us([&]() { launch_abl<2>(v, p, st); }, st), t3 = time_us([&]() { launch_abl<3>(v, p, st); }, st);
        printf(" %s: %.1f / %.1f / %.1f / %.1f |", vn[v], t0, t1, t2, t3);
      }
      printf("  (MFMA roof %.1f us)\n   with the saved pre-activation / with a residual (128x128x16s3):", 2.0 * sh[0] * sh[1] * sh[2] / 157.3e6);
      { GemmP q = p; q.pre = out2; printf(" %.1f", time_us([&]() { launch_abl<0>(0, q, st); }, st)); }
      { GemmP q = p; q.residual = out2; printf(" / %.1f\n", time_us([&]() { launch_abl<0>(0, q, st); }, st)); }
    }
    return 0;
  }
  if (!strcmp(kfilter, "trace")) {  // per-block phase timestamps of one forward product -> stdout (tools/lab/trace_summary.py)
    const int M = argc > 3 ? atoi(argv[3]) : 65536, N = argc > 4 ? atoi(argv[4]) : 512, K = argc > 5 ? atoi(argv[5]) : 128;
    GemmP p; memset(&p, 0, sizeof(p));
    p.drop_inv_keep = 1.f; p.a_vec = p.b_vec = 1;
    p.A = x; p.B = w; p.C = out; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.klen = K;
    const int nb = ((N + 127) / 128) * ((M + 127) / 128);
    long long* dbg; CK(hipMalloc(&dbg, (size_t)nb * 64)); CK(hipMemset(dbg, 0, (size_t)nb * 64));
    p.bias_part = (float*)dbg;
    for (int r = 0; r < 3; ++r) launch_abl<8>(0, p, st);
    CK(hipStreamSynchronize(st));
    std::vector<long long> h((size_t)nb * 8);
    CK(hipMemcpy(h.data(), dbg, (size_t)nb * 64, hipMemcpyDeviceToHost));
    printf("# block wall100MHz t_start t_slab0 t_loop_end t_stores_issued t_stores_done hwid xcc (M %d N %d K %d, 128x128x16s3)\n", M, N, K);
    for (int b = 0; b < nb; ++b) printf("%d %lld %lld %lld %lld %lld %lld %lld %lld\n", b, h[b * 8], h[b * 8 + 1], h[b * 8 + 2], h[b * 8 + 3], h[b * 8 + 4], h[b * 8 + 5], h[b * 8 + 6], h[b * 8 + 7]);
    return 0;
  }
  std::vector<double> tot_base(3, 0.0), tot_best(3, 0.0);
  std::vector<std::vector<double>> tot_var(3, std::vector<double>(kNumVariants, 0.0));
  printf("# kind M N K count | base_us | best variant us | per-variant us (x = wrong, - = not applicable)\n# variants:");
  for (int v = 0; v < kNumVariants; ++v) printf(" %s", kVariants[v].name);
  printf("\n");
  for (const Shape& s : shapes) {
    if (strcmp(kfilter, "all") && s.kind != kfilter) continue;
    if (s.M < minM) continue;
    const int kd = s.kind == "fwd" ? 0 : s.kind == "dgrad" ? 1 : 2;
    GemmP p; memset(&p, 0, sizeof(p));
    p.drop_inv_keep = 1.f; p.a_vec = p.b_vec = 1;
    long a_i, a_k, b_k, b_j;
    if (kd == 0) { p.A = x; p.B = w; p.M = s.M; p.N = s.N; p.K = s.K; p.lda = s.K; p.ldb = s.K; p.ldc = s.N; a_i = s.K; a_k = 1; b_k = 1; b_j = s.K; }
    else if (kd == 1) { p.A = dy; p.B = w; p.M = s.M; p.N = s.K; p.K = s.N; p.lda = s.N; p.ldb = s.K; p.ldc = s.K; a_i = s.N; a_k = 1; b_k = s.K; b_j = 1; }
    else { p.A = dy; p.B = x; p.M = s.N; p.N = s.K; p.K = s.M; p.lda = s.N; p.ldb = s.K; p.ldc = s.K; a_i = 1; a_k = s.N; b_k = s.K; b_j = 1; }
    p.C = out;
    const bool aligned = (p.K % 32 == 0 || kd == 2) && p.N % 4 == 0 && (p.M % 4 == 0 || kd != 2) && p.lda % 4 == 0 && p.ldb % 4 == 0;
    const long nout = (long)p.M * p.N;
    if (check) {
      hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, (const float*)p.A, a_i, a_k, (const float*)p.B, b_k, b_j, ref, p.M, p.N, p.K);
    }
    // baseline: the library entry point
    auto base = [&]() {
      int rc;
      if (kd == 0) rc = lotus_linear_fwd(x, w, nullptr, nullptr, out2, nullptr, s.M, s.N, s.K, 0, 0.f, 0, 0, ws, ws_bytes, counters, st);
      else if (kd == 1) rc = lotus_linear_dgrad(dy, w, out2, nullptr, nullptr, s.M, s.N, s.K, 0, 0.f, 0, 0, ws, ws_bytes, counters, st);
      else rc = lotus_linear_wgrad(dy, x, out2, nullptr, s.M, s.N, s.K, 0, 0, ws, ws_bytes, counters, st);
      if (rc) { printf("library call failed: %s\n", lotus_last_error()); exit(1); }
    };
    const float tb = time_us(base, st);
    // split count of the lab's weight gradient: ~ 2 blocks per CU, >= 256 rows per split
    printf("%s %d %d %d %d | %.1f |", s.kind.c_str(), s.M, s.N, s.K, s.count, tb);
    float best = 1e30f; int bestv = -1;
    std::vector<float> tv(kNumVariants, -1.f);
    for (int v = 0; v < kNumVariants && aligned; ++v) {
      const Variant& V = kVariants[v];
      if ((kd != 2 && p.K % V.bk) || (p.N < V.bn / 2) || (p.M < V.bm / 2)) continue;
      int nz = 1;
      GemmP q = p;
      q.klen = ((p.K + 63) / 64) * 64;
      if (kd == 2) {
        const long tiles = (long)((p.M + V.bm - 1) / V.bm) * ((p.N + V.bn - 1) / V.bn);
        while (nz < 512 && tiles * nz < 512 && p.K / (nz * 2) >= 128) nz *= 2;
        q.klen = (((p.K + nz - 1) / nz + 63) / 64) * 64;
        if (nz > 1) { q.C = ws; q.part_stride = nout; }
      } else {
        const long tiles = (long)((p.M + V.bm - 1) / V.bm) * ((p.N + V.bn - 1) / V.bn);
        while (nz < 16 && tiles * nz < 256 && p.K / (nz * 2) >= 256) nz *= 2;
        q.klen = (((p.K + nz - 1) / nz + 63) / 64) * 64;
        if (nz > 1) { q.C = ws; q.part_stride = nout; }
      }
      auto run = [&]() {
        if (kd == 0) launch_variant<true, true, false>(v, q, nz, st);
        else if (kd == 1) launch_variant<true, false, false>(v, q, nz, st);
        else launch_variant<false, false, false>(v, q, nz, st);
        if (nz > 1) hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, (const float*)ws, out, nout, nout, nz);
      };
      bool ok = true;
      if (check) {
        CK(hipMemsetAsync(out, 0xff, nout * 4, st));
        run();
        CK(hipMemsetAsync(cmp, 0, 16, st));
        hipLaunchKernelGGL(cmp_kernel, dim3(1024), dim3(256), 0, st, (const float*)out, (const double*)ref, nout, cmp);
        double hc[2]; CK(hipMemcpyAsync(hc, cmp, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        ok = hc[0] <= 2e-5 * fmax(hc[1], 1e-30) && hc[0] == hc[0];
        if (!ok) printf(" [%s err %.3g / %.3g]", V.name, hc[0], hc[1]);
      }
      const float t = time_us(run, st);
      tv[v] = ok ? t : -2.f;
      if (ok && t < best) { best = t; bestv = v; }
    }
    printf(" %s %.1f |", bestv >= 0 ? kVariants[bestv].name : "none", bestv >= 0 ? best : 0.f);
    for (int v = 0; v < kNumVariants; ++v) { if (tv[v] == -1.f) printf(" -"); else if (tv[v] == -2.f) printf(" x"); else printf(" %.1f", tv[v]); }
    printf("\n"); fflush(stdout);
    tot_base[kd] += (double)tb * s.count;
    tot_best[kd] += (double)(bestv >= 0 ? std::min(best, tb) : tb) * s.count;
    for (int v = 0; v < kNumVariants; ++v) tot_var[kd][v] += (double)(tv[v] > 0 ? std::min(tv[v], tb) : tb) * s.count;
  }
  const char* kn[3] = {"fwd", "dgrad", "wgrad"};
  for (int kd = 0; kd < 3; ++kd) {
    printf("TOTAL %s: base %.3f ms, best-of-variants %.3f ms; single variant (min with base):", kn[kd], tot_base[kd] / 1e3, tot_best[kd] / 1e3);
    for (int v = 0; v < kNumVariants; ++v) printf(" %.3f", tot_var[kd][v] / 1e3);
    printf("\n");
  }
  return 0;
}
