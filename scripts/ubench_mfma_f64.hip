// v_mfma_f64_16x16x4_f64 on gfx950: (1) what it computes, bit for bit (is D = a chain of fused multiply-adds over k,
// and in which order?), (2) its lane layout, (3) what it costs a SIMD, alone and interleaved with independent VALU
// work (does the matrix pipe run beside the vector ALU for fp64 as it does for the low-precision forms?).
// Build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_mfma_f64.hip -o /tmp/ubench_mfma_f64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;

// ---- semantics: one wave, A 16x4, B 4x16, C 16x16 from memory (row-major), D back
__global__ void k_semantics(const double* A, const double* B, const double* C, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];    // A[i = l & 15][k = l >> 4]
  const double b = B[(l >> 4) * 16 + (l & 15)];   // B[k = l >> 4][j = l & 15]
  d4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];  // row = (l >> 4) + 4 r, col = l & 15
  const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}

// ---- cost: NM independent MFMAs and NV independent VALU instructions per loop trip
template <int NM, int NV, int KIND>
__global__ void __launch_bounds__(256) k_mix(double* out, double a, double b, float fa, float fb) {
  d4 acc[NM > 0 ? NM : 1];
  for (int m = 0; m < (NM > 0 ? NM : 1); ++m) acc[m] = d4{a, a + 1, a + 2, a + 3};
  float x[NV > 0 ? NV : 1];
  double y[NV > 0 ? NV : 1];
  for (int v = 0; v < (NV > 0 ? NV : 1); ++v) { x[v] = fa + v; y[v] = a + v; }
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      // inline asm: the builtin makes the compiler shuttle the accumulators through AGPRs every trip
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
      // the VALU work of this trip is spread behind the MFMAs
#pragma unroll
      for (int v = m * NV / (NM > 0 ? NM : 1); v < (m + 1) * NV / (NM > 0 ? NM : 1); ++v) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v]) : "v"(fa), "v"(fb));
        if (KIND == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(y[v]) : "v"(a), "v"(b));
        if (KIND == 2) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(x[v]) : "v"(y[v])); }
        if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[v]));
      }
    }
    if (NM == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v]) : "v"(fa), "v"(fb));
        if (KIND == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(y[v]) : "v"(a), "v"(b));
        if (KIND == 2) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(x[v]) : "v"(y[v])); }
        if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[v]));
      }
    }
  }
  double s = 0;
  for (int m = 0; m < (NM > 0 ? NM : 1); ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  for (int v = 0; v < (NV > 0 ? NV : 1); ++v) s += x[v] + y[v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
double run_ns(K kern, double* out, int blocks_per_cu) {  // ns per loop trip per wave slot of a SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu;  // blocks of 256 threads = 4 waves = one wave per SIMD each
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001, 0.5, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001, 0.5, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)ms * 1e6 / ((double)ITERS * blocks_per_cu);  // time a SIMD spends per trip of ONE of its waves
}

static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

int main() {
  // ---------------- semantics
  double hA[64], hB[64], hC[256], hD[256];
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
  srand(7);
  auto rnd = [] { return (rand() / (double)RAND_MAX - 0.5) * 200.0 * (1.0 + rand() / (double)RAND_MAX * 1e-3); };
  int n_trials = 200, match[6] = {0, 0, 0, 0, 0, 0}, total = 0;
  for (int t = 0; t < n_trials; ++t) {
    for (double& v : hA) v = rnd();
    for (double& v : hB) v = rnd();
    for (double& v : hC) v = rnd();
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        const double c = hC[i * 16 + j];
        auto a = [&](int k) { return hA[i * 4 + k]; };
        auto b = [&](int k) { return hB[k * 16 + j]; };
        double cand[6];
        cand[0] = fma(a(3), b(3), fma(a(2), b(2), fma(a(1), b(1), fma(a(0), b(0), c))));  // k = 0 first, fused
        cand[1] = fma(a(0), b(0), fma(a(1), b(1), fma(a(2), b(2), fma(a(3), b(3), c))));  // k = 3 first, fused
        cand[2] = ((((c + a(0) * b(0)) + a(1) * b(1)) + a(2) * b(2)) + a(3) * b(3));      // unfused, k = 0 first
        cand[3] = c + (fma(a(1), b(1), a(0) * b(0)) + fma(a(3), b(3), a(2) * b(2)));       // pairwise
        cand[4] = fma(a(1), b(1), fma(a(0), b(0), c)) + fma(a(3), b(3), a(2) * b(2));
        long double e = (long double)c;
        for (int k = 0; k < 4; ++k) e += (long double)a(k) * (long double)b(k);
        cand[5] = (double)e;  // (nearly) exact dot product rounded once
        for (int q = 0; q < 6; ++q) match[q] += bits(cand[q]) == bits(hD[i * 16 + j]);
        ++total;
      }
  }
  printf("semantics over %d results: fused chain k=0..3 %d, fused chain k=3..0 %d, unfused %d, pairwise %d, split %d, "
         "single rounding %d\n", total, match[0], match[1], match[2], match[3], match[4], match[5]);
  // the case the score loop needs: k = 0 and 1 used, k = 2, 3 zero, C = translation
  {
    int ok01 = 0, ok10 = 0, n = 0;
    for (int t = 0; t < 100; ++t) {
      for (int i = 0; i < 16; ++i) { hA[i * 4] = rnd(); hA[i * 4 + 1] = rnd(); hA[i * 4 + 2] = 0; hA[i * 4 + 3] = 0; }
      for (int j = 0; j < 16; ++j) { hB[j] = rnd() / 100; hB[16 + j] = rnd() / 100; hB[32 + j] = 0; hB[48 + j] = 0; }
      for (double& v : hC) v = rnd();
      hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
      hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
      hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          const double c = hC[i * 16 + j];
          ok01 += bits(fma(hA[i * 4 + 1], hB[16 + j], fma(hA[i * 4], hB[j], c))) == bits(hD[i * 16 + j]);
          ok10 += bits(fma(hA[i * 4], hB[j], fma(hA[i * 4 + 1], hB[16 + j], c))) == bits(hD[i * 16 + j]);
          ++n;
        }
    }
    printf("two-term case over %d results: fma(a1,b1,fma(a0,b0,c)) %d, fma(a0,b0,fma(a1,b1,c)) %d\n", n, ok01, ok10);
  }
  // ---------------- cost
  double* out;
  hipMalloc(&out, 256 * 16 * 256 * 8);
  for (int bpc : {1, 2, 4}) {
    printf("--- %d wave(s) per SIMD\n", bpc);
    const double m4 = run_ns(k_mix<4, 0, 0>, out, bpc);
    printf("4 MFMA f64 16x16x4                    %8.2f ns per trip  (%.2f ns per MFMA)\n", m4, m4 / 4);
    const char* names[4] = {"v_fma_f32", "v_fma_f64", "v_cvt_i32_f64", "v_exp_f32"};
    double v[4], mv[4];
    v[0] = run_ns(k_mix<0, 64, 0>, out, bpc);  mv[0] = run_ns(k_mix<4, 64, 0>, out, bpc);
    v[1] = run_ns(k_mix<0, 32, 1>, out, bpc);  mv[1] = run_ns(k_mix<4, 32, 1>, out, bpc);
    v[2] = run_ns(k_mix<0, 32, 2>, out, bpc);  mv[2] = run_ns(k_mix<4, 32, 2>, out, bpc);
    v[3] = run_ns(k_mix<0, 32, 3>, out, bpc);  mv[3] = run_ns(k_mix<4, 32, 3>, out, bpc);
    const int cnt[4] = {64, 32, 32, 32};
    for (int q = 0; q < 4; ++q)
      printf("%2d %-14s alone %8.2f ns, with 4 MFMA %8.2f ns  (sum %.2f, max %.2f)\n", cnt[q], names[q], v[q], mv[q],
             v[q] + m4, v[q] > m4 ? v[q] : m4);
    const double v128 = run_ns(k_mix<0, 128, 0>, out, bpc), mv128 = run_ns(k_mix<4, 128, 0>, out, bpc);
    printf("128 v_fma_f32     alone %8.2f ns, with 4 MFMA %8.2f ns  (sum %.2f, max %.2f)\n", v128, mv128, v128 + m4,
           v128 > m4 ? v128 : m4);
  }
  return 0;
}
