// VALU issue-cost microbenchmark for gfx950: cycles per wave64 instruction per SIMD, measured with
// 4 waves/SIMD resident on every CU and 8 independent dependency chains per wave.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/ubench_valu.hip -o scripts/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 4096;

#define KERNEL_T(NAME, T, ASM)                                                      \
  __global__ void __launch_bounds__(256) NAME(T* out, T a, T b) {                   \
    T x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7; \
    for (int i = 0; i < ITERS; ++i) {                                               \
      asm volatile(ASM : "+v"(x0) : "v"(a), "v"(b) : "vcc");                        \
      asm volatile(ASM : "+v"(x1) : "v"(a), "v"(b) : "vcc");                        \
      asm volatile(ASM : "+v"(x2) : "v"(a), "v"(b) : "vcc");                        \
      asm volatile(ASM : "+v"(x3) : "v"(a), "v"(b) : "vcc");                        \
      asm volatile(ASM : "+v"(x4) : "v"(a), "v"(b) : "vcc");                        \
      asm volatile(ASM : "+v"(x5) : "v"(a), "v"(b) : "vcc");                        \
      asm volatile(ASM : "+v"(x6) : "v"(a), "v"(b) : "vcc");                        \
      asm volatile(ASM : "+v"(x7) : "v"(a), "v"(b) : "vcc");                        \
    }                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; \
  }
#define KERNEL_F64(NAME, ASM) KERNEL_T(NAME, double, ASM)
#define KERNEL_F32(NAME, ASM) KERNEL_T(NAME, float, ASM)

// f64 <-> f32/i32 conversions: 64-bit chain value x, 32-bit temp t
#define KERNEL_CVT(NAME, ASM1, ASM2)                                                \
  __global__ void __launch_bounds__(256) NAME(double* out, double a, double b) {    \
    double x[8]; float t[8];                                                        \
    for (int k = 0; k < 8; ++k) { x[k] = a + k; t[k] = (float)b + k; }              \
    for (int i = 0; i < ITERS; ++i) {                                               \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) {                               \
        asm volatile(ASM1 : "=v"(t[k]) : "v"(x[k]));                                \
        asm volatile(ASM2 : "=v"(x[k]) : "v"(t[k]));                                \
      }                                                                             \
    }                                                                               \
    double s = 0; for (int k = 0; k < 8; ++k) s += x[k];                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                 \
  }

KERNEL_F64(k_fma64, "v_fma_f64 %0, %1, %2, %0")
KERNEL_F64(k_add64, "v_add_f64 %0, %0, %1")
KERNEL_F64(k_mul64, "v_mul_f64 %0, %0, %2")
KERNEL_F64(k_cmp64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL_F64(k_floor64, "v_floor_f64 %0, %0")
KERNEL_F64(k_fract64, "v_fract_f64 %0, %0")
KERNEL_F64(k_rndne64, "v_rndne_f64 %0, %0")
KERNEL_F64(k_trunc64, "v_trunc_f64 %0, %0")
KERNEL_F64(k_max64, "v_max_f64 %0, %0, %1")
KERNEL_F64(k_pkfma32, "v_pk_fma_f32 %0, %0, %0, %0")
KERNEL_F64(k_pkmul32, "v_pk_mul_f32 %0, %0, %0")
KERNEL_F64(k_mov64, "v_mov_b64 %0, %1")
KERNEL_F32(k_fma32, "v_fma_f32 %0, %1, %2, %0")
KERNEL_F32(k_mul32, "v_mul_f32 %0, %0, %2")
KERNEL_F32(k_add32, "v_add_f32 %0, %0, %1")
KERNEL_F32(k_min32, "v_min_f32 %0, %0, %1")
KERNEL_F32(k_exp32, "v_exp_f32 %0, %0")
KERNEL_F32(k_addu, "v_add_u32 %0, %0, %1")
KERNEL_F32(k_subrev, "v_subrev_u32 %0, %1, %0")
KERNEL_F32(k_and, "v_and_b32 %0, %0, %1")
KERNEL_F32(k_lshr, "v_lshrrev_b32 %0, %1, %0")
KERNEL_F32(k_bfe, "v_bfe_u32 %0, %0, %1, %2")
KERNEL_F32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL_F32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_F32(k_cndm, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F32(k_cmp32, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL_F32(k_cmpu, "v_cmp_gt_u32 vcc, %0, %1")
KERNEL_F32(k_cmpu_s, "v_cmp_gt_u32 s[20:21], %0, %1")
KERNEL_F32(k_mov, "v_mov_b32 %0, %1")
KERNEL_F32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL_F32(k_lshladd, "v_lshl_add_u32 %0, %0, 4, %1")
KERNEL_F32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL_F32(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
// the remaining mnemonics of the score loop (scripts/isa_mix.py prices the loop with this table)
KERNEL_F32(k_fmac32, "v_fmac_f32 %0, %1, %2")
KERNEL_F32(k_minu32, "v_min_u32 %0, %0, %1")
KERNEL_F32(k_mulu24, "v_mul_u32_u24 %0, %0, %1")
KERNEL_F32(k_addlshl, "v_add_lshl_u32 %0, %0, %1, 1")
KERNEL_F64(k_pkadd32, "v_pk_add_f32 %0, %0, %0")
KERNEL_F64(k_fmac64, "v_fmac_f64 %0, %1, %2")
KERNEL_CVT(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", "v_cvt_f64_f32 %0, %1")
KERNEL_CVT(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1", "v_cvt_f64_i32 %0, %1")

template <typename K, typename T>
double run(K kern, T* out, T a, T b, int n_instr_per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * 4;  // 4 blocks of 256 threads per CU = 16 waves/CU = 4 waves/SIMD
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, a, b);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, a, b);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)ms * 1e-3 / (4.0 * ITERS * n_instr_per_iter);  // seconds per wave-instruction per SIMD
}

int main() {
  double* d64;
  float* d32;
  hipMalloc(&d64, 256 * 1024 * 8);
  hipMalloc(&d32, 256 * 1024 * 4);
  const double base = run(k_fma32, d32, 1.0001f, 0.5f, 8);
  printf("v_fma_f32: %.3f ns per wave-instr per SIMD (if 2 cycles => clock %.2f GHz)\n", base * 1e9,
         2.0 / (base * 1e9));
#define R64(K, N) printf("%-16s %.2f cyc\n", #K, 2.0 * run(K, d64, 1.0001, 0.5, N) / base);
#define R32(K, N) printf("%-16s %.2f cyc\n", #K, 2.0 * run(K, d32, 1.0001f, 0.5f, N) / base);
  R64(k_fma64, 8) R64(k_add64, 8) R64(k_mul64, 8) R64(k_cmp64, 8) R64(k_floor64, 8) R64(k_fract64, 8)
  R64(k_rndne64, 8) R64(k_trunc64, 8) R64(k_max64, 8) R64(k_pkfma32, 8) R64(k_pkmul32, 8) R64(k_mov64, 8)
  R32(k_mul32, 8) R32(k_add32, 8) R32(k_min32, 8) R32(k_exp32, 8) R32(k_addu, 8) R32(k_subrev, 8) R32(k_and, 8)
  R32(k_lshr, 8) R32(k_bfe, 8) R32(k_bcnt, 8) R32(k_mad24, 8) R32(k_cndm, 8) R32(k_cmp32, 8) R32(k_cmpu, 8)
  R32(k_cmpu_s, 8) R32(k_mov, 8) R32(k_add3, 8) R32(k_lshladd, 8) R32(k_cvt_i32_f32, 8) R32(k_cvt_f32_i32, 8)
  R32(k_fmac32, 8) R32(k_minu32, 8) R32(k_mulu24, 8) R32(k_addlshl, 8) R64(k_pkadd32, 8) R64(k_fmac64, 8)
  printf("pair costs (sum of the two instructions):\n");
  printf("%-34s %.2f cyc\n", "cvt_f32_f64 + cvt_f64_f32", 2.0 * 2 * run(k_cvt_f32_f64, d64, 1.0001, 0.5, 16) / base);
  printf("%-34s %.2f cyc\n", "cvt_i32_f64 + cvt_f64_i32", 2.0 * 2 * run(k_cvt_i32_f64, d64, 1.0001, 0.5, 16) / base);
  return 0;
}
