// What does one tagged-slot exchange between K workgroups cost on MI355X, and can it stay inside one XCD's L2?
// Every workgroup's thread 0 stores {value, tag} into its 16-byte slot (two parities of slots), lane r of wave 0 of every
// workgroup reads slot r until all K carry the round's tag -- the live sequence's cluster exchange (eval_round) without
// the evaluation.  Variants of how the slot is written and read:
//   0  store sc1, load sc1                     (the shipped exchange: past the XCD's L2)
//   1  plain store, 64-bit atomic OR of 0      (atomics execute in the issuing XCD's L2)
//   2  plain store, load sc0
//   3  plain store, load sc0 sc1               (system scope, for reference)
// with the participants spread over the XCDs (consecutive workgroups) or on ONE XCD (every 8th workgroup of the grid:
// workgroups go to the XCDs in turn); each workgroup reports HW_REG_XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_xcd scripts/ubench_xcd_exchange.hip && /tmp/ubench_xcd
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int V>
__device__ __forceinline__ void slot_store(uint4* slot, unsigned lo, unsigned hi, unsigned tag) {
  u32x4 v = {lo, tag, hi, tag};
  if (V == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(slot), "v"(v) : "memory");
}
template <int V>
__device__ __forceinline__ u32x4 slot_load(uint4* slot) {
  u32x4 v;
  if (V == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
  else if (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
  else if (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(slot) : "memory");
  else {
    u32x2 a, b, z = {0u, 0u};
    asm volatile("global_atomic_or_x2 %0, %2, %3, off sc0\n\tglobal_atomic_or_x2 %1, %2, %3, off offset:8 sc0\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(slot), "v"(z) : "memory");
    v = u32x4{a.x, a.y, b.x, b.y};
  }
  return v;
}

template <int V>
__global__ void k_rounds(int K, int stride, int rounds, uint4* xc, unsigned* xcc, unsigned* fail, double* out) {
  if (blockIdx.x % stride != 0) return;
  const int r = blockIdx.x / stride;
  if (r >= K) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[r] = id;
  double acc = 0.;
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  for (int it = 0; it < rounds && !dead; ++it) {
    uint4* buf = xc + (size_t)(it & 1) * 64;
    const unsigned tag = (unsigned)it + 1u;
    if (threadIdx.x == 0) slot_store<V>(&buf[r], (unsigned)(it * 64 + r), 7u, tag);
    if (threadIdx.x < 64) {
      const bool mine = (int)threadIdx.x < K;
      const unsigned long long t0 = wall_clock64();
      u32x4 v = {0, 0, 0, 0};
      for (;;) {
        bool ok = true;
        if (mine) {
          v = slot_load<V>(&buf[threadIdx.x]);
          ok = v.y == tag && v.w == tag;
        }
        if (__all(ok)) break;
        if (wall_clock64() - t0 > 20000ull) {  // 200 us on the 100 MHz counter: nobody is coming
          if (threadIdx.x == 0) {
            atomicAdd(fail, 1u);
            dead = 1;
          }
          break;
        }
      }
      if (mine) acc += (double)v.x;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < K) out[r * 64 + threadIdx.x] = acc;
}

template <int V>
void run(const char* name, uint4* xc, unsigned* xcc, unsigned* fail, double* out) {
  const int rounds = 2000;
  for (int stride : {1, 8}) {
    for (int K : {2, 8, 16}) {
      float best = 1e30f;
      unsigned failed = 0, ids[64];
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(xc, 0, 2 * 64 * 16);
        hipMemset(fail, 0, 4);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_rounds<V>, dim3(K * stride), dim3(256), 0, 0, K, stride, rounds, xc, xcc, fail, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost);
      }
      hipMemcpy(ids, xcc, 64 * 4, hipMemcpyDeviceToHost);
      std::printf("%-34s %s K=%2d: %.3f us per exchange%s   XCC ids:", name, stride == 8 ? "every 8th WG" : "consecutive ", K,
                  1e3 * best / rounds, failed ? "  (SPIN LIMIT HIT)" : "");
      for (int i = 0; i < K; ++i) std::printf(" %u", ids[i] & 15u);
      std::printf("\n");
      std::fflush(stdout);
    }
  }
}

int main() {
  uint4* xc;
  unsigned *xcc, *fail;
  double* out;
  hipMalloc(&xc, 2 * 64 * 16);
  hipMalloc(&xcc, 64 * 4);
  hipMalloc(&fail, 4);
  hipMalloc(&out, 64 * 64 * 8);
  run<0>("store sc1 / load sc1", xc, xcc, fail, out);
  run<1>("plain store / atomic or 0", xc, xcc, fail, out);
  run<2>("plain store / load sc0", xc, xcc, fail, out);
  run<3>("plain store / load sc0 sc1", xc, xcc, fail, out);
  return 0;
}
