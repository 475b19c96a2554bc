// How long does one cluster-wide exchange cost on MI355X?  K workgroups publish a double each (agent-scope atomic
// store), meet at a counter barrier (atomicAdd + bounded spin on an agent-scope load) and read all K values back.
// This is the per-round price a multi-CU version of the exact-order PSO would pay.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_cluster_barrier scripts/ubench_cluster_barrier.hip && ./ubench_cluster_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k_rounds(int K, int stride, int rounds, unsigned* bar, double* xc, double* out, unsigned* fail) {
  if (blockIdx.x % stride != 0) return;  // stride 8: participants share an XCD (workgroups go round robin over 8 XCDs)
  const int r = blockIdx.x / stride;
  if (r >= K) return;
  double acc = 0.;
  for (int it = 0; it < rounds; ++it) {
    double* buf = xc + (size_t)(it & 1) * 64;
    if (threadIdx.x == 0) {
      __hip_atomic_store(&buf[r], (double)(it * 64 + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      atomicAdd(bar, 1u);
      const unsigned want = (unsigned)K * (unsigned)(it + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > 20000000u) {  // never hang the device
          atomicAdd(fail, 1u);
          break;
        }
      }
      __threadfence();
    }
    __syncthreads();
    if ((int)threadIdx.x < K) acc += __hip_atomic_load(&buf[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  if ((int)threadIdx.x < K) out[r * 64 + threadIdx.x] = acc;
}

int main() {
  unsigned *bar, *fail;
  double *xc, *out;
  hipMalloc(&bar, 4);
  hipMalloc(&fail, 4);
  hipMalloc(&xc, 2 * 64 * 8);
  hipMalloc(&out, 64 * 64 * 8);
  const int rounds = 2000;
  for (int stride : {8, 1}) {
    for (int K : {1, 2, 4, 8, 16, 32}) {
      float best = 1e30f;
      unsigned failed = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(bar, 0, 4);
        hipMemset(fail, 0, 4);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_rounds, dim3(K * stride), dim3(256), 0, 0, K, stride, rounds, bar, xc, out, fail);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost);
      }
      std::printf("%s K=%2d: %.3f us per exchange round%s\n", stride == 8 ? "one XCD " : "all XCDs", K,
                  1e3 * best / rounds, failed ? "  (SPIN LIMIT HIT)" : "");
    }
  }
  return 0;
}
