// What does a device-wide barrier cost on MI355X -- the alternative to a kernel boundary between the Runge-Kutta stages of one problem at
// 64 < D <= 512 ("or a persistent multi-CU kernel with a device-wide barrier")?  G co-resident workgroups (one per CU at
// most) meet `iters` times at a counter in device memory (agent-scope atomic add by one lane per workgroup, then a bounded spin on an atomic
// load); between two barriers every workgroup writes one value and reads its neighbour's (the data a stage hands to the next one must be
// visible across XCDs: release fence before the arrival, acquire fence behind the barrier).  Every spin is BOUNDED: a workgroup that does not
// see the others within `limit` polls raises an abort flag and all leave -- the kernel cannot hang.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_barrier(unsigned* counter, unsigned* abort_flag, double* slots, double* out, int iters, unsigned limit, int exchange) {
  const unsigned G = gridDim.x;
  const int wg = blockIdx.x, tid = threadIdx.x;
  __shared__ int leave;
  if (tid == 0) leave = 0;
  __syncthreads();
  double acc = 0.0;
  for (int it = 0; it < iters; it++) {
    if (exchange && tid == 0) slots[(size_t)(it & 1) * G + wg] = (double)(it * 1000 + wg);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(it + 1) * G;
      unsigned polls = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++polls > limit || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          leave = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (leave) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (exchange && tid == 0) {
      const double v = __hip_atomic_load(&slots[(size_t)(it & 1) * G + (wg + 1) % G], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += v - (double)(it * 1000 + (wg + 1) % G);       // 0 when the neighbour's value of THIS round is visible
    }
  }
  if (tid == 0) out[wg] = acc;
}

int main() {
  unsigned *counter, *abort_flag; double *slots, *out;
  const int maxg = 256;
  hipMalloc(&counter, 4); hipMalloc(&abort_flag, 4); hipMalloc(&slots, sizeof(double) * 2 * maxg); hipMalloc(&out, sizeof(double) * maxg);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("# device-wide barrier among G co-resident workgroups of 256 threads (one arrival + bounded spin per barrier); us per barrier\n");
  printf("# %6s %10s %14s %14s %8s\n", "G", "iters", "barrier only", "with exchange", "errors");
  for (int G : {8, 16, 36, 64, 128, 256}) {
    float us[2] = {0, 0}; double err = 0.0; unsigned aborted = 0;
    for (int exchange = 0; exchange < 2; exchange++) {
      const int iters = 2000;
      for (int rep = 0; rep < 2; rep++) {                  // first repetition: warm-up
        hipMemset(counter, 0, 4); hipMemset(abort_flag, 0, 4); hipMemset(slots, 0, sizeof(double) * 2 * maxg);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_barrier, dim3(G), dim3(256), 0, 0, counter, abort_flag, slots, out, iters, 1u << 22, exchange);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        us[exchange] = 1e3f * ms / iters;
      }
      unsigned a = 0; hipMemcpy(&a, abort_flag, 4, hipMemcpyDeviceToHost); aborted |= a;
      std::vector<double> h(G); hipMemcpy(h.data(), out, sizeof(double) * G, hipMemcpyDeviceToHost);
      if (exchange) for (double v : h) err += v < 0 ? -v : v;
    }
    printf("  %6d %10d %14.2f %14.2f %8g%s\n", G, 2000, us[0], us[1], err, aborted ? "  ABORTED (spin limit)" : "");
  }
  return 0;
}
