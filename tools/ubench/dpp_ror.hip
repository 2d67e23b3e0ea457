// Probe: __builtin_amdgcn_update_dpp with ROW_ROR:n (dpp_ctrl 0x120 + n): which source lane does destination lane l read?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x120 + N, 0xf, 0xf, false);
}
template <int N>
void run(int* d) {
  int h[64];
  hipLaunchKernelGGL((k<N>), dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("row_ror:%d  dst<-src:", N);
  for (int l = 0; l < 20; l++) printf(" %d<-%d", l, h[l]);
  int ok = 1;
  for (int l = 0; l < 64; l++) ok &= (h[l] == (l & ~15) + (((l & 15) - N) & 15));
  printf("   [dst l reads lane (l - %d) mod 16 of its row: %s]\n", N, ok ? "yes" : "NO");
}
int main() {
  int* d;
  (void)hipMalloc(&d, 256);
  run<4>(d); run<8>(d); run<12>(d);
  return 0;
}
