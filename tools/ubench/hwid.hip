// Where do the waves of a 4-wave workgroup land?  Reads HW_REG_HW_ID / XCC_ID per wave: workgroups are dealt round-robin
// over XCDs and CUs, and the waves of consecutive workgroups on one CU start on a rotating SIMD (profiles: DESIGN.md 4.3).
// hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(256) k(unsigned* out, int spin) {
  extern __shared__ double s[];
  const int wave = threadIdx.x >> 6;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0000001 + 0.5;
  s[threadIdx.x] = a;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + wave) * 2] = hw; out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc + (s[1] > 1e300 ? 1 : 0); }
}
int main() {
  const int nb = 4096;
  unsigned* d; hipMalloc(&d, nb * 8 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 33 * 1024);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 33 * 1024, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 8); hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
  // HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx9)
  std::map<int, int> hist;   // wave index -> simd histogram key
  int table[4][4] = {};
  for (int b = 0; b < nb; b++) for (int w = 0; w < 4; w++) { unsigned hw = h[(b * 4 + w) * 2]; table[w][(hw >> 4) & 3]++; }
  for (int w = 0; w < 4; w++) printf("wave %d -> simd counts: %d %d %d %d\n", w, table[w][0], table[w][1], table[w][2], table[w][3]);
  for (int b = 0; b < 12; b++) { printf("block %4d:", b); for (int w = 0; w < 4; w++) { unsigned hw = h[(b*4+w)*2]; printf("  [w%d simd %u cu %u se %u xcc %u]", w, (hw>>4)&3, (hw>>8)&15, (hw>>13)&7, h[(b*4+w)*2+1] & 15); } printf("\n"); }
  // blocks sharing a CU: print first CU's blocks
  unsigned key0 = (h[0] >> 8) & 0xff; unsigned x0 = h[1] & 15; int shown = 0;
  for (int b = 0; b < nb && shown < 12; b++) { unsigned hw = h[b*8]; if (((hw >> 8) & 0xff) == key0 && (h[b*8+1] & 15) == x0) { printf("same CU as block 0: block %d wave0 simd %u\n", b, (hw>>4)&3); shown++; } }
  return 0;
}
