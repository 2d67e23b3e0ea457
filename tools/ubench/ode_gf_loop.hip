// The backward RK4 fragment-cover kernel with helper waves, with (grad 1) and without (grad 0) the gradient assembly on them, launched
// back to back for <seconds>.  Inputs are arbitrary finite numbers (the timing does not depend on them); ablation macros of
// ode_sym_impl.h (-DVGPA_GF_ABL=<bits>, wrong results) show what each phase of the assembly costs.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DVGPA_EXPERIMENTS -DVGPA_STAMPS_ROLE | -DVGPA_EXPERIMENTS -DVGPA_GF_ABL=<bits>] -I../../vgpa_amd/csrc -I../../include ode_gf_loop.hip -o ode_gf_loop
// usage: ode_gf_loop <batch> <grad 1|0> <seconds> [fwd 1|0]
#include "ode_sym_impl.h"
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
using namespace vgpa;
int main(int argc, char** argv) {
  const int D = 40, Np = 1001, B = (argc > 1) ? atoi(argv[1]) : 1, grad = (argc > 2) ? atoi(argv[2]) : 1;
  const size_t DD = D * D, PK = D * (D + 1) / 2, len_x = (size_t)Np * DD + (size_t)Np * D;
  std::vector<double> x((size_t)B * len_x);
  std::mt19937_64 rng(1); std::normal_distribution<double> nd(0.0, 1.0);
  for (size_t i = 0; i < len_x; i++) x[i] = 0.05 * nd(rng);
  for (int t = 0; t < Np; t++) for (int i = 0; i < D; i++) x[(size_t)t * DD + i * D + i] += 8.0;
  for (int p = 1; p < B; p++) std::copy(x.begin(), x.begin() + len_x, x.begin() + (size_t)p * len_x);
  std::vector<double> Sg(DD, 0.0);
  for (int i = 0; i < D; i++) Sg[i * D + i] = 4.0;
  double *dx, *dSg, *dv, *dS, *dG, *dg, *dpsi, *dlam;
  hipMalloc(&dx, x.size() * 8); hipMalloc(&dSg, DD * 8); hipMalloc(&dv, (size_t)B * Np * D * 8); hipMalloc(&dS, (size_t)B * Np * PK * 8);
  hipMalloc(&dG, (size_t)B * Np * PK * 8); hipMalloc(&dg, x.size() * 8); hipMalloc(&dpsi, (size_t)B * Np * DD * 8); hipMalloc(&dlam, (size_t)B * Np * D * 8);
  hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dSg, Sg.data(), DD * 8, hipMemcpyHostToDevice);
  hipMemcpy(dv, dx, (size_t)B * Np * D * 8, hipMemcpyDeviceToDevice);
  hipMemcpy(dS, dx, (size_t)B * Np * PK * 8, hipMemcpyDeviceToDevice); hipMemcpy(dG, dx, (size_t)B * Np * PK * 8, hipMemcpyDeviceToDevice);
  OdeArgs a{}; a.D = D; a.Np = Np; a.batch = B; a.dt = 0.01; a.sym_units = 1;
  a.strideA = a.strideB = len_x;
  a.A = dx; a.b = dx + (size_t)Np * DD;
  a.dEm = dv; a.dEs = dG; a.ds_packed = 1; a.lam = dlam; a.psi = dpsi; a.js_const = dSg; a.n_obs = 0;
  a.q_on = 1; a.q_scale = 0.25;
  if (grad) { a.grad_on = 1; a.s_packed = 1; a.S = dS; a.m = dv; a.Ef = dv; a.Am = dv; a.g = dg; }
  const int fwd = (argc > 4) ? atoi(argv[4]) : 0;
  if (fwd) { a.q_on = 0; a.m0 = dv; a.S0 = dSg; a.Sigma = dSg; a.m = dlam; a.S = dpsi; a.s_packed = 0; }
  auto go = [&]() { return fwd ? sym::launch_cover<3, true, 10, 0, true>(a, 0, false) : sym::launch_cover<3, false, 10, 0, true>(a, 0, false); };
  hipError_t e0_ = go(); hipDeviceSynchronize();
  const double seconds = (argc > 3) ? atof(argv[3]) : 3.0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total = 0.0; int n = 0;
  while (total < 1e3 * seconds) {
    hipEventRecord(e0); for (int r = 0; r < 4; r++) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); total += ms; n += 4;
  }
#ifdef VGPA_STAMPS_ROLE
  {
    long long zero[4][16] = {}, st[4][16];
    hipMemcpyToSymbol(HIP_SYMBOL(mfma::g_stamp_role), zero, sizeof(zero));
    go(); hipDeviceSynchronize();
    hipMemcpyFromSymbol(st, HIP_SYMBOL(mfma::g_stamp_role), sizeof(st));
    for (int r = 0; r < 3; r++) {
      printf("%s wave 0 of workgroup 0, cycles per stage [busy | barrier wait]:", r == 2 ? (grad ? "grad   " : "helper2") : r ? "helper " : "product");
      for (int j = 0; j < 4; j++) printf("  j%d %lld | %lld", j, st[r][2 * j] / (Np - 1), st[r][2 * j + 1] / (Np - 1));
      printf("\n");
    }
    if (grad) {
      const char* nm[10] = {"settle", "out", "build", "kp0", "band_u", "prefetch", "kp1", "kp2-3", "kp4", "epilogue"};
      printf("grad wave 0 phases, cycles per step:");
      for (int i = 0; i < 10; i++) printf("  %s %lld", nm[i], st[3][i] / (Np - 1));
      printf("\n");
    }
  }
#endif
  printf("%s RK4 D=40 Np=%d B=%d grad=%d abl=%d: %.3f ms per launch over %.1f s  err=%s / %s\n", fwd ? "fwd" : "bwd", Np, B, grad, VGPA_GF_ABL, total / n, 1e-3 * total,
         hipGetErrorString(e0_), hipGetErrorString(hipGetLastError()));
  return 0;
}
