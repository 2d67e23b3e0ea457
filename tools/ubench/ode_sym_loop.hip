// The symmetric-unit stepping kernel alone, launched back to back for <seconds> (clock / power samples beside it: tools/power_ablations.sh;
// ablation macros of ode_sym_impl.h -- VGPA_ABL_NOFRAG / NOSTORE / NOVEC -- are passed with -D).  Derived from ode_sym_stamp.hip.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DVGPA_EXPERIMENTS -DVGPA_STAMPS] -I../../vgpa_amd/csrc -I../../include ode_sym_stamp.hip -o ode_sym_stamp
// usage: ode_sym_loop <batch> <fwd 1|0> <seconds>
#include "ode_sym_impl.h"
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
using namespace vgpa;
int main(int argc, char** argv) {
  const int D = 40, Np = 1001, B = (argc > 1) ? atoi(argv[1]) : 1, fwd = (argc > 2) ? atoi(argv[2]) : 1;
  const size_t DD = D * D;
  std::vector<double> A((size_t)B * Np * DD), b((size_t)B * Np * D), S0(DD, 0.0), Sg(DD, 0.0), m0(D, 1.0);
  std::mt19937_64 rng(1); std::normal_distribution<double> nd(0.0, 1.0);
  {  // one random problem, repeated (the timing does not depend on the problems being different)
    const size_t n1 = (size_t)Np * DD;
    for (size_t i = 0; i < n1; i++) A[i] = 0.05 * nd(rng);
    for (int t = 0; t < Np; t++) for (int i = 0; i < D; i++) A[(size_t)t * DD + i * D + i] += 8.0;
    for (int p = 1; p < B; p++) std::copy(A.begin(), A.begin() + n1, A.begin() + (size_t)p * n1);
  }
  for (size_t i = 0; i < b.size(); i++) b[i] = 0.001 * (double)((i * 2654435761u) % 1000u);
  for (int i = 0; i < D; i++) { S0[i * D + i] = 0.2; Sg[i * D + i] = 4.0; }
  OdeArgs a{}; a.D = D; a.Np = Np; a.batch = B; a.dt = 0.01;
  a.strideA = (size_t)Np * DD; a.strideB = (size_t)Np * D;
  double *dA, *db, *dS0, *dSg, *dm0, *dm, *dS, *dG;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dS0, DD * 8); hipMalloc(&dSg, DD * 8); hipMalloc(&dm0, D * 8);
  hipMalloc(&dm, (size_t)B * Np * D * 8); hipMalloc(&dS, (size_t)B * Np * DD * 8); hipMalloc(&dG, (size_t)B * Np * DD * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dS0, S0.data(), DD * 8, hipMemcpyHostToDevice); hipMemcpy(dSg, Sg.data(), DD * 8, hipMemcpyHostToDevice); hipMemcpy(dm0, m0.data(), D * 8, hipMemcpyHostToDevice);
  hipMemcpy(dG, dA, A.size() * 8, hipMemcpyDeviceToDevice);      // any finite forcing term
  a.A = dA; a.b = db; a.m0 = dm0; a.S0 = dS0; a.Sigma = dSg; a.m = dm; a.S = dS;
  a.dEm = db; a.dEs = dG; a.lam = dm; a.psi = dS; a.js_const = dSg; a.n_obs = 0;
  auto go = [&]() { return fwd ? sym::launch_sym<3, true, 10>(a, 0) : sym::launch_sym<3, false, 10>(a, 0); };
  go(); hipDeviceSynchronize();
  const double seconds = (argc > 3) ? atof(argv[3]) : 6.0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total = 0.0; int n = 0;
  while (total < 1e3 * seconds) {
    hipEventRecord(e0); for (int r = 0; r < 8; r++) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); total += ms; n += 8;
  }
  printf("%s RK4 D=40 Np=%d B=%d: %.3f ms per launch over %.1f s  err=%s\n", fwd ? "fwd" : "bwd", Np, B, total / n, 1e-3 * total, hipGetErrorString(hipGetLastError()));
  return 0;
}
