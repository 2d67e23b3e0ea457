// Diagnostic build of the role-specialised stepping kernel with in-kernel cycle stamps (share of each segment of a phase).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DVGPA_EXPERIMENTS -DVGPA_STAMPS] -I../../vgpa_amd/csrc -I../../include ode_pe_stamp.hip -o ode_pe_stamp
// usage: ode_pe_stamp <batch> <pair_mode 0|1|2>
#include "ode_mfma_impl.h"
#include <cstdio>
#include <vector>
#include <random>
using namespace vgpa;
int main(int argc, char** argv) {
  const int D = 40, Np = 1001, B = (argc > 1) ? atoi(argv[1]) : 1, pm = (argc > 2) ? atoi(argv[2]) : 0;
  const size_t DD = D * D;
  std::vector<double> A((size_t)B * Np * DD), b((size_t)B * Np * D), S0(DD, 0.0), Sg(DD, 0.0), m0(D, 1.0);
  std::mt19937_64 rng(1); std::normal_distribution<double> nd(0.0, 1.0);
  for (size_t i = 0; i < A.size(); i++) A[i] = 0.05 * nd(rng);
  for (int p = 0; p < B; p++) for (int t = 0; t < Np; t++) for (int i = 0; i < D; i++) A[((size_t)p * Np + t) * DD + i * D + i] += 8.0;
  for (auto& v : b) v = nd(rng);
  for (int i = 0; i < D; i++) { S0[i * D + i] = 0.2; Sg[i * D + i] = 4.0; }
  OdeArgs a{}; a.D = D; a.Np = Np; a.batch = B; a.dt = 0.01;
  a.strideA = (size_t)Np * DD; a.strideB = (size_t)Np * D;
  double *dA, *db, *dS0, *dSg, *dm0, *dm, *dS;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dS0, DD * 8); hipMalloc(&dSg, DD * 8); hipMalloc(&dm0, D * 8);
  hipMalloc(&dm, (size_t)B * Np * D * 8); hipMalloc(&dS, (size_t)B * Np * DD * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dS0, S0.data(), DD * 8, hipMemcpyHostToDevice); hipMemcpy(dSg, Sg.data(), DD * 8, hipMemcpyHostToDevice); hipMemcpy(dm0, m0.data(), D * 8, hipMemcpyHostToDevice);
  a.A = dA; a.b = db; a.m0 = dm0; a.S0 = dS0; a.Sigma = dSg; a.m = dm; a.S = dS;
  mfma::launch_nb<3, true, 10>(a, 0); hipDeviceSynchronize();
#ifdef VGPA_STAMPS
  long long zero[4][8] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(mfma::g_stamp), zero, sizeof(zero));
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); mfma::launch_nb<3, true, 10>(a, 0); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("fwd RK4 D=40 Np=%d B=%d pair_mode=%d: %.3f ms  (%.0f cycles/step at 2.4 GHz)  err=%s\n", Np, B, pm, ms, ms * 1e-3 * 2.4e9 / (Np - 1), hipGetErrorString(hipGetLastError()));
#ifdef VGPA_STAMPS
  long long st[4][8]; hipMemcpyFromSymbol(st, HIP_SYMBOL(mfma::g_stamp), sizeof(st));
  const char* pn[4] = {"product A", "barrier", "product B", "barrier"};
  const char* en[5] = {"mat-vec", "barrier(product)", "element-wise", "staging+stores", "barrier"};
  printf("P wave 0, cycles/step:"); for (int i = 0; i < 4; i++) printf(" [%s %lld]", pn[i], st[0][i] / (Np - 1)); printf("\n");
  for (int r = 1; r <= 2; r++) { printf("E wave 0 of problem %c, cycles/step:", r == 1 ? 'A' : 'B'); for (int i = 0; i < 5; i++) printf(" [%s %lld]", en[i], st[r][i] / (Np - 1)); printf("\n"); }
#endif
  return 0;
}
