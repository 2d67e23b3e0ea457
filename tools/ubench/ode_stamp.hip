// Diagnostic build of the fp64-MFMA stepping kernel with in-kernel cycle stamps (share of each segment of a stage).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVGPA_STAMPS -I../../vgpa_amd/csrc ode_stamp.hip -o ode_stamp
#include "ode_mfma_impl.h"
#include <cstdio>
#include <vector>
#include <random>
using namespace vgpa;
int main(int argc, char** argv) {
  const int D = 40, Np = 1001, B = (argc > 1) ? atoi(argv[1]) : 1, four = (argc > 2) ? atoi(argv[2]) : 0;
  const size_t DD = D * D;
  std::vector<double> A(B * Np * DD), b(B * Np * D), S0(DD, 0.0), Sg(DD, 0.0), m0(D, 1.0);
  std::mt19937_64 rng(1); std::normal_distribution<double> nd(0.0, 1.0);
  for (size_t i = 0; i < A.size(); i++) A[i] = 0.05 * nd(rng);
  for (int p = 0; p < B; p++) for (int t = 0; t < Np; t++) for (int i = 0; i < D; i++) A[((size_t)p * Np + t) * DD + i * D + i] += 8.0;
  for (auto& v : b) v = nd(rng);
  for (int i = 0; i < D; i++) { S0[i * D + i] = 0.2; Sg[i * D + i] = 4.0; }
  OdeArgs a{}; a.four_waves = four; a.D = D; a.Np = Np; a.batch = B; a.dt = 0.01;
  double *dA, *db, *dS0, *dSg, *dm0, *dm, *dS;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dS0, DD * 8); hipMalloc(&dSg, DD * 8); hipMalloc(&dm0, D * 8);
  hipMalloc(&dm, (size_t)B * Np * D * 8); hipMalloc(&dS, (size_t)B * Np * DD * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dS0, S0.data(), DD * 8, hipMemcpyHostToDevice); hipMemcpy(dSg, Sg.data(), DD * 8, hipMemcpyHostToDevice); hipMemcpy(dm0, m0.data(), D * 8, hipMemcpyHostToDevice);
  a.A = dA; a.b = db; a.m0 = dm0; a.S0 = dS0; a.Sigma = dSg; a.m = dm; a.S = dS;
  mfma::launch_nb<3, true, 10>(a, 0); hipDeviceSynchronize();
#ifdef VGPA_STAMPS
  long long zero[8][16] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(mfma::g_stamp), zero, sizeof(zero));
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); mfma::launch_nb<3, true, 10>(a, 0); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("fwd RK4 D=40 Np=%d B=%d waves=%d: %.3f ms  (%.0f cycles/step at 2.4 GHz)\n", Np, B, four ? 4 : 8, ms, ms * 1e-3 * 2.4e9 / (Np - 1));
#ifdef VGPA_STAMPS
  long long st[8][16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(mfma::g_stamp), sizeof(st));
  long long ck[4]; (void)hipMemcpyFromSymbol(ck, HIP_SYMBOL(mfma::g_clk), sizeof(ck));
  printf("in-kernel clock: %.3f GHz (s_memtime %lld ticks over %lld x 10 ns)\n", (double)(ck[2] - ck[0]) / (double)(ck[3] - ck[1]) * 0.1,
         ck[2] - ck[0], ck[3] - ck[1]);
  const char* names[9] = {"elementwise(final)", "mfma product", "mat-vec", "W/pv stores", "barrier A", "W^T/pv loads", "elementwise(stage)", "X stores", "barrier B"};
  for (int w = 0; w < (four ? 4 : 8); w++) {
    long long tot = 0; for (int i = 0; i < 9; i++) tot += st[w][i];
    printf("wave %d: total stamped %lld cycles/step:", w, tot / (Np - 1));
    for (int i = 0; i < 9; i++) printf(" [%s %lld]", names[i], st[w][i] / (Np - 1));
    printf("\n");
  }
#endif
  return 0;
}
