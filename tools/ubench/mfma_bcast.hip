// Probe: do the A-broadcast controls (cbsz / abid) of v_mfma_f64_4x4x4_4b_f64 act on gfx950?  With cbsz = 2 ("broadcast one block
// to four") and abid = a, block q of the product should take its A operand from block a of the A register: D_q = A_a . B_q.  If so,
// ONE A register and ONE B register give all 16 block pairs (a, q) in four instructions -- four times the products per fragment
// read of the symmetric-unit steppers.  Lane maps (profiles/r01_fp64_mfma_layout_probe.txt): A[i = l&3, k = l>>4],
// B[k = l>>4, j = l&3], D[i = l>>4, j = l&3], block (l>>2)&3.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID>
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID>
int run(const double* da, const double* db, double* dd, const double* ha, const double* hb) {
  double hd[64];
  hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  int bad_plain = 0, bad_bcast = 0;
  for (int l = 0; l < 64; l++) {
    const int i = l >> 4, j = l & 3, q = (l >> 2) & 3;
    double plain = 0.0, bc = 0.0;
    for (int kk = 0; kk < 4; kk++) {
      const double bv = hb[16 * kk + 4 * q + j];
      plain += ha[16 * kk + 4 * q + i] * bv;            // D_q = A_q . B_q
      bc += ha[16 * kk + 4 * ABID + i] * bv;            // D_q = A_abid . B_q
    }
    if (hd[l] != plain) bad_plain++;
    if (hd[l] != bc) bad_bcast++;
  }
  printf("cbsz=%d abid=%d: mismatches vs D_q = A_q.B_q: %2d   vs D_q = A_abid.B_q: %2d\n", CBSZ, ABID, bad_plain, bad_bcast);
  return 0;
}
int main() {
  double ha[64], hb[64], *da, *db, *dd;
  for (int l = 0; l < 64; l++) { ha[l] = 1.0 + l; hb[l] = 100.0 + 3.0 * l; }
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  run<0, 0>(da, db, dd, ha, hb);
  run<2, 0>(da, db, dd, ha, hb); run<2, 1>(da, db, dd, ha, hb); run<2, 2>(da, db, dd, ha, hb); run<2, 3>(da, db, dd, ha, hb);
  run<1, 0>(da, db, dd, ha, hb); run<1, 1>(da, db, dd, ha, hb);
  return 0;
}
