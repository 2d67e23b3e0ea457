// Probe: lane -> (block,i,k)/(block,k,j)/(block,i,j) maps of the fp64 MFMA shapes on gfx950.
// For every (la, lb) pair: A = onehot(la), B = onehot(lb); record which (lane, reg) of D is 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe16(int* out_lane, int* out_reg) {   // 64x64 entries
  const int l = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      d4 acc = {0, 0, 0, 0};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, acc, 0, 0, 0);
      for (int r = 0; r < 4; r++)
        if (acc[r] != 0.0) { out_lane[la * 64 + lb] = l; out_reg[la * 64 + lb] = r; }
    }
}
__global__ void probe4(int* out_lane) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
      if (acc != 0.0) out_lane[la * 64 + lb] = l;
    }
}
int main() {
  int *dl, *dr;
  hipMalloc(&dl, 4096 * 4); hipMalloc(&dr, 4096 * 4);
  std::vector<int> hl(4096), hr(4096);
  hipMemset(dl, 0xff, 4096 * 4); hipMemset(dr, 0xff, 4096 * 4);
  hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, dl, dr);
  hipMemcpy(hl.data(), dl, 4096 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), dr, 4096 * 4, hipMemcpyDeviceToHost);
  printf("== v_mfma_f64_16x16x4_f64: for A-lane la, B-lane lb -> D (lane,reg) [only hits]\n");
  // hypothesis: A lane la = (i = la&15, k = la>>4); B lane lb = (k = lb>>4, j = lb&15); D: col j = lane&15, row i = (lane>>4) + 4*reg
  int bad = 0, hits = 0;
  for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) {
    int L = hl[la * 64 + lb], R = hr[la * 64 + lb];
    bool expect = (la >> 4) == (lb >> 4);
    if ((L >= 0) != expect) { bad++; continue; }
    if (L >= 0) { hits++; int i = la & 15, j = lb & 15; if ((L & 15) != j || ((L >> 4) + 4 * R) != i) bad++; }
  }
  printf("   hypothesis A(i=l&15,k=l>>4) B(k=l>>4,j=l&15) D(col=l&15,row=(l>>4)+4r): hits=%d mismatches=%d\n", hits, bad);
  hipMemset(dl, 0xff, 4096 * 4);
  hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, dl);
  hipMemcpy(hl.data(), dl, 4096 * 4, hipMemcpyDeviceToHost);
  printf("== v_mfma_f64_4x4x4_4b_f64 raw table (la: list of lb->lane)\n");
  for (int la = 0; la < 64; la++) {
    printf("la=%2d:", la);
    for (int lb = 0; lb < 64; lb++) if (hl[la * 64 + lb] >= 0) printf(" %d->%d", lb, hl[la * 64 + lb]);
    printf("\n");
  }
  // hypothesis: A lane = (blk = l>>4, i = l&3, k = (l>>2)&3)?  try several
  const char* names[4] = {"A(i=l&3,k=(l>>2)&3)", "A(k=l&3,i=(l>>2)&3)", "", ""};
  for (int ha = 0; ha < 2; ha++) for (int hb = 0; hb < 2; hb++) for (int hd = 0; hd < 2; hd++) {
    int bad2 = 0, hits2 = 0;
    for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) {
      int ba = la >> 4, bb = lb >> 4;
      int ai = ha ? (la >> 2) & 3 : la & 3, ak = ha ? la & 3 : (la >> 2) & 3;
      int bj = hb ? (lb >> 2) & 3 : lb & 3, bk = hb ? lb & 3 : (lb >> 2) & 3;
      bool expect = (ba == bb) && (ak == bk);
      int L = hl[la * 64 + lb];
      if ((L >= 0) != expect) { bad2++; continue; }
      if (L >= 0) { hits2++; int di = hd ? (L >> 2) & 3 : L & 3, dj = hd ? L & 3 : (L >> 2) & 3;
        if ((L >> 4) != ba || di != ai || dj != bj) bad2++; }
    }
    printf("   hyp A:%s B:%s D:%s -> hits=%d mismatches=%d\n", ha ? "i=(l>>2)&3,k=l&3" : "i=l&3,k=(l>>2)&3",
           hb ? "j=(l>>2)&3,k=l&3" : "j=l&3,k=(l>>2)&3", hd ? "i=(l>>2)&3,j=l&3" : "i=l&3,j=(l>>2)&3", hits2, bad2);
  }
  (void)names;
  return 0;
}
