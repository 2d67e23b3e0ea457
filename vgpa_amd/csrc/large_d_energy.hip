// Lorenz-96 energy terms for D > 64 (src/dynamics/lorenz_96.py:316-438 with utilities.py:239-310 and
// variational.py:339-400), batched over grid points, built from a batched fp64-MFMA GEMM and a 64x64 diagonal-block
// kernel.  Same identities as the LDS-resident kernel (energy.hip):
//     L = chol(c S_t),  X = L^-1,  G = A L,  chi(p, i) = m_i +/- L[i][r_p]  (flat np.roll over the (M, D) matrix, Q1)
//     v_p = sum_i isg_i (l96_flat(chi)[p,i] + (A m)_i +/- G[i][r_p] - b_i)^2
//     E_t = 1/2 (w0 v_0 + w sum_r (v_+r + v_-r)),  dE/dm = c/2 X^T delta,  dE/dS = c/2 X^T diag(q) X
// Blocked algorithms with block size 64 (T = ceil(D/64) block columns), all steps batched over `nb` grid points:
//   Cholesky (right-looking):  diag block: L_JJ = chol(C_JJ), X_JJ = L_JJ^-1      (k_diag64, one wave per block)
//                              panel     : L[R,J] = C[R,J] X_JJ^T                 (GEMM, in place)
//                              trailing  : C[R,R] -= L[R,J] L[R,J]^T              (GEMM, lower tiles only)
//   inverse  (by block rows):  T1 = L[I,0:I] X[0:I,0:I];  X[I,0:I] = -X_II T1     (2 GEMMs)
#include <cstdlib>

#include "vgpa_internal.h"
#include "chol_wave.h"

namespace vgpa {
namespace lde {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int BN = 64, BK = 16, NT = 256, NBLK = 64;

// ---- batched general GEMM:  C = alpha * op(A) op(B) + beta * C ----------------------------------------------------
struct GemmB {
  int M, N, K;
  const double* A; long long lda, sA;     // op(A)[i][k] = TA ? A[k*lda + i] : A[i*lda + k]
  const double* B; long long ldb, sB;     // op(B)[k][j] = TB ? B[j*ldb + k] : B[k*ldb + j]
  double* C; long long ldc, sC;
  double alpha, beta;
  int lower_only;                         // skip tiles that lie strictly above the diagonal
  int k_tri;                              // operands with known zero blocks: 1 = op(B)[k][j] = 0 for k < j (B lower
                                          // triangular): the k loop of a tile starts at its first column; 2 = also
                                          // op(A)[i][k] = 0 for k < i: it starts at max(first row, first column).
                                          // Only exact zeros are skipped, so the result is bit-identical.
                                          // 3 = op(B) as in 1 is NOT assumed; op(A)[i][k] = 0 for k > i (A lower triangular,
                                          // whole 64-blocks): the k loop of a tile ends behind its last row.
  const double* epi_u;                    // != nullptr: C = alpha (op(A) op(B) - u v^T), u = epi_u + z1 epi_s, v = epi_v + z1 epi_s (the gradient's
  const double* epi_v;                    // rank-one term in the product's epilogue instead of a pass over the result; beta = 0)
  long long epi_s;
  int k_down;                             // k_tri = 1 in the 16-byte-load kernel: the k loop runs down from the common end (set by gemm_b)
  int mirror;                             // symmetric result (beta = 0, lower_only): the tiles below the diagonal are stored a second
                                          // time, transposed, above it (the skipped tile's own product associates q_k with the other
                                          // factor: equal to it up to rounding, and now symmetric in every bit between tiles)
  int tile_map;                           // which tile a workgroup takes (set by gemm_b).  Workgroup n of a launch runs on XCD n mod 8,
                                          // and with 16 column tiles per row every XCD would always get the same two columns: 2.4 x the
                                          // work on XCD 0 for a triangular operand (k_tri), 24 against 10 live tiles for lower_only.
                                          // 1 = column tile (x + y) mod gridDim.x; 2 = blockIdx.x counts the tiles on and below the
                                          // diagonal row by row (M = N, no idle workgroups)
  int nb1;                                // > 0: two batch levels, blockIdx.z = z2 * nb1 + z1 (z1: grid point, strides sA / sB / sC;
  long long sA2, sB2, sC2;                //      z2: independent sub-problem of the same grid point, strides sA2 / sB2 / sC2)
};

__device__ __forceinline__ bool tile_origin(const GemmB& g, int BM, int& i0, int& j0) {
  int ti = blockIdx.y, tj = blockIdx.x;
  if (g.tile_map == 1) {
    tj = (tj + ti) % (int)gridDim.x;
  } else if (g.tile_map == 2) {
    const int t = blockIdx.x;
    int r = (int)((__builtin_sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) r++;
    while (r * (r + 1) / 2 > t) r--;
    ti = r; tj = t - r * (r + 1) / 2;
  }
  i0 = ti * BM; j0 = tj * BN;
  return !(g.lower_only && j0 >= i0 + BM);
}

__device__ __forceinline__ long long batch_origin(const GemmB& g, const double*& A, const double*& B, double*& C) {
  long long z1 = blockIdx.z, z2 = 0;
  if (g.nb1 > 0) { z2 = blockIdx.z / g.nb1; z1 = blockIdx.z - z2 * g.nb1; }
  A = g.A + z1 * g.sA + z2 * g.sA2;
  B = g.B + z1 * g.sB + z2 * g.sB2;
  C = g.C + z1 * g.sC + z2 * g.sC2;
  return z1;
}

template <bool TA, bool TB, int BM>
__global__ void __launch_bounds__(NT) k_gemm_b(GemmB g) {
  constexpr int LDAS = BM + 17;
  constexpr int LDBS = TB ? (BN + 17) : (BN + 16);
  constexpr int MT = BM / 32;
  constexpr int AQ = BM * BK / NT;
  __shared__ __attribute__((aligned(16))) double As[2][BK * LDAS];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * LDBS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int i0, j0;
  if (!tile_origin(g, BM, i0, j0)) return;
  const double *A, *B;
  double* C;
  const long long z1 = batch_origin(g, A, B, C);
  const int fi = lane & 15, fk = lane >> 4;
  double ra[AQ], rb[4];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int q = 0; q < AQ; q++) {
      int i, k;
      if (TA) { i = tid & (BM - 1); k = (tid / BM) + (NT / BM) * q; }
      else { k = tid & 15; i = (tid >> 4) + 16 * q; }
      const int gi = i0 + i, gk = k0 + k;
      ra[q] = (gi < g.M && gk < g.K) ? (TA ? A[(long long)gk * g.lda + gi] : A[(long long)gi * g.lda + gk]) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int j, k;
      if (TB) { k = tid & 15; j = (tid >> 4) + 16 * q; }
      else { j = tid & 63; k = (tid >> 6) + 4 * q; }
      const int gj = j0 + j, gk = k0 + k;
      rb[q] = (gj < g.N && gk < g.K) ? (TB ? B[(long long)gj * g.ldb + gk] : B[(long long)gk * g.ldb + gj]) : 0.0;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int q = 0; q < AQ; q++) {
      int i, k;
      if (TA) { i = tid & (BM - 1); k = (tid / BM) + (NT / BM) * q; }
      else { k = tid & 15; i = (tid >> 4) + 16 * q; }
      As[buf][k * LDAS + i] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int j, k;
      if (TB) { k = tid & 15; j = (tid >> 4) + 16 * q; }
      else { j = tid & 63; k = (tid >> 6) + 4 * q; }
      Bs[buf][k * LDBS + j] = rb[q];
    }
  };
  d4 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};
  int nk = (g.K + BK - 1) / BK;
  if (g.k_tri == 3 && (i0 + BM) / BK < nk) nk = (i0 + BM) / BK;
  int kt0 = 0;
  if (g.k_tri == 1) kt0 = j0 / BK;
  else if (g.k_tri == 2) kt0 = (i0 > j0 ? i0 : j0) / BK;
  if (kt0 > nk) kt0 = nk;
  if (kt0 < nk) {
    load_tiles(kt0 * BK);
    store_tiles(kt0 & 1);
  }
  __syncthreads();
  for (int kt = kt0; kt < nk; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    const double* as = As[cur] + fk * LDAS + (BM / 2) * wm + fi;
    const double* bs = Bs[cur] + fk * LDBS + 32 * wn + fi;
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      double af[MT], bf[2];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) af[mt] = as[kk * 4 * LDAS + 16 * mt];
#pragma unroll
      for (int nt = 0; nt < 2; nt++) bf[nt] = bs[kk * 4 * LDBS + 16 * nt];
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const int gj = j0 + 32 * wn + 16 * nt + (lane & 15);
      if (gj >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + (BM / 2) * wm + 16 * mt + (lane >> 4) + 4 * r;
        if (gi < g.M) {
          double* cp = C + (long long)gi * g.ldc + gj;
          double v = g.alpha * acc[mt][nt][r];
          if (g.epi_u) v = g.alpha * (acc[mt][nt][r] - g.epi_u[z1 * g.epi_s + gi] * g.epi_v[z1 * g.epi_s + gj]);
          *cp = (g.beta == 0.0) ? v : (v + g.beta * (*cp));
          if (g.mirror && i0 != j0) C[(long long)gj * g.ldc + gi] = v;
        }
      }
    }
}

// Full 64 x 64 tiles, op(B) = B, 16-byte global loads, prefetch distance two (the treatment of ld::k_gemm_v, large_d.hip:
// the A tile keeps its HBM orientation in LDS -- [i][18] for row-major A, k-major for A^T).  Same k order per output
// element as k_gemm_b, i.e. the same bits.
// (BM = 128 -- 57 KB of LDS, two workgroups per CU -- measured slower than 64 for every product of the energy phase: D = 1024, 33 grid
//  points: energy terms 4.57 against 4.26 ms, gradient 1.50 against 1.39 ms; not dispatched)
template <bool TA, int BM = 64>
__global__ void __launch_bounds__(NT) k_gemm_bv(GemmB g) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int LDK = 18, LDT = BM + 16, LDBS = BN + 16;
  constexpr int ASZ = TA ? BK * LDT : BM * LDK;
  constexpr int MT = BM / 32, AV = BM * BK / 2 / NT;
  __shared__ __attribute__((aligned(16))) double As[2][ASZ];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * LDBS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int i0, j0;
  if (!tile_origin(g, BM, i0, j0)) return;
  const double *A, *B;
  double* C;
  const long long z1 = batch_origin(g, A, B, C);
  const int fi = lane & 15, fk = lane >> 4;
  // persistent operand pointers, advanced by one k-tile per load (see ld::k_gemm_v)
  const double* pA[AV];
  const double* pB[2];
  int sa[AV], sb[2];
#pragma unroll
  for (int q = 0; q < AV; q++) {
    if (TA) {
      const int i2 = tid & (BM / 2 - 1), k = tid / (BM / 2) + (NT / (BM / 2)) * q;
      pA[q] = A + (long long)k * g.lda + i0 + 2 * i2;
      sa[q] = k * LDT + 2 * i2;
    } else {
      const int kp = tid & 7, i = (tid >> 3) + 32 * q;
      pA[q] = A + (long long)(i0 + i) * g.lda + 2 * kp;
      sa[q] = i * LDK + 2 * kp;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int j2 = tid & 31, k = (tid >> 5) + 8 * q;
    pB[q] = B + (long long)k * g.ldb + j0 + 2 * j2;
    sb[q] = k * LDBS + 2 * j2;
  }
  // k_tri = 1 (op(B) lower triangular: the k range of a tile starts at its first column): the k loop runs DOWN from the common end.  The
  // workgroups of a tile row share the row's A tiles; going up, each starts at its own k and they never read the same A tile at the same
  // time (D = 4096: up to 113 GB of A reads for 3.5 GB of A); going down they start together like the workgroups of a full product.
  const bool desc = g.k_tri == 1 && g.k_down;
  const long long astep0 = TA ? (long long)BK * g.lda : (long long)BK, bstep0 = (long long)BK * g.ldb;
  const long long astep = desc ? -astep0 : astep0, bstep = desc ? -bstep0 : bstep0;
  // PF register sets of loads in flight, none under a branch (ld::k_gemm_v: a branch makes the wait counts of its two paths merge
  // to the conservative one -- the LDS stores of one set then wait for the loads just issued into another); behind the last tile of
  // this workgroup's k range the pointers stay (the surplus loads re-read that tile and are never multiplied)
  constexpr int PF = 4;
  d2 ra[PF][AV], rb[PF][2];
  int advances = 0;
  auto load_tiles = [&](d2 (&xa)[AV], d2 (&xb)[2]) {       // loads the NEXT k-tile
    const long long as_ = advances > 0 ? astep : 0, bs_ = advances > 0 ? bstep : 0;
    advances--;
#pragma unroll
    for (int q = 0; q < AV; q++) { xa[q] = *reinterpret_cast<const d2*>(pA[q]); pA[q] += as_; }
#pragma unroll
    for (int q = 0; q < 2; q++) { xb[q] = *reinterpret_cast<const d2*>(pB[q]); pB[q] += bs_; }
  };
  auto store_tiles = [&](int buf, const d2 (&xa)[AV], const d2 (&xb)[2]) {
#pragma unroll
    for (int q = 0; q < AV; q++) *reinterpret_cast<d2*>(&As[buf][sa[q]]) = xa[q];
#pragma unroll
    for (int q = 0; q < 2; q++) *reinterpret_cast<d2*>(&Bs[buf][sb[q]]) = xb[q];
  };
  d4 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};
  auto compute = [&](int cur) {
    // (volatile, address space stated: plain ds_read_b64 -- 2 LDS cycles, 64 banks -- instead of the compiler's ds_read2_b64 pairs, see large_d.hip)
    typedef const volatile double __attribute__((address_space(3)))* frag_ptr;
    frag_ptr as = (frag_ptr)(TA ? As[cur] + fk * LDT + (BM / 2) * wm + fi : As[cur] + ((BM / 2) * wm + fi) * LDK + fk);
    frag_ptr bs = (frag_ptr)(Bs[cur] + fk * LDBS + 32 * wn + fi);
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      double af[MT], bf[2];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) af[mt] = TA ? as[kk * 4 * LDT + 16 * mt] : as[16 * mt * LDK + kk * 4];
#pragma unroll
      for (int nt = 0; nt < 2; nt++) bf[nt] = bs[kk * 4 * LDBS + 16 * nt];
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    }
  };
  int nk = g.K / BK;
  if (g.k_tri == 3 && (i0 + BM) / BK < nk) nk = (i0 + BM) / BK;
  int kt0 = 0;
  if (g.k_tri == 1) kt0 = j0 / BK;
  else if (g.k_tri == 2) kt0 = (i0 > j0 ? i0 : j0) / BK;
  if (kt0 > nk) kt0 = nk;
  const int kfirst = desc ? nk - 1 : kt0;                  // (nk - kt0 tiles either way)
#pragma unroll
  for (int q = 0; q < AV; q++) pA[q] += (long long)kfirst * astep0;
#pragma unroll
  for (int q = 0; q < 2; q++) pB[q] += (long long)kfirst * bstep0;
  const int ntile = nk - kt0;                              // k-tiles of this workgroup (k_tri: its triangular operand starts later)
  if (ntile > 0) {
    static_assert(PF % 2 == 0, "LDS buffer parity is static in the unrolled loop");
    advances = ntile - 1;
#pragma unroll
    for (int u = 0; u < PF; u++) load_tiles(ra[u], rb[u]);
    store_tiles(0, ra[0], rb[0]);
    __syncthreads();
    const int main_tiles = ntile / PF * PF;
    for (int t = 0; t < main_tiles; t += PF) {
#pragma unroll
      for (int u = 0; u < PF; u++) {
        load_tiles(ra[u], rb[u]);
        compute(u & 1);
        store_tiles((u + 1) & 1, ra[(u + 1) % PF], rb[(u + 1) % PF]);
        __syncthreads();
      }
    }
#pragma unroll
    for (int u = 0; u < PF - 1; u++) {                     // the last ntile % PF tiles: already in the register sets
      if (main_tiles + u >= ntile) break;
      compute(u & 1);
      store_tiles((u + 1) & 1, ra[(u + 1) % PF], rb[(u + 1) % PF]);
      __syncthreads();
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const int gj = j0 + 32 * wn + 16 * nt + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + (BM / 2) * wm + 16 * mt + (lane >> 4) + 4 * r;
        double* cp = C + (long long)gi * g.ldc + gj;
        double v = g.alpha * acc[mt][nt][r];
        if (g.epi_u) v = g.alpha * (acc[mt][nt][r] - g.epi_u[z1 * g.epi_s + gi] * g.epi_v[z1 * g.epi_s + gj]);
        *cp = (g.beta == 0.0) ? v : (v + g.beta * (*cp));
        if (g.mirror && i0 != j0) C[(long long)gj * g.ldc + gi] = v;
      }
    }
}

// nb: grid points; nsub > 1: that many independent sub-problems per grid point (GemmB::nb1 and the second-level strides are set here)
hipError_t gemm_b(bool ta, bool tb, GemmB g, int nb, hipStream_t st, int nsub = 1) {
  if (g.M <= 0 || g.N <= 0 || nb <= 0 || nsub <= 0) return hipSuccess;
  g.nb1 = nsub > 1 ? nb : 0;
  dim3 grid((g.N + BN - 1) / BN, (g.M + 63) / 64, nb * nsub);
  static const bool plain_map = [] { const char* e = getenv("VGPA_LDE_TILE_MAP"); return e && e[0] == '0'; }();
  static const bool k_up = [] { const char* e = getenv("VGPA_LDE_K_DOWN"); return e && e[0] == '0'; }();
  g.k_down = k_up ? 0 : 1;
  g.tile_map = 0;
  if (!plain_map && g.lower_only && g.M == g.N) {
    g.tile_map = 2;
    grid.x = grid.y * (grid.y + 1) / 2;
    grid.y = 1;
  } else if (!plain_map && g.k_tri != 0 && grid.x > 1) {
    g.tile_map = 1;
  }
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  static const bool scalar_loads = [] { const char* e = getenv("VGPA_GEMM_SCALAR_LOADS"); return e && e[0] == '1'; }();
  const bool vec = !tb && !scalar_loads && g.M % 64 == 0 && g.N % BN == 0 && g.K % BK == 0 && g.lda % 2 == 0 &&
                   g.ldb % 2 == 0 && g.sA % 2 == 0 && g.sB % 2 == 0 && al16(g.A) && al16(g.B) &&
                   (nsub == 1 || (g.sA2 % 2 == 0 && g.sB2 % 2 == 0));
  if (vec) {
    if (ta) hipLaunchKernelGGL((k_gemm_bv<true>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((k_gemm_bv<false>), grid, dim3(NT), 0, st, g);
    return hipGetLastError();
  }
  if (ta && tb) hipLaunchKernelGGL((k_gemm_b<true, true, 64>), grid, dim3(NT), 0, st, g);
  else if (ta) hipLaunchKernelGGL((k_gemm_b<true, false, 64>), grid, dim3(NT), 0, st, g);
  else if (tb) hipLaunchKernelGGL((k_gemm_b<false, true, 64>), grid, dim3(NT), 0, st, g);
  else hipLaunchKernelGGL((k_gemm_b<false, false, 64>), grid, dim3(NT), 0, st, g);
  return hipGetLastError();
}

// ---- 64 x 64 diagonal block: L_JJ = chol(C_JJ) (in place, strict upper zeroed) and X_JJ = L_JJ^-1 ------------------
// One wave per (grid point, block).  Rows / columns beyond D are treated as an identity block.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double lane_value(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(64) k_diag64(int D, int J, double* Cb, double* Xb, long long strideC, long long strideX,
                                               int32_t* status, int status_stride_log) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int LD = NBLK + 1;
  double* Lm = smem;             // [64][65]
  double* Xm = Lm + NBLK * LD;   // [64][65]
  double* rd = Xm + NBLK * LD;   // [64]
  const int l = threadIdx.x, b = blockIdx.x;
  double* C = Cb + (long long)b * strideC;
  double* X = Xb + (long long)b * strideX;
  const int r0 = J * NBLK;
  const int nv = (D - r0 < NBLK) ? (D - r0) : NBLK;      // valid rows / columns of this block
  for (int e = l; e < NBLK * NBLK; e += 64) {
    const int r = e >> 6, c = e & 63;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < nv && c < nv) v = C[(long long)(r0 + r) * D + r0 + c];
    Lm[r * LD + c] = v;
  }
  wave_sync();
  bool bad = false;
  for (int j0 = 0; j0 < NBLK && !bad; j0 += 4) {
    const double* rowi = Lm + l * LD;
    double s0 = rowi[j0], s1 = rowi[j0 + 1], s2 = rowi[j0 + 2], s3 = rowi[j0 + 3];
    const double* p0 = Lm + j0 * LD;
    const double* p1 = p0 + LD; const double* p2 = p1 + LD; const double* p3 = p2 + LD;
#pragma unroll 4
    for (int k = 0; k < j0; k++) {
      const double av = rowi[k];
      s0 = __builtin_fma(-av, p0[k], s0); s1 = __builtin_fma(-av, p1[k], s1);
      s2 = __builtin_fma(-av, p2[k], s2); s3 = __builtin_fma(-av, p3[k], s3);
    }
    double lq[4], sq[4] = {s0, s1, s2, s3};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = j0 + q;
      double s = sq[q];
#pragma unroll
      for (int q2 = 0; q2 < q; q2++) s = __builtin_fma(-lq[q2], lane_value(lq[q2], j), s);
      const double piv = lane_value(s, j);
      if (!(piv > 0.0)) bad = true;
      const double rdv = rsqrt(piv), d = piv * rdv;
      lq[q] = (l > j) ? s * rdv : 0.0;
      Lm[l * LD + j] = (l > j) ? lq[q] : ((l == j) ? d : 0.0);
      if (l == j) rd[j] = rdv;
    }
    wave_sync();
  }
  if (bad) {
    if (l == 0) atomicOr(status + (b >> status_stride_log), 1);
    return;
  }
  // X = L^-1, lane = column
  for (int e = l; e < NBLK * LD; e += 64) Xm[e] = 0.0;
  wave_sync();
  for (int i0 = 0; i0 < NBLK; i0 += 4) {
    const double* r0p = Lm + i0 * LD;
    const double* r1p = r0p + LD; const double* r2p = r1p + LD; const double* r3p = r2p + LD;
    double s0 = (i0 == l) ? 1.0 : 0.0, s1 = (i0 + 1 == l) ? 1.0 : 0.0, s2 = (i0 + 2 == l) ? 1.0 : 0.0,
           s3 = (i0 + 3 == l) ? 1.0 : 0.0;
    const double* xc = Xm + l;
#pragma unroll 4
    for (int k = 0; k < i0; k++) {
      const double xv = xc[k * LD];
      s0 = __builtin_fma(-r0p[k], xv, s0); s1 = __builtin_fma(-r1p[k], xv, s1);
      s2 = __builtin_fma(-r2p[k], xv, s2); s3 = __builtin_fma(-r3p[k], xv, s3);
    }
    const double x0 = s0 * rd[i0];
    s1 = __builtin_fma(-r1p[i0], x0, s1);
    const double x1 = s1 * rd[i0 + 1];
    s2 = __builtin_fma(-r2p[i0], x0, s2); s2 = __builtin_fma(-r2p[i0 + 1], x1, s2);
    const double x2 = s2 * rd[i0 + 2];
    s3 = __builtin_fma(-r3p[i0], x0, s3); s3 = __builtin_fma(-r3p[i0 + 1], x1, s3); s3 = __builtin_fma(-r3p[i0 + 2], x2, s3);
    const double x3 = s3 * rd[i0 + 3];
    double* xw = Xm + i0 * LD + l;
    xw[0] = x0; xw[LD] = x1; xw[2 * LD] = x2; xw[3 * LD] = x3;
  }
  wave_sync();
  for (int e = l; e < NBLK * NBLK; e += 64) {
    const int r = e >> 6, c = e & 63;
    if (r < nv && c < nv) {
      C[(long long)(r0 + r) * D + r0 + c] = Lm[r * LD + c];
      X[(long long)(r0 + r) * D + r0 + c] = Xm[r * LD + c];
    }
  }
}

// The same block on the matrix cores: the one-wave Cholesky and blocked forward substitution of the D <= 64 energy kernel
// (energy.hip::k_energy_l96 phases 1 and 4, NB = 16; chol_wave.h) -- panel updates and the rows of L^-1 as chains of
// v_mfma_f64_4x4x4_4b, every lane factoring the 4 x 4 diagonal block of its panel itself.  The vector-ALU form above walks
// ~2 x 1 900 dependent fused multiply-adds per lane behind LDS reads: 63 us per launch, and D / 64 launches in sequence whatever the
// batch; this one takes a quarter of that.
__global__ void __launch_bounds__(64) k_diag64m(int D, int J, double* Cb, double* Xb, long long strideC, long long strideX,
                                                int32_t* status, int status_stride_log) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NB = NBLK / 4, LD = NBLK + 1;
  double* Lm = smem;             // [64][65]
  double* Xm = Lm + NBLK * LD;   // [64][65]
  double* xdiag = Xm + NBLK * LD;      // [16][16]: inverses of the 4 x 4 diagonal blocks
  double* rdv = xdiag + 16 * NB;       // [64]: 1 / L[i][i]
  const int l = threadIdx.x, bi = blockIdx.x;
  double* C = Cb + (long long)bi * strideC;
  double* X = Xb + (long long)bi * strideX;
  const int r0 = J * NBLK;
  const int nv = (D - r0 < NBLK) ? (D - r0) : NBLK;      // valid rows / columns of this block
  for (int e = l; e < NBLK * NBLK; e += 64) {
    const int r = e >> 6, c = e & 63;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < nv && c < nv) v = C[(long long)(r0 + r) * D + r0 + c];
    Lm[r * LD + c] = v;
  }
  wave_sync();
  const int r4 = l >> 4, c4 = l & 3, b = (l >> 2) & 3;
  double myrd = 1.0;                       // 1 / L[l][l]
  const double* lrow_b = Lm + (4 * b + c4) * LD + r4;      // + 4 (p + 4u) LD: A-operand rows of block b
  double* lout_b = Lm + (4 * b + r4) * LD + c4;            // + 4 (p + 4u) LD + 4p: output position of block b
#pragma unroll
  for (int p = 0; p < NB; p++) {
    const int j0 = 4 * p;
    if (p > 0) {
      const double* brow = Lm + (j0 + c4) * LD + r4;
#pragma unroll
      for (int u = 0; p + 4 * u < NB; u++) {
        const bool full = p + 4 * u + 3 < NB;                // compile-time: all four block-rows of the unit exist
        const bool rowok = full || (p + 4 * u + b < NB);
        const double* arow = full ? lrow_b + 4 * (p + 4 * u) * LD : Lm + (4 * (rowok ? p + 4 * u + b : NB - 1) + c4) * LD + r4;
        double uacc = 0.0;
#pragma unroll
        for (int kk = 0; kk < p; kk++) uacc = __builtin_amdgcn_mfma_f64_4x4x4f64(arow[4 * kk], brow[4 * kk], uacc, 0, 0, 0);
        if (rowok) lout_b[4 * (p + 4 * u) * LD + j0] -= uacc;
      }
      wave_sync();
    }
    cholw::chol_panel_pivots<LD>(Lm, myrd, j0, l, l, true);
    wave_sync();
  }
  if (__any(!(myrd > 0.0 && myrd < __builtin_inf()))) {
    if (l == 0) atomicOr(status + (bi >> status_stride_log), 1);
    return;
  }
  rdv[l] = myrd;
  wave_sync();
  if (l < NB) {
    const double* tb = Lm + (4 * l) * LD + 4 * l;
    const double x00 = rdv[4 * l], x11 = rdv[4 * l + 1], x22 = rdv[4 * l + 2], x33 = rdv[4 * l + 3];
    const double t10 = tb[LD], t20 = tb[2 * LD], t21 = tb[2 * LD + 1], t30 = tb[3 * LD], t31 = tb[3 * LD + 1], t32 = tb[3 * LD + 2];
    const double x10 = -(t10 * x00) * x11;
    const double x21 = -(t21 * x11) * x22;
    const double x32 = -(t32 * x22) * x33;
    const double x20 = -(t20 * x00 + t21 * x10) * x22;
    const double x31 = -(t31 * x11 + t32 * x21) * x33;
    const double x30 = -(t30 * x00 + t31 * x10 + t32 * x20) * x33;
    double* xo = xdiag + 16 * l;
    xo[0] = x00; xo[1] = 0.0; xo[2] = 0.0; xo[3] = 0.0;
    xo[4] = x10; xo[5] = x11; xo[6] = 0.0; xo[7] = 0.0;
    xo[8] = x20; xo[9] = x21; xo[10] = x22; xo[11] = 0.0;
    xo[12] = x30; xo[13] = x31; xo[14] = x32; xo[15] = x33;
  }
  wave_sync();
  // X = L^-1 by blocked forward substitution: for block-row I and the unit of column blocks J_b = 4u + b, T_b = sum_{K < I}
  // L[I][K] X[K][J_b] accumulates in the MFMA result register, which is laid out as the B-operand of X[I][J_b] = -inv(L[I][I]) T_b
  const double* l4_a = Lm + c4 * LD + r4;                  // + 4 I LD + 4 K : A-operand L[4I + c4][4K + r4]
  const double* xd_a = xdiag + 4 * c4 + r4;                // + 16 I
  const double* xd_d = xdiag + 4 * r4 + c4;
#pragma unroll
  for (int I = 0; I < NB; I++) {
    const double* arow = l4_a + 4 * I * LD;
    const double xd = xd_a[16 * I];
    const double xdd = xd_d[16 * I];
#pragma unroll
    for (int u = 0; 4 * u < NB; u++) {
      const int Jb = 4 * u + b;
      double* xcol = Xm + r4 * LD + 4 * Jb + c4;
      double xo = 0.0;
      if (4 * u <= I) {                                      // compile-time: some block of the unit is on / below the diagonal
        double tacc = 0.0;
#pragma unroll
        for (int K = 4 * u; K < I; K++) tacc = __builtin_amdgcn_mfma_f64_4x4x4f64(arow[4 * K], xcol[4 * K * LD], tacc, 0, 0, 0);
        const double prod = __builtin_amdgcn_mfma_f64_4x4x4f64(xd, tacc, 0.0, 0, 0, 0);
        xo = (Jb < I) ? -prod : ((Jb == I) ? xdd : 0.0);
      }
      xcol[4 * I * LD] = xo;                                 // X[4I + r4][4 J_b + c4]
    }
    wave_sync();
  }
  for (int e = l; e < NBLK * NBLK; e += 64) {
    const int r = e >> 6, c = e & 63;
    if (r < nv && c < nv) {
      C[(long long)(r0 + r) * D + r0 + c] = Lm[r * LD + c];
      X[(long long)(r0 + r) * D + r0 + c] = Xm[r * LD + c];
    }
  }
}

// ---- small batched kernels ---------------------------------------------------------------------------------------
// C = c * S_t ; X = 0
__global__ void __launch_bounds__(NT) k_prep(int D, double cfac, const double* S, double* C, double* X, long long sW) {
  const long long DD = (long long)D * D;
  const double* s = S + (long long)blockIdx.y * DD;
  double* c = C + (long long)blockIdx.y * sW;
  double* x = X + (long long)blockIdx.y * sW;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) {
    c[e] = cfac * s[e];
    x[e] = 0.0;
  }
}

// zero the strict upper triangle of L (block rows above the diagonal still hold c*S / stale updates)
__global__ void __launch_bounds__(NT) k_zero_upper(int D, double* C, long long sW) {
  double* c = C + (long long)blockIdx.y * sW;
  const long long DD = (long long)D * D;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) {
    const int r = (int)(e / D), cc = (int)(e - (long long)r * D);
    if (cc > r) c[e] = 0.0;
  }
}

// y = op(A) x, one wave per output row; TA: y_i = sum_k A[k][i] x_k (rows k >= kmin_is_i ? i : 0)
__global__ void __launch_bounds__(NT) k_matvec(int D, int ta, const double* A, long long sA, const double* x, long long sx,
                                               double* y, long long sy, double scale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (NT / 64) + wave;
  if (i >= D) return;
  const double* a = A + (long long)blockIdx.y * sA;
  const double* xv = x + (long long)blockIdx.y * sx;
  double s = 0.0;
  if (ta) for (int k = lane; k < D; k += 64) s = __builtin_fma(a[(long long)k * D + i], xv[k], s);
  else for (int k = lane; k < D; k += 64) s = __builtin_fma(a[(long long)i * D + k], xv[k], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) y[(long long)blockIdx.y * sy + i] = scale * s;
}

// y = scale * A^T x with coalesced row reads: block = 64 columns (lane = column), its four waves take the rows k = wave (mod 4), four
// independent partial sums each; (the one-wave-per-output form above reads a column with one cache line per lane: 285 us per 33 grid
// points at D = 1024 against 73 us for A x)
__global__ void __launch_bounds__(NT) k_matvec_t(int D, const double* A, long long sA, const double* x, long long sx, double* y,
                                                 long long sy, double scale) {
  __shared__ double red[NT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 64 + lane;
  const int ic = i < D ? i : D - 1;
  const double* a = A + (long long)blockIdx.y * sA + ic;
  const double* xv = x + (long long)blockIdx.y * sx;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  int k = wave;
  for (; k + 12 < D; k += 16) {
#pragma unroll
    for (int u = 0; u < 4; u++) s[u] = __builtin_fma(a[(long long)(k + 4 * u) * D], xv[k + 4 * u], s[u]);
  }
  for (; k < D; k += 4) s[0] = __builtin_fma(a[(long long)k * D], xv[k], s[0]);
  red[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (wave == 0 && i < D) y[(long long)blockIdx.y * sy + i] = scale * ((red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]));
}

__device__ __forceinline__ int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// v_p for every sigma point p (thread = sigma point, sequential sliding window over the components i of ONE segment: gridDim.z
// segments of `seg` components each -- one thread per sigma point walked all D rows of L and G with a memory round trip per row:
// 529 us per 33 grid points at D = 1024, 3.8 x the time its bytes take).  nseg > 1: the partial sums go to vpart[seg][t][p] and
// k_vsum adds them in the order of the segments.
__global__ void __launch_bounds__(NT) k_resid(int D, double theta, const double* L, const double* G, long long sW,
                                              const double* m, const double* b, long long sb, const double* am,
                                              const double* isg, double* v, long long sv, int seg) {
  const int M = 2 * D + 1;
  const int p = blockIdx.x * NT + threadIdx.x;
  if (p >= M) return;
  const int t = blockIdx.y;
  const double* Lm = L + (long long)t * sW;
  const double* Gm = G + (long long)t * sW;
  const double* mv = m + (long long)t * D;
  const double* bv = b + (long long)t * sb;
  const double* av = am + (long long)t * D;
  auto col_of = [&](int q) { return q == 0 ? 0 : (q <= D ? q - 1 : q - 1 - D); };
  auto sgn_of = [&](int q) { return q == 0 ? 0.0 : (q <= D ? 1.0 : -1.0); };
  auto chi = [&](int q, int i) { return mv[i] + sgn_of(q) * Lm[(long long)i * D + col_of(q)]; };
  const int rp = col_of(p);
  const double sp = sgn_of(p);
  const int pm = wrapi(p - 1, M), pp = wrapi(p + 1, M);
  // flat roll (quirk Q1): component i - 2 / i - 1 of the first two rows belong to sigma point p - 1, i + 1 of the last row to p + 1
  auto at = [&](int i) { return i < 0 ? chi(pm, i + D) : (i < D ? chi(p, i) : chi(pp, i - D)); };
  const int i0 = blockIdx.z * seg, i1 = (i0 + seg < D) ? (i0 + seg) : D;
  double xm2 = at(i0 - 2), xm1 = at(i0 - 1), x0 = at(i0), x1 = at(i0 + 1);
  double acc = 0.0;
  for (int i = i0; i < i1; i++) {
    const double lin = av[i] + sp * Gm[(long long)i * D + rp];
    const double res = ((x1 - xm2) * xm1 - x0 + theta) + lin - bv[i];
    acc = __builtin_fma(isg[i], res * res, acc);
    xm2 = xm1; xm1 = x0; x0 = x1;
    const int in = i + 2;
    x1 = (in < D) ? chi(p, in) : chi(pp, in - D);
  }
  v[((long long)blockIdx.z * gridDim.y + t) * sv + p] = acc;
}

// v[t][p] = sum over the segments, in their order
__global__ void __launch_bounds__(NT) k_vsum(int M, int nseg, int nb, const double* vpart, double* v) {
  const long long e = (long long)blockIdx.x * NT + threadIdx.x;
  if (e >= (long long)nb * M) return;
  double s = vpart[e];
  for (int q = 1; q < nseg; q++) s += vpart[(long long)q * nb * M + e];
  v[e] = s;
}

// e_t, delta, q from v; also <f> (E96_drift)
__global__ void __launch_bounds__(NT) k_finish(int D, double theta, const double* v, long long sv, const double* S,
                                               const double* m, double* e_t, double* dl, double* qq, double* Ef) {
  __shared__ double red[NT];
  const int t = blockIdx.x, tid = threadIdx.x;
  const double kappa = 1.05 * D, c = D + kappa, w0 = kappa / c, w1 = 1.0 / (2.0 * c);
  const double* vv = v + (long long)t * sv;
  double part = 0.0;
  for (int r = tid; r < D; r += NT) part += vv[1 + r] + vv[1 + D + r];
  red[tid] = part;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const double et = 0.5 * (w0 * vv[0] + w1 * red[0]);
  if (tid == 0) e_t[t] = et;
  const double* St = S + (long long)t * D * D;
  const double* mv = m + (long long)t * D;
  for (int r = tid; r < D; r += NT) {
    dl[(long long)t * D + r] = w1 * (vv[1 + r] - vv[1 + D + r]);
    qq[(long long)t * D + r] = 0.5 * c * (w1 * (vv[1 + r] + vv[1 + D + r])) - et;
    const int ip1 = wrapi(r + 1, D), im1 = wrapi(r - 1, D), im2 = wrapi(r - 2, D);
    const double cxx = St[(long long)ip1 * D + im1] - St[(long long)im2 * D + im1];
    Ef[(long long)t * D + r] = cxx + (mv[ip1] - mv[im2]) * mv[im1] - mv[r] + theta;
  }
}

// Integrands of the hyper-parameter members of Lorenz96.energy (lorenz_96.py:421-434; computed by the reference, consumed by
// nothing): hyp[t][i] = <f>_i + (A m)_i - b_i  (dEsde_dtheta) and hyp[t][D + i] = m_bar_i, the unscented mean of the squared
// residual of component i (dEsde_dSigma) -- the same residuals as k_resid, summed over the sigma points instead of over the
// components: thread = component, sequential over the 2D + 1 sigma points.  Flat roll (quirk Q1): the neighbours of flat index
// f = p D + i are f + 1, f - 1, f - 2 modulo M D.
__global__ void __launch_bounds__(NT) k_hyper(int D, double theta, const double* L, const double* G, long long sW, const double* m,
                                              const double* b, long long sb, const double* am, const double* Ef, double* hyp) {
  const int M = 2 * D + 1;
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= D) return;
  const int t = blockIdx.y;
  const double* Lm = L + (long long)t * sW;
  const double* Gm = G + (long long)t * sW;
  const double* mv = m + (long long)t * D;
  const double kappa = 1.05 * D, c = D + kappa, w0 = kappa / c, w1 = 1.0 / (2.0 * c);
  auto chi = [&](long long f) {          // element of the flattened sigma-point matrix
    const long long MD = (long long)M * D;
    if (f < 0) f += MD;
    if (f >= MD) f -= MD;
    const int q = (int)(f / D), ii = (int)(f - (long long)q * D);
    const int col = q == 0 ? 0 : (q <= D ? q - 1 : q - 1 - D);
    const double sg = q == 0 ? 0.0 : (q <= D ? 1.0 : -1.0);
    return mv[ii] + sg * Lm[(long long)ii * D + col];
  };
  const double bi = b[(long long)t * sb + i], ami = am[(long long)t * D + i];
  double acc = 0.0;
  for (int p = 0; p < M; p++) {
    const long long f = (long long)p * D + i;
    const int col = p == 0 ? 0 : (p <= D ? p - 1 : p - 1 - D);
    const double sg = p == 0 ? 0.0 : (p <= D ? 1.0 : -1.0);
    const double lin = ami + sg * Gm[(long long)i * D + col];
    const double res = ((chi(f + 1) - chi(f - 2)) * chi(f - 1) - chi(f) + theta) + lin - bi;
    acc += (p == 0 ? w0 : w1) * (res * res);
  }
  double* h = hyp + (long long)t * 2 * D;
  h[i] = Ef[(long long)t * D + i] + ami - bi;
  h[D + i] = acc;
}

// Y[k][j] = q_k X[k][j]
__global__ void __launch_bounds__(NT) k_scale_rows(int D, const double* X, const double* q, double* Y, long long sW) {
  const double* x = X + (long long)blockIdx.y * sW;
  double* y = Y + (long long)blockIdx.y * sW;
  const double* qv = q + (long long)blockIdx.y * D;
  const long long DD = (long long)D * D;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) y[e] = qv[e / D] * x[e];
}

// dense <df/dx> (E96_drift_dx)
__global__ void __launch_bounds__(NT) k_edf(int D, const double* m, double* Edf) {
  const double* mv = m + (long long)blockIdx.y * D;
  double* ed = Edf + (long long)blockIdx.y * D * D;
  const long long DD = (long long)D * D;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) {
    const int k = (int)(e / D), j = (int)(e - (long long)k * D);
    const int kp1 = wrapi(k + 1, D), km1 = wrapi(k - 1, D), km2 = wrapi(k - 2, D);
    double val = 0.0;
    if (j == k) val = -1.0;
    if (j == kp1) val = mv[km1];
    if (j == km2) val = -mv[km1];
    if (j == km1) val = mv[kp1] - mv[km2];
    ed[e] = val;
  }
}

// Q = diag(isg) (Edf + A) - 2 Psi, Edf recomputed from m (E96_drift_dx)
__global__ void __launch_bounds__(NT) k_grad_q(int D, const double* isg, const double* A, const double* m, const double* psi,
                                               double* Q) {
  const long long DD = (long long)D * D;
  const double* a = A + (long long)blockIdx.y * DD;
  const double* ps = psi + (long long)blockIdx.y * DD;
  const double* mv = m + (long long)blockIdx.y * D;
  double* q = Q + (long long)blockIdx.y * DD;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) {
    const int k = (int)(e / D), j = (int)(e - (long long)k * D);
    const int kp1 = wrapi(k + 1, D), km1 = wrapi(k - 1, D), km2 = wrapi(k - 2, D);
    double ed = 0.0;
    if (j == k) ed = -1.0;
    if (j == kp1) ed = mv[km1];
    if (j == km2) ed = -mv[km1];
    if (j == km1) ed = mv[kp1] - mv[km2];
    q[e] = isg[k] * (ed + a[e]) - 2.0 * ps[e];
  }
}

// dense Sigma^-1 (variational.py:320,332 take any inverse noise matrix): T = <df/dx>(m) + A and Q0 = -2 Psi, so that one
// batched product Q = Sigma^-1 T + Q0 finishes Q; w = -Ef - A m + b, so that u = Sigma^-1 w + lam
__global__ void __launch_bounds__(NT) k_grad_q_dense(int D, const double* A, const double* m, const double* psi, double* T, double* Q) {
  const long long DD = (long long)D * D;
  const double* a = A + (long long)blockIdx.y * DD;
  const double* ps = psi + (long long)blockIdx.y * DD;
  const double* mv = m + (long long)blockIdx.y * D;
  double* tq = T + (long long)blockIdx.y * DD;
  double* q = Q + (long long)blockIdx.y * DD;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) {
    const int k = (int)(e / D), j = (int)(e - (long long)k * D);
    const int kp1 = wrapi(k + 1, D), km1 = wrapi(k - 1, D), km2 = wrapi(k - 2, D);
    double ed = 0.0;
    if (j == k) ed = -1.0;
    if (j == kp1) ed = mv[km1];
    if (j == km2) ed = -mv[km1];
    if (j == km1) ed = mv[kp1] - mv[km2];
    tq[e] = ed + a[e];
    q[e] = -2.0 * ps[e];
  }
}
__global__ void __launch_bounds__(NT) k_grad_w(int D, const double* am, const double* b, const double* Ef, double* w) {
  const long long vo = (long long)blockIdx.y * D;
  for (int i = blockIdx.x * NT + threadIdx.x; i < D; i += gridDim.x * NT) w[vo + i] = -Ef[vo + i] - am[vo + i] + b[vo + i];
}
// gA = dt (QS - u m^T) in place on QS ; gB = dt u ; u = u0 + lam with u0 = Sigma^-1 w
__global__ void __launch_bounds__(NT) k_grad_fin_dense(int D, double dt, const double* u0, const double* m, const double* lam, double* gA,
                                                       double* gB) {
  const long long DD = (long long)D * D;
  const long long vo = (long long)blockIdx.y * D;
  double* g = gA + (long long)blockIdx.y * DD;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) {
    const int i = (int)(e / D), j = (int)(e - (long long)i * D);
    const double u = u0[vo + i] + lam[vo + i];
    g[e] = dt * (g[e] - u * m[vo + j]);
    if (j == 0) gB[vo + i] = dt * u;
  }
}

// gA = dt (QS - u m^T) in place on QS ; gB = dt u ; u_i = isg_i (-Ef_i - (A m)_i + b_i) + lam_i
__global__ void __launch_bounds__(NT) k_grad_fin(int D, double dt, const double* isg, const double* am, const double* b,
                                                 const double* m, const double* lam, const double* Ef, double* gA, double* gB) {
  const long long DD = (long long)D * D;
  const long long vo = (long long)blockIdx.y * D;
  double* g = gA + (long long)blockIdx.y * DD;
  for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < DD; e += (long long)gridDim.x * NT) {
    const int i = (int)(e / D), j = (int)(e - (long long)i * D);
    const double u = isg[i] * (-Ef[vo + i] - am[vo + i] + b[vo + i]) + lam[vo + i];
    g[e] = dt * (g[e] - u * m[vo + j]);
    if (j == 0) gB[vo + i] = dt * u;
  }
}

// u_i = isg_i (-Ef_i - (A m)_i + b_i) + lam_i ; gB = dt u  (the rank-one term itself rides in the epilogue of the product Q S: GemmB::epi_u)
__global__ void __launch_bounds__(NT) k_grad_u(int D, double dt, const double* isg, const double* am, const double* b, const double* lam,
                                               const double* Ef, double* u, double* gB) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= D) return;
  const long long vo = (long long)blockIdx.y * D;
  const double uv = isg[i] * (-Ef[vo + i] - am[vo + i] + b[vo + i]) + lam[vo + i];
  u[vo + i] = uv;
  gB[vo + i] = dt * uv;
}

// fork / join events of lde_energy's two streams (destroyed on every return path)
struct EventPair {
  hipEvent_t a = nullptr, b = nullptr;
  bool create() {
    return hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess;
  }
  ~EventPair() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

}  // namespace lde

namespace ld {

size_t lde_workspace_doubles(int D, int nb) {
  const size_t DD = (size_t)D * D;
  return (size_t)nb * (3 * DD + (size_t)lde::NBLK * D + 4 * (size_t)D + (2 * (size_t)D + 1));
}

// grid points per batch for a workspace budget in bytes: the diagonal-block kernel runs ONE wave per (grid point,
// 64 x 64 block) and D / 64 of them in sequence, so small batches leave the chip idle (D = 4096 with a 1 GB budget: 2)
int lde_batch(int D, double budget_bytes) {
  const double per_t = 3.1 * D * D * 8.0;
  int nb = (int)(budget_bytes / per_t);
  return nb < 1 ? 1 : (nb > 128 ? 128 : nb);
}

#define LDE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

// Energy terms of Np grid points of ONE problem.  Edf may be nullptr.
hipError_t lde_energy(int D, int Np, double theta, const double* isg, const double* A, const double* b, const double* m,
                      const double* S, double* e_t, double* Ef, double* Edf, double* dEm, double* dEs, int32_t* status,
                      double* ws, int nbmax, hipStream_t st, double* hyp, hipStream_t side) {
  using namespace lde;
  const long long DD = (long long)D * D;
  const int T = (D + NBLK - 1) / NBLK, M = 2 * D + 1;
  const double kappa = 1.05 * D, c = D + kappa;
  const size_t lds_diag = sizeof(double) * (2 * NBLK * (NBLK + 1) + NBLK);
  const size_t lds_diagm = sizeof(double) * (2 * NBLK * (NBLK + 1) + 16 * (NBLK / 4) + NBLK);
  (void)hipFuncSetAttribute((const void*)k_diag64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
  (void)hipFuncSetAttribute((const void*)k_diag64m, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diagm);
  static const bool diag_valu = [] { const char* e = getenv("VGPA_LDE_DIAG"); return e && e[0] == 'v'; }();
  static const bool no_halves = [] { const char* e = getenv("VGPA_LDE_TWO_STREAMS"); return e && e[0] == '0'; }();
  static const bool no_mirror = [] { const char* e = getenv("VGPA_LDE_SYRK_MIRROR"); return e && e[0] == '0'; }();
  static const int wpan = [] { const char* e = getenv("VGPA_LDE_PANEL"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
  static const bool by_rows = [] { const char* e = getenv("VGPA_LDE_INVERSE"); return e && e[0] == 'r'; }();
  const bool two_halves = side != nullptr && side != st && !no_halves;
  EventPair evs;                                      // (destroyed on every return path)
  if (two_halves && !evs.create()) return hipErrorOutOfMemory;
  hipEvent_t evP = evs.a, evB = evs.b;
  for (int t0 = 0; t0 < Np; t0 += nbmax) {
    const int nb = (Np - t0 < nbmax) ? (Np - t0) : nbmax;
    double* C = ws;                                   // [nb][D][D]  c*S -> L -> diag(q) X
    double* X = C + (size_t)nb * DD;                  // [nb][D][D]  L^-1
    double* G = X + (size_t)nb * DD;                  // [nb][D][D]  A L
    double* T1 = G + (size_t)nb * DD;                 // [nb][64][D]
    double* am = T1 + (size_t)nb * NBLK * D;          // [nb][D]
    double* dl = am + (size_t)nb * D;
    double* qq = dl + (size_t)nb * D;
    double* sp = qq + (size_t)nb * D;                 // spare [nb][D]
    double* vv = sp + (size_t)nb * D;                 // [nb][M]
    const double* At = A + (size_t)t0 * DD;
    const double* St = S + (size_t)t0 * DD;
    const double* mt = m + (size_t)t0 * D;
    const double* bt = b + (size_t)t0 * D;
    const int eg = (int)((DD + NT * 8 - 1) / (NT * 8));
    hipLaunchKernelGGL(k_prep, dim3(eg, nb), dim3(NT), 0, st, D, c, St, C, X, DD);
    // ---- blocked Cholesky and X = L^-1, both chains of short launches: k_diag64 is ONE wave per grid point and 50-70 us whatever the
    // batch (D / 64 of them in sequence), the panels and the first levels of the inverse are a few workgroups each.  With a side stream
    // the batch goes through them as two halves on two streams, so that one half's chain runs beside the other half's products (a
    // one-step look-ahead inside one batch was built first: a record / wait pair per step costs the main stream ~20 us, as much as it
    // hid at D = 1024).  Per grid point the same launches in the same order: the same bits.
    const int nh = (two_halves && nb >= 2) ? 2 : 1;
    const hipStream_t hs_[2] = {st, side};
    const int hb_[3] = {0, nh == 2 ? nb - nb / 2 : nb, nb};
    if (nh == 2) {
      LDE_TRY(hipEventRecord(evP, st));                      // fork behind k_prep
      LDE_TRY(hipStreamWaitEvent(side, evP, 0));
    }
    for (int J = 0; J < T; J++) {
      const int r1 = (J + 1) * NBLK;
      const int Mr = D - r1;
      const int kw = (D - J * NBLK < NBLK) ? (D - J * NBLK) : NBLK;
      for (int hh = nh - 1; hh >= 0; hh--) {
        const hipStream_t sh = hs_[hh];
        const int n = hb_[hh + 1] - hb_[hh];
        double* Ch = C + (size_t)hb_[hh] * DD;
        double* Xh = X + (size_t)hb_[hh] * DD;
        if (diag_valu) hipLaunchKernelGGL(k_diag64, dim3(n), dim3(64), lds_diag, sh, D, J, Ch, Xh, DD, DD, status, 30);
        else hipLaunchKernelGGL(k_diag64m, dim3(n), dim3(64), lds_diagm, sh, D, J, Ch, Xh, DD, DD, status, 30);
        if (r1 >= D) continue;
        GemmB p{};   // L[R,J] = C[R,J] X_JJ^T   (in place)
        p.M = Mr; p.N = kw; p.K = kw; p.A = Ch + (size_t)r1 * D + J * NBLK; p.lda = D; p.sA = DD;
        p.B = Xh + (size_t)(J * NBLK) * D + J * NBLK; p.ldb = D; p.sB = DD;
        p.C = Ch + (size_t)r1 * D + J * NBLK; p.ldc = D; p.sC = DD; p.alpha = 1.0; p.beta = 0.0;
        // in place: every workgroup owns a 64-row block of the panel, reads all of it before its epilogue writes it
        LDE_TRY(gemm_b(false, true, p, n, sh));
        // Trailing update in two levels: panel J updates the block columns of its own OUTER panel (`wpan` blocks) only; behind the outer
        // panel's last block everything right of it gets ONE update with all its columns (K = 64 wpan instead of wpan updates with
        // K = 64: the update is bound by reading and writing C -- 8 flop per byte at K = 64 -- and this is a quarter of the traffic).
        const int pend = ((J / wpan + 1) * wpan < T) ? (J / wpan + 1) * wpan * NBLK : D;      // first column behind the outer panel
        GemmB u{};   // C[R, r1:pend] -= L[R,J] L[r1:pend,J]^T  (lower tiles)
        u.M = Mr; u.N = pend - r1; u.K = kw; u.A = Ch + (size_t)r1 * D + J * NBLK; u.lda = D; u.sA = DD;
        u.B = u.A; u.ldb = D; u.sB = DD; u.C = Ch + (size_t)r1 * D + r1; u.ldc = D; u.sC = DD; u.alpha = -1.0; u.beta = 1.0;
        u.lower_only = 1;
        LDE_TRY(gemm_b(false, true, u, n, sh));
        if (r1 == pend && pend < D) {   // J was the outer panel's last block: C[R2,R2] -= L[R2,P] L[R2,P]^T
          const int p0 = (J / wpan) * wpan * NBLK;
          GemmB w{};
          w.M = D - pend; w.N = D - pend; w.K = pend - p0; w.A = Ch + (size_t)pend * D + p0; w.lda = D; w.sA = DD;
          w.B = w.A; w.ldb = D; w.sB = DD; w.C = Ch + (size_t)pend * D + pend; w.ldc = D; w.sC = DD; w.alpha = -1.0; w.beta = 1.0;
          w.lower_only = 1;
          LDE_TRY(gemm_b(false, true, w, n, sh));
        }
      }
      if (r1 >= D) break;
    }
    for (int hh = nh - 1; hh >= 0; hh--)
      hipLaunchKernelGGL(k_zero_upper, dim3(eg, hb_[hh + 1] - hb_[hh]), dim3(NT), 0, hs_[hh], D, C + (size_t)hb_[hh] * DD, DD);
    // ---- X = L^-1 by halves: with X_11, X_22 known, X_21 = -X_22 (L_21 X_11).  Level h (h = 1, 2, 4 ... blocks) joins the
    // neighbouring groups of h blocks; the groups of a level are independent and go into ONE launch per product (second batch level of
    // GemmB), so D / 64 = 16 | 64 takes 8 | 12 launches of growing products instead of 30 | 126 of one 64-row block each.  The
    // intermediate L_21 X_11 sits in G (free until G = A L) at the place of X_21.
    if (by_rows) {      // the block-row form of rounds 1-4, kept for the A/B measurement (VGPA_LDE_INVERSE=rows)
      for (int I = 1; I < T; I++) {
        const int r0 = I * NBLK;
        const int Mi = (D - r0 < NBLK) ? (D - r0) : NBLK;
        for (int hh = nh - 1; hh >= 0; hh--) {
          const int n = hb_[hh + 1] - hb_[hh];
          double* Ch = C + (size_t)hb_[hh] * DD;
          double* Xh = X + (size_t)hb_[hh] * DD;
          double* Th = T1 + (size_t)hb_[hh] * NBLK * D;
          GemmB a1{};  // T1 = L[I, 0:r0] X[0:r0, 0:r0]
          a1.M = Mi; a1.N = r0; a1.K = r0; a1.A = Ch + (size_t)r0 * D; a1.lda = D; a1.sA = DD; a1.B = Xh; a1.ldb = D; a1.sB = DD;
          a1.C = Th; a1.ldc = D; a1.sC = (long long)NBLK * D; a1.alpha = 1.0; a1.beta = 0.0;
          a1.k_tri = 1;                                  // X[0:r0, 0:r0] is lower triangular
          LDE_TRY(gemm_b(false, false, a1, n, hs_[hh]));
          GemmB a2{};  // X[I, 0:r0] = -X_II T1
          a2.M = Mi; a2.N = r0; a2.K = Mi; a2.A = Xh + (size_t)r0 * D + r0; a2.lda = D; a2.sA = DD; a2.B = Th; a2.ldb = D;
          a2.sB = (long long)NBLK * D; a2.C = Xh + (size_t)r0 * D; a2.ldc = D; a2.sC = DD; a2.alpha = -1.0; a2.beta = 0.0;
          LDE_TRY(gemm_b(false, false, a2, n, hs_[hh]));
        }
      }
    }
    for (int h = 1; h < T && !by_rows; h *= 2) {
      const int hs = h * NBLK;
      const int ngr = (T - h + 2 * h - 1) / (2 * h);           // groups whose second half exists
      const long long step2 = (long long)2 * hs * (D + 1);     // from one group's blocks to the next group's
      auto rows_of = [&](int g) { const int mid = (2 * g + 1) * hs, hi = (2 * g + 2) * hs; return (hi < D ? hi : D) - mid; };
      const int nfull = (rows_of(ngr - 1) == hs) ? ngr : ngr - 1;
      for (int part = 0; part < 2; part++) {
        const int g0 = part == 0 ? 0 : nfull, ng = part == 0 ? nfull : ngr - nfull;
        if (ng <= 0) continue;
        const int Mi = rows_of(g0);
        for (int hh = nh - 1; hh >= 0; hh--) {
          const int n = hb_[hh + 1] - hb_[hh];
          const size_t org = (size_t)g0 * (size_t)step2 + (size_t)hb_[hh] * DD;
          GemmB a1{};  // T = L_21 X_11
          a1.M = Mi; a1.N = hs; a1.K = hs; a1.A = C + org + (size_t)hs * D; a1.lda = D; a1.sA = DD; a1.sA2 = step2;
          a1.B = X + org; a1.ldb = D; a1.sB = DD; a1.sB2 = step2;
          a1.C = G + org + (size_t)hs * D; a1.ldc = D; a1.sC = DD; a1.sC2 = step2; a1.alpha = 1.0; a1.beta = 0.0;
          a1.k_tri = 1;                                          // X_11 is lower triangular
          LDE_TRY(gemm_b(false, false, a1, n, hs_[hh], ng));
          GemmB a2{};  // X_21 = -X_22 T
          a2.M = Mi; a2.N = hs; a2.K = Mi; a2.A = X + org + (size_t)hs * (D + 1); a2.lda = D; a2.sA = DD; a2.sA2 = step2;
          a2.B = G + org + (size_t)hs * D; a2.ldb = D; a2.sB = DD; a2.sB2 = step2;
          a2.C = X + org + (size_t)hs * D; a2.ldc = D; a2.sC = DD; a2.sC2 = step2; a2.alpha = -1.0; a2.beta = 0.0;
          a2.k_tri = 3;                                          // X_22 is lower triangular
          LDE_TRY(gemm_b(false, false, a2, n, hs_[hh], ng));
        }
      }
    }
    if (nh == 2) {
      LDE_TRY(hipEventRecord(evB, side));                      // join
      LDE_TRY(hipStreamWaitEvent(st, evB, 0));
    }
    // ---- G = A L ; A m
    GemmB gg{};
    gg.M = D; gg.N = D; gg.K = D; gg.A = At; gg.lda = D; gg.sA = DD; gg.B = C; gg.ldb = D; gg.sB = DD; gg.C = G; gg.ldc = D;
    gg.sC = DD; gg.alpha = 1.0; gg.beta = 0.0;
    gg.k_tri = 1;                                    // L is lower triangular (strict upper part zeroed above)
    LDE_TRY(gemm_b(false, false, gg, nb, st));
    hipLaunchKernelGGL(k_matvec, dim3((D + 3) / 4, nb), dim3(NT), 0, st, D, 0, At, DD, mt, (long long)D, am, (long long)D, 1.0);
    // ---- residuals, scalars
    {
      // segments of >= 128 components (D <= 255: one, the sums of rounds 1-4 bit for bit); the partial sums take T1's place
      const int nseg = D / 128 < 1 ? 1 : (D / 128 > 16 ? 16 : D / 128);
      const int seg = (D + nseg - 1) / nseg;
      hipLaunchKernelGGL(k_resid, dim3((M + NT - 1) / NT, nb, nseg), dim3(NT), 0, st, D, theta, C, G, DD, mt, bt, (long long)D, am,
                         isg, nseg > 1 ? T1 : vv, (long long)M, seg);
      if (nseg > 1)
        hipLaunchKernelGGL(k_vsum, dim3((unsigned)(((long long)nb * M + NT - 1) / NT)), dim3(NT), 0, st, M, nseg, nb, T1, vv);
    }
    hipLaunchKernelGGL(k_finish, dim3(nb), dim3(NT), 0, st, D, theta, vv, (long long)M, St, mt, e_t + t0, dl, qq,
                       Ef + (size_t)t0 * D);
    if (hyp)
      hipLaunchKernelGGL(k_hyper, dim3((D + NT - 1) / NT, nb), dim3(NT), 0, st, D, theta, C, G, DD, mt, bt, (long long)D, am,
                         Ef + (size_t)t0 * D, hyp + (size_t)t0 * 2 * D);
    // ---- dE/dm = c/2 X^T delta ; dE/dS = c/2 X^T diag(q) X
    hipLaunchKernelGGL(k_matvec_t, dim3((D + 63) / 64, nb), dim3(NT), 0, st, D, X, DD, dl, (long long)D, dEm + (size_t)t0 * D,
                       (long long)D, 0.5 * c);
    hipLaunchKernelGGL(k_scale_rows, dim3(eg, nb), dim3(NT), 0, st, D, X, qq, C, DD);
    GemmB sy{};
    sy.M = D; sy.N = D; sy.K = D; sy.A = X; sy.lda = D; sy.sA = DD; sy.B = C; sy.ldb = D; sy.sB = DD;
    sy.C = dEs + (size_t)t0 * DD; sy.ldc = D; sy.sC = DD; sy.alpha = 0.5 * c; sy.beta = 0.0;
    sy.k_tri = 2;                                    // X^T[i][k] = X[k][i] = 0 for k < i and (q X)[k][j] = 0 for k < j
    sy.lower_only = no_mirror ? 0 : 1;               // the result is symmetric: the tiles on and below the diagonal, mirrored on the way out
    sy.mirror = sy.lower_only;
    LDE_TRY(gemm_b(true, false, sy, nb, st));
    if (Edf) hipLaunchKernelGGL(k_edf, dim3(eg, nb), dim3(NT), 0, st, D, mt, Edf + (size_t)t0 * DD);
    LDE_TRY(hipGetLastError());
  }
  return hipSuccess;
}

// Gradient w.r.t. (A_t, b_t) for D > 64 (src/var_bayes/variational.py:263-288), diagonal Sigma^-1:
//   Q = Sigma^-1 (Edf + A) - 2 Psi ;  gLa = dt (Q S - u m^T) ;  u = Sigma^-1 (-Ef - A m + b) + lam ;  gLb = dt u
hipError_t lde_grad(int D, int Np, double dt, const double* isg, const double* A, const double* b, const double* m,
                    const double* S, const double* lam, const double* psi, const double* Ef, double* gA, double* gB,
                    double* ws, int nbmax, hipStream_t st, const double* isig_dense) {
  using namespace lde;
  const long long DD = (long long)D * D;
  for (int t0 = 0; t0 < Np; t0 += nbmax) {
    const int nb = (Np - t0 < nbmax) ? (Np - t0) : nbmax;
    double* Q = ws;                                    // [nb][D][D]
    double* T = Q + (size_t)nb * DD;                   // [nb][D][D]  dense Sigma^-1 only
    double* am = T + (size_t)nb * DD;                  // [nb][D]
    double* wv = am + (size_t)nb * D;                  // [nb][D]     dense only
    double* u0 = wv + (size_t)nb * D;                  // [nb][D]     dense only
    const int eg = (int)((DD + NT * 8 - 1) / (NT * 8));
    if (isig_dense) {
      hipLaunchKernelGGL(k_grad_q_dense, dim3(eg, nb), dim3(NT), 0, st, D, A + (size_t)t0 * DD, m + (size_t)t0 * D, psi + (size_t)t0 * DD, T, Q);
      GemmB q{};     // Q = Sigma^-1 T - 2 Psi   (Sigma^-1 shared by the grid points: stride 0)
      q.M = D; q.N = D; q.K = D; q.A = isig_dense; q.lda = D; q.sA = 0; q.B = T; q.ldb = D; q.sB = DD; q.C = Q; q.ldc = D; q.sC = DD;
      q.alpha = 1.0; q.beta = 1.0;
      LDE_TRY(gemm_b(false, false, q, nb, st));
      hipLaunchKernelGGL(k_matvec, dim3((D + 3) / 4, nb), dim3(NT), 0, st, D, 0, A + (size_t)t0 * DD, DD, m + (size_t)t0 * D,
                         (long long)D, am, (long long)D, 1.0);
      hipLaunchKernelGGL(k_grad_w, dim3((D + NT - 1) / NT, nb), dim3(NT), 0, st, D, am, b + (size_t)t0 * D, Ef + (size_t)t0 * D, wv);
      hipLaunchKernelGGL(k_matvec, dim3((D + 3) / 4, nb), dim3(NT), 0, st, D, 0, isig_dense, 0LL, wv, (long long)D, u0, (long long)D, 1.0);
      GemmB g{};
      g.M = D; g.N = D; g.K = D; g.A = Q; g.lda = D; g.sA = DD; g.B = S + (size_t)t0 * DD; g.ldb = D; g.sB = DD;
      g.C = gA + (size_t)t0 * DD; g.ldc = D; g.sC = DD; g.alpha = 1.0; g.beta = 0.0;
      LDE_TRY(gemm_b(false, false, g, nb, st));
      hipLaunchKernelGGL(k_grad_fin_dense, dim3(eg, nb), dim3(NT), 0, st, D, dt, u0, m + (size_t)t0 * D, lam + (size_t)t0 * D,
                         gA + (size_t)t0 * DD, gB + (size_t)t0 * D);
      LDE_TRY(hipGetLastError());
      continue;
    }
    hipLaunchKernelGGL(k_grad_q, dim3(eg, nb), dim3(NT), 0, st, D, isg, A + (size_t)t0 * DD, m + (size_t)t0 * D,
                       psi + (size_t)t0 * DD, Q);
    hipLaunchKernelGGL(k_matvec, dim3((D + 3) / 4, nb), dim3(NT), 0, st, D, 0, A + (size_t)t0 * DD, DD, m + (size_t)t0 * D,
                       (long long)D, am, (long long)D, 1.0);
    GemmB g{};
    g.M = D; g.N = D; g.K = D; g.A = Q; g.lda = D; g.sA = DD; g.B = S + (size_t)t0 * DD; g.ldb = D; g.sB = DD;
    g.C = gA + (size_t)t0 * DD; g.ldc = D; g.sC = DD; g.alpha = 1.0; g.beta = 0.0;
    static const bool fin_pass = [] { const char* e = getenv("VGPA_LDE_GRAD_EPILOGUE"); return e && e[0] == '0'; }();
    if (fin_pass) {      // rounds 1-4: the rank-one term and dt in a pass over the product (kept for the A/B measurement)
      LDE_TRY(gemm_b(false, false, g, nb, st));
      hipLaunchKernelGGL(k_grad_fin, dim3(eg, nb), dim3(NT), 0, st, D, dt, isg, am, b + (size_t)t0 * D, m + (size_t)t0 * D,
                         lam + (size_t)t0 * D, Ef + (size_t)t0 * D, gA + (size_t)t0 * DD, gB + (size_t)t0 * D);
    } else {             // gA = dt (Q S - u m^T) out of the product's epilogue
      hipLaunchKernelGGL(k_grad_u, dim3((D + NT - 1) / NT, nb), dim3(NT), 0, st, D, dt, isg, am, b + (size_t)t0 * D, lam + (size_t)t0 * D,
                         Ef + (size_t)t0 * D, u0, gB + (size_t)t0 * D);
      g.alpha = dt; g.epi_u = u0; g.epi_v = m + (size_t)t0 * D; g.epi_s = D;
      LDE_TRY(gemm_b(false, false, g, nb, st));
    }
    LDE_TRY(hipGetLastError());
  }
  return hipSuccess;
}
#undef LDE_TRY

}  // namespace ld
}  // namespace vgpa
