// Shared by large_d.hip and large_d_stage.hip: tile constants, the LDS fragment pointer, the argument block of a Runge-Kutta stage and
// its element-wise helpers.  (The one-kernel stages live in their own translation unit because they are compiled with
// -mllvm -amdgpu-mfma-vgpr-form -- vgpa_amd/build.py -- and the products are not: see the head of large_d_stage.hip.)
#pragma once
#include "vgpa_internal.h"

namespace vgpa {
namespace ld {

typedef double d4 __attribute__((ext_vector_type(4)));

// MFMA fragment reads from LDS go through a volatile pointer: that keeps them ds_read_b64 (one wave-instruction in 2 LDS cycles, 64
// banks: the tile layouts below are conflict-free for it).  Left to itself the compiler pairs neighbouring reads into ds_read2_b64, which
// the LDS serves as two accesses of 4 x 16 lanes over 32 banks (MI355X_MICROARCH.md, LDS table): 8 cycles at best, and 16 on the
// row-major tiles ([row][16 k + 2]: rows r and r + 8 then share banks) -- with 8 waves per CU that is as many LDS cycles per k-tile as
// the MFMA pipe has.
#ifndef VGPA_LDS_FRAG_VOLATILE
#define VGPA_LDS_FRAG_VOLATILE 1
#endif
#if VGPA_LDS_FRAG_VOLATILE
typedef const volatile double __attribute__((address_space(3)))* frag_ptr;      // (stated address space: a volatile generic pointer becomes flat loads)
#else
typedef const double __attribute__((address_space(3)))* frag_ptr;
#endif
__device__ __forceinline__ frag_ptr frag(const double* p) { return (frag_ptr)p; }

constexpr int BN = 64, BK = 16, NT = 256;   // block tile BM x 64 (BM = 128, or 64 when 128 would not fill the chip)
constexpr int LDBS = BN + 16;   // 80 = 16 (mod 32): conflict-free fragment reads

// ---------------------------------------------------------------------------------------------------------------
// Fused element-wise stage kernel on the row block [row0, row0 + Mp) of a D x D symmetric recursion.
//   R[r][j] = fwd ? (-(W[r][j]) - Wcol[j][r]) + E[r][j]          (E = Sigma)
//                 : (-E[r][j] + Wcol[j][r]) + W[r][j]            (E = G_stage = dEsde_dS or its mid-point)
//   K-slot bookkeeping follows the reference's expression order:
//     kstore 1: K1 = R     2: K23 = R     3: K23 += R
//     final 0 : Xn = base +/- cx * R                                   (next stage state; + forward, - backward)
//     final 1 : out = base +/- cf * R                        [+ J]     (Euler, RK2 last stage)
//     final 2 : out = base +/- cf * (K1 + R)                 [+ J]     (Heun, cf = dt/2)
//     final 3 : out = base +/- cf * (K1 + 2*K23 + R) / 6     [+ J]     (RK4, cf = dt)
// The trailing blocks of the grid advance the vector recursion (m forward, lam backward) for this rank's rows:
//     y_r = sum_k Aeff[r][k] x[k] ;  r_v = fwd ? -y + e : -e + y ;  same slot logic with vector buffers.
struct StageArgs {
  int D, row0, Mp, cw, fwd, kstore, final, mid_e, has_j;
  int nb = 1;           // problems in grid.z; per-problem strides of the pointer groups below
  size_t zW = 0, zE = 0, zJ = 0, zBase = 0, zK = 0, zOut = 0, zA = 0, zX = 0, zEv = 0, zJv = 0, zVb = 0, zKv = 0, zVo = 0;
  int sym_ok = 0;       // the caller guarantees symmetric E / J / base / slots: the symmetric-tile-pair kernel may serve (internal drivers)
  double cx, cf;
  const double* W;      // [q][Mp][cw] packed row block of the stage product
  const double* Wcol;   // [D][Mp]     column block of the stage product (== W when one rank owns everything)
  const double* E0; const double* E1;   // [Mp rows][D] (row block): Sigma, or G_t / G_{t-1} (mid: 0.5*(E0+E1))
  const double* J;      // [Mp][D] jump added at the end of a backward step (or nullptr)
  const double* base;   // [Mp][D] S_k / Psi_t row block
  double* K1; double* K23;   // [Mp][D]
  double* out;          // [Mp][D]: next stage state row block, or the new S / Psi row block
  // vector part
  const double* A0; const double* A1; int lda, mid_a;   // rows [row0, row0+Mp) of the stage's A (full D columns)
  const double* x;      // [D] stage vector (full)
  const double* e0; const double* e1; int mid_ev;        // [Mp] b or dEsde_dm entries of this rank's rows
  const double* jv;     // [Mp] vector jump or nullptr
  const double* vbase;  // [Mp]
  double* k1v; double* k23v; double* vout;   // [Mp]
  // one-kernel stage (k_stage_prod): the operands of the products instead of W / Wcol
  const double* M0 = nullptr; const double* M1 = nullptr;   // A of the matrix product (M1: mid-point partner or nullptr), [D][D]
  const double* Xm = nullptr;                               // stage state S / Psi (symmetric), [D][D]
  size_t zM = 0, zXm = 0;
};

constexpr int TS = 32;

__device__ __forceinline__ double stage_combine(double r, double base, double k1, double k23, int fin, double cx,
                                                double cf, double sgn, double jump) {
  if (fin == 0) return base + sgn * (cx * r);
  double comb = r;
  if (fin == 2) comb = k1 + r;
  if (fin == 3) comb = (k1 + 2.0 * k23 + r) / 6.0;
  return base + sgn * (cf * comb) + jump;
}

__device__ __forceinline__ void stage_batch_offsets(StageArgs& a) {
  if (a.nb <= 1) return;
  const size_t z = blockIdx.z;
  a.W += z * a.zW; a.Wcol += z * a.zW;
  a.E0 += z * a.zE; if (a.E1) a.E1 += z * a.zE;
  if (a.J) a.J += z * a.zJ;
  a.base += z * a.zBase; a.K1 += z * a.zK; a.K23 += z * a.zK; a.out += z * a.zOut;
  a.A0 += z * a.zA; if (a.A1) a.A1 += z * a.zA;
  a.x += z * a.zX;
  a.e0 += z * a.zEv; if (a.e1) a.e1 += z * a.zEv;
  if (a.jv) a.jv += z * a.zJv;
  a.vbase += z * a.zVb; a.k1v += z * a.zKv; a.k23v += z * a.zKv; a.vout += z * a.zVo;
  if (a.M0) { a.M0 += z * a.zM; if (a.M1) a.M1 += z * a.zM; a.Xm += z * a.zXm; }
}

// the vector recursion of a stage (m forward, lam backward): one wave per row of this rank's block, `blk` counts the
// workgroups behind the matrix tiles
__device__ __forceinline__ void stage_vector_rows(const StageArgs& a, int blk) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blk * (NT / 64) + wave;
  if (r >= a.Mp) return;
  const double sgn = a.fwd ? 1.0 : -1.0;
  const size_t ro = (size_t)(a.row0 + r) * a.lda;
  double s = 0.0;
  for (int k = lane; k < a.D; k += 64) {
    const double av = a.mid_a ? 0.5 * (a.A0[ro + k] + a.A1[ro + k]) : a.A0[ro + k];
    s = __builtin_fma(av, a.x[k], s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    const double e = a.mid_ev ? 0.5 * (a.e1[r] + a.e0[r]) : a.e0[r];
    const double rv = a.fwd ? (-s + e) : (-e + s);
    const double k1 = (a.final >= 2) ? a.k1v[r] : 0.0;
    const double k23 = (a.final == 3) ? a.k23v[r] : 0.0;
    if (a.kstore == 1) a.k1v[r] = rv;
    else if (a.kstore == 2) a.k23v[r] = rv;
    else if (a.kstore == 3) a.k23v[r] = a.k23v[r] + rv;
    const double jump = (a.final && a.jv) ? a.jv[r] : 0.0;
    a.vout[r] = stage_combine(rv, a.vbase[r], k1, k23, a.final, a.cx, a.cf, sgn, jump);
  }
}

// one-kernel stages (large_d_stage.hip): KIND 0 = k_stage_prod (a.M0 / a.M1 / a.Xm set), 1 = k_stage_wide (a.M0 / a.Xm set)
hipError_t launch_stage_fused(int kind, const StageArgs& a, hipStream_t st);

}  // namespace ld
}  // namespace vgpa
