// Large-D time stepping (D > 64): every RK stage is one fp64-MFMA GEMM + one fused element-wise kernel.
//
// Same mathematics and stage structure as the LDS-resident kernels (ode_mfma_impl.h): with S / Psi symmetric
//     forward : f_S   = -(W + W^T) + Sigma,   W  = A S            (src/numerics/ode_solver.py:60)
//     backward: f_Psi = -G + W' + W'^T,       W' = A^T Psi        (ode_solver.py:94)
// steppers of src/numerics/{euler,heun,runge_kutta2,runge_kutta4}.py incl. the RK2 quirk (runge_kutta2.py:96) and
// f_lam = -g + A.lam (ode_solver.py:77).  The stage state lives in HBM / L2 (D = 1024: 8 MB, D = 4096: 134 MB).
//
// Row sharding.  A rank owns the row block I_p = [row0, row0 + Mp) of S / Psi and of A (forward) --
// for the backward product A^T Psi it owns the COLUMN block of A.  Per stage it computes W[I_p, :] with the GEMM,
// written directly in "column-chunk packed" layout [q][Mp][CW] so that an all-to-all of the chunks delivers
// Wcol = W[:, I_p] (shape [D][Mp]) without any repacking, then the element-wise kernel forms
//     R[I_p, :] = -(W[I_p, :] + Wcol^T) + Sigma[I_p, :]
// and the next stage state's row block, which an all-gather completes.  With one rank CW = D and Wcol = W.
// The collectives are issued by the host side (vgpa_amd/large_d.py, torch.distributed = RCCL) between the two
// kernels of a stage; this file only exposes the two kernels through the C ABI (vgpa_ld_*).
//
// GEMM: 128 x 64 block tile, BK = 16, 256 threads = 2 x 2 waves of 64 x 32, v_mfma_f64_16x16x4_f64
// (A lane map i = l & 15, k = l >> 4; B k = l >> 4, j = l & 15; C col = l & 15, row = (l >> 4) + 4 r -- probed,
// profiles/r01_fp64_mfma_layout_probe.txt), LDS tiles stored k-major with leading dimensions chosen for
// conflict-free b64 fragment reads, register-prefetched global loads, two LDS buffers, one barrier per k-step.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <thread>

#include "vgpa_internal.h"

#include "large_d_stage.h"

namespace vgpa {
namespace ld {

// (diagnostic switch: VGPA_GEMM_SCALAR_LOADS=1 keeps the 8-byte-load kernel for full tiles)
static const bool gemm_scalar_loads = [] { const char* e = getenv("VGPA_GEMM_SCALAR_LOADS"); return e && e[0] == '1'; }();

struct GemmArgs {
  int M, N, K;              // C[M x N] = op(A)[M x K] . B[K x N]
  const double* A0;         // TRANSA ? [K x M] : [M x K], leading dimension lda
  const double* A1;         // second operand for the mid-point 0.5*(A0 + A1), or nullptr
  int lda;
  const double* B; int ldb;
  double* C;                // packed by column chunks of width cw: C[(j / cw) * M * cw + i * cw + (j % cw)]
  int cw;
  // K-chunk launches of the pipelined row-sharded stage (shard_stage): the K tiles of THIS launch are `seg_tiles` consecutive
  // k-tiles out of every `seg_stride` k (A0 / B point at the first one; K = k in this launch), i.e. the same sub-chunk of
  // every rank's row block of the stage state; accumulate: the accumulators start from C instead of zero (the K-chunk
  // launches of one product continue ONE fp64 FMA chain per element: bit-identical to a single launch over the same k order).
  int seg_tiles = 0, seg_stride = 0, accumulate = 0;
  // batch of independent problems in grid.z (64 < D <= 512: one problem leaves most of the chip idle): per-problem strides
  int nb = 1;
  size_t zA = 0, zB = 0, zC = 0;
};

// FULL: M, N, K are multiples of the tile sizes -- no bounds checks (each one is an EXEC-masked branch in the k loop)
template <bool TRANSA, bool MID, int BM, bool FULL>
__global__ void __launch_bounds__(NT) k_gemm(GemmArgs g) {
  if (g.nb > 1) { const size_t z = blockIdx.z; g.A0 += z * g.zA; if (MID) g.A1 += z * g.zA; g.B += z * g.zB; g.C += z * g.zC; }
  constexpr int LDAS = BM + 17;   // odd: conflict-free transposing stores (NN), near conflict-free fragment reads
  constexpr int MT = BM / 32;     // 16-row MFMA tiles per wave along M (wave tile = (BM/2) x 32)
  constexpr int AQ = BM * BK / NT;
  __shared__ __attribute__((aligned(16))) double As[2][BK * LDAS];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * LDBS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
  const int fi = lane & 15, fk = lane >> 4;

  // global -> register staging maps
  //  A tile: 128 x 16 = 2048 doubles, 8 per thread.  NN: k = tid & 15 fastest (rows of A are contiguous in k);
  //          TN: i = tid & 127 fastest (rows of A^T storage are contiguous in i).
  //  B tile: 16 x 64 = 1024 doubles, 4 per thread, j = tid & 63 fastest.
  // two register sets: the operands of k-tile t+2 are in flight while tile t is multiplied and tile t+1 moves to LDS
  double ra0[AQ], rb0[4], ra1[AQ], rb1[4];
  const int klim = g.seg_tiles ? 0x7fffffff : g.K;          // segmented launches are made of whole k-tiles
  auto k_of = [&](int kt) { return g.seg_tiles ? (kt / g.seg_tiles) * g.seg_stride + (kt % g.seg_tiles) * BK : kt * BK; };
  auto load_tiles = [&](int k0, double (&ra)[AQ], double (&rb)[4]) {
#pragma unroll
    for (int q = 0; q < AQ; q++) {
      int i, k;
      if (TRANSA) { i = tid & (BM - 1); k = (tid / BM) + (NT / BM) * q; }
      else { k = tid & 15; i = (tid >> 4) + 16 * q; }
      const int gi = i0 + i, gk = k0 + k;
      double v = 0.0;
      if (FULL || (gi < g.M && gk < klim)) {
        const size_t idx = TRANSA ? ((size_t)gk * g.lda + gi) : ((size_t)gi * g.lda + gk);
        v = MID ? 0.5 * (g.A0[idx] + g.A1[idx]) : g.A0[idx];
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = tid & 63, k = (tid >> 6) + 4 * q;
      const int gj = j0 + j, gk = k0 + k;
      rb[q] = (FULL || (gj < g.N && gk < klim)) ? g.B[(size_t)gk * g.ldb + gj] : 0.0;
    }
  };
  auto store_tiles = [&](int buf, const double (&ra)[AQ], const double (&rb)[4]) {
#pragma unroll
    for (int q = 0; q < AQ; q++) {
      int i, k;
      if (TRANSA) { i = tid & (BM - 1); k = (tid / BM) + (NT / BM) * q; }
      else { k = tid & 15; i = (tid >> 4) + 16 * q; }
      As[buf][k * LDAS + i] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = tid & 63, k = (tid >> 6) + 4 * q;
      Bs[buf][k * LDBS + j] = rb[q];
    }
  };

  d4 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};

  const int nk = (g.K + BK - 1) / BK;
  if (g.accumulate) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const int gj = j0 + 32 * wn + 16 * nt + (lane & 15);
        if (gj >= g.N) continue;
        const size_t chunk = (size_t)(gj / g.cw) * g.M * g.cw + (gj % g.cw);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gi = i0 + (BM / 2) * wm + 16 * mt + (lane >> 4) + 4 * r;
          if (gi < g.M) acc[mt][nt][r] = g.C[chunk + (size_t)gi * g.cw];
        }
      }
  }
  auto compute = [&](int cur) {
    frag_ptr as = frag(As[cur] + fk * LDAS + (BM / 2) * wm + fi);
    frag_ptr bs = frag(Bs[cur] + fk * LDBS + 32 * wn + fi);
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
  };
  if constexpr (BM < 128) {
    // prefetch distance two (measured +11 % at D = 1024 where the small tiles leave few waves per CU to hide HBM
    // latency).  Prologue: tile 0 -> LDS buffer 0, tile 1 -> register set 1.
    load_tiles(0, ra0, rb0);
    store_tiles(0, ra0, rb0);
    if (nk > 1) load_tiles(k_of(1), ra1, rb1);
    __syncthreads();
    // iteration kt: loads of tile kt+2 are issued, tile kt is multiplied, tile kt+1 (in registers since the previous
    // iteration) goes to the other LDS buffer.  Unrolled by two so that the register sets are addressed statically.
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 2 < nk) load_tiles(k_of(kt + 2), ra0, rb0);
      compute(0);
      if (kt + 1 < nk) store_tiles(1, ra1, rb1);
      __syncthreads();
      if (kt + 1 >= nk) break;
      if (kt + 3 < nk) load_tiles(k_of(kt + 3), ra1, rb1);
      compute(1);
      if (kt + 2 < nk) store_tiles(0, ra0, rb0);
      __syncthreads();
    }
  } else {
    // prefetch distance one (the 128-row tile measured 3 % faster this way at D = 4096)
    load_tiles(0, ra0, rb0);
    store_tiles(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load_tiles(k_of(kt + 1), ra0, rb0);
      compute(cur);
      if (kt + 1 < nk) store_tiles(cur ^ 1, ra0, rb0);
      __syncthreads();
    }
  }

  // epilogue: C col = lane & 15, row = (lane >> 4) + 4 r
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const int gj = j0 + 32 * wn + 16 * nt + (lane & 15);
      if (gj >= g.N) continue;
      const size_t chunk = (size_t)(gj / g.cw) * g.M * g.cw + (gj % g.cw);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + (BM / 2) * wm + 16 * mt + (lane >> 4) + 4 * r;
        if (gi < g.M) g.C[chunk + (size_t)gi * g.cw] = acc[mt][nt][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Full tiles with 16-byte global loads (M, N, K multiples of the tile sizes, even leading dimensions, 16-byte aligned
// operands): half the global-load and LDS-store instructions of k_gemm.  The A tile keeps the orientation it has in
// HBM, so that its 16-byte pairs stay contiguous in LDS:
//   NN (A row-major [i][k]):  As[i][LDK], LDK = 18 -- fragment read (i = l & 15, k = l >> 4) is conflict-free because
//                             18 i (mod 32) runs over the even residues and the two k of a 32-lane group fill the odd ones
//   TN (A stored  [k][i]):    As[k][BM + 16] as in k_gemm (k-major), b128 stores of 16 consecutive pairs
//   B  (row-major [k][j]):    Bs[k][LDBS], b128 stores
template <bool TRANSA, bool MID, int BM, bool SEG = false, int PFU = 0>
__global__ void __launch_bounds__(NT) k_gemm_v(GemmArgs g) {
  if (g.nb > 1) { const size_t z = blockIdx.z; g.A0 += z * g.zA; if (MID) g.A1 += z * g.zA; g.B += z * g.zB; g.C += z * g.zC; }
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int LDK = 18;                       // NN: doubles per tile row (16 + 2)
  constexpr int LDT = BM + 16;                  // TN: = 16 (mod 32)
  constexpr int ASZ = TRANSA ? BK * LDT : BM * LDK;
  constexpr int MT = BM / 32;
  constexpr int AV = BM * BK / 2 / NT;          // 16-byte pairs of the A tile per thread (BM = 32: 1)
  __shared__ __attribute__((aligned(16))) double As[2][ASZ];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * LDBS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
  const int fi = lane & 15, fk = lane >> 4;

  // per-thread operand pointers (advanced by one k-tile per load_tiles call; dedicated registers -- with addresses
  // recomputed per tile the register allocator recycles the data registers of the previous tile's loads for them, and
  // every new load then waits for those loads: vmcnt(0) in front of the prefetch) and LDS offsets of the pairs
  const double* pA[AV];
  const double* pA1[AV];
  const double* pB[2];
  int sa[AV], sb[2];
#pragma unroll
  for (int q = 0; q < AV; q++) {
    size_t go;
    if (TRANSA) {
      const int i2 = tid & (BM / 2 - 1), k = tid / (BM / 2) + (NT / (BM / 2)) * q;
      go = (size_t)k * g.lda + i0 + 2 * i2;
      sa[q] = k * LDT + 2 * i2;
    } else {
      const int kp = tid & 7, i = (tid >> 3) + 32 * q;
      go = (size_t)(i0 + i) * g.lda + 2 * kp;
      sa[q] = i * LDK + 2 * kp;
    }
    pA[q] = g.A0 + go;
    pA1[q] = MID ? g.A1 + go : nullptr;
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int j2 = tid & 31, k = (tid >> 5) + 8 * q;
    pB[q] = g.B + (size_t)k * g.ldb + j0 + 2 * j2;
    sb[q] = k * LDBS + 2 * j2;
  }
  const size_t astep = TRANSA ? (size_t)BK * g.lda : (size_t)BK, bstep = (size_t)BK * g.ldb;
  // SEG (K-chunk launch of the pipelined sharded stage): after every seg_tiles k-tiles jump to the same sub-chunk of the next
  // rank's row block
  const int seg_skip_k = SEG ? g.seg_stride - g.seg_tiles * BK : 0;
  const size_t askip = TRANSA ? (size_t)seg_skip_k * g.lda : (size_t)seg_skip_k, bskip = (size_t)seg_skip_k * g.ldb;
  int seg_cnt = 0;

  d2 ra0[AV], rb0[2], ra1[AV], rb1[2];
  int advances = g.K / BK - 1;                            // (PFU: loads are unconditional; behind the last tile the pointers stay)
  auto load_tiles = [&](d2 (&ra)[AV], d2 (&rb)[2]) {      // loads the NEXT k-tile (tiles are requested in order)
    const bool adv = !(PFU > 0 && advances <= 0);
    const size_t as_ = adv ? astep : 0, bs_ = adv ? bstep : 0;
    if (PFU > 0) advances--;
#pragma unroll
    for (int q = 0; q < AV; q++) {
      const d2 v0 = *reinterpret_cast<const d2*>(pA[q]);
      pA[q] += as_;
      if (MID) {
        const d2 v1 = *reinterpret_cast<const d2*>(pA1[q]);
        pA1[q] += as_;
        ra[q] = d2{0.5 * (v0[0] + v1[0]), 0.5 * (v0[1] + v1[1])};
      } else {
        ra[q] = v0;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) { rb[q] = *reinterpret_cast<const d2*>(pB[q]); pB[q] += bs_; }
    if (SEG) {
      if (++seg_cnt == g.seg_tiles) {
        seg_cnt = 0;
        const size_t ak = adv ? askip : 0, bk = adv ? bskip : 0;
#pragma unroll
        for (int q = 0; q < AV; q++) { pA[q] += ak; if (MID) pA1[q] += ak; }
#pragma unroll
        for (int q = 0; q < 2; q++) pB[q] += bk;
      }
    }
  };
  auto store_tiles = [&](int buf, const d2 (&ra)[AV], const d2 (&rb)[2]) {
#pragma unroll
    for (int q = 0; q < AV; q++) *reinterpret_cast<d2*>(&As[buf][sa[q]]) = ra[q];
#pragma unroll
    for (int q = 0; q < 2; q++) *reinterpret_cast<d2*>(&Bs[buf][sb[q]]) = rb[q];
  };

  d4 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};

  if (SEG) {
    if (g.accumulate) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
          const int gj = j0 + 32 * wn + 16 * nt + (lane & 15);
          const size_t chunk = (size_t)(gj / g.cw) * g.M * g.cw + (gj % g.cw);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int gi = i0 + (BM / 2) * wm + 16 * mt + (lane >> 4) + 4 * r;
            acc[mt][nt][r] = g.C[chunk + (size_t)gi * g.cw];
          }
        }
    }
  }
  const int nk = g.K / BK;
  auto compute = [&](int cur) {
    frag_ptr as = frag(TRANSA ? As[cur] + fk * LDT + (BM / 2) * wm + fi : As[cur] + ((BM / 2) * wm + fi) * LDK + fk);
    frag_ptr bs = frag(Bs[cur] + fk * LDBS + 32 * wn + fi);
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      double af[MT], bf[2];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) af[mt] = TRANSA ? as[kk * 4 * LDT + 16 * mt] : as[16 * mt * LDK + kk * 4];
#pragma unroll
      for (int nt = 0; nt < 2; nt++) bf[nt] = bs[kk * 4 * LDBS + 16 * nt];
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    }
  };
  if constexpr (PFU > 0) {
    // PFU register sets, no load or LDS store under a branch (K / BK a multiple of PFU): a branch makes the wait counts of its two
    // paths merge to the conservative one -- the LDS stores of one set then wait for the loads just issued into another
    static_assert(PFU % 2 == 0 && !MID, "");
    d2 sa_[PFU][AV], sb_[PFU][2];
#pragma unroll
    for (int u = 0; u < PFU; u++) load_tiles(sa_[u], sb_[u]);
    store_tiles(0, sa_[0], sb_[0]);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += PFU) {
#pragma unroll
      for (int u = 0; u < PFU; u++) {
        load_tiles(sa_[u], sb_[u]);
        compute(u & 1);
        store_tiles((u + 1) & 1, sa_[(u + 1) % PFU], sb_[(u + 1) % PFU]);
        __syncthreads();
      }
    }
  } else if constexpr (BM < 128) {
    load_tiles(ra0, rb0);
    store_tiles(0, ra0, rb0);
    if (nk > 1) load_tiles(ra1, rb1);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 2 < nk) load_tiles(ra0, rb0);
      compute(0);
      if (kt + 1 < nk) store_tiles(1, ra1, rb1);
      __syncthreads();
      if (kt + 1 >= nk) break;
      if (kt + 3 < nk) load_tiles(ra1, rb1);
      compute(1);
      if (kt + 2 < nk) store_tiles(0, ra0, rb0);
      __syncthreads();
    }
  } else {
    load_tiles(ra0, rb0);
    store_tiles(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load_tiles(ra0, rb0);
      compute(cur);
      if (kt + 1 < nk) store_tiles(cur ^ 1, ra0, rb0);
      __syncthreads();
    }
  }

#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const int gj = j0 + 32 * wn + 16 * nt + (lane & 15);
      const size_t chunk = (size_t)(gj / g.cw) * g.M * g.cw + (gj % g.cw);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + (BM / 2) * wm + 16 * mt + (lane >> 4) + 4 * r;
        g.C[chunk + (size_t)gi * g.cw] = acc[mt][nt][r];
      }
    }
}

__global__ void __launch_bounds__(NT) k_stage(StageArgs a) {
  stage_batch_offsets(a);
  __shared__ double tile[TS][TS + 1];
  const int ntx = (a.D + TS - 1) / TS, nty = (a.Mp + TS - 1) / TS;
  const int nmat = ntx * nty;
  const double sgn = a.fwd ? 1.0 : -1.0;
  if ((int)blockIdx.x < nmat) {
    const int by = blockIdx.x / ntx, bx = blockIdx.x - by * ntx;   // tile of R: rows by (local), cols bx (global)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // 32 x 8
    // transposed operand: Wcol[j][r] for j in the column tile, r in the row tile -> read rows j, cols r (coalesced)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int jj = ty + 8 * q;
      const int j = bx * TS + jj, r = by * TS + tx;
      tile[jj][tx] = (j < a.D && r < a.Mp) ? a.Wcol[(size_t)j * a.Mp + r] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int rr = ty + 8 * q;
      const int r = by * TS + rr, j = bx * TS + tx;
      if (r < a.Mp && j < a.D) {
        const size_t o = (size_t)r * a.D + j;
        const double w = a.W[(size_t)(j / a.cw) * a.Mp * a.cw + (size_t)r * a.cw + (j % a.cw)];
        const double wt = tile[tx][rr];
        const double e = a.mid_e ? 0.5 * (a.E1[o] + a.E0[o]) : a.E0[o];
        const double rv = a.fwd ? ((-w - wt) + e) : ((-e + wt) + w);
        const double k1 = (a.final >= 2) ? a.K1[o] : 0.0;
        const double k23 = (a.final == 3) ? a.K23[o] : 0.0;
        if (a.kstore == 1) a.K1[o] = rv;
        else if (a.kstore == 2) a.K23[o] = rv;
        else if (a.kstore == 3) a.K23[o] = a.K23[o] + rv;
        const double jump = (a.final && a.has_j) ? a.J[o] : 0.0;
        a.out[o] = stage_combine(rv, a.base[o], k1, k23, a.final, a.cx, a.cf, sgn, jump);
      }
    }
    return;
  }
  stage_vector_rows(a, (int)blockIdx.x - nmat);
}

template <int BM, bool FULL>
static void launch_gemm_bm_f(bool transa, const GemmArgs& g, hipStream_t st) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nb);
  const bool mid = g.A1 != nullptr;
  if (transa) {
    if (mid) hipLaunchKernelGGL((k_gemm<true, true, BM, FULL>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((k_gemm<true, false, BM, FULL>), grid, dim3(NT), 0, st, g);
  } else {
    if (mid) hipLaunchKernelGGL((k_gemm<false, true, BM, FULL>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((k_gemm<false, false, BM, FULL>), grid, dim3(NT), 0, st, g);
  }
}

template <int BM>
static void launch_gemm_bm_v(bool transa, const GemmArgs& g, hipStream_t st) {
  dim3 grid(g.N / BN, g.M / BM, g.nb);
  const bool mid = g.A1 != nullptr;
  // four register sets of loads in flight and no load under a branch (k_gemm_v, PFU) when the k-tile count allows it: D = 1024
  // 44.6 -> 39.6 us per product (47 -> 54 TFLOP/s; 2 sets 44.1, 8 sets 40.1), D = 2048 forward recursion 44.0 -> 47.7 TFLOP/s;
  // VGPA_GEMM_PF=0: the two-set loop of rounds 2-4
  static const bool pfu = [] { const char* e = getenv("VGPA_GEMM_PF"); return !(e && e[0] == '0'); }();
  if (!mid && pfu && (g.K / BK) % 4 == 0) {
    if (transa) hipLaunchKernelGGL((k_gemm_v<true, false, BM, false, 4>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((k_gemm_v<false, false, BM, false, 4>), grid, dim3(NT), 0, st, g);
    return;
  }
  if (transa) {
    if (mid) hipLaunchKernelGGL((k_gemm_v<true, true, BM>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((k_gemm_v<true, false, BM>), grid, dim3(NT), 0, st, g);
  } else {
    if (mid) hipLaunchKernelGGL((k_gemm_v<false, true, BM>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((k_gemm_v<false, false, BM>), grid, dim3(NT), 0, st, g);
  }
}

template <int BM>
static void launch_gemm_bm(bool transa, const GemmArgs& g, hipStream_t st) {
  const bool full = g.M % BM == 0 && g.N % BN == 0 && g.K % BK == 0;
  auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const bool vec = full && g.lda % 2 == 0 && g.ldb % 2 == 0 && al16(g.A0) && al16(g.A1) && al16(g.B) && !gemm_scalar_loads;
  const bool seg = g.seg_tiles > 0 || g.accumulate;
  if (vec && seg && !g.A1) {        // K-chunk launch (no mid-point operand: the sharded driver forms it once per step)
    GemmArgs h = g;
    if (!h.seg_tiles) { h.seg_tiles = h.K / BK; h.seg_stride = h.K; }
    dim3 grid(g.N / BN, g.M / BM, g.nb);
    static const bool pfu = [] { const char* e = getenv("VGPA_GEMM_PF"); return !(e && e[0] == '0'); }();
    if (pfu && (h.K / BK) % 4 == 0) {        // four register sets of loads in flight (launch_gemm_bm_v)
      if (transa) hipLaunchKernelGGL((k_gemm_v<true, false, BM, true, 4>), grid, dim3(NT), 0, st, h);
      else hipLaunchKernelGGL((k_gemm_v<false, false, BM, true, 4>), grid, dim3(NT), 0, st, h);
      return;
    }
    if (transa) hipLaunchKernelGGL((k_gemm_v<true, false, BM, true>), grid, dim3(NT), 0, st, h);
    else hipLaunchKernelGGL((k_gemm_v<false, false, BM, true>), grid, dim3(NT), 0, st, h);
    return;
  }
  if (vec && !seg) launch_gemm_bm_v<BM>(transa, g, st);
  else if (full) launch_gemm_bm_f<BM, true>(transa, g, st);
  else launch_gemm_bm_f<BM, false>(transa, g, st);
}

hipError_t launch_gemm(bool transa, const GemmArgs& g, hipStream_t st) {
  // Rows of the block tile: the height whose launch fills the chip most evenly, weighted by what the height itself is worth.  Measured
  // with the four-set loop (forward recursion, TFLOP/s, 32 / 64 / 128 rows -- D = 1536: 41.0 / 35.1 / 27.6; 1792: 42.3 / 38.5 / 37.3;
  // 2048: 47.5 / 49.3 / 47.6; 2560: 47.5 / 46.4 / 40.0; 3072: 49.4 / 52.6 / 46.2; one-rank D = 4096 sweep 0.309 / 0.293 / 0.297 s; one
  // rank's 512 x 4096 x 4096 product of configs[4]: 299 / 294 us): every one of these orderings follows from
  //     score = W / (ceil(W / 256) 256)  x  {0.95, 1.0, 0.966}[height]  x  min(1, (W / 512)^0.22),      W = workgroups of the launch,
  // i.e. the balance of the last round over 256 CUs, 5 % for the smaller reuse of 32-row tiles, 3.4 % for the two (instead of four)
  // resident workgroups of 128-row tiles, and what a CU loses below two workgroups (halving them costs 16 %: D = 1024 at 64 rows
  // 44.0 against 39.6 us, D = 768 / 896 at 64 rows 11.8 / 11.6 against 10.7 / 10.5 ms).  The rounds-2-to-4 rule (the tallest tile
  // with at least 512 workgroups) lost 10-19 % at D = 1536 ... 3072.
  static const int force_bm = [] { const char* e = getenv("VGPA_GEMM_BM"); return e ? atoi(e) : 0; }();     // (diagnostic: 32 / 64 / 128)
  int best = force_bm;
  if (!best) {
    double best_score = -1.0;
    const int heights[3] = {32, 64, 128};
    const double worth[3] = {0.95, 1.0, 0.966};
    for (int h = 0; h < 3; h++) {
      const long long w = (long long)((g.N + BN - 1) / BN) * ((g.M + heights[h] - 1) / heights[h]) * g.nb;
      const double score = (double)w / (double)((w + 255) / 256 * 256) * worth[h] * (w < 512 ? std::pow((double)w / 512.0, 0.22) : 1.0);
      if (score >= best_score) { best_score = score; best = heights[h]; }      // (ties: the taller tile)
    }
  }
  if (best == 128) launch_gemm_bm<128>(transa, g, st);
  else if (best == 64) launch_gemm_bm<64>(transa, g, st);
  else launch_gemm_bm<32>(transa, g, st);
  return hipGetLastError();
}

// One rank owns every row (Mp = D, W = Wcol): R = -(W + W^T) + E is symmetric, so a workgroup takes the PAIR of tiles (by, bx)
// and (bx, by), by <= bx: both W tiles are read once (k_stage reads every W tile twice: once straight, once transposed), the
// forcing term / base state / Runge-Kutta slope slots only on and above the diagonal (they are symmetric and only this kernel
// reads the slots), the stage state is written at (r, j) and -- through an LDS transposition, coalesced -- at (j, r).  ~36 MB
// instead of ~64 MB per stage at D = 1024.  The vector recursion rides in the trailing blocks as in k_stage.
__global__ void __launch_bounds__(NT) k_stage_sym(StageArgs a) {
  stage_batch_offsets(a);
  __shared__ double ta[TS][TS + 1], tb[TS][TS + 1], tc[TS][TS + 1];
  const int nt = (a.D + TS - 1) / TS;
  const int npair = nt * (nt + 1) / 2;
  const double sgn = a.fwd ? 1.0 : -1.0;
  const int D = a.D;
  if ((int)blockIdx.x < npair) {
    int by = 0, rem = (int)blockIdx.x;                 // pair index -> (by, bx), by <= bx (row by has nt - by pairs)
    while (rem >= nt - by) { rem -= nt - by; by++; }
    const int bx = by + rem;
    const bool diag = (by == bx);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;         // 32 x 8
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int rr = ty + 8 * q;
      const int r = by * TS + rr, j = bx * TS + tx;
      ta[rr][tx] = (r < D && j < D) ? a.W[(size_t)r * D + j] : 0.0;             // W[by tile][bx tile]
      if (!diag) {
        const int r2 = bx * TS + rr, j2 = by * TS + tx;
        tb[rr][tx] = (r2 < D && j2 < D) ? a.W[(size_t)r2 * D + j2] : 0.0;        // W[bx tile][by tile]
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int rr = ty + 8 * q;
      const int r = by * TS + rr, j = bx * TS + tx;
      double res = 0.0;
      if (r < D && j < D) {
        const size_t o = (size_t)r * D + j;
        const double w = ta[rr][tx];
        const double wt = diag ? ta[tx][rr] : tb[tx][rr];
        const double e = a.mid_e ? 0.5 * (a.E1[o] + a.E0[o]) : a.E0[o];
        const double rv = a.fwd ? ((-w - wt) + e) : ((-e + wt) + w);
        const double k1 = (a.final >= 2) ? a.K1[o] : 0.0;
        const double k23 = (a.final == 3) ? a.K23[o] : 0.0;
        if (a.kstore == 1) a.K1[o] = rv;
        else if (a.kstore == 2) a.K23[o] = rv;
        else if (a.kstore == 3) a.K23[o] = a.K23[o] + rv;
        const double jump = (a.final && a.has_j) ? a.J[o] : 0.0;
        res = stage_combine(rv, a.base[o], k1, k23, a.final, a.cx, a.cf, sgn, jump);
        a.out[o] = res;
      }
      if (!diag) tc[rr][tx] = res;
    }
    if (diag) return;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {                        // the mirror tile: out[bx tile][by tile] = (this tile)^T
      const int rr = ty + 8 * q;
      const int r2 = bx * TS + rr, j2 = by * TS + tx;
      if (r2 < D && j2 < D) a.out[(size_t)r2 * D + j2] = tc[tx][rr];
    }
    return;
  }
  stage_vector_rows(a, (int)blockIdx.x - npair);
}

// selection constants of run_stage (the kernels: large_d_stage.hip)
constexpr int kStageProdMaxD = 512;
constexpr int kStageProdMaxPairs = 64;
constexpr int kStageWideMaxD = 2048;
constexpr int kStageWideFullTileD = 384;
// (read per call, not once per process: the tests run both versions at small D in one process)
static int stage_prod_max_d() {
  const char* e = getenv("VGPA_STAGE_FUSED");       // 0: never the latency version of the one-kernel stage; <n>: up to D = n
  return e ? atoi(e) : kStageProdMaxD;
}

static int stage_wide_max_d() {
  const char* e = getenv("VGPA_STAGE_WIDE");        // 0: never the throughput version (two kernels per stage instead); <n>: up to D = n
  return e ? atoi(e) : kStageWideMaxD;
}

static const bool stage_sym_off = [] { const char* e = getenv("VGPA_STAGE_FULL"); return e && e[0] == '1'; }();

hipError_t launch_stage(const StageArgs& a, hipStream_t st) {
  const int nvec = (a.Mp + (NT / 64) - 1) / (NT / 64);
  if (a.sym_ok && a.Mp == a.D && a.row0 == 0 && a.cw == a.D && a.W == a.Wcol && !stage_sym_off) {
    const int nt = (a.D + TS - 1) / TS;
    hipLaunchKernelGGL(k_stage_sym, dim3(nt * (nt + 1) / 2 + nvec, 1, a.nb), dim3(NT), 0, st, a);
    return hipGetLastError();
  }
  const int ntx = (a.D + TS - 1) / TS, nty = (a.Mp + TS - 1) / TS;
  hipLaunchKernelGGL(k_stage, dim3(ntx * nty + nvec, 1, a.nb), dim3(NT), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Single-rank drivers (one GPU owns every row): the per-step / per-stage loop of the four steppers, all launches on
// one stream.  `ws` is a workspace of at least ld_workspace_doubles(D) doubles.
// mid-point operand 0.5 (A_k + A_{k+1}) (runge_kutta2.py:74, runge_kutta4.py:74): formed ONCE per step and shared by
// the two stages that use it; the GEMM that averages while staging its A tile streams both operands and is ~2.3x
// slower at D = 1024 (measured: 156 us vs 67 us) -- that variant stays for the row-sharded driver.
__global__ void __launch_bounds__(256) k_mid(const double* __restrict__ a0, const double* __restrict__ a1, double* __restrict__ out,
                                             size_t n, size_t zin = 0, size_t zout = 0) {
  a0 += blockIdx.y * zin; a1 += blockIdx.y * zin; out += blockIdx.y * zout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = 0.5 * (a0[i] + a1[i]);
}

// Batched single-rank drivers (64 < D <= 512 with several problems per context: the reference treats every D alike,
// src/numerics/ode_solver.py:31-95, and ONE problem of that size leaves most of the chip idle).  The step / stage loops below
// compute problem 0's pointers as always; which array a pointer belongs to -- hence its per-problem stride -- is looked up by
// address in the ranges the API registered (x, the histories, the workspace ...); pointers outside every range (Sigma, the
// constant jump, S0) are shared by the problems.
thread_local BatchMap g_batch;
void ld_set_batch(const BatchMap* m) { g_batch = m ? *m : BatchMap{}; }
static inline size_t zs(const double* p) { return g_batch.nb > 1 ? g_batch.stride(p) : 0; }

thread_local bool use_library_gemm = false;
// Non-symmetric operator-level inputs (S0, dEsde_dS, jumps that are not symmetric): the shortcut S A^T = (A S)^T does not hold,
// so the second product of ode_solver.py:60,94 is formed literally -- Z = A_s X^T (forward) / A_s^T Psi^T (backward) from a
// transposed copy of the stage state, handed to the stage kernel as its column operand: Z[j][r] = (X A^T)[r][j] / (Psi A)[r][j].
thread_local bool literal_products = false;
void ld_set_literal_products(bool on) { literal_products = on; }

size_t ld_workspace_doubles(int D) { return (size_t)8 * D * D + 4 * (size_t)D; }

__global__ void __launch_bounds__(256) k_transpose(const double* __restrict__ X, double* __restrict__ Xt, int D, size_t zin, size_t zout) {
  __shared__ double tile[32][33];
  X += blockIdx.z * zin; Xt += blockIdx.z * zout;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r = r0 + ty + 8 * q, c = c0 + tx;
    tile[ty + 8 * q][tx] = (r < D && c < D) ? X[(size_t)r * D + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r = c0 + ty + 8 * q, c = r0 + tx;
    if (r < D && c < D) Xt[(size_t)r * D + c] = tile[tx][ty + 8 * q];
  }
}

namespace {
struct Work {
  double *W, *K1, *K23, *XA, *XB, *AM, *XT, *W2, *xvA, *xvB, *k1v, *k23v;
};
// which pair of operands the mid-point buffer AM currently averages (valid inside one step)
struct MidCache { const double* a0 = nullptr; const double* a1 = nullptr; };
Work carve_work(double* ws, int D) {
  const size_t DD = (size_t)D * D;
  Work w;
  w.W = ws; w.K1 = w.W + DD; w.K23 = w.K1 + DD; w.XA = w.K23 + DD; w.XB = w.XA + DD;
  w.AM = w.XB + DD; w.XT = w.AM + DD; w.W2 = w.XT + DD;
  w.xvA = w.W2 + DD; w.xvB = w.xvA + D; w.k1v = w.xvB + D; w.k23v = w.k1v + D;
  return w;
}

struct StageSpec {
  bool fwd;
  const double *Am0, *Am1;   // A operand of the matrix product (Am1: mid-point partner or nullptr)
  const double *Av0, *Av1;   // A of the vector recursion
  const double* X; const double* xv;
  const double *E0, *E1; const double *e0, *e1;
  const double* J; const double* jv;
  const double* base; const double* vbase;
  double* out; double* vout;
  int kstore, final_mode; double cx, cf;
};

StageArgs stage_args(int D, const Work& w, const StageSpec& s) {
  StageArgs a{};
  a.D = D; a.row0 = 0; a.Mp = D; a.cw = D; a.fwd = s.fwd ? 1 : 0; a.kstore = s.kstore; a.final = s.final_mode;
  a.sym_ok = literal_products ? 0 : 1;
  a.mid_e = s.E1 != nullptr; a.has_j = s.J != nullptr; a.cx = s.cx; a.cf = s.cf;
  a.W = w.W; a.Wcol = literal_products ? w.W2 : w.W; a.E0 = s.E0; a.E1 = s.E1; a.J = s.J; a.base = s.base; a.K1 = w.K1; a.K23 = w.K23; a.out = s.out;
  a.A0 = s.Av0; a.A1 = s.Av1; a.lda = D; a.mid_a = s.Av1 != nullptr; a.x = s.xv;
  a.e0 = s.e0; a.e1 = s.e1; a.mid_ev = s.e1 != nullptr; a.jv = s.jv; a.vbase = s.vbase;
  a.k1v = w.k1v; a.k23v = w.k23v; a.vout = s.vout;
  if (g_batch.nb > 1) {
    a.nb = g_batch.nb;
    a.zW = zs(a.W); a.zE = zs(a.E0); a.zJ = zs(a.J); a.zBase = zs(a.base); a.zK = zs(a.K1); a.zOut = zs(a.out);
    a.zA = zs(a.A0); a.zX = zs(a.x); a.zEv = zs(a.e0); a.zJv = zs(a.jv); a.zVb = zs(a.vbase); a.zKv = zs(a.k1v); a.zVo = zs(a.vout);
  }
  return a;
}

hipError_t run_stage(int D, const Work& w, const StageSpec& s, hipStream_t st, MidCache* mc = nullptr) {
  const double *ga0 = s.Am0, *ga1 = s.Am1;
  // Which implementation of the stage (measured on one box, fused sweep, ms: two kernels / k_stage_prod / k_stage_wide,
  // profiles/r05d_stage_versions_ab.txt -- D = 72: 40.4 / 25.4 / 38.4; 128: 39.9 / 29.7 / 36.9; 200: 63.5 / 43.3 / 52.9; 256: 54.1 / 47.9 /
  // 50.7; 384: 35.1 / - / 34.4; 512: 44.0 / - / 44.0; 8 x D = 128: 57.0 / - / 51.3; 640: 26.3 / - / 27.8; 768: 39.5 / - / 38.3; 1000:
  // 61.1 / 68.7 / 51.4; 1024: 46.7 / - / 49.0; 1536: 68.4 / - / 69.0; 2048: 71.3 / - / 78.2; more sizes in EXPERIMENTS.md s.13): the
  // latency version while the launch has at most kStageProdMaxPairs tile pairs (a quarter of the CUs); beyond that the two-kernel
  // stage where the product runs on its full-tile path (D a multiple of 64: 16-byte loads, four register sets) and D >= 384 -- the
  // throughput version leads there by 2-3 % at D = 384 and 768 only and trails by 4-8 % at ten other sizes -- and the throughput
  // version for everything else that is eligible: ragged D, whose product is the bounds-checked 8-byte kernel, and batched contexts
  // of mid-size problems.
  const bool fused_ok = !use_library_gemm && !literal_products;
  const int nt_ = (D + TS - 1) / TS;
  const bool wide = fused_ok && D % 2 == 0 && D <= stage_wide_max_d() && (D % 64 != 0 || D < kStageWideFullTileD || getenv("VGPA_STAGE_WIDE")) && (size_t)D * D * 8 < 0x7ff00000u &&
                    (reinterpret_cast<uintptr_t>(s.X) & 15u) == 0 && (reinterpret_cast<uintptr_t>(s.Am0) & 15u) == 0 &&
                    (!s.Am1 || (reinterpret_cast<uintptr_t>(s.Am1) & 15u) == 0) && zs(s.X) % 2 == 0 && zs(s.Am0) % 2 == 0;
  // (more pairs than that and no throughput version -- odd D: the latency version still beats the bounds-checked product)
  if (fused_ok && D <= stage_prod_max_d() && ((long long)nt_ * (nt_ + 1) / 2 * g_batch.nb <= kStageProdMaxPairs || (!wide && D % 64 != 0))) {
    StageArgs a = stage_args(D, w, s);
    a.M0 = s.Am0; a.M1 = s.Am1; a.Xm = s.X; a.zM = zs(s.Am0); a.zXm = zs(s.X);
    return launch_stage_fused(0, a, st);
  }
  if (ga1 && mc) {
    if (mc->a0 != ga0 || mc->a1 != ga1) {
      const size_t n = (size_t)D * D;
      const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
      hipLaunchKernelGGL(k_mid, dim3(blocks, g_batch.nb), dim3(256), 0, st, ga0, ga1, w.AM, n, zs(ga0), zs(w.AM));
      mc->a0 = ga0; mc->a1 = ga1;
    }
    ga0 = w.AM; ga1 = nullptr;
  }
  if (wide && !ga1) {                     // products + element-wise stage in one kernel (the mid-point operand formed above)
    StageArgs a = stage_args(D, w, s);
    a.M0 = ga0; a.Xm = s.X; a.zM = zs(ga0); a.zXm = zs(s.X);
    return launch_stage_fused(1, a, st);
  }
  hipError_t e;
  if (use_library_gemm && !ga1 && g_batch.nb == 1 && !literal_products) {
    // plain GEMM, one rank: cw = D is plain row-major.  Backward: W' = A^T.Psi would be a transposed-A product (slow in
    // the library at D = 1024); Psi is symmetric, so Z = Psi.A = W'^T is computed instead -- the stage kernel uses W'
    // and W'^T symmetrically (R = -G + W' + W'^T), only the order of its two additions changes.
    // (measured: the library's transposed-A product runs at 31 / 66 / 74 TFLOP/s for D = 1024 / 2048 / 4096, its plain
    // product at 50 / 60 / 67 -- the detour pays below D = 2048 only)
    e = s.fwd ? library_gemm(false, D, D, D, ga0, D, s.X, D, w.W, D, st)
              : (D < 2048 ? library_gemm(false, D, D, D, s.X, D, ga0, D, w.W, D, st)
                          : library_gemm(true, D, D, D, ga0, D, s.X, D, w.W, D, st));
  } else {
    GemmArgs g{D, D, D, ga0, ga1, D, s.X, D, w.W, D};
    g.nb = g_batch.nb; g.zA = zs(ga0); g.zB = zs(s.X); g.zC = zs(w.W);
    e = launch_gemm(!s.fwd, g, st);
  }
  if (e != hipSuccess) return e;
  if (literal_products) {
    const int nt = (D + 31) / 32;
    hipLaunchKernelGGL(k_transpose, dim3(nt, nt, g_batch.nb), dim3(256), 0, st, s.X, w.XT, D, zs(s.X), zs(w.XT));
    GemmArgs g2{D, D, D, ga0, ga1, D, w.XT, D, w.W2, D};
    g2.nb = g_batch.nb; g2.zA = zs(ga0); g2.zB = zs(w.XT); g2.zC = zs(w.W2);
    e = launch_gemm(!s.fwd, g2, st);
    if (e != hipSuccess) return e;
  }
  const StageArgs a = stage_args(D, w, s);
  return launch_stage(a, st);
}
}  // namespace

#define LD_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

hipError_t ld_solve_fwd(int method, double dt, int D, int Np, const double* A, const double* b, const double* m0,
                        const double* S0, const double* Sigma, double* m, double* S, double* ws, hipStream_t st) {
  const size_t DD = (size_t)D * D;
  const Work w = carve_work(ws, D);
  const double h = 0.5 * dt;
  for (int p = 0; p < g_batch.nb; p++) {
    LD_TRY(hipMemcpyAsync(S + p * zs(S), S0, DD * sizeof(double), hipMemcpyDeviceToDevice, st));
    LD_TRY(hipMemcpyAsync(m + p * zs(m), m0, D * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  for (int k = 0; k < Np - 1; k++) {
    const double *Ak = A + k * DD, *Ak1 = Ak + DD, *bk = b + (size_t)k * D, *bk1 = bk + D;
    const double *Sk = S + k * DD, *mk = m + (size_t)k * D;
    double *Sn = S + (k + 1) * DD, *mn = m + (size_t)(k + 1) * D;
    StageSpec s{};
    MidCache mc{};
    s.fwd = true; s.E0 = Sigma; s.base = Sk; s.vbase = mk;
    auto set = [&](const double* am0, const double* am1, const double* av0, const double* av1, const double* X,
                   const double* xv, const double* e0, const double* e1, double* out, double* vout, int ks, int fin,
                   double cx, double cf) {
      s.Am0 = am0; s.Am1 = am1; s.Av0 = av0; s.Av1 = av1; s.X = X; s.xv = xv; s.e0 = e0; s.e1 = e1; s.out = out;
      s.vout = vout; s.kstore = ks; s.final_mode = fin; s.cx = cx; s.cf = cf;
    };
    if (method == VGPA_ODE_EULER) {
      set(Ak, nullptr, Ak, nullptr, Sk, mk, bk, nullptr, Sn, mn, 0, 1, 0.0, dt); LD_TRY(run_stage(D, w, s, st, &mc));
    } else if (method == VGPA_ODE_HEUN) {
      set(Ak, nullptr, Ak, nullptr, Sk, mk, bk, nullptr, w.XA, w.xvA, 1, 0, dt, 0.0); LD_TRY(run_stage(D, w, s, st, &mc));
      set(Ak1, nullptr, Ak1, nullptr, w.XA, w.xvA, bk1, nullptr, Sn, mn, 0, 2, 0.0, h); LD_TRY(run_stage(D, w, s, st, &mc));
    } else if (method == VGPA_ODE_RK2) {
      // covariance predictor: S_k stands in for A_k (reference quirk, runge_kutta2.py:96); mean predictor: A_k
      set(Sk, nullptr, Ak, nullptr, Sk, mk, bk, nullptr, w.XA, w.xvA, 0, 0, h, 0.0); LD_TRY(run_stage(D, w, s, st, &mc));
      set(Ak, Ak1, Ak, Ak1, w.XA, w.xvA, bk1, bk, Sn, mn, 0, 1, 0.0, dt); LD_TRY(run_stage(D, w, s, st, &mc));
    } else {
      set(Ak, nullptr, Ak, nullptr, Sk, mk, bk, nullptr, w.XA, w.xvA, 1, 0, h, 0.0); LD_TRY(run_stage(D, w, s, st, &mc));
      set(Ak, Ak1, Ak, Ak1, w.XA, w.xvA, bk1, bk, w.XB, w.xvB, 2, 0, h, 0.0); LD_TRY(run_stage(D, w, s, st, &mc));
      set(Ak, Ak1, Ak, Ak1, w.XB, w.xvB, bk1, bk, w.XA, w.xvA, 3, 0, dt, 0.0); LD_TRY(run_stage(D, w, s, st, &mc));
      set(Ak1, nullptr, Ak1, nullptr, w.XA, w.xvA, bk1, nullptr, Sn, mn, 0, 3, 0.0, dt); LD_TRY(run_stage(D, w, s, st, &mc));
    }
  }
  return hipSuccess;
}

// One backward step t -> t-1.  At/Am = A_t / A_{t-1}, Gt/Gm = dEsde_dS at t / t-1, gt/gmm = dEsde_dm at t / t-1,
// (Pt, lt) = (Psi_t, lam_t), (Pn, ln) = where (Psi_{t-1}, lam_{t-1}) go, (Jn, jn) = jump of index t-1 or nullptr.
hipError_t ld_bwd_step(int method, double dt, int D, const double* At, const double* Am, const double* Gt, const double* Gm,
                       const double* gt, const double* gmm, const double* Pt, const double* lt, double* Pn, double* ln,
                       const double* Jn, const double* jn, double* ws, hipStream_t st) {
  const Work w = carve_work(ws, D);
  const double h = 0.5 * dt;
  StageSpec s{};
  MidCache mc{};
  s.fwd = false; s.base = Pt; s.vbase = lt;
  auto set = [&](const double* a0, const double* a1, const double* X, const double* xv, const double* E0,
                 const double* E1, const double* e0, const double* e1, double* out, double* vout, int ks, int fin,
                 double cx, double cf, bool jump) {
    s.Am0 = a0; s.Am1 = a1; s.Av0 = a0; s.Av1 = a1; s.X = X; s.xv = xv; s.E0 = E0; s.E1 = E1; s.e0 = e0; s.e1 = e1;
    s.out = out; s.vout = vout; s.kstore = ks; s.final_mode = fin; s.cx = cx; s.cf = cf;
    s.J = jump ? Jn : nullptr; s.jv = jump ? jn : nullptr;
  };
  if (method == VGPA_ODE_EULER) {
    set(At, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, Pn, ln, 0, 1, 0.0, dt, true); LD_TRY(run_stage(D, w, s, st, &mc));
  } else if (method == VGPA_ODE_HEUN) {
    set(At, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, w.XA, w.xvA, 1, 0, dt, 0.0, false); LD_TRY(run_stage(D, w, s, st, &mc));
    set(Am, nullptr, w.XA, w.xvA, Gm, nullptr, gmm, nullptr, Pn, ln, 0, 2, 0.0, h, true); LD_TRY(run_stage(D, w, s, st, &mc));
  } else if (method == VGPA_ODE_RK2) {
    set(At, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, w.XA, w.xvA, 0, 0, h, 0.0, false); LD_TRY(run_stage(D, w, s, st, &mc));
    set(Am, At, w.XA, w.xvA, Gt, Gm, gt, gmm, Pn, ln, 0, 1, 0.0, dt, true); LD_TRY(run_stage(D, w, s, st, &mc));
  } else {
    set(At, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, w.XA, w.xvA, 1, 0, h, 0.0, false); LD_TRY(run_stage(D, w, s, st, &mc));
    set(Am, At, w.XA, w.xvA, Gt, Gm, gt, gmm, w.XB, w.xvB, 2, 0, h, 0.0, false); LD_TRY(run_stage(D, w, s, st, &mc));
    set(Am, At, w.XB, w.xvB, Gt, Gm, gt, gmm, w.XA, w.xvA, 3, 0, dt, 0.0, false); LD_TRY(run_stage(D, w, s, st, &mc));
    set(Am, nullptr, w.XA, w.xvA, Gm, nullptr, gmm, nullptr, Pn, ln, 0, 3, 0.0, dt, true); LD_TRY(run_stage(D, w, s, st, &mc));
  }
  return hipSuccess;
}

// Whole backward recursion with every Psi_t stored.  Jumps: dense arrays (jm, js: [Np][D], [Np][D][D]) or, when
// obs_idx != nullptr, sparse ones (obs_idx[t] = n >= 0: jump vector jm[n][:] and the constant matrix js).
hipError_t ld_solve_bwd(int method, double dt, int D, int Np, const double* A, const double* gm, const double* gs,
                        const double* jm, const double* js, double* lam, double* psi, double* ws, hipStream_t st,
                        const int32_t* obs_idx) {
  const size_t DD = (size_t)D * D;
  for (int p = 0; p < g_batch.nb; p++) {
    LD_TRY(hipMemsetAsync(psi + p * zs(psi) + (size_t)(Np - 1) * DD, 0, DD * sizeof(double), st));
    LD_TRY(hipMemsetAsync(lam + p * zs(lam) + (size_t)(Np - 1) * D, 0, D * sizeof(double), st));
  }
  for (int t = Np - 1; t > 0; t--) {
    const double *Jn, *jn;
    if (obs_idx) {
      const int n = obs_idx[t - 1];
      Jn = n >= 0 ? js : nullptr;
      jn = n >= 0 ? jm + (size_t)n * D : nullptr;
    } else {
      Jn = js + (size_t)(t - 1) * DD; jn = jm + (size_t)(t - 1) * D;
    }
    LD_TRY(ld_bwd_step(method, dt, D, A + t * DD, A + (t - 1) * DD, gs + t * DD, gs + (t - 1) * DD, gm + (size_t)t * D,
                       gm + (size_t)(t - 1) * D, psi + t * DD, lam + (size_t)t * D, psi + (t - 1) * DD,
                       lam + (size_t)(t - 1) * D, Jn, jn, ws, st));
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------
// Row-sharded drivers (BASELINE configs[4]): the whole per-step / per-stage loop of a rank, collectives
// included, behind one C-ABI call.  Rank p owns rows I_p = [row0, row0 + Mp) of the stage state and of the products:
//     W[I_p, :]  = A_s[I_p, :] . X  (forward)   |   (A_s^T)[I_p, :] . Psi  (backward)         fp64-MFMA GEMM, packed
//     Wcol       = all_to_all(W[I_p, :])  = W[:, I_p]                                          (D^2/world^2 per peer)
//     X'[I_p, :] = stage kernel (R = -(W + Wcol^T) + Sigma ..., RK slots, the vector recursion in the same launch)
//     X', x'     = the other ranks' row blocks of the next stage state
// Two schedules for that last step:
//   serial    (chunks = 0): ONE grouped all-gather (matrix rows + vector entries) on the compute stream -- every stage is
//             GEMM -> all-to-all -> stage kernel -> all-gather, strictly in sequence (round 2);
//   pipelined (chunks = C >= 1, needs send / recv in the vgpa_comm table): the row block travels as C sub-blocks of Mp / C rows
//             on a SECOND stream (per sub-block one group of per-peer sends / receives, an event behind each), and the next
//             stage's product runs as C K-chunk launches, launch j over sub-block j of EVERY rank's rows (k_gemm_v<.., SEG>:
//             accumulators continued through W), each waiting only for event j.  xGMI is point to point -- all seven incoming
//             blocks arrive at the same time, each on its own link -- so "the first rank's block first" would overlap nothing;
//             with sub-blocks 1/C of every block is complete after 1/C of the transfer time and the product starts then.  The k
//             order of a pipelined product depends on (world, C) only, never on arrival order: results are deterministic, and
//             equal to the serial schedule's up to the order of the fp64 additions (tests: 1e-12).
// Every rank sees the complete S_{k+1} / Psi_{t-1} after the step's last gather and keeps it only if it owns that grid point:
// the history is TIME-sharded (contiguous slices of the grid), which is the layout the time-parallel energy / gradient phase
// wants, at no extra communication.  The collectives come through a small table of function pointers (vgpa_comm): RCCL in
// production (sharded_rccl.cpp: dlopen'ed librccl, unique id through the C ABI), a test double in the virtual-rank tests.
//
// Failure handling (the reference surfaces every error as an exception that ends the run, vgpa_main.py:126-142; a collective
// program must end on EVERY rank): a failing collective callback marks the shard failed, calls the table's `abort` (RCCL:
// ncclCommAbort) and returns VGPA_ERR_COMM; the fused sweep ends with an agreement step (one all-gather of a status word per
// rank: device faults, a non-positive-definite S_t on ANY rank's time slice, allocation failures) so that all ranks return the
// same code; host waits on the stream are bounded by VGPA_SHARD_OPT_TIMEOUT_MS (a peer that died is a time-out, not a hang).
constexpr int kMaxChunks = 8;

struct ShardWork {
  double *Wp, *Wcol, *K1, *K23, *XA, *XB, *cur, *nxt, *mid, *xvA, *xvB, *vcur, *vnxt, *k1v, *k23v, *agree;
};

size_t shard_workspace_doubles(int D, int Mp, int world) {
  const size_t DD = (size_t)D * D, MD = (size_t)Mp * D;
  return 5 * MD + 4 * DD + 4 * (size_t)D + 2 * (size_t)Mp + 2 * (size_t)world + 16;
}

ShardWork carve_shard(double* ws, int D, int Mp) {
  const size_t DD = (size_t)D * D, MD = (size_t)Mp * D;
  ShardWork w;
  w.Wp = ws; w.Wcol = w.Wp + MD; w.K1 = w.Wcol + MD; w.K23 = w.K1 + MD; w.mid = w.K23 + MD;
  w.XA = w.mid + MD; w.XB = w.XA + DD; w.cur = w.XB + DD; w.nxt = w.cur + DD;
  w.xvA = w.nxt + DD; w.xvB = w.xvA + D; w.vcur = w.xvB + D; w.vnxt = w.vcur + D;
  w.k1v = w.vnxt + D; w.k23v = w.k1v + Mp; w.agree = w.k23v + Mp;
  return w;
}

// out[r][c] = 0.5 (a0[r][c] + a1[r][c]), r < rows, c < cols, both operands with leading dimension ld: the mid-point of a row
// block (forward) or of a column block (backward) of A
__global__ void __launch_bounds__(256) k_mid_cols(const double* __restrict__ a0, const double* __restrict__ a1, double* __restrict__ out,
                                                  int rows, int cols, int ld) {
  const size_t n = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / cols, c = i - r * cols;
    out[i] = 0.5 * (a0[r * ld + c] + a1[r * ld + c]);
  }
}

struct ShardCtx {
  int method = 0, D = 0, rank = 0, world = 1, row0 = 0, Mp = 0;
  double dt = 0.0;
  vgpa_comm comm{};
  hipStream_t st = nullptr;                 // compute stream (also carries the all-to-all)
  hipStream_t cs = nullptr;                 // communication stream of the pipelined gather
  ShardWork w{};
  const double* mid_a0 = nullptr; const double* mid_a1 = nullptr; bool mid_fwd = false;
  int chunks = 0;                           // sub-blocks of the pipelined gather (0: serial schedule)
  hipEvent_t ev_stage = nullptr, ev_tail = nullptr, ev_chunk[kMaxChunks] = {};
  const double* pending = nullptr;          // stage buffer whose remote rows are still arriving on cs (ev_chunk[j] per sub-block)
  bool cs_dirty = false;                    // work enqueued on cs that the compute stream has not joined yet
  bool failed = false;                      // a collective failed: the communicator is gone (VGPA_ERR_COMM from now on)
  int64_t timeout_ms = 600000;              // bound of every host wait on the streams
  bool skip_comm = false;                   // timing only (vgpa_shard_time_stage): a stage's kernels without its collectives
  hipEvent_t ev_phase[7] = {};              // phase boundaries of the last fused sweep (vgpa_shard_phase_ms)
  bool phases_valid = false;
  int group_depth = 0;                      // group_begin calls of this thread that have no group_end yet
  double* pinned = nullptr;                 // page-locked host staging: world status words + F (async copies never target
                                            // pageable, stack or local-vector memory: they may outlive the call that queued them)
};

static int group_begin(ShardCtx& c) {
  if (!c.comm.group_begin) return 0;
  const int rc = c.comm.group_begin(c.comm.user);
  if (rc == 0) c.group_depth++;
  return rc;
}
static int group_end(ShardCtx& c) {
  if (!c.comm.group_end) return 0;
  if (c.group_depth > 0) c.group_depth--;
  return c.comm.group_end(c.comm.user);
}

// the communicator goes down; a group this thread still has open is closed first (its code no longer matters) so that the abort
// -- and whatever else shares the library on this thread afterwards -- does not run inside an unbalanced group
static void comm_failed(ShardCtx& c) {
  while (c.group_depth > 0) { c.group_depth--; if (c.comm.group_end) (void)c.comm.group_end(c.comm.user); }
  if (!c.failed && c.comm.abort) (void)c.comm.abort(c.comm.user);
  c.failed = true;
}
#define SH_COMM(c, expr) do { if ((expr) != 0) { comm_failed(c); return hipErrorUnknown; } } while (0)

static bool pipelined(const ShardCtx& c) { return c.world > 1 && c.chunks > 0; }

// the largest number of sub-blocks <= want that the K-chunk product can use: whole k-tiles per sub-block
static int usable_chunks(const ShardCtx& c, int want) {
  if (c.world < 2 || !c.comm.send || !c.comm.recv || !c.comm.group_begin || !c.comm.group_end) return 0;
  if (want > kMaxChunks) want = kMaxChunks;
  for (int n = want; n >= 1; n--)
    if (c.Mp % (n * BK) == 0) return n;
  return 0;
}

// the compute stream waits for everything enqueued on the communication stream
static hipError_t shard_join(ShardCtx& c) {
  if (c.cs_dirty) {
    LD_TRY(hipEventRecord(c.ev_tail, c.cs));
    LD_TRY(hipStreamWaitEvent(c.st, c.ev_tail, 0));
  }
  c.cs_dirty = false; c.pending = nullptr;
  return hipSuccess;
}

// host wait on the compute stream, bounded: a peer that never arrives must not hang this rank for ever
static int shard_wait(ShardCtx& c) {
  if (c.timeout_ms <= 0) return hipStreamSynchronize(c.st) == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(c.st);
    if (e == hipSuccess) return VGPA_OK;
    if (e != hipErrorNotReady) return VGPA_ERR_DEVICE;
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (ms > c.timeout_ms) { comm_failed(c); return VGPA_ERR_COMM; }
    if (ms > 2) std::this_thread::sleep_for(std::chrono::microseconds(ms > 50 ? 500 : 50));
  }
}

// One stage on this rank's row block; every pointer already addresses the rank's part.
struct ShardStage {
  bool fwd;
  const double *Ag0, *Ag1; int ldag;     // product operand: rows I_p of A_s [Mp][ldag] (fwd) / columns I_p of A_s [D][ldag] (bwd); Ag1: mid-point partner
  const double *Av0, *Av1; int ldav;     // vector recursion: rows I_p of A_s [Mp][ldav]
  const double* X; const double* xv;     // complete stage state [D][D] / [D]
  const double *E0, *E1;                 // rows I_p of Sigma / dEsde_dS(stage)   [Mp][D]
  const double *e0, *e1;                 // entries I_p of b / dEsde_dm(stage)
  const double* J; const double* jv;     // rows / entries I_p of the jump, or nullptr
  const double* base; const double* vbase;   // complete buffers of S_k / Psi_t, m_k / lam_t (the row block is taken here)
  double* out; double* vout;             // complete buffers of the next stage state
  int kstore, final_mode; double cx, cf;
};

static hipError_t shard_gather(ShardCtx& c, double* out, double* vout) {
  const int D = c.D, Mp = c.Mp, row0 = c.row0, world = c.world, rank = c.rank;
  if (!pipelined(c)) {      // complete the next stage state: row blocks of the matrix and of the vector, one group
    SH_COMM(c, group_begin(c));
    SH_COMM(c, c.comm.all_gather(c.comm.user, out + (size_t)row0 * D, out, (uint64_t)Mp * D, c.st));
    SH_COMM(c, c.comm.all_gather(c.comm.user, vout + row0, vout, (uint64_t)Mp, c.st));
    SH_COMM(c, group_end(c));
    return hipSuccess;
  }
  LD_TRY(hipEventRecord(c.ev_stage, c.st));
  LD_TRY(hipStreamWaitEvent(c.cs, c.ev_stage, 0));
  const int sub = Mp / c.chunks;
  for (int j = 0; j < c.chunks; j++) {
    SH_COMM(c, group_begin(c));
    for (int d = 1; d < world; d++) {      // rotated peer order: every rank's d-th send meets its peer's d-th receive
      const int to = (rank + d) % world, from = (rank - d + world) % world;
      SH_COMM(c, c.comm.send(c.comm.user, out + ((size_t)row0 + (size_t)j * sub) * D, (uint64_t)sub * D, to, c.cs));
      SH_COMM(c, c.comm.recv(c.comm.user, out + ((size_t)from * Mp + (size_t)j * sub) * D, (uint64_t)sub * D, from, c.cs));
      if (j == 0) {                        // the vector travels with the first sub-block
        SH_COMM(c, c.comm.send(c.comm.user, vout + row0, (uint64_t)Mp, to, c.cs));
        SH_COMM(c, c.comm.recv(c.comm.user, vout + (size_t)from * Mp, (uint64_t)Mp, from, c.cs));
      }
    }
    SH_COMM(c, group_end(c));
    LD_TRY(hipEventRecord(c.ev_chunk[j], c.cs));
  }
  c.pending = out; c.cs_dirty = true;
  return hipSuccess;
}

static hipError_t shard_stage(ShardCtx& c, const ShardStage& s) {
  const int D = c.D, Mp = c.Mp, row0 = c.row0;
  const ShardWork& w = c.w;
  GemmArgs g{};
  g.M = Mp; g.N = D; g.K = D; g.B = s.X; g.ldb = D; g.C = w.Wp; g.cw = Mp; g.A1 = nullptr;
  if (s.Ag1) {      // mid-point operand of this rank's slab, formed once per step
    if (c.mid_a0 != s.Ag0 || c.mid_a1 != s.Ag1 || c.mid_fwd != s.fwd) {
      const size_t n = (size_t)Mp * D;
      const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
      if (s.fwd) hipLaunchKernelGGL(k_mid_cols, dim3(blocks), dim3(256), 0, c.st, s.Ag0, s.Ag1, w.mid, Mp, D, s.ldag);
      else hipLaunchKernelGGL(k_mid_cols, dim3(blocks), dim3(256), 0, c.st, s.Ag0, s.Ag1, w.mid, D, Mp, s.ldag);
      c.mid_a0 = s.Ag0; c.mid_a1 = s.Ag1; c.mid_fwd = s.fwd;
    }
    g.A0 = w.mid; g.lda = s.fwd ? D : Mp;
  } else {
    g.A0 = s.Ag0; g.lda = s.ldag;
  }
  if (pipelined(c)) {
    // K-chunk launches: launch j multiplies sub-block j of every rank's rows of X and waits only for that sub-block's event
    const int sub = Mp / c.chunks;
    for (int j = 0; j < c.chunks; j++) {
      if (c.pending == s.X) LD_TRY(hipStreamWaitEvent(c.st, c.ev_chunk[j], 0));
      GemmArgs h = g;
      h.K = c.world * sub; h.seg_tiles = sub / BK; h.seg_stride = Mp; h.accumulate = j > 0;
      h.A0 = g.A0 + (s.fwd ? (size_t)j * sub : (size_t)j * sub * g.lda);     // k runs along A's columns (NN) / rows (TN)
      h.B = s.X + (size_t)j * sub * D;
      LD_TRY(launch_gemm(!s.fwd, h, c.st));
    }
    if (c.pending == s.X) c.pending = nullptr;      // every sub-block's event has been waited for
  } else {
    LD_TRY(launch_gemm(!s.fwd, g, c.st));
  }
  const double* wcol = w.Wp;
  if (c.world > 1 && !c.skip_comm) {
    SH_COMM(c, c.comm.all_to_all(c.comm.user, w.Wp, w.Wcol, (uint64_t)Mp * Mp, c.st));
    wcol = w.Wcol;
  }
  StageArgs a{};
  a.D = D; a.row0 = 0; a.Mp = Mp; a.cw = Mp; a.fwd = s.fwd ? 1 : 0; a.kstore = s.kstore; a.final = s.final_mode;
  a.sym_ok = 1;
  a.mid_e = s.E1 != nullptr; a.has_j = s.J != nullptr; a.cx = s.cx; a.cf = s.cf;
  a.W = w.Wp; a.Wcol = wcol;
  a.E0 = s.E0; a.E1 = s.E1; a.J = s.J;
  a.base = s.base + (size_t)row0 * D; a.K1 = w.K1; a.K23 = w.K23; a.out = s.out + (size_t)row0 * D;
  a.A0 = s.Av0; a.A1 = s.Av1; a.lda = s.ldav; a.mid_a = s.Av1 != nullptr; a.x = s.xv;
  a.e0 = s.e0; a.e1 = s.e1; a.mid_ev = s.e1 != nullptr;
  a.jv = s.jv; a.vbase = s.vbase + row0;
  a.k1v = w.k1v; a.k23v = w.k23v; a.vout = s.vout + row0;
  LD_TRY(launch_stage(a, c.st));
  if (c.world > 1 && !c.skip_comm) LD_TRY(shard_gather(c, s.out, s.vout));
  return hipSuccess;
}

static void time_slice(int Np, int rank, int world, int* lo, int* hi) {
  const int base = Np / world, rem = Np % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

// Where a rank finds ITS parts of the time-indexed linear term A_t: replicated input (views into the caller's [Np][D][D]) or
// the row / column blocks a time -> block exchange has delivered (the memory-sharded sweep).
struct AView {
  const double* rows; size_t rows_t; int rows_ld;     // rows I_p of A_t:    rows + t * rows_t, [Mp][rows_ld]
  const double* cols; size_t cols_t; int cols_ld;     // columns I_p of A_t: cols + t * cols_t, [D][cols_ld]
};
static AView aview_replicated(const ShardCtx& c, const double* A) {
  const size_t DD = (size_t)c.D * c.D;
  return AView{A + (size_t)c.row0 * c.D, DD, c.D, A + c.row0, DD, c.D};
}
static AView aview_blocks(const ShardCtx& c, const double* a_rows, const double* a_cols) {
  const size_t MD = (size_t)c.Mp * c.D;
  return AView{a_rows, MD, c.D, a_cols, MD, c.Mp};
}

// a copy of a complete stage buffer into the rank's time slice: on the stream the buffer completes on
static hipError_t shard_keep(ShardCtx& c, const double* M, const double* v, double* M_own, double* v_own) {
  const size_t DD = (size_t)c.D * c.D;
  hipStream_t ks = c.st;
  if (pipelined(c) && c.pending == M) { ks = c.cs; c.cs_dirty = true; }
  LD_TRY(hipMemcpyAsync(M_own, M, DD * sizeof(double), hipMemcpyDeviceToDevice, ks));
  return hipMemcpyAsync(v_own, v, c.D * sizeof(double), hipMemcpyDeviceToDevice, ks);
}

// (m_t, S_t): every rank steps the whole grid, keeps the grid points [t_lo, t_hi) it owns in m_own / S_own.
// b: complete [Np][D].
hipError_t shard_solve_fwd(ShardCtx& c, int Np, const AView& av, const double* b, const double* m0, const double* S0,
                           const double* Sigma, double* m_own, double* S_own) {
  const int D = c.D, row0 = c.row0;
  const size_t DD = (size_t)D * D;
  const double dt = c.dt, h = 0.5 * dt;
  ShardWork& w = c.w;
  int lo, hi;
  time_slice(Np, c.rank, c.world, &lo, &hi);
  c.mid_a0 = c.mid_a1 = nullptr;
  LD_TRY(shard_join(c));
  LD_TRY(hipMemcpyAsync(w.cur, S0, DD * sizeof(double), hipMemcpyDeviceToDevice, c.st));
  LD_TRY(hipMemcpyAsync(w.vcur, m0, D * sizeof(double), hipMemcpyDeviceToDevice, c.st));
  auto keep = [&](int t, const double* S, const double* m) -> hipError_t {
    if (t < lo || t >= hi) return hipSuccess;
    return shard_keep(c, S, m, S_own + (size_t)(t - lo) * DD, m_own + (size_t)(t - lo) * D);
  };
  LD_TRY(keep(0, w.cur, w.vcur));
  for (int k = 0; k < Np - 1; k++) {
    const double *Ar = av.rows + k * av.rows_t, *Ar1 = Ar + av.rows_t;
    const double *bk = b + (size_t)k * D + row0, *bk1 = bk + D;
    const double *Sk = w.cur, *mk = w.vcur;
    double *Sn = w.nxt, *mn = w.vnxt;
    ShardStage s{};
    s.fwd = true; s.E0 = Sigma + (size_t)row0 * D; s.base = Sk; s.vbase = mk; s.ldag = av.rows_ld; s.ldav = av.rows_ld;
    auto set = [&](const double* ag0, const double* ag1, int ldag, const double* av0, const double* av1, const double* X,
                   const double* xv, const double* e0, const double* e1, double* out, double* vout, int ks, int fin,
                   double cx, double cf) {
      s.Ag0 = ag0; s.Ag1 = ag1; s.ldag = ldag; s.Av0 = av0; s.Av1 = av1; s.X = X; s.xv = xv; s.e0 = e0; s.e1 = e1; s.out = out;
      s.vout = vout; s.kstore = ks; s.final_mode = fin; s.cx = cx; s.cf = cf;
    };
    const int la = av.rows_ld;
    if (c.method == VGPA_ODE_EULER) {
      set(Ar, nullptr, la, Ar, nullptr, Sk, mk, bk, nullptr, Sn, mn, 0, 1, 0.0, dt); LD_TRY(shard_stage(c, s));
    } else if (c.method == VGPA_ODE_HEUN) {
      set(Ar, nullptr, la, Ar, nullptr, Sk, mk, bk, nullptr, w.XA, w.xvA, 1, 0, dt, 0.0); LD_TRY(shard_stage(c, s));
      set(Ar1, nullptr, la, Ar1, nullptr, w.XA, w.xvA, bk1, nullptr, Sn, mn, 0, 2, 0.0, h); LD_TRY(shard_stage(c, s));
    } else if (c.method == VGPA_ODE_RK2) {
      // covariance predictor: S_k stands in for A_k (reference quirk, runge_kutta2.py:96); mean predictor: A_k
      set(Sk + (size_t)row0 * D, nullptr, D, Ar, nullptr, Sk, mk, bk, nullptr, w.XA, w.xvA, 0, 0, h, 0.0); LD_TRY(shard_stage(c, s));
      set(Ar, Ar1, la, Ar, Ar1, w.XA, w.xvA, bk1, bk, Sn, mn, 0, 1, 0.0, dt); LD_TRY(shard_stage(c, s));
    } else {
      set(Ar, nullptr, la, Ar, nullptr, Sk, mk, bk, nullptr, w.XA, w.xvA, 1, 0, h, 0.0); LD_TRY(shard_stage(c, s));
      set(Ar, Ar1, la, Ar, Ar1, w.XA, w.xvA, bk1, bk, w.XB, w.xvB, 2, 0, h, 0.0); LD_TRY(shard_stage(c, s));
      set(Ar, Ar1, la, Ar, Ar1, w.XB, w.xvB, bk1, bk, w.XA, w.xvA, 3, 0, dt, 0.0); LD_TRY(shard_stage(c, s));
      set(Ar1, nullptr, la, Ar1, nullptr, w.XA, w.xvA, bk1, nullptr, Sn, mn, 0, 3, 0.0, dt); LD_TRY(shard_stage(c, s));
    }
    LD_TRY(keep(k + 1, Sn, mn));
    double* t1 = w.cur; w.cur = w.nxt; w.nxt = t1;
    double* t2 = w.vcur; w.vcur = w.vnxt; w.vnxt = t2;
  }
  return shard_join(c);
}

// Inputs of the backward recursion, already reduced to what THIS rank reads: rows I_p of dEsde_dS[t] (a view into the
// replicated operator-level array, or the row blocks the time -> row exchange of the fused sweep delivered), the complete
// dEsde_dm [Np][D], and the jumps either dense (operator level; zero rows off the observations) or sparse: observation index
// per grid point (host), one vector per observation, one constant matrix.
struct BwdIn {
  const double* gm = nullptr;                                       // [Np][D]
  const double* g_rows = nullptr; size_t g_rows_t = 0;              // rows I_p of dEsde_dS[t]: g_rows + t * g_rows_t, [Mp][D]
  const double* jm_dense = nullptr; const double* js_dense = nullptr;
  const int32_t* obs_idx = nullptr; const double* jm_sparse = nullptr; const double* js_const = nullptr;
};

// (lam_t, Psi_t); same ownership of the grid as the forward recursion.
hipError_t shard_solve_bwd(ShardCtx& c, int Np, const AView& av, const BwdIn& in, double* lam_own, double* psi_own) {
  const int D = c.D, row0 = c.row0;
  const size_t DD = (size_t)D * D;
  const double dt = c.dt, h = 0.5 * dt;
  ShardWork& w = c.w;
  int lo, hi;
  time_slice(Np, c.rank, c.world, &lo, &hi);
  c.mid_a0 = c.mid_a1 = nullptr;
  LD_TRY(shard_join(c));
  LD_TRY(hipMemsetAsync(w.cur, 0, DD * sizeof(double), c.st));
  LD_TRY(hipMemsetAsync(w.vcur, 0, D * sizeof(double), c.st));
  auto keep = [&](int t, const double* P, const double* l) -> hipError_t {
    if (t < lo || t >= hi) return hipSuccess;
    return shard_keep(c, P, l, psi_own + (size_t)(t - lo) * DD, lam_own + (size_t)(t - lo) * D);
  };
  LD_TRY(keep(Np - 1, w.cur, w.vcur));
  for (int t = Np - 1; t > 0; t--) {
    const double *Act = av.cols + t * av.cols_t, *Acm = av.cols + (t - 1) * av.cols_t;      // product operand: columns I_p
    const double *Art = av.rows + t * av.rows_t, *Arm = av.rows + (t - 1) * av.rows_t;      // vector recursion: rows I_p (Q3: A.lam)
    const double *Gt = in.g_rows + t * in.g_rows_t, *Gm = in.g_rows + (t - 1) * in.g_rows_t;
    const double *gt = in.gm + (size_t)t * D + row0, *gmm = in.gm + (size_t)(t - 1) * D + row0;
    const int nobs = in.obs_idx ? in.obs_idx[t - 1] : 0;        // sparse jumps: only behind a step that ends at an observation
    const bool has_jump = in.obs_idx ? nobs >= 0 : true;
    const double* Jn = (in.obs_idx ? in.js_const : in.js_dense + (size_t)(t - 1) * DD) + (size_t)row0 * D;
    const double* jn = (in.obs_idx ? in.jm_sparse + (size_t)(nobs >= 0 ? nobs : 0) * D : in.jm_dense + (size_t)(t - 1) * D) + row0;
    const double *Pt = w.cur, *lt = w.vcur;
    double *Pn = w.nxt, *ln = w.vnxt;
    ShardStage s{};
    s.fwd = false; s.base = Pt; s.vbase = lt; s.ldag = av.cols_ld; s.ldav = av.rows_ld;
    auto set = [&](const double* ag0, const double* ag1, const double* av0, const double* av1, const double* X, const double* xv,
                   const double* E0, const double* E1, const double* e0, const double* e1, double* out, double* vout, int ks,
                   int fin, double cx, double cf, bool jump) {
      s.Ag0 = ag0; s.Ag1 = ag1; s.Av0 = av0; s.Av1 = av1; s.X = X; s.xv = xv; s.E0 = E0; s.E1 = E1; s.e0 = e0; s.e1 = e1;
      s.out = out; s.vout = vout; s.kstore = ks; s.final_mode = fin; s.cx = cx; s.cf = cf;
      s.J = (jump && has_jump) ? Jn : nullptr; s.jv = (jump && has_jump) ? jn : nullptr;
    };
    if (c.method == VGPA_ODE_EULER) {
      set(Act, nullptr, Art, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, Pn, ln, 0, 1, 0.0, dt, true); LD_TRY(shard_stage(c, s));
    } else if (c.method == VGPA_ODE_HEUN) {
      set(Act, nullptr, Art, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, w.XA, w.xvA, 1, 0, dt, 0.0, false); LD_TRY(shard_stage(c, s));
      set(Acm, nullptr, Arm, nullptr, w.XA, w.xvA, Gm, nullptr, gmm, nullptr, Pn, ln, 0, 2, 0.0, h, true); LD_TRY(shard_stage(c, s));
    } else if (c.method == VGPA_ODE_RK2) {
      set(Act, nullptr, Art, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, w.XA, w.xvA, 0, 0, h, 0.0, false); LD_TRY(shard_stage(c, s));
      set(Acm, Act, Arm, Art, w.XA, w.xvA, Gt, Gm, gt, gmm, Pn, ln, 0, 1, 0.0, dt, true); LD_TRY(shard_stage(c, s));
    } else {
      set(Act, nullptr, Art, nullptr, Pt, lt, Gt, nullptr, gt, nullptr, w.XA, w.xvA, 1, 0, h, 0.0, false); LD_TRY(shard_stage(c, s));
      set(Acm, Act, Arm, Art, w.XA, w.xvA, Gt, Gm, gt, gmm, w.XB, w.xvB, 2, 0, h, 0.0, false); LD_TRY(shard_stage(c, s));
      set(Acm, Act, Arm, Art, w.XB, w.xvB, Gt, Gm, gt, gmm, w.XA, w.xvA, 3, 0, dt, 0.0, false); LD_TRY(shard_stage(c, s));
      set(Acm, nullptr, Arm, nullptr, w.XA, w.xvA, Gm, nullptr, gmm, nullptr, Pn, ln, 0, 3, 0.0, dt, true); LD_TRY(shard_stage(c, s));
    }
    LD_TRY(keep(t - 1, Pn, ln));
    double* t1 = w.cur; w.cur = w.nxt; w.nxt = t1;
    double* t2 = w.vcur; w.vcur = w.vnxt; w.vnxt = t2;
  }
  return shard_join(c);
}

// ---- time -> row / column block exchange ------------------------------------------------------------------------------------
// A time-sharded array (rank q holds the D x D matrices of its grid points) becomes a BLOCK-sharded one (rank p holds rows I_p --
// mode 0 -- or columns I_p -- mode 1 -- of EVERY grid point): what the row-sharded recursions read of dEsde_dS (rows), of A_t
// (rows forward, columns backward).  Per batch of T own grid points: pack [peer][i][Mp * D] -> ONE all-to-all -> the received
// pieces are contiguous runs of the destination.  1/world of the bytes of an all-gather, and no rank ever holds a complete
// (Np, D, D) array.
__global__ void __launch_bounds__(256) k_xpack(int mode, int D, int Mp, int T, int n_valid, const double* __restrict__ src,
                                               double* __restrict__ dst) {
  const size_t DD = (size_t)D * D, MD = (size_t)Mp * D;
  const size_t total = (size_t)n_valid * DD;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t i = e / DD, rem = e - i * DD;
    const int r = (int)(rem / D), col = (int)(rem - (size_t)r * D);
    size_t o;
    if (mode == 0) { const int q = r / Mp; o = ((size_t)q * T + i) * MD + (size_t)(r - q * Mp) * D + col; }
    else { const int q = col / Mp; o = ((size_t)q * T + i) * MD + (size_t)r * Mp + (col - q * Mp); }
    dst[o] = src[e];
  }
}

static int exchange_batch(int D, double budget_bytes, int pad) {
  const double per = 2.0 * 8.0 * (double)D * (double)D;          // send + receive staging per grid point of a batch
  int T = (int)(budget_bytes / per);
  if (T < 1) T = 1;
  return T < pad ? T : (pad > 0 ? pad : 1);
}

// src_own [n_own][D][D] -> dst [Np][Mp][D] (mode 0) / [Np][D][Mp] (mode 1); xs / xr: staging of T * D * D doubles each
static hipError_t exchange_time_to_blocks(ShardCtx& c, int Np, int mode, const double* src_own, double* dst, double* xs, double* xr,
                                          int T) {
  const int D = c.D, Mp = c.Mp, world = c.world;
  const size_t DD = (size_t)D * D, MD = (size_t)Mp * D;
  int lo, hi;
  time_slice(Np, c.rank, world, &lo, &hi);
  const int n_own = hi - lo, pad = (Np + world - 1) / world;
  for (int t0 = 0; t0 < pad; t0 += T) {
    const int n_mine = n_own - t0 < 0 ? 0 : (n_own - t0 < T ? n_own - t0 : T);
    if (n_mine > 0) {
      const size_t total = (size_t)n_mine * DD;
      const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
      double* out = world > 1 ? xs : dst + (size_t)(lo + t0) * MD;      // one rank: the packed layout IS the destination
      hipLaunchKernelGGL(k_xpack, dim3(blocks), dim3(256), 0, c.st, mode, D, Mp, world > 1 ? T : n_mine, n_mine, src_own + (size_t)t0 * DD, out);
    }
    if (world == 1) continue;
    SH_COMM(c, c.comm.all_to_all(c.comm.user, xs, xr, (uint64_t)T * MD, c.st));
    for (int q = 0; q < world; q++) {
      int qlo, qhi;
      time_slice(Np, q, world, &qlo, &qhi);
      const int n_q = (qhi - qlo) - t0 < 0 ? 0 : ((qhi - qlo) - t0 < T ? (qhi - qlo) - t0 : T);
      if (n_q > 0)
        LD_TRY(hipMemcpyAsync(dst + (size_t)(qlo + t0) * MD, xr + (size_t)q * T * MD, (size_t)n_q * MD * sizeof(double),
                              hipMemcpyDeviceToDevice, c.st));
    }
  }
  return hipGetLastError();
}

// ---- the fused sweep of ONE Lorenz-96 problem on the row-sharded recursion -----------------------------------------------
// Observation terms with diagonal R and H = I (gaussian_like.py:87-137), time-sharded: observation n needs m at its grid
// point t_n (owned by one rank) and -- quirk Q4 -- the diagonal of S at grid index n (owned by possibly another one).  Every
// rank writes what it owns (zeros otherwise) into its part of the gather buffer: jm[n][:] and the scalar term of n.
__global__ void __launch_bounds__(256) k_shard_obs(int D, int lo, int hi, const int64_t* __restrict__ obs_t, const double* __restrict__ obs_y,
                                                   const double* __restrict__ rinv, const double* __restrict__ m_own, const double* __restrict__ S_own,
                                                   double* __restrict__ jm, double* __restrict__ part) {
  __shared__ double red[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int64_t tn = obs_t[n];
  const bool own_m = tn >= lo && tn < hi, own_s = n >= lo && n < hi;
  double acc = 0.0;
  for (int i = tid; i < D; i += 256) {
    double j = 0.0;
    if (own_m) {
      const double w = obs_y[(size_t)n * D + i] - m_own[(size_t)(tn - lo) * D + i];
      j = -(rinv[i] * w);
      acc += w * (rinv[i] * w);
    }
    jm[(size_t)n * D + i] = j;
    if (own_s) acc += rinv[i] * S_own[((size_t)(n - lo) * D + i) * D + i];
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) part[n] = red[0];
}

// sums over the ranks' parts (fixed order): jm [M][D], eobs = 0.5 (sum_n term_n + const)
__global__ void __launch_bounds__(256) k_shard_obs_sum(int D, int M, int world, const double* __restrict__ all, double obs_const,
                                                       double* __restrict__ jm, double* __restrict__ eobs) {
  __shared__ double red[256];
  const size_t cnt = (size_t)M * (D + 1);
  const int tid = threadIdx.x;
  for (size_t e = (size_t)blockIdx.x * 256 + tid; e < (size_t)M * D; e += (size_t)gridDim.x * 256) {
    double v = 0.0;
    for (int q = 0; q < world; q++) v += all[q * cnt + e];
    jm[e] = v;
  }
  if (blockIdx.x == 0) {
    double acc = 0.0;
    for (int n = tid; n < M; n += 256)
      for (int q = 0; q < world; q++) acc += all[q * cnt + (size_t)M * D + n];
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) eobs[0] = 0.5 * (red[0] + obs_const);
  }
}

// [world][pad_len][W] gathered time slices -> [Np][W]
__global__ void __launch_bounds__(256) k_shard_unpad(int Np, int W, int world, int pad_len, const double* __restrict__ padded, double* __restrict__ full) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)Np * W) return;
  const int t = (int)(e / W), i = (int)(e - (size_t)t * W);
  const int base = Np / world, rem = Np % world;
  const int q = (t < rem * (base + 1)) ? t / (base + 1) : rem + (base ? (t - rem * (base + 1)) / base : 0);
  const int lo = q * base + (q < rem ? q : rem);
  full[e] = padded[((size_t)q * pad_len + (t - lo)) * W + i];
}

__global__ void k_diag_half(int D, const double* __restrict__ rinv, double* __restrict__ jsc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < D) jsc[(size_t)i * D + i] = 0.5 * rinv[i];
}

// this rank's word of the agreement step: the larger of the host's local code and the device status (bit 0: S_t not PD)
__global__ void k_shard_status(const int32_t* __restrict__ status, double local_code, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double dev = (status && (status[0] & 1)) ? 3.0 : 0.0;       // 3 = -VGPA_ERR_NOT_PD
    out[0] = dev > local_code ? dev : local_code;
  }
}

// Collective: every rank contributes -code (0 = fine), all receive the largest.  Synchronises the compute stream.
// `f_dev` (optional): a device scalar that travels to the host with the status words, returned in *f_host on VGPA_OK only.
static int shard_agree(ShardCtx& c, const int32_t* status_dev, int local_code, const double* f_dev = nullptr, double* f_host = nullptr) {
  double* mine = c.w.agree + c.rank;
  if (shard_join(c) != hipSuccess) return VGPA_ERR_DEVICE;
  hipLaunchKernelGGL(k_shard_status, dim3(1), dim3(64), 0, c.st, status_dev, (double)(-local_code), mine);
  if (c.world > 1) {
    if (c.failed) return VGPA_ERR_COMM;
    if (c.comm.all_gather(c.comm.user, mine, c.w.agree, 1, c.st) != 0) { comm_failed(c); return VGPA_ERR_COMM; }
  }
  if (!c.pinned) return VGPA_ERR_DEVICE;
  if (hipMemcpyAsync(c.pinned, c.w.agree, sizeof(double) * c.world, hipMemcpyDeviceToHost, c.st) != hipSuccess) return VGPA_ERR_DEVICE;
  if (f_dev && hipMemcpyAsync(c.pinned + c.world, f_dev, sizeof(double), hipMemcpyDeviceToHost, c.st) != hipSuccess) return VGPA_ERR_DEVICE;
  const int rc = shard_wait(c);       // (on a time-out the copies stay queued: their target lives as long as the shard)
  if (rc != VGPA_OK) return rc;
  double worst = 0.0;
  for (int q = 0; q < c.world; q++) worst = c.pinned[q] > worst ? c.pinned[q] : worst;
  if (f_dev && f_host) *f_host = c.pinned[c.world];
  return -(int)worst;
}

struct SweepBuffers {       // device memory of the fused sweep, allocated on first use
  double *m_own = nullptr, *S_own = nullptr, *lam_own = nullptr, *Ef_own = nullptr;
  double *gs_own = nullptr;  // dEsde_dS of the own grid points; dead after the time -> row exchange: Psi of the own grid points lives there
  double *g_rows = nullptr;  // [Np][Mp][D] rows I_p of dEsde_dS of every grid point
  double *gm_all = nullptr, *gm_full = nullptr, *et_all = nullptr, *et_full = nullptr, *obuf = nullptr, *jm = nullptr, *jsc = nullptr;
  double *xs = nullptr, *xr = nullptr;      // staging of the exchanges
  double *a_rows = nullptr, *a_cols = nullptr, *b_all = nullptr, *b_full = nullptr;   // memory-sharded x only
  double *scal = nullptr;   // eobs, esde, f
  double* lde_ws = nullptr;
  int64_t* obs_t = nullptr;
  int32_t* status = nullptr;
  int lde_nb = 0, M = -1, xT = 1;
  bool sharded_x = false;
};

struct SweepProblem {
  double theta, obs_const, e0;
  const double *isg, *m0, *S0, *Sigma, *obs_y, *rinv;
  int M; const int64_t* obs_t_host;
};

// forward (row-sharded) -> observation terms + E_sde terms of the own grid points -> time -> row exchange of dEsde_dS ->
// backward (row-sharded) -> gradient of the own grid points -> F.  x either replicated (x_full = [A | b] of the whole grid) or
// memory-sharded (a_own / b_own: the own grid points only; rows and columns of A_t reach the recursions through two more
// exchanges).  Everything is enqueued on the shard's streams; the caller runs the agreement step.
hipError_t shard_sweep(ShardCtx& c, int Np, SweepBuffers& B, const SweepProblem& p, const double* x_full, const double* a_own,
                       const double* b_own, const std::vector<int32_t>& obs_idx, double* gA_own, double* gB_own) {
  const int D = c.D, world = c.world;
  const size_t DD = (size_t)D * D;
  int lo, hi;
  time_slice(Np, c.rank, world, &lo, &hi);
  const int n_own = hi - lo, pad = (Np + world - 1) / world;
  hipStream_t st = c.st;
  auto unpad = [&](int W, const double* padded, double* full) {
    const size_t n = (size_t)Np * W;
    hipLaunchKernelGGL(k_shard_unpad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Np, W, world, pad, padded, full);
  };
  LD_TRY(hipMemsetAsync(B.status, 0, sizeof(int32_t), st));
  c.phases_valid = false;
  auto mark = [&](int i) -> hipError_t { return hipEventRecord(c.ev_phase[i], st); };      // (the recursions end joined: st has waited for cs)
  LD_TRY(mark(0));
  AView av;
  const double* b_full;
  if (x_full) {
    av = aview_replicated(c, x_full);
    b_full = x_full + (size_t)Np * DD;
    a_own = x_full + (size_t)lo * DD;
    b_own = b_full + (size_t)lo * D;
  } else if (world == 1) {
    av = aview_replicated(c, a_own);
    b_full = b_own;
  } else {
    LD_TRY(exchange_time_to_blocks(c, Np, 0, a_own, B.a_rows, B.xs, B.xr, B.xT));
    LD_TRY(exchange_time_to_blocks(c, Np, 1, a_own, B.a_cols, B.xs, B.xr, B.xT));
    av = aview_blocks(c, B.a_rows, B.a_cols);
    double* mine = B.b_all + (size_t)c.rank * pad * D;
    if (n_own > 0) LD_TRY(hipMemcpyAsync(mine, b_own, (size_t)n_own * D * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (world > 1) SH_COMM(c, c.comm.all_gather(c.comm.user, mine, B.b_all, (uint64_t)pad * D, st));
    unpad(D, B.b_all, B.b_full);
    b_full = B.b_full;
  }
  LD_TRY(mark(1));
  LD_TRY(shard_solve_fwd(c, Np, av, b_full, p.m0, p.S0, p.Sigma, B.m_own, B.S_own));
  LD_TRY(mark(2));
  // observation terms
  const size_t ocnt = (size_t)p.M * (D + 1);
  if (p.M > 0) {
    double* mine = B.obuf + (size_t)c.rank * ocnt;
    hipLaunchKernelGGL(k_shard_obs, dim3(p.M), dim3(256), 0, st, D, lo, hi, B.obs_t, p.obs_y, p.rinv, B.m_own, B.S_own, mine, mine + (size_t)p.M * D);
    if (world > 1) SH_COMM(c, c.comm.all_gather(c.comm.user, mine, B.obuf, (uint64_t)ocnt, st));
    const int blocks = (int)(((size_t)p.M * D + 255) / 256 < 1024 ? ((size_t)p.M * D + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_shard_obs_sum, dim3(blocks), dim3(256), 0, st, D, p.M, world, B.obuf, p.obs_const, B.jm, B.scal);
  } else {
    LD_TRY(hipMemsetAsync(B.scal, 0, sizeof(double), st));
  }
  // E_sde terms of the own grid points; the small per-grid-point results go into this rank's part of the gather buffers
  double* gm_mine = B.gm_all + (size_t)c.rank * pad * D;
  double* et_mine = B.et_all + (size_t)c.rank * pad;
  if (n_own > 0)
    LD_TRY(lde_energy(D, n_own, p.theta, p.isg, a_own, b_own, B.m_own, B.S_own, et_mine, B.Ef_own, nullptr, gm_mine, B.gs_own, B.status,
                      B.lde_ws, B.lde_nb, st, nullptr, c.cs));      // (cs is idle here: the recursion ended joined, and lde_energy joins again)
  LD_TRY(mark(3));
  if (world > 1) {
    SH_COMM(c, group_begin(c));
    SH_COMM(c, c.comm.all_gather(c.comm.user, gm_mine, B.gm_all, (uint64_t)pad * D, st));
    SH_COMM(c, c.comm.all_gather(c.comm.user, et_mine, B.et_all, (uint64_t)pad, st));
    SH_COMM(c, group_end(c));
  }
  unpad(D, B.gm_all, B.gm_full);
  unpad(1, B.et_all, B.et_full);
  // dEsde_dS: every rank needs rows I_p of every grid point -- a time -> row exchange, not an all-gather; dEsde_dS of the own
  // grid points is dead behind it and Psi of the own grid points takes its place (one rank: no exchange, Psi in the other buffer)
  BwdIn in;
  in.gm = B.gm_full;
  in.obs_idx = obs_idx.data(); in.jm_sparse = B.jm; in.js_const = B.jsc;
  double* psi_own;
  if (world > 1) {
    LD_TRY(exchange_time_to_blocks(c, Np, 0, B.gs_own, B.g_rows, B.xs, B.xr, B.xT));
    in.g_rows = B.g_rows; in.g_rows_t = (size_t)c.Mp * D;
    psi_own = B.gs_own;
  } else {
    in.g_rows = B.gs_own; in.g_rows_t = DD;
    psi_own = B.g_rows;
  }
  LD_TRY(mark(4));
  LD_TRY(shard_solve_bwd(c, Np, av, in, B.lam_own, psi_own));
  LD_TRY(mark(5));
  if (n_own > 0)
    LD_TRY(lde_grad(D, n_own, c.dt, p.isg, a_own, b_own, B.m_own, B.S_own, B.lam_own, psi_own, B.Ef_own, gA_own, gB_own, B.lde_ws,
                    B.lde_nb, st));
  ReduceArgs r{};
  r.Np = Np; r.batch = 1; r.dt = c.dt; r.pre = 1.0; r.div = 1.0; r.e0 = p.e0;
  r.e_t = B.et_full; r.eobs = B.scal; r.esde = B.scal + 1; r.f = B.scal + 2;
  LD_TRY(launch_reduce(r, st));
  LD_TRY(mark(6));
  c.phases_valid = true;
  return hipSuccess;
}
#undef SH_COMM
#undef LD_TRY

}  // namespace ld
}  // namespace vgpa

// =================================================================================================================
//  C ABI (declared in include/vgpa_hip.h): stage-level entry points on DEVICE pointers, used by the host driver
//  vgpa_amd/large_d.py which interleaves them with the RCCL collectives of the row-sharded recursion.
// =================================================================================================================
using namespace vgpa;

extern "C" {

int vgpa_ld_gemm(void* stream, int transa, int M, int N, int K, const double* A0, const double* A1, int lda,
                 const double* B, int ldb, double* C, int cw) {
  if (M <= 0 || N <= 0 || K <= 0 || !A0 || !B || !C || cw <= 0 || (N % cw) != 0) return VGPA_ERR_ARG;
  ld::GemmArgs g{M, N, K, A0, A1, lda, B, ldb, C, cw};
  return ld::launch_gemm(transa != 0, g, (hipStream_t)stream) == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
}

int vgpa_ld_gemm_chunk(void* stream, int transa, int M, int N, int K, const double* A0, int lda, const double* B, int ldb,
                       double* C, int cw, int seg_tiles, int seg_stride, int accumulate) {
  if (M <= 0 || N <= 0 || K <= 0 || !A0 || !B || !C || cw <= 0 || (N % cw) != 0) return VGPA_ERR_ARG;
  if (seg_tiles < 0 || (seg_tiles > 0 && (seg_stride < seg_tiles * ld::BK || K % (seg_tiles * ld::BK) != 0))) return VGPA_ERR_ARG;
  ld::GemmArgs g{M, N, K, A0, nullptr, lda, B, ldb, C, cw};
  g.seg_tiles = seg_tiles; g.seg_stride = seg_stride; g.accumulate = accumulate != 0;
  return ld::launch_gemm(transa != 0, g, (hipStream_t)stream) == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
}

int vgpa_ld_stage(void* stream, const vgpa_ld_stage_args* p) {
  if (!p || p->D <= 0 || p->Mp <= 0 || p->cw <= 0) return VGPA_ERR_ARG;
  ld::StageArgs a{};
  a.D = p->D; a.row0 = p->row0; a.Mp = p->Mp; a.cw = p->cw; a.fwd = p->fwd; a.kstore = p->kstore; a.final = p->final_mode;
  a.mid_e = p->E1 != nullptr; a.has_j = p->J != nullptr; a.cx = p->cx; a.cf = p->cf;
  a.W = p->W; a.Wcol = p->Wcol; a.E0 = p->E0; a.E1 = p->E1; a.J = p->J; a.base = p->base; a.K1 = p->K1; a.K23 = p->K23;
  a.out = p->out; a.A0 = p->A0; a.A1 = p->A1; a.lda = p->lda; a.mid_a = p->A1 != nullptr; a.x = p->x;
  a.e0 = p->e0; a.e1 = p->e1; a.mid_ev = p->e1 != nullptr; a.jv = p->jv; a.vbase = p->vbase;
  a.k1v = p->k1v; a.k23v = p->k23v; a.vout = p->vout;
  return ld::launch_stage(a, (hipStream_t)stream) == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
}

// ---- row-sharded recursion behind the C ABI ------------------------------------------------------------------------
struct vgpa_shard {
  ld::ShardCtx c;
  int Np = 0, device = 0;
  double* ws = nullptr;
  bool own_stream = false;
  ld::SweepBuffers sb;                 // fused sweep only
  std::vector<void*> sweep_allocs;
  std::vector<int32_t> obs_idx;        // grid point -> observation number or -1 (host)
};

namespace {
// a rank that stops enqueueing leaves its peers inside a collective: whatever went wrong, the communicator goes down with it
int shard_fail(vgpa_shard* s, hipError_t e) {
  if (e == hipSuccess) return VGPA_OK;
  const bool comm = s->c.failed;
  ld::comm_failed(s->c);
  return comm ? VGPA_ERR_COMM : VGPA_ERR_DEVICE;
}
}  // namespace

int vgpa_shard_create(vgpa_shard** out, int method, double dt, int dim_d, int n_pts, int rank, int world, int device,
                      const vgpa_comm* comm, void* stream) {
  if (!out || dim_d < 1 || n_pts < 2 || world < 1 || rank < 0 || rank >= world || !(dt > 0.0)) return VGPA_ERR_ARG;
  if (method < VGPA_ODE_EULER || method > VGPA_ODE_RK4) return VGPA_ERR_ARG;
  if (dim_d % world != 0) return VGPA_ERR_ARG;
  if (world > 1 && (!comm || !comm->all_gather || !comm->all_to_all)) return VGPA_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return VGPA_ERR_DEVICE;
  vgpa_shard* s = new vgpa_shard();
  s->Np = n_pts; s->device = device;
  s->c.method = method; s->c.D = dim_d; s->c.rank = rank; s->c.world = world; s->c.dt = dt;
  s->c.Mp = dim_d / world; s->c.row0 = rank * s->c.Mp;
  if (comm) s->c.comm = *comm; else s->c.comm = vgpa_comm{};
  bool ok = true;
  if (stream) s->c.st = (hipStream_t)stream;
  else {
    ok = hipStreamCreateWithFlags(&s->c.st, hipStreamNonBlocking) == hipSuccess;
    s->own_stream = ok;
  }
  ok = ok && hipStreamCreateWithFlags(&s->c.cs, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&s->c.ev_stage, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&s->c.ev_tail, hipEventDisableTiming) == hipSuccess;
  for (int j = 0; ok && j < ld::kMaxChunks; j++) ok = hipEventCreateWithFlags(&s->c.ev_chunk[j], hipEventDisableTiming) == hipSuccess;
  for (int j = 0; ok && j < 7; j++) ok = hipEventCreate(&s->c.ev_phase[j]) == hipSuccess;
  const size_t n = ld::shard_workspace_doubles(dim_d, s->c.Mp, world);
  ok = ok && hipMalloc((void**)&s->ws, n * sizeof(double)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&s->c.pinned, ((size_t)world + 1) * sizeof(double), hipHostMallocDefault) == hipSuccess;
  if (!ok) { vgpa_shard_destroy(s); return VGPA_ERR_DEVICE; }
  for (int q = 0; q <= world; q++) s->c.pinned[q] = 0.0;
  (void)hipMemsetAsync(s->ws, 0, n * sizeof(double), s->c.st);
  s->c.w = ld::carve_shard(s->ws, dim_d, s->c.Mp);
  // default schedule: the pipelined gather with four sub-blocks wherever the table and the row-block size allow it
  // (VGPA_SHARD_CHUNKS=0 keeps round 2's serial schedule; vgpa_shard_set_option changes it per shard)
  int want = 4;
  if (const char* e = getenv("VGPA_SHARD_CHUNKS")) want = atoi(e);
  s->c.chunks = want > 0 ? ld::usable_chunks(s->c, want) : 0;
  *out = s;
  return VGPA_OK;
}

void vgpa_shard_destroy(vgpa_shard* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->c.st && !s->c.failed) (void)hipStreamSynchronize(s->c.st);
  if (s->c.cs && !s->c.failed) (void)hipStreamSynchronize(s->c.cs);
  if (s->ws) (void)hipFree(s->ws);
  for (void* q : s->sweep_allocs) (void)hipFree(q);
  if (s->c.pinned) (void)hipHostFree(s->c.pinned);    // (hipFree / hipHostFree wait for the device: no queued copy outlives its target)
  if (s->c.ev_stage) (void)hipEventDestroy(s->c.ev_stage);
  if (s->c.ev_tail) (void)hipEventDestroy(s->c.ev_tail);
  for (int j = 0; j < ld::kMaxChunks; j++) if (s->c.ev_chunk[j]) (void)hipEventDestroy(s->c.ev_chunk[j]);
  for (int j = 0; j < 7; j++) if (s->c.ev_phase[j]) (void)hipEventDestroy(s->c.ev_phase[j]);
  if (s->c.cs) (void)hipStreamDestroy(s->c.cs);
  if (s->own_stream) (void)hipStreamDestroy(s->c.st);
  delete s;
}

int vgpa_shard_set_option(vgpa_shard* s, int option, int64_t value) {
  if (!s) return VGPA_ERR_ARG;
  if (option == VGPA_SHARD_OPT_GATHER_CHUNKS) {
    if (value < 0 || value > ld::kMaxChunks) return VGPA_ERR_ARG;
    if (hipSetDevice(s->device) != hipSuccess || ld::shard_join(s->c) != hipSuccess) return VGPA_ERR_DEVICE;
    s->c.chunks = value > 0 ? ld::usable_chunks(s->c, (int)value) : 0;
    return VGPA_OK;
  }
  if (option == VGPA_SHARD_OPT_TIMEOUT_MS) { s->c.timeout_ms = value; return VGPA_OK; }
  return VGPA_ERR_ARG;
}

int vgpa_shard_get_option(const vgpa_shard* s, int option, int64_t* value) {
  if (!s || !value) return VGPA_ERR_ARG;
  if (option == VGPA_SHARD_OPT_GATHER_CHUNKS) { *value = s->c.chunks; return VGPA_OK; }
  if (option == VGPA_SHARD_OPT_TIMEOUT_MS) { *value = s->c.timeout_ms; return VGPA_OK; }
  return VGPA_ERR_ARG;
}

int vgpa_shard_time_slice(const vgpa_shard* s, int* t_lo, int* t_hi) {
  if (!s || !t_lo || !t_hi) return VGPA_ERR_ARG;
  ld::time_slice(s->Np, s->c.rank, s->c.world, t_lo, t_hi);
  return VGPA_OK;
}

// the ownership rule itself, without a shard (pure host arithmetic: callers size their buffers with it, tests check its balance)
int vgpa_time_slice(int n_pts, int rank, int world, int* t_lo, int* t_hi) {
  if (n_pts < 1 || world < 1 || rank < 0 || rank >= world || !t_lo || !t_hi) return VGPA_ERR_ARG;
  ld::time_slice(n_pts, rank, world, t_lo, t_hi);
  return VGPA_OK;
}

void* vgpa_shard_stream(vgpa_shard* s) { return s ? (void*)s->c.st : nullptr; }

int vgpa_shard_synchronize(vgpa_shard* s) {
  if (!s) return VGPA_ERR_ARG;
  if (hipSetDevice(s->device) != hipSuccess) return VGPA_ERR_DEVICE;
  if (s->c.failed) return VGPA_ERR_COMM;
  return ld::shard_wait(s->c);
}

int vgpa_shard_solve_fwd(vgpa_shard* s, const double* lin_a, const double* off_b, const double* m0, const double* s0,
                         const double* sigma, double* m_own, double* s_own) {
  if (!s || !lin_a || !off_b || !m0 || !s0 || !sigma || !m_own || !s_own) return VGPA_ERR_ARG;
  if (s->c.failed) return VGPA_ERR_COMM;
  if (hipSetDevice(s->device) != hipSuccess) return VGPA_ERR_DEVICE;
  return shard_fail(s, ld::shard_solve_fwd(s->c, s->Np, ld::aview_replicated(s->c, lin_a), off_b, m0, s0, sigma, m_own, s_own));
}

int vgpa_shard_solve_bwd(vgpa_shard* s, const double* lin_a, const double* desde_dm, const double* desde_ds,
                         const double* deobs_dm, const double* deobs_ds, double* lam_own, double* psi_own) {
  if (!s || !lin_a || !desde_dm || !desde_ds || !deobs_dm || !deobs_ds || !lam_own || !psi_own) return VGPA_ERR_ARG;
  if (s->c.failed) return VGPA_ERR_COMM;
  if (hipSetDevice(s->device) != hipSuccess) return VGPA_ERR_DEVICE;
  ld::BwdIn in;
  in.gm = desde_dm; in.g_rows = desde_ds + (size_t)s->c.row0 * s->c.D; in.g_rows_t = (size_t)s->c.D * s->c.D;
  in.jm_dense = deobs_dm; in.js_dense = deobs_ds;
  return shard_fail(s, ld::shard_solve_bwd(s->c, s->Np, ld::aview_replicated(s->c, lin_a), in, lam_own, psi_own));
}

namespace {
// every device buffer of the fused sweep; false when an allocation fails (the caller's agreement step tells the peers)
bool shard_sweep_buffers(vgpa_shard* s, int M, bool sharded_x) {
  const int D = s->c.D, Np = s->Np, world = s->c.world;
  const size_t DD = (size_t)D * D;
  int lo, hi;
  ld::time_slice(Np, s->c.rank, world, &lo, &hi);
  const size_t n_own = (size_t)(hi - lo > 0 ? hi - lo : 1), pad = (size_t)(Np + world - 1) / world;
  ld::SweepBuffers& B = s->sb;
  if (B.m_own && B.M == M && B.sharded_x == sharded_x) return true;
  for (void* q : s->sweep_allocs) (void)hipFree(q);
  s->sweep_allocs.clear();
  B = ld::SweepBuffers{};
  auto alloc = [&](double** ptr, size_t n) -> bool {
    if (hipMalloc((void**)ptr, (n ? n : 1) * sizeof(double)) != hipSuccess) return false;
    s->sweep_allocs.push_back(*ptr);
    return hipMemsetAsync(*ptr, 0, (n ? n : 1) * sizeof(double), s->c.st) == hipSuccess;
  };
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  B.lde_nb = ld::lde_batch(D, std::fmin(16.0e9, std::fmax(1.0e9, 0.05 * (double)free_b)));
  if (B.lde_nb > (int)n_own) B.lde_nb = (int)n_own;
  B.xT = ld::exchange_batch(D, std::fmin(2.0e9, std::fmax(2.5e8, 0.01 * (double)free_b)), (int)pad);
  const bool xch = world > 1;               // one rank: the exchanges are identities, no staging
  const size_t MD = (size_t)s->c.Mp * D;
  bool ok = alloc(&B.m_own, n_own * D) && alloc(&B.S_own, n_own * DD) && alloc(&B.lam_own, n_own * D) && alloc(&B.gs_own, n_own * DD) &&
            alloc(&B.Ef_own, n_own * D) && alloc(&B.g_rows, (size_t)Np * MD) && alloc(&B.gm_all, (size_t)world * pad * D) &&
            alloc(&B.gm_full, (size_t)Np * D) && alloc(&B.et_all, (size_t)world * pad) && alloc(&B.et_full, (size_t)Np) &&
            alloc(&B.obuf, (size_t)world * M * (D + 1)) && alloc(&B.jm, (size_t)M * D) && alloc(&B.jsc, DD) && alloc(&B.scal, 4) &&
            alloc(&B.lde_ws, ld::lde_workspace_doubles(D, B.lde_nb));
  if (xch) ok = ok && alloc(&B.xs, (size_t)B.xT * DD) && alloc(&B.xr, (size_t)B.xT * DD);
  if (sharded_x && xch)
    ok = ok && alloc(&B.a_rows, (size_t)Np * MD) && alloc(&B.a_cols, (size_t)Np * MD) && alloc(&B.b_all, (size_t)world * pad * D) &&
         alloc(&B.b_full, (size_t)Np * D);
  double* tmp = nullptr;
  ok = ok && alloc(&tmp, (size_t)M + 1);
  B.obs_t = reinterpret_cast<int64_t*>(tmp);
  tmp = nullptr;
  ok = ok && alloc(&tmp, 1);
  B.status = reinterpret_cast<int32_t*>(tmp);
  if (ok) { B.M = M; B.sharded_x = sharded_x; }
  return ok;
}

int shard_sweep_entry(vgpa_shard* s, const vgpa_shard_problem* p, const double* x_full, const double* a_own, const double* b_own,
                      double* f_host, double* grad_a_own, double* grad_b_own) {
  if (!p->inv_sigma_diag || !p->m0 || !p->s0 || !p->sigma || p->n_obs < 0) return VGPA_ERR_ARG;
  if (p->n_obs > 0 && (!p->obs_t || !p->obs_y || !p->obs_rinv_diag)) return VGPA_ERR_ARG;
  if (s->c.failed) return VGPA_ERR_COMM;
  if (hipSetDevice(s->device) != hipSuccess) return VGPA_ERR_DEVICE;
  const int D = s->c.D, Np = s->Np, M = p->n_obs;
  const size_t DD = (size_t)D * D;
  // argument checks come BEFORE the first collective and depend on replicated data only: every rank takes the same exit
  for (int n = 0; n < M; n++) {
    const int64_t t = p->obs_t[n];
    if (t < 0 || t >= Np) return VGPA_ERR_ARG;
    if (n > 0 && t <= p->obs_t[n - 1]) return VGPA_ERR_ARG;       // increasing: one observation per grid point (the backward
  }                                                               // jumps index by grid point, the energy sums every observation)
  const bool first = !(s->sb.m_own && s->sb.M == M && s->sb.sharded_x == (x_full == nullptr));
  const bool have = shard_sweep_buffers(s, M, x_full == nullptr);
  ld::SweepBuffers& B = s->sb;
  if (first || !have) {      // allocation is a local event: agree on it before any rank enters the sweep's collectives
    const int rc = ld::shard_agree(s->c, nullptr, have ? VGPA_OK : VGPA_ERR_DEVICE);
    if (rc != VGPA_OK) {       // EVERY rank forgets its buffers' key, so that `first` agrees on all of them at the next call
      if (!have) s->sb = ld::SweepBuffers{}; else s->sb.M = -1;
      return rc;
    }
  }
  s->obs_idx.assign((size_t)Np, -1);
  for (int n = 0; n < M; n++) s->obs_idx[(size_t)p->obs_t[n]] = n;
  bool ok = true;
  if (M > 0) {
    ok = hipMemcpyAsync(B.obs_t, p->obs_t, sizeof(int64_t) * M, hipMemcpyHostToDevice, s->c.st) == hipSuccess &&
         hipMemsetAsync(B.jsc, 0, sizeof(double) * DD, s->c.st) == hipSuccess;
    if (ok) hipLaunchKernelGGL(ld::k_diag_half, dim3((D + 255) / 256), dim3(256), 0, s->c.st, D, p->obs_rinv_diag, B.jsc);
  }
  ld::SweepProblem sp{p->theta, p->obs_const, p->e0, p->inv_sigma_diag, p->m0, p->s0, p->sigma, p->obs_y, p->obs_rinv_diag, M, p->obs_t};
  if (!ok || ld::shard_sweep(s->c, Np, B, sp, x_full, a_own, b_own, s->obs_idx, grad_a_own, grad_b_own) != hipSuccess) {
    const bool was_comm = s->c.failed;
    ld::comm_failed(s->c);                   // this rank stops enqueueing: its peers must not wait for it
    return was_comm ? VGPA_ERR_COMM : VGPA_ERR_DEVICE;
  }
  // the outcome is collective: a covariance that lost positive definiteness on ONE rank's time slice (or a device fault there)
  // is every rank's return code (the reference raises LinAlgError and the whole run ends, variational.py:380); F rides to the
  // host with the status words, through the shard's page-locked staging
  return ld::shard_agree(s->c, B.status, VGPA_OK, B.scal + 2, f_host);
}
}  // namespace

int vgpa_shard_sweep(vgpa_shard* s, const vgpa_shard_problem* p, const double* x_dev, double* f_host, double* grad_a_own,
                     double* grad_b_own) {
  if (!s || !p || !x_dev || !f_host || !grad_a_own || !grad_b_own) return VGPA_ERR_ARG;
  return shard_sweep_entry(s, p, x_dev, nullptr, nullptr, f_host, grad_a_own, grad_b_own);
}

int vgpa_shard_sweep_sharded(vgpa_shard* s, const vgpa_shard_problem* p, const double* a_own, const double* b_own, double* f_host,
                             double* grad_a_own, double* grad_b_own) {
  if (!s || !p || !a_own || !b_own || !f_host || !grad_a_own || !grad_b_own) return VGPA_ERR_ARG;
  return shard_sweep_entry(s, p, nullptr, a_own, b_own, f_host, grad_a_own, grad_b_own);
}

// The two per-stage collectives alone, as the recursion issues them (same buffers, sizes, streams and schedule), `reps` times:
// average milliseconds of the all-to-all of the product blocks and of the gather of the next stage state.
int vgpa_shard_time_collectives(vgpa_shard* s, int reps, double* all_to_all_ms, double* gather_ms) {
  if (!s || reps < 1 || !all_to_all_ms || !gather_ms) return VGPA_ERR_ARG;
  *all_to_all_ms = *gather_ms = 0.0;
  if (s->c.world < 2) return VGPA_OK;
  if (s->c.failed) return VGPA_ERR_COMM;
  if (hipSetDevice(s->device) != hipSuccess) return VGPA_ERR_DEVICE;
  ld::ShardCtx& c = s->c;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return VGPA_ERR_DEVICE;
  int rc = VGPA_OK;
  float ms = 0.f;
  for (int which = 0; which < 2 && rc == VGPA_OK; which++) {
    for (int it = -2; it < reps && rc == VGPA_OK; it++) {       // two untimed rounds first
      if (it == 0) (void)hipEventRecord(e0, c.st);
      if (which == 0) {
        if (c.comm.all_to_all(c.comm.user, c.w.Wp, c.w.Wcol, (uint64_t)c.Mp * c.Mp, c.st) != 0) rc = VGPA_ERR_COMM;
      } else {
        if (ld::shard_gather(c, c.w.XA, c.w.xvA) != hipSuccess || ld::shard_join(c) != hipSuccess) rc = VGPA_ERR_COMM;
      }
    }
    if (rc != VGPA_OK) break;
    (void)hipEventRecord(e1, c.st);
    rc = ld::shard_wait(c);
    if (rc == VGPA_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) (which == 0 ? *all_to_all_ms : *gather_ms) = ms / reps;
  }
  if (rc == VGPA_ERR_COMM) ld::comm_failed(c);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

// Milliseconds of the six phases of the LAST fused sweep on this rank (HIP events on the shard's stream): [0] exchanges that bring
// a memory-sharded x to the recursions, [1] forward recursion, [2] observation + E_sde terms of the own grid points, [3] gathers
// of the small terms + the time -> row exchange of dEsde_dS, [4] backward recursion, [5] gradient of the own grid points + F.
int vgpa_shard_phase_ms(vgpa_shard* s, double* ms6) {
  if (!s || !ms6) return VGPA_ERR_ARG;
  if (!s->c.phases_valid) return VGPA_ERR_STATE;
  if (hipSetDevice(s->device) != hipSuccess) return VGPA_ERR_DEVICE;
  if (hipEventSynchronize(s->c.ev_phase[6]) != hipSuccess) return VGPA_ERR_DEVICE;
  for (int i = 0; i < 6; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->c.ev_phase[i], s->c.ev_phase[i + 1]) != hipSuccess) return VGPA_ERR_DEVICE;
    ms6[i] = ms;
  }
  return VGPA_OK;
}

// ONE stage of the row-sharded recursion as the forward RK4 loop issues it -- K-chunk products (waiting for the sub-blocks of the
// previous stage's gather), all-to-all, stage kernel, gather of the next stage state -- `reps` times back to back on the shard's
// own workspace, stage i reading what stage i - 1 completed; with_collectives == 0 leaves the collectives out (same kernels, same
// launches), so that (with) - (without) is the communication a stage does NOT hide.  Timing only: the workspace is overwritten.
// Collective call when with_collectives != 0.
int vgpa_shard_time_stage(vgpa_shard* s, int reps, int with_collectives, double* ms_per_stage) {
  if (!s || reps < 1 || !ms_per_stage) return VGPA_ERR_ARG;
  *ms_per_stage = 0.0;
  if (s->c.failed) return VGPA_ERR_COMM;
  if (hipSetDevice(s->device) != hipSuccess) return VGPA_ERR_DEVICE;
  ld::ShardCtx& c = s->c;
  ld::ShardWork& w = c.w;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return VGPA_ERR_DEVICE;
  const size_t r0 = (size_t)c.row0 * c.D;
  int rc = ld::shard_join(c) == hipSuccess ? VGPA_OK : VGPA_ERR_DEVICE;
  c.skip_comm = !with_collectives;
  c.mid_a0 = c.mid_a1 = nullptr;
  for (int it = -2; it < reps && rc == VGPA_OK; it++) {
    if (it == 0) (void)hipEventRecord(e0, c.st);
    ld::ShardStage st{};
    st.fwd = true;
    st.Ag0 = w.cur + r0; st.Ag1 = nullptr; st.ldag = c.D; st.Av0 = w.cur + r0; st.Av1 = nullptr; st.ldav = c.D;
    st.X = (it & 1) ? w.XB : w.XA; st.xv = (it & 1) ? w.xvB : w.xvA;
    st.E0 = w.cur + r0; st.e0 = w.vcur + c.row0;
    st.base = w.cur; st.vbase = w.vcur;
    st.out = (it & 1) ? w.XA : w.XB; st.vout = (it & 1) ? w.xvA : w.xvB;
    st.kstore = 1; st.final_mode = 0; st.cx = 0.0; st.cf = 0.0;          // out = base + 0 * slope: the data stay what they are
    if (ld::shard_stage(c, st) != hipSuccess) rc = c.failed ? VGPA_ERR_COMM : VGPA_ERR_DEVICE;
  }
  c.skip_comm = false;
  if (rc == VGPA_OK && ld::shard_join(c) != hipSuccess) rc = VGPA_ERR_DEVICE;
  if (rc == VGPA_OK) {
    (void)hipEventRecord(e1, c.st);
    rc = ld::shard_wait(c);
    float ms = 0.f;
    if (rc == VGPA_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *ms_per_stage = ms / reps;
  }
  if (rc != VGPA_OK) ld::comm_failed(c);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

}  // extern "C"
