// The one-kernel Runge-Kutta stages above D = 64 (k_stage_prod, k_stage_wide; large_d.hip::run_stage chooses).  Their own translation
// unit: compiled with -mllvm -amdgpu-mfma-vgpr-form (vgpa_amd/build.py).  Left to its heuristics the compiler keeps the two fp64-MFMA
// accumulators of these kernels in AGPRs inside the k loop and in VGPRs around it -- 16 v_accvgpr_write at the top and 16 v_accvgpr_read at
// the bottom of every trip, each waiting for the matrix pipe to drain -- which costs 4-5 % where the loop has few trips (fused sweep,
// D = 128: 31.1 -> 29.7 ms; ragged D = 1000: 54.2 -> 52.0 ms); the stage products (k_gemm_v, 128-row tiles) lose 1.5 % with the same
// flag, hence the split.
#include <cstdlib>

#include "large_d_stage.h"

namespace vgpa {
namespace ld {

// ---------------------------------------------------------------------------------------------------------------
// One kernel per stage (64 < D <= 512, one rank owns every row, symmetric inputs).  A stage of the two-kernel
// scheme above costs ~6 us between dependent launches whatever their size (measured: 3600 launches of an RK4 recursion over
// 401 grid points take 24 ms at D = 96 AND at D = 128, 32 ms at D = 256; replaying the loop as a hipGraph changes nothing --
// the gap is the device's, not the host's -- EXPERIMENTS.md s.13), so below D ~ 512 the recursion is bound by the NUMBER of
// launches: nine per RK4 step (mid-point operand, four products, four stage kernels).  Here a workgroup owns the pair of
// 32 x 32 tiles (I, J) / (J, I), I <= J, of the stage slope and forms BOTH products it needs itself,
//     forward :  w = (A X)[I, J],      wt = (A X)[J, I]^T   = (X A^T)[I, J]      (X = X^T)
//     backward:  w = (A^T Psi)[I, J],  wt = (A^T Psi)[J, I]^T = (Psi A)[I, J]    (Psi = Psi^T)
// -- two fp64-MFMA accumulators over the same k loop, the same k order as k_gemm, then the element-wise stage of k_stage_sym
// on them in the same expression order (results equal the two-kernel scheme's bit for bit), the mirror tile through an LDS
// transposition.  The mid-point operand 0.5 (A_k + A_{k+1}) is averaged while staging (A is L2-resident at these sizes).
// Four launches per RK4 step instead of nine; every tile product is still computed exactly once per stage.
// By symmetry of X every operand tile is a set of ROW segments: forward all four tiles are [32 rows][16 k] (rows I / J of A
// and of X); backward the A tiles are [16 k][32 columns I / J] and the Psi tiles [32 rows][16 k].
// register sets of operand loads in flight in k_stage_prod (fused sweep, same box, 2 / 4 / 6 sets: D = 72 26.5 / 31.5 / 29.4 ms,
// 128 31.2 / 32.3 / 40.9, 200 45.9 / 50.4 / 55.4, 256 50.7 / 51.2 / 56.1: the tiles a deeper pipeline multiplies beyond the last one
// cost more than its depth hides)
#ifndef VGPA_PROD_PF
#define VGPA_PROD_PF 2
#endif
// Operand tiles come through buffer descriptors (raw_buffer_load: a per-thread 32-bit offset, advanced by one k-tile after every
// load): no 64-bit address arithmetic per tile -- with addresses recomputed per tile the register allocator recycles the destination
// registers of the loads in flight for them and every prefetch waits for the previous one (seen in the ISA of the first version of
// this kernel: 17 us per launch at D = 128) -- and the range check of the descriptor returns zero for rows / columns / k beyond the
// matrix (an offset beyond the D x D doubles), so that no load sits in a branch region either.  Only a row-major tile's k overrun
// into the NEXT ROW needs a select.  (The k-tile offset must travel in the vector offset: the scalar offset operand of a buffer
// instruction is not range-checked.)
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return __builtin_bit_cast(double, v);
}
template <bool TRANSA, bool MID>
__global__ void __launch_bounds__(NT) k_stage_prod(StageArgs a) {
  stage_batch_offsets(a);
  const int D = a.D;
  const int nt = (D + TS - 1) / TS;
  const int npair = nt * (nt + 1) / 2;
  if ((int)blockIdx.x >= npair) { stage_vector_rows(a, (int)blockIdx.x - npair); return; }
  constexpr int LDR = 18;                       // [32 rows][16 k + 2]: conflict-free fragment reads (k_gemm_v, NN)
  constexpr int LDT = 48;                       // [16 k][32 + 16]: = 16 (mod 32)
  constexpr int ASZ = TRANSA ? BK * LDT : TS * LDR;
  __shared__ double sA[2][2][ASZ];              // [buffer][I | J]
  __shared__ double sX[2][2][TS * LDR];
  __shared__ double tc[TS][TS + 1];
  int by = 0, rem = (int)blockIdx.x;
  while (rem >= nt - by) { rem -= nt - by; by++; }
  const int bx = by + rem;
  const bool diag = (by == bx);
  const int I0 = by * TS, J0 = bx * TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, fk = lane >> 4;

  const unsigned mat_bytes = (unsigned)((size_t)D * D * sizeof(double));
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.Xm), 0, mat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.M0), 0, mat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(MID ? a.M1 : a.M0), 0, mat_bytes, 0x00020000);
  // per-thread element of every operand tile: row-major tiles (r = (tid >> 4) + 16 q, k = tid & 15), k-major A tiles of the
  // backward product (k = (tid >> 5) + 8 q, column i = tid & 31); byte offsets without the k-tile part, beyond the matrix when
  // the row / column is
  constexpr unsigned kBeyond = 0x7ff00000u;
  const int kx = tid & 15;
  unsigned ox[2][2], oa[2][2];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int R0 = h ? J0 : I0;
      const int r = R0 + (tid >> 4) + 16 * q;
      ox[h][q] = r < D ? (unsigned)(r * D + kx) * 8u : kBeyond;
      if (TRANSA) { const int i = R0 + (tid & 31); oa[h][q] = i < D ? (unsigned)(((tid >> 5) + 8 * q) * D + i) * 8u : kBeyond; }
      else oa[h][q] = ox[h][q];
    }
  const unsigned step_x = BK * 8u, step_a = TRANSA ? (unsigned)D * BK * 8u : step_x;
  // two register sets (prefetch distance two) of RAW loads: the mid-point average and the k-edge select happen when a set moves to
  // LDS, one k-tile later -- next to the loads they would wait for them on the spot
  struct Regs { double a[2][2], b[2][2], x[2][2]; };          // [I | J][q]
  constexpr int PF = VGPA_PROD_PF;
  Regs rs[PF];
  auto load_tiles = [&](Regs& r) {            // the NEXT k-tile (tiles are requested in order)
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        r.x[h][q] = buf_load(rX, (int)ox[h][q], 0);
        r.a[h][q] = buf_load(rM0, (int)oa[h][q], 0);
        if (MID) r.b[h][q] = buf_load(rM1, (int)oa[h][q], 0);
        ox[h][q] += step_x; oa[h][q] += step_a;
      }
  };
  auto store_tiles = [&](int buf, int k0, const Regs& r) {
    const bool kxok = k0 + kx < D;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const double v = MID ? 0.5 * (r.a[h][q] + r.b[h][q]) : r.a[h][q];
        sX[buf][h][((tid >> 4) + 16 * q) * LDR + (tid & 15)] = kxok ? r.x[h][q] : 0.0;
        if (TRANSA) sA[buf][h][((tid >> 5) + 8 * q) * LDT + (tid & 31)] = v;
        else sA[buf][h][((tid >> 4) + 16 * q) * LDR + (tid & 15)] = kxok ? v : 0.0;
      }
  };
  d4 w1 = d4{0.0, 0.0, 0.0, 0.0}, w2 = d4{0.0, 0.0, 0.0, 0.0};
  auto compute = [&](int cur) {
    // MFMA operands: A-side lane (i = fi, k = fk), B-side lane (k = fk, j = fi)
    frag_ptr aI = frag(TRANSA ? sA[cur][0] + fk * LDT + 16 * wm + fi : sA[cur][0] + (16 * wm + fi) * LDR + fk);
    frag_ptr aJ = frag(TRANSA ? sA[cur][1] + fk * LDT + 16 * wn + fi : sA[cur][1] + (16 * wn + fi) * LDR + fk);
    frag_ptr xI = frag(sX[cur][0] + (16 * wm + fi) * LDR + fk);
    frag_ptr xJ = frag(sX[cur][1] + (16 * wn + fi) * LDR + fk);
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      const int ko = TRANSA ? kk * 4 * LDT : kk * 4;
      const double vaI = aI[ko], vaJ = aJ[ko], vxI = xI[kk * 4], vxJ = xJ[kk * 4];
      w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(vaI, vxJ, w1, 0, 0, 0);       // (A X)[I, J]   | (A^T Psi)[I, J]
      w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(vxI, vaJ, w2, 0, 0, 0);       // (X A^T)[I, J] | (Psi A)[I, J]
    }
  };
  const int nk = (D + BK - 1) / BK;
#pragma unroll
  for (int u = 0; u < PF; u++) load_tiles(rs[u]);              // tiles 0 .. PF - 1
  store_tiles(0, 0, rs[0]);
  // operands of the element-wise stage: requested before the k loop, consumed behind it (C col = lane & 15, row = (lane >> 4) + 4 r).
  // Absent operands (no mid-point partner, no jump) read a valid address and are dropped by a select: no branch around a load.
  size_t eo[4]; bool eok[4];
  double e0[4], e1[4], bs[4], jp[4];
  const double* E1 = a.mid_e ? a.E1 : a.E0;
  const double* Jp = (a.final && a.has_j) ? a.J : a.base;
#pragma unroll
  for (int r4 = 0; r4 < 4; r4++) {
    const int r = I0 + 16 * wm + (lane >> 4) + 4 * r4, j = J0 + 16 * wn + (lane & 15);
    eok[r4] = r < D && j < D;
    eo[r4] = (size_t)min(r, D - 1) * D + min(j, D - 1);
    e0[r4] = a.E0[eo[r4]]; e1[r4] = E1[eo[r4]]; bs[r4] = a.base[eo[r4]]; jp[r4] = Jp[eo[r4]];
  }
  __syncthreads();
  static_assert(PF % 2 == 0, "LDS buffer parity is static in the unrolled loop");
  // step kt: the register set that held tile kt (in LDS since the previous step) takes tile kt + PF; tile kt is multiplied; tile
  // kt + 1 moves to the other LDS buffer.  No branch inside: loads, LDS stores and products of tiles beyond the last one (up to
  // PF - 1 of them) are issued like any other -- they are zeros (range check, k-edge select) -- because a load under a branch, or
  // an exit from the unrolled body, leaves the wait counts of the paths to be merged to the most conservative one: the LDS store
  // of one register set then waits for the loads just issued into another.
  for (int kt = 0; kt < nk; kt += PF) {
#pragma unroll
    for (int u = 0; u < PF; u++) {
      load_tiles(rs[u]);
      compute(u & 1);
      store_tiles((u + 1) & 1, (kt + u + 1) * BK, rs[(u + 1) % PF]);
      __syncthreads();
    }
  }

  // the Runge-Kutta slots: written by the previous stage kernels of this step (the same thread, the same element)
  const double sgn = a.fwd ? 1.0 : -1.0;
  double k1[4], k23[4];
#pragma unroll
  for (int r4 = 0; r4 < 4; r4++) { k1[r4] = a.K1[eo[r4]]; k23[r4] = a.K23[eo[r4]]; }
#pragma unroll
  for (int r4 = 0; r4 < 4; r4++) {
    const int rr = 16 * wm + (lane >> 4) + 4 * r4, cc = 16 * wn + (lane & 15);
    const double w = w1[r4], wt = w2[r4];
    const double e = a.mid_e ? 0.5 * (e1[r4] + e0[r4]) : e0[r4];
    const double rv = a.fwd ? ((-w - wt) + e) : ((-e + wt) + w);
    const double jump = (a.final && a.has_j) ? jp[r4] : 0.0;
    const double res = stage_combine(rv, bs[r4], a.final >= 2 ? k1[r4] : 0.0, a.final == 3 ? k23[r4] : 0.0, a.final, a.cx, a.cf, sgn, jump);
    if (eok[r4]) {
      if (a.kstore == 1) a.K1[eo[r4]] = rv;
      else if (a.kstore == 2) a.K23[eo[r4]] = rv;
      else if (a.kstore == 3) a.K23[eo[r4]] = k23[r4] + rv;
      if (!diag || rr <= cc) a.out[eo[r4]] = res;
    }
    tc[rr][cc] = res;
  }
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
  for (int q = 0; q < 4; q++) {                          // the mirror tile (diagonal pair: its strictly lower part)
    const int rr = ty + 8 * q;
    const int r2 = J0 + rr, j2 = I0 + tx;
    if (r2 < D && j2 < D && (!diag || rr > tx)) a.out[(size_t)r2 * D + j2] = tc[tx][rr];
  }
}

// The same stage for launches of many tile pairs (D even; run_stage holds the measured selection rule).  Differences from
// k_stage_prod:
//   * equal work per workgroup: a DIAGONAL tile needs one product only (wt = w^T), so a diagonal workgroup takes TWO diagonal
//     tiles, one per accumulator -- D = 1024: 496 + 16 = 512 equal workgroups, two per CU -- and transposes its accumulators
//     through LDS (tile a: w = acc1, wt = acc1^T; tile b: wt = acc2, w = acc2^T: the operand slots keep their roles, only their
//     row blocks change: (I, J, I, J) -> (Ia, Ia, Ib, Ib));
//   * 16-byte loads and LDS stores (one per thread and operand tile), the mid-point operand formed once per step by k_mid;
//   * the element-wise operands are read behind the k loop (37 / 43 KB of LDS -- the transposition buffers alias the operand
//     tiles -- three workgroups per CU).
// Where the time goes at D = 1024 (profiles/r05d_stage_{two-kernel,wide}_D1024_*.csv): 55 us per launch
// against 40 + 12 us of GEMM + k_stage_sym (+ 3 us more of launch gaps for those): the k loop takes ~50 us, 64 k-tiles of 8 MFMAs
// per wave at two waves per SIMD = 27 us of MFMA time at 2.4 GHz, matrix pipe busy 0.58 / 0.59 of the SIMD-cycles (GEMM: 0.77) --
// 16 instead of 12 fragment reads per 8 MFMAs, four instead of three 16-byte loads and LDS stores per k-tile.  At THIS size (512
// workgroups, two per CU, one round) neither prefetch depth (2 / 4 / 6 register sets), nor the order of LDS stores and products,
// nor the XCD-aware tile order moved it by more than 1 %; away from it the latter two pay (below).
#ifndef VGPA_WIDE_PF
#define VGPA_WIDE_PF 4
#endif
#ifndef VGPA_WIDE_XCD
#define VGPA_WIDE_XCD 1
#endif
typedef double d2v __attribute__((ext_vector_type(2)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d2v buf_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return __builtin_bit_cast(d2v, v);
}
template <bool TRANSA>
__global__ void __launch_bounds__(NT) k_stage_wide(StageArgs a) {
  stage_batch_offsets(a);
  const int D = a.D;
  const int nt = (D + TS - 1) / TS;
  const int noff = nt * (nt - 1) / 2, ndg = (nt + 1) / 2;
#if VGPA_WIDE_XCD
  const int chunk = (noff + ndg + 7) / 8;                      // matrix workgroups per XCD
  if ((int)blockIdx.x >= 8 * chunk) { stage_vector_rows(a, (int)blockIdx.x - 8 * chunk); return; }
  // workgroup -> tiles.  Workgroup b runs on XCD b % 8, each with its own L2: XCD x takes the x-th CONTIGUOUS eighth of a list in
  // which neighbours share operand strips -- macro-rows of H = 8 tile rows, inside a macro-row its diagonal workgroups, the
  // triangular part of its diagonal block, then whole columns of H pairs -- so that the workgroups an XCD runs at one time cover
  // about an 8 x 8 block of pairs: 16 row strips of A and of X through its 4 MB L2 instead of strips of the whole matrix
  // (against the row-by-row order: D = 1536 78.4 -> 74.8 ms per fused sweep, 384 36.2 -> 35.4; nothing at D = 1024).
  constexpr int H = 8;
  int idx = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (idx >= noff + ndg) return;
  int mr = 0, hm = 0, nd = 0;                                  // first tile row of the macro-row, its rows, its diagonal workgroups
  for (;; mr += H) {
    hm = min(H, nt - mr);
    nd = (hm + 1) / 2;
    const int np = hm * (nt - 1) - mr * hm - hm * (hm - 1) / 2;
    if (idx < nd + np) break;
    idx -= nd + np;
  }
  const bool pair = idx >= nd;
  int Ra, Rb;                                                  // pair: I0, J0 (I < J); diagonal: Ia0, Ib0
  bool two = true;
  if (!pair) {
    const int dg = mr / 2 + idx;
    Ra = 2 * dg * TS; two = 2 * dg + 1 < nt; Rb = two ? Ra + TS : Ra;
  } else {
    idx -= nd;
    int by, bx;
    const int tri = hm * (hm - 1) / 2;
    if (idx < tri) {
      int c = 1;
      while (idx >= c) { idx -= c; c++; }
      by = mr + idx; bx = mr + c;
    } else {
      idx -= tri;
      bx = mr + hm + idx / hm; by = mr + idx % hm;
    }
    Ra = by * TS; Rb = bx * TS;
  }
#else
  if ((int)blockIdx.x >= noff + ndg) { stage_vector_rows(a, (int)blockIdx.x - noff - ndg); return; }
  const bool pair = (int)blockIdx.x < noff;
  int Ra, Rb;                                                  // pair: I0, J0 (I < J); diagonal: Ia0, Ib0
  bool two = true;
  if (pair) {
    int by = 0, rem = (int)blockIdx.x;
    while (rem >= nt - 1 - by) { rem -= nt - 1 - by; by++; }
    Ra = by * TS; Rb = (by + 1 + rem) * TS;
  } else {
    const int dg = (int)blockIdx.x - noff;
    Ra = 2 * dg * TS; two = 2 * dg + 1 < nt; Rb = two ? Ra + TS : Ra;
  }
#endif
  constexpr int LDR = 18, LDT = 48;
  constexpr int SZR = TS * LDR, SZK = BK * LDT;
  constexpr int SZ0 = TRANSA ? SZK : SZR;                    // slots 0 and 3 (tiles of A); slots 1 and 2 (tiles of X) are row-major
  constexpr int BUF = 2 * SZ0 + 2 * SZR;
  __shared__ __attribute__((aligned(16))) double lds[2 * BUF];
  constexpr int slot_off[4] = {0, SZ0, SZ0 + SZR, SZ0 + 2 * SZR};
  static_assert(2 * BUF >= 2 * TS * (TS + 1), "the transposition buffers alias the operand tiles");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, fk = lane >> 4;
  const int R_[4] = {Ra, pair ? Rb : Ra, pair ? Ra : Rb, Rb};  // row (column) blocks of the operand slots

  const unsigned mat_bytes = (unsigned)((size_t)D * D * sizeof(double));
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.Xm), 0, mat_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.M0), 0, mat_bytes, 0x00020000);
  constexpr unsigned kBeyond = 0x7ff00000u;
  // 16-byte element of every operand tile: row-major [32 rows][8 pairs of k] (row = tid >> 3), k-major [16 k][16 pairs of columns]
  const int kp2 = 2 * (tid & 7);
  unsigned vo[4]; int so_[4];                                  // per-thread byte offsets (advanced per k-tile), LDS offsets (doubles)
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const bool km = TRANSA && (s == 0 || s == 3);
    if (km) { const int c = R_[s] + 2 * (tid & 15); vo[s] = c < D ? (unsigned)((tid >> 4) * D + c) * 8u : kBeyond; so_[s] = slot_off[s] + (tid >> 4) * LDT + 2 * (tid & 15); }
    else { const int r = R_[s] + (tid >> 3); vo[s] = r < D ? (unsigned)(r * D + kp2) * 8u : kBeyond; so_[s] = slot_off[s] + (tid >> 3) * LDR + kp2; }
  }
  const unsigned step_rm = BK * 8u, step_km = TRANSA ? (unsigned)D * BK * 8u : step_rm;
  // PF register sets of loads in flight (fused sweep, same box, PF = 2 -> 4: D = 384 37.4 -> 35.4 ms, 640 30.6 -> 28.9, 1000
  // 55.1 -> 54.5, 1536 80.1 -> 74.8; at D = 1024, exactly two workgroups per CU, within 1 %: tools/ab_wide_variants.sh)
  constexpr int PF = VGPA_WIDE_PF;
  struct Regs { d2v v[4]; };
  Regs rs[PF];
  auto load_tiles = [&](Regs& r) {                             // the NEXT k-tile
    r.v[0] = buf_load2(rM, (int)vo[0], 0);
    r.v[1] = buf_load2(rX, (int)vo[1], 0);
    r.v[2] = buf_load2(rX, (int)vo[2], 0);
    r.v[3] = buf_load2(rM, (int)vo[3], 0);
    vo[0] += step_km; vo[1] += step_rm; vo[2] += step_rm; vo[3] += step_km;
  };
  auto store_tiles = [&](int buf, int k0, const Regs& r) {
    const bool kok = k0 + kp2 < D;                             // (row-major tiles: a k beyond the matrix is the next row's data)
    const d2v z = d2v{0.0, 0.0};
    double* b = lds + buf * BUF;
    *reinterpret_cast<d2v*>(b + so_[0]) = (TRANSA || kok) ? r.v[0] : z;
    *reinterpret_cast<d2v*>(b + so_[1]) = kok ? r.v[1] : z;
    *reinterpret_cast<d2v*>(b + so_[2]) = kok ? r.v[2] : z;
    *reinterpret_cast<d2v*>(b + so_[3]) = (TRANSA || kok) ? r.v[3] : z;
  };
  d4 w1 = d4{0.0, 0.0, 0.0, 0.0}, w2 = d4{0.0, 0.0, 0.0, 0.0};
  // MFMA operands: A-side lane (i = fi, k = fk), B-side lane (k = fk, j = fi)
  const int f0 = slot_off[0] + (TRANSA ? fk * LDT + 16 * wm + fi : (16 * wm + fi) * LDR + fk);
  const int f3 = slot_off[3] + (TRANSA ? fk * LDT + 16 * wn + fi : (16 * wn + fi) * LDR + fk);
  const int f1 = slot_off[1] + (16 * wn + fi) * LDR + fk, f2 = slot_off[2] + (16 * wm + fi) * LDR + fk;
  auto compute = [&](int cur) {
    frag_ptr b = frag(lds + cur * BUF);
#pragma unroll
    for (int kk = 0; kk < BK / 4; kk++) {
      const int ko = TRANSA ? kk * 4 * LDT : kk * 4;
      const double v0 = b[f0 + ko], v1 = b[f1 + kk * 4], v2 = b[f2 + kk * 4], v3 = b[f3 + ko];
      w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v1, w1, 0, 0, 0);         // (A X)[I, J]   | (A^T Psi)[I, J]      (diagonal: tile a)
      w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(v2, v3, w2, 0, 0, 0);         // (X A^T)[I, J] | (Psi A)[I, J]        (diagonal: tile b)
    }
  };
  const int nk = (D + BK - 1) / BK;
  static_assert(PF % 2 == 0, "LDS buffer parity is static in the unrolled loop");
#pragma unroll
  for (int u = 0; u < PF; u++) load_tiles(rs[u]);              // tiles 0 .. PF - 1
  store_tiles(0, 0, rs[0]);
  __syncthreads();
  // step kt: the set that held tile kt (in LDS since the previous step) takes tile kt + PF; tile kt is multiplied; tile kt + 1 moves
  // to the other LDS buffer.  No branch inside: tiles beyond the last one (up to PF - 1 of them when PF does not divide the tile
  // count) are loaded, stored and multiplied like any other -- they are zeros (range check, k-edge select) -- because every exit
  // from the unrolled body makes the wait counts of its paths merge to the most conservative one.
  for (int kt = 0; kt < nk; kt += PF) {
#pragma unroll
    for (int u = 0; u < PF; u++) {
      load_tiles(rs[u]);
      compute(u & 1);
      store_tiles((u + 1) & 1, (kt + u + 1) * BK, rs[(u + 1) % PF]);
      __syncthreads();
    }
  }
  __syncthreads();                                             // (the transposition buffers alias the operand tiles)
  double (*tcA)[TS + 1] = reinterpret_cast<double (*)[TS + 1]>(lds);
  double (*tcB)[TS + 1] = reinterpret_cast<double (*)[TS + 1]>(lds + TS * (TS + 1));

  const double sgn = a.fwd ? 1.0 : -1.0;
  // one element of the stage: slope, slots, the new state.  Returns the state; `direct` also stores it at (r, j).
  auto element = [&](int r, int j, double w, double wt, bool direct) {
    const size_t o = (size_t)r * D + j;
    const double e = a.mid_e ? 0.5 * (a.E1[o] + a.E0[o]) : a.E0[o];
    const double rv = a.fwd ? ((-w - wt) + e) : ((-e + wt) + w);
    const double k1 = (a.final >= 2) ? a.K1[o] : 0.0;
    const double k23 = (a.final == 3) ? a.K23[o] : 0.0;
    if (a.kstore == 1) a.K1[o] = rv;
    else if (a.kstore == 2) a.K23[o] = rv;
    else if (a.kstore == 3) a.K23[o] = a.K23[o] + rv;
    const double jump = (a.final && a.has_j) ? a.J[o] : 0.0;
    const double res = stage_combine(rv, a.base[o], k1, k23, a.final, a.cx, a.cf, sgn, jump);
    if (direct) a.out[o] = res;
    return res;
  };
  if (pair) {
    // every operand of the four elements requested before the first is used, none under a branch (clamped addresses; absent
    // operands read a valid address and are dropped by a select): one memory latency per workgroup instead of four in a row
    size_t eo[4]; bool eok[4];
    double e0[4], e1[4], bs[4], jp[4], k1[4], k23[4];
    const double* E1 = a.mid_e ? a.E1 : a.E0;
    const double* Jp = (a.final && a.has_j) ? a.J : a.base;
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {                           // C col = lane & 15, row = (lane >> 4) + 4 r
      const int r = Ra + 16 * wm + (lane >> 4) + 4 * r4, j = Rb + 16 * wn + (lane & 15);
      eok[r4] = r < D && j < D;
      eo[r4] = (size_t)min(r, D - 1) * D + min(j, D - 1);
      e0[r4] = a.E0[eo[r4]]; e1[r4] = E1[eo[r4]]; bs[r4] = a.base[eo[r4]]; jp[r4] = Jp[eo[r4]];
      k1[r4] = a.K1[eo[r4]]; k23[r4] = a.K23[eo[r4]];
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      const int rr = 16 * wm + (lane >> 4) + 4 * r4, cc = 16 * wn + (lane & 15);
      const double w = w1[r4], wt = w2[r4];
      const double e = a.mid_e ? 0.5 * (e1[r4] + e0[r4]) : e0[r4];
      const double rv = a.fwd ? ((-w - wt) + e) : ((-e + wt) + w);
      const double jump = (a.final && a.has_j) ? jp[r4] : 0.0;
      const double res = stage_combine(rv, bs[r4], a.final >= 2 ? k1[r4] : 0.0, a.final == 3 ? k23[r4] : 0.0, a.final, a.cx, a.cf, sgn, jump);
      if (eok[r4]) {
        if (a.kstore == 1) a.K1[eo[r4]] = rv;
        else if (a.kstore == 2) a.K23[eo[r4]] = rv;
        else if (a.kstore == 3) a.K23[eo[r4]] = k23[r4] + rv;
        a.out[eo[r4]] = res;
      }
      tcA[rr][cc] = res;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
    for (int q = 0; q < 4; q++) {                              // the mirror tile
      const int rr = ty + 8 * q;
      const int r2 = Rb + rr, j2 = Ra + tx;
      if (r2 < D && j2 < D) a.out[(size_t)r2 * D + j2] = tcA[tx][rr];
    }
    return;
  }
  // two diagonal tiles: the missing one of (w, wt) is the transposed accumulator
#pragma unroll
  for (int r4 = 0; r4 < 4; r4++) {
    const int rr = 16 * wm + (lane >> 4) + 4 * r4, cc = 16 * wn + (lane & 15);
    tcA[rr][cc] = w1[r4]; tcB[rr][cc] = w2[r4];
  }
  __syncthreads();
#pragma unroll
  for (int r4 = 0; r4 < 4; r4++) {
    const int rr = 16 * wm + (lane >> 4) + 4 * r4, cc = 16 * wn + (lane & 15);
    // (the state is stored from the upper triangle to both sides: exactly symmetric, like a pair's two tiles)
    if (Ra + rr < D && Ra + cc < D) {
      const double res = element(Ra + rr, Ra + cc, w1[r4], tcA[cc][rr], rr <= cc);
      if (rr < cc) a.out[(size_t)(Ra + cc) * D + Ra + rr] = res;
    }
    if (two && Rb + rr < D && Rb + cc < D) {
      const double res = element(Rb + rr, Rb + cc, tcB[cc][rr], w2[r4], rr <= cc);
      if (rr < cc) a.out[(size_t)(Rb + cc) * D + Rb + rr] = res;
    }
  }
}

hipError_t launch_stage_fused(int kind, const StageArgs& a, hipStream_t st) {
  const int D = a.D;
  const int nt = (D + TS - 1) / TS, nvec = (D + (NT / 64) - 1) / (NT / 64);
  if (kind == 0) {
    const dim3 grid(nt * (nt + 1) / 2 + nvec, 1, a.nb);
    if (a.fwd) {
      if (a.M1) hipLaunchKernelGGL((k_stage_prod<false, true>), grid, dim3(NT), 0, st, a);
      else hipLaunchKernelGGL((k_stage_prod<false, false>), grid, dim3(NT), 0, st, a);
    } else {
      if (a.M1) hipLaunchKernelGGL((k_stage_prod<true, true>), grid, dim3(NT), 0, st, a);
      else hipLaunchKernelGGL((k_stage_prod<true, false>), grid, dim3(NT), 0, st, a);
    }
  } else {
    const int nmat = nt * (nt - 1) / 2 + (nt + 1) / 2;
    const dim3 grid((VGPA_WIDE_XCD ? 8 * ((nmat + 7) / 8) : nmat) + nvec, 1, a.nb);
    if (a.fwd) hipLaunchKernelGGL((k_stage_wide<false>), grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((k_stage_wide<true>), grid, dim3(NT), 0, st, a);
  }
  return hipGetLastError();
}

}  // namespace ld
}  // namespace vgpa
