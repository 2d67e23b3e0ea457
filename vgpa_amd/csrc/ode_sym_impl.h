// Symmetric-unit stepping kernels (round 2, second design): the slope of the symmetric recursion on the fp64 matrix cores
// WITHOUT a transpose exchange, one workgroup barrier per Runge-Kutta stage, two workgroups (problems) per CU.
// (Included by ode_mfma_m{0,1,2,3}.hip next to ode_mfma_impl.h, which keeps the role-specialised kernels.)
//
// Math.  With X symmetric (X = S_k forward, Psi_t backward) and Aop the operand matrix (A^T forward, A backward; row index
// = contraction index) the slope of both recursions is f = -Z, Z = Aop^T X + X^T Aop - F  (ode_solver.py:60,94; F = Sigma
// forward, dEsde_dS backward).  Z is symmetric, so only the blocks on and above the diagonal are computed -- each as
//     Z[I][J] = sum_k Aop[k][I] X[k][J] + sum_k X[k][I] Aop[k][J] - F[I][J]
// i.e. TWO chains of 4x4x4 matrix-core products into ONE accumulator.  The lane that holds an accumulator element owns the
// Runge-Kutta state of that element (S_k / Psi_t, the slope sum): it applies the stepper right behind the last product
// and publishes the new stage state at (row, col) AND at (col, row) of the next stage's operand buffer.  Nothing is
// exchanged, nothing is read back, the element-wise work of one group of units runs under the products of the next.
//
// Mapping to gfx950.
//   * Unit = 2x2 blocks of 4x4 = one v_mfma_f64_4x4x4_4b_f64 accumulator: super-block (c, j) of the upper triangle (or
//     (j, c) of the lower one -- same thing by symmetry).  Units are dealt to waves in RUNS of two that share the
//     super-row c, hence their a-fragments (sym_run): D = 40 has 15 units = 8 runs, two runs per wave; per k-pair a wave
//     reads 12 fragments (ds_read_b128, each feeding two k-steps) for 16 products.
//   * LDS: two stage-state buffers (read X_cur, publish into X_next: no second barrier), two A-operand buffers (start /
//     end point, mid-point), all in ONE layout: element (k, c) at (k >> 1) * LD + 2 (c ^ ((k >> 1) & 3)) + (k & 1),
//     LD = 16 (mod 32) doubles (a ds_read_b128 lane group holds lanes of TWO adjacent row pairs, each reading the same 8
//     units of a 2x2-block operand: 128 bytes apart they use disjoint halves of the 64 banks; = 0 (mod 16) keeps the
//     mirror publish below conflict-free).  Rows 2p, 2p+1 share a 16-byte unit (one read feeds two k-steps); the XOR of the unit's
//     column with the row pair's low bits costs the fragment reads nothing (row pair = 4 kp + lane / 16, so the XOR is a
//     per-lane constant) and makes the mirror publish conflict-free (the 16 lanes of a store group hit 16 different
//     8-byte slots).  60 KB at D = 40: two workgroups per CU, 256 registers each.
//   * The vector recursion (m / lambda) is independent of the matrix one.  Its inner product is split over all 256
//     threads (partial sums before the barrier); behind the barrier EVERY wave sums them and advances its own copy of the
//     vector, so no second barrier orders "vector published" before "next partial sums".
//   * A(t) is staged HBM -> registers -> LDS one step ahead; S_k / Psi_t go back to HBM from the stage buffer (coalesced
//     row-pair items).  Everything of a stage that is not the product is ONE LDS round trip behind it (`tail`).
// Measurements, what was tried and what bounds the kernel: DESIGN.md s.4.1, s.7, EXPERIMENTS.md s.2, s.9, s.11; micro-benchmark tools/ubench/sym_product.hip.
#pragma once
#include <type_traits>
#include "ode_mfma_impl.h"
#include <cstdlib>
#include <cstring>

namespace vgpa {
namespace sym {

using mfma::cdiv;
using mfma::cmax;
using mfma::cmin;
using mfma::d2_t;
using mfma::ldg;
using mfma::ldu;
using mfma::n_stages;
using mfma::settle;
using mfma::stage_op;
using mfma::stg;
#ifdef VGPA_STAMPS
using mfma::g_stamp;
#endif
using mfma::OP_M;
using mfma::OP_R;
using mfma::OP_X;

constexpr int kMaxNB = 16;          // D <= 64
#ifndef VGPA_SYM_INTERLEAVE
#define VGPA_SYM_INTERLEAVE 1          // fragment reads between the products (sched_group_barrier) instead of in blocks
#endif
#ifndef VGPA_SYM_TAILPRIO
#define VGPA_SYM_TAILPRIO 1
#endif
#ifndef VGPA_ABL_NOFRAG
#define VGPA_ABL_NOFRAG 0              // DIAGNOSTIC (wrong results): only the first pipeline step of a stage reads fragments -- what the LDS reads cost
#endif
#ifndef VGPA_ABL_NOSTORE
#define VGPA_ABL_NOSTORE 0             // DIAGNOSTIC (wrong results): the stage state is not written to HBM
#endif
#ifndef VGPA_ABL_NOVEC
#define VGPA_ABL_NOVEC 0               // DIAGNOSTIC (wrong results): no vector recursion (partial inner products, per-wave update)
#endif
#ifndef VGPA_SYM_OP_PIN
#define VGPA_SYM_OP_PIN 0              // outer-product cover: 1 = the written order of products, rotations and reads is pinned (slower)
#endif
#ifndef VGPA_SYM_LOOP1
#define VGPA_SYM_LOOP1 1               // cover kernels: the loop unit (diagonal blocks) runs one chain + an in-block transpose
#endif

// ---- runs: a cover of the unordered pairs {c, j} of super-block indices (loops included) by stars of <= 2 edges ---------
// Run = super-row c with up to two partners.  Greedy: every loop (c, c) with the edge to c + 1; then, vertex by vertex, two
// uncovered edges at a time; what is left are single edges.  Returns the `want`-th run (c = -1 beyond the last one) and,
// through `count`, their number.  Bit (8 i + j), i <= j, of `cov` = pair covered.
struct Run { int c, j0, j1; };

__host__ __device__ constexpr unsigned long long pair_bit(int i, int j) {
  return 1ull << (i <= j ? 8 * i + j : 8 * j + i);
}

__host__ __device__ constexpr Run sym_run(int nsb, int want, int* count) {
  unsigned long long cov = 0ull;
  int n = 0;
  Run res{-1, -1, -1};
  for (int c = 0; c < nsb; c++) {
    int j1 = (c + 1) % nsb;
    if (j1 == c || (cov & pair_bit(c, j1))) j1 = -1;
    cov |= pair_bit(c, c);
    if (j1 >= 0) cov |= pair_bit(c, j1);
    if (n == want) res = Run{c, c, j1};
    n++;
  }
  for (int v = 0; v < nsb; v++) {
    for (;;) {
      int a = -1, b = -1;
      for (int j = 0; j < nsb; j++) {
        if (j == v || (cov & pair_bit(v, j))) continue;
        if (a < 0) a = j;
        else if (b < 0) b = j;
      }
      if (b < 0) break;
      cov |= pair_bit(v, a) | pair_bit(v, b);
      if (n == want) res = Run{v, a, b};
      n++;
    }
  }
  for (int v = 0; v < nsb; v++)
    for (int j = v + 1; j < nsb; j++) {
      if (cov & pair_bit(v, j)) continue;
      cov |= pair_bit(v, j);
      if (n == want) res = Run{v, j, -1};
      n++;
    }
  if (count) *count = n;
  return res;
}

__host__ __device__ constexpr int sym_run_count(int nsb) {
  int n = 0;
  (void)sym_run(nsb, -2, &n);
  return n;
}

// ---- fragment covers (round 3): NSB = 5 (D = 33 .. 40) ----------------------------------------------------------------------
// The four blocks of one v_mfma_f64_4x4x4_4b need not form a 2x2 super-block: block q of a product takes its A operand from
// the lanes of block q of one fragment register and its B operand from block q of another, so a fragment is described by a MAP
// q -> block column, and ONE fragment register serves every product whose row side (or column side) uses that map.  A wave
// holds the fragments of four maps (a0, a1 | b0, b1), of both operand buffers, and multiplies the "rectangle" of units (a0,b0),
// (a0,b1), (a1,b0), (a1,b1): 8 fragment reads per k-pair for 16 products (runs: 12 for 16); when b0 IS a1 (`alias`: a
// "triangle" a0-a1-b1 plus the loop unit (a1,a1)) 6 reads.  The maps below cover the 55 unordered block pairs of the upper
// triangle of a 10 x 10 block matrix with one rectangle wave (3 units) and three triangle waves (4 units): 26 fragment reads per
// k-pair and workgroup instead of 45, the same 15 units = 300 products per stage (280 since the loop units run one chain: see
// kCoverLoopSlot).  (Four triangle waves cannot do it: 16
// triangles do not cover K_10.)  Found by simulated annealing under three side conditions: (i) a ds_read_b128 lane group -- it
// mixes lanes of blocks 0, 3 of one row pair with blocks 1, 2 of the next -- touches every bank once
// ({m0, m3, m1 + 2, m2 + 2} distinct mod 4 wherever the blocks differ); (ii) / (iii) the four blocks of a unit differ in
// (row-block parity, column-block parity) as far as possible, which is what keeps the mirror / direct publish at most two-way
// bank conflicted.  Five block pairs are covered twice; the first occurrence owns the elements.
constexpr int kCoverPat[4][2] = {{0, 2}, {0, 3}, {1, 2}, {1, 3}};          // unit slot -> (row-side map, column-side map)
constexpr int kCoverMaps[4][4][4] = {                                      // [wave][map a0, a1, b0, b1][block q]
    {{1, 8, 6, 1}, {2, 6, 7, 3}, {6, 5, 5, 4}, {9, 2, 5, 2}},
    {{9, 9, 8, 0}, {8, 1, 1, 2}, {8, 1, 1, 2}, {4, 5, 0, 5}},
    {{6, 9, 3, 6}, {4, 0, 5, 9}, {4, 0, 5, 9}, {0, 7, 4, 3}},
    {{8, 1, 7, 8}, {6, 7, 2, 3}, {6, 7, 2, 3}, {7, 3, 4, 0}}};
constexpr int kCoverUsed[4] = {0xB, 0xF, 0xF, 0xF};                        // bit s: unit slot s of the wave is in use (the rectangle wave has three)
constexpr bool kCoverAlias[4] = {false, true, true, true};                 // map b0 is map a1
// Slot 2 = (a1, b0) is the LOOP unit of a triangle wave (b0 is a1): its four blocks are diagonal blocks -- and every diagonal block
// of the matrix sits in one of the three loop units (static_assert below).  For a diagonal block the second chain X^T Aop is the
// transpose of the first, block by block, so the loop unit runs ONE chain and adds its accumulator's in-block transpose (one
// cross-lane exchange per stage): 70 products per wave and stage instead of 80.  The rectangle wave parks its unused unit in
// slot 2, so that the instruction stream is the same for all four waves.
constexpr int kCoverLoopSlot = 2;
__host__ __device__ constexpr bool cover_slot_used(int w, int s) { return (kCoverUsed[w] >> s) & 1; }
// does (wave w, slot s, block q) own its block pair (first occurrence in (w, s, q) order)?
__host__ __device__ constexpr bool cover_owner(int w, int s, int q) {
  const int I = kCoverMaps[w][kCoverPat[s][0]][q], J = kCoverMaps[w][kCoverPat[s][1]][q];
  const int lo = I < J ? I : J, hi = I < J ? J : I;
  for (int w2 = 0; w2 <= w; w2++)
    for (int s2 = 0; s2 < 4; s2++)
      for (int q2 = 0; q2 < 4; q2++) {
        if (!cover_slot_used(w2, s2)) continue;
        if (w2 == w && (s2 > s || (s2 == s && q2 >= q))) return true;
        const int I2 = kCoverMaps[w2][kCoverPat[s2][0]][q2], J2 = kCoverMaps[w2][kCoverPat[s2][1]][q2];
        if ((I2 < J2 ? I2 : J2) == lo && (I2 < J2 ? J2 : I2) == hi) return false;
      }
  return true;
}
// the same as a table (bit q of [wave][slot]): a run-time (wave, block) looks it up instead of evaluating the search
struct CoverOwnerTab { unsigned char m[4][4]; };
__host__ __device__ constexpr CoverOwnerTab cover_owner_tab() {
  CoverOwnerTab t{};
  for (int w = 0; w < 4; w++)
    for (int s = 0; s < 4; s++) {
      unsigned char bits = 0;
      for (int q = 0; q < 4; q++)
        if (cover_slot_used(w, s) && cover_owner(w, s, q)) bits = (unsigned char)(bits | (1u << q));
      t.m[w][s] = bits;
    }
  return t;
}
__device__ constexpr CoverOwnerTab kCoverOwner = cover_owner_tab();
__host__ __device__ constexpr int cover_pairs_owned() {
  int n = 0;
  for (int w = 0; w < 4; w++)
    for (int s = 0; s < 4; s++)
      for (int q = 0; q < 4; q++) n += (cover_slot_used(w, s) && cover_owner(w, s, q)) ? 1 : 0;
  return n;
}
static_assert(cover_pairs_owned() == 55, "the fragment cover must own every block pair of the 10 x 10 upper triangle exactly once");
// diagonal block pairs appear in the loop slot of the triangle waves and nowhere else
__host__ __device__ constexpr bool cover_diagonals_in_loop_units() {
  for (int w = 0; w < 4; w++)
    for (int s = 0; s < 4; s++)
      for (int q = 0; q < 4; q++) {
        if (!cover_slot_used(w, s)) continue;
        const bool diag = kCoverMaps[w][kCoverPat[s][0]][q] == kCoverMaps[w][kCoverPat[s][1]][q];
        const bool loop = kCoverAlias[w] && s == kCoverLoopSlot;
        if (diag != loop) return false;
      }
  return !cover_slot_used(0, kCoverLoopSlot);
}
static_assert(cover_diagonals_in_loop_units(), "single-chain loop units: every diagonal block in slot 2 of a triangle wave, slot 2 of the rectangle wave unused");

// ---- outer-product cover (round 4): NSB = 5 (D = 33 .. 40) ------------------------------------------------------------------
// The product phase of the fragment cover above is bound by its LDS fragment reads as much as by the matrix pipe (EXPERIMENTS.md
// s.9): 8 (6) ds_read_b128 per k-pair and wave for 14 products.  A fragment register of a map m also holds, up to a rotation of
// its four 4-lane block groups inside every 16-lane row, the fragment of every ROTATED map rot_r(m)[q] = m[(q - r) & 3] -- and a
// rotation of lanes inside a row is what the DPP operand of a v_mov does (row_ror:4r, no LDS).  So a wave reads the fragments of
// only TWO maps (P, Q) of both operand buffers -- 4 reads per k-pair -- and multiplies the units
//   slot 0: (P, P)          the loop unit: four diagonal blocks, one chain + in-block transpose (as kCoverLoopSlot above)
//   slot 1: (rot_1 P, P)    the 4-cycle P0-P1-P2-P3 of block pairs inside P
//   slot 2: (rot_1 Q, Q)    ... inside Q
//   slot 3: (P, Q)          P[q] with Q[q]
// with ONE instruction stream for all four waves: 14 products per k-pair as before, 16 fragment reads per k-pair and workgroup
// instead of 26, 16 v_mov_dpp per k-pair and wave for the rotated row sides.  The maps below were found by simulated annealing
// (cover all 55 block pairs, every diagonal block in a loop unit; the ds_read_b128 bank condition (i) above on every map; as few
// publish bank conflicts (ii)/(iii) as possible).
struct OpUnit { int rmap, rot, cmap; };              // unit = (rot_rot(map rmap), map cmap); maps: 0 = P, 1 = Q
constexpr OpUnit kOpUnits[4] = {{0, 0, 0}, {0, 1, 0}, {1, 1, 1}, {0, 0, 1}};
constexpr int kOpLoopSlot = 0;
constexpr int kOpMaps[4][2][4] = {                   // [wave][P, Q][block q]
    {{7, 4, 6, 1}, {3, 8, 2, 9}},
    {{8, 9, 8, 9}, {1, 5, 0, 4}},
    {{5, 8, 6, 3}, {6, 7, 9, 0}},
    {{2, 3, 1, 0}, {5, 4, 2, 7}}};
__host__ __device__ constexpr int op_row_block(int w, int s, int q) { return kOpMaps[w][kOpUnits[s].rmap][(q - kOpUnits[s].rot) & 3]; }
__host__ __device__ constexpr int op_col_block(int w, int s, int q) { return kOpMaps[w][kOpUnits[s].cmap][q]; }
__host__ __device__ constexpr bool op_owner(int w, int s, int q) {      // first occurrence of the block pair in (w, s, q) order
  const int I = op_row_block(w, s, q), J = op_col_block(w, s, q);
  const int lo = I < J ? I : J, hi = I < J ? J : I;
  for (int w2 = 0; w2 <= w; w2++)
    for (int s2 = 0; s2 < 4; s2++)
      for (int q2 = 0; q2 < 4; q2++) {
        if (w2 == w && (s2 > s || (s2 == s && q2 >= q))) return true;
        const int I2 = op_row_block(w2, s2, q2), J2 = op_col_block(w2, s2, q2);
        if ((I2 < J2 ? I2 : J2) == lo && (I2 < J2 ? J2 : I2) == hi) return false;
      }
  return true;
}
__host__ __device__ constexpr CoverOwnerTab op_owner_tab() {
  CoverOwnerTab t{};
  for (int w = 0; w < 4; w++)
    for (int s = 0; s < 4; s++) {
      unsigned char bits = 0;
      for (int q = 0; q < 4; q++)
        if (op_owner(w, s, q)) bits = (unsigned char)(bits | (1u << q));
      t.m[w][s] = bits;
    }
  return t;
}
__device__ constexpr CoverOwnerTab kOpOwner = op_owner_tab();
__host__ __device__ constexpr int op_pairs_owned() {
  int n = 0;
  for (int w = 0; w < 4; w++)
    for (int s = 0; s < 4; s++)
      for (int q = 0; q < 4; q++) n += op_owner(w, s, q) ? 1 : 0;
  return n;
}
__host__ __device__ constexpr bool op_pairs_in_range() {
  for (int w = 0; w < 4; w++)
    for (int m = 0; m < 2; m++)
      for (int q = 0; q < 4; q++)
        if (kOpMaps[w][m][q] < 0 || kOpMaps[w][m][q] > 9) return false;
  return true;
}
// every diagonal block must be owned by a loop unit (the single-chain trick is applied to slot kOpLoopSlot and to nothing else;
// a diagonal block in a two-chain unit would be right too, but the count below is what proves the 10 diagonals are covered)
__host__ __device__ constexpr int op_diagonals_in_loop_units() {
  int n = 0;
  for (int w = 0; w < 4; w++)
    for (int q = 0; q < 4; q++) n += op_owner(w, kOpLoopSlot, q) ? 1 : 0;
  return n;
}
static_assert(op_pairs_in_range() && op_pairs_owned() == 55, "the outer-product cover must own every block pair of the 10 x 10 upper triangle exactly once");
static_assert(kOpUnits[kOpLoopSlot].rmap == kOpUnits[kOpLoopSlot].cmap && kOpUnits[kOpLoopSlot].rot == 0 && op_diagonals_in_loop_units() == 10,
              "the loop unit is (m, m) unrotated and the loop units own all ten diagonal blocks");

template <int R>
__device__ __forceinline__ double rot_d(double v) {          // one double of a fragment, rotated like rot_blocks below
  typedef int i2_t __attribute__((ext_vector_type(2)));
  i2_t x = __builtin_bit_cast(i2_t, v), y;
  y[0] = __builtin_amdgcn_mov_dpp(x[0], 0x120 + 4 * R, 0xf, 0xf, false);
  y[1] = __builtin_amdgcn_mov_dpp(x[1], 0x120 + 4 * R, 0xf, 0xf, false);
  return __builtin_bit_cast(double, y);
}
// the four 4-lane block groups of every 16-lane row rotated by R: lane of block q receives the lane of block q - R
// (v_mov_b32 ... row_ror:4R: destination lane l reads lane (l - 4R) mod 16 of its row, tools/ubench/dpp_ror.hip)
template <int R>
__device__ __forceinline__ d2_t rot_blocks(d2_t v) {
  if constexpr (R == 0) return v;
  else {
    typedef int i4_t __attribute__((ext_vector_type(4)));
    i4_t x = __builtin_bit_cast(i4_t, v), y;
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = __builtin_amdgcn_mov_dpp(x[i], 0x120 + 4 * R, 0xf, 0xf, false);
    return __builtin_bit_cast(d2_t, y);
  }
}

template <int NB_, int NW_ = 4>      // NW: waves per workgroup (4: two workgroups per CU; 8: one problem per CU, see k_ode_sym)
struct SGeo {
  static constexpr int NB = NB_;
  static constexpr int NW = NW_, NT = 64 * NW_;
  static constexpr int P = 4 * NB;                 // padded to 4x4 blocks
  static constexpr int NSB = (NB + 1) / 2;         // 8x8 super-blocks per dimension
  static constexpr int KKE = 2 * NSB;              // k-steps (4 rows each), even
  static constexpr int PP = 4 * KKE;               // padded to super-blocks
  static constexpr int RP = 2 * KKE;               // row pairs of a buffer
  static constexpr int NKP = NSB;                  // k-pairs (fragment reads) per product chain
  static constexpr int LD = 32 * cdiv(2 * PP - 16, 32) + 16;   // doubles per row pair, = 16 (mod 32): see the header
  static constexpr int XS = RP * LD;
  static constexpr int NRUNS = sym_run_count(NSB);
  static constexpr int NR = cdiv(NRUNS, 4);        // runs per wave
  static constexpr int MAXS = 2 * NR;              // unit slots per wave
  // vector recursion
  static constexpr int XV = PP + 4;                // one copy of the stage vector per wave
  static constexpr int NPARTF = cmin(NT / PP, RP);          // forward: lane = (column i, part of the row pairs)
  static constexpr int RPP = cdiv(RP, NPARTF);
  static constexpr int NPF = cmin(NPARTF, cdiv(RP, RPP));    // parts that hold a sum
  static constexpr int NCB = PP / 4;                         // backward: lane = (row pair, part of the 4-column blocks)
  static constexpr int NPARTB = cmin(NT / RP, NCB);
  static constexpr int CBP = cdiv(NCB, NPARTB);
  static constexpr int NPART = cmax(NPF, NPARTB);
  static constexpr int PV = NPART * PP;
  static constexpr int NIT = cdiv((P / 2) * P, NT);         // row-pair items per thread (column fastest)
  static constexpr int NTILE = cdiv(P / 2, 8) * cdiv(P, 8);  // forward staging: 8 x 8 tiles of (row pair, column)
  static constexpr int NITF = cdiv(NTILE, NW);
  static constexpr size_t LDS_DOUBLES = (size_t)4 * XS + NW * XV + 2 * PV + 2 * NT;
  static_assert(NPARTF >= 1 && NPARTB >= 1, "dimension too large for the workgroup");
};

// element (k, c) of a buffer
template <int NB>
__host__ __device__ constexpr int elem_off(int k, int c) {
  return (k >> 1) * SGeo<NB>::LD + 2 * (c ^ ((k >> 1) & 3)) + (k & 1);
}
// 16-byte unit (row pair p, column c)
template <int NB>
__host__ __device__ constexpr int unit_off(int p, int c) {
  return p * SGeo<NB>::LD + 2 * (c ^ (p & 3));
}

// row-pair items, column fastest over the threads: the HBM side of the state stores and of the backward operand staging
template <int NB, int NW = 4>
struct ItemTab {
  static constexpr int NIT = SGeo<NB, NW>::NIT;
  int lo[NIT];          // unit offset inside a buffer, or -1
  unsigned g0[NIT];     // byte offset of element (2p, c) in a D x D matrix
  unsigned g1[NIT];     // ... of element (2p+1, c); = g0 when that row does not exist (the value is dropped)
  bool two[NIT];        // row 2p+1 exists
  bool st0[NIT], st1[NIT];   // the state store writes element (2p, c) / (2p+1, c): all that exist, or -- packed lower triangle -- c <= row
};
template <int NB, int NW>
__device__ __forceinline__ void build_items(int D, int tid, ItemTab<NB, NW>& T, bool packed_lower = false) {
  using g = SGeo<NB, NW>;
#pragma unroll
  for (int q = 0; q < g::NIT; q++) {
    const int e = tid + g::NT * q;
    const int p = e / g::P, c = e - p * g::P;
    const bool ok = 2 * p < D && c < D;
    T.lo[q] = ok ? unit_off<NB>(p, c) : -1;
    T.g0[q] = ok ? 8u * (unsigned)(2 * p * D + c) : 0u;
    T.two[q] = ok && 2 * p + 1 < D;
    T.g1[q] = T.two[q] ? T.g0[q] + 8u * (unsigned)D : T.g0[q];
    T.st0[q] = ok; T.st1[q] = T.two[q];
    if (packed_lower) {        // element (r, c), c <= r, at r (r + 1) / 2 + c
      const int r0 = 2 * p, r1 = 2 * p + 1;
      T.st0[q] = ok && c <= r0;
      T.st1[q] = T.two[q] && c <= r1;
      T.g0[q] = T.st0[q] ? 8u * (unsigned)(r0 * (r0 + 1) / 2 + c) : 0u;
      T.g1[q] = T.st1[q] ? 8u * (unsigned)(r1 * (r1 + 1) / 2 + c) : 0u;
    }
  }
}

// forward operand staging (operand = A^T: unit (sp, so) = A[so][2sp], A[so][2sp+1]) in 8 x 8 tiles: the 8 lanes of a
// ds_write_b128 group store 8 consecutive columns of one row pair (conflict-free), a wave reads 8 rows x 128 contiguous bytes
template <int NB, int NW = 4>
struct TileTab {
  static constexpr int NIT = SGeo<NB, NW>::NITF;
  int lo[NIT];
  unsigned g0[NIT], g1[NIT];   // byte offsets of A[so][2sp], A[so][2sp+1] (g1 = g0 when column 2sp+1 does not exist)
  bool two[NIT];
};
template <int NB, int NW>
__device__ __forceinline__ void build_tiles(int D, int wave, int lane, TileTab<NB, NW>& T) {
  using g = SGeo<NB, NW>;
  constexpr int nsot = cdiv(g::P, 8);
#pragma unroll
  for (int q = 0; q < g::NITF; q++) {
    const int tile = wave + NW * q;
    const int spt = tile / nsot, sot = tile - spt * nsot;
    const int sp = 8 * spt + (lane >> 3), so = 8 * sot + (lane & 7);
    const bool ok = 2 * sp < D && so < D;
    T.lo[q] = ok ? unit_off<NB>(sp, so) : -1;
    T.g0[q] = ok ? 8u * (unsigned)(so * D + 2 * sp) : 0u;
    T.two[q] = ok && 2 * sp + 1 < D;
    T.g1[q] = T.two[q] ? T.g0[q] + 8u : T.g0[q];
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier: the fence also waits
// for every outstanding HBM access (vmcnt(0)) -- here the state stores of stage 0 and the prefetches of stage 1, i.e. a full
// HBM round trip in front of two of the four barriers of a step.  The waves of a workgroup exchange nothing through HBM.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Memory waits.  The compiler's wait-count pass cannot tell, across the loop back-edge, how old a loaded value is: the first
// use of ANY value loaded in the previous iteration becomes s_waitcnt vmcnt(0) -- it waits for every load and store issued so
// far.  So a step issues all its HBM loads at ONE point (behind the staging of stage min(1, NS-1): A, the forcing terms, the
// jumps of two steps ahead) and consumes them at ONE point (the rotation at the top of the next step, >= 1 stage later).
// NW = 8 (round 4 experiment; fragment cover only; opt-in, see eight_waves()): ONE problem per CU on EIGHT waves -- wave w and wave w + 4 (same SIMD) split the four units
// of cover wave w at its half-step boundary: units (a0, b0), (a0, b1) | units (a1, b0), (a1, b1).  Each wave multiplies 40 instead of
// 70 products per stage (both chains of every unit: the loop units' single-chain trick would make the two halves different
// instruction streams), owns half the elements and does half the chores, so the per-stage latency chain of a lone problem --
// products, stepper, publish, barrier, first fragments: ~2 300 cycles with four waves -- loses most of its product share.  For
// batches up to one problem per CU (the reference's own use case is ONE optimisation); larger batches keep two four-wave
// workgroups per CU, whose 256 registers per wave an eight-wave workgroup pair cannot have.
// HLP (round 4; fragment cover, four product waves): FOUR HELPER WAVES beside the four product waves of a problem, one of each per SIMD.
// A wave issues one instruction at a time and an fp64 product holds its issue slot for all 16 cycles (a v_mov between two products costs
// its full 8 cycles, tools/ubench/mfma_dpp_overlap.hip), so everything of a stage that is not the product -- the vector recursion,
// operand staging, the state's way to HBM, the step's HBM loads: the "chores" -- lengthens a lone workgroup's stage by what it
// issues (1 120 cycles of products in a 2 500-cycle stage).  With helpers the product waves keep the products, the element-wise
// stepper and the publish; helper wave w + 4 does the chores of product wave w (same thread -> item tables), concurrently, on the
// same SIMD's other issue slot; both meet at the one barrier per stage that was there before.  Nothing a chore writes is read during
// its stage and nothing it reads is written (see tailC), so no new synchronisation.  512 threads, 256 registers each: ONE workgroup
// per CU -- the shape for up to one problem per CU (the reference's own use case is one optimisation); larger batches keep two
// four-wave workgroups per CU.
// GF (round 5; backward RK4 helper-wave kernels with Q'' on): THE GRADIENT ASSEMBLY INSIDE THE BACKWARD KERNEL, on a THIRD set of four waves
// (768 threads: product, helper and gradient wave of every SIMD; <= 168 registers each).  gLa_t = dt ((Sigma^-1 (<df/dx> + A_t) -
// 2 Psi_t) S_t - u m_t^T), u = Sigma^-1 (-<f> - A_t m_t + b_t) + lam_t, gLb_t = dt u (variational.py:263-288, 302-337) is a closed form on
// what the recursion holds in LDS when step k starts (Psi_t in the stage buffer, A_t in the start-point operand, lam_t from the helpers)
// plus S_t (packed), m_t, <f>_t, A_t m_t, b_t from HBM.  One grid point per step, one phase per barrier interval of the step:
//   stage 0: out-buffer of the previous point -> HBM (whole lines); Q''^T = (A_t / sigma^2 - 2 Psi_t)^T from the row-pair units of the
//            two buffers (both are stable during stage 0) and S_t (both triangles from the packed stream) into two more operand
//            buffers of the steppers' layout; m_t; the helpers leave lam_t
//   stage 1: the four-entry band of Sigma^-1 <df/dx> (lorenz_96.py:35-83) is added to Q^T by the rows' lanes; u, gLb -> HBM; the next
//            grid point's HBM loads
//   stage 2, 3: Q S on the matrix cores -- 25 units of four 4 x 4 blocks over the four gradient waves (see kGradA) -- then
//            dt (Q S - u m^T) into the out-buffer
// Q''_t never goes to HBM (-25.6 KB per grid point), the separate assembly kernel and its launch are gone.  Why a third wave and not
// the helpers: a helper's stage is a chain of dependent LDS round trips as long as the product waves' stage (measured: the product
// waves WAIT for the helpers in stages 0 and 1, tools/ubench/ode_gf_loop.hip), so everything added to it lengthened the step by what it
// took (first version of this round: +1.7 ms per 256-problem launch); the SIMD's third wave runs its own chain beside the two.
constexpr int kGradA[3][4] = {{0, 1, 3, 2}, {4, 5, 7, 6}, {8, 9, 8, 9}};      // row-side maps (block rows of Q); column-side map s: (s, s, s+1, s+1) mod 10
__host__ __device__ constexpr int grad_bmap(int s, int q) { return (s + (q >> 1)) % 10; }
// gradient wave w < 3: maps a0, a1 x column maps 3w, 3w+1, 3w+2 (six units); wave 3: (a0, a1) x 9 and a2 x {9, 1, 3, 5, 7} (seven units):
// every ordered block pair (I, J) of the 10 x 10 result exactly once (static_assert below), 23 fragment reads per k-pair and workgroup
__host__ __device__ constexpr int grad_bsel(int w, int i) { return w < 3 ? 3 * w + i : (i == 0 ? 9 : 2 * i - 1); }
__host__ __device__ constexpr bool grad_cover_exact() {
  int cnt[10][10] = {};
  for (int w = 0; w < 4; w++)
    for (int q = 0; q < 4; q++) {
      if (w < 3) {
        for (int ai = 0; ai < 2; ai++)
          for (int bi = 0; bi < 3; bi++) cnt[kGradA[ai][q]][grad_bmap(grad_bsel(w, bi), q)]++;
      } else {
        for (int ai = 0; ai < 3; ai++) cnt[kGradA[ai][q]][grad_bmap(grad_bsel(3, 0), q)]++;
        for (int bi = 1; bi < 5; bi++) cnt[kGradA[2][q]][grad_bmap(grad_bsel(3, bi), q)]++;
      }
    }
  for (int i = 0; i < 10; i++)
    for (int j = 0; j < 10; j++)
      if (cnt[i][j] != 1) return false;
  return true;
}
static_assert(grad_cover_exact(), "the gradient product's units must cover every block of the 10 x 10 result exactly once");

// LDS of the gradient waves, behind the steppers' (offsets in doubles from smem + SGeo::LDS_DOUBLES): two sets of operand buffers
// (Q^T and S_t of the grid point under construction / of the one being multiplied), the out-buffer, dt u and m_t per set, lam_t
template <int NB>
struct GradLds {
  using g = SGeo<NB, 4>;
  static constexpr int QT = 0, S = 2 * g::XS, O = 4 * g::XS, U = O + g::PP * g::PP, M = U + 2 * g::PP, LAM = M + 2 * g::PP;
  static constexpr int DOUBLES = LAM + g::PP;
};

// The gradient waves of k_ode_sym<..., GF> (threads 512 .. 767).  Pipeline over the backward steps (step k starts with Psi_t, t = Np-1-k,
// in stage buffer 0 and A_t in the start-point operand R; both are stable during stage 0):
//   stage 0 of step k: out-buffer (point k-2) -> HBM | build Q^T, S_t, m_t of point k into operand set k & 1 | k-pair 0 of point k-1
//   stage 1:           band of Sigma^-1 <df/dx> into Q^T of point k, u (lam_t from the helpers), gLb -> HBM; HBM loads of point k+1 | k-pair 1 of point k-1
//   stage 2:           k-pairs 2, 3 of point k-1
//   stage 3:           k-pair 4 of point k-1, the result -> out-buffer
// so that the matrix-core work is spread over the whole step.  Barriers: the same count as the other roles (two in the prologue, four
// per step, four behind the loop).
// What this role costs the step is its VECTOR-ALU INSTRUCTION COUNT: while the SIMD's product wave streams fp64 products, another wave's
// vector-ALU instruction issues once per product slot (22 instead of 5 cycles: tools/ubench/mfma_valu_crosswave.hip; LDS and memory
// instructions are not affected).  So: every LDS / HBM address is a precomputed per-lane byte offset (tables below; the operand set is a
// compile-time immediate: the step exists twice), no predicates and no selects -- out-of-range lanes are clamped to a valid duplicate
// that carries the same value -- and the operand is stored as Q^T / -2 = (Psi_t - (A_t + <df/dx>) / (2 sigma^2))^T: one fused
// multiply-add per element to build, the factor -2 dt in the epilogue's multiply-add.
template <int NB>
__device__ __forceinline__ void grad_waves(const OdeArgs& a, double* __restrict__ smem, const int tid) {
#pragma clang fp contract(fast)
  using g = SGeo<NB, 4>;
  using gl = GradLds<NB>;
  constexpr int NT = 256, LD = g::LD, NKP = g::NKP, PP = g::PP;
  static_assert(NKP == 5 && g::NSB == 5, "the gradient product's units are laid out for 33 <= D <= 40");
  constexpr unsigned GB = 8u * (unsigned)g::LDS_DOUBLES;       // byte offset of this role's LDS
  constexpr unsigned SETB = 8u * (unsigned)g::XS;              // bytes between the two operand sets (Q^T, S)
  constexpr unsigned SETV = 8u * (unsigned)PP;                 // ... between the two vector sets (dt u, m)
  constexpr unsigned RB = 8u * 2u * (unsigned)g::XS;           // the start-point operand R (stage buffer 0 is at 0)
  const int lane = tid & 63, wave = tid >> 6;
  const int r4 = lane >> 4, bq = (lane >> 2) & 3, c4 = lane & 3;
  const int prob = (int)blockIdx.x, D = a.D, Np = a.Np, DD = a.D * a.D, n_steps = a.Np - 1, PK = a.D * (a.D + 1) / 2;
  const double dt = a.dt, hq = -0.5 * a.q_scale, m2dt = -2.0 * a.dt;
  char* const lds = reinterpret_cast<char*>(smem);
  auto rd2 = [&](unsigned off) -> d2_t { return *reinterpret_cast<const d2_t*>(lds + off); };
  auto rd1 = [&](unsigned off) -> double { return *reinterpret_cast<const double*>(lds + off); };
  auto wr1 = [&](unsigned off, double v) { *reinterpret_cast<double*>(lds + off) = v; };
  auto tidx = [&](int i) { return Np - 1 - i; };
  const bool vl = lane < D;
  unsigned lane8 = vl ? 8u * (unsigned)lane : 0u;
  double* const gbase = a.g + (size_t)prob * a.strideA;

  // ---- per-lane byte offsets (LDS: from smem; HBM: from the grid point's array) ----------------------------------------------------
  constexpr int NIT = g::NIT;
  constexpr int NPKI = cdiv(PP * (PP + 1) / 2, NT);      // items of the packed S_t per thread
  constexpr int NOUT = cdiv(PP * PP, NT);                // doubles of the out-buffer per thread
  unsigned it_rd[NIT], qt_w0[NIT], qt_w1[NIT];           // row-pair unit (p, c) of Psi / A; Q^T entries (c, 2p), (c, 2p + 1) of set 0
  unsigned s_w0[NPKI], s_w1[NPKI], s_g[NPKI];            // S_t entries (r, c), (c, r) of set 0; the item in the packed stream
#pragma unroll
  for (int q = 0; q < NIT; q++) {
    const int e = tid + NT * q;
    int p = e / g::P, c = e - p * g::P;
    if (!(2 * p < D && c < D)) { p = 0; c = 0; }         // (a duplicate of item (0, 0): same values to the same addresses)
    it_rd[q] = 8u * (unsigned)unit_off<NB>(p, c);
    qt_w0[q] = GB + 8u * (unsigned)(gl::QT + elem_off<NB>(c, 2 * p));
    qt_w1[q] = GB + 8u * (unsigned)(gl::QT + elem_off<NB>(c, 2 * p + 1));      // (row 2p + 1 = D, D odd: a zero of the padding)
  }
#pragma unroll
  for (int q = 0; q < NPKI; q++) {
    int e = tid + NT * q;
    if (e >= PK) e = PK - 1;                             // (a duplicate of the last diagonal element)
    const int r = tri_row(e), c = e - tri_off(r);
    s_w0[q] = GB + 8u * (unsigned)(gl::S + elem_off<NB>(r, c));
    s_w1[q] = GB + 8u * (unsigned)(gl::S + elem_off<NB>(c, r));
    s_g[q] = 8u * (unsigned)e;
  }
  unsigned m_w = GB + 8u * (unsigned)(gl::M + (vl ? lane : 0));      // (every wave writes m_t: the same values)
  // band of <df/dx> (the lanes i < D of wave 0): Q^T entries (i, i), (i+1, i), (i-2, i), (i-1, i); m_{i-1}, m_{i+1}, m_{i-2}; lam_i; u_i
  unsigned bq_o[4], bm_o[3], blam, bu;
  {
    const int i = vl ? lane : 0, ip1 = i + 1 < D ? i + 1 : 0, im1 = i >= 1 ? i - 1 : D - 1, im2 = i >= 2 ? i - 2 : i - 2 + D;
    bq_o[0] = GB + 8u * (unsigned)(gl::QT + elem_off<NB>(i, i));
    bq_o[1] = GB + 8u * (unsigned)(gl::QT + elem_off<NB>(ip1, i));
    bq_o[2] = GB + 8u * (unsigned)(gl::QT + elem_off<NB>(im2, i));
    bq_o[3] = GB + 8u * (unsigned)(gl::QT + elem_off<NB>(im1, i));
    bm_o[0] = GB + 8u * (unsigned)(gl::M + im1);
    bm_o[1] = GB + 8u * (unsigned)(gl::M + ip1);
    bm_o[2] = GB + 8u * (unsigned)(gl::M + im2);
    blam = GB + 8u * (unsigned)(gl::LAM + i);
    bu = GB + 8u * (unsigned)(gl::U + i);
  }
  // fragments of this wave's maps (row side: Q^T, column side: S), the rows' dt u and the columns' m, the out-buffer slot of every
  // result element (wave < 3: units (a, b) = a in {0, 1} x b in {0, 1, 2}; wave 3: (a0, b0), (a1, b0), (a2, b0), (a2, b1..b4))
  unsigned fa_o[3], fb_o[5], eu_o[3], em_o[5], eo_o[7];
  {
    int rowA[3], colJ[5];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int Ia = kGradA[i][bq];
      fa_o[i] = GB + 8u * (unsigned)(gl::QT + r4 * LD + 2 * ((4 * Ia + c4) ^ r4));
      rowA[i] = 4 * Ia + r4;
      eu_o[i] = GB + 8u * (unsigned)(gl::U + rowA[i]);
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int sel = wave < 3 ? 3 * wave + (i < 3 ? i : 0) : (i == 0 ? 9 : 2 * i - 1);      // grad_bsel
      const int Jb = (sel + (bq >> 1)) % 10;
      fb_o[i] = GB + 8u * (unsigned)(gl::S + r4 * LD + 2 * ((4 * Jb + c4) ^ r4));
      colJ[i] = 4 * Jb + c4;
      em_o[i] = GB + 8u * (unsigned)(gl::M + colJ[i]);
    }
#pragma unroll
    for (int u = 0; u < 7; u++) {
      const int ai = wave < 3 ? (u < 6 ? u / 3 : 0) : (u < 3 ? u : 2), bi = wave < 3 ? (u < 6 ? u % 3 : 0) : (u < 3 ? 0 : u - 2);
      const int row = rowA[ai], col = colJ[bi];
      // (results of the padding, D < 4 NB, go to the last slot of the buffer, which D x D elements do not reach)
      eo_o[u] = GB + 8u * (unsigned)(gl::O + ((row < D && col < D) ? row * D + col : PP * PP - 1));
    }
  }
  unsigned oo[NOUT];                                     // the out-buffer's element tid + 256 q: the same offset in LDS and in gLa_t
#pragma unroll
  for (int q = 0; q < NOUT; q++) { const int e = tid + NT * q; oo[q] = 8u * (unsigned)(e < DD ? e : DD - 1); }      // (beyond the matrix: a duplicate of its last element)

  // The tables are all this role keeps across a step.  Left alone, the compiler hoists every address it can derive from them out of
  // the time loop and spills at the 168 registers of three waves per SIMD (and turns wave-uniform base + 32-bit offset into 64-bit
  // per-lane address pairs); made opaque once per step, nothing derived from them lives longer than the step.
  auto opaque_tables = [&]() {
    auto op = [](unsigned& v) { asm volatile("" : "+v"(v)); };
#pragma unroll
    for (int q = 0; q < NIT; q++) { op(it_rd[q]); op(qt_w0[q]); op(qt_w1[q]); }
#pragma unroll
    for (int q = 0; q < NPKI; q++) { op(s_w0[q]); op(s_w1[q]); op(s_g[q]); }
#pragma unroll
    for (int i = 0; i < 4; i++) op(bq_o[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) { op(bm_o[i]); op(fa_o[i]); op(eu_o[i]); }
#pragma unroll
    for (int i = 0; i < 5; i++) { op(fb_o[i]); op(em_o[i]); }
#pragma unroll
    for (int u = 0; u < 7; u++) op(eo_o[u]);
#pragma unroll
    for (int q = 0; q < NOUT; q++) op(oo[q]);
    op(m_w); op(blam); op(bu); op(lane8);
  };

  // ---- phases (SET: compile-time operand set) -------------------------------------------------------------------------------------
  double gsv[NPKI], gvm = 0.0, gvef = 0.0, gvam = 0.0, gvb = 0.0, gacc[7];
  // S_t (packed) and the vector entries of grid point tg, one step ahead of their use
  auto prefetch = [&](int tg) {
    const size_t o = (size_t)prob * Np + tg;
    const double* Sp = a.S + o * PK;
#pragma unroll
    for (int q = 0; q < NPKI; q++) gsv[q] = ldg(Sp, s_g[q]);
    gvm = ldg(a.m + o * D, lane8);
    gvef = ldg(a.Ef + o * D, lane8);
    gvam = ldg(a.Am + o * D, lane8);
    gvb = ldg(a.b + (size_t)prob * a.strideB + (size_t)tg * D, lane8);
  };
  auto settle_loads = [&]() {
#pragma unroll
    for (int q = 0; q < NPKI; q++) settle(gsv[q]);
    settle(gvm); settle(gvef); settle(gvam); settle(gvb);
  };
  // the operands of the grid point whose Psi_t is in stage buffer 0 and whose A_t is in R, and m_t
  auto build = [&](auto set_) {
    constexpr unsigned SB = decltype(set_)::value * SETB, SV = decltype(set_)::value * SETV;
    d2_t ps[NIT], a0[NIT];
#pragma unroll
    for (int q = 0; q < NIT; q++) { ps[q] = rd2(it_rd[q]); a0[q] = rd2(it_rd[q] + RB); }
#pragma unroll
    for (int q = 0; q < NPKI; q++) { wr1(s_w0[q] + SB, gsv[q]); wr1(s_w1[q] + SB, gsv[q]); }
    wr1(m_w + SV, gvm);
#pragma unroll
    for (int q = 0; q < NIT; q++) {
      wr1(qt_w0[q] + SB, __builtin_fma(hq, a0[q][0], ps[q][0]));      // Q / -2 = Psi - A / (2 sigma^2)
      wr1(qt_w1[q] + SB, __builtin_fma(hq, a0[q][1], ps[q][1]));
    }
  };
  // row i of <df/dx> / (-2 sigma^2) (lorenz_96.py:35-83; D >= 5: four distinct entries) is added to column i of the operand; dt u, gLb
  auto band_u = [&](auto set_, int tg) {
    constexpr unsigned SB = decltype(set_)::value * SETB, SV = decltype(set_)::value * SETV;
    if (wave == 0 && vl) {
      const double mm1 = rd1(bm_o[0] + SV), mp1 = rd1(bm_o[1] + SV), mm2 = rd1(bm_o[2] + SV), lam = rd1(blam);
      const double v0 = rd1(bq_o[0] + SB), v1 = rd1(bq_o[1] + SB), v2 = rd1(bq_o[2] + SB), v3 = rd1(bq_o[3] + SB);      // (all reads first: one LDS round trip)
      wr1(bq_o[0] + SB, v0 - hq);                              // <df/dx>_ii = -1
      wr1(bq_o[1] + SB, __builtin_fma(hq, mm1, v1));           // <df/dx>_i,i+1 = m_{i-1}
      wr1(bq_o[2] + SB, __builtin_fma(-hq, mm1, v2));          // <df/dx>_i,i-2 = -m_{i-1}
      wr1(bq_o[3] + SB, __builtin_fma(hq, mp1 - mm2, v3));     // <df/dx>_i,i-1 = m_{i+1} - m_{i-2}
      const double r = -gvef - gvam + gvb;                     // -<f> - A m + b        (variational.py:325-337)
      const double udt = dt * __builtin_fma(-2.0 * hq, r, lam);      // dt (dEsde_db + lam)
      wr1(bu + SV, udt);
      stg(gbase + (size_t)Np * DD + (size_t)tg * D, lane8, udt);
    }
  };
  // k-pairs [KP0, KP1) of the product from operand set SET; unit = (row-side map, column-side map), accumulators carried across the barriers
  auto prod = [&](auto set_, auto kp0_, auto kp1_) {
    constexpr unsigned SB = decltype(set_)::value * SETB;
    constexpr int KP0 = decltype(kp0_)::value, KP1 = decltype(kp1_)::value;
    if (KP0 == 0) {
#pragma unroll
      for (int u = 0; u < 7; u++) gacc[u] = 0.0;
    }
    if (wave < 3) {
#pragma unroll
      for (int kp = KP0; kp < KP1; kp++) {
        constexpr unsigned dummy = 0; (void)dummy;
        const unsigned ko = SB + 8u * (unsigned)(kp * 4 * LD);
        d2_t fa[2], fb[3];
#pragma unroll
        for (int i = 0; i < 2; i++) fa[i] = rd2(fa_o[i] + ko);
#pragma unroll
        for (int i = 0; i < 3; i++) fb[i] = rd2(fb_o[i] + ko);
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
#pragma unroll
          for (int ai = 0; ai < 2; ai++)
#pragma unroll
            for (int bi = 0; bi < 3; bi++) gacc[3 * ai + bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[ai][hh], fb[bi][hh], gacc[3 * ai + bi], 0, 0, 0);
        if (kp + 1 < KP1) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kp = KP0; kp < KP1; kp++) {
        const unsigned ko = SB + 8u * (unsigned)(kp * 4 * LD);
        d2_t fa[3], fb[5];
#pragma unroll
        for (int i = 0; i < 3; i++) fa[i] = rd2(fa_o[i] + ko);
#pragma unroll
        for (int i = 0; i < 5; i++) fb[i] = rd2(fb_o[i] + ko);
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
#pragma unroll
          for (int ai = 0; ai < 3; ai++) gacc[ai] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[ai][hh], fb[0][hh], gacc[ai], 0, 0, 0);
#pragma unroll
          for (int bi = 1; bi < 5; bi++) gacc[2 + bi] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[2][hh], fb[bi][hh], gacc[2 + bi], 0, 0, 0);
        }
        if (kp + 1 < KP1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // gLa = dt (Q S - u m^T) = -2 dt acc - (dt u) m^T of the point in set SET into the out-buffer
  auto epilogue = [&](auto set_) {
    constexpr unsigned SV = decltype(set_)::value * SETV;
    double uu[3], mm[5];
#pragma unroll
    for (int i = 0; i < 3; i++) uu[i] = rd1(eu_o[i] + SV);
#pragma unroll
    for (int i = 0; i < 5; i++) mm[i] = rd1(em_o[i] + SV);
    if (wave < 3) {
#pragma unroll
      for (int u = 0; u < 6; u++) wr1(eo_o[u], __builtin_fma(m2dt, gacc[u], -(uu[u / 3] * mm[u % 3])));
    } else {
#pragma unroll
      for (int u = 0; u < 7; u++) wr1(eo_o[u], __builtin_fma(m2dt, gacc[u], -(uu[u < 3 ? u : 2] * mm[u < 3 ? 0 : u - 2])));
    }
  };
  // the out-buffer -> gLa of grid point tg, whole lines
  auto out = [&](int tg) {
    double* gA = gbase + (size_t)tg * DD;
    double v[NOUT];
#pragma unroll
    for (int q = 0; q < NOUT; q++) v[q] = rd1(oo[q] + GB + 8u * (unsigned)gl::O);
#pragma unroll
    for (int q = 0; q < NOUT; q++) stg(gA, oo[q], v[q]);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

#ifndef VGPA_GF_ABL
#define VGPA_GF_ABL 0                    // DIAGNOSTIC (wrong results; tools/ubench/ode_gf_loop.hip only): bit 0 no product, 1 no band, 2 no build, 3 no out
#endif
#ifdef VGPA_STAMPS_ROLE
  long long ts_prev_ = clock64();
#define VGPA_GF_BARRIER(j)                                                                                               \
  { const long long tsb_ = clock64(); lds_barrier(); const long long tsa_ = clock64();                                   \
    if (lane == 0 && wave == 0 && blockIdx.x == 0) { mfma::g_stamp_role[2][2 * (j)] += tsb_ - ts_prev_; mfma::g_stamp_role[2][2 * (j) + 1] += tsa_ - tsb_; } \
    ts_prev_ = tsa_; }
#else
#define VGPA_GF_BARRIER(j) lds_barrier()
#endif
#ifndef VGPA_GF_PRIO_G
#define VGPA_GF_PRIO_G 3                 // (this role's few instructions first: 4.87 -> 4.60 ms per sweep direction of one problem)
#endif
  // one backward step: HP / HP2 = a previous / a second-previous grid point exists (compile time: the first two steps are peeled)
  auto step = [&](int k, auto set_, auto hp_, auto hp2_) {
    constexpr int SET = decltype(set_)::value;
    constexpr bool HP = decltype(hp_)::value, HP2 = decltype(hp2_)::value;
    using CUR = std::integral_constant<int, SET>; using PRV = std::integral_constant<int, SET ^ 1>;
#ifndef VGPA_GF_OUT_TICK
#define VGPA_GF_OUT_TICK 0
#endif
#ifndef VGPA_GF_PRIO_LOW
#define VGPA_GF_PRIO_LOW VGPA_GF_PRIO_G
#endif
    opaque_tables();
    if (VGPA_GF_PRIO_LOW != VGPA_GF_PRIO_G) __builtin_amdgcn_s_setprio(VGPA_GF_PRIO_G);
    if (HP) settle_loads();              // (the one place that waits for this role's HBM loads)
    // (scheduling barriers between the phases: interleaved, their operands do not fit the 168 registers of three waves per SIMD)
    if (VGPA_GF_OUT_TICK == 0 && HP2 && !(VGPA_GF_ABL & 8)) out(tidx(k - 2));
    __builtin_amdgcn_sched_barrier(0);
    if (!(VGPA_GF_ABL & 4)) build(CUR{});
    __builtin_amdgcn_sched_barrier(0);
    if (HP && !(VGPA_GF_ABL & 1)) prod(PRV{}, I0{}, I1{});
    VGPA_GF_BARRIER(0);
    if (VGPA_GF_PRIO_LOW != VGPA_GF_PRIO_G) __builtin_amdgcn_s_setprio(VGPA_GF_PRIO_LOW);
    if (!(VGPA_GF_ABL & 2)) band_u(CUR{}, tidx(k));
    prefetch(tidx(k + 1));
    __builtin_amdgcn_sched_barrier(0);
    if (HP && !(VGPA_GF_ABL & 1)) prod(PRV{}, I1{}, I2{});
    VGPA_GF_BARRIER(1);
    if (VGPA_GF_OUT_TICK == 2 && HP2 && !(VGPA_GF_ABL & 8)) { out(tidx(k - 2)); __builtin_amdgcn_sched_barrier(0); }
    if (HP && !(VGPA_GF_ABL & 1)) prod(PRV{}, I2{}, I4{});
    VGPA_GF_BARRIER(2);
    if (HP) { if (!(VGPA_GF_ABL & 1)) prod(PRV{}, I4{}, I5{}); __builtin_amdgcn_sched_barrier(0); epilogue(PRV{}); }
    VGPA_GF_BARRIER(3);
  };
  // behind the loop: the last grid point (Psi in stage buffer 0, the end point's operand in R, lam from the helpers' last vector update)
  auto drain = [&](auto set_) {
    constexpr int SET = decltype(set_)::value;
    using CUR = std::integral_constant<int, SET>; using PRV = std::integral_constant<int, SET ^ 1>;
    if (n_steps > 0) settle_loads();
    if (n_steps > 1) out(tidx(n_steps - 2));
    build(CUR{});
    if (n_steps > 0) prod(PRV{}, I0{}, I5{});
    lds_barrier();
    if (n_steps > 0) epilogue(PRV{});    // (every thread has read the out-buffer)
    band_u(CUR{}, tidx(n_steps));
    lds_barrier();
    if (n_steps > 0) out(tidx(n_steps - 1));
    prod(CUR{}, I0{}, I5{});
    lds_barrier();
    epilogue(CUR{});
    lds_barrier();
    out(tidx(n_steps));
  };

  __builtin_amdgcn_s_setprio(VGPA_GF_PRIO_G);
  __syncthreads();                       // LDS zero-filled (by the other roles)
  prefetch(tidx(0));
  settle_loads();
  __syncthreads();                       // prologue published
  using F = std::false_type; using T = std::true_type;
  int k = 0;
  if (n_steps > 0) { step(0, I0{}, F{}, F{}); k = 1; }
  if (n_steps > 1) { step(1, I1{}, T{}, F{}); k = 2; }
  for (; k + 1 < n_steps; k += 2) { step(k, I0{}, T{}, T{}); step(k + 1, I1{}, T{}, T{}); }
  if (k < n_steps) step(k, I0{}, T{}, T{});
  if (n_steps & 1) drain(I1{});
  else drain(I0{});
}

// H2 (round 5; helper-wave kernels without GF): TWO helper roles -- 768 threads, three waves per SIMD.  A helper's stage is a chain of
// dependent LDS round trips (partial sums -> stage vector -> partial products; start-point operand -> mid-point; stage state -> HBM)
// about as long as the product waves' stage, and the product waves wait for it in stages 0 and 1 (tools/ubench/ode_gf_loop.hip).  The
// vector recursion (role 1) and the operand staging / state stores / operand loads (role 2) share nothing but the barriers: side by
// side each chain is shorter than the products.
template <int METHOD, bool FWD, int NB, bool DENSEJ, int GR, int WPE, bool QOUT = false, int NW = 4, bool HLP = false, bool GF = false, bool H2 = false>   // WPE: waves per SIMD the register budget allows for
__global__ void __attribute__((amdgpu_flat_work_group_size(64 * NW * ((GF || H2) ? 3 : (HLP ? 2 : 1)), 64 * NW * ((GF || H2) ? 3 : (HLP ? 2 : 1))), amdgpu_waves_per_eu(WPE, WPE))) k_ode_sym(OdeArgs a) {
#pragma clang fp contract(fast)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  using g = SGeo<NB, NW>;
  constexpr bool W8 = (NW == 8);
  constexpr int NT = g::NT;
  constexpr int NS = n_stages<METHOD>(), NR = g::NR, MAXS = W8 ? 2 : g::MAXS, LD = g::LD, NKP = g::NKP;
  constexpr bool COVER = (GR <= 0);      // fragment cover (NSB = 5) instead of runs: see kCoverMaps
  constexpr bool OPC = (GR == -1);       // ... the outer-product cover (two maps per wave, rotated row sides): see kOpMaps
  constexpr bool LOOP1 = COVER && (OPC || VGPA_SYM_LOOP1) && !W8;      // single-chain loop units: see kCoverLoopSlot
  constexpr int LSLOT = OPC ? kOpLoopSlot : kCoverLoopSlot;
  static_assert(!OPC || !W8, "the outer-product cover runs on four waves");
  static_assert(!COVER || (g::NSB == 5 && g::MAXS == 4), "the fragment cover is built for 33 <= D <= 40");
  static_assert(!W8 || COVER, "eight waves per problem: fragment-cover kernels only");
  static_assert(!HLP || (COVER && NW == 4), "helper waves: fragment-cover kernels on four product waves");
  static_assert(!GF || (HLP && QOUT && !FWD && !DENSEJ && METHOD == VGPA_ODE_RK4), "fused gradient assembly: backward RK4 helper-wave kernels with Q'' on");
  static_assert(!H2 || (HLP && !GF), "two helper roles: helper-wave kernels without the gradient waves");
  constexpr int NITS = FWD ? g::NITF : g::NIT;     // staging items per thread
  constexpr int JSEC = NS > 1 ? 1 : 0;
  // the stage whose chores end with the requests for A_{t+2}, c_{t+2} (consumed at the top of the next step).  Helper-wave RK4 kernels:
  // the last stage -- same box, one problem: 7.82 -> 7.39 ms per sweep (forward 3.74 -> 3.43, backward 3.99 -> 3.87), the fused
  // backward kernel 9.77 -> 9.6 ms per 512 problems; the four-wave kernels (two workgroups per CU) measured no difference and keep JSEC
#ifndef VGPA_SYM_LOADA_STAGE
#define VGPA_SYM_LOADA_STAGE 3
#endif
  constexpr int LSTG = (NS == 4 && HLP) ? VGPA_SYM_LOADA_STAGE : JSEC;
  constexpr double sixth = 1.0 / 6.0;
  const int role = HLP ? (int)threadIdx.x / (64 * NW) : 0;     // (wave-uniform) 0: products; 1: helper -- the chores of product wave `wave` (H2: its vector recursion); 2: gradient assembly (GF) / operand staging and state stores (H2)
  const bool helper = role >= 1;
  const int tid = (int)threadIdx.x - role * 64 * NW, lane = tid & 63, wave = tid >> 6;
  const int wq = wave & 3, half = W8 ? (wave >> 2) : 0;      // cover wave whose units this wave multiplies; which half of them (W8)
  const int prob = (int)blockIdx.x;
  const int D = a.D, Np = a.Np, DD = a.D * a.D, n_steps = a.Np - 1;
  const double dt = a.dt, h = 0.5 * a.dt;
  double* const Xb0 = smem;
  double* const Xb1 = Xb0 + g::XS;
  double* const Rb = Xb1 + g::XS;
  double* const Mb = Rb + g::XS;
  double* const xvw = Mb + g::XS + wave * g::XV;           // this wave's copy of the stage vector
  double* const pvb = Mb + g::XS + NW * g::XV;             // [2][NPART][PP] partial inner products
  double* const trash = pvb + 2 * g::PV + 2 * tid;         // one 16-byte unit per thread
  // GF: the gradient waves (third set of four; grad_waves below) share nothing with this code but the barriers, the buffers they read
  // and lam_t, which the helpers leave for them
  using gl = GradLds<NB>;
  constexpr int GLDS = GF ? gl::DOUBLES : 0;
  if constexpr (GF) {
    if (role == 2) { grad_waves<NB>(a, smem, (int)threadIdx.x - 2 * 64 * NW); return; }
  }
  double* const gLam = smem + g::LDS_DOUBLES + gl::LAM;
  for (int i = tid; i < (int)g::LDS_DOUBLES + GLDS; i += NT) smem[i] = 0.0;
  // QOUT (backward, mid-point methods, Sigma = sigma^2 I): the state store writes Q''_t = A_t / sigma^2 - 2 Psi_t in place of Psi_t --
  // the only combination of A_t and Psi_t the gradient assembly reads (assemble.hip), which then streams one matrix less.  (A
  // general diagonal would need its entries per row pair here: with them the kernel no longer fits its 256 registers.)
  static_assert(!QOUT || (!FWD && (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4)), "Q'' is stored by the backward mid-point kernels");
  auto tidx =[&](int i) { return FWD ? i : Np - 1 - i; };
  auto tclamp = [&](int i) { return tidx(i <= n_steps ? i : n_steps); };

  // ---- unit slots of this lane ------------------------------------------------------------------------------------------
  const int r4 = lane >> 4, bq = (lane >> 2) & 3, c4 = lane & 3, bi = bq >> 1, bj = bq & 1;
  int colI[NR], colJ[MAXS], offD[MAXS], offM[MAXS];
  int colm[4] = {0, 0, 0, 0};            // cover: LDS column offset of this lane's fragment element, per map
  unsigned gofs[MAXS];
  bool own[MAXS], wd[MAXS], wm[MAXS];
  const int cov_used = OPC ? 0xF : (COVER ? kCoverUsed[wq] : 0);
  const bool loop_diag = r4 == c4;                     // diagonal element of a diagonal block (loop unit)
  const int loop_src = 16 * c4 + 4 * bq + r4;          // the lane that holds element (c4, r4) of the same block
  if constexpr (COVER) {
#pragma unroll
    for (int m = 0; m < 4; m++) colm[m] = 2 * ((4 * (OPC ? kOpMaps[wq][m & 1][bq] : kCoverMaps[wq][m][bq]) + c4) ^ r4);
#pragma unroll
    for (int s = 0; s < MAXS; s++) {
      const int so = W8 ? 2 * half + s : s;                       // unit slot of the cover wave
      const int Ib = OPC ? kOpMaps[wq][kOpUnits[so].rmap][(bq - kOpUnits[so].rot) & 3] : kCoverMaps[wq][kCoverPat[so][0]][bq];
      const int Jb = OPC ? kOpMaps[wq][kOpUnits[so].cmap][bq] : kCoverMaps[wq][kCoverPat[so][1]][bq];
      const int row = 4 * Ib + r4, col = 4 * Jb + c4;
      const bool first = ((OPC ? kOpOwner.m[wq][so] : kCoverOwner.m[wq][so]) >> bq) & 1u;      // (a compile-time table, looked up with the run-time wave / block)
      const bool act = ((cov_used >> so) & 1) && first && (Ib != Jb || row <= col);     // diagonal blocks: the upper half represents
      offD[s] = elem_off<NB>(row, col);
      offM[s] = elem_off<NB>(col, row);
      wd[s] = act;
      wm[s] = act && row != col;
      own[s] = act && row < D && col < D;
      gofs[s] = own[s] ? 8u * (unsigned)(row <= col ? row * D + col : col * D + row) : 0u;     // (the upper triangle: all that EnergyArgs::ds_upper writes)
      if (!FWD && a.ds_packed) gofs[s] = own[s] ? 8u * (unsigned)(row <= col ? tri_off(col) + row : tri_off(row) + col) : 0u;      // (packed lower triangle, EnergyArgs::ds_packed)
      colJ[s] = 0;
    }
#pragma unroll
    for (int rl = 0; rl < NR; rl++) colI[rl] = 0;
  } else {
#pragma unroll
  for (int rl = 0; rl < NR; rl++) {
    const Run run = sym_run(g::NSB, wave + 4 * rl, nullptr);
    const bool rv = run.c >= 0;
    const int c = rv ? run.c : 0;
    const int Ib = 2 * c + bi;
    colI[rl] = 2 * ((4 * Ib + c4) ^ r4);
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int s = 2 * rl + t;
      const int jr = t ? run.j1 : run.j0;
      const bool valid = rv && jr >= 0;
      const int j = valid ? jr : 0;
      const int Jb = 2 * j + bj;
      const int row = 4 * Ib + r4, col = 4 * Jb + c4;
      colJ[s] = 2 * ((4 * Jb + c4) ^ r4);
      offD[s] = elem_off<NB>(row, col);
      offM[s] = elem_off<NB>(col, row);
      const bool act = valid && (c != j || row <= col);    // diagonal super-blocks: the upper half represents
      wd[s] = act;
      wm[s] = act && row != col;
      own[s] = act && row < D && col < D;
      gofs[s] = own[s] ? 8u * (unsigned)(row <= col ? row * D + col : col * D + row) : 0u;     // (the upper triangle: all that EnergyArgs::ds_upper writes)
      if (!FWD && a.ds_packed) gofs[s] = own[s] ? 8u * (unsigned)(row <= col ? tri_off(col) + row : tri_off(row) + col) : 0u;      // (packed lower triangle, EnergyArgs::ds_packed)
    }
  }
  }
  const bool spk = FWD && a.s_packed;                       // S_t leaves as its packed lower triangle (OdeArgs::s_packed)
  const int MS = spk ? D * (D + 1) / 2 : DD;                // doubles per stored matrix
  ItemTab<NB, NW> IT;
  build_items<NB, NW>(D, tid, IT, spk);
  TileTab<NB, NW> TT;
  if (FWD) build_tiles<NB, NW>(D, wave, lane, TT);

  // ---- operand staging --------------------------------------------------------------------------------------------------
  const double* A = a.A + (size_t)prob * a.strideA;
  d2_t an[NITS];
#ifndef VGPA_SYM_LOADA_NT
#define VGPA_SYM_LOADA_NT 0
#endif
  auto load_a = [&](const double* At) {
#pragma unroll
    for (int q = 0; q < NITS; q++) {
      if (VGPA_SYM_LOADA_NT) {
        an[q][0] = __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(At) + (FWD ? TT.g0[q] : IT.g0[q])));
        an[q][1] = __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(At) + (FWD ? TT.g1[q] : IT.g1[q])));
      } else {
      an[q][0] = ldg(At, FWD ? TT.g0[q] : IT.g0[q]);
      an[q][1] = ldg(At, FWD ? TT.g1[q] : IT.g1[q]);
      }
    }
  };
  auto unit_ptr = [&](double* buf, int q) -> d2_t* {
    const int lo = FWD ? TT.lo[q] : IT.lo[q];
    return reinterpret_cast<d2_t*>(lo >= 0 ? buf + lo : trash);
  };
  auto store_a = [&](double* buf, const d2_t (&v)[NITS]) {
#pragma unroll
    for (int q = 0; q < NITS; q++) {
      d2_t o = v[q];
      if (!(FWD ? TT.two[q] : IT.two[q])) o[1] = 0.0;
      *unit_ptr(buf, q) = o;
    }
  };
  // ---- matrix state -----------------------------------------------------------------------------------------------------
  const int GS = (!FWD && a.ds_packed) ? D * (D + 1) / 2 : DD;      // doubles per matrix of the forcing-term stream
  const double* G = FWD ? a.Sigma : a.dEs + (size_t)prob * Np * GS;
  double* const mout = (FWD ? a.S : a.psi) + (size_t)prob * Np * (spk ? D * (D + 1) / 2 : DD);
  double xk[MAXS], acc[MAXS], fc[MAXS], fn[MAXS], fnn[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; s++) {
    fc[s] = ldg(G + (FWD ? 0 : (size_t)tidx(0) * GS), gofs[s]);
    fn[s] = FWD ? 0.0 : ldg(G + (size_t)tclamp(1) * GS, gofs[s]);
    fnn[s] = 0.0;
    xk[s] = (FWD && own[s]) ? ldg(a.S0, gofs[s]) : 0.0;
    acc[s] = 0.0;
  }

  // ---- vector state: every wave keeps the whole vector in its lanes < D (the other lanes compute along on element 0) ----
  const bool vl = lane < D;
  const unsigned lane8 = vl ? 8u * (unsigned)lane : 0u;
  double* const xv_mine = vl ? xvw + lane : trash;
  const double* cin = FWD ? a.b + (size_t)prob * a.strideB : a.dEm + (size_t)prob * Np * D;
  double* const vout = (FWD ? a.m : a.lam) + (size_t)prob * Np * D;
  auto vec = [&](int t) { return (size_t)t * D; };
  const bool sparse_j = !FWD && !DENSEJ && a.obs_idx && a.jm_sparse;
  // vector jump behind the step that ends at grid point t (n_obs: its observation index or -1)
  auto jump_vector = [&](int t, int n_obs) -> double {
    if (!FWD && DENSEJ) return ldg(a.jm_dense + ((size_t)prob * Np + t) * D, lane8);
    if (sparse_j && __builtin_amdgcn_readfirstlane(n_obs) >= 0) return ldg(a.jm_sparse + ((size_t)prob * a.n_obs + n_obs) * D, lane8);
    return 0.0;
  };
  double vk = FWD ? ldg(a.m0, lane8) : 0.0, v1 = 0.0;      // v1: Heun's first slope / RK4's running sum k1 + 2 k2 + 2 k3
  double c0 = ldg(cin + vec(tidx(0)), lane8), c1 = ldg(cin + vec(tclamp(1)), lane8), c2 = 0.0;
  int n_obs_cur = sparse_j ? ldu(a.obs_idx, tclamp(1)) : -1, n_obs_next = sparse_j ? ldu(a.obs_idx, tclamp(2)) : -1, n_obs_nn = -1;
  double jm = n_steps >= 1 ? jump_vector(tidx(1), n_obs_cur) : 0.0, jm_next = 0.0;

  __syncthreads();                       // LDS zero-filled
#pragma unroll
  for (int s = 0; s < MAXS; s++) {       // first stage state
    if (FWD) {
      *(wd[s] ? Xb0 + offD[s] : trash) = xk[s];
      *(wm[s] ? Xb0 + offM[s] : trash) = xk[s];
    }
  }
  *xv_mine = vk;
  load_a(A + (size_t)tidx(0) * DD);
  store_a(Rb, an);
  load_a(A + (size_t)tclamp(1) * DD);
#pragma unroll
  for (int s = 0; s < MAXS; s++) { settle(fc[s]); settle(fn[s]); settle(xk[s]); }
  settle(c0); settle(c1); settle(vk); settle(jm);
#pragma unroll
  for (int q = 0; q < NITS; q++) settle(an[q]);
  __syncthreads();                       // prologue published

  // S_k / Psi_t and m_k / lam_t of grid point t to HBM: the matrix from the stage buffer that holds it
  auto store_vector = [&](int t) {
    if (VGPA_ABL_NOSTORE && t > 1) return;
    if (wave == 0 && vl) stg(vout + vec(t), lane8, vk);
  };
  auto store_items = [&](const d2_t (&v)[g::NIT], int t, bool with_vector = true) {
    if (VGPA_ABL_NOSTORE && t > 1) return;
    double* so = mout + (size_t)t * MS;
#pragma unroll
    for (int q = 0; q < g::NIT; q++) {
      if (IT.st0[q]) stg(so, IT.g0[q], v[q][0]);
      if (IT.st1[q]) stg(so, IT.g1[q], v[q][1]);
    }
    if (with_vector) store_vector(t);
  };
  auto load_items = [&](const double* Xc, d2_t (&v)[g::NIT]) {
#pragma unroll
    for (int q = 0; q < g::NIT; q++) v[q] = *reinterpret_cast<const d2_t*>(IT.lo[q] >= 0 ? Xc + IT.lo[q] : trash);
  };

  // every HBM load of a step, issued together (see "Memory waits"): what the step after the next one needs
  auto prefetch = [&](int step, auto helper_role) {
#ifndef VGPA_ABL_NOLOAD
#define VGPA_ABL_NOLOAD 0                // DIAGNOSTIC (wrong results; ubench only): behind the second step no loads of 1: A_t, c_t  2: G_t  4: jumps  8: obs index
#endif
    const bool late = step > 1;
    // role kind: 0 = product waves beside helpers (G_t), 1 = every chore (the one helper role; without helpers: everything),
    //            2 = the vector recursion's helper (H2), 3 = the staging helper (H2)
    constexpr int HK = decltype(helper_role)::value;
    constexpr bool ch_a = !HLP || HK == 1 || HK == 3, ch_v = !HLP || HK == 1 || HK == 2, units = !HLP || HK == 0;
    if (!((VGPA_ABL_NOLOAD & 1) && late) && LSTG == JSEC) {
      if (ch_a) load_a(A + (size_t)tclamp(step + 2) * DD);
      if (ch_v) c2 = ldg(cin + vec(tclamp(step + 2)), lane8);
    }
    if (!FWD) {
      if (units && !((VGPA_ABL_NOLOAD & 2) && late)) {
#pragma unroll
        for (int s = 0; s < MAXS; s++) fnn[s] = ldg(G + (size_t)tclamp(step + 2) * GS, gofs[s]);
      }
      if (ch_v && !((VGPA_ABL_NOLOAD & 4) && late)) jm_next = step + 2 <= n_steps ? jump_vector(tidx(step + 2), n_obs_next) : 0.0;
      if (!((VGPA_ABL_NOLOAD & 8) && late)) n_obs_nn = (sparse_j && step + 3 <= n_steps) ? ldu(a.obs_idx, tidx(step + 3)) : -1;
    }
  };

  // Everything of stage j that is not the matrix product, as ONE LDS round trip: all reads (operand and stage-vector entries
  // of the vector's partial inner products; at stage 0 the start-point operand for the mid-point and the stage state for
  // HBM), then the arithmetic, then all LDS writes, then the HBM stores and -- at stage JSEC -- the step's HBM loads.  (An LDS
  // read cannot be moved above an LDS write by the compiler, so read-compute-write pieces in sequence cost one round trip
  // each.)  Partial inner products: forward sum_k Aop[k][i] v[k], lane = (i, part of the row pairs); backward
  // sum_k Aop[i][k] v[k], lane = (row pair ip, part of the 4-column blocks, rotated by ip / 4: conflict-free).
  // Split-phase chores: every piece of a stage that is not the product is a READ phase (LDS / HBM requests only) and a FINISH
  // phase (arithmetic, LDS writes, HBM stores) that the time loop places one pipeline step apart -- the requests are in flight
  // under a step of products, the finish finds its operands there.  State that lives between the two phases:
  constexpr int NAV = FWD ? g::RPP : 4 * g::CBP;
  constexpr int NPS = FWD ? g::NPF : g::NPARTB;
  d2_t tb_av[NAV], tb_xqf[FWD ? g::RPP : 1];
  double tb_xqb[FWD ? 1 : 4 * g::CBP];
  d2_t tc_mid[NITS], tc_items[g::NIT];
  double ta_ps[NPS];
  const int pdiv = FWD ? g::PP : g::RP;
  const int tb_part0 = tid / pdiv, tb_idx = tid - tb_part0 * pdiv;      // forward: idx = column i; backward: idx = row pair ip
  const bool tb_act = tb_part0 < (FWD ? g::NPF : g::NPARTB);
  const int tb_part = tb_act ? tb_part0 : 0;
  // B: partial inner products of the vector recursion -> pv
  auto tailB_read = [&](const double* Aop) {
    if (FWD) {
#pragma unroll
      for (int r = 0; r < g::RPP; r++) {
        const int rp0 = tb_part * g::RPP + r;
        const int rp = rp0 < g::RP ? rp0 : 0;
        tb_av[r] = *reinterpret_cast<const d2_t*>(Aop + unit_off<NB>(rp, tb_idx));
        tb_xqf[r] = *reinterpret_cast<const d2_t*>(xvw + 2 * rp);
      }
    } else {
#pragma unroll
      for (int q = 0; q < g::CBP; q++) {
        const int cbq = tb_part + q * g::NPARTB;
        int cb = (cbq < g::NCB ? cbq : 0) + (tb_idx >> 2);
        if (cb >= g::NCB) cb -= g::NCB;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          tb_av[4 * q + kk] = *reinterpret_cast<const d2_t*>(Aop + tb_idx * LD + 2 * (4 * cb + (kk ^ (tb_idx & 3))));   // column 4 cb + kk
          tb_xqb[4 * q + kk] = xvw[4 * cb + kk];
        }
      }
    }
  };
  auto tailB_finish = [&](double* pv) {
    double s0 = 0.0, s1 = 0.0;
    if (FWD) {
#pragma unroll
      for (int r = 0; r < g::RPP; r++) {
        const bool in = tb_part * g::RPP + r < g::RP;
        s0 = __builtin_fma(tb_av[r][0], in ? tb_xqf[r][0] : 0.0, s0);
        s0 = __builtin_fma(tb_av[r][1], in ? tb_xqf[r][1] : 0.0, s0);
      }
      *(tb_act ? pv + tb_part * g::PP + tb_idx : trash) = s0;
    } else {
#pragma unroll
      for (int q = 0; q < g::CBP; q++) {
        const bool in = tb_part + q * g::NPARTB < g::NCB;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const double xv = in ? tb_xqb[4 * q + kk] : 0.0;
          s0 = __builtin_fma(tb_av[4 * q + kk][0], xv, s0);
          s1 = __builtin_fma(tb_av[4 * q + kk][1], xv, s1);
        }
      }
      d2_t o; o[0] = s0; o[1] = s1;
      *reinterpret_cast<d2_t*>(tb_act ? pv + tb_part * g::PP + 2 * tb_idx : trash) = o;
    }
  };
  // C: operand staging (see mfma::stage_op -- M <- mid-point / end point at stage 0, R <- end point at JSEC), the stage state of
  // grid point `step` to HBM (stage 0) and the step's HBM loads (stage JSEC).  None of the buffers written here is read by
  // anyone during stage j, none of the values read is written during stage j: free to run anywhere inside the stage.
  constexpr bool MIDP = (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4);
  auto tailC_read = [&](int j, const double* Xc) {
    if (j == 0) {
      if (MIDP) {
#pragma unroll
        for (int q = 0; q < NITS; q++) tc_mid[q] = *unit_ptr(Rb, q);
      }
      if constexpr (!GF) load_items(Xc, tc_items);      // (GF: nothing of the state leaves through the helpers)
    }
  };
  // QOUT: items of Psi_t -> items of Q''_t, with the start-point operand A_t in the same row-pair items
  auto to_q = [&](d2_t (&items)[g::NIT], const d2_t (&a0)[NITS]) {
    if constexpr (QOUT) {
      const double qs = a.q_scale;
#pragma unroll
      for (int q = 0; q < g::NIT; q++) {
        items[q][0] = __builtin_fma(qs, a0[q][0], -2.0 * items[q][0]);
        items[q][1] = __builtin_fma(qs, a0[q][1], -2.0 * items[q][1]);
      }
    }
  };
  using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
  // kind_: 1 = all of it (the one helper role / no helpers), 3 = the staging helper of H2 (the vector's store and loads are role 2's)
  auto tailC_finish = [&](int j, int step, auto kind_) {
    constexpr int KIND = decltype(kind_)::value;
    if (j == 0) {
      double* dst = (METHOD == VGPA_ODE_EULER && (step & 1)) ? Rb : Mb;
      if constexpr (QOUT && !GF) to_q(tc_items, tc_mid);
      if (MIDP) {
#pragma unroll
        for (int q = 0; q < NITS; q++) { tc_mid[q][0] = 0.5 * (tc_mid[q][0] + an[q][0]); tc_mid[q][1] = 0.5 * (tc_mid[q][1] + an[q][1]); }
        store_a(dst, tc_mid);
      } else {
        store_a(dst, an);
      }
      if constexpr (GF) {                 // (Psi_t / Q''_t stay in the kernel; lam_t also goes to the gradient waves)
        if (wave == 0 && vl) { stg(vout + vec(tidx(step)), lane8, vk); gLam[lane] = vk; }
      } else if (!(KIND == 3 && NS > 1)) store_items(tc_items, tidx(step), KIND == 1);
    }
    // (H2's staging helper is the longest chain of stage 0 -- units in, mid-point, operand out, state out -- and nearly idle in stages 1
    //  and 2: the state's way to HBM, which only needs the registers read at stage 0, waits for stage 2 (stage 1 has the end-point
    //  operand's staging).  One problem, stamps: stage 0 2 860 -> 2 370 cycles for this role against 2 320 - 2 420 for the product waves;
    //  one sweep direction 3.64 -> 3.5 ms backward, 3.39 -> 3.27 forward.)
    if (KIND == 3 && NS > 1 && j == (NS > 2 ? 2 : 1)) store_items(tc_items, tidx(step), false);
    if (j == JSEC && NS > 1) store_a(Rb, an);
    if (j == JSEC) prefetch(step, kind_);                  // (overwrites an[]: behind its last use of the step; with helper waves: their part)
    if (LSTG != JSEC && j == LSTG) {     // (helper-wave RK4 kernels: the next operand's loads behind the LAST stage's chores)
      load_a(A + (size_t)tclamp(step + 2) * DD);
      if (KIND == 1) c2 = ldg(cin + vec(tclamp(step + 2)), lane8);
    }
  };
  // H2, role 1: what tailC does for the vector -- its way to HBM (stage 0) and its HBM loads
  auto vec_chores = [&](int j, int step) {
    if (j == 0) store_vector(tidx(step));
    if (j == JSEC) prefetch(step, K2{});
    if (LSTG != JSEC && j == LSTG) c2 = ldg(cin + vec(tclamp(step + 2)), lane8);
  };

  // behind the barrier of stage j: every wave sums the partial products and advances its copy of the vector
  auto vecA_read = [&](const double* pv) {
    const double* pl = pv + (vl ? lane : 0);
#pragma unroll
    for (int q = 0; q < NPS; q++) ta_ps[q] = pl[q * g::PP];
  };
  auto vecA_finish = [&](int j) {
    double vs = ta_ps[0];
#pragma unroll
    for (int q = 1; q < NPS; q++) vs += ta_ps[q];
    const double cmid = 0.5 * (c0 + c1);
    double vn;
    if (METHOD == VGPA_ODE_EULER) { vk = vk + (c0 - vs) * dt + jm; vn = vk; }
    else if (METHOD == VGPA_ODE_HEUN) {
      if (j == 0) { v1 = c0 - vs; vn = vk + v1 * dt; }
      else { vk = vk + h * (v1 + (c1 - vs)) + jm; vn = vk; }
    } else if (METHOD == VGPA_ODE_RK2) {
      if (j == 0) vn = vk + h * (c0 - vs);
      else { vk = vk + dt * (cmid - vs) + jm; vn = vk; }
    } else {
      if (j == 0) { v1 = c0 - vs; vn = vk + h * v1; }
      else if (j == 1) { const double f = cmid - vs; v1 = v1 + 2.0 * f; vn = vk + h * f; }
      else if (j == 2) { const double f = cmid - vs; v1 = v1 + 2.0 * f; vn = vk + dt * f; }
      else { vk = vk + (dt * (v1 + (c1 - vs))) * sixth + jm; vn = vk; }
    }
    *xv_mine = vn;
  };

  // the stepper on one owned element: slope f = -z of stage j; returns the next stage state
  auto element = [&](int s, int j, double z, double js) -> double {
    const double f = -z;
    double xn;
    if (METHOD == VGPA_ODE_EULER) { xk[s] = xk[s] + f * dt + js; xn = xk[s]; }
    else if (METHOD == VGPA_ODE_HEUN) {
      if (j == 0) { acc[s] = f; xn = xk[s] + f * dt; }
      else { xk[s] = xk[s] + h * (acc[s] + f) + js; xn = xk[s]; }
    } else if (METHOD == VGPA_ODE_RK2) {
      if (j == 0) xn = xk[s] + h * f;
      else { xk[s] = xk[s] + dt * f + js; xn = xk[s]; }
    } else {                                            // acc = k1 + 2 k2 + 2 k3 + k4, summed as the stages come
      if (j == 0) { acc[s] = f; xn = xk[s] + h * f; }
      else if (j == 1) { acc[s] = acc[s] + 2.0 * f; xn = xk[s] + h * f; }
      else if (j == 2) { acc[s] = acc[s] + 2.0 * f; xn = xk[s] + dt * f; }
      else { xk[s] = xk[s] + (dt * (acc[s] + f)) * sixth + js; xn = xk[s]; }
    }
    return xn;
  };

  // ---- the product of a stage: NG groups of GR runs, NKP k-pairs each, as ONE software pipeline of NG * NKP steps -----------
  // Step t = (group, k-pair): the fragments of step t + 1 are requested before the products of step t issue -- also across
  // the group boundary (the next group's first fragments are on their way while this group's stepper runs) and across the
  // STAGE boundary: product_begin requests step 0 of the next stage right behind the barrier, in front of the vector work.
  constexpr int GRR = COVER ? 1 : GR;            // (array extents of the run layout; unused under the cover)
  constexpr int NG = COVER ? 1 : cdiv(NR, GRR), NSL = COVER ? (W8 ? 2 : 4) : 2 * GRR, NSTEP = NG * NKP;
  static_assert(COVER || NR % GRR == 0, "a group of runs must be complete (the clamped tail group is not parity-clean)");
  d2_t fa1[2][GRR], fa2[2][GRR], fb1[2][COVER ? 1 : NSL], fb2[2][COVER ? 1 : NSL];
  // cover: fragments of the row-side maps a0, a1 (ONE buffer each: a0 is dead behind the first half of a step, a1 is not needed
  // before the second) and of the column-side maps b0, b1 (two buffers), from the A operand (fA*) / the stage state (fX*)
  // (HALF, backward: 48 fragment registers.  The forward kernel, whose lanes carry no G_t / G_{t-1}, can afford two buffers of
  // all four maps -- 64 registers, one block of eight reads per step -- and measured faster that way: 8.06 vs 8.36 ms per
  // 512-problem launch; the backward kernel spilled 22 registers with it and ran at 9.6 ms instead of 8.95.)
#ifndef VGPA_SYM_HALF_FWD
#define VGPA_SYM_HALF_FWD 0
#endif
#ifndef VGPA_SYM_HALF_BWD
#define VGPA_SYM_HALF_BWD 1
#endif
  #ifndef VGPA_SYM_HALF_HLP
#define VGPA_SYM_HALF_HLP 0            // helper-wave kernels: the product waves have the registers for two buffers of all four maps, backward too
#endif
  constexpr bool HALF = COVER && !OPC && !W8 && (HLP ? VGPA_SYM_HALF_HLP : (FWD ? VGPA_SYM_HALF_FWD : VGPA_SYM_HALF_BWD));
  // W8: three maps per wave -- row side a0 (first half) or a1 (second half), column side b0, b1 -- of both operands, two buffers
  d2_t fA8[2][W8 ? 3 : 1], fX8[2][W8 ? 3 : 1];
  const int col8[3] = {half ? colm[1] : colm[0], colm[2], colm[3]};
  auto frag8 = [&](int buf, int kp, const double* pa, const double* px) {
#pragma unroll
    for (int m = 0; m < 3; m++) {
      fA8[buf][m] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + col8[m]);
      fX8[buf][m] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + col8[m]);
    }
  };
  d2_t fAa[COVER ? 2 : 1], fXa[COVER ? 2 : 1], fAb[2][COVER ? 2 : 1], fXb[2][COVER ? 2 : 1];
  d2_t fAa1[2], fXa1[2];                 // HALF + interleave: map a1 in two buffers (read half a step earlier than it is free)
  d2_t fA[2][(COVER && !OPC && !HALF && !W8) ? 4 : 1], fX[2][(COVER && !OPC && !HALF && !W8) ? 4 : 1];
  d2_t foA[2][2], foX[2][2];             // outer-product cover: maps P, Q of both operand buffers, two pipeline buffers
  auto frag_op = [&](int buf, int kp, const double* pa, const double* px) {
#pragma unroll
    for (int m = 0; m < 2; m++) {
      foA[buf][m] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + colm[m]);
      foX[buf][m] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + colm[m]);
    }
  };
  auto frag_all = [&](int buf, int kp, const double* pa, const double* px) {
#pragma unroll
    for (int m = 0; m < 4; m++) {
      fA[buf][m] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + colm[m]);
      fX[buf][m] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + colm[m]);
    }
  };
  auto frag_a = [&](int m, int kp, const double* pa, const double* px) {
    fAa[m] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + colm[m]);
    fXa[m] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + colm[m]);
  };
  auto frag_a1 = [&](int buf, int kp, const double* pa, const double* px) {
    fAa1[buf] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + colm[1]);
    fXa1[buf] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + colm[1]);
  };
  auto frag_b = [&](int buf, int kp, const double* pa, const double* px) {
#pragma unroll
    for (int m = 0; m < 2; m++) {
      fAb[buf][m] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + colm[2 + m]);
      fXb[buf][m] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + colm[2 + m]);
    }
  };
  auto frag_load = [&](int buf, int t, const double* pa, const double* px) {
    if constexpr (COVER) {
      // (only the first step of a stage comes through here: every other fragment is requested inside the step before it)
      const int kp = t % NKP;
      if constexpr (OPC) {
        frag_op(buf, kp, pa, px);
      } else if constexpr (W8) {
        frag8(buf, kp, pa, px);
      } else if constexpr (HALF) {
        frag_a(0, kp, pa, px);
        frag_b(buf, kp, pa, px);
        if (VGPA_SYM_INTERLEAVE) frag_a1(buf, kp, pa, px);
        else frag_a(1, kp, pa, px);
      } else {
        frag_all(buf, kp, pa, px);
      }
    } else {
      const int g0 = (t / NKP) * GRR, kp = t % NKP;
#pragma unroll
      for (int r = 0; r < GRR; r++) {
        const int rl = g0 + r < NR ? g0 + r : NR - 1;
        fa1[buf][r] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + colI[rl]);
        fa2[buf][r] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + colI[rl]);
      }
#pragma unroll
      for (int u = 0; u < NSL; u++) {
        const int sl = 2 * g0 + u < MAXS ? 2 * g0 + u : MAXS - 1;
        fb1[buf][u] = *reinterpret_cast<const d2_t*>(px + kp * 4 * LD + colJ[sl]);
        fb2[buf][u] = *reinterpret_cast<const d2_t*>(pa + kp * 4 * LD + colJ[sl]);
      }
    }
  };
  auto product_begin = [&](const double* Aop, const double* Xc) {
    frag_load(0, 0, Aop + r4 * LD, Xc + r4 * LD);
#if VGPA_ABL_NOFRAG
    if constexpr (COVER) frag_load(1, 1, Aop + r4 * LD, Xc + r4 * LD);
#endif
  };

  // products (their step-0 fragments are in flight) + stepper + publish of stage j
  auto product_stage = [&](int j, int step, const double* Aop, const double* Xc, double* Xn, auto&& chore) {
    const double* pa = Aop + r4 * LD;
    const double* px = Xc + r4 * LD;
    const bool last = (j == NS - 1);
    const bool jump_now = !FWD && last && !DENSEJ && __builtin_amdgcn_readfirstlane(n_obs_cur) >= 0;
    double jsd[MAXS];
    if (!FWD && last && DENSEJ) {        // dense matrix jump behind the step (euler.py:139-149): requested before the products
#pragma unroll
      for (int s = 0; s < MAXS; s++) jsd[s] = ldg(a.js_dense + ((size_t)prob * Np + tidx(step + 1)) * DD, gofs[s]);
    }
    if (!FWD && last && !DENSEJ) {       // the constant matrix jump 0.5 H^T R^-1 H, only behind a step that ends at an observation:
#pragma unroll                           // fetched (L2) when it is needed instead of riding in eight registers through every step
      for (int s = 0; s < MAXS; s++) jsd[s] = 0.0;
      if (jump_now && a.js_const) {
#pragma unroll
        for (int s = 0; s < MAXS; s++) jsd[s] = ldg(a.js_const, gofs[s]);
      }
    }
    double w[NSL];
    chore(-1);                           // beside the first fragments' way from LDS
#pragma unroll
    for (int t = 0; t < NSTEP; t++) {
      const int gi = t / NKP, kp = t % NKP, g0 = gi * GRR, cur = t & 1;
      if (!COVER && t + 1 < NSTEP) frag_load(cur ^ 1, t + 1, pa, px);
      if (kp == 0) {                     // accumulators start from minus the stage's forcing term
#pragma unroll
        for (int u = 0; u < NSL; u++) {
          const int s = 2 * g0 + u;
          double f = 0.0;
          if (s < MAXS) {
            if (FWD || METHOD == VGPA_ODE_EULER) f = fc[s];
            else if (METHOD == VGPA_ODE_HEUN) f = j == 0 ? fc[s] : fn[s];
            else if (METHOD == VGPA_ODE_RK2) f = j == 0 ? fc[s] : 0.5 * (fn[s] + fc[s]);
            else f = j == 0 ? fc[s] : (j == 3 ? fn[s] : 0.5 * (fn[s] + fc[s]));
            f = own[s] ? -f : 0.0;
            if (LOOP1 && s == LSLOT) f = loop_diag ? 0.5 * f : f;     // (the in-block transpose below adds the diagonal to itself)
          }
          w[u] = f;
        }
      }
      if constexpr (COVER) {
        // unit u = (row-side map u >> 1, column-side map 2 + (u & 1)): Aop^T X with the row side from the A operand and the
        // column side from the stage state, X^T Aop the other way round; the rectangle wave has no fourth unit
        // The wave is in-order and the register allocator, left alone, sinks the next step's fragment reads below this step's
        // products (shorter live ranges): read -> wait -> 16 products, every LDS round trip exposed (measured on a lone
        // workgroup: 2 680 cycles per stage against 1 280 of matrix-core work).  So the order is pinned, half step by half step:
        //   eight products of the units (a0, b0), (a0, b1)  |  reads: a0, b0, b1 of the NEXT step (a0's registers are free now)
        //   eight products of the units (a1, b0), (a1, b1)  |  reads: a1 of the next step
        // -- every read has at least half a step of products in front of its first use, and only the column-side fragments need
        // two buffers (48 fragment registers instead of 64).  Per unit the products come in the same order as before (A^T X
        // then X^T A of the first k-step of the pair, then of the second): same bits.  Straight-line code: a branch inside the
        // pipeline would end the scheduling region (the rectangle wave multiplies its unused fourth unit too; the triangle waves
        // read map a1 twice -- the LDS array is not what bounds this kernel, the exposed latency of its reads was).
        if constexpr (OPC) {
          // outer-product cover: the four reads of the next step between this step's fourteen products; the row sides of the
          // rotated units come out of the fragments by DPP moves (kOpUnits)
          // Pinned order (the scheduler, left alone, rotates every fragment at the top of the step -- behind a wait for the LAST
          // read of the step before): the loop unit and the unrotated unit first, a rotation right in front of the product that
          // needs it, the four reads of the next step behind the first four products.  Per unit the products keep the order
          // A^T X, X^T A of the first k-step, then of the second.
          static_assert(kOpUnits[0].rmap == 0 && kOpUnits[0].cmap == 0 && kOpUnits[1].rmap == 0 && kOpUnits[1].rot == 1 && kOpUnits[1].cmap == 0 &&
                        kOpUnits[2].rmap == 1 && kOpUnits[2].rot == 1 && kOpUnits[2].cmap == 1 && kOpUnits[3].rmap == 0 && kOpUnits[3].rot == 0 &&
                        kOpUnits[3].cmap == 1, "the product order below is written for the units (P,P) (rot1 P,P) (rot1 Q,Q) (P,Q)");
          const bool more = t + 1 < NSTEP && !VGPA_ABL_NOFRAG;
          const int kn = (t + 1) % NKP, nb = cur ^ 1;
          const d2_t PA = foA[cur][0], PX = foX[cur][0], QA = foA[cur][1], QX = foX[cur][1];
          auto SB = [] { if (VGPA_SYM_OP_PIN) __builtin_amdgcn_sched_barrier(0); };
          auto MF = [&](int u, double ra, double rb) { w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(ra, rb, w[u], 0, 0, 0); };
          double rPA[2], rPX[2], rQA[2], rQX[2];
          __builtin_amdgcn_sched_barrier(0);
          MF(0, PA[0], PX[0]); rPA[0] = rot_d<1>(PA[0]);
          if (more) foA[nb][0] = *reinterpret_cast<const d2_t*>(pa + kn * 4 * LD + colm[0]);
          SB();
          MF(0, PA[1], PX[1]); rPX[0] = rot_d<1>(PX[0]);
          if (more) foX[nb][0] = *reinterpret_cast<const d2_t*>(px + kn * 4 * LD + colm[0]);
          SB();
          MF(1, rPA[0], PX[0]); rPA[1] = rot_d<1>(PA[1]);
          if (more) foX[nb][1] = *reinterpret_cast<const d2_t*>(px + kn * 4 * LD + colm[1]);
          SB();
          MF(3, PA[0], QX[0]); rPX[1] = rot_d<1>(PX[1]);
          if (more) foA[nb][1] = *reinterpret_cast<const d2_t*>(pa + kn * 4 * LD + colm[1]);
          SB();
          MF(1, rPX[0], PA[0]); rQA[0] = rot_d<1>(QA[0]);
          SB();
          MF(3, PX[0], QA[0]); rQX[0] = rot_d<1>(QX[0]);
          SB();
          MF(1, rPA[1], PX[1]); rQA[1] = rot_d<1>(QA[1]);
          SB();
          MF(2, rQA[0], QX[0]); rQX[1] = rot_d<1>(QX[1]);
          SB();
          MF(3, PA[1], QX[1]);
          MF(2, rQX[0], QA[0]);
          MF(1, rPX[1], PA[1]);
          MF(3, PX[1], QA[1]);
          MF(2, rQA[1], QX[1]);
          SB();
          MF(2, rQX[1], QA[1]);
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (W8) {
          // eight waves: unit u = (this half's row-side map, column-side map b_u), both chains; the six reads of the next step
          // between the eight products
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NSTEP) frag8(cur ^ 1, (t + 1) % NKP, pa, px);
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
#pragma unroll
            for (int u = 0; u < 2; u++) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fA8[cur][0][hh], fX8[cur][1 + u][hh], w[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; u++) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fX8[cur][0][hh], fA8[cur][1 + u][hh], w[u], 0, 0, 0);
          }
          if (t + 1 < NSTEP) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one matrix-core product
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (!HALF) {
          // forward: unit u = (row-side map u >> 1, column-side map 2 + (u & 1)); eight products, the eight reads of the next
          // step, eight products
#if VGPA_SYM_INTERLEAVE
          // the next step's eight reads BETWEEN this step's sixteen products, one read behind every second product: a block of
          // eight ds_read_b128 takes the LDS queue ~100 cycles to accept when eight waves of the CU do the same, and an in-order
          // wave that sits in it issues no products meanwhile
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NSTEP && !VGPA_ABL_NOFRAG) frag_all(cur ^ 1, (t + 1) % NKP, pa, px);
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
#pragma unroll
            for (int u = 0; u < 4; u++) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fA[cur][u >> 1][hh], fX[cur][2 + (u & 1)][hh], w[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; u++)
              if (!(LOOP1 && u == kCoverLoopSlot)) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fX[cur][u >> 1][hh], fA[cur][2 + (u & 1)][hh], w[u], 0, 0, 0);
          }
          if (t + 1 < NSTEP) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // two matrix-core products
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {                            // (14 products per step with single-chain loop units, 16 without)
              if constexpr (LOOP1) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              else __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
#pragma unroll
            for (int u = 0; u < 4; u++) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fA[cur][u >> 1][hh], fX[cur][2 + (u & 1)][hh], w[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; u++)
              if (!(LOOP1 && u == kCoverLoopSlot)) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fX[cur][u >> 1][hh], fA[cur][2 + (u & 1)][hh], w[u], 0, 0, 0);
            if (hh == 0) {
              __builtin_amdgcn_sched_barrier(0);
              if (t + 1 < NSTEP) frag_all(cur ^ 1, (t + 1) % NKP, pa, px);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#endif
        } else {
#if VGPA_SYM_INTERLEAVE
          // backward: first half = units (a0, b0), (a0, b1) with the next step's b0, b1, a1 reads between the products (a1 in
          // two buffers: 56 fragment registers), second half = units (a1, b0), (a1, b1) with the next step's a0 reads
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NSTEP && !VGPA_ABL_NOFRAG) { frag_b(cur ^ 1, (t + 1) % NKP, pa, px); frag_a1(cur ^ 1, (t + 1) % NKP, pa, px); }
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
#pragma unroll
            for (int m = 0; m < 2; m++) w[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(fAa[0][hh], fXb[cur][m][hh], w[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; m++) w[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(fXa[0][hh], fAb[cur][m][hh], w[m], 0, 0, 0);
          }
          if (t + 1 < NSTEP) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NSTEP && !VGPA_ABL_NOFRAG) frag_a(0, (t + 1) % NKP, pa, px);
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
            if constexpr (LOOP1) {       // loop unit (slot 2): one chain; slot 3's two products are kept apart (dependent accumulator)
              w[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(fAa1[cur][hh], fXb[cur][1][hh], w[3], 0, 0, 0);
              w[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(fAa1[cur][hh], fXb[cur][0][hh], w[2], 0, 0, 0);
              w[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(fXa1[cur][hh], fAb[cur][1][hh], w[3], 0, 0, 0);
            } else {
#pragma unroll
              for (int m = 0; m < 2; m++) w[2 + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(fAa1[cur][hh], fXb[cur][m][hh], w[2 + m], 0, 0, 0);
#pragma unroll
              for (int m = 0; m < 2; m++) w[2 + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(fXa1[cur][hh], fAb[cur][m][hh], w[2 + m], 0, 0, 0);
            }
          }
          if (t + 1 < NSTEP) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
#pragma unroll
            for (int m = 0; m < 2; m++) w[2 * half + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(fAa[half][hh], fXb[cur][m][hh], w[2 * half + m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; m++)
              if (!(LOOP1 && 2 * half + m == kCoverLoopSlot)) w[2 * half + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(fXa[half][hh], fAb[cur][m][hh], w[2 * half + m], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NSTEP) {
            frag_a(half, (t + 1) % NKP, pa, px);
            if (half == 0) frag_b(cur ^ 1, (t + 1) % NKP, pa, px);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
        }
      } else {
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
#pragma unroll
        for (int u = 0; u < NSL; u++) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa1[cur][u >> 1][hh], fb1[cur][u][hh], w[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NSL; u++) w[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa2[cur][u >> 1][hh], fb2[cur][u][hh], w[u], 0, 0, 0);
      }
      }
      chore(t);                          // everything of the stage that is not the product rides between the products
      if (kp == NKP - 1) {
        if constexpr (LOOP1) {           // Z = M + M^T - F on the diagonal blocks: element (r4, c4) of a block adds element (c4, r4)
          w[LSLOT] += __shfl(w[LSLOT], loop_src, 64);
        }
#pragma unroll
        for (int u = 0; u < NSL; u++) {
          const int s = 2 * g0 + u;
          if (s < MAXS) {
            double js = 0.0;
            if (!FWD && last) js = own[s] ? jsd[s] : 0.0;
            const double xn = element(s, j, w[u], js);
            *(wd[s] ? Xn + offD[s] : trash) = xn;
            *(wm[s] ? Xn + offM[s] : trash) = xn;
          }
        }
      }
    }
  };

  // ---- time loop ----------------------------------------------------------------------------------------------------------
  // which buffers stage j of step k reads: the stage buffer that holds the current stage state (with an even number of
  // stages per step a compile-time function of j) and the A operand of the matrix / of the vector
  auto xcur = [&](int k, int j) -> double* { return (((NS & 1) ? (k & 1) : (j & 1)) ? Xb1 : Xb0); };
  auto aop = [&](int k, int j, bool matrix) -> const double* {
    const int op = stage_op<METHOD, FWD>(j, matrix, k);
    return op == OP_X ? xcur(k, j) : (op == OP_M ? Mb : Rb);
  };
  if (role == 0) product_begin(aop(0, 0, true), xcur(0, 0));
  VGPA_STAMP_DECL;
  // Chores between the products (round 3).  Nothing of a stage but the stepper depends on the stage's product, so everything
  // else is issued INSIDE the product pipeline, where its LDS and HBM latencies run under matrix-core work instead of behind it
  // (an in-order wave that does them after the products pays every round trip in full: ~1 700 of a stage's ~3 000 cycles):
  //   in front of the first products (TA), beside their fragments' way from LDS: the vector update of the PREVIOUS stage (its
  //   partial sums were completed by the last barrier),
  //   after step TB: this stage's partial inner products (they read the stage vector written at TA, same wave, in order),
  //   after step TC: operand staging, the state's way to HBM, the step's HBM loads.
  // slots: -1 = in front of the first products (beside their fragments' way from LDS), t >= 0 = behind pipeline step t
  // The split placement is used by the fragment-cover kernels (D = 33 .. 40, the benchmark's size); the run layout keeps every
  // chore in ONE slot (read and finish back to back) and no scheduling barriers: its 8-product steps unrolled 40-fold (D = 64)
  // already fill every register the wave has, and pinning more order made the compiler spill scalars by the hundred.
  constexpr int LAST = NSTEP - 1;
  constexpr bool SPLIT = COVER;
  // which chores are split is a register question (256 per lane with two workgroups per CU; a spilled value costs more than an
  // exposed LDS round trip): forward A and C, backward -- whose lanes also carry G_t, G_{t-1} -- A only, unless overridden
#ifndef VGPA_SYM_SPLIT_A_FWD
#define VGPA_SYM_SPLIT_A_FWD 1
#endif
#ifndef VGPA_SYM_SPLIT_A_BWD
#define VGPA_SYM_SPLIT_A_BWD 1
#endif
#ifndef VGPA_SYM_SPLIT_B_FWD
#define VGPA_SYM_SPLIT_B_FWD 0
#endif
#ifndef VGPA_SYM_SPLIT_B_BWD
#define VGPA_SYM_SPLIT_B_BWD 0
#endif
#ifndef VGPA_SYM_SPLIT_C_FWD
#define VGPA_SYM_SPLIT_C_FWD 1          // (measured with the single-chain loop units: forward 7.80 -> 7.65 ms, same box, no spills)
#endif
#ifndef VGPA_SYM_SPLIT_C_BWD
#define VGPA_SYM_SPLIT_C_BWD 0
#endif
  constexpr bool SPA = SPLIT && (FWD ? VGPA_SYM_SPLIT_A_FWD : VGPA_SYM_SPLIT_A_BWD);
  constexpr bool SPB = SPLIT && (FWD ? VGPA_SYM_SPLIT_B_FWD : VGPA_SYM_SPLIT_B_BWD);
  constexpr bool SPC = SPLIT && (FWD ? VGPA_SYM_SPLIT_C_FWD : VGPA_SYM_SPLIT_C_BWD);
  constexpr int TBU = NSTEP >= 5 ? NSTEP / 4 + 1 : (NSTEP > 1 ? 1 : 0), TCU = NSTEP >= 5 ? NSTEP / 2 + 1 : LAST;   // unsplit slots
  // (W8: a pipeline step is eight products = 128 cycles, shorter than an LDS round trip under load: every chore's read and
  //  finish are TWO steps apart, and the vector's partial products -- which read what the vector update wrote -- come last)
  constexpr int SA_F = W8 ? cmin(1, LAST) : 0, SA_R = (SPA || W8) ? -1 : SA_F;
  constexpr int SB_F = W8 ? LAST : (SPLIT ? cmin(1, LAST) : TBU), SB_R = W8 ? cmin(2, LAST) : (SPB ? 0 : SB_F);
  constexpr int SC_F = W8 ? cmin(3, LAST) : (SPLIT ? cmin(2, LAST) : TCU), SC_R = W8 ? 0 : (SPC ? cmin(1, LAST) : SC_F);
  // (helper waves: the loop exists twice, once per role, so that each role's loop carries only its own state -- one loop with a role
  //  branch inside keeps the union of both alive across the back-edge, and the backward instantiations spill)
  auto time_loop = [&](auto helper_role) {
#ifdef VGPA_STAMPS_ROLE
  long long ts_prev_ = clock64();
#endif
  constexpr int HK = HLP ? decltype(helper_role)::value : 0;    // 0: product waves (or no helpers); 1: the helper role; 2 / 3 (H2): vector / staging helper
  constexpr bool HR = HK != 0;                                  // this copy is a helper role's
  for (int k = 0; k < n_steps; k++) {
    if (k > 0) {                         // what the last step's prefetch brought (the ONE place that waits for HBM)
      if (!HLP || HK == 1 || HK == 3) {
#pragma unroll
        for (int q = 0; q < NITS; q++) settle(an[q]);
      }
      if (!FWD) {
        if (!HR) {
#pragma unroll
          for (int s = 0; s < MAXS; s++) { fc[s] = fn[s]; fn[s] = fnn[s]; }
        }
        n_obs_cur = n_obs_next; n_obs_next = n_obs_nn;
      }
    }
#pragma unroll
    for (int j = 0; j < NS; j++) {
      const double* Xc = xcur(k, j);
      double* Xn = (Xc == Xb0) ? Xb1 : Xb0;
      double* pv = pvb + (Xc == Xb0 ? 0 : g::PV);
      double* pv_prev = pvb + (Xc == Xb0 ? g::PV : 0);         // where the previous stage left its partial sums
      const double* Aopv = aop(k, j, false);
      if constexpr (HR) {
        // every chore of the stage, requests first: the partial sums of the previous stage and (stage 0) the start-point operand and
        // the stage state; then the vector update, its partial products (they read the vector just written, same wave, in order),
        // the staging, the HBM stores and loads
        if (HK != 3 && (j > 0 || k > 0)) vecA_read(pv_prev);
        if (HK != 2) tailC_read(j, Xc);
        if (HK != 3) {
          if (j > 0) vecA_finish(j - 1);
          else if (k > 0) {
            vecA_finish(NS - 1);
            c0 = c1; c1 = c2;
            if (!FWD) jm = jm_next;
          }
          tailB_read(Aopv);
          tailB_finish(pv);
        }
        if (HK == 1) tailC_finish(j, k, K1{});
        if (HK == 2) vec_chores(j, k);
        if (HK == 3) tailC_finish(j, k, K3{});
      } else
      product_stage(j, k, aop(k, j, true), Xc, Xn, [&](int t) {
        if (HLP) {                       // (product waves beside helpers: only their own HBM loads -- G_t of the step after the next)
          if (t == SC_F && j == JSEC) prefetch(k, std::false_type{});
          return;
        }
        if (SPLIT) __builtin_amdgcn_sched_barrier(0);
        if (!VGPA_ABL_NOVEC) {
        if (t == SA_R && (j > 0 || k > 0)) vecA_read(pv_prev);
        if (t == SA_F) {
          if (j > 0) vecA_finish(j - 1);
          else if (k > 0) {
            vecA_finish(NS - 1);
            c0 = c1; c1 = c2;                                   // the vector's forcing terms move on behind its last stage
            if (!FWD) jm = jm_next;
          }
        }
        if (t == SB_R) tailB_read(Aopv);
        if (t == SB_F) tailB_finish(pv);
        }
        if (t == SC_R) tailC_read(j, Xc);
        if (t == SC_F) tailC_finish(j, k, K1{});
        if (SPLIT) __builtin_amdgcn_sched_barrier(0);
      });
#ifdef VGPA_STAMPS_ROLE      // diagnostic build (tools/ubench/ode_gf_loop.hip): first wave of each role of workgroup 0, per stage: [busy | wait at the barrier]
      const long long tsb_ = clock64();
#endif
      VGPA_STAMP(0, 0);
      __builtin_amdgcn_s_setprio(VGPA_SYM_TAILPRIO);
      lds_barrier();
#ifdef VGPA_STAMPS_ROLE
      { const long long tsa_ = clock64();
        if (lane == 0 && wave == 0 && blockIdx.x == 0) { mfma::g_stamp_role[HK == 3 ? 2 : (HR ? 1 : 0)][2 * j] += tsb_ - ts_prev_; mfma::g_stamp_role[HK == 3 ? 2 : (HR ? 1 : 0)][2 * j + 1] += tsa_ - tsb_; }
        ts_prev_ = tsa_; }
#endif
      VGPA_STAMP(0, 2);
      // the next stage's first fragments (Xn is complete now; past the last stage of the sweep they are read and dropped)
      const int kn = j + 1 < NS ? k : k + 1, jn = j + 1 < NS ? j + 1 : 0;
      if (!HR) product_begin(aop(kn, jn, true), Xn);
#ifndef VGPA_HLP_PRIO
#define VGPA_HLP_PRIO 0
#endif
      __builtin_amdgcn_s_setprio(HR ? VGPA_HLP_PRIO : 0);
      VGPA_STAMP(0, 3);
    }
  }
  };
  if (H2 && role == 1) time_loop(K2{});
  else if (H2 && role == 2) time_loop(K3{});
  else if (HLP && helper) time_loop(K1{});
  else time_loop(std::integral_constant<int, 0>{});
  if constexpr (GF) {
    // the last grid point's gradient: three more barrier intervals.  The helpers finish the vector and leave lam; the gradient
    // waves build (Psi in the stage buffer, the end point's operand in R), then band / u, product, out; the product waves only count.
    if (role == 0) { lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier(); return; }
  }
  if (HLP && !helper) return;             // (the last state and the last vector leave through the helper waves)
  if (!(H2 && role == 2) && n_steps > 0) {      // the last stage's vector update
    vecA_read(pvb + (xcur(n_steps, 0) == Xb0 ? g::PV : 0));
    vecA_finish(NS - 1);
  }
  if (H2 && role == 1) { store_vector(tidx(n_steps)); return; }
  if constexpr (GF) {
    if (wave == 0 && vl) { stg(vout + vec(tidx(n_steps)), lane8, vk); gLam[lane] = vk; }
    lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier();
  } else {
    d2_t items[g::NIT];
    load_items(xcur(n_steps, 0), items);
    if constexpr (QOUT) {                // the end point's operand is in R (staged at stage JSEC of the last step, or by the prologue)
      d2_t a0[NITS];
#pragma unroll
      for (int q = 0; q < NITS; q++) a0[q] = *unit_ptr(Rb, q);
      to_q(items, a0);
    }
    store_items(items, tidx(n_steps), !H2);
  }
}


// Eight waves per problem (k_ode_sym, NW = 8) are an EXPERIMENT kept reachable: VGPA_SYM_WAVES=8 in the environment selects them
// (tools/ab_waves.sh, tests).  They measured slower than four waves at every batch size -- the product phase is bound by the LDS
// fragment traffic as much as by the matrix pipe, and the split raises the fragment reads per product from 0.5 to 0.75.
inline bool eight_waves(int batch) {
  (void)batch;
#ifdef VGPA_EXPERIMENTS
  static const int forced = [] { const char* e = getenv("VGPA_SYM_WAVES"); return e ? atoi(e) : 0; }();
  return forced == 8;      // measured slower at every batch size (EXPERIMENTS.md s.9): opt-in only
#else
  return false;            // (not compiled into the product build)
#endif
}

// The outer-product cover (kOpMaps: two maps per wave, rotated row sides by DPP) is an EXPERIMENT kept reachable: VGPA_SYM_COVER=op.
// It measured slower than the round-3 fragment cover (EXPERIMENTS.md s.12): a rotated operand costs two v_mov_dpp, which do not
// issue under a running fp64 product, where the fragment read it replaces costs LDS cycles beside the matrix pipe.
inline bool old_cover() {
#ifdef VGPA_EXPERIMENTS
  static const bool op = [] { const char* e = getenv("VGPA_SYM_COVER"); return e && !strcmp(e, "op"); }();
  return !op;
#else
  return true;             // (not compiled into the product build)
#endif
}
// Helper-wave kernels: one helper role (512 threads) or two (768; VGPA_SYM_HELPERS=2 in the environment forces two, =1 one)
inline bool two_helper_roles() {
  static const int forced = [] { const char* e = getenv("VGPA_SYM_HELPERS"); return e ? atoi(e) : -1; }();
#ifndef VGPA_SYM_TWO_HELPERS
#define VGPA_SYM_TWO_HELPERS 1
#endif
  return forced >= 0 ? forced == 2 : VGPA_SYM_TWO_HELPERS != 0;
}
template <int METHOD, bool FWD, int NB, int GRC, bool HLP = false>
hipError_t launch_cover(const OdeArgs& a, hipStream_t st, bool dense) {
  constexpr size_t lds_c = SGeo<NB>::LDS_DOUBLES * sizeof(double);
  constexpr int WPE_C = (HLP || 2 * lds_c <= 160 * 1024) ? 2 : 1;
  constexpr int threads = HLP ? 512 : 256;
  if constexpr (!FWD && METHOD == VGPA_ODE_RK4 && HLP && GRC == 0) {
    if (a.grad_on) {                                   // the gradient assembly on the helper waves (k_ode_sym, GF)
      if (dense || !a.q_on || !a.s_packed || !a.g || !a.S || !a.m || !a.Ef || !a.Am || !a.b) return hipErrorInvalidValue;
      constexpr size_t lds_g = lds_c + GradLds<NB>::DOUBLES * sizeof(double);
      static_assert(lds_g <= 160 * 1024, "LDS budget");
      auto kg = k_ode_sym<METHOD, FWD, NB, false, GRC, 3, true, 4, true, true>;      // 768 threads: three waves per SIMD, 168 registers each
      (void)hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g);
      hipLaunchKernelGGL(kg, dim3(a.batch), dim3(768), lds_g, st, a);
      return hipGetLastError();
    }
  }
  if (a.grad_on) return hipErrorInvalidValue;          // (only the kernel above assembles the gradient)
  if constexpr (HLP && GRC == 0) {                     // two helper roles (k_ode_sym, H2): 768 threads, three waves per SIMD
    if (two_helper_roles()) {
      if constexpr (!FWD && (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4)) {
        if (a.q_on) {
          if (dense) return hipErrorInvalidValue;
          auto kq2 = k_ode_sym<METHOD, FWD, NB, false, GRC, 3, true, 4, true, false, true>;
          (void)hipFuncSetAttribute((const void*)kq2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
          hipLaunchKernelGGL(kq2, dim3(a.batch), dim3(768), lds_c, st, a);
          return hipGetLastError();
        }
      }
      auto kc2 = dense ? k_ode_sym<METHOD, FWD, NB, true, GRC, 3, false, 4, true, false, true> : k_ode_sym<METHOD, FWD, NB, false, GRC, 3, false, 4, true, false, true>;
      (void)hipFuncSetAttribute((const void*)kc2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
      hipLaunchKernelGGL(kc2, dim3(a.batch), dim3(768), lds_c, st, a);
      return hipGetLastError();
    }
  }
  if constexpr (!FWD && (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4)) {
    if (a.q_on) {
      if (dense) return hipErrorInvalidValue;          // (the fused sweeps bring sparse jumps)
      auto kq = k_ode_sym<METHOD, FWD, NB, false, GRC, WPE_C, true, 4, HLP>;
      if (lds_c > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
      hipLaunchKernelGGL(kq, dim3(a.batch), dim3(threads), lds_c, st, a);
      return hipGetLastError();
    }
  }
  auto kc = dense ? k_ode_sym<METHOD, FWD, NB, true, GRC, WPE_C, false, 4, HLP> : k_ode_sym<METHOD, FWD, NB, false, GRC, WPE_C, false, 4, HLP>;
  if (lds_c > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
  hipLaunchKernelGGL(kc, dim3(a.batch), dim3(threads), lds_c, st, a);
  return hipGetLastError();
}

// Helper waves (k_ode_sym, HLP): up to one problem per CU.  VGPA_SYM_HELPERS=0 / 1 in the environment forces them off / on for
// every batch size (comparison runs, tests).
inline bool helper_waves(int batch) {
  static const int forced = [] { const char* e = getenv("VGPA_SYM_HELPERS"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced != 0;
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n;
  }();
  return batch <= n_cu;
}

template <int METHOD, bool FWD, int NB>
hipError_t launch_sym(const OdeArgs& a, hipStream_t st) {
  constexpr size_t lds = SGeo<NB>::LDS_DOUBLES * sizeof(double);
  static_assert(lds <= 160 * 1024, "LDS budget");
  // runs per pipeline step.  Two (four accumulators in turn, 16 MFMAs per step) spill with 256 registers and were slower with
  // 512 (D = 64: 22.0 vs 15.7 ms forward); the kernel is only exercised with one.
  // NSB = 5 (33 <= D <= 40): the fragment cover (GR = 0; VGPA_SYM_RUNS=1 keeps the run layout for comparison)
  const bool runs_only = runs_only_env();
  constexpr bool can_cover = SGeo<NB>::NSB == 5;
  constexpr int GR = 1;
  const bool dense = !FWD && a.js_dense;
  if constexpr (can_cover) {
#ifdef VGPA_EXPERIMENTS
    if (!runs_only && !dense && eight_waves(a.batch) && !a.grad_on) {
      // up to one problem per CU: eight waves per problem (k_ode_sym, NW = 8), two per SIMD, all 256 registers each
      constexpr size_t lds_8 = SGeo<NB, 8>::LDS_DOUBLES * sizeof(double);
      static_assert(lds_8 <= 160 * 1024, "LDS budget");
      if constexpr (!FWD && (METHOD == VGPA_ODE_RK2 || METHOD == VGPA_ODE_RK4)) {
        if (a.q_on) {
          auto kq8 = k_ode_sym<METHOD, FWD, NB, false, 0, 2, true, 8>;
          (void)hipFuncSetAttribute((const void*)kq8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_8);
          hipLaunchKernelGGL(kq8, dim3(a.batch), dim3(512), lds_8, st, a);
          return hipGetLastError();
        }
      }
      auto k8 = k_ode_sym<METHOD, FWD, NB, false, 0, 2, false, 8>;
      (void)hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_8);
      hipLaunchKernelGGL(k8, dim3(a.batch), dim3(512), lds_8, st, a);
      return hipGetLastError();
    }
#endif
    if (!runs_only && old_cover() && (helper_waves(a.batch) || (!FWD && a.grad_on))) return launch_cover<METHOD, FWD, NB, 0, true>(a, st, dense);
#ifdef VGPA_EXPERIMENTS
    if (!runs_only && !old_cover()) return launch_cover<METHOD, FWD, NB, -1>(a, st, dense);
#endif
    if (!runs_only) return launch_cover<METHOD, FWD, NB, 0>(a, st, dense);
  }
  if (a.grad_on) return hipErrorInvalidValue;
  constexpr int WPE = 2 * lds <= 160 * 1024 ? 2 : 1;     // two workgroups per CU when their LDS fits, else all 512 registers
  auto kern = dense ? k_ode_sym<METHOD, FWD, NB, true, GR, WPE> : k_ode_sym<METHOD, FWD, NB, false, GR, WPE>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(a.batch), dim3(256), lds, st, a);
  return hipGetLastError();
}

// D <= 44 has both kernel families.  Default: the role-specialised 8-wave kernels of ode_mfma_impl.h (faster for one problem
// per CU and for a single problem); VGPA_ODE_KERNEL=sym selects the symmetric-unit kernels (faster from two problems per CU on).
inline bool use_sym_kernels() {
  static const bool sym = [] { const char* e = getenv("VGPA_ODE_KERNEL"); return e && !strcmp(e, "sym"); }();
  return sym;
}
template <int METHOD, bool FWD, int NB>
hipError_t launch_any(const OdeArgs& a, hipStream_t st) {
  if (a.sym_units || use_sym_kernels()) return launch_sym<METHOD, FWD, NB>(a, st);
  return mfma::launch_nb<METHOD, FWD, NB>(a, st);
}

}  // namespace sym
}  // namespace vgpa
