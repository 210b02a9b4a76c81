// FOUR trajectories per wavefront on the small fp64 matrix instruction of gfx950: the forward cubature cell of every model --
// d = nx + nu <= 8 (pendulum ... double cartpole, planar quadrotor), whatever their observation function, and d = 16 (the 12-state
// quadrotor) -- and the backward cell of the d = 16 model (backward_quad_body, at the end of this file).
//
//   v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4 x 4 blocks at once. Block g lives in lanes {16 k + 4 g + m}: lane
//   l = 16 r + 4 g + c holds element (r, c) of block g -- as result / accumulator and as B operand; as A operand the same lane
//   supplies A[c][r], i.e. a resident block is multiplied TRANSPOSED from the left (probed, not assumed:
//   tools/micro/mfma_f64_4x4.hip, profiles/r4_micro_mfma_f64_4x4.txt). So:
//   * trajectory g of a wave owns block g; every matrix of its cell is cut into 4 x 4 blocks, one fp64 register per block, one
//     element per lane (an 8 x 8 joint covariance = 4 registers, the double cartpole's 9 x 9 observation covariance = 9);
//   * X^T Y of two resident block matrices is one instruction per (k, i, j) block triple with NO operand movement, for four
//     trajectories at once (q_tn), at ~16 clocks of issue -- an 8 x 8 x 8 product of four trajectories in 8 instructions.
//     Everything is arranged so that only X^T Y forms occur (as in i2c_wave.hpp); a block transpose, a column sum, a
//     column-form -> row-form change of a vector are each ONE instruction against a constant block (identity / ones);
//   * factorisations are blocked with 4 x 4 pivots: the pivot block goes through 16 doubles of LDS to the sixteen lanes of its
//     trajectory, which factor and invert it in registers (compile-time shorter for a partly filled last block: the 9th
//     observation of the double cartpole is a scalar pivot); scaling and elimination of the block row, the trailing matrix and
//     any number of right-hand-side block columns are matrix instructions. Solves are right-hand sides of eliminations;
//   * the 2 d (+ 1) sigma points of a transform are evaluated ONE PER LANE by the sixteen lanes of a trajectory (lane p < 8:
//     m + sf L[:, p], lane 8 + p: m - sf L[:, p], a spare lane: the centre) through the model functors of i2c_models.hpp --
//     general observation functions included; directions and results change layout through the trajectory's LDS region.
// The math is the reference's I2cCell._forward_msgs_quadrature (i2c/i2c.py:350-447) and QuadratureInference
// (i2c/inference/quadrature.py:15-58) in the centred pairwise form of sp_transform (i2c_cell.hpp). Buffers are the common
// [T][E][B] ones, so the backward schedules of the lane kernels consume what this sweep writes.
//
// Host simulation (tests only): the 64 lanes are 64 threads; every cross-lane instruction is an exchange through a shared
// buffer between two barriers.
#pragma once
#include "i2c_wave.hpp"

namespace i2c {

// LDS region of one trajectory (elements) and the geometry of a model's sigma-point evaluations.
//   d <= 8:  eight pair rows; the sixteen lanes of a trajectory evaluate all 2 d points (and the centre) in ONE pass;
//   d <= 16: sixteen pair rows; two passes (lane p: m + sf L[:, p], then m - sf L[:, p]); the directions share their LDS with the
//            results of the second pass (written after the last direction has been read), so that two waves per SIMD fit a CU.
constexpr int Q_O_DG = 0, Q_O_DG2 = 16;  // 4 x 4 pivot blocks of an elimination / of the second one of a pair (q_elim2)
template <class M> struct QG {
  static constexpr int D = M::NX + M::NU;
  static constexpr bool WIDE = D > 8;
  static constexpr int PR = WIDE ? 16 : 8;        // pair rows
  static constexpr int LDL = WIDE ? 17 : 10;      // row stride of the sigma-point directions
  static constexpr int YLD = 13;                  // row stride of the evaluation outputs: 2 PR rows (points) of <= 12 outputs (odd: the
                                                  // sixteen lanes of a trajectory write their rows to sixteen different bank pairs)
  static constexpr int NMAX = M::NZ > M::NX ? (M::NZ > M::NZT ? M::NZ : M::NZT) : (M::NX > M::NZT ? M::NX : M::NZT);
  static constexpr int QLD = NMAX > 12 ? 16 : 12;  // row stride of the batch constants (QConst)
  static constexpr int O_MV = 32;                 // the mean the points are built around (PR)
  static constexpr int O_Y = O_MV + PR;           // results of the + points (d <= 8: of all points, rows 0 .. 15)
  static constexpr int O_YM = O_Y + PR * YLD;     // results of the - points
  static constexpr int YM_SIZE = (WIDE && PR * LDL > PR * YLD) ? PR * LDL : PR * YLD;
  static constexpr int O_L = WIDE ? O_YM : O_YM + PR * YLD;
  static constexpr int RAW = WIDE ? O_YM + YM_SIZE : O_L + PR * LDL;
  static constexpr int SIZE = RAW + ((24 - RAW % 16) % 16);  // (= 8 mod 16 elements: the four regions of a wave start 16 banks apart)
};

// Diagnostic build only (-DI2C_QUAD_STAMPS, never in the shipped library): s_memtime stamps at the phase boundaries of the forward
// cell, summed per phase and printed by trajectory 0 -- where a lone wave spends its cycles.
#if defined(I2C_QUAD_STAMPS) && !defined(I2C_HOST_SIM)
#define I2C_QSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_last; stamp_last = now_; } while (0)
#else
#define I2C_QSTAMP(i) do { } while (0)
#endif

template <typename R> struct Quad {
  int l, r, g, c;  // lane; block row (l >> 4), trajectory slot ((l >> 2) & 3), block column (l & 3)
  lds_ptr<R> sh;   // this trajectory's LDS region
#ifdef I2C_HOST_SIM
  HostBarrier* bar;
  R* xch;  // 128 slots: operands of the emulated cross-lane instructions
#endif
  I2C_MEM int p() const { return 4 * r + c; }  // lane index inside the trajectory: which sigma point it evaluates
  I2C_MEM void sync() const {
#ifdef I2C_HOST_SIM
    bar->wait();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
  }
};

// the same lane with its block coordinates opaque to loop-invariant code motion (see forward_quad_body)
template <typename R> I2C_FN Quad<R> q_opaque(const Quad<R>& q) {
  Quad<R> o = q;
  o.r = opaque_i(q.r);
  o.c = opaque_i(q.c);
  return o;
}

// ---- the cross-lane instructions -------------------------------------------------------------------------------------------
// acc (block) += A B per trajectory: lane (r, c) supplies A[c][r] and B[r][c] and holds acc[r][c]
template <typename R> I2C_FN void q_mfma(const Quad<R>& q, const R a, const R b, R& acc) {
#ifdef I2C_HOST_SIM
  q.bar->wait();
  q.xch[q.l] = a;
  q.xch[64 + q.l] = b;
  q.bar->wait();
  R s = acc;
  for (int k = 0; k < 4; ++k) s = std::fma(q.xch[16 * k + 4 * q.g + q.r], q.xch[64 + 16 * k + 4 * q.g + q.c], s);
  acc = s;
#else
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
#endif
}
// acc[NI][NJ] += X^T Y for block matrices X[NK][NI], Y[NK][NJ] (row-major arrays of block registers); NEG: -=.
// UPPER: only the blocks i <= j (symmetric results whose lower blocks nobody reads); YUP: Y is block upper triangular.
template <int NK, int NI, int NJ, bool NEG = false, bool UPPER = false, bool YUP = false, typename R>
I2C_FN void q_tn(const Quad<R>& q, const R* x, const R* y, R* acc) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (UPPER && i > j) continue;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (YUP && k > j) continue;
        q_mfma(q, NEG ? -x[k * NI + i] : x[k * NI + i], y[k * NJ + j], acc[i * NJ + j]);
      }
    }
}
// block transpose
template <typename R> I2C_FN R q_tr(const Quad<R>& q, const R x) {
  R t = R(0);
  q_mfma(q, x, q.r == q.c ? R(1) : R(0), t);
  return t;
}
// sum over the four rows of a block, in every row
template <typename R> I2C_FN R q_colsum(const Quad<R>& q, const R x) {
  R t = R(0);
  q_mfma(q, R(1), x, t);
  return t;
}
// value held by column K of the caller's block row (DPP quad_perm)
template <int K, typename R> I2C_FN R q_bcq(const Quad<R>& q, const R x) {
#ifdef I2C_HOST_SIM
  q.bar->wait();
  q.xch[q.l] = x;
  q.bar->wait();
  return q.xch[(q.l & ~3) + K];
#else
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#endif
}
// sum over the four columns of a block row, in every column
template <typename R> I2C_FN R q_rowsum(const Quad<R>& q, const R x) {
#ifdef I2C_HOST_SIM
  q.bar->wait();
  q.xch[q.l] = x;
  q.bar->wait();
  const int b = q.l & ~3;
  return (q.xch[b] + q.xch[b + 1]) + (q.xch[b + 2] + q.xch[b + 3]);
#else
  int lo = __double2loint(x), hi = __double2hiint(x);
  const double y = x + __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false));  // [1,0,3,2]
  lo = __double2loint(y), hi = __double2hiint(y);
  return y + __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false));  // [2,3,0,1]
#endif
}

// Batch-wide constants by (row, column), in LDS, one copy per workgroup: QLD x QLD row-major, zero-padded (d <= 8: 6 KB; with the
// 10 KB of its four trajectory regions a workgroup stays under the 20 KB that let eight of them -- two waves per SIMD -- share a CU)
template <class M, typename R> struct QConst {
  static constexpr int QLD = QG<M>::QLD;
  R xi[QLD * QLD], eta[QLD * QLD], xiT[QLD * QLD], qr[QLD * QLD], qf[QLD * QLD];  // sig_xi0, sig_eta, sig_xiT0, blkdiag(Q, R), Qf
  R zg[QLD], zgT[QLD];
};
template <class M, typename R, class DST> I2C_FN void qconst_fill(DST& k, const Consts<M, R>* c, const int tid, const int nthreads) {
  constexpr int NX = M::NX, NZ = M::NZ, NT = M::NZT > 0 ? M::NZT : 1, QLD = QG<M>::QLD;
  for (int e = tid; e < QLD * QLD; e += nthreads) {
    const int i = e / QLD, j = e % QLD;
    k.xi[e] = (i < NZ && j < NZ) ? c->sig_xi0[tri_any(i, j)] : R(0);
    k.eta[e] = (i < NX && j < NX) ? c->sig_eta_w[tri_any(i, j)] : R(0);  // W sig_eta = sum_p w_p sig_eta (quadrature.py:57; W = 1 unless 1 - alpha^2 + beta != 0)
    k.xiT[e] = (i < NT && j < NT) ? c->sig_xiT0[tri_any(i, j)] : R(0);
    k.qr[e] = (i < NZ && j < NZ) ? c->QR[tri_any(i, j)] : R(0);
    k.qf[e] = (i < NT && j < NT) ? c->Qf[tri_any(i, j)] : R(0);
  }
  for (int e = tid; e < QLD; e += nthreads) {
    k.zg[e] = e < NZ ? c->zg[e] : R(0);
    k.zgT[e] = e < NT ? c->zg_term[e] : R(0);
  }
}
// (kz: an index the compiler cannot see through, always 0 and refreshed per cell: the constants are READ where they are used.
//  Loop-invariant LDS loads are otherwise hoisted out of the time loop into dozens of registers -- and, for d = 16, spilled)
template <int QLD, typename R, class P> I2C_FN R q_ldc(const Quad<R>& q, const P m, const int I, const int J, const int kz = 0) { return m[(4 * I + q.r) * QLD + 4 * J + q.c + kz]; }
template <typename R, class P> I2C_FN R q_ldv(const Quad<R>& q, const P v, const int J, const int kz = 0) { return v[4 * J + q.c + kz]; }  // column form

// ---- blocked Cholesky elimination --------------------------------------------------------------------------------------------
// s: SPD matrix of dimension N in NB x NB blocks, of which the UPPER blocks are read (consumed). On return lt = L^T (upper blocks;
// the others are left alone) and each right-hand side block matrix r1 [NB][NC1], r2 [NB][NC2] is replaced by L^-1 r. Returns
// whether every pivot was positive. Rows / columns >= N of s and of the right-hand sides must be zero (no identity padding: the
// pivot algebra of a partly filled block is compiled for its NL live rows).
// One pivot step = (a) the 4 x 4 pivot block through LDS to the sixteen lanes of its trajectory, (b) every lane factors it and
// forms ITS entry of the inverse factor, (c) matrix instructions scale block row K and eliminate it from everything below.
// q_elim2 runs two independent eliminations in lockstep: one LDS round trip per pair of pivot blocks, and two independent
// instruction streams for the scheduler to interleave (a lone wave per SIMD has nothing else to hide its latencies behind).
// The lower triangle of the pivot block from its LDS slot into registers: d = {d00, d10, d11, d20, d21, d22, d30, d31, d32, d33} (the NL
// live rows). Fetched as ONE batch and settled by ONE wait (q_pivot_settle): left to itself hipcc sinks the reads of the later
// rows into the factorisation to save registers, and a lone wave then pays a full LDS latency three times per pivot block.
template <int NL, typename R, class P> I2C_FN void q_pivot_fetch(const P dg, R* d) {
  static_assert(NL >= 1 && NL <= 4, "pivot block");
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) d[tri(i, j)] = dg[4 * i + j];
}
template <int NL, typename R> I2C_FN void q_pivot_settle(R* d) {
  // (a USE of every fetched value at this point, nothing redefined: "+v" operands made hipcc copy each of them)
#ifndef I2C_HOST_SIM
  if constexpr (NL == 1) asm volatile("" ::"v"(d[0]));
  if constexpr (NL == 2) asm volatile("" ::"v"(d[0]), "v"(d[1]), "v"(d[2]));
  if constexpr (NL == 3) asm volatile("" ::"v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]));
  if constexpr (NL == 4)
    asm volatile("" ::"v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]), "v"(d[9]));
#else
  (void)d;
#endif
}
template <int NL, typename R> I2C_FN void q_pivot_algebra(const Quad<R>& q, const R* d, R* aw_out, R* pl_out) {
  static_assert(NL >= 1 && NL <= 4, "pivot block");
  // 4 x 4 Cholesky of the pivot block (l), column by column; rows >= NL are identity rows
  const R d00 = d[0];
  const R i0 = r_rsqrt(d00);
  R pl = d00, i1 = R(1), i2 = R(1), i3 = R(1);
  R l10 = R(0), l20 = R(0), l21 = R(0), l30 = R(0), l31 = R(0), l32 = R(0);
  if constexpr (NL > 1) {
    const R d10 = d[1], d11 = d[2];
    l10 = d10 * i0;
    pl = d11 - l10 * l10;
    i1 = r_rsqrt(pl);
  }
  if constexpr (NL > 2) {
    const R d20 = d[3], d21 = d[4], d22 = d[5];
    l20 = d20 * i0;
    l21 = (d21 - l20 * l10) * i1;
    pl = d22 - l20 * l20 - l21 * l21;
    i2 = r_rsqrt(pl);
  }
  if constexpr (NL > 3) {
    const R d30 = d[6], d31 = d[7], d32 = d[8], d33 = d[9];
    l30 = d30 * i0;
    l31 = (d31 - l30 * l10) * i1;
    l32 = (d32 - l30 * l20 - l31 * l21) * i2;
    pl = d33 - l30 * l30 - l31 * l31 - l32 * l32;
    i3 = r_rsqrt(pl);
  }
  *pl_out = pl;  // (the last pivot: a failed one poisons everything after it, see chol(), i2c_linalg.hpp)
  // this lane's entry of the inverse of l as the A operand: lane (r, c) supplies A[c][r] = (l^-1)[c][r]. Row a of l^-1 solves
  // y^T l = e_a^T (back substitution; y_k = 0 for k > a falls out of the one-hot right-hand side); entry r picked by a one-hot
  // combination (written without selects: see w_elim_step, i2c_wave.hpp)
  const int a = q.c, cq = q.r;
  const R e0 = a == 0 ? R(1) : R(0), e1 = a == 1 ? R(1) : R(0), e2 = a == 2 ? R(1) : R(0), e3 = a == 3 ? R(1) : R(0);
  const R m0 = cq == 0 ? R(1) : R(0), m1 = cq == 1 ? R(1) : R(0), m2 = cq == 2 ? R(1) : R(0), m3 = cq == 3 ? R(1) : R(0);
  if constexpr (NL == 1) {
    *aw_out = (e0 * m0) * i0 + (e1 * m1 + e2 * m2 + e3 * m3);
  } else {
    const R y3 = e3 * i3;
    const R y2 = (e2 - l32 * y3) * i2;
    const R y1 = (e1 - l21 * y2 - l31 * y3) * i1;
    const R y0 = (e0 - l10 * y1 - l20 * y2 - l30 * y3) * i0;
    *aw_out = (m0 * y0 + m1 * y1) + (m2 * y2 + m3 * y3);
  }
}
// One pivot step, ordered for a lone wave (nothing else hides its latencies): the rows of L^T first, then the NEXT pivot block --
// updated, sent through LDS and fetched -- and only then the rest of the step (the right-hand sides, the trailing blocks), which
// runs while the fetch is in flight (look-ahead).
//   scale block row K by the inverse pivot factor (aw): rows of L^T (the diagonal block masked to its upper triangle: what is left
//   of it is rounding noise) ...
template <int K, int NB, typename R> I2C_FN void q_elim_scale_lt(const Quad<R>& q, const R aw, const R* s, R* lt) {
#pragma unroll
  for (int j = K; j < NB; ++j) {
    R x = R(0);
    q_mfma(q, aw, s[K * NB + j], x);
    lt[K * NB + j] = (j > K || q.c >= q.r) ? x : R(0);
  }
}
// ... and of every right-hand side.
// R1ANTI: the first right-hand side (NC1 = NB) has zero blocks left of its anti-diagonal -- block (i, j) with j < NB - 1 - i --, as
// the row-reversed transpose of a triangular factor has (q_kalman_sqrt); the elimination keeps that pattern and skips those blocks.
// R2LOW: the second right-hand side starts as the identity (NC2 = NB): L^-1 I is block lower triangular, its blocks right of the
// diagonal stay zero and are skipped
template <int K, int NC1, int NC2, bool R2LOW = false, bool R1ANTI = false, typename R> I2C_FN void q_elim_scale_rhs(const Quad<R>& q, const R aw, R* r1, R* r2) {
#pragma unroll
  for (int j = 0; j < NC1; ++j) {
    if (R1ANTI && j < NC1 - 1 - K) continue;
    R x = R(0);
    q_mfma(q, aw, r1[K * NC1 + j], x);
    r1[K * NC1 + j] = x;
  }
#pragma unroll
  for (int j = 0; j < NC2; ++j) {
    if (R2LOW && j > K) continue;
    R x = R(0);
    q_mfma(q, aw, r2[K * NC2 + j], x);
    r2[K * NC2 + j] = x;
  }
}
// the next pivot block: s[K+1][K+1] -= lt[K][K+1]^T lt[K][K+1]
template <int K, int NB, typename R> I2C_FN void q_elim_next_pivot(const Quad<R>& q, R* s, const R* lt) {
  q_mfma(q, -lt[K * NB + K + 1], lt[K * NB + K + 1], s[(K + 1) * NB + K + 1]);
}
// eliminate block row K from everything below, except the next pivot block (q_elim_next_pivot)
template <int K, int NB, int NC1, int NC2, bool R2LOW = false, bool R1ANTI = false, typename R> I2C_FN void q_elim_below(const Quad<R>& q, R* s, R* r1, R* r2, const R* lt) {
#pragma unroll
  for (int i = K + 1; i < NB; ++i) {
    const R nl = -lt[K * NB + i];
#pragma unroll
    for (int j = i; j < NB; ++j) {
      if (i == K + 1 && j == K + 1) continue;
      q_mfma(q, nl, lt[K * NB + j], s[i * NB + j]);
    }
#pragma unroll
    for (int j = 0; j < NC1; ++j) {
      if (R1ANTI && j < NC1 - 1 - K) continue;
      q_mfma(q, nl, r1[K * NC1 + j], r1[i * NC1 + j]);
    }
#pragma unroll
    for (int j = 0; j < NC2; ++j) {
      if (R2LOW && j > K) continue;
      q_mfma(q, nl, r2[K * NC2 + j], r2[i * NC2 + j]);
    }
  }
}
// nothing is scheduled across this point (device only): keeps the look-ahead order hipcc would otherwise undo
I2C_FN void q_sched_fence() {
#ifndef I2C_HOST_SIM
  __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int N, int K> constexpr int q_live_rows() { return (N - 4 * K) >= 4 ? 4 : (N - 4 * K); }
// the pivot block K of s through the LDS slot at `off` to the sixteen lanes of its trajectory (issued, not yet settled)
template <int NL, int NB, int K, typename R> I2C_FN void q_pivot_send(const Quad<R>& q, const R* s, const int off, R* d) {
  const auto dg = q.sh + off;
  q.sync();
  dg[4 * q.r + q.c] = s[K * NB + K];
  q.sync();
  q_pivot_fetch<NL>(dg, d);
}
// d: pivot block K, fetched by the caller (the previous step, or q_elim)
// awk (optional): receives the inverse pivot factors, awk[K] = (l_K^-1)^T as a resident block -- i.e. the INVERSE of the diagonal
// block (K, K) of L^T (rows >= the live rows: identity) -- for a blocked back substitution with L^T afterwards (backward_quad8_body)
template <int K, int NB, int N, int NC1, int NC2, bool R2LOW = false, bool R1ANTI = false, typename R>
I2C_FN void q_elim_step(const Quad<R>& q, R* s, R* r1, R* r2, R* lt, R* last, R* d, R* awk = nullptr) {
  constexpr int NL = q_live_rows<N, K>();
  R aw, pl;
  q_pivot_settle<NL>(d);
  q_pivot_algebra<NL>(q, d, &aw, &pl);
  if (K == NB - 1) *last = pl;
  if (awk) awk[K] = aw;
  q_elim_scale_lt<K, NB>(q, aw, s, lt);
  if constexpr (K + 1 < NB) {
    R dn[10];
    q_sched_fence();
    q_elim_next_pivot<K, NB>(q, s, lt);
    q_pivot_send<q_live_rows<N, K + 1>(), NB, K + 1>(q, s, Q_O_DG, dn);
    q_sched_fence();
    q_elim_scale_rhs<K, NC1, NC2, R2LOW, R1ANTI>(q, aw, r1, r2);
    q_elim_below<K, NB, NC1, NC2, R2LOW, R1ANTI>(q, s, r1, r2, lt);
    q_sched_fence();
    q_elim_step<K + 1, NB, N, NC1, NC2, R2LOW, R1ANTI>(q, s, r1, r2, lt, last, dn, awk);
  } else {
    q_elim_scale_rhs<K, NC1, NC2, R2LOW, R1ANTI>(q, aw, r1, r2);
  }
}
template <int N, int NC1, int NC2, bool R2LOW = false, bool R1ANTI = false, typename R> I2C_FN bool q_elim(const Quad<R>& q, R* s, R* r1, R* r2, R* lt, R* awk = nullptr) {
  constexpr int NB = (N + 3) / 4;
  static_assert(!R1ANTI || NC1 == NB, "R1ANTI: a square right-hand side");
  R last = R(0), d[10];
  q_pivot_send<q_live_rows<N, 0>(), NB, 0>(q, s, Q_O_DG, d);
  q_elim_step<0, NB, N, NC1, NC2, R2LOW, R1ANTI>(q, s, r1, r2, lt, &last, d, awk);
  return last > R(0);
}
// two eliminations of the same dimension in lockstep: (sa; ra1, ra2) -> lta and (sb; rb1) -> ltb
template <int K, int NB, int N, int NA1, int NA2, int NB1, typename R>
I2C_FN void q_elim2_step(const Quad<R>& q, R* sa, R* ra1, R* ra2, R* lta, R* lasta, R* sb, R* rb1, R* ltb, R* lastb, R* da, R* db) {
  constexpr int NL = q_live_rows<N, K>();
  R awa, pla, awb, plb;
  q_pivot_settle<NL>(da);
  q_pivot_settle<NL>(db);
  q_pivot_algebra<NL>(q, da, &awa, &pla);
  q_pivot_algebra<NL>(q, db, &awb, &plb);
  if (K == NB - 1) *lasta = pla, *lastb = plb;
  q_elim_scale_lt<K, NB>(q, awa, sa, lta);
  q_elim_scale_lt<K, NB>(q, awb, sb, ltb);
  if constexpr (K + 1 < NB) {
    constexpr int NLN = q_live_rows<N, K + 1>();
    R dna[10], dnb[10];
    q_sched_fence();
    q_elim_next_pivot<K, NB>(q, sa, lta);
    q_elim_next_pivot<K, NB>(q, sb, ltb);
    {  // both next pivot blocks in one LDS round trip
      const auto dga = q.sh + Q_O_DG, dgb = q.sh + Q_O_DG2;
      q.sync();
      dga[4 * q.r + q.c] = sa[(K + 1) * NB + K + 1];
      dgb[4 * q.r + q.c] = sb[(K + 1) * NB + K + 1];
      q.sync();
      q_pivot_fetch<NLN>(dga, dna);
      q_pivot_fetch<NLN>(dgb, dnb);
    }
    q_sched_fence();
    // (interleaved: the scaled rows of one elimination are consumed after the other's independent instructions)
    q_elim_scale_rhs<K, NA1, NA2, true>(q, awa, ra1, ra2);  // (ra2 starts as the identity: see R2LOW)
    q_elim_scale_rhs<K, NB1, 0>(q, awb, rb1, (R*)nullptr);
    q_elim_below<K, NB, NA1, NA2, true>(q, sa, ra1, ra2, lta);
    q_elim_below<K, NB, NB1, 0>(q, sb, rb1, (R*)nullptr, ltb);
    q_sched_fence();
    q_elim2_step<K + 1, NB, N, NA1, NA2, NB1>(q, sa, ra1, ra2, lta, lasta, sb, rb1, ltb, lastb, dna, dnb);
  } else {
    q_elim_scale_rhs<K, NA1, NA2, true>(q, awa, ra1, ra2);
    q_elim_scale_rhs<K, NB1, 0>(q, awb, rb1, (R*)nullptr);
  }
}
template <int N, int NA1, int NA2, int NB1, typename R>
I2C_FN void q_elim2(const Quad<R>& q, R* sa, R* ra1, R* ra2, R* lta, bool* oka, R* sb, R* rb1, R* ltb, bool* okb) {
  constexpr int NB = (N + 3) / 4, NL0 = q_live_rows<N, 0>();
  R lasta = R(0), lastb = R(0), da[10], db[10];
  {
    const auto dga = q.sh + Q_O_DG, dgb = q.sh + Q_O_DG2;
    q.sync();
    dga[4 * q.r + q.c] = sa[0];
    dgb[4 * q.r + q.c] = sb[0];
    q.sync();
    q_pivot_fetch<NL0>(dga, da);
    q_pivot_fetch<NL0>(dgb, db);
  }
  q_elim2_step<0, NB, N, NA1, NA2, NB1>(q, sa, ra1, ra2, lta, &lasta, sb, rb1, ltb, &lastb, da, db);
  *oka = lasta > R(0);
  *okb = lastb > R(0);
}

// observe() without its last outputs (LASTLIN, forward_quad_body): the first NOUT of them
template <class M, typename R, int NOUT> struct ObserveHeadF {
  const R* p;
  I2C_HD inline void operator()(const R* x, const R* sn, const R* cs, R* y) const {
    R yy[M::NZ];
    M::observe(p, x, sn, cs, yy);
#pragma unroll
    for (int k = 0; k < NOUT; ++k) y[k] = yy[k];
  }
};

// ---- sigma points ------------------------------------------------------------------------------------------------------------
// The points m +/- sf L[:, p] of a DIN-dimensional rule through f, one evaluation per lane of the trajectory and pass
// (d <= 8: lane p < 8: +, lane 8 + p: -, one pass; d <= 16: lane p: + in the first pass, - in the second; geometry G = QG<model>).
// Pairs >= DIN have a zero direction: they evaluate the centre, and the last pair row is used as such. Results as pairwise sums
// and differences about a reference value yc, in blocks [pair][output]:  am = (y+ - yc) + (y- - yc),  dm = y+ - y-.
// yc = the centre value when a pair row is spare. When all rows carry points (DIN = 8 or 16), with no weight on the centre (the
// only rules this family serves) the moments do not depend on the reference value, and yc = the midpoint of pair 0 stands in.
// No masks: a pair beyond the input dimension IS a centre evaluation (its sum and difference vanish exactly), and the output
// columns beyond NOUT are written as zeros by every lane.
//   muc: mean, column form [NBI]; lt: L^T, upper blocks [NBI][NBI]
// CENTRE (general cubature weights, round 6): the reference value must BE the centre value f(m). Where every pair row carries a
// point (DIN = the geometry's pair rows: the planar quadrotor, d = 8; the 12-state quadrotor, d = 16) every lane evaluates the centre
// once more, for itself -- the whole output vector lands in its registers and it keeps the entries of its block column: no exchange.
template <class M, class G, int DIN, int NOUT, bool CENTRE = false, class F, typename R>
I2C_FN void q_points(const Quad<R>& q, const R sf, const R* muc, const R* lt, const F& f, R* am, R* dm, R* yc) {
  constexpr int NBI = (DIN + 3) / 4, NBO = (NOUT + 3) / 4, PR = G::PR, LDL = G::LDL, YLD = G::YLD;
  constexpr int NA1 = M::NA > 0 ? M::NA : 1;
  static_assert(DIN <= PR && NOUT <= 12 && 4 * NBI <= PR, "quad kernels: <= 16 inputs, <= 12 evaluated outputs");
  const auto Lr = q.sh + G::O_L, Y = q.sh + G::O_Y, mv = q.sh + G::O_MV;
  q.sync();
#pragma unroll
  for (int i = 0; i < NBI; ++i)
#pragma unroll
    for (int j = 0; j < NBI; ++j) Lr[(4 * i + q.r) * LDL + 4 * j + q.c] = j >= i ? lt[i * NBI + j] : R(0);
#ifdef I2C_HOST_SIM
  if (q.r == 0)  // (lane-threads must not race; on the device the four rows store the same value to the same address)
#endif
  {
#pragma unroll
    for (int j = 0; j < NBI; ++j) mv[4 * j + q.c] = muc[j];
  }
  q.sync();
  const int p = q.p();
#pragma unroll
  for (int pass = 0; pass < (G::WIDE ? 2 : 1); ++pass) {
    const int pp = G::WIDE ? p : (p & 7);
    const bool pt = pp < DIN;  // pairs beyond the input dimension evaluate the centre (their rows may not even be written)
    const bool plus = G::WIDE ? pass == 0 : p < 8;
    const R sg = pt ? (plus ? sf : -sf) : R(0);
    const int lrow = (pt ? pp : 0) * LDL;
    R x[DIN], sn[NA1], cs[NA1], y[NOUT];
#pragma unroll
    for (int i = 0; i < DIN; ++i) x[i] = mv[i] + sg * Lr[lrow + i];
#pragma unroll
    for (int k = 0; k < M::NA; ++k) {
      if constexpr (G::WIDE) r_sincos_sc(x[M::ang(k)], &sn[k], &cs[k]);  // (constants as scalar operands: the registers are full)
      else r_sincos(x[M::ang(k)], &sn[k], &cs[k]);
    }
    f(x, sn, cs, y);
    if (G::WIDE && pass == 1) q.sync();  // the second half of the results overwrites the directions: after their last read
    const int yrow = (G::WIDE ? pass * PR + p : p) * YLD;
#pragma unroll
    for (int k = 0; k < 4 * NBO; ++k) Y[yrow + k] = k < NOUT ? y[k < NOUT ? k : 0] : R(0);
  }
  if constexpr (CENTRE && DIN >= PR) {
    R x[DIN], sn[NA1], cs[NA1], y[NOUT];
#pragma unroll
    for (int i = 0; i < DIN; ++i) x[i] = mv[i];
#pragma unroll
    for (int k = 0; k < M::NA; ++k) {
      if constexpr (G::WIDE) r_sincos_sc(x[M::ang(k)], &sn[k], &cs[k]);
      else r_sincos(x[M::ang(k)], &sn[k], &cs[k]);
    }
    f(x, sn, cs, y);
#pragma unroll
    for (int j = 0; j < NBO; ++j) {
      R v = R(0);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * j + k < NOUT) v = q.c == k ? y[4 * j + k] : v;
      yc[j] = v;
    }
  }
  q.sync();
#pragma unroll
  for (int j = 0; j < NBO; ++j) {
    const int col = 4 * j + q.c;
    R y0;
    if constexpr (DIN < PR) {
      y0 = Y[(PR - 1) * YLD + col];
    } else if constexpr (CENTRE) {
      y0 = yc[j];  // (evaluated below the points, before this loop)
    } else {
      y0 = R(0.5) * (Y[col] + Y[PR * YLD + col]);
    }
    yc[j] = y0;
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      const int row = 4 * i + q.r;
      const R yp = Y[row * YLD + col], ym = Y[(PR + row) * YLD + col];
      am[i * NBO + j] = (yp - y0) + (ym - y0);
      dm[i * NBO + j] = yp - ym;
    }
  }
}
// Moments from the pairwise sums / differences (quadrature.py:34-44 in the form of sp_transform): mean (column form), the
// centred sums (am is overwritten), and sy [NBO][NBO] upper blocks = wi/2 (am^T am + dm^T dm) (+ nothing: the caller adds noise)
// GENERAL (round 5): any CubatureQuadrature(alpha, beta, kappa) -- a weight on the centre point and weights that need not sum to
// one (W = sum of weights_sig = 2 - alpha^2 + beta; the reference's mean AND covariance use weights_sig, quadrature.py:36,49). With
// y0 the centre value, A = sum_p a_p:   my = W y0 + wi A,
//   sig_y = wi/2 sum_p (a_p a_p^T + d_p d_p^T) - wi^2 A A^T + (W - W^2) y0 y0^T + wi (1 - W) (y0 A^T + A y0^T)
// (the last two terms vanish for W = 1; 2 d wi = 1 -- lam = 0 -- is what lets the unit form centre the a_p instead of the
// explicit wi^2 A A^T). Needs the centre EVALUATED: a spare pair row (DIN < the geometry's pair rows).
template <int DIN, int NOUT, bool GENERAL = false, typename R>
I2C_FN void q_moments(const Quad<R>& q, const Rule<R>& rule, R* am, const R* dm, const R* yc, R* myc, R* sy) {
  constexpr int NBI = (DIN + 3) / 4, NBO = (NOUT + 3) / 4;
  const R wi = rule.wi, hw = R(0.5) * wi;
  const R rm = (4 * (NBI - 1) + q.r < DIN) ? R(1) : R(0);  // live pairs of the last pair block
  R acol[NBO];
#pragma unroll
  for (int j = 0; j < NBO; ++j) {
    R t = am[j];
#pragma unroll
    for (int i = 1; i < NBI; ++i) t += am[i * NBO + j];
    const R asum = q_colsum(q, t);
    acol[j] = asum;
    if constexpr (GENERAL) {
      myc[j] = rule.W * yc[j] + wi * asum;
    } else {
      myc[j] = yc[j] + wi * asum;
      // sig_y = wi/2 sum_p (a_p a_p^T + d_p d_p^T) - wi^2 A A^T: with 2 d wi = 1 the last term centres the a_p
      const R amean = (R(2) * wi) * asum;
#pragma unroll
      for (int i = 0; i < NBI; ++i) am[i * NBO + j] = (i < NBI - 1 || DIN % 4 == 0) ? am[i * NBO + j] - amean : am[i * NBO + j] - amean * rm;
    }
  }
#pragma unroll
  for (int k = 0; k < NBO * NBO; ++k) sy[k] = R(0);
  q_tn<NBI, NBO, NBO, false, true>(q, am, am, sy);
  q_tn<NBI, NBO, NBO, false, true>(q, dm, dm, sy);
#pragma unroll
  for (int k = 0; k < NBO * NBO; ++k) sy[k] *= hw;
  if constexpr (GENERAL) {
    const R W = rule.W, c_yy = W - W * W, c_ya = wi * (R(1) - W), c_aa = -(wi * wi);
    R arow[NBO], y0row[NBO];
#pragma unroll
    for (int i = 0; i < NBO; ++i) {
      arow[i] = q_tr(q, acol[i]);
      y0row[i] = q_tr(q, yc[i]);
    }
#pragma unroll
    for (int i = 0; i < NBO; ++i)
#pragma unroll
      for (int j = i; j < NBO; ++j)
        sy[i * NBO + j] += (c_aa * arow[i] + c_ya * y0row[i]) * acol[j] + (c_yy * y0row[i] + c_ya * arow[i]) * yc[j];
  }
}

// ---- Kalman-style updates ------------------------------------------------------------------------------------------------------
// Update of N(mu, s) (dimension N; mu column form [NB], s upper blocks [NB][NB]) on an observation with predicted mean mzc
// (column form [NBZ]), covariance sz INCLUDING its noise (upper blocks, consumed), cross-covariance szx = cov(z, x) [NBZ][NB]
// and target ztc:  with C = chol(sz), [U | q] = C^-1 [szx | zt - mz]:  s <- s - U^T U,  mu <- mu + U^T q   (i2c.py:398-403).
// The innovation rides as one more column of the right-hand side -- the spare column N of the last block column when N is not
// a multiple of 4, a block column of its own otherwise -- so the mean update is a by-product of the same instructions.
template <int N, int NZ, typename R> I2C_FN bool q_kalman(const Quad<R>& q, R* muc, R* s, const R* mzc, R* sz, const R* szx, const R* ztc) {
  constexpr int NB = (N + 3) / 4, NBZ = (NZ + 3) / 4;
  constexpr bool SPARE = N % 4 != 0;
  constexpr int NC = SPARE ? NB : NB + 1, QC = SPARE ? N % 4 : 0;  // right-hand side block columns; the innovation's column in the last
  R u[NBZ * NC], lt[NBZ * NBZ];
#pragma unroll
  for (int k = 0; k < NBZ; ++k) {
    const R rr = q_tr(q, ztc[k] - mzc[k]);  // row form
#pragma unroll
    for (int j = 0; j < NB; ++j) u[k * NC + j] = szx[k * NB + j];
    if constexpr (SPARE) {
      u[k * NC + NB - 1] = q.c == QC ? rr : u[k * NC + NB - 1];
    } else {
      u[k * NC + NC - 1] = q.c == QC ? rr : R(0);
    }
  }
  const bool ok = q_elim<NZ, NC, 0>(q, sz, u, (R*)nullptr, lt);
  R inc[NB];  // -(U^T q), row form in column QC
  if constexpr (SPARE) {
    q_tn<NBZ, NB, NB, true, true>(q, u, u, s);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      inc[i] = s[i * NB + NB - 1];
      s[i * NB + NB - 1] = q.c == QC ? R(0) : s[i * NB + NB - 1];  // (s had zeros there; the last block also gets -|q|^2 at (N, N))
    }
  } else {
    R un[NBZ * NB], e[NB];
#pragma unroll
    for (int k = 0; k < NBZ; ++k)
#pragma unroll
      for (int j = 0; j < NB; ++j) un[k * NB + j] = u[k * NC + j];
    q_tn<NBZ, NB, NB, true, true>(q, un, un, s);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      e[i] = R(0);
#pragma unroll
      for (int k = 0; k < NBZ; ++k) q_mfma(q, -u[k * NC + i], u[k * NC + NC - 1], e[i]);
      inc[i] = e[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const R ir = q_bcq<QC>(q, inc[i]);                                  // row form, every column
    const R ic = q_tr(q, (4 * i + q.r < N) ? ir : R(0));                // column form
    muc[i] -= ic;
  }
  return ok;
}
// The same update on an IDENTITY observation of the state itself with noise alpha * xi and target zt (i2c.py:394-403 with z = the
// state):  with C = chol(s + alpha xi), U = C^-1 s:  s <- s - U^T U;  the mean uses the posterior-covariance form of the same gain,
// s (s + N)^-1 = s_new N^-1  (N^-1 = W / alpha, W = the cost weight):  mu <- mu + s_new W (zt - mu) / alpha   (see w_kalman).
// s: FULL blocks in, full blocks out. xi_m, w_m: QLD x QLD row-major constants in LDS.
template <int N, int QLD, typename R, class P>
I2C_FN bool q_kalman_identity(const Quad<R>& q, const R alpha, const P xi_m, const P w_m, const bool w_diag, const R* ztc, R* muc, R* s, const int kz = 0) {
  constexpr int NB = (N + 3) / 4;
  R sz[NB * NB], u[NB * NB], lt[NB * NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      sz[i * NB + j] = s[i * NB + j] + alpha * q_ldc<QLD>(q, xi_m, i, j, kz);
      u[i * NB + j] = s[i * NB + j];
    }
  const bool ok = q_elim<N, NB, 0>(q, sz, u, (R*)nullptr, lt);
  q_tn<NB, NB, NB, true, true>(q, u, u, s);  // the upper blocks, then their mirror images (a transpose instead of NB products each)
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) s[i * NB + j] = q_tr(q, s[j * NB + i]);
  // wr = W (zt - mu), row form
  R wr[NB];
  if (w_diag) {
#pragma unroll
    for (int j = 0; j < NB; ++j) wr[j] = q_tr(q, w_m[(4 * j + q.c) * QLD + 4 * j + q.c + kz] * (ztc[j] - muc[j]));
  } else {
    R rr[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) rr[j] = q_tr(q, ztc[j] - muc[j]);
#pragma unroll
    for (int i = 0; i < NB; ++i) {  // (W r)[i], W symmetric: sum_k W[k][i] r[k]
      wr[i] = R(0);
#pragma unroll
      for (int k = 0; k < NB; ++k) q_mfma(q, q_ldc<QLD>(q, w_m, k, i, kz), rr[k], wr[i]);
    }
  }
  const R ia = r_rcp(alpha);
#pragma unroll
  for (int i = 0; i < NB; ++i) {  // (s wr)[i] = sum_k s[k][i] wr[k] (s symmetric), row form; back to column form
    R t = R(0);
#pragma unroll
    for (int k = 0; k < NB; ++k) q_mfma(q, s[k * NB + i], wr[k], t);
    muc[i] += q_tr(q, t) * ia;
  }
  return ok;
}

// The identity-observation update under GENERAL cubature weights (round 6): z = the N-vector itself, so the rule's moments are exact
// (quadrature.py:34-44 with y_p = x_p and 2 wi sf^2 = 1):  mz = W m,  sig_z = S + (W - W^2) m m^T,  cov(z, x) = S  -- the W = 1 forms
// above are the special case. Through the general update (q_kalman). s: upper blocks (the diagonal ones full), in and out.
template <int N, int QLD, typename R, class P>
I2C_FN bool q_kalman_identity_general(const Quad<R>& q, const R W, const R alpha, const P xi_m, const R* ztc, R* muc, R* s, const int kz = 0) {
  constexpr int NB = (N + 3) / 4;
  const R cww = W - W * W;
  R mz[NB], mr[NB], sz[NB * NB], szx[NB * NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    mz[i] = W * muc[i];
    mr[i] = q_tr(q, muc[i]);  // row form
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      szx[i * NB + j] = j >= i ? s[i * NB + j] : q_tr(q, s[j * NB + i]);
      sz[i * NB + j] = j >= i ? s[i * NB + j] + cww * (mr[i] * muc[j]) + alpha * q_ldc<QLD>(q, xi_m, i, j, kz) : R(0);
    }
  return q_kalman<N, N>(q, muc, s, mz, sz, szx, ztc);
}

// The identity-observation update in square-root form (round 5; the mathematics in w_kalman_sqrt, i2c_wave.hpp): the prior arrives
// as l0 = chol(s)^T (upper blocks), and with N^-1 = W / alpha
//   s_new = L (I + L^T N^-1 L)^-1 L^T,   chol(s_new) = L U^-T   for   I + L^T N^-1 L = U U^T  (U upper triangular),
// computed as the ordinary elimination of the index-reversed matrix M' = I + X^T N^-1 X, X = L J (J the exchange matrix), with the
// right-hand side J L^T: its result Y = J chol(s_new)^T holds the rows of the new factor in reverse order. Index reversal in blocks:
// block (I, P) of X is block (NB-1-P, I) of L^T transposed with its columns reversed, block (P, I) of J L^T the same block with its
// rows reversed -- one matrix instruction each, with the 4 x 4 exchange matrix as the other operand. Out: z = the block rows of Y
// in natural order (upper blocks; rows inside a block reversed: a set of sigma-point directions has no order), s = Y^T Y (upper
// blocks), the mean as in q_kalman_identity. Saves the factorisation of s + N AND that of s_new: N / 4 + 1 pivot blocks (with the
// caller's action block) instead of 2 N / 4. N a multiple of 4 (the reversal must not move padding to the front).
template <int N, int QLD, typename R, class P>
I2C_FN bool q_kalman_sqrt(const Quad<R>& q, const R alpha, const P w_m, const bool w_diag, const R* ztc, R* muc, const R* l0, R* z, R* s, const int kz = 0) {
  static_assert(N % 4 == 0, "q_kalman_sqrt: whole blocks");
  constexpr int NB = N / 4;
  const R j4 = (q.r + q.c == 3) ? R(1) : R(0);
  const R ia = r_rcp(alpha);
  R x[NB * NB], rr[NB * NB], yw[NB * NB], m[NB * NB], ltm[NB * NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      x[i * NB + p] = R(0);
      rr[p * NB + i] = R(0);
      if (p < NB - 1 - i) continue;  // block (NB-1-p, i) of L^T is below the diagonal
      q_mfma(q, l0[(NB - 1 - p) * NB + i], j4, x[i * NB + p]);   // (L J)(i, p) = block^T J4
      q_mfma(q, j4, l0[(NB - 1 - p) * NB + i], rr[p * NB + i]);  // (J L^T)(p, i) = J4 block
    }
  if (w_diag) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const R wi = w_m[(4 * i + q.r) * QLD + 4 * i + q.r + kz] * ia;
#pragma unroll
      for (int p = 0; p < NB; ++p) yw[i * NB + p] = wi * x[i * NB + p];
    }
  } else {
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        R t = R(0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (p < NB - 1 - k) continue;
          q_mfma(q, q_ldc<QLD>(q, w_m, k, i, kz), x[k * NB + p], t);  // W symmetric: block (i, k) = block (k, i)^T
        }
        yw[i * NB + p] = t * ia;
      }
  }
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      m[a * NB + b] = (a == b && q.r == q.c) ? R(1) : R(0);
      if (b < a) continue;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (a < NB - 1 - i) continue;
        q_mfma(q, x[i * NB + a], yw[i * NB + b], m[a * NB + b]);
      }
    }
  const bool ok = q_elim<N, NB, 0, false, true>(q, m, rr, (R*)nullptr, ltm);
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      s[a * NB + b] = R(0);
      z[a * NB + b] = b >= a ? rr[(NB - 1 - a) * NB + b] : R(0);
      if (b < a) continue;
#pragma unroll
      for (int p = 0; p < NB; ++p) {
        if (a < NB - 1 - p) continue;
        q_mfma(q, rr[p * NB + a], rr[p * NB + b], s[a * NB + b]);
      }
    }
  // wr = W (zt - mu), row form
  R wr[NB];
  if (w_diag) {
#pragma unroll
    for (int j = 0; j < NB; ++j) wr[j] = q_tr(q, w_m[(4 * j + q.c) * QLD + 4 * j + q.c + kz] * (ztc[j] - muc[j]));
  } else {
    R rv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) rv[j] = q_tr(q, ztc[j] - muc[j]);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      wr[i] = R(0);
#pragma unroll
      for (int k = 0; k < NB; ++k) q_mfma(q, q_ldc<QLD>(q, w_m, k, i, kz), rv[k], wr[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {  // (s wr)[i] = sum_k s[i][k] wr[k]: the A operand is the transpose of what is passed
    R t = R(0);
#pragma unroll
    for (int k = 0; k < NB; ++k) q_mfma(q, k <= i ? s[k * NB + i] : q_tr(q, s[i * NB + k]), wr[k], t);
    muc[i] += q_tr(q, t) * ia;
  }
  return ok;
}

// One cell block of a per-cell buffer for the lanes of a wave. An element index is split into a per-lane part and a part that is
// the same for every lane (a compile-time constant after unrolling): e = lane_e + k.
//   TM (trajectory-major [B][E], the wave-capable models): byte offset = (lane_e * sizeof(S) + b E sizeof(S)) + k * sizeof(S) -- ONE
//   register per distinct lane part (and store predicate), k in the instruction's immediate offset: the 16 x 16 blocks of the
//   12-state quadrotor are addressed from ~a dozen registers instead of one per (block, lane) element;
//   [E][B] rows otherwise: byte offset = (lane_e + k) * B sizeof(S) + b sizeof(S).
// Masked-off lanes of a store park their offset out of the window (dropped by the buffer unit, see GIO::st_if).
template <typename R, typename S, bool TM> struct QIO {
  Window w;
  unsigned rb, bo;
  // TM: the block part goes into the instruction's SCALAR offset, kept apart from the lane part by an opaque move: added to the
  // lane part, hipcc distributes it over the predicate's select and hoists one register PER (block, predicate) out of the time
  // loop (they were spilled to scratch for d = 16); an s_mov per access costs nothing where waves share a SIMD
  I2C_MEM R ld(const int lane_e, const int k) const {
    if constexpr (TM) return (R)wld<S>(w, opaque_uniform((unsigned)k * (unsigned)sizeof(S)), (unsigned)lane_e * (unsigned)sizeof(S) + bo);
    else return (R)wld<S>(w, 0u, (unsigned)(lane_e + k) * rb + bo);
  }
  I2C_MEM void st_if(const bool on, const int lane_e, const int k, const R v) const {
#ifdef I2C_HOST_SIM
    if (on) {
      if constexpr (TM) wst(w, (unsigned)k * (unsigned)sizeof(S), (unsigned)lane_e * (unsigned)sizeof(S) + bo, (S)v);
      else wst(w, 0u, (unsigned)(lane_e + k) * rb + bo, (S)v);
    }
#else
    if constexpr (TM) wst(w, opaque_uniform((unsigned)k * (unsigned)sizeof(S)), on ? (unsigned)lane_e * (unsigned)sizeof(S) + bo : 0x80000000u, (S)v);
    else wst(w, 0u, on ? (unsigned)(lane_e + k) * rb + bo : 0x80000000u, (S)v);
#endif
  }
};

// ------------------------------------------------------------------------------------------
// Forward sweep (i2c.py:876-880 over :350-447)
// ------------------------------------------------------------------------------------------
// The cell is re-ordered around its factorisations (same mathematics):
//   * chol(sig_x3_f) -- needed for the smoother gain J of the cell that produced it -- and chol(P_xx + sig_x3_f) -- the pdf ratio of
//     the NEXT cell's feedback prior -- factor two matrices that are known at the same moment, so they run as ONE pair of
//     eliminations at the top of the next cell (q_elim2), J being stored a cell late; a last single elimination after the loop;
//   * the prior joint is never factored: with Lt3 = chol(sig_x3_f)^T from that pair,  chol(sig_0)^T = [[Lt3, Lt3 Kt^T], [0, chol(S_u|x)^T]],
//     S_u|x = P_uu - Kt P_xu (as the lane kernels do for d <= 5), and sig_0 itself is the product of that factor with its
//     transpose -- no F sig_x F^T products, no d x d factorisation;
//   * feed-forward cells (i2c.py:355-360) are the same arithmetic with Kt = 0.
// `live`: trajectory slot b holds a real trajectory (the last wave of a batch that is not a multiple of four repeats its last
// one in the spare slots: every lane of a wave takes part in the matrix instructions; nothing is stored for them)
// GENERAL: cubature weights with lam != 0 / weights that do not sum to one (q_moments), for every model of the d <= 8 geometry
// (quad_general_exists; round 5: the models whose observations all go through sigma points; round 6: the identity-observation
// models too -- their moments are exact, mean W m, covariance S + (W - W^2) m m^T, cross-covariance S, and go through the general
// Kalman-style update instead of the factor form, which is the W = 1 case; a model whose 2 d points fill the sixteen lanes of a
// trajectory evaluates the centre in an extra pass, q_points<CENTRE>).
template <class M> constexpr bool quad_general_exists() {
  constexpr int NT = M::NZT > 0 ? M::NZT : 1;
  return !QG<M>::WIDE || (st_identity<ObsStruct<M>, M::NZ>() && M::NZ == M::NX + M::NU && (M::NZT == 0 || (st_identity<TermStruct<M>, NT>() && NT == M::NX)));
}
#ifndef I2C_QUAD_SQRT_ID
#define I2C_QUAD_SQRT_ID 1  // (A/B knob: 0 = the covariance form of the identity-observation update, q_kalman_identity)
#endif
template <class M, typename R, typename S, bool GENERAL = false, class KC>
I2C_HD inline void forward_quad_body(const Consts<M, R>& c, const KC& kc, const FwdArgs<R, S>& a, const int b, const bool live, const Quad<R>& qw) {
  using C = Consts<M, R>;
  const Quad<R>& q = qw;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D, NT = C::NZT1;
  constexpr int NBX = (NX + 3) / 4, NBD = (D + 3) / 4, NBZ = (NZ + 3) / 4, NBT = (NT + 3) / 4;
  using G = QG<M>;
  constexpr int QLD = G::QLD;
  static_assert(D <= 16 && NZ <= 16 && NZT <= 16, "quad kernels: d <= 16");
  constexpr bool OBS_ID = st_identity<ObsStruct<M>, NZ>() && NZ == D;
  constexpr bool TERM_ID = NZT > 0 && st_identity<TermStruct<M>, NT>() && NT == NX;
  // identity observation of a joint made of whole blocks (12-state and planar quadrotor): the update works on the factor
  constexpr bool SQRT_ID = I2C_QUAD_SQRT_ID && OBS_ID && D % 4 == 0 && !GENERAL;  // (the factor form is the W = 1 case)
  constexpr bool CENTRE = GENERAL && D >= QG<M>::PR;  // no spare pair row for the centre point: an extra evaluation (q_points)
  // the last observation output as a scalar pre-elimination (see stage 2): a pass-through of a coordinate of the joint's last block
  // that would otherwise be a block row of its own
#ifndef I2C_QUAD_LASTLIN
#define I2C_QUAD_LASTLIN 1
#endif
  constexpr int JZL = M::obs_lin(NZ - 1);
  // (unit weights only: with W != 1 the moments of a pass-through output carry (W - W^2) m^2 terms, q_moments)
  constexpr bool LASTLIN = I2C_QUAD_LASTLIN && !GENERAL && !OBS_ID && NZ % 4 == 1 && NZ > 4 && JZL >= 0 && JZL / 4 == (D + 3) / 4 - 1;
  static_assert(!GENERAL || quad_general_exists<M>(), "general cubature weights: the d <= 8 geometry, or d = 16 with identity observations");
  static_assert(OBS_ID || D % 4 != 0, "quad kernels: a general observation needs a spare column in the joint's last block");
  static_assert(OBS_ID || NU == 1, "quad kernels: a general observation with one action (the factor of S_u|x is a square root)");
  constexpr int JU = NX / 4, CU = NX % 4;  // the block (row and column) and the in-block offset where the action entries start
  static_assert(CU + NU <= 4, "quad kernels: the action entries live in one block");
  constexpr bool SPX = NX % 4 != 0;        // a spare column in the state blocks: the pdf-ratio right-hand side rides in it
  constexpr int NCR = SPX ? 0 : 1;
  constexpr int O_K = D + sym(D), O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  const int r = q.r, cc = q.c;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned WS = sizeof(S), bo = (unsigned)b * WS, rb = (unsigned)(B * WS), bo8 = (unsigned)(b * sizeof(R));
  const Rule<R>& rule = c.rule_xu;
  int fail = 0;
  auto xrow = [&](const int i) { return 4 * i + r; };  // matrix row / column of this lane in block row i / block column j
  auto xcol = [&](const int j) { return 4 * j + cc; };
  // one cell block of a per-cell buffer for this lane's trajectory: [E][B] rows, or trajectory-major [B][E] (the posterior / prior
  // buffers of the wave-capable models, Consts::post_tm; the forward messages when the wave kernels read them, Consts::fwd_tm)
  constexpr bool TM = G::WIDE;  // (Impl::quad_supported: the d = 16 form is only chosen when both buffers are trajectory-major)
  auto cell_io = [&](const S* cell, const int E) {
    if constexpr (TM) return QIO<R, S, TM>{make_window(cell, (unsigned long)E * B * WS), WS, (unsigned)b * (unsigned)E * WS};
    else return QIO<R, S, TM>{make_window(cell, (unsigned long)E * rb), rb, bo};
  };
  // packed-symmetric element (row, col) of block (i, j), i <= j, split into its per-lane and its per-block part:
  //   i < j (row <= col):  col (col + 1) / 2 + row  =  [c (c + 1) / 2 + r + 4 j c]  +  [8 j^2 + 2 j + 4 i]
  //   i = j, read as a full symmetric block (hi / lo = the larger / smaller of r, c):  [hi (hi + 1) / 2 + lo + 4 j hi]  +  [8 j^2 + 6 j]
  const int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
  const int tri_c = cc * (cc + 1) / 2 + r, tri_d = hi * (hi + 1) / 2 + lo;
  auto sym_lane = [&](const int i, const int j) { return i == j ? tri_d + 4 * j * hi : tri_c + 4 * j * cc; };
  auto sym_k = [&](const int i, const int j) { return i == j ? 8 * j * j + 6 * j : 8 * j * j + 2 * j + 4 * i; };
  // is (row of block row i, column of block column j) inside an N x N matrix? (folds to `true` for full blocks)
  auto in_row = [&](const int i, const int N) { return 4 * i + 3 < N || 4 * i + r < N; };
  auto in_col = [&](const int j, const int N) { return 4 * j + 3 < N || 4 * j + cc < N; };
  auto in_n = [&](const int i, const int j, const int N) { return in_row(i, N) && in_col(j, N); };
  // lane masks as multipliers (everything they multiply is finite or belongs to a trajectory that has failed anyway)
  const R m_ucol = (cc >= CU && cc < CU + NU) ? R(1) : R(0);                   // action columns of block column JU
  const R m_uu = (cc >= CU && cc < CU + NU && r >= CU && r < CU + NU) ? R(1) : R(0);

  // state message carried along the chain: mean in column form, covariance in UPPER blocks (nx x nx, zero-padded)
  R mx[NBX], sx[NBX * NBX], sxy[NBX * NBD];
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    const int col = xcol(j);
    const R v = a.x0[(long)(col < NX ? col : 0) * B + b];
    mx[j] = col < NX ? v : R(0);
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
      const int row = xrow(i);
      const bool in = row < NX && col < NX;
      const R sv = a.sig_x0[(long)w_symidx(in ? row : 0, in ? col : 0) * B + b];
      sx[i * NBX + j] = in ? sv : R(0);
    }
  }
#pragma unroll
  for (int k = 0; k < NBX * NBD; ++k) sxy[k] = R(0);
  bool j_pending = false;  // (wave-uniform) the smoother gain of the previous cell still has to be formed and stored

  // The prior rows of a cell do not depend on the recursion: they are fetched ONE CELL AHEAD (see forward_wave_body)
  R nx_pmu[NBD], nx_pj[NBD * NBD], nx_kt[NBX], nx_alpha, nx_zt[NBZ];
  int nx_ff;
  const Window ffw = make_window(a.ff, (unsigned long)T);
  const Window alw = make_window(a.alpha_cell ? a.alpha_cell : a.alpha, (a.alpha_cell ? (unsigned long)T : 1ul) * B * sizeof(R));
  // per-cell targets, or a discarded dummy (rows of sig_x0: distinct addresses, so that the loads stay independent instructions):
  // branch-free buffer loads -- a load behind a run-time branch, or one the compiler can merge with another and then COPY, puts an
  // s_waitcnt vmcnt(0) right behind the prefetch (measured here: 2 us of every 5 us cell)
  static_assert(NBZ <= sym(NX), "dummy rows of the target prefetch");
  const Window zw = make_window(c.z_per_cell ? a.z : a.sig_x0, (c.z_per_cell ? (unsigned long)T * NZ : (unsigned long)sym(NX)) * B * sizeof(R));
  unsigned zlane[NBZ];
#pragma unroll
  for (int j = 0; j < NBZ; ++j) {
    const int col = xcol(j);
    zlane[j] = (unsigned)((((unsigned long)(c.z_per_cell ? (col < NZ ? col : 0) : j)) * B + b) * sizeof(R));
  }
  const unsigned zcell = c.z_per_cell ? (unsigned)((unsigned long)NZ * B * sizeof(R)) : 0u, acell = a.alpha_cell ? (unsigned)(B * sizeof(R)) : 0u;
  auto fetch_prior = [&](const int tc) {
    const int trc = c.row(tc);
    const QIO<R, S, TM> pri = cell_io(a.prior + (unsigned long)trc * C::E_POST * B, C::E_POST);
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      nx_pmu[j] = pri.ld(in_col(j, D) ? cc : -4 * j, 4 * j);
#pragma unroll
      for (int i = 0; i <= j; ++i)  // upper blocks (the diagonal ones as full symmetric blocks)
        nx_pj[i * NBD + j] = pri.ld(in_n(i, j, D) ? sym_lane(i, j) : -sym_k(i, j), D + sym_k(i, j));
    }
#pragma unroll
    for (int i = 0; i < NBX; ++i) {  // K^T in block column JU: row = state index, column = action
      const bool in = in_row(i, NX) && cc >= CU && cc < CU + NU;
      nx_kt[i] = pri.ld(in ? (cc - CU) * NX + r : -4 * i, O_K + 4 * i);
    }
    nx_alpha = wld<R>(alw, (unsigned)trc * acell, bo8);
#pragma unroll
    for (int j = 0; j < NBZ; ++j) nx_zt[j] = wld<R>(zw, (unsigned)trc * zcell, zlane[j]);
    nx_ff = (int)wld_u8(ffw, (unsigned)trc);
  };
  // d <= 8: a cell ahead. d = 16: at the top of the cell itself -- the 22 prefetched doubles would sit on top of a register peak that
  // already fills the 256 registers two waves per SIMD leave each (the second wave covers the round trip)
  constexpr bool PREFETCH = !G::WIDE;
  fetch_prior(0);
  // settled before the loop (see forward_wave_body: loads pending on the loop-entry path cost a vmcnt(0) in every cell)
  nx_alpha = opaque(nx_alpha);
  nx_ff = (int)opaque((unsigned)nx_ff);
#pragma unroll
  for (int k = 0; k < NBD; ++k) nx_pmu[k] = opaque(nx_pmu[k]);
#pragma unroll
  for (int i = 0; i < NBD; ++i)
#pragma unroll
    for (int j = i; j < NBD; ++j) nx_pj[i * NBD + j] = opaque(nx_pj[i * NBD + j]);
#pragma unroll
  for (int k = 0; k < NBX; ++k) nx_kt[k] = opaque(nx_kt[k]);
#pragma unroll
  for (int k = 0; k < NBZ; ++k) nx_zt[k] = opaque(nx_zt[k]);
#pragma unroll
  for (int k = 0; k < NBX; ++k) mx[k] = opaque(mx[k]);
#pragma unroll
  for (int i = 0; i < NBX; ++i)
#pragma unroll
    for (int j = i; j < NBX; ++j) sx[i * NBX + j] = opaque(sx[i * NBX + j]);

  // smoother gain J = sig_xy sig_x3^-1 (i2c.py:423-425) of cell tj from  w3 = W = chol(sig_x3)^-1  and  yj = W sig_xy^T:  J^T = W^T yj
  auto store_gain = [&](const int tj, const R* w3, const R* yj) {
    const QIO<R, S, TM> oj = cell_io(a.fwd + (unsigned long)tj * C::E_FWD * B, C::E_FWD);
    R jt[NBX * NBD];
#pragma unroll
    for (int k = 0; k < NBX * NBD; ++k) jt[k] = R(0);
#pragma unroll
    for (int i = 0; i < NBX; ++i)
#pragma unroll
      for (int j = 0; j < NBD; ++j)
#pragma unroll
        for (int k = i; k < NBX; ++k) q_mfma(q, w3[k * NBX + i], yj[k * NBD + j], jt[i * NBD + j]);  // (W is lower triangular: blocks k >= i)
#pragma unroll
    for (int i = 0; i < NBX; ++i)
#pragma unroll
      for (int j = 0; j < NBD; ++j)  // J[col][row] = J^T[row][col]: element (4 j + c) nx + 4 i + r
        oj.st_if(live && in_row(i, NX) && in_col(j, D), cc * NX + r, O_J + 4 * j * NX + 4 * i, jt[i * NBD + j]);
  };
  auto identity_x = [&](R* w3) {
#pragma unroll
    for (int i = 0; i < NBX; ++i)
#pragma unroll
      for (int j = 0; j < NBX; ++j) w3[i * NBX + j] = (i == j && r == cc && xrow(i) < NX) ? R(1) : R(0);
  };

#if defined(I2C_QUAD_STAMPS) && !defined(I2C_HOST_SIM)
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < T; ++t) {
    const QIO<R, S, TM> out = cell_io(a.fwd + (unsigned long)t * C::E_FWD * B, C::E_FWD);
    if (!PREFETCH && t > 0) fetch_prior(t);
    const int kz = (int)opaque_uniform(0u);  // (see q_ldc)
    // d = 16: the lane's block coordinates made opaque once per cell. Every 0 / 1 lane mask of the cell (one-hot selectors of the
    // pivot algebra, identity blocks, triangle masks: ~25 fp64 values) is otherwise hoisted out of the time loop and pinned in
    // registers the cell needs for its 16 x 16 blocks -- they were spilled to scratch; recomputing one is a compare and a select
    const Quad<R> q = G::WIDE ? q_opaque(qw) : qw;
    const R alpha = nx_alpha;
    const bool ff = w_uniform(nx_ff) != 0;
    R pmu[NBD], pj[NBD * NBD], kt[NBX], zt[NBZ];
#pragma unroll
    for (int k = 0; k < NBD; ++k) pmu[k] = nx_pmu[k];
#pragma unroll
    for (int i = 0; i < NBD; ++i)
#pragma unroll
      for (int j = 0; j < NBD; ++j) pj[i * NBD + j] = j >= i ? nx_pj[i * NBD + j] : R(0);
#pragma unroll
    for (int k = 0; k < NBX; ++k) kt[k] = nx_kt[k];
#pragma unroll
    for (int j = 0; j < NBZ; ++j) zt[j] = xcol(j) < NZ ? (c.z_per_cell ? nx_zt[j] : q_ldv(q, kc.zg, j, kz)) : R(0);
    int cell_bad = 0;

    // ---- 0. the pair of factorisations of the incoming state covariance -----------------------
    //   chol(sig_x) with right-hand sides [sig_xy^T of the previous cell | I]  ->  Lt3, and J of the previous cell;
    //   chol(P_xx + sig_x) with delta = mu_x - P_mu_x  ->  the pdf ratio rho = exp(-delta^T (P_xx + sig_x)^-1 delta / 2) (i2c.py:369-374),
    //   y = L^-1 delta riding as the spare column nx of the sum when nx is not a multiple of 4, a right-hand side otherwise
    R l3[NBX * NBX], dr[NBX];
    R rho;
    {
      R w3[NBX * NBX], yj[NBX * NBD], sa[NBX * NBX], sm[NBX * NBX], lts[NBX * NBX], rhs[NBX];
      identity_x(w3);
#pragma unroll
      for (int k = 0; k < NBX * NBD; ++k) yj[k] = sxy[k];
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        const R dl = xcol(i) < NX ? mx[i] - pmu[i] : R(0);
        dr[i] = q_tr(q, dl);  // row form
      }
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
#pragma unroll
        for (int j = i; j < NBX; ++j) {
          const bool xx = xrow(i) < NX && xcol(j) < NX;
          sa[i * NBX + j] = sx[i * NBX + j];
          sm[i * NBX + j] = xx ? pj[i * NBD + j] + sx[i * NBX + j] : R(0);
        }
        const R dq = (cc == (SPX ? CU : 0) && xrow(i) < NX) ? dr[i] : R(0);
        if constexpr (SPX) sm[i * NBX + NBX - 1] += dq;  // (that column of the sum is zero)
        rhs[i] = dq;
      }
      bool ok3, oks;
      q_elim2<NX, NBD, NBX, NCR>(q, sa, yj, w3, l3, &ok3, sm, rhs, lts, &oks);
      if (j_pending) store_gain(t - 1, w3, yj);
      // a prediction covariance that is not positive definite is the PREVIOUS cell's failure (its stage 4); the state message of
      // cell 0 is the caller's: reported as the prior joint's
      if (t > 0) fail = fold_cell_failure(fail, flag_stage(0, ok3, 4), t - 1);
      else cell_bad = flag_stage(cell_bad, ok3, 1);
      cell_bad = flag_stage(cell_bad, oks || ff, 0);
      R ysq = R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        const R yv = SPX ? lts[i * NBX + NBX - 1] : rhs[i];
        ysq += (cc == (SPX ? CU : 0) && xrow(i) < NX) ? yv * yv : R(0);
      }
      const R maha = q_bcq<(SPX ? CU : 0)>(q, q_colsum(q, ysq));
      rho = ff ? R(0) : (G::WIDE ? r_exp_sc(R(-0.5) * maha) : r_exp(R(-0.5) * maha));  // feed-forward: independent action prior (i2c.py:355-360) = the feedback form with Kt = 0
    }
    // ---- 1. joint prior over (x, u) (i2c.py:361-387): mean column form, covariance UPPER blocks, factor from Lt3 ----
    R mu0[NBD], s0[NBD * NBD], lt0[NBD * NBD];
    {
      // Kt^T = rho K^T in block column JU (rows = state index); G = Lt3 Kt^T
      R ktm[NBX], g[NBX], l3t[NBX * NBX];
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        ktm[i] = (xrow(i) < NX ? m_ucol : R(0)) * (rho * kt[i]);
        g[i] = R(0);
      }
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int k = i; k < NBX; ++k) {
          l3t[i * NBX + k] = q_tr(q, l3[i * NBX + k]);
          q_mfma(q, l3t[i * NBX + k], ktm[k], g[i]);  // (A operand = the transpose of what is passed: Lt3[i][k] itself)
        }
      // Lt0x = [Lt3 | G]: the state rows of chol(sig_0)^T
      R lx[NBX * NBD];
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int j = 0; j < NBD; ++j) {
          const R l3v = (j < NBX && j >= i) ? l3[i * NBX + (j < NBX ? j : 0)] : R(0);
          lx[i * NBD + j] = j == JU ? l3v + g[i] : l3v;
        }
      // S_u|x = P_uu - Kt P_xu in the action entries of block (JU, JU)
      R suu = m_uu * pj[JU * NBD + JU];
#pragma unroll
      for (int k = 0; k < NBX; ++k) {
        const R pxu = (xrow(k) < NX ? m_ucol : R(0)) * pj[k * NBD + JU];
        q_mfma(q, -ktm[k], pxu, suu);
      }
      // sig_0 = Lt0x^T Lt0x + S_u|x  (SQRT_ID: the update works on the factor; only a caller who asked for the prior joint gets it)
#pragma unroll
      for (int k = 0; k < NBD * NBD; ++k) s0[k] = R(0);
      if (!SQRT_ID || a.prior_out) {
#pragma unroll
        for (int i = 0; i < NBD; ++i)
#pragma unroll
          for (int j = i; j < NBD; ++j)
#pragma unroll
            for (int k = 0; k < NBX; ++k) {
              if (k > i && i != JU) continue;  // Lt0x[k][i] = 0 below the diagonal, except in the action columns
              q_mfma(q, lx[k * NBD + i], lx[k * NBD + j], s0[i * NBD + j]);
            }
        s0[JU * NBD + JU] += suu;
      }
      // mean: the state message, and the action prior moved by Kt delta
#pragma unroll
      for (int j = 0; j < NBD; ++j) {
        R kd = R(0);
        if (j == JU) {
          R tt = R(0);
#pragma unroll
          for (int i = 0; i < NBX; ++i) tt += ktm[i] * dr[i];
          kd = q_colsum(q, tt);
        }
        const int col = xcol(j);
        mu0[j] = col < NX ? (j < NBX ? mx[j < NBX ? j : 0] : R(0)) : (col < D ? pmu[j] + kd : R(0));
      }
      if constexpr (!OBS_ID) {
        // chol(sig_0)^T = [[Lt3, G], [0, sqrt(S_u|x)]] (one action): the variance to every lane of the trajectory
        const R v = q_bcq<CU>(q, q_colsum(q, suu));
        cell_bad = flag_stage(cell_bad, v > R(0), 1);
        const R lu = v * r_rsqrt(v);
#pragma unroll
        for (int i = 0; i < NBD; ++i)
#pragma unroll
          for (int j = i; j < NBD; ++j) {
            const R lv = i < NBX ? lx[(i < NBX ? i : 0) * NBD + j] : R(0);
            lt0[i * NBD + j] = (i == JU && j == JU) ? lv + m_uu * lu : lv;
          }
      }
      if constexpr (SQRT_ID) {
        // chol(sig_0)^T = [[Lt3, G], [0, chol(S_u|x)^T]]: the action entries of block (JU, JU) are a pivot block of their own (the
        // state rows that share the block with them -- planar quadrotor: two -- ride along as identity rows)
        R su1[1] = {suu + ((r == cc && (cc < CU || cc >= CU + NU)) ? R(1) : R(0))}, lu1[1] = {R(0)};
        cell_bad = flag_stage(cell_bad, q_elim<4, 0, 0>(q, su1, (R*)nullptr, (R*)nullptr, lu1), 1);
#pragma unroll
        for (int i = 0; i < NBD; ++i)
#pragma unroll
          for (int j = 0; j < NBD; ++j) {
            const R lv = (i < NBX && j >= i) ? lx[(i < NBX ? i : 0) * NBD + j] : R(0);
            lt0[i * NBD + j] = (i == JU && j == JU) ? lv + m_uu * lu1[0] : lv;
          }
      }
    }
    I2C_QSTAMP(0);  // factorisation pair + joint prior
    if (PREFETCH) fetch_prior(t + 1 < T ? t + 1 : t);  // this cell's rows are consumed: the next cell's, a cell ahead
    if (a.prior_out) {
      const WIO<R, S> po = wio<R, S>(a.prior_out + (unsigned long)t * (D + sym(D)) * B, (unsigned long)(D + sym(D)), rb, bo);
#pragma unroll
      for (int j = 0; j < NBD; ++j) {
        po.st_if(live && r == 0 && xcol(j) < D, xcol(j), mu0[j]);
#pragma unroll
        for (int i = 0; i <= j; ++i) po.st_if(live && xrow(i) <= xcol(j) && xcol(j) < D, D + w_symidx(xrow(i), xcol(j) < D ? xcol(j) : 0), s0[i * NBD + j]);
      }
    }

    // ---- 2. cost "observation": measurement update on z (i2c.py:390-407) ----------------
    if constexpr (SQRT_ID) {
      R zf[NBD * NBD];
      cell_bad = flag_stage(cell_bad, q_kalman_sqrt<D, QLD>(q, alpha, kc.qr, c.qr_diag != 0, zt, mu0, lt0, zf, s0, kz), 2);
#pragma unroll
      for (int k = 0; k < NBD * NBD; ++k) lt0[k] = zf[k];
    } else if constexpr (OBS_ID && GENERAL) {
      cell_bad = flag_stage(cell_bad, q_kalman_identity_general<D, QLD>(q, rule.W, alpha, kc.xi, zt, mu0, s0, kz), 2);
    } else if constexpr (OBS_ID) {
      R sf_[NBD * NBD];  // full blocks for the identity form
#pragma unroll
      for (int i = 0; i < NBD; ++i)
#pragma unroll
        for (int j = 0; j < NBD; ++j) sf_[i * NBD + j] = j >= i ? s0[i * NBD + j] : q_tr(q, s0[j * NBD + i]);
      cell_bad = flag_stage(cell_bad, q_kalman_identity<D, QLD>(q, alpha, kc.xi, kc.qr, c.qr_diag != 0, zt, mu0, sf_, kz), 2);
#pragma unroll
      for (int k = 0; k < NBD * NBD; ++k) s0[k] = sf_[k];
    } else if constexpr (LASTLIN) {
      // The last observation output is a PASS-THROUGH of joint coordinate JZL (the double cartpole observes its action as z[8]) and
      // would be a block row of its own (NZ = 4 n + 1). Its moments are exact -- mean mu0[JZL], variance sig_0[JZL][JZL], covariances
      // rows of sig_0 and a column of cov(z, xu) -- so it is conditioned on FIRST, as a scalar (any order of a Gaussian conditioning
      // gives the same posterior: the block elimination below with this pivot taken first): rank-one corrections of sig_0, of
      // the remaining observation covariance and cross-covariance, elementwise. The points evaluate NZ - 1 outputs, the
      // factorisation has one pivot phase less, and every block matrix of the update loses a block row / column.
      I2C_QSTAMP(1);
      constexpr int NZ8 = NZ - 1, NB8 = NZ8 / 4, BJ = JZL / 4, CZ = JZL % 4;
      R am[NBD * NB8], dm[NBD * NB8], yc[NB8], mz[NB8], sz[NB8 * NB8], szx[NB8 * NBD];
      q_points<M, G, D, NZ8>(q, rule.sf, mu0, lt0, ObserveHeadF<M, R, NZ8>{c.params}, am, dm, yc);
      I2C_QSTAMP(2);  // observation points
      q_moments<D, NZ8>(q, rule, am, dm, yc, mz, sz);
#pragma unroll
      for (int i = 0; i < NB8; ++i)
#pragma unroll
        for (int j = i; j < NB8; ++j) sz[i * NB8 + j] += alpha * q_ldc<QLD>(q, kc.xi, i, j, kz);
#pragma unroll
      for (int k = 0; k < NB8 * NBD; ++k) szx[k] = R(0);
      q_tn<NBD, NB8, NBD, false, false, true>(q, dm, lt0, szx);  // cov(z, xu) = wi sf [d_p]^T L^T
      const R cw = rule.wi * rule.sf;
#pragma unroll
      for (int k = 0; k < NB8 * NBD; ++k) szx[k] *= cw;
      I2C_QSTAMP(3);  // observation moments
      // h = sig_0[:, JZL] = cov(xu, z_last), b = cov(z_head, z_last) (+ the noise column), both in row and in column form
      R hrow[NBD], hcol[NBD], brow[NB8], bcol[NB8];
#pragma unroll
      for (int i = 0; i < NBD; ++i) {
        hrow[i] = q_bcq<CZ>(q, s0[i * NBD + BJ]);
        hcol[i] = q_tr(q, hrow[i]);
      }
#pragma unroll
      for (int i = 0; i < NB8; ++i) {
        brow[i] = q_bcq<CZ>(q, szx[i * NBD + BJ]) + alpha * kc.xi[(4 * i + r) * QLD + NZ8 + kz];
        bcol[i] = q_tr(q, brow[i]);
      }
      // c = var(z_last) + its noise, to every lane of the trajectory; nu = target - mean
      const R cvar = q_colsum(q, r == CZ ? hrow[BJ] : R(0)) + alpha * kc.xi[NZ8 * QLD + NZ8 + kz];
      cell_bad = flag_stage(cell_bad, cvar > R(0), 2);
      const R ic = r_rcp(cvar);
      const R nu = (q_bcq<0>(q, zt[NB8]) - q_bcq<CZ>(q, mu0[BJ])) * ic;  // (target of output NZ - 1: column 0 of its block)
      R zth[NB8];
#pragma unroll
      for (int i = 0; i < NB8; ++i) {
        const R bi = brow[i] * ic;
#pragma unroll
        for (int j = i; j < NB8; ++j) sz[i * NB8 + j] -= bi * bcol[j];
#pragma unroll
        for (int j = 0; j < NBD; ++j) szx[i * NBD + j] -= bi * hcol[j];
        zth[i] = zt[i] - bcol[i] * nu;  // the head's innovation given the last output: (zt - mz) - b nu
      }
#pragma unroll
      for (int i = 0; i < NBD; ++i) {
        const R hi_ = hrow[i] * ic;
#pragma unroll
        for (int j = i; j < NBD; ++j) s0[i * NBD + j] -= hi_ * hcol[j];
        mu0[i] += hcol[i] * nu;
      }
      cell_bad = flag_stage(cell_bad, q_kalman<D, NZ8>(q, mu0, s0, mz, sz, szx, zth), 2);
    } else {
      I2C_QSTAMP(1);
      R am[NBD * NBZ], dm[NBD * NBZ], yc[NBZ], mz[NBZ], sz[NBZ * NBZ], szx[NBZ * NBD];
      q_points<M, G, D, NZ>(q, rule.sf, mu0, lt0, ObserveF<M, R>{c.params}, am, dm, yc);
      I2C_QSTAMP(2);  // observation points
      q_moments<D, NZ, GENERAL>(q, rule, am, dm, yc, mz, sz);
#pragma unroll
      for (int i = 0; i < NBZ; ++i)
#pragma unroll
        for (int j = i; j < NBZ; ++j) sz[i * NBZ + j] += alpha * q_ldc<QLD>(q, kc.xi, i, j, kz);
      // cov(z, xu) = wi sf [d_p]^T L^T
#pragma unroll
      for (int k = 0; k < NBZ * NBD; ++k) szx[k] = R(0);
      q_tn<NBD, NBZ, NBD, false, false, true>(q, dm, lt0, szx);
      const R cw = rule.wi * rule.sf;
#pragma unroll
      for (int k = 0; k < NBZ * NBD; ++k) szx[k] *= cw;
      I2C_QSTAMP(3);  // observation moments
      cell_bad = flag_stage(cell_bad, q_kalman<D, NZ>(q, mu0, s0, mz, sz, szx, zt), 2);
    }
    I2C_QSTAMP(4);  // Kalman-style update
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      out.st_if(live && r == 0 && in_col(j, D), cc, 4 * j, mu0[j]);
#pragma unroll
      for (int i = 0; i <= j; ++i) out.st_if(live && (i < j || r <= cc) && in_col(j, D), sym_lane(i, j), D + sym_k(i, j), s0[i * NBD + j]);
    }

    // ---- 3. dynamics push-through (i2c.py:415-421) ----------------------------------------
    {
      R lt[NBD * NBD];
      if constexpr (SQRT_ID) {  // the factor of the updated joint came out of the update
#pragma unroll
        for (int k = 0; k < NBD * NBD; ++k) lt[k] = lt0[k];
      } else {
        R tmp[NBD * NBD];
#pragma unroll
        for (int k = 0; k < NBD * NBD; ++k) tmp[k] = s0[k];
        cell_bad = flag_stage(cell_bad, q_elim<D, 0, 0>(q, tmp, (R*)nullptr, (R*)nullptr, lt), 3);
      }
      I2C_QSTAMP(5);  // stores + chol(updated joint)
      R am[NBD * NBX], dm[NBD * NBX], yc[NBX], sy[NBX * NBX];
      q_points<M, G, D, NX, CENTRE>(q, rule.sf, mu0, lt, DynamicsF<M, R>{c.params}, am, dm, yc);
      I2C_QSTAMP(6);  // dynamics points
      q_moments<D, NX, GENERAL>(q, rule, am, dm, yc, mx, sy);
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int j = 0; j < NBX; ++j) sx[i * NBX + j] = j >= i ? sy[i * NBX + j] + q_ldc<QLD>(q, kc.eta, i, j, kz) : R(0);
#pragma unroll
      for (int k = 0; k < NBX * NBD; ++k) sxy[k] = R(0);
      q_tn<NBD, NBX, NBD, false, false, true>(q, dm, lt, sxy);  // sig_xy^T = wi sf [d_p]^T L^T
      const R cw = rule.wi * rule.sf;
#pragma unroll
      for (int k = 0; k < NBX * NBD; ++k) sxy[k] *= cw;
    }
    j_pending = true;  // J = sig_xy sig_x3^-1 (i2c.py:423-425): at the top of the next cell, next to the factorisation it shares
    // ---- 4. terminal cost observation on the flagged cell, after J (i2c.py:430-443) ----
    if (NZT > 0 && t == c.terminal_cell && c.has_Qf) {  // (uniform: kernel arguments)
      // J uses sig_x3_f BEFORE this update, the next cell the one after it: this cell's gain is formed here
      R tmp[NBX * NBX], w3[NBX * NBX], yj[NBX * NBD], l3t[NBX * NBX];
      identity_x(w3);
#pragma unroll
      for (int k = 0; k < NBX * NBX; ++k) tmp[k] = sx[k];
#pragma unroll
      for (int k = 0; k < NBX * NBD; ++k) yj[k] = sxy[k];
      cell_bad = flag_stage(cell_bad, q_elim<NX, NBD, NBX, true>(q, tmp, yj, w3, l3t), 4);
      store_gain(t, w3, yj);
      j_pending = false;
      R ztT[NBT];
#pragma unroll
      for (int j = 0; j < NBT; ++j) ztT[j] = q_ldv(q, kc.zgT, j, kz);
      if constexpr (TERM_ID && GENERAL) {
        cell_bad = flag_stage(cell_bad, q_kalman_identity_general<NX, QLD>(q, c.rule_x.W, alpha, kc.xiT, ztT, mx, sx, kz), 5);
      } else if constexpr (TERM_ID) {
        R sf_[NBX * NBX];
#pragma unroll
        for (int i = 0; i < NBX; ++i)
#pragma unroll
          for (int j = 0; j < NBX; ++j) sf_[i * NBX + j] = j >= i ? sx[i * NBX + j] : q_tr(q, sx[j * NBX + i]);
        cell_bad = flag_stage(cell_bad, q_kalman_identity<NX, QLD>(q, alpha, kc.xiT, kc.qf, c.qf_diag != 0, ztT, mx, sf_, kz), 5);
#pragma unroll
        for (int i = 0; i < NBX; ++i)
#pragma unroll
          for (int j = 0; j < NBX; ++j) sx[i * NBX + j] = j >= i ? sf_[i * NBX + j] : R(0);
      } else if constexpr (NZT > 0) {
        R am[NBX * NBT], dm[NBX * NBT], yc[NBT], mz[NBT], sz[NBT * NBT], szx[NBT * NBX];
        q_points<M, G, NX, NT>(q, c.rule_x.sf, mx, l3t, ObserveTermF<M, R>{c.params}, am, dm, yc);
        q_moments<NX, NT, GENERAL>(q, c.rule_x, am, dm, yc, mz, sz);
#pragma unroll
        for (int i = 0; i < NBT; ++i)
#pragma unroll
          for (int j = i; j < NBT; ++j) sz[i * NBT + j] += alpha * q_ldc<QLD>(q, kc.xiT, i, j, kz);
#pragma unroll
        for (int k = 0; k < NBT * NBX; ++k) szx[k] = R(0);
        q_tn<NBX, NBT, NBX, false, false, true>(q, dm, l3t, szx);
        const R cw = c.rule_x.wi * c.rule_x.sf;
#pragma unroll
        for (int k = 0; k < NBT * NBX; ++k) szx[k] *= cw;
        cell_bad = flag_stage(cell_bad, q_kalman<NX, NT>(q, mx, sx, mz, sz, szx, ztT), 5);
      }
    }
    fail = fold_cell_failure(fail, cell_bad, t);
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      out.st_if(live && r == 0 && in_col(j, NX), cc, O_MU3 + 4 * j, mx[j]);
#pragma unroll
      for (int i = 0; i <= j; ++i) out.st_if(live && (i < j || r <= cc) && in_col(j, NX), sym_lane(i, j), O_S3 + sym_k(i, j), sx[i * NBX + j]);
    }
    I2C_QSTAMP(7);  // dynamics moments, terminal update, stores
  }
  if (j_pending) {  // the last cell's gain
    R tmp[NBX * NBX], w3[NBX * NBX], yj[NBX * NBD], l3t[NBX * NBX];
    identity_x(w3);
#pragma unroll
    for (int k = 0; k < NBX * NBX; ++k) tmp[k] = sx[k];
#pragma unroll
    for (int k = 0; k < NBX * NBD; ++k) yj[k] = sxy[k];
    fail = fold_cell_failure(fail, flag_stage(0, q_elim<NX, NBD, NBX, true>(q, tmp, yj, w3, l3t), 4), T - 1);
    store_gain(T - 1, w3, yj);
  }
#if defined(I2C_QUAD_STAMPS) && !defined(I2C_HOST_SIM)
  if (b == 0 && q.l == 0)
    printf("quad forward, clocks per cell: pair + prior %llu | - %llu | obs points %llu | obs moments %llu | kalman %llu | stores+chol1 %llu | dyn points %llu | moments+stores %llu\n",
           stamp_acc[0] / T, stamp_acc[1] / T, stamp_acc[2] / T, stamp_acc[3] / T, stamp_acc[4] / T, stamp_acc[5] / T, stamp_acc[6] / T, stamp_acc[7] / T);
#endif
  if (live && q.p() == 0 && fail != 0 && a.status[b] == 0) a.status[b] = fail;
}

// ------------------------------------------------------------------------------------------
// Backward sweep (i2c.py:882-886 over :544-610), d = 16 with identity observations (the 12-state quadrotor)
// ------------------------------------------------------------------------------------------
// The fused walk T-1 .. 0 of FOUR trajectories per wavefront: the same cell as backward_wave_body (i2c_wave.hpp) -- RTS update of
// the joint, expected cost of the posterior observation, controller, stores -- in 4 x 4 blocks: the pivot algebra of the
// controller's factorisation is shared by four trajectories instead of being repeated by the 64 lanes of one (it was half of the
// wave form's vector instructions), and the products J dS J^T are sixteen-lane block products instead of 16 x 16 tiles of which
// a quarter is used. Cubature EM only (Linearize keeps the wave form); trajectory-major forward-message and posterior buffers.
template <class M, typename R> struct QBConst {
  static constexpr int QLD = QG<M>::QLD;
  R qr[QLD * QLD], qf[QLD * QLD], sxT[QLD * QLD];  // blkdiag(Q, R), Qf, sig_x_terminal
  R zg[QLD], zgT[QLD], mxT[QLD];
};
template <class M, typename R, class DST> I2C_FN void qbconst_fill(DST& k, const Consts<M, R>* c, const int tid, const int nthreads) {
  constexpr int NX = M::NX, NZ = M::NZ, NT = M::NZT > 0 ? M::NZT : 1, QLD = QG<M>::QLD;
  for (int e = tid; e < QLD * QLD; e += nthreads) {
    const int i = e / QLD, j = e % QLD;
    k.qr[e] = (i < NZ && j < NZ) ? c->QR[tri_any(i, j)] : R(0);
    k.qf[e] = (i < NT && j < NT) ? c->Qf[tri_any(i, j)] : R(0);
    k.sxT[e] = (i < NX && j < NX) ? c->sig_x_term[tri_any(i, j)] : R(0);
  }
  for (int e = tid; e < QLD; e += nthreads) {
    k.zg[e] = e < NZ ? c->zg[e] : R(0);
    k.zgT[e] = e < NT ? c->zg_term[e] : R(0);
    k.mxT[e] = e < NX ? c->mu_x_term[e] : R(0);
  }
}
// LDS region of one trajectory: the pivot block of an elimination, then one staged cell block (the forward rows in, the posterior rows out)
template <class M> struct QBG {
  using C = Consts<M, double>;
  static constexpr int O_ST = 40;
  static constexpr int NST = C::E_FWD > C::E_POST ? C::E_FWD : C::E_POST;
  static constexpr int RAW = O_ST + NST + (NST & 1);
  static constexpr int SIZE = RAW + ((80 - RAW % 64) % 64);  // (= 16 mod 64 elements: the four regions of a wave start 32 banks apart)
};

// sum over the sixteen lanes of a trajectory, in every lane
template <typename R> I2C_FN R q_sum16(const Quad<R>& q, const R x) { return q_rowsum(q, q_colsum(q, x)); }

// Expected quadratic cost of N(mu, s) about a target (compute_cost_gaussian, i2c.py:1046-1065): this lane's shares of
//   m = err^T W err + tr(s W),   v = 2 tr((s W)^2) + 4 err^T W s W err      (summed over the sixteen lanes by the caller).
// errc: mu - target, column form [NB]; s: upper blocks [NB][NBS] (row stride NBS), the diagonal ones full; wm: the weight, QLD x QLD in LDS.
template <int NB, int NBS, int QLD, typename R, class P>
I2C_FN void q_cost_share(const Quad<R>& q, const bool diag, const P wm, const R* errc, const R* s, R* pm, R* pv, const int kz) {
  R m = R(0), t2 = R(0), qd = R(0);
  if (diag) {
    R wdc[NB], wec[NB], wdr[NB], wer[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      wdc[j] = wm[(4 * j + q.c) * QLD + 4 * j + q.c + kz];
      wdr[j] = wm[(4 * j + q.r) * QLD + 4 * j + q.r + kz];
      wec[j] = wdc[j] * errc[j];
      wer[j] = q_tr(q, wec[j]);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      m += q.r == q.c ? wdc[j] * (errc[j] * errc[j] + s[j * NBS + j]) : R(0);
#pragma unroll
      for (int i = 0; i <= j; ++i) {
        const R sv = s[i * NBS + j], f = i < j ? R(2) : R(1);
        t2 += f * (sv * sv) * (wdr[i] * wdc[j]);
        qd += f * (wer[i] * sv) * wec[j];
      }
    }
  } else {
    R sf[NB * NB], wb[NB * NB], g[NB * NB], er[NB], wec[NB], wer[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        sf[i * NB + j] = j >= i ? s[i * NBS + j] : q_tr(q, s[j * NBS + i]);
        wb[i * NB + j] = q_ldc<QLD>(q, wm, i, j, kz);
        g[i * NB + j] = R(0);
      }
    q_tn<NB, NB, NB>(q, sf, wb, g);  // G = s W
#pragma unroll
    for (int i = 0; i < NB; ++i) er[i] = q_tr(q, errc[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) {  // W err, column and row form
      R t = R(0);
#pragma unroll
      for (int i = 0; i < NB; ++i) t += wb[i * NB + j] * er[i];
      wec[j] = q_colsum(q, t);
      wer[j] = q_tr(q, wec[j]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      m += (q.r == 0 ? errc[i] * wec[i] : R(0)) + (q.r == q.c ? g[i * NB + i] : R(0));
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        t2 += g[i * NB + j] * q_tr(q, g[j * NB + i]);
        qd += (wer[i] * sf[i * NB + j]) * wec[j];
      }
    }
  }
  *pm = m;
  *pv = R(2) * t2 + R(4) * qd;
}

// GENERAL (round 6): any CubatureQuadrature(alpha, beta, kappa): the identity observation's exact moments  mz = W mu,
// sig_z = sig + (W - W^2) mu mu^T  in the expected cost (and the terminal observation's likewise); W = 1 is the joint itself.
template <class M, typename R, typename S, bool GENERAL = false, class KC>
I2C_HD inline void backward_quad_body(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a, const int b, const bool live, const Quad<R>& qw) {
  using C = Consts<M, R>;
  using G = QG<M>;
  using BG = QBG<M>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D, NT = C::NZT1, QLD = G::QLD;
  constexpr int NBX = NX / 4, NBD = D / 4, JU = NBX;
  static_assert(G::WIDE && NX % 4 == 0 && D % 4 == 0 && NU <= 4 && NBD == NBX + 1, "quad backward sweep: d = 16, the actions in a block column of their own");
  static_assert(NZ == D && st_identity<ObsStruct<M>, NZ>(), "quad backward sweep: identity observation of the joint");
  static_assert(NZT == 0 || (NT == NX && st_identity<TermStruct<M>, NT>()), "quad backward sweep: identity terminal observation");
  static_assert(C::E_FWD % 2 == 0 && C::E_POST % 2 == 0, "quad backward sweep: cell blocks move as pairs of elements");
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  constexpr int O_K = D + sym(D), O_k = O_K + NU * NX, O_SK = O_k + NU;
  constexpr int NKF = (C::E_FWD / 2 + 15) / 16, NKP = (C::E_POST / 2 + 15) / 16;  // 16-lane passes over a cell block, two doubles per lane
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned WS = sizeof(S);
  const bool lead = live && qw.p() == 0;
  const auto st = qw.sh + BG::O_ST;  // this trajectory's staged cell block

  // Cell blocks move between HBM and the block layout THROUGH LDS: the sixteen lanes of a trajectory read its forward rows as
  // 256 consecutive bytes per instruction (a cell of a trajectory is contiguous in the trajectory-major buffers) and pick their
  // block elements out of LDS; the posterior rows go the other way. Read / written in block layout directly, every memory
  // instruction of the wave touched 12 - 16 cache lines (four elements of 32 bytes per trajectory), and the sweep was bound by
  // the line rate of the vector-memory pipeline: 90 instructions x ~12 lines per cell against 14 + 7 x ~9.
  // The forward rows of a cell are fetched a cell AHEAD into registers (nx) and written to LDS at the top of their cell.
  // (with them the cell's target: the per-cell one or a discarded dummy -- distinct addresses at the head of the forward-message
  //  buffer, which this sweep only reads -- through the buffer path and without a branch: a load behind a run-time branch costs an
  //  s_waitcnt vmcnt(0) right behind it, and memory operations return in order: a load issued after the prefetch waits for all of it)
  Pair2<S> nx[NKF];
  R nx_zt[NBD];
  const unsigned pofs = (unsigned)qw.p() * 2u * WS;  // (a lane moves two elements per pass: 32 elements per trajectory and instruction)
  const Window zw = make_window(c.z_per_cell ? (const void*)a.z : (const void*)a.fwd, (c.z_per_cell ? (unsigned long)T * NZ : 4ul) * B * sizeof(R));
  const unsigned zcell = c.z_per_cell ? (unsigned)((unsigned long)NZ * B * sizeof(R)) : 0u;
  unsigned zlane[NBD];
#pragma unroll
  for (int j = 0; j < NBD; ++j)
    zlane[j] = c.z_per_cell ? (unsigned)((((unsigned long)(4 * j + qw.c)) * B + b) * sizeof(R)) : (unsigned)((4 * b + j) * sizeof(R));
  auto fetch = [&](const int tc) {
#pragma unroll
    for (int j = 0; j < NBD; ++j) nx_zt[j] = wld<R>(zw, (unsigned)c.row(tc) * zcell, zlane[j]);
    const Window w = make_window(a.fwd + (unsigned long)tc * C::E_FWD * B, (unsigned long)C::E_FWD * B * WS);
    const unsigned bo = (unsigned)b * (unsigned)C::E_FWD * WS + pofs;
#pragma unroll
    for (int k = 0; k < NKF; ++k) {
#ifdef I2C_HOST_SIM
      if (!(16 * k + 15 < C::E_FWD / 2 || 16 * k + qw.p() < C::E_FWD / 2)) {
        nx[k] = Pair2<S>{S(0), S(0)};
        continue;
      }
#endif
      nx[k] = wld2<S>(w, opaque_uniform((unsigned)k * 32u * WS), bo);  // (the last pass reads past the block: into the next one, or zeros past the window)
    }
  };
  auto commit = [&](const Quad<R>& q) {
    q.sync();  // (the staged posterior rows of the previous cell have been read)
#pragma unroll
    for (int k = 0; k < NKF; ++k) {
      const int e2 = 16 * k + q.p();
      if (16 * k + 15 < C::E_FWD / 2 || e2 < C::E_FWD / 2) {
        st[2 * e2] = (R)nx[k].a;
        st[2 * e2 + 1] = (R)nx[k].b;
      }
    }
    q.sync();
  };
  fetch(T - 1);
#pragma unroll
  for (int k = 0; k < NKF; ++k) nx[k].a = opaque(nx[k].a), nx[k].b = opaque(nx[k].b);  // settled before the loop (see forward_wave_body)
#pragma unroll
  for (int j = 0; j < NBD; ++j) nx_zt[j] = opaque(nx_zt[j]);

  // state marginal carried along the chain: mean in column form, covariance in UPPER blocks (the diagonal ones full)
  R m3m[NBX], s3m[NBX * NBX];
  int fail = 0;
  auto note = [&](const bool ok, const int reason, const int t) { fail = (fail == 0 && !ok) ? ((reason << 16) | (t + 1)) : fail; };
  R acc_m = R(0), acc_v = R(0);
  R ztl[NBD];  // the staged cell's target as loaded
  for (int t = T; t >= 0; --t) {
    // iteration T is the end of the chain (i2c.py:546-572): no cell, the terminal state and its statistics (from the rows of cell T - 1,
    // which stay staged for iteration T - 1: nothing is written at the end of the chain)
    const int kz = (int)opaque_uniform(0u);
    const Quad<R> q = q_opaque(qw);
    const int r = q.r, cc = q.c;
    const int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
    const int tri_c = cc * (cc + 1) / 2 + r, tri_d = hi * (hi + 1) / 2 + lo;
    auto sym_lane = [&](const int i, const int j) { return i == j ? tri_d + 4 * j * hi : tri_c + 4 * j * cc; };
    auto sym_k = [&](const int i, const int j) { return i == j ? 8 * j * j + 6 * j : 8 * j * j + 2 * j + 4 * i; };
    const bool up = r <= cc;
    const bool zsel = opaque_i(c.z_per_cell) != 0;
    if (t != T - 1) {
      commit(q);
#pragma unroll
      for (int j = 0; j < NBD; ++j) ztl[j] = nx_zt[j];  // (iteration T: the target of cell T - 1, kept for iteration T - 1)
      fetch(t == T ? (T >= 2 ? T - 2 : 0) : (t >= 1 ? t - 1 : 0));
    }
    R m3f[NBX], s3f[NBX * NBX];
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      m3f[j] = st[cc + O_MU3 + 4 * j + kz];
#pragma unroll
      for (int i = 0; i < NBX; ++i) s3f[i * NBX + j] = i <= j ? st[sym_lane(i, j) + O_S3 + sym_k(i, j) + kz] : R(0);
    }
    if (t == T) {
#pragma unroll
      for (int k = 0; k < NBX; ++k) m3m[k] = m3f[k];
#pragma unroll
      for (int k = 0; k < NBX * NBX; ++k) s3m[k] = s3f[k];
      if (c.has_x_terminal) {
        // covariance control (i2c.py:548-559): the smoothed terminal state is the product of the TEMPERED filtered state
        // N(m3f, temp S3f) with the terminal prior N(mu_T, S_T) -- in Kalman form, an identity observation of the state with
        // noise S_T and target mu_T (see w_end_of_chain, i2c_wave.hpp). temp += dtemp per sweep.
        const R tmp = a.temp[b];
        q.sync();  // (every lane has read the temperature before the leading lane advances it)
        if (lead) a.temp[b] = tmp + c.dtemp;
        R stt[NBX * NBX], stf[NBX * NBX], sum[NBX * NBX], mz[NBX], zt[NBX];
#pragma unroll
        for (int i = 0; i < NBX; ++i) {
          mz[i] = m3m[i];
          zt[i] = q_ldv(q, kc.mxT, i, kz);
#pragma unroll
          for (int j = i; j < NBX; ++j) stt[i * NBX + j] = tmp * s3m[i * NBX + j];
        }
#pragma unroll
        for (int i = 0; i < NBX; ++i)
#pragma unroll
          for (int j = 0; j < NBX; ++j) {
            stf[i * NBX + j] = j >= i ? stt[i * NBX + j] : q_tr(q, stt[j * NBX + i]);
            sum[i * NBX + j] = j >= i ? stt[i * NBX + j] + q_ldc<QLD>(q, kc.sxT, i, j, kz) : R(0);
            if (j < i) stt[i * NBX + j] = R(0);
          }
        note(q_kalman<NX, NX>(q, m3m, stt, mz, sum, stf, zt), 6, T - 1);
#pragma unroll
        for (int k = 0; k < NBX * NBX; ++k) s3m[k] = stt[k];
      }
      // terminal observation statistics (i2c.py:567-570, 989-992): tr(Qf (errT errT^T + sig_z3_m)), identity observation
      R trT = R(0);
      if (NZT > 0 && c.has_Qf) {
        R mzt[NBX], szt[NBX * NBX], errT[NBX], pm, pv;
        const R Wx = GENERAL ? c.rule_x.W : R(1), cwx = Wx - Wx * Wx;
#pragma unroll
        for (int i = 0; i < NBX; ++i) {
          mzt[i] = GENERAL ? Wx * m3m[i] : m3m[i];
          const R mrx = GENERAL ? q_tr(q, m3m[i]) : R(0);
#pragma unroll
          for (int j = 0; j < NBX; ++j) szt[i * NBX + j] = (GENERAL && j >= i) ? s3m[i * NBX + j] + cwx * (mrx * m3m[j]) : s3m[i * NBX + j];
        }
#pragma unroll
        for (int j = 0; j < NBX; ++j) errT[j] = mzt[j] - q_ldv(q, kc.zgT, j, kz);
        q_cost_share<NBX, NBX, QLD>(q, c.qf_diag != 0, kc.qf, errT, szt, &pm, &pv, kz);
        trT = q_sum16(q, pm);
        if (live) {
#pragma unroll
          for (int j = 0; j < NBX; ++j) {
            if (r == 0) a.term_stats[(long)(3 + 4 * j + cc) * B + b] = mzt[j];
#pragma unroll
            for (int i = 0; i <= j; ++i)
              if (i < j || up) a.term_stats[(long)(3 + NT + sym_lane(i, j) + sym_k(i, j)) * B + b] = szt[i * NBX + j];
          }
        }
      }
      if (lead) a.term_stats[b] = trT;
      continue;
    }
    // ---- one backward cell (i2c.py:574-608) ---------------------------------------------------------------------------------
    R mu[NBD], sg[NBD * NBD], jt[NBX * NBD], zt[NBD];
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      mu[j] = st[cc + 4 * j + kz];
#pragma unroll
      for (int i = 0; i < NBD; ++i) sg[i * NBD + j] = i <= j ? st[sym_lane(i, j) + D + sym_k(i, j) + kz] : R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) jt[i * NBD + j] = st[cc * NX + r + O_J + 4 * j * NX + 4 * i + kz];  // J^T: row = state index, column = joint index
      const R zgv = q_ldv(q, kc.zg, j, kz);
      zt[j] = zsel ? ztl[j] : zgv;  // (a select on a per-lane condition: no branch)
    }
    if (a.xm && live) {  // (optional output: the smoothed state marginal that enters cell t)
      S* xo = const_cast<S*>(a.xm) + ((long)t * C::E_XM) * B + b;
#pragma unroll
      for (int j = 0; j < NBX; ++j) {
        if (r == 0) xo[(long)(4 * j + cc) * B] = (S)m3m[j];
#pragma unroll
        for (int i = 0; i <= j; ++i)
          if (i < j || up) xo[(long)(NX + sym_lane(i, j) + sym_k(i, j)) * B] = (S)s3m[i * NBX + j];
      }
    }
    // RTS update of the joint (i2c.py:580-583): mu += J (m3m - m3f), sig += J (S3m - S3f) J^T
    {
      R dsf[NBX * NBX], drr[NBX], p1[NBX * NBD];
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        drr[i] = q_tr(q, m3m[i] - m3f[i]);  // row form
#pragma unroll
        for (int j = i; j < NBX; ++j) dsf[i * NBX + j] = s3m[i * NBX + j] - s3f[i * NBX + j];
      }
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) dsf[i * NBX + j] = q_tr(q, dsf[j * NBX + i]);
#pragma unroll
      for (int j = 0; j < NBD; ++j) {
        R ts = R(0);
#pragma unroll
        for (int i = 0; i < NBX; ++i) ts += jt[i * NBD + j] * drr[i];
        mu[j] += q_colsum(q, ts);
      }
#pragma unroll
      for (int k = 0; k < NBX * NBD; ++k) p1[k] = R(0);
      q_tn<NBX, NBX, NBD>(q, dsf, jt, p1);            // dS J^T
      q_tn<NBX, NBD, NBD, false, true>(q, jt, p1, sg);  // J (dS J^T), upper blocks
    }
    // posterior observation moments = the joint itself (identity observation, i2c.py:594-596) and their expected cost
    R mzg[NBD], szg[NBD * NBD];  // (GENERAL: the observation moments; otherwise unused)
    {
      R err[NBD], pm, pv;
      if constexpr (GENERAL) {
        const R Wd = c.rule_xu.W, cwd = Wd - Wd * Wd;
#pragma unroll
        for (int i = 0; i < NBD; ++i) {
          mzg[i] = Wd * mu[i];
          const R mrd = q_tr(q, mu[i]);
#pragma unroll
          for (int j = 0; j < NBD; ++j) szg[i * NBD + j] = j >= i ? sg[i * NBD + j] + cwd * (mrd * mu[j]) : sg[i * NBD + j];
        }
#pragma unroll
        for (int j = 0; j < NBD; ++j) err[j] = mzg[j] - zt[j];
        q_cost_share<NBD, NBD, QLD>(q, c.qr_diag != 0, kc.qr, err, szg, &pm, &pv, kz);
      } else {
#pragma unroll
        for (int j = 0; j < NBD; ++j) err[j] = mu[j] - zt[j];
        q_cost_share<NBD, NBD, QLD>(q, c.qr_diag != 0, kc.qr, err, sg, &pm, &pv, kz);
      }
      acc_m += pm;
      acc_v += pv;
      if (a.cell_stats) {
        const R cm = q_sum16(q, pm), cv = q_sum16(q, pv);
        if (lead) {
          a.cell_stats[((long)t * 2 + 0) * B + b] = cm;
          a.cell_stats[((long)t * 2 + 1) * B + b] = cv;
        }
      }
    }
    q.sync();  // (every lane has picked its forward rows out of the staged block: the posterior rows take its place)
    // controller (i2c.py:600-608): with [W | Y] = chol(sig_xx)^-1 [I | sig_xu]:  K^T = W^T Y, sigK = sig_uu - Y^T Y
    {
      R sxx[NBX * NBX], y[NBX], w3[NBX * NBX], lt[NBX * NBX], kt[NBX], mr[NBX];
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        y[i] = cc < NU ? sg[i * NBD + JU] : R(0);
        kt[i] = R(0);
        mr[i] = q_tr(q, mu[i]);  // the state mean, row form
#pragma unroll
        for (int j = 0; j < NBX; ++j) {
          sxx[i * NBX + j] = j >= i ? sg[i * NBD + j] : R(0);
          w3[i * NBX + j] = (i == j && r == cc) ? R(1) : R(0);
        }
      }
      note(q_elim<NX, 1, NBX, true>(q, sxx, y, w3, lt), 7, t);
      R sk = (r < NU && cc < NU) ? sg[JU * NBD + JU] : R(0), tk = R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
#pragma unroll
        for (int k = i; k < NBX; ++k) q_mfma(q, w3[k * NBX + i], y[k], kt[i]);  // (W is lower triangular: blocks k >= i)
        q_mfma(q, -y[i], y[i], sk);
        tk += kt[i] * mr[i];
        if (cc < NU) st[cc * NX + r + O_K + 4 * i] = kt[i];  // K[u][x], u = the column, x = 4 i + row
      }
      const R kx = q_colsum(q, tk);  // K mu_x, column form over the actions
      if (r == 0 && cc < NU) st[cc + O_k] = mu[JU] - kx;
      if (up && cc < NU) st[tri_c + O_SK] = sk;
    }
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      if (r == 0) st[cc + 4 * j] = mu[j];
#pragma unroll
      for (int i = 0; i <= j; ++i)
        if (i < j || up) st[sym_lane(i, j) + D + sym_k(i, j)] = sg[i * NBD + j];
    }
    q.sync();
    {  // the posterior rows of the cell, 256 consecutive bytes per trajectory and instruction
      const Window w = make_window(a.post + (unsigned long)c.row(t) * C::E_POST * B, (unsigned long)C::E_POST * B * WS);
      const unsigned bo = (unsigned)b * (unsigned)C::E_POST * WS + pofs;
#pragma unroll
      for (int k = 0; k < NKP; ++k) {
        const int e2 = 16 * k + q.p();
        const bool in = 16 * k + 15 < C::E_POST / 2 || e2 < C::E_POST / 2;
        const int e2c = in ? e2 : 0;
        const Pair2<S> v{(S)st[2 * e2c], (S)st[2 * e2c + 1]};
#ifdef I2C_HOST_SIM
        if (live && in) wst2<S>(w, (unsigned)k * 32u * WS, bo, v);
#else
        wst2<S>(w, opaque_uniform((unsigned)k * 32u * WS), (live && in) ? bo : 0x80000000u, v);  // (out of the window: dropped by the buffer unit)
#endif
      }
    }
    if (a.zpost && live) {
      S* zo = a.zpost + ((long)t * C::E_ZPOST) * B + b;
#pragma unroll
      for (int j = 0; j < NBD; ++j) {
        if (r == 0) zo[(long)(4 * j + cc) * B] = (S)(GENERAL ? mzg[j] : mu[j]);
#pragma unroll
        for (int i = 0; i <= j; ++i)
          if (i < j || up) zo[(long)(NZ + sym_lane(i, j) + sym_k(i, j)) * B] = (S)(GENERAL ? szg[i * NBD + j] : sg[i * NBD + j]);
      }
    }
#ifndef I2C_QB16_SYM
#define I2C_QB16_SYM 1
#endif
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      m3m[j] = mu[j];
#pragma unroll
      for (int i = 0; i < NBX; ++i) s3m[i * NBX + j] = i <= j ? sg[i * NBD + j] : R(0);
      // the CARRIED diagonal blocks exactly symmetric again (round 6, see backward_quad8_body: both triangles of J dS J^T are computed,
      // each with its own rounding, and the antisymmetric part of the carried covariance obeys A <- Jx A Jx^T along the chain). Here,
      // at the end of the cell, and not where the blocks are formed: there the four transposes pushed the kernel over its register
      // budget (2 -> 38 spilled registers, 1.67 -> 1.97 ms at B = 32768)
      if constexpr (I2C_QB16_SYM != 0) {
        const R st_ = q_tr(q, s3m[j * NBX + j]);
        s3m[j * NBX + j] = up ? s3m[j * NBX + j] : st_;
      }
    }
  }
  const R sm = q_sum16(qw, acc_m), sv = q_sum16(qw, acc_v);
  if (lead) {
    a.term_stats[B + b] = sm;
    a.term_stats[2 * B + b] = sv;
    if (fail != 0 && a.status[b] == 0) a.status[b] = fail;
  }
}

// ------------------------------------------------------------------------------------------
// Backward sweep (i2c.py:882-886 over :544-610), d <= 8: every model of the forward kernel's d <= 8 geometry (round 6)
// ------------------------------------------------------------------------------------------
// The fused walk T-1 .. 0 of FOUR trajectories per wavefront on the common [T][E][B] buffers: ONE pass over the forward messages
// (E_FWD elements read, E_POST written per cell: the algorithmic bytes) where the chunked lane schedule makes three (compose +
// stitch + walk: 1.5 - 1.7x the bytes, profiles/r5_*_pmc_traffic.json). What this form has that the d = 16 one lacks:
//   * the posterior observation through SIGMA POINTS (i2c.py:594-596 for a general sys.observe: q_points / q_moments on the smoothed
//     joint, the forward kernel's building blocks), and the terminal observation the same way at the end of the chain (:567-570);
//   * actions that share a block with states: the controller is read off ONE factorisation of the smoothed joint -- lt = chol(sig)^T
//     serves the sigma points AND the controller:  K L_xx = L_ux  as a blocked back substitution with the upper blocks of lt and
//     the inverses of its 4 x 4 diagonal blocks (a by-product of the pivot algebra, q_elim's awk),  sigK = L_uu L_uu^T  (the lane
//     kernels read the same controller off the same factor: cell_posterior, i2c_cell.hpp).
// The forward rows of a cell are fetched a cell AHEAD (registers); cubature rule with lam = 0, or any (alpha, beta, kappa) for the
// models that have the GENERAL moments (quad_general_exists); fp64 or fp32-stored messages.
template <class M> constexpr bool quad_backward8_exists() {
  constexpr int D = M::NX + M::NU;
  return !QG<M>::WIDE && D <= 8 && (M::NX % 4) + M::NU <= 4 && M::NZ <= 12 && M::NZT <= 12;
}
// Cell blocks of the [E][B] buffers for the lanes of a quad wave, with the address of an element split three ways: the block part
// k (a compile-time constant after unrolling) x the row stride goes into the instruction's SCALAR offset; the lane part -- lane_e rows
// + the trajectory's offset, or, for a lane that holds padding, an offset parked out of the buffer window -- is ONE loop-invariant
// register per lane pattern, shared by every access of that pattern. A parked load returns 0 (the buffer unit's range check) and a
// parked store is dropped: zero padding and store predicates cost no instruction inside the time loop. (The forward kernel computes
// (lane_e + k) * rb + bo and a select per access: 370 of the 535 vector instructions of this sweep's first version were such.)
template <typename R, typename S> struct QIO8 {
  unsigned rb, bo;
  static constexpr unsigned PARK = 0x80000000u;
  I2C_MEM unsigned lane(const bool on, const int lane_e) const { return on ? (unsigned)lane_e * rb + bo : PARK; }
  I2C_MEM R ld(const Window& w, const unsigned off, const int k) const {
#ifdef I2C_HOST_SIM
    if (off == PARK) return R(0);
#endif
    return (R)wld<S>(w, (unsigned)k * rb, off);
  }
  I2C_MEM void st(const Window& w, const unsigned off, const int k, const R v) const {
#ifdef I2C_HOST_SIM
    if (off == PARK) return;
#endif
    wst(w, (unsigned)k * rb, off, (S)v);
  }
};
// LEANQ: no optional outputs (smoothed state per cell, observation moments, per-cell cost statistics): what i2c_learn / i2c_mpc_step
// run. Compiled out, not branched over: a store behind a run-time branch costs the lone wave an s_waitcnt vmcnt(0) per cell.
// CHUNK: the walker of the CHUNKED schedule (compose + stitch + walk + reduce, i2c_cell.hpp): cells t_hi - 1 .. t_lo of ONE chunk,
// entered with the smoothed state the stitch pass left in bnd[ch] (the end of the chain and the terminal statistics are the stitch
// pass's), cost sums into part[ch] for the reduction. Small batches are bound by the DEPTH of the chain: the lane walker needs
// ~ 24 k clocks for a d = 8 cell of 64 trajectories, this one 4.2 k for a cell of four -- with chunks supplying the parallelism
// over t, the walk of a batch that leaves most SIMDs idle is as long as its chunk, not as its lane cell.
// MODE = QB8_STITCH: the STITCH pass of the chunked schedule in the same blocks -- end of the chain and terminal statistics (the code
// of the fused walk), then chunk by chunk  bnd[ch] <- (m, S);  (m, S) <- (a + G m, C + G S G^T)  with the composites of the compose
// pass (compose_quad8_body below, or the lane kernels': one format). A step is ~20 matrix + ~60 vector instructions where the lane
// form needs 767 vector instructions (planar quadrotor), and the pass is a chain of NC dependent steps whatever the batch.
template <typename R> struct QChunk {
  R* bnd;         // [NC][NX + sym(NX)][B]  smoothed state entering each chunk (written by the stitch pass, read by the walkers)
  R* part;        // [NC][2][B]             per-chunk cost sums
  int ch, t_lo, t_hi;
  const R* comp;  // [NC][NX + NX*NX + sym(NX)][B]  composite maps (stitch pass)
  int n_chunks;
};
enum { QB8_FUSED = 0, QB8_CHUNK_WALK = 1, QB8_STITCH = 2 };
template <class M, typename R, typename S, bool GENERAL = false, bool LEANQ = false, int MODE = QB8_FUSED, class KC>
I2C_HD inline void backward_quad8_body(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a_in, const int b, const bool live, const Quad<R>& q,
                                       const QChunk<R> qc = QChunk<R>{nullptr, nullptr, 0, 0, 0, nullptr, 0}) {
  constexpr bool CHUNK = MODE == QB8_CHUNK_WALK;
  CellArgs<R, S> a = a_in;
  if (LEANQ) a.xm = nullptr, a.zpost = nullptr, a.cell_stats = nullptr;
  using C = Consts<M, R>;
  using G = QG<M>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D, NT = C::NZT1, QLD = G::QLD;
  constexpr int NBX = (NX + 3) / 4, NBD = (D + 3) / 4, NBZ = (NZ + 3) / 4, NBT = (NT + 3) / 4;
  static_assert(quad_backward8_exists<M>(), "quad backward sweep (d <= 8): the actions live in one block");
  static_assert(!GENERAL || quad_general_exists<M>(), "general cubature weights: the d <= 8 geometry");
  constexpr bool OBS_ID = st_identity<ObsStruct<M>, NZ>() && NZ == D;
  constexpr bool TERM_ID = NZT > 0 && st_identity<TermStruct<M>, NT>() && NT == NX;
  constexpr int JU = NX / 4, CU = NX % 4;  // the block (row and column) and the in-block offset where the action entries start
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  constexpr int O_K = D + sym(D), O_k = O_K + NU * NX, O_SK = O_k + NU;
  const int r = q.r, cc = q.c;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned WS = sizeof(S), bo = (unsigned)b * WS, rb = (unsigned)(B * WS);
  const bool lead = live && q.p() == 0;
  const Rule<R>& rule = c.rule_xu;
  auto xrow = [&](const int i) { return 4 * i + r; };
  auto xcol = [&](const int j) { return 4 * j + cc; };
  // packed-symmetric element (row, col) of block (i, j), i <= j, split into its per-lane and its per-block part (forward_quad_body)
  const int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
  const int tri_c = cc * (cc + 1) / 2 + r, tri_d = hi * (hi + 1) / 2 + lo;
  auto sym_lane = [&](const int i, const int j) { return i == j ? tri_d + 4 * j * hi : tri_c + 4 * j * cc; };
  auto sym_k = [&](const int i, const int j) { return i == j ? 8 * j * j + 6 * j : 8 * j * j + 2 * j + 4 * i; };
  auto in_row = [&](const int i, const int N) { return 4 * i + 3 < N || 4 * i + r < N; };
  auto in_col = [&](const int j, const int N) { return 4 * j + 3 < N || 4 * j + cc < N; };
  auto in_n = [&](const int i, const int j, const int N) { return in_row(i, N) && in_col(j, N); };
  const bool up = r <= cc;
  const bool u_row = r >= CU && r < CU + NU, u_col = cc >= CU && cc < CU + NU;  // action rows / columns of block JU
  // the lane parts of every access of a cell (loop-invariant; a lane that holds padding is parked: loads give 0, stores are dropped)
  const QIO8<R, S> io{rb, bo};
  unsigned l_mu[NBD], l_sg[NBD * NBD], l_m3[NBX], l_s3[NBX * NBX], l_jt[NBX * NBD];                        // loads (forward rows)
  unsigned s_mu[NBD], s_sg[NBD * NBD], s_kt[NBX], s_k, s_sk;                                              // stores (posterior rows)
#pragma unroll
  for (int j = 0; j < NBD; ++j) {
    l_mu[j] = io.lane(in_col(j, D), cc);
    s_mu[j] = io.lane(live && r == 0 && in_col(j, D), cc);
#pragma unroll
    for (int i = 0; i < NBD; ++i) {
      l_sg[i * NBD + j] = io.lane(i <= j && in_n(i, j, D), sym_lane(i, j));  // upper blocks (the diagonal ones as full symmetric blocks)
      s_sg[i * NBD + j] = io.lane(live && i <= j && (i < j || up) && in_col(j, D), sym_lane(i, j));
    }
#pragma unroll
    for (int i = 0; i < NBX; ++i) l_jt[i * NBD + j] = io.lane(in_row(i, NX) && in_col(j, D), cc * NX + r);  // J^T: row = state, column = joint index
  }
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    l_m3[j] = io.lane(in_col(j, NX), cc);
#pragma unroll
    for (int i = 0; i < NBX; ++i) l_s3[i * NBX + j] = io.lane(i <= j && in_n(i, j, NX), sym_lane(i, j));
    s_kt[j] = io.lane(live && in_row(j, NX) && (j != JU || r < CU) && u_col, (cc - CU) * NX + r);  // K[u][x], u = the column, x = 4 j + row
  }
  s_k = io.lane(live && r == 0 && u_col, cc - CU);
  s_sk = io.lane(live && u_row && u_col && r >= cc, (r - CU) * (r - CU + 1) / 2 + (cc - CU));

  // ---- the forward rows of a cell, a cell ahead ---------------------------------------------------------------------------
  R nx_mu[NBD], nx_sg[NBD * NBD], nx_m3[NBX], nx_s3[NBX * NBX], nx_jt[NBX * NBD], nx_zt[NBZ];
  // per-cell targets, or a discarded dummy (the head of the forward-message buffer, which this sweep only reads: distinct addresses,
  // so that the loads stay independent instructions -- see forward_quad_body): branch-free buffer loads
  const Window zw = make_window(c.z_per_cell ? (const void*)a.z : (const void*)a.fwd,
                                c.z_per_cell ? (unsigned long)T * NZ * B * sizeof(R) : (unsigned long)C::E_FWD * B * WS);
  const unsigned zcell = c.z_per_cell ? (unsigned)((unsigned long)NZ * B * sizeof(R)) : 0u;
  unsigned zlane[NBZ];
#pragma unroll
  for (int j = 0; j < NBZ; ++j) {
    const int col = xcol(j);
    zlane[j] = c.z_per_cell ? (unsigned)((((unsigned long)(col < NZ ? col : 0)) * B + b) * sizeof(R)) : (unsigned)(((unsigned long)j * B + b) * WS);
  }
  auto fetch = [&](const int tc) {
    const Window f = make_window(a.fwd + (unsigned long)tc * C::E_FWD * B, (unsigned long)C::E_FWD * rb);
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      nx_mu[j] = io.ld(f, l_mu[j], 4 * j);
#pragma unroll
      for (int i = 0; i <= j; ++i) nx_sg[i * NBD + j] = io.ld(f, l_sg[i * NBD + j], D + sym_k(i, j));
    }
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      nx_m3[j] = io.ld(f, l_m3[j], O_MU3 + 4 * j);
#pragma unroll
      for (int i = 0; i <= j; ++i) nx_s3[i * NBX + j] = io.ld(f, l_s3[i * NBX + j], O_S3 + sym_k(i, j));
    }
#pragma unroll
    for (int i = 0; i < NBX; ++i)
#pragma unroll
      for (int j = 0; j < NBD; ++j) nx_jt[i * NBD + j] = io.ld(f, l_jt[i * NBD + j], O_J + 4 * j * NX + 4 * i);  // stored as J[joint][state]
#pragma unroll
    for (int j = 0; j < NBZ; ++j) nx_zt[j] = wld<R>(zw, (unsigned)c.row(tc) * zcell, zlane[j]);
  };
  const int t_hi = CHUNK ? qc.t_hi : T, t_lo = CHUNK ? qc.t_lo : 0;
  fetch(t_hi - 1);
  // settled before the loop (see forward_wave_body: loads pending on the loop-entry path cost a vmcnt(0) in every cell)
#pragma unroll
  for (int k = 0; k < NBD; ++k) nx_mu[k] = opaque(nx_mu[k]);
#pragma unroll
  for (int i = 0; i < NBD; ++i)
#pragma unroll
    for (int j = i; j < NBD; ++j) nx_sg[i * NBD + j] = opaque(nx_sg[i * NBD + j]);
#pragma unroll
  for (int k = 0; k < NBX; ++k) nx_m3[k] = opaque(nx_m3[k]);
#pragma unroll
  for (int i = 0; i < NBX; ++i)
#pragma unroll
    for (int j = i; j < NBX; ++j) nx_s3[i * NBX + j] = opaque(nx_s3[i * NBX + j]);
#pragma unroll
  for (int k = 0; k < NBX * NBD; ++k) nx_jt[k] = opaque(nx_jt[k]);
#pragma unroll
  for (int k = 0; k < NBZ; ++k) nx_zt[k] = opaque(nx_zt[k]);

  int fail = 0;
  auto note = [&](const bool ok, const int reason, const int t) { fail = (fail == 0 && !ok) ? ((reason << 16) | (t + 1)) : fail; };
  // state marginal carried along the chain: mean in column form, covariance in UPPER blocks (the diagonal ones full), zero-padded
  R m3m[NBX], s3m[NBX * NBX];
  if constexpr (CHUNK) {
    // ---- the smoothed state that enters the chunk's last cell: what the stitch pass composed (R-typed, packed lower) -------------
    const R* bi = qc.bnd + ((long)qc.ch * C::E_XM) * (long)B + b;
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      const bool cj = in_col(j, NX);
      const R v = bi[(long)(cj ? 4 * j + cc : 0) * (long)B];
      m3m[j] = cj ? v : R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        const bool on = i <= j && in_n(i, j, NX);
        const R w = bi[(long)(NX + (on ? sym_lane(i, j) + sym_k(i, j) : 0)) * (long)B];
        s3m[i * NBX + j] = on ? w : R(0);
      }
    }
  } else {
  // ---- end of the chain (i2c.py:546-572): the smoothed terminal state and its statistics, from the rows of cell T - 1 ----------
    const int kz = (int)opaque_uniform(0u);
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      m3m[j] = nx_m3[j];  // (padding lanes loaded zeros)
#pragma unroll
      for (int i = 0; i < NBX; ++i) s3m[i * NBX + j] = i <= j ? nx_s3[i * NBX + j] : R(0);
    }
    if (c.has_x_terminal) {
      // covariance control (i2c.py:548-559): the smoothed terminal state is the product of the TEMPERED filtered state
      // N(m3f, temp S3f) with the terminal prior N(mu_T, S_T) -- in Kalman form, an identity observation of the state with noise
      // S_T and target mu_T (see backward_quad_body / w_end_of_chain). temp += dtemp per sweep.
      const R tmp = a.temp[b];
      q.sync();  // (every lane has read the temperature before the leading lane advances it)
      if (lead) a.temp[b] = tmp + c.dtemp;
      R stt[NBX * NBX], stf[NBX * NBX], sum[NBX * NBX], mz[NBX], zt[NBX];
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        mz[i] = m3m[i];
        zt[i] = q_ldv(q, kc.mxT, i, kz);
#pragma unroll
        for (int j = i; j < NBX; ++j) stt[i * NBX + j] = tmp * s3m[i * NBX + j];
      }
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int j = 0; j < NBX; ++j) {
          stf[i * NBX + j] = j >= i ? stt[i * NBX + j] : q_tr(q, stt[j * NBX + i]);
          sum[i * NBX + j] = j >= i ? stt[i * NBX + j] + q_ldc<QLD>(q, kc.sxT, i, j, kz) : R(0);
          if (j < i) stt[i * NBX + j] = R(0);
        }
      note(q_kalman<NX, NX>(q, m3m, stt, mz, sum, stf, zt), 6, T - 1);
#pragma unroll
      for (int k = 0; k < NBX * NBX; ++k) s3m[k] = stt[k];
    }
    // terminal observation statistics (i2c.py:567-570, 989-992): tr(Qf (errT errT^T + sig_z3_m))
    R trT = R(0);
    if (NZT > 0 && c.has_Qf) {
      if constexpr (TERM_ID) {
        // identity terminal observation: the rule's exact moments  mzT = W m,  sig_zT = S + (W - W^2) m m^T  (W = 1: the state itself)
        R mzt[NBX], szt[NBX * NBX], errT[NBX], pm, pv;
        const R Wx = GENERAL ? c.rule_x.W : R(1), cww = Wx - Wx * Wx;
#pragma unroll
        for (int i = 0; i < NBX; ++i) {
          mzt[i] = GENERAL ? Wx * m3m[i] : m3m[i];
          const R mr = GENERAL ? q_tr(q, m3m[i]) : R(0);
#pragma unroll
          for (int j = 0; j < NBX; ++j) szt[i * NBX + j] = (GENERAL && j >= i) ? s3m[i * NBX + j] + cww * (mr * m3m[j]) : s3m[i * NBX + j];
        }
#pragma unroll
        for (int j = 0; j < NBX; ++j) errT[j] = mzt[j] - q_ldv(q, kc.zgT, j, kz);
        q_cost_share<NBX, NBX, QLD>(q, c.qf_diag != 0, kc.qf, errT, szt, &pm, &pv, kz);
        trT = q_sum16(q, pm);
        if (live) {
#pragma unroll
          for (int j = 0; j < NBX; ++j) {
            if (r == 0 && in_col(j, NX)) a.term_stats[(long)(3 + 4 * j + cc) * B + b] = mzt[j];
#pragma unroll
            for (int i = 0; i <= j; ++i)
              if ((i < j || up) && in_col(j, NX)) a.term_stats[(long)(3 + NT + sym_lane(i, j) + sym_k(i, j)) * B + b] = szt[i * NBX + j];
          }
        }
      } else if constexpr (NZT > 0) {
        // a general terminal observation: the sigma points of the smoothed terminal state through sys.observe_terminal_x
        R tmp[NBX * NBX], l3t[NBX * NBX], am[NBX * NBT], dm[NBX * NBT], yc[NBT], mzt[NBT], szt[NBT * NBT], errT[NBT], pm, pv;
#pragma unroll
        for (int k = 0; k < NBX * NBX; ++k) tmp[k] = s3m[k];
        note(q_elim<NX, 0, 0>(q, tmp, (R*)nullptr, (R*)nullptr, l3t), 6, T - 1);
        q_points<M, G, NX, NT>(q, c.rule_x.sf, m3m, l3t, ObserveTermF<M, R>{c.params}, am, dm, yc);
        q_moments<NX, NT, GENERAL>(q, c.rule_x, am, dm, yc, mzt, szt);
#pragma unroll
        for (int j = 0; j < NBT; ++j) errT[j] = mzt[j] - q_ldv(q, kc.zgT, j, kz);
        q_cost_share<NBT, NBT, QLD>(q, c.qf_diag != 0, kc.qf, errT, szt, &pm, &pv, kz);
        trT = q_sum16(q, pm);
        if (live) {
#pragma unroll
          for (int j = 0; j < NBT; ++j) {
            if (r == 0 && in_col(j, NT)) a.term_stats[(long)(3 + 4 * j + cc) * B + b] = mzt[j];
#pragma unroll
            for (int i = 0; i <= j; ++i)
              if ((i < j || up) && in_col(j, NT)) a.term_stats[(long)(3 + NT + sym_lane(i, j) + sym_k(i, j)) * B + b] = szt[i * NBT + j];
          }
        }
      }
    }
    if (lead) a.term_stats[b] = trT;
  }

  if constexpr (MODE == QB8_STITCH) {
    // ---- the stitch pass: the smoothed state entering every chunk, from the last chunk down ------------------------------------
    constexpr int EC = NX + NX * NX + sym(NX);
    const QIO8<R, R> ior{(unsigned)(B * sizeof(R)), (unsigned)b * (unsigned)sizeof(R)};
    unsigned c_a[NBX], c_g[NBX * NBX], c_c[NBX * NBX], b_m[NBX], b_s[NBX * NBX];
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      c_a[j] = ior.lane(in_col(j, NX), cc);
      b_m[j] = ior.lane(live && r == 0 && in_col(j, NX), cc);
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        c_g[i * NBX + j] = ior.lane(in_n(i, j, NX), r * NX + cc);  // G[4 i + r][4 j + c], row-major
        c_c[i * NBX + j] = ior.lane(i <= j && in_n(i, j, NX), sym_lane(i, j));
        b_s[i * NBX + j] = ior.lane(live && i <= j && (i < j || up) && in_col(j, NX), sym_lane(i, j));
      }
    }
    R ca[NBX], cg[NBX * NBX], ccv[NBX * NBX];  // the next composite, a step ahead
    auto fetchc = [&](const int ch) {
      const Window w = make_window(qc.comp + (unsigned long)(ch > 0 ? ch : 0) * EC * B, (unsigned long)EC * ior.rb);
#pragma unroll
      for (int j = 0; j < NBX; ++j) {
        ca[j] = ior.ld(w, c_a[j], 4 * j);
#pragma unroll
        for (int i = 0; i < NBX; ++i) {
          cg[i * NBX + j] = ior.ld(w, c_g[i * NBX + j], NX + 4 * i * NX + 4 * j);
          if (i <= j) ccv[i * NBX + j] = ior.ld(w, c_c[i * NBX + j], NX + NX * NX + sym_k(i, j));
        }
      }
    };
    fetchc(qc.n_chunks - 1);
    for (int ch = qc.n_chunks - 1; ch >= 0; --ch) {
      R av[NBX], gt[NBX * NBX], cn[NBX * NBX];
#pragma unroll
      for (int j = 0; j < NBX; ++j) {
        av[j] = ca[j];
#pragma unroll
        for (int i = 0; i < NBX; ++i) {
          gt[i * NBX + j] = q_tr(q, cg[j * NBX + i]);  // the blocks of G^T
          cn[i * NBX + j] = i <= j ? ccv[i * NBX + j] : R(0);
        }
      }
      fetchc(ch - 1);
      const Window wb = make_window(qc.bnd + (unsigned long)ch * C::E_XM * B, (unsigned long)C::E_XM * ior.rb);
      R mr[NBX], sf[NBX * NBX], p1[NBX * NBX];
#pragma unroll
      for (int j = 0; j < NBX; ++j) {
        ior.st(wb, b_m[j], 4 * j, m3m[j]);
        mr[j] = q_tr(q, m3m[j]);  // row form
#pragma unroll
        for (int i = 0; i <= j; ++i) ior.st(wb, b_s[i * NBX + j], NX + sym_k(i, j), s3m[i * NBX + j]);
      }
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int j = 0; j < NBX; ++j) {
          sf[i * NBX + j] = j >= i ? s3m[i * NBX + j] : q_tr(q, s3m[j * NBX + i]);
          p1[i * NBX + j] = R(0);
        }
#pragma unroll
      for (int j = 0; j < NBX; ++j) {  // m <- a + G m
        R ts = R(0);
#pragma unroll
        for (int i = 0; i < NBX; ++i) ts += gt[i * NBX + j] * mr[i];
        m3m[j] = av[j] + q_colsum(q, ts);
      }
      q_tn<NBX, NBX, NBX>(q, sf, gt, p1);               // S G^T
      q_tn<NBX, NBX, NBX, false, true>(q, gt, p1, cn);  // C + G (S G^T), upper blocks
#pragma unroll
      for (int i = 0; i < NBX; ++i) {  // (the diagonal blocks exactly symmetric again: see the RTS update of the walk)
        const R st = q_tr(q, cn[i * NBX + i]);
        cn[i * NBX + i] = up ? cn[i * NBX + i] : st;
      }
#pragma unroll
      for (int k = 0; k < NBX * NBX; ++k) s3m[k] = cn[k];
    }
    if (lead && fail != 0 && a.status[b] == 0) a.status[b] = fail;
    return;
  }
  R acc_m = R(0), acc_v = R(0);
  for (int t = t_hi - 1; t >= t_lo; --t) {
    const int kz = (int)opaque_uniform(0u);  // (see q_ldc)
    const Window out = make_window(a.post + (unsigned long)c.row(t) * C::E_POST * B, (unsigned long)C::E_POST * rb);
    // this cell's rows (zero-padded), then the fetch of the next one
    R mu[NBD], sg[NBD * NBD], jt[NBX * NBD], m3f[NBX], s3f[NBX * NBX], zt[NBZ];
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      mu[j] = nx_mu[j];  // (padding lanes loaded zeros)
#pragma unroll
      for (int i = 0; i < NBD; ++i) sg[i * NBD + j] = i <= j ? nx_sg[i * NBD + j] : R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) jt[i * NBD + j] = nx_jt[i * NBD + j];
    }
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      m3f[j] = nx_m3[j];
#pragma unroll
      for (int i = 0; i < NBX; ++i) s3f[i * NBX + j] = i <= j ? nx_s3[i * NBX + j] : R(0);
    }
#pragma unroll
    for (int j = 0; j < NBZ; ++j) zt[j] = xcol(j) < NZ ? (c.z_per_cell ? nx_zt[j] : q_ldv(q, kc.zg, j, kz)) : R(0);
    fetch(t > t_lo ? t - 1 : t_lo);
    if (a.xm && live) {  // (optional output: the smoothed state marginal that enters cell t)
      S* xo = const_cast<S*>(a.xm) + ((long)t * C::E_XM) * B + b;
#pragma unroll
      for (int j = 0; j < NBX; ++j) {
        if (r == 0 && in_col(j, NX)) xo[(long)(4 * j + cc) * B] = (S)m3m[j];
#pragma unroll
        for (int i = 0; i <= j; ++i)
          if ((i < j || up) && in_col(j, NX)) xo[(long)(NX + sym_lane(i, j) + sym_k(i, j)) * B] = (S)s3m[i * NBX + j];
      }
    }
    // ---- RTS update of the joint (i2c.py:580-583): mu += J (m3m - m3f), sig += J (S3m - S3f) J^T --------------------------------
    {
      R dsf[NBX * NBX], drr[NBX], p1[NBX * NBD];
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        drr[i] = q_tr(q, m3m[i] - m3f[i]);  // row form
#pragma unroll
        for (int j = i; j < NBX; ++j) dsf[i * NBX + j] = s3m[i * NBX + j] - s3f[i * NBX + j];
      }
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) dsf[i * NBX + j] = q_tr(q, dsf[j * NBX + i]);
#pragma unroll
      for (int j = 0; j < NBD; ++j) {
        R ts = R(0);
#pragma unroll
        for (int i = 0; i < NBX; ++i) ts += jt[i * NBD + j] * drr[i];
        mu[j] += q_colsum(q, ts);
      }
#pragma unroll
      for (int k = 0; k < NBX * NBD; ++k) p1[k] = R(0);
      q_tn<NBX, NBX, NBD>(q, dsf, jt, p1);              // dS J^T
      q_tn<NBX, NBD, NBD, false, true>(q, jt, p1, sg);  // J (dS J^T), upper blocks
      // the diagonal blocks, exactly symmetric again: both triangles of J dS J^T are computed, each with its own rounding, and the
      // antisymmetric part A of the carried state covariance obeys A <- Jx A Jx^T along the chain -- it GROWS where the smoother gain
      // has a direction > 1 (double cartpole, first iteration: 1e-10 of the joint after 60 cells, 6e-5 of K)
#pragma unroll
      for (int i = 0; i < NBD; ++i) {
        const R st = q_tr(q, sg[i * NBD + i]);
        sg[i * NBD + i] = up ? sg[i * NBD + i] : st;
      }
    }
    // ---- ONE factorisation of the smoothed joint: lt = chol(sig)^T (sigma points, controller) and the inverses of its diagonal blocks
    R lt[NBD * NBD], aw[NBD];
    {
      R tmp[NBD * NBD];
#pragma unroll
      for (int k = 0; k < NBD * NBD; ++k) tmp[k] = sg[k];
      note(q_elim<D, 0, 0>(q, tmp, (R*)nullptr, (R*)nullptr, lt, aw), 7, t);
    }
    // ---- posterior observation moments (i2c.py:594-596) and their expected cost (calc_cost, i2c.py:1046-1065) --------------------
    {
      R pm, pv;
      if constexpr (OBS_ID) {
        // identity observation: the rule's exact moments  mz = W mu,  sig_z = sig + (W - W^2) mu mu^T  (W = 1: the joint itself)
        R mz[NBD], sz[NBD * NBD], err[NBD];
        const R Wd = GENERAL ? rule.W : R(1), cww = Wd - Wd * Wd;
#pragma unroll
        for (int i = 0; i < NBD; ++i) {
          mz[i] = GENERAL ? Wd * mu[i] : mu[i];
          const R mr = GENERAL ? q_tr(q, mu[i]) : R(0);
#pragma unroll
          for (int j = 0; j < NBD; ++j) sz[i * NBD + j] = (GENERAL && j >= i) ? sg[i * NBD + j] + cww * (mr * mu[j]) : sg[i * NBD + j];
        }
#pragma unroll
        for (int j = 0; j < NBD; ++j) err[j] = mz[j] - zt[j];
        q_cost_share<NBD, NBD, QLD>(q, c.qr_diag != 0, kc.qr, err, sz, &pm, &pv, kz);
        if (a.zpost && live) {
          S* zo = a.zpost + ((long)t * C::E_ZPOST) * B + b;
#pragma unroll
          for (int j = 0; j < NBD; ++j) {
            if (r == 0 && in_col(j, D)) zo[(long)(4 * j + cc) * B] = (S)mz[j];
#pragma unroll
            for (int i = 0; i <= j; ++i)
              if ((i < j || up) && in_col(j, D)) zo[(long)(NZ + sym_lane(i, j) + sym_k(i, j)) * B] = (S)sz[i * NBD + j];
          }
        }
      } else {
        R am[NBD * NBZ], dm[NBD * NBZ], yc[NBZ], mz[NBZ], sz[NBZ * NBZ], err[NBZ];
        q_points<M, G, D, NZ>(q, rule.sf, mu, lt, ObserveF<M, R>{c.params}, am, dm, yc);
        q_moments<D, NZ, GENERAL>(q, rule, am, dm, yc, mz, sz);
#pragma unroll
        for (int j = 0; j < NBZ; ++j) err[j] = mz[j] - zt[j];
        q_cost_share<NBZ, NBZ, QLD>(q, c.qr_diag != 0, kc.qr, err, sz, &pm, &pv, kz);
        if (a.zpost && live) {
          S* zo = a.zpost + ((long)t * C::E_ZPOST) * B + b;
#pragma unroll
          for (int j = 0; j < NBZ; ++j) {
            if (r == 0 && in_col(j, NZ)) zo[(long)(4 * j + cc) * B] = (S)mz[j];
#pragma unroll
            for (int i = 0; i <= j; ++i)
              if ((i < j || up) && in_col(j, NZ)) zo[(long)(NZ + sym_lane(i, j) + sym_k(i, j)) * B] = (S)sz[i * NBZ + j];
          }
        }
      }
      acc_m += pm;
      acc_v += pv;
      if (a.cell_stats) {
        const R cm = q_sum16(q, pm), cv = q_sum16(q, pv);
        if (lead) {
          a.cell_stats[((long)t * 2 + 0) * B + b] = cm;
          a.cell_stats[((long)t * 2 + 1) * B + b] = cv;
        }
      }
    }
    // ---- controller (i2c.py:600-608) off the factor, as the lane kernels read it (cell_posterior):  K L_xx = L_ux, i.e. the blocked
    // BACK SUBSTITUTION  L_xx^T K^T = L_ux^T  with the upper blocks of lt -- rows = state index, the action columns of block column JU --
    // and the inverses of its diagonal blocks; sigK = L_uu L_uu^T; k = mu_u - K mu_x.
    // (K = -L_uu W_ux from an identity right-hand side of the elimination is the same matrix, but an explicit inverse loses a factor
    //  cond(L_xx) of accuracy: 6e-5 on the first iteration of the double cartpole, whose state covariance is ~1e-6 of the action's)
    {
      R kt[NBX], tk = R(0);
#pragma unroll
      for (int i = NBX - 1; i >= 0; --i) {
        const bool xr = in_row(i, NX) && (i != JU || r < CU);  // state rows of block row i
        R rhs = (xr && u_col) ? lt[i * NBD + JU] : R(0);
#pragma unroll
        for (int j = i + 1; j < NBX; ++j) {
          const bool xcj = in_col(j, NX) && (j != JU || cc < CU);  // state columns of block column j
          q_mfma(q, -q_tr(q, (xr && xcj) ? lt[i * NBD + j] : R(0)), kt[j], rhs);  // rhs -= L^T(i, j) K^T_j  (A operand = the transpose of what is passed)
        }
        const bool xci = in_col(i, NX) && (i != JU || cc < CU);
        kt[i] = R(0);
        q_mfma(q, q_tr(q, (xr && xci) ? aw[i] : R(0)), rhs, kt[i]);  // K^T_i = (L^T(i, i))^-1 rhs
        tk += kt[i] * q_tr(q, mu[i]);                                 // (row form of the state mean: lane (r, c) holds mu_x[4 i + r])
        io.st(out, s_kt[i], O_K + 4 * i, kt[i]);
      }
      const R kx = q_colsum(q, tk);  // K mu_x, column form over the action columns
      io.st(out, s_k, O_k, mu[JU] - kx);
      const R luu = (u_row && u_col) ? lt[JU * NBD + JU] : R(0);  // L_uu^T: the action entries of the factor's block (JU, JU)
      R sk = R(0);
      q_mfma(q, luu, luu, sk);  // L_uu L_uu^T
      io.st(out, s_sk, O_SK, sk);
    }
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      io.st(out, s_mu[j], 4 * j, mu[j]);
#pragma unroll
      for (int i = 0; i <= j; ++i) io.st(out, s_sg[i * NBD + j], D + sym_k(i, j), sg[i * NBD + j]);
    }
    // the state marginal that enters cell t - 1: the x entries of the smoothed joint
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      m3m[j] = in_col(j, NX) ? mu[j] : R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) s3m[i * NBX + j] = (i <= j && in_n(i, j, NX)) ? sg[i * NBD + j] : R(0);
    }
  }
  const R sm = q_sum16(q, acc_m), sv = q_sum16(q, acc_v);
  if (lead) {
    if constexpr (CHUNK) {
      qc.part[((long)qc.ch * 2 + 0) * (long)B + b] = sm;
      qc.part[((long)qc.ch * 2 + 1) * (long)B + b] = sv;
    } else {
      a.term_stats[B + b] = sm;
      a.term_stats[2 * B + b] = sv;
    }
    if (fail != 0 && a.status[b] == 0) a.status[b] = fail;
  }
}

// The COMPOSE pass of the chunked schedule in the same blocks (chunk_compose_body, i2c_cell.hpp: the x-marginal RTS recursion of one
// chunk as an affine map of the smoothed state that enters it):  a <- mu1 + Jx (a - m3f),  C <- S1 + Jx (C - S3f) Jx^T,  G <- Jx G
// from the chunk's last cell down, four trajectories per wavefront, rows fetched a cell ahead; the composite is stored in the lane
// kernels' format ([a | G row-major | C packed][B]) so that either stitch pass reads it. ~25 matrix + ~50 vector instructions per
// cell (NX = 6) where one lane needs 981 vector instructions.
template <class M, typename R, typename S>
I2C_HD inline void compose_quad8_body(const Consts<M, R>& c, const S* fwd, R* comp, const int ch, const int t_lo, const int t_hi, const int b, const bool live,
                                      const Quad<R>& q) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, D = C::D, NBX = (NX + 3) / 4;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX), EC = NX + NX * NX + sym(NX);
  static_assert(quad_backward8_exists<M>(), "quad compose pass: the d <= 8 geometry");
  const int r = q.r, cc = q.c;
  const unsigned long B = c.B;
  const bool up = r <= cc;
  const int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
  const int tri_c = cc * (cc + 1) / 2 + r, tri_d = hi * (hi + 1) / 2 + lo;
  auto sym_lane = [&](const int i, const int j) { return i == j ? tri_d + 4 * j * hi : tri_c + 4 * j * cc; };
  auto sym_k = [&](const int i, const int j) { return i == j ? 8 * j * j + 6 * j : 8 * j * j + 2 * j + 4 * i; };
  auto in_row = [&](const int i, const int N) { return 4 * i + 3 < N || 4 * i + r < N; };
  auto in_col = [&](const int j, const int N) { return 4 * j + 3 < N || 4 * j + cc < N; };
  auto in_n = [&](const int i, const int j, const int N) { return in_row(i, N) && in_col(j, N); };
  const QIO8<R, S> io{(unsigned)(B * sizeof(S)), (unsigned)b * (unsigned)sizeof(S)};
  const QIO8<R, R> ior{(unsigned)(B * sizeof(R)), (unsigned)b * (unsigned)sizeof(R)};
  unsigned l_v[NBX], l_s[NBX * NBX], l_j[NBX * NBX], o_a[NBX], o_g[NBX * NBX], o_c[NBX * NBX];
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    l_v[j] = io.lane(in_col(j, NX), cc);
    o_a[j] = ior.lane(live && r == 0 && in_col(j, NX), cc);
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
      l_s[i * NBX + j] = io.lane(i <= j && in_n(i, j, NX), sym_lane(i, j));  // upper blocks (the diagonal ones as full symmetric blocks)
      l_j[i * NBX + j] = io.lane(in_n(i, j, NX), cc * NX + r);               // Jx^T: J is stored [joint index][state]
      o_g[i * NBX + j] = ior.lane(live && in_n(i, j, NX), r * NX + cc);
      o_c[i * NBX + j] = ior.lane(live && i <= j && (i < j || up) && in_col(j, NX), sym_lane(i, j));
    }
  }
  R n_mu[NBX], n_s1[NBX * NBX], n_m3[NBX], n_s3[NBX * NBX], n_jt[NBX * NBX];
  auto fetch = [&](const int tc) {
    const Window f = make_window(fwd + (unsigned long)tc * C::E_FWD * B, (unsigned long)C::E_FWD * io.rb);
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      n_mu[j] = io.ld(f, l_v[j], 4 * j);
      n_m3[j] = io.ld(f, l_v[j], O_MU3 + 4 * j);
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        if (i <= j) {
          n_s1[i * NBX + j] = io.ld(f, l_s[i * NBX + j], D + sym_k(i, j));
          n_s3[i * NBX + j] = io.ld(f, l_s[i * NBX + j], O_S3 + sym_k(i, j));
        }
        n_jt[i * NBX + j] = io.ld(f, l_j[i * NBX + j], O_J + 4 * j * NX + 4 * i);
      }
    }
  };
  fetch(t_hi - 1);
  R av[NBX], g[NBX * NBX], cm[NBX * NBX];
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    av[j] = R(0);
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
      g[i * NBX + j] = (i == j && r == cc && 4 * i + r < NX) ? R(1) : R(0);
      cm[i * NBX + j] = R(0);
    }
  }
  for (int t = t_hi - 1; t >= t_lo; --t) {
    R mu1[NBX], s1[NBX * NBX], jt[NBX * NBX], drr[NBX], ds[NBX * NBX], p1[NBX * NBX], gn[NBX * NBX];
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      mu1[j] = n_mu[j];
      drr[j] = q_tr(q, av[j] - n_m3[j]);  // row form
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        s1[i * NBX + j] = i <= j ? n_s1[i * NBX + j] : R(0);
        jt[i * NBX + j] = n_jt[i * NBX + j];
        if (i <= j) ds[i * NBX + j] = cm[i * NBX + j] - n_s3[i * NBX + j];
        p1[i * NBX + j] = R(0);
        gn[i * NBX + j] = R(0);
      }
    }
    fetch(t > t_lo ? t - 1 : t_lo);
#pragma unroll
    for (int i = 0; i < NBX; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) ds[i * NBX + j] = q_tr(q, ds[j * NBX + i]);
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      R ts = R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) ts += jt[i * NBX + j] * drr[i];
      av[j] = mu1[j] + q_colsum(q, ts);
    }
    q_tn<NBX, NBX, NBX>(q, ds, jt, p1);               // (C - S3f) Jx^T
    q_tn<NBX, NBX, NBX, false, true>(q, jt, p1, s1);  // S1 + Jx (...), upper blocks
    q_tn<NBX, NBX, NBX>(q, jt, g, gn);                // Jx G
#pragma unroll
    for (int i = 0; i < NBX; ++i) {  // (the diagonal blocks exactly symmetric again: see the RTS update of the walk)
      const R st = q_tr(q, s1[i * NBX + i]);
      s1[i * NBX + i] = up ? s1[i * NBX + i] : st;
    }
#pragma unroll
    for (int k = 0; k < NBX * NBX; ++k) cm[k] = s1[k], g[k] = gn[k];
  }
  const Window out = make_window(comp + (unsigned long)ch * EC * B, (unsigned long)EC * ior.rb);
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    ior.st(out, o_a[j], 4 * j, av[j]);
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
      ior.st(out, o_g[i * NBX + j], NX + 4 * i * NX + 4 * j, g[i * NBX + j]);
      if (i <= j) ior.st(out, o_c[i * NBX + j], NX + NX * NX + sym_k(i, j), cm[i * NBX + j]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Cubature Kalman filter step of the MPC state estimator (PartiallyObservedMpcPolicy.filter, i2c/policy/mpc.py:125-145), d = 16
// ------------------------------------------------------------------------------------------
// FOUR beliefs per wavefront, the same blocks and phases as the forward cell: chol of the belief, 2 nx (+ centre) state-only sigma
// points through the dynamics with the applied action appended (mpc.py:129-137), chol of the prediction, the points through
// sys.measure, Kalman update on the measurement (mpc.py:139-145). With it a control step of the 12-state quadrotor (filter +
// sweeps, i2c_mpc_step) runs on matrix-instruction kernels end to end; the estimator's rule is CubatureQuadrature(1, 0, 0)
// whatever the graph infers with (mpc.py:121-123: Impl::filter_problem), so the unit-weight forms of q_moments always apply.
template <class M, typename R> struct QKConst {
  static constexpr int QLD = QG<M>::QLD;
  R eta[QLD * QLD], zeta[QLD * QLD];  // sig_eta (nx x nx), sig_zeta (ny x ny), zero-padded
};
template <class M, typename R, class DST> I2C_FN void qkconst_fill(DST& k, const Consts<M, R>* c, const R* zeta, const int tid, const int nthreads) {
  constexpr int NX = M::NX, NY = M::NY, QLD = QG<M>::QLD;
  for (int e = tid; e < QLD * QLD; e += nthreads) {
    const int i = e / QLD, j = e % QLD;
    k.eta[e] = (i < NX && j < NX) ? c->sig_eta[tri_any(i, j)] : R(0);
    k.zeta[e] = (i < NY && j < NY) ? zeta[tri_any(i, j)] : R(0);
  }
}
template <class M> constexpr bool quad_ckf_exists() { return QG<M>::WIDE && M::NX % 4 == 0 && M::NX <= 12 && M::NY <= 12; }

template <class M, typename R, class KC>
I2C_HD inline void ckf_quad_body(const Consts<M, R>& c, const KC& kc, const CkfArgs<R>& a, const int b, const bool live, const Quad<R>& q) {
  using C = Consts<M, R>;
  using G = QG<M>;
  constexpr int NX = C::NX, NU = C::NU, NY = M::NY, NBX = NX / 4, NBY = (NY + 3) / 4, QLD = G::QLD;
  static_assert(quad_ckf_exists<M>(), "quad filter: the d = 16 geometry, nx a multiple of four");
  const int r = q.r, cc = q.c;
  const long B = c.B;
  const Rule<R>& rule = c.rule_x;
  auto xrow = [&](const int i) { return 4 * i + r; };
  auto xcol = [&](const int j) { return 4 * j + cc; };
  R mu[NBX], S[NBX * NBX], u[NU], yt[NBY];
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    mu[j] = a.mu[(long)xcol(j) * B + b];
#pragma unroll
    for (int i = 0; i < NBX; ++i) S[i * NBX + j] = j >= i ? a.cov[(long)w_symidx(xrow(i), xcol(j)) * B + b] : R(0);  // upper blocks, the diagonal ones full
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = a.u[(long)i * B + b];
#pragma unroll
  for (int j = 0; j < NBY; ++j) {
    const int col = xcol(j);
    const R v = a.y[(long)(col < NY ? col : 0) * B + b];
    yt[j] = col < NY ? v : R(0);
  }
  const int kz = (int)opaque_uniform(0u);
  bool ok;
  // prediction: chol(S), the points through the dynamics, moments + process noise
  R mf[NBX], Sf[NBX * NBX];
  {
    R tmp[NBX * NBX], lt[NBX * NBX], am[NBX * NBX], dm[NBX * NBX], yc[NBX];
#pragma unroll
    for (int k = 0; k < NBX * NBX; ++k) tmp[k] = S[k];
    ok = q_elim<NX, 0, 0>(q, tmp, (R*)nullptr, (R*)nullptr, lt);
    q_points<M, G, NX, NX>(q, rule.sf, mu, lt, DynamicsFixedUF<M, R>{c.params, u}, am, dm, yc);
    q_moments<NX, NX>(q, rule, am, dm, yc, mf, Sf);
#pragma unroll
    for (int i = 0; i < NBX; ++i)
#pragma unroll
      for (int j = 0; j < NBX; ++j) Sf[i * NBX + j] = j >= i ? Sf[i * NBX + j] + q_ldc<QLD>(q, kc.eta, i, j, kz) : R(0);
  }
  // innovation: chol(Sf), the points through sys.measure, K = sig_xy sig_y^-1 (as a Cholesky elimination), update
  {
    R tmp[NBX * NBX], lt[NBX * NBX], am[NBX * NBY], dm[NBX * NBY], yc[NBY], my[NBY], Sy[NBY * NBY], Syx[NBY * NBX];
#pragma unroll
    for (int k = 0; k < NBX * NBX; ++k) tmp[k] = Sf[k];
    ok = q_elim<NX, 0, 0>(q, tmp, (R*)nullptr, (R*)nullptr, lt) && ok;
    q_points<M, G, NX, NY>(q, rule.sf, mf, lt, MeasureF<M, R>{c.params}, am, dm, yc);
    q_moments<NX, NY>(q, rule, am, dm, yc, my, Sy);
#pragma unroll
    for (int i = 0; i < NBY; ++i)
#pragma unroll
      for (int j = i; j < NBY; ++j) Sy[i * NBY + j] += q_ldc<QLD>(q, kc.zeta, i, j, kz);
#pragma unroll
    for (int k = 0; k < NBY * NBX; ++k) Syx[k] = R(0);
    q_tn<NBX, NBY, NBX, false, false, true>(q, dm, lt, Syx);  // cov(y, x) = wi sf [d_p]^T L^T
    const R cw = rule.wi * rule.sf;
#pragma unroll
    for (int k = 0; k < NBY * NBX; ++k) Syx[k] *= cw;
    ok = q_kalman<NX, NY>(q, mf, Sf, my, Sy, Syx, yt) && ok;
  }
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    if (live && r == 0) a.mu[(long)xcol(j) * B + b] = mf[j];
#pragma unroll
    for (int i = 0; i <= j; ++i)
      if (live && xrow(i) <= xcol(j)) a.cov[(long)w_symidx(xrow(i), xcol(j)) * B + b] = Sf[i * NBX + j];
  }
  if (live && q.p() == 0 && !ok && a.status[b] == 0) a.status[b] = (9 << 16) | 1;  // I2C_FAIL_FILTER, as set_status(.., 9, 0)
}

// ------------------------------------------------------------------------------------------
// Closed-loop propagation (i2c.py:150-199, 1247-1251), d = 16 with identity observations (the 12-state quadrotor)
// ------------------------------------------------------------------------------------------
// FOUR trajectories per wavefront, the blocks and phases of the forward cell: the (optional) pdf-ratio scaling of the feedback gain
// (i2c.py:160-167: one elimination with the state offset as right-hand side; a failed one leaves K unscaled like the reference,
// which logs the exception), the joint N(xu) from the state marginal and the controller -- sig_0 = [[S, S K^T], [K S, sig_u]] with
// sig_u = the posterior's action block (feed-forward cells) or K S K^T + sig_uu_m - K sig_xx_m K^T (feedback cells,
// i2c.py:169-171) --, the expected cost of the propagated observation (= the joint: identity observation), the 2 d sigma
// points through the dynamics, and at the end of the horizon the KL divergence to the terminal state prior (covariance control,
// i2c.py:1012-1019). With it covariance control / calibrate_alpha of the 12-state model no longer drop to the group kernels,
// whose waves serialise beyond 1024 (4096 trajectories). Posterior buffer trajectory-major, propagation buffer [T][E][B].
template <class M, typename R> struct QPConst {
  static constexpr int QLD = QG<M>::QLD;
  R qr[QLD * QLD], eta[QLD * QLD], sxT[QLD * QLD];  // blkdiag(Q, R), sig_eta, sig_x_terminal
  R zg[QLD], mxT[QLD];
};
template <class M, typename R, class DST> I2C_FN void qpconst_fill(DST& k, const Consts<M, R>* c, const int tid, const int nthreads) {
  constexpr int NX = M::NX, NZ = M::NZ, QLD = QG<M>::QLD;
  for (int e = tid; e < QLD * QLD; e += nthreads) {
    const int i = e / QLD, j = e % QLD;
    k.qr[e] = (i < NZ && j < NZ) ? c->QR[tri_any(i, j)] : R(0);
    k.eta[e] = (i < NX && j < NX) ? c->sig_eta_w[tri_any(i, j)] : R(0);  // W sig_eta (quadrature.py:57; W = 1 for the unit rule)
    k.sxT[e] = (i < NX && j < NX) ? c->sig_x_term[tri_any(i, j)] : R(0);
  }
  for (int e = tid; e < QLD; e += nthreads) {
    k.zg[e] = e < NZ ? c->zg[e] : R(0);
    k.mxT[e] = e < NX ? c->mu_x_term[e] : R(0);
  }
}
template <class M> constexpr bool quad_propagate_exists() {
  return QG<M>::WIDE && M::NX % 4 == 0 && (M::NX + M::NU) % 4 == 0 && M::NU <= 4 && M::NZ == M::NX + M::NU && st_identity<ObsStruct<M>, M::NZ>();
}

template <class M, typename R, bool GENERAL = false, class KC>
I2C_HD inline void propagate_quad_body(const Consts<M, R>& c, const KC& kc, const PropArgs<R>& a, const int b, const bool live, const Quad<R>& qw) {
  using C = Consts<M, R>;
  using G = QG<M>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D, NBX = NX / 4, NBD = D / 4, JU = NX / 4, QLD = G::QLD;
  static_assert(quad_propagate_exists<M>(), "quad propagation: d = 16 geometry, identity observation of the joint");
  constexpr int O_K = D + sym(D), O_X3 = D + sym(D), O_SX3 = O_X3 + NX;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned WS = sizeof(R), bo = (unsigned)b * WS, rb = (unsigned)(B * WS);
  const Rule<R>& rule = c.rule_xu;
  int fail = 0;
  auto note = [&](const bool ok, const int t) { fail = (fail == 0 && !ok) ? ((8 << 16) | (t + 1)) : fail; };  // I2C_FAIL_PROPAGATE

  // state marginal carried along the chain: mean in column form, covariance in UPPER blocks (the diagonal ones full)
  R mx[NBX], sx[NBX * NBX];
#pragma unroll
  for (int j = 0; j < NBX; ++j) {
    mx[j] = a.x0[(long)(4 * j + qw.c) * B + b];
#pragma unroll
    for (int i = 0; i < NBX; ++i) sx[i * NBX + j] = j >= i ? a.sig_x0[(long)w_symidx(4 * i + qw.r, 4 * j + qw.c) * B + b] : R(0);
  }
  const Window ffw = make_window(a.ff, (unsigned long)T), exw = make_window(a.expert ? a.expert : a.ff, (unsigned long)T);
  const Window zw = make_window(c.z_per_cell ? a.z : a.x0, (c.z_per_cell ? (unsigned long)T * NZ : 1ul) * B * sizeof(R));  // (x0 row 0: a valid dummy)
  R acc_m = R(0), acc_v = R(0);

  // The posterior rows of a cell do not depend on the recursion: they are fetched ONE CELL AHEAD (issued once the current cell's have
  // been consumed; see forward_wave_body), through the buffer path and without branches. (trajectory-major: a cell of a trajectory
  // is contiguous)
  R nx_pmu[NBD], nx_pj[NBD * NBD], nx_kt[NBX], nx_zt[NBD];
  int nx_ff, nx_ex;
  auto fetch_rows = [&](const int tc) {
    const int r = qw.r, cc = qw.c;
    const int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
    const int tri_c = cc * (cc + 1) / 2 + r, tri_d = hi * (hi + 1) / 2 + lo;
    auto sym_lane = [&](const int i, const int j) { return i == j ? tri_d + 4 * j * hi : tri_c + 4 * j * cc; };
    auto sym_k = [&](const int i, const int j) { return i == j ? 8 * j * j + 6 * j : 8 * j * j + 2 * j + 4 * i; };
    const int trc = c.row(tc);
    const QIO<R, R, true> pri{make_window(a.post + (unsigned long)trc * C::E_POST * B, (unsigned long)C::E_POST * B * WS), WS, (unsigned)b * (unsigned)C::E_POST * WS};
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      nx_pmu[j] = pri.ld(cc, 4 * j);
#pragma unroll
      for (int i = 0; i < NBD; ++i) nx_pj[i * NBD + j] = i <= j ? pri.ld(sym_lane(i, j), D + sym_k(i, j)) : R(0);  // upper blocks, the diagonal ones full
      // (the per-cell target or a discarded dummy; the choice is made where the value is used)
      nx_zt[j] = wld<R>(zw, 0u, c.z_per_cell ? (unsigned)((((unsigned long)trc * NZ + 4 * j + cc) * B + b) * sizeof(R)) : (unsigned)(b * sizeof(R)));
    }
#pragma unroll
    for (int i = 0; i < NBX; ++i) nx_kt[i] = pri.ld(cc < NU ? cc * NX + r : 0, O_K + 4 * i);  // K^T, block (i, JU): row = state, column = action (masked where used)
    nx_ff = (int)wld_u8(ffw, (unsigned)trc);
    nx_ex = (int)wld_u8(exw, (unsigned)trc);
  };
  fetch_rows(0);
  // settled before the loop (loads pending on the loop-entry path cost a vmcnt(0) in every cell: forward_wave_body)
#pragma unroll
  for (int k = 0; k < NBD; ++k) nx_pmu[k] = opaque(nx_pmu[k]), nx_zt[k] = opaque(nx_zt[k]);
#pragma unroll
  for (int k = 0; k < NBD * NBD; ++k) nx_pj[k] = opaque(nx_pj[k]);
#pragma unroll
  for (int k = 0; k < NBX; ++k) nx_kt[k] = opaque(nx_kt[k]);
  nx_ff = (int)opaque((unsigned)nx_ff), nx_ex = (int)opaque((unsigned)nx_ex);

  for (int t = 0; t < T; ++t) {
    const int kz = (int)opaque_uniform(0u);
    const Quad<R> q = q_opaque(qw);
    const int r = q.r, cc = q.c;
    const int hi = r > cc ? r : cc, lo = r > cc ? cc : r;
    const int tri_c = cc * (cc + 1) / 2 + r, tri_d = hi * (hi + 1) / 2 + lo;
    auto sym_lane = [&](const int i, const int j) { return i == j ? tri_d + 4 * j * hi : tri_c + 4 * j * cc; };
    auto sym_k = [&](const int i, const int j) { return i == j ? 8 * j * j + 6 * j : 8 * j * j + 2 * j + 4 * i; };
    R pmu[NBD], pj[NBD * NBD], kt[NBX], zt[NBD];
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      pmu[j] = nx_pmu[j];
#pragma unroll
      for (int i = 0; i < NBD; ++i) pj[i * NBD + j] = i <= j ? nx_pj[i * NBD + j] : R(0);
      zt[j] = c.z_per_cell ? nx_zt[j] : q_ldv(q, kc.zg, j, kz);
    }
#pragma unroll
    for (int i = 0; i < NBX; ++i) kt[i] = cc < NU ? nx_kt[i] : R(0);
    const bool ff = w_uniform(nx_ff) != 0;
    const bool ex = a.expert ? w_uniform(nx_ex) != 0 : c.use_expert != 0;

    // state offset to the posterior's state mean, row form; pdf-ratio scaling of the gain (expert controller, feedback cells)
    R dr[NBX];
#pragma unroll
    for (int i = 0; i < NBX; ++i) dr[i] = q_tr(q, mx[i] - pmu[i]);
    R rho = R(1);
    if (!ff && ex) {
      R sm[NBX * NBX], rhs[NBX], lts[NBX * NBX];
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
#pragma unroll
        for (int j = 0; j < NBX; ++j) sm[i * NBX + j] = j >= i ? pj[i * NBD + j] + sx[i * NBX + j] : R(0);
        rhs[i] = cc == 0 ? dr[i] : R(0);
      }
      const bool ok = q_elim<NX, 1, 0>(q, sm, rhs, (R*)nullptr, lts);
      R ysq = R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) ysq += cc == 0 ? rhs[i] * rhs[i] : R(0);
      const R maha = q_bcq<0>(q, q_colsum(q, ysq));
      rho = ok ? r_exp_sc(R(-0.5) * maha) : R(1);  // the reference logs the exception and keeps K unscaled (i2c.py:166-167)
    }
    // joint N(xu): sig_0 = [[S, S K^T], [K S, sig_u]], mu_0 = [mx; mu_u + K (mx - x_ref)]
    R mu0[NBD], s0[NBD * NBD];
    {
      R ktm[NBX], sxf[NBX * NBX], g[NBX], kxk = R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) {
        ktm[i] = rho * kt[i];
        g[i] = R(0);
#pragma unroll
        for (int j = 0; j < NBX; ++j) sxf[i * NBX + j] = j >= i ? sx[i * NBX + j] : q_tr(q, sx[j * NBX + i]);
      }
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int k = 0; k < NBX; ++k) q_mfma(q, sxf[k * NBX + i], ktm[k], g[i]);  // (S K^T)[i] = sum_k S[i][k] K^T[k]
#pragma unroll
      for (int i = 0; i < NBX; ++i) q_mfma(q, ktm[i], g[i], kxk);                 // K S K^T
      R suu = pj[JU * NBD + JU];
      if (!ff) {  // sig_u = K S K^T + sig_uu_m - K sig_xx_m K^T (i2c.py:169-171)
        R pxf[NBX * NBX], h[NBX], kpk = R(0);
#pragma unroll
        for (int i = 0; i < NBX; ++i) {
          h[i] = R(0);
#pragma unroll
          for (int j = 0; j < NBX; ++j) pxf[i * NBX + j] = j >= i ? pj[i * NBD + j] : q_tr(q, pj[j * NBD + i]);
        }
#pragma unroll
        for (int i = 0; i < NBX; ++i)
#pragma unroll
          for (int k = 0; k < NBX; ++k) q_mfma(q, pxf[k * NBX + i], ktm[k], h[i]);
#pragma unroll
        for (int i = 0; i < NBX; ++i) q_mfma(q, ktm[i], h[i], kpk);
        suu -= kpk;
      }
#pragma unroll
      for (int i = 0; i < NBD; ++i)
#pragma unroll
        for (int j = 0; j < NBD; ++j) {
          R v = R(0);
          if (i < NBX && j < NBX) v = j >= i ? sx[(i < NBX ? i : 0) * NBX + (j < NBX ? j : 0)] : R(0);
          else if (i < NBX && j == JU) v = g[i < NBX ? i : 0];
          else if (i == JU && j == JU) v = ff ? suu : kxk + suu;  // (feed-forward cells: the action marginal ITSELF, while the joint still carries K S: i2c.py:155-157, 173-179)
          s0[i * NBD + j] = v;
        }
      R tt = R(0);
#pragma unroll
      for (int i = 0; i < NBX; ++i) tt += ktm[i] * dr[i];
      const R kd = ff ? R(0) : q_colsum(q, tt);  // feed-forward cells: the action marginal itself (i2c.py:155-157)
#pragma unroll
      for (int j = 0; j < NBD; ++j) mu0[j] = j < NBX ? mx[j < NBX ? j : 0] : pmu[j] + kd;
    }
    fetch_rows(t + 1 < T ? t + 1 : t);  // this cell's rows are consumed: the next cell's, a cell ahead
    const QIO<R, R, false> out{make_window(a.prop + (unsigned long)t * C::E_PROP * B, (unsigned long)C::E_PROP * rb), rb, bo};
#pragma unroll
    for (int j = 0; j < NBD; ++j) {
      out.st_if(live && r == 0, cc, 4 * j, mu0[j]);
#pragma unroll
      for (int i = 0; i <= j; ++i) out.st_if(live && (i < j || r <= cc), sym_lane(i, j), D + sym_k(i, j), s0[i * NBD + j]);
    }
    // expected cost of the propagated observation (= the joint: identity observation; compute_cost_gaussian, i2c.py:1034-1043)
    {
      R err[NBD], pm, pv;
      if constexpr (GENERAL) {  // the identity observation's exact moments under general weights: W mu, sig + (W - W^2) mu mu^T
        R mzg[NBD], szg[NBD * NBD];
        const R Wd = rule.W, cwd = Wd - Wd * Wd;
#pragma unroll
        for (int i = 0; i < NBD; ++i) {
          mzg[i] = Wd * mu0[i];
          const R mrd = q_tr(q, mu0[i]);
#pragma unroll
          for (int j = 0; j < NBD; ++j) szg[i * NBD + j] = j >= i ? s0[i * NBD + j] + cwd * (mrd * mu0[j]) : s0[i * NBD + j];
        }
#pragma unroll
        for (int j = 0; j < NBD; ++j) err[j] = mzg[j] - zt[j];
        q_cost_share<NBD, NBD, QLD>(q, c.qr_diag != 0, kc.qr, err, szg, &pm, &pv, kz);
      } else {
#pragma unroll
        for (int j = 0; j < NBD; ++j) err[j] = mu0[j] - zt[j];
        q_cost_share<NBD, NBD, QLD>(q, c.qr_diag != 0, kc.qr, err, s0, &pm, &pv, kz);
      }
      acc_m += pm;
      acc_v += pv;
    }
    // dynamics push-through
    {
      R tmp[NBD * NBD], lt[NBD * NBD], am[NBD * NBX], dm[NBD * NBX], yc[NBX], sy[NBX * NBX];
#pragma unroll
      for (int k = 0; k < NBD * NBD; ++k) tmp[k] = s0[k];
      note(q_elim<D, 0, 0>(q, tmp, (R*)nullptr, (R*)nullptr, lt), t);
      q_points<M, G, D, NX, GENERAL>(q, rule.sf, mu0, lt, DynamicsF<M, R>{c.params}, am, dm, yc);  // (d = 16: no spare pair row, the centre is an extra pass)
      q_moments<D, NX, GENERAL>(q, rule, am, dm, yc, mx, sy);
#pragma unroll
      for (int i = 0; i < NBX; ++i)
#pragma unroll
        for (int j = 0; j < NBX; ++j) sx[i * NBX + j] = j >= i ? sy[i * NBX + j] + q_ldc<QLD>(q, kc.eta, i, j, kz) : R(0);
    }
#pragma unroll
    for (int j = 0; j < NBX; ++j) {
      out.st_if(live && r == 0, cc, O_X3 + 4 * j, mx[j]);
#pragma unroll
      for (int i = 0; i <= j; ++i) out.st_if(live && (i < j || r <= cc), sym_lane(i, j), O_SX3 + sym_k(i, j), sx[i * NBX + j]);
    }
  }
  // KL(N(mx, sx) || terminal state prior) (covariance control): with L1 = chol(sx), L2 = chol(sig_T):
  //   2 KL = 2 sum log(L2_ii / L1_ii) + ||L2^-1 L1||_F^2 + |L2^-1 (mu_T - mx)|^2 - nx
  R kl = R(0);
  if (c.has_x_terminal) {
    const Quad<R>& q = qw;
    const int kz = (int)opaque_uniform(0u);
    R t1[NBX * NBX], lt1[NBX * NBX], sT[NBX * NBX], rl[NBX * NBX], dq[NBX], lt2[NBX * NBX];
#pragma unroll
    for (int k = 0; k < NBX * NBX; ++k) t1[k] = sx[k];
    const bool ok1 = q_elim<NX, 0, 0>(q, t1, (R*)nullptr, (R*)nullptr, lt1);
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
#pragma unroll
      for (int j = 0; j < NBX; ++j) {
        sT[i * NBX + j] = j >= i ? q_ldc<QLD>(q, kc.sxT, i, j, kz) : R(0);
        rl[i * NBX + j] = j <= i ? q_tr(q, lt1[j * NBX + i]) : R(0);  // L1 = (L1^T)^T, block lower triangular
      }
      const R dqr = q_tr(q, q_ldv(q, kc.mxT, i, kz) - mx[i]);  // (every lane takes part in the matrix instruction)
      dq[i] = q.c == 0 ? dqr : R(0);
    }
    const bool ok2 = q_elim<NX, NBX, 1>(q, sT, rl, dq, lt2);
    R part = R(0);
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
      part += dq[i] * dq[i];
#pragma unroll
      for (int j = 0; j < NBX; ++j) part += rl[i * NBX + j] * rl[i * NBX + j];
      const R l1 = lt1[i * NBX + i], l2 = lt2[i * NBX + i];
      part += q.r == q.c ? R(2) * (r_log(q.r == q.c ? l2 : R(1)) - r_log(q.r == q.c ? l1 : R(1))) : R(0);
    }
    kl = R(0.5) * (q_sum16(q, part) - R(NX));
    note(ok1 && ok2, T - 1);
  }
  const R sum_m = q_sum16(qw, acc_m), sum_v = q_sum16(qw, acc_v);
  if (live && qw.p() == 0) {
    a.prop_stats[b] = sum_m;
    a.prop_stats[B + b] = sum_v;
    a.prop_stats[2 * B + b] = kl;
    if (fail != 0 && a.status[b] == 0) a.status[b] = fail;
  }
}

}  // namespace i2c
