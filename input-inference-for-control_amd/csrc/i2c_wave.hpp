// One WAVEFRONT per trajectory: the cubature cells for models whose joint state-action dimension is exactly the width of the
// fp64 matrix instruction of gfx950 (d = nx + nu = 16: the 12-state quadrotor).
//
//   * every 16 x 16 block of a cell lives in the ACCUMULATOR LAYOUT of v_mfma_f64_16x16x4_f64: lane l = 16 q + j holds column j,
//     rows q, q + 4, q + 8, q + 12 (four fp64 registers per block). In that layout a block is at once the B operand of the
//     instruction for each of its four row blocks and -- transposed -- the A operand, so  X^T Y  of two resident blocks is four
//     matrix instructions with NO operand movement (w_tn). Symmetric blocks (covariances) are their own transpose: products
//     with them are free of any layout change.
//   * the factorisations are blocked (4 x 4 pivots): the diagonal block goes to every lane (v_readlane), which factors and
//     inverts it in registers; the block row is scaled and the trailing matrix AND up to two right-hand sides are updated by
//     matrix instructions (w_elim). The solves of a cell are right-hand sides of those eliminations -- there is no separate
//     triangular-solve primitive, and only explicit products with  L^-1 R  are ever formed.
//   * vectors live as "column form" (lane (q, j) holds element j, replicated over q) and are turned into "row form" (elements
//     q + 4 v) through a 16-element LDS slot; sums over the four 16-lane rows use v_permlane32_swap / v_permlane16_swap.
//   * the 2 d sigma points of a transform are evaluated by 32 lanes at once (lane p: + column p, lane 16 + p: - column p, lane 32:
//     the centre), one model evaluation per lane.
// The math is the reference's I2cCell (i2c/i2c.py:350-447, 544-610) and QuadratureInference (i2c/inference/quadrature.py:15-58)
// in the centred pairwise form of sp_transform (i2c_cell.hpp); requirements on the model are static_asserted in the bodies.
// HBM layout is the common [T][E][B] one: a wave touches one 8-byte element per row, so the launch maps 16 consecutive
// trajectories (one 128-byte line of every row) onto workgroups of the SAME XCD (k_wave in i2c_impl.hpp).
//
// Host simulation (tests only): the 64 lanes are 64 threads; every cross-lane instruction is an exchange through a shared
// buffer between two barriers.
#pragma once
#include "i2c_group.hpp"
#include "i2c_linearize.hpp"

namespace i2c {

// Diagnostic build only (-DI2C_WAVE_STAMPS, never in the shipped library): s_memtime stamps at the phase boundaries of the forward
// cell, printed by trajectory 0 (tools/build_variant_tu.py).
#if defined(I2C_WAVE_STAMPS) && !defined(I2C_HOST_SIM)
#define I2C_WSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_last; stamp_last = now_; } while (0)
#else
#define I2C_WSTAMP(i) do { } while (0)
#endif

constexpr int WLD = 17;  // row stride (elements) of a 16 x 16 block in LDS: column reads of 16 lanes hit 16 different bank pairs

// LDS region of one wave (elements)
struct WaveLds {
  static constexpr int MAT = 16 * WLD;                 // one 16 x 16 block
  static constexpr int O_MAT = 0;                      // block 0: sigma-point directions (rows of L^T)
  static constexpr int YLD = 13;                       // row stride of the evaluation outputs (nx = 12 values per point)
  static constexpr int O_Y = O_MAT + MAT;              // 33 rows: + points, - points, centre
  static constexpr int NVEC = 6;
  static constexpr int O_VEC = O_Y + 33 * YLD + 3;     // vector slots of 16
  static constexpr int O_DG = O_VEC + NVEC * 16;       // 4 x 4 pivot block (w_pivot_block through LDS)
  static constexpr int O_DUMP = O_DG + 16;             // where masked-off lanes park that store (one slot per lane)
  static constexpr int SIZE = O_DUMP + 64;
};

template <typename R> struct Wave {
  int l, q, j;  // lane, its 16-lane row (l >> 4) and column (l & 15)
  lds_ptr<R> sh;
#ifdef I2C_HOST_SIM
  HostBarrier* bar;
  R* xch;  // 128 slots: operands of the emulated cross-lane instructions
#endif
  I2C_MEM lds_ptr<R> mat() const { return sh + WaveLds::O_MAT; }
  I2C_MEM lds_ptr<R> ybuf() const { return sh + WaveLds::O_Y; }
  I2C_MEM lds_ptr<R> vec(int i) const { return sh + WaveLds::O_VEC + i * 16; }
  I2C_MEM int row(int v) const { return q + 4 * v; }  // matrix row of accumulator register v
  // orders this wave's LDS writes before its later LDS reads (and earlier reads before later writes)
  I2C_MEM void sync() const {
#ifdef I2C_HOST_SIM
    bar->wait();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
  }
};

// a value every lane holds alike, made PROVABLY uniform (an SGPR): branches on it are scalar branches, not EXEC-masked regions
I2C_FN int w_uniform(const int x) {
#ifdef I2C_HOST_SIM
  return x;
#else
  return __builtin_amdgcn_readfirstlane(x);
#endif
}

// ---- the cross-lane instructions -------------------------------------------------------------------------------------------
// acc (16 x 16, accumulator layout) += A (16 x 4) B (4 x 16): lane (q, i) supplies A[i][q], lane (q, j) supplies B[q][j]
template <typename R> I2C_FN void w_mfma(const Wave<R>& w, const R a, const R b, R* acc) {
#ifdef I2C_HOST_SIM
  w.bar->wait();
  w.xch[w.l] = a;
  w.xch[64 + w.l] = b;
  w.bar->wait();
  for (int v = 0; v < 4; ++v) {
    R s = acc[v];
    for (int k = 0; k < 4; ++k) s = std::fma(w.xch[16 * k + w.row(v)], w.xch[64 + 16 * k + w.j], s);
    acc[v] = s;
  }
#else
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 c = {acc[0], acc[1], acc[2], acc[3]};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  acc[0] = c[0];
  acc[1] = c[1];
  acc[2] = c[2];
  acc[3] = c[3];
#endif
}
// One ROW BLOCK (4 x 16, a single accumulator register: lane (q, j) holds row q, column j) times a 4 x 4 matrix from the left, on
// v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 blocks, block n in lanes {16 k + 4 n + m}; tools/micro/mfma_f64_4x4.hip):
//   returns A b  with  A[i][k] supplied by lane (k, .. , i) -- i.e. every lane (q, j) passes A[j & 3][q] -- and b the row block.
// A quarter of the 64 clocks of the 16 x 16 x 4 instruction, whose tile would be 3/4 unused here (round-3 review).
template <typename R> I2C_FN R w_mfma4(const Wave<R>& w, const R a, const R b) {
#ifdef I2C_HOST_SIM
  w.bar->wait();
  w.xch[w.l] = a;
  w.xch[64 + w.l] = b;
  w.bar->wait();
  const int i = w.l >> 4, n = (w.l >> 2) & 3, jj = w.l & 3;
  R s = R(0);
  for (int k = 0; k < 4; ++k) s = std::fma(w.xch[16 * k + 4 * n + i], w.xch[64 + 16 * k + 4 * n + jj], s);
  return s;
#else
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
#endif
}
// acc += X^T Y over the first KB row blocks of X and Y (both in the accumulator layout); NEG: acc -= X^T Y
template <int KB, bool NEG = false, typename R> I2C_FN void w_tn(const Wave<R>& w, const R* x, const R* y, R* acc) {
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) w_mfma(w, NEG ? -x[kb] : x[kb], y[kb], acc);
}
// value held by lane K of the caller's 16-lane row (DPP row_newbcast)
template <int K, typename R> I2C_FN R w_bcast(const Wave<R>& w, const R x) {
#ifdef I2C_HOST_SIM
  w.bar->wait();
  w.xch[w.l] = x;
  w.bar->wait();
  return w.xch[(w.l & ~15) + K];
#else
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#endif
}
// sum over the four 16-lane rows (same column), result in every row: (r0 + r2) + (r1 + r3)
template <typename R> I2C_FN R w_rowsum(const Wave<R>& w, const R x) {
#ifdef I2C_HOST_SIM
  w.bar->wait();
  w.xch[w.l] = x;
  w.bar->wait();
  const int j = w.l & 15;
  return (w.xch[j] + w.xch[32 + j]) + (w.xch[16 + j] + w.xch[48 + j]);
#else
  int lo = __double2loint(x), hi = __double2hiint(x);
  const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double y = __hiloint2double(h32[0], l32[0]) + __hiloint2double(h32[1], l32[1]);
  lo = __double2loint(y), hi = __double2hiint(y);
  const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(h16[0], l16[0]) + __hiloint2double(h16[1], l16[1]);
#endif
}
// The 4 x 4 pivot block of row block KB of a symmetric matrix, to every lane: x = accumulator register KB, whose entry (a, b) of
// the block sits in lane (a, 4 KB + b). d = {d00, d10, d11, d20, d21, d22, d30, d31, d32, d33}. Two ways (PL), a compile-time
// choice per kernel instantiation:
//   v_readlane: ten pairs into scalar registers -- no round trip to wait for, 20 issue slots: the forward sweep while every
//               wave has a SIMD to itself (B <= 1024: 0.382 -> 0.365 ms);
//   LDS:        the 16 lanes that hold the block store it, every lane reads it back (1 + 6 instructions and a round trip that
//               other waves of the SIMD fill): the backward sweep, and the forward sweep of larger batches (the readlanes are
//               17 % of a forward cell's vector instructions, and a full chip is issue-bound).
// (A RUN-TIME switch between the two inside one kernel was measured and dropped: the extra branch and registers cost the
// forward sweep 6 - 10 % at every batch size.)
template <int KB, bool PL, typename R> I2C_FN void w_pivot_block(const Wave<R>& w, const R x, R* d) {
#ifdef I2C_HOST_SIM
  w.bar->wait();
  w.xch[w.l] = x;
  w.bar->wait();
  int n = 0;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b <= a; ++b) d[n++] = w.xch[16 * a + 4 * KB + b];
#else
  if constexpr (PL) {
    const auto dg = w.sh + WaveLds::O_DG;
    const bool inblk = (w.j >> 2) == KB;
    w.sync();
    dg[inblk ? w.q * 4 + (w.j & 3) : 16 + w.l] = x;  // the other lanes park their store
    w.sync();
    d[0] = dg[0], d[1] = dg[4], d[2] = dg[5], d[3] = dg[8], d[4] = dg[9], d[5] = dg[10];
    d[6] = dg[12], d[7] = dg[13], d[8] = dg[14], d[9] = dg[15];
  } else {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    int n = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b)
        d[n++] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * a + 4 * KB + b), __builtin_amdgcn_readlane(lo, 16 * a + 4 * KB + b));
  }
#endif
}
// column form (lane (q, j): x[j]) -> row form (x[q + 4 v], v < NV) through vector slot `slot`
template <int NV, typename R> I2C_FN void w_col2row(const Wave<R>& w, const int slot, const R xc, R* xr) {
  const auto s = w.vec(slot);
  w.sync();
#ifdef I2C_HOST_SIM
  if (w.q == 0)  // (lane-threads must not race; on the device the four rows store the same value to the same address)
#endif
    s[w.j] = xc;
  w.sync();
#pragma unroll
  for (int v = 0; v < NV; ++v) xr[v] = s[w.row(v)];
}
// sum over all 64 lanes, in a fixed order, result in every lane
template <typename R> I2C_FN R w_wavesum(const Wave<R>& w, const int slot, const R x) {
  const R cs = w_rowsum(w, x);
  const auto s = w.vec(slot);
  w.sync();
#ifdef I2C_HOST_SIM
  if (w.q == 0)
#endif
    s[w.j] = cs;
  w.sync();
  R t = R(0);
#pragma unroll
  for (int k = 0; k < 16; ++k) t += s[k];
  return t;
}

// Batch-wide constants by row, in LDS (one copy per workgroup): 16 x 16 row-major, zero-padded
template <class M, typename R> struct WConst {
  R xi[256], eta[256], xiT[256], qr[256], qf[256], sxT[256];  // sig_xi0, sig_eta, sig_xiT0, blkdiag(Q, R), Qf, sig_x_terminal
  R zg[16], zgT[16], mxT[16];
};
template <class M, typename R, class DST> I2C_FN void wconst_fill(DST& k, const Consts<M, R>* c, const int tid, const int nthreads) {
  constexpr int NX = M::NX, NZ = M::NZ, NT = M::NZT > 0 ? M::NZT : 1;
  for (int e = tid; e < 256; e += nthreads) {
    const int i = e >> 4, j = e & 15;
    k.xi[e] = (i < NZ && j < NZ) ? c->sig_xi0[tri_any(i, j)] : R(0);
    k.eta[e] = (i < NX && j < NX) ? c->sig_eta[tri_any(i, j)] : R(0);
    k.xiT[e] = (i < NT && j < NT) ? c->sig_xiT0[tri_any(i, j)] : R(0);
    k.qr[e] = (i < NZ && j < NZ) ? c->QR[tri_any(i, j)] : R(0);
    k.qf[e] = (i < NT && j < NT) ? c->Qf[tri_any(i, j)] : R(0);
    k.sxT[e] = (i < NX && j < NX) ? c->sig_x_term[tri_any(i, j)] : R(0);
  }
  for (int e = tid; e < 16; e += nthreads) {
    k.zg[e] = e < NZ ? c->zg[e] : R(0);
    k.zgT[e] = e < NT ? c->zg_term[e] : R(0);
    k.mxT[e] = e < NX ? c->mu_x_term[e] : R(0);
  }
}
// accumulator-layout registers of a constant block
template <int NV, typename R, class P> I2C_FN void w_ldconst(const Wave<R>& w, const P m, R* x) {
#pragma unroll
  for (int v = 0; v < NV; ++v) x[v] = m[w.row(v) * 16 + w.j];
}

// ---- blocked Cholesky elimination --------------------------------------------------------------------------------------------
// s: SPD matrix of dimension 4 NB (accumulator layout, consumed). On return lt = L^T (upper triangular, accumulator layout,
// rows >= 4 NB zero) and, for each of the NRHS right-hand sides r (4 NB x 16), r = L^-1 r. Returns whether every pivot was
// positive. Per 4 x 4 pivot block: the block goes to every lane (w_pivot_block), which factors it and inverts the factor in
// registers (four dependent rsq chains: the serial core of the kernel); lane (q, i) then holds entry (i mod 4, q) of that inverse
// as the A operand of the matrix instruction that scales block row kb of s and of every right-hand side, and the scaled block
// row of s (= rows of L^T) is A and B operand of the rank-4 update of everything below.
// SIGNED: the matrix is symmetric and non-singular but not necessarily positive definite; the factorisation is L Sigma L^T with
// Sigma = diag(+-1) (a Cholesky factorisation that carries the sign of each pivot), sgn receives Sigma in accumulator layout
// (register v, lane (q, .): the sign of row 4 v + q) and the right-hand sides return L^-1 r as before, so that
// r^T s^-1 r = (L^-1 r)^T Sigma (L^-1 r). Used once per sweep (the Linearize covariance-control multiplier); lt is not produced.
template <int KB, int NB, int NRHS, bool PL, bool SIGNED = false, typename R>
I2C_FN void w_elim_step(const Wave<R>& w, R* s, R* r1, R* r2, R* lt, const R* mq, R* last, R* sgn = nullptr) {
  const int a = w.j & 3, cq = w.q;
  const bool inblk = (w.j >> 2) == KB;
  R d[10];
  w_pivot_block<KB, PL>(w, s[KB], d);
  const R d00 = d[0], d10 = d[1], d11 = d[2], d20 = d[3], d21 = d[4], d22 = d[5], d30 = d[6], d31 = d[7], d32 = d[8], d33 = d[9];
  // 4 x 4 Cholesky (l), column by column. (Measured and dropped: two levels of 2 x 2 blocks, whose two rsq chains per level are
  // independent -- half the dependent depth for 5 more instructions: SLOWER at every batch size, B = 256: 0.333 -> 0.343 ms,
  // B = 8192: 2.07 -> 2.11 ms, profiles/r3_quad12_pivot_2x2_ab.txt. Even the lone wave is bound by instruction count here.)
  R i0, i1, i2, i3, l10, l20, l30, l21, l31, l32, p3, sq = R(1);
  if constexpr (!SIGNED) {
    i0 = r_rsqrt(d00);
    l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
    const R p1 = d11 - l10 * l10;
    i1 = r_rsqrt(p1);
    l21 = (d21 - l20 * l10) * i1, l31 = (d31 - l30 * l10) * i1;
    const R p2 = d22 - l20 * l20 - l21 * l21;
    i2 = r_rsqrt(p2);
    l32 = (d32 - l30 * l20 - l31 * l21) * i2;
    p3 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
    i3 = r_rsqrt(p3);
  } else {
    const R g0 = d00 < R(0) ? R(-1) : R(1);
    i0 = r_rsqrt(g0 * d00);
    l10 = d10 * i0 * g0, l20 = d20 * i0 * g0, l30 = d30 * i0 * g0;
    const R p1 = d11 - g0 * l10 * l10;
    const R g1 = p1 < R(0) ? R(-1) : R(1);
    i1 = r_rsqrt(g1 * p1);
    l21 = (d21 - g0 * l20 * l10) * i1 * g1, l31 = (d31 - g0 * l30 * l10) * i1 * g1;
    const R p2 = d22 - g0 * l20 * l20 - g1 * l21 * l21;
    const R g2 = p2 < R(0) ? R(-1) : R(1);
    i2 = r_rsqrt(g2 * p2);
    l32 = (d32 - g0 * l30 * l20 - g1 * l31 * l21) * i2 * g2;
    p3 = d33 - g0 * l30 * l30 - g1 * l31 * l31 - g2 * l32 * l32;
    const R g3 = p3 < R(0) ? R(-1) : R(1);
    i3 = r_rsqrt(g3 * p3);
    sq = (mq[0] * g0 + mq[1] * g1) + (mq[2] * g2 + mq[3] * g3);
    sgn[KB] = sq;
    p3 = p3 * p3;  // the caller tests `> 0`: a zero (or NaN) pivot fails, a negative one does not
  }
  if (KB == NB - 1) *last = p3;  // a failed pivot poisons everything after it (see chol(), i2c_linalg.hpp)
  // ... and this lane's entry of its inverse: row a of L^-1 solves y^T L = e_a^T (back substitution; y_c = 0 for c > a
  // falls out of the one-hot right-hand side, which is also zero outside block row KB), entry cq picked by a one-hot
  // combination. Written without selects on purpose: with selects hipcc sinks the arithmetic into per-(a, cq) divergent
  // branches, which a wavefront then executes one after the other.
  // (every column block of lanes holds the entries: the small matrix instruction multiplies its four 4 x 4 blocks by the same matrix)
  const R e0 = a == 0 ? R(1) : R(0), e1 = a == 1 ? R(1) : R(0);
  const R e2 = a == 2 ? R(1) : R(0), e3 = a == 3 ? R(1) : R(0);
  const R y3 = e3 * i3;
  const R y2 = (e2 - l32 * y3) * i2;
  const R y1 = (e1 - l21 * y2 - l31 * y3) * i1;
  const R y0 = (e0 - l10 * y1 - l20 * y2 - l30 * y3) * i0;
  const R aw = (mq[0] * y0 + mq[1] * y1) + (mq[2] * y2 + mq[3] * y3);
  // scale block row KB: rows of L^T (masked to the upper triangle: what is left of it is rounding noise) ...
  // (one row block each: the small matrix instruction, w_mfma4 -- lane (q, j) passes entry (j & 3, q) of the inverse, which is aw)
  const R xk = w_mfma4(w, aw, s[KB]);
  const R ltk = (w.j >= 4 * KB + cq) ? xk : R(0);
  lt[KB] = ltk;
  R x1 = R(0), x2 = R(0);
  if (NRHS >= 1) x1 = w_mfma4(w, aw, r1[KB]);
  if (NRHS >= 2) x2 = w_mfma4(w, aw, r2[KB]);
  // ... and eliminate it from everything below
  if (KB < NB - 1) {
    const R nl = SIGNED ? -ltk * sq : -ltk;
    w_mfma(w, nl, ltk, s);
    if (NRHS >= 1) w_mfma(w, nl, x1, r1);
    if (NRHS >= 2) w_mfma(w, nl, x2, r2);
  }
  if (NRHS >= 1) r1[KB] = x1;
  if (NRHS >= 2) r2[KB] = x2;
  if constexpr (KB + 1 < NB) w_elim_step<KB + 1, NB, NRHS, PL, SIGNED>(w, s, r1, r2, lt, mq, last, sgn);
}
// r <- L^-1 r for a symmetric non-singular s = L Sigma L^T (see w_elim_step); sgn <- Sigma, rows >= 4 NB get +1
template <int NB, bool PL = false, typename R> I2C_FN bool w_elim_signed(const Wave<R>& w, R* s, R* r, R* sgn) {
  R last = R(0), mq[4], lt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mq[k] = w.q == k ? R(1) : R(0);
    sgn[k] = R(1);
  }
  w_elim_step<0, NB, 1, PL, true>(w, s, r, (R*)nullptr, lt, mq, &last, sgn);
  return last > R(0);
}
template <int NB, int NRHS, bool PL = false, typename R> I2C_FN bool w_elim(const Wave<R>& w, R* s, R* r1, R* r2, R* lt) {
  R last = R(0), mq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) mq[k] = w.q == k ? R(1) : R(0);
#pragma unroll
  for (int v = 0; v < 4; ++v) lt[v] = R(0);
  w_elim_step<0, NB, NRHS, PL>(w, s, r1, r2, lt, mq, &last);
  return last > R(0);
}

// addressing of one [E][B] cell block for the lanes of a wave (storage type S, arithmetic type R)
template <typename R, typename S> struct WIO {
  Window w;
  unsigned rb, bo;
  I2C_MEM R ld(const int e) const { return (R)wld<S>(w, 0u, (unsigned)e * rb + bo); }
  I2C_MEM void st(const int e, const R v) const { wst(w, 0u, (unsigned)e * rb + bo, (S)v); }
  I2C_MEM void st_if(const bool on, const int e, const R v) const {
#ifdef I2C_HOST_SIM
    if (on) wst(w, 0u, (unsigned)e * rb + bo, (S)v);
#else
    wst(w, 0u, on ? (unsigned)e * rb + bo : 0x80000000u, (S)v);  // out of range: dropped by the buffer unit (see GIO::st_if)
#endif
  }
};
template <typename R, typename S> I2C_FN WIO<R, S> wio(const S* base, const unsigned long elems, const unsigned rb, const unsigned bo) {
  return WIO<R, S>{make_window(base, elems * rb), rb, bo};
}
// The forward-message buffer of the WAVE family is trajectory-major, [T][B][e_fwd]: the e_fwd elements of one cell of one
// trajectory are contiguous, so a wave's load of 64 elements touches 4 - 8 cache lines instead of 64 (with the common
// [T][e][B] layout every lane of a wave hits a different [B]-contiguous row: the vector-memory pipeline, ~0.5 line requests per
// clock and CU, bounded the backward sweep already at one wave per SIMD -- B = 1024: 0.184 ms). The buffer is private to the
// family that writes it (forward sweep) and reads it (backward sweep); callers go through i2c_kernel_family().
template <typename R, typename S> I2C_FN WIO<R, S> w_fwd_cell(const S* fwd, const int e_fwd, const unsigned long B, const int t, const int b) {
  return wio<R, S>(fwd + ((unsigned long)t * B + (unsigned long)b) * (unsigned long)e_fwd, (unsigned long)e_fwd, (unsigned)sizeof(S), 0u);
}
// one cell of the posterior / prior buffer for trajectory b: [T][E][B], or trajectory-major [T][B][E] (Consts::post_tm)
template <typename R, typename S>
I2C_FN WIO<R, S> w_post_cell(const S* post, const int e_post, const unsigned long B, const int row, const int b, const bool tm) {
  const S* cell = post + (unsigned long)row * (unsigned long)e_post * B;
  if (tm) return wio<R, S>(cell + (unsigned long)b * (unsigned long)e_post, (unsigned long)e_post, (unsigned)sizeof(S), 0u);
  return wio<R, S>(cell, (unsigned long)e_post, (unsigned)(B * sizeof(S)), (unsigned)(b * sizeof(S)));
}
I2C_FN int w_symidx(const int i, const int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// Kalman-style update of N(mu, s) of dimension N = 4 NB on an IDENTITY observation of it with noise alpha * xi and target zt
// (i2c.py:394-403 with z a leading slice of the state):  with C = chol(s + alpha xi), U = C^-1 s:  s <- s - U^T U;  the mean uses
// the posterior-covariance form of the same gain,  s (s + N)^-1 = s_new N^-1  (N^-1 = W / alpha, W = the cost weight):
//   mu <- mu + s_new W (zt - mu) / alpha.
// mu, zt, wd (diagonal of W) in column form; xi, wm (W when not diagonal) accumulator-layout constants.
template <int NB, bool PL = false, typename R, class P> I2C_FN bool w_kalman(const Wave<R>& w, const R alpha, const P xi_m, const P w_m, const bool w_diag,
                                                             const R zt, R* mu, R* s) {
  R sz[4], u[4], lt[4], xi[4];
  w_ldconst<NB>(w, xi_m, xi);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    sz[v] = v < NB ? s[v] + alpha * xi[v] : R(0);
    u[v] = v < NB ? s[v] : R(0);
  }
  const bool ok = w_elim<NB, 1, PL>(w, sz, u, (R*)nullptr, lt);
  w_tn<NB, true>(w, u, u, s);
  const R r = zt - *mu;
  R wr[4];
  if (w_diag) {
    const R wd = w_m[w.j * 16 + w.j];
    w_col2row<NB>(w, 0, wd * r, wr);
  } else {
    R rr[4], wm[4];
    w_col2row<NB>(w, 0, r, rr);
    w_ldconst<NB>(w, w_m, wm);
    R t = R(0);
#pragma unroll
    for (int v = 0; v < NB; ++v) t += wm[v] * rr[v];
    w_col2row<NB>(w, 1, w_rowsum(w, t), wr);
  }
  R t = R(0);
#pragma unroll
  for (int v = 0; v < NB; ++v) t += s[v] * wr[v];
  *mu += w_rowsum(w, t) * r_rcp(alpha);
  return ok;
}

// The same update in square-root form, for a prior that arrives as its Cholesky factor (round 5): with s = L L^T and
// N^-1 = W / alpha,  s_new = L (I + L^T N^-1 L)^-1 L^T.  Factor M = I + L^T N^-1 L from its LAST row and column upwards,
// M = U U^T with U upper triangular: then  s_new = (L U^-T)(L U^-T)^T  and L U^-T is lower triangular with a positive diagonal --
// it IS chol(s_new), the factor the cubature rule of the dynamics needs (quadrature.py:17-24), with no second factorisation of a
// 16 x 16 matrix and no difference of two covariances. The reversed elimination is the ordinary one on the index-reversed matrix:
// with J the exchange matrix, J M J = I + X^T N^-1 X (X = L J, read column-reversed from the LDS copy of L^T) = L' L'^T, and
// Y = L'^-1 (J L^T) -- the right-hand side of that elimination, L^T read row-reversed -- is  J chol(s_new)^T: the rows of
// chol(s_new)^T in reverse order. A set of sigma-point directions has no order, and s_new = Y^T Y.
// l: in L^T (accumulator layout), out Y;  s: out s_new;  mu, zt, W as in w_kalman. M >= I: the elimination fails only on
// non-finite input.
template <bool PL = false, typename R, class P> I2C_FN bool w_kalman_sqrt(const Wave<R>& w, const R alpha, const P w_m, const bool w_diag, const R zt,
                                                                   R* mu, R* l, R* s) {
  const auto Lt = w.mat();
  const R ra = r_rcp(alpha);
  w.sync();
#pragma unroll
  for (int v = 0; v < 4; ++v) Lt[w.row(v) * WLD + w.j] = l[v];
  w.sync();
  R x[4], y[4], m[4], lt[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    x[v] = Lt[(15 - w.j) * WLD + w.row(v)];  // (L J)[row][j]
    l[v] = Lt[(15 - w.row(v)) * WLD + w.j];  // (J L^T)[row][j]
  }
  if (w_diag) {
#pragma unroll
    for (int v = 0; v < 4; ++v) y[v] = (w_m[w.row(v) * 17] * ra) * x[v];
  } else {
    R wm[4];
    w_ldconst<4>(w, w_m, wm);
#pragma unroll
    for (int v = 0; v < 4; ++v) y[v] = R(0);
    w_tn<4>(w, wm, x, y);  // W symmetric
#pragma unroll
    for (int v = 0; v < 4; ++v) y[v] *= ra;
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) m[v] = w.row(v) == w.j ? R(1) : R(0);
  w_tn<4>(w, x, y, m);
  const bool ok = w_elim<4, 1, PL>(w, m, l, (R*)nullptr, lt);
#pragma unroll
  for (int v = 0; v < 4; ++v) s[v] = R(0);
  w_tn<4>(w, l, l, s);
  const R r = zt - *mu;
  R wr[4];
  if (w_diag) {
    const R wd = w_m[w.j * 16 + w.j];
    w_col2row<4>(w, 0, wd * r, wr);
  } else {
    R rr[4], wm[4];
    w_col2row<4>(w, 0, r, rr);
    w_ldconst<4>(w, w_m, wm);
    R t = R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) t += wm[v] * rr[v];
    w_col2row<4>(w, 1, w_rowsum(w, t), wr);
  }
  R t = R(0);
#pragma unroll
  for (int v = 0; v < 4; ++v) t += s[v] * wr[v];
  *mu += w_rowsum(w, t) * ra;
  return ok;
}

// ------------------------------------------------------------------------------------------
// Forward sweep (i2c.py:876-880 over :350-447)
// ------------------------------------------------------------------------------------------
// LIN: Linearize() inference (i2c.py:244-348): the same cell with the dynamics push-through replaced by value + Jacobian (one
// forward-mode pass per input direction, lane p carrying the tangent e_p), the pdf-ratio scaling of the gain only with the
// expert controller (:259-265), and no terminal update here (it happens at the end of the backward chain, :475-491).
// PL: the pivot blocks of the eliminations go through LDS instead of v_readlane (w_pivot_block): the variant for batches
// whose waves share a SIMD.
template <class M, typename R, typename S, bool LIN, bool PL, class KC>
I2C_HD inline void forward_wave_body(const Consts<M, R>& c, const KC& kc, const FwdArgs<R, S>& a, const int b, const Wave<R>& w) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D, NBX = NX / 4;
  static_assert(D == 16 && NX % 4 == 0 && NZ == D && (NZT == NX || NZT == 0), "wave kernels: d = 16, identity observations");
  static_assert(st_identity<ObsStruct<M>, NZ>() && st_identity<TermStruct<M>, NZT>(), "wave kernels: identity observations");
  constexpr int O_K = D + sym(D), O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  const int q = w.q, j = w.j;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned WS = sizeof(S), bo = (unsigned)b * WS, rb = (unsigned)(B * WS);
  const bool jx = j < NX;
  const int jxc = jx ? j : 0;
  const R alpha_traj = a.alpha[b];
  const Rule<R>& rule = c.rule_xu;
  int fail = 0;

  // state message carried along the chain: mean in column form, covariance in the accumulator layout (nx x nx, zero-padded)
  R mx = jx ? a.x0[(long)jxc * B + b] : R(0);
  R sx[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) sx[v] = (v < NBX && jx) ? a.sig_x0[(long)w_symidx(w.row(v) < NX ? w.row(v) : 0, jxc) * B + b] : R(0);

  // The prior rows of a cell do not depend on the recursion: they are fetched ONE CELL AHEAD (issued once the current cell's
  // have been consumed), so that the memory round trip hides behind a cell's worth of work.
  R nx_pmu, nx_pj[4], nx_kt[4], nx_alpha, nx_zt;
  int nx_ff, nx_ex;
  const Window ffw = make_window(a.ff, (unsigned long)T), exw = make_window(a.expert ? a.expert : a.ff, (unsigned long)T);
  const Window alw = make_window(a.alpha_cell ? a.alpha_cell : a.alpha, (a.alpha_cell ? (unsigned long)T : 1ul) * B * sizeof(R));
  const Window zw = make_window(c.z_per_cell ? a.z : a.x0, (c.z_per_cell ? (unsigned long)T * NZ : 1ul) * B * sizeof(R));  // (x0 row 0: a valid dummy)
  auto fetch_prior = [&](const int tc) {
    const int trc = c.row(tc);
    const WIO<R, S> pri = w_post_cell<R, S>(a.prior, C::E_POST, B, trc, b, c.post_tm != 0);
    nx_pmu = pri.ld(j);
#pragma unroll
    for (int v = 0; v < 4; ++v) nx_pj[v] = pri.ld(D + w_symidx(w.row(v), j));
#pragma unroll
    for (int v = 0; v < 4; ++v) nx_kt[v] = v < NBX ? pri.ld(O_K + (jx ? 0 : j - NX) * NX + w.row(v)) : R(0);  // K^T columns (action lanes)
    // (buffer loads without branches: a flat load -- global or LDS, picked at run time -- or a global load behind a branch
    //  among buffer stores makes the waitcnt pass wait for vmcnt(0), i.e. for the acknowledgement of the cell's own stores)
    nx_alpha = wld<R>(alw, 0u, a.alpha_cell ? (unsigned)(((unsigned long)trc * B + b) * sizeof(R)) : (unsigned)(b * sizeof(R)));
    // (the per-cell target or a discarded dummy; the choice is made when the value is USED, a cell later: selecting it here would
    //  wait for the load right after issuing it)
    nx_zt = wld<R>(zw, 0u, c.z_per_cell ? (unsigned)((((unsigned long)trc * NZ + j) * B + b) * sizeof(R)) : (unsigned)(b * sizeof(R)));
    nx_ff = (int)wld_u8(ffw, (unsigned)trc);
    nx_ex = LIN ? (int)wld_u8(exw, (unsigned)trc) : 1;  // (Linearize: the per-cell flag or a discarded dummy, picked where it is used)
  };
  fetch_prior(0);
  // settled before the loop: with loads pending on the loop-entry path the waitcnt pass merges their queue positions with the
  // back edge's (where a cell's stores are younger than its prefetch) and waits for vmcnt(0) -- the acknowledgement of the previous
  // cell's stores -- at the top of EVERY cell (see forward_sweep_body)
  nx_pmu = opaque(nx_pmu), nx_alpha = opaque(nx_alpha), nx_zt = opaque(nx_zt);
  nx_ff = (int)opaque((unsigned)nx_ff), nx_ex = (int)opaque((unsigned)nx_ex);
#pragma unroll
  for (int v = 0; v < 4; ++v) nx_pj[v] = opaque(nx_pj[v]), nx_kt[v] = opaque(nx_kt[v]);
  mx = opaque(mx);
#pragma unroll
  for (int v = 0; v < 4; ++v) sx[v] = opaque(sx[v]);

  // Sigma-point cells carry the FACTOR of the state message next to it (round 5): lx = chol(sig_x)^T and, in its action columns,
  // lk = chol(sig_x)^T K^T for the controller of the coming cell (whose rows are fetched a cell ahead). The joint prior of a cell
  // is then known by its factor,  chol(sig_0)^T = [lx | rho lk; 0 | chol(P_uu - rho K P_xu)^T]  -- one 4 x 4 pivot --, and the cost
  // observation updates that factor directly (w_kalman_sqrt): 11 pivot blocks per cell instead of 14, nothing re-factored. Both
  // come out of the elimination of sig_x3 that the smoother gain needs anyway: lk rides in the free action columns of its
  // identity right-hand side.
  R lx[4] = {R(0), R(0), R(0), R(0)}, lk[4] = {R(0), R(0), R(0), R(0)};
  bool lx_ok = true;
  auto state_gain_rhs = [&](R* r) {  // [I | sig_x K^T] with the rows of the coming cell
    R fn[4], mk[4] = {R(0), R(0), R(0), R(0)};
#pragma unroll
    for (int v = 0; v < 4; ++v) fn[v] = v < NBX ? (jx ? (w.row(v) == j ? R(1) : R(0)) : nx_kt[v]) : R(0);
    w_tn<NBX>(w, sx, fn, mk);
#pragma unroll
    for (int v = 0; v < 4; ++v) r[v] = v < NBX ? (jx ? fn[v] : mk[v]) : R(0);
  };
  auto refactor_state = [&]() {  // the chain's entry and the cell after a terminal update: the factor on its own
    R tmp[4], r[4], l3[4];
    state_gain_rhs(r);
#pragma unroll
    for (int v = 0; v < 4; ++v) tmp[v] = sx[v];
    lx_ok = w_elim<NBX, 1, PL>(w, tmp, r, (R*)nullptr, l3);
#pragma unroll
    for (int v = 0; v < 4; ++v) lx[v] = l3[v], lk[v] = r[v];
  };
  if constexpr (!LIN) refactor_state();

#if defined(I2C_WAVE_STAMPS) && !defined(I2C_HOST_SIM)
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < T; ++t) {
    const WIO<R, S> out = w_fwd_cell<R, S>(a.fwd, C::E_FWD, B, t, b);
    const R alpha = nx_alpha, zt = c.z_per_cell ? nx_zt : kc.zg[j], pmu = nx_pmu;
    // the sigma-point cell always scales the gain (i2c.py:366-375); Linearize: per-cell or graph-wide flag (i2c.py:259-265)
    const bool ff = w_uniform(nx_ff) != 0, scale_gain = LIN ? (a.expert ? w_uniform(nx_ex) != 0 : c.use_expert != 0) : true;
    R pj[4], kt[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      pj[v] = nx_pj[v];
      kt[v] = nx_kt[v];
    }
    int cell_bad = 0;

    // ---- 1. joint prior over (x, u) ---------------------------------------------------
    R mu0, s0[4];
    R l0[4];  // sigma-point cells: chol(sig_0)^T, then the sigma-point directions of the updated joint
    if (ff) {  // feed-forward: independent action prior (i2c.py:355-360)
      mu0 = jx ? mx : pmu;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if constexpr (LIN) s0[v] = (v < NBX) ? (jx ? sx[v] : R(0)) : (jx ? R(0) : pj[v]);
        else l0[v] = (v < NBX) ? (jx ? lx[v] : R(0)) : (jx ? R(0) : pj[v]);  // (action block: factored below)
      }
    } else {  // feedback: condition the previous controller on the new state message (i2c.py:361-387)
      // pdf ratio rho = exp(-delta^T (P_xx + sig_x)^-1 delta / 2): delta rides as column NX of the nx x nx matrix ITSELF (its tile
      // has 16 columns): the scaling and the rank-4 updates of the elimination treat that column like any other, so
      // y = L^-1 delta comes out as column NX of L^T with no right-hand-side instructions at all
      const R dl = jx ? mx - pmu : R(0);
      R dr[4];
      w_col2row<NBX>(w, 0, dl, dr);
      R rho = R(1);
      if (scale_gain) {
        R sm[4], lt[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) sm[v] = v < NBX ? (jx ? pj[v] + sx[v] : (j == NX ? dr[v] : R(0))) : R(0);
        cell_bad = flag_stage(cell_bad, w_elim<NBX, 0, PL>(w, sm, (R*)nullptr, (R*)nullptr, lt), 0);
        R ysq = R(0);
#pragma unroll
        for (int v = 0; v < NBX; ++v) ysq += j == NX ? lt[v] * lt[v] : R(0);
        const R maha = w_bcast<NX>(w, w_rowsum(w, ysq));
        rho = r_exp(R(-0.5) * maha);
      }
      // F^T = [I | Kt^T] (nx x 16), Kt = rho K:  sig_0 = F sig_x F^T with (P_uu - Kt P_xu) added to the action block
      R ft[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) ft[v] = v < NBX ? (jx ? (w.row(v) == j ? R(1) : R(0)) : rho * kt[v]) : R(0);
      R kd = R(0);
#pragma unroll
      for (int v = 0; v < NBX; ++v) kd += ft[v] * dr[v];
      kd = w_rowsum(w, kd);  // action lanes: Kt delta
      mu0 = jx ? mx : pmu + kd;
      if constexpr (LIN) {
        R m1[4] = {R(0), R(0), R(0), R(0)}, m1p[4];
        w_tn<NBX>(w, sx, ft, m1);  // sig_x F^T
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          m1p[v] = (v < NBX && !jx) ? m1[v] - pj[v] : m1[v];
          s0[v] = R(0);
        }
        w_tn<NBX>(w, ft, m1p, s0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (v < NBX) s0[v] = jx ? s0[v] : m1[v];  // state-action block: sig_x Kt^T itself
          else s0[v] = jx ? s0[v] : s0[v] + pj[v];  // action block: P_uu - Kt P_xu + Kt sig_x Kt^T
        }
      } else {
        // the Schur complement of the state block, P_uu - Kt P_xu (the action block of sig_0 without Kt sig_x Kt^T)
        R pm[4], su[4] = {R(0), R(0), R(0), R(0)};
#pragma unroll
        for (int v = 0; v < 4; ++v) pm[v] = (v < NBX && !jx) ? pj[v] : R(0);
        w_tn<NBX>(w, ft, pm, su);
#pragma unroll
        for (int v = 0; v < 4; ++v) l0[v] = (v < NBX) ? (jx ? lx[v] : rho * lk[v]) : (jx ? R(0) : pj[v] - su[v]);
      }
    }
    I2C_WSTAMP(0);  // pdf ratio + prior mean / factor rows
    if constexpr (!LIN) {  // chol of the action block: the trailing pivot(s) of the joint's factorisation, on their own
      R sb[4], lu[4] = {R(0), R(0), R(0), R(0)}, mq[4], last = R(0);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        sb[v] = v < NBX ? R(0) : l0[v];
        mq[v] = w.q == v ? R(1) : R(0);
      }
      w_elim_step<NBX, 4, 0, PL>(w, sb, (R*)nullptr, (R*)nullptr, lu, mq, &last);
#pragma unroll
      for (int v = NBX; v < 4; ++v) l0[v] = lu[v];
      cell_bad = flag_stage(cell_bad, lx_ok && last > R(0), 1);
    }
    I2C_WSTAMP(1);  // action-block pivot
    fetch_prior(t + 1 < T ? t + 1 : t);  // this cell's rows are consumed: the next cell's, a cell ahead
    if (a.prior_out) {
      const WIO<R, S> po = wio<R, S>(a.prior_out + (unsigned long)t * (D + sym(D)) * B, D + sym(D), rb, bo);
      po.st_if(q == 0, j, mu0);
      if constexpr (!LIN) {
#pragma unroll
        for (int v = 0; v < 4; ++v) s0[v] = R(0);
        w_tn<4>(w, l0, l0, s0);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) po.st_if(w.row(v) >= j, D + w_symidx(w.row(v), j), s0[v]);
    }

    // ---- 2. cost "observation" z = (x, u): measurement update (i2c.py:390-407) ----------
    if constexpr (LIN) cell_bad = flag_stage(cell_bad, w_kalman<4, PL>(w, alpha, kc.xi, kc.qr, c.qr_diag != 0, zt, &mu0, s0), 2);
    else cell_bad = flag_stage(cell_bad, w_kalman_sqrt<PL>(w, alpha, kc.qr, c.qr_diag != 0, zt, &mu0, l0, s0), 2);
    I2C_WSTAMP(2);  // cost observation update
    out.st_if(q == 0, j, mu0);
#pragma unroll
    for (int v = 0; v < 4; ++v) out.st_if(w.row(v) >= j, D + w_symidx(w.row(v), j), s0[v]);

    // ---- 3. dynamics push-through (i2c.py:415-421; Linearize: :321-341) ----------------------
    R sxy[4] = {R(0), R(0), R(0), R(0)};  // sig_xy^T (nx x 16)
    if constexpr (!LIN) {
      const R* lt = l0;  // rows of chol(sig_xu1_f)^T (in reverse order): the factor came out of the update itself
      const auto Lt = w.mat();
      const auto mv = w.vec(2);
      w.sync();
#pragma unroll
      for (int v = 0; v < 4; ++v) Lt[w.row(v) * WLD + j] = lt[v];  // row p of L^T = direction of sigma-point pair p
#ifdef I2C_HOST_SIM
      if (q == 0)
#endif
        mv[j] = mu0;
      w.sync();
      {
        // lane p: m + sf L[:, p]; lane 16 + p: m - sf L[:, p]; lanes >= 32: the centre (lane 32's copy is used)
        const int p = w.l & 15;
        const R sg = w.l < 16 ? rule.sf : (w.l < 32 ? -rule.sf : R(0));
        R x[D], sn[M::NA > 0 ? M::NA : 1], cs[M::NA > 0 ? M::NA : 1], y[NX];
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = mv[i] + sg * Lt[p * WLD + i];
#pragma unroll
        for (int k = 0; k < M::NA; ++k) r_sincos(x[M::ang(k)], &sn[k], &cs[k]);
        M::dynamics(c.params, x, sn, cs, y);
        const auto Y = w.ybuf();
        if (w.l <= 32) {  // lanes beyond the centre computed a copy of it
#pragma unroll
          for (int k = 0; k < NX; ++k) Y[w.l * WaveLds::YLD + k] = y[k];
        }
      }
      w.sync();
      I2C_WSTAMP(3);  // stores + dynamics at the sigma points
      // a_p = (y+ - y0) + (y- - y0), d_p = y+ - y-, in the accumulator layout (row = point pair, column = output)
      R am[4], dm[4], y0;
      {
        const auto Y = w.ybuf();
        y0 = opaque(Y[32 * WaveLds::YLD + jxc]);  // (opaque: an unconditional read, not an EXEC-masked one)
        y0 = jx ? y0 : R(0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const R yp = Y[w.row(v) * WaveLds::YLD + jxc], ym = Y[(16 + w.row(v)) * WaveLds::YLD + jxc];
          am[v] = jx ? (yp - y0) + (ym - y0) : R(0);
          dm[v] = jx ? yp - ym : R(0);
        }
      }
      const R asum = w_rowsum(w, (am[0] + am[1]) + (am[2] + am[3]));
      mx = y0 + rule.wi * asum;  // mu_x3_f (column form)
      // sig_y = wi/2 sum_p (a_p a_p^T + d_p d_p^T) - wi^2 A A^T: with 2 d wi = 1 the last term centres the a_p
      R at[4], sy[4] = {R(0), R(0), R(0), R(0)};
      const R amean = (R(2) * rule.wi) * asum;
#pragma unroll
      for (int v = 0; v < 4; ++v) at[v] = jx ? am[v] - amean : R(0);
      w_tn<4>(w, at, at, sy);
      w_tn<4>(w, dm, dm, sy);
      w_tn<4>(w, dm, lt, sxy);  // sig_xy^T = wi sf [d_p]^T L^T  (nx x 16)
      {
        R eta[4];
        w_ldconst<NBX>(w, kc.eta, eta);
        const R hw = R(0.5) * rule.wi, cw = rule.wi * rule.sf;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          sx[v] = v < NBX ? hw * sy[v] + eta[v] : R(0);
          sxy[v] = v < NBX ? cw * sxy[v] : R(0);
        }
      }
    } else {
      // value and Jacobian A = d f / d [x; u] at the updated mean: lane p differentiates along input p (single-tangent dual
      // numbers through the SAME model functor), A^T lands in the accumulator layout (row = input, column = output)
      const auto mv = w.vec(2);
      w.sync();
#ifdef I2C_HOST_SIM
      if (q == 0)
#endif
        mv[j] = mu0;
      w.sync();
      {
        constexpr int NA1 = M::NA > 0 ? M::NA : 1, NP1 = M::NP > 0 ? M::NP : 1;
        const int p = w.l & 15;
        Dual<R> pd[NP1], x[D], sn[NA1], cs[NA1], y[NX];
#pragma unroll
        for (int i = 0; i < M::NP; ++i) pd[i] = Dual<R>(c.params[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = Dual<R>(mv[i], i == p ? R(1) : R(0));
#pragma unroll
        for (int k = 0; k < M::NA; ++k) {
          R s0v, c0v;
          r_sincos(x[M::ang(k)].v, &s0v, &c0v);
          const R seed = M::ang(k) == p ? R(1) : R(0);
          sn[k] = Dual<R>(s0v, seed * c0v);
          cs[k] = Dual<R>(c0v, -seed * s0v);
        }
        M::dynamics(pd, x, sn, cs, y);
        const auto Y = w.ybuf();
        if (w.l <= 16) {  // lanes 0..15: their column of A; lane 16 (a copy of lane 0's evaluation): the value f(mu)
          const int yrow = w.l < 16 ? w.l : 32;
#pragma unroll
          for (int k = 0; k < NX; ++k) Y[yrow * WaveLds::YLD + k] = w.l < 16 ? y[k].d : y[k].v;
        }
      }
      w.sync();
      R at[4], pm[4] = {R(0), R(0), R(0), R(0)}, sy[4] = {R(0), R(0), R(0), R(0)};
      {
        const auto Y = w.ybuf();
        const R y0 = opaque(Y[32 * WaveLds::YLD + jxc]);
        mx = jx ? y0 : R(0);  // mu_x3_f = f(mu_xu1_f)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const R av = opaque(Y[w.row(v) * WaveLds::YLD + jxc]);
          at[v] = jx ? av : R(0);
        }
      }
      w_tn<4>(w, at, s0, sxy);  // A sig_1   = sig_xy^T
      w_tn<4>(w, s0, at, pm);   // sig_1 A^T = sig_xy
      w_tn<4>(w, at, pm, sy);   // A sig_1 A^T
      R eta[4];
      w_ldconst<NBX>(w, kc.eta, eta);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        sx[v] = v < NBX ? sy[v] + eta[v] : R(0);
        sxy[v] = v < NBX ? sxy[v] : R(0);
      }
    }
    I2C_WSTAMP(4);  // dynamics moments
    // ---- smoother gain J = sig_xy sig_x3^-1 (i2c.py:423-425): J^T = W^T (W sig_xy^T), W = chol(sig_x3)^-1 ----
    {
      R tmp[4], w3[4], l3[4], jt[4] = {R(0), R(0), R(0), R(0)};
      if constexpr (LIN) {
#pragma unroll
        for (int v = 0; v < 4; ++v) w3[v] = (v < NBX && w.row(v) == j) ? R(1) : R(0);
      } else {
        state_gain_rhs(w3);  // identity | sig_x3 K^T of the next cell (its rows are in flight since the top of this one)
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) tmp[v] = sx[v];
      const bool ok3 = w_elim<NBX, 2, PL>(w, tmp, sxy, w3, l3);
      cell_bad = flag_stage(cell_bad, ok3, 4);
      if constexpr (!LIN) {
        lx_ok = ok3;
#pragma unroll
        for (int v = 0; v < 4; ++v) lx[v] = l3[v], lk[v] = w3[v];
      }
      w_tn<NBX>(w, w3, sxy, jt);  // (rows >= nx of the product, from the action columns of w3, are not part of J)
#pragma unroll
      for (int v = 0; v < NBX; ++v) out.st(O_J + j * NX + w.row(v), jt[v]);
    }
    // ---- 4. terminal cost observation on the flagged cell, after J (i2c.py:430-443) ----
    if (!LIN && NZT > 0 && t == c.terminal_cell && c.has_Qf) {  // (uniform: kernel arguments)
      cell_bad = flag_stage(cell_bad, w_kalman<NBX, PL>(w, alpha, kc.xiT, kc.qf, c.qf_diag != 0, kc.zgT[j], &mx, sx), 5);
      if constexpr (!LIN) {
        if (t + 1 < T) refactor_state();  // the state message changed after its factorisation
      }
    }
    fail = fold_cell_failure(fail, cell_bad, t);
    out.st_if(q == 0 && jx, O_MU3 + jxc, mx);
#pragma unroll
    for (int v = 0; v < NBX; ++v) out.st_if(jx && w.row(v) >= j, O_S3 + w_symidx(w.row(v), jxc), sx[v]);
    I2C_WSTAMP(5);  // smoother gain + factor of the state message, terminal update, stores
  }
#if defined(I2C_WAVE_STAMPS) && !defined(I2C_HOST_SIM)
  if (b == 0 && w.l == 0)
    printf("wave forward, clocks per cell: pdf ratio + prior %llu | action pivot %llu | cost update %llu | stores + dynamics points %llu | moments %llu | gain + state factor %llu\n",
           stamp_acc[0] / T, stamp_acc[1] / T, stamp_acc[2] / T, stamp_acc[3] / T, stamp_acc[4] / T, stamp_acc[5] / T);
#endif
  if (w.l == 0 && fail != 0 && a.status[b] == 0) a.status[b] = fail;
}

// ------------------------------------------------------------------------------------------
// Backward sweep (i2c.py:882-886 over :544-610). Two schedules of the same cell arithmetic:
//   fused    -- one wave walks T-1..0 doing the whole cell (one launch; the choice once the chip is full);
//   two-pass -- only the nx x nx marginal recursion is sequential (backward_wave_scan_body: six matrix instructions per
//               cell), everything else of a cell -- joint update, expected cost, controller -- is independent given the
//               smoothed next state, so a second launch runs ONE WAVE PER (t, b) CELL (backward_wave_cell_body): T times the
//               parallelism of the walk, which is what a batch of ~1000 trajectories needs to fill the chip; the per-cell
//               cost terms are summed over t by the lane kernels' k_reduce in a fixed order.
// LIN: Linearize() inference (i2c.py:449-542): at the end of the chain a terminal state prior PINS the smoothed terminal state
// (:453-472), otherwise the terminal cost observation is applied there (:475-491, with sig_xi_terminal kept in sig_z3_m); per
// cell the alpha statistic uses the linearised marginal observation WITHOUT state-action cross terms (:537-540) while the plan
// cost is priced with the cubature transform of the full joint (i2c.py:841-844, 1034-1053). One schedule (fused).
// term_stats rows: 0 = terminal trace, 1 = sum_t alpha statistic, 2 = sum_t cost variance, last = sum_t plan-cost mean (LIN).
// ------------------------------------------------------------------------------------------
// cost of N(m, s) about a target under weight W (16 x 16 row-major in LDS; diag: only its diagonal): this lane's share of
//   mean  err^T W err + tr(s W)   and   variance  2 tr((s W)^2) + 4 err^T W s W err      (i2c.py:1034-1043)
template <typename R, class P>
I2C_FN void w_cost_share(const Wave<R>& w, const bool diag, const P wmat, const int NB, const R err, const R* s, R* pm, R* pv) {
  const int j = w.j;
  R er[4], wm[4];
  w_col2row<4>(w, 3, err, er);
  w_ldconst<4>(w, wmat, wm);
  if (diag) {
    const R wd = wmat[j * 16 + j];
    R m = R(0), t2 = R(0), qd = R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v < NB) {
        const R wrv = wmat[w.row(v) * 16 + w.row(v)];
        m += (w.row(v) == j) ? wd * (err * err + s[v]) : R(0);
        t2 += s[v] * s[v] * (wrv * wd);
        qd += (wrv * er[v]) * s[v] * (wd * err);
      }
    }
    *pm = m;
    *pv = R(2) * t2 + R(4) * qd;
  } else {
    // P = s W, P^T = W s (matrix instructions); W err in column and row form
    R p[4] = {R(0), R(0), R(0), R(0)}, pt[4] = {R(0), R(0), R(0), R(0)};
    w_tn<4>(w, s, wm, p);
    w_tn<4>(w, wm, s, pt);
    R t = R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) t += wm[v] * er[v];
    const R we = w_rowsum(w, t);
    R wer[4];
    w_col2row<4>(w, 4, we, wer);
    R m = R(0), t2 = R(0), qd = R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v < NB) {
        m += er[v] * wm[v] * err + s[v] * wm[v];
        t2 += p[v] * pt[v];
        qd += wer[v] * s[v] * we;
      }
    }
    *pm = m;
    *pv = R(2) * t2 + R(4) * qd;
  }
}

// End of the chain (i2c.py:546-572): the smoothed terminal state (m3m column form, s3m accumulator layout) and the terminal
// observation statistics (term_stats rows 0, 3..).
template <class M, typename R, typename S, bool LIN, class KC>
I2C_FN void w_end_of_chain(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a, const int b, const Wave<R>& w, R* m3m_out, R* s3m) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZT = C::NZT, D = C::D, NBX = NX / 4, NT = C::NZT1;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX;
  const int q = w.q, j = w.j;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned WS = sizeof(S), bo = (unsigned)b * WS, rb = (unsigned)(B * WS);
  const bool jx = j < NX;
  const int jxc = jx ? j : 0;
  R m3m;
  {  // the smoothed terminal state is the filtered one (i2c.py:563-564)
    const WIO<R, S> fw = w_fwd_cell<R, S>(a.fwd, C::E_FWD, B, T - 1, b);
    const R mv = fw.ld(O_MU3 + jxc);
    m3m = jx ? mv : R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const R sv = v < NBX ? fw.ld(O_S3 + w_symidx(w.row(v) < NX ? w.row(v) : 0, jxc)) : R(0);
      s3m[v] = (v < NBX && jx) ? sv : R(0);
    }
  }
  if (!LIN && c.has_x_terminal) {
    // covariance control (i2c.py:548-559): the smoothed terminal state is the product of the TEMPERED filtered state
    // N(m3f, temp S3f) with the terminal prior N(mu_T, S_T) -- in Kalman form, an identity observation of the state with noise
    // S_T and target mu_T:  with C = chol(S_T + St), [U | y] = C^-1 [St | mu_T - m3f]:  S3m = St - U^T U,  m3m = m3f + U^T y
    // (the reference: S3m (St^-1 m3f + S_T^-1 mu_T) with three inverses: the same Gaussian). temp += dtemp per sweep.
    const R tmp = a.temp[b];
    w.sync();  // (every lane has read the temperature before lane 0 advances it)
    if (w.l == 0) a.temp[b] = tmp + c.dtemp;
    R sxT[4], st[4], sum[4], u[4], r2[4], lt[4], dr[4];
    w_ldconst<NBX>(w, kc.sxT, sxT);
    w_col2row<NBX>(w, 0, jx ? kc.mxT[jxc] - m3m : R(0), dr);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      st[v] = (v < NBX && jx) ? tmp * s3m[v] : R(0);
      sum[v] = (v < NBX && jx) ? sxT[v] + st[v] : R(0);
      u[v] = st[v];
      r2[v] = (v < NBX && j == 0) ? dr[v] : R(0);  // the innovation as column 0 of a second right-hand side
    }
    if (!w_elim<NBX, 2, true>(w, sum, u, r2, lt) && w.l == 0) set_status(a.status, b, 6, T - 1);
    R inc[4] = {R(0), R(0), R(0), R(0)};
    w_tn<NBX, true>(w, u, u, st);
    w_tn<NBX>(w, u, r2, inc);  // column 0: U^T y, element q + 4 v in register v of lane (q, 0)
    {
      const auto sl = w.vec(4);
      w.sync();
#pragma unroll
      for (int v = 0; v < NBX; ++v)
        if (j == 0) sl[w.row(v)] = inc[v];
      w.sync();
      m3m = jx ? m3m + sl[jxc] : R(0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) s3m[v] = (v < NBX && jx) ? st[v] : R(0);
  }
  R xiT[4] = {R(0), R(0), R(0), R(0)};  // Linearize: the sig_xi_terminal that stays in sig_z3_m (i2c.py:460, 488, 497)
  if (LIN) {
    if (c.has_x_terminal) {
      R sxT[4];
      w_ldconst<NBX>(w, kc.sxT, sxT);
      if (NZT > 0 && c.has_Qf) {
        // the back-calculated sig_xi_terminal (the Lagrange multiplier of the pinned covariance, i2c.py:455-462); with the
        // identity observation sig_z = S3f (S3f - S_T)^-1 S3f and sig_xi_terminal = sig_z - S3f
        // (S3f - S_T is symmetric but in general indefinite: the target may be tighter in some directions and looser in others)
        R ds[4], y[4], ys[4], sg[4], g[4] = {R(0), R(0), R(0), R(0)};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          ds[v] = (v < NBX && jx) ? s3m[v] - sxT[v] : R(0);
          y[v] = s3m[v];
        }
        if (!w_elim_signed<NBX, true>(w, ds, y, sg) && w.l == 0) set_status(a.status, b, 6, T - 1);
#pragma unroll
        for (int v = 0; v < 4; ++v) ys[v] = sg[v] * y[v];
        w_tn<NBX>(w, ys, y, g);
#pragma unroll
        for (int v = 0; v < 4; ++v) xiT[v] = g[v] - s3m[v];
      }
      m3m = jx ? kc.mxT[jxc] : R(0);
#pragma unroll
      for (int v = 0; v < 4; ++v) s3m[v] = sxT[v];
    } else if (NZT > 0 && c.has_Qf) {
      const R alpha = a.alpha[b];
      if (!w_kalman<NBX, true>(w, alpha, kc.xiT, kc.qf, c.qf_diag != 0, kc.zgT[j], &m3m, s3m) && w.l == 0) set_status(a.status, b, 6, T - 1);
      w_ldconst<NBX>(w, kc.xiT, xiT);
#pragma unroll
      for (int v = 0; v < 4; ++v) xiT[v] *= alpha;
    }
  }
  // terminal observation statistics (i2c.py:567-570, 989-992): tr(Qf (errT errT^T + sig_z3_m)), identity observation
  R trT = R(0);
  if (NZT > 0 && c.has_Qf) {
    R pm, pv, szt[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) szt[v] = s3m[v] + xiT[v];
    w_cost_share(w, c.qf_diag != 0, kc.qf, NBX, jx ? m3m - kc.zgT[jxc] : R(0), szt, &pm, &pv);
    trT = w_wavesum(w, 5, pm);
    if (q == 0 && jx) a.term_stats[(long)(3 + jxc) * B + b] = m3m;
#pragma unroll
    for (int v = 0; v < NBX; ++v)
      if (jx && w.row(v) >= j) a.term_stats[(long)(3 + NT + w_symidx(w.row(v), jxc)) * B + b] = szt[v];
  }
  if (w.l == 0) a.term_stats[b] = trT;
  *m3m_out = m3m;
}

// The forward rows of one cell as a backward cell needs them (filled from HBM; a prefetch buffer in the fused walk)
template <typename R> struct WFwdRow {
  R mu, m3f, sg[4], s3f[4], jt[4], zt;
};
template <class M, typename R, typename S, class KC>
I2C_FN void w_fetch_fwd(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a, const int b, const Wave<R>& w, const int tc, WFwdRow<R>& f) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZ = C::NZ, D = C::D, NBX = NX / 4;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  const unsigned long B = c.B;
  const unsigned WS = sizeof(S), bo = (unsigned)b * WS, rb = (unsigned)(B * WS);
  const int j = w.j, jxc = j < NX ? j : 0;
  const WIO<R, S> fw = w_fwd_cell<R, S>(a.fwd, C::E_FWD, B, tc, b);
  f.mu = fw.ld(j);
  f.m3f = fw.ld(O_MU3 + jxc);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    f.sg[v] = fw.ld(D + w_symidx(w.row(v), j));
    f.s3f[v] = v < NBX ? fw.ld(O_S3 + w_symidx(w.row(v) < NX ? w.row(v) : 0, jxc)) : R(0);
    f.jt[v] = v < NBX ? fw.ld(O_J + j * NX + (w.row(v) < NX ? w.row(v) : 0)) : R(0);  // J^T
  }
  {  // the per-cell target or a discarded dummy, through the buffer path and without a branch (see forward_wave_body); the choice
     // is made where the value is used (w_bwd_cell)
    // (the dummy: the first B doubles of the forward-message buffer, which this sweep only reads)
    const Window zw = make_window(c.z_per_cell ? (const void*)a.z : (const void*)a.fwd, (c.z_per_cell ? (unsigned long)c.T * NZ : 1ul) * B * sizeof(R));
    f.zt = wld<R>(zw, 0u, c.z_per_cell ? (unsigned)((((unsigned long)c.row(tc) * NZ + j) * B + b) * sizeof(R)) : (unsigned)(b * sizeof(R)));
  }
}

// One backward cell (i2c.py:574-608) given its forward rows and the smoothed next state (m3m column form, s3m accumulator
// layout): RTS update of the joint, expected cost of the posterior observation (this lane's shares: pm, pv; pa = the
// Linearize alpha statistic), controller, stores. On return mu / sg hold the posterior joint.
template <class M, typename R, typename S, bool LIN, class KC>
I2C_FN void w_bwd_cell(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a, const int b, const Wave<R>& w, const int t,
                       const WFwdRow<R>& f, const R m3m, const R* s3m, R* mu_out, R* sg, R* pm, R* pv, R* pa) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D, NBX = NX / 4;
  constexpr int O_K = D + sym(D), O_k = O_K + NU * NX, O_SK = O_k + NU;
  const int q = w.q, j = w.j;
  const unsigned long B = c.B;
  const unsigned WS = sizeof(S), bo = (unsigned)b * WS, rb = (unsigned)(B * WS);
  const bool jx = j < NX;
  const int jxc = jx ? j : 0, ju = jx ? 0 : j - NX;
  const WIO<R, S> po = w_post_cell<R, S>(a.post, C::E_POST, B, c.row(t), b, c.post_tm != 0);
  R mu = f.mu, jt[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    sg[v] = f.sg[v];
    jt[v] = f.jt[v];
  }
  // RTS update of the joint (i2c.py:580-583): mu += J (m3m - m3f), sig += J (S3m - S3f) J^T
  {
    R dr[4], ds[4], p1[4] = {R(0), R(0), R(0), R(0)};
    w_col2row<NBX>(w, 0, m3m - (jx ? f.m3f : R(0)), dr);
    R tsum = R(0);
#pragma unroll
    for (int v = 0; v < NBX; ++v) tsum += jt[v] * dr[v];
    mu += w_rowsum(w, tsum);
#pragma unroll
    for (int v = 0; v < 4; ++v) ds[v] = (v < NBX && jx) ? s3m[v] - f.s3f[v] : R(0);
    w_tn<NBX>(w, ds, jt, p1);
    w_tn<NBX>(w, jt, p1, sg);
  }
  // posterior observation moments = the joint itself (identity observation, i2c.py:594-596) and their expected cost
  const R ztv = c.z_per_cell ? f.zt : kc.zg[j];
  w_cost_share(w, c.qr_diag != 0, kc.qr, 4, mu - ztv, sg, pm, pv);
  *pa = R(0);
  if (LIN) {  // alpha statistic: the linearised marginal observation, block-diagonal in (x, u) (i2c.py:537-540)
    R sbd[4], pva;
#pragma unroll
    for (int v = 0; v < 4; ++v) sbd[v] = ((w.row(v) < NX) == jx) ? sg[v] : R(0);
    w_cost_share(w, c.qr_diag != 0, kc.qr, 4, mu - ztv, sbd, pa, &pva);
  }
  // controller (i2c.py:600-608): with [W | Y] = chol(sig_xx)^-1 [I | sig_xu]:  K^T = W^T Y, sigK = sig_uu - Y^T Y
  {
    R sxx[4], rh[4], lt[4], g[4] = {R(0), R(0), R(0), R(0)}, mr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      sxx[v] = (v < NBX && jx) ? sg[v] : R(0);
      rh[v] = v < NBX ? (jx ? (w.row(v) == j ? R(1) : R(0)) : sg[v]) : R(0);
    }
    if (!w_elim<NBX, 1, true>(w, sxx, rh, (R*)nullptr, lt) && w.l == 0) set_status(a.status, b, 7, t);
    w_tn<NBX>(w, rh, rh, g);
    w_col2row<NBX>(w, 1, jx ? mu : R(0), mr);
    R kx = R(0);
#pragma unroll
    for (int v = 0; v < NBX; ++v) {
      po.st_if(!jx, O_K + ju * NX + (w.row(v) < NX ? w.row(v) : 0), g[v]);
      kx += g[v] * mr[v];
    }
    kx = w_rowsum(w, kx);
    po.st_if(!jx && q == 0, O_k + ju, mu - kx);
    po.st_if(!jx && q >= ju, O_SK + q * (q + 1) / 2 + ju, sg[NBX] - g[NBX]);
  }
  po.st_if(q == 0, j, mu);
#pragma unroll
  for (int v = 0; v < 4; ++v) po.st_if(w.row(v) >= j, D + w_symidx(w.row(v), j), sg[v]);
  if (a.zpost) {
    S* zo = a.zpost + ((long)t * C::E_ZPOST) * B + b;
    if (q == 0) zo[(long)j * B] = (S)mu;
#pragma unroll
    for (int v = 0; v < 4; ++v)  // (Linearize: the linearised marginal observation has no state-action cross terms)
      if (w.row(v) >= j) zo[(long)(NZ + w_symidx(w.row(v), j)) * B] = (S)((LIN && (w.row(v) < NX) != jx) ? R(0) : sg[v]);
  }
  *mu_out = mu;
}
template <class M, typename R, typename S> I2C_FN void w_store_xm(const Consts<M, R>& c, const CellArgs<R, S>& a, const int b, const Wave<R>& w,
                                                                   const int t, const R m3m, const R* s3m) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NBX = NX / 4;
  const long B = c.B;
  const int j = w.j;
  const bool jx = j < NX;
  const int jxc = jx ? j : 0;
  S* xo = const_cast<S*>(a.xm) + ((long)t * C::E_XM) * B + b;
  if (w.q == 0 && jx) xo[(long)jxc * B] = (S)m3m;
#pragma unroll
  for (int v = 0; v < NBX; ++v)
    if (jx && w.row(v) >= j) xo[(long)(NX + w_symidx(w.row(v), jxc)) * B] = (S)s3m[v];
}

// fused schedule: the wave walks T-1..0 doing the whole cell
template <class M, typename R, typename S, bool LIN, class KC>
I2C_HD inline void backward_wave_body(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a, const int b, const Wave<R>& w) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZ = C::NZ, NZT = C::NZT, D = C::D, NBX = NX / 4;
  static_assert(D == 16 && NX % 4 == 0 && NZ == D && (NZT == NX || NZT == 0), "wave kernels: d = 16, identity observations");
  const unsigned long B = c.B;
  const int T = c.T;
  const bool jx = w.j < NX;
  R m3m, s3m[4];
  w_end_of_chain<M, R, S, LIN>(c, kc, a, b, w, &m3m, s3m);
  R acc_m = R(0), acc_v = R(0), acc_a = R(0);  // this lane's share of the cost sums over t (reduced once, after the walk)
  WFwdRow<R> nx;  // forward rows of a cell, fetched one cell ahead (see forward_wave_body)
  w_fetch_fwd<M, R, S>(c, kc, a, b, w, T - 1, nx);
  // settled before the loop (see forward_wave_body: loads pending on the loop-entry path cost a vmcnt(0) in every cell)
  nx.mu = opaque(nx.mu), nx.m3f = opaque(nx.m3f), nx.zt = opaque(nx.zt);
#pragma unroll
  for (int v = 0; v < 4; ++v) nx.sg[v] = opaque(nx.sg[v]), nx.s3f[v] = opaque(nx.s3f[v]), nx.jt[v] = opaque(nx.jt[v]);
  m3m = opaque(m3m);
#pragma unroll
  for (int v = 0; v < 4; ++v) s3m[v] = opaque(s3m[v]);
  for (int t = T - 1; t >= 0; --t) {
    const WFwdRow<R> f = nx;
    w_fetch_fwd<M, R, S>(c, kc, a, b, w, t > 0 ? t - 1 : 0, nx);
    if (a.xm) w_store_xm<M, R, S>(c, a, b, w, t, m3m, s3m);
    R mu, sg[4], pm, pv, pa;
    w_bwd_cell<M, R, S, LIN>(c, kc, a, b, w, t, f, m3m, s3m, &mu, sg, &pm, &pv, &pa);
    acc_m += pm;
    acc_v += pv;
    acc_a += pa;
    if (a.cell_stats) {
      const R cm = w_wavesum(w, 5, pm), cv = w_wavesum(w, 5, pv);
      if (w.l == 0) {
        a.cell_stats[((long)t * 2 + 0) * B + b] = cm;
        a.cell_stats[((long)t * 2 + 1) * B + b] = cv;
      }
    }
    m3m = jx ? mu : R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) s3m[v] = (v < NBX && jx) ? sg[v] : R(0);
  }
  const R sm = w_wavesum(w, 5, acc_m), sv = w_wavesum(w, 5, acc_v), sa = LIN ? w_wavesum(w, 5, acc_a) : sm;
  if (w.l == 0) {
    a.term_stats[B + b] = sa;
    a.term_stats[2 * B + b] = sv;
    if (LIN) a.term_stats[(long)(C::E_TERM - 1) * B + b] = sm;
  }
}

// two-pass schedule, pass 1: the sequential part alone -- the x-marginal of the RTS recursion (i2c.py:580-583 restricted to the
// state block; it is all cell t-1 needs from cell t). Writes xm[t] = (mu_x3_m, sig_x3_m) for every cell.
template <class M, typename R, typename S, class KC>
I2C_HD inline void backward_wave_scan_body(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a, const int b, const Wave<R>& w) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, D = C::D, NBX = NX / 4;
  const int T = c.T;
  const bool jx = w.j < NX;
  R m3m, s3m[4];
  w_end_of_chain<M, R, S, false>(c, kc, a, b, w, &m3m, s3m);
  WFwdRow<R> nx;
  w_fetch_fwd<M, R, S>(c, kc, a, b, w, T - 1, nx);
  for (int t = T - 1; t >= 0; --t) {
    const WFwdRow<R> f = nx;
    w_fetch_fwd<M, R, S>(c, kc, a, b, w, t > 0 ? t - 1 : 0, nx);
    w_store_xm<M, R, S>(c, a, b, w, t, m3m, s3m);
    if (t == 0) break;
    R dr[4], ds[4], jt[4], sg[4], p1[4] = {R(0), R(0), R(0), R(0)};
    w_col2row<NBX>(w, 0, m3m - (jx ? f.m3f : R(0)), dr);
    R tsum = R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      jt[v] = f.jt[v];
      sg[v] = f.sg[v];
      ds[v] = (v < NBX && jx) ? s3m[v] - f.s3f[v] : R(0);
      tsum += v < NBX ? jt[v] * dr[v] : R(0);
    }
    const R mu = f.mu + w_rowsum(w, tsum);
    w_tn<NBX>(w, ds, jt, p1);
    w_tn<NBX>(w, jt, p1, sg);
    m3m = jx ? mu : R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) s3m[v] = (v < NBX && jx) ? sg[v] : R(0);
  }
}
// two-pass schedule, pass 2: one wave per (t, b) cell; its cost terms go to cell_stats[t] (summed over t by k_reduce)
template <class M, typename R, typename S, class KC>
I2C_HD inline void backward_wave_cell_body(const Consts<M, R>& c, const KC& kc, const CellArgs<R, S>& a, const int t, const int b,
                                           const Wave<R>& w) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NBX = NX / 4;
  const long B = c.B;
  const int j = w.j;
  const bool jx = j < NX;
  const int jxc = jx ? j : 0;
  WFwdRow<R> f;
  w_fetch_fwd<M, R, S>(c, kc, a, b, w, t, f);
  R m3m, s3m[4];
  {
    const S* xi = a.xm + ((long)t * C::E_XM) * B + b;
    const R mv = (R)xi[(long)jxc * B];
    m3m = jx ? mv : R(0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const R sv = v < NBX ? (R)xi[(long)(NX + w_symidx(w.row(v) < NX ? w.row(v) : 0, jxc)) * B] : R(0);
      s3m[v] = (v < NBX && jx) ? sv : R(0);
    }
  }
  R mu, sg[4], pm, pv, pa;
  w_bwd_cell<M, R, S, false>(c, kc, a, b, w, t, f, m3m, s3m, &mu, sg, &pm, &pv, &pa);
  const R cm = w_wavesum(w, 5, pm), cv = w_wavesum(w, 5, pv);
  if (w.l == 0) {
    a.cell_stats[((long)t * 2 + 0) * B + b] = cm;
    a.cell_stats[((long)t * 2 + 1) * B + b] = cv;
  }
}

}  // namespace i2c
