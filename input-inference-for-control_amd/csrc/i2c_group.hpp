// Multi-lane ("group") form of the cubature cells: G lanes of one wavefront cooperate on ONE trajectory.
//
// The one-lane-per-trajectory cells of i2c_cell.hpp keep every block of a trajectory in the registers of a single lane.
// That stops scaling at d = nx + nu = 8 (a 16 x 16 joint covariance alone is 272 VGPRs) and it leaves a small batch on a
// few wavefronts, each issuing the whole cell. Here the blocks are ROW-DISTRIBUTED over the G lanes of a group:
//   * lane r owns row r of every matrix of the cell (joint covariance, Cholesky factor, cross-covariances, gains) in
//     registers, as a FULL row (symmetric matrices are not packed inside the kernel; they are packed again in HBM);
//   * vectors (means, targets, innovations) are REPLICATED in every lane of the group;
//   * lane j evaluates the model at the sigma-point pair m +/- sf L[:, j]  (the 2d + 1 points of quadrature.py:15-25
//     are spread over the lanes instead of being walked by one lane);
//   * everything that has to cross lanes (a Cholesky column, the rows of a gain, the sigma-point differences) goes
//     through the group's private LDS region. A group lives inside one wavefront, so LDS traffic is ordered by the
//     hardware (a wave's DS instructions execute in issue order): no s_barrier, only a compiler fence.
// The math is the reference's I2cCell (i2c/i2c.py:350-447, 544-610, 150-199) and QuadratureInference
// (i2c/inference/quadrature.py:15-58), in the centred pairwise form of i2c_cell.hpp::sp_transform.
//
// The host simulation (tests only) runs the G lanes of a group as G threads with a barrier where the device has the
// fence, so the SAME code is checked on a CPU-only box, including its synchronisation.
#pragma once
#include "i2c_cell.hpp"

#ifdef I2C_HOST_SIM
#include <sched.h>
#include <atomic>
#include <thread>
#include <vector>
#endif

#define I2C_MEM I2C_HD inline __attribute__((always_inline))

namespace i2c {

#ifdef I2C_HOST_SIM
struct HostBarrier {  // sense-reversing spin barrier for the G lane-threads of one group
  explicit HostBarrier(int n_) : n(n_) {}
  void wait() {
    const int g = gen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      count.store(0, std::memory_order_relaxed);
      gen.store(g + 1, std::memory_order_release);
    } else {
      while (gen.load(std::memory_order_acquire) == g) sched_yield();
    }
  }
  std::atomic<int> count{0}, gen{0};
  int n;
};
template <typename R> using lds_ptr = R*;
#else
template <typename R> using lds_ptr = __attribute__((address_space(3))) R*;
#endif

// One lane's view of its group: rank, the group's LDS region (three G x (G+2) matrices + four G-vectors).
// Rows are padded to G + 2 elements: lanes writing "their" row then hit different bank pairs (a row step of 2G + 4 dwords
// visits G distinct even residues modulo 64 for G = 4, 8, 16) and every row stays 16-byte aligned.
template <typename R, int G> struct Grp {
  static constexpr int LD = G + 2, MAT = G * LD, NVEC = 4;
  // region size = 2 (mod 32) elements: the regions of the groups of a wave start 4 banks apart, so the broadcast reads
  // (all lanes of a group at one address) of different groups never share a bank
  static constexpr int RAW = 3 * MAT + NVEC * G;
  static constexpr int SIZE = RAW + ((34 - RAW % 32) % 32);
  int r;
  lds_ptr<R> sh;
#ifdef I2C_HOST_SIM
  HostBarrier* bar;
#endif
  I2C_MEM lds_ptr<R> mat(int i) const { return sh + i * MAT; }
  I2C_MEM lds_ptr<R> vec(int i) const { return sh + 3 * MAT + i * G; }
  // Orders the group's LDS writes before its later LDS reads (and earlier reads before later writes).
  I2C_MEM void sync() const {
#ifdef I2C_HOST_SIM
    bar->wait();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
  }
};

// Diagnostic build only (-DI2C_GROUP_STAMPS, never in the shipped library): s_memtime stamps at the phase boundaries of the
// forward cell, summed per phase and printed by trajectory 0 -- where a lone wave spends its cycles.
#if defined(I2C_GROUP_STAMPS) && !defined(I2C_HOST_SIM)
#define I2C_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_last; stamp_last = now_; } while (0)
#else
#define I2C_STAMP(i) do { } while (0)
#endif

// Batch-wide constants a lane needs by ROW (runtime row index): unpacked, in LDS, filled once per workgroup.
template <class M, typename R> struct GConst {
  static constexpr int NX = M::NX, NZ = M::NZ, NT = M::NZT > 0 ? M::NZT : 1, NY = M::NY;
  R sig_xi0[NZ * NZ], sig_eta[NX * NX], sig_xiT0[NT * NT], sig_zeta[NY * NY], sig_x_term[NX * NX];
  R qr[NZ * NZ], qf[NT * NT];  // blkdiag(Q, R) and Qf unpacked (i2c.py:781-789 accepts any symmetric weight)
  R qr_d[NZ], qf_d[NT];
};
// `c` and `zeta` must be addressable memory (the host's structs; on the device the kernel-argument segment itself, see
// k_group): indexing a by-value kernel argument with a per-lane value would make hipcc copy its arrays to scratch.
template <class M, typename R, class DST>
I2C_FN void gconst_fill(DST& k, const Consts<M, R>* c, const R* zeta, const int tid, const int nthreads) {
  constexpr int NX = M::NX, NZ = M::NZ, NT = M::NZT > 0 ? M::NZT : 1, NY = M::NY;
  for (int e = tid; e < NZ * NZ; e += nthreads) k.sig_xi0[e] = c->sig_xi0[tri_any(e / NZ, e % NZ)];
  for (int e = tid; e < NX * NX; e += nthreads) k.sig_eta[e] = c->sig_eta[tri_any(e / NX, e % NX)];
  for (int e = tid; e < NT * NT; e += nthreads) k.sig_xiT0[e] = c->sig_xiT0[tri_any(e / NT, e % NT)];
  for (int e = tid; e < NY * NY; e += nthreads) k.sig_zeta[e] = zeta ? zeta[tri_any(e / NY, e % NY)] : R(0);
  for (int e = tid; e < NX * NX; e += nthreads) k.sig_x_term[e] = c->sig_x_term[tri_any(e / NX, e % NX)];
  for (int e = tid; e < NZ * NZ; e += nthreads) k.qr[e] = c->QR[tri_any(e / NZ, e % NZ)];
  for (int e = tid; e < NT * NT; e += nthreads) k.qf[e] = c->Qf[tri_any(e / NT, e % NT)];
  for (int e = tid; e < NZ; e += nthreads) k.qr_d[e] = c->QR[tri(e, e)];
  for (int e = tid; e < NT; e += nthreads) k.qf_d[e] = c->Qf[tri(e, e)];
}

// A per-lane integer the optimiser cannot see through. Without it hipcc recognises `j <= r` guards of an unrolled loop as
// a loop bound (trip count r + 1, array indexed at run time) and select chains on `r == i` as a table look-up: both turn
// a register array into a scratch array.
I2C_FN int opaque_i(int x) {
#ifndef I2C_HOST_SIM
  asm volatile("" : "+v"(x));
#endif
  return x;
}
// element r of a replicated vector (r is a per-lane value: a select chain, not an indexed register file)
template <int N, typename R> I2C_FN R g_sel(const R* v, const int r) {
  R x = v[0];
#pragma unroll
  for (int i = 1; i < N; ++i) x = (opaque_i(r) == i) ? v[i] : x;
  return x;
}
// packed index of (r, j) of a symmetric matrix, r a per-lane value with tr = r (r + 1) / 2
I2C_FN int symidx(const int r, const int tr, const int j) { return j <= r ? tr + j : tri(j, j) - j + r; }

// every lane contributes one value (its rank's), every lane receives all N
template <int N, typename R, int G> I2C_FN void g_gather(const Grp<R, G>& g, const int slot, const R v, R* out) {
  const auto s = g.vec(slot);
  g.sync();
  s[g.r] = v;
  g.sync();
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = s[i];
}
// sum over the first N ranks of three per-lane values, in rank order (deterministic)
template <int N, typename R, int G>
I2C_FN void g_sum3(const Grp<R, G>& g, const R v0, const R v1, const R v2, R* s0, R* s1, R* s2) {
  const auto a = g.vec(0), b = g.vec(1), c = g.vec(2);
  g.sync();
  a[g.r] = v0;
  b[g.r] = v1;
  c[g.r] = v2;
  g.sync();
  R x = R(0), y = R(0), z = R(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    x += a[i];
    y += b[i];
    z += c[i];
  }
  *s0 = x;
  *s1 = y;
  *s2 = z;
}

// Cholesky of an N x N SPD matrix whose row r sits in lane r (full row). Right-looking, TWO columns per LDS exchange (the
// exchange latency, not the arithmetic, bounds a lone wave): the lanes publish the current columns k and k + 1 UNSCALED,
// everybody reads both (pivots included) and forms, redundantly,
//   l_k = c_k / sqrt(c_k[k]),   c'_{k+1} = c_{k+1} - l_k l_k[k+1]   (the step-k update of column k + 1),
//   l_{k+1} = c'_{k+1} / sqrt(c'_{k+1}[k+1]),
// then updates its own trailing row with both. On return: row = row r of L with zeros above the diagonal, rinv[j] =
// 1 / L[j][j] in every lane, L row-major in LDS matrix `m`. Returns (in every lane) whether the last pivot is positive,
// which is equivalent to all pivots being positive (see chol() in i2c_linalg.hpp).
template <int N, typename R, int G> I2C_FN bool g_chol(const Grp<R, G>& g, const int m, R* row, R* rinv) {
  constexpr int LD = Grp<R, G>::LD;
  constexpr bool FENCE = N >= 6;  // bound the live state per step (see sched_fence, i2c_linalg.hpp)
  const auto Tm = g.mat(m);
  const int r = g.r;
  R last = R(0);
  g.sync();
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    const bool two = k + 1 < N;
    Tm[k * LD + r] = row[k];
    if (two) Tm[(k + 1) * LD + r] = row[k + 1];
    g.sync();
    R c0[N], c1[N];
#pragma unroll
    for (int j = k; j < N; ++j) {
      c0[j] = Tm[k * LD + j];
      if (two && j > k) c1[j] = Tm[(k + 1) * LD + j];
    }
    const R r0 = r_rsqrt(c0[k]);
    rinv[k] = r0;
    const R l0 = row[k] * r0;  // L[r][k]
    if (!two) {
      last = c0[k];
      row[k] = l0;
    } else {
      // column k + 1 after the step-k update, for every row j > k (replicated): c1[j] - c0[j] c0[k+1] / c0[k]
      const R s01 = c0[k + 1] * (r0 * r0);
#pragma unroll
      for (int j = k + 1; j < N; ++j) c1[j] -= c0[j] * s01;
      if (k + 1 == N - 1) last = c1[k + 1];
      const R r1 = r_rsqrt(c1[k + 1]);
      rinv[k + 1] = r1;
      const R t0 = l0 * r0;                                 // L[r][k] / sqrt(pivot_k)
      const R l1 = (row[k + 1] - c0[k + 1] * t0) * r1;      // L[r][k+1]
      const R t1 = l1 * r1;
#pragma unroll
      for (int j = k + 2; j < N; ++j) {  // two fused multiply-adds (written as one expression it is mul + fma + add)
        row[j] -= t0 * c0[j];
        row[j] -= t1 * c1[j];
      }
      row[k] = l0;
      row[k + 1] = l1;
    }
    sched_fence<FENCE>();
  }
  g.sync();  // the unscaled columns are consumed: the matrix now receives L itself
#pragma unroll
  for (int j = 0; j < N; ++j) {
    row[j] = (j <= r) ? row[j] : R(0);
    Tm[r * LD + j] = row[j];
  }
  g.sync();
  return last > R(0);
}

// L y = b (forward substitution) for one or two right-hand sides per lane; L is in LDS matrix m, rinv replicated.
// Row i + 1 of L is fetched while row i is consumed (the LDS latency of every row would otherwise be exposed to the lone
// wave); a scheduling fence per row keeps the live state at two rows.
template <int N, int NRHS, typename R, int G>
I2C_FN void g_fsub(const Grp<R, G>& g, const int m, const R* rinv, R* b0, R* b1) {
  constexpr int LD = Grp<R, G>::LD;
  constexpr bool FENCE = N >= 6;
  const auto L = g.mat(m);
  R lc[N], ln[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i + 1 < N) {
#pragma unroll
      for (int k = 0; k <= i; ++k) ln[k] = L[(i + 1) * LD + k];
    }
    R v0 = b0[i], v1 = NRHS > 1 ? b1[i] : R(0);
#pragma unroll
    for (int k = 0; k < i; ++k) {
      v0 -= lc[k] * b0[k];
      if (NRHS > 1) v1 -= lc[k] * b1[k];
    }
    b0[i] = v0 * rinv[i];
    if (NRHS > 1) b1[i] = v1 * rinv[i];
    if (i + 1 < N) {
#pragma unroll
      for (int k = 0; k <= i; ++k) lc[k] = ln[k];
    }
    sched_fence<FENCE>();
  }
}
// L^T x = y (back substitution), one right-hand side per lane (column i - 1 of L fetched while column i is consumed)
template <int N, typename R, int G> I2C_FN void g_bsub(const Grp<R, G>& g, const int m, const R* rinv, R* y) {
  constexpr int LD = Grp<R, G>::LD;
  constexpr bool FENCE = N >= 6;
  const auto L = g.mat(m);
  R lc[N], ln[N];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    if (i > 0) {
#pragma unroll
      for (int k = i; k < N; ++k) ln[k] = L[k * LD + (i - 1)];
    }
    R v = y[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) v -= lc[k] * y[k];
    y[i] = v * rinv[i];
    if (i > 0) {
#pragma unroll
      for (int k = i; k < N; ++k) lc[k] = ln[k];
    }
    sched_fence<FENCE>();
  }
}

// Walk k = 0..K-1 doing apply(fetch(k)), where fetch only reads LDS into a small struct P and apply only consumes it.
// PIPE: a ROLLED two-stage software pipeline -- the reads of step k + 1 are issued before step k is consumed, pinned there by
// scheduling fences. Left to itself the compiler keeps one or two b128 reads in flight and waits on each in turn, and a lone
// wave then pays the LDS latency a dozen times per step (measured on the 12-state quadrotor: forward 1.14 -> 1.08 ms from
// the transform loop alone). Rolled, because no register array may be indexed by k at d = 16 (fully unrolled, the transform
// loop alone drove the forward kernel 1.3 KB per lane into scratch). Small problems: fully unrolled, the compiler's schedule.
template <int K, bool PIPE, class P, class Fetch, class Apply> I2C_FN void g_walk(const Fetch& fetch, const Apply& apply) {
  if constexpr (PIPE && K % 2 == 0) {
    P p0, p1;
    fetch(0, p0);
#pragma unroll 1
    for (int k = 0; k < K; k += 2) {
      fetch(k + 1, p1);
      sched_fence<true>();
      apply(p0);
      sched_fence<true>();
      fetch(k + 2 < K ? k + 2 : k + 1, p0);  // the last trip re-reads its own second step: no branch
      sched_fence<true>();
      apply(p1);
      sched_fence<true>();
    }
  } else {
    constexpr int UF = PIPE ? 2 : K;
#pragma unroll UF
    for (int k = 0; k < K; ++k) {
      P p;
      fetch(k, p);
      apply(p);
    }
  }
}

template <class ST, int DOUT> constexpr bool st_identity() {
  for (int k = 0; k < DOUT; ++k)
    if (ST::lin(k) != k) return false;
  return true;
}

// Gaussian push-through N(m, L L^T) -> (my, Sy [, Sxy])  (quadrature.py:27-58), rows distributed:
//   in : m [DIN] replicated; Lrow = row r of L (zeros above the diagonal); L row-major in LDS matrix mL
//   out: my [DOUT] replicated; Sy = row r of the output covariance (lanes r < DOUT); Sxy = row r of the DIN x DOUT
//        cross-covariance (lanes r < DIN)
// Lane j evaluates the pair m +/- sf L[:, j] and publishes a_j = (y+ - y0) + (y- - y0), d_j = y+ - y-; pass-through
// outputs (ST::lin(k) >= 0) need no evaluation: a_j = 0 and d_j = 2 sf L[lin(k)][j] exactly. Then every lane
// accumulates ITS row of  Sy = wi/2 sum_j (a_j a_j^T + d_j d_j^T) - wi^2 A A^T (+ terms in 1 - W)  and of
// Sxy = wi sf L [d_0 .. d_{DIN-1}]^T from the published vectors -- the formulas of sp_transform (i2c_cell.hpp).
// LDS matrices mA and mD hold the a_j and d_j (row j); mL may be overwritten afterwards.
template <class M, class ST, int DIN, int DOUT, bool CROSS, typename R, int G, class F>
I2C_FN void g_transform(const Grp<R, G>& g, const int mL, const int mA, const int mD, const Rule<R>& rule, const R* m,
                        const R* Lrow, const F& f, R* my, R* Sy, R* Sxy) {
  constexpr int LD = Grp<R, G>::LD;
  constexpr int NA = M::NA, NA1 = NA > 0 ? NA : 1;
  const int r = g.r;
  const auto Lm = g.mat(mL), Am = g.mat(mA), Dm = g.mat(mD);
  R s0[NA1], c0[NA1];
#pragma unroll
  for (int q = 0; q < NA; ++q) r_sincos(m[M::ang(q)], &s0[q], &c0[q]);
  R y0[DOUT];
  f(m, s0, c0, y0);
  {
    // column r of sf L (zero for the lanes beyond the input dimension)
    R col[DIN], xp[DIN], xm[DIN];
#pragma unroll
    for (int i = 0; i < DIN; ++i) {
      const R l = Lm[i * LD + r];
      col[i] = (r < DIN) ? rule.sf * l : R(0);
      xp[i] = m[i] + col[i];
      xm[i] = m[i] - col[i];
    }
    R sp[NA1], cp[NA1], sm[NA1], cm[NA1];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      R sd, cd;
      r_sincos_small(col[M::ang(q)], &sd, &cd);
      sp[q] = s0[q] * cd + c0[q] * sd;
      cp[q] = c0[q] * cd - s0[q] * sd;
      sm[q] = s0[q] * cd - c0[q] * sd;
      cm[q] = c0[q] * cd + s0[q] * sd;
    }
    R yp[DOUT], ym[DOUT];
    f(xp, sp, cp, yp);
    f(xm, sm, cm, ym);
    g.sync();  // mA / mD may still be read by the previous user
#pragma unroll
    for (int k = 0; k < DOUT; ++k) {
      const bool lin = ST::lin(k) >= 0;
      const R ak = lin ? R(0) : (yp[k] - y0[k]) + (ym[k] - y0[k]);
      const R dk = lin ? R(2) * col[lin ? ST::lin(k) : 0] : yp[k] - ym[k];
      Am[r * LD + k] = ak;
      Dm[r * LD + k] = dk;
    }
    g.sync();
  }
  R A[DOUT], Ar = R(0);
#pragma unroll
  for (int k = 0; k < DOUT; ++k) {
    A[k] = R(0);
    Sy[k] = R(0);
    if (CROSS) Sxy[k] = R(0);
  }
  // base of the mean now, so that neither m nor y0 has to survive the accumulation loop
  R myb[DOUT];
#pragma unroll
  for (int k = 0; k < DOUT; ++k) myb[k] = rule.W * (ST::lin(k) >= 0 ? m[ST::lin(k) >= 0 ? ST::lin(k) : 0] : y0[k]);
  // The walk over the points (g_walk): the lane's own row of L is read back from LDS matrix mL, which phase 1 left intact,
  // so the live state is the three accumulator rows and the published pairs in flight.
  const auto Lr = Lm + r * LD;
  // the LDS reads of one point: its published pair (rows j of mD / mA), this lane's own entries and L[r][j]
  struct Pt {
    R d[DOUT], a[DOUT], ar, dr, lj;
  };
  const auto fetch = [&](const int j, Pt& p) {
    p.ar = Am[j * LD + r];  // junk for r >= DOUT: those lanes' Sy is never used
    p.dr = Dm[j * LD + r];
    p.lj = CROSS ? Lr[j] : R(0);
#pragma unroll
    for (int l = 0; l < DOUT; ++l) {
      p.d[l] = Dm[j * LD + l];
      p.a[l] = ST::lin(l) < 0 ? Am[j * LD + l] : R(0);
    }
  };
  const auto accumulate = [&](const Pt& p) {
    Ar += p.ar;
#pragma unroll
    for (int l = 0; l < DOUT; ++l) {
      R acc = Sy[l] + p.dr * p.d[l];
      if (ST::lin(l) < 0) {
        A[l] += p.a[l];
        acc += p.ar * p.a[l];
      }
      Sy[l] = acc;
      if (CROSS) Sxy[l] += p.lj * p.d[l];
    }
  };
  g_walk<DIN, (DIN * DOUT >= 96), Pt>(fetch, accumulate);
  const R hw = R(0.5) * rule.wi, w2 = rule.wi * rule.wi, cs = rule.wi * rule.sf;
#pragma unroll
  for (int k = 0; k < DOUT; ++k) {
    my[k] = myb[k] + rule.wi * A[k];
    Sy[k] = hw * Sy[k] - w2 * Ar * A[k];
    if (CROSS) Sxy[k] = cs * Sxy[k];
  }
  if (!rule.unit) {  // sum of weights != 1: the reference's m m^T term no longer cancels (see sp_transform)
    const R omw = R(1) - rule.W, iw = r_rcp(rule.W);
    R yc[DOUT];
#pragma unroll
    for (int l = 0; l < DOUT; ++l) yc[l] = myb[l] * iw;
    const R ycr = g_sel<DOUT>(yc, r);
#pragma unroll
    for (int l = 0; l < DOUT; ++l) Sy[l] += omw * (rule.W * ycr * yc[l] + rule.wi * (Ar * yc[l] + ycr * A[l]));
  }
}

// Kalman-style update of N(mu, S) (dimension DX) on an observation with moments (mz, Sz + noise, Sxz) and target zt
// (i2c.py:394-403 / 435-443), rows distributed: mu and the innovation q = zt - mz replicated; S row r (lanes r < DX); Sz
// row r (lanes r < DZ, noise included); Sxz row r (lanes r < DX). With C = chol(Sz): V = Sxz C^-T (row r per lane), mu += V C^-1 q,
// S -= V V^T. Uses LDS matrices 0 (C) and 1 (V) and vector slot 0. *mu_own receives element r of the new mean.
template <int DX, int DZ, typename R, int G>
I2C_FN bool g_kalman(const Grp<R, G>& g, R* mu, R* S, R* q, R* Sz, R* Sxz, R* mu_own) {
  constexpr int LD = Grp<R, G>::LD;
  const int r = g.r;
  R rinv[DZ];
  const bool ok = g_chol<DZ>(g, 0, Sz, rinv);
  g_fsub<DZ, 2>(g, 0, rinv, q, Sxz);
  R dmu = R(0);
#pragma unroll
  for (int k = 0; k < DZ; ++k) dmu += Sxz[k] * q[k];
  const R own = g_sel<DX>(mu, r) + dmu;
  *mu_own = own;
  // V is published TRANSPOSED (row k of the LDS matrix = column k of V): the update S_r -= V_r V^T then walks the
  // observation index reading one contiguous row per step (b128 reads, half the LDS instructions of a column walk), the
  // lane's own V_r[k] among them (g_walk).
  const auto Vt = g.mat(1);
  g.sync();
#pragma unroll
  for (int k = 0; k < DZ; ++k) Vt[k * LD + r] = Sxz[k];
  g_gather<DX>(g, 0, own, mu);  // its syncs also publish V
  struct Col {
    R v[DX], own;
  };
  const auto fetch = [&](const int k, Col& c) {
    c.own = Vt[k * LD + r];
#pragma unroll
    for (int j = 0; j < DX; ++j) c.v[j] = Vt[k * LD + j];
  };
  const auto apply = [&](const Col& c) {
#pragma unroll
    for (int j = 0; j < DX; ++j) S[j] -= c.own * c.v[j];
  };
  g_walk<DZ, (DX * DZ >= 96), Col>(fetch, apply);
  return ok;
}

// Expected quadratic cost of N(mz, Sz) under a DIAGONAL weight (every shipped Q, R, Qf): m = err^T W err + tr(Sz W),
// v = 2 tr((Sz W)^2) + 4 err^T W Sz W err (i2c.py:1034-1043). Sz row r per lane; wd = diag(W) in LDS.
template <int N, typename R, int G, class WD>
I2C_FN void g_cost(const Grp<R, G>& g, const WD wd, const R* mz, const R* Sz, const R* zt, R* m, R* v) {
  const int r = g.r;
  R err[N], we[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    err[i] = mz[i] - zt[i];
    we[i] = wd[i] * err[i];
  }
  const int rc = r < N ? r : N - 1;
  const R wr = wd[rc], er = g_sel<N>(err, r), Srr = g_sel<N>(Sz, r);
  R tr2 = R(0), quad = R(0);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    tr2 += Sz[j] * (Sz[j] * (wr * wd[j]));
    quad += Sz[j] * ((wr * er) * we[j]);
  }
  const bool on = r < N;
  R mm, t2, qd;
  g_sum3<N>(g, on ? wr * er * er + wr * Srr : R(0), on ? tr2 : R(0), on ? quad : R(0), &mm, &t2, &qd);
  *m = mm;
  *v = R(2) * t2 + R(4) * qd;
}

// The same for ANY symmetric weight W (N x N, unpacked in LDS): lane r forms row r of P = Sz W and (W err)_r;
//   m = sum_r [err_r (W err)_r + Sz_r . W_r],  tr((Sz W)^2) = sum_r sum_l P[r][l] P[l][r]  (P exchanged through LDS matrix
//   1, column r read back),  err^T W Sz W err = sum_r (W err)_r (Sz_r . W err).
// The lane kernels' gaussian_cost (i2c_cell.hpp) is the one-lane form of the same sums.
template <int N, typename R, int G, class WM>
I2C_FN void g_cost_full(const Grp<R, G>& g, const WM W, const R* mz, const R* Sz, const R* zt, R* m, R* v) {
  constexpr int LD = Grp<R, G>::LD;
  const int r = g.r, rc = r < N ? r : N - 1;
  const bool on = r < N;
  R err[N];
#pragma unroll
  for (int i = 0; i < N; ++i) err[i] = mz[i] - zt[i];
  R werr_own = R(0), trSW = R(0);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const R w = W[rc * N + j];
    werr_own += w * err[j];
    trSW += Sz[j] * w;
  }
  R werr[N];
  g_gather<N>(g, 0, werr_own, werr);
  const auto Pm = g.mat(1);
  R quad = R(0);
#pragma unroll
  for (int j = 0; j < N; ++j) quad += Sz[j] * werr[j];
  quad *= werr_own;
  g.sync();
#pragma unroll 1
  for (int l = 0; l < N; ++l) {  // row r of P = Sz W (W is symmetric: its column l is its row l)
    R pv = R(0);
#pragma unroll
    for (int j = 0; j < N; ++j) pv += Sz[j] * W[l * N + j];
    if (on) Pm[r * LD + l] = pv;  // lanes beyond the weight's dimension own no row
  }
  g.sync();
  R tr2 = R(0);
#pragma unroll
  for (int l = 0; l < N; ++l) tr2 += Pm[rc * LD + l] * Pm[l * LD + rc];
  const R er = g_sel<N>(err, r);
  R mm, t2, qd;
  g_sum3<N>(g, on ? er * werr_own + trSW : R(0), on ? tr2 : R(0), on ? quad : R(0), &mm, &t2, &qd);
  *m = mm;
  *v = R(2) * t2 + R(4) * qd;
}

// addressing of one [E][B] cell block of a device buffer for the lanes of trajectory b
template <typename R> struct GIO {
  Window w;
  unsigned rb, bo;
  I2C_MEM R ld(const int e) const { return wld<R>(w, 0u, (unsigned)e * rb + bo); }
  I2C_MEM R ldo(const unsigned byte_off) const { return wld<R>(w, 0u, byte_off); }  // offset precomputed by the caller
  I2C_MEM void st(const int e, const R v) const { wst(w, 0u, (unsigned)e * rb + bo, v); }
  // Predicated store without touching EXEC: on the device a lane that must not store gets an offset beyond the window, and
  // the buffer unit drops out-of-range stores (raw buffer, range-checked against num_records); one select instead of a
  // compare + s_and_saveexec + s_or per store -- the packed-symmetric rows have ~60 such stores per cell.
  I2C_MEM void st_if(const bool on, const int e, const R v) const {
#ifdef I2C_HOST_SIM
    if (on) wst(w, 0u, (unsigned)e * rb + bo, v);
#else
    wst(w, 0u, on ? (unsigned)e * rb + bo : 0x80000000u, v);  // launch_group's callers refuse windows of 2 GiB and more (Impl::group_supported)
#endif
  }
};
template <typename R> I2C_FN GIO<R> gio(const R* base, const unsigned long elems, const unsigned rb, const unsigned bo) {
  return GIO<R>{make_window(base, elems * rb), rb, bo};
}
// a cell block of the posterior / prior buffer: [E][B] or, trajectory-major (Consts::post_tm), [B][E] -- the same E * B elements,
// element e of trajectory b at byte e * rbp + bop with (rbp, bop) = (B W, b W) or (W, b E W)
template <typename R> I2C_FN GIO<R> gio_post(const R* base, const unsigned long elems, const unsigned long B, const unsigned rbp, const unsigned bop) {
  return GIO<R>{make_window(base, elems * B * sizeof(R)), rbp, bop};
}

// Joint over (x, u) from a state message (mu_x replicated, sx = row r of sig_x for the state lanes) and a controller
// row block (i2c.py:361-387 forward feedback prior with Kt = rho K; i2c.py:158-179 propagation):
//   mean_u = qmu_u + Kt (mu_x - qmu_x);  cross = Kt sig_x;
//   sig_u[p][q] = puu[p][q] - sub * Kt[p] . pux[q] + Kt (sig_x - sq * qxx) Kt^T [p][q]        (p >= q)
// forward feedback prior: sub = 1, sq = 0 (pux = prior action-state block); propagation: sub = 0, sq = 1 (qxx = the
// posterior state covariance); add_quad = false drops the quadratic term (feed-forward cells of the propagation). Kt rows (scaled) arrive in the action lanes' Krow; prow = row r of the prior joint.
// Uses LDS matrices 0, 1, 2. Out: mu0 replicated, s0 = row r of the joint covariance.
template <int NX, int NU, typename R, int G>
I2C_FN void g_joint(const Grp<R, G>& g, const R* mu_x, const R* sx, const R* Krow, const R* prow, const R* qmu_x,
                    const R* qmu_u, const bool sub_ux, const bool sub_qxx, const bool add_quad, R* mu0, R* s0) {
  constexpr int LD = Grp<R, G>::LD, D = NX + NU;
  static_assert(2 * NU <= LD, "group too narrow for the action block exchange");
  const int r = g.r;
  const bool is_x = r < NX, is_u = r >= NX && r < D;
  const int ru = is_u ? r - NX : 0;
  const auto XC = g.mat(0), Km = g.mat(1), Pm = g.mat(2);
  g.sync();
  if (is_u) {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      Km[ru * LD + k] = Krow[k];
      Pm[ru * LD + k] = prow[k];
    }
  }
  g.sync();
#pragma unroll
  for (int i = 0; i < NX; ++i) mu0[i] = mu_x[i];
  R cr[NU], crd[NU];
#pragma unroll
  for (int a = 0; a < NU; ++a) {
    R v = qmu_u[a], c1 = R(0), c2 = R(0);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const R kt = Km[a * LD + k];
      v += kt * (mu_x[k] - qmu_x[k]);
      c1 += sx[k] * kt;
      c2 += (sub_qxx ? sx[k] - prow[k] : sx[k]) * kt;
    }
    mu0[NX + a] = v;
    cr[a] = c1;
    crd[a] = c2;
    sched_fence<(NX >= 6)>();  // one gain row in flight at a time (all NU * NX reads at once overflow the registers)
  }
  if (is_x) {  // matrix 0 is not read between the sync above and here
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      XC[r * LD + a] = cr[a];
      XC[r * LD + NU + a] = crd[a];
    }
  }
  g.sync();
  // The nu x nu action block: lane r < nu^2 forms entry (r / nu, r % nu) -- one entry per lane instead of every lane walking
  // all of them (only the nu action lanes need the result; the walk was 14 % of the d = 16 forward cell's instructions).
  static_assert(NU * NU <= G, "one action-block entry per lane");
  {
    const int ea = r < NU * NU ? r / NU : 0, ec = r < NU * NU ? r % NU : 0;
    const int p = ea > ec ? ea : ec, q = ea > ec ? ec : ea;
    R v = R(0);
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const R ktq = Km[q * LD + k];
      if (sub_ux) v -= Km[p * LD + k] * Pm[q * LD + k];
      if (add_quad) v += XC[k * LD + NU + p] * ktq;
    }
    g.vec(3)[r] = v;  // slot 3 is not used by the gathers / sums around this call
  }
  g.sync();
  R su[D];
#pragma unroll
  for (int j = 0; j < NX; ++j) su[j] = XC[j * LD + ru];
#pragma unroll
  for (int c = 0; c < NU; ++c) su[NX + c] = prow[NX + c] + g.vec(3)[ru * NU + c];
#pragma unroll
  for (int j = 0; j < D; ++j) s0[j] = is_x ? (j < NX ? sx[j] : cr[j < NX ? 0 : j - NX]) : su[j];
}

// End of the chain with a terminal state prior (covariance control, i2c.py:548-559), rows distributed: the filtered terminal
// state (m3f replicated, s3f = row r of sig_x3_f) is multiplied with the tempered prior N(mu_T, sig_T):
//   St = temp sig_x3_f;  sig_x3_m = St - St (sig_T + St)^-1 St = St - W^T W, W = chol(sig_T + St)^-1 St;
//   mu_x3_m = sig_x3_m (St^-1 mu_x3_f + sig_T^-1 mu_T).
// In place: m3f / s3f become mu_x3_m / row r of sig_x3_m. Returns false if a factorisation fails. Uses LDS matrices 0, 1.
template <class M, typename R, int G, class KC>
I2C_FN bool g_end_of_chain(const Grp<R, G>& g, const Consts<M, R>& c, const KC& kc, const R tmp, R* m3, R* s3) {
  constexpr int NX = M::NX, LD = Grp<R, G>::LD;
  const int r = g.r, rx = r < NX ? r : NX - 1;
  R St[NX], Ss[NX], sT[NX], rinv[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    St[j] = tmp * s3[j];
    sT[j] = kc.sig_x_term[rx * NX + j];
    Ss[j] = sT[j] + St[j];
  }
  bool ok = g_chol<NX>(g, 0, Ss, rinv);
  R w[NX];  // column r of W = C^-1 St (St is symmetric: its column r is this lane's row)
#pragma unroll
  for (int j = 0; j < NX; ++j) w[j] = St[j];
  g_fsub<NX, 1>(g, 0, rinv, w, (R*)nullptr);
  const auto Wm = g.mat(1);
  g.sync();
#pragma unroll
  for (int k = 0; k < NX; ++k) Wm[r * LD + k] = w[k];
  g.sync();
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    R v = St[j];
#pragma unroll
    for (int k = 0; k < NX; ++k) v -= w[k] * Wm[j * LD + k];
    s3[j] = v;
  }
  // rhs = St^-1 mu_x3_f + sig_T^-1 mu_T, every lane for itself (replicated vectors) against the two factors in LDS
  R r1[NX], r2[NX], Lt[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    Lt[j] = St[j];
    r1[j] = m3[j];
    r2[j] = c.mu_x_term[j];
  }
  ok = g_chol<NX>(g, 0, Lt, rinv) && ok;
  g_fsub<NX, 1>(g, 0, rinv, r1, (R*)nullptr);
  g_bsub<NX>(g, 0, rinv, r1);
  ok = g_chol<NX>(g, 0, sT, rinv) && ok;
  g_fsub<NX, 1>(g, 0, rinv, r2, (R*)nullptr);
  g_bsub<NX>(g, 0, rinv, r2);
  R own = R(0);
#pragma unroll
  for (int j = 0; j < NX; ++j) own += s3[j] * (r1[j] + r2[j]);
  g_gather<NX>(g, 0, own, m3);
  return ok;
}

// KL(N(mu, sig) || N(mu_T, sig_T)) of covariance control (i2c.py:1012-1019, mvn_kl_divergence :1223-1229) from the two Cholesky
// factors, rows distributed (sx = row r of sig): log det ratio = 2 sum log(L2_ii / L1_ii), tr(sig_T^-1 sig) = ||L2^-1 L1||_F^2.
template <class M, typename R, int G, class KC>
I2C_FN R g_terminal_kl(const Grp<R, G>& g, const Consts<M, R>& c, const KC& kc, const R* mu, const R* sx, bool* ok) {
  constexpr int NX = M::NX, LD = Grp<R, G>::LD;
  const int r = g.r, rx = r < NX ? r : NX - 1;
  R L1[NX], L2[NX], r1[NX], r2[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    L1[j] = sx[j];
    L2[j] = kc.sig_x_term[rx * NX + j];
  }
  *ok = g_chol<NX>(g, 0, L1, r1);
  *ok = g_chol<NX>(g, 1, L2, r2) && *ok;
  R logdet = R(0), dq[NX], col[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    logdet += r_log(r1[i]) - r_log(r2[i]);  // log(L2_ii) - log(L1_ii) with r = 1 / L_ii
    dq[i] = c.mu_x_term[i] - mu[i];
    col[i] = g.mat(0)[i * LD + rx];  // column r of L1 (zeros above the diagonal)
  }
  g_fsub<NX, 2>(g, 1, r2, col, dq);
  R tr = R(0), maha = R(0);
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    tr += col[i] * col[i];
    maha += dq[i] * dq[i];
  }
  R trs, d0, d1;
  g_sum3<NX>(g, r < NX ? tr : R(0), R(0), R(0), &trs, &d0, &d1);
  return R(0.5) * (R(2) * logdet + trs + maha - R(NX));
}

// ------------------------------------------------------------------------------------------
// Forward sweep (i2c.py:876-880 over :350-447): the group walks its trajectory through all T cells.
// ------------------------------------------------------------------------------------------
// LEANG: the common case fixed at compile time (one shared target, the trajectory's temperature, no joint-prior output, ring at
// its origin). With the per-cell target behind a run-time branch in the MIDDLE of the cell, the s_waitcnt at the join waits for
// vmcnt(0) in every cell -- the rows just prefetched and the acknowledgement of the stores just issued -- whether or not the
// branch is taken (see propagate_body); the generic variant keeps that, both variants fetch flags and temperature through the
// buffer path and settle the first prefetch before the loop.
template <class M, typename R, int G, bool LEANG = false, class KC>
I2C_HD inline void forward_group_body(const Consts<M, R>& c, const KC& kc, const FwdArgs<R>& a, const int b,
                                      const Grp<R, G>& g_in) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D, NT = C::NZT1;
  static_assert(G >= D && G >= NZ && G >= NT, "one matrix row per lane");
  constexpr int O_K = D + sym(D), O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  constexpr bool OBS_ID = st_identity<ObsStruct<M>, NZ>() && NZ <= D;
  constexpr bool TERM_ID = st_identity<TermStruct<M>, NT>() && NT <= NX;
  const Grp<R, G>& g = g_in;
  const int r = g.r;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned W = sizeof(R), bo = (unsigned)b * W, rb0 = (unsigned)(B * W);
  const unsigned rbp = c.post_tm ? W : rb0, bop = c.post_tm ? (unsigned)b * C::E_POST * W : bo;  // prior buffer (gio_post)
  const int rx = r < NX ? r : NX - 1, rd = r < D ? r : D - 1, rz = r < NZ ? r : NZ - 1, rt = r < NT ? r : NT - 1;
  const int trx = rx * (rx + 1) / 2, trd = rd * (rd + 1) / 2;
  const bool is_u = r >= NX && r < D;
  const int ru = is_u ? r - NX : 0;
  const R alpha_traj = a.alpha[b];
  int fail = 0;
  const bool z_per_cell = LEANG ? false : c.z_per_cell != 0;
  const R* const alpha_cell = LEANG ? nullptr : a.alpha_cell;
  R* const prior_out = LEANG ? nullptr : a.prior_out;
  const Window ffw = make_window(a.ff, (unsigned long)T);
  const Window alw = make_window(alpha_cell ? alpha_cell : a.alpha, (alpha_cell ? (unsigned long)T : 1ul) * B * W);

  R mu_x[NX], sx[NX];
  g_gather<NX>(g, 0, a.x0[(long)rx * B + b], mu_x);
#pragma unroll
  for (int j = 0; j < NX; ++j) sx[j] = a.sig_x0[(long)symidx(rx, trx, j) * B + b];

  // The prior rows of a cell do not depend on the recursion: they are fetched ONE CELL AHEAD (all global loads of a cell in
  // one batch, issued right after the previous cell has consumed its own), so that the memory round trip is covered by a
  // cell's worth of work instead of being exposed at the top of every cell -- where the registers allow: for d = 16 the
  // 29 doubles in flight push both sweeps into scratch and cost more than the round trip (measured: forward 1.45 -> 1.50 ms,
  // backward 0.62 -> 0.76 ms at B = 4096), so there the batch is loaded at the top of its own cell.
  constexpr bool PREFETCH = D <= 8;
  R nx_pmu_own, nx_prow[D], nx_Krow[NX], nx_alpha, nx_zt[LEANG ? 1 : NZ];  // (nx_zt: per-cell targets, generic variant only)
  unsigned nx_ff;
  const Window zw = make_window(z_per_cell ? a.z : a.x0, (z_per_cell ? (unsigned long)T * NZ : 1ul) * B * W);
  // byte offsets of this lane's prior rows inside a cell: the one piece of rank-dependent state that IS kept across the
  // time loop (29 dwords; recomputing them costs ~5 instructions per load in every cell)
  unsigned o_pmu, o_prow[D], o_K[NX];
  {
    const int r0 = g.r, rd0 = r0 < D ? r0 : D - 1, trd0 = rd0 * (rd0 + 1) / 2, ru0 = (r0 >= NX && r0 < D) ? r0 - NX : 0;
    o_pmu = (unsigned)rd0 * rbp + bop;
#pragma unroll
    for (int j = 0; j < D; ++j) o_prow[j] = (unsigned)(D + symidx(rd0, trd0, j)) * rbp + bop;
#pragma unroll
    for (int k = 0; k < NX; ++k) o_K[k] = (unsigned)(O_K + ru0 * NX + k) * rbp + bop;
  }
  auto fetch_prior = [&](const int tc, const unsigned rbx, const int, const int, const int) {
    const int trc = c.row(tc);  // row of the persistent buffers (ring, see Consts::t0)
    const GIO<R> pri = gio_post(a.prior + (unsigned long)trc * C::E_POST * B, C::E_POST, B, rbx, bop);
    nx_pmu_own = pri.ldo(o_pmu);
#pragma unroll
    for (int j = 0; j < D; ++j) nx_prow[j] = pri.ldo(o_prow[j]);
#pragma unroll
    for (int k = 0; k < NX; ++k) nx_Krow[k] = pri.ldo(o_K[k]);
    nx_alpha = LEANG ? alpha_traj : wld<R>(alw, 0u, alpha_cell ? (unsigned)(((unsigned long)trc * B + b) * W) : bo);
    nx_ff = wld_u8(ffw, (unsigned)trc);
    if (!LEANG && z_per_cell) {  // with the rows, a cell ahead: a load behind this branch in MID-cell costs a vmcnt(0) at its join
#pragma unroll
      for (int l = 0; l < NZ; ++l) nx_zt[LEANG ? 0 : l] = wld<R>(zw, 0u, (unsigned)((((unsigned long)trc * NZ + l) * B + b) * W));
    }
  };
  if (PREFETCH) {
    const int r0 = g.r, rd0 = r0 < D ? r0 : D - 1;
    fetch_prior(0, rb0, rd0, rd0 * (rd0 + 1) / 2, (r0 >= NX && r0 < D) ? r0 - NX : 0);
    // settled before the loop: loads pending on the loop-entry path cost an s_waitcnt vmcnt(0) at the top of every cell
    nx_pmu_own = opaque(nx_pmu_own), nx_alpha = opaque(nx_alpha), nx_ff = opaque(nx_ff);
#pragma unroll
    for (int j = 0; j < D; ++j) nx_prow[j] = opaque(nx_prow[j]);
#pragma unroll
    for (int k = 0; k < NX; ++k) nx_Krow[k] = opaque(nx_Krow[k]);
#pragma unroll
    for (int j = 0; j < NX; ++j) mu_x[j] = opaque(mu_x[j]), sx[j] = opaque(sx[j]);
  }

#if defined(I2C_GROUP_STAMPS) && !defined(I2C_HOST_SIM)
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < T; ++t) {
    // Nothing that depends on the rank or on the row stride may be hoisted out of the time loop (the byte offsets of ~100
    // rows and the batch constants of "my" row would be pinned in registers for the whole sweep): the cell works on an
    // opaque copy of both.
    Grp<R, G> g = g_in;
    g.r = opaque_i(g_in.r);
    const int r = g.r;
    const unsigned rb = opaque_uniform(rb0);
    const int rx = r < NX ? r : NX - 1, rd = r < D ? r : D - 1, rz = r < NZ ? r : NZ - 1, rt = r < NT ? r : NT - 1;
    const int trx = rx * (rx + 1) / 2, trd = rd * (rd + 1) / 2;
    const bool is_u = r >= NX && r < D;
    const int ru = is_u ? r - NX : 0;

    const int tr = c.row(t);
    const GIO<R> out = gio(a.fwd + (unsigned long)t * C::E_FWD * B, C::E_FWD, rb, bo);
    if (!PREFETCH) fetch_prior(t, rb, rd, trd, ru);  // nothing is carried across cells then
    R pmu[D], prow[D], Krow[NX];
    const R pmu_own = nx_pmu_own, alpha = nx_alpha;
    const unsigned ff_cur = nx_ff;
    R zt_cur[LEANG ? 1 : NZ];
    if (!LEANG) {
#pragma unroll
      for (int l = 0; l < NZ; ++l) zt_cur[l] = nx_zt[l];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) prow[j] = nx_prow[j];
#pragma unroll
    for (int k = 0; k < NX; ++k) Krow[k] = nx_Krow[k];
    g_gather<D>(g, 0, pmu_own, pmu);
    int cell_bad = 0;
    I2C_STAMP(0);  // loads + gather of the prior mean

    // ---- 1. joint prior over (x, u) ---------------------------------------------------
    R mu0[D], s0[D];
    if (ff_cur != 0) {  // feed-forward: independent action prior (i2c.py:355-360)
#pragma unroll
      for (int i = 0; i < D; ++i) mu0[i] = i < NX ? mu_x[i] : pmu[i];
#pragma unroll
      for (int j = 0; j < D; ++j) s0[j] = r < NX ? (j < NX ? sx[j < NX ? j : 0] : R(0)) : (j >= NX ? prow[j] : R(0));
    } else {  // feedback: condition the previous controller on the new state message (i2c.py:361-387)
      R Sr[NX], rinvS[NX], q[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        Sr[j] = prow[j] + sx[j];
        q[j] = mu_x[j] - pmu[j];
      }
      cell_bad = flag_stage(cell_bad, g_chol<NX>(g, 0, Sr, rinvS), 0);
      g_fsub<NX, 1>(g, 0, rinvS, q, (R*)nullptr);
      R maha = R(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) maha += q[i] * q[i];
      const R rho = r_exp(R(-0.5) * maha);
#pragma unroll
      for (int k = 0; k < NX; ++k) Krow[k] *= rho;
      g_joint<NX, NU>(g, mu_x, sx, Krow, prow, pmu, pmu + NX, true, false, true, mu0, s0);
    }
    I2C_STAMP(1);  // joint prior
    if (PREFETCH) fetch_prior(t + 1 < T ? t + 1 : t, rb, rd, trd, ru);  // this cell's rows are consumed: the next cell's, a cell ahead
    if (prior_out) {
      const GIO<R> po = gio(prior_out + (unsigned long)t * (D + sym(D)) * B, D + sym(D), rb, bo);
      po.st_if(r < D, r, g_sel<D>(mu0, r));
#pragma unroll
      for (int j = 0; j < D; ++j)
        po.st_if(r < D && j <= opaque_i(r), D + trd + j, s0[j]);
    }

    // ---- 2. cost "observation": measurement update on z (i2c.py:390-407) --------------
    R mu1_own;
    {
      R mz[NZ], szr[NZ], sxz[NZ];
      if (OBS_ID && c.rule_xu.unit) {  // z is a leading slice of (x, u): its moments are blocks of the prior joint
#pragma unroll
        for (int k = 0; k < NZ; ++k) {
          mz[k] = mu0[k < D ? k : 0];
          szr[k] = sxz[k] = s0[k < D ? k : 0];
        }
      } else {
        R L0[D], rinv0[D];
#pragma unroll
        for (int j = 0; j < D; ++j) L0[j] = s0[j];
        cell_bad = flag_stage(cell_bad, g_chol<D>(g, 0, L0, rinv0), 1);
        g_transform<M, ObsStruct<M>, D, NZ, true>(g, 0, 1, 2, c.rule_xu, mu0, L0, ObserveF<M, R>{c.params}, mz, szr, sxz);
      }
#pragma unroll
      for (int l = 0; l < NZ; ++l) {
        szr[l] += alpha * kc.sig_xi0[rz * NZ + l];
        mz[l] = ((!LEANG && z_per_cell) ? zt_cur[LEANG ? 0 : l] : c.zg[l]) - mz[l];  // the innovation
      }
      cell_bad = flag_stage(cell_bad, g_kalman<D, NZ>(g, mu0, s0, mz, szr, sxz, &mu1_own), 2);
    }
    I2C_STAMP(2);  // cost observation update
    // mu0 / s0 now hold mu_xu1_f / row r of sig_xu1_f
    out.st_if(r < D, r, mu1_own);
#pragma unroll
    for (int j = 0; j < D; ++j)
      out.st_if(r < D && j <= opaque_i(r), D + trd + j, s0[j]);

    // ---- 3. dynamics push-through (i2c.py:415-421) and smoother gain (i2c.py:423-425) --
    R L3[NX], rinv3[NX];
    {
      R L1[D], rinv1[D], sxy[NX];
#pragma unroll
      for (int j = 0; j < D; ++j) L1[j] = s0[j];
      cell_bad = flag_stage(cell_bad, g_chol<D>(g, 0, L1, rinv1), 3);
      I2C_STAMP(3);  // stores + chol(sig_xu1_f)
      g_transform<M, DenseStruct<D>, D, NX, true>(g, 0, 1, 2, c.rule_xu, mu0, L1, DynamicsF<M, R>{c.params}, mu_x, sx, sxy);
      I2C_STAMP(4);  // dynamics transform
#pragma unroll
      for (int l = 0; l < NX; ++l) {
        sx[l] += c.rule_xu.W * kc.sig_eta[rx * NX + l];  // sum_p w_p sig_eta (quadrature.py:57)
        L3[l] = sx[l];
      }
      cell_bad = flag_stage(cell_bad, g_chol<NX>(g, 0, L3, rinv3), 4);
      g_fsub<NX, 1>(g, 0, rinv3, sxy, (R*)nullptr);  // row r of J = sig_xy sig_x3^-1
      g_bsub<NX>(g, 0, rinv3, sxy);
#pragma unroll
      for (int l = 0; l < NX; ++l)
        out.st_if(r < D, O_J + rd * NX + l, sxy[l]);
    }

    // ---- 4. terminal cost observation on the flagged cell, after J (i2c.py:430-443) ----
    if (NZT > 0 && t == c.terminal_cell && c.has_Qf) {
      R mzt[NT], sztr[NT], sxzt[NT], own;
      if (TERM_ID && c.rule_x.unit) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          mzt[k] = mu_x[k < NX ? k : 0];
          sztr[k] = sxzt[k] = sx[k < NX ? k : 0];
        }
      } else {  // chol(sig_x3_f) is still in LDS matrix 0 and its row in L3
        g_transform<M, TermStruct<M>, NX, NT, true>(g, 0, 1, 2, c.rule_x, mu_x, L3, ObserveTermF<M, R>{c.params}, mzt, sztr, sxzt);
      }
#pragma unroll
      for (int l = 0; l < NT; ++l) {
        sztr[l] += alpha * kc.sig_xiT0[rt * NT + l];
        mzt[l] = c.zg_term[l] - mzt[l];
      }
      cell_bad = flag_stage(cell_bad, g_kalman<NX, NT>(g, mu_x, sx, mzt, sztr, sxzt, &own), 5);
    }
    fail = fold_cell_failure(fail, cell_bad, t);
    out.st_if(r < NX, O_MU3 + r, g_sel<NX>(mu_x, r));
#pragma unroll
    for (int j = 0; j < NX; ++j)
      out.st_if(r < NX && j <= opaque_i(r), O_S3 + trx + j, sx[j]);
    I2C_STAMP(5);  // chol(sig_x3_f), smoother gain, terminal update, stores
  }
#if defined(I2C_GROUP_STAMPS) && !defined(I2C_HOST_SIM)
  if (b == 0 && g_in.r == 0)
    printf("group forward stamps (clocks per cell): loads %llu prior %llu observe %llu chol1 %llu dynamics %llu gain+stores %llu\n",
           stamp_acc[0] / T, stamp_acc[1] / T, stamp_acc[2] / T, stamp_acc[3] / T, stamp_acc[4] / T, stamp_acc[5] / T);
#endif
  if (g_in.r == 0 && fail != 0 && a.status[b] == 0) a.status[b] = fail;
}

// ------------------------------------------------------------------------------------------
// Backward sweep (i2c.py:882-886 over :544-610), fused form: the group walks T-1..0 doing the whole cell -- RTS update of
// the joint, posterior observation moments and their expected cost, controller from the factor of the posterior joint.
// With a terminal state prior (covariance control, i2c.py:548-559) the chain starts from its product with the filtered state.
// ------------------------------------------------------------------------------------------
template <class M, typename R, int G, bool FULLW, class KC>
I2C_HD inline void backward_group_body(const Consts<M, R>& c, const KC& kc, const CellArgs<R>& a, const int b,
                                       const Grp<R, G>& g_in) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D, NT = C::NZT1, LD = Grp<R, G>::LD;
  static_assert(G >= D && G >= NZ && G >= NT, "one matrix row per lane");
  constexpr int O_K = D + sym(D), O_k = O_K + NU * NX, O_SK = O_k + NU;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  constexpr bool OBS_ID = st_identity<ObsStruct<M>, NZ>() && NZ <= D;
  constexpr bool TERM_ID = st_identity<TermStruct<M>, NT>() && NT <= NX;
  const Grp<R, G>& g = g_in;
  const int r = g.r;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned W = sizeof(R), bo = (unsigned)b * W, rb0 = (unsigned)(B * W);
  const int rx = r < NX ? r : NX - 1, rd = r < D ? r : D - 1;
  const int trx = rx * (rx + 1) / 2, trd = rd * (rd + 1) / 2, trr = r * (r + 1) / 2;
  const bool is_x = r < NX, is_u = r >= NX && r < D;
  const int ru = is_u ? r - NX : 0;

  // end of the chain (i2c.py:546-564): the smoothed terminal state is the filtered one
  R m3m[NX], s3m[NX];
  {
    const GIO<R> fw = gio(a.fwd + (unsigned long)(T - 1) * C::E_FWD * B, C::E_FWD, rb0, bo);
    g_gather<NX>(g, 0, fw.ld(O_MU3 + rx), m3m);
#pragma unroll
    for (int j = 0; j < NX; ++j) s3m[j] = fw.ld(O_S3 + symidx(rx, trx, j));
  }
  if (c.has_x_terminal) {
    const R tmp = a.temp[b];
    const bool ok = g_end_of_chain<M, R, G>(g, c, kc, tmp, m3m, s3m);
    if (r == 0) {
      a.temp[b] = tmp + c.dtemp;  // every lane has read it (the gather inside is a group-wide synchronisation)
      if (!ok) set_status(a.status, b, 6, T - 1);
    }
  }
  // terminal observation statistics (i2c.py:567-570, 989-992): tr(Qf (errT errT^T + sig_z3_m))
  R trT = R(0);
  if (NZT > 0 && c.has_Qf) {
    R mzt[NT], sztr[NT];
    if (TERM_ID && c.rule_x.unit) {
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        mzt[k] = m3m[k < NX ? k : 0];
        sztr[k] = s3m[k < NX ? k : 0];
      }
    } else {
      R L3[NX], rinv3[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) L3[j] = s3m[j];
      if (!g_chol<NX>(g, 0, L3, rinv3) && r == 0) set_status(a.status, b, 6, T - 1);
      g_transform<M, TermStruct<M>, NX, NT, false>(g, 0, 1, 2, c.rule_x, m3m, L3, ObserveTermF<M, R>{c.params}, mzt, sztr, (R*)nullptr);
    }
    R tv;
    // FULLW: the instantiation for non-diagonal weights (a compile-time variant: in one kernel the general form's live state
    // pushed the d = 16 backward sweep into scratch)
    if constexpr (FULLW) g_cost_full<NT>(g, kc.qf, mzt, sztr, c.zg_term, &trT, &tv);
    else g_cost<NT>(g, kc.qf_d, mzt, sztr, c.zg_term, &trT, &tv);
    if (r < NT) a.term_stats[(long)(3 + r) * B + b] = g_sel<NT>(mzt, r);
#pragma unroll
    for (int l = 0; l < NT; ++l)
      if (r < NT && l <= opaque_i(r)) a.term_stats[(long)(3 + NT + trr + l) * B + b] = sztr[l];
  }
  if (r == 0) a.term_stats[b] = trT;

  R sum_m = R(0), sum_v = R(0);
  // the forward rows of a cell are fetched one cell ahead where the registers allow (see forward_group_body)
  constexpr bool PREFETCH = D <= 8;
  R nx_mu1_own, nx_m3f_own, nx_S[D], nx_s3f[NX], nx_Jr[NX];
  // byte offsets of this lane's forward rows inside a cell, kept across the time loop (see forward_group_body)
  unsigned o_mu1, o_m3f, o_S[D], o_s3f[NX], o_J[NX];
  o_mu1 = (unsigned)rd * rb0 + bo;
  o_m3f = (unsigned)(O_MU3 + rx) * rb0 + bo;
#pragma unroll
  for (int j = 0; j < D; ++j) o_S[j] = (unsigned)(D + symidx(rd, trd, j)) * rb0 + bo;
#pragma unroll
  for (int j = 0; j < NX; ++j) o_s3f[j] = (unsigned)(O_S3 + symidx(rx, trx, j)) * rb0 + bo;
#pragma unroll
  for (int l = 0; l < NX; ++l) o_J[l] = (unsigned)(O_J + rd * NX + l) * rb0 + bo;
  auto fetch_fwd = [&](const int tc, const unsigned rbx, const int, const int, const int, const int) {
    const GIO<R> fw = gio(a.fwd + (unsigned long)tc * C::E_FWD * B, C::E_FWD, rbx, bo);
    nx_mu1_own = fw.ldo(o_mu1);
    nx_m3f_own = fw.ldo(o_m3f);
#pragma unroll
    for (int j = 0; j < D; ++j) nx_S[j] = fw.ldo(o_S[j]);
#pragma unroll
    for (int j = 0; j < NX; ++j) nx_s3f[j] = fw.ldo(o_s3f[j]);
#pragma unroll
    for (int l = 0; l < NX; ++l) nx_Jr[l] = fw.ldo(o_J[l]);
  };
  if (PREFETCH) fetch_fwd(T - 1, rb0, rd, trd, rx, trx);
  for (int t = T - 1; t >= 0; --t) {
    // Nothing that depends on the rank or on the row stride may be hoisted out of the time loop (the byte offsets of ~100
    // rows and the batch constants of "my" row would be pinned in registers for the whole sweep): the cell works on an
    // opaque copy of both.
    Grp<R, G> g = g_in;
    g.r = opaque_i(g_in.r);
    const int r = g.r;
    const unsigned rb = opaque_uniform(rb0);
    const int rx = r < NX ? r : NX - 1, rd = r < D ? r : D - 1;
    const int trx = rx * (rx + 1) / 2, trd = rd * (rd + 1) / 2, trr = r * (r + 1) / 2;
    const bool is_x = r < NX, is_u = r >= NX && r < D;
    const int ru = is_u ? r - NX : 0;

    const GIO<R> po = gio_post(a.post + (unsigned long)c.row(t) * C::E_POST * B, C::E_POST, B, c.post_tm ? W : rb,
                               c.post_tm ? (unsigned)b * C::E_POST * W : bo);
    if (!PREFETCH) fetch_fwd(t, rb, rd, trd, rx, trx);  // nothing is carried across cells then
    R mu[D], S[D], m3f[NX], s3f[NX], Jr[NX];
    const R mu1_own = nx_mu1_own;
    g_gather<NX>(g, 1, nx_m3f_own, m3f);
#pragma unroll
    for (int j = 0; j < D; ++j) S[j] = nx_S[j];
#pragma unroll
    for (int j = 0; j < NX; ++j) s3f[j] = nx_s3f[j];
#pragma unroll
    for (int l = 0; l < NX; ++l) Jr[l] = nx_Jr[l];
    if (a.xm) {
      R* xo = const_cast<R*>(a.xm) + ((long)t * C::E_XM) * B + b;
      if (is_x) xo[(long)r * B] = g_sel<NX>(m3m, r);
#pragma unroll
      for (int j = 0; j < NX; ++j)
        if (is_x && j <= opaque_i(r)) xo[(long)(NX + trx + j) * B] = s3m[j];
    }
    R zt[NZ];
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = c.z_per_cell ? a.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k];

    // RTS update of the joint (i2c.py:580-583): mu += J (m3m - m3f), S += J (S3m - S3f) J^T
    // J is published TRANSPOSED (row l of the LDS matrix = column l of J): both products then read contiguous rows.
    const auto dSm = g.mat(1), Jt = g.mat(2);
    g.sync();
    if (is_x) {
#pragma unroll
      for (int j = 0; j < NX; ++j) dSm[r * LD + j] = s3m[j] - s3f[j];
    }
#pragma unroll
    for (int l = 0; l < NX; ++l) Jt[l * LD + r] = Jr[l];  // lanes r >= D publish junk columns nobody reads
    R mu_own = mu1_own;
#pragma unroll
    for (int l = 0; l < NX; ++l) mu_own += Jr[l] * (m3m[l] - m3f[l]);
    g_gather<D>(g, 0, mu_own, mu);  // its syncs also publish dS and J
    {  // both products as walks over the state index (g_walk): JD = J_r dS, then S_r += JD J^T with JD_r parked in LDS
      R JD[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) JD[k] = R(0);
      struct P1 {
        R row[NX], own;
      };
      g_walk<NX, (D * NX >= 96), P1>(
          [&](const int l, P1& p) {
            p.own = Jt[l * LD + r];
#pragma unroll
            for (int k = 0; k < NX; ++k) p.row[k] = dSm[l * LD + k];
          },
          [&](const P1& p) {
#pragma unroll
            for (int k = 0; k < NX; ++k) JD[k] += p.own * p.row[k];
          });
      const auto JDm = g.mat(0);  // free until the factorisation below
      g.sync();
#pragma unroll
      for (int k = 0; k < NX; ++k) JDm[r * LD + k] = JD[k];
      g.sync();
      struct P2 {
        R col[D], own;
      };
      g_walk<NX, (D * NX >= 96), P2>(
          [&](const int k, P2& p) {
            p.own = JDm[r * LD + k];
#pragma unroll
            for (int j = 0; j < D; ++j) p.col[j] = Jt[k * LD + j];
          },
          [&](const P2& p) {
#pragma unroll
            for (int j = 0; j < D; ++j) S[j] += p.own * p.col[j];
          });
    }
    if (PREFETCH) fetch_fwd(t > 0 ? t - 1 : 0, rb, rd, trd, rx, trx);  // this cell's forward rows are consumed: the next cell's, a cell ahead
    R Lm[D], rinv[D];
#pragma unroll
    for (int j = 0; j < D; ++j) Lm[j] = S[j];
    if (!g_chol<D>(g, 0, Lm, rinv) && r == 0) set_status(a.status, b, 7, t);

    // controller from the factor (i2c.py:600-608): K L_xx = L_ux, sigK = L_uu L_uu^T, k = mu_u - K mu_x
    // (before the transform, which may overwrite LDS matrices 1 and 2 but leaves L in matrix 0 ... and reads it)
    R ctl[NX], sigK[NU];
    {
      const auto L = g.mat(0);
#pragma unroll
      for (int k = 0; k < NX; ++k) ctl[k] = Lm[k];
      g_bsub<NX>(g, 0, rinv, ctl);
#pragma unroll
      for (int q = 0; q < NU; ++q) {
        R v = R(0);
#pragma unroll
        for (int k = 0; k <= q; ++k) v += Lm[NX + k] * L[(NX + q) * LD + NX + k];
        sigK[q] = v;
      }
    }
    // posterior observation moments (i2c.py:594-596) and their expected cost (i2c.py:1034-1043)
    R mz[NZ], szr[NZ], cm, cv;
    if (OBS_ID && c.rule_xu.unit) {
#pragma unroll
      for (int k = 0; k < NZ; ++k) {
        mz[k] = mu[k < D ? k : 0];
        szr[k] = S[k < D ? k : 0];
      }
    } else {
      g_transform<M, ObsStruct<M>, D, NZ, false>(g, 0, 1, 2, c.rule_xu, mu, Lm, ObserveF<M, R>{c.params}, mz, szr, (R*)nullptr);
    }
    if constexpr (FULLW) g_cost_full<NZ>(g, kc.qr, mz, szr, zt, &cm, &cv);
    else g_cost<NZ>(g, kc.qr_d, mz, szr, zt, &cm, &cv);
    sum_m += cm;
    sum_v += cv;

    po.st_if(r < D, r, mu_own);
#pragma unroll
    for (int j = 0; j < D; ++j)
      po.st_if(r < D && j <= opaque_i(r), D + trd + j, S[j]);
    if (is_u) {
      R kk = mu_own;
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        po.st(O_K + ru * NX + k, ctl[k]);
        kk -= ctl[k] * mu[k];
      }
      po.st(O_k + ru, kk);
#pragma unroll
      for (int q = 0; q < NU; ++q)
        po.st_if(q <= opaque_i(ru), O_SK + ru * (ru + 1) / 2 + q, sigK[q]);
    }
    if (a.zpost) {
      R* zo = a.zpost + ((long)t * C::E_ZPOST) * B + b;
      if (r < NZ) zo[(long)r * B] = g_sel<NZ>(mz, r);
#pragma unroll
      for (int l = 0; l < NZ; ++l)
        if (r < NZ && l <= opaque_i(r)) zo[(long)(NZ + trr + l) * B] = szr[l];
    }
    if (a.cell_stats && r == 0) {
      a.cell_stats[((long)t * 2 + 0) * B + b] = cm;
      a.cell_stats[((long)t * 2 + 1) * B + b] = cv;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      m3m[i] = mu[i];
      s3m[i] = S[i];  // state lanes: row r of the xx block
    }
  }
  if (r == 0) {
    a.term_stats[B + b] = sum_m;
    a.term_stats[2 * B + b] = sum_v;
  }
}

// ------------------------------------------------------------------------------------------
// Closed-loop propagation (i2c.py:150-199, 1247-1251) in the group form.
// ------------------------------------------------------------------------------------------
template <class M, typename R, int G, bool FULLW, class KC>
I2C_HD inline void propagate_group_body(const Consts<M, R>& c, const KC& kc, const PropArgs<R>& a, const int b,
                                        const Grp<R, G>& g_in) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D;
  static_assert(G >= D && G >= NZ, "one matrix row per lane");
  constexpr int O_K = D + sym(D), O_X3 = D + sym(D), O_SX3 = O_X3 + NX;
  constexpr bool OBS_ID = st_identity<ObsStruct<M>, NZ>() && NZ <= D;
  const Grp<R, G>& g = g_in;
  const int r = g.r;
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned W = sizeof(R), bo = (unsigned)b * W, rb0 = (unsigned)(B * W);
  const int rx = r < NX ? r : NX - 1, rd = r < D ? r : D - 1;
  const int trx = rx * (rx + 1) / 2, trd = rd * (rd + 1) / 2;
  const bool is_u = r >= NX && r < D;
  const int ru = is_u ? r - NX : 0;

  R mu_x[NX], sx[NX];
  g_gather<NX>(g, 0, a.x0[(long)rx * B + b], mu_x);
#pragma unroll
  for (int j = 0; j < NX; ++j) sx[j] = a.sig_x0[(long)symidx(rx, trx, j) * B + b];
  R sum_m = R(0), sum_v = R(0);

  for (int t = 0; t < T; ++t) {
    // Nothing that depends on the rank or on the row stride may be hoisted out of the time loop (the byte offsets of ~100
    // rows and the batch constants of "my" row would be pinned in registers for the whole sweep): the cell works on an
    // opaque copy of both.
    Grp<R, G> g = g_in;
    g.r = opaque_i(g_in.r);
    const int r = g.r;
    const unsigned rb = opaque_uniform(rb0);
    const int rx = r < NX ? r : NX - 1, rd = r < D ? r : D - 1;
    const int trx = rx * (rx + 1) / 2, trd = rd * (rd + 1) / 2;
    const bool is_u = r >= NX && r < D;
    const int ru = is_u ? r - NX : 0;

    const GIO<R> pri = gio_post(a.post + (unsigned long)c.row(t) * C::E_POST * B, C::E_POST, B, c.post_tm ? W : rb,
                                c.post_tm ? (unsigned)b * C::E_POST * W : bo);
    const GIO<R> out = gio(a.prop + (unsigned long)t * C::E_PROP * B, C::E_PROP, rb, bo);
    R qmu[D], prow[D], Krow[NX];
    g_gather<D>(g, 0, pri.ld(rd), qmu);
#pragma unroll
    for (int j = 0; j < D; ++j) prow[j] = pri.ld(D + symidx(rd, trd, j));
#pragma unroll
    for (int k = 0; k < NX; ++k) Krow[k] = pri.ld(O_K + ru * NX + k);
    const bool ff = a.ff[c.row(t)] != 0;
    if (!ff && (a.expert ? a.expert[c.row(t)] != 0 : c.use_expert != 0)) {  // i2c.py:160-167
      R Sr[NX], rinvS[NX], q[NX];
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        Sr[j] = prow[j] + sx[j];
        q[j] = mu_x[j] - qmu[j];
      }
      const bool ok = g_chol<NX>(g, 0, Sr, rinvS);
      g_fsub<NX, 1>(g, 0, rinvS, q, (R*)nullptr);
      R maha = R(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) maha += q[i] * q[i];
      const R rho = ok ? r_exp(R(-0.5) * maha) : R(1);  // the reference logs the exception and keeps K unscaled
#pragma unroll
      for (int k = 0; k < NX; ++k) Krow[k] *= rho;
    }
    // feed-forward cells use the action marginal but the joint still carries K sig_x (i2c.py:155-157, 173-179)
    R mu0[D], s0[D], qx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) qx[i] = ff ? mu_x[i] : qmu[i];
    g_joint<NX, NU>(g, mu_x, sx, Krow, prow, qx, qmu + NX, false, !ff, !ff, mu0, s0);
    out.st_if(r < D, r, g_sel<D>(mu0, r));
#pragma unroll
    for (int j = 0; j < D; ++j)
      out.st_if(r < D && j <= opaque_i(r), D + trd + j, s0[j]);

    R L0[D], rinv0[D];
#pragma unroll
    for (int j = 0; j < D; ++j) L0[j] = s0[j];
    if (!g_chol<D>(g, 0, L0, rinv0) && r == 0) set_status(a.status, b, 8, t);
    R zt[NZ], mz[NZ], szr[NZ], cm, cv;
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = c.z_per_cell ? a.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k];
    if (OBS_ID && c.rule_xu.unit) {
#pragma unroll
      for (int k = 0; k < NZ; ++k) {
        mz[k] = mu0[k < D ? k : 0];
        szr[k] = s0[k < D ? k : 0];
      }
    } else {
      // writes LDS matrices 1 and 2 only: the factor in matrix 0 stays for the dynamics transform below
      g_transform<M, ObsStruct<M>, D, NZ, false>(g, 0, 1, 2, c.rule_xu, mu0, L0, ObserveF<M, R>{c.params}, mz, szr, (R*)nullptr);
    }
    if constexpr (FULLW) g_cost_full<NZ>(g, kc.qr, mz, szr, zt, &cm, &cv);
    else g_cost<NZ>(g, kc.qr_d, mz, szr, zt, &cm, &cv);
    sum_m += cm;
    sum_v += cv;

    g_transform<M, DenseStruct<D>, D, NX, false>(g, 0, 1, 2, c.rule_xu, mu0, L0, DynamicsF<M, R>{c.params}, mu_x, sx, (R*)nullptr);
#pragma unroll
    for (int l = 0; l < NX; ++l) sx[l] += c.rule_xu.W * kc.sig_eta[rx * NX + l];
    out.st_if(r < NX, O_X3 + r, g_sel<NX>(mu_x, r));
#pragma unroll
    for (int j = 0; j < NX; ++j)
      out.st_if(r < NX && j <= opaque_i(r), O_SX3 + trx + j, sx[j]);
  }
  R kl = R(0);
  if (c.has_x_terminal) {
    bool ok;
    kl = g_terminal_kl<M, R, G>(g_in, c, kc, mu_x, sx, &ok);
    if (!ok && g_in.r == 0) set_status(a.status, b, 8, T - 1);
  }
  if (r == 0) {
    a.prop_stats[b] = sum_m;
    a.prop_stats[B + b] = sum_v;
    a.prop_stats[2 * B + b] = kl;
  }
}

// ------------------------------------------------------------------------------------------
// Cubature Kalman filter step of the MPC state estimator (mpc.py:125-145) in the group form.
// ------------------------------------------------------------------------------------------
template <class M, typename R, int G, class KC>
I2C_HD inline void ckf_group_body(const Consts<M, R>& c, const KC& kc, const CkfArgs<R>& a, const int b,
                                  const Grp<R, G>& g) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NY = M::NY;
  static_assert(G >= NX && G >= NY, "one matrix row per lane");
  constexpr bool MEAS_ID = st_identity<MeasStruct<M>, NY>() && NY <= NX;
  const int r = g.r;
  const long B = c.B;
  const int rx = r < NX ? r : NX - 1, ry = r < NY ? r : NY - 1;
  const int trx = rx * (rx + 1) / 2;
  R mu[NX], S[NX], L[NX], rinv[NX], u[NU], y[NY];
  g_gather<NX>(g, 0, a.mu[(long)rx * B + b], mu);
#pragma unroll
  for (int j = 0; j < NX; ++j) L[j] = S[j] = a.cov[(long)symidx(rx, trx, j) * B + b];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = a.u[(long)i * B + b];
#pragma unroll
  for (int i = 0; i < NY; ++i) y[i] = a.y[(long)i * B + b];
  bool ok = g_chol<NX>(g, 0, L, rinv);
  // prediction (mpc.py:129-137): x-only sigma points, the action is appended unchanged
  R mf[NX], Sf[NX];
  g_transform<M, DenseStruct<NX>, NX, NX, false>(g, 0, 1, 2, c.rule_x, mu, L, DynamicsFixedUF<M, R>{c.params, u}, mf, Sf, (R*)nullptr);
#pragma unroll
  for (int l = 0; l < NX; ++l) L[l] = Sf[l] = Sf[l] + c.rule_x.W * kc.sig_eta[rx * NX + l];
  // innovation (mpc.py:139-145)
  R my[NY], Syr[NY], Sxy[NY], own;
  if (MEAS_ID && c.rule_x.unit) {
#pragma unroll
    for (int k = 0; k < NY; ++k) {
      my[k] = mf[k < NX ? k : 0];
      Syr[k] = Sxy[k] = Sf[k < NX ? k : 0];
    }
  } else {
    ok = g_chol<NX>(g, 0, L, rinv) && ok;
    g_transform<M, MeasStruct<M>, NX, NY, true>(g, 0, 1, 2, c.rule_x, mf, L, MeasureF<M, R>{c.params}, my, Syr, Sxy);
  }
#pragma unroll
  for (int l = 0; l < NY; ++l) {
    Syr[l] += kc.sig_zeta[ry * NY + l];
    my[l] = y[l] - my[l];
  }
  ok = g_kalman<NX, NY>(g, mf, Sf, my, Syr, Sxy, &own) && ok;
  if (r < NX) a.mu[(long)r * B + b] = own;
#pragma unroll
  for (int j = 0; j < NX; ++j)
    if (r < NX && j <= opaque_i(r)) a.cov[(long)(trx + j) * B + b] = Sf[j];
  if (!ok && r == 0) set_status(a.status, b, 9, 0);
}

}  // namespace i2c
