// Per-trajectory cell math of the Gaussian i2c cubature path, one trajectory per lane.
//
// Mathematically identical to the reference's I2cCell (i2c/i2c.py:350-447, 544-610, 150-199)
// and QuadratureInference (i2c/inference/quadrature.py:15-58), but arranged for registers:
//   * sigma points are generated column by column from the Cholesky factor and consumed
//     immediately (pairwise +/- differences), so only O(d * ny) accumulators are live;
//   * covariances are accumulated in the centred, shift-by-centre-point form (no
//     sum w y y^T - m m^T cancellation); the general-weight correction terms
//     reproduce the reference's formula exactly, including its use of w_sig for the mean;
//   * the Kalman-style update uses only forward substitutions with chol(sig_z):
//        V = C^{-1} sig_xz^T,  mu += V^T C^{-1} r,  sig -= V^T V   (symmetric by construction);
//   * the controller (K, k, sigK) falls out of the Cholesky factor of the posterior joint.
#pragma once
#include <stdint.h>
#include "../../include/i2c_hip.h"
#include "i2c_linalg.hpp"
#include "i2c_models.hpp"

namespace i2c {

// Sigma-point rule for one input dimension (i2c/exp_types.py:36-49), precomputed on the host.
template <typename R> struct Rule {
  R sf;   // sqrt(dim + lam)
  R w0;   // weights_sig[0]
  R wi;   // weights_sig[1:]
  R W;    // sum of weights_sig (= 1 unless 1 - alpha^2 + beta != 0)
  int unit;  // W == 1 exactly: skip the correction terms
  // Gauss-Hermite tensor grid (i2c/exp_types.py:52-68), used by the GRID variants only: gh_degree^dim points
  // m + sqrt(2) L xi, xi in gh_x^dim, weight = prod gh_w (the 1-D weights already divided by sqrt(pi))
  int gh_degree, gh_points;
  R gh_x[I2C_MAX_GH_DEGREE], gh_w[I2C_MAX_GH_DEGREE];
};

template <class M, typename R> struct Consts {
  static constexpr int NX = M::NX, NU = M::NU, NZ = M::NZ, NZT = M::NZT, D = NX + NU;
  static constexpr int NZT1 = NZT > 0 ? NZT : 1, NP1 = M::NP > 0 ? M::NP : 1;
  static constexpr int E_PRI = D + sym(D) + NU * NX;                        // rows the forward reads
  static constexpr int E_POST = E_PRI + NU + sym(NU);
  static constexpr int E_FWD = D + sym(D) + NX + sym(NX) + D * NX;
  static constexpr int E_XM = NX + sym(NX);
  static constexpr int E_ZPOST = NZ + sym(NZ);
  static constexpr int E_PROP = D + sym(D) + NX + sym(NX);
  static constexpr int E_TERM = 4 + NZT + sym(NZT);  // last row: plan-cost sum of the Linearize path
  int B, T;
  // Ring offset of the PERSISTENT per-cell buffers (prior/post, z, alpha_cell, feedforward): cell t lives in row
  // (t0 + t) mod T. 0 outside the MPC loop; the receding-horizon shift (mpc.py:174-181) advances it by one instead of
  // moving every row. Scratch buffers of a sweep (fwd, xm, zpost, prior_out, prop, cell_stats) are indexed by t directly.
  int t0;
  I2C_HD inline int row(const int t) const {
    const int r = t + t0;
    return r >= T ? r - T : r;
  }
  int has_Qf, has_x_terminal, z_per_cell, use_expert, terminal_cell, inference;
  int qr_diag, qf_diag;  // cost weights are diagonal: cheap closed forms in gaussian_cost
  int post_tm;           // posterior / prior buffers are trajectory-major, [T][B][E_POST] (I2cProblem.post_layout; wave-capable models)
  int fwd_tm;            // quad forward kernel only: write the forward messages trajectory-major, [T][B][E_FWD] (the wave kernels read them)
  // element e of trajectory b inside one cell block of the posterior / prior buffer: stride of e and offset of b
  I2C_HD inline long post_es() const { return post_tm ? 1L : (long)B; }
  I2C_HD inline long post_bo(const int b) const { return post_tm ? (long)b * E_POST : (long)b; }
  Rule<R> rule_xu, rule_x;
  R dtemp, tol;
  R sig_eta_w[sym(NX)];  // W(d) * sig_eta, W = sum of the d-dimensional rule's weights
  R sig_eta[sym(NX)], sig_xi0[sym(NZ)], QR[sym(NZ)], sig_xiT0[sym(NZT1)], Qf[sym(NZT1)];
  R zg[NZ], zg_term[NZT1], mu_x_term[NX], sig_x_term[sym(NX)], params[NP1];
};

// The three model callbacks as functors: f(x, sin(angles of x), cos(angles of x), y).
template <class M, typename R> struct ObserveF {
  const R* p;
  I2C_HD inline void operator()(const R* x, const R* sn, const R* cs, R* y) const { M::observe(p, x, sn, cs, y); }
};
template <class M, typename R> struct DynamicsF {
  const R* p;
  I2C_HD inline void operator()(const R* x, const R* sn, const R* cs, R* y) const { M::dynamics(p, x, sn, cs, y); }
};
template <class M, typename R> struct MeasureF {
  const R* p;
  I2C_HD inline void operator()(const R* x, const R* sn, const R* cs, R* y) const { M::measure(p, x, sn, cs, y); }
};
// dynamics as a function of the state only, with a known action appended (CKF prediction, mpc.py:129-131)
template <class M, typename R> struct DynamicsFixedUF {
  const R* p;
  const R* u;
  I2C_HD inline void operator()(const R* x, const R* sn, const R* cs, R* y) const {
    R xu[M::NX + M::NU];
#pragma unroll
    for (int i = 0; i < M::NX; ++i) xu[i] = x[i];
#pragma unroll
    for (int i = 0; i < M::NU; ++i) xu[M::NX + i] = u[i];
    M::dynamics(p, xu, sn, cs, y);
  }
};
template <class M, typename R> struct ObserveTermF {
  const R* p;
  I2C_HD inline void operator()(const R* x, const R* sn, const R* cs, R* y) const {
    M::observe_terminal(p, x, sn, cs, y);
  }
};

// Compile-time structure of a model callback, used to skip sigma-point work that is exactly
// known: output k is either a pass-through of input lin(k) (>= 0) or nonlinear (lin(k) < 0) and
// then depends on no input with index above dep(k). (Column j of the cubature points only moves
// inputs i >= j because the Cholesky factor is lower triangular.)
template <class M> struct ObsStruct {
  I2C_HD static constexpr int lin(int k) { return M::obs_lin(k); }
  I2C_HD static constexpr int dep(int k) { return M::obs_dep(k); }
};
template <class M> struct TermStruct {
  I2C_HD static constexpr int lin(int k) { return M::term_lin(k); }
  I2C_HD static constexpr int dep(int k) { return M::term_dep(k); }
};
template <class M> struct MeasStruct {
  I2C_HD static constexpr int lin(int k) { return M::meas_lin(k); }
  I2C_HD static constexpr int dep(int k) { return M::meas_dep(k); }
};
template <int DIN> struct DenseStruct {
  I2C_HD static constexpr int lin(int) { return -1; }
  I2C_HD static constexpr int dep(int) { return DIN - 1; }
};

// Gaussian push-through  N(m, Sin = L L^T) -> (my, Sy [, Sxy])   (quadrature.py:27-58).
//   L     packed lower Cholesky factor of the input covariance Sin (DIN)
//   my    [DOUT]   weighted mean (uses weights_sig, as the reference does)
//   Sy    [sym DOUT] covariance, equal to  sum_p w_p y_p y_p^T - my my^T
//   Sxy   [DIN * DOUT] row-major cross-covariance = sum_p w_p x_p y_p^T - m my^T  (only if CROSS)
//
// With pairwise sums a_j = (y_j+ - y0) + (y_j- - y0) and differences dl_j = y_j+ - y_j-:
//   my = W y0 + wi sum_j a_j,   Sy = wi/2 sum_j (a_j a_j^T + dl_j dl_j^T) - wi^2 A A^T (+ terms in 1-W),
//   Sxy = wi sf L [dl_0 .. dl_{d-1}]^T.
// For a pass-through output a_j = 0 and dl_j = 2 sf L[i][j] exactly, so its moments are rows of
// Sin (2 wi sf^2 = 1 for every alpha, beta, kappa) and cost nothing.
// Trig of the model's angle coordinates is computed once at the mean and once per (angle, column)
// offset d = sf L[i][j], then rotated: sin(m +/- d) = s0 cd +/- c0 sd, cos(m +/- d) = c0 cd -/+ s0 sd.
template <class M, class ST, int DIN, int DOUT, bool CROSS, bool UNITW = false, typename R, class F>
I2C_FN void sp_transform(const Rule<R>& rule, const R* m, const R* Sin, const R* L, const F& f, R* my, R* Sy,
                         R* Sxy, const PolyTab<R>* tab = nullptr) {
  constexpr int NA = M::NA, NA1 = NA > 0 ? NA : 1;
  R s0[NA1], c0[NA1];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if (tab)
      r_sincos(m[M::ang(a)], *tab, &s0[a], &c0[a]);
    else
      r_sincos(m[M::ang(a)], &s0[a], &c0[a]);
  }
  R y0[DOUT];
  f(m, s0, c0, y0);
  R A[DOUT];
#pragma unroll
  for (int k = 0; k < DOUT; ++k) A[k] = R(0);
#pragma unroll
  for (int k = 0; k < sym(DOUT); ++k) Sy[k] = R(0);
  if (CROSS) {
#pragma unroll
    for (int k = 0; k < DIN * DOUT; ++k) Sxy[k] = R(0);
  }
#pragma unroll
  for (int j = 0; j < DIN; ++j) {
    // does any nonlinear output move with this column?
    bool any = false;
#pragma unroll
    for (int k = 0; k < DOUT; ++k) any = any || (ST::lin(k) < 0 && ST::dep(k) >= j);
    R a[DOUT], dl[DOUT];
    if (any) {
      // points m +/- sf L[:, j]; rows above the diagonal of L are structurally zero
      R xp[DIN], xm[DIN];
#pragma unroll
      for (int i = 0; i < DIN; ++i) {
        if (i < j) {
          xp[i] = m[i];
          xm[i] = m[i];
        } else {
          const R d = rule.sf * L[tri(i, j)];
          xp[i] = m[i] + d;
          xm[i] = m[i] - d;
        }
      }
      R sp[NA1], cp[NA1], sm[NA1], cm[NA1];
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        if (M::ang(q) < j) {  // this column does not move the angle
          sp[q] = sm[q] = s0[q];
          cp[q] = cm[q] = c0[q];
        } else {
          R sd, cd;
          if (tab)
            r_sincos(rule.sf * L[tri(M::ang(q), j)], *tab, &sd, &cd);
          else
            r_sincos_small(rule.sf * L[tri(M::ang(q), j)], &sd, &cd);
          sp[q] = s0[q] * cd + c0[q] * sd;
          cp[q] = c0[q] * cd - s0[q] * sd;
          sm[q] = s0[q] * cd - c0[q] * sd;
          cm[q] = c0[q] * cd + s0[q] * sd;
        }
      }
      R yp[DOUT], ym[DOUT];
      f(xp, sp, cp, yp);
      f(xm, sm, cm, ym);
#pragma unroll
      for (int k = 0; k < DOUT; ++k) {
        a[k] = (yp[k] - y0[k]) + (ym[k] - y0[k]);
        dl[k] = yp[k] - ym[k];
      }
    }
    // per-output activity in this column (all compile-time after unrolling)
#pragma unroll
    for (int k = 0; k < DOUT; ++k) {
      const bool nl_k = ST::lin(k) < 0 && ST::dep(k) >= j;
      if (ST::lin(k) >= j) dl[k] = R(2) * rule.sf * L[tri(ST::lin(k) >= j ? ST::lin(k) : j, j)];
      if (nl_k) A[k] += a[k];
#pragma unroll
      for (int l = 0; l <= k; ++l) {
        const bool nl_l = ST::lin(l) < 0 && ST::dep(l) >= j;
        const bool d_k = nl_k || ST::lin(k) >= j, d_l = nl_l || ST::lin(l) >= j;
        const bool both_lin = ST::lin(k) >= 0 && ST::lin(l) >= 0;
        if (nl_k && nl_l) Sy[tri(k, l)] += a[k] * a[l];
        if (d_k && d_l && !both_lin) Sy[tri(k, l)] += dl[k] * dl[l];
      }
      if (CROSS && nl_k) {
#pragma unroll
        for (int i = j; i < DIN; ++i) Sxy[i * DOUT + k] += L[tri(i, j)] * dl[k];
      }
    }
    sched_fence<(DIN >= 6)>();
  }
  const R hw = R(0.5) * rule.wi, w2 = rule.wi * rule.wi, cs = rule.wi * rule.sf;
#pragma unroll
  for (int k = 0; k < DOUT; ++k) {
    const R Wk = UNITW ? R(1) : rule.W;
    my[k] = ST::lin(k) >= 0 ? Wk * m[ST::lin(k) >= 0 ? ST::lin(k) : 0] : Wk * y0[k] + rule.wi * A[k];
#pragma unroll
    for (int l = 0; l <= k; ++l) {
      if (ST::lin(k) >= 0 && ST::lin(l) >= 0)
        Sy[tri(k, l)] = Sin[tri_any(ST::lin(k) >= 0 ? ST::lin(k) : 0, ST::lin(l) >= 0 ? ST::lin(l) : 0)];
      else
        Sy[tri(k, l)] = hw * Sy[tri(k, l)] - w2 * A[k] * A[l];
    }
    if (CROSS) {
#pragma unroll
      for (int i = 0; i < DIN; ++i)
        Sxy[i * DOUT + k] = ST::lin(k) >= 0 ? Sin[tri_any(i, ST::lin(k) >= 0 ? ST::lin(k) : 0)] : cs * Sxy[i * DOUT + k];
    }
  }
  if (!UNITW && !rule.unit) {  // sum of weights != 1: the reference's m m^T term no longer cancels
    const R omw = R(1) - rule.W;
    R yc[DOUT];
#pragma unroll
    for (int k = 0; k < DOUT; ++k) yc[k] = ST::lin(k) >= 0 ? m[ST::lin(k) >= 0 ? ST::lin(k) : 0] : y0[k];
#pragma unroll
    for (int k = 0; k < DOUT; ++k)
#pragma unroll
      for (int l = 0; l <= k; ++l)
        Sy[tri(k, l)] += omw * (rule.W * yc[k] * yc[l] + rule.wi * (A[k] * yc[l] + yc[k] * A[l]));
  }
}

// Kalman-style update of N(mu, S) (dimension DX) on an observation with moments
// (mz, Sz + noise, Sxz) and target zt:  i2c.py:394-403 / 435-443.
// On entry Sz already includes the observation noise. Returns false if Sz is not PD.
template <int DX, int DZ, typename R>
I2C_FN bool kalman_update(R* mu, R* S, const R* mz, R* Sz, R* Sxz, const R* zt) {
  R rinv[DZ];
  const bool ok = chol<DZ>(Sz, rinv);
  R q[DZ];
#pragma unroll
  for (int k = 0; k < DZ; ++k) q[k] = zt[k] - mz[k];
  fsub<DZ>(Sz, rinv, q);
#pragma unroll
  for (int i = 0; i < DX; ++i) {
    fsub<DZ>(Sz, rinv, &Sxz[i * DZ]);  // row i of V^T
    R v = mu[i];
#pragma unroll
    for (int k = 0; k < DZ; ++k) v += Sxz[i * DZ + k] * q[k];
    mu[i] = v;
    sched_fence<(DZ >= 8)>();
  }
#pragma unroll
  for (int i = 0; i < DX; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      R v = S[tri(i, j)];
#pragma unroll
      for (int k = 0; k < DZ; ++k) v -= Sxz[i * DZ + k] * Sxz[j * DZ + k];
      S[tri(i, j)] = v;
    }
  return ok;
}

// exp(-1/2 delta^T S^{-1} delta): the ratio N(x; m, S) / N(m; m, S) of i2c.py:369-374, 162-165.
template <int N, typename R> I2C_FN R pdf_ratio(R* S, const R* delta, bool* ok, const PolyTab<R>* tab = nullptr) {
  R rinv[N], q[N];
  *ok = chol<N>(S, rinv);
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = delta[i];
  fsub<N>(S, rinv, q);
  R maha = R(0);
#pragma unroll
  for (int i = 0; i < N; ++i) maha += q[i] * q[i];
  (void)tab;
  return r_exp(R(-0.5) * maha);
}

// Joint prior over (x, u) from the incoming state message and the previous controller
// (feedback mode: i2c.py:361-387; propagation: i2c.py:158-179). Kt = scaled gain (nu x nx).
//   mu_u = qmu_u + Kt (mu_x - qmu_x);  cross = Kt sig_x;  sig_u given by the caller.
template <int NX, int NU, typename R>
I2C_FN void joint_from_gain(const R* mu_x, const R* sig_x, const R* Kt, const R* qmu_x, const R* qmu_u,
                            const R* sig_u, R* mu0, R* S0) {
  constexpr int D = NX + NU;
#pragma unroll
  for (int i = 0; i < NX; ++i) mu0[i] = mu_x[i];
#pragma unroll
  for (int a = 0; a < NU; ++a) {
    R v = qmu_u[a];
#pragma unroll
    for (int k = 0; k < NX; ++k) v += Kt[a * NX + k] * (mu_x[k] - qmu_x[k]);
    mu0[NX + a] = v;
  }
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) S0[tri(i, j)] = sig_x[tri(i, j)];
#pragma unroll
  for (int a = 0; a < NU; ++a) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      R v = R(0);
#pragma unroll
      for (int k = 0; k < NX; ++k) v += Kt[a * NX + k] * sig_x[tri_any(k, j)];
      S0[tri(NX + a, j)] = v;
    }
#pragma unroll
    for (int c = 0; c <= a; ++c) S0[tri(NX + a, NX + c)] = sig_u[tri(a, c)];
  }
  (void)D;
}

// Kt sig Kt^T (packed sym NU) for packed sym sig (NX)
template <int NX, int NU, typename R> I2C_FN void gain_quad(const R* Kt, const R* sig, R* out) {
  R KS[NU * NX];
#pragma unroll
  for (int a = 0; a < NU; ++a)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      R v = R(0);
#pragma unroll
      for (int k = 0; k < NX; ++k) v += Kt[a * NX + k] * sig[tri_any(k, j)];
      KS[a * NX + j] = v;
    }
#pragma unroll
  for (int a = 0; a < NU; ++a)
#pragma unroll
    for (int c = 0; c <= a; ++c) {
      R v = R(0);
#pragma unroll
      for (int j = 0; j < NX; ++j) v += KS[a * NX + j] * Kt[c * NX + j];
      out[tri(a, c)] = v;
    }
}

// Branch-free bookkeeping of the first failure: keeps the sweep's cells free of control flow, so that
// each cell stays one large scheduling region.
I2C_FN int note_failure(int fail, bool ok, int reason, int t) {
  return (fail == 0 && !ok) ? ((reason << 16) | (t + 1)) : fail;
}
// Cheaper bookkeeping inside a forward cell: each stage only raises a small integer (inline constants, no `fail == 0`
// test), and the first-failure status word is formed once per cell. Stages in execution order 0..5 = pdf ratio, prior
// joint, observation, updated joint, prediction, terminal update (reasons 2, 1, 3, 4, 5, 6); the FIRST failed stage has
// the largest value, and later stages of a poisoned cell cannot override it.
I2C_FN int flag_stage(int cell_bad, bool ok, int order) {
  const int v = ok ? 0 : 7 - order;
  return cell_bad > v ? cell_bad : v;
}
I2C_FN int fold_cell_failure(int fail, int cell_bad, int t) {
  const int order = 7 - cell_bad;
  const int reason = order == 0 ? 2 : (order == 1 ? 1 : order + 1);
  return (fail == 0 && cell_bad != 0) ? ((reason << 16) | (t + 1)) : fail;
}
I2C_FN void set_status(int32_t* status, int b, int reason, int t) {
  if (status[b] == 0) status[b] = (reason << 16) | (t + 1);
}

// ------------------------------------------------------------------------------------------
// Forward sweep: one lane walks one trajectory through all T cells (i2c.py:876-880, 350-447).
// ------------------------------------------------------------------------------------------
// Gaussian push-through with a tensor-grid rule (GaussHermiteQuadrature): the same moments as sp_transform,
//   my = sum_p w_p y_p,  Sy = sum_p w_p y_p y_p^T - my my^T,  Sxy = sum_p w_p x_p y_p^T - m my^T   (quadrature.py:34-44),
// accumulated about the centre value y0 = f(m):  my = y0 + s1,  Sy = S2 - s1 s1^T,  Sxy = sum_p w_p dx_p dy_p^T
// (the weights sum to 1 and the grid is symmetric, so the terms in (1 - W) and sum_p w_p dx_p vanish to rounding).
// The point index runs through a mixed-radix odometer in registers; gh_degree^DIN points, no structure is exploited.
template <class M, int DIN, int DOUT, bool CROSS, typename R, class F>
I2C_FN void grid_transform(const Rule<R>& rule, const R* m, const R* L, const F& f, R* my, R* Sy, R* Sxy) {
  constexpr int NA = M::NA, NA1 = NA > 0 ? NA : 1;
  R sn[NA1], cs[NA1], y0[DOUT], s1[DOUT];
#pragma unroll
  for (int a = 0; a < NA; ++a) r_sincos(m[M::ang(a)], &sn[a], &cs[a]);
  f(m, sn, cs, y0);
#pragma unroll
  for (int k = 0; k < DOUT; ++k) s1[k] = R(0);
#pragma unroll
  for (int k = 0; k < sym(DOUT); ++k) Sy[k] = R(0);
  if (CROSS) {
#pragma unroll
    for (int k = 0; k < DIN * DOUT; ++k) Sxy[k] = R(0);
  }
  int dig[DIN];
#pragma unroll
  for (int i = 0; i < DIN; ++i) dig[i] = 0;
  const int deg = rule.gh_degree;
  for (int p = 0; p < rule.gh_points; ++p) {
    R xi[DIN], w = R(1);
#pragma unroll
    for (int i = 0; i < DIN; ++i) {
      R xv = rule.gh_x[0], wv = rule.gh_w[0];
#pragma unroll
      for (int q = 1; q < I2C_MAX_GH_DEGREE; ++q) {
        xv = dig[i] == q ? rule.gh_x[q] : xv;
        wv = dig[i] == q ? rule.gh_w[q] : wv;
      }
      xi[i] = xv;
      w *= wv;
    }
    R dx[DIN], x[DIN], y[DOUT];
#pragma unroll
    for (int i = 0; i < DIN; ++i) {
      R v = R(0);
#pragma unroll
      for (int j = 0; j <= i; ++j) v += L[tri(i, j)] * xi[j];
      dx[i] = rule.sf * v;
      x[i] = m[i] + dx[i];
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) r_sincos(x[M::ang(a)], &sn[a], &cs[a]);
    f(x, sn, cs, y);
#pragma unroll
    for (int k = 0; k < DOUT; ++k) y[k] -= y0[k];
#pragma unroll
    for (int k = 0; k < DOUT; ++k) {
      const R wy = w * y[k];
      s1[k] += wy;
#pragma unroll
      for (int l = 0; l <= k; ++l) Sy[tri(k, l)] += wy * y[l];
      if (CROSS) {
#pragma unroll
        for (int i = 0; i < DIN; ++i) Sxy[i * DOUT + k] += dx[i] * wy;
      }
    }
    // odometer: dig[0] is the fastest digit
    bool carry = true;
#pragma unroll
    for (int i = 0; i < DIN; ++i) {
      const int nd = dig[i] + (carry ? 1 : 0);
      carry = nd >= deg;
      dig[i] = carry ? 0 : nd;
    }
  }
#pragma unroll
  for (int k = 0; k < DOUT; ++k) my[k] = y0[k] + s1[k];
#pragma unroll
  for (int k = 0; k < DOUT; ++k)
#pragma unroll
    for (int l = 0; l <= k; ++l) Sy[tri(k, l)] -= s1[k] * s1[l];
}

// The same transform with the grid UNROLLED at compile time (degree DEG, DIN <= 3: 27 / 64 points for the pendulum-sized models),
// axis 0 outermost. What the run-time odometer above cannot exploit:
//   * L is lower triangular: axis j only moves the coordinates i >= j, so the offsets of a point are built level by level
//     (one multiply-add per coordinate and level instead of a triangular product per point);
//   * an angle coordinate with input index ia is fixed once the axes 0 .. ia are: its sine / cosine is evaluated at THAT level
//     and shared by every point below it (the pendulum's angle is coordinate 0: DEG evaluations instead of DEG^3);
//   * the weight of a point is the product of its levels' weights: one multiply per point;
//   * the cross-covariance rows of the coordinates that do not move inside the innermost axis take the innermost partial sums
//     sum_k w_k dy_k once per pass instead of once per point.
// Same moments, same order of the points within a level; the sums associate differently (last bits).
template <class M, int DEG, int DIN, int DOUT, bool CROSS, typename R, class F>
I2C_FN void grid_transform_ct(const Rule<R>& rule, const R* m, const R* L, const F& f, R* my, R* Sy, R* Sxy) {
  static_assert(DIN >= 1 && DIN <= 3, "unrolled grid: up to three input dimensions");
  constexpr int NA = M::NA, NA1 = NA > 0 ? NA : 1;
  R sn[NA1], cs[NA1], y0[DOUT], s1[DOUT];
#pragma unroll
  for (int a = 0; a < NA; ++a) r_sincos(m[M::ang(a)], &sn[a], &cs[a]);
  f(m, sn, cs, y0);
#pragma unroll
  for (int k = 0; k < DOUT; ++k) s1[k] = R(0);
#pragma unroll
  for (int k = 0; k < sym(DOUT); ++k) Sy[k] = R(0);
  if (CROSS) {
#pragma unroll
    for (int k = 0; k < DIN * DOUT; ++k) Sxy[k] = R(0);
  }
  R xs[DEG], ws[DEG];  // sf * node, weight
#pragma unroll
  for (int q = 0; q < DEG; ++q) xs[q] = rule.sf * rule.gh_x[q], ws[q] = rule.gh_w[q];
  constexpr int N1 = DIN > 1 ? DEG : 1, N2 = DIN > 2 ? DEG : 1;
#pragma unroll
  for (int i0 = 0; i0 < DEG; ++i0) {
    R d0[DIN];  // offsets after level 0
#pragma unroll
    for (int i = 0; i < DIN; ++i) d0[i] = L[tri(i, 0)] * xs[i0];
#pragma unroll
    for (int a = 0; a < NA; ++a)
      if (M::ang(a) == 0) r_sincos(m[0] + d0[0], &sn[a], &cs[a]);
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1) {
      R d1[DIN];
      R w1 = ws[i0];
#pragma unroll
      for (int i = 0; i < DIN; ++i) d1[i] = d0[i];
      if constexpr (DIN > 1) {
#pragma unroll
        for (int i = 1; i < DIN; ++i) d1[i] += L[tri(i, 1)] * xs[i1];
        w1 *= ws[i1];
#pragma unroll
        for (int a = 0; a < NA; ++a)
          if (M::ang(a) == 1) r_sincos(m[1] + d1[1], &sn[a], &cs[a]);
      }
      R part[DOUT];  // sum over the innermost axis of w dy (cross-covariance rows of the coordinates it does not move)
#pragma unroll
      for (int k = 0; k < DOUT; ++k) part[k] = R(0);
#pragma unroll
      for (int i2 = 0; i2 < N2; ++i2) {
        R dx[DIN], x[DIN], y[DOUT];
        R w = w1;
#pragma unroll
        for (int i = 0; i < DIN; ++i) dx[i] = d1[i];
        if constexpr (DIN > 2) {
          dx[2] += L[tri(2, 2)] * xs[i2];
          w *= ws[i2];
#pragma unroll
          for (int a = 0; a < NA; ++a)
            if (M::ang(a) == 2) r_sincos(m[2] + dx[2], &sn[a], &cs[a]);
        }
#pragma unroll
        for (int i = 0; i < DIN; ++i) x[i] = m[i] + dx[i];
        f(x, sn, cs, y);
#pragma unroll
        for (int k = 0; k < DOUT; ++k) {
          const R wy = w * (y[k] - y0[k]);
          y[k] -= y0[k];
          part[k] += wy;
#pragma unroll
          for (int l = 0; l <= k; ++l) Sy[tri(k, l)] += wy * y[l];
          if (CROSS) Sxy[(DIN - 1) * DOUT + k] += dx[DIN - 1] * wy;  // the coordinate the innermost axis moves
        }
      }
#pragma unroll
      for (int k = 0; k < DOUT; ++k) {
        s1[k] += part[k];
        if (CROSS) {
#pragma unroll
          for (int i = 0; i < DIN - 1; ++i) Sxy[i * DOUT + k] += d1[i] * part[k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < DOUT; ++k) my[k] = y0[k] + s1[k];
#pragma unroll
  for (int k = 0; k < DOUT; ++k)
#pragma unroll
    for (int l = 0; l <= k; ++l) Sy[tri(k, l)] -= s1[k] * s1[l];
}

// GRID selects the tensor-grid rule at compile time; the sigma-point kernels are unchanged by it.
template <bool GRID, class M, class ST, int DIN, int DOUT, bool CROSS, bool UNITW = false, typename R, class F>
I2C_FN void transform(const Rule<R>& rule, const R* m, const R* Sin, const R* L, const F& f, R* my, R* Sy, R* Sxy,
                      const PolyTab<R>* tab = nullptr) {
  if constexpr (GRID) {
    if constexpr (DIN <= 3) {  // degrees 2 .. 4 of the pendulum-sized models: the unrolled grid (wave-uniform choice)
      if (rule.gh_degree == 3) return grid_transform_ct<M, 3, DIN, DOUT, CROSS>(rule, m, L, f, my, Sy, Sxy);
      if (rule.gh_degree == 4) return grid_transform_ct<M, 4, DIN, DOUT, CROSS>(rule, m, L, f, my, Sy, Sxy);
      if (rule.gh_degree == 2) return grid_transform_ct<M, 2, DIN, DOUT, CROSS>(rule, m, L, f, my, Sy, Sxy);
    }
    grid_transform<M, DIN, DOUT, CROSS>(rule, m, L, f, my, Sy, Sxy);
  } else
    sp_transform<M, ST, DIN, DOUT, CROSS, UNITW>(rule, m, Sin, L, f, my, Sy, Sxy, tab);
}


// S = element type of the PER-CELL buffers in HBM (prior/post, fwd, prior_out, xm, zpost): R normally, float with R =
// double in the mixed mode I2C_F64_F32S (fp64 arithmetic on fp32-stored messages). Per-trajectory data stays R.
template <typename R, typename S = R> struct FwdArgs {
  const S* prior;   // [T][E_POST][B]
  S* fwd;           // [T][E_FWD][B]
  S* prior_out;     // [T][D + sym(D)][B] or null
  const R* x0;      // [NX][B]
  const R* sig_x0;  // [sym NX][B]
  const R* z;       // [T][NZ][B] or null
  const R* alpha;   // [B]
  const R* alpha_cell;  // [T][B] or null
  const uint8_t* ff;  // [T]
  int32_t* status;  // [B]
  const uint8_t* expert;  // [T] ring or null: per-cell use_expert_controller (Linearize forward only, i2c.py:259-265)
};

// LEAN = the common case fixed at compile time (weights sum to 1, shared target, trajectory-level alpha, no
// joint-prior output): the corresponding wave-uniform runtime branches disappear from the cell, which keeps
// it one scheduling region. The generic variant (LEAN = false) handles everything.
template <class M, typename R, bool LEAN = false, bool GRID = false, typename ST_ = R>
I2C_HD inline void forward_sweep_body(const Consts<M, R>& c, const FwdArgs<R, ST_>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D;
  constexpr unsigned W = sizeof(ST_);
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned bo = (unsigned)b * W;     // the lane's byte offset inside any row
  const unsigned rb0 = (unsigned)(B * W);  // bytes per row (wave-uniform)
  const R alpha_traj = a.alpha[b];
  int fail = 0;  // first failure of this trajectory, (reason << 16) | (t + 1); kept in a register, stored once

  R mu_x[NX], sig_x[sym(NX)];
#pragma unroll
  for (int i = 0; i < NX; ++i) mu_x[i] = a.x0[i * B + b];
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) sig_x[i] = a.sig_x0[i * B + b];

  // Per-row byte offsets bo + e * rb as VGPR values (small models): a buffer access then needs no scalar multiply for its
  // row offset. The lone wave of the small-batch regime is instruction-issue-bound and a SALU instruction costs it the
  // same 4 clocks as a VALU one (profiles/r1_k_forward_sq_counters.json); VGPRs are plentiful here.
  constexpr bool VOFF = C::D <= 5;
  unsigned voff[VOFF ? C::E_FWD : 1];
  if (VOFF) {
#pragma unroll
    for (int e = 0; e < C::E_FWD; ++e) voff[e] = bo + (unsigned)e * rb0;
  }
  // software prefetch of the next cell's prior rows: the loads do not depend on the recursion.
  // Small models fetch the NEXT cell's prior rows right after a cell has consumed its own (below), straight into the same
  // registers: a whole cell ahead of their use, no copies. For d >= 6 those d + s(d) + nu nx doubles would sit on top of a
  // register peak that already fills the 512-VGPR file, so there EVERY cell (the first included: loaded here and skipped in
  // the loop, the rows become loop-carried values that stay live through every cell to the back edge) loads its rows at the
  // top of its own cell: one exposed HBM round trip (~1 us) per ~15 us cell rather than scratch traffic throughout.
  constexpr bool PREFETCH = C::D <= 5;
  R pri[C::E_PRI], zt[NZ];
  if (PREFETCH) {
    const unsigned rb = rb0;
    const Window w = make_window(a.prior + (unsigned long)(LEAN ? 0 : c.row(0)) * C::E_POST * B, (unsigned long)C::E_POST * rb);
#pragma unroll
    for (int e = 0; e < C::E_PRI; ++e) pri[e] = (R)wld<ST_>(w, VOFF ? 0u : (e) * rb, VOFF ? voff[e] : bo);
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = (!LEAN && c.z_per_cell) ? a.z[((long)c.row(0) * NZ + k) * B + b] : c.zg[k];
  }

  // The feed-forward flag of a cell is a byte in global memory: loaded at the top of its own cell it is a dependent
  // vector load whose full latency (plus, vmcnt being shared, the acknowledgement of the previous cell's last stores)
  // is exposed EVERY cell. It is therefore fetched one cell ahead, together with the prior rows.
  unsigned ff_cur = a.ff[LEAN ? 0 : c.row(0)];
  // Small models never factor the prior joint: its Cholesky factor is assembled from the factor Lx of the incoming
  // sig_x0_f (= chol(sig_x3_f) of the previous cell, already needed for the smoother gain) as
  //   L0 = [[Lx, 0], [K~ Lx, chol(sig_u|x)]],  sig_u|x = sig_u0_m - K~ sig_ux^T   (feed-forward: K~ = 0, sig_u|x = sig_u0_f),
  // which takes one nu x nu factorisation instead of a d x d one off the critical path. (For d >= 6 the carried
  // factor would cost nx(nx+1)/2 more live registers across the whole cell.)
  constexpr bool STRUCT_L0 = C::D <= 5;
  R Lx[STRUCT_L0 ? sym(NX) : 1];
  if (STRUCT_L0) {
    R rx[NX];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) Lx[i] = sig_x[i];
    fail = note_failure(fail, chol<NX>(Lx, rx), 1, 0);
  }

  // Settle the loads issued so far: the waitcnt pass joins the loop-entry state with the back-edge state, and pending
  // loads on the entry path would put an `s_waitcnt vmcnt(0)` at the loop top that executes every cell (see below).
  ff_cur = opaque(ff_cur);
  if (PREFETCH) {
#pragma unroll
    for (int e = 0; e < C::E_PRI; ++e) pri[e] = opaque(pri[e]);
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = opaque(zt[k]);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) mu_x[i] = opaque(mu_x[i]);
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) sig_x[i] = opaque(sig_x[i]);
  const R alpha_settled = opaque(alpha_traj);
  R alpha_cur = opaque((!LEAN && a.alpha_cell) ? a.alpha_cell[(long)c.row(0) * B + b] : alpha_settled);  // per-cell temperature, fetched a cell ahead
  // Per-cell constants as VGPR values (small models): left as kernel arguments they sit in ~30 SGPRs for the whole
  // sweep, and the scalar file then spills (v_readlane) and re-materialises polynomial literals (s_mov) in every cell.
  // Only where the register file has room: the cartpole (21 + 10 doubles) already overflows into AGPRs and slows down.
  constexpr bool CONST_V = VOFF && (sym(NZ) + sym(NX) <= 16);
  PolyTab<R> ptab;  // sincos coefficients as VGPR values (see PolyTab), same condition
  if (CONST_V) poly_tab_init(ptab);
  const PolyTab<R>* const tab = CONST_V ? &ptab : nullptr;
  R xi0_v[CONST_V ? sym(NZ) : 1], eta_v[CONST_V ? sym(NX) : 1];
  if (CONST_V) {
#pragma unroll
    for (int i = 0; i < sym(NZ); ++i) xi0_v[i] = opaque(c.sig_xi0[i]);
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) eta_v[i] = opaque(LEAN ? c.sig_eta[i] : c.sig_eta_w[i]);
  }

  for (int t = 0; t < T; ++t) {
    const int tn = t + 1 < T ? t + 1 : t;
    const int tr = LEAN ? t : c.row(t), tnr = LEAN ? tn : c.row(tn);  // rows of the persistent buffers (ring, see Consts::t0)
    const unsigned rb = opaque_uniform(rb0);  // see opaque_uniform(): no hoisting of e * rb
    // Large models: an index the compiler cannot see through (always 0) makes the big constant tables of the kernel
    // argument (sig_xi0: 45 doubles for the double cartpole, sig_eta: 21) scalar LOADS next to their single use in each
    // cell; hoisted out of the time loop they occupy ~130 SGPRs and the scalar file spills through v_writelane /
    // v_readlane (540 of the 7 800 instructions of that kernel).
    const int kz = (C::D >= 6) ? (int)opaque_uniform(0u) : 0;
    if (!PREFETCH) {  // see PREFETCH above
      const Window w = make_window(a.prior + (unsigned long)tr * C::E_POST * B, (unsigned long)C::E_POST * rb);
#pragma unroll
      for (int e = 0; e < C::E_PRI; ++e) pri[e] = (R)wld<ST_>(w, VOFF ? 0u : (e) * rb, VOFF ? voff[e] : bo);
#pragma unroll
      for (int k = 0; k < NZ; ++k) zt[k] = (!LEAN && c.z_per_cell) ? a.z[((long)tr * NZ + k) * B + b] : c.zg[k];
    }

    // per-cell temperature only in the MPC loop (stale sig_xi of appended cells); else the trajectory's
    const R alpha = opaque(alpha_cur);
    const R* pmu = pri;               // prior joint mean  (== previous posterior, see i2c_hip.h)
    const R* psig = pri + D;          // prior joint covariance
    const R* Kprev = pri + D + sym(D);

    // ---- 1. joint prior over (x, u) ---------------------------------------------------
    int cell_bad = 0;  // 0, or 7 - (execution order of the first failed stage of this cell), see flag_stage()
    R mu0[D], S0[sym(D)];
    R L0[STRUCT_L0 ? sym(D) : 1], Luu[sym(NU)], ruu[NU];  // Luu: conditional action covariance, then its factor
    if (ff_cur != 0) {  // feed-forward: independent action prior (i2c.py:355-360)
#pragma unroll
      for (int i = 0; i < NX; ++i) mu0[i] = mu_x[i];
#pragma unroll
      for (int i = NX; i < D; ++i) mu0[i] = pmu[i];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
          S0[tri(i, j)] = (i < NX) ? sig_x[tri(i, j)] : (j >= NX ? psig[tri(i, j)] : R(0));
      if (STRUCT_L0) {
#pragma unroll
        for (int p = 0; p < NU; ++p)
#pragma unroll
          for (int q = 0; q <= p; ++q) Luu[tri(p, q)] = psig[tri(NX + p, NX + q)];
#pragma unroll
        for (int p = 0; p < NU; ++p)
#pragma unroll
          for (int j = 0; j < NX; ++j) L0[tri(NX + p, j)] = R(0);
      }
    } else {  // feedback: condition the previous controller on the new state message (i2c.py:361-387)
      R S[sym(NX)], delta[NX];
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) S[i] = psig[i] + sig_x[i];  // xx block is the packed prefix
#pragma unroll
      for (int i = 0; i < NX; ++i) delta[i] = mu_x[i] - pmu[i];
      bool ok;
      const R rho = pdf_ratio<NX>(S, delta, &ok, tab);
      cell_bad = flag_stage(cell_bad, ok, 0);
      R Kt[NU * NX];
#pragma unroll
      for (int i = 0; i < NU * NX; ++i) Kt[i] = rho * Kprev[i];
      // sig_u0_f = sig_u0_m - Kt sig_ux^T + Kt sig_x0_f Kt^T
      R sig_u[sym(NU)];
      gain_quad<NX, NU>(Kt, sig_x, sig_u);
#pragma unroll
      for (int p = 0; p < NU; ++p)
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          R v = psig[tri(NX + p, NX + q)];
#pragma unroll
          for (int k = 0; k < NX; ++k) v -= Kt[p * NX + k] * psig[tri(NX + q, k)];
          Luu[tri(p, q)] = v;  // sig_u0_m - K~ sig_ux^T
          sig_u[tri(p, q)] += v;
        }
      joint_from_gain<NX, NU>(mu_x, sig_x, Kt, pmu, pmu + NX, sig_u, mu0, S0);
      if (STRUCT_L0) {
#pragma unroll
        for (int p = 0; p < NU; ++p)
#pragma unroll
          for (int j = 0; j < NX; ++j) {
            R v = R(0);
#pragma unroll
            for (int k = j; k < NX; ++k) v += Kt[p * NX + k] * Lx[tri(k, j)];
            L0[tri(NX + p, j)] = v;
          }
      }
    }
    if (STRUCT_L0) {
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) L0[i] = Lx[i];  // xx block = packed prefix
      cell_bad = flag_stage(cell_bad, chol<NU>(Luu, ruu), 1);
#pragma unroll
      for (int p = 0; p < NU; ++p)
#pragma unroll
        for (int q = 0; q <= p; ++q) L0[tri(NX + p, NX + q)] = Luu[tri(p, q)];
    }
    if (!LEAN && a.prior_out) {
      const Window w = make_window(a.prior_out + (unsigned long)t * (D + sym(D)) * B, (unsigned long)(D + sym(D)) * rb);
#pragma unroll
      for (int e = 0; e < D; ++e) wst(w, VOFF ? 0u : (e) * rb, VOFF ? voff[e] : bo, (ST_)mu0[e]);
#pragma unroll
      for (int e = 0; e < sym(D); ++e) wst(w, VOFF ? 0u : (D + e) * rb, VOFF ? voff[D + e] : bo, (ST_)S0[e]);
    }

    ff_cur = a.ff[tnr];  // the next cell's flag, a whole cell ahead of its use
    if (!LEAN && a.alpha_cell) alpha_cur = a.alpha_cell[(long)tnr * B + b];
    if (PREFETCH) {  // pri is dead from here on: refill it with the next cell's rows
      const Window w = make_window(a.prior + (unsigned long)tnr * C::E_POST * B, (unsigned long)C::E_POST * rb);
#pragma unroll
      for (int e = 0; e < C::E_PRI; ++e) pri[e] = (R)wld<ST_>(w, VOFF ? 0u : (e) * rb, VOFF ? voff[e] : bo);
    }

    // ---- 2. cost "observation": measurement update on z (i2c.py:390-407) --------------
    sched_fence<(D >= 6)>();
    {
      R Lf[STRUCT_L0 ? 1 : sym(D)], rinv[D];
      if (!STRUCT_L0) {
#pragma unroll
        for (int i = 0; i < sym(D); ++i) Lf[i] = S0[i];
        cell_bad = flag_stage(cell_bad, chol<D>(Lf, rinv), 1);
      }
      const R* L = STRUCT_L0 ? L0 : Lf;
      R mz[NZ], Sz[sym(NZ)], Sxz[D * NZ];
      transform<GRID, M, ObsStruct<M>, D, NZ, true, LEAN>(c.rule_xu, mu0, S0, L, ObserveF<M, R>{c.params}, mz, Sz, Sxz, tab);
#pragma unroll
      for (int i = 0; i < sym(NZ); ++i) Sz[i] += alpha * (CONST_V ? xi0_v[i] : c.sig_xi0[i + kz]);
      cell_bad = flag_stage(cell_bad, kalman_update<D, NZ>(mu0, S0, mz, Sz, Sxz, zt), 2);
    }
    if (!LEAN && PREFETCH && c.z_per_cell) {  // the target is consumed: fetch the next cell's
#pragma unroll
      for (int k = 0; k < NZ; ++k) zt[k] = a.z[((long)tnr * NZ + k) * B + b];
    }
    // mu0 / S0 now hold mu_xu1_f / sig_xu1_f
    const Window out = make_window(a.fwd + (unsigned long)t * C::E_FWD * B, (unsigned long)C::E_FWD * rb);
#pragma unroll
    for (int e = 0; e < D; ++e) wst(out, VOFF ? 0u : (e) * rb, VOFF ? voff[e] : bo, (ST_)mu0[e]);
#pragma unroll
    for (int e = 0; e < sym(D); ++e) wst(out, VOFF ? 0u : (D + e) * rb, VOFF ? voff[D + e] : bo, (ST_)S0[e]);

    // ---- 3. dynamics push-through (i2c.py:415-421) and smoother gain (i2c.py:423-425) --
    sched_fence<(D >= 6)>();
    R Sxy[D * NX];
    {
      R L[sym(D)], rinv[D];
#pragma unroll
      for (int i = 0; i < sym(D); ++i) L[i] = S0[i];
      cell_bad = flag_stage(cell_bad, chol<D>(L, rinv), 3);
      transform<GRID, M, DenseStruct<D>, D, NX, true, LEAN>(c.rule_xu, mu0, S0, L, DynamicsF<M, R>{c.params}, mu_x, sig_x, Sxy, tab);
    }
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) sig_x[i] += CONST_V ? eta_v[i] : (LEAN ? c.sig_eta[i + kz] : c.sig_eta_w[i + kz]);  // sum_p w_p sig_eta (quadrature.py:57)
    R L3[sym(NX)], rinv3[NX];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) L3[i] = sig_x[i];
    cell_bad = flag_stage(cell_bad, chol<NX>(L3, rinv3), 4);
    sched_fence<(D >= 6)>();
#pragma unroll
    for (int i = 0; i < D; ++i) {  // J = sig_xy sig_x3^{-1}, row by row
      fsub<NX>(L3, rinv3, &Sxy[i * NX]);
      bsub<NX>(L3, rinv3, &Sxy[i * NX]);
      sched_fence<(D >= 6)>();
    }

    // gfx9 counts loads AND stores in one counter (vmcnt). A wait on the prefetched prior rows placed (by the compiler)
    // at the top of the next cell also waits for the L2 to acknowledge this cell's LAST stores, issued a few
    // instructions earlier. Touching the prefetched registers here, before the tail stores, puts the wait where
    // everything outstanding is old -- the loads were issued most of a cell ago -- and, together with settling the
    // pre-loop loads (above; the waitcnt pass joins the loop-entry state with the back-edge state), leaves the loop
    // top without any s_waitcnt: 372 -> 357 us. (Measured alternative: touching after the tail stores, 363 us.)
    ff_cur = opaque(ff_cur);
    if (!LEAN) {  // the per-cell temperature and targets of the MPC loop were fetched ahead as well
      alpha_cur = opaque(alpha_cur);
      if (PREFETCH) {
#pragma unroll
        for (int k = 0; k < NZ; ++k) zt[k] = opaque(zt[k]);
      }
    }
    if (PREFETCH) {
#pragma unroll
      for (int e = 0; e < C::E_PRI; ++e) pri[e] = opaque(pri[e]);
    }
    // J is written out BEFORE the terminal update so that its d*nx registers are dead there (with J
    // live the terminal block is the register peak of the large models and spills to scratch)
#pragma unroll
    for (int e = 0; e < D * NX; ++e) wst(out, VOFF ? 0u : (D + sym(D) + NX + sym(NX) + e) * rb, VOFF ? voff[D + sym(D) + NX + sym(NX) + e] : bo, (ST_)Sxy[e]);
    sched_fence<(D >= 6)>();

    // ---- 4. terminal cost observation on the flagged cell, after J (i2c.py:430-443) ----
    if (NZT > 0 && t == c.terminal_cell && c.has_Qf) {
      constexpr int NT = C::NZT1;
      R mzt[NT], Szt[sym(NT)], Sxzt[NX * NT];
      transform<GRID, M, TermStruct<M>, NX, NT, true, LEAN>(c.rule_x, mu_x, sig_x, L3, ObserveTermF<M, R>{c.params}, mzt, Szt, Sxzt);
#pragma unroll
      for (int i = 0; i < sym(NT); ++i) Szt[i] += alpha * c.sig_xiT0[i];
      cell_bad = flag_stage(cell_bad, kalman_update<NX, NT>(mu_x, sig_x, mzt, Szt, Sxzt, c.zg_term), 5);
      if (STRUCT_L0) {  // the next cell (MPC: the flagged cell can sit mid-horizon) needs the factor of the UPDATED sig_x
#pragma unroll
        for (int i = 0; i < sym(NX); ++i) L3[i] = sig_x[i];
        cell_bad = flag_stage(cell_bad, chol<NX>(L3, rinv3), 5);
      }
    }
    if (STRUCT_L0) {
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) Lx[i] = L3[i];
    }
    fail = fold_cell_failure(fail, cell_bad, t);
#pragma unroll
    for (int e = 0; e < NX; ++e) wst(out, VOFF ? 0u : (D + sym(D) + e) * rb, VOFF ? voff[D + sym(D) + e] : bo, (ST_)mu_x[e]);
#pragma unroll
    for (int e = 0; e < sym(NX); ++e) wst(out, VOFF ? 0u : (D + sym(D) + NX + e) * rb, VOFF ? voff[D + sym(D) + NX + e] : bo, (ST_)sig_x[e]);

  }
  if (fail != 0 && a.status[b] == 0) a.status[b] = fail;
}

// ------------------------------------------------------------------------------------------
// Backward pass building blocks (i2c.py:544-610), all on registers of one lane.
// ------------------------------------------------------------------------------------------

// End of chain (i2c.py:546-564): the smoothed terminal state, either the filtered one or, for
// covariance control, its product with the tempered terminal prior (i2c.py:548-559).
template <class M, typename R>
I2C_FN void end_of_chain(const Consts<M, R>& c, R* temp, const int b, const R* m3f, const R* S3f, R* m3m, R* S3m,
                         int32_t* status) {
  constexpr int NX = M::NX;
  if (c.has_x_terminal) {
    const R tmp = temp[b];
    temp[b] = tmp + c.dtemp;
    R St[sym(NX)], Ssum[sym(NX)], rinv[NX];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) {
      St[i] = tmp * S3f[i];
      Ssum[i] = c.sig_x_term[i] + St[i];
    }
    bool ok = chol<NX>(Ssum, rinv);
    // sig_x3_m = St - St (sig_T + St)^{-1} St  = St - W^T W,  W = chol(sum)^{-1} St
    R Wm[NX * NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      R col[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) col[i] = St[tri_any(i, j)];
      fsub<NX>(Ssum, rinv, col);
#pragma unroll
      for (int i = 0; i < NX; ++i) Wm[i * NX + j] = col[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        R v = St[tri(i, j)];
#pragma unroll
        for (int k = 0; k < NX; ++k) v -= Wm[k * NX + i] * Wm[k * NX + j];
        S3m[tri(i, j)] = v;
      }
    // mu_x3_m = sig_x3_m (St^{-1} mu_x3_f + sig_T^{-1} mu_T)
    R r1[NX], r2[NX], Lt[sym(NX)], LT[sym(NX)];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) {
      Lt[i] = St[i];
      LT[i] = c.sig_x_term[i];
    }
    ok = chol<NX>(Lt, rinv) && ok;
#pragma unroll
    for (int i = 0; i < NX; ++i) r1[i] = m3f[i];
    fsub<NX>(Lt, rinv, r1);
    bsub<NX>(Lt, rinv, r1);
    ok = chol<NX>(LT, rinv) && ok;
#pragma unroll
    for (int i = 0; i < NX; ++i) r2[i] = c.mu_x_term[i];
    fsub<NX>(LT, rinv, r2);
    bsub<NX>(LT, rinv, r2);
#pragma unroll
    for (int i = 0; i < NX; ++i) r1[i] += r2[i];
    symv<NX>(S3m, r1, m3m);
    if (!ok) set_status(status, b, 6, c.T - 1);
  } else {
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = m3f[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = S3f[i];
  }
}

// Expected quadratic cost of N(mz, Sz) against target zt under weight W (packed sym):
//   m = err^T W err + tr(Sz W),  v = 2 tr((Sz W)^2) + 4 err^T W Sz W err   (i2c.py:1034-1043)
template <int N, typename R>
I2C_FN void gaussian_cost(const R* W, const bool w_diag, const R* mz, const R* Sz, const R* zt, R* m, R* v) {
  R err[N];
#pragma unroll
  for (int i = 0; i < N; ++i) err[i] = mz[i] - zt[i];
  if (w_diag) {  // W = diag(w): every shipped cost (Q, R, Qf diagonal). SW_ij = Sz_ij w_j
    R mm = R(0), tr2 = R(0), quad = R(0), we[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      we[i] = W[tri(i, i)] * err[i];
      mm += we[i] * err[i] + W[tri(i, i)] * Sz[tri(i, i)];
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const R sww = Sz[tri(i, j)] * (W[tri(i, i)] * W[tri(j, j)]);
        tr2 += (i == j ? R(1) : R(2)) * Sz[tri(i, j)] * sww;
        quad += (i == j ? R(1) : R(2)) * Sz[tri(i, j)] * (we[i] * we[j]);
      }
    *m = mm;
    *v = R(2) * tr2 + R(4) * quad;
    return;
  }
  R We[N];
  symv<N>(W, err, We);
  R SW[N * N];  // Sz W (not symmetric)
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) {
      R s = R(0);
#pragma unroll
      for (int k = 0; k < N; ++k) s += Sz[tri_any(i, k)] * W[tri_any(k, j)];
      SW[i * N + j] = s;
    }
  R mm = R(0), tr2 = R(0), quad = R(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    mm += err[i] * We[i] + SW[i * N + i];
    R SWe = R(0);  // (Sz W err)_i = sum_j Sz[i][j] We[j]
#pragma unroll
    for (int j = 0; j < N; ++j) {
      tr2 += SW[i * N + j] * SW[j * N + i];
      SWe += Sz[tri_any(i, j)] * We[j];
    }
    quad += We[i] * SWe;
  }
  *m = mm;
  *v = R(2) * tr2 + R(4) * quad;
}

// Terminal observation statistics (i2c.py:567-570, 989-992): tr(Qf (errT errT^T + sig_z3_m)).
template <class M, typename R, bool GRID = false>
I2C_FN R terminal_obs_stats(const Consts<M, R>& c, const int b, const R* m3m, const R* S3m, R* term_stats,
                            int32_t* status) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZT = C::NZT;
  const long B = c.B;
  R trT = R(0);
  if (NZT > 0 && c.has_Qf) {
    constexpr int NT = C::NZT1;
    R L3[sym(NX)], rinv3[NX], mzt[NT], Szt[sym(NT)];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) L3[i] = S3m[i];
    if (!chol<NX>(L3, rinv3)) set_status(status, b, 6, c.T - 1);
    transform<GRID, M, TermStruct<M>, NX, NT, false>(c.rule_x, m3m, S3m, L3, ObserveTermF<M, R>{c.params}, mzt, Szt,
                                                  (R*)nullptr);
    R tv;
    gaussian_cost<NT>(c.Qf, c.qf_diag != 0, mzt, Szt, c.zg_term, &trT, &tv);
#pragma unroll
    for (int k = 0; k < NT; ++k) term_stats[(long)(3 + k) * B + b] = mzt[k];
#pragma unroll
    for (int k = 0; k < sym(NT); ++k) term_stats[(long)(3 + NT + k) * B + b] = Szt[k];
  }
  term_stats[b] = trT;
  return trT;
}

// One backward cell given the smoothed next state: RTS update of the joint (i2c.py:580-583),
// posterior observation statistics (i2c.py:594-596) with their expected cost (i2c.py:1034-1043),
// and the controller (i2c.py:600-608) read off the Cholesky factor L of the posterior joint:
//   K L_xx = L_ux  ->  K^T = L_xx^{-T} L_ux^T ;  sigK = L_uu L_uu^T ;  k = mu_u - K mu_x.
// In: mu/S = mu_xu1_f / sig_xu1_f, J, dm = mu_x3_m - mu_x3_f, dS = sig_x3_m - sig_x3_f.
// Out: mu/S = mu_xu0_m / sig_xu0_m, ctl = [K | k | sigK], mz/Sz, cost mean / variance.
template <class M, typename R, bool GRID = false>
I2C_FN bool cell_posterior(const Consts<M, R>& c, const R* zt, R* mu, R* S, const R* J, const R* dm, const R* dS,
                           R* ctl, R* mz, R* Sz, R* cm, R* cv, const PolyTab<R>* tab = nullptr) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    R v = mu[i];
#pragma unroll
    for (int k = 0; k < NX; ++k) v += J[i * NX + k] * dm[k];
    mu[i] = v;
  }
  add_JDJt<D, NX>(J, dS, S);
  R Lm[sym(D)], rinv[D];
#pragma unroll
  for (int e = 0; e < sym(D); ++e) Lm[e] = S[e];
  const bool ok = chol<D>(Lm, rinv);
  transform<GRID, M, ObsStruct<M>, D, NZ, false>(c.rule_xu, mu, S, Lm, ObserveF<M, R>{c.params}, mz, Sz, (R*)nullptr, tab);
  gaussian_cost<NZ>(c.QR, c.qr_diag != 0, mz, Sz, zt, cm, cv);
#pragma unroll
  for (int p = 0; p < NU; ++p) {
#pragma unroll
    for (int k = 0; k < NX; ++k) ctl[p * NX + k] = Lm[tri(NX + p, k)];
    bsub<NX>(Lm, rinv, &ctl[p * NX]);  // leading NX x NX block of L is chol(sig_xx)
  }
#pragma unroll
  for (int p = 0; p < NU; ++p) {
    R v = mu[NX + p];
#pragma unroll
    for (int k = 0; k < NX; ++k) v -= ctl[p * NX + k] * mu[k];
    ctl[NU * NX + p] = v;
  }
#pragma unroll
  for (int p = 0; p < NU; ++p)
#pragma unroll
    for (int q = 0; q <= p; ++q) {
      R v = R(0);
#pragma unroll
      for (int k = 0; k <= q; ++k) v += Lm[tri(NX + p, NX + k)] * Lm[tri(NX + q, NX + k)];
      ctl[NU * NX + NU + tri(p, q)] = v;
    }
  return ok;
}

// ------------------------------------------------------------------------------------------
// Backward sweep, TWO-PASS form (small batches): (1) a light sequential scan of the x-marginal
// recursion
//   mu_x0_m = mu_x1_f + Jx (mu_x3_m - mu_x3_f);  sig_x0_m = sig_x1_f + Jx (sig_x3_m - sig_x3_f) Jx^T,
// which is affine in (mu_x3_m, sig_x3_m), touches only nx x nx blocks and is the ONLY sequential
// part of the backward pass; (2) everything else, independent per cell, one lane per (t, b);
// (3) a deterministic reduction of the per-cell cost statistics over t.
// ------------------------------------------------------------------------------------------
template <typename R, typename S = R> struct ScanArgs {
  const S* fwd;  // [T][E_FWD][B]
  S* xm;         // [T][E_XM][B]
  R* temp;       // [B] or null
  int32_t* status;
};

template <int NX, typename R> struct ScanRow {  // the rows of one cell the recursion needs
  R mu1[NX], S1[sym(NX)], m3f[NX], S3f[sym(NX)], Jx[NX * NX];
};

template <class M, typename R, typename S = R>
I2C_HD inline void backward_scan_body(const Consts<M, R>& c, const ScanArgs<R, S>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  // The recursion itself is ~40 FMAs per cell; what bounds this kernel is load latency with only
  // B/64 wavefronts in flight. Cells are therefore fetched U at a time, one chunk ahead.
  constexpr int U = NX <= 2 ? 4 : 1;
  const long B = c.B;
  const int T = c.T;
  using Row = ScanRow<NX, R>;

  auto load = [&](int t, Row& r) {
    const S* in = a.fwd + ((long)(t > 0 ? t : 0) * C::E_FWD) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) r.mu1[i] = in[(long)i * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) r.S1[i] = in[(long)(D + i) * B];
#pragma unroll
    for (int i = 0; i < NX; ++i) r.m3f[i] = in[(long)(O_MU3 + i) * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) r.S3f[i] = in[(long)(O_S3 + i) * B];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) r.Jx[i] = in[(long)(O_J + i) * B];
  };
  Row cur[U], nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) load(T - 1 - u, cur[u]);

  R m3m[NX], S3m[sym(NX)];
  end_of_chain<M, R>(c, a.temp, b, cur[0].m3f, cur[0].S3f, m3m, S3m, a.status);

  for (int t0 = T - 1; t0 >= 0; t0 -= U) {
#pragma unroll
    for (int u = 0; u < U; ++u) load(t0 - U - u, nxt[u]);  // next chunk (clamped at cell 0)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 - u;
      if (t < 0) break;
      const Row& r = cur[u];
      S* out = a.xm + ((long)t * C::E_XM) * B + b;
#pragma unroll
      for (int i = 0; i < NX; ++i) out[(long)i * B] = m3m[i];
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) out[(long)(NX + i) * B] = S3m[i];
      R dm[NX], dS[sym(NX)];
#pragma unroll
      for (int i = 0; i < NX; ++i) dm[i] = m3m[i] - r.m3f[i];
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) dS[i] = S3m[i] - r.S3f[i];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        R v = r.mu1[i];
#pragma unroll
        for (int k = 0; k < NX; ++k) v += r.Jx[i * NX + k] * dm[k];
        m3m[i] = v;
      }
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) S3m[i] = r.S1[i];
      add_JDJt<NX, NX>(r.Jx, dS, S3m);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
  }
}

#ifndef I2C_WALK_VOFF
#define I2C_WALK_VOFF 1
#endif
template <typename R, typename S = R> struct CellArgs {
  const S* fwd;      // [T][E_FWD][B]
  const S* xm;       // [T][E_XM][B]   (two-pass: input; fused: optional output)
  const R* z;        // [T][NZ][B] or null
  S* post;           // [T][E_POST][B]
  S* zpost;          // [T][E_ZPOST][B] or null
  R* cell_stats;     // [T][2][B]      (two-pass: required workspace; fused: optional output)
  R* term_stats;     // [E_TERM][B]
  R* temp;           // [B] or null (fused only)
  int32_t* status;
  const R* alpha;    // [B] (Linearize backward only: terminal cost update at the end of the chain)
};

// voff (small models, see forward_sweep_body): the lane's byte offset of row e, bo + e * rb, as VGPR values -- the stores then go
// through one buffer window per cell and need no scalar address arithmetic (5 - 6 scalar instructions per row otherwise, and
// the lone wave of the small-batch regime pays 4 clocks for each of them).
template <class M, typename R, typename S_>
I2C_FN void store_cell(const Consts<M, R>& c, const CellArgs<R, S_>& a, const int t, const int b, const R* mu, const R* S,
                       const R* ctl, const R* mz, const R* Sz, const R cm, const R cv, const unsigned* voff = nullptr) {
  using C = Consts<M, R>;
  constexpr int NZ = C::NZ, D = C::D;
  const long B = c.B;
  if (voff) {
    const unsigned rb = (unsigned)(B * sizeof(S_));
    const Window wp = make_window(a.post + (unsigned long)c.row(t) * C::E_POST * B, (unsigned long)C::E_POST * rb);
#pragma unroll
    for (int e = 0; e < D; ++e) wst(wp, 0u, voff[e], (S_)mu[e]);
#pragma unroll
    for (int e = 0; e < sym(D); ++e) wst(wp, 0u, voff[D + e], (S_)S[e]);
#pragma unroll
    for (int e = 0; e < C::E_POST - D - sym(D); ++e) wst(wp, 0u, voff[D + sym(D) + e], (S_)ctl[e]);
    if (a.zpost) {
      const Window wz = make_window(a.zpost + (unsigned long)t * C::E_ZPOST * B, (unsigned long)C::E_ZPOST * rb);
#pragma unroll
      for (int k = 0; k < NZ; ++k) wst(wz, 0u, voff[k], (S_)mz[k]);
#pragma unroll
      for (int k = 0; k < sym(NZ); ++k) wst(wz, 0u, voff[NZ + k], (S_)Sz[k]);
    }
    if (a.cell_stats) {
      a.cell_stats[((long)t * 2 + 0) * B + b] = cm;
      a.cell_stats[((long)t * 2 + 1) * B + b] = cv;
    }
    return;
  }
  S_* out = a.post + ((long)c.row(t) * C::E_POST) * B + b;
#pragma unroll
  for (int e = 0; e < D; ++e) out[(long)e * B] = (S_)mu[e];
#pragma unroll
  for (int e = 0; e < sym(D); ++e) out[(long)(D + e) * B] = (S_)S[e];
#pragma unroll
  for (int e = 0; e < C::E_POST - D - sym(D); ++e) out[(long)(D + sym(D) + e) * B] = (S_)ctl[e];
  if (a.zpost) {
    S_* zo = a.zpost + ((long)t * C::E_ZPOST) * B + b;
#pragma unroll
    for (int k = 0; k < NZ; ++k) zo[(long)k * B] = (S_)mz[k];
#pragma unroll
    for (int k = 0; k < sym(NZ); ++k) zo[(long)(NZ + k) * B] = (S_)Sz[k];
  }
  if (a.cell_stats) {
    a.cell_stats[((long)t * 2 + 0) * B + b] = cm;
    a.cell_stats[((long)t * 2 + 1) * B + b] = cv;
  }
}

// pass (2): one lane per (t, b)
template <class M, typename R, typename S_ = R>
I2C_HD inline void backward_cell_body(const Consts<M, R>& c, const CellArgs<R, S_>& a, const int t, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZ = C::NZ, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  const long B = c.B;
  const S_* in = a.fwd + ((long)t * C::E_FWD) * B + b;
  const S_* xin = a.xm + ((long)t * C::E_XM) * B + b;

  R mu[D], S[sym(D)], J[D * NX], dm[NX], dS[sym(NX)], m3m[NX], S3m[sym(NX)];
#pragma unroll
  for (int e = 0; e < D; ++e) mu[e] = in[(long)e * B];
#pragma unroll
  for (int e = 0; e < sym(D); ++e) S[e] = in[(long)(D + e) * B];
#pragma unroll
  for (int e = 0; e < NX; ++e) {
    m3m[e] = xin[(long)e * B];
    dm[e] = m3m[e] - in[(long)(O_MU3 + e) * B];
  }
#pragma unroll
  for (int e = 0; e < sym(NX); ++e) {
    S3m[e] = xin[(long)(NX + e) * B];
    dS[e] = S3m[e] - in[(long)(O_S3 + e) * B];
  }
#pragma unroll
  for (int e = 0; e < D * NX; ++e) J[e] = in[(long)(O_J + e) * B];
  R zt[NZ];
#pragma unroll
  for (int k = 0; k < NZ; ++k) zt[k] = c.z_per_cell ? a.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k];

  R ctl[C::E_POST - D - sym(D)], mz[NZ], Sz[sym(NZ)], cm, cv;
  if (!cell_posterior<M, R>(c, zt, mu, S, J, dm, dS, ctl, mz, Sz, &cm, &cv)) set_status(a.status, b, 7, t);
  store_cell<M, R, S_>(c, a, t, b, mu, S, ctl, mz, Sz, cm, cv);
  if (t == c.T - 1) terminal_obs_stats<M, R>(c, b, m3m, S3m, a.term_stats, a.status);
}

// pass (3): term_stats rows 1, 2 = sum_t m_t, sum_t v_t. `part`/`nparts` split t between the
// cooperating lanes of one trajectory; the caller adds the partial sums in a fixed order.
template <class M, typename R>
I2C_FN void reduce_partial(const Consts<M, R>& c, const R* cell_stats, const int b, const int part, const int nparts,
                           R* m, R* v) {
  const long B = c.B;
  R sm = R(0), sv = R(0);
#pragma unroll 4
  for (int t = part; t < c.T; t += nparts) {
    sm += cell_stats[((long)t * 2 + 0) * B + b];
    sv += cell_stats[((long)t * 2 + 1) * B + b];
  }
  *m = sm;
  *v = sv;
}

// ------------------------------------------------------------------------------------------
// Backward sweep, FUSED form (large batches): one lane walks one trajectory from T-1 to 0 doing
// the whole cell; each forward row is read once and the cost sums stay in registers, so the pass
// moves E_FWD + E_POST elements per cell instead of the two-pass form's ~2x that.
// ------------------------------------------------------------------------------------------
// LEANW: as in chunk_walk_body (no optional outputs, one shared target: a compile-time count of the memory operations of a cell).
template <class M, typename R, bool GRID = false, typename S_ = R, bool LEANW = false>
I2C_HD inline void backward_fused_body(const Consts<M, R>& c, const CellArgs<R, S_>& a_in, const int b) {
  CellArgs<R, S_> a = a_in;
  if (LEANW) a.xm = nullptr, a.zpost = nullptr, a.cell_stats = nullptr;
  const bool z_per_cell = LEANW ? false : c.z_per_cell != 0;
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZ = C::NZ, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  constexpr unsigned W = sizeof(S_);
  const unsigned long B = c.B;
  const int T = c.T;
  const unsigned bo = (unsigned)b * W, rb = (unsigned)(B * W);

  constexpr bool DOUBLE_BUFFER = C::D <= 5;  // see chunk_walk_body
  constexpr bool VOFF = LEANW && !GRID && C::D <= 5;  // row offsets in VGPRs, sincos table in registers (chunk_walk_body)
  constexpr int NOFF = C::E_FWD > C::E_POST ? C::E_FWD : C::E_POST;
  unsigned voff[VOFF ? NOFF : 1];
  if (VOFF) {
#pragma unroll
    for (int e = 0; e < NOFF; ++e) voff[e] = bo + (unsigned)e * rb;
  }
  PolyTab<R> ptab;
  if (VOFF) poly_tab_init(ptab);
  const PolyTab<R>* const tab = VOFF ? &ptab : nullptr;
  R row[C::E_FWD], nxt[DOUBLE_BUFFER ? C::E_FWD : 1];
  R m3m[NX], S3m[sym(NX)];
  {
    const Window w = make_window(a.fwd + (unsigned long)(T - 1) * C::E_FWD * B, (unsigned long)C::E_FWD * rb);
    if (DOUBLE_BUFFER) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) row[e] = (R)wld<S_>(w, VOFF ? 0u : e * rb, VOFF ? voff[e] : bo);
      if (LEANW) {  // settled before the loop (chunk_walk_body)
#pragma unroll
        for (int e = 0; e < C::E_FWD; ++e) row[e] = opaque(row[e]);
      }
      end_of_chain<M, R>(c, a.temp, b, row + O_MU3, row + O_S3, m3m, S3m, a.status);
    } else {
      // Only the filtered terminal state here: the loop below loads EVERY cell's row at the top of its own cell, the last
      // one included. Loaded here and skipped there, the row becomes a loop-carried value and all 100+ doubles of it stay
      // live through every cell to the back edge (that was the 452 B per lane of scratch of the d = 7 fused walk).
      R m3f[NX], S3f[sym(NX)];
#pragma unroll
      for (int e = 0; e < NX; ++e) m3f[e] = (R)wld<S_>(w, (O_MU3 + e) * rb, bo);
#pragma unroll
      for (int e = 0; e < sym(NX); ++e) S3f[e] = (R)wld<S_>(w, (O_S3 + e) * rb, bo);
      end_of_chain<M, R>(c, a.temp, b, m3f, S3f, m3m, S3m, a.status);
    }
  }
  terminal_obs_stats<M, R, GRID>(c, b, m3m, S3m, a.term_stats, a.status);

  R sum_m = R(0), sum_v = R(0);
  for (int t = T - 1; t >= 0; --t) {
    if (DOUBLE_BUFFER) {
      const int tp = t > 0 ? t - 1 : 0;
      const Window w = make_window(a.fwd + (unsigned long)tp * C::E_FWD * B, (unsigned long)C::E_FWD * rb);
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) nxt[DOUBLE_BUFFER ? e : 0] = (R)wld<S_>(w, VOFF ? 0u : e * rb, VOFF ? voff[e] : bo);
    } else {
      const Window w = make_window(a.fwd + (unsigned long)t * C::E_FWD * B, (unsigned long)C::E_FWD * rb);
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) row[e] = (R)wld<S_>(w, e * rb, bo);
    }
    if (a.xm) {
      S_* xo = const_cast<S_*>(a.xm) + ((long)t * C::E_XM) * B + b;
#pragma unroll
      for (int i = 0; i < NX; ++i) xo[(long)i * B] = (S_)m3m[i];
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) xo[(long)(NX + i) * B] = (S_)S3m[i];
    }
    R dm[NX], dS[sym(NX)], zt[NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) dm[i] = m3m[i] - row[O_MU3 + i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) dS[i] = S3m[i] - row[O_S3 + i];
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = z_per_cell ? a.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k];
    R* mu = row;
    R* S = row + D;
    R ctl[C::E_POST - D - sym(D)], mz[NZ], Sz[sym(NZ)], cm, cv;
    if (!cell_posterior<M, R, GRID>(c, zt, mu, S, row + O_J, dm, dS, ctl, mz, Sz, &cm, &cv, tab)) set_status(a.status, b, 7, t);
    store_cell<M, R, S_>(c, a, t, b, mu, S, ctl, mz, Sz, cm, cv, VOFF ? voff : nullptr);
    sum_m += cm;
    sum_v += cv;
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = mu[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = S[i];  // xx block = packed prefix
    if (DOUBLE_BUFFER) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) row[e] = nxt[DOUBLE_BUFFER ? e : 0];
    }
  }
  a.term_stats[B + b] = sum_m;
  a.term_stats[2 * B + b] = sum_v;
}

// ------------------------------------------------------------------------------------------
// Backward sweep, CHUNKED form (small batches): the x-marginal recursion is affine,
//   (m, S)_t = (a_t + Jx_t m_{t+1},  C_t + Jx_t S_{t+1} Jx_t^T),   a_t = mu_x1_f - Jx_t mu_x3_f,
//                                                                   C_t = sig_x1_f - Jx_t sig_x3_f Jx_t^T,
// so maps compose: T cells are cut into NC chunks, (1) every chunk's composite map (a, G, C) is built
// in parallel [one lane per (chunk, b)], (2) one lane per trajectory applies the NC composites to get
// the smoothed state entering every chunk, (3) every chunk is walked in parallel doing the complete
// cell work from its boundary value. The sequential depth drops from T to ~2 T / NC + NC.
// ------------------------------------------------------------------------------------------
template <typename R, typename S_ = R> struct ChunkArgs {
  CellArgs<R, S_> cell;  // fwd, xm (optional out), z, post, zpost, cell_stats (optional out), term_stats, temp, status
  R* comp;           // [NC][NX + NX*NX + sym(NX)][B]  composite maps
  R* bnd;            // [NC][NX + sym(NX)][B]          smoothed state entering each chunk
  R* part;           // [NC][2][B]                     per-chunk cost sums ([NC][3][B] in the Linearize form)
  int n_chunks, chunk_len;
};

template <class M, typename R, typename S_ = R>
I2C_HD inline void chunk_compose_body(const Consts<M, R>& c, const ChunkArgs<R, S_>& a, const int ch, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  const long B = c.B;
  const int t_lo = ch * a.chunk_len, t_hi = (t_lo + a.chunk_len < c.T) ? t_lo + a.chunk_len : c.T;
  R av[NX], G[NX * NX], Cc[sym(NX)];
#pragma unroll
  for (int i = 0; i < NX; ++i) av[i] = R(0);
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < NX; ++j) G[i * NX + j] = i == j ? R(1) : R(0);
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) Cc[i] = R(0);
  using Row = ScanRow<NX, R>;
  auto load = [&](int t, Row& r) {
    const S_* in = a.cell.fwd + ((long)t * C::E_FWD) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) r.mu1[i] = in[(long)i * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) r.S1[i] = in[(long)(D + i) * B];
#pragma unroll
    for (int i = 0; i < NX; ++i) r.m3f[i] = in[(long)(O_MU3 + i) * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) r.S3f[i] = in[(long)(O_S3 + i) * B];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) r.Jx[i] = in[(long)(O_J + i) * B];
  };
  Row cur, nxt;
  load(t_hi - 1, cur);
  for (int t = t_hi - 1; t >= t_lo; --t) {
    load(t > t_lo ? t - 1 : t_lo, nxt);  // one cell ahead: the loads do not depend on the composition
    const R* mu1 = cur.mu1;
    const R* S1 = cur.S1;
    const R* m3f = cur.m3f;
    const R* Jx = cur.Jx;
    R S3n[sym(NX)];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3n[i] = Cc[i] - cur.S3f[i];  // C - sig_x3_f
    // a <- mu1 + Jx (a - m3f);  C <- S1 + Jx (C - S3f) Jx^T;  G <- Jx G
    R an[NX], Gn[NX * NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      R v = mu1[i];
#pragma unroll
      for (int k = 0; k < NX; ++k) v += Jx[i * NX + k] * (av[k] - m3f[k]);
      an[i] = v;
    }
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) Cc[i] = S1[i];
    add_JDJt<NX, NX>(Jx, S3n, Cc);
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        R v = R(0);
#pragma unroll
        for (int k = 0; k < NX; ++k) v += Jx[i * NX + k] * G[k * NX + j];
        Gn[i * NX + j] = v;
      }
#pragma unroll
    for (int i = 0; i < NX; ++i) av[i] = an[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) G[i] = Gn[i];
    cur = nxt;
  }
  constexpr int EC = NX + NX * NX + sym(NX);
  R* out = a.comp + ((long)ch * EC) * B + b;
#pragma unroll
  for (int i = 0; i < NX; ++i) out[(long)i * B] = av[i];
#pragma unroll
  for (int i = 0; i < NX * NX; ++i) out[(long)(NX + i) * B] = G[i];
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) out[(long)(NX + NX * NX + i) * B] = Cc[i];
}

template <class M, typename R, typename S_ = R, bool GRID = false>
I2C_HD inline void chunk_stitch_body(const Consts<M, R>& c, const ChunkArgs<R, S_>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, EC = NX + NX * NX + sym(NX);
  const long B = c.B;
  const S_* last = a.cell.fwd + ((long)(c.T - 1) * C::E_FWD) * B + b;
  R m3f[NX], S3f[sym(NX)], m[NX], S[sym(NX)];
#pragma unroll
  for (int i = 0; i < NX; ++i) m3f[i] = last[(long)(O_MU3 + i) * B];
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) S3f[i] = last[(long)(O_S3 + i) * B];
  end_of_chain<M, R>(c, a.cell.temp, b, m3f, S3f, m, S, a.cell.status);
  terminal_obs_stats<M, R, GRID>(c, b, m, S, a.cell.term_stats, a.cell.status);
  R av[NX], G[NX * NX], Cc[sym(NX)], nav[NX], nG[NX * NX], nCc[sym(NX)];
  auto loadc = [&](int ch, R* la, R* lG, R* lC) {
    const R* cp = a.comp + ((long)(ch > 0 ? ch : 0) * EC) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) la[i] = cp[(long)i * B];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) lG[i] = cp[(long)(NX + i) * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) lC[i] = cp[(long)(NX + NX * NX + i) * B];
  };
  loadc(a.n_chunks - 1, av, G, Cc);
  for (int ch = a.n_chunks - 1; ch >= 0; --ch) {
    loadc(ch - 1, nav, nG, nCc);  // next composite, one step ahead
    R* bo = a.bnd + ((long)ch * C::E_XM) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) bo[(long)i * B] = m[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) bo[(long)(NX + i) * B] = S[i];
    R mn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      R v = av[i];
#pragma unroll
      for (int k = 0; k < NX; ++k) v += G[i * NX + k] * m[k];
      mn[i] = v;
    }
    add_JDJt<NX, NX>(G, S, Cc);  // C + G S G^T
#pragma unroll
    for (int i = 0; i < NX; ++i) m[i] = mn[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S[i] = Cc[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) av[i] = nav[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) G[i] = nG[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) Cc[i] = nCc[i];
  }
}

// LEANW: the common case fixed at compile time (no smoothed-state / observed-marginal / per-cell-cost outputs, one shared target):
// with the optional stores behind run-time branches the compiler cannot count the memory operations between the row prefetch and
// its use, so its s_waitcnt at the top of a cell also waits for the PREVIOUS cell's stores to be acknowledged (vmcnt is one
// in-order counter) -- 41 % of the walk's cycles (profiles/r3_pendulum_B4096_chunked_sq_summary.txt).
// GRID: the Gauss-Hermite tensor-grid transform in the cell (GaussHermiteQuadrature: the chunked schedule of that rule)
template <class M, typename R, typename S_ = R, bool LEANW = false, bool GRID = false>
I2C_HD inline void chunk_walk_body(const Consts<M, R>& c, const ChunkArgs<R, S_>& a, const int ch, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZ = C::NZ, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  constexpr unsigned W = sizeof(S_);
  const unsigned long B = c.B;
  const unsigned bo = (unsigned)b * W, rb = (unsigned)(B * W);
  const int t_lo = ch * a.chunk_len, t_hi = (t_lo + a.chunk_len < c.T) ? t_lo + a.chunk_len : c.T;
  CellArgs<R, S_> ca = a.cell;
  if (LEANW) ca.xm = nullptr, ca.zpost = nullptr, ca.cell_stats = nullptr;
  const bool z_per_cell = LEANW ? false : c.z_per_cell != 0;
  R m3m[NX], S3m[sym(NX)];
  {
    const R* bi = a.bnd + ((long)ch * C::E_XM) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = bi[(long)i * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = bi[(long)(NX + i) * B];
  }
  // The next cell's forward row is fetched while this cell computes -- but only where a second row fits: for d >= 6 a
  // row is 100+ doubles, two of them overflow the register file into scratch and the walk then waits on scratch traffic
  // 80 % of the time (SQ counters); there the row is loaded at the top of its own cell.
  constexpr bool DOUBLE_BUFFER = C::D <= 5;
  // per-row byte offsets as VGPR values (small models; see forward_sweep_body and store_cell)
  constexpr bool VOFF = I2C_WALK_VOFF && C::D <= 5;
  constexpr int NOFF = C::E_FWD > C::E_POST ? (C::E_FWD > C::E_ZPOST ? C::E_FWD : C::E_ZPOST) : (C::E_POST > C::E_ZPOST ? C::E_POST : C::E_ZPOST);
  unsigned voff[VOFF ? NOFF : 1];
  if (VOFF) {
#pragma unroll
    for (int e = 0; e < NOFF; ++e) voff[e] = bo + (unsigned)e * rb;
  }
  PolyTab<R> ptab;  // sincos coefficients as VGPR values (24 literal moves per cell otherwise), same condition
  if (VOFF) poly_tab_init(ptab);
  const PolyTab<R>* const tab = VOFF ? &ptab : nullptr;
  R row[C::E_FWD], nxt[DOUBLE_BUFFER ? C::E_FWD : 1];
  if (DOUBLE_BUFFER) {
    const Window w = make_window(ca.fwd + (unsigned long)(t_hi - 1) * C::E_FWD * B, (unsigned long)C::E_FWD * rb);
#pragma unroll
    for (int e = 0; e < C::E_FWD; ++e) row[e] = (R)wld<S_>(w, VOFF ? 0u : e * rb, VOFF ? voff[e] : bo);
    // settle them before the loop: the waitcnt pass joins the loop-entry state with the back-edge state, and loads still pending
    // on the entry path (youngest operations there, but older than a cell's stores on the back edge) make it wait for
    // vmcnt(0) -- the previous cell's stores -- at the top of EVERY cell (see forward_sweep_body)
    if (LEANW) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) row[e] = opaque(row[e]);
    }
  }
  R sum_m = R(0), sum_v = R(0);
  for (int t = t_hi - 1; t >= t_lo; --t) {
    {
      const int tp = DOUBLE_BUFFER ? (t > t_lo ? t - 1 : t_lo) : t;
      const Window w = make_window(ca.fwd + (unsigned long)tp * C::E_FWD * B, (unsigned long)C::E_FWD * rb);
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e)
        (DOUBLE_BUFFER ? nxt[DOUBLE_BUFFER ? e : 0] : row[e]) = (R)wld<S_>(w, VOFF ? 0u : e * rb, VOFF ? voff[e] : bo);
    }
    if (ca.xm) {
      S_* xo = const_cast<S_*>(ca.xm) + ((long)t * C::E_XM) * B + b;
#pragma unroll
      for (int i = 0; i < NX; ++i) xo[(long)i * B] = (S_)m3m[i];
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) xo[(long)(NX + i) * B] = (S_)S3m[i];
    }
    R dm[NX], dS[sym(NX)], zt[NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) dm[i] = m3m[i] - row[O_MU3 + i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) dS[i] = S3m[i] - row[O_S3 + i];
    if (VOFF && z_per_cell) {  // (NZ <= E_FWD: the row offsets cover it; R-typed rows: 8 / sizeof(S_) scales them)
      static_assert(NZ <= NOFF, "row offsets");
      const Window wz = make_window(ca.z + (unsigned long)c.row(t) * NZ * B, (unsigned long)NZ * B * sizeof(R));
#pragma unroll
      for (int k = 0; k < NZ; ++k) zt[k] = wld<R>(wz, 0u, voff[k] * (unsigned)(sizeof(R) / W));
    } else {
#pragma unroll
      for (int k = 0; k < NZ; ++k) zt[k] = z_per_cell ? ca.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k];
    }
    R* mu = row;
    R* S = row + D;
    R ctl[C::E_POST - D - sym(D)], mz[NZ], Sz[sym(NZ)], cm, cv;
    if (!cell_posterior<M, R, GRID>(c, zt, mu, S, row + O_J, dm, dS, ctl, mz, Sz, &cm, &cv, tab)) set_status(ca.status, b, 7, t);
    store_cell<M, R, S_>(c, ca, t, b, mu, S, ctl, mz, Sz, cm, cv, VOFF ? voff : nullptr);
    sum_m += cm;
    sum_v += cv;
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = mu[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = S[i];
    if (DOUBLE_BUFFER) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) row[e] = nxt[DOUBLE_BUFFER ? e : 0];
    }
  }
  a.part[((long)ch * 2 + 0) * B + b] = sum_m;
  a.part[((long)ch * 2 + 1) * B + b] = sum_v;
}

// ------------------------------------------------------------------------------------------
// M-step on the temperature (i2c.py:913-963, 1045-1053). One lane per trajectory, O(1) work:
// the sums over t were produced by the backward sweep (term_stats rows 1, 2).
// ------------------------------------------------------------------------------------------
template <typename R> struct MstepArgs {
  const R* term_stats;  // [E_TERM][B]
  R* alpha;             // [B]
  R* stats_out;         // [4][B]
  int update;
};

template <class M, typename R>
I2C_HD inline void mstep_body(const Consts<M, R>& c, const MstepArgs<R>& a, const int b) {
  using C = Consts<M, R>;
  const long B = c.B;
  // row 1 is the alpha statistic sum_t tr(QR (err err^T + sig_z0_m)); in the sigma-point path it IS the plan cost,
  // the Linearize path reports the cost of the graph's cubature transform separately (last row)
  const R stat = a.term_stats[B + b], v = a.term_stats[2 * B + b];
  const R m = c.inference == I2C_INF_LINEARIZE ? a.term_stats[(long)(C::E_TERM - 1) * B + b] : stat;
  R tr = stat, sf = R(C::NZ) * R(c.T);
  if (C::NZT > 0 && c.has_Qf) {
    tr += a.term_stats[b];
    sf += R(C::NZT);
  }
  const R alpha_hat = tr / sf;
  const R alpha = a.alpha[b];
  R alpha_new = alpha;
  if (a.update) {
    if (c.tol >= R(0)) {  // i2c.py:953-959
      const R ratio = alpha_hat / alpha;
      alpha_new = alpha_hat;
      if (ratio < c.tol) alpha_new = c.tol * alpha;
      if (ratio > R(2) - c.tol) alpha_new = (R(2) - c.tol) * alpha;
    }
    a.alpha[b] = alpha_new;
  }
  a.stats_out[b] = alpha_hat;
  a.stats_out[B + b] = alpha_new;
  a.stats_out[2 * B + b] = m;
  a.stats_out[3 * B + b] = v;
}

// ------------------------------------------------------------------------------------------
// Closed-loop propagation (i2c.py:150-199, 1247-1251): forward-only rollout of the controller
// distribution through the model. One lane per trajectory.
// ------------------------------------------------------------------------------------------
template <typename R> struct PropArgs {
  const R* post;   // [T][E_POST][B]
  R* prop;         // [T][E_PROP][B]
  R* prop_stats;   // [3][B]: sum_t of the propagated cost mean / variance, KL of the final state to the terminal prior
  const R* x0;
  const R* sig_x0;
  const R* z;
  const uint8_t* ff;
  int32_t* status;
  const uint8_t* expert;  // [T] ring or null: per-cell use_expert_controller (i2c.py:143,160); null: Consts::use_expert
};

template <class M, typename R, bool GRID = false>
I2C_HD inline void propagate_body(const Consts<M, R>& c, const PropArgs<R>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D;
  const long B = c.B;
  const int T = c.T;
  R mu_x[NX], sig_x[sym(NX)];
#pragma unroll
  for (int i = 0; i < NX; ++i) mu_x[i] = a.x0[i * B + b];
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) sig_x[i] = a.sig_x0[i * B + b];
  // the posterior rows of the next cell are fetched a cell ahead where a second set fits (see forward_sweep_body)
  constexpr bool PREFETCH = C::D <= 5;
  R pri[C::E_PRI];
  if (PREFETCH) {
#pragma unroll
    for (int e = 0; e < C::E_PRI; ++e) pri[e] = a.post[((long)c.row(0) * C::E_POST + e) * B + b];
  }
  // The mode and expert flags of a cell are bytes in global memory: loaded at the top of their own cell they are dependent
  // loads whose full latency -- plus, vmcnt being one in-order counter, the acknowledgement of the previous cell's stores and
  // the arrival of the rows just requested -- is exposed EVERY cell (see forward_sweep_body). Fetched one cell ahead, and
  // everything settled before the loop (a load pending on the loop-entry path makes the waitcnt pass wait for vmcnt(0) at the top).
  unsigned ff_cur = a.ff[c.row(0)], ex_cur = a.expert ? (unsigned)a.expert[c.row(0)] : (unsigned)(c.use_expert != 0);
  ff_cur = opaque(ff_cur);
  ex_cur = opaque(ex_cur);
  if (PREFETCH) {
#pragma unroll
    for (int e = 0; e < C::E_PRI; ++e) pri[e] = opaque(pri[e]);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) mu_x[i] = opaque(mu_x[i]);
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) sig_x[i] = opaque(sig_x[i]);
  // The per-cell target too: a load behind a run-time branch in the middle of the cell leaves an s_waitcnt vmcnt(0) at the join,
  // which every cell pays with the acknowledgement of the stores it has just issued. Small models: fetched a cell ahead and
  // WITHOUT a branch (when there are no per-cell targets the loads read the trajectory's x0 row and are discarded).
  constexpr bool ZPRE = PREFETCH && NZ <= NX * 4;
  R zt_cur[ZPRE ? NZ : 1];
  auto fetch_z = [&](const int row, R* zt) {
    const R* src = c.z_per_cell ? a.z + ((long)row * NZ) * B + b : a.x0 + b;
    const long st = c.z_per_cell ? B : 0;
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const R v = src[(long)k * st];
      zt[k] = c.z_per_cell ? v : c.zg[k];
    }
  };
  if (ZPRE) {
    fetch_z(c.row(0), zt_cur);
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt_cur[k] = opaque(zt_cur[k]);
  }
  R sum_m = R(0), sum_v = R(0);

  for (int t = 0; t < T; ++t) {
    R nxt[PREFETCH ? C::E_PRI : 1];
    const int tn1 = t + 1 < T ? t + 1 : t;  // the next cell (flags), and where a second set of rows fits also its rows
    const int tn = PREFETCH ? tn1 : t;
#pragma unroll
    for (int e = 0; e < C::E_PRI; ++e)
      (PREFETCH ? nxt[PREFETCH ? e : 0] : pri[e]) = a.post[((long)c.row(tn) * C::E_POST + e) * B + b];
    const R* qmu = pri;
    const R* qsig = pri + D;
    const R* Kpost = pri + D + sym(D);

    R mu0[D], S0[sym(D)], Kt[NU * NX], sig_u[sym(NU)];
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) Kt[i] = Kpost[i];
    const unsigned ff_now = ff_cur, ex_now = ex_cur;
    R zt_nxt[ZPRE ? NZ : 1];
    if (ZPRE) fetch_z(c.row(tn1), zt_nxt);
    ff_cur = a.ff[c.row(tn1)];  // the next cell's flags, a whole cell ahead of their use
    if (a.expert) ex_cur = a.expert[c.row(tn1)];
    if (ff_now) {  // i2c.py:155-157: action marginal, but the joint still carries K sig_x (i2c.py:173-179)
#pragma unroll
      for (int p = 0; p < NU; ++p)
#pragma unroll
        for (int q = 0; q <= p; ++q) sig_u[tri(p, q)] = qsig[tri(NX + p, NX + q)];
      joint_from_gain<NX, NU>(mu_x, sig_x, Kt, mu_x, qmu + NX, sig_u, mu0, S0);
    } else {
      if (ex_now != 0) {  // i2c.py:160-167
        R S[sym(NX)], delta[NX];
#pragma unroll
        for (int i = 0; i < sym(NX); ++i) S[i] = qsig[i] + sig_x[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) delta[i] = mu_x[i] - qmu[i];
        bool ok;
        const R rho = pdf_ratio<NX>(S, delta, &ok);
        if (ok) {  // the reference logs the exception and keeps K unscaled (i2c.py:166-167)
#pragma unroll
          for (int i = 0; i < NU * NX; ++i) Kt[i] *= rho;
        }
      }
      // sig_u0_pf = K sig_x0_pf K^T + sig_u0_m - K sig_x0_m K^T  (i2c.py:169-171)
      R q1[sym(NU)], q2[sym(NU)];
      gain_quad<NX, NU>(Kt, sig_x, q1);
      gain_quad<NX, NU>(Kt, qsig, q2);
#pragma unroll
      for (int p = 0; p < NU; ++p)
#pragma unroll
        for (int q = 0; q <= p; ++q) sig_u[tri(p, q)] = q1[tri(p, q)] + qsig[tri(NX + p, NX + q)] - q2[tri(p, q)];
      joint_from_gain<NX, NU>(mu_x, sig_x, Kt, qmu, qmu + NX, sig_u, mu0, S0);
    }
    R* out = a.prop + ((long)t * C::E_PROP) * B + b;
#pragma unroll
    for (int e = 0; e < D; ++e) out[(long)e * B] = mu0[e];
#pragma unroll
    for (int e = 0; e < sym(D); ++e) out[(long)(D + e) * B] = S0[e];

    R L0[sym(D)], rinv[D];
#pragma unroll
    for (int e = 0; e < sym(D); ++e) L0[e] = S0[e];
    if (!chol<D>(L0, rinv)) set_status(a.status, b, 8, t);
    R zt[NZ], mz[NZ], Sz[sym(NZ)];
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = ZPRE ? zt_cur[ZPRE ? k : 0] : (c.z_per_cell ? a.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k]);
    transform<GRID, M, ObsStruct<M>, D, NZ, false>(c.rule_xu, mu0, S0, L0, ObserveF<M, R>{c.params}, mz, Sz, (R*)nullptr);
    R cm, cv;
    gaussian_cost<NZ>(c.QR, c.qr_diag != 0, mz, Sz, zt, &cm, &cv);
    sum_m += cm;
    sum_v += cv;

    transform<GRID, M, DenseStruct<D>, D, NX, false>(c.rule_xu, mu0, S0, L0, DynamicsF<M, R>{c.params}, mu_x, sig_x, (R*)nullptr);
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) sig_x[i] += c.sig_eta_w[i];
#pragma unroll
    for (int e = 0; e < NX; ++e) out[(long)(D + sym(D) + e) * B] = mu_x[e];
#pragma unroll
    for (int e = 0; e < sym(NX); ++e) out[(long)(D + sym(D) + NX + e) * B] = sig_x[e];
    if (ZPRE) {
#pragma unroll
      for (int k = 0; k < NZ; ++k) zt_cur[k] = zt_nxt[ZPRE ? k : 0];
    }
    if (PREFETCH) {
#pragma unroll
      for (int e = 0; e < C::E_PRI; ++e) pri[e] = nxt[PREFETCH ? e : 0];
    }
  }
  a.prop_stats[b] = sum_m;
  a.prop_stats[B + b] = sum_v;
  // KL(x3_pf[T-1] || terminal prior) of covariance control (I2cGraph._maximize, i2c.py:1012-1019, mvn_kl_divergence
  // :1223-1229) from the two Cholesky factors: log det ratio = 2 sum log(L2_ii / L1_ii), tr(S2^-1 S1) = ||L2^-1 L1||_F^2.
  R kl = R(0);
  if (c.has_x_terminal) {
    R L1[sym(NX)], r1[NX], L2[sym(NX)], r2[NX];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) {
      L1[i] = sig_x[i];
      L2[i] = c.sig_x_term[i];
    }
    const bool ok = chol<NX>(L1, r1) && chol<NX>(L2, r2);
    if (!ok) set_status(a.status, b, 8, T - 1);
    R logdet = R(0), tr = R(0), dq[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      logdet += r_log(r1[i]) - r_log(r2[i]);  // log(L2_ii) - log(L1_ii) with r = 1 / L_ii
      dq[i] = c.mu_x_term[i] - mu_x[i];
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {  // column j of L2^-1 L1
      R col[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) col[i] = i >= j ? L1[tri(i, j)] : R(0);
      fsub<NX>(L2, r2, col);
#pragma unroll
      for (int i = 0; i < NX; ++i) tr += col[i] * col[i];
    }
    fsub<NX>(L2, r2, dq);
    R maha = R(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) maha += dq[i] * dq[i];
    kl = R(0.5) * (R(2) * logdet + tr + maha - R(NX));
  }
  a.prop_stats[2 * B + b] = kl;
}

// ------------------------------------------------------------------------------------------
// Cubature Kalman filter step of the MPC state estimator (PartiallyObservedMpcPolicy.filter,
// i2c/policy/mpc.py:125-145): predict the belief through the dynamics with the applied action,
// then innovate on the measurement y. One lane per trajectory.
// ------------------------------------------------------------------------------------------
template <typename R> struct CkfArgs {
  const R* y;       // [NY][B]
  const R* u;       // [NU][B]
  R* mu;            // [NX][B]      in/out
  R* cov;           // [sym NX][B]  in/out
  int32_t* status;  // [B]
};

template <class M, typename R>
I2C_HD inline void ckf_filter_body(const Consts<M, R>& c, const R* sig_zeta, const CkfArgs<R>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NY = M::NY;
  const long B = c.B;
  R mu[NX], S[sym(NX)], L[sym(NX)], rinv[NX], u[NU], y[NY];
#pragma unroll
  for (int i = 0; i < NX; ++i) mu[i] = a.mu[(long)i * B + b];
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) L[i] = S[i] = a.cov[(long)i * B + b];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = a.u[(long)i * B + b];
#pragma unroll
  for (int i = 0; i < NY; ++i) y[i] = a.y[(long)i * B + b];
  bool ok = chol<NX>(L, rinv);
  // prediction (mpc.py:129-137): x-only sigma points, the action is appended unchanged
  R mf[NX], Sf[sym(NX)];
  sp_transform<M, DenseStruct<NX>, NX, NX, false>(c.rule_x, mu, S, L, DynamicsFixedUF<M, R>{c.params, u}, mf, Sf,
                                                  (R*)nullptr);
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) L[i] = Sf[i] = Sf[i] + c.rule_x.W * c.sig_eta[i];
  ok = chol<NX>(L, rinv) && ok;
  // innovation (mpc.py:139-145): K = sig_xy sig_y^{-1}; mu = mu_f + K (y - mu_y); cov = sig_f - K sig_y K^T
  R my[NY], Sy[sym(NY)], Sxy[NX * NY];
  sp_transform<M, MeasStruct<M>, NX, NY, true>(c.rule_x, mf, Sf, L, MeasureF<M, R>{c.params}, my, Sy, Sxy);
#pragma unroll
  for (int i = 0; i < sym(NY); ++i) Sy[i] += sig_zeta[i];
  ok = kalman_update<NX, NY>(mf, Sf, my, Sy, Sxy, y) && ok;
#pragma unroll
  for (int i = 0; i < NX; ++i) a.mu[(long)i * B + b] = mf[i];
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) a.cov[(long)i * B + b] = Sf[i];
  if (!ok) set_status(a.status, b, 9, 0);
}

// ------------------------------------------------------------------------------------------
// Policy rollouts through the (noisy) known model: replaces BaseSim.run / batch_eval
// (i2c/env.py:40-103, BaseKnownSim.forward :180-187) driven by the time-indexed linear-Gaussian
// policies of i2c/policy/linear.py. One lane per rollout n = r * B + b (r-th rollout of trajectory
// b), T sequential steps; the disturbance samples are supplied by the caller as standard normals so
// that a rollout is a deterministic function of its inputs.
// ------------------------------------------------------------------------------------------
template <typename R> struct RolloutArgs {
  const R* post;     // [T][E_POST][B]  posterior + controller written by the backward sweep
  const R* x0;       // [NX][B]
  const R* sig_x0;   // [sym NX][B]
  const R* eps_x0;   // [NX][N] or null: x0 ~ N(x0, sig_x0) (BaseLinear.init_env, env.py:195-197)
  const R* eps_x;    // [T][NX][N] or null: process noise chol(sig_eta) eps (env.py:184-186)
  const R* eps_u;    // [T][NU][N] or null: action noise chol(sigK) eps (linear.py:36-41, 86-90)
  R* xu;             // [T][D][N]  or null: (x_t, u_t)                      (env.py:66-68)
  R* z;              // [T][NZ][N] or null: observe(x_t, u_t)               (env.py:70)
  R* x_final;        // [NX][N]    or null: state after the last step
  R* z_term;         // [NZT][N]   or null: observe_terminal(x_T)           (env.py:73)
  int n_rollouts;    // N = n_rollouts * B
  int policy;        // 0: u = K x + k (TimeIndexedLinearGaussianPolicy); 1: expert, soft weight;
                     // 2: expert, hard weight (ExpertTimeIndexedLinearGaussianPolicy, linear.py:46-90)
};

template <class M, typename R>
I2C_HD inline void rollout_body(const Consts<M, R>& c, const RolloutArgs<R>& a, const int n) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, NZT = C::NZT, D = C::D, NA1 = M::NA > 0 ? M::NA : 1;
  const long B = c.B, N = (long)a.n_rollouts * B;
  const int b = n % c.B;
  const int T = c.T;
  R x[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = a.x0[i * B + b];
  if (a.eps_x0) {
    R L[sym(NX)], rinv[NX];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) L[i] = a.sig_x0[i * B + b];
    chol<NX>(L, rinv);
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) x[i] += L[tri(i, j)] * a.eps_x0[(long)j * N + n];
  }
  R Le[sym(NX)], rinv_e[NX];  // chol(sig_eta), constant
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) Le[i] = c.sig_eta[i];
  if (a.eps_x) chol<NX>(Le, rinv_e);

  for (int t = 0; t < T; ++t) {
    const long es = c.post_es();
    const R* row = a.post + ((long)c.row(t) * C::E_POST) * B + c.post_bo(b);
    R u[NU], Kc[NU * NX];
#pragma unroll
    for (int e = 0; e < NU * NX; ++e) Kc[e] = row[(long)(D + sym(D) + e) * es];
    if (a.policy == 0) {  // u = K x + k
#pragma unroll
      for (int p = 0; p < NU; ++p) {
        R v = row[(long)(C::E_PRI + p) * es];
#pragma unroll
        for (int k = 0; k < NX; ++k) v += Kc[p * NX + k] * x[k];
        u[p] = v;
      }
    } else {  // u = mu_u + w K (x - mu_x), w = exp(-maha/2) (soft) or [maha/2 < 3] (hard); lam = sig_x^{-1}
      R dlt[NX], q[NX], S[sym(NX)], rinv[NX];
#pragma unroll
      for (int k = 0; k < NX; ++k) q[k] = dlt[k] = x[k] - row[(long)k * es];
#pragma unroll
      for (int k = 0; k < sym(NX); ++k) S[k] = row[(long)(D + k) * es];
      chol<NX>(S, rinv);
      fsub<NX>(S, rinv, q);
      R half = R(0);
#pragma unroll
      for (int k = 0; k < NX; ++k) half += q[k] * q[k];
      half *= R(0.5);
      const R w = a.policy == 1 ? r_exp(-half) : (half < R(3) ? R(1) : R(0));
#pragma unroll
      for (int p = 0; p < NU; ++p) {
        R v = R(0);
#pragma unroll
        for (int k = 0; k < NX; ++k) v += Kc[p * NX + k] * dlt[k];
        u[p] = row[(long)(NX + p) * es] + w * v;
      }
    }
    if (a.eps_u) {  // u += chol(sigK) eps
      R Lk[sym(NU)], rk[NU];
#pragma unroll
      for (int k = 0; k < sym(NU); ++k) Lk[k] = row[(long)(C::E_PRI + NU + k) * es];
      chol<NU>(Lk, rk);
#pragma unroll
      for (int p = 0; p < NU; ++p)
#pragma unroll
        for (int q2 = 0; q2 <= p; ++q2) u[p] += Lk[tri(p, q2)] * a.eps_u[((long)t * NU + q2) * N + n];
    }
    R xu[D], sn[NA1], cs[NA1], zt[NZ], xn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xu[i] = x[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) xu[NX + i] = u[i];
#pragma unroll
    for (int q2 = 0; q2 < M::NA; ++q2) r_sincos(xu[M::ang(q2)], &sn[q2], &cs[q2]);
    M::observe(c.params, xu, sn, cs, zt);
    M::dynamics(c.params, xu, sn, cs, xn);
    if (a.xu) {
#pragma unroll
      for (int i = 0; i < D; ++i) a.xu[((long)t * D + i) * N + n] = xu[i];
    }
    if (a.z) {
#pragma unroll
      for (int i = 0; i < NZ; ++i) a.z[((long)t * NZ + i) * N + n] = zt[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xn[i];
    if (a.eps_x) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) x[i] += Le[tri(i, j)] * a.eps_x[((long)t * NX + j) * N + n];
    }
  }
  if (a.x_final) {
#pragma unroll
    for (int i = 0; i < NX; ++i) a.x_final[(long)i * N + n] = x[i];
  }
  if (NZT > 0 && a.z_term) {
    R sn[NA1], cs[NA1], zT[C::NZT1];
#pragma unroll
    for (int q2 = 0; q2 < M::NA; ++q2) r_sincos(x[M::ang(q2)], &sn[q2], &cs[q2]);
    M::observe_terminal(c.params, x, sn, cs, zT);
#pragma unroll
    for (int i = 0; i < NZT; ++i) a.z_term[(long)i * N + n] = zT[i];
  }
}

// ------------------------------------------------------------------------------------------
// Receding-horizon shift of the MPC loop (PartiallyObservedMpcPolicy.__call__, i2c/policy/mpc.py:171-181):
//   u = cells[0].mu_u0_m;  cells.pop(0);  cells.append(deepcopy(cell_init)) with the next target.
// The persistent per-cell buffers are a RING (Consts::t0): popping cell 0 and appending a cell is "advance t0 by one"
// (done by the caller after this kernel) plus ONE fresh row, written here over the row cell 0 occupied -- one lane per
// trajectory, after the first action and its covariance were copied out of that row. The fresh cell keeps the temperature
// `alpha_init` it was copied with (see I2cProblem.alpha_cell) and is in feed-forward mode.
// ------------------------------------------------------------------------------------------
template <typename R> struct ShiftArgs {
  R* post;               // [T][E_POST][B]   ring
  const R* cell_init;    // [E_POST][B]
  R* alpha_cell;         // [T][B] ring, or null
  const R* alpha_init;   // [B] or null
  R* z;                  // [T][NZ][B] ring, or null
  const R* z_new;        // [NZ][B] or null: target of the appended cell (null: the previous last cell's)
  uint8_t* ff;           // [T] ring
  R* action;             // [NU + sym(NU)][B]: cells[0].mu_u0_m, sig_u0_m (packed) BEFORE the shift
};

template <class M, typename R>
I2C_HD inline void mpc_shift_body(const Consts<M, R>& c, const ShiftArgs<R>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D;
  const long B = c.B;
  const int r0 = c.row(0), rl = c.row(c.T - 1);  // rows of the first and of the last cell of the current horizon
  const long es = c.post_es();
  const R* p0 = a.post + ((long)r0 * C::E_POST) * B + c.post_bo(b);
  if (a.action) {
#pragma unroll
    for (int i = 0; i < NU; ++i) a.action[(long)i * B + b] = p0[(long)(NX + i) * es];
#pragma unroll
    for (int p = 0; p < NU; ++p)
#pragma unroll
      for (int q = 0; q <= p; ++q) a.action[(long)(NU + tri(p, q)) * B + b] = p0[(long)(D + tri(NX + p, NX + q)) * es];
  }
  // (the fresh cell itself -- cell_init, one cell block in the layout of `post` -- is a plain block copy: Impl::shift)
  if (a.alpha_cell) a.alpha_cell[(long)r0 * B + b] = a.alpha_init[b];
  if (a.z) {
#pragma unroll
    for (int k = 0; k < NZ; ++k)
      a.z[((long)r0 * NZ + k) * B + b] = a.z_new ? a.z_new[(long)k * B + b] : a.z[((long)rl * NZ + k) * B + b];
  }
  if (b == 0) a.ff[r0] = (uint8_t)1;
}

}  // namespace i2c
