// Linearize inference (I2cCell._forward_msgs_linearize / _backward_msgs_linearize, i2c/i2c.py:244-348, 449-542), one
// trajectory per lane, same buffers and layouts as the sigma-point path (i2c_cell.hpp).
//
// The reference linearises the cost observation about the prior mean and the dynamics about the updated mean:
//   mu_z = h(mu), sig_z = H S H^T (+ alpha QR^-1), sig_xz = S H^T      (H = dh/d[x;u], i2c/env_def.py:278-287 ...)
//   mu_x' = f(mu), sig_x' = AB S AB^T + sig_eta, sig_xy = S AB^T        (AB = df/d[x;u], i2c/model.py:158-164)
// and then runs the same Kalman-style update / RTS smoother as the sigma-point path, so this file only supplies the
// "transform" (function value + Jacobian) and the cells built on it. Jacobians come from forward-mode differentiation
// of the SAME model functors (i2c_models.hpp) with single-tangent dual numbers, one input direction per pass; the
// tangent seeds are compile-time constants after unrolling, so pass-through rows fold to 0 / 1 at compile time.
// (The reference: hand-written observation Jacobians, autograd for the dynamics, i2c/env_autograd.py:22,57,170.)
#pragma once
#include <type_traits>
#include "i2c_cell.hpp"

namespace i2c {

template <typename T> struct Dual {
  T v, d;
  I2C_HD Dual() = default;
  I2C_HD Dual(T v_) : v(v_), d(T(0)) {}
  I2C_HD Dual(T v_, T d_) : v(v_), d(d_) {}
  template <typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value && !std::is_same<S, T>::value>::type>
  I2C_HD Dual(S s) : v(T(s)), d(T(0)) {}
  I2C_HD friend Dual operator+(const Dual& a, const Dual& b) { return Dual(a.v + b.v, a.d + b.d); }
  I2C_HD friend Dual operator-(const Dual& a, const Dual& b) { return Dual(a.v - b.v, a.d - b.d); }
  I2C_HD friend Dual operator*(const Dual& a, const Dual& b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
  I2C_HD friend Dual operator/(const Dual& a, const Dual& b) {
    const T q = a.v / b.v;
    return Dual(q, (a.d - q * b.d) / b.v);
  }
  I2C_HD friend Dual operator-(const Dual& a) { return Dual(-a.v, -a.d); }
  I2C_HD Dual& operator+=(const Dual& b) { return *this = *this + b; }
  I2C_HD Dual& operator-=(const Dual& b) { return *this = *this - b; }
  I2C_HD Dual& operator*=(const Dual& b) { return *this = *this * b; }
};
// derivative of clip: 1 strictly inside the limits, 0 on and outside them (autograd's rule for np.clip)
template <typename T> I2C_FN Dual<T> r_clip(const Dual<T>& x, const Dual<T>& lo, const Dual<T>& hi) {
  const bool inside = x.v > lo.v && x.v < hi.v;
  return Dual<T>(r_clip(x.v, lo.v, hi.v), inside ? x.d : T(0));
}
template <typename T> I2C_FN Dual<T> r_rcp(const Dual<T>& x) {
  const T r = r_rcp(x.v);
  return Dual<T>(r, -x.d * r * r);
}

template <typename T> I2C_FN Dual<T> r_exp(const Dual<T>& x) {
  const T e = r_exp(x.v);
  return Dual<T>(e, e * x.d);
}
// (the operations a model functor may apply to its arguments: + - * /, r_clip, r_rcp, r_exp, and the sines / cosines it is handed)

enum { FN_DYNAMICS = 0, FN_OBSERVE = 1, FN_OBSERVE_TERMINAL = 2 };
template <class M, int FN, typename T> I2C_FN void call_model(const T* p, const T* x, const T* sn, const T* cs, T* y) {
  if (FN == FN_DYNAMICS) M::dynamics(p, x, sn, cs, y);
  if (FN == FN_OBSERVE) M::observe(p, x, sn, cs, y);
  if (FN == FN_OBSERVE_TERMINAL) M::observe_terminal(p, x, sn, cs, y);
}

// y = f(m) and Jac[k * DIN + j] = d y_k / d m_j for one of the model callbacks.
template <class M, int FN, int DIN, int DOUT, typename R>
I2C_FN void value_and_jacobian(const R* params, const R* m, R* y, R* Jac) {
  constexpr int NA = M::NA > 0 ? M::NA : 1, NP1 = M::NP > 0 ? M::NP : 1;
  R sn[NA], cs[NA];
#pragma unroll
  for (int a = 0; a < M::NA; ++a) r_sincos(m[M::ang(a)], &sn[a], &cs[a]);
  call_model<M, FN, R>(params, m, sn, cs, y);
  Dual<R> pd[NP1];
#pragma unroll
  for (int i = 0; i < M::NP; ++i) pd[i] = Dual<R>(params[i]);
#pragma unroll
  for (int j = 0; j < DIN; ++j) {
    Dual<R> x[DIN], s[NA], c[NA], yy[DOUT];
#pragma unroll
    for (int i = 0; i < DIN; ++i) x[i] = Dual<R>(m[i], i == j ? R(1) : R(0));
#pragma unroll
    for (int a = 0; a < M::NA; ++a) {
      s[a] = Dual<R>(sn[a], M::ang(a) == j ? cs[a] : R(0));
      c[a] = Dual<R>(cs[a], M::ang(a) == j ? -sn[a] : R(0));
    }
    call_model<M, FN, Dual<R>>(pd, x, s, c, yy);
#pragma unroll
    for (int k = 0; k < DOUT; ++k) Jac[k * DIN + j] = yy[k].d;
  }
}

// Linearised Gaussian push-through: my = f(m), Sxy = Sin Jac^T [DIN x DOUT], Sy = Jac Sin Jac^T (packed).
template <class M, int FN, int DIN, int DOUT, typename R>
I2C_FN void lin_transform(const R* params, const R* m, const R* Sin, R* my, R* Sy, R* Sxy, R* Jac) {
  value_and_jacobian<M, FN, DIN, DOUT, R>(params, m, my, Jac);
#pragma unroll
  for (int i = 0; i < DIN; ++i)
#pragma unroll
    for (int k = 0; k < DOUT; ++k) {
      R v = R(0);
#pragma unroll
      for (int j = 0; j < DIN; ++j) v += Sin[tri_any(i, j)] * Jac[k * DIN + j];
      Sxy[i * DOUT + k] = v;
    }
#pragma unroll
  for (int k = 0; k < DOUT; ++k)
#pragma unroll
    for (int l = 0; l <= k; ++l) {
      R v = R(0);
#pragma unroll
      for (int i = 0; i < DIN; ++i) v += Jac[k * DIN + i] * Sxy[i * DOUT + l];
      Sy[tri(k, l)] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Forward sweep (i2c.py:244-348). Differences to the sigma-point cell: the feedback gain is scaled by the pdf ratio only
// with the expert controller (:259-265), both transforms are linearisations, there is no terminal update here (it
// happens at the end of the backward chain, :475-491) and nothing is symmetrised.
// ------------------------------------------------------------------------------------------
template <class M, typename R>
I2C_HD inline void forward_lin_body(const Consts<M, R>& c, const FwdArgs<R>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D;
  const long B = c.B;
  const int T = c.T;
  int fail = 0;
  R mu_x[NX], sig_x[sym(NX)];
#pragma unroll
  for (int i = 0; i < NX; ++i) mu_x[i] = a.x0[(long)i * B + b];
#pragma unroll
  for (int i = 0; i < sym(NX); ++i) sig_x[i] = a.sig_x0[(long)i * B + b];
  const R alpha_traj = a.alpha[b];
  // Small models: everything a cell reads that does not depend on the recursion -- its prior rows, temperature, target, mode and
  // expert flags -- is fetched ONE CELL AHEAD, without branches (a discarded dummy where an option is off, the choice made where
  // the value is used), and the first set is settled before the loop (see propagate_body / DESIGN.md "Waits on memory").
  constexpr bool PRE = C::D <= 5;
  struct Pre {
    R pri[PRE ? C::E_PRI : 1], zt[PRE ? NZ : 1], alpha;
    unsigned ff, ex;
  } nx;
  auto fetch = [&](const int row, Pre& f) {
    const R* pr = a.prior + ((long)row * C::E_POST) * B + b;
#pragma unroll
    for (int e = 0; e < C::E_PRI; ++e) f.pri[e] = pr[(long)e * B];
    f.alpha = (a.alpha_cell ? a.alpha_cell + (long)row * B : a.alpha)[b];
    const R* zs = c.z_per_cell ? a.z + ((long)row * NZ) * B + b : a.x0 + b;
    const long zst = c.z_per_cell ? B : 0;
#pragma unroll
    for (int k = 0; k < NZ; ++k) f.zt[k] = zs[(long)k * zst];
    f.ff = a.ff[row];
    f.ex = (a.expert ? a.expert : a.ff)[row];
  };
  if (PRE) {
    fetch(c.row(0), nx);
#pragma unroll
    for (int e = 0; e < C::E_PRI; ++e) nx.pri[e] = opaque(nx.pri[e]);
#pragma unroll
    for (int k = 0; k < NZ; ++k) nx.zt[k] = opaque(nx.zt[k]);
    nx.alpha = opaque(nx.alpha), nx.ff = opaque(nx.ff), nx.ex = opaque(nx.ex);
#pragma unroll
    for (int i = 0; i < NX; ++i) mu_x[i] = opaque(mu_x[i]);
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) sig_x[i] = opaque(sig_x[i]);
  }

  for (int t = 0; t < T; ++t) {
    const int tr = c.row(t);  // row of the persistent per-cell buffers (ring of the MPC loop, Consts::t0)
    const R* pri_g = a.prior + ((long)tr * C::E_POST) * B + b;
    R* out = a.fwd + ((long)t * C::E_FWD) * B + b;
    Pre cur;
    if (PRE) {
      cur = nx;
      fetch(c.row(t + 1 < T ? t + 1 : t), nx);
    }
    auto pri = [&](const int e) { return PRE ? cur.pri[PRE ? e : 0] : pri_g[(long)e * B]; };
    const R alpha = PRE ? cur.alpha : (a.alpha_cell ? a.alpha_cell[(long)tr * B + b] : alpha_traj);
    const bool ff_now = PRE ? cur.ff != 0 : a.ff[tr] != 0;
    const bool ex_now = a.expert ? (PRE ? cur.ex != 0 : a.expert[tr] != 0) : c.use_expert != 0;
    R zt[NZ];
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = c.z_per_cell ? (PRE ? cur.zt[PRE ? k : 0] : a.z[((long)tr * NZ + k) * B + b]) : c.zg[k];

    // ---- 1. joint prior over (x, u) (i2c.py:249-276) ----
    R mu0[D], S0[sym(D)];
    if (ff_now) {
#pragma unroll
      for (int i = 0; i < NX; ++i) mu0[i] = mu_x[i];
#pragma unroll
      for (int i = NX; i < D; ++i) mu0[i] = pri(i);
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
          S0[tri(i, j)] = (i < NX) ? sig_x[tri(i, j)] : (j >= NX ? pri(D + tri(i, j)) : R(0));
    } else {
      R pmu[D], psig[sym(D)], Kt[NU * NX];
#pragma unroll
      for (int i = 0; i < D; ++i) pmu[i] = pri(i);
#pragma unroll
      for (int i = 0; i < sym(D); ++i) psig[i] = pri(D + i);
#pragma unroll
      for (int i = 0; i < NU * NX; ++i) Kt[i] = pri(D + sym(D) + i);
      if (ex_now) {
        R S[sym(NX)], delta[NX];
#pragma unroll
        for (int i = 0; i < sym(NX); ++i) S[i] = psig[i] + sig_x[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) delta[i] = mu_x[i] - pmu[i];
        bool ok;
        const R rho = pdf_ratio<NX>(S, delta, &ok);
        fail = note_failure(fail, ok, 2, t);
#pragma unroll
        for (int i = 0; i < NU * NX; ++i) Kt[i] *= rho;
      }
      R sig_u[sym(NU)];
      gain_quad<NX, NU>(Kt, sig_x, sig_u);
#pragma unroll
      for (int p = 0; p < NU; ++p)
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          R v = psig[tri(NX + p, NX + q)] + sig_u[tri(p, q)];
#pragma unroll
          for (int k = 0; k < NX; ++k) v -= Kt[p * NX + k] * psig[tri(NX + q, k)];
          sig_u[tri(p, q)] = v;
        }
      joint_from_gain<NX, NU>(mu_x, sig_x, Kt, pmu, pmu + NX, sig_u, mu0, S0);
    }
    if (a.prior_out) {
      R* po = a.prior_out + ((long)t * (D + sym(D))) * B + b;
#pragma unroll
      for (int e = 0; e < D; ++e) po[(long)e * B] = mu0[e];
#pragma unroll
      for (int e = 0; e < sym(D); ++e) po[(long)(D + e) * B] = S0[e];
    }

    // ---- 2. cost observation linearised about the prior mean (i2c.py:281-305) ----
    {
      R mz[NZ], Sz[sym(NZ)], Sxz[D * NZ], EF[NZ * D];
      lin_transform<M, FN_OBSERVE, D, NZ, R>(c.params, mu0, S0, mz, Sz, Sxz, EF);
#pragma unroll
      for (int i = 0; i < sym(NZ); ++i) Sz[i] += alpha * c.sig_xi0[i];
      fail = note_failure(fail, kalman_update<D, NZ>(mu0, S0, mz, Sz, Sxz, zt), 3, t);
    }
#pragma unroll
    for (int e = 0; e < D; ++e) out[(long)e * B] = mu0[e];
#pragma unroll
    for (int e = 0; e < sym(D); ++e) out[(long)(D + e) * B] = S0[e];

    // ---- 3. dynamics linearised about the updated mean (i2c.py:321-341) ----
    R Sxy[D * NX];
    {
      R AB[NX * D];
      lin_transform<M, FN_DYNAMICS, D, NX, R>(c.params, mu0, S0, mu_x, sig_x, Sxy, AB);
    }
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) sig_x[i] += c.sig_eta[i];
    R L3[sym(NX)], rinv3[NX];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) L3[i] = sig_x[i];
    fail = note_failure(fail, chol<NX>(L3, rinv3), 5, t);
#pragma unroll
    for (int i = 0; i < D; ++i) {  // J = sig_xy sig_x3^{-1}
      fsub<NX>(L3, rinv3, &Sxy[i * NX]);
      bsub<NX>(L3, rinv3, &Sxy[i * NX]);
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) out[(long)(D + sym(D) + e) * B] = mu_x[e];
#pragma unroll
    for (int e = 0; e < sym(NX); ++e) out[(long)(D + sym(D) + NX + e) * B] = sig_x[e];
#pragma unroll
    for (int e = 0; e < D * NX; ++e) out[(long)(D + sym(D) + NX + sym(NX) + e) * B] = Sxy[e];
  }
  if (fail != 0 && a.status[b] == 0) a.status[b] = fail;
}

// ------------------------------------------------------------------------------------------
// Backward sweep (i2c.py:449-542), one lane walks one trajectory from T-1 to 0.
// End of chain: covariance control pins the smoothed terminal state to (mu_x_terminal, sig_x_terminal) (:453-472);
// with a terminal cost the linearised terminal observation is applied HERE (:475-491); otherwise pass-through.
// Per cell: RTS update and controller as in the sigma-point path; the marginal observation is h(mu) with covariance
// C sig_xx C^T + D sig_uu D^T -- no x-u cross terms (:537-540) -- and feeds the alpha M-step (:680-683), while the
// plan cost is evaluated with the graph's cubature transform (i2c.py:841-844, 1034-1053).
// term_stats rows: 0 = terminal trace, 1 = sum_t alpha statistic, 2 = sum_t cost variance, last = sum_t cost mean.
// ------------------------------------------------------------------------------------------
// End of the chain of the Linearize backward sweep (i2c.py:453-501): the smoothed terminal state (m3m, S3m) -- pinned by a terminal
// state prior, or updated by the terminal cost observation, or the filtered one -- and the terminal observation statistics
// (term_stats rows 0, 3..).
template <class M, typename R>
I2C_FN void lin_end_of_chain(const Consts<M, R>& c, const CellArgs<R>& a, const int b, R* m3m, R* S3m) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, D = C::D, NZT = C::NZT, NT = C::NZT1;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX;
  const long B = c.B;
  const int T = c.T;
  const R alpha = a.alpha[b];
  {
    const R* in = a.fwd + ((long)(T - 1) * C::E_FWD) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = in[(long)(O_MU3 + i) * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = in[(long)(O_S3 + i) * B];
  }
  R trT = R(0);
  R xiT[sym(NT)];  // the sig_xi_terminal added to sig_z3_m (i2c.py:460, 488, 497)
#pragma unroll
  for (int i = 0; i < sym(NT); ++i) xiT[i] = R(0);
  if (c.has_x_terminal) {
    if (NZT > 0 && c.has_Qf) {
      // the back-calculated sig_xi_terminal (the Lagrange multiplier of the pinned covariance, i2c.py:455-462): with
      // Szx = E S3f and MP = Szx Szx^T,  sig_z = MP (Szx (S3f - S_T) Szx^T)^-1 MP  and  sig_xi_terminal = sig_z - E S3f E^T
      // (the middle factor is symmetric but in general indefinite: chol_signed)
      R mzt[NT], Szt[sym(NT)], Sxzt[NX * NT], E[NT * NX], W[NX * NT], mid[sym(NT)], rinv[NT], sgn[NT], Y[NT * NT];
      lin_transform<M, FN_OBSERVE_TERMINAL, NX, NT, R>(c.params, m3m, S3m, mzt, Szt, Sxzt, E);
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          R v = R(0);
#pragma unroll
          for (int l = 0; l < NX; ++l) v += (S3m[tri_any(i, l)] - c.sig_x_term[tri_any(i, l)]) * Sxzt[l * NT + k];
          W[i * NT + k] = v;
        }
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int l = 0; l < NT; ++l) {
          R vm = R(0), vp = R(0);
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            vm += Sxzt[i * NT + k] * W[i * NT + l];
            vp += Sxzt[i * NT + k] * Sxzt[i * NT + l];
          }
          if (l <= k) mid[tri(k, l)] = vm;
          Y[l * NT + k] = vp;  // row l = column l of the symmetric MP
        }
      if (!chol_signed<NT>(mid, rinv, sgn)) set_status(a.status, b, 6, T - 1);
#pragma unroll
      for (int l = 0; l < NT; ++l) fsub<NT>(mid, rinv, &Y[l * NT]);  // row l <- L^-1 MP[:, l]
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int l = 0; l <= k; ++l) {
          R v = R(0);
#pragma unroll
          for (int r = 0; r < NT; ++r) v += sgn[r] * Y[k * NT + r] * Y[l * NT + r];
          xiT[tri(k, l)] = v - Szt[tri(k, l)];
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = c.mu_x_term[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = c.sig_x_term[i];
  } else if (NZT > 0 && c.has_Qf) {
    R mzt[NT], Szt[sym(NT)], Sxzt[NX * NT], E[NT * NX];
    lin_transform<M, FN_OBSERVE_TERMINAL, NX, NT, R>(c.params, m3m, S3m, mzt, Szt, Sxzt, E);
#pragma unroll
    for (int i = 0; i < sym(NT); ++i) {
      xiT[i] = alpha * c.sig_xiT0[i];
      Szt[i] += xiT[i];
    }
    if (!kalman_update<NX, NT>(m3m, S3m, mzt, Szt, Sxzt, c.zg_term)) set_status(a.status, b, 6, T - 1);
  }
  if (NZT > 0 && c.has_Qf) {  // mu_z3_m, sig_z3_m = E sig_x3_m E^T + sig_xi_terminal (i2c.py:499-501), alpha statistic :989-992
    R mzt[NT], Szt[sym(NT)], Sxzt[NX * NT], E[NT * NX], tv;
    lin_transform<M, FN_OBSERVE_TERMINAL, NX, NT, R>(c.params, m3m, S3m, mzt, Szt, Sxzt, E);
#pragma unroll
    for (int i = 0; i < sym(NT); ++i) Szt[i] += xiT[i];
    gaussian_cost<NT>(c.Qf, c.qf_diag != 0, mzt, Szt, c.zg_term, &trT, &tv);
#pragma unroll
    for (int k = 0; k < NT; ++k) a.term_stats[(long)(3 + k) * B + b] = mzt[k];
#pragma unroll
    for (int k = 0; k < sym(NT); ++k) a.term_stats[(long)(3 + NT + k) * B + b] = Szt[k];
  }
  a.term_stats[b] = trT;

}

// One cell of the Linearize backward sweep (i2c.py:503-542) given its forward row `ld(e)`, its target zt and the smoothed next
// state, which it replaces by this cell's; adds the cell's alpha statistic, plan cost and cost variance to the sums.
template <class M, typename R, class LD>
I2C_FN void lin_backward_cell(const Consts<M, R>& c, const CellArgs<R>& a, const int t, const int b, const LD& ld, const R* zt_in,
                              R* m3m, R* S3m, R& sum_a, R& sum_m, R& sum_v) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZ = C::NZ, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  const long B = c.B;
  {
    if (a.xm) {
      R* xo = const_cast<R*>(a.xm) + ((long)t * C::E_XM) * B + b;
#pragma unroll
      for (int i = 0; i < NX; ++i) xo[(long)i * B] = m3m[i];
#pragma unroll
      for (int i = 0; i < sym(NX); ++i) xo[(long)(NX + i) * B] = S3m[i];
    }
    R mu[D], S[sym(D)], J[D * NX], dm[NX], dS[sym(NX)], zt[NZ];
#pragma unroll
    for (int e = 0; e < D; ++e) mu[e] = ld(e);
#pragma unroll
    for (int e = 0; e < sym(D); ++e) S[e] = ld(D + e);
#pragma unroll
    for (int e = 0; e < NX; ++e) dm[e] = m3m[e] - ld(O_MU3 + e);
#pragma unroll
    for (int e = 0; e < sym(NX); ++e) dS[e] = S3m[e] - ld(O_S3 + e);
#pragma unroll
    for (int e = 0; e < D * NX; ++e) J[e] = ld(O_J + e);
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = zt_in[k];

    // RTS update, controller and the cubature cost of the posterior (shared with the sigma-point path)
    R ctl[C::E_POST - D - sym(D)], mzq[NZ], Szq[sym(NZ)], cm, cv;
    if (!cell_posterior<M, R>(c, zt, mu, S, J, dm, dS, ctl, mzq, Szq, &cm, &cv)) set_status(a.status, b, 7, t);

    // linearised marginal observation (i2c.py:537-540): block-diagonal use of the posterior covariance
    R mz[NZ], Sz[sym(NZ)], CD[NZ * D];
    value_and_jacobian<M, FN_OBSERVE, D, NZ, R>(c.params, mu, mz, CD);
#pragma unroll
    for (int k = 0; k < NZ; ++k)
#pragma unroll
      for (int l = 0; l <= k; ++l) {
        R v = R(0);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int j = 0; j < D; ++j)
            if ((i < NX) == (j < NX)) v += CD[k * D + i] * S[tri_any(i, j)] * CD[l * D + j];
        Sz[tri(k, l)] = v;
      }
    R ca, cva;
    gaussian_cost<NZ>(c.QR, c.qr_diag != 0, mz, Sz, zt, &ca, &cva);
    store_cell<M, R>(c, a, t, b, mu, S, ctl, mz, Sz, cm, cv);
    sum_a += ca;
    sum_m += cm;
    sum_v += cv;
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = mu[i];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = S[i];
  }
}

template <class M, typename R>
I2C_HD inline void backward_lin_body(const Consts<M, R>& c, const CellArgs<R>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D, NZT = C::NZT, NT = C::NZT1;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_J = O_S3 + sym(NX);
  const long B = c.B;
  const int T = c.T;
  const R alpha = a.alpha[b];

  R m3m[NX], S3m[sym(NX)];
  lin_end_of_chain<M, R>(c, a, b, m3m, S3m);

  // The forward rows (and the per-cell target) of a cell are fetched one cell ahead where a second set of registers fits
  // (see chunk_walk_body / propagate_body: loaded at the top of their own cell, their round trip -- and, vmcnt being one in-order
  // counter, the acknowledgement of the previous cell's stores -- is exposed in every cell; a target behind a branch in mid-cell
  // costs a vmcnt(0) at its join). The first set is settled before the loop.
  constexpr bool PRE = C::D <= 5;
  R row[PRE ? C::E_FWD : 1], zt_cur[PRE ? NZ : 1];
  auto fetch_z = [&](const int r, R* zt) {  // branch-free: without per-cell targets the loads re-read fwd and are discarded
    const R* src = c.z_per_cell ? a.z + ((long)r * NZ) * B + b : a.fwd + b;
    const long st = c.z_per_cell ? B : 0;
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const R v = src[(long)k * st];
      zt[k] = c.z_per_cell ? v : c.zg[k];
    }
  };
  if (PRE) {
    const R* in0 = a.fwd + ((long)(T - 1) * C::E_FWD) * B + b;
#pragma unroll
    for (int e = 0; e < C::E_FWD; ++e) row[e] = opaque(in0[(long)e * B]);
    fetch_z(c.row(T - 1), zt_cur);
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt_cur[k] = opaque(zt_cur[k]);
  }
  R sum_a = R(0), sum_m = R(0), sum_v = R(0);
  for (int t = T - 1; t >= 0; --t) {
    R cur[PRE ? C::E_FWD : 1], nxt[PRE ? C::E_FWD : 1], zt_nxt[PRE ? NZ : 1];
    if (PRE) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) cur[e] = row[e];
      const R* inn = a.fwd + ((long)(t > 0 ? t - 1 : 0) * C::E_FWD) * B + b;
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) nxt[e] = inn[(long)e * B];
      fetch_z(c.row(t > 0 ? t - 1 : 0), zt_nxt);
    }
    const R* in = a.fwd + ((long)t * C::E_FWD) * B + b;
    auto ld = [&](const int e) { return PRE ? cur[PRE ? e : 0] : in[(long)e * B]; };
    R ztc[NZ];
#pragma unroll
    for (int k = 0; k < NZ; ++k) ztc[k] = PRE ? zt_cur[PRE ? k : 0] : (c.z_per_cell ? a.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k]);
    lin_backward_cell<M, R>(c, a, t, b, ld, ztc, m3m, S3m, sum_a, sum_m, sum_v);
    if (PRE) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) row[e] = nxt[e];
#pragma unroll
      for (int k = 0; k < NZ; ++k) zt_cur[k] = zt_nxt[k];
    }
  }
  a.term_stats[B + b] = sum_a;
  a.term_stats[2 * B + b] = sum_v;
  a.term_stats[(long)(C::E_TERM - 1) * B + b] = sum_m;
}

// ------------------------------------------------------------------------------------------
// CHUNKED form of the Linearize backward sweep (small batches; see chunk_compose_body in i2c_cell.hpp): the x-marginal recursion
// is the same affine map as in the sigma-point path, so the composites of the chunks come from the SAME k_chunk_compose; the
// stitch starts from the Linearize end of the chain, the walk does the Linearize cell and keeps three partial sums per chunk
// (alpha statistic, plan cost, cost variance), which one lane per trajectory adds up in chunk order (with the M-step riding on it
// inside i2c_learn). Sequential depth T / NC + NC instead of T.
// ------------------------------------------------------------------------------------------
template <class M, typename R>
I2C_HD inline void chunk_stitch_lin_body(const Consts<M, R>& c, const ChunkArgs<R, R>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, EC = NX + NX * NX + sym(NX);
  const long B = c.B;
  R m[NX], S[sym(NX)];
  lin_end_of_chain<M, R>(c, a.cell, b, m, S);
  for (int ch = a.n_chunks - 1; ch >= 0; --ch) {
    const R* cp = a.comp + ((long)ch * EC) * B + b;
    R av[NX], G[NX * NX], Cc[sym(NX)];
#pragma unroll
    for (int i = 0; i < NX; ++i) av[i] = cp[(long)i * B];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) G[i] = cp[(long)(NX + i) * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) Cc[i] = cp[(long)(NX + NX * NX + i) * B];
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
  }
}

template <class M, typename R>
I2C_HD inline void chunk_walk_lin_body(const Consts<M, R>& c, const ChunkArgs<R, R>& a, const int ch, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NZ = C::NZ;
  const long B = c.B;
  const int t_lo = ch * a.chunk_len, t_hi = (t_lo + a.chunk_len < c.T) ? t_lo + a.chunk_len : c.T;
  const CellArgs<R>& ca = a.cell;
  R m3m[NX], S3m[sym(NX)];
  {
    const R* bi = a.bnd + ((long)ch * C::E_XM) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) m3m[i] = bi[(long)i * B];
#pragma unroll
    for (int i = 0; i < sym(NX); ++i) S3m[i] = bi[(long)(NX + i) * B];
  }
  constexpr bool PRE = C::D <= 5;  // rows and target a cell ahead (see backward_lin_body)
  R row[PRE ? C::E_FWD : 1], zt_cur[PRE ? NZ : 1];
  auto fetch_z = [&](const int r, R* zt) {
    const R* src = c.z_per_cell ? ca.z + ((long)r * NZ) * B + b : ca.fwd + b;
    const long st = c.z_per_cell ? B : 0;
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      const R v = src[(long)k * st];
      zt[k] = c.z_per_cell ? v : c.zg[k];
    }
  };
  if (PRE) {
    const R* in0 = ca.fwd + ((long)(t_hi - 1) * C::E_FWD) * B + b;
#pragma unroll
    for (int e = 0; e < C::E_FWD; ++e) row[e] = opaque(in0[(long)e * B]);
    fetch_z(c.row(t_hi - 1), zt_cur);
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt_cur[k] = opaque(zt_cur[k]);
  }
  R sum_a = R(0), sum_m = R(0), sum_v = R(0);
  for (int t = t_hi - 1; t >= t_lo; --t) {
    R cur[PRE ? C::E_FWD : 1], nxt[PRE ? C::E_FWD : 1], zt_nxt[PRE ? NZ : 1];
    if (PRE) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) cur[e] = row[e];
      const R* inn = ca.fwd + ((long)(t > t_lo ? t - 1 : t_lo) * C::E_FWD) * B + b;
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) nxt[e] = inn[(long)e * B];
      fetch_z(c.row(t > t_lo ? t - 1 : t_lo), zt_nxt);
    }
    const R* in = ca.fwd + ((long)t * C::E_FWD) * B + b;
    auto ld = [&](const int e) { return PRE ? cur[PRE ? e : 0] : in[(long)e * B]; };
    R ztc[NZ];
#pragma unroll
    for (int k = 0; k < NZ; ++k) ztc[k] = PRE ? zt_cur[PRE ? k : 0] : (c.z_per_cell ? ca.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k]);
    lin_backward_cell<M, R>(c, ca, t, b, ld, ztc, m3m, S3m, sum_a, sum_m, sum_v);
    if (PRE) {
#pragma unroll
      for (int e = 0; e < C::E_FWD; ++e) row[e] = nxt[e];
#pragma unroll
      for (int k = 0; k < NZ; ++k) zt_cur[k] = zt_nxt[k];
    }
  }
  a.part[((long)ch * 3 + 0) * B + b] = sum_a;
  a.part[((long)ch * 3 + 1) * B + b] = sum_m;
  a.part[((long)ch * 3 + 2) * B + b] = sum_v;
}

// the three sums over the chunks, in chunk order (high t first, as the sequential walk adds them), and the M-step if asked for
template <class M, typename R>
I2C_HD inline void chunk_reduce_lin_body(const Consts<M, R>& c, const ChunkArgs<R, R>& a, const MstepArgs<R>& ms, const int b) {
  using C = Consts<M, R>;
  const long B = c.B;
  R sa = R(0), sm = R(0), sv = R(0);
  for (int ch = a.n_chunks - 1; ch >= 0; --ch) {
    sa += a.part[((long)ch * 3 + 0) * B + b];
    sm += a.part[((long)ch * 3 + 1) * B + b];
    sv += a.part[((long)ch * 3 + 2) * B + b];
  }
  a.cell.term_stats[B + b] = sa;
  a.cell.term_stats[2 * B + b] = sv;
  a.cell.term_stats[(long)(C::E_TERM - 1) * B + b] = sm;
  if (ms.alpha) mstep_body<M, R>(c, ms, b);
}

// ------------------------------------------------------------------------------------------
// Riccati-form backward messages (I2cCell._backward_ricatti_msgs, i2c.py:612-678; I2cGraph wrapper :888-893): the
// verification helper scripts/lqr_compare.py:175 runs after one Linearize forward/backward pass. Re-derives the cell
// quantities the reference stores during its forward pass (E, F, e, lambda_z1_f, nu_z1_f, lambda_z2_f, nu_z2_f, A, B, a,
// sig_u2_f, sig_x2_f, lambda_x2_f, lambda_x3_f, nu_x3_f; :278-346) from the prior / forward buffers, then walks
// T-1..0. Every matrix the reference inverts here is symmetric; they are inverted through their Cholesky factor and a
// non-positive pivot (improper backward message, e.g. no terminal cost) flags the trajectory with I2C_FAIL_RICCATI.
// Dense row-major N x N helpers; this path is a diagnostic, not a hot loop.
// ------------------------------------------------------------------------------------------
template <int N, typename R> I2C_FN bool spd_inverse(const R* A /* dense, symmetric */, R* Ainv /* dense */) {
  R L[sym(N)], rinv[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) L[tri(i, j)] = R(0.5) * (A[i * N + j] + A[j * N + i]);
  const bool ok = chol<N>(L, rinv);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    R col[N];
#pragma unroll
    for (int i = 0; i < N; ++i) col[i] = i == j ? R(1) : R(0);
    fsub<N>(L, rinv, col);
    bsub<N>(L, rinv, col);
#pragma unroll
    for (int i = 0; i < N; ++i) Ainv[i * N + j] = col[i];
  }
  return ok;
}
template <int NI, int NK, int NJ, typename R> I2C_FN void mm(const R* A, const R* Bm, R* Cm) {  // C = A B
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      R v = R(0);
#pragma unroll
      for (int k = 0; k < NK; ++k) v += A[i * NK + k] * Bm[k * NJ + j];
      Cm[i * NJ + j] = v;
    }
}
template <int NI, int NK, int NJ, typename R> I2C_FN void mm_tn(const R* A, const R* Bm, R* Cm) {  // C = A^T B, A is NK x NI
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      R v = R(0);
#pragma unroll
      for (int k = 0; k < NK; ++k) v += A[k * NI + i] * Bm[k * NJ + j];
      Cm[i * NJ + j] = v;
    }
}
template <int NI, int NK, int NJ, typename R> I2C_FN void mm_nt(const R* A, const R* Bm, R* Cm) {  // C = A B^T, B is NJ x NK
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      R v = R(0);
#pragma unroll
      for (int k = 0; k < NK; ++k) v += A[i * NK + k] * Bm[j * NK + k];
      Cm[i * NJ + j] = v;
    }
}
template <int N, typename R> I2C_FN void unpack_block(const R* S, const int off, R* out) {  // dense N x N block at (off, off)
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) out[i * N + j] = S[tri_any(off + i, off + j)];
}

template <typename R> struct RiccatiArgs {
  const R* prior;   // [T][D + sym(D)][B]  mu_xu0_f, sig_xu0_f of the forward pass (i2c_forward_sweep's prior_out)
  const R* fwd;     // [T][E_FWD][B]
  const R* xm;      // [T][E_XM][B]        only cell T-1 is read (mu_x3_m, sig_x3_m at the end of the chain)
  const R* z;       // [T][NZ][B] or null
  const R* alpha;   // [B]
  R* post;          // [T][E_POST][B]      in: sig_u0_m; out: K, k, sigK overwritten with the Riccati-form controller
  R* ric;           // [T][NX + NX*NX][B]  out: nu_x0_b, lambda_x0_b (row-major)
  int32_t* status;
};

template <class M, typename R>
I2C_HD inline void riccati_body(const Consts<M, R>& c, const RiccatiArgs<R>& a, const int b) {
  using C = Consts<M, R>;
  constexpr int NX = C::NX, NU = C::NU, NZ = C::NZ, D = C::D;
  constexpr int O_MU3 = D + sym(D), O_S3 = O_MU3 + NX, O_CTL = D + sym(D);
  const long B = c.B;
  const int T = c.T;
  const R alpha = a.alpha[b];
  bool ok = true;
  R nu3b[NX], lam3b[NX * NX];
  for (int t = T - 1; t >= 0; --t) {
    const R* pr = a.prior + ((long)t * (D + sym(D))) * B + b;
    const R* in = a.fwd + ((long)t * C::E_FWD) * B + b;
    R* po = a.post + ((long)c.row(t) * C::E_POST) * B + b;
    R mu0[D], S0[sym(D)], mu1[D], S1[sym(D)], m3f[NX], S3f[sym(NX)], zt[NZ];
#pragma unroll
    for (int e = 0; e < D; ++e) {
      mu0[e] = pr[(long)e * B];
      mu1[e] = in[(long)e * B];
    }
#pragma unroll
    for (int e = 0; e < sym(D); ++e) {
      S0[e] = pr[(long)(D + e) * B];
      S1[e] = in[(long)(D + e) * B];
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) m3f[e] = in[(long)(O_MU3 + e) * B];
#pragma unroll
    for (int e = 0; e < sym(NX); ++e) S3f[e] = in[(long)(O_S3 + e) * B];
#pragma unroll
    for (int k = 0; k < NZ; ++k) zt[k] = c.z_per_cell ? a.z[((long)c.row(t) * NZ + k) * B + b] : c.zg[k];

    // lambda_x3_f, nu_x3_f (i2c.py:346)
    R tmpx[NX * NX], lam3f[NX * NX], nu3f[NX];
    unpack_block<NX>(S3f, 0, tmpx);
    ok = spd_inverse<NX>(tmpx, lam3f) && ok;
    mm<NX, NX, 1>(lam3f, m3f, nu3f);
    if (t == T - 1) {  // end of chain (i2c.py:615-617)
      const R* xin = a.xm + ((long)t * C::E_XM) * B + b;
      R m3m[NX], S3m[sym(NX)], lam3m[NX * NX], v[NX];
#pragma unroll
      for (int e = 0; e < NX; ++e) m3m[e] = xin[(long)e * B];
#pragma unroll
      for (int e = 0; e < sym(NX); ++e) S3m[e] = xin[(long)(NX + e) * B];
      unpack_block<NX>(S3m, 0, tmpx);
      ok = spd_inverse<NX>(tmpx, lam3m) && ok;
      mm<NX, NX, 1>(lam3m, m3m, v);
#pragma unroll
      for (int i = 0; i < NX; ++i) nu3b[i] = v[i] - nu3f[i];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) lam3b[i] = lam3m[i] - lam3f[i];
    }

    // observation linearised about the prior mean (i2c.py:281-294, 312-317)
    R mz[NZ], EF[NZ * D], E[NZ * NX], F[NZ * NU], ev[NZ];
    value_and_jacobian<M, FN_OBSERVE, D, NZ, R>(c.params, mu0, mz, EF);
#pragma unroll
    for (int k = 0; k < NZ; ++k) {
      R v = mz[k];
#pragma unroll
      for (int i = 0; i < D; ++i) v -= EF[k * D + i] * mu0[i];
      ev[k] = v;
#pragma unroll
      for (int i = 0; i < NX; ++i) E[k * NX + i] = EF[k * D + i];
#pragma unroll
      for (int i = 0; i < NU; ++i) F[k * NU + i] = EF[k * D + NX + i];
    }
    R sx0[NX * NX], su0[NU * NU], sig_xi[NZ * NZ];
    unpack_block<NX>(S0, 0, sx0);
    unpack_block<NU>(S0, NX, su0);
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
      for (int j = 0; j < NZ; ++j) sig_xi[i * NZ + j] = alpha * c.sig_xi0[tri_any(i, j)];
    R Qm[NX * NX], nu_z1[NX], Rug[NU];
    {
      R t1[NZ * NU], sz[NZ * NZ], lz[NZ * NZ], r[NZ], lr[NZ], t2[NZ * NX];
      mm<NZ, NU, NU>(F, su0, t1);
      mm_nt<NZ, NU, NZ>(t1, F, sz);
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) sz[i] += sig_xi[i];
      ok = spd_inverse<NZ>(sz, lz) && ok;  // lambda_z1_f
#pragma unroll
      for (int k = 0; k < NZ; ++k) {
        R v = zt[k] - ev[k];
#pragma unroll
        for (int i = 0; i < NU; ++i) v -= F[k * NU + i] * mu0[NX + i];
        r[k] = v;
      }
      mm<NZ, NZ, 1>(lz, r, lr);
      mm_tn<NX, NZ, 1>(E, lr, nu_z1);
      mm<NZ, NZ, NX>(lz, E, t2);
      mm_tn<NX, NZ, NX>(E, t2, Qm);
    }
    {
      R t1[NZ * NX], sz[NZ * NZ], lz[NZ * NZ], r[NZ], lr[NZ];
      mm<NZ, NX, NX>(E, sx0, t1);
      mm_nt<NZ, NX, NZ>(t1, E, sz);
#pragma unroll
      for (int i = 0; i < NZ * NZ; ++i) sz[i] += sig_xi[i];
      ok = spd_inverse<NZ>(sz, lz) && ok;  // lambda_z2_f
#pragma unroll
      for (int k = 0; k < NZ; ++k) {
        R v = zt[k];
#pragma unroll
        for (int i = 0; i < NX; ++i) v -= E[k * NX + i] * mu0[i];
        r[k] = v - ev[k];
      }
      mm<NZ, NZ, 1>(lz, r, lr);
      mm_tn<NU, NZ, 1>(F, lr, Rug);  // nu_z2_f
    }
    // dynamics linearised about the updated mean (i2c.py:321-332)
    R f1[NX], AB[NX * D], Am[NX * NX], Bm[NX * NU], av[NX];
    value_and_jacobian<M, FN_DYNAMICS, D, NX, R>(c.params, mu1, f1, AB);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      R v = f1[i];
#pragma unroll
      for (int j = 0; j < D; ++j) v -= AB[i * D + j] * mu1[j];
      av[i] = v;
#pragma unroll
      for (int j = 0; j < NX; ++j) Am[i * NX + j] = AB[i * D + j];
#pragma unroll
      for (int j = 0; j < NU; ++j) Bm[i * NU + j] = AB[i * D + NX + j];
    }
    R sx1[NX * NX], su1[NU * NU], sig_u2[NX * NX], sig_x2[NX * NX], lam2f[NX * NX], eta[NX * NX];
    unpack_block<NX>(S1, 0, sx1);
    unpack_block<NU>(S1, NX, su1);
    unpack_block<NX>(c.sig_eta, 0, eta);
    {
      R t1[NX * NU], t2[NX * NX];
      mm<NX, NU, NU>(Bm, su1, t1);
      mm_nt<NX, NU, NX>(t1, Bm, sig_u2);
      mm<NX, NX, NX>(Am, sx1, t2);
      mm_nt<NX, NX, NX>(t2, Am, sig_x2);
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) sig_x2[i] += eta[i];
      ok = spd_inverse<NX>(sig_x2, lam2f) && ok;
    }
    // backwards Riccati equation (i2c.py:622-676)
    R nu_u_0[NU];
    {
      R iu[NU * NU];
      ok = spd_inverse<NU>(su0, iu) && ok;
      mm<NU, NU, 1>(iu, mu0 + NX, nu_u_0);
    }
    R gamma[NX * NX], Minv[NX * NX], LA[NX * NX], lam0b[NX * NX], nu0b[NX];
    {
      R s[NX * NX], si[NX * NX];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) s[i] = lam2f[i] + lam3b[i];
      ok = spd_inverse<NX>(s, si) && ok;
      mm<NX, NX, NX>(lam2f, si, gamma);
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) s[i] = eta[i] + sig_u2[i];
      ok = spd_inverse<NX>(s, si) && ok;
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) s[i] = si[i] + lam3b[i];  // M
      ok = spd_inverse<NX>(s, Minv) && ok;
    }
    mm<NX, NX, NX>(lam3b, Am, LA);
    {
      R ALA[NX * NX], MLA[NX * NX], LMLA[NX * NX], ALMLA[NX * NX];
      mm_tn<NX, NX, NX>(Am, LA, ALA);
      mm<NX, NX, NX>(Minv, LA, MLA);
      mm<NX, NX, NX>(lam3b, MLA, LMLA);
      mm_tn<NX, NX, NX>(Am, LMLA, ALMLA);
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) lam0b[i] = Qm[i] + ALA[i] - ALMLA[i];
      // AILM = A^T (I - (M^-T lam3b^T)^T) = A^T (I - lam3b M^-1)
      R LM[NX * NX], ILM[NX * NX], AILM[NX * NX], rhs[NX], lb[NX * NU], lbu[NX], la[NX], out[NX];
      mm<NX, NX, NX>(lam3b, Minv, LM);
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) ILM[i * NX + j] = (i == j ? R(1) : R(0)) - LM[i * NX + j];
      mm_tn<NX, NX, NX>(Am, ILM, AILM);
      mm<NX, NX, 1>(lam3b, av, la);
      mm<NX, NX, NU>(lam3b, Bm, lb);
      mm<NX, NU, 1>(lb, mu1 + NX, lbu);
#pragma unroll
      for (int i = 0; i < NX; ++i) rhs[i] = nu3b[i] - la[i] - lbu[i];
      mm<NX, NX, 1>(AILM, rhs, out);
#pragma unroll
      for (int i = 0; i < NX; ++i) nu0b[i] = nu_z1[i] + out[i];
    }
    R Kr[NU * NX], kr[NU], sig_u[NU * NU];
#pragma unroll
    for (int p = 0; p < NU; ++p)
#pragma unroll
      for (int q = 0; q < NU; ++q) sig_u[p * NU + q] = po[(long)(D + tri_any(NX + p, NX + q)) * B];  // sig_u0_m
    {
      R gamma_L[NX * NX], sig3b[NX * NX], s[NX * NX], lam2b[NX * NX], mu_u2[NX], nu2b[NX], psi[NX * NX];
      mm<NX, NX, NX>(gamma, lam3b, gamma_L);
      ok = spd_inverse<NX>(lam3b, sig3b) && ok;
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) s[i] = sig3b[i] + sig_u2[i];
      ok = spd_inverse<NX>(s, lam2b) && ok;
      mm<NX, NU, 1>(Bm, mu1 + NX, mu_u2);
      {
        R ls[NX * NX], v[NX];
        mm<NX, NX, NX>(lam2b, sig3b, ls);
        mm<NX, NX, 1>(ls, nu3b, v);
#pragma unroll
        for (int i = 0; i < NX; ++i) nu2b[i] = v[i] - mu_u2[i];
      }
      {
        R sl[NX * NX], t1[NX * NX];
#pragma unroll
        for (int i = 0; i < NX * NX; ++i) sl[i] = lam2f[i] + lam2b[i];
        mm<NX, NX, NX>(sig_x2, sl, t1);
        mm<NX, NX, NX>(gamma_L, t1, psi);
      }
      // K = -sig_u B^T psi A ;  k = sig_u (nu_u_0 + Rug + B^T (gamma nu3b + (I - gamma) nu2b - psi a))
      R pA[NX * NX], BpA[NU * NX], g1[NX], g2[NX], pa[NX], w[NX], Bw[NU], rr[NU];
      mm<NX, NX, NX>(psi, Am, pA);
      mm_tn<NU, NX, NX>(Bm, pA, BpA);
      mm<NU, NU, NX>(sig_u, BpA, Kr);
#pragma unroll
      for (int i = 0; i < NU * NX; ++i) Kr[i] = -Kr[i];
      mm<NX, NX, 1>(gamma, nu3b, g1);
      mm<NX, NX, 1>(gamma, nu2b, g2);
      mm<NX, NX, 1>(psi, av, pa);
#pragma unroll
      for (int i = 0; i < NX; ++i) w[i] = g1[i] + (nu2b[i] - g2[i]) - pa[i];
      mm_tn<NU, NX, 1>(Bm, w, Bw);
#pragma unroll
      for (int i = 0; i < NU; ++i) rr[i] = nu_u_0[i] + Rug[i] + Bw[i];
      mm<NU, NU, 1>(sig_u, rr, kr);
    }
#pragma unroll
    for (int i = 0; i < NU * NX; ++i) po[(long)(O_CTL + i) * B] = Kr[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) po[(long)(O_CTL + NU * NX + i) * B] = kr[i];
#pragma unroll
    for (int p = 0; p < NU; ++p)
#pragma unroll
      for (int q = 0; q <= p; ++q) po[(long)(O_CTL + NU * NX + NU + tri(p, q)) * B] = sig_u[p * NU + q];
    R* ro = a.ric + ((long)t * (NX + NX * NX)) * B + b;
#pragma unroll
    for (int i = 0; i < NX; ++i) ro[(long)i * B] = nu0b[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) ro[(long)(NX + i) * B] = lam0b[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) nu3b[i] = nu0b[i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) lam3b[i] = lam0b[i];
  }
  if (!ok) set_status(a.status, b, 10, 0);
}

}  // namespace i2c
