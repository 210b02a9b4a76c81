// Device functors for the reference's known-model plugins (the `sys` protocol of
// SURVEY.md section 8b): dynamics / observe / observe_terminal evaluated at one sigma point.
// Citations are relative to the reference root.
//
// Every model lists its NA "angle coordinates" (state entries that only enter through sin/cos).
// The functors receive sn[a] = sin(x[ang(a)]), cs[a] = cos(x[ang(a)]) precomputed by the
// sigma-point transform, which evaluates sincos once at the mean and once per (angle, column)
// offset and rotates (angle addition) instead of calling sincos at all 2d+1 points.
#pragma once
#include "i2c_linalg.hpp"

#ifndef I2C_PENDULUM_GROUP
#define I2C_PENDULUM_GROUP 4
#endif

namespace i2c {

// Kernel-family knobs every model functor carries, with the values of a model that only asks for the one-lane-per-trajectory
// kernels. An OUT-OF-TREE model (`python build.py --model my_model.hpp`) derives from this and states
// its dimensions, its structure hints and its three functions; it may opt into the multi-lane families by overriding a knob:
//   GROUP = 4 / 8 / 16   group kernels (G lanes per trajectory; G >= the largest of d, nz, and a power of two)
//   QUAD = true          quad kernels: d <= 8 with an identity observation, or d % 4 != 0 with one action (i2c_quad.hpp): the forward sweep
//                        inside [QUAD_FORWARD_MIN_B, QUAD_FORWARD_MAX_B] by default, and both sweeps on request (group_lanes = 64)
//   (optional) QUAD_BACKWARD8_MIN_B / _MAX_B: a default batch window for the d <= 8 quad backward walk (fused: none is measured faster
//                        than the chunked lane schedule on MI355X, so no in-tree model sets it)
//   (optional) QUAD_CHUNK_WALK_MIN_B / _MAX_B: a default batch window for the quad WALKER inside the chunked schedule (a few hundred
//                        trajectories at most: there the walk is as long as ONE lane cell x the chunk length, and a quad cell is 5x shorter)
//   (optional) QUAD_CHUNK_PASSES_MIN_B / _MAX_B: a default batch window for the COMPOSE and STITCH passes of that schedule in the
//                        quad form (up to ~1000 trajectories: pure small-matrix products, 20 - 25 matrix instructions per step)
//   (optional) QUAD_CHUNK_STITCH_MAX_B: the stitch pass ALONE stays in the quad form up to here (it is a chain of NC dependent steps
//                        on few wavefronts at any batch size; the compose pass becomes throughput-bound much earlier)
struct ModelDefaults {
  static constexpr int ID = -1;   // in-tree models: their I2C_MODEL_* value; plugins get an id from i2c_register_model
  static constexpr int NP = 0, NA = 0;
  static constexpr int GROUP = 0;
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = false;
  static constexpr bool WAVE = false;
  static constexpr bool QUAD = false;
  static constexpr int QUAD_FORWARD_MAX_B = 0;
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;
  // (optional) static constexpr int BWD_FUSED_MIN_B: batch size from which I2C_BWD_AUTO runs the fused backward walk instead of the
  // chunked schedule; models without it take I2C_BWD_FUSED_MIN_B (include/i2c_hip.h)
  I2C_HD static constexpr int ang(int) { return 0; }
  // structure hints of the observation functions (ObsStruct, i2c_cell.hpp): output k is a pass-through of input obs_lin(k), or
  // (-1) a general function that depends on no input with an index above obs_dep(k). The defaults say "nothing is known"
  // (every output depends on every input): always correct; sharper hints only save work.
  I2C_HD static constexpr int obs_lin(int) { return -1; }
  I2C_HD static constexpr int obs_dep(int) { return 1 << 20; }
  I2C_HD static constexpr int term_lin(int) { return -1; }
  I2C_HD static constexpr int term_dep(int) { return 1 << 20; }
  I2C_HD static constexpr int meas_lin(int) { return -1; }
  I2C_HD static constexpr int meas_dep(int) { return 1 << 20; }
};

// PendulumKnown: i2c/env_def.py:233-309, step i2c/env_autograd.py:5-19
struct Pendulum {
  static constexpr int ID = 0, NX = 2, NU = 1, NZ = 4, NZT = 3, NP = 0, NA = 1;
  // lanes per trajectory of the group kernels (i2c_group.hpp); 0: none compiled. I2C_PENDULUM_GROUP is an experiment knob
  // (-DI2C_PENDULUM_GROUP=8 through I2C_HIPCC_EXTRA): the shipped build is 4, the widest group with a row for every lane but one.
  static constexpr int GROUP = I2C_PENDULUM_GROUP;
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = false;  // see Impl::forward_any (i2c_impl.hpp)
  static constexpr bool WAVE = false;  // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;  // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp): d <= 8 models
  static constexpr int QUAD_FORWARD_MAX_B = 0;  // the quad kernel is the DEFAULT forward sweep up to this batch size (measured crossover, profiles/README.md); 0: only on request
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;  // (d = 16 only; the d <= 8 quad backward walk has no default window: on request)
  I2C_HD static constexpr int ang(int) { return 0; }
  // z = [sin th, cos th, thd, u],  zT = [sin th, cos th, thd]
  I2C_HD static constexpr int obs_lin(int k) { return k < 2 ? -1 : k - 1; }
  I2C_HD static constexpr int obs_dep(int) { return 0; }
  I2C_HD static constexpr int term_lin(int k) { return k < 2 ? -1 : 1; }
  I2C_HD static constexpr int term_dep(int) { return 0; }
  // measurement of the MPC state estimator (build-defined: the reference has no `measure` here)
  static constexpr int NY = 3;
  I2C_HD static constexpr int meas_lin(int k) { return k < 2 ? -1 : 1; }
  I2C_HD static constexpr int meas_dep(int) { return 0; }
  template <typename R> I2C_FN void measure(const R* p, const R* x, const R* sn, const R* cs, R* y) {
    observe_terminal(p, x, sn, cs, y);
  }
  template <typename R> I2C_FN void dynamics(const R*, const R* xu, const R* sn, const R*, R* xn) {
    const R dt = R(0.05), damp = R(1e-2), u_max = R(2.0);
    const R c_grav = R(-3.0 * 9.80665 / (2 * 1.0));  // -3 g / (2 l)
    const R c_torque = R(3.0 / (1.0 * 1.0 * 1.0));    // 3 / (m l^2)
    const R u = r_clip(xu[2], -u_max, u_max);
    R acc = c_grav * (-sn[0]) - damp * xu[1];  // sin(th + pi) = -sin(th)
    acc += c_torque * u;
    const R om = xu[1] + acc * dt;
    xn[0] = xu[0] + om * dt;
    xn[1] = om;
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R* sn, const R* cs, R* z) {  // env_def.py:273-276
    z[0] = sn[0];
    z[1] = cs[0];
    z[2] = xu[1];
    z[3] = xu[2];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R* sn, const R* cs, R* z) {  // env_def.py:288-291
    z[0] = sn[0];
    z[1] = cs[0];
    z[2] = x[1];
  }
};

// PendulumKnownActReg: i2c/env_def.py:312-346 (only the action is observed; no terminal observation)
struct PendulumActReg {
  static constexpr int ID = 1, NX = 2, NU = 1, NZ = 1, NZT = 0, NP = 0, NA = 1;
  static constexpr int GROUP = 4;  // lanes per trajectory of the group kernels (i2c_group.hpp); 0: none compiled
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = false;  // see Impl::forward_any (i2c_impl.hpp)
  static constexpr bool WAVE = false;  // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;  // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp): d <= 8 models
  static constexpr int QUAD_FORWARD_MAX_B = 0;  // the quad kernel is the DEFAULT forward sweep up to this batch size (measured crossover, profiles/README.md); 0: only on request
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;  // (d = 16 only; the d <= 8 quad backward walk has no default window: on request)
  I2C_HD static constexpr int ang(int) { return 0; }
  I2C_HD static constexpr int obs_lin(int) { return 2; }  // z = [u]
  I2C_HD static constexpr int obs_dep(int) { return 0; }
  I2C_HD static constexpr int term_lin(int) { return 0; }
  I2C_HD static constexpr int term_dep(int) { return 0; }
  static constexpr int NY = 3;
  I2C_HD static constexpr int meas_lin(int k) { return k < 2 ? -1 : 1; }
  I2C_HD static constexpr int meas_dep(int) { return 0; }
  template <typename R> I2C_FN void measure(const R* p, const R* x, const R* sn, const R* cs, R* y) {
    Pendulum::observe_terminal(p, x, sn, cs, y);
  }
  template <typename R> I2C_FN void dynamics(const R* p, const R* xu, const R* sn, const R* cs, R* xn) {
    Pendulum::dynamics(p, xu, sn, cs, xn);
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R*, const R*, R* z) { z[0] = xu[2]; }
  template <typename R> I2C_FN void observe_terminal(const R*, const R*, const R*, const R*, R*) {}
};

// CartpoleKnown: i2c/env_def.py:491-612, step i2c/env_autograd.py:25-54
struct Cartpole {
  static constexpr int ID = 2, NX = 4, NU = 1, NZ = 6, NZT = 5, NP = 0, NA = 1;
  static constexpr int GROUP = 8;  // lanes per trajectory of the group kernels (i2c_group.hpp); 0: none compiled
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = false;  // see Impl::forward_any (i2c_impl.hpp)
  static constexpr bool WAVE = false;  // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;  // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp): d <= 8 models
  static constexpr int QUAD_FORWARD_MAX_B = 4096;  // the quad kernel is the DEFAULT forward sweep up to this batch size (measured crossover, profiles/README.md); 0: only on request
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;  // (d = 16 only; the d <= 8 quad backward walk has no default window: on request)
  static constexpr int QUAD_CHUNK_WALK_MIN_B = 1, QUAD_CHUNK_WALK_MAX_B = 64;  // the chunked schedule's walk pass on the quad walker by default (measured, profiles/r6_quad_chunk_walk.txt: B = 1 / 64: backward sweep 0.090 / 0.078 -> 0.069 / 0.074 ms; 256: 0.082 -> 0.091)
  static constexpr int QUAD_CHUNK_PASSES_MIN_B = 1, QUAD_CHUNK_PASSES_MAX_B = 128;  // compose pass (+ stitch) in the quad form by default (measured with the quad stitch on, profiles/r6_quad_chunk_passes.txt: backward sweep B = 64 / 128: 0.066 / 0.072 -> 0.061 / 0.068 ms; 256: 0.074 -> 0.078)
  static constexpr int QUAD_CHUNK_STITCH_MAX_B = 2048;  // ... and the stitch pass alone up to here (backward sweep B = 1024 / 2048: 0.114 / 0.221 -> 0.107 / 0.213 ms; 4096: level)
  I2C_HD static constexpr int ang(int) { return 1; }
  // z = [x, sin th, cos th, xd, thd, u],  zT = [x, sin th, cos th, xd, thd]
  I2C_HD static constexpr int obs_lin(int k) { return k == 0 ? 0 : (k < 3 ? -1 : k - 1); }
  I2C_HD static constexpr int obs_dep(int) { return 1; }
  I2C_HD static constexpr int term_lin(int k) { return k == 0 ? 0 : (k < 3 ? -1 : k - 1); }
  I2C_HD static constexpr int term_dep(int) { return 1; }
  static constexpr int NY = 5;
  I2C_HD static constexpr int meas_lin(int k) { return term_lin(k); }
  I2C_HD static constexpr int meas_dep(int) { return 1; }
  template <typename R> I2C_FN void measure(const R* p, const R* x, const R* sn, const R* cs, R* y) {
    observe_terminal(p, x, sn, cs, y);
  }
  template <typename R> I2C_FN void dynamics(const R*, const R* xu, const R* sn, const R* cs, R* xn) {
    const R grav = R(9.81), m_cart = R(0.37), m_pole = R(0.127), len = R(0.3365);
    const R dt = R(1.0 / 250.0), u_max = R(5.0);
    const R m_tot = m_cart + m_pole;
    const R u = r_clip(xu[4], -u_max, u_max);
    const R om2 = xu[3] * xu[3];
    const R s = sn[0], c = cs[0];
    const R num = -m_pole * len * s * c * om2 + m_tot * grav * s - u * c;
    const R den = len * (R(4.0 / 3.0) * m_tot - m_pole * (c * c));
    const R th_acc = num * r_rcp(den);
    const R x_acc = (m_pole * len * s * om2 - m_pole * len * th_acc * c + u) * (R(1) / m_tot);
    xn[0] = xu[0] + dt * xu[2];
    xn[1] = xu[1] + dt * xu[3];
    xn[2] = xu[2] + dt * x_acc;
    xn[3] = xu[3] + dt * th_acc;
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R* sn, const R* cs, R* z) {  // env_def.py:537-549
    z[0] = xu[0];
    z[1] = sn[0];
    z[2] = cs[0];
    z[3] = xu[2];
    z[4] = xu[3];
    z[5] = xu[4];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R* sn, const R* cs, R* z) {  // env_def.py:567-570
    z[0] = x[0];
    z[1] = sn[0];
    z[2] = cs[0];
    z[3] = x[2];
    z[4] = x[3];
  }
};

// DoubleCartpoleKnown: i2c/env_def.py:615-761, step i2c/env_autograd.py:60-167
struct DoubleCartpole {
  static constexpr int ID = 3, NX = 6, NU = 1, NZ = 9, NZT = 8, NP = 0, NA = 2;
  static constexpr int GROUP = 16;  // lanes per trajectory of the group kernels (i2c_group.hpp); 0: none compiled
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = true;   // see Impl::forward_any (i2c_impl.hpp)
  static constexpr bool WAVE = false;  // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;  // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp): d <= 8 models
  static constexpr int QUAD_FORWARD_MAX_B = 8192;  // the quad kernel is the DEFAULT forward sweep up to this batch size (measured crossover, profiles/README.md); 0: only on request
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;  // (d = 16 only; the d <= 8 quad backward walk has no default window: on request)
  static constexpr int QUAD_CHUNK_WALK_MIN_B = 1, QUAD_CHUNK_WALK_MAX_B = 256;  // the chunked schedule's walk pass on the quad walker by default (measured, profiles/r6_quad_chunk_walk.txt: B = 1 / 64 / 256: backward sweep 0.159 / 0.140 / 0.151 -> 0.119 / 0.123 / 0.145 ms; 512: 0.162 -> 0.188)
  static constexpr int QUAD_CHUNK_PASSES_MIN_B = 1, QUAD_CHUNK_PASSES_MAX_B = 256;  // compose pass (+ stitch) in the quad form by default (measured with the quad stitch on, profiles/r6_quad_chunk_passes.txt: backward sweep B = 64 / 128: 0.082 / 0.087 -> 0.071 / 0.075 ms; 256: level; 512: 0.123 -> 0.141)
  static constexpr int QUAD_CHUNK_STITCH_MAX_B = 8192;  // ... and the stitch pass alone up to here (backward sweep B = 1024 / 4096 / 8192: 0.205 / 0.474 / 0.925 -> 0.164 / 0.463 / 0.854 ms)
  static constexpr int BWD_FUSED_MIN_B = 20480;  // I2C_BWD_AUTO runs the fused backward walk from here up: its rows are the widest (104 doubles), the chunked form keeps its lead longer (measured: 16384: chunked 1.67 / fused 1.86 ms; 24576: 3.44 / 2.20)
  I2C_HD static constexpr int ang(int a) { return a == 0 ? 1 : 2; }
  // z = [x, sin th1, cos th1, sin th2, cos th2, xd, th1d, th2d, u],  zT = z without u
  I2C_HD static constexpr int obs_lin(int k) { return k == 0 ? 0 : (k < 5 ? -1 : k - 2); }
  I2C_HD static constexpr int obs_dep(int k) { return k < 3 ? 1 : 2; }
  I2C_HD static constexpr int term_lin(int k) { return k == 0 ? 0 : (k < 5 ? -1 : k - 2); }
  I2C_HD static constexpr int term_dep(int k) { return k < 3 ? 1 : 2; }
  static constexpr int NY = 8;
  I2C_HD static constexpr int meas_lin(int k) { return term_lin(k); }
  I2C_HD static constexpr int meas_dep(int k) { return term_dep(k); }
  template <typename R> I2C_FN void measure(const R* p, const R* x, const R* sn, const R* cs, R* y) {
    observe_terminal(p, x, sn, cs, y);
  }
  template <typename R> I2C_FN void dynamics(const R*, const R* xu, const R* sn, const R* cs, R* xn) {
    const R dt = R(1.0 / 125.0), grav = R(9.81);
    const R m_c = R(0.37), m1 = R(0.127), m2 = R(0.127);
    const R L1 = R(0.3365), L2 = R(0.3365);
    const R l1 = L1 / 2, l2 = L2 / 2;
    const R J1 = m1 * L1 / 12, J2 = m2 * L2 / 12;
    const R u_max = R(10.0), gear = R(3.0);
    const R m_tot = m_c + m1 + m2;
    const R h1 = m1 * l1 + m2 * L2, h2 = m2 * l2, h3 = L1 * l2 * m2;

    const R s1 = sn[0], c1 = cs[0], s2 = sn[1], c2 = cs[1];
    const R sd = s1 * c2 - c1 * s2, cd = c1 * c2 + s1 * s2;  // sin / cos (th1 - th2)
    const R qd0 = xu[3], qd1 = xu[4], qd2 = xu[5];
    // symmetric mass matrix M(q)
    const R M00 = m_tot, M01 = h1 * c1, M02 = h2 * c2;
    const R M11 = l1 * l1 * m1 + L1 * L1 * m2 + J1, M12 = h3 * cd;
    const R M22 = l2 * l2 * m2 + J2;
    // rhs = B u - C(q, qd) qd - G(q)
    const R u = gear * r_clip(xu[6], -u_max, u_max);
    const R r0 = u - ((-h1 * qd1 * s1) * qd1 + (-h2 * qd2 * s2) * qd2);
    const R r1 = -((h3 * qd2 * sd) * qd2) - (-(m1 * l1 + m2 * L1) * grav * s1);
    const R r2 = -((-h3 * qd1 * sd) * qd1) - (-m2 * l2 * grav * s2);
    // qdd = M^{-1} rhs through the adjugate (M is symmetric 3x3)
    const R A00 = M11 * M22 - M12 * M12, A01 = M02 * M12 - M01 * M22, A02 = M01 * M12 - M02 * M11;
    const R A11 = M00 * M22 - M02 * M02, A12 = M01 * M02 - M00 * M12, A22 = M00 * M11 - M01 * M01;
    const R idet = r_rcp(M00 * A00 + M01 * A01 + M02 * A02);
    const R a0 = (A00 * r0 + A01 * r1 + A02 * r2) * idet;
    const R a1 = (A01 * r0 + A11 * r1 + A12 * r2) * idet;
    const R a2 = (A02 * r0 + A12 * r1 + A22 * r2) * idet;
    xn[3] = qd0 + a0 * dt;
    xn[4] = qd1 + a1 * dt;
    xn[5] = qd2 + a2 * dt;
    xn[0] = xu[0] + xn[3] * dt;
    xn[1] = xu[1] + xn[4] * dt;
    xn[2] = xu[2] + xn[5] * dt;
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R* sn, const R* cs, R* z) {  // env_def.py:682-695
    z[0] = xu[0];
    z[1] = sn[0];
    z[2] = cs[0];
    z[3] = sn[1];
    z[4] = cs[1];
    z[5] = xu[3];
    z[6] = xu[4];
    z[7] = xu[5];
    z[8] = xu[6];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R* sn, const R* cs, R* z) {  // env_def.py:719-732
    z[0] = x[0];
    z[1] = sn[0];
    z[2] = cs[0];
    z[3] = sn[1];
    z[4] = cs[1];
    z[5] = x[3];
    z[6] = x[4];
    z[7] = x[5];
  }
};

// LinearKnown: i2c/env_def.py:139-191, i2c/model.py:226-246.  params = A (2x2 row-major), B (2), a (2)
struct Linear {
  static constexpr int ID = 4, NX = 2, NU = 1, NZ = 3, NZT = 2, NP = 8, NA = 0;
  static constexpr int GROUP = 4;  // lanes per trajectory of the group kernels (i2c_group.hpp); 0: none compiled
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = false;  // see Impl::forward_any (i2c_impl.hpp)
  static constexpr bool WAVE = false;  // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;  // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp): d <= 8 models
  static constexpr int QUAD_FORWARD_MAX_B = 0;  // the quad kernel is the DEFAULT forward sweep up to this batch size (measured crossover, profiles/README.md); 0: only on request
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;  // (d = 16 only; the d <= 8 quad backward walk has no default window: on request)
  I2C_HD static constexpr int ang(int) { return 0; }
  I2C_HD static constexpr int obs_lin(int k) { return k; }  // z = xu, zT = x
  I2C_HD static constexpr int obs_dep(int) { return 0; }
  I2C_HD static constexpr int term_lin(int k) { return k; }
  I2C_HD static constexpr int term_dep(int) { return 0; }
  static constexpr int NY = 2;  // y = x
  I2C_HD static constexpr int meas_lin(int k) { return k; }
  I2C_HD static constexpr int meas_dep(int) { return 0; }
  template <typename R> I2C_FN void measure(const R*, const R* x, const R*, const R*, R* y) {
    y[0] = x[0];
    y[1] = x[1];
  }
  template <typename R> I2C_FN void dynamics(const R* p, const R* xu, const R*, const R*, R* xn) {
    xn[0] = xu[0] * p[0] + xu[1] * p[1] + xu[2] * p[4] + p[6];
    xn[1] = xu[0] * p[2] + xu[1] * p[3] + xu[2] * p[5] + p[7];
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R*, const R*, R* z) {
    z[0] = xu[0];
    z[1] = xu[1];
    z[2] = xu[2];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R*, const R*, R* z) {
    z[0] = x[0];
    z[1] = x[1];
  }
};

// LinearKnownMinimumEnergy: i2c/env_def.py:194-230 (only the action is observed; terminal = state)
struct LinearMinEnergy {
  static constexpr int ID = 5, NX = 2, NU = 1, NZ = 1, NZT = 2, NP = 8, NA = 0;
  static constexpr int GROUP = 0;  // lanes per trajectory of the group kernels (i2c_group.hpp); 0: none compiled
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = false;  // see Impl::forward_any (i2c_impl.hpp)
  static constexpr bool WAVE = false;  // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;  // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp): d <= 8 models
  static constexpr int QUAD_FORWARD_MAX_B = 0;  // the quad kernel is the DEFAULT forward sweep up to this batch size (measured crossover, profiles/README.md); 0: only on request
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;  // (d = 16 only; the d <= 8 quad backward walk has no default window: on request)
  I2C_HD static constexpr int ang(int) { return 0; }
  I2C_HD static constexpr int obs_lin(int) { return 2; }  // z = [u], zT = x
  I2C_HD static constexpr int obs_dep(int) { return 0; }
  I2C_HD static constexpr int term_lin(int k) { return k; }
  I2C_HD static constexpr int term_dep(int) { return 0; }
  static constexpr int NY = 2;  // y = x
  I2C_HD static constexpr int meas_lin(int k) { return k; }
  I2C_HD static constexpr int meas_dep(int) { return 0; }
  template <typename R> I2C_FN void measure(const R*, const R* x, const R*, const R*, R* y) {
    y[0] = x[0];
    y[1] = x[1];
  }
  template <typename R> I2C_FN void dynamics(const R* p, const R* xu, const R* sn, const R* cs, R* xn) {
    Linear::dynamics(p, xu, sn, cs, xn);
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R*, const R*, R* z) { z[0] = xu[2]; }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R*, const R*, R* z) {
    z[0] = x[0];
    z[1] = x[1];
  }
};

// Build-defined analytic planar quadrotor with the reference's interface, dimensions and constants
// (scripts/mpc_state_est/mpc_quad.py:219-383: dim_x 6, dim_u 2, dim_z 8, observe = identity,
// dt = 1/FS = 0.1, arm = vehicle_dx = W/25 = 0.8, angularDamping 0.5, g = 9.81, thrust along the
// body normal at +/- arm, forces clipped to [0, force_mx]). The reference steps a Box2D body, which
// cannot be reproduced (not vendored / pinned); this is Box2D's integrator for a free body:
// velocities first (semi-implicit Euler, damping as 1 / (1 + dt c)), then positions.
// params = {mass, inertia, u_max}.
struct Quadrotor {
  static constexpr int ID = 6, NX = 6, NU = 2, NZ = 8, NZT = 6, NP = 3, NA = 1;
  static constexpr int GROUP = 8;  // lanes per trajectory of the group kernels (i2c_group.hpp); 0: none compiled
  static constexpr bool GROUP_ONLY = false;
  static constexpr bool GROUP_FORWARD_AUTO = true;   // see Impl::forward_any (i2c_impl.hpp)
  static constexpr bool WAVE = false;  // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;  // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp): d <= 8 models
  static constexpr int QUAD_FORWARD_MAX_B = 8192;  // the quad kernel is the DEFAULT forward sweep up to this batch size (measured crossover, profiles/README.md); 0: only on request
  static constexpr int QUAD_FORWARD_MIN_B = 0;
  static constexpr int QUAD_BACKWARD_MIN_B = 0;  // (d = 16 only; the d <= 8 quad backward walk has no default window: on request)
  static constexpr int QUAD_CHUNK_WALK_MIN_B = 1, QUAD_CHUNK_WALK_MAX_B = 256;  // the chunked schedule's walk pass on the quad walker by default (measured, profiles/r6_quad_chunk_walk.txt: B = 1 / 64 / 256: backward sweep 0.065 / 0.059 / 0.059 -> 0.047 / 0.051 / 0.051 ms; 512: 0.062 -> 0.065)
  static constexpr int QUAD_CHUNK_PASSES_MIN_B = 1, QUAD_CHUNK_PASSES_MAX_B = 384;  // compose pass (+ stitch) in the quad form by default (measured with the quad stitch on, profiles/r6_quad_chunk_passes.txt: backward sweep B = 64 / 256 / 384: 0.039 / 0.041 / 0.052 -> 0.035 / 0.035 / 0.047 ms; 512: 0.052 -> 0.054; 1024: 0.056 -> 0.063)
  static constexpr int QUAD_CHUNK_STITCH_MAX_B = 4096;  // ... and the stitch pass alone up to here (backward sweep B = 2048 / 4096: 0.078 / 0.110 -> 0.068 / 0.100 ms; 8192: level)
  I2C_HD static constexpr int ang(int) { return 2; }
  I2C_HD static constexpr int obs_lin(int k) { return k; }  // z = xu, zT = x
  I2C_HD static constexpr int obs_dep(int) { return 0; }
  I2C_HD static constexpr int term_lin(int k) { return k; }
  I2C_HD static constexpr int term_dep(int) { return 0; }
  // measure(): positions and velocities of the two rotor tips, mpc_quad.py:370-383 -- INCLUDING the
  // reference's operator slip in rxd / ryd (`+ vehicle_dx - sin(th) thd`, `+ vehicle_dx + cos(th) thd`).
  static constexpr int NY = 8;
  I2C_HD static constexpr int meas_lin(int) { return -1; }
  I2C_HD static constexpr int meas_dep(int) { return 5; }
  template <typename R> I2C_FN void measure(const R*, const R* x, const R* sn, const R* cs, R* y) {
    const R dx = R(0.8), s = sn[0], c = cs[0];
    y[0] = x[0] - dx * c;
    y[1] = x[1] - dx * s;
    y[2] = x[0] + dx * c;
    y[3] = x[1] + dx * s;
    y[4] = x[3] - dx * (-s) * x[5];
    y[5] = x[4] - dx * c * x[5];
    y[6] = x[3] + dx - s * x[5];
    y[7] = x[4] + dx + c * x[5];
  }
  template <typename R> I2C_FN void dynamics(const R* p, const R* xu, const R* sn, const R* cs, R* xn) {
    const R dt = R(0.1), arm = R(0.8), ang_damp = R(0.5), grav = R(9.81);
    const R mass = p[0], inertia = p[1], u_max = p[2];
    const R f1 = r_clip(xu[6], R(0), u_max), f2 = r_clip(xu[7], R(0), u_max);
    const R thrust = f1 + f2;
    const R s = sn[0], c = cs[0];
    const R imass = R(1) / mass;  // wave-uniform: one scalar-ish divide hoisted by the compiler
    const R ax = -thrust * s * imass;
    const R ay = thrust * c * imass - grav;
    const R al = arm * (f2 - f1) / inertia;
    xn[3] = xu[3] + dt * ax;
    xn[4] = xu[4] + dt * ay;
    xn[5] = (xu[5] + dt * al) / (R(1) + dt * ang_damp);
    xn[0] = xu[0] + dt * xn[3];
    xn[1] = xu[1] + dt * xn[4];
    xn[2] = xu[2] + dt * xn[5];
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R*, const R*, R* z) {
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = xu[i];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R*, const R*, R* z) {
#pragma unroll
    for (int i = 0; i < 6; ++i) z[i] = x[i];
  }
};

// Build-defined 12-state quadrotor (BASELINE config 4 names nx = 12; the reference only has the planar Box2D body of
// scripts/mpc_state_est/mpc_quad.py:219-383, so, like Quadrotor above, this is an analytic model with the reference's
// plugin interface). State x = [p (3) | roll, pitch, yaw | v (3, world) | body rates (3)], action = four rotor thrusts
// clipped to [0, u_max], "+" configuration: roll torque arm (f2 - f4), pitch torque arm (f3 - f1), yaw torque
// kq (f1 - f2 + f3 - f4). Integrator as the planar model (Box2D's order for a free body): velocities first
// (semi-implicit Euler, angular damping as 1 / (1 + dt c)), then positions and Euler angles with the NEW velocities.
// observe = identity on (x, u), observe_terminal = identity on x, measure = [p | angles | body rates].
// params = {mass, Ixx, Iyy, Izz, u_max}. d = 16: no one-lane kernels (GROUP_ONLY); the native tile of the fp64 matrix instruction,
// so the forward / backward sweeps have the wave form (WAVE, i2c_wave.hpp) next to the group form.
struct Quadrotor12 {
  static constexpr int ID = 7, NX = 12, NU = 4, NZ = 16, NZT = 12, NP = 5, NA = 3;
  static constexpr int GROUP = 16;
  static constexpr bool GROUP_ONLY = true;
  static constexpr bool GROUP_FORWARD_AUTO = false;
  static constexpr bool WAVE = true;   // one-wavefront-per-trajectory kernels (i2c_wave.hpp): d = 16 models only
  static constexpr bool QUAD = true;   // four-trajectories-per-wavefront forward kernel (i2c_quad.hpp), d = 16 form
  // the quad kernel is the DEFAULT forward sweep as soon as wave-kernel waves would share a SIMD, i.e. above 1024 trajectories
  // (up to there every wave has a SIMD of its own and the shorter dependent chain of one trajectory per wave wins; round 5, with
  // the square-root update in both forms: B = 1024: wave 0.303 against quad 0.408 ms; 1152: 0.453 / 0.412; 1536: 0.469 / 0.419;
  // 2048: 0.502 / 0.427; 4096: 0.948 / 0.515 -- profiles/r5_quad12_wave_quad_crossover.txt)
  static constexpr int QUAD_FORWARD_MAX_B = 1 << 30;
  static constexpr int QUAD_FORWARD_MIN_B = 1025;
  // ... and the DEFAULT backward sweep above 2048 trajectories (the fused walk of four trajectories per wavefront against one:
  // B = 1536: 0.173 against 0.156 ms; 2048: 0.175 / 0.161; 3072: 0.189 / 0.256; 4096: 0.214 / 0.315)
  static constexpr int QUAD_BACKWARD_MIN_B = 2049;
  I2C_HD static constexpr int ang(int a) { return 3 + a; }
  I2C_HD static constexpr int obs_lin(int k) { return k; }  // z = xu, zT = x
  I2C_HD static constexpr int obs_dep(int) { return 0; }
  I2C_HD static constexpr int term_lin(int k) { return k; }
  I2C_HD static constexpr int term_dep(int) { return 0; }
  static constexpr int NY = 9;  // y = [p, angles, body rates]
  I2C_HD static constexpr int meas_lin(int k) { return k < 6 ? k : k + 3; }
  I2C_HD static constexpr int meas_dep(int) { return 0; }
  template <typename R> I2C_FN void measure(const R*, const R* x, const R*, const R*, R* y) {
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = x[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) y[6 + i] = x[9 + i];
  }
  template <typename R> I2C_FN void dynamics(const R* p, const R* xu, const R* sn, const R* cs, R* xn) {
    const R dt = R(0.05), arm = R(0.25), kq = R(0.05), ang_damp = R(0.5), grav = R(9.81);
    const R mass = p[0], Ixx = p[1], Iyy = p[2], Izz = p[3], u_max = p[4];
    const R f1 = r_clip(xu[12], R(0), u_max), f2 = r_clip(xu[13], R(0), u_max);
    const R f3 = r_clip(xu[14], R(0), u_max), f4 = r_clip(xu[15], R(0), u_max);
    const R thrust = (f1 + f2) + (f3 + f4);
    const R tx = arm * (f2 - f4), ty = arm * (f3 - f1), tz = kq * ((f1 - f2) + (f3 - f4));
    const R sph = sn[0], cph = cs[0], sth = sn[1], cth = cs[1], sps = sn[2], cps = cs[2];
    const R wx = xu[9], wy = xu[10], wz = xu[11];
    // body rates: I w' = tau - w x (I w). Reciprocals of the wave-uniform parameters instead of divisions: an IEEE fp64
    // division is ~30 instructions, and there would be eight of them in each of the three evaluations per transform.
    const R damp = r_rcp(R(1) + dt * ang_damp), imass = r_rcp(mass), iIxx = r_rcp(Ixx), iIyy = r_rcp(Iyy), iIzz = r_rcp(Izz);
    const R wxn = (wx + dt * (tx - (Izz - Iyy) * wy * wz) * iIxx) * damp;
    const R wyn = (wy + dt * (ty - (Ixx - Izz) * wz * wx) * iIyy) * damp;
    const R wzn = (wz + dt * (tz - (Iyy - Ixx) * wx * wy) * iIzz) * damp;
    // world-frame acceleration: thrust along the body z axis R(roll, pitch, yaw) e3
    const R am = thrust * imass;
    const R vxn = xu[6] + dt * am * (cph * sth * cps + sph * sps);
    const R vyn = xu[7] + dt * am * (cph * sth * sps - sph * cps);
    const R vzn = xu[8] + dt * (am * (cph * cth) - grav);
    // Euler-angle rates from the new body rates
    const R icth = r_rcp(cth), tth = sth * icth;
    const R roll_d = wxn + tth * (sph * wyn + cph * wzn);
    const R pitch_d = cph * wyn - sph * wzn;
    const R yaw_d = (sph * wyn + cph * wzn) * icth;
    xn[0] = xu[0] + dt * vxn;
    xn[1] = xu[1] + dt * vyn;
    xn[2] = xu[2] + dt * vzn;
    xn[3] = xu[3] + dt * roll_d;
    xn[4] = xu[4] + dt * pitch_d;
    xn[5] = xu[5] + dt * yaw_d;
    xn[6] = vxn;
    xn[7] = vyn;
    xn[8] = vzn;
    xn[9] = wxn;
    xn[10] = wyn;
    xn[11] = wzn;
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R*, const R*, R* z) {
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = xu[i];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R*, const R*, R* z) {
#pragma unroll
    for (int i = 0; i < 12; ++i) z[i] = x[i];
  }
};

}  // namespace i2c
