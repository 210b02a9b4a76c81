// An OUT-OF-TREE model for tests/test_model_plugin.py: a controlled Van der Pol oscillator, which is not one of the models
// compiled into libi2c_hip.so (and not one of the reference's). Built by `python build.py --model tests/plugins/van_der_pol.hpp`
// into lib/libi2c_model_van_der_pol.so and registered at run time with i2c_load_model -- nothing in the tree names it.
//   x = [position, velocity], u = [force];  x1' = x1 + dt x2',  x2' = x2 + dt (mu (1 - x1^2) x2 - x1 + clip(u, -u_max, u_max))
//   z = [x1, x2, x2 / (1 + x2^2), u]  (a general observation: one output is not a pass-through),  zT = [x1, x2]
//   params = {mu, dt, u_max}.  The NumPy twin (host protocol + oracle) is in tests/test_model_plugin.py.
#pragma once

namespace i2c {

struct VanDerPol : ModelDefaults {
  static constexpr int NX = 2, NU = 1, NZ = 4, NZT = 2, NP = 3, NA = 0, NY = 2;
  static constexpr int GROUP = 4;    // opt into the group kernels (4 lanes per trajectory) ...
  static constexpr bool QUAD = true;  // ... and the quad forward kernel (d = 3: a spare column in the joint's block, one action)
  // structure hints: z0, z1, z3 are pass-throughs of xu[0], xu[1], xu[2]; z2 is a function of inputs <= 1
  I2C_HD static constexpr int obs_lin(int k) { return k < 2 ? k : (k == 3 ? 2 : -1); }
  I2C_HD static constexpr int obs_dep(int) { return 1; }
  I2C_HD static constexpr int term_lin(int k) { return k; }
  I2C_HD static constexpr int meas_lin(int k) { return k; }
  template <typename R> I2C_FN void dynamics(const R* p, const R* xu, const R*, const R*, R* xn) {
    const R mu = p[0], dt = p[1], u_max = p[2];
    const R u = r_clip(xu[2], -u_max, u_max);
    const R acc = mu * (R(1) - xu[0] * xu[0]) * xu[1] - xu[0] + u;
    xn[1] = xu[1] + dt * acc;
    xn[0] = xu[0] + dt * xn[1];
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R*, const R*, R* z) {
    z[0] = xu[0];
    z[1] = xu[1];
    z[2] = xu[1] * r_rcp(R(1) + xu[1] * xu[1]);  // (functors are also instantiated on dual numbers: + - * /, r_rcp, r_clip, r_exp only)
    z[3] = xu[2];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R*, const R*, R* z) {
    z[0] = x[0];
    z[1] = x[1];
  }
  template <typename R> I2C_FN void measure(const R* p, const R* x, const R* sn, const R* cs, R* y) { observe_terminal(p, x, sn, cs, y); }
};

}  // namespace i2c
