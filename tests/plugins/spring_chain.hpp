// A second OUT-OF-TREE model for tests/test_model_plugin_identity.py: a cart on a hardening spring driven through a first-order
// actuator -- three states, one action, the IDENTITY observation of (x, u). Not one of the models compiled into libi2c_hip.so, not
// one of the reference's. d = 4 = one block of the quad kernels, in which the action shares a block with three state rows: the
// shape that exercises the square-root form of the identity-observation update (q_kalman_sqrt) on a single, mixed block.
//   x = [p, v, w], u = [command];   w' = w + dt (-a w + clip(u, -u_max, u_max)),  v' = v + dt (-k p - h p^3 - c v + w),  p' = p + dt v'
//   z = [p, v, w, u],  zT = [p, v, w];   params = {dt, k, h, c, a, u_max}.  NumPy twin: tests/test_model_plugin_identity.py.
#pragma once

namespace i2c {

struct SpringChain : ModelDefaults {
  static constexpr int NX = 3, NU = 1, NZ = 4, NZT = 3, NP = 6, NA = 0, NY = 3;
  static constexpr int GROUP = 4;
  static constexpr bool QUAD = true;
  I2C_HD static constexpr int obs_lin(int k) { return k; }  // z = xu
  I2C_HD static constexpr int obs_dep(int) { return 0; }
  I2C_HD static constexpr int term_lin(int k) { return k; }  // zT = x
  I2C_HD static constexpr int term_dep(int) { return 0; }
  I2C_HD static constexpr int meas_lin(int k) { return k; }
  I2C_HD static constexpr int meas_dep(int) { return 0; }
  template <typename R> I2C_FN void dynamics(const R* p, const R* xu, const R*, const R*, R* xn) {
    const R dt = p[0], k = p[1], h = p[2], c = p[3], a = p[4], u_max = p[5];
    const R u = r_clip(xu[3], -u_max, u_max);
    xn[2] = xu[2] + dt * (u - a * xu[2]);
    xn[1] = xu[1] + dt * (xu[2] - k * xu[0] - h * xu[0] * xu[0] * xu[0] - c * xu[1]);
    xn[0] = xu[0] + dt * xn[1];
  }
  template <typename R> I2C_FN void observe(const R*, const R* xu, const R*, const R*, R* z) {
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = xu[i];
  }
  template <typename R> I2C_FN void observe_terminal(const R*, const R* x, const R*, const R*, R* z) {
#pragma unroll
    for (int i = 0; i < 3; ++i) z[i] = x[i];
  }
  template <typename R> I2C_FN void measure(const R* p, const R* x, const R* sn, const R* cs, R* y) { observe_terminal(p, x, sn, cs, y); }
};

}  // namespace i2c
