// ThreadSanitizer driver for the multi-lane kernels' host simulation (tests only): the G lanes of a group -- or the 64 lanes of a
// wave kernel's wavefront -- run as threads with a barrier wherever the device code has its LDS fence or a cross-lane
// instruction, so a missing fence shows up here as a data race on the exchange region. Drives forward, backward, propagation and
// the filter step of the 12-state quadrotor through the C ABI on a tiny problem: group kernels (16 lanes per trajectory), wave
// kernels (cubature), wave kernels (Linearize), quad forward + quad backward kernels (plain and with the terminal state prior of
// covariance control), quad forward + wave backward. Built and run by tools/tsan_group.sh.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../include/i2c_hip.h"

static int run(const int group_lanes, const int inference, const int post_layout = 0, const bool cov_control = false, const int backward_mode = I2C_BWD_AUTO) {
  I2cDims d;
  if (i2c_query(I2C_MODEL_QUADROTOR12, &d) != I2C_OK) return 2;
  const int B = 3, T = 4, nx = d.nx, nu = d.nu, nz = d.nz, D = nx + nu;
  I2cProblem p;
  std::memset(&p, 0, sizeof(p));
  p.abi_version = I2C_ABI_VERSION;
  p.model_id = I2C_MODEL_QUADROTOR12;
  p.dtype = I2C_F64;
  p.B = B;
  p.T = T;
  p.has_Qf = 1;
  p.terminal_cell = T - 1;
  p.inference = inference;
  p.group_lanes = group_lanes;
  p.post_layout = post_layout;  // 1: trajectory-major posterior (what the quad forward kernel of this model addresses)
  p.quad_alpha = 1.0;
  p.dtemp = 1.0;
  p.backward_mode = backward_mode;
  if (cov_control) {
    p.has_x_terminal = 1;
    for (int i = 0; i < 12; ++i) {
      p.mu_x_term[i] = 0.05 * i;
      p.sig_x_term[i * (i + 1) / 2 + i] = 1e-3;
    }
  }
  auto diag = [](double* packed, int n, double v) {
    for (int i = 0; i < n; ++i) packed[i * (i + 1) / 2 + i] = v;
  };
  diag(p.sig_eta, nx, 1e-5);
  diag(p.sig_xi0, nz, 0.5);
  diag(p.QR, nz, 2.0);
  diag(p.sig_xiT0, d.nzt, 0.5);
  diag(p.Qf, d.nzt, 2.0);
  for (int i = 0; i < 3; ++i) p.zg[i] = p.zg_term[i] = 1.0;
  const double params[5] = {1.0, 0.02, 0.02, 0.04, 6.0};
  for (int i = 0; i < 5; ++i) p.model_params[i] = params[i];
  std::vector<double> x0(nx * B, 0.0), sx0(nx * (nx + 1) / 2 * B, 0.0), alpha(B, 1.0), temp(B, 1.0);
  for (int i = 0; i < nx; ++i)
    for (int b = 0; b < B; ++b) {
      x0[i * B + b] = 1e-2 * (i + b);
      sx0[(i * (i + 1) / 2 + i) * B + b] = 1e-5;
    }
  std::vector<uint8_t> ff(T, 0);  // feedback mode: the larger code path
  std::vector<double> post((size_t)T * d.e_post * B, 0.0), fwd((size_t)T * d.e_fwd * B), prop((size_t)T * d.e_prop * B), pstat(3 * B);
  std::vector<double> term((size_t)(4 + d.nzt + d.nzt * (d.nzt + 1) / 2) * B), stats(4 * B);
  std::vector<int32_t> status(B, 0);
  std::vector<double> xm((size_t)T * (nx + nx * (nx + 1) / 2) * B), cstat((size_t)T * 2 * B);
  for (int t = 0; t < T; ++t)
    for (int b = 0; b < B; ++b) {
      const size_t es = post_layout ? 1 : (size_t)B;  // element stride
      double* c = &post[post_layout ? ((size_t)t * B + b) * d.e_post : (size_t)t * d.e_post * B + b];
      for (int i = 0; i < nx; ++i) c[(size_t)i * es] = x0[i * B + b];
      for (int i = nx; i < D; ++i) c[(size_t)i * es] = 2.45;
      for (int i = 0; i < D; ++i) c[(size_t)(D + i * (i + 1) / 2 + i) * es] = i < nx ? 1e-5 : 1e-2;
    }
  p.x0 = x0.data();
  p.sig_x0 = sx0.data();
  p.alpha = alpha.data();
  p.temp = temp.data();
  p.feedforward = ff.data();
  int rc = 0;
  for (int it = 0; it < 2 && rc == 0; ++it) {
    rc = i2c_forward_sweep(&p, post.data(), fwd.data(), nullptr, status.data(), nullptr);
    if (!rc) rc = i2c_backward_sweep(&p, fwd.data(), backward_mode == I2C_BWD_TWO_PASS ? xm.data() : nullptr, post.data(), nullptr,
                                     backward_mode == I2C_BWD_TWO_PASS ? cstat.data() : nullptr, term.data(), status.data(), nullptr);
    if (!rc) rc = i2c_mstep(&p, term.data(), 0.5, 1, stats.data(), nullptr);
    if (!rc && inference == I2C_INF_CUBATURE) rc = i2c_propagate(&p, post.data(), prop.data(), pstat.data(), 1, status.data(), nullptr);
  }
  std::vector<double> zeta(d.ny * (d.ny + 1) / 2, 0.0), y(d.ny * B, 0.01), u(nu * B, 2.45);
  for (int i = 0; i < d.ny; ++i) zeta[i * (i + 1) / 2 + i] = 1e-4;
  if (!rc && inference == I2C_INF_CUBATURE) rc = i2c_ckf_filter(&p, zeta.data(), y.data(), u.data(), x0.data(), sx0.data(), status.data(), nullptr);
  bool finite = true;
  for (double v : post) finite = finite && std::isfinite(v);
  std::printf("lanes %2d inference %d family fwd %d bwd %d: rc %d status %d %d %d finite %d cost %.6f\n", group_lanes, inference,
              i2c_kernel_family(&p, I2C_SWEEP_FORWARD), i2c_kernel_family(&p, I2C_SWEEP_BACKWARD), rc, status[0], status[1], status[2],
              (int)finite, stats[2 * B]);
  return (rc == 0 && finite && status[0] == 0) ? 0 : 1;
}

int main() {
  int bad = run(16, I2C_INF_CUBATURE);
  bad += run(64, I2C_INF_CUBATURE);
  bad += run(64, I2C_INF_LINEARIZE);
  bad += run(I2C_LANES_QUAD, I2C_INF_CUBATURE, 1);        // quad kernels, both sweeps (four trajectories per wavefront: 3 + one spare slot)
  bad += run(I2C_LANES_QUAD, I2C_INF_CUBATURE, 1, true);  // ... with the tempered terminal state prior at the end of the backward chain
  bad += run(I2C_LANES_QUAD, I2C_INF_CUBATURE, 1, false, I2C_BWD_TWO_PASS);  // quad forward kernel + wave backward sweep (two-pass schedule)
  return bad;
}
