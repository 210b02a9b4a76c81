// Per-(model, dtype) entry points of the library as a table of plain function pointers.
// The kernels of every (model, dtype) pair live in their own translation unit (i2c_model_tu.hip compiled once per
// pair, in parallel); the C-ABI translation unit (i2c_capi.hip) only sees these tables.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/i2c_hip.h"

namespace i2c {

// (the C header names this table I2cModelOps, opaque: a model library hands it to i2c_register_model)
struct ModelOps {
  int (*forward)(const I2cProblem*, const void* prior, void* fwd, void* prior_out, int32_t* status, void* stream);
  int (*backward)(const I2cProblem*, const void* fwd, void* xm, void* post, void* zpost, void* cell_stats,
                  void* term_stats, int32_t* status, void* stream);
  int (*mstep)(const I2cProblem*, const void* term_stats, double tol, int update, void* stats_out, void* stream);
  int (*learn)(const I2cProblem*, void* post, void* fwd, void* xm, void* zpost, void* cell_stats, void* term_stats,
               double tol, int tau, int n_iters, void* stats_hist, int32_t* status, void* stream);
  int (*ckf)(const I2cProblem*, const double* sig_zeta, const void* y, const void* u, void* mu, void* cov,
             int32_t* status, void* stream);
  int (*rollout)(const I2cProblem*, const void* post, int n_rollouts, int policy, const void* eps_x0, const void* eps_x,
                 const void* eps_u, void* xu, void* z, void* x_final, void* z_term, void* stream);
  int (*propagate)(const I2cProblem*, const void* post, void* prop, void* prop_stats, int use_expert, int32_t* status,
                   void* stream);
  int (*riccati)(const I2cProblem*, const void* prior_out, const void* fwd, const void* xm, void* post, void* ric,
                 int32_t* status, void* stream);
  int (*mpc_step)(const I2cProblem*, const I2cMpcStep*, void* stream);
  void (*dims)(I2cDims*);
  size_t (*workspace_elems)(int B, int T);
  int (*plan)(const I2cProblem*);  // the backward schedule that will run (I2C_BWD_*), or an error code
  int (*shift)(const I2cProblem*, void* post, const void* cell_init, const void* alpha_init, const void* z_new, void* action,
               void* stream);
  int (*family)(const I2cProblem*, int sweep);
  int (*learn_propagate)(const I2cProblem*, void* post, void* fwd, void* xm, void* zpost, void* cell_stats, void* term_stats, void* prop,
                         void* prop_hist, double tol, int tau, int n_iters, void* stats_hist, int use_expert, int overlap, int32_t* status,
                         void* stream);
};

// defined by the translation units generated from i2c_model_tu.hip
#define I2C_FOR_EACH_MODEL(X)                                                                               \
  X(I2C_MODEL_PENDULUM, Pendulum, pendulum)                                                                 \
  X(I2C_MODEL_PENDULUM_ACTREG, PendulumActReg, pendulum_actreg)                                             \
  X(I2C_MODEL_CARTPOLE, Cartpole, cartpole)                                                                 \
  X(I2C_MODEL_DOUBLE_CARTPOLE, DoubleCartpole, double_cartpole)                                             \
  X(I2C_MODEL_LINEAR, Linear, linear)                                                                       \
  X(I2C_MODEL_LINEAR_MINENERGY, LinearMinEnergy, linear_minenergy)                                          \
  X(I2C_MODEL_QUADROTOR, Quadrotor, quadrotor)                                                              \
  X(I2C_MODEL_QUADROTOR12, Quadrotor12, quadrotor12)
#define I2C_DECLARE_OPS(ID, MODEL, name) \
  const ModelOps* ops_##name##_f64();    \
  const ModelOps* ops_##name##_f32();    \
  const ModelOps* ops_##name##_f64s();
I2C_FOR_EACH_MODEL(I2C_DECLARE_OPS)
#undef I2C_DECLARE_OPS

}  // namespace i2c
