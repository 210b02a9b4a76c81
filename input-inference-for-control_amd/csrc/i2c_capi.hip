// C ABI (include/i2c_hip.h): argument checks and dispatch over (model_id, dtype) to the per-pair translation units
// (i2c_model_tu.hip -> i2c_impl.hpp). No kernel code is compiled here.
#include <dlfcn.h>

#include <mutex>

#include "i2c_entry.hpp"

namespace {

// Out-of-tree models (i2c_register_model / i2c_load_model): an append-only table of per-dtype ops tables. The only mutable
// state of the library: entries are written once under the mutex and never change or move afterwards, so readers of a
// published id need no lock.
struct PluginEntry {
  const i2c::ModelOps* ops[3];  // I2C_F64, I2C_F32, I2C_F64_F32S (nullptr: that precision was not built)
};  // (a library opened by i2c_load_model stays loaded for the life of the process: its handle is never closed once registered)
PluginEntry g_plugins[I2C_MAX_PLUGIN_MODELS];
int g_n_plugins = 0;
std::mutex g_plugin_mutex;

const i2c::ModelOps* find_ops(int model_id, int dtype) {
  if (dtype != I2C_F64 && dtype != I2C_F32 && dtype != I2C_F64_F32S) return nullptr;
  if (model_id >= I2C_MODEL_PLUGIN_BASE) {
    const int k = model_id - I2C_MODEL_PLUGIN_BASE;
    int n;
    {
      std::lock_guard<std::mutex> lock(g_plugin_mutex);
      n = g_n_plugins;
    }
    return k < n ? g_plugins[k].ops[dtype] : nullptr;
  }
  switch (model_id) {
#define I2C_CASE(ID, MODEL, name) \
  case ID: return dtype == I2C_F64 ? i2c::ops_##name##_f64() : (dtype == I2C_F32 ? i2c::ops_##name##_f32() : i2c::ops_##name##_f64s());
    I2C_FOR_EACH_MODEL(I2C_CASE)
#undef I2C_CASE
    default: return nullptr;
  }
}

// the scalar fields alone (what the resolvers i2c_kernel_family / i2c_backward_schedule read: no buffer needs to exist yet)
int check_problem_shape(const I2cProblem* p);
int check_problem(const I2cProblem* p) {
  const int rc = check_problem_shape(p);
  if (rc != I2C_OK) return rc;
  if (!p->x0 || !p->sig_x0 || !p->alpha || !p->feedforward) return I2C_EINVAL;
  if (p->has_x_terminal && !p->temp) return I2C_EINVAL;
  return I2C_OK;
}
int check_problem_shape(const I2cProblem* p) {
  if (!p || p->abi_version != I2C_ABI_VERSION) return I2C_EINVAL;
  if (p->B < 1 || p->T < 1 || p->T > 65535) return I2C_EINVAL;  // status words keep t + 1 in 16 bits
  if (p->inference < I2C_INF_CUBATURE || p->inference > I2C_INF_GAUSS_HERMITE) return I2C_EINVAL;
  if (p->inference == I2C_INF_GAUSS_HERMITE && (p->gh_degree < 1 || p->gh_degree > I2C_MAX_GH_DEGREE)) return I2C_EINVAL;
  if (p->t0 < 0 || p->t0 >= p->T) return I2C_EINVAL;
  if (p->post_layout != 0 && p->post_layout != 1) return I2C_EINVAL;
  if (p->post_layout == 1) {  // trajectory-major posterior: only the models whose every kernel knows it
    I2cDims d;
    const i2c::ModelOps* ops = find_ops(p->model_id, I2C_F64);
    if (!ops) return I2C_EINVAL;
    ops->dims(&d);
    if (!d.wave) return I2C_ENOTSUP;
  }
  return I2C_OK;
}

}  // namespace

// Check the problem, find the (model, dtype) table, call one of its entries.
#define I2C_DISPATCH(p, CALL)                                             \
  do {                                                                    \
    const int rc_ = check_problem(p);                                     \
    if (rc_ != I2C_OK) return rc_;                                        \
    const i2c::ModelOps* ops_ = find_ops((p)->model_id, (p)->dtype);      \
    if (!ops_) return I2C_EINVAL;                                         \
    return ops_->CALL;                                                    \
  } while (0)

extern "C" {

int i2c_abi_version(void) { return I2C_ABI_VERSION; }

size_t i2c_problem_size(void) { return sizeof(I2cProblem); }

const char* i2c_build_info(void) {
#ifdef I2C_HOST_SIM
  return "i2c host-simulation build (CPU, tests only)";
#else
  return "i2c hip build: gfx950 (MI355X), wave64; lane kernels (one trajectory per lane) + group kernels (4/8/16 lanes per trajectory, LDS exchange) + wave kernels (one wavefront per trajectory, v_mfma_f64_16x16x4_f64) + quad kernels (four trajectories per wavefront, v_mfma_f64_4x4x4_4b_f64)";
#endif
}

// the two resolvers read scalar fields only: they answer before any buffer of the problem exists
#define I2C_DISPATCH_SHAPE(p, CALL)                                       \
  do {                                                                    \
    const int rc_ = check_problem_shape(p);                               \
    if (rc_ != I2C_OK) return rc_;                                        \
    const i2c::ModelOps* ops_ = find_ops((p)->model_id, (p)->dtype);      \
    if (!ops_) return I2C_EINVAL;                                         \
    return ops_->CALL;                                                    \
  } while (0)

int i2c_backward_schedule(const I2cProblem* p) {
  if (p && (p->backward_mode < I2C_BWD_AUTO || p->backward_mode > I2C_BWD_CHUNKED)) return I2C_EINVAL;
  I2C_DISPATCH_SHAPE(p, plan(p));
}

int i2c_kernel_family(const I2cProblem* p, int sweep) {
  if (sweep < I2C_SWEEP_FORWARD || sweep > I2C_SWEEP_CHUNK_STITCH) return I2C_EINVAL;
  I2C_DISPATCH_SHAPE(p, family(p, sweep));
}

size_t i2c_workspace_bytes(int model_id, int dtype, int B, int T) {
  if (B < 1 || T < 1) return 0;
  const i2c::ModelOps* ops = find_ops(model_id, I2C_F64);
  return ops ? (dtype == I2C_F32 ? 4 : 8) * ops->workspace_elems(B, T) : 0;  // the workspace is arithmetic-typed (fp64 in I2C_F64_F32S)
}

int i2c_register_model(int abi_version, const I2cModelOps* ops_f64, const I2cModelOps* ops_f32, const I2cModelOps* ops_f64s,
                       I2cDims* dims_out) {
  if (abi_version != I2C_ABI_VERSION || !ops_f64) return I2C_EINVAL;
  const i2c::ModelOps* t[3] = {reinterpret_cast<const i2c::ModelOps*>(ops_f64), reinterpret_cast<const i2c::ModelOps*>(ops_f32),
                               reinterpret_cast<const i2c::ModelOps*>(ops_f64s)};
  I2cDims d;
  t[0]->dims(&d);
  if (d.nx < 1 || d.nx > I2C_MAX_NX || d.nu < 1 || d.nu > I2C_MAX_NU || d.nz < 1 || d.nz > I2C_MAX_NZ || d.nzt < 0 || d.nzt > I2C_MAX_NZ ||
      d.n_params < 0 || d.n_params > I2C_MAX_PARAMS || d.ny < 0 || d.ny > I2C_MAX_NZ)
    return I2C_EINVAL;  // the host constants of I2cProblem have fixed capacities
  std::lock_guard<std::mutex> lock(g_plugin_mutex);
  for (int k = 0; k < g_n_plugins; ++k)  // registering the same tables again returns the id they already have
    if (g_plugins[k].ops[0] == t[0]) {
      if (dims_out) *dims_out = d;
      return I2C_MODEL_PLUGIN_BASE + k;
    }
  if (g_n_plugins >= I2C_MAX_PLUGIN_MODELS) return I2C_ENOTSUP;
  g_plugins[g_n_plugins] = PluginEntry{{t[0], t[1], t[2]}};
  if (dims_out) *dims_out = d;
  return I2C_MODEL_PLUGIN_BASE + g_n_plugins++;
}

int i2c_load_model(const char* path, I2cDims* dims_out) {
  if (!path) return I2C_EINVAL;
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return I2C_EINVAL;
  typedef int (*abi_fn)(void);
  typedef const I2cModelOps* (*ops_fn)(int);
  abi_fn abi = (abi_fn)dlsym(h, "i2c_model_abi_version");
  ops_fn ops = (ops_fn)dlsym(h, "i2c_model_ops");
  if (!abi || !ops) {
    dlclose(h);
    return I2C_EINVAL;
  }
  const int id = i2c_register_model(abi(), ops(I2C_F64), ops(I2C_F32), ops(I2C_F64_F32S), dims_out);
  if (id < 0) dlclose(h);  // (a library registered twice keeps its first handle: dlopen reference-counts)
  return id;
}

int i2c_query(int model_id, I2cDims* out) {
  const i2c::ModelOps* ops = find_ops(model_id, I2C_F64);
  if (!out || !ops) return I2C_EINVAL;
  ops->dims(out);
  return I2C_OK;
}

int i2c_forward_sweep(const I2cProblem* p, const void* prior, void* fwd, void* prior_out, int32_t* status,
                      void* stream) {
  if (!prior || !fwd || !status) return I2C_EINVAL;
  I2C_DISPATCH(p, forward(p, prior, fwd, prior_out, status, stream));
}

int i2c_backward_sweep(const I2cProblem* p, const void* fwd, void* xm, void* post, void* zpost, void* cell_stats,
                       void* term_stats, int32_t* status, void* stream) {
  if (!fwd || !post || !term_stats || !status) return I2C_EINVAL;
  I2C_DISPATCH(p, backward(p, fwd, xm, post, zpost, cell_stats, term_stats, status, stream));
}

int i2c_mstep(const I2cProblem* p, const void* term_stats, double alpha_update_tol, int update, void* stats_out,
              void* stream) {
  if (!term_stats || !stats_out) return I2C_EINVAL;
  I2C_DISPATCH(p, mstep(p, term_stats, alpha_update_tol, update, stats_out, stream));
}

int i2c_propagate(const I2cProblem* p, const void* post, void* prop, void* prop_stats, int use_expert_controller,
                  int32_t* status, void* stream) {
  if (!post || !prop || !prop_stats || !status) return I2C_EINVAL;
  I2C_DISPATCH(p, propagate(p, post, prop, prop_stats, use_expert_controller, status, stream));
}

int i2c_learn(const I2cProblem* p, void* post, void* fwd, void* xm, void* zpost, void* cell_stats, void* term_stats,
              double alpha_update_tol, int tau, int n_iters, void* stats_hist, int32_t* status, void* stream) {
  if (!post || !fwd || !term_stats || !stats_hist || !status || n_iters < 0) return I2C_EINVAL;
  I2C_DISPATCH(p, learn(p, post, fwd, xm, zpost, cell_stats, term_stats, alpha_update_tol, tau, n_iters, stats_hist,
                        status, stream));
}

int i2c_learn_propagate(const I2cProblem* p, void* post, void* fwd, void* xm, void* zpost, void* cell_stats, void* term_stats, void* prop,
                        void* prop_hist, double alpha_update_tol, int tau, int n_iters, void* stats_hist, int use_expert_controller,
                        int overlap, int32_t* status, void* stream) {
  if (!post || !fwd || !term_stats || !prop || !prop_hist || !stats_hist || !status || n_iters < 0) return I2C_EINVAL;
  I2C_DISPATCH(p, learn_propagate(p, post, fwd, xm, zpost, cell_stats, term_stats, prop, prop_hist, alpha_update_tol, tau, n_iters,
                                  stats_hist, use_expert_controller, overlap, status, stream));
}

int i2c_rollout(const I2cProblem* p, const void* post, int n_rollouts, int policy, const void* eps_x0,
                const void* eps_x, const void* eps_u, void* xu, void* z, void* x_final, void* z_term, void* stream) {
  if (!post || n_rollouts < 1 || policy < 0 || policy > 2) return I2C_EINVAL;
  I2C_DISPATCH(p, rollout(p, post, n_rollouts, policy, eps_x0, eps_x, eps_u, xu, z, x_final, z_term, stream));
}

int i2c_riccati_sweep(const I2cProblem* p, const void* prior_out, const void* fwd, const void* xm, void* post, void* ric,
                      int32_t* status, void* stream) {
  if (!prior_out || !fwd || !xm || !post || !ric || !status) return I2C_EINVAL;
  if (p && p->inference != I2C_INF_LINEARIZE) return I2C_EINVAL;
  I2C_DISPATCH(p, riccati(p, prior_out, fwd, xm, post, ric, status, stream));
}

int i2c_mpc_step(const I2cProblem* p, const I2cMpcStep* m, void* stream) {
  if (!m || !m->post || !m->fwd || !m->term_stats || !m->cell_init || !m->status || m->n_iter < 0) return I2C_EINVAL;
  if (m->do_filter && (!m->y || !m->u)) return I2C_EINVAL;
  if (p && p->alpha_cell && !m->alpha_init) return I2C_EINVAL;
  I2C_DISPATCH(p, mpc_step(p, m, stream));
}

int i2c_shift_horizon(const I2cProblem* p, void* post, const void* cell_init, const void* alpha_init, const void* z_new,
                      void* action, void* stream) {
  if (!post || !cell_init) return I2C_EINVAL;
  if (p && p->alpha_cell && !alpha_init) return I2C_EINVAL;
  I2C_DISPATCH(p, shift(p, post, cell_init, alpha_init, z_new, action, stream));
}

int i2c_ckf_filter(const I2cProblem* p, const double* sig_zeta, const void* y, const void* u, void* mu, void* cov,
                   int32_t* status, void* stream) {
  if (!sig_zeta || !y || !u || !mu || !cov || !status) return I2C_EINVAL;
  I2C_DISPATCH(p, ckf(p, sig_zeta, y, u, mu, cov, status, stream));
}

}  // extern "C"
