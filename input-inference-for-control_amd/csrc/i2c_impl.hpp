// Kernels and their launchers for one (model, dtype) pair: Impl<M, R>, instantiated by i2c_model_tu.hip.
//
// Built two ways:
//   hipcc --offload-arch=gfx950            -> libi2c_hip.so      (THE product; the only library
//                                                                 the Python package ever loads)
//   g++ -x c++ -DI2C_HOST_SIM              -> libi2c_hostsim.so  (tests only: the same cell math
//                                                                 looped on the CPU so that kernel
//                                                                 numerics can be checked against the
//                                                                 oracle on a box without a GPU)
#pragma once
#include "i2c_entry.hpp"
#include "i2c_cell.hpp"
#include "i2c_linearize.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace i2c {

#ifndef I2C_HOST_SIM
// One wavefront per workgroup for the sequential sweeps: at B = 4096 that is 64 workgroups on
// 64 different CUs, each wave with a CU's issue ports, L1 and scalar cache to itself.
constexpr int SWEEP_BLOCK = 64;
constexpr int CELL_BLOCK = 256;
// Experiment knob (not part of the ABI): I2C_SWEEP_LANES=<n<=64> launches the sequential sweeps with
// n active lanes per wavefront (more, emptier waves on more SIMDs).
static int sweep_lanes() {
  static int v = [] {
    const char* e = getenv("I2C_SWEEP_LANES");
    const int n = e ? atoi(e) : SWEEP_BLOCK;
    return (n >= 1 && n <= SWEEP_BLOCK) ? n : SWEEP_BLOCK;
  }();
  return v;
}

template <class M, typename R, bool LEAN, bool GRID = false>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_forward(const Consts<M, R> c, const FwdArgs<R> a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < c.B) forward_sweep_body<M, R, LEAN, GRID>(c, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_forward_lin(const Consts<M, R> c, const FwdArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) forward_lin_body<M, R>(c, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_bwd_lin(const Consts<M, R> c, const CellArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) backward_lin_body<M, R>(c, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_riccati(const Consts<M, R> c, const RiccatiArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) riccati_body<M, R>(c, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_scan(const Consts<M, R> c, const ScanArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) backward_scan_body<M, R>(c, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(CELL_BLOCK) void k_cell(const Consts<M, R> c, const CellArgs<R> a) {
  const int b = blockIdx.x * CELL_BLOCK + threadIdx.x;
  const int t = blockIdx.y;
  if (b < c.B) backward_cell_body<M, R>(c, a, t, b);
}
template <class M, typename R, bool GRID = false>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_bwd_fused(const Consts<M, R> c, const CellArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) backward_fused_body<M, R, GRID>(c, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_chunk_compose(const Consts<M, R> c, const ChunkArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) chunk_compose_body<M, R>(c, a, blockIdx.y, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_chunk_stitch(const Consts<M, R> c, const ChunkArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) chunk_stitch_body<M, R>(c, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_chunk_walk(const Consts<M, R> c, const ChunkArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) chunk_walk_body<M, R>(c, a, blockIdx.y, b);
}
// sum the per-cell cost statistics over t: REDUCE_PARTS lanes per trajectory, fixed summation order
constexpr int REDUCE_PARTS = 8;
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK* REDUCE_PARTS) void k_reduce(const Consts<M, R> c, const CellArgs<R> a,
                                                                      const MstepArgs<R> ms, const int T_mstep) {
  __shared__ R sm[REDUCE_PARTS][SWEEP_BLOCK], sv[REDUCE_PARTS][SWEEP_BLOCK];
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  R m = R(0), v = R(0);
  if (b < c.B) reduce_partial<M, R>(c, a.cell_stats, b, threadIdx.y, REDUCE_PARTS, &m, &v);
  sm[threadIdx.y][threadIdx.x] = m;
  sv[threadIdx.y][threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.y == 0 && b < c.B) {
#pragma unroll
    for (int q = 1; q < REDUCE_PARTS; ++q) {
      m += sm[q][threadIdx.x];
      v += sv[q][threadIdx.x];
    }
    a.term_stats[(long)c.B + b] = m;
    a.term_stats[2 * (long)c.B + b] = v;
    if (ms.alpha) {  // i2c_learn: the temperature M-step rides on the reduction (one launch less per EM iteration)
      Consts<M, R> cm = c;
      cm.T = T_mstep;  // c.T is the number of summands here (cells or chunks), the M-step needs the horizon
      mstep_body<M, R>(cm, ms, b);
    }
  }
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_mstep(const Consts<M, R> c, const MstepArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) mstep_body<M, R>(c, a, b);
}
template <class M, typename R, bool GRID = false>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_propagate(const Consts<M, R> c, const PropArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) propagate_body<M, R, GRID>(c, a, b);
}
template <class M, typename R> struct ZetaArg {
  R v[sym(M::NY)];
};
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_ckf(const Consts<M, R> c, const ZetaArg<M, R> z, const CkfArgs<R> a) {
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (b < c.B) ckf_filter_body<M, R>(c, z.v, a, b);
}
template <class M, typename R>
__global__ __launch_bounds__(CELL_BLOCK) void k_mpc_shift(const Consts<M, R> c, const ShiftArgs<R> a) {
  const int b = blockIdx.x * CELL_BLOCK + threadIdx.x;
  if (b < c.B) mpc_shift_body<M, R>(c, a, blockIdx.y, b);
}
template <class M, typename R>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_rollout(const Consts<M, R> c, const RolloutArgs<R> a) {
  const long n = (long)blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  if (n < (long)a.n_rollouts * c.B) rollout_body<M, R>(c, a, (int)n);
}
static int launch_status() { return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH; }
#endif

template <typename R> static Rule<R> make_rule(const I2cProblem* p, int dim) {
  // CubatureQuadrature.weights, i2c/exp_types.py:40-49
  const double a = p->quad_alpha, lam = a * a * (dim + p->quad_kappa) - dim;
  const double wi = 1.0 / (2.0 * (dim + lam));
  const double w0 = 2.0 * lam * wi + (1.0 - a * a + p->quad_beta);
  const double W = w0 + 2.0 * dim * wi;
  Rule<R> r;
  r.sf = (R)std::sqrt(dim + lam);
  r.w0 = (R)w0;
  r.wi = (R)wi;
  r.unit = std::fabs(W - 1.0) < 1e-14;
  r.W = r.unit ? (R)1 : (R)W;
  r.gh_degree = 0;
  r.gh_points = 0;
  for (int q = 0; q < I2C_MAX_GH_DEGREE; ++q) r.gh_x[q] = r.gh_w[q] = (R)0;
  if (p->inference == I2C_INF_GAUSS_HERMITE) {  // GaussHermiteQuadrature.weights, i2c/exp_types.py:63-68
    r.sf = (R)std::sqrt(2.0);
    r.w0 = r.wi = (R)0;
    r.W = (R)1;
    r.unit = 1;
    r.gh_degree = p->gh_degree;
    long n = 1;
    for (int i = 0; i < dim; ++i) n *= p->gh_degree;
    r.gh_points = (int)n;
    for (int q = 0; q < p->gh_degree; ++q) {
      r.gh_x[q] = (R)p->gh_nodes[q];
      r.gh_w[q] = (R)(p->gh_weights[q] / std::sqrt(3.14159265358979323846));
    }
  }
  return r;
}

template <class M, typename R> static Consts<M, R> make_consts(const I2cProblem* p, double tol, int use_expert) {
  using C = Consts<M, R>;
  C c;
  std::memset(&c, 0, sizeof(c));
  c.B = p->B;
  c.T = p->T;
  c.has_Qf = p->has_Qf && M::NZT > 0;
  c.has_x_terminal = p->has_x_terminal;
  c.z_per_cell = p->z_per_cell && p->z != nullptr;
  c.use_expert = use_expert;
  c.terminal_cell = p->terminal_cell;
  c.inference = p->inference;
  auto is_diag = [](const double* W, int n) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j)
        if (W[i * (i + 1) / 2 + j] != 0.0) return 0;
    return 1;
  };
  c.qr_diag = is_diag(p->QR, C::NZ);
  c.qf_diag = is_diag(p->Qf, C::NZT);
  c.rule_xu = make_rule<R>(p, C::D);
  c.rule_x = make_rule<R>(p, C::NX);
  c.dtemp = (R)p->dtemp;
  c.tol = (R)tol;
  for (int i = 0; i < sym(C::NX); ++i) c.sig_eta[i] = (R)p->sig_eta[i];
  for (int i = 0; i < sym(C::NX); ++i) c.sig_eta_w[i] = c.rule_xu.W * c.sig_eta[i];
  for (int i = 0; i < sym(C::NZ); ++i) c.sig_xi0[i] = (R)p->sig_xi0[i];
  for (int i = 0; i < sym(C::NZ); ++i) c.QR[i] = (R)p->QR[i];
  for (int i = 0; i < sym(C::NZT); ++i) c.sig_xiT0[i] = (R)p->sig_xiT0[i];
  for (int i = 0; i < sym(C::NZT); ++i) c.Qf[i] = (R)p->Qf[i];
  for (int i = 0; i < C::NZ; ++i) c.zg[i] = (R)p->zg[i];
  for (int i = 0; i < C::NZT; ++i) c.zg_term[i] = (R)p->zg_term[i];
  for (int i = 0; i < C::NX; ++i) c.mu_x_term[i] = (R)p->mu_x_term[i];
  for (int i = 0; i < sym(C::NX); ++i) c.sig_x_term[i] = (R)p->sig_x_term[i];
  for (int i = 0; i < M::NP; ++i) c.params[i] = (R)p->model_params[i];
  return c;
}

// Chunk geometry of the chunked backward sweep: enough chunks to put ~64K lanes in flight, at least 4
// cells per chunk, at most 32 chunks.
static void chunk_geometry(int B, int T, int* n_chunks, int* chunk_len) {
  static const int forced = [] {  // experiment knob (not part of the ABI)
    const char* e = getenv("I2C_CHUNKS");
    return e ? atoi(e) : 0;
  }();
  int nc = forced > 0 ? forced : (65536 + B - 1) / B;
  if (nc > 32) nc = 32;
  if (nc > T / 4) nc = T / 4;
  if (nc < 1) nc = 1;
  const int len = (T + nc - 1) / nc;
  *chunk_len = len;
  *n_chunks = (T + len - 1) / len;
}
template <class M> static size_t workspace_elems(int B, int T) {
  int nc, len;
  chunk_geometry(B, T, &nc, &len);
  constexpr int NX = M::NX;
  return (size_t)nc * (size_t)B * (size_t)((NX + NX * NX + sym(NX)) + (NX + sym(NX)) + 2);
}

// ---- per-(model, dtype) entry points ------------------------------------------------------
template <class M, typename R> struct Impl {
  using C = Consts<M, R>;

  static int forward(const I2cProblem* p, const void* prior, void* fwd, void* prior_out, int32_t* status,
                     void* stream) {
    const C c = make_consts<M, R>(p, 0.0, p->inference == I2C_INF_LINEARIZE ? p->expert_controller : 0);
    FwdArgs<R> a{(const R*)prior, (R*)fwd, (R*)prior_out, (const R*)p->x0, (const R*)p->sig_x0,
                 (const R*)p->z,  (const R*)p->alpha, (const R*)p->alpha_cell, p->feedforward, status};
    if (p->inference == I2C_INF_LINEARIZE) {
#ifdef I2C_HOST_SIM
      (void)stream;
      for (int b = 0; b < p->B; ++b) forward_lin_body<M, R>(c, a, b);
      return I2C_OK;
#else
      hipLaunchKernelGGL((k_forward_lin<M, R>), dim3((p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK), dim3(SWEEP_BLOCK), 0,
                         (hipStream_t)stream, c, a);
      return launch_status();
#endif
    }
    if (p->inference == I2C_INF_GAUSS_HERMITE) {
#ifdef I2C_HOST_SIM
      (void)stream;
      for (int b = 0; b < p->B; ++b) forward_sweep_body<M, R, false, true>(c, a, b);
      return I2C_OK;
#else
      hipLaunchKernelGGL((k_forward<M, R, false, true>), dim3((p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK), dim3(SWEEP_BLOCK),
                         0, (hipStream_t)stream, c, a);
      return launch_status();
#endif
    }
#ifdef I2C_HOST_SIM
    (void)stream;
    const bool lean = c.rule_xu.unit && c.rule_x.unit && !c.z_per_cell && !a.alpha_cell && !a.prior_out;
    for (int b = 0; b < p->B; ++b) {
      if (lean)
        forward_sweep_body<M, R, true>(c, a, b);
      else
        forward_sweep_body<M, R, false>(c, a, b);
    }
    return I2C_OK;
#else
    const int lanes = sweep_lanes();
    const int grid = (p->B + lanes - 1) / lanes;
    const bool lean = c.rule_xu.unit && c.rule_x.unit && !c.z_per_cell && !a.alpha_cell && !a.prior_out;
    if (lean)
      hipLaunchKernelGGL((k_forward<M, R, true>), dim3(grid), dim3(lanes), 0, (hipStream_t)stream, c, a);
    else
      hipLaunchKernelGGL((k_forward<M, R, false>), dim3(grid), dim3(lanes), 0, (hipStream_t)stream, c, a);
    return launch_status();
#endif
  }

  // Measured on MI355X (tools/bench_models.py): below ~32k trajectories the sequential depth decides -> chunked.
  // Above, HBM traffic decides -> fused (264 instead of ~490 B/cell for the pendulum), except for the double cartpole,
  // whose fused cell (d = 7, nz = 9 with four trigonometric outputs) still spills ~400 B/lane: its chunk walk does not,
  // and stays the fastest schedule at every batch size (B = 32768: chunked 3.5, two-pass 5.8, fused 6.6 ms).
  static int schedule(int B, int T, int requested) {
    int mode = requested;
    if (mode == I2C_BWD_AUTO)
      mode = B < I2C_BWD_FUSED_MIN_B ? I2C_BWD_CHUNKED : (M::FUSED_BACKWARD_FITS ? I2C_BWD_FUSED : I2C_BWD_CHUNKED);
    if (mode == I2C_BWD_CHUNKED && T < 8) mode = I2C_BWD_TWO_PASS;  // too short to chunk
    return mode;
  }
  static int pick_mode(const I2cProblem* p) {
    int mode = schedule(p->B, p->T, p->backward_mode);
    if (mode == I2C_BWD_CHUNKED && !p->work) mode = I2C_BWD_TWO_PASS;  // no workspace
    return mode;
  }

  // `fuse` (i2c_learn only): run the M-step inside the reduction kernel of the two-pass / chunked schedules.
  struct MstepFuse {
    double tol;
    int update;
    void* stats_out;
    bool done;
  };
  static int backward(const I2cProblem* p, const void* fwd, void* xm, void* post, void* zpost, void* cell_stats,
                      void* term_stats, int32_t* status, void* stream) {
    return backward_impl(p, fwd, xm, post, zpost, cell_stats, term_stats, status, stream, nullptr);
  }
  static int backward_impl(const I2cProblem* p, const void* fwd, void* xm, void* post, void* zpost, void* cell_stats,
                           void* term_stats, int32_t* status, void* stream, MstepFuse* fuse) {
    const C c = make_consts<M, R>(p, fuse ? fuse->tol : 0.0, 0);
    MstepArgs<R> ms{(const R*)term_stats, fuse ? (R*)p->alpha : nullptr, fuse ? (R*)fuse->stats_out : nullptr,
                    fuse ? fuse->update : 0};
    if (p->inference == I2C_INF_LINEARIZE) {  // one schedule: a lane per trajectory walks T-1..0
      if (M::NZT == 0) return I2C_EINVAL;  // no terminal observation: the reference fails at i2c.py:500-501
      CellArgs<R> al{(const R*)fwd, (const R*)xm,    (const R*)p->z, (R*)post, (R*)zpost,
                     (R*)cell_stats, (R*)term_stats, (R*)p->temp,    status,   (const R*)p->alpha};
#ifdef I2C_HOST_SIM
      (void)stream;
      for (int b = 0; b < p->B; ++b) backward_lin_body<M, R>(c, al, b);
      return I2C_OK;
#else
      hipLaunchKernelGGL((k_bwd_lin<M, R>), dim3((p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK), dim3(SWEEP_BLOCK), 0,
                         (hipStream_t)stream, c, al);
      return launch_status();
#endif
    }
    if (p->inference == I2C_INF_GAUSS_HERMITE) {  // one schedule: the fused walk with the grid transform
      CellArgs<R> ag{(const R*)fwd, (const R*)xm,    (const R*)p->z, (R*)post, (R*)zpost,
                     (R*)cell_stats, (R*)term_stats, (R*)p->temp,    status,   (const R*)p->alpha};
#ifdef I2C_HOST_SIM
      (void)stream;
      for (int b = 0; b < p->B; ++b) backward_fused_body<M, R, true>(c, ag, b);
      return I2C_OK;
#else
      hipLaunchKernelGGL((k_bwd_fused<M, R, true>), dim3((p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK), dim3(SWEEP_BLOCK), 0,
                         (hipStream_t)stream, c, ag);
      return launch_status();
#endif
    }
    const int mode = pick_mode(p);
    if (mode == I2C_BWD_TWO_PASS && (!xm || !cell_stats)) return I2C_EINVAL;
    ScanArgs<R> s{(const R*)fwd, (R*)xm, (R*)p->temp, status};
    CellArgs<R> a{(const R*)fwd, (const R*)xm,   (const R*)p->z, (R*)post,  (R*)zpost,
                  (R*)cell_stats, (R*)term_stats, (R*)p->temp,    status,   (const R*)p->alpha};
    ChunkArgs<R> ch{a, nullptr, nullptr, nullptr, 0, 0};
    C cr = c;  // reduction over chunks instead of cells: same kernel, T := number of chunks
    if (mode == I2C_BWD_CHUNKED) {
      chunk_geometry(p->B, p->T, &ch.n_chunks, &ch.chunk_len);
      constexpr int NX = M::NX;
      ch.comp = (R*)p->work;
      ch.bnd = ch.comp + (size_t)ch.n_chunks * (NX + NX * NX + sym(NX)) * p->B;
      ch.part = ch.bnd + (size_t)ch.n_chunks * (NX + sym(NX)) * p->B;
      cr.T = ch.n_chunks;
    }
    CellArgs<R> ared = a;
    ared.cell_stats = ch.part;
#ifdef I2C_HOST_SIM
    (void)stream;
    auto reduce_all = [&](const C& cc, const R* stats) {  // same partition and summation order as k_reduce
      for (int b = 0; b < p->B; ++b) {
        R m = R(0), v = R(0);
        for (int q = 0; q < 8; ++q) {
          R pm, pv;
          reduce_partial<M, R>(cc, stats, b, q, 8, &pm, &pv);
          m += pm;
          v += pv;
        }
        a.term_stats[(long)p->B + b] = m;
        a.term_stats[2 * (long)p->B + b] = v;
        if (ms.alpha) mstep_body<M, R>(c, ms, b);
      }
      if (fuse) fuse->done = true;
    };
    if (mode == I2C_BWD_FUSED) {
      for (int b = 0; b < p->B; ++b) backward_fused_body<M, R>(c, a, b);
      return I2C_OK;
    }
    if (mode == I2C_BWD_CHUNKED) {
      for (int q = 0; q < ch.n_chunks; ++q)
        for (int b = 0; b < p->B; ++b) chunk_compose_body<M, R>(c, ch, q, b);
      for (int b = 0; b < p->B; ++b) chunk_stitch_body<M, R>(c, ch, b);
      for (int q = 0; q < ch.n_chunks; ++q)
        for (int b = 0; b < p->B; ++b) chunk_walk_body<M, R>(c, ch, q, b);
      reduce_all(cr, ch.part);
      return I2C_OK;
    }
    for (int b = 0; b < p->B; ++b) backward_scan_body<M, R>(c, s, b);
    for (int t = 0; t < p->T; ++t)
      for (int b = 0; b < p->B; ++b) backward_cell_body<M, R>(c, a, t, b);
    reduce_all(c, a.cell_stats);
    return I2C_OK;
#else
    const int grid = (p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
    hipStream_t st = (hipStream_t)stream;
    if (mode == I2C_BWD_FUSED) {
      hipLaunchKernelGGL((k_bwd_fused<M, R>), dim3(grid), dim3(SWEEP_BLOCK), 0, st, c, a);
      return launch_status();
    }
    if (mode == I2C_BWD_CHUNKED) {
      hipLaunchKernelGGL((k_chunk_compose<M, R>), dim3(grid, ch.n_chunks), dim3(SWEEP_BLOCK), 0, st, c, ch);
      hipLaunchKernelGGL((k_chunk_stitch<M, R>), dim3(grid), dim3(SWEEP_BLOCK), 0, st, c, ch);
      hipLaunchKernelGGL((k_chunk_walk<M, R>), dim3(grid, ch.n_chunks), dim3(SWEEP_BLOCK), 0, st, c, ch);
      hipLaunchKernelGGL((k_reduce<M, R>), dim3(grid), dim3(SWEEP_BLOCK, REDUCE_PARTS), 0, st, cr, ared, ms, p->T);
      if (fuse) fuse->done = true;
      return launch_status();
    }
    hipLaunchKernelGGL((k_scan<M, R>), dim3(grid), dim3(SWEEP_BLOCK), 0, st, c, s);
    if (launch_status() != I2C_OK) return I2C_ELAUNCH;
    const dim3 cgrid((p->B + CELL_BLOCK - 1) / CELL_BLOCK, p->T);
    hipLaunchKernelGGL((k_cell<M, R>), cgrid, dim3(CELL_BLOCK), 0, st, c, a);
    if (launch_status() != I2C_OK) return I2C_ELAUNCH;
    hipLaunchKernelGGL((k_reduce<M, R>), dim3(grid), dim3(SWEEP_BLOCK, REDUCE_PARTS), 0, st, c, a, ms, p->T);
    if (fuse) fuse->done = true;
    return launch_status();
#endif
  }

  static int riccati(const I2cProblem* p, const void* prior_out, const void* fwd, const void* xm, void* post, void* ric,
                     int32_t* status, void* stream) {
    const C c = make_consts<M, R>(p, 0.0, 0);
    RiccatiArgs<R> a{(const R*)prior_out, (const R*)fwd, (const R*)xm, (const R*)p->z, (const R*)p->alpha,
                     (R*)post,            (R*)ric,       status};
#ifdef I2C_HOST_SIM
    (void)stream;
    for (int b = 0; b < p->B; ++b) riccati_body<M, R>(c, a, b);
    return I2C_OK;
#else
    hipLaunchKernelGGL((k_riccati<M, R>), dim3((p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK), dim3(SWEEP_BLOCK), 0,
                       (hipStream_t)stream, c, a);
    return launch_status();
#endif
  }

  static int mstep(const I2cProblem* p, const void* term_stats, double tol, int update, void* stats_out,
                   void* stream) {
    const C c = make_consts<M, R>(p, tol, 0);
    MstepArgs<R> a{(const R*)term_stats, (R*)p->alpha, (R*)stats_out, update};
#ifdef I2C_HOST_SIM
    (void)stream;
    for (int b = 0; b < p->B; ++b) mstep_body<M, R>(c, a, b);
    return I2C_OK;
#else
    const int grid = (p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
    hipLaunchKernelGGL((k_mstep<M, R>), dim3(grid), dim3(SWEEP_BLOCK), 0, (hipStream_t)stream, c, a);
    return launch_status();
#endif
  }

  static int learn(const I2cProblem* p, void* post, void* fwd, void* xm, void* zpost, void* cell_stats,
                   void* term_stats, double tol, int tau, int n_iters, void* stats_hist, int32_t* status,
                   void* stream) {
    for (int it = 0; it < n_iters; ++it) {
      int rc = forward(p, post, fwd, nullptr, status, stream);
      if (rc != I2C_OK) return rc;
      MstepFuse fuse{tol, 1, (R*)stats_hist + (size_t)it * 4 * p->B, false};
      rc = backward_impl(p, fwd, xm, post, zpost, cell_stats, term_stats, status, stream, &fuse);
      if (rc != I2C_OK) return rc;
      if (!fuse.done) {  // fused / Linearize / Gauss-Hermite schedules have no reduction kernel
        rc = mstep(p, term_stats, tol, 1, (R*)stats_hist + (size_t)it * 4 * p->B, stream);
        if (rc != I2C_OK) return rc;
      }
      if (tau > 0 && it == 0) {  // _update_priors: cells with index <= tau switch to feedback mode (idempotent: once per call)
        const size_t n = (size_t)(tau + 1 < p->T ? tau + 1 : p->T);
#ifdef I2C_HOST_SIM
        std::memset(const_cast<uint8_t*>(p->feedforward), 0, n);
#else
        if (hipMemsetAsync(const_cast<uint8_t*>(p->feedforward), 0, n, (hipStream_t)stream) != hipSuccess)
          return I2C_ELAUNCH;
#endif
      }
    }
    return I2C_OK;
  }

  static int ckf(const I2cProblem* p, const double* sig_zeta, const void* y, const void* u, void* mu, void* cov,
                 int32_t* status, void* stream) {
  const Consts<M, R> c = make_consts<M, R>(p, 0.0, 0);
  R zeta[sym(M::NY)];
  for (int i = 0; i < sym(M::NY); ++i) zeta[i] = (R)sig_zeta[i];
  CkfArgs<R> a{(const R*)y, (const R*)u, (R*)mu, (R*)cov, status};
#ifdef I2C_HOST_SIM
  (void)stream;
  for (int b = 0; b < p->B; ++b) ckf_filter_body<M, R>(c, zeta, a, b);
  return I2C_OK;
#else
  ZetaArg<M, R> z;
  for (int i = 0; i < sym(M::NY); ++i) z.v[i] = zeta[i];
  const int grid = (p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
  hipLaunchKernelGGL((k_ckf<M, R>), dim3(grid), dim3(SWEEP_BLOCK), 0, (hipStream_t)stream, c, z, a);
  return launch_status();
#endif
  }

  // One control step of the MPC loop enqueued by one call (i2c/policy/mpc.py:156-182): filter, n_iter x (forward,
  // backward, _update_priors), first action, horizon shift into the second set of buffers.
  static int mpc_step(const I2cProblem* p, const I2cMpcStep* m, void* stream) {
    int rc = I2C_OK;
    if (m->do_filter) rc = ckf(p, m->sig_zeta, m->y, m->u, const_cast<void*>(p->x0), const_cast<void*>(p->sig_x0), m->status, stream);
    for (int it = 0; it < m->n_iter && rc == I2C_OK; ++it) {
      rc = forward(p, m->post, m->fwd, nullptr, m->status, stream);
      if (rc == I2C_OK) rc = backward(p, m->fwd, m->xm, m->post, m->zpost, m->cell_stats, m->term_stats, m->status, stream);
      if (rc == I2C_OK && m->tau > 0) {  // _update_priors: cells with index <= tau switch to feedback mode
        const size_t n = (size_t)(m->tau + 1 < p->T ? m->tau + 1 : p->T);
#ifdef I2C_HOST_SIM
        std::memset(const_cast<uint8_t*>(p->feedforward), 0, n);
#else
        if (hipMemsetAsync(const_cast<uint8_t*>(p->feedforward), 0, n, (hipStream_t)stream) != hipSuccess) rc = I2C_ELAUNCH;
#endif
      }
    }
    if (rc != I2C_OK) return rc;
    const C c = make_consts<M, R>(p, 0.0, 0);
    ShiftArgs<R> a{(const R*)m->post,       (R*)m->post_next, (const R*)m->cell_init, (const R*)p->alpha_cell, (R*)m->alpha_cell_next,
                   (const R*)m->alpha_init, (const R*)p->z,   (R*)m->z_next,          (const R*)m->z_new,      p->feedforward,
                   m->feedforward_next,     (R*)m->action};
#ifdef I2C_HOST_SIM
    for (int t = 0; t < p->T; ++t)
      for (int b = 0; b < p->B; ++b) mpc_shift_body<M, R>(c, a, t, b);
    return I2C_OK;
#else
    const dim3 grid((p->B + CELL_BLOCK - 1) / CELL_BLOCK, p->T);
    hipLaunchKernelGGL((k_mpc_shift<M, R>), grid, dim3(CELL_BLOCK), 0, (hipStream_t)stream, c, a);
    return launch_status();
#endif
  }

  static int rollout(const I2cProblem* p, const void* post, int n_rollouts, int policy, const void* eps_x0,
                     const void* eps_x, const void* eps_u, void* xu, void* z, void* x_final, void* z_term,
                     void* stream) {
    const C c = make_consts<M, R>(p, 0.0, 0);
    RolloutArgs<R> a{(const R*)post, (const R*)p->x0, (const R*)p->sig_x0, (const R*)eps_x0, (const R*)eps_x,
                     (const R*)eps_u, (R*)xu, (R*)z, (R*)x_final, (R*)z_term, n_rollouts, policy};
    const long N = (long)n_rollouts * p->B;
#ifdef I2C_HOST_SIM
    (void)stream;
    for (long n = 0; n < N; ++n) rollout_body<M, R>(c, a, (int)n);
    return I2C_OK;
#else
    const int grid = (int)((N + SWEEP_BLOCK - 1) / SWEEP_BLOCK);
    hipLaunchKernelGGL((k_rollout<M, R>), dim3(grid), dim3(SWEEP_BLOCK), 0, (hipStream_t)stream, c, a);
    return launch_status();
#endif
  }

  static int propagate(const I2cProblem* p, const void* post, void* prop, void* prop_stats, int use_expert,
                       int32_t* status, void* stream) {
    const C c = make_consts<M, R>(p, 0.0, use_expert);
    PropArgs<R> a{(const R*)post, (R*)prop, (R*)prop_stats, (const R*)p->x0, (const R*)p->sig_x0,
                  (const R*)p->z, p->feedforward, status};
    const bool gh = p->inference == I2C_INF_GAUSS_HERMITE;
#ifdef I2C_HOST_SIM
    (void)stream;
    for (int b = 0; b < p->B; ++b) {
      if (gh)
        propagate_body<M, R, true>(c, a, b);
      else
        propagate_body<M, R>(c, a, b);
    }
    return I2C_OK;
#else
    const int grid = (p->B + SWEEP_BLOCK - 1) / SWEEP_BLOCK;
    if (gh)
      hipLaunchKernelGGL((k_propagate<M, R, true>), dim3(grid), dim3(SWEEP_BLOCK), 0, (hipStream_t)stream, c, a);
    else
      hipLaunchKernelGGL((k_propagate<M, R>), dim3(grid), dim3(SWEEP_BLOCK), 0, (hipStream_t)stream, c, a);
    return launch_status();
#endif
  }
};

template <class M> static void fill_dims(I2cDims* d) {
  using C = Consts<M, double>;
  d->nx = M::NX;
  d->nu = M::NU;
  d->nz = M::NZ;
  d->nzt = M::NZT;
  d->e_post = C::E_POST;
  d->e_fwd = C::E_FWD;
  d->e_xm = C::E_XM;
  d->e_zpost = C::E_ZPOST;
  d->e_prop = C::E_PROP;
  d->n_params = M::NP;
  d->ny = M::NY;
}

template <class M, typename R> const ModelOps* make_ops() {
  using I = Impl<M, R>;
  static const ModelOps ops = {&I::forward, &I::backward,  &I::mstep,        &I::learn,           &I::ckf,
                               &I::rollout, &I::propagate, &I::riccati,   &I::mpc_step,        &fill_dims<M>,
                               &workspace_elems<M>, &I::schedule};
  return &ops;
}

}  // namespace i2c
