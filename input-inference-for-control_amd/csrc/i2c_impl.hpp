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
#include "i2c_group.hpp"
#include "i2c_wave.hpp"
#include "i2c_quad.hpp"
#include "i2c_linearize.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace i2c {

// One wavefront per workgroup for the sequential sweeps: at B = 4096 that is 64 workgroups on
// 64 different CUs, each wave with a CU's issue ports, L1 and scalar cache to itself.
constexpr int SWEEP_BLOCK = 64;   // a wavefront: workgroup of the group kernels and of k_reduce
// Workgroup of the one-lane-per-trajectory kernels: ONE wavefront. (Round 5 measured four -- the hardware deals the waves of one
// workgroup onto the four SIMDs of its CU, while single-wave workgroups may double up on a SIMD, see quad_waves_per_block below --
// and kept one: these kernels move a 512-byte segment of a different row with every memory instruction, and four waves behind ONE
// CU's memory pipeline cost more than a doubled SIMD: pendulum B = 16384 backward 0.19 -> 0.28 ms, planar quadrotor B = 1024
// backward 0.071 -> 0.100, covariance-control iteration 0.18 -> 0.28; no shape got faster. -DI2C_LANE_BLOCK=256 rebuilds that.)
#ifndef I2C_LANE_BLOCK
#define I2C_LANE_BLOCK 64
#endif
constexpr int LANE_BLOCK = I2C_LANE_BLOCK;
constexpr int CELL_BLOCK = 256;

// ---- the ONE place that knows how a per-lane body runs ---------------------------------------------------------------
// Device: a HIP kernel, lane index from the block / thread ids, started by launch() through hipLaunchKernelGGL.
// Host simulation (tests only): the same function with the lane indices as leading arguments, looped by launch().
// A kernel is written once:   template <...> I2C_KERNEL(BLOCK) k_name(I2C_LANE_PARAMS const C c, const A a) {
//                               const int b = I2C_LANE_X(BLOCK); ... I2C_LANE_Y ... }
#ifdef I2C_HOST_SIM
#define I2C_KERNEL(BLOCK) static void
#define I2C_LANE_PARAMS const long lane_x_, const int lane_y_,
#define I2C_LANE_X(BLOCK) ((void)lane_y_, lane_x_)
#define I2C_LANE_Y lane_y_
template <class K, class... A>
static int launch(K kernel, const long n, const int ny, const int /*block*/, void* /*stream*/, const A&... args) {
  for (int y = 0; y < ny; ++y)
    for (long x = 0; x < n; ++x) kernel(x, y, args...);
  return I2C_OK;
}
static int copy_bytes(void* dst, const void* src, size_t n, void*) {
  std::memcpy(dst, src, n);
  return I2C_OK;
}
#else
#define I2C_KERNEL(BLOCK) __global__ __launch_bounds__(BLOCK) void
#define I2C_LANE_PARAMS
#define I2C_LANE_X(BLOCK) ((long)blockIdx.x * (BLOCK) + threadIdx.x)
#define I2C_LANE_Y ((int)blockIdx.y)
// Experiment knob (not part of the ABI): I2C_SWEEP_LANES=<n<=64> launches the sequential sweeps with
// n active lanes per wavefront (more, emptier waves on more SIMDs).
static int sweep_lanes() {
  static int v = [] {
    const char* e = getenv("I2C_SWEEP_LANES");
    const int n = e ? atoi(e) : LANE_BLOCK;
    return (n >= 1 && n <= LANE_BLOCK) ? n : LANE_BLOCK;
  }();
  return v;
}
template <class K, class... A>
static int launch(K kernel, const long n, const int ny, const int block, void* stream, const A&... args) {
  hipLaunchKernelGGL(kernel, dim3((unsigned)((n + block - 1) / block), ny), dim3(block), 0, (hipStream_t)stream, args...);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
static int copy_bytes(void* dst, const void* src, size_t n, void* stream) {
  return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif

template <class M, typename R, bool LEAN, bool GRID = false, typename S = R>
I2C_KERNEL(LANE_BLOCK) k_forward(I2C_LANE_PARAMS const Consts<M, R> c, const FwdArgs<R, S> a, const int block) {
  const long b = I2C_LANE_X(block);  // `block` = active lanes per wave (I2C_SWEEP_LANES experiment), normally 64
  if (b < c.B) forward_sweep_body<M, R, LEAN, GRID, S>(c, a, (int)b);
}
template <class M, typename R> I2C_KERNEL(LANE_BLOCK) k_forward_lin(I2C_LANE_PARAMS const Consts<M, R> c, const FwdArgs<R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) forward_lin_body<M, R>(c, a, (int)b);
}
template <class M, typename R> I2C_KERNEL(LANE_BLOCK) k_bwd_lin(I2C_LANE_PARAMS const Consts<M, R> c, const CellArgs<R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) backward_lin_body<M, R>(c, a, (int)b);
}
template <class M, typename R> I2C_KERNEL(LANE_BLOCK) k_riccati(I2C_LANE_PARAMS const Consts<M, R> c, const RiccatiArgs<R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) riccati_body<M, R>(c, a, (int)b);
}
template <class M, typename R, typename S = R>
I2C_KERNEL(LANE_BLOCK) k_scan(I2C_LANE_PARAMS const Consts<M, R> c, const ScanArgs<R, S> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) backward_scan_body<M, R, S>(c, a, (int)b);
}
template <class M, typename R, typename S = R>
I2C_KERNEL(CELL_BLOCK) k_cell(I2C_LANE_PARAMS const Consts<M, R> c, const CellArgs<R, S> a) {
  const long b = I2C_LANE_X(CELL_BLOCK);
  if (b < c.B) backward_cell_body<M, R, S>(c, a, I2C_LANE_Y, (int)b);
}
template <class M, typename R, bool GRID = false, typename S = R, bool LEANW = false>
I2C_KERNEL(LANE_BLOCK) k_bwd_fused(I2C_LANE_PARAMS const Consts<M, R> c, const CellArgs<R, S> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) backward_fused_body<M, R, GRID, S, LEANW>(c, a, (int)b);
}
template <class M, typename R, typename S = R>
I2C_KERNEL(LANE_BLOCK) k_chunk_compose(I2C_LANE_PARAMS const Consts<M, R> c, const ChunkArgs<R, S> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) chunk_compose_body<M, R, S>(c, a, I2C_LANE_Y, (int)b);
}
template <class M, typename R, typename S = R, bool GRID = false>
I2C_KERNEL(LANE_BLOCK) k_chunk_stitch(I2C_LANE_PARAMS const Consts<M, R> c, const ChunkArgs<R, S> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) chunk_stitch_body<M, R, S, GRID>(c, a, (int)b);
}
#ifndef I2C_WALK_LEAN
#define I2C_WALK_LEAN 1
#endif
template <class M, typename R, typename S = R, bool LEANW = false, bool GRID = false>
I2C_KERNEL(LANE_BLOCK) k_chunk_walk(I2C_LANE_PARAMS const Consts<M, R> c, const ChunkArgs<R, S> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) chunk_walk_body<M, R, S, LEANW, GRID>(c, a, I2C_LANE_Y, (int)b);
}
template <class M, typename R> I2C_KERNEL(LANE_BLOCK) k_chunk_stitch_lin(I2C_LANE_PARAMS const Consts<M, R> c, const ChunkArgs<R, R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) chunk_stitch_lin_body<M, R>(c, a, (int)b);
}
template <class M, typename R> I2C_KERNEL(LANE_BLOCK) k_chunk_walk_lin(I2C_LANE_PARAMS const Consts<M, R> c, const ChunkArgs<R, R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) chunk_walk_lin_body<M, R>(c, a, I2C_LANE_Y, (int)b);
}
template <class M, typename R>
I2C_KERNEL(LANE_BLOCK) k_chunk_reduce_lin(I2C_LANE_PARAMS const Consts<M, R> c, const ChunkArgs<R, R> a, const MstepArgs<R> ms) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) chunk_reduce_lin_body<M, R>(c, a, ms, (int)b);
}
template <class M, typename R> I2C_KERNEL(LANE_BLOCK) k_mstep(I2C_LANE_PARAMS const Consts<M, R> c, const MstepArgs<R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) mstep_body<M, R>(c, a, (int)b);
}
template <class M, typename R, bool GRID = false>
I2C_KERNEL(LANE_BLOCK) k_propagate(I2C_LANE_PARAMS const Consts<M, R> c, const PropArgs<R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) propagate_body<M, R, GRID>(c, a, (int)b);
}
// The forward sweep of one EM iteration and the closed-loop propagation of the PREVIOUS one in ONE launch (grid row 0 / row 1).
// Both walk T dependent cells with one lane per trajectory, both only read the posterior buffer; at the batch sizes of covariance
// control (B = 8192: 128 wavefronts each) run one after the other they are two chains, in one dispatch the workgroup distributor
// deals the 256 workgroups onto 256 different CUs and they are one. (Two STREAMS do not do this: measured, the two kernels' waves
// land on the same SIMDs and the propagation takes 165 instead of 95 us -- profiles/r5_covctrl_overlap.txt.)
template <class M, typename R, bool LEAN>
I2C_KERNEL(LANE_BLOCK) k_forward_propagate(I2C_LANE_PARAMS const Consts<M, R> c, const FwdArgs<R, R> af, const PropArgs<R> ap) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b >= c.B) return;
  if (I2C_LANE_Y == 0) forward_sweep_body<M, R, LEAN, false, R>(c, af, (int)b);
  else propagate_body<M, R, false>(c, ap, (int)b);
}
template <class M, typename R> struct ZetaArg {
  R v[sym(M::NY)];
};
template <class M, typename R>
I2C_KERNEL(LANE_BLOCK) k_ckf(I2C_LANE_PARAMS const Consts<M, R> c, const ZetaArg<M, R> z, const CkfArgs<R> a) {
  const long b = I2C_LANE_X(LANE_BLOCK);
  if (b < c.B) ckf_filter_body<M, R>(c, z.v, a, (int)b);
}
template <class M, typename R> I2C_KERNEL(CELL_BLOCK) k_mpc_shift(I2C_LANE_PARAMS const Consts<M, R> c, const ShiftArgs<R> a) {
  const long b = I2C_LANE_X(CELL_BLOCK);
  if (b < c.B) mpc_shift_body<M, R>(c, a, (int)b);
}
// mode flags of cells 0 .. n-1 (ring rows t0 .. t0+n-1 mod T) to feedback: ONE launch (two hipMemsetAsync spans of odd length
// are split by the runtime into up to five fill kernels, 5 us each: a twentieth of a planar-quadrotor control step)
template <class M> I2C_KERNEL(CELL_BLOCK) k_to_feedback(I2C_LANE_PARAMS uint8_t* ff, const int t0, const int n, const int T) {
  const long i = I2C_LANE_X(CELL_BLOCK);
  if (i < n) ff[(t0 + (int)i) % T] = 0;
}
template <class M, typename R> I2C_KERNEL(LANE_BLOCK) k_rollout(I2C_LANE_PARAMS const Consts<M, R> c, const RolloutArgs<R> a) {
  const long n = I2C_LANE_X(LANE_BLOCK);
  if (n < (long)a.n_rollouts * c.B) rollout_body<M, R>(c, a, (int)n);
}

// Sum of the per-cell cost statistics over t: REDUCE_PARTS lanes per trajectory, fixed summation order; with `ms.alpha`
// set (i2c_learn) the temperature M-step rides on it. The device form exchanges the partial sums through LDS; the host
// form walks the same partition in the same order.
constexpr int REDUCE_PARTS = 8;
template <class M, typename R, class CA>
I2C_FN void reduce_finish(const Consts<M, R>& c, const CA& a, const MstepArgs<R>& ms, const int T_mstep, const int b,
                          const R m, const R v) {
  a.term_stats[(long)c.B + b] = m;
  a.term_stats[2 * (long)c.B + b] = v;
  if (ms.alpha) {
    Consts<M, R> cm = c;
    cm.T = T_mstep;  // c.T is the number of summands here (cells or chunks), the M-step needs the horizon
    mstep_body<M, R>(cm, ms, b);
  }
}
#ifdef I2C_HOST_SIM
template <class M, typename R, class CA>
static int launch_reduce(const Consts<M, R>& c, const CA& a, const MstepArgs<R>& ms, const int T_mstep, void*) {
  for (int b = 0; b < c.B; ++b) {
    R m = R(0), v = R(0);
    for (int q = 0; q < REDUCE_PARTS; ++q) {
      R pm, pv;
      reduce_partial<M, R>(c, a.cell_stats, b, q, REDUCE_PARTS, &pm, &pv);
      m += pm;
      v += pv;
    }
    reduce_finish<M, R>(c, a, ms, T_mstep, b, m, v);
  }
  return I2C_OK;
}
#else
template <class M, typename R, class CA>
__global__ __launch_bounds__(SWEEP_BLOCK* REDUCE_PARTS) void k_reduce(const Consts<M, R> c, const CA a,
                                                                      const MstepArgs<R> ms, const int T_mstep) {
  __shared__ R sm[REDUCE_PARTS][SWEEP_BLOCK], sv[REDUCE_PARTS][SWEEP_BLOCK];
  const int b = blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  R m = R(0), v = R(0);
  if (b < c.B) reduce_partial<M, R>(c, a.cell_stats, b, threadIdx.y, REDUCE_PARTS, &m, &v);
  sm[threadIdx.y][threadIdx.x] = m;
  sv[threadIdx.y][threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.y == 0 && b < c.B) {
    m = sm[0][threadIdx.x];
    v = sv[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < REDUCE_PARTS; ++q) {
      m += sm[q][threadIdx.x];
      v += sv[q][threadIdx.x];
    }
    reduce_finish<M, R>(c, a, ms, T_mstep, b, m, v);
  }
}
template <class M, typename R, class CA>
static int launch_reduce(const Consts<M, R>& c, const CA& a, const MstepArgs<R>& ms, const int T_mstep, void* stream) {
  hipLaunchKernelGGL((k_reduce<M, R, CA>), dim3((c.B + SWEEP_BLOCK - 1) / SWEEP_BLOCK), dim3(SWEEP_BLOCK, REDUCE_PARTS), 0,
                     (hipStream_t)stream, c, a, ms, T_mstep);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif

// ---- group kernels (i2c_group.hpp): G lanes per trajectory, 64 / G trajectories per wavefront ------------------------
// KIND selects the sweep; the bodies share their argument plumbing. Device: one wave per workgroup, the batch constants
// and every group's exchange region in LDS. Host simulation: the G lanes of a group are G threads.
enum { GK_FORWARD = 0, GK_BACKWARD = 1, GK_PROPAGATE = 2, GK_CKF = 3 };
// FULLW: non-diagonal cost weights (only the sweeps that price the cost, backward and propagate, have that variant)
template <int KIND, class M, typename R, int G, bool FULLW, class KC, class A>
I2C_FN void group_body(const Consts<M, R>& c, const KC& kc, const A& a, const int b, const Grp<R, G>& g) {
  if constexpr (KIND == GK_FORWARD) forward_group_body<M, R, G, FULLW>(c, kc, a, b, g);  // (FULLW: the lean variant here)
  if constexpr (KIND == GK_BACKWARD) backward_group_body<M, R, G, FULLW>(c, kc, a, b, g);
  if constexpr (KIND == GK_PROPAGATE) propagate_group_body<M, R, G, FULLW>(c, kc, a, b, g);
  if constexpr (KIND == GK_CKF) ckf_group_body<M, R, G>(c, kc, a, b, g);
}
#ifdef I2C_HOST_SIM
template <int KIND, class M, typename R, int G, bool FULLW, class A>
static int launch_group_w(const Consts<M, R>& c, const ZetaArg<M, R>* zeta, const A& a, void*) {
  GConst<M, R> kc;
  gconst_fill<M, R>(kc, &c, zeta ? zeta->v : (const R*)nullptr, 0, 1);
  for (int b = 0; b < c.B; ++b) {
    std::vector<R> sh((size_t)Grp<R, G>::SIZE, R(0));
    HostBarrier bar(G);
    std::vector<std::thread> lanes;
    for (int r = 0; r < G; ++r)
      lanes.emplace_back([&, r] { group_body<KIND, M, R, G, FULLW>(c, kc, a, b, Grp<R, G>{r, sh.data(), &bar}); });
    for (auto& th : lanes) th.join();
  }
  return I2C_OK;
}
#else
template <int KIND, class M, typename R, int G, bool FULLW, class A>
__global__ __launch_bounds__(SWEEP_BLOCK) void k_group(const Consts<M, R> c, const ZetaArg<M, R> zeta, const int has_zeta, const A a) {
  __shared__ GConst<M, R> kc;
  __shared__ R sh[(SWEEP_BLOCK / G) * Grp<R, G>::SIZE];
  {
    // The batch constants are unpacked into LDS with per-lane indices, read straight from the kernel-argument segment:
    // `c` is the first kernel parameter (offset 0), `zeta` follows it at its natural alignment.
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t zoff = (sizeof(Consts<M, R>) + alignof(ZetaArg<M, R>) - 1) / alignof(ZetaArg<M, R>) * alignof(ZetaArg<M, R>);
    gconst_fill<M, R>(kc, (const Consts<M, R>*)ka, has_zeta ? ((const ZetaArg<M, R>*)(ka + zoff))->v : (const R*)nullptr,
                      (int)threadIdx.x, SWEEP_BLOCK);
  }
  __syncthreads();
  const long lane = (long)blockIdx.x * SWEEP_BLOCK + threadIdx.x;
  const long b = lane / G;
  if (b >= c.B) return;
  const Grp<R, G> g{(int)(threadIdx.x % G), (lds_ptr<R>)(sh + (threadIdx.x / G) * Grp<R, G>::SIZE)};
  group_body<KIND, M, R, G, FULLW>(c, kc, a, (int)b, g);
}
template <int KIND, class M, typename R, int G, bool FULLW, class A>
static int launch_group_w(const Consts<M, R>& c, const ZetaArg<M, R>* zeta, const A& a, void* stream) {
  ZetaArg<M, R> z{};
  if (zeta) z = *zeta;
  const long lanes = (long)c.B * G;
  hipLaunchKernelGGL((k_group<KIND, M, R, G, FULLW, A>), dim3((unsigned)((lanes + SWEEP_BLOCK - 1) / SWEEP_BLOCK)), dim3(SWEEP_BLOCK), 0,
                     (hipStream_t)stream, c, z, zeta ? 1 : 0, a);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif

// ---- wave kernels (i2c_wave.hpp): one wavefront per trajectory, four per workgroup ----------------------------------------------
enum { WK_FORWARD = 0, WK_BACKWARD = 1, WK_SCAN = 2, WK_CELL = 3, WK_FORWARD_PL = 4 };  // _PL: pivot blocks through LDS
constexpr int WAVES_PER_BLOCK = 4;
template <int KIND, class M, typename R, typename S, bool LIN, class KC, class A>
I2C_FN void wave_body(const Consts<M, R>& c, const KC& kc, const A& a, const int t, const int b, const Wave<R>& w) {
  if constexpr (KIND == WK_FORWARD) forward_wave_body<M, R, S, LIN, false>(c, kc, a, b, w);
  if constexpr (KIND == WK_FORWARD_PL) forward_wave_body<M, R, S, LIN, true>(c, kc, a, b, w);
  if constexpr (KIND == WK_BACKWARD) backward_wave_body<M, R, S, LIN>(c, kc, a, b, w);
  if constexpr (KIND == WK_SCAN) backward_wave_scan_body<M, R, S>(c, kc, a, b, w);
  if constexpr (KIND == WK_CELL) backward_wave_cell_body<M, R, S>(c, kc, a, t, b, w);  // one wave per (t, b)
}
#ifdef I2C_HOST_SIM
template <int KIND, class M, typename R, typename S, bool LIN, class A>
static int launch_wave_v(const Consts<M, R>& c, const A& a, void*) {
  WConst<M, R> kc;
  wconst_fill<M, R>(kc, &c, 0, 1);
  for (int t = 0; t < (KIND == WK_CELL ? c.T : 1); ++t)
    for (int b = 0; b < c.B; ++b) {
      std::vector<R> sh((size_t)WaveLds::SIZE, R(0)), xch(128, R(0));
      HostBarrier bar(64);
      std::vector<std::thread> lanes;
      for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l, t, b] { wave_body<KIND, M, R, S, LIN>(c, kc, a, t, b, Wave<R>{l, l >> 4, l & 15, sh.data(), &bar, xch.data()}); });
      for (auto& th : lanes) th.join();
    }
  return I2C_OK;
}
#else
template <int KIND, class M, typename R, typename S, bool LIN, class A>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, 2) void k_wave(const Consts<M, R> c, const A a) {
  __shared__ WConst<M, R> kc;
  __shared__ R sh[WAVES_PER_BLOCK * WaveLds::SIZE];
  wconst_fill<M, R>(kc, (const Consts<M, R>*)__builtin_amdgcn_kernarg_segment_ptr(), (int)threadIdx.x, 64 * WAVES_PER_BLOCK);
  __syncthreads();
  // A wave reads ONE 8-byte element of every [B]-contiguous row: the 16 trajectories that share a 128-byte line of each row
  // are mapped onto four workgroups of the SAME XCD (workgroups are dealt round-robin over the 8 XCDs, so blocks i and i + 8
  // share an L2): every line is then fetched from HBM once per XCD instead of once per wave. Placement is a speed heuristic
  // only -- any mapping computes the same result.
  const unsigned i = blockIdx.x, x = i & 7u, r = (i >> 3) & 3u, g = i >> 5;
  const long b = 16L * (g * 8u + x) + 4 * r + (threadIdx.x >> 6);
  if (b >= c.B) return;
  const int l = (int)(threadIdx.x & 63u);
  const Wave<R> w{l, l >> 4, l & 15, (lds_ptr<R>)(sh + (threadIdx.x >> 6) * WaveLds::SIZE)};
  wave_body<KIND, M, R, S, LIN>(c, kc, a, (int)blockIdx.y, (int)b, w);
}
template <int KIND, class M, typename R, typename S, bool LIN, class A>
static int launch_wave_v(const Consts<M, R>& c, const A& a, void* stream) {
  const unsigned blocks = (unsigned)(((long)c.B + 127) / 128) * 32u;
  hipLaunchKernelGGL((k_wave<KIND, M, R, S, LIN, A>), dim3(blocks, KIND == WK_CELL ? (unsigned)c.T : 1u), dim3(64 * WAVES_PER_BLOCK), 0,
                     (hipStream_t)stream, c, a);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif

// ---- quad kernels (i2c_quad.hpp): four trajectories per wavefront, four wavefronts per workgroup ------------------------------
// Workgroup of the quad kernels: FOUR wavefronts for every model (round 5; d <= 8 had one). With single-wave workgroups the
// dispatcher dealt 1024 waves as four per CU, but inside a CU a wave could land on a SIMD that already had one while another
// stayed empty (s_getreg HW_ID of every wave, profiles/r5_wave_placement.txt: 28 - 35 of 1024 SIMDs doubled) -- depending on the
// batch size and on what ran before, the SAME forward sweep took 1.27 or 1.87 ms (double cartpole B = 4096 / 4032; cartpole
// 1.39 / 2.02): the two-waves-per-SIMD issue rate for the whole launch, which is as slow as its slowest wave. The waves of ONE
// workgroup go to the four SIMDs of its CU: 1.26 ms at every batch size <= 4096. They also cover one whole 128-byte line of every
// [B]-contiguous row between them (four trajectories = 32 bytes per wave).
#ifndef I2C_QUAD_WPB
#define I2C_QUAD_WPB 4
#endif
template <class M> constexpr int quad_waves_per_block() { return QG<M>::WIDE ? 4 : I2C_QUAD_WPB; }
#ifdef I2C_HOST_SIM
template <class M, typename R, typename S, bool GENERAL, class A>
static int launch_quad_forward_g(const Consts<M, R>& c, const A& a, void*) {
  QConst<M, R> kc;
  qconst_fill<M, R>(kc, &c, 0, 1);
  for (int b0 = 0; b0 < c.B; b0 += 4) {
    std::vector<R> sh((size_t)4 * QG<M>::SIZE, R(0)), xch(128, R(0));
    HostBarrier bar(64);
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
      lanes.emplace_back([&, l, b0] {
        const int g = (l >> 2) & 3, b = b0 + g;
        const bool live = b < c.B;
        forward_quad_body<M, R, S, GENERAL>(c, kc, a, live ? b : c.B - 1, live, Quad<R>{l, l >> 4, g, l & 3, sh.data() + g * QG<M>::SIZE, &bar, xch.data()});
      });
    for (auto& th : lanes) th.join();
  }
  return I2C_OK;
}
#else
#ifndef I2C_QF_ATTR  // (experiment knob: extra attributes of the quad forward kernel, e.g. __attribute__((amdgpu_waves_per_eu(1, 1))))
#define I2C_QF_ATTR
#endif
template <class M, typename R, typename S, class A, bool GENERAL = false>
__global__ __launch_bounds__(64 * quad_waves_per_block<M>(), 2) I2C_QF_ATTR void k_quad_forward(const Consts<M, R> c, const A a) {
  constexpr int WPB = quad_waves_per_block<M>();
  __shared__ QConst<M, R> kc;
  __shared__ R sh[WPB * 4 * QG<M>::SIZE];
  qconst_fill<M, R>(kc, (const Consts<M, R>*)__builtin_amdgcn_kernarg_segment_ptr(), (int)threadIdx.x, 64 * WPB);
  __syncthreads();
  const int l = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), g = (l >> 2) & 3;
  long b0;
  if constexpr (WPB == 1) {
    // A wave reads four consecutive trajectories (32 bytes) of every [B]-contiguous row: the four waves that share a 128-byte line
    // of each row are mapped onto workgroups of the SAME XCD (workgroups are dealt round-robin over the 8 XCDs, so blocks i and
    // i + 8 share an L2). Placement is a speed heuristic only -- any mapping computes the same result.
    const unsigned i = blockIdx.x, x = i & 7u, rr = (i >> 3) & 3u, gg = i >> 5;
    b0 = 16L * (gg * 8u + x) + 4 * rr;
  } else {  // (the waves of a workgroup take consecutive groups of four trajectories: one 128-byte line of a [B]-contiguous row)
    b0 = 4L * ((long)blockIdx.x * WPB + wv);
  }
#ifdef I2C_QUAD_PLACEMENT  // (diagnostic build, never shipped: where the dispatcher put this wave -- tools/placement_summary.py)
  if (l == 0)
    printf("placement %u %d %u %u\n", blockIdx.x, b0 < c.B ? 1 : 0, (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4),
           (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20));
#endif
  if (b0 >= c.B) return;  // (wave-uniform: no trajectory in this wave)
  const long b = b0 + g;
#ifdef I2C_QF_NOSTORE  // (experiment, never shipped: the sweep without its stores -- how much of it is the store path)
  const bool live = false;
#else
  const bool live = b < c.B;
#endif
  const Quad<R> q{l, l >> 4, g, l & 3, (lds_ptr<R>)(sh + (wv * 4 + g) * QG<M>::SIZE)};
  forward_quad_body<M, R, S, GENERAL>(c, kc, a, (int)(b < c.B ? b : c.B - 1), live, q);
}
template <class M, typename R, typename S, bool GENERAL, class A>
static int launch_quad_forward_g(const Consts<M, R>& c, const A& a, void* stream) {
  constexpr int WPB = quad_waves_per_block<M>();
  const unsigned blocks = WPB == 1 ? (unsigned)(((long)c.B + 127) / 128) * 32u : (unsigned)(((long)c.B + 4 * WPB - 1) / (4 * WPB));
  hipLaunchKernelGGL((k_quad_forward<M, R, S, A, GENERAL>), dim3(blocks), dim3(64 * WPB), 0, (hipStream_t)stream, c, a);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif
// unit weights (lam = 0, every shipped config), or the GENERAL variant where the model has it (Impl::quad_supported has checked)
template <class M, typename R, typename S, class A>
static int launch_quad_forward(const Consts<M, R>& c, const A& a, void* stream) {
  const bool unit = c.rule_xu.unit && c.rule_x.unit && c.rule_xu.w0 == R(0) && c.rule_x.w0 == R(0);
  if constexpr (quad_general_exists<M>()) {
    if (!unit) return launch_quad_forward_g<M, R, S, true>(c, a, stream);
  }
  return unit ? launch_quad_forward_g<M, R, S, false>(c, a, stream) : I2C_ENOTSUP;
}

// the quad backward sweep: d = 16 models with identity observations (backward_quad_body, trajectory-major buffers), and -- round 6 --
// every model of the d <= 8 geometry (backward_quad8_body: sigma-point observations, actions that share a block with states)
template <class M> constexpr bool quad_backward_exists() {
  if constexpr (!QG<M>::WIDE) return quad_backward8_exists<M>();
  else
    return M::NX % 4 == 0 && (M::NX + M::NU) % 4 == 0 && M::NU <= 4 && M::NZ == M::NX + M::NU && st_identity<ObsStruct<M>, M::NZ>() &&
           (M::NZT == 0 || (M::NZT == M::NX && st_identity<TermStruct<M>, M::NZT>()));
}
// LDS region of one trajectory: the staged cell block of the d = 16 form, the sigma-point geometry of the d <= 8 one
template <class M> constexpr int quad_backward_lds() { return QG<M>::WIDE ? QBG<M>::SIZE : QG<M>::SIZE; }
template <class M, typename R, typename S, bool GENERAL, bool LEANQ, class KC, class A>
I2C_FN void quad_backward_dispatch(const Consts<M, R>& c, const KC& kc, const A& a, const int b, const bool live, const Quad<R>& q) {
  if constexpr (QG<M>::WIDE) backward_quad_body<M, R, S, GENERAL>(c, kc, a, b, live, q);
  else backward_quad8_body<M, R, S, GENERAL, LEANQ>(c, kc, a, b, live, q);
}
#ifdef I2C_HOST_SIM
template <class M, typename R, typename S, bool GENERAL, bool LEANQ, class A>
static int launch_quad_backward_g(const Consts<M, R>& c, const A& a, void*) {
  QBConst<M, R> kc;
  qbconst_fill<M, R>(kc, &c, 0, 1);
  constexpr int LSZ = quad_backward_lds<M>();
  for (int b0 = 0; b0 < c.B; b0 += 4) {
    std::vector<R> sh((size_t)4 * LSZ, R(0)), xch(128, R(0));
    HostBarrier bar(64);
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
      lanes.emplace_back([&, l, b0] {
        const int g = (l >> 2) & 3, b = b0 + g;
        const bool live = b < c.B;
        quad_backward_dispatch<M, R, S, GENERAL, LEANQ>(c, kc, a, live ? b : c.B - 1, live, Quad<R>{l, l >> 4, g, l & 3, sh.data() + g * LSZ, &bar, xch.data()});
      });
    for (auto& th : lanes) th.join();
  }
  return I2C_OK;
}
#else
constexpr int QB_WAVES_PER_BLOCK = 4;
template <class M, typename R, typename S, class A, bool GENERAL = false, bool LEANQ = false>
__global__ __launch_bounds__(64 * QB_WAVES_PER_BLOCK, 2) void k_quad_backward(const Consts<M, R> c, const A a) {
  constexpr int WPB = QB_WAVES_PER_BLOCK, LSZ = quad_backward_lds<M>();
  __shared__ QBConst<M, R> kc;
  __shared__ R sh[WPB * 4 * LSZ];
  qbconst_fill<M, R>(kc, (const Consts<M, R>*)__builtin_amdgcn_kernarg_segment_ptr(), (int)threadIdx.x, 64 * WPB);
  __syncthreads();
  const int l = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), g = (l >> 2) & 3;
  // d = 16: trajectory-major buffers, nothing is shared between waves. d <= 8: the four waves of a workgroup take consecutive
  // groups of four trajectories -- together one 128-byte line of every [B]-contiguous row (as k_quad_forward)
  const long b0 = 4L * ((long)blockIdx.x * WPB + wv);
  if (b0 >= c.B) return;  // (wave-uniform: no trajectory in this wave)
  const long b = b0 + g;
  const bool live = b < c.B;
  const Quad<R> q{l, l >> 4, g, l & 3, (lds_ptr<R>)(sh + (wv * 4 + g) * LSZ)};
  quad_backward_dispatch<M, R, S, GENERAL, LEANQ>(c, kc, a, (int)(live ? b : c.B - 1), live, q);
}
template <class M, typename R, typename S, bool GENERAL, bool LEANQ, class A>
static int launch_quad_backward_g(const Consts<M, R>& c, const A& a, void* stream) {
  constexpr int WPB = QB_WAVES_PER_BLOCK;
  hipLaunchKernelGGL((k_quad_backward<M, R, S, A, GENERAL, LEANQ>), dim3((unsigned)(((long)c.B + 4 * WPB - 1) / (4 * WPB))), dim3(64 * WPB), 0, (hipStream_t)stream, c, a);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif
// unit weights (lam = 0, every shipped config), or the GENERAL variant where the model has it (Impl::quad_supported has checked);
// d <= 8: the lean variant when no optional output is asked for
template <class M, typename R, typename S, class A>
static int launch_quad_backward(const Consts<M, R>& c, const A& a, void* stream) {
  const bool unit = c.rule_xu.unit && c.rule_x.unit && c.rule_xu.w0 == R(0) && c.rule_x.w0 == R(0);
  if constexpr (QG<M>::WIDE) {
    if constexpr (quad_general_exists<M>()) {
      if (!unit) return launch_quad_backward_g<M, R, S, true, false>(c, a, stream);
    }
    return unit ? launch_quad_backward_g<M, R, S, false, false>(c, a, stream) : I2C_ENOTSUP;
  } else {
    const bool lean = !a.xm && !a.zpost && !a.cell_stats;
    if constexpr (quad_general_exists<M>()) {
      if (!unit) return lean ? launch_quad_backward_g<M, R, S, true, true>(c, a, stream) : launch_quad_backward_g<M, R, S, true, false>(c, a, stream);
    }
    if (!unit) return I2C_ENOTSUP;
    return lean ? launch_quad_backward_g<M, R, S, false, true>(c, a, stream) : launch_quad_backward_g<M, R, S, false, false>(c, a, stream);
  }
}

// the quad WALKER of the chunked schedule (d <= 8; backward_quad8_body<CHUNK>): grid = groups of four trajectories x chunks
template <class M, typename R, typename S> static QChunk<R> quad_chunk_of(const Consts<M, R>& c, const ChunkArgs<R, S>& a, const int ch) {
  const int t_lo = ch * a.chunk_len, t_hi = (t_lo + a.chunk_len < c.T) ? t_lo + a.chunk_len : c.T;
  return QChunk<R>{a.bnd, a.part, ch, t_lo, t_hi, a.comp, a.n_chunks};
}
#ifdef I2C_HOST_SIM
template <class M, typename R, typename S, bool GENERAL, bool LEANQ>
static int launch_quad_chunk_walk_g(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void*) {
  if constexpr (!QG<M>::WIDE) {
    QBConst<M, R> kc;
    qbconst_fill<M, R>(kc, &c, 0, 1);
    constexpr int LSZ = quad_backward_lds<M>();
    for (int b0 = 0; b0 < c.B; b0 += 4) {  // (the simulated wavefront of a group walks its chunks one after the other: 64 threads per group)
      std::vector<R> sh((size_t)4 * LSZ, R(0)), xch(128, R(0));
      HostBarrier bar(64);
      std::vector<std::thread> lanes;
      for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l, b0] {
          const int g = (l >> 2) & 3, b = b0 + g;
          const bool live = b < c.B;
          for (int ch = 0; ch < a.n_chunks; ++ch)
            backward_quad8_body<M, R, S, GENERAL, LEANQ, QB8_CHUNK_WALK>(c, kc, a.cell, live ? b : c.B - 1, live,
                                                               Quad<R>{l, l >> 4, g, l & 3, sh.data() + g * LSZ, &bar, xch.data()},
                                                               quad_chunk_of<M, R, S>(c, a, ch));
        });
      for (auto& th : lanes) th.join();
    }
    return I2C_OK;
  }
  return I2C_ENOTSUP;
}
// the COMPOSE and STITCH passes in the quad form (compose_quad8_body, backward_quad8_body<QB8_STITCH>)
template <class M, typename R, typename S>
static int launch_quad_chunk_compose(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void*) {
  if constexpr (!QG<M>::WIDE) {
    for (int b0 = 0; b0 < c.B; b0 += 4) {
      std::vector<R> xch(128, R(0));
      HostBarrier bar(64);
      std::vector<std::thread> lanes;
      for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l, b0] {
          const int g = (l >> 2) & 3, b = b0 + g;
          const bool live = b < c.B;
          for (int ch = 0; ch < a.n_chunks; ++ch) {
            const QChunk<R> qc = quad_chunk_of<M, R, S>(c, a, ch);
            compose_quad8_body<M, R, S>(c, a.cell.fwd, a.comp, ch, qc.t_lo, qc.t_hi, live ? b : c.B - 1, live, Quad<R>{l, l >> 4, g, l & 3, nullptr, &bar, xch.data()});
          }
        });
      for (auto& th : lanes) th.join();
    }
    return I2C_OK;
  }
  return I2C_ENOTSUP;
}
template <class M, typename R, typename S, bool GENERAL>
static int launch_quad_chunk_stitch_g(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void*) {
  if constexpr (!QG<M>::WIDE) {
    QBConst<M, R> kc;
    qbconst_fill<M, R>(kc, &c, 0, 1);
    constexpr int LSZ = quad_backward_lds<M>();
    for (int b0 = 0; b0 < c.B; b0 += 4) {
      std::vector<R> sh((size_t)4 * LSZ, R(0)), xch(128, R(0));
      HostBarrier bar(64);
      std::vector<std::thread> lanes;
      for (int l = 0; l < 64; ++l)
        lanes.emplace_back([&, l, b0] {
          const int g = (l >> 2) & 3, b = b0 + g;
          const bool live = b < c.B;
          backward_quad8_body<M, R, S, GENERAL, true, QB8_STITCH>(c, kc, a.cell, live ? b : c.B - 1, live,
                                                                  Quad<R>{l, l >> 4, g, l & 3, sh.data() + g * LSZ, &bar, xch.data()}, quad_chunk_of<M, R, S>(c, a, 0));
        });
      for (auto& th : lanes) th.join();
    }
    return I2C_OK;
  }
  return I2C_ENOTSUP;
}
#else
template <class M, typename R, typename S>
__global__ __launch_bounds__(64 * QB_WAVES_PER_BLOCK, 2) void k_quad_chunk_compose(const Consts<M, R> c, const ChunkArgs<R, S> a) {
  constexpr int WPB = QB_WAVES_PER_BLOCK;
  const int l = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), g = (l >> 2) & 3;
  const long b0 = 4L * ((long)blockIdx.x * WPB + wv);
  if (b0 >= c.B) return;  // (wave-uniform: no trajectory in this wave)
  const long b = b0 + g;
  const bool live = b < c.B;
  const Quad<R> q{l, l >> 4, g, l & 3, (lds_ptr<R>)nullptr};
  const int ch = (int)blockIdx.y, t_lo = ch * a.chunk_len, t_hi = (t_lo + a.chunk_len < c.T) ? t_lo + a.chunk_len : c.T;
  compose_quad8_body<M, R, S>(c, a.cell.fwd, a.comp, ch, t_lo, t_hi, (int)(live ? b : c.B - 1), live, q);
}
template <class M, typename R, typename S>
static int launch_quad_chunk_compose(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void* stream) {
  if constexpr (!QG<M>::WIDE) {
    constexpr int WPB = QB_WAVES_PER_BLOCK;
    hipLaunchKernelGGL((k_quad_chunk_compose<M, R, S>), dim3((unsigned)(((long)c.B + 4 * WPB - 1) / (4 * WPB)), (unsigned)a.n_chunks), dim3(64 * WPB), 0, (hipStream_t)stream, c, a);
    return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
  }
  return I2C_ENOTSUP;
}
template <class M, typename R, typename S, bool GENERAL = false>
__global__ __launch_bounds__(64 * QB_WAVES_PER_BLOCK, 2) void k_quad_chunk_stitch(const Consts<M, R> c, const ChunkArgs<R, S> a) {
  constexpr int WPB = QB_WAVES_PER_BLOCK, LSZ = quad_backward_lds<M>();
  __shared__ QBConst<M, R> kc;
  __shared__ R sh[WPB * 4 * LSZ];
  qbconst_fill<M, R>(kc, (const Consts<M, R>*)__builtin_amdgcn_kernarg_segment_ptr(), (int)threadIdx.x, 64 * WPB);
  __syncthreads();
  const int l = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), g = (l >> 2) & 3;
  const long b0 = 4L * ((long)blockIdx.x * WPB + wv);
  if (b0 >= c.B) return;  // (wave-uniform: no trajectory in this wave)
  const long b = b0 + g;
  const bool live = b < c.B;
  const Quad<R> q{l, l >> 4, g, l & 3, (lds_ptr<R>)(sh + (wv * 4 + g) * LSZ)};
  backward_quad8_body<M, R, S, GENERAL, true, QB8_STITCH>(c, kc, a.cell, (int)(live ? b : c.B - 1), live, q, QChunk<R>{a.bnd, a.part, 0, 0, c.T, a.comp, a.n_chunks});
}
template <class M, typename R, typename S, bool GENERAL>
static int launch_quad_chunk_stitch_g(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void* stream) {
  if constexpr (!QG<M>::WIDE) {
    constexpr int WPB = QB_WAVES_PER_BLOCK;
    hipLaunchKernelGGL((k_quad_chunk_stitch<M, R, S, GENERAL>), dim3((unsigned)(((long)c.B + 4 * WPB - 1) / (4 * WPB))), dim3(64 * WPB), 0, (hipStream_t)stream, c, a);
    return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
  }
  return I2C_ENOTSUP;
}
template <class M, typename R, typename S, bool GENERAL = false, bool LEANQ = false>
__global__ __launch_bounds__(64 * QB_WAVES_PER_BLOCK, 2) void k_quad_chunk_walk(const Consts<M, R> c, const ChunkArgs<R, S> a) {
  constexpr int WPB = QB_WAVES_PER_BLOCK, LSZ = quad_backward_lds<M>();
  __shared__ QBConst<M, R> kc;
  __shared__ R sh[WPB * 4 * LSZ];
  qbconst_fill<M, R>(kc, (const Consts<M, R>*)__builtin_amdgcn_kernarg_segment_ptr(), (int)threadIdx.x, 64 * WPB);
  __syncthreads();
  const int l = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), g = (l >> 2) & 3;
  const long b0 = 4L * ((long)blockIdx.x * WPB + wv);
  if (b0 >= c.B) return;  // (wave-uniform: no trajectory in this wave)
  const long b = b0 + g;
  const bool live = b < c.B;
  const Quad<R> q{l, l >> 4, g, l & 3, (lds_ptr<R>)(sh + (wv * 4 + g) * LSZ)};
  const int ch = (int)blockIdx.y, t_lo = ch * a.chunk_len, t_hi = (t_lo + a.chunk_len < c.T) ? t_lo + a.chunk_len : c.T;
  backward_quad8_body<M, R, S, GENERAL, LEANQ, QB8_CHUNK_WALK>(c, kc, a.cell, (int)(live ? b : c.B - 1), live, q, QChunk<R>{a.bnd, a.part, ch, t_lo, t_hi, a.comp, a.n_chunks});
}
template <class M, typename R, typename S, bool GENERAL, bool LEANQ>
static int launch_quad_chunk_walk_g(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void* stream) {
  if constexpr (!QG<M>::WIDE) {
    constexpr int WPB = QB_WAVES_PER_BLOCK;
    hipLaunchKernelGGL((k_quad_chunk_walk<M, R, S, GENERAL, LEANQ>), dim3((unsigned)(((long)c.B + 4 * WPB - 1) / (4 * WPB)), (unsigned)a.n_chunks), dim3(64 * WPB), 0,
                       (hipStream_t)stream, c, a);
    return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
  }
  return I2C_ENOTSUP;
}
#endif
template <class M, typename R, typename S>
static int launch_quad_chunk_stitch(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void* stream) {
  const bool unit = c.rule_xu.unit && c.rule_x.unit && c.rule_xu.w0 == R(0) && c.rule_x.w0 == R(0);
  if constexpr (quad_general_exists<M>()) {
    if (!unit) return launch_quad_chunk_stitch_g<M, R, S, true>(c, a, stream);
  }
  return unit ? launch_quad_chunk_stitch_g<M, R, S, false>(c, a, stream) : I2C_ENOTSUP;
}
template <class M, typename R, typename S>
static int launch_quad_chunk_walk(const Consts<M, R>& c, const ChunkArgs<R, S>& a, void* stream) {
  const bool unit = c.rule_xu.unit && c.rule_x.unit && c.rule_xu.w0 == R(0) && c.rule_x.w0 == R(0);
  const bool lean = !a.cell.xm && !a.cell.zpost && !a.cell.cell_stats;
  if constexpr (quad_general_exists<M>()) {
    if (!unit) return lean ? launch_quad_chunk_walk_g<M, R, S, true, true>(c, a, stream) : launch_quad_chunk_walk_g<M, R, S, true, false>(c, a, stream);
  }
  if (!unit) return I2C_ENOTSUP;
  return lean ? launch_quad_chunk_walk_g<M, R, S, false, true>(c, a, stream) : launch_quad_chunk_walk_g<M, R, S, false, false>(c, a, stream);
}

// the quad propagation (propagate_quad_body): d = 16 models with identity observations
#ifdef I2C_HOST_SIM
template <class M, typename R, bool GENERAL>
static int launch_quad_propagate_g(const Consts<M, R>& c, const PropArgs<R>& a, void*) {
  QPConst<M, R> kc;
  qpconst_fill<M, R>(kc, &c, 0, 1);
  for (int b0 = 0; b0 < c.B; b0 += 4) {
    std::vector<R> sh((size_t)4 * QG<M>::SIZE, R(0)), xch(128, R(0));
    HostBarrier bar(64);
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
      lanes.emplace_back([&, l, b0] {
        const int g = (l >> 2) & 3, b = b0 + g;
        const bool live = b < c.B;
        propagate_quad_body<M, R, GENERAL>(c, kc, a, live ? b : c.B - 1, live, Quad<R>{l, l >> 4, g, l & 3, sh.data() + g * QG<M>::SIZE, &bar, xch.data()});
      });
    for (auto& th : lanes) th.join();
  }
  return I2C_OK;
}
#else
template <class M, typename R, bool GENERAL = false>
__global__ __launch_bounds__(64 * quad_waves_per_block<M>(), 2) void k_quad_propagate(const Consts<M, R> c, const PropArgs<R> a) {
  constexpr int WPB = quad_waves_per_block<M>();
  __shared__ QPConst<M, R> kc;
  __shared__ R sh[WPB * 4 * QG<M>::SIZE];
  qpconst_fill<M, R>(kc, (const Consts<M, R>*)__builtin_amdgcn_kernarg_segment_ptr(), (int)threadIdx.x, 64 * WPB);
  __syncthreads();
  const int l = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), g = (l >> 2) & 3;
  const long b0 = 4L * ((long)blockIdx.x * WPB + wv);
  if (b0 >= c.B) return;  // (wave-uniform: no trajectory in this wave)
  const long b = b0 + g;
  const bool live = b < c.B;
  const Quad<R> q{l, l >> 4, g, l & 3, (lds_ptr<R>)(sh + (wv * 4 + g) * QG<M>::SIZE)};
  propagate_quad_body<M, R, GENERAL>(c, kc, a, (int)(live ? b : c.B - 1), live, q);
}
template <class M, typename R, bool GENERAL>
static int launch_quad_propagate_g(const Consts<M, R>& c, const PropArgs<R>& a, void* stream) {
  constexpr int WPB = quad_waves_per_block<M>();
  hipLaunchKernelGGL((k_quad_propagate<M, R, GENERAL>), dim3((unsigned)(((long)c.B + 4 * WPB - 1) / (4 * WPB))), dim3(64 * WPB), 0, (hipStream_t)stream, c, a);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif
// unit weights, or (round 6) the GENERAL variant: any CubatureQuadrature(alpha, beta, kappa)
template <class M, typename R>
static int launch_quad_propagate(const Consts<M, R>& c, const PropArgs<R>& a, void* stream) {
  const bool unit = c.rule_xu.unit && c.rule_xu.w0 == R(0);
  if constexpr (quad_general_exists<M>()) {
    if (!unit) return launch_quad_propagate_g<M, R, true>(c, a, stream);
  }
  return unit ? launch_quad_propagate_g<M, R, false>(c, a, stream) : I2C_ENOTSUP;
}

// the quad filter step (ckf_quad_body): d = 16 models
#ifdef I2C_HOST_SIM
template <class M, typename R>
static int launch_quad_ckf(const Consts<M, R>& c, const ZetaArg<M, R>& zeta, const CkfArgs<R>& a, void*) {
  QKConst<M, R> kc;
  qkconst_fill<M, R>(kc, &c, zeta.v, 0, 1);
  for (int b0 = 0; b0 < c.B; b0 += 4) {
    std::vector<R> sh((size_t)4 * QG<M>::SIZE, R(0)), xch(128, R(0));
    HostBarrier bar(64);
    std::vector<std::thread> lanes;
    for (int l = 0; l < 64; ++l)
      lanes.emplace_back([&, l, b0] {
        const int g = (l >> 2) & 3, b = b0 + g;
        const bool live = b < c.B;
        ckf_quad_body<M, R>(c, kc, a, live ? b : c.B - 1, live, Quad<R>{l, l >> 4, g, l & 3, sh.data() + g * QG<M>::SIZE, &bar, xch.data()});
      });
    for (auto& th : lanes) th.join();
  }
  return I2C_OK;
}
#else
template <class M, typename R>
__global__ __launch_bounds__(64 * quad_waves_per_block<M>(), 2) void k_quad_ckf(const Consts<M, R> c, const ZetaArg<M, R> zeta, const CkfArgs<R> a) {
  constexpr int WPB = quad_waves_per_block<M>();
  __shared__ QKConst<M, R> kc;
  __shared__ R sh[WPB * 4 * QG<M>::SIZE];
  {  // (`c` is the first kernel parameter, `zeta` follows it at its natural alignment: see k_group)
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t zoff = (sizeof(Consts<M, R>) + alignof(ZetaArg<M, R>) - 1) / alignof(ZetaArg<M, R>) * alignof(ZetaArg<M, R>);
    qkconst_fill<M, R>(kc, (const Consts<M, R>*)ka, ((const ZetaArg<M, R>*)(ka + zoff))->v, (int)threadIdx.x, 64 * WPB);
  }
  __syncthreads();
  const int l = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6), g = (l >> 2) & 3;
  const long b0 = 4L * ((long)blockIdx.x * WPB + wv);
  if (b0 >= c.B) return;  // (wave-uniform: no trajectory in this wave)
  const long b = b0 + g;
  const bool live = b < c.B;
  const Quad<R> q{l, l >> 4, g, l & 3, (lds_ptr<R>)(sh + (wv * 4 + g) * QG<M>::SIZE)};
  ckf_quad_body<M, R>(c, kc, a, (int)(live ? b : c.B - 1), live, q);
}
template <class M, typename R>
static int launch_quad_ckf(const Consts<M, R>& c, const ZetaArg<M, R>& zeta, const CkfArgs<R>& a, void* stream) {
  constexpr int WPB = quad_waves_per_block<M>();
  hipLaunchKernelGGL((k_quad_ckf<M, R>), dim3((unsigned)(((long)c.B + 4 * WPB - 1) / (4 * WPB))), dim3(64 * WPB), 0, (hipStream_t)stream, c, zeta, a);
  return hipGetLastError() == hipSuccess ? I2C_OK : I2C_ELAUNCH;
}
#endif

template <int KIND, class M, typename R, typename S, class A>
static int launch_wave(const Consts<M, R>& c, const A& a, void* stream) {
  if constexpr (KIND == WK_FORWARD) {  // batches whose waves share a SIMD: the variant with the pivot blocks through LDS
    if (c.B > 1024 && c.inference != I2C_INF_LINEARIZE) return launch_wave_v<WK_FORWARD_PL, M, R, S, false>(c, a, stream);
  }
  if constexpr (sizeof(S) == sizeof(R) && M::NZT > 0 && (KIND == WK_FORWARD || KIND == WK_BACKWARD)) {
    // the Linearize variant: fp64 storage, models with a terminal observation, one backward schedule
    if (c.inference == I2C_INF_LINEARIZE) return launch_wave_v<KIND, M, R, S, true>(c, a, stream);
  }
  return launch_wave_v<KIND, M, R, S, false>(c, a, stream);
}

template <int KIND, class M, typename R, int G, class A>
static int launch_group(const Consts<M, R>& c, const ZetaArg<M, R>* zeta, const A& a, void* stream) {
  if constexpr (KIND == GK_BACKWARD || KIND == GK_PROPAGATE) {
    const bool fullw = !c.qr_diag || (KIND == GK_BACKWARD && c.has_Qf && !c.qf_diag);
    if (fullw) return launch_group_w<KIND, M, R, G, true>(c, zeta, a, stream);
  }
  if constexpr (KIND == GK_FORWARD) {  // the compile-time lean variant (forward_group_body)
    if (!c.z_per_cell && !a.alpha_cell && !a.prior_out && c.t0 == 0) return launch_group_w<KIND, M, R, G, true>(c, zeta, a, stream);
  }
  return launch_group_w<KIND, M, R, G, false>(c, zeta, a, stream);
}

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
  c.t0 = p->t0;
  c.has_Qf = p->has_Qf && M::NZT > 0;
  c.has_x_terminal = p->has_x_terminal;
  c.z_per_cell = p->z_per_cell && p->z != nullptr;
  c.use_expert = use_expert;
  c.terminal_cell = p->terminal_cell;
  c.inference = p->inference;
  c.post_tm = (p->post_layout == 1 && M::WAVE) ? 1 : 0;
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
  int nc = (65536 + B - 1) / B;
  if (nc > 32) nc = 32;
  if (nc > T / 4) nc = T / 4;
  if (forced > 0) nc = forced < T ? forced : T;  // (the knob overrides the caps: tools/r6_chunk_sweep.sh)
  if (nc < 1) nc = 1;
  const int len = (T + nc - 1) / nc;
  *chunk_len = len;
  *n_chunks = (T + len - 1) / len;
}
template <class M> static size_t workspace_elems(int B, int T) {
  int nc, len;
  chunk_geometry(B, T, &nc, &len);
  constexpr int NX = M::NX;
  return (size_t)nc * (size_t)B * (size_t)((NX + NX * NX + sym(NX)) + (NX + sym(NX)) + 3);  // (3: the Linearize form's partial sums)
}

// batch size from which I2C_BWD_AUTO runs the fused walk: the model's own measured crossover, or the library-wide default
template <class M, class = void> struct bwd_fused_min_b {
  static constexpr int value = I2C_BWD_FUSED_MIN_B;
};
template <class M> struct bwd_fused_min_b<M, std::void_t<decltype(M::BWD_FUSED_MIN_B)>> {
  static constexpr int value = M::BWD_FUSED_MIN_B;
};

// batch window in which the d <= 8 quad backward sweep is the model's DEFAULT (measured per model, i2c_models.hpp:
// QUAD_BACKWARD8_MIN_B / _MAX_B); models without the pair: on request only
template <class M, class = void> struct quad_backward_window {
  static constexpr int min_b = 0, max_b = -1;
};
template <class M> struct quad_backward_window<M, std::void_t<decltype(M::QUAD_BACKWARD8_MAX_B)>> {
  static constexpr int min_b = M::QUAD_BACKWARD8_MIN_B, max_b = M::QUAD_BACKWARD8_MAX_B;
};

// batch window in which the chunked schedule's WALK pass runs on the quad walker by default (backward_quad8_body<CHUNK>; measured per
// model, i2c_models.hpp: QUAD_CHUNK_WALK_MIN_B / _MAX_B); models without the pair: on request only (group_lanes = 64 + "chunked")
template <class M, class = void> struct quad_chunk_walk_window {
  static constexpr int min_b = 0, max_b = -1;
};
template <class M> struct quad_chunk_walk_window<M, std::void_t<decltype(M::QUAD_CHUNK_WALK_MAX_B)>> {
  static constexpr int min_b = M::QUAD_CHUNK_WALK_MIN_B, max_b = M::QUAD_CHUNK_WALK_MAX_B;
};

// batch window in which the COMPOSE and STITCH passes of the chunked schedule run in the quad form by default (compose_quad8_body,
// backward_quad8_body<QB8_STITCH>; measured per model, i2c_models.hpp: QUAD_CHUNK_PASSES_MIN_B / _MAX_B); whatever walker follows
template <class M, class = void> struct quad_chunk_passes_window {
  static constexpr int min_b = 0, max_b = -1;
};
template <class M> struct quad_chunk_passes_window<M, std::void_t<decltype(M::QUAD_CHUNK_PASSES_MAX_B)>> {
  static constexpr int min_b = M::QUAD_CHUNK_PASSES_MIN_B, max_b = M::QUAD_CHUNK_PASSES_MAX_B;
};

// ... and the batch size up to which the STITCH pass alone stays in the quad form (i2c_models.hpp: QUAD_CHUNK_STITCH_MAX_B)
template <class M, class = void> struct quad_chunk_stitch_window {
  static constexpr int max_b = -1;
};
template <class M> struct quad_chunk_stitch_window<M, std::void_t<decltype(M::QUAD_CHUNK_STITCH_MAX_B)>> {
  static constexpr int max_b = M::QUAD_CHUNK_STITCH_MAX_B;
};

// ---- per-(model, dtype) entry points ------------------------------------------------------
// Which kernels serve a call:
//   M::GROUP      lanes per trajectory of the model's group kernels (0: none compiled); fp64 only
//   M::WAVE       the wave kernels (i2c_wave.hpp: one wavefront per trajectory) exist for this model: its default for the forward and
//                 backward sweeps wherever they apply (Impl::wave_supported); I2cProblem.group_lanes = 64 asks for them
//   M::GROUP_ONLY the one-lane-per-trajectory kernels are NOT compiled for this model (d = nx + nu > 8 does not fit one
//                 lane's registers): every call runs the wave or the group kernels
//   I2cProblem.group_lanes   0: the model's default (Impl::family); G = M::GROUP: ask for the group kernels; 64: the wave kernels;
//                 -1: one lane per trajectory; anything else: I2C_ENOTSUP
//   S             storage type of the per-cell buffers: R, or float with R = double (I2C_F64_F32S: the cubature EM path of the
//                 one-lane, the quad and the wave kernels -- forward, backward, M-step, i2c_learn; everything else is I2C_ENOTSUP)
template <class M, typename R, typename S = R> struct Impl {
  using C = Consts<M, R>;
  static constexpr bool MIXED = sizeof(S) != sizeof(R);
  static constexpr int G = M::GROUP;
  static constexpr bool HAS_GROUP = G > 0 && sizeof(R) == 8 && !MIXED;
  static constexpr bool LANE = !M::GROUP_ONLY;  // one-lane-per-trajectory kernels exist
  static constexpr bool HAS_WAVE = M::WAVE && sizeof(R) == 8;  // fp64 matrix instruction; the storage type S may be float
  static constexpr bool HAS_QUAD = M::QUAD && sizeof(R) == 8;  // fp64 matrix instruction (i2c_quad.hpp): forward sweep; the storage type S may be float
  static constexpr bool HAS_QUAD_BACKWARD = HAS_QUAD && quad_backward_exists<M>();  // ... and, for d = 16, the backward sweep

  // 1: group kernels, 0: one lane per trajectory, < 0: error code
  static int use_group(const I2cProblem* p) {
    if (p->group_lanes == 0 || (p->group_lanes == 64 && (M::WAVE || M::QUAD)) || (p->group_lanes == I2C_LANES_QUAD && M::QUAD))  // the wave / quad kernels; their missing sweeps run the default
      return M::GROUP_ONLY ? (HAS_GROUP ? 1 : I2C_ENOTSUP) : 0;
    if (p->group_lanes == -1) return M::GROUP_ONLY ? I2C_ENOTSUP : 0;  // one lane per trajectory, no hybrid forward
    return (HAS_GROUP && p->group_lanes == G) ? 1 : I2C_ENOTSUP;
  }
  // what the group form does not cover: other inference rules; per-cell blocks beyond the 2 GiB the predicated stores of
  // GIO::st_if park their masked-off lanes behind (the parked offset must stay out of the buffer window)
  // (closed-loop propagation under Linearize() IS the unit cubature rule: i2c.py:109-115)
  static int group_supported(const I2cProblem* p, const C&, const int sweep) {
    if (p->inference != I2C_INF_CUBATURE && !(p->inference == I2C_INF_LINEARIZE && sweep == I2C_SWEEP_PROPAGATE)) return I2C_ENOTSUP;
    constexpr long EMAX = C::E_FWD > C::E_POST ? (C::E_FWD > C::E_PROP ? C::E_FWD : C::E_PROP) : (C::E_POST > C::E_PROP ? C::E_POST : C::E_PROP);
    if (EMAX * (long)p->B * (long)sizeof(R) >= (1L << 31)) return I2C_EINVAL;
    return window_32bit_ok(p);
  }
  // per-cell targets [T][NZ][B] and temperatures [T][B] are addressed through 32-bit byte offsets of one buffer window by the
  // group, wave and quad kernels: beyond 4 GiB the offset would wrap and read the wrong cell (round-3 advice)
  static int window_32bit_ok(const I2cProblem* p) {
    if (p->z_per_cell && p->z && (long)p->T * C::NZ * (long)p->B * (long)sizeof(R) >= (1L << 32)) return I2C_EINVAL;
    if (p->alpha_cell && (long)p->T * (long)p->B * (long)sizeof(R) >= (1L << 32)) return I2C_EINVAL;
    return I2C_OK;
  }
  // THE place that decides which kernel family serves a sweep (i2c_kernel_family() reports it): I2C_FAMILY_* or an error code.
  //   explicit request (group_lanes = G or -1) -> that family or I2C_ENOTSUP;
  //   default: the lane kernels, except (a) models that only have group kernels, (b) the hybrid default of the d >= 7 lane
  //   models: the FORWARD sweep runs on the group kernels while the batch leaves every group wave a SIMD of its own
  //   (measured, planar quadrotor d = 8 at B = 4096: forward 0.51 -> 0.40 ms, while its chunked lane backward stays the
  //   faster one; the buffers of the families are the same, so the backward schedules are unaffected).
  // what the wave form covers: the cubature rule with lam = 0 (every shipped config: unit weights, no weight on the centre;
  // the centring of the pairwise sums relies on 2 d wi = 1), windows below 2 GiB (WIO::st_if)
  static int wave_supported(const I2cProblem* p, const C& c) {
    if (p->inference == I2C_INF_LINEARIZE) {  // Linearize(): fp64 storage; needs a terminal observation like the lane form
      if (MIXED) return I2C_ENOTSUP;
      if (M::NZT == 0) return I2C_EINVAL;
    } else if (p->inference != I2C_INF_CUBATURE) {
      return I2C_ENOTSUP;
    }  // (a terminal state prior -- covariance control -- is the backward sweep's end of the chain: w_end_of_chain, round 4)
    if (!c.rule_xu.unit || !c.rule_x.unit || c.rule_xu.w0 != R(0) || c.rule_x.w0 != R(0)) return I2C_ENOTSUP;
    constexpr long EMAX = C::E_FWD > C::E_POST ? C::E_FWD : C::E_POST;
    if (EMAX * (long)p->B * (long)sizeof(S) >= (1L << 31)) return I2C_EINVAL;
    return window_32bit_ok(p);
  }
  // what the quad form (forward sweep) covers: the cubature rule with lam = 0 (unit weights, no weight on the centre: the centring
  // of the pairwise sums relies on 2 d wi = 1, and the d = 8 models evaluate no centre point at all) for every model; any
  // CubatureQuadrature(alpha, beta, kappa) for the models with sigma-point observations and a spare pair row (quad_general_exists:
  // pendulum, cartpole, double cartpole -- the GENERAL variant, round 5); windows below 2 GiB
  static int quad_supported(const I2cProblem* p, const C& c) {
    if (p->inference != I2C_INF_CUBATURE) return I2C_ENOTSUP;
    // cubature weights with lam != 0 (round 5): the GENERAL variant, for the models that have it (quad_general_exists)
    if (!quad_general_exists<M>() && (!c.rule_xu.unit || !c.rule_x.unit || c.rule_xu.w0 != R(0) || c.rule_x.w0 != R(0))) return I2C_ENOTSUP;
    if constexpr (QG<M>::WIDE) {
      // the d = 16 form addresses trajectory-major buffers only: the posterior / prior in that layout (the engine's default for
      // the wave-capable models) and forward messages that the wave backward sweep reads
      if (p->post_layout != 1) return I2C_ENOTSUP;
      if constexpr (!HAS_WAVE) return I2C_ENOTSUP;
      // (what wave_supported checks beyond the rule -- the window sizes -- follows below; general weights are this family's alone)
    }
    constexpr long EMAX = C::E_FWD > C::E_POST ? C::E_FWD : C::E_POST;
    if (EMAX * (long)p->B * (long)sizeof(S) >= (1L << 31)) return I2C_EINVAL;
    return window_32bit_ok(p);
  }
  // d <= 8: is the quad walker the DEFAULT walk pass of this problem? (no request, the chunked schedule is the batch's default and
  // long enough to chunk, the batch inside the model's measured window)
  static bool quad_chunk_default(const I2cProblem* p) {
    if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE && LANE)
      return (p->group_lanes == 0 || p->group_lanes == I2C_LANES_QUAD) && p->backward_mode == I2C_BWD_AUTO && p->inference == I2C_INF_CUBATURE &&
             p->B >= quad_chunk_walk_window<M>::min_b && p->B <= quad_chunk_walk_window<M>::max_b && schedule(p->B, p->T, I2C_BWD_AUTO) == I2C_BWD_CHUNKED;
    return false;
  }
  // the compose / stitch passes of the chunked sigma-point schedule: I2C_FAMILY_QUAD when the quad walker was asked for by name
  // (group_lanes = 64 with "chunked": the whole schedule on matrix instructions) or, by default, inside the model's
  // quad_chunk_passes_window; I2C_FAMILY_LANE otherwise (i2c_kernel_family(problem, I2C_SWEEP_CHUNK_PASSES) reports it)
  static int chunk_passes_family(const I2cProblem* p, const C& c) {
    if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE && LANE) {
      // (the composites are addressed through 32-bit offsets of one window per chunk, masked lanes parked at 2 GiB: arithmetic-typed
      //  rows, which quad_supported's storage-typed bound does not cover under fp32 storage)
      constexpr long EC = M::NX + M::NX * M::NX + sym(M::NX);
      if (p->inference == I2C_INF_CUBATURE && quad_supported(p, c) == I2C_OK && EC * (long)p->B * (long)sizeof(R) < (1L << 31)) {
        if (p->group_lanes == 64 && p->backward_mode == I2C_BWD_CHUNKED) return I2C_FAMILY_QUAD;
        static const int forced_max = [] {  // experiment knob (not part of the ABI): overrides the model's window
          const char* e = getenv("I2C_QUAD_PASSES_MAX_B");
          return e ? atoi(e) : -2;
        }();
        const int min_b = forced_max > -2 ? 1 : quad_chunk_passes_window<M>::min_b, max_b = forced_max > -2 ? forced_max : quad_chunk_passes_window<M>::max_b;
        if ((p->group_lanes == 0 || p->group_lanes == I2C_LANES_QUAD) && p->B >= min_b && p->B <= max_b) return I2C_FAMILY_QUAD;
      }
    }
    return I2C_FAMILY_LANE;
  }
  // the STITCH pass alone in the quad form, beyond the window of the pair: the pass is a chain of NC dependent steps on B / 64 lane
  // wavefronts whatever the batch -- 0.85 us per quad step against 1.6 - 2.5 us per lane step (experiment: I2C_QUAD_STITCH_MAX_B)
  static bool quad_stitch_alone(const I2cProblem* p, const C& c) {
    if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE && LANE) {
      static const int forced_max = [] {
        const char* e = getenv("I2C_QUAD_STITCH_MAX_B");
        return e ? atoi(e) : -2;
      }();
      constexpr long EC = M::NX + M::NX * M::NX + sym(M::NX);
      const int max_b = forced_max > -2 ? forced_max : quad_chunk_stitch_window<M>::max_b;
      return p->inference == I2C_INF_CUBATURE && (p->group_lanes == 0 || p->group_lanes == I2C_LANES_QUAD) && p->B <= max_b && quad_supported(p, c) == I2C_OK &&
             EC * (long)p->B * (long)sizeof(R) < (1L << 31);
    }
    return false;
  }
  static constexpr bool HAS_QUAD_CKF = HAS_QUAD && !MIXED && quad_ckf_exists<M>();  // the filter step of the d = 16 form
  static constexpr bool HAS_QUAD_PROP = HAS_QUAD && !MIXED && quad_propagate_exists<M>();  // the closed-loop propagation of the d = 16 form
  static int family(const I2cProblem* p, const C& c, const int sweep) {
    if constexpr (HAS_QUAD_PROP) {  // the closed-loop propagation of a matrix-instruction graph: the quad form where it applies
      // (unit cubature rule -- a Linearize() graph propagates with it, i2c.py:109-115 --, trajectory-major posterior)
      if (sweep == I2C_SWEEP_PROPAGATE && (p->group_lanes == 0 || p->group_lanes == 64 || p->group_lanes == I2C_LANES_QUAD) &&
          (p->inference == I2C_INF_CUBATURE || p->inference == I2C_INF_LINEARIZE) && p->post_layout == 1 &&
          ((c.rule_xu.unit && c.rule_xu.w0 == R(0)) || quad_general_exists<M>()) && window_32bit_ok(p) == I2C_OK) {
        // (the posterior / propagation cells are addressed through 32-bit offsets of one window per cell, masked stores parked at
        //  2 GiB like the other quad forms: beyond it the offsets would wrap silently -- refused here, as group_supported does)
        constexpr long EP = C::E_POST > C::E_PROP ? C::E_POST : C::E_PROP;
        if (EP * (long)p->B * (long)sizeof(R) >= (1L << 31)) return I2C_EINVAL;
        return I2C_FAMILY_QUAD;
      }
    }
    if constexpr (HAS_QUAD_CKF) {  // the state estimator of a matrix-instruction graph (default, 64 or I2C_LANES_QUAD): the quad filter step
      if (sweep == I2C_SWEEP_FILTER && (p->group_lanes == 0 || p->group_lanes == 64 || p->group_lanes == I2C_LANES_QUAD)) return I2C_FAMILY_QUAD;
    }
    if constexpr (HAS_QUAD) {  // forward sweep: on request, or the model's default inside its batch window
      bool asked = p->group_lanes == I2C_LANES_QUAD || (p->group_lanes == 64 && !M::WAVE);
      bool sweep_ok = sweep == I2C_SWEEP_FORWARD;
      int min_b = M::QUAD_FORWARD_MIN_B, max_b = M::QUAD_FORWARD_MAX_B;
      if (sweep == I2C_SWEEP_BACKWARD && HAS_QUAD_BACKWARD) {
        if constexpr (QG<M>::WIDE) {
          // (the backward sweep of the d = 16 form: the fused walk, with the forward sweep -- an explicit two-pass request keeps the wave form)
          sweep_ok = p->backward_mode != I2C_BWD_TWO_PASS;
          min_b = M::QUAD_BACKWARD_MIN_B > M::QUAD_FORWARD_MIN_B ? M::QUAD_BACKWARD_MIN_B : M::QUAD_FORWARD_MIN_B;
        } else {
          // d <= 8 (round 6), two forms of backward_quad8_body. (i) The fused walk of four trajectories per wavefront, ONE pass over the
          // forward messages: on request (group_lanes = 64 with the schedule left open or "fused"), or inside quad_backward_window
          // (no in-tree model has one). (ii) The WALKER of the chunked schedule (compose / stitch / reduce stay lane kernels): on
          // request (group_lanes = 64 with "chunked"), or the default inside the model's quad_chunk_walk_window where the chunked
          // schedule is the batch's default (quad_chunk_default). An explicit "two_pass" is the lane kernels'.
          // I2C_LANES_QUAD asks for the quad FORWARD sweep only: its backward sweep resolves as the default does (below).
          asked = p->group_lanes == 64;
          sweep_ok = asked ? p->backward_mode != I2C_BWD_TWO_PASS : p->backward_mode == I2C_BWD_AUTO;
          min_b = quad_backward_window<M>::min_b, max_b = quad_backward_window<M>::max_b;
          if (!asked && quad_chunk_default(p)) min_b = quad_chunk_walk_window<M>::min_b, max_b = quad_chunk_walk_window<M>::max_b;
        }
      }
      // (d = 16 with general cubature weights: the wave kernels only have the unit rule -- the quad kernels at every batch size)
      const bool general_wide = QG<M>::WIDE && quad_general_exists<M>() && (!c.rule_xu.unit || !c.rule_x.unit || c.rule_xu.w0 != R(0) || c.rule_x.w0 != R(0));
      if (sweep_ok && (asked || (p->group_lanes == 0 && ((p->B >= min_b && p->B <= max_b) || general_wide)))) {
        const int rc = quad_supported(p, c);
        if (rc == I2C_OK) return I2C_FAMILY_QUAD;
        if (asked) return rc;
      }
    }
    if (p->group_lanes == I2C_LANES_QUAD) {  // the other sweeps of an explicit quad request: the model's default family
      I2cProblem q = *p;
      q.group_lanes = 0;
      return family(&q, c, sweep);
    }
    if constexpr (HAS_WAVE) {  // forward and backward sweeps: on request (group_lanes = 64) or as the model's default
      if ((sweep == I2C_SWEEP_FORWARD || sweep == I2C_SWEEP_BACKWARD) &&
          (p->group_lanes == 64 || p->group_lanes == 0)) {
        const int rc = wave_supported(p, c);
        if (rc == I2C_OK) return I2C_FAMILY_WAVE;
        if (p->group_lanes == 64 || MIXED) return rc;
      }
    }
    if constexpr (MIXED) {  // fp64 arithmetic on fp32-stored messages: the cubature EM path (one-lane, quad and wave kernels)
      if (p->inference != I2C_INF_CUBATURE || use_group(p) != 0) return I2C_ENOTSUP;
      if (sweep != I2C_SWEEP_FORWARD && sweep != I2C_SWEEP_BACKWARD) return I2C_ENOTSUP;
      return LANE ? I2C_FAMILY_LANE : I2C_ENOTSUP;
    }
    int grp = use_group(p);
    if (grp < 0) return grp;
    if constexpr (HAS_GROUP && M::GROUP_FORWARD_AUTO) {
      if (sweep == I2C_SWEEP_FORWARD && grp == 0 && p->group_lanes == 0 && (long)p->B * G <= I2C_GROUP_FORWARD_MAX_LANES &&
          group_supported(p, c, sweep) == I2C_OK)
        grp = 1;
    }
    if (grp) {
      if constexpr (HAS_GROUP) {
        const int rc = group_supported(p, c, sweep);
        return rc != I2C_OK ? rc : I2C_FAMILY_GROUP;
      }
      return I2C_ENOTSUP;
    }
    return LANE ? I2C_FAMILY_LANE : I2C_ENOTSUP;
  }
  // The state estimator's rule is fixed, whatever the graph infers with: CubatureQuadrature(1, 0, 0) (mpc.py:121-123)
  static I2cProblem filter_problem(const I2cProblem* p) {
    I2cProblem q = *p;
    q.inference = I2C_INF_CUBATURE;
    q.quad_alpha = 1.0, q.quad_beta = 0.0, q.quad_kappa = 0.0;
    return q;
  }
  static int family_of(const I2cProblem* p, int sweep) {
    I2cProblem q = sweep == I2C_SWEEP_FILTER ? filter_problem(p) : *p;
    if (sweep == I2C_SWEEP_PROPAGATE && p->inference == I2C_INF_LINEARIZE) q.quad_alpha = 1.0, q.quad_beta = 0.0, q.quad_kappa = 0.0;  // (as propagate())
    const C c = make_consts<M, R>(&q, 0.0, 0);
    if (sweep == I2C_SWEEP_CHUNK_PASSES) {  // the compose / stitch passes: of a problem whose backward sweep runs the chunked sigma-point schedule
      const int mode = plan(&q);
      if (mode < 0) return mode;
      if (mode != I2C_BWD_CHUNKED || q.inference != I2C_INF_CUBATURE) return I2C_ENOTSUP;
      return chunk_passes_family(&q, c);
    }
    if (sweep == I2C_SWEEP_CHUNK_STITCH) {  // the stitch pass of that schedule: with the compose pass, or alone inside its own window
      const int mode = plan(&q);
      if (mode < 0) return mode;
      if (mode != I2C_BWD_CHUNKED || q.inference != I2C_INF_CUBATURE) return I2C_ENOTSUP;
      return (chunk_passes_family(&q, c) == I2C_FAMILY_QUAD || quad_stitch_alone(&q, c)) ? I2C_FAMILY_QUAD : I2C_FAMILY_LANE;
    }
    return family(&q, c, sweep);
  }

  // the sigma-point forward sweep of the one-lane kernels, for either storage type
  static int forward_lane(const I2cProblem* p, const C& c, const FwdArgs<R, S>& a, void* stream) {
    if constexpr (LANE) {
#ifdef I2C_HOST_SIM
      const int lanes = LANE_BLOCK;
#else
      const int lanes = sweep_lanes();
#endif
      const bool lean = c.rule_xu.unit && c.rule_x.unit && !c.z_per_cell && !a.alpha_cell && !a.prior_out && c.t0 == 0;
      if (lean) return launch(k_forward<M, R, true, false, S>, p->B, 1, lanes, stream, c, a, lanes);
      return launch(k_forward<M, R, false, false, S>, p->B, 1, lanes, stream, c, a, lanes);
    }
    return I2C_ENOTSUP;
  }

  static int forward(const I2cProblem* p, const void* prior, void* fwd, void* prior_out, int32_t* status,
                     void* stream) {
    const C c = make_consts<M, R>(p, 0.0, p->inference == I2C_INF_LINEARIZE ? p->expert_controller : 0);
    if constexpr (MIXED) {  // fp64 arithmetic on fp32-stored messages: the cubature path (one-lane, quad and wave kernels)
      const int fam = family(p, c, I2C_SWEEP_FORWARD);
      if (fam < 0) return fam;
      FwdArgs<R, S> am{(const S*)prior, (S*)fwd, (S*)prior_out, (const R*)p->x0, (const R*)p->sig_x0,
                       (const R*)p->z,  (const R*)p->alpha, (const R*)p->alpha_cell, p->feedforward, status, p->expert};
      if (fam == I2C_FAMILY_WAVE) {
        if constexpr (HAS_WAVE) return launch_wave<WK_FORWARD, M, R, S>(c, am, stream);
      }
      if (fam == I2C_FAMILY_QUAD) {
        if constexpr (HAS_QUAD) return launch_quad_forward<M, R, S>(c, am, stream);
      }
      return forward_lane(p, c, am, stream);
    } else {
      return forward_any(p, c, prior, fwd, prior_out, status, stream);
    }
  }
  static int forward_any(const I2cProblem* p, const C& c, const void* prior, void* fwd, void* prior_out, int32_t* status,
                         void* stream) {
    FwdArgs<R> a{(const R*)prior, (R*)fwd, (R*)prior_out, (const R*)p->x0, (const R*)p->sig_x0,
                 (const R*)p->z,  (const R*)p->alpha, (const R*)p->alpha_cell, p->feedforward, status, p->expert};
    const int fam = family(p, c, I2C_SWEEP_FORWARD);
    if (fam < 0) return fam;
    if (fam == I2C_FAMILY_WAVE) {
      if constexpr (HAS_WAVE && !MIXED) return launch_wave<WK_FORWARD, M, R, R>(c, a, stream);
    }
    if (fam == I2C_FAMILY_QUAD) {
      if constexpr (HAS_QUAD) {  // the forward messages go where the backward family of this problem reads them
        C cq = c;
        const int fb = family(p, c, I2C_SWEEP_BACKWARD);
        cq.fwd_tm = ((M::WAVE && fb == I2C_FAMILY_WAVE) || (HAS_QUAD_BACKWARD && fb == I2C_FAMILY_QUAD)) ? 1 : 0;
        return launch_quad_forward<M, R, R>(cq, a, stream);
      }
    }
    if (fam == I2C_FAMILY_GROUP) {
      if constexpr (HAS_GROUP) return launch_group<GK_FORWARD, M, R, G>(c, nullptr, a, stream);
    }
    if constexpr (LANE) {
      if (p->inference == I2C_INF_LINEARIZE) return launch(k_forward_lin<M, R>, p->B, 1, LANE_BLOCK, stream, c, a);
      if (p->inference == I2C_INF_GAUSS_HERMITE)
        return launch(k_forward<M, R, false, true>, p->B, 1, LANE_BLOCK, stream, c, a, LANE_BLOCK);
      if constexpr (!MIXED) return forward_lane(p, c, a, stream);
    }
    return I2C_ENOTSUP;
  }

  // Small batches: the sequential depth decides -> chunked. Large ones: HBM traffic decides -> fused (the chunked form moves
  // compose + stitch + walk = 1.44x the walk's bytes, PMC: pendulum 381 against 264.5 B per cell, double cartpole 1 914 against
  // 1 177). The crossover is PER MODEL (M::BWD_FUSED_MIN_B, i2c_models.hpp), re-derived in round 5 from time AND traffic at
  // B = 8192 .. 32768, beyond the 256 MB Infinity Cache (profiles/r5_backward_crossover.txt): pendulum 8192: chunked 0.111 /
  // fused 0.175 ms, 16384: 0.237 / 0.193 (cartpole 0.72 / 0.99, 1.38 / 1.01; planar quadrotor 0.21 / 0.27, 0.38 / 0.30); double
  // cartpole 16384: 1.67 / 1.86, 24576: 3.44 / 2.20. Models that only have group
  // kernels run the fused walk.
  static int schedule(int B, int T, int requested) {
    if (M::WAVE && M::GROUP_ONLY) {
      // wave kernels: the fused walk, or on request the two-pass schedule (scan + one wave per (t, b) cell). Measured on MI355X
      // (12-state quadrotor, T = 50): the two-pass form is SLOWER at every batch -- B = 256: 0.179 against 0.162 ms, B = 1024:
      // 0.347 against 0.184 ms -- because with T times as many waves in flight the sweep is bound by the vector-memory
      // pipeline: a wave's load touches one 8-byte element in each of 64 different [B]-contiguous rows (64 cache lines per
      // instruction), which a lone wave per SIMD hides behind its dependent arithmetic and 50 waves per SIMD do not.
      return requested == I2C_BWD_TWO_PASS ? I2C_BWD_TWO_PASS : I2C_BWD_FUSED;
    }
    if (M::GROUP_ONLY) return I2C_BWD_FUSED;
    int mode = requested;
    if (mode == I2C_BWD_AUTO)
      mode = B < bwd_fused_min_b<M>::value ? I2C_BWD_CHUNKED : I2C_BWD_FUSED;
    if (mode == I2C_BWD_CHUNKED && T < 8) mode = I2C_BWD_TWO_PASS;  // too short to chunk
    return mode;
  }
  // THE place that decides which backward schedule runs for a problem (i2c_backward_schedule() reports it): the family that
  // serves the sweep, the inference rule, the storage type, then the batch-size rule of the lane kernels. An error code when
  // the sweep would refuse the problem. (What it assumes: the workspaces of its answer are supplied -- pick_mode.)
  static int plan(const I2cProblem* p) {
    const C c = make_consts<M, R>(p, 0.0, 0);
    const int fam = family(p, c, I2C_SWEEP_BACKWARD);
    if (fam < 0) return fam;
    const int lane_rule = schedule(p->B, p->T, p->backward_mode);
    if (fam == I2C_FAMILY_WAVE)  // the fused walk; the two-pass form on request (cubature rule)
      return (lane_rule == I2C_BWD_TWO_PASS && p->inference == I2C_INF_CUBATURE) ? I2C_BWD_TWO_PASS : I2C_BWD_FUSED;
    if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE && LANE) {
      // d <= 8 quad: the fused walk, or the chunked schedule with the quad walker (compose / stitch / reduce are the lane kernels) --
      // when asked for by name, or as the model's default inside its quad_chunk_walk_window
      if (fam == I2C_FAMILY_QUAD && p->T >= 8 && (p->backward_mode == I2C_BWD_CHUNKED || quad_chunk_default(p))) return I2C_BWD_CHUNKED;
    }
    if (fam == I2C_FAMILY_QUAD || fam == I2C_FAMILY_GROUP) return I2C_BWD_FUSED;  // four trajectories / a group of lanes walk T-1..0
    if (!LANE) return I2C_ENOTSUP;
    if (p->inference == I2C_INF_LINEARIZE) {
      if (M::NZT == 0) return I2C_EINVAL;  // no terminal observation: the reference fails at i2c.py:500-501
      return (!MIXED && lane_rule == I2C_BWD_CHUNKED) ? I2C_BWD_CHUNKED : I2C_BWD_FUSED;
    }
    if (p->inference == I2C_INF_GAUSS_HERMITE) return (!MIXED && lane_rule == I2C_BWD_CHUNKED) ? I2C_BWD_CHUNKED : I2C_BWD_FUSED;
    return lane_rule;
  }
  static int pick_mode(const I2cProblem* p) {
    int mode = plan(p);
    if (mode == I2C_BWD_CHUNKED && !p->work) {  // no workspace
      const C c = make_consts<M, R>(p, 0.0, 0);
      mode = (p->inference == I2C_INF_CUBATURE && family(p, c, I2C_SWEEP_BACKWARD) != I2C_FAMILY_QUAD) ? I2C_BWD_TWO_PASS : I2C_BWD_FUSED;
    }
    return mode;
  }

  // `fuse` (i2c_learn only): run the M-step inside the reduction kernel of the two-pass / chunked schedules.
  struct MstepFuse {
    double tol;
    int update;
    void* stats_out;
    bool done;
  };
  // Linearize() applies the terminal cost at the END of the chain, with the temperature of the cell that sits there
  // (i2c.py:475-491: the cell's own sig_xi_terminal). A receding-horizon loop appends cells that keep the temperature they were
  // copied with (I2cProblem.alpha_cell), so after the first shift that is not the graph's alpha.
  static const R* terminal_alpha(const I2cProblem* p, const C& c) {
    return p->alpha_cell ? (const R*)p->alpha_cell + (long)c.row(p->T - 1) * (long)p->B : (const R*)p->alpha;
  }
  static int backward(const I2cProblem* p, const void* fwd, void* xm, void* post, void* zpost, void* cell_stats,
                      void* term_stats, int32_t* status, void* stream) {
    return backward_impl(p, fwd, xm, post, zpost, cell_stats, term_stats, status, stream, nullptr);
  }
  static int backward_impl(const I2cProblem* p, const void* fwd, void* xm, void* post, void* zpost, void* cell_stats,
                           void* term_stats, int32_t* status, void* stream, MstepFuse* fuse) {
    const C c = make_consts<M, R>(p, fuse ? fuse->tol : 0.0, 0);
    MstepArgs<R> ms{(const R*)term_stats, fuse ? (R*)p->alpha : nullptr, fuse ? (R*)fuse->stats_out : nullptr,
                    fuse ? fuse->update : 0};
    if constexpr (MIXED) {
      const int fam = family(p, c, I2C_SWEEP_BACKWARD);
      if (fam < 0) return fam;
      CellArgs<R, S> am{(const S*)fwd, (const S*)xm,   (const R*)p->z, (S*)post,  (S*)zpost,
                        (R*)cell_stats, (R*)term_stats, (R*)p->temp,    status,   terminal_alpha(p, c)};
      if (fam == I2C_FAMILY_WAVE) return backward_wave(p, c, am, ms, fuse, stream);
      if (fam == I2C_FAMILY_QUAD) {
        if constexpr (HAS_QUAD_BACKWARD) {
          if constexpr (!QG<M>::WIDE && LANE) {
            if (pick_mode(p) == I2C_BWD_CHUNKED) return backward_chunked(p, c, am, ms, fuse, stream, true);
          }
          return launch_quad_backward<M, R, S>(c, am, stream);
        }
      }
      return backward_lane(p, c, am, ms, fuse, stream);
    } else {
      return backward_any(p, c, ms, fwd, xm, post, zpost, cell_stats, term_stats, status, stream, fuse);
    }
  }
  // the wave family's backward sweep: two-pass (scan, one wave per cell, reduction -- with the M-step riding on it in
  // i2c_learn) when that schedule is asked for / the default and its workspaces exist; the fused walk otherwise
  template <class CA>
  static int backward_wave(const I2cProblem* p, const C& c, const CA& a, const MstepArgs<R>& ms, MstepFuse* fuse, void* stream) {
    if constexpr (HAS_WAVE) {
      const bool two_pass = plan(p) == I2C_BWD_TWO_PASS && a.xm && a.cell_stats;
      if (!two_pass) return launch_wave<WK_BACKWARD, M, R, S>(c, a, stream);
      int rc = launch_wave<WK_SCAN, M, R, S>(c, a, stream);
      if (rc == I2C_OK) rc = launch_wave<WK_CELL, M, R, S>(c, a, stream);
      if (rc == I2C_OK) rc = launch_reduce<M, R>(c, a, ms, p->T, stream);
      if (fuse) fuse->done = true;
      return rc;
    }
    return I2C_ENOTSUP;
  }
  static int backward_any(const I2cProblem* p, const C& c, const MstepArgs<R>& ms, const void* fwd, void* xm, void* post,
                          void* zpost, void* cell_stats, void* term_stats, int32_t* status, void* stream, MstepFuse* fuse) {
    CellArgs<R> a{(const R*)fwd, (const R*)xm,   (const R*)p->z, (R*)post,  (R*)zpost,
                  (R*)cell_stats, (R*)term_stats, (R*)p->temp,    status,   terminal_alpha(p, c)};
    const int fam = family(p, c, I2C_SWEEP_BACKWARD);
    if (fam < 0) return fam;
    if (fam == I2C_FAMILY_WAVE) {
      if constexpr (!MIXED) return backward_wave(p, c, a, ms, fuse, stream);
    }
    if (fam == I2C_FAMILY_QUAD) {
      if constexpr (HAS_QUAD_BACKWARD) {
        if constexpr (!QG<M>::WIDE && LANE && !MIXED) {
          if (pick_mode(p) == I2C_BWD_CHUNKED) return backward_chunked(p, c, a, ms, fuse, stream, true);
        }
        return launch_quad_backward<M, R, R>(c, a, stream);
      }
    }
    if (fam == I2C_FAMILY_GROUP) {  // one schedule: the group walks T-1..0 (the fused form); backward_mode is ignored
      if constexpr (HAS_GROUP) return launch_group<GK_BACKWARD, M, R, G>(c, nullptr, a, stream);
    }
    if constexpr (LANE) {
      if (p->inference == I2C_INF_LINEARIZE) {  // a lane per trajectory walks T-1..0, or (small batches) the chunked form
        if (M::NZT == 0) return I2C_EINVAL;     // no terminal observation: the reference fails at i2c.py:500-501
        if constexpr (!MIXED) {
          if (pick_mode(p) == I2C_BWD_CHUNKED) {  // (pick_mode: asked for or the default below I2C_BWD_FUSED_MIN_B, with a workspace)
            ChunkArgs<R, R> ch{a, nullptr, nullptr, nullptr, 0, 0};
            chunk_geometry(p->B, p->T, &ch.n_chunks, &ch.chunk_len);
            constexpr int NX = M::NX;
            ch.comp = (R*)p->work;
            ch.bnd = ch.comp + (size_t)ch.n_chunks * (NX + NX * NX + sym(NX)) * p->B;
            ch.part = ch.bnd + (size_t)ch.n_chunks * (NX + sym(NX)) * p->B;
            int rc = launch(k_chunk_compose<M, R, R>, p->B, ch.n_chunks, LANE_BLOCK, stream, c, ch);
            if (rc == I2C_OK) rc = launch(k_chunk_stitch_lin<M, R>, p->B, 1, LANE_BLOCK, stream, c, ch);
            if (rc == I2C_OK) rc = launch(k_chunk_walk_lin<M, R>, p->B, ch.n_chunks, LANE_BLOCK, stream, c, ch);
            if (rc == I2C_OK) rc = launch(k_chunk_reduce_lin<M, R>, p->B, 1, LANE_BLOCK, stream, c, ch, ms);
            if (fuse) fuse->done = true;
            return rc;
          }
        }
        return launch(k_bwd_lin<M, R>, p->B, 1, LANE_BLOCK, stream, c, a);
      }
      if (p->inference == I2C_INF_GAUSS_HERMITE) {  // the fused walk with the grid transform, or (small batches) the chunked form:
        if constexpr (!MIXED) {                     // the composition of the x-marginal recursion has no transform in it
          if (pick_mode(p) == I2C_BWD_CHUNKED) {
            ChunkArgs<R, R> ch{a, nullptr, nullptr, nullptr, 0, 0};
            chunk_geometry(p->B, p->T, &ch.n_chunks, &ch.chunk_len);
            constexpr int NX = M::NX;
            ch.comp = (R*)p->work;
            ch.bnd = ch.comp + (size_t)ch.n_chunks * (NX + NX * NX + sym(NX)) * p->B;
            ch.part = ch.bnd + (size_t)ch.n_chunks * (NX + sym(NX)) * p->B;
            C cr = c;  // reduction over chunks instead of cells: same kernel, T := number of chunks
            cr.T = ch.n_chunks;
            CellArgs<R> ared = a;
            ared.cell_stats = ch.part;
            int rc = launch(k_chunk_compose<M, R, R>, p->B, ch.n_chunks, LANE_BLOCK, stream, c, ch);
            if (rc == I2C_OK) rc = launch(k_chunk_stitch<M, R, R, true>, p->B, 1, LANE_BLOCK, stream, c, ch);
            if (rc == I2C_OK) rc = launch(k_chunk_walk<M, R, R, false, true>, p->B, ch.n_chunks, LANE_BLOCK, stream, c, ch);
            if (rc == I2C_OK) rc = launch_reduce<M, R>(cr, ared, ms, p->T, stream);
            if (fuse) fuse->done = true;
            return rc;
          }
        }
        return launch(k_bwd_fused<M, R, true>, p->B, 1, LANE_BLOCK, stream, c, a);
      }
      if constexpr (!MIXED) return backward_lane(p, c, a, ms, fuse, stream);
    }
    return I2C_ENOTSUP;
  }
  // the chunked schedule of the sigma-point backward sweep: compose (one lane per trajectory and chunk) + stitch + walk + reduce; the
  // walk pass on the lane walker, or -- quad_walk, d <= 8 -- on the quad walker (four trajectories per wavefront and chunk)
  static int backward_chunked(const I2cProblem* p, const C& c, const CellArgs<R, S>& a, const MstepArgs<R>& ms, MstepFuse* fuse, void* stream,
                              const bool quad_walk) {
    if constexpr (LANE) {
      ChunkArgs<R, S> ch{a, nullptr, nullptr, nullptr, 0, 0};
      chunk_geometry(p->B, p->T, &ch.n_chunks, &ch.chunk_len);
      constexpr int NX = M::NX;
      ch.comp = (R*)p->work;
      ch.bnd = ch.comp + (size_t)ch.n_chunks * (NX + NX * NX + sym(NX)) * p->B;
      ch.part = ch.bnd + (size_t)ch.n_chunks * (NX + sym(NX)) * p->B;
      C cr = c;  // reduction over chunks instead of cells: same kernel, T := number of chunks
      cr.T = ch.n_chunks;
      CellArgs<R, S> ared = a;
      ared.cell_stats = ch.part;
      int rc;
      bool quad_compose = false, quad_stitch = false;
      if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE) {
        quad_compose = chunk_passes_family(p, c) == I2C_FAMILY_QUAD;  // compose + stitch, four trajectories per wavefront
        quad_stitch = quad_compose || quad_stitch_alone(p, c);       // ... or the stitch pass alone (a chain of NC steps whatever the batch)
      }
      if (quad_compose) {
        if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE) rc = launch_quad_chunk_compose<M, R, S>(c, ch, stream);
        else rc = I2C_ENOTSUP;
      } else {
        rc = launch(k_chunk_compose<M, R, S>, p->B, ch.n_chunks, LANE_BLOCK, stream, c, ch);
      }
      if (rc == I2C_OK) {
        if (quad_stitch) {
          if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE) rc = launch_quad_chunk_stitch<M, R, S>(c, ch, stream);
          else rc = I2C_ENOTSUP;
        } else {
          rc = launch(k_chunk_stitch<M, R, S>, p->B, 1, LANE_BLOCK, stream, c, ch);
        }
      }
      if (rc == I2C_OK) {
        if (quad_walk) {
          if constexpr (HAS_QUAD_BACKWARD && !QG<M>::WIDE) rc = launch_quad_chunk_walk<M, R, S>(c, ch, stream);
          else rc = I2C_ENOTSUP;
        } else {
          const bool lean = I2C_WALK_LEAN && !a.xm && !a.zpost && !a.cell_stats && !c.z_per_cell;  // see chunk_walk_body
          rc = lean ? launch(k_chunk_walk<M, R, S, true>, p->B, ch.n_chunks, LANE_BLOCK, stream, c, ch)
                    : launch(k_chunk_walk<M, R, S>, p->B, ch.n_chunks, LANE_BLOCK, stream, c, ch);
        }
      }
      if (rc == I2C_OK) rc = launch_reduce<M, R>(cr, ared, ms, p->T, stream);
      if (fuse) fuse->done = true;
      return rc;
    }
    return I2C_ENOTSUP;
  }
  // the three schedules of the sigma-point backward sweep of the one-lane kernels, for either storage type
  static int backward_lane(const I2cProblem* p, const C& c, const CellArgs<R, S>& a, const MstepArgs<R>& ms, MstepFuse* fuse,
                           void* stream) {
    if constexpr (LANE) {
      const int mode = pick_mode(p);
      if (mode == I2C_BWD_FUSED) {
        const bool lean = I2C_WALK_LEAN && !a.xm && !a.zpost && !a.cell_stats && !c.z_per_cell;  // see chunk_walk_body
        return lean ? launch(k_bwd_fused<M, R, false, S, true>, p->B, 1, LANE_BLOCK, stream, c, a)
                    : launch(k_bwd_fused<M, R, false, S>, p->B, 1, LANE_BLOCK, stream, c, a);
      }
      if (mode == I2C_BWD_CHUNKED) return backward_chunked(p, c, a, ms, fuse, stream, false);
      if (!a.xm || !a.cell_stats) return I2C_EINVAL;  // two-pass needs both as workspace
      ScanArgs<R, S> sc{a.fwd, const_cast<S*>(a.xm), (R*)p->temp, a.status};
      int rc = launch(k_scan<M, R, S>, p->B, 1, LANE_BLOCK, stream, c, sc);
      if (rc == I2C_OK) rc = launch(k_cell<M, R, S>, p->B, p->T, CELL_BLOCK, stream, c, a);
      if (rc == I2C_OK) rc = launch_reduce<M, R>(c, a, ms, p->T, stream);
      if (fuse) fuse->done = true;
      return rc;
    }
    return I2C_ENOTSUP;
  }

  static int riccati(const I2cProblem* p, const void* prior_out, const void* fwd, const void* xm, void* post, void* ric,
                     int32_t* status, void* stream) {
    if constexpr (MIXED) return I2C_ENOTSUP;
    if constexpr (LANE) {
      if (use_group(p) != 0) return I2C_ENOTSUP;
      const C c = make_consts<M, R>(p, 0.0, 0);
      RiccatiArgs<R> a{(const R*)prior_out, (const R*)fwd, (const R*)xm, (const R*)p->z, (const R*)p->alpha,
                       (R*)post,            (R*)ric,       status};
      return launch(k_riccati<M, R>, p->B, 1, LANE_BLOCK, stream, c, a);
    }
    return I2C_ENOTSUP;
  }

  static int mstep(const I2cProblem* p, const void* term_stats, double tol, int update, void* stats_out,
                   void* stream) {
    const C c = make_consts<M, R>(p, tol, 0);
    MstepArgs<R> a{(const R*)term_stats, (R*)p->alpha, (R*)stats_out, update};
    return launch(k_mstep<M, R>, p->B, 1, LANE_BLOCK, stream, c, a);
  }

  // _update_priors (i2c.py:1210-1213): cells with index <= tau switch to feedback mode
  static int to_feedback(const I2cProblem* p, int tau, void* stream) {
    const int n = tau + 1 < p->T ? tau + 1 : p->T;  // cells 0 .. n-1 = ring rows t0 .. t0+n-1 (mod T): at most two spans
    return launch(k_to_feedback<M>, n, 1, CELL_BLOCK, stream, const_cast<uint8_t*>(p->feedforward), p->t0, n, p->T);
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
      if (!fuse.done) {  // fused / group / Linearize / Gauss-Hermite schedules have no reduction kernel
        rc = mstep(p, term_stats, tol, 1, (R*)stats_hist + (size_t)it * 4 * p->B, stream);
        if (rc != I2C_OK) return rc;
      }
      if (tau > 0 && it == 0) {  // idempotent: once per call
        rc = to_feedback(p, tau, stream);
        if (rc != I2C_OK) return rc;
      }
    }
    return I2C_OK;
  }

  // n_iters EM iterations WITH closed-loop propagation (covariance control: learn_msgs with _propagate, i2c.py:1238-1251) enqueued by
  // one call: forward, backward, propagate, M-step per iteration. From the second iteration on the propagation of iteration k
  // shares a launch with the forward sweep of iteration k + 1 (k_forward_propagate) where both run on the lane kernels; the
  // first one runs in order -- its M-step is followed by the mode-flag switch the propagation reads. Same kernels' bodies, same
  // inputs as the one-by-one calls: identical results (a trajectory that fails in BOTH overlapped sweeps records either code).
  static int learn_propagate(const I2cProblem* p, void* post, void* fwd, void* xm, void* zpost, void* cell_stats, void* term_stats,
                             void* prop, void* prop_hist, double tol, int tau, int n_iters, void* stats_hist, int use_expert,
                             int overlap, int32_t* status, void* stream) {
    if constexpr (MIXED) return I2C_ENOTSUP;
    auto prop_row = [&](int it) { return (void*)((R*)prop_hist + (size_t)it * 3 * p->B); };
    bool fusable = false;
    if constexpr (LANE && C::D <= 5) {
      const C c0 = make_consts<M, R>(p, 0.0, use_expert);
      fusable = overlap && p->inference == I2C_INF_CUBATURE && family(p, c0, I2C_SWEEP_FORWARD) == I2C_FAMILY_LANE &&
                family(p, c0, I2C_SWEEP_PROPAGATE) == I2C_FAMILY_LANE && c0.rule_xu.unit && c0.rule_x.unit && !c0.z_per_cell &&
                !p->alpha_cell && c0.t0 == 0;
    }
    int pending = -1;  // iteration whose propagation has not run yet
    for (int it = 0; it < n_iters; ++it) {
      int rc = I2C_OK;
      bool fused = false;
      if constexpr (LANE && C::D <= 5) {
        if (fusable && pending >= 0) {
          const C c = make_consts<M, R>(p, 0.0, use_expert);
          FwdArgs<R> af{(const R*)post, (R*)fwd, nullptr, (const R*)p->x0, (const R*)p->sig_x0, (const R*)p->z, (const R*)p->alpha,
                        (const R*)p->alpha_cell, p->feedforward, status, p->expert};
          PropArgs<R> ap{(const R*)post, (R*)prop, (R*)prop_row(pending), (const R*)p->x0, (const R*)p->sig_x0,
                         (const R*)p->z, p->feedforward, status, p->expert};
          rc = launch(k_forward_propagate<M, R, true>, p->B, 2, LANE_BLOCK, stream, c, af, ap);
          fused = true;
          pending = -1;
        }
      }
      if (!fused) {
        if (pending >= 0) {
          rc = propagate(p, post, prop, prop_row(pending), use_expert, status, stream);
          pending = -1;
          if (rc != I2C_OK) return rc;
        }
        rc = forward(p, post, fwd, nullptr, status, stream);
      }
      if (rc != I2C_OK) return rc;
      MstepFuse fuse{tol, 1, (R*)stats_hist + (size_t)it * 4 * p->B, false};
      rc = backward_impl(p, fwd, xm, post, zpost, cell_stats, term_stats, status, stream, &fuse);
      if (rc != I2C_OK) return rc;
      if (it == 0) {  // in order: the mode flags change right after this M-step
        rc = propagate(p, post, prop, prop_row(0), use_expert, status, stream);
        if (rc != I2C_OK) return rc;
      } else {
        pending = it;
      }
      if (!fuse.done) {
        rc = mstep(p, term_stats, tol, 1, (R*)stats_hist + (size_t)it * 4 * p->B, stream);
        if (rc != I2C_OK) return rc;
      }
      if (tau > 0 && it == 0) {
        rc = to_feedback(p, tau, stream);
        if (rc != I2C_OK) return rc;
      }
    }
    if (pending >= 0) return propagate(p, post, prop, prop_row(pending), use_expert, status, stream);
    return I2C_OK;
  }

  static int ckf(const I2cProblem* p, const double* sig_zeta, const void* y, const void* u, void* mu, void* cov,
                 int32_t* status, void* stream) {
    if constexpr (MIXED) return I2C_ENOTSUP;
    const I2cProblem q = filter_problem(p);
    p = &q;
    const C c = make_consts<M, R>(p, 0.0, 0);
    ZetaArg<M, R> z;
    for (int i = 0; i < sym(M::NY); ++i) z.v[i] = (R)sig_zeta[i];
    CkfArgs<R> a{(const R*)y, (const R*)u, (R*)mu, (R*)cov, status};
    const int fam = family(p, c, I2C_SWEEP_FILTER);
    if (fam < 0) return fam;
    if (fam == I2C_FAMILY_QUAD) {
      if constexpr (HAS_QUAD_CKF) return launch_quad_ckf<M, R>(c, z, a, stream);
    }
    if (fam == I2C_FAMILY_GROUP) {
      if constexpr (HAS_GROUP) return launch_group<GK_CKF, M, R, G>(c, &z, a, stream);
    }
    if constexpr (LANE) return launch(k_ckf<M, R>, p->B, 1, LANE_BLOCK, stream, c, z, a);
    return I2C_ENOTSUP;
  }

  // One control step of the MPC loop enqueued by one call (i2c/policy/mpc.py:156-182): filter, n_iter x (forward,
  // backward, _update_priors), first action, and the horizon shift: one fresh row written into the ring of per-cell buffers
  // (the caller then advances I2cProblem.t0 by one and moves terminal_cell).
  static int mpc_step(const I2cProblem* p, const I2cMpcStep* m, void* stream) {
    if constexpr (MIXED) return I2C_ENOTSUP;
    int rc = I2C_OK;
    if (m->do_filter) rc = ckf(p, m->sig_zeta, m->y, m->u, const_cast<void*>(p->x0), const_cast<void*>(p->sig_x0), m->status, stream);
    for (int it = 0; it < m->n_iter && rc == I2C_OK; ++it) {
      rc = forward(p, m->post, m->fwd, nullptr, m->status, stream);
      if (rc == I2C_OK) rc = backward(p, m->fwd, m->xm, m->post, m->zpost, m->cell_stats, m->term_stats, m->status, stream);
      if (rc == I2C_OK && m->tau > 0 && it == 0) rc = to_feedback(p, m->tau, stream);  // (idempotent within a step: _update_priors)
    }
    if (rc != I2C_OK) return rc;
    return shift(p, m->post, m->cell_init, m->alpha_init, m->z_new, m->action, stream);
  }
  // the receding-horizon shift alone (BatchedI2c.shift_horizon, the step-by-step path of the policies)
  static int shift(const I2cProblem* p, void* post, const void* cell_init, const void* alpha_init, const void* z_new, void* action,
                   void* stream) {
    if constexpr (MIXED) return I2C_ENOTSUP;
    const C c = make_consts<M, R>(p, 0.0, 0);
    ShiftArgs<R> a{(R*)post, (const R*)cell_init, (R*)const_cast<void*>(p->alpha_cell), (const R*)alpha_init,
                   (R*)const_cast<void*>(p->z_per_cell ? p->z : nullptr), (const R*)z_new, const_cast<uint8_t*>(p->feedforward),
                   (R*)action};
    // the per-trajectory part (first action out, temperature, target, mode flag of the fresh cell), then the fresh cell: one
    // contiguous block copy (a lane-per-trajectory loop over the e_post elements took 111 us at B = 1024 for the 12-state
    // quadrotor: 230 dependent partial-line stores per lane -- a tenth of the control step)
    int rc = launch(k_mpc_shift<M, R>, p->B, 1, CELL_BLOCK, stream, c, a);
    if (rc == I2C_OK)
      rc = copy_bytes((R*)post + (size_t)c.row(0) * C::E_POST * p->B, cell_init, (size_t)C::E_POST * p->B * sizeof(R), stream);
    return rc;
  }

  static int rollout(const I2cProblem* p, const void* post, int n_rollouts, int policy, const void* eps_x0,
                     const void* eps_x, const void* eps_u, void* xu, void* z, void* x_final, void* z_term,
                     void* stream) {
    if constexpr (MIXED) return I2C_ENOTSUP;
    const C c = make_consts<M, R>(p, 0.0, 0);
    RolloutArgs<R> a{(const R*)post, (const R*)p->x0, (const R*)p->sig_x0, (const R*)eps_x0, (const R*)eps_x,
                     (const R*)eps_u, (R*)xu, (R*)z, (R*)x_final, (R*)z_term, n_rollouts, policy};
    return launch(k_rollout<M, R>, (long)n_rollouts * p->B, 1, LANE_BLOCK, stream, c, a);
  }

  static int propagate(const I2cProblem* p, const void* post, void* prop, void* prop_stats, int use_expert,
                       int32_t* status, void* stream) {
    if constexpr (MIXED) return I2C_ENOTSUP;
    // under Linearize() the closed-loop propagation IS CubatureQuadrature(1, 0, 0) whatever the caller left in the quad fields
    // (i2c.py:109-115), like the state estimator's rule (filter_problem)
    I2cProblem q = *p;
    if (p->inference == I2C_INF_LINEARIZE) q.quad_alpha = 1.0, q.quad_beta = 0.0, q.quad_kappa = 0.0;
    p = &q;
    const C c = make_consts<M, R>(p, 0.0, use_expert);
    PropArgs<R> a{(const R*)post, (R*)prop, (R*)prop_stats, (const R*)p->x0, (const R*)p->sig_x0,
                  (const R*)p->z, p->feedforward, status, p->expert};
    const int fam = family(p, c, I2C_SWEEP_PROPAGATE);
    if (fam < 0) return fam;
    if (fam == I2C_FAMILY_QUAD) {
      if constexpr (HAS_QUAD_PROP) return launch_quad_propagate<M, R>(c, a, stream);
    }
    if (fam == I2C_FAMILY_GROUP) {
      if constexpr (HAS_GROUP) return launch_group<GK_PROPAGATE, M, R, G>(c, nullptr, a, stream);
    }
    if constexpr (LANE) {
      if (p->inference == I2C_INF_GAUSS_HERMITE) return launch(k_propagate<M, R, true>, p->B, 1, LANE_BLOCK, stream, c, a);
      return launch(k_propagate<M, R>, p->B, 1, LANE_BLOCK, stream, c, a);
    }
    return I2C_ENOTSUP;
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
  d->group_lanes = M::GROUP;
  d->group_only = M::GROUP_ONLY ? 1 : 0;
  d->wave = M::WAVE ? 1 : 0;
  d->quad = M::QUAD ? 1 : 0;
}

template <class M, typename R, typename S = R> const ModelOps* make_ops() {
  using I = Impl<M, R, S>;
  static const ModelOps ops = {&I::forward, &I::backward,  &I::mstep,        &I::learn,           &I::ckf,
                               &I::rollout, &I::propagate, &I::riccati,   &I::mpc_step,        &fill_dims<M>,
                               &workspace_elems<M>, &I::plan, &I::shift, &I::family_of, &I::learn_propagate};
  return &ops;
}

}  // namespace i2c
