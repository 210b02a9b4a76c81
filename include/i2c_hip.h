/*
 * i2c_hip.h -- C ABI of the MI355X (gfx950) batched Gaussian i2c solver.
 *
 * The reference (JoeMWatson/input-inference-for-control) is pure Python and has no FFI;
 * its boundary for this path is the object protocol of I2cGraph / I2cCell (i2c/i2c.py).
 * Each entry point below replaces the reference method named in its comment for B
 * independent trajectories at once. All citations are relative to the reference root.
 *
 * Conventions
 *  - No torch / C++ types: plain pointers and sizes. `stream` is a hipStream_t passed as void*.
 *  - DEVICE buffers are struct-of-arrays with the trajectory index innermost:
 *      buf[t][e][b]  at element offset (t*E + e)*B + b,
 *    so that lane b of a wavefront reads element e of step t with one coalesced transaction.
 *    Element type = I2cProblem.dtype (I2C_F64: double, I2C_F32: float).
 *  - Symmetric n x n matrices are stored packed, lower triangle, row-major:
 *      (i,j), i>=j  ->  i*(i+1)/2 + j ;  I2C_SYM(n) = n(n+1)/2 elements.
 *  - Problem constants shared by the whole batch are small HOST arrays of double inside
 *    I2cProblem (they travel in the kernel-argument segment).
 *  - Every function returns 0 on success or a negative I2C_E* code for argument / launch errors.
 *    Per-trajectory numerical failures (a covariance that is not positive definite -- the
 *    reference raises LinAlgError from quadrature.py:17-24 or scipy) never abort the batch:
 *    they are recorded in status[b] = (reason << 16) | (t + 1) for the FIRST failing step.
 *  - The library keeps no global mutable state apart from the append-only table of out-of-tree models (i2c_register_model,
 *    mutex-guarded); calls on different streams are independent (tests/test_streams.py runs two engines on two streams).
 *    Nothing is allocated or freed by the library: every buffer is owned by the caller.
 */
#ifndef I2C_HIP_H
#define I2C_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2C_ABI_VERSION 7

#define I2C_MAX_NX 12
#define I2C_MAX_NU 4
#define I2C_MAX_NZ 16
#define I2C_MAX_PARAMS 16
#define I2C_MAX_GH_DEGREE 8
#define I2C_SYM(n) ((n) * ((n) + 1) / 2)

/* model plugins (the reference's `sys` objects, i2c/model.py:19-44) */
enum {
  I2C_MODEL_PENDULUM = 0,         /* PendulumKnown         env_def.py:233-309, env_autograd.py:5-19   */
  I2C_MODEL_PENDULUM_ACTREG = 1,  /* PendulumKnownActReg   env_def.py:312-346                          */
  I2C_MODEL_CARTPOLE = 2,         /* CartpoleKnown         env_def.py:491-612, env_autograd.py:25-54  */
  I2C_MODEL_DOUBLE_CARTPOLE = 3,  /* DoubleCartpoleKnown   env_def.py:615-761, env_autograd.py:60-167 */
  I2C_MODEL_LINEAR = 4,           /* LinearKnown           env_def.py:139-191, model.py:226-246       */
  I2C_MODEL_LINEAR_MINENERGY = 5, /* LinearKnownMinimumEnergy env_def.py:194-230                      */
  I2C_MODEL_QUADROTOR = 6,        /* build-defined planar quadrotor (Box2D physics is not reproducible) */
  I2C_MODEL_QUADROTOR12 = 7,      /* build-defined 12-state / 4-rotor quadrotor (BASELINE config 4: nx = 12); wave + group kernels */
  I2C_NUM_MODELS = 8
};
/* Out-of-tree models: ids I2C_MODEL_PLUGIN_BASE .. handed out by i2c_register_model / i2c_load_model (below). The reference takes
 * ANY object with dim_*, forward, observe, observe_terminal_x as a model (i2c/model.py:19-44, 154-156; env_def.py:34-82); here a
 * model is a functor struct in a header, compiled into a library of its own without touching this tree. */
#define I2C_MODEL_PLUGIN_BASE 64
#define I2C_MAX_PLUGIN_MODELS 32
typedef struct I2cModelOps I2cModelOps; /* opaque: the per-(model, dtype) table of entry points a model library exports */

/* I2cProblem.dtype. I2C_F64: the reference's arithmetic, parity-grade. I2C_F64_F32S: fp64 ARITHMETIC on fp32-STORED per-cell
 * buffers (prior/post, fwd, xm, zpost, prior_out are float; everything per trajectory -- x0, sig_x0, z, alpha, alpha_cell, temp,
 * term_stats, cell_stats, stats, the chunk workspace -- stays double): half the HBM bytes of the sweeps; deviation from fp64
 * bounded (tests/test_precision.py). Only the cubature EM path (forward, backward, M-step, i2c_learn) of the one-lane, the quad
 * and the wave kernels; other entry points return I2C_ENOTSUP. I2C_F32: fp32 arithmetic and storage -- NOT parity-grade (the sigma-point
 * curvature terms are below fp32 resolution; O(1) deviation after a few EM iterations): for tolerance sweeps only. */
enum { I2C_F64 = 0, I2C_F32 = 1, I2C_F64_F32S = 2 };

/* inference method of the E-step (the `inference` argument of I2cGraph, i2c/exp_types.py:22-68) */
enum {
  I2C_INF_CUBATURE = 0,  /* CubatureQuadrature(alpha, beta, kappa): sigma points (i2c.py:350-447, 544-610)                */
  I2C_INF_LINEARIZE = 1, /* Linearize(): first-order expansion about the means (i2c.py:244-348, 449-542); the plan cost
                            and the closed-loop propagation still use CubatureQuadrature(1, 0, 0) (i2c.py:109-115, 841-844) */
  I2C_INF_GAUSS_HERMITE = 2 /* GaussHermiteQuadrature(degree) (exp_types.py:52-68): the sigma-point cells with a tensor
                            grid of gh_degree^dim points m + sqrt(2) L xi; one backward schedule (a lane per trajectory) */
};

/* how i2c_backward_sweep is scheduled (results are identical up to summation order of the cost) */
enum {
  I2C_BWD_AUTO = 0,     /* resolved by i2c_backward_schedule(): chunked below the model's crossover batch, fused from there on */
  I2C_BWD_TWO_PASS = 1, /* sequential nx x nx scan + one lane per (t, b) + reduction                  */
  I2C_BWD_FUSED = 2,    /* one lane per trajectory does the whole cell: lowest HBM traffic            */
  I2C_BWD_CHUNKED = 3   /* the affine x-recursion composed per chunk of cells: sequential depth ~2T/NC; needs `work` */
};
#define I2C_BWD_FUSED_MIN_B 12288 /* library-wide default, re-derived in round 5 from time and HBM traffic at B = 8192 .. 32768 (pendulum,
                                     cartpole, planar quadrotor all cross between 8192 and 16384; it was 32768); a model may carry its own
                                     (double cartpole: 20480) -- profiles/r5_backward_crossover.txt; i2c_backward_schedule() answers */

/* kernel families (i2c_kernel_family): how the lanes of a wavefront are mapped onto trajectories */
enum {
  I2C_FAMILY_LANE = 1,  /* one trajectory per lane, every block in that lane's registers (d = nx + nu <= 8)              */
  I2C_FAMILY_GROUP = 2, /* G = I2cDims.group_lanes lanes per trajectory, blocks row-distributed, exchanged through LDS  */
  I2C_FAMILY_WAVE = 3,  /* one wavefront per trajectory: 16 x 16 blocks in the MFMA accumulator layout (d = 16)          */
  I2C_FAMILY_QUAD = 4   /* four trajectories per wavefront: 4 x 4 blocks on v_mfma_f64_4x4x4_4b_f64, one element per lane
                           (forward and backward sweep: every model; propagation and filter step: d = 16)                  */
};
/* I2cProblem.group_lanes = I2C_LANES_QUAD asks for the quad kernels of a model that also has wave kernels (64 = the
 * wave kernels there): the 12-state quadrotor -- forward and backward sweep, at any batch size. On a d <= 8 model it asks for the
 * quad FORWARD sweep only: the backward sweep resolves as group_lanes = 0 does (so I2C_LANES_QUAD with I2C_BWD_CHUNKED is the
 * quad forward sweep + the chunked schedule on the lane walker at any batch size; its compose / stitch passes follow the default
 * windows, I2C_SWEEP_CHUNK_PASSES / _STITCH). */
#define I2C_LANES_QUAD 164
enum { I2C_SWEEP_FORWARD = 0, I2C_SWEEP_BACKWARD = 1, I2C_SWEEP_PROPAGATE = 2, I2C_SWEEP_FILTER = 3,
       /* the compose + stitch passes of the CHUNKED backward schedule (sigma-point rules): I2C_FAMILY_LANE, or I2C_FAMILY_QUAD inside
        * the model's measured window / when group_lanes = 64 asks for the chunked schedule; I2C_ENOTSUP when the problem's backward
        * sweep does not run that schedule. The walk pass is what I2C_SWEEP_BACKWARD reports. */
       I2C_SWEEP_CHUNK_PASSES = 4,
       /* the stitch pass of that schedule alone: it keeps the quad form to larger batches than the compose pass (a chain of NC
        * dependent steps on few wavefronts at any batch size) */
       I2C_SWEEP_CHUNK_STITCH = 5 };
/* hybrid default of a d >= 7 lane model WHOSE QUAD FORWARD KERNEL DOES NOT APPLY (none of the in-tree models since round 6: the quad
 * kernels take any cubature weights; an out-of-tree model with GROUP_FORWARD_AUTO and no quad form): its FORWARD sweep runs on the
 * group kernels while B * G stays within this many lanes (every group wave then has a SIMD of its own: 1024 SIMDs x 64 lanes).
 * i2c_kernel_family() reports the family of a sweep. */
#define I2C_GROUP_FORWARD_MAX_LANES 65536
/* The family -- hence the last bits of a result -- of the d >= 5 models depends on the batch size: a caller who needs results
 * that do not depend on how a batch is sharded pins it (group_lanes = -1, 64 or I2C_LANES_QUAD); INTEGRATION.md section 1b. */

enum {
  I2C_OK = 0,
  I2C_EINVAL = -1,   /* bad argument (null pointer, unknown model / dtype, B or T < 1, T > 65535) */
  I2C_ENOTSUP = -2,  /* combination not compiled in */
  I2C_ELAUNCH = -3   /* HIP launch error (hipGetLastError != hipSuccess) */
};

/* status[b] reason codes (upper 16 bits) */
enum {
  I2C_FAIL_PRIOR_JOINT = 1,   /* chol(sig_xu0_f) failed       (quadrature.py:17-24 via i2c.py:391)   */
  I2C_FAIL_PDF_RATIO = 2,     /* chol(sig_xx + sig_x0_f)      (scipy mvn, i2c.py:369-374)            */
  I2C_FAIL_OBS_COV = 3,       /* chol(sig_z0_f + sig_xi)      (la.solve assume_a='pos', i2c.py:398)  */
  I2C_FAIL_UPDATED_JOINT = 4, /* chol(sig_xu1_f)              (quadrature.py:17-24 via i2c.py:415)   */
  I2C_FAIL_PRED_COV = 5,      /* chol(sig_x3_f)               (i2c.py:423)                           */
  I2C_FAIL_TERMINAL = 6,      /* terminal observation update  (i2c.py:430-443, 548-570)              */
  I2C_FAIL_POSTERIOR = 7,     /* chol(sig_xu0_m)              (quadrature.py:17-24 via i2c.py:594)   */
  I2C_FAIL_PROPAGATE = 8,     /* closed-loop propagation      (i2c.py:196-197)                       */
  I2C_FAIL_FILTER = 9,        /* cubature Kalman filter step  (mpc.py:129, 140-142)                  */
  I2C_FAIL_RICCATI = 10       /* improper backward message in the Riccati form (singular inv, i2c.py:615-660) */
};

typedef struct I2cDims {
  int32_t nx, nu, nz, nzt; /* dim_x, dim_u, dim_z, dim_z_term (nzt = 0: no terminal observation) */
  int32_t e_post;          /* elements per cell of the posterior buffer  (see I2cPosterior) */
  int32_t e_fwd;           /* elements per cell of the forward buffer    (see I2cFwdState)  */
  int32_t e_xm;            /* elements per cell of the smoothed-state buffer: nx + SYM(nx)  */
  int32_t e_zpost;         /* elements per cell of the optional z-moment buffer: nz + SYM(nz) */
  int32_t e_prop;          /* elements per cell of the propagation buffer (see I2cProp)     */
  int32_t n_params;        /* number of doubles the model reads from I2cProblem.model_params */
  int32_t ny;              /* dim_y of sys.measure (state estimator of the MPC loop)         */
  int32_t group_lanes;     /* G of the model's group kernels (G lanes of a wavefront per trajectory, blocks row-distributed
                              over the lanes and exchanged through LDS); 0: none compiled                                  */
  int32_t group_only;      /* 1: no one-lane-per-trajectory kernels exist for this model (nx + nu > 8 does not fit one lane)  */
  int32_t wave;            /* 1: the one-wavefront-per-trajectory kernels (I2C_FAMILY_WAVE) exist for this model;
                              I2cProblem.group_lanes = 64 asks for them                                                    */
  int32_t quad;            /* 1: the four-trajectories-per-wavefront forward kernel (I2C_FAMILY_QUAD) exists for this model;
                              I2cProblem.group_lanes = 64 asks for it                                                      */
} I2cDims;

/*
 * One optimisation problem = B independent trajectories of the same model and horizon.
 * Mirrors the constructor of I2cGraph (i2c/i2c.py:735-848) plus the per-cell state that the
 * reference keeps on I2cCell objects (i2c/i2c.py:81-148).
 */
typedef struct I2cProblem {
  int32_t abi_version; /* = I2C_ABI_VERSION */
  int32_t model_id;
  int32_t dtype;
  int32_t B, T;
  int32_t has_Qf;          /* terminal cost observation (Qf given, i2c.py:787-793)             */
  int32_t has_x_terminal;  /* covariance control: terminal state prior (i2c.py:548-559)        */
  int32_t z_per_cell;      /* 0: target = zg for every cell; 1: device targets `z` [T][nz][B]  */
  int32_t backward_mode;   /* I2C_BWD_AUTO | I2C_BWD_TWO_PASS | I2C_BWD_FUSED | I2C_BWD_CHUNKED (see i2c_backward_sweep) */
  int32_t terminal_cell;   /* index of the cell whose FORWARD pass applies the terminal cost update (i2c.py:430-443;
                              `terminal_cell` flag, i2c.py:82,822); T-1 normally, moves with the MPC shift, -1: none */
  int32_t inference;       /* I2C_INF_CUBATURE | I2C_INF_LINEARIZE                                                  */
  int32_t expert_controller; /* Linearize forward pass only: scale the feedback gain by the pdf ratio (use_expert_controller,
                              i2c.py:143,259-265); the cubature forward pass always scales it (i2c.py:366-375)      */
  int32_t gh_degree;       /* I2C_INF_GAUSS_HERMITE: 1 <= degree <= I2C_MAX_GH_DEGREE                               */
  int32_t group_lanes;     /* which kernel family serves a sweep -- resolved in ONE place, reported by i2c_kernel_family():
                              0: the model's default per sweep and batch size (one lane per trajectory; the quad forward kernel
                              of the d >= 5 models inside their batch windows; for the d = 16 model the wave kernels up to 1024
                              trajectories, the quad forward sweep above 1024 and the quad backward sweep above 2048; the
                              group kernels for what those forms do not cover);
                              64: the matrix-instruction family: the wave kernels where they exist (I2cDims.wave: one wavefront per
                              trajectory, forward and backward sweeps), the quad kernels otherwise (I2cDims.quad: four trajectories per
                              wavefront) -- the forward sweep and, since round 6, the backward sweep of every d <= 8 model: the fused walk
                              (one pass over the forward messages, no chunk workspace) with backward_mode I2C_BWD_AUTO or
                              I2C_BWD_FUSED, the chunked schedule with compose, stitch and walk passes in the quad form (the reduction
                              stays a lane kernel; I2cProblem.work as for the lane schedule) with I2C_BWD_CHUNKED. Parts of that are
                              also DEFAULTS of the d >= 5 models at small batches: the quad WALKER up to 64 (cartpole) / 256 (double
                              cartpole, planar quadrotor) trajectories (i2c_kernel_family(.., I2C_SWEEP_BACKWARD) answers
                              I2C_FAMILY_QUAD, i2c_backward_schedule I2C_BWD_CHUNKED), the quad COMPOSE + STITCH passes up to
                              128 / 256 / 384 (I2C_SWEEP_CHUNK_PASSES), the stitch pass alone up to 2048 / 8192 / 4096
                              (I2C_SWEEP_CHUNK_STITCH); I2C_BWD_TWO_PASS keeps the lane kernels;
                              I2C_LANES_QUAD: the quad kernels of a model that also has wave kernels, at any batch size; on a d <= 8
                              model the quad forward sweep only (see the define);
                              (all of these: fp64 or I2C_F64_F32S; the wave kernels: cubature rule with lam = 0; the quad kernels: any
                              CubatureQuadrature(alpha, beta, kappa) -- with general weights the d = 16 model runs on them at every
                              batch size; the closed-loop propagation and the filter step of the d = 16 model run on the quad
                              kernels too: propagate_quad_body, ckf_quad_body)
                              I2cDims.group_lanes: run forward / backward / propagate / filter with that many lanes of
                              a wavefront per trajectory (fp64, cubature rule;
                              the backward sweep then has one schedule, the fused walk); -1: one lane per trajectory for
                              every sweep (I2C_ENOTSUP for a group_only model); anything else: I2C_ENOTSUP                     */
  int32_t t0;              /* ring offset of the PERSISTENT per-cell buffers -- prior/post, z, alpha_cell, feedforward: cell t of the
                              horizon lives in row (t0 + t) mod T. 0 outside the MPC loop; the receding-horizon shift (mpc.py:174-181)
                              is "t0 += 1" plus one fresh row (i2c_mpc_step / i2c_shift_horizon). Buffers a sweep writes for its
                              caller (fwd, xm, zpost, prior_out, prop, cell_stats) are indexed by the cell index t directly.          */
  int32_t post_layout;     /* layout of the posterior / prior buffers (`prior`, `post`, I2cMpcStep.post): 0 = [T][e_post][B] (every
                              model); 1 = trajectory-major [T][B][e_post] -- the e_post elements of one cell of one trajectory are
                              contiguous -- accepted by the models with wave kernels (I2cDims.wave), whose every kernel then
                              reads / writes a cell of a trajectory as a few cache lines instead of e_post scattered elements   */
  /* CubatureQuadrature(alpha, beta, kappa): i2c/exp_types.py:31-49 */
  double quad_alpha, quad_beta, quad_kappa;
  double dtemp;            /* terminal-prior annealing rate (i2c.py:66,552)                    */
  /* I2C_INF_GAUSS_HERMITE: numpy.polynomial.hermite.hermgauss(gh_degree) nodes and weights (exp_types.py:57) */
  double gh_nodes[I2C_MAX_GH_DEGREE], gh_weights[I2C_MAX_GH_DEGREE];
  /* HOST constants (double, packed lower where symmetric) */
  double sig_eta[I2C_SYM(I2C_MAX_NX)];   /* sys.sig_eta                                          */
  double sig_xi0[I2C_SYM(I2C_MAX_NZ)];   /* inv(QR)   (i2c.py:786)  -> sig_xi = alpha * sig_xi0  */
  double QR[I2C_SYM(I2C_MAX_NZ)];        /* blkdiag(Q, R) (i2c.py:781-784)                       */
  double sig_xiT0[I2C_SYM(I2C_MAX_NZ)];  /* inv(Qf)   (i2c.py:789)                               */
  double Qf[I2C_SYM(I2C_MAX_NZ)];
  double zg[I2C_MAX_NZ];                 /* sys.zg      (i2c.py:84)                              */
  double zg_term[I2C_MAX_NZ];            /* sys.zg_term (i2c.py:85)                              */
  double mu_x_term[I2C_MAX_NX];          /* mu_x_terminal (i2c.py:797-801)                       */
  double sig_x_term[I2C_SYM(I2C_MAX_NX)];
  double model_params[I2C_MAX_PARAMS];   /* model-specific (Linear*: A row-major, B, a)          */
  /* DEVICE per-trajectory data */
  const void* x0;       /* [nx][B]        sys.x0       (i2c.py:877)                              */
  const void* sig_x0;   /* [SYM(nx)][B]   sys.sig_x0   (i2c.py:878)                              */
  const void* z;        /* [T][nz][B] per-cell targets (mpc.py:29-31) or NULL when !z_per_cell   */
  void* alpha;          /* [B] temperature; read by the sweeps, updated in place by i2c_mstep    */
  const void* alpha_cell; /* optional [T][B]: per-cell temperature used INSTEAD of alpha[b] for sig_xi /
                             sig_xi_terminal of cell t. NULL normally; the MPC loop needs it because a cell
                             appended by deepcopy(cell_init) keeps its stale sig_xi (mpc.py:175, i2c.py:976-981). Linearize()
                             applies the terminal cost at the END of the chain: with the last cell's entry (i2c.py:475-491) */
  void* temp;           /* [B] terminal-prior temperature (i2c.py:147,552) or NULL               */
  void* work;           /* optional device workspace of i2c_workspace_bytes() bytes for the chunked backward  */
  const uint8_t* feedforward; /* [T] bytes: 1 = cell in feed-forward mode
                                 (state_action_independence, i2c.py:132,355,1212-1213)            */
  const uint8_t* expert;      /* optional [T] bytes (ring): per-cell use_expert_controller (i2c.py:143), read by the
                                 closed-loop propagation (i2c.py:160) and the Linearize forward pass (i2c.py:259); NULL: every
                                 cell takes the scalar flag (expert_controller / the argument of i2c_propagate)  */
} I2cProblem;

/*
 * Posterior + controller buffer, [T][e_post][B]; per cell, in this order:
 *   mu_xu0_m[d] | sig_xu0_m[SYM(d)] | K[nu*nx] (row-major) | k[nu] | sigK[SYM(nu)]
 * Written by i2c_backward_sweep (i2c.py:586-608). The forward sweep READS the first
 * d + SYM(d) + nu*nx elements of `prior` as the previous iteration's posterior and controller
 * (i2c.py:361-387; after _update_priors, i2c.py:1210-1221, prior joint == posterior, and the
 * feed-forward action prior mu_u0_f/sig_u0_f is always the u-block of that joint).
 * Before the first iteration the caller fills it with [x0; mu_u], blkdiag(sig_x0, sig_u), K = 0
 * (I2cCell.__init__, i2c.py:95-100,135).
 */

/*
 * Forward-message buffer, [T][e_fwd][B]; per cell:
 *   mu_xu1_f[d] | sig_xu1_f[SYM(d)] | mu_x3_f[nx] | sig_x3_f[SYM(nx)] | J_dyn[d*nx] (row-major)
 * (the quantities I2cCell keeps for the backward pass, i2c.py:402-428).
 * The buffer is scratch between the forward sweep that writes it and the backward sweep that reads it, and its layout belongs
 * to the kernel family that runs them (i2c_kernel_family): [T][e_fwd][B] for I2C_FAMILY_LANE and I2C_FAMILY_GROUP,
 * trajectory-major [T][B][e_fwd] for I2C_FAMILY_WAVE (a wavefront reads one cell of one trajectory: contiguous). Same size.
 */

/* Smoothed next-state buffer [T][e_xm][B]: mu_x3_m[nx] | sig_x3_m[SYM(nx)] (i2c.py:546-576). */

/* Optional posterior cost-observation moments [T][e_zpost][B]: mu_z0_m[nz] | sig_z0_m[SYM(nz)]. */

/* Query compile-time dimensions of a model. Replaces reading sys.dim_* (env_def.py:34-82). */
int i2c_query(int model_id, I2cDims* out);

/*
 * Bring a model that is not one of the I2C_NUM_MODELS compiled in: replaces "write a *Def mixin + a dynamics function and hand
 * the object to I2cGraph" (i2c/env_def.py:34-82, 233-298; i2c/env_autograd.py:5-19; i2c/model.py:19-44).
 * A model library (lib/libi2c_model_<name>.so, built from ONE header by `python build.py --model <header>`: the same kernels
 * instantiated for the header's functor struct) exports
 *     int i2c_model_abi_version(void);  const char* i2c_model_name(void);  const I2cModelOps* i2c_model_ops(int dtype);
 * i2c_load_model dlopens it, checks the ABI version and registers its tables; i2c_register_model does the registration alone for
 * a caller that linked or loaded the model library itself. Both return the model_id to put into I2cProblem.model_id (>=
 * I2C_MODEL_PLUGIN_BASE; the same library registered twice gets the same id) or a negative I2C_E* code: I2C_EINVAL for an ABI
 * mismatch, a missing symbol, or dimensions beyond the I2C_MAX_* capacities of I2cProblem; I2C_ENOTSUP when the table is full.
 * ops_f32 / ops_f64s may be NULL (that precision was not built). dims_out (optional) receives the model's dimensions.
 * Thread-safe; ids stay valid for the life of the process.
 */
int i2c_register_model(int abi_version, const I2cModelOps* ops_f64, const I2cModelOps* ops_f32, const I2cModelOps* ops_f64s,
                       I2cDims* dims_out);
int i2c_load_model(const char* path, I2cDims* dims_out);

/* Bytes of I2cProblem.work the chunked backward sweep needs for (model_id, dtype, B, T). */
size_t i2c_workspace_bytes(int model_id, int dtype, int B, int T);

/* The schedule i2c_backward_sweep WILL run for problem p -- the one place that resolves I2cProblem.backward_mode together with
 * everything else that decides it: the kernel family of the backward sweep (wave: fused walk, two-pass on request; quad / group:
 * fused walk), the inference rule (Linearize / Gauss-Hermite: fused walk, chunked at small batches), the storage type, and the
 * lane kernels' batch rule (I2C_BWD_AUTO: chunked below I2C_BWD_FUSED_MIN_B trajectories, fused from there on; chunked needs
 * T >= 8, two-pass otherwise). Reads the scalar fields of p only: callers ask BEFORE they allocate, and size their buffers from
 * the answer -- two-pass needs xm and cell_stats, chunked needs I2cProblem.work (without it the sweep falls back to two-pass).
 * Returns I2C_BWD_TWO_PASS / FUSED / CHUNKED, or the negative error code the sweep itself would return for this problem. */
int i2c_backward_schedule(const I2cProblem* p);

/* Which kernel family (I2C_FAMILY_*) will serve `sweep` (I2C_SWEEP_*) of problem p (scalar fields only) -- the one place that resolves
 * I2cProblem.group_lanes, the models' defaults and the batch-size thresholds, so that callers and tests can see (and pin) what
 * ran: the last bits of a result depend on the family. Returns a negative error code if that sweep would refuse the problem. */
int i2c_kernel_family(const I2cProblem* p, int sweep);

/* Library self-description: ABI version, and the gfx target it was compiled for. */
int i2c_abi_version(void);
size_t i2c_problem_size(void); /* sizeof(I2cProblem) as compiled: bindings compare it with their own struct layout */
const char* i2c_build_info(void);

/*
 * Forward filter over all T cells: replaces I2cGraph._forward_msgs (i2c.py:876-880) calling
 * I2cCell._forward_msgs_quadrature (i2c.py:350-447) for every cell.
 *   prior      [T][e_post][B]  in   (see above; only the first d+SYM(d)+nu*nx rows are read)
 *   fwd        [T][e_fwd][B]   out
 *   prior_out  [T][d+SYM(d)][B] out, optional (NULL to skip): the joint prior mu_xu0_f,
 *              sig_xu0_f built in each cell (i2c.py:360,381-387; get_state_action_prior :1191)
 *   status     [B] int32       in/out (only overwritten where it is 0)
 */
int i2c_forward_sweep(const I2cProblem* p, const void* prior, void* fwd, void* prior_out,
                      int32_t* status, void* stream);

/*
 * Backward smoother + controller extraction + M-step statistics: replaces
 * I2cGraph._backward_msgs (i2c.py:882-886) calling I2cCell._backward_msgs_quadrature
 * (i2c.py:544-610) and the per-cell parts of calc_cost / get_z_covar (i2c.py:1034-1053,
 * 680-683, 983-992).
 *   TWO-PASS (small batches): a light sequential scan of the x-marginal recursion (one lane per
 *   trajectory, nx x nx blocks only), then a fully parallel per-cell pass (one lane per (t, b)),
 *   then a deterministic reduction of the per-cell cost over t.
 *   FUSED (large batches): one lane per trajectory does everything; each forward row is read once.
 *   fwd        [T][e_fwd][B]   in
 *   xm         [T][e_xm][B]    out: mu_x3_m, sig_x3_m (two-pass: required workspace;
 *                              fused: optional, NULL to skip)
 *   post       [T][e_post][B]  out (may alias the `prior` given to the forward sweep)
 *   zpost      [T][e_zpost][B] out, optional (NULL to skip)
 *   cell_stats [T][2][B]       out: per-cell expected cost mean m_t and variance v_t
 *                              (compute_cost_gaussian, i2c.py:1034-1043); two-pass: required
 *                              workspace; fused: optional. m_t is also the cell's contribution
 *                              tr(QR (err err^T + sig_z0_m)) to alpha (i2c.py:913-919)
 *   term_stats [4 + nzt + SYM(nzt)][B] out: row 0 = tr(Qf (errT errT^T + sig_z3_m))
 *                              (i2c.py:989-992; 0 when !has_Qf), row 1 = sum_t m_t, row 2 = sum_t v_t,
 *                              then mu_z3_m, sig_z3_m; the last row is used by I2C_INF_LINEARIZE only
 * I2C_INF_LINEARIZE: one schedule (a lane per trajectory walks T-1..0; backward_mode is ignored); the terminal cost
 * update happens here, at the end of the chain (i2c.py:475-491), and needs p->alpha; zpost holds the LINEARISED marginal
 * observation h(mu), C sig_xx C^T + D sig_uu D^T (i2c.py:537-540), whose trace statistic feeds alpha: row 1 of term_stats
 * is sum_t tr(QR (err err^T + sig_z0_m)) while the plan cost of the graph's cubature transform (i2c.py:1034-1053) goes to
 * the last row (mean) and row 2 (variance).
 */
int i2c_backward_sweep(const I2cProblem* p, const void* fwd, void* xm, void* post, void* zpost,
                       void* cell_stats, void* term_stats, int32_t* status, void* stream);

/*
 * M-step on the temperature: replaces calc_cost's sums, calculate_alpha, update_alpha and
 * update_xi (i2c.py:913-981, 1045-1053). O(B): the sums over t come from term_stats.
 *   alpha_update_tol  as I2cGraph.alpha_update_tol (>= 0: clamp ratio to [tol, 2 - tol];
 *                     < 0: keep alpha)
 *   update            0: only report alpha_hat (compute_update_alpha(False))
 *   stats_out [4][B]  out: alpha_hat (alphas_desired), alpha after the clamp (alphas),
 *                     cost mean (costs_m), cost variance (costs_m_var)
 * p->alpha is updated in place when update != 0.
 */
int i2c_mstep(const I2cProblem* p, const void* term_stats, double alpha_update_tol, int update,
              void* stats_out, void* stream);

/*
 * One control step of the model-predictive loop in ONE call with no host round trip: replaces the body of
 * PartiallyObservedMpcPolicy.__call__ / MpcPolicy.__call__ (i2c/policy/mpc.py:156-182, 95-111):
 *   if do_filter: i2c_ckf_filter on (p->x0, p->sig_x0) with measurement y and previous action u (mpc.py:125-145)
 *   n_iter x { i2c_forward_sweep; i2c_backward_sweep; _update_priors (cells <= tau to feedback mode) }
 *   action <- cells[0].mu_u0_m, sig_u0_m;  cells.pop(0); cells.append(deepcopy(cell_init))    (mpc.py:171-181)
 * The shift is in place: the persistent per-cell buffers are a ring (I2cProblem.t0); the step writes the appended cell over
 * the row of the popped one, and the caller advances t0 by one (mod T) and moves terminal_cell before the next call.
 */
typedef struct I2cMpcStep {
  int32_t do_filter, n_iter, tau, reserved0;
  double sig_zeta[I2C_SYM(I2C_MAX_NZ)]; /* packed measurement noise, ny x ny (sys.sig_zeta) */
  const void* y;            /* [ny][B] or NULL */
  const void* u;            /* [nu][B] or NULL: the action applied since the last step */
  void* post;               /* [T][e_post][B] ring, in/out: prior in, posterior out, then the appended cell in row t0 */
  void* fwd;                /* [T][e_fwd][B] */
  void* xm;                 /* [T][e_xm][B] or NULL (as i2c_backward_sweep) */
  void* zpost;              /* optional */
  void* cell_stats;         /* as i2c_backward_sweep */
  void* term_stats;
  const void* cell_init;    /* one cell block in the layout of `post` ([e_post][B]; [B][e_post] with post_layout = 1): the cell
                               appended at the end of the horizon (I2cCell.__init__ state), written by a block copy */
  const void* alpha_init;   /* [B]: temperature the appended cell keeps (NULL iff p->alpha_cell is NULL) */
  const void* z_new;        /* [nz][B] target of the appended cell, or NULL: the previous last cell's */
  void* action;             /* [nu + SYM(nu)][B] out, or NULL: cells[0].mu_u0_m, sig_u0_m before the shift */
  int32_t* status;          /* [B] */
} I2cMpcStep;
int i2c_mpc_step(const I2cProblem* p, const I2cMpcStep* step, void* stream);

/* The receding-horizon shift alone: `cells.pop(0); cells.append(deepcopy(cell_init))` (mpc.py:174-181) on the ring of
 * persistent per-cell buffers: copies cells[0]'s action moments to `action` (optional), then writes the fresh cell (cell_init,
 * alpha_init, target z_new or the previous last cell's, feed-forward mode) over row t0 of post / p->alpha_cell / p->z /
 * p->feedforward. The caller advances p->t0 afterwards. */
int i2c_shift_horizon(const I2cProblem* p, void* post, const void* cell_init, const void* alpha_init, const void* z_new,
                      void* action, void* stream);

/*
 * Riccati-form backward messages after a Linearize forward/backward pass: replaces I2cGraph._backward_ricatti_msgs
 * (i2c.py:888-893) over I2cCell._backward_ricatti_msgs (i2c.py:612-678), the verification helper of
 * scripts/lqr_compare.py:175. Requires p->inference == I2C_INF_LINEARIZE.
 *   prior_out  [T][d+SYM(d)][B]  in: what i2c_forward_sweep wrote as `prior_out` (mu_xu0_f, sig_xu0_f)
 *   fwd        [T][e_fwd][B]     in
 *   xm         [T][e_xm][B]      in: what i2c_backward_sweep wrote (only cell T-1 is read)
 *   post       [T][e_post][B]    in/out: K, k, sigK are OVERWRITTEN with the Riccati-form controller, as in the reference
 *   ric        [T][nx + nx*nx][B] out: nu_x0_b, lambda_x0_b (row-major) -- the backward state message in information form
 */
int i2c_riccati_sweep(const I2cProblem* p, const void* prior_out, const void* fwd, const void* xm, void* post, void* ric,
                      int32_t* status, void* stream);

/*
 * n_iters complete EM iterations enqueued back to back on `stream` with no host round trip:
 * replaces a loop over I2cGraph.learn_msgs (i2c.py:1238-1245; scripts/i2c_run.py:89-94) when
 * closed-loop propagation is off. Per iteration: forward sweep, backward sweep, M-step, and
 * _update_priors' mode switch (cells 0..tau leave feed-forward mode, i2c.py:1210-1213; `tau` <= 0: none).
 *   stats_hist [n_iters][4][B] out: the M-step's stats_out of every iteration
 * Buffers as in the single calls; `prior` and `post` are the same buffer.
 */
int i2c_learn(const I2cProblem* p, void* post, void* fwd, void* xm, void* zpost, void* cell_stats,
              void* term_stats, double alpha_update_tol, int tau, int n_iters, void* stats_hist,
              int32_t* status, void* stream);

/*
 * i2c_learn WITH closed-loop propagation: n_iters x I2cGraph.learn_msgs of a graph whose `_propagate` is on (covariance control,
 * scripts/nonlinear_covariance_control.py:107-113; i2c.py:1238-1251): forward sweep, backward sweep, propagate, M-step per
 * iteration, enqueued back to back. With `overlap` != 0 the propagation of iteration k shares ONE launch with the forward sweep of
 * iteration k + 1 from the second iteration on (both only read the posterior, each is a chain of T dependent cells on a fraction
 * of the chip; lane kernels of the d <= 5 models, cubature rule) -- the results are those of the one-by-one calls.
 *   prop       [T][e_prop][B]     out: the last iteration's propagation (as i2c_propagate)
 *   prop_hist  [n_iters][3][B]    out: the prop_stats of every iteration
 *   stats_hist [n_iters][4][B]    out: as i2c_learn
 */
int i2c_learn_propagate(const I2cProblem* p, void* post, void* fwd, void* xm, void* zpost, void* cell_stats, void* term_stats,
                        void* prop, void* prop_hist, double alpha_update_tol, int tau, int n_iters, void* stats_hist,
                        int use_expert_controller, int overlap, int32_t* status, void* stream);

/*
 * Closed-loop propagation of the controller distribution: replaces I2cGraph.propagate
 * (i2c.py:1247-1251) calling I2cCell._propagate_forward_quadrature (i2c.py:150-199), plus the
 * per-cell propagated cost statistics (i2c.py:685-688, 1055-1063).
 *   post       [T][e_post][B]  in  (posterior + controller from i2c_backward_sweep)
 *   prop       [T][e_prop][B]  out: mu_xu0_pf[d] | sig_xu0_pf[SYM(d)] | mu_x3_pf[nx] | sig_x3_pf[SYM(nx)]
 *   prop_stats [3][B]          out: sum over cells of the propagated cost mean / variance; row 2 = KL(x3_pf[T-1] || terminal
 *                              prior) for covariance control (i2c.py:1012-1019, 1223-1229), 0 otherwise
 *                              (costs_pf, costs_pf_var; row 0 / (nz T) is alpha_pf, i2c.py:934-939)
 *   use_expert_controller      cells' use_expert_controller flag (i2c.py:143,160)
 */
int i2c_propagate(const I2cProblem* p, const void* post, void* prop, void* prop_stats,
                  int use_expert_controller, int32_t* status, void* stream);

/*
 * Policy rollouts through the noisy known model: replaces BaseSim.run / batch_eval
 * (i2c/env.py:40-103; the reference farms n_eval rollouts out to mp.Pool(10)) for the time-indexed
 * linear-Gaussian policies of i2c/policy/linear.py, whose parameters are read straight from `post`.
 * N = n_rollouts * B lanes; rollout n = r * B + b is the r-th rollout of trajectory b.
 *   policy   0: u = K_t x + k_t                       (TimeIndexedLinearGaussianPolicy, linear.py:31-43)
 *            1: u = mu_u,t + exp(-e) K_t (x - mu_x,t), e = (x-mu)^T sig_x,t^{-1} (x-mu) / 2  (expert, soft)
 *            2: as 1 with weight [e < 3]               (expert, hard; linear.py:73-90)
 *   eps_x0 [nx][N], eps_x [T][nx][N], eps_u [T][nu][N]: standard-normal samples, each optional (NULL =
 *            no noise of that kind): x0 ~ N(x0, sig_x0) (env.py:195-197), x' += chol(sig_eta) eps
 *            (env.py:184-186), u += chol(sigK) eps (linear.py:39-41)
 *   xu [T][d][N], z [T][nz][N], x_final [nx][N], z_term [nzt][N]: outputs, each optional
 */
int i2c_rollout(const I2cProblem* p, const void* post, int n_rollouts, int policy, const void* eps_x0,
                const void* eps_x, const void* eps_u, void* xu, void* z, void* x_final, void* z_term,
                void* stream);

/*
 * One cubature-Kalman-filter step of the MPC state estimator: replaces
 * PartiallyObservedMpcPolicy.filter (i2c/policy/mpc.py:125-145; duplicate in
 * scripts/mpc_state_est/mpc_quad.py:135-155): predict the belief N(mu, cov) through sys.forward
 * with the applied action u, then innovate on the measurement y through sys.measure.
 *   sig_zeta  HOST [SYM(ny)] double: measurement noise sys.sig_zeta
 *   y [ny][B], u [nu][B]      in
 *   mu [nx][B], cov [SYM(nx)][B]  in/out
 * Only p->model_id, dtype, B, quad_*, sig_eta and model_params are read from the problem.
 */
int i2c_ckf_filter(const I2cProblem* p, const double* sig_zeta, const void* y, const void* u, void* mu,
                   void* cov, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* I2C_HIP_H */
