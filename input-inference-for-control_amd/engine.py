"""Batched i2c engine: device state as PyTorch tensors, sweeps as HIP kernels through the C ABI.

`BatchedI2c` is the batched counterpart of the reference's I2cGraph (i2c/i2c.py:732-1314): B
independent trajectories of one model and horizon, optimised by the same EM iteration
(learn_msgs, i2c.py:1238-1245). The EM loop lives here in Python; every sweep is one call into
``libi2c_hip.so``. PyTorch is plumbing only: it owns the HBM buffers and the stream.

Device layout (see include/i2c_hip.h): every per-cell buffer is logically ``[T][E][B]``, symmetric matrices packed
(lower, row-major). Physically the trajectory index is innermost -- except for the models with wave kernels, whose
posterior / prior and forward-message buffers are trajectory-major (``[T][B][E]``) behind permuted views.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _native
from ._native import F32, F64, F64_F32S, I2cProblem


def sym_size(n):
    return n * (n + 1) // 2


def tril_index(n):
    r, c = np.tril_indices(n)
    return r, c  # row-major order of the lower triangle == packed order


def pack_sym_np(mat):
    """(..., n, n) -> (..., n(n+1)/2) packed lower, row-major."""
    mat = np.asarray(mat, dtype=np.float64)
    r, c = tril_index(mat.shape[-1])
    return mat[..., r, c]


def unpack_sym(packed, n):
    """torch (..., n(n+1)/2) -> (..., n, n) full symmetric."""
    r, c = tril_index(n)
    r_t = torch.as_tensor(r, device=packed.device)
    c_t = torch.as_tensor(c, device=packed.device)
    out = packed.new_zeros(packed.shape[:-1] + (n, n))
    out[..., r_t, c_t] = packed
    out[..., c_t, r_t] = packed
    return out


class I2cNumericalError(np.linalg.LinAlgError):
    """A covariance lost positive definiteness (the reference raises LinAlgError from
    np.linalg.cholesky / scipy, i2c/inference/quadrature.py:17-24)."""


class BatchedI2c:
    def __init__(self, model, horizon, Q, R, Qf, alpha, alpha_update_tol, mu_u, sig_u,
                 mu_x_terminal=None, sig_x_terminal=None, quad=(1.0, 0.0, 0.0), x0=None, sig_x0=None,
                 z_traj=None, batch=None, dtype=torch.float64, device=None, lib=None, dtemp=1.0,
                 keep_zpost=True, keep_prior=False, keep_xm=True, backward_mode="auto", inference="cubature",
                 gh_degree=None, group_lanes=0, storage_dtype=None, allow_inexact=False, keep_prior_joint=False, post_layout=None,
                 deterministic_family=False, overlap_propagation=True):
        self.lib = lib if lib is not None else _native.load_library()
        if device is None:
            device = "cpu" if self.lib.is_host_sim else "cuda"
        self.device = torch.device(device)
        self._dev_index = None
        if self.lib.is_host_sim != (self.device.type == "cpu"):
            raise RuntimeError(
                f"library '{self.lib.build_info}' cannot run on device {self.device}: the HIP build needs a "
                "GPU tensor device (there is no CPU fallback)"
            )
        # Precision. dtype = ARITHMETIC type: float64 is the reference's arithmetic and the only parity-grade one.
        #   storage_dtype=torch.float32 with dtype=torch.float64: the mixed mode I2C_F64_F32S -- fp64 arithmetic on fp32-STORED
        #     per-cell buffers (post, fwd, xm, zpost, prior_out); half the HBM bytes, deviation from fp64 bounded
        #     (tests/test_precision.py); cubature EM path of the one-lane kernels only.
        #   dtype=torch.float32: fp32 arithmetic. NOT parity-grade (the curvature terms of the sigma-point transform are
        #     below fp32 resolution: O(1) deviation after a few EM iterations with status 0), so it has to be asked for
        #     explicitly with allow_inexact=True (tolerance sweeps).
        assert dtype in (torch.float64, torch.float32)
        if dtype == torch.float32 and not allow_inexact:
            raise ValueError("dtype=torch.float32 runs the kernels in fp32 ARITHMETIC, which is not parity-grade (DESIGN.md section 4): "
                             "pass allow_inexact=True for a tolerance sweep, or storage_dtype=torch.float32 with dtype=torch.float64 "
                             "for fp32 storage with fp64 arithmetic")
        if storage_dtype not in (None, dtype, torch.float32):
            raise ValueError("storage_dtype must be None, the arithmetic dtype, or torch.float32")
        self.dtype = dtype
        self.store_dtype = dtype if storage_dtype is None else storage_dtype
        self.mixed = self.store_dtype != self.dtype
        self.sys = model
        self.model_id = int(model.resolve_model_id(self.lib)) if hasattr(model, "resolve_model_id") else int(model.model_id)
        dims = self.lib.query(self.model_id)
        self.dims = dims
        nx, nu, nz, nzt = dims.nx, dims.nu, dims.nz, dims.nzt
        assert (nx, nu, nz) == (model.dim_x, model.dim_u, model.dim_z), "model plugin / library dimension mismatch"
        self.nx, self.nu, self.nz, self.nzt, self.d = nx, nu, nz, nzt, nx + nu
        self.H = T = int(horizon)
        # group kernels (csrc/i2c_group.hpp): `group_lanes` lanes of a wavefront per trajectory. 0 = the model's default
        # (one lane per trajectory, except for models that only have group kernels; the d >= 7 lane models run their
        # FORWARD sweep on the group kernels at small batches); True = the model's group width; -1 = one lane per
        # trajectory for every sweep.
        if group_lanes is True:
            group_lanes = dims.group_lanes
        # deterministic_family=True: results that do NOT depend on the batch size or on how a batch is sharded over GPUs (round-4
        # review, weak #9). The defaults pick the fastest kernel family and backward schedule for the batch at hand -- the family
        # changes at measured batch windows, the chunk count of the chunked schedule with B -- and families / schedules round
        # differently in the last bits. The switch pins what `group_lanes=0` and `backward_mode="auto"` would otherwise resolve
        # per batch: one lane per trajectory with the sequential (fused) backward walk for the d <= 8 models, the wave kernels
        # with their fused walk for the d = 16 model. A trajectory's result is then a function of its own inputs only (tested).
        # It is the slow-but-reproducible path at small batches; the default stays the fast one.
        self.overlap_propagation = bool(overlap_propagation)  # learn(n) with closed-loop propagation: see _learn_with_propagation
        self.deterministic_family = bool(deterministic_family)
        if self.deterministic_family:
            if not group_lanes:
                group_lanes = 64 if dims.wave else -1
                if dims.wave and inference == "cubature" and tuple(float(v) for v in quad) != (1.0, 0.0, 0.0):
                    group_lanes = _native.LANES_QUAD  # (the wave kernels only have the unit rule: general weights are the quad kernels', round 6)
            if backward_mode == "auto":
                backward_mode = "fused"
        self.group_lanes = int(group_lanes or 0)
        # 64 = the wave kernels (csrc/i2c_wave.hpp: one wavefront per trajectory, blocks in the fp64 matrix-instruction layout),
        # for the models that have them (dims.wave): forward and backward sweeps; propagation and filter run the model's default
        ok = self.group_lanes in (0, dims.group_lanes) or (self.group_lanes == -1 and not dims.group_only) or \
            (self.group_lanes == 64 and (dims.wave or dims.quad)) or (self.group_lanes == _native.LANES_QUAD and dims.quad)
        if not ok:
            raise ValueError(f"group_lanes={self.group_lanes}: this model's group kernels use {dims.group_lanes} lanes"
                             + (" and it has no one-lane kernels" if dims.group_only else " (or -1: one lane per trajectory)")
                             + (" (or 64: one wavefront per trajectory)" if dims.wave else "")
                             + (" (or 64: four trajectories per wavefront, forward sweep)" if dims.quad else ""))
        # 64 on a model with the quad kernel (csrc/i2c_quad.hpp: four trajectories per wavefront on the 4 x 4 x 4 fp64 matrix
        # instruction): the FORWARD sweep runs on it, every other sweep on the model's default kernels
        # (_native.LANES_QUAD asks for it on a model that also has wave kernels, where 64 means those)
        self.quad_requested = bool((self.group_lanes == 64 and dims.quad and not dims.wave) or self.group_lanes == _native.LANES_QUAD)
        self.uses_group_kernels = bool((self.group_lanes > 0 and not self.quad_requested) or dims.group_only)  # a multi-lane family (group or wave) serves the sweeps
        if self.mixed and inference != "cubature":
            raise ValueError("fp32 storage (storage_dtype) is available for the cubature path only")
        if self.mixed and self.uses_group_kernels and not (dims.wave and self.group_lanes in (0, 64, _native.LANES_QUAD)):
            raise ValueError("fp32 storage (storage_dtype) is available for the one-lane, the quad and the wave kernels only")

        mu_u = np.asarray(mu_u, dtype=np.float64)
        if mu_u.ndim == 2:
            mu_u = mu_u[None]
        assert mu_u.shape[1:] == (T, nu), f"mu_u must be (T, nu) or (B, T, nu), got {mu_u.shape}"
        B = mu_u.shape[0]
        if x0 is not None:
            x0 = np.asarray(x0, dtype=np.float64).reshape(-1, nx)
            B = max(B, x0.shape[0])
        if batch is not None:
            B = max(B, int(batch))
        self.B = B
        mu_u = np.broadcast_to(mu_u, (B, T, nu))
        x0 = np.broadcast_to(np.asarray(model.x0, np.float64).reshape(nx) if x0 is None else x0, (B, nx))
        sig_x0 = np.asarray(model.sig_x0 if sig_x0 is None else sig_x0, dtype=np.float64)
        sig_x0 = np.broadcast_to(sig_x0, (B, nx, nx))
        sig_u = np.atleast_2d(np.asarray(sig_u, dtype=np.float64))
        self.mu_u0_base, self.sig_u0_base = np.array(mu_u), np.array(sig_u)  # the initial action prior (I2cCell.mu_u0_base, i2c.py:128-129)

        # cost model (i2c.py:778-793)
        R = np.atleast_2d(np.asarray(R, dtype=np.float64))
        if Q is not None:
            Q = np.atleast_2d(np.asarray(Q, dtype=np.float64))
            assert Q.shape[0] == Q.shape[1] and R.shape[0] == R.shape[1] and Q.shape[0] + R.shape[0] == nz, (
                f"blkdiag(Q, R) must be ({nz},{nz}), got Q {Q.shape}, R {R.shape}")
            QR = np.zeros((nz, nz))
            QR[: Q.shape[0], : Q.shape[0]] = Q
            QR[Q.shape[0]:, Q.shape[0]:] = R
        else:
            QR = R
        assert QR.shape == (nz, nz), f"blkdiag(Q, R) must be ({nz},{nz}), got {QR.shape}"
        assert np.allclose(QR, QR.T), "Q and R must be symmetric"
        self.Q, self.R, self.QR = Q, R, QR
        self.sig_xi0 = np.linalg.inv(QR)
        assert np.linalg.det(self.sig_xi0) > 0.0  # i2c.py:803-804
        self.has_Qf = Qf is not None and nzt > 0
        if Qf is not None:
            self.Qf = np.atleast_2d(np.asarray(Qf, dtype=np.float64))
            self.sig_xi_terminal_base = np.linalg.inv(self.Qf)
        else:
            self.Qf = np.zeros((nx, nx))  # i2c.py:792
            self.sig_xi_terminal_base = None
        self.alpha_update_tol = float(alpha_update_tol)
        # inference method of the E-step (I2cGraph's `inference`, i2c.py:103-127): "cubature" = sigma points with
        # CubatureQuadrature(*quad); "linearize" = Linearize(), whose plan cost and propagation use
        # CubatureQuadrature(1, 0, 0) (i2c.py:109-115, 841-844)
        # "gauss_hermite" = GaussHermiteQuadrature(gh_degree): tensor grid of gh_degree ** dim points (exp_types.py:52-68)
        if inference not in ("cubature", "linearize", "gauss_hermite"):
            raise ValueError(f"unknown inference method {inference!r}")
        self.inference = inference
        self.linearize = inference == "linearize"
        self.gauss_hermite = inference == "gauss_hermite"
        self.gh_degree = 0
        if self.gauss_hermite:
            self.gh_degree = int(gh_degree)
            if not 1 <= self.gh_degree <= _native.MAX_GH_DEGREE:
                raise ValueError(f"gh_degree must be in 1..{_native.MAX_GH_DEGREE}")
            if self.gh_degree ** (nx + nu) > 2 ** 31 - 1:
                raise ValueError("gh_degree ** (nx + nu) points do not fit the kernels' 32-bit point counter")
            quad = (1.0, 0.0, 0.0)
        if self.linearize:
            quad = (1.0, 0.0, 0.0)
            if nzt == 0:
                raise NotImplementedError("Linearize() needs a terminal observation: for this model the reference's "
                                          "observe_terminal_linearize returns None and i2c.py:500-501 fails")
        self.quad = tuple(float(q) for q in quad)
        self.mu_x_terminal = None if mu_x_terminal is None else np.asarray(mu_x_terminal, np.float64).reshape(nx)
        self.sig_x_terminal = None if sig_x_terminal is None else np.asarray(sig_x_terminal, np.float64)
        self.has_x_terminal = self.sig_x_terminal is not None
        if self.has_x_terminal and self.mu_x_terminal is None:
            raise TypeError("sig_x_terminal given without mu_x_terminal: the reference's cubature rule crashes on it (i2c.py:558) and the "
                            "mean its Linearize rule back-calculates (i2c.py:464-470) is not implemented here")
        self.dtemp = float(dtemp)

        dev, dt, st = self.device, self.dtype, self.store_dtype
        to = lambda a: torch.as_tensor(np.array(a, dtype=np.float64, order="C"), dtype=dt, device=dev)  # noqa: E731
        zeros = lambda *s: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        zeros_s = lambda *s: torch.zeros(*s, dtype=st, device=dev)  # noqa: E731  (per-cell buffers: storage type)
        d = self.d
        # initial "posterior" = cell constructor state (i2c.py:95-100, 135-136)
        post = np.zeros((T, dims.e_post, B))
        post[:, :nx, :] = x0.T[None]
        post[:, nx:d, :] = np.transpose(mu_u, (1, 2, 0))
        S0 = np.zeros((B, d, d))
        S0[:, :nx, :nx] = sig_x0
        S0[:, nx:, nx:] = sig_u
        post[:, d: d + sym_size(d), :] = pack_sym_np(S0).T[None]
        o_k = d + sym_size(d) + nu * nx
        post[:, o_k: o_k + nu, :] = np.transpose(mu_u, (1, 2, 0))  # k = mu_u (i2c.py:136)
        post[:, o_k + nu:, :] = pack_sym_np(sig_u)[None, :, None]
        # Layout of the posterior / prior buffers (I2cProblem.post_layout): logically always [T][e_post][B]; for the models with
        # wave kernels the STORAGE is trajectory-major, [T][B][e_post] (a cell of a trajectory is contiguous: a wavefront, which
        # works on one trajectory, reads it as a few cache lines), and self.post is a permuted view of it, so every index
        # expression in this file and its callers is layout-blind; only the library (data_ptr) sees the difference.
        # post_layout: an explicit constructor argument (it is part of the ABI, I2cProblem.post_layout: callers handing raw pointers
        # to the C API must know it); None = trajectory-major for the models with wave kernels, [T][e][B] for every other model
        if post_layout is None:
            post_layout = 1 if dims.wave else 0
        if post_layout not in (0, 1) or (post_layout == 1 and not dims.wave):
            raise ValueError(f"post_layout={post_layout!r}: 0 ([T][e][B]) or 1 (trajectory-major, models with wave kernels only)")
        self.post_layout = int(post_layout)
        self.post = to(post).to(st)
        if self.post_layout == 1:
            self.post = self.post.permute(0, 2, 1).contiguous().permute(0, 2, 1)
        # The forward sweep reads `prior`, the backward sweep writes `post`. Normally they are ONE buffer (after
        # _update_priors the prior IS the posterior, i2c.py:1210-1221, so the copy is free). keep_prior_joint=True keeps
        # them apart until update_priors(): two forward/backward passes without _update_priors() in between then start
        # from the same prior, as the reference's cells do (the drop-in I2cGraph asks for this; costs a second buffer).
        self.keep_prior_joint = bool(keep_prior_joint)
        self.prior = self.post
        self._post_spare = torch.empty_like(self.post) if self.keep_prior_joint else None
        self.fwd = zeros_s(T, dims.e_fwd, B)
        self.zpost = zeros_s(T, dims.e_zpost, B) if keep_zpost else None
        self.prior_out = zeros_s(T, d + sym_size(d), B) if keep_prior else None
        self.e_term = 4 + nzt + sym_size(nzt)
        self.term_stats = zeros(self.e_term, B)
        self.stats_out = zeros(4, B)
        self.prop = None
        self.prop_stats = None
        self.x0 = to(x0.T)
        self.sig_x0 = to(pack_sym_np(sig_x0).T)
        self.alpha = to(np.broadcast_to(np.asarray(alpha, np.float64), (B,)))
        self.temp = torch.ones(B, dtype=dt, device=dev)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.feedforward = torch.ones(T, dtype=torch.uint8, device=dev)  # i2c.py:132
        if z_traj is not None:
            z = np.broadcast_to(np.asarray(z_traj, np.float64), (B, T, nz))
            self.z = to(np.transpose(z, (1, 2, 0)))
        else:
            self.z = None
        self.zg = np.asarray(model.zg, np.float64).reshape(nz)
        self.zg_term = None if model.zg_term is None else np.asarray(model.zg_term, np.float64).reshape(-1)

        # Ring offset of the persistent per-cell buffers (post / prior, z, alpha_cell, feedforward): cell t of the horizon
        # lives in row (t0 + t) mod T (I2cProblem.t0). Only the MPC shift moves it; cells() / the getters undo it.
        self.t0 = 0
        self.alpha_cell = None      # [T][B] per-cell temperature, only in the MPC loop (see enable_per_cell_alpha)
        self.alpha_init = None
        self.terminal_cell = T - 1  # cell whose forward pass applies the terminal cost update (i2c.py:82,822)
        self.cell_init = self.post[0].clone()  # a fresh cell (I2cCell.__init__), appended by shift_horizon(): one cell block in the layout of post
        self.tau = T - 1  # i2c.py:833
        self._propagate = False
        self.use_expert_controller = True
        self.expert_cells = None  # optional [T] uint8 ring: per-cell use_expert_controller (set_cell_expert)
        self.em_iter = 0
        # per-iteration metrics as device tensors (B,): no host sync inside the EM loop
        self.alphas = [self.alpha.clone()]
        self.alphas_desired = [self.alpha.clone()]
        self.alphas_pf = [self.alpha.clone()]
        self.costs_m, self.costs_m_var, self.costs_pf, self.costs_pf_var, self.kl_terms = [], [], [], [], []
        self.backward_mode = {"auto": _native.BWD_AUTO, "two_pass": _native.BWD_TWO_PASS, "fused": _native.BWD_FUSED,
                              "chunked": _native.BWD_CHUNKED}[backward_mode]
        # ONE resolver (round-4 review, weak #8): the library decides which kernel family serves each sweep of THIS problem
        # (i2c_kernel_family) and which backward schedule will run (i2c_backward_schedule: inference rule, family, precision,
        # batch size, the request); the workspaces and the layout of the forward messages follow from its answers. A problem
        # the library refuses is refused here, with the library's code.
        self.work = None
        self._problem = self._make_problem()
        mode = self.lib.i2c_backward_schedule(C.byref(self._problem))
        if mode not in (_native.BWD_TWO_PASS, _native.BWD_FUSED, _native.BWD_CHUNKED):
            raise RuntimeError(f"i2c_backward_schedule() refused the problem with code {mode} (include/i2c_hip.h): model "
                               f"{type(model).__name__}, inference {inference!r}, group_lanes {self.group_lanes}, B {B}, T {T}")
        self.fused_backward = mode == _native.BWD_FUSED
        self.backward_schedule = {_native.BWD_TWO_PASS: "two_pass", _native.BWD_FUSED: "fused", _native.BWD_CHUNKED: "chunked"}[mode]
        # the two-pass backward needs xm / cell_stats as workspace; the fused and chunked ones only write xm on request
        two_pass = mode == _native.BWD_TWO_PASS
        self.xm = zeros_s(T, dims.e_xm, B) if (keep_xm or two_pass) else None
        self.cell_stats = zeros(T, 2, B) if two_pass else None
        if mode == _native.BWD_CHUNKED:
            nbytes = self.lib.i2c_workspace_bytes(self.model_id, F64 if dt == torch.float64 else F32, B, T)
            self.work = torch.empty(nbytes // (8 if dt == torch.float64 else 4), dtype=dt, device=dev)
        # The forward-message buffer is private to the kernel family that writes and reads it: the wave kernels and the d = 16 quad
        # backward sweep keep it trajectory-major, [T][B][e_fwd] (include/i2c_hip.h); forward_messages() reads it through a view
        # either way. The reader decides.
        fam_b, fam_f = self.kernel_family("backward"), self.kernel_family("forward")  # (raise with the library's code on a refusal)
        self.fwd_trajectory_major = fam_b == "wave" or (fam_b == "quad" and bool(dims.wave))
        if fam_f == "wave" and not self.fwd_trajectory_major:
            raise RuntimeError("the wave forward sweep needs the wave backward sweep (forward-message layout)")
        if self.fwd_trajectory_major:
            self.fwd = self.fwd.reshape(T, B, dims.e_fwd)
        self._problem = self._make_problem()  # (now with the workspace)

    # ------------------------------------------------------------------ C-ABI plumbing
    def _make_problem(self):
        p = I2cProblem()
        p.abi_version = _native.ABI_VERSION
        p.model_id = self.model_id
        p.dtype = F64_F32S if self.mixed else (F64 if self.dtype == torch.float64 else F32)
        p.B, p.T = self.B, self.H
        p.has_Qf = int(self.has_Qf)
        p.has_x_terminal = int(self.has_x_terminal)
        p.z_per_cell = int(self.z is not None)
        p.backward_mode = self.backward_mode
        p.terminal_cell = int(self.terminal_cell)
        p.inference = (_native.INF_LINEARIZE if self.linearize else
                       _native.INF_GAUSS_HERMITE if self.gauss_hermite else _native.INF_CUBATURE)
        p.gh_degree = self.gh_degree
        p.group_lanes = self.group_lanes
        p.t0 = int(self.t0)
        p.post_layout = int(self.post_layout)
        if self.gauss_hermite:
            gx, gw = np.polynomial.hermite.hermgauss(self.gh_degree)  # exp_types.py:57
            for i in range(self.gh_degree):
                p.gh_nodes[i], p.gh_weights[i] = float(gx[i]), float(gw[i])
        p.expert_controller = int(bool(self.use_expert_controller))
        p.quad_alpha, p.quad_beta, p.quad_kappa = self.quad
        p.dtemp = self.dtemp

        def put(dst, arr):
            arr = np.asarray(arr, np.float64).reshape(-1)
            for i, v in enumerate(arr):
                dst[i] = float(v)

        put(p.sig_eta, pack_sym_np(np.asarray(self.sys.sig_eta, np.float64)))
        put(p.sig_xi0, pack_sym_np(self.sig_xi0))
        put(p.QR, pack_sym_np(self.QR))
        if self.has_Qf:
            assert self.Qf.shape == (self.nzt, self.nzt)
            put(p.sig_xiT0, pack_sym_np(self.sig_xi_terminal_base))
            put(p.Qf, pack_sym_np(self.Qf))
            put(p.zg_term, self.zg_term)
        put(p.zg, self.zg)
        if self.has_x_terminal:
            put(p.mu_x_term, self.mu_x_terminal)
            put(p.sig_x_term, pack_sym_np(self.sig_x_terminal))
        params = list(self.sys.device_params()) if hasattr(self.sys, "device_params") else []
        assert len(params) == self.dims.n_params, "model plugin / library parameter-count mismatch"
        put(p.model_params, params)
        p.x0 = self.x0.data_ptr()
        p.sig_x0 = self.sig_x0.data_ptr()
        p.z = self.z.data_ptr() if self.z is not None else None
        p.alpha = self.alpha.data_ptr()
        p.alpha_cell = self.alpha_cell.data_ptr() if self.alpha_cell is not None else None
        p.temp = self.temp.data_ptr()
        p.work = self.work.data_ptr() if getattr(self, "work", None) is not None else None
        p.feedforward = self.feedforward.data_ptr()
        p.expert = self.expert_cells.data_ptr() if self.expert_cells is not None else None
        return p

    def kernel_family(self, sweep="forward"):
        """Which kernel family serves a sweep of THIS problem ("lane", "group", "wave" or "quad"): i2c_kernel_family(), the library's
        single resolver of group_lanes, model defaults and batch thresholds. The last bits of a result depend on it.
        "chunk_passes": the compose + stitch passes of the chunked backward schedule, "chunk_stitch": its stitch pass alone (refused
        when another schedule runs)."""
        code = {"forward": _native.SWEEP_FORWARD, "backward": _native.SWEEP_BACKWARD, "propagate": _native.SWEEP_PROPAGATE,
                "filter": _native.SWEEP_FILTER, "chunk_passes": _native.SWEEP_CHUNK_PASSES, "chunk_stitch": _native.SWEEP_CHUNK_STITCH}[sweep]
        rc = self.lib.i2c_kernel_family(C.byref(self._problem), code)
        self._check(0 if rc > 0 else (rc or -1), "i2c_kernel_family")
        return _native.FAMILY_NAMES[rc]

    @property
    def forward_family(self):
        return self.kernel_family("forward")

    @property
    def backward_family(self):
        return self.kernel_family("backward")

    def refresh_problem(self):
        """Rebuild the C problem descriptor after changing host-side constants / tensors."""
        self._problem = self._make_problem()

    def _stream(self):
        """The caller's current HIP stream on the engine's device (what torch.cuda.current_stream(device).cuda_stream returns, read
        through the raw accessor: three sweeps per EM iteration ask for it, and the Stream-object path costs ~9 us each -- a tenth of
        a single-trajectory iteration's host time)."""
        if self.device.type == "cuda":
            if self._dev_index is None:
                self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            if raw is not None:
                return C.c_void_p(raw(self._dev_index))
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    @staticmethod
    def _ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with code {rc} (see include/i2c_hip.h)")

    # ------------------------------------------------------------------ sweeps
    def forward_sweep(self):
        """I2cGraph._forward_msgs (i2c.py:876-880)."""
        self._problem.expert_controller = int(bool(self.use_expert_controller))
        if self.prior_out is not None:  # (the graph facade: the temperature THIS sweep ran at, for the cells' sig_z0_f = S_z + alpha xi)
            self.alpha_fwd = self.alpha.clone()
        rc = self.lib.i2c_forward_sweep(C.byref(self._problem), self._ptr(self.prior), self._ptr(self.fwd),
                                        self._ptr(self.prior_out), self._ptr(self.status), self._stream())
        self._check(rc, "i2c_forward_sweep")

    def backward_sweep(self):
        """I2cGraph._backward_msgs (i2c.py:882-886) + per-cell M-step statistics."""
        if self.keep_prior_joint and self.post is self.prior:  # do not overwrite the prior of this sweep
            self.post, self._post_spare = self._post_spare, None
        rc = self.lib.i2c_backward_sweep(C.byref(self._problem), self._ptr(self.fwd), self._ptr(self.xm),
                                         self._ptr(self.post), self._ptr(self.zpost), self._ptr(self.cell_stats),
                                         self._ptr(self.term_stats), self._ptr(self.status), self._stream())
        self._check(rc, "i2c_backward_sweep")

    def forward_backward(self):
        """I2cGraph._forward_backward_msgs (i2c.py:1231-1236)."""
        self.forward_sweep()
        self.backward_sweep()

    def riccati_sweep(self):
        """I2cGraph._backward_ricatti_msgs (i2c.py:888-893): Riccati-form backward messages after a Linearize
        forward/backward pass. Overwrites K, k, sigK with the Riccati-form controller (as the reference does) and
        returns the backward state message in information form, (nu_x0_b (B, T, nx), lambda_x0_b (B, T, nx, nx))."""
        if not self.linearize:
            raise RuntimeError("the Riccati messages belong to the Linearize() path (i2c.py:612-678)")
        if self.prior_out is None or self.xm is None:
            raise RuntimeError("riccati_sweep() needs keep_prior=True and keep_xm=True")
        nx = self.nx
        self.ric = torch.zeros(self.H, nx + nx * nx, self.B, dtype=self.dtype, device=self.device)
        rc = self.lib.i2c_riccati_sweep(C.byref(self._problem), self._ptr(self.prior_out), self._ptr(self.fwd),
                                        self._ptr(self.xm), self._ptr(self.post), self._ptr(self.ric),
                                        self._ptr(self.status), self._stream())
        self._check(rc, "i2c_riccati_sweep")
        nu = self.ric[:, :nx].permute(2, 0, 1)
        lam = self.ric[:, nx:].permute(2, 0, 1).reshape(self.B, self.H, nx, nx)
        return nu, lam

    def propagate(self, _stats=None):
        """I2cGraph.propagate (i2c.py:1247-1251). (_stats: a [3][B] row that receives the statistics instead of self.prop_stats)"""
        if self.prop is None:
            self.prop = torch.zeros(self.H, self.dims.e_prop, self.B, dtype=self.dtype, device=self.device)
            self.prop_stats = torch.zeros(3, self.B, dtype=self.dtype, device=self.device)
        rc = self.lib.i2c_propagate(C.byref(self._problem), self._ptr(self.post), self._ptr(self.prop),
                                    self._ptr(self.prop_stats if _stats is None else _stats), int(self.use_expert_controller),
                                    self._ptr(self.status), self._stream())
        self._check(rc, "i2c_propagate")

    def set_cell_expert(self, t, value):
        """cells[t].use_expert_controller = value (i2c.py:143): a per-cell flag like the reference's, allocated on first use
        (until then every cell follows `use_expert_controller`)."""
        if self.expert_cells is None:
            self.expert_cells = torch.full((self.H,), int(bool(self.use_expert_controller)), dtype=torch.uint8, device=self.device)
            self._problem.expert = self.expert_cells.data_ptr()
        self.expert_cells[(int(t) + self.t0) % self.H] = int(bool(value))

    def cell_expert(self, t):
        if self.expert_cells is None:
            return bool(self.use_expert_controller)
        return bool(self.expert_cells[(int(t) + self.t0) % self.H].item())

    def update_priors(self):
        """I2cGraph._update_priors (i2c.py:1210-1221). The data copy is free: the forward sweep reads
        the posterior buffer as its prior; only the feed-forward -> feedback flags change."""
        if self.tau > 0:
            n = min(self.tau + 1, self.H)
            if self.t0 == 0:
                self.feedforward[:n] = 0  # one fill, no index arithmetic on the device
            else:
                self.feedforward[self.ring_rows(n)] = 0
        if self.prior is not self.post:  # keep_prior_joint: the posterior becomes the prior, the old prior buffer is free
            self._post_spare, self.prior = self.prior, self.post

    def _alpha_from_propagation(self):
        """calculate_alpha(sum of propagated observation covariances) (i2c.py:901-904, 934-939)."""
        return self.prop_stats[0] / float(self.nz * self.H)

    def alpha_mstep(self, update_alpha=True):
        """compute_update_alpha (i2c.py:921-963) alone: alpha_hat from the backward sweep's statistics, the clamp, and --
        with update_alpha -- the new temperature in every cell (update_xi). No cost bookkeeping, no prior update.
        Returns (alpha_hat, alpha after the clamp) as (B,) tensors and appends them to alphas_desired / alphas."""
        rc = self.lib.i2c_mstep(C.byref(self._problem), self._ptr(self.term_stats), self.alpha_update_tol,
                                int(bool(update_alpha)), self._ptr(self.stats_out), self._stream())
        self._check(rc, "i2c_mstep")
        out = self.stats_out.clone()
        self.alphas_desired.append(out[0])
        self.alphas.append(out[1])
        if update_alpha:
            self._broadcast_alpha()
        return out[0], out[1]

    def record_costs(self):
        """I2cGraph.calc_cost (i2c.py:1045-1066) alone: the expected cost of the current posterior (and of the propagation) appended
        to the cost lists, from the statistics the last backward sweep / propagation left on the device. The temperature is not
        touched (the M-step kernel runs with update_alpha = 0)."""
        rc = self.lib.i2c_mstep(C.byref(self._problem), self._ptr(self.term_stats), self.alpha_update_tol, 0,
                                self._ptr(self.stats_out), self._stream())
        self._check(rc, "i2c_mstep")
        out = self.stats_out.clone()
        self.costs_m.append(out[2])
        self.costs_m_var.append(out[3])
        if self._propagate:
            ps = self.prop_stats.clone()
            self.costs_pf.append(ps[0])
            self.costs_pf_var.append(ps[1])
        else:
            self.costs_pf.append(torch.full_like(out[2], -1.0))  # i2c.py:1065

    def maximize(self, update_alpha=True):
        """I2cGraph._maximize (i2c.py:1004-1019): cost, prior update, temperature M-step."""
        rc = self.lib.i2c_mstep(C.byref(self._problem), self._ptr(self.term_stats), self.alpha_update_tol,
                                int(bool(update_alpha)), self._ptr(self.stats_out), self._stream())
        self._check(rc, "i2c_mstep")
        out = self.stats_out.clone()
        self.costs_m.append(out[2])
        self.costs_m_var.append(out[3])
        ps = None
        if self._propagate:
            ps = self.prop_stats.clone()  # ONE copy per iteration (cost mean, cost variance, terminal KL): the rows below are views
            self.costs_pf.append(ps[0])
            self.costs_pf_var.append(ps[1])
            self.alphas_pf.append(ps[0] / float(self.nz * self.H))  # = _alpha_from_propagation()
        else:
            if getattr(self, "_minus_one", None) is None:
                self._minus_one = torch.full_like(out[2], -1.0)  # (one constant row shared by every entry: no fill kernel per iteration)
            self.costs_pf.append(self._minus_one)  # i2c.py:1065
        self.update_priors()
        self.alphas_desired.append(out[0])
        self.alphas.append(out[1])
        if update_alpha:
            self._broadcast_alpha()
        if self.has_x_terminal:
            self.kl_terms.append(ps[2].to(torch.float64) if (ps is not None and self.prop is not None) else self._terminal_kl())

    def learn_msgs(self):
        """One EM iteration (I2cGraph.learn_msgs, i2c.py:1238-1245)."""
        self.em_iter += 1
        self.forward_backward()
        if self._propagate:
            self.propagate()
        self.maximize()

    def learn(self, n_iters):
        """n_iters EM iterations enqueued from C++ with no host round trip (i2c_learn). Same results as
        calling learn_msgs() n_iters times; available when closed-loop propagation is off."""
        n_iters = int(n_iters)
        # The fused loop updates only alpha[b]; with per-cell temperatures (MPC) every cell has to take the new alpha each
        # iteration (update_xi, i2c.py:961-981), which the stepwise path does through _broadcast_alpha(). Separate prior /
        # posterior buffers (keep_prior_joint) also need the host-side swap of update_priors().
        if (self._propagate and n_iters > 0 and self.alpha_cell is None and not self.keep_prior_joint and self.prior_out is None
                and not self.mixed and not self.uses_group_kernels and not self.linearize and not self.gauss_hermite):
            return self._learn_with_propagation(n_iters)
        if self._propagate or self.prior_out is not None or n_iters <= 0 or self.alpha_cell is not None or self.keep_prior_joint:
            for _ in range(n_iters):
                self.learn_msgs()
            return
        self._problem.expert_controller = int(bool(self.use_expert_controller))
        hist = torch.empty(n_iters, 4, self.B, dtype=self.dtype, device=self.device)
        rc = self.lib.i2c_learn(C.byref(self._problem), self._ptr(self.post), self._ptr(self.fwd), self._ptr(self.xm),
                                self._ptr(self.zpost), self._ptr(self.cell_stats), self._ptr(self.term_stats),
                                self.alpha_update_tol, int(self.tau), n_iters, self._ptr(hist), self._ptr(self.status),
                                self._stream())
        self._check(rc, "i2c_learn")
        self.em_iter += n_iters
        minus_one = torch.full((self.B,), -1.0, dtype=self.dtype, device=self.device)
        for it in range(n_iters):
            self.alphas_desired.append(hist[it, 0])
            self.alphas.append(hist[it, 1])
            self.costs_m.append(hist[it, 2])
            self.costs_m_var.append(hist[it, 3])
            self.costs_pf.append(minus_one)
            if self.has_x_terminal:
                self.kl_terms.append(self._terminal_kl())
        self._broadcast_alpha()

    def _learn_with_propagation(self, n_iters):
        """n x learn_msgs() with closed-loop propagation (covariance control, BASELINE config 5) enqueued by ONE library call
        (i2c_learn_propagate): forward, backward, propagate, M-step per iteration, the statistics of every iteration written
        straight into history rows. From the second iteration on the propagation of iteration k shares a launch with the
        forward sweep of iteration k + 1 (`overlap_propagation`; lane kernels of the d <= 5 models): the two chains of T dependent
        cells become one (pendulum T=100, B=8192: 0.27 -> 0.19 ms per iteration). The same numbers as the one-by-one calls."""
        if self.prop is None:
            self.prop = torch.zeros(self.H, self.dims.e_prop, self.B, dtype=self.dtype, device=self.device)
            self.prop_stats = torch.zeros(3, self.B, dtype=self.dtype, device=self.device)
        hist = torch.empty(n_iters, 4, self.B, dtype=self.dtype, device=self.device)   # alpha_hat, alpha, cost mean, cost variance
        histp = torch.empty(n_iters, 3, self.B, dtype=self.dtype, device=self.device)  # propagated cost mean, variance, terminal KL
        rc = self.lib.i2c_learn_propagate(C.byref(self._problem), self._ptr(self.post), self._ptr(self.fwd), self._ptr(self.xm),
                                          self._ptr(self.zpost), self._ptr(self.cell_stats), self._ptr(self.term_stats),
                                          self._ptr(self.prop), self._ptr(histp), self.alpha_update_tol, int(self.tau), n_iters,
                                          self._ptr(hist), int(self.use_expert_controller), int(self.overlap_propagation),
                                          self._ptr(self.status), self._stream())
        self._check(rc, "i2c_learn_propagate")
        self.em_iter += n_iters
        self.stats_out.copy_(hist[-1])
        self.prop_stats.copy_(histp[-1])
        apf = histp[:, 0] / float(self.nz * self.H)  # = _alpha_from_propagation() of every iteration
        for it in range(n_iters):
            self.costs_m.append(hist[it, 2])
            self.costs_m_var.append(hist[it, 3])
            self.costs_pf.append(histp[it, 0])
            self.costs_pf_var.append(histp[it, 1])
            self.alphas_pf.append(apf[it])
            self.alphas_desired.append(hist[it, 0])
            self.alphas.append(hist[it, 1])
            if self.has_x_terminal:
                self.kl_terms.append(histp[it, 2])
        self._broadcast_alpha()

    def calibrate_alpha(self, only_decrease=False):
        """I2cGraph.calibrate_alpha (i2c.py:895-911)."""
        assert self._propagate
        self.propagate()
        a = self._alpha_from_propagation()
        if only_decrease:
            a = torch.where(a < self.alpha, a, self.alpha)
        self.alpha.copy_(a)
        self._broadcast_alpha()
        self.alphas[-1] = self.alpha.clone()

    # ------------------------------------------------------------------ MPC building blocks
    def set_initial_state(self, mu, cov):
        """sys.x0 / sys.sig_x0 of every trajectory (mpc.py:149-150). mu (B, nx) or (nx,), cov (B, nx, nx) or (nx, nx)."""
        mu = np.broadcast_to(np.asarray(mu, np.float64).reshape(-1, self.nx), (self.B, self.nx))
        cov = np.broadcast_to(np.asarray(cov, np.float64), (self.B, self.nx, self.nx))
        self.x0.copy_(torch.as_tensor(np.array(mu.T, order="C"), dtype=self.dtype))
        self.sig_x0.copy_(torch.as_tensor(np.array(pack_sym_np(cov).T, order="C"), dtype=self.dtype))

    def ckf_filter(self, y, u, sig_zeta, mu=None, cov=None):
        """One cubature Kalman filter step on the belief (mu [nx][B], cov [sym nx][B]; default: the
        solver's x0 / sig_x0 tensors, updated in place) -- PartiallyObservedMpcPolicy.filter, mpc.py:125-145.
        y: [ny][B] tensor, u: [nu][B] tensor, sig_zeta: (ny, ny) array."""
        mu = self.x0 if mu is None else mu
        cov = self.sig_x0 if cov is None else cov
        ny = self.dims.ny
        assert y.shape == (ny, self.B) and u.shape == (self.nu, self.B) and y.dtype == self.dtype
        zeta = (C.c_double * sym_size(ny))(*pack_sym_np(np.asarray(sig_zeta, np.float64)).reshape(-1))
        y, u = y.contiguous(), u.contiguous()  # bound to locals: a temporary copy must outlive the launch
        rc = self.lib.i2c_ckf_filter(C.byref(self._problem), zeta, self._ptr(y), self._ptr(u),
                                     self._ptr(mu), self._ptr(cov), self._ptr(self.status), self._stream())
        self._check(rc, "i2c_ckf_filter")
        return mu, cov

    def enable_per_cell_alpha(self):
        """The reference's MPC appends `deepcopy(cell_init)`: that cell keeps the sig_xi it was copied with
        (the temperature at policy construction), because update_xi (i2c.py:976-981) only reaches the cells
        in the list when alpha changes and alpha is frozen inside the loop. Reproduced with a [T][B]
        per-cell temperature that follows the cells through the shift."""
        if self.alpha_cell is None:
            self.alpha_init = self.alpha.clone()
            self.alpha_cell = self.alpha.reshape(1, -1).repeat(self.H, 1).contiguous()
            self.refresh_problem()

    def _broadcast_alpha(self):
        """update_xi (i2c.py:976-981): every cell currently in the chain takes the graph's alpha."""
        if self.alpha_cell is not None:
            self.alpha_cell.copy_(self.alpha.reshape(1, -1).expand(self.H, -1))

    def ring_rows(self, n=None):
        """Physical rows of cells 0..n-1 (default: the whole horizon) in the ring buffers, as an index tensor."""
        n = self.H if n is None else n
        return (torch.arange(n, device=self.device) + self.t0) % self.H

    def cells(self, buf):
        """A ring buffer ([T][...]) in cell order (a view when the ring has not moved)."""
        return buf if self.t0 == 0 else torch.roll(buf, -self.t0, 0)

    def _advance_ring(self):
        self.t0 = (self.t0 + 1) % self.H
        if self.terminal_cell >= 0:
            self.terminal_cell -= 1
        self._problem.t0 = int(self.t0)
        self._problem.terminal_cell = int(self.terminal_cell)

    def shift_horizon(self, z_new=None, want_action=False):
        """Receding horizon (mpc.py:174-181): drop cell 0, append a fresh feed-forward cell whose target is z_new ([nz][B]
        tensor) or, if None, the previous last cell's. The per-cell buffers are a ring: the fresh cell is written over the
        row of the dropped one (i2c_shift_horizon) and the ring offset advances -- nothing else moves. The `terminal_cell`
        flag stays with the cell it was set on (i2c.py:822), so its index decreases. Returns the dropped cell's action
        moments (mu_u (B, nu), sig_u packed (B, sym nu)) if want_action."""
        assert self.prior is self.post, "shift_horizon() between a sweep and its update_priors()"
        if want_action and getattr(self, "_mpc_action", None) is None:
            self._mpc_action = torch.empty(self.nu + sym_size(self.nu), self.B, dtype=self.dtype, device=self.device)
        if z_new is not None:
            z_new = z_new.contiguous()
        rc = self.lib.i2c_shift_horizon(C.byref(self._problem), self._ptr(self.post), self._ptr(self.cell_init),
                                        self._ptr(self.alpha_init), self._ptr(z_new),
                                        self._ptr(self._mpc_action) if want_action else None, self._stream())
        self._check(rc, "i2c_shift_horizon")
        self._advance_ring()
        if want_action:
            return self._mpc_action[: self.nu].T, self._mpc_action[self.nu:].T

    def mpc_step(self, n_iter, y=None, u=None, sig_zeta=None, z_new=None):
        """One control step of the MPC loop in one library call (i2c_mpc_step): optional filter step on the belief, n_iter
        sweeps, first action, receding-horizon shift -- no host round trip, no torch ops, no copy of the horizon (the per-cell
        buffers are a ring). Returns (mu_u (B, nu), sig_u packed (B, sym nu)) device tensors: the first planned action BEFORE
        the shift (cells[0].mu_u0_m, sig_u0_m). Same numbers as ckf_filter + n_iter x (forward_backward, update_priors) +
        shift_horizon."""
        assert self.prior is self.post, "mpc_step() between a sweep and its update_priors()"
        st = _native.I2cMpcStep()
        st.do_filter = int(y is not None)
        st.n_iter, st.tau = int(n_iter), int(self.tau)
        if y is not None:
            ny = self.dims.ny
            assert y.shape == (ny, self.B) and u.shape == (self.nu, self.B) and y.dtype == self.dtype
            for i, v in enumerate(pack_sym_np(np.asarray(sig_zeta, np.float64)).reshape(-1)):
                st.sig_zeta[i] = float(v)
            y, u = y.contiguous(), u.contiguous()
            st.y, st.u = y.data_ptr(), u.data_ptr()
        if getattr(self, "_mpc_action", None) is None:
            self._mpc_action = torch.empty(self.nu + sym_size(self.nu), self.B, dtype=self.dtype, device=self.device)
        st.post, st.fwd = self.post.data_ptr(), self.fwd.data_ptr()
        opt = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        st.xm, st.zpost, st.cell_stats, st.term_stats = opt(self.xm), opt(self.zpost), opt(self.cell_stats), self.term_stats.data_ptr()
        st.cell_init, st.alpha_init = self.cell_init.data_ptr(), opt(self.alpha_init)
        if z_new is not None:
            z_new = z_new.contiguous()
        st.z_new = opt(z_new)
        st.action, st.status = self._mpc_action.data_ptr(), self.status.data_ptr()
        self._problem.expert_controller = int(bool(self.use_expert_controller))
        rc = self.lib.i2c_mpc_step(C.byref(self._problem), C.byref(st), self._stream())
        self._check(rc, "i2c_mpc_step")
        self._advance_ring()
        return self._mpc_action[: self.nu].T, self._mpc_action[self.nu:].T

    def set_targets(self, z_traj):
        """Per-cell targets (mpc.py:29-31): z_traj (T, nz) or (B, T, nz)."""
        z = np.broadcast_to(np.asarray(z_traj, np.float64), (self.B, self.H, self.nz))
        zt = torch.as_tensor(np.array(np.transpose(z, (1, 2, 0)), order="C"), dtype=self.dtype, device=self.device)
        if self.t0:
            zt = torch.roll(zt, self.t0, 0)  # cell order -> ring rows
        if self.z is None:
            self.z = zt.contiguous()
            self.refresh_problem()
        else:
            self.z.copy_(zt)

    def rollout(self, n_rollouts=1, policy="linear", process_noise=True, action_noise=False, sample_x0=False,
                generator=None, want=("xu", "z", "x_final", "z_term"), eps_x0=None, eps_x=None, eps_u=None):
        """Simulate the current controllers through the noisy model (env.batch_eval, i2c/env.py:93-103):
        n_rollouts per trajectory, all B * n_rollouts in one launch. Returns a dict of (R, B, T, ...) tensors.
        The standard-normal disturbances are drawn here (torch.randn) unless given: eps_x0 [nx][N], eps_x [T][nx][N],
        eps_u [T][nu][N] with N = n_rollouts * B, rollout n = r * B + b."""
        N, T, dev, dt = int(n_rollouts) * self.B, self.H, self.device, self.dtype
        code = {"linear": 0, "expert": 1, "expert_soft": 1, "expert_hard": 2}[policy]
        rnd = lambda *s: torch.randn(*s, dtype=dt, device=dev, generator=generator)  # noqa: E731
        given = lambda t, shape: torch.as_tensor(t, dtype=dt, device=dev).reshape(shape).contiguous()  # noqa: E731
        eps_x0 = given(eps_x0, (self.nx, N)) if eps_x0 is not None else (rnd(self.nx, N) if sample_x0 else None)
        eps_x = given(eps_x, (T, self.nx, N)) if eps_x is not None else (rnd(T, self.nx, N) if process_noise else None)
        eps_u = given(eps_u, (T, self.nu, N)) if eps_u is not None else (rnd(T, self.nu, N) if action_noise else None)
        out = {
            "xu": torch.empty(T, self.d, N, dtype=dt, device=dev) if "xu" in want else None,
            "z": torch.empty(T, self.nz, N, dtype=dt, device=dev) if "z" in want else None,
            "x_final": torch.empty(self.nx, N, dtype=dt, device=dev) if "x_final" in want else None,
            "z_term": torch.empty(self.nzt, N, dtype=dt, device=dev) if ("z_term" in want and self.nzt > 0) else None,
        }
        rc = self.lib.i2c_rollout(C.byref(self._problem), self._ptr(self.post), int(n_rollouts), code, self._ptr(eps_x0),
                                  self._ptr(eps_x), self._ptr(eps_u), self._ptr(out["xu"]), self._ptr(out["z"]),
                                  self._ptr(out["x_final"]), self._ptr(out["z_term"]), self._stream())
        self._check(rc, "i2c_rollout")
        R_, B = int(n_rollouts), self.B
        res = {"eps_x0": eps_x0, "eps_x": eps_x, "eps_u": eps_u}
        for k, v in out.items():
            if v is None:
                res[k] = None
            elif v.dim() == 3:
                res[k] = v.reshape(T, v.shape[1], R_, B).permute(2, 3, 0, 1)  # (R, B, T, n)
            else:
                res[k] = v.reshape(v.shape[0], R_, B).permute(1, 2, 0)  # (R, B, n)
        return res

    def _terminal_kl(self):
        """mvn_kl_divergence(x3_pf[T-1] || terminal prior) (i2c.py:1012-1019, 1223-1229)."""
        nx = self.nx
        if self.prop is not None and self._propagate:
            return self.prop_stats[2].to(torch.float64).clone()  # computed by k_propagate for the state it just propagated
        if self.prop is None:
            mu1 = self.x0.T.to(torch.float64)
            sig1 = unpack_sym(self.sig_x0.T.to(torch.float64), nx)
        else:
            o = self.d + sym_size(self.d)
            mu1 = self.prop[-1, o: o + nx, :].T.to(torch.float64)
            sig1 = unpack_sym(self.prop[-1, o + nx: o + nx + sym_size(nx), :].T.to(torch.float64), nx)
        mu2 = torch.as_tensor(self.mu_x_terminal, dtype=torch.float64, device=self.device)
        sig2 = torch.as_tensor(self.sig_x_terminal, dtype=torch.float64, device=self.device)
        diff = mu2 - mu1
        dist = (diff * torch.linalg.solve(sig2, diff.T).T).sum(-1)
        log_det_ratio = torch.log(torch.linalg.det(sig2) / torch.linalg.det(sig1))
        trace_ratio = torch.diagonal(torch.linalg.solve(sig2, sig1), dim1=-2, dim2=-1).sum(-1)
        return 0.5 * (log_det_ratio + trace_ratio + dist - nx)

    # ------------------------------------------------------------------ failure reporting
    def failures(self):
        """Per-trajectory status words -> list of (b, reason, t) for failed trajectories."""
        st = self.status.cpu().numpy()
        return [(int(b), int(s) >> 16, (int(s) & 0xFFFF) - 1) for b, s in enumerate(st) if s != 0]

    def raise_on_failure(self, status=None):
        """With B == 1 behave like the reference: raise; batched callers read `failures()`."""
        f = self.failures() if status is None else [(int(b), int(s) >> 16, (int(s) & 0xFFFF) - 1) for b, s in enumerate(status) if s != 0]
        if f:
            b, reason, t = f[0]
            raise I2cNumericalError(f"trajectory {b}, cell {t}: {_native.FAIL_REASONS.get(reason, reason)}")

    def iteration_health(self):
        """(status (B,) int32, alpha_hat of the last M-step (B,) float64) on the host after ONE stream synchronisation: what a
        single-trajectory caller checks after every EM iteration (the reference raises inside the iteration: LinAlgError from a
        Cholesky, "Alpha is NaN" from update_alpha, i2c.py:948-951). Two asynchronous copies into page-locked host memory."""
        a = self.alphas_desired[-1]
        if self.device.type != "cuda":
            return self.status.numpy().copy(), a.to(torch.float64).numpy().copy()
        if getattr(self, "_health", None) is None:
            self._health = (torch.empty(self.B, dtype=torch.int32).pin_memory(), torch.empty(self.B, dtype=self.dtype).pin_memory())
        hs, ha = self._health
        hs.copy_(self.status, non_blocking=True)
        ha.copy_(a, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return hs.numpy(), ha.to(torch.float64).numpy()

    # ------------------------------------------------------------------ getters: (B, T, ...) tensors
    def _rows(self, buf, lo, n):
        if buf is self.post or buf is self.prior:  # ring buffers: back to cell order
            buf = self.cells(buf)
        return buf[:, lo: lo + n, :].permute(2, 0, 1)  # (B, T, n)

    def _sym_rows(self, buf, lo, n):
        return unpack_sym(self._rows(buf, lo, sym_size(n)), n)

    def marginal_state_action(self):
        """(mu_xu0_m (B,T,d), sig_xu0_m (B,T,d,d)): i2c.py:1300-1304."""
        return self._rows(self.post, 0, self.d), self._sym_rows(self.post, self.d, self.d)

    def local_linear_policy(self):
        """(K (B,T,nu,nx), k (B,T,nu), sigK (B,T,nu,nu)): i2c.py:1253-1264."""
        o = self.d + sym_size(self.d)
        K = self._rows(self.post, o, self.nu * self.nx).reshape(self.B, self.H, self.nu, self.nx)
        k = self._rows(self.post, o + self.nu * self.nx, self.nu)
        sigK = self._sym_rows(self.post, o + self.nu * self.nx + self.nu, self.nu)
        return K, k, sigK

    def forward_messages(self):
        d, nx = self.d, self.nx
        o = d + sym_size(d)
        fwd = self.fwd.permute(0, 2, 1) if self.fwd_trajectory_major else self.fwd  # [T][e_fwd][B] either way
        return dict(
            mu_xu1_f=self._rows(fwd, 0, d),
            sig_xu1_f=self._sym_rows(fwd, d, d),
            mu_x3_f=self._rows(fwd, o, nx),
            sig_x3_f=self._sym_rows(fwd, o + nx, nx),
            J_dyn=self._rows(fwd, o + nx + sym_size(nx), d * nx).reshape(self.B, self.H, d, nx),
        )

    def smoothed_next_state(self):
        assert self.xm is not None, "constructed with keep_xm=False"
        return self._rows(self.xm, 0, self.nx), self._sym_rows(self.xm, self.nx, self.nx)

    def observed_marginal(self):
        """(mu_z0_m (B,T,nz), sig_z0_m (B,T,nz,nz)) (i2c.py:594-596, 1312-1314)."""
        assert self.zpost is not None, "constructed with keep_zpost=False"
        return self._rows(self.zpost, 0, self.nz), self._sym_rows(self.zpost, self.nz, self.nz)

    def terminal_observed_marginal(self):
        if not self.has_Qf:
            return None, None
        nzt = self.nzt
        return self.term_stats[3: 3 + nzt].T, unpack_sym(self.term_stats[3 + nzt: 3 + nzt + sym_size(nzt)].T, nzt)

    def prior_state_action(self):
        assert self.prior_out is not None, "constructed with keep_prior=False"
        return self._rows(self.prior_out, 0, self.d), self._sym_rows(self.prior_out, self.d, self.d)

    def propagated(self):
        d, nx = self.d, self.nx
        o = d + sym_size(d)
        return dict(
            mu_xu0_pf=self._rows(self.prop, 0, d),
            sig_xu0_pf=self._sym_rows(self.prop, d, d),
            mu_x3_pf=self._rows(self.prop, o, nx),
            sig_x3_pf=self._sym_rows(self.prop, o + nx, nx),
        )

    # ------------------------------------------------------------------ history helpers
    @staticmethod
    def history(lst):
        """list of (B,) device tensors -> (n, B) float64 numpy (one host sync)."""
        if not lst:
            return np.zeros((0, 0))
        return torch.stack([x.to(torch.float64) for x in lst]).cpu().numpy()
