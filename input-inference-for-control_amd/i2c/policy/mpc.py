"""i2c model-predictive controllers with the reference's class names and call protocol
(reference i2c/policy/mpc.py:16-182), batched: one policy drives B closed loops at once.

Every step of the loop is device work: the cubature Kalman filter (`i2c_ckf_filter`), `n_iter`
forward/backward sweeps warm-started from the shifted posterior, and the receding-horizon shift
(the reference's `cells.pop(0); cells.append(deepcopy(cell_init))` list surgery becomes a roll of
the [T][E][B] buffers plus a fresh last row). With B == 1 shapes are the reference's
((n, 1) columns); with B > 1 arrays carry a leading batch axis.
"""
import numpy as np
import torch


def _np(t):
    return np.array(t.detach().to(torch.float64).cpu().numpy(), copy=True)  # never alias a device/host buffer


class _LazyList(list):
    """A history list whose entries may be zero-argument callables -- device-side snapshots taken inside the control loop -- that are
    replaced by their host value when first READ: the single closed loop (the reference's shape) keeps the reference's histories
    (mus, covars, xu_history, z_history: mpc.py:165-170, plots) without a device -> host copy per entry and step."""

    def _at(self, k):
        v = list.__getitem__(self, k)
        if callable(v):
            v = v()
            list.__setitem__(self, k, v)
        return v

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._at(k) for k in range(*i.indices(len(self)))]
        return self._at(i if i >= 0 else len(self) + i)

    def __iter__(self):
        return (self._at(k) for k in range(len(self)))


class MpcPolicy:
    """Fully observed MPC (reference mpc.py:16-111). NB the reference's own __call__ is broken
    (it passes unknown kwargs to compute_update_alpha); this one runs."""

    def plot_history(self, res_path, name=""):
        """Figures are presentation, not part of the solver (I2cGraph._no_plot): the planned trajectories stay in xu_history."""
        return None

    def __init__(self, i2c, n_iter, sig_u, z_traj=None):
        self.i2c = i2c
        self.engine = e = i2c.engine
        self.B = e.B
        self.dim_u, self.dim_x = e.nu, e.nx
        self.sig_u = sig_u
        self.model = i2c.sys
        self.n_iter = n_iter
        self.set_control(True)  # mpc.py:21-22
        e.enable_per_cell_alpha()  # cell_init is copied NOW (mpc.py:26): appended cells carry today's sig_xi ...
        e.cell_init = e.post[e.t0].clone()  # ... and today's cells[0] (deepcopy(i2c.cells[0]), mpc.py:26), not the constructor's
        self.z_traj = None if z_traj is None else np.asarray(z_traj, dtype=float)
        if self.z_traj is not None:
            zt = self.z_traj if self.z_traj.ndim == 3 else self.z_traj[None]
            e.set_targets(zt[:, : e.H])
        self._init_state = {k: getattr(e, k).clone() for k in ("post", "alpha", "alpha_cell", "feedforward", "x0", "sig_x0")}
        self._init_terminal = e.terminal_cell
        self._init_t0 = e.t0  # the snapshot is in physical row order: it is only valid with the ring where it was taken
        self._init_z = None if e.z is None else e.z.clone()
        self.record_history = self.B == 1
        self.xu_history, self.z_history = _LazyList(), _LazyList()

    def set_control(self, feedforward):
        """mpc.py:35-41: feed-forward keeps every cell's action prior independent of the state."""
        e = self.engine
        if feedforward:
            e.tau = 0
        else:
            e.tau = e.H

    def reset(self):
        e = self.engine
        if e.prior is not e.post:  # separate prior / posterior buffers: fold them before restoring
            e._post_spare, e.prior = e.prior, e.post
        for k, v in self._init_state.items():
            getattr(e, k).copy_(v)
        if self._init_z is not None:
            e.z.copy_(self._init_z)
        e.status.zero_()
        e.terminal_cell = self._init_terminal
        e.t0 = self._init_t0  # rows were cloned in physical order, at this offset of the ring
        e._problem.terminal_cell, e._problem.t0 = int(e.terminal_cell), int(e.t0)
        self.xu_history, self.z_history = _LazyList(), _LazyList()
        self._belief_host = None

    def _squeeze(self, a, column=False):
        if self.B == 1:
            a = a[0]
            return a.reshape(-1, 1) if column else a
        return a

    def _next_target(self, i):
        """Target of the cell that enters the horizon at step i (mpc.py:73-77, 177-181)."""
        if self.z_traj is None:
            return None
        e = self.engine
        zt = self.z_traj if self.z_traj.ndim == 3 else self.z_traj[None]
        if i + e.H < zt.shape[1]:
            z = np.broadcast_to(zt[:, i + e.H], (self.B, e.nz))
            return torch.as_tensor(np.array(z.T, order="C"), dtype=e.dtype, device=e.device)
        return None  # keep the previous last cell's target

    def _plan(self, n_iter):
        e = self.engine
        for _ in range(n_iter):
            e.forward_backward()
            e.update_priors()
        if self.B == 1:
            e.raise_on_failure()
        self.i2c._invalidate()

    def _first_action(self, deterministic):
        """cells[0].mu_u0_m (and sig_u0_m when sampling): only cell 0 of the posterior buffer is read back."""
        e = self.engine
        d, nx = e.d, e.nx
        mu_u = _np(e.post[e.t0, nx:d, :].T)  # (B, nu); row t0 of the ring is cell 0
        if not deterministic:
            from .. import core

            sig = core.engine.unpack_sym(e.post[e.t0, d: d + d * (d + 1) // 2, :].T, d)  # (B, d, d), cell 0 only
            sig_u = _np(sig[:, nx:, nx:])
            mu_u = np.stack([np.random.multivariate_normal(m, s) for m, s in zip(mu_u, sig_u)])
        return mu_u

    def _record(self):
        """xu_history / z_history of this step's plan (mpc.py:165-170): device-side copies now, host arrays when read (_LazyList)."""
        if self.record_history:
            e = self.engine
            mu = e._rows(e.post, 0, e.d).clone()  # (B, T, d) in cell order (a copy: the ring row of cell 0 is about to be reused)
            self.xu_history.append(lambda mu=mu: (lambda m: m[0][:, :, None] if self.B == 1 else m)(_np(mu)))
            if e.zpost is not None:
                mz = e._rows(e.zpost, 0, e.nz).clone()
                self.z_history.append(lambda mz=mz: (lambda m: m[0] if self.B == 1 else m)(_np(mz)))
            else:
                self.z_history.append(None)

    def optimize(self, n_iter, x):
        x = np.asarray(x, dtype=float)
        assert x.reshape(-1, self.dim_x).shape[0] in (1, self.B), f"{x.shape}, {(self.dim_x, 1)}"
        e = self.engine
        e.set_initial_state(x.reshape(-1, self.dim_x), _np(unpack(e)))
        if self.B == 1:
            self.i2c.sys.x0 = x.reshape(self.dim_x, 1)
            self.i2c._x0_seen = None
        self._plan(n_iter)

    def _sample_or_mean(self, mu_u, sig_u_packed, deterministic):
        mu_u = _np(mu_u)
        if not deterministic:
            from .. import core

            sig_u = _np(core.engine.unpack_sym(sig_u_packed, self.dim_u))
            mu_u = np.stack([np.random.multivariate_normal(m, s) for m, s in zip(mu_u, sig_u)])
        return mu_u

    def __call__(self, i, x, deterministic=True):
        if not self.record_history:  # batched closed loops: the whole control step is one library call
            x = np.asarray(x, dtype=float)
            e = self.engine
            e.set_initial_state(x.reshape(-1, self.dim_x), _np(unpack(e)))
            mu_u, sig_u = e.mpc_step(self.n_iter, z_new=self._next_target(i))
            self.i2c._invalidate()
            return self._squeeze(self._sample_or_mean(mu_u, sig_u, deterministic), column=True)
        self.optimize(self.n_iter, x)
        self._record()
        u = self._first_action(deterministic)
        self.engine.shift_horizon(self._next_target(i))
        self.i2c._invalidate()
        return self._squeeze(u, column=True)

    def update_models(self, sys):
        raise NotImplementedError("models are compiled device functors; construct a new I2cGraph")


def unpack(e):
    from .. import core

    return core.engine.unpack_sym(e.sig_x0.T, e.nx)


class PartiallyObservedMpcPolicy(MpcPolicy):
    """MPC on a cubature-Kalman-filtered belief (reference mpc.py:113-182)."""

    def __init__(self, i2c, n_iter, sig_u, z_traj=None):
        super().__init__(i2c, n_iter, sig_u, z_traj)
        e = self.engine
        # the belief lives in the solver's x0 / sig_x0 tensors: filter() updates them in place and the
        # next forward sweep starts from them, with no host round trip
        self.mus, self.covars = _LazyList(), _LazyList()
        self._belief_host = None  # (mu, covar) as host arrays while they are known to equal the device belief

    # -- belief as reference-shaped arrays ----------------------------------------------------
    @property
    def mu(self):
        if self._belief_host is not None:
            return np.array(self._belief_host[0], copy=True)
        return self._squeeze(_np(self.engine.x0.T), column=True)

    @property
    def covar(self):
        if self._belief_host is not None:
            return np.array(self._belief_host[1], copy=True)
        return self._squeeze(_np(unpack(self.engine)))

    def _dev(self, a, n):
        e = self.engine
        a = np.broadcast_to(np.asarray(a, dtype=float).reshape(-1, n), (self.B, n))
        return torch.as_tensor(np.array(a.T, order="C"), dtype=e.dtype, device=e.device)

    # The per-step host <-> device traffic of the batched loop: the measurement and the applied action go up in ONE copy from a
    # page-locked staging buffer, the first planned action comes down into one (a pageable numpy array costs a staging copy and a
    # synchronisation per transfer: three of them were a sixth of a planar-quadrotor control step at B = 1024)
    def _stage_yu(self, y, u):
        e = self.engine
        ny, nu = e.dims.ny, e.nu
        if getattr(self, "_yu_host", None) is None:
            self._yu_host = torch.empty(ny + nu, self.B, dtype=e.dtype, pin_memory=e.device.type == "cuda")
            self._yu_np = self._yu_host.numpy()
            self._yu_dev = torch.empty(ny + nu, self.B, dtype=e.dtype, device=e.device)
        self._yu_np[:ny] = np.broadcast_to(np.asarray(y, dtype=float).reshape(-1, ny), (self.B, ny)).T
        self._yu_np[ny:] = np.broadcast_to(np.asarray(u, dtype=float).reshape(-1, nu), (self.B, nu)).T
        self._yu_dev.copy_(self._yu_host, non_blocking=True)  # (the staging buffer is reused only after this step's action has come back)
        return self._yu_dev[:ny], self._yu_dev[ny:]

    def _read_action(self):
        e = self.engine
        if getattr(self, "_act_host", None) is None:
            self._act_host = torch.empty(e.nu, self.B, dtype=e.dtype, pin_memory=e.device.type == "cuda")
        self._act_host.copy_(e._mpc_action[: e.nu])  # (contiguous rows; blocks until the step has run)
        return np.array(self._act_host.numpy().T, dtype=float)

    def filter(self, y, u):
        """mpc.py:125-145. y: (dim_y, 1) or (B, dim_y); u: (dim_u, 1) or (B, dim_u)."""
        e = self.engine
        sig_zeta = self.i2c.sys.sig_zeta
        if sig_zeta is None:
            raise ValueError("sys.sig_zeta (measurement noise) must be set before filtering")
        self._belief_host = None
        e.ckf_filter(self._dev(y, e.dims.ny), self._dev(u, e.nu), sig_zeta)
        if self.B == 1:
            e.raise_on_failure()
        return self.mu, self.covar

    def optimize(self, n_iter, mu=None, covar=None):
        """mpc.py:147-154. With no arguments the current (filtered) belief is used as is."""
        e = self.engine
        if mu is not None:
            mu = np.asarray(mu, dtype=float)
            assert mu.reshape(-1, self.dim_x).shape[0] in (1, self.B), f"{mu.shape}, {(self.dim_x, 1)}"
            self._belief_host = None
            e.set_initial_state(mu.reshape(-1, self.dim_x), covar)
        if self.B == 1:  # keep the facade's view of sys.x0 / sig_x0 consistent (mpc.py:149-150)
            self.i2c.sys.x0, self.i2c.sys.sig_x0 = self.mu, self.covar
            self.i2c._x0_seen = (np.asarray(self.i2c.sys.x0, float).reshape(-1).tobytes(),
                                 np.asarray(self.i2c.sys.sig_x0, float).tobytes())
        self._plan(n_iter)

    def __call__(self, i, y, u, deterministic=True):
        if not self.record_history:  # batched closed loops: filter + plan + first action + shift in one library call
            e = self.engine
            if i > 0:
                sig_zeta = self.i2c.sys.sig_zeta
                if sig_zeta is None:
                    raise ValueError("sys.sig_zeta (measurement noise) must be set before filtering")
                yd, ud = self._stage_yu(y, u)
                mu_u, sig_u = e.mpc_step(self.n_iter, yd, ud, sig_zeta, z_new=self._next_target(i))
            else:
                mu_u, sig_u = e.mpc_step(self.n_iter, z_new=self._next_target(i))
            self.i2c._invalidate()
            if deterministic:
                return self._squeeze(self._read_action(), column=True)
            return self._squeeze(self._sample_or_mean(mu_u, sig_u, deterministic), column=True)
        return self._single_loop_step(i, y, u, deterministic)

    def _single_loop_step(self, i, y, u, deterministic):
        """One closed loop (B = 1, the reference's shape; mpc.py:156-182): filter, plan, record, first action, shift -- the same
        library calls as the separate methods, enqueued without a host round trip in between and settled by ONE synchronisation:
        the failure status (the reference raises inside the step), the first action and the filtered belief come back together
        through page-locked buffers; the histories keep device-side snapshots until they are read (_LazyList). (Round 6: the
        step used to make a dozen blocking device -> host copies -- 0.86 ms per step on MI355X, most of it waiting.)"""
        e = self.engine
        d, nx, nu = e.d, e.nx, e.nu
        if i > 0:
            sig_zeta = self.i2c.sys.sig_zeta
            if sig_zeta is None:
                raise ValueError("sys.sig_zeta (measurement noise) must be set before filtering")
            yd, ud = self._stage_yu(y, u)
            e.ckf_filter(yd, ud, sig_zeta)
        cuda = e.device.type == "cuda"
        if getattr(self, "_step_host", None) is None:
            pin = dict(pin_memory=True) if cuda else {}
            self._step_host = (torch.empty(nx, self.B, dtype=e.dtype, **pin), torch.empty(nx * (nx + 1) // 2, self.B, dtype=e.dtype, **pin),
                               torch.empty(d + d * (d + 1) // 2, self.B, dtype=e.dtype, **pin), torch.empty(self.B, dtype=torch.int32, **pin))
        h_mu, h_cov, h_cell0, h_status = self._step_host
        h_mu.copy_(e.x0, non_blocking=True)      # the belief the plan starts from (planning does not change it)
        h_cov.copy_(e.sig_x0, non_blocking=True)
        for _ in range(self.n_iter):
            e.forward_backward()
            e.update_priors()
        self._record()
        h_cell0.copy_(e.post[e.t0, : d + d * (d + 1) // 2, :], non_blocking=True)  # cells[0]: mu_xu0_m, sig_xu0_m (row t0 of the ring)
        h_status.copy_(e.status, non_blocking=True)
        if cuda:
            torch.cuda.current_stream(e.device).synchronize()
        e.raise_on_failure(h_status.numpy())
        from .. import core

        mu_b = self._squeeze(np.array(h_mu.numpy().T, dtype=float), column=True)
        cov_b = self._squeeze(core.engine.unpack_sym(h_cov.T.clone(), nx).numpy().astype(float))
        self._belief_host = (mu_b, cov_b)
        self.mus.append(np.array(mu_b, copy=True))
        self.covars.append(np.array(cov_b, copy=True))
        # keep the facade's view of sys.x0 / sig_x0 consistent (mpc.py:149-150)
        self.i2c.sys.x0, self.i2c.sys.sig_x0 = self.mu, self.covar
        self.i2c._x0_seen = (np.asarray(self.i2c.sys.x0, float).reshape(-1).tobytes(), np.asarray(self.i2c.sys.sig_x0, float).tobytes())
        cell0 = np.array(h_cell0.numpy().T, dtype=float)  # (B, d + sym d)
        ctrl = cell0[:, nx:d]
        if not deterministic:
            sig = core.engine.unpack_sym(torch.as_tensor(cell0[:, d:]), d).numpy()[:, nx:, nx:]
            ctrl = np.stack([np.random.multivariate_normal(m, s_) for m, s_ in zip(ctrl, sig)])
        e.shift_horizon(self._next_target(i))
        self.i2c._invalidate()
        return self._squeeze(ctrl, column=True)
