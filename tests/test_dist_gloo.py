"""N > 1 path on CPU: two gloo ranks each run the EM loop on their shard of the batch (host
simulation of the kernels), then perform the job's single collective (all-gather of the final
controllers). The gathered result must equal the single-process full-batch run bit for bit --
trajectories are independent, so sharding must not change anything."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hostsim
import parity
from golden_util import load_case


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, iters, out_dir, golden="em_pendulum_T40_quad_general"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("input-inference-for-control_amd")
        g = load_case(golden)
        x0, mu_u = parity.batched_inputs(g, B)
        lo, hi = pkg.dist.shard_range(B, rank, world)
        eng = parity.engine_from_case(g, hostsim.load(), "cpu", x0=x0[lo:hi], mu_u=mu_u[lo:hi])
        for _ in range(iters):
            eng.learn_msgs()
        # count the collectives of the job's one exchange (round-4 review, weak #10: it used to be two -- sizes, then the payload)
        calls = []
        real = {n: getattr(dist, n) for n in ("all_gather", "all_gather_into_tensor", "all_reduce", "broadcast", "all_to_all", "gather")}
        for n, fn in real.items():
            setattr(dist, n, (lambda n_, fn_: lambda *a, **k: (calls.append(n_), fn_(*a, **k))[1])(n, fn))
        try:
            got = pkg.dist.gather_policy(eng, total=B)
        finally:
            for n, fn in real.items():
                setattr(dist, n, fn)
        assert calls == ["all_gather_into_tensor"], calls
        assert got["K"].shape[0] == B
        # without the global size a multi-rank gather is refused BEFORE any collective (round-5 advice: ragged shards without
        # `total` used to enter all_gather_into_tensor with different buffer sizes per rank)
        try:
            pkg.dist.gather_policy(eng)
            raise AssertionError("a multi-rank gather without the global batch size was accepted")
        except ValueError as e:
            assert "total" in str(e)
        if B % world:  # ragged shards need the global size: a silent mis-slice is refused
            try:
                pkg.dist._all_gather_rows(torch.zeros(hi - lo + 1, 2), total=B)
                raise AssertionError("a shard of the wrong size was accepted")
            except ValueError:
                pass
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"), **{k: v.numpy() for k, v in got.items()})
        # every rank holds the same gathered result
        chk = got["k"].sum().reshape(1).clone()
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        assert all(torch.equal(lst[0], x) for x in lst)
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_batch():
    pkg = importlib.import_module("input-inference-for-control_amd")
    for total, world in [(4096, 8), (10, 3), (7, 8), (65536, 8)]:
        spans = [pkg.dist.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("golden,B", [("em_pendulum_T40_quad_general", 11), ("em_quad12_T20", 5)])
def test_two_rank_gloo_matches_single_process(tmp_path, golden, B):
    """(the 12-state quadrotor: wave kernels, trajectory-major posterior storage behind the same views)"""
    hostsim.build()  # build once in the parent so the workers do not race
    iters, world = 3, 2  # ragged shards: 6 + 5, 3 + 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, iters, str(tmp_path), golden), nprocs=world, join=True)
    got = np.load(os.path.join(tmp_path, "gathered.npz"))

    pkg = importlib.import_module("input-inference-for-control_amd")
    g = load_case(golden)
    x0, mu_u = parity.batched_inputs(g, B)
    eng = parity.engine_from_case(g, hostsim.load(), "cpu", x0=x0, mu_u=mu_u)
    for _ in range(iters):
        eng.learn_msgs()
    ref = pkg.dist.gather_policy(eng)  # world size 1: no collective, same packing
    for k in ("K", "k", "sigK", "cost", "alpha", "status"):
        assert np.array_equal(got[k], ref[k].numpy()), k
