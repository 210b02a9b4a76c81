"""Multi-GPU: the batch axis shards embarrassingly (trajectories are independent).

One process per GPU (torchrun); every rank runs the full EM loop on its slice with NO
collective inside the loop. The job's single exchange is one all-gather of the final
controllers / costs / temperatures / status words (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" for the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` trajectories owned by `rank` (remainder spread over the
    first ranks, so shards differ by at most one trajectory)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_rows(t, group=None):
    """All-gather along dim 0 allowing ragged shard sizes."""
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        out = t.new_empty((world * sizes[0],) + tuple(t.shape[1:]))
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
        return out
    mx = max(sizes)
    pad = t.new_zeros((mx,) + tuple(t.shape[1:]))
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def gather_policy(engine, group=None):
    """The one collective: every rank ends up with the whole batch's time-varying
    linear-Gaussian controllers (K, k, sigK), plan cost, temperature and status.
    Payload per trajectory: T (nu nx + nu + nu(nu+1)/2) + 3 scalars."""
    K, k, sigK = engine.local_linear_policy()
    B = engine.B
    flat = torch.cat(
        [
            K.reshape(B, -1),
            k.reshape(B, -1),
            sigK.reshape(B, -1),
            (engine.costs_m[-1] if engine.costs_m else torch.zeros_like(engine.alpha)).reshape(B, 1),
            engine.alpha.reshape(B, 1),
            engine.status.to(engine.dtype).reshape(B, 1),
        ],
        dim=1,
    ).contiguous()
    if dist.is_available() and dist.is_initialized():  # also with one rank: same code path as N > 1
        flat = _all_gather_rows(flat, group)
    n = flat.shape[0]
    T, nu, nx = engine.H, engine.nu, engine.nx
    o1 = T * nu * nx
    o2 = o1 + T * nu
    o3 = o2 + T * nu * nu
    return {
        "K": flat[:, :o1].reshape(n, T, nu, nx),
        "k": flat[:, o1:o2].reshape(n, T, nu),
        "sigK": flat[:, o2:o3].reshape(n, T, nu, nu),
        "cost": flat[:, o3],
        "alpha": flat[:, o3 + 1],
        "status": flat[:, o3 + 2].to(torch.int32),
    }
