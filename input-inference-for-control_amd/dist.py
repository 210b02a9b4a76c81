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


def shard_sizes(total, world):
    """Shard sizes of every rank under shard_range: deterministic, so no rank has to ask another for its size."""
    return [hi - lo for lo, hi in (shard_range(total, r, world) for r in range(world))]


def _all_gather_rows(t, total=None, group=None):
    """All-gather along dim 0 in ONE collective. `total` = the global number of rows: the shards are the contiguous partition of
    shard_range, so every rank computes every shard size itself and no size exchange is needed. With more than one rank `total`
    is REQUIRED (round-5 advice: a default of "equal shards" let ranks with ragged shards enter the collective with different
    buffer sizes -- a hang or silently mis-sliced controllers); a rank whose row count is not its shard_range share raises
    before anything is exchanged. Ragged shards travel in a max-size buffer (at most one padding row per rank) and are cut
    back after the exchange."""
    world = dist.get_world_size(group)
    if total is None:
        if world > 1:
            raise ValueError("gather over more than one rank needs the global batch size (`total`): shard sizes are derived from "
                             "shard_range(total, rank, world), never assumed equal")
        total = int(t.shape[0])
    sizes = shard_sizes(total, world)
    rank = dist.get_rank(group)
    if sizes[rank] != t.shape[0]:
        raise ValueError(f"rank {rank} holds {t.shape[0]} rows but shard_range({total}, {rank}, {world}) has {sizes[rank]}: "
                         "pass the global batch size of a shard_range partition as `total`")
    mx = max(sizes)
    src = t.contiguous()
    if src.shape[0] != mx:  # the short shards: one padding row
        src = torch.cat([src, src.new_zeros((mx - src.shape[0],) + tuple(src.shape[1:]))], dim=0)
    out = t.new_empty((world * mx,) + tuple(t.shape[1:]))
    dist.all_gather_into_tensor(out, src, group=group)
    if len(set(sizes)) == 1:
        return out
    out = out.reshape((world, mx) + tuple(t.shape[1:]))
    return torch.cat([out[r, :n] for r, n in enumerate(sizes)], dim=0)


def gather_policy(engine, group=None, total=None):
    """The one collective (a single all_gather_into_tensor: RCCL over xGMI): every rank ends up with the whole batch's
    time-varying linear-Gaussian controllers (K, k, sigK), plan cost, temperature and status.
    Payload per trajectory: T (nu nx + nu + nu(nu+1)/2) + 3 scalars.
    total: the global batch size (shards from shard_range); required whenever more than one rank takes part."""
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
        flat = _all_gather_rows(flat, total, group)
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
