"""Multi-GPU sharding of the Monte-Carlo loop: independent shot ranges per rank + ONE all-reduce of the tally.

Replaces the reference's process pool + Python tally loop (src/simulation/engine.py:433-457).  One process per GPU
(torchrun); `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.  The Philox streams
are keyed by the GLOBAL shot index, so the reduced tally does not depend on the number of ranks.
"""
import numpy as np

from . import _lib


def shard_range(total, rank, world):
    """Contiguous split of [0, total) into `world` ranges whose sizes differ by at most one -> (begin, count)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, rem = divmod(int(total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def local_device(device=None):
    """The GPU this rank drives: the explicit argument, else LOCAL_RANK (torchrun), else 0."""
    import os
    if device is not None:
        return int(device)
    return int(os.environ.get("LOCAL_RANK", "0"))


def collective_device(backend, device=None):
    """Where the tensors of a collective must live for `backend`: RCCL ("nccl") reduces device tensors only, gloo host tensors.
    The one rule both helpers below follow."""
    import torch
    if backend == "nccl":
        return torch.device("cuda", local_device(device) if device is not None else torch.cuda.current_device())
    return torch.device("cpu")


def allreduce_tally(tally, device=None, group=None, comm=None):
    """Sum the int64[16] tally over all ranks (the single collective of the path); identity when not distributed.
    `comm`: a `_lib.Comm` (native RCCL communicator of the C ABI, qldpc_tally_allreduce) -- used instead of torch.distributed when given,
    so a launcher without PyTorch can run the N > 1 path.  `device`: the rank's GPU index (used when the torch backend is RCCL;
    default = torch's current device)."""
    t = np.ascontiguousarray(tally, dtype=np.int64)
    if comm is not None:
        return comm.allreduce(t)
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t.copy()
    if isinstance(device, torch.device):
        device = device.index if device.type == "cuda" else None
    tt = torch.from_numpy(t.copy()).to(collective_device(dist.get_backend(group), device))
    dist.all_reduce(tt, op=dist.ReduceOp.SUM, group=group)
    return tt.cpu().numpy()


def gather_in_shot_order(local, total, group=None, device=None):
    """Concatenate the per-rank uint8 arrays of a `shard_range` split back into global shot order (all ranks get the result)."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=np.uint8)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local.copy()
    world = dist.get_world_size(group)
    width = -(-int(total) // world)                       # the largest shard; shorter shards are padded
    buf = torch.zeros(max(width, 1), dtype=torch.uint8)
    buf[:local.size] = torch.from_numpy(local.copy())
    dev = collective_device(dist.get_backend(group), device)
    parts = [torch.empty_like(buf, device=dev) for _ in range(world)]
    dist.all_gather(parts, buf.to(dev), group=group)
    return np.concatenate([parts[r].cpu().numpy()[:shard_range(total, r, world)[1]] for r in range(world)])


def cut_at_target(outcomes, errors_so_far, target):
    """In-order early stop (reference engine.py:441-464): the number of leading trials to count so that the running total of
    logical errors reaches `target` exactly at the last counted trial; len(outcomes) when the target is not reached."""
    bad = np.flatnonzero(np.asarray(outcomes) != 0)
    need = int(target) - int(errors_so_far)
    if need <= 0:
        return 0
    return int(bad[need - 1]) + 1 if bad.size >= need else int(np.asarray(outcomes).size)


def run_sharded(total_shots, local_tally_fn, rank=None, world=None, shot_offset=0, device=None, group=None):
    """Every rank tallies its own shot range with `local_tally_fn(shot_begin, count) -> int64[16]`, then one all-reduce."""
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    begin, count = shard_range(total_shots, rank, world)
    local = np.zeros(_lib.TALLY_SLOTS, np.int64) if count == 0 else np.asarray(local_tally_fn(shot_offset + begin, count), dtype=np.int64)
    return allreduce_tally(local, device=device, group=group)


def run_code_capacity(graph, L, p, seed, total_shots, max_iter=50, alpha=1.0, alpha_mode="dynamical", damping=1.0, clip_llr=20.0,
                      use_osd=True, flags=0, device=None, group=None):
    """Code-capacity Monte-Carlo point on all ranks: HIP pipeline per shard + RCCL all-reduce -> global tally int64[16]."""
    def local(begin, count):
        return _lib.cc_sample_decode_tally(graph, L, p, seed, begin, count, max_iter=max_iter, alpha=alpha, alpha_mode=alpha_mode,
                                           damping=damping, clip_llr=clip_llr, use_osd=use_osd, flags=flags)
    return run_sharded(total_shots, local, device=device, group=group)


def tally_to_result(tally):
    """Result dict with the keys of the reference's run_simulation (engine.py:466-472) from a tally."""
    T = _lib.TALLY
    trials = max(1, int(tally[T["trials"]]))
    return {
        "logical_error_rate": int(tally[T["total_err"]]) / trials,
        "z_logical_error_rate": int(tally[T["z_err"]]) / trials,
        "x_logical_error_rate": int(tally[T["x_err"]]) / trials,
        "num_trials": int(tally[T["trials"]]),
        "logical_errors": int(tally[T["total_err"]]),
    }
