"""Batch sharding of the sampling path over the GPUs of one node.

Samples are independent through the predictor and the decoders, so the path shards by batch with no
data-path collective: rank r owns the contiguous rows [r*B/W, (r+1)*B/W) of the global batch, holds
a full replica of the (small) weights, and draws its noise from the Philox stream keyed by the
GLOBAL sample index, so the result does not depend on the world size.  The only collective is one
all-gather of the decoded outputs at the end (RCCL over xGMI with backend "nccl"; "gloo" on CPU in
the tests).

One process per GPU: launch with `python -m torch.distributed.run --nproc-per-node N ...`.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (no-op for a single process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return int(os.environ.get("RANK", "0")), world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def shard_range(global_batch, rank, world):
    """Contiguous split; every rank gets the same count (the global batch must divide evenly)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} does not divide over {world} ranks")
    n = global_batch // world
    return rank * n, n


def all_gather_cat(t, group=None):
    """ONE all-gather of equally-sized shards, concatenated along dim 0 (identity for a single process)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    t = t.contiguous()
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def shared_seed(group=None, seed=None):
    """ONE Philox key for all ranks of `group`: rank 0's draw from torch's default generator (or its `seed`), broadcast.  Every rank
    draws, so the generators stay in step.  This IS a collective -- call it where every rank of the group passes (the sampling helpers
    themselves never do: `_module.fresh_seed` is local), then hand the key down as `seed=`."""
    from ._module import fresh_seed
    key = fresh_seed(seed)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        k = torch.tensor([key - (1 << 64) if key >= (1 << 63) else key], dtype=torch.int64, device=dev)
        dist.broadcast(k, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        key = int(k.cpu()[0].item()) & 0xFFFFFFFFFFFFFFFF
    return key


def sample_sharded(sample_fn, global_batch, group=None, gather=True, share_seed=False):
    """Run `sample_fn(start, count)` on this rank's rows and gather once.

    sample_fn must be a function of the GLOBAL row indices only (e.g. uncond_sampler(...,
    sample_offset=start) followed by the decoders); it returns a tensor or a tuple of tensors with
    `count` rows each.  With `share_seed=True` the ranks first agree on one fresh Philox key
    (`shared_seed`) and the call is `sample_fn(start, count, seed)`: the way to draw fresh noise for a
    sharded batch without choosing a seed by hand."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    start, count = shard_range(global_batch, rank, world)
    local = sample_fn(start, count, shared_seed(group)) if share_seed else sample_fn(start, count)
    if not gather:
        return local
    if isinstance(local, (tuple, list)):
        # pack the per-modality outputs into one buffer so that it is a SINGLE collective
        flat = torch.cat([o.reshape(count, -1) for o in local], dim=1)
        g = all_gather_cat(flat, group)
        outs, col = [], 0
        for o in local:
            n = o[0].numel()
            outs.append(g[:, col:col + n].reshape((g.shape[0],) + tuple(o.shape[1:])))
            col += n
        return tuple(outs)
    return all_gather_cat(local, group)


def all_reduce_sum(t, group=None):
    """In-place sum of `t` over the ranks (identity for a single process): the FID stage's exchange of feature sums
    (fid_score.ActivationStatistics.all_reduce; SURVEY 8(e): 2048 + 2048^2 values)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def langevin_norm_allreduce(norm_sums, group=None):
    """Opt-in exact mode for the corrector: its step size uses BATCH-mean norms (sde_helper2.py:97-99),
    so a sharded run matches the single-device reference only if the two sums are all-reduced."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(norm_sums, op=dist.ReduceOp.SUM, group=group)
    return norm_sums
