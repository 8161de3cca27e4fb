"""Base-sharded MSM across the GPUs of one node.

sum_i s_i*P_i is a sum of independent terms: rank r runs the full single-GPU Pippenger on its
contiguous shard and produces ONE group element; the partial results (96 B for G1, 192 B for G2)
are exchanged with one all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) and folded
locally on every rank with the library's host group law - RCCL has no user-defined reduction op,
so a literal reduce cannot add curve points.  The exchange is a few hundred bytes: latency-bound.
"""

import numpy as np

from .multiexp import point_add


def shard_bounds(n, world, rank):
    """Contiguous split of n terms over `world` ranks (first ranks take the remainder)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_records(record, device=None, process_group=None):
    """every rank's record (a small uint64 array) as the rows of one host array [world, len]: ONE collective into one
    tensor and ONE copy back to the host - a list of `world` tensors brought back one by one was `world` synchronising
    copies per step (8 x ~20 us at 8 ranks, next to a 3.3 ms multiexp)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(process_group)
    mine = torch.from_numpy(np.ascontiguousarray(record, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    try:
        dist.all_gather_into_tensor(out, mine, group=process_group)
    except (RuntimeError, NotImplementedError, AttributeError):   # a backend without the tensor form
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=process_group)
        out = torch.stack(parts)
    return out.cpu().numpy().view(np.uint64)


def fold_partials(partial, group, device=None, process_group=None):
    """all-gather every rank's partial affine record and add them up (same result on all ranks)."""
    parts = _all_gather_records(partial, device=device, process_group=process_group)
    total = parts[0]
    for p in parts[1:]:
        total = point_add(group, total, p)
    return total


def sharded_multiexp(worker, bases_shard, density_map, scalars_shard, group, device=None, process_group=None,
                     **kw):
    """multiexp over this rank's shard, then the fold.  `bases_shard`/`scalars_shard` are the
    rank-local pieces (see shard_bounds)."""
    from .multiexp import multiexp

    part = multiexp(worker, bases_shard, density_map, scalars_shard, **kw).wait()
    return fold_partials(part, group, device=device, process_group=process_group)


def fold_sums(sums, process_group=None, device=None):
    """all-gather every rank's multiexp-result record (960 B) and add them slot-wise."""
    from .groth16 import sums_add

    parts = _all_gather_records(sums, device=device, process_group=process_group)
    total = parts[0]
    for p in parts[1:]:
        total = sums_add(total, p)
    return total


def create_proof_sharded(part_fn, params, r, s, process_group=None, device=None):
    """One Groth16 proof over all ranks of `process_group`: `part_fn(rank, world)` returns this rank's
    multiexp sums (groth16.prove_witness_part / prove_demo_part over its slice of the scalars); one
    all-gather of 960 bytes, slot-wise fold, and every rank assembles the same proof."""
    import torch.distributed as dist

    from .groth16 import assemble

    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
    total = fold_sums(part_fn(rank, world), process_group=process_group, device=device)
    return assemble(params, total, r, s)
