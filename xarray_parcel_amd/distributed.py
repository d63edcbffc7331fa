"""
Multi-GPU: columns are independent, so the grid is cut into contiguous y-slabs (one per rank, one process
per GPU) with no data-path collective; the per-column scalars are gathered to one rank with a single
collective at the end (RCCL `gather` when the process group's backend is nccl, gloo on CPU in the tests).
The reference's counterpart is dask chunking over the horizontal dims (pf.py:343-346, 592, 667) with
per-chunk results concatenated by the scheduler.
"""
import torch
import torch.distributed as dist


def slab_bounds(n_rows, world, rank):
    """Contiguous split of n_rows (the y axis, or the flattened (time, y) axis) into `world` slabs whose sizes
    differ by at most one."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_columns(local, names, dst=0, group=None):
    """Gather per-column results (dict name -> 1-D tensor, same length on a rank, lengths may differ across
    ranks) to `dst`.  One collective: the fields are stacked into a (len(names), ncol_local) buffer, padded to
    the largest slab.  Returns dict name -> concatenated tensor on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = int(local[names[0]].numel())
    dev = local[names[0]].device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.int64, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    dtype = local[names[0]].dtype
    for k in names[1:]:
        dtype = torch.promote_types(dtype, local[k].dtype)
    buf = torch.zeros((len(names), n_max), dtype=dtype, device=dev)
    for i, k in enumerate(names):
        buf[i, :n_local] = local[k].reshape(-1).to(dtype)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    return {k: torch.cat([out[r][i, :sizes[r]] for r in range(world)]).to(local[k].dtype)
            for i, k in enumerate(names)}


def sharded_cape_cin(pressure, temperature, dewpoint, names=('cape', 'cin'), dst=0, group=None, compute=None,
                     **kwargs):
    """Each rank passes ITS slab (nlev, rows_local, nx) (or already flattened (nlev, ncol_local)); results for
    the whole grid arrive on `dst`.  `compute` defaults to the HIP path (numpy_api.cape_cin_columns); the CPU
    rehearsal test injects a stand-in."""
    if compute is None:
        from . import numpy_api
        compute = numpy_api.cape_cin_columns
    res = compute(pressure, temperature, dewpoint, want=tuple(names), **kwargs)
    local = {k: torch.as_tensor(res[k]).reshape(-1) for k in names}
    return gather_columns(local, list(names), dst=dst, group=group)
