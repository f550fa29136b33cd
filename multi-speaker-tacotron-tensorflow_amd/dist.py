"""Multi-GPU: inference shards by batch rows, one process per GPU, NO collective on the data path
(rows are independent at inference -- BatchNorm uses moving statistics, attention and GRUs
are per row).  torch.distributed (RCCL on ROCm, gloo on CPU) is used only for the barrier and the
max-over-ranks reduction of the timed region, as the bench contract requires."""
import os


def shard_range(n_rows, rank, world_size):
    """Contiguous slice [lo, hi) of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend)
    return dist


def max_over_ranks(value, device="cpu"):
    """max of a python float over all ranks (the timed region is the slowest rank's)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value, device="cpu"):
    """every rank's python float, in rank order, on every rank (bench.py: per-rank ms per step)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def gather_rows(local_np, world_size):
    """all_gather of per-rank numpy row blocks (host side; outputs normally stay on their GPU)."""
    import torch.distributed as dist
    if not dist.is_initialized() or world_size == 1:
        return [local_np]
    out = [None] * world_size
    dist.all_gather_object(out, local_np)
    return out
