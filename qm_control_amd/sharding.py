"""Multi-GPU layout of the hot path: instances are independent, so a global batch is cut into
contiguous shards, one per rank/GPU, with NO data-path collective.  torch.distributed (RCCL over xGMI on the GPU box,
gloo in the CPU tests) only synchronises the timed region and gathers timings / optional torques to rank 0."""
import numpy as np


def shard_bounds(rank, world, global_batch):
    """contiguous block [lo, hi) of a global batch for `rank`; the remainder goes to the first ranks"""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_config(cfg_all, rank, world):
    """slice every per-instance array of a scenarios.make_config() dict"""
    G = cfg_all["B"]
    lo, hi = shard_bounds(rank, world, G)
    out = {}
    for k, v in cfg_all.items():
        out[k] = v[lo:hi] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == G else v
    out["B"] = hi - lo
    return out


def init_distributed(backend, local_rank=0):
    """one process per GPU: join the process group torch.distributed.run described in the environment (backend "nccl" IS RCCL on ROCm;
    "gloo" in the CPU tests); returns the torch.distributed module"""
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def barrier(dist=None, device="cpu"):
    """process-group barrier + device synchronisation (a no-op when not distributed)"""
    if dist is None or not dist.is_initialized():
        return
    dist.barrier()
    if device == "cuda":
        import torch
        torch.cuda.synchronize()


def max_over_ranks(value, dist=None, device="cpu"):
    """max of a python float over all ranks (identity when not distributed)"""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, dist=None, device="cpu"):
    """concatenate equally-shaped per-rank f64 arrays [b, k] on every rank in rank order (parity runs: torques)"""
    if dist is None or not dist.is_initialized():
        return np.asarray(local)
    import torch
    t = torch.as_tensor(np.ascontiguousarray(local), dtype=torch.float64, device=device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return np.concatenate([o.cpu().numpy() for o in outs], axis=0)
