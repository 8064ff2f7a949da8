"""Multi-GPU plumbing: one process per GPU, streams sharded across ranks, no data-path collective.
Only two collectives exist, both outside the per-frame path: the max-over-ranks timing reduce and the
gather of fixed-size per-frame records (pose + counts) to every rank (BASELINE config 5).
Backend 'nccl' (= RCCL over xGMI) on GPUs; the CPU tests run the same code with 'gloo'."""
import numpy as np


def stream_offsets(rank, streams_per_rank, spacing=37):
    """global stream ids of this rank -> synthetic time offsets (weak scaling: every rank gets S new streams)"""
    return [spacing * (rank * streams_per_rank + s) for s in range(streams_per_rank)]


def max_over_ranks(dist, value, device):
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_frame_records(dist, Tcw, ninl, nmatch):
    """all_gather of the per-stream record [Tcw(16), inliers, matches] -> (world, S, 18) tensor on every rank"""
    import torch
    S = Tcw.shape[0]
    rec = torch.cat([Tcw.reshape(S, 16).float(), ninl.float().reshape(S, 1), nmatch.float().reshape(S, 1)], 1).contiguous()
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out)


def sum_over_ranks(dist, values, device):
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return t.cpu().numpy()
