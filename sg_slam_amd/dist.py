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


class FrameRecordGather:
    """BASELINE config 5 / SURVEY.md §8(e): after every step batch the per-frame record {n, cv::KeyPoint[cap], desc[cap][32], Tcw} of each stream is
    gathered from device memory to every rank with ONE all_gather (RCCL over xGMI on GPUs, gloo on CPU), on its own stream so the next step's kernels
    overlap the transfer.  Record = 16 B header (n, 3 pad) + cap*28 + cap*32 + 64 B pose, packed per stream into one contiguous uint8 buffer
    (double-buffered: the pack of step t+1 must not overwrite what the collective of step t still reads)."""

    def __init__(self, dist, streams, cap, device, async_stream=True):
        import torch
        self.dist, self.S, self.cap, self.device = dist, streams, cap, device
        self.rec_bytes = 16 + cap * 28 + cap * 32 + 64
        self.world = dist.get_world_size()
        self.send = [torch.zeros((streams, self.rec_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [torch.zeros((self.world, streams, self.rec_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.stream = torch.cuda.Stream() if (async_stream and str(device).startswith('cuda')) else None
        self.packed = [torch.cuda.Event() for _ in range(2)] if self.stream is not None else None
        self.step_idx = 0
        self.bytes_moved = 0
        self.pending = None

    def pack(self, buf, n, keys, desc, Tcw):
        import torch
        S, cap = self.S, self.cap
        buf[:, 0:4] = n.reshape(S, 1).to(torch.int32).view(torch.uint8).reshape(S, 4)
        o = 16
        buf[:, o:o + cap * 28] = keys.reshape(S, cap * 28); o += cap * 28
        buf[:, o:o + cap * 32] = desc.reshape(S, cap * 32); o += cap * 32
        buf[:, o:o + 64] = Tcw.reshape(S, 16).to(torch.float32).contiguous().view(torch.uint8).reshape(S, 64)

    def submit(self, n, keys, desc, Tcw, after_event=None):
        """pack + all_gather of one step's records; returns immediately when an async stream is used (wait() joins)"""
        import torch
        b = self.step_idx & 1
        if self.stream is not None:
            if after_event is not None:
                self.stream.wait_event(after_event)
            else:
                self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.pack(self.send[b], n, keys, desc, Tcw)
                self.packed[b].record(self.stream)
                self.pending = self.dist.all_gather_into_tensor(self.recv[b].view(self.world * self.S, self.rec_bytes), self.send[b], async_op=True)
        else:
            self.pack(self.send[b], n, keys, desc, Tcw)
            self.dist.all_gather_into_tensor(self.recv[b].view(self.world * self.S, self.rec_bytes), self.send[b])
        self.bytes_moved += self.world * self.S * self.rec_bytes
        self.step_idx += 1
        return self.recv[b]

    def wait(self):
        import torch
        if self.pending is not None:
            self.pending.wait(); self.pending = None
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def unpack(self, rec):
        """(world, S, rec_bytes) uint8 -> dict of numpy arrays: n (world,S), keys (world,S,cap,28) u8, desc (world,S,cap,32) u8, Tcw (world,S,16) f32"""
        import numpy as np
        a = rec.cpu().numpy(); cap = self.cap
        n = a[:, :, 0:4].copy().view(np.int32)[:, :, 0]
        o = 16
        keys = a[:, :, o:o + cap * 28].reshape(a.shape[0], a.shape[1], cap, 28); o += cap * 28
        desc = a[:, :, o:o + cap * 32].reshape(a.shape[0], a.shape[1], cap, 32); o += cap * 32
        T = a[:, :, o:o + 64].copy().view(np.float32)
        return dict(n=n, keys=keys, desc=desc, Tcw=T)
