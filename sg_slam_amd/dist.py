"""Multi-GPU plumbing: one process per GPU, streams sharded across ranks, no data-path collective.
Two collectives exist, neither on the per-frame data path: the max-over-ranks timing reduce and the
gather of the fixed-size per-frame records to rank 0 (BASELINE config 5).
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
    """BASELINE config 5 / SURVEY.md §8(e): after every step the per-frame record {n, cv::KeyPoint[cap], desc[cap][32], Tcw} of each of the rank's S streams goes to
    RANK 0 with one `gather` per step (RCCL over xGMI on GPUs — grouped send / recv underneath —, gloo on CPU).  Record = 16 B header (n, 3 pad) + cap*28 + cap*32 +
    64 B pose; the S records of a step are packed into one contiguous uint8 buffer by ONE kernel of the library (sgx_tracker_pack_records_dev) on a side stream
    that waits for the frame's tracking event, so the next step's kernels overlap pack and transfer.  Double-buffered: before send[b] is packed again, the
    collective that read it two steps earlier is waited for (one Work handle per buffer).  Only rank 0 allocates the (world, S, record) receive buffers:
    8 x 512 x 61.5 KB = 252 MB per step land on rank 0, nothing on the other ranks (an all_gather would deliver the same 252 MB to every rank)."""

    def __init__(self, dist, streams, cap, device, async_stream=True, dst=0):
        import torch
        self.dist, self.S, self.cap, self.device, self.dst = dist, streams, cap, device, dst
        self.rec_bytes = 16 + cap * 28 + cap * 32 + 64
        self.world = dist.get_world_size(); self.rank = dist.get_rank()
        self.send = [torch.zeros((streams, self.rec_bytes), dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [torch.zeros((self.world, streams, self.rec_bytes), dtype=torch.uint8, device=device) for _ in range(2)] if self.rank == dst else [None, None]
        self.stream = torch.cuda.Stream() if (async_stream and str(device).startswith('cuda')) else None
        self.step_idx = 0
        self.bytes_moved = 0
        self.pending = [None, None]

    def pack_torch(self, buf, n, keys, desc, Tcw):
        """reference packer (tests, CPU): the same layout with tensor slice assignments"""
        import torch
        S, cap = self.S, self.cap
        buf[:, 0:16] = 0
        buf[:, 0:4] = n.reshape(S, 1).to(torch.int32).view(torch.uint8).reshape(S, 4)
        o = 16
        buf[:, o:o + cap * 28] = keys.reshape(S, cap * 28); o += cap * 28
        buf[:, o:o + cap * 32] = desc.reshape(S, cap * 32); o += cap * 32
        buf[:, o:o + 64] = Tcw.reshape(S, 16).to(torch.float32).contiguous().view(torch.uint8).reshape(S, 64)

    def _step_bytes(self):
        """bytes this RANK moves per step: a sender ships its S records; the destination receives the other ranks' (its own slice is a local copy)"""
        return (self.world - 1) * self.S * self.rec_bytes if self.rank == self.dst else self.S * self.rec_bytes

    def _gather(self, b, async_op):
        out = list(self.recv[b].unbind(0)) if self.rank == self.dst else None
        return self.dist.gather(self.send[b], out, dst=self.dst, async_op=async_op)

    def submit_tracker(self, tracker):
        """pack the records of the frame `tracker` (sg_slam_amd.tracker_native.TrackerNative) tracked last and start the gather; returns immediately"""
        import torch
        b = self.step_idx & 1
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                if self.pending[b] is not None:
                    self.pending[b].wait(); self.pending[b] = None    # orders this side stream after the collective that read send[b] two steps ago
                tracker.pack_records(self.send[b], stream=self.stream.cuda_stream)       # waits for the frame's tracking event inside the library
                self.pending[b] = self._gather(b, True)
        else:
            tracker.pack_records(self.send[b]); self._gather(b, False)
        self.bytes_moved += self._step_bytes()
        self.step_idx += 1
        return self.recv[b]

    def submit(self, n, keys, desc, Tcw, after_event=None):
        """the same from loose tensors (torch packer): CPU tests and callers without the native tracker"""
        import torch
        b = self.step_idx & 1
        if self.stream is not None:
            if after_event is not None:
                self.stream.wait_event(after_event)
            else:
                self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                if self.pending[b] is not None:
                    self.pending[b].wait(); self.pending[b] = None
                self.pack_torch(self.send[b], n, keys, desc, Tcw)
                self.pending[b] = self._gather(b, True)
        else:
            self.pack_torch(self.send[b], n, keys, desc, Tcw)
            self._gather(b, False)
        self.bytes_moved += self._step_bytes()
        self.step_idx += 1
        return self.recv[b]

    def wait(self):
        import torch
        for b in range(2):
            if self.pending[b] is not None:
                self.pending[b].wait(); self.pending[b] = None
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def last(self):
        """receive buffer of the most recent step (rank `dst` only)"""
        return self.recv[(self.step_idx - 1) & 1]

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


class NativeRecordGather:
    """The same gather issued by the LIBRARY (sgx_dist_gather_records: grouped ncclSend / ncclRecv of RCCL loaded by libsgx.so itself) — the path a C++ host uses for BASELINE
    config 5 without Python (include/sgx.h).  The 128-byte RCCL id of rank 0 reaches the other ranks through `broadcast_id` (a callable bytes -> bytes; with
    torch.distributed: a broadcast of a uint8 tensor; a single rank passes None).  Synchronous interface for tests and examples; the production overlap (side stream, double
    buffer) is FrameRecordGather's, which this class does not replace in bench.py."""

    def __init__(self, lib, streams, cap, device, world=1, rank=0, broadcast_id=None, dst=0):
        import ctypes as C
        import torch
        self.lib, self.S, self.cap, self.world, self.rank, self.dst = lib, streams, cap, world, rank, dst
        self.rec_bytes = 16 + cap * 28 + cap * 32 + 64
        ident = (C.c_char * 128)()
        if rank == 0: lib.check(lib.dll.sgx_dist_unique_id(ident))
        raw = bytes(ident) if broadcast_id is None else broadcast_id(bytes(ident))
        h = C.c_void_p()
        lib.check(lib.dll.sgx_dist_create(C.c_char_p(raw), world, rank, C.byref(h)))
        self.h = h
        self.send = torch.zeros((streams, self.rec_bytes), dtype=torch.uint8, device=device)
        self.recv = torch.zeros((world, streams, self.rec_bytes), dtype=torch.uint8, device=device) if rank == dst else None

    def gather_tracker(self, tracker, stream=None):
        """pack the records of the frame the tracker tracked last and gather them to `dst` on `stream` (a raw hipStream_t; None = the legacy stream); asynchronous to the host"""
        import ctypes as C
        tracker.pack_records(self.send, stream=stream)
        self.lib.check(self.lib.dll.sgx_dist_gather_records(self.h, C.c_void_p(self.send.data_ptr()), self.S * self.rec_bytes,
                                                            None if self.recv is None else C.c_void_p(self.recv.data_ptr()), self.dst, None if stream is None else C.c_void_p(stream)))
        return self.recv

    def close(self):
        if self.h: self.lib.dll.sgx_dist_destroy(self.h); self.h = None
