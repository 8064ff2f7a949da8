"""GPU (MI355X): the C-ABI collective of BASELINE config 5 — sgx_dist_gather_records, the grouped ncclSend / ncclRecv the C++ host issues itself (RCCL loaded by libsgx.so on
first use) — with one rank: the only world size a 1-GPU box can form (RCCL refuses two ranks on one device).  Rank 0's block must arrive in slot 0 of the root's receive
buffer, on the stream it was enqueued on, without any torch.distributed process group.  The 2-rank semantics of the same gather are covered on CPU (tests/test_dist_gloo.py)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_native_gather_single_rank(gpulib):
    import torch
    from sg_slam_amd.dist import NativeRecordGather
    S, cap = 8, 64
    g = NativeRecordGather(gpulib, S, cap, 'cuda', world=1, rank=0)
    w, r = C.c_int32(-1), C.c_int32(-1)
    gpulib.check(gpulib.dll.sgx_dist_world(g.h, C.byref(w), C.byref(r)))
    assert (w.value, r.value) == (1, 0)
    payload = torch.randint(0, 256, (S, g.rec_bytes), dtype=torch.uint8, device='cuda')

    class FakeTracker:                      # stands in for TrackerNative.pack_records: puts `payload` into the send buffer on the given stream
        def pack_records(self, buf, stream=None):
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)) if stream else torch.cuda.stream(torch.cuda.current_stream()):
                buf.copy_(payload)

    st = torch.cuda.Stream()
    for _ in range(3):                      # repeated use of one communicator
        out = g.gather_tracker(FakeTracker(), stream=st.cuda_stream)
        st.synchronize()
        assert out.shape == (1, S, g.rec_bytes) and torch.equal(out[0], payload)
        payload = payload.flip(0).contiguous()
    # argument checks: a non-root rank must not pass a receive buffer and the root must
    assert gpulib.dll.sgx_dist_gather_records(g.h, C.c_void_p(g.send.data_ptr()), S * g.rec_bytes, None, 0, None) != 0
    g.close()
