"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask), wrapped for torch.

On gfx950 the mask bits are dealt round-robin over the 8 XCDs (bit b -> XCD b % 8), so the lowest n bits with n a multiple of 8 give n / 8 CUs on every XCD.
Used by bench.py to give the detector (HBM / matrix-pipe bound) and the tracking chain (VALU-issue bound) disjoint CU sets instead of letting the two queues
interleave workgroups on every CU."""
import ctypes as C

_hip = None


def _lib():
    global _hip
    if _hip is None:
        _hip = C.CDLL('libamdhip64.so')
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    return _hip


def cu_mask_words(first, count, total=256):
    """mask words enabling mask bits [first, first + count)"""
    words = [0] * ((total + 31) // 32)
    for b in range(first, min(first + count, total)):
        words[b // 32] |= 1 << (b % 32)
    return words


def masked_stream(first, count, total=256):
    """torch.cuda.ExternalStream whose kernels run only on mask bits [first, first + count) of the device's CUs"""
    import torch
    words = cu_mask_words(first, count, total)
    arr = (C.c_uint32 * len(words))(*words)
    h = C.c_void_p()
    rc = _lib().hipExtStreamCreateWithCUMask(C.byref(h), len(words), arr)
    if rc != 0 or not h.value:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask failed with {rc}')
    return torch.cuda.ExternalStream(h.value)
