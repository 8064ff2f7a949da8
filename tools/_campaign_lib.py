"""Which library the differential campaigns drive: the kernel-logic emulator (default, CPU) or — SGX_CAMPAIGN_LIB=device — the product library on the GPU
(tests/test_campaign_gpu.py runs a fixed-seed slice of every campaign that way, so the random inputs also reach the hipcc build)."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def taps_lib():
    """tests/taps/libsgx_taps.so: the product sources with the test / tuning taps of include/sgx_debug.h and the SGX_* environment switches (make -C sg_slam_amd/csrc taps)"""
    from sg_slam_amd.capi import SgxLib
    lib = SgxLib(os.path.join(ROOT, os.environ.get('SGX_TOOL_TAPS_LIB', os.path.join('tests', 'taps', 'libsgx_taps.so'))))      # SGX_TOOL_TAPS_LIB: an A/B variant (tools/ab_build_det.sh)
    assert 'gfx950' in lib.version() and lib.has_taps
    return lib


def campaign_lib(taps=False):
    """(library, array flavour for the tracker harness); taps=True: the campaign drives a test tap, which on the device only the tap build has"""
    if os.environ.get('SGX_CAMPAIGN_LIB', '') == 'device':
        import torch                      # torch first: its HIP runtime must be initialised before libsgx.so touches the device (the tracker harness allocates through torch)
        torch.cuda.init()
        if taps:
            return taps_lib(), 'torch'
        import sg_slam_amd
        lib = sg_slam_amd.load()
        assert 'gfx950' in lib.version()
        return lib, 'torch'
    from sg_slam_amd.capi import SgxLib
    return SgxLib(os.path.join(ROOT, 'tests', 'emu', 'libsgx_emu.so')), 'numpy'


def tool_lib():
    """the library a measurement tool drives: the product library, or — SGX_BENCH_TAPS_LIB=1, A/B runs of the SGX_* switches — the tap build"""
    if os.environ.get('SGX_BENCH_TAPS_LIB') == '1':
        return taps_lib()
    if os.environ.get('SGX_BENCH_AB_LIB'):      # an A/B build of tools/ab_build.sh
        from sg_slam_amd.capi import SgxLib
        return SgxLib(os.path.join(ROOT, os.environ['SGX_BENCH_AB_LIB']))
    import sg_slam_amd
    return sg_slam_amd.load()
