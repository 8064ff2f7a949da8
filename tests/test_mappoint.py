"""Batched MapPoint post-steps on the kernel-logic emulator against the oracle."""
import mappoint_cases as mpc


def test_mappoint_post_steps_emu(emu, oracle):
    mpc.check_mappoint(emu, oracle, n_cases=2)


def test_triangulation_step_emu(emu, oracle):
    mpc.check_triangulation_step(emu, oracle, n_cases=2, exact=True)
