"""Sim3Solver (LoopClosing::ComputeSim3's RANSAC initialiser): oracle self-checks, then the kernel-logic emulator against the oracle."""
import sim3solver_cases as sc


def test_glibc_rand_replica(oracle):
    sc.check_glibc_rand(oracle)


def test_horn_recovers_similarity(oracle):
    sc.check_horn(oracle)


def test_sim3_solver_emu(emu, oracle):
    sc.check_solver(emu, oracle, n_cases=3, exact=True)
