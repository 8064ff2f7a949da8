"""CPU tier: Optimizer::OptimizeSim3 — the oracle against the generating similarity, the kernel-logic emulator against the oracle."""
import sim3_cases as sc


def test_sim3_oracle_recovers(oracle):
    sc.check_oracle_recovers(oracle)


def test_sim3_emu(emu, oracle):
    sc.check_sim3(emu, oracle, n_cases=4)


def test_essential_graph_oracle_recovers(oracle):
    sc.check_eg_oracle_recovers(oracle)


def test_essential_graph_emu(emu, oracle):
    sc.check_eg(emu, oracle, n_cases=3)
