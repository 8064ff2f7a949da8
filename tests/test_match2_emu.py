"""LocalMapping matcher gates (tier N2) on the kernel-logic emulator against the oracle."""
import match2_cases as mc


def test_hamming_matrix_emu(emu, oracle):
    mc.check_hamming(emu, oracle)


def test_search_for_triangulation_emu(emu, oracle):
    mc.check_triangulation(emu, oracle, n_cases=4)


def test_search_by_bow_emu(emu, oracle):
    mc.check_bow(emu, oracle, n_cases=3)


def test_search_by_bow_kf_emu(emu, oracle):
    mc.check_bow_kf(emu, oracle, n_cases=3)


def test_fuse_search_emu(emu, oracle):
    mc.check_fuse(emu, oracle, n_cases=3)


def test_project_keyframe_emu(emu, oracle):
    mc.check_project_kf(emu, oracle, n_cases=2)


def test_fuse_search_sim3_emu(emu, oracle):
    mc.check_fuse_sim3(emu, oracle, n_cases=2)


def test_project_sim3_emu(emu, oracle):
    mc.check_project_sim3(emu, oracle, n_cases=2)


def test_search_by_sim3_emu(emu, oracle):
    mc.check_search_by_sim3(emu, oracle, n_cases=2)


def test_search_for_initialization_emu(emu, oracle):
    mc.check_search_for_initialization(emu, oracle, n_cases=2)
