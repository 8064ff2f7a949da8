"""DBoW2 vocabulary transform (Frame::ComputeBoW) on the kernel-logic emulator against the oracle."""
import voc_cases as vc


def test_voc_transform_emu(emu, oracle):
    vc.check_transform(emu, oracle, n_cases=2)


def test_voc_files_and_score_emu(emu, oracle, tmp_path):
    vc.check_files_and_score(emu, oracle, str(tmp_path))


def test_voc_bow_chain_emu(emu, oracle):
    vc.check_bow_chain(emu, oracle)
