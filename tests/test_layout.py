"""Repository contract: the oracle is test infrastructure only; required files exist."""

import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for base, _, files in os.walk(os.path.join(ROOT, "learning_to_adapt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert not pat.search(text), "%s imports the oracle" % os.path.join(base, f)
                assert "/root/reference" not in text


def test_gpu_side_files_do_not_read_the_reference():
    for rel in ("bench.py", "__graft_entry__.py"):
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            assert "/root/reference" not in open(path).read()


def test_required_files_exist():
    for rel in ("include/l2a.h", "oracle/__init__.py", "tests/golden/cases.json", "tools/gen_golden.py",
                "learning_to_adapt_amd/csrc/l2a_api.hip", "learning_to_adapt_amd/csrc/l2a_mfma.h",
                "learning_to_adapt_amd/csrc/l2a_kernels.h", "learning_to_adapt_amd/csrc/l2a_valu.h",
                "learning_to_adapt_amd/csrc/l2a_mfma_inst.hip"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
