import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """Properties that one LLaVA and one Qwen configuration establish are not collected for the other full-size models (the module-scoped
    model fixture of tests/test_full_size_gpu.py is parametrised over all of them): deselected here instead of skipped inside the test."""
    keep, drop = [], []
    for it in items:
        if (it.name.startswith(("test_ragged_cohort_at_full_size[", "test_wide_tree_cohort_at_full_size["))
                and not any(f"[{m}]" in it.name for m in ("llava7b", "qwen7b"))):
            drop.append(it)
        else:
            keep.append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
