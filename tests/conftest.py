import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need an MI355X: on a box without one they are skipped, not failed, so that a plain
    `pytest tests` stays readable there.  On a GPU box nothing is skipped: a missing liblwg.so must fail loudly."""
    import torch
    if not torch.cuda.is_available():
        skip = pytest.mark.skip(reason="no GPU visible (gpu-marked tests run through gpurun on an MI355X)")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
