import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The first box with two or more GPUs that runs `pytest -m gpu` must exercise the multi-rank RCCL path before anything
    else can use up the time: tests whose name says rccl / ranks go first there (single-GPU boxes keep the file order)."""
    try:
        import torch

        multi = torch.cuda.is_available() and torch.cuda.device_count() >= 2
    except Exception:
        multi = False
    if multi:
        first = [it for it in items if "rccl" in it.name or "ranks" in it.name]
        rest = [it for it in items if it not in first]
        items[:] = first + rest
