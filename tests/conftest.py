"""Shared test plumbing: path setup, the `gpu` marker, golden-fixture loader."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vit-is-all-you-need_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    # fixtures are plain dicts of tensors / numbers written by oracle/gen_golden.py
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def hip():
    """The C-ABI library, loaded; GPU tests fail (not skip) when it is missing."""
    from vitamd import lib
    return lib.load()
