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


# ---- measured parity errors: every rel-L2 a GPU test computes through oracle.rel_l2 is recorded with the test id and
# the source line that asserts on it; a GPU session writes the table to gpurun_out/parity_errors.json (copied to profiles/rNN/ by hand).
_CURRENT = {"id": None}
_RECORDS = []


def pytest_runtest_setup(item):
    _CURRENT["id"] = item.nodeid


def _install_recorder():
    import linecache
    import vit_oracle as O
    if getattr(O.rel_l2, "_recording", False):
        return
    plain = O.rel_l2

    def rel_l2(a, b):
        v = plain(a, b)
        f = sys._getframe(1)
        if _CURRENT["id"] and "test_gpu" in f.f_code.co_filename:
            _RECORDS.append({"test": _CURRENT["id"], "line": f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}",
                             "source": linecache.getline(f.f_code.co_filename, f.f_lineno).strip()[:160], "rel_l2": float(v)})
        return v

    rel_l2._recording = True
    O.rel_l2 = rel_l2


_install_recorder()


def pytest_sessionfinish(session, exitstatus):
    if not _RECORDS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    worst = {}
    for r in _RECORDS:                      # one row per asserting source line: the worst value any parametrisation produced
        w = worst.setdefault(r["line"], dict(r, n=0))
        w["n"] += 1
        if r["rel_l2"] > w["rel_l2"]:
            w.update(test=r["test"], rel_l2=r["rel_l2"])
    with open(os.path.join(out, "parity_errors.json"), "w") as f:
        json.dump({"per_assert_worst": sorted(worst.values(), key=lambda r: r["line"]), "all": _RECORDS}, f, indent=1)
