import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "longcat-video-tta_amd"
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _library_knobs_follow_the_environment(monkeypatch):
    """liblcv_hip.so reads its LCV_* A/B knobs once (include/lcv_hip.h).  Tests flip them with `monkeypatch.setenv`: re-read after
    every such change, and at the start of every test (monkeypatch has restored the environment by then)."""
    def reload():
        lib = sys.modules.get("lcv_hip.lib")
        if lib is not None and getattr(lib, "_lib", None) is not None:
            lib._lib.lcv_knobs_reload()
    reload()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith("LCV_"):
            reload()

    def delenv_and_reload(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith("LCV_"):
            reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reload, delenv_and_reload
    yield


_PARITY = {}


def _record_parity(value, bound, extra=None):
    """Every rel_l2() of a GPU run lands in gpurun_out/kernel_parity.json: test id -> [{line, value, bound}, ...] in call order (the
    evidence behind the tolerances: profiles/r04_kernel_parity.json is a copy of one such run)."""
    import inspect
    import json
    import torch
    if not torch.cuda.is_available():
        return
    node = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    fr = inspect.stack()[2]
    row = {"line": f"{Path(fr.filename).name}:{fr.lineno}", "rel_l2": float(f"{value:.4g}")}
    if bound is not None:
        row["bound"] = bound
    if extra:
        row.update(extra)
    _PARITY.setdefault(node, []).append(row)
    out = Path(os.environ.get("GRAFT_REPO_ROOT", ROOT)) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "kernel_parity.json").write_text(json.dumps(_PARITY, indent=0))
    except OSError:
        pass


def rel_l2(a, b, bound=None):
    """Relative L2 distance of `a` from the reference `b`.  `bound` (the number the caller asserts against) is only recorded."""
    import torch
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    v = (torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b).clamp_min(1e-30)).item()
    _record_parity(v, bound)
    return v
