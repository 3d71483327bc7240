import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "generic-diffusion-feature_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---- parity margins under the driver's clock ------------------------------------------------------------------------
# `pytest -q` shows dots; the measured margins behind the assertions (worst hook, its error, the bound it was held to) used to exist only
# in builder-run logs.  Full-size tests call record_margin(...); the block below is printed after the dots, so it lands in the tail of the
# driver's own `pytest -m gpu` run.  Reference contract being promised: feature/components/feature_extractor.py:31-76 (FeatureStore hands
# out what the model computed) within the north star's 1e-3.
_MARGINS = []


def record_margin(config_name, worst_id, err, bound, extra=""):
    _MARGINS.append((str(config_name), str(worst_id), float(err), float(bound), str(extra)))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _MARGINS:
        return
    tr = terminalreporter
    tr.write_line("")
    tr.write_line("== parity margins: HIP (libgdf.so, C ABI) vs the fp32 CPU oracle — relative L2 of the WORST hook per configuration ==")
    for name, hid, err, bound, extra in _MARGINS:
        flag = "" if err <= bound else "  <-- ABOVE BOUND"
        tr.write_line(f"  {name:<68s} {err:.2e} / {bound:.2e} ({100.0 * (1.0 - err / bound):4.1f} % spare)  {hid}" + (f"  [{extra}]" if extra else "") + flag)
