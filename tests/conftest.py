import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mri-implicit-neural-representations_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))  # a GPU box exposes 256 cores but gives one job a 16-core share


def pytest_configure(config):
    import torch
    torch.set_num_threads(_usable_cores())
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def record_parity(test: str, **values) -> None:
    """Append the measured errors behind a parity criterion (e_gpu = HIP path vs float64 oracle, e_cpu = the oracle's
    own fp32 evaluation vs float64, ...) to gpurun_out/parity_errors.jsonl, so the margins the criteria leave are
    on record (profiles/r*_parity_errors.jsonl is a committed copy).  Never fails a test."""
    import json
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_errors.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=test, **{k: (float(v) if isinstance(v, (int, float)) else v)
                                                  for k, v in values.items()})) + "\n")
    except OSError:
        pass
