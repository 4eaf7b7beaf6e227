import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "seal-3d_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle_backend as ob
    ob.build()
    return ob


@pytest.fixture()
def oracle_wrappers(oracle, monkeypatch):
    """The build's drop-in Python packages driven by the CPU oracle instead of the HIP library —
    lets the host logic (padding, counters, zero-init contracts, autograd plumbing) run without a GPU."""
    import raymarching.raymarching as rm
    import gridencoder.grid as gg
    import shencoder.sphere_harmonics as sh
    import freqencoder.freq as fq
    import ffmlp.ffmlp as ff
    monkeypatch.setattr(rm, "_backend", oracle.RaymarchingBackend)
    monkeypatch.setattr(gg, "_backend", oracle.GridBackend)
    monkeypatch.setattr(sh, "_backend", oracle.SHBackend)
    monkeypatch.setattr(fq, "_backend", oracle.FreqBackend)
    monkeypatch.setattr(ff, "_backend", oracle.FFMLPBackend)
    import types
    return types.SimpleNamespace(rm=rm, gg=gg, sh=sh, fq=fq, ff=ff)


@pytest.fixture(scope="session")
def hip():
    """The product backend; fails loudly when the extension is missing or there is no GPU."""
    import s3d_hip
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    s3d_hip.lib()
    return s3d_hip


GOLDEN = os.path.join(REPO, "tests", "golden")
