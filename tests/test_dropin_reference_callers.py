"""Authoring container only (skipped where /root/reference does not exist, e.g. on the GPU box): the reference's CALLERS —
nerf/renderer.py (`run_cuda`, `update_extra_state`) and nerf/network.py — imported unmodified on top of the build's drop-in
packages (seal-3d_amd/{raymarching, gridencoder, shencoder, encoding.py, activation.py}; CPU oracle as the native backend)
reproduce tests/golden/wrappers.npz bit for bit.  Runs `oracle/gen_golden.py dropin` in its own process (its import roots
differ from the test session's)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/nerf"), reason="reference checkout not present")


def test_reference_callers_run_unmodified_on_the_dropin_packages():
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "gen_golden.py"), "dropin"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "reproduce wrappers.npz" in r.stdout
