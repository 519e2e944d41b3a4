"""The experimental encoder launch strategies (LDS sweep kernel csrc/msda_sweep.hip, 2-D patch mapping) are
are
selected by an environment variable that the library reads once, so they are exercised in child processes: the
encoder-shape parity tests are re-run with RDETR_MSDA_ALGO=sweep, =tile2d and =hybrid (default: qrun, the direct kernel)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("algo", ["sweep", "tile2d", "hybrid"])
def test_experimental_encoder_kernels_parity_in_subprocess(algo):
    env = dict(os.environ, RDETR_MSDA_ALGO=algo)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu",
                        "-k", "encoder_entry or full_size or module_golden", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
