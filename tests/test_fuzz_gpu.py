"""A short run of tools/fuzz_parity.py (randomized differential test of Join A: every strategy x mode x invert x output
set against the oracle on random index shapes and region mixes); the long campaigns are run by hand on the GPU box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_parity_short(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "25", str(seed)], cwd=ROOT,
                       capture_output=True, timeout=600)
    assert r.returncode == 0 and b"fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_fuzz_lines_short():
    """tools/fuzz_lines.py: Join B, covered bases and depth against their definitions on random inputs."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_lines.py"), "40", "7"], cwd=ROOT,
                       capture_output=True, timeout=600)
    assert r.returncode == 0 and b"fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_fuzz_cli_short():
    """tools/fuzz_cli.py: index + intersect / depth / coverage with random inputs and flags against the oracle's commands."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cli.py"), "5", "9"], cwd=ROOT,
                       capture_output=True, timeout=900)
    assert r.returncode == 0 and b"fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
