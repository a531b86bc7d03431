"""The window index of k_join_pairs / k_join_roots (32-byte lines, 16-bit window-relative coordinates, list tails in win_spill, sweep-only
seqids) against a brute-force scan, on the CPU: tools/win_index_check.hip includes the engine's builder and restates how the
kernel reads a line -- the narrow form's one line per region in all three modes, and the mixed form's reading of a wide region (two
lines, two ranks) in every mode, inverted or not: what a wide lane keeps of the line of its first base and of its run, and the true
ends by position where the lines' clamped 16-bit ends do not tell (round 5).  The GPU parity tests cover the kernel itself; this
covers the tables it reads, and the arithmetic on them, where no GPU exists."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "gffx_amd", "bin", "win_index_check")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_window_lines_answer_like_brute_force(seed):
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    r = subprocess.run([BIN, str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok ("), r.stdout[-500:] + r.stderr[-500:]
