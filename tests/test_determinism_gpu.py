"""ASRK_DETERMINISTIC=1: bit-reproducible training steps (f32 atomics made gradients differ run to
run, while the reference's CPU path is reproducible).  In that mode the library never lets the ORDER of f32 additions
depend on scheduling: GEMMs do not split K across workgroups, column sums and LayerNorm parameter gradients use one row
chunk per column block, the cross-entropy sum and the embedding gradient run in a fixed order (csrc/knobs.h).
Covered: the fused decoder loop + LSTM encoders (the default path of the shipped configurations) at batch <= 32; the
per-step attention kernels of the non-fused variants still accumulate with atomics."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(env_extra, n=3):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "determinism_worker.py"), str(n)], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.splitlines() if l.startswith("RUN")]
    assert len(rows) == n
    return [x[2] for x in rows], [float(x[3]) for x in rows]


def test_deterministic_mode_gives_bit_identical_steps():
    digests, losses = _run({"ASRK_DETERMINISTIC": "1"})
    assert len(set(digests)) == 1, digests
    # and the mode does not change the arithmetic beyond summation order: same loss as the default mode to 1e-6
    _, default_losses = _run({"ASRK_DETERMINISTIC": "0"}, n=1)
    assert abs(default_losses[0] - losses[0]) < 1e-5 * abs(losses[0])
