"""The wavefront measurement harness (rt_experiment_wavefront_device, include/rt_experiments.h) renders the same paths
as the persistent kernel.  The two are separate kernels, so the compiler contracts their f64 arithmetic differently here
and there: a last-bit difference in one scatter direction can send one deep path elsewhere (as between the kernels and the
oracle), hence the same bar as the oracle comparisons: nearly all pixels equal to 1e-9, ray counts equal to 0.1 %.

The harness is not part of the shipped library: the check runs in a child process that loads
libraytracer_amd_experiments.so instead (tests/wavefront_check.py)."""
import os
import subprocess
import sys

import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_shipped_library_has_no_experiment():
    assert not hasattr(rt._lib.lib(), "rt_experiment_wavefront_device")


@pytest.mark.parametrize("case", ["rtiow", "field"])
def test_wavefront_harness_renders_the_same_frame(case):
    env = dict(os.environ, RAYTRACER_AMD_LIB=rt._lib.EXPERIMENTS_LIB_PATH)
    p = subprocess.run([sys.executable, os.path.join(HERE, "wavefront_check.py"), case], env=env, capture_output=True, text=True, timeout=280)
    print(p.stdout)
    assert p.returncode == 0, p.stderr[-2000:]
