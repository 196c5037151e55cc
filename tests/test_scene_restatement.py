"""The library's scene generators against the oracle's independent numpy restatement of src/main.rs:42-82.

Host code only (no GPU): rt_scene_rtiow / rt_scene_field fill arrays on the CPU.
"""
import numpy as np
import pytest

import raytracer_amd as rt


def test_vectorised_philox_equals_the_scalar_one(oracle):
    rng = np.random.default_rng(1)
    for _ in range(64):
        seed = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        pixel, sample, slot, dom = (int(rng.integers(0, 2 ** 32)) for _ in range(4))
        a = oracle.philox_pair(seed, pixel, sample, slot, dom)
        b0, b1 = oracle.philox_pairs(seed, [pixel], sample, slot, dom)
        assert a[0] == b0[0] and a[1] == b1[0]
    # Random123's known-answer vector for Philox4x32-10: counter and key all ones
    out = oracle.philox4x32_10([0xFFFFFFFF, 0xFFFFFFFF], [0xFFFFFFFF] * 4)
    assert [int(x) for x in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


@pytest.mark.parametrize("seed", [1, 2, 3, 0x1234567890ABCDEF])
def test_rtiow_scene_equals_oracle_restatement(oracle, seed):
    spheres, materials = rt.scene_rtiow(seed)
    osp, omt = oracle.scene_rtiow(seed)
    assert spheres.tobytes() == osp.tobytes()
    assert materials.tobytes() == omt.tobytes()
    # src/main.rs:45-55: the four fixed objects; src/main.rs:57-82: 22 x 22 small ones, all at y = 0.2, r = 0.2
    assert len(spheres) == 488 and (spheres["radius"][4:] == 0.2).all() and (spheres["center"][4:, 1] == 0.2).all()
    assert tuple(spheres["center"][0]) == (0.0, -1000.0, 0.0) and spheres["radius"][0] == 1000.0
    kinds = np.bincount(materials["type"][4:], minlength=3)
    assert (kinds > 100).all()  # uniform thirds (src/main.rs:64), not the 80/15/5 mix of the book


@pytest.mark.parametrize("case", [(0, 1 / 3, 1 / 3, 1.0, 3), (1, 0.0, 0.5, 0.3, 5), (2500, 0.0, 0.5, 0.3, 5), (10000, 0.0, 0.5, 0.3, 5),
                                  (100000, 1 / 3, 1 / 3, 1.0, 3), (1000000, 0.15, 0.15, 1.0, 4)])
def test_field_scene_equals_oracle_restatement(oracle, case):
    n, pl, pm, fz, seed = case
    spheres, materials = rt.scene_field(seed, n, pl, pm, fz)
    osp, omt = oracle.scene_field(seed, n, pl, pm, fz)
    assert spheres.tobytes() == osp.tobytes()
    assert materials.tobytes() == omt.tobytes()
