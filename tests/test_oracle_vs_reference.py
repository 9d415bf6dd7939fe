"""Oracle vs the REAL reference binary on fresh random tracks.

oracle/_ref/ref_vtm is compiled from /root/reference by oracle/Makefile in the build
container and travels to the GPU box as a binary; when it is absent (a checkout that
never saw the reference) these tests skip and the committed goldens remain the pin.
"""
import numpy as np
import pytest

import oracle
import tracks

pytestmark = pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref/ref_vtm not built")


@pytest.mark.parametrize("model,delay,layout,fm", [("0", 1, 0, 0), ("2", 1, 0, 0), ("2:2", 2, 0, 0), ("3", 3, 0, 0), ("4", 1, 1, 0),
                                                    ("1", 1, 0, 1), ("2f:2", 2, 0, 1), ("2f:4", 4, 0, 1), ("4f", 1, 1, 1)])
@pytest.mark.parametrize("seed", [11, 12])
def test_bit_identical_on_random_tracks(model, delay, layout, fm, seed, tmp_path):
    tr = tracks.random_track(90, seed, consonant_heavy=bool(seed & 1))
    ref, info = oracle.ref_synthesize(tr, model, tmpdir=str(tmp_path))
    out = oracle.synthesize(oracle.male_config(section_delay=delay, layout=layout, float_model=fm), tr)
    assert out.size == int(info["N"])
    assert np.array_equal(out, ref)
