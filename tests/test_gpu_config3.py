"""BASELINE.json configs[2] in full: "Batch=256 nasal-branch + frication active (consonant-heavy corpus), VTM2, 1xMI355X,
tolerance vs CPU" (SURVEY.md 8d "Config 3"; the reference side is VocalTractModel2<TFloat,1>, vtm/VocalTractModel2.h:626-669,
bit-identical to VocalTractModel0).

All 256 utterances x 500 frames against the oracle: exact sample counts; the fp64 path within 1e-9 of peak or one float32
ulp of the sample; the float path bit-identical to the float oracle; the mixed path within north_star's 1e-5.
(tests/tools/parity_report.py prints the same comparison as a JSON report.)"""
import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import oracle
import tracks

pytestmark = pytest.mark.gpu

BATCH, FRAMES = 256, 500


@pytest.fixture(scope="module")
def corpus():
    params = tracks.random_tracks(BATCH, FRAMES, seed0=2000, consonant_heavy=True)
    # the corpus is what configs[2] names: nasal branch open and frication on for about half of the key frames
    assert (params[:, :, 15] >= 0.5).mean() > 0.4 and (params[:, :, 3] >= 20.0).mean() > 0.3
    # (the references come from spawned worker processes that get their tracks explicitly: oracle.synthesize_many)
    refs64 = oracle.synthesize_many([(params[b], 44100.0, 1, 0, 0) for b in range(BATCH)])
    refs32 = oracle.synthesize_many([(params[b], 44100.0, 1, 0, 1) for b in range(BATCH)])
    return params, refs64, refs32


def _run(precision, params):
    plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 1, precision), 250.0, 0)
    return plan.synthesize_host(params)


def test_config3_fp64_all_256_against_the_double_oracle(corpus):
    params, refs64, _ = corpus
    audio, counts, maxabs = _run(capi.PRECISION_F64, params)
    assert (counts == 88108).all()
    worst, same = 0.0, 0
    for b in range(BATCH):
        r = refs64[b]
        assert r.size == 88108
        d = np.abs(audio[b].astype(np.float64) - r)
        ulp = np.spacing(np.abs(r)).astype(np.float64)
        peak = float(np.abs(r).max())
        assert (d <= np.maximum(ulp, 1e-9 * peak)).all(), (b, float(d.max() / peak))
        worst = max(worst, float(d.max() / peak))
        same += int(np.array_equal(audio[b], r))
    assert worst <= 6e-8      # one float32 ulp of a peak sample at most
    assert same >= BATCH * 9 // 10


def test_config3_float_all_256_bit_identical_to_the_float_oracle(corpus):
    params, _, refs32 = corpus
    audio, counts, _ = _run(capi.PRECISION_F32, params)
    assert (counts == 88108).all()
    for b in range(BATCH):
        assert np.array_equal(audio[b], refs32[b]), b


def test_config3_mixed_all_256_within_north_star_tolerance(corpus):
    params, refs64, _ = corpus
    audio, counts, _ = _run(capi.PRECISION_MIXED, params)
    assert (counts == 88108).all()
    worst = max(float(np.abs(audio[b].astype(np.float64) - refs64[b]).max() / np.abs(refs64[b]).max()) for b in range(BATCH))
    assert worst <= 1e-5, worst
