"""Pins the VocalTractModel5 restatement (oracle/vtm_oracle_body.inc, v5_*) bit for bit against vectors of
the real reference classes (tests/golden/vtm5_golden.npz, made by tests/golden/make_vtm5_golden.py).

Oracle only: the device path does not serve reference model 5 yet; these
pins are what a later device kernel will be tested against.
"""
import hashlib

import numpy as np
import pytest

import golden5_cases
import oracle


def _config(case):
    base = oracle.read_config_file(oracle.VOICE5_MALE)
    base.update({k: str(v) for k, v in case["overrides"].items()})
    return oracle.config5_from_dict(base, case["rate"], case["float_model"])


@pytest.mark.parametrize("case", golden5_cases.CASES, ids=lambda c: c["name"])
def test_oracle5_matches_reference_vector(case, golden, golden5):
    m = golden5["manifest"][case["name"]]
    tr = golden5_cases.track_for(case, golden)
    out, rate = oracle.synthesize5(_config(case), tr, case["crate"])
    assert abs(rate - m["fs"]) < 2e-3  # the internal rate is not an integer (VocalTractModel5.h:465)
    assert round(m["fs"] / case["crate"]) * tr.shape[0] == m["steps"]
    assert out.size == m["n"]
    assert hashlib.sha256(out.tobytes()).hexdigest() == m["sha256"]
    key = case["name"] + ("__out" if case["store"] == "full" else "__strided")
    assert np.array_equal(out if case["store"] == "full" else out[:: golden5_cases.DIGEST_STRIDE], golden5[key])


def test_survey_known_answer_model5(golden5):
    # model 5, 5_male data, const track, 44.1 kHz
    m = golden5["manifest"]["const_m5_44k"]
    assert m["n"] == 88356
    assert m["sum"] == pytest.approx(7.932018498e+00, rel=1e-9)
    assert m["maxabs"] == pytest.approx(1.288747461e+04, rel=1e-9)


def test_model5_internal_rate():
    # speed of sound at 35 C x 30 sections x 100 / 17.5 cm (VocalTractModel5.h:462-465)
    _, rate = oracle.synthesize5(oracle.male5_config(), np.zeros((0, 16), np.float32))
    assert rate == pytest.approx((331.4 + 0.6 * 35.0) * 30 * 100 / 17.5, abs=2e-3)


@pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("model,fm", [("5", 0), ("5f", 1)])
def test_oracle5_matches_reference_binary_on_fresh_tracks(model, fm):
    import tracks
    for seed in (31, 32):
        tr = tracks.random_track(60, seed, seed % 2 == 0)
        ref, _ = oracle.ref_synthesize(tr, model, 48000, 250, config=oracle.VOICE5_MALE)
        out, _ = oracle.synthesize5(oracle.male5_config(48000.0, fm), tr)
        assert np.array_equal(out, ref)
