"""Host side of reference model 5 without a GPU: a design-only plan (GVTM_DEVICE_NONE) derives the same rates,
driver-loop constants and output lengths as the real VocalTractModel5 did for the committed vectors, refuses what the
reference's constructors refuse, and has no CPU synthesis path."""
import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import golden5_cases
import oracle


def _plan(case):
    d = g.read_config_file(oracle.VOICE5_MALE)
    d.update({k: str(v) for k, v in case["overrides"].items()})
    return g.Plan(g.config5_from_dict(d, case["rate"]), case["crate"], capi.DEVICE_NONE)


@pytest.mark.parametrize("case", [c for c in golden5_cases.CASES if not c["float_model"]], ids=lambda c: c["name"])
def test_design_matches_the_reference_vectors(case, golden, golden5):
    m = golden5["manifest"][case["name"]]
    tr = golden5_cases.track_for(case, golden)
    plan = _plan(case)
    i = plan.info
    assert i.model5 == 1 and i.precision == capi.PRECISION_F64
    assert abs(i.internal_rate_hz - m["fs"]) < 1e-9 and i.internal_sample_rate == int(m["fs"])
    assert i.control_steps * tr.shape[0] == m["steps"]
    assert plan.output_count(tr.shape[0]) == m["n"]
    assert i.output_rate == case["rate"] and i.upsampling == int(case["rate"] >= m["fs"])


def test_resampler_tables_are_the_double_tables_of_the_other_models():
    p5 = g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE)), 250.0, capi.DEVICE_NONE)
    p0 = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE)), 250.0, capi.DEVICE_NONE)
    for which in (capi.TABLE_SRC_H, capi.TABLE_SRC_DH):
        assert np.array_equal(p5.table(which), p0.table(which))
    with pytest.raises(g.GvtmError):  # no wavetable / FIR in model 5 (Rosenberg source)
        p5.table(capi.TABLE_FIR)


def test_rejections_and_no_cpu_path():
    d = g.read_config_file(oracle.VOICE5_MALE)
    for key, value in (("output_rate", "0"), ("vocal_tract_length", "25.0"),      # internal rate below 50 kHz
                       ("glottal_pulse_tn_min", "30.0"), ("glottal_pulse_tp", "0.5"),  # RosenbergBGlottalSource's checks
                       ("glottal_noise_cutoff", "0.5"), ("frication_noise_cutoff", "40000"),  # Butterworth update() range
                       ("nasal_radius_4", "0"), ("mix_offset", "0")):
        with pytest.raises(g.GvtmError) as ei:
            g.Plan(g.config5_from_dict(dict(d, **{key: value})), 250.0, capi.DEVICE_NONE)
        assert ei.value.status == 1, key
    with pytest.raises(g.GvtmError):
        g.Plan(g.config5_from_dict(d, precision=capi.PRECISION_MIXED), 250.0, capi.DEVICE_NONE)
    plan = g.Plan(g.config5_from_dict(d), 250.0, capi.DEVICE_NONE)
    with pytest.raises(g.GvtmError) as ei:
        plan.synthesize_host(np.zeros((1, 2, 16), np.float32))
    assert ei.value.status == 2  # GVTM_ERR_NO_DEVICE


def test_output_counts_follow_the_reference_converter():
    """Every frame count gives the oracle's (= the reference converter's) sample count, the two lengths at which the
    reference converts a ring of stale samples more included (SampleRateConverter's flush overrun)."""
    d = g.read_config_file(oracle.VOICE5_MALE)
    overruns = 0
    for rate, frames in ((44100.0, list(range(0, 120, 7)) + [105, 106, 107]), (22050.0, [1, 50, 695, 696, 697]), (48000.0, [0, 1, 2, 3, 100])):
        plan = g.Plan(g.config5_from_dict(d, rate), 250.0, capi.DEVICE_NONE)
        cfg = oracle.male5_config(rate)
        ratio = rate / plan.info.internal_rate_hz
        for f in frames:
            want = oracle.synthesize5(cfg, np.zeros((f, 16), np.float32))[0].size
            assert plan.output_count(f) == want, (rate, f)
            closed = (f * plan.info.control_steps + 2 * plan.info.pad_size) * ratio
            if want - closed > 2:
                overruns += 1
                assert abs((want - closed) / ratio - 1024) < 16, (rate, f, want, closed)  # one ring of stale input more
    assert overruns == 2  # 106 frames at 44.1 kHz, 696 at 22.05 kHz
