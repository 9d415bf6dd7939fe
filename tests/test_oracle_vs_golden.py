"""Pins the CPU oracle: it must reproduce every committed reference vector bit for bit.

The vectors in tests/golden/vtm_golden.npz were produced by the real GamaTTS classes
(oracle/_ref/ref_vtm, see tests/golden/make_golden.py); the reference itself ships no
tests for this path, so these are the pins.
"""
import hashlib

import numpy as np
import pytest

import golden_cases
import oracle
import tracks

# SURVEY.md section 0 known-answer table (reference built -O2 -ffp-contract=off)
KAT = {
    "const_m0": (88108, 7.871065063e+00, 7.537760539e-04),
    "ramp_m0": (88108, 3.076802642e+00, 1.228036941e-03),
    "const_m3": (88077, 1.073991490e+01, 1.668861834e-03),
    "ramp_m3": (88077, 4.196292139e+00, 2.267686650e-03),
    "const_m1": (88108, 7.871396222e+00, 7.547301939e-04),  # model 1 = VocalTractModel0<float>
    "ramp_m1": (88108, 3.077012386e+00, 1.231163274e-03),
    "const_m4": (88077, 1.073981937e+01, 1.675571664e-03),
    "ramp_m4": (88077, 4.196179109e+00, 2.232487779e-03),
}


def _oracle_config(case):
    base = oracle.read_config_file(oracle.VOICE_MALE)
    base.update({k: str(v) for k, v in case["overrides"].items()})
    return oracle.config_from_dict(base, case["rate"], case["delay"], case.get("layout", 0), case.get("float_model", 0))


@pytest.mark.parametrize("case", golden_cases.CASES, ids=lambda c: c["name"])
def test_oracle_matches_reference_vector(case, golden):
    m = golden["manifest"][case["name"]]
    tr = golden_cases.track_for(case, golden)
    cfg = _oracle_config(case)
    d = oracle.derive(cfg, case["crate"])
    assert d.sample_rate == int(m["fs"])
    assert d.control_steps * tr.shape[0] == m["steps"]
    assert oracle.output_count(cfg, tr.shape[0], case["crate"]) == m["n"]
    out = oracle.synthesize(cfg, tr, case["crate"])
    assert out.size == m["n"]
    assert hashlib.sha256(out.tobytes()).hexdigest() == m["sha256"]
    if case["store"] == "full":
        assert np.array_equal(out, golden[case["name"] + "__out"])
    else:
        assert np.array_equal(out[:: golden_cases.DIGEST_STRIDE], golden[case["name"] + "__strided"])


@pytest.mark.parametrize("name", sorted(KAT))
def test_survey_known_answers(name, golden):
    n, total, peak = KAT[name]
    m = golden["manifest"][name]
    assert m["n"] == n
    assert m["sum"] == pytest.approx(total, rel=1e-9)
    assert m["maxabs"] == pytest.approx(peak, rel=1e-9)


def test_derived_constants_male_model0():
    # SURVEY.md E13
    d = oracle.derive(oracle.male_config(44100.0), 250.0)
    assert (d.sample_rate, d.control_steps, d.fir_taps) == (20034, 80, 49)
    assert (d.table_div1, d.table_div2, d.tn_delta) == (205, 328, 0.0)
    assert (d.time_register_increment, d.pad_size, d.upsampling) == (29772, 13, 1)
    d = oracle.derive(oracle.male_config(48000.0), 250.0)
    assert d.time_register_increment == 27353
    d = oracle.derive(oracle.male_config(44100.0, section_delay=3), 250.0)
    assert (d.sample_rate, d.time_register_increment, d.phase_increment, d.pad_size, d.upsampling) == \
        (60102, 89316, 48087, 18, 0)


def test_output_counts_long_form():
    # 7500 frames (30 s)
    assert oracle.output_count(oracle.male_config(44100.0), 7500) == 1320815
    assert oracle.output_count(oracle.male_config(44100.0, section_delay=3), 7500) == 1320785
    assert oracle.output_count(oracle.male_config(44100.0), 0) == 58  # flush of an empty utterance: ceil(26 * 65536 / 29772)


def test_output_scale_rule():
    # Util::calculateOutputScale: 0.95 / max|x|, and 0 below 1e-30 (VTMUtil.cpp:48-67)
    x = np.array([0.1, -0.5, 0.25], dtype=np.float32)
    assert oracle.output_scale(x) == pytest.approx(np.float32(0.95) / np.float32(0.5))
    assert oracle.output_scale(np.zeros(8, dtype=np.float32)) == 0.0


def test_noise_sequence_is_utterance_independent():
    import ctypes
    buf = np.empty(64, dtype=np.float64)
    oracle.lib().vtmo_noise_sequence(buf.ctypes.data, buf.size)
    seed = 0.7892347
    x1 = 0.0
    for i in range(64):
        p = seed * 377.0
        seed = p - int(p)
        w = seed - 0.5
        assert buf[i] == w + x1
        x1 = w


def _overrun_oracle(case):
    tr = golden_cases.track_for(case)
    if case["model5"]:
        cfg = oracle.male5_config(case["rate"])
        return tr, oracle.synthesize5(cfg, tr)[0]
    cfg = oracle.config_from_dict(oracle.read_config_file(oracle.VOICE_MALE), case["rate"], case["delay"], case.get("layout", 0),
                                  case.get("float_model", 0))
    return tr, oracle.synthesize(cfg, tr, case["crate"])


@pytest.mark.parametrize("case", golden_cases.OVERRUN_CASES, ids=lambda c: c["name"])
def test_oracle_matches_reference_at_flush_overrun_lengths(case, golden_overrun):
    """The reference's SampleRateConverter converts one more lap of its ring at these lengths
    (SampleRateConverter.h:298-308 with :462-471); the oracle walks the ring literally and must give the same bytes."""
    m = golden_overrun["manifest"][case["name"]]
    tr, out = _overrun_oracle(case)
    assert out.size == m["n"]
    assert hashlib.sha256(out.tobytes()).hexdigest() == m["sha256"]
    # the sample count really is the overrun's: one ring (1024 input samples) more than the closed form, and the
    # product's count function (no GPU needed) agrees
    import gama_tts_amd as g
    from gama_tts_amd import capi
    if case["model5"]:
        plan = g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE), case["rate"]), case["crate"], capi.DEVICE_NONE)
    else:
        plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), case["rate"], case["delay"],
                                         capi.PRECISION_F32 if case["float_model"] else capi.PRECISION_F64, case.get("layout", 0)),
                      case["crate"], capi.DEVICE_NONE)
    i = plan.info
    fills = tr.shape[0] * i.control_steps + 2 * i.pad_size
    closed = -((-(fills << 16)) // i.time_register_increment)
    lap = -((-((fills + 1024) << 16)) // i.time_register_increment)
    assert plan.output_count(tr.shape[0]) == m["n"] == lap > closed
    if case["store"] == "full":
        assert np.array_equal(out, golden_overrun[case["name"] + "__out"])
    else:
        assert np.array_equal(out[:: golden_cases.DIGEST_STRIDE], golden_overrun[case["name"] + "__strided"])
        assert np.array_equal(out[-golden_cases.OVERRUN_TAIL:], golden_overrun[case["name"] + "__tail"])
        assert np.abs(out[-600:]).max() > 0  # the extra lap is not silence


def _wav_fields(data):
    import struct
    assert data[:4] == b"RIFF" and data[8:16] == b"WAVEfmt " and data[36:40] == b"data"
    riff_len = struct.unpack("<I", data[4:8])[0]
    fmt = struct.unpack("<IHHIIHH", data[16:36])
    n = struct.unpack("<I", data[40:44])[0]
    assert riff_len == len(data) - 8 and n == len(data) - 44
    return fmt, np.frombuffer(data[44:], dtype="<i2")


def pcm16_like_the_reference(x):
    """Controller::writeOutputToFile (Controller.cpp:315-328): scale = 0.95 / max|x| (float), sample * scale (float);
    WAVEFileWriter::writeSample (WAVEFileWriter.cpp:122-125): std::round(v * 32767.0f) to int16."""
    scale = np.float32(oracle.output_scale(x))
    v = (x * scale) * np.float32(32767.0)
    return (np.sign(v) * np.floor(np.abs(v) + np.float32(0.5))).astype(np.int16)


@pytest.mark.parametrize("name", ["hello_m0_48k", "hello_m1_48k", "hello_m0_44k", "hello_m1_44k", "short40_m1_44k", "hello_m4_44k"])
def test_reference_written_wav(name, golden, golden_wav):
    """N, WAV header and int16 samples of the reference's own `gama_tts vtm` run on the captured
    "Hello world" frames.  Pins the oracle AND the restatement of the scaling / rounding rule bit for bit."""
    m = golden_wav["manifest"][name]
    data = bytes(golden_wav[name + "__wav"])
    assert hashlib.sha256(data).hexdigest() == m["sha256"]
    fmt, pcm = _wav_fields(data)
    rate = int(m["output_rate"])
    assert fmt == (16, 1, 1, rate, rate * 2, 2, 16)  # PCM, mono, 16 bit
    tr = np.asarray(golden["hello_params"])[: m["frames"]]
    layout = 1 if m["model"] == "4" else 0
    cfg = oracle.male_config(m["output_rate"], 1, layout, float_model=int(m["model"] == "1"))
    x = oracle.synthesize(cfg, tr)
    assert pcm.size == x.size
    assert np.array_equal(pcm16_like_the_reference(x), pcm)
    assert np.abs(pcm).max() == 31129  # round(0.95 * 32767)
