"""Drop-in boundary on the GPU: the GamaTTS plugin loaded by the REAL reference loader, and
the batched `gama_tts vtm` counterpart (C++ host over the C ABI)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import gama_tts_amd as g
import oracle

pytestmark = pytest.mark.gpu

LIBDIR = os.path.join(os.path.dirname(g.library_path()))
PLUGIN = os.path.join(LIBDIR, "libgama_vtm_plugin.so")
CLI = os.path.join(LIBDIR, "gama_vtm_batch")


@pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref/ref_vtm (the compiled reference) not present")
@pytest.mark.parametrize("name", ["hello_m0", "rand5_m0"])
def test_plugin_through_reference_loader(name, golden, tmp_path):
    """ref_vtm is the reference's own VocalTractModel::getInstance + VocalTractModelPlugin
    (dlopen/dlsym, vtm/VocalTractModelPlugin.cpp:57-91) driven by Controller::synthesize's loop;
    with model = 2000 every virtual call lands in our plugin object."""
    import golden_cases
    case = next(c for c in golden_cases.CASES if c["name"] == name)
    tr = golden_cases.track_for(case, golden)
    out, info = oracle.ref_synthesize(tr, "2000:" + PLUGIN, tmpdir=str(tmp_path))
    ref = golden[name + "__out"]
    assert out.size == ref.size == int(info["N"])
    assert float(info["fs"]) == 20034.0
    from test_gpu_parity import _within, _peak_err
    assert _within(out, ref, 1e-9), _peak_err(out, ref)


@pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref/ref_vtm (the compiled reference) not present")
def test_plugin_float_model_through_reference_loader(golden, tmp_path):
    """`gpu_precision = f32` makes the plugin stand in for model 1 (VocalTractModel0<float>)."""
    keys = oracle.read_config_file(oracle.VOICE_MALE)
    keys["gpu_precision"] = "f32"
    cfg = str(tmp_path / "vtm1.txt")
    with open(cfg, "w") as f:
        for k, v in keys.items():
            f.write("%s = %s\n" % (k, v))
    import golden_cases
    case = next(c for c in golden_cases.CASES if c["name"] == "rand5_m1")
    tr = golden_cases.track_for(case, golden)
    out, info = oracle.ref_synthesize(tr, "2000:" + PLUGIN, config=cfg, tmpdir=str(tmp_path))
    ref = golden["rand5_m1__out"]
    assert out.size == ref.size and float(info["fs"]) == 20034.0
    assert np.array_equal(out, ref)  # the float path is bit-identical to the float model


def test_plugin_thirty_section_tube_through_reference_loader(golden, tmp_path):
    """`tube_layout = 1` in vtm.txt makes the plugin stand in for model 4 (VocalTractModel4)."""
    keys = oracle.read_config_file(oracle.VOICE_MALE)
    keys["tube_layout"] = "1"
    cfg = str(tmp_path / "vtm4.txt")
    with open(cfg, "w") as f:
        for k, v in keys.items():
            f.write("%s = %s\n" % (k, v))
    import golden_cases
    case = next(c for c in golden_cases.CASES if c["name"] == "rand5_m4")
    tr = golden_cases.track_for(case, golden)
    out, info = oracle.ref_synthesize(tr, "2000:" + PLUGIN, config=cfg, tmpdir=str(tmp_path))
    ref = golden["rand5_m4__out"]
    assert out.size == ref.size and float(info["fs"]) == 60102.0
    from test_gpu_parity import _within, _peak_err
    assert _within(out, ref, 1e-9), _peak_err(out, ref)


@pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref/ref_vtm (the compiled reference) not present")
@pytest.mark.parametrize("poll", [64, 1024])
@pytest.mark.parametrize("precision,name", [("f64", "hello_m0"), ("f32", "rand5_m1")])
def test_plugin_interactive_protocol_through_reference_loader(precision, name, poll, golden, tmp_path):
    """interactive = true (VocalTractModelPlugin.cpp:87): the reference's loader constructs our object for the editor's
    caller contract, and the driver polls outputBuffer() after every execSynthesisStep() the way the JACK callback does
    (InteractiveAudio.cpp:141-185; oracle/ref_driver.cpp `poll=`).  The drained samples must be the batch protocol's."""
    keys = oracle.read_config_file(oracle.VOICE_MALE)
    keys["gpu_precision"] = precision
    cfg = str(tmp_path / "vtm_i.txt")
    with open(cfg, "w") as f:
        for k, v in keys.items():
            f.write("%s = %s\n" % (k, v))
    import golden_cases
    case = next(c for c in golden_cases.CASES if c["name"] == name)
    tr = golden_cases.track_for(case, golden)
    out, info = oracle.ref_synthesize(tr, "2000:" + PLUGIN, config=cfg, tmpdir=str(tmp_path), poll=poll)
    ref = golden[name + "__out"]
    assert out.size == ref.size == int(info["N"]) and int(info["callbacks"]) >= ref.size // poll
    if precision == "f32":
        assert np.array_equal(out, ref)
    else:
        from test_gpu_parity import _within, _peak_err
        assert _within(out, ref, 1e-9), _peak_err(out, ref)
    # and the same samples as our own batch protocol, bit for bit
    batch, _ = oracle.ref_synthesize(tr, "2000:" + PLUGIN, config=cfg, tmpdir=str(tmp_path))
    assert np.array_equal(out, batch)


@pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref/ref_vtm (the compiled reference) not present")
def test_plugin_consecutive_utterances_through_reference_loader(golden, tmp_path):
    """Two syntheses on one plugin object, reset() in between as Controller does (Controller.cpp:231): the second
    utterance must not inherit anything from the first (recorded steps, failure flags)."""
    import golden_cases
    case = next(c for c in golden_cases.CASES if c["name"] == "rand5_m0")
    tr = golden_cases.track_for(case, golden)
    out, info = oracle.ref_synthesize(tr, "2000:" + PLUGIN, tmpdir=str(tmp_path), repeat=3)
    ref = golden["rand5_m0__out"]
    assert out.size == ref.size
    from test_gpu_parity import _within, _peak_err
    assert _within(out, ref, 1e-9), _peak_err(out, ref)


def _make_voice_dir(root, model="0", rate=None):
    keys = oracle.read_config_file(oracle.VOICE_MALE)
    keys["model"] = model
    if rate is not None:
        keys["output_rate"] = repr(float(rate))
    os.makedirs(os.path.join(root, "variant"))
    variant_keys = ("vocal_tract_length", "glottal_pulse_tp", "glottal_pulse_tn_min", "glottal_pulse_tn_max",
                    "reference_glottal_pitch", "breathiness", "aperture_radius", "intonation_factor")
    with open(os.path.join(root, "_index.txt"), "w") as f:
        f.write("variant_dir = variant/\nvtm_control_model_file = vtm_control_model.txt\nvtm_file = vtm.txt\n")
    with open(os.path.join(root, "vtm.txt"), "w") as f:
        f.write("# test voice\n")
        for k, v in keys.items():
            if k not in variant_keys:
                f.write("%s = %s\n" % (k, v))
    with open(os.path.join(root, "variant", "male.txt"), "w") as f:
        for k in variant_keys:
            f.write("%s = %s\n" % (k, keys[k]))
    with open(os.path.join(root, "vtm_control_model.txt"), "w") as f:
        f.write("control_period = 4\nvariant_name = male\n")


def _read_wav(path):
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:16] == b"WAVEfmt "
    fmt = struct.unpack("<IHHIIHH", data[16:36])
    assert data[36:40] == b"data"
    n = struct.unpack("<I", data[40:44])[0]
    return fmt, np.frombuffer(data[44:44 + n], dtype="<i2")


@pytest.mark.parametrize("model,layout", [("0", 0), ("4", 1), ("1", 0)])
def test_batched_vtm_cli_writes_reference_wavs(model, layout, golden, tmp_path):
    """`gama_vtm_batch voice_dir out_dir a.txt b.txt` == `gama_tts vtm` per file: same frames in,
    same 16-bit samples out (scale 0.95/max, round(x * 32767), WAVEFileWriter.cpp:62-125)."""
    voice = str(tmp_path / "voice")
    _make_voice_dir(voice, model)
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    tracks_ = {"hello": np.asarray(golden["hello_params"]), "short": np.asarray(golden["hello_params"])[:40]}
    files = []
    for name, tr in tracks_.items():
        p = str(tmp_path / (name + ".txt"))
        with open(p, "w") as f:
            for row in tr:
                f.write(" ".join("%.9g" % v for v in row) + "\n")
        files.append(p)
    r = subprocess.run([CLI, voice, out_dir] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cfg = oracle.male_config(layout=layout, float_model=int(model == "1"))  # `model = 1` is the float model
    for name, tr in tracks_.items():
        fmt, pcm = _read_wav(os.path.join(out_dir, name + ".wav"))
        assert fmt == (16, 1, 1, 44100, 88200, 2, 16)
        ref = oracle.synthesize(cfg, tr)
        assert pcm.size == ref.size
        scaled = (ref * np.float32(oracle.output_scale(ref))) * np.float32(32767.0)
        want = (np.sign(scaled) * np.floor(np.abs(scaled) + np.float32(0.5))).astype(np.int32)
        if model == "1":
            # the float model's float32 samples are bit-identical to the reference's, hence the 16-bit stream is too
            assert np.array_equal(pcm.astype(np.int32), want)
        else:
            # fp64 path: a sample may differ from the reference's by one float32 ulp (tests/test_gpu_parity.py), which
            # can move a value across a rounding boundary: at most one LSB, on a handful of samples
            assert np.abs(pcm.astype(np.int32) - want).max() <= 1
            assert np.count_nonzero(pcm.astype(np.int32) != want) <= max(2, pcm.size // 10000)
    assert np.abs(_read_wav(os.path.join(out_dir, "hello.wav"))[1]).max() == 31129  # round(0.95 * 32767)


@pytest.mark.parametrize("name", ["hello_m1_48k", "hello_m1_44k", "short40_m1_44k", "hello_m0_48k", "hello_m0_44k", "hello_m4_44k"])
def test_batched_vtm_cli_against_wavs_the_reference_wrote(name, golden, golden_wav, tmp_path):
    """SURVEY.md 8(a) row a16 pinned to the reference: the fixture is the WAV `gama_tts vtm` itself wrote for the
    captured "Hello world" frames (Controller::synthesizeToFile -> writeOutputToFile -> WAVEFileWriter,
    tests/golden/make_wav_golden.py).  Float model: the whole FILE byte for byte.  Double models: header identical,
    samples within one LSB on at most 0.01 % of them."""
    m = golden_wav["manifest"][name]
    want_bytes = bytes(golden_wav[name + "__wav"])
    voice = str(tmp_path / "voice")
    _make_voice_dir(voice, m["model"], m["output_rate"])
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    tr = np.asarray(golden["hello_params"])[: m["frames"]]
    p = str(tmp_path / "utt.txt")
    with open(p, "w") as f:
        for row in tr:
            f.write(" ".join("%.9g" % v for v in row) + "\n")
    r = subprocess.run([CLI, voice, out_dir, p], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got_bytes = open(os.path.join(out_dir, "utt.wav"), "rb").read()
    assert len(got_bytes) == len(want_bytes) == m["bytes"]
    assert got_bytes[:44] == want_bytes[:44]  # RIFF / fmt / data headers
    if m["model"] == "1":
        assert got_bytes == want_bytes
    else:
        got = np.frombuffer(got_bytes[44:], dtype="<i2").astype(np.int32)
        want = np.frombuffer(want_bytes[44:], dtype="<i2").astype(np.int32)
        assert np.abs(got - want).max() <= 1
        assert np.count_nonzero(got != want) <= max(2, got.size // 10000)


def test_batched_vtm_cli_ragged_batch_with_a_flush_overrun_length(tmp_path):
    """Two files one frame apart on a down-sampling voice (model 3 at 44.1 kHz): 2334 frames is a length at which the
    reference's converter runs into its flush overrun (SampleRateConverter.h:298-308 with :462-471; pinned by the
    reference-made vectors of tests/test_gpu_overrun.py) and yields 411 798 samples, 575 MORE than the 2335-frame
    neighbour.  The CLI's rows must be sized with gvtm_output_capacity: with the longest utterance's count the first WAV
    would lose its tail (and read into the next row)."""
    voice = str(tmp_path / "voice")
    _make_voice_dir(voice, "3")
    out_dir = str(tmp_path / "out")
    os.makedirs(out_dir)
    import tracks
    pool = tracks.random_tracks(2, 2335, seed0=2334, consonant_heavy=True)
    tracks_ = {"overrun2334": pool[0, :2334], "plain2335": pool[1]}
    files = []
    for name, tr in tracks_.items():
        p = str(tmp_path / (name + ".txt"))
        with open(p, "w") as f:
            for row in tr:
                f.write(" ".join("%.9g" % v for v in row) + "\n")
        files.append(p)
    r = subprocess.run([CLI, voice, out_dir] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cfg = oracle.male_config(44100.0, 3)
    sizes = {}
    for name, tr in tracks_.items():
        fmt, pcm = _read_wav(os.path.join(out_dir, name + ".wav"))
        ref = oracle.synthesize(cfg, tr)
        sizes[name] = pcm.size
        assert pcm.size == ref.size
        scaled = (ref * np.float32(oracle.output_scale(ref))) * np.float32(32767.0)
        want = (np.sign(scaled) * np.floor(np.abs(scaled) + np.float32(0.5))).astype(np.int32)
        assert np.abs(pcm.astype(np.int32) - want).max() <= 1
        assert np.count_nonzero(pcm.astype(np.int32) != want) <= max(2, pcm.size // 10000)
    assert sizes == {"overrun2334": 411798, "plain2335": 411223}


def test_batched_cli_shards_across_device_slots(golden, tmp_path):
    """`-d 0,0,0`: three device slots (the same GPU three times on a one-GPU box), seven utterances -> contiguous
    shards of 3, 2, 2, each on its own host thread and plan (BASELINE configs[4] layout, no exchange between
    devices).  Every WAV must equal the one-device result byte for byte."""
    voice = str(tmp_path / "voice")
    _make_voice_dir(voice, "0")
    hello = np.asarray(golden["hello_params"])
    files = []
    for n, frames in enumerate([332, 40, 7, 120, 1, 250, 60]):
        p = str(tmp_path / ("u%d.txt" % n))
        with open(p, "w") as f:
            for row in hello[:frames]:
                f.write(" ".join("%.9g" % v for v in row) + "\n")
        files.append(p)
    outs = {}
    for tag, dev in (("one", "0"), ("three", "0,0,0")):
        out_dir = str(tmp_path / tag)
        os.makedirs(out_dir)
        r = subprocess.run([CLI, "-d", dev, voice, out_dir] + files, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[tag] = [open(os.path.join(out_dir, "u%d.wav" % n), "rb").read() for n in range(len(files))]
    assert outs["one"] == outs["three"]
    assert len(outs["one"][0]) > 44 and len(outs["one"][4]) > 44


@pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref/ref_vtm (the compiled reference) not present")
def test_plugin_model5_through_reference_loader(golden, golden5, tmp_path):
    """`gpu_model = 5` in vtm.txt makes the plugin stand in for reference model 5 (VocalTractModel5<double,1>); the
    reference's own loader and driver loop see its non-integer internal rate."""
    keys = oracle.read_config_file(oracle.VOICE5_MALE)
    keys["gpu_model"] = "5"
    cfg = str(tmp_path / "vtm5.txt")
    with open(cfg, "w") as f:
        for k, v in keys.items():
            f.write("%s = %s\n" % (k, v))
    import golden5_cases
    case = next(c for c in golden5_cases.CASES if c["name"] == "rand5_m5")
    tr = golden5_cases.track_for(case, golden)
    out, info = oracle.ref_synthesize(tr, "2000:" + PLUGIN, output_rate=48000, config=cfg, tmpdir=str(tmp_path))
    ref = golden5["rand5_m5__out"]
    assert out.size == ref.size == int(info["N"])
    assert abs(float(info["fs"]) - golden5["manifest"]["rand5_m5"]["fs"]) < 1e-6
    from test_gpu_model5 import _check
    _check(out, ref)


@pytest.mark.skipif(oracle.ref_binary() is None, reason="oracle/_ref/ref_vtm (the compiled reference) not present")
@pytest.mark.parametrize("poll", [64, 1024])
def test_plugin_model5_interactive_protocol_through_reference_loader(poll, golden, golden5, tmp_path):
    """interactive = true with `gpu_model = 5`: the reference's loader constructs the object for the editor's caller contract
    (VocalTractModelPlugin.cpp:87, InteractiveAudio.cpp:141-185) and polls outputBuffer() after every execSynthesisStep();
    behind it the steps go to a model-5 stream (VocalTractModel5's state between steps, vtm/VocalTractModel5.h:523-579, kept in
    device memory).  The drained samples must be the batch protocol's bit for bit, and the reference vector's."""
    keys = oracle.read_config_file(oracle.VOICE5_MALE)
    keys["gpu_model"] = "5"
    cfg = str(tmp_path / "vtm5i.txt")
    with open(cfg, "w") as f:
        for k, v in keys.items():
            f.write("%s = %s\n" % (k, v))
    import golden5_cases
    case = next(c for c in golden5_cases.CASES if c["name"] == "rand5_m5")
    tr = golden5_cases.track_for(case, golden)
    out, info = oracle.ref_synthesize(tr, "2000:" + PLUGIN, output_rate=48000, config=cfg, tmpdir=str(tmp_path), poll=poll)
    ref = golden5["rand5_m5__out"]
    assert out.size == ref.size == int(info["N"]) and int(info["callbacks"]) >= ref.size // poll
    from test_gpu_model5 import _check
    _check(out, ref)
    batch, _ = oracle.ref_synthesize(tr, "2000:" + PLUGIN, output_rate=48000, config=cfg, tmpdir=str(tmp_path))
    assert np.array_equal(out, batch)


def test_batched_vtm_cli_model5_voice(golden, tmp_path):
    """A voice directory whose vtm.txt says `model = 5` (the layout of data/voice/english/5_male): the batched CLI
    writes what `gama_tts vtm` writes for it."""
    voice = str(tmp_path / "voice5")
    keys = oracle.read_config_file(oracle.VOICE5_MALE)
    os.makedirs(os.path.join(voice, "variant"))
    variant_keys = ("vocal_tract_length", "glottal_pulse_tp", "glottal_pulse_tn_min", "glottal_pulse_tn_max",
                    "reference_glottal_pitch", "breathiness", "intonation_factor", "nasal_radius_2", "nasal_radius_3")
    with open(os.path.join(voice, "_index.txt"), "w") as f:
        f.write("variant_dir = variant/\nvtm_control_model_file = vtm_control_model.txt\nvtm_file = vtm.txt\n")
    with open(os.path.join(voice, "vtm.txt"), "w") as f:
        for k, v in keys.items():
            if k not in variant_keys:
                f.write("%s = %s\n" % (k, v))
    with open(os.path.join(voice, "variant", "male.txt"), "w") as f:
        for k in variant_keys:
            f.write("%s = %s\n" % (k, keys[k]))
    with open(os.path.join(voice, "vtm_control_model.txt"), "w") as f:
        f.write("control_period = 4\nvariant_name = male\n")
    out_dir = str(tmp_path / "out5")
    os.makedirs(out_dir)
    tracks_ = {"hello": np.asarray(golden["hello_params"]), "short": np.asarray(golden["hello_params"])[:40]}
    files = []
    for name, tr in tracks_.items():
        p = str(tmp_path / (name + "5.txt"))
        with open(p, "w") as f:
            for row in tr:
                f.write(" ".join("%.9g" % v for v in row) + "\n")
        files.append(p)
    r = subprocess.run([CLI, voice, out_dir] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cfg = oracle.male5_config(48000.0)
    for name, tr in tracks_.items():
        fmt, pcm = _read_wav(os.path.join(out_dir, name + "5.wav"))
        assert fmt == (16, 1, 1, 48000, 96000, 2, 16)
        ref, _ = oracle.synthesize5(cfg, tr)
        assert pcm.size == ref.size
        scaled = (ref * np.float32(oracle.output_scale(ref))) * np.float32(32767.0)
        want = (np.sign(scaled) * np.floor(np.abs(scaled) + np.float32(0.5))).astype(np.int32)
        assert np.abs(pcm.astype(np.int32) - want).max() <= 1
        assert np.mean(pcm.astype(np.int32) == want) > 0.999
