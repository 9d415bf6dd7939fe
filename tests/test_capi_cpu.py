"""Host-side checks that need no GPU: the library loads, exports the whole C ABI, the
design tables equal the oracle's bit for bit, and device entry points refuse to run."""
import ctypes
import os
import re

import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gvtm_[a-z_0-9]+|GAMA_TTS_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = g.load_library()
    names = _declared_symbols("gama_vtm.h")
    assert "gvtm_synthesize_batch_device" in names and len(names) >= 12
    for name in names:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("delay,rate,layout", [(1, 44100.0, 0), (1, 48000.0, 0), (2, 44100.0, 0), (3, 44100.0, 0), (1, 16000.0, 0),
                                               (1, 44100.0, 1), (1, 22050.0, 1)])
def test_design_matches_oracle_bit_for_bit(delay, rate, layout):
    d = g.read_config_file(oracle.VOICE_MALE)
    plan = g.Plan(g.config_from_dict(d, rate, delay, tube_layout=layout), 250.0, capi.DEVICE_NONE)
    ocfg = oracle.male_config(rate, delay, layout)
    od = oracle.derive(ocfg)
    i = plan.info
    assert (i.internal_sample_rate, i.control_steps, i.fir_taps) == (od.sample_rate, od.control_steps, od.fir_taps)
    assert (i.time_register_increment, i.phase_increment, i.pad_size, i.upsampling) == \
        (od.time_register_increment, od.phase_increment, od.pad_size, od.upsampling)
    fir = np.empty(401)
    n = oracle.lib().vtmo_fir_coefficients(fir.ctypes.data)
    h, dh, wt = np.empty(3328), np.empty(3328), np.empty(512)
    oracle.lib().vtmo_src_filter(h.ctypes.data, dh.ctypes.data)
    oracle.lib().vtmo_wavetable(ctypes.byref(ocfg), od.sample_rate, wt.ctypes.data)
    assert np.array_equal(plan.table(capi.TABLE_FIR), fir[:n])
    assert np.array_equal(plan.table(capi.TABLE_SRC_H), h)
    assert np.array_equal(plan.table(capi.TABLE_SRC_DH), dh)
    assert np.array_equal(plan.table(capi.TABLE_WAVETABLE), wt)
    for frames in (0, 1, 2, 3, 13, 100, 500, 7500):
        assert plan.output_count(frames) == oracle.output_count(ocfg, frames)


@pytest.mark.parametrize("delay,rate,layout", [(1, 44100.0, 0), (2, 48000.0, 0), (3, 44100.0, 0), (1, 16000.0, 0), (1, 44100.0, 1)])
def test_float_design_matches_float_oracle_bit_for_bit(delay, rate, layout):
    """GVTM_PRECISION_F32 designs every constant and table in float, as the reference's TFloat = float models do."""
    d = g.read_config_file(oracle.VOICE_MALE)
    plan = g.Plan(g.config_from_dict(d, rate, delay, capi.PRECISION_F32, layout), 250.0, capi.DEVICE_NONE)
    ocfg = oracle.male_config(rate, delay, layout, float_model=1)
    od = oracle.derive(ocfg)
    i = plan.info
    assert i.precision == capi.PRECISION_F32
    assert (i.internal_sample_rate, i.control_steps, i.fir_taps) == (od.sample_rate, od.control_steps, od.fir_taps)
    assert i.fir_taps == 47  # 49 taps in double, 47 in float
    assert (i.time_register_increment, i.phase_increment, i.pad_size, i.upsampling) == \
        (od.time_register_increment, od.phase_increment, od.pad_size, od.upsampling)
    fir = np.empty(401, dtype=np.float32)
    n = oracle.lib().vtmo_fir_coefficients_f32(fir.ctypes.data)
    h, dh, wt = np.empty(3328, dtype=np.float32), np.empty(3328, dtype=np.float32), np.empty(512, dtype=np.float32)
    oracle.lib().vtmo_src_filter_f32(h.ctypes.data, dh.ctypes.data)
    oracle.lib().vtmo_wavetable_f32(ctypes.byref(ocfg), od.sample_rate, wt.ctypes.data)
    assert np.array_equal(plan.table(capi.TABLE_FIR), fir[:n].astype(np.float64))
    assert np.array_equal(plan.table(capi.TABLE_SRC_H), h.astype(np.float64))
    assert np.array_equal(plan.table(capi.TABLE_SRC_DH), dh.astype(np.float64))
    assert np.array_equal(plan.table(capi.TABLE_WAVETABLE), wt.astype(np.float64))
    for frames in (0, 1, 3, 100, 500, 7500):
        assert plan.output_count(frames) == oracle.output_count(ocfg, frames)


def test_output_count_sweep_against_oracle_ring_logic():
    """gvtm_output_count vs the oracle's literal ring-buffer walk, incl. the down-sampling flush overrun
    (SampleRateConverter.h:298-308 with :462-471): about one ring (1024 input samples) of extra output at ~0.4 % of the
    lengths.  gvtm_output_capacity bounds every shorter utterance's count."""
    d = g.read_config_file(oracle.VOICE_MALE)
    overruns = 0
    for delay, rate, crate, span in ((3, 44100.0, 250.0, 260), (1, 16000.0, 250.0, 260), (1, 44100.0, 250.0, 260),
                                     (3, 16000.0, 1000.0, 700), (3, 11025.0, 1000.0, 300)):
        plan = g.Plan(g.config_from_dict(d, rate, delay), crate, capi.DEVICE_NONE)
        ocfg = oracle.male_config(rate, delay)
        i = plan.info
        longest = 0
        for frames in range(0, span):
            want = oracle.output_count(ocfg, frames, crate)
            fills = frames * i.control_steps + 2 * i.pad_size
            closed = -((-(fills << 16)) // i.time_register_increment)
            got = plan.output_count(frames)
            assert got == want, (delay, rate, frames)
            if want != closed:
                overruns += 1
                assert not i.upsampling
                # the literal walk converts one ring (1024 input samples) of stale data more
                assert want > closed and abs((want - closed) * i.time_register_increment / 65536.0 - 1024) < 8
            longest = max(longest, got)
            assert longest <= plan.output_capacity(frames)
        if i.upsampling:
            assert plan.output_capacity(span) == plan.output_count(span)
    assert overruns >= 3


def test_product_library_exports_no_test_hooks():
    """The gvtm_debug_* hooks and forced shapes live in libgama_vtm_diag.so only (built with -DGVTM_DIAGNOSTICS)."""
    import subprocess
    names = subprocess.run(["nm", "-D", "--defined-only", capi.library_path()], capture_output=True, text=True, check=True).stdout
    assert "gvtm_synthesize_batch_device" in names and "gvtm_debug" not in names
    diag = subprocess.run(["nm", "-D", "--defined-only", capi.library_path(True)], capture_output=True, text=True, check=True).stdout
    assert "gvtm_debug_set_rows" in diag and "gvtm_synthesize_batch_device" in diag
    blob = open(capi.library_path(), "rb").read()
    assert b"GVTM_ROWS" not in blob and b"GVTM_KERNEL" not in blob  # no environment hooks in the product


def test_invalid_configurations_are_rejected():
    d = g.read_config_file(oracle.VOICE_MALE)
    for key, value in (("output_rate", 0.0), ("nasal_radius_2", 0.0), ("glottal_pulse_tp", 0.0), ("mix_offset", 0.0)):
        bad = dict(d)
        bad[key] = str(value)
        with pytest.raises(g.GvtmError) as ei:
            g.Plan(g.config_from_dict(bad), 250.0, capi.DEVICE_NONE)
        assert ei.value.status == 1
    with pytest.raises(g.GvtmError):
        g.Plan(g.config_from_dict(d, section_delay=9), 250.0, capi.DEVICE_NONE)
    with pytest.raises(g.GvtmError):
        g.Plan(g.config_from_dict(d), 0.0, capi.DEVICE_NONE)


def test_no_cpu_synthesis_path():
    d = g.read_config_file(oracle.VOICE_MALE)
    plan = g.Plan(g.config_from_dict(d), 250.0, capi.DEVICE_NONE)
    with pytest.raises(g.GvtmError) as ei:
        plan.synthesize_host(np.zeros((1, 2, 16), np.float32))
    assert ei.value.status == 2  # GVTM_ERR_NO_DEVICE
    if g.device_count() == 0:
        with pytest.raises(g.GvtmError) as ei:
            g.Plan(g.config_from_dict(d), 250.0, 0)
        assert ei.value.status == 2


def test_plugin_library_exports_the_gamatts_symbols():
    libdir = os.path.dirname(g.library_path())
    g.load_library()  # libgama_vtm.so first: the plugin links against it
    plugin = ctypes.CDLL(os.path.join(libdir, "libgama_vtm_plugin.so"))
    for name in _declared_symbols("gama_vtm_plugin.h"):
        assert hasattr(plugin, name), name
    plugin.GAMA_TTS_construct_vocal_tract_model.restype = ctypes.c_void_p
    plugin.GAMA_TTS_construct_vocal_tract_model.argtypes = [ctypes.c_void_p, ctypes.c_int]
    # a failed construct returns NULL -> the host raises "Could not construct the vocal tract model."
    # (VocalTractModelPlugin.cpp:88-90); both caller contracts (batch and interactive) with no configuration
    assert plugin.GAMA_TTS_construct_vocal_tract_model(None, 0) is None
    assert plugin.GAMA_TTS_construct_vocal_tract_model(None, 1) is None
    assert os.access(os.path.join(libdir, "gama_vtm_batch"), os.X_OK)


def test_short_math_accuracy():
    """csrc/vtm_math.hpp (the kernels' 2^x, 10^x, cos, tan for the parameter conversions) against
    80-bit libm over the argument ranges the model produces: < 4e-16 relative."""
    lib = g.load_library(diagnostics=True)
    rng = np.random.default_rng(7)
    cases = [
        (0, rng.uniform(-30.0, 10.0, 200000), lambda v: np.exp2(v)),
        (1, rng.uniform(-3.2, 0.2, 200000), lambda v: np.power(np.longdouble(10.0), v)),
        (2, rng.uniform(0.0, 3.1, 200000), lambda v: np.cos(v)),
        (3, rng.uniform(1e-6, 1.49, 200000), lambda v: np.tan(v)),
        # out-of-range arguments take the library path
        (0, np.array([-2000.0, 1500.0, 0.0]), lambda v: np.exp2(v)),
        (2, np.array([-1.0, 7.5, 100.0]), lambda v: np.cos(v)),
        (3, np.array([1.55, -0.3, 4.0]), lambda v: np.tan(v)),
    ]
    for kind, x, ref_fn in cases:
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        assert lib.gvtm_debug_short_math(kind, x.ctypes.data, x.size, out.ctypes.data) == 0
        ref = ref_fn(x.astype(np.longdouble))
        with np.errstate(over="ignore"):
            ref64 = ref.astype(np.float64)  # 2^1500 overflows to inf on purpose
        ok = np.isfinite(ref64) & (np.abs(ref) > 1e-3)
        rel = np.abs((out[ok].astype(np.longdouble) - ref[ok]) / ref[ok])
        assert float(rel.max()) < 4e-16, (kind, float(rel.max()))
        small = ~ok & np.isfinite(ref64)
        if small.any():
            assert float(np.abs(out[small].astype(np.longdouble) - ref[small]).max()) < 4e-16


def test_powf_restatement_is_bit_identical_to_libm():
    """The all-float path reproduces glibc's powf(2, x) / powf(10, y) (csrc/vtm_math.hpp): every float in
    the ranges the model can produce (pitch -> |x| < 8; dB -> -3 <= y < 0), plus wider samples."""
    lib = g.load_library(diagnostics=True)
    ol = oracle.lib()
    ol.vtmo_libm_powf.argtypes = [ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]

    def all_floats(lo, hi):
        a = np.array([lo, hi], dtype=np.float32).view(np.uint32)
        return np.arange(a[0], a[1] + 1, dtype=np.uint32).view(np.float32)

    def check(kind, base, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        ref = np.empty_like(x)
        ol.vtmo_libm_powf(base, x.ctypes.data, x.size, ref.ctypes.data)
        xd = x.astype(np.float64)
        out = np.empty_like(xd)
        assert lib.gvtm_debug_short_math(kind, xd.ctypes.data, xd.size, out.ctypes.data) == 0
        assert np.array_equal(out.astype(np.float32).view(np.uint32), ref.view(np.uint32)), (kind, base)

    # 10^y: every float y in [-3.25, -2^-10] (dB 0..60 -> y in [-3, 0))
    check(5, 10.0, -all_floats(2.0 ** -10, 3.25))
    # 2^x: every float in +-[2^-6, 8] (pitch -99..+93 semitones)
    pos = all_floats(2.0 ** -6, 8.0)
    check(4, 2.0, pos)
    check(4, 2.0, -pos)
    rng = np.random.default_rng(11)
    check(4, 2.0, rng.uniform(-90.0, 90.0, 400000))
    check(5, 10.0, rng.uniform(-25.0, 25.0, 400000))
    check(4, 2.0, np.array([0.0, -0.0, 1e-30, -1e-30, 1.0, -1.0, 0.25], dtype=np.float32))


def test_cosf_tanf_restatement_is_bit_identical_to_libm():
    """glibc's cosf / tanf as the all-float band-pass design needs them (csrc/vtm_math.hpp): every float of
    cos on [2^-13, 3.2] and tan on [2^-14, 1.38] (bandwidth up to 0.44 of the internal rate)."""
    lib = g.load_library(diagnostics=True)
    ol = oracle.lib()
    ol.vtmo_libm_tanf_cosf.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]

    def check(kind, which, lo, hi):
        a = np.array([lo, hi], dtype=np.float32).view(np.uint32)
        step = 1 << 22
        for start in range(int(a[0]), int(a[1]) + 1, step):
            x = np.arange(start, min(start + step, int(a[1]) + 1), dtype=np.uint32).view(np.float32)
            ref = np.empty_like(x)
            ol.vtmo_libm_tanf_cosf(which, x.ctypes.data, x.size, ref.ctypes.data)
            xd = x.astype(np.float64)
            out = np.empty_like(xd)
            assert lib.gvtm_debug_short_math(kind, xd.ctypes.data, xd.size, out.ctypes.data) == 0
            assert np.array_equal(out.astype(np.float32).view(np.uint32), ref.view(np.uint32)), (kind, start)

    check(6, 1, 2.0 ** -13, 3.2)
    check(7, 0, 2.0 ** -14, 1.38)
    for kind, which in ((6, 1), (7, 0)):  # zero, and arguments outside the restated ranges (library path)
        x = np.array([0.0, 1e-30, 5.0, 100.0, 1000.0, -0.5], dtype=np.float32)
        ref = np.empty_like(x)
        ol.vtmo_libm_tanf_cosf(which, x.ctypes.data, x.size, ref.ctypes.data)
        xd = x.astype(np.float64)
        out = np.empty_like(xd)
        assert lib.gvtm_debug_short_math(kind, xd.ctypes.data, xd.size, out.ctypes.data) == 0
        assert np.allclose(out, ref, rtol=3e-7, atol=0)


def test_header_is_plain_c_and_the_c_example_builds(tmp_path):
    """include/gama_vtm.h compiles as strict C99, and examples/synthesize_batch.c (no C++, no HIP headers) links against
    libgama_vtm.so; without a device it runs the design-only part and stops at the synthesis call."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    for h in ("gama_vtm.h", "gama_vtm_plugin.h"):
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(inc, h)], check=True)
    libdir = os.path.dirname(g.library_path())
    exe = str(tmp_path / "synthesize_batch")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-O1", "-I" + inc, os.path.join(ROOT, "examples", "synthesize_batch.c"),
                    "-L" + libdir, "-lgama_vtm", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe, "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "internal rate 20034 Hz, 80 steps per frame, 88108 samples" in r.stdout
    if g.device_count() == 0:
        assert "no HIP device" in r.stdout
    else:
        assert "utterance 1: 88108 samples" in r.stdout


def test_ragged_example_sizes_rows_with_output_capacity(tmp_path):
    """examples/synthesize_ragged.c = the INTEGRATION.md snippet compiled: a ragged batch on a down-sampling plan (reference
    model 3 at 44.1 kHz) whose 2334-frame utterance hits the converter's flush overrun and is LONGER than its 2335-frame
    neighbour; the row stride must be gvtm_output_capacity(plan, max_frames).  Design-only without a device (it stops at the
    synthesis call), the whole batch with one; both lengths are pinned by the reference-made vectors of
    tests/golden/vtm_overrun_golden.npz (tests/test_gpu_overrun.py)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(g.library_path())
    exe = str(tmp_path / "synthesize_ragged")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + inc, os.path.join(ROOT, "examples", "synthesize_ragged.c"),
                    "-L" + libdir, "-lgama_vtm", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "utterance 0: 2334 frames -> 411798 samples" in r.stdout
    assert "utterance 1: 2335 frames -> 411223 samples" in r.stdout
    # (the capacity is the maximum over EVERY length up to max_frames: another overrun length below 2334 yields 411974)
    assert "gvtm_output_count(max_frames) = 411223, gvtm_output_capacity(max_frames) = 411974" in r.stdout
    if g.device_count() == 0:
        assert "no HIP device" in r.stdout
    else:
        assert "utterance 0: 411798 samples in a row of 411974" in r.stdout


def test_noise_table_is_the_reference_sequence():
    """The per-plan table of noise samples (csrc/vtm_design.cpp: design_noise_table) against a numpy restatement of
    NoiseSource::getSample + NoiseFilter::filter (vtm/NoiseSource.h:40-44, vtm/NoiseFilter.h:63-68): seed 0.7892347,
    seed = frac(seed * 377) with the product rounded first, sample seed - 0.5, y = x + x1 in TFloat."""
    lib = g.load_library(diagnostics=True)
    lib.gvtm_debug_noise_table.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    n = 100000
    seed = np.float64(0.7892347)
    white = np.empty(n, np.float64)
    for i in range(n):
        product = np.float64(seed * np.float64(377.0))
        seed = np.float64(product - np.floor(product))
        white[i] = seed - np.float64(0.5)
    for as_float, dt in ((0, np.float64), (1, np.float32)):
        w = white.astype(dt)
        want = w.copy()
        want[1:] = (w[1:] + w[:-1]).astype(dt)
        got = np.empty(n, dt)
        assert lib.gvtm_debug_noise_table(n, as_float, got.ctypes.data) == 0
        assert np.array_equal(got.view(np.uint32 if as_float else np.uint64), want.view(np.uint32 if as_float else np.uint64))


@pytest.mark.parametrize("precision", [capi.PRECISION_F64, capi.PRECISION_MIXED, capi.PRECISION_F32])
def test_every_plan_has_a_workgroup_shape_that_fits_the_lds(precision):
    """160 KB of LDS per workgroup: one utterance per workgroup must always fit (the launch falls back to it), and the
    shapes the headline workloads use (four utterances per workgroup, up-sampling plans) must keep fitting."""
    lib = g.load_library(diagnostics=True)
    lib.gvtm_debug_lds_bytes.restype = ctypes.c_size_t
    lib.gvtm_debug_lds_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    cfgd = g.read_config_file(oracle.VOICE_MALE)
    for delay in (1, 2, 3, 4):
        for rate in (16000.0, 22050.0, 44100.0, 48000.0):
            plan = g.Plan(g.config_from_dict(cfgd, rate, delay, precision), 250.0, -1, diagnostics=True)
            assert lib.gvtm_debug_lds_bytes(plan._h, 1) <= 160 * 1024, (delay, rate)
            upsampling = plan.info.upsampling
            if upsampling and delay <= 2:
                assert lib.gvtm_debug_lds_bytes(plan._h, 4) <= 160 * 1024, (delay, rate)
