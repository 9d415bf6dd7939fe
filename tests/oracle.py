"""ctypes binding of the CPU oracle (oracle/vtm_oracle.c) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libvtm_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
VOICE_MALE = os.path.join(GOLDEN_DIR, "voice_male.txt")


class OracleConfig(ctypes.Structure):
    _fields_ = [
        ("output_rate", ctypes.c_double),
        ("waveform", ctypes.c_int),
        ("glottal_pulse_tp", ctypes.c_double),
        ("glottal_pulse_tn_min", ctypes.c_double),
        ("glottal_pulse_tn_max", ctypes.c_double),
        ("breathiness", ctypes.c_double),
        ("vocal_tract_length_offset", ctypes.c_double),
        ("vocal_tract_length", ctypes.c_double),
        ("temperature", ctypes.c_double),
        ("loss_factor", ctypes.c_double),
        ("mouth_coefficient", ctypes.c_double),
        ("nose_coefficient", ctypes.c_double),
        ("throat_cutoff", ctypes.c_double),
        ("throat_volume", ctypes.c_double),
        ("noise_modulation", ctypes.c_int),
        ("mix_offset", ctypes.c_double),
        ("global_radius_coef", ctypes.c_double),
        ("global_nasal_radius_coef", ctypes.c_double),
        ("aperture_radius", ctypes.c_double),
        ("nasal_radius", ctypes.c_double * 5),
        ("radius_coef", ctypes.c_double * 8),
        ("section_delay", ctypes.c_int),
        ("layout", ctypes.c_int),
        ("float_model", ctypes.c_int),
    ]


class OracleDerived(ctypes.Structure):
    _fields_ = [
        ("sample_rate", ctypes.c_int),
        ("control_steps", ctypes.c_uint),
        ("fir_taps", ctypes.c_int),
        ("table_div1", ctypes.c_uint),
        ("table_div2", ctypes.c_uint),
        ("tn_delta", ctypes.c_double),
        ("time_register_increment", ctypes.c_uint),
        ("phase_increment", ctypes.c_uint),
        ("pad_size", ctypes.c_int),
        ("upsampling", ctypes.c_int),
    ]


def read_config_file(path):
    """key = value file in the reference's ConfigurationData format (ConfigurationData.cpp:67-118)."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line.startswith("#"):
                continue
            k, v = line.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def config_from_dict(d, output_rate=None, section_delay=1, layout=0, float_model=0):
    c = OracleConfig()
    c.output_rate = float(d["output_rate"]) if output_rate is None else float(output_rate)
    c.waveform = int(float(d["waveform"]))
    for k in ("glottal_pulse_tp", "glottal_pulse_tn_min", "glottal_pulse_tn_max", "breathiness",
              "vocal_tract_length_offset", "vocal_tract_length", "temperature", "loss_factor",
              "mouth_coefficient", "nose_coefficient", "throat_cutoff", "throat_volume", "mix_offset",
              "global_radius_coef", "global_nasal_radius_coef", "aperture_radius"):
        setattr(c, k, float(d[k]))
    c.noise_modulation = int(float(d["noise_modulation"]))
    for i in range(5):
        c.nasal_radius[i] = float(d["nasal_radius_%d" % (i + 1)])
    for i in range(8):
        c.radius_coef[i] = float(d["radius_%d_coef" % (i + 1)])
    c.section_delay = section_delay
    c.layout = layout
    c.float_model = int(float_model)
    return c


def male_config(output_rate=44100.0, section_delay=1, layout=0, float_model=0, **overrides):
    d = read_config_file(VOICE_MALE)
    d.update({k: str(v) for k, v in overrides.items()})
    return config_from_dict(d, output_rate, section_delay, layout, float_model)


VOICE5_MALE = os.path.join(GOLDEN_DIR, "voice5_male.txt")


class Oracle5Config(ctypes.Structure):
    """vtmo5_config (oracle/vtm_oracle.h): VocalTractModel5's configuration keys as numbers."""
    _fields_ = [
        ("output_rate", ctypes.c_double),
        ("waveform", ctypes.c_int), ("noise_modulation", ctypes.c_int), ("bypass", ctypes.c_int),
        ("constant_radius_mouth_impedance", ctypes.c_int),
        ("glottal_pulse_tp", ctypes.c_double), ("glottal_pulse_tn_min", ctypes.c_double),
        ("glottal_pulse_tn_max", ctypes.c_double), ("breathiness", ctypes.c_double),
        ("vocal_tract_length_offset", ctypes.c_double), ("vocal_tract_length", ctypes.c_double),
        ("temperature", ctypes.c_double), ("loss_factor", ctypes.c_double), ("mix_offset", ctypes.c_double),
        ("global_radius_coef", ctypes.c_double), ("global_nasal_radius_coef", ctypes.c_double),
        ("nasal_radius", ctypes.c_double * 6),
        ("radius_coef", ctypes.c_double * 8),
        ("glottal_noise_cutoff", ctypes.c_double), ("frication_noise_cutoff", ctypes.c_double),
        ("frication_factor", ctypes.c_double), ("min_glottal_loss", ctypes.c_double),
        ("max_glottal_loss", ctypes.c_double), ("glottal_lowpass_cutoff", ctypes.c_double),
        ("mouth_impedance_radius", ctypes.c_double),
        ("float_model", ctypes.c_int),
    ]


def _flag(v):
    return 1 if str(v).strip().lower() in ("1", "true") else 0


def config5_from_dict(d, output_rate=None, float_model=0):
    c = Oracle5Config()
    c.output_rate = float(d["output_rate"]) if output_rate is None else float(output_rate)
    c.waveform = int(float(d["waveform"]))
    c.noise_modulation = int(float(d["noise_modulation"]))
    c.bypass = int(float(d["bypass"]))
    c.constant_radius_mouth_impedance = _flag(d["constant_radius_mouth_impedance"])
    for k in ("glottal_pulse_tp", "glottal_pulse_tn_min", "glottal_pulse_tn_max", "breathiness",
              "vocal_tract_length_offset", "vocal_tract_length", "temperature", "loss_factor", "mix_offset",
              "global_radius_coef", "global_nasal_radius_coef", "glottal_noise_cutoff", "frication_noise_cutoff",
              "frication_factor", "min_glottal_loss", "max_glottal_loss", "glottal_lowpass_cutoff",
              "mouth_impedance_radius"):
        setattr(c, k, float(d[k]))
    for i in range(6):
        c.nasal_radius[i] = float(d["nasal_radius_%d" % (i + 2)])
    for i in range(8):
        c.radius_coef[i] = float(d["radius_%d_coef" % (i + 1)])
    c.float_model = int(float_model)
    return c


def male5_config(output_rate=48000.0, float_model=0, **overrides):
    d = read_config_file(VOICE5_MALE)
    d.update({k: str(v) for k, v in overrides.items()})
    return config5_from_dict(d, output_rate, float_model)


def synthesize5(cfg, params, control_rate=250.0):
    """VocalTractModel5 restatement (oracle only): float32 [F][16] -> (float32 [N], internal rate in Hz)."""
    L = lib()
    L.vtmo5_synthesize.argtypes = [ctypes.POINTER(Oracle5Config), ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t,
                                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.vtmo5_synthesize.restype = ctypes.c_size_t
    params = np.ascontiguousarray(params, dtype=np.float32)
    assert params.ndim == 2 and params.shape[1] == 16
    rate = ctypes.c_int(0)
    n = L.vtmo5_synthesize(ctypes.byref(cfg), control_rate, params.ctypes.data, params.shape[0], None, 0, ctypes.byref(rate))
    out = np.empty(n, dtype=np.float32)
    got = L.vtmo5_synthesize(ctypes.byref(cfg), control_rate, params.ctypes.data, params.shape[0], out.ctypes.data, n, None)
    assert got == n, (got, n)
    return out, rate.value / 1000.0


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("vtm_oracle.c", "vtm_oracle_body.inc", "vtm_oracle_f64.c",
                                                       "vtm_oracle_f32.c", "vtm_oracle.h", "vtm_tracks_oracle.c", "vtm_tracks_oracle.h")]
        if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
            build()
        L = ctypes.CDLL(LIB_PATH)
        P = ctypes.POINTER
        L.vtmo_derive.argtypes = [P(OracleConfig), ctypes.c_double, P(OracleDerived)]
        L.vtmo_derive.restype = ctypes.c_int
        L.vtmo_fir_coefficients.argtypes = [ctypes.c_void_p]
        L.vtmo_fir_coefficients.restype = ctypes.c_int
        L.vtmo_src_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.vtmo_wavetable.argtypes = [P(OracleConfig), ctypes.c_int, ctypes.c_void_p]
        L.vtmo_fir_coefficients_f32.argtypes = [ctypes.c_void_p]
        L.vtmo_fir_coefficients_f32.restype = ctypes.c_int
        L.vtmo_src_filter_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.vtmo_wavetable_f32.argtypes = [P(OracleConfig), ctypes.c_int, ctypes.c_void_p]
        L.vtmo_noise_sequence.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.vtmo_output_count.argtypes = [P(OracleConfig), ctypes.c_double, ctypes.c_size_t]
        L.vtmo_output_count.restype = ctypes.c_size_t
        L.vtmo_synthesize.argtypes = [P(OracleConfig), ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.vtmo_synthesize.restype = ctypes.c_size_t
        L.vtmo_synthesize_batch.argtypes = [P(OracleConfig), ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.vtmo_synthesize_batch.restype = ctypes.c_size_t
        L.vtmo_synthesize_debug.argtypes = [P(OracleConfig), ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.vtmo_synthesize_debug.restype = ctypes.c_size_t
        L.vtmo_output_scale.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.vtmo_output_scale.restype = ctypes.c_float
        _lib = L
    return _lib


class TrackConfig(ctypes.Structure):
    _fields_ = [("control_period", ctypes.c_int), ("macro_intonation", ctypes.c_int), ("micro_intonation", ctypes.c_int),
                ("intonation_drift", ctypes.c_int), ("smooth_intonation", ctypes.c_int),
                ("initial_pitch", ctypes.c_double), ("mean_pitch", ctypes.c_double),
                ("drift_deviation", ctypes.c_double), ("drift_sample_rate", ctypes.c_double),
                ("drift_lowpass_cutoff", ctypes.c_double)]


def track_config(v):
    """v: the 10 numbers of a tracks fixture (control_period, macro, micro, drift, smooth, initial_pitch, mean_pitch,
    drift deviation / sample rate / cutoff)."""
    c = TrackConfig()
    c.control_period, c.macro_intonation, c.micro_intonation, c.intonation_drift, c.smooth_intonation = (int(x) for x in v[:5])
    c.initial_pitch, c.mean_pitch, c.drift_deviation, c.drift_sample_rate, c.drift_lowpass_cutoff = (float(x) for x in v[5:10])
    return c


FRESH_DRIFT = (0.7892347, 0.0, 0.0, 0.0, 0.0)


def tracks_generate(cfg, events, drift=FRESH_DRIFT):
    """events float64 [E][38] -> (frames float32 [F][16], drift state after)   (oracle/vtm_tracks_oracle.c)"""
    L = lib()
    L.vtmo_tracks_generate.argtypes = [ctypes.POINTER(TrackConfig), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_size_t]
    L.vtmo_tracks_generate.restype = ctypes.c_size_t
    events = np.ascontiguousarray(events, dtype=np.float64)
    st = np.array(drift, dtype=np.float64)
    n = L.vtmo_tracks_generate(ctypes.byref(cfg), events.ctypes.data, events.shape[0], st.copy().ctypes.data, None, 0)
    frames = np.zeros((n, 16), dtype=np.float32)
    got = L.vtmo_tracks_generate(ctypes.byref(cfg), events.ctypes.data, events.shape[0], st.ctypes.data, frames.ctypes.data, n)
    assert got == n
    return frames, tuple(st)


def derive(cfg, control_rate=250.0):
    d = OracleDerived()
    rc = lib().vtmo_derive(ctypes.byref(cfg), control_rate, ctypes.byref(d))
    if rc != 0:
        raise ValueError("bad oracle config")
    return d


def output_count(cfg, frames, control_rate=250.0):
    return lib().vtmo_output_count(ctypes.byref(cfg), control_rate, frames)


def synthesize(cfg, params, control_rate=250.0, want_internal=False):
    """params: float32 [F][16] -> float32 [N] (and the internal-rate float64 signal)."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    assert params.ndim == 2 and params.shape[1] == 16
    frames = params.shape[0]
    n = output_count(cfg, frames, control_rate)
    out = np.empty(n, dtype=np.float32)
    internal = None
    ip = None
    if want_internal:
        internal = np.empty(frames * derive(cfg, control_rate).control_steps, dtype=np.float64)
        ip = internal.ctypes.data
    got = lib().vtmo_synthesize(ctypes.byref(cfg), control_rate, params.ctypes.data, frames,
                                out.ctypes.data, n, ip)
    assert got == n, (got, n)
    return (out, internal) if want_internal else out


def synthesize_debug(cfg, params, control_rate=250.0):
    """-> (audio, taps float64 [steps][8]); tap layout in oracle/vtm_oracle.h."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    frames = params.shape[0]
    n = output_count(cfg, frames, control_rate)
    out = np.empty(n, dtype=np.float32)
    taps = np.zeros((frames * derive(cfg, control_rate).control_steps, 8), dtype=np.float64)
    lib().vtmo_synthesize_debug(ctypes.byref(cfg), control_rate, params.ctypes.data, frames, out.ctypes.data, n,
                                taps.ctypes.data)
    return out, taps


def synthesize_batch(cfg, params, control_rate=250.0):
    """params: float32 [B][F][16] -> float32 [B][N]."""
    params = np.ascontiguousarray(params, dtype=np.float32)
    assert params.ndim == 3 and params.shape[2] == 16
    b, frames = params.shape[:2]
    n = output_count(cfg, frames, control_rate)
    out = np.empty((b, n), dtype=np.float32)
    got = lib().vtmo_synthesize_batch(ctypes.byref(cfg), control_rate, params.ctypes.data, b, frames,
                                      out.ctypes.data, n)
    assert got == n, (got, n)
    return out


def output_scale(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().vtmo_output_scale(x.ctypes.data, x.size))


# --- the real reference, compiled here from /root/reference (oracle/_ref/, see oracle/Makefile) ---

def ref_binary(kind="gold"):
    name = {"gold": "ref_vtm", "o3": "ref_vtm_o3", "v3": "ref_vtm_v3"}[kind]
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) else None


def ref_synthesize(params, model="0", output_rate=44100, control_rate=250, config=VOICE_MALE, kind="gold",
                   repeat=1, tmpdir=None, poll=0):
    """Run the compiled reference on float32 frames; returns (audio float32, info dict).
    poll > 0: the interactive caller contract (model constructed with interactive = true, outputBuffer() polled and
    drained the way the editor's JACK callback does, `poll` samples per callback; oracle/ref_driver.cpp)."""
    import tempfile
    exe = ref_binary(kind)
    if exe is None:
        raise FileNotFoundError("oracle/_ref not built")
    params = np.ascontiguousarray(params, dtype=np.float32)
    with tempfile.TemporaryDirectory(dir=tmpdir) as td:
        pin = os.path.join(td, "p.f32")
        pout = os.path.join(td, "o.f32")
        params.tofile(pin)
        r = subprocess.run([exe, config, str(model), repr(float(output_rate)), repr(float(control_rate)), pin,
                            str(params.shape[0]), pout, str(repeat)] + (["poll=%d" % poll] if poll else []),
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("ref_vtm failed: " + r.stderr)
        info = dict(kv.split("=") for kv in r.stdout.split())
        return np.fromfile(pout, dtype=np.float32), info


# ---- many oracle references at once ------------------------------------------------------------------------------
# The C restatement keeps no global state and ctypes releases the GIL for the duration of a call, so a THREAD pool gives
# one core per utterance without any child process: nothing is forked from (or inherited by) a test process that has
# already initialised HIP.

def _job_male(job):
    track, rate, delay, layout, float_model = job
    return synthesize(male_config(rate, delay, layout, float_model=float_model), track)


def _job_male5(job):
    track, rate = job
    return synthesize5(male5_config(rate), track)[0]


def synthesize_many(jobs, workers=8, model5=False):
    """jobs: [(track[F][16], rate, delay, layout, float_model)] (model5: [(track, rate)]) -> list of oracle outputs."""
    from concurrent.futures import ThreadPoolExecutor

    jobs = list(jobs)
    fn = _job_male5 if model5 else _job_male
    lib()  # (loaded once, before the threads start)
    if workers <= 1 or len(jobs) < 2:
        return [fn(j) for j in jobs]
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, jobs))
