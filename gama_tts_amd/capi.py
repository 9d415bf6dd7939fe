"""ctypes binding of include/gama_vtm.h.  No computation happens here."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libgama_vtm.so"
N_PARAM = 16
DEVICE_NONE = -1
PRECISION_F64 = 0
PRECISION_MIXED = 1
PRECISION_F32 = 2  # everything in float, like the reference's TFloat = float models (model 1)
TUBE_10_6 = 0
TUBE_30_18 = 1
TABLE_FIR, TABLE_SRC_H, TABLE_SRC_DH, TABLE_WAVETABLE = 0, 1, 2, 3
_STATUS_UNSUPPORTED = 4


class GvtmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gvtm status %d: %s" % (status, message))
        self.status = status


class Config(ctypes.Structure):
    _fields_ = [
        ("output_rate", ctypes.c_double),
        ("waveform", ctypes.c_int32),
        ("noise_modulation", ctypes.c_int32),
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
        ("mix_offset", ctypes.c_double),
        ("global_radius_coef", ctypes.c_double),
        ("global_nasal_radius_coef", ctypes.c_double),
        ("aperture_radius", ctypes.c_double),
        ("nasal_radius", ctypes.c_double * 5),
        ("radius_coef", ctypes.c_double * 8),
        ("section_delay", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("tube_layout", ctypes.c_int32),
        ("reserved_", ctypes.c_int32),
    ]


class Info(ctypes.Structure):
    _fields_ = [
        ("internal_sample_rate", ctypes.c_int32),
        ("control_steps", ctypes.c_uint32),
        ("output_rate", ctypes.c_double),
        ("control_rate", ctypes.c_double),
        ("fir_taps", ctypes.c_int32),
        ("time_register_increment", ctypes.c_uint32),
        ("phase_increment", ctypes.c_uint32),
        ("pad_size", ctypes.c_int32),
        ("upsampling", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("section_delay", ctypes.c_int32),
        ("model5", ctypes.c_int32),
        ("reserved_", ctypes.c_int32),
        ("internal_rate_hz", ctypes.c_double),
    ]


class Config5(ctypes.Structure):
    """gvtm5_config: VocalTractModel5's configuration keys (reference model 5)."""
    _fields_ = [
        ("output_rate", ctypes.c_double),
        ("waveform", ctypes.c_int32), ("noise_modulation", ctypes.c_int32), ("bypass", ctypes.c_int32),
        ("constant_radius_mouth_impedance", ctypes.c_int32),
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
        ("precision", ctypes.c_int32), ("reserved_", ctypes.c_int32),
    ]


def library_path(diagnostics=False):
    return os.path.join(_HERE, "lib", "libgama_vtm_diag.so" if diagnostics else _LIB_NAME)


_libs = {}


def load_library(diagnostics=False):
    """Loads libgama_vtm.so from the in-tree build; raises if it has not been built.

    diagnostics=True loads libgama_vtm_diag.so instead: the same kernels behind a C ABI built with -DGVTM_DIAGNOSTICS,
    which adds the gvtm_debug_* hooks (tests and tools only; the product library exports none of them)."""
    if diagnostics in _libs:
        return _libs[diagnostics]
    # GVTM_LIBRARY / GVTM_DIAG_LIBRARY: A/B runs of library variants (tools/ab.py, tools/build_variant.sh)
    path = (os.environ.get("GVTM_DIAG_LIBRARY") or library_path(True)) if diagnostics else (os.environ.get("GVTM_LIBRARY") or library_path())
    # libgama_vtm.so and PyTorch-ROCm both need libamdhip64.so.7 and a process can hold only one
    # copy: whichever loads first serves both.  PyTorch only works with the copy bundled in its
    # wheel, so when torch is installed let it load first (bench.py/tests share device pointers
    # and streams with torch).  C/C++ hosts are unaffected: they use the system ROCm runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C gama_tts_amd/csrc). There is no fallback path." % path)
    L = ctypes.CDLL(path)
    vp, sz, dbl, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int
    L.gvtm_status_string.restype = ctypes.c_char_p
    L.gvtm_status_string.argtypes = [i32]
    L.gvtm_last_error.restype = ctypes.c_char_p
    L.gvtm_device_count.restype = i32
    L.gvtm_plan_create.argtypes = [ctypes.POINTER(Config), dbl, i32, ctypes.POINTER(vp)]
    L.gvtm_plan_create.restype = i32
    L.gvtm_plan_create_model5.argtypes = [ctypes.POINTER(Config5), dbl, i32, ctypes.POINTER(vp)]
    L.gvtm_plan_create_model5.restype = i32
    L.gvtm_plan_destroy.argtypes = [vp]
    L.gvtm_plan_destroy.restype = None
    L.gvtm_plan_info.argtypes = [vp, ctypes.POINTER(Info)]
    L.gvtm_plan_info.restype = i32
    L.gvtm_plan_table.argtypes = [vp, i32, vp, sz]
    L.gvtm_plan_table.restype = i32
    L.gvtm_output_count.argtypes = [vp, sz]
    L.gvtm_output_count.restype = sz
    L.gvtm_output_capacity.argtypes = [vp, sz]
    L.gvtm_output_capacity.restype = sz
    L.gvtm_synthesize_batch_device.argtypes = [vp, vp, vp, sz, sz, vp, sz, vp, vp, vp]
    L.gvtm_synthesize_batch_device.restype = i32
    L.gvtm_synthesize_batch_host.argtypes = [vp, vp, vp, sz, sz, vp, sz, vp, vp]
    L.gvtm_synthesize_batch_host.restype = i32
    L.gvtm_synthesize_batch_host_pcm16.argtypes = [vp, vp, vp, sz, sz, vp, sz, vp, vp, vp]
    L.gvtm_synthesize_batch_host_pcm16.restype = i32
    L.gvtm_host_alloc.argtypes = [sz, ctypes.POINTER(vp)]
    L.gvtm_host_alloc.restype = i32
    L.gvtm_host_free.argtypes = [vp]
    L.gvtm_host_free.restype = None
    L.gvtm_normalize_batch_device.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp, vp]
    L.gvtm_normalize_batch_device.restype = i32
    L.gvtm_tracks_frame_count.argtypes = [ctypes.POINTER(TrackConfig), vp, sz]
    L.gvtm_tracks_frame_count.restype = sz
    L.gvtm_generate_tracks_device.argtypes = [i32, ctypes.POINTER(TrackConfig), vp, vp, sz, sz, vp, vp, vp, vp]
    L.gvtm_generate_tracks_device.restype = i32
    L.gvtm_synthesize_events_device.argtypes = [vp, ctypes.POINTER(TrackConfig), vp, vp, sz, sz, vp, sz, vp, vp, vp, vp, vp]
    L.gvtm_synthesize_events_device.restype = i32
    L.gvtm_generate_tracks_host.argtypes = [i32, ctypes.POINTER(TrackConfig), vp, vp, sz, sz, vp, vp, vp]
    L.gvtm_generate_tracks_host.restype = i32
    L.gvtm_stream_create.argtypes = [vp, sz, ctypes.POINTER(vp)]
    L.gvtm_stream_create.restype = i32
    L.gvtm_stream_destroy.argtypes = [vp]
    L.gvtm_stream_destroy.restype = None
    L.gvtm_stream_reset.argtypes = [vp]
    L.gvtm_stream_reset.restype = i32
    L.gvtm_stream_capacity.argtypes = [vp, sz]
    L.gvtm_stream_capacity.restype = sz
    L.gvtm_stream_push.argtypes = [vp, vp, vp, sz, vp, sz, vp]
    L.gvtm_stream_push.restype = i32
    L.gvtm_stream_finish.argtypes = [vp, vp, sz, vp, vp]
    L.gvtm_stream_finish.restype = i32
    L.gvtm_plan_set_timing.argtypes = [vp, i32]
    L.gvtm_plan_set_timing.restype = i32
    L.gvtm_plan_take_kernel_ms.argtypes = [vp, ctypes.POINTER(i32)]
    L.gvtm_plan_take_kernel_ms.restype = dbl
    if diagnostics:
        L.gvtm_debug_set_rows.argtypes = [vp, i32]
        L.gvtm_debug_set_rows.restype = i32
        L.gvtm_debug_set_taps.argtypes = [vp, vp]
        L.gvtm_debug_set_phase_cycles.argtypes = [vp, vp]
        L.gvtm_debug_dpp_selftest.argtypes = [vp, vp]
        L.gvtm_debug_short_math.argtypes = [i32, vp, sz, vp]
        L.gvtm_debug_device_float_math.argtypes = [vp, i32, vp, sz, vp]
    _libs[diagnostics] = L
    return L


def device_count():
    return int(load_library().gvtm_device_count())


class PinnedArray:
    """A numpy array over page-locked host memory from gvtm_host_alloc (the buffers of the host entries: with them the
    H2D / kernel / D2H pipeline really overlaps).  Keep the object alive as long as `.array` is in use."""

    def __init__(self, shape, dtype):
        self._lib = load_library()
        self.array = None
        self._ptr = ctypes.c_void_p()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        rc = self._lib.gvtm_host_alloc(max(nbytes, 1), ctypes.byref(self._ptr))
        if rc != 0:
            raise GvtmError(rc, self._lib.gvtm_last_error().decode())
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._ptr:
            self.array = None
            self._lib.gvtm_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_config_file(path):
    """`key = value` lines, `#` comments — the reference's ConfigurationData file format
    (gama_tts/src/ConfigurationData.cpp:67-118)."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line.startswith("#"):
                continue
            key, value = line.split("=", 1)
            out[key.strip()] = value.strip()
    return out


def config_from_dict(d, output_rate=None, section_delay=1, precision=PRECISION_F64, tube_layout=TUBE_10_6):
    """Builds a gvtm_config from the merged vtm.txt + variant keys (VocalTractModel0.h:266-305)."""
    c = Config()
    c.output_rate = float(d["output_rate"]) if output_rate is None else float(output_rate)
    c.waveform = int(float(d["waveform"]))
    c.noise_modulation = int(float(d["noise_modulation"]))
    for key in ("glottal_pulse_tp", "glottal_pulse_tn_min", "glottal_pulse_tn_max", "breathiness",
                "vocal_tract_length_offset", "vocal_tract_length", "temperature", "loss_factor",
                "mouth_coefficient", "nose_coefficient", "throat_cutoff", "throat_volume", "mix_offset",
                "global_radius_coef", "global_nasal_radius_coef", "aperture_radius"):
        setattr(c, key, float(d[key]))
    for i in range(5):
        c.nasal_radius[i] = float(d["nasal_radius_%d" % (i + 1)])
    for i in range(8):
        c.radius_coef[i] = float(d["radius_%d_coef" % (i + 1)])
    c.section_delay = int(section_delay)
    c.precision = int(precision)
    c.tube_layout = int(tube_layout)
    return c


def config5_from_dict(d, output_rate=None, precision=PRECISION_F64):
    """Builds a gvtm5_config from the merged vtm.txt + variant keys of a model-5 voice (VocalTractModel5.h:375-421)."""
    c = Config5()
    c.output_rate = float(d["output_rate"]) if output_rate is None else float(output_rate)
    for key in ("waveform", "noise_modulation", "bypass"):
        setattr(c, key, int(float(d[key])))
    c.constant_radius_mouth_impedance = 1 if str(d["constant_radius_mouth_impedance"]).strip().lower() in ("1", "true") else 0
    for key in ("glottal_pulse_tp", "glottal_pulse_tn_min", "glottal_pulse_tn_max", "breathiness",
                "vocal_tract_length_offset", "vocal_tract_length", "temperature", "loss_factor", "mix_offset",
                "global_radius_coef", "global_nasal_radius_coef", "glottal_noise_cutoff", "frication_noise_cutoff",
                "frication_factor", "min_glottal_loss", "max_glottal_loss", "glottal_lowpass_cutoff"):
        setattr(c, key, float(d[key]))
    c.mouth_impedance_radius = float(d.get("mouth_impedance_radius", 0.0))
    for i in range(6):
        c.nasal_radius[i] = float(d["nasal_radius_%d" % (i + 2)])
    for i in range(8):
        c.radius_coef[i] = float(d["radius_%d_coef" % (i + 1)])
    c.precision = int(precision)
    return c


class TrackConfig(ctypes.Structure):
    """gvtm_track_config"""
    _fields_ = [("control_period_ms", ctypes.c_int32), ("macro_intonation", ctypes.c_int32), ("micro_intonation", ctypes.c_int32),
                ("intonation_drift", ctypes.c_int32), ("smooth_intonation", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("initial_pitch", ctypes.c_double), ("mean_pitch", ctypes.c_double), ("drift_deviation", ctypes.c_double),
                ("drift_sample_rate", ctypes.c_double), ("drift_lowpass_cutoff", ctypes.c_double)]


# gvtm_event / gvtm_drift_state as numpy record layouts (296 and 40 bytes)
EVENT_DTYPE = np.dtype([("time_ms", "<i4"), ("has_interp", "<i4"), ("interp", "<f8", 4), ("param", "<f8", 16),
                        ("special", "<f8", 16)])
DRIFT_DTYPE = np.dtype([("seed", "<f8"), ("x1", "<f8"), ("x2", "<f8"), ("y1", "<f8"), ("y2", "<f8")])
FRESH_DRIFT = (0.7892347, 0.0, 0.0, 0.0, 0.0)


def events_from_table(table):
    """float64 [E][38] rows (time, has_interp, a, b, c, d, parameters[16], specialParameters[16]) -> gvtm_event records."""
    table = np.asarray(table, dtype=np.float64)
    ev = np.zeros(table.shape[0], dtype=EVENT_DTYPE)
    ev["time_ms"] = table[:, 0].astype(np.int32)
    ev["has_interp"] = (table[:, 1] != 0).astype(np.int32)
    ev["interp"] = table[:, 2:6]
    ev["param"] = table[:, 6:22]
    ev["special"] = table[:, 22:38]
    return ev


def tracks_frame_count(config, events):
    lib = load_library()
    events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
    n = lib.gvtm_tracks_frame_count(ctypes.byref(config), _ptr(events), events.shape[0])
    if n == ctypes.c_size_t(-1).value:
        raise GvtmError(1, lib.gvtm_last_error().decode())
    return n


def generate_tracks_host(config, event_lists, max_frames, drift=None, device=0):
    """event_lists: list of gvtm_event record arrays -> (params float32 [B][max_frames][16], frame_counts int32 [B], drift out)."""
    lib = load_library()
    batch = len(event_lists)
    offsets = np.zeros(batch + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(e) for e in event_lists])
    events = np.concatenate([np.ascontiguousarray(e, dtype=EVENT_DTYPE) for e in event_lists]) if batch else np.zeros(0, EVENT_DTYPE)
    params = np.zeros((batch, max_frames, 16), dtype=np.float32)
    counts = np.zeros(batch, dtype=np.int32)
    dr = None
    if drift is not None:
        dr = np.zeros(batch, dtype=DRIFT_DTYPE)
        for b, st in enumerate(drift):
            dr[b] = tuple(st)
    rc = lib.gvtm_generate_tracks_host(int(device), ctypes.byref(config), _ptr(events), _ptr(offsets), batch, int(max_frames),
                                       _ptr(params), _ptr(counts), _ptr(dr))
    if rc != 0:
        raise GvtmError(rc, lib.gvtm_last_error().decode() or lib.gvtm_status_string(rc).decode())
    return params, counts, dr


def generate_tracks_device(config, d_events, d_offsets, batch, max_frames, d_params, d_frame_counts=None, d_drift=None,
                           stream=None, device=0):
    lib = load_library()
    rc = lib.gvtm_generate_tracks_device(int(device), ctypes.byref(config), _ptr(d_events), _ptr(d_offsets), int(batch),
                                         int(max_frames), _ptr(d_params), _ptr(d_frame_counts), _ptr(d_drift), _ptr(stream))
    if rc != 0:
        raise GvtmError(rc, lib.gvtm_last_error().decode() or lib.gvtm_status_string(rc).decode())


def _ptr(x):
    """Device/host pointer of a torch tensor, numpy array, int or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, np.ndarray):
        return ctypes.c_void_p(x.ctypes.data)
    return ctypes.c_void_p(x.data_ptr())  # torch.Tensor


class Plan:
    """Owns a gvtm_plan.  device=DEVICE_NONE gives a design-only plan (no GPU needed)."""

    def __init__(self, config, control_rate=250.0, device=0, diagnostics=False, rows=0):
        """diagnostics=True binds the plan to libgama_vtm_diag.so (gvtm_debug_* hooks); rows (diagnostics only) forces
        the utterances per workgroup, i.e. the kernel shape a big batch would get."""
        self._lib = load_library(diagnostics)
        self.diagnostics = bool(diagnostics)
        self._h = ctypes.c_void_p()
        self.config = config
        create = self._lib.gvtm_plan_create_model5 if isinstance(config, Config5) else self._lib.gvtm_plan_create
        rc = create(ctypes.byref(config), float(control_rate), int(device), ctypes.byref(self._h))
        self._check(rc)
        info = Info()
        self._check(self._lib.gvtm_plan_info(self._h, ctypes.byref(info)))
        self.info = info
        if rows:
            if not diagnostics:
                raise ValueError("rows can only be forced on a diagnostics plan")
            self._check(self._lib.gvtm_debug_set_rows(self._h, int(rows)))

    def _check(self, rc):
        if rc != 0:
            raise GvtmError(rc, self._lib.gvtm_last_error().decode() or self._lib.gvtm_status_string(rc).decode())

    def close(self):
        if self._h:
            self._lib.gvtm_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self, which):
        buf = np.empty(4096, dtype=np.float64)
        n = self._lib.gvtm_plan_table(self._h, which, buf.ctypes.data, buf.size)
        if n < 0:
            self._check(-n)
        return buf[:n].copy()

    def output_count(self, frames):
        return int(self._lib.gvtm_output_count(self._h, int(frames)))

    def output_capacity(self, max_frames):
        """Row length that holds every utterance of a ragged batch (> output_count(max_frames) on down-sampling plans)."""
        return int(self._lib.gvtm_output_capacity(self._h, int(max_frames)))

    def set_timing(self, enabled):
        self._check(self._lib.gvtm_plan_set_timing(self._h, int(bool(enabled))))

    def take_kernel_ms(self):
        n = ctypes.c_int(0)
        ms = self._lib.gvtm_plan_take_kernel_ms(self._h, ctypes.byref(n))
        return float(ms), int(n.value)

    def synthesize_device(self, d_params, batch, max_frames, d_audio, audio_stride, d_frame_counts=None,
                          d_out_counts=None, d_maxabs=None, stream=None):
        """All arguments are device pointers (torch CUDA tensors or raw ints)."""
        self._check(self._lib.gvtm_synthesize_batch_device(
            self._h, _ptr(d_params), _ptr(d_frame_counts), int(batch), int(max_frames), _ptr(d_audio),
            int(audio_stride), _ptr(d_out_counts), _ptr(d_maxabs), _ptr(stream)))

    def synthesize_host(self, params, frame_counts=None):
        """params: float32 [B][F][16] numpy -> (audio float32 [B][stride], counts int64 [B], maxabs float32 [B])."""
        params = np.ascontiguousarray(params, dtype=np.float32)
        assert params.ndim == 3 and params.shape[2] == N_PARAM
        batch, frames = params.shape[:2]
        stride = self.output_count(frames) if frame_counts is None else self.output_capacity(frames)
        audio = np.zeros((batch, stride), dtype=np.float32)
        counts = np.zeros(batch, dtype=np.int64)
        maxabs = np.zeros(batch, dtype=np.float32)
        fc = None
        if frame_counts is not None:
            fc = np.ascontiguousarray(frame_counts, dtype=np.int32)
            assert fc.shape == (batch,)
        self._check(self._lib.gvtm_synthesize_batch_host(
            self._h, _ptr(params), _ptr(fc), batch, frames, _ptr(audio), stride, _ptr(counts), _ptr(maxabs)))
        return audio, counts, maxabs

    def synthesize_host_into(self, params, out, frame_counts=None, counts=None, maxabs=None, scales=None):
        """The host entries with caller-owned (e.g. page-locked) buffers: `out` float32 [B][stride] takes the unscaled
        samples (gvtm_synthesize_batch_host), `out` int16 [B][stride] the scaled 16-bit ones (.._host_pcm16)."""
        assert params.dtype == np.float32 and params.flags.c_contiguous and params.ndim == 3 and params.shape[2] == N_PARAM
        assert out.flags.c_contiguous and out.ndim == 2 and out.shape[0] == params.shape[0]
        batch, frames = params.shape[:2]
        fc = None
        if frame_counts is not None:
            fc = np.ascontiguousarray(frame_counts, dtype=np.int32)
        if out.dtype == np.int16:
            self._check(self._lib.gvtm_synthesize_batch_host_pcm16(
                self._h, _ptr(params), _ptr(fc), batch, frames, _ptr(out), out.shape[1], _ptr(counts), _ptr(maxabs), _ptr(scales)))
        else:
            assert out.dtype == np.float32 and scales is None
            self._check(self._lib.gvtm_synthesize_batch_host(
                self._h, _ptr(params), _ptr(fc), batch, frames, _ptr(out), out.shape[1], _ptr(counts), _ptr(maxabs)))

    def synthesize_host_pcm16(self, params, frame_counts=None):
        """-> (pcm int16 [B][stride], counts int64 [B], maxabs float32 [B], scales float32 [B])"""
        params = np.ascontiguousarray(params, dtype=np.float32)
        batch, frames = params.shape[:2]
        stride = self.output_count(frames) if frame_counts is None else self.output_capacity(frames)
        pcm = np.zeros((batch, stride), dtype=np.int16)
        counts = np.zeros(batch, dtype=np.int64)
        maxabs = np.zeros(batch, dtype=np.float32)
        scales = np.zeros(batch, dtype=np.float32)
        self.synthesize_host_into(params, pcm, frame_counts, counts, maxabs, scales)
        return pcm, counts, maxabs, scales

    def synthesize_events_device(self, track_config, d_events, d_offsets, batch, max_frames, d_audio, audio_stride,
                                 d_frame_counts=None, d_out_counts=None, d_maxabs=None, d_drift=None, stream=None):
        """Event lists in, samples out, one launch (gvtm_synthesize_events_device); all pointers are device memory."""
        self._check(self._lib.gvtm_synthesize_events_device(
            self._h, ctypes.byref(track_config), _ptr(d_events), _ptr(d_offsets), int(batch), int(max_frames), _ptr(d_audio),
            int(audio_stride), _ptr(d_frame_counts), _ptr(d_out_counts), _ptr(d_maxabs), _ptr(d_drift), _ptr(stream)))

    def normalize_device(self, d_audio, batch, audio_stride, d_maxabs, d_counts=None, d_out_f32=None, d_out_i16=None,
                         d_scales=None, stream=None):
        self._check(self._lib.gvtm_normalize_batch_device(
            self._h, _ptr(d_audio), int(batch), int(audio_stride), _ptr(d_counts), _ptr(d_maxabs), _ptr(d_out_f32),
            _ptr(d_out_i16), _ptr(d_scales), _ptr(stream)))


class Stream:
    """Owns a gvtm_stream: `batch` utterances of a plan synthesized piece by piece (include/gama_vtm.h, "Streams")."""

    def __init__(self, plan, batch):
        self._plan = plan  # keeps the plan alive
        self._lib = plan._lib
        self._h = ctypes.c_void_p()
        self.batch = int(batch)
        plan._check(self._lib.gvtm_stream_create(plan._h, self.batch, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.gvtm_stream_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._plan._check(self._lib.gvtm_stream_reset(self._h))

    def capacity(self, max_new_frames):
        return int(self._lib.gvtm_stream_capacity(self._h, int(max_new_frames)))

    def push(self, params, frame_counts=None):
        """params float32 [batch][F][16] -> list of float32 arrays, the new samples of each utterance."""
        params = np.ascontiguousarray(params, dtype=np.float32)
        assert params.ndim == 3 and params.shape[0] == self.batch and params.shape[2] == N_PARAM
        frames = params.shape[1]
        stride = self.capacity(frames)
        audio = np.zeros((self.batch, stride), dtype=np.float32)
        counts = np.zeros(self.batch, dtype=np.int64)
        fc = None
        if frame_counts is not None:
            fc = np.ascontiguousarray(frame_counts, dtype=np.int32)
        self._plan._check(self._lib.gvtm_stream_push(self._h, _ptr(params), _ptr(fc), frames, _ptr(audio), stride, _ptr(counts)))
        return [audio[b, : counts[b]].copy() for b in range(self.batch)]

    def finish(self):
        """-> (list of float32 arrays, maxabs float32 [batch])"""
        stride = self.capacity(0)
        audio = np.zeros((self.batch, stride), dtype=np.float32)
        counts = np.zeros(self.batch, dtype=np.int64)
        maxabs = np.zeros(self.batch, dtype=np.float32)
        self._plan._check(self._lib.gvtm_stream_finish(self._h, _ptr(audio), stride, _ptr(counts), _ptr(maxabs)))
        return [audio[b, : counts[b]].copy() for b in range(self.batch)], maxabs
