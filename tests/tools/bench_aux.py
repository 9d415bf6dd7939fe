"""GPU-box measurement of the two HBM-bound kernels beside the synthesis kernel:
track generation (EventList::generateOutput) and output scaling (a16)."""
import json, sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gama_tts_amd as g
from gama_tts_amd import capi
import event_lists, oracle

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

out = {}
# --- track generation: batch 4096, ~500 frames each
batch = 4096
tables = [event_lists.random_event_table(s, n_events=80, control_period=4, max_gap_periods=12) for s in range(64)]
cfgv = np.array([4, 1, 1, 1, 1, -20.0, -6.0, 4.0, 250.0, 4.0])
tc = g.TrackConfig()
tc.control_period_ms, tc.macro_intonation, tc.micro_intonation, tc.intonation_drift, tc.smooth_intonation = 4, 1, 1, 1, 1
tc.initial_pitch, tc.mean_pitch, tc.drift_deviation, tc.drift_sample_rate, tc.drift_lowpass_cutoff = -20.0, -6.0, 4.0, 250.0, 4.0
evs = [capi.events_from_table(tables[b % 64]) for b in range(batch)]
frames = [capi.tracks_frame_count(tc, e) for e in evs[:64]]
max_frames = max(frames)
offsets = np.zeros(batch + 1, dtype=np.int64); offsets[1:] = np.cumsum([len(e) for e in evs])
d_events = torch.from_numpy(np.concatenate(evs).view(np.uint8)).to(dev)
d_offsets = torch.from_numpy(offsets).to(dev)
d_params = torch.zeros((batch, max_frames, 16), dtype=torch.float32, device=dev)
d_counts = torch.zeros(batch, dtype=torch.int32, device=dev)
ms = timed(lambda: capi.generate_tracks_device(tc, d_events, d_offsets, batch, max_frames, d_params, d_counts, None, stream))
total_frames = int(d_counts.sum().item())
bytes_alg = total_frames * 64 + int(offsets[-1]) * 296
out["tracks"] = {"batch": batch, "frames_total": total_frames, "events_total": int(offsets[-1]), "ms": ms,
                 "algorithmic_GBps": bytes_alg / (ms * 1e-3) / 1e9, "frames_per_s": total_frames / (ms * 1e-3)}
# --- output scaling: 4096 x 88108 samples
plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 1), 250.0, 0)
n = plan.output_count(500)
d_audio = torch.randn((batch, n), dtype=torch.float32, device=dev)
d_max = d_audio.abs().amax(dim=1).contiguous()
d_f32 = torch.empty_like(d_audio)
d_i16 = torch.empty((batch, n), dtype=torch.int16, device=dev)
ms32 = timed(lambda: plan.normalize_device(d_audio, batch, n, d_max, None, d_out_f32=d_f32, stream=stream), 10)
ms16 = timed(lambda: plan.normalize_device(d_audio, batch, n, d_max, None, d_out_i16=d_i16, stream=stream), 10)
out["normalize_f32"] = {"samples": batch * n, "ms": ms32, "GBps": batch * n * 8 / (ms32 * 1e-3) / 1e9}
out["normalize_i16"] = {"samples": batch * n, "ms": ms16, "GBps": batch * n * 6 / (ms16 * 1e-3) / 1e9}
# --- host-buffer entry (H2D + kernel + D2H, synchronous): the PCIe-inclusive rate; host arrays allocated once
import ctypes, time, tracks
lib = g.load_library()
pool = tracks.random_tracks(64, 500, seed0=1000)
for bsz in (256, 4096):
    params = np.ascontiguousarray(np.tile(pool, (bsz // 64, 1, 1)))
    for prec, name in ((capi.PRECISION_F32, "f32"), (capi.PRECISION_F64, "f64")):
        pl = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 1, prec), 250.0, 0)
        n_out = pl.output_count(500)
        audio = np.empty((bsz, n_out), dtype=np.float32)
        counts = np.zeros(bsz, dtype=np.int64)
        peaks = np.zeros(bsz, dtype=np.float32)
        call = lambda: lib.gvtm_synthesize_batch_host(pl._h, ctypes.c_void_p(params.ctypes.data), None, bsz, 500,
                                                      ctypes.c_void_p(audio.ctypes.data), n_out, ctypes.c_void_p(counts.ctypes.data),
                                                      ctypes.c_void_p(peaks.ctypes.data))
        assert call() == 0
        t0 = time.perf_counter()
        for _ in range(3):
            assert call() == 0
        dt = (time.perf_counter() - t0) / 3
        out["host_entry_%s_batch%d" % (name, bsz)] = {"batch": bsz, "ms": dt * 1e3, "samples_per_s": float(counts.sum()) / dt}
# --- the same entry with page-locked buffers (gvtm_host_alloc), float32 and int16 output: what the three-stream pipeline gives
bsz = 4096
params = g.PinnedArray((bsz, 500, 16), np.float32)
params.array[...] = np.tile(pool, (bsz // 64, 1, 1))
for prec, name in ((capi.PRECISION_F32, "f32"), (capi.PRECISION_MIXED, "mixed"), (capi.PRECISION_F64, "f64")):
    pl = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 1, prec), 250.0, 0)
    n_out = pl.output_count(500)
    raw = g.PinnedArray((bsz * n_out,), np.float32)
    counts = np.zeros(bsz, dtype=np.int64)
    for kind in ("int16", "float32"):
        buf = raw.array.view(np.int16)[: bsz * n_out].reshape(bsz, n_out) if kind == "int16" else raw.array.reshape(bsz, n_out)
        pl.synthesize_host_into(params.array, buf, None, counts, None)
        pl.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(3):
            pl.synthesize_host_into(params.array, buf, None, counts, None)
        dt = (time.perf_counter() - t0) / 3
        kms, launches = pl.take_kernel_ms()
        pl.set_timing(False)
        kernels_ms = kms * launches / 3  # all slices of one call
        out["host_entry_pinned_%s_%s_batch%d" % (name, kind, bsz)] = {
            "batch": bsz, "ms": dt * 1e3, "samples_per_s": float(counts.sum()) / dt, "synthesis_kernels_ms": kernels_ms,
            "vs_kernels_only": kernels_ms / (dt * 1e3)}
    raw.close()
params.close()
print(json.dumps(out))
