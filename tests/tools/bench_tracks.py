"""GPU-box measurement of track generation alone (EventList::generateOutput): the first section of
bench_aux.py, for quick iteration on csrc/vtm_tracks.hip."""
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
import os
fl = [int(c) for c in os.environ.get('TRK_FLAGS', '1111')]
tc.control_period_ms, tc.macro_intonation, tc.micro_intonation, tc.intonation_drift, tc.smooth_intonation = 4, fl[0], fl[1], fl[2], fl[3]
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
print(json.dumps(out))
