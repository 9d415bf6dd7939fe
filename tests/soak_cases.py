"""Ragged random batches through every model / precision against the oracle: shared by tests/test_gpu_soak.py (reduced,
-m gpu) and tests/tools/soak.py (large, builder-run).  Float paths must be bit-identical; fp64 paths within one float32
ulp of the sample or 1e-9 of peak; mixed within 1e-5 of peak (north_star's bar; measured ~3e-7); model 5 within 2e-6 of peak (tests/test_gpu_model5.py explains why)."""
import numpy as np

import gama_tts_amd as g
from gama_tts_amd import capi
import oracle
import tracks

CASES = [  # name, delay, layout, precision, float_model, rate
    ("model1_float", 1, 0, capi.PRECISION_F32, 1, 44100.0),
    ("model2f_d2_float", 2, 0, capi.PRECISION_F32, 1, 44100.0),
    ("model2f_d3_float_48k", 3, 0, capi.PRECISION_F32, 1, 48000.0),
    ("model4f_float", 1, 1, capi.PRECISION_F32, 1, 44100.0),
    ("model2f_d2_float_22k", 2, 0, capi.PRECISION_F32, 1, 22050.0),  # flush overruns at 18, 79, 567 ... frames
    ("model0_double", 1, 0, capi.PRECISION_F64, 0, 44100.0),
    ("model3_double", 3, 0, capi.PRECISION_F64, 0, 44100.0),
    ("model4_double_22k", 1, 1, capi.PRECISION_F64, 0, 22050.0),
    ("model0_mixed", 1, 0, capi.PRECISION_MIXED, 0, 44100.0),
]

def make_inputs(batch, max_frames, seed=20261004):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, max_frames + 1, size=batch).astype(np.int32)
    frames[:4] = [0, 1, 2, max_frames]
    params = tracks.random_tracks(batch, max_frames, seed0=777000, consonant_heavy=True)
    params[::5] = tracks.random_tracks(len(params[::5]), max_frames, seed0=888000, consonant_heavy=False)
    return params, frames


def overrun_lengths(plan, max_frames):
    """Frame counts <= max_frames at which the reference converter's flush overrun happens (down-sampling plans)."""
    i = plan.info
    out = []
    for f in range(max_frames + 1):
        fills = f * i.control_steps + 2 * i.pad_size
        if plan.output_count(f) != -((-(fills << 16)) // i.time_register_increment):
            out.append(f)
    return out


def with_overruns(plan, frames, max_frames, slots=6):
    """`frames` with up to `slots` entries (after the first four) replaced by flush-overrun lengths of this plan."""
    fr = frames.copy()
    hits = overrun_lengths(plan, max_frames)
    for j, f in enumerate(hits[-slots:]):
        fr[4 + j] = f
    return fr, len(hits[-slots:])


def summarize(audio, counts, refs, exact_required, tol):
    worst, same, bad_counts = 0.0, 0, 0
    for b, r in enumerate(refs):
        bad_counts += int(counts[b] != r.size)
        got = audio[b, : r.size]
        same += int(np.array_equal(got, r))
        peak = float(np.abs(r).max()) if r.size else 0.0
        if peak > 0 and got.size == r.size:
            worst = max(worst, float(np.abs(got.astype(np.float64) - r).max() / peak))
    n = len(refs)
    return {"utterances": n, "wrong_counts": bad_counts, "bit_identical_utterances": same, "worst_peak_relative_error": worst,
            "pass": bool(bad_counts == 0 and (same == n if exact_required else worst < tol))}


def run(batch, max_frames, workers=8, names=None, log=None):
    """-> {case name: summary}.  One launch per case; the references come from the oracle in spawned worker processes
    (oracle.synthesize_many: every job carries its own track, nothing is inherited from this process)."""
    params, frames = make_inputs(batch, max_frames)
    out = {}
    cfgd = g.read_config_file(oracle.VOICE_MALE)
    for name, delay, layout, prec, fm, rate in CASES:
        if names is not None and name not in names:
            continue
        plan = g.Plan(g.config_from_dict(cfgd, rate, delay, prec, layout), 250.0, 0)
        fr, n_over = with_overruns(plan, frames, max_frames)
        audio, counts, _ = plan.synthesize_host(params, fr)
        refs = oracle.synthesize_many([(params[b, : int(fr[b])], rate, delay, layout, fm) for b in range(batch)], workers)
        out[name] = summarize(audio, counts, refs, exact_required=bool(fm), tol=1e-5 if prec == capi.PRECISION_MIXED else 2e-7)
        out[name]["flush_overrun_utterances"] = n_over
        out[name]["frame_counts_lowered"] = 0
        if log:
            log(name, out[name])
    if names is None or "model5_double" in names:
        plan = g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE)), 250.0, 0)
        fr, n_over = with_overruns(plan, frames, max_frames)
        audio, counts, _ = plan.synthesize_host(params, fr)
        refs = oracle.synthesize_many([(params[b, : int(fr[b])], 48000.0) for b in range(batch)], workers, model5=True)
        out["model5_double"] = summarize(audio, counts, refs, exact_required=False, tol=2e-6)
        out["model5_double"]["flush_overrun_utterances"] = n_over
        out["model5_double"]["frame_counts_lowered"] = 0
        if log:
            log("model5_double", out["model5_double"])
    return out
