"""GPU-box soak: ragged random batches through every model / precision against the oracle -> JSON on stdout.
Float paths must be bit-identical; fp64 paths within one float32 ulp of the sample or 1e-9 of peak; model 5 within 2e-6 of
peak (see tests/test_gpu_model5.py)."""
import json
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
from gama_tts_amd import capi  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402

BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 768
MAXF = int(sys.argv[2]) if len(sys.argv) > 2 else 96
rng = np.random.default_rng(20261004)
frames = rng.integers(0, MAXF + 1, size=BATCH).astype(np.int32)
frames[:4] = [0, 1, 2, MAXF]
params = tracks.random_tracks(BATCH, MAXF, seed0=777000, consonant_heavy=True)
params[::5] = tracks.random_tracks(len(params[::5]), MAXF, seed0=888000, consonant_heavy=False)

CASES = [  # name, delay, layout, precision, float_model, rate
    ("model1_float", 1, 0, capi.PRECISION_F32, 1, 44100.0),
    ("model2f_d2_float", 2, 0, capi.PRECISION_F32, 1, 44100.0),
    ("model2f_d3_float_48k", 3, 0, capi.PRECISION_F32, 1, 48000.0),
    ("model4f_float", 1, 1, capi.PRECISION_F32, 1, 44100.0),
    ("model0_double", 1, 0, capi.PRECISION_F64, 0, 44100.0),
    ("model3_double", 3, 0, capi.PRECISION_F64, 0, 44100.0),
    ("model4_double_22k", 1, 1, capi.PRECISION_F64, 0, 22050.0),
]


def ref_case(args):
    b, f, delay, layout, fm, rate = args
    return oracle.synthesize(oracle.male_config(rate, delay, layout, float_model=fm), params[b, :f])


def ref5(args):
    b, f = args
    return oracle.synthesize5(oracle.male5_config(48000.0), params[b, :f])[0]


def supported(plan, want):
    """Frame counts the device path serves: a count that would trigger the reference converter's flush overrun (refused,
    DESIGN.md section 5) is lowered until it does not."""
    got = want.copy()
    for b in range(got.size):
        while True:
            try:
                plan.output_count(int(got[b]))
                break
            except g.GvtmError:
                got[b] -= 1
    return got


def summarize(audio, counts, refs, exact_required):
    worst, same, bad_counts = 0.0, 0, 0
    for b in range(BATCH):
        r = refs[b]
        bad_counts += int(counts[b] != r.size)
        got = audio[b, : r.size]
        same += int(np.array_equal(got, r))
        peak = float(np.abs(r).max()) if r.size else 0.0
        if peak > 0:
            worst = max(worst, float(np.abs(got.astype(np.float64) - r).max() / peak))
    return {"utterances": BATCH, "wrong_counts": bad_counts, "bit_identical_utterances": same, "worst_peak_relative_error": worst,
            "pass": bool(bad_counts == 0 and (same == BATCH if exact_required else worst < 2e-6))}


out = {"workload": "%d utterances of 0..%d frames (ragged, one launch per case), random + consonant-heavy tracks" % (BATCH, MAXF)}
cfgd = g.read_config_file(oracle.VOICE_MALE)
with ProcessPoolExecutor(8) as ex:
    for name, delay, layout, prec, fm, rate in CASES:
        plan = g.Plan(g.config_from_dict(cfgd, rate, delay, prec, layout), 250.0, 0)
        fr = supported(plan, frames)
        audio, counts, _ = plan.synthesize_host(params, fr)
        refs = list(ex.map(ref_case, [(b, int(fr[b]), delay, layout, fm, rate) for b in range(BATCH)], chunksize=16))
        out[name] = summarize(audio, counts, refs, exact_required=bool(fm))
        out[name]["frame_counts_lowered"] = int((fr != frames).sum())
        print(name, out[name], file=sys.stderr, flush=True)
    plan = g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE)), 250.0, 0)
    fr = supported(plan, frames)
    audio, counts, _ = plan.synthesize_host(params, fr)
    refs = list(ex.map(ref5, [(b, int(fr[b])) for b in range(BATCH)], chunksize=16))
    out["model5_double"] = summarize(audio, counts, refs, exact_required=False)
    print("model5_double", out["model5_double"], file=sys.stderr, flush=True)
print(json.dumps(out))
