"""GPU-box parity report for reference model 5 (VocalTractModel5<double,1>, 5_male voice, 48 kHz): batch 256 of
consonant-heavy tracks against the double oracle -> JSON on stdout."""
import json
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402

BATCH, FRAMES = 256, 250
params = tracks.random_tracks(BATCH, FRAMES, seed0=5000, consonant_heavy=True)


def ref(b):
    return oracle.synthesize5(oracle.male5_config(48000.0), params[b])[0]


out = {"workload": "batch 256 x 250 frames (1 s), consonant-heavy generator, seed 5000+b, 5_male voice, 48 kHz"}
plan = g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE)), 250.0, 0)
audio, counts, _ = plan.synthesize_host(params)
with ProcessPoolExecutor(8) as ex:
    refs = list(ex.map(ref, range(BATCH), chunksize=8))
errs, same = [], 0
for b in range(BATCH):
    r = refs[b]
    assert counts[b] == r.size
    errs.append(float(np.abs(audio[b, : r.size].astype(np.float64) - r).max() / np.abs(r).max()))
    same += int(np.array_equal(audio[b, : r.size], r))
out["fp64_vs_model5_double"] = {
    "samples_per_utterance": int(counts[0]), "counts_exact": True, "worst_peak_relative_error": max(errs),
    "median_peak_relative_error": float(np.median(errs)), "bit_identical_utterances": same, "utterances": BATCH,
    "bit_identical_samples_fraction": float(np.mean([np.mean(audio[b, : refs[b].size] == refs[b]) for b in range(BATCH)]))}
print(json.dumps(out))
