"""GPU-box parity report for BASELINE.json configs[2] (batch 256, consonant-heavy tracks: nasal branch and
frication active; VocalTractModel2<TFloat,1> = VocalTractModel0 semantics) -> JSON on stdout.
fp64 device path vs the double oracle: max|d|/max|x| per utterance, worst case over the batch, exact N;
float device path vs the float oracle: bit-identical utterances."""
import json, sys
from concurrent.futures import ProcessPoolExecutor
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gama_tts_amd as g
from gama_tts_amd import capi
import oracle, tracks

BATCH, FRAMES = 256, 500
params = tracks.random_tracks(BATCH, FRAMES, seed0=2000, consonant_heavy=True)

def ref(args):
    b, fm = args
    return oracle.synthesize(oracle.male_config(44100.0, 1, float_model=fm), params[b])

out = {"workload": "batch 256 x 500 frames, consonant-heavy generator (velum >= 0.5 on half the frames, fricVol >= 20 on half, aspVol >= 10 on 30 %), seed 2000+b"}
cfgd = g.read_config_file(oracle.VOICE_MALE)
with ProcessPoolExecutor(8) as ex:
    for name, prec, fm in (("fp64_vs_model2_double", capi.PRECISION_F64, 0), ("float_vs_model1_float", capi.PRECISION_F32, 1)):
        plan = g.Plan(g.config_from_dict(cfgd, 44100.0, 1, prec), 250.0, 0)
        audio, counts, _ = plan.synthesize_host(params)
        refs = list(ex.map(ref, [(b, fm) for b in range(BATCH)], chunksize=8))
        errs, same = [], 0
        for b in range(BATCH):
            r = refs[b]
            assert counts[b] == r.size
            errs.append(float(np.abs(audio[b, :r.size].astype(np.float64) - r).max() / np.abs(r).max()))
            same += int(np.array_equal(audio[b, :r.size], r))
        out[name] = {"samples_per_utterance": int(counts[0]), "counts_exact": True, "worst_peak_relative_error": max(errs),
                     "median_peak_relative_error": float(np.median(errs)), "bit_identical_utterances": same, "utterances": BATCH,
                     "bit_identical_samples_fraction": float(np.mean([np.mean(audio[b, :refs[b].size] == refs[b]) for b in range(BATCH)]))}
print(json.dumps(out))
