"""GPU-box report: the all-float device path (GVTM_PRECISION_F32) against the float oracle."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gama_tts_amd as g
from gama_tts_amd import capi
import oracle, tracks

def report(tag, got, ref):
    ref64 = ref.astype(np.float64)
    err = np.abs(got.astype(np.float64) - ref64).max() / max(np.abs(ref64).max(), 1e-300)
    same = float((got == ref).mean())
    first = int(np.argmax(got != ref)) if same < 1 else -1
    print("%-28s n=%7d peak-rel err %.3g  bit-identical %.4f first-diff %d" % (tag, ref.size, err, same, first), flush=True)

d = g.read_config_file(oracle.VOICE_MALE)
for delay, layout, frames in ((1, 0, 60), (2, 0, 60), (3, 0, 60), (1, 1, 60), (1, 0, 500)):
    plan = g.Plan(g.config_from_dict(d, 44100.0, delay, capi.PRECISION_F32, layout), 250.0, 0)
    params = tracks.random_tracks(3, frames, seed0=50 + delay, consonant_heavy=True)
    audio, counts, _ = plan.synthesize_host(params)
    cfg = oracle.male_config(44100.0, delay, layout, float_model=1)
    for b in range(3):
        ref = oracle.synthesize(cfg, params[b])
        assert counts[b] == ref.size
        report("delay %d layout %d frames %d #%d" % (delay, layout, frames, b), audio[b, :ref.size], ref)
