"""GPU-box debugging aid: the plugin's mode of operation (one parameter frame per internal step)."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gama_tts_amd as g
import oracle, tracks

tr = tracks.random_track(120, 5, True)
cfg = oracle.male_config()
ref = oracle.synthesize(cfg, tr)
# Controller::synthesize interpolation in float32
cs = 80
coef = np.float32(1.0) / np.float32(cs)
steps = []
ext = np.vstack([tr, tr[-1:]])
for i in range(1, len(ext)):
    cur = ext[i - 1].copy()
    delta = ((ext[i] - cur) * coef).astype(np.float32)
    for j in range(cs):
        steps.append(cur.copy())
        cur = (cur + delta).astype(np.float32)
steps = np.array(steps, dtype=np.float32)
plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE)), 20034.0, 0)
print("control_steps", plan.info.control_steps, "frames", steps.shape[0])
audio, counts, _ = plan.synthesize_host(steps[None])
out = audio[0, :counts[0]]
print(out.size, ref.size)
err = np.abs(out.astype(np.float64) - ref)
bad = np.nonzero(err > 1e-9 * np.abs(ref).max())[0]
print("max err", err.max() / np.abs(ref).max(), "first bad", bad[:5], "count", bad.size, "last bad", bad[-5:] if bad.size else None)
