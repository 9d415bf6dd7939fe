"""GPU-box debugging aid: compares per-step intermediates of the HIP kernel with the oracle."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
from gama_tts_amd import capi  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402

delay = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
layout = int(sys.argv[4]) if len(sys.argv) > 4 else 0
params = tracks.random_tracks(2, frames, seed0=42, consonant_heavy=True)
cfgd = g.read_config_file(oracle.VOICE_MALE)
plan = g.Plan(g.config_from_dict(cfgd, 44100.0, delay, prec, layout), 250.0, 0, diagnostics=True)
steps = frames * plan.info.control_steps
n = plan.output_count(frames)
dev = torch.device("cuda:0")
d_params = torch.from_numpy(params).to(dev)
d_audio = torch.zeros((2, n), dtype=torch.float32, device=dev)
d_taps = torch.zeros((2, steps, 8), dtype=torch.float64, device=dev)
lib = g.load_library(diagnostics=True)
lib.gvtm_debug_set_taps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.gvtm_debug_set_taps(plan._h, ctypes.c_void_p(d_taps.data_ptr()))
plan.synthesize_device(d_params, 2, frames, d_audio, n)
torch.cuda.synchronize()
taps = d_taps.cpu().numpy()
audio = d_audio.cpu().numpy()
names = ["u", "sig", "thr", "fir", "lpnoise", "pos0", "pos1", "x"]
for b in range(2):
    ref_audio, ref_taps = oracle.synthesize_debug(oracle.male_config(44100.0, delay, layout, float_model=int(prec == 2)), params[b])
    for i, nm in enumerate(names):
        d = np.abs(taps[b, :, i] - ref_taps[:, i])
        first = int(np.argmax(d > 1e-9 * max(1e-30, np.abs(ref_taps[:, i]).max()))) if d.max() > 0 else -1
        exact = float((taps[b, :, i] == ref_taps[:, i]).mean())
        print("utt %d %-8s max|ref| %.3e  max|diff| %.3e  first-bad-step %d  bit-identical %.4f" % (b, nm, np.abs(ref_taps[:, i]).max(), d.max(), first, exact))
    e = np.abs(audio[b].astype(np.float64) - ref_audio)
    print("utt %d audio   max|ref| %.3e  max|diff| %.3e first-bad %d" % (b, np.abs(ref_audio).max(), e.max(), int(np.argmax(e > 1e-9))))
if len(sys.argv) > 5:
    b = 0
    ref_audio, ref_taps = oracle.synthesize_debug(oracle.male_config(44100.0, delay, layout, float_model=int(prec == 2)), params[b])
    np.set_printoptions(precision=10, linewidth=200)
    for i, nm in enumerate(names):
        bad = np.nonzero(taps[b, :, i] != ref_taps[:, i])[0][:6]
        print(nm, "first mismatching steps", bad.tolist())
        for s in bad[:3]:
            print("   step %d dev %r ref %r" % (s, float(taps[b, s, i]).hex(), float(ref_taps[s, i]).hex()))
