"""GPU-box diagnostic: where does a workgroup of vtm_synth_kernel spend its cycles?"""
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

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 500
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
delay = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pool = tracks.random_tracks(min(batch, 64), frames, seed0=1000)
params = np.concatenate([pool] * ((batch + len(pool) - 1) // len(pool)))[:batch]
cfgd = g.read_config_file(oracle.VOICE_MALE)
plan = g.Plan(g.config_from_dict(cfgd, 44100.0, delay, prec), 250.0, 0, diagnostics=True)
n = plan.output_count(frames)
dev = torch.device("cuda:0")
d_params = torch.from_numpy(params).to(dev)
d_audio = torch.zeros((batch, n), dtype=torch.float32, device=dev)
d_cyc = torch.zeros((batch, 16), dtype=torch.int64, device=dev)
lib = g.load_library(diagnostics=True)
lib.gvtm_debug_set_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
for rep in range(2):
    lib.gvtm_debug_set_phase_cycles(plan._h, ctypes.c_void_p(d_cyc.data_ptr()))
    plan.set_timing(True)
    plan.synthesize_device(d_params, batch, frames, d_audio, n)
    torch.cuda.synchronize()
    ms, _ = plan.take_kernel_ms()
cyc = d_cyc.cpu().numpy().astype(np.float64)
steps = frames * plan.info.control_steps
names = ["P1 interp", "P2 convert", "P3 scan", "P4a lookup", "P4b fir+mix", "P5 tube", "P6 src", "carry"]
tot = cyc.sum(axis=1).mean()
print("batch %d frames %d prec %d delay %d: kernel %.3f ms, mean cycles/workgroup %.3e (%.1f cycles/step)" % (
    batch, frames, prec, delay, ms, tot, tot / steps))
for i, nm in enumerate(names):
    print("  %-12s %6.1f cycles/step  %5.1f %%" % (nm, cyc[:, i].mean() / steps, 100 * cyc[:, i].mean() / tot))
