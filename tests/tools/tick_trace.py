"""GPU-box diagnostic: timeline of two ticks of workgroup 0 (library variant built with -DGVTM_TUNE_TRACE=<tick>)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
delay = int(sys.argv[4]) if len(sys.argv) > 4 else 2
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
lib.gvtm_debug_set_phase_cycles(plan._h, ctypes.c_void_p(d_cyc.data_ptr()))
plan.synthesize_device(d_params, batch, frames, d_audio, n)
torch.cuda.synchronize()
tr = d_cyc.cpu().numpy().reshape(-1)[32768:32768 + 16 * 128].reshape(16, 128).astype(np.uint64)
t0 = min(int(v & np.uint64(0xFFFFFFFFFFFF)) for v in tr.reshape(-1) if v)
names = {1: "tick", 2: "stage-done", 3: "barrier"}
for w in range(16):
    ev = [(int(v >> np.uint64(48)), int(v & np.uint64(0xFFFFFFFFFFFF)) - t0) for v in tr[w] if v]
    if not ev:
        continue
    out = []
    for i, c in ev:
        if i & 0x100:
            out.append("%d:get%d" % (c, i & 0xFF))
        elif i & 0x200:
            out.append("%d:end%d" % (c, i & 0xFF))
        else:
            out.append("%d:%s" % (c, names.get(i, str(i))))
    print("wave %2d  %s" % (w, "  ".join(out)))
