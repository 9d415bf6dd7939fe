"""GPU-box diagnostic: kernel time by batch size and utterances per workgroup (forced through the diagnostics library)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402

frames = 500
cfgd = g.read_config_file(oracle.VOICE_MALE)
dev = torch.device("cuda:0")
pool = tracks.random_tracks(64, frames, seed0=1000)
for prec, name in ((2, "f32"), (1, "mixed"), (0, "f64")):
    for batch in (64, 128, 256, 384, 512, 640, 768, 1024, 2048):
        params = np.concatenate([pool] * ((batch + 63) // 64))[:batch]
        d_params = torch.from_numpy(params).to(dev)
        line = []
        for rows in (1, 2, 4):
            plan = g.Plan(g.config_from_dict(cfgd, 44100.0, 1, prec), 250.0, 0, diagnostics=True, rows=rows)
            n = plan.output_count(frames)
            d_audio = torch.zeros((batch, n), dtype=torch.float32, device=dev)
            for _ in range(2):
                plan.synthesize_device(d_params, batch, frames, d_audio, n)
            torch.cuda.synchronize()
            plan.set_timing(True)
            for _ in range(5):
                plan.synthesize_device(d_params, batch, frames, d_audio, n)
            torch.cuda.synchronize()
            ms, cnt = plan.take_kernel_ms()  # the average over the launches since the last call
            line.append("rows %d %.3f ms" % (rows, ms))
        print(name, "batch", batch, " | ".join(line), flush=True)
