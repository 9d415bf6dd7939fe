"""GPU-box: kernel time of the other model semantics behind the same entry point (model 4 = the 30+18-section tube,
model 3 = SectionDelay 3, model 0/1 for reference), fp64 and float, batch 256 .. 4096 x 2 s.
usage: python tests/tools/bench_models.py [layout1-only]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
from gama_tts_amd import capi  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
base = tracks.random_tracks(64, 500, seed0=1000)
only_wide = len(sys.argv) > 1
for prec, name in ((capi.PRECISION_F64, "f64"), (capi.PRECISION_MIXED, "mixed"), (capi.PRECISION_F32, "f32")):
    for layout, delay in ((1, 1),) if only_wide else ((1, 1), (0, 3), (0, 1)):
        plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, delay, prec, layout), 250.0, 0)
        n = plan.output_count(500)
        for batch in (256, 512, 1024, 4096):
            params = torch.from_numpy(np.tile(base, (batch // 64, 1, 1))).to(dev)
            audio = torch.empty((batch, n), dtype=torch.float32, device=dev)
            counts = torch.zeros(batch, dtype=torch.int64, device=dev)
            run = lambda: plan.synthesize_device(params, batch, 500, audio, n, None, counts, None, stream)  # noqa: E731
            run()
            torch.cuda.synchronize()
            plan.set_timing(True)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ms, _ = plan.take_kernel_ms()
            plan.set_timing(False)
            print("%s layout %d delay %d batch %d: %.2f ms, %.2f G samples/s, %.0f ns/step" % (
                name, layout, delay, batch, ms, batch * n / ms / 1e6, ms * 1e6 / (plan.info.control_steps * 500)), flush=True)
