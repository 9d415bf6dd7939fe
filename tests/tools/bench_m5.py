#!/usr/bin/env python3
"""GPU-box measurement of the reference-model-5 kernel (VocalTractModel5<double,1>): configs[1]-shaped batches
(2 s utterances, 5_male voice, 48 kHz), device time from the plan's HIP events, the compiled reference
(oracle/_ref/ref_vtm_o3, model 5) timed beside it on one host core.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gama_tts_amd as g  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402


def main():
    batches = [int(x) for x in sys.argv[1:]] or [256, 1024, 4096]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    plan = g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE)), 250.0, 0)
    frames = 500
    n = plan.output_count(frames)
    out = {"internal_rate_hz": plan.info.internal_rate_hz, "steps_per_utterance": plan.info.control_steps * frames, "runs": []}
    base = tracks.random_tracks(64, frames, seed0=1000, consonant_heavy=True)
    for batch in batches:
        params = torch.from_numpy(np.ascontiguousarray(np.tile(base, ((batch + 63) // 64, 1, 1))[:batch])).to(dev)
        audio = torch.empty((batch, n), dtype=torch.float32, device=dev)
        counts = torch.zeros(batch, dtype=torch.int64, device=dev)
        maxabs = torch.zeros(batch, dtype=torch.float32, device=dev)
        run = lambda: plan.synthesize_device(params, batch, frames, audio, n, None, counts, maxabs, stream)  # noqa: E731
        run()
        torch.cuda.synchronize()
        plan.set_timing(True)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ms, launches = plan.take_kernel_ms()
        plan.set_timing(False)
        samples = int(counts.sum().item())
        out["runs"].append({"batch": batch, "ms": ms, "launches": launches, "samples_per_s": samples / (ms * 1e-3),
                            "x_realtime": samples / 48000.0 / (ms * 1e-3),
                            "ns_per_step_per_utterance_slot": ms * 1e6 / (plan.info.control_steps * frames)})
        del params, audio
    if oracle.ref_binary("o3"):
        _, info = oracle.ref_synthesize(base[0], "5", 48000, 250, config=oracle.VOICE5_MALE, kind="o3", repeat=8)
        out["cpu_reference"] = {"ns_per_step": float(info["ns_per_step"]), "samples_per_s": n * 8 / float(info["sec"]), "cores": 1}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
