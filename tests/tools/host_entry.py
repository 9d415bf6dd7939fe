"""GPU-box: the host-buffer entry point (H2D + kernel + D2H, synchronous) with pageable and with pinned host memory."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
from gama_tts_amd import capi  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402

lib = g.load_library()
for batch in (256, 1024, 4096):
    base = tracks.random_tracks(64, 500, seed0=1000)
    params = np.ascontiguousarray(np.tile(base, (batch // 64, 1, 1)))
    plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 1, capi.PRECISION_F32), 250.0, 0)
    n = plan.output_count(500)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            p = torch.from_numpy(params).pin_memory()
            a = torch.empty((batch, n), dtype=torch.float32).pin_memory()
            c = torch.zeros(batch, dtype=torch.int64).pin_memory()
            m = torch.zeros(batch, dtype=torch.float32).pin_memory()
            pp, ap, cp, mp = p.data_ptr(), a.data_ptr(), c.data_ptr(), m.data_ptr()
        else:
            a = np.empty((batch, n), dtype=np.float32)
            c = np.zeros(batch, dtype=np.int64)
            m = np.zeros(batch, dtype=np.float32)
            pp, ap, cp, mp = params.ctypes.data, a.ctypes.data, c.ctypes.data, m.ctypes.data
        call = lambda: lib.gvtm_synthesize_batch_host(plan._h, ctypes.c_void_p(pp), None, batch, 500, ctypes.c_void_p(ap), n, ctypes.c_void_p(cp), ctypes.c_void_p(mp))  # noqa: E731
        assert call() == 0
        t0 = time.perf_counter()
        for _ in range(3):
            assert call() == 0
        dt = (time.perf_counter() - t0) / 3
        print("batch %d %s: %.2f ms  %.2f G samples/s" % (batch, kind, dt * 1e3, batch * n / dt / 1e9), flush=True)
