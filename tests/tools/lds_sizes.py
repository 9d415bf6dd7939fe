"""LDS bytes per workgroup of every kernel shape (runs without a GPU; diagnostics library)."""
import ctypes
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gama_tts_amd as g  # noqa: E402
import oracle  # noqa: E402

lib = g.load_library(diagnostics=True)
lib.gvtm_debug_lds_bytes.restype = ctypes.c_size_t
lib.gvtm_debug_lds_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
cfgd = g.read_config_file(oracle.VOICE_MALE)
for prec, name in ((2, "f32"), (1, "mixed"), (0, "f64")):
    for delay in (1, 2, 3):
        for rate in (44100.0, 22050.0):
            plan = g.Plan(g.config_from_dict(cfgd, rate, delay, prec), 250.0, -1, diagnostics=True)
            print(name, "SectionDelay", delay, "out", rate, "KB for 1/2/4 utterances per workgroup:",
                  [round(lib.gvtm_debug_lds_bytes(plan._h, r) / 1024, 1) for r in (1, 2, 4)])
