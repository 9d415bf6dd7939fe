"""GPU-box check of the DPP lane directions the tube wavefront relies on."""
import ctypes, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gama_tts_amd as g
import oracle
plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE)), 250.0, 0, diagnostics=True)
lib = g.load_library(diagnostics=True)
out = np.zeros(640, dtype=np.int32)
lib.gvtm_debug_dpp_selftest.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
rc = lib.gvtm_debug_dpp_selftest(plan._h, out.ctypes.data)
print("rc", rc)
for name, row in zip(("from_left (row_shr:1)", "from_right (row_shl:1)", "row_ror:6", "row_ror:10", "wave_shr:1", "wave_shl:1",
                      "row_newbcast:3 (b64)", "row_newbcast:10 (b32)", "row_shr:1 (b32)", "row_ror:10 (b32)"), out.reshape(10, 64)):
    print("%-24s" % name, row[:32].tolist())
lanes = np.arange(64)
assert (out[0:64] == np.where(lanes % 16 == 0, 0, lanes - 1)).all()
assert (out[64:128] == np.where(lanes % 16 == 15, 0, lanes + 1)).all()
assert (out[128:192] == (lanes // 16) * 16 + (lanes - 6) % 16).all()
assert (out[192:256] == (lanes // 16) * 16 + (lanes - 10) % 16).all()
assert (out[256:320] == np.where(lanes == 0, 0, lanes - 1)).all()
assert (out[320:384] == np.where(lanes == 63, 0, lanes + 1)).all()
assert (out[384:448] == (lanes // 16) * 16 + 3).all()
assert (out[448:512] == (lanes // 16) * 16 + 10).all()
assert (out[512:576] == np.where(lanes % 16 == 0, 0, lanes - 1)).all()
assert (out[576:640] == (lanes // 16) * 16 + (lanes - 10) % 16).all()
print("dpp directions ok")
