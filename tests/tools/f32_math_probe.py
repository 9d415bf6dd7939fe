"""GPU-box check: the all-float path's device conversions against this machine's libm (float)."""
import ctypes, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gama_tts_amd as g
from gama_tts_amd import capi
import oracle

plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), precision=capi.PRECISION_F32), 250.0, 0, diagnostics=True)
lib = g.load_library(diagnostics=True)
lib.gvtm_debug_device_float_math.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
ol = oracle.lib()
ol.vtmo_libm_powf.argtypes = [ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
ol.vtmo_libm_tanf_cosf.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
rng = np.random.default_rng(3)

def dev(kind, x):
    out = np.empty_like(x)
    assert lib.gvtm_debug_device_float_math(plan._h, kind, x.ctypes.data, x.size, out.ctypes.data) == 0
    return out

def powf(base, x):
    out = np.empty_like(x)
    ol.vtmo_libm_powf(base, x.ctypes.data, x.size, out.ctypes.data)
    return out

pitch = rng.uniform(-30, 10, 2000000).astype(np.float32)
ref = np.float32(220.0) * powf(2.0, (pitch + np.float32(3.0)) * np.float32(1.0 / 12.0))
got = dev(0, pitch)
print("frequency: mismatches %d of %d" % (int((ref.view(np.uint32) != got.view(np.uint32)).sum()), pitch.size))
db = rng.uniform(-1, 61, 2000000).astype(np.float32)
r = powf(10.0, (db - np.float32(60.0)) * np.float32(1.0 / 20.0))
ref = np.where(db <= 0, np.float32(0), np.where(db == 60, np.float32(1), r)).astype(np.float32)
got = dev(1, db)
print("amplitude: mismatches %d of %d" % (int((ref.view(np.uint32) != got.view(np.uint32)).sum()), db.size))
for which, name, hi in ((0, "tan", 0.78), (1, "cos", 3.1)):
    x = rng.uniform(0.0, hi, 2000000).astype(np.float32)
    ref = np.empty_like(x)
    ol.vtmo_libm_tanf_cosf(which, x.ctypes.data, x.size, ref.ctypes.data)
    got = dev(2 + which, x)
    d = ref.view(np.int32).astype(np.int64) - got.view(np.int32).astype(np.int64)
    print("%s: mismatches %d of %d (max ulp %d)" % (name, int((d != 0).sum()), x.size, int(np.abs(d).max())))
