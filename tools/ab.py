"""GPU-box A/B helper: kernel time of library variants (GVTM_LIBRARY) for one workload, one process each."""
import json, os, subprocess, sys
batch, prec = sys.argv[1], sys.argv[2]
variants = sys.argv[3:]
for v in variants:
    env = dict(os.environ)
    if v != "default":
        env["GVTM_LIBRARY"] = os.path.join("gama_tts_amd", "lib_variants", "libgama_vtm_%s.so" % v)
    r = subprocess.run([sys.executable, "bench.py", "--precision", prec, "--batch", batch, "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--no-end-to-end", "--no-parity-check"],
                       capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-10s batch %s %s: kernel %.3f ms  %.2f G samples/s" % (v, batch, prec, d["roofline"]["kernel_ms"], d["value"] / 1e9), flush=True)
    except Exception:
        print(v, "FAILED", r.stderr[-400:], flush=True)
