"""GPU-box A/B helper: kernel time of library variants (GVTM_LIBRARY) for one workload, one process each.
usage: python tools/ab.py <batch> <precision> [--frames F] [--delay D] <variant>...   ("default" = gama_tts_amd/lib)"""
import json, os, subprocess, sys
args = sys.argv[1:]
batch, prec = args[0], args[1]
rest = args[2:]
extra = []
while rest and rest[0].startswith("--"):
    extra += [rest[0], rest[1]]
    rest = rest[2:]
for v in rest:
    env = dict(os.environ)
    if v != "default":
        env["GVTM_LIBRARY"] = os.path.join("gama_tts_amd", "lib_variants", "libgama_vtm_%s.so" % v)
    r = subprocess.run([sys.executable, "bench.py", "--precision", prec, "--batch", batch, "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras",
                        "--no-end-to-end", "--no-parity-check"] + extra, capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-10s batch %s %s %s: kernel %.3f ms  %.2f G samples/s" % (v, batch, prec, " ".join(extra), d["roofline"]["kernel_ms"], d["value"] / 1e9), flush=True)
    except Exception:
        print(v, "FAILED", r.stderr[-400:], flush=True)
