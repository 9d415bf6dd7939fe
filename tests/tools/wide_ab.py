import os, subprocess, sys
for v in ["default", "wide_a", "wide_b", "wide_c"]:
    for rows in ("2", "4"):
        env = dict(os.environ)
        if v != "default": env["GVTM_LIBRARY"] = os.path.join("gama_tts_amd", "lib_variants", "libgama_vtm_%s.so" % v)
        r = subprocess.run([sys.executable, "tests/tools/bench_models.py", "wide"], capture_output=True, text=True, env=env)
        for l in r.stdout.splitlines():
            if "batch 4096" in l or "batch 1024" in l: print(v, "rows", rows, l, flush=True)
