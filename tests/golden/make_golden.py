#!/usr/bin/env python3
"""Generates tests/golden/vtm_golden.npz by running the REAL reference.

Runs only in the build container: it executes oracle/_ref/ref_vtm and
oracle/_ref/ref_tts_capture, both compiled from /root/reference in place by
oracle/Makefile (`make -C oracle ref ref_full`), with -O2 -ffp-contract=off and
no -march.  The .npz holds data only: input frames, reference
output samples (or their digest) and integer counts.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_cases  # noqa: E402
import oracle  # noqa: E402

REF_VOICE_DIR = "/root/reference/data/voice/english/0_male"


def main():
    cap = os.path.join(oracle.REF_DIR, "ref_tts_capture")
    out = {}
    existing = os.path.join(HERE, "vtm_golden.npz")
    if os.path.exists(existing) and "--recapture" not in sys.argv:
        # `gama_tts tts` draws its intonation from std::random_device: keep the
        # frames captured once, so that regenerating the file does not change existing vectors
        out["hello_params"] = np.load(existing, allow_pickle=False)["hello_params"]
    else:
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "hello.f32")
            subprocess.run([cap, REF_VOICE_DIR, "Hello world.", p], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            out["hello_params"] = np.fromfile(p, dtype=np.float32).reshape(-1, 16)
    base = oracle.read_config_file(oracle.VOICE_MALE)
    manifest = {}
    for case in golden_cases.CASES:
        name = case["name"]
        tr = golden_cases.track_for(case, out)
        cfgd = dict(base)
        cfgd.update({k: str(v) for k, v in case["overrides"].items()})
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            for k, v in cfgd.items():
                f.write("%s = %s\n" % (k, v))
            cfg_path = f.name
        try:
            ref, info = oracle.ref_synthesize(tr, case["model"], case["rate"], case["crate"], config=cfg_path)
        finally:
            os.unlink(cfg_path)
        entry = dict(n=int(ref.size), steps=int(info["steps"]), fs=float(info["fs"]),
                     sum=float(ref.astype(np.float64).sum()), maxabs=float(np.abs(ref).max()) if ref.size else 0.0,
                     sha256=hashlib.sha256(ref.tobytes()).hexdigest())
        if case["store"] == "full":
            out[name + "__out"] = ref
        else:
            out[name + "__strided"] = ref[:: golden_cases.DIGEST_STRIDE].copy()
        manifest[name] = entry
        print(name, entry["n"], entry["sha256"][:12])
    out["manifest_json"] = np.frombuffer(json.dumps(manifest, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "vtm_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "vtm_golden.npz"))


if __name__ == "__main__":
    main()
