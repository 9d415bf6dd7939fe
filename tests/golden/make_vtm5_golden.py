#!/usr/bin/env python3
"""Generates tests/golden/vtm5_golden.npz by running the REAL reference's VocalTractModel5.

Build-container only: executes oracle/_ref/ref_vtm (compiled in place from /root/reference by
oracle/Makefile with -O2 -ffp-contract=off).  The .npz holds data only: reference output samples
(or their digest), counts, the internal rate.  Input frames are the recipes of tests/golden5_cases.py
(the "hello" frames are the ones stored in vtm_golden.npz).

    python tests/golden/make_vtm5_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden5_cases  # noqa: E402
import oracle  # noqa: E402


def main():
    hello = {"hello_params": np.load(os.path.join(HERE, "vtm_golden.npz"), allow_pickle=False)["hello_params"]}
    base = oracle.read_config_file(oracle.VOICE5_MALE)
    out, manifest = {}, {}
    for case in golden5_cases.CASES:
        name = case["name"]
        tr = golden5_cases.track_for(case, hello)
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
        manifest[name] = dict(n=int(ref.size), steps=int(info["steps"]), fs=float(info["fs"]),
                              sum=float(ref.astype(np.float64).sum()), maxabs=float(np.abs(ref).max()) if ref.size else 0.0,
                              sha256=hashlib.sha256(ref.tobytes()).hexdigest())
        if case["store"] == "full":
            out[name + "__out"] = ref
        else:
            out[name + "__strided"] = ref[:: golden5_cases.DIGEST_STRIDE].copy()
        print(name, ref.size, manifest[name]["sha256"][:12])
    out["manifest_json"] = np.frombuffer(json.dumps(manifest, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "vtm5_golden.npz"), **out)


if __name__ == "__main__":
    main()
