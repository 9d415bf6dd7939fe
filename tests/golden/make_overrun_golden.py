#!/usr/bin/env python3
"""Generates tests/golden/vtm_overrun_golden.npz by running the REAL reference at utterance lengths where its
SampleRateConverter runs into the flush overrun (golden_cases.OVERRUN_CASES).

Build-container only: executes oracle/_ref/ref_vtm (compiled in place from /root/reference by oracle/Makefile with
-O2 -ffp-contract=off).  The .npz holds data only: counts, digests, strided subsets and the tails of the reference output.

    python tests/golden/make_overrun_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_cases  # noqa: E402
import oracle  # noqa: E402


def main():
    out, manifest = {}, {}
    for case in golden_cases.OVERRUN_CASES:
        name = case["name"]
        tr = golden_cases.track_for(case)
        voice = oracle.VOICE5_MALE if case["model5"] else oracle.VOICE_MALE
        ref, info = oracle.ref_synthesize(tr, case["model"], case["rate"], case["crate"], config=voice)
        manifest[name] = dict(n=int(ref.size), steps=int(info["steps"]), fs=float(info["fs"]),
                              sum=float(ref.astype(np.float64).sum()), maxabs=float(np.abs(ref).max()),
                              sha256=hashlib.sha256(ref.tobytes()).hexdigest())
        if case["store"] == "full":
            out[name + "__out"] = ref
        else:
            out[name + "__strided"] = ref[:: golden_cases.DIGEST_STRIDE].copy()
            out[name + "__tail"] = ref[-golden_cases.OVERRUN_TAIL:].copy()
        print(name, ref.size, manifest[name]["sha256"][:12])
    out["manifest_json"] = np.frombuffer(json.dumps(manifest, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "vtm_overrun_golden.npz"), **out)


if __name__ == "__main__":
    main()
