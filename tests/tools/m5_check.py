#!/usr/bin/env python3
"""Device model 5 against the oracle: error statistics per utterance (development aid, GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gama_tts_amd as g  # noqa: E402
import oracle  # noqa: E402
import tracks  # noqa: E402


def main():
    d = g.read_config_file(oracle.VOICE5_MALE)
    for name, ov in (("default", {}), ("bypass", {"bypass": "1"}), ("sine", {"waveform": "1"}),
                     ("tn", {"glottal_pulse_tn_min": "16.0", "glottal_pulse_tn_max": "32.0"})):
        dd = dict(d, **ov)
        plan = g.Plan(g.config5_from_dict(dd), 250.0, 0)
        params = tracks.random_tracks(4, 60, seed0=5, consonant_heavy=True)
        audio, counts, maxabs = plan.synthesize_host(params)
        cfg = oracle.config5_from_dict(dd)
        for b in range(params.shape[0]):
            ref, _ = oracle.synthesize5(cfg, params[b])
            got = audio[b, : ref.size]
            dlt = np.abs(got.astype(np.float64) - ref)
            peak = np.abs(ref).max()
            first = int(np.argmax(dlt > 0)) if dlt.max() > 0 else -1
            print("%-8s utt %d n %d/%d peak %.4g maxerr/peak %.3g exact %.4f first-diff %d" % (
                name, b, counts[b], ref.size, peak, dlt.max() / max(peak, 1e-300), float((got == ref).mean()), first), flush=True)


if __name__ == "__main__":
    main()
