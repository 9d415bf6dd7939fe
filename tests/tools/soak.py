"""GPU-box soak: ragged random batches through every model / precision against the oracle -> JSON on stdout.
usage: python tests/tools/soak.py [batch=768] [max_frames=96]      (the reduced -m gpu form: tests/test_gpu_soak.py)"""
import json
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import soak_cases  # noqa: E402

BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 768
MAXF = int(sys.argv[2]) if len(sys.argv) > 2 else 96
out = {"workload": "%d utterances of 0..%d frames (ragged, one launch per case), random + consonant-heavy tracks; up to six of them "
                   "at lengths that trigger the reference converter's flush overrun" % (BATCH, MAXF)}
out.update(soak_cases.run(BATCH, MAXF, log=lambda name, s: print(name, s, file=sys.stderr, flush=True)))
print(json.dumps(out))
