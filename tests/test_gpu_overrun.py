"""The reference converter's flush overrun on the device (SampleRateConverter.h:298-308 with :462-471).

When down-sampling, ~0.4 % of the utterance lengths make flushBuffer()'s final dataEmpty() convert one more lap of the
1024-sample ring from its leftovers.  The device reproduces those samples (kernel epilogues in vtm_kernel_v2.inc /
vtm_kernel_m5.inc); the vectors come from the REAL reference (tests/golden/make_overrun_golden.py)."""
import hashlib

import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import golden_cases
import oracle
import tracks

pytestmark = pytest.mark.gpu


def _plan(case, precision, rows=0):
    if case["model5"]:
        return g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE), case["rate"]), case["crate"], 0)
    d = g.read_config_file(oracle.VOICE_MALE)
    return g.Plan(g.config_from_dict(d, case["rate"], case["delay"], precision, case.get("layout", 0)), case["crate"], 0,
                  diagnostics=bool(rows), rows=rows)


def _check(case, out, m, z, tol):
    if case["store"] == "full":
        pieces = [(out, z[case["name"] + "__out"])]
    else:
        pieces = [(out[:: golden_cases.DIGEST_STRIDE], z[case["name"] + "__strided"]),
                  (out[-golden_cases.OVERRUN_TAIL:], z[case["name"] + "__tail"])]
    for got, ref in pieces:
        if tol == 0.0:
            assert np.array_equal(got, ref)
        else:
            d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
            ulp = np.spacing(np.abs(ref)).astype(np.float64)
            assert (d <= np.maximum(ulp, tol * m["maxabs"])).all(), float(d.max() / m["maxabs"])


@pytest.mark.parametrize("case", golden_cases.OVERRUN_CASES, ids=lambda c: c["name"])
def test_overrun_vectors(case, golden_overrun):
    m = golden_overrun["manifest"][case["name"]]
    tr = golden_cases.track_for(case)
    if case["float_model"]:
        runs = [(capi.PRECISION_F32, 0.0)]           # the float classes: bit-identical
    elif case["model5"]:
        runs = [(capi.PRECISION_F64, 2e-6)]          # model 5's differentiated output (tests/test_gpu_model5.py)
    else:
        # mixed = fp32 resampler: its error grows with the taps per output (26 when up-sampling, ~70-140 at these ratios)
        runs = [(capi.PRECISION_F64, 1e-9), (capi.PRECISION_MIXED, 1e-6)]
    for precision, tol in runs:
        plan = _plan(case, precision)
        assert plan.output_count(tr.shape[0]) == m["n"]
        audio, counts, maxabs = plan.synthesize_host(tr[None])
        assert counts[0] == m["n"] and audio.shape[1] == m["n"]
        out = audio[0]
        if tol == 0.0:
            assert hashlib.sha256(out.tobytes()).hexdigest() == m["sha256"]
        _check(case, out, m, golden_overrun, tol)
        assert maxabs[0] == np.abs(out).max()


@pytest.mark.parametrize("rows", [1, 2, 4])
def test_one_overrun_utterance_in_a_ragged_batch(rows):
    """A batch in which SOME utterances hit the overrun: every row gets its own count and samples (the host entry
    used to refuse the whole batch), whatever the workgroup shape; rows are zero beyond their count."""
    delay, rate = 2, 22050.0   # overruns at 18 and 79 frames (tests/soak_cases.overrun_lengths)
    frames = np.array([79, 40, 18, 80, 17, 79, 0, 19, 78], dtype=np.int32)
    params = tracks.random_tracks(len(frames), 80, seed0=4400, consonant_heavy=True)
    cfg = oracle.male_config(rate, delay, float_model=1)
    d = g.read_config_file(oracle.VOICE_MALE)
    plan = g.Plan(g.config_from_dict(d, rate, delay, capi.PRECISION_F32), 250.0, 0, diagnostics=True, rows=rows)
    longest = max(plan.output_count(int(f)) for f in frames)
    assert plan.output_count(79) > plan.output_count(80)      # the overrun makes the shorter utterance longer
    assert plan.output_capacity(80) >= longest
    audio, counts, maxabs = plan.synthesize_host(params, frames)
    assert audio.shape[1] == plan.output_capacity(80)
    for b, f in enumerate(frames):
        ref = oracle.synthesize(cfg, params[b, :f]) if f else np.zeros(oracle.output_count(cfg, 0), np.float32)
        assert counts[b] == ref.size == plan.output_count(int(f)), (b, f)
        assert np.array_equal(audio[b, : ref.size], ref), (b, f)
        assert not audio[b, ref.size:].any()
        assert maxabs[b] == np.abs(ref).max()


def test_device_resident_frame_counts_with_overrun():
    """gvtm_synthesize_batch_device with d_frame_counts: nothing can be validated on the host, the kernel decides per
    utterance; a stride of gvtm_output_count(max_frames) drops what does not fit and still reports the full count."""
    import torch
    delay, rate = 2, 22050.0
    frames = np.array([79, 80, 18, 60], dtype=np.int32)
    params = tracks.random_tracks(4, 80, seed0=4500, consonant_heavy=True)
    d = g.read_config_file(oracle.VOICE_MALE)
    plan = g.Plan(g.config_from_dict(d, rate, delay, capi.PRECISION_F64), 250.0, 0)
    cfg = oracle.male_config(rate, delay)
    dev = torch.device("cuda:0")
    dp, dfc = torch.from_numpy(params).to(dev), torch.from_numpy(frames).to(dev)
    for stride in (plan.output_capacity(80), plan.output_count(80)):
        da = torch.zeros((4, stride), dtype=torch.float32, device=dev)
        dc = torch.zeros(4, dtype=torch.int64, device=dev)
        plan.synthesize_device(dp, 4, 80, da, stride, dfc, dc, None)
        torch.cuda.synchronize()
        a, c = da.cpu().numpy(), dc.cpu().numpy()
        for b, f in enumerate(frames):
            ref = oracle.synthesize(cfg, params[b, :f])
            assert c[b] == ref.size
            n = min(ref.size, stride)
            err = np.abs(a[b, :n].astype(np.float64) - ref[:n]).max() / np.abs(ref).max()
            assert err < 2e-7, (b, err)
