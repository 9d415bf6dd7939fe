"""Parity of the device path for reference model 5 (VocalTractModel5<double,1>) with the reference vectors
(tests/golden/vtm5_golden.npz) and the oracle, through the C ABI.

Tolerance.  Both sides compute in fp64 and differ at the 1e-13 level (device libm, fused multiply-adds); the
resampler's output is rounded to float32 and THEN differenced and multiplied by the output rate
(VocalTractModel5.h:507-513), so a float32 sample whose rounding flips (one ulp, 6e-8 of the sample) turns into
an output error of ulp * output_rate, a few 1e-7 of the output's peak.  The bar: every sample within
TOL = 2e-6 of the utterance's peak, at least 90 % of the samples bit-identical, sample counts exact.  In bypass
mode (no difference filter) the bar is the fp64 path's of the other models: 1e-9 of peak or one float32 ulp.
"""
import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import golden5_cases
import oracle
import tracks

pytestmark = pytest.mark.gpu

TOL = 2e-6
MIN_EXACT = 0.90


def _plan(overrides=None, rate=48000.0, crate=250.0):
    d = g.read_config_file(oracle.VOICE5_MALE)
    d.update({k: str(v) for k, v in (overrides or {}).items()})
    return g.Plan(g.config5_from_dict(d, rate), crate, 0)


def _check(got, ref, bypass=False, peak=None):
    assert got.shape == ref.shape
    ref64 = ref.astype(np.float64)
    peak = float(np.abs(ref64).max()) if peak is None else float(peak)
    d = np.abs(got.astype(np.float64) - ref64)
    if bypass:
        ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        assert (d <= np.maximum(ulp, 1e-9 * peak)).all(), float(d.max() / max(peak, 1e-300))
    else:
        assert float(d.max()) <= TOL * max(peak, 1e-300), float(d.max() / max(peak, 1e-300))
        if got.size >= 200:
            assert float((got == ref).mean()) >= MIN_EXACT, float((got == ref).mean())


DOUBLE_CASES = [c for c in golden5_cases.CASES if not c["float_model"]]


@pytest.mark.parametrize("case", DOUBLE_CASES, ids=lambda c: c["name"])
def test_reference_vectors(case, golden, golden5):
    m = golden5["manifest"][case["name"]]
    tr = golden5_cases.track_for(case, golden)
    plan = _plan(case["overrides"], case["rate"], case["crate"])
    assert plan.info.model5 == 1
    assert abs(plan.info.internal_rate_hz - m["fs"]) < 1e-6
    assert plan.info.control_steps * tr.shape[0] == m["steps"]
    assert plan.output_count(tr.shape[0]) == m["n"]
    audio, counts, maxabs = plan.synthesize_host(tr[None])
    assert counts[0] == m["n"]
    out = audio[0]
    bypass = int(case["overrides"].get("bypass", 0)) == 1
    if case["store"] == "full":
        _check(out, golden5[case["name"] + "__out"], bypass, m["maxabs"])
    else:
        _check(out[:: golden5_cases.DIGEST_STRIDE], golden5[case["name"] + "__strided"], bypass, m["maxabs"])
    assert maxabs[0] == pytest.approx(m["maxabs"], rel=10 * TOL, abs=1e-12)
    assert maxabs[0] == np.abs(out).max()


def test_ragged_batch_against_oracle():
    # utterances of different lengths in one launch (0, 1, 2, ... frames), consonant-heavy tracks
    frames = [0, 1, 2, 3, 7, 25, 40, 40, 13, 31]
    params = tracks.random_tracks(len(frames), 40, seed0=900, consonant_heavy=True)
    plan = _plan()
    audio, counts, maxabs = plan.synthesize_host(params, frame_counts=frames)
    cfg = oracle.male5_config(48000.0)
    for b, f in enumerate(frames):
        ref, _ = oracle.synthesize5(cfg, params[b, :f])
        assert counts[b] == ref.size
        _check(audio[b, : ref.size], ref)
        assert maxabs[b] == np.abs(audio[b, : ref.size]).max()


def test_batch_of_identical_tracks_is_identical():
    one = tracks.random_track(30, 77, True)
    params = np.repeat(one[None], 70, axis=0)
    audio, _, _ = _plan().synthesize_host(params)
    assert (audio == audio[0]).all()


@pytest.mark.parametrize("rate", [96000.0, 176400.0])
def test_upsampling_branch(rate):
    # an output rate above the 60.4 kHz internal rate: the resampler's other branch (SampleRateConverter.h:320-360);
    # 176.4 kHz is close to the largest ratio the sink's ring takes (3x, refused above)
    tr = tracks.random_track(30, 5, True)
    plan = _plan(rate=rate)
    assert plan.info.upsampling == 1
    audio, counts, _ = plan.synthesize_host(tr[None])
    ref, _ = oracle.synthesize5(oracle.male5_config(rate), tr)
    assert counts[0] == ref.size
    _check(audio[0, : ref.size], ref)
    with pytest.raises(g.GvtmError):
        _plan(rate=192000.0)


def test_rejects_what_the_reference_rejects():
    d = g.read_config_file(oracle.VOICE5_MALE)
    with pytest.raises(g.GvtmError):  # PoleZeroRadiationImpedance: internal rate below 50 kHz
        g.Plan(g.config5_from_dict(dict(d, vocal_tract_length="25.0")), 250.0, 0)
    with pytest.raises(g.GvtmError):  # RosenbergBGlottalSource: tn_min > tn_max
        g.Plan(g.config5_from_dict(dict(d, glottal_pulse_tn_min="30.0")), 250.0, 0)
    with pytest.raises(g.GvtmError):  # the factory only offers VocalTractModel5<double,1>
        g.Plan(g.config5_from_dict(d, precision=capi.PRECISION_F32), 250.0, 0)


def test_long_utterance_next_to_short_ones():
    # BASELINE's long form: 7500 frames (30 s, 1.8 M internal steps, the ring and the 16.16 time register wrap many times),
    # in one launch with a one-frame and an empty utterance
    long_tr = tracks.random_track(7500, 11, True)
    params = np.zeros((3, 7500, 16), np.float32)
    params[0] = long_tr
    params[1, :1] = long_tr[:1]
    plan = _plan()
    audio, counts, maxabs = plan.synthesize_host(params, frame_counts=[7500, 1, 0])
    cfg = oracle.male5_config(48000.0)
    for b, f in enumerate((7500, 1, 0)):
        ref, _ = oracle.synthesize5(cfg, params[b, :f])
        assert counts[b] == ref.size
        _check(audio[b, : ref.size], ref)
        assert maxabs[b] == np.abs(audio[b, : ref.size]).max()


def test_full_batch_on_the_device_by_tiling():
    """4096 utterances x 500 frames resident in HBM: a pool of 64 tracks tiled 64 times, every utterance bit-identical
    to its pool twin wherever it sits in the launch; two pool members against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    batch, frames, pool_n = 4096, 500, 64
    pool = tracks.random_tracks(pool_n, frames, seed0=515100, consonant_heavy=True)
    plan = _plan()
    n = plan.output_count(frames)
    d_params = torch.from_numpy(pool).to(dev).repeat((batch // pool_n, 1, 1)).contiguous()
    d_audio = torch.empty((batch, n), dtype=torch.float32, device=dev)
    d_counts = torch.zeros(batch, dtype=torch.int64, device=dev)
    d_max = torch.zeros(batch, dtype=torch.float32, device=dev)
    plan.synthesize_device(d_params, batch, frames, d_audio, n, None, d_counts, d_max, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((d_counts == n).all())
    tiles = d_audio.view(batch // pool_n, pool_n, n)
    for t in range(1, batch // pool_n):
        assert torch.equal(tiles[t], tiles[0]), t
    assert torch.equal(d_max[:pool_n], tiles[0].abs().amax(dim=1))
    cfg = oracle.male5_config(48000.0)
    for b in (5, 60):
        ref, _ = oracle.synthesize5(cfg, pool[b])
        _check(tiles[33, b].cpu().numpy(), ref)


def test_special_case_frames():
    """tracks.edge_track (volumes 0 / 60 dB, frication at the first / last section incl. the dropped right share, radii
    at the floor, velum 0, pitch and band-pass extremes), alone and inside a batch."""
    tr = tracks.edge_track(48)
    ref, _ = oracle.synthesize5(oracle.male5_config(48000.0), tr)
    plan = _plan()
    batch = np.stack([tr, tracks.random_track(48, 3, True), tr])
    for params in (tr[None], batch):
        audio, counts, _ = plan.synthesize_host(params)
        assert counts[0] == ref.size and np.isfinite(audio).all()
        _check(audio[0, : ref.size], ref)
    assert np.array_equal(audio[2], audio[0])


def test_host_entry_slices_large_batches():
    batch, frames = 2100, 6
    pool = tracks.random_tracks(50, frames, seed0=717000, consonant_heavy=True)
    params = np.ascontiguousarray(np.tile(pool, (batch // 50, 1, 1)))
    fc = (np.arange(batch) % (frames + 1)).astype(np.int32)
    plan = _plan()
    audio, counts, _ = plan.synthesize_host(params, fc)
    cfg = oracle.male5_config(48000.0)
    for b in (0, 1023, 1024, 2047, 2048, batch - 1):
        ref, _ = oracle.synthesize5(cfg, params[b, : fc[b]])
        assert counts[b] == ref.size
        _check(audio[b, : ref.size], ref)
    # the pool repeats every 50 utterances, the frame counts every 7: 350 apart the utterances are the same
    valid = np.arange(audio.shape[1])[None, :] < counts[:350, None]
    assert np.array_equal(audio[:350][valid], audio[1050:1400][valid]) and np.array_equal(audio[:350][valid], audio[1750:2100][valid])


def _plan_rows(rows, overrides=None, rate=48000.0, crate=250.0):
    """A diagnostics plan with the utterances per workgroup forced (1: one tube wavefront, chunk 60; 2: two, chunk 24)."""
    d = g.read_config_file(oracle.VOICE5_MALE)
    d.update({k: str(v) for k, v in (overrides or {}).items()})
    return g.Plan(g.config5_from_dict(d, rate), crate, 0, diagnostics=True, rows=rows)


@pytest.mark.parametrize("case", DOUBLE_CASES, ids=lambda c: c["name"])
def test_reference_vectors_two_utterances_per_workgroup(case, golden, golden5):
    """The reference-made vectors through the two-utterance workgroup shape (vtm5_synth_kernel<24, 5, 1024, 2>: what batches
    beyond one workgroup per compute unit get), the vector in BOTH slots of a workgroup and beside a different neighbour."""
    tr = golden5_cases.track_for(case, golden)
    other = tracks.random_track(tr.shape[0], 77, True)
    plan = _plan_rows(2, case["overrides"], case["rate"], case["crate"])
    audio, counts, _ = plan.synthesize_host(np.stack([tr, other, other, tr, tr]))
    m = golden5["manifest"][case["name"]]
    bypass = int(case["overrides"].get("bypass", 0)) == 1
    for b in (0, 3, 4):
        assert counts[b] == m["n"]
        out = audio[b, : m["n"]]
        if case["store"] == "full":
            _check(out, golden5[case["name"] + "__out"], bypass, m["maxabs"])
        else:
            _check(out[:: golden5_cases.DIGEST_STRIDE], golden5[case["name"] + "__strided"], bypass, m["maxabs"])
    assert np.array_equal(audio[0], audio[3]) and np.array_equal(audio[0], audio[4])


def test_two_utterances_per_workgroup_ragged_against_the_one_utterance_shape_and_the_oracle():
    """Ragged lengths (0, 1, 2 frames, flush-overrun lengths 106 and 353, an odd batch): the two-utterance shape must give the
    one-utterance shape's samples bit for bit, and both the oracle's."""
    frames = np.array([40, 0, 1, 106, 2, 40, 17, 106, 33, 5, 40], dtype=np.int32)
    params = tracks.random_tracks(len(frames), 106, seed0=5150, consonant_heavy=True)
    a1, c1, m1 = _plan_rows(1).synthesize_host(params, frames)
    a2, c2, m2 = _plan_rows(2).synthesize_host(params, frames)
    assert np.array_equal(c1, c2) and np.array_equal(m1, m2)
    assert np.array_equal(a1, a2)
    cfg = oracle.male5_config(48000.0)
    for b in (0, 2, 3, 6, 10):
        ref, _ = oracle.synthesize5(cfg, params[b, : frames[b]])
        assert c2[b] == ref.size
        _check(a2[b, : ref.size], ref)


def test_product_library_on_a_batch_beyond_one_workgroup_per_compute_unit():
    """600 utterances (> 256 compute units) through libgama_vtm.so with nothing forced: a pool of 12 ragged tracks tiled,
    every pool member against the oracle, every copy equal to its first occurrence."""
    pool_f = np.array([30, 0, 7, 30, 19, 1, 30, 12, 25, 30, 3, 28], dtype=np.int32)
    pool = tracks.random_tracks(len(pool_f), 30, seed0=6000, consonant_heavy=True)
    batch = 601
    idx = np.arange(batch) % len(pool_f)
    plan = _plan()
    audio, counts, maxabs = plan.synthesize_host(pool[idx], pool_f[idx])
    cfg = oracle.male5_config(48000.0)
    for t in range(len(pool_f)):
        ref, _ = oracle.synthesize5(cfg, pool[t, : pool_f[t]])
        assert counts[t] == ref.size
        _check(audio[t, : ref.size], ref)
        assert maxabs[t] == (np.abs(audio[t, : ref.size]).max() if ref.size else 0.0)
    for b in range(len(pool_f), batch):
        assert counts[b] == counts[b % len(pool_f)] and np.array_equal(audio[b], audio[b % len(pool_f)]), b
