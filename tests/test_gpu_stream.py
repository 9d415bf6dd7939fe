"""Streams (gvtm_stream_*): utterances handed over in pieces must come out exactly as from the one-shot entry.

What the reference keeps between execSynthesisStep() calls (vtm/VocalTractModel0.h:221-252, :396-445) lives in device
memory between launches; reset() (:309-326) and finishSynthesis() (:720-723) have their counterparts."""
import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import oracle
import tracks

pytestmark = pytest.mark.gpu


def _plan(precision, delay=1, rate=44100.0, layout=0, crate=250.0, rows=0):
    d = g.read_config_file(oracle.VOICE_MALE)
    return g.Plan(g.config_from_dict(d, rate, delay, precision, layout), crate, 0, diagnostics=bool(rows), rows=rows)


def _stream_one(plan, track, blocks):
    st = g.Stream(plan, 1)
    out = []
    at = 0
    for n in blocks:
        if at >= track.shape[0]:
            break
        piece = track[None, at: at + n]
        at += piece.shape[1]
        out.append(st.push(piece)[0])
    if at < track.shape[0]:
        out.append(st.push(track[None, at:])[0])
    tail, maxabs = st.finish()
    out.append(tail[0])
    return np.concatenate(out), maxabs[0], st


@pytest.mark.parametrize("precision", [capi.PRECISION_F32, capi.PRECISION_F64, capi.PRECISION_MIXED], ids=["f32", "f64", "mixed"])
@pytest.mark.parametrize("delay,rate,layout", [(1, 44100.0, 0), (2, 44100.0, 0), (3, 44100.0, 0), (1, 22050.0, 1)], ids=["vtm0", "d2", "model3", "model4_22k"])
def test_uneven_blocks_equal_one_shot_bit_for_bit(precision, delay, rate, layout):
    frames = 83 if rate == 22050.0 else 97   # 83 frames at 22.05 kHz: the converter's flush overrun at finish
    track = tracks.random_track(frames, 5100 + delay, True)
    plan = _plan(precision, delay, rate, layout)
    whole, counts, peak = plan.synthesize_host(track[None])
    got, maxabs, _ = _stream_one(plan, track, [1, 7, 3, 1, 1, 40, 2, 13, 5, 1, 9])
    assert got.size == counts[0]
    assert np.array_equal(got, whole[0, : counts[0]])
    assert maxabs == peak[0]
    if precision == capi.PRECISION_F32:
        ref = oracle.synthesize(oracle.male_config(rate, delay, layout, float_model=1), track)
        assert np.array_equal(got, ref)   # and therefore bit-identical to the reference's float class
    elif precision == capi.PRECISION_F64:
        ref = oracle.synthesize(oracle.male_config(rate, delay, layout), track)
        assert np.abs(got.astype(np.float64) - ref).max() <= 1e-9 * np.abs(ref).max() + np.spacing(np.float32(np.abs(ref).max()))


def test_single_frame_pushes_and_reset():
    """One frame at a time (the converter, the decimator and every recurrence cross a launch boundary again and
    again), then the same stream object reused after reset()."""
    plan = _plan(capi.PRECISION_F32)
    a, b = tracks.random_track(30, 61, True), tracks.random_track(17, 62, False)
    ref_a = oracle.synthesize(oracle.male_config(float_model=1), a)
    ref_b = oracle.synthesize(oracle.male_config(float_model=1), b)
    got, _, st = _stream_one(plan, a, [1] * 30)
    assert np.array_equal(got, ref_a)
    with pytest.raises(g.GvtmError):
        st.push(a[None, :1])          # finished: reset first
    st.reset()
    out = [st.push(b[None, i: i + 1])[0] for i in range(17)]
    tail, _ = st.finish()
    assert np.array_equal(np.concatenate(out + [tail[0]]), ref_b)


def test_empty_and_tiny_utterances():
    plan = _plan(capi.PRECISION_F64)
    st = g.Stream(plan, 1)
    tail, maxabs = st.finish()                       # nothing pushed: the flush of an empty utterance
    assert tail[0].size == plan.output_count(0) and not tail[0].any() and maxabs[0] == 0.0
    st.reset()
    one = tracks.random_track(1, 63, True)
    assert st.push(one[None])[0].size == 0           # a lone frame waits for its successor
    tail, _ = st.finish()
    whole, counts, _ = plan.synthesize_host(one[None])
    assert np.array_equal(tail[0], whole[0, : counts[0]])


@pytest.mark.parametrize("rows", [0, 2, 4])
def test_batch_stream_ragged_pushes(rows):
    """Five utterances pushed with different frame counts each time (one workgroup per utterance), and in lockstep
    (shared workgroups, the shape forced through the diagnostics library)."""
    plan = _plan(capi.PRECISION_F32, delay=2, rows=rows)
    total = np.array([50, 0, 37, 50, 12], dtype=np.int32) if rows == 0 else np.array([48] * 5, dtype=np.int32)
    params = tracks.random_tracks(5, 50, seed0=6400, consonant_heavy=True)
    cfg = oracle.male_config(44100.0, 2, float_model=1)
    st = g.Stream(plan, 5)
    got = [[] for _ in range(5)]
    at = np.zeros(5, dtype=np.int64)
    rng = np.random.default_rng(5)
    while (at < total).any():
        if rows == 0:
            n = np.minimum(rng.integers(0, 9, size=5), total - at).astype(np.int32)
        else:
            n = np.full(5, min(int(rng.integers(1, 9)), int(total[0] - at[0])), dtype=np.int32)
        width = max(int(n.max()), 1)
        block = np.zeros((5, width, 16), dtype=np.float32)
        for b in range(5):
            block[b, : n[b]] = params[b, at[b]: at[b] + n[b]]
        for b, piece in enumerate(st.push(block, n)):
            got[b].append(piece)
        at += n
    tails, maxabs = st.finish()
    for b in range(5):
        ref = oracle.synthesize(cfg, params[b, : total[b]]) if total[b] else np.zeros(oracle.output_count(cfg, 0), np.float32)
        out = np.concatenate(got[b] + [tails[b]])
        assert np.array_equal(out, ref), b
        assert maxabs[b] == np.abs(ref).max()


def test_long_track_in_one_second_pieces():
    """The streaming use the one-shot entry cannot serve in bounded memory: a 30 s utterance, one second at a time."""
    plan = _plan(capi.PRECISION_F32)
    track = tracks.random_track(7500, 6500)
    whole, counts, _ = plan.synthesize_host(track[None])
    got, _, _ = _stream_one(plan, track, [250] * 30)
    assert got.size == counts[0] == 1320815
    assert np.array_equal(got, whole[0])


def test_stream_refusals():
    none = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE)), 250.0, capi.DEVICE_NONE)
    with pytest.raises(g.GvtmError) as ei:
        g.Stream(none, 1)
    assert ei.value.status == 2  # GVTM_ERR_NO_DEVICE


# ---- reference model 5 as a stateful object (vtm/VocalTractModel5.h:523-579) -------------------------------------------

def _plan5(rate=48000.0, crate=250.0, rows=0):
    d = g.read_config_file(oracle.VOICE5_MALE)
    return g.Plan(g.config5_from_dict(d, rate), crate, 0, diagnostics=bool(rows), rows=rows)


@pytest.mark.parametrize("rate,frames", [(48000.0, 97), (44100.0, 106), (96000.0, 41)], ids=["48k", "44k_overrun106", "96k_up"])
def test_model5_uneven_blocks_equal_one_shot_bit_for_bit(rate, frames):
    """Pieces of 1..40 frames (every recurrence, the feed-forward halves' predecessors, the converter's ring and the
    difference filter's look-back cross launch boundaries at arbitrary places); 106 frames at 44.1 kHz finishes on a flush
    overrun of the converter."""
    track = tracks.random_track(frames, 5500, True)
    plan = _plan5(rate)
    whole, counts, peak = plan.synthesize_host(track[None])
    got, maxabs, _ = _stream_one(plan, track, [1, 7, 3, 1, 1, 40, 2, 13, 5, 1, 9])
    assert got.size == counts[0]
    assert np.array_equal(got, whole[0, : counts[0]])
    assert maxabs == peak[0]
    ref, _ = oracle.synthesize5(oracle.male5_config(rate), track)
    assert ref.size == got.size
    assert np.abs(got.astype(np.float64) - ref).max() <= 2e-6 * np.abs(ref).max()


def test_model5_single_frame_pushes_reset_and_ragged_batch():
    plan = _plan5()
    track = tracks.random_track(23, 5600, True)
    whole, counts, _ = plan.synthesize_host(track[None])
    got, _, st = _stream_one(plan, track, [1] * 23)
    assert np.array_equal(got, whole[0, : counts[0]])
    st.reset()
    other = tracks.random_track(9, 5601, True)
    whole2, counts2, _ = plan.synthesize_host(other[None])
    pieces = st.push(other[None, :4])[0], st.push(other[None, 4:])[0]
    tail, _ = st.finish()
    assert np.array_equal(np.concatenate([pieces[0], pieces[1], tail[0]]), whole2[0, : counts2[0]])
    # a batch stream whose utterances get different numbers of frames per push
    batch = tracks.random_tracks(3, 30, seed0=5700, consonant_heavy=True)
    total = np.array([30, 11, 22], dtype=np.int32)
    one, c1, _ = plan.synthesize_host(batch, total)
    st3 = g.Stream(plan, 3)
    outs = [[], [], []]
    done = np.zeros(3, dtype=np.int32)
    for n in (5, 9, 16):
        fc = np.minimum(n, total - done).astype(np.int32)
        buf = np.zeros((3, n, 16), np.float32)
        for b in range(3):
            buf[b, : fc[b]] = batch[b, done[b]: done[b] + fc[b]]
        res = st3.push(buf, fc)
        for b in range(3):
            outs[b].append(res[b])
        done += fc
    tails, _ = st3.finish()
    for b in range(3):
        whole_b = np.concatenate(outs[b] + [tails[b]])
        assert whole_b.size == c1[b] and np.array_equal(whole_b, one[b, : c1[b]]), b
