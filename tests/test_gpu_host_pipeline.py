"""The host-buffer entries (include/gama_vtm.h: gvtm_synthesize_batch_host, gvtm_synthesize_batch_host_pcm16): frames in
host memory -> samples in host memory, the second one ending where the reference's file writer ends
(Controller::writeOutputToFile, vtm_control_model/Controller.cpp:325-340; WAVEFileWriter::writeSample,
WAVEFileWriter.cpp:122-125).  Big batches go through a three-stream pipeline of slices (H2D || kernel + scaling || D2H):
what comes back must not depend on the slicing, on the buffers being page-locked, or on the entry used."""
import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import oracle
import tracks

pytestmark = pytest.mark.gpu


def _plan(precision, delay=1, rate=44100.0, layout=0):
    return g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), rate, delay, precision, layout), 250.0, 0)


def _pcm_rule(x, scale):
    """writeSample: round(x * scale * 32767) in float32, std::round (half away from zero)."""
    p = (x * np.float32(scale)) * np.float32(32767.0)
    return (np.sign(p) * np.floor(np.abs(p) + np.float32(0.5))).astype(np.int16)


def _ragged(batch, max_frames, seed):
    pool = 24
    rng = np.random.default_rng(seed)
    pf = rng.integers(0, max_frames + 1, size=pool).astype(np.int32)
    pf[:3] = [max_frames, 0, 1]
    pp = tracks.random_tracks(pool, max_frames, seed0=seed, consonant_heavy=True)
    idx = np.arange(batch) % pool
    return pp[idx], pf[idx]


@pytest.mark.parametrize("batch", [300, 2100], ids=["one_launch", "three_slices"])
@pytest.mark.parametrize("precision", [capi.PRECISION_F32, capi.PRECISION_F64], ids=["f32", "f64"])
def test_pcm16_entry_is_the_float_entry_scaled_and_rounded(batch, precision):
    """2100 utterances = four per workgroup x 256 compute units twice over: three slices through the pipeline."""
    params, frames = _ragged(batch, 14, 99 + batch)
    plan = _plan(precision)
    audio, counts, maxabs = plan.synthesize_host(params, frames)
    pcm, counts16, maxabs16, scales = plan.synthesize_host_pcm16(params, frames)
    assert np.array_equal(counts, counts16) and np.array_equal(maxabs, maxabs16)
    for b in range(batch):
        n = int(counts[b])
        assert maxabs[b] == (np.abs(audio[b, :n]).max() if n else 0.0)
        want_scale = np.float32(oracle.output_scale(audio[b, :n])) if n else np.float32(0.0)  # (restated rule, pinned to reference-written WAVs)
        assert np.float32(scales[b]) == want_scale, b
        assert np.array_equal(pcm[b, :n], _pcm_rule(audio[b, :n], scales[b])), b
        assert not pcm[b, n:].any() and not audio[b, n:].any()
    # the device entry in ONE launch gives the pipeline's samples bit for bit
    import torch
    dp = torch.from_numpy(params).cuda()
    df = torch.from_numpy(frames).cuda()
    stride = audio.shape[1]
    da = torch.zeros((batch, stride), dtype=torch.float32, device="cuda")
    dc = torch.zeros(batch, dtype=torch.int64, device="cuda")
    plan.synthesize_device(dp, batch, params.shape[1], da, stride, df, dc, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy(), counts)
    got = da.cpu().numpy()
    for b in range(batch):
        assert np.array_equal(got[b, : counts[b]], audio[b, : counts[b]]), b


def test_page_locked_buffers_and_bad_frame_counts():
    """The same call with page-locked buffers from gvtm_host_alloc (where the three streams really overlap) and with a frame
    count outside [0, max_frames]: that utterance fails alone (count -1, zero row), the others are untouched."""
    batch, max_frames = 2060, 10
    params, frames = _ragged(batch, max_frames, 4242)
    plan = _plan(capi.PRECISION_F32)
    pcm_ref, counts_ref, _, scales_ref = plan.synthesize_host_pcm16(params, frames)
    stride = pcm_ref.shape[1]
    p_in = g.PinnedArray(params.shape, np.float32)
    p_out = g.PinnedArray((batch, stride), np.int16)
    p_in.array[...] = params
    p_out.array[...] = 12345
    fr = frames.copy()
    fr[7] = max_frames + 1
    fr[1500] = -3
    counts = np.zeros(batch, np.int64)
    scales = np.zeros(batch, np.float32)
    plan.synthesize_host_into(p_in.array, p_out.array, fr, counts, None, scales)
    for b in range(batch):
        if b in (7, 1500):
            assert counts[b] == -1 and scales[b] == 0.0 and not p_out.array[b].any()
        else:
            assert counts[b] == counts_ref[b] and scales[b] == scales_ref[b]
            assert np.array_equal(p_out.array[b], pcm_ref[b]), b
    # float entry, page-locked too
    f_out = g.PinnedArray((batch, stride), np.float32)
    plan.synthesize_host_into(p_in.array, f_out.array, frames, counts, None)
    audio, counts2, _ = plan.synthesize_host(params, frames)
    assert np.array_equal(counts, counts2) and np.array_equal(f_out.array, audio)
    p_in.close(); p_out.close(); f_out.close()


def test_pcm16_against_the_float_oracle_end_to_end():
    """Float model: the 16-bit samples are exactly what the reference's float class + its WAV writer produce (oracle float32
    samples are bit-identical to VocalTractModel0<float>'s; the scaling rule is pinned to reference-written WAV files in
    tests/test_oracle_vs_golden.py)."""
    params = tracks.random_tracks(6, 30, seed0=31, consonant_heavy=True)
    plan = _plan(capi.PRECISION_F32)
    pcm, counts, _, _ = plan.synthesize_host_pcm16(params)
    cfg = oracle.male_config(44100.0, 1, float_model=1)
    for b in range(6):
        ref = oracle.synthesize(cfg, params[b])
        assert counts[b] == ref.size
        assert np.array_equal(pcm[b], _pcm_rule(ref, oracle.output_scale(ref)))
