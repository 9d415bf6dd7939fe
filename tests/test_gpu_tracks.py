"""Track generation on the device (gvtm_generate_tracks_*) against the reference fixtures and the oracle:
bit-identical frames, frame counts and drift-generator states; and events -> frames -> audio with the frames
never leaving the device."""
import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import event_lists
import oracle
from test_tracks_cpu import TEXTS, _product_config

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", TEXTS)
def test_device_matches_captured_reference_calls(name, golden_tracks):
    state = [oracle.FRESH_DRIFT]
    for call in range(6):
        cfg, events, frames = event_lists.load_golden(golden_tracks, name, call)
        params, counts, drift = capi.generate_tracks_host(_product_config(cfg), [capi.events_from_table(events)], frames.shape[0] + 3, drift=state)
        assert counts[0] == frames.shape[0]
        assert np.array_equal(params[0, : frames.shape[0]].view(np.uint32), frames.view(np.uint32)), (name, call)
        assert not params[0, frames.shape[0]:].any()
        state = [tuple(drift[0])]
        _, want_state = oracle.tracks_generate(oracle.track_config(cfg), events, oracle.FRESH_DRIFT) if call == 0 else (None, None)
        if call == 0:
            assert state[0] == want_state


@pytest.mark.parametrize("flags", [(1, 1, 1, 1), (0, 1, 0, 1), (1, 0, 1, 0), (0, 0, 0, 0)])
def test_ragged_batch_against_oracle(flags):
    macro, micro, drift, smooth = flags
    cfg = np.array([4, macro, micro, drift, smooth, -20.0, -6.0, 4.0, 250.0, 4.0])
    tables = [event_lists.random_event_table(100 + b, n_events=int(n)) for b, n in enumerate([40, 2, 1, 17, 80, 3, 55, 9, 33])]
    want = [oracle.tracks_generate(oracle.track_config(cfg), t) for t in tables]
    max_frames = max(w[0].shape[0] for w in want)
    params, counts, dr = capi.generate_tracks_host(_product_config(cfg), [capi.events_from_table(t) for t in tables], max_frames,
                                                   drift=[oracle.FRESH_DRIFT] * len(tables))
    for b, (frames, state) in enumerate(want):
        assert counts[b] == frames.shape[0]
        assert np.array_equal(params[b, : frames.shape[0]].view(np.uint32), frames.view(np.uint32)), b
        assert tuple(dr[b]) == state


def test_truncation_and_fresh_generator_default():
    cfg = np.array([4, 1, 1, 1, 1, -20.0, -6.0, 4.0, 250.0, 4.0])
    table = event_lists.random_event_table(7, n_events=30)
    frames, _ = oracle.tracks_generate(oracle.track_config(cfg), table)
    params, counts, _ = capi.generate_tracks_host(_product_config(cfg), [capi.events_from_table(table)], 50)  # drift=None: fresh generator
    assert counts[0] == frames.shape[0] > 50
    assert np.array_equal(params[0].view(np.uint32), frames[:50].view(np.uint32))


def test_events_to_audio_on_the_device(golden_tracks):
    """Event lists -> parameter frames -> audio, the frames produced and consumed in device memory: equals the
    oracle of the oracle (EventList::generateOutput then the vocal-tract model)."""
    import torch
    names = ["hello", "question", "count", "hello"]
    cfgv, _, _ = event_lists.load_golden(golden_tracks, "hello", 0)
    tables = [event_lists.load_golden(golden_tracks, n, 0)[1] for n in names]
    want_frames = [oracle.tracks_generate(oracle.track_config(cfgv), t)[0] for t in tables]
    max_frames = max(f.shape[0] for f in want_frames)
    dev = torch.device("cuda:0")
    evs = [capi.events_from_table(t) for t in tables]
    offsets = np.zeros(len(evs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    d_events = torch.from_numpy(np.concatenate(evs).view(np.uint8)).to(dev)
    d_offsets = torch.from_numpy(offsets).to(dev)
    d_params = torch.zeros((len(evs), max_frames, 16), dtype=torch.float32, device=dev)
    d_counts = torch.zeros(len(evs), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    capi.generate_tracks_device(_product_config(cfgv), d_events, d_offsets, len(evs), max_frames, d_params, d_counts, None, stream)
    plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 1), 250.0, 0)
    n_out = plan.output_count(max_frames)
    d_audio = torch.zeros((len(evs), n_out), dtype=torch.float32, device=dev)
    d_n = torch.zeros(len(evs), dtype=torch.int64, device=dev)
    plan.synthesize_device(d_params, len(evs), max_frames, d_audio, n_out, d_counts, d_n, None, stream)
    torch.cuda.synchronize()
    audio = d_audio.cpu().numpy()
    cfg = oracle.male_config()
    for b, frames in enumerate(want_frames):
        assert d_counts[b].item() == frames.shape[0]
        ref = oracle.synthesize(cfg, frames)
        assert d_n[b].item() == ref.size
        from test_gpu_parity import _within, _peak_err
        assert _within(audio[b, : ref.size], ref, 1e-9), (b, _peak_err(audio[b, : ref.size], ref))


def test_events_to_audio_on_the_device_model5(golden_tracks):
    """The same chain into reference model 5 (the event list and its frames do not depend on the vocal-tract model):
    frames made on the device feed the model-5 kernel without leaving HBM."""
    import torch
    from test_gpu_model5 import _check
    names = ["question", "hello", "count"]
    cfgv, _, _ = event_lists.load_golden(golden_tracks, "hello", 0)
    tables = [event_lists.load_golden(golden_tracks, n, 0)[1] for n in names]
    want_frames = [oracle.tracks_generate(oracle.track_config(cfgv), t)[0] for t in tables]
    max_frames = max(f.shape[0] for f in want_frames)
    dev = torch.device("cuda:0")
    evs = [capi.events_from_table(t) for t in tables]
    offsets = np.zeros(len(evs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    d_events = torch.from_numpy(np.concatenate(evs).view(np.uint8)).to(dev)
    d_offsets = torch.from_numpy(offsets).to(dev)
    d_params = torch.zeros((len(evs), max_frames, 16), dtype=torch.float32, device=dev)
    d_counts = torch.zeros(len(evs), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    capi.generate_tracks_device(_product_config(cfgv), d_events, d_offsets, len(evs), max_frames, d_params, d_counts, None, stream)
    plan = g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE)), 250.0, 0)
    n_out = plan.output_count(max_frames)
    d_audio = torch.zeros((len(evs), n_out), dtype=torch.float32, device=dev)
    d_n = torch.zeros(len(evs), dtype=torch.int64, device=dev)
    plan.synthesize_device(d_params, len(evs), max_frames, d_audio, n_out, d_counts, d_n, None, stream)
    torch.cuda.synchronize()
    audio = d_audio.cpu().numpy()
    cfg = oracle.male5_config(48000.0)
    for b, frames in enumerate(want_frames):
        ref, _ = oracle.synthesize5(cfg, frames)
        assert d_n[b].item() == ref.size
        _check(audio[b, : ref.size], ref)
