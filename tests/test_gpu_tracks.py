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
    # (110 events is the longest list the kernel stages in LDS; 111, 150 and 260 walk device memory, next to staged neighbours)
    tables = [event_lists.random_event_table(100 + b, n_events=int(n)) for b, n in enumerate([40, 2, 1, 17, 80, 3, 55, 9, 33, 110, 111, 150, 260])]
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


def _singable_event_table(seed, n_events):
    """event_lists.random_event_table with macro-intonation polynomials that keep the pitch inside the model's range
    (the generator's cubic and quadratic terms reach hundreds of semitones after half a second: fine for comparing FRAMES,
    but the oscillator then steps past its 512-entry wavetable, in the reference as here, and what comes out is whatever
    lies behind the table)."""
    t = event_lists.random_event_table(seed, n_events=n_events)
    t[:, 2] = 0.0
    t[:, 3] = 0.0
    t[:, 4] *= 0.1
    t[:, 5] *= 0.5
    return t


def _events_on_device(tables):
    import torch
    evs = [capi.events_from_table(t) for t in tables]
    offsets = np.zeros(len(evs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(e) for e in evs])
    dev = torch.device("cuda:0")
    return torch.from_numpy(np.concatenate(evs).view(np.uint8)).to(dev), torch.from_numpy(offsets).to(dev)


@pytest.mark.parametrize("batch", [5, 300, 601], ids=["one_row", "two_rows", "four_rows"])
@pytest.mark.parametrize("precision", [capi.PRECISION_F32, capi.PRECISION_F64, capi.PRECISION_MIXED], ids=["f32", "f64", "mixed"])
@pytest.mark.parametrize("delay", [1, 2], ids=["d1", "d2"])
def test_events_entry_equals_the_two_call_chain(batch, precision, delay):
    """gvtm_synthesize_events_device (event lists in, samples out, the frames in a buffer the plan owns): samples, sample
    counts, frame counts, peaks and drift-generator states must be those of gvtm_generate_tracks_device followed by
    gvtm_synthesize_batch_device, bit for bit, in every workgroup shape the product picks; one utterance is also taken
    through the oracle of the oracle (frames by the tracks oracle, pinned to captured generateOutput() calls, then the
    vocal-tract oracle)."""
    import torch
    if batch > 5 and delay == 2 and precision == capi.PRECISION_MIXED:
        pytest.skip("covered by the other combinations")
    cfgv = np.array([4, 1, 1, 1, 1, -20.0, -6.0, 4.0, 250.0, 4.0])
    pool_n = [40, 2, 1, 17, 60, 3, 55, 9, 33, 25, 48]
    pool = [_singable_event_table(300 + b, n) for b, n in enumerate(pool_n)]
    tables = [pool[b % len(pool)] for b in range(batch)]
    tc = _product_config(cfgv)
    frames_of = [capi.tracks_frame_count(tc, capi.events_from_table(t)) for t in pool]
    max_frames = max(frames_of)
    d_events, d_offsets = _events_on_device(tables)
    dev = d_events.device
    stream = torch.cuda.current_stream().cuda_stream
    plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, delay, precision), 250.0, 0)
    stride = plan.output_capacity(max_frames)
    drift0 = np.tile(np.array(oracle.FRESH_DRIFT, dtype=np.float64), (batch, 1))
    drift0[:, 0] = 0.1 + 0.8 * np.random.default_rng(5).random(batch)  # generators that have run before

    def fresh():
        return (torch.zeros((batch, stride), dtype=torch.float32, device=dev), torch.zeros(batch, dtype=torch.int32, device=dev),
                torch.zeros(batch, dtype=torch.int64, device=dev), torch.zeros(batch, dtype=torch.float32, device=dev),
                torch.from_numpy(drift0.copy()).to(dev))

    # the two-call chain, frames in device memory
    a1, f1, n1, m1, dr1 = fresh()
    d_params = torch.zeros((batch, max_frames, 16), dtype=torch.float32, device=dev)
    capi.generate_tracks_device(tc, d_events, d_offsets, batch, max_frames, d_params, f1, dr1, stream)
    plan.synthesize_device(d_params, batch, max_frames, a1, stride, f1, n1, m1, stream)
    # one call
    a2, f2, n2, m2, dr2 = fresh()
    plan.synthesize_events_device(tc, d_events, d_offsets, batch, max_frames, a2, stride, f2, n2, m2, dr2, stream)
    torch.cuda.synchronize()
    # (random event lists can still hold combinations the model itself cannot sing -- a parameter plus its "special" offset
    # outside its range -- and an utterance that goes non-finite says nothing: compared are the ones that stay finite)
    ok = torch.isfinite(a1).all(dim=1) & torch.isfinite(m1)
    assert int(ok[:len(pool)].sum().item()) >= (len(pool) * 2) // 3 or batch <= 5
    assert torch.equal(f1, f2) and torch.equal(n1, n2)
    assert torch.equal(dr1.view(torch.int64), dr2.view(torch.int64))
    assert torch.equal(m1[ok].view(torch.int32), m2[ok].view(torch.int32))
    assert torch.equal(a1[ok].view(torch.int32), a2[ok].view(torch.int32))
    assert f2[:len(pool)].cpu().tolist() == frames_of[:batch]
    # utterance 0 against the oracles (through the chain's own frames, which carry its drift generator's history)
    frames4 = d_params[0, : frames_of[0]].cpu().numpy()
    ref = oracle.synthesize(oracle.male_config(44100.0, delay, float_model=int(precision == capi.PRECISION_F32)), frames4)
    got = a2[0, : ref.size].cpu().numpy()
    assert n2[0].item() == ref.size and np.isfinite(ref).all()
    if precision == capi.PRECISION_F32:
        assert np.array_equal(got, ref)
    else:
        tol = 1e-9 if precision == capi.PRECISION_F64 else 1e-5
        assert np.abs(got.astype(np.float64) - ref).max() <= tol * np.abs(ref).max() + np.spacing(np.float32(np.abs(ref).max()))


def test_events_entry_truncates_at_max_frames_and_serves_every_model():
    """A list that yields more frames than the rows hold is cut at max_frames (frame count still reported in full, drift state
    that of the whole list); SectionDelay 3 and reference model 5 take the same call."""
    import torch
    cfgv = np.array([4, 1, 1, 1, 1, -20.0, -6.0, 4.0, 250.0, 4.0])
    tc = _product_config(cfgv)
    tables = [_singable_event_table(900 + b, n) for b, n in enumerate([30, 12, 30])]
    counts = [capi.tracks_frame_count(tc, capi.events_from_table(t)) for t in tables]
    d_events, d_offsets = _events_on_device(tables)
    dev = d_events.device
    stream = torch.cuda.current_stream().cuda_stream
    cut = min(counts[0], counts[2]) - 7
    assert cut > counts[1]
    for make_plan in (lambda: g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 1, capi.PRECISION_F32), 250.0, 0),
                      lambda: g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, 3, capi.PRECISION_F64), 250.0, 0),
                      lambda: g.Plan(g.config5_from_dict(g.read_config_file(oracle.VOICE5_MALE)), 250.0, 0)):
        plan = make_plan()
        stride = plan.output_capacity(cut)
        outs = []
        for fused in (False, True):
            a = torch.zeros((3, stride), dtype=torch.float32, device=dev)
            f = torch.zeros(3, dtype=torch.int32, device=dev)
            n = torch.zeros(3, dtype=torch.int64, device=dev)
            dr = torch.from_numpy(np.tile(np.array(oracle.FRESH_DRIFT), (3, 1))).to(dev)
            if fused:
                plan.synthesize_events_device(tc, d_events, d_offsets, 3, cut, a, stride, f, n, None, dr, stream)
            else:
                d_params = torch.zeros((3, cut, 16), dtype=torch.float32, device=dev)
                capi.generate_tracks_device(tc, d_events, d_offsets, 3, cut, d_params, f, dr, stream)
                plan.synthesize_device(d_params, 3, cut, a, stride, f, n, None, stream)
            torch.cuda.synchronize()
            outs.append((a, f, n, dr))
        assert outs[0][1].cpu().tolist() == counts and outs[1][1].cpu().tolist() == counts
        fin = torch.isfinite(outs[0][0]).all(dim=1)
        assert bool(fin.any())
        assert torch.equal(outs[0][0][fin].view(torch.int32), outs[1][0][fin].view(torch.int32))
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
        assert torch.equal(outs[0][3].view(torch.int64), outs[1][3].view(torch.int64))
        assert outs[1][2][0].item() == plan.output_count(cut) and outs[1][2][1].item() == plan.output_count(counts[1])
