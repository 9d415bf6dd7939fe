"""Track generation (EventList::generateOutput) without a GPU: the oracle against the captured reference
fixtures, the product's host-side frame counter and argument checks."""
import ctypes

import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import event_lists
import oracle

TEXTS = ("hello", "fox", "question", "count")


@pytest.mark.parametrize("name", TEXTS)
def test_oracle_matches_captured_reference_calls(name, golden_tracks):
    """Six generateOutput() calls per text on the reference's own event list, drift generator running on from
    call to call: bit-identical frames."""
    state = oracle.FRESH_DRIFT
    for call in range(6):
        cfg, events, frames = event_lists.load_golden(golden_tracks, name, call)
        got, state = oracle.tracks_generate(oracle.track_config(cfg), events, state)
        assert got.shape == frames.shape
        assert np.array_equal(got.view(np.uint32), frames.view(np.uint32)), (name, call)


def _product_config(cfg):
    c = g.TrackConfig()
    c.control_period_ms, c.macro_intonation, c.micro_intonation, c.intonation_drift, c.smooth_intonation = (int(x) for x in cfg[:5])
    c.initial_pitch, c.mean_pitch, c.drift_deviation, c.drift_sample_rate, c.drift_lowpass_cutoff = (float(x) for x in cfg[5:10])
    return c


@pytest.mark.parametrize("name", TEXTS)
def test_frame_count_matches_reference(name, golden_tracks):
    cfg, events, frames = event_lists.load_golden(golden_tracks, name, 0)
    assert capi.tracks_frame_count(_product_config(cfg), capi.events_from_table(events)) == frames.shape[0]


def test_frame_count_on_synthetic_lists_and_edges():
    for seed in range(20):
        for cp in (1, 2, 4):
            table = event_lists.random_event_table(seed, n_events=3 + seed, control_period=cp)
            cfg = np.array([cp, 1, 1, 1, 1, -20.0, -6.0, 4.0, 1000.0 / cp, 4.0])
            want = oracle.tracks_generate(oracle.track_config(cfg), table)[0].shape[0]
            assert capi.tracks_frame_count(_product_config(cfg), capi.events_from_table(table)) == want
    cfg = np.array([4, 1, 1, 1, 1, -20.0, -6.0, 4.0, 250.0, 4.0])
    one = capi.events_from_table(event_lists.random_event_table(1, n_events=1))
    assert capi.tracks_frame_count(_product_config(cfg), one) == 0  # fewer than two events: nothing (EventList.cpp:932)


def test_event_record_layout_and_no_cpu_path():
    assert capi.EVENT_DTYPE.itemsize == 296 and capi.DRIFT_DTYPE.itemsize == 40
    bad = _product_config(np.array([4, 1, 1, 1, 1, -20.0, -6.0, 4.0, 250.0, 200.0]))  # cutoff above 0.48 of the rate
    with pytest.raises(capi.GvtmError):
        capi.tracks_frame_count(bad, capi.events_from_table(event_lists.random_event_table(2)))
    if capi.device_count() == 0:
        cfg = _product_config(np.array([4, 1, 1, 1, 1, -20.0, -6.0, 4.0, 250.0, 4.0]))
        with pytest.raises(capi.GvtmError) as ei:
            capi.generate_tracks_host(cfg, [capi.events_from_table(event_lists.random_event_table(3))], 64)
        assert ei.value.status == 2  # GVTM_ERR_NO_DEVICE
