"""Event lists for the track-generation tests: the captured fixtures and a synthetic generator."""
import numpy as np


def load_golden(golden_tracks, name, call):
    cfg = golden_tracks["%s__%d__cfg" % (name, call)]
    return cfg, golden_tracks["%s__%d__events" % (name, call)], golden_tracks["%s__%d__frames" % (name, call)]


def random_event_table(seed, n_events=40, control_period=4, max_gap_periods=12):
    """A synthetic event list in the 38-column form: increasing times (multiples of the control period), every
    parameter set on the first event, later events setting a random subset (+inf = not set), occasional special
    parameters, macro-intonation polynomials on some events."""
    rng = np.random.default_rng(seed)
    t = np.zeros((n_events, 38))
    time = 0
    for i in range(n_events):
        if i:
            time += control_period * int(rng.integers(1, max_gap_periods + 1))
        t[i, 0] = time
        t[i, 6:38] = np.inf
        has = rng.random(16) < (1.0 if i == 0 else 0.55)
        vals = np.concatenate([rng.uniform(-10, 2, 1), rng.uniform(0, 60, 3), rng.uniform(0, 7, 1), rng.uniform(100, 5500, 1),
                               rng.uniform(250, 4500, 1), rng.uniform(0.1, 3.0, 8), rng.uniform(0.1, 1.5, 1)])
        t[i, 6:22][has] = vals[has]
        sp = rng.random(16) < 0.08
        t[i, 22:38][sp] = rng.uniform(-2, 2, 16)[sp]
        if rng.random() < 0.3:
            t[i, 1] = 1.0
            t[i, 2:6] = rng.uniform(-1e-6, 1e-6), rng.uniform(-1e-3, 1e-3), rng.uniform(-0.05, 0.05), rng.uniform(-6, 6)
    return t
