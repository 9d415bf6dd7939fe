"""Deterministic synthetic VTM parameter tracks (test + bench inputs).

Frame layout follows the reference's parameter order (VocalTractModel0.h:160-178):
pitch, glotVol, aspVol, fricVol, fricPos, fricCF, fricBW, r1..r8, velum.
The `const`/`ramp` recipes are the known-answer tracks of SURVEY.md section 0;
`random_tracks` is the generator of SURVEY.md 8(d) config 2/3 (ranges from the
reference's data/voice/english/0_male/interactive.txt).
"""
import numpy as np

N_PARAM = 16

A = np.array([-12, 60, 0, 0, 5.5, 2500, 500, 0.8, 0.65, 0.65, 0.65, 1.31, 1.23, 1.31, 1.67, 0.1],
             dtype=np.float64)
B = np.array([-5, 40, 20, 30, 6.7, 4500, 2000, 0.8, 1.2, 1.0, 0.4, 0.9, 0.3, 1.5, 0.9, 1.0],
             dtype=np.float64)


def const_track(frames=500):
    return np.tile(A.astype(np.float32), (frames, 1))


def ramp_track(frames=500):
    i = np.arange(frames, dtype=np.float64)[:, None]
    return (A + (B - A) * i / (frames - 1)).astype(np.float32)


# (lo, hi) per parameter — editor ranges
_RANGES = np.array([
    (-20, 0), (40, 60), (0, 20), (0, 30), (0, 7), (100, 5500), (250, 4500),
    (0.1, 3.0), (0.1, 3.0), (0.1, 3.0), (0.1, 3.0), (0.1, 3.0), (0.1, 3.0), (0.1, 3.0), (0.1, 3.0),
    (0.1, 1.5)], dtype=np.float64)


def random_track(frames, seed, consonant_heavy=False, unvoiced=0.3):
    """Piecewise-linear track between random key-frames every 20-60 frames; `unvoiced` = share of key-frames with the
    glottal volume at 0 (30 %; 0.0 gives a corpus that never takes the kernel's voicing gate)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    keys_t = [0]
    while keys_t[-1] < frames - 1:
        keys_t.append(min(frames - 1, keys_t[-1] + int(rng.integers(20, 61))))
    keys_t = np.array(keys_t)
    nk = len(keys_t)
    u = rng.random((nk, N_PARAM))
    vals = _RANGES[:, 0] + u * (_RANGES[:, 1] - _RANGES[:, 0])
    silent = rng.random(nk) < 0.3  # (always drawn: the other parameters do not depend on `unvoiced`)
    if unvoiced >= 0.3:
        vals[silent, 1] = 0.0
    elif unvoiced > 0.0:
        vals[silent & (rng.random(nk) < unvoiced / 0.3), 1] = 0.0
    if consonant_heavy:
        m = rng.random(nk) < 0.5
        vals[m, 15] = 0.5 + rng.random(m.sum()) * 1.0          # velum >= 0.5
        m = rng.random(nk) < 0.5
        vals[m, 3] = 20.0 + rng.random(m.sum()) * 10.0         # fricVol >= 20
        m = rng.random(nk) < 0.3
        vals[m, 2] = 10.0 + rng.random(m.sum()) * 10.0         # aspVol >= 10
    t = np.arange(frames)
    out = np.empty((frames, N_PARAM), dtype=np.float64)
    for p in range(N_PARAM):
        out[:, p] = np.interp(t, keys_t, vals[:, p])
    return out.astype(np.float32)


def random_tracks(batch, frames, seed0=1000, consonant_heavy=False, unvoiced=0.3):
    return np.stack([random_track(frames, seed0 + b, consonant_heavy, unvoiced) for b in range(batch)])


def edge_track(frames=48, seed=0):
    """Frames that sit ON the special cases of the per-step conversions: volumes 0 / 60 dB (Util::amplitude60dB's two
    shortcuts), frication position 0 and 7 (first / last injection section, the dropped right share), radii at and below
    the 0.01 floor, velum 0 (nasal junction coefficient -1), pitch at both ends, bandwidth / centre frequency extremes;
    held for two frames each so that the interpolation passes THROUGH the values as well."""
    base = const_track(1)[0]
    rows = []
    specials = [
        {1: 0.0}, {1: 60.0}, {2: 60.0}, {2: 0.0}, {3: 60.0, 4: 0.0}, {3: 60.0, 4: 7.0}, {3: 30.0, 4: 6.999}, {3: 30.0, 4: 3.5},
        {0: -24.0}, {0: 24.0}, {5: 100.0, 6: 250.0, 3: 40.0}, {5: 5500.0, 6: 4500.0, 3: 40.0},
        {7: 0.0, 8: 0.005, 9: 0.01}, {14: 0.0}, {14: 3.0, 13: 0.0}, {15: 0.0}, {15: 1.5, 10: 0.1}, {10: 3.0, 11: 0.1},
    ]
    for sp in specials:
        r = base.copy()
        r[3] = 0.0
        for k, v in sp.items():
            r[k] = v
        rows += [r, r]
    tr = np.array(rows, dtype=np.float32)
    if frames > tr.shape[0]:
        tr = np.concatenate([tr, random_track(frames - tr.shape[0], 9000 + seed, True)])
    return np.ascontiguousarray(tr[:frames], dtype=np.float32)
