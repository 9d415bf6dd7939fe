"""The kernels in the shapes the PRODUCT library picks by itself, against the oracle.

`gvtm_plan_create` + `gvtm_synthesize_batch_*` choose the utterances per workgroup from the batch size (csrc/vtm_kernels.hip,
synth_rows: one up to 256 utterances, two up to 512, four above).  The other GPU tests either stay at <= 256 utterances
(one per workgroup) or force a shape through libgama_vtm_diag.so; here the plans come from libgama_vtm.so with nothing
forced, at the batch sizes that select the two- and four-row kernels -- in particular
vtm_synth_kernel<double,double,2,4,32,7,0> and <double,float,2,4,32,7,0>, the fp64 / mixed kernels bench.py times on
BASELINE.json configs[3] (SectionDelay 2 = VocalTractModel2<double,2>, vtm/VocalTractModel2.h:626-669).

A pool of <= 32 distinct ragged tracks is tiled over the batch: every pool member is compared with the oracle (fp64:
1e-9 of peak or one float32 ulp of the sample; mixed: north_star's 1e-5; float: bit-identical to the float oracle), and
every copy of a track must equal its first occurrence bit for bit, whichever workgroup and DPP row it landed in.
"""
import numpy as np
import pytest

import gama_tts_amd as g
from gama_tts_amd import capi
import oracle
import tracks

pytestmark = pytest.mark.gpu

MAX_FRAMES = 36
POOL = 32


def _within(got, ref, tol):
    ref64 = ref.astype(np.float64)
    peak = float(np.abs(ref64).max())
    d = np.abs(got.astype(np.float64) - ref64)
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    return bool((d <= np.maximum(ulp, tol * max(peak, 1e-300))).all())


def _pool(seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(1, MAX_FRAMES + 1, size=POOL).astype(np.int32)
    frames[:4] = [MAX_FRAMES, 0, 1, 2]
    params = tracks.random_tracks(POOL, MAX_FRAMES, seed0=seed, consonant_heavy=True)
    params[::3] = tracks.random_tracks(len(params[::3]), MAX_FRAMES, seed0=seed + 500, consonant_heavy=False)
    return params, frames


@pytest.mark.parametrize("batch,expect_rows", [(384, 2), (640, 4), (1027, 4)], ids=["b384_two_rows", "b640_four_rows", "b1027_four_rows"])
@pytest.mark.parametrize("precision,float_model,tol", [(capi.PRECISION_F64, 0, 1e-9), (capi.PRECISION_MIXED, 0, 1e-5), (capi.PRECISION_F32, 1, 0.0)],
                         ids=["f64", "mixed", "f32"])
@pytest.mark.parametrize("delay", [2, 1], ids=["d2", "d1"])
def test_product_library_shape_selection_against_the_oracle(batch, expect_rows, precision, float_model, tol, delay):
    if delay == 1 and batch == 1027:
        pytest.skip("SectionDelay 1 is covered at 384 and 640")
    pool_params, pool_frames = _pool(5000 + 10 * delay + expect_rows)
    # a tiling whose period (32) is not a multiple of the rows: every track lands in every DPP row; an odd batch leaves the
    # last workgroup partly empty
    idx = np.arange(batch) % POOL
    params = pool_params[idx]
    frames = pool_frames[idx]
    plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, delay, precision), 250.0, 0)  # product library
    assert not plan.diagnostics
    audio, counts, maxabs = plan.synthesize_host(params, frames)
    refs = oracle.synthesize_many([(pool_params[t, : int(pool_frames[t])], 44100.0, delay, 0, float_model) for t in range(POOL)])
    for t in range(POOL):
        r = refs[t]
        assert counts[t] == r.size, (t, counts[t], r.size)
        got = audio[t, : r.size]
        if float_model:
            assert np.array_equal(got, r), t
        else:
            assert _within(got, r, tol), (t, float(np.abs(got.astype(np.float64) - r).max() / max(np.abs(r).max(), 1e-300)))
        assert maxabs[t] == (np.abs(got).max() if r.size else 0.0)
        assert not audio[t, r.size:].any()  # the rest of a ragged row comes back zero
    for b in range(POOL, batch):
        assert counts[b] == counts[b % POOL]
        assert np.array_equal(audio[b], audio[b % POOL]), (b, b % POOL)
