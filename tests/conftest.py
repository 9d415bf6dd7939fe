import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(HERE, "golden", "vtm_golden.npz"), allow_pickle=False)
    data = {k: z[k] for k in z.files}
    data["manifest"] = json.loads(bytes(data.pop("manifest_json")).decode())
    return data


@pytest.fixture(scope="session")
def golden5():
    """Reference VocalTractModel5 vectors (tests/golden/make_vtm5_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "vtm5_golden.npz"), allow_pickle=False)
    data = {k: z[k] for k in z.files}
    data["manifest"] = json.loads(bytes(data.pop("manifest_json")).decode())
    return data


@pytest.fixture(scope="session")
def golden_tracks():
    """Captured EventList::generateOutput() calls of the reference (tests/golden/make_tracks_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "tracks_golden.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_overrun():
    """Reference vectors at flush-overrun lengths of the sample-rate converter (tests/golden/make_overrun_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "vtm_overrun_golden.npz"), allow_pickle=False)
    data = {k: z[k] for k in z.files}
    data["manifest"] = json.loads(bytes(data.pop("manifest_json")).decode())
    return data


@pytest.fixture(scope="session")
def golden_wav():
    """16-bit WAV files written by the reference's Controller::synthesizeToFile (tests/golden/make_wav_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "wav_golden.npz"), allow_pickle=False)
    data = {k: z[k] for k in z.files}
    data["manifest"] = json.loads(bytes(data.pop("manifest_json")).decode())
    return data
