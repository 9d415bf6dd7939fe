#!/usr/bin/env python3
"""Generates tests/golden/tracks_golden.npz by running the REAL reference (build container only).

oracle/_ref/ref_tracks_capture (oracle/ref_tracks_capture.cpp, compiled against /root/reference by
`make -C oracle ref_full`) runs the reference's text parser, rules and EventList on a sentence and
records, for six EventList::generateOutput() calls with different intonation settings: the event list,
the settings, and the float32 parameter frames.  The text->event pipeline draws its intonation from
std::random_device, so a capture is taken once and kept: the .npz holds data only.

    python tests/golden/make_tracks_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402

REF_VOICE_DIR = "/root/reference/data/voice/english/0_male"
TEXTS = {
    "hello": "Hello world.",
    "fox": "The quick brown fox jumps over the lazy dog.",
    "question": "Is this a question?",
    "count": "Testing one, two, three.",
}


def parse(path):
    data = open(path, "rb").read()
    assert data[:4] == b"GVTR"
    version, n_calls = struct.unpack_from("<ii", data, 4)
    assert version == 1
    off = 12
    calls = []
    for _ in range(n_calls):
        ints = struct.unpack_from("<5i", data, off); off += 20
        dbl = struct.unpack_from("<5d", data, off); off += 40
        (n_events,) = struct.unpack_from("<i", data, off); off += 4
        events = np.frombuffer(data, dtype="<f8", count=n_events * 38, offset=off).reshape(n_events, 38).copy(); off += n_events * 38 * 8
        (n_frames,) = struct.unpack_from("<i", data, off); off += 4
        frames = np.frombuffer(data, dtype="<f4", count=n_frames * 16, offset=off).reshape(n_frames, 16).copy(); off += n_frames * 64
        calls.append((np.array(ints + dbl, dtype=np.float64), events, frames))
    assert off == len(data)
    return calls


def main():
    exe = os.path.join(oracle.REF_DIR, "ref_tracks_capture")
    out = {}
    for name, text in TEXTS.items():
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "t.bin")
            subprocess.run([exe, REF_VOICE_DIR, text, p], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for i, (cfg, events, frames) in enumerate(parse(p)):
                out["%s__%d__cfg" % (name, i)] = cfg        # control_period, macro, micro, drift, smooth, initial_pitch, mean_pitch, drift dev / rate / cutoff
                out["%s__%d__events" % (name, i)] = events  # [E][38]: time, has_interp, a, b, c, d, parameters[16], specialParameters[16]; +inf = empty
                out["%s__%d__frames" % (name, i)] = frames  # [F][16] float32
            print(name, [out["%s__%d__frames" % (name, i)].shape[0] for i in range(6)], out[name + "__0__events"].shape[0], "events")
    np.savez_compressed(os.path.join(HERE, "tracks_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "tracks_golden.npz"))


if __name__ == "__main__":
    main()
