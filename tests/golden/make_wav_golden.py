#!/usr/bin/env python3
"""Generates tests/golden/wav_golden.npz: 16-bit WAV files WRITTEN BY THE REFERENCE (Controller::synthesizeToFile ->
writeOutputToFile -> WAVEFileWriter, through oracle/_ref/ref_wav_capture = the reference's `gama_tts vtm`) for the
captured "Hello world" frames of BASELINE configs[0].

Build-container only (`make -C oracle ref_full`).  The voice directory is the reference's own data/voice/english/0_male
(model 0 = VocalTractModel0<double>, 48 kHz) — for the other cases a temporary directory of symlinks to it whose vtm.txt
differs in `model` / `output_rate` only.  The .npz holds data only: the WAV bytes and their SHA-256; the input frames are
`hello_params` of vtm_golden.npz, written as %.9g text (exact for float32).

    python tests/golden/make_wav_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402

REF_VOICE_DIR = "/root/reference/data/voice/english/0_male"
CASES = [  # name, model, output_rate, frames (None = all 332)
    ("hello_m0_48k", "0", "48000.0", None),   # the shipped voice as it is
    ("hello_m1_48k", "1", "48000.0", None),   # VocalTractModel0<float>
    ("hello_m0_44k", "0", "44100.0", None),
    ("hello_m1_44k", "1", "44100.0", None),
    ("short40_m1_44k", "1", "44100.0", 40),
    ("hello_m4_44k", "4", "44100.0", None),   # VocalTractModel4 (down-sampling converter)
]


def voice_dir(td, model, rate):
    root = os.path.join(td, "voice_%s_%s" % (model, rate))
    if os.path.isdir(root):
        return root
    os.makedirs(root)
    for entry in os.listdir(REF_VOICE_DIR):
        if entry != "vtm.txt":
            os.symlink(os.path.join(REF_VOICE_DIR, entry), os.path.join(root, entry))
    with open(os.path.join(REF_VOICE_DIR, "vtm.txt")) as f, open(os.path.join(root, "vtm.txt"), "w") as g:
        for line in f:
            key = line.split("=")[0].strip()
            if key == "model":
                line = "model = %s\n" % model
            elif key == "output_rate":
                line = "output_rate = %s\n" % rate
            g.write(line)
    return root


def main():
    exe = os.path.join(oracle.REF_DIR, "ref_wav_capture")
    hello = np.load(os.path.join(HERE, "vtm_golden.npz"), allow_pickle=False)["hello_params"]
    out, manifest = {}, {}
    with tempfile.TemporaryDirectory() as td:
        for name, model, rate, frames in CASES:
            tr = hello if frames is None else hello[:frames]
            ptxt = os.path.join(td, name + ".txt")
            with open(ptxt, "w") as f:
                for row in tr:
                    f.write(" ".join("%.9g" % v for v in row) + "\n")
            wav = os.path.join(td, name + ".wav")
            subprocess.run([exe, voice_dir(td, model, rate), ptxt, wav], check=True)
            data = open(wav, "rb").read()
            out[name + "__wav"] = np.frombuffer(data, dtype=np.uint8)
            manifest[name] = dict(model=model, output_rate=float(rate), frames=int(tr.shape[0]), bytes=len(data),
                                  sha256=hashlib.sha256(data).hexdigest())
            print(name, len(data), manifest[name]["sha256"][:12])
    out["manifest_json"] = np.frombuffer(json.dumps(manifest, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "wav_golden.npz"), **out)


if __name__ == "__main__":
    main()
