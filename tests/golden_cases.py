"""The parity fixture list shared by tests/golden/make_golden.py (which runs the REAL
reference, oracle/_ref/ref_vtm, in the build container) and by the tests that replay
the fixtures (oracle on CPU, HIP path on the GPU).

Each case: name, track recipe, model string for ref_vtm, SectionDelay for our side,
output rate, control rate, config overrides (on top of tests/golden/voice_male.txt),
and how much of the reference output is stored ("full" or "digest").
"""
import numpy as np

import tracks


def _track(spec, golden=None):
    kind = spec[0]
    if kind == "const":
        return tracks.const_track(spec[1])
    if kind == "ramp":
        return tracks.ramp_track(spec[1])
    if kind == "random":
        t = tracks.random_track(spec[1], spec[2], spec[3])
        return t[: spec[4]] if len(spec) > 4 else t
    if kind == "hello":
        return np.asarray(golden["hello_params"], dtype=np.float32)
    if kind == "silence":
        t = tracks.const_track(spec[1])
        t[:, 1:4] = 0.0
        return t
    raise ValueError(kind)


def track_for(case, golden=None):
    return _track(case["track"], golden)


def C(name, track, model="0", delay=1, rate=44100.0, crate=250.0, store="full", layout=0, fm=0, **ov):
    # fm = 1: the reference model computes in float (TFloat = float)
    return dict(name=name, track=track, model=model, delay=delay, rate=rate, crate=crate, store=store, layout=layout,
                float_model=fm, overrides=ov)


CASES = [
    # SURVEY.md section 0 known-answer tracks (500 frames -> 88108 samples): digest + strided subset
    C("const_m0", ("const", 500), store="digest"),
    C("ramp_m0", ("ramp", 500), store="digest"),
    C("const_m2d2", ("const", 500), model="2:2", delay=2, store="digest"),
    C("ramp_m2d2", ("ramp", 500), model="2:2", delay=2, store="digest"),
    C("const_m3", ("const", 500), model="3", delay=3, store="digest"),
    C("ramp_m3", ("ramp", 500), model="3", delay=3, store="digest"),
    C("rand1000_m0", ("random", 500, 1000, False), store="digest"),
    C("cons2000_m2", ("random", 500, 2000, True), model="2", store="digest"),
    # the real text->posture track of BASELINE.json configs[0]
    C("hello_m0", ("hello",), store="full"),
    C("hello_m0_48k", ("hello",), rate=48000.0, store="digest"),
    # short tracks, full output stored
    C("rand5_m0", ("random", 120, 5, True)),
    C("rand5_m2d2", ("random", 120, 5, True), model="2:2", delay=2),
    C("rand5_m3", ("random", 120, 5, True), model="3", delay=3),
    C("tn_delta", ("random", 120, 5, True), glottal_pulse_tn_min=16.0, glottal_pulse_tn_max=32.0),
    C("tn_delta_d2", ("random", 120, 5, True), model="2:2", delay=2, glottal_pulse_tn_min=10.0, glottal_pulse_tn_max=40.0),
    C("sine", ("random", 120, 5, True), waveform=1),
    C("no_modulation", ("random", 120, 5, True), noise_modulation=0),
    C("out48k", ("random", 120, 5, True), rate=48000.0),
    C("out22k", ("random", 120, 5, True), rate=22050.0),
    C("out16k_down", ("random", 120, 5, True), rate=16000.0),
    C("crate1000", ("random", 120, 5, True), crate=1000.0),
    C("crate500_m3", ("random", 120, 5, True), model="3", delay=3, crate=500.0),
    C("female", ("random", 120, 6, False), vocal_tract_length=15.0, glottal_pulse_tn_min=32.0,
      glottal_pulse_tn_max=32.0, breathiness=1.5),
    C("small_child", ("random", 120, 7, False), vocal_tract_length=10.0, breathiness=1.5),
    C("radius_coefs", ("random", 120, 8, True), radius_3_coef=1.3, global_radius_coef=0.9,
      global_nasal_radius_coef=1.1, vocal_tract_length_offset=1.0),
    C("silence", ("silence", 40)),
    C("one_frame", ("random", 120, 5, True, 1)),
    C("two_frames_m3", ("random", 120, 5, True, 2), model="3", delay=3),
    C("three_frames", ("random", 120, 5, True, 3)),
    C("thirteen_frames_m3", ("random", 120, 5, True, 13), model="3", delay=3),
    # VocalTractModel4 (30 + 18 sections, 60102 Hz internal): SURVEY.md section 0 + short tracks
    C("const_m4", ("const", 500), model="4", layout=1, store="digest"),
    C("ramp_m4", ("ramp", 500), model="4", layout=1, store="digest"),
    C("cons2000_m4", ("random", 500, 2000, True), model="4", layout=1, store="digest"),
    C("hello_m4", ("hello",), model="4", layout=1, store="digest"),
    C("rand5_m4", ("random", 120, 5, True), model="4", layout=1),
    C("rand5_m4_48k", ("random", 120, 5, True), model="4", layout=1, rate=48000.0),
    C("rand6_m4_22k_crate500", ("random", 120, 6, False), model="4", layout=1, rate=22050.0, crate=500.0),
    C("two_frames_m4", ("random", 120, 5, True, 2), model="4", layout=1),
    # TFloat = float: model 1 = VocalTractModel0<float> from the factory; VocalTractModel2<float,D> and
    # VocalTractModel4<float,1> instantiated directly by oracle/ref_driver.cpp ("2f:D", "4f")
    C("const_m1", ("const", 500), model="1", fm=1, store="digest"),
    C("ramp_m1", ("ramp", 500), model="1", fm=1, store="digest"),
    C("rand1000_m1", ("random", 500, 1000, False), model="1", fm=1, store="digest"),
    C("cons2000_m1", ("random", 500, 2000, True), model="1", fm=1, store="digest"),
    C("hello_m1", ("hello",), model="1", fm=1, store="digest"),
    C("rand5_m1", ("random", 120, 5, True), model="1", fm=1),
    C("rand5_m2f_d2", ("random", 120, 5, True), model="2f:2", delay=2, fm=1),
    C("rand5_m2f_d3", ("random", 120, 5, True), model="2f:3", delay=3, fm=1),
    C("rand5_m4f", ("random", 120, 5, True), model="4f", layout=1, fm=1),
    C("tn_delta_m1", ("random", 120, 5, True), model="1", fm=1, glottal_pulse_tn_min=16.0, glottal_pulse_tn_max=32.0),
    C("sine_m1", ("random", 120, 5, True), model="1", fm=1, waveform=1),
    C("no_modulation_m1", ("random", 120, 5, True), model="1", fm=1, noise_modulation=0),
    C("out48k_m1", ("random", 120, 5, True), model="1", fm=1, rate=48000.0),
    C("out16k_down_m1", ("random", 120, 5, True), model="1", fm=1, rate=16000.0),
    C("crate1000_m1", ("random", 120, 5, True), model="1", fm=1, crate=1000.0),
    C("female_m1", ("random", 120, 6, False), model="1", fm=1, vocal_tract_length=15.0, glottal_pulse_tn_min=32.0,
      glottal_pulse_tn_max=32.0, breathiness=1.5),
    C("radius_coefs_m1", ("random", 120, 8, True), model="1", fm=1, radius_3_coef=1.3, global_radius_coef=0.9,
      global_nasal_radius_coef=1.1, vocal_tract_length_offset=1.0),
    C("silence_m1", ("silence", 40), model="1", fm=1),
    C("one_frame_m1", ("random", 120, 5, True, 1), model="1", fm=1),
    C("three_frames_m2f_d2", ("random", 120, 5, True, 3), model="2f:2", delay=2, fm=1),
]

DIGEST_STRIDE = 97

# Lengths at which the reference's SampleRateConverter runs into its flush overrun (SampleRateConverter.h:298-308 with
# :462-471: the final dataEmpty() finds endPtr < emptyPtr_, adds BUFFER_SIZE and converts one more lap of the ring from
# its leftovers).  Own file (tests/golden/vtm_overrun_golden.npz, made by tests/golden/make_overrun_golden.py from the
# REAL reference); "tail" = SHA-256 of everything + a strided subset + the last OVERRUN_TAIL samples (the extra lap and
# what precedes it).  model5 = True: the 5_male voice / VocalTractModel5.
OVERRUN_TAIL = 1600


def O(name, frames, seed, model, delay=1, rate=44100.0, layout=0, fm=0, model5=False, store="tail"):
    c = C(name, ("random", frames, seed, True), model=model, delay=delay, rate=rate, store=store, layout=layout, fm=fm)
    c["model5"] = model5
    return c


OVERRUN_CASES = [
    O("ovr_d2_22k_18f", 18, 31, "2:2", delay=2, rate=22050.0, store="full"),
    O("ovr_d2_22k_79f_float", 79, 32, "2f:2", delay=2, rate=22050.0, fm=1, store="full"),
    O("ovr_m3_22k_83f", 83, 33, "3", delay=3, rate=22050.0, store="full"),
    O("ovr_m4_22k_83f", 83, 34, "4", layout=1, rate=22050.0, store="full"),
    O("ovr_m4f_22k_202f", 202, 35, "4f", layout=1, rate=22050.0, fm=1),
    # the shipped voices' own rate class: models 3 / 4 at 44.1 kHz overrun at 2334, 2581, 2828 frames
    O("ovr_m3_44k_2334f", 2334, 36, "3", delay=3),
    O("ovr_m4_44k_2334f", 2334, 37, "4", layout=1),
    O("ovr_m3f_44k_2581f", 2581, 38, "2f:3", delay=3, fm=1),
    O("ovr_m4f_44k_2828f", 2828, 39, "4f", layout=1, fm=1),
    # model 5 (5_male voice) at 44.1 kHz: 106 frames; at 22.05 kHz: 696
    O("ovr_m5_44k_106f", 106, 40, "5", model5=True, store="full"),
    O("ovr_m5_22k_696f", 696, 41, "5", rate=22050.0, model5=True),
]
