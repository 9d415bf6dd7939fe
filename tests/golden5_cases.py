"""Parity fixtures of VocalTractModel5 (reference model 5, the 5_male voice) — oracle only this round.

Shared by tests/golden/make_vtm5_golden.py (runs the REAL reference, oracle/_ref/ref_vtm) and
tests/test_oracle5_vs_golden.py.  model "5" = VocalTractModel5<double,1> from the factory
(vtm/VocalTractModel.cpp:47-48), "5f" = VocalTractModel5<float,1> instantiated by oracle/ref_driver.cpp.
"""
import golden_cases


def C(name, track, model="5", rate=48000.0, crate=250.0, store="full", **ov):
    return dict(name=name, track=track, model=model, rate=rate, crate=crate, store=store,
                float_model=1 if model == "5f" else 0, overrides=ov)


CASES = [
    # const track, 44.1 kHz -> 88356 samples
    C("const_m5_44k", ("const", 500), rate=44100.0, store="digest"),
    C("ramp_m5", ("ramp", 500), store="digest"),
    C("cons2000_m5", ("random", 500, 2000, True), store="digest"),
    C("hello_m5", ("hello",), store="digest"),
    C("rand5_m5", ("random", 120, 5, True)),
    C("rand6_m5_44k", ("random", 120, 6, False), rate=44100.0),
    C("rand7_m5_22k_crate500", ("random", 120, 7, True), rate=22050.0, crate=500.0),
    C("bypass_m5", ("random", 120, 5, True), bypass=1),
    C("sine_m5", ("random", 120, 5, True), waveform=1),
    C("tn_delta_m5", ("random", 120, 5, True), glottal_pulse_tn_min=16.0, glottal_pulse_tn_max=32.0),
    C("no_modulation_m5", ("random", 120, 5, True), noise_modulation=0),
    C("constant_mouth_m5", ("random", 120, 5, True), constant_radius_mouth_impedance="true", mouth_impedance_radius=1.2),
    C("female_m5", ("random", 120, 6, False), vocal_tract_length=15.0, glottal_pulse_tn_min=32.0,
      glottal_pulse_tn_max=32.0, breathiness=1.5),
    C("radius_coefs_m5", ("random", 120, 8, True), radius_3_coef=1.3, global_radius_coef=0.9,
      global_nasal_radius_coef=1.1, vocal_tract_length_offset=1.0, loss_factor=0.8, max_glottal_loss=5.0, min_glottal_loss=1.0),
    C("silence_m5", ("silence", 40)),
    C("one_frame_m5", ("random", 120, 5, True, 1)),
    C("three_frames_m5", ("random", 120, 5, True, 3)),
    C("rand5_m5f", ("random", 120, 5, True), model="5f"),
    C("cons2000_m5f", ("random", 500, 2000, True), model="5f", store="digest"),
    C("bypass_m5f", ("random", 120, 5, True), model="5f", bypass=1),
]

DIGEST_STRIDE = golden_cases.DIGEST_STRIDE
track_for = golden_cases.track_for
