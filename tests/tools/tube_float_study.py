"""Study (CPU only, test infrastructure): how far from VocalTractModel2<double,D> does the double model land when ONLY the
tube -- the scattering junctions of vocalTract() -- is computed in float32 and everything else (sources, filters,
coefficients, resampler) stays double?  The question behind it: the double kernels' tick is their fp64 tube wavefront
(DESIGN.md 4a); a float tube would run at the float kernel's pace IF it stayed inside north_star's 1e-5 of peak.

The tool writes a copy of oracle/vtm_oracle_body.inc to a scratch directory in which every arithmetic result of
model_vocal_tract() / propagate_junction() is rounded to float (a double operation on float operands rounded to float is
the float operation, up to double rounding of sums), compiles it next to the oracle's own sources and runs both on the
same tracks.  Nothing here is used by the product or by the parity tests.

usage: python tests/tools/tube_float_study.py [frames] [tracks]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import tracks  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle")

JUNCTION = r'''static void propagate_junction(vtm_model* m, section* l, real k, section* r, real fric)
{
	const real delta = TF(TF(k) * TF(l->top[m->out_ptr] - r->bottom[m->out_ptr]));
	r->top[m->in_ptr] = TF(TF(TF(l->top[m->out_ptr] + delta) * TF(m->damping)) + TF(fric));
	l->bottom[m->in_ptr] = TF(TF(r->bottom[m->out_ptr] + delta) * TF(m->damping));
}
'''

TRACT = r'''static real model_vocal_tract(vtm_model* m, real input, real frication)
{
	m->in_ptr = m->out_ptr;
	m->out_ptr = (m->out_ptr == (unsigned) m->delay) ? 0 : m->out_ptr + 1;
	const unsigned in = m->in_ptr, out = m->out_ptr;
	section* o = m->oro;
	section* n = m->nasal;
	const real d = TF(m->damping);

	o[S1].top[in] = TF(TF(o[S1].bottom[out] * d) + TF(input));
	{
		const real delta = TF(TF(m->oro_k[0]) * TF(o[S1].top[out] - o[S2].bottom[out]));
		o[S2].top[in] = TF(TF(o[S1].top[out] + delta) * d);
		o[S1].bottom[in] = TF(TF(o[S2].bottom[out] + delta) * d);
	}
	for (int i = S2, j = 1, k = 0; i < S4; ++i, ++j, ++k) {
		propagate_junction(m, &o[i], m->oro_k[j], &o[i + 1], m->tap[k] * frication);
	}
	{
		const real jp = TF(TF(TF(TF(m->alpha_l) * o[S4].top[out]) + TF(TF(m->alpha_r) * o[S5].bottom[out])) + TF(TF(m->alpha_u) * n[N1].bottom[out]));
		o[S4].bottom[in] = TF(TF(jp - o[S4].top[out]) * d);
		o[S5].top[in] = TF(TF(TF(jp - o[S5].bottom[out]) * d) + TF(m->tap[2] * frication));
		n[N1].top[in] = TF(TF(jp - n[N1].bottom[out]) * d);
	}
	propagate_junction(m, &o[S5], m->oro_k[3], &o[S6], m->tap[3] * frication);
	o[S7].top[in] = TF(TF(o[S6].top[out] * d) + TF(m->tap[4] * frication));
	o[S6].bottom[in] = TF(o[S7].bottom[out] * d);
	for (int i = S7, j = 4, k = 5; i < S10; ++i, ++j, ++k) {
		propagate_junction(m, &o[i], m->oro_k[j], &o[i + 1], m->tap[k] * frication);
	}
	/* the reflection low-pass belongs to the tube's feedback loop: float too; the radiation filters (outside it) stay double */
	{
		reflection_filter* f = &m->mouth_refl;
		const real y = TF(TF(TF(f->b0) * TF(TF(m->oro_k[7]) * o[S10].top[out])) - TF(TF(f->a1) * f->y1));
		f->y1 = y;
		o[S10].bottom[in] = TF(d * y);
	}
	real output = radiation_run(&m->mouth_rad, (RC(1.0) + m->oro_k[7]) * o[S10].top[out]);
	for (int i = N1; i < N6; ++i) {
		const real delta = TF(TF(m->nasal_k[i]) * TF(n[i].top[out] - n[i + 1].bottom[out]));
		n[i + 1].top[in] = TF(TF(n[i].top[out] + delta) * d);
		n[i].bottom[in] = TF(TF(n[i + 1].bottom[out] + delta) * d);
	}
	{
		reflection_filter* f = &m->nose_refl;
		const real y = TF(TF(TF(f->b0) * TF(TF(m->nasal_k[N6]) * n[N6].top[out])) - TF(TF(f->a1) * f->y1));
		f->y1 = y;
		n[N6].bottom[in] = TF(d * y);
	}
	output += radiation_run(&m->nose_rad, (RC(1.0) + m->nasal_k[N6]) * n[N6].top[out]);
	return output;
}
'''


def build(workdir):
    body = open(os.path.join(ORACLE, "vtm_oracle_body.inc")).read()
    a = body.index("static void propagate_junction(")
    b = body.index("/* vocalTract, VocalTractModel0.h:565-661")
    c = body.index("static real model_vocal_tract(")
    d = body.index("/* Simple copy between sections of one region")
    body = body[:a] + "#define TF(x) ((real) (float) (x))\n" + JUNCTION + "\n" + body[b:c] + TRACT + "\n" + body[d:]
    for name in ("vtm_oracle.h", "vtm_oracle.c", "vtm_oracle_f32.c", "vtm_oracle_f64.c"):
        open(os.path.join(workdir, name), "w").write(open(os.path.join(ORACLE, name)).read())
    open(os.path.join(workdir, "vtm_oracle_body.inc"), "w").write(body)
    so = os.path.join(workdir, "libstudy.so")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(workdir, "vtm_oracle.c"),
                    os.path.join(workdir, "vtm_oracle_f32.c"), os.path.join(workdir, "vtm_oracle_f64.c"), "-lm"], check=True)
    return so


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 7500
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    with tempfile.TemporaryDirectory() as wd:
        so = build(wd)
        ref_lib = oracle.lib()
        oracle._lib = None
        saved = oracle.LIB_PATH
        oracle.LIB_PATH = so
        study_lib = oracle.lib()
        oracle.LIB_PATH = saved
        worst = 0.0
        for delay in (1, 2):
            cfg = oracle.male_config(44100.0, delay)
            for i in range(count):
                tr = tracks.random_track(frames, 9000 + i, consonant_heavy=bool(i & 1))
                oracle._lib = ref_lib
                ref = oracle.synthesize(cfg, tr)
                oracle._lib = study_lib
                got = oracle.synthesize(cfg, tr)
                peak = np.abs(ref).max()
                err = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max() / peak
                worst = max(worst, err)
                print("SectionDelay %d track %d (%d frames%s): max |float-tube - double| = %.3e of peak" % (delay, i, frames, ", consonant-heavy" if i & 1 else "", err), flush=True)
        oracle._lib = ref_lib
        print("worst: %.3e of peak (north_star's tolerance against model 0: 1e-5)" % worst)


if __name__ == "__main__":
    main()
