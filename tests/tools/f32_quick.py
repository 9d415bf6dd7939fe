import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, gama_tts_amd as g, oracle, tracks
from gama_tts_amd import capi
for delay in (1,2,3):
    params = tracks.random_tracks(12, 60, seed0=9100+delay, consonant_heavy=True)
    plan = g.Plan(g.config_from_dict(g.read_config_file(oracle.VOICE_MALE), 44100.0, delay, capi.PRECISION_F32), 250.0, 0, diagnostics=True, rows=4)
    audio,_,_ = plan.synthesize_host(params)
    ref = oracle.synthesize_batch(oracle.male_config(44100.0, delay, float_model=1), params)
    print('delay', delay, 'bit-identical', all(np.array_equal(audio[b], ref[b]) for b in range(12)), 'sha-equal', audio.tobytes()==ref.tobytes())
