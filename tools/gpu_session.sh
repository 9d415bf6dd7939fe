for i in 1 2 3; do
python3 tests/tools/bench_aux.py 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('new tracks ms', d['tracks']['ms'])"
GVTM_LIBRARY=gama_tts_amd/lib_variants/libgama_vtm_oldtracks.so python3 tests/tools/bench_aux.py 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('old tracks ms', d['tracks']['ms'])"
done
