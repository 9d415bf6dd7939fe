#!/bin/bash
# like ab_roles.sh, the double-precision workloads only
tag=$1; shift
out=gpurun_out/${tag}_ab.txt
: > $out
for v in "$@"; do
  if [ "$v" = default ]; then unset GVTM_DIAG_LIBRARY; else export GVTM_DIAG_LIBRARY=gama_tts_amd/lib_variants/libgama_vtm_$v.so; fi
  for cfg in "4096 500 1 1" "4096 500 0 1" "4096 1000 1 2" "4096 1000 0 2"; do
    echo "== $v  $cfg" >> $out
    python3 tests/tools/role_cycles.py $cfg >> $out 2>/dev/null
  done
done
grep -E "^==|kernel" $out
