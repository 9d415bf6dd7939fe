#!/bin/bash
# Everything a reviewer would run, in order: build, CPU suite, and on a GPU box the parity suite, smoke and the bench line.
#   here (no GPU):      tools/verify_all.sh
#   on an MI355X box:   gpurun --timeout 900 -- 'bash tools/verify_all.sh gpu'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -x -q -m "not gpu"
if [ "$1" = "gpu" ]; then
  python -m pytest tests -x -q -m gpu
  python -c "import __graft_entry__ as g; g.smoke()"
  python bench.py
fi
