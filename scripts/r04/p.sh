#!/bin/bash
# round 4, GPU call P: whole -m gpu suite + smoke on the final kernels
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
