#!/bin/bash
# round 3, call j: the whole GPU suite on the current build, then smoke() and a short bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -x -v --timeout 600 --timeout-method thread -m gpu > gpurun_out/r03_j_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r03_j_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
