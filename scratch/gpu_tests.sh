#!/bin/bash
# GPU box: -m gpu tests (default: the whole suite, stop at the first failure), output to gpurun_out/<tag>_tests.log
#   bash scratch/gpu_tests.sh <tag> [pytest args...]
set -o pipefail
tag=${1:-r02}; shift
args=${*:-tests -x}
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest $args -m gpu -q --timeout 400 --timeout-method thread -s > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -8 gpurun_out/${tag}_tests.log
grep "\[parity\]" gpurun_out/${tag}_tests.log > gpurun_out/${tag}_parity_report.txt || true
exit $rc
