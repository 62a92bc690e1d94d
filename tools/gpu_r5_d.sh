#!/bin/bash
# Round 5: alpha tests again (stripe test fixed) and the bench line with its verified fields
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5d; export TMPDIR=/tmp
O=gpurun_out/r5d
timeout 600 python -m pytest tests/test_xt_alpha.py -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 1500 $O/bench.json; echo; tail -5 $O/bench.err
