#!/bin/bash
# JPEG XT profile C: parity tests + throughput of the fused kernel against the three-kernel path
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd $ROOT
timeout 900 python -m pytest tests -q -x -m gpu -k "xt or 12bit" 2>&1 | tail -5
timeout 300 python tools/xt_bench.py 2>&1 | tail -3
MIJPEG_NO_FUSEDXT=1 timeout 300 python tools/xt_bench.py 2>&1 | tail -3
