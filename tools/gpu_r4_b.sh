#!/bin/bash
# Round 4, visit b: the whole GPU suite (DNL, SPEC boxes, damaged XT, full duplex), the host decode of config 5's -rR 4 variant
# with the chain affinity switch, the rocprofv3 passes for profiles/r04, the per-layout table, the bench line.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4b; export TMPDIR=/tmp
O=gpurun_out/r4b
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log
echo "== xt host decode"; timeout 600 python tools/xt_host_bench.py 1 16 64 > $O/xt_host.txt 2>&1; cat $O/xt_host.txt
echo "== profile"; bash tools/gpu_profile.sh r04 > $O/profile.log 2>&1; tail -30 $O/profile.log
echo "== layouts"; LAYOUTS=420,444,422,440,411,gray,cmyk,3x1,1x4,lumasub,3x3,420_12,444_12 timeout 400 python tools/layout_bench.py > $O/layouts.txt 2>&1; W=7678 LAYOUTS=420,444 timeout 300 python tools/layout_bench.py 2>&1 | grep "ms/launch" | sed 's/^/W=7678 /' >> $O/layouts.txt; grep "ms/launch" $O/layouts.txt | cut -c1-160
echo "== bench"; ( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 200 $O/bench.json
