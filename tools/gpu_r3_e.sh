#!/bin/bash
# Round 3, final visit: smoke, the whole GPU suite, the bench line as the driver runs it, the rocprofv3 passes for profiles/r03,
# the per-layout table, the Huffman kernel alone, the no-DRI timeline
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r3e; export TMPDIR=/tmp
O=gpurun_out/r3e
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
echo "== bench"; ( time timeout 1000 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 600 $O/bench.json; tail -3 $O/bench.err
echo "== profile"; bash tools/gpu_profile.sh r03 > $O/profile.log 2>&1; tail -5 $O/profile.log
echo "== layouts"; LAYOUTS=420,444,422,440,411,gray,cmyk,3x1,1x4,lumasub,3x3,420_12,444_12,422_12,gray_12 timeout 400 python tools/layout_bench.py > $O/layouts.txt 2>&1; W=7678 LAYOUTS=420,444,gray,cmyk timeout 300 python tools/layout_bench.py 2>&1 | grep "ms/launch" | sed 's/^/W=7678 /' >> $O/layouts.txt; grep -c "ms/launch" $O/layouts.txt
echo "== huffman"; bash tools/gpu_huff_time.sh > $O/huffman.txt 2>&1; cat $O/huffman.txt
echo "== no-DRI timeline"; bash tools/nodri_timeline.sh > $O/nodri_timeline.txt 2>&1; head -3 $O/nodri_timeline.txt
