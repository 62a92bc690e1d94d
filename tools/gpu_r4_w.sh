#!/bin/bash
# Round 4, final visit: profiler passes, the bench line, every layout's kernel, a short random campaign with the new cases
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4w; export TMPDIR=/tmp
O=gpurun_out/r4w
bash tools/gpu_profile.sh r04 > $O/profile.log 2>&1; tail -3 $O/profile.log
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --no-xt --workload headline > $O/bench_profile_visit.json 2> $O/bench_profile_visit.err; tail -c 300 $O/bench_profile_visit.json; echo
echo "== bench"; ( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 200 $O/bench.json; echo
echo "== layouts"; LAYOUTS=420,444,422,440,411,gray,cmyk,3x1,1x4,lumasub,3x3,420_12,444_12 timeout 900 python tools/layout_bench.py 2>&1 | grep "ms/launch" | tee $O/layouts.txt | cut -c1-160
FLAGS=1 LAYOUTS=444 timeout 300 python tools/layout_bench.py 2>&1 | grep "ms/launch" | sed 's/^/(RGB as such) /' | tee -a $O/layouts.txt | cut -c1-160
echo "== campaign"; N=400 timeout 600 python tools/random_campaign.py > $O/random_campaign.txt 2>&1; tail -6 $O/random_campaign.txt | cut -c1-300
