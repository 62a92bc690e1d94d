# A/B of the packed and the 32-bit flavour of the fused 4:2:0 kernel
for v in "" 1; do echo "MIJPEG_NO_F420P=$v"; MIJPEG_NO_F420P=$v python bench.py --no-cpu-baseline --no-end-to-end --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], d['ms_per_step'], d['roofline']['frac'])"; done
