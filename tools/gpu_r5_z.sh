#!/bin/bash
# Round 5: config 5 with hidden bits end to end after the host-side work (XT tests on the device, the xt leg of the bench)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5z
timeout 900 python -m pytest tests -m gpu -q -k "xt" -x > gpurun_out/r5z/pytest_xt.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r5z/pytest_xt.log
timeout 600 python - > gpurun_out/r5z/xt_profile_c.json 2> gpurun_out/r5z/xt_profile_c.err <<'PY'
import json, torch, bench
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    r = bench.xt_profile_c(0, s, True)
print(json.dumps(r))
PY
echo "xt exit $?"; tail -3 gpurun_out/r5z/xt_profile_c.err
python - <<'PY'
import json
x = json.load(open("gpurun_out/r5z/xt_profile_c.json"))
for k in ("r12", "r12_rR4", "r12_rR4_z8"):
    e = x[k]; print(k, e["kernel"], e["kernel_ms"], e["roofline"]["frac"], e["verified"], e["entropy_decode_ms"], e["bytes_to_half_codes_in_hbm"], e["bytes_to_float32_in_host_memory"]["ms"])
PY
