#!/bin/bash
# Everything profiles/r03/ holds about bench.py, in one gpurun call from the repo root (the GPU box has no .git: pass the commit):
#   gpurun -- 'bash scripts/refresh_r03.sh <git sha>'      then, here:   bash scripts/collect_profiles.sh r03 <git sha>
# 1) scripts/profile_round.sh (kernel trace + stats, FETCH/WRITE/SQ counters, head, batch sweep)  2) the bench records: the default
# command, the driver's command, the head in three element types  3) scripts/profile_test_time.sh (the test-time chain)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
SHA=${1:-unknown}
bash scripts/profile_round.sh r03 "$SHA" > gpurun_out/profile_round.log 2>&1 < /dev/null
O=gpurun_out/prof_r03
timeout 600 python3 bench.py 2>/dev/null < /dev/null | tail -1 > $O/bench_default.json
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1 > $O/bench_k20.json
timeout 300 python3 bench_head.py 2>/dev/null < /dev/null | tail -1 > $O/bench_head_f32.json
timeout 300 python3 bench_head.py --dtype bf16 2>/dev/null < /dev/null | tail -1 > $O/bench_head_bf16.json
timeout 300 python3 bench_head.py --dtype f16 2>/dev/null < /dev/null | tail -1 > $O/bench_head_f16.json
bash scripts/profile_test_time.sh r03 > gpurun_out/profile_test_time.log 2>&1 < /dev/null
ls -la $O | head -30
python3 -c "import json; d=json.load(open('$O/bench_k20.json')); print(d['value'], d['ms_per_step'], d['test_time']['us_per_call_replayed'], {k:(v['ms_per_step'], v['two_launches']['ms_per_step']) for k,v in d['dense'].items()})"
