#!/bin/bash
# Everything profiles/<round>/ holds about bench.py, in one gpurun call from the repo root (the GPU box has no .git: pass the commit):
#   gpurun -- 'bash scripts/refresh_round.sh r04 <git sha>'      then, here:   bash scripts/collect_profiles.sh r04 <git sha>
# 1) scripts/profile_round.sh (kernel trace + stats, FETCH/WRITE/SQ counters, head, batch sweep)  2) the bench records: the default
# command, the driver's command, the head in three element types  3) scripts/profile_test_time.sh (the test-time chain at the reference's
# configs)  4) scripts/profile_next.sh (bench_next.py under rocprofv3: the f-row kernels, fp32 and 16-bit maps)
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
TAG=${1:-r04}
SHA=${2:-unknown}
bash scripts/profile_round.sh "$TAG" "$SHA" > gpurun_out/profile_round.log 2>&1 < /dev/null
O=gpurun_out/prof_$TAG
# the bench line names the counter passes it prices `traffic` / `valu_issue` with (the newest profiles/<round>/pmc_traffic.json): put THIS run's passes
# where it looks before it runs, so the round's records name the round's counters (round 5's and the first of round 6's named the round before)
mkdir -p profiles/$TAG && [ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/$TAG/pmc_traffic.json
timeout 900 python3 bench.py 2>/dev/null < /dev/null | tail -1 > $O/bench_default.json
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1 > $O/bench_k20.json
timeout 300 python3 bench_head.py 2>/dev/null < /dev/null | tail -1 > $O/bench_head_f32.json
timeout 300 python3 bench_head.py --dtype bf16 2>/dev/null < /dev/null | tail -1 > $O/bench_head_bf16.json
timeout 300 python3 bench_head.py --dtype f16 2>/dev/null < /dev/null | tail -1 > $O/bench_head_f16.json
bash scripts/profile_test_time.sh "$TAG" all > gpurun_out/profile_test_time.log 2>&1 < /dev/null
timeout 300 python3 scripts/ubench/select_stamps.py > gpurun_out/test_time_$TAG/select_stamps.txt 2>/dev/null < /dev/null
bash scripts/profile_next.sh "$TAG" > gpurun_out/profile_next.log 2>&1 < /dev/null
timeout 300 python3 bench_next.py 2>/dev/null < /dev/null > gpurun_out/prof_next_$TAG/bench_next.jsonl
ls -la $O | head -30
python3 -c "import json; d=json.load(open('$O/bench_k20.json')); print(d['value'], d['ms_per_step'], {k: v['us_per_call_replayed_200'] for k, v in d['test_time'].items()}, {k:(v['ms_per_step'], v['two_launches']['ms_per_step']) for k,v in d['dense'].items()}, d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
