set -u
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r03 0b84899 > gpurun_out/profile_round.log 2>&1 < /dev/null
O=gpurun_out/prof_r03
timeout 600 python3 bench.py 2>/dev/null < /dev/null | tail -1 > $O/bench_default.json
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1 > $O/bench_k20.json
timeout 300 python3 bench_head.py 2>/dev/null < /dev/null | tail -1 > $O/bench_head_f32.json
timeout 300 python3 bench_head.py --dtype bf16 2>/dev/null < /dev/null | tail -1 > $O/bench_head_bf16.json
timeout 300 python3 bench_head.py --dtype f16 2>/dev/null < /dev/null | tail -1 > $O/bench_head_f16.json
bash scripts/profile_test_time.sh r03 > gpurun_out/profile_test_time.log 2>&1 < /dev/null
ls -la $O | head -30
python3 -c "import json; d=json.load(open('$O/bench_k20.json')); print(d['value'], d['ms_per_step'], d['test_time']['us_per_call_replayed'], {k:(v['ms_per_step'], v['lc_pnp_lm_wide_kernel']['kernel_us'], v['lc_cov_loss_kernel']['kernel_us']) for k,v in d['dense'].items()})"
