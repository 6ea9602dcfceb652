O=gpurun_out/prof_r06; mkdir -p $O
timeout 900 python3 bench.py 2>/dev/null < /dev/null | tail -1 > $O/bench_default.json
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1 > $O/bench_k20.json
python3 -c "import json; d=json.load(open('$O/bench_k20.json')); print(d['value'], d['ms_per_step'], d['roofline']['counters_from']['file'], d['library'])"
