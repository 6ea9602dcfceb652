#!/bin/bash
# What one rank runs under the STRONG split of the metric's batch (256 poses over 8 / 4 / 2 / 1 ranks = 32 / 64 / 128 / 256 poses per launch),
# measured on one MI355X with bench.py's own protocol: the launch is latency-bound, so fewer poses per rank barely shorten it.
for b in 32 64 128 256; do
  python3 bench.py --batch $b --steps 20 --warmup 5 --workload metric --no-cpu-baseline --no-head --steady-batch 0 2>/dev/null | tail -1 > /tmp/strong_$b.json
  python3 - "$b" <<'PY'
import json, sys
b = sys.argv[1]
d = json.load(open(f"/tmp/strong_{b}.json"))
it = d["roofline"]["lm_iterations"]
print(f"B={int(b):4d} per launch: {d['ms_per_step'] * 1e3:6.2f} us per step = {d['value'] / 1e6:6.2f} M poses/s per GPU; LM iterations mean {it['mean']:.2f} max {it['max']}; "
      f"x{256 // int(b)} ranks of a strong split -> {256 / (d['ms_per_step'] * 1e-3) / 1e6:6.2f} M poses/s for the global batch of 256")
PY
done
