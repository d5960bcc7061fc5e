#!/bin/bash
# rate against the number of concurrent games per GPU (the headline workload at other slot counts): one bench line per size, condensed
# usage (on the MI355X box): bash tools/sweep_games.sh > gpurun_out/rXX_sweep.txt
for G in 256 512 1024 2048 4096 8192 16384 32768 65536; do
  python3 bench.py --games $G --steps 2 --warmup 1 --no-cpu-baseline --no-literal-configs 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(f\"games {d['config']['concurrent_games_per_gpu']:6d}  {d['value']:8.1f} games/s  {d['config']['us_per_lockstep']:7.1f} us/lock-step  end_to_end {r['end_to_end_frac']:.3f}  dominant {r['kernel'][:28]:28s} frac {r['frac']:.3f}  avg_launch {1e3 * r['avg_launch_ms']:7.1f} us\")
"
done
