#!/bin/bash
# usage: tools/sweep_env.sh "VAR=a VAR2=b" ...   -- runs bench.py once per environment setting, prints frames/s
for cfg in "$@"; do
  out=$(env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null)
  echo "$cfg -> $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "frames/s", round(d["roofline"]["gpu_ms_per_step"],3), "ms/step")')"
done
