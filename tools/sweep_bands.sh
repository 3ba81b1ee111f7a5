for b in 1 2 3 4 6 8; do
  export DCMT_BANDS=$b
  echo "bands $b: $(python bench.py --total-frames 128 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), {k: round(v['ms'],4) if isinstance(v,dict) else round(v,4) for k,v in d['roofline']['per_kernel'].items()})")"
done
