# slide time vs internal window streams, per precision (python bench.py ... prints one JSON line; ms_per_step is the slide)
for prec in bf16 fp8; do
  for st in 1 2 3 4; do
    echo -n "precision=$prec streams=$st: "
    python3 bench.py --precision $prec --streams $st --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
