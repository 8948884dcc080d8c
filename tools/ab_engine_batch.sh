# slide time vs windows per engine call (two internal streams), per precision
for prec in fp32 bf16; do
  for eb in 0 241 161 121 81 61; do
    echo -n "precision=$prec engine_batch=$eb: "
    python3 bench.py --precision $prec --engine-batch $eb --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
