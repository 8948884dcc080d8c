for f in 0 1; do
  for cfg in "--width baseline32 --batch 32 --precision bf16" "--width reference16 --batch 8 --precision bf16"; do
    echo -n "CS_WGRAD_ALDS=$f  "; CS_WGRAD_ALDS=$f python3 tests/diagnostics/gpu_train_timeline.py $cfg --steps 50 2>&1 | tail -1
  done
done
