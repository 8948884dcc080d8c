for sd in 1 2; do
  for cfg in "--width reference16 --batch 8 --precision fp32" "--width baseline32 --batch 32 --precision bf16" "--width baseline32 --batch 32 --precision fp32"; do
    echo -n "CS_BWD_SIDE=$sd  "; CS_BWD_SIDE=$sd python3 tests/diagnostics/gpu_train_timeline.py $cfg --steps 50 2>&1 | tail -1
  done
done
