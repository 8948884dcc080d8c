for t in 0 1; do
  echo -n "CS_TAPE16=$t: "; CS_TAPE16=$t python3 tests/diagnostics/gpu_train_timeline.py --width baseline32 --batch 32 --precision bf16 --steps 50 2>&1 | tail -1
done
