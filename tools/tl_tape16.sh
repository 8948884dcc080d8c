cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 0 1; do
  export CS_TAPE16=$t
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_t$t -- python3 tests/diagnostics/gpu_train_timeline.py --width baseline32 --batch 32 --precision bf16 > gpurun_out/tl_t$t.log 2>&1
  f=$(find gpurun_out/tl_t$t -name "*kernel_trace.csv" | head -1)
  python3 tools/kernel_timeline.py $f --list > gpurun_out/tl_tape16_$t.txt 2>&1
  rm -rf gpurun_out/tl_t$t
  tail -1 gpurun_out/tl_t$t.log
done
