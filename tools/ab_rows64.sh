for t in 0 100000000; do echo -n "CS_ROWS64_MIN=$t  "; CS_ROWS64_MIN=$t python3 tests/diagnostics/gpu_tiles_forward_sweep.py bf16 2>&1 | tail -1; done
