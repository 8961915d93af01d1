timeout 100 python scripts/init_diag.py 2>&1 | grep "init at"
timeout 200 python scripts/nn_conv_check.py 2>&1 | grep -i "worst\|MISMATCH"
timeout 600 python -m pytest tests/test_gpu_hotpath.py -q -x -k "nn_ or depthcov or network or run_model" 2>&1 | tail -2
timeout 300 python scripts/nn_time.py --layers 2>&1 | grep -v "amdgpu\|Warn\|warn" | head -12
