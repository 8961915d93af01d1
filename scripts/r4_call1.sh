set -x
mkdir -p gpurun_out/r4c1
timeout 120 scripts/micro/bin/chol_pair > gpurun_out/r4c1/chol_pair.txt 2>&1
cat gpurun_out/r4c1/chol_pair.txt
timeout 600 python -m pytest tests/test_gpu_hotpath.py -q -x -k "cholesky" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_r2.py -q -x -k "cholesky" 2>&1 | tail -5
for L in 1 0; do echo "COMO_CHOL_LEAN=$L"; COMO_CHOL_LEAN=$L timeout 300 python scripts/chol_time.py 200 760 1240 2680; done 2>&1 | tee gpurun_out/r4c1/chol_time.txt
timeout 200 bash scripts/micro/mfma_f64_rate.sh 1.2 > gpurun_out/r4c1/mfma_f64_rate.txt 2>&1
tail -50 gpurun_out/r4c1/mfma_f64_rate.txt
