mkdir -p gpurun_out/r4c2
timeout 120 scripts/micro/bin/chol_pair_prof > gpurun_out/r4c2/chol_pair_prof.txt 2>&1
cat gpurun_out/r4c2/chol_pair_prof.txt
timeout 300 python scripts/chol_time.py 760 2>&1 | tee gpurun_out/r4c2/chol_time.txt
timeout 200 bash scripts/micro/mfma_f64_rate.sh 1.2 > gpurun_out/r4c2/mfma_f64_rate.txt 2>&1
grep -v hold gpurun_out/r4c2/mfma_f64_rate.txt | head -60
