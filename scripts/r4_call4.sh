mkdir -p gpurun_out/r4c4
timeout 60 scripts/micro/bin/mfma_f64_4x4_layout 2>&1 | tee gpurun_out/r4c4/mfma_f64_4x4_layout.txt
