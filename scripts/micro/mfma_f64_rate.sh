#!/bin/bash
# Run on the GPU box: the float64 matrix-rate probe in its long mode with rocm-smi sampling sclk / power beside it.
#   bash scripts/micro/mfma_f64_rate.sh [seconds per configuration] > gpurun_out/mfma_f64_rate.txt
SEC=${1:-1.5}
BIN=scripts/micro/bin/mfma_f64_rate
( while true; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed 's/  */ /g'
    echo
    sleep 0.4
  done ) > /tmp/mfma_smi.log &
SMI=$!
$BIN $SEC
kill $SMI 2>/dev/null
echo "---- rocm-smi samples during the run (sclk / power), every ~0.7 s ----"
cat /tmp/mfma_smi.log
