mkdir -p gpurun_out/r4c3
for a in 1 2 4 3 5 6; do echo "ABLATE=$a (bit0 no micro, bit1 no update, bit2 no inverse)"; timeout 60 scripts/micro/bin/chol_pair_ab$a | grep "lean has1=1"; done 2>&1 | tee gpurun_out/r4c3/ablate.txt
