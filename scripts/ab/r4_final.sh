cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4_final
timeout 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4_final/tests.txt
bash scripts/collect_profiles_r4b.sh 2>&1 | tail -30
