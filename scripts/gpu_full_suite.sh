#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/ab/pytest.log 2>&1; tail -8 gpurun_out/ab/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
