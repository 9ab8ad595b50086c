#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
python tools/generic_probe.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o gp -- python $GRAFT_REPO_ROOT/tools/generic_probe.py > /tmp/gp.log 2>&1
find /tmp/gp -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r05/generic_kernel_stats.csv \;
head -20 $GRAFT_REPO_ROOT/gpurun_out/r05/generic_kernel_stats.csv
