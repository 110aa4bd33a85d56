#!/bin/bash
# usage: tools/pmc_run.sh <outdir> "<counters>" <kernel_bench args...>   (run on the GPU box)
# One rocprofv3 --pmc pass (counters only, no tracing domains) of tools/kernel_bench.py.
out=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/kernel_bench.py "$@" > $out.log 2>&1
