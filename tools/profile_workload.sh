#!/bin/bash
# One bench workload, four passes on the GPU box (run through gpurun from the repo root):
#   tools/profile_workload.sh <tag> <name> "<probe env, e.g. N=1000000 MAP=site>" <bench.py args...>
# leaves gpurun_out/<tag>_<name>_{bench.json,stats/,stats.log,pmc_fetch/,pmc_fetch.log,pmc_write/,pmc_write.log};
# `python tools/collect_profiles.py <tag> <name>` (in the build container) turns them into profiles/<tag>_<name>_*.
set -u
tag=$1; name=$2; penv=$3; shift 3
o=gpurun_out/${tag}_${name}
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
o=$R/$o
timeout 900 python bench.py "$@" > ${o}_bench.json 2> ${o}_bench.err
# (rocprofv3 runs from /tmp; --output-format csv: the default is a rocpd database; PMC passes carry --kernel-trace only)
(cd /tmp && timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d ${o}_stats -o s -- python $R/bench.py "$@" --cpu-seconds 0 --sustain-seconds 0 > ${o}_stats.log 2>&1)
(cd /tmp && env $penv timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d ${o}_pmc_fetch -o f -- python $R/tools/pmc_probe.py > ${o}_pmc_fetch.log 2>&1)
(cd /tmp && env $penv timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d ${o}_pmc_write -o w -- python $R/tools/pmc_probe.py > ${o}_pmc_write.log 2>&1)
# keep the merge small: the summaries only
find ${o}_stats ${o}_pmc_fetch ${o}_pmc_write -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
tail -c 600 ${o}_bench.json; echo; grep -h '^{' ${o}_pmc_fetch.log | tail -1 | cut -c1-300
