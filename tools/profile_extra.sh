#!/bin/bash
# MfmaUtil per kernel (its own --pmc pass) of `bench.py --config <cfg>`:   tools/profile_extra.sh <cfg> <tag>
set -e
CFG=$1; TAG=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/pmc_mfma -o $TAG -- python3 $R/bench.py --config $CFG --steps 4 --warmup 2 --no-forward-only --no-cpu-baseline --no-others --soak 0 > $OUT/pmc_mfma_bench.json 2> $OUT/pmc_mfma.err
cd $R
python3 tools/pmc_summary.py $(ls $OUT/pmc_mfma/*counter_collection.csv | head -1) 40 > $OUT/mfma_util.txt
rm -rf $OUT/pmc_mfma/*counter_collection.csv
cat $OUT/mfma_util.txt | head -30
