#!/bin/bash
# Round profiles: kernel-trace statistics and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no other
# tracing) of `bench.py --config <cfg>`.   tools/profile_round.sh <cfg> <tag>   -> gpurun_out/prof_<tag>/
# (run on the GPU box; rocprofv3 gets python3 directly after `--`)
set -e
CFG=$1; TAG=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python3 $R/bench.py --config $CFG --steps 20 --warmup 5 --no-forward-only --no-cpu-baseline --no-others --soak 0 > $OUT/stats_bench.json 2> $OUT/stats.err
echo "stats done" 
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- python3 $R/bench.py --config $CFG --steps 4 --warmup 2 --no-forward-only --no-cpu-baseline --no-others --soak 0 > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- python3 $R/bench.py --config $CFG --steps 4 --warmup 2 --no-forward-only --no-cpu-baseline --no-others --soak 0 > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
echo "write done"
cd $R
python3 tools/make_pmc_json.py $(ls $OUT/pmc_fetch/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*counter_collection.csv | head -1) $OUT/pmc_traffic.json "python bench.py --config $CFG --steps 4 --warmup 2" > $OUT/pmc_traffic.txt
# keep the merged result small: the raw counter CSVs are hundreds of MB
rm -rf $OUT/pmc_fetch/*counter_collection.csv $OUT/pmc_write/*counter_collection.csv $OUT/stats/*kernel_trace.csv
ls -la $OUT $OUT/stats
