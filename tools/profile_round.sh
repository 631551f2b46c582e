#!/bin/bash
# One round of profile evidence for profiles/: kernel-trace stats (1 and 3 batches in flight) and three separate PMC passes.
# Run on the GPU box:  gpurun -- 'bash tools/profile_round.sh'
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r01c
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --streams 1 > $O/stats1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/stats3.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_SQ -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > $O/pmc_SQ.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ > $O/pmc_summary.txt 2>&1
python tools/profile_round_summary.py $O
find $O -name "*.csv" -size +2M -delete
head -12 profiles/r01_kernel_stats.txt; cat profiles/dp_traffic.json
