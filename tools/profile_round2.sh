#!/bin/bash
# Round-2 profile evidence for profiles/: kernel-trace stats of config 2 (1 and 3 batches in flight), configs 3, 4 and 5, and
# separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ counters) for config 2 and config 3.
# Run on the GPU box:  gpurun -- 'bash tools/profile_round2.sh'
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02p
rm -rf $O; mkdir -p $O
B2="--steps 12 --warmup 2 --no-cpu-baseline --no-extras --batches 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python $R/bench.py $B2 --streams 1 > $O/stats1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python $R/bench.py $B2 > $O/stats3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python $R/bench.py --config 3 --regions 2000 --steps 5 > $O/stats_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python $R/bench.py --config 5 --windows 200 --steps 10 --warmup 2 > $O/stats_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python $R/bench.py --config 4 --regions 64 --steps 1 > $O/stats_c4.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --batches 2 --streams 1 > $O/pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc3_$c -- python $R/bench.py --config 3 --regions 2000 --steps 2 > $O/pmc3_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_SQ -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --batches 2 --streams 1 > $O/pmc_SQ.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc3_SQ -- python $R/bench.py --config 3 --regions 2000 --steps 2 > $O/pmc3_SQ.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ > $O/pmc_summary.txt 2>&1
python tools/pmc_summary.py $O/pmc3_FETCH_SIZE $O/pmc3_WRITE_SIZE $O/pmc3_SQ > $O/pmc3_summary.txt 2>&1
python tools/profile_round2_summary.py $O
for f in stats1 stats3 stats_c3 stats_c4 stats_c5; do tail -1 $O/$f.log > $O/$f.json; done
find $O -name "*.csv" -size +1M -delete
head -14 profiles/r02_kernel_stats.txt; head -8 profiles/r02_assemble_stats.txt; cat profiles/dp_traffic.json
