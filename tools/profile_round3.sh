#!/bin/bash
# Round-3 profile evidence for profiles/: kernel-trace stats of config 2 (1 and 3 batches in flight), configs 3, 4, 5, and separate PMC
# passes (FETCH_SIZE | WRITE_SIZE | SQ counters; never combined with a trace domain other than --kernel-trace).
# Run on the GPU box:  gpurun -- 'bash tools/profile_round3.sh [part ...]'   parts: c2 c3 c4 c5 pmc2 pmc3 (default: all)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03p
mkdir -p $O
PARTS=${@:-c2 c3 c4 c5 pmc2 pmc3}
B2="--steps 12 --warmup 2 --no-cpu-baseline --no-extras --batches 3"
prof() { d=$1; shift; rm -rf $O/$d; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; tail -1 $O/$d.log > $O/$d.json; }
pmc() { d=$1; c=$2; shift 2; rm -rf $O/$d; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; }
for p in $PARTS; do case $p in
  c2) prof stats1 python $R/bench.py $B2 --streams 1; prof stats3 python $R/bench.py $B2 ;;
  c3) prof stats_c3 python $R/bench.py --config 3 --regions 2000 --steps 5 --no-extras; prof stats_c3e python $R/bench.py --config 3 --regions 2000 --steps 1 ;;
  c4) prof stats_c4 python $R/bench.py --config 4 --regions 256 --steps 1 ;;
  c5) prof stats_c5 python $R/bench.py --config 5 --windows 200 --steps 10 --warmup 2 ;;
  pmc2) for c in FETCH_SIZE WRITE_SIZE; do pmc pmc_$c $c python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --batches 2 --streams 1; done
        pmc pmc_SQ "SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES" python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --batches 2 --streams 1 ;;
  pmc3) for c in FETCH_SIZE WRITE_SIZE; do pmc pmc3_$c $c python $R/bench.py --config 3 --regions 2000 --steps 2 --no-extras; done
        pmc pmc3_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" python $R/bench.py --config 3 --regions 2000 --steps 2 --no-extras
        pmc pmc3_WAIT "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" python $R/bench.py --config 3 --regions 2000 --steps 2 --no-extras ;;
esac; done
cd $R
python tools/profile_round3_summary.py $O
# (gpurun brings back gpurun_out/ only: the summaries travel in it; copy them to profiles/ afterwards)
mkdir -p $O/out && cp profiles/r03_* profiles/dp_traffic.json $O/out/ 2>/dev/null
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -size +1M -delete
