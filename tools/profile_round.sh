#!/bin/bash
# The ONE profiling / evidence script.  Everything kept under profiles/ comes from here.
#
#   gpurun --timeout 3000 -- "PLAT_COMMIT=$(git rev-parse --short HEAD) bash tools/profile_round.sh r05 [part ...]"
#
# First argument: the round tag the files under profiles/ are named with (r04 -> profiles/r04_*).
# Parts (default: all but the soaks):
#   tests   python -m pytest tests -m gpu                               -> $O/pytest_gpu.txt
#   c2      rocprofv3 --kernel-trace --stats of config 2, one batch at a time and three in flight
#   c3 c4 c5   the same for configs 3 (tiles + end to end), 4 (12288 regions, inputs resident, 128 regions per chunk as the whole-genome line), 5
#   c4solo  config 4 with ONE worker (no two kernels at once): every kernel's duration by itself, the basis bench.py's live timers and the roofline use
#   pmc4    PMC passes of config 4 (FETCH_SIZE | WRITE_SIZE | SQ counters) -> $TAG_pmc_config4.txt, profiles/wgs_profile.json (what bench.py's line quotes)
#   mapa    tools/ubench/dp_mapping_a.hip: mapping A of the DP, bit exact, against the library -> $TAG_dp_mapping_a.json
#   pmc2    PMC passes of config 2 (FETCH_SIZE | WRITE_SIZE | SQ counters; one pass per set, only --kernel-trace next to --pmc)
#   pmc3    PMC passes of the assembler (FETCH | WRITE | SQ | wait counters)
#   pmcseed three SQ passes over k_seed / k_dp_jobs / k_prep_reads (instruction mix, waits, LDS conflicts)
#   dpstall where k_dp_jobs' wave cycles go: parked on waits / issue stalls / active (SQ counters), config 2 as it runs and with every DP executed
#   nextk   kernel stats of the "next row" kernels (tools/next_kernels.py)
#   line    the default bench line, bench.py --config 3/4/5, the two-rank launches on the one GPU
#   soaks   ungapped cross-check (plain + wrap regime), native region-loop soak, assembler soak; SOAK_SECONDS each (default 400)
# gpurun only brings back gpurun_out/: the summaries are written to profiles/ on the box AND copied to $O/out; copy them from there.
TAG=${1:-r06}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}p
mkdir -p $O
PARTS=${@:-tests c2 c3 c4 c5 pmc2 pmc3 pmc4 nextk mapa line}
T=${SOAK_SECONDS:-400}
B2="--config 2 --steps 12 --warmup 2 --no-cpu-baseline --no-extras --batches 3"
cd /tmp && export TMPDIR=/tmp
prof() { d=$1; shift; rm -rf $O/$d; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; tail -1 $O/$d.log > $O/$d.json; }
pmc() { d=$1; c=$2; shift 2; rm -rf $O/$d; rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; }
for p in $PARTS; do case $p in
  tests) (cd $R && python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt) ;;
  c2) prof stats1 python $R/bench.py $B2 --streams 1; prof stats3 python $R/bench.py $B2 ;;
  c3) prof stats_c3 python $R/bench.py --config 3 --regions 2000 --steps 5 --no-extras; prof stats_c3e python $R/bench.py --config 3 --regions 2000 --steps 1 ;;
  c4) PLAT_CALLER_CHUNK=128 prof stats_c4 python $R/bench.py --config 4 --regions 12288 --steps 1 --no-cpu-baseline
      f=$(ls $O/stats_c4/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/trace_overlap.py $f 3 300 > $O/config4_overlap.json ;;
  c4solo) PLAT_CALLER_WORKERS=1 PLAT_CALLER_CHUNK=128 prof stats_c4solo python $R/bench.py --config 4 --regions 2048 --steps 1 --no-cpu-baseline ;;
  pmc4) for c in FETCH_SIZE WRITE_SIZE; do PLAT_CALLER_CHUNK=128 pmc pmc4_$c $c python $R/bench.py --config 4 --regions 12288 --steps 1 --no-cpu-baseline; done
        PLAT_CALLER_CHUNK=128 pmc pmc4_SQ "SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES" python $R/bench.py --config 4 --regions 12288 --steps 1 --no-cpu-baseline ;;
  mapa) (cd $R && hipcc --offload-arch=gfx950 -O3 -o /tmp/dp_mapping_a tools/ubench/dp_mapping_a.hip -Lplatypus_amd -lplat_mi355x && LD_LIBRARY_PATH=platypus_amd /tmp/dp_mapping_a 400000 150 | tail -1 > $O/dp_mapping_a.json; cat $O/dp_mapping_a.json) ;;
  c5) prof stats_c5 python $R/bench.py --config 5 --windows 200 --steps 10 --warmup 2 ;;
  pmc2) for c in FETCH_SIZE WRITE_SIZE; do pmc pmc_$c $c python $R/bench.py --config 2 --steps 4 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-extras --batches 2 --streams 1; done
        pmc pmc_SQ "SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES" python $R/bench.py --config 2 --steps 4 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-extras --batches 2 --streams 1 ;;
  pmc3) for c in FETCH_SIZE WRITE_SIZE; do pmc pmc3_$c $c python $R/bench.py --config 3 --regions 2000 --steps 2 --no-extras; done
        pmc pmc3_SQ "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" python $R/bench.py --config 3 --regions 2000 --steps 2 --no-extras
        pmc pmc3_WAIT "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" python $R/bench.py --config 3 --regions 2000 --steps 2 --no-extras ;;
  pmcseed) S="--config 2 --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-extras --streams 1"
        pmc seed_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" python $R/bench.py $S
        pmc seed_b "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" python $R/bench.py $S
        pmc seed_c "SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT" python $R/bench.py $S
        (cd $R && python tools/pmc_summary.py $O/seed_a $O/seed_b $O/seed_c 2>&1 | grep "plat::\|k_dp_jobs" | tee $O/pmc_seed.txt) ;;
  dpstall) S="--config 2 --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-extras --streams 1 --batches 2"
        # where k_dp_jobs' cycles go (MI355X_MICROARCH.md: WAIT_ANY = wave parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY:
        # the three are disjoint and add up to WAVE_CYCLES), on config 2 as it runs and with every reference DP executed (a full grid)
        pmc dps_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" python $R/bench.py $S
        pmc dps_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" python $R/bench.py $S
        PLAT_NO_UNGAPPED=1 PLAT_NO_EXACT=1 pmc dps_all_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" python $R/bench.py $S
        (cd $R && { echo "# rocprofv3 --kernel-trace --pmc <C> -- python bench.py $S   (MI355X; mean per launch; SQ_*_CYCLES / WAIT / ACTIVE counters in units of 4 cycles, summed over the SIMDs)"
          echo "# config 2 as it runs (0.21 M DPs per launch):"; python tools/pmc_summary.py $O/dps_a $O/dps_b 2>&1 | grep "k_dp_jobs\|k_pairs\|k_sweep"
          echo "# PLAT_NO_UNGAPPED=1 PLAT_NO_EXACT=1: every reference DP executed (1.57 M DPs per launch):"; python tools/pmc_summary.py $O/dps_all_a 2>&1 | grep "k_dp_jobs"; } > $O/dp_stalls.txt; cat $O/dp_stalls.txt) ;;
  nextk) prof nextk python $R/tools/next_kernels.py ;;
  line) (cd $R
        python bench.py > $O/bench_line.json 2> $O/bench_line.err
        for c in 3 4 5; do python bench.py --config $c > $O/bench_config$c.json 2> $O/bench_config$c.err; done
        python bench.py --gpus 2 --steps 5 --no-extras > $O/bench_2ranks.json 2> $O/bench_2ranks.err
        python bench.py --gpus 2 --config 4 --regions 4096 > $O/bench_c4_2ranks.json 2> $O/bench_c4_2ranks.err
        PLAT_CALLER_LOADERS=2 taskset -c 0,1 python bench.py --config 4 --regions 4096 --steps 3 --no-cpu-baseline > $O/bench_config4_2cpus.json 2> $O/bench_config4_2cpus.err
        tail -c 600 $O/bench_line.json) ;;
  soaks) (cd $R
        python tools/ungapped_crosscheck.py $T 70000 --bigq 2>&1 | tail -1 > $O/soak_ungapped_bigq.json
        python tools/ungapped_crosscheck.py $T 50000 2>&1 | tail -1 > $O/soak_ungapped.json
        python tools/native_soak.py $T 2>&1 | tail -3 > $O/soak_native.json
        python tests/soak/assembler_soak.py $T 2>&1 | tail -1 > $O/soak_assembler.json
        tail -n 3 $O/soak_*.json) ;;
esac; done
cd $R
python tools/profile_round_summary.py $O $TAG
mkdir -p $O/out && cp profiles/${TAG}_* profiles/dp_traffic.json profiles/wgs_profile.json $O/out/ 2>/dev/null
[ -s $O/pmc_seed.txt ] && { echo "# rocprofv3 --kernel-trace --pmc <C> (three passes: instruction mix | waits and busy | LDS) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --streams 1; mean per launch (SQ_*_CYCLES / ACTIVE / WAIT counters in units of 4 cycles)"; cat $O/pmc_seed.txt; } > $O/out/${TAG}_pmc_kernels.txt
[ -s $O/dp_stalls.txt ] && cp $O/dp_stalls.txt $O/out/${TAG}_dp_stalls.txt
for f in $O/bench_*.json $O/soak_*.json $O/dp_mapping_a.json $O/config4_overlap.json; do [ -s "$f" ] && grep "^[{[]" $f | tail -1 > $O/out/${TAG}_$(basename $f); done     # the lines / soaks of THIS run (gloo writes to stdout too: the JSON line only)
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -size +1M -delete
