R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/seedpmc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $O/a -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 > $O/c.log 2>&1
cd $R
python tools/pmc_summary.py $O/a $O/b $O/c 2>&1 | grep "k_seed \|k_dp_jobs\|k_prep"
tail -2 $O/b.log $O/c.log
find $O -name "*.csv" -size +1M -delete
