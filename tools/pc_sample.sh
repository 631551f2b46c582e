# PC sampling of the config-2 step (rocprofv3 beta feature): which instructions of k_seed / k_dp_jobs the waves sit on.
# usage: gpurun --timeout 900 -- 'bash tools/pc_sample.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pcs; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras --batches 2 --streams 1"
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 65536 --kernel-trace --output-format csv -d /tmp/pcs_st -- $B > $O/stochastic.log 2>&1
echo "stochastic rc=$?" >> $O/status.txt
ls -la /tmp/pcs_st/* 2>/dev/null | head >> $O/status.txt
if ! ls /tmp/pcs_st/*/*pc_sampling*.csv > /dev/null 2>&1; then
  timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d /tmp/pcs_ht -- $B > $O/host_trap.log 2>&1
  echo "host_trap rc=$?" >> $O/status.txt
  ls -la /tmp/pcs_ht/* 2>/dev/null | head >> $O/status.txt
fi
python - <<'PY' > $O/summary.txt 2>&1
import csv, glob, collections, sys
files = glob.glob('/tmp/pcs_*/*/*pc_sampling*.csv')
print(files)
kt = glob.glob('/tmp/pcs_*/*/*kernel_trace.csv')
disp = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        disp[r.get('Dispatch_Id')] = r.get('Kernel_Name', '').split('(')[0]
for f in files:
    rd = csv.DictReader(open(f))
    print(f, rd.fieldnames)
    per = collections.defaultdict(collections.Counter)
    tot = collections.Counter()
    extra = collections.defaultdict(collections.Counter)
    for r in rd:
        k = disp.get(r.get('Dispatch_Id'), '?')
        ins = r.get('Instruction', '?')
        cm = r.get('Instruction_Comment', '')
        per[k][(ins, cm)] += 1
        tot[k] += 1
        for col in ('Wave_Issued_Instruction', 'Instruction_Type', 'Stall_Reason', 'Wave_Count'):
            if col in r: extra[k][(col, r[col])] += 1
    for k, n in tot.most_common(6):
        print('\n==', k, n, 'samples')
        for (col, v), c in sorted(extra[k].items(), key=lambda x: -x[1])[:24]: print('   %-28s %-24s %6.2f%%' % (col, v, 100.0 * c / n))
        for (ins, cm), c in per[k].most_common(70): print('  %6.2f%%  %-70s %s' % (100.0 * c / n, ins[:70], cm[-60:]))
PY
head -c 3000 $O/summary.txt
