# round 3: where config 4's worker time goes (stage by stage, host vs waiting), and config 2 with more batches in flight
pick='import json,sys
l=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], round(l["ms_per_step"],4), "ms/step", round(l["value"]), "GCUPS")'
for s in 4 6; do python bench.py --no-cpu-baseline --no-extras --streams $s 2>/dev/null | python -c "$pick" streams$s; done
PLAT_CALLER_TRACE=1 python bench.py --config 4 --regions 3875 --steps 1 2> gpurun_out/r3d_c4.err | python -c '
import json,sys
l=json.loads(sys.stdin.read().strip().split("\n")[-1])
print({k: (round(v,6) if isinstance(v,float) else v) for k,v in l.items() if k not in ("config","roofline","cpu_baseline")})'
grep plat_caller gpurun_out/r3d_c4.err
