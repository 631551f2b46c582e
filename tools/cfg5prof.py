import sys; sys.path.insert(0,'/root/repo')
from platypus_amd import synth
from platypus_amd.engine import Engine
eng=Engine(0)
hb=synth.config5(200,100)
db=eng.upload(hb)
eng.call_windows(db); eng.synchronize()
eng.profile_enable(True)
for _ in range(3):
    eng.call_windows(db, want_stats=False); p=eng.profile_last()
    print("ms prepare %.3f seed %.3f dp %.3f fin %.3f geno %.3f jobs %d"%(p.ms_prepare,p.ms_seed,p.ms_dp,p.ms_finalize,p.ms_genotype,p.dp_jobs))
