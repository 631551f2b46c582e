import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from platypus_amd import synth
from platypus_amd.engine import Engine
eng = Engine(0)
hb = synth.config2(10000, seed=2002)
db = eng.upload(hb)
eng.profile_enable(1)
for i in range(6):
    try:
        eng.align(db, want_stats=False)
    except Exception as e:
        print("err", e); break
    eng.synchronize()
    p = eng.profile_last()
print(os.environ.get("PLAT_DBGSTAGE"), "prepare %.3f seed %.3f dp %.3f" % (p.ms_prepare, p.ms_seed, p.ms_dp))
