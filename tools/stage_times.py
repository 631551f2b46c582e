"""Per-stage HIP-event times of plat_align_window_batch on config 2 (optionally with --calculateFlankScore=1)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from platypus_amd import synth
from platypus_amd.engine import Engine

flank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng = Engine(0)
hb = synth.config2(10000, seed=2002)
db = eng.upload(hb)
eng.profile_enable(1)
for i in range(6):
    eng.align(db, want_stats=False, calc_flank_score=flank)
    eng.synchronize()
    p = eng.profile_last()
print("calc_flank_score=%d prepare %.3f seed %.3f dp %.3f finalize %.3f ms" % (flank, p.ms_prepare, p.ms_seed, p.ms_dp, p.ms_finalize))
