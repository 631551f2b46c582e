"""Per-stage HIP-event times of plat_align_window_batch + plat_genotype_window_batch (not a bench line).

    python tools/stage_times.py [config2|config5|long] [calc_flank_score]
"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from platypus_amd import synth
from platypus_amd.engine import Engine

which = sys.argv[1] if len(sys.argv) > 1 else "config2"
flank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = Engine(0)
hb = {"config2": lambda: synth.config2(10000, seed=2002), "config5": lambda: synth.config5(200, 100),
      "long": lambda: synth.make_snp_windows(5000, 11, read_len=250, depth=30)}[which]()
db = eng.upload(hb)
eng.profile_enable(1)
for i in range(6):
    eng.align(db, want_stats=False, calc_flank_score=flank)
    eng.genotype(db)
    eng.synchronize()
    p = eng.profile_last()
print("%s calc_flank_score=%d pairs=%d: prepare %.3f seed %.3f dp %.3f finalize %.3f genotype %.3f ms" % (
    which, flank, hb.n_pairs, p.ms_prepare, p.ms_seed, p.ms_dp, p.ms_finalize, p.ms_genotype))
