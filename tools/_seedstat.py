import sys
sys.path.insert(0, '/root/repo')
import torch
from platypus_amd import synth
from platypus_amd.engine import Engine
eng = Engine(0)
for name, hb in (("cfg2", synth.config2(10000, seed=2002)), ("cfg5", synth.config5(100, 100))):
    db = eng.upload(hb)
    st = eng.align(db, want_stats=True)
    v = st.n_seed_fallback
    print(name, "haps", hb.n_haps, "pairs", st.n_pairs, "aligned", st.n_pairs_aligned, "blocks_needing_index", v >> 40, "pairs_unproven_after_A", (v >> 20) & 0xFFFFF, "fallback pairs", v & 0xFFFFF)
