"""Why pairs still reach the DP: PLAT_SEED_DEBUG=512 makes k_seed count, for every pair that leaves a DP job, the first test of the
ungapped-alignment proof it failed (printed by the library on stderr after each synchronous plat_align_window_batch):
index 1 not exactly one candidate diagonal, 2 the candidate is not the mapping position, 3 proven by hypothesis B only, 5 more
than UNG_KMAX mismatches, 8 not eligible (haplotype N, read not plain ACGT, diagonal < 8, read < 32 bp), 10 a path that never
touches d*, 11 an excursion before a mismatch, 12 after a mismatch, 13 around one mismatch, 14 around several.
usage (GPU box): python tools/dp_reasons.py"""
import os, sys
sys.path.insert(0, "/root/repo")
os.environ["PLAT_SEED_DEBUG"] = "512"
from platypus_amd import synth
from platypus_amd.engine import Engine
eng = Engine(0)
for name, hb in (("config2", synth.config2(10000)), ("hard", synth.config2_hard(10000))):
    db = eng.upload(hb)
    st = eng.align(db)
    eng.synchronize()
    print(name, "pairs", st.n_pairs, "dp ref", st.n_dp_reference, "launched", st.n_dp_launched, "slow", st.n_seed_fallback, file=sys.stderr)
