import sys; sys.path.insert(0,'/root/repo')
from platypus_amd import synth
from platypus_amd.engine import Engine
import time
eng=Engine(0)
hb=synth.config2(2000)
db=eng.upload(hb)
st=eng.call_windows(db); eng.synchronize()
print("pairs",st.n_pairs,"aligned",st.n_pairs_aligned,"dp_launched",st.n_dp_launched,"dp_ref",st.n_dp_reference,"seed_fallback",st.n_seed_fallback)
eng.profile_enable(True)
for _ in range(3):
    eng.call_windows(db, want_stats=False); p=eng.profile_last()
    print("ms prepare %.3f seed %.3f dp %.3f fin %.3f geno %.3f"%(p.ms_prepare,p.ms_seed,p.ms_dp,p.ms_finalize,p.ms_genotype))
