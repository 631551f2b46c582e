"""Bigger / different shapes of the window batch: does the path hold up, and at what rate (not a bench line)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from platypus_amd import synth
from platypus_amd.engine import Engine

eng = Engine(0)
for name, hb in (("config2 x4 (40000 windows)", synth.config2(40000, seed=7)),
                 ("250 bp reads, 5000 windows", synth.make_snp_windows(5000, 11, read_len=250, depth=30)),
                 ("100 bp reads, 20000 windows", synth.make_snp_windows(20000, 12, read_len=100, depth=30))):
    db = eng.upload(hb)
    st = eng.call_windows(db, want_stats=True); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.call_windows(db, want_stats=False, asynchronous=True)
    eng.synchronize()
    t = (time.perf_counter() - t0) / 5
    print("%-30s pairs %9d  ms/step %7.3f  ref-GCUPS %7.1f  windows/s %9.0f  dp launched/ref %.3f  mem %.2f GB" % (
        name, st.n_pairs, 1e3 * t, st.cells_reference / t / 1e9, hb.n_windows / t, st.n_dp_launched / max(1, st.n_dp_reference),
        torch.cuda.max_memory_allocated() / 1e9))
    del db
