import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from platypus_amd import synth
from platypus_amd.engine import Engine
eng = Engine(0)
for depth, lq, err in ((1, 0.0, 1e-3), (1, 0.5, 1e-2), (1, 0.8, 2e-2), (0.5, 0.5, 1e-2), (0.5, 0.9, 3e-2)):
    hb = synth.config5_weak_evidence(200, 100, depth=depth, lowq_frac=lq, err=err)
    db = eng.upload(hb); eng.call_windows(db, want_stats=False); eng.em(db, 100, 0); eng.synchronize()
    it = db.em_iters.cpu().numpy()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.em(db, 100, 0); e1.record(); torch.cuda.synchronize()
    print("depth", depth, lq, err, "iters mean", it.mean(), "max", it.max(), "hit cap", (it>=100).mean(), "em ms", e0.elapsed_time(e1), "pairs", hb.n_pairs)
