"""Cross-check of k_seed's ungapped-alignment shortcut against the DP it replaces: the same batches aligned with the shortcut and
with PLAT_NO_UNGAPPED=1 (every pair through the DP) must give identical scores.  Stress batches (tests/test_gpu_parity.py::
_adversarial_batch: repeats, cheap gaps, mismatches at the read ends, quality minima down to 1) with fresh seeds until the time
budget is used, then BASELINE config 2 and a config-5 sample.   usage: python tools/ungapped_crosscheck.py [seconds] [first seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from platypus_amd import synth                      # noqa: E402
from platypus_amd.engine import Engine              # noqa: E402
from test_gpu_parity import _adversarial_batch      # noqa: E402


def both(eng, hb):
    out = {}
    for mode in ("0", "1"):
        os.environ["PLAT_NO_UNGAPPED"] = mode
        db = eng.upload(hb)
        st = eng.align(db, want_stats=True)
        eng.synchronize()
        out[mode] = (db.score.cpu().numpy()[:hb.n_pairs].copy(), int(st.n_dp_launched))
    os.environ.pop("PLAT_NO_UNGAPPED", None)
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    eng = Engine(0)
    t0 = time.time()
    pairs = shortcut = bad = batches = 0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    fixed = [("config2", lambda: synth.config2(10000)), ("config5", lambda: synth.config5(100, 100))]
    while time.time() - t0 < budget or fixed:
        if time.time() - t0 >= budget:
            name, mk = fixed.pop(0)
            hb = mk()
        else:
            name, hb = "stress", _adversarial_batch(seed, 200, gapped=bool(seed & 1))
            seed += 1
        r = both(eng, hb)
        diff = int((r["0"][0] != r["1"][0]).sum())
        pairs += hb.n_pairs; shortcut += r["1"][1] - r["0"][1]; bad += diff; batches += 1
        if diff:
            w = np.nonzero(r["0"][0] != r["1"][0])[0][:5]
            print("MISMATCH in", name, "seed", seed - 1, "pairs", w.tolist(), r["0"][0][w].tolist(), r["1"][0][w].tolist())
    print(json.dumps(dict(batches=batches, pairs=pairs, finished_without_dp=shortcut, differing_scores=bad, seconds=round(time.time() - t0, 1))))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
