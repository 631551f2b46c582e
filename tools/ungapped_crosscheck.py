"""Cross-check of k_seed's ungapped-alignment shortcut against the DP it replaces: the same batches aligned with the shortcut and
with PLAT_NO_UNGAPPED=1 (every pair through the DP) must give identical scores.  Stress batches (tests/test_gpu_parity.py::
_adversarial_batch: repeats, cheap gaps, mismatches at the read ends, quality minima down to 1) with fresh seeds until the time
budget is used, then BASELINE config 2 and a config-5 sample.   usage: python tools/ungapped_crosscheck.py [seconds] [first seed] [--bigq]

--bigq: every stress batch is drawn in the WRAP regime of the reference's int16 adds (250 bp reads of Q 60..93, quality sums up to
23 000, gap-open penalties down to 1; align.c:81,94-100,520).  The ungapped proof stands aside for such reads (its cost model is exact
arithmetic); the run compares three modes pair by pair: the product (guard on), every pair through the DP, and -- measurement only,
PLAT_UNGAPPED_BIGQ=1 -- the proof let loose on them, which says whether the guard is needed at all."""
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


def both(eng, hb, bigq=False):
    out = {}
    for mode in ("0", "1") + (("loose",) if bigq else ()):
        os.environ["PLAT_NO_UNGAPPED"] = "1" if mode == "1" else "0"
        if mode == "loose":
            os.environ["PLAT_UNGAPPED_BIGQ"] = "1"
        db = eng.upload(hb)
        st = eng.align(db, want_stats=True)
        eng.synchronize()
        out[mode] = (db.score.cpu().numpy()[:hb.n_pairs].copy(), int(st.n_dp_launched))
        os.environ.pop("PLAT_UNGAPPED_BIGQ", None)
    os.environ.pop("PLAT_NO_UNGAPPED", None)
    return out


def main():
    bigq = "--bigq" in sys.argv
    argv = [x for x in sys.argv if x != "--bigq"]
    budget = float(argv[1]) if len(argv) > 1 else 120.0
    eng = Engine(0)
    t0 = time.time()
    pairs = shortcut = bad = batches = 0
    big_pairs = loose_bad = loose_shortcut = 0
    seed = int(argv[2]) if len(argv) > 2 else 1000
    fixed = [("config2", lambda: synth.config2(10000)), ("config5", lambda: synth.config5(100, 100))]
    while time.time() - t0 < budget or fixed:
        if time.time() - t0 >= budget:
            name, mk = fixed.pop(0)
            hb = mk()
        else:
            name, hb = "stress", _adversarial_batch(seed, 200, gapped=bool(seed & 1), bigq=bigq)
            seed += 1
        r = both(eng, hb, bigq and name == "stress")
        if "loose" in r:
            qs = np.add.reduceat(hb.read_qual.astype(np.int64), hb.read_off[:-1])
            nbig = 0                                        # pairs whose read is in the wrap regime
            for w in range(hb.n_windows):
                nbig += int((qs[hb.win_read_begin[w]:hb.win_read_begin[w + 1]] > 15000).sum()) * int(hb.win_hap_begin[w + 1] - hb.win_hap_begin[w])
            big_pairs += nbig
            loose_bad += int((r["loose"][0] != r["1"][0]).sum())
            loose_shortcut += r["1"][1] - r["loose"][1]
        diff = int((r["0"][0] != r["1"][0]).sum())
        pairs += hb.n_pairs; shortcut += r["1"][1] - r["0"][1]; bad += diff; batches += 1
        if diff:
            w = np.nonzero(r["0"][0] != r["1"][0])[0][:5]
            print("MISMATCH in", name, "seed", seed - 1, "pairs", w.tolist(), r["0"][0][w].tolist(), r["1"][0][w].tolist())
    out = dict(batches=batches, pairs=pairs, finished_without_dp=shortcut, differing_scores=bad, seconds=round(time.time() - t0, 1))
    if bigq:
        out.update(wrap_regime=True, pairs_with_quality_sum_above_15000=big_pairs,
                   proof_let_loose=dict(finished_without_dp=loose_shortcut, differing_scores=loose_bad,
                                        what="PLAT_UNGAPPED_BIGQ=1: the ungapped proof NOT standing aside in the wrap regime (measurement only)"))
    print(json.dumps(out))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
