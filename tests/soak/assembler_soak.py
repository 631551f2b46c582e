"""Soak of k_assemble against the oracle's sequential assembler: random regions (the generator of tests/test_gpu_assembler.py, with
its edge flavours: long reads, odd reference bytes, qualities >= 128, unrelated reads that overflow the LDS table) with fresh
seeds until the time budget is used; every region's variant tuples must be identical, in order.
(Lives under tests/ because it calls the oracle, which only tests may do.)
usage (GPU box): python tests/soak/assembler_soak.py [seconds] [first seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from platypus_amd.engine import Engine              # noqa: E402
from oracle.oracle import Oracle                    # noqa: E402
from test_gpu_assembler import synth_region         # noqa: E402


def flavour(rng, seed):
    f = seed % 6
    if f == 0:                                       # long reads, odd reference bytes
        r = synth_region(rng, int(rng.integers(900, 3000)), 2, int(rng.choice([300, 400, 520])), 15, 4)
        ref = bytearray(r["ref"])
        for _ in range(3):
            q = int(rng.integers(0, len(ref) - 10))
            ref[q:q + int(rng.integers(1, 4))] = bytes(rng.choice([ord("N"), ord("R"), ord("a"), ord("n")], 1).tolist()) * 3
        r["ref"] = bytes(ref[:len(r["ref"])])
        return r
    if f == 1:                                       # quality bytes >= 128 and 0
        r = synth_region(rng, 1500, 2, 150, 25, 3)
        qs = []
        for q in r["quals"]:
            q = bytearray(q)
            for _ in range(int(rng.integers(0, 3))):
                q[int(rng.integers(0, len(q)))] = int(rng.choice([0, 128, 200, 255]))
            qs.append(bytes(q))
        r["quals"] = qs
        return r
    if f == 2:                                       # unrelated reads: many distinct k-mers, no bubble starts
        r = synth_region(rng, 3000, 2, 150, 25, 3)
        junk = [bytes(rng.choice(list(b"ACGT"), 150).tolist()) for _ in range(int(rng.integers(50, 500)))]
        at = sorted(int(x) for x in rng.integers(0, len(r["seqs"]) + 1, len(junk)))
        for j, sq in zip(reversed(at), junk):
            r["seqs"].insert(j, sq); r["quals"].insert(j, bytes([35]) * 150)
        return r
    return synth_region(rng, int(rng.integers(600, 4500)), int(rng.choice([1, 2, 2, 4])), int(rng.choice([100, 150, 250])),
                        int(rng.choice([15, 30])), int(rng.choice([0, 3, 6])))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
    eng = Engine(0)
    orc = Oracle()
    t0 = time.time()
    regions = variants = bad = batches = 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        regs = [flavour(rng, seed + i) for i in range(8)]
        k, nc = [(15, 0), (15, 1), (21, 0), (11, 0), (15, 0), (35, 0)][seed % 6]
        seed += 8
        try:
            got = eng.assemble(regs, kmer_size=k, no_cycles=nc)
        except Exception as exc:                     # a capacity error of the device path (more bubble starts than it holds) is not a mismatch
            print("skipped batch", seed - 8, repr(exc)[:120])
            continue
        for r, g in zip(regs, got):
            exp, _ = orc.assemble(r["ref"], r["ref_start"], r["assem_start"], r["assem_end"], r["seqs"], r["quals"], k, 20, 40, nc)
            if g != exp:
                bad += 1
                print("MISMATCH seed", seed - 8, "k", k, "nc", nc, len(g), len(exp))
            variants += len(exp)
        regions += len(regs); batches += 1
    print(json.dumps(dict(tool="tests/soak/assembler_soak.py", batches=batches, regions=regions, variants=variants, differing_regions=bad,
                          seconds=round(time.time() - t0, 1))))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
