#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md: BASELINE configs 3 (assembler) and 5 (population mode), plus the oracle's
single-core time on a sample for scale.  Not the headline bench (that is bench.py / config 2)."""
import json
import sys
import time

import numpy as np

_os = __import__("os")
sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "tests"))
from platypus_amd import synth
from platypus_amd.engine import Engine
from tests.test_gpu_assembler import synth_region


def main():
    import torch
    eng = Engine(0)
    out = {}
    # ---- config 3: 4.5 kb regions, 250 bp reads at 30x, indels + SNPs
    rng = np.random.default_rng(3003)
    nreg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    regions = [synth_region(rng, 4500, 2, 250, 30, 4) for _ in range(nreg)]
    eng.assemble(regions[:8])
    t0 = time.perf_counter(); res = eng.assemble(regions); t = time.perf_counter() - t0
    from oracle.oracle import Oracle
    o = Oracle()
    t0 = time.perf_counter()
    for r in regions[:16]:
        o.assemble(r["ref"], r["ref_start"], r["assem_start"], r["assem_end"], r["seqs"], r["quals"])
    tc = (time.perf_counter() - t0) / 16
    out["config3_assembler"] = dict(regions=nreg, reads_per_region=len(regions[0]["seqs"]), gpu_regions_per_sec_incl_h2d=nreg / t,
                                    oracle_1core_regions_per_sec=1 / tc, variants=sum(len(v) for v in res))
    # ---- config 5: population mode
    hb = synth.config5(200, 100)
    db = eng.upload(hb)
    st = eng.call_windows(db); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.call_windows(db, want_stats=False)
    eng.synchronize(); t = (time.perf_counter() - t0) / 5
    out["config5_population"] = dict(windows=hb.n_windows, n_ind=hb.n_ind, reads=hb.n_reads, pairs=int(st.n_pairs),
                                     ms_per_step=1e3 * t, gcups=st.cells_reference / t / 1e9, windows_per_sec=hb.n_windows / t)
    # EM + genotype calls for the same windows
    eng.em(db, 100, 0); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.em(db, 100, 0)
    eng.synchronize(); t = (time.perf_counter() - t0) / 5
    it = db.em_iters.cpu().numpy()
    out["config5_population"].update(em_ms=1e3 * t, em_iterations_mean=float(it.mean()), em_iterations_max=int(it.max()))
    # ---- candidate generation from CIGARs: 400 regions x 400 reads x 150 bp
    rng = np.random.default_rng(99)
    Bs = np.frombuffer(b"ACGT", dtype=np.uint8)
    regs = []
    for g in range(400):
        ref = Bs[rng.integers(0, 4, 6000)].tobytes()
        reads = []
        for r in range(400):
            p0 = int(rng.integers(2000, 3800))
            seq = bytearray(ref[p0:p0 + 70] + ref[p0 + 73:p0 + 153])           # a 3-bp deletion after 70 matched bases
            for k in rng.integers(12, 138, 2):
                seq[int(k)] = Bs[int(rng.integers(0, 4))]
            reads.append(dict(seq=bytes(seq), qual=bytes([35] * 150), pos=p0, flag=3, cigar=[(0, 70), (2, 3), (0, 80)]))
        regs.append(dict(ref=ref, ref_seq_start=0, contig_len=6000, reads=reads))
    eng.candidates(regs[:4])
    t0 = time.perf_counter()
    res = eng.candidates(regs)
    t = time.perf_counter() - t0
    out["candidates"] = dict(regions=len(regs), reads=160000, records=sum(len(x) for x in res), reads_per_sec_incl_host_packing=160000 / t)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
