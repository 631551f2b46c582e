"""The other BASELINE configs for bench.py: `--config 3|4|5` puts one of them on the JSON line (same contract as the
config-2 line), and the default run carries a compact summary of each under `other_configs`.

    config 3  assembly tiles (4.5 kb reference, 250 bp reads at 30x, indels): plat_assemble_batch, regions/s
    config 4  the region pipeline: reads in host memory -> VCF text, windows/s end to end
    config 5  population mode: 100 samples per window, likelihoods + genotype likelihoods + EM, GCUPS and windows/s
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


def _time_steps(torch, fn, nsteps, dist=None):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(nsteps):
        fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    return t


def _reduce(torch, dist, device, T, sums):
    el = torch.tensor([T], dtype=torch.float64, device=device)
    tot = torch.tensor(sums, dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(el.item()), [float(x) for x in tot.tolist()]


# ---- config 5 -----------------------------------------------------------------------------------------------------------------

def config5(eng, n_windows, n_ind, steps, warmup, seed=5005, dist=None):
    import torch
    from platypus_amd import synth
    hb = synth.config5(n_windows, n_ind, seed=seed)
    db = eng.upload(hb)
    st = eng.call_windows(db, want_stats=True)
    eng.em(db, 100, 0)
    eng.synchronize()
    for _ in range(warmup):
        eng.call_windows(db, want_stats=False, asynchronous=True); eng.em(db, 100, 0)
    eng.synchronize()

    def one(i):
        eng.call_windows(db, want_stats=False, asynchronous=True)    # likelihood arrays + Population.setup
        eng.em(db, 100, 0)                                            # Population.call: EM + callGenotypes
    T = _time_steps(torch, one, steps, dist)
    eng.synchronize()
    eng.profile_enable(True)
    prof = []
    for _ in range(3):
        eng.call_windows(db, want_stats=False)
        prof.append(eng.profile_last())
    eng.profile_enable(False)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); eng.em(db, 100, 0); ev1.record(); torch.cuda.synchronize()
    it = db.em_iters.cpu().numpy()
    return dict(hb=hb, st=st, T=T, steps=steps,
                kernel_ms=dict(prepare=float(np.mean([p.ms_prepare for p in prof])), seed=float(np.mean([p.ms_seed for p in prof])),
                               dp=float(np.mean([p.ms_dp for p in prof])), genotype=float(np.mean([p.ms_genotype for p in prof])),
                               em=float(ev0.elapsed_time(ev1))),
                em_iterations_mean=float(it.mean()), em_iterations_max=int(it.max()), dp_alg_bytes=int(prof[-1].dp_alg_bytes),
                dp_jobs=int(prof[-1].dp_jobs))


def line_config5(a, rank, local, world, dist):
    import torch
    from platypus_amd.engine import Engine
    eng = Engine(local)
    nwin = a.windows or 200
    r = config5(eng, nwin, 100, a.steps, a.warmup, seed=5005 + rank, dist=dist)
    st, hb = r["st"], r["hb"]
    T, (cells, run, nw) = _reduce(torch, dist, eng.device, r["T"], [st.cells_reference * a.steps, st.cells_launched * a.steps, hb.n_windows * a.steps])
    dp_ms = r["kernel_ms"]["dp"]
    ach = r["dp_alg_bytes"] / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
    return {"metric": "pair-HMM GCUPS (reference-equivalent band cells/s, read->haplotype likelihood path)", "value": cells / T / 1e9,
            "unit": "GCUPS", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * T / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE config 5: %d windows/GPU per step x 100 samples at 30x, <= 8 haplotypes; step = alignReads for "
                                   "all haplotypes and samples + genotype likelihoods [100][G] + EM / genotype calls" % nwin,
                       "windows_per_gpu": nwin, "n_ind": 100, "reads_per_step": hb.n_reads, "pairs_per_step": hb.n_pairs},
            "windows_per_sec": nw / T, "gcups_executed": run / T / 1e9, "kernel_ms": r["kernel_ms"],
            "em_iterations_mean": r["em_iterations_mean"], "em_iterations_max": r["em_iterations_max"],
            "roofline": {"bound": "hbm", "kernel": "k_dp_jobs", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": r["dp_alg_bytes"],
                         "avg_launch_ms": dp_ms}}


# ---- config 3 -----------------------------------------------------------------------------------------------------------------

def config3(eng, n_regions, steps, warmup, seed=3003, dist=None):
    import torch
    from platypus_amd import synth
    ab = synth.config3(n_regions, seed=seed)
    adb = eng.upload_assembly(ab)
    for _ in range(max(1, warmup)):
        eng.assemble_device(adb)
    torch.cuda.synchronize()
    T = _time_steps(torch, lambda i: eng.assemble_device(adb), steps, dist)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); eng.assemble_device(adb); ev1.record(); torch.cuda.synchronize()
    res = adb.results()
    # reference bytes + read bases and qualities in, ~0.1 kB of variants out per region
    alg = int(len(ab["ref_seq"]) + 2 * len(ab["read_seq"]) + 100 * n_regions)
    found = sum(len(v) for v in res)
    return dict(ab=ab, T=T, steps=steps, kernel_ms=float(ev0.elapsed_time(ev1)), alg_bytes=alg, variants=found,
                planted=sum(len(t) for t in ab["truth"]))


def line_config3(a, rank, local, world, dist):
    import torch
    from platypus_amd.engine import Engine
    eng = Engine(local)
    nreg = a.regions or 2000
    steps = min(a.steps, 50)
    r = config3(eng, nreg, steps, a.warmup, seed=3003 + rank, dist=dist)
    T, (regs,) = _reduce(torch, dist, eng.device, r["T"], [nreg * steps])
    ach = r["alg_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9
    return {"metric": "assembly tiles/s (assembleReadsAndDetectVariants, coloured de-Bruijn graph + bubble walk)", "value": regs / T,
            "unit": "regions/s", "n_gpus": world, "steps": steps, "warmup": a.warmup, "ms_per_step": 1e3 * T / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: %d tiles/GPU per step, 4.5 kb reference (1.5 kb tile +- 1.5 kb), 250 bp reads at 30x, "
                                   "1-3 indels + 0-3 SNPs per tile, k = 15, minWeight 40" % nreg, "regions_per_gpu": nreg,
                       "reads_per_step": int(r["ab"]["n_reads"])},
            "variants_found": r["variants"], "variants_planted": r["planted"],
            "roofline": {"bound": "hbm", "kernel": "k_assemble", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": r["alg_bytes"],
                         "avg_launch_ms": r["kernel_ms"]}}


def run(a, rank, local, world, dist):
    return {3: line_config3, 4: line_config4, 5: line_config5}[a.config](a, rank, local, world, dist)


# ---- config 4 -----------------------------------------------------------------------------------------------------------------

def config4(device, n_regions, region_len, workers, per_chunk, first_region=0, repeats=1, n_samples=1, pin=True):
    """The region pipeline end to end: reads of `n_regions` regions in host memory (arrays) -> VCF record text, through the native
    region loop (libplat_caller.so: host threads + every device stage batched per chunk of regions)."""
    from concurrent.futures import ThreadPoolExecutor
    from platypus_amd import fastcaller as F, synth
    from platypus_amd.options import default_options
    t0 = time.perf_counter()
    with ThreadPoolExecutor(min(16, n_regions)) as ex:
        regs = list(ex.map(lambda i: synth.config4_region_arrays(first_region + i, region_len=region_len, n_samples=n_samples), range(n_regions)))
    rr = [F.region_from_arrays(r, pin=pin) for r in regs]
    t_synth = time.perf_counter() - t0
    names = ["S%d" % (i + 1) for i in range(n_samples)]
    nc = F.NativeCaller(device, workers, per_chunk)
    nc.call_regions(rr, names, default_options())                          # every worker's scratch buffers at full size, code paths warm
    best = None
    for _ in range(repeats):
        opts = default_options()
        t0 = time.perf_counter()
        text = nc.call_regions(rr, names, opts)
        t = time.perf_counter() - t0
        if best is None or t < best[0]:
            best = (t, text, dict(nc.stats))
    nc.close()
    t, text, st = best
    planted = sum(len(r["variants"]) for r in regs)
    return dict(T=t, text=text, stats=st, regions=n_regions, region_len=region_len, reads=int(st["n_reads"]), windows=int(st["n_windows"]),
                records=int(st["n_records"]), planted=planted, synth_s=t_synth, workers=workers, per_chunk=per_chunk)


def line_config4(a, rank, local, world, dist):
    import torch
    nreg = a.regions or 64
    workers = int(os.environ.get("PLAT_CALLER_WORKERS", "16"))
    per_chunk = int(os.environ.get("PLAT_CALLER_CHUNK", "4"))
    pin = os.environ.get("PLAT_CALLER_PINNED", "1") == "1"
    r = config4(local, nreg, 100000, workers, per_chunk, first_region=rank * nreg, repeats=max(1, min(a.steps, 3)), pin=pin)
    dev = torch.device("cuda", local)
    T, (wins, regs, recs, reads) = _reduce(torch, dist, dev, r["T"], [r["windows"], r["regions"], r["records"], r["reads"]])
    st = r["stats"]
    return {"metric": "variant windows/sec end to end (reads in host memory -> VCF record text)", "value": wins / T, "unit": "windows/s",
            "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": 1e3 * T, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE config 4: %d regions/GPU x 100 kb, 30x 150 bp reads, SNPs 1e-3 + indels 1e-4, one sample; step = "
                                   "candidates -> windows -> haplotypes -> likelihoods / EM / posteriors -> INFO / FILTER -> record text for all "
                                   "regions (native region loop, %d host threads, %d regions per chunk)" % (nreg, r["workers"], r["per_chunk"]),
                       "regions_per_gpu": nreg, "region_len": 100000, "sharding": "regions by rank, records gathered to rank 0"},
            "regions_per_sec": regs / T, "reads_per_sec": reads / T, "records": recs, "windows": wins, "planted_variants": r["planted"],
            "host_seconds_per_region": st["seconds_host"] / max(1, r["regions"]),
            "device_wait_seconds_per_region": st["seconds_device_wait"] / max(1, r["regions"]),
            "host_input_bytes_per_region": 2 * 150 * r["reads"] // max(1, r["regions"]), "input_blobs_pinned": pin,
            "stage_seconds_per_region": {k: v / max(1, r["regions"]) for k, v in st["seconds_stage"].items()},
            "python_region_loop_windows_per_sec_round1": 1100.0}


def summary(eng):
    """Compact figures of configs 3 and 5 for the default line (a few seconds each)."""
    out = {}
    r = config5(eng, 200, 100, 10, 2)
    st, hb = r["st"], r["hb"]
    out["config5_population"] = dict(windows=hb.n_windows, n_ind=hb.n_ind, reads=hb.n_reads, pairs=int(st.n_pairs),
                                     ms_per_step=1e3 * r["T"] / r["steps"], gcups=st.cells_reference * r["steps"] / r["T"] / 1e9,
                                     gcups_executed=st.cells_launched * r["steps"] / r["T"] / 1e9,
                                     windows_per_sec=hb.n_windows * r["steps"] / r["T"], kernel_ms=r["kernel_ms"],
                                     em_iterations_mean=r["em_iterations_mean"])
    r = config3(eng, 500, 5, 1)
    out["config3_assembler"] = dict(regions=500, reads=int(r["ab"]["n_reads"]), regions_per_sec=500 * r["steps"] / r["T"],
                                    kernel_ms=r["kernel_ms"], hbm_frac=r["alg_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    variants_found=r["variants"], variants_planted=r["planted"])
    r = config4(0, 64, 100000, int(os.environ.get("PLAT_CALLER_WORKERS", "16")), int(os.environ.get("PLAT_CALLER_CHUNK", "4")), repeats=3)
    st = r["stats"]
    out["config4_region_pipeline"] = dict(regions=r["regions"], region_len=r["region_len"], reads=r["reads"], windows=r["windows"], records=r["records"],
                                          planted_variants=r["planted"], seconds=r["T"], windows_per_sec=r["windows"] / r["T"],
                                          regions_per_sec=r["regions"] / r["T"], reads_per_sec=r["reads"] / r["T"],
                                          host_seconds_per_region=st["seconds_host"] / r["regions"],
                                          device_wait_seconds_per_region=st["seconds_device_wait"] / r["regions"], host_threads=r["workers"],
                                          regions_per_chunk=r["per_chunk"], what="reads in host memory (arrays) -> VCF record text, native region loop")
    return out
