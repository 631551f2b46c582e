"""The other BASELINE configs for bench.py: `--config 3|4|5` puts one of them on the JSON line (same contract as the
config-2 line), and the default run carries a compact summary of each under `other_configs`.

    config 3  assembly tiles (4.5 kb reference, 250 bp reads at 30x, indels): plat_assemble_batch, regions/s
    config 4  the region pipeline: reads in host memory -> VCF text, windows/s end to end
    config 5  population mode: 100 samples per window, likelihoods + genotype likelihoods + EM, GCUPS and windows/s
"""
import json
import os
import sys
import resource
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


class _Solo:
    """bench.Ranks for a caller that is not one of several ranks (the default line's `other_configs`)."""
    rank, world, dev_index, dist = 0, 1, 0, None

    def barrier(self):
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def reduce(self, seconds, sums):
        return seconds, [float(x) for x in sums]

    def gather(self, payload):
        return [payload]

    def describe(self):
        return {"ranks": 1, "backend": None, "gpus_shared_between_ranks": False}


def _time_steps(torch, fn, nsteps, rk=None):
    rk = rk or _Solo()
    rk.barrier()
    t0 = time.perf_counter()
    for i in range(nsteps):
        fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    rk.barrier()
    return t


# ---- config 5 -----------------------------------------------------------------------------------------------------------------

def config5(eng, n_windows, n_ind, steps, warmup, seed=5005, rk=None, weak=False):
    import torch
    from platypus_amd import synth
    hb = synth.config5_weak_evidence(n_windows, n_ind, seed=seed + 100) if weak else synth.config5(n_windows, n_ind, seed=seed)
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
    T = _time_steps(torch, one, steps, rk)
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


def config5_end_to_end(device, n_regions=48, region_len=5000, n_samples=100, rk=None, first=0, lib=None):
    """BASELINE config 5 END TO END (round 6): regions of 100 samples at 30x each through the native region loop (plat_call_regions_stream) -- candidates of
    every sample's reads, the cohort's windows and haplotypes (HOST stage B: the device stage takes one sample), one likelihood batch of windows x 100 samples,
    EM over the cohort, posteriors, per-sample genotype calls, records with 100 sample columns.  Regions loaded on demand (the reads of 100 samples are
    45 MB per 5 kb region as packed bytes)."""
    from platypus_amd import fastcaller as F
    workers = int(os.environ.get("PLAT_CALLER_WORKERS5", "8"))
    r = config4(device, range(first, first + n_regions), region_len, workers, int(os.environ.get("PLAT_CALLER_CHUNK5", "2")), repeats=2, n_samples=n_samples,
                rk=rk, lib=lib, pin=lib is None, resident=False, region_kw=dict(flank=1000))
    st, T = r["stats"], r["T"]
    out = dict(regions=r["regions"], region_len=region_len, n_samples=n_samples, reads=r["reads"], windows=r["windows"], records=r["records"], pairs=int(st["n_pairs"]),
               timed_s=T, windows_per_sec=r["windows"] / T, regions_per_sec=r["regions"] / T, reads_per_sec=r["reads"] / T,
               host_seconds_per_region=st["seconds_host"] / r["regions"], device_wait_seconds_per_region=st["seconds_device_wait"] / r["regions"],
               host_threads=r["workers"], cpus_granted_to_this_rank=getattr(rk, "cpus", None), regions_per_chunk=r["per_chunk"], inputs="loaded on demand inside the timed region",
               stage_b={"regions_on_the_device": int(st.get("n_regions_stage_b_device", 0)), "regions_left_to_the_host": int(st.get("n_regions_stage_b_host", 0)),
                        "why": "plat_stage_b_batch takes one-sample regions: a cohort's variants / windows / haplotypes are made by host/stage_b_host.hpp"})
    out.update({k: v for k, v in config4_gcups(r.get("counted"), r["regions"], T).items() if k in ("gcups", "gcups_executed", "dp_reference", "dp_launched", "pairs")})
    return out


def line_config5(a, rk):
    from platypus_amd.engine import Engine
    rank, world = rk.rank, rk.world
    eng = Engine(rk.dev_index)
    nwin = a.windows or 200
    r = config5(eng, nwin, 100, a.steps, a.warmup, seed=5005 + rank, rk=rk)
    st, hb = r["st"], r["hb"]
    T, (cells, run, nw) = rk.reduce(r["T"], [st.cells_reference * a.steps, st.cells_launched * a.steps, hb.n_windows * a.steps])
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
            "end_to_end": None if getattr(a, "no_extras", False) else config5_end_to_end(rk.dev_index, rk=rk, first=rank * 48),
            "roofline": {"bound": "hbm", "kernel": "k_dp_jobs", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": r["dp_alg_bytes"],
                         "avg_launch_ms": dp_ms}}


# ---- config 3 -----------------------------------------------------------------------------------------------------------------

def config3(eng, n_regions, steps, warmup, seed=3003, rk=None):
    import torch
    from platypus_amd import synth
    ab = synth.config3(n_regions, seed=seed)
    adb = eng.upload_assembly(ab)
    for _ in range(max(1, warmup)):
        eng.assemble_device(adb)
    torch.cuda.synchronize()
    T = _time_steps(torch, lambda i: eng.assemble_device(adb), steps, rk)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); eng.assemble_device(adb); ev1.record(); torch.cuda.synchronize()
    res = adb.results()
    # reference bytes + read bases and qualities in, ~0.1 kB of variants out per region
    alg = int(len(ab["ref_seq"]) + 2 * len(ab["read_seq"]) + 100 * n_regions)
    found = sum(len(v) for v in res)
    return dict(ab=ab, T=T, steps=steps, kernel_ms=float(ev0.elapsed_time(ev1)), alg_bytes=alg, variants=found,
                planted=sum(len(t) for t in ab["truth"]))


CONFIG3_MODEL = dict(indel_max_len=60, indel_p=0.15, n_indel=(1, 3), n_snp=(0, 3), lowq_frac=0.05)


def config3_end_to_end(device, n_regions, rk=None, first=0, lib=None, region_kw=None):
    """BASELINE config 3 END TO END: indel-heavy regions (4.5 kb contig = 1.5 kb region +- 1.5 kb, 1-3 indels of 1..60 bases +
    0-3 SNPs, 250 bp reads at 30x, 5 % of the bases below Q20) through the native region loop with --assemble=1: assembler tiles of every
    chunk in one plat_assemble_batch, their variants merged with the BAM candidates, then the called windows through the likelihoods at
    250 bp (buf = 500), EM, posteriors, records.  Inputs resident in HBM (round 5; PLAT_CALLER_RESIDENT3=0: loaded on demand by tools/synth)."""
    from platypus_amd import fastcaller as F
    workers = int(os.environ.get("PLAT_CALLER_WORKERS", "20"))
    kw = dict(flank=1500, read_len=250, model=CONFIG3_MODEL, **(region_kw or {}))
    resident = lib is None and os.environ.get("PLAT_CALLER_RESIDENT3", "1") == "1"      # inputs resident in HBM, as on the config-4 line (0: loaded on demand, rounds 2-4)
    r = config4(device, range(first, first + n_regions), 1500, workers, int(os.environ.get("PLAT_CALLER_CHUNK3", "100" if resident else "32")), repeats=3 if resident else 1,
                region_kw=kw, rk=rk, options_kw=dict(assemble=1), lib=lib, pin=lib is None, resident=resident)
    st, T = r["stats"], r["T"]
    return dict(regions=r["regions"], tiles=int(st["n_assembly_tiles"]), assembler_variants=int(st["n_assembler_variants"]), planted_variants=r["planted"],
                windows=r["windows"], records=r["records"], reads=r["reads"], pairs=int(st["n_pairs"]), timed_s=T,
                regions_per_sec=r["regions"] / T, tiles_per_sec=st["n_assembly_tiles"] / T, windows_per_sec=r["windows"] / T,
                gcups_lower_bound=st["n_pairs"] * 16 * 250 / T / 1e9,
                gcups_note="(read, haplotype) pairs of the called windows x 16 x 250 band cells / wall time of the WHOLE pipeline: the reference "
                           "runs at least one DP for every pair it does not skip",
                host_seconds_per_region=st["seconds_host"] / r["regions"], device_wait_seconds_per_region=st["seconds_device_wait"] / r["regions"],
                assemble_seconds_per_region=st["seconds_assemble"] / r["regions"], host_threads=r["workers"], cpus_granted_to_this_rank=getattr(rk, "cpus", None), regions_per_chunk=r["per_chunk"],
                inputs="resident in HBM" if r["resident"] else "loaded on demand inside the timed region", timed_s_runs=r["T_runs"],
                call_seconds=r["T_call"], native_call_seconds=st["seconds_total"], source_wait_seconds_per_region=st["seconds_source_wait"] / r["regions"],
                load_seconds_per_region=st["seconds_load"] / r["regions"], process_cpu_seconds_per_run=r["cpu_user_s"] + r["cpu_sys_s"],
                text=F.text_bytes(r["text"]).decode("ascii"))


def line_config3(a, rk):
    from platypus_amd.engine import Engine
    rank, world = rk.rank, rk.world
    eng = Engine(rk.dev_index)
    nreg = a.regions or 2000
    steps = min(a.steps, 50)
    r = config3(eng, nreg, steps, a.warmup, seed=3003 + rank, rk=rk)
    T, (regs,) = rk.reduce(r["T"], [nreg * steps])
    ach = r["alg_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9
    traffic, traffic_source = _asm_traffic(nreg)
    return {"metric": "assembly tiles/s (assembleReadsAndDetectVariants, coloured de-Bruijn graph + bubble walk)", "value": regs / T,
            "unit": "regions/s", "n_gpus": world, "steps": steps, "warmup": a.warmup, "ms_per_step": 1e3 * T / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: %d tiles/GPU per step, 4.5 kb reference (1.5 kb tile +- 1.5 kb), 250 bp reads at 30x, "
                                   "1-3 indels + 0-3 SNPs per tile, k = 15, minWeight 40" % nreg, "regions_per_gpu": nreg,
                       "reads_per_step": int(r["ab"]["n_reads"])},
            "variants_found": r["variants"], "variants_planted": r["planted"],
            "end_to_end": None if a.no_extras else {k: v for k, v in config3_end_to_end(rk.dev_index, nreg, rk=rk, first=rank * nreg).items() if k != "text"},
            "roofline": {"bound": "hbm", "kernel": "k_assemble", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": r["alg_bytes"], "avg_launch_ms": r["kernel_ms"],
                         "note": "vector issue + LDS round trips at 4 waves per SIMD (one 1024-thread workgroup per CU: the graph takes the CU's LDS), see DESIGN.md section 4"}}


def run(a, rk):
    return {3: line_config3, 4: line_config4, 5: line_config5}[a.config](a, rk)


# ---- config 4 -----------------------------------------------------------------------------------------------------------------

def config4(device, indices, region_len, workers, per_chunk, repeats=1, n_samples=1, pin=True, lib=None, region_kw=None, rk=None, packed=True,
            loaders=None, warm_regions=None, options_kw=None, resident=False, job_regions=None, warm_rounds=2):
    """The region pipeline end to end, sustained: the regions `indices` of the job's region list are LOADED ON DEMAND by a region source
    (tools/synth: generated from seed (+) region index inside the library's loader threads into a bounded set of pinned slots -- where the
    reference's BAM loader stands) and called through the native region loop (plat_call_regions_stream: host threads + every device stage
    batched per chunk of regions) -> VCF record text; then the job's one exchange: record lines to rank 0, merged there.  Every repeat is
    timed; the MEAN is reported."""
    from platypus_amd import fastcaller as F, sharding
    from platypus_amd.options import default_options
    from tools.synth import source
    rk = rk or _Solo()
    indices = list(indices)
    if not loaders:
        try:
            from bench import effective_cpus
            granted = effective_cpus()
        except Exception:
            granted = workers
        loaders = int(os.environ.get("PLAT_CALLER_LOADERS", str(max(2, min(12, granted // 2)))))
    n_slots = per_chunk * (workers + 2) + loaders
    kw = dict(region_len=region_len, n_samples=n_samples, packed=packed, pin=pin and not resident, **(region_kw or {}))
    flank0 = int(kw.get("flank", 1000))                                       # (a synthetic region: contig r<id>, [flank, flank + region_len))
    # resident: the rank's whole share is generated ONCE, outside the timed region, and its read bytes are uploaded to HBM (one slot per
    # region, 3.7 MB each: 14 GB for a GPU's WGS share); a timed run then neither generates nor moves read bytes -- "inputs already resident
    # in HBM when the timed region starts".  Not resident: each region is generated when its turn comes (the source stands where a BAM
    # loader would) and its bytes cross the link inside the timed region.
    src = source.RegionSource(indices, len(indices) if resident else n_slots, **kw)
    if resident:
        import torch
        src.make_resident(None if lib is not None else torch.device("cuda", device), threads=max(2, loaders))
    names = ["S%d" % (i + 1) for i in range(n_samples)]
    nc = F.NativeCaller(device, workers, per_chunk, lib=lib)
    # (the timed runs are the steady state of a long job -- a rank's share of a genome is eight such lists --: ONE UNTIMED PASS over the
    # whole list comes first, as the W untimed steps of the headline do.  After 80 regions the first of three timed runs was still a fifth
    # slower than the others, after 512 a tenth: slots, pinned blocks, scratch buffers, the allocator's arenas for 95 MB of record text
    # and the cores' clocks all take their time)
    nwarm = min(len(indices), warm_regions) if warm_regions is not None else len(indices)
    counted = None
    if nwarm:                                                                 # every worker's scratch buffers at full size, code paths warm
        # The untimed pass also COUNTS: with plat_caller_count_cells on, every likelihood batch goes through the synchronous entry point,
        # whose statistics kernels count the fastAlignmentRoutine calls the reference would make and their band cells (the GCUPS
        # numerator) and the DPs the device ran, and the live kernel timers (HIP events) give k_seed / k_dp_jobs durations per batch.  The
        # timed runs below call the same regions through the asynchronous entry point, counting nothing.
        nc.count_cells(True)
        nc.call_stream(nwarm, src.load_fn, src.h, names, default_options(**(options_kw or {})), n_slots, loaders)
        nc.count_cells(False)
        counted = dict(nc.stats, regions=nwarm)
        for i in range(32):                                                   # every kernel's live timers, flat (they are summed over the ranks like the counts)
            counted["kt_ms_%d" % i] = float(counted.get("kernel_ms", [0.0] * 32)[i]); counted["kt_n_%d" % i] = int(counted.get("kernel_launches", [0] * 32)[i])
        counted["counted_reads"], counted["counted_candidate_records"], counted["counted_variants"], counted["counted_windows"] = (
            counted.get("n_reads", 0), counted.get("n_candidate_records", 0), counted.get("n_variants", 0), counted.get("n_windows", 0))
    runs, text, merged, gather, st = [], "", None, None, None
    # the line's roofline kernel is timed INSIDE the timed steps (two HIP events per launch of that one kernel, nothing else changes)
    tk_name, tk_id, tk_ms, tk_n = None, -1, 0.0, 0
    if counted and lib is None:
        knames = kernel_names()
        tk_name = choose_roofline_kernel({knames[i]: counted.get("kt_ms_%d" % i, 0.0) for i in range(32)})[0]
        if tk_name in knames:
            tk_id = knames.index(tk_name)
            nc.time_kernel(tk_id)
    planted0 = src.planted
    # the exchange: every rank's text to rank 0, merged there as a permutation of whole region blocks (their order follows from the job's
    # region list alone and is worked out here, once, before the timed region: sharding.RegionTextExchange); job_regions = the list of ALL
    # ranks (region i -> rank i % N)
    xch = None
    if job_regions is not None:
        world = getattr(rk, "world", 1)
        per_rank = [[(("r%d" % g), flank0, flank0 + region_len) for g in job_regions[r::world]] for r in range(world)]
        xch = sharding.RegionTextExchange(per_rank, dist=getattr(rk, "dist", None), device=getattr(rk, "coll_device", None), lib=lib,
                                          device_index=getattr(rk, "dev_index", 0))
    nplain = (max(1, min(int(warm_rounds), 32)) if (xch is not None or resident) else 0)     # untimed rounds of the TIMED shape behind the counting pass (a run is
                                                                              # 0.1 s: allocator arenas, clocks and the exchange's pinned block settle over the first two or three)
    # The K timed runs (= the K steps of the bench contract) are bracketed ONCE: barrier + device synchronize, K x (region calls of this rank's
    # share + the exchange + the merge on rank 0), barrier + synchronize; nothing between the runs but the exchange itself (a collective).
    # The untimed rounds before them have the same shape and a barrier each.
    T_bracket = None
    cg0 = cg1 = th0 = th1 = None
    for rep in range(repeats + nplain):
        opts = default_options(**(options_kw or {}))
        if rep <= nplain:
            rk.barrier()
            if rep == nplain:
                cg0 = cgroup_cpu_stat()
                th0 = thread_cpu_seconds()
                T_bracket = time.perf_counter()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        text = nc.call_stream(len(indices), src.load_fn, src.h, names, opts, n_slots, loaders, raw="view")    # the native block itself: no copy, no decode / encode passes
        t1 = time.perf_counter()
        if xch is not None:
            merged = xch.exchange(text, nc.region_text_lengths(len(indices)))
            nranks = xch.world
        else:
            got = rk.gather(text)                                            # every rank's record lines to rank 0 ...
            nranks = len(got) if got else None
            if got is not None:
                merged = F.merge_record_texts(got, lib=lib, raw="view")       # ... merged there by (chromosome, position), runner.py:301-352
        t2 = time.perf_counter()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        st = dict(nc.stats)
        if rep < nplain:
            continue
        if tk_id >= 0:
            tk_ms += float(st["kernel_ms"][tk_id]); tk_n += int(st["kernel_launches"][tk_id])
        runs.append((t2 - t0, t1 - t0, ru1.ru_utime - ru0.ru_utime, ru1.ru_stime - ru0.ru_stime))
        if rep == repeats + nplain - 1:
            rk.barrier()
            T_bracket = time.perf_counter() - T_bracket
            cg1 = cgroup_cpu_stat()
            th1 = thread_cpu_seconds()
            gather = dict(rk.describe(), ms=1e3 * (t2 - t1), ranks=nranks, how=("one rank, regions in order: its text is the merged text (nothing copied)" if (xch is not None and xch.world == 1 and xch.plan.identity and xch.plan.ok)
                                                                                 else "region blocks") if xch is not None else "line merge",
                          records=bytes(memoryview(merged)).count(b"\n") if merged is not None else None)
    planted = (src.planted - planted0) // max(1, repeats)
    phases = {k: v / max(1, (repeats * len(indices) + nwarm)) for k, v in src.phase_seconds.items()}
    nc.close()
    src.close()
    if xch is not None:
        xch.close()
    T = float(T_bracket) / repeats                                            # seconds per step: the bracket over the K runs / K
    return dict(T=T, T_bracket=float(T_bracket), timed_kernel=dict(kernel=tk_name, ms=tk_ms, launches=tk_n), T_call=float(np.mean([r[1] for r in runs])), T_runs=[r[0] for r in runs], cpu_user_s=float(np.mean([r[2] for r in runs])),
                cpu_sys_s=float(np.mean([r[3] for r in runs])), text=text, merged=merged, gather=gather, stats=st,
                regions=len(indices), warm_regions=nwarm, region_len=region_len, reads=int(st["n_reads"]), windows=int(st["n_windows"]), records=int(st["n_records"]),
                planted=int(planted), workers=workers, per_chunk=per_chunk, loaders=loaders, n_slots=n_slots, packed=packed, source_phases=phases,
                input_bytes=int(st["input_bytes"]), counted=counted, resident=bool(resident), warm_rounds_plain=nplain, cgroup_cpu=cgroup_cpu_delta(cg0, cg1),
                thread_cpu=thread_cpu_delta(th0, th1, repeats))


def cgroup_cpu_stat():
    """The process's cgroup-v2 CPU accounting (cpu.stat + cpu.max), or None: how much CPU TIME the box grants (quota / period) and how often the
    kernel's bandwidth controller stopped every thread of the job for the rest of a period because the quota was spent."""
    try:
        st = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        mx = open("/sys/fs/cgroup/cpu.max").read().split()
        return dict(st={k: int(v) for k, v in st.items()}, quota_us=None if mx[0] == "max" else int(mx[0]), period_us=int(mx[1]))
    except (OSError, ValueError, IndexError):
        return None


def thread_cpu_seconds():
    """CPU seconds (user + system) of every thread of this process that is alive now: {tid: (name, seconds)} from /proc/self/task."""
    out = {}
    try:
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/stat" % tid).read()
                name = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tick)
            except (OSError, ValueError, IndexError):
                pass
    except (OSError, ValueError):
        return None
    return out


def thread_cpu_delta(a, b, steps):
    """CPU seconds per step of the threads that lived through the whole timed region (the interpreter's thread, the loaders of the region
    source, the HIP / ROCr runtime's own threads -- the region loop's workers are made per call and are NOT among them: their time is
    worker_cpu_seconds_per_region), by thread name, largest first."""
    if not a or not b:
        return None
    by = {}
    for tid, (name, sec) in b.items():
        if tid in a:
            by[name] = by.get(name, 0.0) + (sec - a[tid][1]) / max(1, steps)
    return dict(sorted(((k, round(v, 4)) for k, v in by.items() if v > 0), key=lambda kv: -kv[1]))


def cgroup_cpu_delta(a, b):
    """What the timed region did to the cgroup's counters (both ends read inside the bracket's barriers): periods, periods in which the quota ran
    out (every thread of the job is then parked until the period ends), the CPU time used."""
    if not a or not b:
        return None
    d = {k: b["st"].get(k, 0) - a["st"].get(k, 0) for k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec")}
    return dict(quota_cpus=None if a["quota_us"] is None else a["quota_us"] / a["period_us"], period_ms=a["period_us"] / 1e3, periods=d["nr_periods"],
                periods_throttled=d["nr_throttled"], throttled_cpu_seconds=d["throttled_usec"] / 1e6, cpu_seconds_used=d["usage_usec"] / 1e6,
                what="cgroup v2 cpu.stat over the timed region (all processes of the box's cgroup: every rank when ranks share it)")


def kernel_source_hash():
    """sha256 over the sources libplat_mi355x.so is built from (the kernels + their build flags): what a stored profile (profiles/wgs_profile.json)
    must carry for its counters to be quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "platypus_amd", "csrc")
    for f in sorted(glob.glob(base + "/*.hip") + glob.glob(base + "/*.hpp") + [base + "/Makefile"]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def kernel_names():
    """Names of the library's kernel timers by id (plat_kernel_timer_name); the header's order when the library is not loadable (CPU tests)."""
    try:
        from platypus_amd import _lib
        lib = _lib.load()
        return [(lib.plat_kernel_timer_name(i) or b"").decode() for i in range(32)]
    except Exception:
        return ["k_candidates", "k_candidates_merge", "k_candidates_filter", "k_unpack_pieces", "k_concat_tables", "k_copy_pieces", "k_gather_reads", "k_sb_variants",
                "k_sb_windows", "k_sb_haps_rank", "k_sb_prefix", "k_sb_scan", "k_sb_haps_write", "k_sb_reads", "k_validate", "k_tile_scan", "k_prep_reads", "k_sweep",
                "k_pairs", "k_seed_slow", "k_dp_jobs", "k_finalize", "k_genotype", "k_haplotype_score", "k_em", "k_variant_posterior", "k_variant_read_stats",
                "k_variant_info", "k_genotype_call", "k_assemble", "k_read_qc", "other"]


def _wgs_profile():
    """(profiles/wgs_profile.json, whether it was collected from this build's kernel sources, this build's hash)."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "wgs_profile.json")))
    except Exception:
        prof = {}
    here = kernel_source_hash()
    return prof, bool(prof) and prof.get("kernel_source_hash") == here, here


def choose_roofline_kernel(kms):
    """The kernel the line's `roofline` names, DETERMINISTICALLY: the first kernel of the rocprofv3 --kernel-trace --stats ranking of this very command
    (profiles/wgs_profile.json: collected with the bench's 24 workers, so it ranks the kernels as they run in the timed region) when that file was made
    from this build's kernel sources; else the kernel with the largest summed live time of the counting pass (every launch bracketed by HIP events, one
    chunk on the chip at a time).  Returns (name, how it was chosen, the counting pass's order)."""
    order = sorted([k for k in kms if kms[k] > 0 and k != "other"], key=lambda k: -kms[k])
    if not order:
        return None, None, order
    prof, same, here = _wgs_profile()
    if same and prof.get("ranking"):
        for k in prof["ranking"]:
            if k in kms and kms[k] > 0:
                how = ("first kernel of the rocprofv3 --kernel-trace --stats ranking of this command (profiles/wgs_profile.json <- %s, same kernel sources, hash %s); "
                       "the counting pass's live timers (every kernel by itself) rank %s first" % (prof.get("ranking_source", "?"), here, order[0]))
                return k, how, order
    return order[0], "largest summed live launch time in the counting pass (HIP events around every launch, one chunk at a time); profiles/wgs_profile.json %s" % (
        "is absent" if not prof else "was collected from other kernel sources"), order


def config4_gcups(counted, regions, T, totals=None, timed_kernel=None):
    """Both halves of BASELINE.json's metric on the WGS workload: the band cells of the DPs the reference would run for the called windows
    and the greedy rounds (counted by the device's statistics kernels during the untimed pass over the same regions) / the timed wall
    time; plus the roofline entries of the loop's kernels from the live timers of that pass.

    The kernel the line's `roofline` names is chosen DETERMINISTICALLY: every kernel of a chunk is timed live in the counting pass (a pair of
    HIP events around each launch, plat_kernel_times, one chunk on the chip at a time); the kernel with the largest SUMMED time is the
    dominant one -- unless profiles/wgs_profile.json (rocprofv3 --kernel-trace --stats of this command, same kernel sources by hash) ranks
    another kernel first, in which case rocprof's ranking wins and the line says so.  HBM traffic (PMC) is quoted from that file only when
    its source hash is this build's; otherwise `traffic` is null and `traffic_source` says why."""
    if not counted or not counted.get("n_dp_reference"):
        return {}
    totals = totals or {}
    scale = regions / max(1, counted["regions"])                             # (the untimed pass normally covers the whole list: 1.0)
    out = {"gcups": counted["cells_reference"] * scale / T / 1e9, "gcups_executed": counted["cells_launched"] * scale / T / 1e9,
           "dp_reference": int(counted["n_dp_reference"] * scale), "dp_launched": int(counted["n_dp_launched"] * scale),
           "pairs": int(counted["n_pairs"] * scale),
           "cells_counted_on": "the untimed pass over %d of the %d regions (plat_caller_count_cells: synchronous likelihood batches + statistics kernels)" % (counted["regions"], regions)}
    nb = max(1, counted["n_align_batches"])
    ndp = counted["n_dp_launched"]
    out["dp_per_launch"] = ndp / nb
    names = kernel_names()
    kms = {names[i]: float(counted.get("kt_ms_%d" % i, 0.0)) for i in range(32)}
    kln = {names[i]: int(counted.get("kt_n_%d" % i, 0)) for i in range(32)}
    if not any(kms.values()):
        return out
    # ALGORITHMIC bytes per launch of each kernel (what the step has to move, once), from the counted quantities of the same pass
    hapb, readb, nreads, npairs = counted["align_hap_bytes"], counted["align_read_bytes"], counted["align_reads"], counted["n_pairs"]
    nh = max(1.0, hapb / 650.0)
    rec = 16 + 24 * 19 + 32
    table_bases = float(counted.get("candidates_bytes", 0))                  # bases of every read of the chunk tables, once
    nreads_tab, ncand, nvar, nwin = float(totals.get("reads", 0)), float(totals.get("candidate_records", 0)), float(totals.get("variants", 0)), float(totals.get("windows", 0))
    alg = {
        "k_dp_jobs": (counted["align_dp_bytes"], "4 x read length + 34 bytes per DP launched (SURVEY 8(d)); VALU-issue bound"),
        "k_sweep": (2 * hapb + rec * nh, "haplotype bytes in, gap-open bytes out, one record per haplotype"),
        "k_pairs": (rec * nh + readb / 4 + 16 * nreads + 32 * npairs + 8 * max(npairs - ndp, 0) + 4 * ndp, "haplotype records + read planes in, pair records / likelihoods out"),
        "k_prep_reads": (2 * readb + readb / 4 + 16 * nreads, "window reads' bases and qualities in, bit planes + descriptors out"),
        "k_candidates": (table_bases + 28 * nreads_tab, "every base of the chunk's reads once -- as 2-bit codes (a quarter of a byte per base) when the scan runs on codes "
                                                        "(round 6: bytes and qualities are read only where codes differ), as one byte otherwise -- + pos / flags / offsets / CIGAR "
                                                        "per read; a lane walks one read"),
        "k_unpack_pieces": (float(counted.get("unpack_bytes", 0)), "one packed byte in, a base and a quality out (+ the base's 2-bit code: a quarter of a byte)"),
        "k_candidates_merge": (20 * ncand + 32 * nvar, "the scan's records in (5 words each), distinct candidates out"),
        "k_candidates_filter": (32 * ncand / 4 + 32 * nvar, "the merge table's occupied slots in, supported candidates out"),
        "k_gather_reads": (2 * 2 * readb + 2 * 17 * nreads, "the called windows' reads: bases + qualities in and out, per-read fields in and out"),
        "k_sb_variants": (32 * nvar * 2 + 36 * nvar + 40 * nwin, "merged candidates in (8 words), variant columns (9) + window table out; latency bound: dependent LDS passes of one workgroup per region"),
        "k_sb_haps_rank": (2 * hapb / 2, "the window's reference bytes in per combination ranked"),
        "k_sb_haps_write": (2 * hapb, "reference bytes in, haplotype bytes out"),
        "k_seed_slow": (64.0 * 256, "a few hundred undecided pairs per launch: latency bound"),
        "k_genotype": (8 * npairs + 8 * 3 * nwin, "per-read log-likelihoods in, genotype likelihoods out"),
        "k_em": (8 * 3 * nwin * 2, "genotype likelihoods in, frequencies / calls out"),
        "k_variant_read_stats": (2 * readb / 2, "the called variants' window reads once"),
    }
    top, chosen_by, order = choose_roofline_kernel(kms)
    prof, same, here = _wgs_profile()
    traffic_source = ("profiles/wgs_profile.json <- %s; collected %s at commit %s (tools/profile_round.sh): same kernel sources as this build (hash %s)" % (
        prof.get("source", "rocprofv3 --pmc passes"), (prof.get("measured") or {}).get("date"), (prof.get("measured") or {}).get("commit"), here)) if same else (
        "null: profiles/wgs_profile.json %s -- counters need rocprofv3 around the process and are never collected by this run" % (
            "is absent" if not prof else "was collected from other kernel sources (hash %s, this build %s)" % (prof.get("kernel_source_hash"), here)))
    if top != order[0]:
        order.remove(top); order.insert(0, top)

    def entry(k):
        n = max(1, kln[k])
        ms = kms[k] / n
        a, note = alg.get(k, (None, None))
        d = {"bound": "hbm", "kernel": k, "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None, "avg_launch_ms": ms,
             "launches": kln[k], "total_ms": kms[k], "share_of_kernel_time": kms[k] / max(1e-12, sum(kms.values()))}
        if a is not None:
            per = float(a) / n
            d.update(achieved=per / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, algorithmic_bytes_per_launch=int(per), note=note)
            d["frac"] = d["achieved"] / HBM_PEAK_GBPS
        pk = (prof.get("kernels") or {}).get(k) if same else None
        if pk and pk.get("hbm_bytes_per_launch") is not None:
            d["traffic"] = int(pk["hbm_bytes_per_launch"])
        return d
    cands = [entry(k) for k in order]
    # the roofline kernel INSIDE the timed region: its launches of the K timed steps bracketed by HIP events on their streams (plat_kernel_timer_only), next to
    # whatever shares the chip with them there -- `achieved` and `frac` are computed from THAT average; the counting pass's (the kernel by itself) stays beside it
    if timed_kernel and timed_kernel.get("kernel") == cands[0]["kernel"] and timed_kernel.get("launches", 0) > 0:
        r0 = cands[0]
        alone = r0["avg_launch_ms"]
        ms = timed_kernel["ms"] / timed_kernel["launches"]
        r0.update(avg_launch_ms=ms, launches=int(timed_kernel["launches"]), avg_launch_ms_by_itself=alone,
                  timed_in="the K timed steps (HIP events around this kernel's launches on the workers' streams, %d launches); avg_launch_ms_by_itself = the counting pass" % timed_kernel["launches"])
        if r0.get("algorithmic_bytes_per_launch") is not None and ms > 0:
            r0["achieved_by_itself"], r0["frac_by_itself"] = r0["achieved"], r0["frac"]
            r0["achieved"] = r0["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e9
            r0["frac"] = r0["achieved"] / HBM_PEAK_GBPS
    cands[0]["traffic_source"] = traffic_source
    cands[0]["kernel_chosen_by"] = chosen_by
    out["roofline"] = cands[0]
    if len(cands) > 1:
        out["roofline_other"] = cands[1]
    if len(cands) > 2:
        out["roofline_more"] = cands[2:8]
    out["kernel_time_ranking"] = [{"kernel": k, "total_ms": round(kms[k], 3), "launches": kln[k], "avg_us": round(1e3 * kms[k] / max(1, kln[k]), 2)} for k in order]
    out["kernel_ms_per_chunk_sum"] = sum(kms.values()) / nb
    return out


def line_config4(a, rk, lib=None, region_len=100000, region_kw=None, resident=None):
    """ONE region list for the whole job, region i -> rank i % N (runner.py:473-474); every rank streams its share through the region loop,
    the record lines travel to rank 0 (sizes all-gather + point to point) and are merged by (chrom, pos) (runner.py:301-352).  The timed
    region covers the calls, the gather and the merge (and, when the inputs are not resident, loading = generating the regions).

    Which scaling the N = 1, 2, 4, 8 lines form is said on the line, never implied:
      default     STRONG (round 6): the SAME list for every N -- the 31 000 regions of the synthetic 30x genome (SURVEY 8(d) cfg 4), region i -> rank
                  i % N; one GPU takes the whole genome (112 GB of packed reads resident in its 288 GB).  Efficiency: T(1) / (N x T(N)) = value(N) / (N x value(1)).
      --weak      WEAK: the list grows with the job, 3 875 regions per GPU (= what a GPU of an 8-GPU job gets).  windows/s(N) / (N x windows/s(1)).
      --regions R R regions for the whole job whatever N (a strong line too, and labelled so)."""
    from platypus_amd import fastcaller as F, sharding
    rank, world = rk.rank, rk.world
    strong = not bool(getattr(a, "weak", False)) or bool(a.regions)          # round 6: the job is the whole synthetic genome unless --weak asks for a share per GPU
    per_gpu = int(os.environ.get("PLAT_BENCH_WGS_REGIONS_PER_GPU", "3875"))   # (tests shrink the share; 3 875 = 31 000 / 8)
    total = (a.regions or (8 * per_gpu if strong else per_gpu * world))
    mine = sharding.regions_for_rank(total, rank, world)
    cpus = getattr(rk, "cpus", 16)                                           # what the box grants this rank (cgroup quota / share of the node)
    # worker and loader threads share the rank's CPUs (workers sleep while the device works on their chunk): 10 + 8 with six regions per
    # chunk on the 16 CPUs one GPU box grants (tools/run_c4_sweep4.sh, run_c4_sweep5.sh: the box's own run-to-run spread, +-8 %, is as large
    # as the differences between 10-12 workers, 6-10 loaders and 4-8 regions per chunk; 16 + 8 and 4 per chunk measured 5 % below)
    if resident is None:
        resident = os.environ.get("PLAT_CALLER_RESIDENT", "1") == "1"        # 0: regions generated and uploaded inside the timed region (rounds 2-3)
    # (resident: the loaders only hand out stored structs, so the workers get the CPUs -- 14 workers x 8 regions per chunk measured best on the
    #  16 CPUs of a one-GPU box: 1.21 M windows/s against 1.05 M with 10 x 6; 2 CPUs: 0.26 M, 4 CPUs: 0.43 M -- host stages 0.6 ms per region)
    # (round 5, resident: one worker per CPU -- 16 x 64 measured 2.40 M windows/s against 2.15 M with 14 workers; their host stages are
    #  0.23 ms per region now and a worker waiting for the device sleeps)
    # (later in round 5: plat_stream_sync naps between polls of its event instead of the runtime's blocking wait, which burned 0.085 ms of CPU
    #  per region while "blocked" -- a waiting worker now costs nothing, so there are three workers for two CPUs: 24 x 64 measured
    #  3.4-3.5 M windows/s against 3.2-3.3 M with 16; GPU_MAX_HW_QUEUES 4 (the default), 8, 16: the same, 24: 40 % slower; 2 CPUs: 3 workers 0.71 M)
    workers = int(os.environ.get("PLAT_CALLER_WORKERS", str(max(2, min(24, cpus * 3 // 2 if resident else cpus * 5 // 8)))))
    os.environ.setdefault("PLAT_CALLER_LOADERS", str(max(2, min(12, cpus // 2))))
    # (round 5: stage B on the device -- the host no longer pays per region for a chunk's size, and the kernels of a chunk are latency bound:
    #  64 regions per chunk measured 2.4 M windows/s against 1.25 M with 8, and a k_dp_jobs launch then holds ~36 k DPs)
    # (round 6, the whole genome on one GPU: 128 regions per chunk -- the latency-bound kernels of a chunk cost the same for 64 and 128 regions, 22.7 -> 16.5 us
    #  of kernel time per region -- measured 5.1 M windows/s against 4.97 M with 64 and 4.8 M with 256; a short list keeps 64: more chunks than workers)
    #  A SHORT list (a rank of an 8-GPU job: 3 875 regions) wants four chunks per worker more than it wants wide launches -- its 0.08 s are mostly the
    #  pipeline filling and draining: 24 / 32 / 40 / 48 / 64 / 96 / 128 per chunk measured 4.35 / 4.76 / 4.32-4.73 / 4.30 / 4.10 / 3.69 / 4.08 M windows/s (+-8 % run to run))
    auto_chunk = max(32, min(128, (len(mine) // max(1, 4 * workers)) // 8 * 8))
    per_chunk = int(os.environ.get("PLAT_CALLER_CHUNK", str(auto_chunk) if resident else "16"))
    pin = os.environ.get("PLAT_CALLER_PINNED", "1") == "1" and lib is None
    packed = os.environ.get("PLAT_CALLER_PACKED", "1") == "1"
    repeats = max(1, min(int(a.steps or 10), 200))                           # K steps = K runs over the rank's share, timed in one bracket (a run is ~0.1 s)
    r = config4(rk.dev_index, mine, region_len, workers, per_chunk, repeats=repeats, pin=pin, lib=lib, region_kw=region_kw, rk=rk, packed=packed,
                resident=resident, job_regions=None if os.environ.get("PLAT_BENCH_LINE_MERGE") == "1" else list(range(total)),
                warm_rounds=getattr(a, "warmup", 2) or 2)
    cnt = r.get("counted") or {}
    ckeys = ("cells_reference", "cells_launched", "n_dp_reference", "n_dp_launched", "n_pairs", "regions", "n_align_batches", "align_hap_bytes", "align_read_bytes",
             "align_reads", "align_dp_bytes", "seconds_kernel_seed", "seconds_kernel_dp", "seconds_kernel_sweep", "seconds_kernel_pairs",
             "seconds_kernel_unpack", "seconds_kernel_candidates", "unpack_bytes", "candidates_bytes", "n_unpack_launches", "n_candidates_launches",
             "counted_reads", "counted_candidate_records", "counted_variants", "counted_windows") + tuple("kt_ms_%d" % i for i in range(32)) + tuple("kt_n_%d" % i for i in range(32))
    tkr = r.get("timed_kernel") or {}
    T, red = rk.reduce(r["T"], [r["windows"], r["regions"], r["records"], r["reads"], r["T_call"], r["input_bytes"], float(tkr.get("ms", 0.0)), float(tkr.get("launches", 0))] +
                       [float(cnt.get(k, 0)) for k in ckeys])
    wins, regs, recs, reads, tcall, inb, tk_ms, tk_n = red[:8]
    counted_all = dict(zip(ckeys, red[8:]))                                   # summed over the ranks
    st = r["stats"]
    line = {"metric": "variant windows/sec end to end (reads in host memory -> VCF record text)", "value": wins / T, "unit": "windows/s",
            "n_gpus": world, "steps": repeats, "warmup": int(getattr(a, "warmup", 2) or 2), "untimed_passes": 1 + r.get("warm_rounds_plain", 0), "ms_per_step": 1e3 * T, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE config 4: %d regions x %d bp for the whole job, 30x 150 bp reads, SNPs 1e-3 + indels 1e-4, one sample, %s; step = "
                                   "candidates -> windows -> haplotypes -> likelihoods / EM / posteriors -> INFO / FILTER -> record text for all "
                                   "regions (native region loop, %d host threads, %d regions per chunk), then the gather of the record lines to "
                                   "rank 0 and their merge; %d timed step(s) in one bracket" % (
                                       total, region_len,
                                       "every region's reads generated from seed (+) index BEFORE the timed region and resident in HBM (plat_read_table.dev_seq; the "
                                       "per-read arrays in host memory): the timed region moves no read bytes" if r["resident"] else
                                       "each region generated from seed (+) index when its turn comes (region source in %d loader threads, %d pinned slots)" % (r["loaders"], r["n_slots"]),
                                       r["workers"], r["per_chunk"], repeats),
                       "inputs": "resident in HBM" if r["resident"] else "generated and uploaded inside the timed region",
                       "regions": total, "region_len": region_len, "sharding": "region i -> rank i % N, records gathered to rank 0 and merged by (chrom, pos)",
                       "read_encoding": "packed: one byte per base (2-bit base | quality << 2)" if r["packed"] else "ASCII bases + quality bytes"},
            "regions_per_sec": regs / T, "reads_per_sec": reads / T, "records": recs, "windows": wins, "regions": regs, "timed_s": T,
            "timed_s_runs": r["T_runs"], "process_cpu_seconds_per_run": {"user": r["cpu_user_s"], "sys": r["cpu_sys_s"],
                                                                           "cpus_busy": (r["cpu_user_s"] + r["cpu_sys_s"]) / r["T"]},
            "planted_variants": r["planted"], "seconds_calls_mean_over_ranks": tcall / world,
            "host_seconds_per_region": st["seconds_host"] / max(1, r["regions"]),
            "worker_cpu_seconds_per_region": st.get("seconds_worker_cpu", 0.0) / max(1, r["regions"]),
            "device_wait_seconds_per_region": st["seconds_device_wait"] / max(1, r["regions"]),
            "source_seconds_per_region": st["seconds_load"] / max(1, r["regions"]), "source_phase_seconds_per_region": r["source_phases"],
            "worker_seconds_waiting_for_the_source_per_region": st["seconds_source_wait"] / max(1, r["regions"]),
            "host_input_bytes_per_region": inb / max(1.0, regs), "h2d_gbytes_per_sec": inb / T / 1e9, "input_blobs_pinned": pin, "cpus_granted_to_this_rank": cpus, "cgroup_cpu": r.get("cgroup_cpu"), "cpu_seconds_per_step_of_the_threads_that_outlive_a_call": r.get("thread_cpu"),
            "stage_seconds_per_region": {k: v / max(1, r["regions"]) for k, v in st["seconds_stage"].items()},
            "record_gather": r["gather"], "untimed_warm_regions_per_rank": r["warm_regions"], "python_region_loop_windows_per_sec_round1": 1100.0,
            # what an efficiency figure over the N = 1, 2, 4, 8 lines has to be computed from (the driver computes it, not this line)
            "scaling_efficiency_basis": {"scaling": "strong" if strong else "weak", "timed_s": T, "regions": int(regs), "windows": int(wins), "ranks": world,
                                         "regions_per_rank": int(regs) // max(1, world), "cpus_per_rank": cpus, "host_threads_per_rank": r["workers"],
                                         "formula": "T(1) / (N x T(N)) over lines with equal `regions`" if strong else
                                                    "windows_per_sec(N) / (N x windows_per_sec(1)): `regions` grows with N (3 875 per GPU)",
                                         "inside_timed_region": "region calls of every rank + gather of the record text to rank 0 + (chrom, pos) merge"}}
    line["inputs"] = "resident in HBM" if r["resident"] else "generated and uploaded inside the timed region"
    line["stage_b"] = {"regions_on_the_device": int(st.get("n_regions_stage_b_device", 0)), "regions_left_to_the_host": int(st.get("n_regions_stage_b_host", 0)),
                       "windows_left_to_the_host": int(st.get("n_windows_stage_b_host", 0)), "of_this_ranks_regions": r["regions"],
                       "regions_with_dictionaries_replayed_on_the_device": int(st.get("n_regions_dict_replay_device", 0))}
    line.update(config4_gcups(counted_all, regs, T, totals=dict(reads=counted_all.get("counted_reads", 0), candidate_records=counted_all.get("counted_candidate_records", 0),
                                                                variants=counted_all.get("counted_variants", 0), windows=counted_all.get("counted_windows", 0)),
                              timed_kernel=dict(kernel=tkr.get("kernel"), ms=tk_ms, launches=int(tk_n))))
    # The HEADLINE shape (round 6): BASELINE.json's metric is "pair-HMM GCUPS + variant windows/sec (synth 30x WGS)" -- this workload.  `value` is the
    # first half (reference-equivalent GCUPS of the whole job), the second half and everything an efficiency figure needs sit INSIDE
    # `config`, which the driver keeps whole.
    wps = line["value"]
    line["windows_per_sec"] = wps
    line["config"].update({"windows_per_sec": wps, "gcups": line.get("gcups"), "gcups_executed": line.get("gcups_executed"), "cpus_per_rank": cpus,
                           "host_threads_per_rank": r["workers"], "regions_per_chunk": r["per_chunk"], "regions_per_rank": int(regs) // max(1, world),
                           "windows": int(wins), "records": int(recs), "scaling": "strong" if strong else "weak", "timed_region_ms": 1e3 * r["T_bracket"],
                           "step": "one pass of the native region loop over this rank's regions + the exchange of the record text to rank 0 + the merge",
                           "gcups_is": "REFERENCE-EQUIVALENT (SURVEY 8(d)): band cells of the fastAlignmentRoutine calls the reference would make for the called windows / wall time; "
                                       "gcups_executed counts only the DPs the device ran"})
    cg = r.get("cgroup_cpu")
    if cg:                                                                    # what bounds the job on this box: the cgroup's CPU-time quota (rank 0's view of its cgroup)
        line["config"].update({"cpu_quota_cpus": cg["quota_cpus"], "cpus_busy": round((r["cpu_user_s"] + r["cpu_sys_s"]) / r["T"], 2),
                               "cpu_quota_periods_throttled": "%d of %d" % (cg["periods_throttled"], cg["periods"])})
    if line.get("gcups") is not None:
        line["metric"] = "pair-HMM GCUPS + variant windows/sec (synthetic 30x WGS): value = reference-equivalent GCUPS of the whole job, config.windows_per_sec = variant windows/sec end to end"
        line["value"], line["unit"] = line["gcups"], "GCUPS"
        line["value_is"] = line["config"]["gcups_is"]
    if rank == 0:
        if lib is None and world == 1 and not getattr(a, "no_cpu_baseline", False):      # (rank 0 at N = 1 only: the bench contract)
            line["cpu_baseline"] = config4_cpu_baseline()
        line["merged_text"] = bytes(memoryview(r["merged"])).decode("ascii")                               # (popped by bench.py before printing; the tests read it)
    return line


def config4_cpu_baseline(seconds=10.0):
    """The reference's own kernel (unmodified align.c, oracle/_ref, traceback on) on one host core over the DPs of a bounded sample of
    config-4 windows: GCUPS, the half of the metric a CPU figure exists for (the reference's region loop itself cannot run here)."""
    try:
        from bench import cpu_baseline
        from platypus_amd import synth
        hb = synth.config2(150, seed=4004)                    # the same read length, depth and window shape as the windows config 4 calls
        d = cpu_baseline(hb, seconds)
        d["sample"] = "150 windows of config-2 shape (150 bp reads, 30x): " + str(d.get("sample", ""))
        return d
    except Exception as exc:                                  # pragma: no cover
        return {"error": repr(exc)[:200]}


def _asm_traffic(nreg):
    """(bytes per launch, where from) of k_assemble out of profiles/dp_traffic.json -- counters need rocprofv3 around the process, so the
    figure of the last profiled run is quoted (scaled to this launch's regions), never measured here."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "dp_traffic.json")))["k_assemble"]
        if d.get("kernel_source_hash") != kernel_source_hash():
            return None, "null: profiles/dp_traffic.json's k_assemble counters were collected from other kernel sources (hash %s, this build %s)" % (d.get("kernel_source_hash"), kernel_source_hash())
        per = float(d["hbm_bytes_per_launch"]) / float(d["regions_per_launch"])
        m = d.get("measured", {})
        return int(per * nreg), "not measured in this run: profiles/dp_traffic.json <- profiles/r%02d_pmc_assemble.txt (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, %d tiles per launch); collected %s at commit %s" % (
            int(m.get("round", 0)), int(d["regions_per_launch"]), m.get("date"), m.get("commit"))
    except Exception:
        return None, None


def _roof(kernel, alg_bytes, ms, note=None):
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    d = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": ms}
    if note:
        d["note"] = note
    return d


def summary(eng):
    """Configs 3, 4 and 5 at their SURVEY 8(d) sizes for the default line, each with the roofline entry of its dominant kernel."""
    out = {}
    # config 5: 2 000 windows x 100 samples, as 10 distinct batches of 200 windows (one batch per step)
    r = config5(eng, 200, 100, 10, 2)
    st, hb = r["st"], r["hb"]
    out["config5_population"] = dict(windows_per_step=hb.n_windows, steps=r["steps"], n_ind=hb.n_ind, reads=hb.n_reads, pairs=int(st.n_pairs),
                                     ms_per_step=1e3 * r["T"] / r["steps"], gcups=st.cells_reference * r["steps"] / r["T"] / 1e9,
                                     gcups_executed=st.cells_launched * r["steps"] / r["T"] / 1e9,
                                     windows_per_sec=hb.n_windows * r["steps"] / r["T"], kernel_ms=r["kernel_ms"],
                                     em_iterations_mean=r["em_iterations_mean"],
                                     roofline=_roof("k_dp_jobs", r["dp_alg_bytes"], r["kernel_ms"]["dp"], "VALU-issue bound, see DESIGN.md"))
    try:
        out["config5_population"]["end_to_end"] = config5_end_to_end(0)
    except Exception as exc:                                  # pragma: no cover
        out["config5_population"]["end_to_end"] = {"error": repr(exc)[:200]}
    # the same geometry with weak evidence (1x, mostly low-quality bases): the EM iterates
    r = config5(eng, 200, 100, 10, 2, weak=True)
    st, hb = r["st"], r["hb"]
    G = np.diff(hb.win_hap_begin).astype(np.int64)
    em_bytes = int(8 * hb.n_ind * (G * (G + 1) // 2).sum() * 2 + 8 * G.sum() + 4 * hb.n_ind * hb.n_windows)   # likelihoods in, EM likelihoods + frequencies + calls out
    out["config5_weak_evidence"] = dict(windows_per_step=hb.n_windows, n_ind=hb.n_ind, reads=hb.n_reads, pairs=int(st.n_pairs),
                                        ms_per_step=1e3 * r["T"] / r["steps"], windows_per_sec=hb.n_windows * r["steps"] / r["T"],
                                        kernel_ms=r["kernel_ms"], em_iterations_mean=r["em_iterations_mean"], em_iterations_max=r["em_iterations_max"],
                                        what="100 samples at 1x, four fifths of the bases below Q20: the EM of Population.call runs for tens of iterations",
                                        roofline=_roof("k_em_wide", em_bytes, r["kernel_ms"]["em"], "latency bound: a chain of dependent fp64 sums per iteration"))
    nt = 2000
    r = config3(eng, nt, 5, 1)
    e2e = {k: v for k, v in config3_end_to_end(0, nt).items() if k != "text"}
    out["config3_assembler"] = dict(regions=nt, reads=int(r["ab"]["n_reads"]), regions_per_sec=nt * r["steps"] / r["T"],
                                    kernel_ms=r["kernel_ms"], variants_found=r["variants"], variants_planted=r["planted"],
                                    roofline=_roof("k_assemble", r["alg_bytes"], r["kernel_ms"], "vector issue + LDS round trips at 4 waves per SIMD, see DESIGN.md"),
                                    end_to_end=e2e)
    nreg = int(os.environ.get("PLAT_BENCH_CONFIG4_REGIONS", "3875"))           # one GPU's share of the 31 000 regions (SURVEY 8(d) cfg 4)
    try:
        from bench import effective_cpus
        cpus = effective_cpus()
    except Exception:
        cpus = 16
    r = config4(0, range(nreg), 100000, int(os.environ.get("PLAT_CALLER_WORKERS", str(max(2, min(16, cpus * 5 // 8))))),
                int(os.environ.get("PLAT_CALLER_CHUNK_STREAMED", "16")), repeats=3)     # the mean of three runs over the whole share
    st = r["stats"]
    out["config4_region_pipeline"] = dict(regions=r["regions"], untimed_warm_regions=r["warm_regions"], region_len=r["region_len"], reads=r["reads"], windows=r["windows"], records=r["records"],
                                          planted_variants=r["planted"], timed_s=r["T"], timed_s_runs=r["T_runs"], windows_per_sec=r["windows"] / r["T"],
                                          regions_per_sec=r["regions"] / r["T"], reads_per_sec=r["reads"] / r["T"],
                                          host_seconds_per_region=st["seconds_host"] / r["regions"],
                                          device_wait_seconds_per_region=st["seconds_device_wait"] / r["regions"],
                                          source_seconds_per_region=st["seconds_load"] / r["regions"],
                                          worker_seconds_waiting_for_the_source_per_region=st["seconds_source_wait"] / r["regions"],
                                          host_input_bytes_per_region=r["input_bytes"] / r["regions"], h2d_gbytes_per_sec=r["input_bytes"] / r["T"] / 1e9,
                                          host_threads=r["workers"], cpus_granted_to_this_rank=cpus, loader_threads=r["loaders"], regions_per_chunk=r["per_chunk"],
                                          read_encoding="packed (1 B/base)" if r["packed"] else "ascii (2 B/base)",
                                          **config4_gcups(r.get("counted"), r["regions"], r["T"]),
                                          what="regions generated on demand into pinned slots (region source) -> native region loop -> VCF record text; "
                                               "timed_s = the MEAN of three runs over the whole share (timed_s_runs), no best-of")
    return out
