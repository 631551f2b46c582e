#!/usr/bin/env python3
"""bench.py -- headline benchmark: BASELINE.json's metric on its own workload, the synthetic 30x WGS (config 4).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Default line (round 6): BASELINE config 4.  One "step" = one pass of the hot path over this rank's share of the genome's regions
(the 31 000 regions x 100 kb of the synthetic genome, region i -> rank i % N: ONE GPU takes the whole genome; `--weak`: 3 875 regions per GPU), reads already resident in HBM: candidates ->
variants -> windows -> haplotypes (device) -> pair-HMM likelihoods / genotype likelihoods / EM / posteriors -> VCF record text through the
native region loop, then the job's ONE exchange (record text to rank 0, RCCL under "nccl") and the merge -- all inside the timed region.
`value` = pair-HMM GCUPS, reference-equivalent (band cells of the fastAlignmentRoutine calls the reference would make for
the called windows / wall time); `config.windows_per_sec` = the metric's second half; `roofline` = the loop's dominant kernel by summed
live launch time (deterministic: tools/bench_other.config4_gcups); `cpu_baseline` = the unmodified reference align.c on the host's cores.
Region i -> rank i % N, no data-path collective (runner.py:470-500).

At N = 1 the line also carries `config2` (the batched pair-HMM of BASELINE config 2 on resident batches: reference-equivalent and
executed GCUPS, every reference DP executed, the hard workload, k_dp_jobs' roofline) and `other_configs` (3, 4 streamed, 5).
`--config 2|3|4|5` puts that config alone on the line (same contract); `--config 4` is the default line without the sub-blocks.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(hb, seconds=10.0):
    """The reference's own kernel (unmodified align.c in oracle/_ref, traceback on = production mode) timed
    on ONE host core over the DP instances of a bounded sample of the same workload, then on every core the
    process may run on; plus the oracle port of the whole per-window path as a secondary figure."""
    from oracle.oracle import LIBREF, Oracle, HERE as ORC_DIR
    o = Oracle()
    nwin = min(hb.n_windows, 150)
    rows = []
    for w in range(nwin):
        haps = hb.window_haps(w)
        rd = hb.window_reads(w)
        ws, we, fl = int(hb.win_start[w]), int(hb.win_end[w]), int(hb.win_flank[w])
        gos = [o.gap_open(h) for h in haps]
        for h, g in zip(haps, gos):
            for r in range(len(rd["seq"])):
                L = len(rd["seq"][r])
                ov = min(we, int(rd["end"][r])) - max(ws, int(rd["pos"][r]))
                if rd["kind"][r] != 2 and ((rd["flags"][r] & 512) or ov < 7):
                    continue
                idx = min(int(rd["pos"][r]) - (ws - fl), len(h) - L - 15)      # calign.pyx:252
                st = max(0, idx - 8)
                rows.append((h[st:st + L + 15], rd["seq"][r], rd["qual"][r], g[st:st + L + 15]))
    n = len(rows)
    lmax = max(len(r[1]) for r in rows)
    H = np.full((n, lmax + 15), ord("A"), dtype=np.uint8); R = np.full((n, lmax), ord("A"), dtype=np.uint8)
    Q = np.zeros((n, lmax), dtype=np.uint8); G = np.ones((n, lmax + 15), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for j, (h, r, q, g) in enumerate(rows):
        L = len(r); lens[j] = L
        H[j, :L + 15] = np.frombuffer(h, dtype=np.uint8); R[j, :L] = np.frombuffer(r, dtype=np.uint8)
        Q[j, :L] = np.frombuffer(q, dtype=np.uint8); G[j, :L + 15] = np.frombuffer(g, dtype=np.uint8)
    out = {"cores": 1, "cpu": cpu_model(),
           "sample": "DP instances (one per aligned read x haplotype pair, at the read's mapping offset) of the "
                     "first %d windows of the workload = %d DPs, repeated to ~%.0f s" % (nwin, n, seconds)}
    lib = C.CDLL(os.path.join(ORC_DIR, "libcpubench.so"))
    lib.cpu_time_reference_dp.restype = C.c_double
    lib.cpu_time_reference_dp.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    if os.path.exists(LIBREF):
        cs, cells = C.c_longlong(0), C.c_longlong(0)

        def run(tb, reps):
            return lib.cpu_time_reference_dp(LIBREF.encode(), n, lmax, H.ctypes.data, R.ctypes.data, Q.ctypes.data,
                                             G.ctypes.data, lens.ctypes.data, tb, reps, C.byref(cs), C.byref(cells))
        t1 = run(1, 1)
        reps = max(1, int(seconds / max(t1, 1e-6)))
        t = run(1, reps)
        out.update(kind="reference", value=cells.value / t / 1e9, unit="GCUPS",
                   what="unmodified src/c/align.c fastAlignmentRoutine, traceback on (production mode), gcc -O2, 1 thread")
        t0 = run(0, max(1, reps // 4))
        out["score_only_gcups"] = cells.value / t0 / 1e9
        # every host core at once ("one worker per core over the same batches"): one thread per CPU this process
        # may run on, each running the same rows through the reference kernel (ctypes releases the GIL; align.c keeps no
        # global state)
        import threading
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        if ncores > 1:
            per = max(1, reps // 5)

            def at(nthreads):
                def worker(k):
                    c1, c2 = C.c_longlong(0), C.c_longlong(0)
                    lib.cpu_time_reference_dp(LIBREF.encode(), n, lmax, H.ctypes.data, R.ctypes.data, Q.ctypes.data,
                                              G.ctypes.data, lens.ctypes.data, 1, per, C.byref(c1), C.byref(c2))
                tw = time.perf_counter()
                th = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                return nthreads * per * float((16 * lens.astype(np.int64)).sum()) / (time.perf_counter() - tw) / 1e9
            # one thread per CPU of the affinity mask is the whole host; a container may be granted fewer CPUs than it shows
            # (then fewer threads do better), so a few counts are tried and all of them reported
            tried = {k: at(k) for k in sorted({min(32, ncores), min(64, ncores), min(128, ncores), ncores})}
            best = max(tried, key=tried.get)
            out["all_cores"] = {"cores": best, "value": tried[best], "unit": "GCUPS", "cpus_in_affinity_mask": ncores,
                                "by_threads": {str(k): v for k, v in tried.items()},
                                "what": "the same kernel, one thread per core, traceback on: best of the thread counts tried "
                                        "(the box shows %d CPUs, the affinity mask holds %d)" % (os.cpu_count() or 0, ncores)}
    else:
        # no prebuilt reference .so on this box: time the oracle port's DP instead
        t0 = time.perf_counter()
        o.dp_batch(H, R, Q, G, lens)
        t = time.perf_counter() - t0
        out.update(kind="port", value=float((16 * lens.astype(np.int64)).sum()) / t / 1e9, unit="GCUPS",
                   what="oracle/plat_oracle.c scalar restatement (no SIMD), 1 thread")
    # secondary: whole per-window path (hash + vote + DP with traceback + log-likelihood) with the oracle port
    t0 = time.perf_counter(); ndp = 0
    for w in range(min(nwin, 40)):
        ndp += o.align_window(hb.window_haps(w), int(hb.win_start[w]), int(hb.win_end[w]), int(hb.win_flank[w]),
                              hb.window_reads(w))[2]
    tp = time.perf_counter() - t0
    out["port_whole_path_windows_per_sec"] = min(nwin, 40) / tp
    return out


class Env:
    """Temporarily set environment switches the library reads per call (PLAT_NO_UNGAPPED, PLAT_NO_EXACT, PLAT_NO_NLOW)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_launch_command(n, argv, port=None):
    """The command `bench.py --gpus N` re-executes itself under when it was started WITHOUT a launcher: one rank per GPU of this node,
    rendezvous on 127.0.0.1 (the reference's fan-out being replaced: runner.py:470-500, one PlatypusSingleProcess per share of the
    region list)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def effective_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (a container often shows every CPU of
    the host and is granted a fraction: the round-3 GPU box shows 256 and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def numa_cpus_of_device(torch, index):
    """CPUs of the NUMA node the GPU hangs off (sysfs), or None when the box does not say."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return cpus
    except Exception:
        return None


def pin_rank_cpus(torch, rank, world, dev_index):
    """Give every rank of the node its own share of the host CPUs, on the GPU's NUMA node when sysfs names it (the ranks' worker
    threads and pinned buffers then stay local); contiguous shares of the affinity mask otherwise.  Returns the number of CPUs."""
    if not hasattr(os, "sched_getaffinity"):
        return os.cpu_count() or 1
    allowed = sorted(os.sched_getaffinity(0))
    if world <= 1 or len(allowed) < 2 * world:
        return len(allowed)
    mine = None
    local = numa_cpus_of_device(torch, dev_index) if torch.cuda.is_available() else None
    if local:
        loc = [c for c in allowed if c in local]
        # the ranks whose GPUs share this node split it between them: assume the usual layout (consecutive GPUs on one node)
        per_node = max(1, world // max(1, len({tuple(sorted(numa_cpus_of_device(torch, d) or [])) for d in range(torch.cuda.device_count())})))
        k = rank % per_node
        share = len(loc) // per_node
        if share >= 2:
            mine = loc[k * share:(k + 1) * share]
    if not mine:
        share = len(allowed) // world
        mine = allowed[rank * share:(rank + 1) * share]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return len(allowed)
    return len(mine)


class Ranks:
    """This process as one rank of the job: RANK / LOCAL_RANK / WORLD_SIZE from the launcher, one GPU per rank, a process group over
    RCCL (backend "nccl") -- or gloo when PLAT_DIST_BACKEND says so or the ranks have to share GPUs (a one-GPU box running the
    two-rank path: RCCL wants one device per rank)."""

    def __init__(self, want_gpus, need_gpu=True):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        ndev = torch.cuda.device_count() if (need_gpu or torch.cuda.is_available()) else 0
        self.shared = ndev > 0 and ndev < min(self.world, int(os.environ.get("LOCAL_WORLD_SIZE", self.world)))
        self.dev_index = self.local % ndev if ndev else 0
        self.device = torch.device("cuda", self.dev_index) if ndev else torch.device("cpu")
        self.backend = None
        self.cpus = effective_cpus()
        if "RANK" in os.environ:                # under a launcher (also with one rank): a process group
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.backend = os.environ.get("PLAT_DIST_BACKEND") or ("nccl" if ndev and not self.shared else "gloo")
            if ndev:
                torch.cuda.set_device(self.dev_index)
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            self.cpus = max(1, min(pin_rank_cpus(torch, self.rank, self.world, self.dev_index), effective_cpus() // max(1, self.world)))
        # tensors of the collectives live where the backend wants them
        self.coll_device = self.device if self.backend in (None, "nccl") else torch.device("cpu")

    def barrier(self):
        if self.device.type == "cuda":
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def reduce(self, seconds, sums):
        """(max over ranks of `seconds`, sums over ranks of `sums`)."""
        torch = self.torch
        el = torch.tensor([seconds], dtype=torch.float64, device=self.coll_device)
        tot = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=self.coll_device)
        if self.dist is not None:
            self.dist.all_reduce(el, op=self.dist.ReduceOp.MAX)
            self.dist.all_reduce(tot, op=self.dist.ReduceOp.SUM)
        return float(el.item()), [float(x) for x in tot.tolist()]

    def gather(self, payload):
        from platypus_amd import sharding
        return sharding.gather_records(payload, self.dist, device=self.coll_device if self.coll_device.type == "cuda" else None)

    def describe(self):
        return {"ranks": self.world, "backend": self.backend, "gpus_shared_between_ranks": bool(self.shared), "cpus_per_rank": self.cpus}

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def selftest_ranks(a):
    """`--selftest-ranks` (CPU suite): the launch path alone -- N ranks, process group, the record gather and merge, one JSON line
    from rank 0 -- with no device work, so that `bench.py --gpus N` can be exercised where there is no GPU."""
    from platypus_amd import sharding
    rk = Ranks(a.gpus, need_gpu=False)
    mine = sharding.regions_for_rank(23, rk.rank, rk.world)
    recs = [("r%d" % g, 100 + g, "r%d\t%d\t.\tA\tC" % (g, 101 + g)) for g in mine]
    t0 = time.perf_counter()
    rk.barrier()
    got = rk.gather(sharding.encode_records(recs))
    T, (n,) = rk.reduce(time.perf_counter() - t0, [len(recs)])
    if rk.rank == 0:
        merged = sharding.merge_record_streams([sharding.decode_records(x) for x in got])
        print(json.dumps({"metric": "selftest", "n_gpus": rk.world, "record_gather": dict(rk.describe(), records=len(merged)),
                          "records_sum": n, "in_order": merged == ["r%d\t%d\t.\tA\tC" % (g, 101 + g) for g in range(23)]}))
    rk.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 10 passes over the WGS share; --config 2: 400 steps)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=None, choices=(2, 3, 4, 5), help="one BASELINE config alone on the line (default: config 4 = the metric's workload, with the config-2 and other-config sub-blocks at N = 1)")
    ap.add_argument("--windows", type=int, default=None, help="windows per GPU per batch (config 2: 10000; config 5: 200)")
    ap.add_argument("--regions", type=int, default=None, help="config 3: assembly tiles per GPU per step (2000); config 4: regions of the WHOLE job "
                                                              "(default 3875 per GPU), region i -> rank i %% N")
    ap.add_argument("--batches", type=int, default=8, help="distinct resident batches the steps walk through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (all-DP, hard workload, other configs)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches in flight per GPU: one plat_ctx + HIP stream each; step i runs on stream i %% S "
                         "(1 = strictly one batch at a time)")
    ap.add_argument("--sync-entry", action="store_true",
                    help="time plat_align_window_batch (two internal read-backs) instead of plat_align_window_batch_async")
    ap.add_argument("--strong", action="store_true", help="(the default since round 6) ONE region list for every N: the 31000 regions of the whole synthetic genome")
    ap.add_argument("--weak", action="store_true", help="config 4: 3875 regions per GPU (the share of a GPU of an 8-GPU job) instead of the whole genome for every N: a weak-scaling line")
    ap.add_argument("--no-other-configs", action="store_true", help="default line: leave out other_configs (configs 3, 4 streamed, 5)")
    ap.add_argument("--no-wgs", action="store_true", help=argparse.SUPPRESS)     # (round 5: the config-2 line carried a wgs block; accepted, ignored)
    ap.add_argument("--min-seconds", type=float, default=0.25, help="a step is made of as many passes (one batch each) as it takes for the K timed steps to last this long")
    ap.add_argument("--selftest-ranks", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # started without a launcher: become N ranks, one per GPU (the driver's N>1 form goes through torch.distributed.run itself)
        import subprocess
        sys.exit(subprocess.call(rank_launch_command(a.gpus, sys.argv[1:])))
    if a.selftest_ranks:
        return selftest_ranks(a)

    import torch
    rk = Ranks(a.gpus)
    if a.config in (3, 5):
        from tools import bench_other
        a.steps = a.steps or 20
        line = bench_other.run(a, rk)
    elif a.config == 2:
        a.steps = a.steps or 400
        line = line_config2(a, rk)
    else:
        line = line_wgs(a, rk)
    if rk.rank == 0 and line is not None:
        line.pop("merged_text", None)
        print(json.dumps(line))
    rk.close()


def line_wgs(a, rk):
    """The default line: BASELINE config 4 on this job's GPUs (tools/bench_other.line_config4: value / config / roofline / cpu_baseline are the WGS
    job's), + at N = 1 the config-2 sub-block and the other configs."""
    from types import SimpleNamespace
    from tools import bench_other
    a.steps = a.steps or 10
    sub = a.config is None and rk.world == 1 and not a.no_extras
    line = bench_other.line_config4(a, rk, resident=True)
    if rk.rank != 0:
        if sub:
            pass
        return None
    if sub:
        try:
            a2 = SimpleNamespace(steps=20, warmup=3, windows=None, batches=a.batches, streams=a.streams, sync_entry=a.sync_entry, min_seconds=a.min_seconds,
                                 no_extras=False, no_other_configs=True, no_cpu_baseline=True)
            c2 = line_config2(a2, rk)
            keep = ("metric", "value", "value_is", "unit", "steps", "ms_per_step", "config", "windows_per_sec", "gcups_executed", "dp_reference_per_step", "dp_launched_per_step",
                    "kernel_ms", "dp_kernel_gcups", "roofline", "roofline_other", "roofline_third", "algorithmic_frac_8d", "step_hbm_frac", "gcups_all_dp", "all_dp",
                    "exact_match_shortcut_only", "hard_workload")
            line["config2"] = {k: c2[k] for k in keep if k in c2}
        except Exception as exc:                # pragma: no cover
            line["config2"] = {"error": repr(exc)[:300]}
        try:
            if a.no_other_configs:
                raise RuntimeError("left out (--no-other-configs)")
            from platypus_amd.engine import Engine
            line["other_configs"] = bench_other.summary(Engine(rk.dev_index))
            c4 = line["other_configs"].get("config4_region_pipeline") or {}
            if "windows_per_sec" in c4:         # the same job with every region generated and uploaded INSIDE the timed region (rounds 2-3's shape)
                line["config"]["windows_per_sec_streamed_inputs"] = c4["windows_per_sec"]
        except Exception as exc:                # pragma: no cover
            line["other_configs"] = {"error": repr(exc)[:300]}
    return line


def line_config2(a, rk):
    """BASELINE config 2 alone: the batched pair-HMM (alignReads for all haplotypes + genotype likelihoods) over resident batches of 10 000 windows."""
    import torch
    rank, local, world, dist = rk.rank, rk.dev_index, rk.world, rk.dist
    from platypus_amd import synth
    from platypus_amd.engine import Engine
    from concurrent.futures import ThreadPoolExecutor

    # B distinct batches stay resident in HBM; consecutive steps go to different plat_ctx / HIP streams so that the
    # latency-bound stages of one batch (prepare, seeding, genotype) overlap the VALU-bound DP of another -- what a caller
    # streaming regions through the library does.  Every step is one full pass over one batch of `--windows` windows.
    B = max(1, a.batches)
    S = max(1, min(a.streams, B))
    nwin_batch = a.windows or 10000
    engs = [Engine(local) for _ in range(S)]
    with ThreadPoolExecutor(min(B, 8)) as ex:   # (numpy releases the GIL in the generator's big array operations)
        hbs = list(ex.map(lambda j: synth.config2(nwin_batch, seed=2002 + rank + 1000 * j), range(B)))
    streams = [torch.cuda.Stream(device=engs[0].device) for _ in range(S)]
    dbs = [engs[0].upload(h) for h in hbs]      # inputs resident in HBM before the timed region
    eng, hb, db = engs[0], hbs[0], dbs[0]
    torch.cuda.synchronize()

    barrier = rk.barrier

    def step(i, dbl=dbs, **kw):
        k = i % len(dbl)
        j = k % S                               # a batch always goes to the same stream: its output buffers are never written by two at once
        with torch.cuda.stream(streams[j]):
            return engs[j].call_windows(dbl[k], **kw)

    def sync_all():
        for j in range(S):
            with torch.cuda.stream(streams[j]):
                engs[j].synchronize()           # also raises any error an asynchronous step recorded on the device
        torch.cuda.synchronize()

    def timed(nsteps, dbl=dbs, passes=1):
        barrier()
        t0 = time.perf_counter()
        for i in range(nsteps * passes):
            step(i, dbl, want_stats=False, asynchronous=not a.sync_entry)
        sync_all()
        t1 = time.perf_counter()
        barrier()
        return t1 - t0

    sts = [eng.call_windows(d, want_stats=True) for d in dbs]      # per-batch statistics (and scratch buffers at full size)
    for i in range(max(a.warmup, 1) * S):
        step(i, want_stats=False, asynchronous=not a.sync_entry)
    sync_all()
    # A step = P consecutive passes, one resident batch each, P chosen so that the K timed steps last at least --min-seconds (a driver run
    # with --steps 20 would otherwise time 9 ms): the figure does not hang on twenty launches.  P comes from an untimed probe, the same on
    # every rank (max over ranks).
    probe = timed(2 * S) / (2 * S)
    P = max(1, int(np.ceil(a.min_seconds / max(a.steps * probe, 1e-9)))) if a.min_seconds > 0 else 1
    P = int(rk.reduce(float(P), [0.0])[0])
    wall = timed(a.steps, passes=P)
    NP = a.steps * P                            # passes inside the timed region

    def over_steps(f):                          # sum of a per-batch statistic over the passes of the K timed steps
        return float(sum(f(sts[i % B], hbs[i % B]) for i in range(NP)))
    T, (cells_ref, cells_run, nwin, ndp_ref, ndp_run) = rk.reduce(wall, [
        over_steps(lambda q, h: q.cells_reference), over_steps(lambda q, h: q.cells_launched), over_steps(lambda q, h: h.n_windows),
        over_steps(lambda q, h: q.n_dp_reference), over_steps(lambda q, h: q.n_dp_launched)])

    # live per-kernel timing of the dominant kernel (HIP events on the launch stream), untimed extra steps
    eng.profile_enable(True)
    dp_ms, seed_ms, seedk_ms, fin_ms, gen_ms, prep_ms, sweep_ms, pairs_ms = [], [], [], [], [], [], [], []
    prof = None
    for _ in range(5):
        # (the entry point the timed passes use: the asynchronous one leaves no record of the pairs k_pairs finishes, the synchronous one
        #  writes 32 bytes per pair -- its k_pairs is 13 % longer, and rocprof of the timed command would not agree with it)
        eng.call_windows(db, want_stats=False, asynchronous=not a.sync_entry)
        prof = eng.profile_last()
        dp_ms.append(prof.ms_dp); seed_ms.append(prof.ms_seed); fin_ms.append(prof.ms_finalize)
        gen_ms.append(prof.ms_genotype); prep_ms.append(prof.ms_prepare); seedk_ms.append(prof.ms_seed_kernel)
        sweep_ms.append(prof.ms_sweep); pairs_ms.append(prof.ms_pairs)
    eng.profile_enable(False)

    # the job's one real exchange -- per-rank result records gathered to rank 0 and merged by (chrom, pos)
    # (runner.py:301-352).  It happens once at the end of a job, so it is exercised here OUTSIDE the timed region (a
    # failure in it is reported, it never hides the measured line).
    gather = None
    try:
        from platypus_amd import sharding
        eng.synchronize()
        nrec = min(hb.n_windows, 256)
        recs = sharding.format_window_records(hb, db.logl.cpu().numpy(), windows=range(nrec), chrom=str(rank + 1))
        recs.sort(key=lambda r: (sharding.chrom_key(r[0]), r[1]))
        tg = time.perf_counter()
        got = rk.gather(sharding.encode_records(recs))
        if rank == 0:
            merged = sharding.merge_record_streams([sharding.decode_records(x) for x in got])
            gather = dict(rk.describe(), records=len(merged), ranks=len(got), ms=1e3 * (time.perf_counter() - tg))
    except Exception as exc:                    # pragma: no cover
        gather = {"error": repr(exc)[:200]}

    if rank == 0:
        ms_step = 1e3 * T / a.steps
        ms_pass = 1e3 * T / NP
        dp_avg = float(np.mean(dp_ms))
        # roofline entries of the two kernels that share the top of the profile: k_dp_jobs (the recurrence) and k_seed
        # (candidate diagonals + the ungapped-alignment proof).  `roofline` is the one with the longer launch.
        pm = {}
        tf = os.path.join(ROOT, "profiles", "dp_traffic.json")      # per-launch PMC figures from rocprofv3 --pmc (see profiles/README.md)
        if os.path.exists(tf):
            try:
                pm = json.load(open(tf))
            except Exception:
                pm = {}

        # the PMC figures are NOT collected by this run (counters need rocprofv3 around the process): they are the last committed pass -- and they are
        # quoted only when that pass was made from THESE kernel sources (hash of csrc/*.hip, *.hpp, Makefile); otherwise traffic is null and says why
        meas = pm.get("measured") or {}
        from tools import bench_other as _bo
        here = _bo.kernel_source_hash()
        if pm and pm.get("kernel_source_hash") != here:
            traffic_source = "null: profiles/dp_traffic.json was collected from other kernel sources (hash %s, this build %s) -- counters are never collected by this run" % (pm.get("kernel_source_hash"), here)
            pm = {}
        else:
            traffic_source = None if not pm else "not measured in this run: profiles/dp_traffic.json <- %s; collected %s at commit %s (tools/profile_round.sh): same kernel sources as this build (hash %s)" % (
                pm.get("source", "rocprofv3 --pmc passes"), meas.get("date", "in round %s" % pm.get("round", "?")), meas.get("commit", "?"), here)

        def entry(kernel, alg_bytes, ms, counters, note):
            ach = alg_bytes / (ms * 1e-3) / 1e9
            sec = None
            if counters.get("valu_insts_per_launch") and counters.get("busy_cycles_per_launch"):
                # what actually bounds these kernels (SURVEY 8(d) "secondary"): wave-instructions issued per SIMD-cycle.
                # A half-rate packed op (v_pk_*, v_alignbit, v_perm: 85 % of the DP's mix) takes ~4.2 cycles.
                cpi = counters["busy_cycles_per_launch"] * 1024.0 / counters["valu_insts_per_launch"]
                sec = {"bound": "valu-issue", "insts_per_launch": counters["valu_insts_per_launch"],
                       "busy_cycles_per_launch": counters["busy_cycles_per_launch"], "simds": 1024,
                       "cycles_per_inst_per_simd": cpi, "frac_of_half_rate_issue_peak": 4.2 / cpi}
            return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBPS, "traffic": counters.get("hbm_bytes_per_launch"), "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": ms, "note": note, "secondary": sec}
        seedk_avg = float(np.mean(seedk_ms))
        # k_seed per launch: haplotype bytes in (1 B/base) + gap-open bytes out (1 B/base; 4-byte DP words until round 3), read bit planes in (2 bits/base)
        # + ReadInfo (16 B/read), one PairRec + one Job out per (haplotype, read) pair (32 B); since round 2 the kernel also
        # finishes the pairs that need no DP: their log-likelihood out (8 B), and one dense-list entry (4 B) per pair that does
        dp_per_step = ndp_run / NP / max(world, 1)
        seed_alg = 2 * int(hb.hap_off[-1]) + int(hb.read_off[-1]) // 4 + 16 * hb.n_reads + 32 * hb.n_pairs \
            + int(8 * max(hb.n_pairs - dp_per_step, 0) + 4 * dp_per_step)
        # (the asynchronous entry keeps no count of the DPs it launched -- plat_profile.dp_alg_bytes is 0 there --: the bytes come from the synchronous
        #  statistics pass over the same batch: cells_launched = 16 x length summed over the launched DPs, so cells / 4 + 34 per DP = 4 x length + 34)
        dp_jobs0 = int(sts[0].n_dp_launched)
        dp_alg_bytes = int(sts[0].cells_launched) // 4 + 34 * dp_jobs0
        r_dp = entry("k_dp_jobs", dp_alg_bytes, dp_avg, pm,
                     "recurrence is VALU-issue bound (packed int16), not HBM bound: see DESIGN.md")
        sweep_avg, pairs_avg = float(np.mean(sweep_ms)), float(np.mean(pairs_ms))
        if sweep_avg > 0 and pairs_avg > 0:
            # round 4: the seeding stage is two kernels.  k_sweep per launch: haplotype bytes in, gap-open bytes out (1 B/base each) + one record
            # out per haplotype (flags, three bit planes, per-chunk minima); k_pairs: those records in, read bit planes in (2 bits/base), ReadInfo
            # (16 B/read), PairRec + Job (32 B/pair), 8 B of log-likelihood per pair it finishes / 4 B per pair it queues
            maxhap = int(np.max(np.diff(hb.hap_off)))
            rec = 16 + 24 * (((maxhap + 63) >> 6) + 8) + (((((maxhap + 63) >> 6) + 8) + 15) & ~15)
            sweep_alg = 2 * int(hb.hap_off[-1]) + rec * hb.n_haps
            pairs_alg = rec * hb.n_haps + int(hb.read_off[-1]) // 4 + 16 * hb.n_reads + 32 * hb.n_pairs + int(8 * max(hb.n_pairs - dp_per_step, 0) + 4 * dp_per_step)
            note = "bit-parallel proofs on bit planes / LDS multiplicity maps: bound by vector issue (64-bit shifts), not by HBM; see DESIGN.md section 4"
            cands = [r_dp, entry("k_pairs", pairs_alg, pairs_avg, pm.get("k_pairs", {}), note), entry("k_sweep", sweep_alg, sweep_avg, pm.get("k_sweep", {}), note)]
        else:
            cands = [r_dp, entry("k_seed", seed_alg, seedk_avg, pm.get("k_seed", {}),
                                 "LDS k-mer maps + bit-parallel proofs: bound by vector issue (secondary), not by HBM; see DESIGN.md section 4")]
        # the sub-block's `roofline` is ALWAYS k_dp_jobs -- the dominant kernel of config 2 by every rocprofv3 --stats summary kept under profiles/ --, never
        # the winner of a 1 % HIP-event tie with k_pairs; the seeding kernels follow by launch time
        rest = sorted(cands[1:], key=lambda r: -r["avg_launch_ms"])
        roof, roof_other = cands[0], rest[0]
        roof_more = rest[1:]
        line = {
            "metric": "pair-HMM GCUPS (reference-equivalent band cells/s, read->haplotype likelihood path)",
            "value": cells_ref / T / 1e9,
            "value_is": "REFERENCE-EQUIVALENT (SURVEY 8(d)): cells of the DPs the reference would run / wall time; see gcups_executed and gcups_all_dp",
            "unit": "GCUPS",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: %d windows/GPU per pass, 150 bp reads, 30x, <=8 haplotypes/window, SNP-only; "
                                   "pass = alignReads for all haplotypes + genotype likelihoods of one resident batch; step = %d pass(es) over "
                                   "consecutive batches (so that the %d timed steps last >= %.2f s)" % (nwin_batch, P, a.steps, a.min_seconds),
                       "windows_per_gpu": nwin_batch, "passes_per_step": P, "ms_per_pass": ms_pass, "read_len": 150, "depth": 30,
                       "sharding": "windows by rank, no collective",
                       "entry": "plat_align_window_batch" if a.sync_entry else "plat_align_window_batch_async",
                       "batches_in_flight": S, "distinct_resident_batches": B, "timed_region_ms": 1e3 * T},
            "windows_per_sec": nwin / T,
            "windows_per_sec_is": "config 2's windows (likelihoods + genotype likelihoods only); the end-to-end figure is the WGS job's (the default line)",
            "gcups_executed": cells_run / T / 1e9,
            "dp_reference_per_step": ndp_ref / NP, "dp_launched_per_step": ndp_run / NP,
            "per_step_figures_are": "per PASS (one batch of %d windows per GPU): dp_*_per_step, kernel_ms, algorithmic bytes, traffic" % nwin_batch,
            "kernel_ms": {"prepare": float(np.mean(prep_ms)), "seed": float(np.mean(seed_ms)), "seed_kernel": seedk_avg, "dp": dp_avg,
                          "finalize": float(np.mean(fin_ms)), "genotype": float(np.mean(gen_ms))},
            "dp_kernel_gcups": 4.0 * (dp_alg_bytes - 34 * dp_jobs0) / (dp_avg * 1e-3) / 1e9,
            "roofline": roof,
            "roofline_other": roof_other,
        }
        if roof_more:
            line["roofline_third"] = roof_more[0]
        if sweep_avg > 0:
            line["kernel_ms"]["sweep"], line["kernel_ms"]["pairs"] = sweep_avg, pairs_avg
        # the honest pair of whole-step figures next to the per-kernel one: HBM traffic of ALL kernels of a step (PMC) and SURVEY 8(d)'s
        # dedup'd algorithmic bytes per window (reads 2 L + 12, haplotypes 2 hapLen, 8 bytes per (haplotype, read) out), both over the
        # pipelined step time
        alg8d = 2 * int(hb.read_off[-1]) + 12 * hb.n_reads + 2 * int(hb.hap_off[-1]) + 8 * hb.n_pairs
        line["algorithmic_bytes_8d_per_step"] = int(alg8d)
        line["algorithmic_frac_8d"] = alg8d / (ms_pass * 1e-3) / 1e9 / HBM_PEAK_GBPS
        if pm.get("step_hbm_bytes"):
            line["step_traffic_bytes"] = int(pm["step_hbm_bytes"])
            line["step_hbm_frac"] = pm["step_hbm_bytes"] / (ms_pass * 1e-3) / 1e9 / HBM_PEAK_GBPS
            line["step_traffic_source"] = traffic_source
        line["record_gather"] = gather
        if world == 1 and not a.no_extras:
            # ---- the same batches with the shortcuts switched off (the library reads the switches per call)
            def mode(nsteps, dbl=dbs, hbl=hbs, **env):
                with Env(**env):
                    st_ = [eng.call_windows(d, want_stats=True) for d in dbl[:2]]
                    for i in range(S):
                        step(i, dbl, want_stats=False, asynchronous=not a.sync_entry)
                    sync_all()
                    pr = timed(2 * S, dbl) / (2 * S)
                    nsteps = max(nsteps, int(np.ceil(a.min_seconds / max(pr, 1e-9))))
                    t_ = timed(nsteps, dbl)
                cells = sum(st_[i % len(st_)].cells_reference for i in range(nsteps))
                run_ = sum(st_[i % len(st_)].cells_launched for i in range(nsteps))
                return {"gcups": cells / t_ / 1e9, "gcups_executed": run_ / t_ / 1e9, "ms_per_step": 1e3 * t_ / nsteps,
                        "dp_launched_per_step": float(st_[0].n_dp_launched), "dp_reference_per_step": float(st_[0].n_dp_reference)}
            n2 = max(20, a.steps // 8)
            alldp = mode(n2, PLAT_NO_UNGAPPED=1, PLAT_NO_EXACT=1)
            line["gcups_all_dp"] = alldp["gcups_executed"]
            line["all_dp"] = dict(alldp, what="both shortcuts off (PLAT_NO_UNGAPPED=1 PLAT_NO_EXACT=1): every reference DP is executed")
            line["exact_match_shortcut_only"] = dict(mode(n2, PLAT_NO_UNGAPPED=1), what="ungapped-alignment proof off")
            # ---- reads the ungapped proof likes less
            hh = synth.config2_hard(nwin_batch)
            dh = [eng.upload(hh)]
            hard = mode(n2, dh, [hh])
            hard_nolow = mode(n2, dh, [hh], PLAT_NO_NLOW=1)
            hard_alldp = mode(max(10, n2 // 2), dh, [hh], PLAT_NO_UNGAPPED=1, PLAT_NO_EXACT=1)
            line["hard_workload"] = {
                "what": "config-2 geometry, 1 % substitution errors, 5 % of the bases below Q20 incl. Q2 tails of 5..40 bases on a "
                        "tenth of the reads, 1e-4 sequencing indels per base (synth.config2_hard)",
                "gcups": hard["gcups"], "gcups_executed": hard["gcups_executed"], "ms_per_step": hard["ms_per_step"],
                "dp_launched_per_step": hard["dp_launched_per_step"], "dp_reference_per_step": hard["dp_reference_per_step"],
                "without_low_quality_valuation": hard_nolow, "all_dp": hard_alldp}
            del dh
            try:
                if a.no_other_configs:
                    raise RuntimeError("left out (--no-other-configs)")
                from tools import bench_other
                line["other_configs"] = bench_other.summary(eng)
            except Exception as exc:            # pragma: no cover
                line["other_configs"] = {"error": repr(exc)[:300]}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(hb)
        return line
    return None


if __name__ == "__main__":
    main()
