"""`bench.py --gpus N` as the driver starts it for N = 1 (no launcher): it must become N ranks by itself; and the strong-scaling
config-4 line (one region list, region i -> rank i % N, records gathered to rank 0 and merged, runner.py:454,470-500,301-352) run by
two gloo ranks must give the text one rank gives.  No GPU here: the launch path runs without device work (`--selftest-ranks`), the
config-4 line on tests/fakedev (the C ABI implemented with the oracle: test infrastructure, never shipped)."""
import json
import os
import socket
import subprocess
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REGION_KW = dict(snp_rate=3e-3, indel_rate=1e-3, read_len=100, depth=25, flank=500)


def test_bench_gpus_flag_starts_that_many_ranks():
    env = dict(os.environ, PLAT_DIST_BACKEND="gloo")
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-ranks"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.split("\n") if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["record_gather"]["ranks"] == 2 and line["record_gather"]["backend"] == "gloo"
    assert line["record_gather"]["records"] == 23 and line["in_order"]


def test_bench_gpus_8_is_eight_ranks():
    """The driver's largest form, `bench.py --gpus 8`: eight ranks (gloo here), the size all_gather, the point-to-point payloads and the ordered
    merge on rank 0 -- the launch path of an 8-GPU node without the devices."""
    env = dict(os.environ, PLAT_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--selftest-ranks"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.split("\n") if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["record_gather"]["ranks"] == 8 and line["record_gather"]["records"] == 23 and line["in_order"] and line["records_sum"] == 23


def test_launch_command_is_the_drivers_form():
    import bench
    cmd = bench.rank_launch_command(8, ["--gpus", "8", "--steps", "5"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")


def _args(regions, strong=False):
    return SimpleNamespace(regions=regions, steps=1, warmup=1, gpus=2, config=4, windows=None, strong=strong, weak=not strong and regions is None)


def _rank_worker(rank, world, port, out_path, regions=5, strong=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), PLAT_DIST_BACKEND="gloo", PLAT_CALLER_WORKERS="2", PLAT_CALLER_CHUNK="2", PLAT_BENCH_WGS_REGIONS_PER_GPU="3")
    import bench
    from tests import fakedev
    from tools import bench_other
    rk = bench.Ranks(world, need_gpu=False)
    line = bench_other.line_config4(_args(regions, strong), rk, lib=fakedev.fake_caller_lib(), region_len=3000, region_kw=REGION_KW)
    if rank == 0:
        json.dump(line, open(out_path, "w"))
    rk.close()


def test_config4_two_ranks_give_the_text_of_one_rank(tmp_path):
    import torch.multiprocessing as mp
    import bench
    from tests import fakedev
    from tools import bench_other
    fakedev.fake_caller_lib()                                                # built once, before the ranks race for it
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "line.json")
    mp.spawn(_rank_worker, args=(2, port, out), nprocs=2, join=True)
    two = json.load(open(out))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    os.environ.update(PLAT_CALLER_WORKERS="2", PLAT_CALLER_CHUNK="2")
    one = bench_other.line_config4(_args(5), bench.Ranks(1, need_gpu=False), lib=fakedev.fake_caller_lib(), region_len=3000, region_kw=REGION_KW)
    assert two["n_gpus"] == 2 and two["record_gather"]["ranks"] == 2 and two["scaling"] == "strong" and one["scaling"] == "strong"
    assert two["regions"] == one["regions"] == 5                             # --regions R: the SAME list whatever N
    assert two["scaling_efficiency_basis"]["scaling"] == "strong" and two["scaling_efficiency_basis"]["regions"] == 5 and two["scaling_efficiency_basis"]["ranks"] == 2
    assert one["n_gpus"] == 1 and one["record_gather"]["ranks"] == 1
    assert two["merged_text"] == one["merged_text"] and one["merged_text"].count("\n") > 10
    assert two["windows"] == one["windows"] and two["records"] == one["records"] == one["merged_text"].count("\n")
    # the reference's DPs of the called windows, counted during the untimed pass (plat_caller_count_cells), summed over the ranks
    assert two["dp_reference"] == one["dp_reference"] > 100 and two["pairs"] == one["pairs"] > 0 and "gcups" in one


def test_config4_scaling_label_follows_the_region_list(tmp_path):
    """--weak: the list grows with the job (a share per GPU) and the line says "weak"; the default (--strong): one list (8 shares) for every N and the
    line says "strong" -- the two N = 1 / N = 2 pairs an efficiency figure may be computed from are never mixed up."""
    import torch.multiprocessing as mp
    import bench
    from tests import fakedev
    from tools import bench_other
    lib = fakedev.fake_caller_lib()
    os.environ.update(PLAT_CALLER_WORKERS="2", PLAT_CALLER_CHUNK="2", PLAT_BENCH_WGS_REGIONS_PER_GPU="3")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    rk1 = bench.Ranks(1, need_gpu=False)
    weak1 = bench_other.line_config4(_args(None), rk1, lib=lib, region_len=3000, region_kw=REGION_KW)
    strong1 = bench_other.line_config4(_args(None, strong=True), rk1, lib=lib, region_len=3000, region_kw=REGION_KW)
    assert weak1["scaling"] == "weak" and weak1["regions"] == 3 and weak1["scaling_efficiency_basis"]["regions_per_rank"] == 3
    assert strong1["scaling"] == "strong" and strong1["regions"] == 24
    out = {}
    for name, (regions, strong) in dict(weak=(None, False), strong=(None, True)).items():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out[name] = str(tmp_path / (name + ".json"))
        mp.spawn(_rank_worker, args=(2, port, out[name], regions, strong), nprocs=2, join=True)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    weak2, strong2 = json.load(open(out["weak"])), json.load(open(out["strong"]))
    assert weak2["scaling"] == "weak" and weak2["regions"] == 6 and weak2["scaling_efficiency_basis"]["regions_per_rank"] == 3
    assert strong2["scaling"] == "strong" and strong2["regions"] == strong1["regions"] == 24
    assert strong2["merged_text"] == strong1["merged_text"]


def test_the_block_merge_writes_the_text_of_the_line_merge(tmp_path):
    """The exchange puts whole region blocks in (chromosome key, start) order without looking at a line (sharding.RegionTextExchange,
    plat_merge_region_blocks); runner.py:301-352 merges line by line (plat_merge_record_texts).  Same text, one rank and two; and a job
    whose regions overlap falls back to the line merge."""
    import torch.multiprocessing as mp
    import bench
    from platypus_amd import fastcaller as F, sharding
    from tests import fakedev
    from tools import bench_other
    lib = fakedev.fake_caller_lib()
    os.environ.update(PLAT_CALLER_WORKERS="2", PLAT_CALLER_CHUNK="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    rk1 = bench.Ranks(1, need_gpu=False)
    blocks = bench_other.line_config4(_args(13), rk1, lib=lib, region_len=3000, region_kw=REGION_KW)      # 13 regions: r0 .. r12 (runner.py:47-50 strips the letters of "CHR": the keys are the integers)
    os.environ["PLAT_BENCH_LINE_MERGE"] = "1"
    try:
        lines = bench_other.line_config4(_args(13), rk1, lib=lib, region_len=3000, region_kw=REGION_KW)
    finally:
        os.environ.pop("PLAT_BENCH_LINE_MERGE", None)
    assert blocks["record_gather"]["how"].startswith("one rank") and lines["record_gather"]["how"] == "line merge"      # (one rank, list in order: its text IS the merged text)
    # ... and the block copies themselves on one rank: the same 13 regions listed in REVERSE (the order of the blocks is then a real permutation)
    from platypus_amd import fastcaller as Fc
    import numpy as np
    txt = blocks["merged_text"].encode()
    starts = [0]
    cur = None
    for ln in txt.split(b"\n")[:-1]:
        c = ln.split(b"\t", 1)[0]
        if cur is not None and c != cur:
            starts.append(starts[-1] + run)
            run = 0
        run = (run if cur == c else 0) + len(ln) + 1
        cur = c
    lens = np.diff(np.array(starts + [len(txt)], dtype=np.int64))
    chroms = [txt[a:].split(b"\t", 1)[0].decode() for a in starts]
    rev = Fc.BlockOrder([[(sharding.chrom_key(c), 500, 3500) for c in reversed(chroms)]])
    assert rev.ok and not rev.identity
    back = b"".join(txt[a:a + n] for a, n in reversed(list(zip(starts, lens))))
    assert bytes(memoryview(rev.merge([back], [lens[::-1]], lib=lib))) == txt
    assert blocks["merged_text"] == lines["merged_text"] and blocks["merged_text"].count("\n") > 20
    order = [ln.split("\t")[0] for ln in blocks["merged_text"].split("\n") if ln]
    assert order == sorted(order, key=lambda c: sharding.chrom_key(c)) and order.index("r2") < order.index("r10")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "two.json")
    mp.spawn(_rank_worker, args=(2, port, out, 13, False), nprocs=2, join=True)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    two = json.load(open(out))
    assert two["record_gather"]["how"] == "region blocks" and two["merged_text"] == lines["merged_text"]
    # overlapping regions: no plan, the texts go through the line merge
    x = sharding.RegionTextExchange([[("c", 0, 100), ("c", 50, 150)]], lib=lib)
    assert not x.plan.ok
    t = b"c\t10\t.\tA\tC\nc\t60\t.\tA\tC\n" + b"c\t55\t.\tG\tT\n"
    merged = x.exchange(t, [len(t) - 13, 13])
    assert bytes(memoryview(merged)) == t                                  # (one text: the line merge keeps a single text's order)


def test_the_roofline_kernel_is_chosen_from_the_profile_only_when_it_is_this_builds(tmp_path, monkeypatch):
    """bench.py's roofline kernel: the first kernel of profiles/wgs_profile.json's rocprof ranking when that file carries THIS build's kernel-source hash,
    else the kernel with the largest summed live time; and config4_gcups hands out the file's traffic only in the first case."""
    from tools import bench_other
    kms = {"k_sb_variants": 50.0, "k_unpack_pieces": 40.0, "k_candidates": 30.0, "k_dp_jobs": 10.0, "other": 99.0}
    prof = {"kernel_source_hash": "feedfacefeedface", "ranking": ["k_nonexistent", "k_unpack_pieces", "k_sb_variants"], "ranking_source": "a test",
            "kernels": {"k_unpack_pieces": {"hbm_bytes_per_launch": 1234}}, "measured": {"date": "d", "commit": "c"}}
    monkeypatch.setattr(bench_other, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "wgs_profile.json").write_text(json.dumps(prof))
    monkeypatch.setattr(bench_other, "kernel_source_hash", lambda: "0123456789abcdef")
    k, how, order = bench_other.choose_roofline_kernel(kms)
    assert k == "k_sb_variants" and order[:3] == ["k_sb_variants", "k_unpack_pieces", "k_candidates"] and "other kernel sources" in how
    monkeypatch.setattr(bench_other, "kernel_source_hash", lambda: "feedfacefeedface")
    k, how, order = bench_other.choose_roofline_kernel(kms)
    assert k == "k_unpack_pieces" and "rocprofv3" in how and order[0] == "k_sb_variants"
    assert bench_other.choose_roofline_kernel({"other": 1.0})[0] is None


def test_cgroup_cpu_accounting_of_the_timed_region():
    """The line says what CPU time the box granted the job and how often the quota ran out inside the timed bracket (bench_other.cgroup_cpu_stat at
    both ends, cgroup_cpu_delta): differences of cpu.stat's counters, the quota in CPUs; None where the box has no cgroup-v2 files (this container)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_other as B
    a = dict(st=dict(nr_periods=100, nr_throttled=7, throttled_usec=1_000_000, usage_usec=50_000_000), quota_us=1_600_000, period_us=100_000)
    b = dict(st=dict(nr_periods=144, nr_throttled=37, throttled_usec=6_882_983, usage_usec=115_662_849), quota_us=1_600_000, period_us=100_000)
    d = B.cgroup_cpu_delta(a, b)
    assert d["quota_cpus"] == 16.0 and d["period_ms"] == 100.0 and d["periods"] == 44 and d["periods_throttled"] == 30
    assert abs(d["throttled_cpu_seconds"] - 5.882983) < 1e-9 and abs(d["cpu_seconds_used"] - 65.662849) < 1e-9
    nolimit = dict(st=dict(nr_periods=0), quota_us=None, period_us=100_000)
    assert B.cgroup_cpu_delta(nolimit, nolimit)["quota_cpus"] is None
    assert B.cgroup_cpu_delta(None, b) is None and B.cgroup_cpu_delta(a, None) is None
    st = B.cgroup_cpu_stat()                                                 # whatever this machine has: a record with the three parts, or None
    assert st is None or {"st", "quota_us", "period_us"} <= set(st)
