"""CPU tests of the N>1 path: region->rank assignment, variable-length gather (gloo, world_size 2) and the
ordered merge (runner.py:301-352 semantics)."""
import os
import socket

import numpy as np
import pytest

from platypus_amd import sharding


def test_round_robin_assignment_matches_reference_rule():
    # runner.py:473-474: regionsForEachProcess[index % nCPU].append(region)
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            mine = sharding.regions_for_rank(31, r, world)
            assert all(i % world == r for i in mine)
            seen += mine
        assert sorted(seen) == list(range(31))


def test_chrom_key_and_merge_order():
    a = [("1", 5, "a"), ("2", 1, "b"), ("X", 3, "c")]
    b = [("1", 7, "d"), ("chr2", 0, "e"), ("10", 4, "f")]
    assert sharding.chrom_key("chr2") == sharding.chrom_key("2")
    merged = sharding.merge_record_streams([a, sorted(b, key=lambda r: (sharding.chrom_key(r[0]), r[1]))])
    assert merged == ["a", "d", "e", "b", "f", "c"]


def _worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    regions = list(range(11))
    mine = sharding.regions_for_rank(len(regions), rank, world)
    recs = [("1", 1000 * g, "region%d" % g) for g in mine] if rank != 1 or True else []
    if rank == 1:
        recs = recs[:2]          # ragged payload sizes
    got = sharding.gather_records(sharding.encode_records(recs), dist)
    if rank == 0:
        streams = [sharding.decode_records(p) for p in got]
        merged = sharding.merge_record_streams(streams)
        open(os.path.join(tmp, "merged.txt"), "w").write("\n".join(merged))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_gather_world2(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    merged = open(tmp_path / "merged.txt").read().split("\n")
    # rank 0 owns even regions (all 6), rank 1 sent only its first two (regions 1 and 3)
    assert merged == ["region0", "region1", "region2", "region3", "region4", "region6", "region8", "region10"]


def test_record_roundtrip():
    recs = [("1", 5, "1\t5\t2\t-1.00,-2.50"), ("X", 7, "x")]
    assert sharding.decode_records(sharding.encode_records(recs)) == recs
    assert sharding.decode_records(b"") == []


def test_vcf_text_to_merge_records():
    text = "r1\t101\t.\tA\tC\t50\tPASS\tx\tGT\t0/1\nr0\t7\t.\tG\tT\t9\tQ20\ty\tGT\t1/1\n#comment\n"
    recs = sharding.records_from_vcf_text(text)
    assert [(c, p) for c, p, _ in recs] == [("r1", 100), ("r0", 6)]
    merged = sharding.merge_record_streams([[recs[1]], [recs[0]]])
    assert merged == [text.split("\n")[1], text.split("\n")[0]]
    assert sharding.chrom_key("r10") > sharding.chrom_key("r9")             # numeric once the letters of "CHR" are stripped


def test_native_merge_of_record_texts_equals_the_python_merge():
    """plat_merge_record_texts (the (chrom, pos) merge of runner.py:301-352 in libplat_caller.so, what rank 0 runs on the gathered
    texts) against merge_record_streams: integer-like names with and without a chr prefix, names that are not integers, header and
    empty lines, empty streams (no two streams hold the same (chrom, pos): regions belong to one rank, and the reference leaves such
    ties to the shape of its heap)."""
    from platypus_amd import fastcaller as F
    from tests import fakedev
    lib = fakedev.fake_caller_lib()
    rng = np.random.default_rng(3)
    names = ["1", "2", "chr2", "10", "chr11", "X", "Y", "MT", "r7", "r12", "hs37d5", "GL000207.1", "chrR3", "c-5"]
    for trial in range(30):
        n = int(rng.integers(1, 6))
        streams, taken = [], set()
        for k in range(n):
            recs = {(str(rng.choice(names)), int(rng.integers(0, 50))) for _ in range(int(rng.integers(0, 40)))}
            recs = {r for r in recs if (sharding.chrom_key(r[0]), r[1]) not in taken}
            taken |= {(sharding.chrom_key(r[0]), r[1]) for r in recs}
            recs = sorted(recs, key=lambda r: (sharding.chrom_key(r[0]), r[1]))
            streams.append([(c, p, "%s\t%d\t.\tA\tC\t%d" % (c, p + 1, k)) for c, p in recs])
        want = "".join(ln + "\n" for ln in sharding.merge_record_streams(streams))
        texts = []
        for k, s in enumerate(streams):
            t = "".join(ln + "\n" for _, _, ln in s)
            if k == 0:
                t = "#header\n\n" + t
            if k == 1 and t:
                t = t[:-1]                                                  # no newline at the end
            texts.append(t.encode())
        assert F.merge_record_texts(texts, lib=lib) == want
    assert F.merge_record_texts([], lib=lib) == "" and F.merge_record_texts([b"", b""], lib=lib) == ""


def test_native_merge_on_a_text_large_enough_for_its_threads():
    """Above a few MB the merge scans its texts in slices and copies the blocks on several threads: ~13 MB in three texts (regions dealt out
    round robin, as ranks get them) against the Python merge."""
    from platypus_amd import fastcaller as F
    from tests import fakedev
    lib = fakedev.fake_caller_lib()
    pad = "X" * 90
    streams = [[], [], []]
    for g in range(900):
        for p in range(0, 5000, 37):
            streams[g % 3].append(("r%d" % g, p, "r%d\t%d\t.\tA\tC\t%s" % (g, p + 1, pad)))
    want = "".join(ln + "\n" for ln in sharding.merge_record_streams(streams))
    texts = ["".join(ln + "\n" for _, _, ln in s).encode() for s in streams]
    assert sum(len(t) for t in texts) > 12 << 20
    got = F.merge_record_texts(texts, lib=lib, raw=True)
    assert got == want.encode()
