"""The three Python-2 behaviours that reach the VCF text (round, str(float), iteration order of a set of filter names), pinned
against PUBLISHED CPython 2.7 facts instead of against each other, for both restatements: platypus_amd/vcfrecords.py (Python
region loop) and platypus_amd/csrc/host/records.hpp (native region loop).

Sources of the literal vectors (no Python 2 interpreter exists in this image):
  * string hash: Objects/stringobject.c `string_hash` of CPython 2.7 on a 64-bit build with hash randomisation off (the default in
    2.7): hash('') == 0 by definition; hash('a') == 12416037344 is the value quoted wherever -R / PYTHONHASHSEED is explained;
    hash('abc') == 1453079729188098211 and hash('hello') == 840651671246116861 are the 64-bit values printed by 2.7 in the
    many "why does hash() differ between 32 and 64 bit / Python 2 and 3" write-ups.
  * set order: Objects/setobject.c (8 slots, probe i = 5i + perturb + 1, perturb >>= 5, resize x4 at 2/3): the well-known
    list(set('abc')) == ['a', 'c', 'b'], list(set('abcd')) == ['a', 'c', 'b', 'd'], list(set('abcde')) == ['a', 'c', 'b', 'e', 'd'].
  * str(float): Python 2.7 tutorial, "Floating Point Arithmetic: Issues and Limitations": str(math.pi) == '3.14159265359',
    `print 0.1 + 0.2` shows 0.3, str(1.0/3) == '0.333333333333' (str() uses 12 significant digits, "%.12g").
  * round(): Python 2.7 library reference, built-in round(): "round(0.5) is 1.0 and round(-0.5) is -1.0" (ties away from zero)
    and its note "round(2.675, 2) gives 2.67" (the exact binary value decides)."""
import ctypes as C
import itertools
import math

import pytest

from platypus_amd import vcfrecords as V

HASHES = {"": 0, "a": 12416037344, "b": 12544037731, "abc": 1453079729188098211, "hello": 840651671246116861}
SET_ORDERS = {"abc": list("acb"), "abcd": list("acbd"), "abcde": list("acbed")}
STRS = [(math.pi, "3.14159265359"), (0.1 + 0.2, "0.3"), (1.0 / 3, "0.333333333333"), (1e16, "1e+16"), (123456789012.0, "123456789012.0"),
        (1234567890123.0, "1.23456789012e+12"), (1e-5, "1e-05"), (100.0, "100.0"), (0.0001, "0.0001"), (-0.0, "-0.0"), (27.9824009652, "27.9824009652")]
ROUND0 = [(0.5, 1.0), (-0.5, -1.0), (1.5, 2.0), (2.5, 3.0), (-2.5, -3.0), (0.49999999999999994, 0.0)]
ROUND2 = [(2.675, 2.67), (0.125, 0.13), (0.375, 0.38), (0.625, 0.63), (0.875, 0.88), (-0.125, -0.13), (1.005, 1.0), (0.005, 0.01), (0.015, 0.01),
          (1e-3, 0.0), (36.965, 36.97), (-42.765, -42.77), (1234.5650000000001, 1234.57), (0.0, 0.0)]
FILTER_NAMES = ["SC", "QD", "HapScore", "MQ", "strandBias", "alleleBias", "badReads", "Q20"]     # the order vcfFILTER appends them in (vcfutils.pyx:1502-1627)


@pytest.fixture(scope="module")
def native():
    from platypus_amd import fastcaller as F
    F.build()
    lib = C.CDLL(F.LIB_PATH)
    lib.plat_caller_debug_round2.restype = lib.plat_caller_debug_round0.restype = C.c_double
    lib.plat_caller_debug_round2.argtypes = lib.plat_caller_debug_round0.argtypes = [C.c_double]
    lib.plat_caller_debug_string_hash.restype = C.c_ulonglong
    lib.plat_caller_debug_string_hash.argtypes = [C.c_char_p]
    lib.plat_caller_debug_str.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    lib.plat_caller_debug_set_order.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    return lib


def _n_str(lib, x):
    b = C.create_string_buffer(128)
    lib.plat_caller_debug_str(x, b, 128)
    return b.value.decode()


def _n_order(lib, names):
    b = C.create_string_buffer(4096)
    lib.plat_caller_debug_set_order("\n".join(names).encode(), b, 4096)
    return b.value.decode().split("\n") if b.value else []


def test_string_hash_matches_cpython27_values(native):
    for s, h in HASHES.items():
        assert V._py2_string_hash(s) == h % (1 << 64) and native.plat_caller_debug_string_hash(s.encode()) == h % (1 << 64), s


def test_set_iteration_order_matches_cpython27_examples(native):
    for s, order in SET_ORDERS.items():
        assert V.py2_set_order(list(s)) == order and _n_order(native, list(s)) == order


def test_str_of_floats_matches_python27_documentation(native):
    for x, t in STRS:
        assert V.py2_str(x) == t and _n_str(native, x) == t, x


def test_round_matches_python27_documentation(native):
    for x, r in ROUND0:
        assert V.py2_round(x) == r and native.plat_caller_debug_round0(x) == r, x
    for x, r in ROUND2:
        assert V.py2_round(x, 2) == r and native.plat_caller_debug_round2(x) == r, x
    # and the two restatements agree on a sweep of values around ties and on random doubles
    import random
    rnd = random.Random(5)
    xs = [k / 8.0 for k in range(-4001, 4001, 2)] + [k / 1000.0 for k in range(-3000, 3000)] + [rnd.uniform(-500, 500) for _ in range(20000)] + \
         [rnd.uniform(-1, 1) * 10 ** rnd.randint(-12, 12) for _ in range(5000)]
    for x in xs:
        assert V.py2_round(x, 2) == native.plat_caller_debug_round2(x), x
        assert V.py2_str(x) == _n_str(native, x), x


def test_every_filter_set_vcfFILTER_can_emit_has_one_order_in_both_restatements(native, golden_dir):
    """All 256 subsets of the FILTER names, inserted in the order vcfFILTER appends them (every variant of a VCF line carries the same
    list, so repeats do not change the order of first insertions).  The committed table was produced by the restatement whose
    ingredients are pinned above; a change of either restatement shows up as a diff against it."""
    import json, os
    table = json.load(open(os.path.join(golden_dir, "filter_set_orders_py27.json")))
    n = 0
    for r in range(len(FILTER_NAMES) + 1):
        for sub in itertools.combinations(FILTER_NAMES, r):
            names = list(sub)
            got = V.py2_set_order(names * 2)                 # (a two-variant line repeats the list)
            assert got == V.py2_set_order(names) == _n_order(native, names) == table[",".join(names)], names
            assert sorted(got) == sorted(names)
            n += 1
    assert n == 256 and len(table) == 256


def test_tuple_hash_and_dict_order_of_variant_keys(native):
    """The candidates of a region leave Python-2 dictionaries keyed by Variant objects, hash((refName, refPos, removed, added))
    (variant.pyx:270-280; variantcaller.pyx:457, variant.pyx:747-751): tupleobject.c's hash and dictobject.c's slot order, both
    restatements.  Published CPython 2.7 (64-bit) facts: hash(()) == 3527539, hash((1, 2, 3)) == 2528502973977326415 (the value every
    "is hash() stable" discussion prints), and a dict of 'a', 'b', 'c' iterates a, c, b (as the set does)."""
    native.plat_caller_debug_tuple_hash.restype = C.c_ulonglong
    native.plat_caller_debug_tuple_hash.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    native.plat_caller_debug_variant_hash.restype = C.c_ulonglong
    native.plat_caller_debug_variant_hash.argtypes = [C.c_char_p, C.c_longlong, C.c_char_p, C.c_char_p]
    native.plat_caller_debug_dict_slot_order.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.POINTER(C.c_int)]

    def n_tuple(hs):
        return native.plat_caller_debug_tuple_hash((C.c_ulonglong * max(1, len(hs)))(*hs), len(hs))

    def n_order(hs):
        out = (C.c_int * max(1, len(hs)))()
        native.plat_caller_debug_dict_slot_order((C.c_ulonglong * max(1, len(hs)))(*hs), len(hs), out)
        return list(out)[:len(hs)]
    for items, want in (([], 3527539), ([1, 2, 3], 2528502973977326415)):
        assert V.py2_tuple_hash(items) == want == n_tuple(items)
    for keys, want in (("abc", "acb"), ("abcd", "acbd"), ("abcde", "acbed")):
        hs = [V._py2_string_hash(k) for k in keys]
        assert "".join(keys[i] for i in V.py2_dict_slot_order(hs)) == want == "".join(keys[i] for i in n_order(hs))
    ints = [5, 1, 9, 17, 3, 1000003, 64, 8, 16, 24, 32, 40]
    assert [ints[i] for i in V.py2_dict_slot_order(ints)] == V.py2_dict_order(ints) == [ints[i] for i in n_order(ints)]
    # both restatements on random keys, through several resizes
    import random
    rnd = random.Random(11)
    for n in (1, 5, 6, 21, 22, 85, 86, 341, 342, 3000):
        vs = [("r%d" % rnd.randint(0, 3), rnd.randint(0, 10 ** 6), "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 4))),
               "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 4)))) for _ in range(n)]
        vs = list(dict.fromkeys(vs))
        hs = [V.py2_variant_hash(*v) for v in vs]
        assert hs == [native.plat_caller_debug_variant_hash(v[0].encode(), v[1], v[2].encode(), v[3].encode()) for v in vs]
        assert V.py2_dict_slot_order(hs) == n_order(hs) and sorted(n_order(hs)) == list(range(len(vs)))
        # `public int hashValue` (variant.pxd:31): the dictionary sees the sign-extended low 32 bits of the tuple hash
        for v, h in zip(vs, hs):
            full = V.py2_tuple_hash([V._py2_string_hash(v[0]), v[1], V._py2_string_hash(v[2]), V._py2_string_hash(v[3])])
            low = full & 0xFFFFFFFF
            want = low if low < 1 << 31 else low + (((1 << 32) - 1) << 32)
            assert h == (want if want != (1 << 64) - 1 else (1 << 64) - 2)
            assert (h >> 31) in (0, (1 << 33) - 1)


def test_native_variant_prior_matches_reference_golden(native, golden_dir):
    """Variant.calculatePrior of the native host (variants.hpp: indelPrior over the bit-plane form of tandem.c's annotate for the
    200-base context) for the 2400 indels in repeats / plain sequence / at contig ends whose priors the reference's own text gave."""
    import gzip, json, os
    native.plat_caller_debug_prior.restype = C.c_double
    native.plat_caller_debug_prior.argtypes = [C.c_char_p, C.c_longlong, C.c_longlong, C.c_char_p, C.c_char_p]
    g = json.load(gzip.open(os.path.join(golden_dir, "indelprior_cases.json.gz"), "rt"))
    n = 0
    for c in g["priors"]:
        ref = c["ref"].encode()
        for v in c["variants"]:
            assert native.plat_caller_debug_prior(ref, len(ref), v["pos"], v["removed"].encode(), v["added"].encode()) == v["prior"], v
            n += 1
    assert n == 2400


def test_fixed_point_text_equals_printf(native):
    """The PP ("%.0f") and FR ("%1.4f") fields are written without printf where the rounding cannot be in doubt: the same characters as
    Python's % operator (the C library's printf) for ties, values next to ties, roll-overs, large, negative and non-finite values."""
    import random
    native.plat_caller_debug_fixed.argtypes = [C.c_double, C.c_int, C.c_char_p, C.c_size_t]
    rng = random.Random(11)
    xs = [0.0, -0.0, 0.5, 1.5, 2.5, 3.5, 0.49999999999999994, 0.99995, 0.99994999, 0.00005, 0.00015, 0.12345, 0.12355, 0.5 / 3, 1.0, 1e-300, 1e5, 99999.99995,
          1e9, 1e15, 1e22, -3.7, 2500.5, 2501.5, float("inf"), float("nan"), 12345.67895, 0.30000000000000004, 1.00005, 7.00015]
    xs += [rng.random() for _ in range(20000)] + [rng.random() * 3000 for _ in range(20000)]
    xs += [(k + 0.5) / 10000.0 for k in range(0, 20000, 7)] + [k + 0.5 for k in range(0, 3000)]
    xs += [math.nextafter((k + 0.5) / 10000.0, d) for k in range(0, 20000, 13) for d in (0.0, 1.0)]
    buf = C.create_string_buffer(512)
    for x in xs:
        for dec, fmt in ((0, "%.0f"), (4, "%1.4f")):
            native.plat_caller_debug_fixed(x, dec, buf, 512)
            assert buf.value.decode() == fmt % x, (x, dec)
