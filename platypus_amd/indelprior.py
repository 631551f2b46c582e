"""The indel prior of Variant.calculatePrior: local tandem-repeat annotation + the tract-length error model.

    annotate(sequence, sizes, displacements, length)     src/c/tandem.c:124-262  (twobit :11-57, foundmatch :87-121)
    calculate_size_and_displacement                      src/cython/cerrormodel.pyx:23-36
    Variant.indelPrior, indel_prior_model                src/cython/variant.pyx:68-95,146-217

Host logic (one 200-base context per indel).  `annotate` is pinned against a build of the UNMODIFIED tandem.c
(tests/test_oracle.py) and by tests/golden/indelprior_cases.json.gz; `indelPrior` by the same fixture (outputs of the
reference's own text on top of tandem.c)."""
import numpy as np

MAX_UNIT_LENGTH = 12                                                             # tandem.c:6
MIN_PARTIAL_MATCH = 5                                                            # tandem.c:7

# phred+33 strings of the model, one per repeat-unit length (variant.pyx:68-91): entry [tract length - 1]
indel_prior_model = {
    1: "LIGC@:62/-*'&%$",
    2: "LIGDB@><9630.,+**)(''&&%%%$$$",
    3: "LIGA@B@><;8763220/.-,+++)*))(((''''&&&&&&%%%%%%%%$$$$$$$",
    4: "LIGA@???=<886533210/.--,+**))))((('''''&&&&&&&&%%%%%%%%%%%$$$$$$$$",
    5: "LIGA@??>=>=;966543210///-,,++*",
    6: "LIGA@??>>=<=;:764532210/----,++",
    7: "LIGA@??>>==<;;987543210/....-,,,++++",
    8: "LIGA@??>>==<<;9876432200/..--,,,+++",
    9: "LIGA@??>>==<<;;9966432100//../..----,,,,,++++++",
    10: "LIGA@??>>==<<;;:986432110//..----,,,,++++",
    11: "LIGA@??>>==<<<;;:87642210////..--,,,,,+++",
    12: "LIGA@??>>==<<<;;;:986532110000/...-----,,,,,+++++",
    13: "LIGA@??>>==<<<;;;::987543111000/////.......--------,,,,,,,,,,,,,+++++++++",
    14: "LIGA@??>>==<<<;;;::987642210/0/.....-------,,,,,,,,+++++++",
    15: "LIGA@??>>==<<<;;;;::988754322110000////////.......------------,,,,,,,,,,,,,,,,,++++++++++",
    16: "LIGA@??>>==<<<;;;;:::98765321110////........-------,,,,,,,,,,,,,,+++++++++",
    17: "LIGA@??>>==<<<;;;;::::988764433211110000000///////.............-----------------,,,,,,,,,,,,,,,,,,,",
    18: "LIGA@??>>==<<<;;;:::::998875433221111000000///////.............-----------------,,,,,,,,,,,,,,,,,,,",
    19: "LIGA@??>>==<<<;;;;::::999887654433222221111111100000000//////////////..................------------",
    20: "LIGA@??>>==<<<;;;;::::9999876543322111000000///////............-----------------,,,,,,,,,,,,,,,,,,,",
    21: "LIGA@??>>==<<<;;;;::::9999988765544433322222221111111100000000000000//////////////////.............",
    22: "LIGA@??>>==<<<;;;;::::9999987765432221000000////////...........-----------------,,,,,,,,,,,,,,,,,,,",
    23: "LIGA@??>>==<<<;;;;::::9999998776543322111100000000////////................-------------------,,,,,,",
    24: "LIGA@??>>==<<<;;;;::::9999998887654433322111111100000000/////////////...................-----------",
}
complex_deletion_prior = 5e-5                                                    # variant.pyx:94
complex_insertion_prior = 5e-6                                                   # variant.pyx:95


def _rate(size, displacement):
    """tandem.c:61-70: -10*phred guess of the indel rate of a tract."""
    if displacement == 1:
        return -360 + 24 * size
    if displacement == 2:
        return -327 + 15 * size
    if displacement == 3:
        return -291 + 8 * size
    return -282 + 6 * size


def _codes(seq, n):
    """Two bits per nucleotide as tandem.c's twobit(): A/C/G/T (either case) = 0..3, anything else a position-dependent
    pseudo-random code, zeros (A) past the end of the string."""
    b = np.frombuffer(seq, dtype=np.uint8) & 0xDF
    idx = np.arange(len(b), dtype=np.int64)
    noise = (((idx % 257) * (1 + idx % 257)) // 2 + (idx % 5)) % 4
    c = np.where(b == ord("A"), 0, np.where(b == ord("C"), 1, np.where(b == ord("G"), 2, np.where(b == ord("T"), 3, noise))))
    return np.concatenate([c, np.zeros(n - len(b), dtype=np.int64)])


def annotate(sequence, markfull=True):
    """-> (sizes, displacements): per position the length of the local repeat tract and its unit length, as tandem.c's
    annotate() leaves them (markfull = the `length < 0` mode calculate_size_and_displacement(seq, True) uses).

    The C code compares 64 nucleotides at a time in groups of 4 start positions; what it computes per start position p
    (group start g = p & ~3) and unit d is the distance from p to the first mismatch between the sequence and itself
    shifted by d, looking only as far as g + 64 (g + 32 when the shifted second word would start past the end)."""
    L = len(sequence)
    sizes, disps = [1] * L, [1] * L
    if L == 0:
        return sizes, disps
    ext = L + 80 + MAX_UNIT_LENGTH
    code = _codes(sequence, ext + MAX_UNIT_LENGTH)
    nxt = {}
    for d in range(1, MAX_UNIT_LENGTH):
        mism = code[:ext] != code[d:ext + d]
        where = np.where(mism, np.arange(ext), ext)
        nxt[d] = np.minimum.accumulate(where[::-1])[::-1]                         # next mismatch at or after each index
    for g in range(0, L, 4):
        for d in range(1, MAX_UNIT_LENGTH):
            if g + d >= L:
                break
            second = g + d + 32 < L
            for k in range(4):
                p = g + k
                m = int(nxt[d][p]) - g                                            # first mismatch >= p, relative to the group
                if m > 31:
                    m = min(m, 64) if second else 32
                size, pos = m - k, p
                # foundmatch (tandem.c:87-121)
                if pos + d + size > L:
                    size = L - d - pos
                size += d
                if size < d + min(MIN_PARTIAL_MATCH, d):
                    continue
                if _rate(sizes[pos], disps[pos]) < _rate(size, d):
                    sizes[pos], disps[pos] = size, d
                    if markfull:
                        for i in range(pos + 1, min(L, pos + size)):
                            sizes[i], disps[i] = size, d
    return sizes, disps


def indelPrior(variant, refFile, indel_length_and_type):
    """variant.pyx:146-217: the smaller of the model's priors for the repeat tracts at the two bases next to the indel; for
    tracts of length <= 3 (no repeat to speak of) a length-dependent prior for complex insertions / deletions instead."""
    context = 100
    leftPos = max(0, variant.refPos - context)
    rightPos = variant.refPos + context
    rel = variant.refPos - leftPos
    try:
        sequence = refFile.getSequence(variant.refName, leftPos + 1, rightPos + 1)
    except IndexError:
        sequence = b""
    sizes, disps = annotate(sequence, True)
    prior, tract = ord(indel_prior_model[1][0]) - 33, 255
    for i in (rel - 1, rel):
        # (the reference reads its NUL-terminated annotation strings here without a bounds check; inside the string this is it)
        disp = disps[i] if 0 <= i < len(disps) else 0
        if disp in indel_prior_model:
            size = min(sizes[i], len(indel_prior_model[disp]))
            q = ord(indel_prior_model[disp][size - 1]) - 33
            if q < prior:
                prior, tract = q, size
    dprior = pow(0.1, prior / 10.0)
    if tract <= 3:
        n = indel_length_and_type
        if n < 0:
            dprior = complex_deletion_prior * pow(0.75, (-n) - 1) * (1.0 - 0.75)
        else:
            dprior = complex_insertion_prior * pow(0.75, n - 1) * (1.0 - 0.75) * pow(0.33, n)
    return dprior
