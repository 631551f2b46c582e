"""plat_variant_info_batch: the LOOPS of INFO ABPV / SbPval (beta-binomial CDF: the 3F2 series and the log-factorial sums, vcfutils.pyx:1156-1222,
platypusutils.pyx:178-315) and MMLQ (sorted(...)[n // 2]) on the device, the three libm calls of a CDF left to the caller -- so the values
carry the bits of the host's own functions (hostapi.computeAlleleBiasPValue / computeStrandBiasPValue, pinned by the reference's
`pvalue` goldens on the CPU suite).  Compared bit for bit."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pvalue_terms_and_median_bit_for_bit():
    import torch
    from platypus_amd import _lib, hostapi as H
    from platypus_amd.engine import Engine
    eng = Engine(0)
    rng = np.random.default_rng(77)
    n = 6000
    counts = np.zeros((n, 16), dtype=np.int64)
    tot = rng.integers(0, 400, n); tot[:50] = 0; tot[50:80] = rng.integers(3000, 6000, 30)          # incl. no reads, and depths beyond the table
    counts[:, 3] = tot
    counts[:, 4] = np.minimum(tot, (tot * rng.uniform(0, 1.1, n)).astype(np.int64))                  # TR_ab (incl. == total, >= half)
    nF, nR = rng.integers(0, 200, n), rng.integers(0, 200, n)
    nF[100:140] = 0; nR[140:180] = 0; nF[180:200] = rng.integers(2000, 5000, 20)                      # one strand empty; skewed -> big beta
    counts[:, 10], counts[:, 9] = nF, nR
    counts[:, 6] = (nF * rng.uniform(0, 1, n)).astype(np.int64); counts[:, 5] = (nR * rng.uniform(0, 1, n)).astype(np.int64)
    counts[200:230, 5] = 0; counts[200:230, 6] = 0                                                   # no variant reads on either strand
    counts[230:260, 5] = 0                                                                            # all variant reads forward (k == n when forward is used)
    nm = rng.integers(0, 150, n).astype(np.int32); nm[:100] = 0; nm[100:130] = 1; nm[130:160] = 64; nm[160:190] = 65
    off = np.concatenate([[0], np.cumsum(np.maximum(nm, 1))]).astype(np.int64)
    mq = rng.integers(0, 60, int(off[-1])).astype(np.int32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
    dc, do, dq, dn = d(counts.reshape(-1)), d(off[:-1].copy()), d(mq), d(nm)
    terms = torch.zeros(8 * n, dtype=torch.float64, device=eng.device)
    mm = torch.zeros(n, dtype=torch.int32, device=eng.device)
    _lib.check(eng.lib.plat_variant_info_batch(eng.ctx, n, dc.data_ptr(), do.data_ptr(), dq.data_ptr(), dn.data_ptr(), terms.data_ptr(), mm.data_ptr(), eng._stream()),
               "plat_variant_info_batch")
    eng._sync()
    t, m = terms.cpu().numpy().reshape(n, 8), mm.cpu().numpy()

    def cdf(x):
        return max(1e-30, 1.0 - math.exp((x[1] + math.log(x[2])) - x[3]))
    states = set()
    for v in range(n):
        c = counts[v]
        ab = H.computeAlleleBiasPValue(int(c[3]), int(c[4]))
        sb = H.computeStrandBiasPValue(int(c[10]), int(c[9]), int(c[6]), int(c[5]))
        for want, x, isab in ((ab, t[v, :4], True), (sb, t[v, 4:], False)):
            states.add((isab, int(x[0])))
            if x[0] == 0.0:
                got = x[1]
            elif x[0] == 1.0:
                p = cdf(x)
                got = min(p, 1.0 - p) if isab else p
            else:
                assert x[0] == 2.0                                        # beyond the log-factorial table: the caller's own loops
                continue
            assert np.float64(got).tobytes() == np.float64(want).tobytes(), (v, c.tolist(), got, want)
        k = int(nm[v])
        assert m[v] == (sorted(mq[off[v]:off[v] + k].tolist())[k // 2] if k else 100)
    assert states >= {(True, 0), (True, 1), (True, 2), (False, 0), (False, 1), (False, 2)}
