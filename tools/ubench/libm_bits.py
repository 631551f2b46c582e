"""How often do the device's double-precision log10 / exp / pow / log differ from the host libm's (glibc) in the last bits?  The numeric
half of the VCF record layer (vcfutils.pyx:338-599,1226-1627) is built from these; the record text is bit-for-bit only while they agree.
torch's ROCm kernels call the same OCML functions a HIP kernel would.   usage: python tools/ubench/libm_bits.py [n]"""
import json
import sys

import numpy as np
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(1)
out = {}
cases = {
    "log10(p), p in (1e-300, 1)": (np.log10, torch.log10, 10.0 ** rng.uniform(-300, 0, n)),
    "log10(1 - p), p in (0, 1)": (np.log10, torch.log10, 1.0 - rng.uniform(0, 1, n)),
    "log(x), x in (1e-12, 1e6)": (np.log, torch.log, 10.0 ** rng.uniform(-12, 6, n)),
    "exp(x), x in (-700, 0)": (np.exp, torch.exp, rng.uniform(-700, 0, n)),
    "10 ** x, x in (-30, 0)": (lambda a: np.power(10.0, a), lambda t: torch.pow(10.0, t), rng.uniform(-30, 0, n)),
    "sqrt(x)": (np.sqrt, torch.sqrt, rng.uniform(0, 1e4, n)),
}
for name, (hf, df, x) in cases.items():
    h = hf(x)
    d = df(torch.from_numpy(x).cuda()).cpu().numpy()
    hb, db = h.view(np.int64), d.view(np.int64)
    diff = np.abs(hb - db)
    # would the 2-decimal text of -10 * value (a phred) differ?
    ht, dt = np.round(-10.0 * h, 2), np.round(-10.0 * d, 2)
    out[name] = {"n": n, "differ_in_bits": int((diff != 0).sum()), "max_ulp": int(diff.max()), "two_decimal_texts_differ": int((ht != dt).sum())}
print(json.dumps(out))
