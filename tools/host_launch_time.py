"""Host-side cost of issuing one step (plat_align_window_batch_async + plat_genotype_window_batch through ctypes) vs the GPU time
of the step: if the host needs as long as the device, the device idles no matter how fast its kernels are."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
from platypus_amd import synth                 # noqa: E402
from platypus_amd.engine import Engine         # noqa: E402

S = 3
engs = [Engine(0) for _ in range(S)]
dbs = [e.upload(synth.config2(10000, seed=2002 + i)) for i, e in enumerate(engs)]
streams = [torch.cuda.Stream(device=engs[0].device) for _ in range(S)]
for i in range(6):
    with torch.cuda.stream(streams[i % S]):
        engs[i % S].call_windows(dbs[i % S], want_stats=False, asynchronous=True)
torch.cuda.synchronize()
N = 60
t0 = time.perf_counter()
for i in range(N):
    with torch.cuda.stream(streams[i % S]):
        engs[i % S].call_windows(dbs[i % S], want_stats=False, asynchronous=True)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host time to issue one step: %.3f ms; wall time per step incl. GPU: %.3f ms" % (1e3 * t_issue / N, 1e3 * t_all / N))
