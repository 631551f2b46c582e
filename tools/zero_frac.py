import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from platypus_amd import synth
from platypus_amd.engine import Engine
eng=Engine(0)
hb=synth.config2(2000)
db=eng.upload(hb); st=eng.call_windows(db); eng.synchronize()
sc=db.score.cpu().numpy()[:hb.n_pairs]
al=sc>=0
print("pairs",len(sc),"aligned",al.sum(),"score==0 frac of aligned", (sc[al]==0).mean())
# duplicates: per window, per read: number of distinct scores among haps is a lower bound on distinct slices
dist=0; tot=0
for w in range(hb.n_windows):
    H=hb.win_hap_begin[w+1]-hb.win_hap_begin[w]; R=hb.win_read_begin[w+1]-hb.win_read_begin[w]
    s=sc[hb.pair_off[w]:hb.pair_off[w+1]].reshape(H,R)
    for r in range(R):
        col=s[:,r]
        if col[0]<0: continue
        nz=col[col>0]
        dist+=len(set(nz.tolist())); tot+=H
print("lower bound on needed DPs (distinct nonzero scores per read) / aligned pairs:", dist/tot)
