import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, cases, oracle_lib as ol
from smcsmc_amd import ParticleFilter
def bits(a): return np.ascontiguousarray(a,dtype=np.float64).view(np.uint64)
n,E,Np,seed = 2,1,300,1
model = cases.make_model(n=n,E=E,L=1.5e5)
segs = cases.make_segments(model, seed=seed, max_seg_len=5000)
o = ol.Oracle(model, Np, seed=seed); o.init_prior(0.0); si=o.pack_segments(model,segs)
g = ParticleFilter(model, Np, seed=seed); g.init_prior(0.0); g.load_segments(segs)
S=len(segs['start'])
for s in range(S):
    o.update_segment(si, s); 
    g.update_segment(s); 
    pos=min(segs['start'][s]+segs['length'][s], model['loci_length'])
    o.count(pos); g.count(s)
    r=o.resample(pos); g.resample(s)
    po,pg=o.particles(),g.particles()
    bad = [k for k in ('heights','w_post','w_pilot','next_base') if not (bits(po[k])==bits(pg[k])).all()]
    if bad or not (po['children']==pg['children']).all():
        print("seg",s,"resampled",r,"bad",bad, "state", segs['state'][s], segs['alleles'][s], segs['length'][s])
        for k in bad:
            idx=np.nonzero(bits(po[k]).reshape(Np,-1)!=bits(pg[k]).reshape(Np,-1))[0]
            print(k, len(idx), idx[:10], po[k].reshape(Np,-1)[idx[:3]], pg[k].reshape(Np,-1)[idx[:3]])
        tr=o.trace(); tg=g.trace()
        print(tr['T'][-3:], tg['T'][-3:], tr['ess'][-3:], tg['ess'][-3:])
        break
else:
    print("all equal")
