"""As l2_replay.py, with the workgroups of a group lagging behind each other by up to `skew` band steps and three node
passes in a row: cfg 3, effective capacity 3 MiB: skew 0 -> 1.13x, 2 -> 1.19x, 4 -> 1.44x (the measured figure), 8 -> 1.74x."""
import numpy as np, sys
from collections import OrderedDict
rp=np.fromfile('/tmp/sim/rp.bin',dtype=np.int32); col=np.fromfile('/tmp/sim/col.bin',dtype=np.int32)
N=len(rp)-1
rng=np.random.default_rng(1)
def run(cap_units, skew_bands, nodes=3, band_wg=16, wgs=32, slices=8, mat_units=0.16, g=3):
    # one XCD group g, `nodes` consecutive node passes; WG w lags by lag[w] steps (random, up to skew_bands), re-drawn smoothly
    sl=(N+slices-1)//slices
    lo=g*sl; hi=min(N,lo+sl)
    steps_per_node=(hi-lo+band_wg*wgs-1)//(band_wg*wgs)
    total_steps=steps_per_node*nodes
    cache=OrderedDict(); used=0.0; miss=0
    def touch(key,size):
        nonlocal used, miss
        if key in cache:
            cache.move_to_end(key); return True
        cache[key]=size; used+=size
        while used>cap_units:
            k,s=cache.popitem(last=False); used-=s
        return False
    lag=rng.integers(0,skew_bands+1,wgs)
    xm=0
    for T in range(total_steps+skew_bands):
        order=rng.permutation(wgs)
        for w in order:
            t=T-lag[w]
            if t<0 or t>=total_steps: continue
            node=t//steps_per_node; s=t%steps_per_node
            r0=lo+s*band_wg*wgs+w*band_wg
            for i in range(r0,min(hi,r0+band_wg)):
                touch(('m',i),mat_units)
                if not touch((node,i),1.0): xm+=1
                for k in range(rp[i],rp[i+1]):
                    j=col[k]
                    if j!=i and not touch((node,j),1.0): xm+=1
                touch(('y',node,i),1.0)
    rows=(hi-lo)*nodes
    return xm/rows
for cap in (4096,3072):
    for S in (0,1,2,4,8):
        x=run(cap,S)
        print(f"cap {cap}: skew {S} steps -> X fetches {x:.3f} per row; traffic/alg {(x+1+0.16)/(2.16):.3f}")
