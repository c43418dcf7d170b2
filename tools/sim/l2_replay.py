"""Replay of the row-per-wave SpMM's X-row accesses (one node pass, renumbered cfg 3 matrix) through an LRU model of one
XCD's L2: how many 1-KB panel rows miss, with the 32 workgroups of a group in perfect lock step.  Companion of
l2_replay_skew.py (workgroups lagging behind each other).  Input: /tmp/sim/rp.bin, col.bin written by dump_renumbered.cpp
(g++ -O2 -std=c++17 tools/sim/dump_renumbered.cpp, fed with the CSR arrays of workloads.laplacian_3d_pencil as int64/f64
binaries /tmp/ia.bin, /tmp/ja.bin, /tmp/va.bin).  Result on cfg 3: 1.10-1.17x algorithmic traffic (measured: 1.44x)."""
import numpy as np, sys
from collections import OrderedDict
rp=np.fromfile('/tmp/sim/rp.bin',dtype=np.int32); col=np.fromfile('/tmp/sim/col.bin',dtype=np.int32)
N=len(rp)-1
def run(cap_units, y_alloc, band=512, slices=8, mat_units=0.16):
    tot_x=0; tot_halo=0
    sl=(N+slices-1)//slices
    for g in range(slices):
        lo=g*sl; hi=min(N,lo+sl)
        cache=OrderedDict(); used=0.0
        miss=0; halo=0
        def touch(key, size):
            nonlocal used, miss
            if key in cache:
                cache.move_to_end(key); return True
            cache[key]=size; used+=size
            while used>cap_units:
                k,s=cache.popitem(last=False); used-=s
            return False
        for b0 in range(lo,hi,band):
            for i in range(b0,min(hi,b0+band)):
                touch(('m',i),mat_units)
                if not touch(('x',i),1.0): miss+=1
                for k in range(rp[i],rp[i+1]):
                    j=col[k]
                    if j==i: continue
                    if not touch(('x',j),1.0):
                        miss+=1
                        if j<lo or j>=hi: halo+=1
                if y_alloc: touch(('y',i),1.0)
        tot_x+=miss; tot_halo+=halo
    return tot_x, tot_halo
for cap in (4096, 3072, 2048):
    for ya in (0,1):
        x,h=run(cap,ya)
        print(f"cap {cap} KB-units, Y allocates {ya}: X row fetches {x} = {x/N:.3f} x N (halo part {h/N:.3f}); traffic/alg = {(x+N+0.16*N)/(2*N+0.16*N):.3f}")
