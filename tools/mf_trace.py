"""Summary of the rocprofv3 kernel trace tools/make_multifrontal_profile.sh leaves under gpurun_out/prof_mf: one factorisation
(wall, kernels by class, slowest groups) and the substitution sweeps of the multifrontal sparse direct solver.  Optional
arguments: first kernel index and count of a listing of the factorisation's launches."""
import csv,os,re,sys
from collections import defaultdict
d='gpurun_out/prof_mf/stats/runc/'
f=sorted([x for x in os.listdir(d) if x.endswith('kernel_trace.csv')], key=lambda x: os.path.getmtime(d+x))[-1]
rows=list(csv.DictReader(open(d+f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('void k_mf_init')]
bp=[i for i,n in enumerate(names) if n.startswith('k_build_perm')]
nfac=3
ng=len(idx)//nfac
print("groups per factorization",ng, len(idx), len(bp))
k=1
i0=idx[k*ng]; i1=bp[(k+1)*ng-1]
seg=rows[i0:i1+1]
wall=(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e6
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e6
print("factorization",k,"kernels",len(seg),"wall %.1f ms busy %.1f ms"%(wall,busy))
agg=defaultdict(lambda:[0,0.0])
for r in seg:
    n=re.sub(r'\(.*','',r['Kernel_Name'])
    agg[n][0]+=1; agg[n][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for n,(c,t) in sorted(agg.items(), key=lambda x:-x[1][1])[:14]: print("   %-60s %5d %8.2f ms"%(n[:60],c,t))
gi=[j for j,r in enumerate(seg) if r['Kernel_Name'].startswith('void k_mf_init')]+[len(seg)]
gt=[]
for a,b in zip(gi,gi[1:]):
    t=(int(seg[b-1]['End_Timestamp'])-int(seg[a]['Start_Timestamp']))/1e6
    gt.append((round(t,2),a,b-a,int(seg[a]['Grid_Size_Y'])))
print("   slowest groups (ms, first kernel, kernels, nmat):", sorted(gt,reverse=True)[:14])
print("   sum of groups %.1f"%sum(t for t,_,_,_ in gt))
if len(sys.argv)>1:
    a=int(sys.argv[1]); b=a+int(sys.argv[2])
    for r in seg[a:b]:
        print("     %-50s grid %6s x %6s  %8.1f us"%(re.sub(r'\(.*','',r['Kernel_Name'])[:50], r['Grid_Size_X'], r['Grid_Size_Y'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
# solve sweeps: from first k_mf_fwd_load to last k_mf_scatter in a sweep
fl=[i for i,n in enumerate(names) if n.startswith('void k_mf_fwd_load')]
sc=[i for i,n in enumerate(names) if n.startswith('void k_mf_scatter')]
nsw=len(fl)//ng
print("sweeps",nsw)
for k in range(nsw):
    seg=rows[fl[k*ng]:sc[(k+1)*ng-1]+1]
    wall=(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e6
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in seg)/1e6
    print("  sweep",k,"kernels",len(seg),"wall %.1f busy %.1f"%(wall,busy))
    if k==1:
        agg=defaultdict(lambda:[0,0.0])
        for r in seg:
            n=re.sub(r'\(.*','',r['Kernel_Name'])
            agg[n][0]+=1; agg[n][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
        for n,(c,t) in sorted(agg.items(), key=lambda x:-x[1][1])[:10]: print("   %-60s %5d %8.2f ms"%(n[:60],c,t))
