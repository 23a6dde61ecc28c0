import sys, os, ctypes as C
os.environ["CS3_PROFILE"]="1"
sys.path.insert(0,'/root/repo')
import numpy as np
from csparse3_amd import csc_hip as hip, synth
m,n,Ap,Ai,Ax=synth.grid_jacobian()
F=hip.Factorization(m,n,Ap,Ai)
for _ in range(5): F.factor(Ax,1e-3)
ns=int(F.info.nsuper)
out=np.zeros((ns,8),dtype=np.int64)
L=hip.lib(); L.cs3_debug_front_stamps.argtypes=[C.c_void_p, C.POINTER(C.c_int64)]
rc=L.cs3_debug_front_stamps(F._h, out.ctypes.data_as(C.POINTER(C.c_int64))); assert rc==0
print("raw sample", out[:3])
valid=out[:,5]>0
d=np.diff(np.concatenate([np.zeros((ns,1),dtype=np.int64),out[:, :6]],axis=1),axis=1)[valid]
names=["desc","zero","gather","elim","stage","store"]
print("fronts with stamps", valid.sum(), "of", ns)
print("mean ticks per phase", dict(zip(names, d.mean(axis=0).round(0))))
print("p95 ticks per phase", dict(zip(names, np.percentile(d,95,axis=0).round(0))))
print("max ticks per phase", dict(zip(names, d.max(axis=0))))
tot=out[valid,5]
print("total: mean %.0f p95 %.0f max %.0f ticks"%(tot.mean(), np.percentile(tot,95), tot.max()))
# the slowest fronts
idx=np.argsort(-out[:,5])[:10]
print("slowest (schedule pos, stamps):")
for i in idx: print(i, out[i,:6])
sched=np.zeros(ns,dtype=np.int32); fr=np.zeros(ns,dtype=np.int32); fw=np.zeros(ns,dtype=np.int32)
L.cs3_debug_schedule.argtypes=[C.c_void_p]+[C.POINTER(C.c_int32)]*3
L.cs3_debug_schedule(F._h, *[a.ctypes.data_as(C.POINTER(C.c_int32)) for a in (sched,fr,fw)])
lvl=sn_lvl=F.supernodes()[2][sched]
el=out[:,3]-out[:,2]; ga=out[:,2]-out[:,1]
print("slowest fronts: pos level r w gather elim total")
for i in np.argsort(-out[:,5])[:12]: print(i, lvl[i], fr[i], fw[i], ga[i], el[i], out[i,5])
for name,lo,hi in [("r<=16",0,16),("r<=32",16,32),("r<=64",32,64),("r<=136",64,136)]:
    m=(fr>lo)&(fr<=hi)&valid&(fw>=4)
    if m.sum(): print(name,"n",m.sum(),"elim cycles/pivot: median %.0f p90 %.0f"%(np.median(el[m]/fw[m]), np.percentile(el[m]/fw[m],90)), " gather median %.0f max %.0f"%(np.median(ga[m]), ga[m].max()), "w median %.0f max %d"%(np.median(fw[m]), fw[m].max()))
# per level: max total
for l in range(int(lvl.max())+1):
    m=(lvl==l)&valid
    if m.sum():
        j=np.flatnonzero(m)[np.argmax(out[m,5])]
        print("level",l,"fronts",m.sum(),"max total",out[j,5],"(r,w)=",fr[j],fw[j],"max w",fw[m].max())
