import sys, numpy as np
sys.path.insert(0,'/root/repo')
from bonnie32_amd import scenegen
from oracle import np_model as M
cfg=sys.argv[1]; BS=int(sys.argv[2]) if len(sys.argv)>2 else 8
sc=scenegen.make_scene(cfg)
V=sc.vertices; F=sc.faces
pos=V['pos'].astype(np.float32)
sx,sy=M.project_fixed(pos, sc.camera, sc.width, sc.height)
idx=F['v']
x=sx[idx].astype(np.int64); y=sy[idx].astype(np.int64); z=(pos[:,2][idx]+np.float32(5.0)).astype(np.float32)
area=(y[:,1]-y[:,2])*(x[:,0]-x[:,2])+(x[:,2]-x[:,1])*(y[:,0]-y[:,2])
# reference backface: signed_area=(v2.x-v1.x)*(v3.y-v1.y)-(v3.x-v1.x)*(v2.y-v1.y) <=0 -> back
sa=(x[:,1]-x[:,0])*(y[:,2]-y[:,0])-(x[:,2]-x[:,0])*(y[:,1]-y[:,0])
vis=sa>0
key=((z[:,0]+z[:,1])+z[:,2])/np.float32(3.0)
prio_all=(-key).astype(np.float64)*1e7 + np.arange(len(key))   # larger = drawn later = nearer; tie -> face id
ids=np.nonzero(vis)[0]
rng=np.random.default_rng(0)
def tile_stats(tx,ty,order_mode,phases):
    X0,Y0=tx*64,ty*64
    sel=[i for i in ids if x[i].max()>=X0 and x[i].min()<X0+64 and y[i].max()>=Y0 and y[i].min()<Y0+64]
    sel=np.array(sel)
    if order_mode=='sorted': sel=sel[np.argsort(-prio_all[sel])]
    else: sel=rng.permutation(sel)
    top=np.full((64,64),-np.inf); sec=np.full((64,64),-np.inf)
    n=len(sel); tot_frag=0; done_frag=0; tot_rows=0; done_rows=0
    hiz=np.full((64//BS,64//BS),-np.inf)
    bounds=[int(n*p) for p in phases]+[n]
    k=0
    xs=np.arange(64)+X0
    for i,s in enumerate(sel):
        if k<len(bounds)-1 and i==bounds[k]:
            hiz=sec.reshape(64//BS,BS,64//BS,BS).min(axis=(1,3)); k+=1
        P=prio_all[s]
        x1,x2,x3=x[s]; y1,y2,y3=y[s]
        a=area[s]; sg=1 if a>0 else -1
        cy0=max(min(y1,y2,y3),Y0); cy1=min(max(y1,y2,y3)+1,Y0+64)
        for yy in range(cy0,cy1):
            e0=sg*((y2-y3)*(xs-x3)+(x3-x2)*(yy-y3)); e1=sg*((y3-y1)*(xs-x3)+(x1-x3)*(yy-y3)); e2=abs(a)-e0-e1
            m=(e0>=0)&(e1>=0)&(e2>=0)
            nz=np.nonzero(m)[0]
            if len(nz)==0: continue
            lo,hi=nz[0],nz[-1]+1
            tot_frag+=hi-lo; tot_rows+=1
            r=yy-Y0
            hb=hiz[r//BS, lo//BS:(hi-1)//BS+1].min()
            if P<hb: continue
            done_frag+=hi-lo; done_rows+=1
            seg=slice(lo,hi)
            t=top[r,seg]; s2=sec[r,seg]
            newtop=np.maximum(t,P); loser=np.minimum(t,P)
            sec[r,seg]=np.maximum(s2,loser); top[r,seg]=newtop
    return n,tot_frag,done_frag,tot_rows,done_rows
for mode in ('random','sorted'):
  for phases in ([0.25,0.5,0.75],[0.1,0.2,0.3,0.4,0.5,0.6,0.7,0.8,0.9]):
    acc=np.zeros(5)
    for (tx,ty) in [(5,5),(20,15),(33,22)]:
        acc+=np.array(tile_stats(tx,ty,mode,phases))
    print(cfg,'BS',BS,mode,len(phases),'phases: entries',acc[0],'frag kept %.3f'%(acc[2]/acc[1]),'rows kept %.3f'%(acc[4]/acc[3]))
