import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, ctypes as C
from raycast_utils import *
from test_raycast_gpu import _offsets
lib = C.CDLL('madrona_amd/_build/librender_prep_hip.so') if False else None
# geometry via the ref lib of render_prep (same meshes)
rl = C.CDLL('oracle/_ref/librender_prep_ref.so')
f = rl.sim_render_geometry; f.restype=C.c_int32; f.argtypes=[C.c_void_p]*7
counts=np.zeros(3,np.uint32); n=f(None,None,None,None,None,None,counts.ctypes.data)
verts=np.zeros((counts[0],3),np.float32); idx=np.zeros((counts[1],3),np.uint32)
voff=np.zeros(n+1,np.uint32); toff=np.zeros(n+1,np.uint32); mats=np.zeros((counts[2],3),np.float32); omat=np.zeros(n,np.int32)
f(verts.ctypes.data,idx.ctypes.data,voff.ctypes.data,toff.ctypes.data,mats.ctypes.data,omat.ctypes.data,None)
geo=Geometry(verts,idx,voff,toff,omat,mats)
d=np.load('gpurun_out/dbg_raycast.npz')
inst,ic,views,lights,lc,hd=d['inst'],d['ic'],d['views'],d['lights'],d['lc'],d['hd']
worlds=len(ic); res=hd.shape[1]
rr,rd=ref_render(geo,worlds,inst,_offsets(ic),ic,views,lights,_offsets(lc),lc,res,threads=16)
print("flipped vs hip", ((rd>0)!=(hd>0)).sum())

def quat_rot(q,v):
    w,x,y,z=q; p=np.array([x,y,z]); a=np.cross(p,v); b=np.cross(p,a); return v+2*(a*w+b)
def brute(view_idx):
    V=views.view(VIEW_DT).ravel()[view_idx]; w=V['worldIDX']
    off=_offsets(ic)[w]; I=inst.view(INSTANCE_DT).ravel()[off:off+ic[w]]
    q=V['rotation'].astype(np.float64); qi=q*np.array([1,-1,-1,-1])
    fwd=quat_rot(qi,np.array([0,1,0.])); u=quat_rot(qi,np.array([1,0,0.])); v=np.cross(fwd,u); v/=np.linalg.norm(v)
    h=1/(-float(V['yScale'])); vp=2*h
    o=V['position'].astype(np.float64)
    out=np.zeros((res,res))
    tris_world=[]
    for it in I:
        ob=it['objectID']; vs=geo.vertices[voff[ob]:voff[ob+1]].astype(np.float64); tr=geo.indices[toff[ob]:toff[ob+1]]
        q2=it['rotation'].astype(np.float64)
        wv=np.array([quat_rot(q2, p*it['scale']) for p in vs])+it['position']
        tris_world.append(wv[tr])
    T=np.concatenate(tris_world)
    for py in range(res):
        for px in range(res):
            dd=(-u*vp/2 - v*vp/2 + fwd) + (px+.5)/res*u*vp + (py+.5)/res*v*vp; dd/=np.linalg.norm(dd)
            e1=T[:,1]-T[:,0]; e2=T[:,2]-T[:,0]; pv=np.cross(dd,e2); det=(e1*pv).sum(-1)
            with np.errstate(all='ignore'):
                inv=1/det; tv=o-T[:,0]; uu=(tv*pv).sum(-1)*inv; qv=np.cross(tv,e1); vv=(qv*dd).sum(-1)*inv; t=(e2*qv).sum(-1)*inv
            ok=(np.abs(det)>1e-12)&(uu>=0)&(vv>=0)&(uu+vv<=1)&(t>0)
            out[py,px]=t[ok].min() if ok.any() else 0
    return out
for vi in (22, 26):
    b=brute(vi)
    print("view",vi,"brute hits",(b>0).sum(),"hip hits",(hd[vi]>0).sum(),"ref hits",(rd[vi]>0).sum(),
          "hip vs brute flipped",((b>0)!=(hd[vi]>0)).sum(),"ref vs brute flipped",((b>0)!=(rd[vi]>0)).sum())
    m=(b>0)&(hd[vi]>0); print("  hip max rel",np.abs(hd[vi][m]-b[m]).max() if m.any() else None)
    m=(b>0)&(rd[vi]>0); print("  ref max rel",np.abs(rd[vi][m]-b[m]).max() if m.any() else None)
print("---- single instances of world 11, view 22")
V=views.view(VIEW_DT).ravel()[22:23].copy(); V['worldIDX']=0
off=_offsets(ic)[11]; I=inst.view(INSTANCE_DT).ravel()[off:off+ic[11]]
L=lights.view(LIGHT_DT).ravel()[:1]
for k in range(len(I)):
    one=I[k:k+1].copy(); one['worldIDX']=0
    _,dep=ref_render(geo,1,one,[0],[1],V,L,[0],[1],res)
    # brute with one instance
    views_bak, inst_bak, ic_bak = views, inst, ic
    views, inst, ic = V, one, np.array([1],np.int32)
    b=brute(0)
    views, inst, ic = views_bak, inst_bak, ic_bak
    print(k,"obj",one['objectID'][0],"brute",(b>0).sum(),"ref",(dep[0]>0).sum(),"flipped",((b>0)!=(dep[0]>0)).sum())
