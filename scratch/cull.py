import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from raycast_utils import *
exec(open('scratch/brute.py').read().split("d=np.load")[0])
def quat_rot(q,v):
    w,x,y,z=q; p=np.array([x,y,z]); a=np.cross(p,v); b=np.cross(p,a); return v+2*(a*w+b)
obj=2; q=np.array([0.7071068,0.7071068,0,0],np.float32); q/=np.linalg.norm(q)
i=np.zeros(1,INSTANCE_DT); i['position']=[0,6,0]; i['rotation']=q; i['scale']=[1,1,1]; i['matID']=-1; i['objectID']=obj
v=np.zeros(1,VIEW_DT); v['rotation']=[1,0,0,0]; v['xScale']=1; v['yScale']=-1
l=np.zeros(1,LIGHT_DT); l['type']=1; l['direction']=[0,1,0]; l['cutoff']=-1
_,d=ref_render(geo,1,i,[0],[1],v,l,[0],[1],32)
vs=geo.vertices[geo.vertex_offsets[obj]:geo.vertex_offsets[obj+1]].astype(np.float64); tr=geo.indices[geo.triangle_offsets[obj]:geo.triangle_offsets[obj+1]]
wv=np.array([quat_rot(q.astype(np.float64),p) for p in vs])+np.array([0,6,0.])
T=wv[tr]; rays=primary_rays(v[0],32)
for py in range(12,20):
    line=""
    for px in range(12,20):
        dd=rays[py,px]; e1=T[:,1]-T[:,0]; e2=T[:,2]-T[:,0]; n=np.cross(e1,e2); pv=np.cross(dd,e2); det=(e1*pv).sum(-1)
        with np.errstate(all='ignore'):
            inv=1/det; tv=-T[:,0]; uu=(tv*pv).sum(-1)*inv; qv=np.cross(tv,e1); vv=(qv*dd).sum(-1)*inv; t=(e2*qv).sum(-1)*inv
        ok=(np.abs(det)>1e-12)&(uu>=0)&(vv>=0)&(uu+vv<=1)&(t>0)
        front=ok&((n*dd).sum(-1)<0)
        line+=f"{int(ok.any())}{int(front.any())}{int(d[0][py,px]>0)} "
    print(line)
