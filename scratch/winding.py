import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from raycast_utils import *
exec(open('scratch/brute.py').read().split("d=np.load")[0])
def quat_rot(q,v):
    w,x,y,z=q; p=np.array([x,y,z]); a=np.cross(p,v); b=np.cross(p,a); return v+2*(a*w+b)
# look at each object from 6 directions by rotating the OBJECT; count ref hits vs silhouette (both-sided brute = any hit)
import itertools
rots=[[1,0,0,0],[0.7071068,0.7071068,0,0],[0.7071068,-0.7071068,0,0],[0.7071068,0,0,0.7071068],[0.7071068,0,0,-0.7071068],[0,0,0,1],[0.5,0.5,0.5,0.5],[0.8,0.2,-0.4,0.4]]
for obj in range(geo.num_objects):
    row=[]
    for q in rots:
        q=np.array(q,np.float32); q/=np.linalg.norm(q)
        i=np.zeros(1,INSTANCE_DT); i['position']=[0,6,0]; i['rotation']=q; i['scale']=[1,1,1]; i['matID']=-1; i['objectID']=obj
        v=np.zeros(1,VIEW_DT); v['rotation']=[1,0,0,0]; v['xScale']=1; v['yScale']=-1
        l=np.zeros(1,LIGHT_DT); l['type']=1; l['direction']=[0,1,0]; l['cutoff']=-1
        _,d=ref_render(geo,1,i,[0],[1],v,l,[0],[1],32)
        # two-sided silhouette
        vs=geo.vertices[geo.vertex_offsets[obj]:geo.vertex_offsets[obj+1]].astype(np.float64); tr=geo.indices[geo.triangle_offsets[obj]:geo.triangle_offsets[obj+1]]
        wv=np.array([quat_rot(q.astype(np.float64),p) for p in vs])+np.array([0,6,0.])
        T=wv[tr]; rays=primary_rays(v[0],32)
        hits=0
        for py in range(32):
            for px in range(32):
                dd=rays[py,px]; e1=T[:,1]-T[:,0]; e2=T[:,2]-T[:,0]; pv=np.cross(dd,e2); det=(e1*pv).sum(-1)
                with np.errstate(all='ignore'):
                    inv=1/det; tv=-T[:,0]; uu=(tv*pv).sum(-1)*inv; qv=np.cross(tv,e1); vv=(qv*dd).sum(-1)*inv; t=(e2*qv).sum(-1)*inv
                hits+= ((np.abs(det)>1e-12)&(uu>=0)&(vv>=0)&(uu+vv<=1)&(t>0)).any()
        row.append((int((d[0]>0).sum()),int(hits)))
    print("object",obj,row)
