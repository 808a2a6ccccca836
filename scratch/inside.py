import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from raycast_utils import *
exec(open('scratch/brute.py').read().split("d=np.load")[0])
def one(obj, pos, scale, campos):
    i=np.zeros(1,INSTANCE_DT); i['position']=pos; i['rotation']=[1,0,0,0]; i['scale']=scale; i['matID']=-1; i['objectID']=obj
    v=np.zeros(1,VIEW_DT); v['rotation']=[1,0,0,0]; v['xScale']=1; v['yScale']=-1; v['position']=campos
    l=np.zeros(1,LIGHT_DT); l['type']=1; l['direction']=[0,1,0]; l['cutoff']=-1
    _,d=ref_render(geo,1,i,[0],[1],v,l,[0],[1],16)
    return (d[0]>0).sum()
print("cube outside", one(0,[0,5,0],[1,1,1],[0,0,0]))
print("cube inside", one(0,[0,0,0],[3,3,3],[0,0,0]))
print("cube inside off", one(0,[0.2,0.1,0.3],[3,3,3],[0,0,0]))
print("ellipsoid outside", one(1,[0,5,-1],[1,1,1],[0,0,0]))
print("ellipsoid inside", one(1,[0,0,-1],[2,8,2],[0,0,0]))
print("ellipsoid inside2", one(1,[0.1,0.05,-0.9],[2,8,2],[0,0,0]))
print("cyl inside", one(2,[0,0,0],[2,2,20],[0,0,0.3]))
