import sys, numpy as np
sys.path.insert(0,'.')
import torch
from nanort_amd import BVHAccel, TriangleMesh, scenes
v,f=scenes.plane(1000,500); a=BVHAccel(np.float32); a.Build(f.shape[0],TriangleMesh(v,f))
r1=scenes.camera_rays(1920,1080); h1,m1=a.TraverseBatch(r1)
for kind in ("shadow","bounce"):
    r=scenes.secondary_rays(kind,v,f,r1,h1,m1)
    d=torch.from_numpy(r.view(np.uint8)).cuda(); o=torch.empty(len(r)*16,dtype=torch.uint8,device='cuda'); m=torch.empty(len(r),dtype=torch.uint8,device='cuda')
    t=[];u=[]
    for _ in range(6):
        a.TraverseBatchDevice(d,o,m); t.append(a.LastTraverseMs())
        a.OccludedBatchDevice(d,m); u.append(a.LastTraverseMs())
    print(kind,'closest %.3f ms  occluded-query %.3f ms  hit fraction %.3f'%(np.median(t),np.median(u),float(m.float().mean())))
