import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, os.path.join(ROOT,'tests')]
import _fixtures as fx
from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML, parameter_from
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
cfg=dict(CORE_PARAM_YAML)
C,N=1024,1000000
w=np.load(os.path.join(ROOT,'tests/golden/weights.npz')); W={k:w[k] for k in w.files}
m=ElevationMap(parameter_from(cfg,C,'reference_fp16',W))
R=np.eye(3,dtype=np.float32); t=np.array([0,0,1],np.float32)
clouds=[fx.cloud(C,N,s,dz=(0.0 if s==0 else -0.02*s)) for s in range(5)]
for i in range(3):
    m.update_map_with_kernel(clouds[i],[],R,t.copy(),1.0,1.0)
    for _ in range(4): m.update_time()
m.update_variance()
for i in range(8):
    m.update_map_with_kernel(clouds[i%5],[],R,t.copy(),1.0,1.0)
    e=m.elevation_map
    inert=(e[2]>=0.5)&(e[4]<0.5)
    blk=inert.reshape(128,8,16,64).all(axis=(1,3))
    print(i,'valid frac',(e[2]>=0.5).mean().round(4),'inert frac (pre-frame state)',inert.mean().round(4),'coarse 8x64 clean',blk.mean().round(4), 'time>=0.5', (e[4]>=0.5).mean().round(4))
