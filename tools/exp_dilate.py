import sys, os, ctypes as ct, numpy as np, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, os.path.join(ROOT,'tests')]
import _fixtures as fx
from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML, parameter_from
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
cfg=dict(CORE_PARAM_YAML, enable_visibility_cleanup=False, enable_overlap_clearance=False)
C,N=1024,1000000
w=np.load(os.path.join(ROOT,'tests/golden/weights.npz')); W={k:w[k] for k in w.files}
m=ElevationMap(parameter_from(cfg,C,'reference_fp16',W))
R=np.eye(3,dtype=np.float32); t=np.array([0,0,1],np.float32)
for i in range(8):
    m.update_map_with_kernel(fx.cloud(C,N,i%5,dz=-0.02*(i%5)),[],R,t.copy(),1.0,1.0)
e=m.elevation_map
mask=(e[2]+e[6])<0.5
tiles=mask.reshape(64,16,16,64).any(axis=(1,3))
print('holes',mask.sum(),'tiles with hole',tiles.sum(),'of',tiles.size)
def timeit(name, n=50):
    m.sync(); t0=time.perf_counter()
    for _ in range(n): m.stage(name)
    m.sync(); return (time.perf_counter()-t0)/n*1e6
print('dilate us (bench state)', timeit('dilate'))
e2=e.copy(); e2[2]=1; m.elevation_map=e2
print('dilate us (all valid)', timeit('dilate'))
e2[2]=0; e2[6]=0; m.elevation_map=e2
print('dilate us (all invalid)', timeit('dilate'))
print('trav_normals us', timeit('traversability_normals'))
print('update_time us', timeit('update_time'))
