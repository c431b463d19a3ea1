"""PCIe-inclusive frame rate through the reference's entry point (input_pointcloud with a float64 host cloud, as the
ROS wrapper hands it over) vs the device-resident rate bench.py reports."""
import sys, os, time, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, os.path.join(ROOT,'tests')]
import _fixtures as fx
from elevation_mapping_cupy_amd.configs import CORE_PARAM_YAML, parameter_from
from elevation_mapping_cupy_amd.elevation_mapping import ElevationMap
cfg=dict(CORE_PARAM_YAML, enable_visibility_cleanup=False, enable_overlap_clearance=False)
C,N=1024,1000000
w=np.load(os.path.join(ROOT,'tests/golden/weights.npz')); W={k:w[k] for k in w.files}
m=ElevationMap(parameter_from(cfg,C,'reference_fp16',W))
R=np.eye(3,dtype=np.float32); t=np.array([0,0,1],np.float32)
c32=[fx.cloud(C,N,s,dz=-0.02*s) for s in range(5)]; c64=[c.astype(np.float64) for c in c32]
for name,clouds in (('float32 host cloud',c32),('float64 host cloud (ROS wrapper)',c64)):
    for i in range(5): m.input_pointcloud(clouds[i],['x','y','z'],R,t.copy(),1.0,1.0)
    m.sync(); t0=time.perf_counter(); K=30
    for i in range(K): m.input_pointcloud(clouds[i%5],['x','y','z'],R,t.copy(),1.0,1.0)
    m.sync(); dt=(time.perf_counter()-t0)/K
    print('%-34s %.3f ms/frame  %.1f Mpoints/s'%(name,dt*1e3,N/dt/1e6))
