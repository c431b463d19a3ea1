mkdir -p gpurun_out/terrain; rm -f gpurun_out/terrain/*
timeout 1500 python -m pytest tests/test_hip_terrain.py tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_randomized.py tests/test_hip_semantic.py tests/test_hip_strips.py tests/test_hip_comm.py tests/test_hip_large_maps.py tests/test_hip_shift.py tests/test_hip_fullsize.py -m gpu -q -x 2>&1 | grep -vE "^RCCL|^HIP ver|^ROCm|^Hostname|^Librccl" | tail -30 > gpurun_out/terrain/tests.txt
for f in base split5 split5; do
  EMAP_HIP_LIB=$PWD/tools/ab/$f.so timeout 600 python bench.py --no-large --no-cpu-baseline 2>/dev/null | head -1 >> gpurun_out/terrain/bench_$f.json
done
for f in base split5; do
  EMAP_HIP_LIB=$PWD/tools/ab/$f.so timeout 600 python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | head -1 > gpurun_out/terrain/bench5_$f.json
done
tail -5 gpurun_out/terrain/tests.txt
