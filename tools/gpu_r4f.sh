O=gpurun_out/r4f; mkdir -p $O
for f in tools/ab/*.so; do for C in 8192 2048; do
  [ $C = 8192 ] && export EMAP_POST_DMA=0
  EMAP_HIP_LIB=$PWD/$f timeout 200 python tools/exp_post_pitch.py --cell-n $C --tag $(basename $f .so) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['tag'], d['cell_n'], 'post_ms', d['post_ms'], 'ns/cell', d['post_ns_per_cell'])"
done; done
