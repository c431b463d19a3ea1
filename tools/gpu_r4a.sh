# round-4 GPU call A: full gpu suite (with the new large strip tests), pitch experiment, strip emulation cfg5, default bench line
O=gpurun_out/r4a; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -q -x) > $O/pytest.log 2>&1
tail -5 $O/pytest.log | grep -E "passed|failed|error"
grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | head -30
for C in 4096 4160 6144 6208 8192 8256; do for V in post dma; do
  if [ $V = dma ]; then export EMAP_POST_DMA_WINDOW="1 100000"; else export EMAP_POST_DMA=0; fi
  timeout 120 python tools/exp_post_pitch.py --cell-n $C --tag $V >> $O/pitch.jsonl 2>> $O/pitch.err
  unset EMAP_POST_DMA_WINDOW EMAP_POST_DMA
done; done
cat $O/pitch.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['cell_n'], d['tag'], d['post_ms'], d['post_ns_per_cell'], d['stage_ns_per_cell'])"
timeout 600 python tools/strip_emulation.py --workload cfg5 --steps 10 --gs 1 8 > $O/strips_cfg5.json 2> $O/strips_cfg5.err
tail -3 $O/strips_cfg5.err; head -c 3000 $O/strips_cfg5.json; echo
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err; head -c 1500 $O/bench_default.json; echo
