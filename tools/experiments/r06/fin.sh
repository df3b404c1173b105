#!/bin/bash
# fused InstanceNorm finalisation: correctness (A/B hash test + generator suite) and the frames/s A/B
O=gpurun_out/r06fin; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_generator.py -x -q 2>&1 | tail -8
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary --no-strict"
for ff in 1 0 1 0; do
  LWG_FUSED_FINALIZE=$ff timeout 300 $B > $O/ff_$ff.json 2> $O/ff_$ff.err
  python -c "
import json; d=json.loads(open('$O/ff_$ff.json').read().strip().splitlines()[-1]); print('LWG_FUSED_FINALIZE=$ff: %.1f fps %.4f ms/step %s' % (d['value'], d['ms_per_step'], d['ms_per_step_windows']))" || tail -5 $O/ff_$ff.err
done
