#!/bin/bash
# Knock-out timing (measurement build, WRONG RESULTS by design): what each launch group costs inside the two-lane pipeline.
#   gpurun --timeout 1200 -- 'bash tools/experiments/r06/ko.sh'
O=gpurun_out/r06ko; mkdir -p $O
B="python bench.py --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline --no-roofline --no-fp32-mode --no-secondary --no-strict"
for ko in 0 1 2 4 8 16 32 64 128 31 0; do
  LWG_LIB=exp LWG_KO=$ko timeout 300 $B > $O/ko_$ko.json 2> $O/ko_$ko.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/ko_$ko.json").read().strip().splitlines()[-1])
    print("KO $ko: %.1f fps  %.4f ms/step  windows %s" % (d["value"], d["ms_per_step"], d["ms_per_step_windows"]))
except Exception as e:
    print("KO $ko failed", e, open("$O/ko_$ko.err").read()[-800:])
PY
done
LWG_LIB=exp LWG_KO=0 timeout 300 $B --lanes 1 > $O/ko_0_l1.json 2> $O/ko_0_l1.err; python -c "
import json; d=json.loads(open('$O/ko_0_l1.json').read().strip().splitlines()[-1]); print('one lane KO 0: %.1f fps %.4f ms' % (d['value'], d['ms_per_step']))"
LWG_LIB=exp LWG_KO=1 timeout 300 $B --lanes 1 > $O/ko_1_l1.json 2> $O/ko_1_l1.err; python -c "
import json; d=json.loads(open('$O/ko_1_l1.json').read().strip().splitlines()[-1]); print('one lane KO 1: %.1f fps %.4f ms' % (d['value'], d['ms_per_step']))"
