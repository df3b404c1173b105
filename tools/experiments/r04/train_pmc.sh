#!/bin/bash
# GPU box: PMC passes over one training iteration (bf16x3 convs): fabric reads and matrix-pipe busy per wgrad launch.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
OUT=$R/gpurun_out/train_pmc; mkdir -p $OUT
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- \
    python $R/tools/bench_train.py --precision bf16x3 --steps 1 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o p -- \
    python $R/tools/bench_train.py --precision bf16x3 --steps 1 > $OUT/mfma.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/wait -o p -- \
    python $R/tools/bench_train.py --precision bf16x3 --steps 1 > $OUT/wait.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob, re
def short(n):
    m = re.search(r"(wgrad_\w+<\d+>|conv_igemm_\w+<[^>]*>|\w+_kernel)", n); return m.group(1) if m else n[:40]
for tag in ("fetch", "mfma", "wait"):
    f = glob.glob("gpurun_out/train_pmc/%s/**/p_counter_collection.csv" % tag, recursive=True)
    if not f: print(tag, "no output"); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "wgrad" not in r["Kernel_Name"]: continue
        key = (short(r["Kernel_Name"]), r["Grid_Size"], r["Counter_Name"])
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items(): print(tag, k, "launches", n, "avg %.4g" % (v / n))
PY
