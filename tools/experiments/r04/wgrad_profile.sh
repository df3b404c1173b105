# rocprofv3 kernel trace of tools/wgrad_bench.py: per-kernel durations of the weight-gradient launches (bash tools/wgrad_profile.sh <tag>)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python $R/tools/wgrad_bench.py > $O/stats.log 2>&1
cd $R; python - $O <<'PY'
import csv, collections, sys, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/stats/**/k_kernel_trace.csv", recursive=True)[0])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "wgrad" in n or "reduce" in n:
        k = (n.split("::")[-1][:52], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    print("%-54s grid %5d %s %s  x%3d  avg %7.1f us  min %7.1f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
