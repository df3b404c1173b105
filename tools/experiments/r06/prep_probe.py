import sys, os, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_train
m = bench_train.build(4, 256, precision="bf16x3")
for _ in range(3):
    m.optimize_parameters()
tr = m._generator_trainer()
pw = tr.prepared
ents = list(pw._entries.values())
print("entries", len(ents), "ready", sum(1 for e in ents if e[4]), "elements", sum(e[2].numel() for e in ents), "matrix floats", sum(e[3].numel() for e in ents))
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        pw.refresh()
    torch.cuda.synchronize()
    print("refresh ms", (time.perf_counter() - t0) / 10 * 1e3)
from collections import Counter
print(Counter((e[1], e[0].k, e[0].stride, e[0].transposed, e[0].Cin, e[0].Cout) for e in ents).most_common(40))
