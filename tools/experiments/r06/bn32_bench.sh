#!/bin/bash
for v in 1 0 1 0; do
LWG_F32_BN32=$v timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, json, torch, sys
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
p = bench.secondary_personalize(dev)
l = bench.secondary_latency(dev)
print("LWG_F32_BN32=%s" % os.environ.get("LWG_F32_BN32"), json.dumps({"personalize": p, "latency": l}))
PY
done
