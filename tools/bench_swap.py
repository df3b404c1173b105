"""Secondary measurement (BASELINE.json config 4, appearance transfer): bench.py's `secondary.swap` block on its own, so that
it can be profiled (rocprofv3 --kernel-trace --stats -- python tools/bench_swap.py)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    print(json.dumps(bench.secondary_swap(torch.device("cuda", 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 30)))
