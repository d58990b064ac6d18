#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_lift_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout 300 python - <<'P' 2>&1 | grep -v "^/opt"
import sys, json, torch
sys.argv = ['bench.py']
sys.path.insert(0, '.')
import bench
torch.cuda.set_stream(torch.cuda.Stream())
torch.cuda.set_device(0)
print(json.dumps(bench._r(bench.k1_record(torch.device('cuda', 0), 2)), indent=1))
P
