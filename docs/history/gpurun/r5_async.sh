# round 5: RALA_HIP_MEM_HOST_ASYNC - the columns uploaded inside rala_hip_initialize: the tests, the bench member
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05u
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "columns" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
from rala_amd.synth import Dataset
for wl in ("c3", "c5"):
    ds = Dataset.config(wl)
    print(wl, json.dumps(bench.from_pinned_host(ds, 0), indent=1), flush=True)
PY
