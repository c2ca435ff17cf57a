cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time, ctypes, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from rala_amd.synth import Dataset
import test_gpu_ingest as T
ds = Dataset.config("c3")
d = tempfile.mkdtemp(dir="/tmp")
paf = os.path.join(d, "ovl.paf")
ds.write_paf(paf)
names = ["r%d" % i for i in range(ds.n_reads)]
L = T._lib()
rl = np.ascontiguousarray(ds.read_len, dtype=np.uint32)
nm = "\n".join(names).encode()
for threads in (4, 8, 12, 16, 16, 8):
    os.environ["RALA_HIP_TRACE"] = "1"
    t0 = time.time()
    h = L.hp_paf_device(paf.encode(), nm, rl.ctypes.data, len(names), 1, threads)
    info = np.zeros(6, dtype=np.int64); L.hp_paf_device_info(h, info.ctypes.data)
    print("threads", threads, "rc", info[0], "irregular", info[1], "records", info[3], "ship ms", info[4] / 1000, "tokenise ms", info[5] / 1000, flush=True)
    L.hp_paf_device_free(h)
os.remove(paf)
PY
