cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/_bin
export TMPDIR=/tmp
g++ -O2 -std=c++17 -Irala_amd/host -o tools/_bin/ingest_probe tools/ingest_probe.cpp rala_amd/host/io.cpp -lz -pthread || exit 1
python - <<'PY'
import sys
sys.path.insert(0, '.')
from rala_amd.synth import Dataset
ds = Dataset.config('c3')
ds.write_paf('/tmp/c3.paf')
PY
RALA_IO_TRACE=1 tools/_bin/ingest_probe /tmp/c3.paf 1000000 16 2>&1 | tail -14
lscpu | grep -i "model name\|^CPU(s)\|thread\|L3\|L2" | head -8
cat /sys/kernel/mm/transparent_hugepage/enabled
hipcc --offload-arch=gfx950 -O3 -o tools/_bin/valu_bench tools/valu_bench.hip 2>/dev/null && tools/_bin/valu_bench
