# ingest A/B on the box's CPU: thread pinning, huge pages, freeing the text during / after the parse
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
cat /sys/kernel/mm/transparent_hugepage/defrag; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
run() { echo "== $*"; env "$@" RALA_IO_TRACE=1 tools/_bin/ingest_probe /tmp/c3.paf 1000000 16 2>&1 | grep "^\[io\] 16\|real table" | tail -4; }
run A=1
run RALA_IO_NO_PIN=1
run RALA_IO_KEEP_TEXT=1
run RALA_IO_NO_HUGEPAGES=1
run RALA_IO_NO_HUGEPAGES=1 RALA_IO_KEEP_TEXT=1
