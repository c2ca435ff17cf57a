# ingest on the box's CPU: pin span sweep (16 threads spread over the first <span> cores of socket 0)
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
run() { echo "== $*"; env "$@" RALA_IO_TRACE=1 tools/_bin/ingest_probe /tmp/c3.paf 1000000 16 2>&1 | grep "^\[io\] 16" | tail -3; }
run RALA_IO_PIN_SPAN=16
run RALA_IO_PIN_SPAN=24
run RALA_IO_PIN_SPAN=32
run RALA_IO_PIN_SPAN=64
