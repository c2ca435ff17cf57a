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
lscpu | grep -i "numa\|socket" | head -8
run() { echo "== $*"; env "$@" RALA_IO_TRACE=1 tools/_bin/ingest_probe /tmp/c3.paf 1000000 16 2>&1 | grep "^\[io\]\|empty table" | tail -2; }
run RALA_IO_PIN_SPAN=16
run RALA_IO_PIN_SPAN=24
run RALA_IO_PIN_SPAN=32
run RALA_IO_PIN_SPAN=48
