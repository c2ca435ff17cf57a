#!/bin/bash
# CPU tests of the host code (assembly graph clean-up stages, PAF / MHAP / FASTA readers) against
# an AddressSanitizer + UBSan build of libassembly_graph.so; the normal build is restored afterwards.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
H=$ROOT/rala_amd/host
cp $H/libassembly_graph.so /tmp/libassembly_graph.so.normal
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I$H \
    -o $H/libassembly_graph.so $H/assembly_graph.cpp $H/assembly_graph_capi.cpp $H/io.cpp $H/io_capi.cpp -lz
cd $ROOT
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_ingest_cpu.py tests/test_layout_cpu.py -x -q || rc=$?
cp /tmp/libassembly_graph.so.normal $H/libassembly_graph.so
exit ${rc:-0}
