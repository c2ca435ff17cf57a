# round 5: final_kernel with the first tile's records kept from its counting pass: kernel statistics (two runs), parity
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 1 2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$k -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/b$k.json 2> $OUT/st$k.log
grep "final_kernel\|l2_scatter_kernel\|l1_scatter_kernel\|query_side" $(ls $OUT/st$k/*/*kernel_stats.csv | head -1) | awk -F, '{print $1, $2, $4}' | cut -c1-120
rm -rf $OUT/st$k
done
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -2
timeout 600 python tests/fuzz_parity.py 60 95000 2>&1 | tail -1
timeout 600 python tests/fuzz_sharded.py 30 96000 2>&1 | tail -1
