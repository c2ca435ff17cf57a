# rocprofv3 kernel statistics of one C5 step (4 M reads / 303.6 M overlaps)
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/c5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/c5/stats -- python3 $ROOT/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/c5/bench.json 2> $ROOT/gpurun_out/c5/stats.log
rm -f $ROOT/gpurun_out/c5/stats/*/*kernel_trace.csv
cd $ROOT
cut -c1-300 gpurun_out/c5/bench.json
python3 -c "
import json; d=json.load(open('gpurun_out/c5/bench.json')); print(d['stage_ms'])"
head -12 gpurun_out/c5/stats/*/*kernel_stats.csv | cut -c1-50,150-260
