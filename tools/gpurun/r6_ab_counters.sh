# round 6 (also the issue model of profiles/r06_c3_pile_issue_model.json): the counters of two (or more) variants of the first pile kernel measured INSIDE one process and one profiler pass each
# (tools/pile_ab.py under rocprofv3 --pmc): what differs between processes cancels
# usage: r6_ab_counters.sh "<variants>" <out.json>     (R6_FLAGS: extra compile flags, e.g. the list of cases)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r06c
mkdir -p $OUT
cd $ROOT
touch rala_amd/csrc/pile_runs_kernel.hip
RALA_HIPCC_FLAGS="-DRALA_PILE_AB $R6_FLAGS" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
cd /tmp && export TMPDIR=/tmp
p=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  p=$((p+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$p -- python3 $ROOT/tools/pile_ab.py c3 "$1" ${R6_ROUNDS:-1} ${R6_STEPS:-3} > $OUT/p$p.log 2>&1 || tail -3 $OUT/p$p.log
done
cd $ROOT
python3 - "$2" <<'PY'
import csv, glob, collections, json, re, sys
PROD = re.compile(r"pile_runs_kernel<512u?, false, 0, true, 16384u?, 2u?, true, (\d+)u?>")
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r06c/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        m = PROD.search(row["Kernel_Name"])
        if m:
            per[int(m.group(1))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob("gpurun_out/r06c/p*/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        m = PROD.search(row["Kernel_Name"])
        if m:
            dur[int(m.group(1))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
out = {}
for v in sorted(per):
    m = {k: sum(x) / len(x) for k, x in per[v].items()}
    ms = sum(dur[v]) / len(dur[v])
    out[v] = {"kernel_ms_under_counters": ms, "dispatches": len(dur[v]), "per_read": {k: round(x / 1e6, 1) for k, x in m.items()}}
    if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
        clock = m["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e6)             # GHz (the counter is summed over the 8 XCDs)
        cycles = ms * 1e6 * clock
        issue = m["SQ_INSTS_VALU"] / 1024.0 * 4.0                # a wave64 vector instruction holds its SIMD's issue for 4 cycles; 1024 SIMDs
        out[v]["model"] = {"clock_ghz": clock, "kernel_cycles": cycles, "vector_issue_cycles_per_simd": issue, "vector_issue_over_kernel": issue / cycles,
                           "vector_issue_ms": issue / (clock * 1e6),
                           "resident_wavefronts_per_simd": m.get("SQ_WAVE_CYCLES", 0) * 4.0 / 1024.0 / cycles if "SQ_WAVE_CYCLES" in m else None}
    print("variant", v, "ms %.3f" % out[v]["kernel_ms_under_counters"], json.dumps(out[v]["per_read"]))
json.dump({"workload": "C3 (1 M reads / 50.86 M overlaps / 10.0 Gbase)", "kernel": "pile_runs_kernel<512, false, 0, true, 16384, 2, true, VARIANT>",
           "variants": "65536 = the product instantiation (non-temporal row stores: what a launch with variant 0 runs where the rows lie in mapped chunks; 0 = the same with plain stores); bit 0 the loop over the items as in round 5 (56 scalar spills), bit 1 the events by ordinary loads, bit 13 the reads as launched instead of XCD ranges: 8195 = round 5's kernel",
           "note": "rocprofv3 --kernel-trace --pmc over tools/pile_ab.py: every variant inside ONE process and one profiler pass per counter group (what differs between processes - the clock state of the box - cancels); per_read = counter / 10^6 reads; SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts",
           "builds": out}, open(sys.argv[1], "w"), indent=1)
PY
rm -rf $OUT/p*/
rm -f $OUT/p*.log
touch rala_amd/csrc/pile_runs_kernel.hip
python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2
