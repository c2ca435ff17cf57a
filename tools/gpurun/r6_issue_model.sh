# round 6: the pile kernel's issue model at C3 on the box's build(s): vector / scalar / LDS / memory instructions per read,
# wave cycles, waits, the vector pipe's busy time, the clock - for every variant build given ("" = the default build)
# usage: r6_issue_model.sh <out.json> "<flags A>" ["<flags B>" ...]
ROOT=$GRAFT_REPO_ROOT
OUTJ=$1; shift
OUT=$ROOT/gpurun_out/r06m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for def in "$@"; do
  i=$((i+1))
  touch $ROOT/rala_amd/csrc/pile_runs_kernel.hip
  (cd $ROOT && RALA_HIPCC_FLAGS="$def" python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2)
  echo "$def" > $OUT/v$i.flags
  p=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
             "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM"; do
    p=$((p+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/v${i}_p$p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $OUT/v${i}_p$p.log || tail -3 $OUT/v${i}_p$p.log
  done
done
cd $ROOT
python3 - "$OUTJ" <<'PY'
import csv, glob, collections, json, os, re, sys
PROD = re.compile(r"pile_runs_kernel<512u?, false, 0, true, 16384u?, 2u?, true>")
out = {"workload": "C3 (1 M reads / 50.86 M overlaps)", "kernel": "pile_runs_kernel<512, false, 0, true, 16384, 2, true>",
       "note": "rocprofv3 --pmc, three passes per build, bench.py --steps 2 --warmup 1: means over the kernel's dispatches; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts; GRBM_GUI_ACTIVE is summed over the 8 XCDs", "builds": []}
for fl in sorted(glob.glob("gpurun_out/r06m/v*.flags")):
    tag = os.path.basename(fl)[:-6]
    c = collections.defaultdict(list); dur = []
    for f in glob.glob("gpurun_out/r06m/%s_p*/*/*counter_collection.csv" % tag):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if PROD.search(k):
                c[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob("gpurun_out/r06m/%s_p1/*/*kernel_trace.csv" % tag):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if PROD.search(k):
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    m = {k: sum(v) / len(v) for k, v in c.items()}
    reads = 1_000_000
    b = {"flags": open(fl).read().strip(), "dispatches": {k: len(v) for k, v in c.items()}, "kernel_ms_under_counters": sum(dur) / len(dur) if dur else None,
         "per_read": {k: round(m[k] / reads, 1) for k in m if k.startswith("SQ_INSTS") or k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_SALU", "SQ_IFETCH", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT")},
         "raw": m}
    if "SQ_INSTS_VALU" in m and "SQ_BUSY_CYCLES" in m and dur:
        ms = b["kernel_ms_under_counters"]
        clock_ghz = m.get("GRBM_GUI_ACTIVE", 0) / 8 / (ms * 1e6) if ms else None
        issue_quad_cycles_per_simd = m["SQ_INSTS_VALU"] / 1024.0            # one quad-cycle (4 cycles) of a SIMD's issue per wave64 vector instruction
        kernel_cycles = ms * 1e6 * clock_ghz if clock_ghz else None
        b["model"] = {"clock_ghz_from_GRBM_GUI_ACTIVE": clock_ghz, "vector_issue_cycles_per_simd": issue_quad_cycles_per_simd * 4,
                      "kernel_cycles": kernel_cycles, "vector_issue_over_kernel": issue_quad_cycles_per_simd * 4 / kernel_cycles if kernel_cycles else None,
                      "vector_issue_ms": issue_quad_cycles_per_simd * 4 / (clock_ghz * 1e6) if clock_ghz else None}
    out["builds"].append(b)
    print(tag, b["flags"], "ms", b["kernel_ms_under_counters"], {k: b["per_read"].get(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU")}, b.get("model"))
json.dump(out, open(sys.argv[1], "w"), indent=1)
PY
rm -rf $OUT/v*_p*/
touch $ROOT/rala_amd/csrc/pile_runs_kernel.hip
(cd $ROOT && python -c "from rala_amd import build; build.build_hip()" 2>&1 | grep -i error | head -2)
