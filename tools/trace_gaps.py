"""Diagnostic: idle gaps on the GPU inside one bench step, from a rocprofv3 --kernel-trace CSV.
python tools/trace_gaps.py <kernel_trace.csv>: takes the last occurrence of the first pile kernel as the step's
anchor, the step = from the bucketing kernel before it to the next bucketing kernel, and lists the gaps between one
kernel's end and the next one's start (any stream) that are longer than 8 us."""
import csv
import sys

def short(name):
    return name.replace("rala_hip::", "").replace("(anonymous namespace)::", "").split("(")[0][-45:]


rows = []
for row in csv.DictReader(open(sys.argv[1])):
    rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "bucket_fixed_kernel" in r[2] or "group_count_kernel" in r[2] or "group_count_dedupe_kernel" in r[2] or "shard_count_kernel" in r[2]]
if len(starts) < 2:
    sys.exit("fewer than two steps in the trace")
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0][0]
busy_end = step[0][1]
prev = step[0][2]
gaps = []
covered = 0
for s, e, name in step:
    if s > busy_end:
        gaps.append((s - busy_end, busy_end - t0, prev, name))
    if e > busy_end:
        covered += e - max(s, busy_end)
        busy_end = e
        prev = name
total = busy_end - t0
print("step: %.1f us from the first kernel's start to the last one's end, %d kernels, busy %.1f us, idle %.1f us in %d gaps"
      % (total / 1e3, len(step), covered / 1e3, (total - covered) / 1e3, len(gaps)))
for g, at, p, n in sorted(gaps, reverse=True)[:25]:
    print("  %7.1f us at %8.1f us  after %-45s before %s" % (g / 1e3, at / 1e3, short(p), short(n)))
small = sum(g for g, _, _, _ in gaps if g < 8000)
print("gaps below 8 us: %.1f us in all" % (small / 1e3))

# python tools/trace_gaps.py <csv> <substring>: every kernel of the step whose name contains it, in time order
if len(sys.argv) > 2:
    for s0, e0, name in step:
        if sys.argv[2] in name:
            print("  at %8.1f us  %7.1f us  %s" % ((s0 - t0) / 1e3, (e0 - s0) / 1e3, name.replace("rala_hip::", "").replace("(anonymous namespace)::", "").split("(")[0][-60:]))

# python tools/trace_gaps.py <csv> ALL: the whole step in time order with the gap in front of every kernel
if len(sys.argv) > 2 and sys.argv[2] == "ALL":
    end = step[0][0]
    for s0, e0, name in step:
        print("  at %8.1f us  gap %6.1f  dur %7.1f  %s" % ((s0 - t0) / 1e3, (s0 - end) / 1e3, (e0 - s0) / 1e3,
                                                          name.replace("rala_hip::", "").replace("(anonymous namespace)::", "").split("(")[0][-60:]))
        end = max(end, e0)
