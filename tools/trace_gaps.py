#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace CSV: per kernel name, average duration and the average gap to the
next kernel on the same queue (second half of the trace only: the first half is warm-up)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"].split("(")[0].replace("void nb::", "")
    dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[k].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k in dur:
    d, g = sorted(dur[k]), sorted(gap[k])
    print("%-50s n=%5d  dur med %8.2f us  min %8.2f   gap-after med %6.2f us  -> %8.2f us per launch" % (
        k[:50], len(d), d[len(d) // 2] / 1e3, d[0] / 1e3, g[len(g) // 2] / 1e3, (d[len(d) // 2] + g[len(g) // 2]) / 1e3))
