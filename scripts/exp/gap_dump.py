"""Kernel trace -> the longest idle gaps (no kernel in flight) of the last N ms, with the dispatches either side of each
(start/end relative to the gap, stream, kernel) - to tell a host-bound hole from a dependency stall."""
import csv
import sys

path, last_ms, top = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 230.0, int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")), r["Kernel_Name"][:70]))
rows.sort()
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
gaps, cur_end = [], rows[0][1]
for i, r in enumerate(rows[1:], 1):
    if r[0] > cur_end:
        gaps.append((r[0] - cur_end, cur_end, r[0], i))
    cur_end = max(cur_end, r[1])
gaps.sort(reverse=True)
for g, a, b, i in gaps[:top]:
    print(f"--- gap {g / 1e3:.1f} us at t = {(a - rows[0][0]) / 1e6:.3f} ms")
    for r in rows[max(0, i - 6):i + 4]:
        print(f"   {'>>' if r[0] >= b else '  '} start {(r[0] - a) / 1e3:9.1f} us  end {(r[1] - a) / 1e3:9.1f} us  stream {r[2]:>3}  {r[3]}")
