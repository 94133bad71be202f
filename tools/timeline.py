"""Print one training step's kernel timeline from a rocprofv3 --kernel-trace CSV: per launch the queue, start offset,
duration and the gap to the previous kernel on the same queue.  Usage: python tools/timeline.py <kernel_trace.csv> [step]
[first-kernel]: a step ends with the Adam kernel; with ``first-kernel`` (e.g. stem_fwd: inference passes) it starts at the
kernel whose name contains that string instead."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def main():
    path = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if len(sys.argv) > 3:
        starts = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
        lo, hi = starts[which - 1], starts[which]
    else:
        ends = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
        lo, hi = ends[which - 1] + 1, ends[which] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    last_end = {}
    queues = {}
    busy = {}
    for r in step:
        q = r["Queue_Id"]
        qi = queues.setdefault(q, len(queues))
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        gap = s - last_end.get(q, s)
        last_end[q] = e
        busy[qi] = busy.get(qi, 0) + (e - s)
        print(f"q{qi} {s / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap / 1e3:6.1f}  {' ' * (qi * 2)}{short(r['Kernel_Name'])}")
    total = max(int(r["End_Timestamp"]) for r in step) - t0
    print(f"step span {total / 1e3:.1f} us, launches {len(step)}, busy per queue (us): " +
          ", ".join(f"q{k}={v / 1e3:.0f}" for k, v in sorted(busy.items())))


if __name__ == "__main__":
    main()
