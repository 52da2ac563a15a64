"""The kernel sequence of ONE replayed step from a rocprofv3 --kernel-trace CSV: every launch between the last two
adam_multi_k launches, in start order, with its duration and the gap to its predecessor.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o x -- python3 bench.py --config C2 --steps 20 ... --no-roofline
    python tools/step_sequence.py <dir>   ->  launches, busy us, gaps us, per-kernel lines"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return re.sub(r"^(v2::igemm2_(?:fwd|bwd|wgrad)<[^>]*), false>$", r"\1>", m.group(1) if m else name)[:70]


def main():
    path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adam_multi_k" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]
    seq = rows[a + 1: b + 1]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq) / 1e3
    span = (int(seq[-1]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
    print(f"launches {len(seq)}  busy {busy:.1f} us  span {span:.1f} us  gaps {span - busy:.1f} us")
    prev = int(rows[a]["End_Timestamp"])
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(e - s) / 1e3:7.1f} us  gap {(s - prev) / 1e3:5.1f}  {short(r['Kernel_Name'])}")
        prev = e


if __name__ == "__main__":
    main()
