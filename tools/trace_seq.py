"""Print the kernel sequence (start offset, duration, registers, LDS, name) of a rocprofv3 --kernel-trace CSV around the
n-th launch of a kernel whose name contains a pattern:  python tools/trace_seq.py trace.csv PATTERN [nth] [before] [after]"""
import csv
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    nth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    before = int(sys.argv[4]) if len(sys.argv) > 4 else 14
    after = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
    i = idx[min(nth, len(idx) - 1)]
    lo = max(0, i - before)
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:i + after]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {d:9.1f} us  grid {r['Grid_Size_X']:>10} vgpr {r['VGPR_Count']:>3} lds {r['LDS_Block_Size']:>6} {r['Kernel_Name'][:100]}")


if __name__ == "__main__":
    main()
