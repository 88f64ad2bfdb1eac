"""Compare the CPU-baseline figures of two bench.py JSON lines (two consecutive runs must agree within
10 % on every CPU figure):  python tools/compare_bench.py a.json b.json"""
import json
import sys


def walk(x, y, path, out):
    if isinstance(x, dict) and isinstance(y, dict):
        for k in x:
            if k in y:
                walk(x[k], y[k], path + "/" + k, out)
    elif isinstance(x, (int, float)) and isinstance(y, (int, float)) and not isinstance(x, bool) and x:
        if "cpu_baseline" in path and path.rsplit("/", 1)[-1] in ("value", "median_ms", "fwd_ms", "bwd_ms"):
            out.append((path, x, y, y / x))


def main():
    a, b = (json.load(open(p)) for p in sys.argv[1:3])
    rows = []
    walk(a, b, "", rows)
    worst = 0.0
    for path, x, y, r in rows:
        worst = max(worst, abs(r - 1.0))
        print(f"{path:72s} {x:12.5g} {y:12.5g}  ratio {r:.3f}")
    print(f"worst deviation between the two runs: {100 * worst:.1f} %")
    print("headline", a["value"], b["value"], "efficiency_vs_single_gpu", a.get("efficiency_vs_single_gpu"))


if __name__ == "__main__":
    main()
