"""Condense a rocprofv3 `*_kernel_stats.csv` into a short table (kernel names trimmed to their identifier).
usage: python tools/summarize_prof.py <kernel_stats.csv> <out.csv> [note]"""
import csv
import re
import sys


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)([A-Za-z_0-9]+)", n)
    if m:
        k = int(m.group(1))
        ident = m.group(2)[:k]
        rest = m.group(2)[k:]
        t = re.match(r"ILi(\d+)E(DF16b|DF16_|f)?", rest)
        return ident + (f"<{t.group(1)},{ {'DF16b': 'bf16', 'DF16_': 'f16'}.get(t.group(2), 'f32') }>" if t else "")
    n = n.split("(")[0]
    n = re.sub(r"at::native::", "aten::", n)
    return n[:90]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        if note:
            f.write(f"# {note}\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", r["Percentage"]])


if __name__ == "__main__":
    main()
