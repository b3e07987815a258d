"""Merge rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES; one pass each, kernel-trace only) into a
per-kernel table.  HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: counters are in KiB, and on gfx950
FETCH_SIZE reports half of a wide coalesced read stream, so read bytes = 2 * FETCH_SIZE * 1024.
usage: python tools/summarize_pmc.py <fetch.csv> <write.csv> <mfma.csv> <out.csv> <traffic.json>"""
import collections
import csv
import json
import os
import sys

from summarize_prof import short


def load(path, counter):
    agg, dur = collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return agg, dur


def main():
    f, df = load(sys.argv[1], "FETCH_SIZE")
    w, _ = load(sys.argv[2], "WRITE_SIZE")
    m, dm = load(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES")
    rows = []
    traffic = {}
    for k in sorted(f, key=lambda k: -sum(df[k])):
        n = len(f[k])
        rd = 2.0 * sum(f[k]) / n * 1024
        wr = sum(w.get(k, [0.0])) / max(1, len(w.get(k, [0.0]))) * 1024
        d = sum(df[k]) / n
        mf = sum(m.get(k, [0.0])) / max(1, len(m.get(k, [0.0])))
        dmu = sum(dm.get(k, [1.0])) / max(1, len(dm.get(k, [1.0])))
        util = mf / (1024 * dmu * 1e-9 * 2.1e9) if dmu else 0.0     # 1024 SIMDs, ~2.1 GHz under load
        rows.append([k, n, f"{d / 1e3:.1f}", f"{rd / 1e6:.1f}", f"{wr / 1e6:.1f}", f"{(rd + wr) / d:.0f}", f"{100 * util:.1f}"])
        traffic[k] = {"hbm_bytes_per_launch": rd + wr, "avg_us": d / 1e3, "launches": n}
    with open(sys.argv[4], "w", newline="") as fo:
        fo.write("# rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES} (separate passes) -- "
                 "python bench.py --steps 3 --warmup 1; read MB = 2*FETCH_SIZE KiB (gfx950 correction)\n")
        wtr = csv.writer(fo)
        wtr.writerow(["kernel", "launches", "avg_us", "hbm_read_MB_per_launch", "hbm_write_MB_per_launch", "hbm_GBps",
                      "mfma_busy_pct_of_1024_SIMDs"])
        wtr.writerows(rows)
    # which kernel sources the passes were taken from: bench.py reports `traffic` only while they are the ones it runs
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    traffic["_meta"] = {"csrc_sha16": bench.csrc_sha16()}
    json.dump(traffic, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
