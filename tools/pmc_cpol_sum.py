"""Summarise tools/pmc_cpol.sh: per variant and NT GEMM shape (identified by EPI template + grid size), mean counter values and
kernel duration.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 B (x2, MI355X guide).
usage: python tools/pmc_cpol_sum.py gpurun_out/pmc_cpol"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
for tagdir in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(tagdir):
        continue
    tag = os.path.basename(tagdir)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(tagdir, "*", "*", "pmc_counter_collection.csv")) + glob.glob(os.path.join(tagdir, "*", "pmc_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_nt_kernel" not in k:
                continue
            epi = k[k.find("<") + 1:k.find(">")].replace(" ", "")
            key = (epi, r["Grid_Size"])
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                agg[key]["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    print(f"== {tag}")
    for (epi, grid), d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        parts = [f"<{epi}> grid={grid:>8s}"]
        if "FETCH_SIZE" in m:
            parts.append(f"read={2 * m['FETCH_SIZE'] * 1024 / 1e6:8.1f} MB")
        if "WRITE_SIZE" in m:
            parts.append(f"write={m['WRITE_SIZE'] * 1024 / 1e6:8.1f} MB")
        if "TCC_HIT_sum" in m:
            parts.append(f"L2hit={m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("dur_us"):      # as tools/summarize_pmc.py: 1024 SIMDs at ~2.1 GHz
            parts.append(f"mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * m['dur_us'] * 1e-6 * 2.1e9):.3f}")
        if "dur_us" in m:
            parts.append(f"dur={m['dur_us']:.1f} us")
        print("  " + "  ".join(parts))
