"""Aggregate rocprofv3 counter_collection csv files: mean counter value per kernel name (short). usage: pmc_agg.py <dir>"""
import collections, csv, glob, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_prof import short
for f in sorted(glob.glob(sys.argv[1] + "/*/pmc_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "gemm_tn" not in k and "tn_reduce" not in k and (len(sys.argv) < 3 or sys.argv[2] not in k):
            continue
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), d in agg.items():
        print(f"{k:32s} grid={g:>8s} " + "  ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in d.items()))
