"""Timeline summary of a rocprofv3 --kernel-trace CSV: for the last `--steps` replays of the bench step, per queue busy
time, the union busy time of the GPU, launch gaps, and the kernels ranked by their share of the step's WALL time
(attributing overlapped intervals to the queue that is the critical path is out of scope: this reports per-queue sums and
the union).  usage: python tools/timeline.py <kernel_trace.csv> [t0_frac t1_frac]"""
import csv
import re
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_prof import short  # noqa: E402


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "0"))))
    ev.sort()
    ends = [e[1] for e in ev if e[2].startswith("adam_kernel")]
    if len(ends) >= 3:          # one optimiser step to the next = one training step: take the last complete one
        a, b = ends[-2], ends[-1]
        print(f"step delimited by adam_kernel: {len(ends)} optimiser steps in the trace, analysing the last one")
    else:
        T0, T1 = ev[0][0], ev[-1][1]
        a, b = T0 + f0 * (T1 - T0), T0 + f1 * (T1 - T0)
    ev = [e for e in ev if e[0] >= a and e[1] <= b]
    span = ev[-1][1] - ev[0][0]
    print(f"window {span / 1e6:.2f} ms, {len(ev)} kernels")
    # union busy
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"GPU busy (union of all queues) {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} % of the window; idle {(span - busy) / 1e6:.2f} ms")
    byq = defaultdict(list)
    for e in ev:
        byq[e[3]].append(e)
    for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        tot = sum(e[1] - e[0] for e in lst)
        gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
        small = [g for g in gaps if 0 < g < 20000]
        print(f"queue {q}: {len(lst)} kernels, busy {tot / 1e6:.2f} ms; gaps<20us: n={len(small)} sum={sum(small) / 1e6:.2f} ms "
              f"median={sorted(small)[len(small) // 2] / 1e3 if small else 0:.2f} us")
    agg = defaultdict(lambda: [0, 0])
    for s, e, n, q in ev:
        agg[(n, q)][0] += 1
        agg[(n, q)][1] += e - s
    # idle gaps of the whole GPU (no queue busy) longer than 3 us, with the kernel that ends / starts them
    ev2 = sorted(ev)
    gaps, cur_e, last = [], None, None
    for s_, e_, n_, q_ in ev2:
        if cur_e is not None and s_ > cur_e + 3000:
            gaps.append((s_ - cur_e, last, n_))
        if cur_e is None or e_ > cur_e:
            cur_e, last = e_, n_
    print(f"GPU-idle gaps > 3 us: {len(gaps)}, sum {sum(g[0] for g in gaps) / 1e6:.2f} ms")
    for g in sorted(gaps, key=lambda t: -t[0])[:12]:
        print(f"   {g[0] / 1e3:8.1f} us after {g[1][:40]} before {g[2][:40]}")
    print("kernel, queue, calls, total_ms, avg_us")
    for (n, q), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{n[:70]:70s} q{q} {c:5d} {t / 1e6:8.3f} {t / c / 1e3:9.2f}")


if __name__ == "__main__":
    main()
