export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_mvit
mkdir -p $OUT
R=$PWD
cd /tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -o tr --output-format csv -- python $R/bench.py --arch mvit --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$OUT/trace.log" 2>&1
cd $OUT
python - <<'PY'
import csv,glob,collections
f=glob.glob('trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step only: take the last half
rows=rows[len(rows)//2:]
want=['pool_wgrad','pool_dgrad','pool_fwd','pattn_bwd_kv','pattn_bwd_q','pattn_fwd','rel_bwd_table','rel_fwd','rel_bwd_q','pool_ln_bwd']
out=collections.defaultdict(list)
for r in rows:
    for w in want:
        if w in r['Kernel_Name']:
            out[w].append(((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','')))
for w in want:
    print(w, ' '.join(f"{d:.0f}" for d,_ in out[w]))
PY
find . -name "*kernel_trace.csv" -delete
