R="$PWD"; OUT=$R/gpurun_out/r05_f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$R"
D=gpurun_out/rp_ct; rm -rf $D
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 tools/coteach_round.py --dtype bf16 --steps 12 > $OUT/coteach_prof.json 2> $OUT/coteach_prof.err
S=$(find $D -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py stats $D $OUT/coteach_kernel_stats.md "coteach round bf16, 12 steps per stage"
T=$(find $D -name "*kernel_trace.csv" | head -1)
python3 - <<'PY' $T > $OUT/trace_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find big gaps: split the trace into segments by > 200 ms idle
t0 = int(rows[0]["Start_Timestamp"])
seg, segs, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 150e6:
        segs.append(seg); seg = []
    seg.append(r); last_end = e
segs.append(seg)
for i, sg in enumerate(segs):
    dur = (int(sg[-1]["End_Timestamp"]) - int(sg[0]["Start_Timestamp"])) / 1e6
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sg) / 1e6
    c = collections.Counter()
    for r in sg:
        c[r["Kernel_Name"][:60]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"segment {i}: {len(sg)} kernels, span {dur:.1f} ms, busy {busy:.1f} ms; top:", [(k, round(v, 1)) for k, v in c.most_common(6)])
PY
rm -rf $D
cat $OUT/trace_summary.txt | cut -c1-900
