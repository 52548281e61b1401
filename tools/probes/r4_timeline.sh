set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
for n in ${WORLDS:-8 1}; do
  D=/tmp/tl$n
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $GRAFT_REPO_ROOT/tools/bench_head.py --emulate-world $n --text-graph --temporal-graph --steps 8 --warmup 2 ${EXTRA:-} > $GRAFT_REPO_ROOT/$O/bench_head_w$n.json 2> /tmp/err$n.txt)
  T=$(find $D -name '*kernel_trace.csv' | head -1)
  python tools/step_timeline.py $T --step 6 --out $O/timeline_w$n.txt
  python - "$T" $O/trace_w$n.csv.gz <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with gzip.open(sys.argv[2], "wt") as f:
    w = csv.writer(f)
    w.writerow(["Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id", "Kernel_Name"])
    for r in rows:
        w.writerow([r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0"), r["Kernel_Name"][:120]])
PY
  tail -2 /tmp/err$n.txt
done
