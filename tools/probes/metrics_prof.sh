cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o x -- python /root/repo/tools/bench_metrics.py > /tmp/mp.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/mp/**/x_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={100*float(r['TotalDurationNs'])/tot:5.1f}")
PY
