#!/bin/bash
# kernel list of the UCF head's test-mode forward at one tile (512 rows): the part of the headline step behind the ViT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/he.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch, bench as B
dev = torch.device("cuda", 0)
net, sd, eot, hc = B.build_net("f32", dev); net.load_from_features = True
nc = torch.zeros(512, device=dev); feats = torch.randn(1, 1, 512, 512, device=dev) * 0.3
with torch.no_grad():
    for _ in range(20): net(feats, None, nc, 1, True)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/he -o x -- python /tmp/he.py > /tmp/he.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/he/**/x_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); n = 20
print("kernel time per forward (us):", round(tot / 1e3 / n, 1), " launches per forward:", sum(int(r["Calls"]) for r in rows) / n)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{r['Name'][:96]:96s} calls/fwd={int(r['Calls'])/n:5.1f} avg_us={float(r['AverageNs'])/1e3:7.1f} us/fwd={float(r['TotalDurationNs'])/1e3/n:7.1f}")
PY
