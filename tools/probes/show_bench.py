"""print the headline numbers of a bench.py JSON line (development aid): python tools/probes/show_bench.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "value_vit_two_streams_opt_in", "value_f16x3_mode_opt_in", "value_bf16x3_mode_opt_in", "value_vit_chunk_256") if k in d})
print("roofline", d["roofline"])
h = d.get("head", {})
print("head", {k: (v.get("train_ms_per_step") if isinstance(v, dict) else v) for k, v in h.items()} if isinstance(h, dict) else h)
print("cpu_baseline", d.get("cpu_baseline"))
