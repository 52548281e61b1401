"""Summarises gpurun_out/prof_bench (tools/profile_bench.sh) into profiles/: per-launch averages of the dominant
kernel (acx_gemm's f32 MFMA kernels) with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (x2 for wide
coalesced reads; WRITE_SIZE checked against the exactly-known output bytes)."""
import csv, json, os, sys, collections
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_bench"
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
def rows(d):
    p = os.path.join(src, d, "p_counter_collection.csv")
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("fetch", "write", "sq", "grbm", "l2"):
    for r in rows(d):
        name = r["Kernel_Name"]
        kind = "gemm" if ("gemm_" in name and "reduce" not in name) else (
            "attn" if ("attn_kernel" in name or "attn16_kernel" in name or "attn_bf16_kernel" in name or "attn_p3_kernel" in name) else None)
        if kind is None:
            continue
        agg[kind][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[kind][r["Counter_Name"]] += 1
out = {}
for kind in agg:
    o = {}
    for c, v in agg[kind].items():
        o[c + "_per_launch"] = v / cnt[kind][c]
    o["launches_sampled"] = max(cnt[kind].values())
    if "FETCH_SIZE_per_launch" in o:
        o["hbm_read_bytes_per_launch"] = o["FETCH_SIZE_per_launch"] * 1024 * 2      # KB units, x2 gfx950 correction
    if "WRITE_SIZE_per_launch" in o:
        o["hbm_write_bytes_per_launch"] = o["WRITE_SIZE_per_launch"] * 1024
    if "hbm_read_bytes_per_launch" in o and "hbm_write_bytes_per_launch" in o:
        o["hbm_bytes_per_launch"] = o["hbm_read_bytes_per_launch"] + o["hbm_write_bytes_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in o and "GRBM_GUI_ACTIVE_per_launch" in o:
        # MFMA busy cycles summed over 1024 SIMDs vs kernel cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        o["mfma_pipe_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / 1024.0 / (o["GRBM_GUI_ACTIVE_per_launch"] / 8.0)
    if "TCC_HIT_sum_per_launch" in o:
        o["l2_hit_rate"] = o["TCC_HIT_sum_per_launch"] / (o["TCC_HIT_sum_per_launch"] + o["TCC_MISS_sum_per_launch"])
    out[kind] = o
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/{tag}_bench_{prec}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
st = os.path.join(src, "stats", "bench_kernel_stats.csv")
if os.path.exists(st):
    import shutil
    shutil.copy(st, f"profiles/{tag}_bench_{prec}_kernel_stats.csv")
    for line in open(os.path.join(src, "stats.log")):
        if line.startswith('{"metric"'):
            open(f"profiles/{tag}_bench_{prec}_under_rocprof.json", "w").write(line)
