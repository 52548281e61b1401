#!/bin/bash
# Copies what tools/refresh_profiles.sh left under gpurun_out/ into profiles/<tag>_* (run in the development container after
# the gpurun call merged its outputs back).  usage: tools/collect_profiles.sh [tag]   (default r06)
set -e
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out
P=$R/profiles
cd $R
cp $G/refresh/bench_auto.json $P/${TAG}_bench_auto.json
cp $G/refresh/bench_f32.json $P/${TAG}_bench_f32.json
cp $G/refresh/bench_bf16.json $P/${TAG}_bench_bf16.json
grep -v amdgpu.ids $G/refresh/x6_probe.txt > $P/${TAG}_gemm_bf16x6_vs_f32.txt
{ echo "# tools/probes/x6_time.py all (the plane-reuse kernel: ViT shapes at 512 frames, the head's convolutions and weight gradients)"; grep -v amdgpu.ids $G/refresh/x6_time.txt;
  echo "# tools/probes/attn_p3_time.py"; grep -v amdgpu.ids $G/refresh/attn_p3_time.txt; } > $P/${TAG}_x6_kernels.txt
[ -f $G/refresh/x3_probe.txt ] && { echo "# tools/probes/x3_probe.py: the three-product modes (acx_gemm_desc.pairs = 3: precision bf16x3 on the bf16 planes, precision f16x3 on two fp16 planes; both opt-in) against the six-product default: GEMM shapes of the ViT, the encode in every mode and on the f32 MFMA kernels, the attention with six / three products"; grep -v amdgpu.ids $G/refresh/x3_probe.txt; } > $P/${TAG}_bf16x3_mode.txt
[ -f $G/refresh/vit_two_streams.txt ] && { echo "# tools/probes/vit_two_streams.py: the 512-frame encode as n half / third / ... batches on n HIP streams (VisionTransformer.streams = 2 is the product's opt-in form)"; grep -v amdgpu.ids $G/refresh/vit_two_streams.txt; } > $P/${TAG}_vit_two_streams.txt
cp $G/refresh/bench_head_f32.json $P/${TAG}_bench_head_f32.json
cp $G/refresh/bench_head_f32_emulated_world8.json $P/${TAG}_bench_head_f32_emulated_world8.json
cp $G/refresh/bench_head.json $P/${TAG}_bench_head.json
cp $G/refresh/bench_head_eager.json $P/${TAG}_bench_head_eager.json
for n in 1 2 4 8; do cp $G/refresh/bench_head_emulated_world$n.json $P/${TAG}_bench_head_emulated_world$n.json; done
cp $G/refresh/bench_head_emulated_world8_eager.json $P/${TAG}_bench_head_emulated_world8_eager.json
for v in world8_autograd_graphs world8_main_chain_only world1_main_chain_only; do cp $G/refresh/bench_head_emulated_$v.json $P/${TAG}_bench_head_emulated_$v.json; done
cp $G/refresh/dp8_step_timeline.txt $P/${TAG}_dp8_step_timeline.txt
# (the refresh script runs under `set -x`: drop its trace lines)
[ -f $G/refresh/step_middle_fused_ab.txt ] && { echo "# tools/bench_head.py --steps 60 --warmup 10 --emulate-world N: the step's middle as fused launches (default) vs the autograd path's separate launches (ACX_STEP_UNFUSED=1), interleaved on one box"; grep -v '^+' $G/refresh/step_middle_fused_ab.txt; } > $P/${TAG}_step_middle_fused_ab.txt
[ -f $G/refresh/step_segments.txt ] && { echo "# tools/probes/step_host_trace.py: HIP-event time stamps in front of every main-stream graph segment of the replayed training step (no profiler): [temporal forward | selector forward .. loss | (SyncBN exchange pieces at N > 1) | temporal backward | optimizer], and the host time per replayed item"; grep -v '^+' $G/refresh/step_segments.txt; } > $P/${TAG}_step_segments.txt
cp $G/refresh/bench_xd_bf16.json $P/${TAG}_bench_xd_bf16.json
grep '^{' $G/refresh/bench_gloo2_smoke.json > $P/${TAG}_bench_gloo2_smoke.json   # gloo prints its own connection lines to stdout
cp $G/refresh/bench_metrics.txt $P/${TAG}_bench_metrics.txt
cp $G/refresh/preprocess.json $P/${TAG}_preprocess.json
cp $G/refresh/feature_stream.json $P/${TAG}_feature_stream.json
{ echo "# tools/gemm_bench.py --frames 512 --epi 1 --inplace --rounds 2 (f32; second round = warm clocks)"; grep -v amdgpu.ids $G/refresh/gemm_f32.txt;
  echo "# tools/gemm_bench.py --frames 512 --epi 1 --inplace --prec bf16 --rounds 2 (f32 C), then --cbf16 --shapes qkv,fc (the in-model bf16 outputs)"; grep -v amdgpu.ids $G/refresh/gemm_bf16.txt;
  echo "# tools/conv_bench.py"; grep -v amdgpu.ids $G/refresh/conv_bench.txt;
  echo "# tools/tn_bench.py"; grep -v amdgpu.ids $G/refresh/tn_bench.txt;
  echo "# tools/attn_bench.py, tools/attn_bf16_bench.py"; grep -v amdgpu.ids $G/refresh/attn.txt;
  echo "# tools/probes/selector_bench.py"; grep -v amdgpu.ids $G/refresh/selector_bench.txt; } > $P/${TAG}_kernel_microbench.txt
grep -v amdgpu.ids $G/refresh/text_gemm.txt > $P/${TAG}_text_gemm.txt
python tools/summarize_pmc.py gpurun_out/prof_bench_auto $TAG auto > /dev/null
python tools/summarize_pmc.py gpurun_out/prof_bench_f32 $TAG f32 > /dev/null
python tools/summarize_pmc.py gpurun_out/prof_bench_bf16 $TAG bf16 > /dev/null
cp $G/prof_extra/train/t_kernel_stats.csv $P/${TAG}_train_step_kernel_stats.csv
cp $G/prof_extra/dp8/dp8_kernel_stats.csv $P/${TAG}_dp8_rank_share_kernel_stats.csv
cp $G/prof_extra/dp2/dp2_kernel_stats.csv $P/${TAG}_dp2_rank_share_kernel_stats.csv
cp $G/prof_extra/xd/xd_kernel_stats.csv $P/${TAG}_xd_bf16_kernel_stats.csv
cp $G/prof_extra/metrics/m_kernel_stats.csv $P/${TAG}_metrics_epilogue_kernel_stats.csv
ls -la $P/${TAG}_*
