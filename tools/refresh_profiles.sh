#!/bin/bash
# One pass over everything profiles/r06_* is made from (run on the GPU box through gpurun; outputs under gpurun_out/refresh/).
set -x
R=/root/repo
OUT=$R/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
cd $R
# the default (precision auto: f32 results, the large products as bf16 x 6 on the bf16 matrix cores) = the driver's command
python bench.py --steps 10 --warmup 3 > $OUT/bench_auto.json 2> $OUT/bench_auto.err
python bench.py --steps 10 --warmup 3 --precision bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
# the f32 MFMA kernels everywhere (a leg of the default run; here as its own line)
python bench.py --steps 10 --warmup 3 --precision f32 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
python tools/probes/x6_probe.py > $OUT/x6_probe.txt 2>&1
python tools/probes/x6_time.py all > $OUT/x6_time.txt 2>&1
python tools/probes/attn_p3_time.py > $OUT/attn_p3_time.txt 2>&1
python tools/probes/x3_probe.py > $OUT/x3_probe.txt 2>&1
python tools/probes/vit_two_streams.py > $OUT/vit_two_streams.txt 2>&1
# train_batch's default = the whole-step graph; --no-step-graph = its autograd fallback (with / without its own graphs)
python tools/bench_head.py --steps 40 --warmup 5 > $OUT/bench_head.json 2>/dev/null
python tools/bench_head.py --steps 20 --no-step-graph > $OUT/bench_head_eager.json 2>/dev/null
for n in 1 2 4 8; do python tools/bench_head.py --steps 40 --warmup 5 --emulate-world $n > $OUT/bench_head_emulated_world$n.json 2>/dev/null; done
python tools/bench_head.py --steps 20 --emulate-world 8 --no-step-graph > $OUT/bench_head_emulated_world8_eager.json 2>/dev/null
python tools/bench_head.py --steps 20 --emulate-world 8 --no-step-graph --text-graph --temporal-graph > $OUT/bench_head_emulated_world8_autograd_graphs.json 2>/dev/null
# the step's middle section as the separate launches of the autograd path (round 5's form) against the fused launches, interleaved
for rep in 1 2; do for u in 0 1; do for n in 1 8; do echo "ACX_STEP_UNFUSED=$u emulate-world $n: $(ACX_STEP_UNFUSED=$u python tools/bench_head.py --steps 60 --warmup 10 --emulate-world $n 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["train_ms_per_step"], "ms/step, loss", d["loss"])')"; done; done; done > $OUT/step_middle_fused_ab.txt 2>&1
# device time stamps in front of every main-stream segment of the replayed step (no profiler attached), with and without the text stream
for n in 1 8; do { echo "# emulate-world $n"; python tools/probes/step_host_trace.py --steps 30 --warmup 5 --emulate-world $n 2>&1 | grep -v '^{' | grep -E 'device|step:' | cut -c1-600 | tail -4;
  echo "# emulate-world $n, ACX_STEP_SKIP_TEXT=1 (main chain only)"; ACX_STEP_SKIP_TEXT=1 python tools/probes/step_host_trace.py --steps 30 --warmup 5 --emulate-world $n 2>&1 | grep -v '^{' | grep device | cut -c1-400 | tail -2; } ; done > $OUT/step_segments.txt 2>&1
ACX_STEP_SKIP_TEXT=1 python tools/bench_head.py --steps 40 --warmup 5 --emulate-world 8 > $OUT/bench_head_emulated_world8_main_chain_only.json 2>/dev/null
ACX_STEP_SKIP_TEXT=1 python tools/bench_head.py --steps 40 --warmup 5 --emulate-world 1 > $OUT/bench_head_emulated_world1_main_chain_only.json 2>/dev/null
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp8 -o dp8 -- python $R/tools/bench_head.py --emulate-world 8 --steps 10 --warmup 2 > /dev/null 2>&1; cp $(find /tmp/pp8 -name '*kernel_stats.csv' | head -1) $OUT/dp8_rank_share_kernel_stats.csv)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp1 -o tr -- python $R/tools/bench_head.py --steps 10 --warmup 2 > /dev/null 2>&1; cp $(find /tmp/pp1 -name '*kernel_stats.csv' | head -1) $OUT/train_step_kernel_stats.csv)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/tl8 -o t -- python $R/tools/bench_head.py --emulate-world 8 --steps 8 --warmup 2 > /dev/null 2>&1; python $R/tools/step_timeline.py $(find /tmp/tl8 -name '*kernel_trace.csv' | head -1) --delim prep_multi_kernel --step 5 --out $OUT/dp8_step_timeline.txt)
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xd -o xd -- python $R/tools/bench_xd.py --head-only --steps 20 > /dev/null 2>&1; cp $(find /tmp/xd -name '*kernel_stats.csv' | head -1) $OUT/xd_bf16_kernel_stats.csv)
python tools/bench_xd.py --steps 6 > $OUT/bench_xd_bf16.json 2>/dev/null
python tools/bench_metrics.py > $OUT/bench_metrics.txt 2>&1
# two rounds each: the first shapes of a process run on cold clocks, read the second round
python tools/gemm_bench.py --frames 512 --epi 1 --inplace --rounds 2 > $OUT/gemm_f32.txt 2>&1
python tools/gemm_bench.py --frames 512 --epi 1 --inplace --prec bf16 --rounds 2 > $OUT/gemm_bf16.txt 2>&1
python tools/gemm_bench.py --frames 512 --epi 1 --prec bf16 --cbf16 --shapes qkv,fc --rounds 2 >> $OUT/gemm_bf16.txt 2>&1
python tools/conv_bench.py > $OUT/conv_bench.txt 2>&1
python tools/attn_bench.py > $OUT/attn.txt 2>&1
python tools/attn_bf16_bench.py >> $OUT/attn.txt 2>&1
python tools/probes/selector_bench.py > $OUT/selector_bench.txt 2>&1
python tools/text_gemm_bench.py > $OUT/text_gemm.txt 2>&1
python tools/tn_bench.py > $OUT/tn_bench.txt 2>&1
python tools/bench_preprocess.py 2>/dev/null | tail -1 > $OUT/preprocess.json
python tools/bench_preprocess.py --hw 720x1280 --frames 64 --cpu-frames 8 2>/dev/null | tail -1 >> $OUT/preprocess.json
python tools/bench_feature_stream.py 2>/dev/null | tail -1 > $OUT/feature_stream.json
ACX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_gloo2_smoke.json 2> $OUT/bench_gloo2_smoke.err
bash tools/profile_bench.sh auto > $OUT/profile_bench_auto.log 2>&1
bash tools/profile_bench.sh f32 > $OUT/profile_bench.log 2>&1
bash tools/profile_bench.sh bf16 > $OUT/profile_bench_bf16.log 2>&1
python tools/bench_head.py --steps 40 --warmup 5 --precision f32 > $OUT/bench_head_f32.json 2>/dev/null
python tools/bench_head.py --steps 40 --warmup 5 --emulate-world 8 --precision f32 > $OUT/bench_head_f32_emulated_world8.json 2>/dev/null
bash tools/profile_extra.sh > $OUT/profile_extra.log 2>&1
ls -la $OUT
