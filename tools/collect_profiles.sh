#!/bin/bash
# one GPU-box call that produces the round's evidence under gpurun_out/$R/: the bench line, the rocprofv3 kernel trace +
# stats of the same command, the PMC passes of the dominant launch (separate runs, --kernel-trace only), the C4 probe,
# the HBM-kernel table and the training step timings.
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err || echo "bench rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-prec > $O/prof_bench.log 2>&1
# the per-layer table needs the launches of one generator call in order: one lane
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_1lane -- python bench.py --lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-second-prec > $O/prof_bench_1lane.log 2>&1
python tools/layer_table.py $(ls $O/prof_bench_1lane/*/*_kernel_trace.csv | tail -1) > $O/per_layer_1lane.txt 2>&1
python bench.py --lanes 1 --no-cpu-baseline > $O/bench_1lane.json 2> $O/bench_1lane.err || echo "bench 1 lane rc $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python tools/roofline_probe.py 2 6 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python tools/roofline_probe.py 2 6 > $O/pmc_write.log 2>&1
python bench.py --workload c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err || echo "bench c4 rc $?"
python tools/probe_8x.py 2 3 > $O/c4_probe.txt 2>&1
python tools/probe_small.py > $O/small_layers.txt 2>&1
python tools/probe_transpose.py > $O/hbm_kernels.md 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_wgrad_fetch -- python tools/roofline_probe_wgrad.py 4 > $O/pmc_wgrad_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wgrad_write -- python tools/roofline_probe_wgrad.py 4 > $O/pmc_wgrad_write.log 2>&1
python bench_train.py > $O/bt_c3.json 2> $O/bt_c3.err || true
python bench_train.py --tile 64 > $O/bt_c3_64.json 2> $O/bt_c3_64.err || true
python bench_train.py --workload c5 > $O/bt_c5.json 2> $O/bt_c5.err || true
# C5 at the "512-slice stage": 16 tiles of 512^2 (tileSize 64)
python bench_train.py --workload c5 --tile 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/bt_c5_tile64.json 2> $O/bt_c5_tile64.err || true
ls -R $O | head -60
