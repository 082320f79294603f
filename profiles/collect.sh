#!/bin/bash
# Collect the round's profile set on the GPU box (run through gpurun from the repo root):
#   bf16 headline:  kernel trace + stats of the bench command, FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (separate runs, counters
#                   only with --kernel-trace), the full bench line;
#   train step:     the same four passes of `bench.py --mode train` (BASELINE config 4);
#   fp32 mode:      kernel trace of `bench.py --dtype fp32`.
# Outputs land in gpurun_out/<tag>/ with the names profiles/ uses (r03_...); copy what should be judged into profiles/.
# usage: profiles/collect.sh [tag=r04] [what=all|bf16|train|trainx3|fp32|x3]
set -e
tag=${1:-r04}; what=${2:-all}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
if [ $what = all ] || [ $what = bf16 ]; then
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-extras --no-repeats > $out/trace.log 2>&1
python profiles/trace_by_shape.py $out/trace/t_kernel_trace.csv auto > $out/${tag}_bf16_bs32_trace_by_shape.txt
cp $out/trace/t_kernel_stats.csv $out/${tag}_bf16_bs32_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-repeats > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-repeats > $out/write.log 2>&1
python profiles/pmc_traffic.py $out/fetch/f_counter_collection.csv $out/write/w_counter_collection.csv $out/${tag}_bf16_bs32_pmc_traffic.json > $out/${tag}_bf16_bs32_pmc_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/mfma -o m --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-repeats > $out/mfma.log 2>&1
python profiles/pmc_mfma_util.py $out/mfma/m_counter_collection.csv $out/${tag}_bf16_bs32_pmc_mfma_util.json > $out/${tag}_bf16_bs32_pmc_mfma_util.txt
echo "bf16 passes done"
fi
if [ $what = all ] || [ $what = train ]; then
rocprofv3 --kernel-trace --stats -d $out/ttrace -o t --output-format csv -- python3 bench.py --mode train --steps 6 --warmup 2 --no-extras > $out/ttrace.log 2>&1
python profiles/trace_by_shape.py $out/ttrace/t_kernel_trace.csv 8 > $out/${tag}_train_trace_by_shape.txt
cp $out/ttrace/t_kernel_stats.csv $out/${tag}_train_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/tfetch -o f --output-format csv -- python3 bench.py --mode train --steps 2 --warmup 1 --no-extras > $out/tfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/twrite -o w --output-format csv -- python3 bench.py --mode train --steps 2 --warmup 1 --no-extras > $out/twrite.log 2>&1
python profiles/pmc_traffic.py $out/tfetch/f_counter_collection.csv $out/twrite/w_counter_collection.csv $out/${tag}_train_pmc_traffic.json > $out/${tag}_train_pmc_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/tmfma -o m --output-format csv -- python3 bench.py --mode train --steps 2 --warmup 1 --no-extras > $out/tmfma.log 2>&1
python profiles/pmc_mfma_util.py $out/tmfma/m_counter_collection.csv $out/${tag}_train_pmc_mfma_util.json > $out/${tag}_train_pmc_mfma_util.txt
echo "train passes done"
fi
if [ $what = all ] || [ $what = trainx3 ]; then          # the train step with the loss network in split-bf16 (custom_loss.set_compute_dtype("bf16x3"))
rocprofv3 --kernel-trace --stats -d $out/t3trace -o t --output-format csv -- python3 bench.py --mode train --loss-dtype x3 --steps 6 --warmup 2 --no-extras > $out/t3trace.log 2>&1
python profiles/trace_by_shape.py $out/t3trace/t_kernel_trace.csv 8 > $out/${tag}_train_x3loss_trace_by_shape.txt
cp $out/t3trace/t_kernel_stats.csv $out/${tag}_train_x3loss_kernel_stats.csv
echo "train (x3 loss network) pass done"
fi
if [ $what = all ] || [ $what = x3 ]; then
rocprofv3 --kernel-trace --stats -d $out/xtrace -o t --output-format csv -- python3 bench.py --dtype x3 --steps 8 --warmup 2 --no-extras --no-repeats > $out/xtrace.log 2>&1
python profiles/trace_by_shape.py $out/xtrace/t_kernel_trace.csv auto > $out/${tag}_x3_bs32_trace_by_shape.txt
cp $out/xtrace/t_kernel_stats.csv $out/${tag}_x3_bs32_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/xfetch -o f --output-format csv -- python3 bench.py --dtype x3 --steps 2 --warmup 1 --no-extras --no-repeats > $out/xfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/xwrite -o w --output-format csv -- python3 bench.py --dtype x3 --steps 2 --warmup 1 --no-extras --no-repeats > $out/xwrite.log 2>&1
python profiles/pmc_traffic.py $out/xfetch/f_counter_collection.csv $out/xwrite/w_counter_collection.csv $out/${tag}_x3_bs32_pmc_traffic.json > $out/${tag}_x3_bs32_pmc_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/xmfma -o m --output-format csv -- python3 bench.py --dtype x3 --steps 2 --warmup 1 --no-extras --no-repeats > $out/xmfma.log 2>&1
python profiles/pmc_mfma_util.py $out/xmfma/m_counter_collection.csv $out/${tag}_x3_bs32_pmc_mfma_util.json > $out/${tag}_x3_bs32_pmc_mfma_util.txt
echo "x3 passes done"
fi
if [ $what = all ] || [ $what = fp32 ]; then
rocprofv3 --kernel-trace --stats -d $out/ftrace -o t --output-format csv -- python3 bench.py --dtype fp32 --steps 8 --warmup 2 --no-extras > $out/ftrace.log 2>&1
python profiles/trace_by_shape.py $out/ftrace/t_kernel_trace.csv auto > $out/${tag}_fp32_bs32_trace_by_shape.txt
echo "fp32 pass done"
fi
# the raw traces are large: keep the summaries only
rm -rf $out/t3trace $out/xtrace $out/xfetch $out/xwrite $out/xmfma $out/trace $out/fetch $out/write $out/mfma $out/ttrace $out/tfetch $out/twrite $out/tmfma $out/ftrace
ls $out
