#!/bin/bash
# Collect the round's profile set on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the bench command, FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (separate runs, counters only
#   with --kernel-trace), the full bench line.  Outputs land in gpurun_out/<tag>/; copy what should be judged into profiles/.
set -e
tag=${1:-v5}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-extras > $out/trace.log 2>&1
python profiles/trace_by_shape.py $out/trace/t_kernel_trace.csv 23 > $out/trace_by_shape.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-extras > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-extras > $out/write.log 2>&1
python profiles/pmc_traffic.py $out/fetch/f_counter_collection.csv $out/write/w_counter_collection.csv $out/pmc_traffic.json > $out/pmc_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/mfma -o m --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-extras > $out/mfma.log 2>&1
python profiles/pmc_mfma_util.py $out/mfma/m_counter_collection.csv $out/pmc_mfma_util.json > $out/pmc_mfma_util.txt
python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json
