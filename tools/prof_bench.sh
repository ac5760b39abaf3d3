#!/bin/bash
# usage: tools/prof_bench.sh <label> [bench args...]   (run on the GPU box from the repo root)
# kernel-trace stats + three PMC passes of bench.py; summaries land in gpurun_out/<label>_*
L=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${L}_trace -o t -- python3 $R/bench.py --no-extras --cpu-budget 0 "$@" > $R/gpurun_out/${L}_bench_profiled.json 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${L}_pmc1 -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${L}_pmc2 -o p -- python3 $R/bench.py --no-extras --cpu-budget 0 --steps 10 --warmup 2 "$@" > /dev/null 2>&1
cd $R
python3 tools/summarise_pmc.py $L
