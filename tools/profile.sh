#!/bin/bash
# tools/profile.sh TAG [CONFIG] -- rocprofv3 passes over `bench.py [--config CONFIG]` on the GPU box (run through gpurun).
# Pass 1: kernel trace + stats.  Passes 2..4: PMC counters, one block group per run (SQ <= 8, TCC <= 4).
set -u
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CFG=${2:-2}
B="python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
# the kernel trace takes more launches, so that its average is not dominated by the first ones after the upload
BT="python3 bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --no-extra"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BT > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- $B > $OUT/pmc_sq.log 2>&1 || { echo "sq pass failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_sq2 -o sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || { echo "sq2 pass failed"; tail -5 $OUT/pmc_sq2.log; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $B > $OUT/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o write -- $B > $OUT/pmc_write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/pmc_write.log; exit 1; }

timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_IFETCH --output-format csv -d $OUT/pmc_lds -o lds -- $B > $OUT/pmc_lds.log 2>&1 || { echo "lds pass failed"; tail -5 $OUT/pmc_lds.log; }
find $OUT -name "*.csv" | head -40
