#!/bin/bash
# Collects the round's measurement artefacts on the GPU box (run through gpurun from the repository root):
#   bench line, rocprofv3 kernel statistics of the same command, and PMC passes (own runs, no trace flags).
# Results land in gpurun_out/prof_<tag>/; copy what should be judged into profiles/<round>/.
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/chunked_bench.json 2> $OUT/chunked_bench.err
python3 $ROOT/bench.py --profile compat --steps 2 --warmup 1 > $OUT/compat_bench.json 2> $OUT/compat_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_compat -- python3 $ROOT/bench.py --profile compat --steps 2 --warmup 1 --no-cpu-baseline > $OUT/kstats_compat.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/scripts/quick_chunked.py 708 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/scripts/quick_chunked.py 708 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/scripts/quick_chunked.py 708 > $OUT/pmc_sq.log 2>&1
cd $ROOT
python3 scripts/summarise_profiles.py $OUT 3 > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
