#!/bin/bash
# Collects the round's measurement artefacts on the GPU box (run through gpurun from the repository root):
#   bench lines, and per workload ("leg") the rocprofv3 kernel statistics + three PMC passes (own runs, no trace flags).
# Results land in gpurun_out/prof_<tag>/<leg>/; copy what should be judged into profiles/<round>/.
#   legs: configs1  scripts/quick_chunked.py 708            BASELINE configs[1], 3 encode + decode passes (k_unpredict3)
#         float1m   scripts/float_chain_time.py 708          the same torus lossless: ONE component, k_unpredict2<float>
#         cfg4share tests/tools/cfg4_check.py 128 221 222    one GPU's share of configs[3]: 128 components, lossless
#         cfg0      scripts/cfg0_time.py                     bunny-class stand-in, lossless, both profiles
#         floatmixed scripts/float_chain_time.py 708 --mixed  ONE mixed-polygon component (40 % quads, 5 % pentagons), lossless
set -u
TAG=${1:-r5}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/chunked_bench.json 2> $OUT/chunked_bench.err
python3 $ROOT/bench.py --profile compat --steps 2 --warmup 1 --no-large > $OUT/compat_bench.json 2> $OUT/compat_bench.err
mkdir -p $OUT/bench
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench/kstats -- python3 $ROOT/bench.py --no-cpu-baseline --no-large > $OUT/bench/kstats.log 2>&1
leg() {
	name=$1; shift
	mkdir -p $OUT/$name
	rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/kstats -- python3 "$@" > $OUT/$name/kstats.log 2>&1
	rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/pmc_fetch -- python3 "$@" > $OUT/$name/pmc_fetch.log 2>&1
	rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/pmc_write -- python3 "$@" > $OUT/$name/pmc_write.log 2>&1
	rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/$name/pmc_sq -- python3 "$@" > $OUT/$name/pmc_sq.log 2>&1
	echo "leg $name done" >> $OUT/progress.log
}
# PART=2: only the 100 M-triangle collections (configs[3] / [4] at the named size; a call of their own: they take minutes)
if [ "${PART:-1}" = "2" ]; then
	cd $ROOT
	HRY_TRACE=1 python3 tests/tools/cfg4_check.py 1024 221 222 --contexts 8 > $OUT/cfg4_full_100M_trace.txt 2>&1
	echo "100M trace + oracle done" >> $OUT/progress.log
	python3 tests/tools/cfg4_check.py 1024 221 222 --no-verify --compat > $OUT/cfg4_full_100M_compat.txt 2>&1
	echo "100M compat done" >> $OUT/progress.log
	HRY_TRACE=1 python3 tests/tools/cfg4_e2e.py 1024 221 222 --gpus 8 > $OUT/cfg4_e2e_100M.txt 2>&1
	python3 tests/tools/cfg4_e2e.py 1024 221 222 > $OUT/cfg4_e2e_100M_one_context.txt 2>&1
	echo "100M end to end done" >> $OUT/progress.log
	cd /tmp
	mkdir -p $OUT/cfg4full
	# (--profile-run: exactly two encodes from resident inputs and two decodes, so that "per pass" means the same for every kernel)
	rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4full/kstats -- python3 $ROOT/tests/tools/cfg4_check.py 1024 221 222 --no-verify --profile-run > $OUT/cfg4full/kstats.log 2>&1
	echo "100M kernel stats done" >> $OUT/progress.log
	rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cfg4full/pmc_fetch -- python3 $ROOT/tests/tools/cfg4_check.py 1024 221 222 --no-verify --profile-run > $OUT/cfg4full/pmc_fetch.log 2>&1
	echo "100M FETCH_SIZE done" >> $OUT/progress.log
	rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cfg4full/pmc_write -- python3 $ROOT/tests/tools/cfg4_check.py 1024 221 222 --no-verify --profile-run > $OUT/cfg4full/pmc_write.log 2>&1
	echo "100M WRITE_SIZE done" >> $OUT/progress.log
	rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/cfg4full/pmc_sq -- python3 $ROOT/tests/tools/cfg4_check.py 1024 221 222 --no-verify --profile-run > $OUT/cfg4full/pmc_sq.log 2>&1
	echo "100M profiles done" >> $OUT/progress.log
	cd $ROOT
	echo "==== cfg4full" >> $OUT/summary.txt
	python3 scripts/summarise_profiles.py $OUT/cfg4full 2 >> $OUT/summary.txt 2>&1
	find $OUT -type d \( -name kstats -o -name pmc_fetch -o -name pmc_write -o -name pmc_sq \) -prune -exec rm -rf {} +
	tail -30 $OUT/summary.txt
	exit 0
fi
leg configs1 $ROOT/scripts/quick_chunked.py 708
leg float1m $ROOT/scripts/float_chain_time.py 708
leg cfg4share $ROOT/tests/tools/cfg4_check.py 128 221 222 --no-verify
leg cfg0 $ROOT/scripts/cfg0_time.py
leg floatmixed $ROOT/scripts/float_chain_time.py 708 --mixed
# the OBJ / general-bindings path (row f-3): timing + kernel statistics
python3 $ROOT/scripts/obj_time.py 300 > $OUT/obj_time.txt 2>&1
python3 $ROOT/scripts/obj_time.py 300 12 > $OUT/obj_time_12bits.txt 2>&1
mkdir -p $OUT/obj
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/obj/kstats -- python3 $ROOT/scripts/obj_time.py 300 > $OUT/obj/kstats.log 2>&1
# both bench modes rehearsed with ranks / contexts sharing this box's one GPU (code paths, not measurements)
python3 $ROOT/bench.py --gpus 2 --inprocess --share-device --comps-per-gpu 16 --steps 3 --warmup 1 > $OUT/rehearsal_inprocess_2ctx_one_gpu.json 2> $OUT/rehearsal_inprocess.err
python3 $ROOT/bench.py --gpus 2 --share-device --comps-per-gpu 16 --steps 2 --warmup 1 2> $OUT/rehearsal_ranks.err | grep '^{' > $OUT/rehearsal_2ranks_one_gpu.json
cd $ROOT
# passes of the workload per run (what traffic.json divides by): quick_chunked 3 encode + decode; float_chain_time 1 encode + 4
# decodes (divide by the decodes: the decode kernels are the subject); cfg4_check 2; cfg0_time 3 per profile
for lp in bench:1 configs1:3 float1m:4 cfg4share:2 cfg0:3 floatmixed:4 obj:1; do
	l=${lp%%:*}; n=${lp##*:}
	echo "==== $l" >> $OUT/summary.txt
	python3 scripts/summarise_profiles.py $OUT/$l $n >> $OUT/summary.txt 2>&1
done
# keep what is small: the condensed files, not the raw rocprofv3 trees
find $OUT -type d \( -name kstats -o -name pmc_fetch -o -name pmc_write -o -name pmc_sq \) -prune -exec rm -rf {} +
tail -60 $OUT/summary.txt
