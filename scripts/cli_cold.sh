#!/bin/bash
# Development: wall time of the one-shot CLI (a cold process each) on BASELINE configs[1]: PLY -> .hry -> PLY, with HRY_TRACE's timeline.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 - <<'PY'
from harry_amd import meshgen as mg
open("gpurun_out/cfg1.ply", "wb").write(mg.torus(708, 708, seed=1).to_ply())
PY
for prof in chunked compat; do
	for rep in 1 2; do
		echo "encode $prof:"; ( TIMEFORMAT="   %R s wall, %U user, %S sys"; time harry_amd/bin/harry --profile $prof -l1 -q14 gpurun_out/cfg1.ply gpurun_out/cfg1.$prof.hry > /dev/null )
		echo "decode $prof:"; ( TIMEFORMAT="   %R s wall, %U user, %S sys"; time harry_amd/bin/harry gpurun_out/cfg1.$prof.hry gpurun_out/cfg1.$prof.ply > /dev/null )
	done
done
HRY_TRACE=1 harry_amd/bin/harry gpurun_out/cfg1.chunked.hry gpurun_out/cfg1.out.ply 2>&1 | grep -v consumer | head -40
