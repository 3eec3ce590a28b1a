"""tests/tools/chain_stress.py and tests/tools/obj_stress.py inside the GPU suite: random noisy meshes (few bits, heavy noise, short
rings, non-manifold, multi-component, lossless floats incl. magnitudes where float sums overflow or go denormal and coordinates that
are 0 everywhere, reference-format streams, forced small slices of the pipelined decode, 1 - 8 wavefronts per chain) and random OBJ
scenes (regions, shared records, corner lists, both profiles) through the product against the oracle.  chain_stress is the tool that
found the uploaded-half-edge hole of the pipelined decode; sixteen seeds of it and eight of obj_stress run here so that the driver's
GPU run meets meshes no fixed case describes.  Damaged sharded containers: tests/test_gpu_shard.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAVES = [None, "1", "2", "3", "4", "5", "6", "7", "8", "12", "16"]


def _run(tool, args, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", tool), *args], capture_output=True, text=True, env=env, timeout=550)
    return r


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_random_meshes_against_the_oracle(seed):
    env = dict(os.environ)
    for k in ("HRY_NO_PIPELINE", "HRY_PIPELINE_MIN_VERTICES", "HRY_PIPELINE_FACES", "HRY_PIPELINE_SLICE", "HRY_CHAIN_WAVES", "STRESS_LOSSLESS", "STRESS_BIG", "STRESS_SLIVERS", "HRY_DEVICE_ANALYSIS_MIN_FACES", "HRY_PARALLEL_MIN_FACES", "HRY_ENCODE_PIPELINE_BATCH", "HRY_NO_ENCODE_PIPELINE", "HRY_STAGED_FETCH_MIN", "HRY_STAGED_FETCH_SLOT", "HRY_SPLIT_UPLOAD_MIN", "HRY_DECODE_LANES", "HRY_DECODE_COUNTS32"):
        env.pop(k, None)
    if WAVES[seed % len(WAVES)]:
        env["HRY_CHAIN_WAVES"] = WAVES[seed % len(WAVES)]
    if seed % 4 == 0:
        env["STRESS_LOSSLESS"] = "1"     # every mesh lossless: the float chain's speculation, its repairs and its exact fallback
    if seed in (3, 8, 13):
        env["STRESS_SLIVERS"] = "1"      # half the meshes: many components with slivers before and after them (shared work lists, remembered owners)
    if seed in (8, 14):                  # the components' analysis on the device and every component walked in place on the host threads, whatever the size
        env["HRY_DEVICE_ANALYSIS_MIN_FACES"] = "1"
        env["HRY_PARALLEL_MIN_FACES"] = "1"
        if seed == 14:
            env["HRY_ENCODE_PIPELINE_BATCH"] = "1"   # ... and the encode's device side beside the walk, a batch per finished group
            env["HRY_STAGED_FETCH_MIN"] = "1"        # ... and every container fetched through the ring of pinned slots (8 KiB each: many rounds)
            env["HRY_STAGED_FETCH_SLOT"] = "8192"
            env["HRY_SPLIT_UPLOAD_MIN"] = "1"        # ... and every payload up in two parts, the attribute streams' beside the connectivity kernel
    if seed in (2, 9, 11):
        env["HRY_DECODE_LANES"] = "1"    # every stream a lane can decode goes to k_chunk_decode_lanes (16-bit counts in LDS) ...
        if seed == 9:
            env["HRY_DECODE_COUNTS32"] = "1"   # ... or its 32-bit form
    if seed in (5, 11):
        env["STRESS_BIG"] = "1"          # some meshes large enough for the pipelined decode with its production parameters
    r = _run("chain_stress.py", ["40", str(100 + seed)], env)
    assert r.returncode == 0 and r.stdout.strip().endswith("all equal"), (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_random_obj_scenes_against_the_oracle(seed):
    env = dict(os.environ)
    env.pop("HRY_HOST_EVENTS", None)
    if seed == 3:
        env["HRY_HOST_EVENTS"] = "1"     # the parallel container's references collected by the host's loop (the device's: events.hip, every other seed)
    r = _run("obj_stress.py", ["20", str(200 + seed)], env)
    assert r.returncode == 0 and r.stdout.strip().endswith("all ok"), (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("waves,whole", [("12", False), ("16", False), ("5", False), ("1", True), ("8", True)])
def test_headline_mesh_with_every_team_size(waves, whole):
    """Round 6: the chain's team got late tiles prepared beside their neighbours, a one-head path, tile-by-tile stores and up to
    sixteen wavefronts -- the full-size mesh (every kind of tile) with several team sizes, in slices beside the replay and as one
    launch, against the reference-format decode."""
    env = dict(os.environ, HRY_CHAIN_WAVES=waves)
    env.pop("HRY_NO_PIPELINE", None)
    if whole:
        env["HRY_NO_PIPELINE"] = "1"
    r = _run("team_sizes.py", ["708"], env)
    assert r.returncode == 0 and r.stdout.strip().endswith("all equal"), (r.stdout + r.stderr)[-3000:]
