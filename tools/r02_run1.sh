#!/bin/bash
# round-2 first GPU pass: tests, driver-style bench, dist paths on a forced 1-rank group, box census
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
cut -c1-1500 $O/bench_driver.json
FRT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"
cut -c1-600 $O/bench_dist1.json; tail -3 $O/bench_dist1.err
FRT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --sharded-gallery --batch 64 --gallery 1250000 --no-cpu-baseline > $O/bench_sharded.json 2> $O/bench_sharded.err; echo "sharded rc=$?"
cut -c1-600 $O/bench_sharded.json; tail -3 $O/bench_sharded.err
timeout 900 python tools/box_census.py --frames 512 --out $O/box_census.json > $O/census.log 2>&1; echo "census rc=$?"
tail -4 $O/census.log
