#!/bin/bash
# Round-end measurement on the GPU box: tests, bench line, rocprofv3 kernel stats and the HBM / MFMA counters (separate passes).
# usage (from the repo root on the box):  bash scripts/final_measure.sh <tag>
set -o pipefail
tag=${1:-r01_f}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest_gpu.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/sq.log 2>&1; echo "sq rc=$?"
find $out -name "*.csv" | head -20
