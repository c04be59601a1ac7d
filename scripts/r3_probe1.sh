#!/bin/bash
# round 3, first GPU call: where does a ViTDet frame's time go (eager vs graph replay, per-kernel timeline)?
set -u
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for g in "" "--graphs"; do
  python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 $g 2>&1 | tail -1 | tee -a $OUT/vd.log
  python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 $g 2>&1 | tail -1 | tee -a $OUT/vd.log
done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t672 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs > $GRAFT_REPO_ROOT/$OUT/t672.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t1024 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs > $GRAFT_REPO_ROOT/$OUT/t1024.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la $OUT/t672 $OUT/t1024
