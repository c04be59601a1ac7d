#!/bin/bash
# GPU job steps for the one-stream (ViTDet) work: $1 = output tag, rest = what to do (see case below)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
case $what in
  stream_tests)
    python -m pytest tests/test_gpu_kernels.py -x -q -k "attention_stream" > $OUT/t_stream.log 2>&1; tail -15 $OUT/t_stream.log
    python -m pytest tests/test_gpu_blocks.py -x -q -k "vitdet" > $OUT/t_vitdet.log 2>&1; tail -15 $OUT/t_vitdet.log ;;
  all_tests)
    python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log ;;
  vd)
    for g in "--graphs"; do
      python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 $g 2>&1 | tail -1 | tee -a $OUT/vd.log
      python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 $g 2>&1 | tail -1 | tee -a $OUT/vd.log
    done ;;
  vd_trace)
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t672 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs > $GRAFT_REPO_ROOT/$OUT/t672.log 2>&1)
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t1024 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs > $GRAFT_REPO_ROOT/$OUT/t1024.log 2>&1)
    python scripts/trace_summary.py $OUT/t672/t_kernel_trace.csv 6000 | tee $OUT/t672_summary.txt
    python scripts/trace_summary.py $OUT/t1024/t_kernel_trace.csv 3000 | tee $OUT/t1024_summary.txt
    rm -f $OUT/t672/t_kernel_trace.csv $OUT/t1024/t_kernel_trace.csv ;;
  osb)
    python scripts/onestream_bench.py 2>&1 | tee $OUT/osb.txt
    for lib in scripts/probes/bin/libevt_*.so; do
      [ -f $lib ] || continue
      echo "== EVT_LIB=$lib" | tee -a $OUT/osb.txt
      EVT_LIB=$PWD/$lib python scripts/onestream_bench.py --only ${OSB_ONLY:-stream,stream_first} 2>&1 | tee -a $OUT/osb.txt
    done ;;
  bench)
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cat $OUT/bench.json ;;
  *) echo "unknown step $what" ;;
esac
done
