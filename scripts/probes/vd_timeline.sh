# detailed kernel timeline of ONE steady-state one-stream ViTDet 672^2 frame (graph replay): rocprofv3 kernel trace + scripts/trace_timeline.py --detail
OUT=gpurun_out/vdtl; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t672 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs > $GRAFT_REPO_ROOT/$OUT/t672.log 2>&1)
python scripts/trace_timeline.py $(find $OUT/t672 -name "*kernel_trace.csv" | head -1) --detail > $OUT/timeline672.txt 2>&1
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
tail -3 $OUT/t672.log; head -20 $OUT/timeline672.txt
