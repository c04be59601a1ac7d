#!/bin/bash
# Steps run ON the GPU box through gpurun: $1 = output tag under gpurun_out/, rest = steps.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_job.sh r04a tests bench'
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
case $what in
  tests)       # the whole GPU suite; the tests append their measured numbers to the parity summary
    rm -f gpurun_out/parity_summary.txt
    EVT_PARITY_SUMMARY=$PWD/$OUT/parity_summary.txt timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest.log 2>&1
    tail -40 $OUT/pytest.log ;;
  tests_x)     # stop at the first failure
    EVT_PARITY_SUMMARY=$PWD/$OUT/parity_summary.txt timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
    tail -40 $OUT/pytest.log ;;
  tests_k=*)   # a subset: tests_k=<pytest -k expression>
    EVT_PARITY_SUMMARY=$PWD/$OUT/parity_summary.txt timeout 1200 python -m pytest tests -m gpu -q -k "${what#tests_k=}" > $OUT/pytest_k.log 2>&1
    tail -40 $OUT/pytest_k.log ;;
  bench)       # the driver's default command
    timeout 1500 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; tail -5 $OUT/bench.err; cat $OUT/bench.json ;;
  bench_short) # headline only
    timeout 900 python bench.py --no-other --no-cpu-baseline 2> $OUT/bench_short.err | tail -1 > $OUT/bench_short.json; cat $OUT/bench_short.json ;;
  envelope)    # bf16-mode free-running agreement with split-precision GEMMs (default) and with exact-fp32 GEMMs (EVT_GEMM=f32):
               # does the split arithmetic explain the distance from the reference's own self-agreement envelope?
    for gm in split f32; do
      echo "== EVT_GEMM=$gm" | tee -a $OUT/envelope_ab.txt
      rm -f $OUT/env_$gm.txt
      EVT_GEMM=$gm EVT_PARITY_SUMMARY=$PWD/$OUT/env_$gm.txt timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -q -k "vivit_b_full_size and bf16" 2>&1 | tail -3 | tee -a $OUT/envelope_ab.txt
      cat $OUT/env_$gm.txt | tee -a $OUT/envelope_ab.txt
    done ;;
  overlap)     # the step's independent resident batches on 1 / 2 / 3 HIP streams (bench.py --overlap), events off, two repetitions
    for rep in 1 2; do for ov in 1 2 3; do
      echo "== rep $rep --overlap $ov" | tee -a $OUT/overlap.txt
      timeout 900 python bench.py --no-other --no-cpu-baseline --no-check --no-exact --no-kernel-events --overlap $ov 2>/dev/null | tail -1 | cut -c1-260 | tee -a $OUT/overlap.txt
    done; done ;;
  smoke)
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
  kbench)
    python scripts/kbench.py 2>&1 | tee $OUT/kbench.txt ;;
  kbench=*)
    python scripts/kbench.py ${what#kbench=} 2>&1 | tee -a $OUT/kbench.txt ;;
  vd_batch)    # ViTDet 672^2: streams per GPU as one batch
    for b in 1 8; do
      echo "== batch $b" | tee -a $OUT/vd_batch.log
      python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs --batch $b 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/vd_batch.log
    done
    for b in 16 32 64; do
      echo "== batch $b" | tee -a $OUT/vd_batch.log
      python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --batch $b 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/vd_batch.log
    done ;;
  vd_trace)    # kernel trace of the graph-replayed one-stream frames -> per-frame totals per kernel
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t672 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs > $GRAFT_REPO_ROOT/$OUT/t672.log 2>&1)
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t1024 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs > $GRAFT_REPO_ROOT/$OUT/t1024.log 2>&1)
    python scripts/trace_summary.py $(find $OUT/t672 -name "*kernel_trace.csv" | head -1) 6000 | tee $OUT/trace_vitdet672_one_stream.txt
    python scripts/trace_summary.py $(find $OUT/t1024 -name "*kernel_trace.csv" | head -1) 3000 | tee $OUT/trace_vitdet1024_one_stream.txt
    find $OUT -name "*kernel_trace.csv" -delete ;;
  vd)          # one-stream ViTDet latency (graph replay)
    python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | tee -a $OUT/vd.log
    python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs 2>&1 | tail -1 | tee -a $OUT/vd.log ;;
  py=*)        # any script with arguments: py=scripts/x.py,--a,1
    IFS=',' read -r -a ARGS <<< "${what#py=}"
    python "${ARGS[@]}" 2>&1 | tee -a $OUT/py.txt ;;
  *) echo "unknown step $what" ;;
esac
done
