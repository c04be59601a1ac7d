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
  tests_lib=*) # a subset against another build: tests_lib=<variant>:<pytest -k expression>
    spec=${what#tests_lib=}; v=${spec%%:*}; k=${spec#*:}
    EVT_LIB=$PWD/scripts/probes/bin/libevt_$v.so timeout 1200 python -m pytest tests -m gpu -q -x -k "$k" > $OUT/pytest_$v.log 2>&1
    tail -5 $OUT/pytest_$v.log ;;
  tests_k=*)   # a subset: tests_k=<pytest -k expression>
    EVT_PARITY_SUMMARY=$PWD/$OUT/parity_summary.txt timeout 1200 python -m pytest tests -m gpu -q -k "${what#tests_k=}" > $OUT/pytest_k.log 2>&1
    tail -40 $OUT/pytest_k.log ;;
  bench)       # the driver's default command
    timeout 1500 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; tail -5 $OUT/bench.err; cat $OUT/bench.json ;;
  bench_short) # headline only
    timeout 900 python bench.py --no-other --no-cpu-baseline 2> $OUT/bench_short.err | tail -1 > $OUT/bench_short.json; cat $OUT/bench_short.json ;;
  bench_events) # headline with and without the per-launch HIP events around the gated linears (what the measurement costs)
    for rep in 1 2; do for fl in "" "--no-kernel-events"; do
      echo "== rep $rep $fl" | tee -a $OUT/bench_events.txt
      timeout 900 python bench.py --no-other --no-cpu-baseline --no-check --no-exact $fl 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/bench_events.txt
    done; done ;;
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
  sharp_ab)    # bf16 cast + sharp attention, projection-gate agreement: split-precision q.k^T (default) vs exact fp32 q.k^T vs all-fp32 GEMMs
    for cfg in "EVT_QK_SPLIT=1" "EVT_QK_SPLIT=0" "EVT_GEMM=f32"; do
      echo "== $cfg" | tee -a $OUT/sharp_ab.txt
      rm -f $OUT/sharp_tmp.txt
      env $cfg EVT_PARITY_SUMMARY=$PWD/$OUT/sharp_tmp.txt timeout 900 python -m pytest tests/test_gpu_blocks.py -m gpu -q -k "sharp_bf16_projection" 2>&1 | tail -2 | tee -a $OUT/sharp_ab.txt
      cat $OUT/sharp_tmp.txt | tee -a $OUT/sharp_ab.txt
    done ;;
  smoke)
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
  kbench)
    python scripts/kbench.py 2>&1 | tee $OUT/kbench.txt ;;
  kbench=*)
    python scripts/kbench.py ${what#kbench=} 2>&1 | tee -a $OUT/kbench.txt ;;
  k9_ab)       # K9 (evt_attention_stream): this tree vs the previous commit's package + build (scripts/probes/bin/base_pkg), one
               # stream and eight, + phase profiles (-DEVT_PROF builds)
    BASE=$PWD/scripts/probes/bin/base_pkg
    for b in 1 8; do
      echo "== previous batch=$b" | tee -a $OUT/k9_ab.txt
      EVT_PKG_ROOT=$BASE python scripts/onestream_bench.py --only stream,stream_first --batch $b 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k9_ab.txt
      echo "== tree batch=$b" | tee -a $OUT/k9_ab.txt
      python scripts/onestream_bench.py --only stream,stream_first --batch $b 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k9_ab.txt
    done
    echo "== previous, phase profile" | tee -a $OUT/k9_ab.txt
    EVT_PKG_ROOT=$BASE EVT_LIB=$BASE/libevt_profbase.so python scripts/onestream_bench.py --only stream,stream_first 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k9_ab.txt
    echo "== tree, phase profile" | tee -a $OUT/k9_ab.txt
    EVT_LIB=$PWD/scripts/probes/bin/libevt_prof.so python scripts/onestream_bench.py --only stream,stream_first 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k9_ab.txt ;;
  k9_var)      # K9 build variants: k9_var (uses scripts/probes/bin/libevt_<v>.so for v in $K9_VARIANTS)
    for v in tree ${K9_VARIANTS:-k9lazy0 k9ahead0}; do
      echo "== $v" | tee -a $OUT/k9_var.txt
      for b in 1 8; do
        if [ $v = tree ]; then python scripts/onestream_bench.py --only stream --batch $b; else EVT_LIB=$PWD/scripts/probes/bin/libevt_$v.so python scripts/onestream_bench.py --only stream --batch $b; fi 2>&1 | grep -v amdgpu.ids | sed "s/^/batch $b  /" | tee -a $OUT/k9_var.txt
      done
    done
    EVT_LIB=$PWD/scripts/probes/bin/libevt_prof.so python scripts/onestream_bench.py --only stream 2>&1 | grep -v amdgpu.ids | tee -a $OUT/k9_var.txt ;;
  k9_ablate)   # K9 timing ablations (scripts/build_variant.sh k9a<mask> -DEVT_K9_ABLATE=<mask>): what the gated launch pays for
    python eventful-transformer_amd/build.py > /dev/null
    echo "== tree" | tee -a $OUT/k9_ablate.txt
    for b in 1 8; do python scripts/onestream_bench.py --only stream --batch $b 2>&1 | grep -v amdgpu.ids | sed "s/^/batch $b  /" | tee -a $OUT/k9_ablate.txt; done
    for m in 1 2 3 4 8 15 16; do
      echo "== EVT_K9_ABLATE=$m" | tee -a $OUT/k9_ablate.txt
      for b in 1 8; do EVT_LIB=$PWD/scripts/probes/bin/libevt_k9a$m.so python scripts/onestream_bench.py --only stream --batch $b 2>&1 | grep -v amdgpu.ids | sed "s/^/batch $b  /" | tee -a $OUT/k9_ablate.txt; done
    done ;;
  osb)
    python scripts/onestream_bench.py 2>&1 | tee $OUT/osb.txt ;;
  dense_occ)   # K8 at ViTDet's window shape: resident kernel (both workgroup shapes) and the tiled kernel
    python scripts/dense_occ.py 2>&1 | tee $OUT/dense_occ.txt
    EVT_WINDOW_NW=8 python scripts/dense_occ.py 2>&1 | tee -a $OUT/dense_occ.txt
    EVT_WINDOW_NW=4 python scripts/dense_occ.py 2>&1 | tee -a $OUT/dense_occ.txt
    EVT_DENSE_TILED=1 python scripts/dense_occ.py 2>&1 | tee -a $OUT/dense_occ.txt ;;
  prof_window) # phase timing inside the resident K8 kernel (scripts/build_variant.sh prof -DEVT_PROF first)
    EVT_LIB=$PWD/scripts/probes/bin/libevt_prof.so EVT_WINDOW_NW=8 python scripts/attn_prof.py --dense window 2>&1 | tee $OUT/k8_resident_phase_profile.txt
    EVT_LIB=$PWD/scripts/probes/bin/libevt_prof.so EVT_WINDOW_NW=4 python scripts/attn_prof.py --dense window 2>&1 | tee -a $OUT/k8_resident_phase_profile.txt ;;
  vd_ab)       # one-stream latency with the round's changes switched off one by one
    for env in "" "EVT_DENSE_TILED=1"; do
      echo "== $env" | tee -a $OUT/vd_ab.log
      env $env python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | tee -a $OUT/vd_ab.log
      env $env python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs 2>&1 | tail -1 | tee -a $OUT/vd_ab.log
    done ;;
  vd_trace)    # kernel trace of the graph-replayed one-stream frames -> per-frame totals per kernel
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t672 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs > $GRAFT_REPO_ROOT/$OUT/t672.log 2>&1)
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t1024 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs > $GRAFT_REPO_ROOT/$OUT/t1024.log 2>&1)
    python scripts/trace_summary.py $(find $OUT/t672 -name "*kernel_trace.csv" | head -1) 6000 | tee $OUT/trace_vitdet672_one_stream.txt
    python scripts/trace_summary.py $(find $OUT/t1024 -name "*kernel_trace.csv" | head -1) 3000 | tee $OUT/trace_vitdet1024_one_stream.txt
    find $OUT -name "*kernel_trace.csv" -delete ;;
  vd_env)      # one-stream latency under runtime environment switches of the HIP graph executor
    for env in "" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "GPU_MAX_HW_QUEUES=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "AMD_DIRECT_DISPATCH=0" "ROC_USE_FGS_KERNARG=0"; do
      echo "== $env" | tee -a $OUT/vd_env.log
      env $env python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | cut -c1-230 | tee -a $OUT/vd_env.log
    done ;;
  vd_batch)    # ViTDet 672^2: streams per GPU as one batch, and the windowed blocks' fused projection norm on / off
    for fuse in 1 0; do for b in 1 8; do
      echo "== EVT_FUSE_PROJ_NORM=$fuse batch $b" | tee -a $OUT/vd_batch.log
      EVT_FUSE_PROJ_NORM=$fuse python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs --batch $b 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/vd_batch.log
    done; done
    for b in 16 32 64; do
      echo "== batch $b" | tee -a $OUT/vd_batch.log
      python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --batch $b 2>&1 | tail -1 | cut -c1-260 | tee -a $OUT/vd_batch.log
    done ;;
  vd_trace672=*) # kernel trace of the 672^2 one-stream frames under an environment setting: vd_trace672=EVT_PREFETCH=0
    (cd /tmp && env ${what#vd_trace672=} rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/t672 -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs > $GRAFT_REPO_ROOT/$OUT/t672.log 2>&1)
    python scripts/trace_summary.py $(find $OUT/t672 -name "*kernel_trace.csv" | head -1) 6000 | tee $OUT/trace672_${what#vd_trace672=}.txt
    find $OUT -name "*kernel_trace.csv" -delete; rm -rf $OUT/t672 ;;
  vd_rider)    # rider shapes: workgroups x loads in flight
    for cfg in ${RIDER_CFGS:-"EVT_PREFETCH=0" "EVT_PREFETCH=1" "EVT_PREFETCH_REFS=1" "EVT_PREFETCH=1" "EVT_PREFETCH_REFS=1"}; do
      echo "== $cfg" | tee -a $OUT/vd_rider.log
      env $cfg python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | cut -c100-200 | tee -a $OUT/vd_rider.log
      env $cfg python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs 2>&1 | tail -1 | cut -c110-210 | tee -a $OUT/vd_rider.log
    done ;;
  vd_touch)    # small-GEMM touch loads on / off: cold-weight probe + one-stream latency
    for t in 1 0; do
      echo "== EVT_SMALL_TOUCH=$t" | tee -a $OUT/vd_touch.log
      EVT_SMALL_TOUCH=$t python scripts/probes/gemm_cold_weights.py 2>&1 | grep -v amdgpu | cut -c1-100 | tee -a $OUT/vd_touch.log
      EVT_SMALL_TOUCH=$t python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | cut -c100-200 | tee -a $OUT/vd_touch.log
      EVT_SMALL_TOUCH=$t python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | cut -c100-200 | tee -a $OUT/vd_touch.log
    done ;;
  vd_prefetch) # one-stream latency with / without the weight prefetch on a side stream
    for rep in 1 2; do for pf in 1 0; do
      echo "== EVT_PREFETCH=$pf" | tee -a $OUT/vd_prefetch.log
      EVT_PREFETCH=$pf python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | cut -c1-230 | tee -a $OUT/vd_prefetch.log
      EVT_PREFETCH=$pf python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs 2>&1 | tail -1 | cut -c1-230 | tee -a $OUT/vd_prefetch.log
    done; done ;;
  vd)          # one-stream ViTDet latency (graph replay)
    python scripts/bench_vitdet.py --grid 42 --policy topk --k 256 --graphs 2>&1 | tail -1 | tee -a $OUT/vd.log
    python scripts/bench_vitdet.py --grid 64 --policy threshold --thr 1.0 --cast bfloat16 --frames 8 --graphs 2>&1 | tail -1 | tee -a $OUT/vd.log ;;
  py=*)        # any script with arguments: py=scripts/x.py,--a,1
    IFS=',' read -r -a ARGS <<< "${what#py=}"
    python "${ARGS[@]}" 2>&1 | tee -a $OUT/py.txt ;;
  *) echo "unknown step $what" ;;
esac
done
