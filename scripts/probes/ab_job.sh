# same-box A/B of two builds of the library: $1 = kbench --only list, $2... = variant names under scripts/probes/bin/libevt_<v>.so ("ship" = the in-tree build)
ONLY=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do for v in "$@"; do
  echo "== $v run $rep" | tee -a gpurun_out/ab/ab.txt
  if [ "$v" = ship ]; then python scripts/kbench.py --clips 256 --only $ONLY 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/ab.txt
  else EVT_LIB=$PWD/scripts/probes/bin/libevt_$v.so python scripts/kbench.py --clips 256 --only $ONLY 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/ab.txt; fi
done; done
