#!/bin/bash
# Runs the reference's UNCHANGED exp_runner_generic_blender_val.py on the MI355X against the REAL ops (libo2345_hip.so) through the drop-in hook.
# The GPU box has no /root/reference: an UNTRACKED working copy of reconstruction/ travels with the gpurun snapshot (_refcopy/, git-ignored, removed
# afterwards; nothing of it is committed).  Log -> gpurun_out/runner_dropin.log (copy to profiles/ by hand).
#   tools/runner_on_gpu.sh [extra gpurun command to append]
set -e
cd "$(dirname "$0")/.."
rm -rf _refcopy && mkdir -p _refcopy
cp -r /root/reference/reconstruction _refcopy/reconstruction
find _refcopy -name '__pycache__' -prune -exec rm -rf {} +
CMD='mkdir -p gpurun_out; L=gpurun_out/runner_dropin.log; : > $L;
for MODE in "export_mesh --resolution 64" "export_mesh --resolution 256" "val"; do
  echo "=== python -m o2345_amd.dropin exp_runner_generic_blender_val.py --mode $MODE --conf confs/one2345_lod0_val_demo.conf --specific_dataset_name scene0" >> $L;
  ( time timeout 600 python tests/run_reference_runner.py --ref _refcopy/reconstruction --work /tmp/runner_work -- --mode $MODE --conf confs/one2345_lod0_val_demo.conf --specific_dataset_name scene0 ) >> $L 2>&1; echo "rc=$?" >> $L;
done;
echo "=== the way run.py starts it (run.py:61-67): python exp_runner_generic_blender_val.py --mode export_mesh --resolution 256 ..., one-2-3-45_amd/autoload first on PYTHONPATH" >> $L;
( time timeout 600 python tests/run_reference_runner.py --ref _refcopy/reconstruction --work /tmp/runner_work --via-autoload -- --mode export_mesh --resolution 256 --conf confs/one2345_lod0_val_demo.conf --specific_dataset_name scene0 ) >> $L 2>&1; echo "rc=$?" >> $L;
grep -c "RUNNER_RESULT" $L; grep "RUNNER_RESULT\|rc=\|real" $L'
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-1500} -- "$CMD; $1"
rc=$?
rm -rf _refcopy
exit $rc
