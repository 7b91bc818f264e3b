#!/bin/bash
# "the reference's CPU path timed on the host cores of the same box".  The GPU box has no /root/reference: an UNTRACKED working copy of
# reconstruction/ travels with the gpurun snapshot (_refcopy/, git-ignored, removed afterwards; nothing of it is committed), tools/time_reference_cpu.py
# runs the reference's own render() / extract_fields on the box's host cores -> gpurun_out/rNN_cpu_reference_gpubox.json (copy to profiles/).
#   tools/reference_cpu_on_gpu_box.sh [seconds, default 25] [extra gpurun command to append]
set -e
cd "$(dirname "$0")/.."
rm -rf _refcopy && mkdir -p _refcopy
cp -r /root/reference/reconstruction _refcopy/reconstruction
find _refcopy -name '__pycache__' -prune -exec rm -rf {} +
CMD="mkdir -p gpurun_out; O2345_COMMIT=$(git rev-parse --short HEAD) O2345_REFERENCE_DIR=\$PWD/_refcopy/reconstruction python tools/time_reference_cpu.py ${1:-25} r06 gpurun_out _gpubox > gpurun_out/cpu_reference_gpubox.log 2>&1; tail -2 gpurun_out/cpu_reference_gpubox.log"
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-1500} -- "$CMD; $2"
rc=$?
rm -rf _refcopy
exit $rc
