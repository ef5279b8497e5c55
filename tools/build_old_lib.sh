#!/bin/bash
# usage: tools/build_old_lib.sh [commit]      (default HEAD)
# Builds physicl_amd/_lib/libphysicl_hip_old.so from the device sources of <commit> (hipRTC text included), for
# tools/ab_lib.sh: same-box A/B of the working tree's library against that commit's.
set -e
C=${1:-HEAD}
W=$(mktemp -d)
git archive $C physicl_amd/csrc physicl_amd/build.py physicl_amd/__init__.py include | tar -x -C $W
cd $W && python3 - <<'PY'
import importlib.util, os, subprocess
spec = importlib.util.spec_from_file_location("b", "physicl_amd/build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b._generate_rtc_source()
subprocess.check_call([b.HIPCC] + b.FLAGS + [b.SOURCES[0], "-o", "old.so", "-ldl"])
PY
cd - > /dev/null
cp $W/old.so physicl_amd/_lib/libphysicl_hip_old.so
rm -rf $W
echo "physicl_amd/_lib/libphysicl_hip_old.so <- $C"
