#!/bin/bash
set -u
TAG=${1:-epa_time}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python tools/diag/epa_time.py > $OUT/climb.txt 2>&1; cat $OUT/climb.txt | grep overlap
(cd ur_gym_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-value -DURGYM_EPA_SCAN -shared -o build/liburgym_scan.so urgym_hip.hip 2>/dev/null)
URGYM_LIB=$R/ur_gym_amd/csrc/build/liburgym_scan.so python tools/diag/epa_time.py > $OUT/scan.txt 2>&1; grep overlap $OUT/scan.txt
