#!/bin/bash
# Builds tools/variants/libradian_hip_r1decode.so: the current library with ROUND 1's decode.hip (commit 05d21b9) linked in place of
# this round's -- the "old" arm of tools/prof_decode.sh.  Run in the build container (needs .git); the .so travels with gpurun.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp "$R"/radian_amd/csrc/*.hip "$R"/radian_amd/csrc/common.h "$T"/
git -C "$R" show 05d21b9:radian_amd/csrc/decode.hip > "$T"/decode.hip
sed -i "s#\"../../include/radian_hip.h\"#\"$R/include/radian_hip.h\"#" "$T"/api.hip
cd "$T"
for f in api forward decode assemble preprocess; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I/opt/rocm/include -c $f.hip -o $f.o &
done
wait
mkdir -p "$R"/tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R"/tools/variants/libradian_hip_r1decode.so api.o forward.o decode.o assemble.o preprocess.o -ldl
rm -rf "$T"
echo "$R/tools/variants/libradian_hip_r1decode.so"
