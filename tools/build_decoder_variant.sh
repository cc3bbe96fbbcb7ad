#!/bin/bash
# Builds tools/variants/libradian_hip_<tag>decode.so: the CURRENT library with decode.hip / glibc_math.h / glibc_tables.h of another
# commit linked in place of this tree's -- the "old" arm of tools/prof_decode.sh (OLD_LIB=...).  Run in the build container (needs
# .git); the .so travels with gpurun.      usage: tools/build_decoder_variant.sh <commit> <tag>     e.g.  f1c66a7 r2
set -eu
C=$1
TAG=$2
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp "$R"/radian_amd/csrc/*.hip "$R"/radian_amd/csrc/*.h "$T"/
for f in decode.hip glibc_math.h glibc_tables.h; do
  git -C "$R" show $C:radian_amd/csrc/$f > "$T"/$f 2>/dev/null || true
done
# (rd_decode_dev gained a trailing parameter in round 3: give the old definition the same signature)
python3 - "$T"/decode.hip <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
if "n_cu_avail" not in s:
    s = s.replace("const int64_t* d_seq_off2, const int32_t* d_seq_split)\n{", "const int64_t* d_seq_off2, const int32_t* d_seq_split, int n_cu_avail)\n{", 1)
open(p, "w").write(s)
PY
sed -i "s#\"../../include/radian_hip.h\"#\"$R/include/radian_hip.h\"#" "$T"/api.hip "$T"/pipe_reads.hip "$T"/stitch.hip
cd "$T"
OBJS=""
for f in api plan pipe_reads forward decode assemble preprocess stitch; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I/opt/rocm/include -c $f.hip -o $f.o &
  OBJS="$OBJS $f.o"
done
wait
mkdir -p "$R"/tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R"/tools/variants/libradian_hip_${TAG}decode.so $OBJS -ldl
rm -rf "$T"
echo "$R/tools/variants/libradian_hip_${TAG}decode.so"
