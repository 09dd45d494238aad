#!/bin/bash
# A second build of the library for a same-box A/B:  bench/tools/build_variant.sh NAME [hipcc flags, e.g. -DH2_ACC9_WAVES=4]
# (-DH2_AB=1 makes the environment switches live in it: the shipped library reads none -- csrc/common.h; `make -C halo2_amd/csrc ab`
#  builds exactly that as build/ab/libhalo2_mi355x_ab.so)
#   -> build/ab/lib_NAME.so   (objects under build/ab/obj_NAME/; the shipped library and build/obj are not touched)
# With SRC_REV=<git rev> the sources of halo2_amd/csrc and include/ are taken from that revision instead of the working tree
# (e.g. SRC_REV=HEAD~1 bench/tools/build_variant.sh before).  Then, on the GPU box (seconds per run, no Python):
#   bench/tools/ab_native.sh build/ab/lib_NAME.so halo2_amd/libhalo2_mi355x.so commit 20 100
set -e
NAME=$1; shift || { echo "usage: $0 NAME [hipcc flags]"; exit 2; }
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/halo2_amd/csrc
OUT=$ROOT/build/ab
mkdir -p $OUT/obj_$NAME
if [ -n "$SRC_REV" ]; then
  rm -rf $OUT/src_$NAME && mkdir -p $OUT/src_$NAME/halo2_amd $OUT/src_$NAME
  (cd $ROOT && git archive "$SRC_REV" halo2_amd/csrc include | tar -x -C $OUT/src_$NAME)
  SRC=$OUT/src_$NAME/halo2_amd/csrc
fi
FILES=$(cd $SRC && ls *.hip | sed "s/\.hip$//" | tr "\n" " ")      # every unit of that revision (round 6 split msm.hip)
for f in $FILES; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $SRC/$f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$NAME.so $(for f in $FILES; do echo $OUT/obj_$NAME/$f.o; done) -ldl
ls -la $OUT/lib_$NAME.so
