#!/bin/bash
# Library of an EARLIER COMMIT beside the shipped one (A/B on one box, profiles/tools/ab/libs_ab.sh; in the build container):
#   bash profiles/tools/ab/build_at.sh <commit> <name>   ->  vegs_amd/_lib/libvegsrast_<name>.so   (git-ignored; travels with gpurun)
set -e
commit=$1; name=$2
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C $root archive $commit vegs_amd/csrc include | tar x -C $tmp
mkdir -p $tmp/obj
for f in $tmp/vegs_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fPIC -fno-fast-math -Wno-unused-function -w -c $f -o $tmp/obj/$b.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $tmp/obj/*.o -o $root/vegs_amd/_lib/libvegsrast_$name.so
rm -rf $tmp
echo $root/vegs_amd/_lib/libvegsrast_$name.so
