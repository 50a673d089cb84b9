#!/bin/bash
# Build libphx.so from the working tree with extra -D flags into tmp_variants/libphx_<name>.so (a copy of the sources in /tmp: the tree's objects stay).
#   bash tools/build_variant.sh <name> "<flags>"
set -e
name=$1; flags=$2
d=/tmp/phxvar_$name
rm -rf $d; mkdir -p $d/phanotate_amd $d/include /root/repo/tmp_variants
cp -r /root/repo/phanotate_amd/csrc $d/phanotate_amd/; cp /root/repo/include/*.h $d/include/
rm -f $d/phanotate_amd/csrc/*.o
make -s -C $d/phanotate_amd/csrc EXTRA="$flags" > $d/build.log 2>&1
cp $d/phanotate_amd/libphx.so /root/repo/tmp_variants/libphx_$name.so
echo built $name
