#!/bin/bash
# A second build of the product library for an A/B run of a kernel experiment:
#   tools/build_variant.sh <name> "<-D switches>" [file.hip ...]
# copies the shipped build's objects to bellman_amd/lib_<name>/, recompiles the named .hip files (default: fft.hip) with the
# switches and links.  Load it with BELLMAN_HIP_ALLOW_LIB_OVERRIDE=1 BELLMAN_HIP_LIB=bellman_amd/lib_<name>/libbellman_hip.so.
set -e
cd "$(dirname "$0")/../bellman_amd/csrc"
name=$1; shift
extra=$1; shift
files=${@:-fft.hip}
out=../lib_$name
make -j8 >/dev/null
mkdir -p $out/obj
cp -p ../lib/obj/*.o $out/obj/
for f in $files; do rm -f $out/obj/${f%.hip}.o; done
make OUT=$out EXTRA="$extra" -j8 2>&1 | grep -v warning | tail -3
ls -la $out/*.so | awk '{print $5, $9}'
