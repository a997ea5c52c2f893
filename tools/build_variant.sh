#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [-D...]: an A/B build of the library with one translation unit recompiled under extra
# flags -> graphtap_amd/lib/variants/NAME.so (run with GRAPHTAP_LIB=$PWD/graphtap_amd/lib/variants/NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; file=$2; shift 2
python graphtap_amd/_build.py > /dev/null
mkdir -p graphtap_amd/lib/variants
obj=graphtap_amd/lib/variants/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -c -o $obj graphtap_amd/csrc/$file
objs=""
for f in engine ingest kernels pb dist tcsc_cf diag; do
  if [ "$f.hip" == "$file" ]; then objs="$objs $obj"; else objs="$objs graphtap_amd/lib/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o graphtap_amd/lib/variants/$name.so $objs -ldl
echo graphtap_amd/lib/variants/$name.so
