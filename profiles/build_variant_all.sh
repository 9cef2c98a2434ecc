#!/bin/bash
# build/variants/lib_<tag>.so = the library with EVERY translation unit that includes csrc/mcorr_device.h recompiled under extra -D flags
# usage: bash profiles/build_variant_all.sh rtn:-DGSH_MC_RTN_FLOOR=1 ...
set -e
R=$(cd $(dirname $0)/.. && pwd); P=$R/gnss-sdr_amd
python -c "import sys; sys.path.insert(0,'$R'); import gnss_sdr_amd; gnss_sdr_amd.build_library()"
mkdir -p $R/build/variants
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}; defs=${defs//,/ }
  objs=""
  for o in $P/_build/*.o; do
    b=$(basename $o .o)
    if grep -q "mcorr_device.h\|multicorrelator.hip" $P/csrc/$b.hip; then
      ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I$R/include -I$P/csrc $defs -c $P/csrc/$b.hip -o $R/build/variants/${b}_$tag.o 2>/dev/null ) &
      objs="$objs $R/build/variants/${b}_$tag.o"
    else
      objs="$objs $o"
    fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$tag.so $objs
  echo built lib_$tag.so "($defs)"
done
