#!/bin/bash
# kernel-tuning helper: build/variants/lib_<tag>.so = the library with multicorrelator.hip recompiled under extra -D flags
# usage: bash profiles/build_variants.sh t64:-DGSH_MC_THREADS=64 t128:-DGSH_MC_THREADS=128 ...
set -e
R=$(cd $(dirname $0)/.. && pwd); P=$R/gnss-sdr_amd
python -c "import sys; sys.path.insert(0,'$R'); import gnss_sdr_amd; gnss_sdr_amd.build_library()"
mkdir -p $R/build/variants
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}; defs=${defs//,/ }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -I$R/include -I$P/csrc $defs -c $P/csrc/multicorrelator.hip -o $R/build/variants/mc_$tag.o 2>/dev/null
  objs=$(ls $P/_build/*.o | grep -v "/multicorrelator.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$tag.so $objs $R/build/variants/mc_$tag.o
  echo built lib_$tag.so "($defs)"
done
