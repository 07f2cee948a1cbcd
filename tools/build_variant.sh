#!/bin/bash
# usage (build container): tools/build_variant.sh NAME [-DMACRO ...] — builds tools/variants/libdcrx_NAME.so: the v2 kernels
# (dcrx_kernels_v2.hip) compiled with the given macros, every other object taken from decombinator_amd/csrc/obj (run `make`
# there first).  DCRX_FAST_BUILD=1 in the macros keeps only the 150-nt uniform two-reads-per-lane instantiations (experiments:
# a quarter of the compile time; other launch shapes return an error).  SRC=path: another version of the kernel source (it must sit in csrc/: it includes its neighbours).
# A -DDCRX_EXP_* macro (an experiment branch: work left out, for timing only) makes the build apply tools/experiment_branches.patch
# to a copy of csrc/ first: the shipped sources carry none of those branches.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/decombinator_amd/csrc
O=$R/tools/variants/obj_$NAME
mkdir -p $O
case " $* " in *" -DDCRX_EXP_"*)
  rm -rf $O/csrc; mkdir -p $O/include_up/decombinator_amd; cp -r $C $O/include_up/decombinator_amd/csrc; cp -r $R/include $O/include_up/include
  (cd $O/include_up/decombinator_amd/csrc && patch -p1 -s < $R/tools/experiment_branches.patch)
  SRC=${SRC:-$O/include_up/decombinator_amd/csrc/dcrx_kernels_v2.hip}
  case "$SRC" in $C/*) SRC=$O/include_up/decombinator_amd/csrc/${SRC#$C/};; esac
  ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -x hip -c ${SRC:-$C/dcrx_kernels_v2.hip} -o $O/dcrx_kernels_v2.hip.o
OBJS=""
for f in dcrx_api.cpp dcrx_tables.cpp dcrx_fastq.cpp dcrx_rows.cpp dcrx_collapse.cpp dcrx_translate.cpp dcrx_rccl.cpp dcrx_kernels.hip dcrx_synth.hip; do OBJS="$OBJS $C/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libdcrx_$NAME.so $O/dcrx_kernels_v2.hip.o $OBJS -lz -lpthread -ldl
echo built tools/variants/libdcrx_$NAME.so
