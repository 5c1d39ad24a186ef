#!/bin/bash
# Builds a variant of libdcvc_amd.so with extra -D flags (or of another git revision) into tools/_bin/<name>.so for
# tools/probes/core_bench.hip.   usage: tools/build_variant.sh <name> [-DFLAG ...]   |   tools/build_variant.sh <name> --rev <git-rev>
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/dcvc_variant_$name
rm -rf $work && mkdir -p $work
if [ "$1" == "--rev" ]; then
  (cd $root && git archive $2) | tar -x -C $work
  defs=""
else
  (cd $root && tar -c --exclude=.git --exclude=gpurun_out --exclude='*.so' --exclude='_obj*' --exclude=_bin --exclude=tests/golden . ) | tar -x -C $work
  defs="$*"
fi
(cd $work && DCVC_EXTRA_DEFS="$defs" python -c "
from dcvc_amd import build
import os
build.build_cli = lambda verbose=False: None
build.CLI_BIN = build.LIB
print(build.build(force=True))")
mkdir -p $root/tools/_bin
cp $work/dcvc_amd/libdcvc_amd.so $root/tools/_bin/$name.so
rm -rf $work
ls -la $root/tools/_bin/$name.so
