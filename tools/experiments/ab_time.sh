#!/bin/bash
# Development aid: 4K frame times of library variants (build/librt_<name>.so; "base" = the in-tree library) on one box.
# usage: tools/experiments/ab_time.sh "<time_render.py arguments>" <name> [<name> ...]
R=$GRAFT_REPO_ROOT; args=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_HOTPATH_LIB; else export RT_HOTPATH_LIB=$R/build/librt_$v.so; fi
  timeout -k 10 300 python3 $R/tools/time_render.py $args 2>&1 | grep ms/frame | sed "s/^/$v: /" || exit 1
done
