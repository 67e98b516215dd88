#!/bin/bash
# Runs on the GPU box: the randomised parity runs of tests/fuzz_parity.py in the exact near-first order over all five scenes with
# fresh seeds, one log per scene under gpurun_out/fuzz/, and their closing lines in gpurun_out/fuzz/summary.txt.
# usage: [FUZZ_SPECS="scene:views ..."] [FUZZ_ARGS="--eye-in-plane"] tools/fuzz_campaign.sh <seed>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz; mkdir -p $O; seed=${1:-1}
: > $O/summary.txt
for spec in ${FUZZ_SPECS:-"planes:1200" "bunny:1200" "small:600" "s1m:300" "deep:300" "s100k:600"}; do
  set -- ${spec/:/ }
  timeout -k 10 ${FUZZ_TIMEOUT:-900} python3 -u $R/tests/fuzz_parity.py --scene $1 --views $2 --seed $seed --order near_first_exact $FUZZ_ARGS > $O/$1.log 2>&1
  rc=$?
  echo "== $1 (seed $seed): $(grep -c -i mismatch.*pixel $O/$1.log) mismatch lines; $(tail -n 1 $O/$1.log) rc=$rc" >> $O/summary.txt
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi     # anything but "ran and found differences" ends the campaign
done
cat $O/summary.txt
