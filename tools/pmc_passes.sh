#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> <kernel-regex> <passes-file> -- <cmd...>
# Runs one rocprofv3 --pmc pass per line of <passes-file> (no trace domains), each under a timeout.
out=$1; regex=$2; passes=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
while read -r counters; do
  [ -z "$counters" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $counters --kernel-include-regex "$regex" --output-format csv -d $out/pass$i -o pmc -- "$@" > $out/pass$i.log 2>&1
  echo "pass $i rc=$?"
done < $passes
