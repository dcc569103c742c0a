#!/bin/bash
# pmc_run.sh <outdir> <counters...> -- <program...>: one rocprofv3 PMC pass (own run, no trace domains besides kernel-trace)
OUT=$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "${CTRS[@]}" -d "$OUT" --output-format csv -- "$@"
