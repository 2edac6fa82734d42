#!/bin/bash
# tools/pmc_one.sh <tag> "<counters>" <bench args...>: one PMC pass, prints the kb_search rows
TAG="$1"; CTRS="$2"; shift; shift
tools/pmc.sh $TAG "$CTRS" "$@"
grep "kb_search" gpurun_out/pmc_$TAG.md | cut -c1-160
