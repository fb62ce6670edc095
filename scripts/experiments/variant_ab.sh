#!/bin/bash
# generic A/B: the committed library (variants/base.so, built from `git stash`-free HEAD by the caller) against the working tree's build
#   bash scripts/experiments/variant_ab.sh <rounds> <a.so> <b.so>     (both under tightly_coupled_sfm_amd/variants/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash scripts/experiments/ab_bench.sh "$@"
