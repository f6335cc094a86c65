#!/bin/bash
# Build HEAD's library into experiments/old (for same-call A/B runs on the GPU box) and the working tree's into lib/.
set -e
cd "$(dirname "$0")/.."
git stash -q
mkdir -p experiments/old
(cd neosr_amd/csrc && NEOSR_AMD_OUT=$PWD/../../experiments/old bash build.sh "$@" 2>&1 | grep "^built")
git stash pop -q
(cd neosr_amd/csrc && bash build.sh "$@" 2>&1 | grep "^built")
