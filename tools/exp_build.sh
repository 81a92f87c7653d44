#!/bin/bash
# Builds a variant of libpt355.so + ptbench into exp/<name>/ (git-ignored) for A/B runs on the GPU box.
#   tools/exp_build.sh <name> [extra hipcc flags...]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/exp/$NAME
make -s -C $ROOT/pathtracer_amd/csrc OUT=$ROOT/exp/$NAME/libpt355.so EXTRA="$*" all
echo "built exp/$NAME"
