#!/bin/bash
# builds the library of a git revision (default HEAD) into scratch/ab/<name>/libdcs_hip.so for A/B timing: DCS_LIB_PATH=scratch/ab/<name>/libdcs_hip.so
# usage: scratch/build_base.sh [rev] [name]
set -e
REV=${1:-HEAD}; NAME=${2:-base}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" orb-slam2-dualcam_amd/csrc include build.sh | tar -x -C "$TMP"
(cd "$TMP" && DCS_OUT_DIR="$ROOT/scratch/ab/$NAME" DCS_OBJ_DIR="$TMP/obj" bash build.sh)
rm -rf "$TMP"
