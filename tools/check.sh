#!/bin/bash
# build (loud on failure) + CPU suite: run before every commit
set -e
cd "$(dirname "$0")/.."
python -m gdmcf_amd.build || { echo "BUILD FAILED"; exit 1; }
python - <<'PY'
import os, hashlib, sys
sys.path.insert(0, ".")
from gdmcf_amd import build as b
stamp = open(os.path.join(b.CSRC, ".build_stamp")).read()
assert stamp == b._digest(), "stale .so: the build stamp does not match the sources"
print("build stamp matches the sources")
PY
python -m pytest tests -x -q -m "not gpu" 2>&1 | tail -2
