#!/bin/bash
# Developer tool: tools/fixed_base_probe.py (random vs repeated scalar, per W) for several builds of the library.
# usage: WS="23 26" tools/probe_ab.sh <alt.so> ...
cd ${GRAFT_REPO_ROOT:-.}
LIB=babyjubjub-rs_amd/csrc/libbjj_hip.so
cp $LIB /tmp/base.so
for V in /tmp/base.so "$@"; do
  echo "== $V"; cp $V $LIB; python3 tools/fixed_base_probe.py ${WS:-23} 2>&1 | grep -v amdgpu.ids
done
cp /tmp/base.so $LIB
