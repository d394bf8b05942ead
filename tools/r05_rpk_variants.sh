#!/bin/bash
for v in 1 2 3 4; do echo "== variant $v"; LSTC_LIBRARY=$PWD/build/rpk$v/liblstc_hip.so python tools/act16_debug.py 2>&1 | grep -v "first bad\|value found\|amdgpu.ids" | head -9; done
