#!/bin/bash
# Per-kernel register / spill / scratch metadata of the gfx950 code object of kernels.hip (device-only compile with the Makefile's
# flags, then the AMDGPU metadata note).  usage: tools/isa_meta.sh > profiles/rNN_isa_meta.txt
set -e
cd "$(dirname "$0")/../minipath_amd/csrc"
make -s kernels.gfx950.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes kernels.gfx950.co | python3 ../../tools/isa_meta.py
