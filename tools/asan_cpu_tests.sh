#!/bin/bash
# AddressSanitizer + UBSan over the HOST code of libminipath_hip.so (builder, OBJ loader, array import, camera, tile ordering,
# C-ABI argument handling, exception guard): builds minipath_amd/csrc/libminipath_hip_asan.so and runs the CPU test-suite's host
# tests against it (MINIPATH_HIP_SO).  CPU only; GPU AddressSanitizer is not available on this pool.
set -e
cd "$(dirname "$0")/.."
make -C minipath_amd/csrc asan > /dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export MINIPATH_HIP_SO=$PWD/minipath_amd/csrc/libminipath_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$RT python -m pytest tests/test_host_cpu.py tests/test_device_tree_cpu.py tests/test_golden_cpu.py tests/test_distributed_cpu.py tests/test_io_cpu.py -x -q -m "not gpu" \
    --deselect tests/test_host_cpu.py::test_no_exception_crosses_the_abi "$@"
