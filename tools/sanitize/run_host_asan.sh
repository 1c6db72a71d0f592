#!/bin/bash
# The CPU tests that drive the host half of the C ABI - argument validation, ABI refusal, ctypes marshalling, exchange-context
# arguments - against the host-instrumented library (build_host_asan.sh), with the sanitizer runtime preloaded into the
# interpreter.  Any AddressSanitizer / UBSan report aborts the run (halt_on_error); leak checking is off (CPython itself).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $R
ESPM_MU_LIB=$R/tools/sanitize/libespm_mu_asan.so LD_PRELOAD=$RT \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:verify_asan_link_order=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
python -m pytest tests/test_host_abi_validation.py "tests/test_host_cpu.py::test_query_layout_without_gpu" \
  "tests/test_host_cpu.py::test_argument_errors_map_to_reference_exceptions" "tests/test_host_cpu.py::test_a_drifted_layout_is_refused" -q -p no:cacheprovider "$@"
