#!/bin/bash
# ThreadSanitizer over the helper-thread pool of the per-frame entry points (librir_amd/csrc/host_copy.cpp), in the container (no GPU):
# ten calling threads, memory and file jobs, sizes around the hand-off threshold, pauses that let the helpers park; then rir_host_touch beside
# the copies that fill the same pages (its byte compare-and-swap is the one intended race: suppressed by name, the copied bytes are checked).
#   bash scripts/tsan_host_copy.sh
set -eu
cd "$(dirname "$0")/.."
OUT=gpurun_out/tsan
mkdir -p $OUT /tmp/tsan
/opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O1 -g -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Ilibrir_amd/csrc -Iinclude \
    librir_amd/csrc/host_copy.cpp scripts/ubench/tsan_host_copy.cpp -o $OUT/tsan_host_copy -lpthread
TSAN_OPTIONS="suppressions=scripts/tsan_host_copy.supp" $OUT/tsan_host_copy
