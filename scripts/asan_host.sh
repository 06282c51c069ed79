#!/bin/bash
# AddressSanitizer (default) or UndefinedBehaviorSanitizer over the HOST side of the library (container, no GPU): the .cpp
# files are rebuilt with -fsanitize=..., linked with the regular device objects, and the CPU test files that drive host
# code (containers, trailers, hostile headers, ZFile, the ABI surface) run against that build.  The regular library is put
# back afterwards.
#   bash scripts/asan_host.sh [address|undefined]
set -eu
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
SAN=${1:-address}
OUT=gpurun_out/san_$SAN
mkdir -p $OUT
if [ "$SAN" = address ]; then RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1); LINK=-shared-libasan;
else RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.ubsan_standalone-x86_64.so" | head -1); LINK=; fi
for f in video_io_abi registration_abi signal_processing_abi runtime file_attributes codec_abi host_copy time_series; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fsanitize=$SAN -fno-omit-frame-pointer -fvisibility=hidden -Ilibrir_amd/csrc -Iinclude \
      -c librir_amd/csrc/$f.cpp -o $OUT/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$SAN $LINK -o $OUT/librir_amd_asan.so $OUT/*.o librir_amd/build/*.hip.o -ldl -lpthread
cp librir_amd/libs/librir_amd.so $OUT/librir_amd.so.keep
trap 'cp $OUT/librir_amd.so.keep librir_amd/libs/librir_amd.so' EXIT
cp $OUT/librir_amd_asan.so librir_amd/libs/librir_amd.so
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 LD_PRELOAD=$RT python -m pytest tests/test_host_io.py tests/test_abi.py tests/test_time_series.py -x -q -s
# the CPU oracle (plain C) under both sanitizers with its own tests
cp oracle/librir_oracle.so $OUT/oracle_keep.so
trap 'cp $OUT/librir_amd.so.keep librir_amd/libs/librir_amd.so; cp $OUT/oracle_keep.so oracle/librir_oracle.so' EXIT
gcc -O1 -g -std=c11 -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o oracle/librir_oracle.so oracle/rir_oracle.c -lm
cp $OUT/librir_amd.so.keep librir_amd/libs/librir_amd.so
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_oracle_golden.py tests/test_codec_oracle.py tests/test_lossy_oracle.py tests/test_registration_oracle.py \
  tests/test_labelling_oracle.py tests/test_time_series.py -x -q
