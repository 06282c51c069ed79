#!/bin/bash
# TEST INFRASTRUCTURE — builds oracle/_ref/librir_ref.so from the UNMODIFIED reference sources
# where they lie under /root/reference (never copied into this repo; oracle/_ref/ is git-ignored).
#
# What is compiled: src/cpp/signal_processing/{BadPixels,Filters,signal_processing}.cpp and
# src/cpp/tools/{SIMD,Misc,Log}.cpp, with the reference Release flags (-O3 -DNDEBUG, no OpenMP,
# no -march: SURVEY.md §2 row 16), plus oracle/ref_driver.cpp (ours).
#
# Two things the reference build system would do that we do by hand, both stated in DESIGN.md:
#  * rir_config.h is produced from the reference's own rir_config.h.in by substituting its five
#    @PROJECT_*@ name/version tokens (cosmetic strings; no arithmetic depends on them);
#  * tools.cpp (handle registry) needs zstd.h and minizip's unzip.h, which this image lacks, so it
#    is NOT built and NOT replaced: the version script below keeps only the symbols we call and
#    --gc-sections discards the three bad_pixels_{create,correct,destroy} wrappers that reference
#    the registry. Their arithmetic (rir::BadPixels) is reached through ref_driver.cpp instead.
#
# The codec half of the reference (h264.cpp -> ffmpeg 7.1 + libx264) is unbuildable here.
set -euo pipefail
R=${RIR_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$R/src/cpp/signal_processing" ]; then
	echo "build_ref: $R not present - skipping (prebuilt oracle/_ref is used if it exists)"
	exit 0
fi
mkdir -p "$OUT"
sed -e 's/@PROJECT_NAME@/librir/' -e 's/@PROJECT_VERSION@/6.1.2/' \
	-e 's/@PROJECT_VERSION_MAJOR@/6/' -e 's/@PROJECT_VERSION_MINOR@/1/' \
	-e 's/@PROJECT_VERSION_PATCH@/2/' "$R/rir_config.h.in" >"$OUT/rir_config.h"
cat >"$OUT/exports.map" <<'EOF'
{
  global:
    translate; gaussian_filter; find_median_pixel; find_median_pixel_mask; hash_bytes;
    extract_times; resample_time_serie; label_image; keep_largest_area;
    ref_*;
  local: *;
};
EOF
# The metadata trailer (tools/FileAttributes.cpp) needs zstd.h and libzstd: both exist in this image
# under /opt/conda (1.4.9; the reference pins 1.5.5 - the zstd frame format is stable).  When they
# are missing the trailer cross-check is simply left out of the build.
ATTR_SRC=()
ATTR_FLAGS=()
ZSTD_INC=${RIR_ZSTD_INCLUDE:-/opt/conda/include}
ZSTD_LIB=${RIR_ZSTD_LIB:-/opt/conda/lib}
if [ -f "$ZSTD_INC/zstd.h" ] && [ -f "$ZSTD_LIB/libzstd.so" ]; then
	mkdir -p "$OUT/zstd_inc"
	cp "$ZSTD_INC/zstd.h" "$OUT/zstd_inc/" # only this header: the rest of that include dir must not shadow system headers
	ATTR_SRC=("$R/src/cpp/tools/FileAttributes.cpp" "$R/src/cpp/tools/ReadFileChunk.cpp")
	ATTR_FLAGS=(-DRIR_REF_WITH_ATTRS -I"$OUT/zstd_inc" -L"$ZSTD_LIB" -Wl,-rpath,"$ZSTD_LIB" -lzstd)
fi
g++ -std=c++14 -O3 -DNDEBUG -fPIC -shared -ffunction-sections -fdata-sections \
	-DBUILD_SIGNAL_PROCESSING_LIB -DBUILD_TOOLS_LIB \
	-I"$OUT" -I"$R/src/cpp/tools" -I"$R/src/cpp/geometry" -I"$R/src/cpp/signal_processing" \
	"$R/src/cpp/signal_processing/BadPixels.cpp" "$R/src/cpp/signal_processing/Filters.cpp" \
	"$R/src/cpp/signal_processing/signal_processing.cpp" \
	"$R/src/cpp/tools/SIMD.cpp" "$R/src/cpp/tools/Misc.cpp" "$R/src/cpp/tools/Log.cpp" \
	"${ATTR_SRC[@]}" "$HERE/ref_driver.cpp" \
	-Wl,--gc-sections -Wl,--version-script="$OUT/exports.map" -Wl,--no-undefined \
	-static-libstdc++ -static-libgcc -lpthread "${ATTR_FLAGS[@]}" \
	-o "$OUT/librir_ref.so"
echo "build_ref: built $OUT/librir_ref.so"

# The reference's ZFile container (video_io/ZFile.cpp: the one video format both libraries exchange) as a library of its
# own: unmodified ZFile.cpp + tools/{ReadFileChunk,FileAttributes,Log,Misc}.cpp + oracle/ref_zfile_driver.cpp (ours: forwarders).  Its
# zstd_* calls (tools.h:185-188; the reference's tools.cpp, which defines them, needs minizip's unzip.h and is not built) are resolved from
# THIS build's libtools.so alias - reference video_io code on this build's `tools`, the drop-in situation.  Needs zstd.h (for
# FileAttributes.cpp) and the product library; left out when either is missing.
LIBS_DIR=$(cd "$HERE/.." && pwd)/librir_amd/libs
if [ ${#ATTR_SRC[@]} -gt 0 ] && [ -e "$LIBS_DIR/libtools.so" ]; then
	cat >"$OUT/exports_zfile.map" <<'EOF2'
{
  global: ref_zfile_*;
  local: *;
};
EOF2
	g++ -std=c++14 -O3 -DNDEBUG -fPIC -shared -ffunction-sections -fdata-sections \
		-DBUILD_IO_LIB -DBUILD_TOOLS_LIB \
		-I"$OUT" -I"$R/src/cpp/tools" -I"$R/src/cpp/video_io" \
		"$R/src/cpp/video_io/ZFile.cpp" "$R/src/cpp/tools/ReadFileChunk.cpp" "$R/src/cpp/tools/FileAttributes.cpp" \
		"$R/src/cpp/tools/Log.cpp" "$R/src/cpp/tools/Misc.cpp" "$HERE/ref_zfile_driver.cpp" \
		-Wl,--gc-sections -Wl,--version-script="$OUT/exports_zfile.map" -Wl,--no-undefined \
		-static-libstdc++ -static-libgcc -lpthread "${ATTR_FLAGS[@]}" \
		-L"$LIBS_DIR" -l:libtools.so -Wl,-rpath,'$ORIGIN/../../librir_amd/libs' \
		-o "$OUT/librir_ref_zfile.so"
	echo "build_ref: built $OUT/librir_ref_zfile.so"
fi
