// C ABI of the block codec on device-resident batches (format RIRB1).
// Host-pointer / file-level entry points (h264_add_image_lossless, load_image, ...) are in
// video_io_abi.cpp and call these.
#include <cstdlib>
#include <cstring>

#include "codec_format.h"
#include "filter_kernels.h"
#include "rir_amd_device.h"
#include "runtime.h"

using namespace rir;

namespace
{
	// the caller's stream, taken literally: NULL is HIP's null (legacy default) stream
	hipStream_t as_stream(void *s) { return (hipStream_t)s; }
	size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
} // namespace

RIR_EXPORT int rir_codec_layout_query(int width, int height, int nframes, int gop, rir_codec_layout *out)
{
	if (!out || width <= 0 || height <= 0 || nframes <= 0 || gop <= 0)
	{
		log_error("rir_codec_layout_query: invalid argument");
		return -1;
	}
	std::memset(out, 0, sizeof(*out));
	const int64_t npx = (int64_t)width * height;
	out->width = width;
	out->height = height;
	out->nframes = nframes;
	out->gop = gop;
	out->ntiles = (int)((npx + RIRB1_TILE_PX - 1) / RIRB1_TILE_PX);
	out->nchunks = (nframes + gop - 1) / gop;
	const int64_t slots = (int64_t)out->nchunks * out->ntiles;
	out->hdr_bytes = slots * gop * 8;
	out->tile_off_bytes = (int64_t)out->nchunks * (out->ntiles + 1) * 4;
	out->chunk_off_bytes = (int64_t)(out->nchunks + 1) * 8;
	out->stream_max_bytes = slots * gop * RIRB1_REC_MAX_WORDS * 8;
	// workspace = look-back control block (first: it is zeroed by a memset before every launch) + spill slots (padded stride,
	// codec_format.h) + seg_words + chunk_words
	out->workspace_bytes = (int64_t)(align256((size_t)encode_ctrl_bytes(out->nchunks, out->ntiles)) + align256((size_t)(slots * RIRB1_SLOT_WORDS(gop) * 8)) +
									 align256((size_t)slots * 4) + align256((size_t)out->nchunks * 8));
	return 0;
}

namespace
{
	struct Workspace
	{
		uint64_t *ctrl;
		uint64_t *sparse;
		uint32_t *seg_words;
		uint64_t *chunk_words;
	};
	bool carve(const rir_codec_layout &L, void *d_workspace, long long workspace_bytes, Workspace &w)
	{
		if (!d_workspace || workspace_bytes < L.workspace_bytes)
			return false;
		char *ws = static_cast<char *>(d_workspace);
		w.ctrl = reinterpret_cast<uint64_t *>(ws);
		ws += align256((size_t)encode_ctrl_bytes(L.nchunks, L.ntiles));
		w.sparse = reinterpret_cast<uint64_t *>(ws);
		ws += align256((size_t)((int64_t)L.nchunks * L.ntiles * RIRB1_SLOT_WORDS(L.gop) * 8));
		w.seg_words = reinterpret_cast<uint32_t *>(ws);
		ws += align256((size_t)L.nchunks * L.ntiles * 4);
		w.chunk_words = reinterpret_cast<uint64_t *>(ws);
		return true;
	}
	bool check_geometry(const rir_codec_layout &L)
	{
		if ((int64_t)L.ntiles * L.gop * RIRB1_REC_MAX_WORDS > 0xffffffffLL)
		{
			log_error("rir_codec_encode: chunk too large for 32-bit tile offsets");
			return false;
		}
		return true;
	}
} // namespace

// Stage 1 of the encoder: the single pass over the raw frames (headers + sparse payload).
RIR_EXPORT int rir_codec_encode_tiles_device(const unsigned short *d_frames, int width, int height, int nframes, int gop,
											 unsigned long long *d_hdr, void *d_workspace, long long workspace_bytes, void *stream)
{
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	Workspace w;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0 || !check_geometry(L))
		return -1;
	if (!d_frames || !d_hdr || !carve(L, d_workspace, workspace_bytes, w))
	{
		log_error("rir_codec_encode_tiles_device: null buffer or workspace too small");
		return -1;
	}
	return hip_ok(launch_encode_tiles(d_frames, (int64_t)width * height, L.ntiles, nframes, gop, reinterpret_cast<uint64_t *>(d_hdr), w.seg_words,
									  w.sparse, as_stream(stream)),
				  "codec encode (tiles)")
			   ? 0
			   : -1;
}

// Stage 2 of the encoder: offsets + compaction of the sparse payload into the dense stream.
RIR_EXPORT int rir_codec_encode_compact_device(int width, int height, int nframes, int gop, unsigned int *d_tile_off,
											   unsigned long long *d_chunk_off, unsigned long long *d_stream, void *d_workspace,
											   long long workspace_bytes, void *stream)
{
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	Workspace w;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0 || !check_geometry(L))
		return -1;
	if (!d_tile_off || !d_chunk_off || !d_stream || !carve(L, d_workspace, workspace_bytes, w))
	{
		log_error("rir_codec_encode_compact_device: null buffer or workspace too small");
		return -1;
	}
	return hip_ok(launch_encode_compact(L.ntiles, nframes, gop, w.seg_words, w.sparse, d_tile_off, w.chunk_words,
										reinterpret_cast<uint64_t *>(d_chunk_off), reinterpret_cast<uint64_t *>(d_stream), as_stream(stream)),
				  "codec encode (compact)")
			   ? 0
			   : -1;
}

RIR_EXPORT int rir_codec_encode_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
									   unsigned int *d_tile_off, unsigned long long *d_chunk_off, unsigned long long *d_stream,
									   void *d_workspace, long long workspace_bytes, void *stream)
{
	if (rir_codec_encode_tiles_device(d_frames, width, height, nframes, gop, d_hdr, d_workspace, workspace_bytes, stream) != 0)
		return -1;
	return rir_codec_encode_compact_device(width, height, nframes, gop, d_tile_off, d_chunk_off, d_stream, d_workspace, workspace_bytes, stream);
}

// The same encode as ONE kernel that writes the dense stream directly (segments staged in LDS, decoupled look-back for their
// offsets): bit-identical tables and stream, 14 % fewer bytes through HBM than the two passes above, and - on MI355X, on the
// headline workload - no faster (the in-order look-back couples every workgroup to the slowest of its
// predecessors).  The first 32-bit word at d_workspace + 1024 is raised when a look-back gave up (2 s clock): the stream is
// then incomplete; rir_codec_encode_status reads it.
RIR_EXPORT int rir_codec_encode_single_pass_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
												   unsigned int *d_tile_off, unsigned long long *d_chunk_off, unsigned long long *d_stream,
												   void *d_workspace, long long workspace_bytes, void *stream)
{
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	Workspace w;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0 || !check_geometry(L))
		return -1;
	if (!d_frames || !d_hdr || !d_tile_off || !d_chunk_off || !d_stream || !carve(L, d_workspace, workspace_bytes, w))
	{
		log_error("rir_codec_encode_single_pass_device: null buffer or workspace too small");
		return -1;
	}
	return hip_ok(launch_encode_dense(d_frames, (int64_t)width * height, L.ntiles, nframes, gop, reinterpret_cast<uint64_t *>(d_hdr), d_tile_off,
									  reinterpret_cast<uint64_t *>(d_chunk_off), reinterpret_cast<uint64_t *>(d_stream), w.ctrl, w.sparse, as_stream(stream)),
				  "codec encode (single pass)")
			   ? 0
			   : -1;
}

// 0: the last rir_codec_encode_single_pass_device on this workspace completed; 1: a look-back gave up; -1: error.  Waits for `stream`.
RIR_EXPORT int rir_codec_encode_status(const void *d_workspace, void *stream)
{
	if (!device_ready() || !d_workspace)
		return -1;
	unsigned int word = 0;
	if (!hip_ok(hipMemcpyAsync(&word, static_cast<const char *>(d_workspace) + 1024, sizeof(word), hipMemcpyDeviceToHost, as_stream(stream)), "D2H") ||
		!hip_ok(hipStreamSynchronize(as_stream(stream)), "sync"))
		return -1;
	return word != 0 ? 1 : 0;
}

RIR_EXPORT int rir_codec_decode_device(const unsigned long long *d_hdr, const unsigned int *d_tile_off, const unsigned long long *d_chunk_off,
									   const unsigned long long *d_stream, long long stream_words, int width, int height, int nframes,
									   int gop, unsigned short *d_frames, int *d_error, void *stream)
{
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0)
		return -1;
	if (!d_frames || !d_hdr || !d_tile_off || !d_chunk_off || !d_stream || !d_error || stream_words < 0)
	{
		log_error("rir_codec_decode_device: null buffer or negative stream length");
		return -1;
	}
	return hip_ok(launch_decode(reinterpret_cast<const uint64_t *>(d_hdr), d_tile_off, reinterpret_cast<const uint64_t *>(d_chunk_off), reinterpret_cast<const uint64_t *>(d_stream),
								(uint64_t)stream_words, (int64_t)width * height, L.ntiles, nframes, gop, nullptr, 0, d_frames, d_error, as_stream(stream)),
				  "codec decode")
			   ? 0
			   : -1;
}

// Decode of chunks that do not form one contiguous batch (chunks gathered from several shards, a selection of a file's
// chunks): table entry k describes one chunk - hdr[k][ntiles][gop], tile_off[k][ntiles+1], chunk_off[k], chunk_off[k+1] -
// and d_chunk_frames[2k], [2k+1] = (first frame, frame count <= gop; 0 = skip the entry) say where its frames go in
// d_frames, which holds frames_capacity frames.  Same checks as rir_codec_decode_device; an entry that does not fit
// d_frames raises *d_error.
RIR_EXPORT int rir_codec_decode_chunks_device(const unsigned long long *d_hdr, const unsigned int *d_tile_off, const unsigned long long *d_chunk_off,
											  const unsigned long long *d_stream, long long stream_words, int width, int height, int nchunks, int gop,
											  const long long *d_chunk_frames, long long frames_capacity, unsigned short *d_frames, int *d_error,
											  void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_frames || !d_hdr || !d_tile_off || !d_chunk_off || !d_stream || !d_error || !d_chunk_frames || stream_words < 0 || width <= 0 ||
		height <= 0 || nchunks <= 0 || nchunks > 65535 || gop <= 0 || frames_capacity <= 0 || frames_capacity > 0x7fffffffLL)
	{
		log_error("rir_codec_decode_chunks_device: invalid argument");
		return -1;
	}
	const int64_t npx = (int64_t)width * height;
	const int ntiles = (int)((npx + RIRB1_TILE_PX - 1) / RIRB1_TILE_PX);
	return hip_ok(launch_decode(reinterpret_cast<const uint64_t *>(d_hdr), d_tile_off, reinterpret_cast<const uint64_t *>(d_chunk_off),
								reinterpret_cast<const uint64_t *>(d_stream), (uint64_t)stream_words, npx, ntiles, (int)frames_capacity, gop,
								reinterpret_cast<const int64_t *>(d_chunk_frames), nchunks, d_frames, d_error, as_stream(stream)),
				  "codec decode (chunks)")
			   ? 0
			   : -1;
}
