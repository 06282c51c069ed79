// C ABI of the block codec on device-resident batches (format RIRB1).
// Host-pointer / file-level entry points (h264_add_image_lossless, load_image, ...) are in
// video_io_abi.cpp and call these.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// Where the slotted form lives inside the encoder workspace (byte offsets from d_workspace): what rir_codec_encode_tiles_device
// leaves there IS a complete encoded batch - record headers in d_hdr, one length per (chunk, tile) segment, every segment's
// payload at the start of a slot whose place does not depend on any length - that rir_codec_decode_slots_device decodes as it
// is and rir_codec_encode_compact_device turns into the dense (file) form.
RIR_EXPORT int rir_codec_slots_query(int width, int height, int nframes, int gop, rir_codec_slots *out)
{
	rir_codec_layout L;
	if (!out || rir_codec_layout_query(width, height, nframes, gop, &L) != 0)
		return -1;
	Workspace w;
	char *base = reinterpret_cast<char *>((uintptr_t)4096);
	carve(L, base, L.workspace_bytes, w);
	out->slots_offset_bytes = reinterpret_cast<char *>(w.sparse) - base;
	out->slot_words = RIRB1_SLOT_WORDS(gop);
	out->seg_words_offset_bytes = reinterpret_cast<char *>(w.seg_words) - base;
	out->nslots = (int64_t)L.nchunks * L.ntiles;
	return 0;
}

// Decode of the slotted form straight out of the encoder workspace: no offsets, no second encoder pass in front of it.
RIR_EXPORT int rir_codec_decode_slots_device(const unsigned long long *d_hdr, const void *d_workspace, long long workspace_bytes, int width, int height,
											 int nframes, int gop, unsigned short *d_frames, int *d_error, void *stream)
{
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	Workspace w;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0 || !check_geometry(L))
		return -1;
	if (!d_hdr || !d_frames || !d_error || !carve(L, const_cast<void *>(d_workspace), workspace_bytes, w))
	{
		log_error("rir_codec_decode_slots_device: null buffer or workspace too small");
		return -1;
	}
	return hip_ok(launch_decode_slots(reinterpret_cast<const uint64_t *>(d_hdr), w.seg_words, w.sparse, (int64_t)width * height, L.ntiles, nframes, gop,
									  d_frames, d_error, as_stream(stream)),
				  "codec decode (slots)")
			   ? 0
			   : -1;
}

// ---- workspace placement -------------------------------------------------------------------------------------------------
// The packing kernel reads the raw frames and writes the slots at the same pace; on MI355X its time has two values, ~135 and
// ~150 us for 1 000 x 640x512, and which one it is depends on nothing but the PAIR of device allocations the two regions live
// in: allocations fall into a few "placement classes" (stretches of the address space handed out by the driver; one
// allocation, however large, is of one class), the kernel is slow exactly when frames and workspace are of the same class,
// whatever the offsets inside the allocations and whatever the cache policy of its loads and stores
// (profiles/r03_placement/README.md: the class matrix over ten allocations, the offset sweeps, the policy variants, the
// counters of the two cases) - the signature of read / write turn-arounds inside one rank of the HBM stacks.  A virtual
// address does not tell the class, so the library finds a workspace of another class than the caller's frames the only way
// there is: it allocates candidates itself (each `spacing_bytes` further along, the spacers freed before it returns), times
// the packing kernel on each with HIP events and keeps the first that is 7 % faster than the first one, else the fastest.
// One-off set-up cost of a few milliseconds; nothing of the caller's is touched, no allocator cache is flushed.
// times_us: HOST float[max_tries + 1] or NULL, *ntimes entries filled (the kept candidate first).  The workspace is released
// with rir_codec_workspace_destroy_device.  Returns 0, or -1 (nothing allocated).
RIR_EXPORT int rir_codec_workspace_create_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, int max_tries,
												 long long spacing_bytes, void **d_workspace, float *times_us, int *ntimes, void *stream)
{
	if (ntimes)
		*ntimes = 0;
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0 || !check_geometry(L))
		return -1;
	if (!d_frames || !d_workspace || max_tries < 0 || max_tries > 64 || spacing_bytes < 0)
	{
		log_error("rir_codec_workspace_create_device: invalid argument");
		return -1;
	}
	*d_workspace = nullptr;
	hipStream_t st = as_stream(stream);
	DeviceBuffer hdr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (!hdr.reserve((size_t)L.hdr_bytes) || !hip_ok(hipEventCreate(&e0), "hipEventCreate") || !hip_ok(hipEventCreate(&e1), "hipEventCreate"))
	{
		if (e0)
			(void)hipEventDestroy(e0);
		return -1;
	}
	constexpr int kReps = 5;
	auto time_on = [&](void *ws, float &us) {
		float best[kReps];
		if (rir_codec_encode_tiles_device(d_frames, width, height, nframes, gop, hdr.as<unsigned long long>(), ws, L.workspace_bytes, stream) != 0)
			return false;
		for (int r = 0; r < kReps; ++r)
		{
			if (!hip_ok(hipEventRecord(e0, st), "event") ||
				rir_codec_encode_tiles_device(d_frames, width, height, nframes, gop, hdr.as<unsigned long long>(), ws, L.workspace_bytes, stream) != 0 ||
				!hip_ok(hipEventRecord(e1, st), "event") || !hip_ok(hipEventSynchronize(e1), "sync") || !hip_ok(hipEventElapsedTime(&best[r], e0, e1), "elapsed"))
				return false;
		}
		std::sort(best, best + kReps);
		us = best[kReps / 2] * 1e3f;
		return true;
	};
	std::vector<void *> spacers, cands;
	std::vector<float> times;
	bool ok = true;
	for (int t = 0; t <= max_tries && ok; ++t)
	{
		void *sp = nullptr, *ws = nullptr;
		if (t > 0 && spacing_bytes > 0)
		{
			if (hipMalloc(&sp, (size_t)spacing_bytes) != hipSuccess)
			{ // (no room for another candidate: the best so far is kept)
				(void)hipGetLastError();
				break;
			}
			spacers.push_back(sp);
		}
		if (hipMalloc(&ws, (size_t)L.workspace_bytes) != hipSuccess)
		{
			(void)hipGetLastError();
			if (t == 0)
			{
				log_error("rir_codec_workspace_create_device: out of device memory");
				ok = false;
			}
			break;
		}
		float us = 0;
		if (!time_on(ws, us))
		{
			(void)hipFree(ws);
			ok = false;
			break;
		}
		cands.push_back(ws);
		times.push_back(us);
		if (t > 0 && us < 0.93f * times[0])
			break;
	}
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	for (void *sp : spacers)
		(void)hipFree(sp);
	size_t keep = 0;
	for (size_t i = 1; i < cands.size(); ++i)
		if (times[i] < times[keep])
			keep = i;
	for (size_t i = 0; i < cands.size(); ++i)
		if (!ok || i != keep)
			(void)hipFree(cands[i]);
	if (!ok || cands.empty())
		return -1;
	*d_workspace = cands[keep];
	if (times_us)
	{
		times_us[0] = times[keep];
		int k = 1;
		for (size_t i = 0; i < times.size(); ++i)
			if (i != keep)
				times_us[k++] = times[i];
	}
	if (ntimes)
		*ntimes = (int)times.size();
	return 0;
}

// The same for ANY pair of buffers a kernel walks at the same pace (the frame-buffer kernels: translate 92-99 us against 86-88, gaussian 119
// against 107, the fused chain 119-129 against 116-117 per 256 frames 640x512, tests/perf/filter_class_probe.py): `bytes` of device memory in
// another placement class than the buffer d_other (other_bytes long), found by timing a plain streaming copy from d_other into each candidate
// (up to 256 MiB of it).  Candidates `spacing_bytes` apart, at most max_tries further ones; the first that copies 5 % faster than the first
// one is kept, else the fastest; spacers and losers are freed before the call returns.  Release with rir_buffer_destroy_device.
RIR_EXPORT int rir_buffer_create_beside_device(const void *d_other, long long other_bytes, long long bytes, int max_tries, long long spacing_bytes,
											   void **d_buffer, float *times_us, int *ntimes, void *stream)
{
	if (ntimes)
		*ntimes = 0;
	if (!device_ready())
		return -1;
	if (!d_other || !d_buffer || other_bytes < 16 || bytes < 16 || max_tries < 0 || max_tries > 64 || spacing_bytes < 0 || ((uintptr_t)d_other & 15))
	{
		log_error("rir_buffer_create_beside_device: invalid argument");
		return -1;
	}
	*d_buffer = nullptr;
	hipStream_t st = as_stream(stream);
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (!hip_ok(hipEventCreate(&e0), "hipEventCreate") || !hip_ok(hipEventCreate(&e1), "hipEventCreate"))
	{
		if (e0)
			(void)hipEventDestroy(e0);
		return -1;
	}
	const int64_t probe = (std::min<long long>(std::min<long long>(other_bytes, bytes), 256ll << 20)) & ~15ll;
	constexpr int kReps = 5;
	auto time_on = [&](void *cand, float &us) {
		float t[kReps];
		if (!hip_ok(launch_stream_copy_probe(d_other, cand, probe, st), "copy probe"))
			return false;
		for (int r = 0; r < kReps; ++r)
			if (!hip_ok(hipEventRecord(e0, st), "event") || !hip_ok(launch_stream_copy_probe(d_other, cand, probe, st), "copy probe") ||
				!hip_ok(hipEventRecord(e1, st), "event") || !hip_ok(hipEventSynchronize(e1), "sync") || !hip_ok(hipEventElapsedTime(&t[r], e0, e1), "elapsed"))
				return false;
		std::sort(t, t + kReps);
		us = t[kReps / 2] * 1e3f;
		return true;
	};
	std::vector<void *> spacers, cands;
	std::vector<float> times;
	bool ok = true;
	for (int k = 0; k <= max_tries && ok; ++k)
	{
		void *sp = nullptr, *b = nullptr;
		if (k > 0 && spacing_bytes > 0)
		{
			if (hipMalloc(&sp, (size_t)spacing_bytes) != hipSuccess)
			{
				(void)hipGetLastError();
				break;
			}
			spacers.push_back(sp);
		}
		if (hipMalloc(&b, (size_t)bytes) != hipSuccess)
		{
			(void)hipGetLastError();
			if (k == 0)
			{
				log_error("rir_buffer_create_beside_device: out of device memory");
				ok = false;
			}
			break;
		}
		float us = 0;
		if (!time_on(b, us))
		{
			(void)hipFree(b);
			ok = false;
			break;
		}
		cands.push_back(b);
		times.push_back(us);
		if (k > 0 && us < 0.95f * times[0])
			break;
	}
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	for (void *sp : spacers)
		(void)hipFree(sp);
	size_t keep = 0;
	for (size_t i = 1; i < cands.size(); ++i)
		if (times[i] < times[keep])
			keep = i;
	for (size_t i = 0; i < cands.size(); ++i)
		if (!ok || i != keep)
			(void)hipFree(cands[i]);
	if (!ok || cands.empty())
		return -1;
	*d_buffer = cands[keep];
	if (times_us)
	{
		times_us[0] = times[keep];
		int k = 1;
		for (size_t i = 0; i < times.size(); ++i)
			if (i != keep)
				times_us[k++] = times[i];
	}
	if (ntimes)
		*ntimes = (int)times.size();
	return 0;
}
RIR_EXPORT void rir_buffer_destroy_device(void *d_buffer)
{
	if (d_buffer)
		(void)hipFree(d_buffer);
}

RIR_EXPORT void rir_codec_workspace_destroy_device(void *d_workspace)
{
	if (d_workspace)
		(void)hipFree(d_workspace);
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

// ---- the packed form (include/rir_amd_device.h; kernel: rirb1_encode_packed) ---------------------------------------------
RIR_EXPORT int rir_codec_packed_query(int width, int height, int nframes, int gop, rir_codec_packed_layout *out)
{
	rir_codec_layout L;
	if (!out || rir_codec_layout_query(width, height, nframes, gop, &L) != 0)
		return -1;
	std::memset(out, 0, sizeof(*out));
	const int64_t slots = (int64_t)L.nchunks * L.ntiles;
	out->ntiles = L.ntiles, out->nchunks = L.nchunks;
	out->hdr_bytes = L.hdr_bytes;
	out->seg_pos_bytes = slots * 8;
	out->seg_words_bytes = slots * 4;
	out->stream_budget_bytes = (int64_t)align256((size_t)((int64_t)width * height * nframes)); // 8 bits per pixel
	out->stream_max_bytes = L.stream_max_bytes;
	// the arena: a spilling wave takes the worst case of its share of a chunk; min = room for one workgroup in sixteen, max = for all
	const int64_t share_words = ((int64_t)(gop + 4) / 4 + 1) * RIRB1_REC_MAX_WORDS;
	out->workspace_max_bytes = RIRB1_PACKED_CTRL_BYTES + (int64_t)align256((size_t)(slots * 4 * share_words * 8));
	out->workspace_min_bytes = RIRB1_PACKED_CTRL_BYTES + (int64_t)align256((size_t)(((slots + 15) / 16) * 4 * share_words * 8));
	return 0;
}

namespace
{
	int encode_packed(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr, unsigned long long *d_seg_pos,
					  unsigned int *d_seg_words, unsigned long long *d_stream, long long stream_capacity_words, void *d_workspace, long long workspace_bytes,
					  bool reset, void *stream);
}
RIR_EXPORT int rir_codec_encode_packed_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
											  unsigned long long *d_seg_pos, unsigned int *d_seg_words, unsigned long long *d_stream,
											  long long stream_capacity_words, void *d_workspace, long long workspace_bytes, void *stream)
{
	return encode_packed(d_frames, width, height, nframes, gop, d_hdr, d_seg_pos, d_seg_words, d_stream, stream_capacity_words, d_workspace, workspace_bytes, true, stream);
}
// The two halves of the call above, for a caller that wants them apart (bench.py's per-kernel HIP events: the reset is a fill launch of its own):
// rir_codec_packed_reset_device zeroes the control block of a workspace, rir_codec_encode_packed_launch_device packs into a workspace that has
// just been reset on the same stream.
RIR_EXPORT int rir_codec_packed_reset_device(void *d_workspace, long long workspace_bytes, void *stream)
{
	if (!device_ready())
		return -1;
	if (!d_workspace || workspace_bytes < RIRB1_PACKED_CTRL_BYTES)
	{
		log_error("rir_codec_packed_reset_device: workspace too small");
		return -1;
	}
	return hip_ok(hipMemsetAsync(d_workspace, 0, RIRB1_PACKED_CTRL_BYTES, as_stream(stream)), "memset") ? 0 : -1;
}
RIR_EXPORT int rir_codec_encode_packed_launch_device(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr,
													 unsigned long long *d_seg_pos, unsigned int *d_seg_words, unsigned long long *d_stream,
													 long long stream_capacity_words, void *d_workspace, long long workspace_bytes, void *stream)
{
	return encode_packed(d_frames, width, height, nframes, gop, d_hdr, d_seg_pos, d_seg_words, d_stream, stream_capacity_words, d_workspace, workspace_bytes, false, stream);
}
namespace
{
int encode_packed(const unsigned short *d_frames, int width, int height, int nframes, int gop, unsigned long long *d_hdr, unsigned long long *d_seg_pos,
				  unsigned int *d_seg_words, unsigned long long *d_stream, long long stream_capacity_words, void *d_workspace, long long workspace_bytes,
				  bool reset, void *stream)
{
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0 || !check_geometry(L))
		return -1;
	if (!d_frames || !d_hdr || !d_seg_pos || !d_seg_words || !d_stream || stream_capacity_words < 0 || !d_workspace ||
		workspace_bytes < RIRB1_PACKED_CTRL_BYTES || ((uintptr_t)d_workspace & 127))
	{
		log_error("rir_codec_encode_packed_device: null buffer, negative capacity, or a workspace that is too small or not 128-byte aligned");
		return -1;
	}
	uint64_t *ctrl = static_cast<uint64_t *>(d_workspace);
	uint64_t *arena = ctrl + RIRB1_PACKED_CTRL_BYTES / 8;
	const uint64_t arena_words = (uint64_t)(workspace_bytes - RIRB1_PACKED_CTRL_BYTES) / 8;
	return hip_ok(launch_encode_packed(d_frames, (int64_t)width * height, L.ntiles, nframes, gop, reinterpret_cast<uint64_t *>(d_hdr),
									   reinterpret_cast<uint64_t *>(d_seg_pos), d_seg_words, reinterpret_cast<uint64_t *>(d_stream),
									   (uint64_t)stream_capacity_words, ctrl, arena, arena_words, reset, as_stream(stream)),
				  "codec encode (packed)")
			   ? 0
			   : -1;
}
} // namespace

RIR_EXPORT int rir_codec_encode_packed_status(const void *d_workspace, unsigned long long *out3, void *stream)
{
	if (!device_ready() || !d_workspace)
		return -1;
	unsigned long long c[RIRB1_PACKED_CTRL_BYTES / 8];
	if (!hip_ok(hipMemcpyAsync(c, d_workspace, sizeof(c), hipMemcpyDeviceToHost, as_stream(stream)), "D2H") ||
		!hip_ok(hipStreamSynchronize(as_stream(stream)), "sync"))
		return -1;
	if (out3)
		out3[0] = c[0], out3[1] = c[16], out3[2] = c[32];
	return (int)(c[48] & 3u) | (c[0] + c[16] > c[64] ? 1 : 0);
}

RIR_EXPORT int rir_codec_decode_packed_device(const unsigned long long *d_hdr, const unsigned long long *d_seg_pos, const unsigned int *d_seg_words,
											  const unsigned long long *d_stream, long long stream_words, int width, int height, int nframes, int gop,
											  unsigned short *d_frames, int *d_error, void *stream)
{
	if (!device_ready())
		return -1;
	rir_codec_layout L;
	if (rir_codec_layout_query(width, height, nframes, gop, &L) != 0)
		return -1;
	if (!d_frames || !d_hdr || !d_seg_pos || !d_seg_words || (!d_stream && stream_words != 0) || !d_error || stream_words < 0)
	{
		log_error("rir_codec_decode_packed_device: null buffer or negative stream length");
		return -1;
	}
	return hip_ok(launch_decode_packed(reinterpret_cast<const uint64_t *>(d_hdr), reinterpret_cast<const uint64_t *>(d_seg_pos), d_seg_words,
									   reinterpret_cast<const uint64_t *>(d_stream), (uint64_t)stream_words, (int64_t)width * height, L.ntiles, nframes, gop,
									   d_frames, d_error, as_stream(stream)),
				  "codec decode (packed)")
			   ? 0
			   : -1;
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
