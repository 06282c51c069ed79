// C ABI of the video_io half of the hot path: saver (h264_* / open_video_write), loader
// (open_camera_* / load_image / attributes) and the read-back filters, on top of the device
// codec (codec_abi.cpp).  Same symbol names, arguments and return codes as the reference header
// src/cpp/video_io/video_io.h (behaviour of src/cpp/video_io/video_io.cpp cited per function).
//
// Container written by the saver ("RIRB" file): an ISO-BMFF style `ftyp` box (so
// that format sniffers that look for "ftyp" at byte 4 - reference IRFileLoader.cpp:118-123 - file
// it with the MP4/H.264 family, which is what FILE_FORMAT_H264 means to callers), a fixed header,
// one record per chunk (GOP) holding the RIRB1 tables and payload, a chunk index, and the
// reference's own "H264ATTRIBUTES" metadata trailer as the last bytes of the file.
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <sstream>

#include "codec_format.h"
#include "file_attributes.h"
#include "filter_kernels.h"
#include "lossy_kernels.h"
#include "rir_amd_device.h"
#include "runtime.h"

using namespace rir;

// Nothing may be thrown across the C boundary: entry points that allocate run through this.
template <class F>
static int guarded(const char *what, int on_error, F &&body)
{
	try
	{
		return body();
	}
	catch (const std::exception &e)
	{
		log_error(std::string(what) + ": " + e.what());
		return on_error;
	}
}

extern "C"
{
	int rir_bad_pixels_create_device(const unsigned short *, int, int, void *);
	int rir_bad_pixels_correct_device(int, const unsigned short *, unsigned short *, int, void *);
	int rir_bad_pixels_create_rows_device(const unsigned short *, int, int, int, void *);
	int rir_remove_bad_pixels_device(int, unsigned short *, int, int, void *);
	int rir_remove_motion_device(const unsigned short *, unsigned short *, int, int, int, int, const float *, void *);
	void bad_pixels_destroy(int);
}

#define FILE_FORMAT_PCR 1
#define FILE_FORMAT_WEST 2
#define FILE_FORMAT_PCR_ENCAPSULATED 3
#define FILE_FORMAT_ZSTD_COMPRESSED 4
#define FILE_FORMAT_H264 5
#define FILE_FORMAT_HCC 6
#define FILE_FORMAT_OTHER 7
#define UNSPECIFIED_CHAR_LENGTH 200

namespace
{
	// ---- container ---------------------------------------------------------------------------
#pragma pack(push, 1)
	struct FtypBox
	{
		uint8_t size_be[4]; // 32
		char type[4];		// "ftyp"
		char major[4];		// "RIRB"
		uint8_t minor_be[4];
		char compat[16]; // "RIRBisom" + padding
	};
	struct FileHeader
	{
		char magic[8]; // "RIRBLOCK"
		uint32_t version;
		uint32_t width, height, gop, fps, flags;
		uint64_t index_offset; // 0 until the file is closed
		uint64_t nframes, nchunks;
		uint8_t reserved[8];
	};
	struct ChunkHeader
	{
		char magic[4]; // "CHNK"
		uint32_t nframes, ntiles, gop;
		uint64_t payload_words;
		uint64_t first_frame;
	};
	struct IndexEntry
	{
		uint64_t file_offset; // of the ChunkHeader
		uint64_t first_frame;
		uint32_t nframes;
		uint32_t reserved;
	};
	struct PcrHeader
	{ // reference IRFileLoader.h:43-61
		int32_t Version, NbImages, X, Y, Band, Bits, Interlaced, Frequency, ImagesPerBuffer, TransfertSize, GrabSizeX, GrabSizeY;
		char reserved[1024 - 48];
	};
	// ZFile container of the reference (ZFile.cpp:18-46): two 128-byte blocks, then one record per image
	//   [int64 timestamp][u32 csize][csize bytes: ZSTD_compress(frame)]      (compression == 1)
	// and the metadata trailer with the global attribute "positions" (ZFile.cpp:434-447).
	struct ZHeader
	{
		uint8_t version, triggers, compression;
		char reserved[125];
	};
	struct ZTrigger
	{
		uint64_t date, rate, samples, samples_pre_trigger, type, nb_channels, data_type, data_format, data_repetition, data_size_x, data_size_y;
		char reserved[128 - 88];
	};
#pragma pack(pop)
	static_assert(sizeof(ZHeader) == 128 && sizeof(ZTrigger) == 128, "layout");
	static_assert(sizeof(FtypBox) == 32 && sizeof(FileHeader) == 64 && sizeof(ChunkHeader) == 32 && sizeof(PcrHeader) == 1024, "layout");

	// ZFile: zstd works on one thread per image (0.25-0.5 ms for a 640x512 image either way).  A loader read image after image and a writer
	// decompress / compress several images at once on threads of their own; RIR_ZFILE_THREADS: how many (default 8, 0: every image in the
	// call that brings or asks for it).
	int zfile_threads()
	{
		static const int n = [] {
			const char *e = std::getenv("RIR_ZFILE_THREADS");
			return e ? std::max(0, std::min(16, std::atoi(e))) : 8;
		}();
		return n;
	}

	bool file_exists(const char *name)
	{
		struct stat st;
		return stat(name, &st) == 0;
	}

	// Device + pinned staging for one chunk geometry.
	struct ChunkCodec
	{
		int w = 0, h = 0, gop = 0;
		rir_codec_layout L{};
		DeviceBuffer d_frames, d_hdr, d_tile_off, d_chunk_off, d_stream, d_ws, d_err, d_tmp, d_shift;
		PinnedBuffer h_frames, h_in; // h_in: a chunk's tables and payload as read from the file (loader)
		bool on_device = true;		 // loader: the decoded chunk is in d_frames (false: it was decoded straight into h_frames)
		bool prepare(int w_, int h_, int gop_, bool encoder = true)
		{ // encoder: the encode workspace is needed too (the loader only decodes)
			if (w == w_ && h == h_ && gop == gop_ && d_frames.ptr)
				return true;
			if (rir_codec_layout_query(w_, h_, gop_, gop_, &L) != 0)
				return false;
			w = w_, h = h_, gop = gop_;
			const size_t fb = (size_t)w * h * 2 * gop;
			return d_frames.reserve(fb) && d_hdr.reserve((size_t)L.hdr_bytes) && d_tile_off.reserve((size_t)L.tile_off_bytes) &&
				   d_chunk_off.reserve((size_t)L.chunk_off_bytes) && d_stream.reserve((size_t)L.stream_max_bytes) &&
				   (!encoder || d_ws.reserve((size_t)L.workspace_bytes)) && d_err.reserve(sizeof(int)) && h_frames.reserve(fb);
		}
		// the buffers change owner (the structs hold raw pointers and free them when they die: member-wise, never by copy)
		void swap_with(ChunkCodec &o)
		{
			std::swap(w, o.w), std::swap(h, o.h), std::swap(gop, o.gop), std::swap(L, o.L), std::swap(on_device, o.on_device);
			DeviceBuffer *a[] = {&d_frames, &d_hdr, &d_tile_off, &d_chunk_off, &d_stream, &d_ws, &d_err, &d_tmp, &d_shift};
			DeviceBuffer *b[] = {&o.d_frames, &o.d_hdr, &o.d_tile_off, &o.d_chunk_off, &o.d_stream, &o.d_ws, &o.d_err, &o.d_tmp, &o.d_shift};
			for (int i = 0; i < 9; ++i)
				std::swap(a[i]->ptr, b[i]->ptr), std::swap(a[i]->cap, b[i]->cap);
			std::swap(h_frames.ptr, o.h_frames.ptr), std::swap(h_frames.cap, o.h_frames.cap);
			std::swap(h_in.ptr, o.h_in.ptr), std::swap(h_in.cap, o.h_in.cap);
		}
	};

	// ---- bounded-loss step ---------------------------------------------------------------------
	// Host side of H264_Saver::addImageLossyNoCamera / addLoss (h264.cpp:2253-2424, :2426-2607): the
	// per-pixel work runs in lossy_kernels.hip; the error budget (40-frame window of the quirky
	// "stdDev" statistics, h264.cpp:2335-2385) is scalar double arithmetic done here from the exact
	// integer sums the GPU returns.
	struct LossyState
	{
		int w = 0, h = 0, hl = 0;
		int frames = 0;
		DeviceBuffer d_img, d_tmp, d_out, d_ref, d_prev, d_last, d_sums, d_cval, d_ccnt, d_ring, d_hist, d_stats, d_min, d_budget, d_decision, d_errs, d_tickets;
		LossyDeviceState dev{};
		DeviceBuffer d_shadow;		   // the speculative form of a run (lossy_kernels.h: LossySpec): where a pass leaves the state after the group
		LossyDeviceState shadow{};	   // ... carved out of it (reserve_shadow)
		int bp_handle = 0;

		~LossyState()
		{
			if (bp_handle > 0)
				bad_pixels_destroy(bp_handle);
		}

		bool prepare(int w_, int h_, int hl_, int running_average, bool subtract_min)
		{
			w = w_, h = h_, hl = std::max(0, std::min(hl_, h_));
			const size_t full = (size_t)w * h, s = (size_t)w * hl;
			const int ra = std::max(0, std::min(running_average, 64));
			if (!d_img.reserve(full * 2) || !d_tmp.reserve(full * 2) || !d_out.reserve(full * 2) || !d_ref.reserve(full * 2) ||
				!d_prev.reserve(full * 2) || !d_last.reserve(full * 2) || !d_sums.reserve(s * 4 + 4) || !d_cval.reserve(s * 2 + 4) ||
				!d_ccnt.reserve(s * 2 + 4) || !d_ring.reserve((size_t)std::max(ra, 1) * s * 2 + 4) || !d_hist.reserve(16384 * 4) ||
				!d_stats.reserve(8 * sizeof(long long)) || !d_min.reserve(4) || !d_budget.reserve(sizeof(LossyBudget)) ||
				!d_decision.reserve(sizeof(LossyDecision)) || !d_errs.reserve(2 * sizeof(int)) || !d_tickets.reserve(2 * sizeof(unsigned int)))
				return false;
			hipStream_t st = default_stream();
			// histogram, sums and budget start at zero; afterwards every frame leaves the first two cleared (lossy_kernels.hip)
			if (!hip_ok(hipMemsetAsync(d_sums.ptr, 0, s * 4 + 4, st), "memset") || !hip_ok(hipMemsetAsync(d_cval.ptr, 0, s * 2 + 4, st), "memset") ||
				!hip_ok(hipMemsetAsync(d_ccnt.ptr, 0, s * 2 + 4, st), "memset") || !hip_ok(hipMemsetAsync(d_hist.ptr, 0, 16384 * 4, st), "memset") ||
				!hip_ok(hipMemsetAsync(d_stats.ptr, 0, 8 * sizeof(long long), st), "memset") ||
				!hip_ok(hipMemsetAsync(d_budget.ptr, 0, sizeof(LossyBudget), st), "memset") ||
				!hip_ok(hipMemsetAsync(d_tickets.ptr, 0, 2 * sizeof(unsigned int), st), "memset"))
				return false;
			dev.refT = d_ref.as<uint16_t>(), dev.prevT = d_prev.as<uint16_t>(), dev.lastDL = d_last.as<uint16_t>();
			dev.ra_sums = d_sums.as<uint32_t>(), dev.ra_const_value = d_cval.as<uint16_t>(), dev.ra_const_count = d_ccnt.as<int16_t>();
			dev.ra_images = d_ring.as<uint16_t>();
			dev.ra_count = 0, dev.ra_head = 0, dev.running_average = ra;
			dev.subtract_min = subtract_min ? 1 : 0;
			dev.min = 0;
			frames = 0;
			return hip_ok(wait_stream(st), "sync"); // the state is ready whatever stream the steps run on
		}

		// img (host, full frame) -> d_out (device, full frame), copied to `out` (host) when it is not NULL.
		bool step(const unsigned short *img, unsigned short *out, bool add_loss, bool remove_bad_pixels, int low_value_error, int high_value_error,
				  double std_factor, int &low_error, int &high_error)
		{
			hipStream_t st = default_stream();
			const int full = w * h;
			int e[2] = {0, 0};
			if (!hip_ok(hipMemcpyAsync(d_img.ptr, img, (size_t)full * 2, hipMemcpyHostToDevice, st), "H2D") ||
				!queue_frame(d_img.as<uint16_t>(), d_out.as<uint16_t>(), add_loss, remove_bad_pixels, low_value_error, high_value_error, std_factor,
							 d_errs.as<int>(), st) ||
				!hip_ok(hipMemcpyAsync(e, d_errs.ptr, sizeof(e), hipMemcpyDeviceToHost, st), "D2H"))
				return false;
			// (when out is NULL the caller consumes d_out on the same stream)
			if (out && !hip_ok(hipMemcpyAsync(out, d_out.ptr, (size_t)full * 2, hipMemcpyDeviceToHost, st), "D2H"))
				return false;
			if (!hip_ok(wait_stream(st), "sync"))
				return false;
			low_error = e[0], high_error = e[1];
			return true;
		}

		// pinned_img (PAGE-LOCKED host memory that stays untouched until the stream has passed this point) -> d_out (device),
		// queued without waiting; the frame's low / high error goes to d_errors (device int[2]).
		bool queue_host_frame(const unsigned short *pinned_img, bool add_loss, bool remove_bad_pixels, int low_value_error, int high_value_error,
							  double std_factor, int *d_errors)
		{
			hipStream_t st = default_stream();
			return hip_ok(hipMemcpyAsync(d_img.ptr, pinned_img, (size_t)w * h * 2, hipMemcpyHostToDevice, st), "H2D") &&
				   queue_frame(d_img.as<uint16_t>(), d_out.as<uint16_t>(), add_loss, remove_bad_pixels, low_value_error, high_value_error, std_factor,
							   d_errors, st);
		}

		// One frame, device to device, and its budget back on the host (one 8-byte read-back and a synchronisation).
		bool step_device(const uint16_t *d_src, uint16_t *d_dst, bool add_loss, bool remove_bad_pixels, int low_value_error, int high_value_error,
						 double std_factor, int &low_error, int &high_error, hipStream_t st)
		{
			int e[2] = {0, 0};
			if (!queue_frame(d_src, d_dst, add_loss, remove_bad_pixels, low_value_error, high_value_error, std_factor, d_errs.as<int>(), st) ||
				!hip_ok(hipMemcpyAsync(e, d_errs.ptr, sizeof(e), hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
				return false;
			low_error = e[0], high_error = e[1];
			return true;
		}

		// One frame, device to device (d_src and d_dst: full frames, distinct), queued on `st` without waiting: statistics,
		// error budget (lossy_budget_kernel) and update all run on the device.  d_errors: device int[2] that receives the
		// frame's low / high error (may be NULL).  Only the very first frame of a subtractMin stream waits (its minimum
		// becomes a parameter of every later launch).
		bool queue_frame(const uint16_t *d_src, uint16_t *d_dst, bool add_loss, bool remove_bad_pixels, int low_value_error, int high_value_error,
						 double std_factor, int *d_errors, hipStream_t st)
		{
			const int full = w * h, s = w * hl;
			const uint16_t *tmp = d_src; // without bad-pixel repair the frame is used as it is
			if (remove_bad_pixels && hl > 0)
			{ // bp.init on the first image (rows < lossy_height), bp.correct on every image (h264.cpp:2259-2266)
				if (frames == 0 && bp_handle <= 0)
					bp_handle = rir_bad_pixels_create_device(d_src, w, hl, st);
				uint16_t *fixed = d_tmp.as<uint16_t>();
				if (bp_handle <= 0 || rir_bad_pixels_correct_device(bp_handle, d_src, fixed, 1, st) != 0)
					return false;
				if (full > s && !hip_ok(hipMemcpyAsync(fixed + s, d_src + s, (size_t)(full - s) * 2, hipMemcpyDeviceToDevice, st), "D2D"))
					return false;
				tmp = fixed;
			}
			if (frames == 0)
			{
				if (dev.subtract_min && s > 0)
				{
					unsigned int mn = 65535;
					if (!hip_ok(launch_lossy_min(tmp, s, d_min.as<unsigned int>(), st), "lossy min") ||
						!hip_ok(hipMemcpyAsync(&mn, d_min.ptr, 4, hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
						return false;
					dev.min = mn;
				}
				if (!hip_ok(launch_lossy_first(tmp, d_dst, dev, s, full, st), "lossy first"))
					return false;
				// the first image is stored as it is: its errors are the configured ones (h264.cpp:2290-2333)
				if (d_errors && (!hip_ok(hipMemsetD32Async((hipDeviceptr_t)d_errors, low_value_error, 1, st), "errors") ||
								 !hip_ok(hipMemsetD32Async((hipDeviceptr_t)(d_errors + 1), high_value_error, 1, st), "errors")))
					return false;
			}
			else
			{
				const LossyStep step = make_step(tmp, d_src, d_dst, add_loss, low_value_error, high_value_error, std_factor, d_errors);
				if (!hip_ok(launch_lossy_step(&step, nullptr, 1, st), "lossy step"))
					return false;
				advance_ring();
			}
			++frames;
			return true;
		}

		// what the kernels of the next frame need (the ring indices as they are now)
		LossyStep make_step(const uint16_t *tmp, const uint16_t *img, uint16_t *dst, bool add_loss, int low_value_error, int high_value_error,
							double std_factor, int *d_errors, int nstreams = 1)
		{
			LossyStep p;
			p.hist_px = lossy_hist_px(w * hl, nstreams), p.reserved = 0;
			p.tmp = tmp, p.img = img, p.out = dst;
			p.st = dev;
			p.hist = d_hist.as<uint32_t>(), p.stats = d_stats.as<long long>();
			p.budget = d_budget.as<LossyBudget>(), p.decision = d_decision.as<LossyDecision>();
			p.errors_out = d_errors;
			p.tickets = d_tickets.as<unsigned int>();
			p.s = w * hl, p.full = w * h;
			p.add_loss = add_loss ? 1 : 0, p.low_value_error = low_value_error, p.high_value_error = high_value_error;
			p.std_factor = std_factor;
			p.next_tmp = p.next_img = nullptr, p.next_background = nullptr, p.next_errors_out = nullptr;
			p.do_update = 1, p.reserved2 = 0;
			return p;
		}
		// the shadow arrays of the speculative form, allocated when a stream is first offered to it: as the state's own, ring included
		bool reserve_shadow()
		{
			if (shadow.refT)
				return true;
			const size_t full = (size_t)w * h, s = (size_t)w * hl;
			auto up = [](size_t v) { return (v + 255) / 256 * 256; };
			const size_t ring = (size_t)std::max(dev.running_average, 1) * s * 2 + 16;
			const size_t total = 3 * up(full * 2 + 16) + up(s * 4 + 16) + 2 * up(s * 2 + 16) + up(ring);
			if (!d_shadow.reserve(total))
				return false;
			char *p = d_shadow.as<char>();
			shadow = dev;
			shadow.refT = reinterpret_cast<uint16_t *>(p), p += up(full * 2 + 16);
			shadow.prevT = reinterpret_cast<uint16_t *>(p), p += up(full * 2 + 16);
			shadow.lastDL = reinterpret_cast<uint16_t *>(p), p += up(full * 2 + 16);
			shadow.ra_sums = reinterpret_cast<uint32_t *>(p), p += up(s * 4 + 16);
			shadow.ra_const_value = reinterpret_cast<uint16_t *>(p), p += up(s * 2 + 16);
			shadow.ra_const_count = reinterpret_cast<int16_t *>(p), p += up(s * 2 + 16);
			shadow.ra_images = reinterpret_cast<uint16_t *>(p);
			return true;
		}
		void advance_ring()
		{
			if (dev.running_average > 0)
			{
				if (dev.ra_count == dev.running_average)
					dev.ra_head = (dev.ra_head + 1) % dev.running_average;
				else
					++dev.ra_count;
			}
		}
	};

	// handle for the device-resident form of the bounded-loss step (rir_lossy_*)
	struct LossyObject : public Object
	{
		const char *type_name() const override { return "LossyStream"; }
		LossyState st;
		int low = 6, high = 2;
		double std_factor = 5;
		bool remove_bad_pixels = false;
		DeviceBuffer batch_errs; // int[nframes][2] of the last rir_lossy_step_device call
		DeviceBuffer multi_table; // rir_lossy_step_multi_device: the steps of the call (this object leads it)
		DeviceBuffer run_exchange; // the run kernel's ticket, error word and exchange words
		DeviceBuffer run_hist, run_tickets, run_bg; // runs of frames: histogram slices and tickets of a group of frames, backgrounds of the call
	DeviceBuffer const_ok, const_partials;		// constant-budget form: one word per group of the last call (1 = stepped by it), the tail frames' sums
	int const_groups = 0;						// groups of the last run call that were OFFERED to the constant-budget form (0: it was not eligible)
	// speculative form (lossy_kernels.h: LossySpec): the budget tables, sums and statistics of a group (reused group after group), the control
	// words of every group and stream of the last call, and - for good - the back-off words of the calls this stream leads
	DeviceBuffer spec_budgets, spec_rows, spec_sd, spec_ctl, spec_backoff, spec_tickets, spec_dplane;
	PinnedBuffer spec_backoff_host;				// the back-off words again, where the host can look without waiting (lossy_kernels.h: LossySpec::backoff_host)
	int spec_groups = 0, spec_streams = 0;		// groups (and streams) of the last run call that went through the speculative launches (0: not eligible)
		PinnedBuffer multi_stage;
		hipEvent_t multi_copied = nullptr; // the copy out of multi_stage of the last call (whatever its stream) has completed
		// A resident run that gave up a wait has advanced the stream's state with invalid frames: the failure is STICKY - every
		// later step / status of a stream that took part in such a call fails until the stream is destroyed.  The error word lives
		// with the call's leading stream and says "some call since the last check"; so the leader numbers the calls it led and
		// keeps the ranges found invalid, and every member remembers who led its last run call and that call's number.
		unsigned int run_epoch = 0, run_arrivals = 0; // residency control block in run_exchange's header (resident_device.h): launches / workgroups so far
		bool failed = false;
		uint64_t led_calls = 0, led_clean = 0;					   // as a leader: run calls led / of those, known good at the last check
		std::vector<std::pair<uint64_t, uint64_t>> led_bad;		   // calls in (first, second] are invalid
		std::weak_ptr<Object> lead;								   // leader of the last run call this stream took part in, when that was another stream
		bool led_by_other = false;								   // ... which `lead` then names (a leader that has been destroyed leaves no record)
		uint64_t lead_call = 0;									   // that call's number with the leader (0: no run call yet)
		LossyObject *leader()
		{
			if (lead_call == 0)
				return nullptr;
			if (!led_by_other)
				return this;
			// a foreign leader that is gone took its books with it: lead_call is a number of ITS numbering and must not be read against this
			// object's own ranges - "no record" (a call that was checked while its leader lived has filed its verdict with every member: is_failed())
			auto l = lead.lock();
			return l ? dynamic_cast<LossyObject *>(l.get()) : nullptr;
		}
		bool is_failed()
		{
			if (failed)
				return true;
			if (LossyObject *l = leader())
				for (const auto &r : l->led_bad)
					if (lead_call > r.first && lead_call <= r.second)
						failed = true;
			return failed;
		}
		// files what a read of the error word (after a wait on the stream the calls were queued on) has shown
		void checked(unsigned int gave_up, hipStream_t st)
		{
			if (gave_up)
			{
				led_bad.emplace_back(led_clean, led_calls);
				(void)hipMemsetAsync(run_exchange.as<unsigned int>() + 16, 0, 4, st);
			}
			led_clean = led_calls;
		}
		~LossyObject() override
		{
			if (multi_copied)
				(void)hipEventDestroy(multi_copied);
		}
	};

	// ---- saver -------------------------------------------------------------------------------
	// reference: struct H264 (video_io.cpp:651-657) + H264_Saver (h264.cpp:1662-1939)
	// (defined with the rir_lossy_* entry points)
	int lossy_step_streams(LossyObject *const *os, int nstreams, const unsigned short *const *d_in, unsigned short *const *d_out, int nframes, int add_loss,
						   int *const *d_errs, int *low_errors, int *high_errors, hipStream_t st, bool force_per_frame = false);

	struct SaverObject : public Object
	{
		const char *type_name() const override { return "H264Saver"; }
		std::string filename;
		int width = 0, height = 0, lossy_height = 0, fps = 50;
		AttrMap global_attrs;
		// parameters (h264.cpp:1709-1781); defaults h264.cpp:1662-1665
		int compressionLevel = 0, lowValueError = 6, highValueError = 2, GOP = RIRB1_DEFAULT_GOP, threads = 1, slices = 1, inputCamera = 0;
		int runningAverage = 32;
		double stdFactor = 5;
		bool removeBadPixels = false, subtractMin = false, subtractLocalMin = false;
		std::string codec = "h264";

		FILE *fp = nullptr;
		bool opened = false;
		int chunk_gop = 0; // GOP frozen at open
		ChunkCodec cc;
		// Chunks are written by a background thread: the call that completes a chunk encodes it, brings tables and payload
		// back into one of two page-locked buffers and hands them over - the 7 MB write into the page cache (most of the
		// 1 ms such a call used to take) overlaps with the caller's next frames.  One job in flight at most.
		struct WriteJob
		{
			ChunkHeader ch;
			size_t hdr_n = 0, toff_n = 0, hdr_b = 0, toff_b = 0;
			uint64_t words = 0;
			int buf = 0;
			uint64_t file_off = 0; // where the chunk goes: the writer uses pwrite (its helper threads write disjoint ranges)
		};
		PinnedBuffer h_out[2];
		int next_buf = 0;
		uint64_t file_pos = 0; // where the next chunk goes (the writer thread owns fp between open and close)
		std::thread writer;
		std::mutex wmu;
		std::condition_variable wcv;
		WriteJob job;
		bool job_ready = false, writer_busy = false, writer_stop = false, write_failed = false;

		void writer_loop()
		{
			std::unique_lock<std::mutex> lk(wmu);
			while (true)
			{
				wcv.wait(lk, [&] { return job_ready || writer_stop; });
				if (!job_ready)
					return;
				const WriteJob j = job;
				job_ready = false, writer_busy = true;
				lk.unlock();
				// chunk header | record headers | tile offsets | payload, the large parts cut over the helper threads (host_copy.cpp: 7 MB into
				// the page cache on one thread take longer than the chunk's trip over the link)
				const char *hb = h_out[j.buf].as<char>();
				const int fd = fileno(fp);
				const uint64_t o_hdr = j.file_off + sizeof(j.ch), o_toff = o_hdr + j.hdr_n * 8, o_pay = o_toff + j.toff_n * 4;
				const bool ok = host_pwrite(fd, &j.ch, sizeof(j.ch), (int64_t)j.file_off) && host_pwrite(fd, hb, j.hdr_n * 8, (int64_t)o_hdr) &&
								host_pwrite(fd, hb + j.hdr_b, j.toff_n * 4, (int64_t)o_toff) &&
								(!j.words || host_pwrite(fd, hb + j.hdr_b + j.toff_b, (size_t)j.words * 8, (int64_t)o_pay));
				lk.lock();
				writer_busy = false;
				if (!ok)
					write_failed = true;
				wcv.notify_all();
			}
		}
		void stop_writer()
		{
			if (!writer.joinable())
				return;
			{
				std::unique_lock<std::mutex> lk(wmu);
				wcv.wait(lk, [&] { return !job_ready && !writer_busy; });
				writer_stop = true;
			}
			wcv.notify_all();
			writer.join();
			writer_stop = false;
		}
		int pending = 0;
		uint64_t nframes = 0;
		// Lossless chunks straight from page-locked memory.  A chunk whose frames all came through add_image is encoded where it lies: the
		// kernels read the staged frames over the link themselves and write tables and payload into the writer's page-locked buffer - no
		// upload calls, no read-back, no wait in the call that completes the chunk (profiles/r05_zero_copy_probe.txt: 711 us for 50 frames
		// of 640x512, the link's rate).  The caller fills the OTHER staging buffer meanwhile; one chunk is in flight at most, the call
		// that completes the next one collects it (and waits for it: the link is then the limit).
		// (decided at open(): the switch, and a chunk of at most kInFlightMaxChunkBytes - the second staging buffer and the second raw
		// buffer are a chunk each, and chunks of gigabytes gain nothing from being in flight)
		static constexpr size_t kInFlightMaxChunkBytes = (size_t)768 << 20;
		bool in_flight_mode = false;
		PinnedBuffer h_stage_b;
		int cur_stage = 0; // 0: cc.h_frames, 1: h_stage_b
		char *stage_ptr() { return cur_stage ? h_stage_b.as<char>() : cc.h_frames.as<char>(); }
		struct Deferred
		{ // a bounded-loss frame whose error budget has not been read back yet: its slot of d_err_slots, where the budget goes
			size_t frame, slot;
			bool with_attrs;
		};
		struct InFlight
		{
			bool active = false;
			int buf = 0, nframes = 0, ntiles = 0;
			uint64_t first_frame = 0;
			std::vector<Deferred> defs; // its bounded-loss frames (their budgets are in h_errs when the event has passed)
			bool have_run = false;
		};
		InFlight flying;
#ifdef RIR_SAVER_DIAG
		double dg_wait_ev = 0, dg_wait_writer = 0, dg_loss = 0, dg_submit = 0, dg_flush = 0;
		int dg_n = 0;
		static double dg_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#endif
		PinnedBuffer h_errs; // [ERR_SLOTS][2] budgets of the chunk in flight, then the run kernel's error word and the poison word
		hipEvent_t fly_ev = nullptr;
		std::vector<IndexEntry> index;
		std::vector<int64_t> times;
		std::vector<AttrMap> frame_attrs;
		std::vector<unsigned short> low_errors, high_errors;
		std::unique_ptr<LossyObject> lossy_obj; // (not in the handle table: only this saver uses it)
		LossyState *lossy = nullptr;			// &lossy_obj->st
		std::vector<unsigned short> lossy_out;
		// Bounded-loss frames whose step has not run yet: frames [raw_from, pending) of the chunk being assembled sit RAW in their
		// page-locked slots; run_deferred_loss() uploads them in one copy and steps them as a run of frames (lossy_kernels.hip:
		// one resident launch instead of three launches and an upload per frame), straight into the chunk's device frames.  Every
		// path that reads or advances the loss state, the chunk's device frames or the error lists calls it first.
		int raw_from = -1;
		int raw_uploaded = 0;	  // raw frames [raw_from, raw_from + raw_uploaded) are in d_raw already (they go up in groups, as they come)
		size_t raw_first_err = 0; // position in `deferred` of frame raw_from
		// The raw frames of a run go up on a COPY STREAM of this saver, into one of two device buffers: the uploads of the next chunk then
		// cross the link while the loss step and the encode of the chunk before it run (on one stream the 0.6 ms of a chunk's uploads
		// queued behind them: 20 us per image, with the copy stream 13-15).  The compute stream waits for a run's uploads (up_ev) before
		// its loss step; the copy stream waits for the loss step that last read a buffer (loss_ev) before it writes that buffer again.
		DeviceBuffer d_raw[2];
		int raw_buf = 0; // the buffer the current run's frames go to
		hipStream_t copy_st = nullptr;
		hipEvent_t up_ev = nullptr, loss_ev[2] = {nullptr, nullptr};
		bool loss_ev_set[2] = {false, false}, copy_st_failed = false;
		hipStream_t upload_stream()
		{
			if (!in_flight_mode || copy_st_failed)
				return default_stream();
			if (!copy_st)
			{
				if (hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&up_ev, hipEventDisableTiming) != hipSuccess ||
					hipEventCreateWithFlags(&loss_ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&loss_ev[1], hipEventDisableTiming) != hipSuccess)
				{
					copy_st_failed = true;
					return default_stream();
				}
			}
			return copy_st;
		}
		bool upload_raw(int upto) // raw frames of the chunk up to slot `upto` -> d_raw[raw_buf]
		{
			const int have = raw_from + raw_uploaded;
			if (raw_from < 0 || upto <= have)
				return true;
			const size_t fbytes = (size_t)width * height * 2;
			hipStream_t us = upload_stream();
			if (!d_raw[raw_buf].reserve((size_t)chunk_gop * fbytes))
				return false;
			if (raw_uploaded == 0 && us == copy_st && loss_ev_set[raw_buf] && !hip_ok(hipStreamWaitEvent(us, loss_ev[raw_buf], 0), "hipStreamWaitEvent"))
				return false; // (the first upload of a run: the loss step that read this buffer last must be through)
			if (!hip_ok(hipMemcpyAsync(d_raw[raw_buf].as<char>() + (size_t)raw_uploaded * fbytes, stage_ptr() + (size_t)have * fbytes, (size_t)(upto - have) * fbytes,
									   hipMemcpyHostToDevice, us),
						"H2D frames"))
				return false;
			raw_uploaded = upto - raw_from;
			return true;
		}

		// The first recording of a process pays for the device runtime's start and for page-locking ~110 MB of staging (160 ms: as much as
		// recording 4 000 frames).  The reference opens its encoder lazily, on the first frame (video_io.cpp:746-752), and so does this saver -
		// but nothing keeps the preparation from starting when the FILE is opened: h264_open_file starts this thread, which brings the device
		// up and page-locks buffers of the sizes the saver will ask for (default GOP; they go to the pool of runtime.cpp, where open() finds
		// them).  Whatever the caller does between opening the file and its first frame - setting parameters, waiting for a camera - then
		// overlaps with it; open() joins the thread first, so a frame that comes at once waits for the same work as before, never for more.
		std::thread warmup;
		void start_warmup()
		{
			const int w = width, h = height, gop = GOP < 1 ? 1 : GOP;
			if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || (int64_t)gop * w * h * 2 >= (1ll << 32))
				return;
			warmup = std::thread([w, h, gop] {
				rir_codec_layout L;
				if (!device_ready() || rir_codec_layout_query(w, h, gop, gop, &L) != 0)
					return;
				{ // the library's stream, and the code object on the device (the first launch of a process loads it: 80 ms)
					DeviceBuffer probe;
					if (probe.reserve(512) && launch_stream_copy_probe(probe.ptr, probe.as<char>() + 256, 256, default_stream()) == hipSuccess)
						(void)hipStreamSynchronize(default_stream());
				}
				PinnedBuffer frames, frames_b, out0, out1;
				const size_t ob = (size_t)L.ntiles * gop * 8 + (((size_t)L.ntiles + 1) * 4 + 7) / 8 * 8 + (size_t)L.stream_max_bytes + 64;
				(void)frames.reserve((size_t)w * h * 2 * gop);
				if (abi_zero_copy() && (size_t)w * h * 2 * gop <= kInFlightMaxChunkBytes)
					(void)frames_b.reserve((size_t)w * h * 2 * gop);
				(void)out0.reserve(ob);
				(void)out1.reserve(ob);
			}); // (the buffers' destructors hand them to the pool)
		}
		void join_warmup()
		{
			if (warmup.joinable())
				warmup.join();
		}

		~SaverObject() override
		{
			join_warmup();
			close();
			stop_writer();
			if (fly_ev)
				(void)hipEventDestroy(fly_ev);
			if (copy_st)
			{
				(void)hipStreamSynchronize(copy_st);
				(void)hipStreamDestroy(copy_st);
			}
			for (hipEvent_t e : {up_ev, loss_ev[0], loss_ev[1]})
				if (e)
					(void)hipEventDestroy(e);
		}

		// the loss-injection state is created on the first lossy call (after the lazy open)
		bool lossy_ready()
		{
			if (lossy)
				return true;
			lossy_obj.reset(new LossyObject());
			lossy = &lossy_obj->st;
			lossy_out.resize((size_t)width * height);
			if (!lossy->prepare(width, height, lossy_height, runningAverage, subtractMin))
			{
				lossy = nullptr;
				lossy_obj.reset();
				return false;
			}
			return true;
		}

		// H264_Saver::addImageLossyNoCamera (h264.cpp:2253-2424)
		bool add_image_lossy(const unsigned short *img, int64_t ts, AttrMap attrs)
		{
			if (!img || !usable() || !open() || !lossy_ready())
				return false;
			// The frame is uploaded and its kernels queued; nothing waits here.  Its error budget is decided on the device and
			// collected later, for all the frames since the last collection at once (resolve_errors: when a chunk is written,
			// when the errors are asked for, at close) - a read-back and a wait per frame cost more than the kernels.
			const bool first = lossy->frames == 0;
			if (deferred.size() >= (size_t)ERR_SLOTS && !resolve_errors())
				return false;
			if (pending >= chunk_gop && !flush_chunk())
				return false;
			// the caller owns its buffer again on return: the image goes through this chunk's page-locked slot (see add_image;
			// the slot is not reused before the chunk has been flushed, i.e. after a wait on the stream)
			const size_t fbytes = (size_t)width * height * 2;
			unsigned short *slot = reinterpret_cast<unsigned short *>(stage_ptr() + (size_t)pending * fbytes);
			host_copy(slot, img, fbytes);
			if (!d_err_slots.reserve((size_t)ERR_SLOTS * 2 * sizeof(int)))
				return false;
			// every frame but the stream's first is left in its slot and stepped later, with the other frames of the chunk, as a run
			// (frames with bad-pixel repair or of a size the run kernels do not take go one by one)
			const int s_px = width * std::max(0, std::min(lossy_height, height)), full_px = width * height;
			if (!first && !removeBadPixels && s_px > 0 && s_px % 8 == 0 && full_px % 8 == 0)
			{
				if (raw_from < 0)
				{
					if (!upload_staged(pending)) // (frames staged before this one are not raw: they go up as they are)
						return false;
					raw_from = pending, raw_uploaded = 0, raw_first_err = deferred.size();
				}
				// (a copy call costs the calling thread ~40 us whatever it moves: with the copy stream, whose transfers overlap the chunk
				// before, the frames go in few large pieces; on the one stream of the copying path small groups keep the link busy early)
				const int group = upload_stream() == copy_st && copy_st ? kRawUploadGroup : kUploadGroup;
				if (pending + 1 - (raw_from + raw_uploaded) >= group && !upload_raw(pending + 1))
					return false;
				deferred.push_back(Deferred{(size_t)nframes, low_errors.size(), true});
				low_errors.push_back(0);
				high_errors.push_back(0);
				return frame_added(ts, attrs);
			}
			if (!run_deferred_loss() ||
				!lossy->queue_host_frame(slot, false, removeBadPixels, lowValueError, highValueError, stdFactor, d_err_slots.as<int>() + 2 * deferred.size()))
				return false;
			if (first)
			{
				if (subtractMin)
				{
					global_attrs["MIN_T"] = std::to_string(lossy->dev.min);
					global_attrs["MIN_T_HEIGHT"] = std::to_string(lossy_height);
				}
				global_attrs["GlobalBackgroundError"] = std::to_string(lowValueError);
				global_attrs["GlobalForegroundError"] = std::to_string(highValueError);
			}
			deferred.push_back(Deferred{(size_t)nframes, low_errors.size(), !first});
			low_errors.push_back(0);
			high_errors.push_back(0);
			return add_image_device(lossy->d_out.as<unsigned short>(), ts, attrs); // stays in HBM: no trip through the host
		}

		bool run_deferred_loss()
		{
			if (raw_from < 0)
				return true;
			const int a = raw_from, n = pending - a;
			const bool up = upload_raw(pending);
			raw_from = -1, raw_uploaded = 0;
			if (n <= 0)
				return true;
			if (!up || !collect_flying()) // (a run call is made below: the verdict on the calls before it is filed first - LossyObject::checked)
				return false;
			const size_t npx = (size_t)width * height;
			hipStream_t st = default_stream();
			const bool side = copy_st && upload_stream() == copy_st; // the run's frames went up on the copy stream
			if (side && (!hip_ok(hipEventRecord(up_ev, copy_st), "hipEventRecord") || !hip_ok(hipStreamWaitEvent(st, up_ev, 0), "hipStreamWaitEvent")))
				return false;
			lossy_obj->low = lowValueError, lossy_obj->high = highValueError, lossy_obj->std_factor = stdFactor, lossy_obj->remove_bad_pixels = false;
			LossyObject *o = lossy_obj.get();
			const unsigned short *in = d_raw[raw_buf].as<unsigned short>();
			unsigned short *out = cc.d_frames.as<unsigned short>() + (size_t)a * npx;
			int *errs = d_err_slots.as<int>() + 2 * raw_first_err;
			if (lossy_step_streams(&o, 1, &in, &out, n, 0, &errs, nullptr, nullptr, st) != 0)
				return false;
			if (side)
			{ // the next run goes to the other buffer; this one is free again when this run's loss step is through
				if (!hip_ok(hipEventRecord(loss_ev[raw_buf], st), "hipEventRecord"))
					return false;
				loss_ev_set[raw_buf] = true;
				raw_buf ^= 1;
			}
			uploaded = pending; // (the chunk's device frames are complete up to here)
			return true;
		}

		// errors of the frames recorded since the last call: into the error lists and the per-frame attributes
		enum
		{
			ERR_SLOTS = 256
		};
		std::vector<Deferred> deferred;
		DeviceBuffer d_err_slots;
		// Sticky failure: once a bounded-loss run has given up (resolve_errors), the frames of the chunk being assembled and the
		// loss state are invalid.  The saver takes no more frames, writes nothing more, and close() finishes the file with the
		// chunks that were complete before.
		bool failed = false;
		bool usable()
		{
			if (!failed)
				return true;
			log_error("h264 saver: a chunk failed earlier (bounded-loss step or encode); this saver takes no more frames (close it: the chunks written before the failure are kept)");
			return false;
		}
		// A chunk that was handed in but cannot be written (its encode failed, its event could not be waited for, the encoder left no
		// plausible length): the recording ends with the chunk before it.  Its frames and whatever came after them leave the books - a chunk
		// that is counted but has no index entry would be a hole whose positions fail on read - and the saver takes no more frames.
		bool abandon_from(int64_t first_frame)
		{
			failed = true;
			nframes = (decltype(nframes))first_frame;
			times.resize((size_t)nframes);
			frame_attrs.resize((size_t)nframes);
			pending = 0, uploaded = 0, raw_from = -1, raw_uploaded = 0;
			deferred.clear();
			flying.defs.clear();
			flying.active = false;
			return false;
		}
		// budgets `e` (pairs, in the order of `defs`) into the error lists and the per-frame attributes; gave_up: the error word of the run
		// kernel (a wait between workgroups that hit its clock) or the poison word of a launch that was called off, read after the frames
		bool file_errors(const std::vector<Deferred> &defs, const int *e, unsigned int gave_up, bool have_run)
		{
			if (have_run)
				lossy_obj->checked(gave_up, default_stream());
			if (gave_up || (lossy_obj && lossy_obj->is_failed()))
			{ // the loss state has been advanced by invalid frames: nothing recorded from here on can be right (usable())
				failed = true;
				log_error("h264 saver: the bounded-loss step of a run of frames gave up waiting (frames invalid)");
				return false;
			}
			for (size_t i = 0; i < defs.size(); ++i)
			{
				const Deferred &d = defs[i];
				low_errors[d.slot] = (unsigned short)e[2 * i];
				high_errors[d.slot] = (unsigned short)e[2 * i + 1];
				if (d.with_attrs && d.frame < frame_attrs.size())
				{
					frame_attrs[d.frame]["BackgroundError"] = std::to_string(e[2 * i]);
					frame_attrs[d.frame]["ForegroundError"] = std::to_string(e[2 * i + 1]);
				}
			}
			return true;
		}
		bool resolve_errors()
		{
			// (the chunk in flight first: its budgets are filed in frame order, and the books of the run calls - LossyObject::checked - are
			// kept in the order the calls were made)
			if (!usable() || !collect_flying() || !run_deferred_loss())
				return false;
			if (deferred.empty())
				return true;
			std::vector<int> e(deferred.size() * 2);
			hipStream_t st = default_stream();
			unsigned int gave_up = 0; // the run kernel's error word (lossy_step_streams): a wait between workgroups that hit its clock
			unsigned int poison = 0;  // ... or a launch that did not become resident and was called off (resident_device.h): its frames were not stepped
			const bool have_run = lossy_obj && lossy_obj->run_exchange.ptr;
			if (!hip_ok(hipMemcpyAsync(e.data(), d_err_slots.ptr, e.size() * sizeof(int), hipMemcpyDeviceToHost, st), "D2H") ||
				(have_run && (!hip_ok(hipMemcpyAsync(&gave_up, lossy_obj->run_exchange.as<unsigned int>() + 16, 4, hipMemcpyDeviceToHost, st), "D2H") ||
							  !hip_ok(hipMemcpyAsync(&poison, lossy_obj->run_exchange.as<unsigned int>() + kLossyRunCtlWord + 2, 4, hipMemcpyDeviceToHost, st), "D2H"))) ||
				!hip_ok(wait_stream(st), "sync"))
				return false;
			const bool ok = file_errors(deferred, e.data(), gave_up | poison, have_run);
			deferred.clear();
			return ok;
		}

		// H264_Saver::addLoss (h264.cpp:2426-2607): the loss is applied to the caller's image, nothing is written
		bool add_loss(unsigned short *img)
		{
			if (!img || !usable() || !open() || !lossy_ready())
				return false;
			int lo = 0, hi = 0;
			if (!resolve_errors() || !lossy->step(img, lossy_out.data(), true, removeBadPixels, lowValueError, highValueError, stdFactor, lo, hi))
				return false;
			low_errors.push_back((unsigned short)lo);
			high_errors.push_back((unsigned short)hi);
			std::memcpy(img, lossy_out.data(), (size_t)width * std::max(0, std::min(lossy_height, height)) * 2);
			return true;
		}

		bool set_parameter(const char *key, const char *value)
		{
			const std::string k = key ? key : "", v = value ? value : "";
			auto as_int = [&]() { return std::atoi(v.c_str()); };
			if (!run_deferred_loss()) // (frames already handed in were recorded under the parameters of their time)
				return false;
			if (k == "lowValueError")
				lowValueError = as_int();
			else if (k == "highValueError")
				highValueError = as_int();
			else if (k == "compressionLevel")
				compressionLevel = as_int();
			else if (k == "codec")
				codec = v;
			else if (k == "GOP")
				GOP = as_int();
			else if (k == "threads")
				threads = as_int();
			else if (k == "slices")
				slices = as_int();
			else if (k == "stdFactor")
				stdFactor = std::atof(v.c_str());
			else if (k == "inputCamera")
				inputCamera = as_int();
			else if (k == "removeBadPixels")
				removeBadPixels = as_int() != 0;
			else if (k == "subtractMin")
				subtractMin = as_int() != 0;
			else if (k == "subtractLocalMin")
				subtractLocalMin = as_int() != 0;
			else if (k == "runningAverage")
				runningAverage = std::min(as_int(), 64);
			else
				return false;
			return true;
		}

		bool open()
		{ // lazy, on the first frame (video_io.cpp:746-752)
			if (opened)
				return true;
			join_warmup();
			if (!device_ready())
				return false;
			if (width <= 0 || height <= 0 || width > 65535 || height > 65535)
			{
				log_error("h264 saver: invalid image size");
				return false;
			}
			chunk_gop = GOP < 1 ? 1 : GOP;
			if ((int64_t)chunk_gop * width * height * 2 >= (1ll << 32))
			{
				log_error("h264 saver: GOP x frame size too large");
				return false;
			}
			if (file_exists(filename.c_str()) && std::remove(filename.c_str()) != 0)
				return false;
			if (!cc.prepare(width, height, chunk_gop))
				return false;
			fp = std::fopen(filename.c_str(), "wb");
			if (!fp)
			{
				log_error("h264 saver: cannot create " + filename);
				return false;
			}
			FtypBox box;
			std::memset(&box, 0, sizeof(box));
			box.size_be[3] = 32;
			std::memcpy(box.type, "ftyp", 4);
			std::memcpy(box.major, "RIRB", 4);
			box.minor_be[3] = 1;
			std::memcpy(box.compat, "RIRBisom", 8);
			FileHeader hd;
			std::memset(&hd, 0, sizeof(hd));
			std::memcpy(hd.magic, "RIRBLOCK", 8);
			hd.version = 1;
			hd.width = width, hd.height = height, hd.gop = chunk_gop, hd.fps = fps;
			if (std::fwrite(&box, sizeof(box), 1, fp) != 1 || std::fwrite(&hd, sizeof(hd), 1, fp) != 1)
				return false;
			if (std::fflush(fp) != 0) // (the chunks are written with pwrite on the descriptor; the FILE comes back into use at close)
				return false;
			file_pos = sizeof(box) + sizeof(hd);
			write_failed = false;
			cur_stage = 0, flying.active = false;
			in_flight_mode = abi_zero_copy() && (size_t)chunk_gop * width * height * 2 <= kInFlightMaxChunkBytes;
			opened = true;
			return true;
		}

		bool flush_chunk()
		{
			hipStream_t st = default_stream();
			rir_codec_layout L;
			if (in_flight_mode && pending > 0)
			{ // nothing is waited for but the chunk BEFORE this one: this chunk's loss step, its encode and its error budgets are queued
#ifdef RIR_SAVER_DIAG
				const double t0 = dg_now();
				struct Fin { SaverObject *s; double t0; ~Fin() { s->dg_flush += dg_now() - t0; ++s->dg_n; } } fin{this, t0};
				if (!usable() || !collect_flying())
					return false;
				const double t1 = dg_now();
				if (!run_deferred_loss())
					return false;
				dg_loss += dg_now() - t1;
#else
				if (!usable() || !collect_flying() || !run_deferred_loss())
					return false;
#endif
				if (rir_codec_layout_query(width, height, pending, chunk_gop, &L) != 0)
					return false;
				const bool staged_only = uploaded == 0; // every frame came through add_image and lies in the staging buffer
				if (!staged_only && !upload_staged(pending))
					return false;
				return submit_chunk(staged_only ? reinterpret_cast<const unsigned short *>(stage_ptr()) : cc.d_frames.as<unsigned short>(), L, st);
			}
			if (!resolve_errors())
				return false;
			if (pending == 0)
				return true;
			if (rir_codec_layout_query(width, height, pending, chunk_gop, &L) != 0)
				return false;
			if (!collect_flying()) // (this path waits for its own chunk: the one before it goes to the file first)
				return false;
			// the frames are in cc.d_frames: uploaded in groups (or produced there) as they were added; the last group goes now
			if (!upload_staged(pending))
				return false;
			if (rir_codec_encode_device(cc.d_frames.as<unsigned short>(), width, height, pending, chunk_gop, cc.d_hdr.as<unsigned long long>(),
										cc.d_tile_off.as<unsigned int>(), cc.d_chunk_off.as<unsigned long long>(),
										cc.d_stream.as<unsigned long long>(), cc.d_ws.ptr, (long long)cc.d_ws.cap, st) != 0)
				return false;
			uint64_t coff[2] = {0, 0};
			if (!hip_ok(hipMemcpyAsync(coff, cc.d_chunk_off.ptr, sizeof(coff), hipMemcpyDeviceToHost, st), "D2H") ||
				!hip_ok(wait_stream(st), "sync"))
				return false;
			const uint64_t words = coff[1];
			// tables and payload come back into page-locked memory (a pageable destination is staged by the runtime at a
			// fraction of the PCIe rate, and a fresh 7 MB vector per chunk costs its page faults) and are written from there
			WriteJob j;
			j.hdr_n = (size_t)L.ntiles * chunk_gop, j.toff_n = (size_t)L.ntiles + 1;
			j.hdr_b = j.hdr_n * 8, j.toff_b = (j.toff_n * 4 + 7) & ~(size_t)7, j.words = words;
			j.buf = next_buf;
			const size_t pay_b = (size_t)words * 8;
			PinnedBuffer &hob = h_out[j.buf]; // (the job in flight, if any, reads the other buffer)
			if (pay_b > (size_t)cc.L.stream_max_bytes || !hob.reserve(out_layout(L.ntiles).total())) // (one size for both ways a chunk is made)
				return false;
			char *hb = hob.as<char>();
			if (!hip_ok(hipMemcpyAsync(hb, cc.d_hdr.ptr, j.hdr_b, hipMemcpyDeviceToHost, st), "D2H") ||
				!hip_ok(hipMemcpyAsync(hb + j.hdr_b, cc.d_tile_off.ptr, j.toff_n * 4, hipMemcpyDeviceToHost, st), "D2H") ||
				(words && !hip_ok(hipMemcpyAsync(hb + j.hdr_b + j.toff_b, cc.d_stream.ptr, pay_b, hipMemcpyDeviceToHost, st), "D2H")) ||
				!hip_ok(wait_stream(st), "sync"))
				return false;
			if (!queue_write(j, pending, nframes - pending, L.ntiles))
				return false;
			next_buf ^= 1;
			pending = 0;
			uploaded = 0;
			return true;
		}

		// a finished chunk (tables and payload in h_out[j.buf]) goes to the writer thread; its place in the file and its index entry are fixed here
		bool queue_write(WriteJob &j, int chunk_frames, uint64_t first_frame, int ntiles)
		{
			std::memset(&j.ch, 0, sizeof(j.ch));
			std::memcpy(j.ch.magic, "CHNK", 4);
			j.ch.nframes = (uint32_t)chunk_frames, j.ch.ntiles = ntiles, j.ch.gop = chunk_gop, j.ch.payload_words = j.words, j.ch.first_frame = first_frame;
			j.file_off = file_pos;
			IndexEntry e{file_pos, first_frame, (uint32_t)chunk_frames, 0};
			file_pos += sizeof(j.ch) + j.hdr_n * 8 + j.toff_n * 4 + (size_t)j.words * 8;
			{
				std::unique_lock<std::mutex> lk(wmu);
				if (!writer.joinable())
					writer = std::thread([this] { writer_loop(); });
#ifdef RIR_SAVER_DIAG
				const double tq = dg_now();
#endif
				wcv.wait(lk, [&] { return !job_ready && !writer_busy; }); // one job in flight: the previous chunk is on disk before this one is queued
#ifdef RIR_SAVER_DIAG
				dg_wait_writer += dg_now() - tq;
#endif
				if (write_failed)
				{
					log_error("h264 saver: write error on " + filename);
					return false;
				}
				job = j;
				job_ready = true;
			}
			wcv.notify_all();
			index.push_back(e);
			return true;
		}

		// layout of a chunk in its page-locked output buffer: record headers | tile offsets (padded to 8) | payload (worst case) | the two chunk offsets
		struct OutLayout
		{
			size_t hdr_n, toff_n, hdr_b, toff_b, pay_max;
			size_t coff_at() const { return hdr_b + toff_b + pay_max; }
			size_t total() const { return coff_at() + 64; }
		};
		OutLayout out_layout(int ntiles) const
		{
			OutLayout o;
			o.hdr_n = (size_t)ntiles * chunk_gop, o.toff_n = (size_t)ntiles + 1;
			o.hdr_b = o.hdr_n * 8, o.toff_b = (o.toff_n * 4 + 7) & ~(size_t)7, o.pay_max = (size_t)cc.L.stream_max_bytes;
			return o;
		}

		// The chunk being assembled - its frames in the current staging buffer (every one of them staged by add_image: the kernels read them
		// over the link) or in cc.d_frames (uploaded, or produced there by the bounded-loss step, which is queued on the same stream) -
		// encoded into h_out[next_buf], nothing waited for.  The error budgets of its bounded-loss frames are copied into page-locked
		// memory behind the loss step and filed when the chunk is collected.  The chunk before it has been collected (one in flight; its
		// staging buffer is the one filled next).
		bool submit_chunk(const unsigned short *chunk_frames, const rir_codec_layout &L, hipStream_t st)
		{
			if (!collect_flying())
				return false;
			flying.defs.clear();
			if (!deferred.empty())
			{ // (slots 0 .. deferred.size() of d_err_slots: the frames of this chunk; the copy runs before any kernel of the next chunk)
				const bool have_run = lossy_obj && lossy_obj->run_exchange.ptr;
				const size_t eb = deferred.size() * 2 * sizeof(int);
				if (!h_errs.reserve((size_t)ERR_SLOTS * 2 * sizeof(int) + 16))
					return false;
				unsigned int *words = reinterpret_cast<unsigned int *>(h_errs.as<char>() + (size_t)ERR_SLOTS * 2 * sizeof(int));
				words[0] = words[1] = 0;
				if (!hip_ok(hipMemcpyAsync(h_errs.ptr, d_err_slots.ptr, eb, hipMemcpyDeviceToHost, st), "D2H") ||
					(have_run && (!hip_ok(hipMemcpyAsync(words, lossy_obj->run_exchange.as<unsigned int>() + 16, 4, hipMemcpyDeviceToHost, st), "D2H") ||
								  !hip_ok(hipMemcpyAsync(words + 1, lossy_obj->run_exchange.as<unsigned int>() + kLossyRunCtlWord + 2, 4, hipMemcpyDeviceToHost, st), "D2H"))))
					return false;
				flying.defs.swap(deferred);
				flying.have_run = have_run;
			}
			const OutLayout o = out_layout(L.ntiles);
			PinnedBuffer &hob = h_out[next_buf]; // (free: the writer's job in flight, if any, reads the other buffer - queue_write waits for the job before)
			const size_t fb = (size_t)width * height * 2 * chunk_gop;
			if (!hob.reserve(o.total()) || !h_stage_b.reserve(fb) || (!fly_ev && !hip_ok(hipEventCreateWithFlags(&fly_ev, hipEventDisableTiming), "hipEventCreate")))
				return abandon_from((int64_t)nframes - pending); // (the chunk's budgets have left `deferred`)
			char *hb = hob.as<char>();
			uint64_t *coff = reinterpret_cast<uint64_t *>(hb + o.coff_at());
			coff[0] = 0, coff[1] = ~0ull; // (what the kernels leave is checked against the buffer before it is believed)
			if (test_hook_is("RIR_DEBUG_SAVER_FAIL_FLYING", "encode") ||
				rir_codec_encode_device(chunk_frames, width, height, pending, chunk_gop,
										reinterpret_cast<unsigned long long *>(hb), reinterpret_cast<unsigned int *>(hb + o.hdr_b),
										reinterpret_cast<unsigned long long *>(coff), reinterpret_cast<unsigned long long *>(hb + o.hdr_b + o.toff_b), cc.d_ws.ptr,
										(long long)cc.d_ws.cap, st) != 0 ||
				!hip_ok(hipEventRecord(fly_ev, st), "hipEventRecord"))
			{ // (the chunk's budgets have left `deferred`: they would never be filed) the recording ends with the chunk before this one
				log_error("h264 saver: a chunk could not be encoded");
				return abandon_from((int64_t)nframes - pending);
			}
			flying.active = true, flying.buf = next_buf, flying.nframes = pending, flying.ntiles = L.ntiles, flying.first_frame = nframes - pending;
			next_buf ^= 1;
			cur_stage ^= 1;
			pending = 0;
			uploaded = 0;
			return true;
		}
		// the chunk in flight, if any: waited for and handed to the writer
		bool collect_flying()
		{
			if (!flying.active)
				return true;
			flying.active = false;
#ifdef RIR_SAVER_DIAG
			const double tw = dg_now();
#endif
			if (!hip_ok(wait_event(fly_ev), "sync") || test_hook_is("RIR_DEBUG_SAVER_FAIL_FLYING", "wait"))
				return abandon_from(flying.first_frame);
#ifdef RIR_SAVER_DIAG
			dg_wait_ev += dg_now() - tw;
#endif
			if (!flying.defs.empty())
			{ // the chunk's bounded-loss frames: their budgets, and whether the run that stepped them gave up (resolve_errors)
				const int *e = h_errs.as<int>();
				const unsigned int *words = reinterpret_cast<const unsigned int *>(h_errs.as<char>() + (size_t)ERR_SLOTS * 2 * sizeof(int));
				const bool ok = file_errors(flying.defs, e, words[0] | words[1], flying.have_run);
				flying.defs.clear();
				if (!ok) // (`failed` is set) this chunk and whatever has been handed in since are invalid: the recording ends with the chunk before it
					return abandon_from(flying.first_frame);
			}
			const OutLayout o = out_layout(flying.ntiles);
			const char *hb = h_out[flying.buf].as<char>();
			const uint64_t *coff = reinterpret_cast<const uint64_t *>(hb + o.coff_at());
			if (coff[0] != 0 || coff[1] > o.pay_max / 8 || test_hook_is("RIR_DEBUG_SAVER_FAIL_FLYING", "length"))
			{
				log_error("h264 saver: the encoder left no chunk length");
				return abandon_from(flying.first_frame);
			}
			WriteJob j;
			j.hdr_n = o.hdr_n, j.toff_n = o.toff_n, j.hdr_b = o.hdr_b, j.toff_b = o.toff_b, j.words = coff[1], j.buf = flying.buf;
			if (!queue_write(j, flying.nframes, flying.first_frame, flying.ntiles))
				return abandon_from(flying.first_frame);
			return true;
		}

		// frames [uploaded, upto) of the chunk being assembled: page-locked staging -> device
		static constexpr int kUploadGroup = 5;
#ifndef RIR_RAW_UPLOAD_GROUP
#define RIR_RAW_UPLOAD_GROUP 25
#endif
		static constexpr int kRawUploadGroup = RIR_RAW_UPLOAD_GROUP;
		int uploaded = 0;
		bool upload_staged(int upto)
		{
			if (upto <= uploaded)
				return true;
			const size_t fbytes = (size_t)width * height * 2;
			if (!hip_ok(hipMemcpyAsync(cc.d_frames.as<char>() + (size_t)uploaded * fbytes, stage_ptr() + (size_t)uploaded * fbytes,
									   (size_t)(upto - uploaded) * fbytes, hipMemcpyHostToDevice, default_stream()),
						"H2D frames"))
				return false;
			uploaded = upto;
			return true;
		}

		bool add_image(const unsigned short *img, int64_t ts, const AttrMap &attrs)
		{
			if (!img || !usable() || !open() || !run_deferred_loss())
				return false;
			if (pending >= chunk_gop && !flush_chunk())
				return false; // an earlier chunk could not be written: no slot is free, never write past the staging buffers
			const size_t fbytes = (size_t)width * height * 2;
			// The caller owns its buffer again when this call returns, whatever kind of host memory it is:
			// copy it into the chunk's page-locked slot, then upload that slot asynchronously - the transfer overlaps
			// with the caller preparing its next frame.  (Uploading straight from the caller's pointer saves 13 us per
			// 640x512 frame but leans on how the runtime treats pageable / pinned sources; not worth the risk.)
			char *slot = stage_ptr() + (size_t)pending * fbytes;
			host_copy(slot, img, fbytes);
			// A chunk made of such frames only is encoded from where they lie (submit_chunk).  Without that (RIR_ABI_ZERO_COPY=0, or a
			// chunk that already holds device frames) uploads go in groups of a few frames, one asynchronous copy each; what is left of a
			// chunk goes when the chunk is flushed
			if ((!in_flight_mode || uploaded > 0) && pending + 1 - uploaded >= kUploadGroup && !upload_staged(pending + 1))
				return false;
			return frame_added(ts, attrs);
		}

		// a frame that is already in device memory (bounded-loss path): device-to-device into the chunk
		bool add_image_device(const unsigned short *d_img, int64_t ts, const AttrMap &attrs)
		{
			if (!d_img || !usable() || !open() || !run_deferred_loss())
				return false;
			if (pending >= chunk_gop && !flush_chunk())
				return false;
			const size_t fbytes = (size_t)width * height * 2;
			if (!upload_staged(pending)) // frames staged on the host before this one go first: `uploaded` is a prefix of the chunk
				return false;
			if (!hip_ok(hipMemcpyAsync(cc.d_frames.as<char>() + (size_t)pending * fbytes, d_img, fbytes, hipMemcpyDeviceToDevice, default_stream()),
						"D2D frame"))
				return false;
			uploaded = pending + 1;
			return frame_added(ts, attrs);
		}

		// a run of frames that lie one behind the other in device memory (a decoded chunk of a loader: rir_transcode_images): as many as the
		// chunk being assembled still takes, in one device-to-device copy.  -> the number taken (>= 1), 0 on failure
		int add_images_device(const unsigned short *d_imgs, int n, const int64_t *ts, const AttrMap *attrs)
		{
			if (!d_imgs || n <= 0 || !usable() || !open() || !run_deferred_loss())
				return 0;
			if (pending >= chunk_gop && !flush_chunk())
				return 0;
			const size_t fbytes = (size_t)width * height * 2;
			const int take = std::min(n, chunk_gop - pending);
			if (!upload_staged(pending)) // frames staged on the host before these go first: `uploaded` is a prefix of the chunk
				return 0;
			if (!hip_ok(hipMemcpyAsync(cc.d_frames.as<char>() + (size_t)pending * fbytes, d_imgs, (size_t)take * fbytes, hipMemcpyDeviceToDevice,
									   default_stream()),
						"D2D frames"))
				return 0;
			uploaded = pending + take;
			for (int k = 0; k < take; ++k)
			{
				++pending;
				++nframes;
				times.push_back(ts[k]);
				frame_attrs.push_back(attrs[k]);
			}
			if (pending == chunk_gop && !flush_chunk())
				return 0;
			return take;
		}

		bool frame_added(int64_t ts, const AttrMap &attrs)
		{
			++pending;
			++nframes;
			times.push_back(ts);
			frame_attrs.push_back(attrs);
			if (pending == chunk_gop)
				return flush_chunk();
			return true;
		}

		int64_t close()
		{ // Finish + trailer (h264.cpp:1883-1912): the file is only valid after this
			if (!opened)
				return 0;
			opened = false;
			bool ok = failed || flush_chunk();
			if (failed)
			{ // (flush_chunk may just have found it out) the frames since the last complete chunk are dropped, the file ends before them:
			  // a valid file of the chunks written before the failure, which the calls that failed have reported
				ok = true;
				nframes -= (uint64_t)pending;
				times.resize((size_t)nframes);
				frame_attrs.resize((size_t)nframes);
				pending = 0, uploaded = 0, raw_from = -1, raw_uploaded = 0;
				deferred.clear();
			}
			if (!failed)
				ok = collect_flying() && ok;
			flying.active = false;
#ifdef RIR_SAVER_DIAG
			if (dg_n)
				fprintf(stderr, "saver diag: %d chunk flushes, per flush: total %.0f us = wait for the chunk in flight %.0f + wait for the writer %.0f + loss step queued %.0f + the rest\n", dg_n,
						dg_flush / dg_n, dg_wait_ev / dg_n, dg_wait_writer / dg_n, dg_loss / dg_n);
#endif
			stop_writer(); // every chunk is in the file from here on
			ok = ok && !write_failed;
			FileHeader hd;
			std::memset(&hd, 0, sizeof(hd));
			std::memcpy(hd.magic, "RIRBLOCK", 8);
			hd.version = 1;
			hd.width = width, hd.height = height, hd.gop = chunk_gop, hd.fps = fps;
			hd.index_offset = file_pos; // (the chunks went through pwrite: the stream's own position has not moved since the header)
			hd.nframes = nframes, hd.nchunks = index.size();
			ok = ok && std::fseek(fp, (long)file_pos, SEEK_SET) == 0;
			if (index.size())
				ok = ok && std::fwrite(index.data(), sizeof(IndexEntry), index.size(), fp) == index.size();
			std::fseek(fp, sizeof(FtypBox), SEEK_SET);
			ok = ok && std::fwrite(&hd, sizeof(hd), 1, fp) == 1;
			std::fclose(fp);
			fp = nullptr;
			if (!ok)
				log_error("h264 saver: error while finishing " + filename);
			FileAttributes fa;
			if (fa.open(filename.c_str()))
			{
				fa.set_global_attributes(global_attrs);
				fa.add_global_attribute("GOP", std::to_string(GOP)); // h264.cpp:1903
				fa.resize((size_t)nframes);
				for (size_t i = 0; i < (size_t)nframes; ++i)
				{
					fa.set_timestamp(i, times[i]);
					fa.set_attributes(i, frame_attrs[i]);
				}
				fa.close();
			}
			struct stat stt;
			const int64_t fsize = stat(filename.c_str(), &stt) == 0 ? (int64_t)stt.st_size : 0;
			times.clear();
			frame_attrs.clear();
			index.clear();
			nframes = 0;
			return fsize;
		}
	};

	// ---- loader ------------------------------------------------------------------------------
	// reference: IRFileLoader (IRFileLoader.cpp) restricted to what the hot path needs:
	// this build's container and raw PCR files.
	struct CameraObject : public Object
	{
		const char *type_name() const override { return "Camera"; }
		enum Kind
		{
			RIRB,
			PCR,
			ZFILE
		} kind = RIRB;
		std::string filename;
		FILE *fp = nullptr;
		std::vector<char> mem; // in-memory file (open_camera_from_memory)
		int width = 0, height = 0, count = 0;
		std::vector<int64_t> times;
		AttrMap global_attrs;
		std::vector<AttrMap> frame_attrs;
		int last_pos = -1;
		std::vector<unsigned short> last_raw;
		int last_raw_pos = -1; // frame held by last_raw (fetched on demand after a filtered read)
		// PCR
		int64_t pcr_start = 0, pcr_transfer = 0;
		int raw_format = FILE_FORMAT_PCR; // kind == PCR: which of the raw formats (PCR, PCR in a BIN envelope, BIN / WEST)
		// ZFile: record offsets, scratch for one compressed frame
		std::vector<int64_t> z_positions;
		std::vector<char> z_buf;
		// ZFile, a reader that goes image after image: the next images are decompressed ahead by a few threads of this object (zstd is one
		// thread per image: 0.45-0.8 ms for a 640x512 image, the whole cost of such a read).  A slot = one image being / having been
		// decompressed ahead; everything under z_mu.
		struct ZSlot
		{
			int pos = -1; // the image this slot is for (-1: free)
			bool busy = false, ready = false, failed = false;
			std::vector<unsigned short> img;
			std::vector<char> comp;
		};
		static constexpr int kZSlots = 12;
		ZSlot z_slots[kZSlots];
		std::mutex z_mu;
		std::condition_variable z_work, z_done;
		std::vector<int> z_todo;
		std::vector<std::thread> z_threads;
		bool z_quit = false;
		int z_last = -2, z_seq = 0;
		// RIRB
		FileHeader hd{};
		std::vector<IndexEntry> index;
		ChunkCodec cc;
		int cached_chunk = -1;
		// Sequential readers: while the images of chunk k are handed out from page-locked memory, helper threads ("lanes") read chunks
		// k + 1 and k + 2 from the file and decode them, each on a stream and into a buffer set (`nx`) of its own, so that their images are
		// in page-locked memory when the reader gets there; the lane's buffer set and the current one are then swapped.
#ifndef RIR_LOADER_LANES
#define RIR_LOADER_LANES 2
#endif
		static constexpr int kLanes = RIR_LOADER_LANES;
		ChunkCodec nx[kLanes];
		struct Prefetch
		{
			std::thread th;
			std::mutex mu;
			std::condition_variable cv;
			int want = -1;	  // chunk the helper should fetch (-1: none)
			int have = -1;	  // chunk whose images are in nx (valid when ok)
			bool busy = false, ok = false, quit = false;
			hipStream_t stream = nullptr;
			int device = 0;
		} pf[kLanes];
		// sequential read-ahead: images [host_first, host_end) of chunk host_chunk are (being) copied to cc.h_frames
		int seq_run = 0, host_chunk = -1, host_base = 0, host_first = 0, host_end = 0;
		bool host_pending = false;
		// read-back filters
		// A sequential reader with a read-back filter switched on: the filters run once over the whole decoded chunk (batched kernels) and the
		// filtered images wait in page-locked memory of their own (filt_host) - a read is then a host copy, like an unfiltered one.
		// filt_chunk: the chunk whose filtered images are there (-1: none); filt_state: a count of the changes to the filters' settings,
		// filt_made_at: its value when filt_chunk was made.
		DeviceBuffer filt_dev, filt_shift;
		PinnedBuffer filt_host;
		int filt_chunk = -1, filt_run = 0;
		unsigned filt_state = 0, filt_made_at = 0;
		bool bp_enabled = false;
		int bp_handle = 0;
		bool motion_enabled = false;
		std::vector<float> shifts; // (x,y) per frame
		int min_T = 0, min_T_rows = 0; // global attributes MIN_T / MIN_T_HEIGHT (IRFileLoader.cpp:905-921)
		std::vector<float> inv_emi;	   // inverse emissivities as set through set_[global_]emissivity (IRVideoLoader.h:29-30)
		float global_emi = 1.f;

		~CameraObject() override
		{
			stop_prefetch();
			stop_zfile_ahead();
			if (fp)
				std::fclose(fp);
			if (bp_handle > 0)
				bad_pixels_destroy(bp_handle);
		}

		size_t total_size()
		{
			if (!fp)
				return mem.size();
			struct stat st;
			return fstat(fileno(fp), &st) == 0 ? (size_t)st.st_size : 0;
		}
		bool read_at(uint64_t off, void *dst, size_t n)
		{
			if (!fp)
			{
				if (off + n > mem.size())
					return false;
				std::memcpy(dst, mem.data() + off, n);
				return true;
			}
			// pread: no shared file position, so the read-ahead thread and the caller's thread may both read
			char *d = static_cast<char *>(dst);
			const int fd = fileno(fp);
			while (n > 0)
			{
				const ssize_t r = pread(fd, d, n, (off_t)off);
				if (r <= 0)
					return false;
				d += r, off += (uint64_t)r, n -= (size_t)r;
			}
			return true;
		}

		// a chunk's tables / payload: the same, cut over the helper threads (host_copy.cpp)
		bool read_at_large(uint64_t off, void *dst, size_t n)
		{
			if (!fp)
			{
				if (off + n > mem.size())
					return false;
				host_copy(dst, mem.data() + off, n);
				return true;
			}
			return host_pread(fileno(fp), dst, n, (int64_t)off);
		}

		bool open_common()
		{
			const size_t fsize = total_size();
			char buf[2000];
			std::memset(buf, 0, sizeof(buf));
			if (!read_at(0, buf, std::min(fsize, sizeof(buf))))
				return false;
			FtypBox box;
			std::memcpy(&box, buf, sizeof(box));
			if (std::memcmp(box.type, "ftyp", 4) == 0)
			{
				if (std::memcmp(box.major, "RIRB", 4) != 0)
				{
					log_error("MP4/H.264 files written by the reference (ffmpeg/x264) are not readable by this library");
					return false;
				}
				std::memcpy(&hd, buf + sizeof(FtypBox), sizeof(hd));
				if (std::memcmp(hd.magic, "RIRBLOCK", 8) != 0 || hd.version != 1 || hd.index_offset == 0)
				{
					log_error("RIRB file not closed properly or unknown version");
					return false;
				}
				// the header comes from a file: nothing in it is trusted before it has been checked against the file size
				if (hd.width == 0 || hd.height == 0 || hd.width > 65535 || hd.height > 65535 || hd.gop == 0 || hd.gop > (1u << 20) ||
					hd.nframes > 0x7fffffffull || hd.nchunks > fsize / sizeof(IndexEntry) || hd.index_offset > fsize ||
					hd.nchunks * sizeof(IndexEntry) > fsize - hd.index_offset || hd.nframes > hd.nchunks * (uint64_t)hd.gop)
				{
					log_error("RIRB file: inconsistent header");
					return false;
				}
				kind = RIRB;
				width = (int)hd.width, height = (int)hd.height, count = (int)hd.nframes;
				index.resize((size_t)hd.nchunks);
				if (hd.nchunks && !read_at(hd.index_offset, index.data(), sizeof(IndexEntry) * index.size()))
					return false;
				for (const IndexEntry &e : index)
					if (e.file_offset > fsize || e.nframes == 0 || e.nframes > hd.gop || e.first_frame > hd.nframes ||
						e.nframes > hd.nframes - e.first_frame)
					{
						log_error("RIRB file: inconsistent chunk index");
						return false;
					}
				if (count > 0 && !cc.prepare(width, height, (int)hd.gop, false))
					return false;
			}
			else
			{
				PcrHeader ph;
				std::memcpy(&ph, buf, sizeof(ph));
				// detection rules of IRFileLoader.cpp:130-165 for plain PCR files
				const bool lab = ph.Bits == 16 && ph.X == 640 && ph.Y == 512 && ph.Frequency == 50;
				const bool pcr = ph.Bits == 16 && std::abs(ph.TransfertSize - ph.X * ph.Y * 2) < 2000 && ph.X > 0 && ph.Y > 0 && ph.X < 2000 && ph.Y < 2000;
				// ... a PCR header behind a 133-byte envelope (IRFileLoader.cpp:166-181), tested next like there
				PcrHeader pe;
				std::memcpy(&pe, buf + 128 + 5, sizeof(pe));
				const bool pcr_enc = !lab && !pcr && pe.Bits == 16 && std::abs(pe.TransfertSize - pe.X * pe.Y * 2) < 2000 && pe.X > 0 && pe.Y > 0 &&
									 pe.X < 1000 && pe.Y < 1000;
				ZHeader zh;
				ZTrigger zt;
				std::memcpy(&zh, buf, sizeof(zh));
				std::memcpy(&zt, buf + sizeof(zh), sizeof(zt));
				// ... the BIN (WEST) file: the two 128-byte blocks of a ZFile with compression 0, raw frames behind them (IRFileLoader.cpp:182-209)
				const bool west = !lab && !pcr && !pcr_enc && zh.version < 10 && zh.compression == 0 && zh.triggers == 1 && zt.data_size_x > 0 &&
								  zt.data_size_x < 1000 && zt.data_size_y > 0 && zt.data_size_y < 1000 && zt.rate > 0 && zt.rate < 1000;
				// IRFileLoader.cpp:213-236, tested after those
				const bool zfile = !lab && !pcr && !pcr_enc && !west && zh.version == 1 && zh.compression >= 1 && zh.compression <= 3 && zh.triggers == 1 &&
								   zt.data_size_x > 0 && zt.data_size_x < 3000 && zt.data_size_y > 0 && zt.data_size_y < 3000 && zt.rate > 0 && zt.rate < 1000;
				if (zfile)
				{
					if (zh.compression != 1)
					{
						log_error("ZFile: only compression method 1 (zstd) is readable, blosc methods are not");
						return false;
					}
					kind = ZFILE;
					width = (int)zt.data_size_x, height = (int)zt.data_size_y;
					count = (int)std::min<uint64_t>(zt.samples, 0x7fffffffull);
				}
				else if (pcr_enc)
					open_pcr(pe, false, fsize, 128 + 5 + (int64_t)sizeof(PcrHeader), FILE_FORMAT_PCR_ENCAPSULATED);
				else if (west)
				{
					PcrHeader wh;
					std::memset(&wh, 0, sizeof(wh));
					wh.X = (int)zt.data_size_x, wh.Y = (int)zt.data_size_y, wh.Bits = 16;
					wh.TransfertSize = wh.X * wh.Y * 2, wh.Frequency = (int)zt.rate;
					open_pcr(wh, false, fsize, (int64_t)(sizeof(ZHeader) + sizeof(ZTrigger)), FILE_FORMAT_WEST);
				}
				else if (!lab && !pcr)
				{
					log_error("unsupported file format (this library reads its own RIRB files, ZFile (zstd) files and the raw formats: PCR, PCR in a BIN "
							  "envelope, BIN)");
					return false;
				}
				else
					open_pcr(ph, lab, fsize, (int64_t)sizeof(PcrHeader), FILE_FORMAT_PCR);
				if (kind == PCR && count <= 0)
					return false;
			}
			// metadata trailer, when present
			std::vector<int64_t> ttimes;
			bool has_trailer = false;
			size_t trailer_size = 0;
			{
				const size_t fs = total_size();
				char tail[30];
				if (fs >= sizeof(tail) && read_at(fs - sizeof(tail), tail, sizeof(tail)) && std::memcmp(tail + 16, "H264ATTRIBUTES", 14) == 0)
				{
					uint64_t tsize;
					std::memcpy(&tsize, tail + 8, 8);
					if (tsize <= fs)
					{
						std::vector<char> tb((size_t)tsize);
						if (read_at(fs - tsize, tb.data(), tb.size()))
							has_trailer = FileAttributes::parse(tb.data(), tb.size(), global_attrs, frame_attrs, ttimes) != 0;
						if (has_trailer)
							trailer_size = (size_t)tsize;
					}
				}
			}
			if (kind == ZFILE && !open_zfile(fsize, has_trailer, trailer_size, ttimes))
				return false;
			if (kind == RIRB)
			{
				times.assign(count, 0);
				for (int i = 0; i < count; ++i)
					times[i] = (has_trailer && i < (int)ttimes.size()) ? ttimes[i] : (int64_t)(i * (1000000000.0 / (hd.fps ? hd.fps : 50)));
			}
			frame_attrs.resize(count);
			last_raw.assign((size_t)width * height, 0);
			min_T = min_T_rows = 0;
			if (kind == RIRB)
			{ // IRFileLoader::open (IRFileLoader.cpp:905-921)
				auto it = global_attrs.find("MIN_T");
				if (it != global_attrs.end())
					min_T = std::atoi(it->second.c_str());
				it = global_attrs.find("MIN_T_HEIGHT");
				if (it != global_attrs.end())
					min_T_rows = std::atoi(it->second.c_str());
				if (min_T_rows == 0)
					min_T_rows = height - 3;
			}
			return true;
		}

		// raw frames behind a header: PCR, PCR in a BIN envelope, BIN / WEST (IRFileLoader.cpp:130-209 detection, :404-451 geometry and
		// timestamps - the three share that code upstream too)
		void open_pcr(PcrHeader ph, bool lab, size_t fsize, int64_t start, int format)
		{
			if (lab)
				ph.TransfertSize = ph.X * ph.Y * 2;
			kind = PCR;
			raw_format = format;
			width = ph.X, height = ph.Y;
			pcr_start = start;
			pcr_transfer = ph.TransfertSize;
			count = (pcr_transfer > 0 && (int64_t)fsize > pcr_start) ? (int)std::min<int64_t>(((int64_t)fsize - pcr_start) / pcr_transfer, 0x7fffffff) : 0;
			if (count <= 0)
				return;
			// timestamps: last 8 bytes of each frame when strictly increasing (IRFileLoader.cpp:255-282)
			times.assign(count, 0);
			bool has_times = true;
			for (int i = 0; i < count && has_times; ++i)
			{
				int64_t t = 0;
				if (!read_at(pcr_start + pcr_transfer * (int64_t)(i + 1) - 8, &t, 8))
					has_times = false;
				times[i] = t;
				if (i > 0 && times[i] <= times[i - 1])
					has_times = false;
			}
			if (!has_times)
			{ // IRFileLoader.cpp:421-431
				int freq = ph.Frequency <= 0 ? 50 : ph.Frequency;
				const double sampling = 1000000000.0 / (double)freq;
				for (int i = 0; i < count; ++i)
					times[i] = (int64_t)(i * sampling);
			}
			else
			{ // IRFileLoader.cpp:433-451
				const int64_t t0 = times[0];
				if (t0 > 28000 && t0 < 32000)
					for (auto &t : times)
						t *= 1000000;
				else if (!(times.front() < -1000000000 || times.back() > 1000000000))
					for (auto &t : times)
						t = (t - t0) * 1000000;
			}
			if (!times.empty() && times.front() > 28000000000LL && times.front() < 32000000000LL)
				for (auto &t : times)
					t -= 32000000000LL;
		}

		// ZFile index (ZFile.cpp:113-262): offsets from the trailer attribute "positions" when it is there and
		// consistent, else by walking the records; timestamps converted as IRFileLoader.cpp:355-376 does.
		bool open_zfile(size_t fsize, bool has_trailer, size_t trailer_size, const std::vector<int64_t> &ttimes)
		{
			if (!ZstdApi::get().ok)
			{
				log_error("ZFile: libzstd is not available on this host");
				return false;
			}
			const size_t npx = (size_t)width * height;
			const size_t data_end = fsize - (has_trailer ? trailer_size : 0);
			z_positions.clear();
			times.clear();
			auto it = global_attrs.find("positions");
			if (has_trailer && it != global_attrs.end() && it->second.size() == ttimes.size() * 8 && !ttimes.empty())
			{
				z_positions.resize(ttimes.size());
				std::memcpy(z_positions.data(), it->second.data(), it->second.size());
				times = ttimes;
			}
			else
			{
				const uint64_t declared = (uint64_t)count; // 0 = unknown: walk to the end of the data
				uint64_t pos = sizeof(ZHeader) + sizeof(ZTrigger);
				while (pos + 12 <= data_end && (declared == 0 || z_positions.size() < declared))
				{
					int64_t t = 0;
					uint32_t csize = 0;
					if (!read_at(pos, &t, 8) || !read_at(pos + 8, &csize, 4) || pos + 12 + csize > data_end)
						break;
					z_positions.push_back((int64_t)pos);
					times.push_back(t);
					pos += 12 + (uint64_t)csize;
				}
			}
			for (int64_t p : z_positions)
				if (p < (int64_t)(sizeof(ZHeader) + sizeof(ZTrigger)) || (uint64_t)p + 12 > fsize)
				{
					log_error("ZFile: inconsistent image positions");
					return false;
				}
			count = (int)z_positions.size();
			if (count == 0)
				return false;
			z_buf.resize(ZstdApi::get().compressBound(npx * 2));
			const int64_t t0 = times[0];
			if (t0 > 28000 && t0 < 32000)
				for (auto &t : times)
					t = t * 1000000 - 10000000;
			else if (!(times.front() < -1000000000 || times.back() > 1000000000))
				for (auto &t : times)
					t = (t - t0) * 1000000;
			return true;
		}

		// one ZFile record -> one frame (ZFile.cpp:544-629)
		// one image of a ZFile, decompressed into `out` (npx cells) through `comp` (any thread: pread and one-shot zstd keep no state here)
		bool zfile_image(int pos, unsigned short *out, std::vector<char> &comp)
		{
			const size_t npx = (size_t)width * height;
			const uint64_t off = (uint64_t)z_positions[pos];
			uint32_t csize = 0;
			if (!read_at(off + 8, &csize, 4) || csize > comp.size() || !read_at(off + 12, comp.data(), csize))
				return false;
			const ZstdApi &z = ZstdApi::get();
			const size_t r = z.decompress(out, npx * 2, comp.data(), csize);
			return !z.isError(r) && r == npx * 2;
		}
		void zfile_ahead_loop()
		{
			std::unique_lock<std::mutex> lk(z_mu);
			for (;;)
			{
				z_work.wait(lk, [&] { return z_quit || !z_todo.empty(); });
				if (z_quit)
					return;
				ZSlot &sl = z_slots[z_todo.front()];
				z_todo.erase(z_todo.begin());
				const int pos = sl.pos;
				lk.unlock();
				if (sl.img.size() != (size_t)width * height)
					sl.img.resize((size_t)width * height);
				if (sl.comp.size() != z_buf.size())
					sl.comp.resize(z_buf.size());
				const bool ok = zfile_image(pos, sl.img.data(), sl.comp);
				lk.lock();
				sl.busy = false, sl.ready = ok, sl.failed = !ok;
				z_done.notify_all();
			}
		}
		void stop_zfile_ahead()
		{
			{
				std::unique_lock<std::mutex> lk(z_mu);
				z_quit = true;
			}
			z_work.notify_all();
			for (auto &t : z_threads)
				if (t.joinable())
					t.join();
			z_threads.clear();
		}
		bool read_zfile(int pos, unsigned short *out)
		{
			const size_t npx = (size_t)width * height;
			bool served = false, found = false;
			{
				std::unique_lock<std::mutex> lk(z_mu);
				for (ZSlot &sl : z_slots)
					if (sl.pos == pos)
					{ // decompressed ahead, or on its way
						found = true;
						z_done.wait(lk, [&] { return !sl.busy; });
						if (sl.ready)
						{
							lk.unlock();
							host_copy(out, sl.img.data(), npx * 2); // (the slot is this image's until it is freed below: nobody else touches it)
							lk.lock();
							served = true;
						}
						sl.pos = -1, sl.ready = sl.failed = false;
						break;
					}
			}
			if (!served && !zfile_image(pos, out, z_buf))
				return false;
			(void)found;
			// the images after this one, once the reader has shown that it goes image after image
			z_seq = (pos == z_last + 1) ? z_seq + 1 : 0;
			z_last = pos;
			const int ahead_threads = zfile_threads();
			if (ahead_threads == 0)
				return true;
			std::unique_lock<std::mutex> lk(z_mu);
			if (z_seq < 2)
			{ // a reader that jumps about: what was made ahead is dropped (slots still being worked on free themselves when asked for - or never)
				for (ZSlot &sl : z_slots)
					if (!sl.busy && sl.pos >= 0)
						sl.pos = -1, sl.ready = sl.failed = false;
				for (size_t i = 0; i < z_todo.size();)
				{
					z_slots[z_todo[i]].pos = -1, z_slots[z_todo[i]].busy = false;
					z_todo.erase(z_todo.begin() + (std::ptrdiff_t)i);
				}
				return true;
			}
			bool queued = false;
			for (int p = pos + 1; p < count && p <= pos + kZSlots; ++p)
			{
				bool there = false;
				for (const ZSlot &sl : z_slots)
					there = there || sl.pos == p;
				if (there)
					continue;
				ZSlot *free_slot = nullptr;
				for (ZSlot &sl : z_slots)
					if (!sl.busy && (sl.pos < 0 || sl.pos <= pos || sl.pos > pos + kZSlots)) // free, an image the reader has passed, or one far from here
					{
						free_slot = &sl;
						break;
					}
				if (!free_slot)
					break;
				free_slot->pos = p, free_slot->busy = true, free_slot->ready = free_slot->failed = false;
				z_todo.push_back((int)(free_slot - z_slots));
				queued = true;
			}
			if (queued)
			{
				if (z_threads.empty())
					for (int t = 0; t < ahead_threads; ++t)
						z_threads.emplace_back([this] { zfile_ahead_loop(); });
				z_work.notify_all();
			}
			return true;
		}

		// Chunk c of the file -> ctx.d_frames (decoded, MIN_T added back) on stream st; with to_host the decoded images follow
		// into ctx.h_frames (one asynchronous copy).  Waits for the stream.  quiet: failures are not logged (the read-ahead
		// thread: the caller's own attempt will say what is wrong).
		bool decode_chunk_into(ChunkCodec &ctx, int c, hipStream_t st, bool to_host, bool quiet)
		{
			auto fail = [&](const char *why) {
				if (!quiet)
					log_error(why);
				return false;
			};
			const IndexEntry &e = index[c];
			ChunkHeader ch;
			if (!read_at(e.file_offset, &ch, sizeof(ch)) || std::memcmp(ch.magic, "CHNK", 4) != 0 || (int)ch.gop != ctx.gop ||
				(int)ch.ntiles != ctx.L.ntiles || ch.nframes == 0 || (int)ch.nframes > ctx.gop || ch.nframes != e.nframes ||
				ch.first_frame != e.first_frame || ch.payload_words > (uint64_t)ctx.L.stream_max_bytes / 8)
				return fail("RIRB file: corrupted chunk header");
			// tables and payload are read straight into page-locked memory (kept for the life of the object): no page faults of a
			// fresh 7 MB vector per chunk, and the uploads below run at the PCIe rate instead of through the runtime's staging
			const size_t hdr_n = (size_t)ch.ntiles * ch.gop, toff_n = (size_t)ch.ntiles + 1, pay_n = (size_t)ch.payload_words + 1;
			const size_t hdr_b = hdr_n * 8, toff_b = (toff_n * 4 + 7) & ~(size_t)7;
			// (+ the guard word, the two chunk offsets and two flag words).  Sized for THIS chunk with a quarter to spare, not for the worst
			// case a chunk can be (the raw size): a recording compresses about five times, and page-locking is what the first read of a
			// process waits for (0.3 ms per megabyte, three buffer sets); a later chunk that needs more makes the buffer grow.
			const size_t in_need = hdr_b + toff_b + pay_n * 8 + 64;
			if (ctx.h_in.cap < in_need && !ctx.h_in.reserve(std::min(in_need + in_need / 4, hdr_b + toff_b + (size_t)ctx.L.stream_max_bytes + 64)))
				return false;
			uint64_t *hdr = ctx.h_in.as<uint64_t>();
			uint32_t *toff = reinterpret_cast<uint32_t *>(ctx.h_in.as<char>() + hdr_b);
			uint64_t *payload = reinterpret_cast<uint64_t *>(ctx.h_in.as<char>() + hdr_b + toff_b);
			// (the previous uploads from this buffer have completed: this function waits for its stream before it returns)
			uint64_t off = e.file_offset + sizeof(ch);
			if (!read_at_large(off, hdr, hdr_b))
				return false;
			off += hdr_b;
			if (!read_at(off, toff, toff_n * 4))
				return false;
			off += toff_n * 4;
			if (ch.payload_words && !read_at_large(off, payload, (size_t)ch.payload_words * 8))
				return false;
			payload[ch.payload_words] = 0;
			// the offsets table comes from the file: monotone and ending exactly at the payload length, or the chunk is
			// refused before anything reaches the device (the kernel checks again against the stream length it is given)
			bool toff_ok = toff[0] == 0 && (uint64_t)toff[ch.ntiles] == ch.payload_words;
			for (size_t t = 0; t < (size_t)ch.ntiles && toff_ok; ++t)
				toff_ok = toff[t] <= toff[t + 1];
			if (!toff_ok)
				return fail("RIRB file: corrupted tile offsets");
			uint64_t *coff = reinterpret_cast<uint64_t *>(payload + pay_n); // (page-locked too: the copy below is asynchronous)
			coff[0] = 0, coff[1] = ch.payload_words;
			int *flags = reinterpret_cast<int *>(coff + 2); // [0] zero for the device's error word, [1] the word read back
			flags[0] = 0, flags[1] = 0;
			// A chunk that is only wanted on the host (the read-ahead of a sequential reader, no read-back filter in play) is decoded where
			// it lies: the kernel reads tables and payload from this page-locked buffer and writes the images into ctx.h_frames, both over
			// the link, at the link's rate (profiles/r05_zero_copy_probe.txt: 667 us for 50 frames of 640x512 against 590 for the bare copy of
			// the images) - no upload calls, no device copy of the chunk, no copy back.  ctx.on_device says which it was.
			const bool host_only = to_host && abi_zero_copy() && ctx.h_frames.ptr && !(min_T && min_T_rows > 0) && !bp_enabled && !motion_enabled;
			ctx.on_device = !host_only;
			if (!hip_ok(hipMemcpyAsync(ctx.d_err.ptr, flags, sizeof(int), hipMemcpyHostToDevice, st), "H2D"))
				return false;
			if (host_only)
			{
				if (rir_codec_decode_device(reinterpret_cast<unsigned long long *>(hdr), toff, reinterpret_cast<unsigned long long *>(coff),
											reinterpret_cast<unsigned long long *>(payload), (long long)ch.payload_words, width, height, (int)ch.nframes, ctx.gop,
											ctx.h_frames.as<unsigned short>(), ctx.d_err.as<int>(), st) != 0 ||
					!hip_ok(hipMemcpyAsync(flags + 1, ctx.d_err.ptr, sizeof(int), hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
					return false;
				if (flags[1])
					return fail("RIRB file: malformed chunk payload");
				return true;
			}
			if (!hip_ok(hipMemcpyAsync(ctx.d_hdr.ptr, hdr, hdr_b, hipMemcpyHostToDevice, st), "H2D") ||
				!hip_ok(hipMemcpyAsync(ctx.d_tile_off.ptr, toff, toff_n * 4, hipMemcpyHostToDevice, st), "H2D") ||
				!hip_ok(hipMemcpyAsync(ctx.d_chunk_off.ptr, coff, 16, hipMemcpyHostToDevice, st), "H2D") ||
				!hip_ok(hipMemcpyAsync(ctx.d_stream.ptr, payload, pay_n * 8, hipMemcpyHostToDevice, st), "H2D"))
				return false;
			if (rir_codec_decode_device(ctx.d_hdr.as<unsigned long long>(), ctx.d_tile_off.as<unsigned int>(), ctx.d_chunk_off.as<unsigned long long>(),
										ctx.d_stream.as<unsigned long long>(), (long long)ch.payload_words, width, height, (int)ch.nframes, ctx.gop,
										ctx.d_frames.as<unsigned short>(), ctx.d_err.as<int>(), st) != 0)
				return false;
			// frames recorded with subtractMin: add the stored minimum back (IRFileLoader.cpp:1173-1179)
			if (min_T && min_T_rows > 0 &&
				!hip_ok(launch_lossy_add_min(ctx.d_frames.as<uint16_t>(), (int64_t)width * height, width * std::min(min_T_rows, height), (int)ch.nframes,
											 (uint32_t)min_T, st),
						"add min"))
				return false;
			// the decoded chunk stays in HBM (ctx.d_frames): frames are filtered there and only the requested
			// frame crosses PCIe (IRFileLoader::readImage hands out one frame per call) - or, for a sequential reader, the whole chunk at once
			if (!hip_ok(hipMemcpyAsync(flags + 1, ctx.d_err.ptr, sizeof(int), hipMemcpyDeviceToHost, st), "D2H"))
				return false;
			if (to_host && ctx.h_frames.ptr &&
				!hip_ok(hipMemcpyAsync(ctx.h_frames.ptr, ctx.d_frames.ptr, (size_t)ch.nframes * width * height * 2, hipMemcpyDeviceToHost, st), "D2H"))
				return false;
			if (!hip_ok(wait_stream(st), "sync"))
				return false;
			if (flags[1])
				return fail("RIRB file: malformed chunk payload");
			return true;
		}

		// ---- read-ahead of the next chunks (sequential readers) ----
		// Two lanes, each a thread with its own stream and buffer set: while the images of chunk k are handed out, one lane has chunk k + 1
		// (read from the file, decoded, its images in page-locked memory) and the other is busy with chunk k + 2 - a lane's file read
		// overlaps with the other lane's trip over the link, so the link, not a lane's latency (read + decode + copy: longer than the
		// reader needs for a chunk), sets the pace.
		void prefetch_loop(int lane)
		{
			Prefetch &p = pf[lane];
			(void)hipSetDevice(p.device);
			std::unique_lock<std::mutex> lk(p.mu);
			for (;;)
			{
				p.cv.wait(lk, [&] { return p.quit || p.want >= 0; });
				if (p.quit)
					return;
				const int c = p.want;
				p.want = -1, p.busy = true, p.ok = false, p.have = c;
				lk.unlock();
				// (the lane's stream is made here, on the lane's thread: creating a stream takes milliseconds, which the reader's call
				// that starts the lane should not wait for)
				if (!p.stream && hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess)
					p.stream = nullptr;
				const bool ok = p.stream && nx[lane].prepare(width, height, (int)hd.gop, false) && decode_chunk_into(nx[lane], c, p.stream, true, true);
				lk.lock();
				p.ok = ok, p.busy = false;
				p.cv.notify_all();
			}
		}
		// the lane that has, is fetching or is about to fetch chunk c (-1: none)
		int lane_of(int c)
		{
			for (int l = 0; l < kLanes; ++l)
			{
				std::unique_lock<std::mutex> lk(pf[l].mu);
				if (pf[l].want == c || (pf[l].have == c && (pf[l].busy || pf[l].ok)))
					return l;
			}
			return -1;
		}
		void start_prefetch(int c)
		{
			if (c < 0 || c >= (int)index.size() || kind != RIRB || lane_of(c) >= 0)
				return;
			for (int l = 0; l < kLanes; ++l)
			{
				Prefetch &p = pf[l];
				std::unique_lock<std::mutex> lk(p.mu);
				// a lane is free when it is idle and what it holds is not one of the chunks the reader comes to next
				const bool holds_next = p.ok && p.have > cached_chunk && p.have <= cached_chunk + kLanes;
				if (p.busy || p.want >= 0 || holds_next)
					continue;
				if (!p.th.joinable())
				{
					if (hipGetDevice(&p.device) != hipSuccess)
						return;
					p.th = std::thread([this, l] { prefetch_loop(l); });
				}
				p.want = c, p.have = -1, p.ok = false;
				p.cv.notify_all();
				return;
			}
		}
		// true when chunk c was fetched ahead: its buffers become the current ones
		bool take_prefetched(int c)
		{
			const int l = lane_of(c);
			if (l < 0)
				return false;
			Prefetch &p = pf[l];
			std::unique_lock<std::mutex> lk(p.mu);
			if (p.want != c && p.have != c)
				return false;
			p.cv.wait(lk, [&] { return p.want < 0 && !p.busy; });
			if (p.have != c || !p.ok)
				return false;
			cc.swap_with(nx[l]);
			p.have = -1, p.ok = false;
			return true;
		}
		void stop_prefetch()
		{
			for (int l = 0; l < kLanes; ++l)
			{
				Prefetch &p = pf[l];
				{
					std::unique_lock<std::mutex> lk(p.mu);
					p.cv.wait(lk, [&] { return p.want < 0 && !p.busy; });
					p.quit = true;
				}
				p.cv.notify_all();
				if (p.th.joinable())
					p.th.join();
				if (p.stream)
					(void)hipStreamDestroy(p.stream);
				p.stream = nullptr;
			}
		}

		bool decode_chunk(int c, bool need_device = true)
		{
			if (c == cached_chunk && (cc.on_device || !need_device))
				return true;
			if (!device_ready())
				return false;
			if (host_pending)
			{ // (images of the current chunk are still on their way into cc.h_frames: those buffers are about to change hands)
				if (!hip_ok(wait_stream(default_stream()), "sync"))
					return false;
				host_pending = false;
			}
			if (c != cached_chunk && take_prefetched(c))
			{ // decoded ahead, images already in page-locked memory
				const IndexEntry &e = index[c];
				cached_chunk = c;
				host_chunk = c, host_base = (int)e.first_frame, host_first = (int)e.first_frame, host_end = (int)(e.first_frame + e.nframes);
				host_pending = false;
				if (cc.on_device || !need_device)
					return true;
			}
			// (also: the chunk is here, but only on the host, and its device copy is asked for - a read-back filter was switched on
			// after the read-ahead had fetched it: it is decoded again, into device memory this time)
			cached_chunk = -1, host_chunk = -1;
			if (!decode_chunk_into(cc, c, default_stream(), false, false))
				return false;
			cached_chunk = c;
			return true;
		}

		// Device address of decoded frame `pos` (RIRB files; decodes its chunk when it is not the cached one).
		int chunk_of(int pos) const
		{
			int c = (int)(pos / (int)hd.gop); // chunks hold `gop` frames except possibly the last
			if (c >= (int)index.size() || (uint64_t)pos < index[c].first_frame || (uint64_t)pos >= index[c].first_frame + index[c].nframes)
			{
				c = -1;
				for (size_t i = 0; i < index.size(); ++i)
					if ((uint64_t)pos >= index[i].first_frame && (uint64_t)pos < index[i].first_frame + index[i].nframes)
						c = (int)i;
			}
			return c;
		}
		const unsigned short *device_frame(int pos)
		{
			const int c = chunk_of(pos);
			if (c < 0 || !decode_chunk(c, true))
				return nullptr;
			return cc.d_frames.as<unsigned short>() + (size_t)(pos - (int)index[c].first_frame) * width * height;
		}

		// Unfiltered frame `pos` into host memory.
		bool read_raw(int pos, unsigned short *out, bool track = true)
		{
			const size_t npx = (size_t)width * height;
			if (kind == PCR)
				return read_at(pcr_start + pcr_transfer * (int64_t)pos, out, npx * 2);
			if (kind == ZFILE)
				return read_zfile(pos, out);
			hipStream_t st = default_stream();
			const size_t fbytes = npx * 2;
			// Sequential readers (IRMovie iteration, the usual case): once three consecutive images have been asked for, the
			// rest of the decoded chunk goes to page-locked host memory in ONE asynchronous copy and the following calls are
			// a host copy - 12 us of PCIe time per image instead of a 44 us blocking copy into pageable memory each.
			if (track)
				seq_run = (pos == last_pos + 1) ? seq_run + 1 : 0;
			auto on_host = [&] { return host_chunk >= 0 && host_chunk == cached_chunk && pos >= host_first && pos < host_end; };
			const unsigned short *d = nullptr;
			if (!on_host())
			{ // the chunk of `pos` becomes the current one - decoded now, or taken over from the read-ahead thread with its images
			  // already in page-locked memory (and then possibly nowhere else)
				const int c = chunk_of(pos);
				if (c < 0 || !decode_chunk(c, false))
					return false;
				if (!on_host())
				{
					d = device_frame(pos);
					if (!d)
						return false;
				}
			}
			if (on_host())
			{
				if (host_pending && !hip_ok(wait_stream(st), "sync"))
					return false;
				host_pending = false;
				host_copy(out, cc.h_frames.as<char>() + (size_t)(pos - host_base) * fbytes, fbytes);
				if (track && seq_run >= 2)
				{ // (no-ops when they are under way, done, or there is no such chunk)
					// (chunks of hundreds of megabytes: one lane - a lane holds a chunk twice in page-locked memory)
					const int lanes = (size_t)hd.gop * width * height * 2 <= ((size_t)256 << 20) ? kLanes : 1;
					for (int a = 1; a <= lanes; ++a)
						start_prefetch(cached_chunk + a);
				}
				return true;
			}
			if (!hip_ok(hipMemcpyAsync(out, d, fbytes, hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
				return false;
			if (host_chunk != cached_chunk)
				host_chunk = -1;
			const IndexEntry &e = index[cached_chunk];
			const int chunk_end = (int)(e.first_frame + e.nframes);
			if (track && seq_run >= 2 && host_chunk < 0 && chunk_end - (pos + 1) >= 2 && cc.h_frames.ptr)
			{ // images pos + 1 .. end of the chunk, queued behind the copy above; the next call finds them (or waits for them)
				const size_t li = (size_t)(pos + 1 - (int)e.first_frame);
				if (hip_ok(hipMemcpyAsync(cc.h_frames.as<char>() + li * fbytes, cc.d_frames.as<char>() + li * fbytes, (size_t)(chunk_end - (pos + 1)) * fbytes,
										  hipMemcpyDeviceToHost, st),
						   "D2H"))
				{
					host_chunk = cached_chunk, host_base = (int)e.first_frame, host_first = pos + 1, host_end = chunk_end;
					host_pending = true;
				}
			}
			return true;
		}

		// get_last_image_raw_value: the unfiltered last image is fetched on demand
		bool ensure_last_raw()
		{
			if (last_pos < 0)
				return false;
			if (last_raw_pos == last_pos)
				return true;
			if (!read_raw(last_pos, last_raw.data(), false))
				return false;
			last_raw_pos = last_pos;
			return true;
		}

		// IRFileLoader::readImage (IRFileLoader.cpp:1148-1247), calibration 0 = digital levels.  Decode, bad-pixel
		// repair and motion correction all run on the device; one D2H copy hands the finished frame over.
		bool read_image(int pos, int calibration, unsigned short *pixels)
		{
			if (pos < 0 || pos >= count || !pixels)
				return false;
			if (calibration != 0)
				return false; // no calibration plugin is shipped
			const bool do_bp = bp_enabled && bp_handle > 0 && global_attrs.count("Type") == 0;
			const bool do_motion = motion_enabled && !shifts.empty();
			if (!do_bp && !do_motion)
			{
				if (!read_raw(pos, pixels))
					return false;
				last_pos = pos;
				last_raw_pos = -1; // (get_last_image_raw_value fetches the image again when it is asked for: no copy per read)
				return true;
			}
			if (!device_ready() || height <= 3)
				return false;
			hipStream_t st = default_stream();
			const size_t npx = (size_t)width * height, fbytes = npx * 2;
			// A reader that goes through the movie image after image (the third in a row onwards) gets its images from the chunk filtered
			// as a whole - one batched pass of each filter over the decoded chunk, one transfer, then a host copy per read: 81-88 us an
			// image one at a time (a device copy, the kernels, an 8-byte upload and a blocking download into the caller's pageable
			// memory per image) become 25.  A reader that jumps about keeps the image-by-image way below.
			filt_run = (pos == last_pos + 1) ? filt_run + 1 : 0;
			if (kind == RIRB && filt_run >= 2 && (size_t)hd.gop * fbytes <= ((size_t)256 << 20))
			{
				const int c = chunk_of(pos);
				if (c < 0)
					return false;
				const int first = (int)index[c].first_frame, nf = (int)index[c].nframes;
				if (filt_chunk != c || filt_made_at != filt_state)
				{
					filt_chunk = -1;
					const unsigned short *d0 = device_frame(first);
					const size_t bytes = (size_t)nf * fbytes;
					if (!d0 || !filt_dev.reserve(2 * bytes) || !filt_host.reserve(bytes) || !filt_shift.reserve((size_t)nf * 8))
						return false;
					unsigned short *f_a = filt_dev.as<unsigned short>(), *f_b = f_a + (size_t)nf * npx, *res = f_a;
					// (the repair works in place: on a copy, the decoded chunk stays intact - get_last_image_raw_value reads from it)
					if (!hip_ok(hipMemcpyAsync(f_a, d0, bytes, hipMemcpyDeviceToDevice, st), "D2D"))
						return false;
					if (do_bp && rir_remove_bad_pixels_device(bp_handle, f_a, height - 3, nf, st) != 0)
						return false;
					if (do_motion)
					{
						if (!hip_ok(hipMemcpyAsync(filt_shift.ptr, &shifts[2 * (size_t)first], (size_t)nf * 8, hipMemcpyHostToDevice, st), "H2D") ||
							rir_remove_motion_device(f_a, f_b, width, height, height - 3, nf, filt_shift.as<float>(), st) != 0)
							return false;
						res = f_b;
					}
					if (!hip_ok(hipMemcpyAsync(filt_host.ptr, res, bytes, hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
						return false;
					filt_chunk = c, filt_made_at = filt_state;
				}
				host_copy(pixels, filt_host.as<char>() + (size_t)(pos - first) * fbytes, fbytes);
				last_pos = pos;
				last_raw_pos = -1;
				return true;
			}
			if (!cc.d_tmp.reserve(fbytes * 2) || !cc.d_shift.reserve(8))
				return false;
			unsigned short *d_a = cc.d_tmp.as<unsigned short>(), *d_b = d_a + npx;
			if (kind != RIRB)
			{ // raw / zstd file: the frame comes from the host
				if (!read_raw(pos, pixels) || !hip_ok(hipMemcpyAsync(d_a, pixels, fbytes, hipMemcpyHostToDevice, st), "H2D"))
					return false;
				std::memcpy(last_raw.data(), pixels, last_raw.size() * 2);
				last_raw_pos = pos;
			}
			else
			{ // the repair works in place: on a copy, the decoded chunk stays intact for the next calls
				const unsigned short *d = device_frame(pos);
				if (!d || !hip_ok(hipMemcpyAsync(d_a, d, fbytes, hipMemcpyDeviceToDevice, st), "D2D"))
					return false;
			}
			last_pos = pos;
			unsigned short *res = d_a;
			if (do_bp && rir_remove_bad_pixels_device(bp_handle, d_a, height - 3, 1, st) != 0)
				return false;
			if (do_motion)
			{
				const float sh[2] = {shifts[2 * pos], shifts[2 * pos + 1]};
				if (!hip_ok(hipMemcpyAsync(cc.d_shift.ptr, sh, sizeof(sh), hipMemcpyHostToDevice, st), "H2D") ||
					rir_remove_motion_device(d_a, d_b, width, height, height - 3, 1, cc.d_shift.as<float>(), st) != 0)
					return false;
				res = d_b;
			}
			return hip_ok(hipMemcpyAsync(pixels, res, fbytes, hipMemcpyDeviceToHost, st), "D2H") && hip_ok(wait_stream(st), "sync");
		}

		// IRFileLoader::setBadPixelsEnabled (IRFileLoader.cpp:693-716): detector on the first image, rows < H-3, once
		bool set_bad_pixels(bool enable)
		{
			if (enable && bp_handle <= 0 && count > 0 && height > 3)
			{
				if (!device_ready())
					return false;
				hipStream_t st = default_stream();
				const unsigned short *d_first = nullptr;
				if (kind != RIRB)
				{
					std::vector<unsigned short> first((size_t)width * height);
					if (!read_raw(0, first.data()) || !cc.d_tmp.reserve(first.size() * 4) ||
						!hip_ok(hipMemcpyAsync(cc.d_tmp.ptr, first.data(), first.size() * 2, hipMemcpyHostToDevice, st), "H2D") ||
						!hip_ok(wait_stream(st), "sync"))
						return false;
					d_first = cc.d_tmp.as<unsigned short>();
				}
				else
					d_first = device_frame(0); // already in HBM
				if (!d_first)
					return false;
				bp_handle = rir_bad_pixels_create_rows_device(d_first, width, height, height - 3, st);
				if (bp_handle <= 0)
					return false;
			}
			bp_enabled = enable;
			++filt_state;
			return true;
		}

		// IRFileLoader::loadTranslationFile (IRFileLoader.cpp:822-847): TSV, one header line, 4 columns, x and y in columns 1 and 2
		bool load_translation_file(const char *fname)
		{
			FILE *f = std::fopen(fname, "r");
			if (!f)
				return false;
			std::vector<float> vals;
			char line[4096];
			bool first = true, ok = true;
			std::vector<float> out;
			while (std::fgets(line, sizeof(line), f))
			{
				if (first)
				{
					first = false;
					continue;
				}
				std::istringstream ss(line);
				std::vector<float> row;
				std::string tok;
				while (ss >> tok)
				{
					std::replace(tok.begin(), tok.end(), ',', '.');
					row.push_back((float)std::atof(tok.c_str()));
				}
				if (row.empty())
					continue;
				if (row.size() != 4)
				{
					ok = false;
					break;
				}
				out.push_back(row[1]);
				out.push_back(row[2]);
			}
			std::fclose(f);
			if (!ok)
			{
				log_error("error while loading motion correction file: 4 columns expected");
				return false;
			}
			if ((int)(out.size() / 2) != count)
			{
				log_error("wrong number of images in motion correction file");
				return false;
			}
			shifts.swap(out);
			++filt_state;
			return true;
		}
	};

	// ---- ZFile writer ------------------------------------------------------------------------------
	// reference: z_open_file_write / z_write_image / z_close_file (ZFile.cpp:325-365, 483-542, 410-447) with
	// compression method 1: every image is one ZSTD_compress of the raw frame on the host (the host's libzstd),
	// so files written here are readable by a reference build that has ZFile enabled, and the reverse.
	struct ZWriterObject : public Object
	{
		const char *type_name() const override { return "ZWriter"; }
		FILE *fp = nullptr;
		std::string filename;
		ZHeader zh{};
		ZTrigger zt{};
		int clevel = 0;
		std::vector<char> buf;
		std::vector<int64_t> times, positions;

		~ZWriterObject() override
		{
			if (fp)
				close();
		}
		bool open(const char *name, int width, int height, int rate, int level)
		{
			const ZstdApi &z = ZstdApi::get();
			if (!z.ok)
			{
				log_error("open_video_write: libzstd is not available on this host");
				return false;
			}
			if (width <= 0 || height <= 0 || width >= 3000 || height >= 3000 || rate <= 0 || rate >= 1000)
			{ // the limits the readers test (ZFile.cpp:149, IRFileLoader.cpp:222)
				log_error("open_video_write: image size must be below 3000x3000 and rate in 1..999 for a ZFile");
				return false;
			}
			fp = std::fopen(name, "wb");
			if (!fp)
				return false;
			filename = name;
			clevel = level;
			zh.version = 1, zh.triggers = 1, zh.compression = 1;
			zt.rate = (uint64_t)rate;
			zt.type = 1; // continuous acquisition
			zt.nb_channels = 1;
			zt.data_format = 3; // u16
			zt.data_repetition = 1;
			zt.data_size_x = (uint64_t)width, zt.data_size_y = (uint64_t)height;
			buf.resize(z.compressBound((size_t)width * height * 2));
			return std::fwrite(&zh, sizeof(zh), 1, fp) == 1 && std::fwrite(&zt, sizeof(zt), 1, fp) == 1;
		}
		// Images are compressed on threads of this object, several at once, and go to the file in the order they came (a record's place in the
		// file depends on the sizes before it): image_write copies the image into a slot and returns; the calls that follow, and close, write
		// what has been compressed in order.  An image that could not be compressed or written fails the call that finds it out, and
		// every call after it.
		struct Slot
		{
			enum State
			{
				FREE,
				QUEUED,
				DONE,
				FAILED
			} state = FREE;
			int64_t ts = 0;
			size_t csize = 0;
			std::vector<unsigned short> img;
			std::vector<char> comp;
		};
		static constexpr int kSlots = 12;
		Slot slots[kSlots];
		uint64_t seq_in = 0, seq_out = 0; // images taken / written
		bool broken = false;
		std::mutex mu;
		std::condition_variable work, done;
		std::vector<int> todo;
		std::vector<std::thread> threads;
		bool quit = false;

		void compress_loop()
		{
			const ZstdApi &z = ZstdApi::get();
			std::unique_lock<std::mutex> lk(mu);
			for (;;)
			{
				work.wait(lk, [&] { return quit || !todo.empty(); });
				if (todo.empty())
					return; // (quit, and nothing left to do)
				Slot &sl = slots[todo.front()];
				todo.erase(todo.begin());
				lk.unlock();
				const size_t c = z.compress(sl.comp.data(), sl.comp.size(), sl.img.data(), sl.img.size() * 2, clevel);
				lk.lock();
				sl.csize = c;
				sl.state = z.isError(c) ? Slot::FAILED : Slot::DONE;
				done.notify_all();
			}
		}
		bool write_record(int64_t timestamp, const void *data, size_t c)
		{
			const int64_t pos = (int64_t)ftello(fp);
			const uint32_t csize = (uint32_t)c;
			if (std::fwrite(&timestamp, 8, 1, fp) != 1 || std::fwrite(&csize, 4, 1, fp) != 1 || std::fwrite(data, 1, c, fp) != c)
				return false;
			zt.samples++;
			times.push_back(timestamp);
			positions.push_back(pos);
			return true;
		}
		// writes the images that are compressed, in order, as far as they go (all: waits for every image taken so far); mu held
		bool commit(std::unique_lock<std::mutex> &lk, bool all)
		{
			while (seq_out < seq_in)
			{
				Slot &sl = slots[seq_out % kSlots];
				if (all)
					done.wait(lk, [&] { return sl.state != Slot::QUEUED; });
				if (sl.state == Slot::QUEUED)
					break;
				const bool ok = sl.state == Slot::DONE;
				lk.unlock();
				const bool written = ok && !broken && write_record(sl.ts, sl.comp.data(), sl.csize);
				lk.lock();
				if (!written)
					broken = true;
				sl.state = Slot::FREE;
				++seq_out;
			}
			return !broken;
		}
		bool add(const unsigned short *img, int64_t timestamp)
		{
			if (!fp || !img || broken)
				return false;
			const size_t npx = (size_t)zt.data_size_x * zt.data_size_y;
			const int nthreads = zfile_threads();
			if (nthreads == 0)
			{
				const ZstdApi &z = ZstdApi::get();
				const size_t c = z.compress(buf.data(), buf.size(), img, npx * 2, clevel);
				if (z.isError(c) || !write_record(timestamp, buf.data(), c))
					return broken = true, false;
				return true;
			}
			std::unique_lock<std::mutex> lk(mu);
			Slot &sl = slots[seq_in % kSlots];
			if (sl.state != Slot::FREE && !commit(lk, false))
				return false;
			if (sl.state != Slot::FREE)
			{ // every slot holds an image that is not written yet: the oldest one first
				Slot &oldest = slots[seq_out % kSlots];
				done.wait(lk, [&] { return oldest.state != Slot::QUEUED; });
				if (!commit(lk, false))
					return false;
			}
			if (sl.img.size() != npx)
				sl.img.resize(npx), sl.comp.resize(buf.size());
			lk.unlock();
			host_copy(sl.img.data(), img, npx * 2); // (the slot is free and nobody else looks at a free slot)
			lk.lock();
			sl.ts = timestamp, sl.state = Slot::QUEUED;
			todo.push_back((int)(seq_in % kSlots));
			++seq_in;
			if (threads.empty())
				for (int t = 0; t < nthreads; ++t)
					threads.emplace_back([this] { compress_loop(); });
			work.notify_one();
			return commit(lk, false);
		}
		void stop_threads()
		{
			{
				std::unique_lock<std::mutex> lk(mu);
				quit = true;
			}
			work.notify_all();
			for (auto &t : threads)
				if (t.joinable())
					t.join();
			threads.clear();
		}
		// size of the image data (the trailer comes after it), as z_close_file returns
		int64_t close()
		{
			if (!fp)
				return -1;
			{
				std::unique_lock<std::mutex> lk(mu);
				(void)commit(lk, true); // every image taken is in the file (or the writer is broken)
			}
			stop_threads();
			const int64_t data_end = (int64_t)ftello(fp);
			bool ok = fseeko(fp, sizeof(ZHeader), SEEK_SET) == 0 && std::fwrite(&zt, sizeof(zt), 1, fp) == 1 && fseeko(fp, 0, SEEK_END) == 0;
			AttrMap global;
			global["positions"] = std::string(reinterpret_cast<const char *>(positions.data()), positions.size() * 8);
			const std::string trailer = FileAttributes::serialize(global, std::vector<AttrMap>(times.size()), times);
			ok = ok && std::fwrite(trailer.data(), 1, trailer.size(), fp) == trailer.size();
			ok = (std::fclose(fp) == 0) && ok && !broken;
			fp = nullptr;
			return ok ? data_end : -1;
		}
	};

	std::shared_ptr<CameraObject> camera(int h) { return lookup_as<CameraObject>(h); }
	std::shared_ptr<SaverObject> saver(int h) { return lookup_as<SaverObject>(h); }

	int format_of(const CameraObject &c)
	{
		return c.kind == CameraObject::PCR ? c.raw_format : c.kind == CameraObject::ZFILE ? FILE_FORMAT_ZSTD_COMPRESSED : FILE_FORMAT_H264;
	}

	int kv_out(const AttrMap &m, int index, char *key, int *key_len, char *value, int *value_len, bool global)
	{
		if (index < 0 || index >= (int)m.size() || !key_len || !value_len)
			return -1;
		auto it = m.begin();
		std::advance(it, index);
		const int s1 = (int)it->first.size(), s2 = (int)it->second.size();
		const int oldk = *key_len, oldv = *value_len;
		if (global)
		{ // video_io.cpp:624-641: the key buffer needs room for the terminator
			if (s1 + 1 > *key_len || s2 > *value_len)
			{
				*key_len = s1 + 1;
				*value_len = s2;
				return -2;
			}
		}
		else if (s1 > *key_len || s2 > *value_len)
		{ // video_io.cpp:573-582
			*key_len = s1;
			*value_len = s2;
			return -2;
		}
		*key_len = s1;
		*value_len = s2;
		std::memcpy(key, it->first.data(), s1);
		std::memcpy(value, it->second.data(), s2);
		if (global || oldk > s1)
			key[s1] = 0;
		if (oldv > s2)
			value[s2] = 0;
		return 0;
	}
} // namespace

// =====================================================================================================
// loader entry points
// =====================================================================================================

static int register_camera(std::shared_ptr<CameraObject> cam, int *file_format)
{
	if (!cam->open_common())
		return 0;
	if (file_format)
		*file_format = format_of(*cam);
	return register_object(cam);
}

// video_io.cpp:16-49: handle > 0, or 0 on failure (file_format set to 0)
RIR_EXPORT int open_camera_file(const char *filename, int *file_format)
{
	if (file_format)
		*file_format = 0;
	int h = 0;
	std::string name = filename ? filename : "";
	try
	{
		auto cam = std::make_shared<CameraObject>();
		cam->filename = name;
		cam->fp = filename ? std::fopen(filename, "rb") : nullptr;
		h = cam->fp ? register_camera(cam, file_format) : 0;
	}
	catch (const std::exception &e)
	{ // nothing may be thrown across the C boundary
		log_error(std::string("open_camera_file: ") + e.what());
		h = 0;
	}
	if (h <= 0)
	{
		log_error("Unable to open camera file " + name + ": wrong file format");
		return 0;
	}
	return h;
}

// video_io.cpp:110-145 (the bytes are copied: the caller may release its buffer)
RIR_EXPORT int open_camera_from_memory(void *ptr, int64_t size, int *file_format)
{
	if (file_format)
		*file_format = 0;
	int h = 0;
	try
	{
		auto cam = std::make_shared<CameraObject>();
		if (ptr && size > 0)
			cam->mem.assign(static_cast<char *>(ptr), static_cast<char *>(ptr) + size);
		h = cam->mem.empty() ? 0 : register_camera(cam, file_format);
	}
	catch (const std::exception &e)
	{
		log_error(std::string("open_camera_from_memory: ") + e.what());
		h = 0;
	}
	if (h <= 0)
	{
		log_error("Unable to open camera file: wrong file format");
		return 0;
	}
	return h;
}

// video_io.cpp:74-108 takes a reference FileReader object; none can exist outside the reference library
RIR_EXPORT int open_camera_file_reader(void *, int *file_format)
{
	if (file_format)
		*file_format = 0;
	log_error("open_camera_file_reader: file reader objects are not supported, use open_camera_file or open_camera_from_memory");
	return 0;
}

// video_io.cpp:51-72
RIR_EXPORT int video_file_format(const char *filename)
{
	try
	{
		auto cam = std::make_shared<CameraObject>();
		cam->filename = filename ? filename : "";
		cam->fp = filename ? std::fopen(filename, "rb") : nullptr;
		if (!cam->fp || !cam->open_common())
			return -1;
		return format_of(*cam);
	}
	catch (const std::exception &e)
	{
		log_error(std::string("video_file_format: ") + e.what());
		return -1;
	}
}

RIR_EXPORT int close_camera(int cam)
{
	if (!camera(cam))
	{
		log_error("close_camera: NULL camera");
		return -1;
	}
	remove_object(cam);
	return 0;
}

RIR_EXPORT int get_image_count(int cam)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("get_image_count: NULL camera");
		return -1;
	}
	return c->count;
}

RIR_EXPORT int get_image_time(int cam, int pos, int64_t *time)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("get_image_time: NULL camera");
		return -1;
	}
	if (pos < 0 || pos >= (int)c->times.size() || !time)
	{
		log_error("get_image_time: position out of range");
		return -1;
	}
	*time = c->times[pos];
	return 0;
}

RIR_EXPORT int get_image_size(int cam, int *width, int *height)
{
	auto c = camera(cam);
	if (!c || !width || !height)
	{
		log_error("get_image_size: NULL camera");
		return -1;
	}
	*width = c->width;
	*height = c->height;
	return 0;
}

RIR_EXPORT int get_filename(int cam, char *filename)
{
	auto c = camera(cam);
	if (!c || !filename)
	{
		log_error("get_filename: NULL camera");
		return -1;
	}
	std::string f = c->filename.substr(0, UNSPECIFIED_CHAR_LENGTH - 1);
	std::memset(filename, 0, UNSPECIFIED_CHAR_LENGTH);
	std::memcpy(filename, f.data(), f.size());
	return 0;
}

// only "Digital Level": no calibration plugin is shipped with the reference either (BaseCalibration.cpp:7-43)
RIR_EXPORT int supported_calibrations(int cam, int *count)
{
	if (!camera(cam) || !count)
	{
		log_error("support_calibration: NULL camera");
		return -1;
	}
	*count = 1;
	return 0;
}
RIR_EXPORT int calibration_name(int cam, int calibration, char *name)
{
	if (!camera(cam) || !name)
	{
		log_error("support_calibration: NULL camera");
		return -1;
	}
	if (calibration != 0)
	{
		log_error("calibration_name: calibration index out of range");
		return -1;
	}
	std::memcpy(name, "Digital Level", 13);
	return 0;
}

// video_io.cpp:361-375
RIR_EXPORT int load_image(int cam, int pos, int calibration, unsigned short *pixels)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("load_image: NULL camera");
		return -1;
	}
	return guarded("load_image", -1, [&] { return c->read_image(pos, calibration, pixels) ? 0 : -1; });
}

// Extension: images first .. first + count - 1 of a recording of this library go into a saver of the same geometry WITHOUT leaving the device:
// a chunk is decoded into device memory, its frames are copied device-to-device into the chunk the saver assembles, with their per-image
// attributes (keep_attributes != 0; else without any) and the given time stamps (IRMovie.to_h264 / split_rush: 6 us an image instead of the 30 of reading each image into host
// memory and recording it from there).  Returns `count`; -2 when this way is not open - another kind of file, another geometry, a
// read-back filter switched on (the images would have to pass through it) - and the caller goes image by image; -1 on failure.
RIR_EXPORT int rir_transcode_images(int cam, int file, int first, int count, const int64_t *timestamps_ns, int keep_attributes)
{
	auto c = camera(cam);
	auto s = saver(file);
	if (!c || !s)
	{
		log_error("rir_transcode_images: NULL camera or saver");
		return -1;
	}
	if (first < 0 || count < 0 || (count > 0 && !timestamps_ns) || (int64_t)first + count > c->count)
		return -1;
	return guarded("rir_transcode_images", -1, [&] {
		const bool filtered = (c->bp_enabled && c->bp_handle > 0 && c->global_attrs.count("Type") == 0) || (c->motion_enabled && !c->shifts.empty());
		if (c->kind != CameraObject::RIRB || c->width != s->width || c->height != s->height || filtered)
			return -2;
		const size_t npx = (size_t)c->width * c->height;
		const bool diag = std::getenv("RIR_TRANSCODE_DIAG") != nullptr;
		const std::vector<AttrMap> none(keep_attributes ? 0 : (size_t)std::max(1, (int)c->hd.gop)); // (a run never exceeds a chunk of the source)
		double t_dec = 0, t_add = 0;
		auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		int done = 0;
		while (done < count)
		{
			const int pos = first + done;
			const int ch = c->chunk_of(pos);
			if (ch < 0)
				return -1;
			const int run = std::min(count - done, (int)(c->index[ch].first_frame + c->index[ch].nframes) - pos);
			const double t0 = diag ? now() : 0;
			const unsigned short *d = c->device_frame(pos);
			if (!d)
				return -1;
			const double t1 = diag ? now() : 0;
			for (int k = 0; k < run;)
			{
				const int took = s->add_images_device(d + (size_t)k * npx, run - k, timestamps_ns + done + k,
													  keep_attributes ? &c->frame_attrs[(size_t)pos + k] : none.data());
				if (took <= 0)
					return -1;
				k += took;
			}
			if (diag)
				t_dec += t1 - t0, t_add += now() - t1;
			done += run;
		}
		if (diag)
			fprintf(stderr, "rir_transcode_images: %d images, chunks into device memory %.0f us, into the saver %.0f us\n", count, t_dec, t_add);
		if (count > 0)
		{
			c->last_pos = first + count - 1;
			c->last_raw_pos = -1;
		}
		return count;
	});
}

// video_io.cpp:377-391: the uint16 image cast to float (IRVideoLoader.h:109-117)
RIR_EXPORT int load_imageF(int cam, int pos, int calibration, float *pixels)
{
	auto c = camera(cam);
	if (!c || !pixels)
	{
		log_error("load_image: NULL camera");
		return -1;
	}
	return guarded("load_imageF", -1, [&] {
		std::vector<unsigned short> tmp((size_t)c->width * c->height);
		if (!c->read_image(pos, calibration, tmp.data()))
			return -1;
		for (size_t i = 0; i < tmp.size(); ++i)
			pixels[i] = (float)tmp[i];
		return 0;
	});
}

RIR_EXPORT int get_last_image_raw_value(int cam, int x, int y, unsigned short *value)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("get_last_image_raw_value: NULL camera");
		return -1;
	}
	if (!value || x < 0 || y < 0 || x >= c->width || y >= c->height || !c->ensure_last_raw())
		return -1;
	*value = c->last_raw[(size_t)y * c->width + x];
	return 0;
}

RIR_EXPORT int enable_bad_pixels(int cam, int enable)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("enable_bad_pixels: NULL camera");
		return -1;
	}
	return c->set_bad_pixels(enable != 0) ? 0 : -1;
}
RIR_EXPORT int bad_pixels_enabled(int cam)
{
	auto c = camera(cam);
	return c ? (c->bp_enabled ? 1 : 0) : 0;
}
RIR_EXPORT int load_motion_correction_file(int cam, const char *filename)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("load_motion_correction_file: NULL camera");
		return -1;
	}
	if (!filename || !c->load_translation_file(filename))
	{
		log_error("unable to load file");
		return -1;
	}
	return 0;
}
RIR_EXPORT int enable_motion_correction(int cam, int enable)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("enable_motion_correction: NULL camera");
		return -1;
	}
	c->motion_enabled = enable != 0;
	++c->filt_state;
	return 0;
}
RIR_EXPORT int motion_correction_enabled(int cam)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("motion_correction_enabled: NULL camera");
		return 0;
	}
	return c->motion_enabled ? 1 : 0;
}

// attributes of the last read image / of the file (video_io.cpp:538-642)
RIR_EXPORT int get_attribute_count(int cam)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("get_attribute_count: NULL camera");
		return -1;
	}
	return (c->last_pos >= 0 && c->last_pos < (int)c->frame_attrs.size()) ? (int)c->frame_attrs[c->last_pos].size() : 0;
}
RIR_EXPORT int get_attribute(int cam, int index, char *key, int *key_len, char *value, int *value_len)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("get_attribute: NULL camera");
		return -1;
	}
	if (c->last_pos < 0 || c->last_pos >= (int)c->frame_attrs.size())
		return -1;
	return kv_out(c->frame_attrs[c->last_pos], index, key, key_len, value, value_len, false);
}
RIR_EXPORT int get_global_attribute_count(int cam)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("get_global_attribute_count: NULL camera");
		return -1;
	}
	return (int)c->global_attrs.size();
}
RIR_EXPORT int get_global_attribute(int cam, int index, char *key, int *key_len, char *value, int *value_len)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("get_global_attribute: NULL camera");
		return -1;
	}
	return kv_out(c->global_attrs, index, key, key_len, value, value_len, true);
}

// ---- calibration / emissivity: no calibration object exists (reference behaviour without plugins) ----
RIR_EXPORT int support_emissivity(int cam)
{ // video_io.cpp:340-349: -1 when there is no calibration
	(void)cam;
	log_error("support_emissivity: NULL camera");
	return -1;
}
// The emissivity map is STATE of the loader whether or not a calibration uses it (IRVideoLoader.h:29-97: a vector of inverse
// emissivities + the global value; IRFileLoader.cpp:1099-1111 forwards to it and answers true): what is set is what is read back
// (tests/python/test_video_io.py:195-205 asserts that on a plain recording).  Host-side book-keeping, no pixel is touched.
RIR_EXPORT int set_global_emissivity(int cam, float emi)
{
	if (emi < 0.f || emi > 1.f)
	{
		log_error("set_emissivity: wrong emissivity value");
		return -1;
	}
	auto c = camera(cam);
	if (!c)
	{
		log_error("set_global_emissivity: NULL camera");
		return -1;
	}
	if (emi != c->global_emi || c->inv_emi.empty())
	{ // IRVideoLoader::setEmissivity
		c->inv_emi.assign((size_t)std::max(c->width, 0) * (size_t)std::max(c->height, 0), 1.f / emi);
		c->global_emi = emi;
	}
	return 0;
}
RIR_EXPORT int set_emissivity(int cam, float *emi, int size)
{
	auto c = camera(cam);
	if (!c)
	{
		log_error("set_emissivity: NULL camera");
		return -1;
	}
	if (size > 0 && emi)
	{ // IRVideoLoader::setEmissivities: the first `size` pixels, 1 for the rest
		const size_t npx = (size_t)std::max(c->width, 0) * (size_t)std::max(c->height, 0), n = std::min((size_t)size, npx);
		c->inv_emi.resize(npx);
		for (size_t i = 0; i < n; ++i)
			c->inv_emi[i] = 1.f / emi[i];
		for (size_t i = n; i < npx; ++i)
			c->inv_emi[i] = 1.f;
		c->global_emi = 0;
	}
	return 0; // (the file loader answers true whatever the size, IRFileLoader.cpp:1105-1111)
}
RIR_EXPORT int get_emissivity(int cam, float *emi, int size)
{ // video_io.cpp:319-338: as many values as asked for and stored; 1 in the first place when nothing is stored; returns the count
	auto c = camera(cam);
	if (!c || !emi)
	{
		log_error("get_emissivity: NULL camera");
		return -1;
	}
	const int s = std::max(0, std::min(size, (int)std::min<size_t>(c->inv_emi.size(), 0x7fffffff)));
	for (int i = 0; i < s; ++i)
		emi[i] = 1.f / c->inv_emi[(size_t)i];
	if (s == 0)
		*emi = 1;
	return s;
}
RIR_EXPORT int calibrate_inplace(int cam, unsigned short *, int, int calibration)
{
	if (!camera(cam))
	{
		log_error("load_image: NULL camera");
		return -1;
	}
	return calibration == 0 ? 0 : -1;
}
RIR_EXPORT int calibrate_image_inplace(int cam, unsigned short *img, int size, int calib) { return calibrate_inplace(cam, img, size, calib); }
RIR_EXPORT int calibrate_image(int cam, unsigned short *img, float *out, int size, int calib)
{
	if (!camera(cam) || !img || !out)
	{
		log_error("calibrate_image: NULL camera");
		return -1;
	}
	if (calib != 0)
		return -1;
	for (int i = 0; i < size; ++i)
		out[i] = (float)img[i];
	return 0;
}
RIR_EXPORT int camera_saturate(int cam)
{
	if (!camera(cam))
	{
		log_error("camera_saturate: NULL camera");
		return -1;
	}
	return 0;
}
RIR_EXPORT int calibration_files(int cam, char *, int *) { return camera(cam) ? -1 : -1; }
RIR_EXPORT int flip_camera_calibration(int cam, int, int) { return camera(cam) ? -2 : -1; } // -2: no calibration (video_io.cpp:266-267)
RIR_EXPORT int get_table_names(int cam, char *, int *dst_size)
{
	if (!camera(cam) || !dst_size)
	{
		log_error("get_table_names: NULL identifier");
		return -1;
	}
	*dst_size = 0;
	return 0;
}
RIR_EXPORT int get_table(int cam, const char *, float *, int *)
{
	if (!camera(cam))
		log_error("get_table_names: NULL identifier");
	return -1;
}

// =====================================================================================================
// saver entry points (video_io.cpp:646-843)
// =====================================================================================================

RIR_EXPORT void set_ffmpeg_log_enabled(int) {} // there is no ffmpeg underneath

// returns the identifier (> 0); 0 on error
RIR_EXPORT int h264_open_file(const char *filename, int width, int height, int lossy_height)
{
	if (!filename)
		return 0;
	auto s = std::make_shared<SaverObject>();
	s->filename = filename;
	s->width = width, s->height = height, s->lossy_height = lossy_height;
	if (file_exists(filename) && std::remove(filename) != 0)
	{
		log_error("h264_open_file: cannot remove output file");
		return 0;
	}
	s->start_warmup();
	return register_object(s);
}

RIR_EXPORT void h264_close_file(int file)
{
	auto s = saver(file);
	if (!s)
	{
		log_error("h264_close_file: NULL identifier");
		return;
	}
	guarded("h264_close_file", 0, [&] {
		s->close();
		return 0;
	});
	remove_object(file);
}

RIR_EXPORT int h264_set_parameter(int file, const char *param, const char *value)
{
	auto s = saver(file);
	if (!s)
	{
		log_error("h264_set_parameter: NULL identifier");
		return -1;
	}
	return s->set_parameter(param, value) ? 0 : -1;
}

RIR_EXPORT int h264_set_global_attributes(int file, int attribute_count, char *keys, int *key_lens, char *values, int *value_lens)
{
	auto s = saver(file);
	if (!s || attribute_count < 0)
	{
		log_error("h264_set_global_attributes: NULL identifier");
		return -1;
	}
	s->global_attrs = attr_map_from_c(attribute_count, keys, key_lens, values, value_lens);
	return 0;
}

RIR_EXPORT int h264_add_image_lossless(int file, unsigned short *img, int64_t timestamps_ns, int attribute_count, char *keys, int *key_lens,
									   char *values, int *value_lens)
{
	auto s = saver(file);
	if (!s || attribute_count < 0)
	{
		log_error("h264_add_image_lossless: NULL identifier");
		return -1;
	}
	return guarded("h264_add_image_lossless", -1,
				   [&] { return s->add_image(img, timestamps_ns, attr_map_from_c(attribute_count, keys, key_lens, values, value_lens)) ? 0 : -1; });
}

// Bounded-loss recording: reference H264_Saver::addImageLossy (h264.cpp:2038-2046) without an input
// camera, i.e. addImageLossyNoCamera (:2253-2424) - the only variant reachable without a calibration plugin.
RIR_EXPORT int h264_add_image_lossy(int file, unsigned short *img_DL, int64_t timestamps_ns, int attribute_count, char *keys, int *key_lens,
									char *values, int *value_lens)
{
	auto s = saver(file);
	if (!s || attribute_count < 0)
	{
		log_error("h264_add_image_lossy: NULL identifier");
		return -1;
	}
	return guarded("h264_add_image_lossy", -1,
				   [&] { return s->add_image_lossy(img_DL, timestamps_ns, attr_map_from_c(attribute_count, keys, key_lens, values, value_lens)) ? 0 : -1; });
}

// h264_add_loss (video_io.cpp:789-806): adds the loss to the caller's image without writing it
RIR_EXPORT int h264_add_loss(int file, unsigned short *img)
{
	auto s = saver(file);
	if (!s)
	{
		log_error("h264_add_loss: NULL identifier");
		return -1;
	}
	return guarded("h264_add_loss", -1, [&] { return s->add_loss(img) ? 0 : -1; });
}

// ---- bounded-loss step on a device-resident stream ---------------------------------------------------------------
// The loss injection of H264_Saver::addImageLossyNoCamera / addLoss (h264.cpp:2253-2607) without the saver around it:
// frames in HBM in, frames in HBM out, one state object per stream.
RIR_EXPORT int rir_lossy_create(int width, int height, int lossy_height, int low_value_error, int high_value_error, double std_factor,
								int running_average, int subtract_min, int remove_bad_pixels)
{
	if (!device_ready())
		return 0;
	if (width <= 0 || height <= 0 || lossy_height < 0)
	{
		log_error("rir_lossy_create: invalid argument");
		return 0;
	}
	auto o = std::make_shared<LossyObject>();
	o->low = low_value_error, o->high = high_value_error, o->std_factor = std_factor, o->remove_bad_pixels = remove_bad_pixels != 0;
	if (!o->st.prepare(width, height, lossy_height, running_average, subtract_min != 0))
		return 0;
	return register_object(o);
}

// A parameter change on a stream in use (H264_Saver::setParameter lowValueError / highValueError / stdFactor, h264.cpp:1709-1781): applies
// from the next frame on; the history the budget is computed from (firstStdDevs, the 40-frame window) stays.
RIR_EXPORT int rir_lossy_set_errors(int handle, int low_value_error, int high_value_error, double std_factor)
{
	auto o = lookup_as<LossyObject>(handle);
	if (!o)
	{
		log_error("rir_lossy_set_errors: invalid handle");
		return -1;
	}
	o->low = low_value_error, o->high = high_value_error, o->std_factor = std_factor;
	return 0;
}

// d_in, d_out: uint16 [nframes][height][width], distinct buffers; low_errors / high_errors: HOST int[nframes] (may be NULL).
// Frames are processed in order (the state is sequential); add_loss != 0 selects the addLoss variant.
RIR_EXPORT int rir_lossy_step_device(int handle, const unsigned short *d_in, unsigned short *d_out, int nframes, int add_loss, int *low_errors,
									 int *high_errors, void *stream)
{
	auto o = lookup_as<LossyObject>(handle);
	if (!o || !d_in || !d_out || d_in == d_out || nframes <= 0)
	{
		log_error("rir_lossy_step_device: invalid argument");
		return -1;
	}
	{
		const size_t span = (size_t)nframes * (size_t)o->st.w * (size_t)o->st.h;
		if (d_in < d_out + span && d_out < d_in + span)
		{
			log_error("rir_lossy_step_device: input and output frames overlap");
			return -1;
		}
	}
	// a batch without bad-pixel repair is a run of frames: one launch per frame instead of three (rir_lossy_step_multi_device)
	if (!o->remove_bad_pixels && nframes >= 3)
	{
		const unsigned short *in1[1] = {d_in};
		unsigned short *out1[1] = {d_out};
		return rir_lossy_step_multi_device(&handle, 1, in1, out1, nframes, add_loss, low_errors, high_errors, stream);
	}
	// every frame is queued without waiting (statistics, budget and update all run on the device); the budgets of the batch
	// come back in one copy at the end - or not at all when the caller does not ask for them: then nothing here waits
	const size_t npx = (size_t)o->st.w * o->st.h;
	hipStream_t st = (hipStream_t)stream;
	const bool want = low_errors || high_errors;
	if (want && !o->batch_errs.reserve((size_t)nframes * 2 * sizeof(int)))
		return -1;
	for (int i = 0; i < nframes; ++i)
		if (!o->st.queue_frame(d_in + (size_t)i * npx, d_out + (size_t)i * npx, add_loss != 0, o->remove_bad_pixels, o->low, o->high, o->std_factor,
							   want ? o->batch_errs.as<int>() + 2 * i : nullptr, st))
			return -1;
	if (want)
	{
		std::vector<int> e((size_t)nframes * 2);
		if (!hip_ok(hipMemcpyAsync(e.data(), o->batch_errs.ptr, e.size() * sizeof(int), hipMemcpyDeviceToHost, st), "D2H") ||
			!hip_ok(wait_stream(st), "sync"))
			return -1;
		for (int i = 0; i < nframes; ++i)
		{
			if (low_errors)
				low_errors[i] = e[2 * i];
			if (high_errors)
				high_errors[i] = e[2 * i + 1];
		}
	}
	return 0;
}

// The same step for `nstreams` INDEPENDENT streams (one state object each, equal geometry and history length) with the streams
// sharing every launch: frame f of all streams = three launches whose grids carry the stream in their second dimension
// (the loss state is sequential in time, so streams - not frames - are what runs side by side).
// d_in[i] / d_out[i]: uint16 [nframes][h][w] of stream i (HOST arrays of nstreams device pointers); low_errors / high_errors:
// HOST int[nstreams][nframes] or NULL (then nothing waits).
// The step of `nframes` frames for the streams os[0..nstreams) (checked by the callers: equal geometry and history length, no
// bad-pixel repair).  d_errs: NULL, or per stream a DEVICE int[nframes][2] that receives the budgets (nothing is read back
// then); otherwise low_errors / high_errors: HOST int[nstreams][nframes] or NULL.
namespace
{
int lossy_step_streams(LossyObject *const *os, int nstreams, const unsigned short *const *d_in, unsigned short *const *d_out, int nframes, int add_loss,
					   int *const *d_errs, int *low_errors, int *high_errors, hipStream_t st, bool force_per_frame)
{
	const size_t npx = (size_t)os[0]->st.w * os[0]->st.h;
	const bool want = low_errors || high_errors;
	LossyObject &lead = *os[0]; // owns the scratch of the call: the table of steps and the budgets
	for (int i = 0; i < nstreams; ++i)
		if (os[i]->is_failed())
		{
			log_error("bounded-loss step: a run of frames of this stream gave up earlier - its state is invalid, destroy the stream");
			return -1;
		}
	if (want && !d_errs && !lead.batch_errs.reserve((size_t)nstreams * nframes * 2 * sizeof(int)))
		return -1;
	// where the budget of frame f of stream i goes on the device: the caller's array, the call's own (read back below), or nowhere
	auto errs_of = [&](int i, int f) {
		return d_errs ? d_errs[i] + (size_t)f * 2 : want ? lead.batch_errs.as<int>() + ((size_t)i * nframes + f) * 2 : (int *)nullptr;
	};
	int f0 = 0;
	if (os[0]->st.frames == 0)
	{ // first frame of every stream: stored as it is, seeds the state - one small launch per stream, once in a stream's life
		for (int i = 0; i < nstreams; ++i)
			if (!os[i]->st.queue_frame(d_in[i], d_out[i], add_loss != 0, false, os[i]->low, os[i]->high, os[i]->std_factor,
									   errs_of(i, 0), st))
				return -1;
		f0 = 1;
	}
	const int nsteps = nframes - f0;
	// resident launches of this call (resident_device.h): their epochs group by group, and the host side of every stream's state as it
	// was before each group - what a launch that did not become resident (it has written nothing) is repeated from
	struct Snap
	{
		int ra_count, ra_head;
		int64_t frames;
	};
	std::vector<unsigned int> run_epochs;	// [ngroups]: epoch of the group's first launch
	std::vector<Snap> snaps;				// [ngroups][nstreams]
	int run_group = 0, run_launches_per_group = 0;
	const int s_px = os[0]->st.w * os[0]->st.hl, full_px = os[0]->st.w * os[0]->st.h;
	// runs: one launch per frame (lossy_frame_kernel) + one histogram launch per group of frames; needs whole groups of 8 pixels
	const bool runs = nsteps >= 2 && s_px > 0 && s_px % 8 == 0 && full_px % 8 == 0;
	if (nsteps > 0)
	{
		lead.const_groups = 0, lead.spec_groups = 0; // (the books of rir_lossy_path_stats / rir_lossy_spec_stats: of THIS call, whatever path it takes)
		// the descriptions of all launches of the call go to the device in one copy (page-locked staging: the copy is asynchronous
		// and the host buffer must outlive it - it is kept by the leading stream's object)
		// persistent: the whole group of frames in one launch (lossy_run_kernel) when the chip holds a stream's workgroups at once
		const int run_wgs = lossy_run_workgroups(full_px);
		bool errors_fit = true; // (the run kernel hands a decision on as two 24-bit fields)
		for (int i = 0; i < nstreams; ++i)
			errors_fit = errors_fit && os[i]->low < (1 << 24) && os[i]->high < (1 << 24);
		// (more streams than the chip holds at once go through the resident kernel a batch of streams after the other: 6 streams of
		// 640x512 per launch keep 0.49 M frames/s whatever the number of streams; a launch per frame for all of them does 0.30-0.43 M)
		const char *max_env = getenv("RIR_LOSSY_RUN_MAX_WORKGROUPS"); // (tests: a smaller limit, to go through the batches with small frames)
		const int capacity = lossy_run_capacity();						 // what THIS device holds of the run kernel at once (0: unknown)
		const int max_wgs = max_env && atoi(max_env) > 0 ? std::min(atoi(max_env), capacity) : capacity;
		// the run kernel has a second form (part of the pixel state parked in LDS, one more wave per SIMD): more streams per launch, each
		// a little slower - taken when it saves a launch (runtime.h; RIR_LOSSY_RUN_FORM=5 / 6: one form only, for tests and measurements)
		const int cap6 = lossy_run_capacity(true), max6 = max_env && atoi(max_env) > 0 ? std::min(atoi(max_env), cap6) : cap6;
		const char *form = getenv("RIR_LOSSY_RUN_FORM");
		bool parked = false;
		ResidentPlan plan = form ? resident_plan(atoi(form) == 6 ? max6 : max_wgs, run_wgs, nstreams) : resident_plan_two_forms(max_wgs, max6, run_wgs, nstreams, &parked);
		if (form)
			parked = atoi(form) == 6; // (units_per_launch 0: a stream does not fit - launch per frame)
		const bool persistent = runs && errors_fit && plan.units_per_launch > 0 && !force_per_frame && !getenv("RIR_LOSSY_LAUNCH_PER_FRAME");
		// streams per launch of the resident kernel: the launches the plan needs, filled evenly (32 streams at 9 per launch: 8, 8, 8, 8 - not 9, 9, 9, 5)
		const int batch = persistent ? resident_batch(plan, nstreams) : nstreams;
		// The constant-budget form (lossy_kernels.hip: lossy_const_run_kernel): with stdFactor == 0 for every stream of the call each group is
		// first offered to an ordinary launch that needs no hand-offs between workgroups (decided below, on the device, group by group).
		bool const_form = persistent && !getenv("RIR_LOSSY_NO_CONST");
		{ // (only in the build with the test hooks: the forms of the kernel that small test frames would never take)
			const char *pairs = test_hook("RIR_LOSSY_CONST_PAIRS");
			lossy_const_force_pairs(pairs ? atoi(pairs) : 0);
		}
		for (int i = 0; i < nstreams; ++i)
			const_form = const_form && os[i]->std_factor == 0.0;
		// The speculative form (lossy_kernels.h: LossySpec): budgets that follow the statistics (stdFactor != 0, the reference's default) are guessed -
		// the configured errors, what a scene that does not move gets - and verified on the device; a group whose guess does not verify within
		// `spec_passes` corrections is left to the resident kernel, exactly as a declined constant-budget group is.
		bool spec_form = persistent && !const_form && !getenv("RIR_LOSSY_NO_SPEC");
		for (int i = 0; i < nstreams; ++i)
			spec_form = spec_form && os[i]->low <= 65535 && os[i]->high <= 65535; // (the budget table holds 16-bit fields)
		int spec_passes = 3;
		if (const char *pe = getenv("RIR_LOSSY_SPEC_PASSES"))
			spec_passes = std::max(1, std::min(16, atoi(pe)));
		if (getenv("RIR_LOSSY_SPEC_FIRST_ONLY")) // (comparison: a pass corrects the first wrong budget only - bit 8 of the launch's pass count)
			spec_passes |= 256;
		if (getenv("RIR_LOSSY_SPEC_NO_GIVE_UP")) // (measurements: every pass that is allowed is made)
			spec_passes |= 512;
		const bool stream_form = const_form || spec_form; // a streaming kernel steps whole groups: long ones
		// frames per histogram launch (one 64 KB histogram slice per frame and stream).  A group costs four small launches beside its
		// frames, so streams that may take the constant-budget form - 0.1 us per stream-frame - get groups four times as long (up to
		// 512 MB of slices, allocated as needed; 32 streams x 200 frames: 64-frame groups 0.83 M frames/s, one group 1.2 M)
		int group = std::min(kLossyConstMaxFrames, std::max(1, (persistent ? (stream_form ? 8192 : 2048) : 512) / nstreams));
		// (the constant-budget kernel addresses a group's frames with 32-bit offsets below 2^31: groups of large frames are cut to fit -
		// 64 = the longest ring, 8 = the most frames it keeps in flight; lossy_const_run_kernel decides for itself, this only keeps it fast)
		if (stream_form)
			group = (int)std::max<long long>(1, std::min<long long>(group, 0x7fffffffll / ((long long)npx * 2) - 64 - 8));
		const int ngroups = runs ? (nsteps + group - 1) / group : 0;
		const size_t nfused = persistent ? 0 : runs ? (size_t)(nsteps + 1) * nstreams : (size_t)nsteps * nstreams, nbg = runs ? (size_t)nsteps * nstreams : 0,
					 nhist = persistent ? 0 : nbg; // (the resident / constant-budget path takes its backgrounds straight from the runs' descriptions: no per-frame table)
		const size_t run_off = (nfused + nhist) * sizeof(LossyStep); // (a multiple of 8)
		const size_t spec_off = run_off + (persistent ? (size_t)ngroups * nstreams * sizeof(LossyRun) : 0);
		static_assert(sizeof(LossyRun) % 8 == 0 && sizeof(LossySpec) % 8 == 0, "the tables of a call lie behind each other in one buffer");
		const size_t nb = spec_off + (persistent && spec_form ? (size_t)ngroups * nstreams * sizeof(LossySpec) : 0);
		if (!lead.multi_table.reserve(nb) || !lead.multi_stage.reserve(nb))
			return -1;
		// an earlier call's copy out of the staging buffer may still be in flight - on whatever stream that call was given
		if (lead.multi_copied && !hip_ok(hipEventSynchronize(lead.multi_copied), "event"))
			return -1;
		if (!lead.multi_copied && !hip_ok(hipEventCreateWithFlags(&lead.multi_copied, hipEventDisableTiming), "event"))
			return -1;
		LossyStep *hs = lead.multi_stage.as<LossyStep>();
		const LossyStep *dt = lead.multi_table.as<LossyStep>();
		if (!runs)
		{
			for (int f = f0; f < nframes; ++f)
				for (int i = 0; i < nstreams; ++i)
				{
					LossyState &ls = os[i]->st;
					hs[(size_t)(f - f0) * nstreams + i] = ls.make_step(d_in[i] + (size_t)f * npx, d_in[i] + (size_t)f * npx, d_out[i] + (size_t)f * npx, add_loss != 0,
																		 os[i]->low, os[i]->high, os[i]->std_factor, errs_of(i, f), nstreams);
					ls.advance_ring();
					++ls.frames;
				}
			if (!hip_ok(hipMemcpyAsync(lead.multi_table.ptr, hs, nb, hipMemcpyHostToDevice, st), "H2D") || !hip_ok(hipEventRecord(lead.multi_copied, st), "event"))
				return -1;
			for (int f = 0; f < nsteps; ++f)
				if (!hip_ok(launch_lossy_step(hs + (size_t)f * nstreams, dt + (size_t)f * nstreams, nstreams, st), "lossy step"))
					return -1;
		}
		else
		{
			// histogram scratch: one zeroed 16 384-bin slice and one ticket per frame of a group (the kernels leave them zeroed)
			const size_t slices = (size_t)std::min(nsteps, group) * nstreams;
			const size_t hist_cap = lead.run_hist.cap, tick_cap = lead.run_tickets.cap;
			if (!lead.run_hist.reserve(slices * 16384 * 4) || !lead.run_tickets.reserve(slices * 4) || !lead.run_bg.reserve(nbg * sizeof(long long)))
				return -1;
			if (lead.run_hist.cap != hist_cap && !hip_ok(hipMemsetAsync(lead.run_hist.ptr, 0, lead.run_hist.cap, st), "memset"))
				return -1;
			if (lead.run_tickets.cap != tick_cap && !hip_ok(hipMemsetAsync(lead.run_tickets.ptr, 0, lead.run_tickets.cap, st), "memset"))
				return -1;
			long long *bg = lead.run_bg.as<long long>();
			LossyStep *hh = hs + nfused;
			if (persistent)
			{
				// scratch of the run kernel: [ticket, error word | 256 B][streams][2][run_wgs][slot_words] exchange words (zeroed before every launch)
				const char *stride_env = getenv("RIR_LOSSY_SLOT_WORDS");
				const int slot_words = stride_env && atoi(stride_env) >= 4 ? atoi(stride_env) : 8;
				const size_t exch_words = (size_t)2 * run_wgs * slot_words + 16, exch_bytes = (size_t)nstreams * exch_words * 8;
				// workgroup 0 of a stream collects and decides (0: every workgroup does - measured slower at every stream count, kept for measurements)
				const char *leader_env = getenv("RIR_LOSSY_LEADER");
				const int leader = leader_env ? atoi(leader_env) : 1;
				const size_t exch_cap = lead.run_exchange.cap;
				if (!lead.run_exchange.reserve(256 + exch_bytes))
					return -1;
				if (lead.run_exchange.cap != exch_cap)
				{
					if (!hip_ok(hipMemsetAsync(lead.run_exchange.ptr, 0, 256, st), "memset"))
						return -1;
					lead.run_arrivals = 0; // (a fresh control block)
				}
				// The constant-budget form (lossy_kernels.hip: lossy_const_run_kernel): with stdFactor == 0 for every stream of the call each
				// group is first offered to an ordinary launch that needs no hand-offs between workgroups; it declines - on the device, nothing
				// waits for the host - when a frame's class may be empty or a NaN sits in a window, and the resident launch behind it, which
				// otherwise finds the group done and leaves, steps it.
				const size_t part_words = (size_t)kLossyConstSlots * lossy_const_workgroups(full_px, nstreams) * 4;
				lead.const_groups = 0;
				if (const_form)
				{
					if (!lead.const_ok.reserve((size_t)ngroups * 4) || !lead.const_partials.reserve((size_t)nstreams * part_words * 8) ||
						!hip_ok(hipMemsetAsync(lead.const_ok.ptr, 0, (size_t)ngroups * 4, st), "memset"))
						return -1;
					lead.const_groups = ngroups;
				}
				lead.spec_groups = 0;
				bool spec_plane = false;
				const int spec_slabs = lossy_spec_stat_workgroups(s_px);
				const size_t spec_group = (size_t)std::min(nsteps, group);
				if (spec_form)
				{
					const size_t bk_cap = lead.spec_backoff.cap, tk_cap = lead.spec_tickets.cap;
					if (!lead.const_ok.reserve((size_t)ngroups * 4) || !hip_ok(hipMemsetAsync(lead.const_ok.ptr, 0, (size_t)ngroups * 4, st), "memset") ||
						!lead.spec_budgets.reserve((size_t)nstreams * (spec_group + 8) * 4) || !lead.spec_rows.reserve((size_t)nstreams * spec_group * spec_slabs * 32) ||
						!lead.spec_sd.reserve((size_t)nstreams * spec_group * 16) || !lead.spec_ctl.reserve((size_t)ngroups * nstreams * 32) ||
						!hip_ok(hipMemsetAsync(lead.spec_ctl.ptr, 0, (size_t)ngroups * nstreams * 32, st), "memset") || !lead.spec_backoff.reserve(8) ||
						!lead.spec_tickets.reserve((size_t)nstreams * spec_group * 4))
						return -1;
					// the byte plane the streaming kernel leaves for the sums kernel (lossy_kernels.h: LossySpec::dplane), a byte per pixel and frame of a group;
					// RIR_LOSSY_SPEC_NO_PLANE (measurements): none - the sums are taken from the frames, as they are when the memory is not to be had
					spec_plane = !getenv("RIR_LOSSY_SPEC_NO_PLANE") && lead.spec_dplane.reserve((size_t)nstreams * spec_group * (size_t)s_px);
					if (lead.spec_tickets.cap != tk_cap && !hip_ok(hipMemsetAsync(lead.spec_tickets.ptr, 0, lead.spec_tickets.cap, st), "memset"))
						return -1;
					if (lead.spec_backoff.cap != bk_cap && !hip_ok(hipMemsetAsync(lead.spec_backoff.ptr, 0, 8, st), "memset"))
						return -1;
					if (!lead.spec_backoff_host.ptr)
					{
						if (!lead.spec_backoff_host.reserve(64))
							return -1;
						std::memset(lead.spec_backoff_host.ptr, 0, 64);
					}
					for (int i = 0; i < nstreams; ++i)
						if (!os[i]->st.reserve_shadow())
							return -1;
					lead.spec_groups = ngroups, lead.spec_streams = nstreams;
				}
				unsigned int *d_ticket = lead.run_exchange.as<unsigned int>(), *d_error = d_ticket + 16;
				unsigned long long *d_exch = reinterpret_cast<unsigned long long *>(lead.run_exchange.as<char>() + 256);
				LossyRun *hr = reinterpret_cast<LossyRun *>(reinterpret_cast<char *>(hs) + run_off);
				const LossyRun *dr = reinterpret_cast<const LossyRun *>(lead.multi_table.as<char>() + run_off);
				LossySpec *hsp = reinterpret_cast<LossySpec *>(reinterpret_cast<char *>(hs) + spec_off);
				const LossySpec *dsp = reinterpret_cast<const LossySpec *>(lead.multi_table.as<char>() + spec_off);
				for (int g = 0; g < ngroups; ++g)
				{
					const int k0 = g * group, in_group = std::min(group, nsteps - k0);
					for (int i = 0; i < nstreams; ++i)
					{
						LossyState &ls = os[i]->st;
						snaps.push_back(Snap{ls.dev.ra_count, ls.dev.ra_head, (int64_t)ls.frames});
						LossyRun r{};
						r.in = d_in[i] + (size_t)(f0 + k0) * npx, r.out = d_out[i] + (size_t)(f0 + k0) * npx;
						r.st = ls.dev;
						r.bg = bg + (size_t)k0 * nstreams + i, r.bg_stride = nstreams;
						r.budget = ls.d_budget.as<LossyBudget>(), r.decision = ls.d_decision.as<LossyDecision>();
						r.errors_out = errs_of(i, f0 + k0);
						r.exchange = d_exch + (size_t)i * exch_words, r.error_word = d_error, r.slot_words = slot_words, r.leader = leader;
						r.frame_px = (long long)npx, r.nsteps = in_group;
						r.s = s_px, r.full = full_px;
						r.add_loss = add_loss ? 1 : 0, r.low_value_error = os[i]->low, r.high_value_error = os[i]->high, r.std_factor = os[i]->std_factor;
						r.partials = const_form ? lead.const_partials.as<unsigned long long>() + (size_t)i * part_words : nullptr;
						hr[(size_t)g * nstreams + i] = r;
						if (spec_form)
						{
							LossySpec sp{};
							sp.shadow = ls.shadow;
							sp.shadow.ra_count = ls.dev.ra_count, sp.shadow.ra_head = ls.dev.ra_head, sp.shadow.running_average = ls.dev.running_average;
							sp.budgets = lead.spec_budgets.as<uint32_t>() + (size_t)i * (spec_group + 8);
							sp.rows = lead.spec_rows.as<unsigned long long>() + (size_t)i * spec_group * spec_slabs * 4;
							sp.sd = lead.spec_sd.as<double>() + (size_t)i * spec_group * 2;
							sp.tickets = lead.spec_tickets.as<unsigned int>() + (size_t)i * spec_group;
							sp.ctl = lead.spec_ctl.as<unsigned int>() + ((size_t)g * nstreams + i) * 8;
							sp.backoff = lead.spec_backoff.as<unsigned int>();
							sp.backoff_host = lead.spec_backoff_host.as<unsigned int>();
							sp.dplane = spec_plane ? lead.spec_dplane.as<uint8_t>() + (size_t)i * spec_group * (size_t)s_px : nullptr;
							hsp[(size_t)g * nstreams + i] = sp;
						}
						for (int k = k0; k < k0 + in_group; ++k)
						{ // (the host side of the stream's state as it will be after the group)
							ls.advance_ring();
							++ls.frames;
						}
					}
				}
				if (!hip_ok(hipMemcpyAsync(lead.multi_table.ptr, hs, nb, hipMemcpyHostToDevice, st), "H2D") || !hip_ok(hipEventRecord(lead.multi_copied, st), "event"))
					return -1;
				// the call is on the leader's books before anything is queued: whatever happens from here on, a raised error word
				// is charged to it
				++lead.led_calls;
				for (int i = 0; i < nstreams; ++i)
				{
					os[i]->lead_call = lead.led_calls;
					os[i]->led_by_other = i != 0;
					if (i == 0)
						os[i]->lead.reset();
					else
						os[i]->lead = lead.weak_from_this();
				}
				if (test_hook("RIR_DEBUG_LOSSY_GIVE_UP") && !hip_ok(hipMemsetAsync(d_error, 1, 4, st), "memset")) // (tests: as if a wait had hit its clock)
					return -1;
				// A stream that backs off (its groups kept failing: budgets that move) is not offered its next groups: the device's counter says how
				// many, the host looks at its page-locked copy - without waiting, so the value may be a call or two old: then fewer groups are left
				// out here and the device skips them itself - and does not even queue their launches (eleven small ones per group).
				int host_skipped = 0;
				if (spec_form)
				{
					const unsigned int left = *reinterpret_cast<volatile unsigned int *>(lead.spec_backoff_host.ptr);
					host_skipped = (int)std::min<unsigned int>(left, (unsigned int)ngroups);
					if (host_skipped > 0 && !hip_ok(launch_lossy_spec_skipped(lead.spec_backoff.as<unsigned int>(), lead.spec_backoff_host.as<unsigned int>(),
																			  (unsigned int)host_skipped, st),
													"lossy speculative skip"))
						return -1;
				}
				for (int g = 0; g < ngroups; ++g)
				{
					const int k0 = g * group, in_group = std::min(group, nsteps - k0);
					if (!hip_ok(launch_lossy_backgrounds_of_runs(dr + (size_t)g * nstreams, nstreams, in_group, lead.run_hist.as<uint32_t>(), lead.run_tickets.as<unsigned int>(), s_px,
																 lossy_hist_px(s_px, in_group * nstreams), st),
								"lossy backgrounds") ||
						!hip_ok(hipMemsetAsync(d_exch, 0, exch_bytes, st), "memset"))
						return -1;
					const unsigned int *d_ok = stream_form ? lead.const_ok.as<unsigned int>() + g : nullptr;
					bool any_ra = false;
					for (int i = 0; i < nstreams; ++i)
						any_ra = any_ra || os[i]->st.dev.running_average > 0;
					if (const_form && !hip_ok(launch_lossy_const(dr + (size_t)g * nstreams, nstreams, full_px, any_ra, add_loss != 0, lead.const_ok.as<unsigned int>() + g,
																  d_ticket + kLossyRunCtlWord + 2, st),
											  "lossy constant-budget run"))
						return -1;
					if (spec_form && g >= host_skipped)
					{ // guess, (step, sums, verify) x passes, commit: every launch returns at once when the one before left it nothing to do
						if (!hip_ok(launch_lossy_spec_begin(dr + (size_t)g * nstreams, dsp + (size_t)g * nstreams, nstreams, spec_passes, lead.const_ok.as<unsigned int>() + g,
															 d_ticket + kLossyRunCtlWord + 2, st),
									"lossy speculative run"))
							return -1;
						for (int pass = 0; pass < (spec_passes & 0xff); ++pass)
							if (!hip_ok(launch_lossy_spec_pass(dr + (size_t)g * nstreams, dsp + (size_t)g * nstreams, nstreams, s_px, full_px, in_group, any_ra, add_loss != 0, st),
										"lossy speculative pass"))
								return -1;
						if (!hip_ok(launch_lossy_spec_commit(dr + (size_t)g * nstreams, dsp + (size_t)g * nstreams, nstreams, s_px, full_px, lead.const_ok.as<unsigned int>() + g, st),
									"lossy speculative commit"))
							return -1;
					}
					for (int s0 = 0; s0 < nstreams; s0 += batch)
					{
						const int nl = std::min(batch, nstreams - s0);
						const unsigned int epoch = ++lead.run_epoch;
						if (s0 == 0)
							run_epochs.push_back(epoch);
						if (test_hook("RIR_DEBUG_LOSSY_BAIL") && atoi(test_hook("RIR_DEBUG_LOSSY_BAIL")) == g)
						{ // (tests: as if group g's launch had not become resident: its decision is BAIL before anybody arrives)
							const unsigned int w_ = ((epoch & 0x3fffffffu) << 2) | 2u;
							if (!hip_ok(hipMemcpyAsync(d_ticket + kLossyRunCtlWord + 1, &w_, 4, hipMemcpyHostToDevice, st), "H2D") || !hip_ok(hipStreamSynchronize(st), "sync"))
								return -1;
						}
						if (!hip_ok(launch_lossy_run(dr + (size_t)g * nstreams + s0, nl, full_px, d_ticket, epoch, lead.run_arrivals, parked, st, d_ok), "lossy run"))
							return -1;
						lead.run_arrivals += (unsigned int)(nl * run_wgs);
					}
				}
				run_group = group, run_launches_per_group = (nstreams + batch - 1) / batch;
			}
			else
			{
			for (int k = 0; k <= nsteps; ++k) // launch k updates frame f0 + k - 1 (k > 0) and takes the sums of frame f0 + k (k < nsteps)
				for (int i = 0; i < nstreams; ++i)
				{
					LossyState &ls = os[i]->st;
					LossyStep p;
					if (k == 0)
					{
						p = ls.make_step(nullptr, nullptr, nullptr, add_loss != 0, os[i]->low, os[i]->high, os[i]->std_factor, nullptr, nstreams);
						p.do_update = 0;
					}
					else
					{
						const size_t f = (size_t)(f0 + k - 1);
						p = ls.make_step(d_in[i] + f * npx, d_in[i] + f * npx, d_out[i] + f * npx, add_loss != 0, os[i]->low, os[i]->high, os[i]->std_factor, nullptr,
										 nstreams);
						ls.advance_ring();
						++ls.frames;
					}
					if (k < nsteps)
					{
						const size_t fn = (size_t)(f0 + k);
						p.next_tmp = p.next_img = d_in[i] + fn * npx;
						p.next_background = bg + (size_t)k * nstreams + i;
						p.next_errors_out = errs_of(i, (int)fn);
						LossyStep h{};
						const int in_group = std::min(group, nsteps - k / group * group);
						const size_t slice = (size_t)(k % group) * nstreams + i;
						h.tmp = h.img = d_in[i] + fn * npx;
						h.hist = lead.run_hist.as<uint32_t>() + slice * 16384;
						h.stats = bg + (size_t)k * nstreams + i;
						h.tickets = lead.run_tickets.as<unsigned int>() + slice;
						h.s = s_px, h.full = full_px;
						h.hist_px = lossy_hist_px(s_px, in_group * nstreams);
						hh[(size_t)k * nstreams + i] = h;
					}
					hs[(size_t)k * nstreams + i] = p;
				}
			if (!hip_ok(hipMemcpyAsync(lead.multi_table.ptr, hs, nb, hipMemcpyHostToDevice, st), "H2D") || !hip_ok(hipEventRecord(lead.multi_copied, st), "event"))
				return -1;
			for (int k = 0; k <= nsteps; ++k)
			{
				if (k < nsteps && k % group == 0)
				{
					const int in_group = std::min(group, nsteps - k);
					if (!hip_ok(launch_lossy_backgrounds(dt + nfused + (size_t)k * nstreams, in_group * nstreams, s_px, lossy_hist_px(s_px, in_group * nstreams), st),
								"lossy backgrounds"))
						return -1;
				}
				if (!hip_ok(launch_lossy_frame(dt + (size_t)k * nstreams, nstreams, full_px, st), "lossy frame"))
					return -1;
			}
			}
		}
	}
	if (want && !d_errs)
	{
		std::vector<int> e((size_t)nstreams * nframes * 2);
		unsigned int gave_up = 0; // the run kernel's error word (a wait between workgroups that hit its clock)
		unsigned int ctl[4] = {0, 0, 0, 0}; // the residency control block: arrivals, decision, poison, epoch of the first launch that bailed out
		if ((lead.run_exchange.ptr && (!hip_ok(hipMemcpyAsync(&gave_up, lead.run_exchange.as<unsigned int>() + 16, 4, hipMemcpyDeviceToHost, st), "D2H") ||
									   !hip_ok(hipMemcpyAsync(ctl, lead.run_exchange.as<unsigned int>() + kLossyRunCtlWord, sizeof(ctl), hipMemcpyDeviceToHost, st), "D2H"))) ||
			!hip_ok(wait_stream(st), "sync"))
			return -1;
		if (ctl[2] != 0 && !run_epochs.empty())
		{
			// A group's launch did not become resident (ordinary kernels of other streams can keep the last workgroups from fitting:
			// resident_device.h).  It and every launch behind it have written nothing.  The call waits here anyway, so the frames
			// from that group on are stepped again on the launch-per-frame path - same results - from the host state of that moment.
			int g_bad = -1;
			for (size_t g = 0; g < run_epochs.size(); ++g)
				if (g_bad < 0 && (int)(ctl[3] - run_epochs[g]) >= 0 && (int)(ctl[3] - run_epochs[g]) < run_launches_per_group)
					g_bad = (int)g;
			const unsigned int zero2[2] = {0, 0};
			if (!hip_ok(hipMemcpyAsync(lead.run_exchange.as<unsigned int>() + kLossyRunCtlWord + 2, zero2, sizeof(zero2), hipMemcpyHostToDevice, st), "H2D") ||
				!hip_ok(hipStreamSynchronize(st), "sync"))
				return -1;
			if (g_bad < 0 || run_launches_per_group != 1)
			{ // (a poison left by an earlier, queue-only call - or streams that went in several batches and stand at different frames)
				gave_up = 1;
			}
			else
			{
				const int fr0 = f0 + g_bad * run_group; // first frame that was not stepped
				for (int i = 0; i < nstreams; ++i)
				{
					const Snap &sn = snaps[(size_t)g_bad * nstreams + i];
					os[i]->st.dev.ra_count = sn.ra_count, os[i]->st.dev.ra_head = sn.ra_head, os[i]->st.frames = (decltype(os[i]->st.frames))sn.frames;
				}
				std::vector<const unsigned short *> in2((size_t)nstreams);
				std::vector<unsigned short *> out2((size_t)nstreams);
				std::vector<int *> errs2((size_t)nstreams);
				for (int i = 0; i < nstreams; ++i)
				{
					in2[i] = d_in[i] + (size_t)fr0 * npx, out2[i] = d_out[i] + (size_t)fr0 * npx;
					errs2[i] = lead.batch_errs.as<int>() + ((size_t)i * nframes + fr0) * 2;
				}
				if (lossy_step_streams(os, nstreams, in2.data(), out2.data(), nframes - fr0, add_loss, errs2.data(), nullptr, nullptr, st, true) != 0 ||
					!hip_ok(wait_stream(st), "sync"))
					return -1;
			}
		}
		if (!hip_ok(hipMemcpyAsync(e.data(), lead.batch_errs.ptr, e.size() * sizeof(int), hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
			return -1;
		if (getenv("RIR_LOSSY_DIAG") && lead.run_exchange.ptr)
		{ // (-DRIR_LOSSY_DIAG builds) where the time of a frame goes, workgroup 0 of stream 0
			unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			if (hip_ok(hipMemcpy(dg, lead.run_exchange.as<char>() + 128, sizeof(dg), hipMemcpyDeviceToHost), "D2H") && dg[4])
				std::fprintf(stderr, "lossy run, per frame (us): loads+pixel sums %.2f  reduce+publish %.2f (wave sums + barrier %.2f, merge + publish %.2f, window sum %.2f)  poll %.2f  budget %.2f  barrier+update %.2f  (%llu frames)\n",
							 dg[5] * 0.01 / dg[4], dg[0] * 0.01 / dg[4], dg[6] * 0.01 / dg[4], dg[7] * 0.01 / dg[4], (dg[0] - dg[6] - dg[7]) * 0.01 / dg[4], dg[1] * 0.01 / dg[4],
							 dg[2] * 0.01 / dg[4], dg[3] * 0.01 / dg[4], dg[4]);
		}
		if (lead.run_exchange.ptr)
			lead.checked(gave_up, st);
		bool any_failed = false;
		for (int i = 0; i < nstreams; ++i)
			any_failed = os[i]->is_failed() || any_failed; // (every stream of the call is marked)
		if (any_failed)
		{
			log_error("rir_lossy_step_multi_device: a run of frames gave up waiting (results invalid; the streams of the call take no more frames)");
			return -1;
		}
		for (size_t i = 0; i < (size_t)nstreams * nframes; ++i)
		{
			if (low_errors)
				low_errors[i] = e[2 * i];
			if (high_errors)
				high_errors[i] = e[2 * i + 1];
		}
	}
	return 0;
}


} // namespace

RIR_EXPORT int rir_lossy_step_multi_device(const int *handles, int nstreams, const unsigned short *const *d_in, unsigned short *const *d_out,
										   int nframes, int add_loss, int *low_errors, int *high_errors, void *stream)
{
	if (!device_ready())
		return -1;
	if (!handles || nstreams <= 0 || nstreams > 65535 || !d_in || !d_out || nframes <= 0)
	{
		log_error("rir_lossy_step_multi_device: invalid argument");
		return -1;
	}
	std::vector<std::shared_ptr<LossyObject>> os((size_t)nstreams);
	std::vector<LossyObject *> ptrs((size_t)nstreams);
	for (int i = 0; i < nstreams; ++i)
	{
		os[i] = lookup_as<LossyObject>(handles[i]);
		if (!os[i] || !d_in[i] || !d_out[i] || d_in[i] == d_out[i])
		{
			log_error("rir_lossy_step_multi_device: invalid handle or buffer");
			return -1;
		}
		{ // (the runs of frames read input frames again - the frame that leaves the running average - after later outputs have been written)
			const size_t span = (size_t)nframes * (size_t)os[i]->st.w * (size_t)os[i]->st.h;
			if (d_in[i] < d_out[i] + span && d_out[i] < d_in[i] + span)
			{
				log_error("rir_lossy_step_multi_device: input and output frames overlap");
				return -1;
			}
		}
		for (int k = 0; k < i; ++k)
			if (os[k] == os[i])
			{
				log_error("rir_lossy_step_multi_device: a stream appears twice");
				return -1;
			}
		ptrs[i] = os[i].get();
		const LossyState &a = os[0]->st, &b = os[i]->st;
		if (a.w != b.w || a.h != b.h || a.hl != b.hl || a.frames != b.frames || os[i]->remove_bad_pixels)
		{
			log_error("rir_lossy_step_multi_device: the streams must share geometry and history length (and not repair bad pixels)");
			return -1;
		}
	}
	return lossy_step_streams(ptrs.data(), nstreams, d_in, d_out, nframes, add_loss, nullptr, low_errors, high_errors, (hipStream_t)stream);
}

// Which form stepped the groups of frames of the last run call this stream LED (a call of its own, or a multi-stream call it was first in):
// out[0] = groups offered to the constant-budget form (0: the call was not eligible - stdFactor != 0, frames stepped one by one, ...),
// out[1] = of those, groups it stepped (the others were declined on the device and stepped by the resident kernel).  Waits for `stream`.
RIR_EXPORT int rir_lossy_path_stats(int handle, int *out2, void *stream)
{
	auto o = lookup_as<LossyObject>(handle);
	if (!o || !out2)
	{
		log_error("rir_lossy_path_stats: invalid argument");
		return -1;
	}
	out2[0] = o->const_groups, out2[1] = 0;
	if (o->const_groups > 0)
	{
		std::vector<unsigned int> w((size_t)o->const_groups);
		if (!hip_ok(hipMemcpyAsync(w.data(), o->const_ok.ptr, w.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)stream), "D2H") ||
			!hip_ok(wait_stream((hipStream_t)stream), "sync"))
			return -1;
		for (unsigned int v : w)
			out2[1] += v != 0 ? 1 : 0;
	}
	return 0;
}

// The speculative form's books for the last run call this stream LED: out[0] = groups that went through its launches (0: the call was not
// eligible - stdFactor 0, frames stepped one by one, ...), out[1] = of those, groups that were offered (precondition held, no back-off),
// out[2] = groups it committed, out[3] = passes over all groups (a group's passes: the most any of its streams took).  Waits for `stream`.
RIR_EXPORT int rir_lossy_spec_stats(int handle, int *out4, void *stream)
{
	auto o = lookup_as<LossyObject>(handle);
	if (!o || !out4)
	{
		log_error("rir_lossy_spec_stats: invalid argument");
		return -1;
	}
	out4[0] = o->spec_groups, out4[1] = out4[2] = out4[3] = 0;
	if (o->spec_groups > 0)
	{
		const size_t ng = (size_t)o->spec_groups, ns = (size_t)o->spec_streams;
		std::vector<unsigned int> w(ng * ns * 8), okw(ng);
		if (!hip_ok(hipMemcpyAsync(w.data(), o->spec_ctl.ptr, w.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)stream), "D2H") ||
			!hip_ok(hipMemcpyAsync(okw.data(), o->const_ok.ptr, okw.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)stream), "D2H") ||
			!hip_ok(wait_stream((hipStream_t)stream), "sync"))
			return -1;
		for (size_t g = 0; g < ng; ++g)
		{
			unsigned int passes = 0;
			for (size_t i = 0; i < ns; ++i)
				passes = std::max(passes, w[(g * ns + i) * 8 + 1]);
			out4[1] += w[g * ns * 8 + 4] != 0 ? 1 : 0;
			out4[2] += okw[g] != 0 ? 1 : 0;
			out4[3] += (int)passes;
		}
	}
	return 0;
}

// 0 when no run of frames this stream took part in has given up a wait between workgroups, -1 otherwise or on an invalid handle.
// Waits for the work queued on `stream` (calls queued on other streams are not covered).  The failure is sticky: the stream's state
// has been advanced by invalid frames, so from then on every status and every step of the stream - and of the streams that shared
// the failed call - returns -1 until it is destroyed.  Calls that return budgets check this themselves; queue-only calls do not.
RIR_EXPORT int rir_lossy_status(int handle, void *stream)
{
	auto o = lookup_as<LossyObject>(handle);
	if (!o)
	{
		log_error("rir_lossy_status: invalid handle");
		return -1;
	}
	hipStream_t st = (hipStream_t)stream;
	// the error word is with the leader of the last run call this stream took part in (itself, for a call of its own)
	LossyObject *l = o->leader();
	std::shared_ptr<Object> keep = o->lead.lock(); // (keeps a foreign leader alive while it is read)
	if (!l || !l->run_exchange.ptr)
		return hip_ok(wait_stream(st), "sync") && !o->is_failed() ? 0 : -1;
	unsigned int gave_up = 0, poison = 0; // a wait that hit its clock / a launch that was called off because it did not become resident
	if (!hip_ok(hipMemcpyAsync(&gave_up, l->run_exchange.as<unsigned int>() + 16, 4, hipMemcpyDeviceToHost, st), "D2H") ||
		!hip_ok(hipMemcpyAsync(&poison, l->run_exchange.as<unsigned int>() + kLossyRunCtlWord + 2, 4, hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(wait_stream(st), "sync"))
		return -1;
	l->checked(gave_up | poison, st);
	if (o->is_failed())
	{
		log_error("rir_lossy_status: a run of frames gave up waiting (results invalid; the stream takes no more frames)");
		return -1;
	}
	return 0;
}

RIR_EXPORT void rir_lossy_destroy(int handle)
{
	if (lookup_as<LossyObject>(handle))
		remove_object(handle);
}

static int errors_out(const std::vector<unsigned short> &err, unsigned short *errors, int *size)
{
	if (!size)
		return -1;
	if (*size < (int)err.size() || !errors)
	{
		*size = (int)err.size();
		return -2;
	}
	*size = (int)err.size();
	if (!err.empty())
		std::memcpy(errors, err.data(), err.size() * sizeof(unsigned short));
	return 0;
}
RIR_EXPORT int h264_get_low_errors(int file, unsigned short *errors, int *size)
{
	auto s = saver(file);
	if (!s)
	{
		log_error("h264_get_low_erros: NULL identifier");
		return -1;
	}
	if (!s->resolve_errors())
		return -1;
	return errors_out(s->low_errors, errors, size);
}
RIR_EXPORT int h264_get_high_errors(int file, unsigned short *errors, int *size)
{
	auto s = saver(file);
	if (!s)
	{
		log_error("h264_get_high_erros: NULL identifier");
		return -1;
	}
	if (!s->resolve_errors())
		return -1;
	return errors_out(s->high_errors, errors, size);
}

// Declared by the reference header (video_io.h:305-314) but defined nowhere upstream; the arguments are those of
// z_open_file_write (ZFile.cpp:325).  method 1 writes the reference's ZFile container (one zstd frame per image, host
// side); any other method writes this build's block-codec container (clevel has no meaning there).
RIR_EXPORT int open_video_write(const char *filename, int width, int height, int rate, int method, int clevel)
{
	if (!filename)
		return -1;
	if (method == 1)
		return guarded("open_video_write", -1, [&] {
			auto w = std::make_shared<ZWriterObject>();
			if (!w->open(filename, width, height, rate, clevel))
				return -1;
			return register_object(w);
		});
	const int h = h264_open_file(filename, width, height, height);
	if (h <= 0)
		return -1;
	saver(h)->fps = rate > 0 ? rate : 50;
	return h;
}
RIR_EXPORT int image_write(int writter, unsigned short *img, int64_t time)
{
	return guarded("image_write", -1, [&] {
		if (auto z = lookup_as<ZWriterObject>(writter))
			return z->add(img, time) ? 0 : -1;
		auto s = saver(writter);
		if (!s)
			return -1;
		return s->add_image(img, time, AttrMap()) ? 0 : -1;
	});
}
RIR_EXPORT int64_t close_video(int writter)
{
	int64_t size = -1;
	guarded("close_video", -1, [&] {
		if (auto z = lookup_as<ZWriterObject>(writter))
			size = z->close();
		else if (auto s = saver(writter))
			size = s->close();
		else
			return -1;
		remove_object(writter);
		return 0;
	});
	return size;
}

// Internal helpers of the reference for vendor files (video_io.cpp:911-958): not part of the path.
RIR_EXPORT int correct_PCR_file(const char *filename, int width, int height, int freq)
{
	FILE *f = filename ? std::fopen(filename, "r+b") : nullptr;
	if (!f)
		return -1;
	PcrHeader h;
	if (std::fread(&h, sizeof(h), 1, f) != 1)
	{
		std::fclose(f);
		return -1;
	}
	h.X = h.GrabSizeX = width;
	h.Y = h.GrabSizeY = height;
	h.Bits = 16;
	h.Version = 0;
	h.TransfertSize = width * height * 2;
	h.Frequency = freq;
	std::fseek(f, 0, SEEK_SET);
	const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
	std::fclose(f);
	return ok ? 0 : -1;
}
RIR_EXPORT int change_hcc_external_blackbody_temperature(const char *, float)
{
	log_error("change_hcc_external_blackbody_temperature: HCC vendor files are outside the accelerated path");
	return -1;
}
