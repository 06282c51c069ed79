// Device state and launchers of the bounded-loss step (lossy_kernels.hip).  Internal, C++ linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rir
{
	// Per-saver state living in HBM; reference members: H264_Saver::PrivateData refT, prevT, lastDL
	// (h264.cpp:1640-1660) and RunningAverage2 (h264.cpp:1526-1615).
	struct LossyDeviceState
	{
		uint16_t *refT, *prevT, *lastDL; // [w*h]
		uint32_t *ra_sums;				 // [s]
		uint16_t *ra_const_value;		 // [s]
		int16_t *ra_const_count;		 // [s]
		uint16_t *ra_images;			 // ring [running_average][s]
		int ra_count, ra_head, running_average;
		int subtract_min;
		uint32_t min;
	};

	// Error budget of the stream (h264.cpp:2335-2385): statistic of the first frame and of the last 40, kept in HBM.
	struct LossyBudget
	{
		double first_std[2];
		double win[40][2]; // a ring once it is full: the oldest entry is win[head]
		int n_first, n_win;
		int head, reserved;
	};
	// What lossy_budget_kernel decides for the current frame and lossy_update_kernel applies.
	struct LossyDecision
	{
		uint32_t background;
		int low_error, high_error;
		int reserved;
	};

	// Everything the three kernels of one frame of one stream need (device pointers; plain data, passed by value or in a table).
	struct LossyStep
	{
		const uint16_t *tmp, *img; // the frame after / before bad-pixel repair (the same without it)
		uint16_t *out;
		LossyDeviceState st; // ring indices as they are for THIS frame
		uint32_t *hist;		 // 16 384 bins
		long long *stats;	 // [8]: background, then the six sums
		LossyBudget *budget;
		LossyDecision *decision;
		int *errors_out;	   // [2] low, high of this frame, or NULL
		unsigned int *tickets; // [2] arrival counters of the two reduction passes
		int s, full;		   // pixels below lossy_height, pixels of the frame
		int hist_px, reserved; // pixels per workgroup of the histogram pass (lossy_hist_px)
		int add_loss, low_value_error, high_value_error;
		double std_factor;
		// Runs of frames (lossy_frame_kernel: one launch per frame): the launch that updates THIS frame also takes the sums of the
		// NEXT one - its pixels against this frame's output, which the thread holds in registers - and its last workgroup decides
		// the next frame's budget.
		const uint16_t *next_tmp, *next_img; // frame f + 1 after / before repair; NULL: this is the last frame of the call
		const long long *next_background;	 // its background, from the histogram pass of the run
		int *next_errors_out;				 // [2] low, high of frame f + 1, or NULL
		int do_update, reserved2;			 // 0: opening launch of a run (sums of its first frame against the stored prevT only)
	};
	// A run of frames of one stream in ONE launch (lossy_run_kernel): the workgroups of the stream stay resident, every thread keeps
	// the state of its 8 pixels in registers from frame to frame, and the frame's sums are exchanged between the workgroups through
	// self-validating words in HBM instead of a kernel boundary.
	struct LossyRun
	{
		const uint16_t *in; // first frame of the run; frames are frame_px pixels apart
		uint16_t *out;
		LossyDeviceState st;	  // ring indices before the first frame of the run
		const long long *bg;	  // background of frame k of the run at bg[k * bg_stride] (histogram pass)
		LossyBudget *budget;
		LossyDecision *decision;
		int *errors_out;			  // [nsteps][2] low, high - or NULL
		unsigned long long *exchange; // [2][workgroups of the stream][slot_words] + 16 words (decisions), zeroed before the launch
		unsigned int *error_word;	  // raised when a wait gives up
		long long frame_px;
		int bg_stride, nsteps;
		int s, full;
		int add_loss, low_value_error, high_value_error;
		int slot_words; // distance of two workgroups' exchange words, in 8-byte words
		int leader, reserved; // 1: workgroup 0 collects the sums and publishes the decision (many streams per launch)
		double std_factor;
		unsigned long long *partials; // constant-budget form: [kLossyConstSlots][workgroups of the stream][4] words, this stream's
	};
	// The constant-budget form of a run (lossy_const_run_kernel + lossy_const_finish_kernel): with stdFactor == 0 and no NaN in play the
	// budgets are the configured errors, frame after frame, and nothing a frame needs comes from another workgroup.  Only the last 40
	// frames of a group (and the stream's very first budget frame) leave sums behind: what the 40-frame window holds afterwards.
	constexpr int kLossyConstTail = 40, kLossyConstSlots = kLossyConstTail + 1;
	constexpr int kLossyConstMaxFrames = 2048; // frames of a group it takes (the callers' groups are at most that long)
	constexpr int kLossyRunThreads = 256; // 8 pixels each
	inline int lossy_run_workgroups(int full) { return (full / 8 + kLossyRunThreads - 1) / kLossyRunThreads; }
	// The run kernel needs ALL its workgroups resident at once: workgroup i runs on XCD i % 8, each XCD starts its own share of the
	// grid as its CUs come free, and a stream whose last workgroups can only start on an XCD that is full of its own waiting
	// workgroups would wait for ever (seen with an oversubscribed grid: one call in a few hundred).  How many the device holds is
	// asked of the runtime (lossy_run_capacity: occupancy of THIS kernel x the device's CUs, less a margin - on an MI355X, 5
	// workgroups per CU: 1 200), and the launch goes through the process-wide gate of runtime.h; more streams than fit go a batch
	// after the other, a frame whose workgroups do not fit (or a device the runtime cannot size) takes the launch-per-frame path.
	constexpr int kLossyRunWavesPerSimd = 5;
	// the second form of the run kernel (lossy_run_parked_kernel): part of the pixel state parked in LDS between the frames, 6 waves per
	// SIMD - more streams per launch (9 of 640x512 instead of 7), each a little slower: taken when it saves a launch
	constexpr int kLossyRunParkedWavesPerSimd = 6;
	int lossy_run_capacity(bool parked = false);

	// Pixels per workgroup of the histogram pass: each workgroup clears and merges a private 16 384-bin histogram, so a launch wants
	// about as many workgroups as the chip holds at once (two per CU) - 4 096 pixels for one 640x512 stream, more with many streams.
	inline int lossy_hist_px(int s, int nstreams)
	{
		long long px = ((long long)s * nstreams + 511) / 512;
		px = (px + 1023) / 1024 * 1024;
		return (int)(px < 4096 ? 4096 : px);
	}
	hipError_t launch_lossy_step(const LossyStep *h_steps, const LossyStep *d_table, int nstreams, hipStream_t st);
	// Runs of frames.  Backgrounds of `entries` frames (any streams) in one launch: d_table[e] describes frame e (tmp, hist = a
	// zeroed 16 384-bin slice of its own, stats -> where its background goes, tickets -> a zeroed word of its own, s, hist_px).
	hipError_t launch_lossy_backgrounds(const LossyStep *d_table, int entries, int s, int hist_px, hipStream_t st);
	// the same for `frames` frames of the nstreams runs d_runs describes (in, frame_px, bg, bg_stride): no per-frame table; d_hist: frames x nstreams
	// zeroed slices of 16 384 bins, d_tickets: as many zeroed words (entry = frame * nstreams + stream)
	hipError_t launch_lossy_backgrounds_of_runs(const LossyRun *d_runs, int nstreams, int frames, uint32_t *d_hist, unsigned int *d_tickets, int s, int hist_px, hipStream_t st);
	// One launch of a run for `nstreams` streams: d_table[i] is stream i's step (next_* filled in).
	hipError_t launch_lossy_frame(const LossyStep *d_table, int nstreams, int full, hipStream_t st);
	// d_table[i]: the run of stream i; d_ticket: the 256-byte header of the exchange buffer, zeroed once: word 0 the ticket (left zeroed), word 16
	// the error word, words kLossyRunCtlWord .. + 3 the residency control block (resident_device.h: arrivals, decision, poison, epoch of the
	// first launch that bailed out); epoch: a number no earlier launch on this header used; arrivals_before: workgroups of those launches
	constexpr int kLossyRunCtlWord = 48;
	hipError_t launch_lossy_run(const LossyRun *d_table, int nstreams, int full, unsigned int *d_ticket, unsigned int epoch, unsigned int arrivals_before, bool parked,
								hipStream_t st, const unsigned int *d_ok = nullptr);
	// The constant-budget form for the group of frames d_table describes (nstreams entries, partials filled in): both launches look at
	// the group's background words (bit 40: a class may be empty), the streams' budget windows (NaN) and *d_poison (an earlier group of
	// the call was not stepped) and do nothing unless everything is clear; *d_ok (zeroed by the caller) says which it was - the resident
	// launch that follows reads it and leaves the group alone when it is 1 (launch_lossy_run's d_ok).
	// The SPECULATIVE form of a run (round 6): streams whose budgets follow the frames' statistics (stdFactor != 0, the reference's default: 5,
	// h264.cpp:1662-1665).  The budget of frame k (h264.cpp:2335-2385) depends on the output of frame k - 1, so the chain is strict - unless the
	// budgets are GUESSED.  On scenes that do not move the rounded correction round(|std - mean| x stdFactor) is 0 frame after frame and the budgets
	// are the configured constants; so a group is first stepped by the streaming kernel with a per-frame budget table filled with that guess, into
	// SHADOW state (the stream's state stays what it was), a second kernel takes every frame's six exact sums from the frames where they lie
	// (input k against output k - 1: fully parallel), and a third runs the reference's double arithmetic for every frame (a thread per frame: the
	// window mean of frame k is its own chain of 40 additions over statistics that are all known).  The sums are right up to and including the first
	// frame m whose true budget is not the table's; from m on the table takes what was computed (right at m, roughly right behind it: a fixed-point
	// iteration that a step change or a flash converges under in two or three passes) and the group is stepped again, up to `passes` times - not at
	// all when the frames off the table do not halve from pass to pass: budgets that move, nothing to iterate on; a group whose table
	// then verifies is committed (shadow -> state, window, budgets), any other is left to the resident kernel behind, exactly as a declined
	// constant-budget group is.  A stream whose groups keep failing is not offered for a while (a counter on the device: 3, 15, 63 groups).
	struct LossySpec
	{
		LossyDeviceState shadow;   // where a pass leaves the state after the group (ring: the slots the group writes)
		uint32_t *budgets;		   // [nsteps] low | high << 16, both clamped to 0 .. 65 535 (a difference never exceeds that)
		unsigned long long *rows;  // [nsteps][stat workgroups][4] the sums of a frame, per slab of kLossySpecSlab pixels (words as lossy_const_run_kernel's partials)
		double *sd;				   // [nsteps][2] the statistic of every frame (sums kernel -> verify, commit)
		unsigned int *tickets;	   // [nsteps] arrival counters of the frames' slabs: zero between launches (the last arriver clears its own)
		unsigned int *ctl;		   // [8] 0: status (0 to be stepped, 1 verified, 2 given up), 1: passes so far, 2: first mismatch of the last pass, 3: passes allowed, 4: offered, 5: frames off the table at the last pass,
								   // 6: flags, 7: a pass met a difference of 128 or more - `dplane` does not hold it: the sums are taken from the frames
		unsigned int *backoff;	   // [2] of the call's leading stream: groups still to skip, failures in a row
		unsigned int *backoff_host; // [2] the same two words in page-locked host memory, written whenever they change: the host looks (without
								   // waiting) before it queues a call, and does not queue the launches of groups that would be skipped anyway
		uint8_t *dplane;		   // [nsteps][s] or null: a byte per pixel and frame, left by the streaming kernel for the sums kernel - |input k - output k - 1| in bits 0..6,
								   // input k > background k in bit 7: what the sums are made of, at a quarter of the bytes of the two frames
	};
	constexpr int kLossySpecSlab = 16384; // pixels of a frame per workgroup of the sums kernel
	inline int lossy_spec_stat_workgroups(int s) { return (s + kLossySpecSlab - 1) / kLossySpecSlab; }
	// begin: precondition (as the constant-budget form's) and back-off, the guess into every stream's table; one pass = streaming kernel + sums + verify;
	// commit: all streams verified -> shadow state, windows, budgets into place, *d_ok = 1.  Every launch looks at the words the one before left and
	// returns at once when there is nothing for it to do; nothing waits for the host.
	hipError_t launch_lossy_spec_begin(const LossyRun *d_table, const LossySpec *d_spec, int nstreams, int passes, unsigned int *d_ok, const unsigned int *d_poison, hipStream_t st);
	hipError_t launch_lossy_spec_pass(const LossyRun *d_table, const LossySpec *d_spec, int nstreams, int s, int full, int max_frames, bool any_ra, bool add_loss, hipStream_t st);
	// groups the host did not queue because the stream backs off: taken off the device's counter (one launch per call)
	hipError_t launch_lossy_spec_skipped(unsigned int *d_backoff, unsigned int *backoff_host, unsigned int count, hipStream_t st);
	hipError_t launch_lossy_spec_commit(const LossyRun *d_table, const LossySpec *d_spec, int nstreams, int s, int full, unsigned int *d_ok, hipStream_t st);
	void lossy_const_force_pairs(int np); // 4, 2, 1: that many pairs of pixels per thread whatever the launch; anything else: chosen by lossy_const_pairs (test hook RIR_LOSSY_CONST_PAIRS)
	int lossy_const_workgroups(int full, int nstreams); // workgroups of a stream in that launch: partials holds kLossyConstSlots x this many x 4 words per stream
	// any_ra: some stream of the launch keeps a running average; add_loss: the addLoss variant of the decision (the same for all streams)
	hipError_t launch_lossy_const(const LossyRun *d_table, int nstreams, int full, bool any_ra, bool add_loss, unsigned int *d_ok, const unsigned int *d_poison, hipStream_t st);
	hipError_t launch_lossy_first(const uint16_t *d_tmp, uint16_t *d_out, const LossyDeviceState &state, int s, int full, hipStream_t st);
	hipError_t launch_lossy_min(const uint16_t *d_tmp, int s, unsigned int *d_result, hipStream_t st);
	hipError_t launch_lossy_add_min(uint16_t *d_frames, int64_t npx, int s, int nframes, uint32_t mn, hipStream_t st);
} // namespace rir
