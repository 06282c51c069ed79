// Bounded-loss recording on the GPU: the per-frame loss-injection step of the reference's lossy saver
// (reference src/cpp/video_io/h264.cpp: get_background :1955-1991, stdDev :1993-2036,
// RunningAverage2 :1526-1615, decision loop :2397-2413 / :2574-2590).  The state is sequential in
// time and parallel in space; all integer, exact.  Three forms of the same arithmetic:
//   one frame            three kernels (two reductions with a sequential tail each, one elementwise update)
//   a run of frames      lossy_run_kernel: the stream's workgroups stay resident, pixel state in registers, the frames' sums
//                        exchanged between workgroups through tagged words (launches of <= 960 workgroups)
//   a run, large frames  lossy_frame_kernel, one launch per frame: update of frame f fused with the sums of frame f + 1
//                        or many streams
// (a run's backgrounds come from one histogram launch per group of frames: they depend on the input only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

#include "lossy_kernels.h"
#include "resident_device.h"
#include "runtime.h"

namespace rir
{
	__device__ __forceinline__ uint32_t sub_min(uint32_t v, uint32_t mn) { return v < mn ? 0u : v - mn; }

	// One frame of one stream = three launches (histogram + mode, sums + budget, update); blockIdx.y selects the stream when
	// several independent streams are stepped by the same launches (rir_lossy_step_multi_device: SURVEY §8e "replicas").  The
	// mode of the histogram and the error budget are sequential tails of a parallel pass: they are run by the LAST workgroup of
	// that pass to arrive (an agent-scope ticket per stream), not by launches of their own - at one frame per call the step is
	// bound by launch boundaries, not by HBM.
	// Hand-off to the last arriver: every word it reads was written by agent-scope atomic adds (executed at the memory side),
	// each adding wave drains them (s_waitcnt vmcnt(0)) before the workgroup's barrier and the ticket, and the last arriver reads
	// them with agent-scope (sc1) loads.
	// The step of this workgroup's stream: the kernel argument (one stream) or entry blockIdx.y of the table (TABLE).  Two
	// instantiations, not a run-time choice: a select between the kernel-argument segment and global memory turns the struct -
	// and every pointer in it - generic, and the whole kernel into flat_ loads, stores and atomics (3x slower, measured).
	// Pointers read from the table are used through global-address-space copies (the compiler cannot know what they point to).
#define RIR_GLOBAL(T) __attribute__((address_space(1))) T
	typedef unsigned int lossy_v4u __attribute__((ext_vector_type(4)));
	template <class T>
	__device__ __forceinline__ RIR_GLOBAL(T) * as_global(T *p)
	{
		return (RIR_GLOBAL(T) *)p;
	}
	// a plain struct out of a table in global memory, in 8-byte words (wave-uniform addresses: scalar loads)
	template <class T>
	__device__ __forceinline__ T lossy_load_struct(const T *p_)
	{
		static_assert(sizeof(T) % 8 == 0, "copied in 8-byte words");
		T v;
		RIR_GLOBAL(const unsigned long long) *src = (RIR_GLOBAL(const unsigned long long) *)p_;
		unsigned long long *dst = reinterpret_cast<unsigned long long *>(&v);
#pragma unroll
		for (size_t k = 0; k < sizeof(T) / 8; ++k)
			dst[k] = src[k];
		return v;
	}
	template <bool TABLE>
	__device__ __forceinline__ LossyStep lossy_step_of(const LossyStep &one, const LossyStep *__restrict__ table)
	{
		if (!TABLE)
			return one;
		static_assert(sizeof(LossyStep) % 8 == 0, "LossyStep is copied in 8-byte words");
		LossyStep p;
		RIR_GLOBAL(const unsigned long long) *src = (RIR_GLOBAL(const unsigned long long) *)(table + blockIdx.y); // (wave-uniform: scalar loads)
		unsigned long long *dst = reinterpret_cast<unsigned long long *>(&p);
#pragma unroll
		for (size_t k = 0; k < sizeof(LossyStep) / 8; ++k)
			dst[k] = src[k];
		return p;
	}
	// Sum of a 64-bit integer over the wave, the same value in every lane: four DPP steps inside each row of 16 lanes (xor 1, xor 2,
	// mirror of 8, mirror of 16 - integer sums do not care about the pairing), then the four rows through readlane.  A butterfly
	// of __shfl_xor is six ds_bpermute round trips per half: six of those per frame were 1.3 us of the resident kernel's frame.
	__device__ __forceinline__ long long lossy_wave_sum(long long v)
	{
#define RIR_DPP_ADD(CTRL)                                                                                     \
	{                                                                                                         \
		const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(unsigned long long)v, CTRL, 0xf, 0xf, false); \
		const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)((unsigned long long)v >> 32), CTRL, 0xf, 0xf, false); \
		v += (long long)(((unsigned long long)(unsigned int)hi_ << 32) | (unsigned int)lo_);                  \
	}
		RIR_DPP_ADD(0xB1)  // quad_perm [1,0,3,2]
		RIR_DPP_ADD(0x4E)  // quad_perm [2,3,0,1]
		RIR_DPP_ADD(0x141) // row_half_mirror
		RIR_DPP_ADD(0x140) // row_mirror
#undef RIR_DPP_ADD
		long long r = 0;
#pragma unroll
		for (int row = 0; row < 4; ++row)
		{
			const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)v, row * 16);
			const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)v >> 32), row * 16);
			r += (long long)(((unsigned long long)hi << 32) | lo);
		}
		return r;
	}

	// the same for a 32-bit integer (wrapping): four DPP adds, four readlanes - a third of the 64-bit form's instructions
	__device__ __forceinline__ uint32_t lossy_wave_sum32(uint32_t v)
	{
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); // row_half_mirror
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false); // row_mirror
		return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) +
			   (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
	}

	// true in every thread of the workgroup whose ticket is the last of `expected`; last_flag: one LDS word the caller can spare
	__device__ __forceinline__ bool lossy_last_arriver(unsigned int *ticket_, unsigned int expected, unsigned int *last_flag)
	{
		RIR_GLOBAL(unsigned int) *ticket = as_global(ticket_);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's atomic adds have been performed
		__syncthreads();
		if (threadIdx.x == 0)
		{
			const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			*last_flag = (t == expected - 1) ? 1u : 0u;
			if (t == expected - 1)
				__hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next frame
		}
		__syncthreads();
		return *last_flag != 0;
	}

	// L1: histogram of (v >> 2) over 16 384 bins, privatised in LDS (64 KiB), 1024-thread workgroups,
	// hist_px pixels per workgroup (lossy_kernels.h: one frame is a small job - 32 768 pixels per workgroup left it on ten CUs and
	// took 17 us - while 32 streams at 4 096 pixels are five rounds of workgroups that each clear and scan 64 KiB; the merge only
	// touches the bins a workgroup has filled, a few hundred for thermal images).
	// The last workgroup to arrive takes the mode of the merged histogram, lowest bin wins ties:
	// stats[0] = background = (bin << 2) + 1 (get_background, h264.cpp:1955-1991), and clears the histogram.
	// smallest value over the wave, the same in every lane (as lossy_wave_sum32: four DPP steps in the rows, the rows through readlane)
	__device__ __forceinline__ uint32_t lossy_wave_min32(uint32_t v)
	{
		v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));
		v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));
		v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));
		v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));
		return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
				   min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
	}
	// The 8 pixels of every lane into the workgroup's LDS histogram.  offs: the pixels' bins as BYTE offsets into it (bin * 4: one mask per
	// pixel, v & 0xfffc, and no shift at the atomic).
	// Thermal scenes are flat - the motion-corrected registration stream of configs[4] has ~20 levels, five or six bins, per frame - and
	// LDS atomics on a handful of addresses serialise lane by lane (scripts/ubench/lds_atomic_rate.hip: 54 ns a wave instruction on one
	// address, 9 on sixteen, 4.7 on 250).  So every wave keeps a WINDOW of sixteen bins (64 levels), anchored a little below the smallest of
	// its pixels since it last emptied it, in registers: a round's eight pixels are counted in the 4-bit fields of two words (the bin's byte
	// offset in a word IS the shift: four instructions a pixel and word), the nibbles are spread into words of four 8-bit counts (five
	// instructions a round and word), and those are added up over the wave and put into the histogram every 31 rounds (8 pixels a round: 248) and at the end
	// of the frame.  No atomic, nothing across lanes, and no 64-bit integer instruction in the round (the first form of this - four 16-bit
	// fields in a 64-bit word - spent its time in v_lshlrev_b64: 2 x slower on flat frames than the plain atomics on spread ones).
	// A round in which more than a quarter of the lanes see pixels outside the window is spread data (one ballot to find out) and goes
	// through plain atomics, none of them under a branch - and so do the seven rounds after it, without a look at the window; the few
	// pixels outside the window in other rounds do too.  Same counts either way.
	constexpr int kHistWindowWords = 2; // the window: 8 bins (32 levels) to a word
	struct LossyHistWindow
	{
		uint32_t even[kHistWindowWords], odd[kHistWindowWords]; // this lane's counts of the bins 0, 2, 4, 6 / 1, 3, 5, 7 of each word of the window: 8 bits each
		uint32_t base;											 // byte offset of the window's first bin (wave-uniform); set whenever the counts are empty
		int rounds;												 // rounds in the counts (wave-uniform)
		int skip;												 // rounds still to go straight to the atomics after a round of spread data (wave-uniform)
	};
	__device__ __forceinline__ uint32_t *lossy_hist_word(uint32_t *lh, uint32_t off) { return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(lh) + off); }
	__device__ __forceinline__ void lossy_hist_window_flush(uint32_t *lh, LossyHistWindow &hw)
	{ // (wave-uniform call) 8-bit counts -> 16-bit fields, two to a word; over the wave: <= 64 lanes x 248
		const int lane = (int)(threadIdx.x & 63);
#pragma unroll
		for (int wd = 0; wd < kHistWindowWords; ++wd)
		{
			const uint32_t b04 = lossy_wave_sum32(hw.even[wd] & 0x00ff00ffu), b26 = lossy_wave_sum32((hw.even[wd] >> 8) & 0x00ff00ffu);
			const uint32_t b15 = lossy_wave_sum32(hw.odd[wd] & 0x00ff00ffu), b37 = lossy_wave_sum32((hw.odd[wd] >> 8) & 0x00ff00ffu);
			if (lane < 8)
			{ // lane = bin of this word of the window
				const uint32_t v = (lane & 1) ? ((lane & 2) ? b37 : b15) : ((lane & 2) ? b26 : b04);
				const uint32_t c = (lane & 4) ? v >> 16 : v & 0xffffu;
				if (c)
					atomicAdd(lossy_hist_word(lh, hw.base + 32u * (uint32_t)wd + 4u * (uint32_t)lane), c);
			}
			hw.even[wd] = hw.odd[wd] = 0u;
		}
		hw.rounds = 0;
	}
	__device__ __forceinline__ void lossy_hist_add8(uint32_t *lh, const uint32_t *offs, bool in, LossyHistWindow &hw)
	{
		const unsigned long long votes = __ballot(in);
		if (!votes)
			return;
		const uint32_t one = in ? 1u : 0u;
		if (__builtin_amdgcn_readfirstlane(hw.skip) > 0)
		{ // spread data a moment ago: the next look at the window is a few rounds away (it costs ~100 instructions that such a frame gets nothing for)
			hw.skip = __builtin_amdgcn_readfirstlane(hw.skip) - 1;
#pragma unroll
			for (int k = 0; k < 8; ++k)
				atomicAdd(lossy_hist_word(lh, offs[k]), one);
			return;
		}
		if (__builtin_amdgcn_readfirstlane(hw.rounds) == 0)
		{ // nothing in the registers: the window is anchored anew, four bins below the (four-bin group of the) smallest pixel of this round
			uint32_t mn = 0xffffffffu;
#pragma unroll
			for (int k = 0; k < 8; ++k)
				mn = min(mn, offs[k]);
			mn = lossy_wave_min32(in ? mn : 0xffffffffu) & ~15u;
			hw.base = mn >= 16u ? mn - 16u : 0u;
		}
		uint32_t nib[kHistWindowWords], inside_all = 1u;
#pragma unroll
		for (int wd = 0; wd < kHistWindowWords; ++wd)
			nib[wd] = 0u;
#pragma unroll
		for (int k = 0; k < 8; ++k)
		{
			const uint32_t rel = offs[k] - hw.base; // (below the window: wraps to a large number)
			uint32_t inw = 0u;
#pragma unroll
			for (int wd = 0; wd < kHistWindowWords; ++wd)
			{
				const uint32_t r = rel - 32u * (uint32_t)wd, here = r < 32u ? 1u : 0u;
				nib[wd] += here << (r & 31u); // bin r / 4 of this word: four bits each, at most 8 in a field
				inw |= here;
			}
			inside_all &= inw;
		}
		const unsigned long long strays = __ballot(in && !inside_all);
		if (__builtin_popcountll(strays) * 4 > __builtin_popcountll(votes))
		{ // spread data: eight atomics, none under a branch (a lane past the end adds 0 - to bin 0, where its zero pixels point)
#pragma unroll
			for (int k = 0; k < 8; ++k)
				atomicAdd(lossy_hist_word(lh, offs[k]), one);
			hw.skip = 7;
			return;
		}
#pragma unroll
		for (int wd = 0; wd < kHistWindowWords; ++wd)
		{
			const uint32_t nb = in ? nib[wd] : 0u;
			hw.even[wd] += nb & 0x0f0f0f0fu, hw.odd[wd] += (nb >> 4) & 0x0f0f0f0fu;
		}
		if (strays)
		{
#pragma unroll
			for (int k = 0; k < 8; ++k)
				if (in && offs[k] - hw.base >= 32u * kHistWindowWords)
					atomicAdd(lossy_hist_word(lh, offs[k]), 1u);
		}
		hw.rounds = __builtin_amdgcn_readfirstlane(hw.rounds) + 1;
		if (hw.rounds == 31)
			lossy_hist_window_flush(lh, hw);
	}

	// what the histogram pass needs of a frame: its pixels, a zeroed 16 384-bin slice, where its background goes, a zeroed ticket
	struct LossyHistJob
	{
		const uint16_t *tmp;
		uint32_t *hist;
		long long *stats;
		unsigned int *tickets;
		int s, hist_px;
	};
	// NT: the pixels are loaded with the non-temporal policy - for launches over more frames than the Infinity Cache holds (the streaming kernel that
	// follows reads them from HBM either way: 1 000 frames of one 640x512 stream +5 % on the whole call); shorter groups keep the default policy,
	// under which the streaming kernel finds the frames in the cache (50-frame groups: 2 % slower with nt)
	template <bool NT = false>
	__device__ __forceinline__ void lossy_hist_mode_body(const LossyHistJob &sp, uint32_t *lh)
	{
		constexpr int kAux = NT ? 2 : 0;
		const int tid = threadIdx.x;
		RIR_GLOBAL(const uint16_t) *tmp = as_global(sp.tmp);
		RIR_GLOBAL(uint32_t) *hist = as_global(sp.hist);
		const int s = sp.s;
		for (int i = tid; i < 16384; i += 1024)
			lh[i] = 0;
		__syncthreads();
		const int i0 = blockIdx.x * sp.hist_px, i1 = min(i0 + sp.hist_px, s);
		if (((i0 | i1) & 7) == 0)
		{ // 8 pixels per 16-byte load, kHistAhead loads of a thread in flight: with one (load, count, load, ...) a workgroup that takes a
		  // whole frame - a run of 1 000 frames is 1 000 such workgroups, two to a CU - waited a memory latency per 16 KB: 32 KB in flight
		  // per CU, 3.9 TB/s over the chip.  Unconditional buffer loads (a lane past the end: an offset out of range, zeros it does not
		  // count), so that the waits stay counted (s_waitcnt vmcnt(N)).
			constexpr int kHistAhead = 4;
			const int n8 = i1 / 8;
			const uint32_t bytes = (uint32_t)n8 * 16u; // (s < 2^27 pixels: lossy_state_create)
			const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(sp.tmp), 0, (int)bytes, 0x00020000);
			lossy_v4u q[kHistAhead];
			LossyHistWindow hw = {};
			int ib = i0 / 8;
#pragma unroll
			for (int a = 0; a < kHistAhead; ++a)
			{ // (in this order, as the loop asks again: the wait at its top is one instruction for both ways in)
				q[a] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(ib + a * 1024 + tid) * 16u, 0, kAux);
				__builtin_amdgcn_sched_barrier(0);
			}
			for (; ib < n8; ib += 1024 * kHistAhead)
			{ // (wave-uniform trip counts: lossy_hist_add8 votes across the wave)
#pragma unroll
				for (int a = 0; a < kHistAhead; ++a)
				{ // (a round past the end: no lane is `in`, nothing is counted)
					const lossy_v4u v = q[a];
					const bool in = ib + a * 1024 + tid < n8;
					uint32_t bins[8] = {v.x & 0xfffcu, (v.x >> 16) & 0xfffcu, v.y & 0xfffcu, (v.y >> 16) & 0xfffcu, v.z & 0xfffcu, (v.z >> 16) & 0xfffcu, v.w & 0xfffcu, (v.w >> 16) & 0xfffcu}; // (as byte offsets)
					// the slot is asked for again when the pixels in it have become bins (here, not where a bin is first used - the empty asm pins
					// that): asked for earlier, old and new are alive together and the compiler rotates the slots through a fifth, with copies at
					// the top of the loop - after waiting for every load
#pragma unroll
					for (int k = 0; k < 8; ++k)
						asm volatile("" : "+v"(bins[k]));
					__builtin_amdgcn_sched_barrier(0);
					q[a] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(ib + (a + kHistAhead) * 1024 + tid) * 16u, 0, kAux);
					__builtin_amdgcn_sched_barrier(0);
					lossy_hist_add8(lh, bins, in, hw);
				}
			}
			if (hw.rounds)
				lossy_hist_window_flush(lh, hw);
		}
		else
			for (int i = i0 + tid; i < i1; i += 1024) // (frames whose lossy part is not a multiple of 8 pixels: one pixel per lane)
				atomicAdd(&lh[tmp[i] >> 2], 1u);
		__syncthreads();
		uint32_t v16[16];
		const bool alone = gridDim.x == 1; // the workgroup has the whole frame: its private histogram IS the frame's (nothing to merge, nothing to clear)
		if (alone)
		{
#pragma unroll
			for (int k = 0; k < 16; ++k)
				v16[k] = lh[k * 1024 + tid];
		}
		else
		{
			for (int i = tid; i < 16384; i += 1024)
				if (lh[i])
					__hip_atomic_fetch_add(&hist[i], lh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (!lossy_last_arriver(sp.tickets, gridDim.x, &lh[0])) // (the private histogram has been merged: its first word is free)
				return;
			// thread t looks at bins t, t + 1024, ... (coalesced: these are agent-scope loads, each one goes to L2 - 16 strided loads per thread cost 10 us)
#pragma unroll
			for (int k = 0; k < 16; ++k)
				v16[k] = __hip_atomic_load(&hist[k * 1024 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		// mode: ascending, strict > keeps the lowest bin; then a tree over the threads on (count, bin)
		uint32_t *best_v = lh, *best_i = lh + 1024;
		uint32_t bv = 0, bi = 0;
#pragma unroll
		for (int k = 0; k < 16; ++k)
		{
			if (!alone)
				hist[k * 1024 + tid] = 0; // ready for the next frame (the histogram is cleared once, when the state is created)
			if (k == 0 || v16[k] > bv)
			{
				bv = v16[k];
				bi = (uint32_t)(k * 1024 + tid);
			}
		}
		__syncthreads(); // (lh is being reused)
		best_v[tid] = bv;
		best_i[tid] = bi;
		__syncthreads();
		for (int d = 512; d >= 1; d >>= 1)
		{
			if (tid < d)
			{
				const uint32_t ov = best_v[tid + d], oi = best_i[tid + d];
				if (ov > best_v[tid] || (ov == best_v[tid] && oi < best_i[tid]))
				{
					best_v[tid] = ov;
					best_i[tid] = oi;
				}
			}
			__syncthreads();
		}
		// Does the frame surely have pixels on BOTH sides of its background ((bin << 2) + 1: every pixel of a higher bin is above it,
		// every pixel of a lower bin below; the mode bin itself holds both kinds)?  The constant-budget form of a run
		// (lossy_const_run_kernel) only takes frames whose two classes cannot be empty - an empty class makes the reference's statistic
		// 0 / 0, and that NaN decides the budgets of the next 40 frames (int_of_double_x86 below).  Bit 40 of the word = "not sure";
		// every reader of the background takes the low 32 bits.
		const uint32_t mode_bin = best_i[0], mode_count = best_v[0];
		if (tid == 0)
			lh[2048] = 0;
		__syncthreads();
		uint32_t below = 0;
#pragma unroll
		for (int k = 0; k < 16; ++k)
			below += (uint32_t)(k * 1024 + tid) < mode_bin ? v16[k] : 0u;
		below = lossy_wave_sum32(below);
		if ((tid & 63) == 0 && below)
			atomicAdd(&lh[2048], below);
		__syncthreads();
		if (tid == 0)
		{
			const uint32_t nb_ = lh[2048], na_ = (uint32_t)s - nb_ - mode_count;
			const long long unsure = (nb_ == 0 || na_ == 0) ? (1ll << 40) : 0ll;
			as_global(sp.stats)[0] = (long long)((mode_bin << 2) + 1) | unsure;
		}
	}
	template <bool TABLE>
	__global__ __launch_bounds__(1024) void lossy_hist_mode_kernel(LossyStep one, const LossyStep *__restrict__ table)
	{
		__shared__ uint32_t lh[16384];
		const LossyStep sp = lossy_step_of<TABLE>(one, table);
		const LossyHistJob job = {sp.tmp, sp.hist, sp.stats, sp.tickets, sp.s, sp.hist_px};
		lossy_hist_mode_body(job, lh);
	}
	// The backgrounds of a group of frames of a run, straight from the runs' descriptions (no per-frame table to build and copy): entry
	// blockIdx.y = frame k of stream i of the group (k major), its histogram slice and ticket the entry's own.
	template <bool NT>
	__global__ __launch_bounds__(1024) void lossy_hist_mode_runs_kernel(const LossyRun *__restrict__ runs, int nstreams, uint32_t *__restrict__ hist_base,
																		unsigned int *__restrict__ tick_base, int s, int hist_px)
	{
		__shared__ uint32_t lh[16384];
		const int e = blockIdx.y, k = e / nstreams, i = e - k * nstreams;
		RIR_GLOBAL(const LossyRun) *r = as_global(runs + i);
		const LossyHistJob job = {r->in + (size_t)k * r->frame_px, hist_base + (size_t)e * 16384, const_cast<long long *>(r->bg) + (size_t)k * r->bg_stride, tick_base + e, s, hist_px};
		lossy_hist_mode_body<NT>(job, lh);
	}

	// (int) of a double as the reference's x86-64 build converts it (cvttsd2si): NaN and values outside int32 give
	// INT_MIN.  It matters: with an empty foreground or background the statistic is 0/0, the NaN stays in the 40-frame
	// window, and the budgets of those frames are whatever this conversion and the wrapping subtraction below make of it
	// (0 / 0 errors: lossless frames) - the device's own conversion (NaN -> 0) would leave the configured errors instead.
	__device__ __forceinline__ int int_of_double_x86(double v) { return (v >= -2147483648.0 && v < 2147483648.0) ? (int)v : (int)0x80000000; }
	__device__ __forceinline__ int sub_wrap(int a, int b) { return (int)((unsigned)a - (unsigned)b); }

	// The error budget of the frame (h264.cpp:2335-2385, :2544-2548 for addLoss): the reference's statistic
	// sqrt((sum d)^2 - sum d^2) / n of this frame against its mean over a 40-frame window, scaled by stdFactor, is
	// taken off lowValueError / highValueError.  Sequential scalar double arithmetic on exact integer sums - one
	// thread, same operations in the same order as the host code it replaces (sqrt, division and round of doubles
	// are IEEE-exact on the device: scripts/ubench/sqrt_f64_check.hip, 1.3e8 samples; this file is compiled with
	// -ffp-contract=off).  Keeping it here means no statistics travel to the host between the kernels of a frame.
	// st[1..6]: the frame's sums as read by the caller.
	// b: the stream's budget state, staged in LDS by the caller (the 40-entry window is shifted and summed element by element:
	// from global memory that was 160 dependent round trips, 8 us).
	struct LossyBudgetParams
	{
		int s, add_loss, low_value_error, high_value_error;
		double std_factor;
	};
	__device__ __forceinline__ LossyDecision lossy_budget(const LossyBudgetParams &sp, const long long *st, LossyBudget &b)
	{
		const int s = sp.s;
		// stdDev (h264.cpp:1993-2036): unsplit for the first 40 frames
		double sd[2];
		if (b.n_win < 40)
		{
			const double sum_diff = (double)(st[1] + st[4]), sum_diff2 = (double)(st[2] + st[5]);
			sd[0] = sd[1] = sqrt(sum_diff * sum_diff - sum_diff2) / s;
		}
		else
		{
			const double fd = (double)st[1], fd2 = (double)st[2], bd = (double)st[4], bd2 = (double)st[5];
			sd[0] = sqrt(bd * bd - bd2) / (int)st[6];
			sd[1] = sqrt(fd * fd - fd2) / (int)st[3];
		}
		if (b.n_first < 1)
		{
			b.first_std[0] = sd[0], b.first_std[1] = sd[1];
			b.n_first = 1;
		}
		if (b.n_win < 40)
		{
			b.win[b.n_win][0] = sd[0], b.win[b.n_win][1] = sd[1];
			++b.n_win;
		}
		else
		{ // the oldest entry makes room (the reference shifts the window; 78 dependent LDS round trips here - a ring instead)
			b.win[b.head][0] = sd[0], b.win[b.head][1] = sd[1];
			b.head = b.head == 39 ? 0 : b.head + 1;
		}
		// summed oldest to newest, as the reference does
		double mean[2] = {b.first_std[0], b.first_std[1]};
		if (b.n_win < 40)
			for (int i = 0; i < b.n_win; ++i)
			{
				mean[0] += b.win[i][0];
				mean[1] += b.win[i][1];
			}
		else
		{
			const int h0 = b.head;
#pragma unroll 4
			for (int i = 0; i < 40; ++i)
			{
				const int j = h0 + i < 40 ? h0 + i : h0 + i - 40;
				mean[0] += b.win[j][0];
				mean[1] += b.win[j][1];
			}
		}
		mean[0] /= (double)(b.n_win + b.n_first);
		mean[1] /= (double)(b.n_win + b.n_first);
		int low_error = sp.low_value_error, high_error = sp.high_value_error;
		if (sp.add_loss)
		{ // one-sided
			const double dh = sd[1] < mean[1] ? 0 : sd[1] - mean[1], dl = sd[0] < mean[0] ? 0 : sd[0] - mean[0];
			high_error = sub_wrap(high_error, int_of_double_x86(round(dh * sp.std_factor)));
			low_error = sub_wrap(low_error, int_of_double_x86(round(dl * sp.std_factor)));
		}
		else
		{ // two-sided
			high_error = sub_wrap(high_error, int_of_double_x86(round(fabs(sd[1] - mean[1]) * sp.std_factor)));
			low_error = sub_wrap(low_error, int_of_double_x86(round(fabs(sd[0] - mean[0]) * sp.std_factor)));
		}
		if (high_error < 0)
			high_error = 0;
		if (low_error < high_error)
			low_error = high_error;
		LossyDecision d;
		d.background = (uint32_t)st[0];
		d.low_error = low_error;
		d.high_error = high_error;
		d.reserved = 0;
		return d;
	}
	__device__ __forceinline__ void lossy_budget(const LossyStep &sp, const long long *st, LossyBudget &b, int *errors_out)
	{
		const LossyBudgetParams bp = {sp.s, sp.add_loss, sp.low_value_error, sp.high_value_error, sp.std_factor};
		const LossyDecision d = lossy_budget(bp, st, b);
		RIR_GLOBAL(LossyDecision) *decision = as_global(sp.decision);
		decision->background = d.background;
		decision->low_error = d.low_error;
		decision->high_error = d.high_error;
		if (errors_out)
		{
			as_global(errors_out)[0] = d.low_error;
			as_global(errors_out)[1] = d.high_error;
		}
	}

	// The same budget by TWO lanes of a wave, lane c = component c (0: background / low error, 1: foreground / high error), each
	// doing exactly the operations lossy_budget does for its component, in the same order.  Split in two so that the sum over the
	// window - 40 dependent additions that do not involve the new frame - is out of the way before the frame's sums arrive:
	// lossy_budget2_prepare (any time after the previous frame's finish), then lossy_budget2_finish.  Both lanes call; b in LDS.
	// (The 39 additions in a row - the ORDER is the reference's and cannot change - are 1.1-1.5 us of the leader's 6 us per frame
	// (RIR_LOSSY_DIAG: "window sum").  Asking for 13 window entries at a time instead of one brought the sum itself from 1.5 to 1.1 us and cost
	// the rest of the frame more - 155 k frames/s against 163 k: the kernel has no registers to spare.  Not kept.)
	__device__ __forceinline__ double lossy_budget2_prepare(const LossyBudget &b, int c)
	{
		double part = b.first_std[c]; // first + the window entries that stay, oldest to newest (the new entry is added last)
		if (b.n_win < 40)
			for (int i = 0; i < b.n_win; ++i)
				part += b.win[i][c];
		else
		{
			const int h0 = b.head;
#pragma unroll 4
			for (int i = 1; i < 40; ++i)
			{
				const int j = h0 + i < 40 ? h0 + i : h0 + i - 40;
				part += b.win[j][c];
			}
		}
		return part;
	}
	__device__ __forceinline__ LossyDecision lossy_budget2_finish(const LossyBudgetParams &sp, const long long *st, LossyBudget &b, int c, double part)
	{
		double sd;
		if (b.n_win < 40)
		{
			const double sum_diff = (double)(st[1] + st[4]), sum_diff2 = (double)(st[2] + st[5]);
			sd = sqrt(sum_diff * sum_diff - sum_diff2) / sp.s;
		}
		else
		{
			const double d = (double)(c ? st[1] : st[4]), d2 = (double)(c ? st[2] : st[5]);
			sd = sqrt(d * d - d2) / (int)(c ? st[3] : st[6]);
		}
		const bool first = b.n_first < 1; // (the very first budget of a stream: `part` was taken before first_std existed)
		if (first)
			b.first_std[c] = sd;
		const int n_win = b.n_win, head = b.head;
		double mean;
		if (n_win < 40)
		{
			b.win[n_win][c] = sd;
			mean = (first ? sd : part) + sd;
		}
		else
		{
			b.win[head][c] = sd;
			mean = part + sd;
		}
		const int n_win_after = n_win < 40 ? n_win + 1 : 40;
		mean /= (double)(n_win_after + 1);
		const int base = c ? sp.high_value_error : sp.low_value_error;
		const double dd = sp.add_loss ? (sd < mean ? 0 : sd - mean) : fabs(sd - mean);
		const int e = sub_wrap(base, int_of_double_x86(round(dd * sp.std_factor)));
		int high_error = __shfl(e, 1, 64), low_error = __shfl(e, 0, 64);
		if (c == 0)
		{ // (one lane moves the counters; the other has read them)
			b.n_first = 1;
			if (n_win < 40)
				b.n_win = n_win + 1;
			else
				b.head = head == 39 ? 0 : head + 1;
		}
		if (high_error < 0)
			high_error = 0;
		if (low_error < high_error)
			low_error = high_error;
		LossyDecision d;
		d.background = (uint32_t)st[0];
		d.low_error = low_error;
		d.high_error = high_error;
		d.reserved = 0;
		return d;
	}

	// L2: sums of |t - prev| and of its (32-bit wrapped) square, split by img > background.
	// stats[1..6] = {fg sum d, fg sum d2, fg count, bg sum d, bg sum d2, bg count}; the last workgroup to arrive turns
	// them into the frame's error budget (lossy_budget) and clears them for the next frame.
	template <bool TABLE>
	__global__ __launch_bounds__(256) void lossy_sums_budget_kernel(LossyStep one, const LossyStep *__restrict__ table)
	{
		const LossyStep sp = lossy_step_of<TABLE>(one, table);
		RIR_GLOBAL(const uint16_t) *prevT = as_global((const uint16_t *)sp.st.prevT), *tmp = as_global(sp.tmp), *img = as_global(sp.img);
		RIR_GLOBAL(long long) *stats = as_global(sp.stats);
		const int s = sp.s, subtract_min = sp.st.subtract_min;
		const uint32_t mn = sp.st.min;
		const uint32_t background = (uint32_t)stats[0];
		long long a[6] = {0, 0, 0, 0, 0, 0};
		auto pixel = [&](uint32_t tv, uint32_t pv, uint32_t iv) {
			const uint32_t t = subtract_min ? sub_min(tv, mn) : tv;
			const int32_t d = abs((int32_t)t - (int32_t)pv);
			const int32_t d2 = (int32_t)((uint32_t)d * (uint32_t)d);
			const int o = iv > background ? 0 : 3;
			a[o] += d;
			a[o + 1] += d2;
			a[o + 2] += 1;
		};
		if ((s & 7) == 0)
		{ // 8 pixels per 16-byte load
			for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s / 8; i += gridDim.x * blockDim.x)
			{
				const lossy_v4u tv = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(tmp + (size_t)i * 8);
				const lossy_v4u pv = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(prevT + (size_t)i * 8);
				const lossy_v4u iv = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(img + (size_t)i * 8);
#pragma unroll
				for (int k = 0; k < 4; ++k)
				{
					pixel(tv[k] & 0xffffu, pv[k] & 0xffffu, iv[k] & 0xffffu);
					pixel(tv[k] >> 16, pv[k] >> 16, iv[k] >> 16);
				}
			}
		}
		else
			for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s; i += gridDim.x * blockDim.x)
				pixel(tmp[i], prevT[i], img[i]);
		// wave reduction, then the four waves of the block through LDS: one atomic per block and sum (a few hundred
		// in total - one per wave made 24 000 contended 64-bit atomics and cost 0.29 ms per frame)
		__shared__ long long red[4][6];
#pragma unroll
		for (int k = 0; k < 6; ++k)
		{
			const long long v = lossy_wave_sum(a[k]);
			if ((threadIdx.x & 63) == 0)
				red[threadIdx.x >> 6][k] = v;
		}
		__syncthreads();
		if (threadIdx.x < 6)
		{
			const long long v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
			if (v)
				__hip_atomic_fetch_add((RIR_GLOBAL(unsigned long long) *)&stats[1 + threadIdx.x], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (!lossy_last_arriver(sp.tickets + 1, gridDim.x, reinterpret_cast<unsigned int *>(&red[0][0])))
			return;
		// the six sums: one lane each (one round trip, not six), handed to thread 0 through LDS
		if (threadIdx.x < 6)
		{
			red[1][threadIdx.x] = (long long)__hip_atomic_load((RIR_GLOBAL(unsigned long long) *)&stats[1 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			stats[1 + threadIdx.x] = 0; // the sums are accumulated with atomics: cleared for the next frame
		}
		// the budget state comes to LDS in one coalesced load, is worked on there by one thread, and goes back the same way
		__shared__ LossyBudget bl;
		static_assert(sizeof(LossyBudget) % 8 == 0 && sizeof(LossyBudget) / 8 <= 256, "LossyBudget is moved by one 8-byte word per thread");
		RIR_GLOBAL(unsigned long long) *gb = (RIR_GLOBAL(unsigned long long) *)as_global(sp.budget);
		unsigned long long *lb = reinterpret_cast<unsigned long long *>(&bl);
		if (threadIdx.x < sizeof(LossyBudget) / 8)
			lb[threadIdx.x] = gb[threadIdx.x];
		__syncthreads();
		if (threadIdx.x == 0)
		{
			long long st[7];
			st[0] = stats[0];
			for (int i = 1; i < 7; ++i)
				st[i] = red[1][i - 1];
			lossy_budget(sp, st, bl, sp.errors_out);
		}
		__syncthreads();
		if (threadIdx.x < sizeof(LossyBudget) / 8)
			gb[threadIdx.x] = lb[threadIdx.x];
	}

	// L3 + L4: running average update and decision loop, one thread per pixel of the whole frame.
	template <bool TABLE>
	__global__ __launch_bounds__(256) void lossy_update_kernel(LossyStep one, const LossyStep *__restrict__ table)
	{
		const LossyStep sp = lossy_step_of<TABLE>(one, table);
		RIR_GLOBAL(const uint16_t) *tmp = as_global(sp.tmp);
		RIR_GLOBAL(uint16_t) *out = as_global(sp.out);
		const LossyDeviceState st_ = sp.st;
		struct
		{ // the state's arrays through global pointers; the scalars as they are
			RIR_GLOBAL(uint16_t) * refT, *prevT, *lastDL, *ra_const_value, *ra_images;
			RIR_GLOBAL(uint32_t) * ra_sums;
			RIR_GLOBAL(int16_t) * ra_const_count;
			int ra_count, ra_head, running_average, subtract_min;
			uint32_t min;
		} st = {as_global(st_.refT), as_global(st_.prevT), as_global(st_.lastDL), as_global(st_.ra_const_value), as_global(st_.ra_images),
				as_global(st_.ra_sums), as_global(st_.ra_const_count), st_.ra_count, st_.ra_head, st_.running_average, st_.subtract_min, st_.min};
		RIR_GLOBAL(const LossyDecision) *decision = as_global((const LossyDecision *)sp.decision);
		const int s = sp.s, full = sp.full, add_loss = sp.add_loss;
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= full)
			return;
		const uint32_t background = decision->background;
		const int low_error = decision->low_error, high_error = decision->high_error;
		const uint32_t v = tmp[i];
		if (i >= s)
		{ // rows past lossy_height: stored as they are
			out[i] = (uint16_t)v;
			st.lastDL[i] = (uint16_t)v;
			return;
		}
		uint32_t t = st.subtract_min ? sub_min(v, st.min) : v;
		const int ra = st.running_average;
		uint32_t sum = 0;
		if (ra > 0)
		{ // RunningAverage2::addImage
			sum = st.ra_sums[i] + t;
			if (st.ra_count == ra)
			{
				const int16_t cc = st.ra_const_count[i];
				if (cc)
				{
					st.ra_const_count[i] = (int16_t)(cc - 1);
					sum -= st.ra_const_value[i];
				}
				else
					sum -= st.ra_images[(size_t)st.ra_head * s + i];
			}
			// the new image takes the free slot (ring not full) or replaces the oldest one
			const int slot = (st.ra_count == ra) ? st.ra_head : (st.ra_head + st.ra_count) % ra;
			st.ra_images[(size_t)slot * s + i] = (uint16_t)t;
		}
		const int n_after = ra > 0 ? (st.ra_count == ra ? ra : st.ra_count + 1) : 0; // images.size() after addImage
		const uint32_t ref = st.refT[i];
		const int diff = abs((int)t - (int)ref);
		const int max_error = v > background ? high_error : low_error;
		bool keep = diff <= max_error;
		if (!add_loss)
			keep = keep && ((st.lastDL[i] >> 13) == (v >> 13));
		if (keep)
			t = ra > 0 ? sum / (uint32_t)n_after : ref;
		else
		{
			st.refT[i] = (uint16_t)t;
			if (ra > 0)
			{
				st.ra_const_value[i] = (uint16_t)t;
				st.ra_const_count[i] = (int16_t)n_after;
				sum = t * (uint32_t)n_after;
			}
		}
		if (ra > 0)
			st.ra_sums[i] = sum;
		out[i] = (uint16_t)t;
		st.prevT[i] = (uint16_t)t;
		st.lastDL[i] = (uint16_t)v;
	}

	// The same update, 8 consecutive pixels per thread through 16-byte loads and stores (used when the lossy region and the frame
	// are whole multiples of 8 pixels: every group is wholly inside or wholly past lossy_height).  One pixel per thread moves 2
	// bytes per lane and instruction: with 32 streams per launch that kernel ran at a third of the bandwidth this one reaches.
	struct U16x8
	{
		uint32_t d[4];
		__device__ __forceinline__ uint32_t get(int k) const { return (k & 1) ? d[k >> 1] >> 16 : d[k >> 1] & 0xffffu; }
		__device__ __forceinline__ void set(int k, uint32_t v) { d[k >> 1] = (k & 1) ? (d[k >> 1] & 0x0000ffffu) | (v << 16) : (d[k >> 1] & 0xffff0000u) | (v & 0xffffu); }
	};
	template <class P>
	__device__ __forceinline__ U16x8 ld8(P p, int i8)
	{
		const lossy_v4u v = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(p + (size_t)i8 * 8);
		U16x8 r;
		r.d[0] = v.x, r.d[1] = v.y, r.d[2] = v.z, r.d[3] = v.w;
		return r;
	}
	template <class P>
	__device__ __forceinline__ void st8(P p, int i8, const U16x8 &r)
	{
		lossy_v4u v;
		v.x = r.d[0], v.y = r.d[1], v.z = r.d[2], v.w = r.d[3];
		*reinterpret_cast<RIR_GLOBAL(lossy_v4u) *>(p + (size_t)i8 * 8) = v;
	}
	// L3 + L4 of ONE pixel (RunningAverage2::addImage h264.cpp:1526-1615, decision loop :2397-2413 / :2574-2590), on values in
	// registers: v = the frame's pixel, old = the ring's oldest image at this pixel (read only when the ring is full and the pixel
	// is not in a constant stretch).  Updates ref / sum / cc / cv, returns the stored pixel; *t_in = what enters the ring.
	struct LossyFrameConsts
	{
		uint32_t min, background;
		int subtract_min, ra, full_ring, n_after, add_loss, low_error, high_error;
		uint32_t div_magic; // lossy_div_magic(n_after)
	};
	// sum / n for sum < 2^22 (at most 64 images of 16 bits) and 1 <= n <= 64 without a division: floor(sum * M / 2^32), M =
	// floor(2^32 / n) + 1, is exact as long as sum * (M - 2^32 / n) / 2^32 < 1 / n, and sum / 2^32 < 2^-10 < 1 / 64.
	__device__ __forceinline__ uint32_t lossy_div_magic(int n) { return n > 1 ? (uint32_t)(0x100000000ull / (uint32_t)n) + 1u : 0u; }
	__device__ __forceinline__ uint32_t lossy_div(uint32_t sum, uint32_t magic) { return magic ? __umulhi(sum, magic) : sum; }
	__device__ __forceinline__ uint32_t lossy_pixel(const LossyFrameConsts &c, uint32_t v, uint32_t old, uint32_t last, uint32_t &ref, uint32_t &sum, uint32_t &cc,
												   uint32_t &cv, uint32_t *t_in, bool &ref_changed, bool &cc_changed)
	{
		uint32_t t = c.subtract_min ? sub_min(v, c.min) : v;
		*t_in = t;
		uint32_t sm = 0;
		if (c.ra > 0)
		{
			sm = sum + t;
			if (c.full_ring)
			{
				if (cc)
				{
					cc = cc - 1u;
					cc_changed = true;
					sm -= cv;
				}
				else
					sm -= old;
			}
		}
		const int diff = abs((int)t - (int)ref);
		const int max_error = v > c.background ? c.high_error : c.low_error;
		bool keep = diff <= max_error;
		if (!c.add_loss)
			keep = keep && ((last >> 13) == (v >> 13));
		if (keep)
			t = c.ra > 0 ? lossy_div(sm, c.div_magic) : ref;
		else
		{
			ref = t;
			ref_changed = true;
			if (c.ra > 0)
			{
				cv = t;
				cc = (uint32_t)c.n_after;
				cc_changed = true;
				sm = t * (uint32_t)c.n_after;
			}
		}
		sum = sm;
		return t;
	}

	// L3 + L4 for the 8 pixels of group i8 (inside the lossy rows), v8 = the frame's pixels: state arrays updated, the lossy pixels
	// stored to `out` (and to prevT when store_prev) and returned.
	__device__ __forceinline__ U16x8 lossy_update8(const LossyStep &sp, int i8, const U16x8 &v8, bool store_prev)
	{
		RIR_GLOBAL(uint16_t) *out = as_global(sp.out);
		const LossyDeviceState st = sp.st;
		RIR_GLOBAL(uint16_t) *refT = as_global(st.refT), *prevT = as_global(st.prevT), *lastDL = as_global(st.lastDL);
		RIR_GLOBAL(uint16_t) *cval = as_global(st.ra_const_value), *ring = as_global(st.ra_images);
		RIR_GLOBAL(uint16_t) *ccnt = (RIR_GLOBAL(uint16_t) *)as_global(st.ra_const_count);
		RIR_GLOBAL(uint32_t) *sums = as_global(st.ra_sums);
		RIR_GLOBAL(const LossyDecision) *decision = as_global((const LossyDecision *)sp.decision);
		const int s = sp.s, add_loss = sp.add_loss;
		const uint32_t background = decision->background;
		const int low_error = decision->low_error, high_error = decision->high_error;
		const int ra = st.running_average;
		const bool full_ring = ra > 0 && st.ra_count == ra;
		const int n_after = ra > 0 ? (full_ring ? ra : st.ra_count + 1) : 0; // images.size() after addImage
		U16x8 ref8 = ld8(refT, i8), last8, cc8, cv8, old8, t8, o8;
		uint32_t sum[8];
		if (!add_loss)
			last8 = ld8(lastDL, i8);
		if (ra > 0)
		{
			const lossy_v4u s0 = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(sums + (size_t)i8 * 8), s1 = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(sums + (size_t)i8 * 8 + 4);
			sum[0] = s0.x, sum[1] = s0.y, sum[2] = s0.z, sum[3] = s0.w, sum[4] = s1.x, sum[5] = s1.y, sum[6] = s1.z, sum[7] = s1.w;
			cc8 = ld8(ccnt, i8);
			cv8 = ld8(cval, i8);
			if (full_ring)
				old8 = ld8(ring + (size_t)st.ra_head * s, i8);
		}
		bool ref_changed = false, cc_changed = false; // (per group of 8: the arrays are written back only where a pixel changed them)
		const LossyFrameConsts fc = {st.min, background, st.subtract_min, ra, full_ring ? 1 : 0, n_after, add_loss, low_error, high_error, lossy_div_magic(n_after)};
#pragma unroll
		for (int k = 0; k < 8; ++k)
		{
			uint32_t ref = ref8.get(k), cc = ra > 0 ? cc8.get(k) : 0u, cv = ra > 0 ? cv8.get(k) : 0u, t_in;
			const uint32_t t = lossy_pixel(fc, v8.get(k), full_ring ? old8.get(k) : 0u, add_loss ? 0u : last8.get(k), ref, sum[k], cc, cv, &t_in, ref_changed,
										   cc_changed);
			ref8.set(k, ref);
			if (ra > 0)
				cc8.set(k, cc), cv8.set(k, cv);
			t8.set(k, t_in);
			o8.set(k, t);
		}
		if (ra > 0)
		{
			// the new image takes the free slot (ring not full) or replaces the oldest one
			const int slot = full_ring ? st.ra_head : (st.ra_head + st.ra_count) % ra;
			st8(ring + (size_t)slot * s, i8, t8);
			lossy_v4u s0, s1;
			s0.x = sum[0], s0.y = sum[1], s0.z = sum[2], s0.w = sum[3], s1.x = sum[4], s1.y = sum[5], s1.z = sum[6], s1.w = sum[7];
			*reinterpret_cast<RIR_GLOBAL(lossy_v4u) *>(sums + (size_t)i8 * 8) = s0;
			*reinterpret_cast<RIR_GLOBAL(lossy_v4u) *>(sums + (size_t)i8 * 8 + 4) = s1;
			if (cc_changed)
			{
				st8(ccnt, i8, cc8);
				st8(cval, i8, cv8);
			}
		}
		if (ref_changed)
			st8(refT, i8, ref8);
		st8(out, i8, o8);
		if (store_prev)
			st8(prevT, i8, o8);
		st8(lastDL, i8, v8);
		return o8;
	}

	template <bool TABLE>
	__global__ __launch_bounds__(256) void lossy_update_vec_kernel(LossyStep one, const LossyStep *__restrict__ table)
	{
		const LossyStep sp = lossy_step_of<TABLE>(one, table);
		const int i8 = blockIdx.x * blockDim.x + threadIdx.x; // group of 8 pixels
		if (i8 * 8 >= sp.full)
			return;
		const U16x8 v8 = ld8(as_global(sp.tmp), i8);
		if (i8 * 8 >= sp.s)
		{ // rows past lossy_height: stored as they are
			st8(as_global(sp.out), i8, v8);
			st8(as_global(sp.st.lastDL), i8, v8);
			return;
		}
		(void)lossy_update8(sp, i8, v8, true);
	}

	// The sums of a frame and, in the last workgroup to arrive, its budget: tail shared by lossy_sums_budget_kernel's form for
	// runs.  a[6]: this thread's share; every thread of the workgroup calls.
	__device__ __forceinline__ void lossy_sums_tail(const LossyStep &sp, const long long *a, long long background, int *errors_out)
	{
		RIR_GLOBAL(long long) *stats = as_global(sp.stats);
		__shared__ long long red[4][6];
		__shared__ LossyBudget bl;
#pragma unroll
		for (int k = 0; k < 6; ++k)
		{
			const long long v = lossy_wave_sum(a[k]);
			if ((threadIdx.x & 63) == 0)
				red[threadIdx.x >> 6][k] = v;
		}
		__syncthreads();
		if (threadIdx.x < 6)
		{
			const long long v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
			if (v)
				__hip_atomic_fetch_add((RIR_GLOBAL(unsigned long long) *)&stats[1 + threadIdx.x], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (!lossy_last_arriver(sp.tickets + 1, gridDim.x, reinterpret_cast<unsigned int *>(&red[0][0])))
			return;
		if (threadIdx.x < 6)
		{
			red[1][threadIdx.x] = (long long)__hip_atomic_load((RIR_GLOBAL(unsigned long long) *)&stats[1 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			stats[1 + threadIdx.x] = 0; // the sums are accumulated with atomics: cleared for the next frame
		}
		RIR_GLOBAL(unsigned long long) *gb = (RIR_GLOBAL(unsigned long long) *)as_global(sp.budget);
		unsigned long long *lb = reinterpret_cast<unsigned long long *>(&bl);
		if (threadIdx.x < sizeof(LossyBudget) / 8)
			lb[threadIdx.x] = gb[threadIdx.x];
		__syncthreads();
		if (threadIdx.x == 0)
		{
			long long st[7];
			st[0] = background;
			for (int i = 1; i < 7; ++i)
				st[i] = red[1][i - 1];
			lossy_budget(sp, st, bl, errors_out);
		}
		__syncthreads();
		if (threadIdx.x < sizeof(LossyBudget) / 8)
			gb[threadIdx.x] = lb[threadIdx.x];
	}

	// One launch per frame of a run (lossy_kernels.h, LossyStep::next_*): L3 + L4 of frame f, and L2 of frame f + 1 taken while
	// frame f's output is still in registers - |t(f+1) - out(f)| is all L2 needs of the state, so the sums never read prevT and
	// prevT is only stored by the last frame of a call.  The decision of frame f is read by every thread before its workgroup
	// takes its ticket; the last workgroup to arrive overwrites it with the decision of frame f + 1, which the next launch reads.
	// Needs s and full to be multiples of 8 (callers fall back to the three-launch step otherwise).  grid.x covers full / 8 threads.
	template <bool TABLE>
	__global__ __launch_bounds__(256) void lossy_frame_kernel(LossyStep one, const LossyStep *__restrict__ table)
	{
		const LossyStep sp = lossy_step_of<TABLE>(one, table);
		const int i8 = blockIdx.x * blockDim.x + threadIdx.x; // group of 8 pixels
		const bool inside = i8 * 8 < sp.full, lossy = i8 * 8 < sp.s;
		const bool next = sp.next_tmp != nullptr;
		U16x8 o8;
		if (sp.do_update)
		{
			if (inside)
			{
				const U16x8 v8 = ld8(as_global(sp.tmp), i8);
				if (lossy)
					o8 = lossy_update8(sp, i8, v8, !next);
				else
				{ // rows past lossy_height: stored as they are
					st8(as_global(sp.out), i8, v8);
					st8(as_global(sp.st.lastDL), i8, v8);
				}
			}
		}
		else if (lossy)
			o8 = ld8(as_global((const uint16_t *)sp.st.prevT), i8);
		if (!next)
			return; // (the whole grid)
		long long a[6] = {0, 0, 0, 0, 0, 0};
		const long long background = *as_global(sp.next_background);
		if (lossy)
		{
			const int subtract_min = sp.st.subtract_min;
			const uint32_t mn = sp.st.min, bg = (uint32_t)background;
			const U16x8 tv = ld8(as_global(sp.next_tmp), i8);
			const U16x8 iv = sp.next_img == sp.next_tmp ? tv : ld8(as_global(sp.next_img), i8);
			int32_t fd = 0, fn = 0, bd = 0; // (8 pixels: |d| < 65 536; the wrapped squares are summed in 64 bits)
			long long f2 = 0, b2 = 0;
#pragma unroll
			for (int k = 0; k < 8; ++k)
			{
				const uint32_t t = subtract_min ? sub_min(tv.get(k), mn) : tv.get(k);
				const int32_t d = abs((int32_t)t - (int32_t)o8.get(k));
				const int32_t d2 = (int32_t)((uint32_t)d * (uint32_t)d);
				if (iv.get(k) > bg)
					fd += d, f2 += d2, fn += 1;
				else
					bd += d, b2 += d2;
			}
			a[0] = fd, a[1] = f2, a[2] = fn, a[3] = bd, a[4] = b2, a[5] = 8 - fn;
		}
		lossy_sums_tail(sp, a, background, sp.next_errors_out);
	}

	// ---- the pixel arithmetic of a run on PAIRS of pixels (packed 16-bit instructions) -----------------------------------
	// The resident kernel is bound by its vector instructions once several streams share the chip (PMC, profiles/r03_pmc_lossy.json:
	// 838 per wave and frame, SIMD utilisation 0.60 with seven streams); most of them unpacked 16-bit lanes, worked on them as
	// 32-bit values and packed them again.  Everything that is 16 bits wide in the reference (pixels, reference and constant
	// values, counters, the per-pixel error bound - errors above 65 535 bound nothing, |t - ref| <= 65 535) is done here two
	// pixels per instruction: saturating subtract for sub_min and for "a > b" (a -sat b != 0), max - min for |a - b|, bit-field
	// inserts under 0 / 0xffff half masks for every select; only the running sums and the squares are 32 bits per pixel.
	// Same values as lossy_pixel / the scalar sums, bit for bit (tests/test_gpu_lossy.py against the oracle).
	typedef unsigned short lossy_u16x2 __attribute__((ext_vector_type(2)));
	__device__ __forceinline__ lossy_u16x2 lp2(uint32_t v) { return __builtin_bit_cast(lossy_u16x2, v); }
	__device__ __forceinline__ uint32_t lu1(lossy_u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
	__device__ __forceinline__ uint32_t lossy_both(uint32_t v) { return (v > 65535u ? 65535u : v) * 0x10001u; } // a 16-bit bound in both halves
	// per half: 0xffff where the half of x is not 0, else 0
	__device__ __forceinline__ uint32_t lossy_nz_mask(lossy_u16x2 x)
	{
#ifndef RIR_LOSSY_NZ_MASK_PLAIN
		// min(x, 1) and 0 - that, two packed instructions.  Spelled as such: written in C (min(x, 1) * 0xffff) the compiler sees through
		// the mask and turns every select under it into two 16-bit compares, two conditional moves and a byte permute.
		uint32_t t, r;
		asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(lu1(x)), "s"(0x00010001u));
		asm("v_pk_sub_u16 %0, 0, %1" : "=v"(r) : "v"(t));
		return r;
#else
		const lossy_u16x2 one = {1, 1}, ffff = {0xffff, 0xffff};
		return lu1(__builtin_elementwise_min(x, one) * ffff);
#endif
	}
	__device__ __forceinline__ uint32_t lossy_bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); } // mask ? a : b, bit by bit
	struct LossyPairConsts
	{
		uint32_t min2, bg2, low2, high2, n2; // both halves: subtract_min bound (0 when off), background, error bounds, n_after
	};
	__device__ __forceinline__ LossyPairConsts lossy_pair_consts(const LossyFrameConsts &c)
	{
		LossyPairConsts p;
		p.min2 = c.subtract_min ? lossy_both(c.min) : 0u;
		p.bg2 = lossy_both(c.background);
		p.low2 = lossy_both((uint32_t)c.low_error), p.high2 = lossy_both((uint32_t)c.high_error); // (both >= 0: lossy_budget*)
		p.n2 = (uint32_t)c.n_after * 0x10001u;
		return p;
	}
	// lossy_pixel for the pixels 2p, 2p + 1 of a group of 8 (word p of the U16x8 values)
	__device__ __forceinline__ void lossy_pixel_pair(const LossyFrameConsts &c, const LossyPairConsts &pc, uint32_t v2, uint32_t old2, uint32_t last2, uint32_t &ref2,
													 uint32_t &sum_lo, uint32_t &sum_hi, uint32_t &cc2, uint32_t &cv2, uint32_t &t_in2, uint32_t &out2)
	{
		const lossy_u16x2 one = {1, 1};
		const lossy_u16x2 v = lp2(v2), ref = lp2(ref2);
		const lossy_u16x2 t = __builtin_elementwise_sub_sat(v, lp2(pc.min2)); // sub_min (min2 = 0 when off)
		t_in2 = lu1(t);
		const uint32_t fgm = lossy_nz_mask(__builtin_elementwise_sub_sat(v, lp2(pc.bg2))); // v > background
		const lossy_u16x2 max_error = lp2(lossy_bfi(fgm, pc.high2, pc.low2));
		const lossy_u16x2 diff = __builtin_elementwise_max(t, ref) - __builtin_elementwise_min(t, ref);
		lossy_u16x2 nk = __builtin_elementwise_sub_sat(diff, max_error); // != 0: the pixel is not kept
		if (!c.add_loss)
			nk = nk | ((lp2(last2) ^ v) >> (lossy_u16x2){13, 13});
		const uint32_t NM = lossy_nz_mask(nk);
		if (c.ra > 0)
		{
			uint32_t sel = 0;
			lossy_u16x2 cc = lp2(cc2);
			if (c.full_ring)
			{ // what leaves the sum: the constant value while its stretch lasts, else the ring's oldest image
				sel = lossy_bfi(lossy_nz_mask(cc), cv2, old2);
				cc = __builtin_elementwise_sub_sat(cc, one);
			}
			const uint32_t t_lo = lu1(t) & 0xffffu, t_hi = lu1(t) >> 16;
			const uint32_t sm_lo = sum_lo + t_lo - (sel & 0xffffu), sm_hi = sum_hi + t_hi - (sel >> 16);
			const uint32_t q = lossy_div(sm_lo, c.div_magic) | (lossy_div(sm_hi, c.div_magic) << 16);
			sum_lo = (NM & 0xffffu) ? __umul24(t_lo, (uint32_t)c.n_after) : sm_lo;
			sum_hi = (NM >> 16) ? __umul24(t_hi, (uint32_t)c.n_after) : sm_hi;
			out2 = lossy_bfi(NM, lu1(t), q);
			cv2 = lossy_bfi(NM, lu1(t), cv2);
			cc2 = lossy_bfi(NM, pc.n2, lu1(cc));
		}
		else
			out2 = lossy_bfi(NM, lu1(t), ref2);
		ref2 = lossy_bfi(NM, lu1(t), ref2);
	}
	// the sums of a frame over the 8 pixels of a thread against the previous output: totals and foreground parts
	//   a[0] = sum d (fg) + (fg pixels << 32), a[1] = sum d2 (fg), a[2] = sum d (all) + (8 << 32), a[3] = sum d2 (all)
	// (the background parts are the differences; d2 = the wrapped 32-bit square, sign-extended, as the reference's int arithmetic has it)
	__device__ __forceinline__ void lossy_sums8_packed(const U16x8 &v8, const U16x8 &o8, uint32_t min2, uint32_t bg2, long long *a)
	{
		const lossy_u16x2 one = {1, 1};
		uint32_t fd = 0, fn = 0, dall = 0;
		long long f2 = 0, t2 = 0;
#pragma unroll
		for (int p = 0; p < 4; ++p)
		{
			const lossy_u16x2 v = lp2(v8.d[p]), o = lp2(o8.d[p]);
			const lossy_u16x2 t = __builtin_elementwise_sub_sat(v, lp2(min2));
			const lossy_u16x2 d = __builtin_elementwise_max(t, o) - __builtin_elementwise_min(t, o);
			const lossy_u16x2 fg = __builtin_elementwise_min(__builtin_elementwise_sub_sat(v, lp2(bg2)), one); // 1 where v > background
			fd = __builtin_amdgcn_udot2(d, fg, fd, false);
			dall = __builtin_amdgcn_udot2(d, one, dall, false);
			fn = __builtin_amdgcn_udot2(fg, one, fn, false);
			const uint32_t d_lo = lu1(d) & 0xffffu, d_hi = lu1(d) >> 16;
			const int32_t q_lo = (int32_t)__umul24(d_lo, d_lo), q_hi = (int32_t)__umul24(d_hi, d_hi); // (d < 2^16: the 24-bit multiplier gives the low 32 bits of d * d)
			const uint32_t m = lu1(fg);
			t2 += (long long)q_lo + (long long)q_hi;
			f2 += (long long)((m & 1u) ? q_lo : 0) + (long long)((m >> 16) ? q_hi : 0);
		}
		a[0] = (long long)(((unsigned long long)fn << 32) | fd);
		a[1] = f2;
		a[2] = (long long)((8ull << 32) | dall);
		a[3] = t2;
	}

	// ---- a run of frames in one launch ---------------------------------------------------------------------------
	//
	// grid = workgroups of a stream (nb = lossy_run_workgroups(full)) x streams, 1-D, all resident at once (lossy_run_capacity(),
	// lossy_kernels.h); a ticket deals (stream, workgroup) in the order workgroups start.  A workgroup only ever waits for
	// workgroups of ITS stream.
	// Thread (b, tid) owns pixels [8 i8, 8 i8 + 8), i8 = 1024 b + tid, for every frame of the run: refT, lastDL, the running sums and
	// the constant-stretch counters of its pixels live in registers from the first frame to the last; per frame it reads the
	// pixels and the ring's oldest image and writes the output and the ring's newest image (8 bytes per pixel and frame instead of 30).
	// Frame k of the run, per workgroup:
	//   1. sums of frame k against the previous output (in registers), reduced over the workgroup; wave 0 publishes them as four
	//      words TAG(k) << 48 | value(s) to exchange[k & 1][b] (agent-scope stores; a word is its own flag)
	//   2. wave 0 of every workgroup polls the words of all nb workgroups of its stream (agent-scope loads, lane l takes workgroups
	//      l, l + 64, ...), adds them up and lane 0 computes the budget - every workgroup the same one, from the same integers
	//   3. update of frame k with that decision.
	// A workgroup publishes frame k + 1 only after it has read all of frame k, and nobody gets to publish frame k + 2 before everybody
	// has published k + 1: two buffers are enough.  Waits are bounded by a clock (2 s): a wait that gives up raises error_word and
	// the run goes on with whatever it has - wrong, flagged, but never hung.
	constexpr int kRunWaves = kLossyRunThreads / 64;
	// PARKED: the running sums, the constant-stretch counters / values, the reference levels and the previous frame's pixels of a thread's
	// pixels live in LDS between the frames (96 B per thread; a thread only ever touches its own slots: no barrier; [pair][thread]: 8- and
	// 4-byte accesses at that lane stride) and come into registers a pixel pair at a time - 24 VGPRs less at the peak, which is what lets
	// the kernel be built for 6 waves per SIMD (80 VGPRs, nothing spilled; 25.6 KB of LDS: six workgroups to a CU) instead of 5: NINE
	// streams of 640x512 per launch instead of seven.  Seven streams run about as fast either way (568 against 580 k frames/s), nine at
	// once are faster than seven (636 k): the host takes this form when it saves a launch (lossy_run_parked_kernel; same arithmetic,
	// same results; scripts/lossy_forms.sh).
	template <bool PARKED>
	__device__ __forceinline__ void lossy_run_body(const LossyRun *__restrict__ table, unsigned int *__restrict__ ticket_, int nb, int nstreams, unsigned int epoch, unsigned int arrivals_before,
												   const unsigned int *__restrict__ ok_word)
	{
		__shared__ unsigned int sh_ticket;
		__shared__ unsigned int sh_flag;
		__shared__ long long red[kRunWaves][6], red2[kRunWaves][6];
		__shared__ LossyBudget bl;
		__shared__ LossyDecision dec;
		constexpr int kParkedThreads = PARKED ? kLossyRunThreads : 1;
		__shared__ uint2 st_sum[4][kParkedThreads];
		__shared__ uint32_t st_cc[4][kParkedThreads], st_cv[4][kParkedThreads];
		__shared__ uint32_t st_ref[4][kParkedThreads], st_last[4][kParkedThreads];
		const int tid = threadIdx.x;
		if (tid == 0)
		{
			RIR_GLOBAL(unsigned int) *ticket = as_global(ticket_);
			const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (t == (unsigned int)(nb * nstreams) - 1u)
				__hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
			sh_ticket = t;
		}
		__syncthreads();
		const int tk = __builtin_amdgcn_readfirstlane((int)sh_ticket);
		// The group has been stepped by the constant-budget form (lossy_const_run_kernel; the word was written by an earlier launch of the
		// same stream): every workgroup ARRIVES - the control block's count stays what the host expects - and leaves without waiting for
		// anybody: a launch that has nothing to do cannot fail to be resident.
		if (ok_word && *as_global(ok_word) != 0u)
		{
			if (tid == 0)
				__hip_atomic_fetch_add(as_global(ticket_ + kLossyRunCtlWord), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		// is the whole launch on the chip (resident_device.h)?  If not - or if an earlier group of the same call was not done - everybody
		// leaves before the state is touched; the control words live in the header of the exchange buffer, behind the ticket and the error word
		if (resident_rendezvous(ticket_ + kLossyRunCtlWord, arrivals_before, gridDim.x, epoch, &sh_flag, true) != RESIDENT_GO)
			return;
		const int stream = tk / nb, b = tk - stream * nb;
		LossyRun rp;
		{
			static_assert(sizeof(LossyRun) % 8 == 0, "LossyRun is copied in 8-byte words");
			RIR_GLOBAL(const unsigned long long) *src = (RIR_GLOBAL(const unsigned long long) *)(table + stream);
			unsigned long long *dst = reinterpret_cast<unsigned long long *>(&rp);
#pragma unroll
			for (size_t k = 0; k < sizeof(LossyRun) / 8; ++k)
				dst[k] = src[k];
		}
		const LossyDeviceState st = rp.st;
		RIR_GLOBAL(uint16_t) *refT = as_global(st.refT), *prevT = as_global(st.prevT), *lastDL = as_global(st.lastDL);
		RIR_GLOBAL(uint16_t) *cval = as_global(st.ra_const_value), *ring = as_global(st.ra_images);
		RIR_GLOBAL(uint16_t) *ccnt = (RIR_GLOBAL(uint16_t) *)as_global(st.ra_const_count);
		RIR_GLOBAL(uint32_t) *sums = as_global(st.ra_sums);
		RIR_GLOBAL(const uint16_t) *in = as_global(rp.in);
		RIR_GLOBAL(uint16_t) *out = as_global(rp.out);
		RIR_GLOBAL(unsigned long long) *exch = as_global(rp.exchange);
		const int s = rp.s, full = rp.full, ra = st.running_average, add_loss = rp.add_loss;
		const int i8 = b * kLossyRunThreads + tid;
		const bool inside = i8 * 8 < full, lossy = i8 * 8 < s;
		const LossyBudgetParams bp = {s, add_loss, rp.low_value_error, rp.high_value_error, rp.std_factor};

		// the budget state of the stream: every workgroup keeps its own copy in LDS and advances it identically
		{
			RIR_GLOBAL(const unsigned long long) *gb = (RIR_GLOBAL(const unsigned long long) *)as_global(rp.budget);
			unsigned long long *lb = reinterpret_cast<unsigned long long *>(&bl);
			if (tid < (int)(sizeof(LossyBudget) / 8))
				lb[tid] = gb[tid];
		}
		// state of this thread's pixels -> registers
		U16x8 ref8{}, last8{}, cc8{}, cv8{}, o8{}, old8{}, t8{};
		uint32_t sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		int count = st.ra_count, head = st.ra_head;
		if (lossy)
		{
			ref8 = ld8(refT, i8);
			last8 = ld8(lastDL, i8);
			o8 = ld8(prevT, i8);
			if (ra > 0)
			{
				const lossy_v4u s0 = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(sums + (size_t)i8 * 8), s1 = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(sums + (size_t)i8 * 8 + 4);
				sum[0] = s0.x, sum[1] = s0.y, sum[2] = s0.z, sum[3] = s0.w, sum[4] = s1.x, sum[5] = s1.y, sum[6] = s1.z, sum[7] = s1.w;
				cc8 = ld8(ccnt, i8);
				cv8 = ld8(cval, i8);
				if (count == ra)
					old8 = ld8(ring + (size_t)head * s, i8);
			}
		}
		if constexpr (PARKED)
		{
#pragma unroll
			for (int p = 0; p < 4; ++p)
			{
				st_sum[p][tid] = make_uint2(sum[2 * p], sum[2 * p + 1]), st_cc[p][tid] = cc8.d[p], st_cv[p][tid] = cv8.d[p];
				st_ref[p][tid] = ref8.d[p], st_last[p][tid] = last8.d[p];
			}
		}
		U16x8 v8{};
		if (inside)
			v8 = ld8(in, i8);
		const unsigned long long t_limit = 200000000ull; // 2 s of the 100 MHz clock
		bool gave_up = false;
		for (int k = 0; k < rp.nsteps; ++k)
		{
#ifdef RIR_LOSSY_DIAG
			const unsigned long long dgA = __builtin_amdgcn_s_memrealtime();
#endif
			// the next frame's pixels are requested now: they arrive while this frame's sums go round
			U16x8 vn{};
			if (inside && k + 1 < rp.nsteps)
				vn = ld8(in + (size_t)(k + 1) * rp.frame_px, i8);
			const long long background = as_global(rp.bg)[(size_t)k * rp.bg_stride];
			// 1. this workgroup's share of the frame's sums (the packed form of the sums - lossy_sums8_packed, four reductions instead of
			// six - is kept behind RIR_LOSSY_PACKED_SUMS: same values, but its extra live registers cost more than its instructions save:
			// 7 streams 509 k frames/s against 593 k; scripts/lossy_ab.sh)
#ifndef RIR_LOSSY_PACKED_SUMS
			int32_t fd = 0, fn = 0, bd = 0, bn = 0;
			long long f2 = 0, b2 = 0;
			if (lossy)
			{
				const uint32_t bg = (uint32_t)background;
#pragma unroll
				for (int q = 0; q < 8; ++q)
				{
					const uint32_t t = st.subtract_min ? sub_min(v8.get(q), st.min) : v8.get(q);
					const int32_t d = abs((int32_t)t - (int32_t)o8.get(q));
					const int32_t d2 = (int32_t)((uint32_t)d * (uint32_t)d);
					if (v8.get(q) > bg)
						fd += d, f2 += d2, fn += 1;
					else
						bd += d, b2 += d2, bn += 1;
				}
			}
#else
			long long ps[4] = {0, 0, 0, 0};
			if (lossy)
				lossy_sums8_packed(v8, o8, st.subtract_min ? lossy_both(st.min) : 0u, lossy_both((uint32_t)background), ps);
#endif
#ifdef RIR_LOSSY_DIAG
			const unsigned long long dg0 = __builtin_amdgcn_s_memrealtime();
			unsigned long long dg1 = 0, dg2 = 0, dg3 = 0, dgB = 0, dgC = 0;
#endif
			// wave sums -> LDS; wave 0 adds the waves up, publishes, collects everybody's words and decides
#ifndef RIR_LOSSY_PACKED_SUMS
			// (The per-pixel sums above as totals + foreground parts without the branch - |t - o| as one v_sad_u16, selects instead of the two
			// sides under exec masks - were measured: 7 streams 589 k frames/s against 635 k; two 64-bit adds per pixel cost more than the exec
			// switches.)
			// (a wave holds 512 pixels: its sums of d stay below 2^25 and its counts below 2^10 - three 32-bit reductions, the two counts in
			// one word, and two 64-bit ones for the squares, instead of six 64-bit ones: 72 vector instructions less per frame)
			long long ws[6];
			{
				const uint32_t wfd = lossy_wave_sum32((uint32_t)fd), wbd = lossy_wave_sum32((uint32_t)bd), wn = lossy_wave_sum32((uint32_t)fn | ((uint32_t)bn << 16));
				ws[0] = (long long)wfd, ws[1] = lossy_wave_sum(f2), ws[2] = (long long)(wn & 0xffffu), ws[3] = (long long)wbd, ws[4] = lossy_wave_sum(b2),
				ws[5] = (long long)(wn >> 16);
			}
#else
			// four reductions instead of six: counts ride in the high halves of the sums of d (a wave's sum of d stays below 2^26),
			// the background parts are totals minus foreground
			long long ws[6];
			{
#pragma unroll
				for (int j = 0; j < 4; ++j)
					ps[j] = lossy_wave_sum(ps[j]);
				const long long wfd = ps[0] & 0xffffffffll, wfn = ps[0] >> 32, wd = ps[2] & 0xffffffffll, wn = ps[2] >> 32;
				ws[0] = wfd, ws[1] = ps[1], ws[2] = wfn, ws[3] = wd - wfd, ws[4] = ps[3] - ps[1], ws[5] = wn - wfn;
			}
#endif
			const int lane = tid & 63, wave = tid >> 6;
			if (lane < 6)
				red[wave][lane] = lane == 0 ? ws[0] : lane == 1 ? ws[1] : lane == 2 ? ws[2] : lane == 3 ? ws[3] : lane == 4 ? ws[4] : ws[5];
			__syncthreads();
#ifdef RIR_LOSSY_DIAG
			dgB = __builtin_amdgcn_s_memrealtime();
#endif
			const unsigned long long tag = (unsigned long long)(((unsigned)k & 0x7fffu) | 0x8000u) << 48;
			const unsigned long long vmask = 0x0000ffffffffffffull;
			RIR_GLOBAL(unsigned long long) *bank = exch + (size_t)(k & 1) * nb * rp.slot_words;
			double part = 0;
			if (wave == 0)
			{
				long long val = 0;
				if (lane < 6)
#pragma unroll
					for (int w = 0; w < kRunWaves; ++w)
						val += red[w][lane];
#pragma unroll
				for (int j = 0; j < 6; ++j)
					ws[j] = __shfl(val, j, 64);
				// four words: TAG | fg count | fg sum d,  TAG | fg sum d2,  TAG | bg count | bg sum d,  TAG | bg sum d2
				// (a workgroup has 2 048 pixels: counts < 2^12, sums of d < 2^28, |sums of d2| < 2^42)
				if (lane < 4)
				{
					const unsigned long long w = lane == 0	 ? ((unsigned long long)ws[2] << 30) | (unsigned long long)ws[0]
												 : lane == 1 ? (unsigned long long)ws[1] & vmask
												 : lane == 2 ? ((unsigned long long)ws[5] << 30) | (unsigned long long)ws[3]
															 : (unsigned long long)ws[4] & vmask;
					__hip_atomic_store(bank + (size_t)b * rp.slot_words + lane, tag | w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
#ifdef RIR_LOSSY_DIAG
				dgC = __builtin_amdgcn_s_memrealtime();
#endif
				// while the words travel: the part of the budget that does not need them
				if (lane < 2 && (!rp.leader || b == 0))
					part = lossy_budget2_prepare(bl, lane);
			}
#ifdef RIR_LOSSY_DIAG
			dg1 = __builtin_amdgcn_s_memrealtime();
#endif
			// 2. everybody's shares: thread p takes workgroup p (p + 256, ...).  With many streams in the launch only workgroup 0 of a
			// stream does that (every workgroup reading every slot is nb^2 polled lines per stream and frame: with 6 streams the polls
			// took 5.8 us instead of 2) and hands the decision on in one word; the others poll that word - one more hop, 1/nb of the traffic.
			const bool collect = !rp.leader || b == 0;
			long long acc[6] = {0, 0, 0, 0, 0, 0};
			if (rp.leader && b == 0)
				__builtin_amdgcn_s_setprio(3); // (the whole stream waits for what this workgroup does next, and it shares its CU with four others: 7 streams 561 -> 579 k frames/s)
			// (wave 0 of a collecting workgroup is in the window sum for another microsecond - RIR_LOSSY_DIAG: "window sum" - so the other three
			// waves take the slots between them when they can: 192 threads for the 160 workgroups of a 640x512 stream.  The round trip of the
			// polls then runs under the window sum instead of behind it.)
			const bool three_waves = nb <= kLossyRunThreads - 64;
			if (collect)
				for (int p = three_waves ? tid - 64 : tid; p >= 0 && p < nb; p += three_waves ? kLossyRunThreads - 64 : kLossyRunThreads)
				{
					unsigned long long w[4];
					const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
					for (;;)
					{
						bool ok = true;
#pragma unroll
						for (int j = 0; j < 4; ++j)
						{
							w[j] = __hip_atomic_load(bank + (size_t)p * rp.slot_words + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							ok = ok && ((w[j] & ~vmask) == tag);
						}
						if (ok || gave_up)
							break;
						__builtin_amdgcn_s_sleep(1);
						if (__builtin_amdgcn_s_memrealtime() - t_start > t_limit)
						{
							gave_up = true;
							__hip_atomic_store(as_global(rp.error_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
					}
					acc[0] += (long long)(w[0] & 0x3fffffffull), acc[2] += (long long)((w[0] & vmask) >> 30);
					acc[1] += (long long)(w[1] << 16) >> 16; // 48-bit two's complement
					acc[3] += (long long)(w[2] & 0x3fffffffull), acc[5] += (long long)((w[2] & vmask) >> 30);
					acc[4] += (long long)(w[3] << 16) >> 16;
				}
			// the decision word of the frame: TAG | low error (24 bits) | high error (24 bits), after the 2 x nb slots of the stream
			RIR_GLOBAL(unsigned long long) *dword = exch + (size_t)2 * nb * rp.slot_words + (size_t)(k & 1) * 8;
			if (collect)
			{
				// (the counts of a stream stay below 2^32: 32-bit reductions for them)
				acc[0] = lossy_wave_sum(acc[0]), acc[1] = lossy_wave_sum(acc[1]), acc[3] = lossy_wave_sum(acc[3]), acc[4] = lossy_wave_sum(acc[4]);
				acc[2] = (long long)lossy_wave_sum32((uint32_t)acc[2]), acc[5] = (long long)lossy_wave_sum32((uint32_t)acc[5]);
				if (lane < 6)
					red2[wave][lane] = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : lane == 3 ? acc[3] : lane == 4 ? acc[4] : acc[5];
				__syncthreads();
#ifdef RIR_LOSSY_DIAG
				dg2 = __builtin_amdgcn_s_memrealtime();
#endif
				if (wave == 0)
				{
					long long val = 0;
					if (lane < 6)
#pragma unroll
						for (int w = 0; w < kRunWaves; ++w)
							val += red2[w][lane];
					long long stt[7];
					stt[0] = background;
#pragma unroll
					for (int j = 0; j < 6; ++j)
						stt[1 + j] = __shfl(val, j, 64);
					if (lane < 2)
					{
						const LossyDecision d = lossy_budget2_finish(bp, stt, bl, lane, part);
						if (lane == 0)
						{
							dec = d;
							if (rp.leader)
								__hip_atomic_store(dword, tag | ((unsigned long long)((unsigned)d.low_error & 0xffffffu) << 24) | ((unsigned)d.high_error & 0xffffffu),
												   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							if (b == 0 && rp.errors_out)
							{
								as_global(rp.errors_out)[2 * k] = d.low_error;
								as_global(rp.errors_out)[2 * k + 1] = d.high_error;
							}
						}
					}
				}
			}
			else if (tid == 0)
			{
				unsigned long long w;
				const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
				for (;;)
				{
					w = __hip_atomic_load(dword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((w & ~vmask) == tag || gave_up)
						break;
					__builtin_amdgcn_s_sleep(1);
					if (__builtin_amdgcn_s_memrealtime() - t_start > t_limit)
					{
						gave_up = true;
						__hip_atomic_store(as_global(rp.error_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
				}
				LossyDecision d;
				d.background = (uint32_t)background;
				d.low_error = (int)((w >> 24) & 0xffffffu), d.high_error = (int)(w & 0xffffffu);
				d.reserved = 0;
				dec = d;
			}
#ifdef RIR_LOSSY_DIAG
			dg3 = __builtin_amdgcn_s_memrealtime();
#endif
			if (rp.leader && b == 0)
				__builtin_amdgcn_s_setprio(0);
			__syncthreads(); // (also keeps the other waves off red / red2 until wave 0 has read them)
			// 3. update
			const bool full_ring = ra > 0 && count == ra;
			const int n_after = ra > 0 ? (full_ring ? ra : count + 1) : 0;
			if (lossy)
			{
				const LossyFrameConsts fc = {st.min, dec.background, st.subtract_min, ra, full_ring ? 1 : 0, n_after, add_loss, dec.low_error, dec.high_error,
											 lossy_div_magic(n_after)};
#ifdef RIR_LOSSY_SCALAR_UPDATE /* (round 2's form, one pixel at a time: 7 streams 570 k frames/s against 593 k with pairs) */
				bool rc = false, ccg = false;
#pragma unroll
				for (int q = 0; q < 8; ++q)
				{
					uint32_t ref = ref8.get(q), cc = cc8.get(q), cv = cv8.get(q), t_in;
					const uint32_t t = lossy_pixel(fc, v8.get(q), old8.get(q), last8.get(q), ref, sum[q], cc, cv, &t_in, rc, ccg);
					ref8.set(q, ref), cc8.set(q, cc), cv8.set(q, cv);
					t8.set(q, t_in);
					o8.set(q, t);
				}
#else
				const LossyPairConsts pc = lossy_pair_consts(fc);
				if constexpr (PARKED)
				{
#pragma unroll
					for (int p = 0; p < 4; ++p)
					{ // (a pair's sums and counters come from LDS right before its update and go back right after: four registers live, not sixteen)
						uint2 s2 = st_sum[p][tid];
						uint32_t c2 = st_cc[p][tid], v2 = st_cv[p][tid];
						uint32_t r2 = st_ref[p][tid];
						const uint32_t l2 = st_last[p][tid];
						lossy_pixel_pair(fc, pc, v8.d[p], old8.d[p], l2, r2, s2.x, s2.y, c2, v2, t8.d[p], o8.d[p]);
						st_ref[p][tid] = r2, st_last[p][tid] = v8.d[p]; // (this frame's pixels are the next frame's "last")
						st_sum[p][tid] = s2, st_cc[p][tid] = c2, st_cv[p][tid] = v2;
					}
				}
				else
				{
#pragma unroll
					for (int p = 0; p < 4; ++p)
						lossy_pixel_pair(fc, pc, v8.d[p], old8.d[p], last8.d[p], ref8.d[p], sum[2 * p], sum[2 * p + 1], cc8.d[p], cv8.d[p], t8.d[p], o8.d[p]);
				}
#endif
				if constexpr (!PARKED)
					last8 = v8;
				st8(out + (size_t)k * rp.frame_px, i8, o8);
				if (ra > 0)
				{
					const int slot = full_ring ? head : (head + count) % ra;
					st8(ring + (size_t)slot * s, i8, t8);
				}
			}
			else if (inside)
				st8(out + (size_t)k * rp.frame_px, i8, v8); // rows past lossy_height: stored as they are
			if (ra > 0)
			{
				if (count == ra)
					head = (head + 1) % ra;
				else
					++count;
				// the oldest image for the next frame: written at least one frame ago (by this thread) unless the ring holds one image
				if (lossy && count == ra && k + 1 < rp.nsteps)
					old8 = ra == 1 ? t8 : ld8(ring + (size_t)head * s, i8);
			}
			if (k + 1 < rp.nsteps)
				v8 = vn;
#ifdef RIR_LOSSY_DIAG
			if (b == 0 && tid == 0)
			{ // ticks (10 ns) of: sums + reduce + publish | poll | reduce | budget | barrier + update, summed over the frames
				RIR_GLOBAL(unsigned long long) *dg = (RIR_GLOBAL(unsigned long long) *)as_global(rp.error_word) + 8;
				const unsigned long long dg4 = __builtin_amdgcn_s_memrealtime();
				dg[0] += dg1 - dg0, dg[1] += dg2 - dg1, dg[2] += dg3 - dg2, dg[3] += dg4 - dg3, dg[4] += 1, dg[5] += dg0 - dgA, dg[6] += dgB - dg0, dg[7] += dgC - dgB;
			}
#endif
		}
		// registers -> state
		if constexpr (PARKED)
		{
#pragma unroll
			for (int p = 0; p < 4; ++p)
			{
				const uint2 s2 = st_sum[p][tid];
				sum[2 * p] = s2.x, sum[2 * p + 1] = s2.y, cc8.d[p] = st_cc[p][tid], cv8.d[p] = st_cv[p][tid];
				ref8.d[p] = st_ref[p][tid], last8.d[p] = st_last[p][tid];
			}
		}
		if (lossy)
		{
			st8(refT, i8, ref8);
			st8(lastDL, i8, last8);
			st8(prevT, i8, o8);
			if (ra > 0)
			{
				lossy_v4u s0, s1;
				s0.x = sum[0], s0.y = sum[1], s0.z = sum[2], s0.w = sum[3], s1.x = sum[4], s1.y = sum[5], s1.z = sum[6], s1.w = sum[7];
				*reinterpret_cast<RIR_GLOBAL(lossy_v4u) *>(sums + (size_t)i8 * 8) = s0;
				*reinterpret_cast<RIR_GLOBAL(lossy_v4u) *>(sums + (size_t)i8 * 8 + 4) = s1;
				st8(ccnt, i8, cc8);
				st8(cval, i8, cv8);
			}
		}
		else if (inside)
			st8(lastDL, i8, v8);
		if (b == 0)
		{
			__syncthreads();
			RIR_GLOBAL(unsigned long long) *gb = (RIR_GLOBAL(unsigned long long) *)as_global(rp.budget);
			const unsigned long long *lb = reinterpret_cast<const unsigned long long *>(&bl);
			if (tid < (int)(sizeof(LossyBudget) / 8))
				gb[tid] = lb[tid];
			if (tid == 0)
			{
				RIR_GLOBAL(LossyDecision) *gd = as_global(rp.decision);
				gd->background = dec.background, gd->low_error = dec.low_error, gd->high_error = dec.high_error;
			}
		}
	}

	__attribute__((amdgpu_waves_per_eu(kLossyRunWavesPerSimd, kLossyRunWavesPerSimd))) __global__ __launch_bounds__(kLossyRunThreads) void lossy_run_kernel(
		const LossyRun *__restrict__ table, unsigned int *__restrict__ ticket_, int nb, int nstreams, unsigned int epoch, unsigned int arrivals_before,
		const unsigned int *__restrict__ ok_word)
	{
		lossy_run_body<false>(table, ticket_, nb, nstreams, epoch, arrivals_before, ok_word);
	}
	__attribute__((amdgpu_waves_per_eu(kLossyRunParkedWavesPerSimd, kLossyRunParkedWavesPerSimd))) __global__ __launch_bounds__(kLossyRunThreads) void lossy_run_parked_kernel(
		const LossyRun *__restrict__ table, unsigned int *__restrict__ ticket_, int nb, int nstreams, unsigned int epoch, unsigned int arrivals_before,
		const unsigned int *__restrict__ ok_word)
	{
		lossy_run_body<true>(table, ticket_, nb, nstreams, epoch, arrivals_before, ok_word);
	}

	// ---- the constant-budget form of a run ---------------------------------------------------------------------------------
	//
	// With stdFactor == 0 - the configuration BASELINE names for configs[4] (reference tests/python/test_video_io.py:112-116) - the
	// budget arithmetic (h264.cpp:2370-2376) multiplies the frame's statistic by zero: lowError / highError are the configured values for
	// every frame, UNLESS a NaN is in play (an empty foreground or background makes the statistic 0 / 0, and NaN x 0 is NaN: the frame
	// and the 39 after it become lossless - lossy_budget above).  What is left of the frame loop (h264.cpp:2387-2413: running mean,
	// decision, refT, lastDL) is per pixel in time: nothing a frame needs comes from another workgroup, so the two cross-workgroup
	// hand-offs per frame that the resident kernel spends most of its time on (profiles/r03_pmc_lossy.json: 69 % of the wave cycles
	// waiting, 0.05 of the HBM peak for one stream) buy nothing here.
	//
	// lossy_const_run_kernel: grid = (workgroups of a stream, streams), an ordinary launch - nobody waits for anybody, any grid fits.
	// A thread owns 8 pixels for every frame of the group and keeps their state in registers, as in lossy_run_kernel; frames come in
	// through a ring of 4 slots (three frames requested ahead).  Per pixel and frame: the pixel in, the pixel out, and the frame that
	// leaves the running average - which, from the ring's length on, is an INPUT frame of this very group (the ring holds the last `ra`
	// inputs less the minimum), re-read where it lies, mostly from the Infinity Cache; the ring itself is only written by the last `ra`
	// frames of the group (what it must hold afterwards) and only read by the first `ra`.  The statistic of a frame is still history
	// (a later set_parameter("stdFactor", 5) must find the window as the reference would have it), but only the last 40 frames of a group
	// are in the window afterwards: they - and the stream's very first budget frame, which seeds firstStdDevs - leave their sums, per
	// workgroup, in `partials`; lossy_const_finish_kernel turns those into the window entries in exact double arithmetic, as
	// lossy_budget does, and writes the budgets of the group's frames.
	// Both kernels first make sure the precondition holds for EVERY stream of the launch - backgrounds without the "not sure" bit, no NaN
	// in any window, no poison from an earlier group - and otherwise return before touching anything; the resident kernel that the host
	// queues behind them then does the group (lossy_run_body's ok_word).  Same inputs, same decision in every workgroup.
	__device__ __forceinline__ bool lossy_const_precondition(const LossyRun *__restrict__ table, int nstreams, const unsigned int *__restrict__ poison, unsigned int *sh)
	{
		const int tid = threadIdx.x, nt = blockDim.x;
		if (tid == 0)
			*sh = 0;
		__syncthreads();
		bool bad = poison && __hip_atomic_load(as_global(poison), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
		for (int q = 0; q < nstreams && !bad; ++q)
		{
			RIR_GLOBAL(const LossyRun) *r = as_global(table + q);
			RIR_GLOBAL(const long long) *bg = as_global(r->bg);
			const int n = r->nsteps, stride = r->bg_stride;
			for (int k = tid; k < n; k += nt)
				bad = bad || ((bg[(size_t)k * stride] >> 40) & 1) != 0;
			RIR_GLOBAL(const LossyBudget) *b = as_global(r->budget);
			const int n_first = b->n_first, n_win = b->n_win;
			if (tid < 2 && n_first > 0)
				bad = bad || b->first_std[tid] != b->first_std[tid];
			for (int j = tid; j < 2 * n_win && j < 80; j += nt)
				bad = bad || b->win[j >> 1][j & 1] != b->win[j >> 1][j & 1];
		}
		if (bad)
			atomicOr(sh, 1u);
		__syncthreads();
		const bool ok = *sh == 0;
		__syncthreads();
		return ok;
	}
	// slot of frame k of a group of n frames in `partials`, -1: the frame leaves no sums
	__device__ __forceinline__ int lossy_const_slot(int k, int n)
	{
		const int tail0 = n > kLossyConstTail ? n - kLossyConstTail : 0;
		return k >= tail0 ? 1 + (k - tail0) : (k == 0 ? 0 : -1);
	}

#ifndef RIR_LOSSY_CONST_DEPTH
#define RIR_LOSSY_CONST_DEPTH 4
#endif
#ifndef RIR_LOSSY_CONST_PIN_STEPS
#define RIR_LOSSY_CONST_PIN_STEPS 1
#endif
	constexpr int kConstDepth = RIR_LOSSY_CONST_DEPTH; // slots of a thread's ring of frames: 3 in flight in the middle of a group (4: 1.49 M frames/s one stream, 6: 1.53, 8: 1.50 - vector issue, not what is in flight, bounds the kernel)
	// NP pairs of pixels per thread (4: one 16-byte access per thread, frame and array, as the resident kernel; 2; 1).  Nobody waits for anybody
	// here, so a stream may be cut as finely as pays: with 8 pixels per thread a 640x512 stream is 640 waves on the chip's 1 024 SIMDs - each
	// working through its ~300 vector instructions per frame alone - with 2 pixels it is 2 560 waves that hide each other's latencies.
	typedef unsigned int lossy_v2u __attribute__((ext_vector_type(2)));
	template <int NP>
	struct PxN
	{
		uint32_t d[NP];
		__device__ __forceinline__ uint32_t get(int k) const { return (k & 1) ? d[k >> 1] >> 16 : d[k >> 1] & 0xffffu; }
	};
	template <int NP, class P>
	__device__ __forceinline__ PxN<NP> ldn(P p, size_t i)
	{
		PxN<NP> r;
		if constexpr (NP == 4)
		{
			const lossy_v4u v = *reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(p + i * 8);
			r.d[0] = v.x, r.d[1] = v.y, r.d[2] = v.z, r.d[3] = v.w;
		}
		else if constexpr (NP == 2)
		{
			const lossy_v2u v = *reinterpret_cast<RIR_GLOBAL(const lossy_v2u) *>(p + i * 4);
			r.d[0] = v.x, r.d[1] = v.y;
		}
		else
			r.d[0] = *reinterpret_cast<RIR_GLOBAL(const uint32_t) *>(p + i * 2);
		return r;
	}
	template <int NP, class P>
	__device__ __forceinline__ void stn(P p, size_t i, const PxN<NP> &r)
	{
		if constexpr (NP == 4)
		{
			lossy_v4u v;
			v.x = r.d[0], v.y = r.d[1], v.z = r.d[2], v.w = r.d[3];
			*reinterpret_cast<RIR_GLOBAL(lossy_v4u) *>(p + i * 8) = v;
		}
		else if constexpr (NP == 2)
		{
			lossy_v2u v;
			v.x = r.d[0], v.y = r.d[1];
			*reinterpret_cast<RIR_GLOBAL(lossy_v2u) *>(p + i * 4) = v;
		}
		else
			*reinterpret_cast<RIR_GLOBAL(uint32_t) *>(p + i * 2) = r.d[0];
	}

	// Every vector-memory operation of the frame loop is an UNCONDITIONAL raw-buffer access - a lane without a pixel, a frame past the end
	// of the group, a ring that is not read or written this frame: an out-of-range offset or an empty descriptor, which the hardware turns
	// into "no access" - in straight-line code: only then does the compiler keep counted waits (s_waitcnt vmcnt(N)) and the loads of the
	// next three frames stay in flight while this one is worked on.  (The first version of this kernel had its loads and stores under
	// `if (inside)` / `if (lossy)`: every frame began with s_waitcnt vmcnt(0) - a full memory latency per frame and wave, 1.4 us.)
	typedef unsigned int lossy_v2u_b __attribute__((ext_vector_type(2)));
#define RIR_LOSSY_BUF_FLAGS 0x00020000 /* raw buffer, 32-bit data format (gfx942 / gfx950 descriptor word 3) */
#define RIR_LOSSY_OOB 0x80000000u		/* beyond any num_records used here (a frame is < 2 GiB) */
	__device__ __forceinline__ __amdgpu_buffer_rsrc_t lossy_rsrc(const void *base, uint32_t bytes)
	{ // (base and bytes are wave-uniform by construction: scalar loads of the launch's table and scalar arithmetic on them)
		return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, RIR_LOSSY_BUF_FLAGS);
	}
	template <int NP>
	__device__ __forceinline__ PxN<NP> buf_ldn(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t soff = 0u)
	{ // soff: a wave-uniform offset (a scalar register of the instruction)
		PxN<NP> x;
		if constexpr (NP == 4)
		{
			const lossy_v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0);
			x.d[0] = v.x, x.d[1] = v.y, x.d[2] = v.z, x.d[3] = v.w;
		}
		else if constexpr (NP == 2)
		{
			const lossy_v2u_b v = __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0);
			x.d[0] = v.x, x.d[1] = v.y;
		}
		else
			x.d[0] = __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0);
		return x;
	}
	template <int NP, int AUX = 0>
	__device__ __forceinline__ void buf_stn(const PxN<NP> &x, __amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t soff = 0u)
	{
		if constexpr (NP == 4)
		{
			lossy_v4u v;
			v.x = x.d[0], v.y = x.d[1], v.z = x.d[2], v.w = x.d[3];
			__builtin_amdgcn_raw_buffer_store_b128(v, r, off, soff, AUX);
			// A 16-byte buffer store followed AT ONCE by a vector instruction that writes one of its data registers stored that instruction's
			// result instead - on gfx950, with the scalar offset in a register (`buffer_store_dwordx4 v[44:47], v127, s[20:23], s40 offen nt` then
			// `v_and_or_b32 v44, ...`: two pixels of a thread's eight, in a few waves of a frame, found when the speculative instance began to
			// compute its byte plane right behind the store of the output frame; eight idle cycles between the two, or another order, and it was
			// gone).  The ISA's table asks for wait states behind stores of more than 8 bytes but excepts those with a register offset, and so does
			// the compiler's hazard pass (GCNHazardRecognizer::createsVALUHazard): nothing was inserted.  scripts/ubench/store_hazard.hip issues the
			// pair from inline assembly on a full chip: 4-7 records in 10 000 arrive overwritten with the offset in a register (nt or not), none with
			// one idle cycle in between, none for 8-byte stores (profiles/r06_store_hazard.txt).  The data registers are kept alive until four idle
			// cycles behind the store - whatever the compiler schedules in between cannot write them.
			asm volatile("s_nop 3" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory");
		}
		else if constexpr (NP == 2)
		{
			lossy_v2u_b v;
			v.x = x.d[0], v.y = x.d[1];
			__builtin_amdgcn_raw_buffer_store_b64(v, r, off, soff, AUX);
		}
		else
			__builtin_amdgcn_raw_buffer_store_b32(x.d[0], r, off, soff, AUX);
	}
#ifndef RIR_SPEC_OUT_STORE_AUX
#define RIR_SPEC_OUT_STORE_AUX 2 /* nt: the speculative instance's output frames are read once more, by the sums kernel, from HBM either way (+3 % on the call) */
#endif
#ifndef RIR_CONST_OUT_STORE_AUX
#define RIR_CONST_OUT_STORE_AUX 0 /* cache policy of the streaming kernel's output frames (2: nt) */
#endif

	// lossy_pixel_pair without its wave-uniform branches (compile-time: a running average or none, the addLoss variant or not; run time, as
	// masks: whether the ring is full).  Same values, bit for bit.
	struct ConstPairConsts
	{
		uint32_t min2, bg2, low2, high2, n2; // both halves: subtract_min bound (0 when off), background, error bounds, n_after
		uint32_t n_after, magic;			 // images in the running average after this frame, lossy_div_magic of it
		uint32_t full_mask, full_one2;		 // ring full: 0xffffffff / 0x00010001, else 0 / 0
		uint32_t ra_mask;					 // this stream keeps a running average (a launch built for one may hold streams without)
	};
	// MID: a frame in the middle of a group - the ring is full (and stays so), see lossy_const_run_kernel
	// DIFF: also hands out what the frame's sums are made of (h264.cpp:1993-2036) - |t - previous output| of the pair in *diff2, the pair's
	// "above the background" mask in *fg2 - for the speculative instance's byte plane (LossySpec::dplane)
	template <bool RA_ON, bool ADD_LOSS, bool MID = false, bool DIFF = false>
	__device__ __forceinline__ void const_pixel_pair(const ConstPairConsts &c, uint32_t v2, uint32_t old2, uint32_t last2, uint32_t &ref2, uint32_t &sum_lo, uint32_t &sum_hi,
													 uint32_t &cc2, uint32_t &cv2, uint32_t &t_in2, uint32_t &out2, uint32_t *diff2 = nullptr, uint32_t *fg2 = nullptr)
	{
		const lossy_u16x2 v = lp2(v2), ref = lp2(ref2);
		const lossy_u16x2 t = __builtin_elementwise_sub_sat(v, lp2(c.min2));
		t_in2 = lu1(t);
		const uint32_t fgm = lossy_nz_mask(__builtin_elementwise_sub_sat(v, lp2(c.bg2))); // v > background
		if constexpr (DIFF)
		{
			const lossy_u16x2 prev = lp2(out2);
			*diff2 = lu1(__builtin_elementwise_max(t, prev) - __builtin_elementwise_min(t, prev));
			*fg2 = fgm;
		}
		const lossy_u16x2 max_error = lp2(lossy_bfi(fgm, c.high2, c.low2));
		const lossy_u16x2 diff = __builtin_elementwise_max(t, ref) - __builtin_elementwise_min(t, ref);
		lossy_u16x2 nk = __builtin_elementwise_sub_sat(diff, max_error); // != 0: the pixel is not kept
		if (!ADD_LOSS)
			nk = nk | ((lp2(last2) ^ v) >> (lossy_u16x2){13, 13});
		const uint32_t NM = lossy_nz_mask(nk);
		if (RA_ON)
		{
			lossy_u16x2 cc = lp2(cc2);
			// what leaves the sum when the ring is full: the constant value while its stretch lasts, else the ring's oldest image
			uint32_t sel = lossy_bfi(lossy_nz_mask(cc), cv2, old2);
			if (!MID)
				sel &= c.full_mask;
			cc = __builtin_elementwise_sub_sat(cc, lp2(MID ? 0x00010001u : c.full_one2));
			const uint32_t t_lo = lu1(t) & 0xffffu, t_hi = lu1(t) >> 16;
			const uint32_t sm_lo = sum_lo + t_lo - (sel & 0xffffu), sm_hi = sum_hi + t_hi - (sel >> 16);
			// (MID: n_after > 1 - lossy_const_run_kernel - the division is the multiplication, without lossy_div's branch)
			const uint32_t q = MID ? (__umulhi(sm_lo, c.magic) | (__umulhi(sm_hi, c.magic) << 16)) : (lossy_div(sm_lo, c.magic) | (lossy_div(sm_hi, c.magic) << 16));
			sum_lo = (NM & 0xffffu) ? __umul24(t_lo, c.n_after) : sm_lo;
			sum_hi = (NM >> 16) ? __umul24(t_hi, c.n_after) : sm_hi;
			out2 = lossy_bfi(NM, lu1(t), lossy_bfi(c.ra_mask, q, ref2));
			cv2 = lossy_bfi(NM, lu1(t), cv2);
			cc2 = lossy_bfi(NM, c.n2, lu1(cc));
		}
		else
			out2 = lossy_bfi(NM, lu1(t), ref2);
		ref2 = lossy_bfi(NM, lu1(t), ref2);
	}

	// The sums of a frame over the workgroup (h264.cpp:1993-2036): threads 0..3 return the four words of the workgroup's share - fg count << 32
	// | fg sum d,  fg sum d2,  bg count << 32 | bg sum d,  bg sum d2.  red: 4 x 6 words of LDS.  Every thread of the workgroup calls.
	template <int NP>
	__device__ __noinline__ unsigned long long const_frame_sums(PxN<NP> v, PxN<NP> o, uint32_t background, bool lossy, int subtract_min, uint32_t mn, long long *red)
	{
		constexpr int PX = 2 * NP;
		const int tid = threadIdx.x;
		int32_t fd = 0, fn = 0, bd = 0, bn = 0;
		long long f2 = 0, b2 = 0;
#pragma unroll
		for (int q = 0; q < PX; ++q)
		{
			const uint32_t tq = subtract_min ? sub_min(v.get(q), mn) : v.get(q);
			const int32_t d = lossy ? abs((int32_t)tq - (int32_t)o.get(q)) : 0;
			const int32_t d2 = (int32_t)((uint32_t)d * (uint32_t)d);
			const int one = lossy ? 1 : 0;
			if (v.get(q) > background)
				fd += d, f2 += d2, fn += one;
			else
				bd += d, b2 += d2, bn += one;
		}
		const uint32_t wfd = lossy_wave_sum32((uint32_t)fd), wbd = lossy_wave_sum32((uint32_t)bd), wn = lossy_wave_sum32((uint32_t)fn | ((uint32_t)bn << 16));
		const long long wf2 = lossy_wave_sum(f2), wb2 = lossy_wave_sum(b2);
		const int lane = tid & 63, wave = tid >> 6;
		if (lane == 0)
			red[wave * 6 + 0] = (long long)wfd, red[wave * 6 + 1] = wf2, red[wave * 6 + 2] = (long long)(wn & 0xffffu), red[wave * 6 + 3] = (long long)wbd,
						 red[wave * 6 + 4] = wb2, red[wave * 6 + 5] = (long long)(wn >> 16);
		__syncthreads();
		long long val = 0;
		if (tid < 4)
		{
			const int a = tid == 0 ? 0 : tid == 1 ? 1 : tid == 2 ? 3 : 4;
			val = red[a] + red[6 + a] + red[12 + a] + red[18 + a];
			if (tid == 0 || tid == 2)
				val |= (red[a + 2] + red[6 + a + 2] + red[12 + a + 2] + red[18 + a + 2]) << 32;
		}
		__syncthreads();
		return (unsigned long long)val;
	}

	// The same sums over ONE WAVE, left in LDS (dst: the six words of this wave and frame: fg sum d, fg sum d2, fg count, bg sum d, bg sum d2,
	// bg count) - no barrier: the frames at the end of a group each have their own words, and the workgroup adds its waves up once, after
	// the last frame.
	template <int NP>
	__device__ __noinline__ void const_frame_sums_wave(PxN<NP> v, PxN<NP> o, uint32_t background, bool lossy, int subtract_min, uint32_t mn, long long *dst)
	{
		constexpr int PX = 2 * NP;
		int32_t fd = 0, fn = 0, bd = 0, bn = 0;
		long long f2 = 0, b2 = 0;
#pragma unroll
		for (int q = 0; q < PX; ++q)
		{
			const uint32_t tq = subtract_min ? sub_min(v.get(q), mn) : v.get(q);
			const int32_t d = lossy ? abs((int32_t)tq - (int32_t)o.get(q)) : 0;
			const int32_t d2 = (int32_t)((uint32_t)d * (uint32_t)d);
			const int one = lossy ? 1 : 0;
			if (v.get(q) > background)
				fd += d, f2 += d2, fn += one;
			else
				bd += d, b2 += d2, bn += one;
		}
		const uint32_t wfd = lossy_wave_sum32((uint32_t)fd), wbd = lossy_wave_sum32((uint32_t)bd), wn = lossy_wave_sum32((uint32_t)fn | ((uint32_t)bn << 16));
		const long long wf2 = lossy_wave_sum(f2), wb2 = lossy_wave_sum(b2);
		if ((threadIdx.x & 63) == 0)
			dst[0] = (long long)wfd, dst[1] = wf2, dst[2] = (long long)(wn & 0xffffu), dst[3] = (long long)wbd, dst[4] = wb2, dst[5] = (long long)(wn >> 16);
	}

	// SPEC: the speculative form (lossy_kernels.h: LossySpec) - the budgets come from the stream's table, frame by frame, the state after the
	// group and the ring's new images go to the SHADOW arrays, and no frame leaves sums (lossy_spec_stats_kernel takes them from the frames).
	template <int NP, bool RA_ON, bool ADD_LOSS, bool SPEC = false>
	__global__ __launch_bounds__(256) void lossy_const_run_kernel(const LossyRun *__restrict__ table, int nstreams, unsigned int *__restrict__ ok_word,
																  const unsigned int *__restrict__ poison, const LossySpec *__restrict__ spec)
	{
		constexpr int PX = 2 * NP;
		constexpr int D = kConstDepth;
		typedef PxN<NP> Px;
		__shared__ unsigned int sh_flag;
		__shared__ long long red[4][6];
		__shared__ __attribute__((aligned(16))) uint32_t sh_bg[kLossyConstMaxFrames + 8];
		// SPEC: the budgets of the group's frames, each in both halves of a word (what the packed comparisons take: no scalar arithmetic per frame)
		__shared__ __attribute__((aligned(16))) uint32_t sh_low2[SPEC ? kLossyConstMaxFrames + 8 : 4], sh_high2[SPEC ? kLossyConstMaxFrames + 8 : 4];
		__shared__ long long red_tail[SPEC ? 1 : kLossyConstTail][4][6]; // the sums of the last frames of a group, per wave (const_frame_sums_wave)
		const int tid = threadIdx.x, b = blockIdx.x, stream = blockIdx.y, nb = gridDim.x;
		LossyDeviceState sto; // where the state goes after the group
		if constexpr (SPEC)
		{ // (the launch before this one has decided: lossy_spec_begin_kernel / lossy_spec_verify_kernel)
			RIR_GLOBAL(const LossySpec) *sp = as_global(spec + stream);
			if (__hip_atomic_load(as_global(sp->ctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
				return;
			sto = lossy_load_struct(&(spec + stream)->shadow);
		}
		else
		{
			const bool ok = lossy_const_precondition(table, nstreams, poison, &sh_flag) && as_global(table + stream)->nsteps <= kLossyConstMaxFrames;
			if (b == 0 && stream == 0 && tid == 0)
				__hip_atomic_store(as_global(ok_word), ok ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (!ok)
				return;
		}
		LossyRun rp;
		{
			RIR_GLOBAL(const unsigned long long) *src = (RIR_GLOBAL(const unsigned long long) *)(table + stream);
			unsigned long long *dst = reinterpret_cast<unsigned long long *>(&rp);
#pragma unroll
			for (size_t k = 0; k < sizeof(LossyRun) / 8; ++k)
				dst[k] = src[k];
		}
		const LossyDeviceState st = rp.st;
		if constexpr (!SPEC)
			sto = st;
		RIR_GLOBAL(uint16_t) *refT = as_global(st.refT), *prevT = as_global(st.prevT), *lastDL = as_global(st.lastDL);
		RIR_GLOBAL(uint16_t) *cval = as_global(st.ra_const_value);
		RIR_GLOBAL(uint16_t) *ccnt = (RIR_GLOBAL(uint16_t) *)as_global(st.ra_const_count);
		RIR_GLOBAL(uint32_t) *sums = as_global(st.ra_sums);
		RIR_GLOBAL(const long long) *bgw = as_global(rp.bg);
		const int s = rp.s, full = rp.full, n = rp.nsteps;
		const int ra = RA_ON ? st.running_average : 0;
		const size_t ig = (size_t)b * 256 + tid; // this thread's group of PX pixels
		const bool inside = ig * PX < (size_t)full, lossy = ig * PX < (size_t)s;
		const uint32_t off_in = inside ? (uint32_t)(ig * PX * 2) : RIR_LOSSY_OOB, off_lossy = lossy ? (uint32_t)(ig * PX * 2) : RIR_LOSSY_OOB;
		const uint32_t lossy_mask = lossy ? 0xffffffffu : 0u;
		const uint32_t full_bytes = (uint32_t)full * 2u, s_bytes = (uint32_t)s * 2u;
		const uint64_t frame_bytes = (uint64_t)rp.frame_px * 2u, ring_bytes = (uint64_t)s * 2u;
		const uint64_t in0 = (uint64_t)rp.in, ring0 = (uint64_t)st.ra_images;
		const uint64_t ring_w0 = (uint64_t)sto.ra_images; // the ring the group's last images are written to (SPEC: the shadow ring, same slots)
		// SPEC: the byte plane of the group (LossySpec::dplane; null: none - every store out of range), a frame's bytes `s` further than the frame's before
		uint32_t dor = 0; // every difference of this thread's pixels, or-ed
		const uint32_t off_d = (SPEC && lossy) ? (uint32_t)(ig * PX) : RIR_LOSSY_OOB;
		__amdgpu_buffer_rsrc_t rs_d = lossy_rsrc(nullptr, 0u);
		if constexpr (SPEC)
		{
			const uint8_t *pl = as_global(spec + stream)->dplane;
			rs_d = lossy_rsrc(pl, pl ? (uint32_t)n * (uint32_t)s : 0u); // (n * s < 2^30: the host cuts groups so that n frames of 2 s bytes stay below 2^31)
		}
		// the bytes of a thread's pixels, from the pairs' differences and masks: byte = difference | 0x80 where the pixel is above the background
		auto plane_store = [&](const uint32_t (&dd)[NP], const uint32_t (&fg)[NP], uint32_t soff) {
			uint32_t w[NP];
#pragma unroll
			for (int p = 0; p < NP; ++p)
			{
				dor |= dd[p];
				w[p] = dd[p] | (fg[p] & 0x00800080u);
			}
			if constexpr (NP == 1)
				__builtin_amdgcn_raw_buffer_store_b16((short)__builtin_amdgcn_perm(w[0], w[0], 0x06040200u), rs_d, off_d, soff, 2);
			else if constexpr (NP == 2)
				__builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(w[1], w[0], 0x06040200u), rs_d, off_d, soff, 2);
			else
			{
				lossy_v2u_b q;
				q.x = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u), q.y = __builtin_amdgcn_perm(w[3], w[2], 0x06040200u);
				__builtin_amdgcn_raw_buffer_store_b64(q, rs_d, off_d, soff, 2);
			}
		};
		// the budget of every frame (lossy_budget with a statistic that is multiplied by zero)
		const int high_error = rp.high_value_error < 0 ? 0 : rp.high_value_error;
		const int low_error = rp.low_value_error < high_error ? high_error : rp.low_value_error;
		ConstPairConsts pc;
		pc.min2 = st.subtract_min ? lossy_both(st.min) : 0u;
		pc.low2 = lossy_both((uint32_t)low_error), pc.high2 = lossy_both((uint32_t)high_error);
		pc.ra_mask = ra > 0 ? 0xffffffffu : 0u;
		const __amdgpu_buffer_rsrc_t part_rsrc = lossy_rsrc(rp.partials, (uint32_t)((size_t)kLossyConstSlots * nb * 32));
		// the group's backgrounds, once, into LDS: a frame's background read from global memory inside the loop is a wave-uniform value the
		// compiler wants in a scalar register at once - a memory latency per frame and wave
		for (int k = tid; k < n; k += 256)
			sh_bg[k] = (uint32_t)bgw[(size_t)k * rp.bg_stride];
		if constexpr (SPEC)
		{
			RIR_GLOBAL(const uint32_t) *bud = as_global(as_global(spec + stream)->budgets);
			for (int k = tid; k < n + 8; k += 256)
			{
				const uint32_t e = k < n ? bud[k] : 0u;
				sh_low2[k] = lossy_both(e & 0xffffu), sh_high2[k] = lossy_both(e >> 16);
			}
		}
		__syncthreads();

		Px ref{}, last{}, cc{}, cv{}, o{}, t{};
		uint32_t sum[PX];
#pragma unroll
		for (int q = 0; q < PX; ++q)
			sum[q] = 0;
		int count = st.ra_count, head = st.ra_head;
		if (lossy)
		{
			ref = ldn<NP>(refT, ig);
			last = ldn<NP>(lastDL, ig);
			o = ldn<NP>(prevT, ig);
			if (ra > 0)
			{
#pragma unroll
				for (int q = 0; q < PX; ++q)
					sum[q] = sums[ig * PX + q];
				cc = ldn<NP>(ccnt, ig);
				cv = ldn<NP>(cval, ig);
			}
		}
		else if (inside)
			last = ldn<NP>(lastDL, ig);
		// Frame kf is REQUESTED: its pixels, its background, and what leaves the running average at it when the ring is full then - the
		// input of frame kf - ra (less the minimum) once that is a frame of this group, else the ring's image at the head as it is at frame
		// kf: an image from before the group, which the group has not overwritten (it only writes the ring in its last `ra` frames, each
		// slot after it has been read).  All wave-uniform choices, all addresses running sums: one load each, whatever the case.
		Px V[D], O[D];
		int kf = 0;					 // the next frame to request
		uint64_t req_in = in0;		 // its pixels
		int req_count = count;		 // images in the ring when it is stepped
		// the ring slot the head is at when that frame is stepped (valid once the ring is full), as a running address
		const uint64_t ring_end = ring0 + (uint64_t)(ra > 0 ? ra : 1) * ring_bytes;
		uint64_t req_old = ring0 + (uint64_t)head * ring_bytes;
		const uint64_t ra_back = (uint64_t)ra * frame_bytes;
		auto request = [&](Px &v, Px &old) {
			const bool more = kf < n;
			v = buf_ldn<NP>(lossy_rsrc((const void *)req_in, more ? full_bytes : 0u), off_in);
			if (RA_ON)
			{
				const bool full_then = req_count == ra;
				const bool need_old = more && ra > 0 && full_then;
				const bool from_in = kf >= ra;
				const uint64_t ob = from_in ? req_in - ra_back : req_old;
				old = buf_ldn<NP>(lossy_rsrc((const void *)ob, need_old ? s_bytes : 0u), off_lossy);
				// (the head moves once per frame from the moment the ring is full; until then the ring grows)
				const uint64_t nxt = req_old + ring_bytes;
				req_old = full_then ? (nxt == ring_end ? ring0 : nxt) : req_old;
				req_count += full_then ? 0 : 1;
			}
			req_in += frame_bytes;
			++kf;
		};
#pragma unroll
		for (int j = 0; j < D; ++j)
			request(V[j], O[j]);
		uint32_t bg_next = sh_bg[0]; // (a frame's background is read from LDS a frame ahead)
		uint32_t low_next = SPEC ? sh_low2[0] : 0u, high_next = SPEC ? sh_high2[0] : 0u;
		// per-frame constants that only move while the ring fills
		auto ring_consts = [&]() {
			const int n_after = ra > 0 ? (count == ra ? ra : count + 1) : 0;
			pc.n_after = (uint32_t)n_after, pc.magic = lossy_div_magic(n_after), pc.n2 = (uint32_t)n_after * 0x10001u;
			pc.full_mask = (ra > 0 && count == ra) ? 0xffffffffu : 0u, pc.full_one2 = pc.full_mask & 0x00010001u;
		};
		ring_consts();
		// (SPEC: no frame leaves sums; the frames from tail0 on only differ in that the frame D ahead of them may not exist)
		const int tail0 = SPEC ? (n > D ? n - D : 0) : (n > kLossyConstTail ? n - kLossyConstTail : 0);
		const int ring_from = n - ra; // frames from here on are in the ring after the group
		uint64_t out_p = (uint64_t)rp.out;
		// where a frame's input goes in the ring: the slot after the newest image - one further per frame, full ring or not
		int wr_slot = (head + (count == ra ? 0 : count)) % (ra > 0 ? ra : 1);
		const uint64_t ring_w_end = ring_w0 + (uint64_t)(ra > 0 ? ra : 1) * ring_bytes;
		uint64_t wr_p = ring_w0 + (uint64_t)wr_slot * ring_bytes;
		auto step = [&](int k, Px &Vj, Px &Oj) {
			const Px v = Vj;
			Px old = Oj;
			if (RA_ON)
			{ // (what leaves the average came from the input - frame k - ra, still with its minimum - or from the ring, where it is stored without)
				const uint32_t mj = k >= ra ? pc.min2 : 0u;
#pragma unroll
				for (int p = 0; p < NP; ++p)
					old.d[p] = lu1(__builtin_elementwise_sub_sat(lp2(old.d[p]), lp2(mj)));
			}
			const uint32_t background = (uint32_t)__builtin_amdgcn_readfirstlane((int)bg_next); // (wave-uniform: what is derived from it stays on the scalar unit)
			bg_next = sh_bg[k + 1 < n ? k + 1 : k];
			if constexpr (SPEC)
			{
				pc.low2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)low_next), pc.high2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)high_next);
				low_next = sh_low2[k + 1 < n ? k + 1 : k], high_next = sh_high2[k + 1 < n ? k + 1 : k];
			}
			request(Vj, Oj);
			// the frame's sums, if its statistic will be in the window (or seeds it): against the previous output, per workgroup (out of
			// line: frame 0 and, where a group is too short for the middle loop, its last 40 frames come here; the loop body is unrolled kConstDepth times)
			lossy_v2u_b pval = {0u, 0u};
			uint32_t poff = RIR_LOSSY_OOB;
			if (!SPEC && (k >= tail0 || k == 0))
			{
				const int slot = k >= tail0 ? 1 + (k - tail0) : 0;
				const unsigned long long val = const_frame_sums<NP>(v, o, background, lossy, st.subtract_min, st.min, &red[0][0]);
				if (tid < 4)
				{
					pval.x = (uint32_t)val, pval.y = (uint32_t)(val >> 32);
					poff = (uint32_t)((((size_t)slot * nb + b) * 4 + tid) * 8);
				}
			}
			if constexpr (!SPEC)
				__builtin_amdgcn_raw_buffer_store_b64(pval, part_rsrc, poff, 0, 0); // (every frame: out of range unless the frame leaves sums)
			pc.bg2 = lossy_both(background);
			Px ov;
			uint32_t dd[NP], fg[NP];
#pragma unroll
			for (int p = 0; p < NP; ++p)
			{
				const_pixel_pair<RA_ON, ADD_LOSS, false, SPEC>(pc, v.d[p], old.d[p], last.d[p], ref.d[p], sum[2 * p], sum[2 * p + 1], cc.d[p], cv.d[p], t.d[p], o.d[p], &dd[p], &fg[p]);
				ov.d[p] = lossy_bfi(lossy_mask, o.d[p], v.d[p]); // rows past lossy_height: stored as they are
			}
			if constexpr (SPEC)
				plane_store(dd, fg, (uint32_t)k * (uint32_t)s);
			last = v;
			buf_stn<NP, (SPEC ? RIR_SPEC_OUT_STORE_AUX : RIR_CONST_OUT_STORE_AUX)>(ov, lossy_rsrc((const void *)out_p, full_bytes), off_in);
			out_p += frame_bytes;
			if (RA_ON)
			{
				// the ring as it must be after the group: the last `ra` inputs
				buf_stn<NP>(t, lossy_rsrc((const void *)wr_p, (ra > 0 && k >= ring_from) ? s_bytes : 0u), off_lossy);
				const uint64_t nxt = wr_p + ring_bytes;
				wr_p = nxt == ring_w_end ? ring_w0 : nxt;
				wr_slot = nxt == ring_w_end ? 0 : wr_slot + 1;
				if (count != ra)
				{ // (only while the ring fills: the constants of the running average move)
					++count;
					ring_consts();
				}
			}
		};
		// ---- the frames in the MIDDLE of a group: [mid0, mid1), both multiples of D ----
		// From frame max(ra, 1) on the ring is full and what leaves the average is an input frame of the group; before frame n - 40 no frame
		// leaves sums, before frame n - ra none is written to the ring; four frames ahead there still is a frame.  For those frames - all but
		// ~50 of a group of 1 000 - every wave-uniform choice of step() is known, and what is left of the scalar side of a frame is ONE
		// addition: the three arrays of the frame (input four frames ahead, the input `ra` frames before that one, output) are one
		// descriptor each for the whole phase - bases shifted so that the same scalar offset k * frame_bytes serves all three - and the
		// frame's background comes four to a 16-byte LDS read.  (step() above is ~60 scalar and ~60 vector instructions per frame and pair,
		// and a wave issues one instruction at a time: the scalar half of that was half of the kernel's time.)  Same values, bit for bit.
		int mid0 = ((ra > 1 ? ra : 1) + D - 1) / D * D, mid1 = mid0;
		{
			const int end = (tail0 < n - ra ? tail0 : n - ra);
			// every offset of the phase below 2^31, the mark of a lane without a pixel (RIR_LOSSY_OOB)
			const bool small = (uint64_t)(n + ra + D) * frame_bytes < 0x80000000ull;
			if (small && end > mid0 && ra != 1) // (ra == 1: an "average" of one image, no division - lossy_div's other branch: step() has it)
				mid1 = mid0 + (end - mid0) / D * D;
		}
		auto middle = [&]() {
			const uint32_t fb = (uint32_t)frame_bytes;
			const uint64_t group_bytes = (uint64_t)n * frame_bytes;
			const __amdgpu_buffer_rsrc_t rs_in = lossy_rsrc((const void *)(in0 + (uint64_t)D * frame_bytes), (uint32_t)(group_bytes - (uint64_t)D * frame_bytes));
			const __amdgpu_buffer_rsrc_t rs_old = lossy_rsrc((const void *)(in0 + (uint64_t)D * frame_bytes - ra_back),
															 ra > 0 ? (uint32_t)(group_bytes - (uint64_t)D * frame_bytes + ra_back) : 0u);
			const __amdgpu_buffer_rsrc_t rs_out = lossy_rsrc((const void *)(uint64_t)rp.out, (uint32_t)group_bytes);
			uint32_t so = (uint32_t)mid0 * fb;
			uint32_t so_d = (uint32_t)mid0 * (uint32_t)s; // (SPEC: where the frame's bytes go in the plane)
			uint32_t bgq[D], bgn[D], loq[D], lon[D], hiq[D], hin[D];
			auto backgrounds = [&](int k) { // of frames k .. k + D - 1 (k a multiple of D; the array is padded) - SPEC: and their budgets
				if constexpr (D == 4)
				{
					const lossy_v4u q = *reinterpret_cast<const lossy_v4u *>(&sh_bg[k]);
					bgn[0] = q.x, bgn[1] = q.y, bgn[2] = q.z, bgn[3] = q.w;
					if constexpr (SPEC)
					{
						const lossy_v4u lo = *reinterpret_cast<const lossy_v4u *>(&sh_low2[k]), hi = *reinterpret_cast<const lossy_v4u *>(&sh_high2[k]);
						lon[0] = lo.x, lon[1] = lo.y, lon[2] = lo.z, lon[3] = lo.w;
						hin[0] = hi.x, hin[1] = hi.y, hin[2] = hi.z, hin[3] = hi.w;
					}
				}
				else
				{
#pragma unroll
					for (int j = 0; j < D; ++j)
					{
						bgn[j] = sh_bg[k + j];
						if constexpr (SPEC)
							lon[j] = sh_low2[k + j], hin[j] = sh_high2[k + j];
					}
				}
			};
			auto uniform = [&]() { // the next D frames' backgrounds (and budgets) into scalar registers
#pragma unroll
				for (int j = 0; j < D; ++j)
				{
					bgq[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)bgn[j]);
					if constexpr (SPEC)
						loq[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lon[j]), hiq[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hin[j]);
				}
			};
			backgrounds(mid0);
			V[D - 1] = last; // (frame mid0 + D - 1, which step() has asked for, is asked for again by the first step below)
			const __amdgpu_buffer_rsrc_t rs_ring_on = lossy_rsrc((const void *)ring_w0, (uint32_t)((uint64_t)(ra > 0 ? ra : 0) * ring_bytes));
			const __amdgpu_buffer_rsrc_t rs_ring_off = lossy_rsrc((const void *)ring_w0, 0u);
			uint32_t so_ring = 0; // (set where the tail begins)
			// One frame, slot j of the ring of registers.  TAIL: a frame of the group's end - it may leave sums (per wave, in LDS), it may go into
			// the ring of images, and the frame D ahead of it may not exist.
			auto frame = [&](auto tail_tag, int j, int k) {
				constexpr bool TAIL = decltype(tail_tag)::value;
				const Px v = V[j];
				Px old = O[j];
				if (RA_ON)
				{
#pragma unroll
					for (int p = 0; p < NP; ++p)
						old.d[p] = lu1(__builtin_elementwise_sub_sat(lp2(old.d[p]), lp2(pc.min2)));
				}
				if (!SPEC && TAIL && k >= tail0)
					const_frame_sums_wave<NP>(v, o, bgq[j], lossy, st.subtract_min, st.min, &red_tail[k - tail0][tid >> 6][0]);
				pc.bg2 = lossy_both(bgq[j]);
				if constexpr (SPEC)
					pc.low2 = loq[j], pc.high2 = hiq[j];
				const Px before = V[(j + D - 1) % D]; // the frame before this one: still in its slot
				Px ov;
				uint32_t dd[NP], fg[NP];
#pragma unroll
				for (int p = 0; p < NP; ++p)
				{
					const_pixel_pair<RA_ON, ADD_LOSS, true, SPEC>(pc, v.d[p], old.d[p], before.d[p], ref.d[p], sum[2 * p], sum[2 * p + 1], cc.d[p], cv.d[p], t.d[p], o.d[p], &dd[p], &fg[p]);
					ov.d[p] = lossy_bfi(lossy_mask, o.d[p], v.d[p]);
				}
				buf_stn<NP, (SPEC ? RIR_SPEC_OUT_STORE_AUX : RIR_CONST_OUT_STORE_AUX)>(ov, rs_out, off_in, so);
				if constexpr (SPEC)
				{
					plane_store(dd, fg, so_d);
					so_d += (uint32_t)s;
				}
				if (TAIL && RA_ON)
				{ // the ring as it must be after the group: the last `ra` inputs (less the minimum), each in the slot after the one before
					buf_stn<NP>(t, k >= ring_from ? rs_ring_on : rs_ring_off, off_lossy, so_ring);
					so_ring += (uint32_t)ring_bytes;
					so_ring = so_ring >= (uint32_t)((uint64_t)(ra > 0 ? ra : 1) * ring_bytes) ? 0u : so_ring;
				}
				if (TAIL && k == n - 1)
					last = v;
				// A slot's next frame is asked for when nothing needs the frame in it any more - in the step AFTER its own, which compares with
				// it (lastDL): asked for earlier the two are alive together, and the compiler copies all D slots out of the way at the top of
				// the iteration, after waiting for every one of them.  D - 1 frames are in flight.
				const uint32_t off_o = !TAIL || k + D < n ? off_lossy : RIR_LOSSY_OOB, off_v = !TAIL || k - 1 + D < n ? off_in : RIR_LOSSY_OOB;
				if (RA_ON)
					O[j] = buf_ldn<NP>(rs_old, off_o, so);
				V[(j + D - 1) % D] = buf_ldn<NP>(rs_in, off_v, so - fb);
				so += fb;
#if RIR_LOSSY_CONST_PIN_STEPS
				// a frame's instructions stay together: left alone, the scheduler gathers the loads of two or three frames at the end of the
				// iteration, and its top then waits for loads issued 40 instructions ago
				__builtin_amdgcn_sched_barrier(0);
#endif
			};
			for (int k0 = mid0; k0 < mid1; k0 += D)
			{
				uniform();
				backgrounds(k0 + D); // (k0 + D <= mid1 < n)
#pragma unroll
				for (int j = 0; j < D; ++j)
					frame(std::false_type{}, j, k0 + j);
			}
			// ---- the frames from mid1 to the end of the group: the same body with the three things a frame of the end may have to do ----
			so_ring = (uint32_t)((uint64_t)((wr_slot + (mid1 - mid0)) % (ra > 0 ? ra : 1)) * ring_bytes); // (the slot frame mid1 goes to: one further per frame)
			for (int k0 = mid1; k0 < n; k0 += D)
			{
				uniform();
				backgrounds(k0 + D); // (the array is padded by D)
#pragma unroll
				for (int j = 0; j < D; ++j)
					if (k0 + j < n)
						frame(std::true_type{}, j, k0 + j);
			}
			// the workgroup's sums of the frames of the end: its four waves added up, where step() leaves them
			if constexpr (SPEC)
				return;
			__syncthreads();
			if (tid < kLossyConstTail * 4 && tail0 + tid / 4 < n)
			{
				const int f = tid / 4, wd = tid & 3; // words: fg count << 32 | fg sum d,  fg sum d2,  bg count << 32 | bg sum d,  bg sum d2
				const int a = wd == 0 ? 0 : wd == 1 ? 1 : wd == 2 ? 3 : 4;
				long long val = red_tail[f][0][a] + red_tail[f][1][a] + red_tail[f][2][a] + red_tail[f][3][a];
				if (wd == 0 || wd == 2)
					val |= (red_tail[f][0][a + 2] + red_tail[f][1][a + 2] + red_tail[f][2][a + 2] + red_tail[f][3][a + 2]) << 32;
				as_global(rp.partials)[((size_t)(1 + f) * nb + b) * 4 + wd] = (unsigned long long)val;
			}
		};
		int k0 = 0;
		bool done = false;
		for (; k0 + D <= n; k0 += D)
		{ // whole iterations of D unconditional steps
			if (k0 == mid0 && mid1 > mid0)
			{
				middle(); // (all the frames from mid0 on)
				done = true;
				break;
			}
#pragma unroll
			for (int j = 0; j < D; ++j)
				step(k0 + j, V[j], O[j]);
		}
		// up to D - 1 left-over frames (the slot rotation stays aligned)
		if (!done)
		{
#pragma unroll
			for (int j = 0; j < D - 1; ++j)
				if (k0 + j < n)
					step(k0 + j, V[j], O[j]);
		}
		if constexpr (SPEC)
		{ // a difference that does not fit its seven bits: this pass's sums are taken from the frames (lossy_spec_stats_kernel<false>)
			if (lossy && (dor & 0xff80ff80u) != 0u)
				__hip_atomic_store(as_global(as_global(spec + stream)->ctl) + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		RIR_GLOBAL(uint16_t) *refT_o = as_global(sto.refT), *prevT_o = as_global(sto.prevT), *lastDL_o = as_global(sto.lastDL);
		if (lossy)
		{
			stn<NP>(refT_o, ig, ref);
			stn<NP>(lastDL_o, ig, last);
			stn<NP>(prevT_o, ig, o);
			if (ra > 0)
			{
				RIR_GLOBAL(uint32_t) *sums_o = as_global(sto.ra_sums);
#pragma unroll
				for (int q = 0; q < PX; ++q)
					sums_o[ig * PX + q] = sum[q];
				stn<NP>((RIR_GLOBAL(uint16_t) *)as_global(sto.ra_const_count), ig, cc);
				stn<NP>(as_global(sto.ra_const_value), ig, cv);
			}
		}
		else if (inside)
			stn<NP>(lastDL_o, ig, last);
	}

	// grid = streams, 1 024 threads.  The window entries of the frames that left sums (exact integers -> the reference's double
	// arithmetic, lossy_budget's), the window's counters as they are after the group's n frames, the budgets of the frames.
	__global__ __launch_bounds__(1024) void lossy_const_finish_kernel(const LossyRun *__restrict__ table, int nb, const unsigned int *__restrict__ ok_word)
	{
		__shared__ double sd[kLossyConstSlots][2];
		if (*as_global(ok_word) == 0u)
			return;
		const int tid = threadIdx.x;
		RIR_GLOBAL(const LossyRun) *r = as_global(table + blockIdx.x);
		RIR_GLOBAL(const unsigned long long) *partials = as_global(r->partials);
		RIR_GLOBAL(LossyBudget) *bud = as_global(r->budget);
		const int n = r->nsteps, s = r->s;
		const int n_first0 = bud->n_first, n_win0 = bud->n_win, head0 = bud->head;
		const int tail0 = n > kLossyConstTail ? n - kLossyConstTail : 0;
		// 16 lanes to a slot (a frame that left sums), each adds up a sixteenth of the workgroups' rows - independent loads, many in flight -
		// and the sixteen are added within their row of the wave
		{
			const int slot = tid >> 4, part = tid & 15;
			const int k = slot == 0 ? 0 : tail0 + slot - 1; // the frame of this slot
			const bool live = slot < kLossyConstSlots && k < n && !(slot == 0 && tail0 == 0); // (slot 0 is only filled when frame 0 is not in the tail)
			long long fd = 0, fn = 0, bd = 0, bn = 0, f2 = 0, b2 = 0;
			if (live)
			{
				RIR_GLOBAL(const lossy_v4u) *rows = reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(partials + (size_t)slot * nb * 4);
#pragma unroll 4
				for (int w = part; w < nb; w += 16)
				{
					const lossy_v4u lo = rows[2 * w], hi = rows[2 * w + 1]; // words 0, 1 | 2, 3
					fd += (long long)lo.x, fn += (long long)lo.y, f2 += (long long)(((unsigned long long)lo.w << 32) | lo.z);
					bd += (long long)hi.x, bn += (long long)hi.y, b2 += (long long)(((unsigned long long)hi.w << 32) | hi.z);
				}
			}
#pragma unroll
			for (int d = 8; d >= 1; d >>= 1)
			{ // (every lane takes part: __shfl_xor within 16 lanes)
				fd += __shfl_xor(fd, d, 16), fn += __shfl_xor(fn, d, 16), bd += __shfl_xor(bd, d, 16), bn += __shfl_xor(bn, d, 16);
				f2 += __shfl_xor(f2, d, 16), b2 += __shfl_xor(b2, d, 16);
			}
			if (live && part == 0)
			{ // stdDev (h264.cpp:1993-2036), as lossy_budget: unsplit while the window is not full
				const int n_win_k = n_win0 + k < 40 ? n_win0 + k : 40;
				if (n_win_k < 40)
				{
					const double sum_diff = (double)(fd + bd), sum_diff2 = (double)(f2 + b2);
					sd[slot][0] = sd[slot][1] = sqrt(sum_diff * sum_diff - sum_diff2) / s;
				}
				else
				{
					const double dfd = (double)fd, dfd2 = (double)f2, dbd = (double)bd, dbd2 = (double)b2;
					sd[slot][0] = sqrt(dbd * dbd - dbd2) / (int)bn;
					sd[slot][1] = sqrt(dfd * dfd - dfd2) / (int)fn;
				}
			}
		}
		__syncthreads();
		if (tid == 0)
		{
			const int first_slot = tail0 == 0 ? 1 : 0;
			if (n_first0 < 1)
			{
				bud->first_std[0] = sd[first_slot][0], bud->first_std[1] = sd[first_slot][1];
				bud->n_first = 1;
			}
			// the frames before the tail go through the window and are gone by the end of the group: only the counters move
			int n_win = n_win0, head = head0;
			{
				const int fill = tail0 < 40 - n_win ? tail0 : 40 - n_win;
				n_win += fill;
				head = (head + (tail0 - fill)) % 40;
			}
			for (int k = tail0; k < n; ++k)
			{
				const int slot = 1 + (k - tail0);
				if (n_win < 40)
				{
					bud->win[n_win][0] = sd[slot][0], bud->win[n_win][1] = sd[slot][1];
					++n_win;
				}
				else
				{
					bud->win[head][0] = sd[slot][0], bud->win[head][1] = sd[slot][1];
					head = head == 39 ? 0 : head + 1;
				}
			}
			bud->n_win = n_win, bud->head = head;
			RIR_GLOBAL(LossyDecision) *gd = as_global(r->decision);
			const int high_error = r->high_value_error < 0 ? 0 : r->high_value_error;
			gd->background = (uint32_t)as_global(r->bg)[(size_t)(n - 1) * r->bg_stride];
			gd->high_error = high_error, gd->low_error = r->low_value_error < high_error ? high_error : r->low_value_error;
		}
		if (r->errors_out)
		{
			const int high_error = r->high_value_error < 0 ? 0 : r->high_value_error;
			const int low_error = r->low_value_error < high_error ? high_error : r->low_value_error;
			RIR_GLOBAL(int) *e = as_global(r->errors_out);
			for (int k = tid; k < n; k += 1024)
				e[2 * k] = low_error, e[2 * k + 1] = high_error;
		}
	}


	// ---- the speculative form of a run (lossy_kernels.h: LossySpec) -------------------------------------------------------------------
	//
	// lossy_spec_begin_kernel: one workgroup.  The group is offered when the constant-budget form's precondition holds for every stream (no
	// class that may be empty, no NaN in a window, no poison), every group fits the streaming kernel and the leading stream is not backing
	// off; then every stream's table gets the guess - the configured errors, clamped as lossy_budget clamps them - and status 0.
	__device__ __forceinline__ uint32_t lossy_spec_pack(int low_error, int high_error)
	{ // (a difference of two 16-bit values never exceeds 65 535: larger budgets decide as 65 535 does)
		const uint32_t lo = (uint32_t)(low_error > 65535 ? 65535 : low_error), hi = (uint32_t)(high_error > 65535 ? 65535 : high_error);
		return lo | (hi << 16);
	}
	__global__ __launch_bounds__(1024) void lossy_spec_begin_kernel(const LossyRun *__restrict__ table, const LossySpec *__restrict__ spec, int nstreams, int passes,
																	unsigned int *__restrict__ ok_word, const unsigned int *__restrict__ poison)
	{
		__shared__ unsigned int sh_flag;
		const int tid = threadIdx.x;
		bool ok = lossy_const_precondition(table, nstreams, poison, &sh_flag);
		for (int q = 0; q < nstreams; ++q)
			ok = ok && as_global(table + q)->nsteps <= kLossyConstMaxFrames;
		RIR_GLOBAL(unsigned int) *bk = as_global(as_global(spec)->backoff);
		const unsigned int skip = bk[0];
		__syncthreads();
		if (ok && skip > 0u)
		{ // a stream whose groups keep failing: this one goes straight to the resident kernel
			if (tid == 0)
			{
				bk[0] = skip - 1u;
				if (as_global(spec)->backoff_host)
					__hip_atomic_store(as_global(as_global(spec)->backoff_host), skip - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
			ok = false;
		}
		if (tid == 0)
			__hip_atomic_store(as_global(ok_word), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		for (int q = 0; q < nstreams; ++q)
		{
			RIR_GLOBAL(const LossyRun) *r = as_global(table + q);
			RIR_GLOBAL(const LossySpec) *sp = as_global(spec + q);
			if (tid == 0)
			{
				RIR_GLOBAL(unsigned int) *ctl = as_global(sp->ctl);
				ctl[0] = ok ? 0u : 2u, ctl[1] = 0u, ctl[2] = 0u, ctl[3] = (unsigned int)(passes & 0xff), ctl[5] = 0u, ctl[6] = (unsigned int)passes >> 8;
				ctl[4] = ok ? 1u : 0u; // (the group was offered: rir_lossy_spec_stats)
			}
			if (ok)
			{
				const int high_error = r->high_value_error < 0 ? 0 : r->high_value_error;
				const int low_error = r->low_value_error < high_error ? high_error : r->low_value_error;
				const uint32_t guess = lossy_spec_pack(low_error, high_error);
				RIR_GLOBAL(uint32_t) *bud = as_global(sp->budgets);
				for (int k = tid; k < r->nsteps; k += 1024)
					bud[k] = guess;
			}
		}
	}

	// lossy_spec_stats_kernel: grid = (slabs of a frame, frames / kLossySpecStatFrames, streams).  The six sums of frame k (h264.cpp:1993-2036) from the frames where
	// they lie: input k (less the minimum) against output k - 1 - the stream's prevT for the group's first frame, which a pass leaves alone -
	// split by input k > background k.  Per slab of kLossySpecSlab pixels four words, as lossy_const_run_kernel's partials; the LAST slab of
	// a frame to arrive (a ticket per frame; rows written and read with agent-scope accesses, as lossy_last_arriver's callers do) adds them up and
	// leaves the frame's statistic, stdDev's double arithmetic on the exact sums.
#ifndef RIR_SPEC_STATS_AUX
#define RIR_SPEC_STATS_AUX 2 /* cache policy of the sums kernel's loads: nt (every byte is read once; measured +4 % on the whole call against the default policy, sc1 no better) */
#endif
	// PLANE: the same sums from the byte plane the streaming kernel left (LossySpec::dplane: difference and class of every pixel in one byte - a
	// quarter of the bytes of the two frames, four pixels to an instruction through v_dot4_u32_u8).  Both forms are queued behind every pass: the one
	// from the plane does the work unless there is no plane or the pass met a difference of 128 or more (ctl[7]), then the one from the frames does.
	// one frame's share of a workgroup (its slab): the sums, the row, and - the last slab of the frame to arrive - the frame's statistic
	template <bool PLANE>
	__device__ __forceinline__ void lossy_spec_stats_frame(RIR_GLOBAL(const LossySpec) *sp, RIR_GLOBAL(const LossyRun) *r, int k, int slab, int nslabs, long long (*red)[6],
														   unsigned int *sh_last_)
	{
		unsigned int &sh_last = *sh_last_;
		const int tid = threadIdx.x;
		const int s = r->s;
		const int i0 = slab * kLossySpecSlab, i1 = min(i0 + kLossySpecSlab, s); // (s is a multiple of 8: the callers' `runs`)
		uint32_t fd = 0, bd = 0, fn = 0, bn = 0;
		long long f2 = 0, b2 = 0;
		if constexpr (PLANE)
		{
			// 16 pixels per 16-byte load; a lane past the end of the lossy rows: out of range - zeros: no difference, no class, counted nowhere but in `valid`
			const __amdgpu_buffer_rsrc_t rs = lossy_rsrc(sp->dplane + (size_t)k * (size_t)s, (uint32_t)i1);
			constexpr int kIter = kLossySpecSlab / 16 / 256;
			lossy_v4u a[kIter];
			uint32_t valid = 0;
#pragma unroll
			for (int j = 0; j < kIter; ++j)
			{
				const uint32_t px = (uint32_t)i0 + (uint32_t)(j * 256 + tid) * 16u;
				a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, px, 0, RIR_SPEC_STATS_AUX);
				valid += px < (uint32_t)i1 ? min(16u, (uint32_t)i1 - px) : 0u;
			}
			// (differences below 128, 64 pixels of a thread: sum d < 2^13, sum d^2 < 2^20)
			uint32_t sd_all = 0, sd_fg = 0, s2_all = 0, s2_fg = 0, n_fg = 0;
#pragma unroll
			for (int j = 0; j < kIter; ++j)
#pragma unroll
				for (int q = 0; q < 4; ++q)
				{
					const uint32_t x = a[j][q], d4 = x & 0x7f7f7f7fu, f4 = (x >> 7) & 0x01010101u, dfg4 = d4 & (f4 * 0x7fu);
					sd_all = __builtin_amdgcn_udot4(d4, 0x01010101u, sd_all, false), sd_fg = __builtin_amdgcn_udot4(dfg4, 0x01010101u, sd_fg, false);
					s2_all = __builtin_amdgcn_udot4(d4, d4, s2_all, false), s2_fg = __builtin_amdgcn_udot4(dfg4, dfg4, s2_fg, false);
					n_fg = __builtin_amdgcn_udot4(f4, 0x01010101u, n_fg, false);
				}
			fd = sd_fg, bd = sd_all - sd_fg, fn = n_fg, bn = valid - n_fg;
			f2 = (long long)s2_fg, b2 = (long long)(s2_all - s2_fg);
		}
		else
		{
			const size_t frame_px = (size_t)r->frame_px;
			const uint16_t *in_k = r->in + (size_t)k * frame_px;
			const uint16_t *prev = k == 0 ? (const uint16_t *)r->st.prevT : (const uint16_t *)r->out + (size_t)(k - 1) * frame_px;
			const uint32_t background = (uint32_t)as_global(r->bg)[(size_t)k * r->bg_stride];
			const int subtract_min = r->st.subtract_min;
			const uint32_t mn = r->st.min;
			const uint32_t bytes = (uint32_t)i1 * 2u;
			const __amdgpu_buffer_rsrc_t rs_in = lossy_rsrc(in_k, bytes), rs_pv = lossy_rsrc(prev, bytes);
			constexpr int kIter = kLossySpecSlab / 8 / 256; // 16-byte loads per thread and array
			lossy_v4u a[kIter], p[kIter];
			uint32_t valid = 0; // pixels of this thread that lie in the lossy rows
#pragma unroll
			for (int j = 0; j < kIter; ++j)
			{ // (a lane past the end of the lossy rows: out of range - zeros against zeros: d = 0, counted nowhere but in `valid`)
				const uint32_t off = (uint32_t)(i0 / 8 + j * 256 + tid) * 16u;
				a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, RIR_SPEC_STATS_AUX);
				p[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_pv, off, 0, RIR_SPEC_STATS_AUX);
				valid += (i0 / 8 + j * 256 + tid) * 8 < i1 ? 8u : 0u;
			}
			// Two pixels to an instruction, the sums through v_dot2_u32_u16: d = max - min, the foreground mask as in const_pixel_pair, sum d and
			// sum d^2 over everything and over the foreground (the background is the difference).  Exact as long as every d is below 4 096 (64 pixels
			// of a thread: sum d^2 < 2^30) - thermal frames are; a wave that meets a larger one does its pixels again one by one, as the host code does
			// (h264.cpp:1993-2036: the square wraps at 32 bits and is added as a signed number).
			{
				const uint32_t min2 = subtract_min ? lossy_both(mn) : 0u, bg2 = lossy_both(background);
				const lossy_u16x2 ones = {1, 1};
				uint32_t sd_all = 0, sd_fg = 0, s2_all = 0, s2_fg = 0, n_fg = 0, dor = 0;
#pragma unroll
				for (int j = 0; j < kIter; ++j)
#pragma unroll
					for (int q = 0; q < 4; ++q)
					{
						const lossy_u16x2 v = lp2(a[j][q]), o = lp2(p[j][q]);
						const lossy_u16x2 t = __builtin_elementwise_sub_sat(v, lp2(min2));
						const lossy_u16x2 d = __builtin_elementwise_max(t, o) - __builtin_elementwise_min(t, o);
						const uint32_t fgm = lossy_nz_mask(__builtin_elementwise_sub_sat(v, lp2(bg2))); // v > background
						const lossy_u16x2 dfg = lp2(lu1(d) & fgm);
						dor |= lu1(d);
						sd_all = __builtin_amdgcn_udot2(d, ones, sd_all, false), sd_fg = __builtin_amdgcn_udot2(dfg, ones, sd_fg, false);
						s2_all = __builtin_amdgcn_udot2(d, d, s2_all, false), s2_fg = __builtin_amdgcn_udot2(dfg, dfg, s2_fg, false);
						n_fg = __builtin_amdgcn_udot2(lp2(fgm & 0x00010001u), ones, n_fg, false);
					}
				if (__builtin_expect(__ballot((dor & 0xf000f000u) != 0u) == 0ull, 1))
				{
					fd = sd_fg, bd = sd_all - sd_fg, fn = n_fg, bn = valid - n_fg;
					f2 = (long long)s2_fg, b2 = (long long)(s2_all - s2_fg);
				}
				else
				{
#pragma unroll
					for (int j = 0; j < kIter; ++j)
					{
						const bool in = (i0 / 8 + j * 256 + tid) * 8 < i1;
#pragma unroll
						for (int q = 0; q < 8; ++q)
						{
							const uint32_t aw = a[j][q >> 1], pw = p[j][q >> 1];
							const uint32_t v = (q & 1) ? aw >> 16 : aw & 0xffffu, o = (q & 1) ? pw >> 16 : pw & 0xffffu;
							const uint32_t t = subtract_min ? sub_min(v, mn) : v;
							const int32_t d = in ? abs((int32_t)t - (int32_t)o) : 0;
							const int32_t d2 = (int32_t)((uint32_t)d * (uint32_t)d);
							const uint32_t one = in ? 1u : 0u;
							if (v > background)
								fd += (uint32_t)d, f2 += d2, fn += one;
							else
								bd += (uint32_t)d, b2 += d2, bn += one;
						}
					}
				}
			}
		}
		const uint32_t wfd = lossy_wave_sum32(fd), wbd = lossy_wave_sum32(bd), wfn = lossy_wave_sum32(fn), wbn = lossy_wave_sum32(bn);
		const long long wf2 = lossy_wave_sum(f2), wb2 = lossy_wave_sum(b2);
		const int lane = tid & 63, wave = tid >> 6;
		if (lane == 0)
			red[wave][0] = (long long)wfd, red[wave][1] = wf2, red[wave][2] = (long long)wfn, red[wave][3] = (long long)wbd, red[wave][4] = wb2, red[wave][5] = (long long)wbn;
		__syncthreads();
		if (tid < 4)
		{ // words: fg count << 32 | fg sum d,  fg sum d2,  bg count << 32 | bg sum d,  bg sum d2
			const int c = tid == 0 ? 0 : tid == 1 ? 1 : tid == 2 ? 3 : 4;
			long long val = red[0][c] + red[1][c] + red[2][c] + red[3][c];
			if (tid == 0 || tid == 2)
				val |= (red[0][c + 2] + red[1][c + 2] + red[2][c + 2] + red[3][c + 2]) << 32;
			__hip_atomic_store(as_global(sp->rows) + ((size_t)k * nslabs + slab) * 4 + tid, (unsigned long long)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (!lossy_last_arriver(sp->tickets + k, (unsigned int)nslabs, &sh_last))
			return;
		// the frame's rows, a lane each (64 lanes x 4 words: up to 64 slabs a round), summed over the wave
		if (tid >= 64)
			return;
		long long tfd = 0, tfn = 0, tbd = 0, tbn = 0, tf2 = 0, tb2 = 0;
		RIR_GLOBAL(unsigned long long) *rows = as_global(sp->rows) + (size_t)k * nslabs * 4;
		for (int q = tid; q < nslabs; q += 64)
		{
			const unsigned long long w0 = __hip_atomic_load(rows + q * 4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w1 = __hip_atomic_load(rows + q * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned long long w2 = __hip_atomic_load(rows + q * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w3 = __hip_atomic_load(rows + q * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			tfd += (long long)(w0 & 0xffffffffull), tfn += (long long)(w0 >> 32), tf2 += (long long)w1;
			tbd += (long long)(w2 & 0xffffffffull), tbn += (long long)(w2 >> 32), tb2 += (long long)w3;
		}
		tfd = lossy_wave_sum(tfd), tfn = lossy_wave_sum(tfn), tf2 = lossy_wave_sum(tf2), tbd = lossy_wave_sum(tbd), tbn = lossy_wave_sum(tbn), tb2 = lossy_wave_sum(tb2);
		if (tid == 0)
		{ // stdDev (h264.cpp:1993-2036), as lossy_budget: unsplit while the window is not full
			RIR_GLOBAL(double) *gsd = as_global(sp->sd);
			const int n_win0 = as_global(r->budget)->n_win;
			if (n_win0 + k < 40)
			{
				const double sum_diff = (double)(tfd + tbd), sum_diff2 = (double)(tf2 + tb2);
				gsd[2 * k] = gsd[2 * k + 1] = sqrt(sum_diff * sum_diff - sum_diff2) / s;
			}
			else
			{
				const double dfd = (double)tfd, dfd2 = (double)tf2, dbd = (double)tbd, dbd2 = (double)tb2;
				gsd[2 * k] = sqrt(dbd * dbd - dbd2) / (int)tbn;
				gsd[2 * k + 1] = sqrt(dfd * dfd - dfd2) / (int)tfn;
			}
		}
	}
	// (a workgroup takes its slab of kLossySpecStatFrames frames, one after the other: a launch that has nothing to do - the other form's, a pass
	// that is not needed - is a quarter of the workgroups to start and to end: 2 us instead of 8 for a group of 1 000 frames)
	constexpr int kLossySpecStatFrames = 4;
	template <bool PLANE>
	__global__ __launch_bounds__(256) void lossy_spec_stats_kernel(const LossyRun *__restrict__ table, const LossySpec *__restrict__ spec)
	{
		__shared__ long long red[4][6];
		__shared__ unsigned int sh_last;
		const int slab = blockIdx.x, stream = blockIdx.z, nslabs = gridDim.x;
		RIR_GLOBAL(const LossySpec) *sp = as_global(spec + stream);
		if (__hip_atomic_load(as_global(sp->ctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
			return;
		{
			const bool from_plane = sp->dplane != nullptr && __hip_atomic_load(as_global(sp->ctl) + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
			if (from_plane != PLANE)
				return;
		}
		RIR_GLOBAL(const LossyRun) *r = as_global(table + stream);
		const int n = r->nsteps;
		for (int kk = 0; kk < kLossySpecStatFrames; ++kk)
		{
			const int k = (int)blockIdx.y * kLossySpecStatFrames + kk;
			if (k >= n)
				break;
			if (kk)
				__syncthreads(); // (`red` and `sh_last` are free again)
			lossy_spec_stats_frame<PLANE>(sp, r, k, slab, nslabs, red, &sh_last);
		}
	}

	// lossy_spec_verify_kernel: grid = streams, 1 024 threads, a thread per frame (two for groups of more than 1 024).  With the sums of
	// every frame known, the statistic of every frame is known, and the budget of frame k is the reference's arithmetic (lossy_budget, operation
	// for operation) over ITS window: the entries the stream had before the group, then the group's statistics up to k - a chain of at most
	// 41 additions of its own.  Everything is right up to and including the first frame whose budget differs from the table's.
	struct LossySpecWindow
	{
		double first[2];
		double old[40][2]; // the window before the group, oldest first
		int n_old, have_first;
	};
	__device__ __forceinline__ uint32_t lossy_spec_budget(const LossySpecWindow &w, const double (*sd)[2], int k, const LossyBudgetParams &bp)
	{
		const double first[2] = {w.have_first ? w.first[0] : sd[0][0], w.have_first ? w.first[1] : sd[0][1]};
		const int total = w.n_old + k + 1;			// statistics so far, this frame's included
		const int c = total < 40 ? total : 40;		// entries of the window once this frame's is in
		double mean[2] = {first[0], first[1]};
		for (int j = total - c; j < total; ++j)
		{ // oldest to newest, as the reference adds them up
			const double *e = j < w.n_old ? w.old[j] : sd[j - w.n_old];
			mean[0] += e[0], mean[1] += e[1];
		}
		mean[0] /= (double)(c + 1);
		mean[1] /= (double)(c + 1);
		int low_error = bp.low_value_error, high_error = bp.high_value_error;
		const double s0 = sd[k][0], s1 = sd[k][1];
		if (bp.add_loss)
		{
			const double dh = s1 < mean[1] ? 0 : s1 - mean[1], dl = s0 < mean[0] ? 0 : s0 - mean[0];
			high_error = sub_wrap(high_error, int_of_double_x86(round(dh * bp.std_factor)));
			low_error = sub_wrap(low_error, int_of_double_x86(round(dl * bp.std_factor)));
		}
		else
		{
			high_error = sub_wrap(high_error, int_of_double_x86(round(fabs(s1 - mean[1]) * bp.std_factor)));
			low_error = sub_wrap(low_error, int_of_double_x86(round(fabs(s0 - mean[0]) * bp.std_factor)));
		}
		if (high_error < 0)
			high_error = 0;
		if (low_error < high_error)
			low_error = high_error;
		return lossy_spec_pack(low_error, high_error);
	}
	__global__ __launch_bounds__(1024) void lossy_spec_verify_kernel(const LossyRun *__restrict__ table, const LossySpec *__restrict__ spec)
	{
		__shared__ double sd[kLossyConstMaxFrames][2];
		__shared__ LossySpecWindow w;
		__shared__ int sh_m, sh_off;
		const int tid = threadIdx.x, stream = blockIdx.x;
		RIR_GLOBAL(const LossySpec) *sp = as_global(spec + stream);
		RIR_GLOBAL(unsigned int) *ctl = as_global(sp->ctl);
		if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
			return;
		RIR_GLOBAL(const LossyRun) *r = as_global(table + stream);
		RIR_GLOBAL(const LossyBudget) *bud = as_global(r->budget);
		const int n = r->nsteps, s = r->s;
		const LossyBudgetParams bp = {s, r->add_loss, r->low_value_error, r->high_value_error, r->std_factor};
		const int n_win0 = bud->n_win, head0 = bud->head;
		if (tid < 80)
		{
			const int j = tid >> 1, c = tid & 1;
			if (j < n_win0)
				w.old[j][c] = bud->win[n_win0 < 40 ? j : (head0 + j) % 40][c];
		}
		if (tid == 0)
		{
			w.n_old = n_win0, w.have_first = bud->n_first >= 1 ? 1 : 0;
			w.first[0] = bud->first_std[0], w.first[1] = bud->first_std[1];
			sh_m = n, sh_off = 0;
		}
		// the statistics of the frames (lossy_spec_stats_kernel's last arrivers)
		RIR_GLOBAL(double) *gsd = as_global(sp->sd);
		for (int k = tid; k < n; k += 1024)
			sd[k][0] = gsd[2 * k], sd[k][1] = gsd[2 * k + 1];
		__syncthreads();
		RIR_GLOBAL(uint32_t) *tab = as_global(sp->budgets);
		uint32_t mine[2] = {0u, 0u};
		for (int k = tid, q = 0; k < n; k += 1024, ++q)
		{
			mine[q] = lossy_spec_budget(w, sd, k, bp);
			if (mine[q] != tab[k])
			{
				atomicMin(&sh_m, k);
				atomicAdd(&sh_off, 1);
			}
		}
		__syncthreads();
		const int m = sh_m;
		if (m == n)
		{ // every budget of the table is the reference's (the statistics stay where they are for the commit)
			if (tid == 0)
			{
				ctl[1] = ctl[1] + 1u, ctl[2] = (unsigned int)n;
				__hip_atomic_store(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			return;
		}
		// The sums of frame m are right - every frame before it was - and so is its budget; behind it the sums were taken against outputs that
		// are only roughly right, and so are the budgets computed from them: roughly.  ALL of them go into the table (ctl[6] == 0; 1: only frame
		// m's, the first form of this - kept for comparison): where the budgets follow the input more than they follow the outputs - a step
		// change whose 40-frame echo in the window mean moves every budget, a slow drift - the table then converges in a few passes instead of
		// one frame per pass; the verification is what it was (a table that equals what is computed from it is right frame by frame, by induction).
		const bool only_first = (ctl[6] & 1u) != 0u;
		for (int k = tid, q = 0; k < n; k += 1024, ++q)
			if (only_first ? k == m : k >= m)
				tab[k] = mine[q];
		if (tid == 0)
		{
			// Does it converge?  The number of frames off the table must at least halve from pass to pass (and a first pass with more than
			// half of the frames off is a scene whose budgets MOVE - S1: nearly every frame -: nothing to iterate on); otherwise the group
			// goes to the general form now, not after the passes that are left.
			const unsigned int passes = ctl[1] + 1u, off = (unsigned int)sh_off, before = ctl[5];
			ctl[1] = passes, ctl[2] = (unsigned int)m, ctl[5] = off;
			const bool hopeless = (ctl[6] & 2u) ? false : only_first ? off > 8u * (ctl[3] - passes) : (passes == 1u ? 2u * off > (unsigned int)n : 2u * off > before);
			if (passes >= ctl[3] || hopeless)
				__hip_atomic_store(ctl, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}

	__global__ void lossy_spec_skipped_kernel(unsigned int *__restrict__ bk_, unsigned int *__restrict__ bh_, unsigned int count)
	{
		RIR_GLOBAL(unsigned int) *bk = as_global(bk_);
		const unsigned int left = bk[0] > count ? bk[0] - count : 0u;
		bk[0] = left;
		if (bh_)
			__hip_atomic_store(as_global(bh_), left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}

	// lossy_spec_commit_kernel: grid = (workgroups, streams).  Every stream verified: the shadow state becomes the state (a copy: 14 bytes a
	// pixel and the ring's new images, against 6 bytes a pixel and FRAME of the passes), workgroup 0 of a stream files its window, its last
	// decision and its budgets, and the group's word says "done" to the resident launch behind.  Anything else: nothing is touched but the
	// leading stream's back-off.
	__global__ __launch_bounds__(256) void lossy_spec_commit_kernel(const LossyRun *__restrict__ table, const LossySpec *__restrict__ spec, int nstreams,
																	 unsigned int *__restrict__ ok_word)
	{
		__shared__ unsigned int sh_all, sh_offered;
		const int tid = threadIdx.x, b = blockIdx.x, stream = blockIdx.y, nb = gridDim.x;
		if (tid == 0)
			sh_all = 1u, sh_offered = 0u;
		__syncthreads();
		for (int q = tid; q < nstreams; q += 256)
		{
			RIR_GLOBAL(unsigned int) *c = as_global(as_global(spec + q)->ctl);
			if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
				atomicAnd(&sh_all, 0u);
			if (c[4] != 0u)
				atomicOr(&sh_offered, 1u);
		}
		__syncthreads();
		const bool all = sh_all != 0u, offered = sh_offered != 0u;
		if (b == 0 && stream == 0 && tid == 0)
		{
			RIR_GLOBAL(unsigned int) *bk = as_global(as_global(spec)->backoff);
			if (all)
				bk[1] = 0u;
			else if (offered)
			{ // (a group that was not offered - precondition, back-off - does not count)
				const unsigned int streak = bk[1] < 3u ? bk[1] + 1u : 3u; // 3, 15, 63 groups without an offer (a scene that moves goes on moving)
				bk[1] = streak, bk[0] = (1u << (2u * streak)) - 1u;
			}
			if (RIR_GLOBAL(unsigned int) *bh = as_global(as_global(spec)->backoff_host))
			{
				__hip_atomic_store(bh, bk[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(bh + 1, bk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
			__hip_atomic_store(as_global(ok_word), all ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (!all)
			return;
		RIR_GLOBAL(const LossyRun) *r = as_global(table + stream);
		RIR_GLOBAL(const LossySpec) *sp = as_global(spec + stream);
		const LossyDeviceState st = lossy_load_struct(&(table + stream)->st), sh = lossy_load_struct(&(spec + stream)->shadow);
		const int s = r->s, full = r->full, n = r->nsteps, ra = st.running_average;
		// 16 bytes a thread and round; s and full are multiples of 8 pixels
		auto copy16 = [&](void *dst_, const void *src_, size_t bytes) {
			RIR_GLOBAL(lossy_v4u) *dst = reinterpret_cast<RIR_GLOBAL(lossy_v4u) *>(as_global((char *)dst_));
			RIR_GLOBAL(const lossy_v4u) *src = reinterpret_cast<RIR_GLOBAL(const lossy_v4u) *>(as_global((const char *)src_));
			for (size_t i = (size_t)b * 256 + tid; i < bytes / 16; i += (size_t)nb * 256)
				dst[i] = src[i];
		};
		copy16(st.refT, sh.refT, (size_t)s * 2);
		copy16(st.prevT, sh.prevT, (size_t)s * 2);
		copy16(st.lastDL, sh.lastDL, (size_t)full * 2);
		if (ra > 0)
		{
			copy16(st.ra_sums, sh.ra_sums, (size_t)s * 4);
			copy16(st.ra_const_value, sh.ra_const_value, (size_t)s * 2);
			copy16(st.ra_const_count, sh.ra_const_count, (size_t)s * 2);
			// the ring's new images: the group's last min(ra, n) inputs, each in the slot after the one before (lossy_const_run_kernel's wr_slot)
			const int wr0 = (st.ra_head + (st.ra_count == ra ? 0 : st.ra_count)) % ra;
			const int nw = n < ra ? n : ra;
			for (int j = 0; j < nw; ++j)
			{
				const int slot = (wr0 + (n - nw) + j) % ra;
				copy16(st.ra_images + (size_t)slot * s, sh.ra_images + (size_t)slot * s, (size_t)s * 2);
			}
		}
		if (b != 0)
			return;
		// the budget state after the group's n frames: the window's last 40 statistics, the seed, the last decision; the budgets of the frames
		RIR_GLOBAL(LossyBudget) *bud = as_global(r->budget);
		RIR_GLOBAL(const double) *gsd = as_global(sp->sd);
		RIR_GLOBAL(const uint32_t) *tab = as_global(sp->budgets);
		// (every thread reads the counters before anybody moves them)
		const int n_first0 = bud->n_first, n_win0 = bud->n_win, head0 = bud->head;
		__syncthreads();
		// Frame k's statistic goes where lossy_budget would have put it: appended while the window fills, then round the ring from its head (a window
		// that is not full has its head at 0).  Only the group's last 40 frames are still there afterwards - a thread each, all to different places.
		if (tid < 40 && tid < n)
		{
			const int k = n - 1 - tid;
			const int pos = n_win0 + k < 40 ? n_win0 + k : (head0 + k - (40 - n_win0)) % 40;
			bud->win[pos][0] = gsd[2 * k], bud->win[pos][1] = gsd[2 * k + 1];
		}
		if (tid == 64)
		{
			if (n_first0 < 1)
			{
				bud->first_std[0] = gsd[0], bud->first_std[1] = gsd[1];
				bud->n_first = 1;
			}
			bud->n_win = n_win0 + n < 40 ? n_win0 + n : 40;
			bud->head = n_win0 + n <= 40 ? head0 : (head0 + n - (40 - n_win0)) % 40;
			RIR_GLOBAL(LossyDecision) *gd = as_global(r->decision);
			gd->background = (uint32_t)as_global(r->bg)[(size_t)(n - 1) * r->bg_stride];
			gd->low_error = (int)(tab[n - 1] & 0xffffu), gd->high_error = (int)(tab[n - 1] >> 16);
		}
		if (r->errors_out)
		{ // (the callers only offer streams whose configured errors are below 65 536: the table's fields ARE the budgets)
			RIR_GLOBAL(int) *e = as_global(r->errors_out);
			for (int k = tid; k < n; k += 256)
				e[2 * k] = (int)(tab[k] & 0xffffu), e[2 * k + 1] = (int)(tab[k] >> 16);
		}
	}

	// first frame: out = tmp minus the optional minimum on rows < lossy_height; seeds refT / prevT / lastDL
	__global__ __launch_bounds__(256) void lossy_first_kernel(const uint16_t *__restrict__ tmp, uint16_t *__restrict__ out, LossyDeviceState st, int s,
															   int full)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= full)
			return;
		const uint32_t v = tmp[i];
		st.lastDL[i] = (uint16_t)v;
		uint32_t t = v;
		if (i < s)
		{
			if (st.subtract_min)
				t = sub_min(v, st.min);
			st.refT[i] = (uint16_t)t;
			st.prevT[i] = (uint16_t)t;
		}
		out[i] = (uint16_t)t;
	}

	// minimum of the first s pixels (subtractMin option)
	__global__ __launch_bounds__(256) void lossy_min_kernel(const uint16_t *__restrict__ tmp, int s, unsigned int *__restrict__ result)
	{
		unsigned int m = 65535;
		for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < s; i += gridDim.x * blockDim.x)
			m = min(m, (unsigned int)tmp[i]);
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
			m = min(m, (unsigned int)__shfl_xor((int)m, d, 64));
		if ((threadIdx.x & 63) == 0)
			atomicMin(result, m);
	}

	// IRFileLoader::readImage (IRFileLoader.cpp:1173-1179): pixels[i] += min_T on the first min_T_height rows
	// (16-bit wrap-around like the reference's unsigned short +=), applied to every frame of a decoded chunk.
	__global__ void __launch_bounds__(256) lossy_add_min_kernel(uint16_t *__restrict__ frames, int64_t npx, int s, int nframes, uint32_t mn)
	{
		const int64_t total = (int64_t)s * nframes;
		for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
		{
			const int64_t f = i / s, p = i - f * s;
			frames[f * npx + p] = (uint16_t)(frames[f * npx + p] + mn);
		}
	}

	// One frame of `nstreams` streams: h_steps[i] describes stream i (device pointers; the ring indices as they are for this
	// frame).  One stream: the description travels as a kernel argument; several: d_table (device, nstreams entries, already
	// filled with h_steps by the caller) is indexed by blockIdx.y.  hist (16 384 bins), stats[1..7] and the two tickets of every
	// stream must be zero on entry: they are when the state is created, and every frame leaves them so.
	hipError_t launch_lossy_step(const LossyStep *h_steps, const LossyStep *d_table, int nstreams, hipStream_t st)
	{
		const LossyStep &one = h_steps[0];
		const int s = one.s, full = one.full; // (equal for all streams of a launch)
		int blocks = (s + 2047) / 2048; // sums: 8 pixels per thread
		if (blocks > 256)
			blocks = 256;
		const int hist_px = h_steps[0].hist_px;
		const dim3 gh((s + hist_px - 1) / hist_px, nstreams), gs(blocks, nstreams), gu((full + 255) / 256, nstreams);
		const dim3 gv((full / 8 + 255) / 256, nstreams);
		const bool vec = (s % 8 == 0) && (full % 8 == 0); // every group of 8 pixels wholly inside or wholly past the lossy rows
		if (nstreams > 1)
		{
			if (s > 0)
			{
				hipLaunchKernelGGL(lossy_hist_mode_kernel<true>, gh, dim3(1024), 0, st, one, d_table);
				hipLaunchKernelGGL(lossy_sums_budget_kernel<true>, gs, dim3(256), 0, st, one, d_table);
			}
			if (vec)
				hipLaunchKernelGGL(lossy_update_vec_kernel<true>, gv, dim3(256), 0, st, one, d_table);
			else
				hipLaunchKernelGGL(lossy_update_kernel<true>, gu, dim3(256), 0, st, one, d_table);
		}
		else
		{
			if (s > 0)
			{
				hipLaunchKernelGGL(lossy_hist_mode_kernel<false>, gh, dim3(1024), 0, st, one, nullptr);
				hipLaunchKernelGGL(lossy_sums_budget_kernel<false>, gs, dim3(256), 0, st, one, nullptr);
			}
			if (vec)
				hipLaunchKernelGGL(lossy_update_vec_kernel<false>, gv, dim3(256), 0, st, one, nullptr);
			else
				hipLaunchKernelGGL(lossy_update_kernel<false>, gu, dim3(256), 0, st, one, nullptr);
		}
		return hipGetLastError();
	}

	hipError_t launch_lossy_backgrounds(const LossyStep *d_table, int entries, int s, int hist_px, hipStream_t st)
	{
		if (entries <= 0 || s <= 0)
			return hipSuccess;
		LossyStep none{};
		hipLaunchKernelGGL(lossy_hist_mode_kernel<true>, dim3((s + hist_px - 1) / hist_px, entries), dim3(1024), 0, st, none, d_table);
		return hipGetLastError();
	}
	hipError_t launch_lossy_backgrounds_of_runs(const LossyRun *d_runs, int nstreams, int frames, uint32_t *d_hist, unsigned int *d_tickets, int s, int hist_px, hipStream_t st)
	{
		if (frames <= 0 || nstreams <= 0 || s <= 0)
			return hipSuccess;
		const dim3 grid((s + hist_px - 1) / hist_px, (unsigned)(frames * nstreams));
		if ((long long)frames * nstreams * s * 2 > (192ll << 20)) // (more than the 256 MB Infinity Cache keeps beside everything else: see lossy_hist_mode_body)
			hipLaunchKernelGGL(lossy_hist_mode_runs_kernel<true>, grid, dim3(1024), 0, st, d_runs, nstreams, d_hist, d_tickets, s, hist_px);
		else
			hipLaunchKernelGGL(lossy_hist_mode_runs_kernel<false>, grid, dim3(1024), 0, st, d_runs, nstreams, d_hist, d_tickets, s, hist_px);
		return hipGetLastError();
	}
	hipError_t launch_lossy_frame(const LossyStep *d_table, int nstreams, int full, hipStream_t st)
	{
		LossyStep none{};
		hipLaunchKernelGGL(lossy_frame_kernel<true>, dim3((full / 8 + 255) / 256, nstreams), dim3(256), 0, st, none, d_table);
		return hipGetLastError();
	}

	// workgroups of lossy_run_kernel the current device holds at once (runtime.h: occupancy x CUs, less the margin; 0 = unknown)
#ifndef RIR_LOSSY_RUN_MARGIN
#define RIR_LOSSY_RUN_MARGIN 1 /* 0: every place of the device - 8 streams of 640x512 per launch instead of 7, 617 k frames/s instead of 577 k; not shipped: a launch
                                  that fills the chip to the last place is called off whenever anything else is resident, and a loss run that is called off
                                  goes on frame by frame - or, for queue-only calls, fails the stream */
#endif
	int lossy_run_capacity(bool parked)
	{
		return resident_capacity(parked ? reinterpret_cast<const void *>(lossy_run_parked_kernel) : reinterpret_cast<const void *>(lossy_run_kernel), kLossyRunThreads,
								 0, RIR_LOSSY_RUN_MARGIN != 0);
	}
	hipError_t launch_lossy_run(const LossyRun *d_table, int nstreams, int full, unsigned int *d_ticket, unsigned int epoch, unsigned int arrivals_before, bool parked,
								hipStream_t st, const unsigned int *d_ok)
	{
		const int nb = lossy_run_workgroups(full);
		if ((long long)nb * nstreams > lossy_run_capacity(parked))
			return hipErrorInvalidConfiguration; // (the callers plan their launches with lossy_run_capacity(): never reached)
		ResidentGate gate(st); // its workgroups wait for each other: not beside any other resident launch of the process
		if (!gate.ok())
			return hipErrorUnknown;
		if (parked)
			hipLaunchKernelGGL(lossy_run_parked_kernel, dim3((unsigned)(nb * nstreams)), dim3(kLossyRunThreads), 0, st, d_table, d_ticket, nb, nstreams, epoch,
							   arrivals_before, d_ok);
		else
			hipLaunchKernelGGL(lossy_run_kernel, dim3((unsigned)(nb * nstreams)), dim3(kLossyRunThreads), 0, st, d_table, d_ticket, nb, nstreams, epoch, arrivals_before, d_ok);
		return hipGetLastError();
	}
	// Pairs of pixels per thread.  A frame costs a wave ~44 vector instructions per pair and ~10 whatever it holds, so beyond the point where
	// the SIMDs are busy - about five waves each - fewer, fatter waves win, and below it more waves do: the largest of 4, 2, 1 pairs that
	// still gives 5 000 waves (640x512, measured through the hook RIR_LOSSY_CONST_PAIRS with 1 / 2 / 4 pairs: two streams 1.98 / 1.89 / 1.77 M
	// frames/s, nine 1.52 / 1.59 / 1.70, thirty-two 1.43 / 1.61 / 1.81; one stream is 2 560 waves with one pair and has no choice).
	static std::atomic<int> g_const_pairs_forced{0}; // (tests and measurements, through the build with the test hooks: lossy_const_force_pairs; calls of several threads all store 0)
	void lossy_const_force_pairs(int np) { g_const_pairs_forced.store((np == 4 || np == 2 || np == 1) ? np : 0, std::memory_order_relaxed); }
	int lossy_const_pairs(int full, int nstreams)
	{
		if (const int forced = g_const_pairs_forced.load(std::memory_order_relaxed))
			return forced;
		for (int np = 4; np > 1; np >>= 1)
			if ((long long)full / (2 * np) / 64 * nstreams >= 5000)
				return np;
		return 1;
	}
	int lossy_const_workgroups(int full, int nstreams) { return (full / (2 * lossy_const_pairs(full, nstreams)) + 255) / 256; }
	hipError_t launch_lossy_const(const LossyRun *d_table, int nstreams, int full, bool any_ra, bool add_loss, unsigned int *d_ok, const unsigned int *d_poison, hipStream_t st)
	{
		const int np = lossy_const_pairs(full, nstreams), nb = lossy_const_workgroups(full, nstreams);
		const dim3 grid((unsigned)nb, (unsigned)nstreams);
		// (the variant - a running average or none, addLoss or not - is the launch's: streams of a call share add_loss, and a call whose streams
		// differ in having a running average goes through the instantiation with one, which reads each stream's own length)
#define RIR_CONST_LAUNCH(NPV)                                                                                                                  \
	{                                                                                                                                          \
		if (any_ra && add_loss)                                                                                                                \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, true, true>), grid, dim3(256), 0, st, d_table, nstreams, d_ok, d_poison, (const LossySpec *)nullptr);          \
		else if (any_ra)                                                                                                                       \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, true, false>), grid, dim3(256), 0, st, d_table, nstreams, d_ok, d_poison, (const LossySpec *)nullptr);         \
		else if (add_loss)                                                                                                                     \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, false, true>), grid, dim3(256), 0, st, d_table, nstreams, d_ok, d_poison, (const LossySpec *)nullptr);         \
		else                                                                                                                                   \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, false, false>), grid, dim3(256), 0, st, d_table, nstreams, d_ok, d_poison, (const LossySpec *)nullptr);        \
	}
		if (np == 4)
			RIR_CONST_LAUNCH(4)
		else if (np == 2)
			RIR_CONST_LAUNCH(2)
		else
			RIR_CONST_LAUNCH(1)
#undef RIR_CONST_LAUNCH
		hipLaunchKernelGGL(lossy_const_finish_kernel, dim3((unsigned)nstreams), dim3(1024), 0, st, d_table, nb, (const unsigned int *)d_ok);
		return hipGetLastError();
	}

	hipError_t launch_lossy_spec_begin(const LossyRun *d_table, const LossySpec *d_spec, int nstreams, int passes, unsigned int *d_ok, const unsigned int *d_poison, hipStream_t st)
	{
		hipLaunchKernelGGL(lossy_spec_begin_kernel, dim3(1), dim3(1024), 0, st, d_table, d_spec, nstreams, passes, d_ok, d_poison);
		return hipGetLastError();
	}
	hipError_t launch_lossy_spec_pass(const LossyRun *d_table, const LossySpec *d_spec, int nstreams, int s, int full, int max_frames, bool any_ra, bool add_loss, hipStream_t st)
	{
		const int np = lossy_const_pairs(full, nstreams), nb = lossy_const_workgroups(full, nstreams);
		const dim3 grid((unsigned)nb, (unsigned)nstreams);
#define RIR_SPEC_LAUNCH(NPV)                                                                                                                                      \
	{                                                                                                                                                             \
		if (any_ra && add_loss)                                                                                                                                   \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, true, true, true>), grid, dim3(256), 0, st, d_table, nstreams, (unsigned int *)nullptr, (const unsigned int *)nullptr, d_spec);   \
		else if (any_ra)                                                                                                                                          \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, true, false, true>), grid, dim3(256), 0, st, d_table, nstreams, (unsigned int *)nullptr, (const unsigned int *)nullptr, d_spec);  \
		else if (add_loss)                                                                                                                                        \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, false, true, true>), grid, dim3(256), 0, st, d_table, nstreams, (unsigned int *)nullptr, (const unsigned int *)nullptr, d_spec);  \
		else                                                                                                                                                      \
			hipLaunchKernelGGL((lossy_const_run_kernel<NPV, false, false, true>), grid, dim3(256), 0, st, d_table, nstreams, (unsigned int *)nullptr, (const unsigned int *)nullptr, d_spec); \
	}
		if (np == 4)
			RIR_SPEC_LAUNCH(4)
		else if (np == 2)
			RIR_SPEC_LAUNCH(2)
		else
			RIR_SPEC_LAUNCH(1)
#undef RIR_SPEC_LAUNCH
		const int nslabs = lossy_spec_stat_workgroups(s);
		// (the sums from the byte plane; from the frames where there is no plane or it does not hold the pass's differences: one of the two returns at once)
		hipLaunchKernelGGL(lossy_spec_stats_kernel<true>, dim3((unsigned)nslabs, (unsigned)((max_frames + kLossySpecStatFrames - 1) / kLossySpecStatFrames), (unsigned)nstreams), dim3(256), 0, st, d_table, d_spec);
		hipLaunchKernelGGL(lossy_spec_stats_kernel<false>, dim3((unsigned)nslabs, (unsigned)((max_frames + kLossySpecStatFrames - 1) / kLossySpecStatFrames), (unsigned)nstreams), dim3(256), 0, st, d_table, d_spec);
		hipLaunchKernelGGL(lossy_spec_verify_kernel, dim3((unsigned)nstreams), dim3(1024), 0, st, d_table, d_spec);
		return hipGetLastError();
	}
	hipError_t launch_lossy_spec_skipped(unsigned int *d_backoff, unsigned int *backoff_host, unsigned int count, hipStream_t st)
	{
		hipLaunchKernelGGL(lossy_spec_skipped_kernel, dim3(1), dim3(1), 0, st, d_backoff, backoff_host, count);
		return hipGetLastError();
	}
	hipError_t launch_lossy_spec_commit(const LossyRun *d_table, const LossySpec *d_spec, int nstreams, int s, int full, unsigned int *d_ok, hipStream_t st)
	{
		(void)s;
		int nb = (full / 8 + 255) / 256; // 16 bytes a thread and round
		if (nb > 1024)
			nb = 1024;
		hipLaunchKernelGGL(lossy_spec_commit_kernel, dim3((unsigned)nb, (unsigned)nstreams), dim3(256), 0, st, d_table, d_spec, nstreams, d_ok);
		return hipGetLastError();
	}

	hipError_t launch_lossy_first(const uint16_t *d_tmp, uint16_t *d_out, const LossyDeviceState &state, int s, int full, hipStream_t st)
	{
		hipLaunchKernelGGL(lossy_first_kernel, dim3((full + 255) / 256), dim3(256), 0, st, d_tmp, d_out, state, s, full);
		return hipGetLastError();
	}

	hipError_t launch_lossy_min(const uint16_t *d_tmp, int s, unsigned int *d_result, hipStream_t st)
	{
		const unsigned int init = 65535;
		hipError_t e = hipMemcpyAsync(d_result, &init, sizeof(init), hipMemcpyHostToDevice, st);
		if (e != hipSuccess)
			return e;
		int blocks = (s + 255) / 256;
		if (blocks > 512)
			blocks = 512;
		hipLaunchKernelGGL(lossy_min_kernel, dim3(blocks), dim3(256), 0, st, d_tmp, s, d_result);
		return hipGetLastError();
	}
	hipError_t launch_lossy_add_min(uint16_t *d_frames, int64_t npx, int s, int nframes, uint32_t mn, hipStream_t st)
	{
		if (s <= 0 || nframes <= 0)
			return hipSuccess;
		int64_t blocks = ((int64_t)s * nframes + 255) / 256;
		if (blocks > 4096)
			blocks = 4096;
		hipLaunchKernelGGL(lossy_add_min_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_frames, npx, s, nframes, mn);
		return hipGetLastError();
	}
} // namespace rir
