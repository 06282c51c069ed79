// Start-up rendezvous of a resident launch (device side of runtime.h's "resident launches").
//
// A kernel whose workgroups wait for each other needs ALL of them on the chip.  The host sizes such a launch from the
// device's occupancy figures and lets no two of them run side by side - but ORDINARY kernels of other streams may run
// beside it, and they leave the register files and the LDS of a CU fragmented: a workgroup of the resident kernel that
// needs, say, 168 contiguous VGPRs on every SIMD then does not fit although more than that is free, the launch never
// becomes fully resident, and the workgroups that did start wait until their clocks run out (measured: an 8-sequence
// alignment launch - 3 workgroups per CU, 504 of a SIMD's 512 VGPRs - beside a flood of gaussian_filter calls from
// another thread ran into its 2 s clock in 5 calls of 8; tests/perf/resident_vs_ordinary_probe.py).
//
// So a resident kernel first finds out whether it IS resident, before it touches anything: every workgroup adds itself
// to a counter and then waits, for a SHORT time, for one common decision - GO when the counter shows that everybody has
// arrived, BAIL when somebody's clock has run out first (one compare-and-swap decides; every workgroup follows the
// decided value, including those that start after the others have left).  A launch that bails out has written nothing:
// the host sees the decision and runs the same work again with a smaller footprint (fewer workgroups per unit, or the
// launch-per-iteration / launch-per-frame kernels) - same results, a detour of milliseconds instead of an error after
// two seconds.
//
// ctl[0]: arrivals, monotone over the launches that use this control block (the host passes the count expected before
// this launch); ctl[1]: decision word, (epoch << 2) | code.  Zeroed once, when the block is allocated.
// Launches that build on each other's results (the groups of frames of a bounded-loss run, queued back to back without a
// host round trip) pass `chained`: then ctl[2] is a POISON word - raised by the launch that bails out, it makes every later
// launch on the block bail out at once, so that nothing is computed on top of work that was not done - and ctl[3] keeps
// the epoch of the first launch that bailed out; the host clears both when it has dealt with it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rir
{
	enum
	{
		RESIDENT_GO = 1,
		RESIDENT_BAIL = 2
	};
	constexpr unsigned long long kResidentRendezvousTicks = 500000ull; // 5 ms of the 100 MHz clock: a launch that fits is complete within tens of microseconds

	// Called by every thread of every workgroup, first thing.  `flag`: one LDS word.  Returns RESIDENT_GO or RESIDENT_BAIL, the
	// same value in every thread of every workgroup of the launch.
	__device__ __forceinline__ int resident_rendezvous(unsigned int *ctl, unsigned int arrivals_before, unsigned int total, unsigned int epoch, unsigned int *flag,
													   bool chained = false)
	{
		if (threadIdx.x == 0)
		{
			const unsigned int tag = (epoch & 0x3fffffffu) << 2;
			__hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
			unsigned int code = 0;
			for (;;)
			{
				const unsigned int d = __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((d & ~3u) == tag && (d & 3u) != 0)
				{
					code = d & 3u;
					break;
				}
				unsigned int want = 0;
				if (chained && __hip_atomic_load(ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
					want = RESIDENT_BAIL; // an earlier launch of the chain was not done
				else if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - arrivals_before >= total)
					want = RESIDENT_GO;
				else if (__builtin_amdgcn_s_memrealtime() - t0 > kResidentRendezvousTicks)
					want = RESIDENT_BAIL;
				if (want)
				{ // the first proposal wins; everybody (the proposer too) then reads what was decided
					unsigned int expected = d;
					__hip_atomic_compare_exchange_strong(ctl + 1, &expected, tag | want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					continue;
				}
				__builtin_amdgcn_s_sleep(2);
			}
			if (chained && code == RESIDENT_BAIL)
			{ // (idempotent: every workgroup of a launch that bails out writes the same things)
				unsigned int zero = 0;
				__hip_atomic_compare_exchange_strong(ctl + 3, &zero, epoch, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			*flag = code;
		}
		__syncthreads();
		const int code = (int)*flag;
		__syncthreads();
		return code;
	}
} // namespace rir
