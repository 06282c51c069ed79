// Host-side copy of a frame between the caller's memory and page-locked staging - and a chunk's trip to / from the file - spread over
// a few helper threads.
//
// Why: on the per-frame C ABI (h264_add_image_lossless, load_image, the signal_processing entry points) the caller owns its buffer
// again when the call returns, so every frame crosses the host once more than the PCIe link asks for - user memory <-> a page-locked
// slot.  Measured on the GPU box (tests/perf/abi_breakdown.py, 640x512 uint16): that one memcpy is 23 us of a 27 us add_image call and
// 15-20 us of a 21 us load_image call, while the link needs 13 us per frame: the calls are bound by ONE core's copy rate.  Four cores
// copy a frame in a quarter of the time; the link then is the limit, as it should be.
//
// How: helpers are assigned explicitly.  A helper is IDLE (spinning on its own state word), BUSY (owned by exactly one caller) or
// PARKED (asleep on its condition variable after kSpinNs without work).  A caller takes the helpers it finds IDLE (one CAS each),
// cuts the copy into that many parts + its own, and waits for the parts it handed out; a PARKED helper is woken for the NEXT call
// and this call does without it.  Nothing is shared between two callers, so several savers / loaders driven from different threads
// simply compete for the idle helpers.  During a burst of per-frame calls (a recording, a sequential read) the helpers stay hot;
// a few hundred microseconds after the last call they sleep and cost nothing.  The helpers of the file group sleep between jobs and are
// woken with their part (a chunk's read or write is long against a wake-up).
// The same helpers move a chunk between page-locked memory and the FILE (host_pread / host_pwrite: disjoint ranges of one descriptor):
// a 7 MB chunk through the page cache is 0.8-1.4 ms on one thread - longer than the chunk's 0.65 ms on the PCIe link.
// RIR_HOST_COPY_THREADS / RIR_HOST_IO_THREADS = helpers for memory copies / for file jobs (default 3 each, 0 = on the calling thread, at
// most 7).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include <errno.h>
#include <pthread.h>
#include <unistd.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "runtime.h"

namespace rir
{
	namespace
	{
		constexpr int kMaxHelpers = 7;
		constexpr size_t kParallelFrom = 192 * 1024; // below this a single memcpy is faster than a hand-off
		constexpr int64_t kSpinNs = 300 * 1000;		 // an idle helper spins this long before it parks

		enum : int
		{
			IDLE = 0,
			BUSY = 1,
			PARKED = 2
		};

		enum : int
		{
			JOB_COPY = 0,
			JOB_PREAD = 1,
			JOB_PWRITE = 2
		};
		// the whole range or failure (short reads / writes are continued, EINTR repeated)
		bool pread_all(int fd, void *dst, size_t bytes, int64_t off)
		{
			char *p = static_cast<char *>(dst);
			while (bytes)
			{
				const ssize_t r = ::pread(fd, p, bytes, (off_t)off);
				if (r < 0 && errno == EINTR)
					continue;
				if (r <= 0)
					return false; // error, or the file ends inside the range
				p += r, off += r, bytes -= (size_t)r;
			}
			return true;
		}
		bool pwrite_all(int fd, const void *src, size_t bytes, int64_t off)
		{
			const char *p = static_cast<const char *>(src);
			while (bytes)
			{
				const ssize_t r = ::pwrite(fd, p, bytes, (off_t)off);
				if (r < 0 && errno == EINTR)
					continue;
				if (r <= 0)
					return false;
				p += r, off += r, bytes -= (size_t)r;
			}
			return true;
		}
		bool do_job(int kind, void *dst, const void *src, size_t bytes, int fd, int64_t off)
		{
			if (kind == JOB_PREAD)
				return pread_all(fd, dst, bytes, off);
			if (kind == JOB_PWRITE)
				return pwrite_all(fd, src, bytes, off);
			std::memcpy(dst, src, bytes);
			return true;
		}

		inline void cpu_relax()
		{
#if defined(__x86_64__)
			_mm_pause();
#else
			std::this_thread::yield();
#endif
		}
		inline int64_t now_ns()
		{
			return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
		}

		struct alignas(128) Helper
		{
			std::atomic<int> state{PARKED}; // a helper starts parked: the first call wakes it
			std::atomic<uint64_t> go{0};	 // written by the owner after the job below
			std::atomic<uint64_t> done{0};	 // written by the helper: the last job it finished
			int kind = 0; // JOB_*
			void *dst = nullptr;
			const void *src = nullptr;
			size_t bytes = 0;
			int fd = -1;
			int64_t file_off = 0;
			std::atomic<int> *failed = nullptr; // the owner's flag, raised when a file job fails (written before `done`)
			std::mutex m;
			std::condition_variable cv;
			bool wake = false;
			int64_t spin_ns = kSpinNs; // how long it spins idle before it parks (0: at once - the file group)
			std::thread th;
		};

		std::atomic<bool> g_forked{false}; // set in the child of a fork(): the helper threads do not exist there

		// Two groups of helpers: memory copies (the calling thread of a per-frame entry point waits for them: 10 us jobs, every 10-20 us
		// during a recording or a read) and file jobs (the saver's writer thread, the loader's read-ahead lanes: a few hundred
		// microseconds each).  With one group a file job held the helpers while the caller's copies ran alone at a third of the rate.
		struct Pool
		{
			Helper h[2][kMaxHelpers];
			int n[2] = {0, 0};
			// CPUs the cgroup of this process may use at once (cgroup v2 cpu.max "quota period", v1 cfs_quota_us / cfs_period_us), rounded up;
			// 0: no limit, or not to be found
			static unsigned cgroup_cpus()
			{
				long long q = -1, per = 0;
				if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r"))
				{
					char w[32] = {0};
					if (std::fscanf(f, "%31s %lld", w, &per) == 2 && std::strcmp(w, "max") != 0)
						q = std::atoll(w);
					std::fclose(f);
				}
				else
				{
					FILE *a = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *b = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
					if (a && b && (std::fscanf(a, "%lld", &q) != 1 || std::fscanf(b, "%lld", &per) != 1))
						q = -1;
					if (a)
						std::fclose(a);
					if (b)
						std::fclose(b);
				}
				if (q <= 0 || per <= 0)
					return 0;
				return (unsigned)((q + per - 1) / per);
			}
			static int wanted(const char *var, int dflt, unsigned hw)
			{
				int want = dflt;
				if (const char *e = std::getenv(var))
					want = std::atoi(e);
				if (want < 0)
					want = 0; // (before the unsigned comparison below: a negative value is no helper, not all of them - ADVICE r5)
				if (hw && (unsigned)want > hw - 1)
					want = (int)hw - 1; // never more threads than other cores
				return want > kMaxHelpers ? kMaxHelpers : want;
			}
			Pool()
			{
				pthread_atfork(nullptr, nullptr, [] { g_forked.store(true, std::memory_order_relaxed); });
				unsigned hw = std::thread::hardware_concurrency(); // (respects the affinity mask)
				const unsigned quota = cgroup_cpus();
				if (quota && (!hw || quota < hw))
					hw = quota; // a container's CPU quota: helpers that spin beyond it only take the caller's time slice
				n[0] = wanted("RIR_HOST_COPY_THREADS", 3, hw);
				n[1] = wanted("RIR_HOST_IO_THREADS", 3, hw > 4 ? hw - 3 : 1);
				for (int g = 0; g < 2; ++g)
					for (int i = 0; i < n[g]; ++i)
					{
						// a copy job comes every 10-20 us and takes 8: its helpers spin between jobs.  A file job comes once a chunk and takes
						// hundreds of microseconds: its helpers sleep in between and are woken WITH their job (50 us of wake-up do not matter)
						h[g][i].spin_ns = g == 0 ? kSpinNs : 0;
						h[g][i].th = std::thread([this, g, i] { run(h[g][i]); });
						h[g][i].th.detach(); // the pool lives as long as the process (it is leaked on purpose: no destructor order to get wrong)
					}
			}
			static void run(Helper &me)
			{
				uint64_t seq = 0;
				int64_t idle_since = now_ns();
				for (;;)
				{
					int s = me.state.load(std::memory_order_acquire);
					if (s == BUSY)
					{ // a caller owns this helper: its job follows at once
						while (me.go.load(std::memory_order_acquire) == seq)
							cpu_relax();
						++seq;
						if (!do_job(me.kind, me.dst, me.src, me.bytes, me.fd, me.file_off))
							me.failed->store(1, std::memory_order_relaxed);
						me.done.store(seq, std::memory_order_release);
						me.state.store(IDLE, std::memory_order_release);
						idle_since = now_ns();
						continue;
					}
					if (s == IDLE && now_ns() - idle_since >= me.spin_ns)
					{
						int expect = IDLE;
						if (me.state.compare_exchange_strong(expect, PARKED, std::memory_order_acq_rel))
							s = PARKED;
					}
					if (s == PARKED)
					{
						{
							std::unique_lock<std::mutex> lk(me.m);
							me.cv.wait(lk, [&] { return me.wake; });
							me.wake = false;
						}
						// woken for the next call (still PARKED: spin now), or woken WITH a job (an owner made it BUSY while it slept)
						int expect = PARKED;
						(void)me.state.compare_exchange_strong(expect, IDLE, std::memory_order_acq_rel);
						idle_since = now_ns();
						continue;
					}
					cpu_relax();
				}
			}
		};

		Pool *pool()
		{
			static Pool *p = new Pool(); // (leaked: see Pool())
			return p;
		}
	} // namespace

	int host_copy_threads() { return g_forked.load(std::memory_order_relaxed) ? 0 : pool()->n[0]; }

	namespace
	{
		// one job cut into parts over the idle helpers + the calling thread; false when a part failed (file jobs)
		bool run_parts(int kind, void *dst, const void *src, size_t bytes, int fd, int64_t file_off)
		{
			if (bytes < kParallelFrom || g_forked.load(std::memory_order_relaxed))
				return do_job(kind, dst, src, bytes, fd, file_off);
			Pool *p = pool();
			Helper *mine[kMaxHelpers];
			int k = 0;
			const int g = kind == JOB_COPY ? 0 : 1;
			bool asleep[kMaxHelpers] = {};
			auto wake = [](Helper &h) {
				std::lock_guard<std::mutex> lk(h.m);
				if (!h.wake)
				{
					h.wake = true;
					h.cv.notify_one();
				}
			};
			for (int i = 0; i < p->n[g]; ++i)
			{
				Helper &h = p->h[g][i];
				int expect = IDLE;
				if (h.state.compare_exchange_strong(expect, BUSY, std::memory_order_acq_rel))
					mine[k++] = &h;
				else if (expect == PARKED)
				{
					if (g == 0)
						wake(h); // a copy: woken for the next call, this one does without it
					else if (h.state.compare_exchange_strong(expect, BUSY, std::memory_order_acq_rel))
						asleep[k] = true, mine[k++] = &h; // a file job: taken asleep, woken below with its part
				}
			}
			// k + 1 parts, cut at multiples of 4 KiB; the caller takes the last one
			const size_t part = ((bytes / (size_t)(k + 1)) + 4095) & ~(size_t)4095;
			uint64_t want[kMaxHelpers];
			std::atomic<int> failed{0}; // (the helpers write it before they report `done`, and this function waits for every `done`)
			size_t off = 0;
			for (int i = 0; i < k; ++i)
			{
				Helper &h = *mine[i];
				const size_t n = off + part <= bytes ? part : bytes - off;
				h.kind = kind, h.fd = fd, h.file_off = file_off + (int64_t)off, h.failed = &failed;
				h.dst = dst ? static_cast<char *>(dst) + off : nullptr, h.src = src ? static_cast<const char *>(src) + off : nullptr, h.bytes = n;
				want[i] = h.go.load(std::memory_order_relaxed) + 1;
				h.go.store(want[i], std::memory_order_release);
				if (asleep[i])
					wake(h);
				off += n;
			}
			bool ok = true;
			if (off < bytes)
				ok = do_job(kind, dst ? static_cast<char *>(dst) + off : nullptr, src ? static_cast<const char *>(src) + off : nullptr, bytes - off, fd,
							file_off + (int64_t)off);
			// (`done` only grows: by the time this caller looks, the helper may have finished a LATER job of another caller already)
			for (int i = 0; i < k; ++i)
				for (unsigned spins = 0; mine[i]->done.load(std::memory_order_acquire) < want[i]; ++spins)
				{
					// a file job: the writer thread / a read-ahead lane waits, for as long as the file system takes.  A copy job takes 8 us - a
					// few thousand spins; a caller still waiting after 20 000 has a helper that was descheduled (a container with a CPU quota):
					// it gives its time slice up instead of spinning through it
					if (spins > (g == 1 ? 2000u : 20000u))
						std::this_thread::yield();
					else
						cpu_relax();
				}
			return ok && failed.load(std::memory_order_relaxed) == 0;
		}
	} // namespace

	void host_copy(void *dst, const void *src, size_t bytes) { (void)run_parts(JOB_COPY, dst, src, bytes, -1, 0); }
	bool host_pread(int fd, void *dst, size_t bytes, int64_t file_off) { return run_parts(JOB_PREAD, dst, nullptr, bytes, fd, file_off); }
	bool host_pwrite(int fd, const void *src, size_t bytes, int64_t file_off) { return run_parts(JOB_PWRITE, nullptr, src, bytes, fd, file_off); }
} // namespace rir

// Test / measurement entry (include/rir_amd_device.h): the copy the per-frame entry points use, on plain host memory.  No device involved.
RIR_EXPORT int rir_host_copy(void *dst, const void *src, int64_t bytes)
{
	if (bytes < 0 || (bytes > 0 && (!dst || !src)))
		return -1;
	rir::host_copy(dst, src, (size_t)bytes);
	return rir::host_copy_threads();
}
// The chunk I/O of the saver and the loader on a caller's descriptor: `write` != 0 writes buf to [file_off, file_off + bytes), else reads that range
// (the whole range or -1: a file that ends inside it is a failure).  0 on success.
RIR_EXPORT int rir_host_file_rw(int fd, void *buf, int64_t bytes, int64_t file_off, int write)
{
	if (fd < 0 || bytes < 0 || file_off < 0 || (bytes > 0 && !buf))
		return -1;
	const bool ok = write ? rir::host_pwrite(fd, buf, (size_t)bytes, file_off) : rir::host_pread(fd, buf, (size_t)bytes, file_off);
	return ok ? 0 : -1;
}
// First touch of fresh memory, ahead of whoever fills it: one atomic compare-and-swap of a byte with itself per 4 KiB page of [buf, buf + bytes), on the calling thread.
// A stack that a slice of a movie is read into is fresh memory: its pages are made (zeroed, 2 MiB at a time where the allocator asked for
// huge pages) under the threads that copy the images in - 25-35 us per 640x512 image on top of a 17 us read.  Called from a thread of its own
// while the images are being read, the pages are there before the copies arrive (profiles/r05_movie_bulk.txt).  It is an atomic
// read-modify-write that changes nothing: a byte the reader has stored already stays what the reader stored, whichever comes first.
// (MADV_POPULATE_WRITE over the helper threads was measured too: 23-30 ms per 655 MB against 15 ms for this loop - not kept.)
RIR_EXPORT int rir_host_touch(void *buf, int64_t bytes)
{
	if (bytes < 0 || (bytes > 0 && !buf))
		return -1;
	// (a compare-and-swap of a byte with itself: an `or 0` / `add 0` is folded into a plain load by the compiler, and a load maps the shared zero
	// page instead of making one)
	auto touch = [](char *q) {
		char seen = __atomic_load_n(q, __ATOMIC_RELAXED);
		(void)__atomic_compare_exchange_n(q, &seen, seen, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
	};
	char *p = static_cast<char *>(buf);
	for (int64_t o = 0; o < bytes; o += 4096)
		touch(p + o);
	if (bytes > 0)
		touch(p + bytes - 1);
	return 0;
}
