// Log, handle registry and HIP device context (see runtime.h).  Also exports the matching part of
// the reference `tools` C ABI: set_print_function / disable_print / reset_print_functions /
// get_last_log_error (tools.h:39-55) and set_void_ptr / get_void_ptr / rm_void_ptr (tools.h:67-78).
#include "runtime.h"

#include <atomic>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#ifndef RIR_SPIN_WAIT
#define RIR_SPIN_WAIT 1
#endif
namespace rir
{
	namespace
	{
		std::mutex &log_mutex()
		{
			static std::mutex m;
			return m;
		}
		struct LogState
		{
			print_function fn = nullptr;
			bool enabled = true;
			std::string last_error;
		};
		LogState &log_state()
		{
			static LogState s;
			return s;
		}
	} // namespace

	void log_message(int level, const char *text)
	{
		std::lock_guard<std::mutex> g(log_mutex());
		LogState &s = log_state();
		if (level == LOG_ERROR)
			s.last_error = text ? text : "";
		if (!s.enabled)
			return;
		if (s.fn)
			s.fn(level, text);
		else
		{
			static const char *prefix[3] = {"Info: ", "Warning: ", "Error: "};
			std::printf("%s%s\n", prefix[level < 0 || level > 2 ? 0 : level], text ? text : "");
			std::fflush(stdout);
		}
	}

	// ---- registry ----------------------------------------------------------------------------
	namespace
	{
		std::mutex &reg_mutex()
		{
			static std::mutex m;
			return m;
		}
		std::map<int, std::shared_ptr<Object>> &reg_map()
		{
			static std::map<int, std::shared_ptr<Object>> m;
			return m;
		}
	} // namespace

	int register_object(const std::shared_ptr<Object> &obj)
	{
		if (!obj)
			return -1;
		std::lock_guard<std::mutex> g(reg_mutex());
		auto &m = reg_map();
		int i = 1; // smallest free slot, starting at 1 (tools.cpp:57-67)
		for (auto it = m.begin(); it != m.end() && it->first == i; ++it)
			++i;
		m[i] = obj;
		return i;
	}
	std::shared_ptr<Object> lookup_object(int handle)
	{
		std::lock_guard<std::mutex> g(reg_mutex());
		auto it = reg_map().find(handle);
		return it == reg_map().end() ? std::shared_ptr<Object>() : it->second;
	}
	void remove_object(int handle)
	{
		std::shared_ptr<Object> keep; // destroy outside the lock
		{
			std::lock_guard<std::mutex> g(reg_mutex());
			auto it = reg_map().find(handle);
			if (it != reg_map().end())
			{
				keep = it->second;
				reg_map().erase(it);
			}
		}
	}

	// ---- device --------------------------------------------------------------------------------
	bool hip_ok(hipError_t e, const char *what)
	{
		if (e == hipSuccess)
			return true;
		log_error(std::string("librir_amd: ") + what + ": " + hipGetErrorString(e));
		return false;
	}

	// Waits for a stream by polling it: the per-frame entry points wait for a few tens of microseconds of work, and a
	// blocking wait adds its wake-up latency (20-50 us here) to every call.  Falls back to the blocking wait after 2 ms.
	hipError_t wait_stream(hipStream_t st)
	{
#if RIR_SPIN_WAIT
		const auto t0 = std::chrono::steady_clock::now();
		for (long spins = 1;; ++spins)
		{
			const hipError_t e = hipStreamQuery(st);
			if (e != hipErrorNotReady)
				return e;
			if ((spins & 0x3f) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
				break;
			__builtin_ia32_pause();
		}
#endif
		return hipStreamSynchronize(st);
	}

	hipError_t wait_event(hipEvent_t ev)
	{
#if RIR_SPIN_WAIT
		const auto t0 = std::chrono::steady_clock::now();
		for (long spins = 1;; ++spins)
		{
			const hipError_t e = hipEventQuery(ev);
			if (e != hipErrorNotReady)
				return e;
			if ((spins & 0x3f) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
				break;
			__builtin_ia32_pause();
		}
#endif
		return hipEventSynchronize(ev);
	}

	namespace
	{
		struct Device
		{
			bool probed = false, ok = false;
			hipStream_t stream = nullptr;
		};
		Device &device()
		{
			static Device d;
			return d;
		}
		std::mutex &dev_mutex()
		{
			static std::mutex m;
			return m;
		}
	} // namespace

	bool device_ready()
	{
		std::lock_guard<std::mutex> g(dev_mutex());
		Device &d = device();
		if (!d.probed)
		{
			d.probed = true;
			int n = 0;
			hipError_t e = hipGetDeviceCount(&n);
			if (e == hipSuccess && n > 0)
			{
				e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
				d.ok = (e == hipSuccess);
			}
			if (!d.ok)
				(void)hipGetLastError();
		}
		if (!d.ok)
			log_message(LOG_ERROR, "librir_amd: no usable HIP device (this library has no CPU fallback)");
		return d.ok;
	}
	hipStream_t default_stream() { return device().stream; }

	// ---- resident launches (runtime.h) ---------------------------------------------------------------
	int resident_capacity_rule(int blocks_per_cu, int cus, int xcds)
	{
		if (blocks_per_cu <= 0 || cus <= 0)
			return 0;
		if (xcds <= 0 || cus % xcds != 0)
			xcds = 1;
		const long long per_xcd = (long long)blocks_per_cu * (cus / xcds);
		const long long margin = std::max(1ll, per_xcd / 16);
		const long long cap = (long long)xcds * (per_xcd - margin);
		return cap <= 0 ? 0 : cap > 0x3fffffff ? 0x3fffffff : (int)cap;
	}
	ResidentPlan resident_plan(int capacity, int wgs_per_unit, int units)
	{
		ResidentPlan p{0, 0};
		if (capacity <= 0 || wgs_per_unit <= 0 || units <= 0 || wgs_per_unit > capacity)
			return p;
		p.units_per_launch = std::min(units, capacity / wgs_per_unit);
		p.launches = (units + p.units_per_launch - 1) / p.units_per_launch;
		return p;
	}
	ResidentPlan resident_plan_two_forms(int capacity_a, int capacity_b, int wgs_per_unit, int units, bool *second)
	{
		const ResidentPlan a = resident_plan(capacity_a, wgs_per_unit, units), b = resident_plan(capacity_b, wgs_per_unit, units);
		const bool take_b = b.units_per_launch > 0 && (a.units_per_launch == 0 || b.launches < a.launches);
		if (second)
			*second = take_b;
		return take_b ? b : a;
	}
	namespace
	{
		constexpr int kMaxDevices = 64;
		struct Gate
		{
			std::mutex mu;
			hipEvent_t last = nullptr;
			bool recorded = false;
		};
		Gate &gate_of(int dev)
		{
			static Gate *gates = new Gate[kMaxDevices]; // (leaked on purpose: the runtime may be gone before static destructors run)
			return gates[dev < 0 || dev >= kMaxDevices ? 0 : dev];
		}
		struct CapKey
		{
			int dev;
			const void *kernel;
			int threads;
			size_t lds;
			bool operator<(const CapKey &o) const
			{
				return std::tie(dev, kernel, threads, lds) < std::tie(o.dev, o.kernel, o.threads, o.lds);
			}
		};
	} // namespace
	int resident_capacity(const void *kernel, int block_threads, size_t dynamic_lds, bool with_margin)
	{
		static std::mutex mu;
		static std::map<CapKey, int> *cache = new std::map<CapKey, int>;
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess)
			return 0;
		std::lock_guard<std::mutex> g(mu);
		const CapKey key{dev, kernel, with_margin ? block_threads : -block_threads, dynamic_lds};
		auto it = cache->find(key);
		if (it != cache->end())
			return it->second;
		int per_cu = 0, cus = 0, xcds = 0, cap = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, dynamic_lds) == hipSuccess &&
			hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
		{
			if (hipDeviceGetAttribute(&xcds, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess)
				xcds = 1;
			cap = with_margin ? resident_capacity_rule(per_cu, cus, xcds) : (int)std::min<long long>((long long)per_cu * cus, 0x3fffffff);
		}
		else
			(void)hipGetLastError();
		if (const char *ev = getenv("RIR_RESIDENT_CAPACITY")) // (tests, partitions the runtime misreports: an upper bound for every kernel)
			if (atoi(ev) >= 0)
				cap = std::min(cap, atoi(ev));
		(*cache)[key] = cap;
		return cap;
	}
	ResidentGate::ResidentGate(hipStream_t st) : st_(st)
	{
		int dev = 0;
		if (!hip_ok(hipGetDevice(&dev), "hipGetDevice"))
			return;
		Gate &g = gate_of(dev);
		g.mu.lock();
		gate_ = &g;
		if (!g.last && !hip_ok(hipEventCreateWithFlags(&g.last, hipEventDisableTiming), "hipEventCreate"))
			return;
		if (g.recorded && !hip_ok(hipStreamWaitEvent(st_, g.last, 0), "hipStreamWaitEvent"))
			return;
		ok_ = true;
	}
	ResidentGate::~ResidentGate()
	{
		if (!gate_)
			return;
		Gate &g = *static_cast<Gate *>(gate_);
		if (ok_ && g.last)
			g.recorded = hipEventRecord(g.last, st_) == hipSuccess || g.recorded;
		g.mu.unlock();
	}

	DeviceBuffer::~DeviceBuffer()
	{
		if (ptr)
			(void)hipFree(ptr);
	}
	void *DeviceBuffer::reserve(size_t bytes)
	{
		if (bytes <= cap && ptr)
			return ptr;
		if (ptr)
		{
			(void)hipFree(ptr);
			ptr = nullptr;
			cap = 0;
		}
		size_t want = bytes < 256 ? 256 : bytes;
		if (!hip_ok(hipMalloc(&ptr, want), "hipMalloc"))
		{
			ptr = nullptr;
			return nullptr;
		}
		cap = want;
		return ptr;
	}
	// Page-locking is slow (a saver's and a loader's staging buffers are 100 - 140 MB: tens of milliseconds, as much as recording
	// or reading a few hundred frames), so large page-locked buffers of objects that die are kept for the next object instead of
	// being returned to the system.  A buffer only enters the pool after the device has gone idle (work of the dead object may
	// still be in flight on some stream; hipHostFree would have waited for it too), and is only handed out for requests it fits
	// without much waste.  The pool is bounded and is never freed at exit (the runtime may already be gone by then).
	namespace
	{
		struct PinnedPool
		{
			std::mutex mu;
			struct Item
			{
				void *ptr;
				size_t cap;
				int device;
			};
			std::vector<Item> items;
			size_t total = 0;
			static constexpr size_t kMin = 1u << 20, kMaxTotal = 768u << 20;
		};
		PinnedPool &pinned_pool()
		{
			static PinnedPool *p = new PinnedPool; // (leaked on purpose)
			return *p;
		}
		void pinned_release(void *ptr, size_t cap)
		{
			int dev = 0;
			if (cap >= PinnedPool::kMin && hipGetDevice(&dev) == hipSuccess)
			{
				PinnedPool &pool = pinned_pool();
				std::unique_lock<std::mutex> lk(pool.mu);
				if (pool.total + cap <= PinnedPool::kMaxTotal)
				{
					lk.unlock();
					(void)hipDeviceSynchronize();
					lk.lock();
					pool.items.push_back({ptr, cap, dev});
					pool.total += cap;
					return;
				}
			}
			(void)hipHostFree(ptr);
		}
		void *pinned_acquire(size_t want, size_t &cap_out)
		{
			int dev = 0;
			if (want < PinnedPool::kMin || hipGetDevice(&dev) != hipSuccess)
				return nullptr;
			PinnedPool &pool = pinned_pool();
			std::lock_guard<std::mutex> g(pool.mu);
			size_t best = pool.items.size();
			for (size_t i = 0; i < pool.items.size(); ++i)
				if (pool.items[i].device == dev && pool.items[i].cap >= want && pool.items[i].cap <= want + want / 2 + PinnedPool::kMin &&
					(best == pool.items.size() || pool.items[i].cap < pool.items[best].cap))
					best = i;
			if (best == pool.items.size())
				return nullptr;
			void *p = pool.items[best].ptr;
			cap_out = pool.items[best].cap;
			pool.total -= cap_out;
			pool.items.erase(pool.items.begin() + (long)best);
			return p;
		}
	} // namespace
	PinnedBuffer::~PinnedBuffer()
	{
		if (ptr)
			pinned_release(ptr, cap);
	}
	void *PinnedBuffer::reserve(size_t bytes)
	{
		if (bytes <= cap && ptr)
			return ptr;
		if (ptr)
		{
			pinned_release(ptr, cap);
			ptr = nullptr;
			cap = 0;
		}
		size_t want = bytes < 256 ? 256 : bytes;
		size_t got = 0;
		if (void *p = pinned_acquire(want, got))
		{
			ptr = p, cap = got;
			return ptr;
		}
		if (!hip_ok(hipHostMalloc(&ptr, want, hipHostMallocDefault), "hipHostMalloc"))
		{
			ptr = nullptr;
			return nullptr;
		}
		cap = want;
		return ptr;
	}

} // namespace rir

namespace rir
{
	namespace
	{
		std::atomic<int> &gauss_order_flag()
		{
			static std::atomic<int> f{-1}; // -1: not decided yet (environment), 0 / 1
			return f;
		}
	} // namespace
	namespace
	{
		struct HostBlocks
		{
			std::mutex mu;
			std::map<uintptr_t, size_t> live; // start -> bytes
			size_t total = 0;
		};
		HostBlocks &host_blocks()
		{
			static HostBlocks *b = new HostBlocks; // (leaked on purpose: blocks may be freed by finalisers at process exit)
			return *b;
		}
		size_t host_blocks_limit()
		{
			static const size_t lim = [] {
				const char *e = std::getenv("RIR_HOST_ALLOC_MAX_MB");
				const long mb = e ? std::atol(e) : 256;
				return (size_t)(mb < 0 ? 0 : mb) << 20;
			}();
			return lim;
		}
	} // namespace
	void *host_block_alloc(size_t bytes)
	{
		if (bytes == 0 || !device_ready())
			return nullptr;
		HostBlocks &b = host_blocks();
		{
			std::lock_guard<std::mutex> g(b.mu);
			if (b.total + bytes > host_blocks_limit())
				return nullptr;
			b.total += bytes; // (reserved before the slow call below)
		}
		void *p = nullptr;
		if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p)
		{
			(void)hipGetLastError();
			std::lock_guard<std::mutex> g(b.mu);
			b.total -= bytes;
			return nullptr;
		}
		std::lock_guard<std::mutex> g(b.mu);
		b.live[(uintptr_t)p] = bytes;
		return p;
	}
	void host_block_free(void *p)
	{
		if (!p)
			return;
		HostBlocks &b = host_blocks();
		{
			std::lock_guard<std::mutex> g(b.mu);
			auto it = b.live.find((uintptr_t)p);
			if (it == b.live.end())
				return; // (not one of ours: left alone)
			b.total -= it->second;
			b.live.erase(it);
		}
		(void)hipHostFree(p);
	}
	bool host_block_contains(const void *p, size_t bytes)
	{
		HostBlocks &b = host_blocks();
		std::lock_guard<std::mutex> g(b.mu);
		if (b.live.empty())
			return false;
		auto it = b.live.upper_bound((uintptr_t)p);
		if (it == b.live.begin())
			return false;
		--it;
		return (uintptr_t)p >= it->first && (uintptr_t)p + bytes <= it->first + it->second;
	}
	bool abi_zero_copy()
	{
		static const bool on = [] {
			const char *e = std::getenv("RIR_ABI_ZERO_COPY");
			return !(e && e[0] == '0');
		}();
		return on;
	}
	bool gaussian_reference_order()
	{
		int v = gauss_order_flag().load(std::memory_order_relaxed);
		if (v < 0)
		{
			const char *e = std::getenv("RIR_GAUSSIAN_REFERENCE_ORDER");
			v = (e && e[0] && e[0] != '0') ? 1 : 0;
			gauss_order_flag().store(v, std::memory_order_relaxed);
		}
		return v != 0;
	}
	void set_gaussian_reference_order(bool on) { gauss_order_flag().store(on ? 1 : 0, std::memory_order_relaxed); }
} // namespace rir
// Page-locked memory for the caller's own images (include/rir_amd_device.h): an entry point that is given buffers inside such a block
// works on them in place.  NULL when there is no device, or when the blocks handed out would exceed RIR_HOST_ALLOC_MAX_MB.
RIR_EXPORT void *rir_host_alloc(int64_t bytes) { return bytes > 0 ? rir::host_block_alloc((size_t)bytes) : nullptr; }
RIR_EXPORT void rir_host_free(void *p) { rir::host_block_free(p); }
RIR_EXPORT int rir_host_is_page_locked(const void *p, int64_t bytes) { return p && bytes > 0 && rir::host_block_contains(p, (size_t)bytes) ? 1 : 0; }
RIR_EXPORT void rir_set_gaussian_reference_order(int on) { rir::set_gaussian_reference_order(on != 0); }
RIR_EXPORT int rir_gaussian_reference_order() { return rir::gaussian_reference_order() ? 1 : 0; }

// ---- exported `tools` symbols ---------------------------------------------------------------------

RIR_EXPORT void set_print_function(rir::print_function function)
{
	std::lock_guard<std::mutex> g(rir::log_mutex());
	rir::log_state().fn = function;
	rir::log_state().enabled = true;
}
RIR_EXPORT void disable_print()
{
	std::lock_guard<std::mutex> g(rir::log_mutex());
	rir::log_state().enabled = false;
}
RIR_EXPORT void reset_print_functions()
{
	std::lock_guard<std::mutex> g(rir::log_mutex());
	rir::log_state().fn = nullptr;
	rir::log_state().enabled = true;
}
// -1 and *len = required length when the buffer is too small; no terminator is written (Log.cpp:72-85)
RIR_EXPORT int get_last_log_error(char *text, int *len)
{
	std::lock_guard<std::mutex> g(rir::log_mutex());
	const std::string &e = rir::log_state().last_error;
	if (!len)
		return -1;
	if (*len < (int)e.size() || !text)
	{
		*len = (int)e.size();
		return -1;
	}
	*len = (int)e.size();
	std::memcpy(text, e.data(), e.size());
	return 0;
}

// ---- the C++ names of the reference's tools/Log.h ------------------------------------------------------
// The reference's libgeometry.so - which a drop-in keeps - is linked against libtools.so and binds
// rir::logError (src/cpp/geometry/geometry.cpp:147,214; declared TOOLS_EXPORT in src/cpp/tools/Log.h:31-35).  ctypes loads
// with RTLD_NOW, so the wrapper's loadDlls() (src/python/librir/low_level/misc.py:129-134) fails on the geometry line unless
// the library that stands in for libtools.so exports the mangled names too.  All of Log.h:20-40 is here, forwarding to the
// one log state above, so that a geometry built at another version still binds.
#define RIR_CXX_EXPORT __attribute__((visibility("default")))
namespace rir
{
	typedef print_function log_print_function; // Log.h:20
	RIR_CXX_EXPORT void logInfo(const char *text) { log_message(LOG_INFO, text); }
	RIR_CXX_EXPORT void logWarning(const char *text) { log_message(LOG_WARNING, text); }
	RIR_CXX_EXPORT void logError(const char *text) { log_message(LOG_ERROR, text); }
	RIR_CXX_EXPORT void set_log_function(log_print_function function)
	{
		std::lock_guard<std::mutex> g(log_mutex());
		log_state().fn = function;
		log_state().enabled = true;
	}
	RIR_CXX_EXPORT void disable_log()
	{
		std::lock_guard<std::mutex> g(log_mutex());
		log_state().enabled = false;
	}
	RIR_CXX_EXPORT log_print_function log_function()
	{
		std::lock_guard<std::mutex> g(log_mutex());
		return log_state().fn;
	}
	RIR_CXX_EXPORT void reset_log_function()
	{
		std::lock_guard<std::mutex> g(log_mutex());
		log_state().fn = nullptr;
		log_state().enabled = true;
	}
	RIR_CXX_EXPORT int getLastErrorLog(char *text, int *len) { return ::get_last_log_error(text, len); }
} // namespace rir

// The reference stores any BaseShared-derived object; here only objects created by this library
// (rir::Object) can be registered.
RIR_EXPORT int set_void_ptr(void *obj)
{
	if (!obj)
		return -1;
	return rir::register_object(static_cast<rir::Object *>(obj)->shared_from_this());
}
RIR_EXPORT void *get_void_ptr(int index) { return rir::lookup_object(index).get(); }
RIR_EXPORT void rm_void_ptr(int index) { rir::remove_object(index); }

// The residency rules of runtime.h as plain functions (no device needed): what tests/test_abi.py checks.
// rir_resident_capacity_rule: workgroups of a kernel with blocks_per_cu resident workgroups per CU that may wait for each other
// in one launch on a device of `cus` CUs in `xcds` XCDs.  rir_resident_plan: out[0] = units per launch (0 = a unit does not fit:
// the caller's launch-per-frame / launch-per-iteration path), out[1] = launches; returns 0, -1 on a NULL pointer.
RIR_EXPORT int rir_resident_capacity_rule(int blocks_per_cu, int cus, int xcds) { return rir::resident_capacity_rule(blocks_per_cu, cus, xcds); }
// rir_resident_plan_two_forms: out[0] = units per launch when filled evenly, out[1] = launches, out[2] = 1 when the second form is taken.
RIR_EXPORT int rir_resident_plan_two_forms(int capacity_a, int capacity_b, int wgs_per_unit, int units, int *out3)
{
	if (!out3)
		return -1;
	bool second = false;
	const rir::ResidentPlan p = rir::resident_plan_two_forms(capacity_a, capacity_b, wgs_per_unit, units, &second);
	out3[0] = p.units_per_launch > 0 ? rir::resident_batch(p, units) : 0, out3[1] = p.launches, out3[2] = second ? 1 : 0;
	return 0;
}
RIR_EXPORT int rir_resident_plan(int capacity, int wgs_per_unit, int units, int *out2)
{
	if (!out2)
		return -1;
	const rir::ResidentPlan p = rir::resident_plan(capacity, wgs_per_unit, units);
	out2[0] = p.units_per_launch, out2[1] = p.launches;
	return 0;
}
