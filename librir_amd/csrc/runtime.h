// Host runtime shared by the C-ABI translation units: log + last error, the int-handle registry,
// and the HIP device context (stream, grow-only device/pinned buffers).
//
// Behavioural model (not code) from the reference:
//   log / last error    src/cpp/tools/Log.cpp:11-85, tools.h:39-55
//   handle registry     src/cpp/tools/tools.cpp:40-84  (ints >= 1, smallest free slot reused,
//                       one namespace for cameras, savers, bad-pixel and attribute objects)
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#define RIR_EXPORT extern "C" __attribute__((visibility("default")))

namespace rir
{
	// ---- log -----------------------------------------------------------------------------------
	enum
	{
		LOG_INFO = 0,
		LOG_WARNING = 1,
		LOG_ERROR = 2
	};
	typedef void (*print_function)(int, const char *);
	void log_message(int level, const char *text);
	inline void log_error(const std::string &s) { log_message(LOG_ERROR, s.c_str()); }
	inline void log_warning(const std::string &s) { log_message(LOG_WARNING, s.c_str()); }
	inline void log_info(const std::string &s) { log_message(LOG_INFO, s.c_str()); }

	// ---- handle registry -----------------------------------------------------------------------
	// Every object reachable through an int handle derives from Object; the registry keeps a
	// shared_ptr so the object lives exactly as long as its handle.
	struct Object : public std::enable_shared_from_this<Object>
	{
		virtual ~Object() {}
		virtual const char *type_name() const = 0;
	};
	int register_object(const std::shared_ptr<Object> &obj); // -> handle >= 1
	std::shared_ptr<Object> lookup_object(int handle);		  // empty when unknown
	void remove_object(int handle);

	template <class T>
	std::shared_ptr<T> lookup_as(int handle)
	{
		return std::dynamic_pointer_cast<T>(lookup_object(handle));
	}

	// ---- device context --------------------------------------------------------------------------
	// The product path has NO CPU fallback: every compute entry point first calls device_ready(),
	// which logs and returns false when no HIP device is usable, and the entry point returns its
	// error code.
	// Test hooks (fault injection: RIR_DEBUG_LOSSY_GIVE_UP, RIR_DEBUG_LOSSY_BAIL, RIR_DEBUG_ECC_BAIL) exist only in the build made with
	// -DRIR_TEST_HOOKS (librir_amd/build.py: libs/librir_amd_testhooks.so, loaded by the tests that need it); the product library reads none of them.
#ifdef RIR_TEST_HOOKS
	inline const char *test_hook(const char *name) { return getenv(name); }
#else
	inline const char *test_hook(const char *) { return nullptr; }
#endif
	inline bool test_hook_is(const char *name, const char *value)
	{
		const char *v = test_hook(name);
		return v && std::strcmp(v, value) == 0;
	}
	bool device_ready();
	// gaussian_filter as the 2-D sum in the reference's own order (signal_processing.cpp:101-148: dx outer, dy inner, one rounding per product
	// and per sum) instead of the separable form: bit-identical results instead of results within 2e-6, three times the time.  Process-wide,
	// default off; RIR_GAUSSIAN_REFERENCE_ORDER=1 in the environment or rir_set_gaussian_reference_order(1) turn it on.
	bool gaussian_reference_order();
	void set_gaussian_reference_order(bool on);
	hipStream_t default_stream(); // one non-blocking stream owned by the library

	struct DeviceBuffer
	{ // grow-only device allocation
		void *ptr = nullptr;
		size_t cap = 0;
		~DeviceBuffer();
		void *reserve(size_t bytes); // nullptr on failure (logged)
		template <class T>
		T *as() { return static_cast<T *>(ptr); }
	};
	struct PinnedBuffer
	{ // grow-only page-locked host allocation
		void *ptr = nullptr;
		size_t cap = 0;
		~PinnedBuffer();
		void *reserve(size_t bytes);
		template <class T>
		T *as() { return static_cast<T *>(ptr); }
	};

	// ---- resident launches ------------------------------------------------------------------------
	// Three kernel families make workgroups of ONE launch wait for each other (lossy_run_kernel, ecc_run_kernel and, through its
	// tickets, rirb1_encode_dense): every workgroup such a launch waits for must be on the chip, so (1) a launch may not be
	// larger than what the device holds at once and (2) two such launches - whatever their kernels, streams or calling threads -
	// may not run side by side, each holding a part of the chip and waiting for the rest.  The handle-based ABI lets different
	// objects be driven from different threads (reference registry mutex, tools.cpp:46-50), so both are enforced here, for the
	// process, per device - not by each kernel family for itself.
	//
	// (1) resident_capacity: workgroups of `kernel` (block of `block_threads` threads, `dynamic_lds` bytes) the current device holds
	// at once, from hipOccupancyMaxActiveBlocksPerMultiprocessor x multiProcessorCount, less a margin (resident_capacity_rule).
	// 0 when the runtime cannot tell: the caller then takes its launch-per-frame / launch-per-iteration path.  Cached.
	// with_margin false: every place of the device - only for kernels that find out at their start whether they are resident and
	// whose host side repeats a launch that was not (resident_device.h): there a launch that does not fit costs a detour, not an error.
	int resident_capacity(const void *kernel, int block_threads, size_t dynamic_lds, bool with_margin = true);
	// The rule alone (no device needed; unit-tested on the CPU): workgroup i of a launch starts on XCD i % xcds and every XCD fills
	// its own CUs, so what counts is an XCD's places, blocks_per_cu x (cus / xcds); one sixteenth of them (at least one) stays
	// free for whatever else is running.  MI355X, a kernel with 5 workgroups per CU: 8 x (160 - 10) = 1 200.
	int resident_capacity_rule(int blocks_per_cu, int cus, int xcds);
	// How `units` independent units (streams, sequences) of `wgs_per_unit` workgroups each go through a resident kernel of the
	// given capacity: units per launch (0: a unit does not fit at all - take the non-resident path) and the number of launches.
	struct ResidentPlan
	{
		int units_per_launch, launches;
	};
	ResidentPlan resident_plan(int capacity, int wgs_per_unit, int units);
	// A kernel built in two forms - the second holds more workgroups at once (capacity_b > capacity_a) but runs a unit a little slower:
	// the second form is taken when it saves a launch (or when only it fits); *second says which.  The plan's launches are then
	// filled evenly (resident_batch): 32 units at 9 per launch go as 8, 8, 8, 8.
	ResidentPlan resident_plan_two_forms(int capacity_a, int capacity_b, int wgs_per_unit, int units, bool *second);
	inline int resident_batch(const ResidentPlan &p, int units) { return p.launches > 0 ? (units + p.launches - 1) / p.launches : 0; }
	// (2) the gate: construct it right before the launch, on the launching thread, with the launch's stream; it makes that stream
	// wait for the previous resident launch of the device (whatever stream that went to) and, when it goes out of scope, leaves
	// its event behind the launch.  Host side it holds the device's mutex for the duration of the launch call only.
	class ResidentGate
	{
	public:
		explicit ResidentGate(hipStream_t st);
		~ResidentGate();
		bool ok() const { return ok_; }
		ResidentGate(const ResidentGate &) = delete;
		ResidentGate &operator=(const ResidentGate &) = delete;

	private:
		hipStream_t st_;
		void *gate_ = nullptr;
		bool ok_ = false;
	};

	// memcpy of a frame between the caller's memory and page-locked staging, spread over a few helper threads (host_copy.cpp): the
	// per-frame entry points are bound by this copy, not by the link.  Plain memcpy for small sizes / RIR_HOST_COPY_THREADS=0.
	void host_copy(void *dst, const void *src, size_t bytes);
	// a chunk between page-locked memory and the file, the same way (disjoint ranges of one descriptor; the whole range or false)
	bool host_pread(int fd, void *dst, size_t bytes, int64_t file_off);
	bool host_pwrite(int fd, const void *src, size_t bytes, int64_t file_off);
	int host_copy_threads(); // helpers a copy may use

	bool hip_ok(hipError_t e, const char *what); // logs "what: hipGetErrorString" on failure
	hipError_t wait_stream(hipStream_t st);		  // polls the stream (short waits without the wake-up latency of a blocking one)
	hipError_t wait_event(hipEvent_t ev);		  // the same for an event
	// The per-frame entry points let the codec kernels work straight on page-locked host memory (the saver's staged frames are read, its
	// tables and payload written, the loader's frames written over the link by the kernels themselves: no copy calls, no device round trip
	// per chunk).  RIR_ABI_ZERO_COPY=0 turns that off (copies through device buffers, as before round 5).
	bool abi_zero_copy();
	// Page-locked memory handed OUT to callers (rir_host_alloc: the Python mirror builds the arrays it returns on such blocks).  An entry
	// point that finds a caller's buffer wholly inside a live block runs its kernel on that memory as it is - nothing is staged, nothing is
	// copied back - as it does on its own staging buffers.  A registry of this library's own blocks (a map under a mutex: ~0.1 us a look-up),
	// not a query of the runtime.  The blocks together stay below RIR_HOST_ALLOC_MAX_MB (256): beyond that host_block_alloc says no and
	// the caller takes ordinary memory.
	void *host_block_alloc(size_t bytes);
	void host_block_free(void *p);
	bool host_block_contains(const void *p, size_t bytes);

} // namespace rir
