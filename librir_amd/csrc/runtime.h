// Host runtime shared by the C-ABI translation units: log + last error, the int-handle registry,
// and the HIP device context (stream, grow-only device/pinned buffers).
//
// Behavioural model (not code) from the reference:
//   log / last error    src/cpp/tools/Log.cpp:11-85, tools.h:39-55
//   handle registry     src/cpp/tools/tools.cpp:40-84  (ints >= 1, smallest free slot reused,
//                       one namespace for cameras, savers, bad-pixel and attribute objects)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
	bool device_ready();
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

	bool hip_ok(hipError_t e, const char *what); // logs "what: hipGetErrorString" on failure
	hipError_t wait_stream(hipStream_t st);		  // polls the stream (short waits without the wake-up latency of a blocking one)

} // namespace rir
