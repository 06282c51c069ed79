// ThreadSanitizer harness for librir_amd/csrc/host_copy.cpp (build container, no GPU): callers on many threads, sizes around the
// hand-off threshold, pauses that let the helpers park, file jobs beside memory jobs.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
namespace rir
{
	void host_copy(void *dst, const void *src, size_t bytes);
	bool host_pread(int fd, void *dst, size_t bytes, int64_t file_off);
	bool host_pwrite(int fd, const void *src, size_t bytes, int64_t file_off);
}
extern "C" int rir_host_touch(void *buf, int64_t bytes);
int main()
{
	std::atomic<int> bad{0};
	char name[] = "/tmp/tsan/io_XXXXXX";
	int fd = mkstemp(name);
	std::vector<std::thread> ts;
	for (int t = 0; t < 10; ++t)
		ts.emplace_back([&, t] {
			unsigned s = 1234u + t;
			auto rnd = [&] { s = s * 1664525u + 1013904223u; return s >> 8; };
			std::vector<unsigned char> a(1 << 20), b(1 << 20);
			for (auto &x : a)
				x = (unsigned char)rnd();
			for (int i = 0; i < 300; ++i)
			{
				const size_t n = (i % 3 == 0) ? rnd() % a.size() : 192 * 1024 + rnd() % (a.size() - 192 * 1024);
				if (t < 7)
				{
					rir::host_copy(b.data(), a.data(), n);
					if (memcmp(a.data(), b.data(), n) != 0)
						bad++;
					memset(b.data(), 0, n);
				}
				else
				{
					const int64_t off = (int64_t)t * (2 << 20);
					if (!rir::host_pwrite(fd, a.data(), n, off) || !rir::host_pread(fd, b.data(), n, off) || memcmp(a.data(), b.data(), n) != 0)
						bad++;
				}
				if (i % 50 == 49)
					std::this_thread::sleep_for(std::chrono::milliseconds(2)); // the helpers park
			}
		});
	for (auto &t : ts)
		t.join();
	close(fd);
	unlink(name);
	// rir_host_touch beside the copies that fill the same pages (low_level/misc.py touch_ahead): the byte compare-and-swap against the copies'
	// plain stores is the one race this library means to have (x86-64: a locked read-modify-write that puts back what it read) - it is
	// suppressed by name (scripts/tsan_host_copy.supp), everything else in the run still counts; what is checked here is what the race
	// could break: the copied bytes
	{
		const size_t n = (size_t)48 << 20;
		std::vector<unsigned char> a(n);
		unsigned s = 99u;
		for (auto &x : a)
			x = (unsigned char)((s = s * 1664525u + 1013904223u) >> 24);
		for (int rep = 0; rep < 4; ++rep)
		{
			unsigned char *b = static_cast<unsigned char *>(malloc(n)); // (fresh pages each time)
			std::thread toucher([&] { rir_host_touch(b, (int64_t)n); });
			for (size_t o = 0; o < n; o += (size_t)1 << 20)
				rir::host_copy(b + o, a.data() + o, (size_t)1 << 20);
			toucher.join();
			if (memcmp(a.data(), b, n) != 0)
				bad++;
			free(b);
		}
	}
	printf("host_copy (+ rir_host_touch beside it) under ThreadSanitizer: %d mismatches\n", bad.load());
	return bad.load() ? 1 : 0;
}
