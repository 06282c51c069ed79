// extract_times / resample_time_serie: the two time-AXIS helpers of the signal_processing C ABI
// (reference src/cpp/signal_processing/signal_processing.cpp:158-195 over Filters.cpp:111-333).
//
// They work on the timestamps of recordings - a few thousand doubles, one dependent step after the other - not on pixels: host
// bookkeeping like the saver's timestamp tables and hash_bytes, with no device work to do.  Restated from the behaviour of the
// reference, double for double (same comparisons in the same order, one rounding per operation: this library is compiled with
// -ffp-contract=off), and checked against the compiled reference (tests/test_time_series.py, tests/golden/time_series.npz).
//
// Inputs on which the reference does not terminate or reads outside its arguments (listed at each function) are refused with -1 and a
// logged message instead.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "runtime.h"

using namespace rir;

namespace
{
	enum
	{ // Filters.h:212-218
		kIntersection = 0x01,
		kPadd = 0x02,
		kInterpolate = 0x04
	};

	struct Span
	{ // a run of one input vector still to be merged
		const double *at, *end;
	};

	// One time axis out of several (Filters.cpp:111-207).  A vector holding a NaN is two runs, before and after it.  Each step emits the
	// lowest head among the runs and moves every run whose head equals it one place on; duplicates ACROSS vectors collapse, duplicates
	// inside one vector stay.  kIntersection first narrows every run to [latest first element, earliest last element] of the vectors.
	// -> false where the reference would never finish: a run that is empty from the start (no element, a NaN at either end of a vector,
	// a second NaN, a run wholly outside the intersection) keeps its place in the reference's list for ever.
	bool merge_times(const std::vector<Span> &vectors, int strategy, std::vector<double> &out)
	{
		for (const Span &v : vectors)
			if (v.at == v.end)
				return false;
		std::vector<Span> runs;
		for (const Span &v : vectors)
		{
			const double *nan_at = v.at;
			while (nan_at != v.end && !std::isnan(*nan_at))
				++nan_at;
			if (nan_at == v.end)
				runs.push_back(v);
			else
			{
				runs.push_back(Span{v.at, nan_at});
				runs.push_back(Span{nan_at + 1, v.end});
			}
		}
		double lo = 0, hi = -1;
		if (strategy & kIntersection)
			for (const Span &v : vectors)
			{
				const double first = *v.at, last = *(v.end - 1);
				if (hi < lo)
				{
					lo = first;
					hi = last;
					continue;
				}
				if (last < lo || first > hi)
					return true; // nothing in common: an empty axis (Filters.cpp:166-169)
				lo = std::max(lo, first);
				hi = std::min(hi, last);
			}
		for (Span &r : runs)
		{
			for (const double *p = r.at; p != r.end; ++p)
				if (std::isnan(*p))
					return false;
			if (strategy & kIntersection)
			{
				while (r.at != r.end && *r.at < lo)
					++r.at;
				while (r.end != r.at && *(r.end - 1) > hi)
					--r.end;
			}
			if (r.at == r.end)
				return false;
		}
		while (!runs.empty())
		{
			double t = *runs.front().at;
			for (size_t i = 1; i < runs.size(); ++i)
				t = std::min(t, *runs[i].at);
			for (size_t i = 0; i < runs.size();)
			{
				if (*runs[i].at == t && ++runs[i].at == runs[i].end)
					runs.erase(runs.begin() + (std::ptrdiff_t)i);
				else
					++i;
			}
			out.push_back(t);
		}
		return true;
	}

	// A series (x, y) read at other times (Filters.cpp:209-333).  One cursor walks the samples as the times go by; a time that meets a
	// sample exactly as the cursor's own sample consumes it, one that meets a sample after the cursor moved does not (so a repeated time
	// reads the same sample twice in the second case and the next one in the first - as upstream).
	// -> false where the reference reads before its first sample (only reachable with a NaN among the times or the samples).
	bool resample(const double *x, const double *y, size_t n, const double *times, size_t m, int strategy, double padd, double *out)
	{
		const bool padded = (strategy & kPadd) != 0, interpolate = (strategy & kInterpolate) != 0;
		if (n == 0)
		{
			for (size_t t = 0; t < m; ++t)
				out[t] = padded ? padd : 0.0;
			return true;
		}
		auto between = [&](size_t k, double time) {
			const double x0 = x[k - 1], y0 = y[k - 1], x1 = x[k], y1 = y[k];
			if (interpolate)
			{
				const double f = (time - x0) / (x1 - x0);
				return y1 * f + (1 - f) * y0;
			}
			return (time - x0 < x1 - time) ? y0 : y1;
		};
		size_t k = 0;
		for (size_t t = 0; t < m; ++t)
		{
			const double time = times[t];
			if (k == n)
				out[t] = padded ? padd : y[n - 1];
			else if (time == x[k])
				out[t] = y[k++];
			else if (time < x[k])
				out[t] = k == 0 ? (padded ? padd : y[0]) : between(k, time);
			else
			{
				while (k != n && x[k] < time)
					++k;
				if (k == n)
					out[t] = padded ? padd : y[n - 1];
				else if (x[k] == time)
					out[t] = y[k];
				else if (k == 0)
					return false;
				else
					out[t] = between(k, time);
			}
		}
		return true;
	}
} // namespace

// signal_processing.cpp:158-181: 0, or -2 with the needed size in *output_size when the output is too small
RIR_EXPORT int extract_times(double *vectors, int vector_count, int *vector_sizes, int s, double *output, int *output_size)
{
	if (vector_count < 0 || !output_size || (vector_count > 0 && (!vectors || !vector_sizes)))
		return -1;
	std::vector<Span> in;
	const double *at = vectors;
	for (int i = 0; i < vector_count; ++i)
	{
		if (vector_sizes[i] < 0)
			return -1;
		in.push_back(Span{at, at + vector_sizes[i]});
		at += vector_sizes[i];
	}
	std::vector<double> res;
	if (vector_count == 1)
		res.assign(in[0].at, in[0].end); // as it is (Filters.cpp:115-118)
	else if (vector_count > 1 && !merge_times(in, s, res))
	{
		log_error("extract_times: a time vector that is empty, starts or ends with a NaN, holds two of them or lies outside the common range "
				  "(the reference does not return on such input)");
		return -1;
	}
	if ((int)res.size() > *output_size)
	{
		*output_size = (int)res.size();
		return -2;
	}
	if (!res.empty())
	{
		if (!output)
			return -1;
		std::memcpy(output, res.data(), res.size() * sizeof(double));
	}
	*output_size = (int)res.size();
	return 0;
}

// signal_processing.cpp:183-195: 0, or -1 with the needed size in *output_size when the output is too small
RIR_EXPORT int resample_time_serie(double *sample_x, double *sample_y, int size, double *times, int times_size, int s, double padds, double *output,
								   int *output_size)
{
	if (size < 0 || times_size < 0 || !output_size || (size > 0 && (!sample_x || !sample_y)) || (times_size > 0 && !times))
		return -1;
	if (times_size > *output_size)
	{
		*output_size = times_size;
		return -1;
	}
	if (times_size > 0 && !output)
		return -1;
	std::vector<double> res((size_t)times_size);
	if (!resample(sample_x, sample_y, (size_t)size, times, (size_t)times_size, s, padds, res.data()))
	{
		log_error("resample_time_serie: a NaN among the times or the samples ahead of the first sample (the reference reads before its input there)");
		return -1;
	}
	if (times_size > 0)
		std::memcpy(output, res.data(), res.size() * sizeof(double));
	*output_size = times_size;
	return 0;
}
