// Frame-buffer image kernels of the librir hot path — gfx950 (CDNA4).
//
// Each kernel is batched over N independent frames laid out [n][h][w] row-major (index = x + y*w,
// the reference layout) and reproduces the reference arithmetic exactly: integer results are
// bit-exact, float results are computed with the same operation order, separate multiply and add
// (this file is compiled with -ffp-contract=off; the reference x86-64 build has no FMA).
//
//   translate          reference src/cpp/signal_processing/Filters.h:249-326
//   gaussian_filter    reference src/cpp/signal_processing/signal_processing.cpp:101-148
//   bad pixel detector reference src/cpp/signal_processing/Filters.h:135-193
//   bad pixel correct  reference src/cpp/signal_processing/BadPixels.cpp:34-66, Filters.cpp:7-50
//   read-back repair   reference src/cpp/video_io/IRFileLoader.cpp:722-802
//   motion removal     reference src/cpp/video_io/IRFileLoader.cpp:617-627
//   quantile           reference src/cpp/signal_processing/Filters.cpp:56-101
//   3x3 median filter  reference src/cpp/signal_processing/Filters.h:71-129
//
// All of them are HBM-bound byte/integer work: loads and stores are coalesced
// along x, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

#include "filter_kernels.h"
#include "runtime.h"

namespace rir
{
	// XCD-aware workgroup order.  The dispatcher deals workgroups round-robin to the 8 XCDs of the MI355X, each
	// with its own L2: neighbours in the launch order never share a cache.  Re-reading the linear id "XCD-major"
	// (XCD k owns the k-th eighth of the logical order) makes logical neighbours - tiles that share halo rows or
	// split cache lines - run on the same XCD one after the other.  Measured: gaussian_filter 0.173 -> 0.131 ms per
	// 256 frames, translate (63-pixel tile rows, never line-aligned) 0.109 -> 0.097 ms; the 3x3 median, bound by
	// its instruction count, gains nothing and keeps the plain order.
	__device__ __forceinline__ unsigned xcd_major(unsigned id, unsigned total)
	{
		const unsigned per = total / 8u;
		return id < per * 8u ? (id % 8u) * per + id / 8u : id;
	}


	// ---- translate ------------------------------------------------------------------------------

	template <class U>
	struct CastTo
	{
		// in-range values: truncation toward zero like static_cast on the host
		__device__ static U from(double v) { return static_cast<U>(v); }
	};
	template <>
	struct CastTo<bool>
	{
		__device__ static bool from(double v) { return v != 0; }
	};
	// sub-int integer types convert through int32 on the host (cvttsd2si + truncation)
	template <>
	struct CastTo<int8_t>
	{
		__device__ static int8_t from(double v) { return (int8_t)(int32_t)v; }
	};
	template <>
	struct CastTo<uint8_t>
	{
		__device__ static uint8_t from(double v) { return (uint8_t)(int32_t)v; }
	};
	template <>
	struct CastTo<int16_t>
	{
		__device__ static int16_t from(double v) { return (int16_t)(int32_t)v; }
	};
	template <>
	struct CastTo<uint16_t>
	{
		__device__ static uint16_t from(double v) { return (uint16_t)(int32_t)v; }
	};
	template <>
	struct CastTo<uint32_t>
	{
		__device__ static uint32_t from(double v) { return (uint32_t)(int64_t)v; }
	};

	// uint16 output of a float translate: the value is first rounded to float (what translate<float,float> stores)
	// and then truncated, i.e. exactly "translate, then convert", without the intermediate frame in memory
	struct u16_via_f32
	{
		uint16_t v;
	};
	template <>
	struct CastTo<u16_via_f32>
	{
		__device__ static u16_via_f32 from(double v) { return u16_via_f32{(uint16_t)(int32_t)(float)v}; }
	};
	template <class T, class U>
	__device__ __forceinline__ U tap_as(T p)
	{
		if constexpr (std::is_same<T, U>::value)
			return p;
		else if constexpr (std::is_same<U, u16_via_f32>::value)
			return CastTo<U>::from((double)p);
		else
			return (U)p;
	}

	__device__ __forceinline__ uint64_t f2sz(float v) { return (uint64_t)(int64_t)v; }
	__device__ __forceinline__ uint64_t wrap_sz(uint64_t value, uint64_t max) { return (value + max) % max; }

	// One output pixel (x, y) of translate<T,U>; returns false when the pixel is left untouched
	// (strategy "noborder" outside the source).  Arithmetic exactly as the reference: float
	// coordinates, double blend, truncating cast.
	// SMALL: |dx|, |dy| < 2^30, so every coordinate fits an int32 and the float -> size_t conversions
	// (a long software sequence for 64 bits) are done in 32 bits, sign-extended: same values.
	template <bool SMALL>
	__device__ __forceinline__ uint64_t f2szT(float v)
	{
		return SMALL ? (uint64_t)(int64_t)(int32_t)v : (uint64_t)(int64_t)v;
	}

	template <class T, class U, bool SMALL = false>
	__device__ __forceinline__ bool translate_px(const T *__restrict__ s, uint64_t w, uint64_t h, uint64_t x, int y, float dx, float dy, int strategy,
												 U background, U &out)
	{
		const float px = (float)x - dx;
		const float py = (float)y - dy;
		if (px < 0 || px >= (float)w || py < 0 || py >= (float)h)
		{
			if (strategy == TRANSLATE_UNCHANGED)
				return false;
			if (strategy == TRANSLATE_SOURCE)
			{
				out = tap_as<T, U>(s[x + (uint64_t)y * w]);
				return true;
			}
			if (strategy == TRANSLATE_CONSTANT)
			{
				out = background;
				return true;
			}
			if (strategy == TRANSLATE_WRAP)
			{
				const uint64_t l = wrap_sz(f2szT<SMALL>(px), w), r = wrap_sz(f2szT<SMALL>(px + 1), w);
				const uint64_t t = wrap_sz(f2szT<SMALL>(py), h), b = wrap_sz(f2szT<SMALL>(py + 1), h);
				const T p1 = s[b * w + l], p2 = s[t * w + l], p3 = s[b * w + r], p4 = s[t * w + r];
				const double u = fabsf(px - (float)(int)px);
				const double v = fabsf(py - (float)(int)py);
				out = CastTo<U>::from(((double)p1 * (1 - v) + (double)p2 * v) * (1 - u) + ((double)p3 * (1 - v) + (double)p4 * v) * u);
				return true;
			}
			uint64_t _x, _y;
			if (px < 0)
				_x = 0;
			else if (px >= (float)w)
				_x = w - 1;
			else
				_x = f2szT<SMALL>(px);
			if (py < 0)
				_y = 0;
			else if (py >= (float)h)
				_y = h - 1;
			else
				_y = f2szT<SMALL>(py);
			out = tap_as<T, U>(s[_x + _y * w]);
			return true;
		}
		// (reference: r == w, b == h.  px + 1 can round up to w + 1 - w a power of two, px one ulp below w - and the
		// reference then reads out of bounds; nothing here may, so that tap is the left / top one as well)
		const uint64_t l = f2szT<SMALL>(px);
		uint64_t r = f2szT<SMALL>(px + 1);
		if (r >= w)
			r = l;
		const uint64_t t = f2szT<SMALL>(py);
		uint64_t b = f2szT<SMALL>(py + 1);
		if (b >= h)
			b = t;
		const T p1 = s[b * w + l], p2 = s[t * w + l], p3 = s[b * w + r], p4 = s[t * w + r];
		const double u = (px - (float)l);
		const double v = ((float)b - py);
		out = CastTo<U>::from(((double)p1 * (1 - v) + (double)p2 * v) * (1 - u) + ((double)p3 * (1 - v) + (double)p4 * v) * u);
		return true;
	}

	// ---- translate, one wave per tile --------------------------------------------------------------------------
	// A kernel that gives every lane a run of 8 output pixels (the first version) spends ~46 vector instructions per pixel, nearly all of them the reference's
	// double-precision blend and the conversions around it: it is bound by the VALU at a third of the memory rate.
	// A translation is uniform, so neighbouring outputs share their operands: here lane i of a wave owns output column
	// x0 + i of a tile of OW x OH pixels and LOADS source column l0 + i for the OH + 1 source rows under the tile (one
	// coalesced raw-buffer load per row).  Then for every row the vertical blend of the lane's own column,
	// c = p_b (1 - v) + p_t v, is computed once; the blend of the right tap column is lane i + 1's c (a DPP wave
	// shift), and the pixel is c (1 - u) + c_right u - the reference's expression (Filters.h:310-322) with every
	// product computed once instead of twice and no per-pixel address arithmetic: ~12 instructions per pixel.
	// That holds for "regular" tiles (every output inside the source, taps at l0 + i / l0 + i + 1 and t0 + j / t0 + j + 1,
	// one vertical weight - checked per tile with the reference's own float expressions, so rounding quirks of
	// px + 1 or of the weights send a tile to the general path, never to a wrong result); the last source column /
	// row (r == w -> l, b == h -> t) and columns left or right of the source under "nearest" / "background" /
	// "noborder" are handled in place.  Every other tile (rows above / below the source, "wrap", ...) runs
	// translate_px pixel by pixel.  Bit-identical to the reference.  (0.135 -> 0.097 ms per 256 frames 640x512 uint16; the XCD-major order is a quarter of that.)
#ifndef RIR_TR_TY
#define RIR_TR_TY 16
#endif
	template <class T>
	__device__ __forceinline__ T buffer_load_px(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
	{
		if constexpr (sizeof(T) == 1)
			return __builtin_bit_cast(T, (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)voff, (int)soff, 0));
		else if constexpr (sizeof(T) == 2)
			return __builtin_bit_cast(T, (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)voff, (int)soff, 0));
		else if constexpr (sizeof(T) == 4)
			return __builtin_bit_cast(T, (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
		else
			return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0));
	}
	__device__ __forceinline__ double wave_shl1_f64(double v) // lane i <- lane i + 1
	{
		const uint64_t b = __builtin_bit_cast(uint64_t, v);
		const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x130, 0xf, 0xf, true);
		const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x130, 0xf, 0xf, true);
		return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
	}

	// rows: rows of the image the translation applies to (h_ for translate; h_ - 3 for the read-back motion removal, whose
	// last rows are copied: copy_tail).  sign: -1 for the motion removal (it shifts by -x, -y).
	template <class T, class U>
	__global__ __launch_bounds__(256) void translate_tile_kernel(const T *__restrict__ src, U *__restrict__ dst, U background, int w, int h_,
																 const float *__restrict__ offsets, int per_frame_offsets, float sign, int strategy, int rows,
																 int copy_tail)
	{
		constexpr int OW = 63, OH = RIR_TR_TY;
		const int lane = threadIdx.x & 63;
		const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		// XCD-major tile order with the column index fastest: tiles that are neighbours along x split cache lines of the
		// same rows (63-pixel rows are not line-aligned) - on one XCD, one after the other, the second half of a line is
		// an L2 hit and the two halves of a written line merge before they leave for HBM
		int bx, by, n;
		{
			const unsigned gx = gridDim.x, gy = gridDim.y;
			const unsigned id2 = xcd_major(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
			bx = (int)(id2 % gx);
			by = (int)((id2 / gx) % gy);
			n = (int)(id2 / (gx * gy));
		}
		const int x0 = bx * OW, y0 = (by * 4 + wv) * OH;
		if (y0 >= h_)
			return;
		const int64_t fbase = (int64_t)n * w * h_;
		const T *s = src + fbase;
		U *d = dst + fbase;
		const float dx = sign * offsets[per_frame_offsets ? 2 * n : 0];
		const float dy = sign * offsets[per_frame_offsets ? 2 * n + 1 : 1];
		const bool small = fabsf(dx) < 1.0e9f && fabsf(dy) < 1.0e9f; // wave-uniform
		const int x = x0 + lane;
		const bool act_x = lane < OW && x < w;
		const int nrows = min(OH, rows - y0); // translated rows of this tile (<= 0: only copied rows)

		if (small && nrows > 0 && (int64_t)w * h_ * (int64_t)sizeof(T) < ((int64_t)1 << 31))
		{
			// columns (this lane), with the reference's expressions
			const float px = (float)x - dx;
			const bool out_x = px < 0 || px >= (float)w;
			const int l = out_x ? 0 : (int)px;
			int r = out_x ? 0 : (int)(px + 1.f);
			const bool r_is_l = r == w; // last source column: the right tap is the left one
			if (r_is_l)
				r = l;
			// the tile's source window starts at the column of lane 0's left tap; when lane 0 lies left of the source the
			// window is extrapolated from the first lane inside (columns < 0 then read as 0 and are never used)
			const uint64_t inside = __ballot(act_x && !out_x);
			const int first = inside ? __builtin_ctzll(inside) : 0;
			const int l0 = __builtin_amdgcn_readlane(l, first) - first;
			const bool col_ok = !act_x || out_x || (l == l0 + lane && (r_is_l || r == l + 1));
			// rows (lane j looks at output row y0 + j)
			const int yj = y0 + lane;
			const bool act_y = lane < nrows;
			const float pyj = (float)yj - dy;
			const bool out_yj = pyj < 0 || pyj >= (float)rows;
			const int tj = out_yj ? 0 : (int)pyj;
			const int bj = out_yj ? 0 : (int)(pyj + 1.f);
			// (rows above / below the source are handled in place too: the window is anchored on the first row inside)
			const uint64_t in_rows = __ballot(act_y && !out_yj);
			const int frow = in_rows ? __builtin_ctzll(in_rows) : 0;
			const int t0 = __builtin_amdgcn_readlane(tj, frow) - frow;
			const float vvj = (float)(bj == rows ? tj : bj) - pyj;
			const float vv0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vvj), frow));
			// regular rows: t = t0 + j, b = t + 1, one vertical weight; the row whose bottom tap falls on the last source row
			// (b == rows -> t, Filters.h:306-309; at most one per tile) has its own weight (float)t - py
			const bool is_last = bj == rows;
			const uint64_t last_rows = __ballot(act_y && !out_yj && is_last);
			const bool row_ok = !act_y || out_yj || (tj == t0 + lane && (is_last || (bj == tj + 1 && __builtin_bit_cast(int, vvj) == __builtin_bit_cast(int, vv0))));
			const bool x_border_ok = strategy == TRANSLATE_NEAREST || strategy == TRANSLATE_CONSTANT || strategy == TRANSLATE_UNCHANGED || strategy == TRANSLATE_SOURCE;
			const uint64_t outs = __ballot(act_x && out_x);
			const bool one_side = (outs & 1) == 0 || (outs >> first) == 0; // columns outside on the left OR on the right of the tile
			// rows outside the source: "nearest" takes source row 0 / rows - 1, which must be one of the OH + 1 rows loaded
			const uint64_t out_rows = __ballot(act_y && out_yj), above_rows = __ballot(act_y && pyj < 0);
			const int i_top = -t0, i_bot = rows - 1 - t0;
			const bool rows_fit = (above_rows == 0 || (i_top >= 0 && i_top <= OH)) && ((out_rows & ~above_rows) == 0 || (i_bot >= 0 && i_bot <= OH));
			if (inside != 0 && in_rows != 0 && __ballot(!(col_ok && row_ok)) == 0 && ((outs == 0 && out_rows == 0) || x_border_ok) && one_side && rows_fit &&
				l0 > -64 && t0 > -64 && ((last_rows >> frow) & 1) == 0)
			{
				const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)s);
				const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)s >> 32));
				const __amdgpu_buffer_rsrc_t rs =
					__builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, w * h_ * (int)sizeof(T), 0x00020000);
				const int col = l0 + lane;
				// (columns outside [0, w) would alias pixels of the neighbouring rows: they are sent out of range -> 0)
				const uint32_t voff = (col >= 0 && col < w) ? (uint32_t)((t0 * w + col) * (int)sizeof(T)) : 0x80000000u;
				const uint32_t step = (uint32_t)w * (uint32_t)sizeof(T);
				T tap[OH + 1];
				if (t0 >= 0 && t0 + OH < h_)
				{ // all OH + 1 rows are in the frame: the row steps on the scalar side (the scalar offset is NOT range-checked)
#pragma unroll
					for (int i = 0; i <= OH; ++i)
						tap[i] = buffer_load_px<T>(rs, voff, (uint32_t)i * step);
				}
				else
				{ // bottom of the frame: per-lane offsets, rows past the end read as 0 and are never used
					uint32_t vo = voff;
#pragma unroll
					for (int i = 0; i <= OH; ++i)
					{
						tap[i] = buffer_load_px<T>(rs, vo, 0);
						vo += step;
					}
				}
				double dl[OH + 1];
#pragma unroll
				for (int i = 0; i <= OH; ++i)
					dl[i] = (double)tap[i];
				const double u = (double)(px - (float)l), u1 = 1 - u;
				const double vv = (double)vv0, v1 = 1 - vv;
				U *o = d + (int64_t)y0 * w + x;
				if (outs == 0 && last_rows == 0 && out_rows == 0)
				{ // the plain tile
#pragma unroll
					for (int j = 0; j < OH; ++j)
					{
						const double cl = dl[j + 1] * v1 + dl[j] * vv;
						const double cs = wave_shl1_f64(cl);
						const double cr = r_is_l ? cl : cs;
						const U res = CastTo<U>::from(cl * u1 + cr * u);
						if (act_x && j < nrows)
						{
							// (plain stores: the streaming policy makes this kernel faster on its own - 0.099 -> 0.084 ms per 256
							// uint16 frames - but the frames are what the next kernel reads, and an encoder that finds them in the
							// Infinity Cache gains four times what the policy saves here)
							o[(int64_t)j * w] = res;
						}
					}
				}
				else
				{
					// edge tiles.  "nearest" left / right of the source: the value of the first / last source column, which one
					// lane of the wave holds
					const int edge_lane = outs ? ((outs & 1) ? -l0 : w - 1 - l0) : 0;
					if (outs != 0 && strategy == TRANSLATE_NEAREST && (edge_lane < 0 || edge_lane > 63))
						goto general;
					const int jl = last_rows ? __builtin_ctzll(last_rows) : -1;
					const double vvl = (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vvj), jl & 63));
					const double v1l = 1 - vvl;
#pragma unroll
					for (int j = 0; j < OH; ++j)
					{
						const bool last = j == jl; // wave-uniform
						const double cl = (last ? dl[j] : dl[j + 1]) * (last ? v1l : v1) + dl[j] * (last ? vvl : vv);
						const double cs = wave_shl1_f64(cl);
						const double cr = r_is_l ? cl : cs;
						U res = CastTo<U>::from(cl * u1 + cr * u);
						bool wr = act_x && j < nrows;
						const bool row_out = (out_rows >> j) & 1; // wave-uniform
						T own = tap[j]; // the lane's own column at the row "nearest" reads: t_j, or the first / last source row
						if (row_out)
						{
							const int ridx = ((above_rows >> j) & 1) ? i_top : i_bot;
#pragma unroll
							for (int i = 0; i <= OH; ++i)
								own = (i == ridx) ? tap[i] : own;
						}
						const bool px_out = out_x || row_out;
						if (strategy == TRANSLATE_NEAREST)
						{
							T e;
							if constexpr (sizeof(T) == 8)
							{
								const uint64_t b = __builtin_bit_cast(uint64_t, own);
								const uint32_t elo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, edge_lane & 63);
								const uint32_t ehi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), edge_lane & 63);
								e = __builtin_bit_cast(T, (uint64_t)elo | ((uint64_t)ehi << 32));
							}
							else
							{
								uint32_t b = 0;
								__builtin_memcpy(&b, &own, sizeof(T));
								b = (uint32_t)__builtin_amdgcn_readlane((int)b, edge_lane & 63);
								__builtin_memcpy(&e, &b, sizeof(T));
							}
							// columns outside: the clamped column (one lane holds it); inside, on a row outside: the lane's own column
							res = out_x ? tap_as<T, U>(e) : (row_out ? tap_as<T, U>(own) : res);
						}
						else if (strategy == TRANSLATE_CONSTANT)
							res = px_out ? background : res;
						else if (strategy == TRANSLATE_SOURCE)
						{ // the input pixel at the output position (clamped address: the load is unconditional)
							const T sp = s[(int64_t)min(y0 + j, h_ - 1) * w + min(x, w - 1)];
							res = px_out ? tap_as<T, U>(sp) : res;
						}
						else
							wr = wr && !px_out; // noborder: left untouched
						if (wr)
							o[(int64_t)j * w] = res;
					}
				}
				if (copy_tail && act_x)
					for (int y = max(y0, rows); y < min(y0 + OH, h_); ++y)
						d[(int64_t)y * w + x] = tap_as<T, U>(s[(int64_t)y * w + x]);
				return;
			}
		}
	general:
		// general path: pixel by pixel
		if (!act_x)
			return;
		for (int j = 0; j < OH; ++j)
		{
			const int y = y0 + j;
			if (y >= h_)
				break;
			if (y >= rows)
			{
				if (copy_tail)
					d[(int64_t)y * w + x] = tap_as<T, U>(s[(int64_t)y * w + x]);
				continue;
			}
			U val;
			const bool wr = small ? translate_px<T, U, true>(s, (uint64_t)w, (uint64_t)rows, (uint64_t)x, y, dx, dy, strategy, background, val)
								  : translate_px<T, U, false>(s, (uint64_t)w, (uint64_t)rows, (uint64_t)x, y, dx, dy, strategy, background, val);
			if (wr)
				d[(int64_t)y * w + x] = val;
		}
	}

	template <class T, class U>
	static hipError_t launch_translate_t(const void *src, void *dst, const void *background, int w, int h, int nframes, const float *d_offsets,
										 int per_frame, float sign, int strategy, int rows, hipStream_t st)
	{
		U back = *reinterpret_cast<const U *>(background);
		dim3 block(256), grid((unsigned)((w + 62) / 63), (unsigned)((rows + 4 * RIR_TR_TY - 1) / (4 * RIR_TR_TY)), nframes);
		hipLaunchKernelGGL((translate_tile_kernel<T, U>), grid, block, 0, st, (const T *)src, (U *)dst, back, w, h, d_offsets, per_frame, sign, strategy, rows,
						   0);
		return hipGetLastError();
	}

	hipError_t launch_translate(int type, const void *src, void *dst, const void *background, int w, int h, int nframes, const float *d_offsets,
								int per_frame, int strategy, hipStream_t st)
	{
		switch (type)
		{
		case '?':
			return launch_translate_t<bool, bool>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'b':
			return launch_translate_t<int8_t, int8_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'B':
			return launch_translate_t<uint8_t, uint8_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'h':
			return launch_translate_t<int16_t, int16_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'H':
			return launch_translate_t<uint16_t, uint16_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'i':
			return launch_translate_t<int32_t, int32_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'I':
			return launch_translate_t<uint32_t, uint32_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'l':
			return launch_translate_t<int64_t, int64_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'L':
			return launch_translate_t<uint64_t, uint64_t>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'f':
			return launch_translate_t<float, float>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'd':
			return launch_translate_t<double, double>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		case 'F': // float32 frames in, uint16 frames out ("translate, then astype(uint16)" in one pass); background is a uint16
			return launch_translate_t<float, u16_via_f32>(src, dst, background, w, h, nframes, d_offsets, per_frame, 1.f, strategy, h, st);
		default:
			return hipErrorInvalidValue;
		}
	}

	// ---- motion removal (read-back) -----------------------------------------------------------
	// translate<u16 -> float>(img, tmp, 0, w, rows, -x[pos], -y[pos], nearest) then the truncating float -> u16 copy
	// (IRFileLoader.cpp:617-627) is translate<u16 -> u16 through float>: the float never leaves registers.  Rows >= `rows`
	// are copied.
	hipError_t launch_remove_motion(const uint16_t *src, uint16_t *dst, int w, int h, int rows, int nframes, const float *d_shifts, hipStream_t st)
	{
		dim3 block(256), grid((unsigned)((w + 62) / 63), (unsigned)((h + 4 * RIR_TR_TY - 1) / (4 * RIR_TR_TY)), nframes);
		hipLaunchKernelGGL((translate_tile_kernel<uint16_t, u16_via_f32>), grid, block, 0, st, src, reinterpret_cast<u16_via_f32 *>(dst), u16_via_f32{0}, w, h,
						   d_shifts, 1, -1.f, (int)TRANSLATE_NEAREST, rows, 1);
		return hipGetLastError();
	}

	// ---- gaussian ------------------------------------------------------------------------------
	// The (2r+1)^2 table is built on the host with the reference's own float sequence and uploaded.
	// Accumulation order = dx outer, dy inner, res = res + k*src (two roundings), exactly as the host.
	// TIN = float, or uint16_t (the conversion of a u16 pixel to float is exact).  Any radius: the form radius > 4 always takes, and - with
	// the reference-order switch on (runtime.h: gaussian_reference_order) - every radius: bit-identical to the reference's loop, at the
	// price of (2r + 1)^2 taps per pixel instead of 2 (2r + 1).
	template <class TIN>
	__global__ __launch_bounds__(256) void gaussian_kernel(const TIN *__restrict__ src, float *__restrict__ dst, int w, int h,
														   const float *__restrict__ kern, int radius)
	{
		const int x = blockIdx.x * blockDim.x + threadIdx.x;
		const int y = blockIdx.y;
		const int n = blockIdx.z;
		if (x >= w)
			return;
		const int64_t fbase = (int64_t)n * w * h;
		const TIN *s = src + fbase;
		const int kw = 2 * radius + 1;
		if (x >= radius && x < w - radius && y >= radius && y < h - radius)
		{
			float res = 0;
			for (int dx = -radius; dx <= radius; ++dx)
				for (int dy = -radius; dy <= radius; ++dy)
				{
					const float p = __fmul_rn(kern[dx + radius + (dy + radius) * kw], (float)s[x + dx + (int64_t)(y + dy) * w]);
					res = __fadd_rn(res, p);
				}
			dst[fbase + x + (int64_t)y * w] = res;
		}
		else
		{
			float res = 0, sum = 0;
			for (int dx = -radius; dx <= radius; ++dx)
				for (int dy = -radius; dy <= radius; ++dy)
				{
					const int _x = x + dx, _y = y + dy;
					if (_x >= 0 && _x < w && _y >= 0 && _y < h)
					{
						const float k = kern[dx + radius + (dy + radius) * kw];
						sum = __fadd_rn(sum, k);
						res = __fadd_rn(res, __fmul_rn(k, (float)s[_x + (int64_t)_y * w]));
					}
				}
			dst[fbase + x + (int64_t)y * w] = __fdiv_rn(res, sum);
		}
	}

	// Radius 1..4: separable form, entirely in registers.  The reference table is k[dx][dy] = g(dx) g(dy) / sum, i.e.
	// the outer product of a 1-D factor a[d] (uploaded behind the 2-D table): a row pass and a column pass of
	// 2R+1 taps each replace the (2R+1)^2-tap sum, which turns the kernel from compute-bound (162 flop per
	// pixel at R = 4) into a streaming one.  Same mathematics as the reference, different rounding: results
	// agree to a few 1e-7 relative (the parity bar for float32 filters is 1e-5, BASELINE.json); border pixels
	// are renormalised by the in-image weight, Sx(x) * Sy(y), as the reference does with its 2-D sum.
	// Each WAVE works alone on a tile of 64 columns (the outer R on each side are halo) x TY rows: every lane
	// owns one column, fetches its TY + 2R values with back-to-back independent raw-buffer loads (row-coalesced
	// across the lanes; pixels outside the image read as 0 through the buffer's range check) and does the COLUMN
	// pass in registers with packed FMAs; the ROW pass takes the 2R neighbouring column sums from the neighbouring
	// lanes with DPP wave shifts.  No LDS, no barrier.  (History: the tile-staging version took 0.25 ms per 256
	// frames whatever the radius, set by its load / barrier / compute phases; the version with a wave-private LDS
	// strip for the row pass and loads under conditions 0.136 ms.)
#ifndef RIR_GAUSS_TY
#define RIR_GAUSS_TY 16 /* output rows per wave tile */
#endif
	// TIN = float, or uint16_t (the integer -> float conversion of a u16 frame folded into the load).
	typedef float v2f __attribute__((ext_vector_type(2)));
	__device__ __forceinline__ float wave_shl1(float f) // lane i <- lane i + 1 (0 into lane 63)
	{
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x130, 0xf, 0xf, true));
	}
	__device__ __forceinline__ float wave_shr1(float f) // lane i <- lane i - 1 (0 into lane 0)
	{
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x138, 0xf, 0xf, true));
	}

	template <int R, class TIN>
	__global__ __launch_bounds__(256) void gaussian_sep_kernel(const TIN *__restrict__ src, float *__restrict__ dst, int w, int h,
															   const float *__restrict__ kern)
	{
		constexpr int TY = RIR_GAUSS_TY, HY = TY / 2, KW = 2 * R + 1, OUTW = 64 - 2 * R, NR = TY + 2 * R;
		static_assert(TY % 2 == 0, "rows are processed in pairs (packed FMAs)");
		const int lane = threadIdx.x & 63;
		const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		// XCD-aware tile order (xcd_major), then (frame, column strip, row band) with the row band fastest:
		// vertically adjacent tiles, which share 2R halo rows, run on the same XCD back to back.
		int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
		{
			const unsigned gx = gridDim.x, gy = gridDim.y;
			const unsigned id2 = xcd_major(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
			by = (int)(id2 % gy);
			bx = (int)((id2 / gy) % gx);
			bz = (int)(id2 / (gy * gx));
		}
		const int x0 = bx * OUTW - R;
		const int x = x0 + lane;				 // this lane's column (halo lanes may fall outside the image)
		const int y0 = (by * 4 + wv) * TY;		 // first output row of this wave
		if (y0 >= h)
			return;
		const int64_t fbase = (int64_t)bz * w * h;
		const TIN *s = src + fbase;
		float a[KW];
#pragma unroll
		for (int d = 0; d < KW; ++d)
			a[d] = kern[KW * KW + d];
		const bool xin = x >= 0 && x < w;
		float v[NR];
		if ((int64_t)w * h * (int64_t)sizeof(TIN) < (int64_t)1 << 31)
		{
			// Raw-buffer loads over the frame: a pixel outside the image has a byte offset outside the buffer (rows above:
			// negative = huge unsigned; rows below: past the end; columns outside: forced there) and reads as 0 - exactly
			// what such a pixel adds to the sums.  No branch, no select, and all TY + 2R loads are in flight together
			// (a load under a condition becomes a branch with its own wait: that many serial latencies).
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)s);
			const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)s >> 32));
			const __amdgpu_buffer_rsrc_t rs =
				__builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, w * h * (int)sizeof(TIN), 0x00020000);
			const uint32_t step = (uint32_t)w * (uint32_t)sizeof(TIN);
			if (x0 >= 0 && x0 + 63 < w && y0 - R >= 0 && y0 - R + NR <= h)
			{ // block inside the image: one lane offset, the row steps on the scalar side
				const uint32_t off = (uint32_t)(((y0 - R) * w + x) * (int)sizeof(TIN));
#pragma unroll
				for (int i = 0; i < NR; ++i)
				{
					if constexpr (sizeof(TIN) == 2)
						v[i] = (float)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, (int)((uint32_t)i * step), 0);
					else
						v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, (int)((uint32_t)i * step), 0));
				}
			}
			else
			{
				uint32_t off = xin ? (uint32_t)(((y0 - R) * w + x) * (int)sizeof(TIN)) : 0x80000000u;
#pragma unroll
				for (int i = 0; i < NR; ++i)
				{
					if constexpr (sizeof(TIN) == 2)
						v[i] = (float)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
					else
						v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
					off += step;
				}
			}
		}
		else
		{ // frames of 2 GiB and more: plain loads at addresses clamped into the frame, values outside dropped afterwards
			const TIN *col = s + min(max(x, 0), w - 1);
#pragma unroll
			for (int i = 0; i < NR; ++i)
			{
				const int gy = y0 - R + i;
				const float val = (float)col[(int64_t)min(max(gy, 0), h - 1) * w];
				v[i] = (xin && gy >= 0 && gy < h) ? val : 0.f; // zeros outside the image add nothing
			}
		}
		// column pass, in registers: rows j and j + TY/2 side by side in packed FMAs (two IEEE fmas per instruction)
		float cs[TY];
#pragma unroll
		for (int j = 0; j < HY; ++j)
		{
			v2f acc = {0.f, 0.f};
#pragma unroll
			for (int d = 0; d < KW; ++d)
				acc = __builtin_elementwise_fma((v2f){a[d], a[d]}, (v2f){v[j + d], v[j + HY + d]}, acc);
			cs[j] = acc.x;
			cs[j + HY] = acc.y;
		}
		// row pass: the 2R neighbouring column sums come from the neighbouring LANES through DPP wave shifts - no LDS,
		// no barrier (all 64 lanes are still active here: the shifts read every lane)
		float g[TY];
#pragma unroll
		for (int j = 0; j < HY; ++j)
		{
			v2f t[KW]; // t[d] = column sums of lane - R + d
			t[R] = (v2f){cs[j], cs[j + HY]};
#pragma unroll
			for (int d = 1; d <= R; ++d)
			{
				t[R - d] = (v2f){wave_shr1(t[R - d + 1].x), wave_shr1(t[R - d + 1].y)};
				t[R + d] = (v2f){wave_shl1(t[R + d - 1].x), wave_shl1(t[R + d - 1].y)};
			}
			v2f acc = {0.f, 0.f};
#pragma unroll
			for (int d = 0; d < KW; ++d)
				acc = __builtin_elementwise_fma((v2f){a[d], a[d]}, t[d], acc);
			g[j] = acc.x;
			g[j + HY] = acc.y;
		}
		if (lane < R || lane >= 64 - R || x >= w)
			return; // halo lanes, and columns past the right edge, have no output
		// in-image weight of the row taps of this column, and the full 1-D sum (not exactly 1)
		float full = 0.f;
#pragma unroll
		for (int d = 0; d < KW; ++d)
			full += a[d];
		const bool xb = x < R || x >= w - R;
		float sx = full;
		if (xb)
		{
			sx = 0.f;
#pragma unroll
			for (int d = -R; d <= R; ++d)
				if (x + d >= 0 && x + d < w)
					sx += a[d + R];
		}
		float *o = dst + fbase + x + (int64_t)y0 * w;
#pragma unroll
		for (int j = 0; j < TY; ++j)
		{
			const int y = y0 + j;
			if (y >= h)
				break;
			float acc = g[j];
			const bool yb = y < R || y >= h - R;
			if (xb || yb)
			{ // border pixels are renormalised by the weight of the taps that fall inside the image
				float sy = full;
				if (yb)
				{
					sy = 0.f;
#pragma unroll
					for (int d = -R; d <= R; ++d)
						if (y + d >= 0 && y + d < h)
							sy += a[d + R];
				}
				acc = acc / (sx * sy);
			}
			o[(int64_t)j * w] = acc;
		}
	}

	// The reference's own 2-D sum (signal_processing.cpp:101-148) with the tiling of the separable kernel above: a wave owns 64 columns
	// (R of halo on each side) x TY rows, a lane one column - its TY + 2R pixels in registers, loaded once; the columns x + dx come from the
	// neighbouring lanes through DPP wave shifts.  For every output pixel the products are added in the reference's order - dx outer, dy
	// inner, one rounding per product and one per sum (no fused multiply-add: the reference's x86-64 build has none) - so the result is the
	// reference's bit for bit.  Pixels closer than R to the image's edge divide by the sum of the weights that fall inside the image,
	// accumulated in the same order (a tap outside the image adds k x 0 to the one and + 0 to the other: what skipping it adds).
	template <int R, class TIN>
	__global__ __launch_bounds__(256) void gaussian_exact_kernel(const TIN *__restrict__ src, float *__restrict__ dst, int w, int h, const float *__restrict__ kern)
	{
		constexpr int TY = RIR_GAUSS_TY, KW = 2 * R + 1, OUTW = 64 - 2 * R, NR = TY + 2 * R;
		const int lane = threadIdx.x & 63;
		const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
		{
			const unsigned gx = gridDim.x, gy = gridDim.y;
			const unsigned id2 = xcd_major(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
			by = (int)(id2 % gy);
			bx = (int)((id2 / gy) % gx);
			bz = (int)(id2 / (gy * gx));
		}
		const int x0 = bx * OUTW - R;
		const int x = x0 + lane;
		const int y0 = (by * 4 + wv) * TY;
		if (y0 >= h)
			return;
		const int64_t fbase = (int64_t)bz * w * h;
		const TIN *s = src + fbase;
		const bool xin = x >= 0 && x < w;
		float v[NR];
		if ((int64_t)w * h * (int64_t)sizeof(TIN) < (int64_t)1 << 31)
		{ // raw-buffer loads: a pixel outside the image has an offset outside the buffer and reads as 0 (gaussian_sep_kernel)
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)s);
			const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)s >> 32));
			const __amdgpu_buffer_rsrc_t rs =
				__builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, w * h * (int)sizeof(TIN), 0x00020000);
			const uint32_t step = (uint32_t)w * (uint32_t)sizeof(TIN);
			uint32_t off = xin ? (uint32_t)(((y0 - R) * w + x) * (int)sizeof(TIN)) : 0x80000000u;
#pragma unroll
			for (int i = 0; i < NR; ++i)
			{
				if constexpr (sizeof(TIN) == 2)
					v[i] = (float)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
				else
					v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
				off += step;
			}
		}
		else
		{
			const TIN *col = s + min(max(x, 0), w - 1);
#pragma unroll
			for (int i = 0; i < NR; ++i)
			{
				const int gy = y0 - R + i;
				const float val = (float)col[(int64_t)min(max(gy, 0), h - 1) * w];
				v[i] = (xin && gy >= 0 && gy < h) ? val : 0.f;
			}
		}
		// every output pixel of the tile an interior pixel of the image?  (wave-uniform: then no weight sums are needed)
		const bool interior = x0 >= 0 && x0 + 63 - R < w - R && y0 >= R && y0 + TY - 1 < h - R;
		float res[TY], sum[TY];
#pragma unroll
		for (int j = 0; j < TY; ++j)
			res[j] = 0.f, sum[j] = 0.f;
		float c[NR];
		float cin = xin ? 1.f : 0.f; // is the column a tap comes from inside the image (travels with the shifts)
		auto taps = [&](int dx) {
#pragma unroll
			for (int dy = -R; dy <= R; ++dy)
			{
				const float k = kern[dx + R + (dy + R) * KW];
#pragma unroll
				for (int j = 0; j < TY; ++j)
					res[j] = __fadd_rn(res[j], __fmul_rn(k, c[j + dy + R]));
				if (!interior)
				{
#pragma unroll
					for (int j = 0; j < TY; ++j)
					{
						const int yy = y0 + j + dy;
						sum[j] = __fadd_rn(sum[j], (cin != 0.f && yy >= 0 && yy < h) ? k : 0.f);
					}
				}
			}
		};
		// dx = -R .. 0: the column R lanes to the left, then one lane closer per step (lane i holds column x + dx of ITS x; the values
		// shifted in at the wave's ends only ever reach halo lanes)
#pragma unroll
		for (int i = 0; i < NR; ++i)
		{
			c[i] = v[i];
#pragma unroll
			for (int d = 0; d < R; ++d)
				c[i] = wave_shr1(c[i]);
		}
		const float cin0 = cin;
#pragma unroll
		for (int d = 0; d < R; ++d)
			cin = wave_shr1(cin);
#pragma unroll
		for (int dx = -R; dx <= 0; ++dx)
		{
			taps(dx);
			if (dx < 0)
			{
#pragma unroll
				for (int i = 0; i < NR; ++i)
					c[i] = wave_shl1(c[i]);
				cin = wave_shl1(cin);
			}
		}
		// dx = 1 .. R: from the lane's own column to the right
#pragma unroll
		for (int i = 0; i < NR; ++i)
			c[i] = v[i];
		cin = cin0;
#pragma unroll
		for (int dx = 1; dx <= R; ++dx)
		{
#pragma unroll
			for (int i = 0; i < NR; ++i)
				c[i] = wave_shl1(c[i]);
			cin = wave_shl1(cin);
			taps(dx);
		}
		if (lane < R || lane >= 64 - R || x >= w)
			return; // halo lanes, and columns past the right edge, have no output
		const bool xb = x < R || x >= w - R;
		float *o = dst + fbase + x + (int64_t)y0 * w;
#pragma unroll
		for (int j = 0; j < TY; ++j)
		{
			const int y = y0 + j;
			if (y >= h)
				break;
			const bool border = xb || y < R || y >= h - R;
			o[(int64_t)j * w] = border ? __fdiv_rn(res[j], sum[j]) : res[j];
		}
	}

	template <class TIN>
	static bool launch_gaussian_sep(const TIN *src, float *dst, int w, int h, int nframes, const float *d_kernel, int radius, hipStream_t st)
	{
		if (radius < 1 || radius > 4)
			return false;
		const int outw = 64 - 2 * radius;
		dim3 block(256), tgrid((w + outw - 1) / outw, (h + 4 * RIR_GAUSS_TY - 1) / (4 * RIR_GAUSS_TY), nframes);
		switch (radius)
		{
		case 1:
			hipLaunchKernelGGL((gaussian_sep_kernel<1, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		case 2:
			hipLaunchKernelGGL((gaussian_sep_kernel<2, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		case 3:
			hipLaunchKernelGGL((gaussian_sep_kernel<3, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		default:
			hipLaunchKernelGGL((gaussian_sep_kernel<4, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		}
	}

	template <class TIN>
	static bool launch_gaussian_exact(const TIN *src, float *dst, int w, int h, int nframes, const float *d_kernel, int radius, hipStream_t st)
	{
		if (radius < 1 || radius > 4)
			return false;
		const int outw = 64 - 2 * radius;
		dim3 block(256), tgrid((w + outw - 1) / outw, (h + 4 * RIR_GAUSS_TY - 1) / (4 * RIR_GAUSS_TY), nframes);
		switch (radius)
		{
		case 1:
			hipLaunchKernelGGL((gaussian_exact_kernel<1, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		case 2:
			hipLaunchKernelGGL((gaussian_exact_kernel<2, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		case 3:
			hipLaunchKernelGGL((gaussian_exact_kernel<3, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		default:
			hipLaunchKernelGGL((gaussian_exact_kernel<4, TIN>), tgrid, block, 0, st, src, dst, w, h, d_kernel);
			return true;
		}
	}

	hipError_t launch_gaussian(const float *src, float *dst, int w, int h, int nframes, const float *d_kernel, int radius, hipStream_t st)
	{
		if (gaussian_reference_order() ? !launch_gaussian_exact<float>(src, dst, w, h, nframes, d_kernel, radius, st)
									   : !launch_gaussian_sep<float>(src, dst, w, h, nframes, d_kernel, radius, st))
		{ // radius > 4: the direct 2-D form, one thread per pixel (the reference's order too)
			dim3 block(256), grid((w + 255) / 256, h, nframes);
			hipLaunchKernelGGL(gaussian_kernel<float>, grid, block, 0, st, src, dst, w, h, d_kernel, radius);
		}
		return hipGetLastError();
	}

	// uint16 frames in, float out (radius <= 4 only: hipErrorInvalidValue otherwise)
	hipError_t launch_gaussian_u16(const uint16_t *src, float *dst, int w, int h, int nframes, const float *d_kernel, int radius, hipStream_t st)
	{
		if (radius < 1 || radius > 4)
			return hipErrorInvalidValue;
		if (gaussian_reference_order())
			launch_gaussian_exact<uint16_t>(src, dst, w, h, nframes, d_kernel, radius, st);
		else
			launch_gaussian_sep<uint16_t>(src, dst, w, h, nframes, d_kernel, radius, st);
		return hipGetLastError();
	}

	// ---- small sorted-window helpers ---------------------------------------------------------

	// value that std::nth_element(p, p+c/2, p+c) leaves at index c/2: the element of rank c/2
	// (rank = number of smaller elements, ties broken by index).  c <= 9.
	__device__ __forceinline__ uint16_t rank_select9(const uint32_t *v, int c)
	{
		const int want = c >> 1;
		uint32_t res = 0;
#pragma unroll
		for (int i = 0; i < 9; ++i)
		{
			int rank = 0;
#pragma unroll
			for (int j = 0; j < 9; ++j)
				rank += (j < c) && ((v[j] < v[i]) || (v[j] == v[i] && j < i));
			if (i < c && rank == want)
				res = v[i];
		}
		return (uint16_t)res;
	}

	// ---- bad pixel correction (BadPixels::correct) ------------------------------------------
	// pass 1: out = max(in, floor)   (copy + clampMin fused; 16 B per lane)
	__global__ __launch_bounds__(256) void clamp_copy_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int64_t total, uint32_t floor_v)
	{
		const int64_t i8 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
		if (i8 >= total)
			return;
		const uint32_t f2 = floor_v | (floor_v << 16);
		if (i8 + 8 <= total && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0)
		{
#ifndef RIR_CLAMP_COPY_NT
#define RIR_CLAMP_COPY_NT 2 /* 1: non-temporal load, 2: non-temporal store, 3: both.  Measured (256 frames 640x512): store alone 0.069 -> 0.0635 ms, load alone or both slower (the repair launch reads the neighbours of the flagged pixels from the cache) */
#endif
			typedef unsigned int clamp_v4u __attribute__((ext_vector_type(4)));
			const clamp_v4u *src4 = reinterpret_cast<const clamp_v4u *>(in + i8);
			clamp_v4u v = (RIR_CLAMP_COPY_NT & 1) ? __builtin_nontemporal_load(src4) : *src4;
#pragma unroll
			for (int k = 0; k < 4; ++k)
			{
				const uint32_t lo = max(v[k] & 0xffffu, f2 & 0xffffu), hi = max(v[k] >> 16, f2 >> 16);
				v[k] = lo | (hi << 16);
			}
			clamp_v4u *dst4 = reinterpret_cast<clamp_v4u *>(out + i8);
			if (RIR_CLAMP_COPY_NT & 2)
				__builtin_nontemporal_store(v, dst4);
			else
				*dst4 = v;
		}
		else
		{
			for (int64_t i = i8; i < total && i < i8 + 8; ++i)
				out[i] = (uint16_t)max((uint32_t)in[i], floor_v);
		}
	}

	// pass 2: one thread per (flagged pixel, frame): upper median of the in-bounds 3x3 (centre
	// included, gathered from `in`), then the clamp.  xy = int32 pairs.
	// table != NULL: the repaired values go to table[n][i] (uint32 each) instead of the frame - the fused filter chain
	// below patches them into its tiles.
	__global__ __launch_bounds__(64) void bad_pixels_fix_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int w, int h,
																const int *__restrict__ xy, int nbad, uint32_t floor_v, uint32_t *__restrict__ table)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		const int n = blockIdx.y;
		if (i >= nbad)
			return;
		const int64_t fbase = (int64_t)n * w * h;
		const int x = xy[2 * i], y = xy[2 * i + 1];
		uint32_t v[9];
		int c = 0;
#pragma unroll
		for (int ddx = -1; ddx <= 1; ++ddx)
#pragma unroll
			for (int ddy = -1; ddy <= 1; ++ddy)
			{
				const int xx = x + ddx, yy = y + ddy;
				const bool ok = xx >= 0 && yy >= 0 && xx < w && yy < h;
				// (always loaded, from an address clamped into the frame: nine independent loads instead of nine branches)
				const uint32_t val = in[fbase + min(max(xx, 0), w - 1) + (int64_t)min(max(yy, 0), h - 1) * w];
				// compact the in-bounds values to the front, keeping gather order
#pragma unroll
				for (int k = 0; k < 9; ++k)
					if (ok && k == c)
						v[k] = val;
				c += ok;
			}
#pragma unroll
		for (int k = 0; k < 9; ++k)
			if (k >= c)
				v[k] = 0xffffffffu;
		const uint32_t m = rank_select9(v, c);
		if (table)
			table[(int64_t)n * nbad + i] = (uint32_t)(uint16_t)max(m, floor_v);
		else
			out[fbase + x + (int64_t)y * w] = (uint16_t)max(m, floor_v);
	}

	hipError_t launch_bad_pixels_correct(const uint16_t *in, uint16_t *out, int w, int h, int nframes, const int *d_xy, int nbad, int floor_v,
										 hipStream_t st)
	{
		const uint32_t fl = floor_v > 0 ? (uint32_t)(uint16_t)floor_v : 0u;
		const int64_t total = (int64_t)w * h * nframes;
		if (in != out || fl > 0)
		{
			const int64_t threads = (total + 7) / 8;
			hipLaunchKernelGGL(clamp_copy_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, in, out, total, fl);
		}
		if (nbad > 0)
			hipLaunchKernelGGL(bad_pixels_fix_kernel, dim3((nbad + 63) / 64, nframes), dim3(64), 0, st, in, out, w, h, d_xy, nbad, fl, (uint32_t *)nullptr);
		return hipGetLastError();
	}

	// ---- fused filter chain: bad_pixels_correct -> gaussian_filter -> translate, one pass over the frame ------
	// The chain of BASELINE configs[2] run as three kernels writes a uint16 frame, reads it, writes a
	// float frame, reads it with four taps per pixel and writes the final uint16 frame: 14 bytes per pixel of HBM
	// traffic for an algorithmic 4 (uint16 in, uint16 out).  Here a WAVE produces a tile of OW x OH output pixels
	// on its own: it finds the block of gaussian-filtered pixels the tile's bilinear taps fall on (translation is
	// uniform, so that block is the tile shifted by (dx, dy), at most OW+2 x OH+2 pixels), loads the raw pixels
	// under it (+R halo) straight into registers with the clamp of BadPixels::correct applied, patches the repaired
	// values of flagged pixels in (a table filled by bad_pixels_fix_kernel beforehand: ~200 values per frame), runs
	// the separable gaussian exactly as gaussian_sep_kernel does - same column sums, same row sums, same border
	// renormalisation, so every filtered value has the bits the stand-alone kernel would have stored - keeps the
	// result in a wave-private LDS tile and evaluates translate_px's expressions on it.  The output is bit-identical
	// to the three-kernel chain.  Strategies whose border pixels are not local to the tile (wrap, noborder) are not
	// offered; a tap outside the wave's block (possible only through float rounding of px + 1) is recomputed from
	// global memory by chain_gauss_point.
// Store policy of the output tiles.  Their rows are 60 (median: 63) pixels long, so neighbouring tiles - other waves, other
// CUs - share 128-byte lines.  With the streaming policy (aux = 2) every partial line went to memory on its own: WRITE_SIZE
// 244 MB for 168 MB of output (1.45x), and the encoder that reads the frames next found nothing cached.  With the default
// policy the lines are completed in L2: 172 MB (1.02x); the kernel alone is 3 % slower (0.149 vs 0.145 ms per 256 frames),
// chain + encode 3 % faster (0.198 vs 0.204 ms) - scripts/chain_variants.sh, profiles/r02_pmc_filters.json.
#ifndef RIR_CHAIN_STORE_AUX
#define RIR_CHAIN_STORE_AUX 0
#endif
#ifndef RIR_CHAIN_TY
#define RIR_CHAIN_TY 16 /* rows of the gaussian block per wave; OH = TY - 2 output rows */
#endif
	typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
	struct ChainBadPixels
	{
		const int *xy;			  // flagged (x, y) pairs, raster order
		const int *row_start;	  // [h + 1]: index of the first flagged pixel of row >= y
		const uint32_t *fix;	  // [nframes][nbad] repaired values
		int nbad;
		uint32_t floor_v;		  // clamp floor (0 = none)
	};

	template <int R>
	__device__ __noinline__ float chain_gauss_point(const uint16_t *__restrict__ s, int w, int h, int c, int r, ChainBadPixels bp, int64_t fix_base,
													const float *__restrict__ kern)
	{
		constexpr int KW = 2 * R + 1;
		if (c < 0 || c >= w || r < 0 || r >= h)
			return 0.f;
		float a[KW];
#pragma unroll
		for (int d = 0; d < KW; ++d)
			a[d] = kern[KW * KW + d];
		float cs[KW];
#pragma unroll
		for (int dc = 0; dc < KW; ++dc)
		{
			const int col = c - R + dc;
			float acc = 0.f;
#pragma unroll
			for (int d = 0; d < KW; ++d)
			{
				const int rr = r - R + d;
				float val = 0.f;
				if (col >= 0 && col < w && rr >= 0 && rr < h)
				{
					uint32_t p = max((uint32_t)s[col + (int64_t)rr * w], bp.floor_v);
					if (bp.nbad > 0)
						for (int i = bp.row_start[rr]; i < bp.row_start[rr + 1]; ++i)
							if (bp.xy[2 * i] == col)
								p = bp.fix[fix_base + i];
					val = (float)p;
				}
				acc = fmaf(a[d], val, acc);
			}
			cs[dc] = acc;
		}
		float acc = 0.f;
#pragma unroll
		for (int d = 0; d < KW; ++d)
			acc = fmaf(a[d], cs[d], acc);
		float full = 0.f;
#pragma unroll
		for (int d = 0; d < KW; ++d)
			full += a[d];
		const bool xb = c < R || c >= w - R, yb = r < R || r >= h - R;
		if (xb || yb)
		{
			float sx = full, sy = full;
			if (xb)
			{
				sx = 0.f;
#pragma unroll
				for (int d = -R; d <= R; ++d)
					if (c + d >= 0 && c + d < w)
						sx += a[d + R];
			}
			if (yb)
			{
				sy = 0.f;
#pragma unroll
				for (int d = -R; d <= R; ++d)
					if (r + d >= 0 && r + d < h)
						sy += a[d + R];
			}
			acc = acc / (sx * sy);
		}
		return acc;
	}

	// first source column / row an output column / row can touch (monotone in x / y)
	__device__ __forceinline__ int chain_anchor(int x, float d, int size)
	{
		const float p = (float)x - d;
		return p < 0 ? 0 : (p >= (float)size ? size - 1 : (int)p);
	}

	// One wave's tile of the chain: output pixels [x0, x0 + OW) x [y0, y0 + OH) of frame n, x0 = bx * OW, y0 = byw * OH.
	// MODE 0: both paths below in one kernel (radius 2..4).  Radius 1 - the configuration the chain is used with, sigma 0.75 -
	// is two kernels: MODE 1, the REGULAR tiles only, entirely in registers (39 VGPRs: 8 waves per SIMD, no scratch); a tile that
	// is not regular (first / last row bands of the image, float rounding of px + 1, ...) is put on a list instead (one 64-bit
	// entry, n << 32 | byw << 16 | bx) and MODE 2, the general path, works the list off.  As one kernel the general path's
	// registers (10 spilled at the regular path's occupancy) cost every wave its scratch set-up and the regular path a wave
	// per SIMD.  `tile`: the wave's LDS strip [TY][64].  MODE 2 never takes the regular path (the list holds what it refused).
	template <int R, int MODE>
	__device__ __forceinline__ void chain_tile(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int w, int h, const ChainBadPixels &bp,
												const float *__restrict__ kern, const float *__restrict__ offsets, int per_frame_offsets, int strategy,
												uint32_t background, int bx, int byw, int n, int lane, float (*tile_w)[64], unsigned long long *__restrict__ worklist,
												unsigned int *__restrict__ work_count)
	{
		constexpr int TY = RIR_CHAIN_TY, KW = 2 * R + 1, OUTW = 64 - 2 * R, OW = OUTW - 2, OH = TY - 2, NR = TY + 2 * R;
		static_assert(TY % 2 == 0 && (R != 1 || OW % 4 == 0), "rows in pairs (packed FMAs); 4-pixel output pieces in the register path");
		static_assert(MODE == 0 || R == 1, "the regular / listed split exists for radius 1");
		const int x0 = bx * OW, y0 = byw * OH;
		if (y0 >= h)
			return;
		const int64_t fbase = (int64_t)n * w * h;
		const uint16_t *s = src + fbase;
		const float dx = offsets[per_frame_offsets ? 2 * n : 0];
		const float dy = offsets[per_frame_offsets ? 2 * n + 1 : 1];
		int gx0 = chain_anchor(x0, dx, w);
		const int gy0 = chain_anchor(y0, dy, h); // (gx0, gy0): top-left filtered pixel of the block
		{ // a tile whose first columns fall left of the source keeps the anchor its own columns would have (negative): lane i's
		  // taps stay at gx0 + i, the columns < 0 of the block read as 0 and the clamped taps (column 0) are still inside it
			const float p0 = (float)x0 - dx;
			if (p0 < 0.f && p0 > -61.f)
				gx0 = (int)floorf(p0);
		}
		const int cx = gx0 - R + lane;											 // this lane's source column
		const bool xin = cx >= 0 && cx < w;
		const int64_t fix_base = (int64_t)n * bp.nbad;

		float a[KW];
#pragma unroll
		for (int d = 0; d < KW; ++d)
			a[d] = kern[KW * KW + d];
		// raw pixels of the column, clamped like BadPixels::correct does (Filters.cpp:7-50)
		// The run of the flagged-pixel list that lies under the block's rows (the list is in raster order): its bounds, and
		// its first 64 entries - one per lane, position and repaired value - are fetched here, ahead of the pixels, so
		// that their latency overlaps the pixel loads instead of following them.  Buffer loads: with no flagged pixels
		// the buffers are empty and everything reads as 0 (an empty run) without a branch.
		int li0, li1;
		int2 lf;
		uint32_t lfix;
		{
			const __amdgpu_buffer_rsrc_t r_rows = __builtin_amdgcn_make_buffer_rsrc((void *)bp.row_start, 0, bp.nbad > 0 ? (h + 1) * 4 : 0, 0x00020000);
			const __amdgpu_buffer_rsrc_t r_xy = __builtin_amdgcn_make_buffer_rsrc((void *)bp.xy, 0, bp.nbad * 8, 0x00020000);
			const __amdgpu_buffer_rsrc_t r_fix = __builtin_amdgcn_make_buffer_rsrc((void *)(bp.fix + fix_base), 0, bp.nbad * 4, 0x00020000);
			const int ra = max(gy0 - R, 0), rb = min(gy0 - R + NR, h);
			li0 = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_raw_buffer_load_b32(r_rows, ra * 4, 0, 0));
			li1 = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_raw_buffer_load_b32(r_rows, rb * 4, 0, 0));
			const int i = li0 + lane;
			const uint32_t eo = i < li1 ? (uint32_t)i : 0x10000000u; // (out of range -> zeros)
			const v2u32 e = __builtin_amdgcn_raw_buffer_load_b64(r_xy, (int)(eo * 8u), 0, 0);
			lf = i < li1 ? make_int2((int)e.x, (int)e.y) : make_int2(-(1 << 30), 0);
			lfix = __builtin_amdgcn_raw_buffer_load_b32(r_fix, (int)(eo * 4u), 0, 0);
		}
		float v[NR];
		if ((int64_t)w * h < (1 << 30))
		{
			// Raw-buffer loads over the frame: the byte offset of a pixel outside the image is out of the buffer's range
			// (rows above: negative, i.e. >= 2^31 unsigned; rows below: past the end; columns outside: forced there) and
			// the hardware returns 0 for it - the value such a pixel contributes.  One add per row, no branch, no select.
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)s);
			const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)s >> 32));
			const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, w * h * 2, 0x00020000);
			const uint32_t step = (uint32_t)w * 2u;
			uint32_t raw[NR];
			if (gx0 - R >= 0 && gx0 - R + 63 < w && gy0 - R >= 0 && gy0 - R + NR <= h)
			{ // the usual case, every pixel under the block is in the image: one lane offset, the row steps on the scalar side
				const uint32_t off = (uint32_t)(((gy0 - R) * w + cx) * 2);
#pragma unroll
				for (int i = 0; i < NR; ++i)
					raw[i] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, (int)((uint32_t)i * step), 0);
#pragma unroll
				for (int i = 0; i < NR; ++i)
					v[i] = (float)max(raw[i], bp.floor_v);
			}
			else
			{
				uint32_t off = xin ? (uint32_t)(((gy0 - R) * w + cx) * 2) : 0x80000000u;
#pragma unroll
				for (int i = 0; i < NR; ++i)
				{
					raw[i] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
					off += step;
				}
#pragma unroll
				for (int i = 0; i < NR; ++i)
				{ // (the clamp floor must not lift the zeros of pixels outside the image)
					const int gy = gy0 - R + i;
					v[i] = (float)((xin && gy >= 0 && gy < h) ? max(raw[i], bp.floor_v) : 0u);
				}
			}
		}
		else
		{ // frames of 2 GiB and more: plain loads at addresses clamped into the frame, values outside dropped afterwards
			const uint16_t *col = s + min(max(cx, 0), w - 1);
#pragma unroll
			for (int i = 0; i < NR; ++i)
			{
				const int gy = gy0 - R + i;
				const float val = (float)max((uint32_t)col[(int64_t)min(max(gy, 0), h - 1) * w], bp.floor_v);
				v[i] = (xin && gy >= 0 && gy < h) ? val : 0.f;
			}
		}
		// flagged pixels under the block take their repaired value (first 64 entries of the run: fetched above)
		{
			const int ridx0 = gy0 - R;
			for (int base = li0; base < li1; base += 64)
			{
				if (base != li0)
				{ // (more than 64 flagged pixels in the rows of one block: rare)
					const int i = base + lane;
					const bool has = i < li1;
					lf = has ? reinterpret_cast<const int2 *>(bp.xy)[i] : make_int2(-(1 << 30), 0);
					lfix = has ? bp.fix[fix_base + i] : 0u;
				}
				const int rel = lf.x - (gx0 - R);
				uint64_t m = __ballot(base + lane < li1 && rel >= 0 && rel < 64);
				while (m)
				{
					const int k = __builtin_ctzll(m);
					m &= m - 1;
					const int rel_k = __builtin_amdgcn_readlane(rel, k);
					const int ridx = __builtin_amdgcn_readlane(lf.y, k) - ridx0;
					const float fv = (float)(uint32_t)__builtin_amdgcn_readlane((int)lfix, k);
#pragma unroll
					for (int q = 0; q < NR; ++q)
						v[q] = (lane == rel_k && q == ridx) ? fv : v[q];
				}
			}
		}
		// gaussian, column pass: in registers, rows j and j + TY/2 side by side in packed FMAs (two IEEE fmas per
		// instruction: the same values as the scalar form, half the instructions)
		// (two rows per packed FMA; which two rows share an instruction does not matter for the values: the regular tiles below take
		// NEIGHBOURING rows, 2p and 2p + 1, and consume them at once - column pass, row pass and blend of a pair of rows before the next
		// pair is touched, so that no array of column sums or filtered pixels is ever held: 8 waves per SIMD without scratch)
		constexpr int HY = TY / 2;
		auto column_pair = [&](int r0, int r1) -> v2f {
			v2f acc = {0.f, 0.f};
#pragma unroll
			for (int d = 0; d < KW; ++d)
				acc = __builtin_elementwise_fma((v2f){a[d], a[d]}, (v2f){v[r0 + d], v[r1 + d]}, acc);
			return acc;
		};
		// this lane's output column and its taps (expressions of translate_px / Filters.h:249-326)
		const int x = x0 + lane;
		const bool act_x = lane < OW && x < w;
		const float px = (float)x - dx;
		const bool out_x = px < 0 || px >= (float)w;
		const int l = out_x ? (px < 0 ? 0 : w - 1) : (int)px;
		int r = out_x ? l : (int)(px + 1.f);
		const bool r_is_l = !out_x && r >= w; // last source column: the right tap is the left one
		if (r >= w)
			r = l;
		const double u = (double)(px - (float)l);

		if constexpr (R == 1 && MODE != 2)
		{
			// ---- regular tiles (all but the first / last row bands of the image): no LDS at all -------------------------
			// When the block does not touch the first / last R rows, lane i's taps are the
			// columns gx0 + i and gx0 + i + 1 (no clamp, px + 1 not rounded across an integer) and output row j's taps
			// are the rows gy0 + j and gy0 + j + 1, then every operand sits in a NEIGHBOUR lane's registers: the row
			// pass of the gaussian takes its left / right column sums through DPP wave shifts, and so do the two tap
			// columns of the blend; the vertical blend of the right column IS the left-column blend of lane i + 1.
			// Same expressions, same order, same bits as the general path below.
			const int yj = y0 + lane; // lane j looks at output row j
			const bool act_y = lane < OH && yj < h;
			const float pyj = (float)yj - dy;
			const bool out_yj = pyj < 0 || pyj >= (float)h;
			const int tj = (int)pyj, bj = (int)(pyj + 1.f);
			bool row_ok = !act_y || (!out_yj && tj == gy0 + lane && bj == tj + 1 && bj < h);
			// columns: regular taps, the last source column (right tap = left tap), or outside the source on ONE side of the
			// tile ("nearest": the first / last source column, held by one lane of the block; "background": the constant)
			const uint64_t outs = __ballot(act_x && out_x);
			const int edge_lane = outs ? ((outs & 1) ? -gx0 : w - 1 - gx0) : 0; // lane whose left tap column is the clamped column
			const uint64_t ins = __ballot(act_x && !out_x);
			const bool one_side = outs == 0 || (((outs & 1) == 0 || ins == 0 || (outs >> __builtin_ctzll(ins)) == 0) && edge_lane >= 0 && edge_lane <= 61);
			const bool col_ok = !act_x || out_x || (l == gx0 + lane && (r_is_l || r == l + 1));
			const float vvj = (float)bj - pyj;
			const float vv0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, vvj))); // row 0 (y0 < h)
			row_ok = row_ok && (!act_y || __builtin_bit_cast(int, vvj) == __builtin_bit_cast(int, vv0)); // one vertical weight for the tile
			// (frames of 2 GiB and more do not fit a buffer descriptor: they take the general path)
			// rows: the block must not touch the first / last R rows (their renormalisation depends on the row); columns may -
			// the renormalisation of the first / last R columns is a per-lane factor, applied below where a tile needs it
			const bool rows_interior = gy0 >= R && gy0 + TY - 1 < h - R && (int64_t)w * h < (1 << 30);
			const bool xb = xin && (cx < R || cx >= w - R);
			const bool needs_norm = __ballot(xb) != 0;
			if (rows_interior && one_side && __ballot(!(row_ok && col_ok)) == 0)
			{
				auto shl1 = [](float f) -> float
				{ return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x130, 0xf, 0xf, true)); }; // lane i <- i + 1
				auto shr1 = [](float f) -> float
				{ return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x138, 0xf, 0xf, true)); }; // lane i <- i - 1
				float norm_den = 1.f;
				if (needs_norm)
				{ // sx * sy of the general path with sy = the full 1-D sum (no row of the block is a border row)
					float full = 0.f;
#pragma unroll
					for (int d = 0; d < KW; ++d)
						full += a[d];
					float sx = 0.f;
#pragma unroll
					for (int d = -R; d <= R; ++d)
						if (cx + d >= 0 && cx + d < w)
							sx += a[d + R];
					norm_den = sx * full;
				}
				const double u1 = 1 - u;
				const double vv = (double)vv0, v1 = 1 - vv;
				// Results leave through raw-buffer stores (written once, not read again here).
				// The vector-memory path handles one wave instruction per few cycles whatever the bytes per lane, and 14
				// two-byte stores per tile were a quarter of the kernel's time: when rows are 8-byte aligned (w % 4 == 0)
				// the tile is turned through the wave's LDS strip so that a lane writes 4 neighbouring pixels at once -
				// 15 pieces per row, OH rows, 4 store instructions instead of OH.
				const uint64_t db = (uint64_t)(dst + fbase);
				const uint32_t dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)db);
				const uint32_t dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(db >> 32));
				const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)dhi << 32) | dlo), 0, w * h * 2, 0x00020000);
				const uint32_t dstep = (uint32_t)w * 2u;
				const bool wide = (w & 3) == 0;
				uint16_t *ot = reinterpret_cast<uint16_t *>(&tile_w[0][0]); // [OH][64] uint16
				uint16_t res[OH];
				// output row j from the filtered pixels of rows j (top tap) and j + 1 (bottom tap) of this lane's left tap column;
				// (rows past the end of the image are computed and not stored: no early exit, the loop unrolls)
				auto blend_row = [&](double top, double bottom) -> uint16_t {
					const double cl = bottom * v1 + top * vv;
					const uint64_t clb = __builtin_bit_cast(uint64_t, cl);
					const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)clb, 0x130, 0xf, 0xf, true);
					const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(clb >> 32), 0x130, 0xf, 0xf, true);
					const double cs1 = __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
					const double cr = r_is_l ? cl : cs1;
					uint16_t out = CastTo<u16_via_f32>::from(cl * u1 + cr * u).v;
					if (outs != 0)
					{ // (wave-uniform) columns outside the source: the constant, or the first / last source column of row t_j
						const uint64_t eb = __builtin_bit_cast(uint64_t, top);
						const uint32_t elo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)eb, edge_lane & 63);
						const uint32_t ehi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(eb >> 32), edge_lane & 63);
						const uint16_t near = CastTo<u16_via_f32>::from(__builtin_bit_cast(double, (uint64_t)elo | ((uint64_t)ehi << 32))).v;
						out = out_x ? (strategy == TRANSLATE_CONSTANT ? (uint16_t)background : near) : out;
					}
					return out;
				};
				double carry = 0.0; // filtered pixel (gx0 + lane, gy0 + 2p - 1): the bottom row of the previous pair
#pragma unroll
				for (int p = 0; p < HY; ++p)
				{
					const v2f cs = column_pair(2 * p, 2 * p + 1); // column sums of rows 2p, 2p + 1
					v2f acc = __builtin_elementwise_fma((v2f){a[0], a[0]}, (v2f){shr1(cs.x), shr1(cs.y)}, (v2f){0.f, 0.f});
					acc = __builtin_elementwise_fma((v2f){a[1], a[1]}, cs, acc);
					acc = __builtin_elementwise_fma((v2f){a[2], a[2]}, (v2f){shl1(cs.x), shl1(cs.y)}, acc);
					if (needs_norm)
					{ // (wave-uniform) first / last R columns of the image: renormalised by the weight of the taps inside it
						acc.x = xb ? acc.x / norm_den : acc.x;
						acc.y = xb ? acc.y / norm_den : acc.y;
					}
					// lane i owns column gx0 - 1 + i: column gx0 + i, this lane's left tap column, is one lane up
					const double d0 = (double)shl1(acc.x), d1 = (double)shl1(acc.y);
					if (p > 0)
						res[2 * p - 1] = blend_row(carry, d0);
					if (2 * p < OH)
						res[2 * p] = blend_row(d0, d1);
					carry = d1;
				}
				if (!wide)
				{
#pragma unroll
					for (int j = 0; j < OH; ++j)
						__builtin_amdgcn_raw_buffer_store_b16(res[j], rd,
															  (int)((act_x && y0 + j < h) ? (uint32_t)((y0 * w + x) * 2) + (uint32_t)j * dstep : 0x80000000u), 0, RIR_CHAIN_STORE_AUX);
				}
				else
				{
#pragma unroll
					for (int j = 0; j < OH; ++j)
						ot[j * 64 + lane] = res[j];
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					constexpr int PPR = OW / 4; // 4-pixel pieces per row
#pragma unroll
					for (int q = 0; q < (PPR * OH + 63) / 64; ++q)
					{
						const int c = q * 64 + lane;
						const int row = c / PPR, k = c - row * PPR;
						const int ox = x0 + 4 * k, oy = y0 + row; // x0 and w are multiples of 4: a piece is inside the row or outside, never across
						const v2u32 px4 = *reinterpret_cast<const v2u32 *>(ot + (row < OH ? row : 0) * 64 + 4 * k);
						const bool st = c < PPR * OH && ox < w && oy < h;
						__builtin_amdgcn_raw_buffer_store_b64(px4, rd, (int)(st ? (uint32_t)((oy * w + ox) * 2) : 0x80000000u), 0, RIR_CHAIN_STORE_AUX);
					}
				}
				return;
			}
		}
		if constexpr (MODE == 1)
		{ // not a regular tile: on the list, for the general path (filter_chain_listed_kernel)
			if (lane == 0)
			{
				const unsigned int slot = atomicAdd(work_count, 1u);
				worklist[slot] = ((unsigned long long)(unsigned int)n << 32) | ((unsigned long long)(unsigned int)byw << 16) | (unsigned long long)(unsigned int)bx;
			}
			return;
		}
		else
		{
		// ---- general path: row pass through the wave's LDS strip (in place), taps read from it ----------------------
#pragma unroll
		for (int j = 0; j < HY; ++j)
		{
			const v2f cs = column_pair(j, j + HY);
			tile_w[j][lane] = cs.x;
			tile_w[j + HY][lane] = cs.y;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		{
			float full = 0.f;
#pragma unroll
			for (int d = 0; d < KW; ++d)
				full += a[d];
			const bool xb = cx < R || cx >= w - R;
			float sx = full;
			if (xb)
			{
				sx = 0.f;
#pragma unroll
				for (int d = -R; d <= R; ++d)
					if (cx + d >= 0 && cx + d < w)
						sx += a[d + R];
			}
			const bool owner = lane >= R && lane < 64 - R && xin;
#pragma unroll
			for (int j = 0; j < TY; ++j)
			{
				const int gy = gy0 + j;
				if (gy >= h)
					break;
				float acc = 0.f;
#pragma unroll
				for (int d = 0; d < KW; ++d)
					acc = fmaf(a[d], tile_w[j][(lane - R + d) & 63], acc);
				const bool yb = gy < R || gy >= h - R;
				if (xb || yb)
				{
					float sy = full;
					if (yb)
					{
						sy = 0.f;
#pragma unroll
						for (int d = -R; d <= R; ++d)
							if (gy + d >= 0 && gy + d < h)
								sy += a[d + R];
					}
					acc = acc / (sx * sy);
				}
				if (owner)
					tile_w[j][lane] = acc; // every lane has read row j before any lane writes it (one wave, in order)
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// translate<float -> uint16> on the filtered block
		if (!act_x)
			return;
		// a filtered pixel of the block: LDS address clamped into the strip (always readable); `ok` says whether it is the
		// pixel asked for - when not (float rounding of px + 1 only), the pixel is recomputed from global memory
		auto in_block = [&](int c, int rr) -> bool { return c >= gx0 && c < gx0 + OUTW && rr >= gy0 && rr < gy0 + TY; };
		auto block_px = [&](int c, int rr) -> float { return tile_w[min(max(rr - gy0, 0), TY - 1)][min(max(c - gx0 + R, 0), 63)]; };
		uint16_t *d = dst + fbase + x;
		for (int j = 0; j < OH; ++j)
		{
			const int y = y0 + j;
			if (y >= h)
				break;
			const float py = (float)y - dy;
			const bool out_y = py < 0 || py >= (float)h;
			const bool outside = out_x || out_y;
			// taps: inside -> (l,b) (l,t) (r,b) (r,t); outside with "nearest" -> the clamped pixel, four times
			int t = out_y ? (py < 0 ? 0 : h - 1) : (int)py;
			int b = out_y ? t : (int)(py + 1.f);
			if (b >= h)
				b = t;
			const int rr = outside ? l : r, bb = outside ? t : b;
			float p1 = block_px(l, bb), p2 = block_px(l, t), p3 = block_px(rr, bb), p4 = block_px(rr, t);
			if (!(in_block(l, t) && in_block(rr, bb)))
			{
				p1 = chain_gauss_point<R>(s, w, h, l, bb, bp, fix_base, kern);
				p2 = chain_gauss_point<R>(s, w, h, l, t, bp, fix_base, kern);
				p3 = chain_gauss_point<R>(s, w, h, rr, bb, bp, fix_base, kern);
				p4 = chain_gauss_point<R>(s, w, h, rr, t, bp, fix_base, kern);
			}
			const double vv = (double)((float)b - py);
			const uint16_t blend =
				CastTo<u16_via_f32>::from(((double)p1 * (1 - vv) + (double)p2 * vv) * (1 - u) + ((double)p3 * (1 - vv) + (double)p4 * vv) * u).v;
			const uint16_t near = CastTo<u16_via_f32>::from((double)p2).v;
			d[(int64_t)y * w] = outside ? (strategy == TRANSLATE_CONSTANT ? (uint16_t)background : near) : blend;
		}
		}
	}

	// register budget of the one-kernel form (radius 2: 6 waves per SIMD - at 7 it spilled 2 VGPRs; wider: the compiler's choice)
#define RIR_CHAIN_OCC __attribute__((amdgpu_waves_per_eu(R == 1 ? 7 : R == 2 ? 5 : 1, R == 1 ? 7 : R == 2 ? 5 : 8)))
	// grid = (tiles along x, groups of 4 tile rows, frames), block = 256 (4 independent waves, one tile each)
	template <int R>
	RIR_CHAIN_OCC __global__ __launch_bounds__(256) void filter_chain_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int w, int h,
															   ChainBadPixels bp, const float *__restrict__ kern, const float *__restrict__ offsets,
															   int per_frame_offsets, int strategy, uint32_t background)
	{
		__shared__ float tile[4][RIR_CHAIN_TY][64];
		const int lane = threadIdx.x & 63;
		const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		// XCD-major tile order, row bands fastest (see gaussian_sep_kernel)
		const unsigned gx = gridDim.x, gy = gridDim.y;
		const unsigned id2 = xcd_major(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
		const int by = (int)(id2 % gy), bx = (int)((id2 / gy) % gx), bz = (int)(id2 / (gy * gx));
		chain_tile<R, 0>(src, dst, w, h, bp, kern, offsets, per_frame_offsets, strategy, background, bx, by * 4 + wv, bz, lane, tile[wv], nullptr, nullptr);
	}
	// radius 1, the regular tiles (same grid): 8 waves per SIMD, no scratch; what it cannot take goes on the list
	__attribute__((amdgpu_waves_per_eu(8, 8))) __global__ __launch_bounds__(256) void filter_chain_regular_kernel(
		const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int w, int h, ChainBadPixels bp, const float *__restrict__ kern,
		const float *__restrict__ offsets, int per_frame_offsets, int strategy, uint32_t background, unsigned long long *__restrict__ worklist,
		unsigned int *__restrict__ work_count)
	{
		__shared__ float tile[4][RIR_CHAIN_TY][64];
		const int lane = threadIdx.x & 63;
		const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		const unsigned gx = gridDim.x, gy = gridDim.y;
		const unsigned id2 = xcd_major(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
		const int by = (int)(id2 % gy), bx = (int)((id2 / gy) % gx), bz = (int)(id2 / (gy * gx));
		chain_tile<1, 1>(src, dst, w, h, bp, kern, offsets, per_frame_offsets, strategy, background, bx, by * 4 + wv, bz, lane, tile[wv], worklist, work_count);
	}
	// radius 1, the listed tiles through the general path: grid = any, block = 256; wave (block, wv) takes entries block * 4 + wv,
	// + 4 * gridDim.x, ... of the list
	__global__ __launch_bounds__(256) void filter_chain_listed_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int w, int h,
																	  ChainBadPixels bp, const float *__restrict__ kern, const float *__restrict__ offsets,
																	  int per_frame_offsets, int strategy, uint32_t background,
																	  const unsigned long long *__restrict__ worklist, const unsigned int *__restrict__ work_count,
																	  unsigned int *__restrict__ other_count)
	{
		__shared__ float tile[4][RIR_CHAIN_TY][64];
		const int lane = threadIdx.x & 63;
		const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		const unsigned int count = __hip_atomic_load(work_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		for (unsigned int e = blockIdx.x * 4u + (unsigned int)wv; e < count; e += gridDim.x * 4u)
		{
			const unsigned long long id = worklist[e];
			chain_tile<1, 2>(src, dst, w, h, bp, kern, offsets, per_frame_offsets, strategy, background, (int)(id & 0xffffu), (int)((id >> 16) & 0xffffu),
							 (int)(id >> 32), lane, tile[wv], nullptr, nullptr);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the strip is reused by the wave's next tile)
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
		// the OTHER counter of the list - the one the previous call on this stream used, long read by then - is emptied for the next
		// call (a ticket that finds the last workgroup to leave costs more than the kernel: 2 048 atomic adds on one address)
		if (blockIdx.x == 0 && threadIdx.x == 0)
			__hip_atomic_store(other_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}

	// the list of a stream's calls: [two counters, used in turn | one 64-bit entry per tile], zeroed once when it is allocated; a call
	// appends under one counter and empties the other one (filter_chain_listed_kernel).  Calls on one stream are ordered, calls on
	// different streams have lists of their own.  Grow-only, kept for the life of the process.  *parity: which counter this call uses.
	static unsigned int *chain_list_of(hipStream_t st, size_t ntile, int *parity)
	{
		struct List
		{
			void *first = nullptr;
			size_t second = 0;
			int calls = 0;
		};
		static std::mutex mu;
		static std::map<hipStream_t, List> *lists = new std::map<hipStream_t, List>;
		std::lock_guard<std::mutex> g(mu);
		if (lists->size() >= 64 && lists->find(st) == lists->end())
		{ // streams come and go (the map is keyed by their handles): the lists of the others are released once in a while
			(void)hipDeviceSynchronize();
			for (auto &kv : *lists)
				if (kv.second.first)
					(void)hipFree(kv.second.first);
			lists->clear();
		}
		auto &e = (*lists)[st];
		if (e.second < ntile)
		{
			void *p = nullptr;
			const size_t want = std::max<size_t>(ntile, 4096);
			if (hipMalloc(&p, 8 + want * 8) != hipSuccess)
				return nullptr; // (the call fails and the counter parity stays where the last successful call left it)
			if (hipMemsetAsync(p, 0, 8, st) != hipSuccess)
			{
				(void)hipFree(p);
				return nullptr;
			}
			if (e.first)
				(void)hipFree(e.first); // (waits for the device: the previous calls that used it are through)
			e.first = p, e.second = want;
		}
		// the parity only moves with a call that goes on to launch: the counter this call uses is the one the previous launch reset
		*parity = e.calls++ & 1;
		return static_cast<unsigned int *>(e.first);
	}

	// d_fix: [nframes][nbad] uint32 scratch (unused when nbad == 0).  radius 1..4, strategy CONSTANT or NEAREST.
	hipError_t launch_filter_chain(const uint16_t *src, uint16_t *dst, int w, int h, int nframes, const int *d_xy, const int *d_row_start, int nbad,
								   int floor_v, uint32_t *d_fix, const float *d_kernel, int radius, const float *d_offsets, int per_frame,
								   int strategy, uint16_t background, hipStream_t st)
	{
		if (radius < 1 || radius > 4 || (strategy != TRANSLATE_CONSTANT && strategy != TRANSLATE_NEAREST))
			return hipErrorInvalidValue;
		const uint32_t fl = floor_v > 0 ? (uint32_t)(uint16_t)floor_v : 0u;
		if (nbad > 0)
			hipLaunchKernelGGL(bad_pixels_fix_kernel, dim3((nbad + 63) / 64, nframes), dim3(64), 0, st, src, (uint16_t *)nullptr, w, h, d_xy, nbad, fl,
							   d_fix);
		ChainBadPixels bp{d_xy, d_row_start, d_fix, nbad, fl};
		const int ow = 64 - 2 * radius - 2, oh = RIR_CHAIN_TY - 2;
		dim3 block(256), grid((w + ow - 1) / ow, (h + 4 * oh - 1) / (4 * oh), nframes);
		switch (radius)
		{
		case 1:
#ifdef RIR_CHAIN_ONE_KERNEL /* (measurements: round 2's form, both paths in one kernel at 7 waves per SIMD) */
			hipLaunchKernelGGL((filter_chain_kernel<1>), grid, block, 0, st, src, dst, w, h, bp, d_kernel, d_offsets, per_frame, strategy, (uint32_t)background);
			break;
#endif
		{
			// regular tiles in registers; the rest through a list (stream-ordered scratch: [count | pad to 8 bytes][one entry per tile])
			const size_t ntile = (size_t)grid.x * grid.y * 4 * grid.z;
			int par = 0;
			unsigned int *counters = chain_list_of(st, ntile, &par);
			if (!counters)
				return hipErrorOutOfMemory;
			unsigned long long *entries = reinterpret_cast<unsigned long long *>(counters + 2);
			unsigned int *count = counters + par;
			// (8 waves per SIMD; with its workgroups per CU capped by unused LDS - 7, 6, 5, 4 waves per SIMD - the kernel is no faster: 0.152-0.159,
			// 0.154-0.156, 0.152-0.157, 0.164, 0.163 ms per 256 frames against 0.153-0.159)
			hipLaunchKernelGGL(filter_chain_regular_kernel, grid, block, 0, st, src, dst, w, h, bp, d_kernel, d_offsets, per_frame, strategy,
							   (uint32_t)background, entries, count);
			// (the list is short - the first / last row bands of every frame: enough workgroups for it, not for every tile)
			const unsigned int lb = (unsigned int)std::min<size_t>((ntile / 8 + 3) / 4 + 1, 2048);
			hipLaunchKernelGGL(filter_chain_listed_kernel, dim3(lb), block, 0, st, src, dst, w, h, bp, d_kernel, d_offsets, per_frame, strategy,
							   (uint32_t)background, entries, count, counters + (par ^ 1));
			break;
		}
		case 2:
			hipLaunchKernelGGL((filter_chain_kernel<2>), grid, block, 0, st, src, dst, w, h, bp, d_kernel, d_offsets, per_frame, strategy, (uint32_t)background);
			break;
		case 3:
			hipLaunchKernelGGL((filter_chain_kernel<3>), grid, block, 0, st, src, dst, w, h, bp, d_kernel, d_offsets, per_frame, strategy, (uint32_t)background);
			break;
		default:
			hipLaunchKernelGGL((filter_chain_kernel<4>), grid, block, 0, st, src, dst, w, h, bp, d_kernel, d_offsets, per_frame, strategy, (uint32_t)background);
			break;
		}
		return hipGetLastError();
	}

	// ---- read-back repair (IRFileLoader::removeBadPixels) ---------------------------------------
	// in place; window shifted inward at the borders, flagged neighbours excluded through the
	// bitmap, so no flagged pixel is ever read: the in-place update has no ordering dependence.
	__global__ __launch_bounds__(64) void remove_bad_pixels_kernel(uint16_t *__restrict__ img, int w, int h, int rows, const int *__restrict__ xy,
																   int nbad, const uint8_t *__restrict__ bitmap)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		const int n = blockIdx.y;
		if (i >= nbad)
			return;
		uint16_t *f = img + (int64_t)n * w * h;
		const int x = xy[2 * i], y = xy[2 * i + 1];
		uint32_t v[9];
		int c = 0;
		const bool small = (w < 3 || rows < 3);
		int dx_st = x - 1, dy_st = y - 1;
		if (!small)
		{
			if (x == 0)
				dx_st = 0;
			else if (x == w - 1)
				dx_st = w - 3;
			if (y == 0)
				dy_st = 0;
			else if (y == rows - 1)
				dy_st = rows - 3;
		}
#pragma unroll
		for (int a = 0; a < 3; ++a)
#pragma unroll
			for (int b = 0; b < 3; ++b)
			{
				const int xx = dx_st + a, yy = dy_st + b;
				bool ok;
				if (small)
					ok = xx >= 0 && yy >= 0 && xx < w && yy < rows;
				else
					ok = bitmap[xx + yy * w] == 0;
				const uint32_t val = ok ? f[xx + (int64_t)yy * w] : 0u;
#pragma unroll
				for (int k = 0; k < 9; ++k)
					if (ok && k == c)
						v[k] = val;
				c += ok;
			}
		if (c == 0)
			return; // reference reads an uninitialised slot here (IRFileLoader.cpp:792-794): leave the pixel
#pragma unroll
		for (int k = 0; k < 9; ++k)
			if (k >= c)
				v[k] = 0xffffffffu;
		f[x + (int64_t)y * w] = rank_select9(v, c);
	}

	hipError_t launch_remove_bad_pixels(uint16_t *img, int w, int h, int rows, int nframes, const int *d_xy, int nbad, const uint8_t *d_bitmap,
										hipStream_t st)
	{
		if (nbad > 0)
			hipLaunchKernelGGL(remove_bad_pixels_kernel, dim3((nbad + 63) / 64, nframes), dim3(64), 0, st, img, w, h, rows, d_xy, nbad, d_bitmap);
		return hipGetLastError();
	}

	// ---- histogram / quantile --------------------------------------------------------------------
	// hist: uint32[nframes][65536].  Each 1024-thread workgroup counts a run of 32 768 pixels in a
	// privatised LDS histogram of 65 536 16-bit counters (128 KiB, two counters per dword: a run
	// cannot overflow 16 bits), then adds its non-empty bins to the frame's histogram in L2.
#define RIR_HIST_RUN 32768
	__global__ __launch_bounds__(1024) void histogram_kernel(const uint16_t *__restrict__ img, const uint8_t *__restrict__ mask, int64_t npx,
															 uint32_t *__restrict__ hist)
	{
		__shared__ uint32_t lh[32768];
		const int n = blockIdx.y, tid = threadIdx.x;
		const uint16_t *f = img + (int64_t)n * npx;
		const uint8_t *m = mask ? mask + (int64_t)n * npx : nullptr;
		uint32_t *hh = hist + (int64_t)n * 65536;
		for (int i = tid; i < 32768; i += 1024)
			lh[i] = 0;
		__syncthreads();
		const int64_t i0 = (int64_t)blockIdx.x * RIR_HIST_RUN;
		const int64_t i1 = i0 + RIR_HIST_RUN < npx ? i0 + RIR_HIST_RUN : npx;
		for (int64_t i = i0 + tid; i < i1; i += 1024)
			if (!m || m[i])
			{
				const uint32_t v = f[i];
				atomicAdd(&lh[v >> 1], 1u << (16 * (v & 1)));
			}
		__syncthreads();
		for (int i = tid; i < 32768; i += 1024)
		{
			const uint32_t c = lh[i];
			if (c & 0xffffu)
				atomicAdd(&hh[2 * i], c & 0xffffu);
			if (c >> 16)
				atomicAdd(&hh[2 * i + 1], c >> 16);
		}
	}

	// find_median_pixel[_mask] without a histogram in memory: one 1024-thread workgroup per frame counts the
	// frame in LDS, one quarter of the value range at a time (16 384 32-bit counters = 64 KiB), carries the
	// cumulative count from quarter to quarter and stops at the quarter that holds the answer - for 14-bit IR
	// data that is the first one, i.e. a single pass over the frame.  Pixels are fetched 8 per lane (16 bytes).
	// Result: first bin b < nbins whose cumulative count >= target, 0 when none (Filters.cpp:56-101), with
	//   target = (size_t)round((float)size * percent)               (Filters.cpp:63, float product)
	//   target = (size_t)(int)round((float)population * percent)     (masked, Filters.cpp:92)
	__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *wave_tot /*[16]*/, uint32_t *total)
	{
		const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
		uint32_t inc = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1)
		{
			const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
			if (lane >= d)
				inc += o;
		}
		if (lane == 63)
			wave_tot[wv] = inc;
		__syncthreads();
		uint32_t pre = 0, tot = 0;
#pragma unroll
		for (int k = 0; k < 16; ++k)
		{
			const uint32_t t = wave_tot[k];
			pre += k < wv ? t : 0u;
			tot += t;
		}
		__syncthreads();
		*total = tot;
		return pre + inc - v;
	}

	__global__ __launch_bounds__(1024) void quantile_select_kernel(const uint16_t *__restrict__ img, const uint8_t *__restrict__ mask, int64_t npx,
																   float percent, int nbins, int *__restrict__ result)
	{
		__shared__ uint32_t cnt[16384];
		__shared__ uint32_t wave_tot[16];
		__shared__ int found;
		const int n = blockIdx.x, tid = threadIdx.x;
		const uint16_t *f = img + (int64_t)n * npx;
		const uint8_t *m = mask ? mask + (int64_t)n * npx : nullptr;
		uint64_t target;
		if (m)
		{
			uint32_t c = 0;
			for (int64_t i = tid; i < npx; i += 1024)
				c += m[i] != 0;
			uint32_t pop;
			(void)block_exclusive_scan_1024(c, wave_tot, &pop);
			target = (uint64_t)(int64_t)(int)roundf(__fmul_rn((float)pop, percent));
		}
		else
			target = (uint64_t)roundf(__fmul_rn((float)(uint64_t)npx, percent));
		const bool vec = !m && ((((uintptr_t)f) & 15) == 0);
		const int64_t nvec = vec ? (npx >> 3) : 0;
		uint64_t cum = 0;
		for (uint32_t q = 0; q < 4; ++q)
		{
			for (int i = tid; i < 16384; i += 1024)
				cnt[i] = 0;
			if (tid == 0)
				found = 0x7fffffff;
			__syncthreads();
			if (nvec > 0)
			{ // Four 16-byte loads of a thread in flight (with one - load, count, load - a workgroup waited a memory latency per 16 KB), every
			  // one an unconditional buffer load (a lane past the end: an offset out of range, zeros it does not count) so that the waits stay
			  // counted; the eight atomics of a round are not under branches either: a pixel of another quarter adds 0.  A slot is asked for
			  // again when its pixels have become counter offsets (the empty asm pins that point: see lossy_hist_mode_body).
				constexpr int kAhead = 4;
				typedef unsigned int q_v4u __attribute__((ext_vector_type(4)));
				const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(f), 0, (int)(uint32_t)(nvec * 16), 0x00020000); // (a frame is < 2^31 bytes: rir_* checks)
				q_v4u qv[kAhead];
#pragma unroll
				for (int a = 0; a < kAhead; ++a)
				{
					qv[a] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(a * 1024 + tid) * 16u, 0, 0);
					__builtin_amdgcn_sched_barrier(0);
				}
				for (int64_t i0 = 0; i0 < nvec; i0 += 1024 * kAhead)
				{
#pragma unroll
					for (int a = 0; a < kAhead; ++a)
					{
						const q_v4u v = qv[a];
						const bool in = i0 + a * 1024 + tid < nvec;
						const uint32_t d[4] = {v.x, v.y, v.z, v.w};
						uint32_t off[8], one[8];
#pragma unroll
						for (int k = 0; k < 4; ++k)
						{
							const uint32_t lo = d[k] & 0xffffu, hi = d[k] >> 16;
							off[2 * k] = (lo & 16383u) * 4u, off[2 * k + 1] = (hi & 16383u) * 4u;
							one[2 * k] = (in && (lo >> 14) == q) ? 1u : 0u, one[2 * k + 1] = (in && (hi >> 14) == q) ? 1u : 0u;
						}
#pragma unroll
						for (int k = 0; k < 8; ++k)
							asm volatile("" : "+v"(off[k]), "+v"(one[k]));
						__builtin_amdgcn_sched_barrier(0);
						qv[a] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(i0 + (a + kAhead) * 1024 + tid) * 16u, 0, 0);
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int k = 0; k < 8; ++k)
							atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(cnt) + off[k]), one[k]);
					}
				}
			}
			for (int64_t i = nvec * 8 + tid; i < npx; i += 1024)
				if (!m || m[i])
				{
					const uint32_t v = f[i];
					if ((v >> 14) == q)
						atomicAdd(&cnt[v & 16383u], 1u);
				}
			__syncthreads();
			// 16 consecutive bins per thread
			uint32_t loc = 0;
#pragma unroll
			for (int k = 0; k < 16; ++k)
				loc += cnt[tid * 16 + k];
			uint32_t tot;
			const uint32_t pre = block_exclusive_scan_1024(loc, wave_tot, &tot);
			uint64_t count = cum + pre;
			for (int k = 0; k < 16; ++k)
			{
				const int b = (int)(q * 16384u) + tid * 16 + k;
				if (b >= nbins)
					break;
				count += cnt[tid * 16 + k];
				if (count >= target)
				{
					atomicMin(&found, b);
					break;
				}
			}
			__syncthreads();
			const int fb = found;
			if (fb != 0x7fffffff)
			{
				if (tid == 0)
					result[n] = fb;
				return;
			}
			cum += tot;
			__syncthreads();
		}
		if (tid == 0)
			result[n] = 0;
	}

	hipError_t launch_quantile_select(const uint16_t *img, const uint8_t *mask, int64_t npx, int nframes, float percent, int nbins, int *d_result,
									  hipStream_t st)
	{
		hipLaunchKernelGGL(quantile_select_kernel, dim3(nframes), dim3(1024), 0, st, img, mask, npx, percent, nbins, d_result);
		return hipGetLastError();
	}

	hipError_t launch_histogram(const uint16_t *img, const uint8_t *mask, int64_t npx, int nframes, uint32_t *d_hist, hipStream_t st)
	{
		hipError_t e = hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * 65536 * (size_t)nframes, st);
		if (e != hipSuccess)
			return e;
		const int blocks = (int)((npx + RIR_HIST_RUN - 1) / RIR_HIST_RUN);
		hipLaunchKernelGGL(histogram_kernel, dim3(blocks, nframes), dim3(1024), 0, st, img, mask, npx, d_hist);
		return hipGetLastError();
	}

	// ---- bad pixel detector ------------------------------------------------------------------
	// global statistics from the 65536-bin histogram of the frame (single block):
	//   median = sorted[size/2]; sum = sum hist[v] * (int32)(v-median)^2   (exact in int64)
	// out[0] = median, out[1..2] = sum as int64 (lo, hi)
	__global__ __launch_bounds__(1024) void bad_pixels_stats_kernel(const uint32_t *__restrict__ hist, uint64_t size, int64_t *__restrict__ out)
	{
		__shared__ uint64_t part[1024];
		__shared__ int med_s;
		__shared__ unsigned long long sum_s;
		const int tid = threadIdx.x;
		uint64_t s = 0;
		for (int k = 0; k < 64; ++k)
			s += hist[tid * 64 + k];
		part[tid] = s;
		if (tid == 0)
		{
			med_s = 0;
			sum_s = 0;
		}
		__syncthreads();
		if (tid == 0)
		{
			uint64_t acc = 0;
			for (int k = 0; k < 1024; ++k)
			{
				const uint64_t t = part[k];
				part[k] = acc;
				acc += t;
			}
		}
		__syncthreads();
		// sorted[size/2] = the value v with cum_before(v) <= size/2 < cum_before(v) + hist[v]
		const uint64_t want = size / 2;
		uint64_t count = part[tid];
		for (int k = 0; k < 64; ++k)
		{
			const int b = tid * 64 + k;
			const uint64_t hb = hist[b];
			if (hb && want >= count && want < count + hb)
				med_s = b;
			count += hb;
		}
		__syncthreads();
		const int med = med_s;
		long long acc = 0;
		for (int k = 0; k < 64; ++k)
		{
			const int b = tid * 64 + k;
			const int32_t d = b - med;
			const int32_t sq = (int32_t)((uint32_t)d * (uint32_t)d); // the reference's 32-bit int product
			acc += (long long)sq * (long long)hist[b];
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
			acc += __shfl_xor(acc, d, 64);
		if ((tid & 63) == 0)
			atomicAdd(&sum_s, (unsigned long long)acc);
		__syncthreads();
		if (tid == 0)
		{
			out[0] = med;
			out[1] = (int64_t)sum_s;
		}
	}

	hipError_t launch_bad_pixels_stats(const uint32_t *d_hist, uint64_t size, int64_t *d_out, hipStream_t st)
	{
		hipLaunchKernelGGL(bad_pixels_stats_kernel, dim3(1), dim3(1024), 0, st, d_hist, size, d_out);
		return hipGetLastError();
	}

	// per-pixel test: sorted in-bounds 5x5 window, median = win[size/2], trimmed std over ranks
	// [size/5, 4*size/5), flag if v < med - f*std, v > med + f*std or v < floor_detect.
	// Ranks are computed by counting (ties by index) - no sort, no scratch.
	__global__ __launch_bounds__(256) void bad_pixels_detect_kernel(const uint16_t *__restrict__ src, int w, int h, double std_factor,
																	int floor_detect, uint8_t *__restrict__ flags)
	{
		const int x = blockIdx.x * blockDim.x + threadIdx.x;
		const int y = blockIdx.y;
		if (x >= w)
			return;
		uint32_t v[25];
		int size = 0;
#pragma unroll
		for (int ddy = -2; ddy <= 2; ++ddy)
#pragma unroll
			for (int ddx = -2; ddx <= 2; ++ddx)
			{
				const int xx = x + ddx, yy = y + ddy;
				const bool ok = xx >= 0 && yy >= 0 && xx < w && yy < h;
				v[(ddy + 2) * 5 + (ddx + 2)] = ok ? (uint32_t)src[xx + (int64_t)yy * w] : 0xffffffffu;
				size += ok;
			}
		const int r_med = size / 2, r_lo = size / 5, r_hi = size * 4 / 5;
		int rank[25];
		long long med = 0;
#pragma unroll
		for (int i = 0; i < 25; ++i)
		{
			int r = 0;
#pragma unroll
			for (int j = 0; j < 25; ++j)
				r += (v[j] < v[i]) || (v[j] == v[i] && j < i);
			rank[i] = r;
			if (r == r_med && v[i] != 0xffffffffu)
				med = (long long)v[i];
		}
		long long sum2 = 0;
#pragma unroll
		for (int i = 0; i < 25; ++i)
			if (v[i] != 0xffffffffu && rank[i] >= r_lo && rank[i] < r_hi)
			{
				const long long d = (long long)v[i] - med;
				sum2 += d * d;
			}
		const long long c = r_hi - r_lo;
		const double var = (double)sum2 / (double)c;
		const double sd = sqrt(var);
		const double lower = (double)med - std_factor * sd;
		const double upper = (double)med + std_factor * sd;
		const uint32_t pv = src[x + (int64_t)y * w];
		flags[x + (int64_t)y * w] = ((double)pv < lower || (double)pv > upper || (int)pv < floor_detect) ? 1 : 0;
	}

	hipError_t launch_bad_pixels_detect(const uint16_t *src, int w, int h, double std_factor, int floor_detect, uint8_t *d_flags, hipStream_t st)
	{
		hipLaunchKernelGGL(bad_pixels_detect_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, src, w, h, std_factor, floor_detect, d_flags);
		return hipGetLastError();
	}

	// ---- 3x3 median filter (medianFilter<u16,u16>) -----------------------------------------------
	__device__ __forceinline__ uint32_t med3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

	// one output pixel, any position (interior: median of 9; edges: median of 3 along the edge; corners: min of 2)
	__device__ __forceinline__ uint16_t median3x3_px(const uint16_t *__restrict__ s, int w, int h, int x, int y)
	{
		const int64_t o = x + (int64_t)y * w;
		const bool row_edge = (y == 0 || y == h - 1), col_edge = (x == 0 || x == w - 1);
		if (row_edge && col_edge)
		{
			const int xn = x == 0 ? 1 : w - 2;
			return min(s[o], s[xn + (int64_t)y * w]);
		}
		if (row_edge)
			return (uint16_t)med3(s[o - 1], s[o], s[o + 1]);
		if (col_edge)
			return (uint16_t)med3(s[o - w], s[o], s[o + w]);
		uint32_t lo[3], mi[3], hi[3];
#pragma unroll
		for (int j = 0; j < 3; ++j)
		{
			const uint32_t a = s[o - w + j - 1], b = s[o + j - 1], c = s[o + w + j - 1];
			lo[j] = min(min(a, b), c);
			hi[j] = max(max(a, b), c);
			mi[j] = med3(a, b, c);
		}
		return (uint16_t)med3(max(max(lo[0], lo[1]), lo[2]), med3(mi[0], mi[1], mi[2]), min(min(hi[0], hi[1]), hi[2]));
	}

	// One wave per tile of 60 x TY outputs, lane per column (as translate / gaussian_filter): lane i loads column x0 - 1 + i
	// for the TY + 2 rows under the tile (raw-buffer loads, one per row), sorts its column triple of every output row once
	// (min / median / max of 3) and takes the sorted triples of the two neighbouring columns from lanes i - 1 and i + 1 by
	// DPP wave shifts: median of 9 = med3(max of the column minima, median of the column medians, min of the column
	// maxima) - 11 vector instructions per pixel.  Pixels on the image border (median of 3 along the edge, min of 2 in the
	// corners) take median3x3_px.  XCD-major tile order with x fastest (60-pixel rows are not line-aligned).
	// (The first version - 8 consecutive outputs per thread, 3 x 10 windows fetched with dword loads - was bound by its
	// instruction count at 0.134 ms per 256 frames 640x512.)
#ifndef RIR_MEDIAN_TY
#define RIR_MEDIAN_TY 16
#endif
	__device__ __forceinline__ uint32_t wave_shl1_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
	__device__ __forceinline__ uint32_t wave_shr1_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }

	__global__ __launch_bounds__(256) void median3x3_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int w, int h)
	{
		constexpr int TY = RIR_MEDIAN_TY, OW = 60; // (62 outputs fit a wave; 60 keeps the tiles 8-byte aligned for the stores)
		__shared__ uint16_t strip[4][TY][64];
		const int lane = threadIdx.x & 63;
		const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		int bx, by, n;
		{
			const unsigned gx = gridDim.x, gy = gridDim.y;
			const unsigned id2 = xcd_major(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
			bx = (int)(id2 % gx);
			by = (int)((id2 / gx) % gy);
			n = (int)(id2 / (gx * gy));
		}
		const int x0 = bx * OW, y0 = (by * 4 + wv) * TY;
		if (y0 >= h)
			return;
		const int64_t fbase = (int64_t)n * w * h;
		const uint16_t *s = src + fbase;
		uint16_t *d = dst + fbase;
		const int x = x0 - 1 + lane; // this lane's column
		const bool xin = x >= 0 && x < w;
		uint32_t v[TY + 2];
		if ((int64_t)w * h < (1 << 30))
		{ // pixels outside the image read as 0 through the buffer's range check (they are never used by an interior output)
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)s);
			const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)s >> 32));
			const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, w * h * 2, 0x00020000);
			const uint32_t step = (uint32_t)w * 2u;
			if (x0 - 1 >= 0 && x0 + 62 < w && y0 - 1 >= 0 && y0 + TY < h)
			{ // block inside the image: the row steps on the scalar side (the scalar offset is not range-checked)
				const uint32_t off = (uint32_t)(((y0 - 1) * w + x) * 2);
#pragma unroll
				for (int i = 0; i < TY + 2; ++i)
					v[i] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, (int)((uint32_t)i * step), 0);
			}
			else
			{
				uint32_t off = xin ? (uint32_t)(((y0 - 1) * w + x) * 2) : 0x80000000u;
#pragma unroll
				for (int i = 0; i < TY + 2; ++i)
				{
					v[i] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
					off += step;
				}
			}
		}
		else
		{
			const uint16_t *col = s + min(max(x, 0), w - 1);
#pragma unroll
			for (int i = 0; i < TY + 2; ++i)
				v[i] = col[(int64_t)min(max(y0 - 1 + i, 0), h - 1) * w];
		}
		const bool out_lane = lane >= 1 && lane <= OW && x < w;
		const bool x_interior = x >= 1 && x < w - 1;
		const bool wide = (w & 3) == 0 && (int64_t)w * h < (1 << 30);
#pragma unroll
		for (int j = 0; j < TY; ++j)
		{
			const int y = y0 + j;
			// (all 64 lanes take part in the shifts; rows past the image are computed and not stored)
			const uint32_t a = v[j], b = v[j + 1], c = v[j + 2];
			const uint32_t lo = min(min(a, b), c), hi = max(max(a, b), c), mi = med3(a, b, c);
			const uint32_t m_lo = max(max(lo, wave_shr1_u32(lo)), wave_shl1_u32(lo));
			const uint32_t m_hi = min(min(hi, wave_shr1_u32(hi)), wave_shl1_u32(hi));
			const uint32_t m_mi = med3(wave_shr1_u32(mi), mi, wave_shl1_u32(mi));
			uint32_t res = med3(m_lo, m_mi, m_hi);
			if (out_lane && y < h && !(x_interior && y >= 1 && y < h - 1))
				res = median3x3_px(s, w, h, x, y);
			if (wide)
				strip[wv][j][(lane - 1) & 63] = (uint16_t)res; // strip column = output column - x0
			else if (out_lane && y < h)
				d[(int64_t)y * w + x] = (uint16_t)res;
		}
		if (wide)
		{ // rows of 8-byte aligned width: the tile leaves in 4-pixel pieces through the wave's LDS strip, with the streaming
		  // policy (0.112 -> 0.090 ms per 256 frames; 2-byte stores with the same policy: 0.098)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const uint64_t db = (uint64_t)d;
			const uint32_t dlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)db);
			const uint32_t dhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(db >> 32));
			const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)dhi << 32) | dlo), 0, w * h * 2, 0x00020000);
			constexpr int PPR = OW / 4;
#pragma unroll
			for (int q = 0; q < (PPR * TY + 63) / 64; ++q)
			{
				const int c = q * 64 + lane;
				const int row = c / PPR, k = c - row * PPR;
				const int ox = x0 + 4 * k, oy = y0 + row;
				const v2u32 px4 = *reinterpret_cast<const v2u32 *>(&strip[wv][row < TY ? row : 0][4 * k]);
				const bool st = c < PPR * TY && ox < w && oy < h;
				__builtin_amdgcn_raw_buffer_store_b64(px4, rd, (int)(st ? (uint32_t)((oy * w + ox) * 2) : 0x80000000u), 0, RIR_CHAIN_STORE_AUX);
			}
		}
	}

	hipError_t launch_median3x3(const uint16_t *src, uint16_t *dst, int w, int h, int nframes, hipStream_t st)
	{
		dim3 block(256), grid((unsigned)((w + 59) / 60), (unsigned)((h + 4 * RIR_MEDIAN_TY - 1) / (4 * RIR_MEDIAN_TY)), nframes);
		hipLaunchKernelGGL(median3x3_kernel, grid, block, 0, st, src, dst, w, h);
		return hipGetLastError();
	}

	// ---- byte-plane split / merge (H264Capture::AddFrame h264.cpp:1066-1082, VideoGrabber::toArray :3016-3051) ----
	// The reference hands a 16-bit frame to its video codec as three 8-bit planes with a row stride (`linesize`):
	// U = v & 0xFF, V = v >> 8, Y = 0 or the 8-bit integration-time image.  This build's codec does not need the
	// planes; the two kernels exist for callers that feed / read an external 8-bit plane codec.  8 pixels per lane.
	__global__ __launch_bounds__(256) void split_planes_kernel(const uint16_t *__restrict__ img, const uint8_t *__restrict__ it, int w, int h,
															   int linesize, uint8_t *__restrict__ Y, uint8_t *__restrict__ U, uint8_t *__restrict__ V)
	{
		const int cpr = (w + 7) / 8;
		const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (idx >= (int64_t)cpr * h)
			return;
		const int y = (int)(idx / cpr), x0 = (int)(idx - (int64_t)y * cpr) * 8;
		const int64_t fi = (int64_t)blockIdx.y * w * h, fo = (int64_t)blockIdx.y * linesize * h;
		const uint16_t *s = img + fi + (int64_t)y * w + x0;
		const int64_t o = fo + (int64_t)y * linesize + x0;
		if (x0 + 8 <= w && ((((uintptr_t)s) & 15) == 0) && ((o & 7) == 0) && (((uintptr_t)U | (uintptr_t)V | (uintptr_t)Y) & 7) == 0 &&
			(!it || ((((uintptr_t)(it + fi + (int64_t)y * w + x0)) & 7) == 0)))
		{
			const uint4 v = *reinterpret_cast<const uint4 *>(s);
			const uint32_t d[4] = {v.x, v.y, v.z, v.w};
			uint32_t lo[2] = {0, 0}, hi[2] = {0, 0};
#pragma unroll
			for (int k = 0; k < 4; ++k)
			{
				// bytes (b0 b1 | b2 b3) of a dword = (lo0 hi0 | lo1 hi1): gather the low and the high bytes of 4 dwords
				lo[k >> 1] |= ((d[k] & 0xffu) | ((d[k] >> 8) & 0xff00u)) << (16 * (k & 1));
				hi[k >> 1] |= (((d[k] >> 8) & 0xffu) | ((d[k] >> 16) & 0xff00u)) << (16 * (k & 1));
			}
			*reinterpret_cast<uint2 *>(U + o) = make_uint2(lo[0], lo[1]);
			*reinterpret_cast<uint2 *>(V + o) = make_uint2(hi[0], hi[1]);
			*reinterpret_cast<uint2 *>(Y + o) = it ? *reinterpret_cast<const uint2 *>(it + fi + (int64_t)y * w + x0) : make_uint2(0, 0);
			return;
		}
		for (int k = 0; k < 8 && x0 + k < w; ++k)
		{
			const uint16_t v = s[k];
			U[o + k] = (uint8_t)(v & 0xff);
			V[o + k] = (uint8_t)(v >> 8);
			Y[o + k] = it ? it[fi + (int64_t)y * w + x0 + k] : (uint8_t)0;
		}
	}

	__global__ __launch_bounds__(256) void merge_planes_kernel(const uint8_t *__restrict__ Y, const uint8_t *__restrict__ U, const uint8_t *__restrict__ V,
															   int linesize, int w, int h, uint16_t *__restrict__ img, uint8_t *__restrict__ it)
	{
		const int cpr = (w + 7) / 8;
		const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (idx >= (int64_t)cpr * h)
			return;
		const int y = (int)(idx / cpr), x0 = (int)(idx - (int64_t)y * cpr) * 8;
		const int64_t fo = (int64_t)blockIdx.y * w * h, fi = (int64_t)blockIdx.y * linesize * h;
		const int64_t i = fi + (int64_t)y * linesize + x0;
		uint16_t *d = img + fo + (int64_t)y * w + x0;
		for (int k = 0; k < 8 && x0 + k < w; ++k)
		{
			d[k] = (uint16_t)(U[i + k] | (V[i + k] << 8));
			if (it)
				it[fo + (int64_t)y * w + x0 + k] = Y[i + k];
		}
	}

	hipError_t launch_split_planes(const uint16_t *img, const uint8_t *it, int w, int h, int nframes, int linesize, uint8_t *Y, uint8_t *U,
								   uint8_t *V, hipStream_t st)
	{
		const int64_t chunks = (int64_t)((w + 7) / 8) * h;
		hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((chunks + 255) / 256), nframes), dim3(256), 0, st, img, it, w, h, linesize, Y, U, V);
		return hipGetLastError();
	}
	hipError_t launch_merge_planes(const uint8_t *Y, const uint8_t *U, const uint8_t *V, int linesize, int w, int h, int nframes, uint16_t *img,
								   uint8_t *it, hipStream_t st)
	{
		const int64_t chunks = (int64_t)((w + 7) / 8) * h;
		hipLaunchKernelGGL(merge_planes_kernel, dim3((unsigned)((chunks + 255) / 256), nframes), dim3(256), 0, st, Y, U, V, linesize, w, h, img, it);
		return hipGetLastError();
	}

	// ---- u16 -> f32 (load_imageF) ---------------------------------------------------------------
	__global__ __launch_bounds__(256) void u16_to_f32_kernel(const uint16_t *__restrict__ src, float *__restrict__ dst, int64_t total)
	{
		for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
			dst[i] = (float)src[i];
	}
	// A plain streaming copy, 16 bytes per lane: the probe of rir_buffer_create_beside_device (which placement class is an allocation of?
	// - a kernel that reads one buffer and writes another at the same pace is 7-10 % slower when both are of one class)
	typedef unsigned int probe_v4u __attribute__((ext_vector_type(4)));
	__global__ __launch_bounds__(256) void stream_copy_probe_kernel(const probe_v4u *__restrict__ src, probe_v4u *__restrict__ dst, int64_t n16)
	{
		for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
			dst[i] = __builtin_nontemporal_load(src + i);
	}
	hipError_t launch_stream_copy_probe(const void *src, void *dst, int64_t bytes, hipStream_t st)
	{
		const int64_t n16 = bytes / 16;
		hipLaunchKernelGGL(stream_copy_probe_kernel, dim3(8192), dim3(256), 0, st, static_cast<const probe_v4u *>(src), static_cast<probe_v4u *>(dst), n16);
		return hipGetLastError();
	}

	hipError_t launch_u16_to_f32(const uint16_t *src, float *dst, int64_t total, hipStream_t st)
	{
		int blocks = (int)(((total + 255) / 256) < 4096 ? ((total + 255) / 256) : 4096);
		hipLaunchKernelGGL(u16_to_f32_kernel, dim3(blocks), dim3(256), 0, st, src, dst, total);
		return hipGetLastError();
	}

} // namespace rir
