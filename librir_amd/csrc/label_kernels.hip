// Connected-component labelling of one image — gfx950 (CDNA4).
//
//   labelImage       reference src/cpp/signal_processing/Filters.h:365-509
//   keepLargestArea  reference src/cpp/signal_processing/Filters.h:511-540
//
// What the reference computes (a sequential raster scan with a table of label equivalences), restated as a property of the image:
//   * a pixel belongs to a component when its value differs from `background` (value == background is the host `==` of the cell type:
//     integers compare their bits, float / double compare as IEEE numbers, so a NaN is never background and never equal to a neighbour);
//   * two pixels one above the other are joined whenever both belong to a component - WHATEVER their values (Filters.h:409-444 and :447
//     look at the label of the pixel above, not at its value); two pixels side by side are joined when their values are equal (:405);
//   * components are numbered 1, 2, ... in the raster order of their first pixel (:458-486: final numbers are handed out in the order of
//     the provisional labels, and a component's lowest provisional label is the one its first pixel opened);
//   * per component: its pixel count and its first pixel in raster order (:493-505).
// So the result does not depend on the scan: the kernels below build the same partition with a union-find forest over pixel indices in
// which a root is always the LOWEST index of its tree (links go from the higher root to the lower one, atomicMin), i.e. the component's
// first pixel in raster order; numbering the roots by a prefix sum over the raster gives the reference's numbers bit for bit.
//
// Integer work on a frame buffer: every pass reads or writes each pixel once, coalesced along x; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "label_kernels.h"

namespace rir
{
	namespace
	{
		constexpr int kBlock = 256;		 // 4 wavefronts
		constexpr int kScanBlock = 1024; // the one workgroup that turns per-block root counts into offsets

		// The forest is read and written by every wave at once: loads and stores of a link are single 32-bit accesses at device scope
		// (never a stale copy from the CU's vector cache, never torn).
		__device__ __forceinline__ int link_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		__device__ __forceinline__ void link_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

		// Root of i's tree; on the way every visited node is re-pointed at its grandparent (path splitting).  A link only ever moves to an
		// ancestor, so a reader that sees the older value still climbs the same tree.  Racing with unite(): a node that unite() found to
		// be a root (atomicMin returned the node itself) was a root until that instant, so no splitting store - which only touches nodes
		// read as non-roots - can overwrite the link unite() just made; when the atomicMin lands on a node that had stopped being a root,
		// unite() carries on with the parent it displaced, and whatever a splitting store does to that node stays inside one tree.
		__device__ __forceinline__ int find_root(int *L, int i)
		{
			int p = link_load(&L[i]);
			while (p != i)
			{
				const int g = link_load(&L[p]);
				if (g != p)
					link_store(&L[i], g);
				i = p;
				p = g;
			}
			return i;
		}

		// Joins the trees of a and b.  Every failed round lowers max(a, b) (the displaced parent is below the node it was read from), so
		// the loop ends for every wave whatever the others do.
		__device__ __forceinline__ void unite(int *L, int a, int b)
		{
			for (;;)
			{
				a = find_root(L, a);
				b = find_root(L, b);
				if (a == b)
					return;
				if (a > b)
				{
					const int t = a;
					a = b;
					b = t;
				}
				const int old = atomicMin(&L[b], a);
				if (old == b)
					return;
				b = old;
			}
		}

		// One wavefront per 64-pixel piece of a row (rows are not cut across wavefronts anywhere else than at multiples of 64).
		struct Piece
		{
			int y, x, i, lane;
			bool in;
		};
		__device__ __forceinline__ bool piece_of(int w, int h, int ppr, Piece &p)
		{
			const int wave = (int)((blockIdx.x * (unsigned)kBlock + threadIdx.x) >> 6);
			p.lane = (int)(threadIdx.x & 63);
			p.y = wave / ppr;
			if (p.y >= h)
				return false; // the whole wavefront
			p.x = (wave - p.y * ppr) * 64 + p.lane;
			p.in = p.x < w;
			p.i = p.y * w + p.x;
			return true;
		}

		// Pass 1: every pixel of a component starts linked to the first pixel of its horizontal run inside the piece (a run = neighbours
		// with equal values); background pixels get -1.
		template <class C>
		__global__ __launch_bounds__(kBlock) void ccl_init_kernel(const C *__restrict__ src, C bg, int w, int h, int ppr, int *__restrict__ L,
																  int *__restrict__ cnt, unsigned long long *__restrict__ best)
		{
			if (blockIdx.x == 0 && threadIdx.x == 0)
				best[0] = 0ull;
			Piece p;
			if (!piece_of(w, h, ppr, p))
				return;
			const C v = p.in ? src[p.i] : bg;
			const bool fg = p.in && v != bg;
			const bool joins_left = fg && p.x > 0 && src[p.i - 1] == v;
			const uint64_t m = __ballot(joins_left);
			const uint64_t open = ~m & ((2ull << p.lane) - 1ull); // lanes at or below this one that do not join their left neighbour
			const int start = open ? 63 - __builtin_clzll(open) : 0;
			if (p.in)
			{
				L[p.i] = fg ? p.i - p.lane + start : -1;
				cnt[p.i] = 0;
			}
		}

		// Pass 2: the joins pass 1 left out - a run that continues from the previous piece, and every pixel with the pixel above it.  A
		// vertical join is implied (and skipped) when the pixel and the one above both continue their left neighbour's run: that
		// neighbour pair is joined already.  A flat image makes one vertical join per row instead of one per pixel.
		template <class C>
		__global__ __launch_bounds__(kBlock) void ccl_merge_kernel(const C *__restrict__ src, C bg, int w, int h, int ppr, int *L)
		{
			Piece p;
			if (!piece_of(w, h, ppr, p))
				return;
			const C v = p.in ? src[p.i] : bg;
			const bool fg = p.in && v != bg;
			const bool joins_left = fg && p.x > 0 && src[p.i - 1] == v;
			bool up = false, up_joins_left = false;
			if (fg && p.y > 0)
			{
				const C u = src[p.i - w];
				up = u != bg;
				up_joins_left = up && p.x > 0 && src[p.i - w - 1] == u;
			}
			if (up && !(joins_left && up_joins_left))
				unite(L, p.i, p.i - w);
			if (joins_left && p.lane == 0)
				unite(L, p.i, p.i - 1);
		}

		// Pass 3: every pixel learns its root (R, -1 on the background), roots collect their pixel counts (one atomic per wavefront and
		// distinct root among its 64 pixels) and every block of 256 pixels counts the roots it holds.
		__global__ __launch_bounds__(kBlock) void ccl_flatten_kernel(int *L, int n, int *__restrict__ R, int *__restrict__ cnt,
																	 int *__restrict__ block_roots)
		{
			__shared__ int wave_roots[kBlock / 64];
			const int i = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			const int lane = (int)(threadIdx.x & 63);
			int r = -1;
			if (i < n)
			{
				if (L[i] >= 0)
					r = find_root(L, i);
				R[i] = r;
			}
			uint64_t todo = __ballot(r >= 0);
			while (todo)
			{
				const int lead = __builtin_ctzll(todo);
				const int rr = __builtin_amdgcn_readlane(r, lead);
				const uint64_t same = __ballot(r == rr);
				if (lane == lead)
					atomicAdd(&cnt[rr], __popcll(same));
				todo &= ~same;
			}
			const uint64_t roots = __ballot(r >= 0 && r == i);
			if (lane == 0)
				wave_roots[threadIdx.x >> 6] = __popcll(roots);
			__syncthreads();
			if (threadIdx.x == 0)
			{
				int s = 0;
				for (int k = 0; k < kBlock / 64; ++k)
					s += wave_roots[k];
				block_roots[blockIdx.x] = s;
			}
		}

		// Pass 4 (one workgroup): per-block root counts -> number of roots before each block; count[0] = components + 1 (the reference's
		// table has an entry 0 for the background, Filters.h:489).
		__global__ __launch_bounds__(kScanBlock) void ccl_scan_kernel(int *__restrict__ block_roots, int nb, int *__restrict__ count)
		{
			__shared__ int wave_sum[kScanBlock / 64];
			__shared__ int carry;
			const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
			if (threadIdx.x == 0)
				carry = 0;
			__syncthreads();
			for (int base = 0; base < nb; base += kScanBlock)
			{
				const int k = base + (int)threadIdx.x;
				const int v = k < nb ? block_roots[k] : 0;
				int inc = v;
				for (int d = 1; d < 64; d <<= 1)
				{
					const int t = __shfl_up(inc, d);
					if (lane >= d)
						inc += t;
				}
				if (lane == 63)
					wave_sum[wv] = inc;
				__syncthreads();
				int before = carry;
				for (int q = 0; q < wv; ++q)
					before += wave_sum[q];
				if (k < nb)
					block_roots[k] = before + inc - v;
				__syncthreads();
				if (threadIdx.x == kScanBlock - 1)
					carry = before + inc;
				__syncthreads();
			}
			if (threadIdx.x == 0)
				count[0] = carry + 1;
		}

		// Pass 5: roots take their numbers (raster order) and publish area and first pixel.  The table's x AND y entries both receive the
		// first pixel's x (signal_processing.cpp:262-263 stores first.x() twice); entry 0 is the background's: (-1, -1), area 0.
		__global__ __launch_bounds__(kBlock) void ccl_number_kernel(const int *__restrict__ R, int n, int w, const int *__restrict__ cnt,
																	const int *__restrict__ block_before, int *__restrict__ number,
																	double *__restrict__ xy, int *__restrict__ area)
		{
			__shared__ int wave_roots[kBlock / 64];
			const int i = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
			const bool root = i < n && R[i] == i;
			const uint64_t roots = __ballot(root);
			if (lane == 0)
				wave_roots[wv] = __popcll(roots);
			__syncthreads();
			if (i == 0)
			{
				xy[0] = -1.0;
				xy[1] = -1.0;
				area[0] = 0;
			}
			if (root)
			{
				int k = block_before[blockIdx.x] + __popcll(roots & ((1ull << lane) - 1ull)) + 1;
				for (int q = 0; q < wv; ++q)
					k += wave_roots[q];
				number[i] = k;
				const double x = (double)(i % w);
				xy[2 * (int64_t)k] = x;
				xy[2 * (int64_t)k + 1] = x;
				area[k] = cnt[i];
			}
		}

		// Pass 6: labels out.
		__global__ __launch_bounds__(kBlock) void ccl_relabel_kernel(const int *__restrict__ R, const int *__restrict__ number, int n,
																	 int *__restrict__ dst)
		{
			const int i = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			if (i >= n)
				return;
			const int r = R[i];
			dst[i] = r >= 0 ? number[r] : 0;
		}

		// keepLargestArea: the component with the most pixels, the first in raster order among equals (Filters.h:524-533 keeps the
		// earlier one unless a later one is strictly larger) = the largest key (area, ~root).
		__global__ __launch_bounds__(kBlock) void ccl_largest_kernel(const int *__restrict__ R, int n, const int *__restrict__ cnt,
																	 unsigned long long *__restrict__ best)
		{
			const int i = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			unsigned long long key = 0ull;
			if (i < n && R[i] == i)
				key = ((unsigned long long)(unsigned)cnt[i] << 32) | (unsigned)~(unsigned)i;
			for (int d = 32; d >= 1; d >>= 1)
			{
				const unsigned long long o = __shfl_xor(key, d);
				key = o > key ? o : key;
			}
			if ((threadIdx.x & 63) == 0 && key)
				atomicMax(best, key);
		}
		__global__ __launch_bounds__(kBlock) void ccl_keep_kernel(const int *__restrict__ R, int n, const unsigned long long *__restrict__ best,
																  int fg_value, int bg_value, int *__restrict__ dst)
		{
			const int i = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			if (i >= n)
				return;
			const unsigned long long b = best[0];
			if (b == 0ull)
			{ // no component: the reference returns with the all-zero label image (Filters.h:518-521)
				dst[i] = 0;
				return;
			}
			const int root = (int)~(unsigned)(b & 0xFFFFFFFFull);
			dst[i] = R[i] == root ? fg_value : bg_value;
		}

		struct Work
		{
			int *L, *cnt, *R, *block_roots;
			unsigned long long *best;
			int n, nb;
		};
		Work carve(void *d_work, int w, int h)
		{
			Work k;
			k.n = w * h;
			k.nb = (k.n + kBlock - 1) / kBlock;
			char *p = static_cast<char *>(d_work);
			const size_t plane = ((size_t)k.n * sizeof(int) + 63) & ~(size_t)63;
			k.L = reinterpret_cast<int *>(p);
			k.cnt = reinterpret_cast<int *>(p + plane);
			k.R = reinterpret_cast<int *>(p + 2 * plane);
			k.block_roots = reinterpret_cast<int *>(p + 3 * plane);
			k.best = reinterpret_cast<unsigned long long *>(p + 3 * plane + (((size_t)k.nb * sizeof(int) + 63) & ~(size_t)63));
			return k;
		}

		template <class C>
		hipError_t forest(const void *d_src, const void *background, int w, int h, const Work &k, hipStream_t st)
		{
			C bg;
			__builtin_memcpy(&bg, background, sizeof(C));
			const int ppr = (w + 63) / 64;
			const unsigned pieces = (unsigned)(((int64_t)ppr * h + kBlock / 64 - 1) / (kBlock / 64));
			hipLaunchKernelGGL(ccl_init_kernel<C>, dim3(pieces), dim3(kBlock), 0, st, static_cast<const C *>(d_src), bg, w, h, ppr, k.L, k.cnt, k.best);
			hipLaunchKernelGGL(ccl_merge_kernel<C>, dim3(pieces), dim3(kBlock), 0, st, static_cast<const C *>(d_src), bg, w, h, ppr, k.L);
			hipLaunchKernelGGL(ccl_flatten_kernel, dim3((unsigned)k.nb), dim3(kBlock), 0, st, k.L, k.n, k.R, k.cnt, k.block_roots);
			return hipGetLastError();
		}
		hipError_t forest_of(int cell_bytes, const void *d_src, const void *background, int w, int h, const Work &k, hipStream_t st)
		{
			switch (cell_bytes)
			{
			case 1:
				return forest<uint8_t>(d_src, background, w, h, k, st);
			case 2:
				return forest<uint16_t>(d_src, background, w, h, k, st);
			case 4:
				return forest<uint32_t>(d_src, background, w, h, k, st);
			case 8:
				return forest<uint64_t>(d_src, background, w, h, k, st);
			case -4:
				return forest<float>(d_src, background, w, h, k, st);
			case -8:
				return forest<double>(d_src, background, w, h, k, st);
			default:
				return hipErrorInvalidValue;
			}
		}
		bool geometry_ok(int w, int h) { return w > 0 && h > 0 && (int64_t)w * h <= 0x7FFFFF00LL; }
	} // namespace

	size_t label_workspace_bytes(int w, int h)
	{
		if (!geometry_ok(w, h))
			return 0;
		const size_t n = (size_t)w * h, nb = (n + kBlock - 1) / kBlock;
		const size_t plane = (n * sizeof(int) + 63) & ~(size_t)63;
		return 3 * plane + ((nb * sizeof(int) + 63) & ~(size_t)63) + 64;
	}

	hipError_t launch_label_image(int cell_bytes, const void *d_src, const void *background, int w, int h, int *d_dst, double *d_xy, int *d_area,
								  int *d_count, void *d_work, hipStream_t st)
	{
		if (!geometry_ok(w, h) || !d_src || !background || !d_dst || !d_xy || !d_area || !d_count || !d_work)
			return hipErrorInvalidValue;
		const Work k = carve(d_work, w, h);
		hipError_t e = forest_of(cell_bytes, d_src, background, w, h, k, st);
		if (e != hipSuccess)
			return e;
		hipLaunchKernelGGL(ccl_scan_kernel, dim3(1), dim3(kScanBlock), 0, st, k.block_roots, k.nb, d_count);
		// (the forest is no longer needed: its plane takes the roots' numbers)
		hipLaunchKernelGGL(ccl_number_kernel, dim3((unsigned)k.nb), dim3(kBlock), 0, st, k.R, k.n, w, k.cnt, k.block_roots, k.L, d_xy, d_area);
		hipLaunchKernelGGL(ccl_relabel_kernel, dim3((unsigned)k.nb), dim3(kBlock), 0, st, k.R, k.L, k.n, d_dst);
		return hipGetLastError();
	}

	hipError_t launch_keep_largest_area(int cell_bytes, const void *d_src, const void *background, int w, int h, int *d_dst, int foreground,
										int background_as_int, void *d_work, hipStream_t st)
	{
		if (!geometry_ok(w, h) || !d_src || !background || !d_dst || !d_work)
			return hipErrorInvalidValue;
		const Work k = carve(d_work, w, h);
		hipError_t e = forest_of(cell_bytes, d_src, background, w, h, k, st);
		if (e != hipSuccess)
			return e;
		hipLaunchKernelGGL(ccl_largest_kernel, dim3((unsigned)k.nb), dim3(kBlock), 0, st, k.R, k.n, k.cnt, k.best);
		hipLaunchKernelGGL(ccl_keep_kernel, dim3((unsigned)k.nb), dim3(kBlock), 0, st, k.R, k.n, k.best, foreground, background_as_int, d_dst);
		return hipGetLastError();
	}
} // namespace rir
