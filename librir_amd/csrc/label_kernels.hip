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
// Integer work on a frame buffer, in five launches: (1) every 64 x 32 tile is labelled on its own in LDS - LDS atomics, no traffic
// but one read of the image - and leaves every pixel linked to its tile-local root with the root's pixel count beside it; (2) the joins
// across tile borders, the only ones made with global atomics; (3) every pixel learns its final root, tile roots hand their counts to
// it; (4) one workgroup turns the per-block root counts into offsets; (5) labels and tables out.  Every pass reads or
// writes each pixel once, coalesced along x; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "label_kernels.h"

namespace rir
{
	namespace
	{
		constexpr int kBlock = 256;		 // 4 wavefronts
		constexpr int kScanBlock = 1024; // the one workgroup that turns per-block root counts into offsets
// Tile of the first launch: 64 columns (a wavefront per row piece) x RIR_LABEL_TILE_H rows, RIR_LABEL_TILE_THREADS threads.  Measured on 640x512
// images (scripts/variants.py; profiles/r05_label_time.txt): heights 8 / 16 / 32 / 64 with 256 / 512 / 1024 threads are within 8 % of each other on
// every kind of image; 32 rows with 1024 threads (two rows a wavefront) is the best by a few per cent (regions 25.4 us against 27.3 for 16 rows
// and 256 threads, vertical stripes 31.6 against 36.0, the spiral 183 against 206).
#ifndef RIR_LABEL_TILE_H
#define RIR_LABEL_TILE_H 32
#endif
#ifndef RIR_LABEL_TILE_THREADS
#define RIR_LABEL_TILE_THREADS 1024
#endif
		constexpr int kTileW = 64, kTileH = RIR_LABEL_TILE_H, kTileThreads = RIR_LABEL_TILE_THREADS, kRowsPerWave = kTileH / (kTileThreads / 64);
		static_assert(kRowsPerWave * (kTileThreads / 64) == kTileH && kRowsPerWave >= 1, "tile rows divide over the wavefronts");

		// The forests are read and written by many waves at once: a link is loaded and stored as one 32-bit access at the scope that
		// shares it (never a stale copy from the CU's vector cache, never torn).
		template <int SCOPE>
		__device__ __forceinline__ int link_load(const int *p)
		{
			return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
		}
		template <int SCOPE>
		__device__ __forceinline__ void link_store(int *p, int v)
		{
			__hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE);
		}

		// Root of i's tree; on the way every visited node is re-pointed at its grandparent (path splitting).  A link only ever moves to an
		// ancestor, so a reader that sees the older value still climbs the same tree.  Racing with unite(): a node that unite() found to
		// be a root (atomicMin returned the node itself) was a root until that instant, so no splitting store - which only touches nodes
		// read as non-roots - can overwrite the link unite() just made; when the atomicMin lands on a node that had stopped being a root,
		// unite() carries on with the parent it displaced, and whatever a splitting store does to that node stays inside one tree.
		template <int SCOPE>
		__device__ __forceinline__ int find_root(int *L, int i)
		{
			int p = link_load<SCOPE>(&L[i]);
			while (p != i)
			{
				const int g = link_load<SCOPE>(&L[p]);
				if (g != p)
					link_store<SCOPE>(&L[i], g);
				i = p;
				p = g;
			}
			return i;
		}

		// Joins the trees of a and b: the higher root is hung under the lower one, so a root is always the lowest index of its tree.
		// Every failed round lowers max(a, b) (the displaced parent is below the node it was read from), so the loop ends for every wave
		// whatever the others do.
		template <int SCOPE>
		__device__ __forceinline__ void unite(int *L, int a, int b)
		{
			for (;;)
			{
				a = find_root<SCOPE>(L, a);
				b = find_root<SCOPE>(L, b);
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
		constexpr int kTile = __HIP_MEMORY_SCOPE_WORKGROUP, kImage = __HIP_MEMORY_SCOPE_AGENT;

		// A launch works on a batch of images [frames][h][w] (blockIdx.y = the image): where the next image's part of every array begins.
		struct Pitch
		{
			int64_t cells;	// image (in cells) and label image (in ints)
			int64_t plane;	// L, cnt (ints)
			int64_t waves;	// root_bits, wave_before (entries)
			int64_t blocks; // block_roots (entries)
			int64_t table;	// entries of the per-image tables the caller gave room for
		};

		// Launch 1: one workgroup per tile, its rows dealt to the wavefronts (two each), row after row.  A pixel starts linked to the first pixel of its
		// horizontal run inside the tile (a run = neighbours with equal values), then every pixel is joined with the pixel above it.  A
		// vertical join is implied (and skipped) when the pixel and the one above both continue their left neighbour's run: that
		// neighbour pair is joined already - a flat tile makes one join per row instead of one per pixel.  Out: L[i] = the image index of
		// the pixel's tile-local root (-1 on the background), cnt[i] = the tile-local component's pixel count at that root, 0 elsewhere.
		template <class C>
		__global__ __launch_bounds__(kTileThreads) void ccl_tile_kernel(const C *__restrict__ src, C bg, int w, int h, int tiles_x, int *__restrict__ L,
																  int *__restrict__ cnt, unsigned long long *__restrict__ best, Pitch pitch)
		{
			__shared__ int lab[kTileW * kTileH];
			__shared__ int num[kTileW * kTileH];
			__shared__ uint8_t flags[kTileW * kTileH]; // 1: belongs to a component, 2: continues its left neighbour's run
			src += blockIdx.y * pitch.cells, L += blockIdx.y * pitch.plane, cnt += blockIdx.y * pitch.plane;
			if (blockIdx.x == 0 && threadIdx.x == 0)
				best[blockIdx.y] = 0ull;
			const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
			const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
			const int x = tx * kTileW + lane, y0 = ty * kTileH;
#pragma unroll
			for (int j = 0; j < kRowsPerWave; ++j)
			{
				const int ly = wv * kRowsPerWave + j, y = y0 + ly, t = ly * kTileW + lane;
				const bool in = x < w && y < h;
				const int i = in ? y * w + x : 0;
				const C v = in ? src[i] : bg;
				const bool fg = in && v != bg;
				const bool joins_left = fg && x > 0 && src[i - 1] == v;
				const uint64_t m = __ballot(joins_left && lane > 0);
				const uint64_t open = ~m & ((2ull << lane) - 1ull); // lanes at or below this one that start a run (lane 0 always does)
				lab[t] = fg ? ly * kTileW + 63 - __builtin_clzll(open) : -1;
				num[t] = 0;
				flags[t] = (uint8_t)((fg ? 1 : 0) | (joins_left ? 2 : 0));
			}
			__syncthreads();
#pragma unroll
			for (int j = 0; j < kRowsPerWave; ++j)
			{
				const int ly = wv * kRowsPerWave + j, t = ly * kTileW + lane;
				if (ly == 0)
					continue;
				const unsigned f = flags[t], u = flags[t - kTileW];
				if ((f & u & 1u) && !((f & u & 2u) && lane > 0))
					unite<kTile>(lab, t, t - kTileW);
			}
			__syncthreads();
			int root[kRowsPerWave];
#pragma unroll
			for (int j = 0; j < kRowsPerWave; ++j)
			{
				const int t = (wv * kRowsPerWave + j) * kTileW + lane;
				root[j] = -1;
				if (flags[t] & 1u)
				{
					root[j] = find_root<kTile>(lab, t);
					atomicAdd(&num[root[j]], 1);
				}
			}
			__syncthreads();
#pragma unroll
			for (int j = 0; j < kRowsPerWave; ++j)
			{
				const int ly = wv * kRowsPerWave + j, y = y0 + ly, t = ly * kTileW + lane;
				if (x < w && y < h)
				{
					const int i = y * w + x, r = root[j];
					L[i] = r >= 0 ? (y0 + (r >> 6)) * w + tx * kTileW + (r & 63) : -1;
					cnt[i] = r == t ? num[t] : 0;
				}
			}
		}

		// Launch 2: the joins across tile borders - the first row of every tile but the top ones with the row above (same skipping rule:
		// the pair to the left is on the same border), the first column of every tile but the left ones with the pixel to its left when
		// the run continues.
		template <class C>
		__global__ __launch_bounds__(kBlock) void ccl_border_kernel(const C *__restrict__ src, C bg, int w, int h, int rows, int cols, int *L, Pitch pitch)
		{
			src += blockIdx.y * pitch.cells, L += blockIdx.y * pitch.plane;
			const int k = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			if (k < rows * w)
			{
				const int y = (k / w + 1) * kTileH, x = k - (k / w) * w, i = y * w + x;
				const C v = src[i], u = src[i - w];
				if (v == bg || u == bg)
					return;
				if (x > 0 && src[i - 1] == v && src[i - w - 1] == u)
					return;
				unite<kImage>(L, i, i - w);
				return;
			}
			const int q = k - rows * w;
			if (q < cols * h)
			{
				const int x = (q / h + 1) * kTileW, y = q - (q / h) * h, i = y * w + x;
				const C v = src[i];
				if (v != bg && src[i - 1] == v)
					unite<kImage>(L, i, i - 1);
			}
		}

		// Launch 3: every pixel learns its final root (L[i] = root from here on, -1 on the background); tile-local roots that are not
		// final hand their pixel count to the final root; every wavefront leaves the bitmap of the roots among its 64 pixels and how
		// many roots its block holds before it, every block its root count.
		// After the tile pass a pixel points at its tile's root and the climb goes on from tile root to tile root: the pixels of a
		// wavefront that share their first link (most of them, in any image with structure) climb once - the first lane of such a run
		// does, the others take its answer - instead of asking the same few addresses 64 times each.
		// Every store to a link that other waves climb through (the tile roots: cnt > 0) lowers it - atomicMin - so that a
		// path-shortening store of another wave that lands late cannot put an ancestor back where the root has been written.
		__global__ __launch_bounds__(kBlock) void ccl_flatten_kernel(int *L, int n, int *__restrict__ cnt, unsigned long long *__restrict__ root_bits,
																	 int *__restrict__ wave_before, int *__restrict__ block_roots, Pitch pitch)
		{
			__shared__ int wave_roots[kBlock / 64];
			L += blockIdx.y * pitch.plane, cnt += blockIdx.y * pitch.plane, root_bits += blockIdx.y * pitch.waves, wave_before += blockIdx.y * pitch.waves,
				block_roots += blockIdx.y * pitch.blocks;
			const int i = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
			const int first = i < n ? link_load<kImage>(&L[i]) : -1;
			const int c = i < n ? cnt[i] : 0;
			const int left = __shfl_up(first, 1);
			const bool head = first >= 0 && (lane == 0 || left != first);
			int r = -1;
			if (head)
			{
				int at = first, p = link_load<kImage>(&L[at]);
				while (p != at)
				{
					const int g = link_load<kImage>(&L[p]);
					if (g != p)
						atomicMin(&L[at], g);
					at = p;
					p = g;
				}
				r = at;
			}
			const uint64_t heads = __ballot(head) & ((2ull << lane) - 1ull);
			r = __shfl(r, heads ? 63 - __builtin_clzll(heads) : 0);
			if (first < 0)
				r = -1;
			else if (r != first)
			{
				if (c > 0)
					atomicMin(&L[i], r);
				else
					L[i] = r;
			}
			if (c > 0 && r != i)
				atomicAdd(&cnt[r], c);
			const uint64_t roots = __ballot(r >= 0 && r == i);
			if (lane == 0)
			{
				root_bits[i >> 6] = roots;
				wave_roots[wv] = __popcll(roots);
			}
			__syncthreads();
			if (lane == 0)
			{
				int before = 0;
				for (int q = 0; q < wv; ++q)
					before += wave_roots[q];
				wave_before[i >> 6] = before;
			}
			if (threadIdx.x == 0)
			{
				int s = 0;
				for (int q = 0; q < kBlock / 64; ++q)
					s += wave_roots[q];
				block_roots[blockIdx.x] = s;
			}
		}

		// (one workgroup) per-block root counts -> number of roots before each block; count[0] = components + 1 (the reference's table
		// has an entry 0 for the background, Filters.h:489).  Its own launch: folded into the pass above - the block that finishes last
		// doing the sums - it needs a device-scope release from every block, which on this part writes the XCD's L2 back each time:
		// 34 us for the 1280 blocks of a 640 x 512 image, against 4 us for this launch.
		__global__ __launch_bounds__(kScanBlock) void ccl_scan_kernel(int *__restrict__ block_roots, int nb, int *__restrict__ count, Pitch pitch)
		{
			block_roots += blockIdx.x * pitch.blocks, count += blockIdx.x; // (one workgroup per image)
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

		// Launch 5: labels and tables out.  A root's number = the roots before it in raster order + 1; the table's x AND y entries both
		// receive the first pixel's x (signal_processing.cpp:262-263 stores first.x() twice); entry 0 is the background's: (-1, -1), area 0.
		__global__ __launch_bounds__(kBlock) void ccl_labels_kernel(const int *__restrict__ R, int n, int w, const int *__restrict__ cnt,
																	const unsigned long long *__restrict__ root_bits, const int *__restrict__ wave_before,
																	const int *__restrict__ block_before, int *__restrict__ dst, double *__restrict__ xy,
																	int *__restrict__ area, Pitch pitch)
		{
			R += blockIdx.y * pitch.plane, cnt += blockIdx.y * pitch.plane, root_bits += blockIdx.y * pitch.waves, wave_before += blockIdx.y * pitch.waves,
				block_before += blockIdx.y * pitch.blocks, dst += blockIdx.y * pitch.cells, xy += blockIdx.y * pitch.table * 2, area += blockIdx.y * pitch.table;
			const int i = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x);
			if (i >= n)
				return;
			if (i == 0)
			{
				xy[0] = -1.0;
				xy[1] = -1.0;
				area[0] = 0;
			}
			const int r = R[i];
			int k = 0;
			if (r >= 0)
				k = block_before[r / kBlock] + wave_before[r >> 6] + __popcll(root_bits[r >> 6] & ((1ull << (r & 63)) - 1ull)) + 1;
			dst[i] = k;
			if (r == i && k < pitch.table) // (a table with less room than the image has components keeps its first entries; the count says how many there are)
			{
				const double x = (double)(i % w);
				xy[2 * (int64_t)k] = x;
				xy[2 * (int64_t)k + 1] = x;
				area[k] = cnt[i];
			}
		}

		// keepLargestArea: the component with the most pixels, the first in raster order among equals (Filters.h:524-533 keeps the
		// earlier one unless a later one is strictly larger) = the largest key (area, ~root).  Four cells a thread (one 16-byte load);
		// roots are few: a wavefront that holds one publishes its best when it beats what is published - compared first with a copy from
		// the CU's cache, which may be old but only ever too low - and the others are through after their load: no barrier, no LDS (a batch
		// of 256 images: 1.45 us an image against 2.5 with one cell a thread, a reduction over the block and a look at the published value by
		// every block; an image of noise 38-43 us a call against 40-50).
		__global__ __launch_bounds__(kBlock) void ccl_largest_kernel(const int *__restrict__ R, int n, const int *__restrict__ cnt,
																	 unsigned long long *best, Pitch pitch)
		{
			R += blockIdx.y * pitch.plane, cnt += blockIdx.y * pitch.plane, best += blockIdx.y;
			const int i0 = (int)(blockIdx.x * (unsigned)kBlock + threadIdx.x) * 4;
			int r[4] = {-1, -1, -1, -1};
			if (i0 + 3 < n)
			{
				const int4 v = *reinterpret_cast<const int4 *>(R + i0); // (a plane starts on a 64-byte boundary)
				r[0] = v.x, r[1] = v.y, r[2] = v.z, r[3] = v.w;
			}
			else
				for (int q = 0; q < 4; ++q)
					if (i0 + q < n)
						r[q] = R[i0 + q];
			unsigned long long key = 0ull;
#pragma unroll
			for (int q = 0; q < 4; ++q)
				if (r[q] == i0 + q)
				{ // (within a thread a later root is a later cell: it wins only when strictly larger)
					const unsigned long long k = ((unsigned long long)(unsigned)cnt[i0 + q] << 32) | (unsigned)~(unsigned)(i0 + q);
					key = k > key ? k : key;
				}
			if (__ballot(key != 0ull) == 0ull)
				return;
			for (int d = 32; d >= 1; d >>= 1)
			{
				const unsigned long long o = __shfl_xor(key, d);
				key = o > key ? o : key;
			}
			if ((threadIdx.x & 63) == 0 && key > *static_cast<const volatile unsigned long long *>(best) &&
				key > __hip_atomic_load(best, __ATOMIC_RELAXED, kImage))
				atomicMax(best, key);
		}
		__global__ __launch_bounds__(kBlock) void ccl_keep_kernel(const int *__restrict__ R, int n, const unsigned long long *__restrict__ best,
																  int fg_value, int bg_value, int *__restrict__ dst, Pitch pitch)
		{
			R += blockIdx.y * pitch.plane, best += blockIdx.y, dst += blockIdx.y * pitch.cells;
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
			int *L, *cnt, *block_roots, *wave_before;
			unsigned long long *root_bits, *best;
			int n, nb;
			Pitch pitch;
		};
		size_t align64(size_t b) { return (b + 63) & ~(size_t)63; }
		Pitch pitch_of(int w, int h, int64_t table)
		{
			const size_t n = (size_t)w * h, nb = (n + kBlock - 1) / kBlock;
			Pitch p;
			p.cells = (int64_t)n;
			p.plane = (int64_t)(align64(n * sizeof(int)) / sizeof(int));
			p.waves = (int64_t)((nb * (kBlock / 64) + 7) & ~(size_t)7);
			p.blocks = (int64_t)((nb + 15) & ~(size_t)15);
			p.table = table;
			return p;
		}
		Work carve(void *d_work, int w, int h, int frames, int64_t table)
		{
			Work k;
			k.n = w * h;
			k.nb = (k.n + kBlock - 1) / kBlock;
			k.pitch = pitch_of(w, h, table);
			// (the planes are read with 16-byte loads: the workspace is used from its first 64-byte boundary on - label_workspace_bytes holds
			// 64 spare bytes for that; the callers only promise 8-byte alignment)
			char *p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(d_work) + 63) & ~(uintptr_t)63);
			const size_t f = (size_t)frames;
			k.L = reinterpret_cast<int *>(p);
			k.cnt = reinterpret_cast<int *>(p += f * k.pitch.plane * sizeof(int));
			k.root_bits = reinterpret_cast<unsigned long long *>(p += f * k.pitch.plane * sizeof(int));
			k.wave_before = reinterpret_cast<int *>(p += f * k.pitch.waves * sizeof(unsigned long long));
			k.block_roots = reinterpret_cast<int *>(p += align64(f * k.pitch.waves * sizeof(int)));
			k.best = reinterpret_cast<unsigned long long *>(p += align64(f * k.pitch.blocks * sizeof(int)));
			return k;
		}

		template <class C>
		hipError_t forest(const void *d_src, const void *background, int w, int h, int frames, const Work &k, hipStream_t st)
		{
			C bg;
			__builtin_memcpy(&bg, background, sizeof(C));
			const int tiles_x = (w + kTileW - 1) / kTileW, tiles_y = (h + kTileH - 1) / kTileH;
			const unsigned fy = (unsigned)frames;
			hipLaunchKernelGGL(ccl_tile_kernel<C>, dim3((unsigned)(tiles_x * tiles_y), fy), dim3(kTileThreads), 0, st, static_cast<const C *>(d_src), bg, w, h,
							   tiles_x, k.L, k.cnt, k.best, k.pitch);
			const int64_t joins = (int64_t)(tiles_y - 1) * w + (int64_t)(tiles_x - 1) * h;
			if (joins > 0)
				hipLaunchKernelGGL(ccl_border_kernel<C>, dim3((unsigned)((joins + kBlock - 1) / kBlock), fy), dim3(kBlock), 0, st,
								   static_cast<const C *>(d_src), bg, w, h, tiles_y - 1, tiles_x - 1, k.L, k.pitch);
			hipLaunchKernelGGL(ccl_flatten_kernel, dim3((unsigned)k.nb, fy), dim3(kBlock), 0, st, k.L, k.n, k.cnt, k.root_bits, k.wave_before, k.block_roots,
							   k.pitch);
			return hipGetLastError();
		}
		hipError_t forest_of(int cell_bytes, const void *d_src, const void *background, int w, int h, int frames, const Work &k, hipStream_t st)
		{
			switch (cell_bytes)
			{
			case 1:
				return forest<uint8_t>(d_src, background, w, h, frames, k, st);
			case 2:
				return forest<uint16_t>(d_src, background, w, h, frames, k, st);
			case 4:
				return forest<uint32_t>(d_src, background, w, h, frames, k, st);
			case 8:
				return forest<uint64_t>(d_src, background, w, h, frames, k, st);
			case -4:
				return forest<float>(d_src, background, w, h, frames, k, st);
			case -8:
				return forest<double>(d_src, background, w, h, frames, k, st);
			default:
				return hipErrorInvalidValue;
			}
		}
		// (a launch's second grid dimension carries the image: 65 535 at most)
		bool geometry_ok(int w, int h, int frames) { return w > 0 && h > 0 && (int64_t)w * h <= 0x7FFF0000LL && frames > 0 && frames <= 65535; }
	} // namespace

	size_t label_workspace_bytes(int w, int h, int frames)
	{
		if (!geometry_ok(w, h, frames))
			return 0;
		const Pitch p = pitch_of(w, h, 0);
		const size_t f = (size_t)frames;
		return 2 * f * p.plane * sizeof(int) + f * p.waves * sizeof(unsigned long long) + align64(f * p.waves * sizeof(int)) + align64(f * p.blocks * sizeof(int)) +
			   align64(f * sizeof(unsigned long long)) + 64;
	}

	hipError_t launch_label_images(int cell_bytes, const void *d_src, const void *background, int w, int h, int frames, int *d_dst, double *d_xy,
								   int *d_area, int64_t table_entries, int *d_count, void *d_work, hipStream_t st)
	{
		if (!geometry_ok(w, h, frames) || table_entries < 1 || !d_src || !background || !d_dst || !d_xy || !d_area || !d_count || !d_work)
			return hipErrorInvalidValue;
		const Work k = carve(d_work, w, h, frames, table_entries);
		hipError_t e = forest_of(cell_bytes, d_src, background, w, h, frames, k, st);
		if (e != hipSuccess)
			return e;
		hipLaunchKernelGGL(ccl_scan_kernel, dim3((unsigned)frames), dim3(kScanBlock), 0, st, k.block_roots, k.nb, d_count, k.pitch);
		hipLaunchKernelGGL(ccl_labels_kernel, dim3((unsigned)k.nb, (unsigned)frames), dim3(kBlock), 0, st, k.L, k.n, w, k.cnt, k.root_bits, k.wave_before,
						   k.block_roots, d_dst, d_xy, d_area, k.pitch);
		return hipGetLastError();
	}

	hipError_t launch_keep_largest_areas(int cell_bytes, const void *d_src, const void *background, int w, int h, int frames, int *d_dst, int foreground,
										 int background_as_int, void *d_work, hipStream_t st)
	{
		if (!geometry_ok(w, h, frames) || !d_src || !background || !d_dst || !d_work)
			return hipErrorInvalidValue;
		const Work k = carve(d_work, w, h, frames, 0);
		hipError_t e = forest_of(cell_bytes, d_src, background, w, h, frames, k, st);
		if (e != hipSuccess)
			return e;
		hipLaunchKernelGGL(ccl_largest_kernel, dim3((unsigned)((k.nb + 3) / 4), (unsigned)frames), dim3(kBlock), 0, st, k.L, k.n, k.cnt, k.best, k.pitch);
		hipLaunchKernelGGL(ccl_keep_kernel, dim3((unsigned)k.nb, (unsigned)frames), dim3(kBlock), 0, st, k.L, k.n, k.best, foreground, background_as_int, d_dst,
						   k.pitch);
		return hipGetLastError();
	}
} // namespace rir
