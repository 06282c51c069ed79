// Translation-only ECC alignment (enhanced correlation coefficient maximisation) — gfx950 kernels.
//
// Replaces, for the registration step of the path, the call the reference makes into OpenCV:
// cv2.findTransformECC(template, image, start_mat, MOTION_TRANSLATION, (EPS|COUNT, 500, 1e-3), mask, 1)
// at reference src/python/librir/registration/masked_registration_ecc.py:166-168.  OpenCV is a third-party
// dependency that is not under /root/reference (constraint opencv-python >= 4.11, pyproject.toml.in:21); what
// is implemented here is the published algorithm (Evangelidis & Psarakis, "Parametric image alignment using
// enhanced correlation coefficient maximization", PAMI 2008, forward additive scheme) in the arrangement of
// its well-known implementation: gradients of the input image by the central difference [-1/2 0 1/2] with
// reflected borders, image and gradients sampled at x + t by bilinear interpolation (zero outside), a
// validity mask sampled at the nearest pixel, zero-mean correlation under that mask, 2x2 normal equations.
// The arithmetic is HBM/L2-bound reductions (15 sums per iteration), not a GEMM: no MFMA.
//
// Two launches per iteration: workgroups write their partial sums, one workgroup adds them in a fixed order
// (deterministic), solves the 2x2 system and updates the state in device memory, so iterations queue back to back
// without a host round trip; the host follows the state through a few words of coherent host memory.
#include <cstddef>

#include "ecc_kernels.h"
#include "resident_device.h"
#include "runtime.h"

namespace rir
{
	static int ecc_blocks(int w, int h)
	{
		const int64_t b = ((int64_t)w * h + ECC_BLOCK - 1) / ECC_BLOCK;
		return (int)(b < RIR_ECC_MAX_BLOCKS ? b : RIR_ECC_MAX_BLOCKS);
	}
	size_t ecc_workspace_bytes(int w, int h) { return (size_t)ecc_blocks(w, h) * 16 * sizeof(double); }

	// central difference with reflect-101 borders: g(0) = g(n-1) = 0
	// (thread 0 also resets the alignment's state: one launch less per frame)
	__global__ __launch_bounds__(256) void ecc_gradient_kernel(const float *__restrict__ img, int w, int h, float *__restrict__ gx,
															  float *__restrict__ gy, EccState *s, float tx, float ty, int max_iter, double eps)
	{
		const int i = blockIdx.x * 256 + threadIdx.x;
		if (i == 0)
		{
			s->tx = tx, s->ty = ty;
			s->rho = -1.0, s->last_rho = -eps;
			s->iter = 0, s->done = 0, s->ticket = 0;
			s->max_iter = max_iter, s->eps = eps;
		}
		if (i >= w * h)
			return;
		const int y = i / w, x = i - y * w;
		const int xl = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xr = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
		const int yu = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yd = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
		gx[i] = 0.5f * img[y * w + xr] - 0.5f * img[y * w + xl];
		gy[i] = 0.5f * img[yd * w + x] - 0.5f * img[yu * w + x];
	}

	// 16-byte hand-off granules {value, flag} of the one-launch alignment: written by ONE write-through (sc1) store each, read by sc1
	// loads - a granule is its own flag, nothing has to be drained or ordered (MI355X_MICROARCH.md: 16-byte sc1 granules observed
	// untorn on gfx950).  The flag's top two bits are free for a payload of their own (ecc_run_kernel: `done`).
	typedef unsigned int ecc_v4u __attribute__((ext_vector_type(4)));
	typedef unsigned int ecc_v2u __attribute__((ext_vector_type(2)));
	constexpr unsigned long long kEccFlagMask = 0x3fffffffffffffffull;
	__device__ __forceinline__ __amdgpu_buffer_rsrc_t ecc_rsrc(const void *base, uint32_t bytes)
	{
		const uint64_t b = (uint64_t)base;
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
		const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
		return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
	}
	__device__ __forceinline__ void ecc_granule_store(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, unsigned long long value, unsigned long long flag)
	{
		ecc_v4u g;
		g.x = (unsigned int)value, g.y = (unsigned int)(value >> 32), g.z = (unsigned int)flag, g.w = (unsigned int)(flag >> 32);
		__builtin_amdgcn_raw_buffer_store_b128(g, rs, byte_off, 0, 16 /* sc1 */);
	}

	// LDS of the reductions (18.5 KB: five workgroups to a CU).  Stage 1 goes through val[k - 8 half][thread], a half of the sums at a
	// time (padded: the 16 lanes that read 16 different k of one chunk hit different banks); a chunk of 16 threads lies inside one
	// wave, so only the wave that wrote an entry reads it - no workgroup barrier, LDS serves a wave's accesses in order.  Stage 2
	// goes through part[call parity][k][chunk]: one barrier per call (the call after the next, which writes this parity again, is
	// behind the next call's barrier).
	struct EccReduceLds
	{
		double val[8][ECC_BLOCK + 1];
		double part[2][ECC_NSUMS][ECC_BLOCK / 16 + 1];
	};
	__device__ __forceinline__ void ecc_lds_barrier()
	{ // a workgroup barrier that orders LDS only: loads from memory that are in flight stay in flight across it (__syncthreads() drains them)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
	}
	__device__ __forceinline__ void ecc_wave_sync()
	{ // orders the wave's own LDS writes and reads for the compiler (the hardware keeps a wave's LDS accesses in order)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
#ifdef RIR_ECC_DIAG
	static __shared__ unsigned long long ecc_diag_loop_end, ecc_diag_reduced; // (written by every thread with about the same value)
#endif
#ifndef RIR_ECC_GRAD_ON_THE_FLY
#define RIR_ECC_GRAD_ON_THE_FLY 0 /* measured: slower (81 k against 90 k frames/s over 8 sequences) although the L2 then holds a sequence - DESIGN.md */
#endif
#ifndef RIR_ECC_PIXELS_PER_ROUND
#define RIR_ECC_PIXELS_PER_ROUND 3 /* one sequence, one wave per SIMD: 19.3 k frames/s; 2: 18.8 k, 6: 19.3 k at 226 VGPRs */
#endif
#ifndef RIR_ECC_MULTI_PIXELS_PER_ROUND
#define RIR_ECC_MULTI_PIXELS_PER_ROUND 1 /* several sequences: latency is hidden by waves (RIR_ECC_MULTI_WAVES per SIMD), not by pixels in flight */
#endif
#ifndef RIR_ECC_MULTI_WAVES
#define RIR_ECC_MULTI_WAVES 5 /* 96 VGPRs, 23 KB of LDS: five workgroups to a CU, 1 280 places - the 1 024 compute workgroups of 8 sequences (4 pairs x
                                 256 slices of one row each: four to every CU, the dispatcher deals them evenly - scripts/ubench/wg_placement.hip) and
                                 their 8 service workgroups.  (The first form of the kernel - no service workgroups, the adding done by each sequence's
                                 slice 0 - at 128 VGPRs, 4 workgroups per CU, (pixels per round, waves per SIMD): (1, 4) 70-75 k frames/s over 8
                                 sequences, (2, 4) 65-69, (3, 4) 62-66, (5, 3) 62-65.)  The pixel loop is bound by latency - 44 % of its L2 accesses miss:
                                 a sequence's four arrays are 5.2 MB, an XCD's L2 4 MB (profiles/r04_pmc_ecc.json) - which more pixels per round did not hide */
#endif
#ifndef RIR_ECC_MULTI_MARGIN
#define RIR_ECC_MULTI_MARGIN 0
#endif
	// the 15 sums of workgroup `blk` of `nblk` at translation (tx, ty), reduced over the workgroup (fixed order); valid in threads < ECC_NSUMS
	template <int R> // pixels per round: their 13 R loads are in flight together; the sums are taken in pixel order whatever R is
	__device__ __forceinline__ double ecc_block_sums(const float *__restrict__ templ, const float *__restrict__ image, const float *__restrict__ gximg,
													 const float *__restrict__ gyimg, const uint8_t *__restrict__ mask, int w, int h, float tx, float ty, int blk,
													 int nblk, EccReduceLds &red, int parity)
	{
		double s[ECC_NSUMS];
#pragma unroll
		for (int k = 0; k < ECC_NSUMS; ++k)
			s[k] = 0.0;
		// grid-stride over the pixels (at most RIR_ECC_MAX_BLOCKS workgroups: one row of partials each), two pixels per round: the
		// 26 loads of both are in flight together (a thread has 5 pixels at 640x512 and one wave per SIMD: one pixel per round was
		// five exposed L2 latencies); sums are taken in pixel order as before
		struct Px
		{
			float I, gx, gy, T;
			bool valid;
		};
		// Pixel coordinates advance by the (wave-uniform) stride instead of a division per pixel, and the 13 taps of a pixel are
		// buffer loads with 32-bit offsets (four offsets shared by the three images) instead of 64-bit address arithmetic per tap:
		// the pixel loop is VALU-bound at one wave per SIMD (2.7 of an iteration's 6.6 us).
		const int npx = w * h, stride = nblk * ECC_BLOCK;
		const int dy = stride / w, dx = stride - dy * w;
		const uint32_t bytes = (uint32_t)npx * 4u;
		const __amdgpu_buffer_rsrc_t r_img = ecc_rsrc(image, bytes), r_gx = ecc_rsrc(gximg, bytes), r_gy = ecc_rsrc(gyimg, bytes), r_t = ecc_rsrc(templ, bytes);
		(void)r_gx, (void)r_gy;
		auto ld = [](__amdgpu_buffer_rsrc_t r, uint32_t off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); };
		auto sample = [&](int i, int x, int y, bool inside) {
			const float sx = (float)x + tx, sy = (float)y + ty;
			const float flx = floorf(sx), fly = floorf(sy);
			const int x0 = (int)flx, y0 = (int)fly;
			const float fx = sx - flx, fy = sy - fly;
			// Where nothing can be outside - all four taps of every lane in the image (then the nearest source pixel is too), no mask, no lane
			// past the end - the validity test, the four range tests of the taps and the four selects on their offsets are left out: same
			// values, bit for bit, a sixth of the loop's vector instructions fewer, and the loop is bound by those (profiles/r04_pmc_ecc.json).
			// Decided per WAVE (64 consecutive pixels of a line): with a translation of t pixels every wave but those within t of two borders.
#if !RIR_ECC_GRAD_ON_THE_FLY && !defined(RIR_ECC_NO_INTERIOR_PATH)
			const bool interior = mask == nullptr && __builtin_amdgcn_ballot_w64(!(inside && (unsigned)x0 < (unsigned)(w - 1) && (unsigned)y0 < (unsigned)(h - 1))) == 0;
#else
			const bool interior = false;
#endif
			bool valid = true;
			if (!interior)
			{ // validity: the nearest source pixel lies inside the image and inside the caller's mask
				const int nx = (int)rintf(sx), ny = (int)rintf(sy);
				valid = inside && (unsigned)nx < (unsigned)w && (unsigned)ny < (unsigned)h;
				const uint8_t mv = mask ? mask[min(max(ny, 0), h - 1) * w + min(max(nx, 0), w - 1)] : (uint8_t)1;
				valid = valid && mv != 0;
			}
			// zero outside the image (constant border); every tap is loaded - a load under a condition is a branch with its own wait - and a
			// tap outside the image is loaded from an offset outside the BUFFER: the hardware's range check returns +0.0 for it.  Four
			// selects on the offsets (shared by the three images) instead of eight clamps and twelve selects on the values; the taps
			// inside need no clamp, and their row offset is one 24-bit multiply-add (v_mul_lo_u32 is a quarter-rate instruction).
			const bool xa = (unsigned)x0 < (unsigned)w, xb = (unsigned)(x0 + 1) < (unsigned)w, ya = (unsigned)y0 < (unsigned)h,
					   yb = (unsigned)(y0 + 1) < (unsigned)h;
			constexpr uint32_t kOutside = 0xfffffff0u; // (beyond any image: bytes < 2^32 - 16)
			const uint32_t lin = (uint32_t)(__mul24(y0, w) + x0) * 4u, w4 = (uint32_t)w * 4u; // (meaningful when the tap is inside: y0 < h < 2^23)
			auto lerp2 = [&](float v00, float v01, float v10, float v11) {
				// fused: one rounding less per term (the file is built with -ffp-contract=off, so fusing is spelled out)
				const float top = __builtin_fmaf(fx, v01 - v00, v00), bot = __builtin_fmaf(fx, v11 - v10, v10);
				return __builtin_fmaf(fy, bot - top, top);
			};
			Px p;
			// (The two taps of an image row as ONE 8-byte load - 7 loads per pixel instead of 13, the lanes at the image's left and right edge
			// fixed up afterwards - was measured: the loop took 5.4 us a row instead of 3.6.  8-byte loads at addresses that are only
			// 4-byte aligned are not what the memory pipeline likes.)
#if RIR_ECC_GRAD_ON_THE_FLY
			// The gradient taps are COMPUTED from the image instead of loaded from the gradient arrays: the same twelve loads per pixel, but
			// all of them from ONE array - a sequence's working set is its image and its template, 2.6 MB instead of 5.2, under an XCD's 4 MB
			// L2 (with four arrays 44 % of the L2 accesses of the sequence kernels missed: profiles/r03_pmc_ecc_caches.json).  Same values,
			// bit for bit: a stored gradient is 0.5f * right - 0.5f * left of the stored image (ecc_gradient_kernel,
			// minmax_apply_grad_frames_kernel), two exact products and one rounding, and at the image's edge, where reflection makes both
			// neighbours the same pixel, it is 0.5f a - 0.5f a = +0.
			{
				const bool xm = (unsigned)(x0 - 1) < (unsigned)w, xp = (unsigned)(x0 + 2) < (unsigned)w, ym = (unsigned)(y0 - 1) < (unsigned)h,
						   yp = (unsigned)(y0 + 2) < (unsigned)h;
				const float i00 = ld(r_img, xa && ya ? lin : kOutside), i01 = ld(r_img, xb && ya ? lin + 4u : kOutside);
				const float i10 = ld(r_img, xa && yb ? lin + w4 : kOutside), i11 = ld(r_img, xb && yb ? lin + w4 + 4u : kOutside);
				const float im0 = ld(r_img, xm && ya ? lin - 4u : kOutside), ip0 = ld(r_img, xp && ya ? lin + 8u : kOutside);
				const float im1 = ld(r_img, xm && yb ? lin + w4 - 4u : kOutside), ip1 = ld(r_img, xp && yb ? lin + w4 + 8u : kOutside);
				const float iu0 = ld(r_img, xa && ym ? lin - w4 : kOutside), iu1 = ld(r_img, xb && ym ? lin - w4 + 4u : kOutside);
				const float id0 = ld(r_img, xa && yp ? lin + 2u * w4 : kOutside), id1 = ld(r_img, xb && yp ? lin + 2u * w4 + 4u : kOutside);
				// a tap (c, r) has a horizontal gradient when it is inside the image and not in its first or last column (there: + 0)
				const bool cx0 = x0 > 0 && x0 < w - 1, cx1 = x0 + 1 > 0 && x0 + 1 < w - 1, cy0 = y0 > 0 && y0 < h - 1, cy1 = y0 + 1 > 0 && y0 + 1 < h - 1;
				const float gx00 = cx0 && ya ? 0.5f * i01 - 0.5f * im0 : 0.0f, gx01 = cx1 && ya ? 0.5f * ip0 - 0.5f * i00 : 0.0f;
				const float gx10 = cx0 && yb ? 0.5f * i11 - 0.5f * im1 : 0.0f, gx11 = cx1 && yb ? 0.5f * ip1 - 0.5f * i10 : 0.0f;
				const float gy00 = cy0 && xa ? 0.5f * i10 - 0.5f * iu0 : 0.0f, gy01 = cy0 && xb ? 0.5f * i11 - 0.5f * iu1 : 0.0f;
				const float gy10 = cy1 && xa ? 0.5f * id0 - 0.5f * i00 : 0.0f, gy11 = cy1 && xb ? 0.5f * id1 - 0.5f * i01 : 0.0f;
				p.I = lerp2(i00, i01, i10, i11);
				p.gx = lerp2(gx00, gx01, gx10, gx11);
				p.gy = lerp2(gy00, gy01, gy10, gy11);
			}
#else
			{
				uint32_t o00 = lin, o01 = lin + 4u, o10 = lin + w4, o11 = lin + w4 + 4u;
				if (!interior)
					o00 = xa && ya ? o00 : kOutside, o01 = xb && ya ? o01 : kOutside, o10 = xa && yb ? o10 : kOutside, o11 = xb && yb ? o11 : kOutside;
				auto blend = [&](__amdgpu_buffer_rsrc_t r) { return lerp2(ld(r, o00), ld(r, o01), ld(r, o10), ld(r, o11)); };
				p.I = blend(r_img);
				p.gx = blend(r_gx);
				p.gy = blend(r_gy);
			}
#endif
			p.T = ld(r_t, (uint32_t)i * 4u);
			p.valid = valid;
			return p;
		};
		auto add = [&](const Px &p) {
			const double I = p.I, gx = p.gx, gy = p.gy, T = p.T;
			if (p.valid)
			{
				s[0] += 1.0, s[1] += I, s[2] = __builtin_fma(I, I, s[2]), s[3] += T, s[4] = __builtin_fma(T, T, s[4]), s[5] = __builtin_fma(T, I, s[5]);
				s[6] += gx, s[7] += gy, s[8] = __builtin_fma(gx, gx, s[8]), s[9] = __builtin_fma(gx, gy, s[9]), s[10] = __builtin_fma(gy, gy, s[10]);
				s[11] = __builtin_fma(gx, I, s[11]), s[12] = __builtin_fma(gy, I, s[12]), s[13] = __builtin_fma(gx, T, s[13]), s[14] = __builtin_fma(gy, T, s[14]);
			}
		};
		{
			const int i0 = blk * ECC_BLOCK + (int)threadIdx.x;
			int y = i0 / w, x = i0 - y * w; // (the thread's first pixel; meaningless, and unused, when i0 >= npx)
			for (int i = i0; i < npx; i += R * stride)
			{ // R pixels per round: their 13 R loads are in flight together (a thread has 5 pixels at 640x512); sums are taken in pixel order
				Px px[R];
				int xx = x, yy = y;
#pragma unroll
				for (int q = 0; q < R; ++q)
				{
					const bool in = i + q * stride < npx; // (a pixel past the end is sampled at the round's first pixel and not counted)
					px[q] = sample(in ? i + q * stride : i, in ? xx : x, in ? yy : y, in);
					xx += dx, yy += dy;
					if (xx >= w)
						xx -= w, ++yy;
				}
#pragma unroll
				for (int q = 0; q < R; ++q)
					add(px[q]);
				x = xx, y = yy;
			}
		}
#ifdef RIR_ECC_DIAG
		ecc_diag_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
		// Reduction over the workgroup through LDS, in a fixed order and without cross-lane operations (fifteen 64-bit butterflies of
		// six ds_bpermute steps each took 3.3 us of a 14 us iteration): every thread leaves its 15 sums in LDS; thread t then adds, for
		// sum k = t % 16, the 16 threads of chunk c = t / 16 in order; thread k < 15 finally adds the chunks in order.
		static_assert(ECC_BLOCK % 64 == 0 && ECC_NSUMS <= 16, "16 lanes per chunk, one per sum; a chunk inside one wave");
		constexpr int NCH = ECC_BLOCK / 16;
		{
			const int k = threadIdx.x & 15, c = threadIdx.x >> 4;
#pragma unroll
			for (int half = 0; half < 2; ++half)
			{
#pragma unroll
				for (int kk = 0; kk < 8; ++kk)
					if (half * 8 + kk < ECC_NSUMS)
						red.val[kk][threadIdx.x] = s[half * 8 + kk];
				ecc_wave_sync();
				if ((k >> 3) == half && k < ECC_NSUMS)
				{
					double a = 0.0;
#pragma unroll
					for (int j = 0; j < 16; ++j)
						a += red.val[k & 7][c * 16 + j];
					red.part[parity][k][c] = a;
				}
				ecc_wave_sync();
			}
		}
		ecc_lds_barrier();
#ifdef RIR_ECC_DIAG
		ecc_diag_reduced = __builtin_amdgcn_s_memrealtime();
#endif
		double v = 0.0;
		if (threadIdx.x < ECC_NSUMS)
#pragma unroll 4
			for (int c = 0; c < NCH; ++c)
				v += red.part[parity][threadIdx.x][c];
		return v;
	}

	__global__ __launch_bounds__(ECC_BLOCK) void ecc_sums_kernel(const float *__restrict__ templ, const float *__restrict__ image,
																 const float *__restrict__ gximg, const float *__restrict__ gyimg,
																 const uint8_t *__restrict__ mask, int w, int h, double *__restrict__ partials,
																 const EccState *__restrict__ state)
	{
		if (state->done)
			return;
		__shared__ EccReduceLds red;
		const double v = ecc_block_sums<RIR_ECC_PIXELS_PER_ROUND>(templ, image, gximg, gyimg, mask, w, h, state->tx, state->ty, blockIdx.x, gridDim.x, red, 0);
		if (threadIdx.x < ECC_NSUMS)
			partials[(size_t)blockIdx.x * 16 + threadIdx.x] = v; // rows of 16 doubles (ecc_rows_total)
	}

	// One update of the alignment from the 15 sums (2x2 normal equations of the forward additive ECC scheme); st: tx, ty, rho,
	// last_rho and iter are advanced.  Returns done: 0 = go on, 1 = converged or iteration limit, 2 = failed.
	__device__ __forceinline__ int ecc_solve_step(const double *tot, EccState &st)
	{
		const double n = tot[0];
		int done = 0;
		double rho = -1.0;
		if (n < 1.0)
			done = 2;
		else
		{
			const double mI = tot[1] / n, mT = tot[3] / n;
			const double imgNorm2 = tot[2] - n * mI * mI, tmpNorm2 = tot[4] - n * mT * mT;
			const double corr = tot[5] - n * mT * mI;
			const double h00 = tot[8], h01 = tot[9], h11 = tot[10];
			const double ip0 = tot[11] - mI * tot[6], ip1 = tot[12] - mI * tot[7];
			const double tp0 = tot[13] - mT * tot[6], tp1 = tot[14] - mT * tot[7];
			const double det = h00 * h11 - h01 * h01;
			rho = corr / (sqrt(imgNorm2) * sqrt(tmpNorm2));
			if (!(det != 0.0) || isnan(rho))
				done = 2;
			else
			{
				const double i00 = h11 / det, i01 = -h01 / det, i11 = h00 / det;
				const double iph0 = i00 * ip0 + i01 * ip1, iph1 = i01 * ip0 + i11 * ip1;
				const double lambda_n = imgNorm2 - (ip0 * iph0 + ip1 * iph1);
				const double lambda_d = corr - (tp0 * iph0 + tp1 * iph1);
				if (lambda_d <= 0.0)
					done = 2;
				else
				{
					const double lambda = lambda_n / lambda_d;
					const double e0 = lambda * tp0 - ip0, e1 = lambda * tp1 - ip1;
					st.tx = (float)((double)st.tx + (i00 * e0 + i01 * e1));
					st.ty = (float)((double)st.ty + (i01 * e0 + i11 * e1));
				}
			}
		}
		const double prev = st.rho;
		st.last_rho = prev;
		st.rho = rho;
		st.iter = st.iter + 1;
		if (!done && (st.iter >= st.max_iter || fabs(rho - prev) < st.eps))
			done = 1;
		return done;
	}
	// results first, then (release, system scope) the two words the host polls
	__device__ __forceinline__ void ecc_report(EccHostView *host_view, const EccState &st, int done)
	{
		host_view->tx = st.tx;
		host_view->ty = st.ty;
		host_view->rho = st.rho;
		__threadfence_system();
		// iter and done are one 8-byte word: the host sees both or neither
		static_assert(offsetof(EccHostView, done) == offsetof(EccHostView, iter) + 4 && offsetof(EccHostView, iter) % 8 == 0, "iter | done << 32");
		__hip_atomic_store(reinterpret_cast<unsigned long long *>(const_cast<int *>(&host_view->iter)),
						   (unsigned long long)(unsigned int)st.iter | ((unsigned long long)(unsigned int)done << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}

	// The 15 totals over `nrows` rows of partial sums ([row][16] doubles, word 15 = the row's flag), by the first 256 threads of a
	// workgroup, in a fixed order: thread t adds, for sum k = t % 16, rows c, c + 16, c + 32, ... (c = t / 16) in order, then thread
	// k < 15 adds the 16 chunks in order -> tot[k].  POLL: every row is waited for first (its flag == want; false when a wait gave up).
	// The 15 totals over `nrows` rows of partial sums, by the first 256 threads of a workgroup, in a fixed order: thread t adds, for sum
	// k = t % 16, rows c, c + 16, c + 32, ... (c = t / 16) in order, then thread k < 15 adds the 16 chunks in order -> tot[k].
	// GRANULES false: rows of 16 doubles left by an earlier launch (ecc_sums_kernel).  GRANULES true: rows of 16 granules written
	// inside this launch; each is waited for (its flag == want under kEccFlagMask; false when a wait gave up).
	template <bool GRANULES>
	__device__ __forceinline__ bool ecc_rows_total(const double *rows, int nrows, unsigned long long want, double (*part)[17], double *tot)
	{
		const int t = threadIdx.x, k = t & 15, c = t >> 4;
		bool ok = true;
		double a = 0.0;
		if (t < 256 && k < ECC_NSUMS)
		{
			const __amdgpu_buffer_rsrc_t rs = ecc_rsrc(rows, (uint32_t)nrows * (GRANULES ? 256u : 128u));
			for (int r0 = c; r0 < nrows; r0 += 16 * 16)
			{ // 16 rows of this chunk at a time, all loads in flight together (a load behind a test of the previous one is 16 round trips)
				double v[16];
				if (GRANULES)
				{
					const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
					for (;;)
					{
						ecc_v4u g[16];
#pragma unroll
						for (int j = 0; j < 16; ++j)
							g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)min(r0 + 16 * j, nrows - 1) * 256u + (uint32_t)k * 16u, 0, 16 /* sc1 */);
						unsigned long long bad = 0;
#pragma unroll
						for (int j = 0; j < 16; ++j)
						{
							bad |= ((((unsigned long long)g[j].w << 32) | g[j].z) ^ want) & kEccFlagMask;
							v[j] = __longlong_as_double((long long)(((unsigned long long)g[j].y << 32) | g[j].x));
						}
						if (bad == 0)
							break;
						__builtin_amdgcn_s_sleep(1);
						if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) // 2 s of the 100 MHz clock
						{
							ok = false;
							break;
						}
					}
				}
				else
				{
#pragma unroll
					for (int j = 0; j < 16; ++j)
						v[j] = rows[(size_t)min(r0 + 16 * j, nrows - 1) * 16 + k];
				}
#pragma unroll
				for (int j = 0; j < 16; ++j)
					a += r0 + 16 * j < nrows ? v[j] : 0.0;
			}
			part[k][c] = a;
		}
		const int all_ok = __syncthreads_and(ok ? 1 : 0);
		if (t < ECC_NSUMS)
		{
			double v = 0.0;
#pragma unroll
			for (int cc = 0; cc < 16; ++cc)
				v += part[t][cc];
			tot[t] = v;
		}
		__syncthreads();
		return all_ok != 0;
	}

	__global__ __launch_bounds__(ECC_SOLVE_BLOCK) void ecc_solve_kernel(const double *__restrict__ partials, int nrows, EccState *__restrict__ state,
																		EccHostView *host_view)
	{
		if (state->done)
			return;
		__shared__ double part[ECC_NSUMS][17];
		__shared__ double tot[ECC_NSUMS];
		(void)ecc_rows_total<false>(partials, nrows, 0ull, part, tot);
		if (threadIdx.x != 0)
			return;
		EccState stt = *state;
		const int done = ecc_solve_step(tot, stt);
		state->tx = stt.tx, state->ty = stt.ty;
		state->last_rho = stt.last_rho, state->rho = stt.rho;
		state->iter = stt.iter;
		__threadfence();
		state->done = done;
		if (host_view)
			ecc_report(host_view, stt, done);
	}

	// ---- all iterations of an alignment in ONE launch -----------------------------------------------------------------
	//
	// grid = nblk (<= RIR_ECC_MAX_BLOCKS = 256: one workgroup per CU, all resident), block = 256.  Per iteration every workgroup
	// leaves its row of 15 sums as granules {value, flag}; workgroup 0 waits for all of them, adds
	// the rows in the order ecc_solve_kernel does - the results are the same bits as with two launches per iteration - solves,
	// and publishes the new translation with a flag the other workgroups wait for.  Two hops across the chip per iteration
	// instead of two launch boundaries (15 us -> see DESIGN.md).  Flags carry (epoch << 32 | frame << 20 | iteration): the host passes a new
	// epoch with every launch, nothing has to be cleared.  Waits are bounded by a clock; a wait that gives up ends the
	// alignment as failed (done = 2).
	// rows:  [nblk][16] granules {sum, flag};   pub: one granule {tx | ty << 32, flag | done << 62}
	__global__ __launch_bounds__(ECC_BLOCK) void ecc_run_kernel(const float *__restrict__ templ, const float *__restrict__ image,
																const float *__restrict__ gximg, const float *__restrict__ gyimg, const uint8_t *__restrict__ mask,
																int w, int h, double *rows, unsigned long long *pub, EccState *state, EccHostView *host_view, float tx0,
																float ty0, int max_iter, double eps, unsigned int epoch, int nframes, EccFrameResult *results,
																unsigned int *__restrict__ ctl, unsigned int arrivals_before)
	{
		static_assert(ECC_BLOCK >= ECC_SOLVE_BLOCK, "workgroup 0 adds the rows the way ecc_solve_kernel does: by its first 256 threads");
		__shared__ EccReduceLds red;
		__shared__ unsigned int sh_flag;
		// is the whole launch on the chip (resident_device.h)?  If not, nothing is written but the host's view: done = 3, "not run" -
		// the host then takes the launch-per-iteration kernels, which need no residency
		if (resident_rendezvous(ctl, arrivals_before, gridDim.x, epoch, &sh_flag) != RESIDENT_GO)
		{
			if (blockIdx.x == 0 && threadIdx.x == 0 && host_view)
			{
				EccState none;
				none.tx = tx0, none.ty = ty0, none.rho = 0.0, none.iter = 0;
				ecc_report(host_view, none, 3);
			}
			return;
		}
		__shared__ double part[ECC_NSUMS][17];
		__shared__ double tot[ECC_NSUMS];
		__shared__ float sh_t[2];
		__shared__ int sh_done;
		const int b = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x;
		const __amdgpu_buffer_rsrc_t rows_rs = ecc_rsrc(rows, (uint32_t)nblk * 256u), pub_rs = ecc_rsrc(pub, 16u);
		EccState st; // (workgroup 0, thread 0 keeps the real one)
		st.tx = tx0, st.ty = ty0;
		float tx = tx0, ty = ty0;
		int done = 0, frames_done = 0;
		// a sequence of `nframes` images (w * h floats apart, their gradients likewise), each aligned from where the previous one
		// ended, as a tracked sequence is; it stops at the first alignment that fails
		for (int f = 0; f < nframes && done != 2; ++f, image += (size_t)w * h, gximg += (size_t)w * h, gyimg += (size_t)w * h)
		{
		st.rho = -1.0, st.last_rho = -eps;
		st.iter = 0, st.done = 0, st.ticket = 0;
		st.max_iter = max_iter, st.eps = eps;
		done = 0;
		for (int it = 1; !done; ++it)
		{
			const unsigned long long flag = ((unsigned long long)(epoch & 0x3fffffffu) << 32) | ((unsigned long long)(unsigned int)f << 20) | (unsigned long long)((unsigned int)it & 0xfffffu);
#ifdef RIR_ECC_DIAG
			const unsigned long long dg0 = __builtin_amdgcn_s_memrealtime();
			unsigned long long dg1 = 0, dg2 = 0, dg3 = 0;
#endif
			const double v = ecc_block_sums<RIR_ECC_PIXELS_PER_ROUND>(templ, image, gximg, gyimg, mask, w, h, tx, ty, b, nblk, red, it & 1);
			// hand-off without fences (a release / acquire pair at agent scope writes back and invalidates whole caches: 227 us per
			// frame against 139 with two launches per iteration) and without a drain: every sum travels as a granule {value, flag}
			if (tid < ECC_NSUMS)
				ecc_granule_store(rows_rs, (uint32_t)b * 256u + (uint32_t)tid * 16u, (unsigned long long)__double_as_longlong(v), flag);
#ifdef RIR_ECC_DIAG
			dg1 = __builtin_amdgcn_s_memrealtime();
#endif
			if (b == 0)
			{
				const bool all_ok = ecc_rows_total<true>(rows, nblk, flag, part, tot);
#ifdef RIR_ECC_DIAG
				dg2 = __builtin_amdgcn_s_memrealtime();
#endif
				if (tid == 0)
				{
					done = all_ok ? ecc_solve_step(tot, st) : 2;
					sh_t[0] = st.tx, sh_t[1] = st.ty;
					sh_done = done;
					// one granule: {tx | ty << 32, flag | done << 62}
					ecc_granule_store(pub_rs, 0, (unsigned long long)__float_as_uint(st.tx) | ((unsigned long long)__float_as_uint(st.ty) << 32),
									  flag | ((unsigned long long)done << 62));
					if (host_view && (done || (it & 63) == 0)) // a sign of life for the host (posted write to coherent host memory, nothing waits for it)
						__hip_atomic_store(const_cast<unsigned int *>(&host_view->progress), (unsigned int)((f << 20) + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			}
			else if (tid == 0)
			{
				const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
				sh_done = 2;
				for (;;)
				{
					const ecc_v4u g = __builtin_amdgcn_raw_buffer_load_b128(pub_rs, 0, 0, 16 /* sc1 */);
					const unsigned long long fl = ((unsigned long long)g.w << 32) | g.z;
					if (((fl ^ flag) & kEccFlagMask) == 0)
					{
						sh_t[0] = __uint_as_float(g.x), sh_t[1] = __uint_as_float(g.y);
						sh_done = (int)(fl >> 62);
						break;
					}
					__builtin_amdgcn_s_sleep(1);
					if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) // 2 s of the 100 MHz clock
						break;
				}
			}
			__syncthreads();
			tx = sh_t[0], ty = sh_t[1];
			done = sh_done;
#ifdef RIR_ECC_DIAG
			if (tid == 0 && (b == 0 || b == nblk - 1))
			{ // ticks (10 ns): sums + publish | wait for the rows | add + solve + publish (workgroup 0) or the whole wait (last workgroup)
				dg3 = __builtin_amdgcn_s_memrealtime();
				unsigned long long *dg = pub + 8 + (b == 0 ? 0 : 8);
				dg[0] += dg1 - dg0, dg[1] += (b == 0 ? dg2 : dg3) - dg1, dg[2] += b == 0 ? dg3 - dg2 : 0, dg[3] += 1;
				dg[4] += ecc_diag_loop_end - dg0, dg[5] += ecc_diag_reduced - ecc_diag_loop_end;
			}
#endif
			__syncthreads(); // (red / tot / sh_* are reused by the next iteration)
		}
		if (b == 0 && tid == 0 && results)
		{
			results[f].tx = st.tx, results[f].ty = st.ty;
			results[f].rho = st.rho;
			results[f].iter = st.iter, results[f].done = done;
		}
		frames_done = f + 1;
		}
		if (b == 0 && tid == 0)
		{
			if (results)
				st.iter = frames_done; // (the host is told how many frames were aligned)
			state->tx = st.tx, state->ty = st.ty;
			state->last_rho = st.last_rho, state->rho = st.rho;
			state->iter = st.iter;
			state->max_iter = max_iter, state->eps = eps;
			state->done = done;
			if (host_view)
				ecc_report(host_view, st, done);
		}
	}

	// ---- S independent tracked sequences in ONE launch --------------------------------------------------------------------
	//
	// An alignment is a dependent chain (iteration after iteration, image after image): one sequence cannot use more of the chip
	// than one iteration's pixels, and between two iterations lie two hand-offs across the chip - the rows of partial sums to
	// whoever adds them, the new translation back: 4-5 us in which the sequence's workgroups have nothing to do.  Independent
	// sequences ("R1 ... replicas") are what can run side by side, and what can fill those gaps:
	//   * sequences are taken in PAIRS (group g = sequences 2g, 2g + 1; the last group may hold one).  A group owns `nslices`
	//     COMPUTE workgroups; each computes its rows of the first sequence, leaves them, computes its rows of the second one, and
	//     only then needs the first one's new translation - which has travelled while it worked;
	//   * every sequence has a SERVICE workgroup of its own (the last S workgroups of the launch) that computes no rows: it waits
	//     for the sequence's rows, adds them, solves and publishes the translation - while the compute workgroups are busy with
	//     the other sequence of the pair.  (With the adding done by a compute workgroup, as in the first form of this kernel, the
	//     hand-off starts only when that workgroup's own rows are done: the interleaving then hides nothing.)
	// With 8 sequences a pair's compute workgroups live on two XCDs (workgroup i starts on XCD i % 8), half of its rows in each.
	//
	// The arithmetic is that of ecc_run_kernel with V = ecc_blocks(w, h) workgroups, bit for bit, whatever S and nslices are: the
	// pixels are cut into the same V "rows" (row b = what workgroup b of a solo run sums: pixels b * 256 + t + k * V * 256 of
	// thread t, added in that order, reduced over the 256 threads in the same fixed order), a slice computes rows j, j + nslices,
	// ... one after the other and leaves each as its row of granules, and the service workgroup adds the V rows in the order
	// ecc_solve_kernel does.  Nothing crosses sequences; a sequence whose alignment fails stops, the others go on.
	// rows: per sequence [V][16] granules, then its pub granule (ecc_run_workspace_bytes).
	// wave-uniform values the compiler cannot see are uniform (read from LDS or through a table pointer): into scalar registers, so
	// that the pixel loop keeps the vector registers
	__device__ __forceinline__ int ecc_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
	__device__ __forceinline__ float ecc_uni(float v) { return __uint_as_float((unsigned int)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
	template <class T>
	__device__ __forceinline__ T *ecc_uni(T *p)
	{
		const uint64_t b = (uint64_t)p;
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
		return (T *)(((uint64_t)hi << 32) | lo);
	}
	__device__ __forceinline__ unsigned long long ecc_flag(unsigned int epoch, int f, int it)
	{
		return ((unsigned long long)(epoch & 0x3fffffffu) << 32) | ((unsigned long long)(unsigned int)f << 20) | (unsigned long long)((unsigned int)it & 0xfffffu);
	}
	// The service workgroup of sequence q of a multi-sequence launch: rows in, translation out, image after image.  Out of line: its
	// registers (the 2x2 solve in double precision) are then not the pixel loop's, which is pinned at 96 by five waves per SIMD.
	__device__ __noinline__ void ecc_multi_service(EccSeq *__restrict__ table, int q, int V, int max_iter, double eps, unsigned int epoch, double (*part)[17],
												   double *tot, int *sh_done_p)
	{
		const int tid = threadIdx.x;
		int &sh_done = *sh_done_p;
		__builtin_amdgcn_s_setprio(3); // (it shares its CU with four compute workgroups, and everybody waits for what it does)
		const EccSeq sq = table[q];
		unsigned long long *pub = reinterpret_cast<unsigned long long *>(sq.rows + (size_t)V * 32);
		const __amdgpu_buffer_rsrc_t pub_rs = ecc_rsrc(pub, 16u);
		EccState st; // (thread 0 keeps the real one)
		st.tx = sq.tx0, st.ty = sq.ty0;
		int done = 0, frames_done = 0;
		for (int f = 0; f < sq.nframes && done != 2; ++f)
		{
			st.rho = -1.0, st.last_rho = -eps;
			st.iter = 0, st.done = 0, st.ticket = 0;
			st.max_iter = max_iter, st.eps = eps;
			done = 0;
			for (int it = 1; !done; ++it)
			{
				const unsigned long long flag = ecc_flag(epoch, f, it);
#ifdef RIR_ECC_DIAG
				const unsigned long long dg0 = __builtin_amdgcn_s_memrealtime();
#endif
				// While the rows are being computed only a few of them are watched - every 16th, one granule each, by 16 lanes - and
				// the whole set (61 KB of write-through granules per look) is asked for when those have come: the rows of a turn are
				// finished within a microsecond of each other, and eight service workgroups that look at everything all the time
				// are 0.3 TB/s of traffic past the L2s that the pixel loops feel.
				if (tid < 16)
				{
					const __amdgpu_buffer_rsrc_t rs = ecc_rsrc(sq.rows, (uint32_t)V * 256u);
					const uint32_t off = (uint32_t)min(tid * 16 + 15, V - 1) * 256u;
					const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
					for (;;)
					{
						const ecc_v4u gr = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16 /* sc1 */);
						const bool here = (((((unsigned long long)gr.w << 32) | gr.z) ^ flag) & kEccFlagMask) == 0;
						if (__builtin_amdgcn_ballot_w64(!here) == 0 || __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull)
							break; // (all 16 have come - or 2 s have passed: the full look below has a clock of its own)
						__builtin_amdgcn_s_sleep(2);
					}
				}
				__syncthreads();
				const bool all_ok = ecc_rows_total<true>(sq.rows, V, flag, part, tot);
#ifdef RIR_ECC_DIAG
				const unsigned long long dg1 = __builtin_amdgcn_s_memrealtime();
#endif
				if (tid == 0)
				{
					done = all_ok ? ecc_solve_step(tot, st) : 2;
					sh_done = done;
					ecc_granule_store(pub_rs, 0, (unsigned long long)__float_as_uint(st.tx) | ((unsigned long long)__float_as_uint(st.ty) << 32),
									  flag | ((unsigned long long)done << 62));
#ifdef RIR_ECC_DIAG
					if (q == 0)
					{ // ticks (10 ns): waiting for + adding the rows | solve + publish
						unsigned long long *dg = pub + 8;
						dg[0] += dg1 - dg0, dg[1] += __builtin_amdgcn_s_memrealtime() - dg1, dg[2] += 1;
					}
#endif
				}
				__syncthreads();
				done = sh_done;
				__syncthreads(); // (part / tot / sh_done are reused by the next iteration)
			}
			if (tid == 0)
			{
				EccFrameResult r;
				r.tx = st.tx, r.ty = st.ty, r.rho = st.rho, r.iter = st.iter, r.done = done;
				sq.results[f] = r;
			}
			frames_done = f + 1;
		}
		if (tid == 0)
			table[q].frames_done = frames_done;
	}

	__attribute__((amdgpu_waves_per_eu(RIR_ECC_MULTI_WAVES, RIR_ECC_MULTI_WAVES))) __global__ __launch_bounds__(ECC_BLOCK) void ecc_run_multi_kernel(EccSeq *__restrict__ table, int S, int w, int h, int V, int max_iter, double eps,
																	  unsigned int epoch, unsigned int *__restrict__ ctl, unsigned int arrivals_before,
																	  unsigned int *host_go)
	{
		__shared__ EccReduceLds red;
		__shared__ double part[ECC_NSUMS][17];
		__shared__ double tot[ECC_NSUMS];
		__shared__ float sh_t[2];
		__shared__ int sh_done;
		__shared__ unsigned int sh_flag;
		// is the whole launch on the chip?  (resident_device.h: if not - ordinary kernels of other streams can keep the last workgroups
		// from fitting - everybody leaves before anything is written and the host runs the chunk again with fewer slices)
		if (resident_rendezvous(ctl, arrivals_before, gridDim.x, epoch, &sh_flag) != RESIDENT_GO)
			return;
		const int tid = threadIdx.x;
		// "the launch is resident", for the host (a word of coherent page-locked memory): from here on other kernels may be started
		// beside it - they can no longer keep it from fitting - and the host uses the time for the pre-processing of the next chunk
		if (host_go && blockIdx.x == 0 && tid == 0)
			__hip_atomic_store(host_go, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		const size_t npx = (size_t)w * h;
		const int ncompute = (int)gridDim.x - S;
		if ((int)blockIdx.x >= ncompute)
		{ // ---- the service workgroup of sequence q ----
			ecc_multi_service(table, (int)blockIdx.x - ncompute, V, max_iter, eps, epoch, part, tot, &sh_done);
			return;
		}
		// ---- a compute workgroup of group g: its rows of the group's sequences in turn ----
		constexpr int G = 2;
		// group and slice of compute workgroup c (it starts on XCD c % 8).  A slice's rows are strips of 256 consecutive pixels, slices
		// next to each other read image lines next to each other: where a group spreads over several XCDs (fewer than 8 groups) each
		// XCD gets a run of consecutive slices, so that an L2 holds a band of the group's images instead of most of every image
		const int NG = (S + G - 1) / G, c = (int)blockIdx.x, nslices = ncompute / NG;
		int g = c % NG, slice = c / NG;
		if (8 % NG == 0 && nslices % (8 / NG) == 0)
		{
			const int x = c & 7, m = c >> 3;
			g = x % NG, slice = (x / NG) * (nslices / (8 / NG)) + m;
		}
		struct Turn
		{
			const float *templ, *image, *gx, *gy;
			__amdgpu_buffer_rsrc_t rows_rs, pub_rs;
			float tx, ty;
			int f, it, nframes;
			bool live, asked; // asked: rows of (f, it) are out, the decision on them has not been read yet
		} tn[G];
#pragma unroll
		for (int j = 0; j < G; ++j)
		{
			const int q = g * G + j;
			tn[j].live = q < S;
			const EccSeq sq = table[tn[j].live ? q : g * G];
			tn[j].templ = ecc_uni(sq.templ), tn[j].image = ecc_uni(sq.image), tn[j].gx = ecc_uni(sq.gx), tn[j].gy = ecc_uni(sq.gy);
			tn[j].rows_rs = ecc_rsrc(sq.rows, (uint32_t)V * 256u);
			tn[j].pub_rs = ecc_rsrc(reinterpret_cast<const unsigned long long *>(sq.rows + (size_t)V * 32), 16u);
			tn[j].tx = ecc_uni(sq.tx0), tn[j].ty = ecc_uni(sq.ty0);
			tn[j].f = 0, tn[j].it = 1, tn[j].nframes = ecc_uni(sq.nframes);
			tn[j].live = tn[j].live && tn[j].nframes > 0;
			tn[j].asked = false;
		}
		int red_calls = 0;
		// (Asking for the decision a turn starts with already between the pixel loop and the reduction of the turn before - its answer
		// kept in flight across the reduction's barrier - was tried: no faster; the decision is rarely there that early.)
#ifdef RIR_ECC_DIAG
		unsigned long long dg_rows = 0, dg_wait = 0, dg_n = 0, dg_loop = 0;
#endif
		for (bool any = true; any;)
		{
			any = false;
#pragma unroll
			for (int j = 0; j < G; ++j)
			{
				Turn &t = tn[j];
				if (!t.live)
					continue;
				if (t.asked)
				{ // the decision on the rows this workgroup left a turn ago
#ifdef RIR_ECC_DIAG
					const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
#endif
					const unsigned long long flag = ecc_flag(epoch, t.f, t.it);
					if (tid == 0)
					{
						const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
						sh_done = 2;
						for (;;)
						{
							const ecc_v4u gr = __builtin_amdgcn_raw_buffer_load_b128(t.pub_rs, 0, 0, 16 /* sc1 */);
							const unsigned long long fl = ((unsigned long long)gr.w << 32) | gr.z;
							if (((fl ^ flag) & kEccFlagMask) == 0)
							{
								sh_t[0] = __uint_as_float(gr.x), sh_t[1] = __uint_as_float(gr.y);
								sh_done = (int)(fl >> 62);
								break;
							}
							__builtin_amdgcn_s_sleep(1);
							if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) // 2 s of the 100 MHz clock
								break;
						}
					}
					__syncthreads();
					const int done = ecc_uni(sh_done);
					if (done != 2)
						t.tx = ecc_uni(sh_t[0]), t.ty = ecc_uni(sh_t[1]);
					__syncthreads(); // (sh_* are reused by the next turn)
#ifdef RIR_ECC_DIAG
					dg_wait += __builtin_amdgcn_s_memrealtime() - w0;
#endif
					t.asked = false;
					if (done == 0)
						++t.it;
					else if (done == 1)
					{ // the image is aligned: the next one starts from its translation
						++t.f, t.it = 1;
						t.image += npx, t.gx += npx, t.gy += npx;
					}
					if (done == 2 || t.f >= t.nframes)
					{
						t.live = false;
						continue;
					}
				}
				any = true;
#ifdef RIR_ECC_DIAG
				const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#endif
				const unsigned long long flag = ecc_flag(epoch, t.f, t.it);
				for (int b = slice; b < V; b += nslices)
				{
					const double v = ecc_block_sums<RIR_ECC_MULTI_PIXELS_PER_ROUND>(t.templ, t.image, t.gx, t.gy, nullptr, w, h, t.tx, t.ty, b, V, red, (red_calls++) & 1);
					if (tid < ECC_NSUMS)
						ecc_granule_store(t.rows_rs, (uint32_t)b * 256u + (uint32_t)tid * 16u, (unsigned long long)__double_as_longlong(v), flag);
#ifdef RIR_ECC_DIAG
					dg_loop += ecc_diag_loop_end - r0; // (one row per turn: r0 is the row's start)
#endif
				}
				t.asked = true;
#ifdef RIR_ECC_DIAG
				dg_rows += __builtin_amdgcn_s_memrealtime() - r0, dg_n += 1;
#endif
			}
		}
#ifdef RIR_ECC_DIAG
		if (g == 0 && tid == 0 && (slice == 0 || slice == nslices - 1))
		{ // ticks (10 ns), per turn: rows | waiting for a decision
			unsigned long long *dg = reinterpret_cast<unsigned long long *>(table[0].rows + (size_t)V * 32) + 8 + (slice == 0 ? 4 : 8);
			dg[0] += dg_rows, dg[1] += dg_wait, dg[2] += dg_n, dg[3] += dg_loop;
		}
#endif
	}

	int ecc_run_grid(int w, int h) { return ecc_blocks(w, h); }
	int ecc_run_multi_capacity() { return resident_capacity(reinterpret_cast<const void *>(ecc_run_multi_kernel), ECC_BLOCK, 0, RIR_ECC_MULTI_MARGIN != 0); }
	int ecc_rows(int w, int h) { return ecc_blocks(w, h); }
	// d_table: nseq entries (device); nslices compute workgroups per PAIR of sequences (1 .. ecc_rows(w, h)), one service workgroup per
	// sequence: ecc_run_multi_grid(nseq, nslices) <= ecc_run_multi_capacity()
	int ecc_run_multi_grid(int nseq, int nslices) { return nseq + (nseq + 1) / 2 * nslices; }
	hipError_t launch_ecc_run_multi(EccSeq *d_table, int nseq, int nslices, int w, int h, int max_iter, double eps, unsigned int epoch, unsigned int *d_ctl,
									unsigned int arrivals_before, unsigned int *host_go, hipStream_t st)
	{
		const int V = ecc_blocks(w, h);
		if (nseq <= 0 || nslices <= 0 || nslices > V || ecc_run_multi_grid(nseq, nslices) > ecc_run_multi_capacity())
			return hipErrorInvalidConfiguration;
		ResidentGate gate(st); // its workgroups wait for each other: not beside any other resident launch of the process
		if (!gate.ok())
			return hipErrorUnknown;
		hipLaunchKernelGGL(ecc_run_multi_kernel, dim3((unsigned)ecc_run_multi_grid(nseq, nslices)), dim3(ECC_BLOCK), 0, st, d_table, nseq, w, h, V, max_iter, eps,
						   epoch, d_ctl, arrivals_before, host_go);
		return hipGetLastError();
	}

	hipError_t launch_ecc_prepare(const float *d_image, int w, int h, float *d_gx, float *d_gy, EccState *d_state, float tx, float ty, int max_iter,
								  double eps, hipStream_t st)
	{
		hipLaunchKernelGGL(ecc_gradient_kernel, dim3((w * h + 255) / 256), dim3(256), 0, st, d_image, w, h, d_gx, d_gy, d_state, tx, ty, max_iter, eps);
		return hipGetLastError();
	}

	hipError_t launch_ecc_iterate(const float *d_templ, const float *d_image, const float *d_gx, const float *d_gy, const uint8_t *d_mask, int w,
								  int h, double *d_partials, EccState *d_state, EccHostView *host_view, hipStream_t st)
	{
		const int nblk = ecc_blocks(w, h);
		hipLaunchKernelGGL(ecc_sums_kernel, dim3(nblk), dim3(ECC_BLOCK), 0, st, d_templ, d_image, d_gx, d_gy, d_mask, w, h, d_partials, d_state);
		hipLaunchKernelGGL(ecc_solve_kernel, dim3(1), dim3(ECC_SOLVE_BLOCK), 0, st, d_partials, nblk, d_state, host_view);
		return hipGetLastError();
	}
	// workgroups of ecc_run_kernel the current device holds at once (runtime.h; 0 = unknown): the one-launch form is only taken when
	// the alignment's grid fits - otherwise two launches per iteration (same sums in the same order, same results)
	int ecc_run_capacity() { return resident_capacity(reinterpret_cast<const void *>(ecc_run_kernel), ECC_BLOCK, 0); }
	bool ecc_run_fits(int w, int h) { return ecc_blocks(w, h) <= ecc_run_capacity(); }
	size_t ecc_run_workspace_bytes(int w, int h) { return (size_t)ecc_blocks(w, h) * 256 + 256; } // rows of 16 granules, then pub (+ diagnostics)
	hipError_t launch_ecc_run(const float *d_templ, const float *d_image, const float *d_gx, const float *d_gy, const uint8_t *d_mask, int w, int h,
							  double *d_rows, EccState *d_state, EccHostView *host_view, float tx, float ty, int max_iter, double eps, unsigned int epoch,
							  int nframes, EccFrameResult *d_results, unsigned int *d_ctl, unsigned int arrivals_before, hipStream_t st)
	{
		const int nblk = ecc_blocks(w, h);
		if (nblk > ecc_run_capacity())
			return hipErrorInvalidConfiguration; // (the callers ask ecc_run_fits() first: never reached)
		ResidentGate gate(st); // its workgroups wait for each other: not beside any other resident launch of the process
		if (!gate.ok())
			return hipErrorUnknown;
		unsigned long long *pub = reinterpret_cast<unsigned long long *>(d_rows + (size_t)nblk * 32);
		hipLaunchKernelGGL(ecc_run_kernel, dim3(nblk), dim3(ECC_BLOCK), 0, st, d_templ, d_image, d_gx, d_gy, d_mask, w, h, d_rows, pub, d_state, host_view, tx,
						   ty, max_iter, eps, epoch, nframes, d_results, d_ctl, arrivals_before);
		return hipGetLastError();
	}
} // namespace rir

// ---- min-max normalisation (MaskedRegistratorECC.compute, masked_registration_ecc.py:162-166) ----------------
// im = (im - min) / (max - min) in float32, exactly the two numpy operations of the reference; the crop to the
// registration window is folded into the read (row stride `src_stride` elements, output dense w x h).
namespace rir
{
	__global__ __launch_bounds__(256) void minmax_partial_kernel(const float *__restrict__ src, int w, int h, int src_stride, float *__restrict__ part)
	{
		float mn = 3.402823466e38f, mx = -3.402823466e38f;
		const int n = w * h;
		for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
		{
			const int y = i / w, x = i - y * w;
			const float v = src[(int64_t)y * src_stride + x];
			mn = fminf(mn, v);
			mx = fmaxf(mx, v);
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
		{
			mn = fminf(mn, __shfl_xor(mn, d, 64));
			mx = fmaxf(mx, __shfl_xor(mx, d, 64));
		}
		__shared__ float smn[4], smx[4];
		if ((threadIdx.x & 63) == 0)
			smn[threadIdx.x >> 6] = mn, smx[threadIdx.x >> 6] = mx;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			part[2 * blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
			part[2 * blockIdx.x + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
		}
	}

	__global__ __launch_bounds__(256) void minmax_apply_kernel(const float *__restrict__ src, int w, int h, int src_stride, const float *__restrict__ part,
															   int nparts, float *__restrict__ dst)
	{
		// min / max of the parts: lane k of every wave takes part k (k + 64, ...), then a butterfly (min and max do not depend on the order)
		float mn = 3.402823466e38f, mx = -3.402823466e38f;
		for (int k = threadIdx.x & 63; k < nparts; k += 64)
		{
			mn = fminf(mn, part[2 * k]);
			mx = fmaxf(mx, part[2 * k + 1]);
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
		{
			mn = fminf(mn, __shfl_xor(mn, d, 64));
			mx = fmaxf(mx, __shfl_xor(mx, d, 64));
		}
		const float range = mx - mn;
		const int n = w * h;
		for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
		{
			const int y = i / w, x = i - y * w;
			dst[i] = (src[(int64_t)y * src_stride + x] - mn) / range;
		}
	}

	// the first step for `nframes` images (blockIdx.y = image; images `src_frame` elements apart): min / max are the same whatever the order
	__global__ __launch_bounds__(256) void minmax_partial_frames_kernel(const float *__restrict__ src, int w, int h, int src_stride, int64_t src_frame,
																		float *__restrict__ part)
	{
		src += (size_t)blockIdx.y * src_frame;
		part += (size_t)blockIdx.y * 2 * gridDim.x;
		float mn = 3.402823466e38f, mx = -3.402823466e38f;
		// part p takes the rows p, p + parts, ...: no division per pixel (min and max do not care how the pixels are dealt)
		for (int y = blockIdx.x; y < h; y += gridDim.x)
		{
			const float *row = src + (int64_t)y * src_stride;
			for (int x = threadIdx.x; x < w; x += 256)
			{
				const float v = row[x];
				mn = fminf(mn, v);
				mx = fmaxf(mx, v);
			}
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
		{
			mn = fminf(mn, __shfl_xor(mn, d, 64));
			mx = fmaxf(mx, __shfl_xor(mx, d, 64));
		}
		__shared__ float smn[4], smx[4];
		if ((threadIdx.x & 63) == 0)
			smn[threadIdx.x >> 6] = mn, smx[threadIdx.x >> 6] = mx;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			part[2 * blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
			part[2 * blockIdx.x + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
		}
	}
	// Normalisation AND gradients of the normalised image in one pass (blockIdx.y = image): neighbours are normalised again from the
	// source - the same two float operations, the same values ecc_gradient_kernel would read back.  A wave owns a tile of 64 columns x
	// kGradRows rows, a lane one column of it: it normalises its kGradRows + 2 pixels once (1.25 divisions per pixel instead of 5 - the
	// kernel was bound by them: 53 us for 32 images of 640x512), takes the vertical neighbours from its own registers and the
	// horizontal ones from the lanes beside it; lanes at a tile's or the image's edge load theirs.
	constexpr int kGradRows = 8;
	__global__ __launch_bounds__(256) void minmax_apply_grad_frames_kernel(const float *__restrict__ src, int w, int h, int src_stride, int64_t src_frame,
																		   const float *__restrict__ part, int nparts, float *__restrict__ dst, float *__restrict__ gxs,
																		   float *__restrict__ gys)
	{
		src += (size_t)blockIdx.y * src_frame;
		part += (size_t)blockIdx.y * 2 * nparts;
		const size_t base = (size_t)blockIdx.y * w * h;
		const int lane = threadIdx.x & 63, tiles_x = (w + 63) / 64, tiles_y = (h + kGradRows - 1) / kGradRows;
		const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
		if (tile >= tiles_x * tiles_y)
			return; // (whole waves)
		float mn = 3.402823466e38f, mx = -3.402823466e38f;
		for (int k = lane; k < nparts; k += 64)
		{
			mn = fminf(mn, part[2 * k]);
			mx = fmaxf(mx, part[2 * k + 1]);
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
		{
			mn = fminf(mn, __shfl_xor(mn, d, 64));
			mx = fmaxf(mx, __shfl_xor(mx, d, 64));
		}
		const float range = mx - mn;
		auto norm = [&](int x, int y) { return (src[(int64_t)y * src_stride + x] - mn) / range; };
		const int by = tile / tiles_x, bx = tile - by * tiles_x, x = bx * 64 + lane, y0 = by * kGradRows;
		const int xc = min(x, w - 1);
		// reflect-101 neighbours (w, h >= 2)
		const int xl = xc > 0 ? xc - 1 : 1, xr = xc < w - 1 ? xc + 1 : w - 2;
		const bool own_l = lane == 0 || xc == 0, own_r = lane == 63 || xc >= w - 1;
		float c[kGradRows + 2]; // rows y0 - 1 .. y0 + kGradRows of this column
#pragma unroll
		for (int r = 0; r < kGradRows + 2; ++r)
		{
			const int yy = y0 - 1 + r;
			c[r] = norm(xc, yy < 0 ? 1 : (yy < h ? yy : (yy == h ? h - 2 : h - 1)));
		}
#pragma unroll
		for (int r = 1; r <= kGradRows; ++r)
		{
			const int y = y0 + r - 1;
			if (y >= h)
				break; // (uniform)
			float l = __shfl_up(c[r], 1, 64), rr = __shfl_down(c[r], 1, 64);
			if (own_l)
				l = norm(xl, y);
			if (own_r)
				rr = norm(xr, y);
			if (x < w)
			{
				const size_t o = base + (size_t)y * w + x;
				dst[o] = c[r];
				gxs[o] = 0.5f * rr - 0.5f * l;
				gys[o] = 0.5f * c[r + 1] - 0.5f * c[r - 1];
			}
		}
	}
	hipError_t launch_minmax_normalize_grad_frames(const float *d_src, int w, int h, int src_stride, int64_t src_frame, int nframes, float *d_dst, float *d_gx,
												   float *d_gy, float *d_part, hipStream_t st)
	{
		const int nparts = nframes == 1 ? kMinMaxParts : kMinMaxPartsFrames;
		hipLaunchKernelGGL(minmax_partial_frames_kernel, dim3(nparts, nframes), dim3(256), 0, st, d_src, w, h, src_stride, src_frame, d_part);
		const int tiles = ((w + 63) / 64) * ((h + kGradRows - 1) / kGradRows);
		hipLaunchKernelGGL(minmax_apply_grad_frames_kernel, dim3((tiles + 3) / 4, nframes), dim3(256), 0, st, d_src, w, h, src_stride, src_frame, d_part, nparts,
						   d_dst, d_gx, d_gy);
		return hipGetLastError();
	}

	hipError_t launch_minmax_normalize(const float *d_src, int w, int h, int src_stride, float *d_dst, float *d_part, hipStream_t st)
	{
		const int nparts = kMinMaxParts; // (one image is a small job: 64 workgroups left it on a quarter of the chip, 9 us)
		hipLaunchKernelGGL(minmax_partial_kernel, dim3(nparts), dim3(256), 0, st, d_src, w, h, src_stride, d_part);
		hipLaunchKernelGGL(minmax_apply_kernel, dim3((w * h + 255) / 256 < 1024 ? (w * h + 255) / 256 : 1024), dim3(256), 0, st, d_src, w, h, src_stride,
						   d_part, nparts, d_dst);
		return hipGetLastError();
	}
} // namespace rir
