// Lossless block codec for 16-bit IR frames — gfx950 (CDNA4) kernels.  Format "RIRB1", DESIGN.md §3.
//
// Replaces, behind the librir C ABI, what libx264 does for the reference saver/loader
// (reference: src/cpp/video_io/h264.cpp:1022-1131 AddFrame, :3096-3229 GetFrame; byte-plane
// split :1066-1082 / merge :3016-3051).  The bitstream is this build's own; the parity contract is
// the reference tests' identity decode(encode(x)) == x plus bit-exact equality with the CPU
// restatement in oracle/rir_oracle.c.
//
// Mapping to the machine
//   * one 64-lane wavefront owns one TILE of 512 consecutive pixels (1 KiB: one 16-byte load per
//     lane, fully coalesced) for every frame of a chunk, and walks the time axis in registers:
//     the previous frame never leaves VGPRs, so every raw pixel crosses HBM exactly once;
//   * the residual bit-planes are transposed with wave ballots: plane word (j,b) bit l = bit b of
//     residual 8l+j — a v_cmp writes it straight into an SGPR pair, all control flow is
//     wave-uniform, no LDS, no barriers;
//   * records are written with one coalesced 8-byte-per-lane store (<= 129 words);
//   * 4 independent waves per 256-thread workgroup; grid = (ntiles/4) x nchunks >> 256 CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec_format.h"

namespace rir
{

	// ---- wave helpers ---------------------------------------------------------------------

	// OR-reduce a dword over the 64 lanes; the result is wave-uniform.  DPP row shifts inside the
	// four 16-lane rows, then the two row broadcasts (gfx9/CDNA wave64 idiom); lane 63 holds the total.
	__device__ __forceinline__ uint32_t wave_or(uint32_t v)
	{
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); // row_shr:2
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); // row_shr:4
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); // row_shr:8
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true); // row_bcast:15
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true); // row_bcast:31
		return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
	}

	// inclusive add-scan over the 64 lanes (Hillis-Steele on ds_bpermute; used once per 64 frames)
	__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v, int lane)
	{
#pragma unroll
		for (int d = 1; d < 64; d <<= 1)
		{
			uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
			if (lane >= d)
				v += o;
		}
		return v;
	}

	__device__ __forceinline__ uint32_t bitlen32(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

	// packed 2 x u16 helpers (low half = even pixel)
	__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)
	{
		return ((a - (b & 0xffffu)) & 0xffffu) | ((a & 0xffff0000u) - (b & 0xffff0000u));
	}
	__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b)
	{
		return ((a + b) & 0xffffu) | ((a & 0xffff0000u) + (b & 0xffff0000u));
	}
	__device__ __forceinline__ uint32_t pk_zigzag16(uint32_t d)
	{
		// per half: (d << 1) ^ (d >>arith 15)
		uint32_t sh = (d << 1) & 0xfffefffeu;
		uint32_t sg = ((d >> 15) & 0x00010001u) * 0xffffu;
		return sh ^ sg;
	}
	__device__ __forceinline__ uint32_t pk_unzigzag16(uint32_t z)
	{
		uint32_t sh = (z >> 1) & 0x7fff7fffu;
		uint32_t sg = (z & 0x00010001u) * 0xffffu;
		return sh ^ sg;
	}

	struct Px8
	{
		uint32_t d[4]; // 8 x u16, d[k] = pixels 2k (low) and 2k+1 (high)
	};

	// 8 consecutive pixels of frame `f` starting at flat index p0 (zero past the end of the frame)
	__device__ __forceinline__ Px8 load8(const uint16_t *__restrict__ frames, int64_t f, int64_t npx, int64_t p0, bool vec_ok)
	{
		Px8 r;
		const uint16_t *src = frames + f * npx + p0;
		if (vec_ok && p0 + 8 <= npx)
		{
			uint4 v = *reinterpret_cast<const uint4 *>(src);
			r.d[0] = v.x;
			r.d[1] = v.y;
			r.d[2] = v.z;
			r.d[3] = v.w;
		}
		else
		{
#pragma unroll
			for (int k = 0; k < 4; ++k)
			{
				uint32_t lo = (p0 + 2 * k < npx) ? src[2 * k] : 0u;
				uint32_t hi = (p0 + 2 * k + 1 < npx) ? src[2 * k + 1] : 0u;
				r.d[k] = lo | (hi << 16);
			}
		}
		return r;
	}

	__device__ __forceinline__ void store8(uint16_t *__restrict__ frames, int64_t f, int64_t npx, int64_t p0, bool vec_ok, const Px8 &r)
	{
		uint16_t *dst = frames + f * npx + p0;
		if (vec_ok && p0 + 8 <= npx)
		{
			uint4 v;
			v.x = r.d[0];
			v.y = r.d[1];
			v.z = r.d[2];
			v.w = r.d[3];
			*reinterpret_cast<uint4 *>(dst) = v;
		}
		else
		{
#pragma unroll
			for (int k = 0; k < 4; ++k)
			{
				if (p0 + 2 * k < npx)
					dst[2 * k] = (uint16_t)(r.d[k] & 0xffffu);
				if (p0 + 2 * k + 1 < npx)
					dst[2 * k + 1] = (uint16_t)(r.d[k] >> 16);
			}
		}
	}

	// three lane-registers hold one record: word p lives in lane (p & 63) of register (p >> 6)
	struct RecRegs
	{
		uint32_t lo[3], hi[3];
	};

	// place the wave-uniform word m at record position p (clang has no writelane builtin: a
	// lane-id compare + two selects; the compare is against a uniform position)
	__device__ __forceinline__ void rec_put(RecRegs &r, int lane, int p, uint64_t m)
	{
		const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
		const bool me = lane == (p & 63);
		if (p < 64)
		{
			r.lo[0] = me ? lo : r.lo[0];
			r.hi[0] = me ? hi : r.hi[0];
		}
		else if (p < 128)
		{
			r.lo[1] = me ? lo : r.lo[1];
			r.hi[1] = me ? hi : r.hi[1];
		}
		else
		{
			r.lo[2] = me ? lo : r.lo[2];
			r.hi[2] = me ? hi : r.hi[2];
		}
	}

	__device__ __forceinline__ uint64_t rec_get(const RecRegs &r, int p)
	{
		const int l = p & 63;
		uint32_t lo, hi;
		if (p < 64)
		{
			lo = (uint32_t)__builtin_amdgcn_readlane((int)r.lo[0], l);
			hi = (uint32_t)__builtin_amdgcn_readlane((int)r.hi[0], l);
		}
		else if (p < 128)
		{
			lo = (uint32_t)__builtin_amdgcn_readlane((int)r.lo[1], l);
			hi = (uint32_t)__builtin_amdgcn_readlane((int)r.hi[1], l);
		}
		else
		{
			lo = (uint32_t)__builtin_amdgcn_readlane((int)r.lo[2], l);
			hi = (uint32_t)__builtin_amdgcn_readlane((int)r.hi[2], l);
		}
		return (uint64_t)lo | ((uint64_t)hi << 32);
	}

	// Bit-plane transpose of the 8 residual slots held by the wave (z: packed pairs) into `rec`,
	// planes [0, wt) of every slot are ballotted, the per-slot widths come out of the ballots.
	// Returns the record length in words (header included), 0 when every residual is zero.
	__device__ __forceinline__ int pack_record(const Px8 &z, uint32_t wt, uint32_t mode, int lane, RecRegs &rec)
	{
		if (wt == 0)
			return 0;
		int k = 1;
		uint64_t hdr = (uint64_t)mode << 5;
#pragma unroll
		for (int j = 0; j < 8; ++j)
		{
			const uint32_t v = (j & 1) ? (z.d[j >> 1] >> 16) : (z.d[j >> 1] & 0xffffu);
			uint32_t wj = 0;
			for (uint32_t b = 0; b < wt; ++b)
			{
				const uint64_t m = __ballot((v >> b) & 1u);
				rec_put(rec, lane, k + (int)b, m);
				if (m)
					wj = b + 1;
			}
			hdr |= (uint64_t)wj << (8 * j);
			k += (int)wj;
		}
		rec_put(rec, lane, 0, hdr);
		return k;
	}

	__device__ __forceinline__ uint32_t lane_or8(const Px8 &z)
	{
		uint32_t o = (z.d[0] | z.d[1]) | (z.d[2] | z.d[3]);
		return (o | (o >> 16)) & 0xffffu;
	}

	// left-delta residual inside a tile (MODE_LEFT): z[i] = zigzag(p[i] - p[i-1]), p[-1] = 0
	__device__ __forceinline__ Px8 left_residual(const Px8 &cur, int lane)
	{
		// previous pixel of this lane's first pixel = last pixel of lane-1
		uint32_t prev_last = (uint32_t)__shfl_up((int)(cur.d[3] >> 16), 1, 64);
		if (lane == 0)
			prev_last = 0;
		Px8 sh; // sh = pixels shifted right by one position
		sh.d[0] = (cur.d[0] << 16) | prev_last;
		sh.d[1] = (cur.d[1] << 16) | (cur.d[0] >> 16);
		sh.d[2] = (cur.d[2] << 16) | (cur.d[1] >> 16);
		sh.d[3] = (cur.d[3] << 16) | (cur.d[2] >> 16);
		Px8 z;
#pragma unroll
		for (int k = 0; k < 4; ++k)
			z.d[k] = pk_zigzag16(pk_sub16(cur.d[k], sh.d[k]));
		return z;
	}

	// ---- encode -----------------------------------------------------------------------------
	//
	// grid  = (ceil(ntiles/4), nchunks), block = 256 (4 independent waves)
	// sizes     [nchunks][ntiles][gop]   u8   record words
	// seg_words [nchunks][ntiles]        u32  segment length (sum over the chunk's frames)
	// sparse    [nchunks][ntiles][gop*129] u64, only the first seg_words words of a slot are written
	__global__ __launch_bounds__(256) void rirb1_encode_tiles(const uint16_t *__restrict__ frames, int64_t npx, int ntiles,
															 int nframes, int gop, uint8_t *__restrict__ sizes,
															 uint32_t *__restrict__ seg_words, uint64_t *__restrict__ sparse)
	{
		const int lane = threadIdx.x & 63;
		const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
		if (tile >= ntiles)
			return;
		const int chunk = blockIdx.y;
		const int f_begin = chunk * gop;
		const int nf = min(gop, nframes - f_begin);
		const bool vec_ok = (npx & 7) == 0;
		const int64_t p0 = (int64_t)tile * RIRB1_TILE_PX + lane * 8;
		const int64_t slot = (int64_t)chunk * ntiles + tile;
		uint8_t *my_sizes = sizes + slot * gop;
		uint64_t *out = sparse + slot * (int64_t)gop * RIRB1_REC_MAX_WORDS;

		Px8 prev;
		prev.d[0] = prev.d[1] = prev.d[2] = prev.d[3] = 0;
		Px8 cur = load8(frames, f_begin, npx, p0, vec_ok);
		uint32_t pos = 0;
		uint32_t sz_reg = 0; // lane (f & 63) keeps the size of frame f until the 64-frame flush
		for (int f = 0; f < nf; ++f)
		{
			Px8 nxt = cur;
			if (f + 1 < nf)
				nxt = load8(frames, f_begin + f + 1, npx, p0, vec_ok); // prefetch, consumed next iteration

			Px8 z;
			uint32_t mode, wt;
			if (f == 0)
			{ // key frame: RAW, or LEFT when strictly smaller
				const uint32_t wt_raw = bitlen32(wave_or(lane_or8(cur)));
				Px8 zl = left_residual(cur, lane);
				const uint32_t wt_left = bitlen32(wave_or(lane_or8(zl)));
				// exact sizes need the per-slot widths: OR-reduce the 4 packed dwords of both candidates
				uint32_t tot_raw = 0, tot_left = 0;
#pragma unroll
				for (int k = 0; k < 4; ++k)
				{
					const uint32_t a = wave_or(cur.d[k]), b = wave_or(zl.d[k]);
					tot_raw += bitlen32(a & 0xffffu) + bitlen32(a >> 16);
					tot_left += bitlen32(b & 0xffffu) + bitlen32(b >> 16);
				}
				if (tot_left < tot_raw)
				{
					z = zl;
					mode = RIRB1_MODE_LEFT;
					wt = wt_left;
				}
				else
				{
					z = cur;
					mode = RIRB1_MODE_RAW;
					wt = wt_raw;
				}
			}
			else
			{
#pragma unroll
				for (int k = 0; k < 4; ++k)
					z.d[k] = pk_zigzag16(pk_sub16(cur.d[k], prev.d[k]));
				mode = RIRB1_MODE_TEMPORAL;
				wt = bitlen32(wave_or(lane_or8(z)));
			}

			RecRegs rec;
#pragma unroll
			for (int r = 0; r < 3; ++r)
				rec.lo[r] = rec.hi[r] = 0;
			const int words = pack_record(z, wt, mode, lane, rec);

			uint64_t *dst = out + pos;
			if (lane < words)
				dst[lane] = (uint64_t)rec.lo[0] | ((uint64_t)rec.hi[0] << 32);
			if (words > 64)
			{
				if (lane + 64 < words)
					dst[lane + 64] = (uint64_t)rec.lo[1] | ((uint64_t)rec.hi[1] << 32);
				if (lane + 128 < words)
					dst[lane + 128] = (uint64_t)rec.lo[2] | ((uint64_t)rec.hi[2] << 32);
			}
			if (lane == (f & 63))
				sz_reg = (uint32_t)words;
			if ((f & 63) == 63 || f == nf - 1)
			{ // one coalesced byte store per 64 frames
				const int fb = f & ~63;
				if (fb + lane <= f)
					my_sizes[fb + lane] = (uint8_t)sz_reg;
			}
			pos += (uint32_t)words;
			prev = cur;
			cur = nxt;
		}
		for (int f = nf + lane; f < gop; f += 64)
			my_sizes[f] = 0; // short last chunk: the unused table entries are defined
		if (lane == 0)
			seg_words[slot] = pos;
	}

	// ---- offsets ------------------------------------------------------------------------------
	// grid = nchunks, block = 256.  tile_off[c][0..ntiles] = exclusive scan of seg_words[c][*],
	// chunk_words[c] = total.
	__global__ __launch_bounds__(256) void rirb1_scan_tiles(const uint32_t *__restrict__ seg_words, int ntiles,
														   uint32_t *__restrict__ tile_off, uint64_t *__restrict__ chunk_words)
	{
		__shared__ uint32_t wave_tot[4];
		__shared__ uint32_t carry_s;
		const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
		const uint32_t *in = seg_words + (int64_t)c * ntiles;
		uint32_t *off = tile_off + (int64_t)c * (ntiles + 1);
		if (tid == 0)
			carry_s = 0;
		__syncthreads();
		for (int base = 0; base < ntiles; base += 256)
		{
			const int i = base + tid;
			const uint32_t v = i < ntiles ? in[i] : 0u;
			const uint32_t inc = wave_scan_add(v, lane);
			if (lane == 63)
				wave_tot[wv] = inc;
			__syncthreads();
			uint32_t pre = carry_s;
			for (int k = 0; k < wv; ++k)
				pre += wave_tot[k];
			if (i < ntiles)
				off[i] = pre + inc - v;
			__syncthreads();
			if (tid == 255)
				carry_s = pre + inc;
			__syncthreads();
		}
		if (tid == 0)
		{
			off[ntiles] = carry_s;
			chunk_words[c] = carry_s;
		}
	}

	// grid = (ntiles, nchunks), block = 256: gathers the sparse slots into the dense stream and
	// publishes chunk_off[c] (first word of chunk c in the stream; chunk_off[nchunks] = total).
	__global__ __launch_bounds__(256) void rirb1_compact(const uint64_t *__restrict__ sparse, const uint32_t *__restrict__ tile_off,
														const uint64_t *__restrict__ chunk_words, int ntiles, int nchunks, int gop,
														uint64_t *__restrict__ chunk_off, uint64_t *__restrict__ stream)
	{
		__shared__ uint64_t base_s;
		const int t = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
		if (tid < 64)
		{ // first wave: sum of the preceding chunks' lengths
			uint64_t s = 0;
			for (int i = tid; i < c; i += 64)
				s += chunk_words[i];
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1)
				s += (uint64_t)__shfl_xor((long long)s, d, 64);
			if (tid == 0)
			{
				base_s = s;
				if (t == 0)
				{
					chunk_off[c] = s;
					if (c == nchunks - 1)
						chunk_off[nchunks] = s + chunk_words[c];
				}
			}
		}
		__syncthreads();
		const uint32_t *off = tile_off + (int64_t)c * (ntiles + 1);
		const uint32_t o0 = off[t], n = off[t + 1] - o0;
		const uint64_t *src = sparse + ((int64_t)c * ntiles + t) * (int64_t)gop * RIRB1_REC_MAX_WORDS;
		uint64_t *dst = stream + base_s + o0;
		for (uint32_t i = tid; i < n; i += 256)
			dst[i] = src[i];
	}

	// ---- decode -----------------------------------------------------------------------------
	// grid = (ceil(ntiles/4), nchunks), block = 256 (4 independent waves)
	__global__ __launch_bounds__(256) void rirb1_decode_tiles(const uint8_t *__restrict__ sizes, const uint32_t *__restrict__ tile_off,
															 const uint64_t *__restrict__ chunk_off, const uint64_t *__restrict__ stream,
															 int64_t npx, int ntiles, int nframes, int gop, uint16_t *__restrict__ frames,
															 int *__restrict__ error_flag)
	{
		const int lane = threadIdx.x & 63;
		const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
		if (tile >= ntiles)
			return;
		const int chunk = blockIdx.y;
		const int f_begin = chunk * gop;
		const int nf = min(gop, nframes - f_begin);
		const bool vec_ok = (npx & 7) == 0;
		const int64_t p0 = (int64_t)tile * RIRB1_TILE_PX + lane * 8;
		const int64_t slot = (int64_t)chunk * ntiles + tile;
		const uint8_t *my_sizes = sizes + slot * gop;
		const uint64_t *in = stream + chunk_off[chunk] + tile_off[(int64_t)chunk * (ntiles + 1) + tile];
		const uint32_t seg_len = tile_off[(int64_t)chunk * (ntiles + 1) + tile + 1] - tile_off[(int64_t)chunk * (ntiles + 1) + tile];

		Px8 prev;
		prev.d[0] = prev.d[1] = prev.d[2] = prev.d[3] = 0;
		uint32_t round_base = 0; // words consumed by the previous 64-frame rounds
		for (int f0 = 0; f0 < nf; f0 += 64)
		{
			const int nr = min(64, nf - f0);
			const uint32_t my_sz = lane < nr ? (uint32_t)my_sizes[f0 + lane] : 0u;
			const uint32_t my_end = wave_scan_add(my_sz, lane);
			const uint32_t my_off = round_base + my_end - my_sz;

			// record of the first frame of the round
			uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)my_sz, 0);
			uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)my_off, 0);
			if (o + s > seg_len)
			{ // malformed tables: never read outside the segment
				if (lane == 0)
					atomicExch(error_flag, 1);
				return;
			}
			uint64_t w0 = (uint32_t)lane < s ? in[o + lane] : 0ull;
			uint64_t w1 = (uint32_t)lane + 64 < s ? in[o + lane + 64] : 0ull;
			uint64_t w2 = (uint32_t)lane + 128 < s ? in[o + lane + 128] : 0ull;

			for (int fr = 0; fr < nr; ++fr)
			{
				RecRegs rec;
				rec.lo[0] = (uint32_t)w0, rec.hi[0] = (uint32_t)(w0 >> 32);
				rec.lo[1] = (uint32_t)w1, rec.hi[1] = (uint32_t)(w1 >> 32);
				rec.lo[2] = (uint32_t)w2, rec.hi[2] = (uint32_t)(w2 >> 32);
				const uint32_t words = s;

				if (fr + 1 < nr)
				{ // prefetch the next record while this one is unpacked
					s = (uint32_t)__builtin_amdgcn_readlane((int)my_sz, fr + 1);
					o = (uint32_t)__builtin_amdgcn_readlane((int)my_off, fr + 1);
					if (o + s > seg_len)
					{
						if (lane == 0)
							atomicExch(error_flag, 1);
						return;
					}
					w0 = (uint32_t)lane < s ? in[o + lane] : 0ull;
					w1 = (s > 64 && (uint32_t)lane + 64 < s) ? in[o + lane + 64] : 0ull;
					w2 = (s > 128 && (uint32_t)lane + 128 < s) ? in[o + lane + 128] : 0ull;
				}

				const int f = f0 + fr;
				uint32_t mode = (f == 0) ? RIRB1_MODE_RAW : RIRB1_MODE_TEMPORAL;
				Px8 z;
				z.d[0] = z.d[1] = z.d[2] = z.d[3] = 0;
				if (words)
				{
					const uint64_t hdr = rec_get(rec, 0);
					mode = (uint32_t)(hdr >> 5) & 3u;
					int k = 1;
					const uint32_t sel = lane & 31;
#pragma unroll
					for (int j = 0; j < 8; ++j)
					{
						const uint32_t wj = (uint32_t)(hdr >> (8 * j)) & 31u;
						if (wj > 16 || k + wj > words)
						{
							if (lane == 0)
								atomicExch(error_flag, 1);
							return;
						}
						uint32_t v = 0;
						for (uint32_t b = 0; b < wj; ++b)
						{
							const uint64_t m = rec_get(rec, k + (int)b);
							const uint32_t half = lane < 32 ? (uint32_t)m : (uint32_t)(m >> 32);
							v |= ((half >> sel) & 1u) << b;
						}
						k += (int)wj;
						z.d[j >> 1] |= (j & 1) ? (v << 16) : v;
					}
				}

				Px8 cur;
				if (mode == RIRB1_MODE_TEMPORAL)
				{
#pragma unroll
					for (int k = 0; k < 4; ++k)
						cur.d[k] = pk_add16(prev.d[k], pk_unzigzag16(z.d[k]));
				}
				else if (mode == RIRB1_MODE_LEFT)
				{
					// inclusive prefix sum (mod 2^16) over the tile: lane-local, then across lanes
					uint32_t a[8];
#pragma unroll
					for (int k = 0; k < 4; ++k)
					{
						const uint32_t d = pk_unzigzag16(z.d[k]);
						a[2 * k] = d & 0xffffu;
						a[2 * k + 1] = d >> 16;
					}
#pragma unroll
					for (int i = 1; i < 8; ++i)
						a[i] = (a[i] + a[i - 1]) & 0xffffu;
					const uint32_t incl = wave_scan_add(a[7], lane);
					const uint32_t carry = (incl - a[7]) & 0xffffu;
#pragma unroll
					for (int k = 0; k < 4; ++k)
						cur.d[k] = ((a[2 * k] + carry) & 0xffffu) | (((a[2 * k + 1] + carry) & 0xffffu) << 16);
				}
				else
				{
					cur = z;
				}
				store8(frames, f_begin + f, npx, p0, vec_ok, cur);
				prev = cur;
			}
			round_base = (uint32_t)__builtin_amdgcn_readlane((int)(round_base + my_end), 63);
		}
	}

} // namespace rir

// ---- host launchers (C++ linkage, used by codec_abi.cpp) -----------------------------------------

namespace rir
{
	hipError_t launch_encode(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint8_t *d_sizes,
							 uint32_t *d_seg_words, uint64_t *d_sparse, uint32_t *d_tile_off, uint64_t *d_chunk_words,
							 uint64_t *d_chunk_off, uint64_t *d_stream, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		dim3 grid((ntiles + 3) / 4, nchunks), block(256);
		hipLaunchKernelGGL(rirb1_encode_tiles, grid, block, 0, st, d_frames, npx, ntiles, nframes, gop, d_sizes, d_seg_words, d_sparse);
		hipLaunchKernelGGL(rirb1_scan_tiles, dim3(nchunks), block, 0, st, d_seg_words, ntiles, d_tile_off, d_chunk_words);
		hipLaunchKernelGGL(rirb1_compact, dim3(ntiles, nchunks), block, 0, st, d_sparse, d_tile_off, d_chunk_words, ntiles, nchunks, gop,
						   d_chunk_off, d_stream);
		return hipGetLastError();
	}

	hipError_t launch_decode(const uint8_t *d_sizes, const uint32_t *d_tile_off, const uint64_t *d_chunk_off, const uint64_t *d_stream,
							 int64_t npx, int ntiles, int nframes, int gop, uint16_t *d_frames, int *d_error, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		dim3 grid((ntiles + 3) / 4, nchunks), block(256);
		hipLaunchKernelGGL(rirb1_decode_tiles, grid, block, 0, st, d_sizes, d_tile_off, d_chunk_off, d_stream, npx, ntiles, nframes, gop,
						   d_frames, d_error);
		return hipGetLastError();
	}
} // namespace rir
