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
//   * residual = prediction error minus its tile minimum (DPP min-reduce), so it is an unsigned
//     range [0, max-min];
//   * the bit-planes of the 8 residual slots are produced by two 64x64 BIT-MATRIX TRANSPOSES
//     across the wavefront (v_permlane32_swap + 5 butterfly stages of {cross-lane move,
//     v_alignbit, v_bfi}): afterwards lane 16j+b holds plane b of slot j, a ballot of the
//     non-zero planes gives the 8 widths on the scalar unit, and each lane stores its own word.
//     No LDS, no barriers, no data-dependent branches, fixed instruction count per record;
//   * decode is the mirror image (the transpose is an involution);
//   * 4 independent waves per 256-thread workgroup; grid = (ntiles/4) x nchunks >> 256 CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec_format.h"
#include "filter_kernels.h"

namespace rir
{

	typedef short short2v __attribute__((ext_vector_type(2)));
	typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

	// ---- packed 2 x u16 arithmetic (low half = even pixel) -----------------------------------
	__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)
	{
		return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2v, a) - __builtin_bit_cast(ushort2v, b));
	}
	__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b)
	{
		return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2v, a) + __builtin_bit_cast(ushort2v, b));
	}
	__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b)
	{
		return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
	}
	__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
	{
		return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, b)));
	}

	// ---- wave reductions / scans ------------------------------------------------------------------
	// DPP row shifts inside the four 16-lane rows, then the two row broadcasts; lane 63 has the total.
	__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
	{
		const int32_t id = 0x7fffffff;
		v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false)); // row_shr:1
		v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false)); // row_shr:2
		v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x114, 0xf, 0xf, false)); // row_shr:4
		v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x118, 0xf, 0xf, false)); // row_shr:8
		v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x142, 0xa, 0xf, false)); // row_bcast:15
		v = min(v, __builtin_amdgcn_update_dpp(id, v, 0x143, 0xc, 0xf, false)); // row_bcast:31
		return __builtin_amdgcn_readlane(v, 63);
	}
	__device__ __forceinline__ uint32_t wave_or(uint32_t v)
	{
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
		v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
		return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
	}
	// inclusive add-scan over the 64 lanes (rare paths only)
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

	// ---- 64x64 bit-matrix transpose across the wavefront ----------------------------------------
	// Lane l holds row l as (lo, hi); afterwards lane q holds column q: bit l of the result = bit q
	// of lane l's input.  Recursive block swap: distance 32 is one v_permlane32_swap, distances
	// 16..1 exchange with lane l^s and merge with a per-lane rotate amount and select mask.
	struct TransposeConsts
	{
		uint32_t k16, k8, k4, k2, k1; // select mask: bits taken from the partner
		uint32_t a16, a8, a4, a2, a1; // right-rotate amount applied to the partner's dword
	};
	__device__ __forceinline__ TransposeConsts make_transpose_consts(int lane)
	{
		TransposeConsts c;
		c.k16 = (lane & 16) ? 0x0000ffffu : 0xffff0000u;
		c.k8 = (lane & 8) ? 0x00ff00ffu : 0xff00ff00u;
		c.k4 = (lane & 4) ? 0x0f0f0f0fu : 0xf0f0f0f0u;
		c.k2 = (lane & 2) ? 0x33333333u : 0xccccccccu;
		c.k1 = (lane & 1) ? 0x55555555u : 0xaaaaaaaau;
		c.a16 = 16;
		c.a8 = (lane & 8) ? 8 : 24;
		c.a4 = (lane & 4) ? 4 : 28;
		c.a2 = (lane & 2) ? 2 : 30;
		c.a1 = (lane & 1) ? 1 : 31;
		return c;
	}
	__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }

#define RIR_XOR16(x) ((uint32_t)__builtin_amdgcn_ds_swizzle((int)(x), (16 << 10) | 0x1f))
#define RIR_XOR8(x) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x128, 0xf, 0xf, false)) /* row_ror:8 */
#define RIR_XOR4(x) ((uint32_t)__builtin_amdgcn_ds_swizzle((int)(x), (4 << 10) | 0x1f))
#define RIR_XOR2(x) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x4e, 0xf, 0xf, false)) /* quad_perm:[2,3,0,1] */
#define RIR_XOR1(x) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0xb1, 0xf, 0xf, false)) /* quad_perm:[1,0,3,2] */
#define RIR_TSTAGE(X, K, A)                                   \
	{                                                         \
		const uint32_t tl = X(lo), th = X(hi);                \
		lo = bfi(K, __builtin_amdgcn_alignbit(tl, tl, A), lo); \
		hi = bfi(K, __builtin_amdgcn_alignbit(th, th, A), hi); \
	}

	__device__ __forceinline__ void transpose64(uint32_t &lo, uint32_t &hi, const TransposeConsts &c)
	{
		auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false); // lo[32..63] <-> hi[0..31]
		lo = r[0];
		hi = r[1];
		RIR_TSTAGE(RIR_XOR16, c.k16, c.a16)
		RIR_TSTAGE(RIR_XOR8, c.k8, c.a8)
		RIR_TSTAGE(RIR_XOR4, c.k4, c.a4)
		RIR_TSTAGE(RIR_XOR2, c.k2, c.a2)
		RIR_TSTAGE(RIR_XOR1, c.k1, c.a1)
	}

	// ---- tile I/O ------------------------------------------------------------------------------------
	struct Px8
	{
		uint32_t d[4]; // 8 x u16, d[k] = pixels 2k (low) and 2k+1 (high)
	};

	// 8 consecutive pixels of frame `f` starting at flat index p0 (zero past the end of the frame)
	__device__ __forceinline__ Px8 load8(const uint16_t *__restrict__ frames, int64_t f, int64_t npx, int64_t p0, bool vec_ok)
	{
		Px8 r;
		const uint16_t *src = frames + f * npx + p0;
		if (vec_ok)
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
		if (vec_ok)
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

	// tile minimum of the 8 packed values of every lane -> wave-uniform 16-bit base
	__device__ __forceinline__ uint32_t tile_base(const Px8 &d, bool is_signed)
	{
		int32_t v;
		if (is_signed)
		{
			const uint32_t m = pk_min_i16(pk_min_i16(d.d[0], d.d[1]), pk_min_i16(d.d[2], d.d[3]));
			v = min((int32_t)(int16_t)(m & 0xffffu), (int32_t)m >> 16);
		}
		else
		{
			const uint32_t m = pk_min_u16(pk_min_u16(d.d[0], d.d[1]), pk_min_u16(d.d[2], d.d[3]));
			v = (int32_t)min(m & 0xffffu, m >> 16);
		}
		return (uint32_t)wave_min_i32(v) & 0xffffu;
	}

	// left-delta prediction error inside a tile (MODE_LEFT): d[i] = p[i] - p[i-1], p[-1] = 0
	__device__ __forceinline__ Px8 left_delta(const Px8 &cur, int lane)
	{
		uint32_t prev_last = (uint32_t)__shfl_up((int)(cur.d[3] >> 16), 1, 64);
		if (lane == 0)
			prev_last = 0;
		Px8 d;
		d.d[0] = pk_sub16(cur.d[0], (cur.d[0] << 16) | prev_last);
		d.d[1] = pk_sub16(cur.d[1], (cur.d[1] << 16) | (cur.d[0] >> 16));
		d.d[2] = pk_sub16(cur.d[2], (cur.d[2] << 16) | (cur.d[1] >> 16));
		d.d[3] = pk_sub16(cur.d[3], (cur.d[3] << 16) | (cur.d[2] >> 16));
		return d;
	}

	// sum over the 8 slots of the bit length of the OR of the slot's residuals (payload words)
	__device__ __forceinline__ uint32_t payload_words(const Px8 &r)
	{
		uint32_t tot = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			const uint32_t o = wave_or(r.d[k]);
			tot += bitlen32(o & 0xffffu) + bitlen32(o >> 16);
		}
		return tot;
	}

	// slot placement word (first word | width << 16) of this lane's 16-lane group
	__device__ __forceinline__ uint32_t select4(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, int grp)
	{
		return grp == 0 ? s0 : (grp == 1 ? s1 : (grp == 2 ? s2 : s3));
	}

	// Emit the payload of one record: residuals r (packed pairs) -> plane words at out[0..words).
	// Returns the header (widths | mode | base).
	__device__ __forceinline__ uint64_t emit_record(const Px8 &r, uint32_t mode, uint32_t base, uint64_t *__restrict__ out, int grp, uint32_t bit,
													const TransposeConsts &tc, uint32_t *words)
	{
		uint32_t alo = r.d[0], ahi = r.d[1], blo = r.d[2], bhi = r.d[3];
		transpose64(alo, ahi, tc); // lane 16j+b: plane b of slot j
		transpose64(blo, bhi, tc); // lane 16j+b: plane b of slot 4+j
		const uint64_t nza = __ballot((alo | ahi) != 0);
		const uint64_t nzb = __ballot((blo | bhi) != 0);
		uint32_t s[8];
		uint32_t pos = 0;
		uint64_t hdr = ((uint64_t)mode << 40) | ((uint64_t)base << 48);
#pragma unroll
		for (int j = 0; j < 8; ++j)
		{
			const uint32_t field = (uint32_t)((j < 4 ? nza : nzb) >> (16 * (j & 3))) & 0xffffu;
			const uint32_t w = bitlen32(field);
			s[j] = pos | (w << 16);
			hdr |= (uint64_t)w << (5 * j);
			pos += w;
		}
		const uint32_t sa = select4(s[0], s[1], s[2], s[3], grp);
		const uint32_t sb = select4(s[4], s[5], s[6], s[7], grp);
		if (bit < (sa >> 16))
			out[(sa & 0xffffu) + bit] = (uint64_t)alo | ((uint64_t)ahi << 32);
		if (bit < (sb >> 16))
			out[(sb & 0xffffu) + bit] = (uint64_t)blo | ((uint64_t)bhi << 32);
		*words = pos;
		return hdr;
	}

	// ---- encode -----------------------------------------------------------------------------
	//
	// grid  = (ceil(ntiles/4), nchunks), block = 256 (4 independent waves)
	// hdr       [nchunks][ntiles][gop]      u64  record headers
	// seg_words [nchunks][ntiles]           u32  segment length (sum over the chunk's frames)
	// sparse    [nchunks][ntiles][gop*128]  u64, only the first seg_words words of a slot are written
	__global__ __launch_bounds__(256) void rirb1_encode_tiles(const uint16_t *__restrict__ frames, int64_t npx, int ntiles,
															 int nframes, int gop, uint64_t *__restrict__ hdr_table,
															 uint32_t *__restrict__ seg_words, uint64_t *__restrict__ sparse)
	{
		const int lane = threadIdx.x & 63;
		const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
		if (tile >= ntiles)
			return;
		const int chunk = blockIdx.y;
		const int f_begin = chunk * gop;
		const int nf = min(gop, nframes - f_begin);
		const int64_t p0 = (int64_t)tile * RIRB1_TILE_PX + lane * 8;
		// lanes of a full tile take the 16-byte path; ragged sizes take the element path
		const bool vec_ok = ((npx & 7) == 0) && (p0 + 8 <= npx);
		const int64_t slot = (int64_t)chunk * ntiles + tile;
		uint64_t *my_hdr = hdr_table + slot * gop;
		uint64_t *out = sparse + slot * (int64_t)gop * RIRB1_REC_MAX_WORDS;
		const TransposeConsts tc = make_transpose_consts(lane);
		const int grp = lane >> 4;
		const uint32_t bit = lane & 15;

		uint32_t pos = 0;
		uint64_t hdr_reg = 0; // lane (f & 63) keeps the header of frame f until the 64-frame flush

		// ---- key frame: RAW, or LEFT when its payload is strictly smaller ----
		Px8 cur = load8(frames, f_begin, npx, p0, vec_ok);
		Px8 nxt = cur;
		if (nf > 1)
			nxt = load8(frames, f_begin + 1, npx, p0, vec_ok);
		{
			const uint32_t base_raw = tile_base(cur, false);
			const uint32_t b2 = base_raw | (base_raw << 16);
			Px8 r_raw;
#pragma unroll
			for (int k = 0; k < 4; ++k)
				r_raw.d[k] = pk_sub16(cur.d[k], b2);
			const Px8 dl = left_delta(cur, lane);
			const uint32_t base_left = tile_base(dl, true);
			const uint32_t bl2 = base_left | (base_left << 16);
			Px8 r_left;
#pragma unroll
			for (int k = 0; k < 4; ++k)
				r_left.d[k] = pk_sub16(dl.d[k], bl2);
			const bool use_left = payload_words(r_left) < payload_words(r_raw);
			Px8 r_sel;
#pragma unroll
			for (int k = 0; k < 4; ++k)
				r_sel.d[k] = use_left ? r_left.d[k] : r_raw.d[k];
			uint32_t words;
			const uint64_t h =
				emit_record(r_sel, use_left ? RIRB1_MODE_LEFT : RIRB1_MODE_RAW, use_left ? base_left : base_raw, out, grp, bit, tc, &words);
			if (lane == 0)
				hdr_reg = h;
			pos += words;
		}

		// ---- temporal frames ----
		for (int f = 1; f < nf; ++f)
		{
			const Px8 prev = cur;
			cur = nxt;
			if (f + 1 < nf)
				nxt = load8(frames, f_begin + f + 1, npx, p0, vec_ok); // prefetch, consumed next iteration
			Px8 d;
#pragma unroll
			for (int k = 0; k < 4; ++k)
				d.d[k] = pk_sub16(cur.d[k], prev.d[k]);
			const uint32_t base = tile_base(d, true);
			const uint32_t b2 = base | (base << 16);
#pragma unroll
			for (int k = 0; k < 4; ++k)
				d.d[k] = pk_sub16(d.d[k], b2);
			uint32_t words;
			const uint64_t h = emit_record(d, RIRB1_MODE_TEMPORAL, base, out + pos, grp, bit, tc, &words);
			if ((f & 63) == 0)
			{ // flush the previous 64 headers (one coalesced 8-byte store per lane)
				my_hdr[f - 64 + lane] = hdr_reg;
				hdr_reg = 0;
			}
			if (lane == (f & 63))
				hdr_reg = h;
			pos += words;
		}
		{
			const int fb = (nf - 1) & ~63;
			if (fb + lane < nf)
				my_hdr[fb + lane] = hdr_reg;
		}
		for (int f = nf + lane; f < gop; f += 64)
			my_hdr[f] = 0; // short last chunk: the unused table entries are defined
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
	// One record in flight: header + the two plane words of this lane.
	struct Fetched
	{
		uint64_t hdr;
		uint64_t a, b;
		uint32_t words;
		bool bad;
	};

	__device__ __forceinline__ Fetched fetch_record(uint64_t hdr, const uint64_t *__restrict__ in, uint32_t pos, uint32_t seg_len, int grp, uint32_t bit)
	{
		Fetched r;
		r.hdr = hdr;
		uint32_t s[8];
		uint32_t p = 0, wmax = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j)
		{
			const uint32_t w = (uint32_t)(hdr >> (5 * j)) & 31u;
			s[j] = p | (w << 16);
			p += w;
			wmax = max(wmax, w);
		}
		r.words = p;
		r.bad = (wmax > 16) || (pos + p > seg_len) || (((hdr >> 40) & 3u) == 3u);
		r.a = 0;
		r.b = 0;
		if (!r.bad)
		{ // never read outside the segment
			const uint32_t sa = select4(s[0], s[1], s[2], s[3], grp);
			const uint32_t sb = select4(s[4], s[5], s[6], s[7], grp);
			if (bit < (sa >> 16))
				r.a = in[pos + (sa & 0xffffu) + bit];
			if (bit < (sb >> 16))
				r.b = in[pos + (sb & 0xffffu) + bit];
		}
		return r;
	}

	__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
	{
		return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) |
			   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
	}

	// grid = (ceil(ntiles/4), nchunks), block = 256 (4 independent waves)
	__global__ __launch_bounds__(256) void rirb1_decode_tiles(const uint64_t *__restrict__ hdr_table, const uint32_t *__restrict__ tile_off,
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
		const int64_t p0 = (int64_t)tile * RIRB1_TILE_PX + lane * 8;
		const bool vec_ok = ((npx & 7) == 0) && (p0 + 8 <= npx);
		const int64_t slot = (int64_t)chunk * ntiles + tile;
		const uint64_t *my_hdr = hdr_table + slot * gop;
		const uint32_t t0 = tile_off[(int64_t)chunk * (ntiles + 1) + tile];
		const uint32_t seg_len = tile_off[(int64_t)chunk * (ntiles + 1) + tile + 1] - t0;
		const uint64_t *in = stream + chunk_off[chunk] + t0;
		const TransposeConsts tc = make_transpose_consts(lane);
		const int grp = lane >> 4;
		const uint32_t bit = lane & 15;

		Px8 prev;
		prev.d[0] = prev.d[1] = prev.d[2] = prev.d[3] = 0;
		uint32_t pos = 0;
		for (int f0 = 0; f0 < nf; f0 += 64)
		{
			const int nr = min(64, nf - f0);
			const uint64_t my_h = lane < nr ? my_hdr[f0 + lane] : 0ull; // 64 headers, one coalesced load
			Fetched cur = fetch_record(readlane64(my_h, 0), in, pos, seg_len, grp, bit);
			for (int fr = 0; fr < nr; ++fr)
			{
				if (cur.bad)
				{
					if (lane == 0)
						atomicExch(error_flag, 1);
					return;
				}
				pos += cur.words;
				Fetched nxt = cur;
				if (fr + 1 < nr) // prefetch the next record while this one is transposed
					nxt = fetch_record(readlane64(my_h, fr + 1), in, pos, seg_len, grp, bit);

				uint32_t alo = (uint32_t)cur.a, ahi = (uint32_t)(cur.a >> 32), blo = (uint32_t)cur.b, bhi = (uint32_t)(cur.b >> 32);
				transpose64(alo, ahi, tc);
				transpose64(blo, bhi, tc);
				const uint32_t mode = (uint32_t)(cur.hdr >> 40) & 3u;
				const uint32_t base = (uint32_t)(cur.hdr >> 48) & 0xffffu;
				const uint32_t b2 = base | (base << 16);
				Px8 d;
				d.d[0] = pk_add16(alo, b2);
				d.d[1] = pk_add16(ahi, b2);
				d.d[2] = pk_add16(blo, b2);
				d.d[3] = pk_add16(bhi, b2);

				Px8 out;
				if (mode == RIRB1_MODE_TEMPORAL)
				{
#pragma unroll
					for (int k = 0; k < 4; ++k)
						out.d[k] = pk_add16(prev.d[k], d.d[k]);
				}
				else if (mode == RIRB1_MODE_LEFT)
				{ // inclusive prefix sum (mod 2^16) over the tile: lane-local, then across lanes
					uint32_t a[8];
#pragma unroll
					for (int k = 0; k < 4; ++k)
					{
						a[2 * k] = d.d[k] & 0xffffu;
						a[2 * k + 1] = d.d[k] >> 16;
					}
#pragma unroll
					for (int i = 1; i < 8; ++i)
						a[i] = (a[i] + a[i - 1]) & 0xffffu;
					const uint32_t incl = wave_scan_add(a[7], lane);
					const uint32_t carry = (incl - a[7]) & 0xffffu;
#pragma unroll
					for (int k = 0; k < 4; ++k)
						out.d[k] = ((a[2 * k] + carry) & 0xffffu) | (((a[2 * k + 1] + carry) & 0xffffu) << 16);
				}
				else
				{
					out = d;
				}
				store8(frames, f_begin + f0 + fr, npx, p0, vec_ok, out);
				prev = out;
				cur = nxt;
			}
		}
		if (pos != seg_len && lane == 0)
			atomicExch(error_flag, 1);
	}

	// ---- host launchers --------------------------------------------------------------------------------

	// stage 1: one pass over the raw frames -> headers, per-tile segment lengths, sparse payload
	hipError_t launch_encode_tiles(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint64_t *d_hdr,
								   uint32_t *d_seg_words, uint64_t *d_sparse, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		dim3 grid((ntiles + 3) / 4, nchunks), block(256);
		hipLaunchKernelGGL(rirb1_encode_tiles, grid, block, 0, st, d_frames, npx, ntiles, nframes, gop, d_hdr, d_seg_words, d_sparse);
		return hipGetLastError();
	}

	// stage 2: offsets (exclusive scans) + gather of the sparse slots into the compact stream
	hipError_t launch_encode_compact(int ntiles, int nframes, int gop, const uint32_t *d_seg_words, const uint64_t *d_sparse,
									 uint32_t *d_tile_off, uint64_t *d_chunk_words, uint64_t *d_chunk_off, uint64_t *d_stream, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		dim3 block(256);
		hipLaunchKernelGGL(rirb1_scan_tiles, dim3(nchunks), block, 0, st, d_seg_words, ntiles, d_tile_off, d_chunk_words);
		hipLaunchKernelGGL(rirb1_compact, dim3(ntiles, nchunks), block, 0, st, d_sparse, d_tile_off, d_chunk_words, ntiles, nchunks, gop,
						   d_chunk_off, d_stream);
		return hipGetLastError();
	}

	hipError_t launch_decode(const uint64_t *d_hdr, const uint32_t *d_tile_off, const uint64_t *d_chunk_off, const uint64_t *d_stream,
							 int64_t npx, int ntiles, int nframes, int gop, uint16_t *d_frames, int *d_error, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		dim3 grid((ntiles + 3) / 4, nchunks), block(256);
		hipLaunchKernelGGL(rirb1_decode_tiles, grid, block, 0, st, d_hdr, d_tile_off, d_chunk_off, d_stream, npx, ntiles, nframes, gop, d_frames,
						   d_error);
		return hipGetLastError();
	}
} // namespace rir
