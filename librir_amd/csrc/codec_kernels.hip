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
#include <stdlib.h>

#include "codec_format.h"
#include "filter_kernels.h"
#include "runtime.h"

// Cache policy of each traffic class (aux operand of the buffer instructions: 2 = nt, "streaming").
// Raw frames are read once by the encoder and written once by the decoder, the sparse slots are read once by
// the compaction: marking those accesses non-temporal keeps them from evicting what the NEXT kernel of the
// pipeline reads (the dense stream, which should stay in L2 / Infinity Cache between compaction and decode).
// Measured on the headline workload (scripts/variants.py): all-default 438 us per encode+decode pass,
// frames nt 391 us, + compaction loads nt 351 us; nt on the stream loads or the sparse stores is worse.
#ifndef RIR_FRAME_LOAD_AUX
#define RIR_FRAME_LOAD_AUX 2
#endif
#ifndef RIR_FRAME_STORE_AUX
#define RIR_FRAME_STORE_AUX 2
#endif
#ifndef RIR_SPARSE_STORE_AUX
#define RIR_SPARSE_STORE_AUX 0
#endif
#ifndef RIR_STREAM_LOAD_AUX
#define RIR_STREAM_LOAD_AUX 0
#endif
#ifndef RIR_COMPACT_NT_LOAD
#define RIR_COMPACT_NT_LOAD 1
#endif
#ifndef RIR_COMPACT_NT_STORE
#define RIR_COMPACT_NT_STORE 0
#endif

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
	// (mask & a) | (~mask & b) in one VALU op: v_bitop3_b32 with the truth table of a 3-input mux
	__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xCA); }

#define RIR_XOR16(x) ((uint32_t)__builtin_amdgcn_ds_swizzle((int)(x), (16 << 10) | 0x1f))
#define RIR_XOR8(x) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), 0x128, 0xf, 0xf, false)) /* row_ror:8 */
#define RIR_XOR4(x) ((uint32_t)__builtin_amdgcn_ds_swizzle((int)(x), (4 << 10) | 0x1f))
#define RIR_XOR2(x) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), 0x4e, 0xf, 0xf, false)) /* quad_perm:[2,3,0,1] */
#define RIR_XOR1(x) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), 0xb1, 0xf, 0xf, false)) /* quad_perm:[1,0,3,2] */
#define RIR_TSTAGE2(X, K, A)                                    \
	{                                                           \
		const uint32_t t0 = X(a0), t1 = X(a1), t2 = X(b0), t3 = X(b1); \
		a0 = bfi(K, __builtin_amdgcn_alignbit(t0, t0, A), a0);   \
		a1 = bfi(K, __builtin_amdgcn_alignbit(t1, t1, A), a1);   \
		b0 = bfi(K, __builtin_amdgcn_alignbit(t2, t2, A), b0);   \
		b1 = bfi(K, __builtin_amdgcn_alignbit(t3, t3, A), b1);   \
	}

	// two independent transposes, stage by stage (4 independent dword chains fill the DPP/LDS latency)
	__device__ __forceinline__ void transpose64x2(uint32_t &a0, uint32_t &a1, uint32_t &b0, uint32_t &b1, const TransposeConsts &c)
	{
		auto ra = __builtin_amdgcn_permlane32_swap(a0, a1, false, false); // lo[32..63] <-> hi[0..31]
		auto rb = __builtin_amdgcn_permlane32_swap(b0, b1, false, false);
		a0 = ra[0];
		a1 = ra[1];
		b0 = rb[0];
		b1 = rb[1];
		RIR_TSTAGE2(RIR_XOR16, c.k16, c.a16)
		RIR_TSTAGE2(RIR_XOR8, c.k8, c.a8)
		RIR_TSTAGE2(RIR_XOR4, c.k4, c.a4)
		RIR_TSTAGE2(RIR_XOR2, c.k2, c.a2)
		RIR_TSTAGE2(RIR_XOR1, c.k1, c.a1)
	}

	// 32x32 bit-matrix transpose inside each 32-lane half of the wave (one dword per lane): lane q of a
	// half ends with bit l = bit q of that half's lane l.  Same butterfly, without the 32-lane stage.
	__device__ __forceinline__ uint32_t transpose32(uint32_t x, const TransposeConsts &c)
	{
		uint32_t t;
		t = RIR_XOR16(x);
		x = bfi(c.k16, __builtin_amdgcn_alignbit(t, t, c.a16), x);
		t = RIR_XOR8(x);
		x = bfi(c.k8, __builtin_amdgcn_alignbit(t, t, c.a8), x);
		t = RIR_XOR4(x);
		x = bfi(c.k4, __builtin_amdgcn_alignbit(t, t, c.a4), x);
		t = RIR_XOR2(x);
		x = bfi(c.k2, __builtin_amdgcn_alignbit(t, t, c.a2), x);
		t = RIR_XOR1(x);
		x = bfi(c.k1, __builtin_amdgcn_alignbit(t, t, c.a1), x);
		return x;
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

	// ---- buffer-resource addressing -------------------------------------------------------------
	// Every vector-memory operation of the steady-state loops is an UNCONDITIONAL raw-buffer access:
	// lanes that must not touch memory get an out-of-range offset (the hardware drops the store /
	// returns 0 for the load), so no branch surrounds a memory instruction and the compiler can keep
	// exact s_waitcnt vmcnt(N) counts - which is what lets three frames stay in flight per wave.
	typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
	typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
#define RIR_BUF_FLAGS 0x00020000 /* raw buffer, 32-bit data format (gfx942/gfx950 descriptor word 3) */
#define RIR_OOB 0x40000000u		 /* beyond any num_records used here */

	__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes)
	{
		// the base is wave-uniform by construction; readfirstlane makes that provable
		const uint64_t b = (uint64_t)base;
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
		const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
		return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)bytes, RIR_BUF_FLAGS);
	}
	__device__ __forceinline__ Px8 buf_load8(const void *tile_base, uint32_t lane_off)
	{
		const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(tile_base, RIRB1_TILE_PX * 2), lane_off, 0, RIR_FRAME_LOAD_AUX);
		Px8 r;
		r.d[0] = v.x;
		r.d[1] = v.y;
		r.d[2] = v.z;
		r.d[3] = v.w;
		return r;
	}
	__device__ __forceinline__ void buf_store8(void *tile_base, uint32_t lane_off, const Px8 &r)
	{
		v4u32 v;
		v.x = r.d[0];
		v.y = r.d[1];
		v.z = r.d[2];
		v.w = r.d[3];
		__builtin_amdgcn_raw_buffer_store_b128(v, make_rsrc(tile_base, RIRB1_TILE_PX * 2), lane_off, 0, RIR_FRAME_STORE_AUX);
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

	// ---- per-lane bookkeeping of a record (no scalar-unit work) ---------------------------------
	// After the transposes lane 16q+k holds plane k of slot q (half A) and of slot 4+q (half B).
	struct LaneConsts
	{
		uint32_t bit;	 // k = lane & 15
		uint32_t bitp1;	 // k + 1
		uint32_t onehot; // 1 << k
		uint32_t sh4;	 // 4 * q   (base nibble of this row)
		uint32_t sh16;	 // 16 * (q & 1)
		bool row_hi;	 // q >= 2 : this row's header field is in the high dword
		bool row0;		 // q == 0
		// narrow tier (all widths <= 4): lane (lane & 31) = 4*slot + plane, the two 32-lane halves
		// hold the low / high dword of the plane word
		uint32_t n_bit;	   // plane = lane & 3
		uint32_t n_sh;	   // 4 * slot : position of the slot's planes in the 32-bit non-zero mask
		uint32_t n_half;   // byte offset of this half inside the 64-bit word (0 or 4)
		uint32_t n_hsh;	   // bit position of the slot's width inside its header dword
		bool n_hhi;		   // the slot's width lives in the high header dword
		bool n_first;	   // plane 0 of its slot
		// narrow tier, packed-nibble order (see emit_record_narrow): nibble n of the packed dword holds slot
		// NIB_SLOT[n] = {0,4,2,6,1,5,3,7}; lane (lane & 31) = 4n + b holds plane b of that slot
		uint32_t p_before; // mask of the lanes whose plane words precede this lane's word in the stream
		uint32_t p_hshw;   // header lane 16q+k: left shift that brings its bit of the width word W to bit 31
		uint32_t p_hshb;   // same for the base/mode word
	};
	__device__ __forceinline__ LaneConsts make_lane_consts(int lane)
	{
		LaneConsts c;
		const uint32_t q = (uint32_t)lane >> 4;
		c.bit = lane & 15;
		c.bitp1 = c.bit + 1;
		c.onehot = 1u << c.bit;
		c.sh4 = 4 * q;
		c.sh16 = 16 * (q & 1);
		c.row_hi = q >= 2;
		c.row0 = q == 0;
		const uint32_t slot = ((uint32_t)lane & 31u) >> 2;
		c.n_bit = lane & 3;
		c.n_sh = 4 * slot;
		c.n_half = ((uint32_t)lane >> 5) * 4;
		c.n_hsh = 16 * (slot & 1) + (slot >= 4 ? 5 : 0);
		c.n_hhi = (slot & 2) != 0;
		c.n_first = (lane & 3) == 0;
		{
			const uint32_t nib_slot[8] = {0, 4, 2, 6, 1, 5, 3, 7};
			const uint32_t n = ((uint32_t)lane & 31u) >> 2, b = lane & 3;
			uint32_t before = ((1u << b) - 1u) << (4 * n);
			for (uint32_t m = 0; m < 8; ++m)
				if (nib_slot[m] < nib_slot[n])
					before |= 0xfu << (4 * m);
			c.p_before = before;
			// header field q = w_q | w_{4+q} << 5 | base nibble q << 10 | mode << 14 (q == 0); lane 16q+k tests bit k.
			// Slot q sits in nibble nq = {0,4,2,6}[q] of W, slot 4+q in nibble nq+1; widths are <= 4 (3 bits).
			const uint32_t k = lane & 15, nq = (q & 1) * 4 + (q >> 1) * 2;
			uint32_t bw = 3; // a bit of W that is always 0
			if (k < 3)
				bw = 4 * nq + k;
			else if (k >= 5 && k < 8)
				bw = 4 * (nq + 1) + (k - 5);
			c.p_hshw = 31 - bw;
			uint32_t bb = 31; // a bit of (base | mode << 16) that is always 0
			if (k >= 10 && k < 14)
				bb = 4 * q + (k - 10);
			else if (k >= 14 && q == 0)
				bb = 16 + (k - 14);
			c.p_hshb = 31 - bb;
		}
		return c;
	}

	// max over the 16 lanes of a row, result in every lane (4 DPP rotations)
	__device__ __forceinline__ uint32_t row_allmax(uint32_t v)
	{
		v = max(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x121, 0xf, 0xf, false)); // row_ror:1
		v = max(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x122, 0xf, 0xf, false)); // row_ror:2
		v = max(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xf, 0xf, false)); // row_ror:4
		v = max(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, false)); // row_ror:8
		return v;
	}
	// w: row-uniform value.  Returns the inclusive sum over rows 0..q (row-uniform); row 3 holds the total.
	__device__ __forceinline__ uint32_t rows_inclusive_sum(uint32_t w)
	{
		const uint32_t s = w + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
		return s + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false);			  // row_bcast:31 -> rows 2,3
	}

	// inclusive sum over the 32 lanes of each half (DPP row shifts + one row broadcast)
	__device__ __forceinline__ uint32_t half_inclusive_sum(uint32_t v)
	{
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true); // row_shr:2
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true); // row_shr:4
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true); // row_shr:8
		v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
		return v;
	}
	// bit length of a value < 2^31 without a zero test: bitlen(2f+1) = bitlen(f)+1
	__device__ __forceinline__ uint32_t bitlen_nz(uint32_t f) { return 31u - (uint32_t)__builtin_clz((f << 1) | 1u); }

	// The two 8-byte stores of a record (offsets are out of range for lanes without a plane).  They are issued
	// by the caller, outside the wave-uniform tier branch: with every vector-memory operation of the frame loop
	// unconditional the compiler keeps exact s_waitcnt vmcnt(N) counts (a store inside the branch made it fall
	// back to vmcnt(0) on half of the steps).
	struct RecordStores
	{
		v2u32 va, vb;
		uint32_t oa, ob;
	};

	// Narrow tier: every residual < 16.  The 8 nibbles of a lane are packed in one dword (3 shift-ors, nibble
	// order r0 r4 r2 r6 r1 r5 r3 r7) and ONE 32x32 transpose per half-wave produces the 32 candidate planes
	// (lane 4n+b: plane b of the slot in nibble n; the upper half holds bits 32..63 of the same plane word).
	// Everything after the ballot is wave-uniform and runs on the scalar unit: M = the mask of stored planes
	// (non-zero mask smeared down inside each nibble), W = per-nibble popcount of M = the 8 widths.  A lane's
	// word index is a popcount of M under a per-lane constant mask; the header is two ballots of per-lane
	// bit tests of W and of (base | mode << 16).
	__device__ __forceinline__ uint64_t emit_record_narrow(const Px8 &r, uint32_t mode, uint32_t base, RecordStores &rs, uint32_t pos,
														   const LaneConsts &lc, const TransposeConsts &tc, uint32_t *words)
	{
		uint32_t x = (r.d[0] | (r.d[1] << 8)) | ((r.d[2] | (r.d[3] << 8)) << 4);
		x = transpose32(x, tc);
		const uint64_t nz64 = __ballot(x != 0);
		const uint32_t nz = (uint32_t)nz64 | (uint32_t)(nz64 >> 32); // bit 4n+b: plane b of nibble n is not empty
		uint32_t M = nz | ((nz >> 1) & 0x77777777u);
		M |= (M >> 2) & 0x33333333u; // nibble n = (1 << w) - 1
		uint32_t W = M - ((M >> 1) & 0x55555555u);
		W = (W & 0x33333333u) + ((W >> 2) & 0x33333333u); // nibble n = w (0..4)
		*words = (uint32_t)__builtin_popcount(M);
		// lower-half lanes assemble the 64-bit plane word (their dword | the partner lane's dword << 32)
		auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false); // sw[1] lower lanes = x of lane + 32
		rs.va.x = x, rs.va.y = sw[1];
		const uint32_t off = (pos + (uint32_t)__builtin_popcount(M & lc.p_before)) * 8u;
		rs.oa = __builtin_amdgcn_inverse_ballot_w64((uint64_t)M) ? off : RIR_OOB; // lane l < 32 stores iff bit l of M
		rs.vb = rs.va, rs.ob = RIR_OOB;
		const uint32_t B = base | (mode << 16);
		const uint64_t hw = __ballot((int32_t)(W << lc.p_hshw) < 0) & 0x00E700E700E700E7ull;
		const uint64_t hb = __ballot((int32_t)(B << lc.p_hshb) < 0) & 0x3C003C003C00FC00ull;
		return hw | hb;
	}

	// Wide tier (any width up to 16): two 64x64 transposes, lane 16q+k holds plane k of slots q and 4+q.
	__device__ __forceinline__ uint64_t emit_record_wide(const Px8 &r, uint32_t mode, uint32_t base, RecordStores &rs, uint32_t pos,
														 const LaneConsts &lc, const TransposeConsts &tc, uint32_t *words)
	{
		uint32_t alo = r.d[0], ahi = r.d[1], blo = r.d[2], bhi = r.d[3];
		transpose64x2(alo, ahi, blo, bhi, tc);
		const uint32_t wa = row_allmax((alo | ahi) != 0 ? lc.bitp1 : 0u); // width of slot q
		const uint32_t wb = row_allmax((blo | bhi) != 0 ? lc.bitp1 : 0u); // width of slot 4+q
		const uint32_t ia = rows_inclusive_sum(wa), ib = rows_inclusive_sum(wb);
		const uint32_t tot_a = (uint32_t)__builtin_amdgcn_readlane((int)ia, 63);
		const uint32_t tot_b = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
		// one unconditional 8-byte store per half; lanes without a plane point out of range
		v2u32 va, vb;
		va.x = alo, va.y = ahi, vb.x = blo, vb.y = bhi;
		const uint32_t oa = lc.bit < wa ? (pos + ia - wa + lc.bit) * 8u : RIR_OOB;
		const uint32_t ob = lc.bit < wb ? (pos + tot_a + ib - wb + lc.bit) * 8u : RIR_OOB;
		rs.va = va, rs.oa = oa;
		rs.vb = vb, rs.ob = ob;
		*words = tot_a + tot_b;
		// header = ballot of the row's 16-bit field (w_q | w_{4+q} << 5 | base nibble q << 10 | mode << 14)
		const uint32_t nib = (base >> lc.sh4) & 15u;
		const uint32_t field = ((nib << 10) | (lc.row0 ? mode << 14 : 0u)) | ((wb << 5) | wa);
		return __ballot((field & lc.onehot) != 0);
	}

	// Emit the payload of one record: residuals r (packed pairs) -> plane words at out[pos..pos+words).
	// Returns the header.  The tier is a property of
	// the data (wave-uniform branch), the bitstream is the same either way.
	__device__ __forceinline__ uint64_t emit_record(const Px8 &r, uint32_t mode, uint32_t base, __amdgpu_buffer_rsrc_t out, uint32_t pos,
													const LaneConsts &lc, const TransposeConsts &tc, uint32_t *words)
	{
		const uint32_t any = (r.d[0] | r.d[1]) | (r.d[2] | r.d[3]);
		RecordStores rs;
		uint64_t h;
		if (__ballot((any & 0xfff0fff0u) != 0) == 0)
			h = emit_record_narrow(r, mode, base, rs, pos, lc, tc, words);
		else
			h = emit_record_wide(r, mode, base, rs, pos, lc, tc, words);
		__builtin_amdgcn_raw_buffer_store_b64(rs.va, out, rs.oa, 0, RIR_SPARSE_STORE_AUX);
		__builtin_amdgcn_raw_buffer_store_b64(rs.vb, out, rs.ob, 0, RIR_SPARSE_STORE_AUX);
		return h;
	}

	// One wave = one tile over the frames of one chunk (see the kernel below).
	template <bool FAST>
	__device__ __forceinline__ void encode_tile(const uint16_t *__restrict__ frames, int64_t npx, int nf, int64_t frame0, int tile, int lane,
												uint64_t *__restrict__ my_hdr, uint64_t *__restrict__ out_ptr, uint32_t out_bytes,
												uint32_t *__restrict__ seg_words_slot, int gop)
	{
		const int64_t p0 = (int64_t)tile * RIRB1_TILE_PX + lane * 8;
		const TransposeConsts tc = make_transpose_consts(lane);
		const LaneConsts lc = make_lane_consts(lane);
		const __amdgpu_buffer_rsrc_t out = make_rsrc(out_ptr, out_bytes);
		const uint32_t lane_off = (uint32_t)lane * 16u;
		const uint16_t *tile0 = frames + frame0 * npx + (int64_t)tile * RIRB1_TILE_PX; // tile of the chunk's first frame
		// FAST: whole tile inside the frame and 16-byte aligned rows -> raw-buffer loads, unconditional;
		// past the end of the chunk the prefetch re-reads the last frame (an L2 hit, no HBM traffic).
		// Frame loads are compiler-visible builtins: the waits are the compiler's own counted s_waitcnt, which
		// stay exact because no vector-memory operation of the frame loop sits inside a branch.
		// Frames are loaded strictly in order (0,1,2,3 up front, then frame f+3 at step f), so the address is a
		// running wave-uniform pointer (2 scalar adds per load instead of a 64-bit multiply chain); past the end
		// of the chunk it stays on the last frame.
		const uint16_t *next_ptr = tile0;
		int next_f = 0;
		auto load = [&](int f, v4u32 &dst) {
			if (FAST)
			{
				const Px8 p = buf_load8(next_ptr, lane_off);
				dst.x = p.d[0], dst.y = p.d[1], dst.z = p.d[2], dst.w = p.d[3];
				const bool more = next_f < nf - 1;
				next_ptr += more ? npx : 0;
				next_f += more ? 1 : 0;
			}
			else
			{
				const Px8 p = load8(frames, frame0 + min(f, nf - 1), npx, p0, false);
				dst.x = p.d[0], dst.y = p.d[1], dst.z = p.d[2], dst.w = p.d[3];
			}
		};
		auto as_px8 = [](const v4u32 &v) {
			Px8 p;
			p.d[0] = v.x, p.d[1] = v.y, p.d[2] = v.z, p.d[3] = v.w;
			return p;
		};

		uint32_t pos = 0;
		uint64_t hdr_reg = 0; // lane (f & 63) keeps the header of frame f until the 64-frame flush

		// Ring of 4 frame slots: frame f lives in slot f % 4, three frames are in flight ahead of the
		// one being packed.  The loop is unrolled by 4 so that slots are fixed registers (a register
		// copy of a loaded value would force an early wait on the load).
		v4u32 s0, s1, s2, s3;
		load(0, s0);
		load(1, s1);
		load(2, s2);
		load(3, s3);

		// ---- key frame: RAW, or LEFT when its payload is strictly smaller ----
		{
			const Px8 cur = as_px8(s0);
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
			const uint32_t key_mode = use_left ? RIRB1_MODE_LEFT : RIRB1_MODE_RAW;
			const uint64_t h = emit_record(r_sel, key_mode, use_left ? base_left : base_raw, out, pos, lc, tc, &words);
			if (lane == 0)
				my_hdr[0] = h;
			pos += words;
		}

		// ---- temporal frames ----
		// Steps come in groups of 64 (one header per lane of hdr_reg, flushed with one coalesced store per
		// group) and the steady-state loop runs whole iterations of 4 steps with NO condition around any
		// vector-memory operation: that is what lets the compiler keep counted waits, i.e. keeps the loads
		// of the next frames in flight across the packing of the current one.
#define RIR_ENC_STEP(F, CUR, PREV)                                                             \
	{                                                                                          \
		const int f = (F);                                                                     \
		Px8 d;                                                                                 \
		{                                                                                      \
			const Px8 c_ = as_px8(CUR), p_ = as_px8(PREV);                                     \
			_Pragma("unroll") for (int k = 0; k < 4; ++k) d.d[k] = pk_sub16(c_.d[k], p_.d[k]); \
		}                                                                                      \
		load(f + 3, PREV); /* the slot of frame f-1 is free: prefetch frame f+3 into it */     \
		const uint32_t base = tile_base(d, true);                                              \
		const uint32_t b2 = base | (base << 16);                                               \
		_Pragma("unroll") for (int k = 0; k < 4; ++k) d.d[k] = pk_sub16(d.d[k], b2);          \
		uint32_t words;                                                                        \
		const uint64_t h = emit_record(d, RIRB1_MODE_TEMPORAL, base, out, pos, lc, tc, &words);      \
		if (lane == ((f - 1) & 63))                                                            \
			hdr_reg = h;                                                                       \
		pos += words;                                                                          \
	}
		int fb = 1;
		while (fb < nf)
		{
			const int g0 = fb, gend = min(fb + 64, nf); // this group: frames [g0, gend), header of frame f in lane f - g0
			for (; fb + 3 < gend; fb += 4)
			{
				RIR_ENC_STEP(fb, s1, s0)
				RIR_ENC_STEP(fb + 1, s2, s1)
				RIR_ENC_STEP(fb + 2, s3, s2)
				RIR_ENC_STEP(fb + 3, s0, s3)
			}
			if (fb < gend)
			{ // up to three left-over steps of the chunk's last group (64 % 4 == 0: the slot rotation stays aligned)
				RIR_ENC_STEP(fb, s1, s0)
				if (fb + 1 < gend)
				{
					RIR_ENC_STEP(fb + 1, s2, s1)
					if (fb + 2 < gend)
						RIR_ENC_STEP(fb + 2, s3, s2)
				}
				fb = gend;
			}
			if (g0 + lane < gend)
				my_hdr[g0 + lane] = hdr_reg;
			hdr_reg = 0;
		}
#undef RIR_ENC_STEP
		for (int f = nf + lane; f < gop; f += 64)
			my_hdr[f] = 0; // short last chunk: the unused table entries are defined
		if (lane == 0)
			*seg_words_slot = pos;
	}

	// ---- encode -----------------------------------------------------------------------------
	//
	// grid  = (ceil(tile_count/4), nchunks), block = 256 (4 independent waves); tiles [tile_first, tile_first + tile_count)
	// hdr       [nchunks][ntiles][gop]      u64  record headers
	// seg_words [nchunks][ntiles]           u32  segment length (sum over the chunk's frames)
	// sparse    [nchunks][ntiles][gop*128]  u64, only the first seg_words words of a slot are written
	// Two kernels: FAST for tiles that lie whole inside 16-byte aligned frames (every tile but the last one of a frame whose
	// size is not a multiple of 512 pixels) - unconditional raw-buffer loads, 64 registers, no scratch - and the ragged form
	// (element-wise loads with bounds tests) for the rest; as one kernel the ragged instantiation's spills cost the fast path
	// its scratch set-up.
	// 8 waves per SIMD (<= 64 VGPRs; the allocator's natural choice was 71 -> 7 waves): 12 800 waves then take 1.56
	// rounds instead of 1.79 and the run-to-run spread of the kernel (141 / 154 us) collapses onto the fast mode
	template <bool FAST>
	__attribute__((amdgpu_waves_per_eu(8, 8))) __global__ __launch_bounds__(256) void rirb1_encode_tiles(const uint16_t *__restrict__ frames, int64_t npx, int ntiles,
																 int tile_first, int tile_count, int nframes, int gop,
																 uint64_t *__restrict__ hdr_table, uint32_t *__restrict__ seg_words,
																 uint64_t *__restrict__ sparse)
	{
		const int lane = threadIdx.x & 63;
		const int ti = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
		if (ti >= tile_count)
			return;
		const int tile = tile_first + ti;
		const int chunk = blockIdx.y;
		const int f_begin = chunk * gop;
		const int nf = min(gop, nframes - f_begin);
		const int64_t slot = (int64_t)chunk * ntiles + tile;
		uint64_t *my_hdr = hdr_table + slot * gop;
		uint64_t *out = sparse + slot * RIRB1_SLOT_WORDS(gop);
		const uint32_t out_bytes = (uint32_t)gop * RIRB1_REC_MAX_WORDS * 8u;
		encode_tile<FAST>(frames, npx, nf, f_begin, tile, lane, my_hdr, out, out_bytes, seg_words + slot, gop);
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
		const int t = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
		// every wave derives the chunk's base itself (sum of the preceding chunks' lengths): no LDS, no barrier,
		// and the loads of the table entries are in flight while the reduction runs
		const uint32_t *off = tile_off + (int64_t)c * (ntiles + 1);
		const uint32_t o0 = off[t], o1 = off[t + 1];
		uint64_t s = 0;
		for (int i = tid & 63; i < c; i += 64)
			s += chunk_words[i];
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
			s += (uint64_t)__shfl_xor((long long)s, d, 64);
		if (t == 0 && tid == 0)
		{
			chunk_off[c] = s;
			if (c == nchunks - 1)
				chunk_off[nchunks] = s + chunk_words[c];
		}
		const uint32_t n = o1 - o0;
		const uint64_t *src = sparse + ((int64_t)c * ntiles + t) * RIRB1_SLOT_WORDS(gop);
		uint64_t *dst = stream + s + o0;
		uint32_t i = tid;
		for (; i + 768 < n; i += 1024)
		{ // four independent 8-byte copies per thread and iteration
			const uint64_t v0 = RIR_COMPACT_NT_LOAD ? __builtin_nontemporal_load(src + i) : src[i];
			const uint64_t v1 = RIR_COMPACT_NT_LOAD ? __builtin_nontemporal_load(src + i + 256) : src[i + 256];
			const uint64_t v2 = RIR_COMPACT_NT_LOAD ? __builtin_nontemporal_load(src + i + 512) : src[i + 512];
			const uint64_t v3 = RIR_COMPACT_NT_LOAD ? __builtin_nontemporal_load(src + i + 768) : src[i + 768];
			dst[i] = v0, dst[i + 256] = v1, dst[i + 512] = v2, dst[i + 768] = v3;
		}
		for (; i < n; i += 256)
			dst[i] = RIR_COMPACT_NT_LOAD ? __builtin_nontemporal_load(src + i) : src[i];
	}

	// ==== single-pass dense encoder ===================================================================
	//
	// One WORKGROUP owns one (chunk, tile) segment; its WAVES waves split the chunk's frames in time (wave 0 takes the
	// key frame, a later wave loads the frame before its first one as its starting `prev`).  The payload words of the
	// segment are staged in LDS (R words per wave; what does not fit goes to a spill slot in the workspace - noisy data
	// degrades to the traffic of a two-pass layout, the reference's recipe never spills).  When a workgroup has packed
	// its frames it knows the length of its segment, publishes it, and looks back for the lengths of the segments in
	// front of it (decoupled look-back): the sum is its place in the dense stream, to which it copies its words from LDS.
	// The stream, tile_off and chunk_off are bit-identical to the two-pass layout (scan + gather) this replaces.
	//
	// Look-back state (control block in the workspace, zeroed by a memset node before every launch):
	//   gran   [chunk][tile]    u64  TAG | segment length                        published by the segment's workgroup
	//   gtotal [chunk][tile/64] u64  TAG | sum of the lengths of a group of 64   published by the group's last arriver
	//   P      [chunk]          u64  TAG | first word of the chunk in the stream published by the previous chunk's last arriver
	//   gcount [chunk][tile/64], ccount [chunk]  u64  arrivals << 40 | sum: returning agent-scope atomic adds that find the
	//          last arriver of a group / of a chunk; on lines of their own, never polled
	// Every POLLED word is written once, by one sc1 store, and carries its own data (granule = flag).  What is polled is
	// never what is added to (counters that 64 workgroups add to while hundreds poll them serialise on their cache line:
	// 60 ns per add, 770 us per launch), and no published word depends on another one except P[c + 1] on P[c] (prefixes
	// chained through the groups cost 200 hops of 2.5 us).
	// Workgroup (c, t) needs  A = sum of lengths of tiles < t in chunk c  (= its tile_off) and P[c]: it polls the granules of
	// the tiles in front of it in its own group of 64, the totals of the groups in front and P[c] - 63 + ntiles / 64 + 1
	// words at most, by ONE wave, with relaxed agent-scope loads (sc1: served by L2 / memory, never a stale L1 line).  No
	// payload is handed between workgroups inside the launch (the stream is read by the NEXT kernel).
	// Forward progress: segments are dealt by tickets (8 heads, head = blockIdx % 8, segment = ticket * 8 + head), so a
	// workgroup only ever waits for segments whose workgroups have taken their ticket earlier on the same head or are
	// not behind it in dispatch; every spin is bounded by a 2 s clock and a shared error word, a wait that gives up
	// raises the error instead of hanging.
#define RIR_LB_TAG (1ull << 63)
#define RIR_LB_TIMEOUT_TICKS 200000000ull /* s_memrealtime runs at 100 MHz: 2 s */
#define RIR_NONE 0xffffffffu

	struct RecordWords
	{
		v2u32 va, vb;
		uint32_t ia, ib; // word index inside the wave's payload, RIR_NONE for a lane without a plane
	};

	__device__ __forceinline__ uint64_t emit_words_narrow(const Px8 &r, uint32_t mode, uint32_t base, RecordWords &rw, uint32_t pos, const LaneConsts &lc,
														  const TransposeConsts &tc, uint32_t *words)
	{
		uint32_t x = (r.d[0] | (r.d[1] << 8)) | ((r.d[2] | (r.d[3] << 8)) << 4);
		x = transpose32(x, tc);
		const uint64_t nz64 = __ballot(x != 0);
		const uint32_t nz = (uint32_t)nz64 | (uint32_t)(nz64 >> 32);
		uint32_t M = nz | ((nz >> 1) & 0x77777777u);
		M |= (M >> 2) & 0x33333333u;
		uint32_t W = M - ((M >> 1) & 0x55555555u);
		W = (W & 0x33333333u) + ((W >> 2) & 0x33333333u);
		*words = (uint32_t)__builtin_popcount(M);
		auto sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
		rw.va.x = x, rw.va.y = sw[1];
		rw.ia = __builtin_amdgcn_inverse_ballot_w64((uint64_t)M) ? pos + (uint32_t)__builtin_popcount(M & lc.p_before) : RIR_NONE;
		rw.vb = rw.va, rw.ib = RIR_NONE;
		const uint32_t B = base | (mode << 16);
		const uint64_t hw = __ballot((int32_t)(W << lc.p_hshw) < 0) & 0x00E700E700E700E7ull;
		const uint64_t hb = __ballot((int32_t)(B << lc.p_hshb) < 0) & 0x3C003C003C00FC00ull;
		return hw | hb;
	}

	__device__ __forceinline__ uint64_t emit_words_wide(const Px8 &r, uint32_t mode, uint32_t base, RecordWords &rw, uint32_t pos, const LaneConsts &lc,
														const TransposeConsts &tc, uint32_t *words)
	{
		uint32_t alo = r.d[0], ahi = r.d[1], blo = r.d[2], bhi = r.d[3];
		transpose64x2(alo, ahi, blo, bhi, tc);
		const uint32_t wa = row_allmax((alo | ahi) != 0 ? lc.bitp1 : 0u);
		const uint32_t wb = row_allmax((blo | bhi) != 0 ? lc.bitp1 : 0u);
		const uint32_t ia = rows_inclusive_sum(wa), ib = rows_inclusive_sum(wb);
		const uint32_t tot_a = (uint32_t)__builtin_amdgcn_readlane((int)ia, 63);
		const uint32_t tot_b = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
		rw.va.x = alo, rw.va.y = ahi, rw.vb.x = blo, rw.vb.y = bhi;
		rw.ia = lc.bit < wa ? pos + ia - wa + lc.bit : RIR_NONE;
		rw.ib = lc.bit < wb ? pos + tot_a + ib - wb + lc.bit : RIR_NONE;
		*words = tot_a + tot_b;
		const uint32_t nib = (base >> lc.sh4) & 15u;
		const uint32_t field = ((nib << 10) | (lc.row0 ? mode << 14 : 0u)) | ((wb << 5) | wa);
		return __ballot((field & lc.onehot) != 0);
	}

	// Where a wave's payload goes: the first records into its LDS region, from the first record that does not fit
	// (wave-uniform decision) everything into its spill area in the workspace.
	// Two kinds of spill area.  STATIC (arena == nullptr): `spill` is the wave's own worst-case slot, there from the start
	// (rirb1_encode_dense).  DYNAMIC (rirb1_encode_packed): nothing is reserved; the wave that starts to spill takes an extent
	// of `need` words - the worst case of its share of the chunk - from a bump cursor in the workspace, once (a returning
	// atomic inside the rare, wave-uniform branch), and an arena that is full raises bit 1 of the error word and leaves the
	// descriptor empty: the stores go nowhere, the segment is reported unusable, nothing is written out of bounds.
	struct Staging
	{
		uint64_t *lds;	   // this wave's region
		uint32_t cap;	   // its capacity in words
		uint32_t lds_used; // words in LDS once spilling has started
		bool spilling;
		__amdgpu_buffer_rsrc_t spill;
		// dynamic spill extents
		uint64_t *arena;			 // nullptr: static
		unsigned long long *cursor;	 // words handed out so far
		uint64_t arena_words;
		uint32_t need;				 // words this wave asks for when it starts to spill
		uint64_t extent;			 // first word of its extent (valid once spilling and granted)
		uint32_t *error_word;
	};

	// first word of an extent of `need` words, ~0 when the arena is full (the launch is marked); wave-uniform
	__device__ __noinline__ uint64_t staging_take_extent(unsigned long long *cursor, uint32_t need, uint64_t arena_words, uint32_t *error_word)
	{
		unsigned long long off = 0;
		if ((threadIdx.x & 63u) == 0)
			off = __hip_atomic_fetch_add(cursor, (unsigned long long)need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off);
		const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(off >> 32));
		off = ((unsigned long long)hi << 32) | lo;
		if (off + need > arena_words)
		{ // no room: the caller's descriptor stays empty (every store is dropped by the range check)
			if ((threadIdx.x & 63u) == 0)
				__hip_atomic_fetch_or(error_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return ~0ull;
		}
		return off;
	}

	__device__ __forceinline__ uint64_t emit_staged(const Px8 &r, uint32_t mode, uint32_t base, Staging &sg, uint32_t pos, const LaneConsts &lc,
													const TransposeConsts &tc, uint32_t *words)
	{
		const uint32_t any = (r.d[0] | r.d[1]) | (r.d[2] | r.d[3]);
		RecordWords rw;
		uint64_t h;
		if (__ballot((any & 0xfff0fff0u) != 0) == 0)
			h = emit_words_narrow(r, mode, base, rw, pos, lc, tc, words);
		else
			h = emit_words_wide(r, mode, base, rw, pos, lc, tc, words);
		if (!sg.spilling && pos + *words > sg.cap)
		{
			sg.spilling = true, sg.lds_used = pos;
			if (sg.arena)
			{
				// (the extent comes back from a call: vector registers, "divergent" for the compiler - and with it the branch below and the
				// descriptor it sets, which then lived in VGPRs and cost every record two waterfall loops, 12 vector + 12 scalar
				// instructions, around its two stores.  It is wave-uniform: say so.)
				const uint64_t ext = staging_take_extent(sg.cursor, sg.need, sg.arena_words, sg.error_word);
				sg.extent = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ext >> 32)) << 32) |
							(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ext);
				if (sg.extent != ~0ull)
					sg.spill = make_rsrc(sg.arena + sg.extent, sg.need * 8u);
			}
		}
		if (!sg.spilling)
		{ // LDS operations only inside this (wave-uniform) branch: the vector-memory stream below stays unconditional
			if (rw.ia != RIR_NONE)
				sg.lds[rw.ia] = ((uint64_t)rw.va.y << 32) | rw.va.x;
			if (rw.ib != RIR_NONE)
				sg.lds[rw.ib] = ((uint64_t)rw.vb.y << 32) | rw.vb.x;
		}
		// two stores per record whatever happens (counted waits, see RecordStores): out of range unless the wave spills
		const uint32_t add = sg.spilling ? 0u - sg.lds_used * 8u : RIR_OOB;
		const uint32_t oa = rw.ia != RIR_NONE ? rw.ia * 8u + add : RIR_OOB;
		const uint32_t ob = rw.ib != RIR_NONE ? rw.ib * 8u + add : RIR_OOB;
		__builtin_amdgcn_raw_buffer_store_b64(rw.va, sg.spill, oa, 0, RIR_SPARSE_STORE_AUX);
		__builtin_amdgcn_raw_buffer_store_b64(rw.vb, sg.spill, ob, 0, RIR_SPARSE_STORE_AUX);
		return h;
	}

	// One wave packs the records of frames [rec0, rec0 + nrec) of a chunk for one tile.  `first` points at the tile in the
	// first frame the wave LOADS: the key frame when has_key, else the frame before its first record.  hdr_first: table
	// entry of that loaded frame (headers of records go to hdr_first[1..] / hdr_first[0] for the key frame).
	// Returns the number of payload words produced.
	template <bool FAST>
	__device__ __forceinline__ uint32_t encode_run(const uint16_t *__restrict__ frames, int64_t npx, int nload, int64_t frame_first, bool has_key,
												   int tile, int lane, uint64_t *__restrict__ hdr_first, Staging &sg)
	{
		const int64_t p0 = (int64_t)tile * RIRB1_TILE_PX + lane * 8;
		const TransposeConsts tc = make_transpose_consts(lane);
		const LaneConsts lc = make_lane_consts(lane);
		const uint32_t lane_off = (uint32_t)lane * 16u;
		const uint16_t *next_ptr = frames + frame_first * npx + (int64_t)tile * RIRB1_TILE_PX;
		int next_f = 0;
		// frames are loaded strictly in order; a request past the wave's last frame is an out-of-range offset: the
		// instruction is issued (the waits stay counted) and touches no memory
		auto load = [&](int f, v4u32 &dst) {
			if (FAST)
			{
				const Px8 p = buf_load8(next_ptr, next_f < nload ? lane_off : RIR_OOB);
				dst.x = p.d[0], dst.y = p.d[1], dst.z = p.d[2], dst.w = p.d[3];
				next_ptr += npx;
				next_f += 1;
			}
			else
			{
				const Px8 p = load8(frames, frame_first + min(f, nload - 1), npx, p0, false);
				dst.x = p.d[0], dst.y = p.d[1], dst.z = p.d[2], dst.w = p.d[3];
			}
		};
		auto as_px8 = [](const v4u32 &v) {
			Px8 p;
			p.d[0] = v.x, p.d[1] = v.y, p.d[2] = v.z, p.d[3] = v.w;
			return p;
		};
		uint32_t pos = 0;
		uint64_t hdr_reg = 0;
		v4u32 s0, s1, s2, s3;
		load(0, s0);
		load(1, s1);
		load(2, s2);
		load(3, s3);
		if (has_key)
		{ // key frame: RAW, or LEFT when its payload is strictly smaller
			const Px8 cur = as_px8(s0);
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
			const uint32_t key_mode = use_left ? RIRB1_MODE_LEFT : RIRB1_MODE_RAW;
			const uint64_t h = emit_staged(r_sel, key_mode, use_left ? base_left : base_raw, sg, pos, lc, tc, &words);
			if (lane == 0)
				hdr_first[0] = h;
			pos += words;
		}
#define RIR_ENC2_STEP(F, CUR, PREV)                                                            \
	{                                                                                          \
		const int f = (F);                                                                     \
		Px8 d;                                                                                 \
		{                                                                                      \
			const Px8 c_ = as_px8(CUR), p_ = as_px8(PREV);                                     \
			_Pragma("unroll") for (int k = 0; k < 4; ++k) d.d[k] = pk_sub16(c_.d[k], p_.d[k]); \
		}                                                                                      \
		load(f + 3, PREV);                                                                     \
		const uint32_t base = tile_base(d, true);                                              \
		const uint32_t b2 = base | (base << 16);                                               \
		_Pragma("unroll") for (int k = 0; k < 4; ++k) d.d[k] = pk_sub16(d.d[k], b2);          \
		uint32_t words;                                                                        \
		const uint64_t h = emit_staged(d, RIRB1_MODE_TEMPORAL, base, sg, pos, lc, tc, &words); \
		if (lane == ((f - 1) & 63))                                                            \
			hdr_reg = h;                                                                       \
		pos += words;                                                                          \
	}
		int fb = 1;
		while (fb < nload)
		{
			const int g0 = fb, gend = min(fb + 64, nload);
			for (; fb + 3 < gend; fb += 4)
			{
				RIR_ENC2_STEP(fb, s1, s0)
				RIR_ENC2_STEP(fb + 1, s2, s1)
				RIR_ENC2_STEP(fb + 2, s3, s2)
				RIR_ENC2_STEP(fb + 3, s0, s3)
			}
			if (fb < gend)
			{
				RIR_ENC2_STEP(fb, s1, s0)
				if (fb + 1 < gend)
				{
					RIR_ENC2_STEP(fb + 1, s2, s1)
					if (fb + 2 < gend)
						RIR_ENC2_STEP(fb + 2, s3, s2)
				}
				fb = gend;
			}
			if (g0 + lane < gend)
				hdr_first[g0 + lane] = hdr_reg;
			hdr_reg = 0;
		}
#undef RIR_ENC2_STEP
		return pos;
	}

	__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
	{
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
			v += (uint64_t)__shfl_xor((long long)v, d, 64);
		return v;
	}

	// Look-back of workgroup (c, t), run by ONE wave.  A = sum of the lengths of tiles < t of chunk c, Pc = first word of
	// chunk c.  false: gave up (clock or error word); the error word is raised.
	__device__ __noinline__ bool encode_lookback(const uint64_t *P, const uint64_t *gtotal, const uint64_t *gran, uint32_t *error_word, int c, int t,
												 int ntiles, int ngroups, int lane, uint64_t *A_out, uint64_t *P_out, uint64_t *dbg_words)
	{
		const int g = t >> 6, e1 = t & 63, E = e1 + g + (c > 0 ? 1 : 0);
		uint64_t accA = 0, accP = 0;
		const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
		for (int e0 = 0; e0 < E; e0 += 64)
		{
			const int e = e0 + lane;
			const bool valid = e < E;
			int kind = 2;
			const uint64_t *ptr = P + c;
			if (e < e1)
				kind = 0, ptr = gran + (int64_t)c * ntiles + ((int64_t)g << 6) + e;
			else if (e < e1 + g)
				kind = 1, ptr = gtotal + (int64_t)c * ngroups + (e - e1);
			uint64_t v = 0;
			bool done = !valid;
			for (uint32_t spins = 0;; ++spins)
			{
				if (!done)
				{
					v = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					done = (v >> 63) != 0;
				}
				if (__ballot(!done) == 0)
					break;
				if ((spins & 15u) == 15u)
				{
					const uint32_t err = __hip_atomic_load(error_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (err != 0 || __builtin_amdgcn_s_memrealtime() - t_start > RIR_LB_TIMEOUT_TICKS)
					{
						if (lane == 0)
							__hip_atomic_store(error_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						return false;
					}
				}
#ifndef RIR_LB_SLEEP
#define RIR_LB_SLEEP 8
#endif
				__builtin_amdgcn_s_sleep(RIR_LB_SLEEP);
			}
			if (valid)
			{
				if (kind == 2)
					accP = v & ~RIR_LB_TAG;
				else
					accA += v & ~RIR_LB_TAG;
			}
		}
#ifdef RIR_DIAG_LB_STATS
		if (lane == 0)
		{ // diagnostic build only: time spent looking back, into the pad at the end of the control block (nothing reads it in the kernel)
			dbg_words[0] = __builtin_amdgcn_s_memrealtime() - t_start; // (the pad word of this segment's spill slot)
			dbg_words[1] = t_start;
		}
#endif
		*A_out = wave_sum_u64(accA);
		*P_out = wave_sum_u64(accP);
		return true;
	}

	// frames of wave w when nf frames are split over `waves` waves with the key frame counted twice: [enc_split(w), enc_split(w + 1))
	__device__ __host__ __forceinline__ int enc_split(int w, int waves, int nf)
	{
		if (w <= 0)
			return 0;
		if (w >= waves)
			return nf;
		const int b = (w * (nf + 1) + waves / 2) / waves - 1;
		return b < 1 ? (nf < 1 ? nf : 1) : (b > nf ? nf : b);
	}

	// control block: 8 ticket heads on lines of their own, the error word, then P / group / gran
	__device__ __host__ __forceinline__ int64_t enc_ctrl_words64(int nchunks, int ntiles)
	{
		const int64_t ngroups = (ntiles + 63) / 64;
		return 8 * 16 + 16 + ((int64_t)nchunks + 1) + (int64_t)nchunks * ngroups + (((int64_t)nchunks * ntiles + 15) & ~(int64_t)15) +
			   (int64_t)nchunks * ngroups + nchunks + 16;
	}

	template <int WAVES>
	__attribute__((amdgpu_waves_per_eu(8, 8))) __global__ __launch_bounds__(WAVES * 64) void rirb1_encode_dense(
		const uint16_t *__restrict__ frames, int64_t npx, int ntiles, int nframes, int gop, int nchunks, uint64_t *__restrict__ hdr_table,
		uint32_t *__restrict__ tile_off, uint64_t *__restrict__ chunk_off, uint64_t *__restrict__ stream, uint64_t *__restrict__ ctrl,
		uint64_t *__restrict__ spill, int cap)
	{
		extern __shared__ __attribute__((aligned(16))) uint64_t enc_lds[];
		uint64_t *sh_base = enc_lds + (size_t)WAVES * cap;				   // [0] segment's first word in the stream
		uint32_t *sh_u32 = reinterpret_cast<uint32_t *>(sh_base + 1); // [0..WAVES) words of each wave, [WAVES] segment index, [WAVES + 2..] words of each wave in LDS
		const int lane = threadIdx.x & 63;
		const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
		const int ngroups = (ntiles + 63) >> 6;
		uint32_t *heads = reinterpret_cast<uint32_t *>(ctrl);
		uint32_t *error_word = reinterpret_cast<uint32_t *>(ctrl + 8 * 16);
		uint64_t *P = ctrl + 8 * 16 + 16;
		uint64_t *gtotal = P + nchunks + 1;
		uint64_t *gran = gtotal + (int64_t)nchunks * ngroups;
		uint64_t *gcount = gran + (((int64_t)nchunks * ntiles + 15) & ~(int64_t)15); // counters: on lines of their own
		uint64_t *ccount = gcount + (int64_t)nchunks * ngroups;

#ifdef RIR_DIAG_NO_TICKET
		const int seg = blockIdx.x;
#else
		if (threadIdx.x == 0)
		{
			const uint32_t head = blockIdx.x & 7u;
			const uint32_t ticket = __hip_atomic_fetch_add(heads + head * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			sh_u32[WAVES] = ticket * 8u + head;
		}
		__syncthreads();
		const int seg = __builtin_amdgcn_readfirstlane((int)sh_u32[WAVES]);
#endif
#ifdef RIR_DIAG_LB_STATS
		const uint64_t dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
		const int chunk = seg / ntiles, tile = seg - chunk * ntiles;
		const int f_begin = chunk * gop;
		const int nf = min(gop, nframes - f_begin);
		uint64_t *my_hdr = hdr_table + (int64_t)seg * gop;

		// this wave's share of the chunk
		const int rec0 = enc_split(w, WAVES, nf), rec1 = enc_split(w + 1, WAVES, nf);
		const int nrec = rec1 - rec0;
		const bool has_key = w == 0;
		const int first_load = has_key ? 0 : rec0 - 1;
		const int nload = nrec > 0 ? rec1 - first_load : 0;
		Staging sg;
		sg.lds = enc_lds + (size_t)w * cap;
		sg.cap = (uint32_t)cap;
		sg.lds_used = 0;
		sg.spilling = false;
		sg.arena = nullptr, sg.cursor = nullptr, sg.arena_words = 0, sg.need = 0, sg.extent = 0, sg.error_word = nullptr;
		uint64_t *my_spill = spill + (int64_t)seg * RIRB1_SLOT_WORDS(gop) + (int64_t)rec0 * RIRB1_REC_MAX_WORDS;
		sg.spill = make_rsrc(my_spill, (uint32_t)(nrec > 0 ? nrec : 0) * RIRB1_REC_MAX_WORDS * 8u);
		uint32_t pos = 0;
		if (nrec > 0)
		{
			const bool fast = ((npx & 7) == 0) && ((int64_t)(tile + 1) * RIRB1_TILE_PX <= npx) && ((((uintptr_t)frames) & 15) == 0);
			if (fast)
				pos = encode_run<true>(frames, npx, nload, (int64_t)f_begin + first_load, has_key, tile, lane, my_hdr + first_load, sg);
			else
				pos = encode_run<false>(frames, npx, nload, (int64_t)f_begin + first_load, has_key, tile, lane, my_hdr + first_load, sg);
		}
		for (int f = nf + (int)threadIdx.x; f < gop; f += WAVES * 64)
			my_hdr[f] = 0; // short last chunk: the unused table entries are defined
		if (lane == 0)
		{
			sh_u32[w] = pos;
			sh_u32[WAVES + 2 + w] = sg.spilling ? sg.lds_used : pos; // words of this wave that are in LDS
		}
		if (sg.spilling)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the spilled words have left this wave before wave 0 is told of them
		__syncthreads();
		// Only wave 0 goes on: the segment is complete in LDS (+ spill), what is left is to wait for its place in the stream and
		// to copy it there.  The other waves END here - a workgroup that waits holds one wave slot, not four, so the CU can
		// start the next workgroup's waves while this one looks back.
		if (w != 0)
			return;
		uint32_t total = 0;
#pragma unroll
		for (int i = 0; i < WAVES; ++i)
			total += sh_u32[i];
		bool chunk_last = false; // this workgroup completed its chunk (lane 0 knows)
		uint64_t chunk_words = 0;
		if (lane == 0)
		{ // publish first, then look back
			__hip_atomic_store(gran + (int64_t)chunk * ntiles + tile, RIR_LB_TAG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const int g = tile >> 6, gsize = min(64, ntiles - (g << 6));
			const uint64_t one = 1ull << 40, sum_mask = one - 1ull;
			const uint64_t og = __hip_atomic_fetch_add(gcount + (int64_t)chunk * ngroups + g, one | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((int)(og >> 40) == gsize - 1)
			{ // last arriver of its group: the group's total becomes visible; it may complete the chunk too
				const uint64_t gsum = (og & sum_mask) + total;
				__hip_atomic_store(gtotal + (int64_t)chunk * ngroups + g, RIR_LB_TAG | gsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint64_t oc = __hip_atomic_fetch_add(ccount + chunk, one | gsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((int)(oc >> 40) == ngroups - 1)
					chunk_last = true, chunk_words = (oc & sum_mask) + gsum;
			}
		}
		uint64_t A = 0, Pc = 0;
#ifdef RIR_DIAG_NO_LOOKBACK
		const bool ok = true;
#else
		const bool ok = encode_lookback(P, gtotal, gran, error_word, chunk, tile, ntiles, ngroups, lane, &A, &Pc,
										spill + (int64_t)seg * RIRB1_SLOT_WORDS(gop) + (int64_t)gop * RIRB1_REC_MAX_WORDS);
#endif
		if (!ok)
			return; // the look-back gave up: the error word is raised, nothing is copied
#ifdef RIR_DIAG_LB_STATS
		const uint64_t dbg_t2 = __builtin_amdgcn_s_memrealtime();
#endif
		if (lane == 0)
		{
			tile_off[(int64_t)chunk * (ntiles + 1) + tile] = (uint32_t)A;
			if (seg == 0)
				chunk_off[0] = 0;
			if (chunk_last)
			{ // the chunk's length is known and so is its first word: the next chunk's first word follows
				tile_off[(int64_t)chunk * (ntiles + 1) + ntiles] = (uint32_t)chunk_words;
				__hip_atomic_store(P + chunk + 1, RIR_LB_TAG | (Pc + chunk_words), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				chunk_off[chunk + 1] = Pc + chunk_words;
			}
		}
#ifdef RIR_DIAG_NO_COPY
		return;
#endif
		uint64_t *dst = stream + (Pc + A);
		for (int i = 0; i < WAVES; ++i)
		{
			const uint64_t *src = enc_lds + (size_t)i * cap;
			const uint32_t n_all = sh_u32[i], n_lds = sh_u32[WAVES + 2 + i];
			uint32_t j = (uint32_t)lane;
			for (; j + 192 < n_lds; j += 256)
			{
				const uint64_t v0 = src[j], v1 = src[j + 64], v2 = src[j + 128], v3 = src[j + 192];
				dst[j] = v0, dst[j + 64] = v1, dst[j + 128] = v2, dst[j + 192] = v3;
			}
			for (; j < n_lds; j += 64)
				dst[j] = src[j];
			if (n_all > n_lds)
			{ // what wave i spilled comes back from its area of the workspace (sc1 loads: from L2, where its stores went)
				const int r0 = enc_split(i, WAVES, nf), r1 = enc_split(i + 1, WAVES, nf);
				const __amdgpu_buffer_rsrc_t sp =
					make_rsrc(spill + (int64_t)seg * RIRB1_SLOT_WORDS(gop) + (int64_t)r0 * RIRB1_REC_MAX_WORDS, (uint32_t)(r1 - r0) * RIRB1_REC_MAX_WORDS * 8u);
				for (uint32_t q = (uint32_t)lane; q < n_all - n_lds; q += 64)
				{
					const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(sp, q * 8u, 0, 16 /* sc1 */);
					dst[n_lds + q] = ((uint64_t)v.y << 32) | v.x;
				}
			}
			dst += n_all;
		}
#ifdef RIR_DIAG_LB_STATS
		if (lane == 0)
		{
			uint64_t *dbg = spill + (int64_t)seg * RIRB1_SLOT_WORDS(gop) + (int64_t)gop * RIRB1_REC_MAX_WORDS;
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			dbg[2] = dbg_t0, dbg[3] = dbg_t2, dbg[4] = __builtin_amdgcn_s_memrealtime();
			unsigned xcc;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
			unsigned hwid;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 32)" : "=s"(hwid));
			dbg[5] = ((uint64_t)xcc << 32) | hwid;
		}
#endif
	}

	// ==== packed form: the dense stream without the order ===============================================
	//
	// What made the single-pass DENSE encoder slow was not the staging but the ORDER: a segment's place is the sum of all
	// lengths in front of it, so every workgroup waits for the slowest of its predecessors.  The
	// reference's own ZFile container does not order its records by anything either: it keeps a table of positions
	// (ZFile.cpp:434-447).  The PACKED form does the same on the device: a segment is packed into LDS exactly as above, and
	// when its length is known ONE returning atomic add on a cursor hands it the next free words of the stream - first come,
	// first placed, nobody waits for anybody.  The batch is
	//   hdr       [nchunks][ntiles][gop]  u64   record headers (as everywhere)
	//   seg_pos   [nchunks][ntiles]       u64   first word of the segment in `stream`
	//   seg_words [nchunks][ntiles]       u32   its length
	//   stream    cursor words, no holes: the segments in order of arrival
	// i.e. exactly C bytes of payload plus tables - what can be kept, sent or written - from ONE pass over the frames.  The
	// layout differs from run to run (arrival order); every segment's words are the oracle's.
	// ONE cursor is not enough: 12 800 returning atomic adds on one address take 16 ns each at the memory side - 205 us, longer
	// than the kernel (measured: 207 us with one cursor, 162-165 with two or more or with none).  So there are TWO, and so that
	// they share the caller's capacity instead of halving it they work from its two ends: segments with an even (tile + chunk)
	// are placed upwards from word 0, the others downwards from word `capacity`.  The batch is two extents without holes,
	// [0, low) and [capacity - high, capacity); it fits when low + high <= capacity.
	// ctrl (zeroed before the launch, a line each): [0] low cursor, [16] high cursor, [32] spill cursor, [48] error word (u32): bit 0
	// = a segment did not fit its side (the cursors still say how many words the batch needs), bit 1 = the spill arena was
	// exceeded; [64] the capacity the launch was given (for the status call: the extents cross when low + high > capacity).
#ifndef RIR_PACKED_COPY_ALL_WAVES
#define RIR_PACKED_COPY_ALL_WAVES 0
#endif
#ifndef RIR_PACKED_WAVES
#define RIR_PACKED_WAVES 4
#endif
	// Two kernels, as for rirb1_encode_tiles: FAST for the tiles that lie whole inside 16-byte aligned frames, the ragged form for
	// the last tile of a frame whose size is not a multiple of 512 pixels.  grid = (tile_count, nchunks).
	template <int WAVES, bool FAST>
	__attribute__((amdgpu_waves_per_eu(8, 8))) __global__ __launch_bounds__(WAVES * 64) void rirb1_encode_packed(
		const uint16_t *__restrict__ frames, int64_t npx, int ntiles, int tile_first, int nframes, int gop, uint64_t *__restrict__ hdr_table,
		uint64_t *__restrict__ seg_pos, uint32_t *__restrict__ seg_words, uint64_t *__restrict__ stream, uint64_t capacity_words,
		uint64_t *__restrict__ ctrl, uint64_t *__restrict__ arena, uint64_t arena_words, int cap, int diag)
	{
		extern __shared__ __attribute__((aligned(16))) uint64_t enc_lds[];
		uint64_t *sh_u64 = enc_lds + (size_t)WAVES * cap;			  // [0] segment's first word in the stream, [1 + w] wave w's spill extent
		uint32_t *sh_u32 = reinterpret_cast<uint32_t *>(sh_u64 + 1 + WAVES); // [0..WAVES) words of each wave, [WAVES..2 WAVES) of them in LDS
		const int lane = threadIdx.x & 63;
		const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
		const bool downwards = ((blockIdx.x + blockIdx.y) & 1u) != 0 && diag != 1; // (diag 1: one cursor)
		unsigned long long *cursor = reinterpret_cast<unsigned long long *>(ctrl) + (downwards ? 16 : 0);
		unsigned long long *spill_cursor = reinterpret_cast<unsigned long long *>(ctrl + 32);
		uint32_t *error_word = reinterpret_cast<uint32_t *>(ctrl + 48);
		if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
			ctrl[64] = capacity_words;

		const int chunk = blockIdx.y, tile = tile_first + blockIdx.x;
		const int64_t seg = (int64_t)chunk * ntiles + tile;
		const int f_begin = chunk * gop;
		const int nf = min(gop, nframes - f_begin);
		uint64_t *my_hdr = hdr_table + seg * gop;

		const int rec0 = enc_split(w, WAVES, nf), rec1 = enc_split(w + 1, WAVES, nf);
		const int nrec = rec1 - rec0;
		const bool has_key = w == 0;
		const int first_load = has_key ? 0 : rec0 - 1;
		const int nload = nrec > 0 ? rec1 - first_load : 0;
		Staging sg;
		sg.lds = enc_lds + (size_t)w * cap;
		sg.cap = (uint32_t)cap;
		sg.lds_used = 0;
		sg.spilling = false;
		sg.spill = make_rsrc(arena, 0); // empty until the wave takes an extent
		sg.arena = arena, sg.cursor = spill_cursor, sg.arena_words = arena_words, sg.error_word = error_word;
		sg.need = (uint32_t)(nrec > 0 ? nrec : 0) * RIRB1_REC_MAX_WORDS;
		sg.extent = ~0ull;
		uint32_t pos = 0;
		if (nrec > 0)
			pos = encode_run<FAST>(frames, npx, nload, (int64_t)f_begin + first_load, has_key, tile, lane, my_hdr + first_load, sg);
		for (int f = nf + (int)threadIdx.x; f < gop; f += WAVES * 64)
			my_hdr[f] = 0; // short last chunk: the unused table entries are defined
		if (lane == 0)
		{
			sh_u32[w] = pos;
			sh_u32[WAVES + w] = sg.spilling ? sg.lds_used : pos;
			sh_u64[1 + w] = sg.extent;
		}
		if (sg.spilling)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the spilled words have left this wave before another wave reads them
		__syncthreads();
#if !RIR_PACKED_COPY_ALL_WAVES
		if (w != 0)
			return; // the segment is complete in LDS (+ extents): one wave places and copies it, the others make room
#endif
		uint32_t total = 0, before = 0;
		bool lost = false; // a wave of this segment wanted an extent and got none
		(void)before;
#pragma unroll
		for (int i = 0; i < WAVES; ++i)
		{
			const uint32_t n = sh_u32[i];
			before += i < w ? n : 0u;
			total += n;
			lost |= n > sh_u32[WAVES + i] && sh_u64[1 + i] == ~0ull;
		}
		if (w == 0)
		{
			unsigned long long at = 0;
			if (lane == 0)
				at = __hip_atomic_fetch_add(cursor, (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)at);
			const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(at >> 32));
			at = ((unsigned long long)hi << 32) | lo;
			const bool fits = at + total <= capacity_words;
			if (downwards)
				at = capacity_words - at - total; // (unused when it does not fit)
			if (lane == 0)
			{
				seg_pos[seg] = at;
				seg_words[seg] = total;
				if (!fits)
					__hip_atomic_fetch_or(error_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				sh_u64[0] = (fits && !lost) ? at : ~0ull;
			}
#if !RIR_PACKED_COPY_ALL_WAVES
			if (!fits || lost)
				return;
			uint64_t *dst = stream + at;
			for (int i = 0; i < WAVES; ++i)
			{
				const uint64_t *src = enc_lds + (size_t)i * cap;
				const uint32_t n_all = sh_u32[i], n_lds = sh_u32[WAVES + i];
				uint32_t j = (uint32_t)lane;
				for (; j + 192 < n_lds; j += 256)
				{
					const uint64_t v0 = src[j], v1 = src[j + 64], v2 = src[j + 128], v3 = src[j + 192];
					dst[j] = v0, dst[j + 64] = v1, dst[j + 128] = v2, dst[j + 192] = v3;
				}
				for (; j < n_lds; j += 64)
					dst[j] = src[j];
				if (n_all > n_lds)
				{ // what wave i spilled comes back from its extent (sc1 loads: from L2, where its stores went)
					const int r0 = enc_split(i, WAVES, nf), r1 = enc_split(i + 1, WAVES, nf);
					const __amdgpu_buffer_rsrc_t sp = make_rsrc(arena + sh_u64[1 + i], (uint32_t)(r1 - r0) * RIRB1_REC_MAX_WORDS * 8u);
					for (uint32_t q = (uint32_t)lane; q < n_all - n_lds; q += 64)
					{
						const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(sp, q * 8u, 0, 16 /* sc1 */);
						dst[n_lds + q] = ((uint64_t)v.y << 32) | v.x;
					}
				}
				dst += n_all;
			}
#endif
		}
#if RIR_PACKED_COPY_ALL_WAVES
		__syncthreads();
		const uint64_t at = sh_u64[0];
		if (at == ~0ull)
			return;
		{ // every wave moves its own words
			uint64_t *dst = stream + at + before;
			const uint64_t *src = enc_lds + (size_t)w * cap;
			const uint32_t n_all = sh_u32[w], n_lds = sh_u32[WAVES + w];
			uint32_t j = (uint32_t)lane;
			for (; j + 192 < n_lds; j += 256)
			{
				const uint64_t v0 = src[j], v1 = src[j + 64], v2 = src[j + 128], v3 = src[j + 192];
				dst[j] = v0, dst[j + 64] = v1, dst[j + 128] = v2, dst[j + 192] = v3;
			}
			for (; j < n_lds; j += 64)
				dst[j] = src[j];
			if (n_all > n_lds)
			{
				const __amdgpu_buffer_rsrc_t sp = make_rsrc(arena + sh_u64[1 + w], (uint32_t)nrec * RIRB1_REC_MAX_WORDS * 8u);
				for (uint32_t q = (uint32_t)lane; q < n_all - n_lds; q += 64)
				{
					const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(sp, q * 8u, 0, 16 /* sc1 */);
					dst[n_lds + q] = ((uint64_t)v.y << 32) | v.x;
				}
			}
		}
#endif
	}

	// ---- decode -----------------------------------------------------------------------------
	// One record in flight: header + the two plane words of this lane.
	struct Fetched
	{
		uint64_t hdr;
		v2u32 a, b; // wide tier: the two plane words of this lane; narrow tier: a = plane word (lanes 0..31)
		uint32_t words;
		bool bad;
		bool narrow; // every width <= 4 (wave-uniform)
	};

	// true when every 5-bit width field of the header is <= 4 (scalar-unit arithmetic on the uniform header)
	__device__ __forceinline__ bool header_is_narrow(uint64_t hdr)
	{
		const uint64_t m0 = 0x0021002100210021ull; // bit 0 of w_q and of w_{4+q} in each 16-bit field
		const uint64_t hi = hdr & (m0 * 0x18u);	   // bits 3,4 of any width
		const uint64_t b2 = (hdr >> 2) & m0;
		const uint64_t lo = (hdr | (hdr >> 1)) & m0;
		return (hi | (b2 & lo)) == 0;
	}

	__device__ __forceinline__ Fetched fetch_record(uint64_t hdr, __amdgpu_buffer_rsrc_t in, uint32_t pos, uint32_t seg_len, const LaneConsts &lc)
	{
		Fetched r;
		r.hdr = hdr;
		r.narrow = header_is_narrow(hdr);
		const bool hdr_bad = ((hdr & 0xC000C000C0000000ull) != 0) || (((hdr >> 14) & 3u) == 3u);
		// Offsets depend on the tier; the two loads below do not (no branch around a memory operation,
		// no copy of an in-flight load result: the ring keeps its counted waits).  Lanes without a plane
		// - and every lane of a malformed record - point out of range and read 0; the descriptor's
		// num_records is the segment length, so nothing outside the segment is ever touched.
		uint32_t oa, ob;
		if (r.narrow)
		{ // lanes 0..31: lane 4*slot + plane reads the whole 64-bit plane word
			const uint32_t w = ((lc.n_hhi ? (uint32_t)(hdr >> 32) : (uint32_t)hdr) >> lc.n_hsh) & 31u;
			const uint32_t incl = half_inclusive_sum(lc.n_first ? w : 0u);
			r.words = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
			r.bad = hdr_bad || (pos + r.words > seg_len);
			oa = (!r.bad && lc.n_half == 0 && lc.n_bit < w) ? (pos + incl - w + lc.n_bit) * 8u : RIR_OOB;
			ob = RIR_OOB;
		}
		else
		{ // wide tier: this row's header field is w_q | w_{4+q} << 5 | base nibble << 10 | mode << 14
			const uint32_t field = ((lc.row_hi ? (uint32_t)(hdr >> 32) : (uint32_t)hdr) >> lc.sh16) & 0xffffu;
			const uint32_t wa = field & 31u, wb = (field >> 5) & 31u;
			const uint32_t ia = rows_inclusive_sum(wa), ib = rows_inclusive_sum(wb);
			const uint32_t tot_a = (uint32_t)__builtin_amdgcn_readlane((int)ia, 63);
			const uint32_t tot_b = (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
			r.words = tot_a + tot_b;
			const bool too_wide = __ballot(wa > 16u || wb > 16u) != 0;
			r.bad = hdr_bad || too_wide || (pos + r.words > seg_len);
			oa = (!r.bad && lc.bit < wa) ? (pos + ia - wa + lc.bit) * 8u : RIR_OOB;
			ob = (!r.bad && lc.bit < wb) ? (pos + tot_a + ib - wb + lc.bit) * 8u : RIR_OOB;
		}
		// exactly two loads per record, whatever the tier (the consumer's wait counts younger loads)
		r.a = __builtin_amdgcn_raw_buffer_load_b64(in, oa, 0, RIR_STREAM_LOAD_AUX);
		r.b = __builtin_amdgcn_raw_buffer_load_b64(in, ob, 0, RIR_STREAM_LOAD_AUX);
		return r;
	}

	// residuals of one record back in pixel order (packed pairs), from the fetched plane words
	__device__ __forceinline__ Px8 planes_to_residuals(v2u32 a, v2u32 b, bool narrow, const TransposeConsts &tc)
	{
		Px8 r;
		if (narrow)
		{
			// upper-half lanes take the high dword of their partner's plane word
			auto sw = __builtin_amdgcn_permlane32_swap(a.x, a.y, false, false);
			const uint32_t x = transpose32(sw[0], tc); // nibble j = residual of pixel j
			const uint32_t lo = x & 0x0f0f0f0fu, hi = (x >> 4) & 0x0f0f0f0fu; // bytes r0,r2,r4,r6 / r1,r3,r5,r7
			r.d[0] = __builtin_amdgcn_perm(hi, lo, 0x0c040c00u);
			r.d[1] = __builtin_amdgcn_perm(hi, lo, 0x0c050c01u);
			r.d[2] = __builtin_amdgcn_perm(hi, lo, 0x0c060c02u);
			r.d[3] = __builtin_amdgcn_perm(hi, lo, 0x0c070c03u);
		}
		else
		{
			uint32_t alo = a.x, ahi = a.y, blo = b.x, bhi = b.y;
			transpose64x2(alo, ahi, blo, bhi, tc);
			r.d[0] = alo, r.d[1] = ahi, r.d[2] = blo, r.d[3] = bhi;
		}
		return r;
	}

	__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
	{
		return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) |
			   ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
	}

	// MODE_LEFT reconstruction: inclusive prefix sum (mod 2^16) over the tile, lane-local then across lanes
	__device__ __forceinline__ Px8 left_integrate(const Px8 &d, int lane)
	{
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
		Px8 o;
#pragma unroll
		for (int k = 0; k < 4; ++k)
			o.d[k] = ((a[2 * k] + carry) & 0xffffu) | (((a[2 * k + 1] + carry) & 0xffffu) << 16);
		return o;
	}

	template <bool FAST>
	__device__ __forceinline__ void decode_tile(const uint64_t *__restrict__ my_hdr, const uint64_t *__restrict__ in_ptr, uint32_t seg_len,
												int64_t npx, int nf, int64_t frame0, int tile, int lane, uint16_t *__restrict__ frames,
												int *__restrict__ error_flag)
	{
		const int64_t p0 = (int64_t)tile * RIRB1_TILE_PX + lane * 8;
		const TransposeConsts tc = make_transpose_consts(lane);
		const LaneConsts lc = make_lane_consts(lane);
		const __amdgpu_buffer_rsrc_t in = make_rsrc(in_ptr, seg_len * 8u);
		const uint32_t lane_off = (uint32_t)lane * 16u;
		uint16_t *tile0 = frames + frame0 * npx + (int64_t)tile * RIRB1_TILE_PX;

		Px8 prev;
		prev.d[0] = prev.d[1] = prev.d[2] = prev.d[3] = 0;
		uint32_t pos = 0;  // fetch position: words of every record requested so far
		bool err = false; // malformed input seen (wave-uniform); reported once, at the end
		for (int f0 = 0; f0 < nf; f0 += 64)
		{
			const int nr = min(64, nf - f0);
			const uint64_t my_h = lane < nr ? my_hdr[f0 + lane] : 0ull; // 64 headers, one coalesced load
			// Ring of 4 records in flight (fixed registers, loop unrolled by 4, see the encoder).
			// Requests past the end of the round use an all-zero header = empty record (no memory
			// touched); a malformed record points every lane out of range, so the loop body has no
			// data-dependent branch and the compiler keeps counted s_waitcnt vmcnt(N).
			Fetched r0 = fetch_record(readlane64(my_h, 0), in, pos, seg_len, lc);
			pos += r0.words;
			Fetched r1 = fetch_record(nr > 1 ? readlane64(my_h, 1) : 0ull, in, pos, seg_len, lc);
			pos += r1.words;
			Fetched r2 = fetch_record(nr > 2 ? readlane64(my_h, 2) : 0ull, in, pos, seg_len, lc);
			pos += r2.words;
			Fetched r3 = fetch_record(nr > 3 ? readlane64(my_h, 3) : 0ull, in, pos, seg_len, lc);
			pos += r3.words;
#define RIR_DEC_STEP(FR, R)                                                                                          \
	if ((FR) < nr)                                                                                                   \
	{                                                                                                                \
		const int fr = (FR);                                                                                         \
		const uint64_t hdr = R.hdr;                                                                                  \
		err |= R.bad;                                                                                                \
		const v2u32 wa_ = R.a, wb_ = R.b;                                                                             \
		const bool narrow_ = R.narrow;                                                                               \
		/* refill the slot with the record four frames ahead */                                                       \
		R = fetch_record(fr + 4 < nr ? readlane64(my_h, fr + 4) : 0ull, in, pos, seg_len, lc);                        \
		pos += R.words;                                                                                              \
		const Px8 res = planes_to_residuals(wa_, wb_, narrow_, tc);                                                  \
		const uint32_t mode = (uint32_t)(hdr >> 14) & 3u;                                                            \
		const uint32_t base = ((uint32_t)(hdr >> 10) & 0xfu) | ((uint32_t)(hdr >> 22) & 0xf0u) |                     \
							  ((uint32_t)(hdr >> 34) & 0xf00u) | ((uint32_t)(hdr >> 46) & 0xf000u);                  \
		const uint32_t b2 = base | (base << 16);                                                                     \
		Px8 o;                                                                                                       \
		_Pragma("unroll") for (int k = 0; k < 4; ++k) o.d[k] = pk_add16(res.d[k], b2);                              \
		if (f0 + fr == 0)                                                                                            \
		{ /* key frame: RAW or LEFT */                                                                               \
			if (mode == RIRB1_MODE_LEFT)                                                                             \
				o = left_integrate(o, lane);                                                                         \
			err |= (mode == RIRB1_MODE_TEMPORAL);                                                                    \
		}                                                                                                            \
		else                                                                                                         \
		{ /* TEMPORAL (or RAW); LEFT is only legal on the key frame */                                               \
			const uint32_t keep = (mode == RIRB1_MODE_TEMPORAL) ? 0xffffffffu : 0u;                                  \
			_Pragma("unroll") for (int k = 0; k < 4; ++k) o.d[k] = pk_add16(prev.d[k] & keep, o.d[k]);              \
			err |= (mode == RIRB1_MODE_LEFT);                                                                        \
		}                                                                                                            \
		if (FAST)                                                                                                    \
			buf_store8(tile0 + (int64_t)(f0 + fr) * npx, lane_off, o);                                               \
		else                                                                                                         \
			store8(frames, frame0 + f0 + fr, npx, p0, false, o);                                                     \
		prev = o;                                                                                                    \
	}
			for (int fb = 0; fb < nr; fb += 4)
			{
				RIR_DEC_STEP(fb, r0)
				RIR_DEC_STEP(fb + 1, r1)
				RIR_DEC_STEP(fb + 2, r2)
				RIR_DEC_STEP(fb + 3, r3)
			}
#undef RIR_DEC_STEP
		}
		if ((err || pos != seg_len) && lane == 0)
			atomicExch(error_flag, 1);
	}

	// grid = (ceil(ntiles/4), nchunks), block = 256 (4 independent waves)
	// Two layouts of the payload, one record format:
	//   dense   (seg_words == NULL)  segment (c, t) = stream[chunk_off[c] + tile_off[c][t] ...) - the file form, tables untrusted
	//   slotted (seg_words != NULL)  segment (c, t) = stream[(c * ntiles + t) * RIRB1_SLOT_WORDS(gop) ...), seg_words[c][t] words
	//            long - what rirb1_encode_tiles leaves in its workspace: every segment's place is known before anything is
	//            packed, so the encoder needs no second pass and the decoder no offsets (tile_off / chunk_off unused).
	//   packed  (seg_pos != NULL too) segment (c, t) = stream[seg_pos[c][t] ...), seg_words[c][t] words long - what rirb1_encode_packed
	//            leaves: a stream without holes whose segments lie in order of arrival.
	__global__ __launch_bounds__(256) void rirb1_decode_tiles(const uint64_t *__restrict__ hdr_table, const uint32_t *__restrict__ tile_off,
															 const uint64_t *__restrict__ chunk_off, const uint64_t *__restrict__ stream,
															 uint64_t stream_words, int64_t npx, int ntiles, int nframes, int gop,
															 const int64_t *__restrict__ chunk_frames, const uint32_t *__restrict__ seg_words,
															 const uint64_t *__restrict__ seg_pos, uint16_t *__restrict__ frames,
															 int *__restrict__ error_flag)
	{
		const int lane = threadIdx.x & 63;
		const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
		if (tile >= ntiles)
			return;
		const int chunk = blockIdx.y;
		// chunk -> frames of the output: consecutive runs of `gop` frames, or (chunk_frames != NULL, nframes = capacity of
		// `frames`) an explicit (first frame, frame count) pair per chunk - chunks gathered from several shards decode
		// straight to their place in the reassembled stream
		int64_t f_begin = (int64_t)chunk * gop;
		int nf = (int)min((int64_t)gop, (int64_t)nframes - f_begin);
		if (chunk_frames)
		{
			const int64_t cf = chunk_frames[2 * chunk], cn = chunk_frames[2 * chunk + 1];
			if (cn == 0)
				return; // padding entry
			if (cf < 0 || cn < 0 || cn > gop || cf > (int64_t)nframes - cn)
			{
				if (lane == 0)
					atomicExch(error_flag, 1);
				return;
			}
			f_begin = cf, nf = (int)cn;
		}
		const int64_t slot = (int64_t)chunk * ntiles + tile;
		const uint64_t *my_hdr = hdr_table + slot * gop;
		uint32_t seg_len;
		const uint64_t *in;
		if (seg_words)
		{ // slotted: the slot's place is fixed, its length comes from the encoder's table (clamped to the slot all the same);
		  // packed (seg_pos != NULL): the place comes from a table too - untrusted like the length (a batch received from another
		  // device), so the segment must lie inside the stream before a descriptor is built on it
			const uint64_t base = seg_pos ? seg_pos[slot] : (uint64_t)slot * (uint64_t)RIRB1_SLOT_WORDS(gop);
			seg_len = min(seg_words[slot], (uint32_t)gop * RIRB1_REC_MAX_WORDS);
			if (base > stream_words || (uint64_t)seg_len > stream_words - base)
			{
				if (lane == 0)
					atomicExch(error_flag, 1);
				return;
			}
			in = stream + base;
		}
		else
		{
			const uint32_t t0 = tile_off[(int64_t)chunk * (ntiles + 1) + tile];
			const uint32_t t1 = tile_off[(int64_t)chunk * (ntiles + 1) + tile + 1];
			const uint64_t c0 = chunk_off[chunk], c1 = chunk_off[chunk + 1];
			// The tables are untrusted (they come from a file): the segment [c0 + t0, c0 + t1) must lie inside the chunk
			// [c0, c1) and the chunk inside the stream allocation BEFORE a descriptor is built on it - the descriptor's
			// num_records only bounds accesses relative to its base.
			if (c0 > c1 || c1 > stream_words || t1 < t0 || (uint64_t)t1 > c1 - c0)
			{ // malformed offsets tables
				if (lane == 0)
					atomicExch(error_flag, 1);
				return;
			}
			seg_len = min(t1 - t0, (uint32_t)gop * RIRB1_REC_MAX_WORDS);
			in = stream + c0 + t0;
		}
		const bool fast = ((npx & 7) == 0) && ((int64_t)(tile + 1) * RIRB1_TILE_PX <= npx) && ((((uintptr_t)frames) & 15) == 0);
		if (fast)
			decode_tile<true>(my_hdr, in, seg_len, npx, nf, f_begin, tile, lane, frames, error_flag);
		else
			decode_tile<false>(my_hdr, in, seg_len, npx, nf, f_begin, tile, lane, frames, error_flag);
	}

	// ---- host launchers --------------------------------------------------------------------------------

	// stage 1: one pass over the raw frames -> headers, per-tile segment lengths, sparse payload
	hipError_t launch_encode_tiles(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint64_t *d_hdr,
								   uint32_t *d_seg_words, uint64_t *d_sparse, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		const bool aligned = ((npx & 7) == 0) && ((((uintptr_t)d_frames) & 15) == 0);
		const int nfast = aligned ? (int)(npx / RIRB1_TILE_PX) : 0; // tiles that lie whole inside the frame
		dim3 block(256);
		if (nfast > 0)
			hipLaunchKernelGGL(rirb1_encode_tiles<true>, dim3((nfast + 3) / 4, nchunks), block, 0, st, d_frames, npx, ntiles, 0, nfast, nframes, gop, d_hdr,
							   d_seg_words, d_sparse);
		if (ntiles > nfast)
			hipLaunchKernelGGL(rirb1_encode_tiles<false>, dim3((ntiles - nfast + 3) / 4, nchunks), block, 0, st, d_frames, npx, ntiles, nfast, ntiles - nfast,
							   nframes, gop, d_hdr, d_seg_words, d_sparse);
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

	// single-pass dense encode: control block zeroed by a memset node, then one kernel.  d_ctrl: enc_ctrl_words64() x 8 bytes at
	// the start of an allocation; d_spill: one RIRB1_SLOT_WORDS(gop) slot per (chunk, tile).
	int64_t encode_ctrl_bytes(int nchunks, int ntiles) { return enc_ctrl_words64(nchunks, ntiles) * 8; }
	hipError_t launch_encode_dense(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint64_t *d_hdr, uint32_t *d_tile_off,
								   uint64_t *d_chunk_off, uint64_t *d_stream, uint64_t *d_ctrl, uint64_t *d_spill, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		const int64_t ctrl_bytes = (encode_ctrl_bytes(nchunks, ntiles) + 15) & ~(int64_t)15;
		hipError_t e = hipMemsetAsync(d_ctrl, 0, (size_t)ctrl_bytes, st);
		if (e != hipSuccess)
			return e;
		constexpr int WAVES = 4;
		// LDS words per wave: what 8 workgroups per CU leave (160 KiB / 8 = 20 KiB per workgroup), never more than the
		// worst case of the wave's share of the chunk
		const int share = (gop + 1 + WAVES - 1) / WAVES + 1;
		int cap = 448; // 14.4 KB per workgroup: 11 workgroups per CU, 7 packing (28 waves) while 4 wait for their offsets (1 wave each)
		if (cap > share * RIRB1_REC_MAX_WORDS)
			cap = share * RIRB1_REC_MAX_WORDS;
		static int cap_env = -1;
		if (cap_env < 0)
		{
			const char *ev = getenv("RIR_ENC_LDS_WORDS"); // tuning aid
			cap_env = ev ? atoi(ev) : 0;
		}
		if (cap_env > 0)
			cap = cap_env;
		const size_t lds = (size_t)WAVES * cap * 8 + 8 + (2 * WAVES + 2) * 4;
		const int64_t total = (int64_t)nchunks * ntiles;
		// its workgroups wait for the segments in front of them (dealt by tickets, so any grid size makes progress on its own - but
		// not beside a launch that holds the chip waiting for ITS missing workgroups): through the process-wide gate (runtime.h)
		ResidentGate gate(st);
		if (!gate.ok())
			return hipErrorUnknown;
		hipLaunchKernelGGL(rirb1_encode_dense<WAVES>, dim3((unsigned)total), dim3(WAVES * 64), lds, st, d_frames, npx, ntiles, nframes, gop, nchunks, d_hdr,
						   d_tile_off, d_chunk_off, d_stream, d_ctrl, d_spill, cap);
		return hipGetLastError();
	}

	hipError_t launch_decode(const uint64_t *d_hdr, const uint32_t *d_tile_off, const uint64_t *d_chunk_off, const uint64_t *d_stream,
							 uint64_t stream_words, int64_t npx, int ntiles, int nframes, int gop, const int64_t *d_chunk_frames, int nchunks_tab,
							 uint16_t *d_frames, int *d_error, hipStream_t st)
	{
		const int nchunks = d_chunk_frames ? nchunks_tab : (nframes + gop - 1) / gop;
		dim3 grid((ntiles + 3) / 4, nchunks), block(256);
		hipLaunchKernelGGL(rirb1_decode_tiles, grid, block, 0, st, d_hdr, d_tile_off, d_chunk_off, d_stream, stream_words, npx, ntiles, nframes, gop,
						   d_chunk_frames, (const uint32_t *)nullptr, (const uint64_t *)nullptr, d_frames, d_error);
		return hipGetLastError();
	}
	// the slotted form: d_slots = the encoder's slot array ([nchunks][ntiles] slots of RIRB1_SLOT_WORDS(gop) words), d_seg_words its lengths
	hipError_t launch_decode_slots(const uint64_t *d_hdr, const uint32_t *d_seg_words, const uint64_t *d_slots, int64_t npx, int ntiles, int nframes,
								   int gop, uint16_t *d_frames, int *d_error, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		dim3 grid((ntiles + 3) / 4, nchunks), block(256);
		hipLaunchKernelGGL(rirb1_decode_tiles, grid, block, 0, st, d_hdr, (const uint32_t *)nullptr, (const uint64_t *)nullptr, d_slots,
						   (uint64_t)nchunks * ntiles * (uint64_t)RIRB1_SLOT_WORDS(gop), npx, ntiles, nframes, gop, (const int64_t *)nullptr,
						   d_seg_words, (const uint64_t *)nullptr, d_frames, d_error);
		return hipGetLastError();
	}

	// the packed form.  d_ctrl: RIRB1_PACKED_CTRL_BYTES at the start of the workspace (zeroed here), d_arena: the rest of it.
	int packed_lds_words(int gop)
	{
		constexpr int WAVES = RIR_PACKED_WAVES;
		const int share = (gop + 1 + WAVES - 1) / WAVES + 1;
		int cap = 448 * 4 / WAVES; // 14.4 KB per workgroup, as the dense kernel: the reference's recipe needs 290-355 words per wave of four
		if (cap > share * RIRB1_REC_MAX_WORDS)
			cap = share * RIRB1_REC_MAX_WORDS;
		static int cap_env = -1;
		if (cap_env < 0)
		{
			const char *ev = getenv("RIR_ENC_LDS_WORDS"); // tuning aid
			cap_env = ev ? atoi(ev) : 0;
		}
		return cap_env > 0 ? cap_env : cap;
	}
	hipError_t launch_encode_packed(const uint16_t *d_frames, int64_t npx, int ntiles, int nframes, int gop, uint64_t *d_hdr, uint64_t *d_seg_pos,
									uint32_t *d_seg_words, uint64_t *d_stream, uint64_t capacity_words, uint64_t *d_ctrl, uint64_t *d_arena,
									uint64_t arena_words, bool reset, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		if (reset)
		{
			hipError_t e = hipMemsetAsync(d_ctrl, 0, RIRB1_PACKED_CTRL_BYTES, st);
			if (e != hipSuccess)
				return e;
		}
		constexpr int WAVES = RIR_PACKED_WAVES;
		const int cap = packed_lds_words(gop);
		// (a build with -DRIR_PACKED_ONE_CURSOR sends every segment through the low cursor: the measurement of DESIGN.md §3 - 207 us a launch
		// against 162-165 with two; the product library reads no such switch)
#ifdef RIR_PACKED_ONE_CURSOR
		constexpr int diag = 1;
#else
		constexpr int diag = 0;
#endif
		const size_t lds = (size_t)WAVES * cap * 8 + (1 + WAVES) * 8 + 2 * WAVES * 4;
		const bool aligned = ((npx & 7) == 0) && ((((uintptr_t)d_frames) & 15) == 0);
		const int nfast = aligned ? (int)(npx / RIRB1_TILE_PX) : 0; // tiles that lie whole inside the frame
		if (nfast > 0)
			hipLaunchKernelGGL((rirb1_encode_packed<WAVES, true>), dim3(nfast, nchunks), dim3(WAVES * 64), lds, st, d_frames, npx, ntiles, 0, nframes, gop, d_hdr,
							   d_seg_pos, d_seg_words, d_stream, capacity_words, d_ctrl, d_arena, arena_words, cap, diag);
		if (ntiles > nfast)
			hipLaunchKernelGGL((rirb1_encode_packed<WAVES, false>), dim3(ntiles - nfast, nchunks), dim3(WAVES * 64), lds, st, d_frames, npx, ntiles, nfast,
							   nframes, gop, d_hdr, d_seg_pos, d_seg_words, d_stream, capacity_words, d_ctrl, d_arena, arena_words, cap, diag);
		return hipGetLastError();
	}
	hipError_t launch_decode_packed(const uint64_t *d_hdr, const uint64_t *d_seg_pos, const uint32_t *d_seg_words, const uint64_t *d_stream,
									uint64_t stream_words, int64_t npx, int ntiles, int nframes, int gop, uint16_t *d_frames, int *d_error, hipStream_t st)
	{
		const int nchunks = (nframes + gop - 1) / gop;
		dim3 grid((ntiles + 3) / 4, nchunks), block(256);
		hipLaunchKernelGGL(rirb1_decode_tiles, grid, block, 0, st, d_hdr, (const uint32_t *)nullptr, (const uint64_t *)nullptr, d_stream, stream_words, npx,
						   ntiles, nframes, gop, (const int64_t *)nullptr, d_seg_words, d_seg_pos, d_frames, d_error);
		return hipGetLastError();
	}
} // namespace rir
