// Micro-benchmark: issue rate of the VALU/cross-lane instructions used by the codec, 8 waves/SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2000
#define UNROLL 16
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
	uint32_t a0 = threadIdx.x * 7 + seed, a1 = a0 ^ 0x1234567, a2 = a0 * 3, a3 = a0 + 99, m = 0xf0f0f0f0u ^ threadIdx.x, sh = threadIdx.x & 31;
	for (int i = 0; i < N_ITER; ++i)
	{
#pragma unroll
		for (int u = 0; u < UNROLL / 4; ++u)
		{
			if (OP == 0) { a0 = a0 * 3 + a1; a1 = a1 * 5 + a2; a2 = a2 * 7 + a3; a3 = a3 * 9 + a0; } // v_mad_u32_u24? generic int
			if (OP == 1) { a0 = __builtin_amdgcn_bitop3_b32(m, a1, a0, 0xCA); a1 = __builtin_amdgcn_bitop3_b32(m, a2, a1, 0xCA); a2 = __builtin_amdgcn_bitop3_b32(m, a3, a2, 0xCA); a3 = __builtin_amdgcn_bitop3_b32(m, a0, a3, 0xCA); }
			if (OP == 2) { a0 = __builtin_amdgcn_alignbit(a0, a0, sh); a1 = __builtin_amdgcn_alignbit(a1, a1, sh); a2 = __builtin_amdgcn_alignbit(a2, a2, sh); a3 = __builtin_amdgcn_alignbit(a3, a3, sh); }
			if (OP == 3) { a0 = __builtin_amdgcn_mov_dpp(a0, 0x4e, 0xf, 0xf, false); a1 = __builtin_amdgcn_mov_dpp(a1, 0x4e, 0xf, 0xf, false); a2 = __builtin_amdgcn_mov_dpp(a2, 0x4e, 0xf, 0xf, false); a3 = __builtin_amdgcn_mov_dpp(a3, 0x4e, 0xf, 0xf, false); }
			if (OP == 4) { auto r = __builtin_amdgcn_permlane32_swap(a0, a1, false, false); a0 = r[0]; a1 = r[1]; auto q = __builtin_amdgcn_permlane32_swap(a2, a3, false, false); a2 = q[0]; a3 = q[1]; }
			if (OP == 5) { a0 = __builtin_amdgcn_ds_swizzle(a0, (16 << 10) | 0x1f); a1 = __builtin_amdgcn_ds_swizzle(a1, (16 << 10) | 0x1f); a2 = __builtin_amdgcn_ds_swizzle(a2, (4 << 10) | 0x1f); a3 = __builtin_amdgcn_ds_swizzle(a3, (4 << 10) | 0x1f); }
			if (OP == 6) { a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a0; } // v_add_u32
			if (OP == 7) { a0 = a0 ^ a1; a1 = a1 ^ a2; a2 = a2 ^ a3; a3 = a3 ^ a0; }
			if (OP == 8) { typedef unsigned short us2 __attribute__((ext_vector_type(2))); a0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a0) - __builtin_bit_cast(us2, a1)); a1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a1) - __builtin_bit_cast(us2, a2)); a2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a2) - __builtin_bit_cast(us2, a3)); a3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a3) - __builtin_bit_cast(us2, a0)); }
			if (OP == 10) { a0 = __builtin_amdgcn_alignbit(a0, a0, 8); a1 = __builtin_amdgcn_alignbit(a1, a1, 8); a2 = __builtin_amdgcn_alignbit(a2, a2, 8); a3 = __builtin_amdgcn_alignbit(a3, a3, 8); }
			if (OP == 11) { a0 = __builtin_amdgcn_perm(a0, a1, m); a1 = __builtin_amdgcn_perm(a1, a2, m); a2 = __builtin_amdgcn_perm(a2, a3, m); a3 = __builtin_amdgcn_perm(a3, a0, m); }
			if (OP == 12) { a0 = (a0 << 3) | a1; a1 = (a1 << 3) | a2; a2 = (a2 << 3) | a3; a3 = (a3 << 3) | a0; }
			if (OP == 13) { a0 = (a0 & m) | a1; a1 = (a1 & m) | a2; a2 = (a2 & m) | a3; a3 = (a3 & m) | a0; }
			if (OP == 14) { a0 = a0 > m ? a1 : a0; a1 = a1 > m ? a2 : a1; a2 = a2 > m ? a3 : a2; a3 = a3 > m ? a0 : a3; }
			if (OP == 15) { a0 = a0 << sh; a1 = a1 >> sh; a2 = a2 << sh; a3 = a3 >> sh; a0 += 3; a1 += 5; a2 += 7; a3 += 9; }
			if (OP == 16) { a0 = max(a0, a1); a1 = max(a1, a2); a2 = max(a2, a3); a3 = max(a3, a0 + 1); }
			if (OP == 17) { a0 = __builtin_amdgcn_update_dpp(a0, a1, 0x4e, 0xf, 0xf, false) + 1; a1 = __builtin_amdgcn_update_dpp(a1, a2, 0x4e, 0xf, 0xf, false) + 1; a2 = __builtin_amdgcn_update_dpp(a2, a3, 0x4e, 0xf, 0xf, false) + 1; a3 = __builtin_amdgcn_update_dpp(a3, a0, 0x4e, 0xf, 0xf, false) + 1; }
			if (OP == 18) { a0 = min((int)a0, __builtin_amdgcn_update_dpp(0x7fffffff, (int)a0, 0x111, 0xf, 0xf, false)); a1 = min((int)a1, __builtin_amdgcn_update_dpp(0x7fffffff, (int)a1, 0x112, 0xf, 0xf, false)); a2 = min((int)a2, __builtin_amdgcn_update_dpp(0x7fffffff, (int)a2, 0x114, 0xf, 0xf, false)); a3 = min((int)a3, __builtin_amdgcn_update_dpp(0x7fffffff, (int)a3, 0x142, 0xa, 0xf, false)); }
			if (OP == 19) { a0 = __builtin_popcount(a0 & m) + a1; a1 = __builtin_popcount(a1 & m) + a2; a2 = __builtin_popcount(a2 & m) + a3; a3 = __builtin_popcount(a3 & m) + a0; }
			if (OP == 20) { a0 = __builtin_amdgcn_ubfe(a1, sh, 8) + a0; a1 = __builtin_amdgcn_ubfe(a2, sh, 8) + a1; a2 = __builtin_amdgcn_ubfe(a3, sh, 8) + a2; a3 = __builtin_amdgcn_ubfe(a0, sh, 8) + a3; }
			if (OP == 21) { a0 = min((int)a0, (int)a1); a1 = min((int)a1, (int)a2); a2 = min((int)a2, (int)a3); a3 = min((int)a3, (int)a0 + 1); }
			if (OP == 22) { a0 = min((int)a0, __builtin_amdgcn_ds_swizzle((int)a0, (1 << 10) | 0x1f)); a1 = min((int)a1, __builtin_amdgcn_ds_swizzle((int)a1, (2 << 10) | 0x1f)); a2 = min((int)a2, __builtin_amdgcn_ds_swizzle((int)a2, (4 << 10) | 0x1f)); a3 = min((int)a3, __builtin_amdgcn_ds_swizzle((int)a3, (8 << 10) | 0x1f)); }
			if (OP == 23) { typedef short s2 __attribute__((ext_vector_type(2))); a0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s2, a0), __builtin_bit_cast(s2, a1))); a1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s2, a1), __builtin_bit_cast(s2, a2))); a2 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s2, a2), __builtin_bit_cast(s2, a3))); a3 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s2, a3), __builtin_bit_cast(s2, a0 + 1))); }
			if (OP == 9) { a0 = max(a0, (uint32_t)__builtin_amdgcn_mov_dpp(a0, 0x121, 0xf, 0xf, false)); a1 = max(a1, (uint32_t)__builtin_amdgcn_mov_dpp(a1, 0x121, 0xf, 0xf, false)); a2 = max(a2, (uint32_t)__builtin_amdgcn_mov_dpp(a2, 0x121, 0xf, 0xf, false)); a3 = max(a3, (uint32_t)__builtin_amdgcn_mov_dpp(a3, 0x121, 0xf, 0xf, false)); }
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int OP>
void run(const char *name, uint32_t *d)
{
	const int blocks = 256 * 8; // 8 blocks of 256 threads per CU = 8 waves per SIMD
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	// instructions per SIMD = waves_per_simd(8) * N_ITER * UNROLL
	double instr_per_simd = 8.0 * N_ITER * UNROLL;
	double cycles = ms * 1e-3 * 2.4e9;
	printf("%-28s %8.3f ms  ~%.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cycles / instr_per_simd);
}
int main()
{
	uint32_t *d;
	hipMalloc(&d, 256 * 8 * 256 * 4);
	run<6>("v_add_u32", d);
	run<7>("v_xor_b32", d);
	run<0>("mul-add (int)", d);
	run<1>("v_bitop3_b32", d);
	run<2>("v_alignbit_b32", d);
	run<3>("v_mov_b32_dpp", d);
	run<9>("v_max_u32 + dpp", d);
	run<4>("v_permlane32_swap (x2 regs)", d);
	run<5>("ds_swizzle_b32", d);
	run<8>("v_pk_sub_u16", d);
	run<10>("v_alignbit const", d);
	run<11>("v_perm_b32", d);
	run<12>("v_lshl_or_b32", d);
	run<13>("v_and_or_b32", d);
	run<14>("v_cmp+v_cndmask (2)", d);
	run<15>("shift var + add (2)", d);
	run<16>("v_max_u32 (+1 add in 4th)", d);
	run<17>("mov_dpp + add (2)", d);
	run<18>("v_min_i32_dpp (fused)", d);
	run<19>("v_and + v_bcnt (2)", d);
	run<20>("v_bfe + add (2)", d);
	run<21>("v_min_i32 (+1 add in 4th)", d);
	run<22>("ds_swizzle + v_min (1+lds)", d);
	run<23>("v_pk_min_i16 (+1 add)", d);
	return 0;
}
