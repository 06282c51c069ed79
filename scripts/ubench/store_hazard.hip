// A buffer store of 16 bytes followed AT ONCE by a vector instruction that writes the first of its data registers: does the store send what
// the registers held when it was issued?  Round 6 met a case where it did not (lossy_kernels.hip: buf_stn) - `buffer_store_dwordx4 ... sN offen nt`
// with the scalar offset in a register, the form for which the ISA's wait-state table asks nothing and the compiler inserts nothing.  This
// program issues the two instructions back to back from inline assembly, in every wave of a full chip, behind a burst of other stores of the
// same wave (a busy memory pipe), and counts the 16-byte records whose first word arrived overwritten.  Variants: scalar offset in a register /
// the constant 0; nt / default policy; 0-4 idle cycles (s_nop) between the store and the write.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_hazard scripts/ubench/store_hazard.hip && /tmp/store_hazard
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

#define BODY(STORE_TEXT, NOPS_TEXT)                                                                                                     \
	asm volatile("v_mov_b32 v10, %[good]\n"                                                                                             \
				 "v_mov_b32 v11, %[good]\n"                                                                                             \
				 "v_mov_b32 v12, %[good]\n"                                                                                             \
				 "v_mov_b32 v13, %[good]\n"                                                                                             \
				 "s_nop 7\n" STORE_TEXT "\n" NOPS_TEXT "v_mov_b32 v10, %[bad]\n"                                                         \
				 "s_nop 7\n"                                                                                                            \
				 :                                                                                                                      \
				 : [good] "v"(good), [bad] "v"(bad), [off] "v"(off), [rs] "s"(rs), [so] "s"(so)                                          \
				 : "v10", "v11", "v12", "v13", "memory")

template <int VARIANT, int NOPS>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t *scratch, int rounds, int burst)
{
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	const uint32_t nthreads = gridDim.x * 256;
	// (the descriptor of `out` as four words: base, base high | stride 0, num_records, raw buffer with 32-bit data format)
	const uint64_t a = (uint64_t)out;
	const v4i rs = {(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
	const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(scratch, 0, (int)0x7fffffff, 0x00020000);
	for (int r = 0; r < rounds; ++r)
	{
		// a burst of other 16-byte stores of this wave first: the store under test is issued into a busy pipe
		for (int b = 0; b < burst; ++b)
		{
			v4i x = {(int)tid, r, b, 7};
			__builtin_amdgcn_raw_buffer_store_b128(x, rsb, (tid + (uint32_t)b * nthreads) * 16u, 0, 2);
		}
		const uint32_t good = 0x600d0000u | (uint32_t)r, bad = 0xbad00000u | (uint32_t)r;
		const uint32_t off = tid * 16u;
		const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)r * nthreads * 16u); // (a scalar register)
		(void)so;
		if constexpr (VARIANT == 0) // register offset, nt: the form met in lossy_kernels.hip
		{
			if constexpr (NOPS == 0) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], %[so] offen nt", "");
			if constexpr (NOPS == 1) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], %[so] offen nt", "s_nop 0\n");
			if constexpr (NOPS == 2) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], %[so] offen nt", "s_nop 1\n");
			if constexpr (NOPS == 4) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], %[so] offen nt", "s_nop 3\n");
		}
		if constexpr (VARIANT == 1) // register offset, default policy
		{
			if constexpr (NOPS == 0) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], %[so] offen", "");
			if constexpr (NOPS == 2) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], %[so] offen", "s_nop 1\n");
		}
		if constexpr (VARIANT == 2) // the constant 0 as scalar offset (the form the table DOES ask wait states for), nt; the records of a round lie behind each other through the vector offset
		{
			const uint32_t off2 = off + (uint32_t)r * nthreads * 16u;
			const uint32_t off_keep = off;
			(void)off_keep;
			{
				const uint32_t off = off2;
				if constexpr (NOPS == 0) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], 0 offen nt", "");
				if constexpr (NOPS == 2) BODY("buffer_store_dwordx4 v[10:13], %[off], %[rs], 0 offen nt", "s_nop 1\n");
			}
		}
		if constexpr (VARIANT == 3) // 8 bytes, register offset, nt: the form every table holds for safe
		{
			if constexpr (NOPS == 0) BODY("buffer_store_dwordx2 v[10:11], %[off], %[rs], %[so] offen nt", "");
		}
	}
}

template <int VARIANT, int NOPS>
static void run(const char *what, int rounds, int burst)
{
	const int blocks = 256 * 8, nthreads = blocks * 256;
	uint32_t *out, *scratch;
	const size_t bytes = (size_t)rounds * nthreads * 16;
	hipMalloc(&out, bytes);
	hipMalloc(&scratch, (size_t)burst * nthreads * 16 + 16);
	hipMemset(out, 0, bytes);
	hipLaunchKernelGGL((k<VARIANT, NOPS>), dim3(blocks), dim3(256), 0, 0, out, scratch, rounds, burst);
	hipDeviceSynchronize();
	std::vector<uint32_t> h(bytes / 4);
	hipMemcpy(h.data(), out, bytes, hipMemcpyDeviceToHost);
	size_t wrong = 0, other = 0, total = 0;
	const int words = VARIANT == 3 ? 2 : 4;
	for (int r = 0; r < rounds; ++r)
		for (int t = 0; t < nthreads; ++t, ++total)
		{
			const uint32_t *q = &h[((size_t)r * nthreads + t) * 4];
			const uint32_t good = 0x600d0000u | (uint32_t)r, bad = 0xbad00000u | (uint32_t)r;
			if (q[0] == bad)
				++wrong;
			else if (q[0] != good)
				++other;
			for (int w = 1; w < words; ++w)
				if (q[w] != good)
					++other;
		}
	std::printf("%-72s idle cycles %d: %zu of %zu records with the first word overwritten (%.4f %%), %zu other mismatches\n", what, NOPS, wrong, total,
				100.0 * (double)wrong / (double)total, other);
	hipFree(out);
	hipFree(scratch);
}

int main()
{
	const int rounds = 24, burst = 12;
	run<0, 0>("16 bytes, scalar offset in a register, nt", rounds, burst);
	run<0, 1>("16 bytes, scalar offset in a register, nt", rounds, burst);
	run<0, 2>("16 bytes, scalar offset in a register, nt", rounds, burst);
	run<0, 4>("16 bytes, scalar offset in a register, nt", rounds, burst);
	run<1, 0>("16 bytes, scalar offset in a register, default policy", rounds, burst);
	run<1, 2>("16 bytes, scalar offset in a register, default policy", rounds, burst);
	run<2, 0>("16 bytes, scalar offset the constant 0, nt (hand-written: no wait states)", rounds, burst);
	run<2, 2>("16 bytes, scalar offset the constant 0, nt", rounds, burst);
	run<3, 0>("8 bytes, scalar offset in a register, nt", rounds, burst);
	return 0;
}
