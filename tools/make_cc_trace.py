"""Debug aid: libkmx_trace.so whose k_round_check_claim / k_round_verify_commit print a phase timeline (wall_clock64,
10 ns ticks) of workgroup (0,0) for late rounds.  Not part of the product."""
import os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "kmcex_amd/csrc/kernels.hip")).read()
def rep(old, new, count=1):
    global src
    assert old in src, old
    src = src.replace(old, new, count)
rep("template <int W, int NHM> __global__ __launch_bounds__(256) void k_round_check_claim(",
    "__device__ int g_ccp;\n#define CT(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && t >= 3) { ct_[ctn_] = wall_clock64(); ck_[ctn_++] = k; } } while (0)\n"
    "template <int W, int NHM> __global__ __launch_bounds__(256) void k_round_check_claim(")
rep("	__shared__ int s_fail;\n	const int i = blockIdx.y;\n	const int n = bd.n[pp][i];",
    "	__shared__ int s_fail;\n	long long ct_[12]; int ck_[12]; int ctn_ = 0;\n	CT(0);\n	const int i = blockIdx.y;\n	const int n = bd.n[pp][i];\n	CT(1);")
rep("			load_kmer<W>(bd.kmers, row + idx, v);\n			const u32 bin = md.bin_of_occ[bd.counts[row + idx]];\n			Premixed<W> pm = premix_string<W>(left_align<W>(v, md.k), md.gfull);\n			Touches<NHM> tc;\n			gather_touches<W, NHM, false>(md, pm, a, tc);\n			failed = touches_conflict<NHM>(md, tc, bin);",
    "			load_kmer<W>(bd.kmers, row + idx, v);\n			const u32 bin = md.bin_of_occ[bd.counts[row + idx]];\n			CT(2);\n			Premixed<W> pm = premix_string<W>(left_align<W>(v, md.k), md.gfull);\n			Touches<NHM> tc;\n			gather_touches<W, NHM, false>(md, pm, a, tc);\n			failed = touches_conflict<NHM>(md, tc, bin);\n			CT(3);")
rep("		const u64 mask = __ballot(failed);\n		if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&s_fail, (int)__popcll(mask));\n		__syncthreads();",
    "		CT(4);\n		const u64 mask = __ballot(failed);\n		if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&s_fail, (int)__popcll(mask));\n		__syncthreads();\n		CT(5);")
rep("		if (threadIdx.x == 0 && s_fail) atomicAdd(bd.tile_cnt[pp] + i * KMX_NTILES + (base >> 10), s_fail);\n		__syncthreads();\n	}\n}",
    "		if (threadIdx.x == 0 && s_fail) atomicAdd(bd.tile_cnt[pp] + i * KMX_NTILES + (base >> 10), s_fail);\n		__syncthreads();\n		CT(6);\n	}\n"
    "	if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && t >= 3 && g_ccp < 40) { g_ccp++; printf(\"CC t=%d n=%d grid=%d:\", t, n, (int)gridDim.x); for (int q = 1; q < ctn_; q++) printf(\" %d:%.1f\", ck_[q], (double)(ct_[q] - ct_[0]) / 100.0); printf(\"\\n\"); }\n}")
d = os.path.join(root, "kmcex_amd/csrc")
open(os.path.join(d, "kernels_trace.hip"), "w").write(src)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-I" + os.path.join(root, "include"), "-c", os.path.join(d, "kernels_trace.hip"), "-o", os.path.join(d, "kernels_trace.o")])
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "kmcex_amd/libkmx_trace.so"), os.path.join(d, "kernels_trace.o")] + [os.path.join(d, f) for f in ("rest_device.o", "kmx_api.o", "kmc_reader.o")])
os.remove(os.path.join(d, "kernels_trace.hip")); os.remove(os.path.join(d, "kernels_trace.o"))
print("built kmcex_amd/libkmx_trace.so")
