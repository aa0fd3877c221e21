"""Debug aid: build kmcex_amd/libkmx_trace.so, a copy of the library whose finisher prints a phase timeline
(wall_clock64, 10 ns ticks) for list 0 of the first launches.  Not part of the product; nothing is committed from it."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "kmcex_amd/csrc/kernels.hip")).read()
def rep(old, new, count=1):
    global src
    assert old in src, old
    src = src.replace(old, new, count)
rep("template <int W, int NHM, int RPT>\n__device__ __forceinline__ u64 finish_lds(",
    "__device__ long long g_tr[96]; __device__ int g_trn; __device__ int g_prints;\n"
    "#define TR(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { int q_ = g_trn++; if (q_ < 96) g_tr[q_] = ((long long)(k) << 48) | (wall_clock64() & 0xFFFFFFFFFFFFll); } } while (0)\n"
    "template <int W, int NHM, int RPT>\n__device__ __forceinline__ u64 finish_lds(")
rep("#define FIN_BIT(r, j) ((u32)(bits[r] >> (4 * (j))) & 15u)", "TR(3);\n#define FIN_BIT(r, j) ((u32)(bits[r] >> (4 * (j))) & 15u)")
rep("	int succ = 0;\n	u64 iters = 0;\n	for (;; iters++) {\n		const int par = (int)(iters % 3), par_next", "	TR(7);\n	int succ = 0;\n	u64 iters = 0;\n	for (;; iters++) {\n		const int par = (int)(iters % 3), par_next")
rep("		if (threadIdx.x == 0) s_pending[par_next] = 0;               // last read two iterations ago\n		lds_barrier();",
    "		if (threadIdx.x == 0) s_pending[par_next] = 0;               // last read two iterations ago\n		lds_barrier();\nTR(4);")
rep("		lds_barrier();\n		if (!s_pending[par]) break;", "		lds_barrier();\nTR(5);\n		if (!s_pending[par]) break;")
rep("	constexpr int CAP = RPT * 1024;\n	const u64 row = (u64)i * KMX_BUCKET;\n",
    "	constexpr int CAP = RPT * 1024;\n	const u64 row = (u64)i * KMX_BUCKET;\n	if (threadIdx.x == 0 && blockIdx.x == 0) g_trn = 0;\n	TR(1);\n")
rep("				if (cnt <= CAP) break;\n", "				TR(6);\n				if (cnt <= CAP) break;\n")
rep("	__syncthreads();\n	if (threadIdx.x == 0) {\n		atomicAdd(bd.stats + ST_FIN_ITERS, iters);",
    "	__syncthreads();\n	TR(9);\n	if (threadIdx.x == 0 && blockIdx.x == 0 && g_prints < 60) { g_prints++; printf(\"FIN n=%d snap=%d:\", n, (int)snapshot); for (int q = 1; q < g_trn && q < 96; q++) printf(\" %d:%.1f\", (int)(g_tr[q] >> 48), (double)((g_tr[q] & 0xFFFFFFFFFFFFll) - (g_tr[0] & 0xFFFFFFFFFFFFll)) / 100.0); printf(\"\\n\"); }\n	if (threadIdx.x == 0) {\n		atomicAdd(bd.stats + ST_FIN_ITERS, iters);")
d = os.path.join(root, "kmcex_amd/csrc")
open(os.path.join(d, "kernels_trace.hip"), "w").write(src)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-I" + os.path.join(root, "include"), "-c", os.path.join(d, "kernels_trace.hip"), "-o", os.path.join(d, "kernels_trace.o")])
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "kmcex_amd/libkmx_trace.so"), os.path.join(d, "kernels_trace.o")] + [os.path.join(d, f) for f in ("rest_device.o", "kmx_api.o", "kmc_reader.o")])
os.remove(os.path.join(d, "kernels_trace.hip")); os.remove(os.path.join(d, "kernels_trace.o"))
print("built kmcex_amd/libkmx_trace.so")
