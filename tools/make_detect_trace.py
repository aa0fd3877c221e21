"""Debug aid: build kmcex_amd/libkmx_dtrace.so, a copy of the library whose k_round_detect stamps wall_clock64 (10 ns ticks)
at its phase boundaries for a few (bin, list) workgroups and prints them for the first launches of a build.  Not part of the
product; nothing is committed from it.  usage: python tools/make_detect_trace.py ; KMX_TEST_HOOKS=1 KMX_LIBRARY=.../libkmx_dtrace.so python tools/soak_big.py 1e8 1"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "kmcex_amd/csrc/kernels.hip")).read()
def rep(old, new, count=1):
    global src
    assert old in src, old
    src = src.replace(old, new, count)
rep("template <int NHM, int BT> __global__ __launch_bounds__(BT) __attribute__((amdgpu_waves_per_eu(NHM <= 8 ? 8 : 4)))\nvoid k_round_detect(",
    "__device__ long long g_dt[8][16]; __device__ unsigned long long g_dt0; __device__ int g_dlaunch;\n"
    "#define DTR(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x & 31) == 0 && blockIdx.x < 256) g_dt[blockIdx.x >> 5][k] = (long long)wall_clock64(); } while (0)\n"
    "template <int NHM, int BT> __global__ __launch_bounds__(BT) __attribute__((amdgpu_waves_per_eu(NHM <= 8 ? 8 : 4)))\nvoid k_round_detect(")
rep("	const int i = (int)blockIdx.y, b = blockIdx.x;\n	const int id = (i + 1) % nb;", "	DTR(0);\n	const int i = (int)blockIdx.y, b = blockIdx.x;\n	const int id = (i + 1) % nb;")
rep("	if (!use_delta) dcnt = 0;\n	if (cnt) {", "	if (!use_delta) dcnt = 0;\n	DTR(1);\n	if (cnt) {")
rep("			__syncthreads();\n#pragma unroll\n			for (int u = 0; u < U; u++) if (e[u] != ~0ULL) dt_insert(", "			__syncthreads();\n			DTR(2);\n			{ volatile u64 sink_ = e[0] ^ d[0]; (void)sink_; }\n			DTR(3);\n#pragma unroll\n			for (int u = 0; u < U; u++) if (e[u] != ~0ULL) dt_insert(")
rep("			__syncthreads();\n			if (dcnt) {                                              // (uniform)", "			__syncthreads();\n			DTR(4);\n			if (dcnt) {                                              // (uniform)")
rep("#pragma unroll\n			for (int u = 0; u < U; u++) if (e[u] != ~0ULL) dt_lookup(s_t, tmask, e[u], status, dfail);", "			DTR(5);\n#pragma unroll\n			for (int u = 0; u < U; u++) if (e[u] != ~0ULL) dt_lookup(s_t, tmask, e[u], status, dfail);\n			DTR(6);")
rep("	if (threadIdx.x == 0) {\n		*gd = 0;", "	__syncthreads();\n	DTR(7);\n	if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { g_dt[0][8] = cnt; g_dt[0][9] = dcnt; }\n	if (threadIdx.x == 0) {\n		*gd = 0;")
# a tiny kernel after detect prints the stamps (same stream: ordered)
rep("// ------------------------------------------------------------------------------------------ S: ordered slow path (helpers)",
    "__global__ void k_dtrace_print(int t) { if (g_dlaunch++ >= 40) return; long long t0 = g_dt[0][0]; for (int w = 0; w < 8; w++) if (g_dt[w][0] < t0) t0 = g_dt[w][0];\n"
    "	printf(\"DET t=%d cnt=%lld dcnt=%lld:\", t, g_dt[0][8], g_dt[0][9]); for (int w = 0; w < 8; w += 1) { printf(\" [wg%d\", w * 32); for (int k = 0; k < 8; k++) printf(\" %.1f\", (double)(g_dt[w][k] - t0) / 100.0); printf(\"]\"); } printf(\"\\n\"); }\n"
    "// ------------------------------------------------------------------------------------------ S: ordered slow path (helpers)")
rep("	KPROF_END(prof, st);\n	KPROF_BEGIN(prof, KC_FILE, st);", "	hipLaunchKernelGGL(k_dtrace_print, dim3(1), dim3(1), 0, st, t);\n	KPROF_END(prof, st);\n	KPROF_BEGIN(prof, KC_FILE, st);")
d = os.path.join(root, "kmcex_amd/csrc")
open(os.path.join(d, "kernels_dtrace.hip"), "w").write(src)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-c", os.path.join(d, "kernels_dtrace.hip"), "-o", os.path.join(d, "kernels_dtrace.o")])
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(root, "kmcex_amd/libkmx_dtrace.so"), os.path.join(d, "kernels_dtrace.o")] + [os.path.join(d, f) for f in ("rest_device.o", "kmx_api.o", "kmc_reader.o")])
os.remove(os.path.join(d, "kernels_dtrace.hip")); os.remove(os.path.join(d, "kernels_dtrace.o"))
print("built kmcex_amd/libkmx_dtrace.so")
