// Launch floor of one block's kernel chain, eager against hipGraph replay.
// The chain is the shape of process_block at nh = 7, nb = 5: per round commit_check, detect, file, finish, reorder with the
// product's grid / workgroup / LDS sizes, bodies empty (one predicated store so that nothing is optimised away), 5 rounds +
// block_init + rest_append = 27 launches, 67 blocks.  Built on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/graph_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_body(int *flag, int tag)
{
	extern __shared__ int s_any[];
	if (flag[0] == tag) { s_any[threadIdx.x] = tag; __syncthreads(); flag[1 + blockIdx.x % 7] = s_any[(threadIdx.x + 1) % blockDim.x]; }
}

struct Shape { int wgs, threads, lds; const char *name; };

static int chain(hipStream_t st, int *flag, const std::vector<Shape> &sh)
{
	for (size_t i = 0; i < sh.size(); i++)
		hipLaunchKernelGGL(k_body, dim3(sh[i].wgs), dim3(sh[i].threads), sh[i].lds, st, flag, -1 - (int)i);
	return 0;
}

static double run(const char *what, hipStream_t st, int *flag, const std::vector<Shape> &sh, int blocks, bool graph)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
	if (graph) {
		hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
		chain(st, flag, sh);
		hipStreamEndCapture(st, &g);
		hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
	}
	double best = 1e30, host_best = 1e30;
	for (int rep = 0; rep < 5; rep++) {
		hipStreamSynchronize(st);
		auto h0 = std::chrono::steady_clock::now();
		hipEventRecord(e0, st);
		for (int b = 0; b < blocks; b++) { if (graph) hipGraphLaunch(ge, st); else chain(st, flag, sh); }
		hipEventRecord(e1, st);
		auto h1 = std::chrono::steady_clock::now();
		hipStreamSynchronize(st);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
		double hm = std::chrono::duration<double, std::milli>(h1 - h0).count();
		if (hm < host_best) host_best = hm;
	}
	const double n = (double)blocks * sh.size();
	printf("%-44s %s: %8.3f ms device for %5.0f launches = %6.2f us per launch; host enqueue %8.3f ms = %5.2f us per launch\n",
	       what, graph ? "graph" : "eager", best, n, best * 1e3 / n, host_best, host_best * 1e3 / n);
	if (ge) hipGraphExecDestroy(ge);
	if (g) hipGraphDestroy(g);
	hipEventDestroy(e0); hipEventDestroy(e1);
	return best;
}

int main()
{
	int *flag;
	CK(hipMalloc(&flag, 64));
	CK(hipMemset(flag, 0, 64));
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	CK(hipFuncSetAttribute((const void *)k_body, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
	const int blocks = 67;
	auto block_of = [](std::vector<Shape> round) {
		std::vector<Shape> b;
		b.push_back({1024 * 5, 256, 0, "block_init"});
		for (int t = 0; t < 5; t++) for (auto &s : round) { Shape x = s; if (x.name[0] == 'c' && x.wgs > 1000) x.wgs = 5 * ((1024 >> t) + (1024 >> ((t + 4) % 5))); b.push_back(x); }
		b.push_back({1024 * 5, 256, 0, "rest_append"});
		return b;
	};
	std::vector<Shape> product = {{5440, 256, 17 * 1024, "commit_check"}, {1280, 1024, 64 * 1024, "detect"}, {320, 1024, 8 * 1024, "file"}, {133, 1024, 139 * 1024, "finish"}, {5 * 260, 256, 1024, "reorder"}};
	std::vector<Shape> small = {{256, 256, 0, "commit_check"}, {256, 256, 0, "detect"}, {256, 256, 0, "file"}, {256, 256, 0, "finish"}, {256, 256, 0, "reorder"}};
	std::vector<Shape> tiny = {{8, 64, 0, "commit_check"}, {8, 64, 0, "detect"}, {8, 64, 0, "file"}, {8, 64, 0, "finish"}, {8, 64, 0, "reorder"}};
	for (int g = 0; g < 2; g++) {
		run("product grids (empty bodies)", st, flag, block_of(product), blocks, g);
		run("256 x 256-thread workgroups per launch", st, flag, block_of(small), blocks, g);
		run("8 x 64-thread workgroups per launch", st, flag, block_of(tiny), blocks, g);
	}
	// one launch class at a time: where does the floor of the product grids come from?
	for (auto &s : product) { std::vector<Shape> one(27, s); run(s.name, st, flag, one, blocks, false); }
	return 0;
}
