// strpack.h -- host side of the vector<string> front door (kmodel.hpp:90-98): k-mer strings -> what the query kernels take.
#pragma once
#include <cstdint>

// Strings [lo, hi) of a batch: string i is ptrs[i] (separate strings) or flat + i * stride (one buffer), `len` characters each.
struct KmxStrBatch {
	const char *const *ptrs;     // null: use flat / stride
	const char *flat;
	int stride, len;
};

// 2 bits per base (A C G T = 0 1 2 3), first base most significant, W = ceil(len / 32) words per k-mer, word 0 = the
// first len - 32 bases when W == 2 -- the layout of tools.hpp:63-76 that k_query reads.  dst[(i - lo) * W ...].
// A string that holds anything but ACGT cannot be expressed this way (its word is then meaningless): its index is appended
// to `dirty` (when given) -- the caller ships those strings as bytes (k_query_ascii).  Returns false if there was one.
#include <vector>
bool kmx_pack_strings(const KmxStrBatch &b, int W, uint64_t lo, uint64_t hi, uint64_t *dst, std::vector<uint64_t> *dirty = nullptr);

// the same strings laid out back to back, `len` bytes each: dst[(i - lo) * len ...]
void kmx_gather_strings(const KmxStrBatch &b, uint64_t lo, uint64_t hi, unsigned char *dst);

// CPUs this process may use: the smaller of its affinity mask and its cgroup CPU quota (a container sees every CPU of the
// host in the mask and is throttled to its quota: more runnable threads than that only queue), at least 1
int kmx_host_cpus(void);
