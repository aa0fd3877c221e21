// rest_device.hip -- KRestData::build on the device (rest.hpp:95-135, :157-161).
//
// The reference collects the survivors in per-prefix vectors, std::sorts every vector by suffix bytes and
// flattens them.  Sorting the whole list by packed k-mer value gives the same arrays (the prefix is the top
// 2*pre_len bits), so here: one device radix sort (rocPRIM, the vendor primitive for a plain key sort -- not a
// hot-path kernel) + three small kernels that emit hash2index / pre_buffer / the suffix integers k_query
// searches.  The on-disk byte arrays are materialised on the host only when save() is called.
#include "kmx_types.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

__global__ __launch_bounds__(256) void k_iota(u32 *idx, u64 n)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i < n) idx[i] = (u32)i;
}
// split [n][2] k-mers into hi / lo planes
__global__ __launch_bounds__(256) void k_split2(const u64 *km, u64 n, u64 *hi, u64 *lo)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { hi[i] = km[2 * i]; lo[i] = km[2 * i + 1]; }
}
__global__ __launch_bounds__(256) void k_gather_u64(const u64 *src, const u32 *idx, u64 n, u64 *dst)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void k_gather_final2(const u64 *km, const int *cnt, const u32 *idx, u64 n, u64 *okm, int *ocnt)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	u32 j = idx[i];
	okm[2 * i] = km[2 * (u64)j];
	okm[2 * i + 1] = km[2 * (u64)j + 1];
	ocnt[i] = cnt[j];
}

__device__ __forceinline__ u32 prefix_of(const u64 *km, u64 e, int W, int sbits)
{
	if (W == 1) return (u32)(km[e] >> sbits);
	const u64 hi = km[2 * e], lo = km[2 * e + 1];
	if (sbits >= 64) return (u32)(hi >> (sbits - 64));
	return (u32)((hi << (64 - sbits)) | (lo >> sbits));
}

// flag[e] = 1 where a new prefix group starts; suffix integers for k_query
__global__ __launch_bounds__(256) void k_rest_flags(const u64 *km, u64 n, int W, int sbits, int *flag, u64 *suf)
{
	u64 e = (u64)blockIdx.x * 256 + threadIdx.x;
	if (e >= n) return;
	const u32 p = prefix_of(km, e, W, sbits);
	flag[e] = (e == 0 || prefix_of(km, e - 1, W, sbits) != p) ? 1 : 0;
	if (W == 1) suf[e] = km[e] & ((1ULL << sbits) - 1);
	else {
		const u64 hi = km[2 * e], lo = km[2 * e + 1];
		if (sbits >= 64) { suf[2 * e] = sbits >= 128 ? hi : hi & ((1ULL << (sbits - 64)) - 1); suf[2 * e + 1] = lo; }
		else { suf[2 * e] = 0; suf[2 * e + 1] = lo & ((1ULL << sbits) - 1); }
	}
}
// gid = inclusive_scan(flag) - 1: hash2index[prefix] = gid, pre_buffer[gid] = first row, pre_buffer[groups] = n
__global__ __launch_bounds__(256) void k_rest_index(const u64 *km, u64 n, int W, int sbits, const int *flag, const int *scan, int *h2i, int *pre, int *groups)
{
	u64 e = (u64)blockIdx.x * 256 + threadIdx.x;
	if (e >= n) return;
	const int gid = scan[e] - 1;
	if (flag[e]) { h2i[prefix_of(km, e, W, sbits)] = gid; pre[gid] = (int)e; }
	if (e == n - 1) { pre[gid + 1] = (int)n; *groups = gid + 1; }
}

// rows of a loaded rest.bin carry only suffixes: rebuild the full packed k-mers, prefix by prefix
__global__ __launch_bounds__(256) void k_rest_expand(const int *h2i, const int *pre, const u64 *suf, int map_size, int W, int sbits, u64 *out)
{
	int P = blockIdx.x * 256 + threadIdx.x;
	if (P >= map_size) return;
	const int g = h2i[P];
	if (g < 0) return;
	for (int e = pre[g]; e < pre[g + 1]; e++) {
		if (W == 1) out[e] = ((u64)P << sbits) | suf[e];
		else {
			unsigned __int128 v = ((unsigned __int128)suf[2 * (u64)e] << 64) | suf[2 * (u64)e + 1];
			v |= (unsigned __int128)(u64)P << sbits;
			out[2 * (u64)e] = (u64)(v >> 64);
			out[2 * (u64)e + 1] = (u64)v;
		}
	}
}

// fine[b] = first row whose top F bits are >= b (lower bound over the sorted k-mers); fine[2^F] = n
__global__ __launch_bounds__(256) void k_fine_index(const u64 *km, u64 n, int W, int shift, u64 nbuckets, u32 *fine)
{
	u64 b = (u64)blockIdx.x * 256 + threadIdx.x;
	if (b > nbuckets) return;
	if (b == nbuckets) { fine[b] = (u32)n; return; }
	const unsigned __int128 thr = (unsigned __int128)b << shift;
	u64 lo = 0, hi = n;
	while (lo < hi) {
		const u64 mid = (lo + hi) >> 1;
		const unsigned __int128 v = W == 1 ? (unsigned __int128)km[mid] : (((unsigned __int128)km[2 * mid] << 64) | km[2 * mid + 1]);
		if (v < thr) lo = mid + 1; else hi = mid;
	}
	fine[b] = (u32)lo;
}

// q[P] = suffix of the first row of the group after P's group (all-ones: none)
__global__ __launch_bounds__(256) void k_next_first(const int *h2i, const int *pre, const u64 *suf, u64 n, int map_size, int W, u64 *q)
{
	int P = blockIdx.x * 256 + threadIdx.x;
	if (P >= map_size) return;
	const int g = h2i[P];
	u64 e = g >= 0 ? (u64)pre[g + 1] : n;
	for (int w = 0; w < W; w++) q[(u64)P * W + w] = e < n ? suf[e * W + w] : ~0ULL;
}

inline unsigned nblk(u64 n) { return (unsigned)((n + 255) / 256); }

}   // namespace

// rest.bin's suffix_bin (Appendix B.2): the low suff_group bytes of every sorted k-mer, most significant byte first
__global__ __launch_bounds__(256) void k_rest_suffix_bytes(const u64 *km, u64 n, int W, int suff_group, unsigned char *out)
{
	const u64 e = (u64)blockIdx.x * 256 + threadIdx.x;
	if (e >= n) return;
	const u64 lo = km[e * W + (W - 1)], hi = W == 2 ? km[e * W] : 0;
	for (int g = 0; g < suff_group; g++) {
		const int sh = 8 * (suff_group - 1 - g);
		out[e * (u64)suff_group + g] = (unsigned char)(sh >= 64 ? hi >> (sh - 64) : lo >> sh);
	}
}

namespace kmxk {

#define RCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

// Sort n (k-mer, count) pairs ascending by k-mer.  in/out buffers are distinct; scratch is allocated here
// (rest tables are a few % of the input, and this runs once per build).
hipError_t rest_sort(const u64 *km_in, const int *cnt_in, u64 n, int W, int k, u64 *km_out, int *cnt_out, hipStream_t st)
{
	if (!n) return hipSuccess;
	size_t tmp_bytes = 0;
	void *tmp = nullptr;
	if (W == 1) {
		RCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, km_in, km_out, cnt_in, cnt_out, n, 0, 2 * k, st));
		RCHK(hipMalloc(&tmp, tmp_bytes));
		hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, km_in, km_out, cnt_in, cnt_out, n, 0, 2 * k, st);
		hipStreamSynchronize(st);
		hipFree(tmp);
		return e;
	}
	// two-word keys: LSD -- stable sort by the low word, then by the high word, carrying a permutation
	u64 *hi = nullptr, *lo = nullptr, *key2 = nullptr, *key3 = nullptr;
	u32 *i0 = nullptr, *i1 = nullptr;
	RCHK(hipMalloc((void **)&hi, n * 8)); RCHK(hipMalloc((void **)&lo, n * 8));
	RCHK(hipMalloc((void **)&key2, n * 8)); RCHK(hipMalloc((void **)&key3, n * 8));
	RCHK(hipMalloc((void **)&i0, n * 4)); RCHK(hipMalloc((void **)&i1, n * 4));
	hipLaunchKernelGGL(k_split2, dim3(nblk(n)), dim3(256), 0, st, km_in, n, hi, lo);
	hipLaunchKernelGGL(k_iota, dim3(nblk(n)), dim3(256), 0, st, i0, n);
	size_t t1 = 0, t2 = 0;
	RCHK(rocprim::radix_sort_pairs(nullptr, t1, lo, key2, i0, i1, n, 0, 64, st));
	RCHK(rocprim::radix_sort_pairs(nullptr, t2, key2, key3, i1, i0, n, 0, 2 * k - 64, st));
	tmp_bytes = t1 > t2 ? t1 : t2;
	RCHK(hipMalloc(&tmp, tmp_bytes));
	hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, lo, key2, i0, i1, n, 0, 64, st);            // by low word -> perm i1
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_gather_u64, dim3(nblk(n)), dim3(256), 0, st, (const u64 *)hi, (const u32 *)i1, n, key2);   // high words in that order
		e = rocprim::radix_sort_pairs(tmp, tmp_bytes, key2, key3, i1, i0, n, 0, 2 * k - 64, st);          // stable by high word -> perm i0
	}
	if (e == hipSuccess) hipLaunchKernelGGL(k_gather_final2, dim3(nblk(n)), dim3(256), 0, st, km_in, cnt_in, (const u32 *)i0, n, km_out, cnt_out);
	hipStreamSynchronize(st);
	hipFree(tmp); hipFree(hi); hipFree(lo); hipFree(key2); hipFree(key3); hipFree(i0); hipFree(i1);
	return e;
}

// From the sorted table: h2i[map_size] (pre-set to -1 by the caller), pre[map_size+1], suf[n*W], *groups.
hipError_t rest_index(const u64 *km_sorted, u64 n, int W, int k, int pre_len, int *h2i, int *pre, u64 *suf, int *groups, hipStream_t st)
{
	const int sbits = 2 * (k - pre_len);
	RCHK(hipMemsetAsync(groups, 0, 4, st));
	RCHK(hipMemsetAsync(pre, 0, 4, st));
	if (!n) return hipSuccess;
	int *flag = nullptr, *scan = nullptr;
	void *tmp = nullptr;
	size_t tmp_bytes = 0;
	RCHK(hipMalloc((void **)&flag, n * 4));
	RCHK(hipMalloc((void **)&scan, n * 4));
	hipLaunchKernelGGL(k_rest_flags, dim3(nblk(n)), dim3(256), 0, st, km_sorted, n, W, sbits, flag, suf);
	RCHK(rocprim::inclusive_scan(nullptr, tmp_bytes, flag, scan, n, rocprim::plus<int>(), st));
	RCHK(hipMalloc(&tmp, tmp_bytes));
	hipError_t e = rocprim::inclusive_scan(tmp, tmp_bytes, flag, scan, n, rocprim::plus<int>(), st);
	if (e == hipSuccess) hipLaunchKernelGGL(k_rest_index, dim3(nblk(n)), dim3(256), 0, st, km_sorted, n, W, sbits, (const int *)flag, (const int *)scan, h2i, pre, groups);
	hipStreamSynchronize(st);
	hipFree(tmp); hipFree(flag); hipFree(scan);
	return e;
}

void rest_expand(const int *h2i, const int *pre, const u64 *suf, int map_size, int W, int k, int pre_len, u64 *out, hipStream_t st)
{
	hipLaunchKernelGGL(k_rest_expand, dim3(nblk(map_size)), dim3(256), 0, st, h2i, pre, suf, map_size, W, 2 * (k - pre_len), out);
}

// lookup accelerators for k_query: bucket index over the top F bits and the per-prefix "next group's first row"
void rest_accel(const u64 *km_sorted, u64 n, int W, int k, int F, const int *h2i, const int *pre, const u64 *suf, int map_size, u32 *fine, u64 *q, hipStream_t st)
{
	const u64 nbuckets = 1ULL << F;
	hipLaunchKernelGGL(k_fine_index, dim3(nblk(nbuckets + 1)), dim3(256), 0, st, km_sorted, n, W, 2 * k - F, nbuckets, fine);
	hipLaunchKernelGGL(k_next_first, dim3(nblk(map_size)), dim3(256), 0, st, h2i, pre, suf, n, map_size, W, q);
}

void rest_suffix_bytes(const u64 *km_sorted, u64 n, int W, int suff_group, unsigned char *out, hipStream_t st)
{
	if (n) hipLaunchKernelGGL(k_rest_suffix_bytes, dim3(nblk(n)), dim3(256), 0, st, km_sorted, n, W, suff_group, out);
}

}   // namespace kmxk
