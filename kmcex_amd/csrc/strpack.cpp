// strpack.cpp -- host side of the vector<string> front door (kmodel.hpp:90-98).  The reference walks the vector with t_num
// threads and hashes every string as it is; here worker threads turn the strings into packed k-mers (8 bytes instead of k
// over the link) 16 characters per SSSE3 step, and the kernels do the rest.  Plain C++ (no HIP): compiled for the host only.
#include "strpack.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <sched.h>

namespace {

struct Lut { unsigned char c[256]; Lut() { memset(c, 0x80, sizeof c); c['A'] = 0; c['C'] = 1; c['G'] = 2; c['T'] = 3; } };
const Lut g_lut;

// cnt <= 32 characters -> their 2-bit codes, first character most significant; `bad` collects bit 7 of anything that is not ACGT
inline uint64_t run_scalar(const unsigned char *s, int cnt, unsigned &bad)
{
	uint64_t v = 0;
	for (int j = 0; j < cnt; j++) { const unsigned c = g_lut.c[s[j]]; bad |= c; v = (v << 2) | (c & 3); }
	return v;
}

// 16 characters in one step.  (c >> 1) & 3 maps A C G T to 0 1 3 2; x ^ (x >> 1) turns that into 0 1 2 3.  Pairs are
// merged with two multiply-adds (4a + b, then 16p + q), the four bytes that result are put in order by one shuffle.
__attribute__((target("ssse3"))) inline uint32_t pack16(const unsigned char *s, unsigned &bad)
{
	const __m128i v = _mm_loadu_si128((const __m128i *)s);
	const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('A')), _mm_cmpeq_epi8(v, _mm_set1_epi8('C'))),
	                                _mm_or_si128(_mm_cmpeq_epi8(v, _mm_set1_epi8('G')), _mm_cmpeq_epi8(v, _mm_set1_epi8('T'))));
	bad |= (unsigned)(_mm_movemask_epi8(ok) != 0xFFFF) << 7;
	const __m128i t = _mm_and_si128(_mm_srli_epi16(v, 1), _mm_set1_epi8(3));
	const __m128i code = _mm_xor_si128(t, _mm_and_si128(_mm_srli_epi16(t, 1), _mm_set1_epi8(1)));
	const __m128i p4 = _mm_maddubs_epi16(code, _mm_set1_epi16(0x0104));          // bytes (4, 1): 4 * c[2i] + c[2i+1], 8 x 16 bit
	const __m128i p8 = _mm_madd_epi16(p4, _mm_set1_epi32(0x00010010));            // words (16, 1): 16 * p[2i] + p[2i+1], 4 x 32 bit
	const __m128i sh = _mm_shuffle_epi8(p8, _mm_set_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12));
	return (uint32_t)_mm_cvtsi128_si32(sh);
}

__attribute__((target("ssse3"))) inline uint64_t run_simd(const unsigned char *s, int cnt, unsigned &bad)
{
	if (cnt < 16) return run_scalar(s, cnt, bad);
	if (cnt == 32) return ((uint64_t)pack16(s, bad) << 32) | pack16(s + 16, bad);
	const int r = cnt - 16;                                            // the last 16 characters overlap the first 16 by 16 - r
	const uint64_t hi = pack16(s, bad), lo = pack16(s + r, bad);
	return (hi << (2 * r)) | (lo & ((1ULL << (2 * r)) - 1));
}

__attribute__((target("ssse3"))) bool pack_range_simd(const KmxStrBatch &b, int W, uint64_t lo, uint64_t hi, uint64_t *dst, std::vector<uint64_t> *dirty)
{
	unsigned any = 0;
	const int len = b.len;
	for (uint64_t i = lo; i < hi; i++) {
		const unsigned char *s = (const unsigned char *)(b.ptrs ? b.ptrs[i] : b.flat + i * (uint64_t)b.stride);
		if (b.ptrs && i + 8 < hi) __builtin_prefetch(b.ptrs[i + 8]);
		unsigned bad = 0;
		if (W == 1) dst[i - lo] = run_simd(s, len, bad);
		else {
			dst[2 * (i - lo)] = run_simd(s, len - 32, bad);
			dst[2 * (i - lo) + 1] = run_simd(s + (len - 32), 32, bad);
		}
		if (bad & 0x80) { any = 1; if (dirty) dirty->push_back(i); }
	}
	return !any;
}

bool pack_range_scalar(const KmxStrBatch &b, int W, uint64_t lo, uint64_t hi, uint64_t *dst, std::vector<uint64_t> *dirty)
{
	unsigned any = 0;
	const int len = b.len;
	for (uint64_t i = lo; i < hi; i++) {
		const unsigned char *s = (const unsigned char *)(b.ptrs ? b.ptrs[i] : b.flat + i * (uint64_t)b.stride);
		unsigned bad = 0;
		if (W == 1) dst[i - lo] = run_scalar(s, len, bad);
		else {
			dst[2 * (i - lo)] = run_scalar(s, len - 32, bad);
			dst[2 * (i - lo) + 1] = run_scalar(s + (len - 32), 32, bad);
		}
		if (bad & 0x80) { any = 1; if (dirty) dirty->push_back(i); }
	}
	return !any;
}

}   // namespace

bool kmx_pack_strings(const KmxStrBatch &b, int W, uint64_t lo, uint64_t hi, uint64_t *dst, std::vector<uint64_t> *dirty)
{
	static const bool simd = __builtin_cpu_supports("ssse3");
	return simd ? pack_range_simd(b, W, lo, hi, dst, dirty) : pack_range_scalar(b, W, lo, hi, dst, dirty);
}

void kmx_gather_strings(const KmxStrBatch &b, uint64_t lo, uint64_t hi, unsigned char *dst)
{
	const uint64_t len = (uint64_t)b.len;
	if (!b.ptrs && (uint64_t)b.stride == len) { memcpy(dst, b.flat + lo * len, (hi - lo) * len); return; }
	for (uint64_t i = lo; i < hi; i++) memcpy(dst + (i - lo) * len, b.ptrs ? b.ptrs[i] : b.flat + i * (uint64_t)b.stride, len);
}

int kmx_host_cpus(void)
{
	int n = 1;
	cpu_set_t set;
	CPU_ZERO(&set);
	if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
	long long quota = -1, period = 100000;
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {              // cgroup v2: "<quota|max> <period>"
		char q[32] = "";
		if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
		fclose(f);
	} else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
		if (fscanf(g, "%lld", &quota) != 1) quota = -1;
		fclose(g);
		if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
	}
	if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
	return n;
}
