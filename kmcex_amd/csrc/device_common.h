// device_common.h -- per-lane arithmetic shared by every gfx950 kernel of the KModel hot path.
//
// What it restates (behaviour, not code) from the reference:
//   MurmurHash64A over the ASCII k-mer string      tools.hpp:16-50
//   `% length` on 64-bit positions                   kmodel.hpp:378,503,600,633
//   MSB-first bit order inside a byte                kmodel.hpp:576-588
//   canonicalisation through one u64                 tools.hpp:63-76,130-139,160-167
// MI355X-first choices: k-mers stay packed (2 bits/base) in registers; the ASCII bytes the hash is
// defined over are rebuilt 4 bases at a time with one v_perm_b32; the seed-independent block pre-mix
// (k*=m; k^=k>>47; k*=m) is hoisted out of the per-seed loop; the modulo is an exact mul-hi reciprocal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define KMX_BUCKET (1u << 18)        // kmodel.hpp:276 bucket_size
#define KMX_MAX_NH 16
#define KMX_MAX_NB 16

static constexpr u64 MURMUR_M = 0xc6a4a7935bd1e995ULL;

// ---------------------------------------------------------------- exact u64 modulo by a runtime constant
struct ModU64 {
	u64 d;   // divisor (0 = empty filter: never probed)
	u64 M;   // floor(2^64 / d)   (d == 1: 2^64-1)
};

static inline ModU64 make_mod(u64 d)
{
	ModU64 m;
	m.d = d;
	m.M = d <= 1 ? ~0ULL : (u64)((((unsigned __int128)1) << 64) / d);
	return m;
}

// q' = mulhi(h, M) is floor(h/d) or one less, so one conditional subtract makes the remainder exact.
__device__ __forceinline__ u64 mod_u64(u64 h, const ModU64 m)
{
	u64 q = __umul64hi(h, m.M);
	u64 r = h - q * m.d;
	return r >= m.d ? r - m.d : r;
}

// ---------------------------------------------------------------- packed k-mer -> ASCII blocks
// 4 bases (base 0 in bits 7:6) -> 4 ASCII bytes, base 0 in the lowest byte.
__device__ __forceinline__ u32 ascii4(u32 c8)
{
	u32 sel = ((c8 << 24) | (c8 << 14) | (c8 << 4) | (c8 >> 6)) & 0x03030303u;
	return __builtin_amdgcn_perm(0x54474341u, 0x54474341u, sel);   // bytes: 'A','C','G','T'
}

// 8 bases (base 0 in bits 15:14) -> one little-endian 8-byte block of the string
__device__ __forceinline__ u64 ascii8(u32 c16)
{
	return (u64)ascii4((c16 >> 8) & 0xFFu) | ((u64)ascii4(c16 & 0xFFu) << 32);
}

__device__ __forceinline__ u64 premix(u64 w)
{
	w *= MURMUR_M;
	w ^= w >> 47;
	w *= MURMUR_M;
	return w;
}

// geometry of one hashed string length (k or k-2)
struct StrGeom {
	int nblk;   // len / 8
	int rem;    // len & 7
	u64 lenm;   // (u64)len * m
};
static inline StrGeom make_geom(int len)
{
	StrGeom g;
	g.nblk = len / 8;
	g.rem = len & 7;
	g.lenm = (u64)(long long)len * MURMUR_M;
	return g;
}

// A k-mer left-aligned in W words (base 0 in the top 2 bits of x[0]).
template <int W> struct Aligned { u64 x[W]; };

template <int W> __device__ __forceinline__ Aligned<W> left_align(const u64 *v, int k);
template <> __device__ __forceinline__ Aligned<1> left_align<1>(const u64 *v, int k)
{
	Aligned<1> a;
	a.x[0] = v[0] << (64 - 2 * k);
	return a;
}
template <> __device__ __forceinline__ Aligned<2> left_align<2>(const u64 *v, int k)
{
	Aligned<2> a;
	int s = 128 - 2 * k;                       // 0..62 for k in 33..64
	a.x[0] = s ? (v[0] << s) | (v[1] >> (64 - s)) : v[0];
	a.x[1] = v[1] << s;
	return a;
}
template <int W> __device__ __forceinline__ Aligned<W> drop_first_base(Aligned<W> a)
{
	Aligned<W> r;
#pragma unroll
	for (int w = 0; w < W; w++) r.x[w] = (a.x[w] << 2) | (w + 1 < W ? a.x[w + 1] >> 62 : 0);
	return r;
}

// Pre-mixed blocks + tail of one string: everything of MurmurHash64A that does not depend on the seed.
template <int W> struct Premixed {
	u64 blk[4 * W];
	u64 tail;
};

template <int W> __device__ __forceinline__ Premixed<W> premix_string(const Aligned<W> a, const StrGeom g)
{
	Premixed<W> p;
	p.tail = 0;
#pragma unroll
	for (int b = 0; b < 4 * W; b++) {
		u32 c16 = (u32)(a.x[b >> 2] >> (48 - 16 * (b & 3))) & 0xFFFFu;
		u64 w = ascii8(c16);
		p.blk[b] = premix(w);
		if (b == g.nblk) p.tail = g.rem ? (w & ((1ULL << (8 * g.rem)) - 1)) : 0;
	}
	return p;
}

template <int W> __device__ __forceinline__ u64 murmur_seeded(const Premixed<W> &p, const StrGeom g, u32 seed)
{
	u64 h = (u64)seed ^ g.lenm;
#pragma unroll
	for (int b = 0; b < 4 * W; b++)
		if (b < g.nblk) { h ^= p.blk[b]; h *= MURMUR_M; }
	if (g.rem) { h ^= p.tail; h *= MURMUR_M; }
	h ^= h >> 47;
	h *= MURMUR_M;
	h ^= h >> 47;
	return h;
}

// ---------------------------------------------------------------- bit addressing
// On-disk arrays are bytes with bit `pos` at byte pos>>3, mask 0x80>>(pos&7).  Viewed as little-endian
// u32 words: word pos>>5, bit 8*((pos>>3)&3) + 7-(pos&7).
__device__ __forceinline__ u32 bit_in_word32(u64 pos) { return (u32)(8 * ((pos >> 3) & 3) + 7 - (pos & 7)); }

__device__ __forceinline__ void bloom_set(u32 *bits, u64 pos) { atomicOr(bits + (pos >> 5), 1u << bit_in_word32(pos)); }
__device__ __forceinline__ bool bloom_get(const u32 *bits, u64 pos) { return (bits[pos >> 5] >> bit_in_word32(pos)) & 1u; }

// Coupled arrays live in HBM as one 32-bit "cell" per 16 positions (= 2 on-disk bytes of each array):
//   bits  0..15 value (bit_array_1)   bits 16..31 tag (bit_array_2)
// so one 4-byte access reads or commits a position (tag and value travel together).  save() de-interleaves to the on-disk
// layout.  (Round 1 kept two 16-bit claim fields in an upper half; claims are a partitioned stream now, kernels.hip.)
typedef u32 cell_t;
__device__ __forceinline__ u32 bit_in_cell(u64 pos) { return (u32)(8 * ((pos >> 3) & 1) + 7 - (pos & 7)); }
#define CELL_VAL(bit)      (1u << (bit))
#define CELL_TAG(bit)      (1u << (16 + (bit)))

// ---------------------------------------------------------------- canonicalisation (A.7)
__device__ __forceinline__ u64 rev2_u64(u64 x)
{
	x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
	x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
	x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
	return (x >> 32) | (x << 32);
}

// The reference canonicalises through ONE u64 (tools.hpp:160-167): u = the last min(k,32) bases,
// rc = the loop of tools.hpp:130-139 run k times.  Correct for k <= 32; for k > 32 it overflows and the
// "canonical" k-mer becomes (k-32) x 'A' followed by rc -- reproduced here bit for bit (Q4).
template <int W> __device__ __forceinline__ void min_kmer(u64 *v, int k)
{
	u64 u = v[W - 1];
	u64 r32 = rev2_u64(~u);
	u64 rc;
	if (k <= 32) rc = r32 >> (64 - 2 * k);
	else { int s = 2 * (k - 32); rc = s >= 64 ? ~0ULL : ((r32 << s) | ((1ULL << s) - 1)); }
	if (u <= rc) return;
	v[W - 1] = rc;
	if (W == 2) v[0] = 0;
}
