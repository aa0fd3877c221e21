// kernels.hip -- hand-written gfx950 kernels of the KModel insert/query hot path.
//
// One lane = one k-mer.  Wave64; 256-thread workgroups; grids of >= 1024 workgroups on the hot kernels.
// The path is HBM random-access bound (1-bit gathers/scatters addressed by MurmurHash64A), so there is
// no MFMA anywhere: the levers are (i) one 4-byte cell per touched position (tag + value co-located), (ii) all nh
// touches of a k-mer in flight at once, (iii) everything that needs no order -- the claims of a round, the km_back and
// Bloom bits -- travels as LDS-staged, partitioned streams instead of one memory-side atomic per item, (iv) no host round
// trips inside a block: list lengths, contended sets and statistics stay in HBM and every kernel reads them there.
//
// Insert (reference: kmodel.hpp:543-622, sequential greedy) is reproduced bit for bit:
//   round = nb independent (buffer i, array (i+t)%nb) pairs.  Per pair the reference walks the buffer
//   in order; a k-mer fits iff no set tag carries a different value bit.  Here:
//   A  check_emit    every k-mer checks the committed state; the ones that fit ("candidates") emit one claim tuple
//                    (position, wanted value, slot) per still-untagged position into hash-partitioned bins;
//   D  detect        bin by bin, an LDS table finds the positions wanted with both values and marks the candidates
//                    involved as contended;
//   B  commit        an unmarked candidate cannot interact with any other candidate of the list, so it commits
//                    (atomic OR per untagged position) in parallel; the marked ones form the contended set U;
//   S  slow path     U is resolved in list order by priority reservations: a k-mer that holds the smallest
//                    index on all its slots has no earlier undecided k-mer touching them, so its outcome is
//                    the sequential one.  A single-workgroup finisher per list decides in LDS (k_slow_finish);
//                    grid-wide passes (k_slow_resolve0 / k_slow_reserve / k_slow_resolve) run first when U is
//                    larger than what its registers hold.
//   R  reorder       applies what the finisher decided, then the reference's unstable compaction
//                    (kmodel.hpp:529-540) in one launch with lazily filled holes.
//   Fused launches: in rounds 0 and 1 one group of lists commits beside the check of the next (k_round_commit_check:
//   memory-side atomics and gathers side by side); the finisher launches of those rounds (one workgroup per list) carry
//   the km_back emission of the previous block as extra workgroups.
#include "kmx_types.h"
#include <cstdlib>

__constant__ u32 c_seeds[128] = {   // tools.hpp:9 -- 128 consecutive primes (data)
	46757, 46769, 46771, 46807, 46811, 46817, 46819, 46829, 46831, 46853, 46861, 46867, 46877, 46889, 46901, 46919,
	46933, 46957, 46993, 46997, 47017, 47041, 47051, 47057, 47059, 47087, 47093, 47111, 47119, 47123, 47129, 47137,
	47143, 47147, 47149, 47161, 47189, 47207, 47221, 47237, 47251, 47269, 47279, 47287, 47293, 47297, 47303, 47309,
	47317, 47339, 47351, 47353, 47363, 47381, 47387, 47389, 47407, 47417, 47419, 47431, 47441, 47459, 47491, 47497,
	47501, 47507, 47513, 47521, 47527, 47533, 47543, 47563, 47569, 47581, 47591, 47599, 47609, 47623, 47629, 47639,
	47653, 47657, 47659, 47681, 47699, 47701, 47711, 47713, 47717, 47737, 47741, 47743, 47777, 47779, 47791, 47797,
	47807, 47809, 47819, 47837, 47843, 47857, 47869, 47881, 47903, 47911, 47917, 47933, 47939, 47947, 47951, 47963,
	47969, 47977, 47981, 48017, 48023, 48029, 48049, 48073, 48079, 48091, 48109, 48119, 48121, 48131, 48157, 48163};

// ------------------------------------------------------------------------------------------ helpers
template <int W> __device__ __forceinline__ void load_kmer(const u64 *base, u64 i, u64 *v)
{
#pragma unroll
	for (int w = 0; w < W; w++) v[w] = base[i * W + w];
}
template <int W> __device__ __forceinline__ void store_kmer(u64 *base, u64 i, const u64 *v)
{
#pragma unroll
	for (int w = 0; w < W; w++) base[i * W + w] = v[w];
}

// Bloom insert / probe of one pre-mixed string with HashSeeds[0..nhash-1] (kmodel.hpp:498-506, :373-383).
// Empty filter: nothing to set, every probe misses (divergence D2).
template <int W> __device__ __forceinline__ void bloom_insert_pm(const Premixed<W> &pm, const StrGeom g, u32 *bits, const ModU64 md, int nhash)
{
	if (!md.d) return;
	for (int j = 0; j < nhash; j++) {
		const u64 pos = mod_u64(murmur_seeded<W>(pm, g, c_seeds[j]), md);
		// bits are only ever set, so a bit that reads 1 (even from a stale cache line) needs no atomic: late in a
		// build most Bloom bits are already set and a load is 2-3x cheaper than a memory-side atomic
		if (!bloom_get(bits, pos)) bloom_set(bits, pos);
	}
}
template <int W> __device__ __forceinline__ bool bloom_check_pm(const Premixed<W> &pm, const StrGeom g, const u32 *bits, const ModU64 md, int nhash)
{
	if (!md.d) return false;
	bool ok = true;
	for (int j = 0; j < nhash && ok; j++) ok = bloom_get(bits, mod_u64(murmur_seeded<W>(pm, g, c_seeds[j]), md));
	return ok;
}

// Wave-aggregated counters: one atomic per wave instead of one per lane (same-address atomics serialise).
// Call from converged code; lanes that already returned simply do not take part in the ballot.
__device__ __forceinline__ void wave_count_add(u64 *ctr, bool pred)
{
	const u64 mask = __ballot(pred);
	if (!mask) return;
	const int lane = threadIdx.x & 63;
	if (lane == __ffsll((long long)mask) - 1) atomicAdd(ctr, (u64)__popcll(mask));
}
template <typename T> __device__ __forceinline__ T wave_append_slot(T *ctr, bool pred)
{
	const u64 mask = __ballot(pred);
	if (!mask) return 0;
	const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
	T base = 0;
	if (lane == leader) base = atomicAdd(ctr, (T)__popcll(mask));
	base = __shfl(base, leader, 64);
	return base + (T)__popcll(mask & ((1ULL << lane) - 1));
}

// wait until every vector-memory operation of this wave (loads, stores, atomics) has been acknowledged
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T> __device__ __forceinline__ T cell_load_coherent(const T *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load sc1: bypasses this CU's L1
}

// The nh touches of one k-mer on one coupled array.
template <int NHM> struct Touches {
	u64 pos[NHM];
	cell_t cell[NHM];
};

template <int W, int NHM, bool COHERENT>
__device__ __forceinline__ void gather_touches(const ModelDev &md, const Premixed<W> &pm, int a, Touches<NHM> &t)
{
	const cell_t *cells = md.cells[a];
	const int sbase = a * md.nh;
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if (j < md.nh) {
			t.pos[j] = mod_u64(murmur_seeded<W>(pm, md.gfull, c_seeds[(sbase + j) & 127]), md.km_mod);
			t.cell[j] = COHERENT ? cell_load_coherent(cells + (t.pos[j] >> 4)) : cells[t.pos[j] >> 4];
		}
}

// The fused launches are bound by the number of random DRAM operations, and a failing attempt usually fails
// on one of its first positions: gather md.nh_first positions, and the others only for the lanes none of them refused.
// Same verdict as gather_touches + touches_conflict (the conflicts are OR-ed, kmodel.hpp:604-610); pos / cell beyond the
// first group are only defined for lanes that return false.
// (Hashing the next group while the loads of the current one are in flight -- pinned there with an asm -- changed nothing.)
template <int W, int NHM> struct StagedGather {
	const ModelDev &md;
	const Premixed<W> &pm;
	const cell_t *cells;
	int sbase;
	u32 bin;
	Touches<NHM> &t;
	__device__ __forceinline__ void hash(int lo, int hi)
	{
#pragma unroll
		for (int j = 0; j < NHM; j++)
			if (j >= lo && j < hi) {
				t.pos[j] = mod_u64(murmur_seeded<W>(pm, md.gfull, c_seeds[(sbase + j) & 127]), md.km_mod);
			}
	}
	__device__ __forceinline__ void load(int lo, int hi)
	{
#pragma unroll
		for (int j = 0; j < NHM; j++)
			if (j >= lo && j < hi) t.cell[j] = cells[t.pos[j] >> 4];
	}
	__device__ __forceinline__ bool conflict(int lo, int hi)
	{
		bool fail = false;
#pragma unroll
		for (int j = 0; j < NHM; j++)
			if (j >= lo && j < hi) {
				const u32 b = bit_in_cell(t.pos[j]);
				fail |= ((u32)(t.cell[j] >> (16 + b)) & 1u) && (((u32)(t.cell[j] >> b) & 1u) != ((bin >> j) & 1u));
			}
		return fail;
	}
};
// `stages`: how many of the three groups this lane fetched (accounting: random loads actually issued).
template <int W, int NHM>
__device__ __forceinline__ bool gather_touches_staged(const ModelDev &md, const Premixed<W> &pm, int a, u32 bin, Touches<NHM> &t, int &stages)
{
	StagedGather<W, NHM> g = {md, pm, md.cells[a], a * md.nh, bin, t};
	const int j1 = md.nh_first, j2 = md.nh_second, nh = md.nh;
	stages = 1;
	g.hash(0, j1);
	g.load(0, j1);
	if (g.conflict(0, j1)) return true;
	stages = 2;
	g.hash(j1, j2);
	g.load(j1, j2);
	if (g.conflict(j1, j2)) return true;
	stages = 3;
	g.hash(j2, nh);
	g.load(j2, nh);
	return g.conflict(j2, nh);
}

// insert_to_array's check (kmodel.hpp:604-610): fail iff a set tag carries the other value
template <int NHM> __device__ __forceinline__ bool touches_conflict(const ModelDev &md, const Touches<NHM> &t, u32 bin)
{
	bool fail = false;
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if (j < md.nh) {
			u32 b = bit_in_cell(t.pos[j]);
			u32 tag = (u32)(t.cell[j] >> (16 + b)) & 1u, val = (u32)(t.cell[j] >> b) & 1u;
			fail |= tag && (val != ((bin >> j) & 1u));
		}
	return fail;
}

// The value a k-mer leaves at the position of its hash j: a k-mer may hit one position with several of its hashes, and the
// reference's set loop ORs their value bits (kmodel.hpp:611-618).  Every atomic for that position must carry the FINAL
// value: set with one atomic per hash (tag|0, then tag|1), the position is for a moment tagged with value 0, a concurrent
// gatherer that wants 0 there takes it for settled, drops it from what it must hold, and wins ahead of its turn -- the
// timing-dependent result of tools/soak_case.py.  `among`: the hashes that take part (bit mask).
template <int NHM> __device__ __forceinline__ u32 value_at_position(const ModelDev &md, const u64 *pos, u32 among, u32 bin, int j)
{
	u32 v = (bin >> j) & 1u;
#pragma unroll
	for (int j2 = 0; j2 < NHM; j2++)
		if (j2 < md.nh && j2 != j && ((among >> j2) & 1u) && pos[j2] == pos[j]) v |= (bin >> j2) & 1u;
	return v;
}

// kmodel.hpp:611-618: set tag (and value) bits.  (The (k-2)-mer of a success goes into km_back -- :548-550 -- once per block
// or per ring round: k_kmback_emit.)
template <int W, int NHM>
__device__ __forceinline__ void commit_touches(const ModelDev &md, const Touches<NHM> &t, u32 bin, int a)
{
	cell_t *cells = md.cells[a];
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if (j < md.nh) {
			u32 b = bit_in_cell(t.pos[j]);
			if (!((t.cell[j] >> (16 + b)) & 1u))          // already tagged => already carries this value
				atomicOr(cells + (t.pos[j] >> 4), CELL_TAG(b) | (value_at_position<NHM>(md, t.pos, ~0u, bin, j) ? CELL_VAL(b) : 0u));
		}
}

// ------------------------------------------------------------------------------------------ pass 1
// get_km_kmer_count (kmodel.hpp:423-428): histogram of the bf_num lowest counts.
__global__ __launch_bounds__(256) void k_histogram(const u32 *counts, u64 n, int ci, int cs, int bf_num, u64 *n_bf, u64 *stats)
{
	u64 local[3] = {0, 0, 0}, bad = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		u32 c = counts[i];
		if (c < (u32)ci || c > (u32)cs) bad++;
		else if (c < (u32)(ci + bf_num)) local[c - ci]++;
	}
	for (int f = 0; f < 3; f++)
		if (local[f]) atomicAdd(n_bf + f, local[f]);
	if (bad) atomicAdd(stats + ST_BAD_COUNT, bad);
}

// block-wide exclusive scan of one int per thread over the first 256 threads (the other threads of a larger workgroup
// only take part in the barriers and get the total)
__device__ __forceinline__ int block_excl_scan_256(int v, int *s_tmp, int *total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int incl = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		int t = __shfl_up(incl, d, 64);
		if (lane >= d) incl += t;
	}
	if (lane == 63 && wave < 4) s_tmp[wave] = incl;
	__syncthreads();
	int wbase = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < 4; w++) {
		int c = s_tmp[w];
		if (w < wave) wbase += c;
		tot += c;
	}
	__syncthreads();
	if (total) *total = tot;
	return wbase + incl - v;
}

// ------------------------------------------------------------------------------------------ partitioned bit-set
// bs_block_emit: every thread of a BT-thread workgroup brings up to K bit addresses of the filter; they are counted per
// bin in LDS, each bin's run is reserved with ONE global atomic, the tuples are sorted by bin in LDS and written out run
// by run.  All threads must call it (barriers inside); the LDS arrays are the caller's.
#define BS_LDS(K) __shared__ int s_bs_cnt[BS_BINS], s_bs_off[BS_BINS], s_bs_base[BS_BINS], s_bs_tmp[4]; __shared__ u64 s_bs_stage[256 * (K)]
// the same arrays carved out of a byte pool the kernel owns (kernels that run one of several bodies per workgroup)
#define BS_LDS_BYTES(K, BT) ((BT) * (K) * 8 + (3 * BS_BINS + 4) * 4)
#define BS_LDS_AT(K, BT, p) u64 *s_bs_stage = (u64 *)(p); int *s_bs_cnt = (int *)((p) + (BT) * (K) * 8), *s_bs_off = s_bs_cnt + BS_BINS, *s_bs_base = s_bs_off + BS_BINS, *s_bs_tmp = s_bs_base + BS_BINS
template <int K, int BT = 256> __device__ __forceinline__ void bs_block_emit(const BitScatter &bs, const u64 *v, u32 valid, int *s_cnt, int *s_off, int *s_base, int *s_tmp, u64 *s_stage)
{
	static_assert(BT >= BS_BINS, "one thread per bin");
	if (threadIdx.x < BS_BINS) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	int rank[K];
#pragma unroll
	for (int j = 0; j < K; j++)
		if ((valid >> j) & 1u) rank[j] = atomicAdd(&s_cnt[v[j] >> bs.wshift], 1);
	__syncthreads();
	int total;
	{
		const int c = threadIdx.x < BS_BINS ? s_cnt[threadIdx.x] : 0;
		const int ex = block_excl_scan_256(c, s_tmp, &total);
		if (threadIdx.x < BS_BINS) {
			s_off[threadIdx.x] = ex;
			s_base[threadIdx.x] = c ? atomicAdd(bs.cnt + threadIdx.x, c) : 0;
		}
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < K; j++)
		if ((valid >> j) & 1u) {
			const u32 b = (u32)(v[j] >> bs.wshift);
			s_stage[s_off[b] + rank[j]] = ((u64)b << 32) | (u32)(v[j] - ((u64)b << bs.wshift));
		}
	__syncthreads();
	for (int q = threadIdx.x; q < total; q += BT) {
		const u64 e = s_stage[q];
		const u32 b = (u32)(e >> 32), o = (u32)e;
		const u32 g = (u32)s_base[b] + (u32)(q - s_off[b]);
		if (g < bs.cap) bs.tup[(u64)b * bs.cap + g] = o;
		else atomicOr(bs.words + (((u64)b << bs.wshift) >> 5) + (o >> 5), 1u << (o & 31));   // bin full: set the bit directly (exact either way)
	}
	__syncthreads();
}

// (defined with k_kmback_emit; the finisher launch runs it too, as rider workgroups)
template <int W, int NHM, int KPT, int BT = 256>
__device__ __forceinline__ void kmback_emit_body(const ModelDev &md, const BlockDev &bd, const u64 *kmers, const unsigned char *surv, int pp, int n_in_block,
                                                 const BitScatter &bs, int i, int bx, int gx, unsigned char *lds);

// ------------------------------------------------------------------------------------------ classification
// Pass 2 front end (kmodel.hpp:70-73): Bloom-class k-mers are inserted right here (commutative ORs, any
// order); coupled-array k-mers are compacted, in listing order, into the staging stream.
#define CLS_TILE KMX_CLS_TILE
template <int W, int NHM> __global__ __launch_bounds__(256) void k_classify_count(ModelDev md, const u64 *kmers, const u32 *counts, u64 n, int *tile_cnt, u64 *stats, BitScatter bs)
{
	constexpr int K = 2 * NHM - 3;                                   // (nh-1) + (nh-2) positions of one Bloom-class k-mer
	BS_LDS(K);
	__shared__ int s_cnt, s_nb;
	__shared__ unsigned short s_bloom[CLS_TILE];                     // the tile's Bloom-class k-mers (offsets in the tile)
	if (threadIdx.x == 0) { s_cnt = 0; s_nb = 0; }
	__syncthreads();
	int mine = 0;
	u64 base = (u64)blockIdx.x * CLS_TILE;
	for (int q = 0; q < CLS_TILE / 256; q++) {
		u64 i = base + (u64)q * 256 + threadIdx.x;
		if (i >= n) break;
		u32 c = counts[i];
		if (c < (u32)md.ci || c > (u32)md.cs) { atomicAdd(stats + ST_BAD_COUNT, 1ULL); continue; }
		if (c < (u32)(md.ci + md.bf_num)) {
			if (md.bloom_direct) {
				int f = (int)(c - (u32)md.ci);
				u64 v[W];
				load_kmer<W>(kmers, i, v);
				Aligned<W> al = left_align<W>(v, md.k);
				Premixed<W> pf = premix_string<W>(al, md.gfull);
				bloom_insert_pm<W>(pf, md.gfull, md.bf[f], md.bf_mod[f], md.nh - 1);                 // kmodel.hpp:474
				Premixed<W> pb = premix_string<W>(drop_first_base<W>(al), md.gback);
				bloom_insert_pm<W>(pb, md.gback, md.bf_back[f], md.bf_back_mod[f], md.nh - 2);       // kmodel.hpp:475-476
			} else s_bloom[atomicAdd(&s_nb, 1)] = (unsigned short)(q * 256 + threadIdx.x);
		} else mine++;
	}
	if (mine) atomicAdd(&s_cnt, mine);
	__syncthreads();
	if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_cnt;
	if (md.bloom_direct) return;
	// the Bloom-class k-mers of the tile, 256 at a time: their bit addresses in the slab go to the BitScatter
	const int nbl = s_nb;
	for (int c0 = 0; c0 < nbl; c0 += 256) {                          // uniform
		u64 v[K];
		u32 valid = 0;
		if (c0 + (int)threadIdx.x < nbl) {
			const u64 i = base + s_bloom[c0 + threadIdx.x];
			const int f = (int)(counts[i] - (u32)md.ci);
			u64 km[W];
			load_kmer<W>(kmers, i, km);
			Aligned<W> al = left_align<W>(km, md.k);
			Premixed<W> pf = premix_string<W>(al, md.gfull);
			Premixed<W> pb = premix_string<W>(drop_first_base<W>(al), md.gback);
#pragma unroll
			for (int j = 0; j < NHM - 1; j++)
				if (j < md.nh - 1 && md.bf_mod[f].d) {
					const u64 pos = mod_u64(murmur_seeded<W>(pf, md.gfull, c_seeds[j]), md.bf_mod[f]);
					v[j] = ((md.bf_woff[f] + (pos >> 5)) << 5) | bit_in_word32(pos);
					valid |= 1u << j;
				}
#pragma unroll
			for (int j = 0; j < NHM - 2; j++)
				if (j < md.nh - 2 && md.bf_back_mod[f].d) {
					const u64 pos = mod_u64(murmur_seeded<W>(pb, md.gback, c_seeds[j]), md.bf_back_mod[f]);
					v[NHM - 1 + j] = ((md.bf_back_woff[f] + (pos >> 5)) << 5) | bit_in_word32(pos);
					valid |= 1u << (NHM - 1 + j);
				}
		}
		bs_block_emit<K>(bs, v, valid, s_bs_cnt, s_bs_off, s_bs_base, s_bs_tmp, s_bs_stage);
	}
}

// exclusive scan of up to 2^20 tile counts by one workgroup; total -> *total_out
__global__ __launch_bounds__(1024) void k_scan_tiles(const int *cnt, int *off, int n_tiles, int *total_out)
{
	__shared__ int s[1024];
	__shared__ int carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int base = 0; base < n_tiles; base += 1024) {
		int i = base + threadIdx.x;
		int v = i < n_tiles ? cnt[i] : 0;
		s[threadIdx.x] = v;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) {
			int t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
			__syncthreads();
			s[threadIdx.x] += t;
			__syncthreads();
		}
		if (i < n_tiles) off[i] = carry + s[threadIdx.x] - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry += s[1023];
		__syncthreads();
	}
	if (threadIdx.x == 0) *total_out = carry;
}

// compaction of the coupled-array k-mers of a tile into the staging stream, in listing order: lane-contiguous loads, a
// ballot per 256 elements for the positions, consecutive stores
template <int W> __global__ __launch_bounds__(256) void k_classify_scatter(ModelDev md, const u64 *kmers, const u32 *counts, u64 n, const int *tile_off, u64 *stg_kmers, u32 *stg_counts, u64 stg_base)
{
	__shared__ int s_w[2][4];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const u64 tile = (u64)blockIdx.x * CLS_TILE;
	u64 off = stg_base + (u64)tile_off[blockIdx.x];
	for (int q = 0; q < CLS_TILE / 256; q++) {
		const u64 i = tile + (u64)q * 256 + threadIdx.x;
		u32 c = 0;
		bool f = false;
		if (i < n) { c = counts[i]; f = c >= (u32)(md.ci + md.bf_num) && c <= (u32)md.cs; }
		const u64 mask = __ballot(f);
		if (lane == 0) s_w[q & 1][wave] = (int)__popcll(mask);
		__syncthreads();                                             // (the two buffers alternate: one barrier per step is enough)
		int before = 0, total = 0;
#pragma unroll
		for (int w = 0; w < 4; w++) { const int t = s_w[q & 1][w]; before += w < wave ? t : 0; total += t; }
		if (f) {
			const u64 o = off + (u64)before + (u64)__popcll(mask & ((1ULL << lane) - 1));
			u64 v[W];
			load_kmer<W>(kmers, i, v);
			store_kmer<W>(stg_kmers, o, v);
			stg_counts[o] = c;
		}
		off += (u64)total;
	}
}

// Block-aggregated append: ONE global atomic per workgroup (same-address atomics serialise at ~11 ns each on
// gfx950, so per-lane or even per-wave counters on a hot word dominate a kernel).  Every thread of the
// workgroup must call it; s_cnt must have been zeroed and barriered.  Returns the global slot for pred lanes.
__device__ __forceinline__ int block_append_slot(int *gctr, bool pred, int *s_cnt, int *s_base)
{
	const u64 mask = __ballot(pred);
	const int lane = threadIdx.x & 63;
	int wbase = 0;
	if (lane == 0 && mask) wbase = atomicAdd(s_cnt, (int)__popcll(mask));
	wbase = __shfl(wbase, 0, 64);
	__syncthreads();
	if (threadIdx.x == 0) *s_base = *s_cnt ? atomicAdd(gctr, *s_cnt) : 0;
	__syncthreads();
	return *s_base + wbase + (int)__popcll(mask & ((1ULL << lane) - 1));
}
__device__ __forceinline__ void block_count_add(u64 *gctr, bool pred, int *s_cnt)
{
	const u64 mask = __ballot(pred);
	if ((threadIdx.x & 63) == 0 && mask) atomicAdd(s_cnt, (int)__popcll(mask));
	__syncthreads();
	if (threadIdx.x == 0 && *s_cnt) atomicAdd(gctr, (u64)*s_cnt);
}

// slot x of list i -> index of its k-mer in buffer i, filling a pending hole of the last reorder on the way
__device__ __forceinline__ u32 list_entry(const BlockDev &bd, int pp, u64 row, int x)
{
	u32 e = bd.list[pp][row + x];
	return (e & LIST_HOLE) ? bd.mover[pp][row + (e & ~LIST_HOLE)] : e;
}

// lists of a fresh block: identity permutation (kmodel.hpp:509-513 fills row-major in listing order)
__global__ __launch_bounds__(256) void k_block_init(BlockDev bd, int nb, int pp, int n_in_block)
{
	int i = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
	bd.list[pp][(u64)i * KMX_BUCKET + x] = (u32)x;
	bd.surv[(u64)i * KMX_BUCKET + x] = 0;
	if (x < KMX_NTILES) { bd.tile_cnt[0][i * KMX_NTILES + x] = 0; bd.tile_cnt[1][i * KMX_NTILES + x] = 0; }
	if (x == 0) {
		int lo = i * (int)KMX_BUCKET;
		int ni = n_in_block - lo;
		bd.n[pp][i] = ni < 0 ? 0 : (ni > (int)KMX_BUCKET ? (int)KMX_BUCKET : ni);      // kmodel.hpp:521-525
		for (int s = 0; s < KMX_NSLOW; s++) bd.Un[UN_IDX(s, i, nb)] = 0;
	}
}

// ------------------------------------------------------------------------------------------ A: check + emit claims
// Claims are a partitioned stream, not atomics on the cells.  A candidate (no conflict with the state it saw) emits
// one tuple (position, wanted value, list slot) per position it saw untagged.  The tuples are hash-partitioned by position
// into KMX_CL_BINS bins per list: inside a workgroup they are counted per bin in LDS, each run is reserved with ONE global
// atomic per (workgroup, bin), the tuples are sorted by bin in LDS and written out run by run -- LDS-staged write combining:
// the memory system sees 8-byte stores that fill lines instead of one memory-side atomic per claimed position.
// k_round_detect then finds, bin by bin in an LDS table, the positions wanted with both values.
//
// The state a check sees may be STALE by exactly one visit: the winners of the previous visit to the same array (round
// r-1: list i+1) commit in the very launch that checks round r (k_round_commit_check: memory-side atomics and gathers side
// by side).  That is sound because the state is monotone (kmodel.hpp:604-618 never clears a bit): a conflict seen is final,
// and the verdict of a candidate can only be changed by tags it did not see yet -- exactly the claim tuples of those
// winners, which k_round_detect enters into its tables as SETTLED positions before it looks at this round's claims.
// For that the identity of a position inside a table must be exact: bin and fingerprint are the two halves of ONE
// bijective mix of the position (36 bits: arrays of up to 2^36 positions; above that the host commits before it checks).
// A tuple carries the position in its MIXED form (cl_mix below: bin = top 8 bits, fingerprint = the rest), computed once by
// the producer: k_round_detect spent half its time re-deriving it (two 64-bit multiplies per tuple and phase, 10^7 per launch).
#define CL_POS_BITS 44
#define CL_TUPLE(mixed, want, x) ((u64)(mixed) | ((u64)(want) << CL_POS_BITS) | ((u64)(x) << (CL_POS_BITS + 1)))
#define CL_MIXED(tp) ((tp) & ((1ULL << CL_POS_BITS) - 1))
#define CL_WANT(tp) ((u32)((tp) >> CL_POS_BITS) & 1u)
#define CL_X(tp) ((u32)((tp) >> (CL_POS_BITS + 1)) & (KMX_BUCKET - 1))
#define CL_MIX_BITS KMX_CL_MIX_BITS
// position-range partition (range_kernels.h): a claim names the received triple it came from instead of a list slot -- 27 bits, 8 above
// the 36 mixed bits and 19 above the wanted value (k_round_detect<..., RANGE> answers in that triple's verdict byte)
#define CL_RANGE_TUPLE(mixed, want, q) ((u64)(mixed) | ((u64)((q) & 0xFFu) << CL_MIX_BITS) | ((u64)(want) << CL_POS_BITS) | ((u64)((q) >> 8) << (CL_POS_BITS + 1)))
#define CL_RQ(tp) (((u32)((tp) >> CL_MIX_BITS) & 0xFFu) | ((u32)((tp) >> (CL_POS_BITS + 1)) << 8))
static_assert(KMX_RANGE_QBITS == 8 + 19 && CL_MIX_BITS + 8 <= CL_POS_BITS, "8 bits above the mixed position + the 19 bits above the wanted value");
#define CL_FP_BITS (CL_MIX_BITS - 8)
static_assert(KMX_CL_BINS_LOG2(8) == 8 && KMX_CL_BINS_LOG2(16) == 8 && CL_FP_BITS + 4 == 32, "bin (8 bits) + fingerprint (28 bits) = the mixed position; an entry = fingerprint + 4 flags");
// multiply by an odd constant and xor-shift right are bijections on 36-bit integers: distinct positions below 2^36 get
// distinct (bin, fingerprint) pairs
__device__ __forceinline__ u64 cl_mix(u64 pos)
{
	constexpr u64 MASK = (1ULL << CL_MIX_BITS) - 1;
	u64 m = (pos * 0x9E3779B97F4A7C15ULL) & MASK;
	m ^= m >> 19;
	m = (m * 0xD6E8FEB86659FD93ULL) & MASK;
	m ^= m >> 18;
	return m;
}
__device__ __forceinline__ u32 cl_bin(u64 m) { return (u32)(m >> CL_FP_BITS); }
__device__ __forceinline__ u32 cl_fp(u64 m) { return (u32)m & ((1u << CL_FP_BITS) - 1); }
// position & 15 <-> bit of the cell (device_common.h bit_in_cell)
__device__ __forceinline__ u32 bit_of_nibble(u32 nib) { return 8 * ((nib >> 3) & 1) + 7 - (nib & 7); }

// Per-slot record of a candidate: the cell index of each of its nh positions, then position & 15 of hash j in nibble j --
// whole 16-byte groups, so that a wave writes (and the commit reads) full lines: 32 bytes per slot at nh = 7.
__host__ __device__ __forceinline__ int crec_words(int nh) { return (nh + (nh <= 8 ? 1 : 2) + 3) & ~3; }
template <int NHM> struct CRec { u32 w[(NHM + (NHM <= 8 ? 1 : 2) + 3) & ~3]; };
template <int NHM> __device__ __forceinline__ void crec_store(u32 *dst, int nh, const u64 *pos)
{
	CRec<NHM> r;
	constexpr int WMAX = sizeof(r.w) / 4;
#pragma unroll
	for (int q = 0; q < WMAX; q++) r.w[q] = 0;
	u64 nib = 0;
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if (j < nh) { nib |= (pos[j] & 15) << (4 * j); }
	// (nh is uniform: the selects below are scalar)
#pragma unroll
	for (int q = 0; q < WMAX; q++) {
		u32 v = 0;
		if (q < NHM && q < nh) v = (u32)(pos[q < NHM ? q : 0] >> 4);
		else if (q == nh) v = (u32)nib;
		else if (q == nh + 1 && nh > 8) v = (u32)(nib >> 32);
		r.w[q] = v;
	}
	const int words = crec_words(nh);
#pragma unroll
	for (int q = 0; q < WMAX; q += 4)
		if (q < words) *(uint4 *)(dst + q) = make_uint4(r.w[q], r.w[q + 1], r.w[q + 2], r.w[q + 3]);
}
template <int NHM> __device__ __forceinline__ CRec<NHM> crec_load(const u32 *src, int nh)
{
	CRec<NHM> r;
	constexpr int WMAX = sizeof(r.w) / 4;
	const int words = crec_words(nh);
#pragma unroll
	for (int q = 0; q < WMAX; q += 4) {
		uint4 v = make_uint4(0, 0, 0, 0);
		if (q < words) v = *(const uint4 *)(src + q);
		r.w[q] = v.x; r.w[q + 1] = v.y; r.w[q + 2] = v.z; r.w[q + 3] = v.w;
	}
	return r;
}
// hash j of a record: cell index / position & 15
template <int NHM> __device__ __forceinline__ u32 crec_cell(const CRec<NHM> &r, int j) { return r.w[j]; }
template <int NHM> __device__ __forceinline__ u32 crec_nib(const CRec<NHM> &r, int nh, int j)
{
	constexpr int WMAX = sizeof(r.w) / 4;
	u64 nib = 0;
#pragma unroll
	for (int q = 0; q < WMAX; q++) {
		if (q == nh) nib |= r.w[q];
		if (q == nh + 1 && nh > 8) nib |= (u64)r.w[q] << 32;
	}
	return (u32)(nib >> (4 * j)) & 15u;
}

// (bx of gx workgroups work on list i: the kernels below map their grids onto these bodies)
// LDS of one check_emit workgroup, carved out of a byte pool the kernel owns (kernels that run one of several bodies per workgroup)
#define CHECK_LDS_BYTES(NHM) (256 * (NHM) * 8 + (3 * KMX_CL_BINS(NHM) + 8) * 4)
// COUNT: also count the random loads the launch issues (ST_PIPE_GATHERS) -- only the profiled leg of bench.py asks for it:
// the three ballots per wave and the live `stages` cost the fused launches 8 % (same-box A/B, profiles/r04_*).
template <int W, int NHM, bool COUNT = false> __device__ __forceinline__ void check_emit_body(const ModelDev &md, const BlockDev &bd, int t, int pp, int i, int bx, int gx, unsigned char *lds, int stat_slot)
{
	constexpr int NBIN = KMX_CL_BINS(NHM);
	u64 *s_tup = (u64 *)lds;                                         // [256 * NHM]
	int *s_cnt = (int *)(lds + 256 * NHM * 8), *s_off = s_cnt + NBIN, *s_base = s_off + NBIN, *s_tmp = s_base + NBIN;
	int &s_fail = s_tmp[4], &s_gath = s_tmp[5];
	const int n = bd.n[pp][i];
	const u64 row = (u64)i * KMX_BUCKET;
	const int a = (i + t) % md.nb;                                  // kmodel.hpp:563
	if (bx == 0 && threadIdx.x == 0) {
		if (n) atomicAdd(bd.stats + ST_ATTEMPTS, (u64)n);
		if (COUNT && stat_slot && n) atomicAdd(bd.stats + stat_slot, (u64)n);   // accounting: attempts examined inside fused launches
		for (int s = 0; s < KMX_NSLOW; s++) bd.Un[UN_IDX(s, i, md.nb)] = 0;   // k_round_file files this round's records
	}
	if (COUNT && threadIdx.x == 0) s_gath = 0;                       // (the loop's first barrier orders this)
	constexpr int CAP = KMX_CL_CAP_OF(NHM);
	u64 *tup = bd.cl_tup[pp] + (u64)i * NBIN * CAP;
	int *gcnt = bd.cl_cnt[pp] + i * KMX_CL_MAXBINS;
	unsigned char *status = bd.status[pp] + row;
	// grid-stride over the list: later rounds are launched with fewer workgroups (lists shrink round by round)
	for (int base = bx * 256; base < n; base += gx * 256) {
		if (threadIdx.x == 0) s_fail = 0;
		if (threadIdx.x < NBIN) s_cnt[threadIdx.x] = 0;
		__syncthreads();
		const int x = base + threadIdx.x;
		bool failed = false;
		u32 um = 0, bin = 0;
		int rank[NHM], stages = 0;
		Touches<NHM> tc;
		if (x < n) {
			const u32 raw = bd.list[pp][row + x];
			const u32 idx = (raw & LIST_HOLE) ? bd.mover[pp][row + (raw & ~LIST_HOLE)] : raw;
			if (raw & LIST_HOLE) bd.list[pp][row + x] = idx;            // later kernels of the round read plain entries
			u64 v[W];
			load_kmer<W>(bd.kmers, row + idx, v);
			bin = md.bin_of_occ[bd.counts[row + idx]];
			Premixed<W> pm = premix_string<W>(left_align<W>(v, md.k), md.gfull);
			failed = gather_touches_staged<W, NHM>(md, pm, a, bin, tc, stages);
			status[x] = failed ? SLOT_FAILED : SLOT_UNDECIDED;
			if (!failed) {
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh) {
						const u32 b = bit_in_cell(tc.pos[j]);
						if (!((tc.cell[j] >> (16 + b)) & 1u)) {
							um |= 1u << j;
							rank[j] = atomicAdd(&s_cnt[cl_bin(cl_mix(tc.pos[j]))], 1);
						}
					}
				bd.uw[pp][row + x] = um | (bin << 16);
				crec_store<NHM>(bd.crec[pp] + (row + x) * (u64)crec_words(md.nh), md.nh, tc.pos);   // what the deferred commit needs: no k-mer, no hash
			}
		}
		// survivors are counted per 1024-slot tile as they fail, so the reorder needs no counting pass
		const u64 mask = __ballot(failed);
		if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&s_fail, (int)__popcll(mask));
		if (COUNT && stat_slot) {                                        // random 4-byte loads this wave issued (the staged fetch stops early)
			const int g1 = (int)__popcll(__ballot(stages >= 1)), g2 = (int)__popcll(__ballot(stages >= 2)), g3 = (int)__popcll(__ballot(stages >= 3));
			if ((threadIdx.x & 63) == 0 && g1) atomicAdd(&s_gath, g1 * md.nh_first + g2 * (md.nh_second - md.nh_first) + g3 * (md.nh - md.nh_second));
		}
		__syncthreads();
		if (threadIdx.x == 0 && s_fail) atomicAdd(bd.tile_cnt[pp] + i * KMX_NTILES + (base >> 10), s_fail);
		// one run per bin: its offset in the LDS staging area (scan) and its place in the bin (ONE global atomic per run)
		int total;
		{
			const int c = threadIdx.x < NBIN ? s_cnt[threadIdx.x] : 0;
			const int ex = block_excl_scan_256(c, s_tmp, &total);
			if (threadIdx.x < NBIN) {
				s_off[threadIdx.x] = ex;
				s_base[threadIdx.x] = c ? atomicAdd(gcnt + threadIdx.x, c) : 0;
			}
		}
		__syncthreads();
		if (um) {
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((um >> j) & 1u)) {
					const u64 mx = cl_mix(tc.pos[j]);
					s_tup[s_off[cl_bin(mx)] + rank[j]] = CL_TUPLE(mx, (bin >> j) & 1u, x);
				}
		}
		__syncthreads();
		for (int q = threadIdx.x; q < total; q += 256) {               // consecutive lanes, consecutive tuples of a run
			const u64 tp = s_tup[q];
			const u32 b = cl_bin(CL_MIXED(tp));
			const int g = s_base[b] + (q - s_off[b]);
			if (g < CAP) tup[(u64)b * CAP + g] = tp;
			else bd.cl_ovf[i] = 1;                                     // the whole list takes the ordered path this round
		}
		__syncthreads();
	}
	if (COUNT && stat_slot && threadIdx.x == 0 && s_gath) atomicAdd(bd.stats + ST_PIPE_GATHERS, (u64)s_gath);
}
template <int W, int NHM> __global__ __launch_bounds__(256) void k_round_check_emit(ModelDev md, BlockDev bd, int t, int pp)
{
	__shared__ __align__(16) unsigned char lds[CHECK_LDS_BYTES(NHM)];
	check_emit_body<W, NHM>(md, bd, t, pp, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, lds, 0);
}

// ------------------------------------------------------------------------------------------ D: settled positions and opposite claims, bin by bin
// One workgroup per (bin, list).  An LDS table entry = fingerprint of a position (exact within the bin, see cl_mix) + 4 flags:
//   bit 0 / bit 1   value 0 / value 1 is wanted there by a candidate of this round
//   bit 2 / bit 3   SETTLED with value 0 / 1: a winner of the previous visit to this array tagged the position after (or
//                   while) this round's check looked at it
// Phase 1: this round's tuples enter the table (at most KMX_CL_CAP <= 3/4 of it: a free slot always exists).
// Phase 2 (use_delta): the claim tuples of the previous round's list i+1 -- the list that visited this array -- only PROBE the
//   table: nearly all of them (99.7 %) name a position nobody claims this round and are done after one LDS read; for a hit
//   the slot's status is looked up, and an uncontended winner (status 0: its commit rides with this round's check) settles the entry.
// Phase 3: every tuple of this round looks its position up: settled with the other value -> the candidate has failed after
//   all (dfail); wanted with both values by candidates -> those candidates take the ordered path (SLOT_CONTENDED).  A settled
//   position causes no contention: everybody who wants the other value there fails, the others are compatible with it.
#define DT_SETTLED(v) (4u << (v))
__device__ __forceinline__ void dt_insert(u32 *s_t, u32 tmask, u32 fp, u32 w)
{
	u32 slot = fp & tmask;                                           // (the low bits of cl_mix are as good as any)
	for (;;) {                                                       // a free slot always exists: 2^TBITS > 4/3 * capacity
		const u32 old = atomicCAS(&s_t[slot], 0u, (fp << 4) | w);
		if (old == 0) return;
		if ((old >> 4) == fp) { if (!(old & w)) atomicOr(&s_t[slot], w); return; }
		slot = (slot + 1) & tmask;
	}
}
// slot of the entry of fp, or -1
__device__ __forceinline__ int dt_find(const u32 *s_t, u32 tmask, u32 fp)
{
	u32 slot = fp & tmask, cur;
	while ((cur = s_t[slot]) != 0) {
		if ((cur >> 4) == fp) return (int)slot;
		slot = (slot + 1) & tmask;
	}
	return -1;
}
__device__ __forceinline__ void dt_settle(u32 *s_t, u32 tmask, u64 d, const unsigned char *dstatus)
{
	const int slot = dt_find(s_t, tmask, cl_fp(CL_MIXED(d)));
	if (slot >= 0 && dstatus[CL_X(d)] == SLOT_UNDECIDED) atomicOr(&s_t[slot], DT_SETTLED(CL_WANT(d)));
}
// RANGE (the owner of a position range, range_kernels.h): no settled positions there, and the contention is reported per CLAIM --
// in the verdict byte of the triple the claim came from (it was 2 = untagged; 2 | 4 = untagged, wanted with both values) --
// because the list rank orders its contended candidates on the both-wanted positions alone
template <bool RANGE = false>
__device__ __forceinline__ void dt_lookup(const u32 *s_t, u32 tmask, u64 e, unsigned char *status, unsigned char *dfail, unsigned char *rverdict = nullptr)
{
	const u32 fp = cl_fp(CL_MIXED(e));
	u32 slot = fp & tmask, cur;
	while (((cur = s_t[slot]) >> 4) != fp) slot = (slot + 1) & tmask;  // (it was inserted in phase 1)
	const u32 w = CL_WANT(e);
	if (RANGE) { if ((cur >> (w ^ 1u)) & 1u) rverdict[CL_RQ(e)] = 2 | 4; return; }
	if (cur & 12u) { if (w != ((cur >> 3) & 1u)) dfail[CL_X(e)] = 1; }             // tagged meanwhile with the other value (two winners never settle
	else if ((cur >> (w ^ 1u)) & 1u) status[CL_X(e)] = SLOT_CONTENDED;             // different values; one that hits it twice leaves the OR: 1)
}

// keep_own: this round's winners will be committed beside the next round's check, whose k_round_detect reads (and resets) the bins.
// TBITS: log2 of the table.  The full-size form (KMX_CL_TBITS: 64 KB at nh <= 8, two 1024-thread workgroups per CU) takes every
// tuple of a full bin.  The SMALL form (KMX_CL_TBITS_SMALL, 256 threads, 16 KB: every bin of every list resident at once) is
// for the late rounds, whose bins hold a few hundred tuples: a launch of it is one generation of workgroups instead of three.
// A bin that does not fit a small table raises cl_ovf[i] -- the whole list takes the ordered path, exact like any other
// overflow -- and reports its fill (`late`: ST_MAX_LATE_BIN), from which the host decides which form the late rounds get.
template <int NHM, int BT, int TBITS, bool RANGE = false> __global__ __launch_bounds__(BT) __attribute__((amdgpu_waves_per_eu(NHM <= 8 ? 8 : 4)))
void k_round_detect(BlockDev bd, int nb, int pp, int use_delta, int keep_own, int late)
{
	constexpr int NBIN = KMX_CL_BINS(NHM), T = 1 << TBITS, CAP = KMX_CL_CAP_OF(NHM), TCAP = T / 4 * 3;
	static_assert(TBITS < KMX_CL_TBITS(NHM) || CAP <= TCAP, "the full-size table of a bin must take every tuple of a full bin with room to spare");
	__shared__ u32 s_t[T];                                           // exactly 64 KB at nh <= 8 (full size): two workgroups per CU
	const int i = (int)blockIdx.y, b = blockIdx.x;
	const int id = (i + 1) % nb;                                     // list that visited this array one round earlier
	int *gc = bd.cl_cnt[pp] + (i * KMX_CL_MAXBINS + b) * (RANGE ? KMX_CTR_STRIDE : 1);   // (RANGE: one counter per 128-byte line, k_range_verdict)
	int *gd = bd.cl_cnt[pp ^ 1] + id * KMX_CL_MAXBINS + b;
	int cnt = *gc, dcnt = RANGE ? 0 : *gd;
	if (late && threadIdx.x == 0 && cnt > 1024) atomicMax(bd.stats + ST_MAX_LATE_BIN, (u64)cnt);
	if (cnt > CAP) cnt = CAP;                                        // check_emit has raised cl_ovf[i]
	if (cnt > TCAP) {                                                // (small form only) the bin does not fit: the list takes the ordered path
		if (threadIdx.x == 0) bd.cl_ovf[i] = 1;
		cnt = 0;
	}
	if (dcnt > CAP) dcnt = CAP;                                      // (that list had no uncontended winner then: nothing of it passes the filter)
	if (!use_delta) dcnt = 0;
	if (cnt) {                                                       // uniform
		const u64 *tp = bd.cl_tup[pp] + ((u64)i * NBIN + b) * CAP;
		const u64 *dp = bd.cl_tup[pp ^ 1] + ((u64)id * NBIN + b) * CAP;
		const unsigned char *dstatus = bd.status[pp ^ 1] + (u64)id * KMX_BUCKET;
		unsigned char *status = bd.status[pp] + (u64)i * KMX_BUCKET, *dfail = bd.dfail + (u64)i * KMX_BUCKET;
		unsigned char *rverdict = RANGE ? bd.rverdict : nullptr;
		int tb = TBITS < 10 ? TBITS : 10;
		while ((1 << tb) < 4 * cnt && tb < TBITS) tb++;              // load <= 1/4 (<= 3/4 for a full bin): short probe chains
		const u32 tmask = (1u << tb) - 1;
		for (int q = threadIdx.x; q < (1 << tb); q += BT) s_t[q] = 0;
		constexpr int U = 8;                                         // tuples per thread in flight: the loads of a batch are issued together
		if (cnt <= U * BT && dcnt <= U * BT / 2) {                   // the usual case: the bin's tuples stay in registers between the phases
			u64 e[U], d[U / 2];
#pragma unroll
			for (int u = 0; u < U; u++) { const int q = u * BT + (int)threadIdx.x; e[u] = q < cnt ? tp[q] : ~0ULL; }
#pragma unroll
			for (int u = 0; u < U / 2; u++) { const int q = u * BT + (int)threadIdx.x; d[u] = q < dcnt ? dp[q] : ~0ULL; }
			__syncthreads();
#pragma unroll
			for (int u = 0; u < U; u++) if (e[u] != ~0ULL) dt_insert(s_t, tmask, cl_fp(CL_MIXED(e[u])), 1u << CL_WANT(e[u]));
			__syncthreads();
			if (dcnt) {                                              // (uniform)
#pragma unroll
				for (int u = 0; u < U / 2; u++) if (d[u] != ~0ULL) dt_settle(s_t, tmask, d[u], dstatus);
				__syncthreads();
			}
#pragma unroll
			for (int u = 0; u < U; u++) if (e[u] != ~0ULL) dt_lookup<RANGE>(s_t, tmask, e[u], status, dfail, rverdict);
		} else {
			__syncthreads();
			for (int q = threadIdx.x; q < cnt; q += BT) { const u64 e = tp[q]; dt_insert(s_t, tmask, cl_fp(CL_MIXED(e)), 1u << CL_WANT(e)); }
			__syncthreads();
			for (int q = threadIdx.x; q < dcnt; q += BT) dt_settle(s_t, tmask, dp[q], dstatus);
			__syncthreads();
			for (int q = threadIdx.x; q < cnt; q += BT) dt_lookup<RANGE>(s_t, tmask, tp[q], status, dfail, rverdict);
		}
	}
	// (every thread has read the counters: barriers above, or nothing else happened)
	if (threadIdx.x == 0) {
		if (!RANGE) *gd = 0;                                         // last reader of the previous round's bin: ready for round r+1
		if (!keep_own) *gc = 0;                                      // nobody will read this round's tuples again
	}
}

// ------------------------------------------------------------------------------------------ S: ordered slow path (helpers)
__device__ __forceinline__ u64 resv_key(u64 epoch, u32 x) { return (epoch << 20) | (u64)(0xFFFFFu - x); }
__device__ __forceinline__ u64 *resv_slot(const BlockDev &bd, int i, u64 pos) { return bd.R + (u64)i * KMX_RSIZE + (pos & (KMX_RSIZE - 1)); }
__device__ __forceinline__ void mark_failed(const BlockDev &bd, int pp, int i, u64 row, u32 x)
{
	bd.status[pp][row + x] = SLOT_FAILED;
	atomicAdd(bd.tile_cnt[pp] + i * KMX_NTILES + (x >> 10), 1);      // contended k-mers only: rare
}
template <int NHM> __device__ __forceinline__ void reserve_untagged(const ModelDev &md, const BlockDev &bd, int i, const Touches<NHM> &tc, u64 key)
{
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if (j < md.nh) {
			u32 b = bit_in_cell(tc.pos[j]);
			if (!((tc.cell[j] >> (16 + b)) & 1u)) atomicMax(resv_slot(bd, i, tc.pos[j]), key);
		}
}
// A k-mer owns its outcome when it does not conflict with what is committed by now and holds the reservation of every
// position that is still untagged.  (A position another k-mer of this pass has tagged meanwhile was won by a smaller
// index -- it held our common slot -- so it is judged like committed state.)
template <int NHM, bool COHERENT>
__device__ __forceinline__ bool owns_outcome(const ModelDev &md, const BlockDev &bd, int i, const Touches<NHM> &tc, u32 bin, u64 key)
{
	bool mine = !touches_conflict<NHM>(md, tc, bin);
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if (j < md.nh) {
			u32 b = bit_in_cell(tc.pos[j]);
			if (!((tc.cell[j] >> (16 + b)) & 1u)) {
				const u64 r = COHERENT ? cell_load_coherent(resv_slot(bd, i, tc.pos[j])) : *resv_slot(bd, i, tc.pos[j]);
				mine &= (r == key);
			}
		}
	return mine;
}

// record of a contended k-mer: list index | REC_WON | bin << 32 | (positions that were untagged when check_emit looked) << 48
#define REC_WON (1ULL << 31)       // set by the finisher: decided to fit, k_reorder applies it
template <int W> __device__ __forceinline__ void rec_store(u64 *rec, u64 slot, u32 x, u32 bin, const u64 *v, u32 untagged = 0)
{
	u64 *r = rec + slot * (1 + W);
	r[0] = (u64)x | ((u64)bin << 32) | ((u64)untagged << 48);
#pragma unroll
	for (int w = 0; w < W; w++) r[1 + w] = v[w];
}
template <int W> __device__ __forceinline__ u32 rec_load(const u64 *rec, u64 slot, u32 &x, u32 &bin, u64 *v)
{
	const u64 *r = rec + slot * (1 + W);
	const u64 h = r[0];
	x = (u32)h & (KMX_BUCKET - 1);
	bin = (u32)(h >> 32) & 0xFFFFu;
#pragma unroll
	for (int w = 0; w < W; w++) v[w] = r[1 + w];
	return (u32)(h >> 48);
}

// ------------------------------------------------------------------------------------------ F: file the contended candidates
// After k_round_detect every slot of the round has its verdict spread over two bytes: status (undecided / failed /
// contended) and dfail.  A dfail slot becomes a failure (counted for the reorder like the ones check_emit found); a contended
// one -- every candidate of a list whose claims overflowed a bin -- files a record in U[0] (slot, bin, packed k-mer, untagged
// mask of the check) and places its priority reservations (epoch `epoch`), which saves the first reserve pass of the
// ordered slow path.  What stays SLOT_UNDECIDED is a winner: nobody wants the other value on any of its positions.
// A scan with few, fat workgroups (a launch pays ~6 ns of dispatch per workgroup, and same-address atomics on the
// record counter serialise at ~11 ns each): a workgroup takes chunks of 4096 slots, 4 consecutive ones per thread as one
// 32-bit word of status and of dfail, collects the contended slots of the chunk in an LDS queue, reserves their records
// with ONE global atomic, and files them.
#define KMX_FILE_WGS 64                        // one chunk per workgroup at full length
template <int W, int NHM> __global__ __launch_bounds__(1024) void k_round_file(ModelDev md, BlockDev bd, int pp, u64 epoch)
{
	__shared__ u32 s_q[4096];
	__shared__ int s_nq, s_base, s_tf[4], s_df;
	const int i = blockIdx.y;
	const int n = bd.n[pp][i];
	const u64 row = (u64)i * KMX_BUCKET;
	const bool all_contended = bd.cl_ovf[i] != 0;
	u32 *status32 = (u32 *)(bd.status[pp] + row), *dfail32 = (u32 *)(bd.dfail + row);
	if (threadIdx.x == 0) { s_df = 0; s_nq = 0; }
	if (threadIdx.x < 4) s_tf[threadIdx.x] = 0;
	__syncthreads();
	for (int base = (int)blockIdx.x * 4096; base < n; base += KMX_FILE_WGS * 4096) {     // uniform trip count
		const int x0 = base + 4 * (int)threadIdx.x;
		if (x0 < n) {
			const u32 st = status32[x0 >> 2], df = dfail32[x0 >> 2];
			u32 st_new = st;
			int nfail = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				if (x0 + k >= n) break;
				const u32 sk = (st >> (8 * k)) & 0xFFu;
				if (sk == SLOT_FAILED) continue;
				if ((df >> (8 * k)) & 0xFFu) {                           // (only candidates emit claims, so only they can be flagged)
					st_new = (st_new & ~(0xFFu << (8 * k))) | ((u32)SLOT_FAILED << (8 * k));
					nfail++;
				} else if (sk == SLOT_CONTENDED || (sk == SLOT_UNDECIDED && all_contended)) {
					st_new &= ~(0xFFu << (8 * k));                         // SLOT_UNDECIDED: the ordered path decides it
					s_q[atomicAdd(&s_nq, 1)] = (u32)(x0 + k);
				}
			}
			if (st_new != st) status32[x0 >> 2] = st_new;
			if (df) dfail32[x0 >> 2] = 0;
			if (nfail) atomicAdd(&s_tf[threadIdx.x >> 8], nfail);       // survivors are counted per 1024-slot tile (k_reorder)
		}
		__syncthreads();
		const int nq = s_nq;
		if (threadIdx.x == 0 && nq) s_base = atomicAdd(bd.Un + UN_IDX(0, i, md.nb), nq);
		if (threadIdx.x < 4 && s_tf[threadIdx.x]) {
			atomicAdd(bd.tile_cnt[pp] + i * KMX_NTILES + (base >> 10) + (int)threadIdx.x, s_tf[threadIdx.x]);
			atomicAdd(&s_df, s_tf[threadIdx.x]);
			s_tf[threadIdx.x] = 0;
		}
		__syncthreads();
		for (int e = threadIdx.x; e < nq; e += 1024) {
			const u32 x = s_q[e];
			const u32 idx = bd.list[pp][row + x];
			u64 v[W];
			load_kmer<W>(bd.kmers, row + idx, v);
			const u32 uw = bd.uw[pp][row + x], um = uw & 0xFFFFu, bin = uw >> 16;
			const CRec<NHM> rec = crec_load<NHM>(bd.crec[pp] + (row + x) * (u64)crec_words(md.nh), md.nh);
			const u64 key = resv_key(epoch, x);
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((um >> j) & 1u)) {
					const u64 pos = ((u64)crec_cell<NHM>(rec, j) << 4) | crec_nib<NHM>(rec, md.nh, j);
					atomicMax(resv_slot(bd, i, pos), key);
				}
			rec_store<W>(bd.Urec[0], row + (u64)(s_base + e), x, bin, v, um);
		}
		__syncthreads();
		if (threadIdx.x == 0) s_nq = 0;
		__syncthreads();
	}
	if (threadIdx.x == 0 && s_df) atomicAdd(bd.stats + ST_DELTA_FAILS, (u64)s_df);
}

// ------------------------------------------------------------------------------------------ B: commit
// The winners k_round_file left undecided cannot interact with any other candidate of their list (no position of theirs
// is wanted with the other value), so they commit in parallel: one atomic OR per position they saw untagged (tag + value
// in one word, kmodel.hpp:611-618) -- straight from what check_emit stored per slot (cell indices, low position bits,
// wanted values): no k-mer, no hash, no second look at the cells.  Deferred by one round: the commit of round r rides with
// the check of round r+1 (k_round_commit_check), whose detect treats these positions as settled.
// (a = the array list i visited in the round being committed; pp = that round's parity)
template <int NHM, bool COUNT = false> __device__ __forceinline__ void commit_body(const ModelDev &md, const BlockDev &bd, int a, int pp, int i, int bx, int gx, unsigned char *lds, int stat_slot)
{
	int &s_cnt = ((int *)lds)[0], &s_atom = ((int *)lds)[1];
	const int n = bd.n[pp][i];
	const u64 row = (u64)i * KMX_BUCKET;
	cell_t *cells = md.cells[a];
	const unsigned char *status = bd.status[pp] + row;
	if (COUNT && stat_slot && threadIdx.x == 0) { s_cnt = 0; s_atom = 0; }
	if (COUNT && stat_slot) __syncthreads();
	for (int base = bx * 256; base < n; base += gx * 256) {
		const int x = base + threadIdx.x;
		const bool win = x < n && status[x] == SLOT_UNDECIDED;
		u32 um = 0;
		if (win) {
			const u32 uw = bd.uw[pp][row + x], want = uw >> 16;
			um = uw & 0xFFFFu;
			const CRec<NHM> rec = crec_load<NHM>(bd.crec[pp] + (row + x) * (u64)crec_words(md.nh), md.nh);
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((um >> j) & 1u)) {
					const u32 b = bit_of_nibble(crec_nib<NHM>(rec, md.nh, j));
					atomicOr(cells + crec_cell<NHM>(rec, j), CELL_TAG(b) | (((want >> j) & 1u) ? CELL_VAL(b) : 0u));
				}
		}
		if (COUNT && stat_slot) {
			const u64 wm = __ballot(win);
			if (wm) {                                                    // (uniform per wave) winners; atomics issued: one per untagged position of a winner
				int at = 0;
#pragma unroll
				for (int j = 0; j < NHM; j++) at += (int)__popcll(__ballot((um >> j) & 1u));
				if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt, (int)__popcll(wm)); atomicAdd(&s_atom, at); }
			}
		}
	}
	if (COUNT && stat_slot) {
		__syncthreads();
		if (threadIdx.x == 0 && s_cnt) { atomicAdd(bd.stats + stat_slot, (u64)s_cnt); atomicAdd(bd.stats + ST_PIPE_ATOMICS, (u64)s_atom); }
	}
}
// commit of the round with parity pp, whose list i visited array (i + t) % nb
template <int NHM> __global__ __launch_bounds__(256) void k_round_commit(ModelDev md, BlockDev bd, int t, int pp)
{
	__shared__ __align__(16) unsigned char lds[16];
	commit_body<NHM>(md, bd, ((int)blockIdx.y + t) % md.nb, pp, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, lds, 0);
}

// ------------------------------------------------------------------------------------------ B|A: the commit of round r-1 beside the check of round r
// The check is bound by random 4-byte gathers, the commit by memory-side atomics; side by side the two take ~3/4 of what
// they take one after the other (tools/microbench_pair.py: 4.8 ms + 9.9 ms alone, 10.9 ms together).  Round 2 overlapped
// two groups of lists inside a round; now the whole previous round commits under the whole check: list i is examined on array
// (i + t) % nb while list i + 1 commits on that very array -- what the check misses of it, k_round_detect supplies.
// t: the round being checked (parity pp); the round being committed is the one before it (parity pp ^ 1; across a block
// boundary: the last round of the previous block), whose list i visited array (i + t + nb - 1) % nb.
// Consecutive workgroups alternate between the two kinds and cycle through the lists, so every list advances at the same
// pace, both kinds are resident on every CU from the first wave of workgroups to the last, and the launch does not end on a
// tail of atomics.
// gk / gc workgroups per list check / commit (powers of two; the committed lists are about twice as long as the checked
// ones -- or, across a block boundary, 16 times shorter -- and every workgroup that is launched costs dispatch time).
template <int W, int NHM, bool COUNT> __global__ __launch_bounds__(256) void k_round_commit_check(ModelDev md, BlockDev bd, int t, int pp, int gk, int gc)
{
	__shared__ __align__(16) unsigned char lds[CHECK_LDS_BYTES(NHM)];
	const int nb = md.nb, G = gk < gc ? gk : gc, rk = gk / G, rc = gc / G, per = nb * (rk + rc);
	const int p = (int)blockIdx.x / per, r = (int)blockIdx.x % per, l = r % nb, s = r / nb;
	if (s < rk) check_emit_body<W, NHM, COUNT>(md, bd, t, pp, l, p * rk + s, gk, lds, ST_PIPE_ATTEMPTS);
	else commit_body<NHM, COUNT>(md, bd, (l + t + nb - 1) % nb, pp ^ 1, l, p * rc + (s - rk), gc, lds, ST_PIPE_SUCC);
}

// ------------------------------------------------------------------------------------------ S: ordered slow path
// Grid-stride over the records of level s (the set is ~0.5 % of a round at 10^8 k-mers, most of it in round 0).
#define SLOW_BLOCKS 64
template <int W, int NHM> __global__ __launch_bounds__(256) void k_slow_reserve(ModelDev md, BlockDev bd, int t, int pp, int s, u64 epoch)
{
	const int i = blockIdx.y, lv = s & 1;
	const int cnt = bd.Un[UN_IDX(lv, i, md.nb)];
	const u64 row = (u64)i * KMX_BUCKET;
	const int a = (i + t) % md.nb;
	if (blockIdx.x == 0 && threadIdx.x == 0) bd.Un[UN_IDX(lv ^ 1, i, md.nb)] = 0;   // the other level was consumed by the previous pass
	for (int u = blockIdx.x * 256 + threadIdx.x; u < cnt; u += SLOW_BLOCKS * 256) {
		u32 x, bin;
		u64 v[W];
		rec_load<W>(bd.Urec[lv], row + u, x, bin, v);
		Premixed<W> pm = premix_string<W>(left_align<W>(v, md.k), md.gfull);
		Touches<NHM> tc;
		gather_touches<W, NHM, false>(md, pm, a, tc);
		if (touches_conflict<NHM>(md, tc, bin)) { mark_failed(bd, pp, i, row, x); continue; }
		reserve_untagged<NHM>(md, bd, i, tc, resv_key(epoch, x));
	}
}

// resolve pass of level s: winners commit, the rest move to level s+1.  Level 0 holds k_round_file's reservations.
template <int W, int NHM> __device__ __forceinline__ void slow_resolve_body(const ModelDev &md, const BlockDev &bd, int t, int pp, int s, u64 epoch)
{
	__shared__ int s_cnt, s_base, s_succ;
	const int i = blockIdx.y, lv = s & 1;
	const int cnt = bd.Un[UN_IDX(lv, i, md.nb)];
	const u64 row = (u64)i * KMX_BUCKET;
	const int a = (i + t) % md.nb;
	if (threadIdx.x == 0) {
		s_succ = 0;
		if (s == 0 && blockIdx.x == 0 && cnt) { atomicAdd(bd.stats + ST_CONTENDED, (u64)cnt); atomicMax(bd.stats + ST_MAX_U0, (u64)cnt); }
	}
	for (int base = blockIdx.x * 256; base < cnt; base += SLOW_BLOCKS * 256) {      // uniform trip count per workgroup
		if (threadIdx.x == 0) s_cnt = 0;
		__syncthreads();
		const int u = base + threadIdx.x;
		bool mine = false, defer = false;
		u32 x = 0, bin = 0;
		u64 v[W];
		if (u < cnt) {
			rec_load<W>(bd.Urec[lv], row + u, x, bin, v);
			if (bd.status[pp][row + x] == SLOT_UNDECIDED) {
				Aligned<W> al = left_align<W>(v, md.k);
				Premixed<W> pm = premix_string<W>(al, md.gfull);
				Touches<NHM> tc;
				gather_touches<W, NHM, false>(md, pm, a, tc);
				mine = owns_outcome<NHM, false>(md, bd, i, tc, bin, resv_key(epoch, x));
				if (mine) {
					commit_touches<W, NHM>(md, tc, bin, a);
					bd.status[pp][row + x] = SLOT_INSERTED;
				}
				defer = !mine;
			}
		}
		const int p = block_append_slot(bd.Un + UN_IDX(lv ^ 1, i, md.nb), defer, &s_cnt, &s_base);
		if (defer) rec_store<W>(bd.Urec[lv ^ 1], row + p, x, bin, v);
		const u64 mk = __ballot(mine);
		if ((threadIdx.x & 63) == 0 && mk) atomicAdd(&s_succ, (int)__popcll(mk));
		__syncthreads();
	}
	__syncthreads();
	if (threadIdx.x == 0 && s_succ) atomicAdd(bd.stats + ST_SLOW_SUCC, (u64)s_succ);
}
template <int W, int NHM> __global__ __launch_bounds__(256) void k_slow_resolve(ModelDev md, BlockDev bd, int t, int pp, int s, u64 epoch)
{
	slow_resolve_body<W, NHM>(md, bd, t, pp, s, epoch);
}

// The untagged mask of the check is a sound basis for the ordered path as long as every claim of the list went through
// k_round_detect: a position in it is still untagged, or was tagged since by a still-uncommitted winner of the previous
// visit with the value this k-mer wants (anything else made it a dfail).  A list whose claims overflowed a bin lost tuples,
// so its masks are unchecked against those winners: it takes the gathering forms (the commits are visible by then).
__device__ __forceinline__ bool snapshot_ok(const BlockDev &bd, int i) { return bd.cl_ovf[i] == 0; }

// resolve pass of level 0 without a gather: the untagged mask check_emit left in the record is still right for
// this purpose (see finish_lds), so a k-mer that holds the reservation of every position in it commits, and one that
// does not is deferred -- also when the position was tagged meanwhile by a winner of this very pass, which held it.
template <int W, int NHM> __global__ __launch_bounds__(256) void k_slow_resolve0(ModelDev md, BlockDev bd, int t, int pp, u64 epoch)
{
	__shared__ int s_cnt, s_base, s_succ;
	const int i = blockIdx.y;
	if (!snapshot_ok(bd, i)) { slow_resolve_body<W, NHM>(md, bd, t, pp, 0, epoch); return; }
	const int cnt = bd.Un[UN_IDX(0, i, md.nb)];
	const u64 row = (u64)i * KMX_BUCKET;
	const int a = (i + t) % md.nb;
	cell_t *cells = md.cells[a];
	const int sbase = a * md.nh;
	if (threadIdx.x == 0) {
		s_succ = 0;
		if (blockIdx.x == 0 && cnt) { atomicAdd(bd.stats + ST_CONTENDED, (u64)cnt); atomicMax(bd.stats + ST_MAX_U0, (u64)cnt); }
	}
	for (int base = blockIdx.x * 256; base < cnt; base += SLOW_BLOCKS * 256) {      // uniform trip count per workgroup
		if (threadIdx.x == 0) s_cnt = 0;
		__syncthreads();
		const int u = base + threadIdx.x;
		bool mine = false, defer = false;
		u32 x = 0, bin = 0;
		u64 v[W];
		if (u < cnt) {
			const u32 um = rec_load<W>(bd.Urec[0], row + u, x, bin, v);
			Aligned<W> al = left_align<W>(v, md.k);
			Premixed<W> pm = premix_string<W>(al, md.gfull);
			const u64 key = resv_key(epoch, x);
			u64 pos[NHM], held[NHM];
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((um >> j) & 1u)) {
					pos[j] = mod_u64(murmur_seeded<W>(pm, md.gfull, c_seeds[(sbase + j) & 127]), md.km_mod);
					held[j] = *resv_slot(bd, i, pos[j]);
				}
			mine = true;
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((um >> j) & 1u)) mine &= held[j] == key;
			if (mine) {
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((um >> j) & 1u)) {
						const u32 b = bit_in_cell(pos[j]);
						atomicOr(cells + (pos[j] >> 4), CELL_TAG(b) | (value_at_position<NHM>(md, pos, um, bin, j) ? CELL_VAL(b) : 0u));
					}
				bd.status[pp][row + x] = SLOT_INSERTED;
			}
			defer = !mine;
		}
		const int p = block_append_slot(bd.Un + UN_IDX(1, i, md.nb), defer, &s_cnt, &s_base);
		if (defer) rec_store<W>(bd.Urec[1], row + p, x, bin, v);
		const u64 mk = __ballot(mine);
		if ((threadIdx.x & 63) == 0 && mk) atomicAdd(&s_succ, (int)__popcll(mk));
		__syncthreads();
	}
	__syncthreads();
	if (threadIdx.x == 0 && s_succ) atomicAdd(bd.stats + ST_SLOW_SUCC, (u64)s_succ);
}

// Finisher: ONE workgroup per list decides whatever is still undecided at level s, in list order.
//
// finish_lds -- at most KMX_FIN_RPT*1024 records at a time.  Every thread keeps its records (cell indices and bit
// numbers of their positions, mask of the positions still untagged) in registers; the reservations live in LDS.
// The state of the array is read ONCE -- the untagged mask check_emit left in the record when nothing was committed
// on contended positions since (snapshot), one round of coherent gathers otherwise -- and afterwards only this
// workgroup changes it, so everything else is learnt through LDS:
//   1. every undecided record reserves its untagged positions with atomicMax(priority << 13 | tag) in table 1
//      (priority: smaller list index first; tag: 13 position bits above the slot index);
//   2. a position whose table-1 slot went to ANOTHER position (tag mismatch -- all reservers of a position see the same
//      winner) is reserved again in table 2 under a second hash, which removes nearly all false sharing;
//   3. a record that holds every untagged position has no earlier undecided record on any of them: its outcome is the
//      sequential one.  It wins, and replaces each slot it holds by a MARK word carrying the full position identity
//      (tag + 9 more bits; the slot index gives the rest) and the value it commits there;
//   4. a record that lost a slot to a MARK of exactly its position now knows the position is tagged: with the other
//      value it has failed for good, with its own value the position leaves its untagged mask.  (Every undecided
//      record interested in a position reserves it in every iteration and takes the same route to table 1 or 2, so
//      nobody misses the iteration in which the position is won.)
//   5. the slots are cleared, and the loop ends when nothing is undecided; the smallest undecided index always wins.
// Sharing a slot with a foreign position only delays a record.  `defer`: the winners do not touch the array at all;
// they set REC_WON in their record and k_reorder -- 256 workgroups per list instead of one -- applies tag/value bits
// and the km_back insert.  One CU issues a random access every ~2.8 ns, which is what bounded this kernel before.
// KMX_FIN_LOG2 can be lowered at compile time (tools/stress_small_tables.py builds such a library) so that slot sharing,
// the table-2 route and MARK words of foreign positions are exercised constantly; the product uses 2^14 slots per table.
#ifndef KMX_FIN_LOG2
#define KMX_FIN_LOG2 14
#endif
#define KMX_FIN_T (1 << KMX_FIN_LOG2)
#define KMX_FIN_RIDER_KPT(NHM) ((NHM) <= 8 ? 2 : 1)            // k-mers per thread of a rider workgroup in the finisher launch (its LDS must fit the finisher's)
#define KMX_FIN_MAX_POS (1ULL << (KMX_FIN_LOG2 + 22))   // slot index + 22 identity bits name a position exactly below this
#define FIN_MARK 0x80000000u
// workgroup barrier for LDS-only hand-offs: __syncthreads() would also wait for every outstanding global store and
// atomic of the wave (status bytes, failure counters), 2-4 us each time
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ u32 fin_ident(u64 q) { return (u32)(q >> KMX_FIN_LOG2) & 0x3FFFFFu; }        // tag13 | hi9 << 13
__device__ __forceinline__ u32 fin_key(u32 x, u64 q) { return ((0x3FFFFu - x) << 13) | ((u32)(q >> KMX_FIN_LOG2) & 0x1FFFu); }
__device__ __forceinline__ u32 fin_slot1(u64 q) { return (u32)q & (KMX_FIN_T - 1); }
__device__ __forceinline__ u32 fin_slot2(u64 q) { return ((u32)q ^ ((fin_ident(q) * 0x9E3779B1u) >> (32 - KMX_FIN_LOG2))) & (KMX_FIN_T - 1); }

template <int W, int NHM, int RPT>
__device__ __forceinline__ u64 finish_lds(const ModelDev &md, const BlockDev &bd, int pp, int i, int a, int lv, int n, const u32 *s_list, bool snapshot, bool defer,
                                           u32 *s_t1, u32 *s_t2, int *s_pending, int *s_succ)
{
	// the n records are Urec[lv][s_list[0..n)], or Urec[lv][0..n) when s_list is null; the tables are empty on entry and on exit
	const u64 row = (u64)i * KMX_BUCKET;
	cell_t *cells = md.cells[a];
	u32 x[RPT], bin[RPT], rec[RPT], um[RPT], ghost[RPT], cidx[RPT][NHM];   // ghost: won, but these positions are not published yet
	u64 bits[RPT];                                                   // bit_in_cell of position j in nibble j
	bool live[RPT], won[RPT];
#pragma unroll
	for (int r = 0; r < RPT; r++) {
		const int slot = threadIdx.x + r * 1024;
		live[r] = slot < n;
		won[r] = false;
		x[r] = bin[r] = 0;
		bits[r] = 0;
		rec[r] = um[r] = ghost[r] = 0;
		if (live[r]) {
			rec[r] = s_list ? s_list[slot] : (u32)slot;
			u64 v[W];
			um[r] = rec_load<W>(bd.Urec[lv], row + rec[r], x[r], bin[r], v);    // every record of the finisher's level is undecided
			// the positions: what check_emit left for the slot (every record is a candidate of this round's check) -- no hashing
			const CRec<NHM> cr = crec_load<NHM>(bd.crec[pp] + (row + x[r]) * (u64)crec_words(md.nh), md.nh);
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh) {
					cidx[r][j] = crec_cell<NHM>(cr, j);
					bits[r] |= (u64)bit_of_nibble(crec_nib<NHM>(cr, md.nh, j)) << (4 * j);
				}
		}
	}
#define FIN_BIT(r, j) ((u32)(bits[r] >> (4 * (j))) & 15u)
#define FIN_Q(r, j) (((u64)cidx[r][j] << 4) | FIN_BIT(r, j))
	if (!snapshot) {
		// the state of the array now (two records per thread at a time: bounds the registers in flight)
		constexpr int G = RPT < 2 ? RPT : 2;
#pragma unroll
		for (int g0 = 0; g0 < RPT; g0 += G) {
			if (g0 * 1024 >= n) break;                                 // uniform
			u32 w[G][NHM];                                             // value16 | tag16 halves of the cells
#pragma unroll
			for (int g = 0; g < G; g++)
				if (live[g0 + g]) {
#pragma unroll
					for (int j = 0; j < NHM; j++)
						if (j < md.nh) w[g][j] = __hip_atomic_load(cells + cidx[g0 + g][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
#pragma unroll
			for (int g = 0; g < G; g++) {
				const int r = g0 + g;
				if (!live[r]) continue;
				bool conflict = false;
				um[r] = 0;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh) {
						const u32 b = FIN_BIT(r, j), want = (bin[r] >> j) & 1u;
						const u32 tag = (w[g][j] >> (16 + b)) & 1u, val = (w[g][j] >> b) & 1u;
						conflict |= tag && val != want;
						um[r] |= tag ? 0u : 1u << j;
					}
				if (conflict) { mark_failed(bd, pp, i, row, x[r]); live[r] = false; um[r] = 0; }
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	int succ = 0;
	u64 iters = 0;
	for (;; iters++) {
		const int par = (int)(iters % 3), par_next = (int)((iters + 1) % 3);
		// keep only the cell indices alive across iterations: slots, keys and addresses are recomputed (a few ALU ops)
		// instead of being hoisted into ~10 registers per position
#pragma unroll
		for (int r = 0; r < RPT; r++) {
#pragma unroll
			for (int j = 0; j < NHM; j++) asm volatile("" : "+v"(cidx[r][j]));
			asm volatile("" : "+v"(bits[r]), "+v"(x[r]), "+v"(bin[r]), "+v"(um[r]), "+v"(ghost[r]));
		}
		u32 resv[RPT], second[RPT], hold[RPT];                       // positions reserved in this iteration / moved to table 2 / held
		bool mine[RPT];
		// 1
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			resv[r] = live[r] ? um[r] : ghost[r];
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((resv[r] >> j) & 1u)) atomicMax(s_t1 + fin_slot1(FIN_Q(r, j)), fin_key(x[r], FIN_Q(r, j)));
		}
		if (threadIdx.x == 0) s_pending[par_next] = 0;               // last read two iterations ago
		lds_barrier();
		// 2
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			second[r] = 0;
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((resv[r] >> j) & 1u)) {
					const u64 q = FIN_Q(r, j);
					const u32 key = fin_key(x[r], q);
					if ((s_t1[fin_slot1(q)] ^ key) & 0x1FFFu) {
						second[r] |= 1u << j;
						atomicMax(s_t2 + fin_slot2(q), key);
					}
				}
			__builtin_amdgcn_sched_barrier(0);
		}
		lds_barrier();
		// 3 (reads only).  In table 2 the priority alone decides: two positions of ONE record may share a slot there,
		// and whoever has the highest priority on a slot has it on every position that maps to it.
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			hold[r] = 0;
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((resv[r] >> j) & 1u)) {
					const u64 q = FIN_Q(r, j);
					const u32 key = fin_key(x[r], q);
					const bool held = ((second[r] >> j) & 1u) ? (s_t2[fin_slot2(q)] >> 13) == (key >> 13) : s_t1[fin_slot1(q)] == key;
					hold[r] |= held ? 1u << j : 0u;
				}
			mine[r] = live[r] && hold[r] == resv[r];
			__builtin_amdgcn_sched_barrier(0);
		}
		lds_barrier();
		// 3 (the winners publish what they commit).  Only the holder of a slot writes to it in this phase, so a MARK it
		// finds there is its own: the same position again (a k-mer may hit a position twice; the values OR, as in the
		// reference's set loop, kmodel.hpp:611-618) or ANOTHER of its positions in the same slot -- that one is published
		// in a later iteration, the record staying on as a "ghost" that keeps reserving it with its priority, so that
		// nobody interested in it can hold it before the MARK has been seen.
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			if (!mine[r] && !ghost[r]) continue;
			const u32 pub = mine[r] ? resv[r] : (ghost[r] & hold[r]);
			u32 later = mine[r] ? 0u : (ghost[r] & ~hold[r]);
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((pub >> j) & 1u)) {
					const u64 q = FIN_Q(r, j);
					u32 *slot = ((second[r] >> j) & 1u) ? s_t2 + fin_slot2(q) : s_t1 + fin_slot1(q);
					const u32 cur = *slot, mark = FIN_MARK | fin_ident(q) | (((bin[r] >> j) & 1u) << 30);
					if (!(cur & FIN_MARK)) *slot = mark;
					else if ((cur & 0x3FFFFFu) == fin_ident(q)) *slot = cur | mark;
					else later |= 1u << j;
				}
			if (mine[r]) {
				if (!defer) {
#pragma unroll
					for (int j = 0; j < NHM; j++)
						if (j < md.nh && ((resv[r] >> j) & 1u)) {
							const u32 b = FIN_BIT(r, j);
							atomicOr(cells + cidx[r][j], CELL_TAG(b) | (((bin[r] >> j) & 1u) ? CELL_VAL(b) : 0u));
						}
				}
				bd.status[pp][row + x[r]] = SLOT_INSERTED;
				live[r] = false;
				won[r] = true;
				succ++;
			}
			ghost[r] = later;
			if (later) s_pending[par] = 1;
			__builtin_amdgcn_sched_barrier(0);
		}
		lds_barrier();
		// 4
#pragma unroll
		for (int r = 0; r < RPT; r++) {
			if (!live[r]) continue;
			bool conflict = false;
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((resv[r] >> j) & 1u)) {
					const u64 q = FIN_Q(r, j);
					const u32 m = ((second[r] >> j) & 1u) ? s_t2[fin_slot2(q)] : s_t1[fin_slot1(q)];
					if ((m & FIN_MARK) && (m & 0x3FFFFFu) == fin_ident(q)) {
						conflict |= ((m >> 30) & 1u) != ((bin[r] >> j) & 1u);
						um[r] &= ~(1u << j);
					}
				}
			if (conflict) { mark_failed(bd, pp, i, row, x[r]); live[r] = false; }
			else s_pending[par] = 1;
			__builtin_amdgcn_sched_barrier(0);
		}
		lds_barrier();
		// 5
#pragma unroll
		for (int r = 0; r < RPT; r++)
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((resv[r] >> j) & 1u)) {
					const u64 q = FIN_Q(r, j);
					s_t1[fin_slot1(q)] = 0;
					if ((second[r] >> j) & 1u) s_t2[fin_slot2(q)] = 0;
				}
		lds_barrier();
		if (!s_pending[par]) break;
	}
#undef FIN_BIT
#undef FIN_Q
#pragma unroll
	for (int r = 0; r < RPT; r++)
		if (won[r] && defer) bd.Urec[lv][(row + rec[r]) * (1 + W)] |= REC_WON;   // k_reorder applies it
	if (!defer) drain_vmem();                                        // the next range gathers what this one committed
	if (succ) atomicAdd(s_succ, succ);
	return iters + 1;
}

// More records than the registers hold: the list order lets the finisher take them in index ranges, one range to
// completion before the next (a record only ever waits for smaller indices).  The ranges are cut by index so that about
// 7/8 of the capacity falls into each; a range that turns out too full is halved.
template <int W, int NHM, int RPT>
__device__ __forceinline__ void finish_lds_ranges(const ModelDev &md, const BlockDev &bd, int pp, int i, int a, int lv, int n, bool snapshot,
                                                   u32 *s_t1, u32 *s_t2, u32 *s_list, int *s_count, int *s_pending, int *s_succ)
{
	constexpr int CAP = RPT * 1024;
	const u64 row = (u64)i * KMX_BUCKET;
	for (int q = threadIdx.x; q < KMX_FIN_T; q += 1024) { s_t1[q] = 0; s_t2[q] = 0; }
	u64 iters = 0;
	if (n <= CAP) {
		__syncthreads();
		iters = finish_lds<W, NHM, RPT>(md, bd, pp, i, a, lv, n, nullptr, snapshot, true, s_t1, s_t2, s_pending, s_succ);
	} else {
		const int n_list = bd.n[pp][i];
		int lo = 0, remaining = n;
		while (remaining > 0) {
			int hi = n_list;
			if (remaining > CAP) hi = lo + max(1, (int)((long long)(n_list - lo) * (CAP * 7 / 8) / remaining));
			int cnt;
			for (;;) {
				if (threadIdx.x == 0) *s_count = 0;
				__syncthreads();
				for (int u = threadIdx.x; u < n; u += 1024) {
					const int x = (int)((u32)bd.Urec[lv][(row + u) * (1 + W)] & (KMX_BUCKET - 1));
					if (x >= lo && x < hi) {
						const int slot = atomicAdd(s_count, 1);
						if (slot < CAP) s_list[slot] = (u32)u;
					}
				}
				__syncthreads();
				cnt = *s_count;
				__syncthreads();
				if (cnt <= CAP) break;
				hi = lo + max(1, (hi - lo) / 2);
			}
			iters += finish_lds<W, NHM, RPT>(md, bd, pp, i, a, lv, cnt, s_list, snapshot && lo == 0, false, s_t1, s_t2, s_pending, s_succ);
			__syncthreads();
			if (threadIdx.x == 0) s_pending[0] = 0;                  // finish_lds starts with s_pending[0] clear
			remaining -= cnt;
			lo = hi;
			if (lo >= n_list) break;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		atomicAdd(bd.stats + ST_FIN_ITERS, iters);
		if (*s_succ) atomicAdd(bd.stats + ST_SLOW_SUCC, (u64)*s_succ);
	}
}

// finish_global -- more records than finish_lds holds (the host adds grid-wide passes when it sees that happen): the
// same iteration through the records and the epoch-tagged reservation table in global memory.  The first iteration
// of s == 0 resolves k_round_commit's reservations (epoch_b); afterwards it reserves for itself with epoch0, epoch0+1, ...
template <int W, int NHM>
__device__ __forceinline__ void finish_global(const ModelDev &md, const BlockDev &bd, int pp, int i, int a, int lv, int n, bool first, u64 epoch_b, u64 epoch0,
                                               int *s_pending, int *s_succ)
{
	const u64 row = (u64)i * KMX_BUCKET;
	u64 epoch = epoch0;
	u64 iters = 0;
	for (;; iters++) {
		if (threadIdx.x == 0) *s_pending = 0;
		__syncthreads();
		if (!first) {
			for (int u = threadIdx.x; u < n; u += 1024) {
				u32 x, bin;
				u64 v[W];
				rec_load<W>(bd.Urec[lv], row + u, x, bin, v);
				if (bd.status[pp][row + x] != SLOT_UNDECIDED) continue;
				Premixed<W> pm = premix_string<W>(left_align<W>(v, md.k), md.gfull);
				Touches<NHM> tc;
				gather_touches<W, NHM, true>(md, pm, a, tc);
				if (touches_conflict<NHM>(md, tc, bin)) { mark_failed(bd, pp, i, row, x); continue; }
				reserve_untagged<NHM>(md, bd, i, tc, resv_key(epoch, x));
				*s_pending = 1;
			}
			drain_vmem();                                      // our atomics are performed; readers use sc1 loads
			__syncthreads();
			const int pending = *s_pending;
			__syncthreads();
			if (!pending) break;
		}
		int succ = 0;
		for (int u = threadIdx.x; u < n; u += 1024) {
			u32 x, bin;
			u64 v[W];
			const u32 um = rec_load<W>(bd.Urec[lv], row + u, x, bin, v);
			if (bd.status[pp][row + x] != SLOT_UNDECIDED) continue;
			Aligned<W> al = left_align<W>(v, md.k);
			Premixed<W> pm = premix_string<W>(al, md.gfull);
			if (first) {
				// k_round_commit's reservations are resolved exactly as k_slow_resolve0 does it, from the untagged mask
				// check_emit left in the record: whoever holds the reservation of EVERY position in it commits those
				// positions -- no gather.  (Until the end of round 2 this iteration gathered the cells afresh and asked only
				// for the positions still untagged by then; that is sound too, now that a position is committed with its
				// final value in one atomic -- see value_at_position -- but it costs a gather the snapshot makes unnecessary.)
				const u64 key = resv_key(epoch_b, x);
				const int sbase = a * md.nh;
				cell_t *cells = md.cells[a];
				u64 pos[NHM];
				bool mine = true;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((um >> j) & 1u)) {
						pos[j] = mod_u64(murmur_seeded<W>(pm, md.gfull, c_seeds[(sbase + j) & 127]), md.km_mod);
						mine &= cell_load_coherent(resv_slot(bd, i, pos[j])) == key;
					}
				if (mine) {
#pragma unroll
					for (int j = 0; j < NHM; j++)
						if (j < md.nh && ((um >> j) & 1u)) {
							const u32 b = bit_in_cell(pos[j]);
							atomicOr(cells + (pos[j] >> 4), CELL_TAG(b) | (value_at_position<NHM>(md, pos, um, bin, j) ? CELL_VAL(b) : 0u));
						}
					bd.status[pp][row + x] = SLOT_INSERTED;
					succ++;
				}
				continue;
			}
			Touches<NHM> tc;
			gather_touches<W, NHM, true>(md, pm, a, tc);
			if (owns_outcome<NHM, true>(md, bd, i, tc, bin, resv_key(epoch, x))) {
				commit_touches<W, NHM>(md, tc, bin, a);
				bd.status[pp][row + x] = SLOT_INSERTED;
				succ++;
			}
		}
		if (succ) atomicAdd(s_succ, succ);
		drain_vmem();
		__syncthreads();
		if (!first) epoch++;
		first = false;
	}
	if (threadIdx.x == 0) {
		atomicAdd(bd.stats + ST_FIN_ITERS, iters);
		if (*s_succ) atomicAdd(bd.stats + ST_SLOW_SUCC, (u64)*s_succ);
	}
}

// One workgroup per list.
// Workgroups past the first nb are riders: the finisher is ONE workgroup per list that walks a chain of dependent steps
// (33 us in round 0, 10-17 us later) while the other 250 CUs idle, so the km_back emission the PREVIOUS block still owes
// (`job`: its staging region and survivor flags) runs in the same launch, one 1024-thread workgroup per 1024 k-mers --
// throughput work that costs the launch (almost) nothing.  (Hosted by the late rounds' check launches instead it cost
// 15-24 us per launch: its LDS cut the check's occupancy from 8 to 3 workgroups per CU.)
template <int W, int NHM> __global__ __launch_bounds__(1024) void k_slow_finish(ModelDev md, BlockDev bd, int t, int pp, int s, u64 epoch_b, u64 epoch0, int force_global, KmbackJob job, BitScatter kmb)
{
	constexpr int RKPT = KMX_FIN_RIDER_KPT(NHM);
	constexpr int FIN_BYTES = (2 * KMX_FIN_T + KMX_FIN_RPT(NHM) * 1024) * 4 + 8 * 4, RIDER_BYTES = BS_LDS_BYTES(RKPT * (NHM - 2), 1024);
	__shared__ __align__(16) unsigned char pool[FIN_BYTES > RIDER_BYTES ? FIN_BYTES : RIDER_BYTES];
	if ((int)blockIdx.x >= md.nb) {
		const int r = (int)blockIdx.x - md.nb, gxk = KMX_BUCKET / (1024 * RKPT);
		kmback_emit_body<W, NHM, RKPT, 1024>(md, bd, job.kmers, job.surv, pp, job.n_in_block, kmb, job.i0 + r / gxk, r % gxk, gxk, pool);
		return;
	}
	u32 *s_t1 = (u32 *)pool, *s_t2 = s_t1 + KMX_FIN_T, *s_list = s_t2 + KMX_FIN_T;
	int *s_pending = (int *)(s_list + KMX_FIN_RPT(NHM) * 1024);
	int &s_succ = s_pending[3], &s_count = s_pending[4];
	const int i = blockIdx.x, lv = s & 1;
	const int n = bd.Un[UN_IDX(lv, i, md.nb)];
	if (n == 0) return;
	const int a = (i + t) % md.nb;
	constexpr int RPT = KMX_FIN_RPT(NHM);
	const bool lds_path = n <= KMX_FIN_RANGES * RPT * 1024 && !force_global && md.km_mod.d < KMX_FIN_MAX_POS;
	if (threadIdx.x == 0) {
		s_succ = 0;
		s_pending[0] = 0;
		atomicMax(bd.stats + ST_MAX_UFIN, (u64)n);
		if (s == 0) { atomicAdd(bd.stats + ST_CONTENDED, (u64)n); atomicMax(bd.stats + ST_MAX_U0, (u64)n); }
	}
	__syncthreads();
	const bool snap = s == 0 && snapshot_ok(bd, i);
	if (lds_path) finish_lds_ranges<W, NHM, RPT>(md, bd, pp, i, a, lv, n, snap, s_t1, s_t2, s_list, &s_count, s_pending, &s_succ);
	else finish_global<W, NHM>(md, bd, pp, i, a, lv, n, snap, epoch_b, epoch0, s_pending, &s_succ);
}

// ------------------------------------------------------------------------------------------ R: reorder
// reorder_buffer (kmodel.hpp:529-540): m survivors; survivors already below m stay; the i-th hole from the left
// (below m) receives the i-th survivor from the right (at or above m).  ONE launch: the tile survivor counts were
// accumulated while the slots failed; holes are written as (LIST_HOLE | rank) and filled lazily from mover[] by the
// next round's check_emit (or by rest_append after the last round), so no grid-wide fill pass is needed.
// Also closes the round's books: successes = n - m.
// Before that, it applies what the finisher decided but left undone (REC_WON records of level lv): tag/value bits
// (kmodel.hpp:611-618, every position: an already tagged one carries the same value) and the km_back insert (:548-550).
template <int W, int NHM> __global__ __launch_bounds__(256) void k_reorder(ModelDev md, BlockDev bd, int t, int pp, int lv)
{
	__shared__ int s_tmp[4];
	__shared__ int s_m, s_off;
	const int nb = md.nb;
	const int i = blockIdx.y, tile = blockIdx.x;
	const int n = bd.n[pp][i];
	const u64 row = (u64)i * KMX_BUCKET;
	if (tile >= (int)KMX_NTILES) {                                   // the extra workgroups only apply; the others only reorder
		const int nrec = bd.Un[UN_IDX(lv, i, nb)];
		for (int u = (tile - (int)KMX_NTILES) * 256 + threadIdx.x; u < nrec; u += KMX_APPLY_WGS * 256) {
			const u64 *rec = bd.Urec[lv] + (row + u) * (1 + W);
			if (rec[0] & REC_WON) {
				const int a = (i + t) % nb;
				u32 x, bin;
				u64 v[W];
				rec_load<W>(bd.Urec[lv], row + u, x, bin, v);
				Aligned<W> al = left_align<W>(v, md.k);
				Premixed<W> pm = premix_string<W>(al, md.gfull);
				cell_t *cells = md.cells[a];
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh) {
						const u64 pos = mod_u64(murmur_seeded<W>(pm, md.gfull, c_seeds[(a * md.nh + j) & 127]), md.km_mod);
						const u32 b = bit_in_cell(pos);
						atomicOr(cells + (pos >> 4), CELL_TAG(b) | (((bin >> j) & 1u) ? CELL_VAL(b) : 0u));
					}
			}
		}
		return;
	}
	{   // every workgroup scans the 256 tile counts of its list
		int c = bd.tile_cnt[pp][i * KMX_NTILES + threadIdx.x];
		int tot;
		int ex = block_excl_scan_256(c, s_tmp, &tot);
		if ((int)threadIdx.x == tile) s_off = ex;
		if (threadIdx.x == 0) s_m = tot;
		__syncthreads();
	}
	const int m = s_m;
	int f[4], c = 0;
#pragma unroll
	for (int q = 0; q < 4; q++) {
		int x = tile * KMX_TILE + threadIdx.x * 4 + q;
		f[q] = (x < n && bd.status[pp][row + x] == SLOT_FAILED) ? 1 : 0;
		c += f[q];
	}
	int before = block_excl_scan_256(c, s_tmp, nullptr) + s_off;
	const u32 *oldl = bd.list[pp] + row;
	u32 *newl = bd.list[pp ^ 1] + row;
	u32 *mv = bd.mover[pp ^ 1] + row;
#pragma unroll
	for (int q = 0; q < 4; q++) {
		int x = tile * KMX_TILE + threadIdx.x * 4 + q;
		if (x < n) {
			if (x < m) newl[x] = f[q] ? oldl[x] : (LIST_HOLE | (u32)(x - before));     // hole number x-before, left to right
			else if (f[q]) mv[m - before - 1] = oldl[x];                                // survivor number m-before-1 from the right
		}
		before += f[q];
	}
	bd.tile_cnt[pp ^ 1][i * KMX_NTILES + tile] = 0;                 // next round counts into the other buffer
	if (tile == 0 && threadIdx.x == 0) {
		bd.cl_ovf[i] = 0;
		if (n > m) atomicAdd(bd.stats + ST_SUCCESSES, (u64)(n - m));
		bd.n[pp ^ 1][i] = m;                                       // (the record counters are reset by the next check_emit)
	}
}

// ------------------------------------------------------------------------------------------ partitioned bit-set: the sweep
// One workgroup per bin: the bin's slice of the filter is swept tile by tile (2^20 positions in LDS); a tile collects
// its bits with LDS atomics and is OR-ed back with coalesced whole-word accesses.  Only this workgroup writes these
// words during the launch (producers that met a full bin used atomics in EARLIER launches).
__global__ __launch_bounds__(1024) void k_bs_apply(BitScatter bs)
{
	__shared__ u32 s_tile[BS_TILE_WORDS];
	const u32 TW = 1u << (bs.tlog2 - 5);                             // words per tile (BS_TILE_WORDS outside the test hook)
	const u32 b = blockIdx.x;
	u32 cnt = (u32)bs.cnt[b];
	if (cnt == 0) return;                                            // uniform
	if (cnt > bs.cap) cnt = bs.cap;
	const u64 wpb = 1ULL << (bs.wshift - 5), word0 = (u64)b * wpb;
	if (word0 >= bs.nwords) return;
	const u32 nw = (u32)((bs.nwords - word0) < wpb ? (bs.nwords - word0) : wpb);
	const u32 *tup = bs.tup + (u64)b * bs.cap;
	u32 *words = bs.words + word0;
	for (u32 t0 = 0; t0 < nw; t0 += TW) {
		const u32 tw = nw - t0 < TW ? nw - t0 : TW;
		for (u32 w = threadIdx.x; w < tw; w += 1024) s_tile[w] = 0;
		__syncthreads();
		for (u32 q = threadIdx.x; q < cnt; q += 1024) {
			const u32 o = tup[q], w = (o >> 5) - t0;
			if (w < tw) atomicOr(&s_tile[w], 1u << (o & 31));         // (o >> 5) < t0 wraps to a huge value: skipped too
		}
		__syncthreads();
		for (u32 w = threadIdx.x; w < tw; w += 1024) {
			const u32 x = s_tile[w];
			if (x) words[t0 + w] |= x;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) bs.cnt[b] = 0;
}

// Second level for big filters.  k_bs_split: the tuples of bin blockIdx.y are dealt to the bin's tiles the way
// bs_block_emit dealt them to the bins (per-tile counts in LDS, ONE global atomic per run, LDS-staged run-by-run
// write-out); a tile that is full gets the bit set with an atomic instead (exact either way).
#define BS_SPLIT_K 8
__global__ __launch_bounds__(256) void k_bs_split(BitScatter bs)
{
	__shared__ int s_cnt[1 << BS_MAX_TILES_LOG2], s_off[1 << BS_MAX_TILES_LOG2], s_base[1 << BS_MAX_TILES_LOG2], s_tmp[4];
	__shared__ u32 s_stage[256 * BS_SPLIT_K];
	const u32 b = blockIdx.y, tl = bs.wshift - bs.tlog2, tmask = (1u << bs.tlog2) - 1;
	u32 cnt = (u32)bs.cnt[b];
	if (cnt > bs.cap) cnt = bs.cap;
	const u32 *tup = bs.tup + (u64)b * bs.cap;
	u32 *out = bs.tup2 + (((u64)b << tl) * bs.cap2);
	int *gcnt = bs.cnt2 + ((u64)b << tl);
	for (u32 base = blockIdx.x * (256 * BS_SPLIT_K); base < cnt; base += gridDim.x * (256 * BS_SPLIT_K)) {   // uniform trip count
		s_cnt[threadIdx.x] = 0;
		__syncthreads();
		u32 o[BS_SPLIT_K];
		int rank[BS_SPLIT_K];
#pragma unroll
		for (int j = 0; j < BS_SPLIT_K; j++) {
			const u32 q = base + j * 256 + threadIdx.x;
			o[j] = q < cnt ? tup[q] : ~0u;
		}
#pragma unroll
		for (int j = 0; j < BS_SPLIT_K; j++)
			if (o[j] != ~0u) rank[j] = atomicAdd(&s_cnt[o[j] >> bs.tlog2], 1);
		__syncthreads();
		int total;
		{
			const int c = s_cnt[threadIdx.x];
			const int ex = block_excl_scan_256(c, s_tmp, &total);
			s_off[threadIdx.x] = ex;
			s_base[threadIdx.x] = c ? atomicAdd(gcnt + threadIdx.x, c) : 0;
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < BS_SPLIT_K; j++)
			if (o[j] != ~0u) s_stage[s_off[o[j] >> bs.tlog2] + rank[j]] = o[j];
		__syncthreads();
		for (int q = threadIdx.x; q < total; q += 256) {
			const u32 e = s_stage[q], t = e >> bs.tlog2;
			const u32 g = (u32)s_base[t] + (u32)(q - s_off[t]);
			if (g < bs.cap2) out[(u64)t * bs.cap2 + g] = e & tmask;
			else atomicOr(bs.words + (((u64)b << bs.wshift) >> 5) + (e >> 5), 1u << (e & 31));
		}
		__syncthreads();
	}
}
// k_bs_apply2: one workgroup per (tile, bin).  Only this workgroup writes these words during the launch (k_bs_split's
// atomics for full tiles came in the launch before).
__global__ __launch_bounds__(1024) void k_bs_apply2(BitScatter bs)
{
	__shared__ u32 s_tile[BS_TILE_WORDS];
	const u32 b = blockIdx.y, t = blockIdx.x, tl = bs.wshift - bs.tlog2, TW = 1u << (bs.tlog2 - 5);
	int *gc = bs.cnt2 + (((u64)b << tl) + t);
	u32 cnt = (u32)*gc;
	if (t == 0 && threadIdx.x == 0) bs.cnt[b] = 0;                   // (k_bs_split has read it, a launch ago)
	if (cnt == 0) return;                                            // uniform
	if (cnt > bs.cap2) cnt = bs.cap2;
	const u64 word0 = (((u64)b << bs.wshift) >> 5) + (u64)t * TW;
	if (word0 >= bs.nwords) { if (threadIdx.x == 0) *gc = 0; return; }
	const u32 tw = (u32)((bs.nwords - word0) < TW ? (bs.nwords - word0) : TW);
	const u32 *tup = bs.tup2 + (((u64)b << tl) + t) * bs.cap2;
	u32 *words = bs.words + word0;
	for (u32 w = threadIdx.x; w < tw; w += 1024) s_tile[w] = 0;
	__syncthreads();
	for (u32 q = threadIdx.x; q < cnt; q += 1024) {
		const u32 o = tup[q];
		if ((o >> 5) < tw) atomicOr(&s_tile[o >> 5], 1u << (o & 31));
	}
	__syncthreads();
	for (u32 w = threadIdx.x; w < tw; w += 1024) {
		const u32 x = s_tile[w];
		if (x) words[w] |= x;
	}
	if (threadIdx.x == 0) *gc = 0;
}

// km_back insert of a round (kmodel.hpp:548-550), deferred: every slot the round decided as inserted -- by the parallel
// commit or by the ordered path -- contributes the nh-2 positions of its (k-2)-mer to the BitScatter of km_back.
// n_in_block >= 0: once per block instead -- every k-mer of the block that did not go to the rest table was inserted in
// one of the rounds (the single-GPU build; in the multi-GPU ring a rank sees a list for one round only).
template <int W, int NHM, int KPT, int BT>
__device__ __forceinline__ void kmback_emit_body(const ModelDev &md, const BlockDev &bd, const u64 *kmers, const unsigned char *surv, int pp, int n_in_block,
                                                 const BitScatter &bs, int i, int bx, int gx, unsigned char *lds)
{
	constexpr int K = KPT * (NHM - 2);
	BS_LDS_AT(K, BT, lds);
	const bool whole = n_in_block >= 0;
	int n;
	if (whole) { n = n_in_block - i * (int)KMX_BUCKET; n = n < 0 ? 0 : (n > (int)KMX_BUCKET ? (int)KMX_BUCKET : n); }
	else n = bd.n[pp][i];
	const u64 row = (u64)i * KMX_BUCKET;
	for (int base = bx * BT * KPT; base < n; base += gx * BT * KPT) {     // uniform trip count per workgroup
		u64 v[K];
		u32 valid = 0;
#pragma unroll
		for (int q = 0; q < KPT; q++) {
			const int x = base + q * BT + (int)threadIdx.x;
			if (x < n && (whole ? !surv[row + x] : bd.status[pp][row + x] != SLOT_FAILED)) {
				const u32 idx = whole ? (u32)x : bd.list[pp][row + x];
				u64 km[W];
				load_kmer<W>(kmers, row + idx, km);
				Premixed<W> pb = premix_string<W>(drop_first_base<W>(left_align<W>(km, md.k)), md.gback);
				if (md.kmb_direct) {                                     // filter too small or too big for the partitioned bit-set: test + atomic OR per bit
					bloom_insert_pm<W>(pb, md.gback, md.km_back, md.km_back_mod, md.nh - 2);
					continue;
				}
#pragma unroll
				for (int j = 0; j < NHM - 2; j++)
					if (j < md.nh - 2) {
						const u64 pos = mod_u64(murmur_seeded<W>(pb, md.gback, c_seeds[j]), md.km_back_mod);
						v[q * (NHM - 2) + j] = ((pos >> 5) << 5) | bit_in_word32(pos);
						valid |= 1u << (q * (NHM - 2) + j);
					}
			}
		}
		if (!md.kmb_direct) bs_block_emit<K, BT>(bs, v, valid, s_bs_cnt, s_bs_off, s_bs_base, s_bs_tmp, s_bs_stage);   // (uniform)
	}
}
template <int W, int NHM, int KPT> __global__ __launch_bounds__(256) void k_kmback_emit(ModelDev md, BlockDev bd, const u64 *kmers, const unsigned char *surv, int i0, int istride, int pp, int n_in_block, BitScatter bs)
{
	__shared__ __align__(16) unsigned char lds[BS_LDS_BYTES(KPT * (NHM - 2), 256)];
	kmback_emit_body<W, NHM, KPT>(md, bd, kmers, surv, pp, n_in_block, bs, i0 + (int)blockIdx.y * istride, (int)blockIdx.x, (int)gridDim.x, lds);
}

// survivors of the block go to the rest table (kmodel.hpp:567-571); slot 0 is remembered for the
// stale-slot duplicate of the final block (quirk Q1)
// (i0: first list of the launch -- the whole block with grid.y = nb, ONE list that a rank of the multi-GPU ring retires, or the
// lists i0, i0 + istride, ... a rank of the range partition holds)
template <int W> __global__ __launch_bounds__(256) void k_rest_append(BlockDev bd, int pp, int i0, int istride, u64 *rest_kmers, int *rest_counts, unsigned long long *rest_n, u64 *stale_kmers, int *stale_counts, u64 *feedback)
{
	__shared__ int s_cnt;
	__shared__ unsigned long long s_base;
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // the block's contention figures go straight to the host's pinned words
		feedback[0] = bd.stats[ST_MAX_U0]; feedback[1] = bd.stats[ST_MAX_UFIN]; feedback[2] = bd.stats[ST_MAX_LATE_BIN];
		bd.stats[ST_MAX_U0] = 0; bd.stats[ST_MAX_UFIN] = 0; bd.stats[ST_MAX_LATE_BIN] = 0;
	}
	if (threadIdx.x == 0) s_cnt = 0;
	__syncthreads();
	const int i = i0 + (int)blockIdx.y * istride, x = blockIdx.x * 256 + threadIdx.x;
	const int n = bd.n[pp][i];
	if (x == 0 && n == 0) stale_counts[i] = 0;
	const bool act = x < n;
	const u64 mask = __ballot(act);
	const int lane = threadIdx.x & 63;
	int wbase = 0;
	if (lane == 0 && mask) wbase = atomicAdd(&s_cnt, (int)__popcll(mask));
	wbase = __shfl(wbase, 0, 64);
	__syncthreads();
	if (threadIdx.x == 0) s_base = s_cnt ? atomicAdd(rest_n, (unsigned long long)s_cnt) : 0ULL;
	__syncthreads();
	if (!act) return;
	const u64 p = s_base + (u64)wbase + (u64)__popcll(mask & ((1ULL << lane) - 1));
	const u64 row = (u64)i * KMX_BUCKET;
	const u32 idx = list_entry(bd, pp, row, x);
	u64 v[W];
	load_kmer<W>(bd.kmers, row + idx, v);
	const int c = (int)bd.counts[row + idx];
	store_kmer<W>(rest_kmers, p, v);
	rest_counts[p] = c;
	bd.surv[row + idx] = 1;
	if (x == 0) { store_kmer<W>(stale_kmers, (u64)i, v); stale_counts[i] = c; }
}

// ------------------------------------------------------------------------------------------ multi-GPU ring (one model, several GPUs)
// The rotation insert_array(buff[i], (i + t) % n_thread, ...) (kmodel.hpp:560-565) with the arrays owned whole by
// different GPUs: after a round the survivors of a list travel, in list order, to the GPU that owns the next array.
// A list in flight is a *message* in device memory: u64 header[8] (header[0] = n), BUCKET*W packed k-mers, BUCKET
// counts.  The receiver starts the next round from a fresh identity list over the message (the reference physically
// moves its KmerBuff entries in reorder_buffer, so the order of a list is all there is to it).
template <int W> __global__ __launch_bounds__(256) void k_ring_import(BlockDev bd, int nb, int pp, RingLists rl, u64 *stg_kmers, u32 *stg_counts)
{
	const int i = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
	const RingList e = rl.e[i];
	const u64 row = (u64)i * KMX_BUCKET;
	int n = 0;
	if (e.active) {
		const u64 *src_k = e.n_host >= 0 ? e.src_kmers : e.src_msg + KMX_MSG_HDR;
		const u32 *src_c = e.n_host >= 0 ? e.src_counts : (const u32 *)(e.src_msg + KMX_MSG_HDR + (u64)KMX_BUCKET * W);
		n = e.n_host >= 0 ? e.n_host : (int)e.src_msg[0];
		n = n < 0 ? 0 : (n > (int)KMX_BUCKET ? (int)KMX_BUCKET : n);   // a malformed header must not walk out of the buffers (or give a negative length)
		if (x < n) {
			u64 v[W];
			load_kmer<W>(src_k, (u64)x, v);
			store_kmer<W>(stg_kmers, row + x, v);
			stg_counts[row + x] = src_c[x];
		}
		bd.list[pp][row + x] = (u32)x;
		bd.surv[row + x] = 0;                                     // (the range partition emits km_back once per block, from the survivor flags)
	}
	if (x < (int)KMX_NTILES) { bd.tile_cnt[0][i * KMX_NTILES + x] = 0; bd.tile_cnt[1][i * KMX_NTILES + x] = 0; }
	if (x == 0) {
		bd.n[pp][i] = n;                                           // lists that are elsewhere in the ring this round are empty here
		for (int s = 0; s < KMX_NSLOW; s++) bd.Un[UN_IDX(s, i, nb)] = 0;
	}
}

// survivors of the round (list[pp], after k_reorder) in list order -> message
template <int W> __global__ __launch_bounds__(256) void k_ring_export(BlockDev bd, int pp, RingLists rl, u64 *feedback)
{
	const int i = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
	// a rank that hands every list on never runs k_rest_append, which reports the contention figures to the host: the fullest late
	// claim bin (it picks the form of the late rounds' k_round_detect) is reported here too (a maximum over the build so far)
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
		const u64 late = bd.stats[ST_MAX_LATE_BIN];                     // (0 unless a bin went beyond 1024 tuples: the host's words are then not touched)
		if (late && late > feedback[2]) feedback[2] = late;
	}
	const RingList e = rl.e[i];
	if (!e.active || !e.dst_msg) return;
	const int n = bd.n[pp][i];
	const u64 row = (u64)i * KMX_BUCKET;
	if (x == 0) e.dst_msg[0] = (u64)n;
	if (x >= n) return;
	const u32 idx = list_entry(bd, pp, row, x);
	u64 v[W];
	load_kmer<W>(bd.kmers, row + idx, v);
	store_kmer<W>(e.dst_msg + KMX_MSG_HDR, (u64)x, v);
	((u32 *)(e.dst_msg + KMX_MSG_HDR + (u64)KMX_BUCKET * W))[x] = bd.counts[row + idx];
}

// dst |= src, 32-bit words (merging the partial Bloom / back filters of the ranks: set_bit is an OR, kmodel.hpp:576-581)
__global__ __launch_bounds__(256) void k_or_words(u32 *dst, const u32 *src, u64 n)
{
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
		const u32 s = src[i];
		if (s) dst[i] |= s;
	}
}

// ------------------------------------------------------------------------------------------ query
template <int W> __device__ __forceinline__ u64 shr128_lo(const u64 *v, int s)   // low 64 bits of (value >> s)
{
	if (W == 1) return s >= 64 ? 0 : v[0] >> s;
	if (s >= 64) return s >= 128 ? 0 : v[0] >> (s - 64);
	return s ? (v[0] << (64 - s)) | (v[1] >> s) : v[1];
}

// KRestData::check_kmer (rest.hpp:223-251), literally: binary search over the prefix group with an INCLUSIVE upper
// bound, so the first row of the next group can match too.  Rows are suffix integers (same order as the byte rows).
template <int W> __device__ __forceinline__ int rest_check_reference(const ModelDev &md, u32 pre, const u64 *key)
{
	const int g = md.rest_h2i[pre];
	if (g < 0) return 0;
	int low = md.rest_pre[g], high = md.rest_pre[g + 1], mid = 0;
	bool found = false;
	while (low <= high) {
		mid = (low + high) / 2;
		if ((u64)mid >= md.rest_entries) break;             // divergence D3: the row past the table never matches
		int c = 0;
#pragma unroll
		for (int w = 0; w < W; w++) {
			u64 r = md.rest_suf[(u64)mid * W + w];
			if (c == 0) c = key[w] < r ? -1 : (key[w] > r ? 1 : 0);
		}
		if (c < 0) high = mid - 1;
		else if (c > 0) low = mid + 1;
		else { found = true; break; }
	}
	return found ? md.rest_cnt[mid] : 0;
}

// The same answer with ~3 touches instead of ~12: rows are bucketed by the top F bits of the k-mer (a bucket lies
// inside one prefix group and holds < 1 row on average), so an exact match is found by scanning the bucket; when
// there is none, the only row the reference could still return is the first row of the next group, and only if
// its suffix equals ours -- then, and only then, the literal search above is replayed.
template <int W> __device__ __forceinline__ int rest_check(const ModelDev &md, const u64 *v)
{
	if (!md.rest_entries) return 0;
	const int sbits = 2 * (md.k - md.rest_pre_len);
	const u32 pre = (u32)shr128_lo<W>(v, sbits);
	u64 key[W];
	if (W == 1) key[0] = v[0] & ((1ULL << sbits) - 1);
	else {
		if (sbits >= 64) { key[0] = sbits >= 128 ? v[0] : v[0] & ((1ULL << (sbits - 64)) - 1); key[W - 1] = v[W - 1]; }
		else { key[0] = 0; key[W - 1] = v[W - 1] & ((1ULL << sbits) - 1); }
	}
	const u32 b = (u32)shr128_lo<W>(v, 2 * md.k - md.rest_fbits);
	u32 lo = md.rest_fine[b], hi = md.rest_fine[b + 1];
	while (hi - lo > 4) {                                   // skewed data: narrow long buckets first
		const u32 mid = (lo + hi) >> 1;
		bool less = false, eq = true;
#pragma unroll
		for (int w = 0; w < W; w++) {
			const u64 r = md.rest_suf[(u64)mid * W + w];
			if (eq && r != key[w]) { less = r < key[w]; eq = false; }
		}
		if (eq) return md.rest_cnt[mid];
		if (less) lo = mid + 1; else hi = mid;
	}
	for (u32 e = lo; e < hi; e++) {
		bool eq = true;
#pragma unroll
		for (int w = 0; w < W; w++) eq &= md.rest_suf[(u64)e * W + w] == key[w];
		if (eq) return md.rest_cnt[e];
	}
	bool q_eq = true;
#pragma unroll
	for (int w = 0; w < W; w++) q_eq &= md.rest_q[(u64)pre * W + w] == key[w];
	return q_eq ? rest_check_reference<W>(md, pre, key) : 0;
}

// check_all_bf (kmodel.hpp:361-371): filter order {0} for ci==1, {1,0,2} otherwise.  PW = words of the pre-mixed string.
template <int PW> __device__ __forceinline__ int check_all_bf(const ModelDev &md, const StrGeom gf, const StrGeom gb, const Premixed<PW> &pf, const Premixed<PW> &pb)
{
	for (int j = 0; j < md.bf_num; j++) {
		int i = md.ci == 1 ? j : (j == 0 ? 1 : (j == 1 ? 0 : 2));
		bool a = bloom_check_pm<PW>(pf, gf, md.bf[i], md.bf_mod[i], md.nh - 1);
		bool b = a && bloom_check_pm<PW>(pb, gb, md.bf_back[i], md.bf_back_mod[i], md.nh - 2);
		if (a && b) return i + md.ci;
	}
	return 0;
}

// one coupled array: -1 if a tag is missing, else the value bits, hash j -> bit j (kmodel.hpp:630-642)
template <int PW> __device__ __forceinline__ int decode_array(const ModelDev &md, const StrGeom gf, const Premixed<PW> &pf, int a)
{
	if (!md.km_mod.d) return -1;
	const cell_t *cells = md.cells[a];
	int v = 0;
	bool ok = true;
	for (int j = 0; j < md.nh && ok; j++) {
		u64 pos = mod_u64(murmur_seeded<PW>(pf, gf, c_seeds[(a * md.nh + j) & 127]), md.km_mod);
		const cell_t cell = cells[pos >> 4];
		u32 b = bit_in_cell(pos);
		ok = (cell >> (16 + b)) & 1u;
		v |= (int)((cell >> b) & 1u) << j;
	}
	return ok ? v : -1;
}

// the part of get_candidates (kmodel.hpp:326-342) after canonicalisation and the rest lookup; -2 = contributes nothing
template <int PW> __device__ __forceinline__ int candidate_from_filters(const ModelDev &md, const StrGeom gf, const StrGeom gb, const Premixed<PW> &pf, const Premixed<PW> &pb)
{
	int occ = check_all_bf<PW>(md, gf, gb, pf, pb);
	if (occ != 0) return occ;
	if (bloom_check_pm<PW>(pb, gb, md.km_back, md.km_back_mod, md.nh - 2)) {
		int result = -1;                                     // find_bitarray_one (kmodel.hpp:650-671, quirk Q3)
		for (int a = 0; a < md.nb; a++) {
			int d = decode_array<PW>(md, gf, pf, a);
			if (d >= 0) { result = d; if (d != 0) break; }
		}
		if (result > -1) return result;
	}
	return -2;
}

// kmer_to_occ after canonicalisation and the rest lookup (kmodel.hpp:107-115) + kmer_to_bin (:286-323).
// `neighbours(cand)` fills the up-to-8 neighbour candidates (get_neighbor_kmer_bin, :344-359) and returns their number.
template <int PW, typename NEIGH>
__device__ __forceinline__ int occ_from_filters(const ModelDev &md, const StrGeom gf, const StrGeom gb, const Premixed<PW> &pf, const Premixed<PW> &pb, NEIGH neighbours)
{
	const bool in_back = bloom_check_pm<PW>(pb, gb, md.km_back, md.km_back_mod, md.nh - 2);
	const int occ = check_all_bf<PW>(md, gf, gb, pf, pb);
	if (!in_back) return occ;                                // :109-111
	int nv = 0, first = 0;
	for (int a = 0; a < md.nb; a++) {                        // find_bitarray (:625-646)
		int d = decode_array<PW>(md, gf, pf, a);
		if (d > 0) { if (nv == 0) first = d; nv++; }
	}
	int bin;
	if (nv == 0) bin = occ;
	else if (nv == 1) {
		bin = first;
		if (occ) {
			int cand[8];
			int nc = neighbours(cand), cnt = 0;
			for (int c = 0; c < 8; c++) cnt += (c < nc && cand[c] < md.ci + md.bf_num) ? 1 : 0;
			if (cnt >= nc / 2) bin = occ;
		}
	} else {
		int cand[8];
		int nc = neighbours(cand);
		if (nc <= 0) bin = 0;
		else {
			int min_dist = 2 << 20;
			bin = first;
			for (int a = 0; a < md.nb; a++) {
				int d = decode_array<PW>(md, gf, pf, a);
				if (d <= 0) continue;
				int cur = 2 << 20;
				for (int c = 0; c < 8; c++)
					if (c < nc) { int dd = d > cand[c] ? d - cand[c] : cand[c] - d; cur = dd < cur ? dd : cur; }
				if (min_dist > cur) { min_dist = cur; bin = d; }
			}
		}
	}
	return bin < (1 << md.nh) ? (int)md.mean_of_bin[bin] : 0;       // a Bloom-class count beyond the table: the reference's map yields 0 (occu_bin.hpp:79-83)
}

// get_candidates (kmodel.hpp:326-342) on a packed k-mer; returns -2 when the neighbour contributes nothing
template <int W> __device__ __forceinline__ int neighbour_candidate(const ModelDev &md, u64 *nv)
{
	min_kmer<W>(nv, md.k);
	int r = rest_check<W>(md, nv);
	if (r > 0) return (int)md.bin_of_occ[r];
	Aligned<W> al = left_align<W>(nv, md.k);
	Premixed<W> pf = premix_string<W>(al, md.gfull);
	Premixed<W> pb = premix_string<W>(drop_first_base<W>(al), md.gback);
	return candidate_from_filters<W>(md, md.gfull, md.gback, pf, pb);
}

// get_neighbor_kmer_bin (kmodel.hpp:344-359): 4 successors then 4 predecessors, bases in ACGT order
template <int W> __device__ __forceinline__ int neighbour_bins(const ModelDev &md, const u64 *v, int *cand)
{
	int nc = 0;
	const int k = md.k;
	for (int x = 0; x < 8; x++) {
		u64 nv[W];
		if (x < 4) {              // drop the first base, append X
			if (W == 1) nv[0] = ((v[0] << 2) | (u64)x) & (k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1));
			else {
				nv[0] = ((v[0] << 2) | (v[W - 1] >> 62)) & (k == 64 ? ~0ULL : ((1ULL << (2 * k - 64)) - 1));
				nv[W - 1] = (v[W - 1] << 2) | (u64)x;
			}
		} else {                  // prepend X, drop the last base
			u64 X = (u64)(x - 4);
			if (W == 1) nv[0] = (v[0] >> 2) | (X << (2 * (k - 1)));
			else {
				nv[W - 1] = (v[W - 1] >> 2) | (v[0] << 62);
				nv[0] = (v[0] >> 2) | (X << (2 * (k - 1) - 64));
			}
		}
		int c = neighbour_candidate<W>(md, nv);
		if (c != -2) cand[nc++] = c;
	}
	return nc;
}

// KModel::kmer_to_occ (kmodel.hpp:100-116) on packed k-mers
// ACCT (never the timed kernel: kmx_set_profile(m, 2)): counts the queries that enter the neighbour disambiguation
// (get_neighbor_kmer_bin, kmodel.hpp:344-359: up to 8 nested lookups in one lane) in acct[0]
template <int W, bool ACCT> __global__ __launch_bounds__(256) void k_query(ModelDev md, const u64 *kmers, u64 n, int *out, u64 *acct)
{
	const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
	if (q >= n) return;
	u64 v[W];
	load_kmer<W>(kmers, q, v);
	if (md.k & 31) v[0] &= (1ULL << (2 * (md.k & 31))) - 1;          // bits above 2k are not part of a packed k-mer (a caller's stray bits would index past the tables)
	min_kmer<W>(v, md.k);
	int occ = rest_check<W>(md, v);
	if (occ != 0) { out[q] = occ; return; }
	Aligned<W> al = left_align<W>(v, md.k);
	Premixed<W> pf = premix_string<W>(al, md.gfull);
	Premixed<W> pb = premix_string<W>(drop_first_base<W>(al), md.gback);
	out[q] = occ_from_filters<W>(md, md.gfull, md.gback, pf, pb, [&](int *cand) { if (ACCT) atomicAdd(acct, 1ULL); return neighbour_bins<W>(md, v, cand); });
}

// ------------------------------------------------------------------------------------------ query on raw strings
// kmer_to_occ for strings the packed form cannot hold (characters outside ACGT, length != k): the reference hashes
// the bytes it is given, treats unknown characters as 'A' only where it converts to 2-bit codes (tools.hpp:63-76,
// rest.hpp:22-34) and canonicalises through one u64 (tools.hpp:160-167).  Same decision tree, on byte strings of one
// common length L <= 64 (string byte i lives in b[i >> 3], bits 8*(i & 7)).
struct AStr { u64 b[8]; };

__device__ __forceinline__ u32 code_of_byte(u32 c) { return c == 'C' ? 1u : (c == 'G' ? 2u : (c == 'T' ? 3u : 0u)); }

// 2-bit codes of all L characters as a 128-bit integer (first character most significant)
__device__ __forceinline__ void astr_codes(const AStr &s, int L, u64 &hi, u64 &lo)
{
	hi = lo = 0;
#pragma unroll
	for (int w = 0; w < 8; w++)
#pragma unroll
		for (int j = 0; j < 8; j++)
			if (8 * w + j < L) {
				const u32 c = code_of_byte((u32)(s.b[w] >> (8 * j)) & 0xFFu);
				hi = (hi << 2) | (lo >> 62);
				lo = (lo << 2) | c;
			}
}
// zero everything from byte L on
__device__ __forceinline__ void astr_clip(AStr &s, int L)
{
#pragma unroll
	for (int w = 0; w < 8; w++) {
		const int keep = L - 8 * w;                             // bytes of this word that belong to the string
		if (keep <= 0) s.b[w] = 0;
		else if (keep < 8) s.b[w] &= (1ULL << (8 * keep)) - 1;
	}
}
__device__ __forceinline__ AStr astr_drop_first(const AStr &s)              // s[1:]
{
	AStr r;
#pragma unroll
	for (int w = 0; w < 8; w++) r.b[w] = (s.b[w] >> 8) | (w + 1 < 8 ? s.b[w + 1] << 56 : 0);
	return r;
}
__device__ __forceinline__ AStr astr_shift_in_front(const AStr &s, u32 c)    // c + s
{
	AStr r;
#pragma unroll
	for (int w = 0; w < 8; w++) r.b[w] = (s.b[w] << 8) | (w ? s.b[w - 1] >> 56 : (u64)c);
	return r;
}
__device__ __forceinline__ void astr_set(AStr &s, int i, u32 c)
{
#pragma unroll
	for (int w = 0; w < 8; w++)
		if ((i >> 3) == w) s.b[w] = (s.b[w] & ~(0xFFULL << (8 * (i & 7)))) | ((u64)c << (8 * (i & 7)));
}
// Tools::get_min_kmer on a byte string (tools.hpp:160-167, quirk Q4 for L > 32)
__device__ __forceinline__ void astr_min_kmer(AStr &s, int L)
{
	u64 hi, u;
	astr_codes(s, L, hi, u);
	const u64 r32 = rev2_u64(~u);
	u64 rc;
	if (L <= 32) rc = r32 >> (64 - 2 * L);
	else { const int sh = 2 * (L - 32); rc = sh >= 64 ? ~0ULL : ((r32 << sh) | ((1ULL << sh) - 1)); }
	if (u <= rc) return;
#pragma unroll
	for (int w = 7; w >= 0; w--) {                             // uint64_to_string: last character from the low bits
		u64 word = 0;
#pragma unroll
		for (int j = 7; j >= 0; j--)
			if (8 * w + j < L) {
				const u32 c = (u32)(rc & 3);
				word |= (u64)(c == 0 ? 'A' : (c == 1 ? 'C' : (c == 2 ? 'G' : 'T'))) << (8 * j);
				rc >>= 2;
			}
		s.b[w] = word;
	}
}
__device__ __forceinline__ Premixed<2> astr_premix(const AStr &s, const StrGeom g)
{
	Premixed<2> p;
	p.tail = 0;
#pragma unroll
	for (int b = 0; b < 8; b++) {
		p.blk[b] = premix(s.b[b]);
		if (b == g.nblk) p.tail = g.rem ? (s.b[b] & ((1ULL << (8 * g.rem)) - 1)) : 0;
	}
	return p;
}
// KRestData::check_kmer on a byte string: 0 unless the length is the table's k (rest.hpp:224-226, quirk Q7)
template <int W> __device__ __forceinline__ int astr_rest_check(const ModelDev &md, const AStr &s, int L)
{
	if (L != md.k) return 0;
	u64 hi, lo, v[W];
	astr_codes(s, L, hi, lo);
	v[W - 1] = lo;
	if (W == 2) v[0] = hi;
	return rest_check<W>(md, v);
}
template <int W> __device__ __forceinline__ int astr_candidate(const ModelDev &md, const StrGeom gf, const StrGeom gb, AStr t, int L)
{
	astr_min_kmer(t, L);
	const int r = astr_rest_check<W>(md, t, L);
	if (r > 0) return (int)md.bin_of_occ[r];
	Premixed<2> pf = astr_premix(t, gf);
	Premixed<2> pb = astr_premix(astr_drop_first(t), gb);
	return candidate_from_filters<2>(md, gf, gb, pf, pb);
}

template <int W> __global__ __launch_bounds__(256) void k_query_ascii(ModelDev md, StrGeom gf, StrGeom gb, int L, const unsigned char *strs, int stride, u64 n, int *out)
{
	const u64 q = (u64)blockIdx.x * 256 + threadIdx.x;
	if (q >= n) return;
	AStr s;
	const unsigned char *p = strs + q * (u64)stride;
#pragma unroll
	for (int w = 0; w < 8; w++) {
		u64 word = 0;
#pragma unroll
		for (int j = 0; j < 8; j++)
			if (8 * w + j < L) word |= (u64)p[8 * w + j] << (8 * j);
		s.b[w] = word;
	}
	astr_min_kmer(s, L);
	const int occ = astr_rest_check<W>(md, s, L);
	if (occ != 0) { out[q] = occ; return; }
	Premixed<2> pf = astr_premix(s, gf);
	Premixed<2> pb = astr_premix(astr_drop_first(s), gb);
	out[q] = occ_from_filters<2>(md, gf, gb, pf, pb, [&](int *cand) {
		int nc = 0;
		const AStr t1 = astr_drop_first(s);                      // kmer.substr(1): still holds s[L-1] at byte L-2
		for (int x = 0; x < 8; x++) {
			const u32 X = (u32)"ACGT"[x & 3];
			AStr t;
			if (x < 4) { t = t1; astr_set(t, L - 1, X); }      // drop the first character, append X
			else { t = astr_shift_in_front(s, X); astr_clip(t, L); }   // prepend X, drop the last character
			const int c = astr_candidate<W>(md, gf, gb, t, L);
			if (c != -2) cand[nc++] = c;
		}
		return nc;
	});
}

// ------------------------------------------------------------------------------------------ KMC listing on the device
// CKMCFile::ReadNextKmer (kmc_file.cpp:428-515) for a whole batch of records at once: the host only moves the raw record
// bytes (pinned hipMemcpyAsync); prefix lookup, byte swaps and packing happen here.  Every record of the batch must be
// listed (count within the header's [min_count, max_count]); the caller checks that in pass 1.
__device__ __forceinline__ u64 lut_last_le(const u64 *lut, u64 lo, u64 hi, u64 rec)      // largest idx in [lo, hi] with lut[idx] <= rec
{
	while (lo < hi) {
		const u64 mid = lo + (hi - lo + 1) / 2;
		if (lut[mid] <= rec) lo = mid; else hi = mid - 1;
	}
	return lo;
}
template <int W> __global__ __launch_bounds__(256) void k_kmc_decode(KmcDecode d, u64 rec0, u64 n, u64 *kmers, u32 *counts)
{
	__shared__ u64 s_lo, s_hi;
	const u64 j0 = (u64)blockIdx.x * 256;
	if (threadIdx.x < 2) {                                           // the LUT entries of the workgroup's first and last record
		u64 j = j0 + (threadIdx.x ? 255 : 0);
		if (j >= n) j = n - 1;
		const u64 idx = lut_last_le(d.lut, 0, d.n_lut - 1, rec0 + j);
		if (threadIdx.x) s_hi = idx; else s_lo = idx;
	}
	__syncthreads();
	const u64 j = j0 + threadIdx.x;
	if (j >= n) return;
	const u64 prefix = lut_last_le(d.lut, s_lo, s_hi, rec0 + j) & d.prefix_mask;
	const unsigned char *r = d.recs + j * d.rec_bytes;
	u32 c = 0;
	for (u32 b = 0; b < d.cnt_bytes; b++) c |= (u32)r[d.suf_bytes + b] << (8 * b);
	u64 hi = 0, lo = prefix;
	for (u32 b = 0; b < d.suf_bytes; b++) { hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | r[b]; }
	if (W == 1) kmers[j] = lo;
	else { kmers[2 * j] = hi; kmers[2 * j + 1] = lo; }
	counts[j] = c;
}

// ------------------------------------------------------------------------------------------ layout conversion
// on-disk value/tag bytes <-> cells (kmodel.hpp:199-201, :227-229).  One thread per cell (2 bytes of each).
__global__ __launch_bounds__(256) void k_cells_from_disk(const unsigned char *val, const unsigned char *tag, u64 nbytes, cell_t *cells, u64 ncells)
{
	u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
	if (c >= ncells) return;
	u64 b0 = 2 * c, b1 = 2 * c + 1;
	u32 v = (u32)val[b0] | (b1 < nbytes ? (u32)val[b1] << 8 : 0);
	u32 t = (u32)tag[b0] | (b1 < nbytes ? (u32)tag[b1] << 8 : 0);
	cells[c] = v | (t << 16);
}
// which: 0 value, 1 tag, 2 insert-time scratch left in the model (there is none any more: always zero)
__global__ __launch_bounds__(256) void k_cells_to_disk(const cell_t *cells, u64 ncells, u64 nbytes, int which, unsigned char *out)
{
	u64 c = (u64)blockIdx.x * 256 + threadIdx.x;
	if (c >= ncells) return;
	const cell_t cell = cells[c];
	u32 x = which == 0 ? (cell & 0xFFFF) : which == 1 ? (cell >> 16) : 0;
	out[2 * c] = (unsigned char)(x & 0xFF);
	if (2 * c + 1 < nbytes) out[2 * c + 1] = (unsigned char)(x >> 8);
}

// ------------------------------------------------------------------------------------------ KAT surface
template <int W> __global__ void k_debug_hash(int k, const u64 *kmers, u64 n, const u32 *seeds, int n_seeds, int whole, u64 *out)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	u64 v[W];
	load_kmer<W>(kmers, i, v);
	Aligned<W> al = left_align<W>(v, k);
	StrGeom g;
	int len = whole ? k : k - 2;
	g.nblk = len / 8; g.rem = len & 7; g.lenm = (u64)len * MURMUR_M;
	Premixed<W> pm = premix_string<W>(whole ? al : drop_first_base<W>(al), g);
	for (int s = 0; s < n_seeds; s++) out[i * n_seeds + s] = murmur_seeded<W>(pm, g, seeds[s]);
}
__global__ void k_debug_mod(const u64 *h, u64 n, ModU64 md, u64 *out)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = mod_u64(h[i], md);
}
template <int W> __global__ void k_debug_min_kmer(int k, const u64 *kmers, u64 n, u64 *out)
{
	u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	u64 v[W];
	load_kmer<W>(kmers, i, v);
	min_kmer<W>(v, k);
	store_kmer<W>(out, i, v);
}

// ------------------------------------------------------------------------------------------ microbenchmarks
// The random-access ceiling the roofline fraction is quoted against: each lane issues 8
// independent 8-byte touches at splitmix64 addresses, like one k-mer's touches on one array.
__device__ __forceinline__ u64 splitmix(u64 z)
{
	z += 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_micro_gather(const u64 *buf, u64 ncell, u64 n_lanes, u64 salt, u64 *sink)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
	u64 acc = 0;
#pragma unroll
	for (int j = 0; j < 8; j++) acc ^= buf[splitmix(i * 8 + j + salt) % ncell];
	if (acc == 0x123456789ULL) *sink = acc;
}
__global__ __launch_bounds__(256) void k_micro_gather32(const u32 *buf, u64 nword, u64 n_lanes, u64 salt, u64 *sink)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
	u32 acc = 0;
#pragma unroll
	for (int j = 0; j < 8; j++) acc ^= buf[splitmix(i * 8 + j + salt) % nword];
	if (acc == 0x12345678u) *sink = acc;
}
// variants of the 4-byte gather: 1 non-temporal (no allocation in the caches on the way), 2 agent-scope relaxed
// (global_load sc1: bypasses this CU's L1), 3 plain with 16 loads in flight per lane instead of 8
template <int V> __global__ __launch_bounds__(256) void k_micro_gather32v(const u32 *buf, u64 nword, u64 n_lanes, u64 salt, u64 *sink)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
	u32 acc = 0;
	constexpr int N = V == 3 ? 16 : 8;
	if (V == 3 && (i & 1)) return;                                  // half the lanes, twice the loads: same number of touches
#pragma unroll
	for (int j = 0; j < N; j++) {
		const u32 *p = buf + splitmix(i * 8 + j + salt) % nword;
		acc ^= V == 1 ? __builtin_nontemporal_load(p) : (V == 2 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p);
	}
	if (acc == 0x12345678u) *sink = acc;
}
// 32-bit atomic ORs with only ~3/8 of the lanes active in each instruction (what a commit looks like: a lane is a list
// slot, 55 % are candidates, 74 % of their positions untagged); touches = 3/8 of the nominal count
__global__ __launch_bounds__(256) void k_micro_atomic_or32_sparse(u32 *buf, u64 nword, u64 n_lanes, u64 salt)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		u64 r = splitmix(i * 8 + j + salt);
		if (((r >> 40) & 7) < 3) atomicOr(buf + r % nword, 1u << (r >> 59));
	}
}
__global__ __launch_bounds__(256) void k_micro_atomic_or(u64 *buf, u64 ncell, u64 n_lanes, u64 salt)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		u64 r = splitmix(i * 8 + j + salt);
		atomicOr(buf + r % ncell, 1ULL << (r >> 58));
	}
}

__global__ __launch_bounds__(256) void k_micro_byte_store(unsigned char *buf, u64 nbytes, u64 n_lanes, u64 salt)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
#pragma unroll
	for (int j = 0; j < 8; j++) buf[splitmix(i * 8 + j + salt) % nbytes] = (unsigned char)(salt | 1);
}
__global__ __launch_bounds__(256) void k_micro_byte_gather(const unsigned char *buf, u64 nbytes, u64 n_lanes, u64 salt, u64 *sink)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
	u32 acc = 0;
#pragma unroll
	for (int j = 0; j < 8; j++) acc += buf[splitmix(i * 8 + j + salt) % nbytes];
	if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_micro_store8(u64 *buf, u64 ncell, u64 n_lanes, u64 salt)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
#pragma unroll
	for (int j = 0; j < 8; j++) buf[splitmix(i * 8 + j + salt) % ncell] = salt;
}
__global__ __launch_bounds__(256) void k_micro_atomic_or32(u32 *buf, u64 nword, u64 n_lanes, u64 salt)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		u64 r = splitmix(i * 8 + j + salt);
		atomicOr(buf + r % nword, 1u << (r >> 59));
	}
}

__global__ __launch_bounds__(256) void k_micro_atomic_or_wg(u64 *buf, u64 ncell, u64 n_lanes, u64 salt)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		u64 r = splitmix(i * 8 + j + salt);
		__hip_atomic_fetch_or(buf + r % ncell, 1ULL << (r >> 58), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}
__global__ __launch_bounds__(256) void k_micro_atomic_or_ret(u64 *buf, u64 ncell, u64 n_lanes, u64 salt, u64 *sink)
{
	u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_lanes) return;
	u64 acc = 0;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		u64 r = splitmix(i * 8 + j + salt);
		acc ^= atomicOr(buf + r % ncell, 1ULL << (r >> 58));
	}
	if (acc == 0x123456789ULL) *sink = acc;
}

// ------------------------------------------------------------------------------------------ launchers
namespace kmxk {

#define DISPATCH_W(W_, ...)                     \
	do {                                        \
		if ((W_) == 1) { constexpr int W = 1; __VA_ARGS__; } \
		else { constexpr int W = 2; __VA_ARGS__; }           \
	} while (0)
#define DISPATCH_W_NH(W_, nh_, ...)             \
	do {                                        \
		if ((W_) == 1) { constexpr int W = 1; if ((nh_) <= 8) { constexpr int NHM = 8; __VA_ARGS__; } else { constexpr int NHM = 16; __VA_ARGS__; } } \
		else { constexpr int W = 2; if ((nh_) <= 8) { constexpr int NHM = 8; __VA_ARGS__; } else { constexpr int NHM = 16; __VA_ARGS__; } } \
	} while (0)

static inline int words(const ModelDev &md) { return (md.k + 31) / 32; }
void bs_apply(const BitScatter &bs, hipStream_t st);

void histogram(const u32 *counts, u64 n, int ci, int cs, int bf_num, u64 *n_bf, u64 *stats, hipStream_t st)
{
	if (!n) return;
	u64 blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(k_histogram, dim3((unsigned)blocks), dim3(256), 0, st, counts, n, ci, cs, bf_num, n_bf, stats);
}

// returns the number of classification tiles
int classify_tiles(u64 n) { return (int)((n + CLS_TILE - 1) / CLS_TILE); }

// Front end of `n` k-mers cut into chunks of `chunk` k-mers (a multiple of CLS_TILE): the Bloom-class k-mers are inserted
// (directly, or a chunk at a time through the BitScatter `bs`, swept after every `sweep_every` chunks and at the end) and
// the coupled-array k-mers are counted per tile; one small scan per chunk turns the counts into offsets relative to the
// chunk start and the chunk's total (totals[c]).
void classify_count(const ModelDev &md, const u64 *kmers, const u32 *counts, u64 n, u64 chunk, int *tile_cnt, int *tile_off, int *totals, u64 *stats, const BitScatter &bs, int sweep_every, hipStream_t st, KernelProf *prof)
{
	if (!n) return;
	const int tiles = classify_tiles(n), tiles_per_chunk = (int)(chunk / CLS_TILE);
	const int W_ = words(md);
	KPROF_BEGIN(prof, KC_CLASSIFY, st);
	const u64 piece = md.bloom_direct ? n : chunk;                  // k-mers per launch: what the bins of the BitScatter take (a bin that fills up falls back to atomics)
	int since = 0;
	for (u64 lo = 0; lo < n; lo += piece) {
		const u64 c = n - lo < piece ? n - lo : piece;
		const int t0 = (int)(lo / CLS_TILE), nt = classify_tiles(c);
		DISPATCH_W_NH(W_, md.nh, hipLaunchKernelGGL((k_classify_count<W, NHM>), dim3(nt), dim3(256), 0, st, md, kmers + lo * W_, counts + lo, c, tile_cnt + t0, stats, bs));
		if (!md.bloom_direct && (++since >= sweep_every || lo + piece >= n)) { bs_apply(bs, st); since = 0; }
	}
	KPROF_END(prof, st);
	for (int c = 0, t0 = 0; t0 < tiles; c++, t0 += tiles_per_chunk) {
		const int nt = tiles - t0 < tiles_per_chunk ? tiles - t0 : tiles_per_chunk;
		hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, (const int *)(tile_cnt + t0), tile_off + t0, nt, totals + c);
	}
}

void classify_scatter(const ModelDev &md, const u64 *kmers, const u32 *counts, u64 n, const int *tile_off, u64 *stg_kmers, u32 *stg_counts, u64 stg_base, hipStream_t st)
{
	int tiles = classify_tiles(n);
	DISPATCH_W(words(md), hipLaunchKernelGGL(k_classify_scatter<W>, dim3(tiles), dim3(256), 0, st, md, kmers, counts, n, tile_off, stg_kmers, stg_counts, stg_base));
}

void block_init(const BlockDev &bd, int nb, int pp, int n_in_block, hipStream_t st)
{
	hipLaunchKernelGGL(k_block_init, dim3(KMX_BUCKET / 256, nb), dim3(256), 0, st, bd, nb, pp, n_in_block);
}

// lists shrink by roughly half per round; the kernels are grid-stride, so a smaller grid is only a speed choice
static inline int round_gx(int t) { return (int)(KMX_BUCKET / 256) >> (t < 4 ? t : 4); }

// One round t (list parity pp) of one block: A check (+ the commit of the previous round beside it when
// KMX_ROUND_PENDING: its winners are still uncommitted), D detect, F file, `nsub` grid-wide ordered sub-rounds, finisher,
// reorder.  `epoch` advances.  The winners of THIS round stay uncommitted: the caller passes KMX_ROUND_PENDING to the next
// round, or calls commit_flush.  The single-workgroup finisher decides whatever the sub-rounds leave (everything when
// nsub == 0), so nsub only trades launches for finisher iterations; the host picks it from the contention it has observed.
// t_prev: the round index of the pending commit (t - 1, or nb - 1 of the previous block).
// PROBE (KMX_PREGATHER_PROBE under KMX_TEST_HOOKS; never in a product build's path): how much of the next round's gathers could hide
// under this round's ordered chain?  The k-mers that a round's check has just failed on a set tag -- ~95 % of the next round's
// attempts -- could fetch their next array's cells while detect -> file -> finisher -> reorder leave the memory system idle.
// This kernel only IMITATES that traffic: as many random 4-byte loads on the next array as those k-mers would issue (nh * 13/16,
// the staged fetch stops early), on a side stream beside the chain, results discarded.  If a build takes no longer with it, the
// real thing would take the same loads out of the fused launches; if the build grows by what the kernel takes, nothing can hide.
__global__ __launch_bounds__(256) void k_probe_pregather(ModelDev md, BlockDev bd, int t, int pp, u64 ncells, u32 *sink)
{
	__shared__ int s_sum;
	const int i = blockIdx.y;
	if (threadIdx.x == 0) s_sum = 0;
	__syncthreads();
	if (threadIdx.x < (int)KMX_NTILES) atomicAdd(&s_sum, bd.tile_cnt[pp][i * KMX_NTILES + threadIdx.x]);     // the failures counted so far in this round
	__syncthreads();
	const u64 total = (u64)s_sum * (u64)md.nh * 13 / 16;
	const cell_t *cells = md.cells[(i + t + 1) % md.nb];
	u32 acc = 0;
	for (u64 g = (u64)blockIdx.x * 256 + threadIdx.x; g < total; g += (u64)gridDim.x * 256) {
		u64 z = (g + ((u64)i << 40) + ((u64)t << 48)) * 0x9E3779B97F4A7C15ULL;
		z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 32;
		acc ^= cells[z % ncells];
	}
	if (acc == 0x12345u) sink[0] = acc;                               // (keeps the loads)
}

void round(const ModelDev &md, const BlockDev &bd, int t, int pp, int nsub, u64 *epoch, int flags, hipStream_t st, KernelProf *prof, const KmbackJob *job, const BitScatter *kmb, const RoundProbe *probe)
{
	const int nb = md.nb;
	if (probe && probe->on) hipStreamWaitEvent(st, probe->done, 0);   // the imitation of the last round's pre-gather ends before this round's check starts
	if (nsub < 0) nsub = 0;
	if (nsub > KMX_MAX_NSUB) nsub = KMX_MAX_NSUB;
	const int gx = round_gx(t);
	const dim3 grid(gx, nb), blk(256), sgrid(SLOW_BLOCKS, nb);
	const u64 eb = (*epoch)++;                                 // k_round_file's reservations
	const int pending = (flags & KMX_ROUND_PENDING) ? 1 : 0;
	if (pending) {
		KPROF_BEGIN(prof, KC_COMMIT_CHECK, st);
		const int gc = round_gx(t > 0 ? t - 1 : nb - 1);
		if (prof && prof->count) DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_round_commit_check<W, NHM, true>), dim3(nb * (gx + gc), 1), blk, 0, st, md, bd, t, pp, gx, gc));   // + the issue counters
		else DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_round_commit_check<W, NHM, false>), dim3(nb * (gx + gc), 1), blk, 0, st, md, bd, t, pp, gx, gc));
		KPROF_END(prof, st);
	} else {
		KPROF_BEGIN(prof, KC_CHECK_CLAIM, st);
		DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_round_check_emit<W, NHM>), grid, blk, 0, st, md, bd, t, pp));
		KPROF_END(prof, st);
	}
	if (probe && probe->on) {
		hipEventRecord(probe->fork, st);
		hipStreamWaitEvent(probe->side, probe->fork, 0);
		hipLaunchKernelGGL(k_probe_pregather, dim3(256, nb), dim3(256), 0, probe->side, md, bd, t, pp, (u64)((md.km_mod.d + 15) / 16), probe->sink);
		hipEventRecord(probe->done, probe->side);
	}
	KPROF_BEGIN(prof, KC_DETECT, st);
	const int keep_own = (flags & KMX_ROUND_KEEP) ? 1 : 0;
	const int late = t >= 2 ? 1 : 0;
	if (late && (flags & KMX_ROUND_SMALL_DETECT)) {                  // the late rounds' few hundred tuples per bin: small tables, every bin resident at once
		if (md.nh <= 8) hipLaunchKernelGGL((k_round_detect<8, 256, KMX_CL_TBITS_SMALL>), dim3(KMX_CL_BINS(8), nb), dim3(256), 0, st, bd, nb, pp, pending, keep_own, late);
		else hipLaunchKernelGGL((k_round_detect<16, 256, KMX_CL_TBITS_SMALL>), dim3(KMX_CL_BINS(16), nb), dim3(256), 0, st, bd, nb, pp, pending, keep_own, late);
	} else if (md.nh <= 8) hipLaunchKernelGGL((k_round_detect<8, 1024, KMX_CL_TBITS(8)>), dim3(KMX_CL_BINS(8), nb), dim3(1024), 0, st, bd, nb, pp, pending, keep_own, late);
	else hipLaunchKernelGGL((k_round_detect<16, 1024, KMX_CL_TBITS(16)>), dim3(KMX_CL_BINS(16), nb), dim3(1024), 0, st, bd, nb, pp, pending, keep_own, late);
	KPROF_END(prof, st);
	KPROF_BEGIN(prof, KC_FILE, st);
	DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_round_file<W, NHM>), dim3(KMX_FILE_WGS, nb), dim3(1024), 0, st, md, bd, pp, eb));
	KPROF_END(prof, st);
	KPROF_BEGIN(prof, KC_SLOW, st);
	const bool legacy0 = flags & KMX_ROUND_RESOLVE_GATHER;      // test hook: the gathering resolve kernel for level 0 too
	for (int s = 0; s < nsub; s++) {
		u64 e = eb;
		if (s > 0) {
			e = (*epoch)++;
			DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_slow_reserve<W, NHM>), sgrid, blk, 0, st, md, bd, t, pp, s, e));
		}
		if (s == 0 && !legacy0) DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_slow_resolve0<W, NHM>), sgrid, blk, 0, st, md, bd, t, pp, e));
		else DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_slow_resolve<W, NHM>), sgrid, blk, 0, st, md, bd, t, pp, s, e));
	}
	const int force_global = (flags & KMX_ROUND_FIN_GLOBAL) ? 1 : 0;   // test hook: the finisher's global-memory path for every set
	const u64 e0 = *epoch;
	*epoch += (1ULL << 19);                                    // the finisher may use up to |U| <= 2^18 epochs
	{
		KmbackJob none = {nullptr, nullptr, 0, 0, 0};
		const KmbackJob &jb = job && job->n_lists > 0 && kmb ? *job : none;
		const int riders = jb.n_lists * (int)(KMX_BUCKET / (1024 * KMX_FIN_RIDER_KPT(md.nh <= 8 ? 8 : 16)));   // one workgroup per 2048 (1024) k-mers of the hosted lists
		DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_slow_finish<W, NHM>), dim3(nb + riders, 1), dim3(1024), 0, st, md, bd, t, pp, nsub, eb, e0, force_global, jb, kmb ? *kmb : BitScatter()));
	}
	KPROF_END(prof, st);
	KPROF_BEGIN(prof, KC_REORDER, st);
	DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_reorder<W, NHM>), dim3(KMX_NTILES + KMX_APPLY_WGS, nb), dim3(256), 0, st, md, bd, t, pp, nsub & 1));
	KPROF_END(prof, st);
}

// the commit of round t (parity pp) in a launch of its own: end of the build, or a caller that needs the arrays up to date
void commit_flush(const ModelDev &md, const BlockDev &bd, int t, int pp, hipStream_t st, KernelProf *prof)
{
	KPROF_BEGIN(prof, KC_VERIFY_COMMIT, st);
	if (md.nh <= 8) hipLaunchKernelGGL((k_round_commit<8>), dim3(round_gx(t), md.nb), dim3(256), 0, st, md, bd, t, pp);
	else hipLaunchKernelGGL((k_round_commit<16>), dim3(round_gx(t), md.nb), dim3(256), 0, st, md, bd, t, pp);
	KPROF_END(prof, st);
}

// lists [i0, i0 + n_lists) of the block go to the rest table
void rest_append(const ModelDev &md, const BlockDev &bd, int pp, int i0, int n_lists, u64 *rest_kmers, int *rest_counts, unsigned long long *rest_n, u64 *stale_kmers, int *stale_counts, u64 *feedback, hipStream_t st, int istride)
{
	DISPATCH_W(words(md), hipLaunchKernelGGL(k_rest_append<W>, dim3(KMX_BUCKET / 256, n_lists), dim3(256), 0, st, bd, pp, i0, istride, rest_kmers, rest_counts, rest_n, stale_kmers, stale_counts, feedback));
}

// n_in_block < 0: the slots round t inserted (list[pp], status); >= 0: every k-mer of the block that is not a survivor.
// Lists [i0, i0 + n_lists) of the block whose k-mers / survivor flags are given (the current block's: bd.kmers, bd.surv).
void kmback_emit(const ModelDev &md, const BlockDev &bd, const u64 *kmers, const unsigned char *surv, int i0, int n_lists, int t, int pp, int n_in_block, const BitScatter &bs, hipStream_t st, int istride)
{
	if (!md.km_back_mod.d || n_lists <= 0) return;
	const int gx = n_in_block >= 0 ? KMX_BUCKET / 1024 : (KMX_BUCKET / 1024) >> (t < 4 ? t : 4);   // 4 (2) slots per thread; lists shrink round by round
	const dim3 grid(gx, n_lists), blk(256);
	if (words(md) == 1) {
		if (md.nh <= 8) hipLaunchKernelGGL((k_kmback_emit<1, 8, 4>), grid, blk, 0, st, md, bd, kmers, surv, i0, istride, pp, n_in_block, bs);
		else hipLaunchKernelGGL((k_kmback_emit<1, 16, 2>), grid, blk, 0, st, md, bd, kmers, surv, i0, istride, pp, n_in_block, bs);
	} else {
		if (md.nh <= 8) hipLaunchKernelGGL((k_kmback_emit<2, 8, 4>), grid, blk, 0, st, md, bd, kmers, surv, i0, istride, pp, n_in_block, bs);
		else hipLaunchKernelGGL((k_kmback_emit<2, 16, 2>), grid, blk, 0, st, md, bd, kmers, surv, i0, istride, pp, n_in_block, bs);
	}
}
void bs_apply(const BitScatter &bs, hipStream_t st)
{
	if (!bs.cap2) { hipLaunchKernelGGL(k_bs_apply, dim3(BS_BINS), dim3(1024), 0, st, bs); return; }
	const unsigned per_bin = (bs.cap + 256 * BS_SPLIT_K - 1) / (256 * BS_SPLIT_K);
	hipLaunchKernelGGL(k_bs_split, dim3(per_bin < 64 ? per_bin : 64, BS_BINS), dim3(256), 0, st, bs);
	hipLaunchKernelGGL(k_bs_apply2, dim3(1u << (bs.wshift - bs.tlog2), BS_BINS), dim3(1024), 0, st, bs);
}

void kmc_decode(const KmcDecode &d, int W_, u64 rec0, u64 n, u64 *kmers, u32 *counts, hipStream_t st)
{
	if (!n) return;
	DISPATCH_W(W_, hipLaunchKernelGGL(k_kmc_decode<W>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, rec0, n, kmers, counts));
}

void ring_import(const ModelDev &md, const BlockDev &bd, int pp, const RingLists &rl, u64 *stg_kmers, u32 *stg_counts, hipStream_t st)
{
	DISPATCH_W(words(md), hipLaunchKernelGGL(k_ring_import<W>, dim3(KMX_BUCKET / 256, md.nb), dim3(256), 0, st, bd, md.nb, pp, rl, stg_kmers, stg_counts));
}
void ring_export(const ModelDev &md, const BlockDev &bd, int pp, const RingLists &rl, u64 *feedback, hipStream_t st)
{
	DISPATCH_W(words(md), hipLaunchKernelGGL(k_ring_export<W>, dim3(KMX_BUCKET / 256, md.nb), dim3(256), 0, st, bd, pp, rl, feedback));
}
void or_words(u32 *dst, const u32 *src, u64 n, hipStream_t st)
{
	if (!n) return;
	u64 blocks = (n + 1023) / 1024;
	if (blocks > 65536) blocks = 65536;
	hipLaunchKernelGGL(k_or_words, dim3((unsigned)blocks), dim3(256), 0, st, dst, src, n);
}

// acct != null: the accounting variant (queries that enter the neighbour disambiguation are counted there)
void query(const ModelDev &md, const u64 *kmers, u64 n, int *out, hipStream_t st, KernelProf *prof, u64 *acct)
{
	if (!n) return;
	KPROF_BEGIN(prof, KC_QUERY, st);
	if (acct) DISPATCH_W(words(md), hipLaunchKernelGGL((k_query<W, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, md, kmers, n, out, acct));
	else DISPATCH_W(words(md), hipLaunchKernelGGL((k_query<W, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, md, kmers, n, out, (u64 *)nullptr));
	KPROF_END(prof, st);
}

void query_ascii(const ModelDev &md, int L, const unsigned char *strs, int stride, u64 n, int *out, hipStream_t st)
{
	if (!n) return;
	const StrGeom gf = make_geom(L), gb = make_geom(L >= 2 ? L - 2 : 0);
	DISPATCH_W(words(md), hipLaunchKernelGGL(k_query_ascii<W>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, md, gf, gb, L, strs, stride, n, out));
}

void cells_from_disk(const unsigned char *val, const unsigned char *tag, u64 nbytes, cell_t *cells, u64 ncells, hipStream_t st)
{
	if (!ncells) return;
	hipLaunchKernelGGL(k_cells_from_disk, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st, val, tag, nbytes, cells, ncells);
}
void cells_to_disk(const cell_t *cells, u64 ncells, u64 nbytes, int which, unsigned char *out, hipStream_t st)
{
	if (!ncells) return;
	hipLaunchKernelGGL(k_cells_to_disk, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st, cells, ncells, nbytes, which, out);
}

void debug_hash(int k, const u64 *kmers, u64 n, const u32 *seeds, int n_seeds, int whole, u64 *out, hipStream_t st)
{
	DISPATCH_W((k + 31) / 32, hipLaunchKernelGGL(k_debug_hash<W>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k, kmers, n, seeds, n_seeds, whole, out));
}
void debug_mod(const u64 *h, u64 n, u64 d, u64 *out, hipStream_t st)
{
	hipLaunchKernelGGL(k_debug_mod, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h, n, make_mod(d), out);
}
void debug_min_kmer(int k, const u64 *kmers, u64 n, u64 *out, hipStream_t st)
{
	DISPATCH_W((k + 31) / 32, hipLaunchKernelGGL(k_debug_min_kmer<W>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k, kmers, n, out));
}

void micro(int mode, u64 *buf, u64 ncell, u64 n_lanes, u64 salt, u64 *sink, hipStream_t st)
{
	dim3 grid((unsigned)((n_lanes + 255) / 256));
	if (mode == 0) hipLaunchKernelGGL(k_micro_gather, grid, dim3(256), 0, st, (const u64 *)buf, ncell, n_lanes, salt, sink);
	else if (mode == 1) hipLaunchKernelGGL(k_micro_atomic_or, grid, dim3(256), 0, st, buf, ncell, n_lanes, salt);
	else if (mode == 2) hipLaunchKernelGGL(k_micro_byte_store, grid, dim3(256), 0, st, (unsigned char *)buf, ncell * 8, n_lanes, salt);
	else if (mode == 3) hipLaunchKernelGGL(k_micro_byte_gather, grid, dim3(256), 0, st, (const unsigned char *)buf, ncell * 8, n_lanes, salt, sink);
	else if (mode == 4) hipLaunchKernelGGL(k_micro_store8, grid, dim3(256), 0, st, buf, ncell, n_lanes, salt);
	else if (mode == 5) hipLaunchKernelGGL(k_micro_atomic_or32, grid, dim3(256), 0, st, (u32 *)buf, ncell * 2, n_lanes, salt);
	else if (mode == 6) hipLaunchKernelGGL(k_micro_atomic_or_wg, grid, dim3(256), 0, st, buf, ncell, n_lanes, salt);
	else if (mode == 8) hipLaunchKernelGGL(k_micro_gather32, grid, dim3(256), 0, st, (const u32 *)buf, ncell * 2, n_lanes, salt, sink);
	else if (mode == 9) hipLaunchKernelGGL(k_micro_gather32v<1>, grid, dim3(256), 0, st, (const u32 *)buf, ncell * 2, n_lanes, salt, sink);
	else if (mode == 10) hipLaunchKernelGGL(k_micro_gather32v<2>, grid, dim3(256), 0, st, (const u32 *)buf, ncell * 2, n_lanes, salt, sink);
	else if (mode == 11) hipLaunchKernelGGL(k_micro_gather32v<3>, grid, dim3(256), 0, st, (const u32 *)buf, ncell * 2, n_lanes, salt, sink);
	else if (mode == 12) hipLaunchKernelGGL(k_micro_atomic_or32_sparse, grid, dim3(256), 0, st, (u32 *)buf, ncell * 2, n_lanes, salt);
	else hipLaunchKernelGGL(k_micro_atomic_or_ret, grid, dim3(256), 0, st, buf, ncell, n_lanes, salt, sink);
}

}   // namespace kmxk

#include "range_kernels.h"
