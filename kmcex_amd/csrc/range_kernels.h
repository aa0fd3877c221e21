// range_kernels.h -- the north star's multi-GPU partition on the device: every coupled array cut by POSITION RANGE across
// the ranks.  Included at the end of kernels.hip (one translation unit: c_seeds, the claim tuples, k_round_detect
// and k_reorder are shared with the single-GPU path).
//
// Rank q owns the cells [cell_lo[q], cell_lo[q+1]) -- 16 positions each -- of EVERY array.  List i of a block lives on rank
// i % P for the whole block: it hashes its k-mers, keeps the list order and reorders locally (kmodel.hpp:529-540); k-mers
// never move.  Between a list rank s and an owner q lies ONE REGION of 64-bit words (RangeDev::out[q]): the winners' commits of
// round t in front, the triples of round t + 1 behind them, and a two-word header {commits, triples} (k_range_seal) -- the
// counts travel IN BAND, no kernel of a round needs a number from the host.  Where the region lives is the transport's
// business: in the owner's memory, written through a peer mapping (kmx_build_from_kmc_multi_ex: one process, the rounds ordered
// by events), or in the sender's, moved by the caller's all-to-all (kmx_range_*_dev, kmcex_amd/dist.py).  A round (all lists at
// once, list i against array (i + t) % nb, kmodel.hpp:560-565):
//   1. k_range_emit        list rank: one TRIPLE (position, wanted value, list, slot, hash index) per position of every attempt,
//      k_range_seal        by owner rank and, inside a workgroup's run, sorted by claim bin; header out        -> owner
//   2. k_range_commit_apply owner: the commit words of every region first (the set loop kmodel.hpp:611-618 of the round before:
//      k_range_verdict     the order the sequential algorithm has), then the cell of every triple -- conflict with a set tag
//      k_round_detect      (kmodel.hpp:604-610) | untagged --; every untagged claim is filed in the position-hashed bins of the
//      <..., RANGE>        single-GPU path (one atomic per wave and bin) and ITS detect kernel finds the positions wanted with both
//      k_range_ship        values by claims of this round: the owner sees every claim on its positions, so contention is found
//                          where the bits live.  One verdict byte per triple, shipped to where the sender reads it  -> list rank
//   3. k_range_apply       list rank: a slot with a conflict anywhere has failed (final: bits are never cleared); a candidate
//      k_range_resolve     none of whose untagged positions is wanted with both values wins outright and its commit words go
//                          straight to the front of the regions; the others are decided in list order from the verdicts alone, on
//                          their both-wanted positions (the only writers that can matter to them are earlier contended winners):
//                          priority reservations on an exact position table, the smallest undecided slot always wins its turn.
// Then k_reorder and, after the last round, k_rest_append and the block's km_back emission -- the single-GPU kernels.
#pragma once

// triple: position (36 bits) | want << 36 | list << 37 | slot << 41 | hash index << 59
#define RT_POS(tr) ((tr) & ((1ULL << 36) - 1))
#define RT_WANT(tr) ((u32)((tr) >> 36) & 1u)
#define RT_LIST(tr) ((u32)((tr) >> 37) & 15u)
#define RT_X(tr) ((u32)((tr) >> 41) & (KMX_BUCKET - 1))
#define RT_J(tr) ((u32)((tr) >> 59) & 15u)
#define RT_MAKE(pos, want, i, x, j) ((u64)(pos) | ((u64)(want) << 36) | ((u64)(i) << 37) | ((u64)(x) << 41) | ((u64)(j) << 59))
#define RT_COMMIT (1ULL << 63)      // a commit word: position | value << 36 | ARRAY << 37
enum { RV_CONFLICT = 1, RV_UNTAGGED = 2, RV_BOTH = 4 };

__device__ __forceinline__ int range_owner(const RangePlan &pl, u64 cell)
{
	int q = 0;
#pragma unroll 1
	while (q + 1 < pl.world && cell >= pl.cell_lo[q + 1]) q++;
	return q;
}

// What an owner received: one region per sender; the counts from the headers (device memory) or, when the caller moved the
// words and knows them, by value.  Staged in LDS (lanes index the regions independently: not from the kernel arguments).
struct RangeInLds {
	u32 pc[KMX_MAX_RANKS + 1], pt[KMX_MAX_RANKS + 1];   // exclusive prefix sums of the commits taken / the triples over the regions
	u32 c0[KMX_MAX_RANKS];                              // first commit word taken of region s
	u32 nc[KMX_MAX_RANKS];                              // commits in front of region s (the triples start there)
	const u64 *reg[KMX_MAX_RANKS];
	unsigned char *vout[KMX_MAX_RANKS];
};
// part: which commit words the caller is after (kmx_types.h RANGE_ALL / RANGE_BULK / RANGE_LATE)
__device__ __forceinline__ void range_in_stage(const RangeIn &in, RangeInLds &L, int part = RANGE_ALL)
{
	if (threadIdx.x == 0) {
		u32 c = 0, t = 0;
		for (int s = 0; s < in.world; s++) {
			u32 nc = in.hdr ? in.hdr[in.hdr_stride * s] : in.nc[s], nt = in.hdr ? in.hdr[in.hdr_stride * s + 1] : in.nt[s];
			if (in.cap) { if (nc > in.cap) nc = in.cap; if (nt > in.cap - nc) nt = in.cap - nc; }     // (the sender dropped what did not fit and says so: the build is repeated)
			// (RANGE_BULK runs while the sender is still writing: only word 2 of the header is final then)
			const u32 nbulk = in.hdr ? in.hdr[in.hdr_stride * s + 2] : 0;
			const u32 lo = part == RANGE_LATE ? (nbulk < nc ? nbulk : nc) : 0, hi = part == RANGE_BULK ? nbulk : nc;
			L.pc[s] = c; L.pt[s] = t; L.nc[s] = nc; L.c0[s] = lo;
			L.reg[s] = in.reg[s]; L.vout[s] = in.vout[s];
			c += hi - lo; t += nt;
		}
		L.pc[in.world] = c; L.pt[in.world] = t;
	}
	__syncthreads();
}
__device__ __forceinline__ int range_seg_of(const u32 *pre, int world, u32 q)
{
	int s = 0;
#pragma unroll 1
	while (s + 1 < world && q >= pre[s + 1]) s++;
	return s;
}
// the list side's view of its regions, staged the same way
struct RangeOutLds { u64 *out[KMX_MAX_RANKS]; };
__device__ __forceinline__ void range_out_stage(const RangeDev &rd, int world, RangeOutLds &L)
{
	if ((int)threadIdx.x < world) L.out[threadIdx.x] = rd.out[threadIdx.x];
	__syncthreads();
}

// Appends this thread's `cnt` (<= NHM) COMMIT words, each to the front part of the region of its destination rank: the workgroup
// counts per destination in LDS and reserves each run with ONE global atomic.
template <int NHM> __device__ __forceinline__ void range_block_append(const RangeDev &rd, const RangeOutLds &L, int world, const u64 *word, const int *dest, u32 valid, int *s_cnt, int *s_base)
{
	if ((int)threadIdx.x < world) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	int rank[NHM];
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if ((valid >> j) & 1u) rank[j] = atomicAdd(&s_cnt[dest[j]], 1);
	__syncthreads();
	if ((int)threadIdx.x < world) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(rd.ccnt + (int)threadIdx.x * KMX_CTR_STRIDE, s_cnt[threadIdx.x]) : 0;
	__syncthreads();
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if ((valid >> j) & 1u) {
			const u64 off = (u64)(s_base[dest[j]] + rank[j]);
			if (off < rd.cap) L.out[dest[j]][off] = word[j];
			else *rd.ovf = 1;
		}
	__syncthreads();
}

// The same for a round's triples, behind the commits, SORTED by (destination, claim bin of the position) inside the workgroup's
// run: the owner files untagged claims in the position-hashed bins of its list, and with the triples of one bin next to each other
// a wave there reserves a run of tuples with ONE atomic instead of one per lane (k_range_verdict).  s_key: KMX_MAX_RANKS * 256 + 1
// counters.  where[j]: destination << 28 | index in its region (where the verdict byte will be).
template <int NHM> __device__ __forceinline__ void range_block_append_sorted(const RangeDev &rd, const RangeOutLds &L, int world, const u64 *word, const int *dest, u32 valid, u32 *where, int *s_key, int *s_base, int *s_wsum)
{
	const int K = world * KMX_CL_MAXBINS;
	for (int q = threadIdx.x; q <= K; q += 256) s_key[q] = 0;
	__syncthreads();
	int rank[NHM], key[NHM];
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if ((valid >> j) & 1u) {
			key[j] = dest[j] * KMX_CL_MAXBINS + (int)cl_bin(cl_mix(RT_POS(word[j])));
			rank[j] = atomicAdd(&s_key[key[j]], 1);
		}
	__syncthreads();
	// exclusive scan of the K counters in place: thread t owns the `world` consecutive keys [t * world, (t + 1) * world)
	int loc[KMX_MAX_RANKS], sum = 0;
#pragma unroll
	for (int e = 0; e < KMX_MAX_RANKS; e++)
		if (e < world) { loc[e] = sum; sum += s_key[(int)threadIdx.x * world + e]; }
	int incl = sum;                                                       // inclusive scan over the 256 threads: wave scan + wave totals
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
	if (lane == 63) s_wsum[wv] = incl;
	__syncthreads();
	int before = incl - sum;
	for (int w2 = 0; w2 < wv; w2++) before += s_wsum[w2];
#pragma unroll
	for (int e = 0; e < KMX_MAX_RANKS; e++)
		if (e < world) s_key[(int)threadIdx.x * world + e] = before + loc[e];
	if (threadIdx.x == 255) s_key[K] = before + sum;                      // the grand total closes the last destination
	__syncthreads();
	if ((int)threadIdx.x < world) {
		const int tot = s_key[((int)threadIdx.x + 1) * KMX_CL_MAXBINS] - s_key[(int)threadIdx.x * KMX_CL_MAXBINS];
		// (the commits of the round before are all in place: their counter is only read here)
		s_base[threadIdx.x] = rd.ccnt[(int)threadIdx.x * KMX_CTR_STRIDE] + (tot ? atomicAdd(rd.tcnt + (int)threadIdx.x * KMX_CTR_STRIDE, tot) : 0);
	}
	__syncthreads();
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if ((valid >> j) & 1u) {
			const u64 off = (u64)(s_base[dest[j]] + (s_key[key[j]] - s_key[dest[j] * KMX_CL_MAXBINS]) + rank[j]);
			if (off < rd.cap) L.out[dest[j]][off] = word[j];
			else *rd.ovf = 1;
			where[j] = ((u32)dest[j] << 28) | (u32)(off < rd.cap ? off : 0);
		}
	__syncthreads();
}

// the header of every region goes to its owner: {commits in front, triples behind them, bulk commits}; the counters start over
// (the next words this rank writes into a region are the commits of the round being decided, from its front)
__device__ __forceinline__ void range_seal_regions(const RangeDev &rd, int world)
{
	const int q = threadIdx.x;
	if (q >= world) return;
	const u32 nc = (u32)rd.ccnt[q * KMX_CTR_STRIDE], nt = (u32)rd.tcnt[q * KMX_CTR_STRIDE];
	if ((u64)nc + nt > rd.cap) *rd.ovf = 1;
	rd.hdr_out[q][0] = nc;
	rd.hdr_out[q][1] = nt;
	rd.ccnt[q * KMX_CTR_STRIDE] = 0;
	rd.tcnt[q * KMX_CTR_STRIDE] = 0;
}

// 1. every attempt of the lists this rank holds -> triples by owner rank; positions kept per slot (crec) for steps 3
template <int W, int NHM> __global__ __launch_bounds__(256) void k_range_emit(ModelDev md, BlockDev bd, RangeDev rd, RangePlan pl, int t, int pp)
{
	__shared__ int s_key[KMX_MAX_RANKS * KMX_CL_MAXBINS + 1], s_base[KMX_MAX_RANKS], s_wsum[4];
	__shared__ RangeOutLds L;
	const int i = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
	const int n = bd.n[pp][i];
	if ((int)blockIdx.x * 256 < n) {                                    // uniform
		range_out_stage(rd, pl.world, L);
		const u64 row = (u64)i * KMX_BUCKET;
		const int a = (i + t) % md.nb;
		if (x == 0) atomicAdd(bd.stats + ST_ATTEMPTS, (u64)n);
		u64 word[NHM];
		int dest[NHM];
		u32 where[NHM], valid = 0;
		if (x < n) {
			const u32 raw = bd.list[pp][row + x];
			const u32 idx = (raw & LIST_HOLE) ? bd.mover[pp][row + (raw & ~LIST_HOLE)] : raw;
			if (raw & LIST_HOLE) bd.list[pp][row + x] = idx;            // later kernels of the round read plain entries
			u64 v[W];
			load_kmer<W>(bd.kmers, row + idx, v);
			const u32 bin = md.bin_of_occ[bd.counts[row + idx]];
			Premixed<W> pm = premix_string<W>(left_align<W>(v, md.k), md.gfull);
			u64 pos[NHM];
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh) {
					pos[j] = mod_u64(murmur_seeded<W>(pm, md.gfull, c_seeds[(a * md.nh + j) & 127]), md.km_mod);
					word[j] = RT_MAKE(pos[j], (bin >> j) & 1u, i, x, j);
					dest[j] = range_owner(pl, pos[j] >> 4);
					valid |= 1u << j;
				}
			crec_store<NHM>(bd.crec[pp] + (row + x) * (u64)crec_words(md.nh), md.nh, pos);
			bd.uw[pp][row + x] = bin << 16;                              // the untagged mask arrives with the verdicts
			bd.status[pp][row + x] = SLOT_UNDECIDED;
		}
		range_block_append_sorted<NHM>(rd, L, pl.world, word, dest, valid, where, s_key, s_base, s_wsum);
		if (x < n)
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh) rd.tidx[(row + x) * (u64)md.nh + j] = where[j];
	}
}

// 1b. the headers, when the launch that filled the regions has ended (a launch of one wave: a ticket per workgroup in the
// producer -- same-address atomics serialise at ~11 ns -- costs more than it saves)
__global__ __launch_bounds__(64) void k_range_seal(RangeDev rd, int world) { range_seal_regions(rd, world); }

// 2a. owner: the winners' tag / value bits of the round before (kmodel.hpp:611-618), the commit words in front of every region
// (ovf: the owner's overflow flags of the round before -- k_range_ship has read them -- are reset here, ahead of this round's k_range_verdict)
__global__ __launch_bounds__(256) void k_range_commit_apply(ModelDev md, RangeIn in, int *ovf, int part)
{
	__shared__ RangeInLds L;
	if (ovf && blockIdx.x == 0 && (int)threadIdx.x < md.nb) ovf[threadIdx.x] = 0;
	range_in_stage(in, L, part);
	const u32 n = L.pc[in.world];
	for (u32 c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
		const int s = range_seg_of(L.pc, in.world, c);
		const u64 tr = L.reg[s][L.c0[s] + (c - L.pc[s])];
		const u64 pos = RT_POS(tr);
		const u32 b = bit_in_cell(pos);
		atomicOr(md.cells[RT_LIST(tr)] + (pos >> 4), CELL_TAG(b) | (RT_WANT(tr) ? CELL_VAL(b) : 0u));
	}
}

// 2b. owner: the state of every claimed position; untagged claims go to the detect bins of their list
// (The bin counters of this kernel are PADDED, one per 128-byte line (pcnt[(list * bins + bin) * KMX_CTR_STRIDE]): every lane's
// returning atomic goes to a counter of its own choosing, and with the 1280 counters of the claim bins packed into 80 lines the
// lines serialised them -- 761 us per launch against 150 for the gathers; k_round_detect<..., RANGE> reads and resets the padded
// form.  A claim tuple names the triple it came from -- its index q among ALL the triples received this round, regions in rank
// order --, so that detect answers in that triple's verdict byte: lver[q].)
// (The triples of a sender's workgroup arrive sorted by claim bin -- range_block_append_sorted --, so neighbouring lanes mostly file
// into the same bin: the lanes of a wave that follow each other with the same (list, bin) reserve their tuples with ONE atomic.)
template <int NHM> __global__ __launch_bounds__(256) void k_range_verdict(ModelDev md, BlockDev obd, int *pcnt, int t, RangeIn in, unsigned char *lver)
{
	constexpr int NBIN = KMX_CL_BINS(NHM), CAP = KMX_CL_CAP_OF(NHM);
	__shared__ RangeInLds L;
	range_in_stage(in, L);
	u32 n = L.pt[in.world];
	if (n >> KMX_RANGE_QBITS) n = 0;                                    // (cannot happen: at most nb * 2^18 * nh <= 2^26 triples exist in a round)
	const int lane = threadIdx.x & 63;
	for (u32 base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {   // (uniform trip count: the wave votes below)
		const u32 q = base + threadIdx.x;
		const bool act = q < n;
		u64 tr = 0;
		if (act) {
			const int s = range_seg_of(L.pt, in.world, q);
			tr = L.reg[s][L.nc[s] + (q - L.pt[s])];
		}
		const u64 pos = RT_POS(tr);
		const u32 i = RT_LIST(tr), want = RT_WANT(tr);
		unsigned char v = 0;
		u64 mx = 0;
		u32 key = 0x80000000u | (u32)lane;                              // lanes without a claim: a key of their own
		bool untagged = false;
		if (act) {
			const int a = (int)((i + (u32)t) % (u32)md.nb);
			const cell_t c = md.cells[a][pos >> 4];
			const u32 b = bit_in_cell(pos);
			const u32 tag = (c >> (16 + b)) & 1u, val = (c >> b) & 1u;
			mx = cl_mix(pos);
			key = i * KMX_CL_MAXBINS + cl_bin(mx);
			if (tag) v = val != want ? RV_CONFLICT : 0;
			else { v = RV_UNTAGGED; untagged = true; }
		}
		// runs of equal keys among the lanes: heads, this lane's run [start, end)
		const u32 prev = __shfl_up(key, 1, 64);
		const u64 heads = __ballot(lane == 0 || key != prev), um = __ballot(untagged);
		const u64 upto = lane == 63 ? ~0ULL : ((2ULL << lane) - 1);     // lanes 0..lane
		const int start = 63 - __clzll((long long)(heads & upto));
		const u64 after = heads & ~upto;
		const int end = after ? __ffsll((long long)after) - 1 : 64;
		const u64 run = (end == 64 ? ~0ULL : ((1ULL << end) - 1)) & ~((1ULL << start) - 1);
		const int cnt = (int)__popcll(um & run), rk = (int)__popcll(um & run & ((1ULL << lane) - 1));
		int g0 = 0;
		if (lane == start && cnt) g0 = atomicAdd(pcnt + (int)key * KMX_CTR_STRIDE, cnt);
		g0 = __shfl(g0, start, 64);
		if (untagged) {
			const int g = g0 + rk;
			if (g < CAP) obd.cl_tup[0][((u64)i * NBIN + (key & (KMX_CL_MAXBINS - 1))) * CAP + g] = CL_RANGE_TUPLE(mx, want, q);
			else obd.cl_ovf[i] = 1;                                     // tuples lost: every untagged claim of the list counts as contended
		}
		if (act) lver[q] = v;
	}
}
// 2c. k_round_detect<..., RANGE> on the owner's bins has marked the claims that sit on positions wanted with both values (lver).
// The verdict bytes go to where the senders read them: byte `off` of region s for the word `off` of that region.  (A bin that lost
// tuples or does not fit its table: every untagged claim of that list counts as contended.)
__global__ __launch_bounds__(256) void k_range_ship(BlockDev obd, int nb, RangeIn in, const unsigned char *lver)
{
	__shared__ RangeInLds L;
	range_in_stage(in, L);
	u32 n = L.pt[in.world];
	if (n >> KMX_RANGE_QBITS) n = 0;
	bool any = false;
	for (int i = 0; i < nb; i++) any |= obd.cl_ovf[i] != 0;             // uniform
	for (u32 q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
		const int s = range_seg_of(L.pt, in.world, q);
		const u32 off = L.nc[s] + (q - L.pt[s]);
		unsigned char v = lver[q];
		if (any && (v & RV_UNTAGGED) && obd.cl_ovf[RT_LIST(L.reg[s][off])]) v |= RV_BOTH;
		L.vout[s][off] = v;
	}
}

// the BULK of a round's commits -- the uncontended winners', 99.6 % -- is complete when k_range_apply ends: its count goes to the
// owners right away (header word 2), so that they can set those bits while this rank still orders its contended candidates
__global__ __launch_bounds__(64) void k_range_seal_bulk(RangeDev rd, int world)
{
	const int q = threadIdx.x;
	if (q < world) rd.hdr_out[q][2] = (u32)rd.ccnt[q * KMX_CTR_STRIDE];
}

// 3a. list rank: the verdicts of a slot's positions, read from where its triples went; an uncontended winner's commit words --
// one per position it saw untagged (kmodel.hpp:611-618) -- go straight to the front of the regions, by owner rank
template <int NHM> __global__ __launch_bounds__(256) void k_range_apply(ModelDev md, BlockDev bd, RangeDev rd, RangePlan pl, int t, int pp)
{
	__shared__ int s_fail, s_cnt, s_base, s_acnt[KMX_MAX_RANKS], s_abase[KMX_MAX_RANKS];
	__shared__ RangeOutLds L;
	__shared__ const unsigned char *s_vin[KMX_MAX_RANKS];
	const int i = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
	const int n = bd.n[pp][i];
	if ((int)blockIdx.x * 256 >= n) return;                             // uniform
	if (threadIdx.x == 0) { s_fail = 0; s_cnt = 0; }
	if ((int)threadIdx.x < pl.world) s_vin[threadIdx.x] = rd.vin[threadIdx.x];
	range_out_stage(rd, pl.world, L);
	const u64 row = (u64)i * KMX_BUCKET;
	bool failed = false, contended = false;
	u32 both_of_slot = 0, valid = 0;
	u64 word[NHM];
	int dest[NHM];
	if (x < n) {
		u32 um = 0, both = 0;
#pragma unroll
		for (int j = 0; j < NHM; j++)
			if (j < md.nh) {
				const u32 w = rd.tidx[(row + x) * (u64)md.nh + j];
				dest[j] = (int)(w >> 28);
				const unsigned char v = s_vin[w >> 28][w & 0x0FFFFFFFu];
				failed |= (v & RV_CONFLICT) != 0;
				um |= (v & RV_UNTAGGED) ? 1u << j : 0u;
				both |= (v & RV_BOTH) ? 1u << j : 0u;
			}
		contended = !failed && both != 0;
		both_of_slot = both;
		const u32 want = bd.uw[pp][row + x] >> 16;
		bd.uw[pp][row + x] = (want << 16) | um;
		if (failed) bd.status[pp][row + x] = SLOT_FAILED;
		else if (!contended && um) {                                    // nobody wants the other value on any of its positions: it has won
			const CRec<NHM> cr = crec_load<NHM>(bd.crec[pp] + (row + x) * (u64)crec_words(md.nh), md.nh);
			u64 pos[NHM];
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh) pos[j] = ((u64)crec_cell<NHM>(cr, j) << 4) | crec_nib<NHM>(cr, md.nh, j);
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((um >> j) & 1u)) {
					word[j] = RT_COMMIT | RT_MAKE(pos[j], value_at_position<NHM>(md, pos, um, want, j), (i + t) % md.nb, 0, 0);
					valid |= 1u << j;
				}
		}
	}
	const u64 fm = __ballot(failed);
	if ((threadIdx.x & 63) == 0 && fm) atomicAdd(&s_fail, (int)__popcll(fm));
	const int slot = block_append_slot(rd.n_contended + i * KMX_CTR_STRIDE, contended, &s_cnt, &s_base);
	if (contended) { rd.contended[row + slot] = (u32)x; rd.rt_um[row + slot] = both_of_slot; }
	if (threadIdx.x == 0 && s_fail) atomicAdd(bd.tile_cnt[pp] + i * KMX_NTILES + (int)(blockIdx.x >> 2), s_fail);   // survivors per 1024-slot tile (k_reorder)
	range_block_append<NHM>(rd, L, pl.world, word, dest, valid, s_acnt, s_abase);
}

// the contended candidates that won their turn in k_range_resolve: their commit words, AFTER the ordered loop (inside it every
// returning atomic would add its latency to each of the loop's dependent iterations).  Every thread of the workgroup calls it;
// `won` says whether this thread's record x has won.  The workgroup counts per destination in LDS and reserves with one atomic.
template <int NHM> __device__ __forceinline__ void range_winner_commits(const ModelDev &md, const BlockDev &bd, const RangeDev &rd, const RangeOutLds &L, const RangePlan &pl, int t, int pp, int i, u64 row, u32 x, bool won,
                                                                        int *s_cc, int *s_cb)
{
	if ((int)threadIdx.x < pl.world) s_cc[threadIdx.x] = 0;
	__syncthreads();
	u64 word[NHM];
	int dest[NHM], rank[NHM];
	u32 valid = 0;
	if (won) {
		const u32 uw = bd.uw[pp][row + x], um = uw & 0xFFFFu, want = uw >> 16;
		const CRec<NHM> cr = crec_load<NHM>(bd.crec[pp] + (row + x) * (u64)crec_words(md.nh), md.nh);
		u64 pos[NHM];
#pragma unroll
		for (int j = 0; j < NHM; j++)
			if (j < md.nh) pos[j] = ((u64)crec_cell<NHM>(cr, j) << 4) | crec_nib<NHM>(cr, md.nh, j);
#pragma unroll
		for (int j = 0; j < NHM; j++)
			if (j < md.nh && ((um >> j) & 1u)) {
				dest[j] = range_owner(pl, pos[j] >> 4);
				word[j] = RT_COMMIT | RT_MAKE(pos[j], value_at_position<NHM>(md, pos, um, want, j), (i + t) % md.nb, 0, 0);
				rank[j] = atomicAdd(&s_cc[dest[j]], 1);
				valid |= 1u << j;
			}
	}
	__syncthreads();
	if ((int)threadIdx.x < pl.world) s_cb[threadIdx.x] = s_cc[threadIdx.x] ? atomicAdd(rd.ccnt + (int)threadIdx.x * KMX_CTR_STRIDE, s_cc[threadIdx.x]) : 0;
	__syncthreads();
#pragma unroll
	for (int j = 0; j < NHM; j++)
		if ((valid >> j) & 1u) {
			const u64 off = (u64)(s_cb[dest[j]] + rank[j]);
			if (off < rd.cap) L.out[dest[j]][off] = word[j];
			else *rd.ovf = 1;
		}
	__syncthreads();
}

// 3b. the contended candidates of a list, in list order, from the verdicts alone.  ONE workgroup per list.  Only the
// positions the owners reported as wanted with BOTH values take part (rt_um: bit j of a record): on any other untagged
// position every claim of the round wants the same value, so whoever wins it leaves what the others want there -- it can
// neither fail a record nor be lost by one.  An exact position table (open addressing on the position itself): a record
// reserves every both-wanted position it still believes untagged with its priority (smaller slot first); whoever holds all
// of its positions has no earlier undecided record on any of them -- it wins and MARKS them with the value it commits (a
// k-mer that hits a position twice leaves the OR, kmodel.hpp:611-618); a record that meets a mark with the other value has
// failed, with its own value the position leaves its mask.  The smallest undecided slot wins every iteration, and a later
// record can never win a position an earlier undecided one still wants, so the outcome is the sequential one (the scheme
// of finish_lds).  Up to 2048 records on 2048 positions are decided in LDS with their state in registers (the usual case:
// a few hundred records per list and round); larger sets use the table in global memory, without a size limit.
__device__ __forceinline__ u32 rt_hash(u64 pos) { return (u32)((pos * 0x9E3779B97F4A7C15ULL) >> 32); }
constexpr u32 RT_WON = 1u << 30;                                     // in a record's mask word: it won its turn
constexpr int RT_LDS_BITS = 12, RT_LDS = 1 << RT_LDS_BITS, RT_RPT = 2;        // 4096 entries: key | mark in one u64 (32 KB) + reservations (16 KB)
#define RT_LDS_MARK(v) (1ULL << (62 + (v)))
template <int NHM> __device__ __forceinline__ u64 range_resolve_lds(const ModelDev &md, const BlockDev &bd, int pp, int i, u64 row, int nc, const u32 *cont, const u32 *bothm, u32 *wonm,
                                                                    u64 *s_key, u32 *s_resv, int *s_pending, int *s_succ)
{
	constexpr u32 TM = RT_LDS - 1;
	for (int q = threadIdx.x; q < RT_LDS; q += 1024) { s_key[q] = 0; s_resv[q] = 0; }
	__syncthreads();
	u32 x[RT_RPT], c[RT_RPT], want[RT_RPT];
	bool live[RT_RPT];
	unsigned short e[RT_RPT][NHM];
#pragma unroll
	for (int k = 0; k < RT_RPT; k++) {
		const int r = (int)threadIdx.x + k * 1024;
		live[k] = r < nc;
		x[k] = c[k] = want[k] = 0;
#pragma unroll
		for (int j = 0; j < NHM; j++) e[k][j] = 0;
		if (!live[k]) continue;
		x[k] = cont[r];
		c[k] = bothm[r] & 0xFFFFu;
		want[k] = bd.uw[pp][row + x[k]] >> 16;
		const CRec<NHM> cr = crec_load<NHM>(bd.crec[pp] + (row + x[k]) * (u64)crec_words(md.nh), md.nh);
#pragma unroll
		for (int j = 0; j < NHM; j++)
			if (j < md.nh && ((c[k] >> j) & 1u)) {
				const u64 pos = ((u64)crec_cell<NHM>(cr, j) << 4) | crec_nib<NHM>(cr, md.nh, j);
				u32 q = rt_hash(pos) & TM;
				for (;;) {                                               // (at most 2048 positions in 4096 entries: a free one always exists)
					const u64 old = atomicCAS(&s_key[q], 0ULL, pos + 1);
					if (old == 0 || old == pos + 1) break;
					q = (q + 1) & TM;
				}
				e[k][j] = (unsigned short)q;
			}
	}
	__syncthreads();
	u64 iters = 0;
	for (;; iters++) {
		if (threadIdx.x == 0) *s_pending = 0;
#pragma unroll
		for (int k = 0; k < RT_RPT; k++)                                 // 1: reserve
			if (live[k]) {
				const u32 prio = 0x40000u - x[k];
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c[k] >> j) & 1u)) atomicMax(&s_resv[e[k][j]], prio);
			}
		__syncthreads();
		int succ = 0;
#pragma unroll
		for (int k = 0; k < RT_RPT; k++)                                 // 2: whoever holds everything wins and marks
			if (live[k]) {
				const u32 prio = 0x40000u - x[k];
				bool mine = true;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c[k] >> j) & 1u)) mine &= s_resv[e[k][j]] == prio;
				if (!mine) continue;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c[k] >> j) & 1u)) {
						u32 v = (want[k] >> j) & 1u;
#pragma unroll
						for (int j2 = 0; j2 < NHM; j2++)
							if (j2 < md.nh && j2 != j && ((c[k] >> j2) & 1u) && e[k][j2] == e[k][j]) v |= (want[k] >> j2) & 1u;
						atomicOr(&s_key[e[k][j]], RT_LDS_MARK(v));
					}
				bd.status[pp][row + x[k]] = SLOT_INSERTED;
				wonm[(int)threadIdx.x + k * 1024] = RT_WON;                  // (read again by this very thread when the loop is over: k_range_resolve's tail)
				live[k] = false;
				succ++;
			}
		if (succ) atomicAdd(s_succ, succ);
		__syncthreads();
#pragma unroll
		for (int k = 0; k < RT_RPT; k++)                                 // 3: what the winners marked
			if (live[k]) {
				bool conflict = false;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c[k] >> j) & 1u)) {
						const u32 mk = (u32)(s_key[e[k][j]] >> 62);
						if (mk) { conflict |= ((mk >> 1) & 1u) != ((want[k] >> j) & 1u); c[k] &= ~(1u << j); }
						else s_resv[e[k][j]] = 0;                        // everybody who still wants it reserves it again
					}
				if (conflict) { mark_failed(bd, pp, i, row, x[k]); live[k] = false; }
				else *s_pending = 1;
			}
		__syncthreads();
		const int pending = *s_pending;
		__syncthreads();
		if (!pending) break;
	}
	return iters;
}
template <int NHM> __global__ __launch_bounds__(1024) void k_range_resolve(ModelDev md, BlockDev bd, RangeDev rd, RangePlan pl, int t, int pp)
{
	__shared__ int s_pending, s_succ, s_ent, s_cc[KMX_MAX_RANKS], s_cb[KMX_MAX_RANKS];
	__shared__ u64 s_key[RT_LDS];
	__shared__ u32 s_resv[RT_LDS];
	const int i = blockIdx.x, tab = i / pl.world;                       // the lists a rank holds are i = rank, rank + world, ...: one table each
	const int nc = rd.n_contended[i * KMX_CTR_STRIDE];
	if (nc == 0) return;
	__shared__ RangeOutLds Lo;
	range_out_stage(rd, pl.world, Lo);
	const u64 row = (u64)i * KMX_BUCKET;
	const u32 *cont = rd.contended + row;
	u32 *cur = rd.rt_um + row;                                          // per record: mask of both-wanted positions still believed untagged (| live)
	if (threadIdx.x == 0) { s_succ = 0; s_ent = 0; atomicAdd(bd.stats + ST_CONTENDED, (u64)nc); atomicMax(bd.stats + ST_MAX_U0, (u64)nc); }
	__syncthreads();
	{
		int ent = 0;
		for (int r = threadIdx.x; r < nc; r += 1024) ent += __popc(cur[r] & 0xFFFFu);
		if (ent) atomicAdd(&s_ent, ent);
	}
	__syncthreads();
	u64 iters = 0;
	if (nc <= 1024 * RT_RPT && s_ent * 2 <= RT_LDS) iters = range_resolve_lds<NHM>(md, bd, pp, i, row, nc, cont, cur, cur, s_key, s_resv, &s_pending, &s_succ);
	else {
		int tb = 10;
		while ((1u << tb) < 4u * (u32)nc * (u32)md.nh && tb < (int)rd.rt_bits) tb++;
		const u32 tmask = (1u << tb) - 1;
		u64 *key = rd.rt_key + ((u64)tab << rd.rt_bits);
		u32 *resv = rd.rt_resv + ((u64)tab << rd.rt_bits), *mark = rd.rt_mark + ((u64)tab << rd.rt_bits);
		u32 *eidx = rd.rt_eidx + row * (u64)md.nh;                      // per record: table entry of each position
		constexpr u32 LIVE = 1u << 31, WON = RT_WON;
		for (u32 q = threadIdx.x; q <= tmask; q += 1024) { key[q] = 0; resv[q] = 0; mark[q] = 0; }
		drain_vmem();
		__syncthreads();
		for (int r = threadIdx.x; r < nc; r += 1024) {
			const u32 x = cont[r];
			const u32 c0 = cur[r] & 0xFFFFu;
			const CRec<NHM> cr = crec_load<NHM>(bd.crec[pp] + (row + x) * (u64)crec_words(md.nh), md.nh);
#pragma unroll
			for (int j = 0; j < NHM; j++)
				if (j < md.nh && ((c0 >> j) & 1u)) {
					const u64 pos = ((u64)crec_cell<NHM>(cr, j) << 4) | crec_nib<NHM>(cr, md.nh, j);
					u32 e = rt_hash(pos) & tmask;
					for (;;) {
						const u64 old = atomicCAS(&key[e], 0ULL, pos + 1);
						if (old == 0 || old == pos + 1) break;
						e = (e + 1) & tmask;
					}
					eidx[(u64)r * md.nh + j] = e;
				}
			cur[r] = c0 | LIVE;
		}
		drain_vmem();
		__syncthreads();
		for (;; iters++) {
			if (threadIdx.x == 0) s_pending = 0;
			for (int r = threadIdx.x; r < nc; r += 1024) {                 // 1: reserve
				const u32 c = cur[r];
				if (!(c & LIVE)) continue;
				const u32 prio = 0x40000u - cont[r];
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c >> j) & 1u)) atomicMax(&resv[eidx[(u64)r * md.nh + j]], prio);
			}
			drain_vmem();
			__syncthreads();
			int succ = 0;
			for (int r = threadIdx.x; r < nc; r += 1024) {                 // 2: whoever holds everything wins and marks
				const u32 c = cur[r];
				if (!(c & LIVE)) continue;
				const u32 x = cont[r], prio = 0x40000u - x, want = bd.uw[pp][row + x] >> 16;
				bool mine = true;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c >> j) & 1u)) mine &= cell_load_coherent(&resv[eidx[(u64)r * md.nh + j]]) == prio;
				if (!mine) continue;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c >> j) & 1u)) {
						const u32 e = eidx[(u64)r * md.nh + j];
						u32 v = (want >> j) & 1u;
#pragma unroll
						for (int j2 = 0; j2 < NHM; j2++)
							if (j2 < md.nh && j2 != j && ((c >> j2) & 1u) && eidx[(u64)r * md.nh + j2] == e) v |= (want >> j2) & 1u;
						atomicOr(&mark[e], 1u << v);
					}
				bd.status[pp][row + x] = SLOT_INSERTED;
				cur[r] = WON;                                            // (read again below by this very thread)
				succ++;
			}
			if (succ) atomicAdd(&s_succ, succ);
			drain_vmem();
			__syncthreads();
			for (int r = threadIdx.x; r < nc; r += 1024) {                 // 3: what the winners marked
				u32 c = cur[r];
				if (!(c & LIVE)) continue;
				const u32 x = cont[r], want = bd.uw[pp][row + x] >> 16;
				bool conflict = false;
#pragma unroll
				for (int j = 0; j < NHM; j++)
					if (j < md.nh && ((c >> j) & 1u)) {
						const u32 e = eidx[(u64)r * md.nh + j];
						const u32 mk = cell_load_coherent(&mark[e]);
						if (mk) { conflict |= ((mk >> 1) & 1u) != ((want >> j) & 1u); c &= ~(1u << j); }
						else resv[e] = 0;                                    // everybody who still wants it reserves it again
					}
				if (conflict) { mark_failed(bd, pp, i, row, x); cur[r] = 0; }
				else { cur[r] = c; s_pending = 1; }
			}
			drain_vmem();
			__syncthreads();
			const int pending = s_pending;
			__syncthreads();
			if (!pending) break;
		}
	}
	// the winners' commit words, now that the ordered loop is over (its state is dead: the registers are free for this)
	for (int r0 = 0; r0 < nc; r0 += 1024) {                              // uniform
		const int r = r0 + (int)threadIdx.x;
		const bool won = r < nc && (cur[r] & RT_WON);
		range_winner_commits<NHM>(md, bd, rd, Lo, pl, t, pp, i, row, r < nc ? cont[r] : 0u, won, s_cc, s_cb);
	}
	if (threadIdx.x == 0) {
		atomicAdd(bd.stats + ST_FIN_ITERS, iters + 1);
		if (s_succ) atomicAdd(bd.stats + ST_SLOW_SUCC, (u64)s_succ);
		rd.n_contended[i * KMX_CTR_STRIDE] = 0;                      // (every thread read it before the first barrier) ready for the next round's k_range_apply
	}
}

namespace kmxk {

// (the counts live on the device: every launch has a fixed grid that strides over what the headers say)
static inline unsigned range_grid(const ModelDev &md) { return (unsigned)std::min<u64>(((u64)md.nb * KMX_BUCKET * (u64)md.nh / 4 + 255) / 256, 4096); }

// step 1, list rank: the triples of round t behind the commits of the round before, then the headers.  fresh: a build starts
// (no commits are pending and the counters may hold what an aborted build left)
void range_emit(const ModelDev &md, const BlockDev &bd, const RangeDev &rd, const RangePlan &pl, int t, int pp, bool fresh, hipStream_t st)
{
	if (fresh) { hipMemsetAsync(rd.ccnt, 0, sizeof(int) * KMX_MAX_RANKS * KMX_CTR_STRIDE, st); hipMemsetAsync(rd.tcnt, 0, sizeof(int) * KMX_MAX_RANKS * KMX_CTR_STRIDE, st); }
	DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_range_emit<W, NHM>), dim3(KMX_BUCKET / 256, md.nb), dim3(256), 0, st, md, bd, rd, pl, t, pp));
	hipLaunchKernelGGL(k_range_seal, dim3(1), dim3(64), 0, st, rd, pl.world);
}
// the headers alone: the commits of the last round of a build
void range_seal(const RangeDev &rd, const RangePlan &pl, hipStream_t st) { hipLaunchKernelGGL(k_range_seal, dim3(1), dim3(64), 0, st, rd, pl.world); }
// step 2, owner.  obd: the owner's view (its own overflow flags and padded bin counters; the claim bins of the handle, unused by the
// list side here); lver: one byte per triple received
// commits: which commit words are still to be applied (RANGE_ALL, or RANGE_LATE when the bulk went ahead on another stream)
void range_verdict(const ModelDev &md, const BlockDev &obd_, int *pcnt, int t, const RangeIn &in, unsigned char *lver, int commits, bool small_late, hipStream_t st)
{
	BlockDev obd = obd_;
	obd.rverdict = lver;
	const unsigned g = range_grid(md);
	hipLaunchKernelGGL(k_range_commit_apply, dim3(commits == RANGE_LATE ? 64u : g), dim3(256), 0, st, md, in, obd.cl_ovf, commits);      // the previous round's winners first
	const int late = t >= 2 ? 1 : 0;                                    // (reports the fullest late bin: ST_MAX_LATE_BIN, like the single-GPU rounds)
	if (md.nh <= 8) {
		hipLaunchKernelGGL((k_range_verdict<8>), dim3(g), dim3(256), 0, st, md, obd, pcnt, t, in, lver);
		// late rounds: a few hundred claims per bin -- the small tables (16 KB, 256 threads: every bin resident at once); a bin that does
		// not fit sends its list down the ordered path (cl_ovf), exact like any other overflow
		if (late && small_late) hipLaunchKernelGGL((k_round_detect<8, 256, KMX_CL_TBITS_SMALL, true>), dim3(KMX_CL_BINS(8), md.nb), dim3(256), 0, st, obd, md.nb, 0, 0, 0, late);
		else hipLaunchKernelGGL((k_round_detect<8, 1024, KMX_CL_TBITS(8), true>), dim3(KMX_CL_BINS(8), md.nb), dim3(1024), 0, st, obd, md.nb, 0, 0, 0, late);
	} else {
		hipLaunchKernelGGL((k_range_verdict<16>), dim3(g), dim3(256), 0, st, md, obd, pcnt, t, in, lver);
		if (late && small_late) hipLaunchKernelGGL((k_round_detect<16, 256, KMX_CL_TBITS_SMALL, true>), dim3(KMX_CL_BINS(16), md.nb), dim3(256), 0, st, obd, md.nb, 0, 0, 0, late);
		else hipLaunchKernelGGL((k_round_detect<16, 1024, KMX_CL_TBITS(16), true>), dim3(KMX_CL_BINS(16), md.nb), dim3(1024), 0, st, obd, md.nb, 0, 0, 0, late);
	}
	hipLaunchKernelGGL(k_range_ship, dim3(std::min(g, 1024u)), dim3(256), 0, st, obd, md.nb, in, (const unsigned char *)lver);
}
// step 3, list rank: the verdicts of every slot; the uncontended winners' commits (the bulk) are in the regions when it ends ...
void range_apply(const ModelDev &md, const BlockDev &bd, const RangeDev &rd, const RangePlan &pl, int t, int pp, bool seal_bulk, hipStream_t st)
{
	const dim3 grid(KMX_BUCKET / 256, md.nb);
	if (md.nh <= 8) hipLaunchKernelGGL((k_range_apply<8>), grid, dim3(256), 0, st, md, bd, rd, pl, t, pp);
	else hipLaunchKernelGGL((k_range_apply<16>), grid, dim3(256), 0, st, md, bd, rd, pl, t, pp);
	if (seal_bulk) hipLaunchKernelGGL(k_range_seal_bulk, dim3(1), dim3(64), 0, st, rd, pl.world);
}
// ... then the contended in list order (their commits behind the bulk) and reorder_buffer
// (seal_all: every commit of the round is in the regions now: their count goes to the owners' headers -- as the "bulk" --, so that
// the owners can set the bits while this rank reorders and hashes the next round)
void range_resolve(const ModelDev &md, const BlockDev &bd, const RangeDev &rd, const RangePlan &pl, int t, int pp, bool seal_all, hipStream_t st)
{
	if (md.nh <= 8) hipLaunchKernelGGL((k_range_resolve<8>), dim3(md.nb), dim3(1024), 0, st, md, bd, rd, pl, t, pp);
	else hipLaunchKernelGGL((k_range_resolve<16>), dim3(md.nb), dim3(1024), 0, st, md, bd, rd, pl, t, pp);
	if (seal_all) hipLaunchKernelGGL(k_range_seal_bulk, dim3(1), dim3(64), 0, st, rd, pl.world);
	DISPATCH_W_NH(words(md), md.nh, hipLaunchKernelGGL((k_reorder<W, NHM>), dim3(KMX_NTILES + KMX_APPLY_WGS, md.nb), dim3(256), 0, st, md, bd, t, pp, 0));   // (no REC_WON records here: Un stays 0)
}
// the commit words of a last exchange
void range_commit_apply(const ModelDev &md, const RangeIn &in, int part, hipStream_t st)
{
	hipLaunchKernelGGL(k_range_commit_apply, dim3(range_grid(md)), dim3(256), 0, st, md, in, (int *)nullptr, part);
}

}   // namespace kmxk
