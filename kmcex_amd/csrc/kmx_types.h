// kmx_types.h -- plain structs passed by value from the host orchestration to the gfx950 kernels.
#pragma once
#include "device_common.h"

// Everything a kernel needs to address the model in HBM (A.1-A.3 of SURVEY.md).
struct ModelDev {
	int k, nh, nb, ci, cs, bf_num;
	int nh_first, nh_second;                   // the check gathers positions [0, nh_first) first, then [nh_first, nh_second), then the rest -- each group only if no earlier one conflicts
	StrGeom gfull, gback;                      // geometry of the k-mer and (k-2)-mer strings
	u32 *bf[3];      ModU64 bf_mod[3];         // Bloom filters, on-disk byte layout (kmodel.hpp:250,253)
	u32 *bf_back[3]; ModU64 bf_back_mod[3];    // their (k-2)-mer back filters       (kmodel.hpp:251,255)
	u32 *km_back;    ModU64 km_back_mod;       // back filter of the coupled arrays  (kmodel.hpp:267-269)
	cell_t *cells[KMX_MAX_NB];                 // coupled arrays, cell layout (device_common.h)
	ModU64 km_mod;                             // bit_array_length                   (kmodel.hpp:33,445)
	int bloom_direct;                          // 1: Bloom-class k-mers OR their bits in right away; 0: through the BitScatter of the Bloom slab
	u64 bf_woff[3], bf_back_woff[3];           // word offsets of the filters inside the slab (they are allocated back to back)
	int kmb_direct;                            // 1: a success ORs its (k-2)-mer into km_back right away (atomics); 0: the bits are
	                                           // scattered round by round through a BitScatter (k_kmback_emit / k_bs_apply)
	const u32 *bin_of_occ;                     // occ -> bin  (occu_bin.hpp:67-77)
	const u32 *mean_of_bin;                    // bin -> mean (occu_bin.hpp:79-83)
	// exact rest table (rest.hpp), device form: suffixes as integers instead of byte rows
	int rest_pre_len, rest_W;
	u64 rest_entries;
	const int *rest_h2i;                       // hash2index[4^pre_len]
	const int *rest_pre;                       // pre_buffer[groups+1]
	const u64 *rest_suf;                       // [entries][W] suffix value (low 2*(k-pre_len) bits)
	const int *rest_cnt;                       // count_bin[entries]
	// device-only accelerators of the lookup (same answers as the reference's binary search, far fewer touches)
	int rest_fbits;                            // F: rows are bucketed by the top F bits of the 2k-bit k-mer
	const u32 *rest_fine;                      // [2^F + 1] first row of every bucket
	const u64 *rest_q;                         // [4^pre_len][W] suffix of the first row of the NEXT group (the one row the
	                                           // reference's inclusive upper bound can still match), all-ones if none
};

// Partitioned bit-set into one order-free filter (set_bit is an OR, kmodel.hpp:576-581): producers append the bit
// addresses they want set to BS_BINS position-range bins (LDS-staged runs, bs_block_emit); k_bs_apply then sweeps the
// filter bin by bin, tile by tile: the bits of a tile are collected in LDS and the tile is OR-ed back with whole-word
// coalesced accesses -- streaming traffic instead of one random read-modify-write per bit.
#define BS_BINS 256
#define BS_TILE_WORDS (1u << 15)               // 2^15 words = 128 KB of LDS = 2^20 positions per tile
struct BitScatter {
	u32 *words;              // the filter (32-bit words; bit p of the filter = bit bit_in_word32(p) of word p >> 5)
	u64 nwords;
	u32 wshift;              // log2(positions per bin): bin = address >> wshift (a power of two of words, at most 2^32 positions)
	u32 cap;                 // tuples per bin; a producer that finds its bin full sets the bit with an atomic instead
	u32 *tup;                // [BS_BINS][cap] bit offsets inside the bin: (word - bin start) << 5 | bit in word
	int *cnt;                // [BS_BINS], reset by k_bs_apply
	u32 tlog2;               // log2(positions per tile), <= 20 (smaller only under the KMX_BS_TILE_LOG2 test hook)
	// second level, for bins of more than 8 tiles (filters above 256 MB): k_bs_split deals a bin's tuples to its tiles,
	// k_bs_apply2 sweeps one tile per workgroup -- every tuple and every word of the filter is then read once per sweep
	u32 cap2;                // tuples per (bin, tile); 0 = single level
	u32 *tup2;               // [BS_BINS << (wshift - tlog2)][cap2] bit offsets inside the tile
	int *cnt2;               // [BS_BINS << (wshift - tlog2)], reset by k_bs_apply2
};
// km_back tuples a finished block still owes (the whole-block form of k_kmback_emit): hosted by the finisher launches of
// the NEXT block's first two rounds -- one workgroup per list, the rest of the chip idle (k_slow_finish)
struct KmbackJob {
	const u64 *kmers;            // the finished block's k-mers (its staging region, identity order)
	const unsigned char *surv;   // its survivor flags (the other half of the double buffer)
	int n_in_block;
	int i0, n_lists;             // this launch emits lists [i0, i0 + n_lists) of that block; n_lists == 0: nothing hosted
};
#define BS_MAX_TILES_LOG2 8                    // tiles per bin at most, two-level (the split counts them in 256 LDS counters)

// Raw KMC records on the device (k_kmc_decode): fixed-size [suffix bytes big-endian | counter bytes little-endian];
// record r carries the prefix idx & prefix_mask of the LUT entry with lut[idx] <= r < lut[idx + 1] (kmc_file.cpp:439-478).
struct KmcDecode {
	const unsigned char *recs;   // records of this batch
	const u64 *lut;              // n_lut entries + sentinel
	u64 n_lut, prefix_mask;
	u32 rec_bytes, suf_bytes, cnt_bytes;
};

// device-side statistics (one u64 each)
enum { ST_ATTEMPTS = 0, ST_SUCCESSES, ST_SLOW_SUCC, ST_CONTENDED, ST_FIN_ITERS, ST_BAD_COUNT, ST_MAX_U0, ST_MAX_UFIN,
       ST_PIPE_ATTEMPTS, ST_PIPE_SUCC,        // attempts examined / winners committed inside fused commit|check launches (accounting only)
       ST_PIPE_GATHERS, ST_PIPE_ATOMICS,      // random 4-byte loads / 32-bit atomic ORs issued inside those launches (the staged fetch stops early; one atomic per NEW tag bit)
       ST_MAX_LATE_BIN,                       // fullest claim bin of a late round (t >= 2) since the last block: picks the form of their k_round_detect
       ST_DELTA_FAILS,                        // candidates of a stale check that a still-uncommitted winner of the previous visit ruled out (k_round_detect)
       ST_QUERY_NEIGH, ST_QUERY_N,            // accounting queries (kmx_set_profile(m, 2)): how many entered the neighbour disambiguation / were asked
       ST_N };

#define KMX_CLS_TILE 2048                      // k-mers per classification tile (front end)
#define KMX_RSIZE_LOG2 20
#define KMX_RSIZE (1u << KMX_RSIZE_LOG2)      // reservation slots per list (ordered slow path)
#define KMX_FIN_RPT(NHM) ((NHM) <= 8 ? 2 : 1)      // records per finisher thread kept in registers (1024 threads)
#define KMX_FIN_RANGES 8                        // the finisher takes up to this many register loads, in index ranges
#define KMX_APPLY_WGS 8                         // extra workgroups per list in k_reorder that apply the finisher's decisions
enum { KMX_ROUND_FIN_GLOBAL = 1, KMX_ROUND_RESOLVE_GATHER = 2,     // test hooks of kmxk::round (older code paths)
       KMX_ROUND_PENDING = 4,                                        // the previous round's winners are not committed yet: their commit rides with this round's check
       KMX_ROUND_KEEP = 8,
       KMX_ROUND_SMALL_DETECT = 16 };                               // rounds t >= 2: k_round_detect with small tables (their bins hold a few hundred tuples)                                         // this round's winners will be committed beside the next round's check (its detect re-reads the claims)
#define KMX_NSLOW 2                            // contended-record levels, ping-pong: pass s reads level s&1, defers to (s+1)&1
#define KMX_MAX_NSUB 16                        // most grid-wide ordered passes per round
#define KMX_CTR_STRIDE 32                      // ints between per-list counters: one 128-byte line each (same-line atomics serialise)
#define UN_IDX(s, i, nb) ((((s) * (nb)) + (i)) * KMX_CTR_STRIDE)
#define KMX_TILE 1024                          // slots per reorder tile
#define KMX_NTILES (KMX_BUCKET / KMX_TILE)

// Working set of one nb*2^18 block (kmodel.hpp:508-573): lists are index permutations into the block.
struct BlockDev {
	const u64 *kmers;        // [nb*BUCKET][W] this block's k-mers in listing order (buffer i = slice i)
	const u32 *counts;       // [nb*BUCKET]
	// Lists are index permutations, ping-pong by round parity pp.  An entry with bit 31 set is a hole of the last
	// reorder that has not been filled yet: its value is mover[pp][entry & 0x7FFFFFFF] (resolved by check_emit).
	u32 *list[2];            // list[pp][i*BUCKET + x] = index into buffer i of slot x
	u32 *mover[2];           // mover[pp][i*BUCKET + r] = r-th survivor from the right of the previous round
	int *n[2];               // n[pp][i] current list lengths (buff_real_n, kmodel.hpp:277)
	int *tile_cnt[2];        // tile_cnt[pp][i*NTILES + tile] survivors (failed slots) per 1024-slot tile, counted as they fail
	// Per-slot state of a round, double-buffered by the round parity pp like the lists: the winners of round r are committed
	// beside the check of round r+1 (k_round_commit_check), which writes the other half.
	unsigned char *status[2]; // [nb*BUCKET] per slot: 0 undecided -- after the round: a winner nobody contended, committed one round
	                          // later from cidx/cnib/want --, 1 failed (survivor), 2 inserted by the ordered path, 3 contended (transient)
	unsigned char *dfail;    // [nb*BUCKET] set by k_round_detect: a winner of the previous visit to the array, not yet committed when
	                          // this slot was checked, holds one of its positions with the other value; k_round_file makes it a failure
	unsigned char *surv;     // [nb*BUCKET] per k-mer of the block: 1 = it went to the rest table (k_rest_append); the others were inserted
	// contended k-mers; level (s & 1) = still undecided after s grid-wide resolve passes.  A record carries what the
	// ordered slow path needs -- word 0 = slot | bin << 32, then the W packed k-mer words -- so that its latency-bound
	// kernels reach the cells after ONE dependent load.
	u64 *Urec[KMX_NSLOW];    // [nb*BUCKET][1 + W]
	int *Un;                 // [KMX_NSLOW*nb*KMX_CTR_STRIDE], use UN_IDX
	u64 *R;                  // [nb*KMX_RSIZE] epoch-tagged reservations
	u64 *stats;              // [ST_N]
	// claims of a round as a partitioned stream (k_round_check_emit -> k_round_detect -> k_round_commit)
	u32 *uw[2];              // [nb*BUCKET] per candidate: positions it saw untagged (bit j = hash j) | values it wants there << 16 (= its occurrence bin)
	u32 *crec[2];            // [nb*BUCKET][crec_words(nh)] per candidate: cell index of every position + their low 4 bits (kernels.hip CRec)
	u64 *cl_tup[2];          // [nb][bins][KMX_CL_CAP] claim tuples, hash-partitioned by position
	int *cl_cnt[2];          // [nb][KMX_CL_MAXBINS] tuples per bin; a round's counts are reset by the NEXT round's k_round_detect, which reads
	                         // the tuples once more as the delta of the still-uncommitted winners
	int *cl_ovf;             // [nb] a bin of the list overflowed: every candidate of the round takes the ordered path (reset by k_reorder)
	unsigned char *rverdict; // position-range partition, owner's view only: the verdict bytes of the triples it received this round -- a claim
	                         // on a position wanted with both values is reported there (k_round_detect<..., RANGE>); null elsewhere
};

// claim partition: bins per list and slots of the LDS table of one bin (kernels.hip, "claims as a partitioned stream")
#define KMX_CL_BINS_LOG2(NHM) 8
#define KMX_CL_BINS(NHM) (1 << KMX_CL_BINS_LOG2(NHM))
#define KMX_CL_MAXBINS 256
// a position is identified inside a bin by the rest of a bijective 36-bit mix of it (kernels.hip cl_mix): exact for arrays
// of up to 2^36 positions (2e10 coupled k-mers at nh = 7) -- beyond that winners are committed before the next check
#define KMX_CL_MIX_BITS 36
// tuples per bin: 2^18 * nh / bins at most on average (7168 at nh = 7, 16384 at nh = 16), + 25 % and more
#ifndef KMX_CL_CAP                             // (tools/stress_small_tables.py builds a library with a tiny capacity: the overflow path every round)
#define KMX_CL_CAP_OF(NHM) ((NHM) <= 8 ? 12288 : 20480)
#else
#define KMX_CL_CAP_OF(NHM) KMX_CL_CAP
#endif
// 2^TBITS > capacity slots: the table of a bin can always take every tuple.  64 KB of LDS at nh <= 8: two 1024-thread
// workgroups per CU, so one's loads run under the other's table phase
#define KMX_CL_TBITS(NHM) ((NHM) <= 8 ? 14 : 15)
#define KMX_CL_TBITS_SMALL 12                  // the late rounds' form: 16 KB, 256 threads; takes bins of up to 3072 tuples

#define LIST_HOLE 0x80000000u

// One list of a multi-GPU ring round (kernels.hip, k_ring_import / k_ring_export), indexed by list number.
#define KMX_MSG_HDR 8                          // u64 words of message header; word 0 = number of entries
struct RingList {
	int active;              // this rank attempts the list this round
	int n_host;              // >= 0: entries, known on the host (fresh from the stream); < 0: read from the message header
	const u64 *src_kmers;    // n_host >= 0
	const u32 *src_counts;
	const u64 *src_msg;      // n_host < 0: message received from the rank that owns the previous array
	u64 *dst_msg;            // survivors of the round as a message; null in the last round (they go to the rest table)
};
struct RingLists { RingList e[KMX_MAX_NB]; };

// ---- the position-range partition of the coupled arrays over several GPUs (range_kernels.h)
#define KMX_MAX_RANKS 16
#define KMX_RANGE_HDR 4                        // u32 words of a region's header: commits, triples, bulk commits (the first part of the commits, complete early), pad
#define KMX_RANGE_QBITS 27                     // a claim tuple on the owner names the received triple it came from in this many bits (kernels.hip CL_RANGE_TUPLE)
struct RangePlan {
	int rank, world;
	u64 cell_lo[KMX_MAX_RANKS + 1];   // rank q owns the cells [cell_lo[q], cell_lo[q+1]) -- 16 positions each -- of every array
};
// Between a list rank and an owner lies one REGION of `cap` 64-bit words: the winners' commits of a round in front, the next
// round's triples behind them, a header {commits, triples} beside it.  The list rank's side:
struct RangeDev {
	u64 *out[KMX_MAX_RANKS];             // out[q]: the region for owner q -- this rank's own memory (the caller moves the words) or owner q's inbox through a peer mapping
	u32 *hdr_out[KMX_MAX_RANKS];         // hdr_out[q][0..2]: where the header {commits, triples, bulk commits} of that region is left for owner q
	const unsigned char *vin[KMX_MAX_RANKS];   // vin[q][off]: verdict byte of word `off` of the region for owner q
	int *ccnt, *tcnt;        // [KMX_MAX_RANKS * KMX_CTR_STRIDE] commits / triples per destination written since the last seal
	u64 cap;                 // words a region takes.  The worst case of a round (nothing is ever dropped) -- or, when the regions travel as
	                         // fixed-size messages, what the transport ships: a word beyond it is DROPPED and *ovf raised (the build is void:
	                         // the caller repeats it with counted messages; uniform hashing makes that a once-in-never event)
	int *ovf;
	u32 *tidx;               // [nb*BUCKET][nh] where the triple of hash j of a slot went: destination << 28 | index in its region
	u32 *contended;          // [nb*BUCKET] slots of the contended candidates of a list (any order)
	int *n_contended;        // [KMX_MAX_NB * KMX_CTR_STRIDE]
	// k_range_resolve: exact position table per held list (2^rt_bits entries), per-record scratch
	u64 *rt_key;
	u32 *rt_resv, *rt_mark;
	u32 *rt_eidx;            // [nb*BUCKET][nh]
	u32 *rt_um;              // [nb*BUCKET] per contended record (same index as `contended`): its positions wanted with both values
	u32 rt_bits;
};
// which commit words of a region: all, the BULK (the uncontended winners', complete when k_range_apply ends -- the sender still
// orders its contended candidates then) or the LATE ones behind them (k_range_resolve's)
enum { RANGE_ALL = 0, RANGE_BULK = 1, RANGE_LATE = 2 };
// ... and the owner's: what it received from every sender, and where the verdict bytes go
struct RangeIn {
	const u64 *reg[KMX_MAX_RANKS];       // region of sender s
	unsigned char *vout[KMX_MAX_RANKS];  // vout[s][off]: where sender s reads the verdict of word `off` of its region (its own memory, possibly through a peer mapping)
	const u32 *hdr;                      // hdr[hdr_stride s + 0 .. 2]: commits / triples / bulk commits of region s, in device memory (in band) -- or null and
	u32 nc[KMX_MAX_RANKS], nt[KMX_MAX_RANKS];   // the counts by value (the caller moved the words and knows them)
	u32 hdr_stride;                      // u32 words between the headers of consecutive regions
	u32 cap;                             // words of a region that arrived (0: all of them): the counts of a header are cut to it
	int world;
};

// the pre-gather probe of kmxk::round (test hook)
struct RoundProbe {
	bool on;
	hipStream_t side;
	hipEvent_t fork, done;
	u32 *sink;
};

enum { SLOT_UNDECIDED = 0, SLOT_FAILED = 1, SLOT_INSERTED = 2, SLOT_CONTENDED = 3 };

// Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
enum { KC_CLASSIFY = 0, KC_CHECK_CLAIM, KC_VERIFY_COMMIT, KC_SLOW, KC_REORDER, KC_REST, KC_QUERY, KC_DETECT, KC_COMMIT_CHECK, KC_FILE, KC_N };   // CHECK_CLAIM = check + emit, VERIFY_COMMIT = commit
struct KernelProf {
	bool on = false;             // HIP events around every launch of a kernel class
	bool count = false;          // the fused launches also count what they examine, commit and issue (the ST_PIPE_* statistics): their
	                             // accounting variant is slower than the product's, so it is never the one that is timed
	void *events = nullptr;      // std::vector<hipEvent_t>* owned by the host side
	void *spans = nullptr;       // std::vector<int>* : class of span i uses events 2i, 2i+1
	void (*begin)(KernelProf *, int cls, hipStream_t) = nullptr;
	void (*end)(KernelProf *, hipStream_t) = nullptr;
};
#define KPROF_BEGIN(p, cls, st) do { if ((p) && (p)->on) (p)->begin((p), (cls), (st)); } while (0)
#define KPROF_END(p, st) do { if ((p) && (p)->on) (p)->end((p), (st)); } while (0)
