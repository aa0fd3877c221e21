/* include/kmx.h -- C ABI of libkmx.so, the MI355X-native KModel insert/query hot path.
 *
 * The reference (lzhLab/kmcEx) has no FFI layer: its boundary is the header-only C++ class KModel plus
 * two factory functions (kmodel.hpp).  This C ABI is what a binding for that class binds; every entry
 * point cites the reference interface it replaces.  include/kmodel.hpp is the C++ facade with the
 * reference's own names (get_model / init / init_KModel / kmer_to_occ / save / load) on top of it.
 *
 * Conventions
 *   - plain C types, caller-owned buffers, no exceptions cross the boundary;
 *   - every function returns 0 on success, a negative KMX_E_* code otherwise; kmx_last_error() gives
 *     a thread-local message;
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     KMX_E_NODEVICE;
 *   - packed k-mers: W = ceil(k/32) uint64 words per k-mer, word 0 most significant, holding the
 *     2k-bit integer right-aligned, A=0 C=1 G=2 T=3, first base most significant (tools.hpp:63-76);
 *   - "_dev" variants take DEVICE pointers valid on the model's device and enqueue on the model's
 *     stream (kmx_set_stream); they synchronise that stream only where a count has to reach the host.
 */
#ifndef KMX_H
#define KMX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMX_OK            0
#define KMX_E_ARG        -1   /* bad argument / parameter combination                         */
#define KMX_E_NODEVICE   -2   /* no HIP device, or a HIP call failed                           */
#define KMX_E_IO         -3   /* file missing / short / malformed                              */
#define KMX_E_STATE      -4   /* call out of order (e.g. insert before begin, query before build) */
#define KMX_E_RANGE      -5   /* a count outside [ci, cs] (reference: out-of-bounds index)      */
#define KMX_E_NOMEM      -6

typedef struct kmx_model kmx_model;

typedef struct kmx_stats {
	uint64_t n_total;         /* k-mers listed (CKMCFile::KmerCount, kmodel.hpp:429)            */
	uint64_t n_km;            /* k-mers routed to the coupled arrays (kmodel.hpp:433)           */
	uint64_t n_bf[3];         /* k-mers per Bloom-filter class (kmodel.hpp:425-428)             */
	uint64_t attempts;        /* coupled-array insert attempts  (insert_to_array calls, :590)   */
	uint64_t successes;       /* ... that succeeded                                             */
	uint64_t rest_entries;    /* rows of rest.bin (rest.hpp:53 suffix_bin_count)                */
	uint64_t km_byte_size;    /* bytes per tag / value array (kmodel.hpp:437)                   */
	uint64_t byte_km_back;    /* kmodel.hpp:439                                                 */
	uint64_t byte_bf[3], byte_bf_back[3];                          /* kmodel.hpp:411-416        */
	uint64_t fast_commits;    /* successes decided by the uncontended fast path                 */
	uint64_t contended;       /* attempts that went through the ordered slow path -- TIMING-DEPENDENT (a few units in 7e5 from run  */
	uint64_t finisher_iters;  /* to run, like the next one: which candidates meet in flight); iterations of the ordered finisher.   */
	                          /* Every other count of this struct is a function of the input alone.                                 */
	uint64_t blocks, rounds;  /* nb*2^18 blocks and rounds executed                             */
	int32_t  k, ci, cs, nh, nb, bf_num;
	int32_t  device, reserved;
	uint64_t rest_bytes;      /* KRestData::get_all_byte_size (rest.hpp:257-259)                */
	uint64_t piped_attempts;  /* attempts examined / successes committed inside the fused commit|check launches   */
	uint64_t piped_commits;   /* (accounting for the per-kernel roofline: collected only by builds under kmx_set_profile(m, 2)) */
	uint64_t piped_gathers;   /* random 4-byte loads / 32-bit atomic ORs those launches actually ISSUED: the check     */
	uint64_t piped_atomics;   /* stops at the first conflicting group, a winner sets only its untagged positions      */
	uint64_t query_neighbour_calls;  /* packed device queries answered under kmx_set_profile(m, 2) since the last build: how many entered   */
	uint64_t query_accounted;        /* the neighbour disambiguation (kmodel.hpp:344-359) / how many were asked                          */
} kmx_stats;

const char *kmx_last_error(void);
int kmx_device_count(void);
/* Layout version of the structs and array sizes this header declares (kmx_stats, KMX_KERNEL_CLASSES ...): a caller built
 * against an older header must not hand its smaller structs to a newer library -- compare before the first call that
 * takes one (include/kmodel.hpp does, and exits like the reference does on a bad model directory).                    */
#define KMX_ABI_VERSION 5
int kmx_abi_version(void);

/* get_model(ci, cs, num_hash, num_bit)                                     kmodel.hpp:674-677 */
int kmx_create(int ci, int cs, int nh, int nb, kmx_model **out);
/* the reference never frees a model; the handle owns all device memory                        */
int kmx_destroy(kmx_model *m);
/* stream (hipStream_t) used by every later call on this model; NULL = the default stream       */
int kmx_set_stream(kmx_model *m, void *hip_stream);

/* a handle on a given HIP device (kmx_create uses the calling thread's current device, which this call leaves as it was) */
int kmx_create_on(int device, int ci, int cs, int nh, int nb, kmx_model **out);
/* KModel::init(db_file) by several GPUs from ONE process (kmodel.hpp:57-86; main.cpp:143-149 is the caller): models[d] was
 * created on the device it is to use (devices may repeat), all with the same parameters.  One host thread per handle
 * drives the ring of whole arrays (below) with hipMemcpyPeerAsync hand-offs; on return EVERY handle holds the whole model. */
int kmx_build_from_kmc_multi(kmx_model **models, int n_models, const char *db_prefix);
/* The same with the partition chosen.  KMX_PARTITION_RANGE is the north star's: every coupled array cut by POSITION RANGE over
 * the handles (handle q owns the cells [q n/P, (q+1) n/P) of every array; list i of a block lives on handle i % P and its k-mers
 * never move).  A round of the rotation (kmodel.hpp:560-565; check :604-610, set :611-618) is words written by the list
 * handle's kernels STRAIGHT INTO the owner's inbox through a peer mapping (hipDeviceEnablePeerAccess) -- the winners' commits of
 * the round before, then one triple per position of every attempt, with the counts in a header beside them -- and one verdict
 * byte per triple written straight back into the sender's box; two events per handle order the three steps of a round and the
 * host threads only enqueue: no host wait inside a round.  Up to 16 handles; one handle is allowed (the partition's kernels
 * alone).  On return every handle holds the whole model.                                                                       */
#define KMX_PARTITION_RING  0
#define KMX_PARTITION_RANGE 1
/* the same partition with the words of a round as fixed-size RCCL messages instead of peer-mapped stores: one communicator per
 * handle (ncclCommInitAll: the handles must sit on DIFFERENT devices), every region = [header with the counts | capx words]
 * (kmx_range_inband below), a round = two groups of ncclSend / ncclRecv on each handle's stream, nothing of it on the host.
 * librccl.so is opened with dlopen when this is asked for (libkmx.so does not link it).  Should a region overflow, the build
 * is repeated with KMX_PARTITION_RANGE.                                                                                       */
#define KMX_PARTITION_RANGE_RCCL 2
int kmx_build_from_kmc_multi_ex(kmx_model **models, int n_models, const char *db_prefix, int partition);
/* KModel::init(db_file): two passes over the KMC listing + rest build      kmodel.hpp:57-86   */
int kmx_build_from_kmc(kmx_model *m, const char *db_prefix);

/* Host-only view of the KMC listing that init() consumes (CKMCFile::OpenForListing / ReadNextKmer / KmerCount /
 * KmerLength, kmc_file.cpp:66-99, :428-515, :763, :740); needs no GPU.  kmx_kmc_read fills up to `capacity`
 * k-mers (W words each) in listing order, already filtered by the header's [min_count, max_count].          */
int kmx_kmc_info(const char *db_prefix, int *k, uint64_t *total_kmers);
int kmx_kmc_read(const char *db_prefix, uint64_t *kmers, uint32_t *counts, uint64_t capacity, uint64_t *n_read);

/* The same build, streamed.  begin = get_km_kmer_count's result + init_km_bit (kmodel.hpp:423-471):
 * n_bf[i] = number of k-mers with count ci+i (i < bf_num), n_total = KmerCount().                */
int kmx_begin(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total);
/* pass 2 (kmodel.hpp:68-74): batches in LISTING ORDER; host or device pointers                   */
int kmx_insert_batch(kmx_model *m, const uint64_t *kmers, const uint32_t *counts, uint64_t n);
int kmx_insert_batch_dev(kmx_model *m, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n);
/* push_last_to_array / push_last_to_bloomfilter / kld->build()             kmodel.hpp:76-80   */
int kmx_finish(kmx_model *m);
/* one call = pass 1 (device histogram) + begin + insert + finish on a device-resident listing     */
int kmx_build_dev(kmx_model *m, int k, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n);
int kmx_build_host(kmx_model *m, int k, const uint64_t *kmers, const uint32_t *counts, uint64_t n);

/* ---- ONE model built by several GPUs (one process per GPU) -------------------------------------
 * The reference's insert is n_bits OpenMP threads in a rotation: thread i walks buffer i against array (i + t) % n_bits,
 * barrier, next t (insert_with_thread, kmodel.hpp:557-573).  Here the arrays are owned whole by different GPUs and the
 * survivors of a list travel round the ring as *messages*; the order-free filters (Bloom, back, km_back: set_bit is an
 * atomic OR, kmodel.hpp:576-581) are built as per-rank partial filters and merged by OR.  These entry points are the
 * per-rank compute; the exchange between them (RCCL all-to-all / send-recv / broadcast over xGMI) belongs to the caller
 * (kmcex_amd/dist.py, torch.distributed).  All pointers are DEVICE pointers; work is enqueued on the model's stream.     */
typedef struct kmx_ring_list {
	int32_t list;             /* buffer index i of the block; the round attempts array (i + t) % nb (kmodel.hpp:563) */
	int32_t n_host;           /* >= 0: entries, known on the host (round 0, fresh from the stream); -1: read it from src_msg */
	const void *src_kmers;    /* n_host >= 0: packed k-mers / uint32 counts of the list                                */
	const void *src_counts;
	const void *src_msg;      /* n_host < 0: the list as a message (kmx_ring_msg_bytes) left by kmx_ring_round_dev      */
	void *dst_msg;            /* survivors of this round, in list order, as a message; NULL = last round: rest table     */
} kmx_ring_list;
/* pass 1 on this rank's slice (get_km_kmer_count's histogram, kmodel.hpp:423-428); the caller sums over the ranks      */
int kmx_count_classes_dev(kmx_model *m, const uint32_t *d_counts, uint64_t n, uint64_t n_bf[3]);
/* kmx_begin with whole-model figures (every rank sizes and allocates the whole model, kmodel.hpp:402-456)             */
int kmx_shard_begin(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world);
/* pass 2 front end on this rank's slice (kmodel.hpp:70-73): Bloom-class k-mers -> this rank's partial filters;
 * coupled-array k-mers compacted in listing order into d_out_* (capacity n); *n_out on the host                       */
int kmx_shard_classify_dev(kmx_model *m, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n, uint64_t *d_out_kmers, uint32_t *d_out_counts, uint64_t *n_out);
/* bytes of one list message: 64-byte header (word 0 = entries) + 2^18 k-mers + 2^18 counts                             */
uint64_t kmx_ring_msg_bytes(int k);
/* insert_array(buff[i], (i + t) % nb, ...) for the lists this rank holds in round t (kmodel.hpp:543-555, :560-565)     */
int kmx_ring_round_dev(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists);
/* stale-slot duplicate of the final partial block for the lists this rank retired (kmodel.hpp:520-527)                 */
int kmx_ring_stale_dup_dev(kmx_model *m, int first_unused_row);
/* this rank's statistics (attempts, successes, ..., rest_entries = survivors it holds) and its survivor list           */
int kmx_shard_local(kmx_model *m, kmx_stats *partial, void **d_rest_kmers, void **d_rest_counts);
/* kld->build() on the survivors of ALL ranks + the summed statistics; the handle becomes a full replica (kmodel.hpp:80) */
int kmx_shard_complete(kmx_model *m, const uint64_t *d_rest_kmers, const int32_t *d_rest_counts, uint64_t n_rest, const kmx_stats *totals);
/* ---- the same model with every coupled array cut by POSITION RANGE over the ranks: rank q owns the cells
 * [cell_lo[q], cell_lo[q+1]) -- 16 positions each -- of every array; list i of a block lives on rank i % world for the whole
 * block and its k-mers never move.  A round of insert_array (kmodel.hpp:543-555, :560-565; check :604-610, set :611-618) is
 * two exchanges of 64-bit words / bytes that the CALLER moves between the ranks (all-to-all over RCCL): triples out (behind the
 * commits of the round before), verdicts back.  kmx_count_classes_dev, kmx_shard_classify_dev, kmx_ring_stale_dup_dev,
 * kmx_shard_local / _complete and kmx_dev_view are shared with the ring.                                                */
int kmx_range_begin(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world);
/* the regions: region q (cap_words 64-bit words apart) holds what the last emit left for rank q -- commits in front, triples behind  */
int kmx_range_buffers(kmx_model *m, void **d_send, uint64_t *cap_words, uint64_t *cell_lo /* [world + 1] */);
/* step 1, list rank: every position of every attempt of its lists as a triple, by owner rank, appended behind the commits
 * the last kmx_range_resolve_dev left in front of the regions; on the host counts[q] = words for rank q, counts[world + q] =
 * the commit words among them (the header of the region: the owner needs both).
 * t == 0: `lists` = the fresh buffers of the block this rank holds (list, n_host, src_kmers, src_counts)              */
int kmx_range_emit_dev(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists, uint64_t *counts /* [2 world] */);
/* step 2, owner: d_words = the regions of the n_src senders back to back (totals[s] words, the first commits[s] of them commit
 * words).  Applies the commit words (the set loop :611-618 of the round before), then answers every triple with one byte at the
 * same index of d_verdict: conflict | untagged | wanted with both values this round (the bytes of commit words stay unwritten);
 * fewer than 2^27 triples per call (KMX_E_ARG beyond: a round of nb = nh = 16 holds 2^26)                                  */
int kmx_range_verdict_dev(kmx_model *m, int t, const uint64_t *d_words, const uint64_t *totals, const uint64_t *commits, int n_src, uint8_t *d_verdict);
/* step 3, list rank: verdicts in the order the words left (regions back to back, in rank order) -> winners (the contended
 * ones decided in list order); their commits go to the front of the regions for the next emit; reorder_buffer (:529-540),
 * km_back, rest.  Nothing is exchanged now, and no host wait is spent                                                   */
int kmx_range_resolve_dev(kmx_model *m, int t, const uint8_t *d_verdict);
/* FIXED-SIZE MESSAGES (no count on the host, two equal-split all-to-alls per round).  Call between kmx_range_begin and the first
 * emit: every region becomes [header: commits, triples, 0, 0 as uint32 | capx_words 64-bit words], region_words = 2 + capx apart
 * in *d_send -- capx = the mean of a round's fullest exchange + 25 % + 8192 words, far beyond what uniformly hashed positions
 * deviate.  Then kmx_range_emit_dev / kmx_range_flush_dev take counts = NULL and do not wait; the owner reads what arrived -- the
 * world regions of its senders, same layout -- with kmx_range_verdict_inband_dev (verdict byte of word j of region s at
 * d_verdict[s * capx + j]) and kmx_range_commit_inband_dev; kmx_range_resolve_dev takes the bytes that came back, capx per
 * destination.  A word that does not fit its region is DROPPED and the build is void: kmx_shard_local reports it in
 * kmx_stats.reserved (non-zero), and the caller repeats the build with counted messages (kmcex_amd/dist.py does).           */
int kmx_range_inband(kmx_model *m, void **d_send, uint64_t *region_words, uint64_t *capx_words);
int kmx_range_verdict_inband_dev(kmx_model *m, int t, const uint64_t *d_recv, int n_src, uint8_t *d_verdict);
int kmx_range_commit_inband_dev(kmx_model *m, const uint64_t *d_recv, int n_src);
/* end of the build: what is still pending in front of the regions (counts[q] = counts[world + q] = commit words) for a last exchange ... */
int kmx_range_flush_dev(kmx_model *m, uint64_t *counts /* [2 world] */);
/* ... and its application on the owner                                                                                  */
int kmx_range_commit_dev(kmx_model *m, const uint64_t *d_commits, uint64_t n);

/* device memory of filter / array storage for the caller's collectives: which 0 bf[i], 1 bf_back[i], 2 km_back
 * (bytes rounded up to 32-bit words), 3 the cells of coupled array i (value+tag interleaved, 8 bytes per 16 positions)  */
int kmx_dev_view(kmx_model *m, int which, int index, void **ptr, uint64_t *bytes);
/* dst |= src over n 32-bit words (merging partial filters)                                                              */
int kmx_or_words_dev(kmx_model *m, void *d_dst, const void *d_src, uint64_t n_words);

/* vector<int> KModel::kmer_to_occ(vector<string>, t_num)                   kmodel.hpp:90-98   */
int kmx_query_packed(kmx_model *m, const uint64_t *kmers, uint64_t n, int32_t *out);
int kmx_query_packed_dev(kmx_model *m, const uint64_t *d_kmers, uint64_t n, int32_t *d_out);
/* n records of `stride` bytes holding `len` characters each (not NUL-terminated), 2 <= len <= 64.  Strings of
 * the model's k over ACGT take the packed kernel; anything else is hashed byte for byte like the reference does. */
int kmx_query_ascii(kmx_model *m, const char *strs, int len, int stride, uint64_t n, int32_t *out);
/* the same for n separate strings of `len` characters each (strs[i] need not be NUL-terminated): what a
 * vector<string> holds, without concatenating it first */
int kmx_query_strings(kmx_model *m, const char *const *strs, int len, uint64_t n, int32_t *out);

/* KModel::save(dir) -> header, km.bin, rest.bin (dir must exist)           kmodel.hpp:173-206 */
int kmx_save(kmx_model *m, const char *dir);
/* get_model(save_dir) = header parse + KModel::load                        kmodel.hpp:680-696, :209-235 */
int kmx_load(const char *dir, kmx_model **out);

int kmx_get_stats(kmx_model *m, kmx_stats *st);
/* the same, writing at most `size` bytes: kmx_stats only ever grows at its end, so a caller that passes sizeof of ITS kmx_stats
 * is safe against a newer library (kmx_get_stats and kmx_shard_local write the library's full struct: compare kmx_abi_version()
 * first).  ABI 5 (this header): kmx_stats + query_neighbour_calls / query_accounted; kmx_range_* move regions with in-band
 * headers; piped_* are filled only by builds under kmx_set_profile(m, 2) (since ABI 4).                                    */
int kmx_get_stats_n(kmx_model *m, void *st, uint64_t size);

/* Raw on-disk-layout views for byte-level parity checks (kmodel.hpp:183-202).
 * which: 0 bf[i], 1 bf_back[i], 2 km_back, 3 value array i (bit_array_1), 4 tag array i (bit_array_2),
 *        5 insert-time scratch left in array i (none any more: always zero; kept for the tests' invariant).   */
int kmx_download(kmx_model *m, int which, int index, uint8_t *dst, uint64_t capacity, uint64_t *written);

/* Primitive known-answer surface, evaluated ON THE DEVICE (tools.hpp:16-50, :160-167).
 * hashes[n*n_seeds]: murmur_hash64 of the k-mer string (whole=1) or its (k-2)-mer (whole=0).       */
int kmx_debug_hash(int k, const uint64_t *kmers, uint64_t n, const uint32_t *seeds, int n_seeds, int whole, uint64_t *hashes);
int kmx_debug_min_kmer(int k, const uint64_t *kmers, uint64_t n, uint64_t *out);
/* out[i] = h[i] % d with the device's exact reciprocal modulo (`% length`, kmodel.hpp:378,503,600,633), d up to 2^63 */
int kmx_debug_mod(const uint64_t *h, uint64_t n, uint64_t d, uint64_t *out);
/* Host half of kmx_query_strings / kmx_query_ascii, on its own (no device needed): strings -> packed k-mers
 * (2 bits per base, first base most significant, ceil(len/32) words each, tools.hpp:63-76).  strs != NULL: n separate
 * strings; else n records of `stride` bytes in `flat`.  *clean = 0 when a string holds anything but ACGT (such a
 * batch travels as bytes and is answered by the byte-string kernel).                                */
int kmx_debug_pack_strings(const char *const *strs, const char *flat, int len, int stride, uint64_t n, uint64_t *packed, int *clean);
/* OccuBin tables (occu_bin.hpp:27-83): bin_of_occ[cs+1], mean_of_bin[2^nh]                         */
int kmx_occubin(int cs, int nh, uint32_t *bin_of_occ, uint32_t *mean_of_bin);

/* Random-access ceiling of the memory system for this access pattern: `touches` random touches over
 * `bytes` of device memory, 8 per lane like one k-mer on one array; seconds per launch out.  mode: 0 8-byte loads,
 * 1 64-bit atomic OR (agent scope, what the insert uses), 2 byte stores, 3 byte loads, 4 8-byte stores,
 * 5 32-bit atomic OR (what the insert uses on its 4-byte cells), 6 64-bit atomic OR at workgroup scope, 7 64-bit atomic OR
 * returning the old word, 8 4-byte loads (what the insert's check uses); 9 / 10 / 11 the same non-temporal / agent-scope
 * (sc1) / with 16 loads in flight per lane; 12 32-bit atomic OR with 3/8 of the lanes active (`touches` counts all lanes);
 * 20 modes 8 and 5 side by side on two streams, each on its own buffer of `bytes` (seconds until both are done).          */
int kmx_microbench(int mode, uint64_t bytes, uint64_t touches, int iters, double *seconds);

/* Per-kernel-class timing with HIP events recorded on the model's stream around each launch (off by default).
 * classes: 0 classify(+Bloom insert) 1 check (+ claim emission) 2 commit 3 ordered slow path 4 reorder 5 rest table (sort + index) 6 query
 * 7 detect (opposite claims) 8 commit of the previous round beside the check of the next (k_round_commit_check)
 * 9 file (k_round_file).  seconds[KMX_KERNEL_CLASSES], launches[KMX_KERNEL_CLASSES] accumulate until reset: a caller
 * must size both arrays with the macro of the header it was compiled against and check kmx_kernel_classes() == it.  */
#define KMX_KERNEL_CLASSES 10
int kmx_kernel_classes(void);           /* what this library writes: KMX_KERNEL_CLASSES of ITS header */
/* on = 1: time the kernel classes of the next builds / queries; on = 2: no timing, but the fused launches of the next builds
 * run their ACCOUNTING variant (kmx_stats piped_*: what they examined, committed and issued) -- it is slower than the
 * product's kernel, so it is never the one that is timed; 0: neither                                                    */
/* (any other value: KMX_E_ARG)                                                                                          */
int kmx_set_profile(kmx_model *m, int on);
int kmx_get_kernel_times(kmx_model *m, double *seconds, uint64_t *launches, int reset);

/* timing of the last build with HIP events on the model's stream: total_s = whole build call;
 * insert_kernels_s = time inside the insert kernels (only collected while kmx_set_profile is on, else 0) */
int kmx_last_build_seconds(kmx_model *m, double *insert_kernels_s, double *total_s);

#ifdef __cplusplus
}
#endif
#endif
