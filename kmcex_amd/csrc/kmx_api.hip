// kmx_api.hip -- host orchestration + the C ABI of include/kmx.h.
//
// Mirrors KModel's life cycle (kmodel.hpp:45-235, :674-696) with the bit arrays resident in HBM.
// There is no CPU compute path in this file: every insert/query runs in the gfx950 kernels of
// kernels.hip; the host only sizes, streams, sorts the (small) rest table and does file I/O.
#include "../../include/kmx.h"
#include "kmc_reader.h"
#include "kmx_types.h"
#include "strpack.h"

#include <algorithm>
#include <atomic>
#include <exception>
#include <new>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>                                             // types and prototypes only: librccl.so is opened with dlopen when the RCCL transport is asked for
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace kmxk {
void histogram(const u32 *, u64, int, int, int, u64 *, u64 *, hipStream_t);
int classify_tiles(u64 n);
void classify_count(const ModelDev &, const u64 *, const u32 *, u64, u64, int *, int *, int *, u64 *, const BitScatter &, int, hipStream_t, KernelProf *);
void classify_scatter(const ModelDev &, const u64 *, const u32 *, u64, const int *, u64 *, u32 *, u64, hipStream_t);
void block_init(const BlockDev &, int, int, int, hipStream_t);
void round(const ModelDev &, const BlockDev &, int, int, int, u64 *, int, hipStream_t, KernelProf *, const KmbackJob *, const BitScatter *, const RoundProbe *probe = nullptr);
void commit_flush(const ModelDev &, const BlockDev &, int, int, hipStream_t, KernelProf *);
void rest_append(const ModelDev &, const BlockDev &, int, int, int, u64 *, int *, unsigned long long *, u64 *, int *, u64 *, hipStream_t, int istride = 1);
void kmback_emit(const ModelDev &, const BlockDev &, const u64 *, const unsigned char *, int, int, int, int, int, const BitScatter &, hipStream_t, int istride = 1);
void bs_apply(const BitScatter &, hipStream_t);
void ring_import(const ModelDev &, const BlockDev &, int, const RingLists &, u64 *, u32 *, hipStream_t);
void ring_export(const ModelDev &, const BlockDev &, int, const RingLists &, u64 *, hipStream_t);
void or_words(u32 *, const u32 *, u64, hipStream_t);
void range_emit(const ModelDev &, const BlockDev &, const RangeDev &, const RangePlan &, int, int, bool, hipStream_t);
void range_seal(const RangeDev &, const RangePlan &, hipStream_t);
void range_verdict(const ModelDev &, const BlockDev &, int *, int, const RangeIn &, unsigned char *, int, bool, hipStream_t);
void range_apply(const ModelDev &, const BlockDev &, const RangeDev &, const RangePlan &, int, int, bool, hipStream_t);
void range_resolve(const ModelDev &, const BlockDev &, const RangeDev &, const RangePlan &, int, int, bool, hipStream_t);
void range_commit_apply(const ModelDev &, const RangeIn &, int, hipStream_t);
void query(const ModelDev &, const u64 *, u64, int *, hipStream_t, KernelProf *, u64 *acct = nullptr);
void query_ascii(const ModelDev &, int, const unsigned char *, int, u64, int *, hipStream_t);
void cells_from_disk(const unsigned char *, const unsigned char *, u64, cell_t *, u64, hipStream_t);
void cells_to_disk(const cell_t *, u64, u64, int, unsigned char *, hipStream_t);
void debug_hash(int, const u64 *, u64, const u32 *, int, int, u64 *, hipStream_t);
void debug_min_kmer(int, const u64 *, u64, u64 *, hipStream_t);
void debug_mod(const u64 *, u64, u64, u64 *, hipStream_t);
void micro(int, u64 *, u64, u64, u64, u64 *, hipStream_t);
hipError_t rest_sort(const u64 *, const int *, u64, int, int, u64 *, int *, hipStream_t);
hipError_t rest_index(const u64 *, u64, int, int, int, int *, int *, u64 *, int *, hipStream_t);
void rest_expand(const int *, const int *, const u64 *, int, int, int, int, u64 *, hipStream_t);
void rest_accel(const u64 *, u64, int, int, int, const int *, const int *, const u64 *, int, u32 *, u64 *, hipStream_t);
void rest_suffix_bytes(const u64 *, u64, int, int, unsigned char *, hipStream_t);
}   // namespace kmxk

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

// Every behaviour-changing environment variable of this library is a TEST HOOK (forced code paths, shrunken tables): they
// are read only when KMX_TEST_HOOKS=1 is set as well, so that a stray variable cannot move a user onto an untuned path.
// (KMX_INIT_TRACE only prints phase times and is not gated.)
static const char *hook_env(const char *name)
{
	static const bool on = [] { const char *e = getenv("KMX_TEST_HOOKS"); return e && atoi(e) != 0; }();
	return on ? getenv(name) : nullptr;
}
static int fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}
#define HIPCHK(call)                                                                             \
	do {                                                                                         \
		hipError_t e_ = (call);                                                                  \
		if (e_ != hipSuccess) return fail(KMX_E_NODEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
	} while (0)

extern "C" const char *kmx_last_error(void) { return g_err; }

// temporaries of one call: freed on every way out
struct DevMem {
	void *p = nullptr;
	DevMem() = default;
	DevMem(const DevMem &) = delete;
	DevMem &operator=(const DevMem &) = delete;
	~DevMem() { if (p) hipFree(p); }
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
	template <typename T> T *as() const { return (T *)p; }
	void *release() { void *q = p; p = nullptr; return q; }
};
template <typename F> struct ScopeExit {
	F f;
	explicit ScopeExit(F g) : f(g) {}
	~ScopeExit() { f(); }
};
template <typename F> static ScopeExit<F> scope_exit(F f) { return ScopeExit<F>(f); }

static int kmx_device_count_impl(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// ------------------------------------------------------------------------------------------ OccuBin
// occu_bin.hpp:27-83: three zones -- identity below e1 = 2^nh/4, 2^nh/2 bins 3 wide, 2^nh/4 bins `cap` wide.
static int occubin_tables(int cs, int nh, std::vector<u32> &bin_of_occ, std::vector<u32> &mean_of_bin)
{
	const int mc = cs + 1, e3 = 1 << nh, e1 = e3 / 4, e2 = e1 + e3 / 2;
	bin_of_occ.assign(mc, 0xFFFFFFFFu);
	std::vector<u32> mean_of_occ(mc, 0xFFFFFFFFu);
	int start = e1;
	for (int i = 0; i < e3 / 2; i++, start += 3)
		for (int j = 0; j < 3; j++) {
			if (start + j >= mc) return -1;                    // cs too small for nh (reference overruns, Q11)
			mean_of_occ[start + j] = start + 1;
			bin_of_occ[start + j] = e1 + i;
		}
	const int nz3 = e3 / 4, cap = (mc - start) / nz3;
	for (int i = 0; i < nz3; i++, start += cap)
		for (int j = 0; j < cap; j++) {
			mean_of_occ[start + j] = (2 * start + cap) / 2;
			bin_of_occ[start + j] = e2 + i;
		}
	for (int i = start; i < mc; i++) {
		mean_of_occ[i] = (2 * start - cap) / 2;
		bin_of_occ[i] = e3 - 1;
	}
	for (int i = 0; i < e1 && i < mc; i++) bin_of_occ[i] = i;
	mean_of_bin.assign(e3, 0);                                 // unordered_map::operator[] on a missing bin yields 0
	for (int b = 0; b < e1; b++) mean_of_bin[b] = b;
	std::vector<char> seen(e3, 0);
	for (int i = e1; i < mc; i++) {                            // map::insert keeps the first mean per bin
		u32 b = bin_of_occ[i];
		if (b < (u32)e3 && !seen[b]) { seen[b] = 1; if ((int)b >= e1) mean_of_bin[b] = mean_of_occ[i]; }
	}
	return 0;
}

static int kmx_occubin_impl(int cs, int nh, uint32_t *bin_of_occ, uint32_t *mean_of_bin)
{
	if (nh < 3 || nh > KMX_MAX_NH || cs < 1) return fail(KMX_E_ARG, "bad cs/nh");
	std::vector<u32> b, m;
	if (occubin_tables(cs, nh, b, m)) return fail(KMX_E_ARG, "cs=%d too small for nh=%d", cs, nh);
	memcpy(bin_of_occ, b.data(), b.size() * 4);
	memcpy(mean_of_bin, m.data(), m.size() * 4);
	return KMX_OK;
}

// ------------------------------------------------------------------------------------------ rest table (host side)
// rest.hpp:46-65 arrays in their on-disk form (Appendix B.2) + the device form used by k_query.
struct RestTable {
	int k = 0, pre_len = 0, map_size = 0, pre_buffer_size = 0, suff_group = 0;
	u64 suff_bin_size = 0, entries = 0;
	std::vector<int> hash2index, pre_buffer, count_bin;
	std::vector<unsigned char> suffix_bin;
	bool host_valid = false;       // the four on-disk arrays above are materialised (load, or lazily at save)
	// device
	int *d_h2i = nullptr, *d_pre = nullptr, *d_cnt = nullptr;     // d_cnt: counts in sorted order
	u64 *d_suf = nullptr;
	u64 *d_sorted = nullptr;       // sorted k-mers [entries][W] (kept for save after a device build)
	int fbits = 0;                 // lookup accelerators (see ModelDev)
	u32 *d_fine = nullptr;
	u64 *d_q = nullptr;
};
static int rest_prefix_len(int k) { for (int i = 7; i >= 3; i--) if ((k - i) % 4 == 0) return i; return 3; }   // rest.hpp:78-83

struct RestEnt { u64 w[2]; int c; };

// ------------------------------------------------------------------------------------------ the model
enum { ST_EMPTY = 0, ST_BUILDING = 1, ST_READY = 2 };

struct kmx_model {
	int ci = 1, cs = 1023, nh = 7, nb = 5, k = 0, bf_num = 1, W = 1;
	int device = 0, state = ST_EMPTY;
	hipStream_t stream = nullptr;
	std::vector<u32> h_bin_of_occ, h_mean_of_bin;
	u32 *d_bin_of_occ = nullptr, *d_mean_of_bin = nullptr;
	u64 n_total = 0, n_km = 0, n_bf[3] = {0, 0, 0};
	u64 byte_bf[3] = {0, 0, 0}, byte_bf_back[3] = {0, 0, 0}, km_byte_size = 0, byte_km_back = 0, ncells = 0;
	u32 *d_bloom = nullptr;                                    // ONE slab for bf[i] / bf_back[i], back to back (d_bf / d_bf_back point into it)
	u64 cap_bloom = 0, bloom_words = 0, bf_woff[3] = {0, 0, 0}, bf_back_woff[3] = {0, 0, 0};
	u32 *d_bf[3] = {nullptr, nullptr, nullptr}, *d_bf_back[3] = {nullptr, nullptr, nullptr}, *d_km_back = nullptr;
	cell_t *d_cells[KMX_MAX_NB] = {nullptr};
	u64 cap_km_back = 0, cap_cells[KMX_MAX_NB] = {0};          // bytes allocated
	RestTable rest;
	ModelDev md;
	// ---- build-time state
	u64 *d_stg_kmers = nullptr;
	u32 *d_stg_counts = nullptr;
	u64 stg_n = 0, stg_cap = 0;
	BlockDev bd;
	void *d_block_scratch = nullptr;
	u64 scratch_bytes = 0;
	int scratch_nb = 0, scratch_W = 0;
	u64 *d_rest_kmers = nullptr;
	int *d_rest_counts = nullptr;
	unsigned long long *d_rest_n = nullptr;
	u64 rest_cap = 0, rest_upper = 0;
	u64 *d_stale_kmers = nullptr;
	int *d_stale_counts = nullptr;
	u64 *d_stats = nullptr, *d_nbf = nullptr;
	int *d_tile_cnt = nullptr, *d_tile_off = nullptr, *d_total = nullptr;
	u64 tile_cap = 0;                                          // tiles the two arrays above can hold
	int *d_totals = nullptr, *h_totals = nullptr;              // per-chunk totals of one insert_batch call (h_: pinned)
	u64 totals_cap = 0;
	int *h_total = nullptr;                                    // pinned
	u64 *h_feedback = nullptr;                                 // pinned: ST_MAX_U0 as of some earlier block (heuristic input)
	u64 *d_feedback = nullptr;                                 // the same words as the device sees them (k_rest_append writes them)
	u64 epoch = 1, blocks = 0, rounds = 0;
	// Bloom-class k-mers of the front end as a partitioned bit-set over the slab (k_classify_count emits, k_bs_apply sweeps)
	BitScatter blm;
	u32 *d_blm_tup = nullptr;
	u64 blm_tup_cap = 0;
	int *d_blm_cnt = nullptr;
	bool blm_deferred = false;
	int blm_sweep_every = 1;                                   // chunks of the front end per sweep of the slab
	// km_back insert as a partitioned bit-set (k_kmback_emit per round, k_bs_apply every few blocks)
	BitScatter kmb;
	u32 *d_kmb_tup = nullptr;
	int *d_kmb_cnt = nullptr;
	u64 kmb_tup_cap = 0;                                       // tuples allocated
	u64 kmb_pending = 0, kmb_budget = 0;                       // upper bound of tuples emitted since the last apply / what the bins take
	bool kmb_deferred = false;
	bool dbg_kmb_direct = false;                               // KMX_KMB_DIRECT=1: the atomic path at every size (test hook)
	unsigned char *d_surv[2] = {nullptr, nullptr};             // survivor flags of the current and of the previous block (inside the scratch slab)
	KmbackJob kmb_job = {nullptr, nullptr, 0, 0, 0};            // km_back emission the last block still owes (hosted by the next block's finisher launches)
	bool dbg_kmb_host = true;                                  // KMX_KMB_HOST=0: every block emits in a launch of its own (test hook)
	u32 *d_bs2_tup = nullptr;                                  // second level of the two bit-sets (big filters), shared: [bins * tiles][cap2]
	int *d_bs2_cnt = nullptr;
	u64 bs2_tup_cap = 0, bs2_cnt_cap = 0;
	// feed of KModel::init(db): pinned slots, device buffers, copy stream -- kept across calls on the handle (allocating and
	// freeing ~300 MB of pinned + device memory costs ~30 ms per call on this stack)
	struct KmcFeed {
		unsigned char *raw[2] = {nullptr, nullptr}, *draw[2] = {nullptr, nullptr};
		u64 *km[2] = {nullptr, nullptr}, *dk[2] = {nullptr, nullptr}, *d_lut = nullptr, *h_lut = nullptr;   // h_lut: pinned
		u32 *cnt[2] = {nullptr, nullptr}, *dc[2] = {nullptr, nullptr};
		size_t raw_cap = 0, km_cap = 0, dk_cap = 0, lut_cap = 0;   // bytes / bytes / k-mer words / entries
		hipStream_t copy = nullptr;
		hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
	} feed;
	// feed of kmer_to_occ(vector<string>): three slots of pinned + device buffers (strings in, answers out), a copy stream
	// each way -- kept on the handle like the KMC feed
	struct QueryFeed {
		static const int S = 3;
		unsigned char *h_in[S] = {nullptr, nullptr, nullptr}, *d_in[S] = {nullptr, nullptr, nullptr};
		int32_t *h_out[S] = {nullptr, nullptr, nullptr}, *d_out[S] = {nullptr, nullptr, nullptr};
		size_t in_cap = 0, out_cap = 0;                            // bytes per slot / answers per slot
		hipStream_t to_dev = nullptr, to_host = nullptr;
		hipEvent_t ev_in[S] = {nullptr, nullptr, nullptr}, ev_k[S] = {nullptr, nullptr, nullptr}, ev_out[S] = {nullptr, nullptr, nullptr};
	} qfeed;
	// position-range partition over several GPUs (kmx_range_*): this rank's exchange buffers, its resolver tables, and the
	// owner-side view of the block working set (overflow flags + padded bin counters for the detect kernel on received claims)
	struct RangeState {
		bool on = false;
		bool pending = false;                                      // a round was resolved since the last seal: its winners' commits sit in front of the regions
		bool mailbox = false;                                      // the regions live in the owners' inboxes (kmx_build_from_kmc_multi_ex), not in d_send
		RangeDev rd = {};
		RangePlan plan = {};
		RangeIn in = {};                                           // mailbox transport: this rank's inbox as its owner-side kernels see it
		BlockDev obd = {};
		int *d_oovf = nullptr;
		int *d_opcnt = nullptr;                                    // the owner's claim-bin counters, one per 128-byte line (k_range_verdict)
		unsigned char *d_lver = nullptr;                           // one verdict byte per triple received in a round (detect answers there, k_range_ship sends it on)
		// caller-moved transport (kmx_range_*_dev): the regions and their headers in this rank's memory
		u64 *d_send = nullptr;
		u32 *d_hdr = nullptr, *h_hdr = nullptr;                    // [KMX_MAX_RANKS][2]; h_: pinned copy
		u64 sent_tot[KMX_MAX_RANKS] = {};                          // words per destination of the last emit (where each region's verdicts start in what comes back)
		bool inband = false;                                       // the regions travel as fixed-size messages [header | capx words] (kmx_range_inband)
		u64 capx = 0, cap_full = 0;                                // words per region shipped / the worst case of a round
		int *d_ovf = nullptr;                                      // raised by a launch that had to drop a word (fixed-size messages only)
		// mailbox transport: what the other ranks write into (through peer mappings when they sit on other devices)
		u64 *d_inbox = nullptr;                                    // [world][cap] region of sender s
		u32 *d_in_hdr = nullptr;                                   // [world][2]
		unsigned char *d_vbox = nullptr;                           // [world][cap] verdict bytes of the words this rank sent to owner q
		u64 alloc_key = 0;                                         // nb, nh, world, transport the buffers were sized for
		int n0[KMX_MAX_NB] = {};                                   // entries of the held lists when the block came in (km_back is emitted once per block)
	} range;
	RoundProbe probe = {false, nullptr, nullptr, nullptr, nullptr};   // KMX_PREGATHER_PROBE=1 (test hook): kernels.hip k_probe_pregather
	bool ring = false;                                         // built by several GPUs (kmx_shard_begin): this handle holds ONE rank's share
	int ring_rank = 0, ring_world = 1;
	int nsub = 1;                                              // grid-wide ordered passes in round 0 (see process_block)
	// The winners of a round are committed beside the check of the NEXT round (k_round_commit_check), also across a block
	// boundary: `pending` says that the round with list parity pp ^ 1 and round index pending_t still owes its commit.
	int pp = 0;                                                // list / per-slot-state parity of the next round (alternates from round to round, never reset inside a build)
	bool pending = false;
	int pending_t = 0;
	bool defer = true;                                         // false: every round commits in a launch of its own before the next check (arrays above 2^KMX_CL_MIX_BITS = 2^36 positions; KMX_PIPE=0)
	bool dbg_no_defer = false;
	// test hooks, read from the environment by kmx_begin: forced pass counts, forced older code
	// paths (KMX_ROUND_* flags of kmx_types.h), a trace of the pass-count controller
	int dbg_nsub0 = -1, dbg_nsub1 = -1, dbg_flags = 0;
	int dbg_small_detect = -1;                                 // KMX_SMALL_DETECT=0/1: force the form of the late rounds' k_round_detect (test hook)
	bool dbg_ctrl = false;
	u64 h_stats[ST_N] = {0};
	double t_insert_kernels = 0, t_total = 0;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	// per-kernel-class timing (kmx_set_profile)
	KernelProf prof;
	std::vector<hipEvent_t> prof_events;
	std::vector<int> prof_spans;
	size_t prof_used = 0;
	double kc_seconds[KC_N] = {0};
	uint64_t kc_launches[KC_N] = {0};
};

static void prof_begin(KernelProf *p, int cls, hipStream_t st)
{
	auto *ev = (std::vector<hipEvent_t> *)p->events;
	auto *sp = (std::vector<int> *)p->spans;
	const size_t need = 2 * sp->size() + 2;
	while (ev->size() < need) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev->push_back(e); }
	sp->push_back(cls);
	hipEventRecord((*ev)[2 * sp->size() - 2], st);
}
static void prof_end(KernelProf *p, hipStream_t st)
{
	auto *ev = (std::vector<hipEvent_t> *)p->events;
	auto *sp = (std::vector<int> *)p->spans;
	if (sp->empty() || ev->size() < 2 * sp->size()) return;
	hipEventRecord((*ev)[2 * sp->size() - 1], st);
}
// fold the recorded spans into kc_seconds / kc_launches (stream must be idle)
static void prof_collect(kmx_model *m)
{
	for (size_t i = 0; i < m->prof_spans.size(); i++) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, m->prof_events[2 * i], m->prof_events[2 * i + 1]) == hipSuccess) {
			m->kc_seconds[m->prof_spans[i]] += ms * 1e-3;
			m->kc_launches[m->prof_spans[i]]++;
		}
	}
	m->prof_spans.clear();
}

static const u64 kChunk = u64(1) << 23;                       // k-mers classified per pass of the front end

static std::atomic<unsigned long long> g_malloc_ns{0};       // time inside hipMalloc (KMX_CTRL_DEBUG=1 prints it per build)
static hipError_t timed_malloc(void **p, u64 bytes)
{
	const auto t0 = std::chrono::steady_clock::now();
	const hipError_t e = hipMalloc(p, bytes);
	g_malloc_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
	return e;
}

template <typename T> static int dalloc(T **p, u64 n_elems, bool zero, hipStream_t st)
{
	*p = nullptr;
	u64 bytes = n_elems * sizeof(T);
	if (!bytes) bytes = 16;
	HIPCHK(timed_malloc((void **)p, bytes));
	if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, st));
	return KMX_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// grow-only device buffer: reallocated only when `need` exceeds what is there (bench loops rebuild the same sizes)
template <typename T> static int ensure(T **p, u64 *cap_bytes, u64 need_bytes, bool zero, hipStream_t st)
{
	if (!need_bytes) need_bytes = 16;
	if (!*p || *cap_bytes < need_bytes) {
		if (*p) { HIPCHK(hipStreamSynchronize(st)); hipFree(*p); *p = nullptr; }
		HIPCHK(timed_malloc((void **)p, need_bytes));
		*cap_bytes = need_bytes;
	}
	if (zero) HIPCHK(hipMemsetAsync(*p, 0, need_bytes, st));
	return KMX_OK;
}

static void free_build_state(kmx_model *m)
{
	hipFree(m->d_stg_kmers); m->d_stg_kmers = nullptr;
	hipFree(m->d_stg_counts); m->d_stg_counts = nullptr;
	hipFree(m->d_block_scratch); m->d_block_scratch = nullptr;
	hipFree(m->d_rest_kmers); m->d_rest_kmers = nullptr;
	hipFree(m->d_rest_counts); m->d_rest_counts = nullptr;
	hipFree(m->d_rest_n); m->d_rest_n = nullptr;
	hipFree(m->d_stale_kmers); m->d_stale_kmers = nullptr;
	hipFree(m->d_stale_counts); m->d_stale_counts = nullptr;
	hipFree(m->d_tile_cnt); m->d_tile_cnt = nullptr;
	hipFree(m->d_tile_off); m->d_tile_off = nullptr;
	m->tile_cap = 0;
	hipFree(m->d_total); m->d_total = nullptr;
	hipFree(m->d_blm_tup); m->d_blm_tup = nullptr;
	hipFree(m->d_blm_cnt); m->d_blm_cnt = nullptr;
	m->blm_tup_cap = 0;
	hipFree(m->d_kmb_tup); m->d_kmb_tup = nullptr;
	hipFree(m->d_kmb_cnt); m->d_kmb_cnt = nullptr;
	m->kmb_tup_cap = 0;
	hipFree(m->d_bs2_tup); m->d_bs2_tup = nullptr;
	hipFree(m->d_bs2_cnt); m->d_bs2_cnt = nullptr;
	m->bs2_tup_cap = m->bs2_cnt_cap = 0;
	m->stg_n = m->stg_cap = 0;
}

static void free_rest_dev(RestTable &r)
{
	hipFree(r.d_h2i); hipFree(r.d_pre); hipFree(r.d_cnt); hipFree(r.d_suf); hipFree(r.d_sorted); hipFree(r.d_fine); hipFree(r.d_q);
	r.d_h2i = r.d_pre = r.d_cnt = nullptr;
	r.d_suf = r.d_sorted = r.d_q = nullptr;
	r.d_fine = nullptr;
}

static void free_arrays(kmx_model *m)
{
	hipFree(m->d_bloom); m->d_bloom = nullptr; m->cap_bloom = 0;
	for (int i = 0; i < 3; i++) m->d_bf[i] = m->d_bf_back[i] = nullptr;
	hipFree(m->d_km_back); m->d_km_back = nullptr;
	for (int a = 0; a < KMX_MAX_NB; a++) { hipFree(m->d_cells[a]); m->d_cells[a] = nullptr; m->cap_cells[a] = 0; }
	m->cap_km_back = 0;
	free_rest_dev(m->rest);
}

static int create_device_side(kmx_model *m);
static int kmx_destroy_impl(kmx_model *m);
static int kmx_create_impl(int ci, int cs, int nh, int nb, kmx_model **out)
{
	if (!out) return fail(KMX_E_ARG, "null out");
	*out = nullptr;
	if (nh < 3 || nh > KMX_MAX_NH || nb < 1 || nb > KMX_MAX_NB || ci < 1 || cs < ci)
		return fail(KMX_E_ARG, "bad parameters ci=%d cs=%d nh=%d nb=%d", ci, cs, nh, nb);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		return fail(KMX_E_NODEVICE, "no HIP device: libkmx has no CPU fallback");
	kmx_model *m = new kmx_model();
	m->ci = ci; m->cs = cs; m->nh = nh; m->nb = nb;
	m->bf_num = ci == 1 ? 1 : 3;                               // kmodel.hpp:50
	if (occubin_tables(cs, nh, m->h_bin_of_occ, m->h_mean_of_bin)) { delete m; return fail(KMX_E_ARG, "cs=%d too small for nh=%d", cs, nh); }
	const int rc = create_device_side(m);
	if (rc) { kmx_destroy_impl(m); return rc; }                // whatever was allocated so far goes with the handle
	*out = m;
	return KMX_OK;
}

static int create_device_side(kmx_model *m)
{
	HIPCHK(hipGetDevice(&m->device));
	HIPCHK(hipMalloc((void **)&m->d_bin_of_occ, m->h_bin_of_occ.size() * 4));
	HIPCHK(hipMalloc((void **)&m->d_mean_of_bin, m->h_mean_of_bin.size() * 4));
	HIPCHK(hipMemcpy(m->d_bin_of_occ, m->h_bin_of_occ.data(), m->h_bin_of_occ.size() * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(m->d_mean_of_bin, m->h_mean_of_bin.data(), m->h_mean_of_bin.size() * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMalloc((void **)&m->d_stats, ST_N * 8));
	HIPCHK(hipMalloc((void **)&m->d_nbf, 3 * 8));
	HIPCHK(hipMemset(m->d_stats, 0, ST_N * 8));
	HIPCHK(hipHostMalloc((void **)&m->h_total, 64));
	HIPCHK(hipHostMalloc((void **)&m->h_feedback, 64, hipHostMallocMapped));
	HIPCHK(hipHostGetDevicePointer((void **)&m->d_feedback, m->h_feedback, 0));
	// [2]: the fullest claim bin of a late round (t >= 2), which picks the form of their k_round_detect: until the device reports
	// one, a guess -- a quarter of a list still alive, every survivor a candidate on all its positions -- so that wide
	// configurations (nh >= 9: 2304 > 2048) start on the full-size tables instead of overflowing the small ones
	m->h_feedback[0] = ~0ULL; m->h_feedback[1] = 0; m->h_feedback[2] = (u64)KMX_BUCKET * (u64)m->nh / KMX_CL_MAXBINS / 4;
	{
		auto env_int = [](const char *name, int dflt) { const char *v = hook_env(name); return v ? atoi(v) : dflt; };
		m->dbg_nsub0 = env_int("KMX_NSUB0", -1);
		m->dbg_nsub1 = env_int("KMX_NSUB1", -1);
		m->dbg_flags = (env_int("KMX_FIN_GLOBAL", 0) ? KMX_ROUND_FIN_GLOBAL : 0) | (env_int("KMX_RESOLVE_GATHER", 0) ? KMX_ROUND_RESOLVE_GATHER : 0);
		m->dbg_small_detect = env_int("KMX_SMALL_DETECT", -1);
		m->dbg_no_defer = env_int("KMX_PIPE", 1) == 0;                      // KMX_PIPE=0: commit after every round, check against the committed state only
		m->dbg_ctrl = env_int("KMX_CTRL_DEBUG", 0) != 0;
		m->dbg_kmb_direct = env_int("KMX_KMB_DIRECT", 0) != 0;
		m->dbg_kmb_host = env_int("KMX_KMB_HOST", 1) != 0;
	}
	HIPCHK(hipEventCreate(&m->ev0));
	HIPCHK(hipEventCreate(&m->ev1));
	m->prof.events = &m->prof_events; m->prof.spans = &m->prof_spans; m->prof.begin = prof_begin; m->prof.end = prof_end;
	return KMX_OK;
}

static void free_feed(kmx_model *m)
{
	auto &f = m->feed;
	if (f.copy) { hipStreamSynchronize(f.copy); hipStreamDestroy(f.copy); f.copy = nullptr; }
	for (int s = 0; s < 2; s++) {
		if (f.raw[s]) hipHostFree(f.raw[s]);
		if (f.km[s]) hipHostFree(f.km[s]);
		if (f.cnt[s]) hipHostFree(f.cnt[s]);
		hipFree(f.draw[s]); hipFree(f.dk[s]); hipFree(f.dc[s]);
		if (f.ev_copied[s]) hipEventDestroy(f.ev_copied[s]);
		if (f.ev_free[s]) hipEventDestroy(f.ev_free[s]);
		f.raw[s] = f.draw[s] = nullptr; f.km[s] = f.dk[s] = nullptr; f.cnt[s] = f.dc[s] = nullptr;
		f.ev_copied[s] = f.ev_free[s] = nullptr;
	}
	hipFree(f.d_lut); f.d_lut = nullptr;
	if (f.h_lut) hipHostFree(f.h_lut);
	f.h_lut = nullptr;
	f.raw_cap = f.km_cap = f.dk_cap = f.lut_cap = 0;
}

static void free_range(kmx_model *m)
{
	auto &R = m->range;
	hipFree(R.d_send); hipFree(R.d_hdr); hipFree(R.rd.ccnt); hipFree(R.rd.tcnt); hipFree(R.rd.tidx); hipFree(R.rd.contended); hipFree(R.rd.n_contended);
	hipFree(R.rd.rt_key); hipFree(R.rd.rt_resv); hipFree(R.rd.rt_mark); hipFree(R.rd.rt_eidx); hipFree(R.rd.rt_um);
	hipFree(R.d_oovf); hipFree(R.d_opcnt); hipFree(R.d_lver); hipFree(R.d_inbox); hipFree(R.d_in_hdr); hipFree(R.d_vbox); hipFree(R.d_ovf);
	if (R.h_hdr) hipHostFree(R.h_hdr);
	R = kmx_model::RangeState();
}

static void free_query_feed(kmx_model *m)
{
	auto &f = m->qfeed;
	if (f.to_dev) { hipStreamSynchronize(f.to_dev); hipStreamDestroy(f.to_dev); f.to_dev = nullptr; }
	if (f.to_host) { hipStreamSynchronize(f.to_host); hipStreamDestroy(f.to_host); f.to_host = nullptr; }
	for (int s = 0; s < f.S; s++) {
		if (f.h_in[s]) hipHostFree(f.h_in[s]);
		if (f.h_out[s]) hipHostFree(f.h_out[s]);
		hipFree(f.d_in[s]); hipFree(f.d_out[s]);
		f.h_in[s] = f.d_in[s] = nullptr; f.h_out[s] = f.d_out[s] = nullptr;
		for (hipEvent_t *e : {&f.ev_in[s], &f.ev_k[s], &f.ev_out[s]}) { if (*e) hipEventDestroy(*e); *e = nullptr; }
	}
	f.in_cap = f.out_cap = 0;
}

static int kmx_destroy_impl(kmx_model *m)
{
	if (!m) return KMX_OK;
	hipSetDevice(m->device);
	hipStreamSynchronize(m->stream);
	free_build_state(m);
	free_arrays(m);
	hipFree(m->d_bin_of_occ); hipFree(m->d_mean_of_bin); hipFree(m->d_stats); hipFree(m->d_nbf);
	if (m->h_total) hipHostFree(m->h_total);
	if (m->h_feedback) hipHostFree(m->h_feedback);
	if (m->ev0) hipEventDestroy(m->ev0);
	if (m->ev1) hipEventDestroy(m->ev1);
	hipFree(m->d_totals);
	if (m->h_totals) hipHostFree(m->h_totals);
	free_feed(m);
	free_query_feed(m);
	free_range(m);
	if (m->probe.side) { hipStreamSynchronize(m->probe.side); hipStreamDestroy(m->probe.side); hipEventDestroy(m->probe.fork); hipEventDestroy(m->probe.done); hipFree(m->probe.sink); }
	for (hipEvent_t e : m->prof_events) hipEventDestroy(e);
	delete m;
	return KMX_OK;
}

static int kmx_set_stream_impl(kmx_model *m, void *s)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	m->stream = (hipStream_t)s;
	return KMX_OK;
}

// sizes (kmodel.hpp:402-420, :436-456) and the by-value kernel argument block
static void compute_sizes(kmx_model *m)
{
	u64 nbf = 0;
	for (int i = 0; i < m->bf_num; i++) {
		m->byte_bf[i] = (u64)((double)m->n_bf[i] / 5.5 * (double)(m->nh - 1));          // f64 then truncation (Q5)
		m->byte_bf_back[i] = (m->n_bf[i] >> 3) * (u64)(m->nh - 2);
		nbf += m->n_bf[i];
	}
	m->n_km = m->n_total - nbf;                                                        // kmodel.hpp:433
	m->km_byte_size = (m->n_km >> 4) * (u64)m->nh;
	m->byte_km_back = (m->n_km >> 4) * (u64)(m->nh - 2);
	m->ncells = (m->km_byte_size + 1) / 2;
}

static void fill_model_dev(kmx_model *m)
{
	ModelDev &md = m->md;
	memset(&md, 0, sizeof md);
	md.k = m->k; md.nh = m->nh; md.nb = m->nb; md.ci = m->ci; md.cs = m->cs; md.bf_num = m->bf_num;
	{
		// a failing attempt usually fails on its first few positions: fetch those first (KMX_NH_FIRST >= nh: all at once)
		// (measured at nh = 7: 2 + 2 + 3 positions, profiles/r03_experiments_measured_and_rejected.txt)
		const char *e = hook_env("KMX_NH_FIRST"), *e2 = hook_env("KMX_NH_SECOND");
		const int f = e ? atoi(e) : (2 * m->nh + 3) / 7, f2 = e2 ? atoi(e2) : (4 * m->nh + 3) / 7;
		md.nh_first = f < 1 ? 1 : (f > m->nh ? m->nh : f);
		md.nh_second = f2 < md.nh_first ? md.nh_first : (f2 > m->nh ? m->nh : f2);
	}
	md.gfull = make_geom(m->k);
	md.gback = make_geom(m->k - 2);
	for (int i = 0; i < 3; i++) {
		md.bf[i] = m->d_bf[i]; md.bf_mod[i] = make_mod(i < m->bf_num ? m->byte_bf[i] * 8 : 0);
		md.bf_back[i] = m->d_bf_back[i]; md.bf_back_mod[i] = make_mod(i < m->bf_num ? m->byte_bf_back[i] * 8 : 0);
	}
	md.km_back = m->d_km_back; md.km_back_mod = make_mod(m->byte_km_back * 8);
	md.kmb_direct = m->kmb_deferred ? 0 : 1;
	md.bloom_direct = m->blm_deferred ? 0 : 1;
	for (int i = 0; i < 3; i++) { md.bf_woff[i] = m->bf_woff[i]; md.bf_back_woff[i] = m->bf_back_woff[i]; }
	for (int a = 0; a < m->nb; a++) md.cells[a] = m->d_cells[a];
	md.km_mod = make_mod(m->km_byte_size * 8);
	md.bin_of_occ = m->d_bin_of_occ; md.mean_of_bin = m->d_mean_of_bin;
	md.rest_pre_len = m->rest.pre_len; md.rest_W = m->W; md.rest_entries = m->rest.entries;
	md.rest_h2i = m->rest.d_h2i; md.rest_pre = m->rest.d_pre; md.rest_suf = m->rest.d_suf; md.rest_cnt = m->rest.d_cnt;
	md.rest_fbits = m->rest.fbits; md.rest_fine = m->rest.d_fine; md.rest_q = m->rest.d_q;
}

static int alloc_arrays(kmx_model *m)
{
	free_rest_dev(m->rest);
	{   // the Bloom filters and their back filters live back to back in one slab (the BitScatter of the front end sweeps it)
		u64 off = 0;
		for (int i = 0; i < 3; i++) {
			m->bf_woff[i] = off; off += i < m->bf_num ? (m->byte_bf[i] + 3) / 4 + 1 : 0;
			m->bf_back_woff[i] = off; off += i < m->bf_num ? (m->byte_bf_back[i] + 3) / 4 + 1 : 0;
		}
		m->bloom_words = off;
		TRY(ensure(&m->d_bloom, &m->cap_bloom, (off + 1) * 4, true, m->stream));
		for (int i = 0; i < 3; i++) {
			m->d_bf[i] = i < m->bf_num ? m->d_bloom + m->bf_woff[i] : nullptr;
			m->d_bf_back[i] = i < m->bf_num ? m->d_bloom + m->bf_back_woff[i] : nullptr;
		}
	}
	TRY(ensure(&m->d_km_back, &m->cap_km_back, ((m->byte_km_back + 3) / 4 + 1) * 4, true, m->stream));
	// (hipMalloc of fresh device memory costs ~90 ms per GB on this stack, also from several threads at once:
	// a cold build of 2.5e9 k-mers spends 2.3 s here, a rebuild on the same handle nothing)
	const u64 need = (m->ncells + 1) * sizeof(cell_t);
	for (int a = 0; a < m->nb; a++) TRY(ensure(&m->d_cells[a], &m->cap_cells[a], need, true, m->stream));
	return KMX_OK;
}

// km_back as a partitioned bit-set: bins of a power-of-two number of positions, swept in tiles of 2^20.  Up to 8 tiles per
// bin (a filter of 256 MB: ~8*10^8 coupled k-mers at nh = 7) one workgroup per bin sweeps its tiles one after the other and
// re-reads the bin's tuples for each (k_bs_apply); doing that with 32 tiles LOSES against plain atomics (2.5*10^9 k-mers:
// warm build 2.66 s -> 4.14 s).  Bigger filters get a second level: k_bs_split deals every bin's tuples to its tiles
// (up to 256 per bin: filters of 8 GB), k_bs_apply2 sweeps one tile per workgroup, so every tuple and every word of the
// filter is read once per sweep.  Beyond 2^36 positions the direct atomic path stays.
static const u32 kBsMaxTileShift = 3;
// test hooks: KMX_BS_CAP=<tuples per bin> (read once per process) shrinks the bins of both partitioned bit-sets, so that
// producers meet full bins all the time and set those bits with atomics instead (the result must not change);
// KMX_BS_TILE_LOG2=<t> (10..20) shrinks the tiles, so that small filters take the multi-tile and the two-level sweeps
static u64 bs_cap_hook()
{
	static const u64 v = [] { const char *e = hook_env("KMX_BS_CAP"); const long long x = e ? atoll(e) : 0; return x > 0 ? (u64)std::min<long long>(x, 1 << 18) : 0ull; }();
	return v;
}
static u32 bs_tile_log2()
{
	const char *e = hook_env("KMX_BS_TILE_LOG2");                    // (read at every kmx_begin, so one test process can try several)
	const int x = e ? atoi(e) : 20;
	return (u32)std::min(std::max(x, 10), 20);
}
// geometry of a partitioned bit-set over `nwords` words: false when the filter is too big for it.  L2 storage (shared by
// the two bit-sets of a model: their sweeps never overlap) is grown to what this one needs.
static int bs_geometry(kmx_model *m, BitScatter &bs, u64 nwords, u64 cap, bool *ok)
{
	*ok = false;
	const u32 tlog2 = bs_tile_log2();
	u32 wshift = 5;
	while ((((u64)BS_BINS) << wshift) < nwords * 32) wshift++;
	if (wshift > tlog2 + BS_MAX_TILES_LOG2 || wshift > 32) return KMX_OK;
	bs.nwords = nwords; bs.wshift = wshift; bs.tlog2 = tlog2; bs.cap = (u32)cap;
	bs.cap2 = 0; bs.tup2 = nullptr; bs.cnt2 = nullptr;
	if (wshift > tlog2 + kBsMaxTileShift) {
		const u32 tiles = 1u << (wshift - tlog2);
		// uniform hashing: a tile receives cap/tiles tuples on average when the bin is full; 1/4 more + 256 never fills in
		// practice (and a full tile only costs atomics)
		const u64 cap2 = bs_cap_hook() ? std::max<u64>(bs_cap_hook() / 4, 8) : (cap / tiles + cap / (4 * tiles) + 256);
		const u64 need = (u64)BS_BINS * tiles * cap2, ncnt = (u64)BS_BINS * tiles;
		if (m->bs2_tup_cap < need || m->bs2_cnt_cap < ncnt) {
			HIPCHK(hipStreamSynchronize(m->stream));
			hipFree(m->d_bs2_tup); hipFree(m->d_bs2_cnt);
			m->d_bs2_tup = nullptr; m->d_bs2_cnt = nullptr; m->bs2_tup_cap = m->bs2_cnt_cap = 0;
			TRY(dalloc(&m->d_bs2_tup, need, false, m->stream));
			TRY(dalloc(&m->d_bs2_cnt, ncnt, true, m->stream));
			m->bs2_tup_cap = need; m->bs2_cnt_cap = ncnt;
		}
		bs.cap2 = (u32)cap2;
	}
	*ok = true;
	return KMX_OK;
}
static int setup_kmback_scatter(kmx_model *m)
{
	m->kmb_deferred = false;
	m->kmb_pending = 0;
	m->kmb_job.n_lists = 0;
	const u64 nwords = (m->byte_km_back + 3) / 4;
	if (m->dbg_kmb_direct || nwords == 0) return KMX_OK;
	u32 wshift = 5;
	while ((((u64)BS_BINS) << wshift) < nwords * 32) wshift++;
	const u64 blk = (u64)m->nb * KMX_BUCKET, per_block = blk * (u64)(m->nh - 2);
	// tuples per bin: 1 MB while a bin is a few tiles; more when the sweep re-reads the tuples tile after tile or the filter
	// is big, so that a sweep of the whole filter is shared by more blocks
	u64 cap = wshift <= 22 ? (1u << 18) : (wshift == 23 ? (1u << 19) : (1u << 20));
	// two levels: every sweep reads and writes the whole filter, so the bins grow with it (a sweep per ~nwords/3 tuples:
	// at 10^10 k-mers 2^22 per bin = 4 GB + 5 GB of second-level storage, 52 sweeps of 15.6 GB instead of 207)
	while (wshift > 23 && cap < (1u << 22) && cap * 512 < nwords) cap <<= 1;
	const u64 cap_hook = bs_cap_hook();
	if (cap_hook) cap = cap_hook;
	bool ok;
	TRY(bs_geometry(m, m->kmb, nwords, cap, &ok));
	if (!ok) return KMX_OK;                                        // too big: direct atomics
	const u64 bins_used = (nwords * 32 + (1ULL << m->kmb.wshift) - 1) >> m->kmb.wshift;
	if (!m->d_kmb_tup || m->kmb_tup_cap < (u64)BS_BINS * cap) {
		HIPCHK(hipStreamSynchronize(m->stream));
		hipFree(m->d_kmb_tup); hipFree(m->d_kmb_cnt);
		m->d_kmb_tup = nullptr; m->d_kmb_cnt = nullptr; m->kmb_tup_cap = 0;
		TRY(dalloc(&m->d_kmb_tup, (u64)BS_BINS * cap, false, m->stream));
		TRY(dalloc(&m->d_kmb_cnt, (u64)BS_BINS, true, m->stream));
		m->kmb_tup_cap = (u64)BS_BINS * cap;
	}
	m->kmb.words = m->d_km_back;
	m->kmb.tup = m->d_kmb_tup; m->kmb.cnt = m->d_kmb_cnt;
	// positions are hashed uniformly over the used bins: keep the expected fill of a bin below 3/4 (a full bin is still exact)
	m->kmb_budget = bins_used * cap * 3 / 4;
	if (m->kmb_budget < per_block && !cap_hook) return KMX_OK;     // a single block would not fit: tiny filter, direct path
	m->kmb_deferred = true;
	return KMX_OK;
}

// The Bloom slab as a partitioned bit-set: the front end emits a chunk (2^23 k-mers) and sweeps; the bins take a chunk of
// which ~2/3 is Bloom class (more falls back to atomics, bit by bit).
static int setup_bloom_scatter(kmx_model *m)
{
	m->blm_deferred = false;
	if (m->dbg_kmb_direct || m->bloom_words == 0) return KMX_OK;
	m->blm_sweep_every = 1;
	// a slab above 256 MB (second level) costs > 0.5 GB of traffic per sweep: bigger bins, so that a sweep serves more chunks
	const u64 cap = bs_cap_hook() ? bs_cap_hook() : (m->bloom_words > (1ull << 26) ? (1u << 20) : (1u << 18));
	bool ok;
	TRY(bs_geometry(m, m->blm, m->bloom_words, cap, &ok));
	if (!ok) return KMX_OK;
	if (!m->d_blm_tup || m->blm_tup_cap < (u64)BS_BINS * cap) {
		HIPCHK(hipStreamSynchronize(m->stream));
		hipFree(m->d_blm_tup); hipFree(m->d_blm_cnt);
		m->d_blm_tup = nullptr; m->d_blm_cnt = nullptr; m->blm_tup_cap = 0;
		TRY(dalloc(&m->d_blm_tup, (u64)BS_BINS * cap, false, m->stream));
		TRY(dalloc(&m->d_blm_cnt, (u64)BS_BINS, true, m->stream));
		m->blm_tup_cap = (u64)BS_BINS * cap;
	}
	{
		// chunks per sweep from the EXPECTED number of tuples of a chunk (the share of Bloom-class k-mers is known from pass 1):
		// positions are hashed uniformly over the used bins, aim at 3/4 of their capacity; a bin that fills up all the same
		// only costs atomics
		const u64 bins_used = (m->bloom_words * 32 + (1ULL << m->blm.wshift) - 1) >> m->blm.wshift;
		u64 nbf = 0;
		for (int i = 0; i < m->bf_num; i++) nbf += m->n_bf[i];
		const double per_chunk = (double)kChunk * ((double)nbf / (double)std::max<u64>(m->n_total, 1)) * (double)(2 * m->nh - 3) * 1.25;
		const double budget = (double)bins_used * (double)cap * 0.75;
		const double e = per_chunk > 0 ? budget / per_chunk : 1.0;
		m->blm_sweep_every = (int)std::min(std::max(e, 1.0), 64.0);
	}
	m->blm.words = m->d_bloom;
	m->blm.tup = m->d_blm_tup; m->blm.cnt = m->d_blm_cnt;
	m->blm_deferred = true;
	return KMX_OK;
}
// (the two bit-sets share the second-level storage: point both at it once both geometries are known)
static void bs_attach_level2(kmx_model *m)
{
	m->kmb.tup2 = m->blm.tup2 = m->d_bs2_tup;
	m->kmb.cnt2 = m->blm.cnt2 = m->d_bs2_cnt;
}

// sweep km_back with what the rounds have emitted so far
static int kmback_flush(kmx_model *m)
{
	if (!m->kmb_deferred || !m->kmb_pending) return KMX_OK;
	kmxk::bs_apply(m->kmb, m->stream);
	m->kmb_pending = 0;
	HIPCHK(hipGetLastError());
	return KMX_OK;
}
// the (k-2)-mers of the successes -> bins: of round t (n_in_block < 0; the ring, where a rank holds a list for one round)
// or of a whole block after its last round (n_in_block >= 0); `bound` = most k-mers that can have been inserted
static int kmback_reserve(kmx_model *m, u64 bound)
{
	const u64 add = bound * (u64)(m->nh - 2);
	if (m->kmb_pending + add > m->kmb_budget) TRY(kmback_flush(m));
	m->kmb_pending += add;
	return KMX_OK;
}
static int kmback_emit(kmx_model *m, int t, int pp, int n_in_block, u64 bound)
{
	if (m->kmb_deferred) TRY(kmback_reserve(m, bound));          // (else: md.kmb_direct, the kernel ORs the bits in itself)
	kmxk::kmback_emit(m->md, m->bd, m->bd.kmers, m->bd.surv, 0, m->nb, t, pp, n_in_block, m->kmb, m->stream);
	return KMX_OK;
}
// what is left of the job a finished block handed on: emitted in a launch of its own (the k-mers it points at are about to
// be overwritten, the build ends, or no late round can host it)
static int kmback_job_flush(kmx_model *m)
{
	KmbackJob &j = m->kmb_job;
	if (j.n_lists <= 0) return KMX_OK;
	const u64 lo = (u64)j.i0 * KMX_BUCKET, nbk = (u64)j.n_in_block;
	TRY(kmback_reserve(m, nbk > lo ? std::min<u64>(nbk - lo, (u64)j.n_lists * KMX_BUCKET) : 0));
	kmxk::kmback_emit(m->md, m->bd, j.kmers, j.surv, j.i0, j.n_lists, 0, 0, j.n_in_block, m->kmb, m->stream);
	j.n_lists = 0;
	return KMX_OK;
}

// ------------------------------------------------------------------------------------------ streamed build
static int kmx_begin_impl(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (k < 3 || k > 64) return fail(KMX_E_ARG, "k=%d out of range [3,64]", k);
	HIPCHK(hipSetDevice(m->device));
	u64 s = 0;
	for (int i = 0; i < m->bf_num; i++) s += n_bf[i];
	if (s > n_total) return fail(KMX_E_ARG, "n_bf exceeds n_total");
	m->k = k; m->W = (k + 31) / 32;
	m->n_total = n_total;
	for (int i = 0; i < 3; i++) m->n_bf[i] = i < m->bf_num ? n_bf[i] : 0;
	compute_sizes(m);
	TRY(alloc_arrays(m));
	m->rest = RestTable();
	TRY(setup_kmback_scatter(m));
	TRY(setup_bloom_scatter(m));
	bs_attach_level2(m);
	fill_model_dev(m);
	const int nb = m->nb;
	const u64 B = KMX_BUCKET, blk = (u64)nb * B;
	// Build-time buffers are kept across builds of the same shape (nb, W): only the counters are re-zeroed.
	if (m->d_block_scratch && (m->scratch_nb != nb || m->scratch_W != m->W)) free_build_state(m);
	if (!m->d_block_scratch) {
		// staging stream for coupled-array k-mers: one front-end chunk plus one block of carry-over
		m->stg_cap = kChunk + blk;
		TRY(dalloc(&m->d_stg_kmers, m->stg_cap * m->W, false, m->stream));
		TRY(dalloc(&m->d_stg_counts, m->stg_cap, false, m->stream));
		// one slab for the block working set
		u64 off = 0;
		auto carve = [&](u64 bytes) { u64 o = off; off += (bytes + 255) & ~u64(255); return o; };
		u64 o_list0 = carve(blk * 4), o_list1 = carve(blk * 4), o_mv0 = carve(blk * 4), o_mv1 = carve(blk * 4);
		u64 o_n0 = carve(nb * 4), o_n1 = carve(nb * 4), o_status0 = carve(blk), o_status1 = carve(blk), o_dfail = carve(blk);
		u64 o_U[KMX_NSLOW];
		for (int s2 = 0; s2 < KMX_NSLOW; s2++) o_U[s2] = carve(blk * 8 * (1 + m->W));
		u64 o_Un = carve((u64)KMX_NSLOW * nb * KMX_CTR_STRIDE * 4), o_R = carve((u64)nb * KMX_RSIZE * 8);
		u64 o_tc0 = carve((u64)nb * KMX_NTILES * 4), o_tc1 = carve((u64)nb * KMX_NTILES * 4);
		const u64 cl_bins = m->nh <= 8 ? KMX_CL_BINS(8) : KMX_CL_BINS(16);
		u64 o_surv = carve(blk), o_surv1 = carve(blk);
		const u64 cl_bytes = (u64)nb * cl_bins * (u64)(m->nh <= 8 ? KMX_CL_CAP_OF(8) : KMX_CL_CAP_OF(16)) * 8;
		u64 o_uw[2], o_crec[2], o_cl_tup[2], o_cl_cnt[2];
		const u64 rec_words = (u64)((m->nh + (m->nh <= 8 ? 1 : 2) + 3) & ~3);      // kernels.hip crec_words
		for (int q = 0; q < 2; q++) {
			o_uw[q] = carve(blk * 4); o_crec[q] = carve(blk * 4 * rec_words);
			o_cl_tup[q] = carve(cl_bytes); o_cl_cnt[q] = carve((u64)nb * KMX_CL_MAXBINS * 4);
		}
		u64 o_cl_ovf = carve((u64)nb * 4);
		HIPCHK(hipMalloc(&m->d_block_scratch, off));
		HIPCHK(hipMemsetAsync(m->d_block_scratch, 0, off, m->stream));     // R starts at epoch 0; epochs only grow
		m->scratch_bytes = off; m->scratch_nb = nb; m->scratch_W = m->W;
		char *base = (char *)m->d_block_scratch;
		BlockDev &bd = m->bd;
		bd.kmers = nullptr; bd.counts = nullptr;
		bd.list[0] = (u32 *)(base + o_list0); bd.list[1] = (u32 *)(base + o_list1);
		bd.mover[0] = (u32 *)(base + o_mv0); bd.mover[1] = (u32 *)(base + o_mv1);
		bd.n[0] = (int *)(base + o_n0); bd.n[1] = (int *)(base + o_n1);
		bd.status[0] = (unsigned char *)(base + o_status0); bd.status[1] = (unsigned char *)(base + o_status1); bd.dfail = (unsigned char *)(base + o_dfail);
		for (int s2 = 0; s2 < KMX_NSLOW; s2++) bd.Urec[s2] = (u64 *)(base + o_U[s2]);
		bd.Un = (int *)(base + o_Un); bd.R = (u64 *)(base + o_R);
		bd.tile_cnt[0] = (int *)(base + o_tc0); bd.tile_cnt[1] = (int *)(base + o_tc1);
		bd.surv = (unsigned char *)(base + o_surv);
		m->d_surv[0] = bd.surv; m->d_surv[1] = (unsigned char *)(base + o_surv1);   // block b flags into d_surv[b & 1]
		for (int q = 0; q < 2; q++) {
			bd.uw[q] = (u32 *)(base + o_uw[q]); bd.crec[q] = (u32 *)(base + o_crec[q]);
			bd.cl_tup[q] = (u64 *)(base + o_cl_tup[q]); bd.cl_cnt[q] = (int *)(base + o_cl_cnt[q]);
		}
		bd.cl_ovf = (int *)(base + o_cl_ovf);                        // zeroed with the slab; the kernels keep it zero between rounds
		bd.stats = m->d_stats;
		TRY(dalloc(&m->d_rest_n, 1, false, m->stream));
		TRY(dalloc(&m->d_stale_kmers, (u64)nb * 2, false, m->stream));
		TRY(dalloc(&m->d_stale_counts, (u64)nb, false, m->stream));
		TRY(dalloc(&m->d_total, 4, true, m->stream));
		m->rest_cap = 0;
	}
	m->stg_n = 0;
	// a build that ended on an error may have left claim counters or dfail marks behind
	for (int q = 0; q < 2; q++) HIPCHK(hipMemsetAsync(m->bd.cl_cnt[q], 0, (u64)nb * KMX_CL_MAXBINS * 4, m->stream));
	HIPCHK(hipMemsetAsync(m->bd.dfail, 0, blk, m->stream));
	HIPCHK(hipMemsetAsync(m->bd.cl_ovf, 0, (u64)nb * 4, m->stream));
	m->pp = 0; m->pending = false; m->pending_t = 0;
	m->defer = !m->dbg_no_defer && m->km_byte_size * 8 <= (1ULL << KMX_CL_MIX_BITS);      // position identity inside a detect table is exact up to 2^36
	// rest accumulators: grown on demand (see ensure_rest_capacity)
	const u64 want_rest = std::max<u64>(m->n_km / 8, 2 * blk) + blk;
	if (m->rest_cap < want_rest) {
		hipFree(m->d_rest_kmers); hipFree(m->d_rest_counts);
		m->d_rest_kmers = nullptr; m->d_rest_counts = nullptr;
		TRY(dalloc(&m->d_rest_kmers, want_rest * m->W, false, m->stream));
		TRY(dalloc(&m->d_rest_counts, want_rest, false, m->stream));
		m->rest_cap = want_rest;
	}
	m->rest_upper = 0;
	HIPCHK(hipMemsetAsync(m->d_rest_n, 0, 8, m->stream));
	HIPCHK(hipMemsetAsync(m->d_stale_kmers, 0, (u64)nb * 16, m->stream));
	HIPCHK(hipMemsetAsync(m->d_stale_counts, 0, (u64)nb * 4, m->stream));
	HIPCHK(hipMemsetAsync(m->d_stats, 0, ST_N * 8, m->stream));
	m->blocks = 0; m->rounds = 0;
	// [2]: the fullest claim bin of a late round (t >= 2), which picks the form of their k_round_detect: until the device reports
	// one, a guess -- a quarter of a list still alive, every survivor a candidate on all its positions -- so that wide
	// configurations (nh >= 9: 2304 > 2048) start on the full-size tables instead of overflowing the small ones
	m->h_feedback[0] = ~0ULL; m->h_feedback[1] = 0; m->h_feedback[2] = (u64)KMX_BUCKET * (u64)m->nh / KMX_CL_MAXBINS / 4;
	{   // first guess of the contended set per list in round 0: a candidate is contended when one of its nh positions
		// is also claimed with the other value by one of the ~2^18*nh/2 opposite claims spread over the L positions
		const double L = (double)m->km_byte_size * 8.0;
		const double p = L > 0 ? std::min(1.0, (double)m->nh * m->nh * 131072.0 / L) : 1.0;
		const double u0 = 157000.0 * p;
		const double fin = 1024.0 * KMX_FIN_RPT(m->nh <= 8 ? 8 : 16);       // what the LDS finisher holds per list
		// measured at 10^8 k-mers: one load of the finisher 50-60 us; two loads (index ranges) 115 us; one grid-wide pass
		// (37 us) decides half of the set and leaves one load (40 us)
		m->nsub = u0 <= fin ? 0 : (u0 <= 6 * fin ? 1 : (u0 <= 16 * fin ? 3 : (u0 <= 40 * fin ? 5 : 8)));
	}
	m->t_insert_kernels = 0; m->t_total = 0;
	memset(m->h_stats, 0, sizeof m->h_stats);
	{
		const char *pe = hook_env("KMX_PREGATHER_PROBE");
		const bool want = pe && pe[0] == '1';
		if (want && !m->probe.side) {
			HIPCHK(hipStreamCreateWithFlags(&m->probe.side, hipStreamNonBlocking));
			HIPCHK(hipEventCreateWithFlags(&m->probe.fork, hipEventDisableTiming));
			HIPCHK(hipEventCreateWithFlags(&m->probe.done, hipEventDisableTiming));
			HIPCHK(hipMalloc((void **)&m->probe.sink, 256));
		}
		if (want) HIPCHK(hipEventRecord(m->probe.done, m->probe.side));
		m->probe.on = want;
	}
	m->ring = false; m->ring_rank = 0; m->ring_world = 1;
	m->state = ST_BUILDING;
	return KMX_OK;
}

static int ensure_rest_capacity(kmx_model *m, u64 add)
{
	if (m->rest_upper + add <= m->rest_cap) { m->rest_upper += add; return KMX_OK; }
	unsigned long long actual = 0;
	HIPCHK(hipMemcpyAsync(&actual, m->d_rest_n, 8, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	m->rest_upper = actual;
	if (m->rest_upper + add > m->rest_cap) {
		u64 ncap = std::max<u64>(m->rest_cap * 2, m->rest_upper + add);
		DevMem nk, nc;
		HIPCHK(nk.alloc(ncap * m->W * 8));
		HIPCHK(nc.alloc(ncap * 4));
		HIPCHK(hipMemcpyAsync(nk.p, m->d_rest_kmers, actual * m->W * 8, hipMemcpyDeviceToDevice, m->stream));
		HIPCHK(hipMemcpyAsync(nc.p, m->d_rest_counts, actual * 4, hipMemcpyDeviceToDevice, m->stream));
		HIPCHK(hipStreamSynchronize(m->stream));
		hipFree(m->d_rest_kmers); hipFree(m->d_rest_counts);
		m->d_rest_kmers = (u64 *)nk.release(); m->d_rest_counts = (int *)nc.release(); m->rest_cap = ncap;
	}
	m->rest_upper += add;
	return KMX_OK;
}

// Stale-slot duplicate (quirk Q1, kmodel.hpp:520-527 + :539): in the final partial block every unused
// buffer row whose slot 0 still holds a survivor of the previous block re-offers it to nb-1 arrays (it
// fails on all of them: bits are never cleared) and pushes it to the rest table a second time.
__global__ void k_stale_dup(int first_unused_row, int nb, int W, const u64 *stale_kmers, const int *stale_counts, u64 *rest_kmers, int *rest_counts, unsigned long long *rest_n, u64 *stats)
{
	int i = first_unused_row + threadIdx.x;
	if (i >= nb || stale_counts[i] == 0) return;
	u64 p = atomicAdd(rest_n, 1ULL);
	for (int w = 0; w < W; w++) rest_kmers[p * W + w] = stale_kmers[(u64)i * W + w];
	rest_counts[p] = stale_counts[i];
	atomicAdd(stats + ST_ATTEMPTS, (u64)(nb - 1));
}

// Contention feedback (pinned words that k_rest_append writes, read here while the device is some blocks behind; they
// steer a launch-count heuristic only, never the result): the largest contended set per list, and the largest set that
// reached the single-workgroup finisher.  Small sets are decided by the finisher alone; when too much reaches it,
// grid-wide ordered passes are added in front of it.
static void steer_passes(kmx_model *m)
{
	const u64 u0 = ((volatile u64 *)m->h_feedback)[0], ufin = ((volatile u64 *)m->h_feedback)[1];
	if (u0 != ~0ULL) {
		const u64 fin = 1024ull * KMX_FIN_RPT(m->nh <= 8 ? 8 : 16);
		if (ufin > fin) m->nsub = std::min(m->nsub + (ufin > 4 * fin ? 2 : 1), KMX_MAX_NSUB);
		else if (m->nsub > 0 && u0 <= fin * 7 / 8) m->nsub--;               // the finisher could have taken all of it in one load
		else if (m->nsub > 1 && ufin < fin / 4) m->nsub--;
	}
	if (m->dbg_ctrl) fprintf(stderr, "[kmx] block %llu feedback u0=%lld ufin=%lld -> nsub=%d\n", (unsigned long long)m->blocks, (long long)u0, (long long)ufin, m->nsub);
}
static int passes_of_round(const kmx_model *m, int t)
{
	int nsub = t == 0 ? m->nsub : m->nsub / 2;
	if (t == 0 && m->dbg_nsub0 >= 0) nsub = m->dbg_nsub0;
	if (t > 0 && m->dbg_nsub1 >= 0) nsub = m->dbg_nsub1;
	return nsub;
}

// insert_with_thread (kmodel.hpp:557-573) for the block at staging offset `head`
// the commit the last round still owes (see kmx_model::pending), in a launch of its own
static int flush_pending_commit(kmx_model *m)
{
	if (!m->pending) return KMX_OK;
	kmxk::commit_flush(m->md, m->bd, m->pending_t, m->pp ^ 1, m->stream, &m->prof);
	m->pending = false;
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

// one round of the block in m->bd (list parity m->pp): the previous round's commit rides with its check; its own winners
// stay pending -- or are committed right away when the build does not defer (m->defer false: arrays beyond 2^36 positions,
// KMX_PIPE=0).  A rank of the multi-GPU ring defers too (kmx_ring_round_dev passes m->defer): it owns its arrays whole, so
// the claims its detect needs as settled positions are its own.  Deferring is sound only while cl_mix is a bijection on the
// positions of an array -- the gate in kmx_begin and the static_assert below tie the two together.
static_assert(KMX_CL_MIX_BITS == 36, "kmx_begin's defer gate, cl_mix's mask and the 8 + 28-bit table entry of k_round_detect all assume 36 bits");
static int run_round(kmx_model *m, int t, bool defer, const KmbackJob *job)
{
	// late rounds: small detect tables while the fullest late bin the device reported stays far below what they take (a
	// launch-shape heuristic like steer_passes: a bin that does not fit only sends its list down the ordered path, never changes the result)
	const u64 late_bin = ((volatile u64 *)m->h_feedback)[2];
	const bool small_detect = m->dbg_small_detect >= 0 ? m->dbg_small_detect != 0 : late_bin <= 2048;
	const int flags = m->dbg_flags | (m->pending ? KMX_ROUND_PENDING : 0) | (defer ? KMX_ROUND_KEEP : 0) | (small_detect ? KMX_ROUND_SMALL_DETECT : 0);
	kmxk::round(m->md, m->bd, t, m->pp, passes_of_round(m, t), &m->epoch, flags, m->stream, &m->prof, job, &m->kmb, m->probe.on ? &m->probe : nullptr);
	m->pending = true; m->pending_t = t;
	m->pp ^= 1;
	m->rounds++;
	if (!defer) TRY(flush_pending_commit(m));
	return KMX_OK;
}

// insert_with_thread (kmodel.hpp:557-573) for the block at staging offset `head`
static int process_block(kmx_model *m, u64 head, u64 n_in_block, bool final_partial)
{
	const int nb = m->nb;
	m->bd.kmers = m->d_stg_kmers + head * m->W;
	m->bd.counts = m->d_stg_counts + head;
	m->bd.surv = m->d_surv[m->blocks & 1];                        // (the previous block's flags stay readable for its job)
	kmxk::block_init(m->bd, nb, m->pp, (int)n_in_block, m->stream);   // (the pending commit of the previous block reads the OTHER parity)
	steer_passes(m);
	// The previous block's km_back emission rides along with this block's finisher launches, one list per round: 128
	// rider workgroups beside the nb finisher workgroups, one wave of workgroups on 256 CUs.
	KmbackJob &job = m->kmb_job;
	for (int t = 0; t < nb; t++) {
		KmbackJob part = {nullptr, nullptr, 0, 0, 0};
		if (job.n_lists > 0) {
			part = job;
			part.n_lists = 1;
			const u64 lo = (u64)part.i0 * KMX_BUCKET, nbk = (u64)part.n_in_block;
			TRY(kmback_reserve(m, nbk > lo ? std::min<u64>(nbk - lo, (u64)part.n_lists * KMX_BUCKET) : 0));
			job.i0 += part.n_lists;
			job.n_lists -= part.n_lists;
		}
		TRY(run_round(m, t, m->defer, part.n_lists ? &part : nullptr));
	}
	const int pp = m->pp;                                          // the lists the last reorder wrote: the block's survivors
	TRY(ensure_rest_capacity(m, n_in_block + (u64)nb));
	if (final_partial) {
		int row = (int)((n_in_block - 1) / KMX_BUCKET);
		if (row + 1 < nb && m->blocks > 0)
			hipLaunchKernelGGL(k_stale_dup, dim3(1), dim3(64), 0, m->stream, row + 1, nb, m->W, (const u64 *)m->d_stale_kmers,
			                   (const int *)m->d_stale_counts, m->d_rest_kmers, m->d_rest_counts, m->d_rest_n, m->d_stats);
	}
	kmxk::rest_append(m->md, m->bd, pp, 0, nb, m->d_rest_kmers, m->d_rest_counts, m->d_rest_n, m->d_stale_kmers, m->d_stale_counts, m->d_feedback, m->stream);
	// km_back insert of everything the block inserted (kmodel.hpp:548-550): handed to the next block's finisher launches,
	// or done here when there is no next block to host it (or the filter takes the direct path)
	if (m->kmb_deferred && m->dbg_kmb_host && !final_partial) {
		job.kmers = m->bd.kmers; job.surv = m->bd.surv; job.n_in_block = (int)n_in_block; job.i0 = 0; job.n_lists = nb;
	} else TRY(kmback_emit(m, 0, pp, (int)n_in_block, n_in_block));
	m->blocks++;
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

// per-tile counters / per-chunk totals of the classification front end for a batch of n k-mers
static int ensure_front_end(kmx_model *m, u64 n)
{
	const u64 n_chunks = (n + kChunk - 1) / kChunk, tiles = (u64)kmxk::classify_tiles(n);
	if (tiles > m->tile_cap) {
		HIPCHK(hipStreamSynchronize(m->stream));
		hipFree(m->d_tile_cnt); hipFree(m->d_tile_off);
		m->d_tile_cnt = m->d_tile_off = nullptr;
		m->tile_cap = 0;
		TRY(dalloc(&m->d_tile_cnt, tiles, false, m->stream));
		TRY(dalloc(&m->d_tile_off, tiles, false, m->stream));
		m->tile_cap = tiles;
	}
	if (n_chunks > m->totals_cap) {
		HIPCHK(hipStreamSynchronize(m->stream));
		hipFree(m->d_totals);
		if (m->h_totals) hipHostFree(m->h_totals);
		m->d_totals = nullptr; m->h_totals = nullptr;
		m->totals_cap = 0;
		TRY(dalloc(&m->d_totals, n_chunks, false, m->stream));
		HIPCHK(hipHostMalloc((void **)&m->h_totals, n_chunks * 4));
		m->totals_cap = n_chunks;
	}
	return KMX_OK;
}

// Pass 2 for one batch (kmodel.hpp:68-74).  The batch is cut into chunks of kChunk k-mers.  The front end of chunk 0
// (classification + Bloom insert, commutative) runs on the model's stream; the front end of all later chunks runs as
// ONE launch on a side stream, underneath the ordered coupled-array rounds of the earlier chunks, which leave the
// memory system idle in their latency-bound tails.  The compaction into the staging stream and the rounds stay in
// order on the model's stream.
static int kmx_insert_batch_dev_impl(kmx_model *m, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_BUILDING) return fail(KMX_E_STATE, "insert_batch before begin");
	if (!n) return KMX_OK;
	HIPCHK(hipSetDevice(m->device));
	const u64 blk = (u64)m->nb * KMX_BUCKET;
	const u64 n_chunks = (n + kChunk - 1) / kChunk;
	TRY(ensure_front_end(m, n));
	const u64 *km0 = (const u64 *)d_kmers;
	const u32 *ct0 = (const u32 *)d_counts;
	// the order-free front end of the whole batch first (classification, Bloom classes), then the ordered rounds chunk by chunk
	kmxk::classify_count(m->md, km0, ct0, n, kChunk, m->d_tile_cnt, m->d_tile_off, m->d_totals, m->d_stats, m->blm, m->blm_sweep_every, m->stream, &m->prof);
	HIPCHK(hipMemcpyAsync(m->h_totals, m->d_totals, n_chunks * 4, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	for (u64 ci = 0, done = 0; ci < n_chunks; ci++) {
		const u64 c = std::min<u64>(kChunk, n - done);
		const u64 *km = km0 + done * m->W;
		const u32 *ct = ct0 + done;
		const u64 add = (u64)m->h_totals[ci];
		done += c;
		if (m->km_byte_size == 0 && add) continue;               // divergence D2: no arrays to insert into
		if (m->stg_n + add > m->stg_cap) return fail(KMX_E_STATE, "staging overflow");
		kmxk::classify_scatter(m->md, km, ct, c, m->d_tile_off + ci * (kChunk / KMX_CLS_TILE), m->d_stg_kmers, m->d_stg_counts, m->stg_n, m->stream);
		m->stg_n += add;
		u64 head = 0;
		while (m->stg_n - head >= blk) {
			TRY(process_block(m, head, blk, false));
			head += blk;
		}
		TRY(kmback_job_flush(m));                                // (the next scatter overwrites the k-mers the job points at)
		if (head) {                                              // carry the remainder (< one block) to the front
			const u64 rem = m->stg_n - head;
			if (rem) {
				HIPCHK(hipMemcpyAsync(m->d_stg_kmers, m->d_stg_kmers + head * m->W, rem * m->W * 8, hipMemcpyDeviceToDevice, m->stream));
				HIPCHK(hipMemcpyAsync(m->d_stg_counts, m->d_stg_counts + head, rem * 4, hipMemcpyDeviceToDevice, m->stream));
			}
			m->stg_n = rem;
		}
	}
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

// Host buffers reach the device the way the KMC feed does: two pinned slots filled by a parallel memcpy, a copy stream
// that moves slot b+1 (hipMemcpyAsync) while the model's stream inserts batch b.  The buffers live on the handle.
static int ensure_host_feed(kmx_model *m, size_t B, int W)
{
	auto &F = m->feed;
	if (!F.copy) HIPCHK(hipStreamCreateWithFlags(&F.copy, hipStreamNonBlocking));
	for (int s = 0; s < 2; s++) {
		if (!F.ev_copied[s]) { HIPCHK(hipEventCreateWithFlags(&F.ev_copied[s], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&F.ev_free[s], hipEventDisableTiming)); }
		if (F.km_cap < B * (size_t)W) {
			if (F.km[s]) hipHostFree(F.km[s]);
			if (F.cnt[s]) hipHostFree(F.cnt[s]);
			F.km[s] = nullptr; F.cnt[s] = nullptr;
			HIPCHK(hipHostMalloc((void **)&F.km[s], B * W * 8));
			HIPCHK(hipHostMalloc((void **)&F.cnt[s], B * 4));
		}
		if (F.dk_cap < B * (size_t)W) {
			hipFree(F.dk[s]); hipFree(F.dc[s]);
			F.dk[s] = nullptr; F.dc[s] = nullptr;
			HIPCHK(hipMalloc((void **)&F.dk[s], B * W * 8));
			HIPCHK(hipMalloc((void **)&F.dc[s], B * 4));
		}
	}
	F.km_cap = std::max(F.km_cap, B * (size_t)W);
	F.dk_cap = std::max(F.dk_cap, B * (size_t)W);
	return KMX_OK;
}

static void parallel_copy(void *dst, const void *src, size_t bytes)
{
	const unsigned hw = std::thread::hardware_concurrency();
	const int T = (int)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 16), bytes / (4u << 20) + 1));
	if (T == 1) { memcpy(dst, src, bytes); return; }
	std::vector<std::thread> th;
	const size_t per = ((bytes + T - 1) / T + 4095) & ~size_t(4095);
	for (int t = 0; t < T; t++)
		th.emplace_back([=] {
			const size_t lo = (size_t)t * per, hi = std::min(bytes, lo + per);
			if (lo < hi) memcpy((char *)dst + lo, (const char *)src + lo, hi - lo);
		});
	for (auto &x : th) x.join();
}

static int kmx_insert_batch_impl(kmx_model *m, const uint64_t *kmers, const uint32_t *counts, uint64_t n)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_BUILDING) return fail(KMX_E_STATE, "insert_batch before begin");
	if (!n) return KMX_OK;
	if (!kmers || !counts) return fail(KMX_E_ARG, "null argument");
	HIPCHK(hipSetDevice(m->device));
	const size_t B = size_t(1) << 23;
	const int W = m->W;
	TRY(ensure_host_feed(m, B, W));
	auto &F = m->feed;
	auto drain = scope_exit([&] { hipStreamSynchronize(F.copy); hipStreamSynchronize(m->stream); });   // the caller's buffers and the slots are free on return
	HIPCHK(hipEventRecord(F.ev_free[0], m->stream));
	HIPCHK(hipEventRecord(F.ev_free[1], m->stream));
	auto stage = [&](int s, uint64_t lo, uint64_t c) -> int {        // host -> pinned slot s -> device buffers s, on the copy stream
		HIPCHK(hipEventSynchronize(F.ev_copied[s]));                 // the previous copy out of this slot (a fresh event is complete)
		parallel_copy(F.km[s], kmers + lo * W, c * W * 8);
		parallel_copy(F.cnt[s], counts + lo, c * 4);
		HIPCHK(hipStreamWaitEvent(F.copy, F.ev_free[s], 0));        // the insert that read the device buffers two batches ago
		HIPCHK(hipMemcpyAsync(F.dk[s], F.km[s], c * W * 8, hipMemcpyHostToDevice, F.copy));
		HIPCHK(hipMemcpyAsync(F.dc[s], F.cnt[s], c * 4, hipMemcpyHostToDevice, F.copy));
		HIPCHK(hipEventRecord(F.ev_copied[s], F.copy));
		return KMX_OK;
	};
	TRY(stage(0, 0, std::min<uint64_t>(B, n)));
	int s = 0;
	for (uint64_t done = 0; done < n; s ^= 1) {
		const uint64_t c = std::min<uint64_t>(B, n - done), next = done + c;
		if (next < n) TRY(stage(s ^ 1, next, std::min<uint64_t>(B, n - next)));      // copied under this batch's rounds
		HIPCHK(hipStreamWaitEvent(m->stream, F.ev_copied[s], 0));
		TRY(kmx_insert_batch_dev(m, (const uint64_t *)F.dk[s], F.dc[s], c));
		HIPCHK(hipEventRecord(F.ev_free[s], m->stream));
		done = next;
	}
	return KMX_OK;
}

// KRestData::build (rest.hpp:95-135,157-161) on the device: radix sort of the survivors + index kernels
// (rest_device.hip).  The on-disk byte arrays are produced lazily by rest_materialize_host (save only).
static int rest_to_device(kmx_model *m);
static int rest_build_accel(kmx_model *m);

static int build_rest(kmx_model *m, u64 n)
{
	RestTable &r = m->rest;
	free_rest_dev(r);
	const int W = m->W, k = m->k;
	r.k = k;
	r.pre_len = rest_prefix_len(k);
	r.map_size = 1 << (2 * r.pre_len);
	r.suff_group = (k - r.pre_len) / 4;
	r.entries = n;
	r.suff_bin_size = n * (u64)r.suff_group;
	r.host_valid = false;
	HIPCHK(hipMalloc((void **)&r.d_sorted, n * W * 8 + 16));
	HIPCHK(hipMalloc((void **)&r.d_cnt, n * 4 + 16));
	HIPCHK(hipMalloc((void **)&r.d_suf, n * W * 8 + 16));
	HIPCHK(hipMalloc((void **)&r.d_h2i, (u64)r.map_size * 4));
	HIPCHK(hipMalloc((void **)&r.d_pre, ((u64)r.map_size + 2) * 4));
	HIPCHK(hipMemsetAsync(r.d_h2i, 0xFF, (u64)r.map_size * 4, m->stream));
	KPROF_BEGIN(&m->prof, KC_REST, m->stream);                        // class 5: the rest table -- radix sort + index kernels (the accelerators follow)
	HIPCHK(kmxk::rest_sort(m->d_rest_kmers, m->d_rest_counts, n, W, k, r.d_sorted, r.d_cnt, m->stream));
	HIPCHK(kmxk::rest_index(r.d_sorted, n, W, k, r.pre_len, r.d_h2i, r.d_pre, r.d_suf, m->d_total, m->stream));
	KPROF_END(&m->prof, m->stream);
	HIPCHK(hipMemcpyAsync(m->h_total, m->d_total, 4, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	r.pre_buffer_size = *m->h_total + 1;
	return rest_build_accel(m);
}

// bucket index + next-group table for k_query's lookup (device only)
static int rest_build_accel(kmx_model *m)
{
	RestTable &r = m->rest;
	int F = 1;
	while ((1ULL << F) < r.entries * 2 && F < 26) F++;
	F = std::max(F, 2 * r.pre_len);
	F = std::min(F, 2 * r.k);
	r.fbits = F;
	HIPCHK(hipMalloc((void **)&r.d_fine, ((1ULL << F) + 2) * 4));
	HIPCHK(hipMalloc((void **)&r.d_q, (u64)r.map_size * m->W * 8));
	kmxk::rest_accel(r.d_sorted, r.entries, m->W, r.k, F, r.d_h2i, r.d_pre, r.d_suf, r.map_size, r.d_fine, r.d_q, m->stream);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

// on-disk arrays of rest.bin (Appendix B.2) from the device-resident sorted table
static int rest_materialize_host(kmx_model *m)
{
	RestTable &r = m->rest;
	if (r.host_valid) return KMX_OK;
	const u64 n = r.entries;
	r.hash2index.resize(r.map_size); r.pre_buffer.resize(r.pre_buffer_size); r.count_bin.resize(n);
	r.suffix_bin.assign(r.suff_bin_size, 0);
	HIPCHK(hipMemcpy(r.hash2index.data(), r.d_h2i, (u64)r.map_size * 4, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(r.pre_buffer.data(), r.d_pre, (u64)r.pre_buffer_size * 4, hipMemcpyDeviceToHost));
	if (n) {
		unsigned char *d_bytes = nullptr;                           // the byte rows are cut on the device
		HIPCHK(hipMalloc((void **)&d_bytes, r.suff_bin_size));
		kmxk::rest_suffix_bytes(r.d_sorted, n, m->W, r.suff_group, d_bytes, m->stream);
		hipError_t e = hipMemcpyAsync(r.suffix_bin.data(), d_bytes, r.suff_bin_size, hipMemcpyDeviceToHost, m->stream);
		if (e == hipSuccess) e = hipMemcpyAsync(r.count_bin.data(), r.d_cnt, n * 4, hipMemcpyDeviceToHost, m->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
		hipFree(d_bytes);
		if (e != hipSuccess) return fail(KMX_E_NODEVICE, "rest table download failed");
	}
	r.host_valid = true;
	return KMX_OK;
}

// load path: device form (suffix integers, W words per row) from the byte rows read from rest.bin
static int rest_to_device(kmx_model *m)
{
	RestTable &r = m->rest;
	const int W = m->W;
	const u64 n = r.entries;
	std::vector<u64> suf(n * W + 1, 0);
	for (u64 e = 0; e < n; e++) {
		unsigned __int128 val = 0;
		for (int g = 0; g < r.suff_group; g++) val = (val << 8) | r.suffix_bin[e * (u64)r.suff_group + g];
		if (W == 1) suf[e] = (u64)val;
		else { suf[2 * e] = (u64)(val >> 64); suf[2 * e + 1] = (u64)val; }
	}
	HIPCHK(hipMalloc((void **)&r.d_h2i, (u64)r.map_size * 4));
	HIPCHK(hipMalloc((void **)&r.d_pre, (u64)r.pre_buffer_size * 4 + 4));
	HIPCHK(hipMalloc((void **)&r.d_cnt, n * 4 + 4));
	HIPCHK(hipMalloc((void **)&r.d_suf, suf.size() * 8));
	HIPCHK(hipMemcpy(r.d_h2i, r.hash2index.data(), (u64)r.map_size * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(r.d_pre, r.pre_buffer.data(), (u64)r.pre_buffer_size * 4, hipMemcpyHostToDevice));
	if (n) HIPCHK(hipMemcpy(r.d_cnt, r.count_bin.data(), n * 4, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(r.d_suf, suf.data(), suf.size() * 8, hipMemcpyHostToDevice));
	HIPCHK(hipMalloc((void **)&r.d_sorted, n * W * 8 + 16));
	kmxk::rest_expand(r.d_h2i, r.d_pre, r.d_suf, r.map_size, W, r.k, r.pre_len, r.d_sorted, m->stream);
	r.host_valid = true;
	return rest_build_accel(m);
}

static int kmx_finish_impl(kmx_model *m)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_BUILDING) return fail(KMX_E_STATE, "finish before begin");
	HIPCHK(hipSetDevice(m->device));
	if (m->stg_n) TRY(process_block(m, 0, m->stg_n, true));   // push_last_to_array; an empty tail is skipped (divergence D1)
	m->stg_n = 0;
	TRY(flush_pending_commit(m));
	TRY(kmback_flush(m));
	unsigned long long n_rest = 0;
	int range_ovf = 0;
	HIPCHK(hipMemcpyAsync(m->h_stats, m->d_stats, ST_N * 8, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipMemcpyAsync(&n_rest, m->d_rest_n, 8, hipMemcpyDeviceToHost, m->stream));
	if (m->range.on && m->range.d_ovf) HIPCHK(hipMemcpyAsync(&range_ovf, m->range.d_ovf, sizeof(int), hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	if (m->h_stats[ST_BAD_COUNT]) {
		m->state = ST_EMPTY;
		return fail(KMX_E_RANGE, "%llu k-mers with a count outside [ci=%d, cs=%d]", (unsigned long long)m->h_stats[ST_BAD_COUNT], m->ci, m->cs);
	}
	if (m->prof.on) {
		double before = 0, after = 0;
		for (int c = 0; c < KC_N; c++) if (c != KC_QUERY) before += m->kc_seconds[c];
		prof_collect(m);
		for (int c = 0; c < KC_N; c++) if (c != KC_QUERY) after += m->kc_seconds[c];
		m->t_insert_kernels = after - before;                   // HIP-event time inside the insert kernels of this build
	}
	TRY(build_rest(m, n_rest));
	fill_model_dev(m);
	m->state = ST_READY;
	return KMX_OK;
}

static int build_common(kmx_model *m, int k, const u64 *d_kmers, const u32 *d_counts, u64 n, u64 n_total)
{
	HIPCHK(hipSetDevice(m->device));
	HIPCHK(hipEventRecord(m->ev0, m->stream));
	HIPCHK(hipMemsetAsync(m->d_nbf, 0, 24, m->stream));
	HIPCHK(hipMemsetAsync(m->d_stats, 0, ST_N * 8, m->stream));
	kmxk::histogram(d_counts, n, m->ci, m->cs, m->bf_num, m->d_nbf, m->d_stats, m->stream);    // pass 1
	u64 nbf[3], bad = 0;
	HIPCHK(hipMemcpyAsync(nbf, m->d_nbf, 24, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipMemcpyAsync(&bad, m->d_stats + ST_BAD_COUNT, 8, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	if (bad) return fail(KMX_E_RANGE, "%llu k-mers with a count outside [ci=%d, cs=%d]", (unsigned long long)bad, m->ci, m->cs);
	const auto t0 = std::chrono::steady_clock::now();
	TRY(kmx_begin(m, k, (const uint64_t *)nbf, n_total));
	if (m->dbg_ctrl) hipStreamSynchronize(m->stream);
	const auto t1 = std::chrono::steady_clock::now();
	TRY(kmx_insert_batch_dev(m, (const uint64_t *)d_kmers, d_counts, n));
	if (m->dbg_ctrl) hipStreamSynchronize(m->stream);
	const auto t2 = std::chrono::steady_clock::now();
	TRY(kmx_finish(m));
	if (m->dbg_ctrl) {
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
		fprintf(stderr, "[kmx] build: begin (allocate + clear) %.1f ms of which hipMalloc %.1f ms, insert %.1f ms, finish (last block + rest table) %.1f ms\n", ms(t0, t1), (double)g_malloc_ns.exchange(0) * 1e-6, ms(t1, t2), ms(t2, std::chrono::steady_clock::now()));
	}
	HIPCHK(hipEventRecord(m->ev1, m->stream));
	HIPCHK(hipEventSynchronize(m->ev1));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, m->ev0, m->ev1));
	m->t_total = ms * 1e-3;
	return KMX_OK;
}

static int kmx_build_dev_impl(kmx_model *m, int k, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (k < 3 || k > 64) return fail(KMX_E_ARG, "k=%d out of range [3,64]", k);
	return build_common(m, k, (const u64 *)d_kmers, (const u32 *)d_counts, n, n);
}

static int kmx_build_host_impl(kmx_model *m, int k, const uint64_t *kmers, const uint32_t *counts, uint64_t n)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (k < 3 || k > 64) return fail(KMX_E_ARG, "k=%d out of range [3,64]", k);
	if (n && (!kmers || !counts)) return fail(KMX_E_ARG, "null argument");
	HIPCHK(hipSetDevice(m->device));
	HIPCHK(hipEventRecord(m->ev0, m->stream));
	// pass 1 on the host cores (kmodel.hpp:423-428): the class histogram of the caller's counts, in parallel
	u64 nbf[3] = {0, 0, 0}, bad = 0;
	{
		const unsigned hw = std::thread::hardware_concurrency();
		const int T = (int)std::max<u64>(1, std::min<u64>(std::min<unsigned>(hw ? hw : 1, 16), n / 1000000 + 1));
		std::vector<u64> acc((size_t)T * 8, 0);
		auto work = [&](int t) {
			const u64 per = (n + T - 1) / T, lo = (u64)t * per, hi = std::min<u64>(n, lo + per);
			u64 a[4] = {0, 0, 0, 0};
			for (u64 i = lo; i < hi; i++) {
				const u32 c = counts[i];
				if (c < (u32)m->ci || c > (u32)m->cs) a[3]++;
				else if (c < (u32)(m->ci + m->bf_num)) a[c - (u32)m->ci]++;
			}
			for (int q = 0; q < 4; q++) acc[(size_t)t * 8 + q] = a[q];
		};
		if (T == 1) work(0);
		else {
			std::vector<std::thread> th;
			for (int t = 0; t < T; t++) th.emplace_back(work, t);
			for (auto &x : th) x.join();
		}
		for (int t = 0; t < T; t++) { for (int q = 0; q < 3; q++) nbf[q] += acc[(size_t)t * 8 + q]; bad += acc[(size_t)t * 8 + 3]; }
	}
	if (bad) return fail(KMX_E_RANGE, "%llu k-mers with a count outside [ci=%d, cs=%d]", (unsigned long long)bad, m->ci, m->cs);
	TRY(kmx_begin(m, k, (const uint64_t *)nbf, n));
	TRY(kmx_insert_batch(m, kmers, counts, n));                  // pinned double-buffered hipMemcpyAsync under the rounds
	TRY(kmx_finish(m));
	HIPCHK(hipEventRecord(m->ev1, m->stream));
	HIPCHK(hipEventSynchronize(m->ev1));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, m->ev0, m->ev1));
	m->t_total = ms * 1e-3;
	return KMX_OK;
}

// KModel::init(db_file) (kmodel.hpp:57-87).  The host does I/O only: it counts the classes over the record file (pass 1,
// kmodel.hpp:423-428: one parallel scan of the counter bytes) while a producer thread already copies the RAW records of
// the first batches into pinned slots; a copy stream moves a slot to the device (hipMemcpyAsync) while the model's stream
// is still inserting the previous batch; the records are decoded there (k_kmc_decode: prefix lookup, byte swaps,
// packing) and inserted.  ONE decode of the database, none of it on the host.  Databases that hold records outside the
// header's [min_count, max_count] (ReadNextKmer skips those; KMC itself never writes them) take the host decoder
// instead, and so does KMX_KMC_HOST_DECODE=1 (test hook).
namespace kmxk { void kmc_decode(const KmcDecode &, int, u64, u64, u64 *, u32 *, hipStream_t); }
namespace {
struct FeedSlot {
	unsigned char *raw = nullptr;     // GPU decode: record bytes
	u64 *km = nullptr;                // host decode: packed k-mers / counts
	u32 *cnt = nullptr;
	size_t n = 0;
	uint64_t rec0 = 0;
	bool full = false, last = false;
};
}   // namespace

// the feed of KModel::init(db) on the handle: two pinned slots of B raw records + their device twins, decoded k-mers / counts
// of a batch, a copy stream, events, the prefix LUT on the device -- kept across calls (kmx_model::KmcFeed)
static bool feed_alloc(kmx_model *m, size_t B, size_t rb, int W, const kmx::KmcListing &db)
{
	auto &F = m->feed;
	if (hipSetDevice(m->device) != hipSuccess) return false;
	bool good = F.copy || hipStreamCreateWithFlags(&F.copy, hipStreamNonBlocking) == hipSuccess;
	for (int s = 0; s < 2 && good; s++) {
		if (!F.ev_copied[s]) good = hipEventCreateWithFlags(&F.ev_copied[s], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&F.ev_free[s], hipEventDisableTiming) == hipSuccess;
		if (good && F.raw_cap < B * rb + 16) {
			if (F.raw[s]) hipHostFree(F.raw[s]);
			hipFree(F.draw[s]);
			F.raw[s] = nullptr; F.draw[s] = nullptr;
			good = hipHostMalloc((void **)&F.raw[s], B * rb + 16) == hipSuccess && hipMalloc((void **)&F.draw[s], B * rb + 16) == hipSuccess;
		}
		if (good && F.dk_cap < B * (size_t)W) {
			hipFree(F.dk[s]); hipFree(F.dc[s]);
			F.dk[s] = nullptr; F.dc[s] = nullptr;
			good = hipMalloc((void **)&F.dk[s], B * W * 8) == hipSuccess && hipMalloc((void **)&F.dc[s], B * 4) == hipSuccess;
		}
	}
	if (good) { F.raw_cap = std::max(F.raw_cap, B * rb + 16); F.dk_cap = std::max(F.dk_cap, B * (size_t)W); }
	// the prefix LUT(s): file -> the handle's pinned buffer (parallel preads into memory that is resident already) -> device
	const size_t n_lut = db.lut_entries();
	if (good && F.lut_cap < n_lut) {
		hipFree(F.d_lut); F.d_lut = nullptr;
		if (F.h_lut) hipHostFree(F.h_lut);
		F.h_lut = nullptr;
		good = hipMalloc((void **)&F.d_lut, n_lut * 8) == hipSuccess && hipHostMalloc((void **)&F.h_lut, n_lut * 8) == hipSuccess;
		F.lut_cap = good ? n_lut : 0;
	}
	if (good) good = n_lut && db.read_lut((uint64_t *)F.h_lut) && hipMemcpy(F.d_lut, F.h_lut, n_lut * 8, hipMemcpyHostToDevice) == hipSuccess;
	if (!good) F.raw_cap = F.dk_cap = 0;                         // whatever is half there is replaced next time
	return good;
}

static int kmx_build_from_kmc_impl(kmx_model *m, const char *db_prefix)
{
	if (!m || !db_prefix) return fail(KMX_E_ARG, "null argument");
	const char *tr = getenv("KMX_INIT_TRACE");                       // wall-clock phase times on stderr (no extra synchronisation)
	const bool trace = tr && atoi(tr);
	const auto t_start = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) { if (trace) fprintf(stderr, "[kmx] init(db) %-28s at %7.2f ms\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() * 1e3); };
	kmx::KmcListing db;
	if (!db.open(db_prefix, false)) return fail(KMX_E_IO, "can't open the kmer_data_base %s: %s", db_prefix, db.error().c_str());
	unsigned hw = std::thread::hardware_concurrency();
	const int T = hw > 32 ? 16 : (hw > 1 ? (int)hw / 2 : 1);          // per activity: pass 1 and the producer run side by side
	db.set_threads(T);
	const int k = (int)db.kmer_length(), W = db.words();
	const size_t B = size_t(1) << 23;
	const size_t rb = db.record_bytes();
	HIPCHK(hipSetDevice(m->device));
	lap("database open");
	// ONE PASS (KMX_ONE_PASS=1 under KMX_TEST_HOOKS): the whole listing is streamed to the device and decoded there
	// once -- pread of batch b + 1 under copy + decode of batch b --, the classes are counted ON the device (k_histogram), and the
	// build runs on the resident listing (build_common = kmx_build_dev).  No host scan; but no insert can start before the last
	// batch has arrived (every length is a function of the class counts), so the stream is not hidden under the rounds any more:
	// measured 10^8 31-mers, same box, profiles/r05_one_pass_ab.txt.  Needs N * (8 W + 4) bytes of HBM beside the model; records
	// outside the header's count range (KMC never writes them) are not skipped here -- the two-pass path below is the product.
	if (const char *op = hook_env("KMX_ONE_PASS")) if (op[0] == '1') {
		const u64 N = db.records();
		if (!feed_alloc(m, B, rb, W, db)) return fail(KMX_E_NOMEM, "pinned / device buffers for the listing feed could not be allocated");
		DevMem all_k, all_c;
		HIPCHK(all_k.alloc(std::max<u64>(N, 1) * W * 8));
		HIPCHK(all_c.alloc(std::max<u64>(N, 1) * 4));
		auto &F = m->feed;
		hipStream_t st = m->stream;
		KmcDecode kd;
		kd.lut = F.d_lut; kd.n_lut = db.lut_entries() - 1; kd.prefix_mask = db.prefix_mask();
		kd.rec_bytes = (u32)rb; kd.suf_bytes = db.suffix_bytes(); kd.cnt_bytes = db.counter_bytes();
		HIPCHK(hipEventRecord(F.ev_free[0], st)); HIPCHK(hipEventRecord(F.ev_free[1], st));
		HIPCHK(hipEventRecord(F.ev_copied[0], F.copy)); HIPCHK(hipEventRecord(F.ev_copied[1], F.copy));
		int s2 = 0;
		for (u64 done = 0; done < N; done += B, s2 ^= 1) {
			const u64 c = std::min<u64>(B, N - done);
			HIPCHK(hipEventSynchronize(F.ev_copied[s2]));                     // the pinned slot has been read (two batches ago)
			db.copy_records(done, c, F.raw[s2]);
			HIPCHK(hipStreamWaitEvent(F.copy, F.ev_free[s2], 0));
			HIPCHK(hipMemcpyAsync(F.draw[s2], F.raw[s2], c * rb, hipMemcpyHostToDevice, F.copy));
			HIPCHK(hipEventRecord(F.ev_copied[s2], F.copy));
			HIPCHK(hipStreamWaitEvent(st, F.ev_copied[s2], 0));
			kd.recs = F.draw[s2];
			kmxk::kmc_decode(kd, W, done, c, (u64 *)all_k.p + done * W, (u32 *)all_c.p + done, st);
			HIPCHK(hipEventRecord(F.ev_free[s2], st));
		}
		if (db.io_failed()) return fail(KMX_E_IO, "reading %s.kmc_suf failed", db_prefix);
		lap("listing enqueued (one pass)");
		const int rc1 = build_common(m, k, (const u64 *)all_k.p, (const u32 *)all_c.p, N, db.kmer_count());
		lap("finish returned");
		hipStreamSynchronize(st);
		return rc1;
	}
	const char *force_host = hook_env("KMX_KMC_HOST_DECODE");
	bool gpu_decode = !(force_host && atoi(force_host));
	FeedSlot slot[2];
	auto &F = m->feed;
	std::mutex mu;
	std::condition_variable cv;
	bool stop = false, alloc_done = false;
	std::thread producer, allocator;
	auto cleanup = [&] {
		{ std::lock_guard<std::mutex> lk(mu); stop = true; for (auto &sl : slot) sl.full = false; }
		cv.notify_all();
		if (producer.joinable()) producer.join();
		if (allocator.joinable()) allocator.join();
		if (F.copy) hipStreamSynchronize(F.copy);
		hipStreamSynchronize(m->stream);
	};
	auto guard = scope_exit(cleanup);                                // also when something throws (bad_alloc in pass 1): threads are joined before the frame goes
	hipEventRecord(m->ev0, m->stream);
	// the buffers of pass 2 are (re)allocated beside pass 1 when this database needs larger ones than the handle holds
	bool ok = false;
	allocator = std::thread([&] {
		const bool good = feed_alloc(m, B, rb, W, db);
		{ std::lock_guard<std::mutex> lk(mu); ok = good; alloc_done = true; }
		cv.notify_all();
	});
	// The producer fills the two slots in turn: raw records (a parallel pread out of the page cache) or, in host mode, decoded
	// k-mers.  In raw mode it needs nothing pass 1 finds out, so it starts as soon as the slots exist and works BESIDE pass 1:
	// when the class counts are known the first two batches already sit in pinned memory.
	auto start_producer = [&](bool raw) {
		producer = std::thread([&, raw] {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return alloc_done || stop; });
				if (stop || !ok) return;
			}
			for (int s = 0; s < 2; s++) { slot[s].raw = F.raw[s]; slot[s].km = F.km[s]; slot[s].cnt = F.cnt[s]; }
			uint64_t rec = 0;
			if (!raw) db.restart();                                          // kmodel.hpp:430
			for (int s = 0;; s ^= 1) {
				{
					std::unique_lock<std::mutex> lk(mu);
					cv.wait(lk, [&] { return !slot[s].full || stop; });
					if (stop) return;
				}
				size_t got;
				const uint64_t rec0 = rec;
				if (raw) {
					got = (size_t)std::min<uint64_t>(B, db.records() > rec ? db.records() - rec : 0);
					if (got) db.copy_records(rec, got, slot[s].raw);
					rec += got;
				} else got = db.next_batch((uint64_t *)slot[s].km, slot[s].cnt, B);
				{ std::lock_guard<std::mutex> lk(mu); slot[s].n = got; slot[s].rec0 = rec0; slot[s].last = got == 0; slot[s].full = true; }
				cv.notify_all();
				if (!got) return;
			}
		});
	};
	if (gpu_decode) start_producer(true);
	uint64_t nbf[3] = {0, 0, 0}, bad = 0, not_listed = 0;
	int rc = KMX_OK;
	const auto t_p1 = std::chrono::steady_clock::now();
	// pass 1 (kmodel.hpp:423-428), counts only.  Everything waits for it (the sizes depend on it), and it is a copy out of the
	// page cache + a scan: it takes as many threads as the machine has to spare beside the producer's
	const int T1 = hw > 32 ? 32 : T;
	db.count_classes((u32)m->ci, (u32)m->cs, m->bf_num, nbf, &bad, &not_listed, T1);
	const double s_p1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_p1).count();
	lap("pass 1 done");
	allocator.join();
	lap("buffers ready");
	if (!ok) return fail(KMX_E_NOMEM, "pinned / device buffers for the listing feed could not be allocated");
	if (not_listed && gpu_decode) {                                  // records outside [min_count, max_count]: only the host decoder skips them
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv.notify_all();
		producer.join();
		{ std::lock_guard<std::mutex> lk(mu); stop = false; for (auto &sl : slot) sl = FeedSlot(); }
		gpu_decode = false;
	}
	if (!gpu_decode && F.km_cap < B * (size_t)W) {                   // host decoder: pinned k-mer / count slots
		for (int s = 0; s < 2 && !rc; s++) {
			if (F.km[s]) hipHostFree(F.km[s]);
			if (F.cnt[s]) hipHostFree(F.cnt[s]);
			F.km[s] = nullptr; F.cnt[s] = nullptr;
			if (hipHostMalloc((void **)&F.km[s], B * W * 8) != hipSuccess || hipHostMalloc((void **)&F.cnt[s], B * 4) != hipSuccess) rc = fail(KMX_E_NOMEM, "pinned buffers for the listing feed could not be allocated");
		}
		F.km_cap = rc ? 0 : B * (size_t)W;
	}
	if (!rc && !gpu_decode) start_producer(false);
	if (!rc && db.io_failed()) rc = fail(KMX_E_IO, "reading %s.kmc_suf failed during pass 1", db_prefix);
	if (!rc && bad) rc = fail(KMX_E_RANGE, "%llu k-mers with a count outside [ci=%d, cs=%d]", (unsigned long long)bad, m->ci, m->cs);
	if (!rc) rc = kmx_begin(m, k, nbf, db.kmer_count());
	lap("begin returned");
	double s_wait = 0;
	if (!rc) {
		KmcDecode kd;
		kd.lut = F.d_lut; kd.n_lut = db.lut_entries() - 1; kd.prefix_mask = db.prefix_mask();
		kd.rec_bytes = (u32)rb; kd.suf_bytes = db.suffix_bytes(); kd.cnt_bytes = db.counter_bytes();
		// pass 2 (kmodel.hpp:68-74).  Batch b travels through slot b%2 and device buffers b%2; its copy is enqueued one
		// batch ahead of its insert, so it runs under the rounds of batch b-1.
		auto enqueue_copy = [&](int s) -> bool {                               // slot s is full
			if (hipStreamWaitEvent(F.copy, F.ev_free[s], 0) != hipSuccess) return false;      // the insert that read the device buffers two batches ago
			if (gpu_decode) { if (hipMemcpyAsync(F.draw[s], slot[s].raw, slot[s].n * rb, hipMemcpyHostToDevice, F.copy) != hipSuccess) return false; }
			else {
				if (hipMemcpyAsync(F.dk[s], slot[s].km, slot[s].n * W * 8, hipMemcpyHostToDevice, F.copy) != hipSuccess) return false;
				if (hipMemcpyAsync(F.dc[s], slot[s].cnt, slot[s].n * 4, hipMemcpyHostToDevice, F.copy) != hipSuccess) return false;
			}
			return hipEventRecord(F.ev_copied[s], F.copy) == hipSuccess;
		};
		auto wait_full = [&](int s) {
			const auto t0 = std::chrono::steady_clock::now();
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return slot[s].full; });
			s_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		};
		auto release = [&](int s) {
			{ std::lock_guard<std::mutex> lk(mu); slot[s].full = false; }
			cv.notify_all();
		};
		hipEventRecord(F.ev_free[0], m->stream);
		hipEventRecord(F.ev_free[1], m->stream);
		wait_full(0);
		bool copied = !slot[0].last && enqueue_copy(0);
		if (!slot[0].last && !copied) rc = fail(KMX_E_NODEVICE, "H2D copy failed");
		for (int s = 0; !slot[s].last; s ^= 1) {
			const size_t n = slot[s].n;
			// the next batch: produced meanwhile, copied under this batch's rounds
			wait_full(s ^ 1);
			if (!rc && !slot[s ^ 1].last && !enqueue_copy(s ^ 1)) rc = fail(KMX_E_NODEVICE, "H2D copy failed");
			if (!rc) {
				if (hipStreamWaitEvent(m->stream, F.ev_copied[s], 0) != hipSuccess) rc = fail(KMX_E_NODEVICE, "stream wait failed");
				else {
					if (gpu_decode) { kd.recs = F.draw[s]; kmxk::kmc_decode(kd, W, slot[s].rec0, n, F.dk[s], F.dc[s], m->stream); }
					rc = kmx_insert_batch_dev(m, (const uint64_t *)F.dk[s], F.dc[s], n);
				}
				hipEventRecord(F.ev_free[s], m->stream);                       // everything that reads the device buffers of s is enqueued by now
			}
			hipEventSynchronize(F.ev_copied[s]);                               // the pinned slot has been read
			release(s);
		}
	}
	lap("all batches enqueued");
	if (!rc && db.io_failed()) { rc = fail(KMX_E_IO, "reading %s.kmc_suf failed during pass 2: the model is dropped", db_prefix); m->state = ST_EMPTY; }
	if (!rc) rc = kmx_finish(m);
	lap("finish returned");
	if (!rc) {
		hipEventRecord(m->ev1, m->stream);
		hipEventSynchronize(m->ev1);
		float ms = 0;
		hipEventElapsedTime(&ms, m->ev0, m->ev1);
		m->t_total = ms * 1e-3;
		if (m->dbg_ctrl) fprintf(stderr, "[kmx] init(db): pass 1 %.1f ms, waited %.1f ms for the listing in pass 2, total %.1f ms (%s decode)\n", s_p1 * 1e3, s_wait * 1e3, (double)ms, gpu_decode ? "GPU" : "host");
	}
	return rc;
}

// ------------------------------------------------------------------------------------------ one model, several GPUs
// SURVEY §8e.  One process per GPU; each handle holds ONE rank's share of a single model.  What the reference does with
// n_bits OpenMP threads -- thread i walks buffer i against array (i + t) % n_bits, barrier, rotate (kmodel.hpp:560-565) --
// is done by a ring of GPUs that own the arrays whole: after a round the survivors of a list travel to the owner of the
// next array.  The order-free parts (Bloom filters, back filters, km_back: set_bit is an OR, kmodel.hpp:576-581) are
// built as per-rank partial filters and merged by OR.  The exchange itself (RCCL all-to-all / send-recv / broadcast)
// is the caller's: kmcex_amd/dist.py drives these entry points through torch.distributed.
static int kmx_count_classes_dev_impl(kmx_model *m, const uint32_t *d_counts, uint64_t n, uint64_t n_bf[3])
{
	if (!m || !n_bf) return fail(KMX_E_ARG, "null argument");
	HIPCHK(hipSetDevice(m->device));
	HIPCHK(hipMemsetAsync(m->d_nbf, 0, 24, m->stream));
	HIPCHK(hipMemsetAsync(m->d_stats, 0, ST_N * 8, m->stream));
	kmxk::histogram((const u32 *)d_counts, n, m->ci, m->cs, m->bf_num, m->d_nbf, m->d_stats, m->stream);    // pass 1 (kmodel.hpp:423-428)
	u64 bad = 0;
	HIPCHK(hipMemcpyAsync(n_bf, m->d_nbf, 24, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipMemcpyAsync(&bad, m->d_stats + ST_BAD_COUNT, 8, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	if (bad) return fail(KMX_E_RANGE, "%llu k-mers with a count outside [ci=%d, cs=%d]", (unsigned long long)bad, m->ci, m->cs);
	return KMX_OK;
}

static int kmx_shard_begin_impl(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world)
{
	if (world < 1 || rank < 0 || rank >= world) return fail(KMX_E_ARG, "bad rank %d of %d", rank, world);
	TRY(kmx_begin_impl(m, k, n_bf, n_total));                   // whole-model sizes (kmodel.hpp:402-456): every rank allocates every array
	m->ring = true; m->ring_rank = rank; m->ring_world = world;
	m->bd.kmers = m->d_stg_kmers;                              // the lists a rank attempts are imported into the block staging area
	m->bd.counts = m->d_stg_counts;
	return KMX_OK;
}

// Front end of this rank's slice of the listing (kmodel.hpp:70-73): Bloom-class k-mers go into this rank's PARTIAL
// filters; coupled-array k-mers are compacted, in listing order, into the caller's buffers (capacity >= n).
static int kmx_shard_classify_dev_impl(kmx_model *m, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n, uint64_t *d_out_kmers, uint32_t *d_out_counts, uint64_t *n_out)
{
	if (!m || !n_out) return fail(KMX_E_ARG, "null argument");
	if (m->state != ST_BUILDING || !m->ring) return fail(KMX_E_STATE, "shard_classify before shard_begin");
	*n_out = 0;
	if (!n) return KMX_OK;
	HIPCHK(hipSetDevice(m->device));
	TRY(ensure_front_end(m, n));
	const u64 n_chunks = (n + kChunk - 1) / kChunk;
	kmxk::classify_count(m->md, (const u64 *)d_kmers, (const u32 *)d_counts, n, kChunk, m->d_tile_cnt, m->d_tile_off, m->d_totals, m->d_stats, m->blm, m->blm_sweep_every, m->stream, &m->prof);
	HIPCHK(hipMemcpyAsync(m->h_totals, m->d_totals, n_chunks * 4, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	u64 base = 0;
	for (u64 ci = 0, done = 0; ci < n_chunks; ci++) {
		const u64 c = std::min<u64>(kChunk, n - done);
		kmxk::classify_scatter(m->md, (const u64 *)d_kmers + done * m->W, (const u32 *)d_counts + done, c, m->d_tile_off + ci * (kChunk / KMX_CLS_TILE),
		                       (u64 *)d_out_kmers, (u32 *)d_out_counts, base, m->stream);
		base += (u64)m->h_totals[ci];
		done += c;
	}
	HIPCHK(hipGetLastError());
	*n_out = base;
	return KMX_OK;
}

static uint64_t ring_msg_bytes(int k) { return 8ull * KMX_MSG_HDR + (u64)KMX_BUCKET * (8ull * ((k + 31) / 32) + 4); }

// One round t on the lists this rank holds: import, A/B/S/R (the same kernels as the single-GPU build; the lists that
// are elsewhere in the ring are empty here), then every list leaves as a message or, after the last round, goes to the
// rest table (kmodel.hpp:567-571).
static int kmx_ring_round_dev_impl(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists)
{
	if (!m || (n_lists && !lists)) return fail(KMX_E_ARG, "null argument");
	if (m->state != ST_BUILDING || !m->ring) return fail(KMX_E_STATE, "ring_round before shard_begin");
	const int nb = m->nb;
	if (t < 0 || t >= nb || n_lists < 0 || n_lists > nb) return fail(KMX_E_ARG, "bad round %d / %d lists", t, n_lists);
	if (m->km_byte_size == 0) return KMX_OK;                      // divergence D2: no arrays to insert into
	HIPCHK(hipSetDevice(m->device));
	RingLists rl;
	memset(&rl, 0, sizeof rl);
	for (int e = 0; e < n_lists; e++) {
		const kmx_ring_list &l = lists[e];
		if (l.list < 0 || l.list >= nb || rl.e[l.list].active) return fail(KMX_E_ARG, "bad or repeated list %d", l.list);
		if (l.n_host > (int)KMX_BUCKET) return fail(KMX_E_ARG, "list %d longer than a buffer", l.list);
		if (l.n_host >= 0 ? (l.n_host > 0 && (!l.src_kmers || !l.src_counts)) : !l.src_msg) return fail(KMX_E_ARG, "list %d has no source", l.list);
		RingList &r = rl.e[l.list];
		r.active = 1; r.n_host = l.n_host;
		r.src_kmers = (const u64 *)l.src_kmers; r.src_counts = (const u32 *)l.src_counts; r.src_msg = (const u64 *)l.src_msg;
		r.dst_msg = (u64 *)l.dst_msg;
	}
	if (t == 0) steer_passes(m);
	// The winners of a list commit beside the NEXT round's check on this rank, like in the single-GPU build: arrays are owned
	// whole, so the list that visits array a in round t+1 is examined by the rank that committed round t on a -- the claims its
	// detect needs as settled positions are this rank's own (kernels.hip, "A: check + emit claims").
	// (A rank that had no list in the round before -- the short final block -- was not called for it: what it still owes would
	// be committed a round late, beside a check whose detect looks for the delta in the wrong list.  Commit it first.)
	if (m->pending && t != (m->pending_t + 1) % nb) TRY(flush_pending_commit(m));
	const int pp = m->pp;
	kmxk::ring_import(m->md, m->bd, pp, rl, m->d_stg_kmers, m->d_stg_counts, m->stream);
	TRY(run_round(m, t, m->defer, nullptr));
	TRY(kmback_emit(m, t, pp, -1, (u64)n_lists * KMX_BUCKET));
	bool any_out = false;
	for (int i = 0; i < nb; i++) any_out |= rl.e[i].active && rl.e[i].dst_msg;
	if (any_out) kmxk::ring_export(m->md, m->bd, pp ^ 1, rl, m->d_feedback, m->stream);
	for (int i = 0; i < nb; i++)
		if (rl.e[i].active && !rl.e[i].dst_msg) {
			TRY(ensure_rest_capacity(m, (u64)KMX_BUCKET + (u64)nb));
			kmxk::rest_append(m->md, m->bd, pp ^ 1, i, 1, m->d_rest_kmers, m->d_rest_counts, m->d_rest_n, m->d_stale_kmers, m->d_stale_counts, m->d_feedback, m->stream);
		}
	if (t == nb - 1) m->blocks++;
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

// Quirk Q1 in the ring: the rank that retired list i in the previous block holds its stale slot 0 (kmodel.hpp:520-527, :539)
static int kmx_ring_stale_dup_dev_impl(kmx_model *m, int first_unused_row)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_BUILDING || !m->ring) return fail(KMX_E_STATE, "ring_stale_dup before shard_begin");
	if (first_unused_row < 0 || first_unused_row >= m->nb || m->km_byte_size == 0) return KMX_OK;
	HIPCHK(hipSetDevice(m->device));
	TRY(ensure_rest_capacity(m, (u64)m->nb));
	hipLaunchKernelGGL(k_stale_dup, dim3(1), dim3(64), 0, m->stream, first_unused_row, m->nb, m->W, (const u64 *)m->d_stale_kmers,
	                   (const int *)m->d_stale_counts, m->d_rest_kmers, m->d_rest_counts, m->d_rest_n, m->d_stats);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

// this rank's share when its last round is enqueued: statistics and the survivors it retired (device pointers)
static int kmx_shard_local_impl(kmx_model *m, kmx_stats *partial, void **d_rest_kmers, void **d_rest_counts)
{
	if (!m || !partial) return fail(KMX_E_ARG, "null argument");
	if (m->state != ST_BUILDING || !m->ring) return fail(KMX_E_STATE, "shard_local before shard_begin");
	HIPCHK(hipSetDevice(m->device));
	TRY(flush_pending_commit(m));                                 // the arrays are about to be read by the caller's broadcasts
	TRY(kmback_flush(m));
	unsigned long long n_rest = 0;
	int range_ovf = 0;
	HIPCHK(hipMemcpyAsync(m->h_stats, m->d_stats, ST_N * 8, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipMemcpyAsync(&n_rest, m->d_rest_n, 8, hipMemcpyDeviceToHost, m->stream));
	if (m->range.on && m->range.d_ovf) HIPCHK(hipMemcpyAsync(&range_ovf, m->range.d_ovf, sizeof(int), hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	if (m->h_stats[ST_BAD_COUNT]) return fail(KMX_E_RANGE, "%llu k-mers with a count outside [ci=%d, cs=%d]", (unsigned long long)m->h_stats[ST_BAD_COUNT], m->ci, m->cs);
	if (m->prof.on) prof_collect(m);
	memset(partial, 0, sizeof *partial);
	partial->attempts = m->h_stats[ST_ATTEMPTS]; partial->successes = m->h_stats[ST_SUCCESSES];
	partial->fast_commits = m->h_stats[ST_SUCCESSES] - m->h_stats[ST_SLOW_SUCC];
	partial->contended = m->h_stats[ST_CONTENDED]; partial->finisher_iters = m->h_stats[ST_FIN_ITERS];
	partial->rest_entries = n_rest; partial->blocks = m->blocks; partial->rounds = m->rounds;
	partial->reserved = range_ovf;                                // a fixed-size region of the range partition dropped words: this build is void
	if (d_rest_kmers) *d_rest_kmers = m->d_rest_kmers;
	if (d_rest_counts) *d_rest_counts = m->d_rest_counts;
	return KMX_OK;
}

// The merged model: the caller has OR-merged the filters and broadcast the arrays into this handle's memory
// (kmx_dev_view); `d_rest_*` hold the survivors of ALL ranks (any order: KRestData::build sorts, rest.hpp:95-135).
static int kmx_shard_complete_impl(kmx_model *m, const uint64_t *d_rest_kmers, const int32_t *d_rest_counts, uint64_t n_rest, const kmx_stats *totals)
{
	if (!m || !totals) return fail(KMX_E_ARG, "null argument");
	if (m->state != ST_BUILDING || !m->ring) return fail(KMX_E_STATE, "shard_complete before shard_begin");
	if (n_rest && (!d_rest_kmers || !d_rest_counts)) return fail(KMX_E_ARG, "null rest list");
	HIPCHK(hipSetDevice(m->device));
	if (n_rest && (const u64 *)d_rest_kmers != m->d_rest_kmers) {
		if (n_rest > m->rest_cap) {
			HIPCHK(hipStreamSynchronize(m->stream));
			hipFree(m->d_rest_kmers); hipFree(m->d_rest_counts);
			m->d_rest_kmers = nullptr; m->d_rest_counts = nullptr; m->rest_cap = 0;
			TRY(dalloc(&m->d_rest_kmers, n_rest * m->W, false, m->stream));
			TRY(dalloc(&m->d_rest_counts, n_rest, false, m->stream));
			m->rest_cap = n_rest;
		}
		HIPCHK(hipMemcpyAsync(m->d_rest_kmers, d_rest_kmers, n_rest * m->W * 8, hipMemcpyDeviceToDevice, m->stream));
		HIPCHK(hipMemcpyAsync(m->d_rest_counts, d_rest_counts, n_rest * 4, hipMemcpyDeviceToDevice, m->stream));
	}
	TRY(build_rest(m, n_rest));
	m->h_stats[ST_ATTEMPTS] = totals->attempts; m->h_stats[ST_SUCCESSES] = totals->successes;
	m->h_stats[ST_SLOW_SUCC] = totals->successes - totals->fast_commits;
	m->h_stats[ST_CONTENDED] = totals->contended; m->h_stats[ST_FIN_ITERS] = totals->finisher_iters;
	m->blocks = totals->blocks; m->rounds = totals->rounds;
	fill_model_dev(m);
	m->ring = false;
	m->range.on = false;
	m->state = ST_READY;
	return KMX_OK;
}

#include "multi_build.h"

#include "range_host.h"

// ------------------------------------------------------------------------------------------ KMC listing (host only)
static int kmx_kmc_info_impl(const char *db_prefix, int *k, uint64_t *total_kmers)
{
	if (!db_prefix) return fail(KMX_E_ARG, "null argument");
	kmx::KmcListing db;
	if (!db.open(db_prefix)) return fail(KMX_E_IO, "can't open the kmer_data_base %s: %s", db_prefix, db.error().c_str());
	if (k) *k = (int)db.kmer_length();
	if (total_kmers) *total_kmers = db.kmer_count();
	return KMX_OK;
}

static int kmx_kmc_read_impl(const char *db_prefix, uint64_t *kmers, uint32_t *counts, uint64_t capacity, uint64_t *n_read)
{
	if (!db_prefix || !kmers || !counts || !n_read) return fail(KMX_E_ARG, "null argument");
	kmx::KmcListing db;
	if (!db.open(db_prefix)) return fail(KMX_E_IO, "can't open the kmer_data_base %s: %s", db_prefix, db.error().c_str());
	const int W = db.words();
	uint64_t n = 0;
	for (size_t got; n < capacity && (got = db.next_batch(kmers + n * W, counts + n, (size_t)(capacity - n))) > 0;) n += got;
	*n_read = n;
	if (db.io_failed()) return fail(KMX_E_IO, "reading %s.kmc_suf failed", db_prefix);
	return KMX_OK;
}

// ------------------------------------------------------------------------------------------ query
static int kmx_query_packed_dev_impl(kmx_model *m, const uint64_t *d_kmers, uint64_t n, int32_t *d_out)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_READY) return fail(KMX_E_STATE, "query before the model is built or loaded");
	HIPCHK(hipSetDevice(m->device));
	if (m->prof.count && m->d_stats) {                            // accounting (kmx_set_profile(m, 2)): never the timed kernel
		kmxk::query(m->md, (const u64 *)d_kmers, n, d_out, m->stream, &m->prof, m->d_stats + ST_QUERY_NEIGH);
		u64 got = 0;
		HIPCHK(hipMemcpyAsync(&got, m->d_stats + ST_QUERY_NEIGH, 8, hipMemcpyDeviceToHost, m->stream));
		HIPCHK(hipStreamSynchronize(m->stream));
		m->h_stats[ST_QUERY_NEIGH] = got; m->h_stats[ST_QUERY_N] += n;
		return KMX_OK;
	}
	kmxk::query(m->md, (const u64 *)d_kmers, n, d_out, m->stream, &m->prof);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

static int kmx_query_packed_impl(kmx_model *m, const uint64_t *kmers, uint64_t n, int32_t *out)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_READY) return fail(KMX_E_STATE, "query before the model is built or loaded");
	if (!n) return KMX_OK;
	HIPCHK(hipSetDevice(m->device));
	DevMem dk, dout;
	HIPCHK(dk.alloc(n * m->W * 8));
	HIPCHK(dout.alloc(n * 4));
	int rc = hipMemcpyAsync(dk.p, kmers, n * m->W * 8, hipMemcpyHostToDevice, m->stream) == hipSuccess ? KMX_OK : fail(KMX_E_NODEVICE, "H2D copy failed");
	if (!rc) rc = kmx_query_packed_dev(m, dk.as<uint64_t>(), n, dout.as<int32_t>());
	if (!rc && hipMemcpyAsync(out, dout.p, n * 4, hipMemcpyDeviceToHost, m->stream) != hipSuccess) rc = fail(KMX_E_NODEVICE, "D2H copy failed");
	if (hipStreamSynchronize(m->stream) != hipSuccess && !rc) rc = fail(KMX_E_NODEVICE, "query failed");
	return rc;
}

// vector<string> front door (kmodel.hpp:90-116).  The reference splits the vector over t_num threads (:93-96); here the
// batch flows through a three-slot pipeline in chunks: worker threads turn chunk c into packed k-mers inside a pinned slot
// (strpack.cpp: 8 bytes per string over the link instead of k), a copy stream moves it (hipMemcpyAsync), the model's stream
// answers it (k_query), a second copy stream brings the answers back, and the workers hand them to the caller's array --
// chunk c is packed while c-1 is on the GPU and c-2 is copied out.  The work is dealt in tasks of 2^14 strings from one
// atomic counter (pack tasks of chunk c, then copy-out tasks of chunk c-2, and so on): no barrier between the threads.
// A string the packed form cannot express (other characters) is remembered by its index and answered afterwards by the
// byte-string kernel, which hashes the bytes as they are, exactly like the reference does -- the cost of dirty strings is
// proportional to their number; a batch of another length than the model's k travels as bytes altogether.
static const u64 kQuerySub = u64(1) << 14;                     // strings per task
static const size_t kQuerySlotBytes = size_t(32) << 20;        // input bytes per slot

static int ensure_query_feed(kmx_model *m, size_t in_bytes, size_t answers)
{
	auto &F = m->qfeed;
	if (!F.to_dev) HIPCHK(hipStreamCreateWithFlags(&F.to_dev, hipStreamNonBlocking));
	if (!F.to_host) HIPCHK(hipStreamCreateWithFlags(&F.to_host, hipStreamNonBlocking));
	for (int s = 0; s < F.S; s++)
		for (hipEvent_t *e : {&F.ev_in[s], &F.ev_k[s], &F.ev_out[s]})
			if (!*e) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
	if (in_bytes > F.in_cap) {
		HIPCHK(hipStreamSynchronize(m->stream));
		for (int s = 0; s < F.S; s++) {
			if (F.h_in[s]) hipHostFree(F.h_in[s]);
			hipFree(F.d_in[s]);
			F.h_in[s] = F.d_in[s] = nullptr;
		}
		F.in_cap = 0;
		for (int s = 0; s < F.S; s++) {
			HIPCHK(hipHostMalloc((void **)&F.h_in[s], in_bytes));
			HIPCHK(timed_malloc((void **)&F.d_in[s], in_bytes));
		}
		F.in_cap = in_bytes;
	}
	if (answers > F.out_cap) {
		HIPCHK(hipStreamSynchronize(m->stream));
		for (int s = 0; s < F.S; s++) {
			if (F.h_out[s]) hipHostFree(F.h_out[s]);
			hipFree(F.d_out[s]);
			F.h_out[s] = F.d_out[s] = nullptr;
		}
		F.out_cap = 0;
		for (int s = 0; s < F.S; s++) {
			HIPCHK(hipHostMalloc((void **)&F.h_out[s], answers * 4));
			HIPCHK(timed_malloc((void **)&F.d_out[s], answers * 4));
		}
		F.out_cap = answers;
	}
	return KMX_OK;
}

// n items of item_bytes each through the pipeline.  stage(worker, lo, hi, dst): items [lo, hi) -> dst (their place in the
// slot); launch(slot, count): the kernel from d_in[slot] to d_out[slot] on the model's stream; answers -> out[0 .. n).
template <typename STAGE, typename LAUNCH>
static int query_pipeline(kmx_model *m, u64 n, size_t item_bytes, int T, STAGE stage, LAUNCH launch, int32_t *out)
{
	auto &F = m->qfeed;
	u64 C = std::max<u64>(kQuerySub, (kQuerySlotBytes / item_bytes) & ~(kQuerySub - 1));
	C = std::min<u64>(C, (n + kQuerySub - 1) & ~(kQuerySub - 1));
	const u64 nc = (n + C - 1) / C;
	TRY(ensure_query_feed(m, (size_t)C * item_bytes, (size_t)C));
	auto count_of = [&](u64 c) { return std::min<u64>(C, n - c * C); };
	auto subs_of = [&](u64 c) { return (count_of(c) + kQuerySub - 1) / kQuerySub; };
	auto enqueue = [&](u64 c) -> bool {
		const int s = (int)(c % F.S);
		const u64 cn = count_of(c);
		if (hipMemcpyAsync(F.d_in[s], F.h_in[s], cn * item_bytes, hipMemcpyHostToDevice, F.to_dev) != hipSuccess || hipEventRecord(F.ev_in[s], F.to_dev) != hipSuccess ||
		    hipStreamWaitEvent(m->stream, F.ev_in[s], 0) != hipSuccess) return false;
		launch(s, cn);
		return hipEventRecord(F.ev_k[s], m->stream) == hipSuccess && hipStreamWaitEvent(F.to_host, F.ev_k[s], 0) == hipSuccess &&
		       hipMemcpyAsync(F.h_out[s], F.d_out[s], cn * 4, hipMemcpyDeviceToHost, F.to_host) == hipSuccess && hipEventRecord(F.ev_out[s], F.to_host) == hipSuccess;
	};
	if (nc == 1 && T == 1) {                                      // a handful of strings: no threads
		stage(0, 0, n, F.h_in[0]);
		if (!enqueue(0) || hipEventSynchronize(F.ev_out[0]) != hipSuccess) return fail(KMX_E_NODEVICE, "query failed");
		memcpy(out, F.h_out[0], n * 4);
		return KMX_OK;
	}
	// phase p = the pack tasks of chunk p, then the copy-out tasks of chunk p - 2
	const u64 lag = F.S - 1, n_phase = nc + lag;
	std::vector<u64> first(n_phase + 1, 0);
	for (u64 p = 0; p < n_phase; p++) first[p + 1] = first[p] + (p < nc ? subs_of(p) : 0) + (p >= lag ? subs_of(p - lag) : 0);
	std::unique_ptr<std::atomic<u32>[]> packed(new std::atomic<u32>[nc]), copied(new std::atomic<u32>[nc]), ready(new std::atomic<u32>[nc]);
	for (u64 c = 0; c < nc; c++) { packed[c] = 0; copied[c] = 0; ready[c] = 0; }
	std::atomic<u64> next{0};
	std::atomic<bool> abort{false};
	auto wait_for = [&](auto cond) {                              // short waits: spin, then give the core away
		for (unsigned spins = 0; !cond(); spins++) {
			if (abort.load(std::memory_order_relaxed)) return false;
			if (spins < 256) __builtin_ia32_pause(); else std::this_thread::yield();
		}
		return true;
	};
	std::atomic<bool> worker_threw{false};
	auto worker = [&](int t) {
		u64 p = 0;
		try {                                                        // (`stage` may grow a vector: a worker that throws ends the pipeline with KMX_E_NOMEM, not the process)
		for (;;) {
			const u64 id = next.fetch_add(1, std::memory_order_relaxed);
			if (id >= first[n_phase]) return;
			while (id >= first[p + 1]) p++;
			u64 s = id - first[p];
			const u64 n_pack = p < nc ? subs_of(p) : 0;
			if (s < n_pack) {                                        // pack sub-piece s of chunk p
				const u64 c = p;
				if (c >= (u64)F.S && !wait_for([&] { return copied[c - F.S].load(std::memory_order_acquire) == subs_of(c - F.S); })) return;
				const u64 lo = c * C + s * kQuerySub, hi = std::min<u64>(lo + kQuerySub, c * C + count_of(c));
				stage(t, lo, hi, F.h_in[c % F.S] + (lo - c * C) * item_bytes);
				packed[c].fetch_add(1, std::memory_order_release);
			} else {                                                 // hand sub-piece s of chunk p - lag's answers to the caller
				s -= n_pack;
				const u64 c = p - lag;
				if (!wait_for([&] { return ready[c].load(std::memory_order_acquire) != 0; })) return;
				const u64 lo = c * C + s * kQuerySub, hi = std::min<u64>(lo + kQuerySub, c * C + count_of(c));
				memcpy(out + lo, F.h_out[c % F.S] + (lo - c * C), (hi - lo) * 4);
				copied[c].fetch_add(1, std::memory_order_release);
			}
		}
		} catch (...) { worker_threw = true; abort = true; }
	};
	// the tasks come from one counter, so the pipeline works with however many workers could be started (a pid-limited
	// container may refuse some of up to 32); with none at all the calling thread cannot both pack and drive: an error
	std::vector<std::thread> th;
	try { th.reserve((size_t)T); } catch (...) { return fail(KMX_E_NOMEM, "out of memory"); }
	for (int t = 0; t < T; t++) {
		try { th.emplace_back(worker, t); } catch (...) { break; }
	}
	if (th.empty()) return fail(KMX_E_NOMEM, "cannot start a worker thread for the query pipeline");
	int rc = KMX_OK;
	auto publish = [&](u64 c) {                                   // chunk c's answers are in its pinned slot
		hipError_t e;
		while ((e = hipEventQuery(F.ev_out[c % F.S])) == hipErrorNotReady) std::this_thread::yield();
		if (e != hipSuccess && !rc) rc = fail(KMX_E_NODEVICE, "query failed");
		if (rc) abort = true;
		ready[c].store(1, std::memory_order_release);
	};
	for (u64 c = 0; c < nc && !rc; c++) {
		if (!wait_for([&] { return packed[c].load(std::memory_order_acquire) == subs_of(c); })) { rc = fail(KMX_E_NOMEM, "out of memory in a query worker"); break; }
		if (!enqueue(c)) { rc = fail(KMX_E_NODEVICE, "query pipeline: enqueue failed"); break; }
		if (c >= 1) publish(c - 1);
	}
	if (!rc) publish(nc - 1);
	else abort = true;
	for (auto &x : th) x.join();
	hipStreamSynchronize(F.to_dev); hipStreamSynchronize(F.to_host);
	if (worker_threw && !rc) rc = fail(KMX_E_NOMEM, "out of memory in a query worker");
	if (hipStreamSynchronize(m->stream) != hipSuccess && !rc) rc = fail(KMX_E_NODEVICE, "query failed");
	if (!rc && hipGetLastError() != hipSuccess) rc = fail(KMX_E_NODEVICE, "query kernel failed");
	return rc;
}

static int query_text(kmx_model *m, const KmxStrBatch &sb, uint64_t n, int32_t *out)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_READY) return fail(KMX_E_STATE, "query before the model is built or loaded");
	const int len = sb.len, W = m->W;
	if (len < 2 || len > 64 || sb.stride < len) return fail(KMX_E_ARG, "k-mer strings must hold 2..64 characters (got %d, stride %d)", len, sb.stride);
	if (!n) return KMX_OK;
	HIPCHK(hipSetDevice(m->device));
	const int T = (int)std::max<u64>(1, std::min<u64>(std::min(kmx_host_cpus(), 32), n / 8192 + 1));
	auto bytes_pass = [&](const KmxStrBatch &b, u64 cnt, int32_t *dst) {   // the strings as they are -> k_query_ascii
		return query_pipeline(m, cnt, (size_t)len, (int)std::max<u64>(1, std::min<u64>(T, cnt / 8192 + 1)),
		                      [&](int, u64 lo, u64 hi, unsigned char *d) { kmx_gather_strings(b, lo, hi, d); },
		                      [&](int s, u64 cn) { kmxk::query_ascii(m->md, len, m->qfeed.d_in[s], len, cn, m->qfeed.d_out[s], m->stream); }, dst);
	};
	if (len != m->k) return bytes_pass(sb, n, out);              // (another length hashes differently from any k-mer of the model)
	std::vector<std::vector<uint64_t>> dirty((size_t)T);
	TRY(query_pipeline(m, n, (size_t)W * 8, T,
	                   [&](int t, u64 lo, u64 hi, unsigned char *d) { kmx_pack_strings(sb, W, lo, hi, (uint64_t *)d, &dirty[(size_t)t]); },
	                   [&](int s, u64 cn) { kmxk::query(m->md, (const u64 *)m->qfeed.d_in[s], cn, m->qfeed.d_out[s], m->stream, &m->prof); }, out));
	std::vector<const char *> dptr;
	std::vector<uint64_t> didx;
	for (auto &v : dirty) for (uint64_t i : v) { didx.push_back(i); dptr.push_back(sb.ptrs ? sb.ptrs[i] : sb.flat + i * (uint64_t)sb.stride); }
	if (didx.empty()) return KMX_OK;
	std::vector<int32_t> ans(didx.size());
	TRY(bytes_pass(KmxStrBatch{dptr.data(), nullptr, len, len}, (u64)didx.size(), ans.data()));
	for (size_t j = 0; j < didx.size(); j++) out[didx[j]] = ans[j];
	return KMX_OK;
}

static int kmx_query_ascii_impl(kmx_model *m, const char *strs, int len, int stride, uint64_t n, int32_t *out)
{
	if (n && (!strs || !out)) return fail(KMX_E_ARG, "null argument");
	return query_text(m, KmxStrBatch{nullptr, strs, stride, len}, n, out);
}

// the same for n separate strings of `len` characters each (what a vector<string> holds), without concatenating them
static int kmx_query_strings_impl(kmx_model *m, const char *const *strs, int len, uint64_t n, int32_t *out)
{
	if (n && (!strs || !out)) return fail(KMX_E_ARG, "null argument");
	return query_text(m, KmxStrBatch{strs, nullptr, len, len}, n, out);
}

// ------------------------------------------------------------------------------------------ persistence
static int download_array(kmx_model *m, int which, int index, std::vector<unsigned char> &out)
{
	out.clear();
	if (which == 0 || which == 1 || which == 2) {
		if (which != 2 && (index < 0 || index >= m->bf_num)) return fail(KMX_E_ARG, "bad filter index");
		const u64 nbytes = which == 0 ? m->byte_bf[index] : which == 1 ? m->byte_bf_back[index] : m->byte_km_back;
		const u32 *src = which == 0 ? m->d_bf[index] : which == 1 ? m->d_bf_back[index] : m->d_km_back;
		out.resize(nbytes);
		if (nbytes) HIPCHK(hipMemcpy(out.data(), src, nbytes, hipMemcpyDeviceToHost));
		return KMX_OK;
	}
	if (which < 3 || which > 5 || index < 0 || index >= m->nb) return fail(KMX_E_ARG, "bad array selector");
	const u64 nbytes = m->km_byte_size;
	out.resize(nbytes);
	if (!nbytes) return KMX_OK;
	DevMem tmp;
	HIPCHK(tmp.alloc(m->ncells * 2));
	kmxk::cells_to_disk(m->d_cells[index], m->ncells, nbytes, which - 3, tmp.as<unsigned char>(), m->stream);
	hipError_t e = hipMemcpyAsync(out.data(), tmp.p, nbytes, hipMemcpyDeviceToHost, m->stream);
	if (hipStreamSynchronize(m->stream) != hipSuccess || e != hipSuccess) return fail(KMX_E_NODEVICE, "D2H copy failed");
	return KMX_OK;
}

static int kmx_download_impl(kmx_model *m, int which, int index, uint8_t *dst, uint64_t capacity, uint64_t *written)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state == ST_EMPTY) return fail(KMX_E_STATE, "no arrays yet");
	HIPCHK(hipSetDevice(m->device));
	HIPCHK(hipStreamSynchronize(m->stream));
	std::vector<unsigned char> v;
	TRY(download_array(m, which, index, v));
	if (written) *written = v.size();
	if (v.size() > capacity) return fail(KMX_E_ARG, "buffer too small: need %llu bytes", (unsigned long long)v.size());
	if (!v.empty()) memcpy(dst, v.data(), v.size());
	return KMX_OK;
}

// km.bin (Appendix B.1): u64 n_km, u64 n_bf[bf_num], then bf / bf_back per filter, km_back, and bit_array_1 (value) /
// bit_array_2 (tag) per coupled array.  Streamed: the model's stream de-interleaves one array at a time and copies it
// into a ring of pinned chunks; writer threads pwrite the chunks at their final offsets, so the copy, the page-cache
// writes and the next array's kernel overlap.
namespace {
struct SaveChunk {
	unsigned char *buf = nullptr;
	hipEvent_t ready = nullptr;
	u64 len = 0, off = 0;
	int state = 0;                                              // 0 free, 1 filled (copy enqueued), 2 being written
};
}

static int save_km_bin(kmx_model *m, const std::string &path)
{
	const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (fd < 0) return fail(KMX_E_IO, "cannot write %s", path.c_str());
	u64 head[4] = {m->n_km, 0, 0, 0};
	for (int i = 0; i < m->bf_num; i++) head[1 + i] = m->n_bf[i];
	const u64 head_bytes = 8 * (1 + (u64)m->bf_num);
	bool io_ok = pwrite(fd, head, head_bytes, 0) == (ssize_t)head_bytes;
	constexpr int R = 6, T = 4;
	constexpr u64 CH = 16ull << 20;
	SaveChunk ring[R];
	unsigned char *tmp = nullptr;
	int rc = KMX_OK;
	for (auto &c : ring)
		if (hipHostMalloc((void **)&c.buf, CH) != hipSuccess || hipEventCreateWithFlags(&c.ready, hipEventDisableTiming) != hipSuccess) rc = fail(KMX_E_NOMEM, "pinned buffers for save could not be allocated");
	if (!rc && m->km_byte_size && hipMalloc((void **)&tmp, m->ncells * 2) != hipSuccess) rc = fail(KMX_E_NOMEM, "device buffer for save could not be allocated");
	std::mutex mu;
	std::condition_variable cv;
	bool done = false;
	std::vector<std::thread> writers;
	if (!rc)
		for (int t = 0; t < T; t++)
			writers.emplace_back([&] {
				for (;;) {
					SaveChunk *c = nullptr;
					{
						std::unique_lock<std::mutex> lk(mu);
						cv.wait(lk, [&] {
							for (auto &x : ring) if (x.state == 1) { c = &x; return true; }
							return done;
						});
						if (!c) return;
						c->state = 2;
					}
					bool ok = hipEventSynchronize(c->ready) == hipSuccess;
					for (u64 w = 0; ok && w < c->len;) {
						const ssize_t k = pwrite(fd, c->buf + w, c->len - w, (off_t)(c->off + w));
						if (k <= 0) ok = false; else w += (u64)k;
					}
					{ std::lock_guard<std::mutex> lk(mu); c->state = 0; if (!ok) io_ok = false; }
					cv.notify_all();
				}
			});
	u64 off = head_bytes;
	auto emit = [&](const unsigned char *dsrc, u64 nbytes) {                 // device bytes -> file at `off`
		for (u64 pos = 0; pos < nbytes && !rc; pos += CH) {
			const u64 len = std::min<u64>(CH, nbytes - pos);
			SaveChunk *c = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { for (auto &x : ring) if (x.state == 0) { c = &x; return true; } return false; });
			}
			if (hipMemcpyAsync(c->buf, dsrc + pos, len, hipMemcpyDeviceToHost, m->stream) != hipSuccess || hipEventRecord(c->ready, m->stream) != hipSuccess) { rc = fail(KMX_E_NODEVICE, "D2H copy failed"); break; }
			c->len = len; c->off = off + pos;
			{ std::lock_guard<std::mutex> lk(mu); c->state = 1; }
			cv.notify_all();
		}
		off += nbytes;
	};
	if (!rc) {
		for (int i = 0; i < m->bf_num; i++) { emit((const unsigned char *)m->d_bf[i], m->byte_bf[i]); emit((const unsigned char *)m->d_bf_back[i], m->byte_bf_back[i]); }
		emit((const unsigned char *)m->d_km_back, m->byte_km_back);
		for (int a = 0; a < m->nb && !rc && m->km_byte_size; a++)
			for (int which = 0; which < 2; which++) {                          // bit_array_1 (value), bit_array_2 (tag)
				kmxk::cells_to_disk(m->d_cells[a], m->ncells, m->km_byte_size, which, tmp, m->stream);   // waits, in stream order, for the copies out of tmp
				emit(tmp, m->km_byte_size);
			}
	}
	{ std::lock_guard<std::mutex> lk(mu); done = true; }
	cv.notify_all();
	// the writers drain what is filled before they see `done` with nothing left
	for (auto &w : writers) w.join();
	hipStreamSynchronize(m->stream);
	for (auto &c : ring) { if (c.buf) hipHostFree(c.buf); if (c.ready) hipEventDestroy(c.ready); }
	hipFree(tmp);
	if (close(fd) != 0) io_ok = false;
	if (!rc && !io_ok) rc = fail(KMX_E_IO, "short write to %s", path.c_str());
	return rc;
}

// KModel::save (kmodel.hpp:173-206) + KRestData::save_file (rest.hpp:197-221); layouts: Appendix B.1/B.2
static int kmx_save_impl(kmx_model *m, const char *dir)
{
	if (!m || !dir) return fail(KMX_E_ARG, "null argument");
	if (m->state != ST_READY) return fail(KMX_E_STATE, "save before the model is built or loaded");
	HIPCHK(hipSetDevice(m->device));
	HIPCHK(hipStreamSynchronize(m->stream));
	std::string d(dir);
	FILE *f = fopen((d + "/header").c_str(), "w");
	if (!f) return fail(KMX_E_IO, "cannot write %s/header", dir);
	const bool header_ok = fprintf(f, "number_hash %d\nnumber_bit %d\nci %d\ncs %d\n", m->nh, m->nb, m->ci, m->cs) > 0;
	if ((fclose(f) != 0) | !header_ok) return fail(KMX_E_IO, "short write to %s/header", dir);
	const auto t0 = std::chrono::steady_clock::now();
	TRY(save_km_bin(m, d + "/km.bin"));
	const auto t1 = std::chrono::steady_clock::now();
	TRY(rest_materialize_host(m));
	const auto t2 = std::chrono::steady_clock::now();
	const RestTable &r = m->rest;
	if (!(f = fopen((d + "/rest.bin").c_str(), "wb"))) return fail(KMX_E_IO, "cannot write %s/rest.bin", dir);
	int h[4] = {r.k, r.pre_len, r.map_size, r.pre_buffer_size};
	bool w_ok = fwrite(h, 4, 4, f) == 4 && fwrite(&r.suff_bin_size, 8, 1, f) == 1 && fwrite(&r.entries, 8, 1, f) == 1;
	w_ok = w_ok && fwrite(r.hash2index.data(), 4, r.hash2index.size(), f) == r.hash2index.size();
	w_ok = w_ok && fwrite(r.pre_buffer.data(), 4, r.pre_buffer.size(), f) == r.pre_buffer.size();
	w_ok = w_ok && fwrite(r.suffix_bin.data(), 1, r.suffix_bin.size(), f) == r.suffix_bin.size();
	w_ok = w_ok && fwrite(r.count_bin.data(), 4, r.count_bin.size(), f) == r.count_bin.size();
	if ((fclose(f) != 0) | !w_ok) return fail(KMX_E_IO, "short write to %s/rest.bin", dir);
	if (m->dbg_ctrl) {
		const auto t3 = std::chrono::steady_clock::now();
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count() * 1e3; };
		fprintf(stderr, "[kmx] save: km.bin %.1f ms, rest table to host form %.1f ms, rest.bin %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
	}
	return KMX_OK;
}

// get_model(save_dir) (kmodel.hpp:680-696) + KModel::load (:209-235) + KRestData::from_file (rest.hpp:163-195)
static int kmx_load_impl(const char *dir, kmx_model **out)
{
	if (!dir || !out) return fail(KMX_E_ARG, "null argument");
	*out = nullptr;
	std::string d(dir);
	FILE *f = fopen((d + "/header").c_str(), "r");
	if (!f) return fail(KMX_E_IO, "load_model: cant't open the header of the model ! (%s/header)", dir);
	char key[64];
	int nh, nb, ci, cs;
	int got = fscanf(f, "%63s %d %63s %d %63s %d %63s %d", key, &nh, key, &nb, key, &ci, key, &cs);
	fclose(f);
	if (got != 8) return fail(KMX_E_IO, "malformed header in %s", dir);
	kmx_model *m = nullptr;
	TRY(kmx_create(ci, cs, nh, nb, &m));
	auto bail = [&](int code) { kmx_destroy(m); return code; };
	// rest.bin first: it carries k
	if (!(f = fopen((d + "/rest.bin").c_str(), "rb"))) return bail(fail(KMX_E_IO, "cannot open %s/rest.bin", dir));
	RestTable &r = m->rest;
	int h[4];
	bool ok = fread(h, 4, 4, f) == 4 && fread(&r.suff_bin_size, 8, 1, f) == 1 && fread(&r.entries, 8, 1, f) == 1;
	if (ok) {
		r.k = h[0]; r.pre_len = h[1]; r.map_size = h[2]; r.pre_buffer_size = h[3];
		ok = r.k >= 3 && r.k <= 64 && r.pre_len >= 1 && r.pre_len <= 12 && r.map_size == (1 << (2 * r.pre_len)) && r.pre_buffer_size >= 1;
	}
	if (ok) {
		r.suff_group = (r.k - r.pre_len) / 4;
		ok = r.suff_bin_size == r.entries * (u64)r.suff_group;
	}
	if (ok) {
		r.hash2index.resize(r.map_size); r.pre_buffer.resize(r.pre_buffer_size);
		r.suffix_bin.resize(r.suff_bin_size); r.count_bin.resize(r.entries);
		ok = fread(r.hash2index.data(), 4, r.map_size, f) == (size_t)r.map_size &&
		     fread(r.pre_buffer.data(), 4, r.pre_buffer_size, f) == (size_t)r.pre_buffer_size &&
		     fread(r.suffix_bin.data(), 1, r.suff_bin_size, f) == r.suff_bin_size &&
		     fread(r.count_bin.data(), 4, r.entries, f) == r.entries;
	}
	fclose(f);
	if (ok) {   // the device lookup indexes with these: a corrupt file must end as KMX_E_IO, not as an out-of-bounds read
		ok = r.pre_buffer[0] == 0 && (u64)r.pre_buffer[r.pre_buffer_size - 1] == r.entries && r.pre_buffer_size - 1 <= r.map_size;
		for (int g = 1; g < r.pre_buffer_size && ok; g++) ok = r.pre_buffer[g] >= r.pre_buffer[g - 1];
		int next_group = 0;                                         // groups are numbered in ascending prefix order (rest.hpp:113-127)
		for (int q = 0; q < r.map_size && ok; q++)
			if (r.hash2index[q] != -1) ok = r.hash2index[q] == next_group++;
		ok = ok && next_group == r.pre_buffer_size - 1;
		for (u64 e = 0; e < r.entries && ok; e++) ok = r.count_bin[e] >= ci && r.count_bin[e] <= cs;
	}
	if (!ok) return bail(fail(KMX_E_IO, "malformed %s/rest.bin", dir));
	m->k = r.k; m->W = (r.k + 31) / 32;
	// km.bin is mapped and copied to the device section by section, without a staging read
	const int fd = open((d + "/km.bin").c_str(), O_RDONLY);
	if (fd < 0) return bail(fail(KMX_E_IO, "cannot open %s/km.bin", dir));
	struct stat sb;
	const unsigned char *map = nullptr;
	if (fstat(fd, &sb) == 0 && sb.st_size >= 16) map = (const unsigned char *)mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (!map || map == (const unsigned char *)MAP_FAILED) return bail(fail(KMX_E_IO, "malformed %s/km.bin", dir));
	const u64 fsize = (u64)sb.st_size;
	auto unmap = [&] { munmap((void *)map, (size_t)fsize); };
	u64 off = 0;
	memcpy(&m->n_km, map, 8); off = 8;
	u64 nbf = 0;
	ok = fsize >= 8 * (1 + (u64)m->bf_num);
	for (int i = 0; i < m->bf_num && ok; i++) { memcpy(&m->n_bf[i], map + off, 8); off += 8; nbf += m->n_bf[i]; }
	if (!ok) { unmap(); return bail(fail(KMX_E_IO, "malformed %s/km.bin", dir)); }
	m->n_total = m->n_km + nbf;
	compute_sizes(m);
	{
		u64 need = off + m->byte_km_back + 2 * m->km_byte_size * (u64)m->nb;
		for (int i = 0; i < m->bf_num; i++) need += m->byte_bf[i] + m->byte_bf_back[i];
		if (fsize < need) { unmap(); return bail(fail(KMX_E_IO, "short or unreadable %s/km.bin", dir)); }
	}
	int rc = alloc_arrays(m);
	if (rc) { unmap(); return bail(rc); }
	auto upload = [&](void *dst, u64 nbytes) -> bool {
		const bool good = !nbytes || hipMemcpyAsync(dst, map + off, nbytes, hipMemcpyHostToDevice, m->stream) == hipSuccess;
		off += nbytes;
		return good;
	};
	for (int i = 0; i < m->bf_num && ok; i++) ok = upload(m->d_bf[i], m->byte_bf[i]) && upload(m->d_bf_back[i], m->byte_bf_back[i]);
	ok = ok && upload(m->d_km_back, m->byte_km_back);
	unsigned char *dv = nullptr, *dt = nullptr;
	if (ok && m->km_byte_size) {
		ok = hipMalloc((void **)&dv, m->ncells * 2) == hipSuccess && hipMalloc((void **)&dt, m->ncells * 2) == hipSuccess;
		for (int a = 0; a < m->nb && ok; a++) {
			ok = upload(dv, m->km_byte_size) && upload(dt, m->km_byte_size);
			if (ok) kmxk::cells_from_disk(dv, dt, m->km_byte_size, m->d_cells[a], m->ncells, m->stream);
		}
		ok = ok && hipStreamSynchronize(m->stream) == hipSuccess;
		hipFree(dv); hipFree(dt);
	}
	ok = ok && hipStreamSynchronize(m->stream) == hipSuccess;
	unmap();
	if (!ok) return bail(fail(KMX_E_IO, "short or unreadable %s/km.bin", dir));
	rc = rest_to_device(m);
	if (rc) return bail(rc);
	fill_model_dev(m);
	m->state = ST_READY;
	*out = m;
	return KMX_OK;
}

static int kmx_get_stats_impl(kmx_model *m, kmx_stats *st)
{
	if (!m || !st) return fail(KMX_E_ARG, "null argument");
	memset(st, 0, sizeof *st);
	st->n_total = m->n_total; st->n_km = m->n_km;
	for (int i = 0; i < 3; i++) { st->n_bf[i] = m->n_bf[i]; st->byte_bf[i] = m->byte_bf[i]; st->byte_bf_back[i] = m->byte_bf_back[i]; }
	st->attempts = m->h_stats[ST_ATTEMPTS]; st->successes = m->h_stats[ST_SUCCESSES];
	st->fast_commits = m->h_stats[ST_SUCCESSES] - m->h_stats[ST_SLOW_SUCC]; st->contended = m->h_stats[ST_CONTENDED]; st->finisher_iters = m->h_stats[ST_FIN_ITERS];
	st->rest_entries = m->rest.entries; st->km_byte_size = m->km_byte_size; st->byte_km_back = m->byte_km_back;
	st->blocks = m->blocks; st->rounds = m->rounds;
	st->piped_attempts = m->h_stats[ST_PIPE_ATTEMPTS]; st->piped_commits = m->h_stats[ST_PIPE_SUCC];
	st->piped_gathers = m->h_stats[ST_PIPE_GATHERS]; st->piped_atomics = m->h_stats[ST_PIPE_ATOMICS];
	st->query_neighbour_calls = m->h_stats[ST_QUERY_NEIGH]; st->query_accounted = m->h_stats[ST_QUERY_N];
	st->rest_bytes = m->rest.suff_bin_size + 4 * m->rest.entries + 4 * (u64)m->rest.pre_buffer_size + 4 * (u64)m->rest.map_size;
	st->k = m->k; st->ci = m->ci; st->cs = m->cs; st->nh = m->nh; st->nb = m->nb; st->bf_num = m->bf_num; st->device = m->device;
	return KMX_OK;
}

// the first `size` bytes of the struct: a caller built against an older (shorter) kmx_stats names its own sizeof
static int kmx_get_stats_n_impl(kmx_model *m, void *st, uint64_t size)
{
	if (!st) return fail(KMX_E_ARG, "null argument");
	kmx_stats full;
	TRY(kmx_get_stats_impl(m, &full));
	memcpy(st, &full, (size_t)std::min<uint64_t>(size, sizeof full));
	return KMX_OK;
}

static int kmx_last_build_seconds_impl(kmx_model *m, double *insert_kernels_s, double *total_s)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (insert_kernels_s) *insert_kernels_s = m->t_insert_kernels;
	if (total_s) *total_s = m->t_total;
	return KMX_OK;
}

// ------------------------------------------------------------------------------------------ KAT surface / microbench
static int kmx_debug_hash_impl(int k, const uint64_t *kmers, uint64_t n, const uint32_t *seeds, int n_seeds, int whole, uint64_t *hashes)
{
	if (k < 3 || k > 64 || n_seeds < 1) return fail(KMX_E_ARG, "bad arguments");
	const int W = (k + 31) / 32;
	DevMem dk, dh, ds;
	HIPCHK(dk.alloc(n * W * 8 + 8));
	HIPCHK(dh.alloc(n * n_seeds * 8 + 8));
	HIPCHK(ds.alloc(n_seeds * 4));
	HIPCHK(hipMemcpy(dk.p, kmers, n * W * 8, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(ds.p, seeds, n_seeds * 4, hipMemcpyHostToDevice));
	kmxk::debug_hash(k, dk.as<u64>(), n, ds.as<u32>(), n_seeds, whole, dh.as<u64>(), nullptr);
	HIPCHK(hipMemcpy(hashes, dh.p, n * n_seeds * 8, hipMemcpyDeviceToHost));
	return KMX_OK;
}

static int kmx_debug_min_kmer_impl(int k, const uint64_t *kmers, uint64_t n, uint64_t *out)
{
	if (k < 3 || k > 64) return fail(KMX_E_ARG, "bad arguments");
	const int W = (k + 31) / 32;
	DevMem dk, dout;
	HIPCHK(dk.alloc(n * W * 8 + 8));
	HIPCHK(dout.alloc(n * W * 8 + 8));
	HIPCHK(hipMemcpy(dk.p, kmers, n * W * 8, hipMemcpyHostToDevice));
	kmxk::debug_min_kmer(k, dk.as<u64>(), n, dout.as<u64>(), nullptr);
	HIPCHK(hipMemcpy(out, dout.p, n * W * 8, hipMemcpyDeviceToHost));
	return KMX_OK;
}

static int kmx_debug_mod_impl(const uint64_t *h, uint64_t n, uint64_t d, uint64_t *out)
{
	if (!d || !n) return fail(KMX_E_ARG, "bad arguments");
	DevMem dh, dout;
	HIPCHK(dh.alloc(n * 8));
	HIPCHK(dout.alloc(n * 8));
	HIPCHK(hipMemcpy(dh.p, h, n * 8, hipMemcpyHostToDevice));
	kmxk::debug_mod(dh.as<u64>(), n, d, dout.as<u64>(), nullptr);
	HIPCHK(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
	return KMX_OK;
}

// mode 20: the 4-byte gathers (mode 8) on one stream and the 32-bit atomic ORs (mode 5) on another, side by side, each on
// its own buffer; seconds = wall time until both are done.  Compared with the sum of the two alone it says whether the
// gather-bound and the atomic-bound kernels of a round could hide behind each other.
static int microbench_pair(uint64_t bytes, uint64_t touches, int iters, double *seconds)
{
	DevMem b0, b1, sinkm;
	const u64 ncell = bytes / 8, lanes = touches / 8;
	HIPCHK(b0.alloc(ncell * 8));
	HIPCHK(b1.alloc(ncell * 8));
	HIPCHK(sinkm.alloc(8));
	HIPCHK(hipMemset(b0.as<u64>(), 0, ncell * 8));
	HIPCHK(hipMemset(b1.as<u64>(), 0, ncell * 8));
	hipStream_t s0 = nullptr, s1 = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr, j = nullptr;
	auto guard = scope_exit([&] {
		if (e0) hipEventDestroy(e0);
		if (e1) hipEventDestroy(e1);
		if (j) hipEventDestroy(j);
		if (s0) hipStreamDestroy(s0);
		if (s1) hipStreamDestroy(s1);
	});
	HIPCHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	HIPCHK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
	kmxk::micro(8, b0.as<u64>(), ncell, lanes, 12345, sinkm.as<u64>(), s0);
	kmxk::micro(5, b1.as<u64>(), ncell, lanes, 12345, sinkm.as<u64>(), s1);
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipEventRecord(e0, s0));
	HIPCHK(hipStreamWaitEvent(s1, e0, 0));
	for (int it = 0; it < iters; it++) {
		kmxk::micro(8, b0.as<u64>(), ncell, lanes, 1000003ULL * (it + 1), sinkm.as<u64>(), s0);
		kmxk::micro(5, b1.as<u64>(), ncell, lanes, 1000003ULL * (it + 1), sinkm.as<u64>(), s1);
	}
	HIPCHK(hipEventRecord(j, s1));
	HIPCHK(hipStreamWaitEvent(s0, j, 0));
	HIPCHK(hipEventRecord(e1, s0));
	HIPCHK(hipEventSynchronize(e1));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	*seconds = ms * 1e-3 / iters;
	return KMX_OK;
}

static int kmx_microbench_impl(int mode, uint64_t bytes, uint64_t touches, int iters, double *seconds)
{
	if (bytes < 4096 || touches < 8 || iters < 1 || !seconds) return fail(KMX_E_ARG, "bad arguments");
	if (mode == 20) return microbench_pair(bytes, touches, iters, seconds);
	DevMem bufm, sinkm;
	const u64 ncell = bytes / 8, lanes = touches / 8;
	HIPCHK(bufm.alloc(ncell * 8));
	HIPCHK(sinkm.alloc(8));
	u64 *buf = bufm.as<u64>(), *sink = sinkm.as<u64>();
	HIPCHK(hipMemset(buf, 0, ncell * 8));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	auto ev_guard = scope_exit([&] { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); });
	HIPCHK(hipEventCreate(&e0));
	HIPCHK(hipEventCreate(&e1));
	kmxk::micro(mode, buf, ncell, lanes, 12345, sink, nullptr);            // warm-up
	HIPCHK(hipEventRecord(e0, nullptr));
	for (int it = 0; it < iters; it++) kmxk::micro(mode, buf, ncell, lanes, 1000003ULL * (it + 1), sink, nullptr);
	HIPCHK(hipEventRecord(e1, nullptr));
	HIPCHK(hipEventSynchronize(e1));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	*seconds = ms * 1e-3 / iters;
	return KMX_OK;
}

// ------------------------------------------------------------------------------------------ kernel-class timing
static int kmx_set_profile_impl(kmx_model *m, int on)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (on < 0 || on > 2) return fail(KMX_E_ARG, "kmx_set_profile: 0 (off), 1 (timing) or 2 (accounting), got %d", on);
	m->prof.on = on == 1;                                         // 1: per-class timing of the product's kernels
	m->prof.count = on == 2;                                      // 2: no timing; the fused launches run their accounting variant (kmx_stats piped_*)
	return KMX_OK;
}

static int kmx_get_kernel_times_impl(kmx_model *m, double *seconds, uint64_t *launches, int reset)
{
	if (!m || !seconds || !launches) return fail(KMX_E_ARG, "null argument");
	HIPCHK(hipSetDevice(m->device));
	HIPCHK(hipStreamSynchronize(m->stream));
	prof_collect(m);
	for (int c = 0; c < KC_N; c++) { seconds[c] = m->kc_seconds[c]; launches[c] = m->kc_launches[c]; }
	if (reset) for (int c = 0; c < KC_N; c++) { m->kc_seconds[c] = 0; m->kc_launches[c] = 0; }
	return KMX_OK;
}

// ------------------------------------------------------------------------------------------ exception firewall
// Nothing may unwind through the C ABI: every entry point runs its implementation inside this guard.
template <typename F> static int guarded(F f)
{
	try { return f(); }
	catch (const std::bad_alloc &) { return fail(KMX_E_NOMEM, "out of host memory"); }
	catch (const std::exception &e) { return fail(KMX_E_ARG, "internal error: %s", e.what()); }
	catch (...) { return fail(KMX_E_ARG, "internal error"); }
}

extern "C" int kmx_device_count(void) { return guarded([&] { return kmx_device_count_impl(); }); }
extern "C" int kmx_occubin(int cs, int nh, uint32_t *bin_of_occ, uint32_t *mean_of_bin) { return guarded([&] { return kmx_occubin_impl(cs, nh, bin_of_occ, mean_of_bin); }); }
extern "C" int kmx_create(int ci, int cs, int nh, int nb, kmx_model **out) { return guarded([&] { return kmx_create_impl(ci, cs, nh, nb, out); }); }
extern "C" int kmx_destroy(kmx_model *m) { return guarded([&] { return kmx_destroy_impl(m); }); }
extern "C" int kmx_set_stream(kmx_model *m, void *s) { return guarded([&] { return kmx_set_stream_impl(m, s); }); }
extern "C" int kmx_begin(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total) { return guarded([&] { return kmx_begin_impl(m, k, n_bf, n_total); }); }
extern "C" int kmx_insert_batch_dev(kmx_model *m, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n) { return guarded([&] { return kmx_insert_batch_dev_impl(m, d_kmers, d_counts, n); }); }
extern "C" int kmx_insert_batch(kmx_model *m, const uint64_t *kmers, const uint32_t *counts, uint64_t n) { return guarded([&] { return kmx_insert_batch_impl(m, kmers, counts, n); }); }
extern "C" int kmx_finish(kmx_model *m) { return guarded([&] { return kmx_finish_impl(m); }); }
extern "C" int kmx_build_dev(kmx_model *m, int k, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n) { return guarded([&] { return kmx_build_dev_impl(m, k, d_kmers, d_counts, n); }); }
extern "C" int kmx_build_host(kmx_model *m, int k, const uint64_t *kmers, const uint32_t *counts, uint64_t n) { return guarded([&] { return kmx_build_host_impl(m, k, kmers, counts, n); }); }
extern "C" int kmx_build_from_kmc(kmx_model *m, const char *db_prefix) { return guarded([&] { return kmx_build_from_kmc_impl(m, db_prefix); }); }
extern "C" int kmx_count_classes_dev(kmx_model *m, const uint32_t *d_counts, uint64_t n, uint64_t n_bf[3]) { return guarded([&] { return kmx_count_classes_dev_impl(m, d_counts, n, n_bf); }); }
extern "C" int kmx_shard_begin(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world) { return guarded([&] { return kmx_shard_begin_impl(m, k, n_bf, n_total, rank, world); }); }
extern "C" int kmx_shard_classify_dev(kmx_model *m, const uint64_t *d_kmers, const uint32_t *d_counts, uint64_t n, uint64_t *d_out_kmers, uint32_t *d_out_counts, uint64_t *n_out) { return guarded([&] { return kmx_shard_classify_dev_impl(m, d_kmers, d_counts, n, d_out_kmers, d_out_counts, n_out); }); }
extern "C" uint64_t kmx_ring_msg_bytes(int k) { return ring_msg_bytes(k); }
extern "C" int kmx_ring_round_dev(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists) { return guarded([&] { return kmx_ring_round_dev_impl(m, t, lists, n_lists); }); }
extern "C" int kmx_ring_stale_dup_dev(kmx_model *m, int first_unused_row) { return guarded([&] { return kmx_ring_stale_dup_dev_impl(m, first_unused_row); }); }
extern "C" int kmx_shard_local(kmx_model *m, kmx_stats *partial, void **d_rest_kmers, void **d_rest_counts) { return guarded([&] { return kmx_shard_local_impl(m, partial, d_rest_kmers, d_rest_counts); }); }
extern "C" int kmx_shard_complete(kmx_model *m, const uint64_t *d_rest_kmers, const int32_t *d_rest_counts, uint64_t n_rest, const kmx_stats *totals) { return guarded([&] { return kmx_shard_complete_impl(m, d_rest_kmers, d_rest_counts, n_rest, totals); }); }
extern "C" int kmx_create_on(int device, int ci, int cs, int nh, int nb, kmx_model **out) { return guarded([&] { return kmx_create_on_impl(device, ci, cs, nh, nb, out); }); }
extern "C" int kmx_build_from_kmc_multi(kmx_model **models, int n_models, const char *db_prefix) { return guarded([&] { return kmx_build_from_kmc_multi_ex_impl(models, n_models, db_prefix, KMX_PARTITION_RING); }); }
extern "C" int kmx_build_from_kmc_multi_ex(kmx_model **models, int n_models, const char *db_prefix, int partition) { return guarded([&] { return kmx_build_from_kmc_multi_ex_impl(models, n_models, db_prefix, partition); }); }
extern "C" int kmx_range_begin(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world) { return guarded([&] { return kmx_range_begin_impl(m, k, n_bf, n_total, rank, world); }); }
extern "C" int kmx_range_buffers(kmx_model *m, void **d_send, uint64_t *cap_words, uint64_t *cell_lo) { return guarded([&] { return kmx_range_buffers_impl(m, d_send, cap_words, cell_lo); }); }
extern "C" int kmx_range_emit_dev(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists, uint64_t *counts) { return guarded([&] { return kmx_range_emit_dev_impl(m, t, lists, n_lists, counts); }); }
extern "C" int kmx_range_verdict_dev(kmx_model *m, int t, const uint64_t *d_words, const uint64_t *totals, const uint64_t *commits, int n_src, uint8_t *d_verdict) { return guarded([&] { return kmx_range_verdict_dev_impl(m, t, d_words, totals, commits, n_src, d_verdict); }); }
extern "C" int kmx_range_resolve_dev(kmx_model *m, int t, const uint8_t *d_verdict) { return guarded([&] { return kmx_range_resolve_dev_impl(m, t, d_verdict); }); }
extern "C" int kmx_range_inband(kmx_model *m, void **d_send, uint64_t *region_words, uint64_t *capx_words) { return guarded([&] { return kmx_range_inband_impl(m, d_send, region_words, capx_words); }); }
extern "C" int kmx_range_verdict_inband_dev(kmx_model *m, int t, const uint64_t *d_recv, int n_src, uint8_t *d_verdict) { return guarded([&] { return kmx_range_verdict_inband_dev_impl(m, t, d_recv, n_src, d_verdict); }); }
extern "C" int kmx_range_commit_inband_dev(kmx_model *m, const uint64_t *d_recv, int n_src) { return guarded([&] { return kmx_range_commit_inband_dev_impl(m, d_recv, n_src); }); }
extern "C" int kmx_range_commit_dev(kmx_model *m, const uint64_t *d_commits, uint64_t n) { return guarded([&] { return kmx_range_commit_dev_impl(m, d_commits, n); }); }
extern "C" int kmx_range_flush_dev(kmx_model *m, uint64_t *counts) { return guarded([&] { return kmx_range_flush_dev_impl(m, counts); }); }
extern "C" int kmx_dev_view(kmx_model *m, int which, int index, void **ptr, uint64_t *bytes) { return guarded([&] { return kmx_dev_view_impl(m, which, index, ptr, bytes); }); }
extern "C" int kmx_or_words_dev(kmx_model *m, void *d_dst, const void *d_src, uint64_t n_words) { return guarded([&] { return kmx_or_words_dev_impl(m, d_dst, d_src, n_words); }); }
extern "C" int kmx_kmc_info(const char *db_prefix, int *k, uint64_t *total_kmers) { return guarded([&] { return kmx_kmc_info_impl(db_prefix, k, total_kmers); }); }
extern "C" int kmx_kmc_read(const char *db_prefix, uint64_t *kmers, uint32_t *counts, uint64_t capacity, uint64_t *n_read) { return guarded([&] { return kmx_kmc_read_impl(db_prefix, kmers, counts, capacity, n_read); }); }
extern "C" int kmx_query_packed_dev(kmx_model *m, const uint64_t *d_kmers, uint64_t n, int32_t *d_out) { return guarded([&] { return kmx_query_packed_dev_impl(m, d_kmers, n, d_out); }); }
extern "C" int kmx_query_packed(kmx_model *m, const uint64_t *kmers, uint64_t n, int32_t *out) { return guarded([&] { return kmx_query_packed_impl(m, kmers, n, out); }); }
extern "C" int kmx_query_ascii(kmx_model *m, const char *strs, int len, int stride, uint64_t n, int32_t *out) { return guarded([&] { return kmx_query_ascii_impl(m, strs, len, stride, n, out); }); }
extern "C" int kmx_query_strings(kmx_model *m, const char *const *strs, int len, uint64_t n, int32_t *out) { return guarded([&] { return kmx_query_strings_impl(m, strs, len, n, out); }); }
extern "C" int kmx_download(kmx_model *m, int which, int index, uint8_t *dst, uint64_t capacity, uint64_t *written) { return guarded([&] { return kmx_download_impl(m, which, index, dst, capacity, written); }); }
extern "C" int kmx_save(kmx_model *m, const char *dir) { return guarded([&] { return kmx_save_impl(m, dir); }); }
extern "C" int kmx_load(const char *dir, kmx_model **out) { return guarded([&] { return kmx_load_impl(dir, out); }); }
extern "C" int kmx_get_stats(kmx_model *m, kmx_stats *st) { return guarded([&] { return kmx_get_stats_impl(m, st); }); }
extern "C" int kmx_last_build_seconds(kmx_model *m, double *insert_kernels_s, double *total_s) { return guarded([&] { return kmx_last_build_seconds_impl(m, insert_kernels_s, total_s); }); }
extern "C" int kmx_debug_hash(int k, const uint64_t *kmers, uint64_t n, const uint32_t *seeds, int n_seeds, int whole, uint64_t *hashes) { return guarded([&] { return kmx_debug_hash_impl(k, kmers, n, seeds, n_seeds, whole, hashes); }); }
extern "C" int kmx_debug_min_kmer(int k, const uint64_t *kmers, uint64_t n, uint64_t *out) { return guarded([&] { return kmx_debug_min_kmer_impl(k, kmers, n, out); }); }
extern "C" int kmx_debug_mod(const uint64_t *h, uint64_t n, uint64_t d, uint64_t *out) { return guarded([&] { return kmx_debug_mod_impl(h, n, d, out); }); }
extern "C" int kmx_debug_pack_strings(const char *const *strs, const char *flat, int len, int stride, uint64_t n, uint64_t *packed, int *clean)
{
	return guarded([&] {
		if (len < 2 || len > 64 || (!strs && (!flat || stride < len)) || !packed || !clean) return fail(KMX_E_ARG, "bad argument");
		*clean = kmx_pack_strings(KmxStrBatch{strs, flat, strs ? len : stride, len}, (len + 31) / 32, 0, n, packed) ? 1 : 0;
		return (int)KMX_OK;
	});
}
extern "C" int kmx_kernel_classes(void) { return KMX_KERNEL_CLASSES; }
extern "C" int kmx_abi_version(void) { return KMX_ABI_VERSION; }
extern "C" int kmx_get_stats_n(kmx_model *m, void *st, uint64_t size) { return guarded([&] { return kmx_get_stats_n_impl(m, st, size); }); }
extern "C" int kmx_microbench(int mode, uint64_t bytes, uint64_t touches, int iters, double *seconds) { return guarded([&] { return kmx_microbench_impl(mode, bytes, touches, iters, seconds); }); }
extern "C" int kmx_set_profile(kmx_model *m, int on) { return guarded([&] { return kmx_set_profile_impl(m, on); }); }
static_assert(KC_N == KMX_KERNEL_CLASSES, "include/kmx.h promises KMX_KERNEL_CLASSES entries");
extern "C" int kmx_get_kernel_times(kmx_model *m, double *seconds, uint64_t *launches, int reset) { return guarded([&] { return kmx_get_kernel_times_impl(m, seconds, launches, reset); }); }
