// multi_build.h -- KModel::init(db) by several GPUs from ONE process (kmx_build_from_kmc_multi_ex): a host thread per handle; the
// ring of whole arrays, or the north star's position-range partition through peer-mapped inboxes or RCCL messages.  Host code;
// included by kmx_api.hip (one translation unit: it uses the handle, the error plumbing and the per-rank entry points there).
#pragma once

// ------------------------------------------------------------------------------------------ KModel::init(db) on several GPUs, from C++
// The ring of whole arrays (above) driven from ONE process: a host thread per device, the hand-offs of the ring as
// hipMemcpyPeerAsync between the devices' own streams (events order them), the merges as peer copies -- no Python, no
// torch.distributed.  A caller of the reference's API (main.cpp:143-149 -> KModel::init) reaches it through
// include/kmodel.hpp with KMX_DEVICES=0,1,...  `hs[d]` was created on the device it is to use (kmx_create_on); on return
// EVERY handle holds the whole model (replicas, as after dist.build_sharded).  Devices may repeat (several handles on one
// GPU: how the one-GPU pool tests the protocol).
#define KMX_MAX_RANKS_ANY 64                     // handles of one multi-GPU build (the ring takes any number; the range partition KMX_MAX_RANKS)
#ifndef KMX_RANGE_OVERLAP_DEFAULT
#define KMX_RANGE_OVERLAP_DEFAULT 0
#endif
namespace {
// The host threads only ENQUEUE (nobody waits for a device between two barriers of a round), so a barrier is crossed within
// microseconds: spin, and give the core away only when there are more threads than cores.
struct HostBarrier {
	std::atomic<int> waiting{0};
	std::atomic<unsigned> gen{0};
	int n;
	explicit HostBarrier(int n_) : n(n_) {}
	void wait()
	{
		const unsigned g = gen.load(std::memory_order_acquire);
		if (waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { waiting.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_release); return; }
		for (int spins = 0; gen.load(std::memory_order_acquire) == g; spins++) { if (spins > 4000) std::this_thread::yield(); else __builtin_ia32_pause(); }
	}
};
}   // namespace

static int kmx_dev_view_impl(kmx_model *m, int which, int index, void **ptr, uint64_t *bytes);
static int owner_of_array_ring(int a, int nb, int world) { return a * std::min(world, nb) / nb; }   // kmcex_amd/dist.py owner_of_array

static int range_begin_common(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world, bool mailbox);
static int range_link(kmx_model **hs, int P, int d);
// the calling thread's current device `from` may map the memory of device `to` (hipDeviceEnablePeerAccess, once per pair)
static int peer_access(int from, int to)
{
	if (from == to) return KMX_OK;
	int can = 0;
	HIPCHK(hipDeviceCanAccessPeer(&can, from, to));
	if (!can) return fail(KMX_E_NODEVICE, "device %d cannot map the memory of device %d: no peer access", from, to);
	const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
	if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(KMX_E_NODEVICE, "hipDeviceEnablePeerAccess(%d): %s", to, hipGetErrorString(e));
	(void)hipGetLastError();
	return KMX_OK;
}
static int kmx_range_inband_impl(kmx_model *m, void **d_send, uint64_t *region_words, uint64_t *capx_words);
static int range_in_inband(kmx_model *m, const uint64_t *d_recv, int n_src, uint8_t *d_verdict, RangeIn &in);
namespace {
// RCCL, bound at run time: libkmx.so does not link it (a one-GPU host needs no collective library), the range partition's RCCL
// transport opens it on demand
struct Rccl {
	void *lib = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::vector<int> devs;                                           // the devices the communicators below were made for
	std::vector<ncclComm_t> comms;
	bool load(std::string &why)
	{
		if (lib) return true;
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (lib) break;
		}
		if (!lib) { why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return false; }
		auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) why = std::string("librccl.so lacks ") + n; return p; };
		CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
		CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
		Send = (decltype(Send))sym("ncclSend");
		Recv = (decltype(Recv))sym("ncclRecv");
		GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
		GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
		GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
		if (CommInitAll && CommDestroy && Send && Recv && GroupStart && GroupEnd && GetErrorString) return true;
		dlclose(lib); lib = nullptr;
		return false;
	}
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;
}   // namespace
static int range_list_emit(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists);
static int range_owner_round(kmx_model *m, int t, const RangeIn &in, int commits);
static int range_list_apply(kmx_model *m, int t, bool seal_bulk = false);
static int range_list_order(kmx_model *m, int t, bool seal_all = false);
static int kmx_build_from_kmc_multi_ex_impl(kmx_model **hs, int P, const char *db_prefix, int partition)
{
	if (!hs || !db_prefix || P < 1) return fail(KMX_E_ARG, "null argument");
	if (partition != KMX_PARTITION_RING && partition != KMX_PARTITION_RANGE && partition != KMX_PARTITION_RANGE_RCCL) return fail(KMX_E_ARG, "partition %d: KMX_PARTITION_RING, KMX_PARTITION_RANGE or KMX_PARTITION_RANGE_RCCL", partition);
	const bool by_range = partition != KMX_PARTITION_RING, over_rccl = partition == KMX_PARTITION_RANGE_RCCL;
	// ---- the RCCL transport: one communicator per handle (ncclCommInitAll: one rank per DEVICE), fixed-size messages with in-band
	// counts, ncclSend / ncclRecv fused in a group on each handle's stream -- nothing of a round passes through the host
	// (Creating the communicators costs ~350 ms: they are kept for the next build on the same devices; one RCCL build at a time
	// per process uses them -- the lock is held for the whole build.)
	std::vector<ncclComm_t> comms;
	std::unique_lock<std::mutex> rccl_lock(g_rccl_mu, std::defer_lock);
	if (over_rccl) {
		std::vector<int> devs((size_t)P);
		for (int d = 0; d < P; d++) {
			if (!hs[d]) return fail(KMX_E_ARG, "null model");
			devs[(size_t)d] = hs[d]->device;
			for (int e = 0; e < d; e++) if (devs[(size_t)e] == devs[(size_t)d]) return fail(KMX_E_ARG, "the RCCL transport takes one handle per device (device %d appears twice): use KMX_PARTITION_RANGE there", devs[(size_t)d]);
		}
		rccl_lock.lock();
		std::string why;
		if (!g_rccl.load(why)) return fail(KMX_E_NODEVICE, "RCCL transport: %s", why.c_str());
		if (g_rccl.devs != devs) {
			for (ncclComm_t c : g_rccl.comms) if (c) g_rccl.CommDestroy(c);
			g_rccl.comms.assign((size_t)P, nullptr);
			g_rccl.devs.clear();
			const ncclResult_t rc = g_rccl.CommInitAll(g_rccl.comms.data(), P, devs.data());
			if (rc != ncclSuccess) { g_rccl.comms.clear(); return fail(KMX_E_NODEVICE, "ncclCommInitAll: %s", g_rccl.GetErrorString(rc)); }
			g_rccl.devs = devs;
		}
		comms = g_rccl.comms;
	}
	for (int d = 0; d < P; d++) {
		if (!hs[d]) return fail(KMX_E_ARG, "null model");
		if (hs[d]->ci != hs[0]->ci || hs[d]->cs != hs[0]->cs || hs[d]->nh != hs[0]->nh || hs[d]->nb != hs[0]->nb) return fail(KMX_E_ARG, "the handles of one model must share ci, cs, nh, nb");
		for (int e = 0; e < d; e++) if (hs[e] == hs[d]) return fail(KMX_E_ARG, "a handle appears twice");
	}
	if (P == 1 && !by_range) return kmx_build_from_kmc_impl(hs[0], db_prefix);      // (one handle, by range: the partition's kernels alone, every word "sent" to itself)
	kmx::KmcListing db;
	if (!db.open(db_prefix, false)) return fail(KMX_E_IO, "can't open the kmer_data_base %s: %s", db_prefix, db.error().c_str());
	const int k = (int)db.kmer_length(), W = db.words(), nb = hs[0]->nb;
	const u64 N = db.records();
	const size_t rb = db.record_bytes();
	uint64_t nbf_all[3] = {0, 0, 0};
	{   // pass 1 (kmodel.hpp:423-428) over the whole listing, on the host cores: a scan of the counter bytes, like the one-GPU init.
		// A database with unlisted records (KMC never writes one) needs the host decoder: one device builds it
		uint64_t bad = 0, not_listed = 0;
		const unsigned hw = std::thread::hardware_concurrency();
		db.count_classes((u32)hs[0]->ci, (u32)hs[0]->cs, hs[0]->bf_num, nbf_all, &bad, &not_listed, hw > 32 ? 32 : (hw > 1 ? (int)hw : 1));
		if (db.io_failed()) return fail(KMX_E_IO, "reading %s.kmc_suf failed during pass 1", db_prefix);
		if (bad) return fail(KMX_E_RANGE, "%llu k-mers with a count outside [ci=%d, cs=%d]", (unsigned long long)bad, hs[0]->ci, hs[0]->cs);
		if (not_listed) return fail(KMX_E_ARG, "the database holds records outside its header's count range: build it on one device");
		db.set_threads(std::max(1, (int)std::min(hw ? hw : 1u, 32u) / P));     // every handle's thread reads its slice with this many preads side by side
	}
	struct Rank {
		u64 *d_km = nullptr, *d_ck = nullptr, *d_rk = nullptr, *d_allk = nullptr;
		u32 *d_cnt = nullptr, *d_cc = nullptr, *d_rc = nullptr;
		int *d_allc = nullptr;
		u32 *d_tmp = nullptr;
		u64 n = 0, n_c = 0, n_r = 0, rec_lo = 0;
		std::vector<u64 *> msg;                                    // [nb * 2] list i, parity
		hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // ring: round r is enqueued (r & 1); range: the regions are sealed / the verdicts shipped / the bulk of the commits is out / ... is applied
		hipStream_t side = nullptr;                               // range: the owner's stream for the bulk of the commits
		u64 *x_recv = nullptr;                                    // RCCL transport: what the senders' regions arrive in, the verdict bytes
		unsigned char *x_ver = nullptr, *x_back = nullptr;        // this rank answers with / gets back
		u64 x_stride = 0, x_capx = 0;
		kmx_stats st;
		void *rest_k = nullptr, *rest_c = nullptr;
	};
	std::vector<Rank> R((size_t)P);
	std::vector<int> own((size_t)nb);
	for (int a = 0; a < nb; a++) own[(size_t)a] = by_range ? a % P : owner_of_array_ring(a, nb, P);   // ring: who owns array a = who meets list a first; range: who holds list a
	const u64 msg_words = ring_msg_bytes(k) / 8, blk = (u64)nb * KMX_BUCKET;
	std::atomic<int> err{0};
	std::mutex err_mu;
	std::string err_msg;
	auto note = [&](int rc) { if (rc) { std::lock_guard<std::mutex> lk(err_mu); if (!err.load()) { err = rc; err_msg = g_err; } } return rc; };
	auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess) { fail(KMX_E_NODEVICE, "%s: %s", what, hipGetErrorString(e)); note(KMX_E_NODEVICE); return false; } return true; };
	HostBarrier bar(P);
	u64 n_km = 0, n_blocks = 0, n_rest_all = 0;
	std::vector<u64> offs((size_t)P + 1, 0), rest_off((size_t)P + 1, 0);
	kmx_stats totals;
	memset(&totals, 0, sizeof totals);
	bool overflowed = false;
	auto list_len = [&](u64 b, int i) { const u64 lo = (b * (u64)nb + (u64)i) * KMX_BUCKET; return (int)std::min<u64>(n_km > lo ? n_km - lo : 0, KMX_BUCKET); };
	const bool trace = getenv("KMX_INIT_TRACE") != nullptr;          // wall-clock phase times of handle 0 on stderr; the rounds are bracketed by a stream synchronisation only when tracing
	const auto t_start = std::chrono::steady_clock::now();
	auto mark = [&](int d, const char *what) {
		if (trace && d == 0) fprintf(stderr, "[kmx multi %s P=%d] %-28s %8.2f ms\n", over_rccl ? "range-rccl" : (by_range ? "range" : "ring"), P, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
	};
	auto body = [&](int d) {
		kmx_model *m = hs[d];
		Rank &r = R[(size_t)d];
		if (!hip_ok(hipSetDevice(m->device), "hipSetDevice")) { /* keep going to the barriers */ }
		hipStream_t st = m->stream;
		// ---- this rank's slice of the listing: raw records -> the two pinned slots of the handle's feed -> device -> decoded there
		// (k_kmc_decode); the pread of batch b + 1 runs under the copy and the decode of batch b
		r.rec_lo = N * (u64)d / (u64)P;
		r.n = N * (u64)(d + 1) / (u64)P - r.rec_lo;
		if (!err) {
			const u64 B = u64(1) << 23;
			auto &F = m->feed;
			bool ok = hip_ok(hipEventCreateWithFlags(&r.ev[0], hipEventDisableTiming), "event") && hip_ok(hipEventCreateWithFlags(&r.ev[1], hipEventDisableTiming), "event") &&
			          hip_ok(hipEventCreateWithFlags(&r.ev[2], hipEventDisableTiming), "event") && hip_ok(hipEventCreateWithFlags(&r.ev[3], hipEventDisableTiming), "event") &&
			          (!by_range || hip_ok(hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking), "stream"));
			if (ok && !feed_alloc(m, (size_t)B, rb, W, db)) { fail(KMX_E_NOMEM, "pinned / device buffers for the listing feed could not be allocated"); note(KMX_E_NOMEM); ok = false; }
			if (ok) ok = hip_ok(hipMalloc((void **)&r.d_km, std::max<u64>(r.n, 1) * W * 8), "hipMalloc") && hip_ok(hipMalloc((void **)&r.d_cnt, std::max<u64>(r.n, 1) * 4), "hipMalloc") &&
			             hip_ok(hipMalloc((void **)&r.d_ck, std::max<u64>(r.n, 1) * W * 8), "hipMalloc") && hip_ok(hipMalloc((void **)&r.d_cc, std::max<u64>(r.n, 1) * 4), "hipMalloc");
			mark(d, "  buffers");
			KmcDecode kd;
			kd.lut = F.d_lut; kd.n_lut = db.lut_entries() - 1; kd.prefix_mask = db.prefix_mask();
			kd.rec_bytes = (u32)rb; kd.suf_bytes = db.suffix_bytes(); kd.cnt_bytes = db.counter_bytes();
			if (ok) ok = hip_ok(hipEventRecord(F.ev_free[0], st), "event") && hip_ok(hipEventRecord(F.ev_free[1], st), "event") &&
			             hip_ok(hipEventRecord(F.ev_copied[0], F.copy), "event") && hip_ok(hipEventRecord(F.ev_copied[1], F.copy), "event");
			int s2 = 0;
			for (u64 done = 0; ok && done < r.n; done += B, s2 ^= 1) {
				const u64 c = std::min<u64>(B, r.n - done);
				ok = hip_ok(hipEventSynchronize(F.ev_copied[s2]), "copy");        // the pinned slot has been read (two batches ago)
				if (!ok) break;
				db.copy_records(r.rec_lo + done, c, F.raw[s2]);
				ok = hip_ok(hipStreamWaitEvent(F.copy, F.ev_free[s2], 0), "wait") &&            // the decode that read the device slot two batches ago
				     hip_ok(hipMemcpyAsync(F.draw[s2], F.raw[s2], c * rb, hipMemcpyHostToDevice, F.copy), "H2D copy") &&
				     hip_ok(hipEventRecord(F.ev_copied[s2], F.copy), "event") && hip_ok(hipStreamWaitEvent(st, F.ev_copied[s2], 0), "wait");
				if (!ok) break;
				kd.recs = F.draw[s2];
				kmxk::kmc_decode(kd, W, r.rec_lo + done, c, r.d_km + done * W, r.d_cnt + done, st);
				ok = hip_ok(hipEventRecord(F.ev_free[s2], st), "event");
			}
			if (ok && db.io_failed()) { fail(KMX_E_IO, "reading %s.kmc_suf failed", db_prefix); note(KMX_E_IO); ok = false; }
			mark(d, "  slice read and enqueued");
		}
		// ---- sizes from the whole database (kmodel.hpp:402-456), front end on the slice (partial Bloom filters), the coupled class in order
		if (!err) note(by_range ? range_begin_common(m, k, nbf_all, db.kmer_count(), d, P, !over_rccl) : kmx_shard_begin_impl(m, k, nbf_all, db.kmer_count(), d, P));
		if (!err && over_rccl) {
			uint64_t stride = 0, capx = 0;
			if (!note(kmx_range_inband_impl(m, nullptr, &stride, &capx))) {
				r.x_stride = stride; r.x_capx = capx;
				hip_ok(hipMalloc((void **)&r.x_recv, (u64)P * stride * 8), "hipMalloc") && hip_ok(hipMalloc((void **)&r.x_ver, (u64)P * capx), "hipMalloc") && hip_ok(hipMalloc((void **)&r.x_back, (u64)P * capx), "hipMalloc");
			}
		}
		if (!err) { uint64_t nc = 0; if (!note(kmx_shard_classify_dev_impl(m, (const uint64_t *)r.d_km, r.d_cnt, r.n, (uint64_t *)r.d_ck, r.d_cc, &nc))) r.n_c = nc; }
		if (!err) hip_ok(hipStreamSynchronize(st), "classify");
		mark(d, "  begin + classify");
		bar.wait();
		if (d == 0) {
			for (int q = 0; q < P; q++) offs[(size_t)q + 1] = offs[(size_t)q] + R[(size_t)q].n_c;
			n_km = offs[(size_t)P];
			n_blocks = (n_km + blk - 1) / blk;
		}
		bar.wait();
		// ---- routing: every buffer of the stream to the owner of the array it meets first, in block order (dist.plan_routing)
		if (!err) {
			for (u64 b = 0; b < n_blocks; b++) for (int i = 0; i < nb; i++) if (own[(size_t)i] == d) r.n_r += (u64)list_len(b, i);
			bool ok = hip_ok(hipMalloc((void **)&r.d_rk, std::max<u64>(r.n_r, 1) * W * 8), "hipMalloc") && hip_ok(hipMalloc((void **)&r.d_rc, std::max<u64>(r.n_r, 1) * 4), "hipMalloc");
			u64 at = 0;
			for (u64 b = 0; ok && b < n_blocks; b++)
				for (int i = 0; ok && i < nb; i++) {
					if (own[(size_t)i] != d) continue;
					const u64 g0 = (b * (u64)nb + (u64)i) * KMX_BUCKET, g1 = g0 + (u64)list_len(b, i);
					for (int q = 0; ok && q < P; q++) {
						const u64 lo = std::max(g0, offs[(size_t)q]), hi = std::min(g1, offs[(size_t)q + 1]);
						if (lo >= hi) continue;
						const Rank &sr = R[(size_t)q];
						ok = hip_ok(hipMemcpyPeerAsync(r.d_rk + (at + lo - g0) * W, m->device, sr.d_ck + (lo - offs[(size_t)q]) * W, hs[q]->device, (hi - lo) * W * 8, st), "routing copy") &&
						     hip_ok(hipMemcpyPeerAsync(r.d_rc + (at + lo - g0), m->device, sr.d_cc + (lo - offs[(size_t)q]), hs[q]->device, (hi - lo) * 4, st), "routing copy");
					}
					at += g1 - g0;
				}
			if (by_range && !over_rccl) { if (ok) note(range_link(hs, P, d)); }          // (every handle allocated its inbox in range_begin, two barriers ago)
			else if (by_range) { /* RCCL: the regions stay local, ncclSend / ncclRecv move them */ }
			else {
				try { r.msg.assign((size_t)nb * 2, nullptr); } catch (...) { fail(KMX_E_NOMEM, "out of memory"); note(KMX_E_NOMEM); ok = false; }
				for (int q = 0; ok && q < P; q++) if (note(peer_access(m->device, hs[q]->device))) ok = false;      // k_ring_export stores into the next owner's buffer
				for (auto &p : r.msg) if (ok) { ok = hip_ok(hipMalloc((void **)&p, msg_words * 8), "hipMalloc") && hip_ok(hipMemsetAsync(p, 0, msg_words * 8, st), "memset"); }
			}
			if (ok) hip_ok(hipStreamSynchronize(st), "routing");
		}
		if (trace) hipStreamSynchronize(st);
		bar.wait();
		mark(d, "decode, classify, routing");
		// ---- the rounds (kmodel.hpp:560-565), by range: this rank holds the lists i = d, d + P, ... for the whole block and owns a
		// cell range of every array; the words of a round are written straight into the owners' inboxes (peer mappings), the
		// verdict bytes straight into the senders' boxes, and two events per rank order the three steps -- the host threads
		// only enqueue, nothing here waits for a device
		u64 pos = 0;
		if (over_rccl) {
			// ---- the same rounds over RCCL: every region is a fixed-size message [header | capx words] (kmx_range_inband: the counts
			// travel in band), a round is two groups of ncclSend / ncclRecv on this handle's stream -- words out, verdict bytes back --
			// and RCCL orders the devices: no event, no barrier, no number on the host.  A region that overflows (uniformly hashed
			// positions do not) voids the build; it is then repeated through the inboxes (below), which take any round whole.
			ncclComm_t comm = comms[(size_t)d];
			const u64 stride = r.x_stride, capx = r.x_capx;
			auto nccl_ok = [&](ncclResult_t rc, const char *what) { if (rc != ncclSuccess) { fail(KMX_E_NODEVICE, "%s: %s", what, g_rccl.GetErrorString(rc)); note(KMX_E_NODEVICE); return false; } return true; };
			auto exchange = [&](const void *out, void *in, u64 count, ncclDataType_t ty, u64 elem) {
				bool ok = nccl_ok(g_rccl.GroupStart(), "ncclGroupStart");
				for (int q = 0; ok && q < P; q++)
					ok = nccl_ok(g_rccl.Send((const char *)out + (u64)q * count * elem, count, ty, q, comm, st), "ncclSend") &&
					     nccl_ok(g_rccl.Recv((char *)in + (u64)q * count * elem, count, ty, q, comm, st), "ncclRecv");
				const bool ended = nccl_ok(g_rccl.GroupEnd(), "ncclGroupEnd");      // (always closed: an open group would swallow the next build's calls)
				return ok && ended;
			};
			for (u64 b = 0; b < n_blocks; b++) {
				const u64 n_in_block = std::min<u64>(blk, n_km - b * blk);
				if (!err && n_in_block < blk && b > 0) {                     // quirk Q1 (kmodel.hpp:520-527), on the rank that holds the list
					const int row = (int)((n_in_block - 1) / KMX_BUCKET);
					if (row + 1 < nb) note(kmx_ring_stale_dup_dev_impl(m, row + 1));
				}
				for (int t = 0; t < nb; t++) {
					kmx_ring_list lists[KMX_MAX_NB];
					int n_lists = 0;
					if (t == 0)
						for (int i = d; i < nb; i += P) {
							kmx_ring_list &l = lists[n_lists++];
							memset(&l, 0, sizeof l);
							l.list = i; l.n_host = list_len(b, i);
							l.src_kmers = r.d_rk + pos * W; l.src_counts = r.d_rc + pos;
							pos += (u64)l.n_host;
						}
					// (every rank makes every call of a round whatever happened to it: a rank that left out a group would hang the others)
					hip_ok(hipSetDevice(m->device), "hipSetDevice");
					if (!err) note(range_list_emit(m, t, lists, n_lists));                          // 1. [header | commits of the round before + this round's triples] per owner
					exchange(m->range.d_send, r.x_recv, stride, ncclUint64, 8);
					RangeIn in;
					if (!err && !note(range_in_inband(m, (const uint64_t *)r.x_recv, P, r.x_ver, in))) note(range_owner_round(m, t, in, RANGE_ALL));   // 2. commits applied, one verdict byte per word
					exchange(r.x_ver, r.x_back, capx, ncclUint8, 1);
					if (!err) {                                                                     // 3. winners decided; their commits go to the front of the regions
						for (int q = 0; q < P; q++) m->range.rd.vin[q] = r.x_back + (u64)q * capx;
						note(range_list_apply(m, t));
						if (!err) note(range_list_order(m, t));
					}
				}
			}
			if (n_blocks) {                                                   // the last round's commits
				if (!err) { kmxk::range_seal(m->range.rd, m->range.plan, st); m->range.pending = false; }
				exchange(m->range.d_send, r.x_recv, stride, ncclUint64, 8);
				RangeIn in;
				if (!err && !note(range_in_inband(m, (const uint64_t *)r.x_recv, P, nullptr, in))) { kmxk::range_commit_apply(m->md, in, RANGE_ALL, st); hip_ok(hipGetLastError(), "commit"); }
			}
		} else if (by_range) {
			// A round's commits can be set by the owners on a SIDE stream, beside what the list ranks do next, instead of in front of
			// the next verdicts.  KMX_RANGE_OVERLAP (under KMX_TEST_HOOKS) picks from where: 1 -- from the end of k_range_apply (the
			// uncontended winners': 99.6 % of them; the resolver's follow in front of the verdicts): measured on one GPU it buys
			// nothing, k_range_resolve takes 67 us instead of 35 under the atomics and k_reorder 20 instead of 6.6 -- the ordered
			// chain stretches by what the overlap hides; 2 -- from the end of k_reorder: ALL commits, beside the hashing of the
			// next round's triples (k_range_emit: ALU and streaming stores, not slowed by the atomics); 0 -- none.
			const char *ov = hook_env("KMX_RANGE_OVERLAP");
			const int overlap = ov ? atoi(ov) : KMX_RANGE_OVERLAP_DEFAULT;
			enum { EV_EMIT = 0, EV_VER, EV_BULK, EV_BULK_DONE };
			auto wait_all = [&](hipStream_t on, int which, bool self) {
				for (int q = 0; q < P; q++)
					if (q != d || self) hip_ok(hipStreamWaitEvent(on, R[(size_t)q].ev[which], 0), "wait");
			};
			bool bulk_ahead = false;                                          // the side stream holds a commit launch the next verdicts depend on
			for (u64 b = 0; b < n_blocks; b++) {
				const u64 n_in_block = std::min<u64>(blk, n_km - b * blk);
				if (!err && n_in_block < blk && b > 0) {                     // quirk Q1 (kmodel.hpp:520-527), on the rank that holds the list
					const int row = (int)((n_in_block - 1) / KMX_BUCKET);
					if (row + 1 < nb) note(kmx_ring_stale_dup_dev_impl(m, row + 1));
				}
				for (int t = 0; t < nb; t++) {
					kmx_ring_list lists[KMX_MAX_NB];
					int n_lists = 0;
					if (t == 0)
						for (int i = d; i < nb; i += P) {
							kmx_ring_list &l = lists[n_lists++];
							memset(&l, 0, sizeof l);
							l.list = i; l.n_host = list_len(b, i);
							l.src_kmers = r.d_rk + pos * W; l.src_counts = r.d_rc + pos;
							pos += (u64)l.n_host;
						}
					if (!err) { hip_ok(hipSetDevice(m->device), "hipSetDevice"); note(range_list_emit(m, t, lists, n_lists)); }   // 1. this round's triples behind the commits of the round before -> the owners' inboxes
					if (!err) hip_ok(hipEventRecord(r.ev[EV_EMIT], st), "event");
					bar.wait();
					if (!err) {                                                                     // 2. commits applied, one verdict byte per triple -> the senders' boxes
						wait_all(st, EV_EMIT, false);
						if (bulk_ahead) hip_ok(hipStreamWaitEvent(st, r.ev[EV_BULK_DONE], 0), "wait");
						note(range_owner_round(m, t, m->range.in, bulk_ahead ? RANGE_LATE : RANGE_ALL));
						hip_ok(hipEventRecord(r.ev[EV_VER], st), "event");
					}
					bar.wait();
					auto commits_aside = [&] {                                                      // what the headers call the bulk, on the side stream
						if (!err) hip_ok(hipEventRecord(r.ev[EV_BULK], st), "event");
						bar.wait();
						if (!err) {
							wait_all(r.side, EV_BULK, true);
							kmxk::range_commit_apply(m->md, m->range.in, RANGE_BULK, r.side);
							hip_ok(hipEventRecord(r.ev[EV_BULK_DONE], r.side), "event");
						}
						bulk_ahead = true;
					};
					if (!err) { wait_all(st, EV_VER, false); note(range_list_apply(m, t, overlap == 1)); }   // 3. failures, winners: the bulk of the commits is in the owners' inboxes
					if (overlap == 1) commits_aside();
					if (!err) note(range_list_order(m, t, overlap == 2));                          //    ... the contended in list order, reorder
					if (overlap == 2) commits_aside();
				}
			}
			if (n_blocks) {                                                   // the last round's commits
				if (!err) { kmxk::range_seal(m->range.rd, m->range.plan, st); m->range.pending = false; hip_ok(hipEventRecord(r.ev[EV_EMIT], st), "event"); }
				bar.wait();
				if (!err) {
					wait_all(st, EV_EMIT, false);
					if (bulk_ahead) hip_ok(hipStreamWaitEvent(st, r.ev[EV_BULK_DONE], 0), "wait");
					kmxk::range_commit_apply(m->md, m->range.in, bulk_ahead ? RANGE_LATE : RANGE_ALL, st);
					hip_ok(hipGetLastError(), "commit");
				}
			}
		} else
		// ---- the rounds, ring: this rank attempts the lists whose array it owns.  The survivors of a list are written by k_ring_export
		// STRAIGHT INTO the message buffer of the rank that owns the next array (a peer mapping): exactly the survivors cross the
		// link -- no copy of a whole 3.1 MB buffer, no count on the host -- and ONE event per rank and round orders the hand-offs:
		// before round r a rank waits for round r - 1 of the ranks it imports from (their export is complete) and of the ranks
		// it exports to (they have read what it is about to overwrite).
		{
		u64 round_no = 0;
		for (u64 b = 0; b < n_blocks; b++) {
			const u64 n_in_block = std::min<u64>(blk, n_km - b * blk);
			if (!err && n_in_block < blk && b > 0) {                         // quirk Q1 (kmodel.hpp:520-527)
				const int row = (int)((n_in_block - 1) / KMX_BUCKET);
				if (row + 1 < nb) note(kmx_ring_stale_dup_dev_impl(m, row + 1));
			}
			for (int t = 0; t < nb; t++, round_no++) {
				kmx_ring_list lists[KMX_MAX_NB];                             // (fixed arrays: nothing in a body may throw between two barriers)
				bool peer[KMX_MAX_RANKS_ANY] = {false};                      // ranks whose last round this one depends on
				int n_lists = 0;
				for (int i = 0; i < nb; i++) {
					const int n_i = list_len(b, i), a = (i + t) % nb;
					if (n_i == 0 || own[(size_t)a] != d) continue;
					kmx_ring_list l;
					memset(&l, 0, sizeof l);
					l.list = i;
					const int next = own[(size_t)((a + 1) % nb)], prev = own[(size_t)((a + nb - 1) % nb)];
					l.dst_msg = t + 1 < nb ? R[(size_t)next].msg[(size_t)i * 2 + ((t + 1) & 1)] : nullptr;      // the next owner's buffer (possibly this rank's own)
					if (t + 1 < nb && next != d) peer[next] = true;
					if (t == 0) { l.n_host = n_i; l.src_kmers = r.d_rk + pos * W; l.src_counts = r.d_rc + pos; pos += (u64)n_i; }
					else { l.n_host = -1; l.src_msg = r.msg[(size_t)i * 2 + (t & 1)]; if (prev != d) peer[prev] = true; }
					lists[n_lists++] = l;
				}
				if (!err && round_no > 0)
					for (int q = 0; q < P; q++) if (peer[q]) hip_ok(hipStreamWaitEvent(st, R[(size_t)q].ev[(round_no - 1) & 1], 0), "wait");
				if (!err && n_lists) note(kmx_ring_round_dev_impl(m, t, lists, n_lists));
				if (!err) hip_ok(hipEventRecord(r.ev[round_no & 1], st), "event");
				bar.wait();                                                  // (an event is recorded again two rounds later: everybody's waits on it are enqueued by then)
			}
		}
		}
		// ---- merge: survivors to every rank, filters OR-ed (set_bit is an OR, kmodel.hpp:576-581), every array from its owner
		if (trace && d == 0) mark(d, "rounds enqueued");
		if (!err) note(kmx_shard_local_impl(m, &r.st, &r.rest_k, &r.rest_c));
		bar.wait();
		mark(d, "rounds done on every handle");
		if (d == 0) {
			for (int q = 0; q < P; q++) {
				const kmx_stats &s2 = R[(size_t)q].st;
				if (s2.reserved) overflowed = true;                              // a fixed-size region dropped words: the build is void
				rest_off[(size_t)q + 1] = rest_off[(size_t)q] + s2.rest_entries;
				totals.attempts += s2.attempts; totals.successes += s2.successes; totals.fast_commits += s2.fast_commits;
				totals.contended += s2.contended; totals.finisher_iters += s2.finisher_iters;
			}
			n_rest_all = rest_off[(size_t)P];
			totals.blocks = n_blocks; totals.rounds = n_blocks * (u64)nb;
		}
		bar.wait();
		if (!err) {
			bool ok = hip_ok(hipMalloc((void **)&r.d_allk, std::max<u64>(n_rest_all, 1) * W * 8), "hipMalloc") && hip_ok(hipMalloc((void **)&r.d_allc, std::max<u64>(n_rest_all, 1) * 4), "hipMalloc");
			for (int q = 0; ok && q < P; q++) {
				const u64 c = rest_off[(size_t)q + 1] - rest_off[(size_t)q];
				if (!c) continue;
				ok = hip_ok(hipMemcpyPeerAsync(r.d_allk + rest_off[(size_t)q] * W, m->device, R[(size_t)q].rest_k, hs[q]->device, c * W * 8, st), "survivor copy") &&
				     hip_ok(hipMemcpyPeerAsync(r.d_allc + rest_off[(size_t)q], m->device, R[(size_t)q].rest_c, hs[q]->device, c * 4, st), "survivor copy");
			}
			if (ok) hip_ok(hipStreamSynchronize(st), "survivor gather");
		}
		bar.wait();
		auto filter = [&](kmx_model *mm, int which, int idx, u32 **p, u64 *words) {
			void *vp = nullptr; uint64_t bytes = 0;
			kmx_dev_view_impl(mm, which, idx, &vp, &bytes);
			*p = (u32 *)vp; *words = bytes / 4;
		};
		if (!err && d == 0) {                                              // rank 0 ORs everybody's partial filters ...
			u64 wmax = 0;
			for (int f = 0; f < 2 * m->bf_num + 1; f++) { u32 *p; u64 w; filter(m, f < m->bf_num ? 0 : (f < 2 * m->bf_num ? 1 : 2), f % m->bf_num, &p, &w); wmax = std::max(wmax, w); }
			bool ok = hip_ok(hipMalloc((void **)&r.d_tmp, std::max<u64>(wmax, 1) * 4), "hipMalloc");
			for (int f = 0; ok && f < 2 * m->bf_num + 1; f++) {
				const int which = f < m->bf_num ? 0 : (f < 2 * m->bf_num ? 1 : 2), idx = which == 2 ? 0 : f % m->bf_num;
				u32 *mine; u64 w;
				filter(m, which, idx, &mine, &w);
				for (int q = 1; ok && q < P && w; q++) {
					u32 *theirs; u64 w2;
					filter(hs[q], which, idx, &theirs, &w2);
					ok = hip_ok(hipMemcpyPeerAsync(r.d_tmp, m->device, theirs, hs[q]->device, w * 4, st), "filter copy");
					kmxk::or_words(mine, r.d_tmp, w, st);
				}
			}
			if (ok) hip_ok(hipStreamSynchronize(st), "filter merge");
		}
		bar.wait();
		if (!err) {                                                        // ... and everybody takes the merged filters and the arrays it does not own
			if (d != 0)
				for (int f = 0; f < 2 * m->bf_num + 1; f++) {
					const int which = f < m->bf_num ? 0 : (f < 2 * m->bf_num ? 1 : 2), idx = which == 2 ? 0 : f % m->bf_num;
					u32 *mine, *theirs; u64 w, w2;
					filter(m, which, idx, &mine, &w);
					filter(hs[0], which, idx, &theirs, &w2);
					if (w) hip_ok(hipMemcpyPeerAsync(mine, m->device, theirs, hs[0]->device, w * 4, st), "filter copy");
				}
			if (by_range) {                                                   // every rank's cell range of every array
				for (int a = 0; a < nb; a++)
					for (int q = 0; q < P; q++) {
						const u64 lo = m->range.plan.cell_lo[q], hi = m->range.plan.cell_lo[q + 1];
						if (q != d && hi > lo) hip_ok(hipMemcpyPeerAsync(m->d_cells[a] + lo, m->device, hs[q]->d_cells[a] + lo, hs[q]->device, (hi - lo) * sizeof(cell_t), st), "array copy");
					}
			} else
			for (int a = 0; a < nb; a++)
				if (own[(size_t)a] != d) hip_ok(hipMemcpyPeerAsync(m->d_cells[a], m->device, hs[own[(size_t)a]]->d_cells[a], hs[own[(size_t)a]]->device, m->ncells * sizeof(cell_t), st), "array copy");
			hip_ok(hipStreamSynchronize(st), "merge");
		}
		bar.wait();
		if (!err) note(kmx_shard_complete_impl(m, (const uint64_t *)r.d_allk, r.d_allc, n_rest_all, &totals));
		hipStreamSynchronize(st);
		bar.wait();                                                         // nobody frees what a peer may still be reading
		mark(d, "merged and complete");
		hipFree(r.d_km); hipFree(r.d_cnt); hipFree(r.d_ck); hipFree(r.d_cc); hipFree(r.d_rk); hipFree(r.d_rc); hipFree(r.d_allk); hipFree(r.d_allc);
		hipFree(r.d_tmp); hipFree(r.x_recv); hipFree(r.x_ver); hipFree(r.x_back);
		for (u64 *p : r.msg) hipFree(p);
		for (hipEvent_t e : r.ev) if (e) hipEventDestroy(e);
		if (r.side) { hipStreamSynchronize(r.side); hipStreamDestroy(r.side); }
	};
	// (a body must reach every barrier whatever happens to it: an exception -- bad_alloc in one of its vectors -- is noted like
	// any other error and the walk goes on with empty steps; a thread that cannot be created leaves the build with an error
	// before anybody waits for it)
	std::vector<std::thread> th;
	std::atomic<int> started{1};
	try { th.reserve((size_t)P); } catch (...) { return fail(KMX_E_NOMEM, "out of memory"); }
	bool spawned = true;
	std::atomic<bool> go{false};
	std::atomic<bool> cancel{false};
	auto guarded_body = [&](int d) {
		while (!go.load(std::memory_order_acquire)) { if (cancel.load(std::memory_order_acquire)) return; std::this_thread::yield(); }
		body(d);
	};
	for (int d = 1; d < P && spawned; d++) {
		try { th.emplace_back(guarded_body, d); started++; } catch (...) { spawned = false; }
	}
	if (!spawned) { cancel.store(true, std::memory_order_release); for (auto &x : th) x.join(); return fail(KMX_E_NOMEM, "cannot start %d host threads", P); }
	go.store(true, std::memory_order_release);
	body(0);
	for (auto &x : th) x.join();
	if (err) { snprintf(g_err, sizeof g_err, "%s", err_msg.c_str()); for (int d = 0; d < P; d++) if (hs[d]->state == ST_BUILDING) hs[d]->state = ST_EMPTY; return err; }
	if (overflowed) {                                                   // (the merge ran on void arrays: harmless, everything is rebuilt)
		if (getenv("KMX_INIT_TRACE")) fprintf(stderr, "[kmx multi] a fixed-size region overflowed: the build is repeated through the inboxes\n");
		if (rccl_lock.owns_lock()) rccl_lock.unlock();
		return kmx_build_from_kmc_multi_ex_impl(hs, P, db_prefix, KMX_PARTITION_RANGE);
	}
	return KMX_OK;
}

static int kmx_create_on_impl(int device, int ci, int cs, int nh, int nb, kmx_model **out)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(KMX_E_NODEVICE, "no HIP device: libkmx has no CPU fallback");
	if (device < 0 || device >= n) return fail(KMX_E_ARG, "device %d: this process sees %d", device, n);
	int prev = -1;
	if (hipGetDevice(&prev) != hipSuccess) prev = -1;
	HIPCHK(hipSetDevice(device));
	const int rc = kmx_create_impl(ci, cs, nh, nb, out);
	if (prev >= 0 && prev != device) (void)hipSetDevice(prev);      // the calling thread keeps its current device
	return rc;
}

