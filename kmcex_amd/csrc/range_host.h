// range_host.h -- host side of the position-range partition (range_kernels.h has the kernels): buffers, the three steps of a round,
// and the transports that move the regions (the caller's collectives, the inboxes of kmx_build_from_kmc_multi_ex).  Included by
// kmx_api.hip (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------------ position-range partition (range_kernels.h)
// Every coupled array cut by position range over the ranks (the north star's partition).  Between a list rank
// and an owner lies one region of words with an in-band header; two transports move them: the CALLER's (kmx_range_*_dev below:
// the regions live here, kmcex_amd/dist.py moves them with all-to-alls) and the MAILBOX (kmx_build_from_kmc_multi_ex: the
// regions live in the owners' inboxes, written through peer mappings, the rounds ordered by events -- no host wait in a round).
static int range_begin_common(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world, bool mailbox)
{
	if (world > KMX_MAX_RANKS) return fail(KMX_E_ARG, "the range partition takes up to %d ranks", KMX_MAX_RANKS);
	TRY(kmx_shard_begin_impl(m, k, n_bf, n_total, rank, world));   // whole-model sizes; a rank works on its cell range of every array
	if (m->km_byte_size * 8 > (1ULL << 36)) { m->state = ST_EMPTY; return fail(KMX_E_ARG, "the range partition addresses up to 2^36 positions per array"); }
	auto &R = m->range;
	const int nb = m->nb, nh = m->nh;
	const u64 key = ((u64)nb << 32) | ((u64)nh << 16) | ((u64)world << 1) | (mailbox ? 1u : 0u);
	if (R.alloc_key != key) {
		HIPCHK(hipStreamSynchronize(m->stream));
		free_range(m);
		const u64 held = (u64)((nb + world - 1) / world), slots = (u64)nb * KMX_BUCKET;
		R.rd.cap = 2 * held * KMX_BUCKET * (u64)nh;                  // a round's triples behind the previous round's commits
		R.rd.rt_bits = nh <= 8 ? 22 : 23;
		if (mailbox) {
			TRY(dalloc(&R.d_inbox, (u64)world * R.rd.cap, false, m->stream));
			TRY(dalloc(&R.d_in_hdr, (u64)KMX_MAX_RANKS * KMX_RANGE_HDR, true, m->stream));
			TRY(dalloc(&R.d_vbox, (u64)world * R.rd.cap, false, m->stream));
		} else {
			TRY(dalloc(&R.d_send, (u64)world * R.rd.cap, false, m->stream));
			TRY(dalloc(&R.d_hdr, (u64)KMX_MAX_RANKS * KMX_RANGE_HDR, true, m->stream));
			HIPCHK(hipHostMalloc((void **)&R.h_hdr, sizeof(u32) * KMX_MAX_RANKS * KMX_RANGE_HDR));
		}
		TRY(dalloc(&R.rd.ccnt, (u64)KMX_MAX_RANKS * KMX_CTR_STRIDE, true, m->stream));
		TRY(dalloc(&R.rd.tcnt, (u64)KMX_MAX_RANKS * KMX_CTR_STRIDE, true, m->stream));
		TRY(dalloc(&R.rd.tidx, slots * nh, false, m->stream));
		TRY(dalloc(&R.rd.contended, slots, false, m->stream));
		TRY(dalloc(&R.rd.n_contended, (u64)KMX_MAX_NB * KMX_CTR_STRIDE, true, m->stream));
		TRY(dalloc(&R.rd.rt_key, held << R.rd.rt_bits, false, m->stream));
		TRY(dalloc(&R.rd.rt_resv, held << R.rd.rt_bits, false, m->stream));
		TRY(dalloc(&R.rd.rt_mark, held << R.rd.rt_bits, false, m->stream));
		TRY(dalloc(&R.rd.rt_eidx, slots * nh, false, m->stream));
		TRY(dalloc(&R.rd.rt_um, slots, false, m->stream));
		TRY(dalloc(&R.d_oovf, (u64)KMX_MAX_NB, true, m->stream));
		TRY(dalloc(&R.d_opcnt, (u64)KMX_MAX_NB * KMX_CL_MAXBINS * KMX_CTR_STRIDE, true, m->stream));
		TRY(dalloc(&R.d_lver, slots * nh, false, m->stream));        // every triple of a round may come to one owner
		TRY(dalloc(&R.d_ovf, (u64)KMX_CTR_STRIDE, true, m->stream));
		R.cap_full = R.rd.cap;
		R.alloc_key = key;
	}
	R.mailbox = mailbox;
	R.inband = false;
	R.rd.cap = R.cap_full;
	R.rd.ovf = R.d_ovf;
	HIPCHK(hipMemsetAsync(R.d_ovf, 0, sizeof(int), m->stream));
	HIPCHK(hipMemsetAsync(R.d_oovf, 0, sizeof(int) * KMX_MAX_NB, m->stream));
	HIPCHK(hipMemsetAsync(R.rd.n_contended, 0, sizeof(int) * KMX_MAX_NB * KMX_CTR_STRIDE, m->stream));   // (k_range_resolve leaves them zero round by round)
	HIPCHK(hipMemsetAsync(R.d_opcnt, 0, sizeof(int) * (u64)KMX_MAX_NB * KMX_CL_MAXBINS * KMX_CTR_STRIDE, m->stream));
	HIPCHK(hipMemsetAsync(R.rd.ccnt, 0, sizeof(int) * KMX_MAX_RANKS * KMX_CTR_STRIDE, m->stream));         // (k_range_seal leaves them zero; an aborted build may not have)
	HIPCHK(hipMemsetAsync(R.rd.tcnt, 0, sizeof(int) * KMX_MAX_RANKS * KMX_CTR_STRIDE, m->stream));
	if (mailbox) HIPCHK(hipMemsetAsync(R.d_in_hdr, 0, sizeof(u32) * KMX_MAX_RANKS * KMX_RANGE_HDR, m->stream));
	else HIPCHK(hipMemsetAsync(R.d_hdr, 0, sizeof(u32) * KMX_MAX_RANKS * KMX_RANGE_HDR, m->stream));
	R.plan.rank = rank; R.plan.world = world;
	for (int q = 0; q <= world; q++) R.plan.cell_lo[q] = (u64)(((unsigned __int128)m->ncells * (unsigned)q) / (unsigned)world);
	for (int q = 0; q < KMX_MAX_RANKS; q++) {
		R.rd.out[q] = (!mailbox && q < world) ? R.d_send + (u64)q * R.rd.cap : nullptr;      // (mailbox: range_link points them at the owners' inboxes)
		R.rd.hdr_out[q] = (!mailbox && q < world) ? R.d_hdr + KMX_RANGE_HDR * q : nullptr;
		R.rd.vin[q] = nullptr;
		R.sent_tot[q] = 0;
	}
	R.obd = m->bd;                                                  // (kmx_begin carved it; the claim bins are the owner's here)
	R.obd.cl_cnt[0] = R.obd.cl_cnt[1] = R.d_opcnt;                  // (its detect reads the padded counters and reports per claim, in the verdict bytes)
	R.obd.cl_ovf = R.d_oovf;
	R.on = true;
	R.pending = false;
	return KMX_OK;
}
static int kmx_range_begin_impl(kmx_model *m, int k, const uint64_t n_bf[3], uint64_t n_total, int rank, int world)
{
	return range_begin_common(m, k, n_bf, n_total, rank, world, false);
}

static int range_check(kmx_model *m, int t)
{
	if (!m) return fail(KMX_E_ARG, "null model");
	if (m->state != ST_BUILDING || !m->ring || !m->range.on) return fail(KMX_E_STATE, "kmx_range_* before kmx_range_begin");
	if (t < 0 || t >= m->nb) return fail(KMX_E_ARG, "bad round %d", t);
	HIPCHK(hipSetDevice(m->device));
	return KMX_OK;
}

// ---- the three steps of a round, whatever moves the words (all enqueue only)
// step 1, list rank: (t == 0: the fresh buffers of the block this rank holds, i = rank, rank + world, ...) triples by owner rank
// behind the commits the last round left in front of the regions; the headers
static int range_list_emit(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists)
{
	auto &R = m->range;
	if (t == 0) {
		RingLists rl;
		memset(&rl, 0, sizeof rl);
		for (int e = 0; e < n_lists; e++) {
			const kmx_ring_list &l = lists[e];
			if (l.list < 0 || l.list >= m->nb || rl.e[l.list].active || l.list % R.plan.world != R.plan.rank) return fail(KMX_E_ARG, "list %d is not this rank's", l.list);
			if (l.n_host < 0 || l.n_host > (int)KMX_BUCKET || (l.n_host > 0 && (!l.src_kmers || !l.src_counts))) return fail(KMX_E_ARG, "list %d: bad source", l.list);
			RingList &r = rl.e[l.list];
			r.active = 1; r.n_host = l.n_host;
			r.src_kmers = (const u64 *)l.src_kmers; r.src_counts = (const u32 *)l.src_counts;
		}
		for (int i = 0; i < m->nb; i++) R.n0[i] = rl.e[i].active ? rl.e[i].n_host : 0;
		kmxk::ring_import(m->md, m->bd, m->pp, rl, m->d_stg_kmers, m->d_stg_counts, m->stream);
	}
	kmxk::range_emit(m->md, m->bd, R.rd, R.plan, t, m->pp, false, m->stream);
	R.pending = false;                                             // the regions are sealed: commits of the last round + these triples
	HIPCHK(hipGetLastError());
	return KMX_OK;
}
// step 2, owner: the commit words first, then one verdict byte per triple, to where its sender reads it
// (commits: RANGE_ALL, or RANGE_LATE when the bulk of them was applied ahead, on the side stream)
static int range_owner_round(kmx_model *m, int t, const RangeIn &in, int commits)
{
	const u64 late_bin = ((volatile u64 *)m->h_feedback)[2];      // (a launch-shape heuristic like run_round's: never changes the result)
	const bool small_late = m->dbg_small_detect >= 0 ? m->dbg_small_detect != 0 : late_bin <= 2048;
	kmxk::range_verdict(m->md, m->range.obd, m->range.d_opcnt, t, in, m->range.d_lver, commits, small_late, m->stream);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}
// step 3, list rank: verdicts -> failures and winners; the uncontended winners' commits (the bulk) are in front of the regions
// and their count in the headers when this launch ends
static int range_list_apply(kmx_model *m, int t, bool seal_bulk)
{
	kmxk::range_apply(m->md, m->bd, m->range.rd, m->range.plan, t, m->pp, seal_bulk, m->stream);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}
// ... then the contended in list order (their commits behind the bulk); reorder_buffer (:529-540); after the last round km_back
// and the rest table
static int range_list_order(kmx_model *m, int t, bool seal_all)
{
	auto &R = m->range;
	const int nb = m->nb, pp = m->pp;
	kmxk::range_resolve(m->md, m->bd, R.rd, R.plan, t, pp, seal_all, m->stream);
	R.pending = true;
	HIPCHK(hipGetLastError());
	m->pp ^= 1;
	m->rounds++;
	if (t == nb - 1) {
		// the lists never leave their rank: survivors -> rest table, then km_back ONCE for the block -- every k-mer of a held list
		// that is not a survivor was inserted in one of the rounds (kmodel.hpp:548-550), as in the single-GPU build
		const int held = R.plan.rank < nb ? (nb - 1 - R.plan.rank) / R.plan.world + 1 : 0;
		if (held) {
			TRY(ensure_rest_capacity(m, (u64)held * KMX_BUCKET + (u64)nb));
			kmxk::rest_append(m->md, m->bd, m->pp, R.plan.rank, held, m->d_rest_kmers, m->d_rest_counts, m->d_rest_n, m->d_stale_kmers, m->d_stale_counts, m->d_feedback, m->stream, R.plan.world);
			u64 n_mine = 0;
			int n_in_block = 0;                                        // (the held lists' lengths are what a block of this many k-mers gives them)
			for (int i = R.plan.rank; i < nb; i += R.plan.world) { n_mine += (u64)R.n0[i]; if (R.n0[i] > 0) n_in_block = i * (int)KMX_BUCKET + R.n0[i]; }
			if (n_mine) {
				if (m->kmb_deferred) TRY(kmback_reserve(m, n_mine));
				kmxk::kmback_emit(m->md, m->bd, m->bd.kmers, m->bd.surv, R.plan.rank, held, 0, m->pp, n_in_block, m->kmb, m->stream, R.plan.world);
			}
		}
		m->blocks++;
	}
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

// ---- the mailbox transport (kmx_build_from_kmc_multi_ex): handle d's regions ARE the owners' inboxes.  Every handle of `hs`
// has been through range_begin_common(..., mailbox); devices that differ get peer access to each other's memory.
static int range_link(kmx_model **hs, int P, int d)
{
	kmx_model *m = hs[d];
	auto &R = m->range;
	HIPCHK(hipSetDevice(m->device));
	for (int q = 0; q < P; q++) {
		if (!hs[q]->range.on || !hs[q]->range.mailbox || hs[q]->range.rd.cap != R.rd.cap) return fail(KMX_E_STATE, "handle %d is not part of this range-partitioned build", q);
		TRY(peer_access(m->device, hs[q]->device));
		R.rd.out[q] = hs[q]->range.d_inbox + (u64)d * R.rd.cap;          // sender d's region in owner q's inbox
		R.rd.hdr_out[q] = hs[q]->range.d_in_hdr + KMX_RANGE_HDR * d;
		R.rd.vin[q] = R.d_vbox + (u64)q * R.rd.cap;                      // owner q answers into this rank's box
		R.in.reg[q] = R.d_inbox + (u64)q * R.rd.cap;                     // ... and as an owner: sender q's region here,
		R.in.vout[q] = hs[q]->range.d_vbox + (u64)d * R.rd.cap;          // its verdicts into sender q's box
	}
	R.in.hdr = R.d_in_hdr;
	R.in.hdr_stride = KMX_RANGE_HDR;
	R.in.cap = 0;
	R.in.world = P;
	return KMX_OK;
}

// ---- the caller-moved transport (kmcex_amd/dist.py): the regions stay here, the caller reads the headers and ships the words
static int range_read_headers(kmx_model *m, uint64_t *counts)
{
	auto &R = m->range;
	if (R.inband) {
		const u64 stride = KMX_RANGE_HDR / 2 + R.capx;
		for (int q = 0; q < R.plan.world; q++) HIPCHK(hipMemcpyAsync(R.h_hdr + KMX_RANGE_HDR * q, R.d_send + (u64)q * stride, sizeof(u32) * KMX_RANGE_HDR, hipMemcpyDeviceToHost, m->stream));
	} else HIPCHK(hipMemcpyAsync(R.h_hdr, R.d_hdr, sizeof(u32) * KMX_MAX_RANKS * KMX_RANGE_HDR, hipMemcpyDeviceToHost, m->stream));
	HIPCHK(hipStreamSynchronize(m->stream));
	for (int q = 0; q < R.plan.world; q++) {
		counts[q] = (uint64_t)R.h_hdr[KMX_RANGE_HDR * q] + (uint64_t)R.h_hdr[KMX_RANGE_HDR * q + 1];
		counts[R.plan.world + q] = (uint64_t)R.h_hdr[KMX_RANGE_HDR * q];
		R.sent_tot[q] = counts[q];
	}
	return KMX_OK;
}
static int range_caller_moved(kmx_model *m)
{
	if (m->range.mailbox) return fail(KMX_E_STATE, "this build moves its words through the owners' inboxes");
	return KMX_OK;
}
static int kmx_range_emit_dev_impl(kmx_model *m, int t, const kmx_ring_list *lists, int n_lists, uint64_t *counts)
{
	TRY(range_check(m, t));
	TRY(range_caller_moved(m));
	auto &R = m->range;
	if ((!counts && !R.inband) || (n_lists && !lists)) return fail(KMX_E_ARG, "null argument");
	for (int q = 0; counts && q < 2 * R.plan.world; q++) counts[q] = 0;
	if (m->km_byte_size == 0) return KMX_OK;                       // divergence D2: no arrays to insert into
	TRY(range_list_emit(m, t, lists, n_lists));
	return counts ? range_read_headers(m, counts) : KMX_OK;         // (fixed-size messages carry their counts: no host wait)
}

static int kmx_range_buffers_impl(kmx_model *m, void **d_send, uint64_t *cap_words, uint64_t *cell_lo)
{
	if (!m || !m->range.on) return fail(KMX_E_STATE, "kmx_range_* before kmx_range_begin");
	if (d_send) *d_send = m->range.d_send;
	if (cap_words) *cap_words = m->range.rd.cap;
	if (cell_lo) for (int q = 0; q <= m->range.plan.world; q++) cell_lo[q] = m->range.plan.cell_lo[q];
	return KMX_OK;
}

// what came in: the regions of the `n_src` senders back to back, totals[s] words each with commits[s] commit words in front;
// the verdict bytes are laid out the same way (one per word; those of the commit words stay unwritten)
static int range_in_of(kmx_model *m, const uint64_t *d_words, const uint64_t *totals, const uint64_t *commits, int n_src, uint8_t *d_verdict, RangeIn &in)
{
	memset(&in, 0, sizeof in);
	if (n_src < 0 || n_src > KMX_MAX_RANKS) return fail(KMX_E_ARG, "%d regions", n_src);
	u64 off = 0;
	for (int s = 0; s < n_src; s++) {
		if (commits[s] > totals[s] || totals[s] >> 32) return fail(KMX_E_ARG, "bad counts of region %d", s);
		in.reg[s] = (const u64 *)d_words + off;
		in.vout[s] = d_verdict ? d_verdict + off : nullptr;
		in.nc[s] = (u32)commits[s]; in.nt[s] = (u32)(totals[s] - commits[s]);
		off += totals[s];
	}
	in.world = n_src;
	return KMX_OK;
}
// ... as fixed-size messages: region s = [header (KMX_RANGE_HDR u32) | capx words], the verdict bytes capx apart
static int range_in_inband(kmx_model *m, const uint64_t *d_recv, int n_src, uint8_t *d_verdict, RangeIn &in)
{
	memset(&in, 0, sizeof in);
	auto &R = m->range;
	if (!R.inband) return fail(KMX_E_STATE, "kmx_range_inband was not called on this build");
	if (n_src != R.plan.world) return fail(KMX_E_ARG, "%d regions for %d ranks", n_src, R.plan.world);
	const u64 stride = KMX_RANGE_HDR / 2 + R.capx;
	for (int s = 0; s < n_src; s++) {
		in.reg[s] = (const u64 *)d_recv + (u64)s * stride + KMX_RANGE_HDR / 2;
		in.vout[s] = d_verdict ? d_verdict + (u64)s * R.capx : nullptr;
	}
	in.hdr = (const u32 *)d_recv;
	in.hdr_stride = (u32)(2 * stride);
	in.cap = (u32)R.capx;
	in.world = n_src;
	return KMX_OK;
}
// Fixed-size messages for the caller-moved transport: every region becomes [header | capx words] -- capx = the mean of a round's
// fullest exchange + 25 % + 8192, far beyond what uniformly hashed positions ever deviate -- so that a round is two equal-split
// all-to-alls with NO count on the host.  A word that does not fit is dropped and the build marked void (kmx_shard_local reports
// it in kmx_stats.reserved): the caller repeats it with counted messages.  Call between kmx_range_begin and the first emit.
static int kmx_range_inband_impl(kmx_model *m, void **d_send, uint64_t *region_words, uint64_t *capx_words)
{
	TRY(range_check(m, 0));
	TRY(range_caller_moved(m));
	auto &R = m->range;
	const u64 held = (u64)((m->nb + R.plan.world - 1) / R.plan.world);
	const u64 mean0 = held * KMX_BUCKET * (u64)m->nh / (u64)R.plan.world;
	u64 capx = mean0 + mean0 / 4 + 8192;
	if (const char *e = hook_env("KMX_RANGE_CAPX")) capx = std::max<u64>(64, strtoull(e, nullptr, 10));      // (test hook: a capacity that overflows)
	capx = std::min(capx, R.cap_full);
	capx = (capx + 7) & ~u64(7);
	const u64 stride = KMX_RANGE_HDR / 2 + capx;
	if ((u64)R.plan.world * stride > (u64)R.plan.world * R.cap_full) return fail(KMX_E_STATE, "region buffer too small");     // (cannot happen: capx <= cap_full - header only when cap_full is tiny)
	R.inband = true;
	R.capx = capx;
	R.rd.cap = capx;
	HIPCHK(hipMemsetAsync(R.d_send, 0, (u64)R.plan.world * stride * 8, m->stream));       // (headers of regions nobody writes stay zero)
	for (int q = 0; q < R.plan.world; q++) {
		R.rd.hdr_out[q] = (u32 *)(R.d_send + (u64)q * stride);
		R.rd.out[q] = R.d_send + (u64)q * stride + KMX_RANGE_HDR / 2;
	}
	if (d_send) *d_send = R.d_send;
	if (region_words) *region_words = stride;
	if (capx_words) *capx_words = capx;
	return KMX_OK;
}
static int kmx_range_verdict_inband_dev_impl(kmx_model *m, int t, const uint64_t *d_recv, int n_src, uint8_t *d_verdict)
{
	TRY(range_check(m, t));
	TRY(range_caller_moved(m));
	if (!d_recv || !d_verdict) return fail(KMX_E_ARG, "null argument");
	if (m->km_byte_size == 0) return KMX_OK;
	RangeIn in;
	TRY(range_in_inband(m, d_recv, n_src, d_verdict, in));
	return range_owner_round(m, t, in, RANGE_ALL);
}
static int kmx_range_commit_inband_dev_impl(kmx_model *m, const uint64_t *d_recv, int n_src)
{
	TRY(range_check(m, 0));
	TRY(range_caller_moved(m));
	if (!d_recv) return fail(KMX_E_ARG, "null argument");
	if (m->km_byte_size == 0) return KMX_OK;
	RangeIn in;
	TRY(range_in_inband(m, d_recv, n_src, nullptr, in));
	kmxk::range_commit_apply(m->md, in, RANGE_ALL, m->stream);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}
static int kmx_range_verdict_dev_impl(kmx_model *m, int t, const uint64_t *d_words, const uint64_t *totals, const uint64_t *commits, int n_src, uint8_t *d_verdict)
{
	TRY(range_check(m, t));
	TRY(range_caller_moved(m));
	if (!totals || !commits) return fail(KMX_E_ARG, "null argument");
	u64 n = 0, ntr = 0;
	for (int s = 0; s < n_src && s < KMX_MAX_RANKS; s++) { n += totals[s]; ntr += totals[s] - commits[s]; }
	if (n && (!d_words || !d_verdict)) return fail(KMX_E_ARG, "null argument");
	if (ntr >> KMX_RANGE_QBITS) return fail(KMX_E_ARG, "a round's exchange holds up to 2^%d triples", KMX_RANGE_QBITS);   // (a claim tuple names its triple in that many bits)
	if (m->km_byte_size == 0 || !n) return KMX_OK;
	RangeIn in;
	TRY(range_in_of(m, d_words, totals, commits, n_src, d_verdict, in));
	return range_owner_round(m, t, in, RANGE_ALL);
}

// verdicts in the order the words left (regions back to back, in rank order)
static int kmx_range_resolve_dev_impl(kmx_model *m, int t, const uint8_t *d_verdict)
{
	TRY(range_check(m, t));
	TRY(range_caller_moved(m));
	auto &R = m->range;
	if (m->km_byte_size == 0) return KMX_OK;
	u64 off = 0;
	for (int q = 0; q < R.plan.world; q++) { R.rd.vin[q] = d_verdict ? d_verdict + off : nullptr; off += R.inband ? R.capx : R.sent_tot[q]; }
	if (off && !d_verdict) return fail(KMX_E_ARG, "null argument");
	TRY(range_list_apply(m, t));
	return range_list_order(m, t);
}

// ... and on the owner: the winners' tag / value bits of a last exchange (kmodel.hpp:611-618)
static int kmx_range_commit_dev_impl(kmx_model *m, const uint64_t *d_commits, uint64_t n)
{
	TRY(range_check(m, 0));
	TRY(range_caller_moved(m));
	if (n && !d_commits) return fail(KMX_E_ARG, "null argument");
	if (m->km_byte_size == 0 || !n) return KMX_OK;
	RangeIn in;
	const uint64_t tot[1] = {n};
	TRY(range_in_of(m, d_commits, tot, tot, 1, nullptr, in));
	kmxk::range_commit_apply(m->md, in, RANGE_ALL, m->stream);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}
// end of the build: what is still pending in front of the regions (the commits of the last round) for a last exchange
static int kmx_range_flush_dev_impl(kmx_model *m, uint64_t *counts)
{
	TRY(range_check(m, 0));
	TRY(range_caller_moved(m));
	auto &R = m->range;
	if (!counts && !R.inband) return fail(KMX_E_ARG, "null argument");
	for (int q = 0; counts && q < 2 * R.plan.world; q++) counts[q] = 0;
	if (m->km_byte_size == 0) return KMX_OK;
	if (!R.pending && !R.inband) return KMX_OK;                     // (fixed-size messages: the headers are sealed -- with zeros -- whatever is pending)
	R.pending = false;
	kmxk::range_seal(R.rd, R.plan, m->stream);
	HIPCHK(hipGetLastError());
	return counts ? range_read_headers(m, counts) : KMX_OK;
}

// device memory of one filter / array of this handle, for the collectives of the caller (which: as kmx_download; 3 = the
// cells of coupled array `index`, value and tag interleaved -- see device_common.h)
static int kmx_dev_view_impl(kmx_model *m, int which, int index, void **ptr, uint64_t *bytes)
{
	if (!m || !ptr || !bytes) return fail(KMX_E_ARG, "null argument");
	if (m->state == ST_EMPTY) return fail(KMX_E_STATE, "no arrays yet");
	if (which >= 0 && which <= 1 && (index < 0 || index >= m->bf_num)) return fail(KMX_E_ARG, "bad filter index");
	if (which == 3 && (index < 0 || index >= m->nb)) return fail(KMX_E_ARG, "bad array index");
	auto words = [](u64 nbytes) { return ((nbytes + 3) / 4) * 4; };
	switch (which) {
	case 0: *ptr = m->d_bf[index]; *bytes = words(m->byte_bf[index]); break;
	case 1: *ptr = m->d_bf_back[index]; *bytes = words(m->byte_bf_back[index]); break;
	case 2: *ptr = m->d_km_back; *bytes = words(m->byte_km_back); break;
	case 3: *ptr = m->d_cells[index]; *bytes = m->ncells * sizeof(cell_t); break;
	default: return fail(KMX_E_ARG, "bad selector");
	}
	return KMX_OK;
}

static int kmx_or_words_dev_impl(kmx_model *m, void *d_dst, const void *d_src, uint64_t n_words)
{
	if (!m || (n_words && (!d_dst || !d_src))) return fail(KMX_E_ARG, "null argument");
	HIPCHK(hipSetDevice(m->device));
	kmxk::or_words((u32 *)d_dst, (const u32 *)d_src, n_words, m->stream);
	HIPCHK(hipGetLastError());
	return KMX_OK;
}

