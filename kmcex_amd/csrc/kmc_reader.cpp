// kmc_reader.cpp -- see kmc_reader.h.  Format per SURVEY.md Appendix B.3 (kmc_file.cpp:132-292, :428-515).
#include "kmc_reader.h"
#include <algorithm>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace kmx {

namespace {
bool read_file(const std::string &path, std::vector<unsigned char> &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	rewind(f);
	out.resize(sz > 0 ? (size_t)sz : 0);
	size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
	fclose(f);
	return got == out.size();
}
bool pread_fully(int fd, void *dst, size_t bytes, uint64_t off)
{
	unsigned char *d = (unsigned char *)dst;
	while (bytes) {
		const ssize_t got = ::pread(fd, d, bytes, (off_t)off);
		if (got <= 0) return false;
		d += got; off += (uint64_t)got; bytes -= (size_t)got;
	}
	return true;
}
template <typename T> T rd(const unsigned char *p)
{
	T v;
	memcpy(&v, p, sizeof(T));
	return v;
}
}   // namespace

// the prefix file is read where it is needed: markers, the tail, the header -- and the LUT(s) straight into lut_ with parallel
// preads (a KMC2 database with 512 bins has 67 MB of them: through a buffer of the whole file and a copy that is 60 ms)
bool KmcListing::open(const std::string &prefix, bool load_lut)
{
	close();
	const int pfd = ::open((prefix + ".kmc_pre").c_str(), O_RDONLY);
	struct stat psb;
	auto pread_all = [&](void *dst, size_t bytes, uint64_t off) { return pread_fully(pfd, dst, bytes, off); };
	unsigned char head[4], tail[12];
	if (pfd < 0 || fstat(pfd, &psb) != 0 || (size_t)psb.st_size < 4 + 4 + 12 || !pread_all(head, 4, 0) || !pread_all(tail, 12, (uint64_t)psb.st_size - 12) ||
	    memcmp(head, "KMCP", 4) || memcmp(tail + 8, "KMCP", 4)) {
		if (pfd >= 0) ::close(pfd);
		err_ = "cannot open " + prefix + ".kmc_pre (missing or no KMCP markers)";
		return false;
	}
	pre_fd_ = pfd;                                                  // (closed by close(): the LUT may be read later, into the caller's memory)
	const size_t end = (size_t)psb.st_size;
	// the last 8 + header_offset bytes: header, offset field, marker (header_offset is one byte: at most 255)
	std::vector<unsigned char> pre_tail(std::min<size_t>(end, 8 + 255 + 4));
	if (!pread_all(pre_tail.data(), pre_tail.size(), end - pre_tail.size())) { err_ = "cannot read " + prefix + ".kmc_pre"; return false; }
	struct TailView {                                               // pre[i] for the bytes of the file that were read
		const std::vector<unsigned char> &t; size_t end;
		const unsigned char &operator[](size_t i) const { return t[i - (end - t.size())]; }
	} pre{pre_tail, end};
	version_ = rd<uint32_t>(&pre[end - 12]);                        // kmc_file.cpp:184-187
	const uint32_t header_offset = pre[end - 8];                    // low byte only (:193, :243)
	if (version_ != 0 && version_ != 0x200) { err_ = "unsupported KMC database version"; return false; }
	if ((size_t)header_offset + 8 > end - 4) { err_ = "corrupt KMC header offset"; return false; }
	const unsigned char *h = &pre[end - 8 - header_offset];
	size_t lut_bytes;
	if (version_ == 0x200) {                                        // KMC2 header (:197-208)
		if (header_offset < 37) { err_ = "corrupt KMC2 header"; return false; }
		k_ = rd<uint32_t>(h);
		mode_ = rd<uint32_t>(h + 4);
		counter_size_ = rd<uint32_t>(h + 8);
		p_ = rd<uint32_t>(h + 12);
		const uint32_t sig_len = rd<uint32_t>(h + 16);
		min_count_ = rd<uint32_t>(h + 20);
		max_count_ = rd<uint32_t>(h + 24);
		total_ = rd<uint64_t>(h + 28);
		const size_t sig_map_bytes = ((size_t(1) << (2 * sig_len)) + 1) * 4;
		const size_t inner = end - 8 - 4;                           // without both markers and the offset field
		if (inner < sig_map_bytes + header_offset + 8) { err_ = "corrupt KMC2 prefix file"; return false; }
		lut_bytes = inner - (sig_map_bytes + header_offset + 8);    // (:212); one guard u64 follows the LUTs
	} else {                                                        // KMC1 header (:253-279)
		if (header_offset < 40) { err_ = "corrupt KMC1 header"; return false; }
		k_ = rd<uint32_t>(h);
		mode_ = rd<uint32_t>(h + 4);
		counter_size_ = rd<uint32_t>(h + 8);
		p_ = rd<uint32_t>(h + 12);
		min_count_ = rd<uint32_t>(h + 16);
		max_count_ = rd<uint32_t>(h + 20);
		total_ = rd<uint64_t>(h + 24);
		max_count_ += rd<uint64_t>(h + 32) & 0xFFFFFFFF00000000ULL;
		lut_bytes = end - 8 - 4 - header_offset;
	}
	if (mode_ != 0) { err_ = "KMC databases with quality-aware (float) counters are not supported"; return false; }
	if (k_ == 0 || k_ > 64 || p_ > 15 || p_ >= k_ || (k_ - p_) % 4 || counter_size_ == 0 || counter_size_ > 4) {
		err_ = "unsupported KMC parameters (k, prefix length or counter size)";
		return false;
	}
	const size_t n_lut = lut_bytes / 8;
	if (n_lut == 0 || 4 + n_lut * 8 > end) { err_ = "corrupt KMC LUT"; return false; }
	n_lut_ = n_lut;
	if (load_lut && !lut_loaded()) { err_ = "cannot read the LUT of " + prefix + ".kmc_pre"; return false; }
	prefix_mask_ = (uint64_t(1) << (2 * p_)) - 1;
	suf_bytes_ = (k_ - p_) / 4;
	rec_bytes_ = suf_bytes_ + counter_size_;
	const int fd = ::open((prefix + ".kmc_suf").c_str(), O_RDONLY);
	struct stat sb;
	if (fd < 0 || fstat(fd, &sb) != 0 || sb.st_size < 8) {
		if (fd >= 0) ::close(fd);
		err_ = "cannot open " + prefix + ".kmc_suf";
		close();
		return false;
	}
	fd_ = fd;
	file_len_ = (size_t)sb.st_size;
	char marker[4];
	if (!read_at(0, marker, 4) || memcmp(marker, "KMCS", 4)) { err_ = prefix + ".kmc_suf has no KMCS marker"; close(); return false; }
	// a truncated file ends the listing early, like the reference's EOF
	const uint64_t recs_in_file = (file_len_ - 4 >= 4 ? file_len_ - 8 : 0) / rec_bytes_;
	avail_ = std::min<uint64_t>(total_, recs_in_file);
	restart();
	return true;
}

// n_lut_ LUT entries + the sentinel (= total) -> dst, with parallel preads
bool KmcListing::read_lut(uint64_t *dst) const
{
	if (pre_fd_ < 0 || !n_lut_) return false;
	const size_t bytes = n_lut_ * 8;
	const int T = (int)std::max<size_t>(1, std::min<size_t>(8, bytes / (4u << 20) + 1));
	std::atomic<bool> bad{false};
	auto part = [&](int t) {
		const size_t per = ((bytes + T - 1) / T + 4095) & ~size_t(4095), lo = (size_t)t * per, hi = std::min(bytes, lo + per);
		if (lo < hi && !pread_fully(pre_fd_, (unsigned char *)dst + lo, hi - lo, 4 + lo)) bad = true;
	};
	std::vector<std::thread> th;
	for (int t = 1; t < T; t++) th.emplace_back(part, t);
	part(0);
	for (auto &x : th) x.join();
	dst[n_lut_] = total_;
	return !bad;
}
// the host copy of the LUT (the host decoder, the tests): loaded on first use
bool KmcListing::lut_loaded() const
{
	if (lut_.size() == n_lut_ + 1) return true;
	lut_.resize(n_lut_ + 1);
	if (read_lut(lut_.data())) return true;
	lut_.clear();
	io_failed_.store(true, std::memory_order_relaxed);
	return false;
}

void KmcListing::close()
{
	if (fd_ >= 0) ::close(fd_);
	if (pre_fd_ >= 0) ::close(pre_fd_);
	pre_fd_ = -1;
	n_lut_ = 0;
	fd_ = -1;
	file_len_ = 0;
	lut_.clear();
	stage_.clear();
}

bool KmcListing::read_at(uint64_t off, void *dst, size_t bytes) const
{
	char *p = (char *)dst;
	while (bytes) {
		const ssize_t got = pread(fd_, p, bytes, (off_t)off);
		if (got <= 0) { io_failed_.store(true, std::memory_order_relaxed); return false; }
		p += got; off += (uint64_t)got; bytes -= (size_t)got;
	}
	return true;
}

void KmcListing::restart() { rec_ = 0; }

// Decode records [rec0, rec0 + n_recs) (fixed size, so any range can be decoded on its own): the prefix of a record is
// the index of the LUT entry that contains it (kmc_file.cpp:439-449; "& prefix_mask" folds KMC2's per-bin LUTs).
// Counts outside [min_count, max_count] are skipped exactly like ReadNextKmer does (kmc_file.cpp:513).
size_t KmcListing::decode_range(const unsigned char *recs, uint64_t rec0, size_t n_recs, uint64_t *kmers, uint32_t *counts) const
{
	const int W = words();
	if (!lut_loaded()) return 0;                                    // (io_failed() is set)
	const size_t n_lut = lut_.size() - 1;
	// last LUT entry <= rec0 that is followed by a larger one
	size_t idx = (size_t)(std::upper_bound(lut_.begin(), lut_.begin() + n_lut, rec0) - lut_.begin());
	idx = idx ? idx - 1 : 0;
	size_t out = 0;
	// fast path (one-word k-mers, suffix of at most 8 bytes): one unaligned 8-byte load per field instead of byte loops.
	// It reads 8 bytes from the start of a field, so the last few records of the range take the byte loops.
	const size_t margin = 8 / rec_bytes_ + 2;
	const size_t n_fast = (W == 1 && suf_bytes_ >= 1 && suf_bytes_ <= 8 && counter_size_ <= 4 && n_recs > margin) ? n_recs - margin : 0;
	const uint32_t suf_shift = 64 - 8 * suf_bytes_, cnt_mask = counter_size_ == 4 ? 0xFFFFFFFFu : ((1u << (8 * counter_size_)) - 1);
	size_t j = 0;
	for (; j < n_fast; j++) {
		const uint64_t rec = rec0 + j;
		while (idx + 1 < n_lut && lut_[idx + 1] <= rec) idx++;
		const unsigned char *r = recs + j * rec_bytes_;
		uint64_t s8, c8;
		memcpy(&s8, r, 8);
		memcpy(&c8, r + suf_bytes_, 8);
		const uint32_t c = (uint32_t)c8 & cnt_mask;                        // little-endian counter
		if (c < min_count_ || c > max_count_) continue;
		const uint64_t suffix = __builtin_bswap64(s8) >> suf_shift;          // big-endian suffix
		kmers[out] = suf_bytes_ == 8 ? suffix : (((uint64_t)(idx & prefix_mask_) << (8 * suf_bytes_)) | suffix);
		counts[out++] = c;
	}
	for (; j < n_recs; j++) {
		const uint64_t rec = rec0 + j;
		while (idx + 1 < n_lut && lut_[idx + 1] <= rec) idx++;
		const unsigned char *r = recs + j * rec_bytes_;
		uint32_t c = 0;
		for (uint32_t b = 0; b < counter_size_; b++) c |= (uint32_t)r[suf_bytes_ + b] << (8 * b);
		if (c < min_count_ || c > max_count_) continue;
		if (W == 1) {
			uint64_t v = idx & prefix_mask_;
			for (uint32_t b = 0; b < suf_bytes_; b++) v = (v << 8) | r[b];
			kmers[out] = v;
		} else {
			unsigned __int128 v = idx & prefix_mask_;
			for (uint32_t b = 0; b < suf_bytes_; b++) v = (v << 8) | r[b];
			kmers[2 * out] = (uint64_t)(v >> 64);
			kmers[2 * out + 1] = (uint64_t)v;
		}
		counts[out++] = c;
	}
	return out;
}

size_t KmcListing::next_batch(uint64_t *kmers, uint32_t *counts, size_t max_n)
{
	if (fd_ < 0 || rec_ >= avail_ || !max_n) return 0;
	if (!lut_loaded()) return 0;                                    // (before the decode threads start: the load is not theirs to race for)
	const int W = words();
	const size_t got = (size_t)std::min<uint64_t>(max_n, avail_ - rec_);
	stage_.resize(got * rec_bytes_ + 16);                            // the decoders' 8-byte loads may reach past the last field
	copy_records(rec_, got, stage_.data());
	const unsigned char *recs = stage_.data();
	const uint64_t rec0 = rec_;
	rec_ += got;
	int T = threads_;
	if ((size_t)T > got / 65536 + 1) T = (int)(got / 65536 + 1);
	if (T <= 1) return decode_range(recs, rec0, got, kmers, counts);
	// each thread decodes a contiguous slice into the matching slice of the output; slices are closed up afterwards
	// if the count filter dropped anything (rare: KMC already applied -ci/-cs)
	std::vector<size_t> kept(T, 0);
	std::vector<std::thread> th;
	const size_t per = (got + T - 1) / T;
	for (int t = 0; t < T; t++)
		th.emplace_back([&, t] {
			const size_t lo = (size_t)t * per, hi = std::min(got, lo + per);
			if (lo < hi) kept[t] = decode_range(recs + lo * rec_bytes_, rec0 + lo, hi - lo, kmers + lo * W, counts + lo);
		});
	for (auto &x : th) x.join();
	size_t out = 0;
	for (int t = 0; t < T; t++) {
		const size_t lo = (size_t)t * per;
		if (kept[t] && out != lo) {
			memmove(kmers + out * W, kmers + lo * W, kept[t] * W * sizeof(uint64_t));
			memmove(counts + out, counts + lo, kept[t] * sizeof(uint32_t));
		}
		out += kept[t];
	}
	return out;
}

void KmcListing::copy_records(uint64_t rec0, uint64_t n, unsigned char *dst) const
{
	if (fd_ < 0 || rec0 >= avail_ || !n) return;
	n = std::min<uint64_t>(n, avail_ - rec0);
	const uint64_t src = 4 + rec0 * rec_bytes_;
	const uint64_t bytes = n * rec_bytes_;
	const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads_, bytes / (1u << 20) + 1));
	if (T == 1) { read_at(src, dst, bytes); return; }
	std::vector<std::thread> th;
	const uint64_t per = ((bytes + T - 1) / T + 4095) & ~uint64_t(4095);
	for (int t = 0; t < T; t++)
		th.emplace_back([=] {
			const uint64_t lo = (uint64_t)t * per, hi = std::min(bytes, lo + per);
			if (lo < hi) read_at(src + lo, dst + lo, hi - lo);
		});
	for (auto &x : th) x.join();
}

void KmcListing::count_classes(uint32_t ci, uint32_t cs, int bf_num, uint64_t n_bf[3], uint64_t *out_of_range, uint64_t *not_listed, int threads) const
{
	n_bf[0] = n_bf[1] = n_bf[2] = 0;
	*out_of_range = 0;
	if (not_listed) *not_listed = 0;
	if (fd_ < 0) return;
	const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(threads > 0 ? threads : threads_), avail_ / 65536 + 1));
	std::vector<uint64_t> acc((size_t)T * 8, 0);
	auto work = [&](int t) {
		const uint64_t per = (avail_ + T - 1) / T, lo = (uint64_t)t * per, hi = std::min(avail_, lo + per);
		uint64_t a[5] = {0, 0, 0, 0, 0};
		const uint32_t cnt_mask = counter_size_ == 4 ? 0xFFFFFFFFu : ((1u << (8 * counter_size_)) - 1);
		const uint64_t chunk = std::max<uint64_t>(1, (256u << 10) / rec_bytes_);   // records per read: ~256 KB, so that the scan finds in this core's L2 what pread just copied there
		std::vector<unsigned char> buf(chunk * rec_bytes_ + 8);
		for (uint64_t r0 = lo; r0 < hi; r0 += chunk) {
			const uint64_t nr = std::min(chunk, hi - r0);
			if (!read_at(4 + r0 * rec_bytes_, buf.data(), nr * rec_bytes_)) break;
			for (uint64_t r = 0; r < nr; r++) {                              // one unaligned load per record (the buffer has 8 spare bytes)
				uint32_t c;
				memcpy(&c, buf.data() + r * rec_bytes_ + suf_bytes_, 4);
				c &= cnt_mask;
				if (c < min_count_ || c > max_count_) { a[4]++; continue; }   // not listed
				if (c < ci || c > cs) a[3]++;
				else if (c < ci + (uint32_t)bf_num) a[c - ci]++;
			}
		}
		for (int q = 0; q < 5; q++) acc[(size_t)t * 8 + q] = a[q];
	};
	if (T == 1) work(0);
	else {
		std::vector<std::thread> th;
		for (int t = 0; t < T; t++) th.emplace_back(work, t);
		for (auto &x : th) x.join();
	}
	for (int t = 0; t < T; t++) {
		for (int q = 0; q < 3; q++) n_bf[q] += acc[(size_t)t * 8 + q];
		*out_of_range += acc[(size_t)t * 8 + 3];
		if (not_listed) *not_listed += acc[(size_t)t * 8 + 4];
	}
}

}   // namespace kmx
